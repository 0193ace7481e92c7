"""tools/valu_busy_pmc.sh's passes -> per opcode and occupancy: every collected SQ counter per VALU instruction next to the calibrated issue cycles
(profiles/valu_calib.json), and the counter (with its unit) whose per-instruction value follows that cost.  usage: valu_busy_pmc.py <dir> <valu_calib.json>"""
import csv, glob, json, os, sys, collections
root, calib = sys.argv[1], json.load(open(sys.argv[2]))["cycles"]
CUS = 256
rows = collections.defaultdict(lambda: collections.defaultdict(dict))   # pass -> (kernel, grid, dispatch) -> counter -> value
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    p = os.path.relpath(f, root).split(os.sep)[0]
    for r in csv.DictReader(open(f)):
        rows[p][(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Dispatch_Id"]))][r["Counter_Name"]] = float(r["Counter_Value"])
per = collections.defaultdict(dict)   # (opcode, waves per SIMD) -> counter -> value per VALU instruction (chip-wide sums divided alike)
for p, ks in rows.items():
    for (name, grid, disp), c in ks.items():
        insts = c.get("SQ_INSTS_VALU", 0.0)
        if insts / (grid / 64.0) < 4000 * 64 * 0.9:      # warm-up launches
            continue
        wps = grid // (CUS * 256)
        for k, v in c.items():
            if k != "SQ_INSTS_VALU":
                per[(name[2:], wps)][k] = v / insts
counters = sorted({k for d in per.values() for k in d})
print("per VALU instruction (wave64); calibrated = issue cycles from GRBM_GUI_ACTIVE (profiles/valu_calib.json)")
print("%-16s %3s %10s " % ("opcode", "w", "calibrated") + " ".join("%22s" % k[-22:] for k in counters))
fit = collections.defaultdict(list)
for (op, w) in sorted(per, key=lambda k: (calib.get(k[0], {}).get("w4", 0), k[0], k[1])):
    if op not in calib or w not in (1, 4, 8):
        continue
    cal = calib[op]["w%d" % w]
    print("%-16s %3d %10.2f " % (op, w, cal) + " ".join("%22.4f" % per[(op, w)].get(k, float("nan")) for k in counters))
    if w == 4:
        for k in counters:
            if k in per[(op, w)]:
                fit[k].append((cal, per[(op, w)][k]))
print("\nat 4 waves per SIMD: counter per instruction = a x calibrated cycles (least squares through the origin), worst relative deviation over the opcodes")
for k, pts in fit.items():
    a = sum(x * y for x, y in pts) / max(1e-30, sum(x * x for x, y in pts))
    dev = max(abs(y - a * x) / max(1e-30, a * x) for x, y in pts) if a > 0 else float("nan")
    print("   %-28s a = %.5f   worst deviation %.1f %%   (%d opcodes)" % (k, a, 100 * dev, len(pts)))
