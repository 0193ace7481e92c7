"""profiles/pmc_traffic.json from the PMC passes of tools/pmc_profile.sh.

usage: ROUND_TAG=r02 python tools/pmc_traffic.py <workload>=<pmc out dir> [...]
HBM bytes per launch of the dominant kernel (k_trace / k_trace_pool) = 2 x FETCH_SIZE (KB, the gfx950 correction of
MI355X_MICROARCH.md's HBM section) + WRITE_SIZE (KB), each from its own pass; the SQ ratios come from the sq1 pass."""
import csv, glob, json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
res = json.load(open(out_path)) if os.path.exists(out_path) else {}
TAG = os.environ.get("ROUND_TAG", "r02")
for arg in sys.argv[1:]:
    wl, d = arg.split("=", 1)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = max((k for k in acc if "k_trace" in k), key=lambda k: sum(acc[k].get("SQ_WAVE_CYCLES", [0])))
    m = {c: sum(v) / len(v) for c, v in acc[kern].items()}
    simds = 256 * 4
    rec = {
        "kernel": kern.strip(),
        "hbm_bytes_per_launch": 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024,
        "fetch_bytes_corrected_x2": 2 * m["FETCH_SIZE"] * 1024,
        "write_bytes": m["WRITE_SIZE"] * 1024,
        "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
        "tcc_hit_rate": m["TCC_HIT_sum"] / max(1.0, m["TCC_REQ_sum"]),
        "valu_lane_utilisation": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]),
        "valu_insts_per_launch": m["SQ_INSTS_VALU"],
        # a wave64 VALU instruction occupies its SIMD-32's issue for 2 cycles (MI355X_MICROARCH.md constants table; tools/valu_calib.hip
        # measures 0.5 wave-instructions per cycle per SIMD at saturation on the bench box).  Round 1 used 4 and got 1.08.
        "cycles_per_launch": m["GRBM_GUI_ACTIVE"] / 8.0,
        "valu_busy_fraction": 2.0 * m["SQ_INSTS_VALU"] / simds / (m["GRBM_GUI_ACTIVE"] / 8.0),
        "wave_wait_fraction": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        "waves_per_simd": m["SQ_WAVES"] / simds,
        "salu_insts_per_launch": m.get("SQ_INSTS_SALU"),
        "round": TAG,
        "source": "rocprofv3 --pmc passes of tools/pmc_profile.sh (FETCH_SIZE and WRITE_SIZE in separate passes), mean per launch; "
                  "valu_busy = 2 cycles x SQ_INSTS_VALU / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); profiles/%s_pmc_%s.txt" % (TAG, wl),
    }
    res[wl] = rec
    print(wl, json.dumps(rec, indent=1))
json.dump(res, open(out_path, "w"), indent=1)
