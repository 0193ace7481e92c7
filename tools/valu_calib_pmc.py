"""Issue cycles per wave64 VALU instruction from COUNTED cycles: `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES -- tools/valu_calib`.
cycles per instruction = (GRBM_GUI_ACTIVE / 8 XCDs) / (instructions per wave x waves per SIMD): free of the clock the chip held in each kernel
(the time ratios tools/valu_calib prints are not: a v_fma_f32 stream draws more power and runs at a lower clock than a v_add_f32 stream).
usage: python tools/valu_calib_pmc.py <rocprofv3 output dir> [out.json]   -> profiles/valu_calib.json"""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
ITERS, PER_ITER, CUS = 4000, 64, 256
rows = collections.defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
best = {}
for (disp, name, grid), c in rows.items():
    wps = grid // (CUS * 256)
    insts_per_wave = c.get("SQ_INSTS_VALU", 0) / (grid / 64.0)
    if insts_per_wave < ITERS * PER_ITER * 0.9:      # the 50-iteration warm-up launches
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0 / (insts_per_wave * wps)
    k = (name[2:], wps)
    best[k] = min(best.get(k, 1e30), cyc)
CLASS = {"fma_f32": "FMA_F32", "mul_f32": "MUL_F32", "add_f32": "ADD_F32", "fma_f64": "FMA_F64", "mul_f64": "MUL_F64", "add_f64": "ADD_F64",
         "rcp_f32": "TRANS_F32", "sqrt_f32": "TRANS_F32", "rsq_f32": "TRANS_F32", "exp_f32": "TRANS_F32", "log_f32": "TRANS_F32", "rcp_f64": "TRANS_F64", "sqrt_f64": "TRANS_F64",
         "add_u32": "INT32", "and_b32": "INT32", "lshrrev_b32": "INT32", "mul_lo_u32": "INT32", "mul_hi_u32": "INT32", "mad_u64_u32": "INT64", "lshlrev_b64": "INT64",
         "cvt_f32_u32": "CVT", "cvt_f32_ubyte0": "CVT", "cvt_f64_f32": "CVT", "cvt_f32_f64": "CVT", "cvt_i32_f64": "CVT"}
names = sorted({n for n, _ in best}, key=lambda n: (CLASS.get(n, "~"), n))
out = {"note": "issue cycles per wave64 VALU instruction = GRBM_GUI_ACTIVE / 8 / (SQ_INSTS_VALU per wave x waves per SIMD), an independent stream of the one opcode on every SIMD "
               "(tools/valu_calib.hip under rocprofv3 --pmc; includes the loop's own s_add / s_cmp / s_cbranch, which issue beside the VALU)", "cycles": {}}
print("%-16s %-10s %8s %8s %8s   (cycles per instruction at 1 / 4 / 8 waves per SIMD)" % ("opcode", "class", "w1", "w4", "w8"))
for n in names:
    v = [best.get((n, w)) for w in (1, 4, 8)]
    # the cmp kernel counts two VALU instructions (v_cmp + v_nop) per asm statement: SQ_INSTS_VALU already counts both
    print("%-16s %-10s %8.2f %8.2f %8.2f" % (n, CLASS.get(n, ""), *v))
    out["cycles"][n] = {"class": CLASS.get(n, ""), "w1": round(v[0], 3), "w4": round(v[1], 3), "w8": round(v[2], 3)}
# class costs the weighting uses (at 4 waves per SIMD, the megakernels' occupancy): the mean of the class's opcodes; INT32 = the simple ones (add / and /
# shift dominate address and flag arithmetic; v_mul_lo/hi_u32 are listed separately)
def mean(keys): return sum(best[(k, 4)] for k in keys) / len(keys)
out["class_cost_w4"] = {
    "FMA_F32": mean(["fma_f32"]), "MUL_F32": mean(["mul_f32"]), "ADD_F32": mean(["add_f32"]),
    "FMA_F64": mean(["fma_f64"]), "MUL_F64": mean(["mul_f64"]), "ADD_F64": mean(["add_f64"]),
    "TRANS_F32": mean(["rcp_f32", "sqrt_f32", "rsq_f32", "exp_f32", "log_f32"]), "TRANS_F64": mean(["rcp_f64", "sqrt_f64"]),
    "INT32": mean(["add_u32", "and_b32", "lshrrev_b32"]), "INT32_MUL": mean(["mul_lo_u32", "mul_hi_u32"]), "INT64": mean(["mad_u64_u32", "lshlrev_b64"]),
    "CVT": mean(["cvt_f32_u32", "cvt_f32_ubyte0", "cvt_f64_f32", "cvt_f32_f64", "cvt_i32_f64"]),
    # everything the class counters do not cover: compares, selects, min / max, moves, DPP, the divide helpers
    "OTHER": mean(["max_f32", "mov_dpp", "cmp_f32", "div_scale_f32", "div_fmas_f32", "div_fixup_f32"]),
}
print(json.dumps(out["class_cost_w4"], indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
