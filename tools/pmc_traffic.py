"""profiles/pmc_traffic.json from the PMC passes of tools/pmc_profile.sh.

usage: ROUND_TAG=r03 python tools/pmc_traffic.py <workload>=<pmc out dir> [...]

Per workload, for the dominant kernel (k_trace / k_trace_pool), mean per launch:
  * HBM bytes = 2 x FETCH_SIZE (KB; the gfx950 correction of MI355X_MICROARCH.md's HBM section) + WRITE_SIZE (KB), each from its own pass;
  * VALU instructions by class (SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 / _F64, TRANS_F32 / _F64, INT32, INT64, CVT; the rest = SQ_INSTS_VALU minus
    those) priced with the issue costs measured by tools/valu_calib.hip under rocprofv3 (profiles/valu_calib.json: 2 cycles per wave64 for
    v_fma / v_mul / v_add / v_sub_f32 and the plain 32-bit and / or / xor / add / sub / shift-right / mov, 4 for everything else, 8 / 16 for
    f32 / f64 transcendentals).  INT32 and the unclassified rest mix 2- and 4-cycle opcodes that no counter separates: they are priced
    with the static mix of the kernel's code (tools/static_mix.py), and the all-2 / all-4 bounds are kept next to the estimate.
    valu_weighted_busy_fraction = sum(count x cost) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the launch's cycles the SIMDs'
    vector issue ports were taken;
  * the build id of the library the passes ran on (RaylibAMD_BuildId): bench.py calls the record STALE when it is not the loaded library's."""
import csv, glob, json, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
out_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
res = json.load(open(out_path)) if os.path.exists(out_path) else {}
TAG = os.environ.get("ROUND_TAG", "r04")
SIMDS = 256 * 4
NOMINAL = {"FMA_F32": 2, "MUL_F32": 2, "ADD_F32": 2, "FMA_F64": 4, "MUL_F64": 4, "ADD_F64": 4, "TRANS_F32": 8, "TRANS_F64": 16, "CVT": 4, "INT64": 4}


def build_id():
    from raylib_amd import binding
    return binding.load().RaylibAMD_BuildId().decode()


def static_mix():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "static_mix.py")], capture_output=True, text=True).stdout
    return json.loads(out)


BUILD = build_id()
MIX = static_mix()
for arg in sys.argv[1:]:
    wl, d = arg.split("=", 1)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    kern = max((k for k in acc if "k_trace" in k), key=lambda k: sum(acc[k].get("SQ_WAVE_CYCLES", [0])))
    m = {c: sum(v) / len(v) for c, v in acc[kern].items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0
    # static mix of this instantiation: "void rl::k_trace<16, false, true, 2>" -> "k_traceILi16ELb0ELb1ELi2E"
    args = kern.strip().split("<", 1)[1].rstrip(">").split(",")
    mangled = ("k_trace_pool" if "k_trace_pool" in kern else "k_trace") + "I" + "".join(("Lb1" if a.strip() == "true" else "Lb0" if a.strip() == "false" else "Li" + a.strip()) + "E" for a in args)
    assert mangled in MIX, (mangled, list(MIX)[:4])
    mix = MIX[mangled]["mean_cost"]
    counts = {c: m.get("SQ_INSTS_VALU_" + c, 0.0) for c in ("ADD_F32", "MUL_F32", "FMA_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F32", "TRANS_F64", "CVT", "INT32", "INT64")}
    other = m["SQ_INSTS_VALU"] - sum(counts.values())
    counts["OTHER"] = other
    fixed = sum(counts[c] * NOMINAL[c] for c in NOMINAL)
    est = fixed + counts["INT32"] * mix.get("INT32", 3.0) + other * mix.get("OTHER", 3.5)
    lo = fixed + (counts["INT32"] + other) * 2.0
    hi = fixed + (counts["INT32"] + other) * 4.0
    rec = {
        "kernel": kern.strip(),
        "build_id": BUILD,
        "hbm_bytes_per_launch": 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024,
        "fetch_bytes_corrected_x2": 2 * m["FETCH_SIZE"] * 1024,
        "write_bytes": m["WRITE_SIZE"] * 1024,
        "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
        "tcc_hit_rate": m["TCC_HIT_sum"] / max(1.0, m["TCC_REQ_sum"]),
        "valu_lane_utilisation": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]),
        "valu_insts_per_launch": m["SQ_INSTS_VALU"],
        "valu_class_counts": counts,
        "valu_class_costs": dict(NOMINAL, INT32=mix.get("INT32", 3.0), OTHER=mix.get("OTHER", 3.5)),
        "valu_weighted_cycles_per_launch": est,
        "valu_weighted_cycles_bounds": [lo, hi],
        "cycles_per_launch": cycles,
        "profiled_clock_ghz": None,
        "valu_weighted_busy_fraction": est / SIMDS / cycles,
        "valu_weighted_busy_bounds": [lo / SIMDS / cycles, min(1.0, hi / SIMDS / cycles)],   # (the port cannot be busier than always)
        # The one hardware counter of the VALU port's TIME, and what it can and cannot see (round 4, tools/valu_busy_pmc.sh over the single-opcode streams of
        # tools/valu_calib: profiles/r04_valu_busy_counters.txt): SQ_ACTIVE_INST_VALU counts max(1, issue cycles / 4) per instruction -- 1 for a 2-cycle AND for a
        # 4-cycle opcode, 2 for the 8-cycle transcendentals, 4 for the 16-cycle f64 ones -- i.e. the port's time in units of 4 clocks with every 2-cycle instruction
        # charged 4.  x4 over the launch's SIMD cycles it is an UPPER bound (above 1 where the code is full of 2-cycle opcodes: that excess is itself the
        # hardware saying they exist); taking 2 clocks back for every instruction the hardware classes as ADD / MUL / FMA F32 and for the 2-cycle share of
        # INT32 / OTHER (static mix) gives a figure that agrees with the class-weighted one because it leans on the same model for that share.
        "valu_hw_active_inst_valu": m["SQ_ACTIVE_INST_VALU"],
        "valu_hw_active_x4_fraction": 4.0 * m["SQ_ACTIVE_INST_VALU"] / SIMDS / cycles,
        "valu_hw_refined_fraction": (4.0 * m["SQ_ACTIVE_INST_VALU"] - 2.0 * (counts["FMA_F32"] + counts["MUL_F32"] + counts["ADD_F32"])
                                     - 2.0 * (counts["INT32"] * (4.0 - mix.get("INT32", 3.0)) / 2.0 + other * (4.0 - mix.get("OTHER", 3.5)) / 2.0)) / SIMDS / cycles,
        # round 2's figure, kept for comparison: every VALU instruction charged the 2 cycles of a v_fma_f32
        "valu_busy_fraction": 2.0 * m["SQ_INSTS_VALU"] / SIMDS / cycles,
        "wave_wait_fraction": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
        # the vector memory path's request rate: TCP accesses (one per lane and load instruction, whatever its width) per clock and CU, against the 1.14 a kernel
        # of nothing but scattered loads reaches (tools/vmem_width_bench.hip: 0.88 clocks per lane-load)
        "tcp_accesses_per_launch": m.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "tcp_accesses_per_clock_and_cu": (m["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / cycles) if m.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
        "tcp_request_ceiling_per_clock_and_cu": 1.0 / 0.88,
        "vmem_rd_insts_per_launch": m.get("SQ_INSTS_VMEM_RD"), "lds_insts_per_launch": m.get("SQ_INSTS_LDS"),
        "waves_per_simd": m["SQ_WAVES"] / SIMDS,
        "salu_insts_per_launch": m.get("SQ_INSTS_SALU"),
        "round": TAG,
        "source": "rocprofv3 --pmc passes of tools/pmc_profile.sh (FETCH_SIZE and WRITE_SIZE in separate passes), mean per launch; VALU cycles = class counts x "
                  "issue costs (profiles/valu_calib.json; INT32 and the unclassified rest at the kernel's static mix, tools/static_mix.py) over 1024 SIMDs x "
                  "(GRBM_GUI_ACTIVE / 8 XCDs); profiles/%s_pmc_%s.txt" % (TAG, wl),
    }
    res[wl] = rec
    print(wl, json.dumps(rec, indent=1))
json.dump(res, open(out_path, "w"), indent=1)
