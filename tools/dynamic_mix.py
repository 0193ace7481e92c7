"""Dynamic VALU mix of a megakernel: the hardware's instruction counters by class, explained by the code that runs.

tools/static_mix.py prices the two instruction classes that mix 2- and 4-clock opcodes (INT32, and everything no class counter covers) with the kernel's STATIC
mix -- every opcode once, cold code like hot code: its CVT share is 2.9 % where the hardware counts 11.6 %.  This tool weights the code by how often it runs:

  1. both device translation units are compiled again with -gline-tables-only (same flags otherwise: the code is the library's), disassembled, and every
     instruction of the kernel is given its inline stack by llvm-symbolizer -- the kernel-level source line it belongs to and the functions inlined on the way;
  2. instructions are grouped into REGIONS of the kernel's loop: the phases between the kernel's own stamps (RL_STAMP / RL_PSTAMP: refill, traversal, shading,
     miss + fold), split where a loop inside a phase runs a different number of times (node step, triangle step, the leaf list's box loop / pick / triangle loop,
     the Beckmann sampler's Newton iteration); out-of-line functions (glibc's powf / tanf / acosf / atan2f ..., TexFetch) are charged to the regions that call them,
     once per call site;
  3. how often each region runs (wave level) is the unknown: the eleven class counters of the hardware, its VALU / SALU / LDS / vector-memory instruction totals
     are linear in those counts, and a non-negative least-squares fit finds them.  The fit's residual per counter says how well the regions explain the hardware's
     numbers (the gate: 2 %); the counts that a diagnostic build measures directly (wave-level node and triangle steps, trips) are printed next to the fitted ones;
  4. the dynamic mix = sum over regions of count x the region's opcode histogram: mean issue cost of the INT32 class and of the unclassified rest, the weighted
     VALU issue cycles of the launch, and a budget by region (issue cycles, share of the launch).

usage: python tools/dynamic_mix.py <workload>[,<workload>...]     (reads profiles/pmc_traffic.json, profiles/valu_classes.json; writes the `valu_dynamic_mix` record
       of each workload back into profiles/pmc_traffic.json and prints the tables)"""
import collections, json, os, re, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import static_mix as sm

LLVM = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "software-raytracing_amd")
WORK = os.path.join(ROOT, "tools", "_scratch", "dyn")
CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F32", "TRANS_F64", "INT32", "INT64", "CVT")


def hw_class(op, measured):
    """The hardware class counter an opcode is counted under: measured where tools/valu_class_pmc.sh ran the opcode (profiles/valu_classes.json), by family otherwise."""
    b = sm.base(op)
    if b in measured:
        return measured[b]
    if b.startswith(sm.TRANS):
        return "TRANS_F64" if b.endswith("f64") else "TRANS_F32"
    if b.startswith("v_cvt_"):
        return "CVT"
    if b in ("v_add_f32", "v_sub_f32", "v_subrev_f32"):
        return "ADD_F32"
    if b in ("v_mul_f32", "v_mul_legacy_f32"):
        return "MUL_F32"
    if b in ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32"):
        return "FMA_F32"
    if b in ("v_add_f64",):
        return "ADD_F64"
    if b in ("v_mul_f64",):
        return "MUL_F64"
    if b in ("v_fma_f64", "v_fmac_f64"):
        return "FMA_F64"
    if b in sm.INT64:
        return "INT64"
    if b in sm.INT32 or re.match(r"v_cmpx?_\w+_[ui](16|32)$", b):
        return "INT32"
    return "OTHER"


def measured_classes():
    """profiles/valu_classes.json (kernel of tools/valu_calib -> class shares) -> {opcode: class} for the opcodes whose stream landed (>= 90 %) in one class or in none."""
    path = os.path.join(ROOT, "profiles", "valu_classes.json")
    out = {}
    if not os.path.exists(path):
        return out
    alias = {"cndmask_b32": "v_cndmask_b32", "cndmask_vcc": "v_cndmask_b32", "cndmask_e64": "v_cndmask_b32", "cmp_only_f32": "v_cmp_lt_f32", "cmp_u32": "v_cmp_lt_u32",
             "mov_dpp": "v_mov_b32", "readlane": "v_readlane_b32", "readfirstlane": "v_readfirstlane_b32", "writelane": "v_writelane_b32", "mbcnt_lo": "v_mbcnt_lo_u32_b32",
             "mbcnt_hi": "v_mbcnt_hi_u32_b32", "bcnt_u32": "v_bcnt_u32_b32", "cmp_class_f32": "v_cmp_class_f32", "cmpx_f32": "v_cmpx_le_f32"}
    for name, shares in json.load(open(path)).items():
        op = alias.get(name, "v_" + name)
        if any(s in name for s in ("_sgpr", "_inline", "_literal", "_mod", "_e64", "_sdwa", "_x2_", "cmp_cndmask", "cmp_f32", "_v", "bpermute", "_lo", "_hi", "_f32_f32")) and name not in alias:
            continue
        best = max(shares.items(), key=lambda kv: kv[1]) if shares else (None, 0.0)
        if best[1] >= 0.9:
            out[op] = best[0]
        elif sum(shares.values()) <= 0.1:
            out[op] = "OTHER"
    # families the calibration holds one member of
    for fam, members in (("v_cmp_lt_f32", [c + t for c in ("v_cmp_lt_", "v_cmp_gt_", "v_cmp_le_", "v_cmp_ge_", "v_cmp_eq_", "v_cmp_neq_", "v_cmp_nlt_", "v_cmp_ngt_", "v_cmp_nge_", "v_cmp_nle_", "v_cmp_lg_", "v_cmp_u_", "v_cmp_o_", "v_cmp_ngt_") for t in ("f32",)]),
                         ("v_cmp_lt_u32", [c + t for c in ("v_cmp_lt_", "v_cmp_gt_", "v_cmp_le_", "v_cmp_ge_", "v_cmp_eq_", "v_cmp_ne_") for t in ("u32", "i32", "u16", "i16", "u64", "i64")]),
                         ("v_max_f32", ["v_min_f32"]), ("v_max3_f32", ["v_min3_f32", "v_med3_f32"]), ("v_maximum3_f32", ["v_minimum3_f32"])):
        if fam in out:
            for m in members:
                out.setdefault(m, out[fam])
    return out


def build_objects():
    """The two device translation units with line tables (the library's own flags otherwise) -> ELF code objects under tools/_scratch/dyn."""
    os.makedirs(WORK, exist_ok=True)
    flags = "-std=c++17 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize -fvisibility=hidden -DRAYLIB_EXPORTS=1 -I%s/../include -I%s/csrc --offload-arch=gfx950 --cuda-device-only -gline-tables-only" % (SRC, SRC)
    units = {"rl_render": "", "rl_render_pool": "-mllvm -amdgpu-sched-strategy=iterative-ilp -DRL_EXACT_FAST_RCP_SQRT=0"}
    srcs = [os.path.join(SRC, "csrc", f) for f in os.listdir(os.path.join(SRC, "csrc")) if f.endswith((".hip", ".h", ".inl"))] + [os.path.join(ROOT, "include", "raylib_amd_rng.h")]
    newest = max(os.path.getmtime(f) for f in srcs)
    out = {}
    for u, extra in units.items():
        elf = os.path.join(WORK, u + ".elf")
        if not os.path.exists(elf) or os.path.getmtime(elf) < newest:
            co = os.path.join(WORK, u + ".co")
            r = subprocess.run("/opt/rocm/bin/hipcc %s %s -c %s/csrc/%s.hip -o %s" % (flags, extra, SRC, u, co), shell=True, capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stderr[-3000:])
            cos = sm.code_objects(co)
            open(elf, "wb").write(cos[0])
        out[u] = elf
    return out


def disassemble(elf):
    """{function: [(address, opcode, text)]} and the function start addresses."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
    funcs, starts, cur = collections.OrderedDict(), {}, None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = m.group(2); funcs[cur] = []; starts[int(m.group(1), 16)] = cur
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
        if m and cur is not None:
            funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs, starts


def call_targets(insts, starts):
    """addresses of s_swappc_b64 -> callee (s_getpc_b64 sN / s_add_u32 sN, sN, literal / s_addc_u32 ... / s_swappc_b64 s[30:31], s[N:N+1])."""
    out = {}
    getpc, lit = {}, {}
    for addr, op, text in insts:
        if op == "s_getpc_b64":
            m = re.match(r"s\[(\d+):", text)
            if m:
                getpc[int(m.group(1))] = addr + 4; lit.pop(int(m.group(1)), None)
        elif op == "s_add_u32":
            m = re.match(r"s(\d+), s(\d+), (0x[0-9a-f]+|-?\d+)$", text)
            if m and m.group(1) == m.group(2) and int(m.group(1)) in getpc and int(m.group(1)) not in lit:
                v = int(m.group(3), 0)
                if v >= 1 << 31:
                    v -= 1 << 32
                lit[int(m.group(1))] = v
        elif op == "s_swappc_b64":
            m = re.search(r", s\[(\d+):", text)
            if m and int(m.group(1)) in lit:
                out[addr] = starts.get(getpc[int(m.group(1))] + lit[int(m.group(1))], None)
    return out


def symbolize(elf, addrs):
    """address -> [(function, line)] innermost first (llvm-symbolizer --inlines)."""
    p = subprocess.run([os.path.join(LLVM, "llvm-symbolizer"), "--obj=" + elf, "--inlines", "--functions=short"], input="\n".join("0x%x" % a for a in addrs) + "\n", capture_output=True, text=True)
    out, frames, lines = {}, [], p.stdout.split("\n")
    it = iter(addrs)
    i = 0
    while i < len(lines):
        if lines[i] == "":
            if frames:
                out[next(it)] = frames; frames = []
            i += 1
            continue
        fn = lines[i]; loc = lines[i + 1] if i + 1 < len(lines) else ""
        m = re.match(r"(.*):(\d+):\d+$", loc)
        name = fn.split("<")[0].split("::")[-1]
        if name == "TraverseLeafList" and "<true>" in fn:
            name = "TraverseLeafListAny"   # the occlusion query's copy of the walk (ANYHIT): its own regions -- it runs only for sun queries
        frames.append((name, int(m.group(2)) if m else 0, os.path.basename(m.group(1)) if m else "?"))
        i += 2
    return out


def source_markers():
    """Line numbers the region rules hang on: the kernels' phase stamps and the loops inside TraverseLeafList / BeckmannSample11."""
    src = open(os.path.join(SRC, "csrc", "rl_render.hip")).read().split("\n")
    mk = {"stamp": [], "pstamp": []}
    for i, l in enumerate(src, 1):
        s = l.strip()
        if re.match(r"RL_STAMP\(\d\);", s):
            mk["stamp"].append(i)
        if re.match(r"RL_PSTAMP\(\d\);", s):
            mk["pstamp"].append(i)
        if "bool TraverseLeafList(" in l:
            mk["ll"] = i
        if "ll" in mk and "ll_from" not in mk and s.startswith("uint32_t from = 0u;"):
            mk["ll_from"] = i
        if "ll_from" in mk and "ll_tri" not in mk and s.startswith("for (int i = 0; i < count; ++i) {"):
            mk["ll_tri"] = i
        if "ll_tri" in mk and "ll_inner" not in mk and s.startswith("const V3 p = o + t * d;"):
            mk["ll_inner"] = i
        if "ll_tri" in mk and "ll_tri_end" not in mk and i > mk["ll_tri"] and s.startswith("m = 0xffffffffu;"):
            mk["ll_tri_end"] = i
        if "void BeckmannSample11(" in l:
            mk["b11"] = i
        if "b11" in mk and "newton" not in mk and s.startswith("while (++it < "):
            mk["newton"] = i
        if "newton" in mk and "newton_end" not in mk and i > mk["newton"] and s.startswith("b -= value / derivative;"):
            mk["newton_end"] = i + 1
        m = re.search(r"bool (LeafStep8?)\(", l)
        if m:
            mk["_leaf"] = m.group(1)
        if mk.get("_leaf") and s.startswith("const V3 pp = o + t * d;") and mk["_leaf"] + "_inner" not in mk:
            mk[mk["_leaf"] + "_inner"] = (i, i + 9)
        if "RL_WSTEP(7);" in l:
            mk["pool_hits"] = i
        if "pool_hits" in mk and "pool_hits_end" not in mk and s.startswith("shadedEnd += 64u;"):
            mk["pool_hits_end"] = i
        if "if (nextSlot < (uint32_t)PP) {" in l and "pool_fetch" not in mk:
            mk["pool_fetch"] = i
        if "pool_fetch" in mk and "pool_fetch_end" not in mk and s.startswith("nextSlot += (uint32_t)__popcll(idle);"):
            mk["pool_fetch_end"] = i
        if "if (fin) {" in l and "pool_fin" not in mk and "pool_fetch_end" in mk:
            mk["pool_fin"] = i
        if "pool_fin" in mk and "pool_fin_end" not in mk and s.startswith("T.cur = IDLE;"):
            mk["pool_fin_end"] = i + 1
        if "__device__ __forceinline__ bool Scatter(" in l:
            mk["scatter"] = i; mk["arms"] = []
        if "scatter" in mk and "scatter_end" not in mk:
            m = re.match(r"(case (MAT_\w+)|default):", s)
            if m:
                mk["arms"].append((i, (m.group(2) or "MAT_MICROFACET").replace("MAT_", "").lower()))
            if i > mk["scatter"] and l.startswith("}"):
                mk["scatter_end"] = i
    return mk


def region_of(frames, pool, mk):
    names = [f for f, _, _ in frames]
    for f, line, _ in frames:
        if f == "BeckmannSample11" and mk.get("newton", 1 << 30) <= line <= mk.get("newton_end", 0):
            return "newton iteration"
        if f in ("TexFetch", "TexSample", "AlphaTestCandidate", "AlphaTestCandidateNI"):
            return "texel fetch"
    for f, line, _ in frames:
        if f == "Scatter" and mk.get("arms"):
            arm = [name for (l0, name) in mk["arms"] if l0 <= line]
            if arm:
                return "scatter: " + arm[-1]
    for f, line, _ in frames:
        if f in ("TraverseLeafList", "TraverseLeafListAny"):
            pre = "leaf list: " if f == "TraverseLeafList" else "sun query's leaf list: "
            if line < mk["ll_from"]:
                return pre + ("boxes" if line > mk["ll"] + 12 else "set-up")
            if mk["ll_tri"] <= line < mk["ll_tri_end"]:
                return pre + ("inside test" if line >= mk.get("ll_inner", 1 << 30) else "triangle step")
            return pre + "pick"
    if any(n.startswith("NodeStep") for n in names):
        return "node step"
    for f, line, _ in frames:
        if f.startswith("LeafStep"):
            lo, hi = mk.get(f + "_inner", (1 << 30, 0))
            return "triangle step: inside test" if lo <= line <= hi else "triangle step"
    if any(n.startswith("LeafStep") for n in names):
        return "triangle step"
    if any(n in ("Traverse4", "Traverse") for n in names):
        return "tree walk"
    kline = frames[-1][1]
    stamps = mk["pstamp"] if pool else mk["stamp"]
    phase = sum(1 for s in stamps if kline > s)
    if pool:
        if phase == 1:
            if mk.get("pool_fetch", 1 << 30) <= kline <= mk.get("pool_fetch_end", 0):
                return "fetch a ray"
            if mk.get("pool_fin", 1 << 30) <= kline <= mk.get("pool_fin_end", 0):
                return "traversal: a ray ends"
            return "traversal: turn"
        if phase == 2:
            if mk.get("pool_hits", 1 << 30) - 12 <= kline <= mk.get("pool_hits_end", 0):
                return "shade: a round of hits"
            return "shade: misses + hand-back"
        return ("refill", "traversal: turn", "shade", "epilogue")[min(phase, 3)]
    return ("refill", "traverse: glue", "shade", "miss + fold", "epilogue")[min(phase, 4)]


def analyse(kernel_mangled, elf, measured):
    funcs, starts = disassemble(elf)
    kname = [k for k in funcs if kernel_mangled in k]
    assert kname, kernel_mangled
    kname = kname[0]
    pool = "k_trace_pool" in kname
    mk = source_markers()
    insts = funcs[kname]
    sym = symbolize(elf, [a for a, _, _ in insts])
    # histograms of the out-of-line functions, their own calls flattened in
    calls = {f: call_targets(funcs[f], starts) for f in funcs}
    memo = {}

    def flat(f, depth=0):
        if f in memo:
            return memo[f]
        h = collections.Counter(op for _, op, _ in funcs[f])
        if depth < 6:
            for _, callee in calls[f].items():
                if callee and callee in funcs and callee != f:
                    h += flat(callee, depth + 1)
        memo[f] = h
        return h

    regions = collections.defaultdict(collections.Counter)
    ktargets = calls[kname]
    for addr, op, _ in insts:
        fr = sym.get(addr) or [("?", 0, "?")]
        r = region_of(fr, pool, mk)
        regions[r][op] += 1
        if op == "s_swappc_b64" and ktargets.get(addr) in funcs:
            regions[r] += flat(ktargets[addr])
    # The compiler unrolls the Beckmann sampler's Newton loop (nine copies whatever `#pragma nounroll` says: its trip count is a constant); one RUN of the region
    # is one iteration, i.e. one copy: every copy evaluates ErfInv once, and ErfInv holds one v_sqrt_f32.
    nw = regions.get("newton iteration")
    if nw:
        copies = max(1, sum(c for op, c in nw.items() if op.startswith("v_sqrt_f32")))
        if copies > 1:
            regions["newton iteration"] = collections.Counter({op: c / copies for op, c in nw.items()})
            print("   (the Newton loop is %d copies in the binary; a run of the region is one of them)" % copies)
    return kname, regions


def features(hist, measured):
    """A region's contribution per execution to the hardware's counters."""
    f = collections.Counter()
    for op, n in hist.items():
        if op.startswith("v_"):
            f["VALU"] += n
            c = hw_class(op, measured)
            f[c] += n
        elif op.startswith("s_"):
            if op.startswith(("s_load_", "s_buffer_load", "s_memtime", "s_dcache", "s_atc")):
                f["SMEM"] += n
            elif not op.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_branch", "s_cbranch", "s_setpc", "s_swappc", "s_getpc")) or op.startswith(("s_cbranch", "s_branch")):
                f["SALU"] += n
        elif op.startswith("ds_"):
            f["LDS"] += n
        elif op.startswith(("global_load", "scratch_load", "flat_load", "buffer_load")):
            f["VMEM_RD"] += n
        elif op.startswith(("global_store", "scratch_store", "flat_store", "buffer_store", "global_atomic", "flat_atomic")):
            f["VMEM_WR"] += n
    return f


def issue_cycles(hist):
    cyc = 0
    for op, n in hist.items():
        if op.startswith("v_"):
            c = sm.cost(op)
            if sm.base(op).startswith("v_bitop3"):
                c = 2
            cyc += n * c
    return cyc


def fit(regions, y, measured, known=None):
    """Wave-level run counts of the regions: `known` ones (a diagnostic build's step counters) are taken as they are, the others fitted to what the known ones
    leave of the hardware's counters (non-negative least squares, every counter weighted by its own size)."""
    from scipy.optimize import nnls
    known = known or {}
    names = sorted(regions)
    keys = [k for k in list(CLASSES) + ["OTHER", "VALU"] if y.get(k)]
    A = np.array([[features(regions[r], measured).get(k, 0) for r in names] for k in keys], float)
    b = np.array([y[k] for k in keys], float)
    fixed = np.array([known.get(r, 0.0) for r in names])
    free = np.array([r not in known for r in names])
    w = 1.0 / np.maximum(b, b.max() * 1e-4)
    x = fixed.copy()
    if free.any():
        xf, _ = nnls((A[:, free]) * w[:, None], (b - A @ fixed) * w)
        x[free] = xf
    pred = A @ x
    return names, x, {k: (float(p), float(t)) for k, p, t in zip(keys, pred, b)}


def main():
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    pmc = json.load(open(pmc_path))
    measured = measured_classes()
    elfs = build_objects()
    for wl in sys.argv[1].split(","):
        rec = pmc[wl]
        args = rec["kernel"].split("<", 1)[1].rstrip(">").split(",")
        kernel = ("k_trace_pool" if "k_trace_pool" in rec["kernel"] else "k_trace") + "I" + "".join(("Lb1" if a.strip() == "true" else "Lb0" if a.strip() == "false" else "Li" + a.strip()) + "E" for a in args) + "EEvNS"
        elf = elfs["rl_render_pool" if "k_trace_pool" in rec["kernel"] else "rl_render"]
        kname, regions = analyse(kernel, elf, measured)
        y = dict(rec["valu_class_counts"]); y["VALU"] = rec["valu_insts_per_launch"]
        known = {}
        kn_path = os.path.join(ROOT, "profiles", "dynamic_mix_counts.json")      # wave-level step counts of the level-2 diagnostic build (tools/gpu_stamps.py), per workload
        if os.path.exists(kn_path):
            known = json.load(open(kn_path)).get(wl, {})
        names, x, table = fit(regions, y, measured, known)
        print("== %s  (%s)" % (wl, rec["kernel"]))
        print("   counter        hardware          regions x counts    error")
        worst = 0.0
        for k, (p, t) in table.items():
            err = (p - t) / t if t else 0.0
            worst = max(worst, abs(err)) if t > 0.005 * y["VALU"] else worst
            print("   %-12s %16.4g %16.4g   %+6.2f %%" % (k, t, p, 100 * err))
        total_cyc = 0.0
        dyn = collections.Counter()
        rows = []
        for r, n in zip(names, x):
            cyc = issue_cycles(regions[r])
            total_cyc += n * cyc
            for op, k in regions[r].items():
                if op.startswith("v_"):
                    dyn[op] += n * k
            rows.append((r, n, sum(k for op, k in regions[r].items() if op.startswith("v_")), cyc, n * cyc))
        print("   region                      runs (wave level)   VALU instr / run   issue clocks / run   share of the VALU issue clocks")
        for r, n, vi, cyc, tot in sorted(rows, key=lambda t: -t[4]):
            print("   %-26s %16.4g %12d %18d %14.1f %%" % (r, n, vi, cyc, 100 * tot / max(total_cyc, 1)))
        by = collections.defaultdict(lambda: [0.0, 0.0])
        for op, n in dyn.items():
            c = hw_class(op, measured)
            cost = 2 if sm.base(op).startswith("v_bitop3") else sm.cost(op)
            by[c][0] += n; by[c][1] += n * cost
        mean = {c: v[1] / v[0] for c, v in by.items() if v[0] > 0}
        cycles = rec.get("cycles_per_launch")
        out = {"regions": {r: {"runs_wave_level": float(n), "valu_instructions_per_run": int(vi), "issue_clocks_per_run": int(cyc), "share_of_valu_issue_clocks": float(tot / max(total_cyc, 1))} for r, n, vi, cyc, tot in rows},
               "fit": {k: {"hardware": t, "model": p, "rel_error": (p - t) / t if t else None} for k, (p, t) in table.items()},
               "worst_rel_error_of_a_counter_above_half_a_per_cent_of_the_instructions": worst,
               "mean_issue_cost": {c: round(v, 3) for c, v in sorted(mean.items())},
               "valu_weighted_cycles_per_launch": total_cyc,
               "valu_weighted_busy_fraction": total_cyc / (1024 * cycles) if cycles else None,
               "method": "tools/dynamic_mix.py: regions of the kernel's loop (inline stacks of a -gline-tables-only build) x wave-level run counts fitted to the hardware's class counters (non-negative least squares)"}
        # ---- anchored accounting: the regions whose runs a diagnostic build COUNTED and whose code is straight-line are charged exactly (opcode by opcode); what the
        # hardware's class counters hold beyond them is the remainder, priced class by class -- its INT32 and unclassified instructions at the static mix of the
        # code that is not in the anchored regions (between 2 and 4 clocks: the band)
        anchors = [r for r in names if r in known and known[r] > 0 and r in ("node step", "triangle step", "traversal: turn", "leaf list: boxes", "leaf list: set-up", "leaf list: pick", "leaf list: triangle step")]
        anch_feat = collections.Counter(); anch_cyc = 0.0
        for r in anchors:
            f = features(regions[r], measured)
            for k, v in f.items():
                anch_feat[k] += known[r] * v
            anch_cyc += known[r] * issue_cycles(regions[r])
        rest_hist = collections.Counter()
        for r in names:
            if r not in anchors:
                rest_hist += regions[r]
        rb = collections.defaultdict(lambda: [0.0, 0.0])
        for op, n in rest_hist.items():
            if op.startswith("v_"):
                c = hw_class(op, measured); cost = 2 if sm.base(op).startswith("v_bitop3") else sm.cost(op)
                rb[c][0] += n; rb[c][1] += n * cost
        rest_cost = {c: (v[1] / v[0] if v[0] else 3.0) for c, v in rb.items()}
        nominal = {"FMA_F32": 2, "MUL_F32": 2, "ADD_F32": 2, "FMA_F64": 4, "MUL_F64": 4, "ADD_F64": 4, "TRANS_F32": 8, "TRANS_F64": 16, "CVT": 4, "INT64": 4}
        rem, rem_cyc, rem_lo, rem_hi, over = {}, 0.0, 0.0, 0.0, {}
        for c in list(CLASSES) + ["OTHER"]:
            left = y.get(c, 0.0) - anch_feat.get(c, 0.0)
            if left < -0.02 * max(y.get(c, 0.0), 1.0):
                over[c] = left / max(y.get(c, 0.0), 1.0)
            left = max(left, 0.0)
            rem[c] = left
            if c in nominal:
                rem_cyc += left * nominal[c]; rem_lo += left * nominal[c]; rem_hi += left * nominal[c]
            else:
                rem_cyc += left * rest_cost.get(c, 3.0); rem_lo += left * 2.0; rem_hi += left * 4.0
        anch_insts = anch_feat.get("VALU", 0.0)
        simd_cycles = 1024 * cycles if cycles else None
        out["anchored"] = {"regions": anchors, "share_of_valu_instructions": anch_insts / y["VALU"], "issue_clocks": anch_cyc,
                           "remainder_instructions_by_class": rem, "remainder_cost_of_INT32_and_OTHER": {c: rest_cost.get(c) for c in ("INT32", "OTHER")},
                           "classes_the_anchored_regions_overshoot": over,
                           "valu_weighted_cycles_per_launch": anch_cyc + rem_cyc, "bounds": [anch_cyc + rem_lo, anch_cyc + rem_hi],
                           "valu_weighted_busy_fraction": (anch_cyc + rem_cyc) / simd_cycles if simd_cycles else None,
                           "valu_weighted_busy_bounds": [(anch_cyc + rem_lo) / simd_cycles, min(1.0, (anch_cyc + rem_hi) / simd_cycles)] if simd_cycles else None}
        a = out["anchored"]
        print("   anchored: %s = %.1f %% of the launch's VALU instructions, counted and priced opcode by opcode; remainder priced by class (INT32 %.2f, rest %.2f clocks)" % (
            ", ".join(anchors), 100 * a["share_of_valu_instructions"], rest_cost.get("INT32", 3.0), rest_cost.get("OTHER", 3.0)))
        print("   -> VALU issue %.3f of the launch's SIMD cycles, band [%.3f, %.3f] (static mix of round 4: %.3f, band %s)%s" % (
            a["valu_weighted_busy_fraction"] or 0, (a["valu_weighted_busy_bounds"] or [0, 0])[0], (a["valu_weighted_busy_bounds"] or [0, 0])[1], rec.get("valu_weighted_busy_fraction") or 0.0,
            rec.get("valu_weighted_busy_bounds"), ("; anchored regions hold MORE than the hardware counted in: %s" % over) if over else ""))
        print("   mean issue cost by class (dynamic): %s" % out["mean_issue_cost"])
        print("   weighted VALU issue clocks per launch %.4g -> %.3f of the launch's SIMD cycles (static mix: %.3f); worst counter error %.2f %%" % (
            total_cyc, out["valu_weighted_busy_fraction"] or 0.0, rec.get("valu_weighted_busy_fraction") or 0.0, 100 * worst))
        rec["valu_dynamic_mix"] = out
    json.dump(pmc, open(pmc_path, "w"), indent=1)


if __name__ == "__main__":
    main()
