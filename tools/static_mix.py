"""Static VALU opcode mix of a megakernel, priced with the measured issue costs (profiles/valu_calib.json).

The hardware counts VALU instructions by CLASS (SQ_INSTS_VALU_{ADD,MUL,FMA}_F32/F64, TRANS, INT32, INT64, CVT); two classes mix
2-cycle and 4-cycle opcodes -- INT32 (and / or / xor / add / sub / shift-right: 2; multiplies, bit-field ops, shift-left, 3-operand forms: 4)
and everything the class counters leave out (v_mov: 2; compares, selects, min / max, lane ops, the divide helpers: 4).  No counter
separates those, so tools/pmc_traffic.py prices the two classes with the STATIC mix of the kernel's code (this script): the kernel's
own body plus every out-of-line device function it can call, each opcode once.  Lower and upper bounds (all 2 / all 4) are reported
next to it.

usage: python tools/static_mix.py [libraylib.so] -> JSON on stdout: per kernel {"INT32": cost, "OTHER": cost, "histogram": {...}}"""
import json, os, re, struct, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_mov_b32", "v_fmaak_f32", "v_fmamk_f32", "v_ashrrev_i32", "v_not_b32", "v_nop"}   # 2 issue cycles per wave64 (valu_calib); the rest 4, transcendentals 8 / 16
F32 = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmaak_f32", "v_fmamk_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_mul_legacy_f32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
# which opcodes the hardware counts under SQ_INSTS_VALU_INT32 was MEASURED in round 5 (tools/valu_class_pmc.sh -> profiles/valu_classes.json): integer add / sub / multiply /
# mad, compares of integers, bit counts and field extracts, the arithmetic shift right -- and NOT the bit-wise and / or / xor / not, the logical shifts, v_and_or,
# v_or3, v_bfi, v_lshl_or, v_alignbit, v_perm, v_bitop3, which no class counter sees (rounds 3 - 4 listed them here by guess)
INT32 = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_mul_lo_u32", "v_mul_hi_u32",
         "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_i32_i24", "v_bfe_u32", "v_bfe_i32", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32",
         "v_xad_u32", "v_min_u32", "v_max_u32", "v_min3_u32", "v_max3_u32", "v_min_i32", "v_max_i32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32",
         "v_subrev_co_u32", "v_bcnt_u32_b32", "v_ffbh_u32", "v_ffbl_b32", "v_mul_hi_i32", "v_sad_u32", "v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32",
         "v_med3_u32", "v_med3_i32", "v_min3_i32", "v_max3_i32", "v_ffbh_i32")
INT64 = ("v_mad_u64_u32", "v_mad_i64_i32", "v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64", "v_lshl_add_u64", "v_mov_b64")


def code_objects(path):
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], 0
    while True:
        at = data.find(magic, at)
        if at < 0:
            break
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[at + off:at + off + size])
        at += 24
    return out


def base(op):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)


def classify(op):
    b = base(op)
    if b.startswith(TRANS):
        return "TRANS_F64" if b.endswith("f64") else "TRANS_F32"
    if b.startswith("v_cvt_"):
        return "CVT"
    if b in F32:
        return "F32"
    if b.endswith("_f64") and b.split("_")[1] in ("fma", "fmac", "mul", "add"):
        return "F64"
    if b in INT64:
        return "INT64"
    if b in INT32 or re.match(r"v_cmpx?_\w+_[ui](16|32)$", b):
        return "INT32"
    return "OTHER"


def cost(op):
    b = base(op)
    if b.startswith(TRANS):
        return 16 if b.endswith("f64") else 8
    return 2 if (b in FAST and not op.endswith("dpp")) else 4


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "software-raytracing_amd", "libraylib.so")
    funcs = {}
    for i, co in enumerate(code_objects(lib)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1); funcs.setdefault(cur, collections.Counter()); continue
            m = re.match(r"^\s+(v_[a-z0-9_]+)\s", line)
            if m and cur:
                funcs[cur][m.group(1)] += 1
    helpers = collections.Counter()
    for name, h in funcs.items():
        if "k_" not in name.split("rl")[-1][:6] and not re.search(r"\dk_", name):   # out-of-line device functions (libm, TexFetch, Erf ...)
            helpers += h
    res = {}
    for name, h in funcs.items():
        if "k_trace" not in name:
            continue
        tot = h + helpers
        by = collections.defaultdict(lambda: [0, 0])
        for op, n in tot.items():
            c = classify(op)
            by[c][0] += n; by[c][1] += n * cost(op)
        short = re.sub(r"^_ZN2rl\d+", "", name)
        short = re.sub(r"EEvNS_.*$", "", short)
        res[short] = {"static_valu_instructions": sum(tot.values()),
                      "mean_cost": {c: round(v[1] / v[0], 3) for c, v in sorted(by.items()) if v[0]},
                      "share": {c: round(v[0] / sum(tot.values()), 4) for c, v in sorted(by.items()) if v[0]},
                      "top_other": {op: n for op, n in collections.Counter({o: n for o, n in tot.items() if classify(o) == "OTHER"}).most_common(14)}}
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
