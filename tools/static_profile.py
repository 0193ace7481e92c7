"""Static cost profile of one kernel by SOURCE LINE: hipcc -gline-tables-only -S keeps a .loc in front of every instruction; each VALU instruction is
priced with the measured issue cost of its opcode (tools/static_mix.py cost table = profiles/valu_calib.json) and charged to the source function it
was inlined from (the line's enclosing function in rl_render.hip / rl_glibc_math.h ...).  Every instruction counts ONCE: loops and branches are not
weighted, so this shows how dear one pass through each piece of code is, not how often it runs (wave-step counters give that: RL_DIAG_STAMPS=2).

usage: python tools/static_profile.py <file.s from hipcc -gline-tables-only --cuda-device-only -S> <mangled kernel substring>"""
import re, sys, collections, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from static_mix import cost, base
asm, want = sys.argv[1], sys.argv[2]
files = {}
func_of_line = {}


def functions_of(path):
    """line -> name of the function whose body holds it (a brace-depth scan good enough for this code base)."""
    if path in func_of_line:
        return func_of_line[path]
    out = {}
    try:
        src = open(path).read().split("\n")
    except OSError:
        func_of_line[path] = out; return out
    cur, depth, start_depth = None, 0, 0
    sig = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:__device__|__global__|RLM_FN|RTM_FN|__host__|static|inline|__forceinline__|__noinline__|\w+\s+)*[\w:<>\*&\s]+?\b(\w+)\s*\([^;]*$")
    for i, line in enumerate(src, 1):
        if depth == 0:
            m = sig.match(line.strip())
            if m and not line.strip().startswith(("#", "//", "}", "if", "for", "while")):
                cur = m.group(1)
        out[i] = cur if (depth > 0 or "{" in line) else out.get(i - 1, cur)
        depth += line.count("{") - line.count("}")
        if depth == 0 and "}" in line:
            pass
    func_of_line[path] = out
    return out


by_func = collections.defaultdict(lambda: [0, 0])
by_line = collections.defaultdict(lambda: [0, 0])
inside, cur_loc = False, None
for line in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', line)
    if m:
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if not m.group(3).startswith("/") else m.group(3)
        continue
    if re.match(r"^_Z\w+:", line):
        inside = want in line
        continue
    if not inside:
        continue
    if "s_endpgm" in line:
        inside = False
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur_loc = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r"\s+(v_[a-z0-9_]+)\s", line)
    if m and cur_loc:
        path = files.get(cur_loc[0], "?")
        fn = functions_of(path).get(cur_loc[1]) or "?"
        key = "%s:%s" % (os.path.basename(path), fn)
        c = cost(m.group(1))
        by_func[key][0] += 1; by_func[key][1] += c
        by_line[(os.path.basename(path), cur_loc[1])][0] += 1; by_line[(os.path.basename(path), cur_loc[1])][1] += c
tot_i = sum(v[0] for v in by_func.values()); tot_c = sum(v[1] for v in by_func.values())
print("kernel %s: %d static VALU instructions, %d issue cycles (each instruction once)" % (want, tot_i, tot_c))
for k, (n, c) in sorted(by_func.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %-52s %5d instr %6d cycles  %5.1f %%" % (k, n, c, 100.0 * c / tot_c))
print("dearest source lines:")
for (f, l), (n, c) in sorted(by_line.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  %s:%d  %d instr %d cycles" % (f, l, n, c))
