"""Time library variants (RAYLIB_LIB) on the bench workload, interleaved rounds in separate processes."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = os.environ.get("WL", "cornell_1080p_64spp")
variants = sys.argv[1:]
res = {v: [] for v in variants}
for rnd in range(3):
    for v in variants:
        env = dict(os.environ)
        name, _, extra = v.partition(":")
        env["RAYLIB_LIB"] = os.path.join(ROOT, "software-raytracing_amd", "libraylib%s.so" % (("_" + name) if name != "base" else ""))
        for kv in extra.split(","):
            if kv: k, _, val = kv.partition("="); env[k] = val
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", os.environ.get("STEPS", "3"), "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--workload", workload], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line: print(v, "FAILED", out.stderr[-300:]); continue
        d = json.loads(line[-1]); res[v].append((d["value"], d["roofline"]["avg_launch_ms"]))
for v in variants:
    print("%-28s" % v, " ".join("%.0f Mrays/s (%.2f ms)" % r for r in res[v]))
