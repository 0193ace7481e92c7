"""profiles/dynamic_mix_counts.json from the level-2 diagnostic build's wave-level step counters (tools/gpu_stamps.py cornell | breakfast | interior | textured, the lines
"<workload>: ... diagnostic slots (wave level): [4] .. [5] .. [6] .. [7] .. [16] .. [17] .. [18] .. [19] .. trips .."; the LAST such line of a workload is the full frame).
usage: python tools/dynamic_mix_counts.py <file with those lines>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
last = {}
for line in open(sys.argv[1]):
    m = re.match(r"(\w+): .*diagnostic slots \(wave level\): \[4\] (\d+) \[5\] (\d+) \[6\] (\d+) \[7\] (\d+) \[16\] (\d+) \[17\] (\d+) \[18\] (\d+) \[19\] (\d+) trips (\d+)", line)
    if m:
        last[m.group(1)] = [int(x) for x in m.groups()[1:]]
zero = {"texel fetch": 0, "scatter: lambertian": 0, "scatter: metal": 0, "scatter: dielectric": 0, "scatter: diffuse_light": 0}
out = {"_source": "wave-level step counters of the level-2 diagnostic build (make variant VARIANT=stamps2 EXTRA=-DRL_DIAG_STAMPS=2; tools/gpu_stamps.py): " + os.path.basename(sys.argv[1])}
if "cornell" in last:
    s4, s5, s6, s7, s16, s17, s18, s19, trips = last["cornell"]
    out["cornell_1080p_64spp"] = dict(zero, **{"leaf list: boxes": s4 // 6, "leaf list: set-up": s4 // 6, "leaf list: pick": s6, "leaf list: triangle step": s5, "newton iteration": s16,
                                                 "scatter: microfacet": s18, "traverse: glue": trips, "shade": trips, "miss + fold": trips, "tree walk": 0})
for key, wl in (("breakfast", "breakfast_300k_1080p_128spp"), ("interior", "breakfast_interior_300k_1080p_128spp"), ("textured", "breakfast_textured_interior_300k_1080p_128spp")):
    if key in last:
        s4, s5, s6, s7, s16, s17, s18, s19, trips = last[key]
        if key == "textured":
            zero = dict(zero); zero.pop("texel fetch")      # (texel fetches run there: their count is not measured at wave level, the fit keeps it free)
        out[wl] = dict(zero, **{"node step": s4, "triangle step": s5, "traversal: turn": s4 + s5, "refill": s6, "shade: a round of hits": s7, "newton iteration": s16, "scatter: microfacet": s18,
                                "shade: misses + hand-back": trips})
json.dump(out, open(os.path.join(ROOT, "profiles", "dynamic_mix_counts.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
