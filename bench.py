#!/usr/bin/env python3
"""bench.py -- Mrays/s of the hot path (Raylib_Render's megakernel) on N MI355X GPUs.

A "step" is one full frame of BASELINE.json's configs[1]: synthetic Cornell box (36 triangles), 1920x1080, 64 spp,
maxPathLength 5, fixed RNG seed, scene resident in HBM.  Two ways to run N GPUs, the same JSON schema from both:

  library mode  `python bench.py --gpus N`  (no torch.distributed.run).  The timed call is Raylib_Render itself -- the reference's
                entry point (raylib/raylib.cc:231-239), the frame left resident in HBM.  N > 1 sets RAYLIB_NUM_GPUS=N before
                Raylib_Initialize: the library splits the frame's 8x8 cells over N devices of this process behind that same call
                (csrc/rl_runtime.inl RenderMulti: RCCL grouped send / recv or peer copies, one scatter kernel).  Exits non-zero when fewer
                than N devices are visible (RAYLIB_GPU_MAP, a test aid, may name one device several times).
  process mode  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
                --gpus N ...` (what the driver launches for N > 1): one process per GPU, each renders its cells into device memory
                (RaylibAMD_RenderDevice), one RCCL gather per frame brings them to rank 0.  --gpus must equal WORLD_SIZE.

Strong scaling: the total work is fixed.  Outside the timed region the frame is checked against windows rendered by the reference
build (tests/golden/bench_windows.npz).  Rank 0 prints ONE JSON line; DESIGN.md section 5 describes every field.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RAYLIB_QUIET", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

# ---- hardware constants (MI355X_MICROARCH.md, chip-level parameters): peaks are CONSTANTS, never derived from a run -----------------
HBM_PEAK_GBS = 8000.0            # HBM3E spec peak
NUM_SIMDS = 256 * 4              # 256 CUs x 4 SIMD-32
MAX_CLOCK_HZ = 2.4e9             # "Max clock 2400 MHz"
# VALU roof: a SIMD's vector issue port is busy `cost` cycles per wave64 instruction (2 for v_fma_f32 = 157.3 TFLOP/s f32 at 2.4 GHz;
# the other classes: tools/valu_calib.hip, measured on the bench box, profiles/valu_calib.json).  Unit: SIMD issue cycles per second.
VALU_PEAK_GCYC = NUM_SIMDS * MAX_CLOCK_HZ / 1e9

WORKLOADS = {
    # BASELINE.json configs[1]
    "cornell_1080p_64spp": dict(scene="cornell", kw={}, w=1920, h=1080, spp=64, max_path=5, camera="cornell"),
    # configs[2]-sized stress: a memory-bound megakernel (the pool schedule on 298 k triangles)
    "breakfast_300k_1080p_128spp": dict(scene="cornell", kw=dict(tess=91, displace_fraction=0.2), w=1920, h=1080, spp=128, max_path=5, camera="breakfast"),
    # the same scene from INSIDE (what the named scenes are: interiors): every pixel looks at geometry, no cell can be dropped (round 4)
    "breakfast_interior_300k_1080p_128spp": dict(scene="cornell", kw=dict(tess=91, displace_fraction=0.2), w=1920, h=1080, spp=128, max_path=5, camera="breakfast_interior"),
    # ... and WITH textures and alpha cut-outs (round 5; SURVEY 8d: Breakfast Room / San Miguel use map_Kd throughout, C5 is "foliage cards with an alpha-cut-out
    # texture"): albedo maps on every wall, a fifth of the triangles as foliage cards whose map is two thirds holes -- the cut-out test runs inside traversal for
    # every candidate of a mapped material (reference geom/triangle.cc:54, render/material.cc:397-404, render/texture.cc:30-53); camera inside the room
    "breakfast_textured_interior_300k_1080p_128spp": dict(scene="textured", kw=dict(tess=91, displace_fraction=0.2), w=1920, h=1080, spp=128, max_path=5, camera="breakfast_interior"),
}
HEADLINE = "cornell_1080p_64spp"
EXTRA = "breakfast_300k_1080p_128spp"
EXTRA2 = "breakfast_interior_300k_1080p_128spp"
EXTRA3 = "breakfast_textured_interior_300k_1080p_128spp"


def cpu_baseline(workload, cam, ref_rays_per_sample, gpu_frame_s):
    """The reference's own CPU path (Renderer::RenderScene + its ThreadPool, built in place into
    oracle/_ref/libref_native.so) timed on this host's cores on a bounded sample of the same
    workload.  Falls back to the oracle restatement (kind "port") when the prebuilt binary is absent."""
    sys.path.insert(0, ROOT)
    from oracle import ffi, objflat           # checker only: never on the product path
    from raylib_amd import scenes
    orc = ffi.load_oracle()
    tmp = tempfile.mkdtemp()
    obj, _ = getattr(scenes, workload["scene"])(os.path.join(tmp, "cpu.obj"), **workload["kw"])
    tex = scenes.textured_textures()
    flat = objflat.load_obj(obj, orc, sun_illuminance=cam["sun"], sun_direction=cam["sun_dir"],
                            texture_loader=lambda p: scenes.texture_as_float(tex[os.path.basename(p)]) if os.path.basename(p) in tex else None)
    w, h = workload["w"], workload["h"]
    cores = os.cpu_count() or 1
    # bounded sample: ~10-30 s of CPU work.  Few cores: 1/8 of the samples; many cores: the whole step
    # (thread start-up and the reference's 100 ms completion poll would otherwise dominate).
    if flat.triangles.shape[0] < 1000:
        spp = workload["spp"] if cores >= 32 else max(1, workload["spp"] // 8)
    else:
        spp = 4 if cores >= 32 else 1
    camera = ffi.make_camera(cam["origin"], cam["look_at"], cam["fov"], w / h)
    st = ffi.make_settings(w, h, spp, max_path=workload["max_path"])
    ref = ffi.load_ref(seeded=False)
    if ref is not None:
        scene = ref.scene_create(flat, 1)
        t0 = time.time(); ref.render_native(scene, camera, st); dt = time.time() - t0
        kind = "reference"
    else:
        scene = orc.scene_create(flat, 1)
        t0 = time.time(); orc.render(scene, camera, st, seed=1, threads=cores); dt = time.time() - t0
        kind = "port"
    samples = w * h * spp
    cpu_frame_s = dt * workload["spp"] / spp
    return {"value": samples * ref_rays_per_sample / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
            "seconds": dt, "camera_samples_per_s": samples / dt,
            # what "x times the CPU" means here: the TIME of the same frame, CPU reference over GPU -- not a ratio of ray rates (the reference traces
            # every camera sample of the frame, the GPU path leaves the cells outside the scene's silhouette out of its job list)
            "frame_time_ratio_cpu_over_gpu": cpu_frame_s / gpu_frame_s if gpu_frame_s > 0 else None,
            "cpu_frame_seconds_scaled_to_full_spp": cpu_frame_s,
            "sample": "%dx%d at %d spp of the same scene/camera (%.1f%% of the step's camera samples), all %d host threads; "
                      "rays = camera samples x the reference-equivalent queries per camera sample (%.4f): the reference's TraceScene issues one accel->Hit per path "
                      "segment and one per sun test for EVERY camera sample of the frame (renderer.cc:129,194), which is what the GPU path's counters come to "
                      "over the whole frame -- executed queries plus the one root-box query (two with a sun) each sample of a dropped cell stands for; the "
                      "reference itself has no ray counter"
                      % (w, h, spp, 100.0 * spp / workload["spp"], cores, ref_rays_per_sample)}


def golden_windows(workload_name, frame_hw):
    """The timed frame against windows of the same frame rendered by the REAL reference build (tests/golden/gen_golden.py ->
    bench_windows.npz: data; only windows in which no sample met two surfaces at exactly the same t).  Bit for bit."""
    import numpy as np
    g = None
    for name in ("bench_windows.npz", "bench_windows_textured.npz"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(path):
            f = np.load(path)
            if workload_name + "_pos" in f.files:
                g = f
    if g is None:
        return None
    pos, px = g[workload_name + "_pos"], g[workload_name + "_px"]
    bad = 0
    for (x0, y0), want in zip(pos, px):
        got = frame_hw[y0:y0 + 16, x0:x0 + 16]
        bad += int((np.ascontiguousarray(got[..., :3]).view(np.uint32) != np.ascontiguousarray(want[..., :3]).view(np.uint32)).any(-1).sum())
    n = len(pos) * 256
    return ("%d / %d pixels of %d reference-rendered windows bit-identical" % (n - bad, n, len(pos))) if bad == 0 else \
           ("MISMATCH: %d of %d pixels of the reference-rendered windows differ" % (bad, n))


def step_times(ts):
    """min / median / max of the host-visible time of each timed step (ms)."""
    if not ts:
        return None
    return {"min": min(ts) * 1e3, "median": statistics.median(ts) * 1e3, "max": max(ts) * 1e3, "n": len(ts)}


def roofline_block(workload_name, acc, world, build_id, tree_walk=None, gpus_in_acc=1, frame_samples_per_launch=None):
    """The roofline object of the line.

    Measured in THIS run: launches, average megakernel launch duration (HIP events on the library's stream), the launch's record
    counters -> algorithmic bytes.  Hardware counters cannot be read from inside this process: HBM traffic and the VALU instruction
    counts by class are REPLAYED from the committed rocprofv3 --pmc passes of this same command (profiles/pmc_traffic.json, written by
    tools/profile_round.sh -> tools/pmc_traffic.py); they sit in `replayed_pmc` with the build id of the library they were taken from,
    and when that is not the loaded library's (`RaylibAMD_BuildId`) the line says STALE and takes its headline figure from what this run
    measured alone.  Peaks are constants.

    N > 1: `acc` holds either the sum over the N ranks of one process (library mode: gpus_in_acc = N, the launch time is
    the slowest rank's) or this rank's own launches (process mode: gpus_in_acc = 1); `frame_samples_per_launch` = the camera samples ALL ranks
    executed per launch.  The N = 1 counter passes are then replayed SCALED by the share of the frame's executed camera samples `acc` covers
    and priced against gpus_in_acc GPUs' peaks -- labelled as such; bytes a kernel was served from LDS are never put over the HBM peak."""
    launches = max(1, acc["launches"])
    avg_launch_ms = acc["trace_ms"] / launches
    sec = avg_launch_ms * 1e-3
    samples_per_launch = acc["samples"] / launches
    # ---- algorithmic bytes, SURVEY 8(d): a property of the WORKLOAD -- the records a tree walk of this scene fetches: 64 B per 4-wide
    # grid-node step (a float-box step = 2 records) or 80 B per 8-wide one (the scenes deep enough to get that tree, RaylibAMDStats.treeWidth /
    # nodeBytes say which), 64 B per triangle test, 64 B per shading record, 16 B per texel and per pixel.
    # Where the kernel that ran walks something else (the Cornell class: the leaf list, every box of it per ray, from LDS), the tree-walk
    # counts come from one untimed frame with RAYLIB_LEAF_LIST=0 and the kernel's own LDS-served bytes are reported next to them.
    ran_bytes = acc["bytes"] / launches
    leaf_list_ran = acc.get("tree_width") == 0        # k_trace<.., LDS = 2>: no tree is walked, every ray reads every leaf box from LDS
    hbm_peak = HBM_PEAK_GBS * gpus_in_acc             # the peak of the GPUs whose launches `acc` adds up
    valu_peak = VALU_PEAK_GCYC * gpus_in_acc
    served = None
    if leaf_list_ran:
        served = {"schedule": "leaf list (k_trace<.., LDS = 2>): every traced ray reads all leaf-box records, from LDS",
                  "lds_served_bytes_per_launch": ran_bytes, "lds_served_bytes_per_camera_sample": ran_bytes / max(1.0, samples_per_launch)}
    if tree_walk is not None:
        alg_bytes = tree_walk["bytes_per_launch"]
    elif leaf_list_ran:
        alg_bytes = None                              # (no tree-walk frame was rendered: the LDS-served bytes are NOT the workload's HBM demand)
    else:
        alg_bytes = ran_bytes
    alg_gbs = alg_bytes / sec / 1e9 if (sec > 0 and alg_bytes is not None) else None
    algorithmic = {"bytes_per_launch": alg_bytes, "bytes_per_camera_sample": None if alg_bytes is None else alg_bytes / max(1.0, samples_per_launch), "gbs": alg_gbs,
                   "frac_of_hbm_peak": None if alg_gbs is None else alg_gbs / hbm_peak, "hbm_peak_gbs": hbm_peak, "gpus": gpus_in_acc,
                   "definition": "node bytes x node records of the tree walk (64 B on the 4-wide tree, 80 B on the 8-wide one) + 64 B x (triangle records + shading records) + 16 B x (texels + pixels): the tree walk's counts whatever schedule ran"
                                 + ("; summed over the %d ranks of this process, against %d GPUs' peak" % (gpus_in_acc, gpus_in_acc) if gpus_in_acc > 1 else ""),
                   "tree_width": acc.get("tree_width"), "node_bytes": acc.get("node_bytes"),
                   "served_elsewhere": served}
    out = {"kernel": "k_trace" if acc["paths_per_wave"] <= 64 else "k_trace_pool", "paths_per_wave": int(acc["paths_per_wave"]),
           "avg_launch_ms": avg_launch_ms, "launches": acc["launches"], "job_heads": acc.get("job_heads"), "algorithmic": algorithmic}
    # ---- replayed hardware counters (of the N = 1 run of this workload; N > 1: scaled by the share of the frame's executed samples `acc` covers)
    rec, stale, source = None, None, None
    share = 1.0
    if world > 1:
        share = (samples_per_launch / frame_samples_per_launch) if frame_samples_per_launch else None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if share is not None and os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(workload_name)
        except (OSError, ValueError) as e:
            source = "profiles/pmc_traffic.json could not be read: %s" % e
    if rec:
        stale = rec.get("build_id") != build_id
        source = ("STALE: " if stale else "") + "replayed from profiles/pmc_traffic.json (%s, build %s; loaded library is build %s)" % (rec.get("round"), rec.get("build_id"), build_id)
        if world > 1:
            source += "; N = 1 counters SCALED by %.4f = the share of the frame's executed camera samples these launches cover, against %d GPU(s)' peak" % (share, gpus_in_acc)
    traffic = rec["hbm_bytes_per_launch"] * share if rec else None
    hbm = None
    if rec and stale and rec.get("cycles_per_launch"):
        # counters of another build: priced against THAT build's launch (its cycles at the spec clock), not against this run's time -- old counts over a
        # new time are a fraction of nothing; the line says STALE and the headline falls back to this run's algorithmic bytes
        sec = rec["cycles_per_launch"] / 2.4e9 * share / gpus_in_acc
    if rec and sec > 0:
        gbs = traffic / sec / 1e9
        hbm = {"measured_bytes_per_launch": traffic, "gbs": gbs, "frac": gbs / hbm_peak, "tcc_hit_rate": rec.get("tcc_hit_rate"), "scaled_from_n1": world > 1}
    valu = None
    if rec and sec > 0 and rec.get("valu_weighted_cycles_per_launch"):
        wc = rec["valu_weighted_cycles_per_launch"] * share
        ach = wc / sec / 1e9
        valu = {"weighted_issue_cycles_per_launch": wc, "insts_per_launch": rec.get("valu_insts_per_launch") * share if rec.get("valu_insts_per_launch") else None,
                "scaled_from_n1": world > 1,
                "mean_cost_cycles_per_inst": rec["valu_weighted_cycles_per_launch"] / rec["valu_insts_per_launch"], "class_counts": rec.get("valu_class_counts"),
                "achieved_gcyc_per_s": ach, "peak_gcyc_per_s": valu_peak, "frac_of_spec_peak": ach / valu_peak,
                # the same cycles against the cycles the launch really had (GRBM_GUI_ACTIVE / 8 of the profiled pass: the clock the chip held)
                "frac_of_profiled_pass_cycles": rec.get("valu_weighted_busy_fraction"), "unweighted_2cyc_busy_fraction": rec.get("valu_busy_fraction"),
                "weighted_busy_bounds_of_profiled_pass": rec.get("valu_weighted_busy_bounds"),
                # hardware's own view (SQ_ACTIVE_INST_VALU: the port's time in 4-clock units, a 2-cycle instruction charged like a 4-cycle one -- an upper bound)
                "hw_active_inst_valu_x4_over_pass_cycles": rec.get("valu_hw_active_x4_fraction"), "hw_refined_fraction_of_pass_cycles": rec.get("valu_hw_refined_fraction"),
                "lane_utilisation": rec.get("valu_lane_utilisation"), "wave_wait_fraction": rec.get("wave_wait_fraction"),
                "waves_per_simd": rec.get("waves_per_simd"), "salu_insts_per_launch": rec.get("salu_insts_per_launch"),
                "profiled_clock_ghz": rec.get("profiled_clock_ghz"),
                # round 5: the mix weighted by the code that runs (tools/dynamic_mix.py): the loop's straight-line regions counted by a diagnostic build and priced
                # opcode by opcode, the rest by the hardware's class counters -- a narrower band than the static mix's
                "dynamic_mix": None if not rec.get("valu_dynamic_mix") else {
                    "anchored_regions": rec["valu_dynamic_mix"]["anchored"]["regions"],
                    "anchored_share_of_valu_instructions": rec["valu_dynamic_mix"]["anchored"]["share_of_valu_instructions"],
                    "frac_of_profiled_pass_cycles": rec["valu_dynamic_mix"]["anchored"]["valu_weighted_busy_fraction"],
                    "bounds_of_profiled_pass": rec["valu_dynamic_mix"]["anchored"]["valu_weighted_busy_bounds"],
                    "issue_clock_budget_by_region": {r: v["share_of_valu_issue_clocks"] for r, v in rec["valu_dynamic_mix"]["regions"].items() if v["share_of_valu_issue_clocks"] >= 0.005}}}
    # ---- the vector memory path's request rate (round 5): one access per lane and load instruction whatever the width; with the VALU port the tree walk's other ceiling
    vmem = None
    if rec and sec > 0 and rec.get("tcp_accesses_per_launch") and not stale:
        clk = rec["cycles_per_launch"]
        vmem = {"tcp_accesses_per_launch": rec["tcp_accesses_per_launch"] * share, "per_clock_and_cu_in_the_profiled_pass": rec.get("tcp_accesses_per_clock_and_cu"),
                "ceiling_per_clock_and_cu": rec.get("tcp_request_ceiling_per_clock_and_cu"), "frac": (rec.get("tcp_accesses_per_clock_and_cu") or 0.0) / (rec.get("tcp_request_ceiling_per_clock_and_cu") or 1.0),
                "definition": "TCP_TOTAL_CACHE_ACCESSES per clock and CU of the profiled pass over the rate a kernel of nothing but scattered 16-byte loads reaches (tools/vmem_width_bench.hip)"}
    # ---- headline: the resource nearest its roof among those measured with fresh counters; else this run's algorithmic bytes
    scaled = " -- replayed from the N = 1 passes, scaled by executed camera samples" if world > 1 else ""
    if valu and not stale and (not hbm or valu["frac_of_spec_peak"] >= hbm["frac"]):
        head = {"achieved": valu["achieved_gcyc_per_s"], "peak": valu_peak, "unit": "G SIMD-issue-cycles/s", "frac": valu["frac_of_spec_peak"], "resource": "VALU issue (class-weighted)" + scaled}
    elif hbm and not stale:
        head = {"achieved": hbm["gbs"], "peak": hbm_peak, "unit": "GB/s", "frac": hbm["frac"], "resource": "HBM (measured bytes)" + scaled}
    elif alg_gbs is not None and alg_gbs <= hbm_peak:
        head = {"achieved": alg_gbs, "peak": hbm_peak, "unit": "GB/s", "frac": alg_gbs / hbm_peak, "resource": "HBM (algorithmic bytes of the tree walk; no fresh counters)"}
    elif alg_gbs is not None:
        # more bytes per second than HBM can deliver: the walk's records came out of the caches -- a demand figure, not a fraction of a roof
        head = {"achieved": alg_gbs, "peak": hbm_peak, "unit": "GB/s", "frac": None,
                "resource": "none: the tree walk's algorithmic bytes (%.2f x the HBM peak in this time) were served by the caches, and there are no fresh counter passes for this build" % (alg_gbs / hbm_peak)}
    else:
        head = {"achieved": None, "peak": hbm_peak, "unit": "GB/s", "frac": None,
                "resource": "none: no fresh counter passes for this build, and the kernel that ran reads its scene from LDS (its bytes are not an HBM figure)"}
    vf = valu["frac_of_profiled_pass_cycles"] if valu and valu["frac_of_profiled_pass_cycles"] and world == 1 else (valu["frac_of_spec_peak"] if valu else 0.0)
    hf = hbm["frac"] if hbm else 0.0
    if hbm and hf >= 0.6 and hf >= vf:
        bound = "hbm"
    elif valu and vf >= 0.75:
        bound = "valu"
    elif valu or hbm:
        bound = "latency (VALU issue %.0f %% of the launch's cycles, HBM %.0f %% of peak: neither saturated)" % (100 * vf, 100 * hf)
    else:
        bound = "unknown (no counter passes for this workload)"
    line = {"bound": bound}
    line.update(head)
    line.update({"traffic": traffic, "traffic_source": source})
    line.update(out)
    line.update({"hbm": hbm, "valu": valu, "vmem_requests": vmem,
                 "replayed_pmc": None if not rec else {"file": "profiles/pmc_traffic.json", "round": rec.get("round"), "build_id": rec.get("build_id"),
                                                       "loaded_build_id": build_id, "stale": stale, "kernel": rec.get("kernel")}})
    return line


def accumulate(acc, stats, binding):
    acc["rays"] += stats.rays; acc["trace_ms"] += stats.traceKernelMs; acc["launches"] += stats.traceLaunches
    acc["kernel_ms"] += stats.kernelMs
    acc["bytes"] += binding.algorithmic_bytes(stats)
    acc["samples"] += stats.cameraSamples
    acc["texels"] = acc.get("texels", 0) + stats.texFetches; acc["nodes"] = acc.get("nodes", 0) + stats.nodesVisited; acc["tris"] = acc.get("tris", 0) + stats.trisTested
    acc["culled_rays"] += stats.culledRays; acc["culled_samples"] += stats.culledSamples; acc["culled_cells"] = stats.culledCells; acc["listed_cells"] = stats.listedCells
    acc["paths_per_wave"] = stats.pathsPerWave
    acc["job_heads"] = stats.jobHeads
    acc["tree_width"] = int(stats.treeWidth); acc["node_bytes"] = int(stats.nodeBytes)


def new_acc():
    return dict(rays=0, trace_ms=0.0, launches=0, bytes=0, samples=0, kernel_ms=0.0, paths_per_wave=64, job_heads=None,
                culled_rays=0, culled_samples=0, culled_cells=0, listed_cells=0)


def work_block(acc, steps):
    """What one step consisted of: executed on the device (what `value` is made of) next to what the silhouette cull left out (csrc/rl_cull.cc)."""
    k = max(1, steps)
    return {"rays_executed_per_step": acc["rays"] / k, "camera_samples_executed_per_step": acc["samples"] / k,
            "rays_accounted_not_traced_per_step": acc["culled_rays"] / k, "camera_samples_not_traced_per_step": acc["culled_samples"] / k,
            "cells_culled": acc["culled_cells"], "cells_listed": acc["listed_cells"],
            "node_records_per_ray": acc.get("nodes", 0) / max(1, acc["rays"]), "triangle_records_per_ray": acc.get("tris", 0) / max(1, acc["rays"]),
            "texel_fetches_per_ray": acc.get("texels", 0) / max(1, acc["rays"]),
            "definition": "a ray = one closest-hit or occlusion query EXECUTED by a kernel (reference renderer.cc:129,194); `value` counts only those. "
                          "Samples of cells outside the scene's silhouette are neither generated nor traced: k_resolve writes the miss shader's constant for them"}


def library_run(lib, binding, scenes, workload_name, steps, warmup, n_gpus, rank_tag="0", want_device_entry=True):
    """K timed Raylib_Render calls on one workload (library mode).  Returns the measurements and, untimed, the frame check, the
    render + dump time, the device-pointer entry's time and the tree-walk record counts."""
    import numpy as np
    wl = WORKLOADS[workload_name]
    cam = scenes.CONFIG_CAMERAS[wl["camera"]]
    w, h = wl["w"], wl["h"]
    tmp = tempfile.mkdtemp()
    obj, ntris = getattr(scenes, wl["scene"])(os.path.join(tmp, "bench_%s.obj" % rank_tag), **wl["kw"])
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
    st = ses.settings(w, h, wl["spp"], max_path=wl["max_path"])
    image = lib.Raylib_CreateImage(w, h)          # what a front-end hands to Raylib_Render
    stats = binding.Stats()
    acc = new_acc()
    for _ in range(warmup):
        lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, image)
    import torch

    def fence():
        for d in range(torch.cuda.device_count()):
            torch.cuda.synchronize(d)

    lib.RaylibAMD_GetLastStats(C.byref(stats))    # (waits for what the warm-up left in flight)
    fence()
    per_step = []
    t0 = time.perf_counter()
    if n_gpus == 1:
        for _ in range(steps):
            ts = time.perf_counter()
            lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, image)     # the boundary itself (synchronous); the frame stays in HBM
            per_step.append(time.perf_counter() - ts)
            lib.RaylibAMD_GetLastStats(C.byref(stats))
            accumulate(acc, stats, binding)
        fence()
        elapsed = time.perf_counter() - t0
    else:
        # several ranks behind the call: Raylib_Render returns with frame i in flight and frame i + 1 is enqueued behind it (rl_runtime.inl
        # RenderMulti), so nothing asks for stats or pixels inside the timed region; the K frames are identical (same seed, same scene), the
        # last one's counters stand for each
        for _ in range(steps):
            lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, image)
        lib.RaylibAMD_GetLastStats(C.byref(stats))                            # waits for the last frame (every rank's stream, the gather, the assembly)
        fence()
        elapsed = time.perf_counter() - t0
        for _ in range(steps):
            accumulate(acc, stats, binding)
    last = stats.as_dict()
    # ---- untimed from here on -------------------------------------------------------------------------------------------------
    host = np.zeros(w * h * 3, np.float32)
    lib.Raylib_DumpImageData(image, host.ctypes.data_as(C.POINTER(C.c_float)))
    frame_check = golden_windows(workload_name, host.reshape(h, w, 3))
    # what a front-end observes: the render plus the read-back of the frame (the reference's Raylib_Render returns with the pixels
    # in host memory, render/renderer.cc:292-356; here the first reader pays the 33 MB copy)
    k2 = max(1, min(steps, 10))
    t1 = time.perf_counter()
    for _ in range(k2):
        lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, image)
        lib.Raylib_DumpImageData(image, host.ctypes.data_as(C.POINTER(C.c_float)))
    render_plus_dump = (time.perf_counter() - t1) / k2
    boundary = {"timed_entry": "Raylib_Render", "raylib_render_ms_per_step": elapsed / steps * 1e3,
                "render_plus_dump_ms_per_step": render_plus_dump * 1e3}
    if want_device_entry and n_gpus == 1:
        dev = torch.zeros(w * h * 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(k2):
            if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, C.c_void_p(dev.data_ptr())) != 1:
                raise SystemExit("RaylibAMD_RenderDevice failed")
        torch.cuda.synchronize()
        boundary["render_device_ms_per_step"] = (time.perf_counter() - t2) / k2 * 1e3
    tree_walk = None
    if acc.get("tree_width") == 0:
        # the workload's algorithmic bytes: the same frame on the BVH4 walk (the leaf-list kernel reads every leaf box per ray, from LDS); with several ranks
        # behind the call the counters are the ranks' sum and the launch count one rank's, as in the timed frames
        keep = os.environ.get("RAYLIB_LEAF_LIST")
        lib.RaylibAMD_GetLastStats(C.byref(stats))     # (nothing of the frames above is still in flight when the switch changes)
        os.environ["RAYLIB_LEAF_LIST"] = "0"
        lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, image)
        lib.RaylibAMD_GetLastStats(C.byref(stats))
        if keep is None:
            del os.environ["RAYLIB_LEAF_LIST"]
        else:
            os.environ["RAYLIB_LEAF_LIST"] = keep
        tree_walk = {"bytes_per_launch": binding.algorithmic_bytes(stats) / max(1, stats.traceLaunches), "launch_ms": stats.traceKernelMs / max(1, stats.traceLaunches)}
    lib.Raylib_DestroyImage(image)
    ses.close()
    return dict(acc=acc, elapsed=elapsed, per_step=per_step, frame_check=frame_check, boundary=boundary, tree_walk=tree_walk,
                ntris=ntris, last_stats=last, wl=wl, cam=cam)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=HEADLINE, choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the second workload (a memory-bound megakernel) reported under \"extra\"")
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1:
        raise SystemExit("--gpus and --steps must be positive")

    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if distributed else 1
    if distributed and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: under torch.distributed.run --gpus must equal --nproc-per-node" % (args.gpus, world))
    import torch
    from raylib_amd import binding, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if distributed:
        return process_mode(args, rank, local_rank, world, torch, binding, scenes)

    # ---- library mode ---------------------------------------------------------------------------------------------------------
    n = args.gpus
    env_n = os.environ.get("RAYLIB_NUM_GPUS")
    if env_n is not None and int(env_n) != n:
        raise SystemExit("RAYLIB_NUM_GPUS=%s in the environment but --gpus %d: the line would attribute the rays to the wrong number of GPUs" % (env_n, n))
    gpu_map = os.environ.get("RAYLIB_GPU_MAP")
    visible = torch.cuda.device_count()
    if n > visible and not gpu_map:
        raise SystemExit("--gpus %d but %d device(s) visible: refusing to measure fewer GPUs than asked for "
                         "(RAYLIB_GPU_MAP=0,0,.. puts several logical ranks on one device, for tests)" % (n, visible))
    if n > 1:
        os.environ["RAYLIB_NUM_GPUS"] = str(n)
    lib = binding.load()
    if lib.Raylib_Initialize() != 1:
        raise SystemExit("Raylib_Initialize failed")
    lib.RaylibAMD_SetSeed(1)
    build_id = lib.RaylibAMD_BuildId().decode()

    r = library_run(lib, binding, scenes, args.workload, args.steps, args.warmup, n)
    ranks = int(r["last_stats"]["ranks"])
    if ranks != n:
        raise SystemExit("asked for %d GPU(s) but the library rendered on %d rank(s)" % (n, ranks))
    acc, elapsed, wl = r["acc"], r["elapsed"], r["wl"]
    out = {
        "metric": "Mrays/sec (primary+secondary) + frame time, Cornell Box 1080p 64spp",
        "value": acc["rays"] / elapsed / 1e6,
        "unit": "Mrays/s",
        "n_gpus": ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_spread": step_times(r["per_step"]),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": args.workload, "mode": "library (Raylib_Render; RAYLIB_NUM_GPUS=%d)" % n, "scene_triangles": r["ntris"],
                   "width": wl["w"], "height": wl["h"], "spp": wl["spp"], "max_path_length": wl["max_path"], "seed": 1,
                   "tiling": "8x8 cells round-robin over %d rank(s)" % ranks, "rays_per_step": acc["rays"] / args.steps,
                   "camera_samples_per_step": acc["samples"] / args.steps, "work": work_block(acc, args.steps),
                   "frame_check": r["frame_check"], "boundary": r["boundary"],
                   "build_id": build_id, "timed_region_s": elapsed},
        "roofline": roofline_block(args.workload, acc, ranks, build_id, r["tree_walk"], gpus_in_acc=ranks,
                                   frame_samples_per_launch=acc["samples"] / max(1, acc["launches"])),
    }
    # the two N = 1 figures a scaling curve can start from: this mode's (Raylib_Render, what `python bench.py --gpus 1` -- the driver's
    # N = 1 run -- times) and process mode's (RaylibAMD_RenderDevice into a caller's device buffer + one gather, what torch.distributed.run launches for N > 1)
    out["config"]["n1_reference"] = {
        "library_mode_ms_per_step": out["ms_per_step"] if ranks == 1 else None,
        "process_mode_entry_ms_per_step": r["boundary"].get("render_device_ms_per_step"),
        "note": "the driver's N = 1 line is library mode (Raylib_Render); its N > 1 lines are process mode (RaylibAMD_RenderDevice per rank + one RCCL gather per frame): "
                "a SCALE curve compares those two, so process mode's own one-rank entry time stands here next to it"}
    if ranks > 1 or gpu_map:
        s = r["last_stats"]
        out["multi_gpu"] = {"ranks": ranks, "devices": s["devices"], "gpu_map": gpu_map, "gather": s["gatherMode"], "rccl_comm_size": s["rcclCommSize"],
                            "gather_ms": s["gatherMs"], "scatter_ms": s["scatterMs"], "rank_kernel_ms": s["rankKernelMs"], "rank_trace_ms": s["rankTraceMs"],
                            "note": "last timed frame; gather_ms = rank 0's stream from the end of its own kernels until every rank's cells are on its device"}
    if ranks == 1 and not args.no_extra and args.workload == HEADLINE:
        # further objects in the same invocation: the configs[2]-sized scene, whose megakernel (the pool schedule) waits on memory -- seen from outside
        # (SURVEY 8d's stand-in camera: 89 % of that frame is empty sky and is not traced) and from INSIDE (every pixel is geometry, nothing is dropped)
        k = max(3, min(20, args.steps // 5))
        for key, name in (("extra", EXTRA), ("extra_interior", EXTRA2), ("extra_textured", EXTRA3)):
            e = library_run(lib, binding, scenes, name, k, 1, 1, rank_tag="x", want_device_entry=False)
            ea = e["acc"]
            out[key] = {"workload": name, "value": ea["rays"] / e["elapsed"] / 1e6, "unit": "Mrays/s", "steps": k, "warmup": 1,
                        "ms_per_step": e["elapsed"] / k * 1e3, "ms_per_step_spread": step_times(e["per_step"]), "scene_triangles": e["ntris"],
                        "spp": e["wl"]["spp"], "work": work_block(ea, k), "frame_check": e["frame_check"], "boundary": e["boundary"],
                        "roofline": roofline_block(name, ea, 1, build_id, e["tree_walk"])}
    if ranks == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(wl, r["cam"], (acc["rays"] + acc["culled_rays"]) / max(1.0, acc["samples"] + acc["culled_samples"]), elapsed / args.steps)
    print(json.dumps(out))
    sys.stdout.flush()


def process_mode(args, rank, local_rank, world, torch, binding, scenes):
    """One process per GPU (what the driver launches): RaylibAMD_RenderDevice per rank, one RCCL gather per frame."""
    import torch.distributed as dist
    from raylib_amd import tiling
    # BENCH_SHARE_GPU=1 (test aid, 1-GPU box): several ranks on the one device, the gather staged through the host over gloo --
    # RCCL refuses two ranks per device.  It exercises the N > 1 frame assembly, not its speed.
    share = os.environ.get("BENCH_SHARE_GPU", "0") == "1"
    visible = torch.cuda.device_count()
    if world > visible and not share:
        raise SystemExit("WORLD_SIZE=%d but %d device(s) visible (BENCH_SHARE_GPU=1 shares one device, for tests)" % (world, visible))
    if os.environ.get("RAYLIB_NUM_GPUS", "1") != "1":
        raise SystemExit("RAYLIB_NUM_GPUS must not be set in process mode: every process drives one GPU")
    # one visible device per rank (a launcher that masks devices per process) or all of the node's devices visible to every rank
    device_index = local_rank % max(1, visible)
    os.environ["RAYLIB_DEVICE"] = str(device_index)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if share:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    lib = binding.load()
    if lib.Raylib_Initialize() != 1:
        raise SystemExit("Raylib_Initialize failed")
    lib.RaylibAMD_SetSeed(1)
    build_id = lib.RaylibAMD_BuildId().decode()

    wl = WORKLOADS[args.workload]
    cam = scenes.CONFIG_CAMERAS[wl["camera"]]
    w, h = wl["w"], wl["h"]
    tmp = tempfile.mkdtemp()
    obj, ntris = getattr(scenes, wl["scene"])(os.path.join(tmp, "bench_r%d.obj" % rank), **wl["kw"])
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
    st = ses.settings(w, h, wl["spp"], max_path=wl["max_path"])

    # device buffers: this rank's cells (padded to equal size for the gather) and, on rank 0, the frame
    pad_floats = max(tiling.padded_cells(w, h, world) * 64 * 4, w * h * 4 if world == 1 else 0)
    # two send buffers, used alternately: the gather of frame i (RCCL's stream) may still be reading its buffer while the
    # library (its own stream) already renders frame i + 1 into the other one
    mines = [torch.zeros(pad_floats, dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.zeros(pad_floats, dtype=torch.float32, device=dev) for _ in range(world)] if rank == 0 else None
    frame = torch.zeros(h * w, 4, dtype=torch.float32, device=dev) if rank == 0 else None
    plan = tiling.torch_scatter_plan(w, h, world, dev) if rank == 0 else None

    stats = binding.Stats()
    acc = new_acc()
    pending = [None, None]
    frame_no = [0]

    def step(record):
        k = frame_no[0] & 1
        frame_no[0] += 1
        out = mines[k]
        if pending[k] is not None:
            pending[k].wait()                             # the gather that read this buffer two frames ago ...
            torch.cuda.current_stream().synchronize()     # ... is complete before the library's stream overwrites it (returns at once in steady state)
        if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, rank, world, C.c_void_p(out.data_ptr())) != 1:
            raise SystemExit("RaylibAMD_RenderDevice failed")
        if share:
            host = out.cpu()
            host_list = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, host_list, dst=0)
            if rank == 0:
                for dst_t, src_t in zip(gathered, host_list):
                    dst_t.copy_(src_t)
        else:
            pending[k] = dist.gather(out, gathered, dst=0, async_op=True)   # one RCCL gather per frame
        if rank == 0:
            if pending[k] is not None:
                pending[k].wait()                         # stream-level: the assembly below is ordered after the gather
            if world == 1:
                frame.copy_(gathered[0][: h * w * 4].view(h * w, 4))   # one rank renders the row-major frame directly
            else:
                frame[plan[1]] = torch.stack(gathered).reshape(-1, 4)[plan[0]]
        if record:
            lib.RaylibAMD_GetLastStats(C.byref(stats))
            accumulate(acc, stats, binding)

    for _ in range(args.warmup):
        step(False)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    fence()
    per_step = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step(True)
        per_step.append(time.perf_counter() - ts)
    fence()
    elapsed = time.perf_counter() - t0

    red_dev = torch.device("cpu") if share else dev
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    tot = torch.tensor([float(acc["rays"]), float(acc["samples"]), float(acc["culled_rays"]), float(acc["culled_samples"])], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    total_rays, total_samples = float(tot[0].item()), float(tot[1].item())
    total_frame_rays, total_frame_samples = total_rays + float(tot[2].item()), total_samples + float(tot[3].item())
    # every rank's megakernel time of the last frame, for attributing a scaling loss
    km = torch.zeros(world, dtype=torch.float64, device=red_dev)
    km[rank] = stats.traceKernelMs
    dist.all_reduce(km, op=dist.ReduceOp.SUM)

    # outside the timed region: this rank's share once more on the tree walk, where the kernel that ran read its scene from LDS (the leaf list) -- the workload's
    # algorithmic bytes are the tree walk's, whatever schedule ran
    tree_walk = None
    if acc.get("tree_width") == 0:
        keep = os.environ.get("RAYLIB_LEAF_LIST")
        os.environ["RAYLIB_LEAF_LIST"] = "0"
        torch.cuda.synchronize()
        if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, rank, world, C.c_void_p(mines[0].data_ptr())) == 1:
            lib.RaylibAMD_GetLastStats(C.byref(stats))
            tree_walk = {"bytes_per_launch": binding.algorithmic_bytes(stats) / max(1, stats.traceLaunches), "launch_ms": stats.traceKernelMs / max(1, stats.traceLaunches)}
        if keep is None:
            del os.environ["RAYLIB_LEAF_LIST"]
        else:
            os.environ["RAYLIB_LEAF_LIST"] = keep
    frame_check = None
    if rank == 0:
        # outside the timed region: the frame assembled from the ranks' cells against this rank's own render of the whole frame
        torch.cuda.synchronize()
        whole = torch.zeros(h * w * 4, dtype=torch.float32, device=dev)
        if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, C.c_void_p(whole.data_ptr())) == 1:
            same = bool(torch.equal(whole.view(torch.int32), frame.reshape(-1).view(torch.int32)))
            frame_check = "assembled frame bit-identical to a one-GPU render" if same else "MISMATCH between the assembled frame and a one-GPU render"
            gw = golden_windows(args.workload, frame.reshape(h, w, 4).cpu().numpy())
            if gw is not None:
                frame_check += "; " + gw
        out = {
            "metric": "Mrays/sec (primary+secondary) + frame time, Cornell Box 1080p 64spp",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_spread": step_times(per_step),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "mode": "one process per GPU (RaylibAMD_RenderDevice + one RCCL gather per frame)", "scene_triangles": ntris,
                       "width": w, "height": h, "spp": wl["spp"], "max_path_length": wl["max_path"], "seed": 1,
                       "tiling": "8x8 cells round-robin over %d rank(s)" % world, "rays_per_step": total_rays / args.steps,
                       "camera_samples_per_step": total_samples / args.steps,
                       "work": {"rays_executed_per_step": total_rays / args.steps, "camera_samples_executed_per_step": total_samples / args.steps,
                                "rays_accounted_not_traced_per_step": (total_frame_rays - total_rays) / args.steps,
                                "camera_samples_not_traced_per_step": (total_frame_samples - total_samples) / args.steps},
                       "frame_check": frame_check,
                       "boundary": {"timed_entry": "RaylibAMD_RenderDevice + torch.distributed gather", "ms_per_step": elapsed / args.steps * 1e3},
                       "n1_reference": {"note": "the driver's N = 1 line is library mode (`python bench.py --gpus 1`: Raylib_Render); this line is process mode "
                                                "(RaylibAMD_RenderDevice per rank + one gather per frame), whose own one-rank time is `boundary.render_device_ms_per_step` of the N = 1 line "
                                                "(or this line at WORLD_SIZE = 1)"},
                       "build_id": build_id, "timed_region_s": elapsed},
            # rank 0's own launches (the other ranks run the same kernel on their share of the cells: rank_trace_ms below)
            "roofline": roofline_block(args.workload, acc, world, build_id, tree_walk, gpus_in_acc=1,
                                       frame_samples_per_launch=total_samples / max(1, acc["launches"])),
            "multi_gpu": {"ranks": world, "devices": min(world, visible), "gather": "gloo via host (BENCH_SHARE_GPU)" if share else "rccl (torch.distributed gather)",
                          "rank_trace_ms": [float(x) for x in km.tolist()], "note": "rank_trace_ms: every rank's megakernel time in the last timed frame"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, cam, total_frame_rays / max(1.0, total_frame_samples), elapsed / args.steps)
        print(json.dumps(out))
        sys.stdout.flush()
    ses.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
