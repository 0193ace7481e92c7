#!/usr/bin/env python3
"""bench.py -- Mrays/s of the hot path (Raylib_Render's megakernel) on N MI355X GPUs.

A "step" is one full frame of BASELINE.json's configs[1]: synthetic Cornell box
(36 triangles), 1920x1080, 64 spp, maxPathLength 5, fixed RNG seed, scene resident
in HBM.  At N = 1 the timed call is Raylib_Render itself -- the reference's entry point
(raylib/raylib.cc:231-239), frame left resident in HBM -- and the frame it produced is
checked, outside the timed region, against windows rendered by the reference build
(tests/golden/bench_windows.npz).  With N ranks (one process per GPU) the 8x8-pixel cells
of the frame are dealt round-robin to the ranks, each rank renders its cells into device
memory (RaylibAMD_RenderDevice), and one RCCL gather per frame brings them to rank 0
(strong scaling: total work fixed).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "software-raytracing_amd"))
os.environ.setdefault("RAYLIB_QUIET", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
# wave64 VALU instructions per cycle per SIMD-32 at saturation: 0.5 (2 cycles each; MI355X_MICROARCH.md constants table, and measured on
# the bench box by tools/valu_calib.hip: 944 G wave-instructions/s over 1024 SIMDs at the ~1.9 GHz the chip holds under that load)
VALU_CYCLES_PER_WAVE_INST = 2.0
NUM_SIMDS = 256 * 4

WORKLOADS = {
    # BASELINE.json configs[1]
    "cornell_1080p_64spp": dict(scene="cornell", kw={}, w=1920, h=1080, spp=64, max_path=5, camera="cornell"),
    # configs[2]-sized stress (parity-test size by default; selectable for profiling)
    "breakfast_300k_1080p_128spp": dict(scene="cornell", kw=dict(tess=91, displace_fraction=0.2), w=1920, h=1080, spp=128, max_path=5, camera="breakfast"),
}


def cpu_baseline(workload, cam, gpu_rays_per_sample):
    """The reference's own CPU path (Renderer::RenderScene + its ThreadPool, built in place into
    oracle/_ref/libref_native.so) timed on this host's cores on a bounded sample of the same
    workload.  Falls back to the oracle restatement (kind "port") when the prebuilt binary is absent."""
    sys.path.insert(0, ROOT)
    from oracle import ffi, objflat           # checker only: never on the product path
    from raylib_amd import scenes
    orc = ffi.load_oracle()
    tmp = tempfile.mkdtemp()
    obj, _ = getattr(scenes, workload["scene"])(os.path.join(tmp, "cpu.obj"), **workload["kw"])
    flat = objflat.load_obj(obj, orc, sun_illuminance=cam["sun"], sun_direction=cam["sun_dir"])
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
    return {"value": samples * gpu_rays_per_sample / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": kind,
            "seconds": dt, "camera_samples_per_s": samples / dt,
            "sample": "%dx%d at %d spp of the same scene/camera (%.1f%% of the step's camera samples), all %d host threads; "
                      "rays = camera samples x the GPU run's measured rays per camera sample (%.4f), since the reference has no ray counter"
                      % (w, h, spp, 100.0 * spp / workload["spp"], cores, gpu_rays_per_sample)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_64spp", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # under torch.distributed.run the gather path is used even with one rank (so it can be exercised on a 1-GPU box)
    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    from raylib_amd import binding, scenes, tiling

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    # BENCH_SHARE_GPU=1 (test aid, 1-GPU box): several ranks on the one device, the gather staged through the host over gloo --
    # RCCL refuses two ranks per device.  It exercises the N > 1 frame assembly, not its speed.
    share = os.environ.get("BENCH_SHARE_GPU", "0") == "1"
    # one visible device per rank (a launcher that masks devices per process) or all of the node's devices visible to every rank
    device_index = local_rank % max(1, torch.cuda.device_count())
    os.environ["RAYLIB_DEVICE"] = str(device_index)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if distributed:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm

    lib = binding.load()
    if lib.Raylib_Initialize() != 1:
        raise SystemExit("Raylib_Initialize failed")
    lib.RaylibAMD_SetSeed(1)

    wl = WORKLOADS[args.workload]
    cam = scenes.CONFIG_CAMERAS[wl["camera"]]
    w, h = wl["w"], wl["h"]
    tmp = tempfile.mkdtemp()
    obj, ntris = getattr(scenes, wl["scene"])(os.path.join(tmp, "bench_r%d.obj" % rank), **wl["kw"])
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
    st = ses.settings(w, h, wl["spp"], max_path=wl["max_path"])

    # device buffers: this rank's cells (padded to equal size for the gather) and, on rank 0, the frame
    if distributed:
        pad_floats = max(tiling.padded_cells(w, h, world) * 64 * 4, w * h * 4 if world == 1 else 0)
        # two send buffers, used alternately: the gather of frame i (RCCL's stream) may still be reading its buffer while the
        # library (its own stream) already renders frame i + 1 into the other one
        mines = [torch.zeros(pad_floats, dtype=torch.float32, device=dev) for _ in range(2)]
        mine = mines[0]
        gathered = [torch.zeros(pad_floats, dtype=torch.float32, device=dev) for _ in range(world)] if rank == 0 else None
        frame = torch.zeros(h * w, 4, dtype=torch.float32, device=dev) if rank == 0 else None
        plan = tiling.torch_scatter_plan(w, h, world, dev) if rank == 0 else None
    else:
        mine = torch.zeros(w * h * 4, dtype=torch.float32, device=dev)
        boundary_image = lib.Raylib_CreateImage(w, h)       # what a front-end hands to Raylib_Render

    stats = binding.Stats()
    acc = dict(rays=0, trace_ms=0.0, launches=0, bytes=0, nodes=0, tris=0, shaded=0, texels=0, samples=0, kernel_ms=0.0)

    pending = [None, None]
    frame_no = [0]

    def step(record):
        out = mine
        if distributed:
            k = frame_no[0] & 1
            frame_no[0] += 1
            out = mines[k]
            if pending[k] is not None:
                pending[k].wait()                             # the gather that read this buffer two frames ago ...
                torch.cuda.current_stream().synchronize()     # ... is complete before the library's stream overwrites it (returns at once in steady state)
        if distributed:
            ok = lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, rank, world, C.c_void_p(out.data_ptr()))
            if ok != 1:
                raise SystemExit("RaylibAMD_RenderDevice failed")
        else:
            lib.Raylib_Render(C.byref(st), ses.scene, ses.camera, boundary_image)    # the boundary itself; the frame stays in HBM
        if distributed:
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
            acc["rays"] += stats.rays; acc["trace_ms"] += stats.traceKernelMs; acc["launches"] += stats.traceLaunches
            acc["kernel_ms"] += stats.kernelMs
            acc["bytes"] += binding.algorithmic_bytes(stats)
            acc["nodes"] += stats.nodesVisited; acc["tris"] += stats.trisTested; acc["shaded"] += stats.shadedHits
            acc["texels"] += stats.texFetches; acc["samples"] += stats.cameraSamples

    for _ in range(args.warmup):
        step(False)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0

    if distributed:
        red_dev = torch.device("cpu") if share else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(acc["rays"]), float(acc["samples"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_rays, total_samples = float(tot[0].item()), float(tot[1].item())
    else:
        total_rays, total_samples = float(acc["rays"]), float(acc["samples"])

    def golden_windows(frame_hw4):
        """The timed frame against windows of the same frame rendered by the REAL reference build (tests/golden/gen_golden.py ->
        bench_windows.npz: data; only windows in which no sample met two surfaces at exactly the same t).  Bit for bit."""
        import numpy as np
        path = os.path.join(ROOT, "tests", "golden", "bench_windows.npz")
        if not os.path.exists(path):
            return None
        g = np.load(path)
        if args.workload + "_pos" not in g.files:
            return None
        pos, px = g[args.workload + "_pos"], g[args.workload + "_px"]
        bad = 0
        for (x0, y0), want in zip(pos, px):
            got = frame_hw4[y0:y0 + 16, x0:x0 + 16]
            bad += int((np.ascontiguousarray(got[..., :3]).view(np.uint32) != np.ascontiguousarray(want[..., :3]).view(np.uint32)).any(-1).sum())
        n = len(pos) * 256
        return ("%d / %d pixels of %d reference-rendered windows bit-identical" % (n - bad, n, len(pos))) if bad == 0 else \
               ("MISMATCH: %d of %d pixels of the reference-rendered windows differ" % (bad, n))

    frame_check = None
    boundary = None
    if not distributed:
        # outside the timed region: (1) the frame the LAST timed Raylib_Render call produced, read back through the reference's own
        # accessor and compared with the reference build's pixels; (2) the same K steps through the device-pointer entry, for comparison
        import numpy as np
        host = np.zeros(w * h * 3, np.float32)
        lib.Raylib_DumpImageData(boundary_image, host.ctypes.data_as(C.POINTER(C.c_float)))
        frame_check = golden_windows(host.reshape(h, w, 3))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, C.c_void_p(mine.data_ptr())) != 1:
                raise SystemExit("RaylibAMD_RenderDevice failed")
        torch.cuda.synchronize()
        boundary = {"timed_entry": "Raylib_Render", "raylib_render_ms_per_step": elapsed / args.steps * 1e3,
                    "render_device_ms_per_step": (time.perf_counter() - t1) / args.steps * 1e3}
    if distributed and rank == 0:
        # outside the timed region: the frame assembled from the ranks' cells against this rank's own render of the whole frame
        torch.cuda.synchronize()
        whole = torch.zeros(h * w * 4, dtype=torch.float32, device=dev)
        if lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, C.c_void_p(whole.data_ptr())) == 1:
            same = bool(torch.equal(whole.view(torch.int32), frame.reshape(-1).view(torch.int32)))
            frame_check = "assembled frame bit-identical to a one-GPU render" if same else "MISMATCH between the assembled frame and a one-GPU render"
            gw = golden_windows(frame.reshape(h, w, 4).cpu().numpy())
            if gw is not None:
                frame_check += "; " + gw
        lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, rank, world, C.c_void_p(mines[0].data_ptr()))   # stats of a timed-style call again

    if rank == 0:
        launches = max(1, acc["launches"])
        avg_launch_ms = acc["trace_ms"] / launches
        bytes_per_launch = acc["bytes"] / launches
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        # Hardware counters cannot be read from inside this process: `traffic` and the VALU figures are REPLAYED from the committed
        # rocprofv3 --pmc passes of this same command (tools/pmc_profile.sh -> tools/pmc_traffic.py -> profiles/pmc_traffic.json),
        # rescaled to this run's launch time where they are rates.  traffic_source says so in the line.
        traffic = None
        valu = None
        hbm_measured = None
        bound = "unknown (no PMC passes committed for this workload)"
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc)).get(args.workload)
                if rec and world == 1:
                    traffic = rec["hbm_bytes_per_launch"]
                    traffic_source = "replayed from profiles/pmc_traffic.json (%s)" % rec.get("round", "rocprofv3 --pmc passes of this workload")
                    hbm_measured = traffic / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else None
                    # VALU issue: wave-level VALU instructions x 2 cycles / (SIMDs x the launch's cycles), and the share of lanes doing work
                    cycles = rec.get("cycles_per_launch")
                    busy = (VALU_CYCLES_PER_WAVE_INST * rec["valu_insts_per_launch"] / NUM_SIMDS / cycles) if cycles else rec.get("valu_busy_fraction")
                    valu = {"busy_fraction": busy, "lane_utilisation": rec.get("valu_lane_utilisation"),
                            "insts_per_launch": rec.get("valu_insts_per_launch"), "cycles_per_wave_inst": VALU_CYCLES_PER_WAVE_INST,
                            "wave_wait_fraction": rec.get("wave_wait_fraction"), "waves_per_simd": rec.get("waves_per_simd")}
                    hbm_frac = (hbm_measured or 0.0) / HBM_PEAK_GBS
                    # what binds: the larger of the two utilisations -- and neither when both are low (then the waves are waiting on
                    # each other's latencies: too few of them per SIMD to fill the issue slots)
                    if hbm_frac >= 0.6 and hbm_frac >= (busy or 0.0):
                        bound = "hbm"
                    elif (busy or 0.0) >= 0.75:
                        bound = "valu-issue"
                    else:
                        bound = "latency (VALU issue %.0f %%, HBM %.0f %% of peak: neither saturated)" % (100 * (busy or 0.0), 100 * hbm_frac)
            except Exception:
                traffic = None
        # achieved / peak / unit / frac describe the resource the counters show nearest its roof:
        #   HBM        -- ALGORITHMIC bytes per launch over the launch time against the HBM peak, as SURVEY 8(d) defines the figure;
        #   VALU issue -- wave-level VALU instructions per second against 1024 SIMDs x (1 instruction / 2 cycles) x the launch's clock
        #                 (tools/valu_calib.hip measures that 0.5 per cycle on this part): what binds a scene that lives in LDS or L2,
        #                 where the algorithmic bytes never reach HBM and "bytes / HBM peak" can exceed 1 without meaning anything.
        # The algorithmic figure is kept in every line (algorithmic_gbs, algorithmic_frac_of_hbm_peak) next to what the memory system
        # really moved (hbm_measured_*); `bound` says whether anything is saturated at all.
        algorithmic = {"algorithmic_bytes_per_launch": bytes_per_launch, "algorithmic_gbs": achieved, "algorithmic_frac_of_hbm_peak": achieved / HBM_PEAK_GBS}
        busy_now = valu["busy_fraction"] if valu else None
        hbm_now = (hbm_measured / HBM_PEAK_GBS) if hbm_measured else None
        if busy_now is not None and busy_now >= (hbm_now or 0.0) and avg_launch_ms > 0:
            a = valu["insts_per_launch"] / (avg_launch_ms * 1e-3) / 1e9
            head = {"achieved": a, "peak": a / busy_now, "unit": "Gwave-inst/s", "frac": busy_now, "resource": "VALU issue"}
        else:
            head = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "resource": "HBM (algorithmic bytes)"}
        roofline = {"bound": bound}
        roofline.update(head)
        roofline.update({"traffic": traffic, "traffic_source": traffic_source,
                         "hbm_measured_gbs": hbm_measured, "hbm_measured_frac": hbm_now,
                         "kernel": "k_trace" if stats.pathsPerWave <= 64 else "k_trace_pool", "paths_per_wave": int(stats.pathsPerWave), "avg_launch_ms": avg_launch_ms, "launches": acc["launches"]})
        roofline.update(algorithmic)
        roofline.update({"valu": valu,
                         "note": "rank 0's launches; algorithmic bytes = 64 B x (BVH node / leaf-list records + triangle records + shading records) + 16 B x (texels + pixels)"})
        out = {
            "metric": "Mrays/sec (primary+secondary) + frame time, Cornell Box 1080p 64spp",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "scene_triangles": ntris, "width": w, "height": h, "spp": wl["spp"],
                       "max_path_length": wl["max_path"], "seed": 1, "tiling": "8x8 cells round-robin over %d rank(s)" % world,
                       "rays_per_step": total_rays / args.steps, "camera_samples_per_step": total_samples / args.steps,
                       "frame_check": frame_check, "boundary": boundary},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, cam, total_rays / max(1.0, total_samples))
        print(json.dumps(out))
        sys.stdout.flush()

    if not distributed:
        lib.Raylib_DestroyImage(boundary_image)
    ses.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
