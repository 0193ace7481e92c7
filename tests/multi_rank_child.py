"""Child process of tests/test_gpu_multi_rank.py: the library reads RAYLIB_NUM_GPUS / RAYLIB_GPU_MAP / RAYLIB_GATHER* once, when it
initialises, so every rank layout needs its own process.  Renders a fixed list of frames through Raylib_Render (the reference's
own entry point, nothing rank-aware in the call) and stores them.   usage: python multi_rank_child.py <workdir> <out.npz>
or, for one frame of an OBJ scene at a BASELINE config's size (tests/test_gpu_configs.py):
python multi_rank_child.py <workdir> <out.npz> obj <file.obj> <camera name> <width> <height> <spp>"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import helpers  # noqa: E402
from raylib_amd import binding  # noqa: E402

FRAMES = [  # (case, width, height, spp, mode)
    ("cornell", 64, 64, 4, 0), ("cornell", 40, 28, 3, 0), ("cornell", 8, 8, 2, 0), ("cornell", 17, 1, 2, 0),
    ("cornell_glass_sun", 64, 64, 4, 0), ("cornell_glass_sun", 64, 64, 1, 2), ("cutout_sky", 64, 48, 4, 0), ("pbr_maps", 48, 64, 4, 0),
    ("cornell", 1920, 1080, 2, 0),
]


def render_all(lib, workdir):
    out = {}
    stats = []
    for i, (name, w, h, spp, mode) in enumerate(FRAMES):
        ses = helpers.session_for_case(lib, name, workdir)
        out["f%d" % i] = ses.render(w, h, spp, mode=mode)
        s = ses.stats()
        stats.append((s.ranks, s.cameraSamples, s.pixels, s.rays, s.gatherMode, s.devices, s.rcclCommSize))
        ses.close()
    out["stats"] = np.asarray(stats, np.int64)
    out.update(back_to_back(lib, workdir))
    out["caller_buffer"] = caller_buffer(lib, workdir)
    return out


def caller_buffer(lib, workdir):
    """RaylibAMD_RenderDevice(whole frame) into device memory of the CALLER's, read back at once with a plain hipMemcpy (the null stream does not
    wait for the library's non-blocking streams): the entry is documented as synchronous, also when the frame is assembled from several ranks
    on the gather stream (csrc/rl_runtime.inl RenderMulti, RenderRequest::callerOwnsOut).  Then the buffer is freed straight away."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]; hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    w, h = 1920, 1080
    ses = helpers.session_for_case(lib, "cornell", workdir)
    st = ses.settings(w, h, 2)
    frames = []
    for _ in range(2):                                   # twice: the second call finds the first one's slot
        dev = C.c_void_p()
        assert hip.hipMalloc(C.byref(dev), w * h * 16) == 0 and hip.hipMemset(dev, 0xff, w * h * 16) == 0
        assert lib.RaylibAMD_RenderDevice(C.byref(st), ses.scene, ses.camera, 0, 1, dev) == 1
        host = np.zeros((h, w, 4), np.float32)
        assert hip.hipMemcpy(host.ctypes.data, dev, w * h * 16, 2) == 0
        assert hip.hipFree(dev) == 0
        frames.append(host)
    s = ses.stats()
    assert s.cameraSamples + s.culledSamples == w * h * 2 and s.kernelMs > 0      # the call's own numbers, complete when it returned
    ses.close()
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    return frames[0]


def back_to_back(lib, workdir):
    """Calls that leave no time between them: with several ranks Raylib_Render returns with the frame still in flight (csrc/rl_runtime.inl
    RenderMulti), and whatever comes next -- another render into the same image, into another image, PostProcess, a destroyed image, the
    stats -- must see the finished frame.  The one-rank run of the same sequence is the reference."""
    import ctypes as C
    a = helpers.session_for_case(lib, "cornell", workdir)
    b = helpers.session_for_case(lib, "pbr_maps", workdir)
    sta, stb = a.settings(64, 64, 4), b.settings(48, 64, 4)
    img1, img2, img3 = lib.Raylib_CreateImage(64, 64), lib.Raylib_CreateImage(48, 64), lib.Raylib_CreateImage(64, 64)
    for _ in range(3):
        lib.Raylib_Render(C.byref(sta), a.scene, a.camera, img1)          # same image, nothing read in between
    lib.Raylib_Render(C.byref(stb), b.scene, b.camera, img2)              # another scene and image straight behind
    lib.Raylib_Render(C.byref(sta), a.scene, a.camera, img3)
    lib.Raylib_DestroyImage(img3)                                         # destroyed with its frame in flight
    s = binding.Stats(); lib.RaylibAMD_GetLastStats(C.byref(s))           # of the LAST call (the cornell frame into img3)
    p1 = np.zeros((64, 64, 4), np.float32); lib.RaylibAMD_DumpImageRGBA(img1, p1.ctypes.data_as(C.POINTER(C.c_float)))
    p2 = np.zeros((64, 48, 4), np.float32); lib.RaylibAMD_DumpImageRGBA(img2, p2.ctypes.data_as(C.POINTER(C.c_float)))
    lib.Raylib_Render(C.byref(sta), a.scene, a.camera, img1)
    lib.Raylib_PostProcess(img1)                                          # works on the frame that was just enqueued
    p3 = np.zeros((64, 64, 4), np.float32); lib.RaylibAMD_DumpImageRGBA(img1, p3.ctypes.data_as(C.POINTER(C.c_float)))
    lib.Raylib_DestroyImage(img1); lib.Raylib_DestroyImage(img2)
    b.close()
    lib.Raylib_Render(C.byref(sta), a.scene, a.camera, lib.Raylib_CreateImage(64, 64))   # (image leaked on purpose) scene destroyed with a frame in flight:
    a.close()
    return {"p1": p1, "p2": p2, "p3": p3, "pstats": np.asarray([s.cameraSamples, s.pixels, s.rays, int(s.kernelMs > 0), int(s.traceKernelMs > 0)], np.int64)}


def render_obj(lib, obj, camera, w, h, spp):
    from raylib_amd import scenes
    cam = scenes.CONFIG_CAMERAS[camera]
    ses = binding.SceneSession(lib, obj, cam["origin"], cam["look_at"], cam["fov"], w / h, sun=cam["sun"], sun_dir=cam["sun_dir"])
    img = ses.render(w, h, spp)
    s = ses.stats()
    ses.close()
    return {"img": img, "stats": np.asarray([s.ranks, s.cameraSamples, s.pixels, s.rays, s.gatherMode, s.devices], np.int64)}


if __name__ == "__main__":
    lib = binding.load()
    assert lib.Raylib_Initialize() == 1
    lib.RaylibAMD_SetSeed(1)
    if len(sys.argv) > 3 and sys.argv[3] == "obj":
        np.savez(sys.argv[2], **render_obj(lib, sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])))
    else:
        np.savez(sys.argv[2], **render_all(lib, sys.argv[1]))
