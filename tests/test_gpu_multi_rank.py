"""The frame split over N devices BEHIND the reference's entry point (north_star: "the image is tiled across the 8 GPUs
... behind raylib.h so the front-ends stay untouched").  RAYLIB_NUM_GPUS=N makes Raylib_Render deal the 8x8 cells to N ranks of
this process, gather them on rank 0's device and scatter them into the frame (csrc/rl_runtime.inl).  A 1-GPU box covers the whole
path by naming its one device N times in RAYLIB_GPU_MAP; the copy mechanisms between devices (RCCL grouped send / recv, peer
copies) are exercised on rank 0's own cells with RAYLIB_GATHER_SELF=1.  Every frame must equal the one-rank frame bit for bit."""
import os
import subprocess
import sys
import numpy as np
import pytest

import helpers
import multi_rank_child

pytestmark = pytest.mark.gpu

LAYOUTS = [
    dict(RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,0"),
    dict(RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,0", RAYLIB_PIPELINE="0"),
    dict(RAYLIB_NUM_GPUS="3", RAYLIB_GPU_MAP="0,0,0"),
    dict(RAYLIB_NUM_GPUS="8", RAYLIB_GPU_MAP="0,0,0,0,0,0,0,0"),
    dict(RAYLIB_NUM_GPUS="1", RAYLIB_GATHER_SELF="1", RAYLIB_GATHER="rccl"),
    dict(RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,0", RAYLIB_GATHER_SELF="1", RAYLIB_GATHER="peer"),
]
# Two PHYSICAL devices (per-device scene copies, SyncSky's peer copy, cross-device stream waits, ncclCommInitAll over two devices, grouped
# send / recv between them, peer access): run where the box has them, SKIPPED -- and not counted as covered -- where it has one.
LAYOUTS_2DEV = [
    dict(RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,1", RAYLIB_GATHER="rccl"),
    dict(RAYLIB_NUM_GPUS="2", RAYLIB_GPU_MAP="0,1", RAYLIB_GATHER="peer"),
    dict(RAYLIB_NUM_GPUS="4", RAYLIB_GPU_MAP="0,1,0,1", RAYLIB_GATHER="rccl"),
    dict(RAYLIB_NUM_GPUS="4", RAYLIB_GPU_MAP="0,1,0,1", RAYLIB_GATHER="peer"),
]
GATHER_MODE = {"none": 0, "rccl": 1, "peer": 2}


def visible_devices():
    import torch
    return torch.cuda.device_count()


@pytest.fixture(scope="module")
def one_rank_frames(gpu_lib, workdir):
    return multi_rank_child.render_all(gpu_lib, workdir)


@pytest.mark.parametrize("layout", LAYOUTS + LAYOUTS_2DEV, ids=lambda d: "-".join("%s%s" % (k.replace("RAYLIB_", "").lower(), v) for k, v in d.items()))
def test_raylib_render_over_n_ranks_is_bit_identical(layout, one_rank_frames, workdir):
    need = 1 + max(int(x) for x in layout.get("RAYLIB_GPU_MAP", "0").split(","))
    if visible_devices() < need:
        pytest.skip("needs %d physical devices, %d visible: NOT covered on this box" % (need, visible_devices()))
    out = os.path.join(str(workdir), "multi_%s.npz" % "_".join(layout.values()).replace(",", ""))
    env = dict(os.environ, **layout)
    for k in ("RAYLIB_POOL", "RAYLIB_LIB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi_rank_child.py"), str(workdir), out],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    got = np.load(out)
    n = int(layout["RAYLIB_NUM_GPUS"])
    for i in range(len(multi_rank_child.FRAMES)):
        assert helpers.same(got["f%d" % i], one_rank_frames["f%d" % i]).all(), multi_rank_child.FRAMES[i]
    # calls with nothing between them (frames in flight behind Raylib_Render): same pixels, and the last call's counters arrive with GetLastStats
    for k in ("p1", "p2", "p3"):
        assert helpers.same(got[k], one_rank_frames[k]).all(), k
    assert np.array_equal(got["pstats"], one_rank_frames["pstats"]) and got["pstats"][3] == 1 and got["pstats"][4] == 1
    # a frame rendered into a device buffer of the caller's and read back at once with a plain hipMemcpy: complete when RaylibAMD_RenderDevice returned
    assert helpers.same(got["caller_buffer"], one_rank_frames["caller_buffer"]).all() and helpers.same(got["caller_buffer"], one_rank_frames["f8"]).all()
    # the same camera samples, pixels and rays, however they were dealt; `ranks` says who rendered
    assert np.array_equal(got["stats"][:, 1:4], one_rank_frames["stats"][:, 1:4])
    assert (got["stats"][:, 0] == n).all() and (one_rank_frames["stats"][:, 0] == 1).all()
    # the gather mechanism that actually ran is the one asked for: a silent fall-back from RCCL to peer copies fails here
    remote = layout.get("RAYLIB_GATHER_SELF") == "1" or need > 1
    want_mode = GATHER_MODE[layout.get("RAYLIB_GATHER", "rccl")] if remote else 0
    assert (got["stats"][:, 4] == want_mode).all(), (got["stats"][:, 4], want_mode)
    assert (got["stats"][:, 5] == need).all()
    if want_mode == 1:
        assert (got["stats"][:, 6] == need).all()      # the communicator spans the distinct devices


def test_more_ranks_than_devices_without_a_map_fails_loudly(workdir):
    env = dict(os.environ, RAYLIB_NUM_GPUS="16")
    env.pop("RAYLIB_GPU_MAP", None)
    code = "import sys; sys.path.insert(0, %r); import helpers; from raylib_amd import binding; lib = binding.load(); sys.exit(0 if lib.Raylib_Initialize() == 0 else 1)" % os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_sky_panorama_is_read_at_render_time(gpu_lib, workdir, oracle):
    """The reference keeps the ImageHandle and reads its pixels at every miss (renderer.cc:159-176): a panorama set or changed after
    Raylib_FinalizeScene is what the next render sees, and nothing of the scene is rebuilt for it."""
    import ctypes as C
    from raylib_amd import binding
    lib = gpu_lib
    obj, c = helpers.build_case("cornell", workdir)
    ses = binding.SceneSession(lib, obj, (0, 1, 9), (0, 1, -1), 60.0, 1.0)      # far enough back to see the sky around the box
    hash0 = lib.RaylibAMD_SceneBVHHash(ses.scene)
    dark = ses.render(32, 32, 2)
    sky = helpers.scenes.sky_panorama()
    img = lib.RaylibAMD_CreateImageFromData(sky.shape[1], sky.shape[0], sky.ctypes.data_as(C.POINTER(C.c_float)))
    lib.Raylib_SetSkyPanorama(ses.scene, img)                                 # after FinalizeScene
    lit = ses.render(32, 32, 2)
    assert lit[..., :3].sum() > dark[..., :3].sum() + 1.0
    assert lib.RaylibAMD_SceneBVHHash(ses.scene) == hash0
    # against the oracle with that panorama
    flat = helpers.objflat.load_obj(obj, oracle)
    flat.textures.append(np.ascontiguousarray(sky, np.float32)); flat.sky_texture = len(flat.textures) - 1
    want = oracle.render(oracle.scene_create(flat, 1), helpers.ffi.make_camera((0, 1, 9), (0, 1, -1), 60.0, 1.0), helpers.ffi.make_settings(32, 32, 2), seed=1)
    corner = np.s_[0:6, 0:6]                                                  # pure sky pixels: no closest-hit ties there
    assert helpers.same(lit[corner], want[corner]).all()
    # the image's pixels change (a render into the same handle): the next render of the scene sees the new ones
    st = ses.settings(sky.shape[1], sky.shape[0], 1)
    other = binding.SceneSession(lib, obj, (0, 1, 4), (0, 1, -1), 45.0, sky.shape[1] / sky.shape[0])
    lib.Raylib_Render(C.byref(st), other.scene, other.camera, img)
    relit = ses.render(32, 32, 2)
    assert not helpers.same(relit, lit).all()
    # a destroyed panorama is not read (the reference would read freed memory)
    lib.Raylib_DestroyImage(img)
    assert helpers.same(ses.render(32, 32, 2), dark).all()
    other.close(); ses.close()
