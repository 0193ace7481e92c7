"""Child process of tests/test_gpu_multi_rank.py: the library reads RAYLIB_NUM_GPUS / RAYLIB_GPU_MAP / RAYLIB_GATHER* once, when it
initialises, so every rank layout needs its own process.  Renders a fixed list of frames through Raylib_Render (the reference's
own entry point, nothing rank-aware in the call) and stores them.   usage: python multi_rank_child.py <workdir> <out.npz>"""
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
    return out


if __name__ == "__main__":
    lib = binding.load()
    assert lib.Raylib_Initialize() == 1
    lib.RaylibAMD_SetSeed(1)
    np.savez(sys.argv[2], **render_all(lib, sys.argv[1]))
