import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RAYLIB_QUIET", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _helpers():
    import helpers
    return helpers


@pytest.fixture(scope="session")
def oracle():
    import helpers
    return helpers.ffi.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference build (oracle/_ref); None where it was never built."""
    import helpers
    return helpers.ffi.load_ref(True)


@pytest.fixture(scope="session")
def lib():
    import helpers  # noqa: F401  (sets sys.path)
    from raylib_amd import binding
    return binding.load()


@pytest.fixture(scope="session")
def gpu_lib(lib):
    """libraylib.so initialised on a real device; fails loudly when there is none."""
    assert lib.Raylib_Initialize() == 1, "Raylib_Initialize failed: no HIP device -- GPU tests cannot run on a fallback"
    lib.RaylibAMD_SetSeed(1)
    return lib


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return tmp_path_factory.mktemp("scenes")


# ---- scenes shared by the GPU test files (session scope: one load, whichever file asks first) ------------------------------------------
@pytest.fixture(scope="session")
def sessions(gpu_lib, workdir):
    s = {name: _helpers().session_for_case(gpu_lib, name, workdir) for name in _helpers().CASES}
    yield s
    for v in s.values():
        v.close()


@pytest.fixture(scope="session")
def full_size(gpu_lib, workdir):
    from raylib_amd import binding
    obj, c = _helpers().build_case("cornell", workdir)
    ses = binding.SceneSession(gpu_lib, obj, c["origin"], c["look_at"], 45.0, 1920 / 1080)
    img = ses.render(1920, 1080, 64)
    yield ses, img, ses.stats().as_dict()
    ses.close()


@pytest.fixture(scope="session")
def mid_scene(gpu_lib, workdir):
    """Tessellated room with displaced triangles (about 21 k triangles, sun): BVH deeper than 16 -> pool schedule by default."""
    from raylib_amd import binding
    d = os.path.join(str(workdir), "mid"); os.makedirs(d, exist_ok=True)
    obj, n = _helpers().scenes.cornell(os.path.join(d, "mid.obj"), tess=24, displace_fraction=0.2)
    ses = binding.SceneSession(gpu_lib, obj, (0, 1, 5), (0, 1, -1), 60.0, 96 / 64, sun=(20, 20, 20), sun_dir=(-1.0, -1.0, 0.0))
    yield ses, obj, n
    ses.close()


@pytest.fixture(scope="session")
def config2_scene(gpu_lib, oracle, workdir):
    """BASELINE configs[2]'s stand-in (tessellated room, 298 116 triangles, a fifth of them displaced, sun) through the OBJ path, loaded once for the contract tier's
    whole-frame comparison and for tests/test_gpu_01_configs.py; the product must have loaded exactly the generator's arrays.  Yields (session, flat scene, obj path)."""
    from raylib_amd import binding
    h = _helpers()
    cam = h.scenes.CONFIG_CAMERAS["breakfast"]
    d = os.path.join(str(workdir), "config2"); os.makedirs(d, exist_ok=True)
    obj, flat = h.big_scene(os.path.join(d, "c2.obj"), h.scenes.cornell_objects(), h.scenes.CORNELL_MTL, oracle, 91, 0.2, sun=cam["sun"], sun_dir=cam["sun_dir"])
    assert len(flat.triangles) == 298116
    ses = binding.SceneSession(gpu_lib, obj, cam["origin"], cam["look_at"], cam["fov"], 1920 / 1080, sun=cam["sun"], sun_dir=cam["sun_dir"])
    tris, mats = ses.export_flat()
    assert tris.tobytes() == flat.triangles.tobytes(), "the product's loader and the generator's arrays disagree"
    assert mats.tobytes() == flat.materials.tobytes()
    yield ses, flat, obj
    ses.close()
