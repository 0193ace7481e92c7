import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RAYLIB_QUIET", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
