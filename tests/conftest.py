import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def smpl_table():
    from anim_nerf_amd import synthetic
    return synthetic.make_smpl_table(0)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The tests load libanimnerf_hip.so (symbol surface on the CPU, everything on the GPU): (re)build it whenever a source
    or header is newer than the binary (hipcc cross-compiles gfx950 without a GPU; about 90 s, a no-op otherwise).  On a
    box without hipcc (never the case in this image) the shipped binary is used as is."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("anr_build", os.path.join(ROOT, "anim-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if os.environ.get("ANIMNERF_HIP_LIB"):
        return
    import shutil
    if os.path.exists(mod.LIB_PATH) and not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        return
    mod.build()
