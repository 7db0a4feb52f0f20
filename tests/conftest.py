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


@pytest.fixture(autouse=True)
def _collect_dead_trainers_between_tests():
    """A Trainer that captured its step holds an instantiated HIP graph (three parallel branches); a test's dead Trainers are
    reference cycles, so WITHOUT this fixture their graphs stay alive until Python's cyclic collector happens to run — by the
    end of tests/test_gpu_training.py a dozen of them at once.  On ROCm 7.2 that ends in a segmentation fault inside
    hipGraphLaunch (hip::Graph::UpdateStreams <- hip::GraphExec::Run) of whichever graph is replayed next: round 6 met it when
    the file ran as a process of its own (7 of 7 runs; never in isolation, never in the whole-suite run; with four graph
    deaths logged before the crash and none near it — it is the number of LIVE instantiated graphs, not a destruction, that
    matters; native backtrace and the bisection in profiles/r06/graph_many_live_segfault.txt).  Collecting between tests
    keeps one test's graphs from piling onto the next's.  A training process holds one Trainer, i.e. one or two graphs; a
    caller that builds Trainers in a loop should drop the old one and gc.collect() first."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
    except Exception:                                            # noqa: BLE001
        pass
