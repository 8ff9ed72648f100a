import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # seeded inputs with the same arguments are generated once per session (bench_inputs.py: every load
    # is a fresh copy, a test that edits its inputs does not reach another test's); minutes of numpy on a GPU box
    if "M3T_INPUT_CACHE" not in os.environ:
        import atexit
        import shutil
        import tempfile
        d = tempfile.mkdtemp(prefix="m3t_inputs_")
        os.environ["M3T_INPUT_CACHE"] = d
        atexit.register(shutil.rmtree, d, True)


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("3dobjecttracking_amd")


def pytest_sessionstart(session):
    """A fresh checkout has no built library (it is git-ignored): build it once, here, so that the ABI tests do not
    depend on a step run before pytest.  On a GPU box initialise torch's HIP runtime BEFORE libm3t_hip.so loads /opt/rocm's: torch
    bundles its own libamdhip64 and reports "no GPUs" if it comes second (the RCCL test needs it)."""
    build = importlib.import_module("3dobjecttracking_amd.build")
    if not os.path.exists(build.LIB):
        build.build()
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
