"""MI355X-native implementation of M3T's per-frame pose-optimisation hot path.

RegionModality correspondence lines + DepthModality ICP + the Gauss-Newton /
Tikhonov solve behind Tracker::ExecuteTrackingStep, as hand-written HIP kernels
for gfx950 behind a C-ABI (include/m3t_hip.h, csrc/libm3t_hip.so).  This Python
package is only the thin host mirror used by tests and bench.py; a C++ host
binds the same C-ABI directly.

The directory name starts with a digit, so import it with
    importlib.import_module("3dobjecttracking_amd")
"""
import os

from . import config, evaluation, generator, host, sharding, synthetic  # noqa: F401
from ._capi import CApi, M3TError  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC_DIR, "libm3t_hip.so")


def open_context(device_id=0):
    """Create one device context (one per GPU per host thread).

    Fails loudly when the HIP extension has not been built or no GPU is
    usable: there is no CPU fallback in the product path.

    Small batches (fewer objects than CUs) run several workgroups per object that
    exchange results inside one launch; that launch shape assumes the context has
    the GPU to itself.  A process that shares the GPU with other contexts, streams
    or processes calls ``ctx.call("set_object_split", 0)`` (include/m3t_hip.h,
    m3t_hip_set_object_split)."""
    path = os.environ.get("M3T_HIP_LIBRARY", LIB_PATH)  # another build of the same HIP library (A/B measurements)
    if not os.path.exists(path):
        raise RuntimeError("%s missing: build it with `python __graft_entry__.py`" % path)
    return CApi(path, "m3t_hip_", device_id)
