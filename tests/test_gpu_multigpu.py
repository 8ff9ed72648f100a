"""N > 1 on real GPUs (skipped on boxes with one GPU; the N > 1 logic is also covered on CPU with gloo in
test_sharding_gloo.py): the kinematic structure spread over two GPUs with the library's own RCCL all-reduce."""
import os
import subprocess
import sys

import pytest

import util

pytestmark = pytest.mark.gpu


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs")
def test_kinematic_chain_over_two_gpus_with_the_native_all_reduce():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(29800 + os.getpid() % 100),
                          os.path.join(util.ROOT, "tests", "multigpu_chain_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "multigpu chain ok: 2 ranks" in out.stdout
