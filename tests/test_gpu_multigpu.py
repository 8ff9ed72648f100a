"""N > 1 processes: the kinematic structure spread over ranks.  On two real GPUs with the library's own RCCL all-reduce
(skipped on boxes with one GPU), and -- the same worker, placement and checks -- as 2 and 4 processes that share ONE GPU
with the link sums summed over gloo through m3t_hip_comm_set_reduce_callback (round 6).  The N > 1 host logic is also
covered on CPU with gloo in test_sharding_gloo.py."""
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


@pytest.mark.skipif(_n_gpus() < 1, reason="needs a GPU")
@pytest.mark.parametrize("world", [2, 4])
def test_kinematic_chain_over_processes_that_share_one_gpu(world):
    """the worker of the two-GPU test with every rank on device 0: body i's modality in process i mod N, the whole link
    tree in each, tracking_step_tree_segment_kernel with partial ownership, 14 reductions per frame carried by gloo;
    every process ends on the oracle's single-process poses, bit for bit"""
    env = dict(os.environ, M3T_TEST_SHARE_ONE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29900 + (os.getpid() + world) % 90),
                          os.path.join(util.ROOT, "tests", "multigpu_chain_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "multigpu chain ok: %d ranks" % world in out.stdout
