"""nfcgpu_gather_frames with more than one rank: two processes, two GPUs, RCCL behind the C ABI. Skips on a box with one
GPU (the driver's multi-GPU node runs it)."""
import os
import subprocess
import sys

import pytest

import nfc_testlib as T


@pytest.mark.gpu
def test_frame_gather_through_the_c_abi_two_ranks(built):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(T.ROOT, "tests", "gather_driver.py")],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-3000:]
    assert "rank 0 ok" in run.stdout and "rank 1 ok" in run.stdout, run.stdout[-3000:]
