"""`bench.py --gpus N` rehearsed without GPUs: the driver launches the multi-GPU bench as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`, one process per rank, and no run of
ours has ever had more than one GPU. Here the same command line runs with eight (and two) PROCESSES on this box:
NFC_BENCH_DRY_CPU=1 puts the tensors on the host and gloo between the ranks, NFCGPU_LIB is the emulated test build of the host
runtime (tests/hostsim), NFCGPU_FAKE_RCCL=shm its stand-in for RCCL with the ranks as processes (a shared-memory world named
by the unique id). What it exercises is everything of the N > 1 path that is ours: the environment torch.distributed.run
hands over, build() on rank 0 behind a barrier, the streams cut over the ranks, nfcgpu_comm_unique_id -> broadcast ->
nfcgpu_comm_init on every rank, the timed region with its barriers, nfcgpu_gather_frames_packed, the parse of the gathered
records on rank 0 and the one JSON line. What it cannot: RCCL itself and hipSetDevice on a real device."""
import glob
import json
import os
import socket
import subprocess
import sys

import pytest

import nfc_testlib as T

EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def emulated(built):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    return EMU


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_gpus_n_as_processes_on_the_emulated_runtime(emulated, ranks):
    before = set(glob.glob("/dev/shm/nfcfake_*"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("NFCGPU_")}
    env.update(NFC_BENCH_DRY_CPU="1", NFCGPU_LIB=emulated, NFCGPU_FAKE_RCCL="shm", NFCGPU_FAKE_DEVICES=str(ranks), NFCGPU_WINDOWED_MIN="4096", OMP_NUM_THREADS="1")
    streams = 2 * ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1",
           "--streams", str(streams), "--samples", "65536", "--no-cpu", "--no-points"]
    run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500, cwd=T.ROOT)
    try:
        assert run.returncode == 0, run.stderr[-4000:]
        lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, run.stdout[-2000:]          # rank 0 prints ONE line
        out = json.loads(lines[0])
        assert out["n_gpus"] == ranks and out["steps"] == 1 and out["warmup"] == 1 and out["scaling"] == "strong", out
        assert out["config"]["streams_total"] == streams and out["config"]["streams_per_gpu"] == 2, out["config"]
        assert "ncclAllGather behind the C ABI" in out["config"]["parallelism"], out["config"]["parallelism"]
        assert "DRY RUN" in out["data"], out["data"]
        # every rank's records reached rank 0 (two dense streams x 2 submissions of 65536 samples a rank: frames on every rank)
        assert out["config"]["frame_words_gathered"] > out["config"]["frames_decoded_rank0"] > 0, out["config"]
        assert out["frames_dropped"] == 0, out
    finally:
        left = set(glob.glob("/dev/shm/nfcfake_*")) - before
        for f in left:
            os.unlink(f)
    assert not left, "the stand-in's shared-memory world was not unlinked: %r" % (left,)


DEVICES = r'''
import ctypes, json, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import nfc_testlib as T, nfclab_amd
lib = nfclab_amd.load_library()
cur = lib.nfcgpu_emulated_current_device
cur.restype = ctypes.c_int
mag = np.abs(T.load_fixture("test_NFC-A_106kbps_001")).astype(np.float32)[:40000]
a = nfclab_amd.NfcGpu(device=0, max_streams=64)
b = nfclab_amd.NfcGpu(device=2, max_streams=64)
seen = [("init b", cur())]
sa = a.open(); seen.append(("open a", cur()))
sb = b.open(); seen.append(("open b", cur()))
a.submit(sa, mag, 10000000); seen.append(("submit a", cur()))
b.submit(sb, mag, 10000000); seen.append(("submit b", cur()))
fa = a.poll(sa); seen.append(("poll a", cur()))          # (collects what the submission left: a synchronisation of a's stream)
fb = b.poll(sb); seen.append(("poll b", cur()))
a.submit(sa, mag[:4096], 10000000); seen.append(("submit a", cur()))
b.submit(sb, mag[:4096], 10000000); seen.append(("submit b", cur()))
a.sync(); seen.append(("sync a", cur()))
b.flush(sb); seen.append(("flush b", cur()))
a.stats(); seen.append(("stats a", cur()))
ident = b.comm_unique_id(); seen.append(("unique id b", cur()))
a.comm_init(ident, 0, 1); seen.append(("comm init a", cur()))
b.close(); a.close()
print(json.dumps({"seen": seen, "same_frames": fa == fb and len(fa) > 0}))
'''


def test_every_entry_point_makes_its_contexts_device_current(emulated, tmp_path):
    """a host with a context per GPU calls the C ABI in any order; every entry point that touches the runtime sets its
    context's own device first (csrc/nfcgpu.hip). The stand-in HIP of the emulated build records the device a call leaves current."""
    with open(tmp_path / "devices.py", "w") as f:
        f.write(DEVICES)
    env = {k: v for k, v in os.environ.items() if not k.startswith("NFCGPU_")}
    env.update(NFCGPU_LIB=emulated, NFCGPU_NO_TORCH="1", NFCGPU_FAKE_RCCL="1", NFCGPU_FAKE_DEVICES="4")
    run = subprocess.run([sys.executable, str(tmp_path / "devices.py"), os.path.join(T.ROOT, "nfc-laboratory_amd"), os.path.join(T.ROOT, "tests")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    out = json.loads(run.stdout.splitlines()[-1])
    assert out["same_frames"], out
    # (opening a stream, the statistics and the unique id are the host's business: they leave the current device alone)
    touched = [(what, device) for what, device in out["seen"] if what.split()[0] in ("init", "submit", "sync", "poll", "flush", "comm")]
    assert len(touched) == 10, out["seen"]
    for what, device in touched:
        assert device == (0 if what.endswith(" a") else 2), out["seen"]
