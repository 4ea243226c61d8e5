"""The host runtime behind the C ABI (nfc-laboratory_amd/csrc/nfcgpu.hip) and the lab::NfcDecoder shim exercised on a box
without a GPU: tests/hostsim/build_emulated.sh compiles the runtime, unchanged, against a stand-in HIP whose kernel
launches call CPU twins of the kernels built on the product's device step machine (test infrastructure, not a CPU path
of the product: the real library refuses to work without a GPU, tests/test_abi.py). What this covers is everything the
kernels do not: batching and ragged work tables, configuration resolution, (re)initialisation, the clock mirror that
decides whether the exact-modulo kernels are launched, staging, frame sink draining, flush / reset / close. The same
tests run against the real library and kernels with `-m gpu`."""
import os
import subprocess
import sys

import pytest

import nfc_testlib as T

EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")

# need a real GPU: device tensors, RCCL, or binaries linked against the real library (those run below with the emulated
# runtime preloaded instead)
NEEDS_GPU = ["test_uniform_device_batch_synthetic_streams", "test_frame_gather_over_rccl_single_rank", "test_frame_gather_through_the_c_abi_single_rank",
             "test_reference_test_sdr_harness_runs_unchanged_on_the_gpu_decoder",
             "test_reference_radio_decoder_task_runs_unchanged_on_the_gpu_decoder", "test_radio_decoder_task_fed_with_iq_buffers",
             "test_radio_decoder_task_with_the_shim_in_block_mode", "test_a_closed_context_gives_its_device_memory_back"]


@pytest.fixture(scope="module")
def emulated(built):
    sources = [os.path.join(T.ROOT, "nfc-laboratory_amd", "csrc", f) for f in os.listdir(os.path.join(T.ROOT, "nfc-laboratory_amd", "csrc"))]
    sources += [os.path.join(T.ROOT, "tests", "hostsim", f) for f in ("emu_kernels.cpp", "build_emulated.sh", "fakehip/hip/hip_runtime.h")]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in sources):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    return EMU


def test_c_abi_parity_suite_on_the_emulated_runtime(emulated):
    env = dict(os.environ, NFCGPU_LIB=emulated, NFCGPU_NO_TORCH="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(T.ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
           "-p", "no:cacheprovider"]
    # the tests of that file are independent of each other (a context of its own per test module and worker): a few of them
    # at a time where pytest-xdist is there (ten minutes -> four on eight cores: this is the longest test of the CPU suite)
    try:
        import xdist  # noqa: F401
        workers = max(1, min(4, (os.cpu_count() or 2) // 2))
        if workers > 1:
            cmd += ["-n", str(workers)]
    except ImportError:
        pass
    for name in NEEDS_GPU:
        cmd += ["--deselect", "tests/test_gpu_parity.py::" + name]
    run = subprocess.run(cmd, cwd=T.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = run.stdout[-3000:]
    assert run.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def _preloaded(emulated):
    return dict(os.environ, LD_PRELOAD=emulated)


def test_reference_test_sdr_harness_and_shim_on_the_emulated_runtime(emulated, tmp_path):
    """The reference's own golden harness, compiled unmodified against the lab::NfcDecoder shim, with the emulated runtime
    preloaded in place of libnfcgpu.so: shim + host runtime reproduce the goldens (the GPU twin of this test is
    test_reference_test_sdr_harness_runs_unchanged_on_the_gpu_decoder)."""
    import shutil
    exe = os.path.join(T.ROOT, "oracle", "_ref", "test-sdr-gpu")
    if not os.path.exists(exe):
        pytest.skip("test-sdr-gpu not built (needs the reference tree at build time)")
    names = ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_001", "test_NFC-F_212kbps_002", "test_NFC-V_26kbps_002", "test_POLL_ABF_001"]
    for name in names:
        T.write_wav(str(tmp_path / (name + ".wav")), T.load_fixture_i16(name))
        shutil.copyfile(os.path.join(T.GOLDEN, "wav", name + ".json"), tmp_path / (name + ".json"))
    out = subprocess.run([exe, str(tmp_path) + "/"], env=_preloaded(emulated), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True, timeout=600).stdout
    for name in names:
        assert "TEST FILE %s.wav: PASS" % name in out, out


def test_reference_radio_decoder_task_and_shim_on_the_emulated_runtime(emulated, tmp_path, monkeypatch):
    """The reference's RadioDecoderTask driven through its subjects (tests/dropin/task_harness.cpp), magnitude and IQ
    buffers, with the emulated runtime preloaded."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-gpu")
    if not os.path.exists(exe):
        pytest.skip("task-gpu not built (needs the reference tree at build time)")
    monkeypatch.setenv("LD_PRELOAD", emulated)
    names = ["test_NFC-A_106kbps_001", "test_POLL_ABF_001"]
    for iq in (False, True):
        got = T.run_task_harness(exe, names, tmp_path, iq=iq)
        for name in names:
            assert got[name] == T.load_golden(name), (name, iq)


@pytest.mark.parametrize("block", ["300000", "auto"])
def test_shim_block_mode_on_the_emulated_runtime(emulated, tmp_path, monkeypatch, block):
    """NFCGPU_SHIM_BLOCK: the shim collects the task's 65536-sample buffers into long submissions (time-parallel path) and
    hands a block's frames out when the next one goes in; same frames, same order, magnitude and IQ buffers."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-gpu")
    if not os.path.exists(exe):
        pytest.skip("task-gpu not built (needs the reference tree at build time)")
    monkeypatch.setenv("LD_PRELOAD", emulated)
    monkeypatch.setenv("NFCGPU_SHIM_BLOCK", block)  # ("auto": the shim picks and grows the block itself, host/NfcDecoder.cpp)
    names = ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_002", "test_NFC-F_212kbps_001", "test_POLL_ABF_001"]
    for iq in (False, True):
        got = T.run_task_harness(exe, names, tmp_path, iq=iq)
        for name in names:
            assert got[name] == T.load_golden(name), (name, iq)


UNIFORM_DRIVER = r'''
import sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import nfc_testlib as T, nfclab_amd
from test_oracle_goldens import _fuzz_stream
S, L, K = 70, 8192, 5            # 70 streams: one full stream block and a partial one
data = np.stack([np.abs(_fuzz_stream(4000 + s, L * K)) for s in range(S)]).astype(np.float32)
iq = np.zeros((S, L * K, 2), np.float32)
iq[:, :, 0] = data               # phase 0: |I + j0| is the magnitude itself
bad = []
with nfclab_amd.NfcGpu(device=0, max_streams=128) as gpu:
    extra = gpu.open()           # a stream ahead of the range: the range starts in the middle of a block
    first = gpu.open(count=S)
    for k in range(K):
        if k % 2:
            gpu.submit_uniform(first, S, iq.ctypes.data + k * L * 8, L * K * 8, L, 10000000, stride=2, location=nfclab_amd.LOC_DEVICE)
        else:
            gpu.submit_uniform(first, S, data.ctypes.data + k * L * 4, L * K * 4, L, 10000000, stride=1, location=nfclab_amd.LOC_HOST)
    frames = 0
    refs = []
    for s in range(S):
        ref, _ = T.reference_decode(data[s], chunk=L, keep_carrier=True, cap=8192, defined_storage=True)
        refs.append(ref)
        frames += len(ref)
        if gpu.poll(first + s, capacity=8192) != ref:
            bad.append(s)

# the same once more the way bench.py collects frames: a caller-provided sink, held (never drained by the runtime),
# parsed from the packed records
import frames as framelib
sink = np.zeros(1 << 22, np.int32)
ctl = np.zeros(4, np.int32)
bad_sink = []
with nfclab_amd.NfcGpu(device=0, max_streams=128, frame_sink_bytes=1 << 20) as gpu:
    gpu.sink_attach(sink.ctypes.data, sink.size, ctl.ctypes.data)
    gpu.sink_hold(True)
    first = gpu.open(count=S)
    for k in range(K):
        gpu.submit_uniform(first, S, data.ctypes.data + k * L * 4, L * K * 4, L, 10000000, stride=1, location=nfclab_amd.LOC_DEVICE)
    gpu.sync()
    parsed = framelib.parse_sink(sink, int(ctl[0]), 10000000)
    for s in range(S):
        if parsed.get(first + s, []) != refs[s]:
            bad_sink.append(s)
    dropped = int(ctl[1])
print(json.dumps({"bad": bad, "frames": frames, "bad_sink": bad_sink, "dropped": dropped}))
'''


def test_uniform_layout_submissions_on_the_emulated_runtime(emulated, tmp_path):
    """nfcgpu_submit_uniform ([stream][sample] layout, what bench.py uses with HBM-resident input): host-resident rows
    and rows handed over as device memory (plain memory here), magnitude and IQ, a range that starts inside a stream
    block; every stream against the reference, collected by polling and, as bench.py does, from a caller-provided held sink."""
    import json
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    with open(tmp_path / "driver.py", "w") as f:
        f.write(UNIFORM_DRIVER)
    run = subprocess.run([sys.executable, str(tmp_path / "driver.py"), os.path.join(T.ROOT, "nfc-laboratory_amd"), os.path.join(T.ROOT, "tests")],
                         env=dict(os.environ, NFCGPU_LIB=emulated, NFCGPU_NO_TORCH="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    out = json.loads(run.stdout.splitlines()[-1])
    assert out["bad"] == [] and out["bad_sink"] == [] and out["dropped"] == 0 and out["frames"] > 200, out


CONFIG_DRIVER = r'''
import sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np
import nfc_testlib as T, nfclab_amd
x = T.load_fixture("test_NFC-A_106kbps_001")[:18000]
out = {}
with nfclab_amd.NfcGpu(device=0, max_streams=320) as gpu:
    # one stream whose thresholds change 300 times: every change is a new configuration, only one is in use at a time
    sid = gpu.open()
    for k in range(300):
        p = nfclab_amd.default_params()
        p.corr_threshold[1] = 0.2 + 0.002 * k          # NFC-B threshold: no effect on this NFC-A capture
        gpu.configure(sid, p)
        gpu.submit(sid, x[k * 60:(k + 1) * 60], 10000000)
    ref, _ = T.reference_decode(x, keep_carrier=True)
    out["changing"] = gpu.poll(sid) == ref
    gpu.close_stream(sid)
    # 260 streams with 260 distinct configurations at once: more than the table holds
    first = gpu.open(count=260)
    code = 0
    for s in range(260):
        p = nfclab_amd.default_params()
        p.corr_threshold[2] = 0.1 + 0.003 * s
        gpu.configure(first + s, p)
        try:
            gpu.submit(first + s, x[:64], 10000000)
        except nfclab_amd.NfcGpuError as e:
            code = e.code
            out["failed_at"] = s
            break
    out["code"] = code
    # room again once streams are closed
    for s in range(10):
        gpu.close_stream(first + s)
    try:
        gpu.submit(first + out.get("failed_at", 259), x[:64], 10000000)
        out["after_close"] = True
    except nfclab_amd.NfcGpuError:
        out["after_close"] = False
print(json.dumps(out))
'''


def test_configuration_table_is_recycled(emulated, tmp_path):
    """Every distinct set of thresholds is one entry of a 256-entry device table. Entries no stream refers to any more are
    reused (a stream whose thresholds change 300 times keeps decoding), and more than 256 in use at once is refused loudly."""
    import json
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    with open(tmp_path / "driver.py", "w") as f:
        f.write(CONFIG_DRIVER)
    run = subprocess.run([sys.executable, str(tmp_path / "driver.py"), os.path.join(T.ROOT, "nfc-laboratory_amd"), os.path.join(T.ROOT, "tests")],
                         env=dict(os.environ, NFCGPU_LIB=emulated, NFCGPU_NO_TORCH="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    out = json.loads(run.stdout.splitlines()[-1])
    assert out["changing"] is True
    assert out["code"] == -3 and out["failed_at"] == 256, out     # NFCGPU_ENOMEM at the 257th configuration in use
    assert out["after_close"] is True


def test_a_block_is_taken_by_one_kernel_per_launch_at_the_clock_wrap(emulated):
    """Two stream blocks of one launch whose clocks sit one buffer apart just before the exact-modulo zone of the 32-bit
    wrap: the common kernel runs first and advances the clocks of the blocks it takes; the exact kernel of the same launch
    must not look at those blocks again (their clocks may have moved into its zone meanwhile). Before the launch stamp
    (NfcStreamState::served) the second block was decoded twice and its clock ran 4096 samples ahead of the host mirror."""
    import ctypes
    import numpy as np
    lib = ctypes.CDLL(emulated)
    vp, u32 = ctypes.c_void_p, ctypes.c_uint32
    ctx = vp()
    assert lib.nfcgpu_init(0, None, ctypes.byref(ctx)) == 0
    try:
        first = u32()
        lib.nfcgpu_stream_open_many.argtypes = [vp, vp, u32, ctypes.POINTER(u32)]
        assert lib.nfcgpu_stream_open_many(ctx, None, 128, ctypes.byref(first)) == 0
        n = 4096
        idle = (0.25 + (np.arange(n) % 3) / 32768.0).astype(np.float32)
        ids = (u32 * 2)(first.value, first.value + 64)
        ptrs = (vp * 2)(idle.ctypes.data, idle.ctypes.data)
        cnts = (u32 * 2)(n, n)

        class Batch(ctypes.Structure):
            _fields_ = [("n_streams", u32), ("stride", u32), ("location", u32), ("sample_rate", u32), ("stream_ids", ctypes.POINTER(u32)),
                        ("data", ctypes.POINTER(vp)), ("n_samples", ctypes.POINTER(u32))]

        b = Batch(2, 1, 0, 10000000, ids, ptrs, cnts)
        lib.nfcgpu_submit_batch.argtypes = [vp, ctypes.POINTER(Batch)]
        assert lib.nfcgpu_submit_batch(ctx, ctypes.byref(b)) == 0   # initialises both streams
        lib.nfcgpu_test_set_clock.argtypes = [vp, u32, u32]
        lib.nfcgpu_test_get_clock.argtypes = [vp, u32, ctypes.POINTER(u32), ctypes.POINTER(u32)]
        zone = (1 << 32) - 1024 - 1          # clocks from here on need the exact-modulo kernel
        c0 = zone - 2000                      # block 0: enters the zone during this buffer -> exact kernel
        c1 = c0 - n                           # block 1: one buffer behind -> common kernel, then inside the span test
        assert lib.nfcgpu_test_set_clock(ctx, ids[0], c0) == 0 and lib.nfcgpu_test_set_clock(ctx, ids[1], c1) == 0
        assert lib.nfcgpu_submit_batch(ctx, ctypes.byref(b)) == 0
        for sid, start in ((ids[0], c0), (ids[1], c1)):
            dev, mir = u32(), u32()
            assert lib.nfcgpu_test_get_clock(ctx, sid, ctypes.byref(dev), ctypes.byref(mir)) == 0
            assert dev.value == (start + n) & 0xFFFFFFFF, (sid, dev.value - start)
            assert mir.value == dev.value
    finally:
        lib.nfcgpu_shutdown(ctx)
