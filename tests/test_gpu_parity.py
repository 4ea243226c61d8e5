"""GPU parity tests: every call goes through the C ABI of libnfcgpu.so (HIP kernels on the MI355X) and is
compared bit-for-bit with the oracle: the reference's golden vectors (tests/golden) and the reference decoder
itself (oracle/_ref/libnfcref.so, built from /root/reference and shipped with the snapshot)."""
import os

import numpy as np
import pytest

import nfc_testlib as T

pytestmark = pytest.mark.gpu

NAMES = T.fixture_names()
FS = 10000000


@pytest.fixture(scope="module")
def gpu(built):
    import nfclab_amd
    g = nfclab_amd.NfcGpu(device=0, max_streams=4096, frame_sink_bytes=64 << 20)
    yield g
    g.close()


def data_frames(frames):
    return [f for f in frames if f[1] in (0x102, 0x103)]


def decode_chunked(gpu, samples, chunk, stride=1, flush=False):
    sid = gpu.open()
    n = samples.size // stride
    for pos in range(0, n, chunk):
        gpu.submit(sid, np.ascontiguousarray(samples[pos * stride:(pos + chunk) * stride]), FS, stride=stride)
    if flush:
        gpu.flush(sid)
    frames = gpu.poll(sid)
    gpu.close_stream(sid)
    return frames


@pytest.mark.parametrize("name", NAMES)
def test_fixture_matches_golden(gpu, name):
    """BASELINE configs 2-4: every wav fixture, all four technologies enabled, 65536-sample buffers
    like test-sdr (src/nfc-test/test-sdr/src/main/cpp/main.cpp:163)."""
    frames = decode_chunked(gpu, T.load_fixture(name), 65536)
    assert data_frames(frames) == T.load_golden(name)


@pytest.mark.parametrize("name", ["test_NFC-A_106kbps_003", "test_NFC-B_106kbps_002", "test_NFC-F_212kbps_002",
                                  "test_NFC-V_26kbps_001", "test_NFC-A_106kbps_212kbps_001"])
def test_streaming_resume_at_odd_buffer_sizes(gpu, name):
    """Decoder state is carried across launches at any sample boundary (checkpoint/resume)."""
    x = T.load_fixture(name)
    assert data_frames(decode_chunked(gpu, x, 4099)) == T.load_golden(name)
    assert data_frames(decode_chunked(gpu, x, 777)) == T.load_golden(name) if x.size < 400000 else True


@pytest.mark.parametrize("name", ["test_POLL_ABF_001", "test_NFC-A_424kbps_001", "test_NFC-F_212kbps_004"])
def test_carrier_and_eof_frames_match_reference(gpu, name):
    """Carrier on/off frames and the end-of-stream frame are not pinned by the goldens: pin them on the live reference."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    x = T.load_fixture(name)
    ref, _ = T.reference_decode(x, keep_carrier=True, send_eof=True)
    assert decode_chunked(gpu, x, 65536, flush=True) == ref


def test_ragged_batch_all_fixtures_one_launch_per_step(gpu):
    """All 18 captures decoded concurrently as one ragged batch (different lengths, same stream block)."""
    xs = [T.load_fixture(n) for n in NAMES]
    first = gpu.open(count=len(xs))
    chunk = 32768
    longest = max(x.size for x in xs)
    for pos in range(0, longest, chunk):
        ids, ptrs, cnts, keep = [], [], [], []
        for i, x in enumerate(xs):
            part = np.ascontiguousarray(x[pos:pos + chunk])
            if part.size:
                keep.append(part)
                ids.append(first + i)
                ptrs.append(part.ctypes.data)
                cnts.append(part.size)
        gpu.submit_batch(ids, ptrs, cnts, FS)
    for i, n in enumerate(NAMES):
        assert data_frames(gpu.poll(first + i)) == T.load_golden(n), n
    for i in range(len(xs)):
        gpu.close_stream(first + i)


def test_iq_entry_matches_magnitude_entry(gpu):
    """float2 IQ path: magnitude computed on the GPU must equal the reference's sqrtf(I*I+Q*Q) (no FMA)."""
    rng = np.random.default_rng(5)
    m = T.load_fixture("test_NFC-A_106kbps_002")
    iq_exact = T.magnitude_to_iq(m, seed=3)
    assert data_frames(decode_chunked(gpu, iq_exact, 65536, stride=2)) == T.load_golden("test_NFC-A_106kbps_002")
    if T.reference_lib() is None:
        return
    # arbitrary phase + small noise: magnitudes are no longer on the int16 grid
    phi = rng.uniform(0, 2 * np.pi, m.size).astype(np.float32)
    iq = np.empty(2 * m.size, np.float32)
    iq[0::2] = m * np.cos(phi) + rng.normal(0, 0.001, m.size).astype(np.float32)
    iq[1::2] = m * np.sin(phi) + rng.normal(0, 0.001, m.size).astype(np.float32)
    mag = np.empty(m.size, np.float32)
    T.reference_lib().nfcref_magnitude(iq.ctypes.data, m.size, mag.ctypes.data)
    ref, _ = T.reference_decode(mag, keep_carrier=True)
    assert decode_chunked(gpu, iq, 65536, stride=2) == ref


def test_iq_magnitude_is_bit_exact(gpu):
    """nfcgpu_magnitude (the device function the decoder applies to stride-2 input) against the reference's
    conversion (RadioDeviceTask.cpp:626-642 through oracle/_ref) on values spanning the float range: normal,
    tiny (denormal squares), large, signed zeros, exact squares."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    rng = np.random.default_rng(11)
    n = 1 << 20
    expo = rng.integers(-70, 60, n).astype(np.float32)
    iq = (rng.standard_normal(2 * n).astype(np.float32) * np.exp2(np.repeat(expo, 2)).astype(np.float32)).astype(np.float32)
    iq[:8] = [0.0, 0.0, -0.0, 0.0, 3.0, 4.0, 1e-30, 1e-30]
    iq[8:16] = [1.0, 0.0, 0.0, -1.0, 0.5, 0.5, 1e19, 1e19]
    want = np.empty(n, np.float32)
    T.reference_lib().nfcref_magnitude(iq.ctypes.data, n, want.ctypes.data)
    got = gpu.magnitude(iq)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name,per_buffer", [("test_NFC-A_106kbps_001", 65536), ("test_POLL_ABF_001", 65536), ("test_NFC-V_26kbps_001", 10007)])
def test_adaptive_resampler_matches_reference_task(gpu, tmp_path, name, per_buffer):
    """nfcgpu_resample_radio against the reference's SignalResamplingTask: the (value, offset)
    control points of every buffer must be bit-identical, including the shorter last buffer."""
    want = T.reference_resample(name, tmp_path, per_buffer)
    if want is None:
        pytest.skip("oracle/_ref/resample-ref not available")
    x = T.load_fixture(name)
    full = x.size // per_buffer
    got = []
    if full:
        got += gpu.resample_radio(x[:full * per_buffer].reshape(full, per_buffer))
    if x.size % per_buffer:
        got += gpu.resample_radio(x[full * per_buffer:].reshape(1, -1))
    assert len(got) == len(want)
    for b, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g.reshape(-1).view(np.uint32), w.view(np.uint32)), "buffer %d: %d vs %d floats" % (b, g.size, w.size)


def test_generic_kernels_at_the_specialised_rate(gpu):
    """10 MS/s normally runs the kernels with the derived constants compiled in; the generic kernels (any sample
    rate) must give the same frames at that rate. A second context with NFCGPU_GENERIC_KERNELS=1 decodes a fixture."""
    import nfclab_amd
    name = "test_POLL_ABF_001"
    os.environ["NFCGPU_GENERIC_KERNELS"] = "1"
    try:
        with nfclab_amd.NfcGpu(device=0, max_streams=64) as generic:
            first = generic.open(count=1)
            x = T.load_fixture(name)
            for pos in range(0, x.size, 50000):
                generic.submit(first, np.ascontiguousarray(x[pos:pos + 50000]), FS)
            got = data_frames(generic.poll(first))
    finally:
        del os.environ["NFCGPU_GENERIC_KERNELS"]
    assert got == T.load_golden(name)


def test_uniform_device_batch_synthetic_streams(gpu):
    """BASELINE config 5 shape at test size: many synthetic streams resident in HBM, uniform pitch, IQ.
    Each stream is checked against the reference decoder run on the same magnitudes."""
    import torch
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    streams, length, step = 96, 1 << 17, 1 << 15
    mags = [T.synthetic_stream_i16(s, length).astype(np.float32) / np.float32(32768.0) for s in range(streams)]
    iq = np.stack([T.magnitude_to_iq(m, seed=s) for s, m in enumerate(mags)])
    dev = torch.from_numpy(iq).cuda()
    torch.cuda.synchronize()
    first = gpu.open(count=streams)
    for pos in range(0, length, step):
        gpu.submit_uniform(first, streams, dev.data_ptr() + pos * 8, dev.stride(0) * 4, step, FS, stride=2)
    gpu.sync()
    total = 0
    for s in range(streams):
        ref, _ = T.reference_decode(np.abs(mags[s]), keep_carrier=True, defined_storage=True)
        got = gpu.poll(first + s)
        assert got == ref, "stream %d" % s
        total += len(got)
        gpu.close_stream(first + s)
    assert total > streams  # the synthetic streams do contain traffic


def test_tech_mask_and_thresholds_follow_reference(gpu):
    import nfclab_amd
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    x = T.load_fixture("test_POLL_ABF_001")
    for mask in (0x1, 0x2, 0x4, 0xA):
        ref, _ = T.reference_decode(x, tech_mask=mask, keep_carrier=True)
        sid = gpu.open(nfclab_amd.default_params(tech_mask=mask))
        gpu.submit(sid, x, FS)
        assert gpu.poll(sid) == ref
        gpu.close_stream(sid)


def test_empty_and_short_inputs(gpu):
    sid = gpu.open()
    gpu.submit(sid, np.zeros(0, np.float32), FS)
    assert gpu.poll(sid) == []
    gpu.submit(sid, np.zeros(5, np.float32), FS)
    gpu.flush(sid)
    frames = gpu.poll(sid)
    if T.reference_lib() is not None:
        ref, _ = T.reference_decode(np.zeros(5, np.float32), keep_carrier=True, send_eof=True)
        assert frames == ref
    assert frames[-1][1] == 0x100 and frames[-1][5] == 4  # end-of-stream frame: CarrierOff at signalClock
    gpu.close_stream(sid)


def test_sample_rate_change_reinitialises_like_reference(gpu):
    """NfcDecoder::nextFrames re-runs initialize() when the buffer's sample rate differs (NfcDecoder.cpp:383-388)."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    x = T.load_fixture("test_NFC-A_106kbps_001")
    sid = gpu.open()
    gpu.submit(sid, x[:30000], 5000000)  # wrong rate first: garbage in, state reset afterwards
    gpu.poll(sid)
    gpu.submit(sid, x, FS)
    assert data_frames(gpu.poll(sid)) == T.load_golden("test_NFC-A_106kbps_001")
    gpu.close_stream(sid)


def test_frame_sink_overflow_is_reported_not_silent(built):
    import nfclab_amd
    x = T.load_fixture("test_NFC-A_424kbps_002")
    with nfclab_amd.NfcGpu(device=0, max_streams=64, frame_sink_bytes=4096) as g:
        first = g.open(count=8)
        for i in range(8):  # ~600 frames in total: more than a 4 KiB sink can hold
            g.submit(first + i, x, FS)
        with pytest.raises(nfclab_amd.NfcGpuError) as e:
            g.poll(first)
        assert e.value.code == -6


def test_a_closed_context_gives_its_device_memory_back(built):
    """three contexts in a row, each taking a capture through the time-parallel path (scan records, planes, lane storage): the
    free device memory afterwards is what it was (a context that kept its front-end planes - 16 B per sample - made the
    third large point of a bench run fail)"""
    import torch
    import nfclab_amd
    x = np.abs(T.load_fixture("test_NFC-F_212kbps_003")).astype(np.float32)
    torch.cuda.synchronize()
    free = []
    for k in range(4):
        with nfclab_amd.NfcGpu(device=0, max_streams=64) as g:
            sid = g.open()
            g.submit(sid, x, FS)
            assert len(g.poll(sid)) > 0
            st = g.stats()
            assert st.windowed_streams == 1 and st.planes_ms >= 0
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info(0)[0])
    assert abs(free[-1] - free[0]) < (8 << 20), free   # (the first context may leave the runtime's own pools behind)


def test_reference_test_sdr_harness_runs_unchanged_on_the_gpu_decoder(built, tmp_path):
    """Drop-in check: the reference's own test-sdr main.cpp, linked against our lab::NfcDecoder shim
    (nfc-laboratory_amd/host/NfcDecoder.cpp -> C ABI -> HIP), must print PASS for its golden files."""
    import os
    import shutil
    import struct
    import subprocess
    exe = os.path.join(T.ROOT, "oracle", "_ref", "test-sdr-gpu")
    if not os.path.exists(exe):
        pytest.skip("test-sdr-gpu not built (needs the reference tree at build time)")
    names = ["test_NFC-A_106kbps_001", "test_NFC-A_424kbps_001", "test_NFC-B_106kbps_001", "test_POLL_AB_001"]
    for name in names:
        raw = T.load_fixture_i16(name).tobytes()
        with open(tmp_path / (name + ".wav"), "wb") as f:
            f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
            f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, FS, FS * 2, 2, 16))
            f.write(b"data" + struct.pack("<I", len(raw)) + raw)
        shutil.copyfile(os.path.join(T.GOLDEN, "wav", name + ".json"), tmp_path / (name + ".json"))
    out = subprocess.run([exe, str(tmp_path) + "/"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900).stdout
    for name in names:
        assert "TEST FILE %s.wav: PASS" % name in out, out


def test_reference_radio_decoder_task_runs_unchanged_on_the_gpu_decoder(built, tmp_path):
    """Task-level drop-in (SURVEY 8(b) outer contract): the reference's RadioDecoderTask, compiled unmodified,
    submitted to rt::Executor, configured / started over radio.decoder.command and fed SignalBuffers over
    radio.signal.raw; its frames on radio.decoder.frame must equal the goldens with the GPU decoder underneath."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-gpu")
    if not os.path.exists(exe):
        pytest.skip("task-gpu not built (needs the reference tree at build time)")
    names = ["test_NFC-A_106kbps_001", "test_NFC-F_212kbps_001", "test_NFC-V_26kbps_001", "test_POLL_ABF_001"]
    got = T.run_task_harness(exe, names, tmp_path)
    for name in names:
        assert got[name] == T.load_golden(name), name


@pytest.mark.parametrize("block", ["300000", "auto"])
def test_radio_decoder_task_with_the_shim_in_block_mode(built, tmp_path, monkeypatch, block):
    """NFCGPU_SHIM_BLOCK: buffers collected into long asynchronous submissions (time-parallel path, pinned double-buffered
    staging), frames of a block handed out when the next block goes in: the task still publishes the golden frames."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-gpu")
    if not os.path.exists(exe):
        pytest.skip("task-gpu not built (needs the reference tree at build time)")
    monkeypatch.setenv("NFCGPU_SHIM_BLOCK", block)  # ("auto": the shim picks and grows the block itself, host/NfcDecoder.cpp)
    names = ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_002", "test_NFC-F_212kbps_001", "test_NFC-V_26kbps_001", "test_POLL_ABF_001"]
    for iq in (False, True):
        got = T.run_task_harness(exe, names, tmp_path, iq=iq)
        for name in names:
            assert got[name] == T.load_golden(name), (name, iq)


def test_radio_decoder_task_fed_with_iq_buffers(built, tmp_path):
    """the task publishes interleaved IQ (SIGNAL_TYPE_RADIO_IQ) instead of host-computed
    magnitudes; the GPU decoder behind the unchanged RadioDecoderTask demodulates from IQ and yields the goldens."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-gpu")
    if not os.path.exists(exe):
        pytest.skip("task-gpu not built (needs the reference tree at build time)")
    names = ["test_NFC-B_106kbps_001", "test_NFC-A_424kbps_001"]
    got = T.run_task_harness(exe, names, tmp_path, iq=True)
    for name in names:
        assert got[name] == T.load_golden(name), name


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzzed_streams_match_reference(gpu, seed):
    """Random cut-and-paste of captures with arbitrary gains, offsets and noise (general fp32, not on the int16 grid)."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    from test_oracle_goldens import _fuzz_stream
    streams = [_fuzz_stream(seed * 100 + i, 200000) for i in range(12)]
    first = gpu.open(count=len(streams))
    for pos in range(0, 200000, 50000):
        parts = [np.ascontiguousarray(x[pos:pos + 50000]) for x in streams]
        gpu.submit_batch([first + i for i in range(len(parts))], [p.ctypes.data for p in parts], [p.size for p in parts], FS)
    for i, x in enumerate(streams):
        ref, _ = T.reference_decode(x, keep_carrier=True, cap=16384, defined_storage=True)
        assert gpu.poll(first + i, capacity=16384) == ref, "stream %d" % i
        gpu.close_stream(first + i)


@pytest.mark.parametrize("rate,step", [(5000000, 2), (2500000, 4)])
def test_other_sample_rates_match_reference(gpu, rate, step):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    for name in ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_001", "test_NFC-F_212kbps_002", "test_NFC-V_26kbps_002"]:
        x = np.ascontiguousarray(T.load_fixture(name)[::step])
        ref, _ = T.reference_decode(x, sample_rate=rate, keep_carrier=True)
        sid = gpu.open()
        gpu.submit(sid, x, rate)
        assert gpu.poll(sid) == ref, name
        gpu.close_stream(sid)


def test_unsupported_sample_rate_is_rejected_loudly(gpu):
    import nfclab_amd
    sid = gpu.open()
    with pytest.raises(nfclab_amd.NfcGpuError) as e:
        gpu.submit(sid, np.zeros(100, np.float32), 20000000)  # look-back would exceed the 512-deep history rings
    assert e.value.code == -5
    gpu.close_stream(sid)


def test_frame_gather_over_rccl_single_rank(gpu):
    """The frame gather of bench.py on the GPU backend (torch.distributed "nccl" = RCCL) with a one-rank group: the
    collectives run on device tensors and return this rank's records unchanged. (world_size 2 is covered on CPU with
    gloo in test_distributed_gather.py; a one-GPU box cannot host two RCCL ranks.)"""
    import torch
    import torch.distributed as dist
    import frames as framelib
    if dist.is_initialized():
        pytest.skip("a process group already exists in this interpreter")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        want = {3: [(0x0101, 0x0102, 0, 0x0102, 105938, 1000, 1940, FS, bytes([0x26]))],
                9: [(0x0102, 0x0103, 2, 0x0103, 105938, 5000, 9000, FS, bytes(range(37)))]}
        words = framelib.pack_frames(want)
        sink = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
        sink[:words.size] = torch.from_numpy(words).to("cuda:0")
        gathered, counts = framelib.gather_sinks(sink, int(words.size), 1)
        torch.cuda.synchronize()
        assert counts == [int(words.size)]
        assert framelib.parse_sink(gathered[0, :counts[0]].cpu().numpy(), counts[0], FS) == want
    finally:
        dist.destroy_process_group()


def test_frame_gather_through_the_c_abi_single_rank(gpu):
    """nfcgpu_comm_* / nfcgpu_gather_frames: the frame gather in C++ over RCCL (ncclAllGather of the counts, then the
    records at their exact sizes), here with a one-rank communicator on the context's own sink after a real decode; and the streaming-read
    measurement used as the second roofline denominator."""
    import torch
    import frames as framelib
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    mag = np.abs(T.load_fixture("test_NFC-A_106kbps_001")).astype(np.float32)
    want, _ = T.reference_decode(mag, keep_carrier=True)
    sink = torch.zeros(1 << 20, dtype=torch.int32, device="cuda:0")
    ctl = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    out = torch.zeros(1 << 20, dtype=torch.int32, device="cuda:0")
    g = gpu
    g.sink_attach(sink.data_ptr(), sink.numel(), ctl.data_ptr())
    g.sink_hold(True)
    try:
        sid = g.open()
        g.submit(sid, mag, FS)
        g.sync()
        g.comm_init(g.comm_unique_id(), 0, 1)
        counts, stride = g.gather_frames(out.data_ptr(), out.numel())
        torch.cuda.synchronize()
        assert counts == [int(ctl[0].item())] and counts[0] > 0 and stride == 0  # records packed at their exact sizes
        got = framelib.parse_sink(out[:counts[0]].cpu().numpy(), counts[0], FS)
        assert got[sid] == want
        g.comm_destroy()
        gbps = g.read_bandwidth(out.data_ptr(), out.numel() * 4, repeats=3)
        assert gbps > 10.0
    finally:
        g.sink_hold(False)
        g.sink_attach(None, 0, None)


def test_streams_joining_a_block_at_different_times(gpu):
    """Streams of one block opened at different moments and fed ragged buffer lengths: blocks then hold streams that
    need the exact-modulo kernel (first 1024 samples) next to streams that do not, launches mix both kernels, and the
    host-side clock mirrors must keep agreeing with the device. Every stream is compared with the reference."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    rng = np.random.default_rng(77)
    from test_oracle_goldens import _fuzz_stream
    total = 120000
    n = 24
    data = [_fuzz_stream(7000 + i, total) for i in range(n)]
    start_step = [0 if i < 6 else int(rng.integers(1, 12)) for i in range(n)]
    fed = [0] * n
    ids = [None] * n
    step = 0
    while any(f < total for f in fed):
        sel, ptrs, cnts, keep = [], [], [], []
        for i in range(n):
            if step < start_step[i] or fed[i] >= total:
                continue
            if ids[i] is None:
                ids[i] = gpu.open()
            if rng.random() < 0.2:
                continue  # this stream skips the step
            c = int(min(total - fed[i], rng.integers(1, 9000)))
            part = np.ascontiguousarray(data[i][fed[i]:fed[i] + c])
            keep.append(part)
            sel.append(ids[i]); ptrs.append(part.ctypes.data); cnts.append(c)
            fed[i] += c
        if sel:
            gpu.submit_batch(sel, ptrs, cnts, FS)
        step += 1
    for i in range(n):
        ref, _ = T.reference_decode(data[i], keep_carrier=True, cap=16384, defined_storage=True)
        assert gpu.poll(ids[i], capacity=16384) == ref, "stream %d" % i
        gpu.close_stream(ids[i])


def test_streams_with_own_parameters_between_streams_already_running(gpu):
    """Three neighbouring slots with three tech masks, opened with their sample rate set; the middle one is fed only
    after its neighbours have run (a batch initialises only the streams it lists, each with its own configuration)."""
    import nfclab_amd
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    x = T.load_fixture("test_POLL_ABF_001")
    masks = (0x1, 0x4, 0x2)
    sids = [gpu.open(nfclab_amd.default_params(sample_rate=FS, tech_mask=m)) for m in masks]
    assert sids == list(range(sids[0], sids[0] + 3))
    half = x.size // 2
    outer = [np.ascontiguousarray(x[:half]), np.ascontiguousarray(x[:half])]
    gpu.submit_batch([sids[0], sids[2]], [p.ctypes.data for p in outer], [half, half], FS)
    rest = np.ascontiguousarray(x[half:])
    gpu.submit_batch(sids, [rest.ctypes.data, x.ctypes.data, rest.ctypes.data], [rest.size, x.size, rest.size], FS)
    for sid, mask in zip(sids, masks):
        ref, _ = T.reference_decode(x, tech_mask=mask, keep_carrier=True)
        assert gpu.poll(sid) == ref, hex(mask)
        gpu.close_stream(sid)


def test_rejected_batch_leaves_no_trace(gpu):
    """A batch refused half way through validation (unknown stream, stream listed twice) must not mark its streams."""
    import nfclab_amd
    x = T.load_fixture("test_NFC-A_106kbps_001")
    sid = gpu.open()
    for ids in ([sid, 0x7FFFFFFF], [sid, sid]):
        with pytest.raises(nfclab_amd.NfcGpuError):
            gpu.submit_batch(ids, [x.ctypes.data] * 2, [x.size] * 2, FS)
    gpu.submit_batch([sid], [x.ctypes.data], [x.size], FS)
    assert data_frames(gpu.poll(sid)) == T.load_golden("test_NFC-A_106kbps_001")
    gpu.close_stream(sid)


def test_one_batch_of_streams_with_different_thresholds(gpu):
    """Eight fuzzed captures in one ragged batch, each stream with its own tech mask, power level, correlation and
    modulation-depth thresholds: one launch per configuration inside the batch, each stream against the reference run
    with the same parameters."""
    import ctypes
    import nfclab_amd
    from test_oracle_goldens import _fuzz_stream
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    rng = np.random.default_rng(77)
    captures, sids, refs = [], [], []
    for i in range(8):
        x = _fuzz_stream(900 + i, 200000)
        p = nfclab_amd.default_params(tech_mask=int(rng.integers(1, 16)))
        rp = T.RefParams(p.tech_mask, float("nan"), *[(ctypes.c_float * 4)(*([float("nan")] * 4)) for _ in range(3)])
        if i % 2:
            p.power_level_threshold = rp.power_level_threshold = float(np.float32(rng.uniform(0.004, 0.05)))
        for t in range(4):
            if rng.random() < 0.5:
                p.corr_threshold[t] = rp.corr_threshold[t] = float(np.float32(rng.uniform(0.2, 0.9)))
            if rng.random() < 0.5:
                p.min_modulation_depth[t] = rp.min_depth[t] = float(np.float32(rng.uniform(0.05, 0.95)))
            if rng.random() < 0.5:
                p.max_modulation_depth[t] = rp.max_depth[t] = float(np.float32(rng.uniform(0.5, 1.0)))
        ref, _ = T.reference_decode(x, keep_carrier=True, cap=16384, params=rp, defined_storage=True)
        captures.append(x)
        refs.append(ref)
        sids.append(gpu.open(p))
    fed = [0] * 8
    while any(f < 200000 for f in fed):
        ids, parts = [], []
        for i in range(8):
            if fed[i] < 200000 and rng.random() < 0.8:
                c = int(min(200000 - fed[i], rng.integers(1, 50000)))
                parts.append(np.ascontiguousarray(captures[i][fed[i]:fed[i] + c]))
                ids.append(sids[i])
                fed[i] += c
        if ids:
            gpu.submit_batch(ids, [q.ctypes.data for q in parts], [q.size for q in parts], FS)
    for i in range(8):
        assert gpu.poll(sids[i], capacity=16384) == refs[i], i
        gpu.close_stream(sids[i])
    assert sum(len(r) for r in refs) > 40


def test_synthetic_nfcv_one_of_256_frames(gpu):
    """NFC-V 1-of-256 pulse-position frames (no capture of the reference uses that coding) next to 1-of-4 ones."""
    from test_oracle_goldens import _nfcv_capture
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    x, want = _nfcv_capture(2)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=4096, defined_storage=True)
    got = decode_chunked(gpu, x, 65536)
    assert got == ref
    assert [(256 if f[4] == 1655 else 4, f[-1]) for f in got if f[0] == 0x104 and f[1] == 0x102] == want


def test_special_sample_values(gpu):
    """Negative and zero runs, denormals, huge values and sign flips in a float capture (finite values only here; the
    CPU suite also covers NaN and infinities on the step machine): one ragged batch of six captures."""
    from test_oracle_goldens import _special_values_capture
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    captures = [_special_values_capture(20 + i, finite_only=True) for i in range(6)]
    first = gpu.open(count=6)
    for pos in range(0, 200000, 50000):
        parts = [np.ascontiguousarray(c[pos:pos + 50000 - 7 * i]) for i, c in enumerate(captures)]
        gpu.submit_batch([first + i for i in range(6)], [p.ctypes.data for p in parts], [p.size for p in parts], FS)
        rest = [np.ascontiguousarray(c[pos + 50000 - 7 * i:pos + 50000]) for i, c in enumerate(captures)]
        gpu.submit_batch([first + i for i in range(6)], [p.ctypes.data for p in rest], [p.size for p in rest], FS)
    for i, c in enumerate(captures):
        ref, _ = T.reference_decode(c, keep_carrier=True, cap=16384, defined_storage=True)
        assert gpu.poll(first + i, capacity=16384) == ref, i
        gpu.close_stream(first + i)


def test_resampled_capture_at_rtl_sdr_rate(gpu):
    """3.2 MS/s (RTL-SDR, the slowest receiver class of the reference): captures resampled by linear interpolation, all
    periods and windows rounded from a non-integer ratio to 10 MS/s; generic kernels."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not available")
    rate = 3200000
    for name in ["test_NFC-A_424kbps_002", "test_NFC-F_212kbps_003"]:
        x = T.load_fixture(name)
        t = np.arange(int(x.size * rate / 10e6), dtype=np.float64) * (10e6 / rate)
        y = np.interp(t, np.arange(x.size), x).astype(np.float32)
        ref, _ = T.reference_decode(y, sample_rate=rate, keep_carrier=True, cap=16384, defined_storage=True)
        sid = gpu.open()
        gpu.submit(sid, y, rate)
        assert gpu.poll(sid, capacity=16384) == ref, name
        gpu.close_stream(sid)
