"""CPU tests: the oracle (the real reference, oracle/_ref) is pinned against the reference's golden
vectors, and the CPU build of the device state machine (tests/hostsim) is checked against both."""
import hashlib
import os

import numpy as np
import pytest

import nfc_testlib as T

NAMES = T.fixture_names()


def test_manifest_matches_fixture_files():
    m = T.manifest()
    assert len(m) == 18
    total = 0
    for name, info in m.items():
        raw = T.load_fixture_i16(name)
        assert raw.size == info["samples"]
        assert hashlib.sha256(raw.tobytes()).hexdigest() == info["sha256"]
        assert len(T.load_golden(name)) == info["frames"]
        total += info["frames"]
    assert total == 284


@pytest.mark.parametrize("name", NAMES)
def test_reference_oracle_reproduces_goldens(built, name):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    frames, _ = T.reference_decode(T.load_fixture(name))
    assert frames == T.load_golden(name)


@pytest.mark.parametrize("name", NAMES)
def test_step_machine_matches_goldens(built, name):
    frames = T.hostsim_decode(T.load_fixture(name), lane=hash(name) % 64)
    assert frames == T.load_golden(name)


@pytest.mark.parametrize("name", ["test_POLL_ABF_001", "test_NFC-A_106kbps_001", "test_NFC-V_26kbps_002"])
def test_step_machine_matches_reference_with_carrier_frames(built, name):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = T.load_fixture(name)
    ref, _ = T.reference_decode(x, keep_carrier=True)
    assert T.hostsim_decode(x, keep_carrier=True) == ref


def test_reference_is_chunking_invariant(built):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = T.load_fixture("test_NFC-A_106kbps_002")
    a, _ = T.reference_decode(x, chunk=65536)
    b, _ = T.reference_decode(x, chunk=4099)
    assert a == b == T.load_golden("test_NFC-A_106kbps_002")


def test_iq_magnitude_is_exact_for_axis_aligned_iq(built):
    """The synthetic IQ used by bench.py puts the int16 magnitude on one axis: sqrt(m*m) == |m| in fp32."""
    m = T.load_fixture("test_NFC-B_106kbps_001")
    iq = T.magnitude_to_iq(m, seed=7)
    mag = np.sqrt((iq[0::2] * iq[0::2]).astype(np.float32) + (iq[1::2] * iq[1::2]).astype(np.float32), dtype=np.float32)
    assert np.array_equal(mag, np.abs(m))
    assert T.hostsim_decode(iq, stride=2) == T.hostsim_decode(np.abs(m))


def test_empty_and_tiny_inputs(built):
    assert T.hostsim_decode(np.zeros(0, np.float32)) == []
    assert T.hostsim_decode(np.zeros(10, np.float32)) == []
    if T.reference_lib() is not None:
        ref, _ = T.reference_decode(np.zeros(2000, np.float32), keep_carrier=True)
        assert T.hostsim_decode(np.zeros(2000, np.float32), keep_carrier=True) == ref


def _fuzz_stream(seed, length=400000):
    """Random cut-and-paste of fixture pieces with arbitrary (non int16-grid) gains, offsets and noise: exercises
    resets, truncated frames, stale correlator sums after locks and general fp32 rounding."""
    rng = np.random.default_rng(seed)
    names = T.fixture_names()
    out = np.empty(length, np.float32)
    pos = 0
    while pos < length:
        x = T.load_fixture(names[rng.integers(len(names))])
        n = int(rng.integers(20000, 200000))
        a = int(rng.integers(0, max(1, x.size - n)))
        piece = x[a:a + n] * np.float32(rng.uniform(0.5, 1.5)) + np.float32(rng.uniform(-0.003, 0.003))
        piece = piece + rng.normal(0, rng.uniform(0, 0.002), piece.size).astype(np.float32)
        m = min(length - pos, piece.size)
        out[pos:pos + m] = piece[:m]
        pos += m
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_step_machine_matches_reference_on_fuzzed_streams(built, seed):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = _fuzz_stream(seed)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=16384, defined_storage=True)
    got = T.hostsim_decode(x, keep_carrier=True, cap=16384, lane=seed)
    assert got == ref
    assert len(ref) > 0


_SPECIAL = np.array([-0.25, 0.0, -0.0, 1e-40, -1e-40, 3e38, 1e-30, 65504.0, -1.0, np.nan, np.inf, -np.inf], np.float32)


def _special_values_capture(seed, finite_only=False):
    """a fuzzed capture with samples no receiver delivers but a float WAV may hold: negative and zero runs, denormals,
    huge values, sign flips, NaN and infinities"""
    rng = np.random.default_rng(seed)
    x = _fuzz_stream(seed, 200000).copy()
    pos = rng.integers(0, x.size, int(rng.integers(4, 40)))
    kind = seed % (3 if finite_only else 4)
    if kind == 0:
        x[pos] = _SPECIAL[rng.integers(0, 9, pos.size)]
    elif kind == 1:
        for p in pos[:6]:
            x[p:p + int(rng.integers(1, 3000))] = _SPECIAL[rng.integers(0, 9)]
    elif kind == 2:
        x[pos[0]:] = -x[pos[0]:]
    else:
        x[pos[:3]] = _SPECIAL[rng.integers(9, 12, 3)]
    return x


@pytest.mark.parametrize("seed", range(8))
def test_step_machine_follows_reference_on_special_sample_values(built, seed):
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = _special_values_capture(seed)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=16384, defined_storage=True)
    assert T.hostsim_decode(x, keep_carrier=True, cap=16384, lane=seed) == ref


def _nfcv_capture(seed):
    """several synthetic NFC-V reader frames, 1-of-4 and 1-of-256 mixed, random payloads, levels and depths"""
    rng = np.random.default_rng(seed)
    parts, want = [], []
    for k in range(4):
        body = bytes(rng.integers(0, 256, int(rng.integers(2, 6)), dtype=np.uint8))
        if k % 2 == 0:
            body += T.crc_iso15693(body)
        mode = 256 if (k + seed) % 2 else 4
        parts.append(T.synth_nfcv_poll(body, mode, lead=20000 if k else 30000, tail=int(rng.integers(20000, 60000)), level=0.4,
                                       depth=float(rng.uniform(0.93, 1.0)), noise=float(rng.uniform(0, 0.001)), seed=seed * 8 + k))
        want.append((mode, body))
    return np.ascontiguousarray(np.concatenate(parts)), want


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_synthetic_nfcv_one_of_256_frames(built, seed):
    """NFC-V pulse-position frames in 1-of-256 coding (symbol rate / 32), which none of the reference's captures uses,
    next to 1-of-4 ones: the reference decodes the synthetic frames to their payloads and the step machine agrees."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x, want = _nfcv_capture(seed)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=4096, defined_storage=True)
    got = T.hostsim_decode(x, keep_carrier=True, cap=4096, lane=seed)
    assert got == ref
    polls = [f for f in ref if f[0] == 0x104 and f[1] == 0x102]
    assert [(256 if f[4] == 1655 else 4, f[-1]) for f in polls] == want


@pytest.mark.parametrize("seed", range(8))
def test_step_machine_matches_reference_with_random_parameters(built, seed):
    """Random tech mask, power level, correlation and modulation-depth thresholds and sample rate (some of them only a
    label: the capture is not resampled) on a fuzzed capture; profiles/tools/cpu_fuzz.py --params runs thousands."""
    import ctypes
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(1000 + seed)
    nan = float("nan")
    x = _fuzz_stream(500 + seed, 250000)
    rate = [10000000, 5000000, 2500000, 8000000, 6000000, 10500000][int(rng.integers(6))]
    if rate in (5000000, 2500000):
        x = np.ascontiguousarray(x[::10000000 // rate])
    mask = int(rng.integers(1, 16))

    def pick(a, b):
        return nan if rng.random() < 0.4 else float(np.float32(rng.uniform(a, b)))

    def f4(v):
        return (ctypes.c_float * 4)(*v)
    power = pick(0.002, 0.08)
    corr, lo, hi = [pick(0.05, 1.0) for _ in range(4)], [pick(0.02, 1.0) for _ in range(4)], [pick(0.3, 1.0) for _ in range(4)]
    out = (T.Frame * 16384)()
    n = T.hostsim_lib().hostsim_decode(x.ctypes.data, len(x), 1, rate, seed, mask, power, f4(corr), f4(lo), f4(hi),
                                       ctypes.byref(out), 16384)
    assert 0 <= n <= 16384
    ref, _ = T.reference_decode(x, sample_rate=rate, keep_carrier=True, cap=16384, defined_storage=True,
                                params=T.RefParams(mask, power, f4(corr), f4(lo), f4(hi)))
    assert T.frames_to_tuples(out, n, keep_carrier=True) == ref


# captures found by profiles/tools/long_fuzz.py and cpu_fuzz.py on which the plain reference answers differently from run
# to run: (seed of _fuzz_stream, length, the truncated frame, what the reference reads beyond it)
_TRUNCATED = [
    (31337 + 13 * 292, 250000, (0x103, "057833"), "ATS ending before TB: FWI from frame[3], NfcA.cpp:1736-1769"),
    (225057, 395353, (0x103, "ab019955"), "4-byte answer taken as ATQB: FSDI/FWI from frame[10], frame[11], NfcB.cpp:1186-1187"),
    (220294, 393913, (0x102, "0600ffff00"), "5-byte REQC: time slots from frame[5], NfcF.cpp:1151-1160"),
    (209959, 277521, (0x102, "06"), "1-byte NFC-F poll: command from frame[1], NfcF.cpp:1151-1160"),
]


@pytest.mark.parametrize("seed,length,trigger,what", _TRUNCATED, ids=[t[3].split(":")[0] for t in _TRUNCATED])
def test_truncated_frames_are_classified_as_with_cleared_reference_storage(built, seed, length, trigger, what):
    """The reference classifies these frames from bytes beyond their length, i.e. from leftovers in recycled RawFrame
    storage, and derives the waiting time of later frames from them; with that storage defined (cleared, not recycled:
    oracle/ref_capi.cpp nfcref_decode_defined) it agrees frame for frame with the step machine, whose out-of-frame bytes
    read as zero (nfc_byte)."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = _fuzz_stream(seed, length)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=16384, defined_storage=True)
    got = T.hostsim_decode(x, keep_carrier=True, cap=16384, lane=3)
    assert got == ref
    assert any(f[1] == trigger[0] and f[-1] == bytes.fromhex(trigger[1]) for f in ref), what


def test_defined_reference_storage_holds_under_concurrent_callers(built):
    """The at-size parity runs (tests/parity_sweep_driver.py, bench.py) call nfcref_decode_defined from a pool of threads.
    The reference's frame pool is process-wide: round 3's wrapper cleared its flag when the FIRST of several concurrent
    callers returned, and the others then classified truncated frames from uncleared storage (found by bench.py's parity leg
    over 25 submissions: one NFC-F poll of one byte in 37 000 frames with another frame phase). Every concurrent caller
    must see what a lone caller sees."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    from concurrent.futures import ThreadPoolExecutor
    captures = [_fuzz_stream(seed, length) for seed, length, _, _ in _TRUNCATED]
    alone = [T.reference_decode(x, keep_carrier=True, cap=16384, defined_storage=True)[0] for x in captures]

    # (plain decodes first: they leave used frame storage in the allocator and in the pool)
    for x in captures:
        T.reference_decode(x, keep_carrier=True, cap=16384)
    with ThreadPoolExecutor(max_workers=12) as pool:
        together = list(pool.map(lambda i: T.reference_decode(captures[i % len(captures)], keep_carrier=True, cap=16384, defined_storage=True)[0], range(96)))
    for i, fr in enumerate(together):
        assert fr == alone[i % len(captures)], i


@pytest.mark.parametrize("rate,step", [(5000000, 2), (2500000, 4)])
@pytest.mark.parametrize("name", ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_001", "test_NFC-F_212kbps_002", "test_NFC-V_26kbps_002"])
def test_step_machine_matches_reference_at_other_sample_rates(built, name, rate, step):
    """Symbol periods, delays and protocol timings are all derived from the sample rate (decimated captures)."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    x = np.ascontiguousarray(T.load_fixture(name)[::step])
    ref, _ = T.reference_decode(x, sample_rate=rate, keep_carrier=True)
    got = T.hostsim_decode(x, sample_rate=rate, keep_carrier=True, lane=9)
    assert got == ref
    assert any(f[1] in (0x102, 0x103) for f in ref)


@pytest.mark.parametrize("rate", [8000000, 6000000, 3200000, 2400000, 10500000])
def test_step_machine_matches_reference_on_resampled_captures(built, rate):
    """Captures resampled (linear interpolation) to the rates of the other receivers the reference supports (RTL-SDR 2.4 and
    3.2 MS/s, Airspy 6 MS/s) and to the edge of the history depth: periods, delays and windows all round differently."""
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    frames = 0
    for name in ("test_NFC-A_424kbps_002", "test_NFC-B_106kbps_002", "test_NFC-F_212kbps_003", "test_NFC-V_26kbps_001"):
        x = T.load_fixture(name)
        t = np.arange(int(x.size * rate / 10e6), dtype=np.float64) * (10e6 / rate)
        y = np.interp(t, np.arange(x.size), x).astype(np.float32)
        ref, _ = T.reference_decode(y, sample_rate=rate, keep_carrier=True, cap=16384, defined_storage=True)
        assert T.hostsim_decode(y, sample_rate=rate, keep_carrier=True, cap=16384, lane=7) == ref, name
        frames += sum(f[1] in (0x102, 0x103) for f in ref)
    assert frames > 60


def test_reference_radio_decoder_task_plumbing(tmp_path):
    """BASELINE configs[0]: a fixture through the reference's own RadioDecoderTask (subjects + executor, reference CPU
    decoder underneath, oracle/_ref/task-ref) yields the golden frames; the GPU twin of this test is in
    test_gpu_parity.py with the same harness linked against libnfcgpu.so."""
    exe = os.path.join(T.ROOT, "oracle", "_ref", "task-ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/task-ref not built")
    names = ["test_NFC-A_106kbps_001", "test_POLL_ABF_001"]
    got = T.run_task_harness(exe, names, tmp_path)
    for name in names:
        assert got[name] == T.load_golden(name), name


def test_reference_resampler_oracle_runs(tmp_path):
    """The adaptive-resampler oracle (reference SignalResamplingTask behind its subjects) runs here and obeys the
    invariants of SignalResamplingTask.cpp:168-226: first pair = (first sample, 0), offsets non-decreasing, gaps of at
    most 255 samples, last offset = last sample of the buffer."""
    out = T.reference_resample("test_NFC-A_106kbps_001", tmp_path)
    if out is None:
        pytest.skip("oracle/_ref/resample-ref not built")
    x = T.load_fixture("test_NFC-A_106kbps_001")
    assert len(out) == (x.size + 65535) // 65536
    pos = 0
    for buf in out:
        pairs = buf.reshape(-1, 2)
        n = min(65536, x.size - pos)
        assert pairs[0, 0] == x[pos] and pairs[0, 1] == 0.0
        offsets = pairs[:, 1]
        assert np.all(np.diff(offsets) >= 0) and np.max(np.diff(offsets)) <= 255
        assert offsets[-1] == n - 1
        assert np.array_equal(pairs[:, 0], x[pos + offsets.astype(np.int64)])
        pos += n
