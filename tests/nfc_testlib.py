"""Shared helpers for the parity tests (fixtures, oracle binding, frame comparison)."""
import ctypes
import json
import lzma
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Frame(ctypes.Structure):
    """Same layout as include/nfcgpu.h nfcgpu_frame, oracle/ref_capi.cpp nfcref_frame."""
    _fields_ = [
        ("stream_id", ctypes.c_uint32),
        ("tech_type", ctypes.c_uint32),
        ("frame_type", ctypes.c_uint32),
        ("frame_flags", ctypes.c_uint32),
        ("frame_phase", ctypes.c_uint32),
        ("frame_rate", ctypes.c_uint32),
        ("length", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("sample_start", ctypes.c_uint64),
        ("sample_end", ctypes.c_uint64),
        ("sample_rate", ctypes.c_uint64),
        ("data", ctypes.c_uint8 * 512),
    ]


def frame_tuple(f):
    """The eight fields RawFrame::operator== compares (lab-data RawFrame.cpp:82-98) + payload."""
    return (f.tech_type, f.frame_type, f.frame_flags, f.frame_phase, f.frame_rate,
            f.sample_start, f.sample_end, f.sample_rate, bytes(f.data[:f.length]))


def frames_to_tuples(arr, n, keep_carrier=False):
    out = []
    for i in range(n):
        t = frame_tuple(arr[i])
        if keep_carrier or t[1] in (0x0102, 0x0103):
            out.append(t)
    return out


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def fixture_names():
    return sorted(manifest().keys())


def load_fixture_i16(name):
    with open(os.path.join(GOLDEN, "wav", name + ".i16.xz"), "rb") as f:
        raw = lzma.decompress(f.read())
    return np.frombuffer(raw, dtype="<i2")


def load_fixture(name):
    """float32 magnitude exactly as hw::RecordDevice reads it: int16 / 32768.0f."""
    return (load_fixture_i16(name).astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def load_golden(name):
    """Golden frames as tuples in frame_tuple() order (test-sdr main.cpp:47-93)."""
    with open(os.path.join(GOLDEN, "wav", name + ".json")) as f:
        data = json.load(f)
    out = []
    for e in data["frames"]:
        payload = bytes(int(x, 16) for x in e["frameData"].split(":")) if e["frameData"] else b""
        out.append((e["techType"], e["frameType"], e["frameFlags"], e["framePhase"], e["frameRate"],
                    e["sampleStart"], e["sampleEnd"], e["sampleRate"], payload))
    return out


class RefParams(ctypes.Structure):
    _fields_ = [
        ("tech_mask", ctypes.c_uint32),
        ("power_level_threshold", ctypes.c_float),
        ("corr_threshold", ctypes.c_float * 4),
        ("min_depth", ctypes.c_float * 4),
        ("max_depth", ctypes.c_float * 4),
    ]


_ref = None


def reference_lib():
    """oracle/_ref/libnfcref.so: the real reference decoder (test infrastructure)."""
    global _ref
    if _ref is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libnfcref.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        lib.nfcref_decode.restype = ctypes.c_long
        lib.nfcref_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
        lib.nfcref_decode_defined.restype = ctypes.c_long
        lib.nfcref_decode_defined.argtypes = lib.nfcref_decode.argtypes
        lib.nfcref_magnitude.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
        _ref = lib
    return _ref


def reference_decode(samples, sample_rate=10000000, chunk=65536, tech_mask=0xF, keep_carrier=False, send_eof=False,
                     cap=4096, params=None, defined_storage=False):
    """The reference decoder on one capture. defined_storage: run it with cleared, unrecycled frame storage
    (oracle/ref_capi.cpp, nfcref_decode_defined): the reference classifies some truncated frames from bytes beyond the
    frame length, which otherwise are leftovers of earlier frames, earlier captures or malloc."""
    lib = reference_lib()
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    out = (Frame * cap)()
    secs = ctypes.c_double(0)
    nan = float("nan")
    p = params or RefParams(tech_mask, nan, (ctypes.c_float * 4)(nan, nan, nan, nan),
                            (ctypes.c_float * 4)(nan, nan, nan, nan), (ctypes.c_float * 4)(nan, nan, nan, nan))
    decode = lib.nfcref_decode_defined if defined_storage else lib.nfcref_decode
    n = decode(samples.ctypes.data, len(samples), sample_rate, chunk, ctypes.byref(p),
               int(keep_carrier), int(send_eof), ctypes.byref(out), cap, ctypes.byref(secs))
    assert 0 <= n <= cap, n
    return frames_to_tuples(out, n, keep_carrier=True), secs.value


_sim = None


def hostsim_lib():
    global _sim
    if _sim is None:
        path = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
        lib = ctypes.CDLL(path)
        lib.hostsim_decode.restype = ctypes.c_long
        lib.hostsim_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_uint32]
        _sim = lib
    return _sim


def hostsim_decode(samples, sample_rate=10000000, lane=0, tech_mask=0xF, stride=1, cap=4096, keep_carrier=False):
    lib = hostsim_lib()
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    out = (Frame * cap)()
    count = len(samples) // stride
    n = lib.hostsim_decode(samples.ctypes.data, count, stride, sample_rate, lane, tech_mask, float("nan"),
                           None, None, None, ctypes.byref(out), cap)
    assert 0 <= n <= cap, n
    return frames_to_tuples(out, n, keep_carrier=keep_carrier)


def write_wav(path, samples_i16, sample_rate=10000000):
    """Mono 16-bit PCM WAV as the reference's hw::RecordDevice reads it."""
    import struct
    raw = np.ascontiguousarray(samples_i16, dtype=np.int16).tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sample_rate, sample_rate * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)


def run_task_harness(exe, names, tmp_path, timeout=900, iq=False):
    """Run tests/dropin/task_harness.cpp (reference RadioDecoderTask driven through its subjects) on fixtures;
    returns {name: [frame tuples in load_golden() order]}."""
    import subprocess
    paths = []
    for name in names:
        wav = os.path.join(str(tmp_path), name + ".wav")
        write_wav(wav, load_fixture_i16(name))
        paths.append(wav)
    proc = subprocess.run([exe] + (["--iq"] if iq else []) + paths, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-2000:]
    out = {name: [] for name in names}
    done = set()
    for line in proc.stdout.splitlines():
        w = line.split()
        if w and w[0] == "FRAME":
            name = os.path.basename(w[1])[:-4]
            payload = bytes.fromhex(w[10]) if w[10] != "-" else b""
            out[name].append((int(w[2]), int(w[3]), int(w[4]), int(w[5]), int(w[6]), int(w[7]), int(w[8]), int(w[9]), payload))
        elif w and w[0] == "DONE":
            done.add(os.path.basename(w[1])[:-4])
    assert done == set(names), proc.stdout[-2000:] + proc.stderr[-2000:]
    return out


def reference_resample(name, tmp_path, per_buffer=65536):
    """Control points published by the reference's SignalResamplingTask for a fixture (oracle/_ref/resample-ref: the
    real task behind its subjects); returns None when the oracle binary is absent, else a list of float32 arrays."""
    import struct
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "resample-ref")
    if not os.path.exists(exe):
        return None
    wav = os.path.join(str(tmp_path), name + ".wav")
    dst = os.path.join(str(tmp_path), name + ".adaptive.bin")
    write_wav(wav, load_fixture_i16(name))
    proc = subprocess.run([exe, wav, dst, str(per_buffer)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    raw = open(dst, "rb").read()
    out, pos = [], 0
    while pos < len(raw):
        count = struct.unpack_from("<I", raw, pos)[0]
        pos += 4
        out.append(np.frombuffer(raw, dtype=np.float32, count=count, offset=pos).copy())
        pos += 4 * count
    return out


def describe(t):
    return "tech=%x type=%x flags=%x phase=%x rate=%d start=%d end=%d data=%s" % (
        t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[8].hex(":"))


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def magnitude_to_iq(mag, seed=0, period=4096):
    """Interleaved IQ whose magnitude is exactly `mag`: the sample is put on +I, +Q, -I or -Q, the axis
    advancing every `period` samples (SURVEY.md 8(d) set S1). sqrtf(m*m + 0) == |m| exactly in fp32."""
    mag = np.ascontiguousarray(mag, dtype=np.float32)
    n = mag.size
    phase = ((np.arange(n) // period) + (splitmix64(seed) & 3)) & 3
    iq = np.zeros(2 * n, dtype=np.float32)
    i = np.where(phase == 0, mag, np.where(phase == 2, -mag, 0)).astype(np.float32)
    q = np.where(phase == 1, mag, np.where(phase == 3, -mag, 0)).astype(np.float32)
    iq[0::2] = i
    iq[1::2] = q
    return iq


def synthetic_stream_i16(stream, length, names=None):
    """Deterministic synthetic capture: fixtures (seeded choice, circular shift) tiled to `length` samples
    with an integer gain of 3/4 or 1 on the int16 grid; values stay exact multiples of 2^-15."""
    names = names or fixture_names()
    state = splitmix64(0x9E3779B97F4A7C15 * (stream + 1) & 0xFFFFFFFFFFFFFFFF)
    out = np.empty(length, dtype=np.int16)
    pos = 0
    while pos < length:
        state = splitmix64(state)
        src = load_fixture_i16(names[state % len(names)])
        state = splitmix64(state)
        shift = state % src.size
        state = splitmix64(state)
        piece = np.roll(src, -int(shift))
        if state & 1:
            piece = ((piece.astype(np.int32) * 3) // 4).astype(np.int16)
        n = min(length - pos, piece.size)
        out[pos:pos + n] = piece[:n]
        pos += n
    return out


def crc_iso15693(data):
    """CRC of ISO/IEC 15693 frames (ISO/IEC 13239: reflected 0x1021, preset 0xFFFF, inverted), low byte first"""
    crc = 0xFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    crc ^= 0xFFFF
    return bytes([crc & 0xFF, crc >> 8])


def synth_nfcv_poll(payload, mode, lead=30000, tail=60000, level=0.5, depth=0.97, noise=0.0005, seed=1, sample_rate=10000000):
    """Synthetic ISO 15693 reader frame (pulse-position coding, 9.44 us pauses): SOF, `payload` in 1-of-4 (mode 4) or
    1-of-256 (mode 256) coding, EOF, on an unmodulated carrier. No fixture of the reference uses 1-of-256."""
    unit = 9.44e-6 * sample_rate           # one half slot
    pauses = [(0, 1), (7, 8)] if mode == 256 else [(0, 1), (5, 6)]
    t = 8
    for b in payload:
        if mode == 256:
            pauses.append((t + 2 * b + 1, t + 2 * b + 2))
            t += 512
        else:
            for k in range(4):
                v = (b >> (2 * k)) & 3
                pauses.append((t + 2 * v + 1, t + 2 * v + 2))
                t += 8
    pauses.append((t + 2, t + 3))
    t += 4
    x = np.full(int(lead + t * unit + tail), level, np.float32)
    for a, b in pauses:
        x[int(round(lead + a * unit)):int(round(lead + b * unit))] = level * (1 - depth)
    x = np.convolve(x, np.array([0.25, 0.5, 0.25], np.float32), mode="same").astype(np.float32)
    x += np.random.default_rng(seed).normal(0, noise, x.size).astype(np.float32)
    x[:200] *= np.linspace(0, 1, 200, dtype=np.float32)
    return x
