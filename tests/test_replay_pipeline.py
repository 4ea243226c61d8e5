"""The application's replay pipeline: the reference's SignalStorageTask reads a two-channel I/Q WAV,
turns it into magnitudes itself (SignalStorageTask.cpp:372-440, the SSE2 twin of RadioDeviceTask's conversion) and
publishes radio.signal.raw, which the reference's RadioDecoderTask decodes (tests/dropin/replay_harness.cpp: both tasks
compiled where they lie, in one executor). Linked with the reference decoder (replay-ref) and with the shim (replay-gpu)
the pipeline must give the same frames; and the magnitudes the storage task computed are the yardstick of the IQ -> magnitude
step on our side (nfcgpu_magnitude, the kernels' input staging): bit for bit."""
import os
import struct
import subprocess

import numpy as np
import pytest

import nfc_testlib as T

REF = os.path.join(T.ROOT, "oracle", "_ref", "replay-ref")
GPU = os.path.join(T.ROOT, "oracle", "_ref", "replay-gpu")
EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")
FS = 10000000

needs_harness = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)),
                                   reason="replay-ref / replay-gpu not built (need the reference tree at build time)")


def _iq_wav(path, name):
    """the capture as I/Q with a slowly turning phase, rounded to int16: general I/Q pairs whose magnitude is close to, not
    equal to, the original. The length is cut to a multiple of 8 samples: the reference's SSE2 loop writes whole groups
    of 8 magnitudes and runs past its buffer otherwise."""
    m = T.load_fixture_i16(name).astype(np.float64)
    m = m[:m.size // 8 * 8]
    phase = 2 * np.pi * np.arange(m.size) / 5000.0
    iq = np.empty(2 * m.size, np.int16)
    iq[0::2] = np.clip(np.rint(m * np.cos(phase)), -32768, 32767)
    iq[1::2] = np.clip(np.rint(m * np.sin(phase)), -32768, 32767)
    raw = iq.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, FS, FS * 4, 4, 16))
        f.write(b"data" + struct.pack("<I", len(raw)) + raw)
    return iq


def _replay(exe, wav, dump=None, env=None):
    run = subprocess.run([exe, wav, str(FS)] + ([dump] if dump else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=900, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = run.stdout.splitlines()
    assert lines and lines[-1].startswith("DONE")
    return lines


def _check(tmp_path, env, magnitude):
    for name in ("test_POLL_ABF_001", "test_NFC-V_26kbps_002"):
        wav = str(tmp_path / (name + "_iq.wav"))
        iq = _iq_wav(wav, name)
        dump = str(tmp_path / "magnitudes.f32")
        want = _replay(REF, wav, dump=dump)
        got = _replay(GPU, wav, env=env)
        assert got == want
        assert sum(l.startswith("FRAME") and l.split()[2] in ("258", "259") for l in want) >= 4

        reference_magnitudes = np.fromfile(dump, np.float32)
        assert reference_magnitudes.size == iq.size // 2
        floats = iq.astype(np.float32) / np.float32(32768.0)
        assert np.array_equal(magnitude(floats), reference_magnitudes)


@needs_harness
def test_replay_pipeline_on_the_emulated_runtime(built, tmp_path):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])

    def magnitude(iq):
        # the oracle's restatement of the scalar formula (ref_capi.cpp): pinned here against the reference's own results
        out = np.empty(iq.size // 2, np.float32)
        T.reference_lib().nfcref_magnitude(iq.ctypes.data, iq.size // 2, out.ctypes.data)
        return out

    _check(tmp_path, dict(os.environ, LD_PRELOAD=EMU), magnitude)


@needs_harness
@pytest.mark.gpu
def test_replay_pipeline_on_the_gpu(built, tmp_path):
    import nfclab_amd
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        _check(tmp_path, None, lambda iq: gpu.magnitude(np.ascontiguousarray(iq.reshape(-1, 2))))
