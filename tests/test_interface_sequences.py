"""The lab::NfcDecoder interface (the drop-in seam) driven by random call sequences: setters between
buffers, initialize() in mid-stream, sample-rate changes, empty and invalid buffers. tests/dropin/api_harness.cpp is
linked once with the reference decoder (oracle/_ref/api-ref) and once with the shim on libnfcgpu.so (oracle/_ref/api-gpu);
both run the same script on the same samples and must print the same frames and getter values. On the CPU the shim runs
on the emulated host runtime (tests/test_host_runtime_emulated.py); with -m gpu on the real library and kernels."""
import os
import subprocess

import numpy as np
import pytest

import nfc_testlib as T
from test_oracle_goldens import _fuzz_stream

REF = os.path.join(T.ROOT, "oracle", "_ref", "api-ref")
GPU = os.path.join(T.ROOT, "oracle", "_ref", "api-gpu")
EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")


def _script(seed, total, prefix="", start=0):
    """a random but plausible life of a decoder: mostly buffers of the capture in order, now and then something else"""
    rng = np.random.default_rng(seed)
    lines, pos, rate = [], start, 10000000
    # the start as the reference's callers do it: either nothing (test-sdr: the first buffer brings the rate) or the rate
    # followed by initialize() (RadioDecoderTask: Configure, then Start). initialize() while the rate is still unknown,
    # or the rate without initialize(), leaves the reference with parameters derived from rate 0 (NaN filter weights,
    # no output): the one piece of its behaviour the shim does not reproduce
    def start():
        if rng.random() < 0.6:
            lines.append("rate %d" % rate)
            lines.append("init")

    start()
    fed = False   # the decoder has seen a buffer since it was created
    if rng.random() < 0.5:
        lines.append("time %d" % int(rng.integers(0, 2000000000)))
    while pos < total:
        r = rng.random()
        if prefix and r < 0.01:
            lines.append("drop")   # destroyed; the next line creates a new decoder under the same number
            start()
            fed = False
        elif r < 0.70:
            fed = True
            n = int(rng.choice([65536, 65536, 16384, 4099, 1, 0, int(rng.integers(1, 100000))]))
            lines.append("feed %d %d %d" % (pos, n, rate))
            pos += n
        elif r < 0.74:
            lines.append("enable %s %d" % ("ABFV"[int(rng.integers(4))], int(rng.integers(2))))
        elif r < 0.78:
            lines.append("power %.6f" % rng.uniform(0.003, 0.05))
        elif r < 0.82:
            lines.append("corr %s %s" % ("ABFV"[int(rng.integers(4))], "nan" if rng.random() < 0.15 else "%.6f" % rng.uniform(0.2, 0.9)))
        elif r < 0.86:
            lo = rng.uniform(0.05, 0.9)
            pair = ["%.6f" % lo, "%.6f" % min(1.0, lo + rng.uniform(0.05, 0.6))]
            if rng.random() < 0.2:
                pair[int(rng.integers(2))] = "nan"   # NaN leaves that bound as it is (NfcDecoder.cpp setters)
            lines.append("depth %s %s %s" % ("ABFV"[int(rng.integers(4))], pair[0], pair[1]))
        elif r < 0.90:
            lines.append("init")
        elif r < 0.93:
            lines.append("invalid")
        elif r < 0.96:
            rate = int(rng.choice([10000000, 5000000, 8000000, 10000000]))
            if rng.random() < 0.5 and fed:
                # announced through the setter as well (otherwise the buffers just change); not before the decoder has been
                # initialised by a first buffer: see the note on rate 0 above
                lines.append("rate %d" % rate)
        else:
            lines.append("time %d" % int(rng.integers(0, 2000000000)))
    lines.append("invalid")
    if prefix:
        return [prefix + " " + l for l in lines]
    return "\n".join(lines) + "\n"


def _interleaved_script(seed, total):
    """three decoders side by side (with the shim: three streams of one GPU context, neighbours in one stream block), their
    calls interleaved at random; now and then one is destroyed and created again"""
    rng = np.random.default_rng(seed + 5000)
    lives = [_script(seed * 3 + k, total, prefix="@%d" % k, start=int(rng.integers(0, total // 2))) for k in range(3)]
    lines = []
    while any(lives):
        k = int(rng.integers(3))
        if not lives[k]:
            continue
        take = int(rng.integers(1, 5))
        lines += lives[k][:take]
        lives[k] = lives[k][take:]
    return "\n".join(lines) + "\n"


def _run(exe, raw, script, env=None):
    run = subprocess.run([exe, raw, script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-2000:]
    return run.stdout.splitlines()


def _check(seed, tmp_path, env, interleaved=False):
    x = _fuzz_stream(7000 + seed, 400000)
    raw = str(tmp_path / "x.f32")
    x.tofile(raw)
    script = str(tmp_path / "script.txt")
    with open(script, "w") as f:
        f.write(_interleaved_script(seed, x.size) if interleaved else _script(seed, x.size))
    want = _run(REF, raw, script)
    got = _run(GPU, raw, script, env=env)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a == b
    assert any(l[0] == "F" for l in want)   # frames did come out of it


needs_harness = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)),
                                   reason="api-ref / api-gpu not built (need the reference tree at build time)")


@needs_harness
@pytest.mark.parametrize("seed", range(6))
def test_random_call_sequences_on_the_emulated_runtime(built, seed, tmp_path):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    _check(seed, tmp_path, dict(os.environ, LD_PRELOAD=EMU))


@needs_harness
@pytest.mark.parametrize("seed", range(4))
def test_interleaved_decoders_on_the_emulated_runtime(built, seed, tmp_path):
    if not os.path.exists(EMU):
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    _check(seed, tmp_path, dict(os.environ, LD_PRELOAD=EMU), interleaved=True)


@needs_harness
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(3))
def test_random_call_sequences_on_the_gpu(built, seed, tmp_path):
    _check(seed, tmp_path, None)


@needs_harness
@pytest.mark.gpu
def test_interleaved_decoders_on_the_gpu(built, tmp_path):
    _check(1, tmp_path, None, interleaved=True)
