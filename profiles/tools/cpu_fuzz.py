#!/usr/bin/env python3
"""Randomized parity of the device step machine (tests/hostsim: the product's device code compiled for the CPU) with the
reference decoder, far beyond the test suite and without a GPU. Each capture is a random cut-and-paste of the fixtures
(general fp32: arbitrary gains, offsets, noise), optionally with random decoder parameters, tech masks and sample rates.
The yardstick is the reference with defined frame storage (oracle/ref_capi.cpp, nfcref_decode_defined); the plain
reference, whose answer for truncated frames depends on leftovers in recycled storage, is counted beside it.

  cpu_fuzz.py FIRST LAST [--params] [--timescale]     prints one JSON line

--timescale: pieces are also decimated or linearly interpolated by 2 or 4, which turns the fixtures into traffic at the
rates no fixture covers (NFC-B 212 kbps, NFC-F 424 kbps listen frames, NFC-A 212/424 kbps polls from the 106 kbps ones).
"""
import ctypes
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nfc_testlib as T  # noqa: E402
from test_oracle_goldens import _fuzz_stream  # noqa: E402

NAN = float("nan")
RATES = [10000000, 10000000, 10000000, 5000000, 2500000, 8000000, 6000000, 10500000]
CAP = 32768
PARAMS = "--params" in sys.argv
TIMESCALE = "--timescale" in sys.argv


def _up2(x):
    y = np.empty(x.size * 2, np.float32)
    y[0::2] = x
    y[1::2] = np.concatenate([(x[:-1] + x[1:]) * np.float32(0.5), x[-1:]])
    return y


def _scaled_stream(seed, length):
    """_fuzz_stream with each piece played at x1, x2, x4, /2 or /4 speed"""
    rng = np.random.default_rng(seed)
    names = T.fixture_names()
    out = np.empty(length, np.float32)
    pos = 0
    while pos < length:
        x = T.load_fixture(names[rng.integers(len(names))])
        n = int(rng.integers(20000, 200000))
        a = int(rng.integers(0, max(1, x.size - n)))
        piece = x[a:a + n]
        how = int(rng.integers(0, 6))
        if how == 1:
            piece = piece[::2]
        elif how == 2:
            piece = piece[::4]
        elif how == 3:
            piece = _up2(piece)
        elif how == 4:
            piece = _up2(_up2(piece[:n // 2]))
        piece = piece * np.float32(rng.uniform(0.5, 1.5)) + np.float32(rng.uniform(-0.003, 0.003))
        piece = piece + rng.normal(0, rng.uniform(0, 0.002), piece.size).astype(np.float32)
        m = min(length - pos, piece.size)
        out[pos:pos + m] = piece[:m]
        pos += m
    return out


def _f4(v):
    return (ctypes.c_float * 4)(*v)


def one(seed):
    rng = np.random.default_rng(seed * 7 + 1)
    x = (_scaled_stream if TIMESCALE else _fuzz_stream)(seed, int(rng.integers(100000, 400000)))
    rate, mask, power = 10000000, 0xF, NAN
    corr, lo, hi = [NAN] * 4, [NAN] * 4, [NAN] * 4
    if PARAMS:
        rate = RATES[int(rng.integers(len(RATES)))]
        if rate == 5000000:
            x = np.ascontiguousarray(x[::2])
        if rate == 2500000:
            x = np.ascontiguousarray(x[::4])
        mask = int(rng.integers(1, 16))

        def pick(a, b):
            return NAN if rng.random() < 0.5 else float(np.float32(rng.uniform(a, b)))
        power = pick(0.002, 0.08)
        corr = [pick(0.05, 1.0) for _ in range(4)]
        lo = [pick(0.02, 1.0) for _ in range(4)]
        hi = [pick(0.3, 1.0) for _ in range(4)]
    out = (T.Frame * CAP)()
    n = T.hostsim_lib().hostsim_decode(x.ctypes.data, len(x), 1, rate, int(seed % 64), mask, power, _f4(corr), _f4(lo),
                                       _f4(hi), ctypes.byref(out), CAP)
    if n == -1:
        return seed, 0, "rate not decodable", None
    assert 0 <= n <= CAP, n
    got = T.frames_to_tuples(out, n, keep_carrier=True)
    kw = dict(sample_rate=rate, keep_carrier=True, cap=CAP)
    ref, _ = T.reference_decode(x, params=T.RefParams(mask, power, _f4(corr), _f4(lo), _f4(hi)), defined_storage=True, **kw)
    plain, _ = T.reference_decode(x, params=T.RefParams(mask, power, _f4(corr), _f4(lo), _f4(hi)), **kw)
    detail = None
    if got != ref:
        k = next((j for j, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
        detail = [k, T.describe(got[k]) if k < len(got) else None, T.describe(ref[k]) if k < len(ref) else None]
    return seed, len(ref), "ok" if got == ref else "MISMATCH", (plain == ref, detail)


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    res = []
    with Pool(os.cpu_count(), maxtasksperchild=64) as pool:
        for r in pool.imap_unordered(one, range(first, last), chunksize=4):
            res.append(r)
            if r[2] == "MISMATCH":
                print(r, file=sys.stderr, flush=True)
    done = [r for r in res if r[3] is not None]
    print(json.dumps({"tool": "profiles/tools/cpu_fuzz.py", "seeds": [first, last], "random_parameters": PARAMS, "time_scaled_pieces": TIMESCALE,
                      "captures": len(done), "rate_not_decodable": len(res) - len(done),
                      "reference_frames": sum(r[1] for r in done),
                      "mismatching_defined_storage_reference": sum(r[2] == "MISMATCH" for r in done),
                      "captures_where_plain_reference_differs_from_itself_with_defined_storage":
                          sum(not r[3][0] for r in done),
                      "seconds": round(time.time() - t0, 1)}))
