#!/usr/bin/env python3
"""Runs the tile-major schedule prototype (tilemajor_sim.cpp) against the reference decoder: fixtures, then fuzzed captures.
  tilemajor_check.py [tile] [first_seed last_seed]"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nfc_testlib as T  # noqa: E402
from test_oracle_goldens import _fuzz_stream  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = "/tmp/libtilemajor.so"
subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-msse3", "-mno-avx", "-fPIC", "-shared",
                       "-Wno-unknown-pragmas", os.path.join(HERE, "tilemajor_sim.cpp"), "-o", LIB])
lib = ctypes.CDLL(LIB)


class Stats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("searchSegments", "locks", "takeBacks", "detectorSteps", "samples")]


lib.tilemajor_decode.restype = ctypes.c_long
lib.tilemajor_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(Stats)]
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 200)
total = Stats()
bad = 0
frames = 0


def check(x, tag):
    global bad, frames
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = (T.Frame * 32768)()
    st = Stats()
    n = lib.tilemajor_decode(x.ctypes.data, len(x), 10000000, 5, 0xF, tile, ctypes.byref(out), 32768, ctypes.byref(st))
    assert 0 <= n <= 32768, (tag, n)
    got = T.frames_to_tuples(out, n, keep_carrier=True)
    ref, _ = T.reference_decode(x, keep_carrier=True, cap=32768, defined_storage=True)
    frames += len(ref)
    for f, _ in Stats._fields_:
        setattr(total, f, getattr(total, f) + getattr(st, f))
    if got != ref:
        bad += 1
        k = next((j for j, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
        print("MISMATCH", tag, "frame", k, T.describe(got[k]) if k < len(got) else None, "|", T.describe(ref[k]) if k < len(ref) else None)


for name in T.fixture_names():
    check(T.load_fixture(name), name)
print("fixtures done, mismatching:", bad)
for seed in range(lo, hi):
    check(_fuzz_stream(seed, 300000), seed)
print({"tile": tile, "captures": len(T.fixture_names()) + hi - lo, "reference_frames": frames, "mismatching": bad,
       **{f: getattr(total, f) for f, _ in Stats._fields_}})
