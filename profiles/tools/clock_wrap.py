#!/usr/bin/env python3
"""Parity across the 2^32 wrap of the sample clock (7 minutes of signal per stream at 10 MS/s), CPU only: the reference
offers no way to preset its clock, so both it and the CPU build of the device step machine (tests/hostsim) are fed
2^32 - OFFSET samples of idle carrier (one 65536-sample buffer over and over) and then a fixture that straddles the
wrap; frames are compared bit for bit, carrier frames included. Minutes of CPU per capture.

  clock_wrap.py NAME [NAME ...]     prints one JSON line per capture
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

FS = 10000000
IDLE = 65536
CAP = 32768
NAN = float("nan")


def idle_buffer():
    """unmodulated carrier at the level of the fixtures with 2-LSB deterministic noise: every detector searches, none locks"""
    n = np.arange(IDLE, dtype=np.uint64)
    return (np.float32(0.25) + ((n * 2654435761) % 5).astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def one(name):
    t0 = time.time()
    x = T.load_fixture(name)
    idle = idle_buffer()
    # the wrap falls inside the fixture: `lead` samples of the fixture come before it
    golden = T.load_golden(name)
    lead = int(golden[len(golden) // 2][5]) + 37 if golden else x.size // 2
    total = int(os.environ.get("CLOCK_WRAP_TOTAL", str(1 << 32)))  # smaller only to try the plumbing
    repeats = (total - lead) // IDLE
    head = total - lead - repeats * IDLE          # idle samples that do not fill a whole buffer
    capture = np.ascontiguousarray(np.concatenate([idle[:head], x]))
    wrap_at = head + lead                              # index in `capture` of the sample with clock 0 again
    f4 = (ctypes.c_float * 4)(NAN, NAN, NAN, NAN)
    p = T.RefParams(0xF, NAN, f4, f4, f4)

    ref_lib = T.reference_lib()
    ref_lib.nfcref_decode_after_idle.restype = ctypes.c_long
    ref_lib.nfcref_decode_after_idle.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_uint32]
    out = (T.Frame * CAP)()
    n = ref_lib.nfcref_decode_after_idle(idle.ctypes.data, IDLE, repeats, capture.ctypes.data, capture.size, FS, 65536,
                                         ctypes.byref(p), 1, ctypes.byref(out), CAP)
    assert 0 <= n <= CAP, n
    ref = T.frames_to_tuples(out, n, keep_carrier=True)
    t1 = time.time()

    sim = T.hostsim_lib()
    sim.hostsim_decode_after_idle.restype = ctypes.c_long
    sim.hostsim_decode_after_idle.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                              ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    out2 = (T.Frame * CAP)()
    m = sim.hostsim_decode_after_idle(idle.ctypes.data, IDLE, repeats, capture.ctypes.data, capture.size, 1, FS, 5, 0xF, NAN,
                                      None, None, None, ctypes.byref(out2), CAP)
    assert 0 <= m <= CAP, m
    got = T.frames_to_tuples(out2, m, keep_carrier=True)
    t2 = time.time()

    data = [f for f in ref if f[1] in (0x102, 0x103)]
    after = [f for f in data if f[5] < (1 << 31)]     # sample stamps are 32-bit clocks: small again after the wrap
    straddling = [f for f in data if f[5] > f[6]]
    return {"capture": name, "idle_samples": repeats * IDLE + head, "wrap_at_fixture_sample": lead,
            "reference_frames": len(ref), "data_frames": len(data), "data_frames_after_wrap": len(after),
            "frames_straddling_wrap": len(straddling), "equal": got == ref,
            "golden_frames": len(golden), "reference_seconds": round(t1 - t0, 1), "step_machine_seconds": round(t2 - t1, 1)}


if __name__ == "__main__":
    names = sys.argv[1:] or ["test_NFC-A_106kbps_001"]
    with Pool(min(len(names), os.cpu_count())) as pool:
        for r in pool.imap_unordered(one, names):
            print(json.dumps(r), flush=True)
