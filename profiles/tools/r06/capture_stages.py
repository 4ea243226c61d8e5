#!/usr/bin/env python3
"""Round 6: stage log (NFCGPU_WINDOW_DEBUG=2) of single bundled captures decoded as one submission each: where the time of the
slowest of BASELINE's configs 2-4 goes. usage: capture_stages.py name [name ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np
import nfc_testlib as T, nfclab_amd
for name in sys.argv[1:]:
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        for rep in range(3):
            sid = gpu.open()
            if rep == 2:
                os.environ["NFCGPU_WINDOW_DEBUG"] = os.environ.get("CAPTURE_STAGES_LEVEL", "2")
                sys.stderr.write("==== %s: %d samples\n" % (name, mag.size))
            t0 = time.perf_counter()
            gpu.submit(sid, mag, 10000000)
            fr = gpu.poll(sid, capacity=1 << 16)
            dt = time.perf_counter() - t0
            if rep == 2:
                del os.environ["NFCGPU_WINDOW_DEBUG"]
                sys.stderr.write("==== %s: %.2f ms, %d frames\n" % (name, dt * 1e3, len(fr)))
            gpu.close_stream(sid)
