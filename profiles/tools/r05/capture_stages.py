#!/usr/bin/env python3
"""Where the time of a short capture goes (one stream, magnitudes resident in HBM, second decode of each length timed):
NFCGPU_WINDOW_DEBUG=1 stage lines on stderr, wall milliseconds on stdout. usage: capture_stages.py [capture names...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np, torch
import nfc_testlib as T, nfclab_amd
names = sys.argv[1:] or ["test_NFC-A_106kbps_002", "test_NFC-B_106kbps_001", "test_NFC-A_424kbps_001", "test_NFC-A_106kbps_001"]
dev = torch.device("cuda", 0)
g = nfclab_amd.NfcGpu(device=0, max_streams=64, frame_sink_bytes=8 << 20)
for name in names:
    mag = torch.from_numpy(T.load_fixture(name)).to(dev)
    n = int(mag.numel())
    torch.cuda.synchronize()
    for attempt in range(2):
        sid = g.open(nfclab_amd.default_params(), count=1)
        g.sync()
        sys.stderr.write("=== %s attempt %d (%d samples)\n" % (name, attempt, n)); sys.stderr.flush()
        ta = time.perf_counter()
        g.submit_uniform(sid, 1, mag.data_ptr(), n * 4, n, 10000000, stride=1)
        g.sync()
        tb = time.perf_counter()
        got = g.poll(sid, capacity=1 << 16)
        g.close_stream(sid)
    print(name, n, "samples", round((tb - ta) * 1e3, 2), "ms", round(n / (tb - ta) / 1e6, 2), "MS/s", "golden" if [f for f in got if f[1] in (0x102, 0x103)] == T.load_golden(name) else "DIFFERS")
g.close()
