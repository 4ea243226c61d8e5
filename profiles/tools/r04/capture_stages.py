"""Where the time of one short capture goes: the capture decoded three times as one 10 MS/s stream (magnitudes resident in HBM),
the third time with NFCGPU_WINDOW_DEBUG-style marks read from the library's stderr. usage: python capture_stages.py <fixture> [...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np, torch
import nfc_testlib as T, nfclab_amd
FS = 10000000
for name in sys.argv[1:]:
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    dev = torch.from_numpy(mag).cuda()
    torch.cuda.synchronize()
    for rep in range(3):
        with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
            first = gpu.open(count=1)
            gpu.sync()
            if rep == 2:
                sys.stderr.write("---- %s (%d samples), third decode ----\n" % (name, mag.size)); sys.stderr.flush()
            t0 = time.perf_counter()
            gpu.submit_uniform(first, 1, dev.data_ptr(), mag.size * 4, mag.size, FS, stride=1)
            gpu.sync()
            t1 = time.perf_counter()
            n = len(gpu.poll(first, capacity=4096))
        print("%s rep %d: %.2f ms, %d frames" % (name, rep, (t1 - t0) * 1e3, n), flush=True)
