"""Development check of the time-parallel path on the emulated runtime (CPU): every fixture decoded in one submission
(and in a few large buffers) with the windowed path forced on, compared frame by frame (carrier frames included) with the
reference decoder. Usage: python profiles/tools/r02/windowed_check.py [name-substring] [--buffers N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
os.environ.setdefault("NFCGPU_LIB", os.path.join(ROOT, "tests", "hostsim", "libnfcgpu_emulated.so"))
os.environ.setdefault("NFCGPU_WINDOWED_MIN", "4096")
os.environ.setdefault("NFCGPU_SCAN_CHUNK", "32768")
import numpy as np
import nfc_testlib as T
import nfclab_amd

pick = [a for a in sys.argv[1:] if not a.startswith("--")]
nbuf = 1
if "--buffers" in sys.argv:
    nbuf = int(sys.argv[sys.argv.index("--buffers") + 1])

bad = 0
for name in T.fixture_names():
    if pick and not any(p in name for p in pick):
        continue
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    want, _ = T.reference_decode(mag, keep_carrier=True)
    t0 = time.time()
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        s = gpu.open()
        step = (mag.size + nbuf - 1) // nbuf
        for pos in range(0, mag.size, step):
            gpu.submit(s, np.ascontiguousarray(mag[pos:pos + step]), 10000000)
        got = gpu.poll(s)
        st = gpu.stats()
    ok = got == want
    bad += 0 if ok else 1
    print("%-34s %s frames %3d/%3d  windows %4d passes %d windowed %d fallback %d  %.1fs" % (
        name, "ok  " if ok else "FAIL", len(got), len(want), st.windows, st.window_passes, st.windowed_streams, st.fallback_streams, time.time() - t0))
    if not ok:
        for i, (a, b) in enumerate(zip(got, want)):
            if a != b:
                print("   first difference at frame", i, "\n    got ", a, "\n    want", b)
                break
        else:
            print("   length differs; extra:", (got[len(want):] or want[len(got):])[:2])
print("bad:", bad)
