"""Where the time of one short capture goes: each bundled capture as one stream, one submission, kernel spans from the
library's own event pairs (nfcgpu_profile) beside the wall clock of submit..sync. Run on the GPU box from the repo root."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))

import torch

import nfc_testlib as TL
import nfclab_amd

dev = torch.device("cuda:0")
g = nfclab_amd.NfcGpu(device=0, max_streams=64, frame_sink_bytes=8 << 20)
g.profile(True)
for name in TL.fixture_names():
    mag = torch.from_numpy(TL.load_fixture(name)).to(dev)
    n = int(mag.numel())
    torch.cuda.synchronize()
    for attempt in range(3):
        sid = g.open(nfclab_amd.default_params(), count=1)
        g.sync()
        g.stats_reset()
        ta = time.perf_counter()
        g.submit_uniform(sid, 1, mag.data_ptr(), n * 4, n, 10000000, stride=1)
        tm = time.perf_counter()
        g.sync()
        tb = time.perf_counter()
        st = g.stats()
        got = g.poll(sid, capacity=1 << 16)
        g.close_stream(sid)
    print("%-28s n=%7d wall %7.3f ms (submit call %6.3f) scan %6.3f planes %6.3f wave %7.3f (%d launches) windowed %6.3f seq %6.3f lanes %d passes %d repairs %d frames %d" % (
        name, n, (tb - ta) * 1e3, (tm - ta) * 1e3, st.scan_ms, st.planes_ms, st.wave_ms, st.wave_launches, st.window_ms, st.kernel_ms,
        st.windows, st.window_passes, st.scan_repairs, len(got)), flush=True)
g.close()
