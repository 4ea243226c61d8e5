#!/usr/bin/env python3
"""Measurement of nfcgpu_resample_radio on one GPU: N magnitude buffers of L samples resident in
HBM (the synthetic streams of bench.py), HIP-event time of the call, algorithmic bytes = 4 B read per sample + 8 B per
control point written; the reference's SignalResamplingTask (oracle/_ref/resample-ref) timed on one host core beside it.
Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nfclab_amd  # noqa: E402
import synth  # noqa: E402
import nfc_testlib as T  # noqa: E402

N = int(os.environ.get("RS_BUFFERS", "65536"))
L = int(os.environ.get("RS_SAMPLES", "8192"))
dev = torch.device("cuda:0")

template = torch.from_numpy(synth.load_template(os.path.join(ROOT, "tests", "golden")).astype(np.int16)).to(dev)
iq = torch.empty((N, L, 2), dtype=torch.float32, device=dev)
synth.fill_iq_torch(iq, template, first_stream=0)
mag = torch.sqrt(iq[:, :, 0] ** 2 + iq[:, :, 1] ** 2).contiguous()
del iq
cap = L + L // 255 + 2
out = torch.zeros((N, 2 * cap), dtype=torch.float32, device=dev)
counts = torch.zeros(N, dtype=torch.int32, device=dev)
torch.cuda.synchronize()

with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
    for _ in range(2):
        gpu.resample_radio_device(mag.data_ptr(), L * 4, N, L, out.data_ptr(), 2 * cap * 4, cap, counts.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gpu.resample_radio_device(mag.data_ptr(), L * 4, N, L, out.data_ptr(), 2 * cap * 4, cap, counts.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3

pairs = int(counts.sum().item())
alg = N * L * 4 + pairs * 8
result = {"op": "nfcgpu_resample_radio", "buffers": N, "samples_per_buffer": L, "ms_per_call": round(ms, 3),
          "msamples_per_s": round(N * L / ms / 1e3, 1), "control_points": pairs,
          "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(alg / ms / 1e6 / 8000.0, 4)}}

exe = os.path.join(ROOT, "oracle", "_ref", "resample-ref")
if os.path.exists(exe):
    with tempfile.TemporaryDirectory() as tmp:
        name = "test_POLL_ABF_001"
        wav = os.path.join(tmp, name + ".wav")
        x = T.load_fixture_i16(name)
        T.write_wav(wav, np.tile(x, 8))
        t0 = time.perf_counter()
        subprocess.run([exe, wav, os.path.join(tmp, "o.bin")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        secs = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": round(x.size * 8 / secs / 1e6, 1), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                                  "sample": "reference SignalResamplingTask on %d samples (WAV read and task plumbing included, %.1f s)" % (x.size * 8, secs)}
print(json.dumps(result))
