#!/usr/bin/env python3
"""Long randomized parity run on the GPU box (beyond the test suite): N fuzzed captures (random cut-and-paste of the
fixtures with arbitrary gains, offsets and noise: general fp32, not on the int16 grid) decoded in ragged batches through
the C ABI and compared frame by frame, carrier frames included, with the reference decoder run with defined frame
storage (oracle/ref_capi.cpp, nfcref_decode_defined: the plain reference classifies some truncated frames from leftovers
beyond the frame length, so its answer depends on what the process decoded before). Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import nfc_testlib as T  # noqa: E402
import nfclab_amd  # noqa: E402
from test_oracle_goldens import _fuzz_stream  # noqa: E402

N = int(os.environ.get("FUZZ_STREAMS", "768"))
L = int(os.environ.get("FUZZ_SAMPLES", "300000"))
SEED = int(os.environ.get("FUZZ_SEED", "20260925"))
FS = 10000000

rng = np.random.default_rng(SEED)
t0 = time.time()
streams = [_fuzz_stream(SEED + 13 * i, L) for i in range(N)]
bad = []
undefined = []   # captures on which the plain reference differs from itself with defined frame storage
frames = 0
with nfclab_amd.NfcGpu(device=0, max_streams=N, frame_sink_bytes=256 << 20) as gpu:
    first = gpu.open(count=N)
    fed = [0] * N
    trace = {int(k): [] for k in os.environ.get("FUZZ_TRACE", "").split(",") if k}
    while any(f < L for f in fed):
        ids, ptrs, cnts, keep = [], [], [], []
        for i in range(N):
            if fed[i] >= L or rng.random() < 0.1:
                continue
            c = int(min(L - fed[i], rng.integers(1, 40000)))
            part = np.ascontiguousarray(streams[i][fed[i]:fed[i] + c])
            keep.append(part)
            ids.append(first + i); ptrs.append(part.ctypes.data); cnts.append(c)
            fed[i] += c
            if i in trace:
                trace[i].append(c)
        if ids:
            gpu.submit_batch(ids, ptrs, cnts, FS)
    for i in range(N):
        ref, _ = T.reference_decode(streams[i], keep_carrier=True, cap=32768, defined_storage=True)
        plain, _ = T.reference_decode(streams[i], keep_carrier=True, cap=32768)
        got = gpu.poll(first + i, capacity=32768)
        frames += len(ref)
        if plain != ref:
            undefined.append(i)
        if got != ref:
            bad.append(i)
            if os.environ.get("FUZZ_DUMP"):
                k = next((j for j, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
                print("stream", i, "frames", len(got), "vs", len(ref), "first difference at", k, file=sys.stderr)
                for j in range(max(0, k - 1), min(k + 3, max(len(got), len(ref)))):
                    print("   got ", T.describe(got[j]) if j < len(got) else None, file=sys.stderr)
                    print("   want", T.describe(ref[j]) if j < len(ref) else None, file=sys.stderr)
if trace:
    json.dump(trace, open(os.path.join(ROOT, "gpurun_out", "fuzz_trace.json"), "w"))
print(json.dumps({"streams": N, "samples_per_stream": L, "seed": SEED, "reference_frames": frames,
                  "streams_mismatching": len(bad), "first_bad": bad[:8],
                  "plain_reference_differs_from_defined_storage": undefined[:8], "seconds": round(time.time() - t0, 1)}))
