#!/usr/bin/env python3
"""VERDICT r05 item 5, measured on the same work (round 6): the pieces the first decode pass of the headline runs - a wavefront
per piece (nfc_wave_kernel) - decoded again as independent streams by the sequential kernel (nfc_demod_fixed_kernel: a lane per
piece, 64 pieces per wavefront, rings in HBM), through the C ABI.

  1. the headline's shape (4096 dense streams x 2^20, IQ resident) is submitted to the time-parallel path of the tuning build,
     the second submission with NFCGPU_WINDOW_DEBUG=1 (stage log: what pass 0 took) and NFCGPU_DUMP_WINDOWS (every piece of pass 0
     as it ended: stream, first sample, first sample not consumed);
  2. a second context (NFCGPU_WINDOWED=0: sequential kernels only) opens one stream per piece and is given the very sample ranges
     (device pointers into the same IQ, longest pieces first so that the 64 lanes of a wavefront have pieces of about one length),
     twice; the second submission is timed.

What the lanes do is not identical (a sequential lane walks the front end itself and starts as a fresh stream, a piece starts at
rest with 768 samples of warm-up taken from the planes) but it is the same samples through the same step machine: the figure
is what the *shape* - 64 independent pieces per wavefront against one - does to the pass, ragged lengths and divergence included.
Usage (GPU box): NFCGPU_LIB=nfc-laboratory_amd/libnfcgpu_tuning.so python profiles/tools/r06/item5_pieces_ab.py [--streams 4096]"""
import argparse
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
sys.path.insert(0, ROOT)

FS = 10000000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--cap", type=int, default=0, help="pieces longer than this many samples are cut off at it (0: as they ran)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import nfclab_amd
    import synth

    dev = torch.device("cuda", 0)
    S, L = args.streams, args.samples
    template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
    template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)
    data = torch.empty((S, 2 * L, 2), dtype=torch.float32, device=dev)
    synth.fill_iq_torch(data, template_dev, first_stream=0, chunk_streams=max(1, min(1024, (1 << 27) // (2 * L))))
    torch.cuda.synchronize()
    pitch = 2 * L * 8

    dump = tempfile.mktemp(suffix=".windows")
    log = tempfile.mktemp(suffix=".log")

    # ---- 1. the time-parallel path, pieces dumped ----
    words = 64 << 20
    sink = torch.zeros(words, dtype=torch.int32, device=dev)
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    gpu = nfclab_amd.NfcGpu(device=0, max_streams=S, frame_sink_bytes=1 << 20)
    gpu.sink_attach(sink.data_ptr(), words, ctl.data_ptr())
    gpu.sink_hold(True)
    first = gpu.open(nfclab_amd.default_params(), count=S)
    gpu.submit_uniform(first, S, data.data_ptr(), pitch, L, FS, stride=2)
    gpu.sync()

    os.environ["NFCGPU_DUMP_WINDOWS"] = dump
    os.environ["NFCGPU_WINDOW_DEBUG"] = "1"
    sys.stderr.flush()
    saved = os.dup(2)
    fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(fd, 2)
    t0 = time.perf_counter()
    gpu.submit_uniform(first, S, data.data_ptr() + L * 8, pitch, L, FS, stride=2)
    gpu.sync()
    t1 = time.perf_counter()
    os.dup2(saved, 2)
    os.close(fd)
    del os.environ["NFCGPU_DUMP_WINDOWS"]
    del os.environ["NFCGPU_WINDOW_DEBUG"]
    gpu.close()
    del sink

    stage = open(log).read()
    m = re.search(r"windowed pass 0: (\d+) lanes, (\d+) tiles of \d+ \(([\d.]+) tiles per us\).*?, ([\d.]+) ms", stage)
    wave = {"lanes": int(m.group(1)), "tiles": int(m.group(2)), "tiles_per_us": float(m.group(3)), "pass0_ms": float(m.group(4))} if m else None

    rec = np.fromfile(dump, dtype=np.uint32).reshape(-1, 5)
    os.unlink(dump)
    job, start, stop = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64), rec[:, 3].astype(np.int64)
    keep = stop > start
    job, start, stop = job[keep], start[keep], stop[keep]
    length = stop - start
    if args.cap:
        length = np.minimum(length, args.cap)
    order = np.argsort(-length, kind="stable")
    job, start, length = job[order], start[order], length[order]
    P = int(job.size)

    # ---- 2. the same sample ranges as independent streams of the sequential kernels ----
    os.environ["NFCGPU_WINDOWED"] = "0"
    words = 96 << 20
    sink = torch.zeros(words, dtype=torch.int32, device=dev)
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    seq = nfclab_amd.NfcGpu(device=0, max_streams=P, frame_sink_bytes=1 << 20)
    seq.sink_attach(sink.data_ptr(), words, ctl.data_ptr())
    seq.sink_hold(True)
    base = seq.open(nfclab_amd.default_params(), count=P)
    ptrs = (data.data_ptr() + L * 8 + job * pitch + start * 8).tolist()
    ids = list(range(base, base + P))
    cnts = length.tolist()

    times = []
    for _ in range(2):
        seq.stats_reset()
        t0s = time.perf_counter()
        seq.submit_batch(ids, ptrs, cnts, FS, stride=2, location=nfclab_amd.LOC_DEVICE)
        seq.sync()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0s) * 1e3)
    st = seq.stats()
    seq.close()

    total = int(length.sum())
    out = {
        "workload": "the pieces of pass 0 of one headline submission (%d dense streams x %d samples, second slice)" % (S, L),
        "pieces": P,
        "samples_in_pieces": total,
        "tiles_in_pieces": int(((length + 63) // 64).sum()),
        "piece_length": {"median": int(np.median(length)), "p90": int(np.percentile(length, 90)), "p99": int(np.percentile(length, 99)), "longest": int(length.max())},
        "cap": args.cap,
        "wave_per_piece": {"kernel": "nfc_wave_kernel", "pass0": wave, "whole_submission_ms": round((t1 - t0) * 1e3, 1)},
        "lane_per_piece": {"kernel": "nfc_demod_fixed_kernel (64 pieces per wavefront, longest first)", "ms_first": round(times[0], 1), "ms_second": round(times[1], 1),
                           "kernel_ms_second": round(float(st.kernel_ms), 1), "launches": int(st.launches),
                           "tiles_per_us_second": round(((length + 63) // 64).sum() / (times[1] * 1e3), 1),
                           "lane_samples_per_s_second": round(total / (times[1] * 1e-3) / 1e9, 2)},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
