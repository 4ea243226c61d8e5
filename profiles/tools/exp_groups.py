#!/usr/bin/env python3
"""Experiment (round 5): config 5 dense decoded as G groups of streams, each group a context of its own driven by a host
thread of its own. Do the latency-bound phases of one group (repair rounds, late passes) fill with the bulk of another?
Usage: exp_groups.py [--streams 4096] [--samples 1048576] [--steps 4] [--groups 1,2,4] [--stagger-ms 0]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--groups", default="1,2,4")
    ap.add_argument("--stagger-ms", type=float, default=0.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__
    __graft_entry__.build()
    import nfclab_amd
    import synth

    dev = torch.device("cuda", 0)
    S, L, K, W = args.streams, args.samples, args.steps, args.warmup
    NS = 2
    T = NS * L
    template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
    template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)
    data = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
    synth.fill_iq_torch(data, template_dev, first_stream=0, chunk_streams=max(1, min(1024, (1 << 27) // T)))
    torch.cuda.synchronize()
    pitch = T * 8
    out = {}

    for G in [int(g) for g in args.groups.split(",")]:
        per = S // G
        ctxs = []
        for g in range(G):
            words = max(16 << 20, (K + W) * (1024 * per + 65536))
            sink = torch.zeros(words, dtype=torch.int32, device=dev)
            ctl = torch.zeros(4, dtype=torch.int32, device=dev)
            gpu = nfclab_amd.NfcGpu(device=0, max_streams=per, frame_sink_bytes=1 << 20)
            gpu.sink_attach(sink.data_ptr(), words, ctl.data_ptr())
            gpu.sink_hold(True)
            first = gpu.open(nfclab_amd.default_params(), count=per)
            ctxs.append((gpu, first, sink, ctl))
        torch.cuda.synchronize()

        def run(g, k0, k1):
            gpu, first, _, _ = ctxs[g]
            if args.stagger_ms and g:
                time.sleep(args.stagger_ms * g / 1e3)
            for k in range(k0, k1):
                gpu.submit_uniform(first, per, data.data_ptr() + g * per * pitch + (k % NS) * L * 8, pitch, L, 10000000, stride=2)
            gpu.sync()

        def all_groups(k0, k1):
            ts = [threading.Thread(target=run, args=(g, k0, k1)) for g in range(G)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            torch.cuda.synchronize()

        all_groups(0, W)
        t0 = time.perf_counter()
        all_groups(W, W + K)
        t1 = time.perf_counter()
        frames = sum(int(c[3][0].item()) for c in ctxs)
        ms = (t1 - t0) / K * 1e3
        out[G] = {"ms_per_step": round(ms, 2), "MS/s": round(S * L / ms / 1e3, 1), "sink_words": frames}
        print("groups", G, out[G], flush=True)
        for gpu, _, _, _ in ctxs:
            gpu.close()
        del ctxs
        torch.cuda.empty_cache()

    print(json.dumps(out))


if __name__ == "__main__":
    main()
