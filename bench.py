#!/usr/bin/env python3
"""bench.py — IQ Msamples/s demodulated on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one pass of the demodulation hot path (IQ -> magnitude -> front end -> NFC-A/B/F/V detector bank ->
symbol/bit/frame assembly) over one 8192-sample buffer (default) of every stream of this rank, with the IQ already
resident in HBM. Streams are independent capture streams (BASELINE config 5 shape, sharded by rank: weak
scaling, no data-path collective); decoded frames are gathered at the end of the timed region (RCCL all_gather
when N > 1, D2H when N == 1).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      HBM roofline of the demodulation kernel: 8 algorithmic bytes per IQ sample / HIP-event kernel time
  cpu_baseline  the reference's own lab::NfcDecoder (oracle/_ref/libnfcref.so) timed on this box's host cores on a
                bounded sample of the same streams; the same leg checks GPU frames against the reference bit for bit
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))

FS = 10000000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("NFC_BENCH_STREAMS", "131072")), help="streams per GPU")
    ap.add_argument("--samples", type=int, default=8192, help="samples per stream per step")
    ap.add_argument("--cpu-streams", type=int, default=24576, help="streams of rank 0 replayed on the host CPU")
    ap.add_argument("--check-streams", type=int, default=64, help="streams compared frame-by-frame with the reference")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    if args.gpus != world:
        # one process per GPU: N > 1 is launched by torch.distributed.run (see the module docstring); a bare
        # `python bench.py --gpus N` is a single-rank run and reports itself as such (n_gpus = 1)
        print("bench.py: --gpus %d but WORLD_SIZE is %d: running %d rank(s); launch with python -m torch.distributed.run "
              "--nproc-per-node %d bench.py --gpus %d for the multi-GPU figure" % (args.gpus, world, world, args.gpus, args.gpus),
              file=sys.stderr)

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    import nfclab_amd
    import synth
    import frames as framelib

    S, L, K, W = args.streams, args.samples, args.steps, args.warmup
    T = (K + W) * L

    template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
    template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)

    data = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
    synth.fill_iq_torch(data, template_dev, first_stream=rank * S)
    if os.environ.get("NFC_BENCH_IDLE") == "1":
        # diagnostic only (not a benchmark configuration): unmodulated carrier with 2-LSB noise, every lane stays in search mode
        noise = (torch.arange(T, device=dev) * 2654435761 % 5).to(torch.float32) / 32768.0
        data[:, :, 0] = 0.25 + noise[None, :]
        data[:, :, 1] = 0.0

    sink_words = 64 << 20
    sink = torch.zeros(sink_words, dtype=torch.int32, device=dev)
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    gpu = nfclab_amd.NfcGpu(device=local, max_streams=S, frame_sink_bytes=1 << 20)
    gpu.sink_attach(sink.data_ptr(), sink_words, ctl.data_ptr())
    gpu.sink_hold(True)
    gpu.profile(True)
    params = nfclab_amd.default_params(tech_mask=int(os.environ.get("NFC_BENCH_TECH_MASK", "15")))  # diagnostic switch, default all four
    first = gpu.open(params, count=S)

    pitch = T * 8

    def step(k):
        gpu.submit_uniform(first, S, data.data_ptr() + k * L * 8, pitch, L, FS, stride=2)

    def fence():
        gpu.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(W):
        step(k)

    fence()
    gpu.stats_reset()
    t0 = time.perf_counter()

    for k in range(W, W + K):
        step(k)
    gpu.sync()

    # frame gather: every rank's packed records to every rank (RCCL over xGMI), or to the host when N == 1
    host_used = int(ctl[0].item())
    if world > 1:
        gathered, counts = framelib.gather_sinks(sink, host_used, world)
        host_words = gathered[rank, :host_used].cpu().numpy()
        total_words = sum(counts)
    else:
        host_words = sink[:host_used].cpu().numpy()
        total_words = host_used

    fence()
    t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    st = gpu.stats()
    dropped = int(ctl[1].item())

    samples_per_step = S * L * world
    value = samples_per_step * K / elapsed / 1e6

    kernel_ms = st.kernel_ms / max(st.launches, 1)
    bytes_per_launch = 8.0 * S * L
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            if tj.get("streams") == S and tj.get("samples") == L:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "IQ Msamples/s demodulated",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 5 shape on one node: %d independent 10 MS/s IQ streams per GPU (fixture-derived synthetic "
                        "float2 IQ resident in HBM), all four tech decoders (NFC-A/B/F/V) enabled, %d-sample buffers per step; "
                        "configs[1] (single stream) cannot fill a GPU with a per-stream sequential state machine" % (S, L),
            "streams_per_gpu": S,
            "samples_per_stream_per_step": L,
            "sample_rate": FS,
            "frames_decoded_rank0": int(st.frames) if not st.frames == 0 else None,
            "parallelism": "stream-parallel x%d (one lane per stream, one process per GPU, RCCL frame all_gather)" % world,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 3),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic,
            "kernel": "nfc_demod_fixed_kernel" if FS == 10000000 else "nfc_demod_kernel",
            "kernel_ms_avg": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "traffic_frac_of_peak": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and kernel_ms > 0 else None,
        },
        "frames_dropped": dropped,
    }

    frames = framelib.parse_sink(host_words, host_used, FS) if rank == 0 else {}
    if rank == 0:
        result["config"]["frames_decoded_rank0"] = sum(len(v) for v in frames.values())
        result["config"]["frame_words_gathered"] = int(total_words)

    # ---- CPU baseline + parity check against the real reference (rank 0, N == 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import nfc_testlib as TL
        lib = TL.reference_lib()
        if lib is not None:
            C = min(S, args.cpu_streams)
            mags = torch.sqrt(data[:C, :, 0] * data[:C, :, 0] + data[:C, :, 1] * data[:C, :, 1]).cpu().numpy()
            mags = np.ascontiguousarray(mags, dtype=np.float32)
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)

            lib.nfcref_decode_many.restype = ctypes.c_long
            lib.nfcref_decode_many.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32,
                                               ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]

            def timed(streams, threads):
                secs = ctypes.c_double(0)
                lib.nfcref_decode_many(mags.ctypes.data, T, streams, T, FS, L, threads, ctypes.byref(secs))
                return streams * T / secs.value / 1e6, secs.value

            n_single = max(1, min(C, int(150e6 // T)))
            single, single_seconds = timed(n_single, 1)
            multi, multi_seconds = timed(C, cores)

            checked = min(C, args.check_streams)
            outs = []
            for s in range(checked):
                fr, _ = TL.reference_decode(mags[s], sample_rate=FS, chunk=L, keep_carrier=True, cap=8192, defined_storage=True)
                outs.append((s, fr, 0.0))

            bad = 0
            for s, fr, _ in outs:
                if frames.get(first + s, []) != fr:
                    bad += 1

            result["cpu_baseline"] = {
                "value": round(multi, 3),
                "unit": "Msamples/s",
                "cores": cores,
                "kind": "reference",
                "sample": "reference lab::NfcDecoder (oracle/_ref, built from /root/reference) on the magnitudes of the first %d "
                          "streams x %d samples (%.1f s), %d-sample buffers, one decoder per stream, %d threads; single thread on %d "
                          "streams: %.1f Msamples/s (%.1f s)" % (C, T, multi_seconds, L, cores, n_single, single, single_seconds),
                "single_thread_value": round(single, 3),
            }
            # BASELINE configs[0]: the reference's RadioDecoderTask itself (subjects + executor + reference decoder, one stream)
            task = os.path.join(ROOT, "oracle", "_ref", "task-ref")
            if os.path.exists(task):
                import subprocess
                import tempfile
                try:
                    with tempfile.TemporaryDirectory() as tmp:
                        wav = os.path.join(tmp, "plumbing.wav")
                        one = np.clip(np.rint(mags[0] * 32768.0), -32768, 32767).astype(np.int16)
                        TL.write_wav(wav, np.tile(one, max(1, int(40e6 // one.size))))
                        out = subprocess.run([task, wav], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300).stdout
                        done = [l.split() for l in out.splitlines() if l.startswith("DONE")]
                        if done:
                            n_task = one.size * max(1, int(40e6 // one.size))
                            result["cpu_baseline"]["radio_decoder_task_value"] = round(n_task / float(done[0][3]) / 1e6, 3)
                            result["cpu_baseline"]["sample"] += "; reference RadioDecoderTask (WAV -> radio.signal.raw -> task -> radio.decoder.frame," \
                                                               " one stream, %d samples): %.1f Msamples/s" % (n_task, n_task / float(done[0][3]) / 1e6)
                except Exception as exc:  # the plumbing figure is informative only
                    result["cpu_baseline"]["radio_decoder_task_value"] = None
                    result["cpu_baseline"]["sample"] += "; RadioDecoderTask run failed: %r" % (exc,)

            result["parity"] = {"streams_checked": checked, "streams_mismatching": bad,
                                "reference_frames": sum(len(o[1]) for o in outs[:checked])}
        else:
            result["cpu_baseline"] = None
            result["parity"] = "oracle/_ref not available on this box"

    if rank == 0:
        print(json.dumps(result))

    gpu.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
