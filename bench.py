#!/usr/bin/env python3
"""bench.py — IQ Msamples/s demodulated on MI355X (BASELINE.json metric), one process per GPU.

Headline = BASELINE config 5 as SURVEY 8(d) writes it: 4096 independent 10 MS/s streams of the fixture-derived set S1
(dense traffic: the bundled captures tiled end to end, float2 IQ resident in HBM), all four decoders enabled, 2^20 samples
per stream and step. A "step" is one submission of all streams of this rank through the C ABI (nfcgpu_submit_uniform):
scan kernel -> front-end planes -> windows -> wave decoder passes -> chain -> finish, frames into the
device sink. Streams are independent: with --gpus N the same 4096-stream dataset is cut over the ranks (strong scaling,
rank r decodes streams [r*4096/N, (r+1)*4096/N), no data-path collective); the decoded frames are gathered at the end of
the timed region through the C ABI (ncclAllGather over RCCL when N > 1, D2H when N == 1).

At N == 1 the same line also carries, as `points` (outside the timed region, each with its own context and clock):
BASELINE configs 2-4 (every bundled capture as one stream), config 5 with sparse and with no traffic, one long stream, the
share of config 5 one of eight GPUs holds, set S2 (off the capture grid: rotated IQ + noise), the same share fed from host
memory (H2D included), and the round-1/2 headline (131072 streams x 8192-sample buffers: the sequential kernel's
saturating point).

Extra objects on the JSON line:
  roofline         HBM roofline of the dominant kernel of the headline (nfc_wave_kernel, all its launches of a step)
  roofline_search  the same for the scan kernel (the per-sample search kernel of the time-parallel path), measured on the
                   config-5 point without traffic; `on_sparse_traffic`: on the sparse point
  cpu_baseline     the reference's own lab::NfcDecoder (oracle/_ref/libnfcref.so) on this box's host cores on a bounded
                   sample of the headline streams; the same leg checks GPU frames against the reference bit for bit
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))

FS = 10000000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def git_head():
    """the commit of this tree: from git where there is a repository, otherwise the stamp __graft_entry__.build() left in
    nfc-laboratory_amd/build/ when it last ran in one (the GPU box gets a snapshot without .git)"""
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                              text=True, timeout=10).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "nfc-laboratory_amd", "build", "git_head.txt")) as f:
            return f.read().strip() or None
    except Exception:
        return None


# files under csrc/ that are not part of the kernels the stored counter figures are about (nfc_wave_kernel, nfc_scan_kernel,
# nfc_demod_fixed_kernel): the host runtime, and the kernel of the envelope tracker's second walks (round 4, a translation unit
# of its own that includes the others' headers and changes none of them)
NOT_IN_DIGEST = ("nfcgpu.hip", "nfc_envelope.hip", "nfc_envelope.hpp")


def sources_digest():
    """sha1 over the sources of the kernels the stored counter figures belong to: everything under csrc/ but NOT_IN_DIGEST"""
    import hashlib
    src = os.path.join(ROOT, "nfc-laboratory_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(src)):
        if not f.endswith((".h", ".hpp", ".hip")) or f in NOT_IN_DIGEST:
            continue
        with open(os.path.join(src, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def stored_traffic(kernel, streams, samples):
    """HBM bytes per launch from the PMC passes kept under profiles/ (profiles/tools/r02/round_profile.sh), or None when
    there is no record for this kernel and shape, or when the sources have changed since the record was taken."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f).get(kernel)
        if not rec or rec.get("streams") != streams or rec.get("samples") != samples:
            return None, None
        if rec.get("sources_sha1") != sources_digest():
            return None, "profiles/traffic.json was taken from other kernel sources (%s)" % rec.get("git")
        return rec.get("hbm_bytes_per_launch"), "profiles/traffic.json @ %s (%s)" % (rec.get("git"), rec.get("from"))
    except Exception:
        return None, None


def host_cpu():
    """(model name, physical cores, logical CPUs) of this box, from /proc/cpuinfo"""
    model, physical, logical = None, set(), 0
    try:
        pkg = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                k, _, v = line.partition(":")
                k, v = k.strip(), v.strip()
                if k == "model name" and model is None:
                    model = v
                elif k == "processor":
                    logical += 1
                elif k == "physical id":
                    pkg = v
                elif k == "core id":
                    core = v
                elif not k and pkg is not None and core is not None:
                    physical.add((pkg, core))
                    pkg = core = None
    except Exception:
        pass
    return model, (len(physical) or None), (logical or None)


def host_limits():
    """what the box lets this process have: the processors it may run on and the processor quota of its control group (a
    container with a quota of N processors runs 128 threads at N processors' worth, whatever the harness does)"""
    out = {"sched_affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    for name, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("cgroup_v1_cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
                       ("cgroup_v1_cfs_period_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"), ("cgroup_cpuset", "/sys/fs/cgroup/cpuset.cpus.effective"),
                       ("loadavg", "/proc/loadavg")):
        try:
            with open(path) as f:
                out[name] = f.read().strip()
        except Exception:
            out[name] = None
    return out


def processor_quota(limits):
    """processors' worth of time the control group gives this process (cgroup v2 cpu.max 'quota period', or the v1 pair); None: no quota"""
    try:
        v2 = (limits.get("cgroup_cpu_max") or "").split()
        if len(v2) == 2 and v2[0] != "max":
            return float(v2[0]) / float(v2[1])
        q, per = limits.get("cgroup_v1_cfs_quota_us"), limits.get("cgroup_v1_cfs_period_us")
        if q and per and int(q) > 0:
            return float(q) / float(per)
    except Exception:
        pass
    return None


def clamp_used(cursor, dropped, capacity):
    """valid words of a held sink: the cursor keeps counting past the capacity on overflow (nfc_emit)"""
    limit = capacity - (9 + 128) + 1
    return min(cursor, limit) if dropped else min(cursor, capacity)


def headline_layout(steps, warmup, samples, slices):
    """(slices resident, samples per stream resident) of the headline dataset: SURVEY 8(d) fixes it at L samples per stream
    (32 GiB for 4096 streams x 2^20) so that the identical data serves every run; `slices` of them are kept (default 2) and
    step k submits slice k % slices, whatever --steps and --warmup are"""
    n = max(1, min(steps + warmup, slices))
    return n, n * samples


def headline_sink_words(streams, samples, steps, warmup):
    """capacity of the held frame sink: dense S1 traffic leaves about 500 words per stream and 2^20 samples (2.04 M words per
    step of config 5); twice that for every step and warm-up step, at least 16 Mi words"""
    per_step = 1024 * streams * max(1, samples >> 20) + 65536
    return max(16 << 20, (steps + warmup) * per_step)


def headline_device_bytes(streams, samples, steps, warmup, slices=2, world=1):
    """device memory one rank of the headline asks for: resident IQ, held sink (+ gather buffer when N > 1) and the library's
    work buffers of one submission (front-end planes 16 B per sample, tiles / points / lanes < 1 B per sample)"""
    _, resident = headline_layout(steps, warmup, samples, slices)
    iq = 8 * streams * resident
    sink = 4 * headline_sink_words(streams, samples, steps, warmup) * (1 + (world if world > 1 else 0))
    work = 17 * streams * samples
    return iq + sink + work


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=2,
                    help="slices of --samples per stream kept resident (SURVEY 8(d): the dataset is L = 2^20 per stream, 32 GiB for 4096 "
                         "streams); step k submits slice k %% slices, the decoder's carried state goes on from step to step")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("NFC_BENCH_STREAMS", "4096")),
                    help="streams of the headline dataset (BASELINE config 5: 4096); with --scaling strong cut over the ranks, with weak per GPU")
    ap.add_argument("--samples", type=int, default=1 << 20, help="samples per stream per step (headline)")
    ap.add_argument("--cpu-streams", type=int, default=256, help="streams of rank 0 replayed on the host CPU")
    ap.add_argument("--check-streams", type=int, default=64, help="streams compared frame-by-frame with the reference")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (BASELINE config 5: one 4096-stream dataset, rank r decodes streams [r*S/N, (r+1)*S/N)); weak: --streams per GPU")
    ap.add_argument("--require-abi-gather", action="store_true", help="(accepted for the command lines of rounds 3-4: it is the default again)")
    ap.add_argument("--allow-torch-gather", action="store_true",
                    help="N > 1: should the C ABI's RCCL communicator not come up on every rank, measure the line with torch.distributed's "
                         "all_gather instead (and say so in config.parallelism) rather than fail. Default: the product's gather or an error")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-points", action="store_true", help="headline only")
    ap.add_argument("--points-budget", type=float, default=780.0, help="no new point is started later than this many seconds after the start")
    ap.add_argument("--points", default="fixtures_single,config5_sparse,config5_idle,share_sparse,share_dense,share_dense_h2d,single_sparse,single_dense,s2_share,s2_config5,saturating")
    ap.add_argument("--saturating-streams", type=int, default=131072, help="streams of the `saturating` point (8192-sample buffers, sequential kernel)")
    ap.add_argument("--share-streams", type=int, default=512, help="streams one GPU holds when BASELINE's 4096 are spread over 8")
    ap.add_argument("--single-dense-samples", type=int, default=1 << 23)
    ap.add_argument("--config5-streams", type=int, default=4096)
    ap.add_argument("--config5-samples", type=int, default=1 << 20)
    ap.add_argument("--single-samples", type=int, default=1 << 26)
    args = ap.parse_args()
    t_start = time.perf_counter()

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

    # NFC_BENCH_DRY_CPU=1: a rehearsal of this script's control flow on a box without GPUs (tests/test_bench_multi_rank_dry.py): tensors
    # on the host, gloo between the ranks, the library under NFCGPU_LIB the emulated test build with its stand-in for RCCL
    # (tests/hostsim). Not a measurement: the line it prints says so.
    dry = os.environ.get("NFC_BENCH_DRY_CPU") == "1"

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="gloo" if dry else "nccl", rank=rank, world_size=world)

    if dry:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.empty_cache = lambda *a, **k: None
    else:
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
    if args.scaling == "strong":
        S = max(1, args.streams // world)  # the same dataset cut over the ranks
    NS, T = headline_layout(K, W, L, args.slices)  # slices resident, samples per stream resident
    order = [k % NS for k in range(W + K)]           # the slice step k submits

    template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
    template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)

    data = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
    synth.fill_iq_torch(data, template_dev, first_stream=rank * S, chunk_streams=max(1, min(1024, (1 << 27) // T)))
    if os.environ.get("NFC_BENCH_IDLE") == "1":
        # diagnostic only (not a benchmark configuration): unmodulated carrier with 2-LSB noise, every lane stays in search mode
        noise = (torch.arange(T, device=dev) * 2654435761 % 5).to(torch.float32) / 32768.0
        data[:, :, 0] = 0.25 + noise[None, :]
        data[:, :, 1] = 0.0

    sink_words = headline_sink_words(S, L, K, W)
    sink = torch.zeros(sink_words, dtype=torch.int32, device=dev)
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    gpu = nfclab_amd.NfcGpu(device=local, max_streams=S, frame_sink_bytes=1 << 20)
    gpu.sink_attach(sink.data_ptr(), sink_words, ctl.data_ptr())
    gpu.sink_hold(True)
    gpu.profile(True)
    params = nfclab_amd.default_params(tech_mask=int(os.environ.get("NFC_BENCH_TECH_MASK", "15")))  # diagnostic switch, default all four
    first = gpu.open(params, count=S)

    # the frame gather lives behind the C ABI (ncclAllGather in C++): the unique id travels over torch.distributed
    gathered = None
    abi_gather = world > 1
    if world > 1:
        ok = 1
        try:
            ident = [gpu.comm_unique_id() if rank == 0 else None]
        except Exception:
            ident, ok = [None], 0
        dist.broadcast_object_list(ident, src=0)
        try:
            if ident[0] is None:
                raise RuntimeError("no RCCL unique id")
            gpu.comm_init(ident[0], rank, world)
        except Exception as exc:
            ok = 0
            sys.stderr.write("rank %d: nfcgpu_comm_init failed (%r)\n" % (rank, exc))
        # every rank takes the same way: the C ABI's gather, or (should RCCL not come up behind it on some rank) its torch twin
        agreed = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        abi_gather = bool(int(agreed.item()))
        if not abi_gather and not args.allow_torch_gather:
            raise SystemExit("bench.py: the frame gather behind the C ABI (nfcgpu_comm_init over RCCL) did not come up on every rank "
                             "(--allow-torch-gather measures the line with torch.distributed's all_gather instead)")
        if not abi_gather and rank == 0:
            sys.stderr.write("bench.py: nfcgpu_comm_init did not come up on every rank: the frames are gathered with torch.distributed's "
                             "all_gather (RCCL as well; the data path has no collective either way)\n")
        if abi_gather:
            gathered = torch.zeros(sink_words * world, dtype=torch.int32, device=dev)  # room for every rank's records, packed

    pitch = T * 8

    def step(k):
        gpu.submit_uniform(first, S, data.data_ptr() + order[k] * L * 8, pitch, L, FS, stride=2)

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

    # frame gather: every rank's packed records to every rank (RCCL over xGMI, C ABI), or to the host when N == 1
    dropped = int(ctl[1].item())
    host_used = clamp_used(int(ctl[0].item()), dropped, sink_words)
    if world > 1 and abi_gather:
        counts, stride = gpu.gather_frames(gathered.data_ptr(), gathered.numel())
        mine_at = rank * stride if stride else sum(counts[:rank])  # (packed at exact sizes: stride 0)
        host_words = gathered[mine_at:mine_at + host_used].cpu().numpy()
        total_words = sum(counts)
    elif world > 1:
        everyone, counts = framelib.gather_sinks(sink, host_used, world)
        host_words = everyone[rank, :host_used].cpu().numpy()
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

    samples_per_step = S * L * world
    value = samples_per_step * K / elapsed / 1e6

    # The dominant kernel of the headline: the wave decoder. A step launches it several times (carry lanes, the windows
    # of every decode pass, final lanes); `kernel_ms` is the HIP-event time of all its launches of one step, on the
    # streams they were launched on (nfcgpu_stats::wave_ms; rocprofv3's per-kernel total of the same command agrees:
    # profiles/r03). Submissions the time-parallel path does not take run the sequential kernel instead.
    wave_ms = st.wave_ms / K
    wave_busy_ms = st.wave_busy_ms / K if st.wave_busy_ms > 0 else wave_ms
    seq_ms = st.kernel_ms / K
    dominant = "nfc_wave_kernel" if wave_ms >= seq_ms else "nfc_demod_fixed_kernel"
    kernel_ms = max(wave_ms, seq_ms)
    bytes_per_launch = 8.0 * S * L
    achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # the carry lanes of a pass run on a second HIP stream beside the speculative lanes: the sum of the launch durations counts
    # the overlap twice; the time the kernel was running at all is the union of the launch intervals (nfcgpu_stats::wave_busy_ms)
    busy_ms = wave_busy_ms if dominant == "nfc_wave_kernel" else kernel_ms
    achieved_busy = bytes_per_launch / (busy_ms * 1e-3) / 1e9 if busy_ms > 0 else 0.0

    # the measured denominator: streaming read of this very buffer with 16-byte loads
    read_peak = 0.0 if dry else gpu.read_bandwidth(data.data_ptr(), min(data.numel() * 4, 32 << 30), repeats=5)

    traffic, traffic_source = stored_traffic(dominant, S, L)
    traffic_step, traffic_step_source = stored_traffic("step", S, L)  # every kernel of a step, not only the dominant one

    result = {
        "metric": "IQ Msamples/s demodulated",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(elapsed / K * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not dry else "synthetic - DRY RUN on the emulated runtime (NFC_BENCH_DRY_CPU=1): a rehearsal of the script, not a measurement",
        "config": {
            "workload": "BASELINE config 5: %d independent 10 MS/s IQ streams (set S1: fixture-derived synthetic float2 IQ resident in HBM, "
                        "dense traffic - the bundled captures tiled end to end), %d per GPU, all four tech decoders (NFC-A/B/F/V) enabled, "
                        "%d samples per stream and step, one submission per step through the C ABI (time-parallel path: scan -> planes -> "
                        "windows -> wave decoder passes -> chain -> finish)" % (S * world if args.scaling == "strong" else S, S, L),
            "streams_total": S * world,
            "streams_per_gpu": S,
            "samples_per_stream_per_step": L,
            "sample_rate": FS,
            "frames_decoded_rank0": None,
            "parallelism": "stream-parallel x%d (one process per GPU, %s scaling, frames gathered with %s)" % (
                world, args.scaling, "ncclAllGather behind the C ABI" if abi_gather or world == 1 else "torch.distributed all_gather (the C ABI's RCCL communicator did not come up)"),
            "time_parallel": {"streams": int(st.windowed_streams), "streams_sequential": int(st.fallback_streams), "lanes": int(st.windows),
                              "decode_passes": int(st.window_passes), "chunks_rescanned": int(st.scan_repairs), "scan_kernel_ms_per_step": round(st.scan_ms / K, 3),
                              "planes_kernel_ms_per_step": round(st.planes_ms / K, 3), "windowed_decode_ms_per_step": round(st.window_ms / K, 3),
                              "wave_kernel_ms_per_step": round(wave_ms, 3), "wave_kernel_busy_ms_per_step": round(wave_busy_ms, 3), "wave_kernel_launches_per_step": round(st.wave_launches / K, 1),
                              "sequential_kernel_ms_per_step": round(seq_ms, 3)},
            "git": git_head(),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 3),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6),
            "traffic": traffic,
            "traffic_source": traffic_source,
            # HBM bytes of a WHOLE step by the same counters: scan, second walks, planes, tile tests, windows, every pass of the wave
            # decoder, chain and finish summed (profiles/tools/r06/make_traffic.py); `traffic` above is the dominant kernel's alone
            "traffic_step": traffic_step,
            "traffic_step_source": traffic_step_source,
            "traffic_step_over_algorithmic": round(traffic_step / bytes_per_launch, 3) if traffic_step else None,
            "kernel": dominant,
            "kernel_ms_avg": round(kernel_ms, 4),
            "kernel_ms_is": "sum over all launches of the kernel in one step (HIP events on the launching streams; launches on the "
                            "library's two streams may overlap, so this is a sum of launch durations, not exclusive time)",
            "kernel_launches_per_step": round((st.wave_launches if dominant == "nfc_wave_kernel" else st.launches) / K, 1),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "peak_measured_streaming_read": round(read_peak, 1),
            "frac_of_measured_peak": round(achieved / read_peak, 6) if read_peak > 0 else None,
            "kernel_busy_ms": round(busy_ms, 4),
            "kernel_busy_ms_is": "time per step during which the kernel was running at all: union of its launch intervals (HIP events against one epoch event)",
            "achieved_over_busy_time": round(achieved_busy, 3),
            "frac_over_busy_time": round(achieved_busy / HBM_PEAK_GBS, 6),
        },
        "frames_dropped": dropped,
    }

    frames = framelib.parse_sink(host_words, host_used, FS) if rank == 0 else {}
    if rank == 0:
        result["config"]["frames_decoded_rank0"] = sum(len(v) for v in frames.values())
        result["config"]["frame_words_gathered"] = int(total_words)

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nfc_testlib as TL
    lib = TL.reference_lib() if rank == 0 else None

    # ---- CPU baseline + parity check against the real reference (rank 0, N == 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu:
        if lib is not None:
            C = min(S, args.cpu_streams)
            # what the decoder saw: slice order[k] of the resident data in step k. The CPU legs are bounded (about
            # 10-30 s of CPU work): the timing leg takes the first min(K + W, 3) steps of C streams, the parity leg the whole
            # submission sequence of --check-streams streams
            base = torch.sqrt(data[:C, :, 0] * data[:C, :, 0] + data[:C, :, 1] * data[:C, :, 1]).cpu().numpy()
            base = np.ascontiguousarray(base, dtype=np.float32)

            def sequence(rows, steps_taken):
                return np.ascontiguousarray(np.concatenate([base[rows, order[k] * L:(order[k] + 1) * L] for k in range(steps_taken)], axis=1))

            timed_steps = min(K + W, 3)
            mags = sequence(slice(0, C), timed_steps)
            TT = timed_steps * L
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cpu_model, cpu_physical, cpu_logical = host_cpu()
            chunk = 65536 if TT >= 65536 else TT  # the reference harness's buffer length (TS/main.cpp:163)

            # (oracle/ref_capi.cpp: threads started, decoders constructed and every thread's samples first touched by that thread
            # before the clock; the clock is the decode alone. detail: slowest / fastest thread, sum over threads, set-up seconds)
            lib.nfcref_decode_many_detail.restype = ctypes.c_long
            lib.nfcref_decode_many_detail.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32,
                                                      ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
            spread = {}

            def timed(streams, threads):
                secs = ctypes.c_double(0)
                detail = (ctypes.c_double * 5)()
                lib.nfcref_decode_many_detail(mags.ctypes.data, TT, streams, TT, FS, chunk, threads, ctypes.byref(secs), detail)
                spread[threads] = {"wall_s": round(secs.value, 4), "slowest_thread_s": round(detail[0], 4), "fastest_thread_s": round(detail[1], 4),
                                   "threads_busy_fraction": round(detail[2] / (threads * secs.value), 3) if secs.value > 0 else None,
                                   "setup_s_outside_the_clock": round(detail[3], 3),
                                   # processor seconds the threads got inside the clock / wall seconds: processors really at work
                                   "processors_at_work": round(detail[4] / secs.value, 2) if secs.value > 0 else None}
                return streams * TT / secs.value / 1e6, secs.value

            n_single = max(1, min(C, int(150e6 // TT)))
            single, single_seconds = timed(n_single, 1)
            multi, multi_seconds = timed(C, cores)
            physical = min(cores, cpu_physical or cores)
            if physical < cores:
                multi_phys, multi_phys_seconds = timed(C, physical)
            else:
                multi_phys, multi_phys_seconds = multi, multi_seconds
            del mags

            checked = min(C, args.check_streams)
            from concurrent.futures import ThreadPoolExecutor

            def reference_frames_of(s):
                full = sequence(slice(s, s + 1), K + W)[0]
                fr, _ = TL.reference_decode(full, sample_rate=FS, chunk=L, keep_carrier=True, cap=4096 * (K + W), defined_storage=True)
                return (s, fr, 0.0)

            with ThreadPoolExecutor(max_workers=max(1, min(32, cores))) as pool:  # (as tests/parity_sweep_driver.py does)
                outs = list(pool.map(reference_frames_of, range(checked)))

            bad, bad_ids, first_diff = 0, [], None
            for s, fr, _ in outs:
                mine = frames.get(first + s, [])
                if mine != fr:
                    bad += 1
                    bad_ids.append(s)
                    if first_diff is None:
                        at = next((i for i, (a, b) in enumerate(zip(mine, fr)) if a != b), min(len(mine), len(fr)))
                        first_diff = {"stream": s, "frame_index": at, "gpu_frames": len(mine), "reference_frames": len(fr),
                                      "gpu": repr(mine[at]) if at < len(mine) else None, "reference": repr(fr[at]) if at < len(fr) else None}
                    dump = os.environ.get("NFC_BENCH_DUMP")
                    if dump:
                        os.makedirs(dump, exist_ok=True)
                        with open(os.path.join(dump, "stream_%d.json" % s), "w") as fh:
                            json.dump({"stream": s, "order": order, "samples": L, "gpu": [list(map(lambda v: v if not isinstance(v, bytes) else v.hex(), f)) for f in mine],
                                       "reference": [list(map(lambda v: v if not isinstance(v, bytes) else v.hex(), f)) for f in fr]}, fh)

            # the baseline is the better of the two multi-thread runs (SMT siblings share a core's load ports: on the 2 x 64-core
            # host of the GPU boxes one thread per physical core is the faster run); both are in the object
            best, best_threads = (multi_phys, physical) if multi_phys > multi else (multi, cores)
            limits = host_limits()
            quota = processor_quota(limits)
            result["cpu_baseline"] = {
                "value": round(best, 3),
                "unit": "Msamples/s",
                "cores": best_threads,
                "all_threads_value": round(multi, 3),
                "all_threads": cores,
                "cpu_model": cpu_model,
                "physical_cores": cpu_physical,
                "logical_cpus": cpu_logical,
                "kind": "reference",
                "sample": "reference lab::NfcDecoder (oracle/_ref, built from /root/reference) on the magnitudes of the first %d "
                          "streams x %d samples (%.1f s), %d-sample buffers, one decoder per stream, %d threads; single thread on %d "
                          "streams: %.1f Msamples/s (%.1f s); %d threads (one per physical core): %.1f Msamples/s (%.1f s)" % (
                              C, TT, multi_seconds, chunk, cores, n_single, single, single_seconds, physical, multi_phys, multi_phys_seconds),
                "single_thread_value": round(single, 3),
                "physical_cores_value": round(multi_phys, 3),
                "physical_cores_threads": physical,
                # what the cores would do if every one of them ran like the single thread, and how much of that the threaded run got
                "ideal_single_thread_x_physical_cores": round(single * physical, 3),
                "parallel_efficiency": round(multi_phys / (single * physical), 4) if single > 0 and physical else None,
                "parallel_efficiency_all_threads": round(multi / (single * cores), 4) if single > 0 and cores else None,
                "thread_spread": {str(k): v for k, v in spread.items()},
                "host_limits": limits,
                # the box's control group may give the process fewer processors than it shows (thread_spread.processors_at_work is
                # what the threads really got): the threaded run against the single thread times that quota
                "processor_quota": quota,
                "parallel_efficiency_vs_quota": round(multi_phys / (single * min(quota, physical)), 4) if single > 0 and physical and quota else None,
                "harness": "threads started, decoders constructed and each thread's samples first touched by that thread before the clock "
                           "(oracle/ref_capi.cpp: nfcref_decode_many_detail); the clock is the decode alone",
            }
            # BASELINE configs[0]: the reference's RadioDecoderTask itself (subjects + executor + reference decoder, one stream)
            task = os.path.join(ROOT, "oracle", "_ref", "task-ref")
            if os.path.exists(task):
                import tempfile
                try:
                    with tempfile.TemporaryDirectory() as tmp:
                        wav = os.path.join(tmp, "plumbing.wav")
                        one = np.clip(np.rint(base[0] * 32768.0), -32768, 32767).astype(np.int16)
                        TL.write_wav(wav, np.tile(one, max(1, int(40e6 // one.size))))
                        out = subprocess.run([task, wav], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=300).stdout
                        done = [l.split() for l in out.splitlines() if l.startswith("DONE")]
                        if done:
                            n_task = one.size * max(1, int(40e6 // one.size))
                            result["cpu_baseline"]["radio_decoder_task_value"] = round(n_task / float(done[0][3]) / 1e6, 3)
                            result["cpu_baseline"]["sample"] += "; reference RadioDecoderTask (WAV -> radio.signal.raw -> task -> radio.decoder.frame," \
                                                               " one stream, %d samples): %.1f Msamples/s" % (n_task, n_task / float(done[0][3]) / 1e6)
                        # the same task, same WAVs, on top of the GPU decoder (host/NfcDecoder.cpp in block mode): the shim path
                        # end to end, host buffers through the pinned staging of the C ABI
                        task_gpu = os.path.join(ROOT, "oracle", "_ref", "task-gpu")
                        if os.path.exists(task_gpu):
                            segs0 = synth.sparse_segments(template)
                            # (lengths that end on a partial buffer: the shim takes the short last buffer as the end of the stream)
                            sparse = synth.sparse_magnitude_f32(template, segs0, 0, 0, (1 << 25) + 12345)
                            wav_s = os.path.join(tmp, "sparse.wav")
                            TL.write_wav(wav_s, np.clip(np.rint(sparse * 32768.0), -32768, 32767).astype(np.int16))
                            dense_n = (1 << 23) + 12345
                            wav_d = os.path.join(tmp, "dense.wav")
                            dense = synth.magnitude_f32(template, 0, 0, dense_n)
                            TL.write_wav(wav_d, np.clip(np.rint(dense * 32768.0), -32768, 32767).astype(np.int16))
                            env = dict(os.environ, NFCGPU_SHIM_BLOCK=str(1 << 22))
                            env_auto = dict(os.environ, NFCGPU_SHIM_BLOCK="auto")
                            env_default = {k: v for k, v in os.environ.items() if k != "NFCGPU_SHIM_BLOCK"}
                            shim = {"block_samples": 1 << 22,
                                    "modes": "gpu_task: NFCGPU_SHIM_BLOCK=4194304; gpu_task_auto: NFCGPU_SHIM_BLOCK=auto (the shim grows its block while "
                                             "the decoder does not keep up with the samples' own duration); gpu_task_default_mode: no variable - every "
                                             "65536-sample buffer of the task submitted and collected inside its nextFrames() call, frames delivered "
                                             "by the very call the reference delivers them by"}
                            for label, path, n_w in (("sparse", wav_s, sparse.size), ("dense", wav_d, dense_n)):
                                row = {"samples": int(n_w)}
                                for who, exe, e in (("reference_task", task, os.environ), ("gpu_task", task_gpu, env), ("gpu_task_auto", task_gpu, env_auto),
                                                    ("gpu_task_default_mode", task_gpu, env_default)):
                                    o = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=600, env=e).stdout
                                    dn = [l.split() for l in o.splitlines() if l.startswith("DONE")]
                                    row[who + "_Msamples_per_s"] = round(n_w / float(dn[0][3]) / 1e6, 3) if dn else None
                                    row[who + "_frames"] = int(dn[0][2]) if dn and len(dn[0]) > 2 and dn[0][2].isdigit() else None
                                shim[label] = row
                            result["cpu_baseline"]["radio_decoder_task_gpu"] = shim
                except Exception as exc:  # the plumbing figure is informative only
                    result["cpu_baseline"].setdefault("radio_decoder_task_value", None)
                    result["cpu_baseline"]["sample"] += "; RadioDecoderTask run failed: %r" % (exc,)

            result["parity"] = {"streams_checked": checked, "streams_mismatching": bad, "submissions_compared": K + W,
                                "reference_frames": sum(len(o[1]) for o in outs[:checked]), "mismatching_streams": bad_ids,
                                "first_difference": first_diff}
            del base
        else:
            result["cpu_baseline"] = None
            result["parity"] = "oracle/_ref not available on this box"

    gpu.close()
    del data
    torch.cuda.empty_cache()

    # ---- BASELINE's own configurations (N == 1): each point on its own context and clock, outside the headline's timed region ----
    if rank == 0 and world == 1 and not args.no_points:
        segs = synth.sparse_segments(template)
        points = {}
        point_sink_words = 64 << 20  # (a point is at most 8 submissions)

        def run_point(name, n_streams, n_samples, sparse, steps, warm, check, idle=False, offgrid=False, from_host=False):
            total = (steps + warm) * n_samples
            buf = torch.empty((n_streams, total, 2), dtype=torch.float32, device=dev)
            if idle:
                # unmodulated carrier with a few LSB of per-stream, non-periodic noise on the int16 grid: nothing to decode
                tix = torch.arange(total, device=dev, dtype=torch.int64)
                for s0 in range(0, n_streams, 256):
                    s1 = min(n_streams, s0 + 256)
                    sid = torch.arange(s0, s1, device=dev, dtype=torch.int64)[:, None]
                    h = (tix[None, :] * 2654435761 + sid * 40503) & 0xFFFFFFFF
                    h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
                    h = (h ^ (h >> 13)) * 3266489917 & 0xFFFFFFFF
                    h = h ^ (h >> 16)
                    buf[s0:s1, :, 0] = 0.25 + ((h % 9) - 4).to(torch.float32) / 32768.0
                    buf[s0:s1, :, 1] = 0.0
                    del h
            elif sparse:
                synth.fill_sparse_iq_torch(buf, template_dev, segs, first_stream=0, chunk_streams=max(1, min(256, (1 << 27) // total)))
            else:
                synth.fill_iq_torch(buf, template_dev, first_stream=0, chunk_streams=max(1, min(1024, (1 << 27) // total)))
            if offgrid:
                # set S2: the S1 magnitudes on a random per-stream phase plus white noise of sigma 0.002 on both
                # components, fp32 IQ - what a radio delivers; off the capture grid (running sums no longer order-independent)
                gen = torch.Generator(device=dev)
                gen.manual_seed(20260926)
                for s0 in range(0, n_streams, 64):
                    s1 = min(n_streams, s0 + 64)
                    m = torch.sqrt(buf[s0:s1, :, 0] ** 2 + buf[s0:s1, :, 1] ** 2)
                    phi = torch.rand((s1 - s0, 1), device=dev, generator=gen) * 6.283185307179586
                    buf[s0:s1, :, 0] = m * torch.cos(phi) + torch.randn(m.shape, device=dev, generator=gen) * 0.002
                    buf[s0:s1, :, 1] = m * torch.sin(phi) + torch.randn(m.shape, device=dev, generator=gen) * 0.002
                    del m
            host = buf.cpu().numpy() if from_host else None  # (the shim's way in: host buffers through the pinned staging of the C ABI)
            psink = torch.zeros(point_sink_words, dtype=torch.int32, device=dev)
            pctl = torch.zeros(4, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()  # what the generators above left in torch's cache is memory the library cannot allocate
            g = nfclab_amd.NfcGpu(device=local, max_streams=max(64, n_streams), frame_sink_bytes=1 << 20)
            try:
                return run_point_on(g, name, n_streams, n_samples, sparse, steps, warm, check, idle, offgrid, from_host, buf, host, psink, pctl, total)
            finally:
                g.close()
                del buf, psink, host
                torch.cuda.empty_cache()

        def run_point_on(g, name, n_streams, n_samples, sparse, steps, warm, check, idle, offgrid, from_host, buf, host, psink, pctl, total):
            g.sink_attach(psink.data_ptr(), point_sink_words, pctl.data_ptr())
            g.sink_hold(True)
            g.profile(True)
            f0 = g.open(nfclab_amd.default_params(), count=n_streams)

            def submit(k):
                if from_host:
                    ids = list(range(f0, f0 + n_streams))
                    ptrs = [host.ctypes.data + (s * total + k * n_samples) * 8 for s in range(n_streams)]
                    g.submit_batch(ids, ptrs, [n_samples] * n_streams, FS, stride=2)
                else:
                    g.submit_uniform(f0, n_streams, buf.data_ptr() + k * n_samples * 8, total * 8, n_samples, FS, stride=2)

            for k in range(warm):
                submit(k)
            g.sync()
            torch.cuda.synchronize()
            g.stats_reset()
            ta = time.perf_counter()
            for k in range(warm, warm + steps):
                submit(k)
            g.sync()
            torch.cuda.synchronize()
            tb = time.perf_counter()
            ps = g.stats()
            pdrop = int(pctl[1].item())
            used = clamp_used(int(pctl[0].item()), pdrop, point_sink_words)
            pframes = framelib.parse_sink(psink[:used].cpu().numpy(), used, FS)
            point = {
                "workload": "%d stream(s) x %d samples per step, %s traffic (%s), %s, all four decoders" % (
                    n_streams, n_samples, "no" if idle else ("sparse" if sparse else "dense"),
                    "unmodulated carrier with +-4 LSB of noise" if idle else
                    ("S1q: one exchange of the captures per 2^19 samples in quiet carrier" if sparse else
                     ("S2: the S1 magnitudes on a random phase per stream + white noise sigma 0.002, fp32 IQ off the capture grid" if offgrid else
                      "S1: the captures tiled end to end")),
                    "IQ in host memory, staged and copied by the library inside the timed region (H2D included)" if from_host else "IQ resident in HBM"),
                "value": round(n_streams * n_samples * steps / (tb - ta) / 1e6, 3),
                "unit": "Msamples/s",
                "ms_per_step": round((tb - ta) / steps * 1e3, 3),
                "steps": steps,
                "warmup": warm,
                "real_time_factor_per_stream": round(n_samples * steps / (tb - ta) / FS, 3),
                "frames": sum(len(v) for v in pframes.values()),
                "frames_dropped": pdrop,
                "time_parallel": {"streams": int(ps.windowed_streams), "streams_sequential": int(ps.fallback_streams), "lanes": int(ps.windows),
                                  "decode_passes": int(ps.window_passes), "chunks_rescanned": int(ps.scan_repairs),
                                  "scan_kernel_ms": round(ps.scan_ms, 3), "planes_kernel_ms": round(ps.planes_ms, 3), "windowed_decode_ms": round(ps.window_ms, 3),
                                  "wave_kernel_ms": round(ps.wave_ms, 3), "wave_kernel_launches": int(ps.wave_launches),
                                  "sequential_kernel_ms": round(ps.kernel_ms, 3)},
            }
            if ps.scan_ms > 0:
                point["scan_kernel_GBps"] = round(8.0 * ps.scan_samples / (ps.scan_ms * 1e-3) / 1e9, 1)
            # HBM roofline of the point's dominant kernel: algorithmic bytes of a step over all the kernel's launches of a step
            kern = max((("nfc_wave_kernel", ps.wave_ms), ("nfc_scan_kernel", ps.scan_ms), ("nfc_demod_fixed_kernel", ps.kernel_ms)), key=lambda kv: kv[1])
            if kern[1] > 0:
                ach = 8.0 * n_streams * n_samples * steps / (kern[1] * 1e-3) / 1e9
                point["roofline"] = {"bound": "hbm", "kernel": kern[0], "kernel_ms_per_step": round(kern[1] / steps, 3), "achieved": round(ach, 3),
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6)}
            if lib is not None and check:
                badp, ref_frames = 0, 0
                for s in sorted(set([0, n_streams // 2, n_streams - 1] + list(range(min(n_streams, check)))))[:max(check, 1)]:
                    mag = torch.sqrt(buf[s, :, 0] ** 2 + buf[s, :, 1] ** 2).cpu().numpy().astype(np.float32)
                    fr, _ = TL.reference_decode(mag, sample_rate=FS, chunk=65536, keep_carrier=True, cap=1 << 17, defined_storage=True)
                    ref_frames += len(fr)
                    badp += 0 if pframes.get(f0 + s, []) == fr else 1
                point["parity"] = {"streams_checked": min(n_streams, max(check, 1)), "streams_mismatching": badp, "reference_frames": ref_frames}
            return point, ps

        def run_fixtures():
            """BASELINE configs 2-4 as they are written: every bundled capture as ONE 10 MS/s stream through the library in one
            submission (magnitudes resident in HBM), frames against the golden vectors the reference's own test holds"""
            per = {}
            total_n, total_t, bad = 0, 0.0, 0
            torch.cuda.empty_cache()
            g = nfclab_amd.NfcGpu(device=local, max_streams=64, frame_sink_bytes=8 << 20)
            for name in TL.fixture_names():
                mag = torch.from_numpy(TL.load_fixture(name)).to(dev)
                n = int(mag.numel())
                torch.cuda.synchronize()
                times, got = [], None
                for attempt in range(2):  # the first decode of a length sizes the library's work buffers
                    sid = g.open(nfclab_amd.default_params(), count=1)
                    g.sync()
                    ta = time.perf_counter()
                    g.submit_uniform(sid, 1, mag.data_ptr(), n * 4, n, FS, stride=1)
                    g.sync()
                    tb = time.perf_counter()
                    got = g.poll(sid, capacity=1 << 16)
                    g.close_stream(sid)
                    times.append(tb - ta)
                want = TL.load_golden(name)
                same = [f for f in got if f[1] in (0x0102, 0x0103)] == want  # poll / listen frames, as test-sdr writes them
                per[name] = {"samples": n, "ms": round(times[-1] * 1e3, 3), "Msamples_per_s": round(n / times[-1] / 1e6, 3), "frames": len(want),
                             "matches_golden": same}
                total_n += n
                total_t += times[-1]
                bad += 0 if same else 1
                del mag
            g.close()
            rates = sorted(v["Msamples_per_s"] for v in per.values())
            return {"workload": "each of the %d bundled captures as one 10 MS/s stream, one submission, magnitudes resident in HBM, all four "
                                "decoders; second decode of each length timed" % len(per),
                    "value": round(total_n / total_t / 1e6, 3), "unit": "Msamples/s", "slowest": rates[0], "median": rates[len(rates) // 2],
                    "fastest": rates[-1], "real_time_factor_slowest": round(rates[0] * 1e6 / FS, 3),
                    "captures_not_matching_golden": bad, "captures": per}

        want = [p for p in args.points.split(",") if p]
        scan_stats = None
        idle_stats = None
        for name in want:
            if time.perf_counter() - t_start > args.points_budget:
                points[name] = {"skipped": "the points' time budget (%d s since start) was used up" % args.points_budget}
                continue
            t_point = time.perf_counter()
            try:
                if name == "fixtures_single":
                    points[name] = run_fixtures()
                elif name == "config5_dense":
                    points[name], _ = run_point(name, args.config5_streams, args.config5_samples, False, 1, 1, 64)
                elif name == "config5_sparse":
                    points[name], scan_stats = run_point(name, args.config5_streams, args.config5_samples, True, 2, 1, 64)
                elif name == "config5_idle":
                    points[name], idle_stats = run_point(name, args.config5_streams, args.config5_samples, False, 2, 1, 64, idle=True)
                elif name == "share_dense":
                    points[name], _ = run_point(name, args.share_streams, args.config5_samples, False, 1, 1, 64)
                elif name == "share_sparse":
                    points[name], _ = run_point(name, args.share_streams, args.config5_samples, True, 2, 1, 64)
                elif name == "share_dense_h2d":
                    points[name], _ = run_point(name, 128, args.config5_samples, False, 1, 1, 8, from_host=True)
                elif name == "s2_share":
                    points[name], _ = run_point(name, args.share_streams, args.config5_samples, False, 1, 1, 64, offgrid=True)
                elif name == "s2_config5":
                    # config 5's shape off the capture grid: what 4096 radios deliver (every stream by its carry lane alone)
                    points[name], _ = run_point(name, args.config5_streams, args.config5_samples, False, 1, 1, 64, offgrid=True)
                elif name == "saturating":
                    points[name], _ = run_point(name, args.saturating_streams, 8192, False, 6, 2, 64)
                elif name == "single_dense":
                    points[name], _ = run_point(name, 1, args.single_dense_samples, False, 1, 1, 1)
                elif name == "single_sparse":
                    points[name], _ = run_point(name, 1, args.single_samples, True, 2, 1, 1)
            except Exception as exc:  # a point that fails must not take the headline with it
                points[name] = {"error": repr(exc)}
            if isinstance(points.get(name), dict):
                points[name]["wall_s"] = round(time.perf_counter() - t_point, 1)
        result["config"]["points"] = points

        # the search kernel of the time-parallel path (nfc_scan_kernel): on the idle point it is the whole job; on the sparse
        # point its time includes the second walk of the chunks whose seams did not verify
        def scan_line(stats, launches, note):
            n_scan = args.config5_streams * args.config5_samples
            ms = stats.scan_ms / launches
            ach = 8.0 * n_scan / (ms * 1e-3) / 1e9
            return {"achieved": round(ach, 3), "frac": round(ach / HBM_PEAK_GBS, 6), "kernel_ms_avg": round(ms, 4),
                    "frac_of_measured_peak": round(ach / read_peak, 6) if read_peak > 0 else None, "workload": note}

        if idle_stats is not None and idle_stats.scan_ms > 0:
            line = scan_line(idle_stats, 2, "config5_idle: %d streams x %d samples of unmodulated carrier" % (args.config5_streams, args.config5_samples))
            straffic, ssource = stored_traffic("nfc_scan_kernel", args.config5_streams, args.config5_samples)
            result["roofline_search"] = {
                "bound": "hbm",
                "achieved": line["achieved"],
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": line["frac"],
                "traffic": straffic,
                "traffic_source": ssource,
                "kernel": "nfc_scan_kernel",
                "kernel_ms_avg": line["kernel_ms_avg"],
                "algorithmic_bytes_per_launch": 8.0 * args.config5_streams * args.config5_samples,
                "peak_measured_streaming_read": round(read_peak, 1),
                "frac_of_measured_peak": line["frac_of_measured_peak"],
                "workload": line["workload"],
                "note": "the per-sample search kernel of the time-parallel path: exact front end + tile tests over every sample; it reads "
                        "every sample once plus the warm-up overlap of its chunks (4096 samples per chunk of up to 32768: 1.125 x the "
                        "algorithmic bytes at this size). Two timed launches, HIP events on the library's stream.",
            }
            if scan_stats is not None and scan_stats.scan_ms > 0:
                result["roofline_search"]["on_sparse_traffic"] = scan_line(
                    scan_stats, 2, "config5_sparse (re-walks of chunks whose seams did not verify included in the time)")

    if rank == 0:
        asked = headline_device_bytes(S, L, K, W, args.slices, world)
        result["consistency"] = {
            "headline_device_bytes_asked": int(asked),
            "headline_dataset": "%d slice(s) of %d samples per stream resident (%.1f GiB on this rank), step k submits slice k %% %d" % (
                NS, L, 8.0 * S * T / 2 ** 30, NS),
            "fits_in_driver_run": bool(asked < 200 * 2 ** 30),
            "wall_s": round(time.perf_counter() - t_start, 1),
        }
        print(json.dumps(result))

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
