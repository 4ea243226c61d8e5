"""GPU probe of the time-parallel path: S streams x L samples of the S1 synthetic set resident in HBM, one submission per
step; prints wall time, scan / windowed-decode kernel spans, windows, passes, fallbacks, and checks a few streams frame by
frame against the reference. Usage: python profiles/tools/r02/probe.py S L [steps] [check] (env NFCGPU_* knobs apply)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np, torch
import nfclab_amd, synth, frames as framelib
import nfc_testlib as TL

S, L = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
CHECK = int(sys.argv[4]) if len(sys.argv) > 4 else 4
FS = 10000000
dev = torch.device("cuda", 0)
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)
T = K * L
data = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
if os.environ.get("PROBE_SPARSE") == "1":
    segs = synth.sparse_segments(template)
    synth.fill_sparse_iq_torch(data, template_dev, segs, first_stream=0, chunk_streams=max(1, min(256, (1 << 26) // T)))
else:
    synth.fill_iq_torch(data, template_dev, first_stream=0, chunk_streams=max(1, min(1024, (1 << 26) // T)))
if os.environ.get("PROBE_IDLE") == "1":
    # unmodulated carrier with a few LSB of (non-periodic, per-stream) noise on the int16 grid
    t = torch.arange(T, device=dev, dtype=torch.int64)
    for s0 in range(0, S, 256):
        s1 = min(S, s0 + 256)
        sid = torch.arange(s0, s1, device=dev, dtype=torch.int64)[:, None]
        h = (t[None, :] * 2654435761 + sid * 40503) & 0xFFFFFFFF
        h = (h ^ (h >> 15)) * 2246822519 & 0xFFFFFFFF
        h = (h ^ (h >> 13)) * 3266489917 & 0xFFFFFFFF
        h = h ^ (h >> 16)
        data[s0:s1, :, 0] = 0.25 + ((h % 9) - 4).to(torch.float32) / 32768.0
        data[s0:s1, :, 1] = 0.0
sink_words = 64 << 20
sink = torch.zeros(sink_words, dtype=torch.int32, device=dev)
ctl = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
gpu = nfclab_amd.NfcGpu(device=0, max_streams=S, frame_sink_bytes=1 << 20)
gpu.sink_attach(sink.data_ptr(), sink_words, ctl.data_ptr())
gpu.sink_hold(True)
gpu.profile(True)
first = gpu.open(nfclab_amd.default_params(), count=S)
out = {"streams": S, "samples": L, "steps": []}
for k in range(K):
    gpu.stats_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gpu.submit_uniform(first, S, data.data_ptr() + k * L * 8, T * 8, L, FS, stride=2)
    gpu.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = gpu.stats()
    out["steps"].append({"wall_ms": round(dt * 1e3, 3), "GS_per_s": round(S * L / dt / 1e9, 3), "scan_ms": round(st.scan_ms, 3),
                         "scan_GBps": round(8.0 * st.scan_samples / max(st.scan_ms, 1e-9) / 1e6, 1), "window_ms": round(st.window_ms, 3),
                         "legacy_kernel_ms": round(st.kernel_ms, 3), "windows": st.windows, "passes": st.window_passes,
                         "windowed": st.windowed_streams, "fallback": st.fallback_streams, "launches": st.launches})
used = int(ctl[0].item())
out["dropped"] = int(ctl[1].item())
words = sink[:used].cpu().numpy()
frames = framelib.parse_sink(words, used, FS)
out["frames"] = sum(len(v) for v in frames.values())
lib = TL.reference_lib()
if lib is not None and CHECK:
    bad = 0
    pick = sorted(set(i for i in [0, 1, S // 2, S - 1] + list(range(min(S, CHECK))) if 0 <= i < S))[:max(CHECK, 1)]
    for s in pick:
        mag = torch.sqrt(data[s, :, 0] ** 2 + data[s, :, 1] ** 2).cpu().numpy().astype(np.float32)
        fr, _ = TL.reference_decode(mag, sample_rate=FS, chunk=65536, keep_carrier=True, cap=65536, defined_storage=True)
        got = frames.get(first + s, [])
        if got != fr:
            bad += 1
            if bad <= 2:
                n = min(len(got), len(fr))
                d = next((i for i in range(n) if got[i] != fr[i]), n)
                print("stream", s, "differs at frame", d, "of", len(got), "/", len(fr), file=sys.stderr)
                print("  got ", got[d] if d < len(got) else None, "\n  want", fr[d] if d < len(fr) else None, file=sys.stderr)
    out["parity"] = {"checked": len(pick), "mismatching": bad}
print(json.dumps(out))
gpu.close()
