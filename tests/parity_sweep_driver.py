"""Driver of tests/test_parity_at_size.py. Parity at size, on the GPU: S streams x L samples x K submissions of the sparse /
dense synthetic set through the C ABI (IQ resident in HBM), EVERY stream compared frame by frame with the reference
decoder (oracle/_ref, one decoder per stream, a pool of host threads). Prints one JSON object.
usage: python tests/parity_sweep_driver.py sparse|dense|offgrid S L K   (NFCGPU_* knobs apply; the test sets none)"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np, torch
import nfclab_amd, synth, frames as framelib
import nfc_testlib as TL

kind, S, L, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
FIRST = int(os.environ.get("PARITY_FIRST_STREAM", "0"))  # (another draw of the synthetic set: streams FIRST .. FIRST + S - 1)
FS = 10000000
dev = torch.device("cuda", 0)
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
template_dev = torch.from_numpy(template.astype(np.int16)).to(dev)
T = K * L
data = torch.empty((S, T, 2), dtype=torch.float32, device=dev)
if kind == "sparse":
    synth.fill_sparse_iq_torch(data, template_dev, synth.sparse_segments(template), first_stream=FIRST, chunk_streams=max(1, min(256, (1 << 26) // T)))
else:
    synth.fill_iq_torch(data, template_dev, first_stream=FIRST, chunk_streams=max(1, min(1024, (1 << 26) // T)))
if kind == "offgrid":
    # set S2: the dense S1 magnitudes on a random phase per stream plus white noise of sigma 0.002 on both components,
    # fp32 IQ - what a radio delivers, off the capture grid (the generator of bench.py's s2 points)
    gen = torch.Generator(device=dev)
    gen.manual_seed(20260927)
    for s0 in range(0, S, 64):
        s1 = min(S, s0 + 64)
        m = torch.sqrt(data[s0:s1, :, 0] ** 2 + data[s0:s1, :, 1] ** 2)
        phi = torch.rand((s1 - s0, 1), device=dev, generator=gen) * 6.283185307179586
        data[s0:s1, :, 0] = m * torch.cos(phi) + torch.randn(m.shape, device=dev, generator=gen) * 0.002
        data[s0:s1, :, 1] = m * torch.sin(phi) + torch.randn(m.shape, device=dev, generator=gen) * 0.002
        del m
sink_words = 128 << 20
sink = torch.zeros(sink_words, dtype=torch.int32, device=dev)
ctl = torch.zeros(4, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
gpu = nfclab_amd.NfcGpu(device=0, max_streams=max(64, S), frame_sink_bytes=1 << 20)
gpu.sink_attach(sink.data_ptr(), sink_words, ctl.data_ptr())
gpu.sink_hold(True)
first = gpu.open(nfclab_amd.default_params(), count=S)
t0 = time.perf_counter()
for k in range(K):
    gpu.submit_uniform(first, S, data.data_ptr() + k * L * 8, T * 8, L, FS, stride=2)
gpu.sync()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = gpu.stats()
used = int(ctl[0].item())
got = framelib.parse_sink(sink[:used].cpu().numpy(), used, FS)
lib = TL.reference_lib()
assert lib is not None, "oracle/_ref not available"

def check(s0):
    bad, frames = [], 0
    # the oracle's input is the reference's own IQ -> magnitude step (RadioDeviceTask.cpp:626-642, the scalar formula compiled with
    # the reference's flags: oracle/ref_capi.cpp nfcref_magnitude), not a root taken on the device
    for s1 in range(s0, min(S, s0 + 64), 8):  # (eight streams of IQ on the host at a time per worker)
        iq = np.ascontiguousarray(data[s1:s1 + 8].cpu().numpy())
        for i in range(iq.shape[0]):
            mag = np.empty(iq.shape[1], dtype=np.float32)
            lib.nfcref_magnitude(iq[i].ctypes.data, iq.shape[1], mag.ctypes.data)
            fr, _ = TL.reference_decode(mag, sample_rate=FS, chunk=65536, keep_carrier=True, cap=1 << 17, defined_storage=True)
            frames += len(fr)
            if got.get(first + s1 + i, []) != fr:
                bad.append(s1 + i)
    return bad, frames

bad, ref_frames = [], 0
with ThreadPoolExecutor(max_workers=min(96, (os.cpu_count() or 8))) as pool:
    for b, f in pool.map(check, range(0, S, 64)):
        bad += b
        ref_frames += f
print(json.dumps({"set": kind, "streams": S, "samples_per_submission": L, "submissions": K, "knobs": {k: v for k, v in os.environ.items() if k.startswith("NFCGPU_")},
                  "gpu_seconds": round(dt, 3), "time_parallel_streams": int(st.windowed_streams), "sequential_streams": int(st.fallback_streams),
                  "lanes": int(st.windows), "decode_passes": int(st.window_passes), "frames_dropped": int(ctl[1].item()),
                  "streams_compared": S, "reference_frames": ref_frames, "streams_mismatching": bad}))
gpu.close()
