import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"nfc-laboratory_amd")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np
import nfclab_amd, synth
template = synth.load_template(os.path.join(ROOT,"tests","golden"))
mag = synth.magnitude_f32(template, 7, 0, 1<<22)
N=65536
with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
    sid = gpu.open()
    # warm
    for k in range(8):
        gpu.submit(sid, np.ascontiguousarray(mag[k*N:(k+1)*N]), 10000000)
        gpu.poll(sid, capacity=4096)
    os.environ["NFCGPU_WINDOW_DEBUG"]="1"
    t0=time.perf_counter()
    for k in range(8, 14):
        t1=time.perf_counter()
        gpu.submit(sid, np.ascontiguousarray(mag[k*N:(k+1)*N]), 10000000)
        fr=gpu.poll(sid, capacity=4096)
        sys.stderr.write("== buffer %d: %.2f ms, %d frames\n" % (k, (time.perf_counter()-t1)*1e3, len(fr)))
    del os.environ["NFCGPU_WINDOW_DEBUG"]
    t0=time.perf_counter()
    for k in range(14, 62):
        gpu.submit(sid, np.ascontiguousarray(mag[k*N:(k+1)*N]), 10000000)
        gpu.poll(sid, capacity=4096)
    dt=time.perf_counter()-t0
    sys.stderr.write("48 buffers: %.2f ms each, %.2f MS/s\n" % (dt/48*1e3, 48*N/dt/1e6))
