import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"nfc-laboratory_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import nfclab_amd, synth
dev=torch.device("cuda",0)
S,L=4096,1<<20
template = synth.load_template(os.path.join(ROOT,"tests","golden"))
tdev=torch.from_numpy(template.astype(np.int16)).to(dev)
data=torch.empty((S,2*L,2),dtype=torch.float32,device=dev)
synth.fill_iq_torch(data,tdev,first_stream=0,chunk_streams=64)
torch.cuda.synchronize()
words=256<<20
sink=torch.zeros(words,dtype=torch.int32,device=dev); ctl=torch.zeros(4,dtype=torch.int32,device=dev)
gpu=nfclab_amd.NfcGpu(device=0,max_streams=S,frame_sink_bytes=1<<20)
gpu.sink_attach(sink.data_ptr(),words,ctl.data_ptr()); gpu.sink_hold(True)
first=gpu.open(nfclab_amd.default_params(),count=S)
pitch=2*L*8
for k in range(3):
    gpu.submit_uniform(first,S,data.data_ptr()+(k%2)*L*8,pitch,L,10000000,stride=2)
gpu.sync()
for dbg in (None,"1"):
    if dbg: os.environ["NFCGPU_WINDOW_DEBUG"]=dbg
    ts=[]
    for k in range(3,7):
        t0=time.perf_counter()
        gpu.submit_uniform(first,S,data.data_ptr()+(k%2)*L*8,pitch,L,10000000,stride=2)
        t1=time.perf_counter()
        gpu.sync()
        t2=time.perf_counter()
        ts.append((round((t1-t0)*1e3,1),round((t2-t1)*1e3,2)))
    sys.stderr.write("debug=%s submit/sync ms: %s\n"%(dbg,ts))
