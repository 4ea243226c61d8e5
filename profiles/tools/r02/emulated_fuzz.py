"""Randomised parity of the host runtime + time-parallel path on the emulated runtime (test infrastructure; see profiles/r02/emulated_fuzz.json).
usage: NFCGPU_LIB=tests/hostsim/libnfcgpu_emulated.so python profiles/tools/r02/emulated_fuzz.py <seed> <seconds> [small]
(small: at most 6 streams of at most 9 x 32768 samples per scenario - what the CPU suite runs; round 4: a quarter of the scenarios is taken off the
int16 grid - a gain and white noise, what a radio delivers -: carry lanes with walked sums)"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT+"/tests"); sys.path.insert(0, ROOT+"/nfc-laboratory_amd")
import numpy as np
import nfc_testlib as T, nfclab_amd, synth
FS=10000000
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
segs = synth.sparse_segments(template)
rng=np.random.default_rng(int(sys.argv[1]))
deadline=time.time()+float(sys.argv[2])
rounds=0; bad=[]
while time.time()<deadline:
    kind=rng.choice(["sparse","dense","mixed"])
    small=len(sys.argv)>3 and sys.argv[3]=="small"
    S=int(rng.integers(1,7 if small else 24)); L=int(rng.integers(3,10 if small else 40))*32768+int(rng.integers(0,4))*8191
    offgrid=bool(rng.random()<0.25)
    base=int(rng.integers(0,100000))
    streams=[]
    for i in range(S):
        k=kind if kind!="mixed" else rng.choice(["sparse","dense"])
        m=synth.sparse_magnitude_f32(template, segs, base+i, 0, L) if k=="sparse" else synth.magnitude_f32(template, base+i, 0, L)
        if offgrid:
            m=np.abs(m*np.float32(rng.uniform(0.7,1.1))+rng.normal(0.0,0.0006,m.size).astype(np.float32)).astype(np.float32)
        streams.append(m)
    nb=int(rng.integers(1,6))
    cuts=sorted(set([0,L]+[int(x) for x in rng.integers(1,L,size=nb-1)]))
    if os.environ.get("NFC_FUZZ_TRACE"):
        print("SCENARIO", json.dumps({"kind":str(kind),"S":S,"L":L,"base":base,"cuts":cuts,"round":rounds,"offgrid":offgrid}), file=sys.stderr, flush=True)
    want=[T.reference_decode(m, keep_carrier=True, cap=65536, defined_storage=True)[0] for m in streams]
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        first=gpu.open(count=S)
        for a,b in zip(cuts[:-1],cuts[1:]):
            parts=[np.ascontiguousarray(m[a:b]) for m in streams]
            gpu.submit_batch([first+i for i in range(S)],[p.ctypes.data for p in parts],[p.size for p in parts],FS,stride=1)
        got=[gpu.poll(first+i,capacity=65536) for i in range(S)]
    wrong=[i for i in range(S) if got[i]!=want[i]]
    rounds+=1
    if wrong:
        bad.append({"kind":str(kind),"S":S,"L":L,"base":base,"cuts":cuts,"wrong":wrong,"offgrid":offgrid})
        print("MISMATCH", bad[-1], flush=True)
print(json.dumps({"seed":int(sys.argv[1]),"rounds":rounds,"mismatches":bad}))
