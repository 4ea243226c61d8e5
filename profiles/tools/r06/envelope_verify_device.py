#!/usr/bin/env python3
"""Round 6, on the MI355X: the envelope kernel built with -DNFC_ENVELOPE_VERIFY_BUILD (libnfcgpu_envverify.so: every tile its
grouped walk takes is walked again by the statement nfc_envelope_step and compared bit for bit) over the 18 bundled captures and
256 dense streams x 2^20; frames compared with the golden vectors / the reference. The library prints its running totals on
stdout at every launch of the kernel ("[envelope verify] tiles differing"); this script runs the decodes in a child process and
reports the last totals. usage: NFCGPU_LIB=.../libnfcgpu_envverify.so python envelope_verify_device.py"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CHILD = r'''
import json, os, sys
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np
import nfc_testlib as T, nfclab_amd, synth
out = {"captures_not_matching_golden": []}
for name in T.fixture_names():
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        sid = gpu.open()
        gpu.submit_batch([sid], [mag.ctypes.data], [mag.size], 10000000, stride=1)
        got = [f for f in gpu.poll(sid, capacity=1 << 16) if f[1] in (0x0102, 0x0103)]
    if got != T.load_golden(name):
        out["captures_not_matching_golden"].append(name)
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
S, L = 256, 1 << 20
streams = [synth.magnitude_f32(template, s, 0, L) for s in range(S)]
with nfclab_amd.NfcGpu(device=0, max_streams=S) as gpu:
    first = gpu.open(count=S)
    gpu.submit_batch([first + i for i in range(S)], [m.ctypes.data for m in streams], [m.size for m in streams], 10000000, stride=1)
    got = [gpu.poll(first + i, capacity=1 << 16) for i in range(S)]
    # one more small submission: its first launch of the kernel prints the totals of everything before
    gpu.submit_batch([first], [streams[1].ctypes.data], [1 << 17], 10000000, stride=1)
    gpu.poll(first, capacity=1 << 16)
bad = 0
for i in range(0, S, 8):
    want = T.reference_decode(streams[i], keep_carrier=True, cap=1 << 16, defined_storage=True)[0]
    bad += int(got[i] != want)
out["dense_streams"] = S
out["dense_streams_compared"] = len(range(0, S, 8))
out["dense_streams_differing"] = bad
print("RESULT " + json.dumps(out))
'''
r = subprocess.run([sys.executable, "-c", CHILD, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
totals = [tuple(int(v) for v in m.groups()) for m in re.finditer(r"\[envelope verify\] (\d+) (\d+)\n", r.stdout)]
res = [json.loads(l[7:]) for l in r.stdout.splitlines() if l.startswith("RESULT ")]
out = res[0] if res else {"error": r.stderr[-2000:]}
out["library"] = os.environ.get("NFCGPU_LIB")
out["launches_of_the_kernel"] = len(totals)
out["tiles_walked_by_groups_and_again_by_the_statement"] = max([t[0] for t in totals] or [0])
out["tiles_differing"] = max([t[1] for t in totals] or [0])
out["differences_printed"] = [l for l in r.stdout.splitlines() if "differs" in l][:8]
print(json.dumps(out, indent=1))
