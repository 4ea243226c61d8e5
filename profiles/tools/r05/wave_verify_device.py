#!/usr/bin/env python3
"""Once a round, on the MI355X: the wave decoder built with -DNFC_WAVE_VERIFY decodes every tile twice - with the
bulk paths of nfc_wave_fast.hpp and again sample by sample by the step machine alone - and counts the tiles whose results differ
in any word (decoder state, protocol state, rings, frame bytes, usage marks). Not a product build.

    make -C nfc-laboratory_amd verify                   (here: hipcc cross-compiles)
    gpurun -- python profiles/tools/r05/wave_verify_device.py > profiles/r05/wave_verify_device.json

Inputs: the 18 bundled captures, each as one stream (speculative windows forced on the short ones too), and 64 dense synthetic
streams x 2^20 in two submissions; the frames are compared with the golden vectors / the reference as well."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
DRIVER = r'''
import json, os, sys
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))
import numpy as np
import nfc_testlib as T, nfclab_amd, synth
out = {"captures": {}, "dense": {}}
for name in T.fixture_names():
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        sid = gpu.open()
        gpu.submit_batch([sid], [mag.ctypes.data], [mag.size], 10000000, stride=1)
        got = [f for f in gpu.poll(sid, capacity=1 << 16) if f[1] in (0x0102, 0x0103)]
    out["captures"][name] = {"samples": int(mag.size), "matches_golden": got == T.load_golden(name)}
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
S, L = 64, 1 << 20
streams = [synth.magnitude_f32(template, s, 0, L) for s in range(S)]
with nfclab_amd.NfcGpu(device=0, max_streams=S) as gpu:
    first = gpu.open(count=S)
    for pos in (0, L // 2):
        parts = [np.ascontiguousarray(m[pos:pos + L // 2]) for m in streams]
        gpu.submit_batch([first + i for i in range(S)], [p.ctypes.data for p in parts], [p.size for p in parts], 10000000, stride=1)
    got = [gpu.poll(first + i, capacity=1 << 16) for i in range(S)]
bad = 0
if T.reference_lib() is not None:
    for i in range(S):
        want, _ = T.reference_decode(streams[i], keep_carrier=True, cap=1 << 16, defined_storage=True)
        bad += 0 if got[i] == want else 1
    out["dense"] = {"streams": S, "samples_per_stream": L, "submissions": 2, "streams_differing_from_the_reference": bad}
else:
    out["dense"] = {"streams": S, "samples_per_stream": L, "submissions": 2, "streams_differing_from_the_reference": None}
print(json.dumps(out))
'''


def main():
    lib = os.path.join(ROOT, "nfc-laboratory_amd", "libnfcgpu_verify.so")
    if not os.path.exists(lib):
        raise SystemExit("build it first: make -C nfc-laboratory_amd verify")
    env = dict(os.environ, NFCGPU_LIB=lib, NFCGPU_WAVE_VERIFY_REPORT="1", NFCGPU_SOLO_SAMPLES="0", NFCGPU_WINDOWED_MIN="4096")
    run = subprocess.run([sys.executable, "-c", DRIVER, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=3000)
    if run.returncode != 0:
        raise SystemExit(run.stderr[-4000:])
    result = json.loads(run.stdout.strip().splitlines()[-1])
    tiles = differ = reports = 0
    first = None
    for line in run.stderr.splitlines():
        m = re.search(r"wave verify: (\d+) tiles decoded twice .* (\d+) differ(.*)", line)
        if m:
            reports += 1
            tiles += int(m.group(1))
            differ += int(m.group(2))
            if int(m.group(2)) and first is None:
                first = m.group(3).strip()
    result["wave_verify"] = {"library": "libnfcgpu_verify.so (csrc/nfc_wave.hip with -DNFC_WAVE_VERIFY)", "submissions_reported": reports,
                             "tiles_decoded_twice": tiles, "tiles_differing": differ, "first_difference": first,
                             "compared": "decoder state, protocol state (but the chain of frame records), history and correlation rings (but the "
                                         "product ring), frame bytes, usage marks, position in the tile - word for word after every tile"}
    result["captures_not_matching_golden"] = sum(0 if v["matches_golden"] else 1 for v in result["captures"].values())
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
