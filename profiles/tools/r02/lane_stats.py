"""Where the lanes of the time-parallel path spend their steps, on the emulated runtime (same device functions as the
kernels, lanes one after the other): per decode pass the lanes, their steps and the longest lane, and - at every sample where
a lane met the lane after it and could not hand over - what state it was in. Backs the statements of DESIGN.md section 4b.
usage: python profiles/tools/r02/lane_stats.py > profiles/r02/lane_stats.json   (needs tests/hostsim/libnfcgpu_emulated.so)"""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
DRIVER = r'''
import os, sys
ROOT = %r
sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/nfc-laboratory_amd")
import numpy as np, nfc_testlib as T, nfclab_amd, synth
kind, S, L = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
template = synth.load_template(os.path.join(ROOT, "tests", "golden"))
segs = synth.sparse_segments(template)
if kind == "fixture":
    streams = [T.load_fixture(sys.argv[4])]
else:
    streams = [(synth.sparse_magnitude_f32(template, segs, s, 0, L) if kind == "sparse" else synth.magnitude_f32(template, s, 0, L)) for s in range(S)]
with nfclab_amd.NfcGpu(device=0, max_streams=max(64, len(streams))) as gpu:
    first = gpu.open(count=len(streams))
    gpu.submit_batch([first + i for i in range(len(streams))], [m.ctypes.data for m in streams], [m.size for m in streams], 10000000, stride=1)
    for i in range(len(streams)):
        gpu.poll(first + i, capacity=65536)
''' % ROOT
TECH = {"101": "NFC-A", "102": "NFC-B", "103": "NFC-F", "104": "NFC-V", "0": "searching"}

def run(kind, S, L, extra=None):
    env = dict(os.environ, NFCGPU_LIB=os.path.join(ROOT, "tests", "hostsim", "libnfcgpu_emulated.so"), NFCGPU_NO_TORCH="1",
               NFCGPU_WINDOW_DEBUG="1", NFC_EMU_DEBUG5="1")
    out = subprocess.run([sys.executable, "-c", DRIVER, kind, str(S), str(L)] + (extra or []), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True).stderr
    passes, blocked = [], collections.Counter()
    for line in out.splitlines():
        m = re.search(r"windowed pass (\d+): (\d+) lanes, (\d+) lane-steps .* longest lane (\d+) steps, (\d+) streams unsettled", line)
        if m:
            passes.append({"pass": int(m.group(1)), "lanes": int(m.group(2)), "lane_steps": int(m.group(3)), "longest_lane": int(m.group(4)), "streams_unsettled": int(m.group(5))})
        m = re.search(r"not comparable: lock (\w+) unlock \w+ bank (\d) run (\d+) .* type (\d+) fstart (\d+) towait (-?\d+)", line)
        if m:
            lock, bank, run_, ftype, fstart = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5))
            if lock == "0":
                why = "searching, detector bank stepped for fewer than 1024 samples since the last lock or gap"
            elif ftype == 259 and fstart == 0:
                why = TECH.get(lock, lock) + " locked: waiting for an answer"
            elif ftype == 259:
                why = TECH.get(lock, lock) + " locked: inside a listen frame"
            else:
                why = TECH.get(lock, lock) + " locked: inside a poll frame"
            blocked[why] += 1
    return {"workload": ("the capture %s as one stream" % extra[0]) if extra else "%s synthetic set, %d stream(s) x %d samples" % (kind, S, L), "passes": passes,
            "could_not_hand_over_at_a_meeting_sample": dict(blocked.most_common())}

out = {"note": __doc__.split("usage:")[0].strip(), "runs": [
    run("dense", 1, 1 << 22), run("sparse", 64, 1 << 20), run("fixture", 1, 0, ["test_NFC-V_26kbps_001"]), run("fixture", 1, 0, ["test_NFC-A_106kbps_004"]),
    run("fixture", 1, 0, ["test_POLL_ABF_001"])]}
print(json.dumps(out, indent=1))
