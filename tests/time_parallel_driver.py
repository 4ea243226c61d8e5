"""Driver of tests/test_time_parallel.py (run in a subprocess so that the library under test - the real libnfcgpu.so or the
emulated runtime of tests/hostsim - and the knobs of the time-parallel path are chosen through the environment).
Prints one JSON object: per case the number of reference frames, whether the frames matched, and the path statistics."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))

import numpy as np

import nfc_testlib as T
import nfclab_amd
import synth

FS = 10000000


def decode(streams, buffers=1, stride=1):
    """streams: list of float32 magnitude arrays; returns (frames per stream, stats)"""
    with nfclab_amd.NfcGpu(device=0, max_streams=max(64, len(streams))) as gpu:
        first = gpu.open(count=len(streams))
        longest = max(m.size for m in streams)
        step = (longest + buffers - 1) // buffers
        for pos in range(0, longest, step):
            parts = []
            for m in streams:
                part = np.ascontiguousarray(m[pos:pos + step])
                if stride == 2:
                    part = np.ascontiguousarray(T.magnitude_to_iq(part, seed=3))
                parts.append(part)
            ids = [first + i for i, p in enumerate(parts) if p.size]
            ptrs = [p.ctypes.data for p in parts if p.size]
            cnts = [p.size // stride for p in parts if p.size]
            gpu.submit_batch(ids, ptrs, cnts, FS, stride=stride)
        got = [gpu.poll(first + i, capacity=16384) for i in range(len(streams))]
        st = gpu.stats()
    return got, {"windowed": int(st.windowed_streams), "fallback": int(st.fallback_streams), "windows": int(st.windows),
                 "passes": int(st.window_passes), "repairs": int(st.scan_repairs)}


def case(name, streams, **kw):
    want = [T.reference_decode(m, keep_carrier=True, cap=16384, defined_storage=True)[0] for m in streams]
    got, st = decode(streams, **kw)
    bad = [i for i in range(len(streams)) if got[i] != want[i]]
    return {"name": name, "frames": sum(len(w) for w in want), "mismatching": bad, "stats": st}


def main():
    which = sys.argv[1:]
    out = []
    template = synth.load_template(os.path.join(ROOT, "tests", "golden"))

    # "fixture:<name>": one capture; "fixture:<name>@<n>": its first n samples
    for name in [w.split(":", 1)[1] for w in which if w.startswith("fixture:")]:
        head = None
        if "@" in name:
            name, head = name.split("@", 1)
        mag = np.abs(T.load_fixture(name)).astype(np.float32)
        if head is not None:
            mag = np.ascontiguousarray(mag[:int(head)])
        out.append(case(name if head is None else "%s@%s" % (name, head), [mag]))

    if not which or "fixtures" in which:
        for name in T.fixture_names():
            mag = np.abs(T.load_fixture(name)).astype(np.float32)
            out.append(case(name, [mag]))

    if not which or "buffers" in which:
        for name in ("test_NFC-A_106kbps_003", "test_NFC-B_106kbps_002", "test_NFC-F_212kbps_001", "test_NFC-V_26kbps_002", "test_POLL_ABF_001"):
            mag = np.abs(T.load_fixture(name)).astype(np.float32)
            out.append(case(name + " in 3 buffers", [mag], buffers=3))

    if not which or "synthetic" in which:
        streams = [synth.magnitude_f32(template, s, 0, 1 << 18) for s in range(12)]
        out.append(case("12 synthetic streams x 2^18, IQ entry, 2 buffers", streams, buffers=2, stride=2))

    if not which or "carried" in which:
        # what a lane leaves behind without having looked at it (protocol state of a technology it never locked, an NFC-F
        # record it never ran) goes on to the next submission: dense streams in three submissions, sparse ones in four
        streams = [synth.magnitude_f32(template, s, 0, 1 << 20) for s in (29, 5, 17)]
        out.append(case("3 dense synthetic streams x 2^20 in 3 buffers", streams, buffers=3))
        segs = synth.sparse_segments(template)
        streams = [synth.sparse_magnitude_f32(template, segs, s, 0, 1 << 20) for s in range(24)]
        out.append(case("24 sparse synthetic streams x 2^20 in 4 buffers", streams, buffers=4))

    if not which or "offgrid" in which:
        mag = np.abs(T.load_fixture("test_NFC-A_106kbps_001")).astype(np.float32)
        mag = (mag * np.float32(1.0000153)).astype(np.float32)  # off the int16 grid: the carry lane alone, running sums walked
        out.append(case("off-grid magnitudes", [mag]))
        # what a radio delivers (set S2): every technology's captures scaled off the grid with white noise on top,
        # in two submissions (the sums a stream carries are not on a grid either)
        rng = np.random.default_rng(11)
        streams = []
        for name, gain in (("test_NFC-B_106kbps_001", 0.83), ("test_NFC-F_212kbps_001", 1.07), ("test_NFC-V_26kbps_002", 0.91), ("test_NFC-A_424kbps_001", 0.77),
                           ("test_POLL_ABF_001", 1.0))[:int(os.environ.get("NFC_TEST_OFFGRID_CAPTURES", "5"))]:
            m = np.abs(T.load_fixture(name)).astype(np.float32)
            streams.append(np.abs(m * np.float32(gain) + rng.normal(0.0, 0.0007, m.size).astype(np.float32)).astype(np.float32))
        out.append(case("captures off the grid with noise, 2 buffers", streams, buffers=2))
        n_syn = int(os.environ.get("NFC_TEST_OFFGRID_SYNTHETIC", "3"))
        streams = [np.abs(synth.magnitude_f32(template, 40 + s, 0, 1 << 18) * np.float32(0.93) + rng.normal(0.0, 0.0005, 1 << 18).astype(np.float32)).astype(np.float32)
                   for s in range(n_syn)]
        out.append(case("%d dense synthetic streams x 2^18 off the grid, IQ entry, 2 buffers" % n_syn, streams, buffers=2, stride=2))

    if "carried_dense" in which:
        streams = [synth.magnitude_f32(template, s, 0, 3 << 16) for s in (29, 5)]
        out.append(case("2 dense synthetic streams x 3 * 2^16 in 3 buffers", streams, buffers=3))

    if "beside" in which:
        # (tests/test_time_parallel.py: a submission of more than 4 Mi samples - its planes are written beside the rounds of second walks)
        streams = [synth.magnitude_f32(template, 90 + 7 * s, 0, 1 << 20) for s in range(5)]
        out.append(case("5 dense synthetic streams x 2^20", streams))

    if "planes" in which:
        # (tests/test_time_parallel.py: the front-end planes - 16 bytes per sample - do not fit the device)
        streams = [synth.magnitude_f32(template, 60 + s, 0, 1 << 19) for s in range(2)]
        out.append(case("2 dense synthetic streams x 2^19", streams))

    if not which or "quiet" in which:
        # long quiet carrier around one exchange: the case the path is for (nearly everything skipped)
        rng = np.random.default_rng(5)
        fix = np.abs(T.load_fixture("test_NFC-A_106kbps_001")).astype(np.float32)
        level = np.float32(np.median(fix[:8000]))
        def idle(n):
            return (np.round((level + rng.normal(0, 0.0008, n)) * 32768.0) / 32768.0).astype(np.float32)
        mag = np.concatenate([idle(300000), fix[9000:70000], idle(500000)])
        out.append(case("one exchange in quiet carrier", [mag]))

    print(json.dumps(out))


if __name__ == "__main__":
    main()
