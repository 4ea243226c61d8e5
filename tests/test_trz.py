"""Trace writer: what nfclab_amd/trz.py writes must be readable by the reference's own
tools/py_nfclab reader (when the reference tree is present) and carry exactly the fields the reference's
TraceStorageTask::readFrameEntry requires."""
import json
import os
import sys
import tarfile

import pytest

import nfc_testlib as T

sys.path.insert(0, os.path.join(T.ROOT, "nfc-laboratory_amd"))
import trz  # noqa: E402

REQUIRED = ("techType", "frameType", "framePhase", "frameFlags", "frameRate", "sampleStart", "sampleEnd", "sampleRate",
            "timeStart", "timeEnd", "dateTime")


def test_trace_has_the_fields_the_reference_reader_requires(tmp_path):
    frames = T.load_golden("test_POLL_ABF_001")
    path = str(tmp_path / "poll.trz")
    trz.write_trz(path, frames, stream_time=1700000000)
    with tarfile.open(path, "r:gz") as tar:
        data = json.load(tar.extractfile(tar.getmember("frame.json")))
    assert len(data["frames"]) == len(frames)
    golden = json.load(open(os.path.join(T.GOLDEN, "wav", "test_POLL_ABF_001.json")))["frames"]
    for entry, frame, ref in zip(data["frames"], frames, golden):
        for key in REQUIRED:
            assert key in entry
        # same values as the reference wrote into its own golden JSON for these frames
        for key in ("techType", "frameType", "framePhase", "frameFlags", "frameRate", "sampleStart", "sampleEnd", "sampleRate"):
            assert entry[key] == ref[key]
        assert entry.get("frameData", "") == ref["frameData"]
        assert entry["timeStart"] == ref["timeStart"] and entry["timeEnd"] == ref["timeEnd"]
        assert entry["dateTime"] == 1700000000 + ref["timeStart"]


def test_trace_opens_with_the_reference_python_reader(tmp_path):
    tools = os.path.join(os.environ.get("NFC_REFERENCE_ROOT", "/root/reference"), "tools")
    if not os.path.isdir(os.path.join(tools, "py_nfclab")):
        pytest.skip("reference tree not present")
    sys.path.insert(0, tools)
    try:
        from py_nfclab.readers import read_trz
    except Exception as exc:  # the reference package may need modules this image lacks
        pytest.skip("py_nfclab not importable here: %r" % (exc,))
    frames = T.load_golden("test_NFC-A_106kbps_001") + T.load_golden("test_NFC-V_26kbps_001")
    path = str(tmp_path / "mixed.trz")
    trz.write_trz(path, frames)
    got = read_trz(path)
    assert len(got) == len(frames)
    for g, f in zip(got, frames):
        assert (int(g.tech_type), int(g.frame_type), g.sample_start, g.sample_end, g.sample_rate, g.frame_rate) == \
               (f[0], f[1], f[5], f[6], f[7], f[4])
        assert bytes(g.data) == f[8]


TRACE_REF = os.path.join(T.ROOT, "oracle", "_ref", "trace-ref")


def _frame_lines(entries):
    return ["%d %d %d %d %d %d %d %d %.17g %.17g %.17g %s" % (
        e["techType"], e["frameType"], e["frameFlags"], e["framePhase"], e["frameRate"], e["sampleStart"], e["sampleEnd"],
        e["sampleRate"], e["timeStart"], e["timeEnd"], e["dateTime"], e.get("frameData", "").replace(":", "") or "-") for e in entries]


@pytest.mark.skipif(not os.path.exists(TRACE_REF), reason="trace-ref not built (needs the reference tree and zlib at build time)")
def test_trace_against_the_reference_trace_storage_task(tmp_path):
    """The task behind "open trace" / "save trace" of the application (lab-tasks TraceStorageTask.cpp, built in place with the
    reference's own tar + zlib code, driven through its subjects by tests/dropin/trace_harness.cpp): it reads what trz.py
    wrote and publishes the same frames, and what it writes itself for those frames is entry for entry what trz.py wrote."""
    import subprocess
    if T.reference_lib() is None:
        pytest.skip("oracle/_ref not built")
    frames, _ = T.reference_decode(T.load_fixture("test_POLL_ABF_001"), keep_carrier=True)
    more, _ = T.reference_decode(T.load_fixture("test_NFC-V_26kbps_001"), keep_carrier=True)
    frames = frames + more
    mine = str(tmp_path / "mine.trz")
    trz.write_trz(mine, frames, stream_time=1700000000)
    with tarfile.open(mine, "r:gz") as tar:
        entries = json.load(tar.extractfile(tar.getmember("frame.json")))["frames"]
    assert len(entries) == len(frames)

    # read by the reference
    run = subprocess.run([TRACE_REF, "read", mine], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    got = run.stdout.splitlines()
    assert len(got) == len(frames)
    for line, f, e in zip(got, frames, entries):
        w = line.split()
        assert [int(v) for v in w[:8]] == list(f[:8])
        assert (bytes.fromhex(w[11]) if w[11] != "-" else b"") == f[8]
        assert float(w[8]) == pytest.approx(e["timeStart"], abs=1e-9) and float(w[10]) == pytest.approx(e["dateTime"], abs=1e-6)

    # written by the reference
    listing = tmp_path / "frames.txt"
    listing.write_text("\n".join(_frame_lines(entries)) + "\n")
    theirs = str(tmp_path / "theirs.trz")
    run = subprocess.run([TRACE_REF, "write", theirs, str(listing)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    with tarfile.open(theirs, "r:gz") as tar:
        assert tar.getnames() == ["frame.json"]
        ref_entries = json.load(tar.extractfile(tar.getmember("frame.json")))["frames"]
    assert ref_entries == entries


# ---- the writer behind the C ABI (nfc-laboratory_amd/csrc/nfc_trace.hip): nfcgpu_trace_write_frames / nfcgpu_trace_write ----

def _c_frames(frames):
    import nfclab_amd
    arr = (nfclab_amd.Frame * len(frames))()
    for a, (tech, ftype, flags, phase, rate, start, end, fs, payload) in zip(arr, frames):
        a.tech_type, a.frame_type, a.frame_flags, a.frame_phase, a.frame_rate = tech, ftype, flags, phase, rate
        a.sample_start, a.sample_end, a.sample_rate, a.length = start, end, fs, len(payload)
        for i, b in enumerate(payload):
            a.data[i] = b
    return arr


def _read_entries(path):
    with tarfile.open(path, "r:gz") as tar:
        assert tar.getnames() == ["frame.json"]
        return json.load(tar.extractfile(tar.getmember("frame.json")))["frames"]


EMU = os.path.join(T.ROOT, "tests", "hostsim", "libnfcgpu_emulated.so")


def _abi():
    """the C ABI without a device: the trace writer is host code, the emulated test build of the library exports it too"""
    import ctypes
    if not os.path.exists(EMU):
        import subprocess
        subprocess.check_call(["bash", os.path.join(T.ROOT, "tests", "hostsim", "build_emulated.sh")])
    lib = ctypes.CDLL(EMU)
    lib.nfcgpu_trace_write_frames.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int64, ctypes.c_double, ctypes.c_double,
                                              ctypes.POINTER(ctypes.c_uint32)]
    return lib


def test_c_abi_writer_writes_what_the_python_writer_writes(tmp_path):
    import ctypes
    frames = T.load_golden("test_POLL_ABF_001") + T.load_golden("test_NFC-V_26kbps_001") + [(0x100, 0x101, 0, 0x101, 0, 77, 77, 10000000, b"")]
    mine, theirs = str(tmp_path / "c.trz"), str(tmp_path / "py.trz")
    written = ctypes.c_uint32()
    arr = _c_frames(frames)
    assert _abi().nfcgpu_trace_write_frames(mine.encode(), ctypes.byref(arr), len(frames), 1700000000, 0.0, 0.0, ctypes.byref(written)) == 0
    assert written.value == len(frames)
    trz.write_trz(theirs, frames, stream_time=1700000000)
    assert _read_entries(mine) == _read_entries(theirs)


def test_c_abi_writer_keeps_the_frames_of_a_time_range_and_shifts_them(tmp_path):
    """TraceStorageTask.cpp:461-483: frames that start before the range or end after it are left out, the others move to
    its start (times, sample numbers); dateTime stays"""
    import ctypes
    frames = T.load_golden("test_NFC-A_106kbps_001")
    t0 = frames[3][5] / 1e7 - 1e-4
    t1 = frames[9][6] / 1e7 + 1e-4
    path = str(tmp_path / "range.trz")
    written = ctypes.c_uint32()
    arr = _c_frames(frames)
    assert _abi().nfcgpu_trace_write_frames(path.encode(), ctypes.byref(arr), len(frames), 5, t0, t1, ctypes.byref(written)) == 0
    entries = _read_entries(path)
    assert written.value == len(entries) == 7
    offset = int(1e7 * t0)
    for e, f in zip(entries, frames[3:10]):
        assert e["sampleStart"] == f[5] - offset and e["sampleEnd"] == f[6] - offset
        assert e["timeStart"] == f[5] / 1e7 - t0 and e["dateTime"] == 5 + f[5] / 1e7


def test_c_abi_trace_opens_with_the_reference_python_reader(tmp_path):
    import ctypes
    tools = os.path.join(os.environ.get("NFC_REFERENCE_ROOT", "/root/reference"), "tools")
    if not os.path.isdir(os.path.join(tools, "py_nfclab")):
        pytest.skip("reference tree not present")
    sys.path.insert(0, tools)
    try:
        from py_nfclab.readers import read_trz
    except Exception as exc:
        pytest.skip("py_nfclab not importable here: %r" % (exc,))
    frames = T.load_golden("test_NFC-B_106kbps_001") + T.load_golden("test_NFC-F_212kbps_001")
    path = str(tmp_path / "c.trz")
    arr = _c_frames(frames)
    assert _abi().nfcgpu_trace_write_frames(path.encode(), ctypes.byref(arr), len(frames), 0, 0.0, 0.0, None) == 0
    got = read_trz(path)
    assert len(got) == len(frames)
    for g, f in zip(got, frames):
        assert (int(g.tech_type), int(g.frame_type), g.sample_start, g.sample_end, g.sample_rate, g.frame_rate) == (f[0], f[1], f[5], f[6], f[7], f[4])
        assert bytes(g.data) == f[8]


@pytest.mark.skipif(not os.path.exists(TRACE_REF), reason="trace-ref not built (needs the reference tree and zlib at build time)")
def test_c_abi_trace_is_read_by_the_reference_trace_storage_task(tmp_path):
    """the reference's own task (tar + zlib inflate of the reference) reads the archive the C ABI wrote - a gzip member of
    stored deflate blocks - and publishes the same frames"""
    import ctypes
    import subprocess
    frames = T.load_golden("test_POLL_ABF_001")
    path = str(tmp_path / "c.trz")
    arr = _c_frames(frames)
    assert _abi().nfcgpu_trace_write_frames(path.encode(), ctypes.byref(arr), len(frames), 1700000000, 0.0, 0.0, None) == 0
    run = subprocess.run([TRACE_REF, "read", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    got = run.stdout.splitlines()
    assert len(got) == len(frames)
    for line, f in zip(got, frames):
        w = line.split()
        assert [int(v) for v in w[:8]] == list(f[:8])
        assert (bytes.fromhex(w[11]) if w[11] != "-" else b"") == f[8]


@pytest.mark.gpu
def test_trace_of_device_decoded_frames_through_the_c_abi(built, tmp_path):
    """nfcgpu_trace_write: the frames the device decoded for a stream (waiting in its queue) as a .trz - the capture decoded
    on the GPU, the trace equal to what the golden frames give"""
    import numpy as np
    import nfclab_amd
    name = "test_NFC-A_106kbps_001"
    mag = np.abs(T.load_fixture(name)).astype(np.float32)
    path = str(tmp_path / "gpu.trz")
    with nfclab_amd.NfcGpu(device=0, max_streams=64) as gpu:
        sid = gpu.open()
        for pos in range(0, mag.size, 65536):
            gpu.submit(sid, mag[pos:pos + 65536], 10000000)
        n = gpu.trace_write(sid, path)
        queued = gpu.poll(sid)          # (a trace does not take the frames out of the queue)
    want = [f for f in queued]
    assert n == len(want) and [f for f in want if f[1] in (0x102, 0x103)] == T.load_golden(name)
    ref = str(tmp_path / "py.trz")
    trz.write_trz(ref, want)
    assert _read_entries(path) == _read_entries(ref)
