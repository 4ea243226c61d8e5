"""Trace (.trz) writer for decoded frames: the on-disk format the reference application opens (File > Open) and
`tools/py_nfclab` reads, so that the output of a many-stream GPU run can be inspected with the reference's own tools
(SURVEY.md 8(f) rank 4).

Format, as written by the reference's TraceStorageTask::writeFrameEntry
(src/nfc-lib/lib-lab/lab-tasks/src/main/cpp/tasks/TraceStorageTask.cpp:461-520) and read back by its readFrameEntry
(:380-449): a gzip-compressed tar archive with one entry `frame.json` = {"frames": [ {...}, ... ]}, one object per frame
with sampleStart, sampleEnd, sampleRate, timeStart, timeEnd, techType, frameType, frameRate, frameFlags, framePhase,
dateTime and, for frames with payload, frameData ("26" / "04:00" ... upper-case hex separated by colons) and length.
Times follow the decoder: timeStart = sampleStart / sampleRate, dateTime = streamTime + timeStart
(NfcA.cpp frame construction, NfcDecoder.cpp:449-463).

Frames are the tuples used throughout the tests and bench (nfclab_amd.Frame.as_tuple()):
    (techType, frameType, frameFlags, framePhase, frameRate, sampleStart, sampleEnd, sampleRate, payload bytes)
"""
import io
import json
import tarfile


def frame_entry(frame, stream_time=0.0):
    tech, ftype, flags, phase, rate, start, end, fs, payload = frame
    time_start = float(start) / float(fs) if fs else 0.0
    time_end = float(end) / float(fs) if fs else 0.0
    entry = {
        "sampleStart": int(start),
        "sampleEnd": int(end),
        "sampleRate": int(fs),
        "timeStart": time_start,
        "timeEnd": time_end,
        "techType": int(tech),
        "frameType": int(ftype),
        "frameRate": int(rate),
        "frameFlags": int(flags),
        "framePhase": int(phase),
        "dateTime": float(stream_time) + time_start,
    }
    if payload:
        entry["frameData"] = ":".join("%02X" % b for b in payload)
        entry["length"] = len(payload)
    return entry


def write_trz(path, frames, stream_time=0.0):
    """Write one stream's frames (in stream order) to `path` (must end in .trz)."""
    content = json.dumps({"frames": [frame_entry(f, stream_time) for f in frames]}).encode("ascii")
    with tarfile.open(path, "w:gz", compresslevel=9, format=tarfile.USTAR_FORMAT) as tar:
        info = tarfile.TarInfo("frame.json")
        info.size = len(content)
        info.mode = 0o664
        tar.addfile(info, io.BytesIO(content))
    return len(content)


def write_trz_per_stream(directory, frames_by_stream, stream_time=0.0, prefix="stream"):
    """One trace per stream of a batch run ({stream id: [frames]}, as frames.parse_sink returns); returns the paths."""
    import os
    paths = []
    for sid in sorted(frames_by_stream):
        path = os.path.join(directory, "%s_%06d.trz" % (prefix, sid))
        write_trz(path, frames_by_stream[sid], stream_time)
        paths.append(path)
    return paths
