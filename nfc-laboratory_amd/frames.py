"""Frame-record plumbing shared by bench.py and the tests: parsing the packed frame sink and gathering the
variable-length per-rank sinks across ranks (RCCL over xGMI on GPUs, gloo in the CPU tests).

Sink layout (nfc_types.h NfcFrameRecord): 32-bit words
    [stream, tech, type, flags, phase, rate, start, end, length, payload((length+3)//4 words) ...]
"""
import numpy as np

HEADER_WORDS = 9


def parse_sink(words, used, sample_rate):
    """packed records -> {stream: [frame tuples in nfclab_amd.Frame.as_tuple() order]}."""
    frames = {}
    pos = 0
    words = np.asarray(words)
    while pos + HEADER_WORDS <= used:
        sid, tech, typ, flags, phase, rate, start, end, length = (int(v) & 0xFFFFFFFF for v in words[pos:pos + HEADER_WORDS])
        nwords = (length + 3) // 4
        payload = words[pos + HEADER_WORDS:pos + HEADER_WORDS + nwords].tobytes()[:length]
        frames.setdefault(sid, []).append((tech, typ, flags, phase, rate, start, end, sample_rate, payload))
        pos += HEADER_WORDS + nwords
    return frames


def pack_frames(frames_by_stream):
    """inverse of parse_sink (used by tests to fabricate sinks)."""
    out = []
    for sid, frames in frames_by_stream.items():
        for (tech, typ, flags, phase, rate, start, end, _fs, payload) in frames:
            out.extend([sid, tech, typ, flags, phase, rate, start, end, len(payload)])
            padded = payload + b"\0" * (-len(payload) % 4)
            out.extend(np.frombuffer(padded, dtype="<u4").tolist())
    return np.array(out, dtype=np.uint32).view(np.int32)


def gather_sinks(sink, used, world):
    """All-gather the used part of every rank's sink.

    sink: 1-D int32 tensor (device of the process group's backend); used: number of valid words on this rank.
    Returns (gathered, counts): gathered is [world, longest] int32 on the same device, counts a python list.
    One tiny all_gather for the counts, one padded all_gather_into_tensor for the records: frames are KBs-MBs,
    the exchange is latency bound, so two collectives beat per-peer send/recv on xGMI's point-to-point links."""
    import torch
    import torch.distributed as dist

    mine = torch.tensor([used], dtype=torch.int32, device=sink.device)
    counts = [torch.zeros(1, dtype=torch.int32, device=sink.device) for _ in range(world)]
    dist.all_gather(counts, mine)
    counts = [int(c.item()) for c in counts]
    longest = max(max(counts), 1)

    local = torch.zeros(longest, dtype=torch.int32, device=sink.device)
    local[:used] = sink[:used]
    gathered = torch.empty(world * longest, dtype=torch.int32, device=sink.device)
    dist.all_gather_into_tensor(gathered, local)
    return gathered.view(world, longest), counts
