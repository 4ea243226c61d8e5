"""N > 1 path on CPU: two gloo processes shard the streams by rank exactly like bench.py (weak scaling,
no data-path collective) and exchange their decoded frames with frames.gather_sinks()."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import nfc_testlib as T

sys.path.insert(0, os.path.join(T.ROOT, "nfc-laboratory_amd"))


def _worker(rank, world, port, names, queue):
    sys.path.insert(0, os.path.join(T.ROOT, "tests"))
    sys.path.insert(0, os.path.join(T.ROOT, "nfc-laboratory_amd"))
    import frames as framelib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # this rank's shard of the streams: stream id = global index, decoded with the CPU build of the step machine
        mine = {}
        for g, name in enumerate(names):
            if g % world == rank:
                mine[g] = T.hostsim_decode(T.load_fixture(name), keep_carrier=True)
        words = framelib.pack_frames(mine)
        sink = torch.zeros(max(len(words), 1) + 100, dtype=torch.int32)
        sink[:len(words)] = torch.from_numpy(words.copy())
        gathered, counts = framelib.gather_sinks(sink, len(words), world)
        merged = {}
        for r in range(world):
            merged.update(framelib.parse_sink(gathered[r].numpy(), counts[r], 10000000))
        queue.put((rank, counts, {k: v for k, v in merged.items()}))
    finally:
        dist.destroy_process_group()


def test_two_rank_stream_sharding_and_frame_gather(built):
    names = ["test_NFC-A_106kbps_001", "test_NFC-B_106kbps_001", "test_NFC-A_424kbps_001", "test_POLL_AB_001", "test_NFC-A_106kbps_002"]
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, names, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = [queue.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = {g: T.hostsim_decode(T.load_fixture(n), keep_carrier=True) for g, n in enumerate(names)}
    for rank, counts, merged in results:
        assert len(counts) == 2 and all(c > 0 for c in counts)
        assert merged == expected  # every rank ends up with every stream's frames, in stream order


def test_pack_parse_roundtrip():
    import frames as framelib
    frames = {3: [(0x101, 0x102, 1, 0x102, 105938, 10, 20, 10000000, b"\x52")],
              7: [(0x103, 0x103, 0, 0x103, 211875, 5, 99, 10000000, bytes(range(19))), (0x100, 0x101, 0, 0x101, 0, 1, 1, 10000000, b"")]}
    words = framelib.pack_frames(frames)
    assert framelib.parse_sink(words, len(words), 10000000) == frames
