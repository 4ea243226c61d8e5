"""Driver of tests/test_gather_two_ranks.py: one process per GPU (launched by torch.distributed.run), each decodes its own
captures on its own GPU into a held sink, then every rank gathers every rank's frame records through the C ABI
(nfcgpu_gather_frames over RCCL) and checks them against the reference decoder. Prints "rank R ok"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "nfc-laboratory_amd"))

import numpy as np
import torch
import torch.distributed as dist

import frames as framelib
import nfc_testlib as T
import nfclab_amd

FS = 10000000
NAMES = ["test_NFC-A_106kbps_001", "test_POLL_AB_001", "test_NFC-B_106kbps_001", "test_NFC-A_424kbps_001"]


def main():
    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # (only carries the unique id)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ranks hold different amounts of records (rank r decodes r + 1 captures): the exact-size gather is what is tested
    mine = [NAMES[(rank + k) % len(NAMES)] for k in range(rank + 1)]
    sink = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
    ctl = torch.zeros(4, dtype=torch.int32, device=dev)
    out = torch.zeros(1 << 21, dtype=torch.int32, device=dev)

    with nfclab_amd.NfcGpu(device=local, max_streams=64) as g:
        g.sink_attach(sink.data_ptr(), sink.numel(), ctl.data_ptr())
        g.sink_hold(True)
        first = g.open(count=len(mine))
        for i, name in enumerate(mine):
            g.submit(first + i, np.abs(T.load_fixture(name)).astype(np.float32), FS)
        g.sync()

        ident = [g.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        g.comm_init(ident[0], rank, world)
        counts, stride = g.gather_frames(out.data_ptr(), out.numel())
        torch.cuda.synchronize()
        assert stride == 0 and len(counts) == world and counts[rank] == int(ctl[0].item()), (counts, stride)

        at = 0
        for r in range(world):
            got = framelib.parse_sink(out[at:at + counts[r]].cpu().numpy(), counts[r], FS)
            names = [NAMES[(r + k) % len(NAMES)] for k in range(r + 1)]
            assert len(got) == len(names), (r, sorted(got))
            for i, name in enumerate(names):
                want, _ = T.reference_decode(np.abs(T.load_fixture(name)).astype(np.float32), keep_carrier=True)
                assert got[sorted(got)[i]] == want, (r, name)
            at += counts[r]

        # a rank whose receive buffer is too small: the same verdict on every rank, nobody left waiting
        small = out.numel() if rank else 16
        try:
            g.gather_frames(out.data_ptr(), small)
            raise AssertionError("expected NFCGPU_ENOMEM on every rank")
        except nfclab_amd.NfcGpuError:
            pass

        g.comm_destroy()
        g.sink_hold(False)
        g.sink_attach(None, 0, None)

    dist.barrier()
    print("rank %d ok" % rank)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
