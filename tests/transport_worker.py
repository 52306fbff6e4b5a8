"""One rank of the record-transport test (started by test_transport.py): a chain of ranks exchanges
record buffers whose sizes jump around, including zero and far beyond the agreed message size.

With --rccl-self a single rank on cuda:0 runs the same sequence over the "nccl" backend (RCCL) with itself
as the peer: the device-buffer branch of the transport (no host staging, header written to and read from
device memory, messages sliced out of the device buffers), which a box with one GPU cannot reach with two
ranks."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fluidsolver_amd import slabs  # noqa: E402

RB = 48
COUNTS = [10, 3000, 5, 0, 2000, 2100, 0, 0, 7000, 1]


def payload(src, dst, step, count):
    """deterministic bytes both ends can compute"""
    rng = np.random.Generator(np.random.PCG64(1000 * step + 10 * src + dst))
    return torch.from_numpy(rng.integers(0, 256, count * RB, dtype=np.uint8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--rccl-self", action="store_true")
    a = ap.parse_args()
    if a.rccl_self:
        return rccl_self(a)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
    tr = slabs.NeighbourTransport(a.rank, a.world, torch.device("cpu"))
    peers = [p for p in (a.rank - 1, a.rank + 1) if 0 <= p < a.world]
    cap = 8000
    send_buf = {p: torch.zeros((cap + 1) * RB, dtype=torch.uint8) for p in peers}
    recv_buf = {p: torch.zeros((cap + 1) * RB, dtype=torch.uint8) for p in peers}
    for step, base in enumerate(COUNTS):
        send = {}
        for p in peers:
            cnt = base if p > a.rank else (base * 2) % 7500  # the two directions differ
            send_buf[p][RB:(1 + cnt) * RB] = payload(a.rank, p, step, cnt)
            send[p] = (send_buf[p], cnt)
        got = tr.exchange(send, recv_buf, RB)
        for p in peers:
            want_cnt = base if a.rank > p else (base * 2) % 7500
            assert got[p] == want_cnt, (step, p, got[p], want_cnt)
            want = payload(p, a.rank, step, want_cnt)
            assert torch.equal(recv_buf[p][RB:(1 + want_cnt) * RB], want), (step, p)
    dist.barrier()
    dist.destroy_process_group()


def rccl_self(a):
    assert a.world == 1 and a.rank == 0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{a.port}", rank=0, world_size=1, device_id=dev)
    tr = slabs.NeighbourTransport(0, 1, dev)
    assert not tr.staged and tr.backend == "nccl"
    cap = 8000
    send_buf = torch.zeros((cap + 1) * RB, dtype=torch.uint8, device=dev)
    recv_buf = {0: torch.zeros((cap + 1) * RB, dtype=torch.uint8, device=dev)}
    side = torch.cuda.Stream(dev)  # not the default stream: the slab solver runs on a stream of its own
    with torch.cuda.stream(side):
        for step, cnt in enumerate(COUNTS):
            want = payload(0, 0, step, cnt)
            send_buf[RB:(1 + cnt) * RB] = want.to(dev)
            got = tr.exchange({0: (send_buf, cnt)}, recv_buf, RB)
            assert got[0] == cnt, (step, got[0], cnt)
            assert torch.equal(recv_buf[0][RB:(1 + cnt) * RB].cpu(), want), step
            recv_buf[0].zero_()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("rccl-self ok")


if __name__ == "__main__":
    main()
