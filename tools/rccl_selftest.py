#!/usr/bin/env python3
"""The torch.distributed calls of bench.py and collector.py against RCCL itself, in a group of ONE rank (the box this
round has one GPU; two ranks cannot share a device under RCCL): process-group init with device_id, barrier, the float64
MAX / SUM all-reduces of the bench line, and gather_records' counts + padded all_gather on device tensors.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
        tools/rccl_selftest.py
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from alphazero_amd.collector import gather_records
    dist.barrier()
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    w = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    assert t.item() == 1.5 and w.tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    recs = torch.arange(5 * 48, device="cuda").remainder(251).to(torch.uint8).reshape(5, 48)
    out = gather_records(recs, force=True)
    assert torch.equal(out, recs)
    out = gather_records(recs[:0], force=True)
    assert out.shape == (0, 48)
    dist.barrier()
    torch.cuda.synchronize()
    print("rccl selftest ok: backend", dist.get_backend(), "world", dist.get_world_size())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
