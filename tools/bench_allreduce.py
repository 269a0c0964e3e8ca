#!/usr/bin/env python3
"""All-reduce micro-benchmark of EXACTLY the collective sequence a training step issues (cim_amd.nn.DataParallel's
buckets for the given config, in issue order), on N GPUs of one node over RCCL / xGMI:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_allreduce.py [--config resnet50_voc]

Prints per bucket: bytes, ms, algorithm bandwidth (bytes / time) and bus bandwidth (x 2 (N-1) / N, the per-link figure to
hold against the ~153 GB/s of one xGMI link for a ring; a direct reduce-scatter + all-gather over all 7 links is bounded
by 2 S / N per link instead), and the whole sequence back to back - the number to hold against the ~9 ms of backward
that follow the fc1 weight gradient in a 17 ms step.  `--rs-ag` times reduce_scatter + all_gather
for the big buckets instead of all_reduce."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="resnet50_voc")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rs-ag", action="store_true")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from cim_amd.nn import DataParallel
    apply_preset(args.config)
    torch.manual_seed(3)
    model = Generalized_RCNN().to(dev)
    dp = DataParallel(model, minibatch=True)
    sizes = []
    for bk in dp.buckets:
        sizes.append(bk["tensor"].numel() if "tensor" in bk else bk["end"] - bk["start"])
    del dp, model
    torch.cuda.empty_cache()
    bufs = [torch.randn(n, device=dev) for n in sizes]
    op = dist.ReduceOp.AVG

    def reduce(b):
        if args.rs_ag and b.numel() * 4 >= (16 << 20) and b.numel() % world == 0:
            out = torch.empty(b.numel() // world, device=dev)
            dist.reduce_scatter_tensor(out, b, op=op)
            dist.all_gather_into_tensor(b, out)
        else:
            dist.all_reduce(b, op=op)

    def timed(fn):
        fn()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        t = torch.tensor([a.elapsed_time(e) / args.iters], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    rows = []
    for i, b in enumerate(bufs):
        ms = timed(lambda b=b: reduce(b))
        nb = b.numel() * 4
        rows.append(dict(bucket=i, bytes=nb, ms=ms, algbw_GBs=nb / ms / 1e6, busbw_GBs=nb / ms / 1e6 * 2 * (world - 1) / world))
    seq = timed(lambda: [reduce(b) for b in bufs])
    if rank == 0:
        total = sum(r["bytes"] for r in rows)
        print(json.dumps(dict(config=args.config, n_gpus=world, mode="reduce_scatter+all_gather" if args.rs_ag else "all_reduce",
                              total_bytes=total, sequence_ms=seq, sequence_algbw_GBs=total / seq / 1e6,
                              ring_bound_ms=1e3 * 2.0 * (world - 1) / world * total / 153e9,
                              direct_bound_ms=1e3 * 2.0 * total / world / 153e9, buckets=rows)))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
