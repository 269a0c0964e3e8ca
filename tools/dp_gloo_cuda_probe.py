import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from torch import nn
from cim_amd.nn import DataParallel
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = nn.Sequential(nn.Linear(256, 512), nn.ReLU(), nn.Linear(512, 512), nn.ReLU(), nn.Linear(512, 8)).to(dev)
dp = DataParallel(m, minibatch=True, bucket_bytes=1 << 18)
print(rank, "buckets", len(dp.buckets), flush=True)
x = torch.randn(32, 256, device=dev) + rank
for it in range(3):
    t0 = time.time()
    dp.zero_grad()
    y = dp(x)
    (y.pow(2).mean() * dp.loss_scale()).backward()
    print(rank, "bwd done, pending", len(dp._pending), flush=True)
    dp.finish_gradient_sync()
    torch.cuda.synchronize()
    print(rank, "iter", it, "ok", round(time.time() - t0, 3), float(dp.flat_grad.abs().sum()), flush=True)
dist.barrier()
tt = torch.tensor([1.0 + rank], device=dev, dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
print(rank, "max", float(tt), flush=True)
dist.destroy_process_group()
