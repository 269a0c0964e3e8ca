cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pool.py -x -q 2>&1 | tail -3
python - <<'P'
import torch, sys
sys.path.insert(0,'.')
from cim_amd.ops import max_pool2d
import torch.nn as nn
x=torch.randn(1,64,258,344,device='cuda')
m=nn.MaxPool2d(3,2,1)
def t(f,n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
print("hip us", t(lambda: max_pool2d(x,m)), "aten us", t(lambda: m(x)))
P
