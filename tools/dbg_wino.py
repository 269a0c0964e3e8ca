import sys, torch
sys.path.insert(0, '.')
from cim_amd import _lib
from cim_amd.ops import gemm as G
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(9)
R, P, C = 13, 7, 72
x = torch.randn(R, P, P, C, generator=g).to(dev)
mt, npos, st = R * 4, 36, _lib.stream_ptr()
V = torch.zeros(npos, mt, C, device=dev)
vr = torch.zeros(npos * mt, dtype=torch.int32, device=dev)
_lib.call("cim_wino_input_transform_amax", x.data_ptr(), V.data_ptr(), vr.data_ptr(), R, P, C, 4, st)
V0 = torch.zeros_like(V)
_lib.call("cim_wino_input_transform", x.data_ptr(), V0.data_ptr(), R, P, C, 4, st)
torch.cuda.synchronize()
d = (V - V0).abs()
print("maxdiff", float(d.max()), "nmismatch", int((V != V0).sum()), "of", V.numel(), "V0 absmax", float(V0.abs().max()))
bad = (V != V0).nonzero()
print(bad[:10])
print("by pos:", (V != V0).sum(dim=(1, 2)).tolist())
ra = vr.view(torch.float32).view(npos, mt)
print("row amax match vs V:", bool(torch.equal(ra, V.abs().amax(dim=2))), "vs V0:", bool(torch.equal(ra, V0.abs().amax(dim=2))))
