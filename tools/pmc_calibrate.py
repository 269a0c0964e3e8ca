#!/usr/bin/env python3
"""Known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md asks for it):
run this under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and feed the two
output directories to tools/pmc_traffic.py --calibrate.  Each launch moves an exactly known number of bytes with one of
the access widths the product kernels use:

    cal_copy16   libcim_hip's split-K reduce of the region-form ROIAlign backward used as a copy (groups = 1):
                 16 B per lane nontemporal load + 16 B per lane store, 1 GiB in / 1 GiB out
    cal_copy4    ATen elementwise copy of a strided (non-vectorisable) view: 4 B per lane, 256 MiB in / out
    cal_clone16  ATen vectorised copy (float4), 1 GiB in / 1 GiB out
Buffers are larger than the 256 MiB Infinity Cache and each is touched once per launch."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
n = (1 << 30) // 4
src = torch.randn(n, device=dev)
dst = torch.empty_like(src)
lib = _lib.load()
torch.cuda.synchronize()
for _ in range(3):
    torch.add(src, 1.0, out=dst)                                # vectorized_elementwise_kernel (16 B per lane): 1 GiB in, 1 GiB out
    torch.cuda.synchronize()
    a = src.view(-1, 2)[: n // 8, 0]                             # stride-2 view: 4 B per lane, 256 MiB of lines touched... read
    b = dst.view(-1, 2)[: n // 8, 0]
    torch.add(a, 1.0, out=b)                                     # elementwise_kernel (scalar, strided): 4 B per lane
    torch.cuda.synchronize()
print("calibration launches done: 1 GiB vector copy x3, strided 4-byte copy x3")
