#!/usr/bin/env python3
"""Micro-benchmark of the MaskFuse contractions at BASELINE cfg2 sizes: cim_amd HIP kernels vs the
PyTorch/MIOpen/hipBLASLt calls they replace, plus an accuracy probe of both against fp64."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd.ops import gemm as G  # noqa: E402
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    res = {}
    N, C = 1000, 1024
    g = torch.Generator(device=dev).manual_seed(0)
    # accuracy probe: is the library "fp32" GEMM really fp32?
    A = torch.randn(512, 4096, device=dev, generator=g)
    B = torch.randn(4096, 512, device=dev, generator=g)
    ref = A.double() @ B.double()
    res["relerr_torch_matmul"] = float(((A @ B).double() - ref).abs().max() / ref.abs().max())
    res["relerr_cim_gemm"] = float((G.gemm(A, B, 512, 512, 4096, 4096, 512).double() - ref).abs().max() / ref.abs().max())

    x = torch.randn(N, 7, 7, 2 * C, device=dev, generator=g).permute(0, 3, 1, 2)
    w = torch.randn(C, 2 * C, 3, 3, device=dev, generator=g) * 0.01
    b = torch.zeros(C, device=dev)
    dy = torch.randn(N, 7, 7, C, device=dev, generator=g)
    whwio = w.permute(2, 3, 1, 0).contiguous()
    w2 = w.flip(2, 3).permute(2, 3, 0, 1).contiguous()
    xp = x.permute(0, 2, 3, 1).contiguous()
    y = torch.empty(N, 7, 7, C, device=dev)
    dx = torch.empty(N, 7, 7, 2 * C, device=dev)
    dwh = torch.empty(3, 3, 2 * C, C, device=dev)
    st = _lib.stream_ptr()
    import cim_amd.ops.gemm as GG0
    fl = 2.0 * 49 * N * 18 * C * C
    t = timeit(lambda: _lib.call("cim_conv3x3_f32", xp.data_ptr(), whwio.data_ptr(), b.data_ptr(), y.data_ptr(), N, 7, 2 * C, C, 1, GG0.engine_code(), st))
    res["conv_fwd_cim_ms"], res["conv_fwd_cim_tf"] = t, fl / t / 1e9
    t = timeit(lambda: F.conv2d(x, w, b, padding=1))
    res["conv_fwd_torch_ms"], res["conv_fwd_torch_tf"] = t, fl / t / 1e9
    t = timeit(lambda: _lib.call("cim_conv3x3_f32", dy.data_ptr(), w2.data_ptr(), None, dx.data_ptr(), N, 7, C, 2 * C, 0, GG0.engine_code(), st))
    res["conv_dgrad_cim_ms"], res["conv_dgrad_cim_tf"] = t, fl / t / 1e9
    sp = _lib.call("cim_gemm_f32_splits", 18 * C, C, 49 * N, GG0.engine_code())
    ws = torch.empty(max(sp, 1) * 18 * C * C, device=dev)
    t = timeit(lambda: _lib.call("cim_conv3x3_wgrad_f32", xp.data_ptr(), dy.data_ptr(), dwh.data_ptr(), N, 7, 2 * C, C, sp, ws.data_ptr(), GG0.engine_code(), st))
    res["conv_wgrad_cim_ms"], res["conv_wgrad_cim_tf"], res["conv_wgrad_splits"] = t, fl / t / 1e9, sp
    # Winograd F(2x2,3x3) pipeline (autograd wrapper: fwd, and fwd+bwd)
    from cim_amd.ops import conv3x3
    import cim_amd.ops.gemm as GG
    for algo in ("winograd", "winograd4", "direct"):
        GG.CONV_ALGO = algo
        xr_ = x.detach().clone().requires_grad_(True)
        wr_ = w.detach().clone().requires_grad_(True)
        t = timeit(lambda: conv3x3(xr_, wr_, b, relu=True))
        res["conv_fwd_%s_ms" % algo] = t
        go = torch.randn(N, C, 7, 7, device=dev).contiguous(memory_format=torch.channels_last)
        def fb():
            xr_.grad = None; wr_.grad = None
            conv3x3(xr_, wr_, b, relu=True).backward(go)
        t = timeit(fb)
        res["conv_fwd_bwd_%s_ms" % algo] = t
    # fc1: [N, 49C] x [4096, 49C]^T
    K1 = 49 * C
    xf = torch.randn(N, K1, device=dev, generator=g)
    w1 = torch.randn(4096, K1, device=dev, generator=g) * 0.01
    dyf = torch.randn(N, 4096, device=dev, generator=g)
    fl1 = 2.0 * N * K1 * 4096
    t = timeit(lambda: G.gemm(xf, w1, N, 4096, K1, K1, K1, b_kcontig=True))
    res["fc1_fwd_cim_ms"], res["fc1_fwd_cim_tf"] = t, fl1 / t / 1e9
    t = timeit(lambda: F.linear(xf, w1))
    res["fc1_fwd_torch_ms"], res["fc1_fwd_torch_tf"] = t, fl1 / t / 1e9
    t = timeit(lambda: G.gemm(dyf, w1, N, K1, 4096, 4096, K1))
    res["fc1_dgrad_cim_ms"], res["fc1_dgrad_cim_tf"] = t, fl1 / t / 1e9
    t = timeit(lambda: dyf @ w1)
    res["fc1_dgrad_torch_ms"], res["fc1_dgrad_torch_tf"] = t, fl1 / t / 1e9
    t = timeit(lambda: G.gemm(dyf, xf, 4096, K1, N, 4096, K1, a_mcontig=True))
    res["fc1_wgrad_cim_ms"], res["fc1_wgrad_cim_tf"] = t, fl1 / t / 1e9
    t = timeit(lambda: dyf.t() @ xf)
    res["fc1_wgrad_torch_ms"], res["fc1_wgrad_torch_tf"] = t, fl1 / t / 1e9
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) and v > 1e-3 else v) for k, v in res.items()}, indent=1))


if __name__ == "__main__":
    main()
