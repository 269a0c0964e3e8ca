"""world_size-2 data-parallel path on CPU (gloo): flat-bucket gradient all-reduce of
cim_amd.nn.DataParallel == mean of the per-rank gradients; no_sync() accumulation; minibatch kwargs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 16)
        self.frozen = nn.Linear(16, 16)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.b = nn.Linear(16, 3)

    def forward(self, data, scale=None, skip_b=False):
        h = torch.relu(self.frozen(torch.relu(self.a(data))))
        y = (h if skip_b else self.b(h)).pow(2).sum()
        return {"losses": {"l": (y * scale).unsqueeze(0)}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5, 6, generator=g)


def _grads(dp):
    """All gradients in reverse parameter order (flat-buffer views and in-place reduced big tensors alike)."""
    params = [p for p in dp.module.parameters() if p.requires_grad]
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in reversed(params)]).clone()


def _worker(rank, world, port, out, big_bytes=16 << 20):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cim_amd.nn import DataParallel
    torch.manual_seed(rank)            # ranks start from DIFFERENT weights: DataParallel must broadcast rank 0's
    dp = DataParallel(Tiny(), cpu_keywords=["im_info"], minibatch=True, bucket_bytes=64, big_bytes=big_bytes)
    assert len(dp.buckets) > 1 and dp.world_size == world
    assert any("tensor" in b for b in dp.buckets) == (big_bytes < (16 << 20))
    # construction broadcast rank 0's parameters (rank 1 was initialised differently on purpose)
    ref = Tiny.__new__(Tiny)
    torch.manual_seed(0)
    Tiny.__init__(ref)
    for a, b in zip(dp.module.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    # step 1: plain synchronised backward - exactly the reference driver's calls (tools/train.py:419-438):
    # zero_grad, forward, backward; the reduction finishes by itself at the end of backward
    dp.zero_grad()
    o = dp(data=[_data(rank)], scale=[torch.tensor(1.0)])
    o["losses"]["l"].sum().backward()
    assert not dp._pending and dp._next_bucket == 0
    g1 = _grads(dp)
    # step 2: accumulate one un-synced micro-step, then a synced one (iter_size = 2)
    dp.zero_grad()
    with dp.no_sync():
        o = dp(data=[_data(rank)], scale=[torch.tensor(0.5)])
        (o["losses"]["l"].sum() * dp.loss_scale()).backward()
    o = dp(data=[_data(rank + 10)], scale=[torch.tensor(2.0)])
    (o["losses"]["l"].sum() * dp.loss_scale()).backward()
    dp.finish_gradient_sync()
    g2 = _grads(dp)
    # the same accumulation driven by iter_size (the unchanged driver loop: no no_sync() in tools/train.py:420-438)
    dp.iter_size = 2
    dp.zero_grad()
    for d, sc in ((_data(rank), 0.5), (_data(rank + 10), 2.0)):
        o = dp(data=[d], scale=[torch.tensor(sc)])
        o["losses"]["l"].sum().backward()
    g2b = _grads(dp)
    dp.iter_size = 1
    # step 3: rank 1 skips one head entirely (its parameters get no gradient on that rank):
    # the strict bucket order must still pair up the collectives of both ranks
    dp.zero_grad()
    o = dp(data=[_data(rank)], scale=[torch.tensor(1.0)], skip_b=[rank == 1])
    (o["losses"]["l"].sum() * dp.loss_scale()).backward()
    dp.finish_gradient_sync()
    g3 = _grads(dp)
    if rank == 0:
        torch.save({"g1": g1, "g2": g2, "g2b": g2b, "g3": g3}, out)
    dist.destroy_process_group()


def _local_grad(data, scale, skip_b=False):
    torch.manual_seed(0)
    m = Tiny()
    m(data, scale, skip_b)["losses"]["l"].sum().backward()
    params = [p for p in m.parameters() if p.requires_grad]
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in reversed(params)])


import pytest


@pytest.mark.parametrize("big_bytes", [16 << 20, 100])       # all parameters in the flat buffer / weight matrices reduced in place
def test_dp_allreduce_world2(tmp_path, big_bytes):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, big_bytes), nprocs=2, join=True)
    got = torch.load(out)
    want1 = (_local_grad(_data(0), 1.0) + _local_grad(_data(1), 1.0)) / 2
    torch.testing.assert_close(got["g1"], want1, rtol=1e-5, atol=1e-6)
    # the un-synced micro-step stays local to rank 0 only until the synced all-reduce sums everything
    want2 = (_local_grad(_data(0), 0.5) + _local_grad(_data(1), 0.5)
             + _local_grad(_data(10), 2.0) + _local_grad(_data(11), 2.0)) / 2
    torch.testing.assert_close(got["g2"], want2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got["g2b"], want2, rtol=1e-5, atol=1e-6)
    want3 = (_local_grad(_data(0), 1.0) + _local_grad(_data(1), 1.0, skip_b=True)) / 2
    torch.testing.assert_close(got["g3"], want3, rtol=1e-5, atol=1e-6)


def test_dp_single_process_passthrough():
    from cim_amd.nn import DataParallel
    torch.manual_seed(0)
    dp = DataParallel(Tiny(), minibatch=True, force_flat_grads=True)
    assert dp.world_size == 1 and hasattr(dp, "module")
    o = dp(data=[_data(0)], scale=[torch.tensor(1.0)])
    o["losses"]["l"].sum().backward()
    torch.testing.assert_close(_grads(dp), _local_grad(_data(0), 1.0))      # (views are 16-byte aligned: the buffer has padding)
    assert all(p.grad.data_ptr() % 16 == 0 for p in dp.module.parameters() if p.requires_grad)
    dp.zero_grad()
    assert float(dp.flat_grad.abs().sum()) == 0 and dp.module.a.weight.grad.data_ptr() >= dp.flat_grad.data_ptr()
    # default single-process mode: no flat buffer, gradients handed over by autograd as they are
    torch.manual_seed(0)
    dp2 = DataParallel(Tiny(), minibatch=True)
    assert dp2.flat_grad is None
    dp2(data=[_data(0)], scale=[torch.tensor(1.0)])["losses"]["l"].sum().backward()
    dp2.finish_gradient_sync()
    got = torch.cat([p.grad.reshape(-1) for p in reversed([q for q in dp2.module.parameters() if q.requires_grad])])
    torch.testing.assert_close(got, _local_grad(_data(0), 1.0))
    dp2.zero_grad()
    assert all(p.grad is None for p in dp2.module.parameters())


# ---- the reference driver's loop, literally (tools/train.py:418-438): optimizer.zero_grad() - torch's default drops the
# gradients (set_to_none=True), i.e. DataParallel's flat views - then iter_size x (forward, loss.backward()), optimizer.step()
def _loop_worker(rank, world, port, out, no_engine_callback):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if no_engine_callback:
        os.environ["CIM_NO_ENGINE_CALLBACK"] = "1"      # public-API fallbacks of cim_amd/utils/engine.py
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cim_amd.nn import DataParallel
    from cim_amd.utils import engine
    assert engine.HAS_ENGINE_CALLBACK == (not no_engine_callback)
    torch.manual_seed(rank)
    results = {}
    for iter_size in (1, 2):
        torch.manual_seed(rank)
        dp = DataParallel(Tiny(), minibatch=True, bucket_bytes=64, big_bytes=100, iter_size=iter_size)
        optimizer = torch.optim.SGD([p for p in dp.parameters() if p.requires_grad], lr=0.1, momentum=0.9)
        for step in range(3):
            optimizer.zero_grad()
            for inner in range(iter_size):
                net_outputs = dp(data=[_data(rank + 10 * inner + 100 * step)], scale=[torch.tensor(1.0 + inner)])
                loss = net_outputs["losses"]["l"].sum()
                loss.backward(retain_graph=True)
            optimizer.step()
        results[iter_size] = torch.cat([p.detach().reshape(-1) for p in dp.module.parameters()]).clone()
        if hasattr(dp, "_step_hook"):
            dp._step_hook.remove()
    if rank == 0:
        torch.save(results, out)
    dist.destroy_process_group()


def _loop_reference(iter_size):
    """The same three optimizer steps in one process on the mean of the two ranks' gradients."""
    torch.manual_seed(0)
    m = Tiny()
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.1, momentum=0.9)
    for step in range(3):
        opt.zero_grad()
        total = [torch.zeros_like(p) for p in params]
        for rank in range(2):
            for inner in range(iter_size):
                out = m(_data(rank + 10 * inner + 100 * step), torch.tensor(1.0 + inner))["losses"]["l"].sum()
                gs = torch.autograd.grad(out, params)
                for t, g in zip(total, gs):
                    t += g / 2
        for p, t in zip(params, total):
            p.grad = t
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


@pytest.mark.parametrize("no_engine_callback", [False, True])
def test_reference_driver_loop_world2(tmp_path, no_engine_callback):
    out = str(tmp_path / "loop.pt")
    mp.spawn(_loop_worker, args=(2, _free_port(), out, no_engine_callback), nprocs=2, join=True)
    got = torch.load(out)
    for iter_size in (1, 2):
        torch.testing.assert_close(got[iter_size], _loop_reference(iter_size), rtol=1e-5, atol=1e-6)


def test_finish_gradient_sync_twice_is_a_noop(tmp_path):
    """a legacy driver calling finish_gradient_sync() after a backward that already reduced must not launch the
    buckets again (a second all-reduce of an averaged buffer is value-neutral but costs a full round - and deadlocks when
    only some ranks do it)."""
    out = str(tmp_path / "g.pt")
    mp.spawn(_twice_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.load(out)["launched"] == 0


def _twice_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cim_amd.nn import DataParallel
    torch.manual_seed(0)
    dp = DataParallel(Tiny(), minibatch=True, bucket_bytes=64)
    dp.zero_grad()
    dp(data=[_data(rank)], scale=[torch.tensor(1.0)])["losses"]["l"].sum().backward()
    dp.comm_works = []
    if rank == 0:                 # only ONE rank calls it: with a re-launch this would hang
        dp.finish_gradient_sync()
    n = len(dp.comm_works)
    dist.barrier()
    if rank == 0:
        torch.save({"launched": n}, out)
    dist.destroy_process_group()
