"""The mining calls alone (cim_asy_prep + cim_mining_step through heads.mine_step) on a BASELINE-shaped image: HIP-event time of the
calls on the step's stream, checked against oracle/mining.py first; with CIM_HIP_LIB pointing at a -DCIM_MINING_CLOCKS=1 build
(bash tools/build_alt.sh clk mining.hip -DCIM_MINING_CLOCKS=1) also the in-kernel phase stamps.

    python tools/bench_mining.py [--config resnet50_voc] [--n 1000] [--iters 200]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cim_amd import _lib, mask_iou, synthetic          # noqa: E402
from cim_amd.modeling import heads                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="resnet50_voc")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfgd = synthetic.CONFIGS[args.config]
    n = args.n or cfgd["n"]
    C = cfgd["classes"]
    inp = synthetic.make_image_inputs(args.config, seed=3, n=n)
    rng = np.random.RandomState(7)
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    labels = torch.from_numpy(inp["labels"]).to(dev)
    thr = [(0.25, 0.5), (0.35, 0.6), (0.45, 0.7)]
    layers = [heads.CIM_layer(0.1, c, i, 0.85, True) for c, i in thr]
    sc_np = [synthetic.make_scores(n, C, rng) for _ in thr]
    # the heads' fused score matrix layout: column blocks read in place through their row stride
    fused = torch.zeros((n, 8 * (C + 1)), dtype=torch.float32, device=dev)
    scores = []
    for i, (cls, det, iouscore) in enumerate(sc_np):
        a = fused[:, (2 * i) * (C + 1):(2 * i + 1) * (C + 1)]
        b = fused[:, (2 * i + 1) * (C + 1):(2 * i + 2) * (C + 1)]
        a.copy_(torch.from_numpy(cls))
        b.copy_(torch.from_numpy(det if i == 0 else iouscore))
        scores.append((a, b))
    if not args.no_check:
        from oracle import mining as om
        np.random.seed(11)
        ref = [om.cim_layer_forward(s[0].cpu().numpy(), s[1].cpu().numpy(), inp["labels"], iou.cpu().numpy(), asy.cpu().numpy(),
                                    cls_thr=c, iou_thr=i) for s, (c, i) in zip(scores, thr)]
        np.random.seed(11)
        res = heads.mine_step(layers, scores, labels, iou, asy).commit()
        heads.settle_rng()
        for l, r in enumerate(ref):
            G = int(res.host[2 + 2 * l])
            assert (r[0] is None) == (G == 0), l
            if r[0] is not None:
                assert np.array_equal(res.pseudo[l][0].cpu().numpy(), r[0]) and np.array_equal(res.pseudo[l][2].cpu().numpy(), r[2]), l
        print("# parity with oracle/mining.py: ok; pseudo GTs per layer (before / after sampling):",
              [(int(res.host[2 + 2 * l]), int(res.host[3 + 2 * l])) for l in range(3)],
              "seeds per class:", [res.debug[l]["n_seeds"].cpu().numpy()[np.nonzero(inp["labels"].reshape(-1))[0]].tolist() for l in range(3)])
    heads.LAZY_SETTLE = True
    out = {}
    for name, ahead in (("prep_inline", False), ("prep_ahead", True)):
        ts = []
        for it in range(args.iters + 20):
            np.random.seed(5)
            prep = heads.prepare_containment(layers, asy) if ahead else None
            if ahead:
                torch.cuda.current_stream().wait_event(prep.event)      # (in the step the backbone forward lies in between)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(3000000)      # ~1.4 ms of GPU spin in front: the host enqueues the calls meanwhile (as in the step, where it
            a.record()                      # runs ahead of the GPU) - without it the span holds the host's ~0.25 ms of launch work
            res = heads.mine_step(layers, scores, labels, iou, asy, prep=prep)
            b.record()
            res.commit()
            heads.settle_rng()
            torch.cuda.synchronize()
            if it >= 20:
                ts.append(a.elapsed_time(b))
        out[name + "_ms"] = float(np.median(ts))
        out[name + "_ms_min"] = float(np.min(ts))
    out.update(config=args.config, n=n, classes=int(inp["labels"].sum()))
    print(json.dumps(out))
    lib_path = os.environ.get("CIM_HIP_LIB")
    if lib_path:
        lib = ctypes.CDLL(lib_path)
        if hasattr(lib, "cim_debug_mining_clocks"):
            buf = (ctypes.c_ulonglong * (2 * 8 * 16))()
            assert lib.cim_debug_mining_clocks(buf) == 0
            a = np.array(buf, dtype=np.uint64).reshape(2, 8, 16).astype(np.int64)
            t0 = a[a > 0].min()
            for k, nst in ((0, 6), (1, 9)):
                for wg in range(8):
                    st = a[k, wg, :nst]
                    if st[0] == 0:
                        continue
                    st = st[st > 0]
                    print("phase", "seed" if k == 0 else "arb ", "wg", wg, "start %6.2f us" % ((st[0] - t0) / 100.0),
                          "deltas us:", [round(float(d) / 100.0, 2) for d in np.diff(st)], "total %.2f" % ((st[-1] - st[0]) / 100.0))


if __name__ == "__main__":
    main()
