"""-m gpu: how far each arithmetic configuration of the MaskFuse contractions is from the reference, measured.

north_star: "within stated fp tolerance on losses/logits".  The build's stated tolerance (README):
relative deviation of each of the four losses <= 1e-5, of every parameter gradient (||g - g_ref|| / ||g_ref||) <= 6e-3,
against the fp32 CPU reference.  This test RECORDS the deviation for every engine x algorithm combination (the product's pair engine
and the superseded ones of experiments/)
 - against the reference's own whole-step golden (cfg1, tests/golden/e2e_vgg16_voc.npz: the REFERENCE code ran it) and
 - against the fp32/direct run of the same step at cfg2 full size (1000 proposals, 516 x 688 image),
prints the table (profiles/r2/parity_deviation.json is a copy of one run) and asserts 3x-margin bounds per engine."""
import json
import os

import numpy as np
import pytest
import torch

from cases import E2E, GRAD_RTOL, e2e_inputs, gradient_deviation, procedural_init, record_deviation

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
COMBOS = [("fp32", "direct"), ("fp32", "winograd7"), ("bf16x3", "direct"), ("bf16x3", "winograd7"),
          ("f16x2", "direct"), ("f16x2", "winograd7"), ("f16x2p", "winograd7")]
# bounds = ~3x the values measured on MI355X (see the printed table): (loss rel., gradient rel. vs reference / vs fp32-direct)
LOSS_TOL = 1e-5
GRAD_TOL = GRAD_RTOL          # 5e-4 per parameter; gradients that cancel to nothing are bounded absolutely (cases.py)


def _set(monkeypatch, engine, algo):
    """f16x2p + winograd7 = the product as it is.  Every other combination routes MaskFuse.forward through the superseded engines of
    experiments/ (test infrastructure: the product has one engine and one algorithm)."""
    from cim_amd.modeling import maskfuse
    from cim_amd.ops import gemm
    from experiments import engines
    gemm.forget_weight_scales()
    monkeypatch.undo()
    if engine == "f16x2p":
        assert algo == "winograd7"
        return
    monkeypatch.setattr(engines, "ENGINE", engine)
    monkeypatch.setattr(engines, "CONV_ALGO", algo)
    monkeypatch.setattr(maskfuse.MaskFuse, "forward", lambda self, x, rois, masks: engines.maskfuse_forward(self, x, rois, masks))


def _step(model, batch, seed):
    model.zero_grad(set_to_none=True)
    np.random.seed(seed)
    out = model(**batch)
    sum(v.sum() for v in out["losses"].values()).backward()
    torch.cuda.synchronize()
    losses = {k: float(v.detach()) for k, v in out["losses"].items()}
    grads = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return losses, grads


def _grad_dev(g, ref):
    worst, where, _ = gradient_deviation(g, ref, check=False)
    return worst, where


def test_engine_and_algorithm_deviation(monkeypatch, golden_dir):
    from cim_amd import _lib, mask_iou, synthetic
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    _lib.load()
    table = {}
    # ---- cfg1 against the reference's own run
    g = np.load(os.path.join(golden_dir, "e2e_vgg16_voc.npz"))
    apply_preset(E2E["config"])
    m = Generalized_RCNN().train()
    procedural_init(m)
    m = m.to(DEV)
    inp = e2e_inputs()
    t = lambda a: torch.from_numpy(a).unsqueeze(0).to(DEV)
    batch = dict(data=torch.from_numpy(inp["data"]).to(DEV), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                 gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]), iou_map=torch.from_numpy(inp["iou"]).to(DEV),
                 asy_iou_map=torch.from_numpy(inp["asy"]).to(DEV))
    names = [str(n) for n in g["grad_names"]]
    for engine, algo in COMBOS:
        _set(monkeypatch, engine, algo)
        losses, grads = _step(m, batch, E2E["np_seed"])
        ldev = max(abs(losses[k] - float(g["loss_" + k])) / abs(float(g["loss_" + k])) for k in losses)
        # (gradients that are exactly zero in exact arithmetic - the detector bias under the softmax over proposals - are
        # pure rounding noise of ~1e-9 in both implementations: skipped)
        devs = [(abs(float(grads[n].norm()) - norm) / norm, n) for n, norm in zip(names, g["grad_norms"]) if norm > 1e-6]
        ndev, where = max(devs)
        table["cfg1 vs reference | %s + %s" % (engine, algo)] = dict(loss_rel=ldev, grad_norm_rel=ndev, worst_param=where)
    del m, batch
    # ---- cfg2 at full size against the fp32 / direct run
    apply_preset("resnet50_voc")
    torch.manual_seed(3)
    m = Generalized_RCNN()
    for mod in m.modules():
        if hasattr(mod, "bn3"):
            torch.nn.init.constant_(mod.bn3.weight, 0.25)
    import copy
    cpu_m = copy.deepcopy(m).train()                     # the same weights for the fp32 CPU oracle step (oracle/cpu_step.py)
    m = m.to(DEV).train()
    inp = synthetic.make_image_inputs("resnet50_voc", seed=3)
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(DEV))
    from oracle import cpu_step
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cpu_losses = cpu_step.step(cpu_m, inp, iou.cpu().numpy(), asy.cpu().numpy(), seed=77)
    cpu_g = {n: p.grad.detach().double() for n, p in cpu_m.named_parameters() if p.grad is not None}
    batch = dict(data=torch.from_numpy(inp["data"]).to(DEV), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                 gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy)
    ref_l = ref_g = None
    for engine, algo in COMBOS:
        _set(monkeypatch, engine, algo)
        losses, grads = _step(m, batch, 77)
        # ... and against the fp32 CPU oracle step (ATen CPU kernels + oracle ROIAlign / mining): which parameter deviates most
        cl = max(abs(losses[k] - cpu_losses[k]) / abs(cpu_losses[k]) for k in cpu_losses)
        cg, cwhere = _grad_dev({n: grads[n] for n in cpu_g if n in grads}, {n: cpu_g[n] for n in cpu_g if n in grads})
        table["cfg2 vs CPU oracle | %s + %s" % (engine, algo)] = dict(loss_rel=cl, grad_rel=cg, worst_param=cwhere)
        if ref_l is None:
            ref_l, ref_g = losses, grads
            l2, g2 = _step(m, batch, 77)                          # run-to-run noise floor of the reference configuration itself
            table["cfg2 run-to-run | fp32 + direct"] = dict(loss_rel=max(abs(l2[k] - ref_l[k]) / abs(ref_l[k]) for k in ref_l),
                                                            grad_rel=_grad_dev(g2, ref_g)[0])
            continue
        ldev = max(abs(losses[k] - ref_l[k]) / abs(ref_l[k]) for k in ref_l)
        gdev, where = _grad_dev(grads, ref_g)
        table["cfg2 vs fp32+direct | %s + %s" % (engine, algo)] = dict(loss_rel=ldev, grad_rel=gdev, worst_param=where)
    del m, cpu_m
    # ---- cfg2 at full size with weights under which the LOSSES move with the contraction: with the random initialisation above
    # the heads' logits are ~1e-2 wide, the scores flat, and the four losses bit-equal in fp32 whatever the engine (loss_rel
    # exactly 0.0 in rounds 2-3: a vacuous check).  Here the scoring heads' weights are scaled up until the classifier's logits
    # have unit spread over the proposals: the scores then depend visibly on seg_x, the mining still finds seeds, and the engines
    # must agree to LOSS_TOL on losses that DO move.
    torch.manual_seed(3)
    m = Generalized_RCNN()
    for mod in m.modules():
        if hasattr(mod, "bn3"):
            torch.nn.init.constant_(mod.bn3.weight, 0.25)
    m = m.to(DEV).train()
    _set(monkeypatch, "fp32", "direct")
    got = {}
    hook = m.Box_Head.register_forward_hook(lambda mod, i, o: got.update(seg_x=o.detach()))
    with torch.no_grad():
        np.random.seed(77)
        m(**batch)
        hook.remove()
        spread = float(torch.nn.functional.linear(got["seg_x"], m.cls_iou_model.classifier.weight).std())
        for p in m.cls_iou_model.parameters():
            p.mul_(1.0 / spread)
    from cim_amd.modeling import heads as _heads
    _heads.settle_rng()
    ref_l = ref_g = None
    moved = 0.0
    for engine, algo in [("fp32", "direct"), ("f16x2", "winograd7"), ("f16x2p", "winograd7")]:
        _set(monkeypatch, engine, algo)
        losses, grads = _step(m, batch, 77)
        assert all(np.isfinite(v) for v in losses.values()), losses
        if ref_l is None:
            ref_l, ref_g = losses, grads
            print("cfg2 scaled heads: losses", losses)
            assert sum(1 for v in losses.values() if abs(v) > 1e-6) >= 3, "degenerate case (losses %s)" % losses
            continue
        ldev = max(abs(losses[k] - ref_l[k]) / abs(ref_l[k]) for k in ref_l if abs(ref_l[k]) > 1e-6)
        gdev, where = _grad_dev(grads, ref_g)
        moved = max(moved, ldev)
        table["cfg2 scaled heads vs fp32+direct | %s + %s" % (engine, algo)] = dict(loss_rel=ldev, grad_rel=gdev, worst_param=where)
    assert moved > 0.0, "the scaled-heads case is vacuous too: no loss moved between engines"
    print("\nPARITY-DEVIATION " + json.dumps(table))
    for k, v in table.items():
        record_deviation(k, v)
    for k, v in table.items():
        assert v["loss_rel"] <= LOSS_TOL, (k, v)
        # (scaled heads: the gradients reaching the body are ~30x larger relative to its activations; measured 5.3e-4 on a
        # BatchNorm scale gradient of res4 - a sum of ~10^6 products - between the fp16-split Winograd engine and fp32 + direct)
        assert v.get("grad_rel", v.get("grad_norm_rel")) <= (2 * GRAD_TOL if k.startswith("cfg2 scaled heads") else GRAD_TOL), (k, v)


def test_tf32_class_single_product_deviation(monkeypatch, golden_dir):
    """SURVEY section 7 (hard parts): "decide explicitly and report both".  The reference's own arithmetic on its hardware is TF32-class
    (torch 1.10 defaults; tools/train.py:153-154 sets cudnn.deterministic / benchmark only); the build's step is fp32-class (three fp16
    MFMA products per multiply-add).  ops.pair.PRODUCTS = 1 evaluates the h * h product alone - 11-bit operands, fp32 accumulation -
    on the same pair images.  RECORDED here (gpurun_out/parity_deviation.json): how far that run is from the reference's cfg1 golden
    and from the three-product run of the same cfg2 step, and whether the mined pseudo labels are still the same."""
    from cim_amd import _lib, mask_iou, synthetic
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from cim_amd.ops import pair
    _lib.load()
    table = {}
    t = lambda a: torch.from_numpy(a).unsqueeze(0).to(DEV)
    _set(monkeypatch, "f16x2p", "winograd7")
    # ---- cfg1 against the reference's own run
    g = np.load(os.path.join(golden_dir, "e2e_vgg16_voc.npz"))
    apply_preset(E2E["config"])
    m = Generalized_RCNN().train()
    procedural_init(m)
    m = m.to(DEV)
    inp = e2e_inputs()
    batch = dict(data=torch.from_numpy(inp["data"]).to(DEV), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                 gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]), iou_map=torch.from_numpy(inp["iou"]).to(DEV),
                 asy_iou_map=torch.from_numpy(inp["asy"]).to(DEV))
    names = [str(n) for n in g["grad_names"]]
    for products in (3, 1):
        monkeypatch.setattr(pair, "PRODUCTS", products)
        losses, grads = _step(m, batch, E2E["np_seed"])
        ldev = max(abs(losses[k] - float(g["loss_" + k])) / abs(float(g["loss_" + k])) for k in losses)
        ndev, where = max((abs(float(grads[n].norm()) - norm) / norm, n) for n, norm in zip(names, g["grad_norms"]) if norm > 1e-6)
        table["cfg1 vs reference | pair engine, %d product%s" % (products, "s" if products > 1 else "")] = \
            dict(loss_rel=ldev, grad_norm_rel=ndev, worst_param=where)
    del m, batch
    # ---- cfg2 at full size: one product against three
    apply_preset("resnet50_voc")
    torch.manual_seed(3)
    m = Generalized_RCNN()
    for mod in m.modules():
        if hasattr(mod, "bn3"):
            torch.nn.init.constant_(mod.bn3.weight, 0.25)
    m = m.to(DEV).train()
    inp = synthetic.make_image_inputs("resnet50_voc", seed=3)
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(DEV))
    batch = dict(data=torch.from_numpy(inp["data"]).to(DEV), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                 gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy)
    runs = {}
    for products in (3, 1):
        monkeypatch.setattr(pair, "PRODUCTS", products)
        losses, grads = _step(m, batch, 77)
        mined = m.__dict__["_last_mining"]
        runs[products] = (losses, grads, [tuple(x.clone() for x in p) for p in mined.pseudo], mined.valid.clone())
    l3, g3, p3, v3 = runs[3]
    l1, g1, p1, v1 = runs[1]
    same = bool(torch.equal(v3, v1)) and all(torch.equal(a, b) for pa, pb in zip(p3, p1) for a, b in zip(pa, pb))
    gdev, where = _grad_dev(g1, g3)
    table["cfg2 one product vs three | pair engine"] = dict(loss_rel=max(abs(l1[k] - l3[k]) / abs(l3[k]) for k in l3), grad_rel=gdev,
                                                            worst_param=where, pseudo_labels_identical=same)
    print("\nPARITY-DEVIATION " + json.dumps(table))
    for k, v in table.items():
        record_deviation(k, v)
    # TF32-class arithmetic: operands rounded to 11 bits (2^-12 relative each) - three decimal orders above the fp32-class engine.
    # Measured (round 5): cfg1 losses 1.4e-3 / gradient norms 1.3e-3 from the reference (fp32-class: 2.7e-7 / 2.4e-5); at cfg2 the
    # losses move by 8e-4 and the MINED PSEUDO LABELS CHANGE (scores within 1e-3 of each other swap ranks), after which the
    # refinement heads' gradients differ by 0.19 - no gradient bound can be stated for that configuration, which is why the
    # step's default is the three-product evaluation ("mining indices bit-identical to reference" needs fp32-class scores).
    assert table["cfg1 vs reference | pair engine, 3 products"]["loss_rel"] <= LOSS_TOL
    for k, v in table.items():
        assert v["loss_rel"] <= 5e-3, (k, v)
        if k.startswith("cfg1"):
            assert v["grad_norm_rel"] <= 5e-2, (k, v)
