"""Chaining the backward of two fused conv + frozen-statistics-BatchNorm + ReLU layers without a BatchNorm-backward launch in between.

When layer P produces x = relu(bn_P(conv_P(.))) (no residual) and layer Q is its ONLY consumer, the data-gradient product of Q
applies P's BatchNorm + ReLU backward in its epilogue (csrc/conv1x1.hip: SmallArgs.mask):
    dz = x > 0 ? dx : 0,   dconv_P = dz * gamma_P rsqrt(var_P + eps_P)
so P's backward starts from the gradient of its convolution output and skips its `bn_act_bwd` launch (torchvision Bottleneck
conv1 -> conv2 -> conv3 of /root/reference/lib/modeling/resnet50.py:17-44: 2 of 3 such launches per block).
The reference freezes the STATISTICS of its BatchNorm layers only (resnet50.py:59-60: the freeze of the affine parameters is
commented out), so gamma_P / beta_P train: their gradients  dbeta = sum dz,  dgamma = rsqrt(var + eps) sum dz (conv_P - mean)  are
per-channel sums over the pixels.  Q's epilogue writes them as partial sums per 32-pixel group (SmallArgs.mpart: plain stores, no
atomics); they are finished in group order by ONE launch for all chained layers at the end of the backward pass (ops/conv1x1.py:
finish_affine through ops/gemm.py: defer_finisher) - deterministic, and the same gradients as the separate launch up to the
summation order.

Protocol (host side, per backward pass; everything hangs off P's own `state` dict - no process-global marks):
  * P's forward leaves its convolution output in state["xr"] (it lives as long as P's autograd node: a retained graph can run its
    backward again); P's wrapper tags its output: y._cim_bn = InputBn(..., state);
  * Q is called with fuse_input_bn=True by code that KNOWS x has no other consumer (the bottleneck): it marks state["taken"],
    applies the epilogue in its backward and hands the result over: hand_over(in_bn, dx, part) -> state["handed"];
  * P's backward asks take(dy, state): (True, part) -> dy is already dconv_P.  A taken P that receives anything else (the gradient
    was accumulated with another one: x had a second consumer after all) raises - it cannot be repaired silently.  A mark that is
    never taken (P's backward did not run) dies with P's state.
  * Someone who looks AT x's gradient would see dconv_P instead (the chain rewrites what flows along that edge).  So Q checks, at
    its backward, that x is still private (still_private): no tensor hook, no retain_grad() on x, and the running pass is a complete
    .backward() - not torch.autograd.grad(...) / .backward(inputs=...), which stop at (or capture) tensors inside the graph:
    restricted_pass() asks the engine whether Q's own parameters are accumulated into by this pass.  If not, Q computes the plain
    data gradient and P its own BatchNorm backward - the ordinary autograd path, correct for every observer (the decision is per pass:
    state["unchained_pass"], consumed by P's take(); the forward-time mark state["taken"] is never cleared).  What cannot be seen
    from here - torch.autograd.grad whose inputs hold x AND every parameter of Q - is documented in INTEGRATION.md section 4.
"""
import collections

import torch

InputBn = collections.namedtuple("InputBn", "gamma var eps mean affine state")


def tag(y, gamma, beta, mean, var, eps, relu, has_res, state):
    """Tag P's output `y` for a chaining consumer; -> state (None when P is not eligible).  A BatchNorm whose mean carries a
    folded convolution bias that trains (HRNet) is not eligible: its gradient is derived from dbeta on the caller's stream."""
    if state is None or not relu or has_res or mean.requires_grad or not y.requires_grad:
        return None
    y._cim_bn = InputBn(gamma, var, float(eps), mean, bool(gamma.requires_grad or beta.requires_grad), state)
    return state


def input_bn(x, enabled):
    """The InputBn of the layer that produced x, marked as taken - or None."""
    t = getattr(x, "_cim_bn", None) if enabled else None
    if t is None:
        return None
    t.state["taken"] = True
    return t


def node_runs(fn):
    """Will the autograd node `fn` be executed by the backward pass that is running now?  (torch.autograd.grad(outputs, inputs) only
    runs the nodes between them.)  True when the engine cannot be asked."""
    if fn is None:
        return False
    try:
        return bool(torch._C._will_engine_execute_node(fn))
    except (AttributeError, RuntimeError):
        return True


def restricted_pass(node, indices=(1, 3, 4)):
    """Is the running backward pass known NOT to be a complete .backward()?  torch.autograd.grad(outputs, inputs) and
    .backward(inputs=...) only run the nodes between their ends and CAPTURE gradients at tensors inside the graph - there the
    bypasses of this package (deferred weight gradients, chained BatchNorm backward, branch hand-over) must step aside.  The engine
    tells whether a node runs (or is captured); a complete pass accumulates into every trainable leaf, so a parameter of `node`
    (the Function's ctx; `indices`: its parameter inputs) whose AccumulateGrad node neither runs nor is captured gives a
    restricted pass away.  (A captured leaf reads as "runs": all parameters are looked at, not the first.)"""
    try:
        nf = node.next_functions
    except AttributeError:
        return False
    for i in indices:
        fn = nf[i][0] if i < len(nf) else None
        if fn is not None and type(fn).__name__ == "AccumulateGrad" and not node_runs(fn):
            return True
    return False


def still_private(x, in_bn, node=None):
    """Q's backward: may x's incoming gradient still be rewritten into the gradient of P's convolution output?  Not when somebody
    observes x's gradient (hook, retain_grad) or the pass is not a complete one - then the chain is undone for this pass."""
    ok = (not getattr(x, "retains_grad", False) and not getattr(x, "_backward_hooks", None)
          and not getattr(x, "_post_accumulate_grad_hooks", None) and node_runs(x.grad_fn)
          and not (node is not None and restricted_pass(node)))
    if not ok:
        # P sees an ordinary gradient in THIS pass and runs its own BatchNorm backward.  The forward-time mark state["taken"] stays:
        # a later complete pass over a retained graph chains again, and take()'s second-consumer check must still be armed for it.
        in_bn.state["unchained_pass"] = True
    return ok


def c_args(in_bn, part):
    """(in_gamma, in_var, in_eps, in_xr, in_mean, in_part) of the C entry points."""
    if in_bn is None:
        return None, None, 0.0, None, None, None
    if part is None:
        return in_bn.gamma.data_ptr(), in_bn.var.data_ptr(), in_bn.eps, None, None, None
    return (in_bn.gamma.data_ptr(), in_bn.var.data_ptr(), in_bn.eps, in_bn.state["xr"].data_ptr(), in_bn.mean.data_ptr(),
            part.data_ptr())


def hand_over(in_bn, dx, part=None):
    in_bn.state.pop("unchained_pass", None)      # (a mark left by an earlier restricted pass in which P's backward never ran)
    in_bn.state["handed"] = (dx.data_ptr(), dx.numel(), part)


def take(dy, state):
    """-> (dy is already the gradient of this layer's convolution output, partial sums of its affine gradients or None)"""
    if state is None:
        return False, None
    h = state.pop("handed", None)
    if h is not None and h[0] == dy.data_ptr() and h[1] == dy.numel():
        return True, h[2]
    if state.pop("unchained_pass", False):       # Q stepped aside in this pass (still_private): an ordinary gradient
        return False, None
    if state.get("taken"):
        raise RuntimeError("cim_amd: a layer whose BatchNorm backward was fused into its consumer's data gradient received a "
                           "different gradient tensor - its output has a second consumer (fuse_input_bn=True is only valid for "
                           "a single consumer)")
    return False, None
