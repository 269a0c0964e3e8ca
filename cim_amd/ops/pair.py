"""Pair images: fp32 matrices stored pre-split for the f16x2p GEMM engine (cim_amd/csrc/gemm_pair.hip).

A pair image of a logical matrix [rows][cols] is an int32 tensor [batch][rows_pad][ld] (4 bytes per logical element: the
bytes hold [col / 8][h: 8 x f16 | l: 8 x f16]) plus ONE power-of-two scale per batch entry with x * scale = h + l.  Rows
>= rows are zero (so the image can be contracted over its rows in slabs of 32).  The contractions of MaskFuse
(/root/reference/lib/modeling/resnet50.py:104-110,135-136) consume such images directly; the kernels that produce the
tensors write them (winograd.hip transforms, flatten, SGD), `split()` is the generic producer.
"""
import torch

from .. import _lib


def pad32(n):
    return (n + 31) & ~31


class Pair:
    __slots__ = ("buf", "rows", "cols", "batch", "scale")

    def __init__(self, buf, rows, cols, batch, scale):
        self.buf, self.rows, self.cols, self.batch, self.scale = buf, rows, cols, batch, scale

    @property
    def rows_pad(self):
        return self.buf.shape[-2]

    @property
    def ld(self):
        return self.buf.shape[-1]

    @property
    def bs(self):
        return self.buf.shape[-2] * self.buf.shape[-1]

    def record_stream(self, s):
        self.buf.record_stream(s)
        self.scale.record_stream(s)


def amax_of(x, out=None):
    """max |x| of a dense fp32 tensor as an int32[1] bit pattern (device)."""
    if out is None:
        out = torch.zeros(1, dtype=torch.int32, device=x.device)
    if x.data_ptr() % 16:
        x = x.clone()                    # (16-byte lanes)
    _lib.call("cim_pair_amax", x.data_ptr(), x.numel(), out.data_ptr(), _lib.stream_ptr())
    return out


def masked_stats(dy, y, amax, want_colsum, reduce=True):
    """ReLU backward in front of a split, dz = y > 0 ? dy : 0 (not stored: split(dy, relu_y=y) applies the same mask): max |dz| into
    the zeroed int32[1] `amax`; -> column sums of dz (the layer's bias gradient) or None.  reduce=False: the per-64-row partial sums
    [ceil(rows / 64)][cols] instead (the caller sums them over dim 0 where it suits it: ops/maskfuse_pair.py, off the critical chain)."""
    rows, cols = dy.shape
    part = torch.empty(((rows + 63) // 64, cols), dtype=torch.float32, device=dy.device) if want_colsum else None
    _lib.call("cim_pair_masked_stats", dy.data_ptr(), y.data_ptr(), rows, cols, _lib.ptr(part), amax.data_ptr(), _lib.stream_ptr())
    if not want_colsum:
        return None
    return part.sum(dim=0) if reduce else part


def scales_from(amax, n=1, factor=None, reduce_all=False):
    """float32[n] power-of-two scales 2^(14 - exponent(amax * factor[i])) (amax: int32 bit patterns, [1] or [n]); reduce_all: every
    scale from the maximum of ALL words of `amax` (the row maxima of a weight -> the one scale of its image)."""
    s = torch.empty(n, dtype=torch.float32, device=amax.device)
    _lib.call("cim_pair_scales", amax.data_ptr(), amax.numel(), _lib.ptr(factor), s.data_ptr(), n, int(reduce_all), _lib.stream_ptr())
    return s


def empty(rows, cols, batch, dev, scale=None, zero_pad=True):
    """Uninitialised image whose pad rows (rows .. pad32(rows)) are zeroed."""
    rp = pad32(rows)
    buf = torch.empty((batch, rp, cols), dtype=torch.int32, device=dev)
    if zero_pad and rp != rows:
        buf[:, rows:, :].zero_()
    return Pair(buf, rows, cols, batch, scale)


def split(x, rows=None, cols=None, ld=None, batch=1, x_bs=0, scale=None, relu_y=None):
    """Generic producer: fp32 x ([batch][rows][ld], cols used) -> Pair.  scale: float32[batch] or None (from max |x|)."""
    if rows is None:
        rows, cols = x.shape[-2], x.shape[-1]
        ld = cols
    if scale is None:
        scale = scales_from(amax_of(x), batch)
    p = empty(rows, cols, batch, x.device, scale, zero_pad=False)
    _lib.call("cim_pair_split", x.data_ptr(), p.buf.data_ptr(), rows, p.rows_pad, cols, ld, cols, batch, x_bs, p.bs,
              scale.data_ptr(), _lib.ptr(relu_y), _lib.stream_ptr())
    return p


def tail_columns(m, n, k):
    """A product of T = ceil(m / 256) ceil(n / 256) tiles runs in ceil(T / 256) rounds of one workgroup per CU; when the last round
    holds only a few tiles (fc1's data gradient at <= 1024 proposals: 784 tiles = 3 rounds + 16 tiles - those 16 cost a fourth
    round: 1.07 ms where 980 tiles take 1.12) the last column tiles are better run as a SEPARATE product with split-K over the
    idle CUs.  -> (columns of the main product, k-splits of the tail) or None."""
    tm, tn = (m + 255) // 256, (n + 255) // 256
    total = tm * tn
    if total <= 256 or total % 256 == 0 or n % 256:
        return None
    main_tn = tn
    while main_tn > 0 and (main_tn * tm) % 256:
        main_tn -= 1
    tail = (tn - main_tn) * tm
    if main_tn == 0 or tail > 64:
        return None
    s = 1
    while s * 2 * tail <= 256 and k // (s * 2 * 32) >= 8 and s < 16:
        s *= 2
    return (main_tn * 256, s) if s > 1 else None


def tail_entries(m, n, k, batch):
    """Batched form of tail_columns: when the last round of batch * T tiles holds only a few of them (the Winograd data gradient at
    <= 1024 proposals: 121 x 32 = 15 rounds of 256 + 32 tiles - one more round, 82 us, for an eighth of the chip), the entries those
    tiles belong to are better run as single split-K products.  -> (entries, k-splits) or None."""
    t = ((m + 255) // 256) * ((n + 255) // 256)
    rem = (batch * t) % 256
    if batch * t <= 256 or rem == 0 or rem > 64 or rem % t:
        return None
    s = 1
    while s * 2 * t <= 256 and k // (s * 2 * 32) >= 8 and s < 16:
        s *= 2
    return (rem // t, s) if s > 1 else None


# MFMA products per multiply-add of the pair engine: 3 = fp32-class (h*l + l*h + h*h, the product's arithmetic); 1 = the h*h term
# alone (TF32-class: 11-bit operands, fp32 accumulation).  An explicit setting of the caller (bench.py's extra.tf32_class and
# tests/test_gpu_tolerance.py set it around a run); never read from the environment.
PRODUCTS = 3


def gemm(a, b, m, n, k, a_mcontig=False, b_kcontig=False, bias=None, relu=False, out=None, c_amax=None, balance=False, limit=0,
         products=None, form=0):
    """C[m,n] = A . B (+ bias)(ReLU) on pair images; k must be a multiple of 32 (the images' zero rows / columns pad it).
    Batched when a.batch > 1 (then no bias / ReLU / split-K).  balance=True: a short last round of tiles is run as its own
    split-K product over the last columns (tail_columns) / over the last batch entries (tail_entries).
    limit > 0: the product goes out as consecutive launches of at most `limit` workgroups (a workgroup of this engine owns its CU:
    a capped product leaves the rest of the chip to concurrent streams).
    form = 1 (a_mcontig and not b_kcontig only): the co-resident form of the kernel (include/cim_hip.h) - 128 x 256 tiles whose workgroups
    leave half a CU to other streams' kernels: MaskFuse's late weight gradients, ops/maskfuse_pair.py.  Same bits as form 0."""
    dev = a.buf.device
    products = PRODUCTS if products is None else products
    if a.batch > 1:
        c = out if out is not None else torch.empty((a.batch, m, n), dtype=torch.float32, device=dev)
        tail = tail_entries(m, n, k, a.batch) if balance else None
        main = a.batch - (tail[0] if tail else 0)
        _lib.call("cim_gemm_pair_batched", a.buf.data_ptr(), b.buf.data_ptr(), c.data_ptr(), m, n, k, a.ld, b.ld, n,
                  int(a_mcontig), int(b_kcontig), main, a.bs, b.bs, m * n, a.scale.data_ptr(), b.scale.data_ptr(),
                  limit, products, form, _lib.stream_ptr())
        if tail:
            ws = torch.empty(tail[1] * m * n, dtype=torch.float32, device=dev)
            for z in range(main, a.batch):
                _lib.call("cim_gemm_pair", a.buf.data_ptr() + 4 * z * a.bs, b.buf.data_ptr() + 4 * z * b.bs, c.data_ptr() + 4 * z * m * n,
                          None, m, n, k, a.ld, b.ld, n, int(a_mcontig), int(b_kcontig), 0, tail[1], ws.data_ptr(),
                          a.scale.data_ptr() + 4 * z, b.scale.data_ptr() + 4 * z, None, limit, products, form, _lib.stream_ptr())
        return c
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=dev)
    splits = _lib.call("cim_gemm_pair_splits", m, n, k)
    tail = tail_columns(m, n, k) if (balance and splits == 1) else None
    if tail is not None:
        n_main, s = tail
        n_tail = n - n_main
        ws = torch.empty(s * m * n_tail, dtype=torch.float32, device=dev)
        # column n_main of B: a row offset for a K-contiguous B ([n][k]), a column offset (4 bytes per element) otherwise
        b_off = (n_main * b.ld if b_kcontig else n_main) * 4
        for (cols, col0, boff, sp, w) in ((n_main, 0, 0, 1, None), (n_tail, n_main, b_off, s, ws)):
            _lib.call("cim_gemm_pair", a.buf.data_ptr(), b.buf.data_ptr() + boff, c.data_ptr() + 4 * col0,
                      (bias.data_ptr() + 4 * col0) if bias is not None else None, m, cols, k, a.ld, b.ld, n,
                      int(a_mcontig), int(b_kcontig), int(relu), sp, _lib.ptr(w), a.scale.data_ptr(), b.scale.data_ptr(),
                      _lib.ptr(c_amax), limit, products, form, _lib.stream_ptr())
        return c
    ws = torch.empty(splits * m * n, dtype=torch.float32, device=dev) if splits > 1 else None
    _lib.call("cim_gemm_pair", a.buf.data_ptr(), b.buf.data_ptr(), c.data_ptr(), _lib.ptr(bias), m, n, k, a.ld, b.ld, n,
              int(a_mcontig), int(b_kcontig), int(relu), splits, _lib.ptr(ws), a.scale.data_ptr(), b.scale.data_ptr(),
              _lib.ptr(c_amax), limit, products, form, _lib.stream_ptr())
    return c
