"""Token re-orderings of the tri-directional ("v3") Mamba block as two fused HIP ops (mamba_simple.py:212-270).

``tri_split(x, ns)``  -> ``(x, x_flip, x_slice)``: ``x.flip(-1)`` and the slice-interleaved copy
(token i of slice s -> position i*ns + s, :245-247) in one pass over ``x``.
``tri_combine(a, b_flip, c_slice, ns)`` -> ``a + b_flip.flip(-1) + unslice(c_slice)`` (:263-270) in one pass.
Each is the other's adjoint, so each backward is one launch of the other kernel and the three gradients
meet without the two ``add_`` passes autograd would use.  float32 (bfloat16 for 5..64 slices) tensors ``(B, C, L)``
with unit L-stride and dense rows (the ``[C][B][L]`` layout of the fused path or plain contiguous); ``supported()`` tells.
"""
import torch

from . import _lib


def _dense_rows(t):
    """(B, C, L) tensor whose (b, c) rows are dense in memory in some order -> permutation tag, else None."""
    if t.dim() != 3 or t.stride(2) != 1 or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda:
        return None
    B, C, L = t.shape
    if t.stride(0) == C * L and t.stride(1) == L:
        return "bc"
    if t.stride(1) == B * L and t.stride(0) == L:
        return "cb"
    return None


ENABLED = True   # False: Mamba's tri-directional branch uses flip / stack / permute (fused_paths.plain_aten)


def supported(*tensors, nslices=None):
    """float32, or bfloat16 when 5 <= nslices <= 64 (the kernels' tiled form; pass nslices to have it checked)."""
    if not ENABLED:
        return False
    tags = [_dense_rows(t) for t in tensors]
    lowp = tensors[0].dtype != torch.float32
    return tags[0] is not None and all(tag == tags[0] for tag in tags) and \
        all(t.shape == tensors[0].shape and t.dtype == tensors[0].dtype for t in tensors) and \
        tensors[0].shape[0] * tensors[0].shape[1] < 65536 and (not lowp or (nslices is not None and 4 < nslices <= 64))


def _like(t):
    return torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype)


def _params(ref, ns):
    p = _lib.TriParams()
    p.rows, p.seqlen, p.nslices = ref.shape[0] * ref.shape[1], ref.shape[2], ns
    p.dtype = _lib.dtype_code(ref)
    return p


def _split(x, ns):
    if not supported(x, nslices=ns) or x.shape[2] % ns != 0:
        raise RuntimeError("tri_split: float32 (or, for 5..64 slices, bfloat16) (B, C, L) tensor with dense rows and L "
                           "divisible by nslices required")
    xf, xs = _like(x), _like(x)
    p = _params(x, ns)
    p.a, p.flip, p.slice = x.data_ptr(), xf.data_ptr(), xs.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_tri_split(p, _lib.stream_of(x)))
    return xf, xs


def _combine(a, bf, cs, ns):
    if not supported(a, bf, cs, nslices=ns):   # gradients can arrive in another layout (or dtype): bring them to a's
        if _dense_rows(a) is None or not supported(a, nslices=ns):
            a = a.float().contiguous()
        bf, cs = (t if (_dense_rows(t) == _dense_rows(a) and t.dtype == a.dtype) else _relayout(t, a) for t in (bf, cs))
    if not supported(a, bf, cs, nslices=ns):
        raise RuntimeError("tri_combine: float32 / bfloat16 (B, C, L) tensors with dense rows required")
    out = _like(a)
    p = _params(a, ns)
    p.a, p.flip, p.slice, p.out = a.data_ptr(), bf.data_ptr(), cs.data_ptr(), out.data_ptr()
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().mmu_tri_combine(p, _lib.stream_of(a)))
    return out


def _relayout(t, ref):
    o = _like(ref)
    o.copy_(t)
    return o


class TriSplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ns):
        _lib.require_gpu(x)
        ctx.ns = ns
        xf, xs = _split(x, ns)
        return x.view_as(x), xf, xs

    @staticmethod
    def backward(ctx, ga, gf, gs):
        ref = next(g for g in (ga, gf, gs) if g is not None)
        ga, gf, gs = (torch.zeros_like(ref) if g is None else g for g in (ga, gf, gs))
        return _combine(ga, gf, gs, ctx.ns), None


class TriCombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, bf, cs, ns):
        _lib.require_gpu(a, bf, cs)
        ctx.ns = ns
        return _combine(a, bf, cs, ns)

    @staticmethod
    def backward(ctx, g):
        if not supported(g, nslices=ctx.ns):
            g = g.float().contiguous()
        gf, gs = _split(g, ctx.ns)
        return g, gf, gs, None


def tri_split(x, nslices):
    """x (B, C, L) -> (x, x.flip(-1), slice-interleaved x)."""
    return TriSplitFn.apply(x, nslices)


def tri_combine(a, b_flip, c_slice, nslices):
    """a + b_flip.flip(-1) + unslice(c_slice)."""
    return TriCombineFn.apply(a, b_flip, c_slice, nslices)
