"""Host-side helpers for the projections around the scan, whose reductions run over all B*L tokens.

The reference writes them as plain ``F.linear`` / ``einsum`` (selective_scan_interface.py:181-182,
272-277, 394; mamba_simple.py:201-205,270).  Their *weight gradients* are ``[I x T] @ [T x J]`` products
with T = B*L up to 524,288 and I*J <= 4,608: hipBLASLt has no split-K for these and takes up to 1.5 ms
per call (measured on MI355X; 44 MMConv blocks x 3 such products per step).  Splitting T into slabs
turns each into one strided batched GEMM + a small sum: 27 us.
"""
import torch

_SLAB = 2048


def nt_splitk(X, Y):
    """X (I, T), Y (J, T), both with unit stride along T  ->  X @ Y^T  (I, J), fp32-accumulated."""
    I, T = X.shape
    J = Y.shape[0]
    if T < 4 * _SLAB:
        return X @ Y.t()
    S = T // _SLAB
    Tm = S * _SLAB
    Xs = X[:, :Tm].reshape(I, S, _SLAB).transpose(0, 1)          # (S, I, slab) view
    Ys = Y[:, :Tm].reshape(J, S, _SLAB).permute(1, 2, 0)         # (S, slab, J) view
    out = torch.bmm(Xs, Ys).float().sum(0)   # slab sums added in fp32 also for bf16 operands
    if Tm < T:
        out = out + X[:, Tm:] @ Y[:, Tm:].t()
    return out


class _ProjTokensFn(torch.autograd.Function):
    """W (O, I) applied to a tokens-last matrix Xm (I, T): returns W @ Xm (O, T); the weight gradient
    uses split-K."""

    @staticmethod
    def forward(ctx, W, Xm):
        ctx.w_dtype, ctx.x_dtype = W.dtype, Xm.dtype
        if torch.is_autocast_enabled():          # behave like F.linear under autocast
            ac = torch.get_autocast_dtype("cuda")
            W, Xm = W.to(ac), Xm.to(ac)
        elif W.dtype != Xm.dtype:
            W = W.to(Xm.dtype)
        ctx.save_for_backward(W, Xm)
        return W @ Xm

    @staticmethod
    def backward(ctx, G):
        W, Xm = ctx.saved_tensors
        G = G.to(W.dtype)
        dW = dX = None
        if ctx.needs_input_grad[0]:
            dW = nt_splitk(G if G.stride(-1) == 1 else G.contiguous(),
                           Xm if Xm.stride(-1) == 1 else Xm.contiguous()).to(ctx.w_dtype)
        if ctx.needs_input_grad[1]:
            dX = (W.t() @ G).to(ctx.x_dtype)
        return dW, dX


def proj_tokens(W, Xm):
    """W @ Xm for Xm of shape (in_features, tokens) -- tall-K-safe autograd."""
    return _ProjTokensFn.apply(W, Xm)
