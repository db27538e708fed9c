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


class _DscGemmFn(torch.autograd.Function):
    """The K x 1 / stride K x 1 ``dsc_conv_x`` of MMConv (MMUNet.py:262) on the tokens-last sampler output:
    W2 (Cout, Cin*K) times samples (Cin*K, B*T) -> (B, Cout, T) **batch-major** (what GroupNorm wants), as
    a strided batched GEMM -- batch b's operand is rows of ``samples`` with leading dimension B*T, so
    neither the K x inflated samples nor the output are ever copied.  Backward: the (small) output
    gradient is brought to tokens-last once; dX = W2^T @ G lands in the samples' layout, dW uses split-K."""

    @staticmethod
    def forward(ctx, W2, samples, batch):
        O, I = W2.shape
        T = samples.shape[1] // batch
        ctx.save_for_backward(W2, samples)
        ctx.batch = batch
        Xb = samples.view(I, batch, T).permute(1, 0, 2)                   # (B, I, T) view, ld = B*T
        return torch.bmm(W2.unsqueeze(0).expand(batch, O, I), Xb)         # (B, O, T) contiguous

    @staticmethod
    def backward(ctx, G):
        W2, samples = ctx.saved_tensors
        B = ctx.batch
        O = W2.shape[0]
        G2 = G.permute(1, 0, 2).reshape(O, -1)                             # (O, B*T): the one small copy
        dW = dX = None
        if ctx.needs_input_grad[0]:
            dW = nt_splitk(G2, samples).to(W2.dtype)
        if ctx.needs_input_grad[1]:
            dX = W2.t() @ G2
        return dW, dX, None


def dsc_gemm(W2, samples, batch):
    """(Cout, Cin*K) x (Cin*K, B*T) tokens-last samples -> (B, Cout, T)."""
    return _DscGemmFn.apply(W2, samples, batch)
