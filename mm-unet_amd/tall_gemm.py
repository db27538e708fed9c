"""Host-side helpers for the projections around the scan, whose reductions run over all B*L tokens.

The reference writes them as plain ``F.linear`` / ``einsum`` (selective_scan_interface.py:181-182,
272-277, 394; mamba_simple.py:201-205,270).  Their *weight gradients* are ``[I x T] @ [T x J]`` products
with T = B*L up to 524,288 and I*J <= 4,608: hipBLASLt has no split-K for these and takes up to 1.5 ms
per call (measured on MI355X; 44 MMConv blocks x 3 such products per step).  float32 operands go to the
token-contraction matrix-core kernel (csrc/gemm_nt_splitk.hip: both layouts read in place, ordered slab sums);
other dtypes split T into slabs as one strided batched GEMM + a small sum.
"""
import contextlib
import os

import torch

from . import _lib, deferred, mfma_gemm

_SLAB = 2048
_NT_MIN = int(os.environ.get("MMUNET_GEMM_NT_MIN_TOKENS", "256"))   # fewer tokens: one library GEMM


def nt_splitk(X, Y):
    """X (I, T), Y (J, T), both with unit stride along T  ->  X @ Y^T  (I, J), fp32-accumulated."""
    I, T = X.shape
    J = Y.shape[0]
    if T < _NT_MIN:
        return X @ Y.t()
    if X.dtype == Y.dtype and mfma_gemm.nt_supported(X, Y, T) and X.stride(0) % 4 == 0 and Y.stride(0) % 4 == 0:
        # one kernel + its slab sum.  bf16 operands (autocast): the callers cast the float32 result to the dtype of a
        # weight that may itself be a bf16 COPY of the parameter -- the result is then a temporary, which a deferred sum
        # must never write to (deferred.py: the leaf-parameter contract): summed on the spot
        with (contextlib.nullcontext() if X.dtype == torch.float32 else deferred.paused()):
            return mfma_gemm.gemm_nt(X, Y, I, J, 1, T, X.stride(0), 0, Y.stride(0), 0)
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
    neither the K x inflated samples nor the output are ever copied.  Backward: the batch-major output gradient is read
    in place too (row stride T, batch stride O*T); dX = W2^T @ G lands in the samples' layout, dW is the
    token-contraction product (csrc/gemm_nt_splitk.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, W2, samples, batch):
        O, I = W2.shape
        T = samples.shape[1] // batch
        ctx.save_for_backward(W2, samples)
        ctx.batch = batch
        # inside MM_Net's prepared_weights block: the image of W2^T for the input gradient, kept for the backward pass
        ctx.prep_t = mfma_gemm.prepared_for(W2, I, O, True) if W2.is_contiguous() else None
        if mfma_gemm.supported(O, I, batch * T, W2, samples) and W2.stride(1) == 1 and samples.is_contiguous() \
                and T % 4 == 0:
            out = torch.empty((batch, O, T), device=samples.device, dtype=torch.float32)
            return mfma_gemm.gemm_tokens(W2, samples, out, O, I, T, batch, batch * T, T, T, O * T)
        Xb = samples.view(I, batch, T).permute(1, 0, 2)                   # (B, I, T) view, ld = B*T
        return torch.bmm(W2.unsqueeze(0).expand(batch, O, I), Xb)         # (B, O, T) contiguous

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, G):
        W2, samples = ctx.saved_tensors
        G = G.float()
        B = ctx.batch
        O, I = W2.shape
        T = G.shape[2]
        dW = dX = None
        # the batch-major output gradient is read in place by both products (row stride T, batch stride O*T)
        inplace = G.is_contiguous() and samples.is_contiguous() and T % 4 == 0 and W2.stride(1) == 1
        G2 = None
        if ctx.needs_input_grad[0]:
            if inplace and T % 32 == 0 and B * T >= _NT_MIN and mfma_gemm.nt_supported(G, samples, T):
                dW = mfma_gemm.gemm_nt(G, samples, O, I, B, T, T, O * T, B * T, T).to(W2.dtype)
            else:
                G2 = G.permute(1, 0, 2).reshape(O, -1)                         # (O, B*T): one small copy
                dW = nt_splitk(G2, samples).to(W2.dtype)
        if ctx.needs_input_grad[1]:
            if inplace and mfma_gemm.supported(I, O, B * T, W2, G):
                dX = torch.empty((I, B * T), device=G.device, dtype=torch.float32)
                mfma_gemm.gemm_tokens(W2, G, dX, I, O, T, B, T, O * T, B * T, T, transposed_weight=True,
                                      prepared=ctx.prep_t)
            else:
                if G2 is None:
                    G2 = G.permute(1, 0, 2).reshape(O, -1)
                if mfma_gemm.supported(I, O, G2.shape[1], W2, G2) and W2.stride(1) == 1 and G2.is_contiguous():
                    NT = G2.shape[1]
                    dX = torch.empty((I, NT), device=G2.device, dtype=torch.float32)
                    mfma_gemm.gemm_tokens(W2, G2, dX, I, O, NT, 1, NT, 0, NT, 0, transposed_weight=True,
                                          prepared=ctx.prep_t)
                else:
                    dX = W2.t() @ G2
        return dW, dX, None


def dsc_gemm(W2, samples, batch):
    """(Cout, Cin*K) x (Cin*K, B*T) tokens-last samples -> (B, Cout, T)."""
    return _DscGemmFn.apply(W2, samples, batch)


def _mfma_ok(rows, inner, B, L, W, X, out):
    """One strided-batch launch of the matrix-core GEMM instead of B library calls (both operands addressed in
    place): unit token stride, strides that keep 16-byte alignment."""
    return (mfma_gemm.supported(rows, inner, B * L, W, X, out) and L % 4 == 0 and W.is_contiguous()
            and X.stride(2) == 1 and out.stride(2) == 1
            and all(t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 for t in (X, out)))


def _channel_major_2d(t):
    """(B, C, L) tensor -> its (C, B*L) matrix; a view if the storage is already [C][B][L], one copy otherwise."""
    B, C, L = t.shape
    tc = t.permute(1, 0, 2)
    if not tc.is_contiguous():
        tc = tc.contiguous()
    return tc.view(C, B * L)


class _ProjBclFn(torch.autograd.Function):
    """``W (O, I)`` applied per batch item between the two layouts that meet at the Mamba block of RCG
    (MMUNet.py:398-412): feature maps are ``[B][C][L]`` (batch-major), the fused Mamba path works on
    ``[C][B][L]`` (tokens-last).  ``to_cb=True``:  X (B, I, L) contiguous -> (B, O, L) tensor laid out
    [O][B][L];  ``to_cb=False``: X (B, I, L) laid out [I][B][L] -> (B, O, L) contiguous.  Each batch item is
    one GEMM whose strided operand / result is addressed in place (leading dimension B*L), so neither side is
    ever transposed or copied (the reference's ``(B, L, C)`` interface costs a 134 MB transposing copy each
    way at 256 x 256, plus one more for the weight gradient).  Weight gradient: ONE split-K product over all
    (batch, token) pairs on (channels, B*L) matrices (the batch-major operand is transposed once for it)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, W, X, to_cb):
        B, I, L = X.shape
        O = W.shape[0]
        # May the weight gradient's final sum wait for a deferred.Scope's launch?  Only if nothing reads it during the
        # backward pass: W is a parameter, or a view of one (view backward ops move no data).  A COPY of a parameter
        # (morph_mix's permuted, padded weight) has its gradient sliced / permuted / cloned by autograd on the spot.
        base = W._base if W._is_view() else None
        ctx.defer_dw = bool(W.is_leaf or (base is not None and base.is_leaf and W.is_contiguous()))
        if to_cb:
            out = torch.empty((O, B, L), device=X.device, dtype=X.dtype).permute(1, 0, 2)   # [O][B][L]
        else:
            out = torch.empty((B, O, L), device=X.device, dtype=X.dtype)
        if _mfma_ok(O, I, B, L, W, X, out):
            mfma_gemm.gemm_tokens(W, X, out, O, I, L, B, X.stride(1), X.stride(0), out.stride(1), out.stride(0))
        else:
            for b in range(B):
                torch.mm(W, X[b], out=out[b])
        ctx.save_for_backward(W, X)
        ctx.to_cb = to_cb
        # (inside a prepared_weights block: W^T's image for the input gradient, see _DscGemmFn)
        ctx.prep_t = mfma_gemm.prepared_for(W, I, O, True) if W.dim() == 2 and W.is_contiguous() else None
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, G):
        W, X = ctx.saved_tensors
        G = G.float()
        B, I, L = X.shape
        dW = dX = None
        if ctx.needs_input_grad[1]:
            if ctx.to_cb:
                dX = torch.empty((B, I, L), device=X.device, dtype=X.dtype)
            else:
                dX = torch.empty((I, B, L), device=X.device, dtype=X.dtype).permute(1, 0, 2)
            if _mfma_ok(I, W.shape[0], B, L, W, G, dX):
                mfma_gemm.gemm_tokens(W, G, dX, I, W.shape[0], L, B, G.stride(1), G.stride(0), dX.stride(1),
                                      dX.stride(0), transposed_weight=True, prepared=getattr(ctx, "prep_t", None))
            else:
                Wt = W.t()
                for b in range(B):
                    torch.mm(Wt, G[b], out=dX[b])
        if ctx.needs_input_grad[0]:
            # one split-K product over all (batch, token) pairs: both operands as (channels, B*L) matrices.  One of the
            # two is batch-major by construction (the layouts meet here); it is brought to channel-major with one
            # transposing copy -- 2-3 launches instead of three per batch item (24 -> 4 for B = 8)
            if (G.stride(2) == 1 and X.stride(2) == 1 and mfma_gemm.nt_supported(G, X, L) and B * L >= _NT_MIN
                    and all(t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 for t in (G, X))):
                # both layouts addressed in place: no transposing copy of the batch-major operand
                with (contextlib.nullcontext() if ctx.defer_dw else deferred.paused()):
                    dW = mfma_gemm.gemm_nt(G, X, W.shape[0], I, B, L, G.stride(1), G.stride(0), X.stride(1),
                                           X.stride(0)).to(W.dtype)
            else:
                with (contextlib.nullcontext() if ctx.defer_dw else deferred.paused()):
                    dW = nt_splitk(_channel_major_2d(G), _channel_major_2d(X)).to(W.dtype)
        return dW, dX, None


class _ProjBclLowpFn(torch.autograd.Function):
    """``_ProjBclFn`` for bfloat16 activations (autocast): the same two layouts and no transposing copies, bf16 operands
    and results, float32 accumulation.  Round 4: ONE launch of ``gemm_tokens``' bf16 form over all batch items with the
    FLOAT32 weight (its two-part image: products exact to 16 bits of the weight, where a weight cast to bf16 keeps 8) --
    before: a cast of the weight and one library GEMM per batch item.  The weight gradient is the split-K product over
    all tokens, summed in float32."""

    @staticmethod
    def forward(ctx, W, X, to_cb):
        with torch.autocast("cuda", enabled=False):
            B, I, L = X.shape
            O = W.shape[0]
            Xc = X.to(torch.bfloat16)
            if to_cb:
                out = torch.empty((O, B, L), device=X.device, dtype=torch.bfloat16).permute(1, 0, 2)   # [O][B][L]
            else:
                out = torch.empty((B, O, L), device=X.device, dtype=torch.bfloat16)
            base = W._base if W._is_view() else None   # (the leaf-parameter contract of deferred sums: _ProjBclFn)
            ctx.defer_dw = bool(W.dtype == torch.float32 and
                                (W.is_leaf or (base is not None and base.is_leaf and W.is_contiguous())))
            own = W.is_contiguous() and mfma_gemm.tokens_lowp_supported(W, Xc, out)
            if own:
                mfma_gemm.gemm_tokens(W, Xc, out, O, I, L, B, Xc.stride(1), Xc.stride(0), out.stride(1), out.stride(0))
                Wc = W
            else:
                Wc = W.to(torch.bfloat16)
                for b in range(B):
                    torch.mm(Wc, Xc[b], out=out[b])
        ctx.save_for_backward(Wc, Xc)
        ctx.to_cb, ctx.w_dtype, ctx.x_dtype, ctx.own = to_cb, W.dtype, X.dtype, own
        return out

    @staticmethod
    def backward(ctx, G):
        Wc, Xc = ctx.saved_tensors
        with torch.autocast("cuda", enabled=False):
            G = G.to(torch.bfloat16)
            B, I, L = Xc.shape
            dW = dX = None
            if ctx.needs_input_grad[1]:
                if ctx.to_cb:
                    dX = torch.empty((B, I, L), device=G.device, dtype=torch.bfloat16)
                else:
                    dX = torch.empty((I, B, L), device=G.device, dtype=torch.bfloat16).permute(1, 0, 2)
                if ctx.own and G.stride(2) == 1 and mfma_gemm.tokens_lowp_supported(Wc, G, dX):
                    mfma_gemm.gemm_tokens(Wc, G, dX, I, Wc.shape[0], L, B, G.stride(1), G.stride(0), dX.stride(1),
                                          dX.stride(0), transposed_weight=True)
                else:
                    Wt = Wc.to(torch.bfloat16).t()
                    for b in range(B):
                        torch.mm(Wt, G[b], out=dX[b])
                dX = dX.to(ctx.x_dtype)
            if ctx.needs_input_grad[0]:
                if (G.stride(2) == 1 and Xc.stride(2) == 1 and mfma_gemm.nt_supported(G, Xc, L) and B * L >= _NT_MIN
                        and all(t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0 for t in (G, Xc))):
                    # both layouts read in place by the token-contraction kernel's bf16 form (exact products)
                    with (contextlib.nullcontext() if ctx.defer_dw else deferred.paused()):
                        dW = mfma_gemm.gemm_nt(G, Xc, Wc.shape[0], I, B, L, G.stride(1), G.stride(0), Xc.stride(1),
                                               Xc.stride(0)).to(ctx.w_dtype)
                else:
                    with deferred.paused():
                        dW = nt_splitk(_channel_major_2d(G), _channel_major_2d(Xc)).to(ctx.w_dtype)
        return dW, dX, None


def proj_bcl(W, X, to_cb):
    if X.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return _ProjBclLowpFn.apply(W, X, to_cb)
    return _ProjBclFn.apply(W, X, to_cb)


class _Conv1x1Stride2Fn(torch.autograd.Function):
    """``nn.Conv2d(I, O, kernel_size=1, stride=2, bias=False)`` -- the shortcut of the down-sampling residual blocks
    (src/UM_Net/MMUNet.py:448) -- as a strided gather + ``W @ tokens`` per batch item.  The library runs it as an
    implicit GEMM behind layout transposes (30 us forward, 86-99 us backward per call) and, like its 3 x 3 stride-2
    kernels, not run-to-run reproducibly (tools/dbg/fwd_determinism.py); this path is deterministic: ordered GEMMs,
    token-contraction weight gradient, plain strided scatter for the input gradient."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, slot=None):
        B, I, H, W = x.shape
        O = weight.shape[0]
        xs = x[:, :, ::2, ::2].contiguous()
        Ho, Wo = xs.shape[2], xs.shape[3]
        T = Ho * Wo
        W2 = weight.view(O, I)
        out = torch.empty((B, O, T), device=x.device, dtype=torch.float32)
        xs3 = xs.view(B, I, T)
        ctx.prep_t = None
        if mfma_gemm.supported(O, I, B * T, W2, xs3, out) and T % 4 == 0:
            mfma_gemm.gemm_tokens(W2, xs3, out, O, I, T, B, T, I * T, T, O * T)
            ctx.prep_t = mfma_gemm.prepared_for(W2, I, O, True)     # (the W^T image, for the backward pass)
        else:
            torch.matmul(W2, xs3, out=out)
        ctx.save_for_backward(W2, xs3)
        ctx.in_hw = (H, W)
        # slot: a conv3x3_small.SharedGrad of all consumers of x -- their input gradients leave as one
        ctx.slot = slot if (slot is not None and ctx.needs_input_grad[0]) else None
        if ctx.slot is not None:
            ctx.slot.join()
        return out.view(B, O, Ho, Wo)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, G):
        W2, xs3 = ctx.saved_tensors
        B, I, T = xs3.shape
        O = W2.shape[0]
        H, W = ctx.in_hw
        G3 = G.float().contiguous().view(B, O, T)
        dx = dW = None
        if ctx.needs_input_grad[0]:
            # d xs = W^T G per batch item (strided-batch matrix-core GEMM: torch.matmul folds the batch into the token
            # axis behind two transposing copies), then the even pixels of d x -- in place on the gradient another
            # consumer of x left, or a zero-filled tensor
            dxs = torch.empty((B, I, T), device=G.device, dtype=torch.float32)
            # (the matrix-core kernel also below its usual break-even of 192 tiles: the library's strided-batch kernel
            #  for 64 x 128 against 8 x 4,096 tokens takes 121 us, 64 workgroups of this one ~10)
            if (mfma_gemm.ENABLED and T % 4 == 0 and W2.is_contiguous()
                    and not torch.is_autocast_enabled() and G3.data_ptr() % 16 == 0 and W2.data_ptr() % 16 == 0):
                mfma_gemm.gemm_tokens(W2, G3, dxs, I, O, T, B, T, O * T, T, I * T, transposed_weight=True,
                                      prepared=ctx.prep_t)
            else:
                torch.matmul(W2.t(), G3, out=dxs)
            parked = ctx.slot.take() if ctx.slot is not None else None
            if parked is not None and (tuple(parked.shape) != (B, I, H, W) or parked.dtype != torch.float32
                                       or not parked.is_contiguous()):
                raise RuntimeError("conv1x1_stride2: parked input gradient does not match the input")
            dx = parked if parked is not None else torch.empty((B, I, H, W), device=G.device, dtype=torch.float32)
            with torch.cuda.device(G.device):
                _lib.check(_lib.lib().mmu_scatter_stride2(dxs.data_ptr(), dx.data_ptr(), _lib.ptr(parked), B * I, H, W,
                                                          _lib.stream_of(G)))
            if ctx.slot is not None:
                dx = ctx.slot.give(dx)
        if ctx.needs_input_grad[1]:
            if T % 32 == 0 and B * T >= _NT_MIN and mfma_gemm.nt_supported(G3, xs3, T):
                dW = mfma_gemm.gemm_nt(G3, xs3, O, I, B, T, T, O * T, T, I * T)
            else:
                dW = torch.einsum("bot,bit->oi", G3, xs3)
            dW = dW.view(O, I, 1, 1)
        return dx, dW, None


def conv1x1_stride2(x, weight, slot=None):
    return _Conv1x1Stride2Fn.apply(x, weight, slot)


STRIDE2_ENABLED = True   # False: the shortcut convolutions stay nn.Conv2d calls (fused_paths.plain_aten)


def conv1x1_stride2_supported(m, x):
    return (STRIDE2_ENABLED and isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1) and m.stride == (2, 2) and m.padding == (0, 0)
            and m.bias is None and m.groups == 1 and m.dilation == (1, 1) and x.is_cuda and x.dim() == 4
            and x.dtype == torch.float32 and m.weight.dtype == torch.float32 and not torch.is_autocast_enabled())
