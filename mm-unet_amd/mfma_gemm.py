"""``gemm_tokens`` -- ``W (M, K) @ X (K, T)`` for tokens-last ``X`` on the bf16 matrix cores with float32 accuracy
(csrc/gemm_tokens_mfma.hip; hi/lo bf16 split, three MFMAs per product).  float32 only; any M / K since ABI 6 (the weight
image is zero-padded to 64 rows / 16 columns); ``dt_proj`` / ``dt_proj_input_grad``: csrc/dt_proj.hip."""
import os

import torch

from . import _lib, deferred

ENABLED = True   # False: callers use their ATen GEMMs (tests compare the two)
ANY_SHAPE = os.environ.get("MMUNET_GEMM_TOKENS_ANY_SHAPE", "1") != "0"   # rows / inner that are no multiples of 64 / 16
# products with fewer 64-row x 512-token tiles than this stay with the library (0: none do -- below 192 tiles the C entry
# point switches to its 32-token kernel with the inner dimension split over the waves, csrc/gemm_tokens_mfma.hip)
MIN_TILES = int(os.environ.get("MMUNET_GEMM_TOKENS_MIN_TILES", "0"))


def supported(rows, inner, tokens, *tensors):
    """Shapes the matrix-core kernels cover: any rows / inner, tokens a multiple of 4, float32, 16-byte aligned."""
    tiles = ((rows + 63) // 64) * ((tokens + 511) // 512)      # any rows / inner since ABI 6 (zero-padded weight images)
    return (ENABLED and (ANY_SHAPE or (rows % 64 == 0 and inner % 16 == 0)) and tokens % 4 == 0 and tiles >= MIN_TILES
            and not torch.is_autocast_enabled()
            and all(t.is_cuda and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 for t in tensors))


_PREPARED = {}   # (weight data_ptr, rows, inner, transposed) -> uint8 image; alive only inside a prepared_weights block


class prepared_weights:
    """``with prepared_weights(weights):`` -- the bf16 hi/lo images ``gemm_tokens`` multiplies with, for every 2-D float32
    weight in the list and both orientations (``W . X`` and ``W^T . G``), written by ONE launch on entry instead of one
    4.8 us launch in front of each product (MM_Net: 48 per training step).  Inside the block ``gemm_tokens`` finds the
    image by the weight's address; ``prepared_for`` hands a caller the transposed image to keep for its backward pass
    (which runs after the block has been left -- the arena is only rewritten by the next entry, i.e. by the next forward
    pass, after the weights may have changed).  The table of addresses is built once: parameters keep their storage."""

    def __init__(self, weights_fn):
        # weights_fn() -> the weights as they are NOW (parameters may have been moved or replaced since the last pass);
        # each is read as the (shape[0], numel / shape[0]) matrix of its contiguous storage
        self.weights_fn = weights_fn
        self.weights = []
        self._key = None
        self._images = {}
        self._table = None
        self._arena = None
        self._max = 0

    def _build(self):
        dev = self.weights[0].device
        L = _lib.lib()
        items, off = [], 0
        for w in self.weights:
            m, k = w.shape[0], w.numel() // w.shape[0]
            for rows, inner, trans in ((m, k, 0), (k, m, 1)):
                if True:   # any rows / inner since ABI 6 (zero-padded images)
                    nbytes = (int(L.mmu_gemm_tokens_workspace_bytes(rows, inner)) + 255) // 256 * 256
                    items.append((w, rows, inner, trans, off, nbytes))
                    off += nbytes
        self._arena = torch.empty(max(off, 16), device=dev, dtype=torch.uint8)
        rows_ = []
        self._images = {}
        for w, rows, inner, trans, o, nbytes in items:
            img = self._arena[o:o + nbytes]
            self._images[(w.data_ptr(), rows, inner, trans)] = img
            rows_.append([w.data_ptr(), w.numel() // w.shape[0], img.data_ptr(), rows, inner, trans])
            self._max = max(self._max, ((rows + 63) // 64 * 64) * ((inner + 15) // 16 * 16))
        self._table = torch.tensor(rows_, dtype=torch.int64).to(dev) if rows_ else None
        self._key = tuple(w.data_ptr() for w in self.weights)

    def __enter__(self):
        ws = self.weights = [w for w in self.weights_fn() if w.dtype == torch.float32 and w.is_contiguous()]
        if not ENABLED or not ws or not all(w.is_cuda for w in ws) or torch.is_autocast_enabled():
            return self
        if self._key != tuple(w.data_ptr() for w in ws):   # first use, or the parameters were moved / reloaded
            self._build()
        if self._table is not None:
            dev = ws[0].device
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().mmu_gemm_tokens_prepare_batch(self._table.data_ptr(), self._table.shape[0], self._max,
                                                                    torch.cuda.current_stream(dev).cuda_stream))
            _PREPARED.update(self._images)
        return self

    def __exit__(self, *exc):
        _PREPARED.clear()
        return False


def prepared_for(weight, rows, inner, transposed_weight):
    """The prepared image of ``weight`` for this orientation, or None (outside a prepared_weights block / not listed)."""
    return _PREPARED.get((weight.data_ptr(), rows, inner, int(transposed_weight)))


def gemm_tokens(weight, x, out, rows, inner, tokens, batch, x_rs, x_bs, out_rs, out_bs, transposed_weight=False,
                prepared=None, accumulate=False):
    """out[b] = W . X[b]; ``weight`` is (rows, inner) -- or (inner, rows) read transposed -- with unit inner stride;
    ``x`` / ``out`` are float32 tensors whose storage holds the strided operands described by the element strides.
    ``prepared``: this weight's image from :class:`prepared_weights` (looked up by address when not given)."""
    _lib.require_gpu(weight, x, out)
    if weight.dtype != torch.float32 or x.dtype not in (torch.float32, torch.bfloat16) or out.dtype != x.dtype:
        raise RuntimeError("gemm_tokens: a float32 weight and float32 or bfloat16 x / out of one type required")
    if weight.stride(-1) != 1 or weight.dim() != 2:
        raise RuntimeError("gemm_tokens: weight must be a 2-D matrix with unit column stride")
    want = (inner, rows) if transposed_weight else (rows, inner)
    if tuple(weight.shape) != want:
        raise RuntimeError(f"gemm_tokens: weight shape {tuple(weight.shape)} != {want}")
    if prepared is None and weight.is_contiguous():
        prepared = prepared_for(weight, rows, inner, transposed_weight)
    need = _lib.lib().mmu_gemm_tokens_workspace_bytes(rows, inner)
    if prepared is not None and (prepared.numel() < need or prepared.device != x.device or prepared.dtype != torch.uint8):
        raise RuntimeError("gemm_tokens: the prepared weight image does not belong to this product")
    ws = prepared if prepared is not None else torch.empty(need, device=x.device, dtype=torch.uint8)
    p = _lib.GemmTokensParams()
    p.rows, p.inner, p.tokens, p.batch, p.transposed_weight = rows, inner, tokens, batch, int(transposed_weight)
    p.weight, p.w_ld = (None if prepared is not None else weight.data_ptr()), weight.stride(0)
    p.x, p.x_rs, p.x_bs = x.data_ptr(), x_rs, x_bs
    p.out, p.out_rs, p.out_bs = out.data_ptr(), out_rs, out_bs
    p.workspace = ws.data_ptr()
    p.accumulate = int(bool(accumulate))   # out += W . X
    p.x_dtype = p.out_dtype = _lib.dtype_code(x)   # (bf16 activations: the 512-token kernel's XB form, any tile count)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_gemm_tokens_mfma(p, _lib.stream_of(x)))
    return out


def dt_proj(weight, dt_rows, delta=None):
    """delta (dim, T) = weight (dim, rank) @ dt_rows (rank, T) -- Mamba's dt_proj on tokens-last rows
    (csrc/dt_proj.hip; selective_scan_interface.py:182).  float32, rank <= 8, T % 4 == 0."""
    _lib.require_gpu(weight, dt_rows)
    dim, rank = weight.shape
    T = dt_rows.shape[1]
    if delta is None:
        delta = torch.empty((dim, T), device=dt_rows.device, dtype=dt_rows.dtype)
    _dt_proj_call("mmu_dt_proj_fwd", weight, dt_rows, delta)
    return delta


def dt_proj_input_grad(weight, ddelta, ddt_rows):
    """ddt_rows (rank, T) = weight^T (rank, dim) @ ddelta (dim, T), written in place (csrc/dt_proj.hip;
    selective_scan_interface.py:274)."""
    _lib.require_gpu(weight, ddelta, ddt_rows)
    _dt_proj_call("mmu_dt_proj_bwd", weight, ddt_rows, ddelta)
    return ddt_rows


def _io_rows_ok(a, b):
    """Token rows of csrc/dt_proj.hip's streaming kernels: two 2-D matrices of ONE dtype with unit token stride -- float32
    (outside autocast; 16-byte aligned) or bfloat16 (8-byte aligned; read and written natively, float32 arithmetic)."""
    if a.dtype != b.dtype or not all(t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 for t in (a, b)):
        return False
    if a.dtype == torch.bfloat16:
        return LOWP and a.data_ptr() % 8 == 0 and b.data_ptr() % 8 == 0
    return (a.dtype == torch.float32 and not torch.is_autocast_enabled() and a.data_ptr() % 16 == 0
            and b.data_ptr() % 16 == 0)


def dt_proj_supported(weight, dt_rows, delta):
    return (ENABLED and weight.dim() == 2 and 1 <= weight.shape[1] <= 8 and weight.stride(1) == 1
            and all(t is not None and t.is_cuda for t in (weight, dt_rows, delta)) and weight.dtype == torch.float32
            and _io_rows_ok(dt_rows, delta)
            and dt_rows.shape[1] % 4 == 0 and dt_rows.shape[1] == delta.shape[1]
            and dt_rows.shape[0] == weight.shape[1] and delta.shape[0] == weight.shape[0])


def _dt_proj_call(name, weight, dt_rows, delta):
    if not dt_proj_supported(weight, dt_rows, delta):
        raise RuntimeError(f"{name}: float32 (or, both, bfloat16) tokens-last matrices with unit token stride, float32 weight, rank <= 8, tokens % 4 == 0 required")
    p = _lib.DtProjParams()
    p.rank, p.dim, p.tokens = weight.shape[1], weight.shape[0], dt_rows.shape[1]
    p.dt, p.dt_rs = dt_rows.data_ptr(), dt_rows.stride(0)
    p.weight, p.w_ld = weight.data_ptr(), weight.stride(0)
    p.delta, p.delta_rs = delta.data_ptr(), delta.stride(0)
    p.io_dtype = _lib.dtype_code(delta)
    with torch.cuda.device(delta.device):
        _lib.check(getattr(_lib.lib(), name)(p, _lib.stream_of(delta)))


X_PROJ_ROWS = (33, 34, 36, 40)   # dt_rank + 2 * d_state the streaming x_proj kernels are instantiated for


def x_proj_supported(weight, x, x_dbl):
    return (ENABLED and weight.dim() == 2 and weight.shape[0] in X_PROJ_ROWS and weight.stride(1) == 1
            and weight.shape[1] % 4 == 0 and weight.shape[1] <= 1024 and weight.shape[1] * ((weight.shape[0] + 3) // 4 * 4) * 4 <= 65536
            and all(t is not None and t.is_cuda for t in (weight, x, x_dbl)) and weight.dtype == torch.float32
            and _io_rows_ok(x, x_dbl)
            and x.shape[1] % 4 == 0 and x.shape[1] == x_dbl.shape[1] and x.shape[0] == weight.shape[1]
            and x_dbl.shape[0] == weight.shape[0])


def _x_proj_call(name, weight, x, x_dbl):
    if not x_proj_supported(weight, x, x_dbl):
        raise RuntimeError(f"{name}: float32 (or, both, bfloat16) tokens-last matrices with unit token stride, float32 weight, rows in {X_PROJ_ROWS}, dim % 4 == 0 required")
    p = _lib.XProjParams()
    p.rows, p.dim, p.tokens = weight.shape[0], weight.shape[1], x.shape[1]
    p.x, p.x_rs = x.data_ptr(), x.stride(0)
    p.weight, p.w_ld = weight.data_ptr(), weight.stride(0)
    p.x_dbl, p.x_dbl_rs = x_dbl.data_ptr(), x_dbl.stride(0)
    p.io_dtype = _lib.dtype_code(x)
    with torch.cuda.device(x.device):
        _lib.check(getattr(_lib.lib(), name)(p, _lib.stream_of(x)))


def x_proj(weight, x, x_dbl=None):
    """x_dbl (rows, T) = weight (rows, dim) @ x (dim, T): exact float32 products, streaming (csrc/dt_proj.hip;
    selective_scan_interface.py:181)."""
    _lib.require_gpu(weight, x)
    if x_dbl is None:
        x_dbl = torch.empty((weight.shape[0], x.shape[1]), device=x.device, dtype=x.dtype)
    _x_proj_call("mmu_x_proj_fwd", weight, x, x_dbl)
    return x_dbl


def x_proj_input_grad_add(weight, dx_dbl, dx):
    """dx (dim, T) += weight^T @ dx_dbl (rows, T), in place (selective_scan_interface.py:277)."""
    _lib.require_gpu(weight, dx_dbl, dx)
    _x_proj_call("mmu_x_proj_bwd", weight, dx, dx_dbl)
    return dx


def tokens_lowp_supported(weight, *tensors):
    """gemm_tokens with bfloat16 activations (csrc/gemm_tokens_mfma.hip, XB form): float32 2-D weight, bf16 operands /
    results with unit token stride, 8-byte aligned, row / batch strides multiples of 4."""
    return (ENABLED and LOWP and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_cuda
            and all(t.is_cuda and t.dtype == torch.bfloat16 and t.data_ptr() % 8 == 0 and t.stride(-1) == 1
                    and all(st % 4 == 0 for st in t.stride()[:-1]) and t.shape[-1] % 4 == 0 for t in tensors))


LOWP = os.environ.get("MMUNET_GEMM_TOKENS_LOWP", "1") != "0"   # "0": bf16 projections stay library GEMMs (A/B)


def tokens_supported(*tensors):
    """gemm_tokens on ANY rows / inner (zero-padded weight image, masked rows): what the operands must satisfy."""
    return (ENABLED and not torch.is_autocast_enabled()
            and all(t.is_cuda and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 for t in tensors))


# False (or MMUNET_GEMM_NT=0): callers keep their batched-GEMM split-K (tests compare the two)
NT_ENABLED = os.environ.get("MMUNET_GEMM_NT", "1") != "0"
NT_LOWP = os.environ.get("MMUNET_GEMM_NT_LOWP", "1") != "0"   # "0": bf16 weight gradients stay on the slab-batched library GEMM


def nt_supported(a, b, seqlen):
    """Token-contraction product on the matrix cores (csrc/gemm_nt_splitk.hip): float32 operands with unit token
    stride, 16-byte aligned, seqlen a multiple of 32 -- or bfloat16 operands (autocast), 8-byte aligned, seqlen a
    multiple of 128."""
    if not (NT_ENABLED and a.is_cuda and a.dtype == b.dtype):
        return False
    if a.dtype == torch.bfloat16:
        return NT_LOWP and seqlen % 128 == 0 and a.data_ptr() % 8 == 0 and b.data_ptr() % 8 == 0
    return a.dtype == torch.float32 and seqlen % 32 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0


def gemm_nt(a, b, m, n, batch, seqlen, a_rs, a_bs, b_rs, b_bs, exact=False, narrow=False):
    """C (m, n) = sum over the batch * seqlen tokens of A[i][t] * B[j][t]; token (bi, l) of row i of ``a`` lies at
    element offset i * a_rs + bi * a_bs + l of its storage (same for ``b``): channel-major and batch-major operands are
    both read in place.  float32 operands, or both bfloat16 (exact products, one MFMA each); float32 result,
    deterministic."""
    _lib.require_gpu(a, b)
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("gemm_nt: two float32 or two bfloat16 tensors required")
    if seqlen % (128 if a.dtype == torch.bfloat16 else 32) != 0 or any(v % 4 != 0 for v in (a_rs, a_bs, b_rs, b_bs)):
        raise RuntimeError("gemm_nt: seqlen must be a multiple of 32 (bfloat16: 128) and the strides multiples of 4")
    L = _lib.lib()
    c = torch.empty((m, n), device=a.device, dtype=torch.float32)
    with torch.cuda.device(a.device):   # the slab count follows the CU count of the device that runs the kernel
        nws = L.mmu_gemm_nt_splitk_workspace_floats(m, n, batch, seqlen)
    ws = torch.empty(nws, device=a.device, dtype=torch.float32)
    p = _lib.GemmNtParams()
    p.m, p.n, p.batch, p.seqlen, p.exact_products, p.narrow_steps = m, n, batch, seqlen, int(exact), int(narrow)
    p.a, p.a_rs, p.a_bs = a.data_ptr(), a_rs, a_bs
    p.b, p.b_rs, p.b_bs = b.data_ptr(), b_rs, b_bs
    p.c, p.workspace = c.data_ptr(), ws.data_ptr()
    p.ab_dtype = _lib.dtype_code(a)
    with torch.cuda.device(a.device):
        _lib.check(L.mmu_gemm_nt_splitk(p, _lib.stream_of(a)))
    # inside a deferred.Scope the slab sums run later: the partials must outlive this call.  (NOT the result: an extra
    # reference to a gradient makes autograd's AccumulateGrad clone it -- i.e. read it -- on the spot)
    deferred.keep(ws)
    return c
