"""``gemm_tokens`` -- ``W (M, K) @ X (K, T)`` for tokens-last ``X`` on the bf16 matrix cores with float32 accuracy
(csrc/gemm_tokens_mfma.hip; hi/lo bf16 split, three MFMAs per product).  float32 only, M % 64 == 0, K % 16 == 0;
callers keep their hipBLASLt path for everything else."""
import torch

from . import _lib

ENABLED = True   # False: callers use their ATen GEMMs (tests compare the two)
MIN_TILES = 192   # measured break-even against hipBLASLt on the DSC shapes (csrc/gemm_tokens_mfma.hip header)


def supported(rows, inner, tokens, *tensors):
    """Shapes the matrix-core kernel covers AND wins on: enough output tiles of 64 rows x 512 tokens to fill the
    chip (deep-K problems with few tokens would need split-K and stay with hipBLASLt)."""
    tiles = (rows // 64) * ((tokens + 511) // 512)
    return (ENABLED and rows % 64 == 0 and inner % 16 == 0 and tokens % 4 == 0 and tiles >= MIN_TILES
            and not torch.is_autocast_enabled()
            and all(t.is_cuda and t.dtype == torch.float32 and t.data_ptr() % 16 == 0 for t in tensors))


def gemm_tokens(weight, x, out, rows, inner, tokens, batch, x_rs, x_bs, out_rs, out_bs, transposed_weight=False):
    """out[b] = W . X[b]; ``weight`` is (rows, inner) -- or (inner, rows) read transposed -- with unit inner stride;
    ``x`` / ``out`` are float32 tensors whose storage holds the strided operands described by the element strides."""
    _lib.require_gpu(weight, x, out)
    if weight.dtype != torch.float32 or x.dtype != torch.float32 or out.dtype != torch.float32:
        raise RuntimeError("gemm_tokens: float32 tensors required")
    if weight.stride(-1) != 1 or weight.dim() != 2:
        raise RuntimeError("gemm_tokens: weight must be a 2-D matrix with unit column stride")
    want = (inner, rows) if transposed_weight else (rows, inner)
    if tuple(weight.shape) != want:
        raise RuntimeError(f"gemm_tokens: weight shape {tuple(weight.shape)} != {want}")
    ws = torch.empty(_lib.lib().mmu_gemm_tokens_workspace_bytes(rows, inner), device=x.device, dtype=torch.uint8)
    p = _lib.GemmTokensParams()
    p.rows, p.inner, p.tokens, p.batch, p.transposed_weight = rows, inner, tokens, batch, int(transposed_weight)
    p.weight, p.w_ld = weight.data_ptr(), weight.stride(0)
    p.x, p.x_rs, p.x_bs = x.data_ptr(), x_rs, x_bs
    p.out, p.out_rs, p.out_bs = out.data_ptr(), out_rs, out_bs
    p.workspace = ws.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_gemm_tokens_mfma(p, _lib.stream_of(x)))
    return out
