"""One switch for every fused / hand-written replacement of a plain tensor-op sequence in the model code.

``with plain_aten():`` runs ``MM_Net`` / ``Unet`` with the module calls the reference makes -- ``nn.Conv2d``, ``nn.GroupNorm``,
``nn.BatchNorm2d``, ``F.grid_sample``, ``F.interpolate``, ``flip`` / ``stack`` / ``permute``, the ``(B, L, C)`` Mamba interface --
on ATen-ROCm; only the five boundary kernels (selective scan, causal conv1d: the reference's own extension modules) stay
HIP.  It exists for A/B tests: the same weights through both routes must agree to rounding, in fp32 and -- where a wrong
low-precision kernel would otherwise hide behind bf16's own rounding -- under bf16 autocast
(tests/test_baseline_configs_gpu.py).  Not a product mode: ~3x the launches.
"""
import contextlib

from . import loss as loss_mod
from . import (conv3x3_mfma, conv3x3_small, conv_s2, mamba_simple, mamba_small_fused, maxpool, mfma_gemm, morph_coords,
               morph_mix, morph_sample, norm_fused, pointwise, resize, selective_scan_interface, stem7, tall_gemm, tri_inner, tri_order)

_FLAGS = ((conv3x3_mfma, "ENABLED"), (conv3x3_small, "ENABLED"), (conv_s2, "ENABLED"), (mamba_simple, "BCL_ENABLED"),
          (mamba_small_fused, "ENABLED"), (maxpool, "ENABLED"), (mfma_gemm, "ENABLED"), (mfma_gemm, "NT_ENABLED"),
          (morph_coords, "ENABLED"), (morph_mix, "ENABLED"), (morph_sample, "ENABLED"), (norm_fused, "ENABLED"), (pointwise, "ENABLED"),
          (pointwise, "GATED_MUL"), (pointwise, "STATS"), (pointwise, "CBAM_GATE"), (resize, "ENABLED"), (tall_gemm, "STRIDE2_ENABLED"),
          (tri_order, "ENABLED"), (tri_inner, "ENABLED"), (stem7, "ENABLED"), (selective_scan_interface, "PRE_SMALL_FUSED"),
          (selective_scan_interface, "POST_SMALL_FUSED"), (selective_scan_interface, "OWN_PROJ"), (loss_mod, "FUSED"))


@contextlib.contextmanager
def plain_aten():
    saved = [(m, n, getattr(m, n)) for m, n in _FLAGS]
    try:
        for m, n in _FLAGS:
            setattr(m, n, False)
        yield
    finally:
        for m, n, v in saved:
            setattr(m, n, v)
