"""ctypes binding of libmmunet_hip.so (the C-ABI declared in include/mmunet_amd.h).

The product path has NO fallback: if the library is missing or cannot be loaded,
every op raises ``RuntimeError`` -- it never routes to a CPU implementation.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMUNET_HIP_LIB selects another build of the same library (kernel experiments, tools/ablate_scan.py)
LIB_PATH = os.environ.get("MMUNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libmmunet_hip.so")

MMU_DTYPE_F32 = 0
MMU_DTYPE_BF16 = 1

_i32, _i64, _vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p


class ScanFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "dstate", "ngroups", "dtype", "delta_softplus", "n_chunks")]
        + [(n, _vp) for n in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "out", "out_z", "x")]
        + [(n, _i64) for n in ("u_bs", "u_ds", "delta_bs", "delta_ds", "z_bs", "z_ds", "out_bs", "out_ds",
                               "out_z_bs", "out_z_ds", "A_ds", "A_ns", "B_bs", "B_gs", "B_ns", "C_bs", "C_gs",
                               "C_ns")]
    )


class ScanBwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "dstate", "ngroups", "dtype", "delta_softplus", "n_chunks")]
        + [(n, _vp) for n in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias", "dout", "x", "du", "ddelta",
                              "dA", "dB", "dC", "dD", "ddelta_bias", "dz", "out_z", "workspace")]
        + [(n, _i64) for n in ("u_bs", "u_ds", "delta_bs", "delta_ds", "z_bs", "z_ds", "dout_bs", "dout_ds",
                               "du_bs", "du_ds", "ddelta_bs", "ddelta_ds", "dz_bs", "dz_ds", "out_z_bs",
                               "out_z_ds", "A_ds", "A_ns", "B_bs", "B_gs", "B_ns", "C_bs", "C_gs", "C_ns",
                               "dB_bs", "dB_gs", "dB_ns", "dC_bs", "dC_gs", "dC_ns")]
        + [("dA_times_A", _i32), ("out", _vp), ("out_bs", _i64), ("out_ds", _i64)]
    )


class Conv1dFwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "width", "dtype", "silu")]
        + [(n, _vp) for n in ("x", "weight", "bias", "out")]
        + [(n, _i64) for n in ("x_bs", "x_ds", "out_bs", "out_ds", "w_ds", "w_ws")]
    )


class Conv1dBwdParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "seqlen", "width", "dtype", "silu")]
        + [(n, _vp) for n in ("x", "weight", "bias", "dout", "dx", "dweight", "dbias")]
        + [(n, _i64) for n in ("x_bs", "x_ds", "dout_bs", "dout_ds", "dx_bs", "dx_ds", "w_ds", "w_ws")]
        + [("workspace", _vp)]
    )


class Conv1dUpdateParams(ctypes.Structure):
    _fields_ = (
        [(n, _i32) for n in ("batch", "dim", "width", "dtype", "silu")]
        + [(n, _vp) for n in ("x", "conv_state", "weight", "bias", "out")]
        + [(n, _i64) for n in ("x_bs", "x_ds", "cs_bs", "cs_ds", "cs_ws", "out_bs", "out_ds", "w_ds", "w_ws")]
    )


class MorphParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "channels", "height", "width", "taps", "out_layout")]
                + [(n, _vp) for n in ("input", "y", "out", "dout", "dinput", "dy")]
                + [("in_dtype", _i32), ("y_parts", _i32), ("y_sum", _vp), ("dinput_addend", _vp)])


class ResizeParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("planes", "in_h", "in_w", "out_h", "out_w")]
                + [(n, _vp) for n in ("input", "out", "dout", "dinput")]
                + [("dtype", _i32), ("dinput_addend", _vp)])


class Conv3x3sParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "in_channels", "out_channels", "height", "width")]
                + [(n, _vp) for n in ("input", "weight_t", "bias", "out", "dout", "dinput", "dweight", "dbias",
                                      "workspace")]
                + [("in_dtype", _i32), ("dinput_addend", _vp), ("weight_native", _i32)])


class TriParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("rows", "seqlen", "nslices")] + [(n, _vp) for n in ("a", "flip", "slice", "out")]
                + [("dtype", _i32)])


class TriConvParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "dim", "seqlen", "nslices", "dtype")] + [("x", _vp)]
                + [(n, _i64) for n in ("x_bs", "x_ds")]
                + [(n, _vp) for n in ("weight_f", "weight_b", "weight_s", "bias_f", "bias_b", "bias_s", "out_f", "out_b",
                                      "out_s", "dout_f", "dout_b", "dout_s", "dx")]
                + [(n, _i64) for n in ("dx_bs", "dx_ds")]
                + [(n, _vp) for n in ("dweight_f", "dweight_b", "dweight_s", "dbias_f", "dbias_b", "dbias_s", "workspace")])


class TriGateParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "dim", "seqlen", "nslices", "dtype")] + [("z", _vp)]
                + [(n, _i64) for n in ("z_bs", "z_ds")] + [(n, _vp) for n in ("y_f", "y_b", "y_s", "out")]
                + [(n, _i64) for n in ("out_bs", "out_ds")] + [("dout", _vp)] + [(n, _i64) for n in ("dout_bs", "dout_ds")]
                + [("dz", _vp)] + [(n, _i64) for n in ("dz_bs", "dz_ds")] + [(n, _vp) for n in ("dy_f", "dy_b", "dy_s")])


class Stem7Params(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "height", "width")]
                + [(n, _vp) for n in ("input", "weight", "out", "dout", "dweight", "workspace")])


class NormParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "channels", "groups", "hw", "has_bn", "training", "act", "has_gn",
                                       "dinput_channel_major")]
                + [(n, ctypes.c_float) for n in ("gn_eps", "bn_eps", "momentum")]
                + [(n, _vp) for n in ("input", "gn_weight", "gn_bias", "bn_weight", "bn_bias", "pre_bias", "residual",
                                      "running_mean",
                                      "running_var", "out", "s1", "s2", "mu", "rstd", "bn_mean", "bn_rstd", "scale",
                                      "shift", "dout", "act_out", "dinput", "dresidual", "dgn_weight", "dgn_bias", "dbn_weight", "dbn_bias",
                                      "dpre_bias", "workspace")]
                + [("x_dtype", _i32), ("act_dtype", _i32)])


class Conv3x3MfmaParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "in_channels", "out_channels", "height", "width", "transposed")]
                + [(n, _vp) for n in ("input", "weight", "bias", "out", "workspace")] + [("io_dtype", _i32)])


class MorphMixParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("out_channels", _i32), ("height", _i32), ("width", _i32), ("taps", _i32),
                ("mixed", _vp), ("mixed_bs", _i64), ("mixed_cs", _i64), ("y", _vp), ("out", _vp), ("dout", _vp),
                ("dmixed", _vp), ("dy", _vp)]


class AdamWParams(ctypes.Structure):
    _fields_ = [("n_tensors", _i32), ("n_work", _i32), ("table", _vp), ("work", _vp), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_float)]


class GemmTokensParams(ctypes.Structure):
    _fields_ = [("rows", _i32), ("inner", _i32), ("tokens", _i32), ("batch", _i32), ("transposed_weight", _i32),
                ("weight", _vp), ("w_ld", _i64), ("x", _vp), ("x_rs", _i64), ("x_bs", _i64),
                ("out", _vp), ("out_rs", _i64), ("out_bs", _i64), ("workspace", _vp), ("accumulate", _i32),
                ("x_dtype", _i32), ("out_dtype", _i32)]


class DtProjParams(ctypes.Structure):
    _fields_ = [("rank", _i32), ("dim", _i32), ("tokens", _i64), ("dt", _vp), ("dt_rs", _i64), ("weight", _vp), ("w_ld", _i64),
                ("delta", _vp), ("delta_rs", _i64), ("io_dtype", _i32)]


class XProjParams(ctypes.Structure):
    _fields_ = [("rows", _i32), ("dim", _i32), ("tokens", _i64), ("x", _vp), ("x_rs", _i64), ("weight", _vp), ("w_ld", _i64),
                ("x_dbl", _vp), ("x_dbl_rs", _i64), ("io_dtype", _i32)]


class GemmNtParams(ctypes.Structure):
    _fields_ = [("m", _i32), ("n", _i32), ("batch", _i32), ("seqlen", _i32), ("exact_products", _i32), ("narrow_steps", _i32),
                ("a", _vp), ("a_rs", _i64), ("a_bs", _i64), ("b", _vp), ("b_rs", _i64), ("b_bs", _i64),
                ("c", _vp), ("workspace", _vp), ("ab_dtype", _i32)]


class CbamStatsParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("channels", _i32), ("mode", _i32), ("hw", _i64)] + \
               [(n, _vp) for n in ("input", "mean", "max", "out", "argmax", "dmean", "dmax", "dout", "dinput", "dinput_addend")]


class CbamGateParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("channels", _i32), ("hidden", _i32)] + \
               [(n, _vp) for n in ("avg", "max", "w1", "w2", "gate", "dgate", "davg", "dmax", "dw1", "dw2")]


class DiceBceParams(ctypes.Structure):
    _fields_ = [("n", _i64), ("smooth", ctypes.c_float)] + \
               [(n, _vp) for n in ("logits", "targets", "workspace", "out", "dloss", "dlogits")]


class GatedMulParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("channels", _i32), ("mode", _i32), ("hw", _i64)] + \
               [(n, _vp) for n in ("input", "gate", "out", "dout", "dinput", "dgate", "stats_dout", "stats_argmax", "input2", "addend", "dinput2", "workspace")]


class Conv7x7Params(ctypes.Structure):
    _fields_ = [(n, _i32) for n in ("batch", "height", "width")] + \
               [(n, _vp) for n in ("input", "weight", "out", "dout", "dinput", "dweight", "workspace")]


class MaxPoolParams(ctypes.Structure):
    _fields_ = [("planes", _i64)] + [(n, _i32) for n in ("height", "width", "out_height", "out_width")] + \
               [(n, _vp) for n in ("dout", "indices", "dinput", "input", "out", "codes", "dinput_addend")] + [("io_dtype", _i32)]


class SumPartsParams(ctypes.Structure):
    _fields_ = [("n", _i64), ("nparts", _i32), ("out_dtype", _i32), ("parts", _vp * 4), ("out", _vp)]


class Conv1x1OneParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("channels", _i32), ("hw", _i64)] + \
               [(n, _vp) for n in ("input", "weight", "bias", "out", "dout", "dinput", "dweight", "dbias", "workspace", "scale")]


class MambaPreParams(ctypes.Structure):
    _fields_ = [("batch", _i32), ("dim", _i32), ("seqlen", _i32), ("rows", _i32),
                ("x", _vp), ("x_bs", _i64), ("x_ds", _i64), ("conv_weight", _vp), ("conv_bias", _vp),
                ("x_proj_weight", _vp), ("dt_proj_weight", _vp), ("conv_out", _vp), ("conv_bs", _i64), ("conv_ds", _i64),
                ("x_dbl", _vp), ("delta", _vp), ("delta_bs", _i64), ("delta_ds", _i64)]


class MambaPostParams(ctypes.Structure):
    _fields_ = ([("dim", _i32), ("rows", _i32), ("tokens", _i64)]
                + [(n, _vp) for n in ("ddelta", "dt", "dx_dbl", "conv_out", "dconv_out", "x_proj_weight",
                                      "dt_proj_weight", "dx_proj_weight", "ddt_proj_weight", "workspace")])


class CoordsParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "height", "width", "taps")] + [("extend_scope", ctypes.c_float)]
                + [(n, _vp) for n in ("offset", "in_proj_weight", "out_proj_weight", "altho", "xz", "dxz", "out_z",
                                      "y", "dy", "doffset", "din_proj_weight", "dout_z", "dout_proj_weight",
                                      "daltho", "workspace")] + [("accumulate_doffset", _i32)])


class ConvS2Params(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "in_channels", "out_channels", "in_height", "in_width", "out_height",
                                     "out_width", "kernel")]
                + [(n, _vp) for n in ("input", "weight", "bias", "out", "workspace")] + [("io_dtype", _i32)])


class MambaSmallParams(ctypes.Structure):
    _fields_ = ([(n, _i32) for n in ("batch", "height", "width", "taps", "dstate", "parts")]
                + [("extend_scope", ctypes.c_float)]
                + [(n, _vp) for n in ("offset", "in_proj_weight", "conv_weight", "conv_bias", "x_proj_weight",
                                      "dt_proj_weight", "dt_bias", "A", "D", "out_proj_weight", "altho", "y",
                                      "dy", "doffset", "workspace", "dweights")])


# every symbol include/mmunet_amd.h declares (tests check that the library exports all of them)
EXPORTS = (
    "mmu_abi_version", "mmu_last_error", "mmu_scan_chunk_len", "mmu_scan_bwd_workspace_bytes",
    "mmu_selective_scan_fwd", "mmu_selective_scan_bwd", "mmu_causal_conv1d_fwd", "mmu_causal_conv1d_bwd",
    "mmu_causal_conv1d_bwd_workspace_floats", "mmu_causal_conv1d_update", "mmu_morph_sample_fwd", "mmu_morph_sample_bwd", "mmu_zigzag_inproj_fwd",
    "mmu_zigzag_inproj_bwd", "mmu_coords_outproj_fwd", "mmu_coords_outproj_bwd", "mmu_bilinear_resize_fwd",
    "mmu_bilinear_resize_bwd", "mmu_conv3x3_small_fwd_splits", "mmu_conv3x3_small_fwd", "mmu_conv3x3_small_bwd",
    "mmu_conv3x3_small_wgrad_workspace_floats",
    "mmu_tri_split", "mmu_tri_combine", "mmu_tri_conv_fwd", "mmu_tri_conv_bwd", "mmu_tri_conv_bwd_workspace_floats",
    "mmu_tri_gate_fwd", "mmu_tri_gate_bwd", "mmu_stem7_workspace_bytes", "mmu_stem7_fwd", "mmu_stem7_wgrad", "mmu_conv3x3_mfma", "mmu_conv3x3_mfma_workspace_bytes", "mmu_conv3x3_wgrad_mfma",
    "mmu_conv3x3_wgrad_mfma_workspace_floats", "mmu_gemm_tokens_mfma", "mmu_dt_proj_fwd", "mmu_dt_proj_bwd", "mmu_x_proj_fwd", "mmu_x_proj_bwd",
    "mmu_gemm_tokens_workspace_bytes", "mmu_gemm_tokens_prepare_batch", "mmu_morph_mix_sample_fwd", "mmu_morph_mix_sample_bwd", "mmu_adamw_multi", "mmu_coords_bwd_workspace_floats", "mmu_channel_sum", "mmu_scatter_stride2", "mmu_dice_bce_fwd", "mmu_dice_bce_bwd", "mmu_dice_bce_workspace_floats", "mmu_gated_mul_bwd_workspace_floats", "mmu_cbam_gate_fwd", "mmu_cbam_gate_bwd", "mmu_maxpool3s2_fwd", "mmu_maxpool3s2_bwd_codes", "mmu_deferred_begin", "mmu_deferred_pause", "mmu_deferred_end", "mmu_deferred_jobs", "mmu_deferred_job_workgroups", "mmu_deferred_launch", "mmu_gemm_nt_splitk", "mmu_gemm_nt_splitk_workspace_floats", "mmu_mamba_pre_small", "mmu_mamba_post_small",
    "mmu_mamba_post_small_workspace_floats", "mmu_norm_fused_workspace_floats", "mmu_norm_fused_fwd", "mmu_norm_fused_bwd",
    "mmu_cbam_stats_fwd", "mmu_cbam_stats_bwd", "mmu_gated_mul_fwd", "mmu_gated_mul_bwd", "mmu_conv7x7_2to1_fwd", "mmu_conv7x7_2to1_bwd", "mmu_conv7x7_2to1_workspace_floats", "mmu_maxpool3s2_bwd", "mmu_sum_parts", "mmu_conv1x1_one_fwd", "mmu_conv1x1_one_bwd", "mmu_conv1x1_one_workspace_floats",
    "mmu_mamba_small_supported", "mmu_mamba_small_parts", "mmu_mamba_small_bwd_workspace_floats",
    "mmu_mamba_small_grad_floats", "mmu_mamba_small_fwd", "mmu_mamba_small_bwd",
    "mmu_conv_s2_workspace_bytes", "mmu_conv_s2_mfma", "mmu_conv_s2_transposed_mfma",
    "mmu_conv_s2_wgrad_workspace_floats", "mmu_conv_s2_wgrad_mfma",
    "mmu_debug_wave_scan",
)

_lib = None
ABI_VERSION = 12   # = MMU_ABI_VERSION of include/mmunet_amd.h


def lib():
    """Loads the HIP library; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"mm-unet_amd: HIP library not built ({LIB_PATH}); run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or mm-unet_amd/csrc/build.sh.  There is no CPU fallback.")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. no ROCm runtime on this host
        raise RuntimeError(f"mm-unet_amd: cannot load {LIB_PATH}: {e}.  There is no CPU fallback.") from e
    L.mmu_abi_version.restype = ctypes.c_int
    if L.mmu_abi_version() != ABI_VERSION:   # struct layouts below are those of include/mmunet_amd.h at this version
        raise RuntimeError(f"mm-unet_amd: {LIB_PATH} has ABI version {L.mmu_abi_version()}, this binding was written "
                           f"against {ABI_VERSION} (include/mmunet_amd.h: MMU_ABI_VERSION); rebuild with csrc/build.sh")
    L.mmu_last_error.restype = ctypes.c_char_p
    L.mmu_scan_chunk_len.restype = ctypes.c_int
    L.mmu_scan_chunk_len.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mmu_scan_bwd_workspace_bytes.restype = ctypes.c_size_t
    L.mmu_scan_bwd_workspace_bytes.argtypes = [ctypes.c_int] * 6
    for name, st in (("mmu_selective_scan_fwd", ScanFwdParams), ("mmu_selective_scan_bwd", ScanBwdParams),
                     ("mmu_causal_conv1d_fwd", Conv1dFwdParams), ("mmu_causal_conv1d_bwd", Conv1dBwdParams),
                     ("mmu_causal_conv1d_update", Conv1dUpdateParams),
                     ("mmu_morph_sample_fwd", MorphParams), ("mmu_morph_sample_bwd", MorphParams),
                     ("mmu_zigzag_inproj_fwd", CoordsParams), ("mmu_zigzag_inproj_bwd", CoordsParams),
                     ("mmu_coords_outproj_fwd", CoordsParams), ("mmu_coords_outproj_bwd", CoordsParams),
                     ("mmu_bilinear_resize_fwd", ResizeParams), ("mmu_bilinear_resize_bwd", ResizeParams),
                     ("mmu_conv3x3_small_fwd", Conv3x3sParams), ("mmu_conv3x3_small_bwd", Conv3x3sParams),
                     ("mmu_tri_split", TriParams), ("mmu_tri_combine", TriParams),
                     ("mmu_tri_conv_fwd", TriConvParams), ("mmu_tri_conv_bwd", TriConvParams),
                     ("mmu_tri_gate_fwd", TriGateParams), ("mmu_tri_gate_bwd", TriGateParams),
                     ("mmu_stem7_fwd", Stem7Params), ("mmu_stem7_wgrad", Stem7Params),
                     ("mmu_norm_fused_fwd", NormParams), ("mmu_norm_fused_bwd", NormParams),
                     ("mmu_mamba_pre_small", MambaPreParams), ("mmu_mamba_post_small", MambaPostParams),
                     ("mmu_conv3x3_mfma", Conv3x3MfmaParams), ("mmu_conv3x3_wgrad_mfma", Conv3x3MfmaParams), ("mmu_gemm_tokens_mfma", GemmTokensParams),
                     ("mmu_gemm_nt_splitk", GemmNtParams), ("mmu_dt_proj_fwd", DtProjParams), ("mmu_dt_proj_bwd", DtProjParams),
                     ("mmu_x_proj_fwd", XProjParams), ("mmu_x_proj_bwd", XProjParams),
                     ("mmu_cbam_stats_fwd", CbamStatsParams), ("mmu_cbam_stats_bwd", CbamStatsParams),
                     ("mmu_gated_mul_fwd", GatedMulParams), ("mmu_gated_mul_bwd", GatedMulParams),
                     ("mmu_conv7x7_2to1_fwd", Conv7x7Params), ("mmu_conv7x7_2to1_bwd", Conv7x7Params),
                     ("mmu_dice_bce_fwd", DiceBceParams), ("mmu_dice_bce_bwd", DiceBceParams), ("mmu_cbam_gate_fwd", CbamGateParams), ("mmu_cbam_gate_bwd", CbamGateParams), ("mmu_maxpool3s2_bwd", MaxPoolParams), ("mmu_maxpool3s2_fwd", MaxPoolParams), ("mmu_maxpool3s2_bwd_codes", MaxPoolParams), ("mmu_sum_parts", SumPartsParams), ("mmu_conv1x1_one_fwd", Conv1x1OneParams), ("mmu_conv1x1_one_bwd", Conv1x1OneParams),
                     ("mmu_mamba_small_fwd", MambaSmallParams), ("mmu_mamba_small_bwd", MambaSmallParams),
                     ("mmu_conv_s2_mfma", ConvS2Params), ("mmu_conv_s2_transposed_mfma", ConvS2Params),
                     ("mmu_conv_s2_wgrad_mfma", ConvS2Params),
                     ("mmu_morph_mix_sample_fwd", MorphMixParams), ("mmu_morph_mix_sample_bwd", MorphMixParams),
                     ("mmu_adamw_multi", AdamWParams)):
        fn = getattr(L, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(st), _vp]
    L.mmu_causal_conv1d_bwd_workspace_floats.restype = ctypes.c_size_t
    L.mmu_causal_conv1d_bwd_workspace_floats.argtypes = [ctypes.c_int] * 3
    L.mmu_stem7_workspace_bytes.restype = ctypes.c_size_t
    L.mmu_stem7_workspace_bytes.argtypes = [ctypes.c_int] * 4
    L.mmu_tri_conv_bwd_workspace_floats.restype = ctypes.c_size_t
    L.mmu_tri_conv_bwd_workspace_floats.argtypes = [ctypes.c_int] * 4
    L.mmu_conv3x3_small_fwd_splits.restype = ctypes.c_int
    L.mmu_conv3x3_small_fwd_splits.argtypes = [ctypes.c_int] * 4
    L.mmu_conv3x3_small_wgrad_workspace_floats.restype = ctypes.c_size_t
    L.mmu_conv3x3_small_wgrad_workspace_floats.argtypes = [ctypes.c_int] * 5
    L.mmu_channel_sum.restype = ctypes.c_int
    L.mmu_channel_sum.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _vp, _vp, _vp]
    L.mmu_gated_mul_bwd_workspace_floats.restype = ctypes.c_size_t
    L.mmu_gated_mul_bwd_workspace_floats.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    L.mmu_dice_bce_workspace_floats.restype = ctypes.c_size_t
    L.mmu_dice_bce_workspace_floats.argtypes = [ctypes.c_int64]
    L.mmu_scatter_stride2.restype = ctypes.c_int
    L.mmu_scatter_stride2.argtypes = [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp]
    L.mmu_coords_bwd_workspace_floats.restype = ctypes.c_size_t
    L.mmu_coords_bwd_workspace_floats.argtypes = [ctypes.c_int] * 4
    L.mmu_deferred_begin.restype = None
    L.mmu_deferred_begin.argtypes = []
    L.mmu_deferred_pause.restype = None
    L.mmu_deferred_pause.argtypes = [ctypes.c_int]
    L.mmu_deferred_end.restype = None
    L.mmu_deferred_end.argtypes = []
    L.mmu_deferred_jobs.restype = ctypes.c_int
    L.mmu_deferred_jobs.argtypes = [_vp, ctypes.c_int]
    L.mmu_deferred_job_workgroups.restype = ctypes.c_int
    L.mmu_deferred_job_workgroups.argtypes = [_vp]
    L.mmu_deferred_launch.restype = ctypes.c_int
    L.mmu_deferred_launch.argtypes = [_vp, _vp, ctypes.c_int, _vp]
    L.mmu_gemm_tokens_prepare_batch.restype = ctypes.c_int
    L.mmu_gemm_tokens_prepare_batch.argtypes = [_vp, ctypes.c_int, ctypes.c_int64, _vp]
    L.mmu_gemm_tokens_workspace_bytes.restype = ctypes.c_size_t
    L.mmu_gemm_tokens_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mmu_conv7x7_2to1_workspace_floats.restype = ctypes.c_size_t
    L.mmu_conv7x7_2to1_workspace_floats.argtypes = [ctypes.c_int] * 3
    L.mmu_conv1x1_one_workspace_floats.restype = ctypes.c_size_t
    L.mmu_conv1x1_one_workspace_floats.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
    L.mmu_gemm_nt_splitk_workspace_floats.restype = ctypes.c_size_t
    L.mmu_gemm_nt_splitk_workspace_floats.argtypes = [ctypes.c_int] * 4
    L.mmu_conv3x3_wgrad_mfma_workspace_floats.restype = ctypes.c_size_t
    L.mmu_conv3x3_wgrad_mfma_workspace_floats.argtypes = [ctypes.c_int] * 5
    L.mmu_conv3x3_mfma_workspace_bytes.restype = ctypes.c_size_t
    L.mmu_conv3x3_mfma_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mmu_mamba_post_small_workspace_floats.restype = ctypes.c_size_t
    L.mmu_mamba_post_small_workspace_floats.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_long]
    L.mmu_norm_fused_workspace_floats.restype = ctypes.c_size_t
    L.mmu_norm_fused_workspace_floats.argtypes = [ctypes.c_int] * 3
    L.mmu_conv_s2_workspace_bytes.restype = ctypes.c_size_t
    L.mmu_conv_s2_workspace_bytes.argtypes = [ctypes.c_int] * 2
    L.mmu_conv_s2_wgrad_workspace_floats.restype = ctypes.c_size_t
    L.mmu_conv_s2_wgrad_workspace_floats.argtypes = [ctypes.c_int] * 5
    L.mmu_mamba_small_supported.restype = ctypes.c_int
    L.mmu_mamba_small_supported.argtypes = [ctypes.c_int] * 4
    L.mmu_mamba_small_parts.restype = ctypes.c_int
    L.mmu_mamba_small_parts.argtypes = [ctypes.c_int] * 6
    L.mmu_mamba_small_bwd_workspace_floats.restype = ctypes.c_size_t
    L.mmu_mamba_small_bwd_workspace_floats.argtypes = [ctypes.c_int] * 6
    L.mmu_mamba_small_grad_floats.restype = ctypes.c_size_t
    L.mmu_mamba_small_grad_floats.argtypes = [ctypes.c_int] * 2
    L.mmu_debug_wave_scan.restype = ctypes.c_int
    L.mmu_debug_wave_scan.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp]
    _lib = L
    return L


def check(status):
    if status != 0:
        raise RuntimeError(lib().mmu_last_error().decode())


def ptr(t):
    return None if t is None else t.data_ptr()


def dtype_code(t):
    if t.dtype == torch.float32:
        return MMU_DTYPE_F32
    if t.dtype == torch.bfloat16:
        return MMU_DTYPE_BF16
    raise RuntimeError(f"mm-unet_amd kernels support float32 and bfloat16 I/O, got {t.dtype}")


def stream_of(t):
    """Current HIP stream of the tensor's device (the reference launches on the current stream
    after a device guard, selective_scan.cpp:326-327)."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mm-unet_amd: expected a GPU (HIP) tensor; there is no CPU path in this package")
