"""CPU-side checks of the host layer: the C-ABI library builds/loads/exports what the header declares,
module construction reproduces the reference's weights by RNG draw order (checksums from the
reference), state_dict surface, layout helpers, and the no-fallback rule."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


def test_library_exports_every_declared_symbol():
    from mm_unet_amd import _lib
    header = open(os.path.join(ROOT, "include", "mmunet_amd.h")).read()
    declared = set(re.findall(r"\b(mmu_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"libmmunet_hip.so does not export {name}"
    lib = _lib.lib()
    assert lib.mmu_abi_version() == _lib.ABI_VERSION
    assert lib.mmu_scan_chunk_len(16, 0) == 128 and lib.mmu_scan_chunk_len(32, 0) == 256
    assert lib.mmu_scan_chunk_len(64, 0) == 128
    assert lib.mmu_scan_chunk_len(300, 0) == 0


def test_param_structs_match_header_field_order():
    from mm_unet_amd import _lib
    header = open(os.path.join(ROOT, "include", "mmunet_amd.h")).read()

    def fields(struct):
        end = header.index("} " + struct + ";")
        body = header[header.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            decl = re.sub(r"^(const\s+)?(void|float|double|int32_t|int64_t|uint8_t)\s*", "", stmt)
            names += [n.strip().lstrip("*").strip() for n in decl.split(",")]
        return names

    for struct, cls in (("mmu_scan_fwd_params", _lib.ScanFwdParams), ("mmu_scan_bwd_params", _lib.ScanBwdParams),
                        ("mmu_conv1d_fwd_params", _lib.Conv1dFwdParams),
                        ("mmu_conv1d_bwd_params", _lib.Conv1dBwdParams),
                        ("mmu_conv1d_update_params", _lib.Conv1dUpdateParams),
                        ("mmu_morph_params", _lib.MorphParams), ("mmu_coords_params", _lib.CoordsParams),
                        ("mmu_resize_params", _lib.ResizeParams), ("mmu_conv3x3s_params", _lib.Conv3x3sParams),
                        ("mmu_tri_params", _lib.TriParams), ("mmu_norm_params", _lib.NormParams),
                        ("mmu_tri_conv_params", _lib.TriConvParams), ("mmu_tri_gate_params", _lib.TriGateParams),
                        ("mmu_stem7_params", _lib.Stem7Params),
                        ("mmu_mamba_pre_params", _lib.MambaPreParams),
                        ("mmu_mamba_post_params", _lib.MambaPostParams),
                        ("mmu_conv3x3_mfma_params", _lib.Conv3x3MfmaParams),
                        ("mmu_gemm_tokens_params", _lib.GemmTokensParams),
                        ("mmu_gemm_nt_params", _lib.GemmNtParams),
                        ("mmu_dt_proj_params", _lib.DtProjParams), ("mmu_x_proj_params", _lib.XProjParams),
                        ("mmu_conv1x1_one_params", _lib.Conv1x1OneParams),
                        ("mmu_maxpool_params", _lib.MaxPoolParams),
                        ("mmu_conv7x7_params", _lib.Conv7x7Params),
                        ("mmu_gated_mul_params", _lib.GatedMulParams),
                        ("mmu_mamba_small_params", _lib.MambaSmallParams),
                        ("mmu_conv_s2_params", _lib.ConvS2Params),
                        ("mmu_morph_mix_params", _lib.MorphMixParams),
                        ("mmu_adamw_params", _lib.AdamWParams),
                        ("mmu_cbam_gate_params", _lib.CbamGateParams),
                        ("mmu_dice_bce_params", _lib.DiceBceParams)):
        # (mmu_cbam_stats_params declares two pointers per line: not parsed by this check)
        assert fields(struct) == [f[0] for f in cls._fields_], struct


@pytest.mark.parametrize("which", ["MM_Net", "Unet"])
def test_seeded_construction_reproduces_reference_weights(which):
    """torch.manual_seed(50) + construction must give the reference's tensors: same module creation
    order and initialisers (per-tensor sum / abs-sum recorded from the reference state_dict)."""
    import mm_unet_amd.mmunet as pm
    import mm_unet_amd.unet as pu
    g = golden("mmnet_64" if which == "MM_Net" else "unet_64")
    torch.manual_seed(50)
    m = pm.MM_Net(num_classes=1) if which == "MM_Net" else pu.Unet(3, 1)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(s) for s in g["ck_names"]], "state_dict keys/order differ from the reference"
    for k, s, a in zip(g["ck_names"], g["ck_sum"], g["ck_abs"]):
        v = sd[str(k)].double()
        assert abs(float(v.sum()) - float(s)) <= 1e-9 * max(1.0, abs(float(a))), k
        assert abs(float(v.abs().sum()) - float(a)) <= 1e-9 * max(1.0, abs(float(a))), k


def test_mmnet_parameter_census():
    import mm_unet_amd.mmunet as pm
    from mm_unet_amd.mamba_simple import Mamba
    m = pm.MM_Net(num_classes=1)
    assert sum(p.numel() for p in m.parameters()) == 16318391
    assert sum(isinstance(x, pm.MMConv) for x in m.modules()) == 47
    assert sum(isinstance(x, Mamba) for x in m.modules()) == 50
    g = golden("mmnet_64")
    never = set(str(s) for s in g["no_grad_names"])
    live = sum(p.numel() for k, p in m.named_parameters() if k not in never)
    assert live == int(g["n_live"]) == 9562699


def test_unet_cpu_plumbing_config1():
    """BASELINE config 1: plain Unet forward on CPU, 1x3x64x64 (pure ATen, no HIP involved)."""
    import mm_unet_amd.unet as pu
    g = golden("unet_64")
    torch.manual_seed(50)
    m = pu.Unet(3, 1).eval()
    with torch.no_grad():
        out = m(torch.from_numpy(g["x"]))
    assert out.shape == (1, 1, 64, 64)
    assert torch.allclose(out, torch.from_numpy(g["out"]), rtol=1e-4, atol=1e-4)


def test_dice_bce_loss_matches_reference():
    from mm_unet_amd.loss import DICE_BCE_Loss
    g = golden("loss_dice_bce")
    lg = torch.from_numpy(g["logits"]).requires_grad_()
    loss = DICE_BCE_Loss()(lg, torch.from_numpy(g["targets"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    loss.backward()
    assert torch.allclose(lg.grad, torch.from_numpy(g["dlogits"]), rtol=1e-5, atol=1e-7)


def test_zigzag_token_order_round_trip():
    from mm_unet_amd.mmunet import MMConv
    for h, w in ((4, 5), (5, 3), (1, 4), (2, 2)):
        x = torch.arange(2 * 3 * h * w, dtype=torch.float32).reshape(2, 3, h, w)
        f = MMConv.two_row_columnwise_flatten_grad_safe(x)
        assert f.shape == (2, 3, h * w)
        assert torch.equal(MMConv.inverse_two_row_columnwise_flatten(f, h, w), x)
    x = torch.arange(12.).reshape(1, 1, 3, 4)
    f = MMConv.two_row_columnwise_flatten_grad_safe(x)[0, 0]
    assert f.tolist() == [0, 4, 1, 5, 2, 6, 3, 7, 8, 9, 10, 11]   # pairs column-wise, odd last row appended


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing somewhere else."""
    from mm_unet_amd.mamba_simple import Mamba
    from mm_unet_amd.selective_scan_interface import selective_scan_fn
    from mm_unet_amd.causal_conv1d_interface import causal_conv1d_fn
    with pytest.raises(RuntimeError, match="no CPU path|GPU"):
        causal_conv1d_fn(torch.randn(1, 4, 8), torch.randn(4, 4), None, "silu")
    with pytest.raises(RuntimeError, match="no CPU path|GPU"):
        selective_scan_fn(torch.randn(1, 4, 8), torch.rand(1, 4, 8), -torch.rand(4, 16), torch.randn(1, 16, 8),
                          torch.randn(1, 16, 8))
    with pytest.raises(RuntimeError, match="no CPU path|GPU"):
        Mamba(8, bimamba_type="v1")(torch.randn(1, 16, 8))
    import mm_unet_amd, pathlib
    src = "\n".join(p.read_text() for p in pathlib.Path(mm_unet_amd.__path__[0]).rglob("*.py"))
    assert "import oracle" not in src and "from oracle" not in src, "product code must never import the oracle"


def test_mamba_type_resolution():
    from mm_unet_amd.mamba_simple import Mamba
    for t in ("none", "v1", "v2", "v3"):
        m = Mamba(8, bimamba_type=t)
        assert {"A_b_log", "conv1d_s.weight", "dt_proj_b.bias", "D_s"} <= set(m.state_dict().keys())
    with pytest.raises(ValueError):
        Mamba(8, bimamba_type="v9")


def test_deferred_job_workgroups_come_from_the_library():
    """mmu_deferred_job_workgroups: the block decomposition of every job kind is computed where the kernel that indexes
    by it lives (csrc/deferred_reduce.hip), no GPU needed.  Spot values: 64 results per block and 8 blocks per workgroup
    for the gemm_nt slab sums; 16 channels per workgroup for the conv1d-style sums; a workgroup per channel for the scan."""
    import ctypes
    from mm_unet_amd import _lib
    L = _lib.lib()

    def wg(row):
        buf = (ctypes.c_int64 * 8)(*row)
        return L.mmu_deferred_job_workgroups(ctypes.cast(buf, ctypes.c_void_p))

    assert wg([0, 0, 0, 0, 128 * 64, 5, 0, 0]) == 16          # 8,192 results / 64 / 8
    assert wg([0, 0, 0, 0, 36 * 128, 5, 0, 0]) == 9
    assert wg([2, 0, 0, 0, 0, 128, 0, 0]) == 8                # 128 channels / 16
    assert wg([4, 0, 0, 0, 0, 0, 128 | (7 << 32), 0]) == 128  # low word of field 6: channels
    assert wg([7, 0, 0, 0, 98, 0, 0, 0]) == 7
    assert wg([6, 0, 0, (64 << 32) | 64, 0, 2, 16, 64 | (32 << 32)]) == 36    # 2 x 1 x 64 x 32 x 9 / 1,024
    assert L.mmu_deferred_job_workgroups(None) == 0


def test_deferred_scope_rejects_results_that_are_not_parameter_gradients():
    """deferred.Scope.verify_destinations (TrainStep runs it on what the capture recorded): a deferred sum whose result
    is not inside exactly one param.grad -- a non-leaf weight's gradient, a second gradient of the same parameter --
    must stop the capture instead of training on stale gradients."""
    import torch
    from mm_unet_amd import deferred
    a, b = torch.nn.Parameter(torch.zeros(8, 4)), torch.nn.Parameter(torch.zeros(16))
    a.grad, b.grad = torch.zeros(8, 4), torch.zeros(16)
    params = [("a", a), ("b", b)]
    sc = deferred.Scope("cpu")
    pa, pb = a.grad.data_ptr(), b.grad.data_ptr()
    sc._rows = [[0, 1, pa, 0, 32, 2, 0, 0], [2, 1, pb, pb + 32, 1, 8, 1, 4], [4, 1, 0, pa + 64, 0, 1, 16, 0]]
    sc.verify_destinations(params)                                  # inside the gradients (kind 4 with a NULL dA), no repeats
    orphan = torch.zeros(8, 4)
    sc._rows = [[0, 1, orphan.data_ptr(), 0, 32, 2, 0, 0]]
    with pytest.raises(RuntimeError, match="no parameter gradient"):
        sc.verify_destinations(params)
    sc._rows = [[0, 1, pa, 0, 32, 2, 0, 0], [6, 1, pa, 0, 0, 0, 0, 0]]
    with pytest.raises(RuntimeError, match="two deferred reductions"):
        sc.verify_destinations(params)
