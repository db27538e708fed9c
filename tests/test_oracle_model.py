"""Pins the model-level CPU oracle (oracle/model_ref.py) to fixtures produced by the REFERENCE's own
modules (tools/make_golden_modules.py): Mamba uni/bi/tri-directional, MMConv, MM_Net (eval logits,
train-mode loss + gradients), Unet, Dice+BCE.  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import model_ref as M


def _sd(g, grad=True):
    sd = {}
    for k, v in g.items():
        if k.startswith("sd."):
            t = torch.from_numpy(v.copy())
            if grad and t.is_floating_point() and "running_" not in k:
                t.requires_grad_()
            sd[k[3:]] = t
    return sd


def close(a, b, rtol, atol, what):
    a, b = torch.as_tensor(a).detach().float(), torch.as_tensor(b).detach().float()
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (ref max {b.abs().max():.3e})"


@pytest.mark.parametrize("name", ["mamba_v1_d3", "mamba_v3_d64", "mamba_v2_d8", "mamba_none_d16"])
def test_mamba_oracle_vs_reference(name):
    g = golden(name)
    sd = _sd(g)
    x = torch.from_numpy(g["x"]).requires_grad_()
    out, o1, o2, o3 = M.mamba(M.P(sd), x, str(g["btype"]), int(g["nslices"]))
    close(out, g["out"], 1e-4, 1e-4, "out")
    if "o_1" in g:
        close(o1, g["o_1"], 1e-4, 1e-4, "o_1")
        close(o2, g["o_2"], 1e-4, 1e-4, "o_2")
        close(o3, g["o_3"], 1e-4, 1e-4, "o_3")
    out.backward(torch.from_numpy(g["dout"]))
    close(x.grad, g["dx"], 1e-3, 1e-4, "dx")
    for k in g:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, g[k], 2e-3, 2e-3, k)


# (19 x 19 and 38 x 38: the deepest maps of the reference's own training resolution, 608 x 608 -- config.yml:26)
@pytest.mark.parametrize("name", ["mmconv_c16_k3_16x16", "mmconv_c16_k3_15x16", "mmconv_c32to8_k1_8x8",
                                  "mmconv_c16_k3_19x19", "mmconv_c8_k3_38x38"])
def test_mmconv_oracle_vs_reference(name):
    g = golden(name)
    sd = _sd(g)
    cin, cout, k = (int(v) for v in g["cfg"])
    x = torch.from_numpy(g["x"]).requires_grad_()
    out = M.mmconv(M.P(sd), x, k, cout, 4)
    close(out, g["out"], 1e-4, 1e-4, "out")
    out.backward(torch.from_numpy(g["dout"]))
    close(x.grad, g["dx"], 1e-3, 2e-4, "dx")
    for kk in g:
        if kk.startswith("grad."):
            close(sd[kk[5:]].grad, g[kk], 2e-3, 2e-3, kk)


BLOCK_FNS = {
    "block_residual_32": lambda p, ins, cx: M.residual_block(p, ins[0], 32, 32, False, cx, 4),
    "block_residual_down_32to64": lambda p, ins, cx: M.residual_block(p, ins[0], 32, 64, True, cx, 4),
    "block_decoder_64to32": lambda p, ins, cx: M.decoder_block(p, ins[0], 64, 32, cx, 4),
    "block_sideout_64": lambda p, ins, cx: M.sideout(p, ins[0], cx, 4),
    "block_cbam_64": lambda p, ins, cx: M.cbam(p, ins[0]),
    "block_rcg_ns4": lambda p, ins, cx: M.rcg(p, ins[0], ins[1], ins[2], cx, 4),
}


@pytest.mark.parametrize("name", sorted(BLOCK_FNS))
def test_block_oracle_vs_reference(name):
    """ResidualBlock / DecoderBlock / SideoutBlock / CBAM / RCG of the oracle against the reference's own modules
    (train mode, every parameter gradient; tolerances as in tests/test_modules_gpu.py::test_block_vs_reference)."""
    g = golden(name)
    sd = _sd(g)
    ins = []
    while f"in{len(ins)}" in g:
        ins.append(torch.from_numpy(g[f"in{len(ins)}"]).requires_grad_())
    out = BLOCK_FNS[name](M.P(sd), ins, M.Ctx(True))
    close(out, g["out"], 1e-4, max(2e-4, 4 * float(g["sens_out"])), "out")
    out.backward(torch.from_numpy(g["dout"]))

    def check(t, ref, sens, what):
        scale = float(np.abs(ref).max())
        close(t, ref, 2e-3, max(2e-3 * scale, 4 * float(sens), 2e-6), what)

    for j, x in enumerate(ins):
        check(x.grad, g[f"din{j}"], g[f"sens.din{j}"], f"din{j}")
    for kk in g:
        if kk.startswith("grad."):
            check(sd[kk[5:]].grad, g[kk], g["sens." + kk[5:]], kk)


def _seeded_state(cls_name):
    """Weights by RNG draw order: seed 50 then construct (identity with the reference proven by the
    checksum test in test_host_logic.py)."""
    import mm_unet_amd.mmunet as pm
    import mm_unet_amd.unet as pu
    torch.manual_seed(50)
    m = pm.MM_Net(num_classes=1) if cls_name == "MM_Net" else pu.Unet(3, 1)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def test_mmnet_oracle_eval_logits_vs_reference():
    g = golden("mmnet_64")
    sd = _seeded_state("MM_Net")
    with torch.no_grad():
        logits = M.mm_net(sd, torch.from_numpy(g["x"]), training=False)
    close(logits, g["logits"], 1e-3, 1e-3, "logits")


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_mmnet_oracle_fwd_bwd_vs_reference(mode):
    """Dice+BCE fwd+bwd.  Tolerances are tied to the reference's OWN response to a 1e-6 input
    perturbation stored in the fixture (``*_sens``): bilinear sampling at learned coordinates makes the
    deep gradients only piecewise smooth, and train-mode BatchNorm sees 8 samples per channel at the
    deepest stage, so two correct fp32 implementations cannot agree better than that."""
    g = golden("mmnet_64")
    sd = _seeded_state("MM_Net")
    for k, v in sd.items():
        if v.is_floating_point() and "running_" not in k:
            v.requires_grad_()
    lt = M.mm_net(sd, torch.from_numpy(g["xb"]), training=(mode == "train"))
    lsens = float(g[f"{mode}_logits_sens"])
    close(lt, g[f"{mode}_logits"], 1e-3, max(1e-3, 4 * lsens), f"{mode} logits")
    loss = M.dice_bce_loss(lt, torch.from_numpy(g["tb"]))
    assert abs(float(loss) - float(g[f"{mode}_loss"])) < max(1e-4, lsens)
    loss.backward()
    never = set(str(s) for s in g["no_grad_names"])
    for k, v in sd.items():
        if v.requires_grad:
            assert (v.grad is None) == (k in never), f"live/never-used mismatch for {k}"
    pre = f"{mode}_grad."
    for k in g:
        if k.startswith(pre):
            name = k[len(pre):]
            ref = torch.from_numpy(g[k])
            tol = max(2e-3, 4 * float(g[f"{mode}_sens.{name}"]))
            close(sd[name].grad, ref, tol, tol * float(ref.abs().max()), k)


def test_unet_and_loss_oracle_vs_reference():
    g = golden("unet_64")
    sd = _seeded_state("Unet")
    with torch.no_grad():
        out = M.unet(sd, torch.from_numpy(g["x"]), training=False)
    close(out, g["out"], 1e-4, 1e-4, "unet eval")
    gl = golden("loss_dice_bce")
    lg = torch.from_numpy(gl["logits"]).requires_grad_()
    loss = M.dice_bce_loss(lg, torch.from_numpy(gl["targets"]))
    assert abs(float(loss) - float(gl["loss"])) < 1e-6
    loss.backward()
    close(lg.grad, gl["dlogits"], 1e-5, 1e-7, "dlogits")
