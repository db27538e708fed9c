"""Module-level golden fixtures from the REFERENCE modules on CPU (build container only; see
tools/ref_import.py for how the reference is imported).  Called by tools/make_golden.py.

Small modules: the reference state_dict is stored next to inputs/outputs/grads.
Large modules (MM_Net 65 MB, Unet 124 MB): weights are reproduced by RNG draw order --
``torch.manual_seed(seed)`` followed by construction -- and proven identical by per-tensor
checksums stored in the fixture (SURVEY.md section 8c "Weights").
"""
import os

import numpy as np
import torch

import ref_import
from make_golden import _np, save


def _sd(m, prefix="sd."):
    return {prefix + k: _np(v) for k, v in m.state_dict().items()}


def _checksums(m):
    names, sums, abss = [], [], []
    for k, v in m.state_dict().items():
        names.append(k)
        v = v.double()
        sums.append(float(v.sum()))
        abss.append(float(v.abs().sum()))
    return dict(ck_names=np.array(names), ck_sum=np.array(sums), ck_abs=np.array(abss))


def make_mamba(R):
    for name, d_model, btype, ns, b, l in (("v1_d3", 3, "v1", 4, 2, 300), ("v3_d64", 64, "v3", 16, 2, 256),
                                           ("v2_d8", 8, "v2", 4, 1, 128), ("none_d16", 16, "none", 4, 1, 64)):
        torch.manual_seed(7)
        m = R.Mamba(d_model=d_model, d_state=16, d_conv=4, expand=2, bimamba_type=btype, nslices=ns)
        x = torch.randn(b, l, d_model, requires_grad=True)
        out, o1, o2, o3 = m(x)
        g = torch.randn_like(out)
        out.backward(g)
        grads = {"grad." + k: _np(p.grad) for k, p in m.named_parameters() if p.grad is not None}
        extra = {}
        if btype == "v3":
            extra = dict(o_1=_np(o1), o_2=_np(o2), o_3=_np(o3))
        save("mamba_" + name, x=_np(x), out=_np(out), dout=_np(g), dx=_np(x.grad), d_model=np.array(d_model),
             nslices=np.array(ns), btype=np.array(btype), **_sd(m), **grads, **extra)


def make_mmconv(R, only=None):
    # (round 4: 19 x 19 and 38 x 38 -- the two deepest maps of the reference's own training resolution, 608 x 608
    #  (config.yml:26): neither side a power of two, 361 / 1,444 tokens)
    for name, cin, cout, k, h, w in (("c16_k3_16x16", 16, 16, 3, 16, 16), ("c16_k3_15x16", 16, 16, 3, 15, 16),
                                     ("c32to8_k1_8x8", 32, 8, 1, 8, 8), ("c16_k3_19x19", 16, 16, 3, 19, 19),
                                     ("c8_k3_38x38", 8, 8, 3, 38, 38)):
        if only is not None and name not in only:
            continue
        torch.manual_seed(3)
        m = R.MMConv(cin, cout, kernel_size=k, num_slices=4)
        m.train()
        x = torch.randn(2, cin, h, w, requires_grad=True)
        out = m(x)
        g = torch.randn_like(out)
        out.backward(g)
        grads = {"grad." + kk: _np(p.grad) for kk, p in m.named_parameters() if p.grad is not None}
        save("mmconv_" + name, x=_np(x), out=_np(out), dout=_np(g), dx=_np(x.grad),
             cfg=np.array([cin, cout, k]), **_sd(m), **grads)


GRAD_KEYS = ["encoder1.0.weight", "line_predict.weight", "rcg4.mamba.x_proj_s.weight", "rcg4.mamba.A_b_log",
             "rcg2.mamba.conv1d.weight", "rcg2.mamba.dt_proj.bias", "encoder2.0.block1.0.mamba.in_proj.weight",
             "encoder2.0.block1.0.offset_conv.weight", "encoder2.0.block1.0.altho",
             "encoder3.1.block1.0.dsc_conv_x.weight", "decoder2.conv2.0.mamba.A_log", "side5.conv2.weight",
             "rcg3.upsample.weight", "down5.0.mamba.D"]


def make_mmnet(R):
    """MM_Net at 64x64, weights = seed 50 + construction order.

    Stored: (a) eval-mode forward logits (the north-star forward-parity case); (b) fwd+bwd with Dice+BCE
    in eval mode (BatchNorm running stats) and in train mode (batch stats; Dropout2d p forced to 0 so no
    RNG is involved): logits, loss, gradients of GRAD_KEYS, |grad| sums of every live parameter.
    Because bilinear sampling at learned coordinates is only piecewise smooth (and train-mode BN sees
    8 samples per channel at the deepest stage), deep-layer gradients respond by several percent to a
    1e-6 input perturbation.  The REFERENCE'S OWN response to such a perturbation is recorded per key
    (``*sens.*``); parity tests may not ask for better agreement than the reference has with itself."""
    import io
    import contextlib
    torch.manual_seed(50)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.MM_Net(num_classes=1)
    ck = _checksums(m)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    m.eval()
    torch.manual_seed(0)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        logits = m(x)
    torch.manual_seed(1)
    xb = torch.randn(2, 3, 64, 64)
    tb = (torch.rand(2, 1, 64, 64) > 0.88).float()
    torch.manual_seed(2)
    noise = torch.randn_like(xb)

    def step(train, eps):
        m.load_state_dict(state0)  # reset BatchNorm running stats
        m.train(train)
        m.zero_grad(set_to_none=True)
        lt = m(xb + eps * noise)
        loss = R.DICE_BCE_Loss()(lt, tb)
        loss.backward()
        params = dict(m.named_parameters())
        return lt.detach(), float(loss), {k: p.grad.clone() for k, p in params.items() if p.grad is not None}

    out = {}
    for tag, train in (("train", True), ("eval", False)):
        lt, loss, g0 = step(train, 0.0)
        lt1, _, g1 = step(train, 1e-6)
        out[f"{tag}_logits"] = _np(lt)
        out[f"{tag}_loss"] = np.array(loss)
        out[f"{tag}_logits_sens"] = np.array(float((lt - lt1).abs().max()))
        for k in GRAD_KEYS:
            out[f"{tag}_grad.{k}"] = _np(g0[k])
            out[f"{tag}_sens.{k}"] = np.array(float((g0[k] - g1[k]).abs().max() / g0[k].abs().max()))
        out[f"{tag}_gabs"] = np.array([float(v.double().abs().sum()) for v in g0.values()])
        out[f"{tag}_gabs_sens"] = np.array([abs(float(a.double().abs().sum()) - float(b.double().abs().sum()))
                                            / max(float(a.double().abs().sum()), 1e-30)
                                            for a, b in zip(g0.values(), g1.values())])
        live_names = list(g0.keys())
        print(f"  MM_Net {tag}: loss {loss:.6f}  logits response to 1e-6 noise {float((lt - lt1).abs().max()):.2e}  "
              f"grad response encoder1 {float(out[tag + '_sens.encoder1.0.weight']):.2e}")
    params = dict(m.named_parameters())
    no_grad = np.array([k for k in params if k not in live_names])
    n_live = sum(params[k].numel() for k in live_names)
    save("mmnet_64", x=_np(x), logits=_np(logits), xb=_np(xb), tb=_np(tb), no_grad_names=no_grad,
         n_live=np.array(n_live), gabs_names=np.array(live_names), **ck, **out)
    print(f"  MM_Net: live grads {n_live}, never-used params {len(no_grad)}")


def make_mmnet_train128(R):
    """MM_Net in TRAIN mode on 4 x 3 x 128 x 128 (round 3): the train-mode half of ``mmnet_64`` runs 2 images at 64 x 64 -- the
    deepest BatchNorms see 2 x 2 x 2 = 8 samples per channel and the network amplifies a 1e-6 input perturbation to
    16 % on the stem gradient, so gradient checksums cannot be compared more tightly than tens of per cent there.
    Here the deepest maps are 4 x 4 with 4 images (64 samples per channel).  Stored: logits, loss, |grad| sums of every
    live parameter and the reference's OWN response of each to input noise of 1e-6 and of 1e-5 (the size of the
    bf16 hi/lo-split error of the matrix-core convolutions, 2^-16)."""
    import io
    import contextlib
    torch.manual_seed(50)
    with contextlib.redirect_stdout(io.StringIO()):
        m = R.MM_Net(num_classes=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    torch.manual_seed(11)
    xb = torch.randn(4, 3, 128, 128)
    tb = (torch.rand(4, 1, 128, 128) > 0.88).float()
    torch.manual_seed(12)
    noise = torch.randn_like(xb)

    def step(eps):
        m.load_state_dict(state0)
        m.train(True)
        m.zero_grad(set_to_none=True)
        lt = m(xb + eps * noise)
        loss = R.DICE_BCE_Loss()(lt, tb)
        loss.backward()
        return lt.detach(), float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    lt, loss, g0 = step(0.0)
    lt6, _, g6 = step(1e-6)
    lt5, _, g5 = step(1e-5)
    gabs = lambda g: np.array([float(v.double().abs().sum()) for v in g.values()])   # noqa: E731
    a0, a6, a5 = gabs(g0), gabs(g6), gabs(g5)
    save("mmnet_128_train", xb=_np(xb), tb=_np(tb), logits=_np(lt), loss=np.array(loss),
         logits_sens6=np.array(float((lt - lt6).abs().max())), logits_sens5=np.array(float((lt - lt5).abs().max())),
         gabs_names=np.array(list(g0.keys())), gabs=a0, gabs_sens6=np.abs(a6 - a0) / np.maximum(a0, 1e-30),
         gabs_sens5=np.abs(a5 - a0) / np.maximum(a0, 1e-30))
    print(f"  MM_Net train 4x128x128: loss {loss:.6f}; logits response to 1e-6 / 1e-5 noise "
          f"{float((lt - lt6).abs().max()):.2e} / {float((lt - lt5).abs().max()):.2e}; |grad|-sum response median "
          f"{np.median(np.abs(a6 - a0) / np.maximum(a0, 1e-30)):.2e} / {np.median(np.abs(a5 - a0) / np.maximum(a0, 1e-30)):.2e}, "
          f"max {np.max(np.abs(a6 - a0) / np.maximum(a0, 1e-30)):.2e} / {np.max(np.abs(a5 - a0) / np.maximum(a0, 1e-30)):.2e}")


def make_unet(R):
    torch.manual_seed(50)
    m = R.Unet(3, 1)
    ck = _checksums(m)
    m.eval()
    torch.manual_seed(0)
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        out = m(x)
    m.train()
    torch.manual_seed(1)
    xb = torch.randn(2, 3, 64, 64)
    tb = (torch.rand(2, 1, 64, 64) > 0.88).float()
    lt = m(xb)
    loss = R.DICE_BCE_Loss()(lt, tb)
    loss.backward()
    p = dict(m.named_parameters())
    save("unet_64", x=_np(x), out=_np(out), xb=_np(xb), tb=_np(tb), train_logits=_np(lt),
         loss=np.array(float(loss)), **ck,
         **{"grad.inc.conv.0.weight": _np(p["inc.conv.0.weight"].grad),
            "grad.outc.conv.weight": _np(p["outc.conv.weight"].grad),
            "grad.up1.up.weight_abs": np.array(float(p["up1.up.weight"].grad.double().abs().sum()))})


def make_loss(R):
    torch.manual_seed(2)
    logits = torch.randn(2, 1, 32, 32, requires_grad=True)
    t = (torch.rand(2, 1, 32, 32) > 0.8).float()
    loss = R.DICE_BCE_Loss()(logits, t)
    loss.backward()
    save("loss_dice_bce", logits=_np(logits), targets=_np(t), loss=np.array(float(loss)), dlogits=_np(logits.grad))


def make_blocks(R):
    """Block-level fixtures of the reference's ResidualBlock (both forms), DecoderBlock, SideoutBlock, CBAM and RCG
    (MMUNet.py:313-467): train mode (BatchNorm batch statistics), Dropout2d p forced to 0, output + input gradients
    + every parameter gradient.  The reference's own response to a 1e-6 input perturbation is stored with each
    fixture (``sens_out`` / ``sens_grad``): a parity test may not ask for better agreement than that."""
    def run(name, build, inputs):
        torch.manual_seed(11)
        m = build()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout2d):
                mod.p = 0.0
        m.train()
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        torch.manual_seed(12)
        xs = [torch.randn(*shp) for shp in inputs]
        noise = [torch.randn(*shp) for shp in inputs]
        g = None

        def step(eps):
            nonlocal g
            m.load_state_dict(state0)
            m.zero_grad(set_to_none=True)
            ins = [(x + eps * nz).requires_grad_(True) for x, nz in zip(xs, noise)]
            out = m(*ins)
            if g is None:
                g = torch.randn_like(out)
            out.backward(g)
            return out.detach(), [i.grad for i in ins], {k: p.grad.clone() for k, p in m.named_parameters()
                                                         if p.grad is not None}
        out, dins, grads = step(0.0)
        out1, dins1, grads1 = step(1e-6)
        sens_out = float((out - out1).abs().max())
        # per-tensor response (absolute): gradients that are analytically zero (a GroupNorm bias in front of a
        # train-mode BatchNorm) are pure rounding noise and respond by 100 % of themselves
        sens = {"sens." + k: np.array(float((grads[k] - grads1[k]).abs().max())) for k in grads}
        sens.update({f"sens.din{i}": np.array(float((a - b).abs().max())) for i, (a, b) in enumerate(zip(dins, dins1))})
        sens_grad = max(float((grads[k] - grads1[k]).abs().max() / max(float(grads[k].abs().max()), 1e-30))
                        for k in grads if float(grads[k].abs().max()) > 1e-3)
        arrs = {f"in{i}": _np(x) for i, x in enumerate(xs)}
        arrs.update({f"din{i}": _np(d) for i, d in enumerate(dins)})
        arrs.update({"grad." + k: _np(v) for k, v in grads.items()})
        arrs.update(sens)
        sd = {"sd." + k: _np(v) for k, v in state0.items()}
        save("block_" + name, out=_np(out), dout=_np(g), sens_out=np.array(sens_out), sens_grad=np.array(sens_grad),
             **arrs, **sd)
        print(f"    {name}: response of out / grads to 1e-6 input noise {sens_out:.2e} / {sens_grad:.2e}")

    run("residual_32", lambda: R.ResidualBlock(32, 32, 4, downsample=False), [(2, 32, 16, 16)])
    run("residual_down_32to64", lambda: R.ResidualBlock(32, 64, 4, downsample=True), [(2, 32, 16, 16)])
    run("decoder_64to32", lambda: R.DecoderBlock(64, 32, num_slices=4), [(2, 64, 8, 8)])
    run("sideout_64", lambda: R.SideoutBlock(64, 1, num_slices=4), [(2, 64, 16, 16)])
    run("cbam_64", lambda: R.CBAM(64), [(2, 64, 16, 16)])
    run("rcg_ns4", lambda: R.RCG(num_slices=4), [(2, 1, 8, 8), (2, 64, 16, 16), (2, 64, 8, 8)])


def main():
    R = ref_import.load_reference_model()

    print("Mamba fixtures (reference mamba_simple.Mamba on CPU)")
    make_mamba(R)
    print("MMConv fixtures")
    make_mmconv(R)
    print("block fixtures")
    make_blocks(R)
    print("loss fixture")
    make_loss(R)
    print("Unet fixture")
    make_unet(R)
    print("MM_Net fixture")
    make_mmnet(R)
