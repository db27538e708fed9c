"""CPU ORACLE, model level -- test infrastructure, not the product.

Functional restatement (plain torch-CPU ops + the C oracle for scan / conv1d) of the reference's
model code, evaluated directly on a ``state_dict``-style mapping ``{name: tensor}`` with the
reference's parameter names.  It shares no code with ``mm-unet_amd/`` (which is nn.Module based
and GPU-only), so agreement between the two is an independent check.

Follows:
    mamba_inner (conv1d -> x_proj -> dt_proj -> scan [-> out_proj])  selective_scan_interface.py:636-670
    Mamba.forward uni / bi / tri-directional                        requirements/mamba_simple.py:185-362
    MMConv.forward + coordinate map + zig-zag token order           src/UM_Net/MMUNet.py:68-274
    CBAM / SideoutBlock / RCG / DecoderBlock / ResidualBlock        src/UM_Net/MMUNet.py:313-467
    MM_Net.forward                                                  src/UM_Net/MMUNet.py:532-585
    Unet.forward                                                    model.py:71-85
    DICE_BCE_Loss.forward                                           loss.py:10-19

Pinned by tests/golden/{mamba_*,mmconv_*,mmnet_64,unet_64,loss_dice_bce}.npz, which were produced by
the reference's own modules (tools/make_golden_modules.py) -- see tests/test_oracle_model.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F

import oracle


class P:
    """Prefix view over a flat parameter mapping."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def get(self, k):
        return self.sd.get(self.prefix + k)

    def sub(self, p):
        return P(self.sd, self.prefix + p + ".")


# ----------------------------------------------------------------------------- Mamba
def _inner(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, dt_bias):
    """out_z of one direction: selective_scan_interface.py:642-669."""
    batch, two_d, L = xz.shape
    d = two_d // 2
    r = dt_proj_w.shape[1]
    n = A.shape[1]
    x, z = xz[:, :d], xz[:, d:]
    x = oracle.causal_conv1d(x, conv_w.reshape(d, -1), conv_b, "silu")
    x_dbl = x.permute(0, 2, 1).reshape(batch * L, d) @ x_proj_w.t()
    delta = (dt_proj_w @ x_dbl[:, :r].t()).reshape(d, batch, L).permute(1, 0, 2)
    Bm = x_dbl[:, r:r + n].reshape(batch, L, n).permute(0, 2, 1)
    Cm = x_dbl[:, r + n:].reshape(batch, L, n).permute(0, 2, 1)
    return oracle.selective_scan(x, delta, A, Bm, Cm, D, z, dt_bias, True)


def mamba(p, h, bimamba_type, nslices):
    """h: (B, L, d_model) -> (out, o_1, o_2, o_3); mamba_simple.py:185-362 with the v1/v3 resolution of
    SURVEY.md section 8a-8."""
    batch, L, dm = h.shape
    xz = (p["in_proj.weight"] @ h.reshape(batch * L, dm).t()).reshape(-1, batch, L).permute(1, 0, 2)
    d = xz.shape[1] // 2

    def branch(xz_, sfx, a_name):
        return _inner(xz_, p[f"conv1d{sfx}.weight"], p[f"conv1d{sfx}.bias"], p[f"x_proj{sfx}.weight"],
                      p[f"dt_proj{sfx}.weight"], -torch.exp(p[a_name].to(torch.get_default_dtype())), p[f"D{sfx}"].to(torch.get_default_dtype()),
                      p[f"dt_proj{sfx}.bias"].to(torch.get_default_dtype()))

    wo, bo = p["out_proj.weight"], p.get("out_proj.bias")
    if bimamba_type == "v3":
        o1 = branch(xz, "", "A_log")
        o2 = branch(xz.flip([-1]), "_b", "A_b_log")
        xs = torch.stack(xz.chunk(nslices, dim=-1), dim=-1).flatten(-2)
        o3 = branch(xs, "_s", "A_s_log")
        o3 = o3.reshape(batch, d, L // nslices, nslices).permute(0, 1, 3, 2).flatten(-2)
        return F.linear((o1 + o2.flip([-1]) + o3).permute(0, 2, 1), wo, bo), o1, o2, o3
    if bimamba_type == "v2":
        o1 = branch(xz, "", "A_log")
        o2 = branch(xz.flip([-1]), "_b", "A_b_log")
        return F.linear((o1 + o2.flip([-1])).permute(0, 2, 1), wo, bo), None, None, None
    o1 = branch(xz, "", "A_log")
    return F.linear(o1.permute(0, 2, 1), wo, bo), None, None, None


# ----------------------------------------------------------------------------- MMConv
def _zigzag(x):
    """MMUNet.py:68-93."""
    B, C, H, W = x.shape
    He = H // 2 * 2
    out = x[:, :, :He].reshape(B, C, He // 2, 2, W).transpose(-1, -2).reshape(B, C, -1)
    if H % 2:
        out = torch.cat([out, x[:, :, He:].reshape(B, C, -1)], 2)
    return out


def _unzigzag(f, H, W):
    """MMUNet.py:95-121."""
    B, C, _ = f.shape
    He = H // 2 * 2
    out = f[:, :, :He * W].reshape(B, C, He // 2, W, 2).transpose(-1, -2).reshape(B, C, He, W)
    if H % 2:
        out = torch.cat([out, f[:, :, He * W:].reshape(B, C, 1, W)], 2)
    return out


def mmconv(p, x, K, cout, nslices=4, extend_scope=1.0):
    """MMUNet.py:245-274 (morph = 0)."""
    B, cin, H, W = x.shape
    off = torch.tanh(F.group_norm(F.conv2d(x, p["offset_conv.weight"], p["offset_conv.bias"], padding=1), K,
                                  p["gn_offset.weight"], p["gn_offset.bias"]))
    y_off = off[:, :K]
    c = K // 2
    rows = torch.arange(H, dtype=torch.get_default_dtype()).view(1, 1, H, 1)
    cols = torch.arange(W, dtype=torch.get_default_dtype()).view(1, 1, 1, W)
    # MMUNet.py:156-172: the sums are written into a detached clone but are built from the live offsets
    cum = [None] * K
    cum[c] = torch.zeros(B, H, W)
    for i in range(1, c + 1):
        cum[c + i] = cum[c + i - 1] + y_off[:, c + i]
        cum[c - i] = cum[c - i + 1] + y_off[:, c - i]
    y_new = rows + extend_scope * torch.stack(cum, 1)
    x_new = (cols + torch.linspace(-c, c, K).view(1, K, 1, 1)).expand(B, K, H, W)
    seq, _, _, _ = mamba(p.sub("mamba"), _zigzag(y_off).transpose(1, 2), "v1", nslices)
    y_keep = _unzigzag(seq.transpose(1, 2), H, W)
    wgt = torch.clamp(F.softplus(p["altho"]), min=0.01)
    y = wgt * y_keep + y_new
    ymap = y.permute(0, 2, 1, 3).reshape(B, H * K, W)      # "b k w h -> b (w k) h"
    xmap = x_new.permute(0, 2, 1, 3).reshape(B, H * K, W)
    ys = -1 + 2.0 / (H - 1) * torch.clamp(ymap, 0, H - 1)  # MMUNet.py:229-242
    xs = -1 + 2.0 / (W - 1) * torch.clamp(xmap, 0, W - 1)
    deformed = F.grid_sample(x, torch.stack([xs, ys], -1), mode="bilinear", padding_mode="zeros",
                             align_corners=True)
    out = F.conv2d(deformed, p["dsc_conv_x.weight"], p["dsc_conv_x.bias"], stride=(K, 1))
    return F.group_norm(out, cout // 4, p["gn.weight"], p["gn.bias"])


# ----------------------------------------------------------------------------- blocks
class Ctx:
    """training flag + the BatchNorm buffers to update (functional batch_norm mutates them in place,
    like nn.BatchNorm2d in train mode)."""

    def __init__(self, training):
        self.training = training


def bn(p, x, cx):
    return F.batch_norm(x, p["running_mean"], p["running_var"], p["weight"], p["bias"], training=cx.training,
                        momentum=0.1, eps=1e-5)


def mmconv_bn_relu(p, x, K, cout, cx, ns):
    """nn.Sequential(MMConv, BatchNorm2d, ReLU) -- e.g. MMUNet.py:344,358,423."""
    return F.relu(bn(p.sub("1"), mmconv(p.sub("0"), x, K, cout, ns), cx))


def residual_block(p, x, cin, cout, down, cx, ns):
    """MMUNet.py:433-467."""
    b1 = p.sub("block1")
    if down:
        t = F.conv2d(x, b1["0.weight"], None, stride=2, padding=1)
        t = F.relu(bn(b1.sub("1"), t, cx))
        t = bn(b1.sub("4"), mmconv(b1.sub("3"), t, 3, cout, ns), cx)
        b2 = p.sub("block2")
        s = bn(b2.sub("1"), F.conv2d(x, b2["0.weight"], None, stride=2), cx)
        return F.relu(s + t)
    t = F.relu(bn(b1.sub("1"), mmconv(b1.sub("0"), x, 3, cout, ns), cx))
    t = bn(b1.sub("4"), mmconv(b1.sub("3"), t, 3, cout, ns), cx)
    return F.relu(t + x)


def decoder_block(p, x, cin, cout, cx, ns):
    """MMUNet.py:420-431."""
    t = mmconv_bn_relu(p.sub("conv1"), x, 3, cin // 4, cx, ns)
    t = mmconv_bn_relu(p.sub("conv2"), t, 3, cout, cx, ns)
    return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)


def sideout(p, x, cx, ns):
    """MMUNet.py:341-352; Dropout2d is identity here (eval, or p forced to 0 in the train fixture)."""
    t = mmconv_bn_relu(p.sub("conv1"), x, 3, 16, cx, ns)
    return F.conv2d(t, p["conv2.weight"], p["conv2.bias"])


def cbam(p, x):
    """MMUNet.py:313-338."""
    def mlp(v):
        return F.conv2d(F.relu(F.conv2d(v, p["mlp.0.weight"])), p["mlp.2.weight"])
    c = torch.sigmoid(mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1)))
    y1 = c * x
    s = torch.cat((y1.max(dim=1, keepdim=True)[0], y1.mean(dim=1, keepdim=True)), 1)
    return torch.sigmoid(F.conv2d(s, p["conv.weight"], padding=3)) * y1


def rcg(p, pre, edge, f, cx, ns):
    """MMUNet.py:389-418."""
    r = (-1 * torch.sigmoid(pre) + 1) * f
    e1 = F.interpolate(edge, size=f.shape[2:], mode="bilinear", align_corners=True)
    x2 = mmconv_bn_relu(p.sub("conv1"), torch.cat((e1, r), 1), 3, 64, cx, ns)
    x0 = F.conv_transpose2d(x2, p["upsample.weight"], p["upsample.bias"], stride=2, padding=1)
    B, C, H, W = x0.shape
    out, _, _, _ = mamba(p.sub("mamba"), x0.reshape(B, C, H * W).transpose(1, 2), "v3", ns)
    x0 = F.conv2d(out.transpose(1, 2).reshape(B, C, H, W), p["downsample.weight"], p["downsample.bias"], stride=2,
                  padding=1)
    x3 = torch.sigmoid(F.conv2d(x2, p["mlp.0.weight"], p["mlp.0.bias"]))
    return x0 * x3 * x2 + f


def mm_net(sd, x, training=False, num_slices_list=(64, 32, 16, 8)):
    """MM_Net.forward, MMUNet.py:532-585."""
    p, cx, s = P(sd), Ctx(training), num_slices_list
    e1 = F.relu(bn(p.sub("encoder1.1"), F.conv2d(x, p["encoder1.0.weight"], None, stride=2, padding=3), cx))
    t = F.max_pool2d(e1, 3, 2, 1)
    plan = (("encoder2", [(64, 64, False)] * 3, s[0]),
            ("encoder3", [(64, 128, True)] + [(128, 128, False)] * 3, s[1]),
            ("encoder4", [(128, 256, True)] + [(256, 256, False)] * 5, s[2]),
            ("encoder5", [(256, 512, True)] + [(512, 512, False)] * 2, s[3]))
    feats = []
    for name, blocks, ns in plan:
        for i, (ci, co, dn) in enumerate(blocks):
            t = residual_block(p.sub(f"{name}.{i}"), t, ci, co, dn, cx, ns)
        feats.append(t)
    e2, e3, e4, e5 = feats
    e3 = mmconv_bn_relu(p.sub("down3"), e3, 1, 64, cx, s[-1])
    e4 = mmconv_bn_relu(p.sub("down4"), e4, 1, 64, cx, s[-1])
    e5 = mmconv_bn_relu(p.sub("down5"), e5, 1, 64, cx, s[-1])
    d5 = decoder_block(p.sub("decoder5"), e5, 64, 64, cx, s[3])
    out5 = sideout(p.sub("side5"), d5, cx, s[3])
    cb = p.sub("cbam")
    c1 = F.relu(bn(cb.sub("1"), F.conv2d(e1, cb["0.weight"], cb["0.bias"], padding=1), cx))
    c1 = cbam(cb.sub("3"), c1)
    c1 = F.relu(bn(cb.sub("5"), F.conv2d(c1, cb["4.weight"], cb["4.bias"], padding=1), cx))
    p_c = F.conv2d(c1, p["line_predict.weight"], p["line_predict.bias"], padding=1)
    r4 = rcg(p.sub("rcg4"), out5, c1, e4, cx, s[2])
    d4 = decoder_block(p.sub("decoder4"), torch.cat((d5, r4), 1), 128, 64, cx, s[2])
    out4 = sideout(p.sub("side4"), d4, cx, s[2])
    r3 = rcg(p.sub("rcg3"), out4, c1, e3, cx, s[1])
    d3 = decoder_block(p.sub("decoder3"), torch.cat((d4, r3), 1), 128, 64, cx, s[1])
    out3 = sideout(p.sub("side3"), d3, cx, s[1])
    r2 = rcg(p.sub("rcg2"), out3, c1, e2, cx, s[0])
    d2 = decoder_block(p.sub("decoder2"), torch.cat((d3, r2), 1), 128, 64, cx, s[0])
    out2 = sideout(p.sub("side2"), d2, cx, s[0])
    size = x.shape[2:]
    up = lambda v: F.interpolate(v, size=size, mode="bilinear", align_corners=True)  # noqa: E731
    return up(out2) + up(out3) + up(out4) + up(out5) + up(p_c)


# ----------------------------------------------------------------------------- plain U-Net, loss
def _inconv(p, x, cx):
    c = p.sub("conv")
    x = F.relu(bn(c.sub("1"), F.conv2d(x, c["0.weight"], c["0.bias"], padding=1), cx))
    return F.relu(bn(c.sub("4"), F.conv2d(x, c["3.weight"], c["3.bias"], padding=1), cx))


def unet(sd, x, training=False):
    """model.py:71-85."""
    p, cx = P(sd), Ctx(training)
    x1 = _inconv(p.sub("inc"), x, cx)
    xs = [x1]
    for i in range(1, 5):
        xs.append(_inconv(p.sub(f"down{i}.down.1"), F.max_pool2d(xs[-1], 2), cx))
    t = xs[4]
    for i, skip in zip(range(1, 5), (xs[3], xs[2], xs[1], xs[0])):
        u = p.sub(f"up{i}")
        t = F.conv_transpose2d(t, u["up.weight"], u["up.bias"], stride=2)
        dy, dx = skip.shape[2] - t.shape[2], skip.shape[3] - t.shape[3]
        t = F.pad(t, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        t = _inconv(u.sub("conv"), torch.cat([skip, t], 1), cx)
    return F.conv2d(t, p["outc.conv.weight"], p["outc.conv.bias"])


def dice_bce_loss(logits, targets, smooth=1):
    """loss.py:10-19."""
    pr = torch.sigmoid(logits)
    inter = 2 * (pr * targets).sum() + smooth
    union = (pr + targets).sum() + smooth
    return 1.0 - inter / union + F.binary_cross_entropy(pr, targets)
