"""Import shim for the READ-ONLY reference at /root/reference (build container only).

Used by tools/make_golden.py to produce tests/golden/*.npz and by nothing else.
Never shipped, never imported by the product or by tests (the reference does not
exist on the GPU box).  Recipe follows SURVEY.md section 8c:

  * empty ``types.ModuleType`` stubs for the two CUDA extension modules that the
    reference imports at module top (selective_scan_interface.py:9-11,
    causal_conv1d_interface.py:7);
  * bare package shells for ``mamba_ssm`` / ``mamba_ssm.ops`` with ``__path__`` set
    so that ``mamba_ssm/__init__.py`` (LM-generation imports) is not executed;
  * stubs for ``torchvision`` / ``timm`` (imported, never used: MMUNet.py:3,5);
  * ``mamba_simple.py``'s fused-op globals replaced by CPU compositions of the
    reference's own ``causal_conv1d_ref`` + ``F.linear`` + ``selective_scan_ref``
    (the steps of mamba_inner_ref, selective_scan_interface.py:636-670);
  * a ``Mamba`` subclass that passes the ``assert bimamba_type == "v3"``
    (mamba_simple.py:125) at construction, then restores the requested type and,
    for non-v3 types, returns ``(out, None, None, None)`` (mamba_simple.py:362 would
    raise UnboundLocalError).
"""
import importlib.util
import os
import sys
import types

REF = "/root/reference"
MAMBA = os.path.join(REF, "requirements/Mamba/mamba")
CC1D = os.path.join(REF, "requirements/Mamba/causal-conv1d")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_leaf_refs():
    """Returns (selective_scan_ref, causal_conv1d_ref) from the reference."""
    _stub("causal_conv1d_cuda")
    _stub("selective_scan_cuda")
    if CC1D not in sys.path:
        sys.path.insert(0, CC1D)
    pkg = _stub("mamba_ssm")
    pkg.__path__ = [os.path.join(MAMBA, "mamba_ssm")]
    ops = _stub("mamba_ssm.ops")
    ops.__path__ = [os.path.join(MAMBA, "mamba_ssm", "ops")]
    import torch.cuda.amp  # noqa: F401  (custom_fwd/custom_bwd used as decorators)
    from causal_conv1d.causal_conv1d_interface import causal_conv1d_ref
    from mamba_ssm.ops.selective_scan_interface import selective_scan_ref
    return selective_scan_ref, causal_conv1d_ref


def load_reference_model():
    """Returns a namespace with the reference's Mamba (patched as documented above),
    MMConv, RCG, MM_Net, Unet, DICE_BCE_Loss -- all running on CPU."""
    import torch
    import torch.nn.functional as F
    from einops import rearrange

    selective_scan_ref, causal_conv1d_ref = load_leaf_refs()
    _stub("torchvision").models = _stub("torchvision.models")
    _stub("timm")

    def _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias):
        # steps of mamba_inner_ref (selective_scan_interface.py:642-669) up to out_z
        L = xz.shape[-1]
        delta_rank = delta_proj_weight.shape[1]
        d_state = A.shape[-1]
        x, z = xz.chunk(2, dim=1)
        x = causal_conv1d_ref(x, rearrange(conv1d_weight, "d 1 w -> d w"), conv1d_bias, "silu")
        x_dbl = F.linear(rearrange(x, "b d l -> (b l) d"), x_proj_weight)
        delta = delta_proj_weight @ x_dbl[:, :delta_rank].t()
        delta = rearrange(delta, "d (b l) -> b d l", l=L)
        B = rearrange(x_dbl[:, delta_rank:delta_rank + d_state], "(b l) n -> b n l", l=L).contiguous()
        C = rearrange(x_dbl[:, -d_state:], "(b l) n -> b n l", l=L).contiguous()
        return selective_scan_ref(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=True)

    def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                                   C_proj_bias=None, delta_softplus=True):
        assert B is None and C is None and delta_softplus
        return _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias)

    def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                       out_proj_weight, out_proj_bias, A, B=None, C=None, D=None, delta_bias=None,
                       B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
        y = _inner(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias)
        return F.linear(rearrange(y, "b d l -> b l d"), out_proj_weight, out_proj_bias)

    spec = importlib.util.spec_from_file_location("ref_mamba_simple",
                                                  os.path.join(REF, "requirements/mamba_simple.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    ms.mamba_inner_fn_no_out_proj = mamba_inner_fn_no_out_proj
    ms.mamba_inner_fn = mamba_inner_fn

    class Mamba(ms.Mamba):
        def __init__(self, *a, bimamba_type="none", **kw):
            super().__init__(*a, bimamba_type="v3", **kw)
            self.bimamba_type = bimamba_type

        def forward(self, hidden_states, inference_params=None):
            if self.bimamba_type == "v3":
                return super().forward(hidden_states, inference_params)
            if self.bimamba_type == "v2":
                # mamba_simple.py:272-302 then the (unbound) tuple return
                batch, seqlen, dim = hidden_states.shape
                xz = rearrange(self.in_proj.weight @ rearrange(hidden_states, "b l d -> d (b l)"),
                               "d (b l) -> b d l", l=seqlen)
                A = -torch.exp(self.A_log.float())
                A_b = -torch.exp(self.A_b_log.float())
                out = mamba_inner_fn_no_out_proj(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                                 self.dt_proj.weight, A, None, None, self.D.float(),
                                                 delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
                out_b = mamba_inner_fn_no_out_proj(xz.flip([-1]), self.conv1d_b.weight, self.conv1d_b.bias,
                                                   self.x_proj_b.weight, self.dt_proj_b.weight, A_b, None, None,
                                                   self.D_b.float(), delta_bias=self.dt_proj_b.bias.float(),
                                                   delta_softplus=True)
                out = F.linear(rearrange(out + out_b.flip([-1]), "b d l -> b l d"),
                               self.out_proj.weight, self.out_proj.bias)
                return out, None, None, None
            # uni-directional branch mamba_simple.py:303-318
            batch, seqlen, dim = hidden_states.shape
            xz = rearrange(self.in_proj.weight @ rearrange(hidden_states, "b l d -> d (b l)"),
                           "d (b l) -> b d l", l=seqlen)
            A = -torch.exp(self.A_log.float())
            out = mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                 self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias, A, None, None,
                                 self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
            return out, None, None, None

    sys.modules["mamba_ssm"].Mamba = Mamba

    spec = importlib.util.spec_from_file_location("ref_mmunet", os.path.join(REF, "src/UM_Net/MMUNet.py"))
    mmu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mmu)
    d = list(mmu.MMConv.__init__.__defaults__)
    d[6] = "cpu"  # device default (MMUNet.py:19)
    mmu.MMConv.__init__.__defaults__ = tuple(d)

    spec = importlib.util.spec_from_file_location("ref_model", os.path.join(REF, "model.py"))
    model = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(model)
    spec = importlib.util.spec_from_file_location("ref_loss", os.path.join(REF, "loss.py"))
    loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss)

    return types.SimpleNamespace(Mamba=Mamba, MMConv=mmu.MMConv, RCG=mmu.RCG, MM_Net=mmu.MM_Net,
                                 CBAM=mmu.CBAM, ResidualBlock=mmu.ResidualBlock, DecoderBlock=mmu.DecoderBlock,
                                 SideoutBlock=mmu.SideoutBlock, Unet=model.Unet, DICE_BCE_Loss=loss.DICE_BCE_Loss,
                                 selective_scan_ref=selective_scan_ref, causal_conv1d_ref=causal_conv1d_ref)
