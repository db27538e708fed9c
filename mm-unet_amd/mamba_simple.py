"""``Mamba`` block with the reference's patched interface (requirements/mamba_simple.py:34-362):
uni- ("none"/"v1"), bi- ("v2") and tri-directional ("v3") selective scan, 4-tuple return.

Parameter names, shapes, creation order (hence RNG draw order under ``torch.manual_seed``) and
initialisers are those of the reference, so reference state_dicts load and identical seeds give
identical weights (mamba_simple.py:69-183):

    in_proj, conv1d, x_proj, dt_proj (special dt init :89-108), A_log, D,
    A_b_log, conv1d_b, x_proj_b, dt_proj_b, D_b, A_s_log, conv1d_s, x_proj_s, dt_proj_s, D_s, out_proj

The shipped reference is inconsistent about ``bimamba_type`` (``assert == "v3"`` at :125 while
MMConv passes "v1", MMUNet.py:32, and the tuple return at :362 only binds o_1..o_3 in the v3
branch).  Resolution: every type constructs; "none"/"v1" run the
uni-directional branch (:303-318) and return ``(out, None, None, None)``; "v2" runs forward +
reversed (:272-302) and returns ``(out, None, None, None)``; "v3" returns
``(out, o_1, o_2, o_3)`` exactly as :267-270 (o_2 stays in reversed token order).

Not provided (never reached from MM-UNet): ``step`` / inference cache, the ``Block`` wrapper.
The compute goes through the HIP kernels only (no CPU path).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .causal_conv1d_interface import causal_conv1d_fn
from .selective_scan_interface import _dbl_view, mamba_inner_fn, mamba_inner_fn_no_out_proj, selective_scan_fn
from . import deferred, tri_inner, tri_order
from .tall_gemm import proj_bcl, proj_tokens


# A = -exp(A_log) of every Mamba block, formed once per model forward by two multi-tensor launches (and one in the
# backward) instead of exp / neg / mul kernels per block -- 47 blocks x 5 launches of 2.5 us each in MM_Net.
_A_CACHE = {}


class _NegExpAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        outs = torch._foreach_exp([p.detach().float() for p in params])
        torch._foreach_neg_(outs)
        for o, p in zip(outs, params):     # a scan may return d A_log = dA * A for these (selective_scan_interface)
            o._mmu_neg_exp = p.dtype == torch.float32
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*outs)
        ctx.dtypes = [p.dtype for p in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        outs = ctx.saved_tensors
        res = [None] * len(grads)
        idx = []
        for i, g in enumerate(grads):
            if g is None:
                continue
            if deferred.take_prescaled(g):      # already d A_log (possibly still to be filled by a deferred.Scope)
                res[i] = g
            else:
                idx.append(i)
        if deferred._PRESCALED:
            deferred.clear_prescaled()
            raise RuntimeError("_NegExpAll: a pre-scaled dA did not arrive as it was returned (accumulated with another "
                               "gradient?)")
        if idx:
            prod = torch._foreach_mul([grads[i] for i in idx], [outs[i] for i in idx])     # dA_log = dA * A
            for i, v in zip(idx, prod):
                res[i] = v.to(ctx.dtypes[i])
        return tuple(res)


class precomputed_A:
    """``with precomputed_A(model):`` -- every ``Mamba`` inside ``model`` takes its ``A`` from one batched
    ``-exp(A_log)`` for the duration of the block (the cache never outlives the forward pass it was built in)."""

    def __init__(self, model):
        self.params = []
        for m in model.modules():
            if isinstance(m, Mamba):
                self.params += m.used_A_params()

    def __enter__(self):
        deferred.clear_prescaled()
        if self.params and all(p.is_cuda for p in self.params):
            for p, a in zip(self.params, _NegExpAll.apply(*self.params)):
                _A_CACHE[id(p)] = a
        return self

    def __exit__(self, *exc):
        _A_CACHE.clear()
        return False


def neg_exp(param):
    """``-exp(param.float())`` (mamba_simple.py:209,230,251,304), from the per-forward cache when there is one."""
    a = _A_CACHE.get(id(param))
    return a if a is not None else -torch.exp(param.float())


BCL_ENABLED = True   # False: forward_bcl() is forward() between two transposes (fused_paths.plain_aten)
BCL_LOWP = os.environ.get("MMUNET_BCL_LOWP", "1") != "0"   # 0: under autocast the tri-directional block takes the (B, L, C) route


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None, bimamba_type="none", nslices=5):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        if bimamba_type not in ("none", "v1", "v2", "v3"):
            raise ValueError(f"unknown bimamba_type {bimamba_type!r}")
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        self.bimamba_type = bimamba_type
        self.nslices = nslices
        self.activation = "silu"

        def conv():
            return nn.Conv1d(self.d_inner, self.d_inner, kernel_size=d_conv, groups=self.d_inner,
                             padding=d_conv - 1, bias=conv_bias, **factory_kwargs)

        def a_log():
            a = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device)
            p = nn.Parameter(torch.log(a.repeat(self.d_inner, 1).contiguous()))  # S4D-real, fp32
            p._no_weight_decay = True
            return p

        def skip():
            p = nn.Parameter(torch.ones(self.d_inner, device=device))  # fp32
            p._no_weight_decay = True
            return p

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.conv1d = conv()
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        # dt projection init: weight ~ U(+-dt_rank^-0.5 * scale); bias = softplus^-1(dt), dt log-uniform
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True
        self.A_log = a_log()
        self.D = skip()
        # reversed-direction branch (default nn.Linear init for dt_proj_b, mamba_simple.py:149)
        self.A_b_log = a_log()
        self.conv1d_b = conv()
        self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        self.D_b = skip()
        # slice-interleaved ("spatial") branch
        self.A_s_log = a_log()
        self.conv1d_s = conv()
        self.x_proj_s = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj_s = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        self.D_s = skip()
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)
        # v3 returns (out, o_1, o_2, o_3) like the reference; a caller that drops the three branch outputs
        # (RCG, MMUNet.py:409) can switch them off and save the re-ordering copy of o_3
        self.return_branch_outputs = True

    def used_A_params(self):
        """The A_log parameters this block's forward reads (the others never receive a gradient)."""
        return {"v3": [self.A_log, self.A_b_log, self.A_s_log], "v2": [self.A_log, self.A_b_log]}.get(
            self.bimamba_type, [self.A_log])

    def _branch(self, xz, suffix):
        g = lambda n: getattr(self, n + suffix)  # noqa: E731
        A = neg_exp(getattr(self, {"": "A_log", "_b": "A_b_log", "_s": "A_s_log"}[suffix]))
        return mamba_inner_fn_no_out_proj(xz, g("conv1d").weight, g("conv1d").bias, g("x_proj").weight,
                                          g("dt_proj").weight, A, None, None, g("D").float(),
                                          delta_bias=g("dt_proj").bias.float(), delta_softplus=True)

    def _dir_params(self, suffix):
        g = lambda n: getattr(self, n + suffix)  # noqa: E731
        A = neg_exp(getattr(self, {"": "A_log", "_b": "A_b_log", "_s": "A_s_log"}[suffix]))
        return (g("conv1d").weight, g("conv1d").bias, g("x_proj").weight, g("dt_proj").weight, A, g("D").float(),
                g("dt_proj").bias.float())

    def _out_proj(self, y):
        """(B, d_inner, L) -> (B, L, d_model): out_proj applied tokens-last (mamba_simple.py:270)."""
        batch, _, seqlen = y.shape
        o = proj_tokens(self.out_proj.weight, _dbl_view(y))
        if self.out_proj.bias is not None:
            o = o + self.out_proj.bias.view(-1, 1)
        return o.view(-1, batch, seqlen).permute(1, 2, 0)

    def _v3(self, xz, batch, seqlen):
        """The three scans of the tri-directional block and their sum (mamba_simple.py:212-270), before out_proj:
        xz (B, 2*d_inner, L) -> (total (B, d_inner, L), o_1, o_2, o_3)."""
        o_1 = o_2 = o_3 = None
        if seqlen % self.nslices != 0:
            raise RuntimeError(f"Mamba v3: seqlen {seqlen} must be divisible by nslices {self.nslices}")
        ns = self.nslices
        if not self.return_branch_outputs:
            # one conv1d pass for the three token orders, three scans without z, one sum + gate pass (tri_inner.py): the
            # re-ordered copies of xz and the three gated outputs are never formed
            dirs = [self._dir_params(sfx) for sfx in ("", "_b", "_s")]
            if tri_inner.supported(xz, ns, [d[0] for d in dirs], [t for d in dirs for t in d]):
                return tri_inner.tri_mamba_inner(xz, ns, dirs), None, None, None
        fused = tri_order.supported(xz, nslices=ns)
        if fused:   # flip + slice-interleave in one pass; the three input gradients meet in one kernel
            xz_a, xz_f, xz_s = tri_order.tri_split(xz, ns)
        else:
            xz_a, xz_f = xz, xz.flip([-1])
            # token i of slice s -> position i*nslices + s   (mamba_simple.py:245-247)
            xz_s = xz.reshape(batch, 2 * self.d_inner, ns, seqlen // ns).transpose(-1, -2) \
                .reshape(batch, 2 * self.d_inner, seqlen)
        out = self._branch(xz_a, "")
        out_b = self._branch(xz_f, "_b")
        out_sp = self._branch(xz_s, "_s")          # still in slice-interleaved order
        unslice = lambda t: t.reshape(batch, self.d_inner, seqlen // ns, ns).permute(0, 1, 3, 2).flatten(-2)  # noqa: E731
        if self.return_branch_outputs:             # (out, o_1, o_2, o_3) of mamba_simple.py:267-270,362
            o_1, o_2, o_3 = out, out_b, unslice(out_sp)
        if fused and tri_order.supported(out, out_b, out_sp, nslices=ns):
            total = tri_order.tri_combine(out, out_b, out_sp, ns)
        else:
            total = out + out_b.flip([-1]) + (o_3 if o_3 is not None else unslice(out_sp))
        return total, o_1, o_2, o_3

    def forward_bcl(self, x):
        """Channels-first variant for callers that hold feature maps: x (B, d_model, L) contiguous ->
        (out (B, d_model, L) contiguous, o_1, o_2, o_3).  Same computation as ``forward(x.transpose(1, 2))``
        followed by ``.transpose(1, 2)``, without the transposing copies (see tall_gemm.proj_bcl)."""
        lowp = x.dtype == torch.bfloat16 or (torch.is_autocast_enabled() and
                                             torch.get_autocast_dtype("cuda") == torch.bfloat16)
        # bf16 activations (autocast): the same channels-first route with bf16 tensors between the kernels, when the slice
        # count is one the re-ordering kernels take in bf16 (MM-UNet: 16 / 32 / 64)
        ok = (BCL_ENABLED and self.use_fast_path and self.bimamba_type == "v3" and self.in_proj.bias is None and x.is_cuda
              and self.out_proj.bias is None and x.dtype in (torch.float32, torch.bfloat16)
              and (not lowp or (BCL_LOWP and 4 < self.nslices <= 64))
              and (lowp or not torch.is_autocast_enabled()))
        if not ok:
            res = self.forward(x.transpose(1, 2))
            return (res[0].transpose(1, 2),) + tuple(res[1:])
        xz = proj_bcl(self.in_proj.weight, x.contiguous(), True)
        total, o_1, o_2, o_3 = self._v3(xz, x.shape[0], x.shape[2])
        return proj_bcl(self.out_proj.weight, total, False), o_1, o_2, o_3

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> (out (B, L, D), o_1, o_2, o_3)."""
        if inference_params is not None:
            raise NotImplementedError("incremental decoding (inference_params/step) is outside the MM-UNet path")
        batch, seqlen, dim = hidden_states.shape
        # in_proj fused with the BLD -> [2*d_inner][B][L] transpose (mamba_simple.py:201-205)
        # (proj_tokens = W @ X with a split-K weight gradient: the reduction runs over all B*L tokens)
        xz = proj_tokens(self.in_proj.weight, hidden_states.reshape(batch * seqlen, dim).t()) \
            .view(2 * self.d_inner, batch, seqlen).permute(1, 0, 2)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(dtype=xz.dtype).view(1, -1, 1)
        o_1 = o_2 = o_3 = None
        if self.use_fast_path:
            if self.bimamba_type == "v3":
                total, o_1, o_2, o_3 = self._v3(xz, batch, seqlen)
                out = self._out_proj(total)
            elif self.bimamba_type == "v2":
                out = self._branch(xz, "")
                out_b = self._branch(xz.flip([-1]), "_b")
                out = self._out_proj(out + out_b.flip([-1]))
            else:
                A = neg_exp(self.A_log)
                out = mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                     self.dt_proj.weight, self.out_proj.weight, self.out_proj.bias, A, None, None,
                                     self.D.float(), delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
        else:
            # un-fused path (mamba_simple.py:319-361), uni-directional
            A = neg_exp(self.A_log)
            x, z = xz.chunk(2, dim=1)
            x = causal_conv1d_fn(x, self.conv1d.weight.view(self.d_inner, self.d_conv), self.conv1d.bias,
                                 self.activation)
            x_dbl = self.x_proj(x.permute(0, 2, 1).reshape(batch * seqlen, self.d_inner))
            dt, B, C = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
            dt = (self.dt_proj.weight @ dt.t()).view(self.d_inner, batch, seqlen).permute(1, 0, 2)
            B = B.reshape(batch, seqlen, self.d_state).permute(0, 2, 1).contiguous()
            C = C.reshape(batch, seqlen, self.d_state).permute(0, 2, 1).contiguous()
            y = selective_scan_fn(x, dt, A, B, C, self.D.float(), z=z, delta_bias=self.dt_proj.bias.float(),
                                  delta_softplus=True)
            out = self.out_proj(y.permute(0, 2, 1))
        return out, o_1, o_2, o_3
