"""Autograd layer over the HIP causal-conv1d kernels; public surface of the reference's
``causal_conv1d/causal_conv1d_interface.py``:

    causal_conv1d_fn(x, weight, bias=None, activation=None)          (:37-46)
    causal_conv1d_update(x, conv_state, weight, bias=None, activation=None)   (:68-80)
    class CausalConv1dFn                                             (:10-34)

No CPU path.
"""
import torch

from . import causal_conv1d_hip


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in [None, "silu", "swish"]:
            raise NotImplementedError("activation must be None, silu, or swish")
        if x.stride(2) != 1 and x.stride(1) != 1:
            x = x.contiguous()
        bias = bias.contiguous() if bias is not None else None
        ctx.save_for_backward(x, weight, bias)
        ctx.activation = activation in ["silu", "swish"]
        return causal_conv1d_hip.causal_conv1d_fwd(x, weight, bias, ctx.activation)

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias = ctx.saved_tensors
        if dout.stride(2) != 1 and dout.stride(1) != 1:
            dout = dout.contiguous()
        dx, dweight, dbias = causal_conv1d_hip.causal_conv1d_bwd(x, weight, bias, dout, None, ctx.activation)
        return dx, dweight, dbias if bias is not None else None, None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim,); activation: None | "silu" | "swish".
    Returns (batch, dim, seqlen)."""
    return CausalConv1dFn.apply(x, weight, bias, activation)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """x: (batch, dim); conv_state: (batch, dim, width), updated in place.  Returns (batch, dim)."""
    if activation not in [None, "silu", "swish"]:
        raise NotImplementedError("activation must be None, silu, or swish")
    return causal_conv1d_hip.causal_conv1d_update(x, conv_state, weight, bias, activation in ["silu", "swish"])
