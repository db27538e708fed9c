"""``causal_conv1d_hip`` -- stands where the reference's pybind extension ``causal_conv1d_cuda``
stands (requirements/Mamba/causal-conv1d/csrc/causal_conv1d.cpp:329-333): same three function
names, positional arguments, returns and error behaviour; the work is done by the gfx950
kernels behind the C-ABI (include/mmunet_amd.h).

The kernels are channel-first (unit stride along seqlen), which is the only layout MM-UNet
produces.  A channel-last input (``x.stride(1) == 1``) is accepted
like in the reference but is first brought to channel-first here.
Weights/bias are read as float32 (cast here when they arrive in a 16-bit dtype).
"""
import torch

from . import _lib, deferred


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _prep(x, weight, bias_):
    _lib.require_gpu(x, weight, bias_)
    _check(x.dtype in (torch.float32, torch.bfloat16), f"causal_conv1d: unsupported input dtype {x.dtype}")
    _check(x.dim() == 3 and weight.dim() == 2, "causal_conv1d: x must be (batch, dim, seqlen), weight (dim, width)")
    batch, dim, seqlen = x.shape
    width = weight.shape[-1]
    _check(tuple(weight.shape) == (dim, width), "causal_conv1d: weight has the wrong shape")
    _check(x.stride(2) == 1 or x.stride(1) == 1, "causal_conv1d: x needs unit stride along seqlen or channels")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    if bias_ is not None:
        _check(bias_.dtype == weight.dtype and tuple(bias_.shape) == (dim,) and bias_.stride(-1) == 1,
               "causal_conv1d: bias must be a contiguous (dim,) tensor with the dtype of weight")
    w32 = weight if weight.dtype == torch.float32 else weight.float()
    b32 = None if bias_ is None else (bias_ if bias_.dtype == torch.float32 else bias_.float())
    return batch, dim, seqlen, width, w32, b32


def causal_conv1d_fwd(x, weight, bias_, silu_activation):
    """causal_conv1d.cpp:130-189.  Returns ``out`` (same layout as ``x``)."""
    batch, dim, seqlen, width, w32, b32 = _prep(x, weight, bias_)
    if x.stride(2) != 1:  # channel-last caller
        x = x.contiguous()
    out = torch.empty_like(x)
    p = _lib.Conv1dFwdParams()
    p.batch, p.dim, p.seqlen, p.width = batch, dim, seqlen, width
    p.dtype, p.silu = _lib.dtype_code(x), int(bool(silu_activation))
    p.x, p.weight, p.bias, p.out = x.data_ptr(), w32.data_ptr(), _lib.ptr(b32), out.data_ptr()
    p.x_bs, p.x_ds, p.out_bs, p.out_ds = x.stride(0), x.stride(1), out.stride(0), out.stride(1)
    p.w_ds, p.w_ws = w32.stride(0), w32.stride(1)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_causal_conv1d_fwd(p, _lib.stream_of(x)))
    return out


def causal_conv1d_bwd(x, weight, bias_, dout, dx_, silu_activation):
    """causal_conv1d.cpp:191-268.  Returns ``[dx, dweight, dbias]``; ``dx_`` may be a pre-allocated
    view (selective_scan_interface.py:244-245,281-283)."""
    batch, dim, seqlen, width, w32, b32 = _prep(x, weight, bias_)
    _lib.require_gpu(dout, dx_)
    _check(tuple(dout.shape) == (batch, dim, seqlen) and dout.dtype == x.dtype, "causal_conv1d_bwd: dout mismatch")
    if x.stride(2) != 1:
        x = x.contiguous()
    if dout.stride(2) != 1:
        dout = dout.contiguous()
    if dx_ is not None:
        _check(dx_.dtype == x.dtype and tuple(dx_.shape) == (batch, dim, seqlen) and dx_.stride(2) == 1,
               "causal_conv1d_bwd: dx has the wrong dtype/shape/stride")
        dx = dx_
    else:
        dx = torch.empty_like(x)
    # weight-gradient sums through a workspace of per-block partials added in a fixed order: no float atomics,
    # bit-reproducible, nothing to zero (the reference: atomicAdd into zeroed dweight / dbias)
    dweight = torch.empty((dim, width), device=x.device, dtype=torch.float32)
    dbias = torch.empty((dim,), device=x.device, dtype=torch.float32) if bias_ is not None else None
    ws = torch.empty(_lib.lib().mmu_causal_conv1d_bwd_workspace_floats(batch, dim, seqlen), device=x.device,
                     dtype=torch.float32)
    p = _lib.Conv1dBwdParams()
    p.batch, p.dim, p.seqlen, p.width = batch, dim, seqlen, width
    p.dtype, p.silu = _lib.dtype_code(x), int(bool(silu_activation))
    p.x, p.weight, p.bias = x.data_ptr(), w32.data_ptr(), _lib.ptr(b32)
    p.dout, p.dx, p.dweight, p.dbias = dout.data_ptr(), dx.data_ptr(), dweight.data_ptr(), _lib.ptr(dbias)
    p.x_bs, p.x_ds = x.stride(0), x.stride(1)
    p.dout_bs, p.dout_ds = dout.stride(0), dout.stride(1)
    p.dx_bs, p.dx_ds = dx.stride(0), dx.stride(1)
    p.w_ds, p.w_ws = w32.stride(0), w32.stride(1)
    p.workspace = ws.data_ptr()
    # inside a deferred.Scope the ordered sum of the per-block partials runs with all the others of the backward pass --
    # unless the result is converted (read) right here
    same = (weight.dtype == torch.float32 and (bias_ is None or bias_.dtype == torch.float32)
            and deferred.may_defer(weight, bias_))   # (parameters, or contiguous views of them: deferred.py's contract)
    with torch.cuda.device(x.device):
        if same:
            _lib.check(_lib.lib().mmu_causal_conv1d_bwd(p, _lib.stream_of(x)))
            deferred.keep(ws)
        else:
            with deferred.paused():
                _lib.check(_lib.lib().mmu_causal_conv1d_bwd(p, _lib.stream_of(x)))
    return [dx, dweight.to(weight.dtype), dbias.to(bias_.dtype) if bias_ is not None else None]


def causal_conv1d_update(x, conv_state, weight, bias_, silu_activation):
    """causal_conv1d.cpp:270-327.  ``conv_state`` (batch, dim, width) is updated in place."""
    _lib.require_gpu(x, conv_state, weight, bias_)
    _check(x.dim() == 2, "causal_conv1d_update: x must be (batch, dim)")
    batch, dim = x.shape
    width = weight.shape[-1]
    _check(tuple(weight.shape) == (dim, width) and tuple(conv_state.shape) == (batch, dim, width),
           "causal_conv1d_update: shape mismatch")
    _check(conv_state.dtype == x.dtype, "causal_conv1d_update: conv_state must have the dtype of x")
    _check(2 <= width <= 4, "causal_conv1d only supports width between 2 and 4")
    w32 = weight if weight.dtype == torch.float32 else weight.float()
    b32 = None if bias_ is None else (bias_ if bias_.dtype == torch.float32 else bias_.float())
    out = torch.empty_like(x)
    p = _lib.Conv1dUpdateParams()
    p.batch, p.dim, p.width = batch, dim, width
    p.dtype, p.silu = _lib.dtype_code(x), int(bool(silu_activation))
    p.x, p.conv_state, p.weight, p.bias, p.out = (x.data_ptr(), conv_state.data_ptr(), w32.data_ptr(),
                                                  _lib.ptr(b32), out.data_ptr())
    p.x_bs, p.x_ds = x.stride(0), x.stride(1)
    p.cs_bs, p.cs_ds, p.cs_ws = conv_state.stride(0), conv_state.stride(1), conv_state.stride(2)
    p.out_bs, p.out_ds = out.stride(0), out.stride(1)
    p.w_ds, p.w_ws = w32.stride(0), w32.stride(1)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().mmu_causal_conv1d_update(p, _lib.stream_of(x)))
    return out
