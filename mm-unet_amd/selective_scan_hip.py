"""``selective_scan_hip`` -- the module that stands where the reference's pybind extension
``selective_scan_cuda`` stands (requirements/Mamba/mamba/csrc/selective_scan/selective_scan.cpp:494-497).

Same function names, positional arguments, return lists and error behaviour
(``RuntimeError`` for what the reference rejects with ``TORCH_CHECK``); the work is done by
the hand-written gfx950 kernels behind the C-ABI (include/mmunet_amd.h).  Tensor allocation
that the reference does inside the pybind function happens here, so the ABI carries raw
pointers only.

Differences a caller can observe:
  * the chunk-state tensor ``x`` is ``(batch, dim, n_chunks, 2*dstate)`` with
    ``n_chunks = ceil(seqlen / chunk_len(dstate))`` (chunk_len 128 for dstate<=16) instead of
    the reference's fixed 2048; ``x[:, :, -1, 1::2]`` is still the last state
    (selective_scan_interface.py:40);
  * real ``A`` and input-dependent (``dim() >= 3``) ``B``/``C`` only -- the only variant
    MM-UNet uses; complex ``A`` / constant ``B``,``C`` raise ``RuntimeError``;
  * float32 and bfloat16 I/O (no float16); dstate <= 128.
"""
import os

import torch

from . import _lib, deferred


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def chunk_len(dstate, dtype=torch.float32):
    code = _lib.MMU_DTYPE_F32 if dtype == torch.float32 else _lib.MMU_DTYPE_BF16
    n = _lib.lib().mmu_scan_chunk_len(int(dstate), code)
    _check(n > 0, f"selective_scan only supports state dimension <= 128 (got {dstate})")
    return n


def _common_checks(u, delta, A, B, C, D_, z_, delta_bias_):
    _lib.require_gpu(u, delta, A, B, C, D_, z_, delta_bias_)
    _check(u.dtype in (torch.float32, torch.bfloat16), f"selective_scan: unsupported input dtype {u.dtype}")
    _check(A.dtype == torch.float32, "selective_scan: A must be float32 (complex A is not on the MM-UNet path)")
    _check(B.dim() == 4 and C.dim() == 4,
           "selective_scan: only input-dependent B and C of shape (batch, groups, dstate, seqlen) are supported")
    _check(delta.dtype == u.dtype and B.dtype == u.dtype and C.dtype == u.dtype,
           "selective_scan: delta, B, C must have the dtype of u")
    _check(u.dim() == 3, "selective_scan: u must be (batch, dim, seqlen)")
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    g = B.shape[1]
    _check(u.stride(-1) == 1 and delta.stride(-1) == 1, "selective_scan: u and delta need unit stride in seqlen")
    _check(tuple(delta.shape) == (batch, dim, seqlen), "selective_scan: delta has the wrong shape")
    _check(tuple(A.shape) == (dim, dstate), "selective_scan: A has the wrong shape")
    _check(tuple(B.shape) == (batch, g, dstate, seqlen) and B.stride(-1) == 1, "selective_scan: B has the wrong shape/stride")
    _check(tuple(C.shape) == (batch, g, dstate, seqlen) and C.stride(-1) == 1, "selective_scan: C has the wrong shape/stride")
    _check(dim % g == 0, "selective_scan: dim must be divisible by the number of groups")
    for name, t in (("D", D_), ("delta_bias", delta_bias_)):
        if t is not None:
            _check(t.dtype == torch.float32 and tuple(t.shape) == (dim,) and t.stride(-1) == 1,
                   f"selective_scan: {name} must be a contiguous float32 (dim,) tensor")
    if z_ is not None:
        _check(z_.dtype == u.dtype and tuple(z_.shape) == (batch, dim, seqlen) and z_.stride(-1) == 1,
               "selective_scan: z has the wrong dtype/shape/stride")
    return batch, dim, seqlen, dstate, g


def _fwd_one(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, want_out=True, x_out=None):
    """selective_scan_cuda.fwd: returns ``[out, x]`` or ``[out, x, out_z]`` (selective_scan.cpp:226-336).

    ``want_out=False`` (extension over the reference signature, only valid with ``z_``): do not
    materialise the un-gated ``out``; the list then holds ``None`` in its place.  The fused
    ``mamba_inner`` path uses it -- its backward recomputes y, so writing ``out`` would be a wasted
    [batch, dim, L] store per call."""
    batch, dim, seqlen, dstate, g = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    T = chunk_len(dstate, u.dtype)
    n_chunks = (seqlen + T - 1) // T
    _check(want_out or z_ is not None, "selective_scan_fwd: want_out=False needs z")
    out = torch.empty_like(delta) if want_out else None
    out_z = torch.empty_like(z_) if z_ is not None else None
    x = x_out if x_out is not None else torch.empty((batch, dim, n_chunks, 2 * dstate), device=u.device,
                                                    dtype=torch.float32)
    _check(x.is_contiguous() and x.dtype == torch.float32 and tuple(x.shape) == (batch, dim, n_chunks, 2 * dstate),
           "selective_scan_fwd: x_out must be a contiguous float32 (batch, dim, n_chunks, 2 * dstate) tensor")
    p = _lib.ScanFwdParams()
    p.batch, p.dim, p.seqlen, p.dstate, p.ngroups = batch, dim, seqlen, dstate, g
    p.dtype = _lib.dtype_code(u)
    p.delta_softplus = int(bool(delta_softplus))
    p.n_chunks = n_chunks
    p.u, p.delta, p.A, p.B, p.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    p.D, p.z, p.delta_bias = _lib.ptr(D_), _lib.ptr(z_), _lib.ptr(delta_bias_)
    p.out, p.out_z, p.x = _lib.ptr(out), _lib.ptr(out_z), x.data_ptr()
    p.u_bs, p.u_ds = u.stride(0), u.stride(1)
    p.delta_bs, p.delta_ds = delta.stride(0), delta.stride(1)
    if out is not None:
        p.out_bs, p.out_ds = out.stride(0), out.stride(1)
    if z_ is not None:
        p.z_bs, p.z_ds = z_.stride(0), z_.stride(1)
        p.out_z_bs, p.out_z_ds = out_z.stride(0), out_z.stride(1)
    p.A_ds, p.A_ns = A.stride(0), A.stride(1)
    p.B_bs, p.B_gs, p.B_ns = B.stride(0), B.stride(1), B.stride(2)
    p.C_bs, p.C_gs, p.C_ns = C.stride(0), C.stride(1), C.stride(2)
    with torch.cuda.device(u.device):
        _lib.check(_lib.lib().mmu_selective_scan_fwd(p, _lib.stream_of(u)))
    return [out, x, out_z] if z_ is not None else [out, x]


def _bwd_one(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z,
             dB_out=None, dC_out=None, dA_times_A=False, defer=False):
    """selective_scan_cuda.bwd: returns ``[du, ddelta, dA, dB, dC, dD, ddelta_bias, (dz), (out_z)]``
    (selective_scan.cpp:338-492).  ``out_`` (the forward's y before gating) is read when given; without it y is
    recomputed from the states the kernel rebuilds anyway.

    ``dB_out`` / ``dC_out`` (extension): pre-allocated float32 (batch, groups, dstate, L) views with unit
    L stride to receive dB / dC (e.g. rows of the projection-gradient matrix), saving a copy.
    ``dA_times_A`` (extension): dA is returned multiplied by A -- d A_log for A = -exp(A_log).
    ``defer`` (extension): inside a deferred.Scope the final sums of dA / dD / ddelta_bias may run at ``Scope.launch()``
    -- only for a caller whose three results go straight into parameter gradients."""
    batch, dim, seqlen, dstate, g = _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
    _lib.require_gpu(dout, x_, dz_)
    _check(dout.dtype == u.dtype and tuple(dout.shape) == (batch, dim, seqlen) and dout.stride(-1) == 1,
           "selective_scan_bwd: dout has the wrong dtype/shape/stride")
    T = chunk_len(dstate, u.dtype)
    n_chunks = (seqlen + T - 1) // T
    if x_ is not None:
        _check(x_.dtype == torch.float32 and x_.is_contiguous() and
               tuple(x_.shape) == (batch, dim, n_chunks, 2 * dstate),
               "selective_scan_bwd: x must be the contiguous float32 chunk-state tensor returned by fwd")
    has_z = z_ is not None
    dz = out_z = None
    if has_z:
        if dz_ is not None:
            _check(dz_.dtype == u.dtype and tuple(dz_.shape) == (batch, dim, seqlen) and dz_.stride(-1) == 1,
                   "selective_scan_bwd: dz has the wrong dtype/shape/stride")
            dz = dz_
        else:
            dz = torch.empty_like(z_)
        if recompute_out_z:
            out_z = torch.empty_like(out_) if out_ is not None else torch.empty_like(delta)
    du = torch.empty_like(u)
    ddelta = torch.empty_like(delta)
    dA = torch.empty((dim, dstate), device=u.device, dtype=torch.float32)
    def _acc(t):
        if t is None:
            return torch.empty((batch, g, dstate, seqlen), device=u.device, dtype=torch.float32)
        _check(t.dtype == torch.float32 and tuple(t.shape) == (batch, g, dstate, seqlen) and t.stride(-1) == 1,
               "selective_scan_bwd: dB_out/dC_out must be float32 (batch, groups, dstate, L) with unit L stride")
        return t
    dB, dC = _acc(dB_out), _acc(dC_out)
    dD = torch.empty_like(D_) if D_ is not None else None
    ddelta_bias = torch.empty_like(delta_bias_) if delta_bias_ is not None else None
    L = _lib.lib()
    ws_bytes = L.mmu_scan_bwd_workspace_bytes(batch, dim, seqlen, dstate, _lib.dtype_code(u), int(x_ is not None))
    ws = torch.empty(ws_bytes // 4, device=u.device, dtype=torch.float32)
    p = _lib.ScanBwdParams()
    p.batch, p.dim, p.seqlen, p.dstate, p.ngroups = batch, dim, seqlen, dstate, g
    p.dtype = _lib.dtype_code(u)
    p.delta_softplus = int(bool(delta_softplus))
    p.n_chunks = n_chunks
    p.u, p.delta, p.A, p.B, p.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    p.D, p.z, p.delta_bias = _lib.ptr(D_), _lib.ptr(z_), _lib.ptr(delta_bias_)
    p.dout, p.x = dout.data_ptr(), _lib.ptr(x_)
    p.du, p.ddelta, p.dA, p.dB, p.dC = du.data_ptr(), ddelta.data_ptr(), dA.data_ptr(), dB.data_ptr(), dC.data_ptr()
    p.dD, p.ddelta_bias, p.dz, p.out_z = _lib.ptr(dD), _lib.ptr(ddelta_bias), _lib.ptr(dz), _lib.ptr(out_z)
    p.workspace = ws.data_ptr()
    p.u_bs, p.u_ds = u.stride(0), u.stride(1)
    p.delta_bs, p.delta_ds = delta.stride(0), delta.stride(1)
    p.dout_bs, p.dout_ds = dout.stride(0), dout.stride(1)
    p.du_bs, p.du_ds = du.stride(0), du.stride(1)
    p.ddelta_bs, p.ddelta_ds = ddelta.stride(0), ddelta.stride(1)
    if has_z:
        p.z_bs, p.z_ds = z_.stride(0), z_.stride(1)
        p.dz_bs, p.dz_ds = dz.stride(0), dz.stride(1)
        if out_z is not None:
            p.out_z_bs, p.out_z_ds = out_z.stride(0), out_z.stride(1)
    p.A_ds, p.A_ns = A.stride(0), A.stride(1)
    p.B_bs, p.B_gs, p.B_ns = B.stride(0), B.stride(1), B.stride(2)
    p.C_bs, p.C_gs, p.C_ns = C.stride(0), C.stride(1), C.stride(2)
    p.dB_bs, p.dB_gs, p.dB_ns = dB.stride(0), dB.stride(1), dB.stride(2)
    p.dC_bs, p.dC_gs, p.dC_ns = dC.stride(0), dC.stride(1), dC.stride(2)
    _check(not dA_times_A or (A.is_contiguous() and A.dtype == torch.float32), "selective_scan_bwd: dA_times_A needs a "
           "contiguous float32 A")
    p.dA_times_A = int(bool(dA_times_A))
    if out_ is not None and has_z:
        # the forward's y before gating (selective_scan.cpp:338 `out_`): read instead of recomputed where the kernel can
        _check(out_.shape == u.shape and out_.dtype == u.dtype and out_.stride(-1) == 1 and out_.device == u.device,
               "selective_scan_bwd: out must have u's shape, dtype and device and unit stride along L")
        p.out, p.out_bs, p.out_ds = out_.data_ptr(), out_.stride(0), out_.stride(1)
    with torch.cuda.device(u.device):
        if defer:
            _lib.check(L.mmu_selective_scan_bwd(p, _lib.stream_of(u)))
            deferred.keep(ws)
        else:
            with deferred.paused():
                _lib.check(L.mmu_selective_scan_bwd(p, _lib.stream_of(u)))
    result = [du, ddelta, dA, dB if dB.dtype == B.dtype else dB.to(B.dtype),
              dC if dC.dtype == C.dtype else dC.to(C.dtype), dD, ddelta_bias]
    if has_z:
        result.append(dz)
    if recompute_out_z:
        result.append(out_z)
    return result


# ---------------------------------------------------------------------------------------------------------------
# State groups: dstate = 32, 48, 64, ... on the dstate-16 kernels
# ---------------------------------------------------------------------------------------------------------------
# The fast kernels (streaming / packed-pair forward, state-split register backward) are cut for 16 states.  A larger
# state (BASELINE config 5: d_state = 64) is a SUM over groups of 16 states of independent recurrences:
#     y = sum_g y_g + D u,   h of group g depends on A[:, 16g:16g+16], B[:, :, 16g:16g+16], C[:, :, 16g:16g+16] only,
# so it runs as one dstate-16 call per group on strided VIEWS of A / B / C (no copies; dB / dC of a group are written
# straight into their rows) and the per-token results -- all linear in the group's y, the gate included -- are added
# up in float32, instead of the generic-dstate kernels, whose backward adds dB / dC with
# LDS float atomics (137 of 212 GPU-ms per training step of config 5, profiles/r02_config5_before.txt).
# Worth it from 512 K elements (batch * dim * seqlen) on: since round 2's small-block kernels four dstate-16 launches also
# beat the generic kernel on MMConv's 6-channel blocks at 256 x 256 and up; tiny scans keep the single generic launch.
GROUP_SPLIT = True
GROUP_SPLIT_MIN_ELEMENTS = int(os.environ.get("MMUNET_GROUP_SPLIT_MIN", str(1 << 19)))   # batch * dim * seqlen (1 << 22 until the small-block kernels of round 2: 93.9 vs 86.1 ms on config 5)


def group_split(dstate, u):
    """True when fwd / bwd run this call as dstate // 16 launches of the dstate-16 kernels."""
    if not (GROUP_SPLIT and dstate > 16 and dstate % 16 == 0 and u.numel() >= GROUP_SPLIT_MIN_ELEMENTS
            and u.shape[-1] % 512 == 0):
        return False
    # the groups' chunk states are laid out with the dstate-16 kernels' chunk length: only states whose own chunk
    # length is the same (48, 64 today) can be split; 32 and 80..128 stay on the generic kernels (_fwd_one / _bwd_one)
    return chunk_len(dstate, torch.float32) == chunk_len(16, torch.float32)


def _groups(A, B, C):
    for g in range(A.shape[1] // 16):
        sl = slice(16 * g, 16 * g + 16)
        yield g, sl, A[:, sl], B[:, :, sl], C[:, :, sl]


def _f32(*ts):
    """bf16 I/O: the group launches run on float32 copies -- a group's y can be two orders of magnitude larger than
    the sum over the groups, and rounding each partial to bf16 first costs the sum its last bits (seen: 1.0 absolute
    on outputs of magnitude 20).  The kernels are bound by instruction issue, not bytes."""
    return [t if (t is None or t.dtype == torch.float32) else t.float() for t in ts]


def _acc(total, part):
    return part if total is None else total.add_(part)


def _sum_parts(parts, dtype, out=None):
    """Sum of the groups' float32 partial outputs in the I/O type: one pass (csrc/sum_parts.hip) instead of three in-place
    adds and a cast.  ``out``: optional destination (contiguous, ``dtype``)."""
    parts = [t for t in parts if t is not None]
    if not parts:
        return None
    # element-wise over the storage: the parts (and out) only have to be dense with the SAME strides -- the fused Mamba path
    # hands over [D][B][L]-ordered tensors, which are not "contiguous" as (B, D, L)
    order = sorted(range(parts[0].dim()), key=lambda i: -parts[0].stride(i))
    same = lambda t: t.shape == parts[0].shape and t.stride() == parts[0].stride()   # noqa: E731
    ok = (parts[0].permute(order).is_contiguous() and all(t.dtype == torch.float32 and same(t) and t.data_ptr() % 16 == 0
                                                          for t in parts)
          and dtype in (torch.float32, torch.bfloat16) and len(parts) <= 4
          and (out is None or (same(out) and out.dtype == dtype and out.data_ptr() % 16 == 0)))
    if not ok:
        total = parts[0]
        for t in parts[1:]:
            total = total.add_(t) if total.dtype == torch.float32 else total + t
        if out is not None:
            return out.copy_(total)
        return total.to(dtype)
    if len(parts) == 1 and dtype == torch.float32 and out is None:
        return parts[0]
    res = out if out is not None else torch.empty_like(parts[0], dtype=dtype)    # (keeps the parts' strides)
    p = _lib.SumPartsParams()
    p.n, p.nparts, p.out_dtype = parts[0].numel(), len(parts), _lib.dtype_code(res)
    for k, t in enumerate(parts):
        p.parts[k] = t.data_ptr()
    p.out = res.data_ptr()
    with torch.cuda.device(res.device):
        _lib.check(_lib.lib().mmu_sum_parts(p, _lib.stream_of(res)))
    return res


def _fwd_groups(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, want_out=True, group_major_x=False):
    """Everything the kernel computes per token is LINEAR in the group's y -- out = y, out_z = y silu(z) -- so every
    group launch gets z (and group 0 gets D) and the outputs are simply added up."""
    io_dtype = u.dtype
    u, delta, z_, B, C = _f32(u, delta, z_, B, C)
    batch, dim, seqlen = u.shape
    dstate = A.shape[1]
    T = chunk_len(dstate, u.dtype)
    _check(T == chunk_len(16, u.dtype), "selective_scan: group split needs equal chunk lengths")
    n_chunks = (seqlen + T - 1) // T
    # group_major_x (internal callers that only hand x back to bwd): the groups' chunk states stay where their launches
    # wrote them, [group][batch][dim][chunk][32]; the reference's layout needs one strided copy per group and direction
    if group_major_x:
        x = torch.empty((dstate // 16, batch, dim, n_chunks, 32), device=u.device, dtype=torch.float32)
    else:
        x = torch.empty((batch, dim, n_chunks, 2 * dstate), device=u.device, dtype=torch.float32)
    outs, out_zs = [], []
    for g, sl, Ag, Bg, Cg in _groups(A, B, C):
        r = _fwd_one(u, delta, Ag, Bg, Cg, D_ if g == 0 else None, z_, delta_bias_, delta_softplus,
                     want_out=want_out or z_ is None, x_out=x[g] if group_major_x else None)
        if not group_major_x:
            x[..., 32 * g:32 * g + 32] = r[1]
        outs.append(r[0])
        if z_ is not None:
            out_zs.append(r[2])
    out = _sum_parts(outs, io_dtype)
    return [out, x] if z_ is None else [out, x, _sum_parts(out_zs, io_dtype)]


def _bwd_groups(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, dz_, delta_softplus, recompute_out_z,
                dB_out=None, dC_out=None):
    """Likewise du, ddelta, dz (= dout * y * d silu / dz: linear in y), ddelta_bias and the recomputed out_z are sums
    over the groups; dA, dB, dC of a group are its own rows (dB / dC written in place through strided views)."""
    io_dtype, bc_dtype = u.dtype, B.dtype
    batch, dim, seqlen = u.shape
    dstate, g_ = A.shape[1], B.shape[1]
    if x_ is None:
        x_ = _fwd_groups(u, delta, A, B, C, D_, None, delta_bias_, delta_softplus, group_major_x=True)[1]
    u, delta, z_, dout, B, C = _f32(u, delta, z_, dout, B, C)
    f32 = dict(device=u.device, dtype=torch.float32)
    dB = dB_out if dB_out is not None else torch.empty((batch, g_, dstate, seqlen), **f32)
    dC = dC_out if dC_out is not None else torch.empty((batch, g_, dstate, seqlen), **f32)
    dA = torch.empty((dim, dstate), **f32)
    dbias = dD = None
    dus, ddeltas, dzs, out_zs = [], [], [], []
    for g, sl, Ag, Bg, Cg in _groups(A, B, C):
        r = _bwd_one(u, delta, Ag, Bg, Cg, D_ if g == 0 else None, z_, delta_bias_, dout,
                     x_[g] if x_.dim() == 5 else x_[..., 32 * g:32 * g + 32].contiguous(), None, None, delta_softplus,
                     recompute_out_z,
                     dB_out=dB[:, :, sl], dC_out=dC[:, :, sl])
        dA[:, sl] = r[2]
        dus.append(r[0])
        ddeltas.append(r[1])
        if g == 0:
            dD = r[5]
        if r[6] is not None:
            dbias = _acc(dbias, r[6])
        if z_ is not None:
            dzs.append(r[7])
            if recompute_out_z:
                out_zs.append(r[8])
    result = [_sum_parts(dus, io_dtype), _sum_parts(ddeltas, io_dtype), dA, dB if dB.dtype == bc_dtype else dB.to(bc_dtype),
              dC if dC.dtype == bc_dtype else dC.to(bc_dtype), dD, dbias]
    if z_ is not None:
        if dz_ is not None and dz_.shape == dzs[0].shape and dz_.stride() == dzs[0].stride():
            dz = _sum_parts(dzs, dz_.dtype, out=dz_)
        elif dz_ is not None:
            dz = dz_.copy_(_sum_parts(dzs, torch.float32))
        else:
            dz = _sum_parts(dzs, io_dtype)
        result.append(dz)
        if recompute_out_z:
            result.append(_sum_parts(out_zs, io_dtype))
    return result


def fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, want_out=True, opaque_x=False):
    """selective_scan_cuda.fwd (selective_scan.cpp:226-336): ``[out, x]`` or ``[out, x, out_z]``; see ``_fwd_one``.
    ``want_out=False`` (extension, only with ``z_``): ``None`` in place of the un-gated ``out``.
    ``opaque_x=True`` (extension): the caller only hands ``x`` back to :func:`bwd`; a state-group call may then keep it
    group-major (5-D) instead of the reference's (batch, dim, n_chunks, 2 * dstate) layout."""
    if A.dim() == 2 and u.dim() == 3 and group_split(A.shape[1], u):
        _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
        _check(want_out or z_ is not None, "selective_scan_fwd: want_out=False needs z")
        return _fwd_groups(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, want_out, group_major_x=opaque_x)
    return _fwd_one(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, want_out)


def bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z,
        dB_out=None, dC_out=None, dA_times_A=False, defer=False):
    """selective_scan_cuda.bwd (selective_scan.cpp:338-492); see ``_bwd_one``."""
    if A.dim() == 2 and u.dim() == 3 and group_split(A.shape[1], u):
        _common_checks(u, delta, A, B, C, D_, z_, delta_bias_)
        res = _bwd_groups(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, dz_, delta_softplus, recompute_out_z,
                          dB_out, dC_out)
        if dA_times_A:
            res[2] = res[2] * A
        return res
    return _bwd_one(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z,
                    dB_out, dC_out, dA_times_A, defer)
