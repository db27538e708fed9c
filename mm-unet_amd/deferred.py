"""The final ordered sums of a backward pass's weight gradients in ONE launch (csrc/deferred_reduce.hip).

``gemm_nt`` (projection / DSC weight gradients), ``conv3x3_small``'s weight gradient and ``causal_conv1d_bwd`` each end in a
small kernel that adds per-workgroup partial sums in a fixed order: 131 of them per MM-UNet training step, every one a
dependent launch at the ~4.6 us floor in the middle of the backward chain.  Nothing reads a weight gradient before the
optimizer, so inside a :class:`Scope` those launchers only record their reduction and ``Scope.launch()`` runs them all
in one kernel, with the summation order of the kernels it replaces.

    scope = Scope(device); scope.reserve()          # device tables (outside any capture)
    with scope:                                     # mmu_deferred_begin ... mmu_deferred_end
        loss.backward()                             # weight gradients are allocated, their partials kept alive
        scope.launch()                              # one launch (capturable); the gradients are final after it
    scope.bind()                                    # writes the job table; BEFORE the first replay of a captured graph

Eagerly ``launch()`` binds by itself.  Contract: the weight gradients are leaves of the backward pass (stored into
``param.grad``, not accumulated into or read before ``launch()``) -- ``TrainStep`` opens a scope only around the backward
pass it captures and CHECKS the contract there (``Scope.verify_destinations``: every result address inside exactly one
``param.grad``, none twice).  Process-wide: one scope at a time.
"""
import os

import torch

from . import _lib

_active = None


def keep(*tensors):
    """Called by the wrappers whose final sum may have been deferred: keeps their partial-sum buffers alive."""
    if _active is not None:
        _active._keep.extend(t for t in tensors if t is not None)


def active():
    return _active is not None


# Gradients of A that already carry the factor A (the scan's dA_times_A): mamba_simple._NegExpAll.backward passes them
# through as d A_log instead of multiplying.  Keyed by storage address between a scan's backward and _NegExpAll's, which
# empties the set (as does every model forward).
_PRESCALED = set()


def mark_prescaled(t):
    _PRESCALED.add(t.data_ptr())


def take_prescaled(t):
    ptr = t.data_ptr()
    if ptr in _PRESCALED:
        _PRESCALED.discard(ptr)
        return True
    return False


def clear_prescaled():
    _PRESCALED.clear()


_GUARD = os.environ.get("MMUNET_DEFER_GUARD", "1") != "0"   # "0": every wrapper defers unconditionally (A/B only)


def may_defer(*weights):
    """The leaf-parameter contract, checked where a wrapper is called: every tensor whose gradient the call produces is a
    float32 LEAF (a parameter) or a contiguous view of one -- its gradient then goes into ``param.grad`` untouched (view
    backward ops move no data).  A copy, a cast, a permuted or re-parametrised weight has its gradient READ by autograd on
    the spot (clone / permute / add): such a call must sum at once -- wrap it in ``guard(False)``."""
    if not _GUARD:
        return True
    for w in weights:
        if w is None:
            continue
        if w.dtype != torch.float32:
            return False
        if w.is_leaf:
            continue
        base = w._base
        if base is None or not base.is_leaf or not w.is_contiguous():
            return False
    return True


class guard:
    """``with guard(ok):`` -- defers inside an open scope only when ``ok`` (= :func:`may_defer` at forward time)."""

    def __init__(self, ok):
        self._p = None if ok else paused()

    def __enter__(self):
        if self._p is not None:
            self._p.__enter__()

    def __exit__(self, *exc):
        if self._p is not None:
            self._p.__exit__(*exc)
        return False


class paused:
    """``with paused():`` -- calls inside reduce at once although a scope is open (their result is read right away)."""

    def __enter__(self):
        if _active is not None:
            _lib.lib().mmu_deferred_pause(1)

    def __exit__(self, *exc):
        if _active is not None:
            _lib.lib().mmu_deferred_pause(0)
        return False


class Scope:
    MAX_JOBS = 1024

    def __init__(self, device):
        self.device = torch.device(device)
        self.table = self.work = None
        self.n_jobs = self.n_work = 0
        self._keep, self._rows, self._work = [], None, None
        self._captured = False

    def reserve(self, max_work=262144):
        self.table = torch.zeros((self.MAX_JOBS, 8), dtype=torch.int64, device=self.device)
        self.work = torch.zeros((max_work, 2), dtype=torch.int32, device=self.device)

    def __enter__(self):
        global _active
        if _active is not None:
            raise RuntimeError("deferred.Scope: another scope is open")
        if self.table is None:
            self.reserve()
        self._keep = []
        _lib.lib().mmu_deferred_begin()
        _active = self
        return self

    def __exit__(self, *exc):
        global _active
        _active = None
        _lib.lib().mmu_deferred_end()
        return False

    def _collect(self):
        import ctypes
        L = _lib.lib()
        n = L.mmu_deferred_jobs(None, 0)
        if n > self.MAX_JOBS:
            raise RuntimeError(f"deferred.Scope: {n} deferred reductions, table holds {self.MAX_JOBS}")
        buf = (ctypes.c_int64 * (8 * max(n, 1)))()
        L.mmu_deferred_jobs(ctypes.cast(buf, ctypes.c_void_p), n)
        rows = [[buf[8 * j + k] for k in range(8)] for j in range(n)]
        work = []
        for j in range(n):      # (the block decomposition of each kind is the library's: csrc/deferred_reduce.hip)
            nb = L.mmu_deferred_job_workgroups(ctypes.byref(buf, 8 * 8 * j))
            work += [[j, b] for b in range(nb)]
        if len(work) > self.work.shape[0]:
            raise RuntimeError(f"deferred.Scope: {len(work)} workgroups, work list holds {self.work.shape[0]}")
        self._rows, self._work = rows, work
        self.n_jobs, self.n_work = len(rows), len(work)

    # which fields of a job row are result pointers (csrc/deferred_reduce.hip: kind -> row layout)
    _DEST_FIELDS = {0: (2,), 1: (2, 3), 2: (2, 3), 3: (2, 3), 4: (2, 3, 4), 5: (2, 3, 4), 6: (2,), 7: (2, 3)}

    def destinations(self):
        """(kind, address) of every result the recorded jobs will write."""
        out = []
        for r in self._rows or []:
            out += [(int(r[0]), int(r[f])) for f in self._DEST_FIELDS.get(int(r[0]), (2,)) if r[f]]
        return out

    def verify_destinations(self, params):
        """The contract of a deferred sum is that NOTHING reads its result before ``launch()`` has run -- true for a
        gradient that autograd stores straight into ``param.grad``, false as soon as the gradient takes a detour (a
        non-leaf weight: a cast, a permuted copy, a re-parametrisation; a parameter used twice, whose second gradient
        AccumulateGrad ADDS to the unfilled first; a tensor hook): the step would then train on stale or garbage
        gradients, silently (it happened once: DESIGN.md 5.1).  Called by ``TrainStep`` after the captured backward:
        every destination must lie inside exactly one ``param.grad`` of the model and none may repeat."""
        spans = sorted((p.grad.data_ptr(), p.grad.data_ptr() + p.grad.numel() * p.grad.element_size(), n)
                       for n, p in params if p.grad is not None)
        starts = [a for a, _, _ in spans]
        import bisect
        seen = {}
        for kind, ptr in self.destinations():
            i = bisect.bisect_right(starts, ptr) - 1
            if i < 0 or not (spans[i][0] <= ptr < spans[i][1]):
                raise RuntimeError(f"deferred.Scope: a deferred reduction of kind {kind} writes to {ptr:#x}, which is no "
                                   "parameter gradient of the model: its result is read (cloned, cast, accumulated) before "
                                   "the deferred launch.  Compute that gradient inside deferred.paused(), or construct "
                                   "TrainStep(deferred_reductions=False).")
            if ptr in seen:
                raise RuntimeError(f"deferred.Scope: two deferred reductions (kinds {seen[ptr]}, {kind}) write to {ptr:#x} "
                                   f"({spans[i][2]}.grad): a parameter that receives two gradients cannot have them deferred")
            seen[ptr] = kind

    def bind(self):
        """Writes the recorded jobs into the device tables (not inside a capture)."""
        if self.n_jobs:
            self.table[:self.n_jobs].copy_(torch.tensor(self._rows, dtype=torch.int64))
            self.work[:self.n_work].copy_(torch.tensor(self._work, dtype=torch.int32))

    def launch(self):
        """Runs every reduction recorded since ``__enter__`` (inside the scope, after the backward pass)."""
        if _active is not self:
            raise RuntimeError("deferred.Scope.launch: call inside the scope")
        self._collect()
        if self.n_jobs == 0:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.bind()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mmu_deferred_launch(self.table.data_ptr(), self.work.data_ptr(), self.n_work,
                                                      torch.cuda.current_stream(self.device).cuda_stream))
        if not capturing:
            self._keep = []     # (stream order: the partial buffers may be reused by later allocations from here on)
        _lib.lib().mmu_deferred_begin()   # further reductions of this scope start a new list
