"""``torch.optim.AdamW.step()`` as two launches for a training step whose tensors are static (csrc/adamw_multi.hip).

``MultiTensorAdamW(optimizer)`` wraps a ``torch.optim.AdamW`` (what train.py:197-201 builds through timm) and runs its
update on the optimizer's OWN state tensors (``exp_avg``, ``exp_avg_sq``, ``step``) and hyper-parameters: ``state_dict()``
/ checkpoints / schedulers keep working on the wrapped optimizer; only the arithmetic of ``step()`` moves into one kernel
over a device table of addresses.  The table is bound to the tensors' addresses: made for ``TrainStep(use_graph=True)``,
where parameters, gradients and state are the same storage at every replay: ``reserve()`` before the capture,
``plan()`` + ``launch()`` inside it after the backward pass, ``bind()`` after it (the table's CONTENT may be written after
the capture -- a captured launch only keeps the table's address).  Requirements (checked): float32 contiguous CUDA tensors, initialised state with ``step`` tensors on the device
(``capturable=True``), no amsgrad / maximize, learning rate a device tensor or a number.
"""
import struct

import torch

from . import _lib

CHUNK = 4096


def supported(optimizer):
    if not isinstance(optimizer, torch.optim.AdamW):
        return False
    for g in optimizer.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable"):
            return False
        for p in g["params"]:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                return False
    return True


class MultiTensorAdamW:
    def __init__(self, optimizer):
        if not supported(optimizer):
            raise RuntimeError("MultiTensorAdamW: a torch.optim.AdamW over float32 contiguous CUDA parameters without "
                               "amsgrad / maximize is required")
        self.optimizer = optimizer
        self.table = self.work = None
        self._keep = []
        self.n_tensors = self.n_work = 0
        g0 = optimizer.param_groups[0]
        self.beta1, self.beta2, self.eps = float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"])
        for g in optimizer.param_groups:
            if (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) != (self.beta1, self.beta2, self.eps):
                raise RuntimeError("MultiTensorAdamW: betas / eps must be the same in every parameter group")

    def _entries(self):
        rows, work, keep = [], [], []
        for g in self.optimizer.param_groups:
            lr = g["lr"]
            if not isinstance(lr, torch.Tensor):
                lr = torch.tensor(float(lr), dtype=torch.float32, device=g["params"][0].device)
            if lr.dtype != torch.float32 or not lr.is_cuda:
                raise RuntimeError("MultiTensorAdamW: the learning rate must be a float32 device tensor or a number")
            keep.append(lr)
            wd_bits = struct.unpack("<I", struct.pack("<f", float(g["weight_decay"])))[0]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.optimizer.state.get(p)
                if not st or not isinstance(st.get("step"), torch.Tensor) or not st["step"].is_cuda:
                    raise RuntimeError("MultiTensorAdamW: optimizer state missing or step counters not on the device "
                                       "(run one optimizer.step() first; build the optimizer with capturable=True)")
                gr, m, v, stp = p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"]
                for t in (gr, m, v):
                    if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel() or t.device != p.device:
                        raise RuntimeError("MultiTensorAdamW: gradients and state must be float32, contiguous, on the parameter's device")
                if stp.dtype != torch.float32 or stp.numel() != 1:
                    raise RuntimeError("MultiTensorAdamW: step counters must be float32 scalars")
                t_idx = len(rows)
                rows.append([p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), stp.data_ptr(), p.numel(),
                             lr.data_ptr(), wd_bits])
                work += [[t_idx, c] for c in range((p.numel() + CHUNK - 1) // CHUNK)]
        return rows, work, keep

    def reserve(self):
        """Allocates the device table / work list for EVERY parameter of the optimizer (an upper bound: which of them
        receive gradients is only known after a backward pass).  Call outside any capture: an allocation inside one would
        come with a captured fill that wipes the table at every replay."""
        dev = self.optimizer.param_groups[0]["params"][0].device
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        self.table = torch.zeros((max(len(params), 1), 8), dtype=torch.int64, device=dev)
        self.work = torch.zeros((max(sum((p.numel() + CHUNK - 1) // CHUNK for p in params), 1), 2), dtype=torch.int32,
                                device=dev)

    def plan(self):
        """Which tensors the step covers = the parameters that have a gradient NOW (host side only: legal inside a
        capture, after the captured backward has defined the gradients).  ``launch()`` uses these counts, ``bind()``
        writes the addresses."""
        if self.table is None:
            raise RuntimeError("MultiTensorAdamW.plan: reserve() first")
        self._rows, self._work, self._keep = self._entries()
        self.n_tensors, self.n_work = len(self._rows), len(self._work)
        if self.n_tensors > self.table.shape[0] or self.n_work > self.work.shape[0]:
            raise RuntimeError("MultiTensorAdamW.plan: more tensors than reserve() saw")

    def bind(self):
        """Writes the planned addresses into the device table (outside any capture; before the first replay)."""
        if self.n_tensors:
            self.table[:self.n_tensors].copy_(torch.tensor(self._rows, dtype=torch.int64))
            self.work[:self.n_work].copy_(torch.tensor(self._work, dtype=torch.int32))

    def launch(self):
        """The update, on the current stream (capturable: two kernel launches reading the device table)."""
        if self.table is None or self.n_tensors == 0:
            raise RuntimeError("MultiTensorAdamW.launch: reserve() and plan() first")
        p = _lib.AdamWParams()
        p.n_tensors, p.n_work = self.n_tensors, self.n_work
        p.table, p.work = self.table.data_ptr(), self.work.data_ptr()
        p.beta1, p.beta2, p.eps = self.beta1, self.beta2, self.eps
        dev = self.table.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mmu_adamw_multi(p, torch.cuda.current_stream(dev).cuda_stream))
