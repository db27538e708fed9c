"""One MM-UNet training step with the semantics of the reference's ``train_one_epoch`` body
(train.py:35-54): forward -> loss -> backward (+ DP gradient all-reduce, train.py:52/252) -> AdamW
step -> zero_grad.  The per-step host syncs of the reference (``float(loss)`` and the MONAI metrics on
the training batch, train.py:42-48,61) are not part of the step; the loss is returned as a device tensor.

``make_optimizer`` mirrors the reference's ``create_optimizer_v2('adamw', lr 1e-3, weight_decay .05,
betas (.9,.95))`` (train.py:197-201): decoupled weight decay, none on 1-D parameters / biases, groups in
timm's order.

``use_graph=True`` captures the step into a HIP graph (``torch.cuda.CUDAGraph``): a step is ~7,700
kernel launches and the host needs as long to issue them (108 ms, measured) as the GPU needs to run
them, so replaying a graph is what keeps the GPU fed once kernels get faster.  One rank: the whole step
(forward, loss, backward, AdamW) is one graph.  Several ranks: forward+backward is the graph; the
gradient exchange (RCCL) and the optimizer run after the replay.  Inputs are copied into static
buffers; shapes are fixed at capture time.
"""
import torch

from . import adamw_multi, deferred
from .dp import GradAllReducer


def make_optimizer(module, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.95), fused=None, capturable=False):
    """AdamW as the reference builds it (train.py:197-201: timm ``create_optimizer_v2(model, opt='adamw', lr,
    weight_decay=.05, betas=(.9,.95))``): decoupled weight decay, none on parameters with ``ndim <= 1`` or a name
    ending in ``.bias``.

    timm is not in the image, so the rule is restated from its published source (``optim_factory.py``:
    ``param_groups_weight_decay`` in 0.9.7 = pyproject.toml's pin; ``add_weight_decay`` in 0.4.12 =
    requirements.txt's pin, which tests ``len(shape) == 1`` and therefore decays MMConv's 0-dim ``altho``):
    **parity unpinned**.  Both versions return the groups in the order ``[no_decay (wd 0), decay]``;
    ``optimizer.state_dict()`` numbers parameters in group order, so a reference-trained ``optimizer.bin`` loads
    (checkpoint.py) only if that order is kept -- it is.

    ``capturable=True`` (the optimizer step is recorded into a HIP graph, ``TrainStep(use_graph=True)`` on one
    rank): the learning rate is a device tensor.  A Python-float lr is passed to the fused kernel by value and
    baked into the captured launch -- the reference steps a warm-up + cosine scheduler every epoch from
    ``warmup_start_lr = 0`` (src/optimizer.py:26,56), i.e. a graph captured in epoch 0 would train at lr 0 for
    ever.  Schedulers (and ``set_lr`` below) update the tensor in place, replays see it."""
    decay, no_decay = [], []
    for name, p in module.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if p.ndim <= 1 or name.endswith(".bias") else decay).append(p)
    if fused is None:
        fused = all(p.is_cuda for p in decay + no_decay)
    capturable = capturable and fused
    if capturable:
        lr = torch.tensor(float(lr), dtype=torch.float32, device=(decay + no_decay)[0].device)
    return torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0},
                              {"params": decay, "weight_decay": weight_decay}], lr=lr, betas=betas, fused=fused,
                             capturable=capturable)


def set_lr(optimizer, lr):
    """Sets the learning rate of every group; in place when it is a device tensor (captured optimizer step)."""
    for g in optimizer.param_groups:
        if isinstance(g["lr"], torch.Tensor):
            g["lr"].fill_(float(lr))
        else:
            g["lr"] = float(lr)


class TrainStep:
    def __init__(self, model, loss_fn, optimizer, group=None, amp_dtype=None, bucket_bytes=16 << 20, overlap=True,
                 use_graph=False, multi_tensor_adamw=True, deferred_reductions=True):
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.amp_dtype = amp_dtype
        self.use_graph = use_graph
        self.multi_tensor_adamw = multi_tensor_adamw
        self.deferred_reductions = deferred_reductions
        self._scope = None
        self.reducer = GradAllReducer(model, group=group, bucket_bytes=bucket_bytes, overlap=overlap,
                                      static_grads=use_graph)
        self._graph = None
        self._warm = 0

    def forward_backward(self, images, targets):
        if self.amp_dtype is not None:
            with torch.autocast(images.device.type, dtype=self.amp_dtype):
                logits = self.model(images)
            loss = self.loss_fn(logits.float(), targets)
        else:
            loss = self.loss_fn(self.model(images), targets)
        scope = self._scope
        if scope is not None:   # captured step: the final sums of the weight gradients as ONE launch (deferred.py)
            with scope:
                loss.backward()
                scope.launch()
        else:
            loss.backward()
        return loss.detach()

    def _eager(self, images, targets):
        loss = self.forward_backward(images, targets)
        self.reducer.finish()
        self.optimizer.step()
        self.reducer.zero_grad()
        return loss

    def _capture(self, images, targets):
        dev = images.device
        self._x = images.clone()
        self._t = targets.clone()
        whole = self.reducer.world == 1
        self.model.zero_grad(set_to_none=True)
        self._graph = torch.cuda.CUDAGraph()
        # one rank: AdamW is part of the graph -- as two launches over a device table of the (static) parameter / gradient
        # / state addresses (adamw_multi.py) where the optimizer allows it, as optimizer.step() (41 launches) otherwise
        self._scope = None
        if self.deferred_reductions and all(p.dtype == torch.float32 for p in self.model.parameters()):
            self._scope = deferred.Scope(dev)
            self._scope.reserve()
        self._adamw = None
        # (several ranks: the gradients are static storage too -- the same two launches run after the all-reduce, outside
        #  the graph, instead of optimizer.step()'s 41; only with the learning rate as a device tensor, which a scheduler
        #  updates in place)
        if self.multi_tensor_adamw and adamw_multi.supported(self.optimizer) and \
                all(isinstance(s.get("step"), torch.Tensor) and s["step"].is_cuda for s in self.optimizer.state.values()) \
                and len(self.optimizer.state) > 0 and \
                all(isinstance(g["lr"], torch.Tensor) and g["lr"].is_cuda for g in self.optimizer.param_groups):
            # (a Python-float learning rate would be copied to the device inside the capture -- a pageable host-to-device
            #  copy in a capturing stream -- and baked into the graph: such optimizers keep optimizer.step())
            self._adamw = adamw_multi.MultiTensorAdamW(self.optimizer)
            self._adamw.reserve()
        with torch.cuda.graph(self._graph):
            self._loss = self.forward_backward(self._x, self._t)
            if whole:
                if self._adamw is not None:
                    self._adamw.plan()         # (the parameters the captured backward has just given gradients)
                    self._adamw.launch()
                else:
                    self.optimizer.step()
        if self._adamw is not None:
            if not whole:
                self._adamw.plan()             # (the parameters the captured backward has given gradients)
            self._adamw.bind()                 # addresses into the table the captured launches read
        if self._scope is not None:
            # the deferred sums' contract, checked on what the capture actually recorded: raises (before any replay) when
            # a result is not a parameter gradient or is written twice
            self._scope.verify_destinations(list(self.model.named_parameters()))
            self._scope.bind()                 # (the same for the deferred reductions' job table)
            self._scope_captured, self._scope = self._scope, None   # its tables and partial buffers live with the graph
        self._whole = whole
        torch.cuda.synchronize(dev)

    def refresh_optimizer_state(self):
        """After ``optimizer.load_state_dict`` (checkpoint.load_state) on a step that is already captured: torch puts the
        loaded moments into NEW tensors, and a captured ``optimizer.step()`` would go on updating the old ones.  The
        table-driven AdamW only needs its address table rewritten -- no re-capture.  Returns False when the captured step
        runs ``optimizer.step()`` itself (then the graph must be captured again: ``self._graph = None``)."""
        if self._graph is None:
            return True
        if getattr(self, "_adamw", None) is None:
            return not getattr(self, "_whole", False)
        self._adamw.plan()
        self._adamw.bind()
        return True

    def __call__(self, images, targets):
        if not self.use_graph:
            return self._eager(images, targets)
        if self._graph is None:
            # eager warm-up first: MIOpen kernel selection, lazy optimizer state, live-gradient discovery
            if self._warm < 2:
                self._warm += 1
                loss = self.forward_backward(images, targets)
                self.reducer.finish()
                self.optimizer.step()
                self.model.zero_grad(set_to_none=True)
                return loss
            self._capture(images, targets)
        self._x.copy_(images)
        self._t.copy_(targets)
        self._graph.replay()
        if not self._whole:
            self.reducer.finish()
            if self._adamw is not None:
                self._adamw.launch()
            else:
                self.optimizer.step()
        return self._loss


class InferStep:
    """Forward-only counterpart of TrainStep for serving (BASELINE config 2: inference, bs 8, 3x512x512):
    ``model.eval()`` forward under ``no_grad``, captured into a HIP graph after one eager warm-up call
    (MIOpen kernel selection) and replayed from then on -- an eager MM_Net forward is ~2,000 launches and
    host-bound (150 ms for bs 8; the GPU needs a fraction of that).  Shapes are fixed at capture time."""

    def __init__(self, model, amp_dtype=None, use_graph=True):
        self.model, self.amp_dtype, self.use_graph = model.eval(), amp_dtype, use_graph
        self._graph = None
        self._warm = False

    def _forward(self, images):
        with torch.no_grad():
            if self.amp_dtype is not None:
                with torch.autocast(images.device.type, dtype=self.amp_dtype):
                    return self.model(images)
            return self.model(images)

    def __call__(self, images):
        if not self.use_graph:
            return self._forward(images)
        if self._graph is None:
            if not self._warm:
                self._warm = True
                return self._forward(images)
            self._x = images.clone()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = self._forward(self._x)
            torch.cuda.synchronize(images.device)
        self._x.copy_(images)
        self._graph.replay()
        return self._out
