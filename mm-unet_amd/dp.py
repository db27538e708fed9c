"""Data-parallel gradient exchange for MM-UNet: one process per GPU, ``torch.distributed``
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" on CPU for tests).

Replaces what ``accelerator.prepare`` + ``accelerator.backward`` do in the reference
(train.py:52,252: DistributedDataParallel with NCCL bucketed all-reduce) with a design sized
for this model and this fabric:

  * MM-UNet has 16.3 M parameters but only 9.56 M ever receive a gradient (``MMConv.dsc_conv_y`` and
    the ``_b``/``_s`` Mamba branches of the 47 "v1" blocks are created and never used).
    Stock DDP needs ``find_unused_parameters=True`` (a graph walk every step).  Here the live set is
    discovered ONCE (first backward), is identical on every rank by construction, and only those
    38 MB are ever exchanged.
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU) and the payload is small, so the exchange is
    latency-bound: few, large buckets (default 16 MiB -> 3 collectives) rather than many small ones.
    A bucket is packed with ONE multi-tensor copy (``torch._foreach_copy_``) and its all-reduce is
    issued on a side stream from a post-accumulate hook as soon as its last gradient is ready,
    overlapping the rest of the backward; after the wait the averaged values are copied back.
  * autograd ASSIGNS gradients (``zero_grad(set_to_none=True)``): letting it accumulate into
    pre-existing bucket views costs one ``add_`` launch per parameter per step (1,069 launches, 8 ms
    of GPU time and as much host time on MI355X, measured).
  * with a single rank nothing is copied or launched at all;
  * under HIP-graph replay (``static_grads=True``, what ``bench.py`` runs) the hooks are off: forward + backward
    are one graph, and the exchange -- pack, all-reduce, unpack, three buckets -- runs after the replay, not
    overlapped with the backward (38 MB over xGMI is < 1 % of a 52 ms step);
  * per-replica BatchNorm batch statistics and per-replica loss, as under DDP (no SyncBN in the reference).
    BUFFERS: DDP's default ``broadcast_buffers=True`` re-broadcasts rank 0's BatchNorm running statistics at
    every forward, so under the reference every rank validates / checkpoints with rank 0's statistics.  Here
    nothing is exchanged per step (18,968 floats that only matter in eval mode); call ``sync_buffers`` before
    validation and before saving a checkpoint to get the same state.
"""
import torch
import torch.distributed as dist


def broadcast_module_state(module, src=0, group=None):
    """DDP construction semantics: parameters and buffers of rank ``src`` everywhere."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


def sync_buffers(module, src=0, group=None):
    """Rank ``src``'s buffers (BatchNorm running statistics, ``num_batches_tracked``) everywhere: what DDP's
    ``broadcast_buffers=True`` leaves every rank with.  Call before validation / checkpointing."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in module.buffers():
            dist.broadcast(t, src=src, group=group)


class GradAllReducer:
    """Averages the live gradients of ``module`` across the process group.

    Usage per step:  loss.backward();  reducer.finish();  optimizer.step();  reducer.zero_grad()
    The first backward runs un-overlapped (it discovers the live set and builds the buckets).
    ``static_grads=True``: gradients are persistent tensors (HIP-graph replay) -- never set to None.
    """

    def __init__(self, module, group=None, bucket_bytes=16 << 20, overlap=True, static_grads=False):
        self.module = module
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.overlap = overlap and not static_grads
        self.static_grads = static_grads
        self.buckets = None          # list of dict(flat, views, params, pending, handle)
        self._hooks = []
        self._side = None
        self.live_names = None

    # -- one-time layout ---------------------------------------------------------------------------
    def _build(self):
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        live = [(n, p) for n, p in named if p.grad is not None]
        self.live_names = [n for n, _ in live]
        if self.world > 1:
            # the live set is structural; verify cheaply that every rank found the same one
            sig = torch.tensor([len(live), sum(p.numel() for _, p in live)], dtype=torch.int64,
                               device=live[0][1].device)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not torch.equal(lo, hi):
                raise RuntimeError("GradAllReducer: ranks disagree on the set of parameters that receive gradients")
        # buckets in reverse registration order ~ the order backward produces gradients
        groups, cur, cur_bytes = [], [], 0
        for n, p in reversed(live):
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > self.bucket_bytes or cur[0][1].dtype != p.dtype):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append((n, p))
            cur_bytes += nb
        if cur:
            groups.append(cur)
        self.buckets = []
        for members in groups:
            p0 = members[0][1]
            total = sum(p.numel() for _, p in members)
            flat = torch.zeros(total, dtype=p0.dtype, device=p0.device) if self.world > 1 else None
            views, off = [], 0
            for _, p in members:
                if flat is not None:
                    views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self.buckets.append(dict(flat=flat, views=views, params=[p for _, p in members], nbytes=total * p0.element_size(),
                                     pending=len(members), handle=None))
        if self.overlap and self.world > 1:
            for bi, b in enumerate(self.buckets):
                for p in b["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
            if self.buckets[0]["params"][0].is_cuda:
                self._side = torch.cuda.Stream(device=self.buckets[0]["params"][0].device)

    def _make_hook(self, bi):
        def hook(_param):
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        """Pack the bucket (one multi-tensor copy) and start its all-reduce."""
        grads = [p.grad for p in b["params"]]

        def go():
            torch._foreach_copy_(b["views"], grads)
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(b["flat"].device))
            with torch.cuda.stream(self._side):
                go()
        else:
            go()

    # -- per step -------------------------------------------------------------------------------------
    def finish(self):
        """Call after ``backward``: on return every live ``p.grad`` holds the mean over ranks."""
        first = self.buckets is None
        if first:
            self._build()
        if self.world == 1:
            return
        for b in self.buckets:
            if b["handle"] is None:      # first step / overlap off / hook did not fire
                self._launch(b)
        for b in self.buckets:
            b["handle"].wait()
            b["handle"] = None
            b["pending"] = len(b["params"])
        if self._side is not None:
            torch.cuda.current_stream(self.buckets[0]["flat"].device).wait_stream(self._side)
        for b in self.buckets:
            b["flat"].mul_(1.0 / self.world)
            torch._foreach_copy_([p.grad for p in b["params"]], b["views"])

    def zero_grad(self):
        if not self.static_grads:
            self.module.zero_grad(set_to_none=True)

    def payload_bytes(self):
        return 0 if self.buckets is None else sum(b["nbytes"] for b in self.buckets)

    def live_parameters(self):
        return [p for b in (self.buckets or []) for p in b["params"]]
