"""Data-parallel path on CPU: 2 processes over gloo.  Checks that GradAllReducer (mm-unet_amd/dp.py)
 (a) discovers the live-parameter set once and skips never-used parameters (MM-UNet has 6.76 M of them),
 (b) leaves in every p.grad the MEAN over ranks of the local gradients (== DDP semantics, train.py:52,252),
 (c) keeps replicas bit-identical after optimizer steps, with and without hook-driven overlap,
 (d) broadcast_module_state makes replicas equal at start,
 (e) BatchNorm running statistics stay per replica during training; sync_buffers gives every rank rank 0's (DDP's
     broadcast_buffers state) for validation / checkpoints.
The model is a small pure-ATen network (the MM-UNet blocks themselves are GPU-only)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.b = nn.Conv2d(8, 4, 3, padding=1)
        self.unused = nn.Conv2d(8, 4, 1)          # created, never used (like MMConv.dsc_conv_y)
        self.head = nn.Linear(4, 1)

    def forward(self, x):
        h = torch.relu(self.bn(self.a(x)))
        return self.head(self.b(h).mean(dim=(2, 3)))


def _worker(rank, world, port, overlap, bucket_bytes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mm_unet_amd.dp import GradAllReducer, broadcast_module_state, sync_buffers
        torch.manual_seed(100 + rank)            # different init per rank on purpose
        net = Net()
        broadcast_module_state(net)
        ref = Net()
        ref.load_state_dict(net.state_dict())
        red = GradAllReducer(net, bucket_bytes=bucket_bytes, overlap=overlap)
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        out = {}
        for step in range(3):
            g = torch.Generator().manual_seed(1000 * step + rank)
            x = torch.randn(4, 3, 8, 8, generator=g)
            t = torch.randn(4, 1, generator=g)
            loss = ((net(x) - t) ** 2).mean()
            loss.backward()
            red.finish()
            # reference: local grads of an identical replica, averaged by hand with all_reduce
            ref.zero_grad(set_to_none=True)
            ((ref(x) - t) ** 2).mean().backward()
            for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                if pr.grad is None:
                    assert p.grad is None, f"{n} should have no gradient"
                    continue
                want = pr.grad.clone()
                dist.all_reduce(want)
                want /= world
                assert torch.allclose(p.grad, want, rtol=1e-6, atol=1e-7), f"step {step} {n}"
            opt.step()
            red.zero_grad()
            with torch.no_grad():                # keep the reference replica in lock-step
                for p, pr in zip(net.parameters(), ref.parameters()):
                    pr.copy_(p)
                for bf, br in zip(net.buffers(), ref.buffers()):
                    br.copy_(bf)
        out["live"] = red.live_names
        out["payload"] = red.payload_bytes()
        out["n_buckets"] = len(red.buckets)
        out["w"] = torch.cat([p.detach().flatten() for p in net.parameters()]).numpy().copy()  # by value
        out["bn_before"] = net.bn.running_mean.numpy().copy()   # per-replica statistics (different data per rank)
        sync_buffers(net)                                        # what DDP's broadcast_buffers leaves behind
        out["bn_after"] = net.bn.running_mean.numpy().copy()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,bucket_bytes", [(True, 16 << 20), (False, 16 << 20), (True, 512)])
def test_grad_allreduce_two_ranks_gloo(overlap, bucket_bytes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if overlap else 0) + (2 if bucket_bytes < 1024 else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    live0, live1 = res[0]["live"], res[1]["live"]
    assert live0 == live1 and not any(n.startswith("unused") for n in live0)
    assert res[0]["payload"] == sum(p.numel() for n, p in Net().named_parameters() if not n.startswith("unused")) * 4
    if bucket_bytes < 1024:
        assert res[0]["n_buckets"] > 1           # several buckets, launched from hooks as they fill
    assert (res[0]["w"] == res[1]["w"]).all(), "replicas diverged"
    assert not (res[0]["bn_before"] == res[1]["bn_before"]).all()       # running statistics are per replica ...
    assert (res[1]["bn_after"] == res[0]["bn_before"]).all() and (res[0]["bn_after"] == res[0]["bn_before"]).all()


def _loss_and_data(step, rank):
    g = torch.Generator().manual_seed(1000 * step + rank)
    x = torch.randn(4, 3, 8, 8, generator=g)
    t = (torch.rand(4, 1, generator=g) > 0.5).float()
    return x, t


class _DiceBceHead(nn.Module):
    """loss.py's Dice + BCE on the head's logits: the Dice sums run over the WHOLE (per-replica) batch, like the
    reference's DICE_BCE_Loss (loss.py:12-14) -- the one loss term that is not a mean of per-sample terms."""

    def forward(self, logits, target):
        p = torch.sigmoid(logits)
        dice = 1 - (2 * (p * target).sum() + 1) / (p.sum() + target.sum() + 1)
        return dice + nn.functional.binary_cross_entropy(p, target)


def _train_step_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mm_unet_amd.dp import broadcast_module_state
        from mm_unet_amd.train_step import TrainStep, make_optimizer
        torch.manual_seed(7 + rank)              # different init per rank: the broadcast must make them equal
        net = Net()
        broadcast_module_state(net)
        step = TrainStep(net, _DiceBceHead(), make_optimizer(net, lr=1e-2, fused=False), overlap=True, use_graph=False,
                         bucket_bytes=512)       # several buckets, launched from the hooks during backward
        losses = []
        for s in range(4):
            x, t = _loss_and_data(s, rank)
            losses.append(float(step(x, t)))
        q.put((rank, {"w": {k: v.detach().clone().numpy() for k, v in net.state_dict().items()}, "losses": losses}))
    finally:
        dist.destroy_process_group()


def test_train_step_two_ranks_gloo_equals_single_process_emulation():
    """the whole data-parallel step (TrainStep, eager mode, hook-driven overlapped all-reduce in several
    buckets, AdamW with the reference's parameter groups) over 2 gloo ranks for 4 steps, against ONE process that
    reproduces the semantics by hand: per-replica BatchNorm batch statistics and per-replica batch-global Dice sums
    (each replica's forward / backward on its own 4 samples), gradients averaged over the replicas, one AdamW step.
    Post-step weights agree to float32 rounding; the never-used parameter keeps its initial value."""
    from mm_unet_amd.train_step import make_optimizer
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_step_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # --- the same four steps in one process ---------------------------------------------------------------------
    torch.manual_seed(7)                         # rank 0's initial weights (what broadcast_module_state distributes)
    ref = Net()
    replicas = [Net() for _ in range(world)]     # one module per replica: per-replica BatchNorm buffers
    for r_ in replicas:
        r_.load_state_dict(ref.state_dict())
    opt = make_optimizer(ref, lr=1e-2, fused=False)
    loss_fn = _DiceBceHead()
    ref_losses = []
    for s in range(4):
        grads = None
        for r, rep in enumerate(replicas):
            with torch.no_grad():
                for pr, p in zip(rep.parameters(), ref.parameters()):
                    pr.copy_(p)
            rep.zero_grad(set_to_none=True)
            x, t = _loss_and_data(s, r)
            loss = loss_fn(rep(x), t)
            loss.backward()
            if r == 0:
                ref_losses.append(float(loss))
            gs = [None if p.grad is None else p.grad.clone() for p in rep.parameters()]
            grads = gs if grads is None else [a if b_ is None else a + b_ for a, b_ in zip(grads, gs)]
        for p, g_ in zip(ref.parameters(), grads):
            p.grad = None if g_ is None else g_ / world
        opt.step()
    w0, w1 = res[0]["w"], res[1]["w"]
    for k, v in ref.state_dict().items():
        if "running_" in k or "num_batches" in k:
            continue                              # per-replica buffers (checked in the test above)
        assert (w0[k] == w1[k]).all(), f"{k}: replicas diverged"
        assert abs(w0[k] - v.numpy()).max() <= 2e-6 * max(1.0, float(abs(v).max())), k
    assert max(abs(a - b_) for a, b_ in zip(res[0]["losses"], ref_losses)) < 1e-6
    init = Net.__new__(Net)
    torch.manual_seed(7)
    init = Net()
    assert (w0["unused.weight"] == init.unused.weight.detach().numpy()).all()
