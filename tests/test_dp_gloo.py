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
