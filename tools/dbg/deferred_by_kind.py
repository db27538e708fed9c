"""Time of the deferred-reduction launch of one MM_Net backward pass (3x512x512, bs 8), in total and per job kind
(debug aid: which kind holds the launch up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mm_unet_amd  # noqa
from mm_unet_amd import deferred, _lib
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss

torch.manual_seed(50)
dev = "cuda"
model = MM_Net(num_classes=1).to(dev).train()
loss_fn = DICE_BCE_Loss()
x = torch.randn(8, 3, 512, 512, device=dev)
t = (torch.rand(8, 1, 512, 512, device=dev) > 0.5).float()
scope = deferred.Scope(dev)
scope.reserve()
for it in range(2):
    model.zero_grad(set_to_none=True)
    loss = loss_fn(model(x), t)
    with scope:
        loss.backward()
        scope._collect()
        rows, work = scope._rows, scope._work
        kinds = sorted(set(r[0] for r in rows))
        print("jobs", len(rows), "workgroups", len(work), {k: sum(1 for r in rows if r[0] == k) for k in kinds},
              {k: sum(1 for w in work if rows[w[0]][0] == k) for k in kinds})
        if it == 1:
            print("kind 0 bytes", sum(r[4] * r[5] * 4 for r in rows if r[0] == 0) / 1e6, "MB; slabs",
                  sorted(set(r[5] for r in rows if r[0] == 0)))
            print("kind 1 bytes", sum(r[4] * ((r[6] * 10 + 3) & ~3) * r[5] * 4 for r in rows if r[0] == 1) / 1e6, "MB")
            print("kind 4 bytes", sum(r[5] * (r[6] & 0xffffffff) * 128 for r in rows if r[0] == 4) / 1e6, "MB")

            def timed(sel, label):
                sub = [w for w in work if sel(rows[w[0]])]
                if not sub:
                    return
                scope.table[:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
                scope.work[:len(sub)].copy_(torch.tensor(sub, dtype=torch.int32))
                torch.cuda.synchronize()
                st = torch.cuda.current_stream().cuda_stream
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for rep in range(3):
                    e0.record()
                    _lib.check(_lib.lib().mmu_deferred_launch(scope.table.data_ptr(), scope.work.data_ptr(), len(sub), st))
                    e1.record()
                    torch.cuda.synchronize()
                print(f"{label:28s} workgroups {len(sub):7d}  {e0.elapsed_time(e1) * 1e3:8.1f} us")
            timed(lambda r: True, "all")
            for k in kinds:
                timed(lambda r, k=k: r[0] == k, f"kind {k}")
            # the scan jobs one by one
            for j, r in enumerate(rows):
                if r[0] in (4, 5) and (r[5] >= 512):
                    timed(lambda rr, r=r: rr is r, f"kind {r[0]} rows {r[5]} dim {r[6] & 0xffffffff}")
        scope.launch()
