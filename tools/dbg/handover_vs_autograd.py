"""The gradient hand-overs (conv3x3_small.GradSlot / SharedGrad / park_extra, morph_coords.OffsetGradSlot,
pointwise.ChannelStatsSlot) against autograd's own accumulation, same kernels otherwise, eager, train mode, at the
benchmark's image size: the forward pass is identical, so every parameter gradient must agree to rounding (the order of
the additions differs).  conv3x3_small.HANDOVER switches all slots off.  Debug aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_unet_amd import conv3x3_small
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.mmunet import MM_Net

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = "cuda"
torch.manual_seed(50)
model = MM_Net(num_classes=1).to(dev).eval()      # (running statistics: nothing in the forward depends on the batch order)
gen = torch.Generator().manual_seed(1)
x = torch.randn(bs, 3, size, size, generator=gen).to(dev)
t = (torch.rand(bs, 1, size, size, generator=gen) > 0.8).float().to(dev)
loss_fn = DICE_BCE_Loss()
grads, losses = {}, {}
for mode in (True, False, True):
    conv3x3_small.HANDOVER = mode
    model.zero_grad(set_to_none=True)
    loss = loss_fn(model(x), t)
    loss.backward()
    torch.cuda.synchronize()
    key = "on" if mode and "on" not in grads else ("off" if not mode else "on2")
    grads[key] = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    losses[key] = float(loss)
print("loss", losses)
for a, b in (("on", "on2"), ("on", "off")):
    ga, gb = grads[a], grads[b]
    assert ga.keys() == gb.keys()
    total = sum(float(v.double().pow(2).sum()) for v in gb.values()) ** 0.5
    rows = sorted(((float((ga[k] - gb[k]).double().norm()) / max(float(gb[k].double().norm()), 1e-30),
                    float(gb[k].double().norm()) / total, k) for k in gb), reverse=True)
    live = [r for r in rows if r[1] > 1e-9]
    print(f"{a} vs {b}: overall rel diff {sum(float((ga[k] - gb[k]).double().pow(2).sum()) for k in ga) ** 0.5 / total:.3e}; "
          f"largest among {len(live)} live tensors:")
    for rel, share, k in live[:5]:
        print(f"  rel {rel:9.3e}  share {share:9.3e}  {k}")
