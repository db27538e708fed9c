"""60 graph-replayed training steps on a fixed synthetic batch set: the loss must fall and stay finite
(exercises running statistics, AdamW state and every fused backward over many steps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.loss import DICE_BCE_Loss
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.train_step import TrainStep, make_optimizer
dev = torch.device("cuda", 0)
torch.manual_seed(50)
model = MM_Net(num_classes=1).to(dev).train()
step = TrainStep(model, DICE_BCE_Loss(), make_optimizer(model, capturable=True), use_graph=True)
g = torch.Generator(device=dev).manual_seed(7)
xs = [torch.randn(4, 3, 256, 256, device=dev, generator=g) for _ in range(4)]
# targets correlated with the input so that there is something to learn
ts = [(x.mean(1, keepdim=True) > 0.4).float() for x in xs]
losses = []
for i in range(60):
    losses.append(float(step(xs[i % 4], ts[i % 4])))
print("first 4:", [round(v, 4) for v in losses[:4]], "last 4:", [round(v, 4) for v in losses[-4:]])
assert all(v == v and abs(v) < 1e4 for v in losses), "non-finite loss"
assert sum(losses[-4:]) < 0.8 * sum(losses[:4]), "loss did not fall"
print("ok")
