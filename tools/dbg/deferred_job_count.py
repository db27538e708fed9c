"""How many final sums does the captured training step defer (deferred.Scope)?  264 at the end of round 4."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd import train_step
from mm_unet_amd.loss import DICE_BCE_Loss
torch.manual_seed(0)
m = MM_Net(num_classes=1).cuda().train()
opt = train_step.make_optimizer(m, capturable=True)
st = train_step.TrainStep(m, DICE_BCE_Loss(), opt, use_graph=True)
x = torch.randn(8, 3, 512, 512, device="cuda"); t = (torch.rand(8, 1, 512, 512, device="cuda") > 0.9).float()
for _ in range(3):
    st(x, t)
torch.cuda.synchronize()
print("deferred jobs:", st._scope_captured.n_jobs)
