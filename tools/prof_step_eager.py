"""One warm + N eager training steps (float32, 8 x 3 x 512 x 512) for whole-step counter passes
(tools/pmc_step_traffic.sh): every kernel of the step under rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss
torch.manual_seed(50)
m = MM_Net(num_classes=1).cuda().train()
x = torch.randn(8, 3, 512, 512, device="cuda")
t = (torch.rand(8, 1, 512, 512, device="cuda") > 0.88).float()
for _ in range(1 + (int(sys.argv[1]) if len(sys.argv) > 1 else 1)):
    m.zero_grad(set_to_none=True)
    DICE_BCE_Loss()(m(x), t).backward()
torch.cuda.synchronize()
