import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd.conv3x3_small import conv3x3_small
import torch.nn.functional as F
for (B, Cin, H, W) in [(8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32), (8, 64, 256, 256), (8, 512, 16, 16)]:
    x = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    w = torch.randn(6, Cin, 3, 3, device="cuda", requires_grad=True)
    b = torch.randn(6, device="cuda", requires_grad=True)
    g = torch.randn(B, 6, H, W, device="cuda")
    for _ in range(3):
        conv3x3_small(x, w, b).backward(g)
        F.conv2d(x, w, b, padding=1).backward(g)
torch.cuda.synchronize()
