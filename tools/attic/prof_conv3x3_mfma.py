"""Timing legs for csrc/conv3x3_mfma.hip vs ATen (MIOpen) at CBAM's shape and two Unet shapes (run under tools/kstats.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from mm_unet_amd.conv3x3_mfma import conv3x3_mfma
for (B, Cin, Cout, H, W) in [(8, 64, 64, 256, 256), (8, 128, 128, 128, 128), (8, 256, 256, 64, 64)]:
    x = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).requires_grad_()
    b = torch.randn(Cout, device="cuda", requires_grad=True)
    g = torch.randn(B, Cout, H, W, device="cuda")
    for _ in range(5):
        conv3x3_mfma(x, w, b).backward(g)
    for _ in range(5):
        F.conv2d(x, w, b, padding=1).backward(g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    print(B, Cin, Cout, H, W, "max err mfma", float((conv3x3_mfma(x, w, b) - ref).abs().max()),
          "aten", float((F.conv2d(x, w, b, padding=1) - ref).abs().max()), flush=True)
torch.cuda.synchronize()
