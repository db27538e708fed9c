"""The offset convolution's input gradient (Cin -> 6, 3 x 3) at the step's shapes, under tools/kstats.sh; compare with
MMUNET_OFFSET_DGRAD_MFMA=0 (the direct vector-pipe kernel): python3 tools/prof_offset_conv.py [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from mm_unet_amd.conv3x3_small import conv3x3_small
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
for (B, Cin, H) in ((8, 64, 128), (8, 128, 64), (8, 256, 32), (8, 64, 256), (8, 32, 256), (8, 512, 16)):
    x = torch.randn(B, Cin, H, H, device=dev, requires_grad=True)
    w = torch.randn(6, Cin, 3, 3, device=dev) * 0.1
    b = torch.randn(6, device=dev)
    g = torch.randn(B, 6, H, H, device=dev)
    for _ in range(reps):
        x.grad = None
        conv3x3_small(x, w, b).backward(g)
    torch.cuda.synchronize()
print("ok")
