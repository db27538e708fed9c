"""The stem convolution (8 x 3 x 512 x 512 -> 64 channels, 7 x 7, stride 2) on csrc/stem7_mfma.hip and in MIOpen, under
tools/kstats.sh: python3 tools/prof_stem.py [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from mm_unet_amd import stem7
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
conv = torch.nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False).cuda()
x = torch.randn(8, 3, 512, 512, device="cuda")
dout = torch.randn(8, 64, 256, 256, device="cuda")
for _ in range(reps):
    conv.weight.grad = None
    stem7.stem_conv(conv, x).backward(dout)
    conv.weight.grad = None
    conv(x).backward(dout)
torch.cuda.synchronize()
print("ok")
