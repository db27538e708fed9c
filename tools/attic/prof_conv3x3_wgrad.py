"""Weight gradient of the dense 3x3 conv: matrix-core kernel vs MIOpen at CBAM's shape and two Unet shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd.conv3x3_mfma as cm
for (B, Cin, Cout, H, W) in [(8, 64, 64, 256, 256), (8, 128, 128, 128, 128), (8, 512, 512, 32, 32)]:
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)).requires_grad_()
    g = torch.randn(B, Cout, H, W, device="cuda")
    for on in (True, False):
        cm.WGRAD_MFMA = on
        for _ in range(5):
            w.grad = None
            cm.conv3x3_mfma(x, w, None).backward(g)
torch.cuda.synchronize()
