"""Times conv3x3_small forward / backward kernels alone at MM-UNet's offset-conv shapes (run on the GPU box).
usage: python tools/bench_conv3x3s.py [fwd|bwd]   -- knobs through the environment (MMU_CONV3X3S_*), one process per setting."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd  # noqa
from mm_unet_amd.conv3x3_small import conv3x3_small
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
shapes = [(8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32), (8, 512, 16, 16), (8, 64, 256, 256)]
dev = "cuda"
res = []
for B, C, H, W in shapes:
    xs = [torch.randn(B, C, H, W, device=dev, requires_grad=True) for _ in range(4)]   # rotate: 4 x input > L2
    w = (torch.randn(6, C, 3, 3, device=dev) * 0.1).requires_grad_()
    b = torch.randn(6, device=dev, requires_grad=True)
    g = torch.randn(B, 6, H, W, device=dev)
    def run(i):
        out = conv3x3_small(xs[i % 4], w, b)
        if what == "bwd":
            out.backward(g)
    for i in range(8):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    if what == "fwd":
        with torch.no_grad():
            e0.record()
            for i in range(n):
                run(i)
            e1.record()
    else:
        e0.record()
        for i in range(n):
            run(i)
        e1.record()
    torch.cuda.synchronize()
    res.append(f"{B}x{C}x{H}x{W}: {e0.elapsed_time(e1) / n * 1e3:7.1f} us")
print(what, {k: v for k, v in os.environ.items() if k.startswith("MMU_CONV3X3S")}, " | ".join(res))
