"""causal conv1d forward / backward at the headline shape on batch-major and channel-major storage (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd  # noqa
from mm_unet_amd import causal_conv1d_hip as cc
dev = "cuda"
b, d, l = 8, 128, 65536
w, cb = torch.randn(d, 4, device=dev), torch.randn(d, device=dev)
for layout in ("bdl", "dbl"):
    mk = (lambda: torch.randn(d, b, l, device=dev).permute(1, 0, 2)) if layout == "dbl" else (lambda: torch.randn(b, d, l, device=dev))
    u, g = mk(), mk()
    for name, fn in (("fwd", lambda: cc.causal_conv1d_fwd(u, w, cb, True)), ("bwd", lambda: cc.causal_conv1d_bwd(u, w, cb, g, None, True))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        byt = (2 if name == "fwd" else 3) * 4 * b * d * l
        print(f"{layout} {name}: {ms*1e3:7.1f} us  {byt/ms/1e6:7.1f} GB/s  ({byt/ms/1e6/8000:.3f} of HBM)")
