"""Times the four kernels of csrc/tri_fused.hip at the three RCG block sizes of the headline step (run under
tools/kstats.sh for per-kernel durations): python3 tools/prof_tri.py [reps]"""
import sys
import torch
sys.path.insert(0, ".")
from mm_unet_amd import tri_inner

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
B0 = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = "cuda:0"
for B, D, L, ns in ((B0, 128, 65536, 64), (B0, 128, 16384, 32), (B0, 128, 4096, 16)):
    xz = torch.randn(2 * D, B, L, device=dev).to(dt).permute(1, 0, 2)
    x, z = xz[:, :D], xz[:, D:]
    ws = [torch.randn(D, 4, device=dev) for _ in range(3)]
    bs = [torch.randn(D, device=dev) for _ in range(3)]
    dxz = torch.empty_like(xz)
    for _ in range(reps):
        convs = tri_inner.tri_conv_fwd(x, ns, ws, bs)
        out = tri_inner.tri_gate_fwd(z, ns, convs)
        dys = tri_inner.tri_gate_bwd(z, ns, convs, out, dxz[:, D:])
        tri_inner.tri_conv_bwd(x, ns, ws, bs, dys, dxz[:, :D])
    torch.cuda.synchronize()
print("ok")
