"""Reads the s_memtime stamps of the diagnostic streaming-scan build (tools/stream_stamps.sh) and prints, per wave,
the cycles spent in each phase of a tile.  Run on the GPU box with MMUNET_HIP_LIB=tools/_abl/libmmunet_stamps.so."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm_unet_amd import _lib, selective_scan_hip as ss
DEV = "cuda:0"
b, d, l, n = 8, 128, 65536, 16
g = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
B = torch.randn(b, 1, n, l, device=DEV, generator=g); C = torch.randn(b, 1, n, l, device=DEV, generator=g)
D = torch.randn(d, device=DEV, generator=g); bias = 0.5 * torch.rand(d, device=DEV, generator=g)
mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).permute(1, 0, 2)
u, z = mk(), mk()
delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).permute(1, 0, 2)
for _ in range(3):
    ss.fwd(u, delta, A, B, C, D, z, bias, True, want_out=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 8 * 8 * 16))()
assert _lib.lib().mmu_debug_stream_stamps(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 8, 8, 16).astype(np.int64)
names = ["fetch", "bar1", "dl/du rd", "pair0", "pair1", "pair2", "pair3", "y wr", "bar2", "finalize", "prepare"]
for blk in range(2):
    print(f"block sel {blk}: cycles per phase, averaged over tiles 16..23 (s_memtime ticks)")
    for w in range(8):
        dt = np.diff(t[blk, w, :, :12], axis=1).mean(axis=0)
        tile = (t[blk, w, 1:, 0] - t[blk, w, :-1, 0]).mean()
        print(f"  wave {w}: " + " ".join(f"{nm}={v:.0f}" for nm, v in zip(names, dt)) + f" | tile={tile:.0f}")
