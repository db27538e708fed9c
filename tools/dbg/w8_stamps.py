"""Reads the s_memtime stamps of the diagnostic w8 scan-backward build (tools/dbg/w8_stamps.sh) and prints, per wave,
the ticks (10 ns) spent in each phase of a channel.  Run on the GPU box with
MMUNET_HIP_LIB=tools/_abl/libmmunet_w8stamps.so."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mm_unet_amd import _lib, selective_scan_hip as ss
DEV = "cuda:0"
b, d, l, n = 8, 128, 65536, 16
g = torch.Generator(device=DEV).manual_seed(0)
A = -0.5 * torch.rand(d, n, device=DEV, generator=g)
B = torch.randn(b, 1, n, l, device=DEV, generator=g); C = torch.randn(b, 1, n, l, device=DEV, generator=g)
D = torch.randn(d, device=DEV, generator=g); bias = 0.5 * torch.rand(d, device=DEV, generator=g)
mk = lambda: torch.randn(d, b, l, device=DEV, generator=g).permute(1, 0, 2)
u, z, dout = mk(), mk(), mk()
delta = (0.5 * torch.rand(d, b, l, device=DEV, generator=g)).permute(1, 0, 2)
res = ss.fwd(u, delta, A, B, C, D, z, bias, True)
for _ in range(3):
    ss.bwd(u, delta, A, B, C, D, z, bias, dout, res[1], None, None, True, False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 8 * 8 * 8))()
fn = _lib.lib().mmu_debug_w8_stamps
assert fn(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 8, 8, 8).astype(np.int64)
for blk in range(2):
    print(f"block sel {blk}: cycles (s_memtime) per phase, averaged over channels 16..23")
    for w in range(8):
        T = t[blk, w]
        fin = (T[1:, 0] - T[:-1, 5]).mean()          # barrier -> top of main: finishing the previous channel
        rd = (T[:, 7] - T[:, 0]).mean()              # slots + prepared rows read from LDS
        scan = (T[:, 6] - T[:, 7]).mean()            # exps, lane chains, the two row scans, h chain
        walk = (T[:, 1] - T[:, 6]).mean()            # adjoint walk (+ exchange stores)
        xw = (T[:, 2] - T[:, 1]).mean()
        prep = (T[:, 3] - T[:, 2]).mean()
        fetch = (T[:, 4] - T[:, 3]).mean()
        bar = (T[:, 5] - T[:, 4]).mean()
        per = (T[1:, 0] - T[:-1, 0]).mean()
        print(f"  wave {w}: finish={fin:.0f} lds rd={rd:.0f} recompute+scans={scan:.0f} walk={walk:.0f} xch tail={xw:.0f} prepare={prep:.0f} "
              f"fetch={fetch:.0f} barrier={bar:.0f} | channel={per:.0f}")
    print("  barrier arrival spread (max-min over waves): " + " ".join(str(int(t[blk, :, c, 4].max() - t[blk, :, c, 4].min())) for c in range(8)))
