"""Phase times of csrc/mamba_small_fused.hip from its s_memtime stamps (tools/dbg/small_stamps.sh builds the library):
MMUNET_HIP_LIB=tools/_abl/libmmunet_smallstamps.so python tools/dbg/small_stamps.py [B K H W N parts]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_unet_amd import _lib
import bench_small_fused as bsf

args = [int(v) for v in sys.argv[1:]] or [8, 3, 32, 32, 16, 8]
B, K, H, W, N, parts = args
f, b = bsf.run(B, K, H, W, N, parts, iters=20)
print(f"fwd {f:.1f} us  bwd+reduce {b:.1f} us")
L = _lib.lib()
buf = (ctypes.c_ulonglong * (2 * 8 * 16))()
L.mmu_debug_small_stamps.argtypes = [ctypes.c_void_p]
assert L.mmu_debug_small_stamps(buf) == 0
nw = 2 * K
names = (["stage", "pre+coop", "scan", "epilogue"],
         ["stage", "pre+coop", "softplus", "scan", "gate+ddt", "conv+in_proj", "d offset", "wave sums", "final"])
for d, nm in enumerate(names):
    print("forward" if d == 0 else "backward", "(s_memtime cycles)")
    for w in range(nw):
        st = [buf[(d * 8 + w) * 16 + i] for i in range(len(nm) + 1)]
        print(f"  wave {w}: " + "  ".join(f"{n} {st[i + 1] - st[i]}" for i, n in enumerate(nm)) + f"   total {st[-1] - st[0]} cycles")
