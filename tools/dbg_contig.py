import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mm_unet_amd.morph_sample as ms
from mm_unet_amd.mmunet import MMConv
orig_b = ms.MorphSampleFn.backward
def fwd_hook(ctx, input, y):
    print("fwd input", tuple(input.shape), input.stride(), input.is_contiguous(), "y", y.stride(), y.is_contiguous())
def bwd(ctx, dout):
    print("bwd dout", tuple(dout.shape), dout.stride(), dout.is_contiguous(), dout.dtype)
    return orig_b(ctx, dout)
ms.MorphSampleFn.backward = staticmethod(bwd)
m = MMConv(64, 64, kernel_size=3).cuda()
x = torch.randn(8, 64, 128, 128, device="cuda", requires_grad=True)
from torch.profiler import profile, ProfilerActivity
out = m(x); out.sum().backward(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    out = m(x); out.sum().backward(); torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total)[:14]:
    print(f"{e.self_device_time_total/1e3:8.3f} ms n={e.count:3d} {e.key[:80]}")
print("---- copies")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    out = m(x); out.sum().backward(); torch.cuda.synchronize()
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::copy_", "aten::contiguous", "aten::clone") and e.self_device_time_total > 0:
        print(f"{e.self_device_time_total/1e3:8.3f} ms n={e.count:3d} {e.key} {e.input_shapes}")
