"""Every gemm_tokens / gemm_nt call of one MM_Net training pass (4 x 3 x 128 x 128 by default: the train-mode fixtures'
size) checked on the spot against the float64 product of its strided operands: prints the worst relative error per shape."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mm_unet_amd import mfma_gemm
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss

B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
worst = collections.defaultdict(float)
orig = mfma_gemm.gemm_tokens

def audited(weight, x, out, rows, inner, tokens, batch, x_rs, x_bs, out_rs, out_bs, transposed_weight=False, prepared=None,
            accumulate=False):
    before = out.clone() if accumulate else None
    r = orig(weight, x, out, rows, inner, tokens, batch, x_rs, x_bs, out_rs, out_bs, transposed_weight, prepared, accumulate)
    W = (weight.t() if transposed_weight else weight).double()
    so = lambda t, rs, bs, n: torch.as_strided(t, (batch, n, tokens), (bs, rs, 1), t.storage_offset())
    X = so(x, x_rs, x_bs, inner).double()
    ref = torch.matmul(W.unsqueeze(0), X)
    got = so(out, out_rs, out_bs, rows).double()
    if accumulate:
        ref = ref + so(before, out_rs, out_bs, rows).double()
    err = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    key = (rows, inner, tokens, batch, bool(transposed_weight), bool(accumulate))
    worst[key] = max(worst[key], err)
    return r

mfma_gemm.gemm_tokens = audited
orig_nt = mfma_gemm.gemm_nt
worst_nt = collections.defaultdict(float)

def audited_nt(a, b, m, n, batch, seqlen, a_rs, a_bs, b_rs, b_bs, exact=False, narrow=False):
    from mm_unet_amd import deferred
    with deferred.paused():
        c = orig_nt(a, b, m, n, batch, seqlen, a_rs, a_bs, b_rs, b_bs, exact, narrow)
    so = lambda t, rs, bs, r: torch.as_strided(t, (batch, r, seqlen), (bs, rs, 1), t.storage_offset()).double()
    ref = torch.einsum("bit,bjt->ij", so(a, a_rs, a_bs, m), so(b, b_rs, b_bs, n))
    worst_nt[(m, n, batch, seqlen)] = max(worst_nt[(m, n, batch, seqlen)], float((c.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)))
    return c

mfma_gemm.gemm_nt = audited_nt
import mm_unet_amd.tall_gemm as tg, mm_unet_amd.selective_scan_interface as ssi
torch.manual_seed(50)
m = MM_Net(num_classes=1).cuda().train()
x = torch.randn(B, 3, S, S, device="cuda"); t = (torch.rand(B, 1, S, S, device="cuda") > 0.88).float()
DICE_BCE_Loss()(m(x), t).backward()
print("gemm_nt (m, n, batch, seqlen): worst relative error")
for k, e in sorted(worst_nt.items(), key=lambda kv: -kv[1])[:25]:
    print("   %.3e  %s" % (e, k))
print("gemm_tokens")
for k, e in sorted(worst.items(), key=lambda kv: -kv[1])[:12]:
    print("%.3e  rows %4d inner %4d tokens %6d batch %d trans %d acc %d" % ((e,) + tuple(int(v) for v in k)))
