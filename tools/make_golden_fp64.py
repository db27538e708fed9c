"""Float64 "truth" for the train-mode gradient test (tests/golden/mmnet_128_train_fp64.npz).

The fixture mmnet_128_train holds the REFERENCE's float32 logits / loss / |grad| sums on 4 x 3 x 128 x 128 (made by
tools/make_golden_modules.py from the imported reference).  Both the reference and this build compute in float32 and
differ from the exact result by rounding that the network amplifies (train-mode BatchNorm, the sampler's piecewise
constant d(row)); comparing the two float32 results with each other cannot tell which side is off.  This script runs the
CPU oracle (oracle/model_ref.py + the all-double build of oracle/mmu_oracle.c) on the same inputs and weights

  1. in float32 -- printed next to the reference's numbers: the oracle reproduces the reference (pinning), and
  2. in float64 -- stored: logits, loss and the |grad| sum of every live parameter,

so that the GPU test can bound |ours - fp64| by a multiple of |reference_fp32 - fp64| tensor by tensor.
Runs in the build container (CPU, a few minutes); nothing of the reference is imported here -- its numbers come from
the committed fixture."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import model_ref  # noqa: E402
from mm_unet_amd.mmunet import MM_Net  # noqa: E402  (constructor only: the seeded initial weights)


def run(dtype, xb, tb):
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(50)
        with torch.random.fork_rng():
            torch.set_default_dtype(torch.float32)      # the weights are DRAWN in float32 (as the reference draws them) ...
            torch.manual_seed(50)
            sd0 = {k: v.detach().clone() for k, v in MM_Net(num_classes=1).state_dict().items()}
            torch.set_default_dtype(dtype)
        sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}   # ... and then widened
        for k, v in sd.items():
            if v.is_floating_point() and "running_" not in k:
                v.requires_grad_()
        logits = model_ref.mm_net(sd, xb.to(dtype), training=True)
        loss = model_ref.dice_bce_loss(logits, tb.to(dtype))
        loss.backward()
        grads = {k: v.grad.detach() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
        return logits.detach(), float(loss.detach()), grads
    finally:
        torch.set_default_dtype(torch.float32)


def main():
    oracle.build()
    # default: the 4 x 3 x 128 x 128 fixture; "mmnet_64": the train-mode half of the 2 x 3 x 64 x 64 fixture (round 4:
    # its distributional check against the float32 reference alone moved with every change of a rounding -- the stem on
    # the build's own kernel took its 90th percentile from 31 % to 39.5 % -- so it, too, is judged against the truth)
    which = sys.argv[1] if len(sys.argv) > 1 else "mmnet_128_train"
    g0 = np.load(os.path.join(ROOT, "tests", "golden", which + ".npz"), allow_pickle=False)
    if which == "mmnet_64":
        g = {"xb": g0["xb"], "tb": g0["tb"], "gabs_names": g0["gabs_names"], "gabs": g0["train_gabs"],
             "loss": g0["train_loss"], "logits": g0["train_logits"]}
    else:
        g = g0
    xb, tb = torch.from_numpy(g["xb"]), torch.from_numpy(g["tb"])
    names = [str(s) for s in g["gabs_names"]]
    t0 = time.time()
    l32, loss32, g32 = run(torch.float32, xb, tb)
    print(f"oracle float32: {time.time() - t0:.0f} s; loss {loss32:.7f} (reference {float(g['loss']):.7f}); "
          f"logits max |oracle - reference| {float((l32 - torch.from_numpy(g['logits'])).abs().max()):.2e}")
    live = {k for k, v in g32.items() if float(v.abs().sum()) != 0.0 or k in names}
    assert set(names) <= set(g32), sorted(set(names) - set(g32))[:5]
    a32 = np.array([float(g32[k].double().abs().sum()) for k in names])
    ref = np.asarray(g["gabs"], dtype=np.float64)
    d = np.abs(a32 - ref) / np.maximum(ref, 1e-30)
    print(f"  |grad| sums, oracle float32 vs reference float32: median {np.median(d):.2e}, p90 {np.quantile(d, 0.9):.2e}, max {d.max():.2e}")
    t0 = time.time()
    l64, loss64, g64 = run(torch.float64, xb, tb)
    a64 = np.array([float(g64[k].abs().sum()) for k in names])
    dr = np.abs(ref - a64) / np.maximum(a64, 1e-30)
    do = np.abs(a32 - a64) / np.maximum(a64, 1e-30)
    print(f"oracle float64: {time.time() - t0:.0f} s; loss {loss64:.12f}; logits max |reference32 - fp64| "
          f"{float((torch.from_numpy(g['logits']).double() - l64).abs().max()):.2e}")
    print(f"  |grad| sums vs float64: reference float32 median {np.median(dr):.2e} p90 {np.quantile(dr, 0.9):.2e} max {dr.max():.2e}; "
          f"oracle float32 median {np.median(do):.2e} p90 {np.quantile(do, 0.9):.2e} max {do.max():.2e}")
    # the float32 oracle again on an input perturbed by 1.5e-5 relative = 2^-16, the size of the error of a two-part bf16
    # product (conv3x3_mfma / conv_s2_mfma / the 512-token GEMM): how far a float32 implementation moves under the
    # perturbation those kernels ARE -- the yardstick for the build's distance from the truth in the chaotic 64 x 64 case
    gen = torch.Generator().manual_seed(11)
    _, _, g32n = run(torch.float32, xb * (1.0 + 1.5e-5 * torch.randn(xb.shape, generator=gen)), tb)
    a32n = np.array([float(g32n[k].double().abs().sum()) for k in names])
    dn = np.abs(a32n - a64) / np.maximum(a64, 1e-30)
    print(f"  oracle float32 with 2^-16 input noise vs float64: median {np.median(dn):.2e} p90 {np.quantile(dn, 0.9):.2e}")
    out = os.path.join(ROOT, "tests", "golden", ("mmnet_64_train" if which == "mmnet_64" else which) + "_fp64.npz")
    np.savez_compressed(out, gabs_names=np.array(names), gabs64=a64, logits64=l64.numpy(), loss64=np.array(loss64),
                        gabs_oracle32=a32, gabs_oracle32_noise16=a32n)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
