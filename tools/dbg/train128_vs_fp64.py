"""The build's train-mode |grad| sums on the mmnet_128_train inputs against the float64 truth (fixture mmnet_128_train_fp64),
next to the float32 reference's own deviation from it.  Writes gpurun_out/train128_ours.npz; prints the distributions."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mm_unet_amd.mmunet import MM_Net
from mm_unet_amd.loss import DICE_BCE_Loss
g = np.load(os.path.join(ROOT, "tests/golden/mmnet_128_train.npz"))
t = np.load(os.path.join(ROOT, "tests/golden/mmnet_128_train_fp64.npz"))
names = [str(s) for s in g["gabs_names"]]
runs = []
for rep in range(2):
    torch.manual_seed(50)
    m = MM_Net(num_classes=1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout2d):
            mod.p = 0.0
    m = m.to("cuda:0").train()
    lt = m(torch.from_numpy(g["xb"]).cuda())
    loss = DICE_BCE_Loss()(lt, torch.from_numpy(g["tb"]).cuda())
    loss.backward()
    p = dict(m.named_parameters())
    runs.append((lt.detach().cpu().double().numpy(), float(loss), np.array([float(p[n].grad.double().abs().sum()) for n in names])))
a64, ref = t["gabs64"], np.asarray(g["gabs"], dtype=np.float64)
ours = runs[0][2]
fl = 2e-4
dev = lambda a: np.abs(a - a64) / (np.abs(a64) + fl)
do, dr, dorc = dev(ours), dev(ref), dev(t["gabs_oracle32"])
print("loss ours %.7f ref32 %.7f fp64 %.9f" % (runs[0][1], float(g["loss"]), float(t["loss64"])))
print("logits max |ours - fp64| %.2e   |ref32 - fp64| %.2e   run-to-run %.2e" % (np.abs(runs[0][0] - t["logits64"]).max(), np.abs(g["logits"].astype(np.float64) - t["logits64"]).max(), np.abs(runs[0][0] - runs[1][0]).max()))
for nm, d in (("ours", do), ("ref32", dr), ("oracle32", dorc), ("ours run-to-run", np.abs(runs[1][2] - ours) / (np.abs(a64) + fl))):
    print("%-16s median %.2e p90 %.2e p99 %.2e max %.2e" % (nm, np.median(d), np.quantile(d, .9), np.quantile(d, .99), d.max()))
s5, s6 = np.asarray(g["gabs_sens5"]), np.asarray(g["gabs_sens6"])
for k in (2, 3, 4):
    for nm, sens in (("max(ref, sens5, sens6)", np.maximum(dr, np.maximum(s5, s6))), ("ref only", dr)):
        band = k * np.maximum(sens, np.median(dr))
        bad = np.nonzero(do > band)[0]
        print(f"k={k} band = k*max({nm}, median ref): {len(bad)} of {len(names)} outside; worst ratio {np.max(do / np.maximum(band, 1e-30)):.2f}", [(names[i], float(do[i]), float(band[i])) for i in bad[:4]])
np.savez(os.path.join(ROOT, "gpurun_out", "train128_ours.npz"), ours=ours, ours2=runs[1][2])
