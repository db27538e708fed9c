"""Is the forward bit-reproducible?  Runs it N times on fixed weights / inputs and reports the first module (in
execution order) whose output differs from run 0."""
import os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from test_modules_gpu import _mmnet
DEV = "cuda:0"
gen = torch.Generator().manual_seed(4)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(2, 3, size, size, generator=gen).to(DEV)
m = _mmnet().train()
names = {mod: n for n, mod in m.named_modules()}
log = []
def hook(mod, inp, out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    for i, o in enumerate(outs):
        if torch.is_tensor(o):
            log.append((names[mod] + (f"[{i}]" if len(outs) > 1 else ""), o.detach().clone()))
for mod in m.modules():
    mod.register_forward_hook(hook)
runs = []
with torch.no_grad():
    for r in range(8):
        log.clear()
        m(x); torch.cuda.synchronize()
        runs.append(list(log))
for r in range(1, len(runs)):
    first = None
    nd = 0
    for (n0, t0), (n1, t1) in zip(runs[r - 1], runs[r]):
        if not torch.equal(t0, t1):
            nd += 1
            if first is None:
                first = (n0, float((t0 - t1).abs().max()), float(t0.abs().max()))
    print(f"run {r} vs run {r - 1}: {nd} of {len(runs[0])} module outputs differ; first: {first}", flush=True)
