"""Timing-by-ablation of the forward scan kernels: builds variants of selective_scan.hip with one cost
removed each (results are then wrong -- timing only) into tools/_abl/, to be timed on the GPU with
    MMUNET_HIP_LIB=tools/_abl/libabl_<name>.so python tools/prof_scan_fwd.py
Usage: python tools/ablate_scan.py build        (here, cross-compiles)
       python tools/ablate_scan.py run          (on the GPU box: rocprofv3 per variant, prints K1/K3 averages)
"""
import csv, glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mm-unet_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_abl")

def sub(s, old, new, count=1):
    assert old in s, old
    return s.replace(old, new, count)

import os as _os
NP = (lambda s: s) if _os.environ.get("ABL_FULL") else (lambda s: sub(s, "for (int pr = 0; pr < NE / 2; ++pr)\n                fwd_pair8", "for (int pr = 0; pr < (int)(p.softplus > 5); ++pr)\n                fwd_pair8"))
K3 = "__global__ __launch_bounds__(1024) void chunk_apply_fwd_kernel(ScanArgs p) {"

def in_k3(s, old, new):
    i = s.index(K3)
    return s[:i] + sub(s[i:], old, new)

VARIANTS = {
    "base": lambda s: s,
    "nopairs": NP,
    "np_nosoft": lambda s: in_k3(in_k3(NP(s), "if (p.softplus) v = softplus_thr(v);", ""), "y[i] *= zv[i] * sigmoidf_(zv[i]);", "y[i] *= zv[i];"),
    "np_noAx": lambda s: in_k3(in_k3(NP(s), "sA[n] = n < N ? p.A[(long)d * p.A_ds + (long)n * p.A_ns] * MMU_LOG2E : 0.f;", "sA[n] = -0.5f;"),
                               "sH[n] = (cprev >= 0 && n < N) ? xprev[2 * n + 1] : 0.f;", "sH[n] = 0.f;"),
    "np_noz": lambda s: in_k3(NP(s), "        if (p.z)  // consumed after the state loop\n            load_k<io_t, KX, FULL>((const io_t *)p.z + (long)b * p.z_bs + (long)d * p.z_ds + t0 + tl, nvalid, p.vec_io,\n                                   zv);",
                              "        for (int i = 0; i < KX; ++i) zv[i] = dl[i];"),
    "np_nostore": lambda s: in_k3(NP(s), "            store_k<io_t, KX, FULL>((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds + t0 + tl, nvalid,\n                                    p.vec_io, y);",
                                  "            if (y[0] == 123.456f) store_k<io_t, KX, FULL>((io_t *)p.out_z + (long)b * p.out_z_bs + (long)d * p.out_z_ds + t0 + tl, nvalid,\n                                    p.vec_io, y);"),
    "np_nostage": lambda s: sub(NP(s), "    if constexpr (KX == 8) {\n        stage_pair8<io_t, FULL>(sB,", "    if constexpr (false) {\n        stage_pair8<io_t, FULL>(sB,"),
}

def build():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(SRC, "selective_scan.hip")).read()
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -munsafe-fp-atomics".split()
    others = [os.path.join(SRC, f) for f in ("mmu_abi.o", "causal_conv1d.o", "morph_sample.o", "morph_coords.o")]
    procs = []
    for name, fn in VARIANTS.items():
        f = os.path.join(SRC, f"_abl_{name}.hip")
        open(f, "w").write(fn(src))
        o = os.path.join(OUT, f"abl_{name}.o")
        procs.append((name, f, o, subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", f, "-o", o])))
    for name, f, o, pr in procs:
        assert pr.wait() == 0, name
        os.remove(f)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(OUT, f"libabl_{name}.so"), o, *others])
        os.remove(o)
        print("built", name)

def run():
    res = {}
    for name in VARIANTS:
        d = os.path.join(ROOT, "gpurun_out", "abl", name)
        env = dict(os.environ, MMUNET_HIP_LIB=os.path.join(OUT, f"libabl_{name}.so"), TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--",
                        sys.executable, os.path.join(ROOT, "tools", "prof_scan_fwd.py"), "4"], env=env, cwd="/tmp",
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        row = {}
        for f in glob.glob(d + "/*/*kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                for k in ("chunk_reduce8", "chunk_carry", "chunk_apply_fwd"):
                    if k in r["Name"]:
                        row[k] = float(r["AverageNs"]) / 1e3
        res[name] = row
        print(name, {k: round(v, 1) for k, v in row.items()}, flush=True)

if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
