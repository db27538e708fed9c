"""Timing-by-ablation of chunk_apply_bwd_p4 (results wrong -- timing only).  build here, run on the GPU box."""
import csv, glob, os, subprocess, sys, statistics as st
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate_scan as ab
K = "void chunk_apply_bwd_p4_kernel(ScanArgs p) {"
def ink(s, old, new, count=1):
    i = s.index(K)
    assert old in s[i:], old
    return s[:i] + s[i:].replace(old, new, count)
ab.VARIANTS = {
    "base": lambda s: s,
    "nobarrier": lambda s: ink(s, "        MMU_LDS_BARRIER();\n        const int arr", "        const int arr"),
    "noreverse": lambda s: ink(ink(s, "float Q0 = wave_reverse(P.x), R0 = wave_reverse(R.x), Q1 = wave_reverse(P.y), R1 = wave_reverse(R.y);", "float Q0 = P.x, R0 = R.x, Q1 = P.y, R1 = R.y;"),
                               "gam.x = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((62 - lane) << 2, __builtin_bit_cast(int, R0)));\n            gam.y = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((62 - lane) << 2, __builtin_bit_cast(int, R1)));", "gam.x = R0; gam.y = R1;"),
    "noscans": lambda s: ink(s, "            wave_scan_affine_x2(P0, S0, P1, S1);\n            wave_scan_affine_x2(Q0, R0, Q1, R1);\n            v2f h = v2f{wave_shift_up1(S0, h0.x)", "            v2f h = v2f{wave_shift_up1(S0, h0.x)"),
    "nowavesum": lambda s: ink(s, "            dAq[2 * pi] = wave_sum(dAp.x);\n            dAq[2 * pi + 1] = wave_sum(dAp.y);", "            dAq[2 * pi] = dAp.x;\n            dAq[2 * pi + 1] = dAp.y;"),
    "nostores": lambda s: ink(s, "        float ov[4];\n        if (w == 0) {", "        float ov[4];\n        if (p.softplus > 5) {"),
}
def run():
    for name in ab.VARIANTS:
        d = os.path.join(ab.ROOT, "gpurun_out", "ablb", name)
        env = dict(os.environ, MMUNET_HIP_LIB=os.path.join(ab.OUT, f"libabl_{name}.so"), TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                        os.path.join(ab.ROOT, "tools", "prof_scan_bwd.py"), "6"], env=env, cwd="/tmp",
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        v = []
        for f in glob.glob(d + "/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                if "bwd_p4" in r["Kernel_Name"]:
                    v.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        print(name, "p4 med %.1f min %.1f" % (st.median(v), min(v)), flush=True)
if __name__ == "__main__":
    {"build": ab.build, "run": run}[sys.argv[1]]()
