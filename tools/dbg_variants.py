import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ablate_scan as ab
K1 = "void chunk_reduce8_kernel(ScanArgs p) {"
def in_k1(s, old, new):
    i = s.index(K1)
    return s[:i] + ab.sub(s[i:], old, new)
ab.VARIANTS = {
    "v_splat": lambda s: in_k1(in_k1(s, "s = fma2(mul_bcast<0>(wv2[k], exp2_2(mul_bcast<0>(cum2[k], a2))), row[2 * k], s);", "s = fma2(exp2_2(a2 * cum[2 * k]) * wv[2 * k], row[2 * k], s);"),
                               "s = fma2(mul_bcast<1>(wv2[k], exp2_2(mul_bcast<1>(cum2[k], a2))), row[2 * k + 1], s);", "s = fma2(exp2_2(a2 * cum[2 * k + 1]) * wv[2 * k + 1], row[2 * k + 1], s);"),
    "v_vol": lambda s: s.replace('asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r)', 'asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(r)').replace('asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r)', 'asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=&v"(r)'),
    "v_nops": lambda s: in_k1(s, "const float s0 = BWD ? row_scan_add_up(s.x) : row_scan_add_down(s.x);", "asm volatile(\"s_nop 4\" ::: \"memory\"); const float s0 = BWD ? row_scan_add_up(s.x) : row_scan_add_down(s.x); asm volatile(\"s_nop 4\" ::: \"memory\");"),
}
if sys.argv[1] == "build":
    ab.build()
else:
    root = ab.ROOT
    for name in ["old"] + list(ab.VARIANTS):
        lib = os.path.join(root, "tools", "_abl", "libold.so" if name == "old" else f"libabl_{name}.so")
        subprocess.run([sys.executable, os.path.join(root, "tools", "dump_scan.py"), f"{root}/gpurun_out/dbg/{name}.npz", "1", "6", "256", "16"],
                       env=dict(os.environ, MMUNET_HIP_LIB=lib), check=True)
        if name != "old":
            print(name)
            subprocess.run(f"{sys.executable} {root}/tools/cmp_npz.py {root}/gpurun_out/dbg/old.npz {root}/gpurun_out/dbg/{name}.npz | grep -E 'gx_x|x_du'", shell=True)
