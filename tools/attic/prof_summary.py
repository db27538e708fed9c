"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: per-step ms by kernel.
usage: python tools/prof_summary.py <dir> <n_steps_profiled> [top]"""
import glob, re, sys
import pandas as pd
d, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
df = pd.read_csv(f)
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'at::native::', '', n)
    if n.startswith('Cijk'):
        m = re.search(r'MT\d+x\d+x\d+', n)
        return 'GEMM ' + n[:22] + ' ' + (m.group(0) if m else '')
    return n[:100]
df['s'] = df.Name.map(short)
print(f'total GPU ms per step: {df.TotalDurationNs.sum()/1e6/steps:.1f}   launches per step: {df.Calls.sum()/steps:.0f}')
for _, r in df.head(top).iterrows():
    print(f"{r.TotalDurationNs/1e6/steps:8.2f} ms/step  calls/step {r.Calls/steps:6.0f}  avg {r.AverageNs/1e3:9.1f} us  {r.s}")
print('GEMM total ms/step', df[df.s.str.startswith('GEMM')].TotalDurationNs.sum()/1e6/steps)
print('elementwise/copy/fill total ms/step', df[df.s.str.contains('elementwise|copy|Fill|transpose')].TotalDurationNs.sum()/1e6/steps)
for name, pat in (("scan", "chunk_|scan_fwd_stream|reduce_partials|reduce_slices|sum_splits_kernel"), ("fused normalisation (nf_*)", "nf_"),
                  ("library normalisation", "BatchNorm|Rowwise|GroupNorm|ComputeInternal|batch_norm"),
                  ("library convolution", "igemm|miopen|Sp3Asm|naive_conv|ck::|_ZN2ck")):
    print(name, 'total ms/step', df[df.s.str.contains(pat)].TotalDurationNs.sum() / 1e6 / steps)
