"""Steady-state per-step kernel breakdown from a rocprofv3 kernel_trace.csv of bench.py:
steps are delimited by the fused-AdamW kernel; the first `skip` steps (warm-up, MIOpen find) are dropped.
usage: python tools/prof_steady.py <dir> [skip_steps=1] [top=45] [detail_regex]
(detail_regex: those kernels again, split by launch grid -- one line per distinct shape)"""
import glob, re, sys
import pandas as pd
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
adam = df.index[df.Kernel_Name.str.contains('FusedAdam|adamw_multi_kernel')].tolist()
# the optimizer step launches several multi-tensor kernels back to back; a step ends at the last one of a burst
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
print('optimizer bursts (step ends) at rows', ends, 'of', len(df))
start = ends[skip - 1] + 1 if skip > 0 else 0
stop = ends[-1] + 1
n_steps = len(ends) - skip
sub = df.iloc[start:stop].copy()
sub['dur'] = sub.End_Timestamp - sub.Start_Timestamp
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'at::native::', '', n)
    if n.startswith('Cijk'):
        m = re.search(r'MT\d+x\d+x\d+', n)
        return 'GEMM ' + n[:22] + ' ' + (m.group(0) if m else '')
    return n[:110]
sub['s'] = sub.Kernel_Name.map(short)
wall = (sub.End_Timestamp.max() - sub.Start_Timestamp.min()) / 1e6 / n_steps
g = sub.groupby('s').dur.agg(['sum', 'count']).sort_values('sum', ascending=False)
print(f'steady steps: {n_steps}; GPU-busy ms/step {sub.dur.sum()/1e6/n_steps:.1f}; span ms/step {wall:.1f}; launches/step {len(sub)/n_steps:.0f}')
for name, r in g.head(top).iterrows():
    print(f"{r['sum']/1e6/n_steps:8.2f} ms/step  calls/step {r['count']/n_steps:6.0f}  avg {r['sum']/r['count']/1e3:9.1f} us  {name}")
cat = lambda pat: sub[sub.s.str.contains(pat)].dur.sum() / 1e6 / n_steps
# (own kernels by family; library convolutions = MIOpen / ck / igemm only: the own conv kernels have their own buckets)
print(f"GEMM(rocBLAS) {cat('^GEMM'):.1f} | conv(MIOpen/ck/igemm) {cat('igemm|miopen|Sp3Asm|naive_conv|ck::|_ZN2ck'):.1f} | "
      f"ATen elementwise/copy/fill/reduce {cat('at::native|elementwise|Fill|SubTensor|batched_transpose'):.1f} | "
      f"scan {cat('chunk_|scan_fwd_stream|reduce_partials|reduce_slices|sum_splits_kernel'):.1f} | conv1d {cat('conv1d_'):.1f} | "
      f"morph {cat('morph_|zigzag_|coords_'):.1f} | norm(own nf_*) {cat('nf_'):.1f} | norm(library) "
      f"{cat('BatchNorm|Rowwise|GroupNorm|ComputeInternal|batch_norm'):.1f} | small conv3x3 {cat('conv3x3s_'):.1f} | "
      f"matrix-core conv/GEMM {cat('conv3x3_mfma|conv3x3_wgrad_mfma|gemm_tokens'):.1f} | small Mamba pre/post {cat('mamba_pre|mamba_post'):.1f}")

if len(sys.argv) > 4:
    gx = [c for c in sub.columns if c.lower() in ('grid_size_x', 'grid_size_y', 'grid_size_z', 'grid_size', 'workgroup_size_x', 'lds_block_size')]
    det = sub[sub.s.str.contains(sys.argv[4])]
    print('\nby launch grid:', gx)
    gd = det.groupby(['s'] + gx).dur.agg(['sum', 'count']).sort_values('sum', ascending=False)
    for key, r in gd.iterrows():
        print(f"{r['sum']/1e6/n_steps:8.3f} ms/step  calls/step {r['count']/n_steps:5.1f}  avg {r['sum']/r['count']/1e3:9.1f} us  {key[0][:60]}  {key[1:]}")

# the launch sequence of ONE steady step (name, grid, duration), for reading who follows whom
seq_path = d.rstrip('/') + '_step_sequence.txt'
lo_, hi_ = ends[-2] + 1, ends[-1] + 1
with open(seq_path, 'w') as fh:
    t0 = df.Start_Timestamp[lo_]
    for i in range(lo_, hi_):
        r = df.iloc[i]
        fh.write(f"{(r.Start_Timestamp - t0)/1e3:10.1f} us  {(r.End_Timestamp - r.Start_Timestamp)/1e3:8.1f} us  "
                 f"grid ({r.Grid_Size_X},{r.Grid_Size_Y},{r.Grid_Size_Z}) wg {r.Workgroup_Size_X}  {short(r.Kernel_Name)}\n")
