"""Busy/idle accounting of the HIP-graph replay steps from a rocprofv3 kernel trace of bench.py (graph mode):
union of kernel intervals vs wall span between optimizer bursts."""
import glob, sys
import pandas as pd
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
df = pd.read_csv(f).sort_values('Start_Timestamp').reset_index(drop=True)
adam = df.index[df.Kernel_Name.str.contains('FusedAdam')].tolist()
ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
print('step ends at rows', ends)
for a, b in zip(ends[-4:-1], ends[-3:]):
    sub = df.iloc[a + 1:b + 1]
    iv = sorted(zip(sub.Start_Timestamp, sub.End_Timestamp))
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = iv[-1][1] - iv[0][0]
    ksum = (sub.End_Timestamp - sub.Start_Timestamp).sum()
    print(f'kernels {len(sub)}  span {span/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms  idle {100*(1-busy/span):.1f}%  sum-of-kernels {ksum/1e6:.2f} ms')
