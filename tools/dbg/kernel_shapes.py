"""Per-launch-shape durations of the kernels of ONE replayed step, from a rocprofv3 kernel trace.
usage: python tools/dbg/kernel_shapes.py <trace dir> <launches per step> [name substring ...]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))[-int(sys.argv[2]):]
pats = sys.argv[3:]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(void )?([A-Za-z0-9_:]+)", n)
    return m.group(2) if m else n[:40]
agg = collections.defaultdict(list)
for r in rows:
    n = short(r["Kernel_Name"])
    if not pats or any(p in n for p in pats):
        agg[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0]:34s} grid {k[1]:>8s} {k[2]:>5s} {k[3]:>4s} wg {k[4]:>4s}  x{len(v):3d}  avg {sum(v)/len(v):7.1f} us  total {sum(v):8.1f}")
