"""Kernel durations in launch order, abbreviated: python tools/seq_parse.py <dir> [skip-substring ...]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
        if any(s in n for s in sys.argv[2:]):
            continue
        rows.append((int(r["Start_Timestamp"]), n[:14], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
out = [(n, round(d, 1)) for _, n, d in rows]
for i in range(0, len(out), 10):
    print(out[i:i + 10])
