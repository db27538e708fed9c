"""Per (kernel, grid) medians from a rocprofv3 kernel_trace.csv: python tools/kshape.py <dir> <name-substring>"""
import csv, collections, glob, statistics as st, sys
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            d[(r["Kernel_Name"][:60], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, "n", len(v), "med %.1f min %.1f" % (st.median(v), min(v)))
