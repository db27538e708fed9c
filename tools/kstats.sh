#!/bin/bash
# rocprofv3 kernel trace of `python3 <script> [args]`: per-kernel calls / median / total, sorted by total (run on the GPU box)
# usage: tools/kstats.sh <tag> <script.py> [args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/ks_$tag
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$@" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, statistics as st, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print("%9.1f us total  n=%4d  med %8.1f  min %8.1f  %s" % (sum(v), len(v), st.median(v), min(v), k))
PY
