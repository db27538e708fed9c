#!/bin/bash
# rocprofv3 per-kernel median/min durations of the forward scan at the headline shape (run on the GPU box)
# usage: tools/time_scan_fwd.sh <tag> [launches] [lib.so]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tsf_${1:-x}
[ -n "$3" ] && export MMUNET_HIP_LIB="$3"
rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 ${PROF_SCRIPT:-tools/prof_scan_fwd.py} ${2:-20} > /dev/null 2>&1
python3 - "$out" "${1:-x}" <<'PY'
import csv, glob, sys, statistics as st, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "anonymous namespace" in r["Kernel_Name"] and "at::" not in r["Kernel_Name"]:
            d[r["Kernel_Name"].split("(anonymous namespace)::")[1][:28]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
line = []
for k, v in d.items():
    line.append("%s med %.1f min %.1f" % (k, st.median(v), min(v)))
    tot += st.median(v)
print("[%s] " % sys.argv[2] + " | ".join(line) + " | sum(med) %.1f us" % tot)
PY
