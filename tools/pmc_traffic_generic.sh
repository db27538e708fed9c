#!/bin/bash
# HBM traffic per launch (FETCH_SIZE x2 for gfx950 streaming reads + WRITE_SIZE, separate passes, KiB units) of the
# kernels whose name contains <match>, per (kernel, grid).  usage: tools/pmc_traffic_generic.sh <tag> <match> <script.py> [args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; match=$2; shift 2
out=gpurun_out/traffic_$tag
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- python3 "$@" > /dev/null 2>&1
python3 - $out "$match" <<'PY'
import csv, glob, sys, collections
def per_launch(d, counter):
    tot = collections.defaultdict(list); dur = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == counter:
                k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:30], r.get("Grid_Size", ""), r.get("Dispatch_Id", "0"))
                tot[k[:2]].append(float(r["Counter_Value"]))
                dur[k[:2]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return tot, dur
fetch, dur = per_launch(sys.argv[1] + "/f", "FETCH_SIZE")
write, _ = per_launch(sys.argv[1] + "/w", "WRITE_SIZE")
for k in fetch:
    f = [v * 2048 for v in fetch[k]]; w = [v * 1024 for v in write.get(k, [0])]
    print(k, "launches", len(f), "fetch_x2_MB per launch", [round(v / 1e6, 1) for v in f], "write_MB", [round(v / 1e6, 1) for v in w],
          "dur_us", [round(v, 1) for v in dur[k]])
PY
