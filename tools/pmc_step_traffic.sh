#!/bin/bash
# HBM traffic of every kernel of one eager training step: FETCH_SIZE (x2, gfx950 streaming reads) and WRITE_SIZE in
# separate passes, summed per kernel name over the LAST step of tools/prof_step_eager.py.  usage (GPU box):
# tools/pmc_step_traffic.sh > gpurun_out/step_traffic.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/traffic_step
rm -rf $out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -- python3 tools/prof_step_eager.py 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -- python3 tools/prof_step_eager.py 1 > /dev/null 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
def load(d, counter):
    rows = []
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60],
                             float(r["Counter_Value"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "stem7_fwd_kernel" in r[1]]   # first kernel of a step's forward pass
    return rows[starts[-1]:]              # the last step
f = load(sys.argv[1] + "/f", "FETCH_SIZE"); w = load(sys.argv[1] + "/w", "WRITE_SIZE")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for _, k, v, d in f:
    a = agg[k]; a[0] += 1; a[1] += v * 2048; a[3] += d
for _, k, v, d in w:
    agg[k][2] += v * 1024
tot = [sum(a[i] for a in agg.values()) for i in (1, 2, 3)]
print("step total: fetch(x2) %.0f MB, write %.0f MB, kernel time under PMC %.1f ms" % (tot[0] / 1e6, tot[1] / 1e6, tot[2] / 1e3))
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:60]:
    print("%-60s calls %3d  fetch %8.1f MB  write %8.1f MB  time %8.1f us  -> %5.2f TB/s" %
          (k, a[0], a[1] / 1e6, a[2] / 1e6, a[3], (a[1] + a[2]) / a[3] / 1e6 if a[3] else 0))
PY
