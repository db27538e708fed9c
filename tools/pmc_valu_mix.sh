#!/bin/bash
# Instruction mix and stall counters (two passes) of kernels whose name contains <match>, averaged per (kernel, grid).
# usage: tools/pmc_valu_mix.sh <tag> <match> <script.py> [args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; match=$2; shift 2
out=gpurun_out/pmc_$tag
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $out/a -- python3 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SMEM --output-format csv -d $out/b -- python3 "$@" > /dev/null 2>&1
python3 - $out "$match" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            k = (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:34], r.get("Grid_Size", ""))
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print(k, " ".join("%s=%.4g" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
