#!/bin/bash
# PMC counters of the forward-scan kernels (two passes), averaged per launch.  usage: pmc_scan_fwd.sh <tag> [lib.so]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
[ -n "$2" ] && export MMUNET_HIP_LIB="$2"
out=gpurun_out/pmc_$1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $out/a -- python3 ${PROF_SCRIPT:-tools/prof_scan_fwd.py} 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $out/b -- python3 ${PROF_SCRIPT:-tools/prof_scan_fwd.py} 3 > /dev/null 2>&1
python3 - $out $1 <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "anonymous namespace" in r["Kernel_Name"] and "at::" not in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(anonymous namespace)::")[1][:24]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    print("[%s] %s" % (sys.argv[2], k), " ".join("%s=%.4g" % (c.replace("SQ_", ""), sum(x) / len(x)) for c, x in sorted(v.items())))
PY
