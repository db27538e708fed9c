#!/bin/bash
# Stall / instruction-mix counters of EVERY kernel of an eager training step (two PMC passes), summed per kernel name and
# sorted by time: which kernels are parked at s_waitcnt (WAIT_ANY), which stall at issue (WAIT_INST_ANY), which are busy.
# usage (GPU box): tools/pmc_step_mix.sh <tag>   -> gpurun_out/r3/pmc_step_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-step}
out=gpurun_out/pmc_step_$tag
args="bench.py --no-graph --steps 2 --warmup 2 --no-cpu-baseline --no-roofline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/a -- python3 $args > /dev/null 2>&1
python3 - $out <<'PY' > gpurun_out/r3/pmc_step_$tag.txt
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for f in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], n)
        if key not in seen:
            seen.add(key)
            agg[n]["_dur_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            cnt[n] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["_dur_us"])
print("all launches of the profiled process (4 eager steps incl. warm-up); per kernel: total us, launches, share of wave cycles parked at s_waitcnt / stalled at issue / issuing, VALU instructions per wave")
for n, v in rows[:70]:
    wc = max(v["SQ_WAVE_CYCLES"], 1.0)
    print(f"{v['_dur_us']:10.0f} us {cnt[n]:6d}  wait_any {v['SQ_WAIT_ANY']/wc:5.2f}  wait_inst {v['SQ_WAIT_INST_ANY']/wc:5.2f}  active {v['SQ_ACTIVE_INST_ANY']/wc:5.2f}  valu/wave {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):8.0f}  salu/wave {v['SQ_INSTS_SALU']/max(v['SQ_WAVES'],1):7.0f}  {n}")
PY
head -45 gpurun_out/r3/pmc_step_$tag.txt
