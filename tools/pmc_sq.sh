#!/bin/bash
# Developer tool: SQ instruction-mix counters of one chr21-scale -n 3 batch (separate rocprofv3 --pmc passes).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 500 > /dev/null 2>&1   # leaves genome/index/reads in /tmp/bwb_bench
BWB_DEBUG=1 python3 $R/tools/prof_bench.py 48000000 1000000 3 2>&1 | grep -E "iterations|k_search"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq_$i -o run -- python3 $R/tools/prof_bench.py 48000000 1000000 3 > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/pmc_sq_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "kl_search" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(f"{k:28s} {v:.4g}")
PY
