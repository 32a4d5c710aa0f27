#!/bin/bash
# Developer tool: TLB / L1 / L2 counters of one chr21-scale -n 3 batch (small counter sets, one rocprofv3 --pmc pass each,
# every pass under its own timeout: a large TCP/TCC set once hung a box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 500 > /dev/null 2>&1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_mem_$i -o run -- python3 $R/tools/prof_bench.py 48000000 1000000 3 > /dev/null 2>&1 || echo "pass $i failed or timed out"
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R + "/gpurun_out/pmc_mem_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "kl_search" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()): print(f"{k:36s} {v:.4g}")
PY
