#!/bin/bash
# Developer tool: TLB / L1 / L2 counters of kl_search (small counter sets, one rocprofv3 --pmc pass each, every pass under its own
# timeout: a large TCP/TCC set once hung a box).  usage: pmc_mem.sh [genome_mb=48] [pool=4000000] [reads=1000000] [ndiff=3]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
[ -f "$R/bench.py" ] || { echo "pmc_mem.sh: $R/bench.py not found"; exit 2; }
MB=${1:-48}; POOL=${2:-4000000}; READS=${3:-1000000}; ND=${4:-3}
python3 $R/bench.py --genome-mb $MB --pool $POOL --reads $READS --ndiff $ND --steps 1 --warmup 0 --no-extras > /dev/null 2>&1   # leaves genome/index/reads in /tmp/bwb_bench
python3 $R/tools/prof_bench.py $MB $POOL $READS $ND 3
i=0
ONLY=${PMC_SETS:-1 2 3 4 5 6 7 8}   # PMC_SETS="5 6 7": only those counter groups (the SQ ones)
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS_LOAD"; do
  i=$((i+1))
  case " $ONLY " in *" $i "*) ;; *) continue;; esac
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_mem_$i -o run -- python3 $R/tools/prof_bench.py $MB $POOL $READS $ND 3 > /dev/null 2>&1 || echo "pass $i failed or timed out"
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
