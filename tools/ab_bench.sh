#!/bin/bash
# Developer tool: A/B of library builds on one workload in ONE GPU session (the genome, index and reads are built once and
# cached in /tmp/bwb_bench).  Replaces the one-shot tools/r2_probe*.sh of round 2: every profiles/r3_* A/B table names the
# command line of this script that produced it.
#
#   tools/ab_bench.sh <out-dir-name> "<bench.py args>" <variant> [<variant> ...]
#     variant = name[:lib[:ENV=val,ENV=val...]]   lib relative to the repo root ("" or "product" = bwbble_amd/libbwbble_hip.so)
#   e.g. tools/ab_bench.sh r3s1 "--steps 4 --warmup 1 --no-extras" product hist:bwbble_amd/tools_exp/libbwbble_hip_hist.so
#
# Prints one summary line per variant (and keeps the full JSON line and stderr under gpurun_out/<out-dir-name>/).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export GRAFT_REPO_ROOT=$R
[ -f "$R/bench.py" ] || { echo "ab_bench.sh: $R/bench.py not found"; exit 2; }
OUT=$R/gpurun_out/$1; shift
ARGS=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; [ "$rest" = "$v" ] && rest=""
  lib=${rest%%:*}; envs=${rest#*:}; [ "$envs" = "$rest" ] && envs=""
  [ "$lib" = "product" ] && lib=""
  (
    [ -n "$lib" ] && export BWB_LIB=$R/$lib
    for kv in ${envs//,/ }; do export "$kv"; done
    timeout ${AB_TIMEOUT:-1800} python3 $R/bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err"
  )
  python3 $R/tools/ab_show.py "$name" < "$OUT/$name.json" | tee -a "$OUT/summary.txt"
  grep -h "bwb hist" "$OUT/$name.err" | tail -1 > "$OUT/$name.hist" 2>/dev/null; [ -s "$OUT/$name.hist" ] || rm -f "$OUT/$name.hist"
  grep -h "stamps" "$OUT/$name.err" | tail -1 >> "$OUT/summary.txt"
done
