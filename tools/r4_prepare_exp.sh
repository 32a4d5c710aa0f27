#!/bin/bash
# Round-4 preparation, on the CPU (hipcc cross-compiles): builds the experiment libraries of branch exp-r4 from the patch that main
# carries (bwbble_amd/tools_exp/r4_late_side_experiment.patch) WITHOUT touching bwbble_amd/csrc - the product's kernel sources and
# their hash (bench.py: source_hash, profiles/r3_c3_pmc.json) stay as they are.  Output: bwbble_amd/tools_exp/libbwbble_hip_<name>.so
# for tools/ab_bench.sh ... <name>:bwbble_amd/tools_exp/libbwbble_hip_<name>.so   (run from the repo root, then tools/r4_s1.sh on the GPU)
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $T/bwbble_amd $T/include
cp -r $R/bwbble_amd/csrc $T/bwbble_amd/ && cp $R/include/*.h $T/include/
grep -v '^#' $R/bwbble_amd/tools_exp/r4_late_side_experiment.patch | patch -s -p1 -d $T
for v in "laterec:-DBWB_LATE_REC" "lateside:-DBWB_LATE_SIDE" "priv:-DBWB_PRIV_FIRST" "fnext:-DBWB_FNEXT" "g2:-DBWB_GATHER2" "combo:-DBWB_GATHER2 -DBWB_PRIV_FIRST -DBWB_LATE_REC" "karg:-DBWB_KARG" "kg:-DBWB_KARG -DBWB_GATHER2" "kgall:-DBWB_KARG -DBWB_GATHER2 -DBWB_LATE_REC -DBWB_PRIV_FIRST"; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-value $f -o $R/bwbble_amd/tools_exp/libbwbble_hip_$n.so $T/bwbble_amd/csrc/bwb_hip.hip 2>&1 | grep -v "MD5\|\.file\|\^" || true
  echo "built bwbble_amd/tools_exp/libbwbble_hip_$n.so ($f)"
done
rm -rf $T
