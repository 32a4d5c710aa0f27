#!/usr/bin/env python3
"""Developer tool: a few steps of the streamed pipeline on the files bench.py left in its work directory (for rocprofv3 --pmc passes:
no child processes, nothing but the library's kernels).  usage: prof_bench.py <genome_mb> <pool> <reads_per_step> <n_diff> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bwbble_amd as bw
work = os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench")
n_fwd, pool, B, nd = int(float(sys.argv[1]) * 1e6), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
fa, fq = f"{work}/genome_{n_fwd}.fa", f"{work}/reads_{n_fwd}_{pool}_100_i0.1_r0.fq"
ctx = bw.Context(bw.BwtFile(fa + ".bwt"))
seqs, lens = bw.load_fastq_codes(fq, max_reads=min(pool, B * bw.MAX_SLOTS))
p = bw.params(["-n", nd])
nb = max(1, min(bw.MAX_SLOTS, len(lens) // B))
for j in range(nb):
    ctx.slot_upload(j, p, seqs[j * B:(j + 1) * B], lens[j * B:(j + 1) * B])
ctx.flush()
ctx.reset_stats()
for s in range(steps):
    if s >= nb:
        ctx.slot_wait(s % nb)
    ctx.slot_submit(s % nb)
ctx.flush()
st = ctx.stats()
print("ms", st.ms_calc_d, st.ms_search, "launches", st.launches_search, "visits", st.visits_single + st.visits_alphabet, "lane iterations", st.lane_iterations,
      "wave iterations", st.wave_iterations)
