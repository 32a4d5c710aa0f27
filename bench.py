#!/usr/bin/env python3
"""bench.py - headline benchmark: 100 bp reads aligned per second on a synthetic IUPAC multi-genome, MI355X.

Default workload = config C3 of SURVEY.md 8(d) / BASELINE.json configs[2]: GRCh37-scale synthetic multi-genome (3.1 G forward
characters -> 6.85 G BWT rows, 64-bit positions, 13.7 GB device index for the alignment kernels), 10 M x 100 bp reads per GPU, `align -n 3`.  Genome,
index (the product's own host indexer) and reads are built inside the run and cached under --workdir (about 4 minutes on the
GPU box's 256 cores the first time).

One "step" = one pass of the hot path (kl_calc_d + one slice of kl_search, include/bwbble_hip.h) over one batch of
`--reads` reads that is resident in HBM before the timed region starts: the read pool is cut into batches, each uploaded into
a slot of the context up front, and step s runs batch s mod n_batches.  Steps are queued back to back like the steps of any
GPU job; the timed region ends with a flush, so every read of every step is finished inside it.

Multi-GPU: one process per GPU (torch.distributed, RCCL), FM-index replicated, ONE logical FASTQ of N x pool reads cut into
contiguous shards (rank r owns reads [r*pool, (r+1)*pool), stored as its own file), no data-path collective; weak scaling.
`--gpus N` without a launcher starts the N ranks itself (a child `python -m torch.distributed.run`, before anything touches
the GPU).  See DESIGN.md "Measurement".

  python bench.py [--gpus N --steps K --warmup W] [--genome-mb 3100 --pool 10000000 --reads 2500000 --ndiff 3]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
ALG_BYTES_PER_VISIT = 192  # one reference checkpoint row (128 B) + one packed BWT block (64 B)
REC_BYTES = 16  # what kl_calc_d hands to kl_search: one 16-byte record per four read positions (bwb_kernels.h: rec_put)


def rec_bytes_written(read_len):
    """bytes of records kl_calc_d writes per read: the zero-initialised records, then their bytes (each once, some twice)"""
    return 2 * REC_BYTES * ((read_len >> 2) + 1)
DEV_BYTES_PER_BUCKET = 128  # what the device layout fetches per rank visit (bwb_device.h)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["C2", "C3", "C5"], default=os.environ.get("BWB_BENCH_CONFIG", "C3"),
                    help="SURVEY 8(d) configuration: C3 = GRCh37 scale, 100 bp, -n 3 (the default, and C4's per-GPU workload); C2 = chr21 scale; "
                         "C5 = GRCh37 scale, 150 bp reads with indels, -n 5 -o 1 -e 6 -l 32 -k 2.  The options below override its parts")
    ap.add_argument("--genome-mb", type=float, default=None,
                    help="forward characters of the synthetic genome, in millions (3100 = GRCh37 scale, configs C3-C5; 48 = chr21 scale, C2)")
    ap.add_argument("--pool", type=int, default=None, help="reads per GPU in the FASTQ shard (C3: 10 M; with --gpus N > 1: 12.5 M = config C4's 100 M over 8 GPUs)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU per step (one batch)")
    ap.add_argument("--ndiff", type=int, default=None, help="-n (the reference default is 0; reported next to it)")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("BWB_BENCH_CPU_SAMPLE", 0)), help="reads in the CPU baseline sample (0 = auto)")
    ap.add_argument("--workdir", default=os.environ.get("BWB_BENCH_DIR", "/tmp/bwb_bench"))
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline, also.n0, end_to_end and rank_micro (profiling runs)")
    a = ap.parse_args()
    cfg = {"C2": dict(genome_mb=48.0, pool=4000000, reads=1000000, ndiff=3, read_len=100, indel="0.1", extra=[]),
           "C3": dict(genome_mb=3100.0, pool=10000000, reads=2500000, ndiff=3, read_len=100, indel="0.1", extra=[]),
           "C5": dict(genome_mb=3100.0, pool=8000000, reads=2000000, ndiff=5, read_len=150, indel="0.2",
                      extra=["-o", "1", "-e", "6", "-l", "32", "-k", "2"])}[a.config]
    env = lambda k, d: type(d)(os.environ[k]) if os.environ.get(k) else d
    if a.genome_mb is None: a.genome_mb = env("BWB_BENCH_GENOME_MB", cfg["genome_mb"])
    if a.pool is None: a.pool = env("BWB_BENCH_POOL", 12500000 if (a.gpus > 1 and a.config == "C3") else cfg["pool"])
    if a.reads is None: a.reads = env("BWB_BENCH_READS", cfg["reads"])
    if a.ndiff is None: a.ndiff = env("BWB_BENCH_NDIFF", cfg["ndiff"])
    if a.read_len is None: a.read_len = cfg["read_len"]
    a.indel_pct, a.extra_flags = cfg["indel"], cfg["extra"]
    return a


def self_launch(a):
    """--gpus N without a launcher: start the N ranks as a child process.  This process has not touched the GPU (no HIP call, no
    torch.cuda query) and only waits for the child; it never exec()s."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def wait_for(path, what, timeout=3600):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError(f"timed out waiting for {what}")
        time.sleep(1.0)


T_PROCESS_START = time.time()


def over_budget(need_s):
    """the extras after the timed region are optional evidence: one that would take the run past BWB_BENCH_BUDGET_S (default 900 s from process start)
    is skipped and says so"""
    return time.time() - T_PROCESS_START + need_s > float(os.environ.get("BWB_BENCH_BUDGET_S", 900))


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    import numpy as np
    import bwbble_amd as bw

    # ---- workload: synthetic genome + index (built once, by rank 0, with the product's own indexer) ----------
    n_fwd = int(a.genome_mb * 1e6)
    os.makedirs(a.workdir, exist_ok=True)
    fa = os.path.join(a.workdir, f"genome_{n_fwd}.fa")
    ok = fa + ".bwt.ok"
    t_build = time.time()
    if rank == 0:
        bw.build()
        if not os.path.exists(ok):
            n_rec = 1 if n_fwd <= 60_000_000 else 24
            subprocess.run([bw.SYNTH_BIN, "genome", fa, str(n_fwd), str(n_rec), str(max(4, n_fwd // 2400)), "21"], check=True)
            import resource
            t_ix = time.time()
            subprocess.run([bw.HOST_BIN, "index", fa], check=True, stdout=subprocess.DEVNULL)  # f3: the product's own index builder (host/index.c)
            json.dump({"index_s": round(time.time() - t_ix, 1), "peak_rss_GB": round(resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6, 1),
                       "bwt_bytes": os.path.getsize(fa + ".bwt"), "cores": os.cpu_count()}, open(fa + ".index_stats.json", "w"))
            if os.path.exists(fa + ".ref"):
                os.remove(fa + ".ref")  # 2 bytes per forward character that nothing here reads
            open(ok, "w").write("ok\n")
    else:
        wait_for(ok, "rank 0 to build the index")
    a.pool = max(a.pool, a.reads)
    shard = lambda r: os.path.join(a.workdir, f"reads_{n_fwd}_{a.pool}_{a.read_len}_i{a.indel_pct}_r{r}.fq")
    fq = shard(rank)  # shard `rank` of the logical FASTQ
    if not os.path.exists(fq + ".ok"):
        subprocess.run([bw.SYNTH_BIN, "reads", fa, fq, str(a.pool), str(a.read_len), str(1000 + rank), "1.0", a.indel_pct, "0.0"], check=True)
        open(fq + ".ok", "w").write("ok\n")
    t_build = time.time() - t_build

    import torch  # plumbing only (device census, barrier, max-over-ranks)
    from bwbble_amd import dist as bdist
    ndev = torch.cuda.device_count()
    share = bool(os.environ.get("BWB_BENCH_SHARE_DEVICE"))  # test knob: several ranks on one GPU (gloo for the timing protocol)
    device = local_rank % max(ndev, 1) if share else local_rank
    if ndev < device + 1:
        sys.exit(f"bench.py: rank {rank} needs HIP device {device}, {ndev} visible (no CPU path)")
    grp = bdist.Group(backend="gloo" if share and world > 1 else None)  # one process per GPU; nccl (= RCCL) under torch.distributed.run
    barrier = grp.barrier
    for r in range(world):  # every shard exists before anybody needs a neighbour's
        wait_for(shard(r) + ".ok", f"rank {r}'s FASTQ shard")

    seqs, lens = bw.load_fastq_codes(fq)
    flags = ["-n", str(a.ndiff)] + a.extra_flags
    p = bw.params(flags)
    bwt = bw.BwtFile(fa + ".bwt")
    t0 = time.time()
    ctx = bw.Context(bwt, device=device)
    t_ctx = time.time() - t0
    B = a.reads
    nb = max(1, a.pool // B)  # distinct batches of the pool
    ns = bw.MAX_SLOTS  # batches resident in HBM (slots of the context): slot j holds batch j mod nb; step s runs slot s mod ns.
                              # A slot runs again only when its previous pass is complete - its heaviest read has ns - 1 further slices for
                              # that, which is why small batches on a small index want all the slots
    batch = lambda j: (seqs[(j % nb) * B:(j % nb + 1) * B], lens[(j % nb) * B:(j % nb + 1) * B])
    for j in range(ns):
        ctx.slot_upload(j, p, *batch(j))
    ctx.flush()

    def run_steps(k):
        """k steps queued back to back; returns when all of them are complete"""
        for s in range(k):
            slot = s % ns
            if s >= ns:
                ctx.slot_wait(slot)  # its previous pass must be complete before the batch runs again
            ctx.slot_submit(slot)
        ctx.flush()

    if a.warmup:
        run_steps(a.warmup)
    ctx.reset_stats()
    barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    t0 = time.perf_counter()
    run_steps(a.steps)  # ends with a flush: the library's streams are idle when it returns
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    dt_rank = time.perf_counter() - t0  # this rank's own time for its K steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    st = ctx.stats()
    kern_ms = st.ms_calc_d + st.ms_search
    visits = st.visits_single + st.visits_alphabet
    dt, kern_ms, visits_all = grp.reduce_step(dt, kern_ms, float(visits))  # MAX time over ranks, SUM of visits
    gathered = grp.all_gather_pairs(int(a.reads * a.steps / dt_rank), int(t_ctx * 1000))
    rank_rates = [v[0] for v in gathered]  # every rank's own reads/s (its own clock between the barriers)
    rank_ctx_ms = [v[1] for v in gathered]  # every rank's .bwt -> HBM time (the ranks map one file: its pages are shared)
    off0, alns0 = ctx.slot_result(0)  # hits of batch 0 = the first B reads of this shard

    # every rank re-aligns a sample of its right neighbour's shard: the bytes must not depend on which GPU did the work
    shard_check = None
    if world > 1:
        nbr = (rank + 1) % world
        n_chk = min(2000, B)  # (its own name: `ns` is the number of resident slots and goes into the line)
        nseqs, nlens = bw.load_fastq_codes(shard(nbr), max_reads=n_chk)
        noff, nalns = ctx.align(p, nseqs, nlens)
        mine = zlib.crc32(bw.aln_bytes(off0[:n_chk + 1], alns0[:int(off0[n_chk])]))
        theirs = zlib.crc32(bw.aln_bytes(noff, nalns))
        allc = grp.all_gather_pairs(mine, theirs)
        shard_check = all(allc[(r + 1) % world][0] == allc[r][1] for r in range(world))
        if not shard_check:
            sys.exit("bench.py: a shard's sample aligned on a neighbouring GPU gave different .aln bytes")

    if rank != 0:
        grp.close()
        return
    total_reads = B * world * a.steps
    value = total_reads / dt
    # Roofline per kernel.  The unit is the rank visit; its bytes in THIS layout are counted in the kernels: 128 B per bucket actually
    # fetched (an L-1/U pair in one bucket is fetched once) + the heap entries the search stored and loaded + the per-position
    # records it loaded - bytes the memory system cannot avoid moving, so the fraction cannot exceed 1.
    # `ref_layout_GBs` prices the same visits at the reference layout's 192 B: a rate for comparison, not a fraction.
    vis_calcd = st.visits_calc_d
    vis_search = visits - vis_calcd
    esz = 32 if (p.max_gapo > 1 or max(p.mm_score, p.gapo_score, p.gape_score) > 63) else 16
    def kernel(vis, ms, launches, bkt, extra_dev=0):
        sec = ms * 1e-3
        dev = bkt * DEV_BYTES_PER_BUCKET + extra_dev
        return {"launches": int(launches), "ms_per_launch": round(ms / max(launches, 1), 3), "ms_total": round(ms, 3),
                "visits_per_step": int(vis / a.steps), "bucket_bytes_per_step": int(bkt * DEV_BYTES_PER_BUCKET / a.steps),
                "device_bytes_per_step": int(dev / a.steps), "device_bytes_per_launch": int(dev / max(launches, 1)),
                "device_GBs": round(dev / sec / 1e9, 1) if ms else 0.0, "device_frac": round(dev / sec / 1e9 / HBM_PEAK_GBS, 4) if ms else 0.0,
                "Gvisits_per_s": round(vis / sec / 1e9, 2) if ms else 0.0, "ref_layout_GBs": round(vis * ALG_BYTES_PER_VISIT / sec / 1e9, 1) if ms else 0.0}
    heap_bytes = (st.heap_entries_stored + st.heap_entries_loaded) * esz + st.record_loads * REC_BYTES
    k_search = kernel(vis_search, st.ms_search, st.launches_search, st.bucket_loads_search, heap_bytes)
    k_calcd = kernel(vis_calcd, st.ms_calc_d, st.launches_calc_d, st.bucket_loads_calc_d, B * a.steps * rec_bytes_written(a.read_len))
    dom_name, dom = ("kl_search", k_search) if st.ms_search >= st.ms_calc_d else ("kl_calc_d", k_calcd)
    traffic, traffic_src = measured_traffic(a, B, dom_name, dom)
    index_mb = 2 * bwt.length / 1e6  # one 128-byte bucket per 64 BWT characters
    scale = {3_100_000_000: "C3 GRCh37-scale", 48_000_000: "C2 chr21-scale"}.get(n_fwd, f"{n_fwd / 1e6:.0f} M-char")
    residency = (f"device index {index_mb:.0f} MB (64-character buckets): Infinity-Cache (256 MB) resident, so this is the fraction of the HBM peak reached from cache"
                 if index_mb <= 256 else f"device index {index_mb:.0f} MB (64-character buckets): larger than the 256 MB Infinity Cache, bucket loads come from HBM")
    out = {
        "metric": f"{a.read_len}bp reads aligned/sec (inexact BWT backward search, IUPAC FM-index)", "value": round(value, 1), "unit": "reads/s",
        "n_gpus": world if not share else len({r % max(ndev, 1) for r in range(world)}), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32" if bwt.length < 0xFFFFFFFF else "u64", "data": "synthetic",
        "config": {"workload": f"{scale} synthetic multi-genome: {n_fwd} fwd chars (BWT length {bwt.length}), FASTQ shard of {a.pool} x {a.read_len} bp "
                               f"reads per GPU, one step = one resident batch of {B} reads (step s runs batch s mod {nb} from slot s mod {ns}), align {' '.join(flags)} (other params default)",
                   "name": a.config + ("" if world == 1 or a.config != "C3" else " (C4: the C3 workload sharded over the GPUs)"), "genome_mb": a.genome_mb, "flags": " ".join(flags),
                   "reads_per_gpu_per_step": B, "read_pool_per_gpu": a.pool, "batches_in_pool": nb, "slots_resident": ns, "read_len": a.read_len, "max_diff": a.ndiff,
                   "bwt_length": int(bwt.length), "sharding": f"reads x{world} (contiguous shards of one logical FASTQ), index replicated",
                   "steps_are_pipelined": "a slice parks its unfinished reads for the next step's slice; the timed region ends with a flush"},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": dom["device_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["device_frac"], "traffic": traffic, "traffic_measured_in_run": False, "traffic_source": traffic_src,
                     "traffic_over_device_bytes": round(traffic / dom["device_bytes_per_launch"], 3) if traffic and dom["device_bytes_per_launch"] else None,
                     "device_bytes_per_launch": dom["device_bytes_per_launch"],
                     "algorithmic_bytes_per_launch": int(dom["visits_per_step"] * a.steps / max(dom["launches"], 1) * ALG_BYTES_PER_VISIT),
                     "kernel_ms_per_launch": dom["ms_per_launch"], "achieved_ref_layout_GBs": dom["ref_layout_GBs"],
                     "s8d_check": s8d_check(dom, a),
                     "kernels": {"kl_search": k_search, "kl_calc_d": k_calcd},
                     "lanes_busy_of_64": round(st.lane_iterations / max(st.wave_iterations, 1), 1),
                     "reads_parked_per_step": int(st.n_parked_reads / a.steps),
                     "visit_split_per_step": {"calculate_d": int(vis_calcd / a.steps), "exact_tail": int((st.visits_single - vis_calcd) / a.steps),
                                              "expansion_O_alphabet": int(st.visits_alphabet / a.steps)},
                     "note": residency + "; `achieved` / `frac` count the bytes this layout has to move for the launch's rank visits, counted in-kernel: "
                             "128-byte buckets actually fetched (an L-1/U pair in one bucket is fetched once) + heap entries stored and loaded + "
                             "per-position records; `achieved_ref_layout_GBs` prices the same visits at the reference layout's 192 B (SURVEY 8d): a rate, not a fraction"},
        "calculate_d_table": dict(ctx.dtab_info(), note="kl_calc_d starts a read and its seed from the state after the first K steps of calculate_d, looked up by the last K bases "
                                  "(bit-identical records; DESIGN.md 3.4): `visits` stay the reference's count, the buckets of those steps are not fetched - kl_calc_d's device bytes and "
                                  "`device_frac` count what it really moves; built once per context, outside the timed region"),
        "hits_batch0": int(off0[-1]), "rerun_reads": int(st.n_overflow_reads), "kernel_ms_of_step_ms": round(kern_ms / (dt * 1e3), 4),
        "setup_s": {"genome_index_reads": round(t_build, 1), "index_to_hbm": round(t_ctx, 1),
                    "index": (json.load(open(fa + ".index_stats.json")) if os.path.exists(fa + ".index_stats.json") else None)},
    }
    if shard_check is not None:
        out["shard_sample_parity"] = shard_check
    if world > 1:
        out["per_rank_reads_per_s"] = {"min": min(rank_rates), "max": max(rank_rates)}
        out["per_rank_index_to_hbm_s"] = {"min": round(min(rank_ctx_ms) / 1e3, 2), "max": round(max(rank_ctx_ms) / 1e3, 2)}
        n1 = stored_n1_value(a)
        if n1:
            out["efficiency_vs_stored_n1"] = {"value": round(value / world / n1["value"], 4), "n1_reads_per_s": n1["value"], "n1_source": n1["source"],
                                              "note": "informational: this run's reads/s per GPU over a stored 1-GPU line of the same workload (the driver computes the official curve)"}
    if world == 1 and not a.no_extras:
        out["cpu_baseline"] = cpu_baseline(a, fa, fq, flags, off0, alns0, bw)
        out["end_to_end"] = end_to_end(ctx, p, batch, ns, B, value)
        out["rank_micro"] = rank_micro(ctx, index_mb)
        if a.ndiff != 0:
            out["also"] = {"n0": also_n0(ctx, bw, batch, nb, B)}
        ctx.flush()
        out["cli_end_to_end"] = cli_end_to_end(a, fa, fq, flags, value, ctx, bw)  # (last: it closes this process's context)
    print(json.dumps(out))
    grp.close()


def s8d_check(dom, a):
    """SURVEY 8(d)'s unit (192 B per rank visit: the REFERENCE layout's checkpoint row + BWT block) against the HBM peak.  It is no upper-bound
    check for this layout - a visit moves one 128-byte bucket, and an L-1 / U pair in one bucket is fetched once - so the priced rate may exceed
    the peak; the line says so here instead of in a note."""
    vis, bkt = dom["visits_per_step"], dom["bucket_bytes_per_step"] / DEV_BYTES_PER_BUCKET
    rate = dom["ref_layout_GBs"]
    return {"bytes": int(vis * a.steps / max(dom["launches"], 1) * ALG_BYTES_PER_VISIT), "rate_GBs": rate, "over_peak": bool(rate > HBM_PEAK_GBS),
            "buckets_per_visit": round(bkt / vis, 3) if vis else None,
            "why": f"{(bkt / vis if vis else 0):.2f} buckets fetched per visit x {DEV_BYTES_PER_BUCKET} B in this layout against 192 B per visit in the reference's: "
                   "`frac` follows the device bytes, this figure prices the reference layout and is not a fraction of anything"}


def source_hash():
    """sha256 over the kernel sources: a stored PMC profile is quoted only for the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bwbble_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(a, B, dom_name, dom):
    """HBM traffic of the dominant kernel per launch from the committed rocprofv3 --pmc passes of this workload (profiles/r<N>_*_pmc.json,
    written by tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate passes, corrected with the calibration measured by
    tools_exp/gather_bench: bucket requests x2, metadata requests x k) - only when the profile was taken on THIS kernel source."""
    import glob
    for prof in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")), reverse=True):
        pj = json.load(open(prof))
        if (pj.get("genome_mb"), pj.get("reads"), pj.get("ndiff"), pj.get("read_len", 100)) != (a.genome_mb, B, a.ndiff, a.read_len) or dom_name not in pj:
            continue
        if pj.get("source_hash") != source_hash():
            return None, f"{os.path.relpath(prof, ROOT)} was measured on other kernel sources (hash {pj.get('source_hash')}, now {source_hash()}): not quoted"
        pl = pj[dom_name].get("per_launch")
        if isinstance(pl, dict) and pl["slice"]["launches"] and (pl["drain"]["launches"] or dom_name != "kl_search"):
            # every dispatch of the profile was priced on its own (tools/pmc_traffic_summary.py): this run's traffic is a SUM - its slices at a
            # slice's measured bytes plus its draining launches at a drain's - over its launches
            n_drain = max(dom["launches"] - a.steps, 0) if dom_name == "kl_search" else 0
            n_slice = dom["launches"] - n_drain
            tot = n_slice * pl["slice"]["hbm_bytes_per_launch"] + n_drain * (pl["drain"]["hbm_bytes_per_launch"] or 0)
            return (tot / max(dom["launches"], 1),
                    f"{os.path.relpath(prof, ROOT)}: separate rocprofv3 --pmc passes of this command ({pj.get('steps_in_pass')} steps) on this kernel source (hash {pj['source_hash']}), "
                    f"every dispatch priced on its own; {pj.get('method', '')}; here: ({n_slice} slices x {pl['slice']['hbm_bytes_per_launch']:.4g} B + {n_drain} draining "
                    f"launch(es) x {(pl['drain']['hbm_bytes_per_launch'] or 0):.4g} B) / {dom['launches']} launches; NOT measured in this run (counters need the profiler attached)")
        # (the profile's own passes have fewer steps per draining launch than this run: scale its bytes per STEP to this run's launches)
        return (pj[dom_name]["hbm_bytes_per_step"] * a.steps / max(dom["launches"], 1),
                f"{os.path.relpath(prof, ROOT)}: separate rocprofv3 --pmc passes of this command on this kernel source (hash {pj['source_hash']}); "
                f"{pj.get('method', '')}; bytes per step there x steps / launches here; NOT measured in this run")
    return None, "not measured in this run (PMC counters need rocprofv3 passes: tools/pmc_traffic.sh) and no stored profile of this workload"


def stored_n1_value(a):
    """the 1-GPU bench line of the same workload kept under profiles/ (for the informational efficiency figure of a multi-GPU run)"""
    import glob
    # (newest round first; within a round the line taken with the driver's arguments before the short profiling runs)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_bench_line*.json")), key=lambda f: (int("".join(ch for ch in os.path.basename(f).split("_")[0] if ch.isdigit()) or 0), "driver_args" in f, f), reverse=True):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        c = j.get("config", {})
        if j.get("n_gpus") == 1 and (c.get("genome_mb"), c.get("read_len"), c.get("max_diff")) == (a.genome_mb, a.read_len, a.ndiff):
            return {"value": j["value"], "source": os.path.relpath(f, ROOT)}
    return None


def end_to_end(ctx, p, batch, ns, B, value):
    """The same batches with everything a caller pays: host -> pinned staging -> HBM, kernels, hit log -> host, in read order.
    Uploads and result copies ride their own streams next to the kernels, and a batch's result is fetched when its slot is needed
    again, ns batches later (the product's `align` loop, host/align_gpu.c).  10 batches: the draining launch at the end of the stream
    (the heaviest reads of the last batches, alone on the GPU for about a slice's time) weighs 1/10 here against 1/20 in `value`."""
    ctx.flush()
    ne = ns + 2
    t0 = time.perf_counter()
    for j in range(ne):
        if j >= ns:
            ctx.slot_result(j % ns)
        ctx.slot_upload(j % ns, p, *batch(j))
        ctx.slot_submit(j % ns)
    for j in range(max(0, ne - ns), ne):
        ctx.slot_result(j % ns)
    ctx.flush()
    dt = time.perf_counter() - t0
    v = ne * B / dt
    return {"value": round(v, 1), "unit": "reads/s", "batches": ne, "slots": ns, "of_value": round(v / value, 4),
            "includes": "H2D of reads (pinned staging), kl_calc_d + kl_search, D2H of the hit log, reordering into read order"}


def cli_end_to_end(a, fa, fq, flags, value, ctx, bw):
    """What a user of the product sees: the wall time of the C CLI - `bwbble align` from process start to exit - .bwt file -> host -> HBM,
    FASTQ parsing, alignment, .aln writing included.  The CLI overlaps all of it (host/align_gpu.c: loader threads, streamed context
    creation, parallel FASTQ scanner, one worker per GPU that also serialises its chunks' records, ordered writer).  Three runs:
    this rank's read pool (10 M reads) with the bench's flags; `long_stream`: the pool two and a half times over (>= 25 M reads: the one
    draining launch at the end of a stream and the index load weigh less, as in a real run); `n0`: the pool with the CLI's default -n 0,
    where the host stages matter most (2.3 M+ reads/s per GPU).  `host_pipeline`: the two host stages alone, on this box's cores.
    The Python context gives its memory back first (the CLI's context sizes its heap chunk pool from what is free)."""
    try:
        ctx.close()
    except Exception:
        pass
    # This process has just given ~200 GB of device memory back.  The driver hands such memory to ANOTHER process only after it has gone over
    # it, and a process that starts right away sits out those seconds in its first hipMalloc (measured: `context created at +5.39 s` for the
    # first CLI run after a bench process, +0.97 s for the next one - profiles/r6_ab_steps.txt session 4).  That is this benchmark's own
    # residue, not the CLI's start-up, so the CLI runs start after a pause; BWB_BENCH_SETTLE_S=0 shows the other number.
    # (the same before every CLI run: a CLI run that follows another pays for ITS 230 GB in the same way - 4 s of start-up in
    # profiles/r6_ab_steps.txt session 8)
    settle = float(os.environ.get("BWB_BENCH_SETTLE_S", 8))

    def run_cli(fl, fastq, n_reads, keep=None):
        out_aln = fastq + ".cli.aln"
        time.sleep(settle)
        t0 = time.perf_counter()
        r = subprocess.run([bw.HOST_BIN, "align"] + fl + [fa, fastq, out_aln], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-400:]}
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("GPUs:")]
        startup = [ln for ln in r.stdout.splitlines() if ln.startswith("start-up:")]
        res = {"value": round(n_reads / dt, 1), "unit": "reads/s", "reads": n_reads, "wall_s": round(dt, 2),
               "command": "bwbble align " + " ".join(fl) + " <fasta> <fastq> <out.aln>", "cli_summary": line[-1] if line else "",
               "cli_startup": startup[-1] if startup else ""}
        if keep:
            os.replace(out_aln, keep)
        else:
            try:
                os.remove(out_aln)
            except OSError:
                pass
        return res

    kept = fq + ".kept.aln"
    res = run_cli(flags, fq, a.pool, keep=kept)
    if "error" in res:
        return res
    res["of_value"] = round(res["value"] / value, 4)
    res["settle_s_before_each_run"] = settle
    # the two host stages of the pipeline on their own (`bwbble hostbench`: reads.c's scanner + encoder, aln_io.c's chunk serialiser)
    try:
        hb = subprocess.run([bw.HOST_BIN, "hostbench", fq, kept], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, check=True)
        hj = json.loads(hb.stdout.strip().splitlines()[-1])
        res["host_pipeline"] = {"parse_reads_per_s": hj["parse_reads_per_s"], "write_records_per_s": hj.get("write_records_per_s"), "reads": hj["parse_reads"],
                                "parse_stages_s": hj.get("parse_stages_s"), "cores": os.cpu_count(), "openmp_team": "at most 32 threads per region (bwb_host.h)", "what": "FASTQ record scan (parallel, verified against the sequential scan) + base encoding; .aln record serialisation of the run above"}
    except Exception as e:  # noqa: BLE001
        res["host_pipeline"] = {"error": str(e)[-200:]}
    # f1: `bwbble aln2sam` on the same records - SA(L) of every mapped read by the invPsi walk on the GPU (k_locate: up to 31
    # dependent rank-block visits per row), MAPQ / CIGAR / text on the host's cores
    try:
        if over_budget(60):
            raise RuntimeError("skipped: time budget (BWB_BENCH_BUDGET_S)")
        sam = fq + ".cli.sam"
        t0 = time.perf_counter()
        r = subprocess.run([bw.HOST_BIN, "aln2sam"] + (["-n", str(a.ndiff)] if a.ndiff else []) + [fa, fq, kept, sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            res["aln2sam"] = {"error": (r.stdout + r.stderr)[-300:]}
        else:
            ln = [x for x in r.stdout.splitlines() if x.startswith("SA lookups on the GPU:")]
            f = ln[-1].split() if ln else []
            rows, visits, kms = (int(f[6]), int(f[9]), float(f[11])) if len(f) > 11 else (0, 0, 0.0)
            res["aln2sam"] = {"wall_s": round(dt, 2), "reads": a.pool, "reads_per_s": round(a.pool / dt, 1), "sam_bytes": os.path.getsize(sam),
                              "k_locate": {"rows": rows, "visits": visits, "kernel_ms": kms, "Gvisits_per_s": round(visits / kms / 1e6, 2) if kms else 0.0,
                                           "device_GBs": round(visits * DEV_BYTES_PER_BUCKET / kms / 1e6, 1) if kms else 0.0,
                                           "device_frac": round(visits * DEV_BYTES_PER_BUCKET / kms / 1e6 / HBM_PEAK_GBS, 4) if kms else 0.0,
                                           "note": "a chain of dependent visits per row (31 at most, SA samples every 32 rows): latency-bound by construction, the fraction says how much of it the rows in flight hide"},
                              "command": "bwbble aln2sam <fasta> <fastq> <aln> <sam> (wall: .bwt + SA -> host -> HBM, .aln and FASTQ loaded, SA lookups, SAM text by all cores)"}
        try:
            os.remove(sam)
        except OSError:
            pass
    except Exception as e:  # noqa: BLE001
        res["aln2sam"] = {"error": str(e)[-200:]}
    try:
        os.remove(kept)
    except OSError:
        pass
    if a.config == "C3":
        res["n0"] = run_cli(["-n", "0"], fq, a.pool) if not over_budget(30) else {"skipped": "time budget (BWB_BENCH_BUDGET_S)"}
    if a.config == "C3" and over_budget(110):
        res["long_stream"] = {"skipped": "time budget (BWB_BENCH_BUDGET_S): a 25 M-read CLI run takes about 100 s"}
    elif a.config == "C3":
        # >= 25 M reads: the pool's FASTQ two and a half times over (the same reads again: only the length of the stream matters here)
        long_fq = fq + ".long.fq"
        with open(long_fq, "wb") as g:
            for rep in range(3):
                with open(fq, "rb") as f:
                    if rep < 2:
                        shutil.copyfileobj(f, g, 1 << 24)
                    else:  # half of the file, cut at a record boundary
                        half = os.path.getsize(fq) // 2
                        data = f.read(half)
                        cut = data.rfind(b"\n@")  # (this generator's qualities never start a line with '@')
                        g.write(data[:cut + 1])
                        n_half = data[:cut + 1].count(b"\n") // 4
        n_long = 2 * a.pool + n_half
        lr = run_cli(flags, long_fq, n_long)
        if "error" not in lr:
            lr["of_value"] = round(lr["value"] / value, 4)
        res["long_stream"] = lr
        try:
            os.remove(long_fq)
        except OSError:
            pass
    return res


def rank_micro(ctx, index_mb):
    """Stand-alone random Occ16: one 128-byte bucket per query, all 15 codes ranked, nothing else."""
    n = 1 << 26
    res = {"queries": n, "index_MB": round(index_mb, 1), "peak_GBs": HBM_PEAK_GBS}
    for name, lane in (("octet_layout", False), ("lane_layout", True)):
        ms, _ = ctx.rank_bench(n, iters=3, seed=7, lane=lane)
        res[name] = {"ms": round(ms, 3), "Gvisits_per_s": round(n / ms / 1e6, 2), "device_GBs": round(n * DEV_BYTES_PER_BUCKET / ms / 1e6, 1),
                     "device_frac": round(n * DEV_BYTES_PER_BUCKET / ms / 1e6 / HBM_PEAK_GBS, 4),
                     "ref_layout_GBs": round(n * ALG_BYTES_PER_VISIT / ms / 1e6, 1)}
    res["note"] = ("device_frac = 128-byte buckets actually fetched / launch time / HBM peak: the roofline fraction of this kernel. "
                   "ref_layout_GBs prices the same visits at the reference layout's 192 B (SURVEY 8d), a rate for comparison; "
                   "it is 1.5 x device_GBs by construction and not a fraction of anything")
    return res


def also_n0(ctx, bw, batch, nb, B):
    """the reference's literal CLI default is -n 0 (align.c:26): same reads, same index, reported next to the main line"""
    p0 = bw.params(["-n", "0"])
    k = min(nb, 2)
    for j in range(k):
        ctx.slot_upload(j, p0, *batch(j))
    for j in range(k):
        ctx.slot_submit(j)
    ctx.flush()  # warm-up
    ctx.reset_stats()
    t0 = time.perf_counter()
    for j in range(k):
        ctx.slot_submit(j)
    ctx.flush()
    d0 = time.perf_counter() - t0
    s0 = ctx.stats()
    ach0 = s0.visits_calc_d * ALG_BYTES_PER_VISIT / (s0.ms_calc_d * 1e-3) / 1e9
    dev0 = (s0.bucket_loads_calc_d * DEV_BYTES_PER_BUCKET + k * B * rec_bytes_written(100)) / (s0.ms_calc_d * 1e-3) / 1e9
    return {"workload": f"{k} of the same batches, align -n 0 (CLI default)", "value": round(k * B / d0, 1), "unit": "reads/s",
            "ms_per_step": round(d0 / k * 1e3, 3), "dominant_kernel": "kl_calc_d", "kernel_ms_per_launch": round(s0.ms_calc_d / max(s0.launches_calc_d, 1), 3),
            "device_GBs": round(dev0, 1), "device_frac": round(dev0 / HBM_PEAK_GBS, 4), "ref_layout_GBs": round(ach0, 1),
            "visits_per_step": int((s0.visits_single + s0.visits_alphabet) / k)}


def cpu_baseline(a, fa, fq, flags, off, alns, bw):
    """Times the REAL reference (oracle/_ref/bwbble, OpenMP) when its prebuilt binary is present, else the CPU oracle port, on
    the first reads of the same FASTQ, and re-checks parity of the sample against the GPU result.

    The reference cuts a batch into one contiguous chunk per thread (inexact_match.c:111-116), so with a heavy-tailed cost
    per read a small sample measures its slowest thread, and 2 threads per core on a memory-latency-bound walk need not help.
    So: a sweep of -t over {cores/4, cores/2, cores} at 64 reads per thread, then the best -t again at >= 200 reads per
    thread; the load (index + FASTQ) is timed with a 1-read FASTQ at the same -t and subtracted.  value = the best rate."""
    import oracle_lib
    cores = os.cpu_count() or 1
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "bwbble")
    res = {"cores": cores, "unit": "reads/s"}
    def head_fastq(n_reads, path):
        with open(fq) as f, open(path, "w") as g:
            for i, line in enumerate(f):
                if i >= 4 * n_reads:
                    break
                g.write(line)

    if os.path.exists(ref_bin):
        # The reference times with clock() (CPU time summed over threads, inexact_match.c:104,150): wall-time the whole process.
        def t_run(path, out, t):
            t0 = time.perf_counter()
            subprocess.run([ref_bin, "align"] + flags + ["-t", str(t), fa, path, out], check=True, stdout=subprocess.DEVNULL)
            return time.perf_counter() - t0
        one = fq + ".one"
        head_fastq(1, one)
        def measure(t, n_reads):
            n_reads = min(n_reads, a.reads)
            sfq = fq + f".sample{n_reads}"
            if not os.path.exists(sfq):
                head_fastq(n_reads, sfq)
            t_load = t_run(one, one + ".aln", t)
            t_all = t_run(sfq, sfq + ".aln", t)
            sec = max(t_all - t_load, 1e-3)
            return {"threads": t, "reads": n_reads, "reads_per_thread": round(n_reads / t, 1), "wall_s": round(t_all, 2), "load_s": round(t_load, 2),
                    "reads_per_s": round(n_reads / sec, 1), "reads_per_s_per_thread": round(n_reads / sec / t, 2)}, sfq
        ts = sorted({max(1, cores // 4), max(1, cores // 2), cores})
        sweep = [measure(t, 64 * t)[0] for t in ts] if not a.cpu_sample else []
        best_t = max(sweep, key=lambda r: r["reads_per_s"])["threads"] if sweep else cores
        final, sfq = measure(best_t, a.cpu_sample or 200 * best_t)
        sample = final["reads"]
        ref_bytes = open(sfq + ".aln", "rb").read()
        phys = cores // 2 if cores >= 4 else cores  # the GPU boxes expose 2 hardware threads per core
        # The reference freads the 12 GB index from one thread (bwt.c:90-125): first touch puts it on that thread's NUMA node, and the
        # other socket's threads walk it over the inter-socket link.  The same binary under `numactl --interleave=all` (when the box
        # has it) says what the reference does once its pages are spread - reported next to the plain run, not instead of it.
        import shutil
        il_helper = os.path.join(ROOT, "oracle", "interleave_exec")  # (the same policy through the raw system call: the GPU boxes have no numactl)
        il_cmd = ["numactl", "--interleave=all"] if shutil.which("numactl") else ([il_helper] if os.path.exists(il_helper) else None)
        if il_cmd and not a.cpu_sample:
            try:
                nodes_now = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
            except OSError:
                nodes_now = 1
            if nodes_now < 2:
                il_cmd = None
        if il_cmd and not a.cpu_sample:
            def t_run_il(path, out, t):
                t0 = time.perf_counter()
                subprocess.run(il_cmd + [ref_bin, "align"] + flags + ["-t", str(t), fa, path, out], check=True, stdout=subprocess.DEVNULL)
                return time.perf_counter() - t0
            il = []
            for t in [best_t]:  # (the plain sweep's best -t, on the same sample as `final`: the interleaved sweeps of round 4 had the same best -t)
                n_il = min(200 * t if t == best_t else 64 * t, a.reads)
                sfq_il = fq + f".sample{n_il}"
                if not os.path.exists(sfq_il):
                    head_fastq(n_il, sfq_il)
                tl, ta = t_run_il(one, one + ".il.aln", t), t_run_il(sfq_il, sfq_il + ".il.aln", t)
                sec = max(ta - tl, 1e-3)
                il.append({"threads": t, "reads": n_il, "wall_s": round(ta, 2), "load_s": round(tl, 2), "reads_per_s": round(n_il / sec, 1),
                           "reads_per_s_per_thread": round(n_il / sec / t, 2)})
            bi = max(il, key=lambda r: r["reads_per_s"])
            res["interleaved"] = {"value": bi["reads_per_s"], "threads": bi["threads"], "reads_per_s_per_thread": bi["reads_per_s_per_thread"],
                                  "sweep": il, "command": " ".join(os.path.basename(x) for x in il_cmd) + " oracle/_ref/bwbble align ..."}
        else:
            res["interleaved"] = None if a.cpu_sample else "one NUMA node, or neither numactl nor oracle/interleave_exec on this box"
        try:
            nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
        except OSError:
            nodes = None
        res["numa_nodes"] = nodes
        res.update({"cores": best_t, "host_hardware_threads": cores,  # (`cores` = the threads the quoted run used, as the bench contract asks)
                    "value": final["reads_per_s"], "kind": "reference", "threads": best_t, "reads_per_s_per_thread": final["reads_per_s_per_thread"],
                    "reads_per_s_per_core": round(final["reads_per_s"] / min(best_t, phys), 2), "sweep": sweep, "final": final,
                    "sample": f"first {sample} reads of the same FASTQ ({final['reads_per_thread']} per thread), oracle/_ref/bwbble align -t {best_t} "
                              f"(best of the -t sweep {ts}); wall {final['wall_s']}s minus {final['load_s']}s load at the same -t"})
    else:
        orc = oracle_lib.load()
        seqs, lens = bw.load_fastq_codes(fq, max_reads=a.cpu_sample or min(a.reads, 50000))
        sample = len(lens)
        idx = orc.load_index(fa + ".bwt")
        ref_bytes, _, sec = orc.align_encoded(idx, seqs, lens, orc.params(flags + ["-t", str(cores)]))
        res.update({"value": round(sample / sec, 1), "kind": "port", "threads": cores,
                    "sample": f"first {sample} reads, oracle/libbwb_oracle.so with {cores} OpenMP threads, align loop only"})
    res["parity_on_sample"] = bool(bw.aln_bytes(off[:sample + 1], alns[:int(off[sample])]) == ref_bytes)
    return res


if __name__ == "__main__":
    main()
