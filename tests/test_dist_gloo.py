"""N>1 path on CPU: two processes over gloo exercise the sharding and the benchmark's timing protocol
(barrier, max-over-ranks, sum of work) that bench.py uses with RCCL on the GPU box.  No GPU, no collective on data."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
from bwbble_amd import dist
g = dist.Group(backend="gloo")
lo, hi = dist.shard_bounds(1000003, g.world, g.rank)
g.barrier()
dt, kms, vis = g.reduce_step(1.0 + g.rank, 10.0 * (g.rank + 1), float(hi - lo))
pairs = g.all_gather_pairs(100 + g.rank, 100 + (g.rank + 1) %% g.world)  # (own checksum, checksum of the right neighbour's sample)
ring_ok = all(pairs[(r + 1) %% g.world][0] == pairs[r][1] for r in range(g.world))
print(json.dumps({"rank": g.rank, "world": g.world, "lo": lo, "hi": hi, "dt": dt, "kms": kms, "vis": vis, "ring_ok": ring_ok}))
g.close()
""" % ROOT


def test_two_rank_gloo_protocol(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(eval(o.strip().splitlines()[-1].replace("true", "True")))
    outs.sort(key=lambda d: d["rank"])
    assert [o["world"] for o in outs] == [2, 2]
    # contiguous, disjoint, complete shards in rank order (the .aln of rank r precedes that of rank r+1)
    assert outs[0]["lo"] == 0 and outs[0]["hi"] == outs[1]["lo"] and outs[1]["hi"] == 1000003
    for o in outs:
        assert o["dt"] == 2.0 and o["kms"] == 20.0 and o["vis"] == 1000003.0 and o["ring_ok"]


def test_shard_bounds_cover_everything():
    from bwbble_amd import dist
    for n in (0, 1, 7, 262144, 1000003):
        for world in (1, 2, 3, 8):
            b = [dist.shard_bounds(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


def test_bench_refuses_a_rank_count_it_was_not_launched_with():
    """`bench.py --gpus N` inside a launcher that started a different number of ranks is an error, before anything touches a GPU
    (round 1 measured one GPU and printed n_gpus: 1 whatever --gpus said)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--no-extras"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "launcher started 1 rank" in r.stderr
