"""CPU checks of bench.py's host logic: configuration defaults (C2, C3 = C4 per GPU, C5) and the hash that guards the
stored PMC traffic profile (no GPU, no oracle)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(argv, monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("BWB_BENCH_GENOME_MB", "BWB_BENCH_POOL", "BWB_BENCH_READS", "BWB_BENCH_NDIFF", "BWB_BENCH_CONFIG"):
        monkeypatch.delenv(k, raising=False)
    return bench.parse_args()


def test_config_defaults(monkeypatch):
    a = parse([], monkeypatch)
    assert (a.config, a.genome_mb, a.pool, a.reads, a.ndiff, a.read_len, a.gpus) == ("C3", 3100.0, 10000000, 2500000, 3, 100, 1)
    a = parse(["--gpus", "8"], monkeypatch)  # config C4: 100 M reads over 8 GPUs
    assert a.pool == 12500000 and a.reads == 2500000
    a = parse(["--config", "C5"], monkeypatch)
    assert (a.read_len, a.ndiff, a.extra_flags) == (150, 5, ["-o", "1", "-e", "6", "-l", "32", "-k", "2"]) and a.genome_mb == 3100.0
    a = parse(["--config", "C2", "--reads", "500000"], monkeypatch)
    assert (a.genome_mb, a.reads, a.pool) == (48.0, 500000, 4000000)


def test_stored_pmc_profile_is_quoted_only_for_the_kernels_it_was_measured_on(monkeypatch):
    """bench.py quotes the newest profiles/r<N>_c3_pmc.json as roofline.traffic only when its hash equals that of bwbble_amd/csrc/*; on
    other sources it says so instead (a kernel edit makes the stored traffic figure disappear from the line, not go stale)."""
    import bench
    import glob
    pj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_c3_pmc.json")))[-1]))
    assert pj["kl_search"]["hbm_bytes_per_step"] > pj["kl_search"]["device_bytes_per_step"] > 0
    cal = pj["calibration_7GiB_table"]
    assert abs(cal["k_coop"]["RDREQ_per_request"] - 1.0) < 0.01 and abs(cal["k_meta8"]["RDREQ_per_request"] - 1.0) < 0.01
    a = parse([], monkeypatch)
    dom = {"launches": 21}
    a.steps = 20
    traffic, src = bench.measured_traffic(a, a.reads, "kl_search", dom)
    if pj["source_hash"] == bench.source_hash():
        pl = pj["kl_search"].get("per_launch")
        if isinstance(pl, dict):  # (round 6) every dispatch priced on its own: the run's traffic is a sum over its slices and its draining launch
            want = (20 * pl["slice"]["hbm_bytes_per_launch"] + 1 * pl["drain"]["hbm_bytes_per_launch"]) / 21
            assert abs(traffic - want) < 1e-6 * want and "every dispatch priced on its own" in src
            assert pl["slice"]["launches"] >= 8 and pl["drain"]["launches"] >= 1
        else:
            assert traffic == pj["kl_search"]["hbm_bytes_per_step"] * 20 / 21
        assert "NOT measured in this run" in src
    else:
        assert traffic is None and "other kernel sources" in src
    monkeypatch.setattr(bench, "source_hash", lambda: "0" * 16)
    traffic, src = bench.measured_traffic(a, a.reads, "kl_search", dom)
    assert traffic is None and "other kernel sources" in src
