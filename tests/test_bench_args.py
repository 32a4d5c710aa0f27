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


def test_pmc_summary_prices_every_dispatch_on_its_own(tmp_path):
    """tools/pmc_traffic_summary.py on synthetic counter CSVs and launch logs: a slice's and the draining launch's bytes are told apart
    (bucket requests tallied at 64 B get their second 64 B from the library's own per-launch bucket count), and bench.py turns them into a
    SUM over a run's launches."""
    import csv
    import subprocess
    import sys
    out = tmp_path / "pmc"
    for sub in ("calib", "rd", "wr"):
        (out / sub).mkdir(parents=True)
    def write(sub, rows):
        with open(out / sub / "run_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            w.writeheader()
            w.writerows(rows)
    write("calib", [{"Dispatch_Id": 1, "Kernel_Name": "k_coop", "Counter_Name": "TCC_EA0_RDREQ_sum", "Counter_Value": float(1 << 27)}])
    ks = "void kl_search<unsigned long, false, true>(...)"
    rd, wr, log = [], [], []
    for d, (bk, meta, wreq, drains) in enumerate([(1000, 500, 300, False), (1200, 700, 340, False), (100, 90, 20, True)], start=1):
        rd += [{"Dispatch_Id": d, "Kernel_Name": ks, "Counter_Name": "TCC_EA0_RDREQ_sum", "Counter_Value": bk + meta},
               {"Dispatch_Id": d, "Kernel_Name": ks, "Counter_Name": "TCC_EA0_RDREQ_32B_sum", "Counter_Value": 0}]
        wr += [{"Dispatch_Id": d, "Kernel_Name": ks, "Counter_Name": "WRITE_SIZE", "Counter_Value": wreq * 32 / 1024},
               {"Dispatch_Id": d, "Kernel_Name": ks, "Counter_Name": "TCC_EA0_WRREQ_sum", "Counter_Value": wreq}]
        log.append({"kernel": "kl_search", "class": 0, "slot": d - 1, "drains": drains, "ms": 5.0, "buckets": bk, "entries_stored": 10, "entries_loaded": 20, "records_loaded": 30})
    write("rd", rd)
    write("wr", wr)
    for name in ("rd_launches.jsonl", "wr_launches.jsonl"):
        open(out / name, "w").write("".join(json.dumps(l) + "\n" for l in log))
    line = {"metric": "x", "config": {"genome_mb": 1.0, "reads_per_gpu_per_step": 10, "max_diff": 3, "read_len": 100, "flags": "-n 3"},
            "roofline": {"kernels": {"kl_search": {"bucket_bytes_per_step": 1150 * 128, "device_bytes_per_step": 1150 * 128 + 60 * 16, "ms_per_launch": 5.0}}}}
    open(out / "rd.json", "w").write(json.dumps(line) + "\n")
    dst = tmp_path / "x_pmc.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic_summary.py"), str(out), str(dst), "2"], check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    pj = json.load(open(dst))
    pl = pj["kl_search"]["per_launch"]
    assert pl["slice"]["launches"] == 2 and pl["drain"]["launches"] == 1
    # a launch's bytes: RDREQ x 64 + its buckets x 64 (the second half of a 128-byte bucket request) + its writes
    want = [(1500 * 64 + 1000 * 64 + 300 * 32), (1900 * 64 + 1200 * 64 + 340 * 32), (190 * 64 + 100 * 64 + 20 * 32)]
    assert [q["hbm_bytes"] for q in pl["launches"]] == want
    assert pl["slice"]["hbm_bytes_per_launch"] == (want[0] + want[1]) / 2 and pl["drain"]["hbm_bytes_per_launch"] == want[2]
    assert pl["launches"][0]["device_bytes"] == 1000 * 128 + 30 * 16 + 30 * 16
