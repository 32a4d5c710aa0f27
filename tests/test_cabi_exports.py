"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/bwbble_hip.h declares."""
import ctypes
import os
import re

import bwbble_amd as bw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "bwbble_hip.h")).read()
    declared = set(re.findall(r"\b(bwb_(?:hip_)?[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(bw.EXPORTS), declared ^ set(bw.EXPORTS)
    lib = ctypes.CDLL(bw.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name


def test_default_params_mirror_reference_defaults(built):
    p = bw.params()  # set_default_aln_params, mg-aligner/align.c:22-38
    assert (p.max_diff, p.max_gapo, p.max_gape, p.max_entries) == (0, 1, 6, 3000000)
    assert (p.mm_score, p.gapo_score, p.gape_score) == (3, 11, 4)
    assert (p.seed_length, p.max_diff_seed, p.max_best, p.no_indel_length, p.is_multiref) == (32, 2, 30, 5, 1)
    q = bw.params(["-n", "3", "-o", "2", "-l", "20"])
    assert (q.max_diff, q.max_gapo, q.seed_length) == (3, 2, 20)


def test_code_object_targets_gfx950(built):
    data = open(bw.LIB_PATH, "rb").read()
    assert b"gfx950" in data
    for k in (b"kl_search", b"kl_calc_d", b"k_rank16", b"k_relayout", b"k_rank_bench_lane"):
        assert k in data
