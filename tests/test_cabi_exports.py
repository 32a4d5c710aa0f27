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


def test_ctypes_mirror_matches_the_header_layout(built, tmp_path):
    """The Python mirror (bwbble_amd/__init__.py) and include/bwbble_hip.h must agree on every struct: sizes and field offsets
    as the C compiler sees them."""
    import subprocess
    import numpy as np
    src = tmp_path / "layout.c"
    fields = {"bwb_params": [n for n, _ in bw.Params._fields_], "bwb_stats": [n for n, _ in bw.Stats._fields_],
              "bwb_result": [n for n, _ in bw.Result._fields_], "bwb_aln": ["L", "U", "score", "num_mm", "num_gapo", "num_gape", "reserved", "aln_length", "gap_run"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "bwbble_hip.h"', 'int main(void) {']
    for st, fs in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            lines.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines += ['printf("slots %d\\n", BWB_MAX_SLOTS);', 'return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, text=True).stdout.splitlines())
    for st, cls in (("bwb_params", bw.Params), ("bwb_stats", bw.Stats), ("bwb_result", bw.Result)):
        assert int(got[st]) == ctypes.sizeof(cls), st
        for f, _ in cls._fields_:
            assert int(got[f"{st}.{f}"]) == getattr(cls, f).offset, f"{st}.{f}"
    assert int(got["bwb_aln"]) == bw.ALN_DTYPE.itemsize
    for f in fields["bwb_aln"]:
        assert int(got[f"bwb_aln.{f}"]) == bw.ALN_DTYPE.fields[f][1], f
    assert int(got["slots"]) == bw.MAX_SLOTS
