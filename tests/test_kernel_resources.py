"""CPU-only (hipcc cross-compiles): properties of the generated gfx950 code that the design depends on.

  * kl_search keeps three waves per SIMD (at most 168 VGPRs) and the headline instantiation <u64 positions, 16-byte entries> has no
    scratch and no VGPR spill;
  * the registers written by the in-place prefetches (bwb_lane.h: prefetch128 / prefetch32) are not touched before a vmcnt(0) wait
    (tools/check_prefetch_regs.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


BUILD = os.path.join(ROOT, "bwbble_amd", "build")


def kept(name):
    """the assembly and the resource remarks that `make` kept from the hipcc invocation that produced the shipped library `name` (and on
    which the Makefile rule has already run the proof: a library that fails it is deleted)"""
    import subprocess
    import check_prefetch_regs as cpr
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "bwbble_amd"), "all", "testlib"], check=True)
    d = os.path.join(BUILD, name)
    return cpr.check_asm(os.path.join(d, "bwb_hip-hip-amdgcn-amd-amdhsa-gfx950.s"), remarks_file=os.path.join(d, "remarks.txt"))


@pytest.fixture(scope="module")
def compiled():
    return kept("lib")


def test_prefetched_registers_are_not_read_before_the_wait(compiled):
    res, _ = compiled
    assert len([k for k in res if "kl_search" in k]) == 8  # <u32|u64 positions> x <16|32-byte entries> x <multi-genome | -S>
    assert len([k for k in res if "kl_calc_d" in k]) == 2  # <u32|u64 positions>: the four-interval group of the current list (round 5)
    for k, (sites, errs) in res.items():
        assert sites >= (2 if "kl_search" in k else 4), k  # the uncovered heap entry and the chunk header word; kl_calc_d: the group's four loads
        assert not errs, errs


def test_prefetched_registers_in_the_test_build():
    """the small-superblock test build (make testlib) is a different compilation: the same proof for it"""
    res, _ = kept("testlib")
    assert len(res) == 10
    for k, (sites, errs) in res.items():
        assert sites >= 2 and not errs, (k, errs)


def test_search_kernel_register_budget(compiled):
    _, remarks = compiled
    ks = {k: v for k, v in remarks.items() if "kl_search" in k}
    assert len(ks) == 8
    for k, ru in ks.items():
        assert ru["Occupancy"] >= 3 and ru["VGPRs"] <= 168, (k, ru)
    for k, ru in ks.items():
        # the compiler turns an atomic with a provably wave-uniform address into its own reduction, a scalar loop over the 64 lanes: twelve of
        # them in the rare paths (read end, allocation, statistics); a thirteenth once sat in the per-iteration path and cost 21 % (session 10)
        assert ru["ComputeLoops"] <= 12, (k, ru["ComputeLoops"])
    head = [v for k, v in ks.items() if "kl_searchImLb0ELb1E" in k][0]  # 64-bit positions (GRCh37 scale), 16-byte heap entries (-o <= 1), multi-genome
    # no vector register spilled and no scratch instruction in it (the frame itself may keep a few bytes that nothing touches)
    assert head["VGPRs Spill"] == 0 and head["ScratchOps"] == 0 and head["ScratchSize"] <= 64, head


def test_the_shipped_libraries_are_the_checked_ones():
    """the proof is about the .so that ships: the assembly is kept by the same hipcc invocation, and is not older than the library"""
    for name, so in (("lib", "libbwbble_hip.so"), ("testlib", "libbwbble_hip_test.so")):
        kept(name)
        d = os.path.join(BUILD, name)
        so_path = os.path.join(ROOT, "bwbble_amd", so)
        assert os.path.exists(so_path) and "HAZARD" not in open(os.path.join(d, "prefetch_check.txt")).read()
        assert abs(os.path.getmtime(so_path) - os.path.getmtime(os.path.join(d, "bwb_hip-hip-amdgcn-amd-amdhsa-gfx950.s"))) < 120


def test_checker_sees_a_hazard():
    """the checker is not vacuous: a register that is read between the prefetch and the wait is reported"""
    import check_prefetch_regs as cpr
    body = ["s_and_saveexec_b64 s[12:13], s[10:11]", "global_load_dwordx4 v[38:41], v[50:51], off", "s_mov_b64 exec, s[12:13]",
            "v_add_u32_e32 v1, v2, v3", "v_mov_b32_e32 v7, v40", "s_waitcnt vmcnt(0)"]
    n, errs = cpr.check_kernel("k", body)
    assert n == 1 and len(errs) == 1
    body[4] = "v_mov_b32_e32 v7, v42"
    assert cpr.check_kernel("k", body) == (1, [])
    body[5] = "s_waitcnt vmcnt(1)"  # not a full wait: the scan runs on to the end without finding one, but nothing touches the registers
    body.append("s_cbranch_execz .LBB0_1"); body.append(".LBB0_1:"); body.append("v_mov_b32_e32 v8, v39"); body.append("s_endpgm")
    n, errs = cpr.check_kernel("k", body)
    assert n == 1 and len(errs) == 1
