#!/usr/bin/env python3
"""Developer tool: static instruction statistics of the library's kernels from the ISA that hipcc writes with --save-temps.

    tools/isa_stats.py [-D<flag> ...]                 per kernel: vector / scalar / LDS / memory instructions, v_bcnt, scalar-spill moves
                                                      (v_readlane / v_writelane), registers, scratch, occupancy
    tools/isa_stats.py --lines <kernel-substring> [-D<flag> ...]
                                                      the same kernel's instructions attributed to the source lines of bwb_lane.h
                                                      (line tables), summed over the regions of kl_search's loop

Static counts: a loop body counts once, both sides of a branch count.  What they are good for: comparing two builds of the same
kernel (an experiment flag against the product), and finding where the instructions of the straight-line hot path are.  Round 3 used
them for the 64-character buckets (3074 -> 2796 vector instructions in kl_search<u64>) and for the exp-r4 builds.
No GPU needed (hipcc cross-compiles)."""
import bisect
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_hip.hip")
INSTR = re.compile(r"\s+((v|s|ds|global|buffer|flat|scratch)_\w+)")


def compile_isa(flags, lines=False):
    d = tempfile.mkdtemp(prefix="isa_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Wno-unused-value",
           "--save-temps", "-o", os.path.join(d, "lib.so"), SRC] + flags + (["-gline-tables-only"] if lines else [])
    subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(d, "bwb_hip-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(path):
    cur, out = None, collections.OrderedDict()
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is not None:
            out[cur].append(ln) # (up to the next kernel's label: the resource-usage comments follow .end_amdhsa_kernel)
    return out


def file_numbers(path):
    fileno = {}
    for ln in open(path):
        m = re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if m:
            fileno[int(m.group(1))] = m.group(2)
    return fileno


def summary(path):
    for name, body in kernels(path).items():
        if not any(k in name for k in ("kl_", "rank_bench_lane")):
            continue
        c = collections.Counter()
        for ln in body:
            m = INSTR.match(ln)
            if m:
                c[m.group(2)] += 1
                if "v_readlane" in ln or "v_writelane" in ln:
                    c["spill_moves"] += 1
                if "v_bcnt" in ln:
                    c["v_bcnt"] += 1
            m = re.match(r"; (NumVgprs|ScratchSize|Occupancy): (\d+)", ln)
            if m:
                c[m.group(1)] = int(m.group(2))
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.split("(")[0]
        print(f"{short:42s} vector {c['v']:5d} scalar {c['s']:5d} lds {c['ds']:4d} global {c['global']:4d} flat {c['flat']:3d} v_bcnt {c['v_bcnt']:4d} "
              f"spill_moves {c['spill_moves']:4d} | vgprs {c['NumVgprs']} scratch {c['ScratchSize']} waves/SIMD {c['Occupancy']}")


REGIONS = [("pair_setup", "void pair_setup"), ("wave_gather", "void wave_gather"), ("rank passes", "void block_pops16"), ("kid_get", "struct KidCtx"),
           ("wave_children", "uint32_t wave_children"), ("list_add", "struct ListW"), ("kl_calc_d", "void kl_calc_d"), ("LHeap", "struct LEntry"),
           ("emit_entry", "void emit_entry"), ("search: prologue", "void kl_search"), ("search: admission / park", "bool admit = !active"),
           ("A pick", "---- A: pick"), ("B record", "---- B: one round"), ("rank call", "KidCtx<P> kc;"), ("C group", "---- C: act on it"),
           ("C prune / hit", "} else if (from_pop) {"), ("C expansion logic", "---- expansion :377-504"), ("C reserve", "uint32_t st0 = h.reserve(h.cst, k0, ovf);"),
           ("C templates", "/* child entry templates */"), ("C gap pushes", "{ /* gap pushes"), ("C mismatch / match loops", "const uint32_t sm = (uint32_t)STATE_M | (alen1 << 2);"),
           ("C commit", "h.num_entries += nGc + nX + n0;"), ("exact step", "if (exact_step && need_rank) {"), ("exact done", "if (exact_done && !ovf && seeding) {"),
           ("reload top", "const bool reload = active"), ("finish", "const SlotDesc &d = descs[myslot];"), ("epilogue", "atomicAdd(&stats[STAT_BKT_SEARCH]")]


def by_region(path, want):
    src = open(os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_lane.h")).read().split("\n")
    marks, last = [], 0
    for name, pat in REGIONS:
        for i in range(last, len(src)):
            if pat in src[i]:
                marks.append((i + 1, name))
                last = i
                break
    starts = [m[0] for m in marks]
    for kname, body in kernels(path).items():
        if want not in kname:
            continue
        v, s, cur, fileno = collections.Counter(), collections.Counter(), None, file_numbers(path)
        for ln in body:
            m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
            if m:
                cur = (int(m.group(1)), int(m.group(2)))
                continue
            m = INSTR.match(ln)
            if m and cur:
                if fileno.get(cur[0], "") != "bwb_lane.h" or cur[1] == 0:
                    key = "(no line / other headers)"
                else:
                    i = bisect.bisect_right(starts, cur[1]) - 1
                    key = marks[i][1] if i >= 0 else "(before)"
                (v if m.group(2) == "v" else s if m.group(2) == "s" else collections.Counter())[key] += 1
        print(subprocess.run(["c++filt", kname], capture_output=True, text=True).stdout.split("(")[0])
        for name in [m[1] for m in marks] + ["(no line / other headers)"]:
            print(f"  {name:28s} vector {v[name]:5d} scalar {s[name]:5d}")
        print(f"  {'total':28s} vector {sum(v.values()):5d} scalar {sum(s.values()):5d}")


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--lines":
        by_region(compile_isa(args[2:], lines=True), args[1])
    else:
        summary(compile_isa(args))
