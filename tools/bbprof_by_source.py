#!/usr/bin/env python3
"""Developer tool: sums a `tools/bbprof.py listing` (every executed block of kl_search with its executions per wave iteration) by source
region of bwb_lane.h - the functions and the lettered sections of the search loop, found by their text in the source file.

    tools/bbprof.py listing counts.json bwbble_amd/tools_exp/libbwbble_hip_bbprof.s kl_searchImLb0ELb1 0.02 > listing.txt
    tools/bbprof_by_source.py listing.txt [bwbble_amd/csrc/bwb_lane.h]
"""
import collections, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(sys.argv[1]).read().split("\n")
src = open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_lane.h")).read().split("\n")


def find(pat, start=0):
    for i in range(start or 0, len(src)):
        if re.search(pat, src[i]):
            return i + 1
    return None


ks = find(r"void kl_search")
marks = [("pair_setup", find(r"void pair_setup")), ("wave_gather", find(r"void wave_gather")), ("sub_pops16", find(r"void sub_pops16")),
         ("side_read", find(r"void side_read")), ("side_finish", find(r"void side_finish")), ("kid_get", find(r"void kid_get")),
         ("wave_children", find(r"uint32_t wave_children")), ("list_add", find(r"struct ListW")), ("publish / grab_read", find(r"void publish_done")),
         ("prefetch asm", find(r"void prefetch128")), ("IntvRegs / ListBuf", find(r"struct IntvRegs;")), ("kl_calc_d", find(r"void kl_calc_d")),
         ("LHeap: members (copies of the side registers land here)", find(r"struct LHeap \{")), ("LHeap: side_of / switch_cache", find(r"int side_of\(int pen\)")),
         ("LHeap: alloc", find(r"uint32_t alloc\(bool")), ("LHeap: release_excess", find(r"uint32_t release_excess")), ("LHeap: reserve", find(r"uint32_t reserve\(uint32_t st")),
         ("LHeap: pack / unpack / store", find(r"static __device__ __forceinline__ void pack\(")), ("LHeap: pop", find(r"void pop\(LEntry")),
         ("LHeap: give_back / prefetch", find(r"void give_back")), ("wave_sum5", find(r"uint32_t wave_sum5")), ("search: prologue", ks),
         ("search: loop top (admission, park)", find(r"^\tfor \(;;\) \{", ks)), ("A: pick / pop", find(r"---- A: pick the SA interval")),
         ("B: record, issue", find(r"---- B: one round of memory")), ("B: rank call + record unpack", find(r"KidCtx<P> kc;", find(r"---- B: one round"))),
         ("C: deletion group", find(r"---- C: act on it")), ("C: prune, dispatch, hit, start of a tail", find(r"const int e_i = e\.f & 255")),
         ("C: expansion rules (allow_*)", find(r"---- expansion :377-504")), ("C: child counts, reserve", find(r"push sequence \(:434-504\)")),
         ("C: child templates", find(r"child entry templates")), ("C: gap pushes", find(r"gap pushes: insertion")),
         ("C: mismatch loop", find(r"const uint32_t sm = \(uint32_t\)STATE_M")), ("C: match children", find(r"The last match child is the next entry popped")),
         ("C: commit", find(r"h\.num_entries \+= nGc \+ nX \+ n0")), ("E: exact step", find(r"if \(exact_step && need_rank\)")),
         ("E: end of an exact tail", find(r"if \(exact_done && !ovf && seeding\)")), ("F: end of the iteration", find(r"STAMP\(15\)")), ("after the loop", find(r"#undef myalns"))]
marks = sorted([(n, l) for n, l in marks if l], key=lambda x: x[1])


def region(ln):
    r = "(before the first function)"
    for n, l in marks:
        if ln >= l:
            r = n
        else:
            break
    return r


def cls(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    return "other"


w, agg, tot = 0.0, collections.defaultdict(collections.Counter), collections.Counter()
for l in lines:
    m = re.match(r"### x([\d.]+) per wave iteration", l)
    if m:
        w = float(m.group(1))
        continue
    t = l.strip()
    if not t or t.startswith((".", ";", "#")) or w <= 0:
        continue
    m = re.match(r"(\S+):(\d+)", t.split(";")[-1].strip())
    if not m:
        continue
    f, ln = m.group(1), int(m.group(2))
    r = ("(compiler glue: no source line)" if ln == 0 else region(ln)) if f == "bwb_lane.h" else "other file: " + f
    c = cls(t.split()[0])
    agg[r][c] += w
    tot[c] += w
print("# instructions per wave iteration by source region of bwb_lane.h: vector  scalar  branch  wait/nop  LDS  vector-memory")
print("# total" + " " * 50 + f"{tot['valu']:7.1f} {tot['salu']:7.1f} {tot['branch']:6.1f} {tot['wait']:6.1f} {tot['lds']:6.1f} {tot['vmem']:6.1f}")
for r, c in sorted(agg.items(), key=lambda x: -sum(x[1].values())):
    print(f"{r:57s}{c['valu']:7.1f} {c['salu']:7.1f} {c['branch']:6.1f} {c['wait']:6.1f} {c['lds']:6.1f} {c['vmem']:6.1f}")
