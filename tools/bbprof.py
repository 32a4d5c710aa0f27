#!/usr/bin/env python3
"""Developer tool: a basic-block execution profile of the alignment kernels - "which instructions does a wave really issue per iteration".

The loop of kl_search is bound by instruction issue and ROCm 7.2 ships no thread-trace decoder for gfx950, so
this tool counts itself: it compiles bwb_hip.hip to assembly, gives every basic block of the chosen kernels a counter - one LANE of a
spare VGPR per block, incremented with v_readlane / s_add / v_writelane through M0, i.e. once per WAVE execution of the block whatever the
exec mask - adds the counters to the end of the statistics buffer when a wave leaves (64-bit global atomics), and re-runs the rest of
hipcc's own pipeline (assembler, lld, bundler, host compile) on the patched assembly.

    tools/bbprof.py build [-D<flag> ...]          -> bwbble_amd/tools_exp/libbwbble_hip_bbprof.so + ..._bbprof.map.json (no GPU needed)
    BWB_LIB=.../libbwbble_hip_bbprof.so BWB_BBPROF_OUT=counts.json python3 bench.py ...     (GPU: the library dumps the counters with
                                                                                             every get_stats; the last line counts)
    tools/bbprof.py report counts.json [map.json]  -> dynamic instructions per wave iteration by class, by source line region, hottest blocks
    tools/bbprof.py listing counts.json <instrumented.s> <kernel-substring> [min executions per iteration]
                                                   -> the kernel's assembly, block by block, each block headed by its executions per wave
                                                      iteration (what sessions 9-10 of round 5 worked from; tools/bbprof_by_source.py sums it by source region)

The instrumented kernels need more registers (two waves per SIMD instead of three) and are several times slower: the COUNTS per read are
the product's (same code, same control flow), the timings are not.  A block whose start has M0 or SCC live (the increment clobbers both)
is not counted directly; `report` lists them (none on the hot path of the shipped kernels).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_hip.hip")
OUT_LIB = os.path.join(ROOT, "bwbble_amd", "tools_exp", "libbwbble_hip_bbprof.so")
OUT_MAP = os.path.join(ROOT, "bwbble_amd", "tools_exp", "libbwbble_hip_bbprof.map.json")
DEFAULT_KERNELS = ["_Z9kl_searchImLb0ELb1EE", "_Z9kl_calc_dImE"]  # the instantiations every headline number runs (u64 positions, 16-byte entries, multi-genome)
BBPROF_BYTE_OFF = 104 * 8  # STAT_BBPROF (bwb_kernels.h): the counters follow the ordinary statistics words
MAX_COUNTERS = 2048
TEMP_SREG = "xnack_mask_lo"  # the scalar the increment goes through: no compiler-generated code touches it (XNACK is off on this pool)

INSTR = re.compile(r"^\s+((?:v|s|ds|global|buffer|flat|scratch)_\w+)\s*(.*)$")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
SCC_READ = re.compile(r"^(s_cbranch_scc[01]|s_cselect_b(32|64)|s_addc_u32|s_subb_u32|s_cmov_b(32|64)|s_cmovk_i32)$")
SCC_NOWRITE = re.compile(r"^(s_mov_|s_movk_|s_cmov|s_mul_i32|s_mul_hi|s_mulk_|s_load_|s_buffer_load|s_waitcnt|s_nop|s_brev|s_ff[01]_|s_flbit|s_sext|s_bitset|s_getpc|s_setpc|s_swappc|"
                         r"s_bfm|s_pack|s_cselect|s_cbranch|s_branch|s_barrier|s_sleep|s_setprio|s_endpgm|s_getreg|s_setreg|s_memtime|s_memrealtime|s_dcache|s_sendmsg|s_trap|s_sethalt|"
                         r"s_icache|s_incperflevel|s_decperflevel|s_ttrace|s_code_end|s_set_gpr|s_rfe|s_movrel|s_setvskip|s_scratch|s_store|s_atc|s_wakeup|s_barrier)")
M0_IMPLICIT_READ = re.compile(r"^(global_load_lds_|buffer_load_.*lds|s_movrel|v_movrel|s_sendmsg|v_interp|ds_\w+_addtid|ds_gws|ds_ordered)")


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def compile_steps(flags, d):
    """hipcc --save-temps -v: returns the device assembly path and the sub-commands that follow the device code generation"""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Wno-unused-value",
           "-gline-tables-only", "-DBWB_BBPROF", "--save-temps", "-v", "-o", os.path.join(d, "lib.so"), SRC] + flags
    r = subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    steps = [ln.strip() for ln in r.stderr.splitlines() if ln.startswith(' "')]
    asm = os.path.join(d, "bwb_hip-hip-amdgcn-amd-amdhsa-gfx950.s")
    k = next(i for i, s in enumerate(steps) if "-cc1as" in s and "amdgcn-amd-amdhsa" in s)
    return asm, steps[k:]


class Block:
    def __init__(self, idx, label, first_line):
        self.idx, self.label, self.first_line = idx, label, first_line
        self.ins = []  # (mnemonic, operands, loc)
        self.succ = []
        self.fallthrough = True
        self.counter = None
        self.skip_reason = None


def split_kernel(lines, start, end):
    """basic blocks of lines[start:end] (the kernel's body after its label)"""
    blocks, cur, loc = [], None, None
    cur = Block(0, "entry", start)
    blocks.append(cur)
    new_after_branch = False
    for i in range(start, end):
        ln = lines[i]
        m = LABEL.match(ln)
        if m:
            cur = Block(len(blocks), m.group(1), i + 1)
            blocks.append(cur)
            new_after_branch = False
            continue
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        m = INSTR.match(ln)
        if not m:
            continue
        if new_after_branch:
            cur = Block(len(blocks), None, i)
            blocks.append(cur)
            new_after_branch = False
        op, args = m.group(1), m.group(2).split(";")[0].strip()
        cur.ins.append((op, args, loc))
        if op.startswith("s_cbranch"):
            new_after_branch = True
        if op in ("s_setpc_b64", "s_swappc_b64"):
            raise SystemExit("bbprof: indirect branch / call in the kernel: not supported")
    by_label = {b.label: b for b in blocks if b.label}
    for k, b in enumerate(blocks):
        b.fallthrough = True
        for op, args, _ in b.ins:
            if op.startswith("s_cbranch") or op == "s_branch":
                t = re.search(r"(\.LBB\d+_\d+)", args)
                if t and t.group(1) in by_label:
                    b.succ.append(by_label[t.group(1)].idx)
            if op in ("s_branch", "s_endpgm"):
                b.fallthrough = False
        if b.fallthrough and k + 1 < len(blocks):
            b.succ.append(k + 1)
    return blocks


def first_operand(args):
    return args.split(",")[0].strip() if args else ""


def liveness(blocks):
    """live-in of M0 and SCC per block (backward dataflow)"""
    use = {"m0": [False] * len(blocks), "scc": [False] * len(blocks)}
    dfn = {"m0": [False] * len(blocks), "scc": [False] * len(blocks)}
    for b in blocks:
        seen_def = {"m0": False, "scc": False}
        for op, args, _ in b.ins:
            dst = first_operand(args)
            srcs = args[len(dst):]
            # M0
            reads_m0 = bool(M0_IMPLICIT_READ.match(op)) or bool(re.search(r"\bm0\b", srcs))
            writes_m0 = dst == "m0" and not op.startswith(("s_cmp", "s_bitcmp", "v_cmp"))
            if op.startswith(("s_cmp", "s_bitcmp")) and re.search(r"\bm0\b", args):
                reads_m0 = True
            if reads_m0 and not seen_def["m0"]:
                use["m0"][b.idx] = True
            if writes_m0:
                seen_def["m0"] = True
            # SCC
            if SCC_READ.match(op) and not seen_def["scc"]:
                use["scc"][b.idx] = True
            if op.startswith("s_") and not SCC_NOWRITE.match(op):
                seen_def["scc"] = True
        for r in ("m0", "scc"):
            dfn[r][b.idx] = seen_def[r]
    live_in = {r: list(use[r]) for r in use}
    changed = True
    while changed:
        changed = False
        for b in reversed(blocks):
            for r in ("m0", "scc"):
                out = any(live_in[r][s] for s in b.succ)
                v = use[r][b.idx] or (out and not dfn[r][b.idx])
                if v != live_in[r][b.idx]:
                    live_in[r][b.idx] = v
                    changed = True
    return live_in


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait_nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


def instrument(asm_path, kernels):
    lines = open(asm_path).read().split("\n")
    file_names = {}
    for ln in lines:
        m = re.match(r'\s+\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
        if m:
            file_names[int(m.group(1))] = m.group(2)
    out_map = {"kernels": {}, "files": file_names}
    inserts = collections.defaultdict(list)  # line index -> text inserted BEFORE that line
    replace = {}
    counter_base = 0
    for kern_sub in kernels:
        ks = [k for k, ln in enumerate(lines) if re.match(r"^(_Z\w+):", ln) and kern_sub in ln]
        if not ks:
            raise SystemExit("bbprof: no kernel matches " + kern_sub)
        start = ks[0]
        name = re.match(r"^(_Z\w+):", lines[start]).group(1)
        end = next(k for k in range(start, len(lines)) if lines[k].startswith(".Lfunc_end"))
        blocks = split_kernel(lines, start + 1, end)
        live = liveness(blocks)
        # kernel descriptor
        d0 = next(k for k, ln in enumerate(lines) if ln.strip() == ".amdhsa_kernel " + name)
        d1 = next(k for k in range(d0, len(lines)) if lines[k].strip() == ".end_amdhsa_kernel")
        nfv = next(k for k in range(d0, d1) if ".amdhsa_next_free_vgpr" in lines[k])
        acc = next(k for k in range(d0, d1) if ".amdhsa_accum_offset" in lines[k])
        vbase = int(lines[nfv].split()[-1])
        vbase = (vbase + 3) // 4 * 4
        n = 0
        for b in blocks:
            if not b.ins:
                b.skip_reason = "empty"
                continue
            if live["scc"][b.idx]:
                b.skip_reason = "scc live"
                continue
            b.counter = n
            n += 1
        nreg = (n + 63) // 64
        if counter_base + nreg * 64 > MAX_COUNTERS:
            raise SystemExit("bbprof: too many blocks")
        vptr = vbase + nreg  # lanes 0 and 1 of this register keep the `stats` pointer (the kernel may overwrite its kernarg pointer)
        total = vbase + nreg + 1
        total = (total + 7) // 8 * 8
        replace[nfv] = re.sub(r"\d+\s*$", str(total), lines[nfv])
        for k in range(d0, d1):
            if ".amdhsa_reserve_xnack_mask" in lines[k]:
                replace[k] = re.sub(r"\d+\s*$", "1", lines[k])
        replace[acc] = re.sub(r"\d+\s*$", str((total + 3) // 4 * 4), lines[acc])
        # metadata vgpr_count
        mi = next(k for k, ln in enumerate(lines) if re.match(r"\s+\.name:\s+" + re.escape(name) + r"\s*$", ln))
        for k in range(mi, min(mi + 40, len(lines))):
            if ".vgpr_count:" in lines[k]:
                replace[k] = re.sub(r"\d+\s*$", str(total), lines[k])
                break
        # counter init at entry (exec is full there), increments, dump at every s_endpgm
        init = "".join("\tv_mov_b32_e32 v%d, 0\n" % (vbase + r) for r in range(nreg))
        init += "\ts_load_dwordx2 s[4:5], s[0:1], 0x%x\n\ts_waitcnt lgkmcnt(0)\n\tv_writelane_b32 v%d, s4, 0\n\tv_writelane_b32 v%d, s5, 1\n" % (0, vptr, vptr)
        first_ins = next(k for k in range(start + 1, end) if INSTR.match(lines[k]))
        inserts[first_ins].append(init)
        init_at = len(inserts[first_ins]) - 1
        for b in blocks:
            if b.counter is None:
                continue
            reg, ln_ = vbase + b.counter // 64, b.counter % 64
            seq = "\tv_readlane_b32 {t}, v%d, %d\n\ts_nop 3\n\ts_add_u32 {t}, {t}, 1\n\ts_nop 3\n\tv_writelane_b32 v%d, {t}, %d\n\ts_nop 1\n".format(t=TEMP_SREG) % (reg, ln_, reg, ln_)
            k = b.first_line
            while not INSTR.match(lines[k]):
                k += 1
            inserts[k].append(seq)
        # the `stats` pointer is the LAST explicit argument of both kernels: find its offset (the last .offset before the hidden arguments)
        mi0 = mi
        while not re.match(r"\s+- \.a", lines[mi0]):
            mi0 -= 1
        args = []
        k = mi0
        cur = {}
        while k < mi:
            m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)", lines[k])
            if m:
                if lines[k].lstrip().startswith("- .") and cur:
                    args.append(cur)
                    cur = {}
                cur[m.group(1)] = m.group(2)
            k += 1
        if cur:
            args.append(cur)
        explicit = [a for a in args if "offset" in a and not a.get("value_kind", "").startswith("hidden")]
        stats_off = int(explicit[-1]["offset"])
        init = init.replace("s[0:1], 0x0", "s[0:1], 0x%x" % stats_off)
        inserts[first_ins][init_at] = init
        dump = "\ts_mov_b64 exec, -1\n\tv_readlane_b32 s2, v%d, 0\n\tv_readlane_b32 s3, v%d, 1\n\ts_nop 3\n" % (vptr, vptr)
        dump += "\ts_add_u32 s2, s2, %d\n\ts_addc_u32 s3, s3, 0\n" % (BBPROF_BYTE_OFF + counter_base * 8)
        dump += "\tv_mbcnt_lo_u32_b32 v2, -1, 0\n\tv_mbcnt_hi_u32_b32 v2, -1, v2\n\tv_lshlrev_b32_e32 v2, 3, v2\n\tv_mov_b32_e32 v1, 0\n"
        for r in range(nreg):
            dump += "\tv_mov_b32_e32 v0, v%d\n\ts_nop 1\n\tglobal_atomic_add_x2 v2, v[0:1], s[2:3] offset:%d\n" % (vbase + r, 0)
            dump += "\ts_add_u32 s2, s2, 512\n\ts_addc_u32 s3, s3, 0\n\ts_waitcnt vmcnt(0)\n"
        for k in range(start + 1, end):
            m = INSTR.match(lines[k])
            if m and m.group(1) == "s_endpgm":
                inserts[k].append(dump)
        info = []
        for b in blocks:
            cls = collections.Counter(classify(op) for op, _, _ in b.ins)
            locs = collections.Counter(l for _, _, l in b.ins if l)
            ops = collections.Counter(op for op, _, _ in b.ins)
            info.append({"idx": b.idx, "label": b.label, "counter": None if b.counter is None else counter_base + b.counter, "skip": b.skip_reason,
                         "succ": b.succ, "classes": dict(cls), "locs": [[f, l, c] for (f, l), c in sorted(locs.items())],
                         "movs": ops.get("v_mov_b32_e32", 0) + ops.get("v_mov_b32_e64", 0) + ops.get("v_mov_b64_e32", 0), "readlane": ops.get("v_readlane_b32", 0) + ops.get("v_writelane_b32", 0),
                         "cndmask": sum(c for o, c in ops.items() if o.startswith("v_cndmask")), "cmp": sum(c for o, c in ops.items() if o.startswith("v_cmp"))})
        out_map["kernels"][name] = {"counter_base": counter_base, "counters": n, "vgpr_base": vbase, "vgprs": total, "blocks": info}
        skipped = [(b.idx, b.label, b.skip_reason, len(b.ins)) for b in blocks if b.counter is None and b.ins]
        print("bbprof: %s: %d blocks, %d counted, %d not counted %s, VGPRs %d -> %d" % (name, len(blocks), n, len(skipped), skipped[:12], vbase, total))
        counter_base += nreg * 64
    out = []
    for k, ln in enumerate(lines):
        for t in inserts.get(k, []):
            out.append(t.rstrip("\n"))
        out.append(replace.get(k, ln))
    open(asm_path, "w").write("\n".join(out))
    return out_map


def build(flags, kernels):
    d = tempfile.mkdtemp(prefix="bbprof_")
    asm, steps = compile_steps(flags, d)
    # The in-place prefetches of kl_search / kl_calc_d rely on a property of the GENERATED code (tools/check_prefetch_regs.py); this build does
    # not go through the Makefile's build_checked rule, so the proof runs here, on the assembly BEFORE it is instrumented (the counters added
    # below live in VGPRs above the kernel's own and never touch a prefetched register).  A build that fails it is not written.
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import check_prefetch_regs as cpr
    res, _ = cpr.check_asm(asm)
    bad = ["%s: %s" % (k, "; ".join(errs) if errs else "no prefetch site found") for k, (n, errs) in res.items() if errs or n == 0]
    if bad or not res:
        raise SystemExit("bbprof: this build fails the in-flight prefetch register proof, no library written:\n  " + "\n  ".join(bad or ["no kernel found"]))
    m = instrument(asm, kernels)
    for s in steps:
        subprocess.run(s, cwd=d, shell=True, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    os.makedirs(os.path.dirname(OUT_LIB), exist_ok=True)
    run(["cp", os.path.join(d, "lib.so"), OUT_LIB])
    run(["cp", asm, OUT_LIB[:-3] + ".s"])  # (for `listing`)
    json.dump(m, open(OUT_MAP, "w"))
    print("bbprof: wrote", OUT_LIB, "and", OUT_MAP)


def load_regions():
    """regions of bwb_lane.h by function / by the section comments of kl_search's loop"""
    src = open(os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_lane.h")).read().split("\n")
    marks = []
    pats = [(r"^__device__ __forceinline__ void pair_setup", "pair_setup"), (r"^__device__ __forceinline__ void wave_gather", "wave_gather"),
            (r"^__device__ __forceinline__ void sub_pops16", "rank: sub_pops16"), (r"^__device__ __forceinline__ void side_read", "rank: side_read"),
            (r"^__device__ __forceinline__ void side_finish", "rank: side_finish"), (r"kid_get\(const KidCtx", "kid_get"),
            (r"^__device__ __forceinline__ uint32_t wave_children", "wave_children"), (r"^template <typename P> struct ListW", "list_add"),
            (r"^__device__ __forceinline__ void publish_done", "publish/grab"), (r"void kl_calc_d\(", "kl_calc_d"), (r"^__device__ __forceinline__ void prefetch128", "prefetch asm"),
            (r"^template <typename P, bool WIDE> struct LHeap", "LHeap: decl/side regs"), (r"void switch_cache\(int s\)", "LHeap: switch_cache"), (r"uint32_t alloc\(bool &ovf", "LHeap: alloc"),
            (r"uint32_t release_excess\(", "LHeap: release_excess"), (r"uint32_t reserve\(uint32_t st", "LHeap: reserve"), (r"static __device__ __forceinline__ void pack\(", "LHeap: pack/unpack/store"),
            (r"void pop\(LEntry<P> &e", "LHeap: pop"), (r"void give_back\(uint32_t c\)", "LHeap: give_back/prefetch"), (r"uint32_t wave_sum5", "misc"),
            (r"void kl_search\(", "search: prologue"), (r"^\tfor \(;;\) \{\s*$", None), (r"auto add_aln = ", "add_aln"), (r"---- A: pick the SA interval", "A pick/pop"),
            (r"---- B: one round of memory", "B record/issue"), (r"KidCtx<P> kc;\s*$", "rank call + record unpack"), (r"---- C: act on it", "C: group"),
            (r"\} else if \(from_pop\) \{", "C: prune/hit/exact start"), (r"---- expansion :377-504", "C: expansion logic"), (r"const uint32_t cst_old = h.cst;\s*$", "C: reserve"),
            (r"/\* child entry templates \*/", "C: templates"), (r"\{ /\* gap pushes", "C: gap pushes"), (r"const uint32_t sm = \(uint32_t\)STATE_M", "C: mismatch/match loops"),
            (r"h.num_entries \+= nGc \+ nX \+ n0;", "C: commit"), (r"if \(exact_step && need_rank\) \{", "E: exact step"), (r"if \(exact_done && !ovf && seeding\)", "E: exact done (seed)"),
            (r"\} else if \(exact_done && !ovf\) \{", "E: exact done (hits)"), (r"^\t\tSTAMP\(15\);", "F: top reload / stats"), (r"^\t\tif \(finish\) \{", "F: finish read"),
            (r"if \(lane == 0 && n_bkt\) atomicAdd\(&R_stats\[STAT_BKT_SEARCH\]", "search: epilogue"), (r"void k_rank_bench_lane", "rank bench")]
    loop_seen = 0
    for i, ln in enumerate(src, 1):
        for p, name in pats:
            if re.search(p, ln):
                if name is None:
                    loop_seen += 1
                    if loop_seen == 2:
                        marks.append((i, "search: admission/park/top"))
                elif name == "C: reserve" and any(n == "C: reserve" for _, n in marks):
                    pass
                else:
                    marks.append((i, name))
    marks.sort()
    return marks


def region_of(marks, line):
    name = "(before)"
    for l, n in marks:
        if l <= line:
            name = n
        else:
            break
    return name


def report(counts_path, map_path):
    m = json.load(open(map_path))
    last = None
    for ln in open(counts_path):
        ln = ln.strip()
        if ln:
            last = json.loads(ln)
    counts = last["counters"]
    lane_file = next((int(k) for k, v in m["files"].items() if v.endswith("bwb_lane.h")), None)
    marks = load_regions()
    for name, K in m["kernels"].items():
        blocks = K["blocks"]
        entry_counter = next(b["counter"] for b in blocks if b["counter"] is not None)
        waves = counts[entry_counter] or max(counts[b["counter"]] for b in blocks if b["counter"] is not None and not b["succ"]) if any(counts[b["counter"]] for b in blocks if b["counter"] is not None) else 0
        if not waves:
            continue
        # wave iterations = executions of the loop header: the block with the largest count whose label exists
        cnt = {b["idx"]: (counts[b["counter"]] if b["counter"] is not None else None) for b in blocks}
        # blocks that could not be counted: estimate from the predecessors that have them as only successor, else from their successor
        preds = collections.defaultdict(list)
        for b in blocks:
            for s in b["succ"]:
                preds[s].append(b["idx"])
        for b in blocks:
            if cnt[b["idx"]] is None and b["classes"]:
                ps = preds[b["idx"]]
                if ps and all(cnt[p] is not None and len(blocks[p]["succ"]) == 1 for p in ps):
                    cnt[b["idx"]] = sum(cnt[p] for p in ps)
                elif len(b["succ"]) == 1 and cnt.get(b["succ"][0]) is not None and len(preds[b["succ"][0]]) == 1:
                    cnt[b["idx"]] = cnt[b["succ"][0]]
        iters = last.get("wave_iterations_search" if "kl_search" in name else "wave_iterations_calc_d") or max(v for v in cnt.values() if v)
        print("\n== %s: %d waves, %d wave iterations (loop header executions), counted blocks %d/%d" % (name, waves, iters, sum(1 for b in blocks if b["counter"] is not None), sum(1 for b in blocks if b["classes"])))
        tot = collections.Counter()
        by_region = collections.defaultdict(collections.Counter)
        extra = collections.Counter()
        hot = []
        for b in blocks:
            c = cnt[b["idx"]]
            if not c:
                if c is None and b["classes"]:
                    print("   (not counted: block %d %s %s, %d instructions)" % (b["idx"], b["label"], b["skip"], sum(b["classes"].values())))
                continue
            n_ins = sum(b["classes"].values())
            for k, v in b["classes"].items():
                tot[k] += v * c
            for k in ("movs", "readlane", "cndmask", "cmp"):
                extra[k] += b[k] * c
            nloc = sum(x[2] for x in b["locs"]) or 1
            unatt = n_ins - sum(x[2] for x in b["locs"])
            for f, l, k in b["locs"]:
                reg = region_of(marks, l) if f == lane_file else ("other file: " + os.path.basename(m["files"].get(str(f), str(f))) if l else "(line 0)")
                if l == 0:
                    reg = "(line 0: compiler glue)"
                # classes are per block, not per line: spread them in proportion
                for cl, v in b["classes"].items():
                    by_region[reg][cl] += v * c * k / nloc * (1.0 - unatt / n_ins if n_ins else 1.0)
            if unatt:
                for cl, v in b["classes"].items():
                    by_region["(no line)"][cl] += v * c * unatt / n_ins
            hot.append((c * n_ins, c, b))
        print("   per wave iteration: " + "  ".join("%s %.1f" % (k, tot[k] / iters) for k in ("valu", "salu", "branch", "lds", "vmem", "smem", "wait_nop")))
        print("   of the VALU: v_mov %.1f  v_readlane/writelane %.1f  v_cndmask %.1f  v_cmp %.1f" % tuple(extra[k] / iters for k in ("movs", "readlane", "cndmask", "cmp")))
        print("   by source region (instructions per wave iteration: valu / salu+branch / lds / vmem):")
        rows = sorted(by_region.items(), key=lambda kv: -(kv[1]["valu"] + kv[1]["salu"] + kv[1]["branch"]))
        for reg, c in rows:
            v, s, l, vm = c["valu"] / iters, (c["salu"] + c["branch"]) / iters, c["lds"] / iters, c["vmem"] / iters
            if v + s >= 0.5:
                print("     %-34s %7.1f %7.1f %6.1f %6.1f" % (reg, v, s, l, vm))
        hot.sort(key=lambda t: -t[0])
        print("   hottest blocks (executions per wave iteration x instructions):")
        for w, c, b in hot[:40]:
            top = sorted(b["locs"], key=lambda x: -x[2])[:3]
            print("     blk %4d %-12s x%6.3f  valu %3d salu %3d lds %2d vmem %2d  lines %s" % (b["idx"], b["label"] or "(fallthrough)", c / iters, b["classes"].get("valu", 0),
                  b["classes"].get("salu", 0) + b["classes"].get("branch", 0), b["classes"].get("lds", 0), b["classes"].get("vmem", 0), ["%d:%d" % (l, k) for f, l, k in top]))


def listing(counts_path, asm_path, map_path, kern_sub, min_ratio):
    """the instrumented assembly of one kernel, every counted block headed by its executions per wave iteration (gcov style)"""
    m = json.load(open(map_path))
    last = None
    for ln in open(counts_path):
        if ln.strip():
            last = json.loads(ln)
    counts = last["counters"]
    name = next(k for k in m["kernels"] if kern_sub in k)
    K = m["kernels"][name]
    iters = (last.get("wave_iterations_search") if "kl_search" in name else 0) or max(counts[b["counter"]] for b in K["blocks"] if b["counter"] is not None)
    vbase, cbase = K["vgpr_base"], K["counter_base"]
    lines = open(asm_path).read().split("\n")
    start = next(k for k, ln in enumerate(lines) if ln.startswith(name + ":"))
    end = next(k for k in range(start, len(lines)) if lines[k].startswith(".Lfunc_end"))
    out, show, loc, skip = [], True, "", 0
    k = start
    while k < end:
        ln = lines[k]
        mm = re.match(r"\s+v_readlane_b32 %s, v(\d+), (\d+)" % TEMP_SREG, ln)
        if mm and int(mm.group(1)) >= vbase:
            c = counts[cbase + (int(mm.group(1)) - vbase) * 64 + int(mm.group(2))]
            show = c / iters >= min_ratio
            if show:
                out.append("### x%.3f per wave iteration (%d)" % (c / iters, c))
            k += 6
            continue
        mm = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if mm:
            loc = "%s:%s" % (os.path.basename(m["files"].get(mm.group(1), mm.group(1))), mm.group(2))
        elif show and (INSTR.match(ln) or LABEL.match(ln)):
            out.append("%-90s ; %s" % (ln.split(";")[0].rstrip()[:90], loc) if INSTR.match(ln) else ln.split(";")[0])
        k += 1
    print("\n".join(out))


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "listing":  # listing counts.json instrumented.s <kernel-substring> [min executions per iteration]
        listing(sys.argv[2], sys.argv[3], OUT_MAP, sys.argv[4], float(sys.argv[5]) if len(sys.argv) > 5 else 0.05)
        return 0
    if len(sys.argv) < 2 or sys.argv[1] not in ("build", "report"):
        print(__doc__)
        return 2
    if sys.argv[1] == "build":
        flags = [a for a in sys.argv[2:] if a.startswith("-D")]
        kernels = [a for a in sys.argv[2:] if not a.startswith("-")] or DEFAULT_KERNELS
        build(flags, kernels)
    else:
        report(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else OUT_MAP)
    return 0


if __name__ == "__main__":
    sys.exit(main())
