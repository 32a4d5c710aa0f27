#!/usr/bin/env python3
"""Developer tool + CPU test helper: verifies on the ISA that the registers written by the in-place prefetches of kl_search and kl_calc_d
(prefetch128 / 64 / 32 in bwb_lane.h: global loads issued ahead of the gather by an asm that narrows the exec mask) are not touched
by any instruction until a vmcnt(0) wait has been executed.

The compiler does not know that those registers are in flight: a copy, a spill or a use it placed between the load and the wait would
read bytes that have not arrived.  The source is written so that there is none (the loaded values are first looked at after the
gather's wait); this script proves it for the code hipcc actually generated, for every kl_search instantiation:

  * every prefetch site  s_and_saveexec_b64 / global_load_dword[x4] vD, ... / s_mov_b64 exec  is found;
  * from there the code is followed along fall-through and branch edges (both ways of a conditional branch) until an
    `s_waitcnt` with vmcnt(0) is met on the path; any instruction on the way that names a register of vD (as a source or as a
    destination) is an error - except the other prefetch sites' own loads into their own registers.

usage: check_prefetch_regs.py --asm <file.s> [--remarks <file>]   the assembly (and the compiler's remarks) that bwbble_amd/Makefile kept from the
                                                                  hipcc invocation that produced the shipped .so: what `make` runs, and fails on
       check_prefetch_regs.py [-D<flag> ...]                      compile here with extra flags (experiment builds)
exit code 0 = clean; prints one line per kernel"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bwbble_amd", "csrc", "bwb_hip.hip")


def compile_isa(flags, remarks=None):
    """compiles the library with --save-temps; returns the path of the gfx950 assembly.  remarks: a dict that receives, per kernel symbol,
    the compiler's resource-usage remarks (VGPRs, ScratchSize, SGPRs Spill, VGPRs Spill, Occupancy)"""
    d = tempfile.mkdtemp(prefix="isa_")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function", "-Wno-unused-value",
           "--save-temps", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(d, "lib.so"), SRC] + flags
    r = subprocess.run(cmd, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    if remarks is not None:
        parse_remarks(r.stderr, remarks)
    return os.path.join(d, "bwb_hip-hip-amdgcn-amd-amdhsa-gfx950.s")


def parse_remarks(text, remarks):
    cur = None
    for ln in text.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = m.group(1)
            remarks[cur] = {}
            continue
        m = re.search(r"\s{2,}([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", ln)
        if m and cur:
            remarks[cur][m.group(1).strip()] = int(m.group(2))


def check_all(flags=()):
    """-> ({kernel: (sites, [errors])}, {kernel: resource remarks}) for every kl_search instantiation, compiled here"""
    remarks = {}
    path = compile_isa(list(flags), remarks)
    return check_asm(path, remarks)


def check_asm(path, remarks=None, remarks_file=None):
    """the same for an assembly file that a build kept (bwbble_amd/build/<lib>/: the .s of the very hipcc invocation that made the .so)"""
    remarks = {} if remarks is None else remarks
    if remarks_file and os.path.exists(remarks_file):
        parse_remarks(open(remarks_file).read(), remarks)
    cur, kernels = None, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w*(?:kl_search|kl_calc_d)\w*):", ln)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur:
            if ln.startswith(".Lfunc_end"):
                cur = None
                continue
            kernels[cur].append(ln.strip())
    for k, body in kernels.items():  # scratch instructions actually present (the frame may keep a few bytes no instruction uses)
        remarks.setdefault(k, {})["ScratchOps"] = sum(1 for t in body if t.startswith("scratch_"))
        # wave reductions the compiler put in place of atomics with a provably uniform address (a scalar loop over the lanes each): fine in the
        # rare paths, 21 % of the kernel's time when one landed in the loop (profiles/r4_ab_steps.txt, session 10)
        remarks[k]["ComputeLoops"] = sum(1 for t in body if "%ComputeLoop" in t)
    return {k: check_kernel(k, body) for k, body in kernels.items()}, remarks


def vregs(text):
    """all VGPR numbers an operand string names: v12, v[38:41]"""
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def is_vm0_wait(ins):
    if not ins.startswith("s_waitcnt"):
        return False
    m = re.search(r"vmcnt\((\d+)\)", ins)
    if m:
        return int(m.group(1)) == 0
    m = re.match(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins)  # raw immediate: vmcnt = bits 3:0 and 15:14
    if m:
        v = int(m.group(1), 0)
        return (v & 0xF) == 0 and ((v >> 14) & 3) == 0
    return False


def check_kernel(name, body):
    """body: list of stripped lines (labels and instructions) of one kernel"""
    ins, labels = [], {}
    for ln in body:
        if not ln or ln.startswith(";") or ln.startswith(".loc") or ln.startswith(".Ltmp") or ln.startswith(".cfi") or ln.startswith(".p2align"):
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if ln.startswith("."):
            continue
        ins.append(ln.split(";")[0].strip())
    sites = [i for i in range(len(ins) - 2) if ins[i].startswith("s_and_saveexec_b64") and ins[i + 1].startswith("global_load_dword") and ins[i + 2].startswith("s_mov_b64 exec")]
    site_loads = {i + 1 for i in sites}
    errors = []
    for i in sites:
        dst = vregs(ins[i + 1].split(",")[0])
        seen, work = set(), [i + 3]
        while work:
            pc = work.pop()
            while pc < len(ins) and pc not in seen:
                seen.add(pc)
                t = ins[pc]
                if is_vm0_wait(t):
                    break
                if pc not in site_loads or vregs(t.split(",")[0]) & dst:
                    hit = vregs(t) & dst
                    if hit and not (pc in site_loads and not (vregs(t.split(",")[0]) & dst)):
                        errors.append(f"{name}: `{t}` touches v{sorted(hit)} of the prefetch `{ins[i + 1]}` before a vmcnt(0) wait")
                        break
                m = re.match(r"^(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)", t)
                if m:
                    if m.group(2) in labels:
                        work.append(labels[m.group(2)])
                    if m.group(1) == "s_branch":
                        break
                if t.startswith("s_endpgm"):
                    break
                pc += 1
    return len(sites), errors


def main():
    flags = [a for a in sys.argv[1:] if a.startswith("-D")]
    if "--asm" in sys.argv:
        asm = sys.argv[sys.argv.index("--asm") + 1]
        rem = sys.argv[sys.argv.index("--remarks") + 1] if "--remarks" in sys.argv else None
        res, remarks = check_asm(asm, remarks_file=rem)
    else:
        res, remarks = check_all(flags)
    bad = 0
    for k, (n, errs) in res.items():
        ru = remarks.get(k, {})
        print(f"{k}: {n} prefetch site(s), {'OK' if not errs else 'HAZARD'} | VGPRs {ru.get('VGPRs')} scratch {ru.get('ScratchSize')} B/lane ({ru.get('ScratchOps')} scratch instructions), spills: {ru.get('SGPRs Spill')} SGPR, {ru.get('VGPRs Spill')} VGPR, {ru.get('Occupancy')} waves/SIMD")
        for e in errs:
            print("   " + e)
        bad += len(errs)
        if n == 0:
            print("   no prefetch site found: the check is vacuous")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
