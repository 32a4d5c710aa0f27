#!/usr/bin/env python3
"""Developer tool: `bwbble align` end to end on the files bench.py left in its work directory, with the .aln checked in TWO places:
the first N records against the real reference (oracle/_ref/bwbble, when present) and N records from the middle of the file (a later
chunk: its parked reads finish inside the next chunk's slice while the host already fetches results) against the library's one-batch
interface, which reads back only after every kernel has ended.  usage: cli_check.py <genome.fa> <reads.fq> [N=5000] [align flags]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bwbble_amd as bw

fa, fq = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
flags = sys.argv[4:] or ["-n", "3"]
out = fq + ".cli.aln"
t = time.time()
log = subprocess.run([bw.HOST_BIN, "align"] + flags + [fa, fq, out], check=True, stdout=subprocess.PIPE, text=True).stdout
print([l for l in log.splitlines() if l.startswith("GPUs:")][0], f"| process wall {time.time() - t:.1f} s")
print("\n".join(l for l in log.splitlines() if l.startswith("start-up:")))


def records(path, first, count):
    """byte ranges of records [first, first + count) of a .aln file (align.c:345-382)"""
    data = np.memmap(path, dtype=np.uint8, mode="r")
    pos, r, start = 0, 0, None
    i32 = lambda p: int.from_bytes(bytes(data[p:p + 4]), "little", signed=True)
    while r < first + count and pos < len(data):
        if r == first:
            start = pos
        n = i32(pos); pos += 4
        for _ in range(n):
            pairs = i32(pos + 36); pos += 40 + 4 * pairs
        r += 1
    return bytes(data[start:pos])


seqs, lens = bw.load_fastq_codes(fq)
mid = (len(lens) // 2 // 1000) * 1000 + 123
ctx = bw.Context(fa + ".bwt")
off, alns = ctx.align(bw.params(flags), seqs[mid:mid + N], lens[mid:mid + N])
print(f"records [{mid}, {mid + N}) of the CLI's .aln identical to the one-batch interface on the same reads:", records(out, mid, N) == bw.aln_bytes(off, alns))
ctx.close()
ref = os.path.join(ROOT, "oracle", "_ref", "bwbble")
if os.path.exists(ref):
    head = fq + ".head"
    with open(fq) as f, open(head, "w") as g:
        for i, line in enumerate(f):
            if i >= 4 * N:
                break
            g.write(line)
    subprocess.run([ref, "align"] + flags + ["-t", str(os.cpu_count()), fa, head, head + ".aln"], check=True, stdout=subprocess.DEVNULL)
    print(f"first {N} records identical to oracle/_ref/bwbble align on the same reads:", records(out, 0, N) == open(head + ".aln", "rb").read())
