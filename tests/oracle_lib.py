"""ctypes binding of oracle/libbwb_oracle.so (TEST INFRASTRUCTURE: the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "libbwb_oracle.so")


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "max_diff", "max_gapo", "max_gape", "max_entries", "mm_score", "gapo_score", "gape_score",
        "seed_length", "max_diff_seed", "max_best", "no_indel_length", "matched_Ncontig", "use_precalc",
        "is_multiref", "n_threads")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "visits_single", "visits_alphabet", "heap_pops", "heap_pushes", "max_heap_entries", "n_alignments")]


class Index(C.Structure):
    _fields_ = [("length", C.c_uint64), ("num_words", C.c_uint64), ("num_sa", C.c_uint64), ("num_occ", C.c_uint64),
                ("sa0_index", C.c_uint64), ("C", C.c_uint64 * 17), ("bwt", C.POINTER(C.c_uint32)),
                ("O", C.POINTER(C.c_uint64)), ("SA", C.POINTER(C.c_uint64))]


class Dlb(C.Structure):
    _fields_ = [("num_diff", C.c_int), ("sa_intv_width", C.c_int)]


_FLAG = {"-M": "mm_score", "-O": "gapo_score", "-E": "gape_score", "-n": "max_diff", "-k": "max_diff_seed",
         "-o": "max_gapo", "-e": "max_gape", "-l": "seed_length", "-m": "max_entries", "-t": "n_threads"}


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.bwb_or_load_bwt.restype = C.POINTER(Index)
        L.bwb_or_load_bwt.argtypes = [C.c_char_p, C.c_int]
        L.bwb_or_free_index.argtypes = [C.POINTER(Index)]
        L.bwb_or_O.restype = C.c_uint64
        L.bwb_or_O.argtypes = [C.POINTER(Index), C.c_int, C.c_uint64]
        L.bwb_or_O_alphabet_many.argtypes = [C.POINTER(Index), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.bwb_or_calculate_d.argtypes = [C.POINTER(Index), C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Params), C.c_void_p]
        L.bwb_or_align_fastq.restype = C.c_long
        L.bwb_or_align_fastq.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(Params), C.c_int, C.c_long,
                                         C.POINTER(Stats), C.POINTER(C.c_double)]
        L.bwb_or_align_encoded.restype = C.c_long
        L.bwb_or_align_encoded.argtypes = [C.POINTER(Index), C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.POINTER(Params),
                                           C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats),
                                           C.POINTER(C.c_double)]
        L.bwb_or_free.argtypes = [C.c_void_p]
        L.bwb_or_SA.restype = C.c_uint64
        L.bwb_or_SA.argtypes = [C.POINTER(Index), C.c_uint64]

    def params(self, flags=()):
        p = Params()
        self.lib.bwb_or_default_params(C.byref(p))
        flags = list(flags)
        i = 0
        while i < len(flags):
            if flags[i] == "-S":
                p.is_multiref = 0
                i += 1
                continue
            if flags[i] == "-P":
                p.use_precalc = 1
                i += 1
                continue
            setattr(p, _FLAG[flags[i]], int(flags[i + 1]))
            i += 2
        return p

    def load_index(self, bwt_path, load_sa=False):
        idx = self.lib.bwb_or_load_bwt(bwt_path.encode(), int(load_sa))
        if not idx:
            raise IOError("oracle could not load " + bwt_path)
        return idx

    def O_alphabet(self, idx, pos, inc):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((len(pos), 16), dtype=np.uint64)
        self.lib.bwb_or_O_alphabet_many(idx, pos.ctypes.data, len(pos), inc, out.ctypes.data)
        return out

    def O_single(self, idx, pos):
        out = np.zeros((len(pos), 16), dtype=np.uint64)
        for q, p in enumerate(pos):
            for c in range(1, 16):
                out[q, c] = self.lib.bwb_or_O(idx, c, int(p))
        return out

    def calculate_d(self, idx, seq, p):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        D = np.zeros((len(seq) + 1, 2), dtype=np.int32)
        self.lib.bwb_or_calculate_d(idx, seq.ctypes.data, len(seq), D.ctypes.data, C.byref(p), None)
        return D

    def align_fastq(self, bwt_path, fastq, aln_out, p, fresh_dseed=0, max_reads=0):
        st, sec = Stats(), C.c_double()
        n = self.lib.bwb_or_align_fastq(bwt_path.encode(), fastq.encode(), aln_out.encode(), C.byref(p), fresh_dseed,
                                         max_reads, C.byref(st), C.byref(sec))
        if n < 0:
            raise RuntimeError(f"oracle align failed ({n})")
        return n, st, sec.value

    def align_encoded(self, idx, seqs, lens, p, fresh_dseed=0):
        """seqs: (n, stride) uint8 codes A0 G1 C2 T3 N4; returns (.aln bytes, Stats, seconds).
        fresh_dseed=0 with p.n_threads <= 1 is the serial reference (one D_seed buffer for the whole file, inexact_match.c:35):
        a read not longer than the seed sees the bounds of the last longer read before it - what the GPU path reproduces.
        With threads, every thread has its own buffer over its own share of a 262 144-read batch (:115-121)."""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        buf, blen, st, sec = C.c_void_p(), C.c_size_t(), Stats(), C.c_double()
        self.lib.bwb_or_align_encoded(idx, seqs.ctypes.data, lens.ctypes.data, seqs.shape[1], len(lens), C.byref(p),
                                      fresh_dseed, C.byref(buf), C.byref(blen), C.byref(st), C.byref(sec))
        data = C.string_at(buf, blen.value)
        self.lib.bwb_or_free(buf)
        return data, st, sec.value


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ROOT, "oracle", "bwb_oracle.c")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), LIB], check=True)
    return Oracle(C.CDLL(LIB))


def parse_aln(data):
    """Decode .aln bytes (align.c:345-382) -> list (per read) of lists of dicts."""
    out, off = [], 0
    mv = memoryview(data)
    while off < len(data):
        n = int(np.frombuffer(mv[off:off + 4], dtype=np.int32)[0]); off += 4
        ents = []
        for _ in range(n):
            score = int(np.frombuffer(mv[off:off + 4], dtype=np.int32)[0]); off += 4
            L, U = (int(v) for v in np.frombuffer(mv[off:off + 16], dtype=np.uint64)); off += 16
            mm, go, ge, alen, pairs = (int(v) for v in np.frombuffer(mv[off:off + 20], dtype=np.int32)); off += 20
            states = [int(v) for v in np.frombuffer(mv[off:off + 4 * pairs], dtype=np.int32)]; off += 4 * pairs
            ents.append(dict(score=score, L=L, U=U, mm=mm, gapo=go, gape=ge, aln_length=alen, states=states))
        out.append(ents)
    return out
