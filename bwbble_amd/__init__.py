"""bwbble_amd - host-side Python mirror of the C-ABI in include/bwbble_hip.h (ctypes, no torch types).

The product is libbwbble_hip.so (hand-written HIP for gfx950) plus the C host tools in
bwbble_amd/host; this module only binds the library for tests and bench.py.  It fails loudly when
the library is missing or no GPU is present: there is no CPU path here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BWB_LIB") or os.path.join(HERE, "libbwbble_hip.so")  # BWB_LIB: tests load the `make testlib` build
TEST_LIB_PATH = os.path.join(HERE, "libbwbble_hip_test.so")
HOST_BIN = os.path.join(HERE, "bin", "bwbble")
SYNTH_BIN = os.path.join(HERE, "bin", "bwb_synth")

EXPORTS = [
    "bwb_hip_device_count", "bwb_hip_last_error", "bwb_default_params", "bwb_hip_ctx_create", "bwb_hip_ctx_destroy",
    "bwb_hip_align_batch", "bwb_hip_batch_upload", "bwb_hip_batch_run", "bwb_hip_batch_result", "bwb_hip_get_stats",
    "bwb_hip_calc_d", "bwb_hip_rank16", "bwb_hip_rank_bench", "bwb_hip_rank_bench_lane", "bwb_hip_set_sa", "bwb_hip_locate", "bwb_hip_locate_stats",
    "bwb_hip_reset_stats", "bwb_hip_slot_upload", "bwb_hip_slot_submit", "bwb_hip_slot_wait", "bwb_hip_slot_result", "bwb_hip_flush", "bwb_hip_abi_version", "bwb_hip_ctx_create_streamed", "bwb_hip_device_numa_node",
    "bwb_hip_ctx_create_async", "bwb_hip_ctx_index_wait", "bwb_hip_setup_times", "bwb_hip_dtab_info",
]
ABI_VERSION = 3  # BWB_HIP_ABI_VERSION (include/bwbble_hip.h)
MAX_SLOTS = 8  # BWB_MAX_SLOTS


class Params(C.Structure):
    """bwb_params == aln_params_t (mg-aligner/align.h:48-79)."""
    _fields_ = [(n, C.c_int32) for n in (
        "max_diff", "max_gapo", "max_gape", "max_entries", "mm_score", "gapo_score", "gape_score",
        "seed_length", "max_diff_seed", "max_best", "no_indel_length", "matched_Ncontig", "use_precalc",
        "is_multiref", "n_threads")]


class Result(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("aln_off", C.POINTER(C.c_uint64)), ("alns", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("visits_single", C.c_uint64), ("visits_calc_d", C.c_uint64), ("visits_alphabet", C.c_uint64), ("heap_pops", C.c_uint64),
                ("heap_pushes", C.c_uint64), ("n_alignments", C.c_uint64), ("n_overflow_reads", C.c_uint64), ("n_parked_reads", C.c_uint64),
                ("bucket_loads_search", C.c_uint64), ("bucket_loads_calc_d", C.c_uint64), ("lane_iterations", C.c_uint64), ("wave_iterations", C.c_uint64),
                ("heap_entries_stored", C.c_uint64), ("heap_entries_loaded", C.c_uint64), ("record_loads", C.c_uint64),
                ("ms_calc_d", C.c_double), ("ms_search", C.c_double), ("ms_total", C.c_double),
                ("launches_calc_d", C.c_uint32), ("launches_search", C.c_uint32)]


ALN_DTYPE = np.dtype([("L", "<u8"), ("U", "<u8"), ("score", "<u2"), ("num_mm", "u1"), ("num_gapo", "u1"),
                      ("num_gape", "u1"), ("reserved", "u1"), ("aln_length", "<u2"), ("gap_run", "<u2", (8,)), ("reserved2", "<u8")])
assert ALN_DTYPE.itemsize == 48

_FLAG = {"-M": "mm_score", "-O": "gapo_score", "-E": "gape_score", "-n": "max_diff", "-k": "max_diff_seed",
         "-o": "max_gapo", "-e": "max_gape", "-l": "seed_length", "-m": "max_entries", "-t": "n_threads"}

_lib = None


class BwbError(RuntimeError):
    pass


def build(testlib=False):
    """Compiles the HIP library and the host tools in-tree (hipcc --offload-arch=gfx950); testlib: the test build as well."""
    subprocess.run(["make", "-s", "-C", HERE] + (["all", "testlib"] if testlib else []), check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BwbError(f"{LIB_PATH} is missing: run `make -C bwbble_amd` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        if not hasattr(L, "bwb_hip_abi_version") or L.bwb_hip_abi_version() != ABI_VERSION:
            raise BwbError(f"{LIB_PATH} implements another version of the C-ABI than this mirror (want {ABI_VERSION}): rebuild it")
        L.bwb_hip_last_error.restype = C.c_char_p
        L.bwb_hip_ctx_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        L.bwb_hip_ctx_destroy.argtypes = [C.c_void_p]
        L.bwb_hip_batch_upload.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.bwb_hip_batch_run.argtypes = [C.c_void_p]
        L.bwb_hip_batch_result.argtypes = [C.c_void_p, C.POINTER(Result)]
        L.bwb_hip_align_batch.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(Result)]
        L.bwb_hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.bwb_hip_calc_d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.bwb_hip_rank16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]
        L.bwb_hip_rank_bench.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.bwb_hip_rank_bench_lane.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.bwb_hip_reset_stats.argtypes = [C.c_void_p]
        L.bwb_hip_slot_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.bwb_hip_slot_submit.argtypes = [C.c_void_p, C.c_int]
        L.bwb_hip_slot_wait.argtypes = [C.c_void_p, C.c_int]
        L.bwb_hip_slot_result.argtypes = [C.c_void_p, C.c_int, C.POINTER(Result)]
        L.bwb_hip_flush.argtypes = [C.c_void_p]
        L.bwb_hip_set_sa.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.bwb_hip_locate.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.bwb_hip_dtab_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.bwb_hip_setup_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.bwb_hip_locate_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise BwbError(f"bwbble_hip error {rc}: {lib().bwb_hip_last_error().decode()}")


def device_count():
    return lib().bwb_hip_device_count()


def params(flags=()):
    """Defaults of set_default_aln_params (align.c:22-38) + `bwbble align` style flags (main.c:100-117)."""
    p = Params()
    lib().bwb_default_params(C.byref(p))
    flags = list(flags)
    i = 0
    while i < len(flags):
        if flags[i] == "-S":
            p.is_multiref = 0
            i += 1
        elif flags[i] == "-P":
            p.use_precalc = 1
            i += 1
        else:
            setattr(p, _FLAG[flags[i]], int(flags[i + 1]))
            i += 2
    return p


class BwtFile:
    """In-memory bwt_t (bwt.h:19-40) read from a reference-format .bwt file (bwt.c:66-82)."""

    def __init__(self, path, load_sa=False):
        hdr = np.fromfile(path, dtype="<u8", count=22)
        self.hdr = hdr[:5].copy()
        self.length, self.num_words, self.num_sa, self.num_occ, self.sa0_index = (int(v) for v in hdr[:5])
        self.C = hdr[5:22].copy()
        off = 22 * 8
        # memory-mapped: a GRCh37-scale file is 12 GB, and the ranks of a multi-GPU run share its pages
        self.bwt = np.memmap(path, dtype="<u4", mode="r", offset=off, shape=(self.num_words,))
        off += 4 * self.num_words
        self.O = np.memmap(path, dtype="<u8", mode="r", offset=off, shape=(self.num_occ * 16,))
        off += 8 * self.num_occ * 16
        self.SA = np.memmap(path, dtype="<u8", mode="r", offset=off, shape=(self.num_sa,)) if load_sa else None


class Context:
    """One GPU context (bwb_hip_ctx): device-resident FM-index + batch state."""

    def __init__(self, bwt, device=0):
        if isinstance(bwt, str):
            bwt = BwtFile(bwt)
        if device_count() < 1:
            raise BwbError("no HIP device visible: the alignment path has no CPU fallback")
        self.bwt = bwt
        self._h = C.c_void_p()
        _chk(lib().bwb_hip_ctx_create(device, bwt.hdr.ctypes.data, bwt.C.ctypes.data, bwt.bwt.ctypes.data,
                                      bwt.O.ctypes.data, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().bwb_hip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- batch API ------------------------------------------------------------------------------
    def upload(self, p, seqs, lens):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        if seqs.ndim != 2 or len(lens) != seqs.shape[0]:
            raise ValueError("seqs must be (n_reads, stride) uint8 with one length per row")
        self._keep = (seqs, lens)
        _chk(lib().bwb_hip_batch_upload(self._h, C.byref(p), seqs.ctypes.data, lens.ctypes.data, seqs.shape[0], max(seqs.shape[1], 1)))

    def run(self):
        _chk(lib().bwb_hip_batch_run(self._h))

    # -- streaming API: up to MAX_SLOTS batches resident, slices that park instead of draining ---------------------
    def slot_upload(self, slot, p, seqs, lens, carry=None):
        """carry: codes of the last read longer than the seed that precedes this batch in the file (D_seed of leading short reads)"""
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint16)
        if seqs.ndim != 2 or len(lens) != seqs.shape[0]:
            raise ValueError("seqs must be (n_reads, stride) uint8 with one length per row")
        carry = None if carry is None else np.ascontiguousarray(carry, dtype=np.uint8)
        _chk(lib().bwb_hip_slot_upload(self._h, slot, C.byref(p), seqs.ctypes.data, lens.ctypes.data, seqs.shape[0], max(seqs.shape[1], 1),
                                       None if carry is None else carry.ctypes.data, 0 if carry is None else len(carry)))

    def slot_submit(self, slot):
        _chk(lib().bwb_hip_slot_submit(self._h, slot))

    def slot_wait(self, slot):
        _chk(lib().bwb_hip_slot_wait(self._h, slot))

    def slot_result(self, slot):
        r = Result()
        _chk(lib().bwb_hip_slot_result(self._h, slot, C.byref(r)))
        return self._unpack(r)

    def flush(self):
        _chk(lib().bwb_hip_flush(self._h))

    def reset_stats(self):
        _chk(lib().bwb_hip_reset_stats(self._h))

    def result(self):
        r = Result()
        _chk(lib().bwb_hip_batch_result(self._h, C.byref(r)))
        return self._unpack(r)

    @staticmethod
    def _unpack(r):
        n = r.n_reads
        off = np.ctypeslib.as_array(r.aln_off, shape=(n + 1,)).copy()
        total = int(off[n])
        alns = np.frombuffer(C.string_at(r.alns, total * ALN_DTYPE.itemsize), dtype=ALN_DTYPE) if total else np.zeros(0, dtype=ALN_DTYPE)
        return off, alns

    def align(self, p, seqs, lens):
        self.upload(p, seqs, lens)
        self.run()
        return self.result()

    def stats(self):
        s = Stats()
        _chk(lib().bwb_hip_get_stats(self._h, C.byref(s)))
        return s

    def calc_d(self, p, seqs, lens):
        """D / D_seed of every read as (n, maxlen+1, 2) and (n, seed_length+1, 2) int32 (num_diff, sa_intv_width)."""
        self.upload(p, seqs, lens)
        n, maxlen = len(lens), int(max(lens)) if len(lens) else 0
        D = np.zeros((n, maxlen + 1, 2), dtype=np.int32)
        Ds = np.zeros((n, p.seed_length + 1, 2), dtype=np.int32)
        _chk(lib().bwb_hip_calc_d(self._h, D.ctypes.data, Ds.ctypes.data))
        return D, Ds

    # -- rank -----------------------------------------------------------------------------------
    def rank16(self, pos, inc=0, exact=False):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((len(pos), 16), dtype=np.uint64)
        _chk(lib().bwb_hip_rank16(self._h, pos.ctypes.data, len(pos), inc, int(exact), out.ctypes.data))
        return out

    def rank_bench(self, n, iters=5, seed=1, lane=False):
        """Random Occ16 micro-benchmark: octet layout (8 lanes share a bucket) or the alignment kernels' lane layout."""
        ms, cs = C.c_double(), C.c_uint64()
        f = lib().bwb_hip_rank_bench_lane if lane else lib().bwb_hip_rank_bench
        _chk(f(self._h, n, iters, seed, C.byref(ms), C.byref(cs)))
        return ms.value, cs.value

    def set_sa(self, SA):
        SA = np.ascontiguousarray(SA, dtype=np.uint64)
        _chk(lib().bwb_hip_set_sa(self._h, SA.ctypes.data, len(SA)))

    def locate(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(len(rows), dtype=np.uint64)
        _chk(lib().bwb_hip_locate(self._h, rows.ctypes.data, len(rows), out.ctypes.data))
        return out

    def dtab_info(self):
        """the context's calculate_d table: {K (0: none), build seconds, bytes}"""
        k, sec, nb = C.c_int(), C.c_double(), C.c_uint64()
        _chk(lib().bwb_hip_dtab_info(self._h, C.byref(k), C.byref(sec), C.byref(nb)))
        return {"K": k.value, "build_s": round(sec.value, 3), "GB": round(nb.value / 1e9, 2)}

    def locate_stats(self):
        """(rows, invPsi steps = rank-block visits, kernel ms) of the last locate()"""
        n, st, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        _chk(lib().bwb_hip_locate_stats(self._h, C.byref(n), C.byref(st), C.byref(ms)))
        return n.value, st.value, ms.value


# -- formats ------------------------------------------------------------------------------------

def encode_reads(ascii_reads):
    """read->seq codes of fastq2reads (io.c:467; io.h:112-130): A0 G1 C2 T3, anything else 4."""
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        lut[ord(ch)] = v
        lut[ord(ch.lower())] = v
    lens = np.array([len(r) for r in ascii_reads], dtype=np.uint16)
    stride = int(lens.max()) if len(lens) else 1
    seqs = np.full((len(ascii_reads), max(stride, 1)), 4, dtype=np.uint8)
    for i, r in enumerate(ascii_reads):
        seqs[i, :len(r)] = lut[np.frombuffer(r.encode(), dtype=np.uint8)]
    return seqs, lens


def load_fastq_codes(path, max_reads=0, chunk=500000):
    """FASTQ -> (codes (n, stride) uint8, lens uint16) without a Python loop per read (same encoding as encode_reads);
    the gather runs over `chunk` reads at a time so that a 10 M-read file does not need tens of GB of index temporaries."""
    data = np.fromfile(path, dtype=np.uint8)
    nl = np.flatnonzero(data == 10)
    if len(data) and data[-1] != 10:
        nl = np.append(nl, len(data))
    n = len(nl) // 4
    if max_reads:
        n = min(n, max_reads)
    if n == 0:
        return np.full((0, 1), 4, dtype=np.uint8), np.zeros(0, dtype=np.uint16)
    e0 = nl[1:4 * n:4]
    s0 = nl[0:4 * n:4] + 1
    lens = (e0 - s0).astype(np.uint16)
    stride = int(lens.max())
    lut = np.full(256, 4, dtype=np.uint8)
    for ch, v in (("A", 0), ("G", 1), ("C", 2), ("T", 3)):
        lut[ord(ch)] = v
        lut[ord(ch.lower())] = v
    seqs = np.empty((n, stride), dtype=np.uint8)
    cols = np.arange(stride)[None, :]
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        idx = s0[a:b, None] + cols
        valid = cols < lens[a:b, None]
        seqs[a:b] = np.where(valid, lut[data[np.minimum(idx, len(data) - 1)]], 4)
    return seqs, lens


def read_fastq(path, max_reads=0):
    """Sequence lines of a 4-line FASTQ (the subset of fastq2reads, io.c:410-515, that align needs)."""
    out = []
    with open(path) as f:
        while True:
            name = f.readline()
            if not name:
                break
            if not name.startswith("@"):
                continue
            seq = f.readline().rstrip("\n")
            f.readline()
            f.readline()
            out.append(seq)
            if max_reads and len(out) >= max_reads:
                break
    return out


def aln_path(rec):
    """Edit path bytes (index 0 = first backward-search step) of one hit from its gap runs."""
    alen = int(rec["aln_length"])
    path = np.zeros(256 + 8, dtype=np.uint8)
    for run in rec["gap_run"]:
        run = int(run)
        if run == 0xFFFF:
            continue
        start, ln, is_del = run & 0xFF, (run >> 8) & 0x7F, run >> 15
        path[start:start + ln] = 2 if is_del else 1
    return path[:alen]


def aln_bytes(aln_off, alns):
    """Serialises hits exactly like alns2alnf_bin (align.c:345-382): the bytes of a .aln file."""
    out = bytearray()
    i32 = lambda v: int(v).to_bytes(4, "little", signed=True)
    for r in range(len(aln_off) - 1):
        lo, hi = int(aln_off[r]), int(aln_off[r + 1])
        out += i32(hi - lo)
        for rec in alns[lo:hi]:
            out += i32(rec["score"]) + int(rec["L"]).to_bytes(8, "little") + int(rec["U"]).to_bytes(8, "little")
            out += i32(rec["num_mm"]) + i32(rec["num_gapo"]) + i32(rec["num_gape"]) + i32(rec["aln_length"])
            path = aln_path(rec)
            if len(path) == 0:
                out += i32(0)
                continue
            pairs = []
            state, count = int(path[-1]), 1
            for s in path[-2::-1]:
                if int(s) == state:
                    count += 1
                else:
                    pairs.append(state | (count << 2))
                    state, count = int(s), 1
            pairs.append(state | (count << 2))
            out += i32(len(pairs))
            for v in pairs:
                out += i32(v)
    return bytes(out)
