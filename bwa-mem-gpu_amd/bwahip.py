"""ctypes binding of libbwahip.so -- the host-side mirror of the reference interface.

The product is the C-ABI shared library (include/bwahip.h); this module only makes it callable from
the Python test-suite and bench.py.  It never computes anything itself and it never falls back to a
CPU implementation: if the HIP library has not been built, importing the library fails loudly.

Mirrors: mem_opt_t / mem_opt_init (bwa.h:86, bwamem.c:74), mem_process_seqs (bwamem.h:69),
mem_align1_core (bwamem.c:1061) and the stage boundaries used by the parity tests.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BWAHIP_LIB") or os.path.join(_HERE, "libbwahip.so")
TOOLS = os.path.join(_HERE, "tools")


class Opt(C.Structure):  # bwahip_opt_t == mem_opt_t (bwa.h:86-118)
    _fields_ = [("max_mem_intv", C.c_uint64), ("a", C.c_int), ("b", C.c_int), ("o_del", C.c_int), ("e_del", C.c_int),
                ("o_ins", C.c_int), ("e_ins", C.c_int), ("pen_unpaired", C.c_int), ("pen_clip5", C.c_int),
                ("pen_clip3", C.c_int), ("w", C.c_int), ("zdrop", C.c_int), ("T", C.c_int), ("flag", C.c_int),
                ("min_seed_len", C.c_int), ("min_chain_weight", C.c_int), ("max_chain_extend", C.c_int),
                ("split_factor", C.c_float), ("split_width", C.c_int), ("max_occ", C.c_int), ("max_chain_gap", C.c_int),
                ("n_threads", C.c_int), ("chunk_size", C.c_int), ("mask_level", C.c_float), ("drop_ratio", C.c_float),
                ("XA_drop_ratio", C.c_float), ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float),
                ("mapQ_coef_fac", C.c_int), ("max_ins", C.c_int), ("max_matesw", C.c_int), ("max_XA_hits", C.c_int),
                ("max_XA_hits_alt", C.c_int), ("mat", C.c_int8 * 25)]


class Seq(C.Structure):  # bwahip_seq_t == bseq1_t (bwa.h:58-63)
    _fields_ = [("l_seq", C.c_int), ("id", C.c_int), ("name", C.c_char_p), ("comment", C.c_char_p),
                ("seq", C.POINTER(C.c_char)), ("qual", C.c_char_p), ("sam", C.POINTER(C.c_char)),
                ("l_name", C.c_int8), ("l_comment", C.c_int8), ("l_qual", C.c_int16)]


class AlnReg(C.Structure):  # bwahip_alnreg_t == mem_alnreg_t (bwa.h:145-163)
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64), ("hash", C.c_uint64), ("frac_rep", C.c_float),
                ("qb", C.c_int), ("qe", C.c_int), ("rid", C.c_int), ("score", C.c_int), ("truesc", C.c_int),
                ("sub", C.c_int), ("alt_sc", C.c_int), ("csub", C.c_int), ("sub_n", C.c_int), ("w", C.c_int),
                ("seedcov", C.c_int), ("secondary", C.c_int), ("secondary_all", C.c_int), ("seedlen0", C.c_int),
                ("n_comp", C.c_int, 30), ("is_alt", C.c_int, 2)]


class AlnRegV(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("a", C.POINTER(AlnReg))]


class Bwt(C.Structure):  # bwahip_bwt_t == bwt_t (bwt.h:48-60)
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.c_void_p), ("cnt_table", C.c_uint32 * 256), ("sa_intv", C.c_int), ("n_sa", C.c_uint64), ("sa", C.c_void_p)]


class Ann(C.Structure):  # bwahip_ann_t == bntann1_t (bntseq.h:41-48)
    _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("n_ambs", C.c_int32), ("gi", C.c_uint32), ("is_alt", C.c_int32),
                ("name", C.c_char_p), ("anno", C.c_char_p)]


class Bns(C.Structure):  # bwahip_bns_t == bntseq_t (bntseq.h:56-64)
    _fields_ = [("l_pac", C.c_int64), ("n_seqs", C.c_int32), ("seed", C.c_uint32), ("anns", C.POINTER(Ann)), ("n_holes", C.c_int32),
                ("ambs", C.c_void_p), ("fp_pac", C.c_void_p)]


class PeStat(C.Structure):
    _fields_ = [("low", C.c_int), ("high", C.c_int), ("failed", C.c_int), ("avg", C.c_double), ("std", C.c_double)]


class StreamStats(C.Structure):  # bwahip_stream_t
    _fields_ = [("chunk_bases", C.c_int64), ("max_reads", C.c_int64), ("keep_comments", C.c_int), ("reader_threads", C.c_int),
                ("n_reads", C.c_int64), ("n_batches", C.c_int64), ("sam_bytes", C.c_int64), ("seconds", C.c_double),
                ("reader_wait_s", C.c_double), ("gpu_busy_s", C.c_double), ("write_s", C.c_double)]


ERRORS = {0: "ok", -1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "EIO", -5: "ECAPACITY", -6: "EINTERNAL"}

STAGE_INTV, STAGE_CHAIN, STAGE_CHAIN_FLT, STAGE_REGS, STAGE_REGS_PRE, STAGE_SEEDS = 1, 2, 3, 4, 5, 6
TAG_READ = 100


class BwahipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libbwahip.so.  There is no fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BwahipError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the hot path)")
    L = C.CDLL(LIB_PATH)
    vp, i64p, u64p, ip = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)
    L.bwahip_version.restype = C.c_char_p
    L.bwahip_opt_init.argtypes = [C.POINTER(Opt)]
    L.bwahip_opt_fill_scmat.argtypes = [C.POINTER(Opt)]
    L.bwahip_ctx_tune.argtypes = [vp, C.c_char_p, C.c_int]
    L.bwahip_kat_ksw_align.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.bwahip_init_from_files.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.bwahip_rccl_unique_id.argtypes = [vp]
    L.bwahip_init_rccl.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.bwahip_ctx_clone.argtypes = [vp, C.POINTER(vp)]
    L.bwahip_ctx_clone_on.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.bwahip_stream_run.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Opt), C.POINTER(PeStat), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(StreamStats)]
    L.bwahip_process_seqs_text.argtypes = [vp, C.POINTER(Opt), C.c_int64, C.c_int, C.POINTER(Seq), C.c_void_p, C.POINTER(C.c_char_p), i64p, C.POINTER(i64p)]
    L.bwahip_fastq_open.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.bwahip_fastq_next.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(C.POINTER(Seq)), C.POINTER(C.c_int)]
    L.bwahip_fastq_close.argtypes = [vp]
    L.bwahip_fastq_close.restype = None
    L.bwahip_fastq_open_mt.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.bwahip_fastq_next_batch.argtypes = [vp, C.c_int64, C.c_int, C.POINTER(vp), C.POINTER(C.POINTER(Seq)), C.POINTER(C.c_int)]
    L.bwahip_fastq_batch_seqs.argtypes = [vp, C.POINTER(C.c_int)]
    L.bwahip_fastq_batch_seqs.restype = C.POINTER(Seq)
    L.bwahip_fastq_batch_release.argtypes = [vp]
    L.bwahip_fastq_batch_release.restype = None
    L.bwahip_ctx_set_rg_id.argtypes = [vp, C.c_char_p]
    L.bwahip_init_device.argtypes = [C.POINTER(Bwt), C.POINTER(Bns), vp, C.c_int, C.POINTER(vp)]
    L.bwahip_destroy.argtypes = [vp]
    L.bwahip_run_stages.argtypes = [vp, C.POINTER(Opt), C.c_int, vp, vp, C.c_int, C.POINTER(i64p), i64p]
    L.bwahip_batch_upload.argtypes = [vp, C.c_int, vp, vp]
    L.bwahip_batch_attach.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int64]
    L.bwahip_batch_attach_text.argtypes = [vp, vp, vp, vp, vp]
    L.bwahip_batch_run_sam.argtypes = [vp, C.POINTER(Opt), C.c_int64, C.POINTER(PeStat), C.POINTER(C.c_float), C.c_int]
    L.bwahip_batch_sam.argtypes = [vp, C.POINTER(vp), i64p, vp]
    L.bwahip_batch_run.argtypes = [vp, C.POINTER(Opt), C.POINTER(C.c_float), C.c_int]
    L.bwahip_batch_counters.argtypes = [vp, u64p, C.c_int]
    L.bwahip_kernel_name.restype = C.c_char_p
    L.bwahip_kat_introsort.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    L.bwahip_kat_occ4.argtypes = [vp, C.c_int, vp, vp]
    L.bwahip_index_footprint.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.bwahip_kat_sa.argtypes = [vp, C.c_int, vp, vp]
    L.bwahip_kat_kmer_table.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.bwahip_kat_extend.argtypes = [vp, C.c_int, vp, vp, vp]
    L.bwahip_kat_ksw_extend.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.bwahip_align_batch.argtypes = [vp, C.POINTER(Opt), C.c_int, C.POINTER(Seq), C.POINTER(AlnRegV)]
    L.bwahip_process_seqs.argtypes = [vp, C.POINTER(Opt), C.c_int64, C.c_int, C.POINTER(Seq), C.POINTER(PeStat)]
    L.bwahip_batch_download.argtypes = [vp, C.POINTER(AlnRegV)]
    L.bwahip_last_pe_stats.argtypes = [vp, C.POINTER(PeStat), u64p]
    L.bwahip_seqs_take_sam.argtypes = [C.POINTER(Seq), C.c_int, C.POINTER(vp), i64p]
    _lib = L
    return L


def big_bytes(ptr, n):
    """n bytes at ptr as a bytes object; ctypes.string_at takes a C int, and a batch's SAM text can exceed 2^31 bytes."""
    addr = ptr.value if hasattr(ptr, "value") else ptr
    if isinstance(addr, bytes):                                # a c_char_p: take the address of its buffer
        addr = C.cast(ptr, C.c_void_p).value
    if n <= 0:
        return b""
    return bytes((C.c_char * n).from_address(addr))


def _check(rc, what):
    if rc != 0:
        raise BwahipError(f"{what} failed: {ERRORS.get(rc, rc)}")


def default_opt():
    o = Opt()
    lib().bwahip_opt_init(C.byref(o))
    return o


NT4 = np.full(256, 4, dtype=np.uint8)
for _c, _v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    NT4[_c] = _v
NT4[ord("-")] = 5


def pack_reads(reads):
    """list of ASCII/bytes reads -> (codes uint8 concatenated, offsets int64[n+1])."""
    lens = np.fromiter((len(r) for r in reads), dtype=np.int64, count=len(reads))
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    buf = b"".join(r if isinstance(r, bytes) else r.encode() for r in reads)
    codes = NT4[np.frombuffer(buf, dtype=np.uint8)] if buf else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(codes), off


def parse_records(words):
    """int64 record stream [tag, n, n values]* -> list of (tag, ndarray)."""
    out, i, n = [], 0, len(words)
    while i < n:
        tag, cnt = int(words[i]), int(words[i + 1])
        out.append((tag, words[i + 2:i + 2 + cnt]))
        i += 2 + cnt
    return out


def read_record_file(path):
    return parse_records(np.fromfile(path, dtype=np.int64))


class Context:
    """One GPU, one index resident in HBM (bwahip_ctx)."""

    def __init__(self, prefix, device=0):
        self._h = C.c_void_p()
        self._keep = None
        if prefix is not None:
            _check(lib().bwahip_init_from_files(os.fsencode(prefix), device, C.byref(self._h)), "bwahip_init_from_files")

    @classmethod
    def from_device_arrays(cls, meta, bwt_ptr, sa_ptr, pac_ptr, device=0):
        """Adopt index arrays that already sit in HBM (bwahip_init_device), e.g. after the RCCL broadcast."""
        self = cls(None, device)
        b = Bwt()
        b.primary = meta["primary"]
        for i in range(5):
            b.L2[i] = meta["L2"][i]
        b.seq_len, b.bwt_size, b.bwt = meta["seq_len"], meta["bwt_words"], bwt_ptr
        b.sa_intv, b.n_sa, b.sa = meta["sa_intv"], meta["n_sa"], sa_ptr
        anns = (Ann * len(meta["contigs"]))()
        for i, (name, off, ln, alt) in enumerate(meta["contigs"]):
            anns[i].offset, anns[i].len, anns[i].is_alt = off, ln, alt
            anns[i].name, anns[i].anno = name.encode(), b""
        n = Bns()
        n.l_pac, n.n_seqs, n.seed, n.anns = meta["l_pac"], len(meta["contigs"]), 11, anns
        self._keep = (b, anns, n)
        _check(lib().bwahip_init_device(C.byref(b), C.byref(n), pac_ptr, device, C.byref(self._h)), "bwahip_init_device")
        return self

    @staticmethod
    def rccl_unique_id():
        """128-byte ncclUniqueId (made on one rank; hand it to the others before from_rccl)."""
        buf = C.create_string_buffer(128)
        _check(lib().bwahip_rccl_unique_id(buf), "bwahip_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_rccl(cls, prefix, rank, world, unique_id, device=0):
        """Collective: rank 0 loads `prefix`, all ranks receive the index over RCCL into their own HBM (bwahip_init_rccl)."""
        self = cls(None, device)
        idb = C.create_string_buffer(bytes(unique_id), 128)
        _check(lib().bwahip_init_rccl(os.fsencode(prefix) if prefix is not None else None, rank, world, idb, device, C.byref(self._h)), "bwahip_init_rccl")
        return self

    def clone(self):
        """A further context on the same GPU sharing this one's index in HBM (bwahip_ctx_clone); keep `self` alive longer."""
        other = Context(None)
        _check(lib().bwahip_ctx_clone(self._h, C.byref(other._h)), "bwahip_ctx_clone")
        other._keep = self
        return other

    def clone_on(self, device):
        """A context on another GPU with its own copy of the index, made device to device (bwahip_ctx_clone_on)."""
        other = Context(None)
        _check(lib().bwahip_ctx_clone_on(self._h, device, C.byref(other._h)), "bwahip_ctx_clone_on")
        other._keep = self
        return other

    def close(self):
        if self._h:
            lib().bwahip_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run_stages(self, codes, off, stages, opt=None):
        opt = opt or default_opt()
        mask = 0
        for s in stages:
            mask |= 1 << s
        out, n = C.POINTER(C.c_int64)(), C.c_int64()
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        _check(lib().bwahip_run_stages(self._h, C.byref(opt), len(off) - 1, codes.ctypes.data, off.ctypes.data, mask,
                                       C.byref(out), C.byref(n)), "bwahip_run_stages")
        words = np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)
        C.CDLL(None).free(out)
        return parse_records(words)

    def tune(self, **kw):
        """Set hand-off thresholds of the heavy-read kernels (bwahip_ctx_tune): intv_cap, smem_lanes, heavy_mult, ..."""
        for k, v in kw.items():
            _check(lib().bwahip_ctx_tune(self._h, k.encode(), int(v)), f"bwahip_ctx_tune({k})")

    def set_rg_id(self, rg_id):
        """Read-group id printed as RG:Z: on every record (bwa mem -R '@RG\\tID:<id>...', bwa_set_rg bwa.c:562); None / '' = none."""
        _check(lib().bwahip_ctx_set_rg_id(self._h, rg_id.encode() if rg_id else None), "bwahip_ctx_set_rg_id")

    def process_seqs(self, names, seqs, quals=None, opt=None, n_processed=0, pes0=None, comments=None):
        """mem_process_seqs: list of names / ASCII reads (/ quals) -> list of SAM text (bytes) per read."""
        opt = opt or default_opt()
        n = len(seqs)
        arr = (Seq * n)()
        keep = []
        for i in range(n):
            sb = C.create_string_buffer(bytes(seqs[i]), len(seqs[i]) + 1)
            keep.append(sb)
            arr[i].l_seq, arr[i].id = len(seqs[i]), i
            arr[i].name = bytes(names[i])
            arr[i].comment = bytes(comments[i]) if comments is not None and comments[i] is not None else None
            arr[i].seq = C.cast(sb, C.POINTER(C.c_char))
            arr[i].qual = bytes(quals[i]) if quals is not None else None
        _check(lib().bwahip_process_seqs(self._h, C.byref(opt), n_processed, n, arr, pes0), "bwahip_process_seqs")
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        out = []
        for i in range(n):
            out.append(C.string_at(arr[i].sam))
            libc.free(C.cast(arr[i].sam, C.c_void_p))
        return out

    def process_seqs_array(self, arr, n, opt=None, n_processed=0, pes0=None):
        """bwahip_process_seqs on a bseq1_t array (e.g. a FastqReader batch); returns the batch's SAM text (bwahip_seqs_take_sam)."""
        opt = opt or default_opt()
        _check(lib().bwahip_process_seqs(self._h, C.byref(opt), n_processed, n, arr, pes0), "bwahip_process_seqs")
        out, ln = C.c_void_p(), C.c_int64()
        _check(lib().bwahip_seqs_take_sam(arr, n, C.byref(out), C.byref(ln)), "bwahip_seqs_take_sam")
        sam = big_bytes(out, ln.value)
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(out)
        return sam

    def process_seqs_text_array(self, arr, n, opt=None, n_processed=0, pes0=None, want_offsets=False):
        """bwahip_process_seqs_text on a bseq1_t array: the batch's SAM as one bytes object (copied out of the context's buffer)."""
        opt = opt or default_opt()
        sam, ln, off = C.c_char_p(), C.c_int64(), C.POINTER(C.c_int64)()
        _check(lib().bwahip_process_seqs_text(self._h, C.byref(opt), n_processed, n, arr, pes0, C.byref(sam), C.byref(ln), C.byref(off)), "bwahip_process_seqs_text")
        text = big_bytes(sam, ln.value)
        return (text, [off[i] for i in range(n + 1)]) if want_offsets else text

    def last_pe_stats(self):
        """(pestat[4] as dicts, mate-rescue alignments run on the GPU, regions they added) of the last PE batch."""
        pes = (PeStat * 4)()
        cnt = (C.c_uint64 * 4)()
        _check(lib().bwahip_last_pe_stats(self._h, pes, cnt), "bwahip_last_pe_stats")
        self.pe_counters = dict(sw=int(cnt[0]), added=int(cnt[1]), max_sw_per_pair=int(cnt[2]), pairs_rescued=int(cnt[3]))
        return [dict(low=p.low, high=p.high, failed=p.failed, avg=p.avg, std=p.std) for p in pes], int(cnt[0]), int(cnt[1])

    def batch_upload(self, codes, off):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        _check(lib().bwahip_batch_upload(self._h, len(off) - 1, codes.ctypes.data, off.ctypes.data), "bwahip_batch_upload")

    def batch_attach(self, n, seq_ptr, off_ptr, max_len, total_bases):
        """Use reads already resident in HBM (device pointers); nothing is copied."""
        _check(lib().bwahip_batch_attach(self._h, n, seq_ptr, off_ptr, max_len, total_bases), "bwahip_batch_attach")

    def batch_attach_text(self, qual_ptr, qual_off_ptr, names_ptr, name_off_ptr):
        _check(lib().bwahip_batch_attach_text(self._h, qual_ptr, qual_off_ptr, names_ptr, name_off_ptr), "bwahip_batch_attach_text")

    def batch_run_sam(self, opt=None, n_processed=0, pes0=None):
        """Hot path + finalisation + SAM text, all on the GPU, over the attached batch; returns per-stage milliseconds."""
        opt = opt or default_opt()
        nk = lib().bwahip_n_kernels()
        ms = (C.c_float * nk)()
        _check(lib().bwahip_batch_run_sam(self._h, C.byref(opt), n_processed, pes0, ms, nk), "bwahip_batch_run_sam")
        return {lib().bwahip_kernel_name(i).decode(): float(ms[i]) for i in range(nk)}

    def batch_sam(self):
        out, ln = C.c_void_p(), C.c_int64()
        _check(lib().bwahip_batch_sam(self._h, C.byref(out), C.byref(ln), None), "bwahip_batch_sam")
        sam = big_bytes(out, ln.value)
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(out)
        return sam

    def batch_run(self, opt=None):
        opt = opt or default_opt()
        nk = lib().bwahip_n_kernels()
        ms = (C.c_float * nk)()
        _check(lib().bwahip_batch_run(self._h, C.byref(opt), ms, nk), "bwahip_batch_run")
        return {lib().bwahip_kernel_name(i).decode(): float(ms[i]) for i in range(nk)}

    @staticmethod
    def stage_names():
        return ["k_smem(passes 1-2)+k_smem_heavy+k_smem3(pass 3)+k_intv_sort", "k_seeds", "k_chain(+k_chain_big,k_chain_flt)", "k_seed_sw (only with -W / reads > 700 bp)",
                "k_extend_spec+k_extend(+k_extend_big, dedup/patch)", "pe_rescue = PE: k_pestat + k_pe_prepare + k_pe_copy (insert sizes, lists; the rescue kernels k_matesw_sw / k_matesw start on the second stream)",
                "k_mark (mark primary) + PE: k_pair (pairing) beside the rescue kernels, then both for the rescued pairs",
                "k_cigar (mem_reg2aln: mapQ, CIGAR by ksw_global2 backtrack, NM/MD)", "k_sam size + scan + write (SAM text)"]

    @staticmethod
    def output_description(paired):
        return "SAM text of the batch in HBM (== seqs[i].sam of mem_process_seqs, " + ("paired-end: mate rescue, pairing, paired records)" if paired else "single-end)")

    def counters(self):
        buf = (C.c_uint64 * 32)()
        _check(lib().bwahip_batch_counters(self._h, buf, 32), "bwahip_batch_counters")
        names = ["extend", "blocks", "sa", "lf", "intv", "seeds", "cells", "max_extends", "chain_build_max", "chain_sort_max", "chain_flt_max",
                 "chain_write_max", "max_seeds", "max_chains", "ext_max", "ext_dedup_max", "heavy_blocks", "heavy_intv", "heavy_reads",
                 "dp_rows_1col", "dp_rows_ncol", "dedup_sort1_max", "dedup_loop_max", "dedup_sort2_max",
                 "pass3_blocks", "pass3_intv", "pass3_jumped", "_27", "_28", "_29", "_30", "_31"]
        return {k: int(buf[i]) for i, k in enumerate(names)}

    def kat_introsort(self, k64, score, qb, mode):
        """(idx_par, idx_seq, ran_parallel): the wavefront's exact introsort and the one-lane restatement of ksort.h on the same keys."""
        k64 = np.ascontiguousarray(k64, dtype=np.int64); score = np.ascontiguousarray(score, dtype=np.int32); qb = np.ascontiguousarray(qb, dtype=np.int32)
        n = len(k64)
        a = np.empty(n, dtype=np.int32); b = np.empty(n, dtype=np.int32); st = np.zeros(2, dtype=np.int32)
        _check(lib().bwahip_kat_introsort(self._h, n, mode, k64.ctypes.data, score.ctypes.data, qb.ctypes.data, a.ctypes.data, b.ctypes.data, st.ctypes.data), "bwahip_kat_introsort")
        assert st[1] == 0
        return a, b, bool(st[0])

    def kat_occ4(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros((len(k), 4), dtype=np.uint64)
        _check(lib().bwahip_kat_occ4(self._h, len(k), k.ctypes.data, out.ctypes.data), "bwahip_kat_occ4")
        return out

    def kat_sa(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64)
        out = np.zeros(len(k), dtype=np.uint64)
        _check(lib().bwahip_kat_sa(self._h, len(k), k.ctypes.data, out.ctypes.data), "bwahip_kat_sa")
        return out

    def index_footprint(self):
        """{'sa_intv', 'kmer_k', 'bwt_gb', 'sa_gb', 'pac_gb', 'interval_table_gb'} of the index in HBM (bwahip_index_footprint)."""
        a, k, b = C.c_int(), C.c_int(), (C.c_uint64 * 4)()
        _check(lib().bwahip_index_footprint(self._h, C.byref(a), C.byref(k), b), "bwahip_index_footprint")
        return {"sa_intv": a.value, "kmer_k": k.value, "bwt_gb": round(b[0] / 1e9, 2), "sa_gb": round(b[1] / 1e9, 2), "pac_gb": round(b[2] / 1e9, 2),
                "interval_table_gb": round(b[3] / 1e9, 2)}

    def kat_kmer_table(self):
        """(K, mismatches): the interval table against forward bwt_extend calls (bwahip_kat_kmer_table)."""
        k, bad = C.c_int(), C.c_uint64()
        _check(lib().bwahip_kat_kmer_table(self._h, C.byref(k), C.byref(bad)), "bwahip_kat_kmer_table")
        return k.value, bad.value

    def kat_ksw_extend(self, params, q, qoff, t, toff):
        params = np.ascontiguousarray(params, dtype=np.int32)
        q, t = np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)
        qoff, toff = np.ascontiguousarray(qoff, dtype=np.int64), np.ascontiguousarray(toff, dtype=np.int64)
        out = np.zeros((len(params), 6), dtype=np.int32)
        _check(lib().bwahip_kat_ksw_extend(self._h, len(params), params.ctypes.data, q.ctypes.data, qoff.ctypes.data, t.ctypes.data,
                                           toff.ctypes.data, out.ctypes.data), "bwahip_kat_ksw_extend")
        return out

    def kat_ksw_align(self, params, q, qoff, t, toff, mat=None):
        """ksw_align2 on the device; params n x 8 (qlen, tlen, xtra, o_del, e_del, o_ins, e_ins, 0) -> n x 7."""
        params = np.ascontiguousarray(params, dtype=np.int32)
        q, t = np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)
        qoff, toff = np.ascontiguousarray(qoff, dtype=np.int64), np.ascontiguousarray(toff, dtype=np.int64)
        out = np.zeros((len(params), 7), dtype=np.int32)
        m = np.ascontiguousarray(mat, dtype=np.int8) if mat is not None else None
        _check(lib().bwahip_kat_ksw_align(self._h, len(params), params.ctypes.data, m.ctypes.data if m is not None else None, q.ctypes.data,
                                          qoff.ctypes.data, t.ctypes.data, toff.ctypes.data, out.ctypes.data), "bwahip_kat_ksw_align")
        return out

    def kat_extend(self, ik3, is_back):
        ik3 = np.ascontiguousarray(ik3, dtype=np.uint64)
        is_back = np.ascontiguousarray(is_back, dtype=np.int32)
        out = np.zeros((len(is_back), 12), dtype=np.uint64)
        _check(lib().bwahip_kat_extend(self._h, len(is_back), ik3.ctypes.data, is_back.ctypes.data, out.ctypes.data), "bwahip_kat_extend")
        return out


# ---------------------------------------------------------------- tooling wrappers (synthetic data, index build)
class FastqReader:
    """bwahip_fastq_*: batches of a FASTA/FASTQ file (pair), plain or gzip, as bseq_read (bwa.c:191) cuts them."""

    def __init__(self, path1, path2=None, threads=0):
        self._h = C.c_void_p()
        _check(lib().bwahip_fastq_open_mt(os.fsencode(path1), os.fsencode(path2) if path2 else None, threads, C.byref(self._h)), "bwahip_fastq_open_mt")

    def next(self, chunk_bases, keep_comments=False):
        """-> (pointer to the bseq1_t array owned by the reader, n); n == 0 at the end of the input."""
        arr, n = C.POINTER(Seq)(), C.c_int()
        _check(lib().bwahip_fastq_next(self._h, chunk_bases, 1 if keep_comments else 0, C.byref(arr), C.byref(n)), "bwahip_fastq_next")
        return arr, n.value

    def next_batch(self, chunk_bases, keep_comments=False):
        """-> (batch handle, pointer to its bseq1_t array, n): an owned batch, valid until release_batch(handle); (None, None, 0) at the end."""
        h, arr, n = C.c_void_p(), C.POINTER(Seq)(), C.c_int()
        _check(lib().bwahip_fastq_next_batch(self._h, chunk_bases, 1 if keep_comments else 0, C.byref(h), C.byref(arr), C.byref(n)), "bwahip_fastq_next_batch")
        return (h, arr, n.value) if n.value else (None, None, 0)

    @staticmethod
    def release_batch(h):
        if h:
            lib().bwahip_fastq_batch_release(h)

    def close(self):
        if self._h:
            lib().bwahip_fastq_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def stream_run(ctxs, fq1, fq2=None, out_fd=-1, opt=None, chunk_bases=0, max_reads=0, keep_comments=False, reader_threads=0, pes0=None):
    """bwahip_stream_run: FASTQ files -> SAM text on out_fd over the given contexts (the library's own reader, workers and writer;
    no Python in the data path).  Returns the filled StreamStats."""
    opt = opt or default_opt()
    st = StreamStats()
    st.chunk_bases, st.max_reads, st.keep_comments, st.reader_threads = chunk_bases, max_reads, int(keep_comments), reader_threads
    arr = (C.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().bwahip_stream_run(arr, len(ctxs), C.byref(opt), C.byref(pes0) if pes0 is not None else None, os.fsencode(fq1),
                                   os.fsencode(fq2) if fq2 else None, out_fd, C.byref(st)), "bwahip_stream_run")
    return st


def _tool(name):
    path = os.path.join(TOOLS, name)
    if not os.path.exists(path):
        raise BwahipError(f"{path} is missing: run __graft_entry__.build()")
    return path


def make_genome(fa, seed, lens, repeats=True):
    subprocess.check_call([_tool("simgen"), "genome", fa, str(seed), "1" if repeats else "0"] + [str(x) for x in lens])


def make_reads(fa, fq1, fq2, n, length, sub_ppm, indel_ppm, n_ppm, seed, chim_ppm=0):
    subprocess.check_call([_tool("simgen"), "reads", fa, fq1, fq2 or "-", str(n), str(length), str(sub_ppm), str(indel_ppm),
                           str(n_ppm), str(seed), str(chim_ppm)])


def make_index(fa, prefix):
    subprocess.check_call([_tool("mkindex"), fa, prefix])


def read_fastq(path):
    names, seqs, quals = [], [], []
    with open(path, "rb") as f:
        while True:
            h = f.readline()
            if not h:
                break
            s = f.readline().rstrip(b"\r\n")
            f.readline()
            q = f.readline().rstrip(b"\r\n")
            names.append(h[1:].split()[0])
            seqs.append(s)
            quals.append(q)
    return names, seqs, quals
