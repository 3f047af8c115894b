"""Python-side tooling for bench.py and the tests: synthetic genomes / reads through libsimgen (ctypes),
FASTA/FASTQ writers, and the one collective of the system -- the start-up broadcast of the index arrays
(bwt, sa, pac) from rank 0 over torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_sg = None


def _simgen():
    global _sg
    if _sg is None:
        path = os.path.join(_HERE, "tools", "libsimgen.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build()")
        L = C.CDLL(path)
        L.simgen_random_bases.argtypes = [C.c_uint64, C.c_int64, C.c_void_p]
        L.simgen_add_repeats.argtypes = [C.c_uint64, C.c_int64, C.c_void_p]
        L.simgen_add_human_repeats.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_void_p]
        L.simgen_reads.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]
        _sg = L
    return _sg


def contig_lengths(total):
    """Split `total` bases into contigs of at most 150 Mbp (contig length is an int32 in the index)."""
    n = max(1, -(-total // 150000000))
    base = total // n
    return [base + (1 if i < total - base * n else 0) for i in range(n)]


def make_genome(seed, lens, repeats=True, profile="default"):
    """Same bytes as `simgen genome <fa> seed repeat_mode lens...` writes (ASCII ACGTN, contigs concatenated).  profile "human-like":
    genome-wide Alu/L1-like families of 10^4..10^6 copies at 5-15 % divergence and satellite arrays (simgen_add_human_repeats)."""
    L = _simgen()
    g = np.empty(int(sum(lens)), dtype=np.uint8)
    o = 0
    for c, ln in enumerate(lens):
        view = g[o:o + ln]
        L.simgen_random_bases(seed + 1000003 * c, ln, view.ctypes.data)
        if repeats and profile == "human-like":
            L.simgen_add_human_repeats(seed, seed + 7919 * c, ln, view.ctypes.data)
        elif repeats:
            L.simgen_add_repeats(seed + 7919 * c, ln, view.ctypes.data)
        o += ln
    return g


def write_fasta(path, genome, lens, width=60):
    with open(path, "wb") as f:
        o = 0
        for c, ln in enumerate(lens):
            f.write(b">ctg%d\n" % (c + 1))
            seq = genome[o:o + ln]
            full = (ln // width) * width
            if full:
                block = np.empty((full // width, width + 1), dtype=np.uint8)
                block[:, :width] = seq[:full].reshape(-1, width)
                block[:, width] = 10
                block.tofile(f)
            if ln > full:
                f.write(seq[full:].tobytes() + b"\n")
            o += ln


def make_reads(genome, lens, n, read_len, sub_ppm=10000, indel_ppm=0, n_ppm=0, chim_ppm=0, seed=102, paired=False):
    """n reads of read_len bases (ASCII), uniform starts, 50 % reverse strand, substitutions to a different base."""
    L = _simgen()
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lens, dtype=np.int64), out=off[1:])
    out = np.empty((n, read_len), dtype=np.uint8)
    L.simgen_reads(seed, genome.ctypes.data, len(lens), off.ctypes.data, n, read_len, sub_ppm, indel_ppm, n_ppm, chim_ppm,
                   1 if paired else 0, out.ctypes.data)
    return out


def write_fastq(path, reads, paired_second=None, qual=b"I"):
    n, rl = reads.shape
    with open(path, "wb") as f:
        q = qual * rl
        for i in range(n):
            f.write(b"@r%d\n" % i)
            f.write(reads[i].tobytes())
            f.write(b"\n+\n" + q + b"\n")


def read_fasta_bases(path, lens, width=60):
    """The bases of a FASTA written by write_fasta (headers and newlines stripped), without parsing line by line."""
    raw = np.fromfile(path, dtype=np.uint8)
    out = np.empty(int(sum(lens)), dtype=np.uint8)
    o = p = 0
    for c, ln in enumerate(lens):
        p += len(b">ctg%d\n" % (c + 1))
        full = (ln // width) * width
        if full:
            out[o:o + full] = raw[p:p + (full // width) * (width + 1)].reshape(-1, width + 1)[:, :width].reshape(-1)
            p += (full // width) * (width + 1)
        if ln > full:
            out[o + full:o + ln] = raw[p:p + ln - full]
            p += ln - full + 1
        o += ln
    return out


def fixed_names(n, paired):
    """(n, 9) uint8: NUL-terminated fixed-width names r0000000..; mates of a pair share their name (bwamem_pair.c:386)."""
    idx = np.arange(n, dtype=np.int64) >> (1 if paired else 0)
    a = np.empty((n, 9), dtype=np.uint8)
    a[:, 0] = ord("r")
    for k in range(7):
        a[:, 7 - k] = (idx // 10 ** k) % 10 + 48
    a[:, 8] = 0
    return a


def write_fastq_fixed(prefix, reads, paired, qual=b"I"):
    """FASTQ with the names of fixed_names(); PE: <prefix>_1.fq / _2.fq (even / odd rows).  Vectorised: fixed-size records."""
    n, rl = reads.shape
    names = fixed_names(n, paired)[:, :8]
    outs = []
    for m, sel in ([(1, slice(0, n, 2)), (2, slice(1, n, 2))] if paired else [(0, slice(0, n))]):
        sub, nm = reads[sel], names[sel]
        rec = np.empty((len(sub), 1 + 8 + 1 + rl + 3 + rl + 1), dtype=np.uint8)
        rec[:, 0] = ord("@"); rec[:, 1:9] = nm; rec[:, 9] = 10
        rec[:, 10:10 + rl] = sub
        rec[:, 10 + rl:13 + rl] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 13 + rl:13 + 2 * rl] = qual[0]
        rec[:, 13 + 2 * rl] = 10
        fn = f"{prefix}_{m}.fq" if paired else f"{prefix}.fq"
        rec.tofile(fn)
        outs.append(fn)
    return outs


SEQ_DTYPE = np.dtype([("l_seq", "<i4"), ("id", "<i4"), ("name", "<u8"), ("comment", "<u8"), ("seq", "<u8"), ("qual", "<u8"), ("sam", "<u8"),
                      ("l_name", "i1"), ("l_comment", "i1"), ("l_qual", "<i2"), ("pad", "<i4")])   # bseq1_t, bwa.h:58-63 (56 bytes)


def process_seqs_bulk(bw, ctx, opt, names, reads, qual=b"I", n_processed=0, pes0=None):
    """bwahip_process_seqs (== mem_process_seqs) on a big batch without per-read Python work: the bseq1_t array is laid out
    with numpy.  names: (n, w) NUL-terminated rows; reads: (n, rl) ASCII.  Returns (seconds inside the call, SAM bytes)."""
    import ctypes as C
    import time
    big_bytes = bw.big_bytes
    assert SEQ_DTYPE.itemsize == C.sizeof(bw.Seq) == 56
    n, rl = reads.shape
    seqbuf = np.ascontiguousarray(reads).copy()             # converted to 0..4 codes in place, as the reference does
    names = np.ascontiguousarray(names)
    qbuf = np.frombuffer(qual * rl + b"\0", dtype=np.uint8).copy()
    arr = np.zeros(n, dtype=SEQ_DTYPE)
    arr["l_seq"] = rl
    arr["id"] = np.arange(n)
    arr["name"] = names.ctypes.data + np.arange(n, dtype=np.uint64) * names.shape[1]
    arr["seq"] = seqbuf.ctypes.data + np.arange(n, dtype=np.uint64) * rl
    arr["qual"] = qbuf.ctypes.data
    L = bw.lib()
    t0 = time.time()
    rc = L.bwahip_process_seqs(ctx._h, C.byref(opt), n_processed, n, C.cast(arr.ctypes.data, C.POINTER(bw.Seq)), pes0)
    dt = time.time() - t0
    if rc != 0:
        raise bw.BwahipError(f"bwahip_process_seqs failed: {bw.ERRORS.get(rc, rc)}")
    out, ln = C.c_void_p(), C.c_int64()
    rc = L.bwahip_seqs_take_sam(C.cast(arr.ctypes.data, C.POINTER(bw.Seq)), n, C.byref(out), C.byref(ln))
    if rc != 0:
        raise bw.BwahipError(f"bwahip_seqs_take_sam failed: {bw.ERRORS.get(rc, rc)}")
    sam = big_bytes(out, ln.value)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(out)
    return dt, sam


def bulk_caller(bw, ctx, opt, names, reads, qual=b"I", one_piece=False):
    """A prepared bseq1_t array + a function that pushes it through bwahip_process_seqs once and returns (seconds, SAM length, crc32 of the
    SAM): the caller's side of a batch without any per-batch Python work, for timing several callers at once."""
    import ctypes as C
    import time
    import zlib
    big_bytes = bw.big_bytes
    n, rl = reads.shape
    pristine = np.ascontiguousarray(reads)
    seqbuf = pristine.copy()
    names = np.ascontiguousarray(names)
    qbuf = np.frombuffer(qual * rl + b"\0", dtype=np.uint8).copy()
    arr = np.zeros(n, dtype=SEQ_DTYPE)
    arr["l_seq"] = rl
    arr["id"] = np.arange(n)
    arr["name"] = names.ctypes.data + np.arange(n, dtype=np.uint64) * names.shape[1]
    arr["seq"] = seqbuf.ctypes.data + np.arange(n, dtype=np.uint64) * rl
    arr["qual"] = qbuf.ctypes.data
    L = bw.lib()
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    keep = (pristine, seqbuf, names, qbuf, arr)

    def once(check=False):
        np.copyto(seqbuf, pristine)                            # the library turns the bases into codes in place
        t0 = time.time()
        if one_piece:                                          # bwahip_process_seqs_text: the SAM stays in the context's buffer
            sam, ln = C.c_char_p(), C.c_int64()
            rc = L.bwahip_process_seqs_text(ctx._h, C.byref(opt), 0, n, C.cast(arr.ctypes.data, C.POINTER(bw.Seq)), None, C.byref(sam), C.byref(ln), None)
            dt = time.time() - t0
            if rc != 0:
                raise bw.BwahipError(f"bwahip_process_seqs_text failed: {bw.ERRORS.get(rc, rc)}")
            crc = zlib.crc32(big_bytes(sam, ln.value)) if check else None
            return dt, ln.value, crc
        rc = L.bwahip_process_seqs(ctx._h, C.byref(opt), 0, n, C.cast(arr.ctypes.data, C.POINTER(bw.Seq)), None)
        if rc != 0:
            raise bw.BwahipError(f"bwahip_process_seqs failed: {bw.ERRORS.get(rc, rc)}")
        out, ln = C.c_void_p(), C.c_int64()
        rc = L.bwahip_seqs_take_sam(C.cast(arr.ctypes.data, C.POINTER(bw.Seq)), n, C.byref(out), C.byref(ln))
        dt = time.time() - t0
        crc = zlib.crc32((C.c_char * ln.value).from_address(out.value)) if check else None
        libc.free(out)
        return dt, ln.value, crc
    once.keep = keep
    return once


# ------------------------------------------------------------------------------------------ index broadcast
def load_index_arrays(prefix):
    """Read a stock index file set into numpy arrays + metadata (format: bwt.c:385-462, bntseq.c:65-211)."""
    raw = np.fromfile(prefix + ".bwt", dtype=np.uint8)
    hdr = raw[:40].view(np.uint64)
    bwt = raw[40:]
    sraw = np.fromfile(prefix + ".sa", dtype=np.uint64)
    sa_intv, seq_len = int(sraw[5]), int(sraw[6])
    sa = np.concatenate([np.array([2**64 - 1], dtype=np.uint64), sraw[7:]])
    with open(prefix + ".ann") as f:
        l_pac, n_seqs, _seed = f.readline().split()
        contigs = []
        for _ in range(int(n_seqs)):
            name = f.readline().split()[1]
            o, ln, _ = f.readline().split()
            contigs.append([name, int(o), int(ln), 0])
    alt = set()
    if os.path.exists(prefix + ".alt"):
        with open(prefix + ".alt") as f:
            alt = {ln.split("\t")[0].strip() for ln in f if ln and ln[0] != "@"}
    for c in contigs:
        c[3] = 1 if c[0] in alt else 0
    pac = np.fromfile(prefix + ".pac", dtype=np.uint8)[:int(l_pac) // 4 + 1]
    meta = {"primary": int(hdr[0]), "L2": [0] + [int(x) for x in hdr[1:5]], "seq_len": seq_len, "sa_intv": sa_intv,
            "n_sa": int(len(sa)), "bwt_words": int(len(bwt) // 4), "l_pac": int(l_pac), "contigs": contigs,
            "sizes": [int(bwt.nbytes), int(sa.nbytes), int(pac.nbytes)]}
    return meta, {"bwt": bwt, "sa": sa.view(np.uint8), "pac": pac}


def broadcast_index_arrays(dist, torch, prefix, rank, device):
    """The system's only collective: rank 0 sends the three index arrays, everybody returns (meta, tensors)."""
    box = [None]
    arrays = None
    if rank == 0:
        meta, arrays = load_index_arrays(prefix)
        box[0] = meta
    dist.broadcast_object_list(box, src=0)
    meta = box[0]
    tensors = {}
    for name, size in zip(("bwt", "sa", "pac"), meta["sizes"]):
        if rank == 0:
            t = torch.from_numpy(arrays[name]).to(device)
        else:
            t = torch.empty(size, dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        tensors[name] = t
    return meta, tensors


def broadcast_index(bw, dist, torch, prefix, rank, local_rank):
    """Rank 0 (which already holds its own context) only sends; the others build a context on the received
    device arrays (bwahip_init_device) and must keep the returned tensors alive."""
    meta, tensors = broadcast_index_arrays(dist, torch, prefix, rank, torch.device(f"cuda:{local_rank}"))
    if rank == 0:
        return None
    torch.cuda.synchronize()
    ctx = bw.Context.from_device_arrays(meta, tensors["bwt"].data_ptr(), tensors["sa"].data_ptr(), tensors["pac"].data_ptr(), local_rank)
    return ctx, tensors


# ------------------------------------------------------------------------------------------ read sharding over ranks
def shard_batches(n_reads, batch_reads, rank, world):
    """Whole mem_process_seqs batches round-robin over the ranks: [(batch index, first read, end read)] of `rank`.
    A batch keeps its true n_processed (= first read), so hash tie-breaks and per-batch insert-size statistics are those of
    a single-process run with the same -K."""
    out = []
    for b, b0 in enumerate(range(0, n_reads, batch_reads)):
        if b % world == rank:
            out.append((b, b0, min(n_reads, b0 + batch_reads)))
    return out


def align_sharded(dist, rank, world, n_reads, batch_reads, align_batch):
    """Every rank runs align_batch(first, end, n_processed) -> bytes on its batches; rank 0 returns the SAM of all batches in
    input order (the others return None).  No data-path collective: only the final gather of the text."""
    mine = [(b, align_batch(b0, b1, b0)) for b, b0, b1 in shard_batches(n_reads, batch_reads, rank, world)]
    if world == 1:
        return b"".join(t for _, t in sorted(mine))
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    allb = sorted(x for part in gathered for x in part)
    return b"".join(t for _, t in allb)
