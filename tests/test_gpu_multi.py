"""Multi-GPU path on the one-GPU box: (1) the RCCL index broadcast behind the C ABI (bwahip_init_rccl) executed for real with a
one-rank communicator; (2) two processes (gloo for coordination, both on cuda:0 -- RCCL refuses two ranks on one device) that
shard whole batches round-robin with their true n_processed, rank 1 on index arrays it received by broadcast
(bwahip_init_device); the reassembled SAM must equal the single-process CPU path's with the same -K, SE and PE."""
import os
import socket
import subprocess
import sys
import textwrap
import pytest
import common
from common import bw

pytestmark = pytest.mark.gpu


def test_rccl_broadcast_init_one_rank(small_index, tmp_path):
    fq = str(tmp_path / "r.fq")
    bw.make_reads(small_index["fa"], fq, None, 1500, 150, 10000, 2000, 500, 141, 20000)
    names, seqs, quals = bw.read_fastq(fq)
    want = subprocess.run([common.ORACLE, "mem", "-t", "4", small_index["prefix"], fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    uid = bw.Context.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    with bw.Context.from_rccl(small_index["prefix"], 0, 1, uid) as c:
        opt = bw.default_opt()
        opt.n_threads = 4
        assert b"".join(c.process_seqs(names, seqs, quals, opt)) == want


def test_rccl_broadcast_init_reports_a_missing_index_to_every_rank(tmp_path):
    """Rank 0 loads the index before the communicator exists and broadcasts its verdict first (a zero metadata length): a missing file set is
    BWAHIP_EIO on every rank, not a hang of the others in a broadcast that never comes (one-rank communicator here)."""
    uid = bw.Context.rccl_unique_id()
    with pytest.raises(bw.BwahipError):
        bw.Context.from_rccl(str(tmp_path / "no_such_index"), 0, 1, uid)


def test_two_contexts_sharing_the_index_run_batches_concurrently(small_index, tmp_path):
    """bwahip_ctx_clone: a second context on the same GPU (index arrays shared in HBM), both driven at once from two host threads
    (double buffering).  Every batch's SAM must equal the CPU path's whichever context took it, SE and PE."""
    import threading
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, 3000, 150, 10000, 2000, 500, 151, 20000)
    n1, s1, q1 = bw.read_fastq(fq1)
    n2, s2, q2 = bw.read_fastq(fq2)
    names = [x for p in zip(n1, n2) for x in p]; seqs = [x for p in zip(s1, s2) for x in p]; quals = [x for p in zip(q1, q2) for x in p]
    batch = 1000 * 150                                           # -K in bases: 6 batches of 1000 reads
    want_pe = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(batch), small_index["prefix"], fq1, fq2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    want_se = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(batch), small_index["prefix"], fq1], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    with bw.Context(small_index["prefix"]) as c0:
        c1 = c0.clone()
        try:
            for pe, nm, sq, ql, want in ((True, names, seqs, quals, want_pe), (False, n1, s1, q1, want_se)):
                opt = bw.default_opt()
                opt.n_threads = 2
                if pe:
                    opt.flag |= 0x2
                out = {}

                def work(t, cx):
                    for k, b0 in enumerate(range(0, len(sq), 1000)):
                        if k % 2 == t:
                            out[b0] = b"".join(cx.process_seqs(nm[b0:b0 + 1000], sq[b0:b0 + 1000], ql[b0:b0 + 1000], opt, n_processed=b0))
                th = [threading.Thread(target=work, args=(t, cx)) for t, cx in enumerate((c0, c1))]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                assert b"".join(out[k] for k in sorted(out)) == want
        finally:
            c1.close()


@pytest.mark.parametrize("n_ctx", [1, 2, 3])
def test_stream_driver_files_to_sam_over_several_contexts(small_index, tmp_path, n_ctx):
    """bwahip_stream_run -- the product's batch driver (the role of superBatchMain, cuda/superbatch_process.cpp:133): one reader, one worker
    per context (here clones on the one GPU; bwahip_ctx_clone_on gives the same on further devices), whole -K batches dealt with their
    true n_processed, SAM written in input order by the writer thread.  FASTQ (plain and gzip) in, file out: must equal the CPU path's
    output with the same -K, SE and PE, however many contexts share the work."""
    import gzip
    fq1, fq2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    bw.make_reads(small_index["fa"], fq1, fq2, 4300, 150, 10000, 2000, 500, 171, 20000)
    gz1 = fq1 + ".gz"
    with gzip.open(gz1, "wb", compresslevel=1) as g:
        g.write(open(fq1, "rb").read())
    K = 600 * 150                                                # 8 batches of 600 reads (PE: 4300 reads) / 4 (SE: 2150 reads)
    want_pe = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(K), small_index["prefix"], fq1, fq2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    want_se = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(K), small_index["prefix"], fq1], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    with bw.Context(small_index["prefix"]) as c0:
        ctxs = [c0] + [c0.clone_on(0) for _ in range(n_ctx - 1)]
        try:
            opt = bw.default_opt()
            opt.n_threads = 4
            for a, b, want in ((fq1, fq2, want_pe), (gz1, None, want_se)):
                out = str(tmp_path / "out.sam")
                fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                try:
                    st = bw.stream_run(ctxs, a, b, fd, opt, chunk_bases=K, reader_threads=2)
                finally:
                    os.close(fd)
                got = open(out, "rb").read()
                assert got == want
                assert st.n_reads == (4300 if b else 2150) and st.sam_bytes == len(want) and st.n_batches == (8 if b else 4)
            # max_reads stops at a batch boundary; a missing file is an error, not an empty output
            st = bw.stream_run(ctxs, fq1, fq2, -1, opt, chunk_bases=K, max_reads=1000)
            assert st.n_reads == 1200 and st.n_batches == 2
            with pytest.raises(bw.BwahipError):
                bw.stream_run(ctxs, str(tmp_path / "missing.fq"), None, -1, opt, chunk_bases=K)
        finally:
            for c in ctxs[1:]:
                c.close()


WORKER = textwrap.dedent('''
    import os, sys
    root, prefix, fq1, fq2, batch, out = sys.argv[1:7]
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "bwa-mem-gpu_amd"))
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    import common
    from common import bw
    import tools_py as tp
    if rank == 0:
        ctx, holder = bw.Context(prefix, 0), None
        tp.broadcast_index(bw, dist, torch, prefix, rank, 0)
    else:
        ctx, holder = tp.broadcast_index(bw, dist, torch, None, rank, 0)     # index arrays received, context on the adopted device arrays
    n1, s1, q1 = bw.read_fastq(fq1)
    pe = fq2 != "-"
    if pe:
        n2, s2, q2 = bw.read_fastq(fq2)
        names = [x for p in zip(n1, n2) for x in p]; seqs = [x for p in zip(s1, s2) for x in p]; quals = [x for p in zip(q1, q2) for x in p]
    else:
        names, seqs, quals = n1, s1, q1
    opt = bw.default_opt(); opt.n_threads = 4
    if pe: opt.flag |= 0x2
    def align(b0, b1, n_processed):
        return b"".join(ctx.process_seqs(names[b0:b1], seqs[b0:b1], quals[b0:b1], opt, n_processed=n_processed))
    sam = tp.align_sharded(dist, rank, world, len(seqs), int(batch), align)
    if rank == 0:
        open(out, "wb").write(sam)
    dist.barrier(); ctx.close(); dist.destroy_process_group()
    print("WORKER_OK", rank, flush=True)
''')


@pytest.mark.parametrize("pe", [False, True])
def test_two_ranks_shard_batches_and_reproduce_single_process_sam(small_index, tmp_path, pe):
    fq1, fq2 = str(tmp_path / "x_1.fq"), str(tmp_path / "x_2.fq")
    if pe:
        bw.make_reads(small_index["fa"], fq1, fq2, 3500, 150, 20000, 2000, 500, 151)
        batch, K = 1000, 150 * 1000                              # 1000 reads = 500 pairs per batch
        want = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(K), small_index["prefix"], fq1, fq2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    else:
        bw.make_reads(small_index["fa"], fq1, None, 5500, 150, 10000, 2000, 500, 152, 20000)
        batch, K = 1000, 150 * 1000
        want = subprocess.run([common.ORACLE, "mem", "-t", "4", "-K", str(K), small_index["prefix"], fq1], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    out = str(tmp_path / "out.sam")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", port,
           str(script), common.ROOT, small_index["prefix"], fq1, fq2 if pe else "-", str(batch), out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.count("WORKER_OK") == 2, r.stdout[-3000:]
    got = open(out, "rb").read()
    assert got == want
