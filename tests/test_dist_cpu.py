"""The N>1 path on CPU: two ranks over gloo rehearse the system's only collective (the start-up broadcast of
the index arrays from rank 0) and the per-rank read sharding of bench.py."""
import os
import subprocess
import sys
import textwrap
import numpy as np
import common
from common import bw

WORKER = textwrap.dedent('''
    import os, sys, hashlib, json
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "bwa-mem-gpu_amd"))
    import numpy as np, torch, torch.distributed as dist
    import tools_py as tp
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank = dist.get_rank()
    meta, tensors = tp.broadcast_index_arrays(dist, torch, sys.argv[2] if rank == 0 else None, rank, torch.device("cpu"))
    h = {k: hashlib.sha256(v.numpy().tobytes()).hexdigest() for k, v in tensors.items()}
    lens = [c[2] for c in meta["contigs"]]
    genome = tp.make_genome(11, lens, True)
    reads = tp.make_reads(genome, lens, 64, 100, seed=102 + rank)        # every rank its own shard
    out = {"rank": rank, "hash": h, "meta": {k: meta[k] for k in ("primary", "seq_len", "l_pac", "sizes")},
           "reads": hashlib.sha256(reads.tobytes()).hexdigest()}
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                               # bench.py's max-over-ranks timing
    out["max"] = float(t.item())
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier(); dist.destroy_process_group()
''')


def test_index_broadcast_and_sharding_world_size_2(small_index, tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    with socket.socket() as sk:                      # a port that is free right now (a fixed one can linger in TIME_WAIT)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script), common.ROOT, small_index["prefix"]]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l), key=lambda x: x["rank"])
    assert [x["rank"] for x in res] == [0, 1]
    assert res[0]["hash"] == res[1]["hash"] and res[0]["meta"] == res[1]["meta"]       # every rank holds the same index
    assert res[0]["reads"] != res[1]["reads"]                                           # but aligns its own reads
    assert res[0]["max"] == res[1]["max"] == 2.0
    import hashlib
    want = hashlib.sha256(np.fromfile(small_index["prefix"] + ".bwt", dtype=np.uint8)[40:].tobytes()).hexdigest()
    assert res[1]["hash"]["bwt"] == want


SHARD_WORKER = textwrap.dedent('''
    import os, sys, json, hashlib
    sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "bwa-mem-gpu_amd"))
    import torch, torch.distributed as dist
    import tools_py as tp
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    n, batch = 10007, 1000                                   # ragged last batch
    seen = []
    def align(b0, b1, n_processed):                          # stand-in for bwahip_process_seqs: records what it was asked to do
        seen.append((b0, b1, n_processed))
        return ("".join(f"read{i}@{n_processed}\\n" for i in range(b0, b1))).encode()
    sam = tp.align_sharded(dist, rank, world, n, batch, align)
    out = {"rank": rank, "seen": seen, "sam": hashlib.sha256(sam).hexdigest() if sam is not None else None, "n_lines": sam.count(b"\\n") if sam is not None else None}
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier(); dist.destroy_process_group()
''')


def test_round_robin_batches_keep_n_processed_and_input_order(tmp_path):
    """Two ranks take whole batches round-robin, each with its true n_processed; rank 0 reassembles the text in input order:
    the result must equal what one process produces."""
    import json, hashlib, socket
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, str(script), common.ROOT]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    res = sorted((json.loads(l.split("RESULT ", 1)[1]) for l in r.stdout.splitlines() if "RESULT " in l), key=lambda x: x["rank"])
    n, batch = 10007, 1000
    want = "".join(f"read{i}@{(i // batch) * batch}\n" for i in range(n)).encode()
    assert res[0]["sam"] == hashlib.sha256(want).hexdigest() and res[0]["n_lines"] == n and res[1]["sam"] is None
    assert [tuple(x) for x in res[0]["seen"]] == [(b, min(n, b + batch), b) for b in range(0, n, 2 * batch)]
    assert [tuple(x) for x in res[1]["seen"]] == [(b, min(n, b + batch), b) for b in range(batch, n, 2 * batch)]
