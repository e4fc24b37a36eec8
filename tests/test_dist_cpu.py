"""CPU test of the N>1 exchange step: world_size-2 gloo processes run ntjoin_amd.dist.gather_sketches and must
produce the rank-ordered concatenation with shifted record indices (what the union graph is built from)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import REPO


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local(rank):
    rng = np.random.default_rng(100 + rank)
    n = [5, 0, 9][rank % 3] if rank < 3 else 4
    nrec = rank + 2
    rec = np.sort(rng.integers(0, nrec, size=n)).astype(np.int32)
    return {"out_hash": torch.from_numpy(rng.integers(-2**62, 2**62, size=n, dtype=np.int64)),
            "pos": torch.from_numpy(rng.integers(0, 10**6, size=n).astype(np.int32)),
            "record": torch.from_numpy(rec),
            "forward": torch.from_numpy(rng.integers(0, 2, size=n).astype(np.uint8))}, nrec


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ntjoin_amd.dist import gather_sketches
    local, nrec = _local(rank)
    g = gather_sketches(local, nrec)
    q.put((rank, {k: (v.numpy().copy() if torch.is_tensor(v) else v) for k, v in g.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_sketches_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    locs = [_local(r) for r in range(world)]
    off = np.cumsum([0] + [nrec for _, nrec in locs])
    want = {k: np.concatenate([l[k].numpy() + (off[r] if k == "record" else 0) for r, (l, _) in enumerate(locs)])
            for k in ("out_hash", "pos", "record", "forward")}
    for r in range(world):
        g = results[r]
        for k in want:
            assert np.array_equal(g[k], want[k]), (r, k)
        assert g["n_records"] == off[-1] and g["record_offset"] == off[r]
        assert g["counts"] == [l["out_hash"].numel() for l, _ in locs]


def _ids_of(rank):
    rng = np.random.default_rng(7 + rank)
    n = [0, 5, 2000, 3][rank % 4]
    return [f"r{rank}:" + "".join(chr(int(c)) for c in rng.integers(33, 127, size=int(rng.integers(0, 40)))) + ("_é" if i % 7 == 0 else "")
            for i in range(n)]


def _worker_ids(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ntjoin_amd.dist import all_gather_strings
    got = all_gather_strings(_ids_of(rank), torch.device("cpu"))
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_record_ids_travel_as_length_prefixed_bytes(world):
    """the record ids of the exchange step (the only strings it moves): two tensor collectives instead of all_gather_object;
    empty lists, empty strings, non-ASCII ids, lists of very different sizes"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ids, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_ids_of(r) for r in range(world)]
    for r in range(world):
        assert results[r] == want


def test_shard_records_balanced():
    from ntjoin_amd.dist import shard_records
    rng = np.random.default_rng(0)
    lens = np.exp(rng.uniform(np.log(1e4), np.log(2e6), size=500)).astype(np.int64)
    for world in (1, 2, 4, 8):
        shards = shard_records(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(500))
        loads = [int(lens[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= lens.max()
        assert all(s == sorted(s) for s in shards)


def test_concat_tsv_parts(tmp_path):
    """rank-ordered TSV parts of a split load: a part whose first line continues the previous part's last record is
    appended to that line (with or without entries on either side); other lines pass through"""
    from ntjoin_amd.dist import concat_tsv_parts
    parts = [b"r0\t1:2:AC 3:4:GT\nr1\t5:6:AA\n",      # rank 0: r0 complete, r1 begun
             b"r1\t7:8:CC\n",                            # rank 1: continues r1, nothing else
             b"r1\t\nr2\t\n",                           # rank 2: continues r1 without entries, begins r2 without entries
             b"",                                          # rank 3: holds nothing
             b"r2\t9:10:GG\nr3\t11:12:TT\n",            # rank 4: continues r2 (first entries of that line), r3
             b"r4\t13:14:AT\n"]                          # rank 5: a new record, no continuation
    cont = [False, True, True, False, True, False]
    paths = []
    for i, data in enumerate(parts):
        path = tmp_path / f"p{i}"
        path.write_bytes(data)
        paths.append(str(path))
    out = tmp_path / "joined.tsv"
    concat_tsv_parts(paths, cont, str(out))
    assert out.read_bytes() == (b"r0\t1:2:AC 3:4:GT\nr1\t5:6:AA 7:8:CC\nr2\t9:10:GG\nr3\t11:12:TT\nr4\t13:14:AT\n")
