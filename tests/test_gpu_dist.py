"""GPU test of the N>1 code path on one GPU: a 1-rank RCCL ("nccl") group runs the real all-gather + union-graph
path on device tensors; its graph must equal the graph built directly.  (World sizes 2/3 of the exchange logic are
covered on CPU with gloo in test_dist_cpu.py; the driver runs 2/4/8 GPUs.)"""
import os
import socket

import numpy as np
import pytest

from tests.conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu
FASTA = os.path.join(GOLDEN, "fasta")


def test_union_graph_single_rank_nccl():
    import torch
    import torch.distributed as dist
    from ntjoin_amd.dist import allgather_union_graph
    from ntjoin_amd.engine import MxEngine
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        meta = load_case("synth3_w50")["meta"]
        asms = meta["refs"] + [meta["target"]]
        with MxEngine(k=meta["k"], w=meta["w"]) as eng:
            for a in asms:
                eng.add_fasta(a["tsv"], a["weight"], os.path.join(FASTA, a["fasta"]))
            eng.sketch()
            union = allgather_union_graph(eng, meta["k"], meta["w"], 0)
            union2 = allgather_union_graph(eng, meta["k"], meta["w"], 0, union)  # reuse of the union handle
            assert union2 is union
            eng.build_graph()
            g0, g1 = eng.get_graph(), union.get_graph()
            for key in g0:
                assert np.array_equal(np.asarray(g0[key]), np.asarray(g1[key])), key
            union_flags = [union.get_mx_flags(a).copy() for a in range(len(asms))]
            for a in range(len(asms)):
                assert np.array_equal(eng.get_mx_flags(a), union_flags[a])
                assert union.record_ids(a, union.n_records(a)) == [f"r0:{x}" for x in eng.record_ids(a, eng.n_records(a))]
            union.close()
            # the hash-partitioned graph stage over RCCL (one rank: every all-to-all is a copy, the code path is whole)
            from ntjoin_amd.dist import partitioned_graph, partitioned_totals
            owner = partitioned_graph(eng, meta["k"], meta["w"], 0)
            owner = partitioned_graph(eng, meta["k"], meta["w"], 0, owner)
            partitioned_totals(owner)
            g2 = owner.get_graph()
            assert owner.dg["base"] == 0 and owner.dg["vertices"] == len(g0["vertex_hash"]) and owner.dg["edges"] == len(g0["edge_u"])
            e0 = {(int(g0["vertex_hash"][u]), int(g0["vertex_hash"][v])): (int(s_), float(w_)) for u, v, s_, w_ in
                  zip(g0["edge_u"], g0["edge_v"], g0["edge_support"], g0["edge_weight"])}
            e2 = {(int(g2["vertex_hash"][u]), int(g2["vertex_hash"][v])): (int(s_), float(w_)) for u, v, s_, w_ in
                  zip(g2["edge_u"], g2["edge_v"], g2["edge_support"], g2["edge_weight"])}
            assert e0 == e2 and sorted(g0["vertex_hash"].tolist()) == sorted(g2["vertex_hash"].tolist())
            for a in range(len(asms)):
                assert np.array_equal(eng.get_mx_flags(a), union_flags[a])
            owner.close()
    finally:
        dist.destroy_process_group()


def test_run_dist_single_rank(tmp_path):
    """the torchrun-able FASTA driver with one rank: TSVs byte-identical to the committed sketches, canonical
    .mx.dot identical to the reference's (sharding with world=1 is the identity; world>1 differs only in who holds
    which contiguous record range, covered by test_shard_range_partitions_records + the gloo exchange tests)"""
    import shutil
    import subprocess
    import sys
    from oracle import graph_oracle as go
    from tests.conftest import REPO
    meta = load_case("synth3_w50")["meta"]
    asms = meta["refs"] + [meta["target"]]
    for a in asms:
        shutil.copy(os.path.join(FASTA, a["fasta"]), tmp_path / a["fasta"])
    env = dict(os.environ, PYTHONPATH=REPO, MASTER_PORT="29577")
    cmd = [sys.executable, "-m", "ntjoin_amd.run_dist", "-k", str(meta["k"]), "-w", str(meta["w"]), "-p", "out",
           "--target", meta["target"]["fasta"], "--target_weight", str(meta["target"]["weight"]),
           "--references"] + [a["fasta"] for a in meta["refs"]] + ["--reference_weights"] + \
          [str(a["weight"]) for a in meta["refs"]]
    subprocess.check_call(cmd, cwd=tmp_path, env=env)
    import filecmp
    for a in asms:
        assert filecmp.cmp(str(tmp_path / a["tsv"]), os.path.join(GOLDEN, "cases", meta["name"], a["tsv"]), shallow=False)
    with open(os.path.join(GOLDEN, "cases", meta["name"], "reference.mx.dot"), encoding="utf-8") as fh:
        want = go.canonical_dot_from_text(fh.read())
    assert go.canonical_dot_from_text((tmp_path / "out.mx.dot").read_text(encoding="utf-8")) == want


def test_sharded_load_equals_full_load():
    """add_fasta_shard(s, n) for all s, concatenated in shard order, equals add_fasta (sketch arrays, global records)"""
    from ntjoin_amd.engine import MxEngine
    fa = os.path.join(FASTA, "scaf.more_seqs.fa")
    with MxEngine(k=32, w=100) as eng:
        eng.add_fasta("x", 1.0, fa)
        eng.sketch()
        full = eng.get_sketch(0)
    for world in (2, 3):
        parts = []
        covered = []
        for s in range(world):
            with MxEngine(k=32, w=100) as eng:
                eng.add_fasta_shard("x", 1.0, fa, s, world)
                eng.sketch()
                sk = eng.get_sketch(0)
                lo, hi = eng.assembly_shard(0)
                covered.append((lo, hi))
                assert sk["record_ids"] == full["record_ids"]
                assert all(lo <= r < hi for r in sk["record"].tolist())
                parts.append(sk)
        assert covered[0][0] == 0 and covered[-1][1] == len(full["record_ids"])
        for key in ("out_hash", "pos", "record", "forward"):
            assert np.array_equal(np.concatenate([p[key] for p in parts]), full[key]), (world, key)


def test_pack_unpack_two_ranks_on_one_gpu():
    """the exchange kernels with world=2 without a second GPU: two engines play the ranks (disjoint record sets of
    every assembly), their packed buffers are laid side by side as an all-gather would, and mxg_set_sketch_gathered
    unpacks them (rank order, record indices shifted) into a union engine; the union graph must equal the graph of
    one engine that holds everything."""
    import torch
    from ntjoin_amd.engine import MxEngine
    from tests import _oracle
    meta = load_case("synth3_w50")["meta"]
    asms = meta["refs"] + [meta["target"]]
    k, w = meta["k"], meta["w"]
    all_recs = [_oracle.read_fasta(os.path.join(FASTA, a["fasta"])) for a in asms]
    with MxEngine(k=k, w=w) as full, MxEngine(k=k, w=w) as r0, MxEngine(k=k, w=w) as r1, MxEngine(k=k, w=w) as union:
        splits = []
        for a, recs in zip(asms, all_recs):
            cut = max(1, len(recs) // 2)
            splits.append(cut)
            full.add_records(a["tsv"], a["weight"], recs)
            r0.add_records(a["tsv"], a["weight"], recs[:cut])
            r1.add_records(a["tsv"], a["weight"], recs[cut:])
            union.add_minimizers(a["tsv"], a["weight"], np.zeros(0, np.uint64), np.zeros(0, np.uint32),
                                 np.zeros(0, np.uint32), [rid for rid, _ in recs])
        for e in (full, r0, r1):
            e.sketch()
        for a in range(len(asms)):
            counts = np.array([r0.sketch_size(a), r1.sketch_size(a)], dtype=np.uint64)
            nmax = (max(int(counts.max()), 1) + 7) // 8 * 8
            recv = torch.zeros(2 * 16 * nmax, dtype=torch.uint8, device="cuda")
            r0.pack_sketch_device(a, recv[:16 * nmax].data_ptr(), nmax)
            r1.pack_sketch_device(a, recv[16 * nmax:].data_ptr(), nmax)
            union.set_sketch_gathered(a, recv.data_ptr(), nmax, counts, np.array([0, splits[a]], dtype=np.uint64))
        full.build_graph()
        union.build_graph()
        for a in range(len(asms)):
            s0, s1 = full.get_sketch(a), union.get_sketch(a)
            for key in ("out_hash", "pos", "record", "record_first"):
                assert np.array_equal(s0[key], s1[key]), (a, key)
            assert np.array_equal(full.get_mx_flags(a), union.get_mx_flags(a))
        g0, g1 = full.get_graph(), union.get_graph()
        for key in g0:
            assert np.array_equal(np.asarray(g0[key]), np.asarray(g1[key])), key
