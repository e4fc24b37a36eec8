"""worker of tests/test_gpu_dist2.py: one of WORLD_SIZE processes that share cuda:0 and talk over gloo (RCCL cannot put two
ranks on one GPU; the exchange code is the same).  Every rank owns a slice of the records of every assembly; rank 0
checks the union graph against a single handle holding all records."""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntjoin_amd.dist import allgather_union_graph, partitioned_graph, partitioned_totals, sketch_union_graph  # noqa: E402
from ntjoin_amd.engine import MxEngine  # noqa: E402


def records(seed, n, genome=90000):
    rng0 = random.Random(0)
    base = "".join(rng0.choice("ACGT") for _ in range(genome))   # the genome every assembly is a copy of
    rng = random.Random(seed)
    cuts = sorted(rng.sample(range(2000, len(base) - 2000), n - 1))          # its own contig boundaries
    out = []
    for r, (lo, hi) in enumerate(zip([0] + cuts, cuts + [len(base)])):
        s = list(base[lo:hi])
        for _ in range(len(s) // 300):          # a few substitutions: assemblies share most minimizers, not all
            s[rng.randrange(len(s))] = rng.choice("ACGT")
        if r % 3 == 1:                          # some contigs reverse-complemented
            s = ["TGCA"["ACGT".index(c)] for c in reversed(s)]
        out.append((f"s{seed}_{r}", "".join(s)))
    return out


def records_np(seed, n, genome):
    """the same construction with numpy (sizes at which the sketches take the one-batch device route: Mbp per rank)"""
    import numpy as np
    base = np.random.default_rng(0).integers(0, 4, genome).astype(np.uint8)
    rng = np.random.default_rng(seed)
    cuts = np.sort(rng.choice(np.arange(2000, genome - 2000), n - 1, replace=False)).tolist()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for r, (lo, hi) in enumerate(zip([0] + cuts, cuts + [genome])):
        s_ = base[lo:hi].copy()
        at = rng.integers(0, len(s_), len(s_) // 300)
        s_[at] = rng.integers(0, 4, len(at))
        if r % 3 == 1:
            s_ = (3 - s_[::-1]).astype(np.uint8)
        out.append((f"s{seed}_{r}", acgt[s_].tobytes().decode()))
    return out


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k, w = 32, 50
    asms = [("refA", 2.0, records(1, 9)), ("refB", 1.5, records(2, 7)), ("tgt", 1.0, records(3, 11))]
    if os.environ.get("MXG_TEST_CONFIG3") == "1":   # BASELINE configs[3]'s shape: target + 3 references, w=500, weights 1/2/2/2
        w = 500
        asms = [("ref1", 2.0, records(11, 17, 700000)), ("ref2", 2.0, records(12, 13, 700000)), ("ref3", 2.0, records(13, 19, 700000)),
                ("tgt", 1.0, records(14, 41, 700000))]
    if os.environ.get("MXG_TEST_CONFIG3") == "2":   # configs[2]'s shape at 6 Mbp per assembly: every rank's sketches are packed on the device
        w = 1000
        asms = [("refA", 2.0, records_np(1, 9, 6_000_000)), ("refB", 1.5, records_np(2, 7, 6_000_000)), ("tgt", 1.0, records_np(3, 11, 6_000_000))]
    use_stream = os.environ.get("MXG_TEST_STREAM") == "1"
    xs = torch.cuda.Stream() if use_stream else None
    kw = {"cand_per_window": int(os.environ["MXG_TEST_CAND"])} if os.environ.get("MXG_TEST_CAND") else {}
    eng = MxEngine(k=k, w=w, device=0, stream=xs.cuda_stream if xs is not None else None, **kw)
    mine = []
    for name, wt, recs in asms:
        part = [r for i, r in enumerate(recs) if i * world // len(recs) == rank]   # contiguous slices, rank order
        mine.append(part)
        eng.add_records(name, wt, part)
    union = None
    for _step in range(3):                      # later steps reuse the union handle and its fixed-capacity slots
        if _step == 1:
            eng.sketch(-2)                      # the two-call form ...
            union = allgather_union_graph(eng, k, w, 0, union, stream=xs)
        else:                                   # ... and the one-call form (no host sync between sketch and exchange when on a stream)
            union = sketch_union_graph(eng, k, w, 0, union, stream=xs)
    assert all(eng.sketch_size(a) > 0 for a in range(len(asms)) if mine[a])   # (8 ranks: some own no record of an assembly)
    # the same graph, distributed by hash range: every rank ends up with its own vertices and edges
    owner = None
    for _step in range(3):                       # exact exchange, then twice with the fixed-capacity slots
        if os.environ.get("MXG_TEST_DG_SKETCH", "1") == "1":   # the call sketches too (on a stream and with slots: every assembly's
            owner = partitioned_graph(eng, k, w, 0, owner, stream=xs, sketch=True)   # items leave while the next one is sketched)
        else:
            eng.sketch(-2)
            owner = partitioned_graph(eng, k, w, 0, owner, stream=xs)
    os.write(1, f"slots in use: {owner._slots is not None}\n".encode())
    os.write(1, f"ITEMS rank {rank}: {'behind the sketches' if getattr(owner, '_comm', None) is None or getattr(owner, '_no_overlap', False) else 'beside the sketches'}\n".encode())
    partitioned_totals(owner)
    pg = owner.get_graph()
    part = {"base": owner.dg["base"], "vhash": pg["vertex_hash"].tolist(),
            "vpos": pg["vertex_pos"].tolist(), "vrec": pg["vertex_record"].tolist(),
            "edges": list(zip(pg["edge_u"].tolist(), pg["edge_v"].tolist(), pg["edge_support"].tolist(), pg["edge_weight"].tolist())),
            "rank": rank, "flags": [eng.get_mx_flags(a).tolist() for a in range(len(asms))], "totals": (owner.dg["vertices"], owner.dg["edges"])}
    parts = [None] * world
    dist.all_gather_object(parts, part)
    ok = True
    if rank == 0:
        with MxEngine(k=k, w=w, device=0) as whole:
            for name, wt, recs in asms:
                ids = []
                for r in range(world):
                    ids += [(f"r{r}:{rid}", seq) for i, (rid, seq) in enumerate(recs) if i * world // len(recs) == r]
                whole.add_records(name, wt, ids)
            whole.sketch()
            whole.build_graph()
            g0, g1 = whole.get_graph(), union.get_graph()
            for key in g0:
                if not np.array_equal(np.asarray(g0[key]), np.asarray(g1[key])):
                    ok = False
                    os.write(1, f"MISMATCH {key}\n".encode())
            for a in range(len(asms)):
                if not np.array_equal(whole.get_mx_flags(a), union.get_mx_flags(a)):
                    ok = False
                    os.write(1, f"MISMATCH flags {a}\n".encode())
                if whole.record_ids(a, whole.n_records(a)) != union.record_ids(a, union.n_records(a)):
                    ok = False
                    os.write(1, f"MISMATCH ids {a}\n".encode())
            # --- partitioned graph against the same single handle ---
            parts.sort(key=lambda p: p["base"])
            ghash = [h for p in parts for h in p["vhash"]]
            names0 = g0["vertex_hash"].tolist()
            want_e = {(names0[u], names0[v]): (s_, w_) for u, v, s_, w_ in
                      zip(g0["edge_u"].tolist(), g0["edge_v"].tolist(), g0["edge_support"].tolist(), g0["edge_weight"].tolist())}
            got_e = {}
            for p in parts:
                for u, v, s_, w_ in p["edges"]:
                    got_e[(p["vhash"][u], ghash[v])] = (s_, w_)
            n_e = sum(len(p["edges"]) for p in parts)
            want_v = {h: (tuple(int(g0["vertex_pos"][a][i]) for a in range(len(asms))),
                          tuple(int(g0["vertex_record"][a][i]) for a in range(len(asms)))) for i, h in enumerate(names0)}
            got_v = {h: (tuple(p["vpos"][a][i] for a in range(len(asms))), tuple(p["vrec"][a][i] for a in range(len(asms))))
                     for p in parts for i, h in enumerate(p["vhash"])}
            if sorted(ghash) != sorted(names0) or got_v != want_v:
                ok = False
                os.write(1, f"MISMATCH partitioned vertices {len(ghash)} vs {len(names0)}\n".encode())
            if got_e != want_e or n_e != len(want_e):
                ok = False
                os.write(1, f"MISMATCH partitioned edges {n_e} / {len(got_e)} vs {len(want_e)}\n".encode())
                bad = [(k_, got_e.get(k_), want_e.get(k_)) for k_ in list(want_e) if got_e.get(k_) != want_e[k_]][:5]
                extra = [(k_, got_e[k_]) for k_ in got_e if k_ not in want_e][:5]
                os.write(1, f"  differing {bad}\n  extra {extra}\n".encode())
            if any(p["totals"] != (len(names0), len(want_e)) for p in parts):
                ok = False
                os.write(1, f"MISMATCH partitioned totals {[p['totals'] for p in parts]}\n".encode())
            byrank = sorted(parts, key=lambda p: p["base"])
            for a in range(len(asms)):
                cat = [f for p in sorted(parts, key=lambda q: q["rank"]) for f in p["flags"][a]]
                if cat != whole.get_mx_flags(a).tolist():
                    ok = False
                    os.write(1, f"MISMATCH partitioned flags {a}\n".encode())
            os.write(1, f"union: {len(g1['vertex_hash'])} vertices, {len(g1['edge_u'])} edges; sketches "
                        f"{[union.sketch_size(a) for a in range(len(asms))]} whole {[whole.sketch_size(a) for a in range(len(asms))]}\n".encode())
            if len(g1["vertex_hash"]) < 100 or len(g1["edge_u"]) < 50:
                ok = False
                os.write(1, b"graph too small to mean anything\n")
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    sl = getattr(union, "_slots", None)
    os.write(1, f"EXCHANGE rank {rank}: {'none' if sl is None else 'per-assembly' if 'send_parts' in sl else 'one all-gather'}\n".encode())
    union.close()
    owner.close()
    eng.close()
    dist.destroy_process_group()
    os.write(1, f"DIST2 {'OK' if int(flag) == 1 else 'FAILED'} rank {rank}\n".encode())  # one write: no interleaving
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
