"""worker of tests/test_gpu_dist2.py::test_partitioned_graph_with_records_cut_between_ranks: WORLD_SIZE processes share cuda:0 over
gloo; every rank holds an EQUAL BASE RANGE of every assembly (records cut at the borders travel as pieces with a halo,
mxg_add_assembly_fasta_split), both exchanges run, and the graph must be that of one handle holding the whole files -- in
particular no edge may be lost where a record is cut (the last shared minimizer of a rank travels to the ranks after it)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntjoin_amd.dist import partitioned_graph, partitioned_totals, sketch_union_graph  # noqa: E402
from ntjoin_amd.engine import MxEngine  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fastas = os.environ["MXG_TEST_FASTAS"].split(":")
    k, w = 32, int(os.environ.get("MXG_TEST_W", "60"))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs = torch.cuda.Stream()
    eng = MxEngine(k=k, w=w, device=0, stream=xs.cuda_stream)
    for i, fa in enumerate(fastas):
        eng.add_fasta_split(f"a{i}", float(i + 1), fa, rank, world)
    eng.global_records = True
    n_cut = sum(int(eng.assembly_continues(a)) for a in range(len(fastas)))
    owner, union = None, None
    for _step in range(3):   # exact exchange, then the fixed-capacity slots
        eng.sketch(-2)
        owner = partitioned_graph(eng, k, w, 0, owner, stream=xs)
    union = sketch_union_graph(eng, k, w, 0, union, stream=xs)
    tot = partitioned_totals(owner)
    pg = owner.get_graph()
    part = {"base": owner.dg["base"], "vhash": pg["vertex_hash"].tolist(),
            "edges": list(zip(pg["edge_u"].tolist(), pg["edge_v"].tolist(), pg["edge_support"].tolist(), pg["edge_weight"].tolist())),
            "cut": n_cut}
    parts = [None] * world
    dist.all_gather_object(parts, part)
    ok = True
    if rank == 0:
        with MxEngine(k=k, w=w, device=0) as whole:
            for i, fa in enumerate(fastas):
                whole.add_fasta(f"a{i}", float(i + 1), fa)
            whole.sketch()
            whole.build_graph()
            g0, g1 = whole.get_graph(), union.get_graph()
        for key in g0:
            if not np.array_equal(np.asarray(g0[key]), np.asarray(g1[key])):
                ok = False
                os.write(1, f"MISMATCH union {key}\n".encode())
        parts.sort(key=lambda p: p["base"])
        ghash = [h for p in parts for h in p["vhash"]]
        names0 = g0["vertex_hash"].tolist()
        want_e = {(names0[u], names0[v]): (s_, w_) for u, v, s_, w_ in
                  zip(g0["edge_u"].tolist(), g0["edge_v"].tolist(), g0["edge_support"].tolist(), g0["edge_weight"].tolist())}
        got_e = {(p["vhash"][u], ghash[v]): (s_, w_) for p in parts for u, v, s_, w_ in p["edges"]}
        n_e = sum(len(p["edges"]) for p in parts)
        if sorted(ghash) != sorted(names0):
            ok = False
            os.write(1, f"MISMATCH partitioned vertices {len(ghash)} vs {len(names0)}\n".encode())
        if got_e != want_e or n_e != len(want_e):
            ok = False
            missing = [k_ for k_ in want_e if k_ not in got_e][:5]
            os.write(1, f"MISMATCH partitioned edges {n_e} / {len(got_e)} vs {len(want_e)}; missing {missing}\n".encode())
        if tot["vertices"] != len(names0) or tot["edges"] != len(want_e):
            ok = False
            os.write(1, f"MISMATCH totals {tot['vertices']}, {tot['edges']}\n".encode())
        if sum(p["cut"] for p in parts) < 2 or len(want_e) < 100:
            ok = False
            os.write(1, b"the test did not cut enough records / graph too small\n")
        os.write(1, f"split: {len(names0)} vertices, {len(want_e)} edges, {sum(p['cut'] for p in parts)} cut pieces, slots in use: {owner._slots is not None}\n".encode())
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    union.close()
    owner.close()
    eng.close()
    dist.destroy_process_group()
    os.write(1, f"SPLIT {'OK' if int(flag) == 1 else 'FAILED'} rank {rank}\n".encode())
    sys.exit(0 if int(flag) == 1 else 1)


if __name__ == "__main__":
    main()
