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
            for a in range(len(asms)):
                assert np.array_equal(eng.get_mx_flags(a), union.get_mx_flags(a))
                assert union.record_ids(a, union.n_records(a)) == [f"r0:{x}" for x in eng.record_ids(a, eng.n_records(a))]
            union.close()
    finally:
        dist.destroy_process_group()
