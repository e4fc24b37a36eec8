"""Soak of the k = 32 route on repeat-rich records (satellite arrays, low-complexity runs, N gaps, repeat families): sketches against
the C oracle, the joins against one another, over seeds / window sizes / candidate densities.  Test infrastructure (it loads the
oracle): python tools/soak_repeats.py [first_seed n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _oracle
from tests.test_gpu_scale_paths import _check
from ntjoin_amd import synth
from ntjoin_amd.engine import MxEngine

oracle = _oracle.load()
s0, ns = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 12)
os.environ["MXG_DEV_GAPS"] = "1"
n_ok = 0
for seed in range(s0, s0 + ns):
    ref = synth.repeat_rich_records(seed, 3, 250_000 + 1000 * (seed % 7))
    recs = [(f"c{i}", synth.to_ascii5(c).decode()) for i, c in enumerate(ref)]
    for w, cand in ((1000, 10), (500, 10), (1000, 4), (250, 6)):
        for knobs in ({}, {"MXG_SPARSE_BATCH_KMERS": "120000"}, {"MXG_GAP_POOL": "2000"}):
            os.environ.update(knobs)
            st = _check(oracle, recs, 32, w, cand_per_window=cand)
            for k_ in knobs:
                os.environ.pop(k_)
            n_ok += 1
    # the joins: the sketches of the reference and of a target derived from it, every route
    tgt = synth.derive_target_with_n(ref, seed + 1, min_len=5_000, max_len=120_000)
    res = []
    for env in ({}, {"MXG_GRAPH_JOIN": "global"}, {"MXG_PJ_TWO_LEVEL": "1"}, {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_SKEW": "1"},
                {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_PIPE": "0"}):
        os.environ.update(env)
        with MxEngine(k=32, w=250) as eng:
            eng.add_records("ref", 2.0, recs)
            eng.add_records("tgt", 1.0, [(f"t{i}", synth.to_ascii5(c).decode()) for i, c in enumerate(tgt)])
            eng.sketch()
            eng.build_graph()
            out = {f"flags{a}": eng.get_mx_flags(a).copy() for a in range(2)}
            out.update({k_: np.asarray(v).copy() for k_, v in eng.get_graph().items()})
            out["join"] = eng.stats()["graph_join"]
            res.append(out)
        for k_ in env:
            os.environ.pop(k_)
    for other in res[1:]:
        for k_ in res[0]:
            if k_ != "join":
                assert np.array_equal(res[0][k_], other[k_]), (seed, k_)
    n_ok += 1
    print(f"seed {seed}: ok ({len(res[0]['flags0'])} + {len(res[0]['flags1'])} minimizers, joins {[hex(r['join']) for r in res]})", flush=True)
print("soak passed:", n_ok, "checks")
