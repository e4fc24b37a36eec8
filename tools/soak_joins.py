"""Soak of the graph stage's joins on synthetic minimizer lists: keys with Zipf-like multiplicities (a few of them 10^4 - 10^5
fold, scattered or in runs), keys missing from some assemblies, 2-5 assemblies -- every route against the global table and the
flags against numpy.  python tools/soak_joins.py [first_seed n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ntjoin_amd.engine import MxEngine

s0, ns = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 8)
for seed in range(s0, s0 + ns):
    r = np.random.default_rng(seed)
    n_asm = int(r.integers(2, 6))
    n_base = int(r.integers(50_000, 400_000))
    base = r.integers(0, 2**63, size=n_base, dtype=np.int64).astype(np.uint64)
    base[3] = np.uint64(0xFFFFFFFFFFFFFFFF)
    sets = []
    for a in range(n_asm):
        hs = base[r.random(n_base) >= r.uniform(0.0, 0.1)].copy()
        r.shuffle(hs)
        parts = [hs]
        for _ in range(int(r.integers(0, 4))):  # heavy keys
            key = base[int(r.integers(0, 50))]
            mult = int(10 ** r.uniform(2, 5.2))
            if r.random() < 0.5:  # one run, somewhere
                at = int(r.integers(0, hs.size))
                parts = [np.concatenate([p[:at], np.full(mult, key, np.uint64), p[at:]]) if p is hs else p for p in parts]
                hs = parts[0]
            else:  # scattered
                parts.append(np.full(mult, key, np.uint64))
        hs = np.concatenate(parts)
        if len(parts) > 1:
            tail = hs[parts[0].size:]
            body = hs[:parts[0].size]
            pos_ins = np.sort(r.integers(0, body.size + 1, size=tail.size))
            hs = np.insert(body, pos_ins, tail)
        n_rec = int(r.integers(1, 40))
        rec = np.sort(r.integers(0, n_rec, size=hs.size)).astype(np.uint32)
        sets.append((hs, np.arange(hs.size, dtype=np.uint32), rec, [f"c{i}" for i in range(n_rec)]))
    # truth for the flags
    cnt = []
    for hs, *_ in sets:
        u, c = np.unique(hs, return_counts=True)
        cnt.append((u, c))
    truth = []
    for a, (hs, *_) in enumerate(sets):
        cs = []
        for u, c in cnt:
            idx = np.searchsorted(u, hs)
            idx[idx >= u.size] = 0
            cs.append(np.where(u[idx] == hs, c[idx], 0))
        cs = np.stack(cs)
        inall = (cs > 0).all(axis=0)
        fl = (cs[a] == 1).astype(np.uint8) | ((inall & (cs.max(axis=0) == 1)).astype(np.uint8) << 1) | (inall.astype(np.uint8) << 2)
        truth.append(fl)
    res = []
    for env in ({}, {"MXG_GRAPH_JOIN": "global"}, {"MXG_PJ_TWO_LEVEL": "1"}, {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_SKEW": "1"},
                {"MXG_PJ_TWO_LEVEL": "1", "MXG_PJ_PIPE": "0"}):
        os.environ.update(env)
        with MxEngine(k=32, w=1000) as eng:
            for i, (hs, pos, rec, ids) in enumerate(sets):
                eng.add_minimizers(f"a{i}", 1.0 + i / 4, hs, pos, rec, ids)
            eng.build_graph()
            eng.build_graph()
            out = {f"flags{a}": eng.get_mx_flags(a).copy() for a in range(n_asm)}
            for a in range(n_asm):
                assert np.array_equal(out[f"flags{a}"], truth[a]), (seed, env, a)
            out.update({k_: np.asarray(v).copy() for k_, v in eng.get_graph().items()})
            res.append((out, eng.stats()["graph_join"]))
        for k_ in env:
            os.environ.pop(k_)
    for other, _ in res[1:]:
        for k_ in res[0][0]:
            assert np.array_equal(res[0][0][k_], other[k_]), (seed, k_)
    print(f"seed {seed}: ok ({n_asm} assemblies, {sum(s[0].size for s in sets)} minimizers, joins {[hex(j) for _, j in res]})", flush=True)
print("soak passed")
