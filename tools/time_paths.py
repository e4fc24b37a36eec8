"""time mxg_find_paths on a configs[1]-shaped graph (2 x MBP Mbp, k=32, w=1000): python tools/time_paths.py [mbp]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ntjoin_amd import synth
from ntjoin_amd.engine import MxEngine

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ref, tgt = synth.config2(seed=1, n_bases=mbp * 1_000_000)
with MxEngine(k=32, w=w) as eng:
    for nm, wt, recs in (("ref", 2.0, ref), ("tgt", 1.0, tgt)):
        words, starts, lens = synth.pack_records(recs)
        d = torch.from_numpy(words.view(np.int32)).cuda()
        eng.add_packed_device(nm, wt, d.data_ptr(), starts, lens, keepalive=d)
    eng.sketch(-2)
    eng.build_graph()
    st = eng.stats()
    for n in (1, 2, 1, 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        import ctypes as C
        from ntjoin_amd import capi
        view = capi.PathsView()
        assert eng._lib.mxg_find_paths(eng._h, n, C.byref(view)) == 0
        dt = time.perf_counter() - t0
        found = eng.find_paths(n)
        print(f"mbp={mbp} w={w} n={n}: vertices={st['vertices']} edges={st['edges']} components={eng.n_components} "
              f"paths={len(found)} path_vertices={sum(len(p) for _c, p in found)} find_paths={dt * 1e3:.2f} ms (C call, results on host)")
