"""what the first step of a fresh handle costs (allocations: MXG_DEBUG_ALLOC=1): [WL=configs3 W=500] python tools/first_step.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ntjoin_amd.engine import MxEngine
WL, W = os.environ.get("WL", "configs2"), int(os.environ.get("W", "1000"))
cfg, asms, label = bench.workload_tables(WL, float(os.environ.get("MBP", "3000")), W, seed=1)
torch.cuda.set_device(0)
for rep in range(2):
    eng = MxEngine(k=32, w=W, device=0)
    keep = [bench.add_rank_share(eng, n, wt, segs, cfg["seed"], ss, sub, 0, 1, 0)[0] for n, wt, segs, _, sub, ss in asms]
    torch.cuda.synchronize()
    for step in range(3):
        t0 = time.perf_counter(); eng.sketch(-2); torch.cuda.synchronize(); t1 = time.perf_counter(); eng.build_graph(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"handle {rep} step {step}: sketch {1e3*(t1-t0):.2f} ms, graph {1e3*(t2-t1):.2f} ms, join {hex(eng.stats()['graph_join'])}", file=sys.stderr)
    eng.close()
