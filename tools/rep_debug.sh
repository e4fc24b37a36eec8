#!/bin/bash
# the repeat-rich workload with the batches' reports (MXG_DEBUG_BATCH) and the per-kernel pass: tools/rep_debug.sh [mbp]
cd "$(dirname "$0")/.."
MXG_DEBUG_BATCH=1 python bench.py --workload repeats --mbp ${1:-1000} --steps 3 --warmup 2 2> gpurun_out/rep_debug.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['fallbacks'])"
grep -c "asm" gpurun_out/rep_debug.err; head -30 gpurun_out/rep_debug.err
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ntjoin_amd import synth
from ntjoin_amd.engine import MxEngine
recs = synth.repeat_rich_records(1, 24, 1000_000_000 // 24)
eng = MxEngine(k=32, w=1000, device=0, timing_fine=True)
eng.add_records("ref", 2.0, [(f"r{i}", synth.to_ascii5(c)) for i, c in enumerate(recs)])
for i in range(3):
    eng.sketch(-2)
eng.reset_timers()
t0 = time.perf_counter(); eng.sketch(-2); torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = eng.stats()
print("one assembly sketch ms", dt * 1e3, {k: st[k] for k in ("select_slices", "ms_hash", "ms_reorder", "ms_resolve_kernel", "ms_emit", "candidates", "dense_kmers", "deferred_stretches", "retried_assemblies", "batches_redone", "sync_assemblies")})
PY
