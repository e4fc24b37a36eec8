#!/bin/bash
# The scaling curve in one command, on a node with several MI355X: one JSON line of bench.py per GPU count (the driver's own
# launch line), N = 1 2 4 8 as far as the node has GPUs.   usage: tools/scale.sh [steps] [warmup]   -> stdout, gpurun_out/scale/
# Beside each measured line the committed one-GPU prediction for that N (profiles/r05/bench_dry<N>.json) is printed, if present.
cd "$(dirname "$0")/.."
steps=${1:-20}; warm=${2:-3}
n_gpu=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)
mkdir -p gpurun_out/scale
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 1 2 4 8; do
  [ "$n" -le "$n_gpu" ] || { echo "{\"n_gpus\": $n, \"skipped\": \"the node has $n_gpu GPU(s)\"}"; continue; }
  out=gpurun_out/scale/bench_n$n.json
  # bench.py starts its own N ranks when no launcher did (WORLD_SIZE unset): the same line for every N
  extra=""; [ "$n" -eq 1 ] && extra="--no-end-to-end"
  timeout 1500 python bench.py --gpus $n --steps $steps --warmup $warm $extra > $out 2> gpurun_out/scale/bench_n$n.err
  grep '^{' $out | tail -1
  [ -f profiles/r05/bench_dry$n.json ] && python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"profiles/r05/bench_dry{n}.json").read().strip().splitlines()[-1])
print(json.dumps({"n_gpus": int(n), "prediction_from_one_gpu": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"]}))
PY
done
