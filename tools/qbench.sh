#!/bin/bash
# quick A/B on the GPU box: the k = 32 route's tests, then the default bench line's headline figures.  usage: tools/qbench.sh tag [bench args]
tag=$1; shift
python -m pytest tests/test_gpu_bs.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-end-to-end --no-cpu-baseline --no-repeats "$@" > gpurun_out/qb_$tag.json 2> gpurun_out/qb_$tag.err
python - "$tag" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/qb_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("kernels", {}).get("ms_per_step"))
PY
