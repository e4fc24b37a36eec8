#!/bin/bash
# quick A/B of the graph stage on the GPU box: its tests, then the headline figures of configs[2] and configs[3].
# usage: tools/qgraph.sh tag [ENV=value ...]   (the assignments apply to the bench runs only)
tag=$1; shift
timeout 600 python -m pytest tests/test_gpu_join.py tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_paths.py tests/test_gpu_dist2.py tests/test_gpu_dist.py -x -q > gpurun_out/qg_tests_$tag.txt 2>&1
tail -3 gpurun_out/qg_tests_$tag.txt
for v in "$@" ""; do
  n=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-end-to-end --no-cpu-baseline --no-repeats > gpurun_out/qg_${tag}_$n.json 2> gpurun_out/qg_${tag}_$n.err
  env $v timeout 300 python bench.py --workload configs3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > gpurun_out/qg3_${tag}_$n.json 2> gpurun_out/qg3_${tag}_$n.err
  python - "$tag" "$n" <<'PY'
import json, sys
for p in ("qg", "qg3"):
    try:
        d = json.loads(open(f"gpurun_out/{p}_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
        print(p, sys.argv[2] or "default", d["value"], d["ms_per_step"], d.get("kernels", {}).get("ms_per_step"))
    except Exception as e:
        print(p, sys.argv[2], "FAILED", e)
PY
done
