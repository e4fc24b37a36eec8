#!/bin/bash
# everything profiles/r04 is made from, at the current commit (run on the GPU box): tools/r04_final.sh
mkdir -p gpurun_out/r04
timeout 900 python bench.py > gpurun_out/r04/bench_configs2.json 2> gpurun_out/r04/bench_configs2.err
timeout 600 python bench.py --workload configs3 --steps 5 --warmup 2 --no-end-to-end --cpu-seconds 4 > gpurun_out/r04/bench_configs3.json 2>/dev/null
timeout 600 python bench.py --workload configs4 --steps 3 --warmup 1 --no-end-to-end --cpu-seconds 4 > gpurun_out/r04/bench_configs4.json 2>/dev/null
tools/bench_pair.sh 1000 > gpurun_out/r04/bench_pair.txt 2>&1
for n in 2 4 8; do timeout 900 python bench.py --gpus $n --dry --steps 3 --warmup 1 > gpurun_out/r04/bench_dry$n.json 2>/dev/null; done
tools/prof_bench.sh r04_stats --steps 10 --warmup 3 --no-end-to-end > /dev/null 2>&1
tools/pmc_r04.sh r04_pmc > /dev/null 2>&1
for f in configs2 configs3 configs4 repeats iid dry2 dry4 dry8; do python - "$f" <<PY
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r04/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("avg_launch_ms"), (d.get("valu") or {}).get("frac_of_issue_bound"),
          (d.get("end_to_end") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(n, "FAILED", e)
PY
done
