#!/bin/bash
# everything profiles/<round> is made from, at the current commit (run on the GPU box): [ROUND=r06] tools/round_final.sh
# then, in the build container: python tools/make_profile_summaries.py gpurun_out profiles/<round> <round>
R=${ROUND:-r06}
r=gpurun_out/$R
mkdir -p $r
timeout 1200 python bench.py > $r/bench_configs2.json 2> $r/bench_configs2.err
timeout 600 python bench.py --workload configs3 --steps 5 --warmup 2 --no-end-to-end --cpu-seconds 4 > $r/bench_configs3.json 2>/dev/null
timeout 600 python bench.py --workload configs4 --steps 3 --warmup 1 --no-end-to-end --cpu-seconds 4 > $r/bench_configs4.json 2>/dev/null
timeout 600 python bench.py --workload repeats --steps 10 --warmup 3 > $r/bench_repeats.json 2>/dev/null
timeout 600 python bench.py --workload configs2 --mbp 1000 --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-kernels > $r/bench_iid.json 2>/dev/null
for n in 2 4 8; do timeout 900 python bench.py --gpus $n --dry --steps 3 --warmup 1 > $r/bench_dry$n.json 2>/dev/null; done
tools/prof_bench.sh ${R}_stats_configs2 --steps 10 --warmup 3 --no-end-to-end --no-repeats > /dev/null 2>&1
tools/prof_bench.sh ${R}_stats_configs3 --workload configs3 --steps 5 --warmup 2 --no-end-to-end > /dev/null 2>&1
tools/pmc_round.sh ${R}_pmc_configs2 > /dev/null 2>&1
tools/pmc_round.sh ${R}_pmc_configs3 --workload configs3 > /dev/null 2>&1
tools/pmc_calib.sh $R > $r/calib.txt 2>&1
for f in configs2 configs3 configs4 repeats iid dry2 dry4 dry8; do python - "$f" "$R" <<PY
import json, sys
n = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/{sys.argv[2]}/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("kernel", "")[:12], (d.get("roofline") or {}).get("avg_launch_ms"),
          (d.get("end_to_end") or {}).get("value"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(n, "FAILED", e)
PY
done
