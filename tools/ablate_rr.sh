#!/bin/bash
# one-stream kernel-trace profiles of bench.py with the resolve / reorder ablation builds (profiling only)
export MXG_ONE_STREAM=1
bash tools/prof_bench.sh ab_base --steps 10 --warmup 2
for v in 1 2 3; do MXG_ABLATE_RESOLVE=$v bash tools/prof_bench.sh ab_res$v --steps 10 --warmup 2; done
MXG_ABLATE_REORDER=1 bash tools/prof_bench.sh ab_reo1 --steps 10 --warmup 2
