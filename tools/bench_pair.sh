#!/bin/bash
# repeat-rich workload beside i.i.d. sequence of the same size (usage: tools/bench_pair.sh [mbp])
mbp=${1:-1000}
mkdir -p gpurun_out/${MXG_ROUND:-r04}
timeout 900 python bench.py --workload repeats --mbp $mbp --steps 5 --warmup 2 > gpurun_out/${MXG_ROUND:-r04}/bench_repeats.json 2> gpurun_out/${MXG_ROUND:-r04}/bench_repeats.err
tail -3 gpurun_out/${MXG_ROUND:-r04}/bench_repeats.err
timeout 300 python bench.py --workload configs2 --mbp $mbp --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline --no-kernels > gpurun_out/${MXG_ROUND:-r04}/bench_iid.json
python - <<PY
import json
for n in ("repeats", "iid"):
    d = json.loads(open(f"gpurun_out/${MXG_ROUND:-r04}/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["config"]["bases_per_step"], d["config"]["minimizers"], d["config"]["vertices"], d["fallbacks"])
PY
