mkdir -p gpurun_out/s4
(time timeout 1200 python -m pytest tests -m gpu -x -q) > gpurun_out/s4/gputests_final.txt 2>&1
grep -E "passed|failed" gpurun_out/s4/gputests_final.txt | tail -2
(time bash tools/r05_final.sh) > gpurun_out/s4/r05_final.txt 2>&1
tail -12 gpurun_out/s4/r05_final.txt
