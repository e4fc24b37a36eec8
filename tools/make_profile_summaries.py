"""Turn the raw profiler output that tools/prof_bench.sh and tools/pmc_bench.sh left under gpurun_out/ into the tracked
summaries under profiles/:  python tools/make_profile_summaries.py gpurun_out/<stats dir> gpurun_out/<pmc dir> [tag]"""
import json
import os
import shutil
import sys

stats_dir, pmc_dir = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
shutil.copy(os.path.join(stats_dir, "b_kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(pmc_dir, "pmc_by_kernel.json"), os.path.join(prof, f"{tag}_pmc_by_kernel.json"))
bench = json.loads(open(os.path.join(stats_dir, "bench.json")).read().strip().splitlines()[-1])
bases = bench["roofline"]["bases_per_launch"]
pmc = json.load(open(os.path.join(pmc_dir, "pmc_by_kernel.json")))
name = next(k for k in pmc if "k_hash_sparse<0, 0>" in k)
c = pmc[name]
fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0   # the counters report KiB
traffic = {
    "kernel": name, "FETCH_SIZE_KiB_per_launch": round(c["FETCH_SIZE"], 1), "WRITE_SIZE_KiB_per_launch": round(c["WRITE_SIZE"], 1),
    "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B; MI355X_MICROARCH.md HBM section)",
    "k_hash_bytes_per_launch": 2 * fetch + write, "k_hash_bytes_per_base": (2 * fetch + write) / bases,
    "workload": "bench.py configs[1], one 100 Mbp assembly per launch",
    "command": "tools/pmc_bench.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, MXG_ONE_STREAM=1)",
}
json.dump(traffic, open(os.path.join(prof, "hbm_traffic.json"), "w"), indent=1)
cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
valu = {
    "kernel": name, "waves": c["SQ_WAVES"], "valu_wave_instr": c["SQ_INSTS_VALU"],
    "valu_per_base": c["SQ_INSTS_VALU"] * 64.0 / bases,
    "valu_busy": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cycles),
    "lds_instr": c["SQ_INSTS_LDS"], "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
    "wait_any_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "gpu_cycles": cycles,
    "source": f"profiles/{tag}_pmc_by_kernel.json (tools/pmc_bench.sh)",
}
json.dump(valu, open(os.path.join(prof, "hash_kernel_pmc.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
print(json.dumps(valu, indent=1))
