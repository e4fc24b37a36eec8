"""Turn what tools/prof_bench.sh and tools/pmc_r04.sh left under gpurun_out/ into the tracked round-4 summaries:
   python tools/make_profile_summaries_r03.py gpurun_out/<stats dir> gpurun_out/<pmc dir> [out dir = profiles/r04]
(stamped with the commit and the digest of the kernel sources: bench.py quotes a static figure only while they are unchanged)
Writes  configs2_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py),  configs2_bench.json (that run's line),
        configs2_pmc_by_kernel.json (counter averages per launch),  hbm_traffic.json (HBM bytes per kernel and per step
        against the algorithmic bytes; bench.py quotes the hash kernel's figure when it runs this workload),
        configs2_hash_kernel_pmc.json (integer-VALU view of the dominant kernel)."""
import json
import os
import shutil
import sys

stats_dir, pmc_dir = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, "profiles", "r04")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(stats_dir, "b_kernel_stats.csv"), os.path.join(out, "configs2_kernel_stats.csv"))
bench = json.loads(open(os.path.join(stats_dir, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(out, "configs2_bench_under_rocprofv3.json"), "w"), indent=1)
pmc = json.load(open(os.path.join(pmc_dir, "pmc_by_kernel.json")))
json.dump(pmc, open(os.path.join(out, "configs2_pmc_by_kernel.json"), "w"), indent=1, sort_keys=True)
pb = json.loads(open(os.path.join(pmc_dir, "bench_pass_3.json")).read().strip().splitlines()[-1])
steps = pb["steps"] + pb["warmup"]
bases_step, mx = pb["config"]["bases_per_step"], pb["config"]["minimizers"]
alg = 0.25 * bases_step + 70.0 * mx
rows, total = [], 0.0
for k, c in pmc.items():
    if "mxg::" not in k or "k_synth" in k:
        continue
    f, w, n = c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0, c["launches"]
    per_step = (2.0 * f + w) * n / steps
    total += per_step
    rows.append({"kernel": k, "launches_per_step": round(n / steps, 2), "fetch_bytes_per_launch_x2": round(2 * f), "write_bytes_per_launch": round(w),
                 "bytes_per_step": round(per_step)})
rows.sort(key=lambda r: -r["bytes_per_step"])
hname = next(k for k in pmc if "k_hash_bs" in k)
hc = pmc[hname]
bases_launch = pb["roofline"]["bases_per_launch"]
hbytes = 2.0 * hc["FETCH_SIZE"] * 1024.0 + hc["WRITE_SIZE"] * 1024.0
traffic = {
    "workload": "configs2", "mbp": 3000.0, "config": pb["config"]["workload"],
    "command": "tools/pmc_r04.sh (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, MXG_ONE_STREAM=1, "
               "bench.py --steps 3 --warmup 1)",
    "fetch_correction": "x2 (gfx950: 128-B requests tallied at 64 B; MI355X_MICROARCH.md HBM section; confirmed in round 2: at strip "
                        "length 128 the rolling-hash kernel's FETCH_SIZE x 2 equalled the packed bases it must read)",
    "k_hash_kernel": hname, "k_hash_bytes_per_launch": round(hbytes), "k_hash_bytes_per_base": hbytes / bases_launch,
    "k_hash_algorithmic_bytes_per_base": 0.25,
    "step": {"hbm_bytes": round(total), "algorithmic_bytes": round(alg), "ratio": round(total / alg, 3),
             "formula": "0.25 B x bases + 70 B x minimizers (SURVEY.md 8d)"},
    "per_kernel": rows,
}
import subprocess
sys.path.insert(0, root)
import bench as _bench
try:
    traffic["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    traffic["commit"] = "?"
traffic["kernel_sources_digest"] = pb.get("kernel_sources_digest") or _bench.kernel_sources_digest()
traffic["k_hash_min_traffic_bytes_per_base"] = 0.25 + 0.25 / 32 + 0.125
json.dump(traffic, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
cycles = hc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
valu = {
    "kernel": hname, "bases_per_launch": bases_launch, "waves": hc["SQ_WAVES"], "valu_wave_instr": hc["SQ_INSTS_VALU"],
    "valu_wave_instr_per_base": hc["SQ_INSTS_VALU"] / bases_launch,
    "valu_lane_ops_per_base": hc["SQ_INSTS_VALU"] * 64.0 / bases_launch,
    "gpu_cycles_per_valu_instr_per_simd": cycles * 1024.0 / hc["SQ_INSTS_VALU"],
    "issue_floor_cycles_per_instr": 2.0,
    "note": "SQ_ACTIVE_INST_VALU equals the instruction count for this kernel (one count per wave instruction), so it says nothing "
            "about busy cycles; the kernel alone on the GPU takes gpu_cycles (GRBM_GUI_ACTIVE / 8 XCDs) for valu_wave_instr / 1024 "
            "instructions per SIMD",
    "lds_instr": hc["SQ_INSTS_LDS"], "wait_any_frac_of_wave_cycles": hc["SQ_WAIT_ANY"] / hc["SQ_WAVE_CYCLES"], "gpu_cycles": cycles,
    "source": "profiles/r04/configs2_pmc_by_kernel.json (tools/pmc_r04.sh)",
}
json.dump(valu, open(os.path.join(out, "configs2_hash_kernel_pmc.json"), "w"), indent=1)
print(json.dumps(traffic["step"], indent=1))
for r in rows[:10]:
    print(r)
print(json.dumps(valu, indent=1))
