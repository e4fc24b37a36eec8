"""Turn what tools/round_final.sh left under gpurun_out/ into the tracked summaries of a round:
   python tools/make_profile_summaries.py [gpurun_out] [out dir = profiles/<round>] [round = r06]
(stamped with the commit and the digest of the kernel sources: bench.py quotes a static figure only while they are unchanged)
Per workload W in (configs2, configs3):
   W_kernel_stats.csv           rocprofv3 --kernel-trace --stats of bench.py on W (what bench.py's roofline.kernel is chosen from)
   W_bench_under_rocprofv3.json that run's line
   W_step_timeline.txt          one step of the trace, kernel by kernel, and how much of it had 0 / 1 / >= 2 kernels running
   W_pmc_by_kernel.json         counter averages per launch (tools/pmc_round.sh: three SQ groups, FETCH_SIZE, WRITE_SIZE, separate passes)
   W_hbm_traffic.json           (configs2: hbm_traffic.json) HBM bytes per kernel and per step against the algorithmic bytes, with the
                                counters' correction per access pattern (traffic_calibration.json)
configs2 only:
   configs2_select_pmc.json     k_bs_select per slice: VALU / SALU / LDS / vector-memory instructions, waits, LDS bank conflicts
   configs2_hash_kernel_pmc.json  k_hash_bs: instructions per base, cycles per instruction
and copies of the bench lines (bench_*.json), the calibration (traffic_calibration.json), and the runs under gpurun_out/<round>/*.txt."""
import glob
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
RND = sys.argv[3] if len(sys.argv) > 3 else "r06"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", RND)
os.makedirs(out, exist_ok=True)
sys.path.insert(0, root)
import bench as _bench  # noqa: E402

try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = "?"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


# which access pattern a kernel's reads have (what the FETCH_SIZE counter tallies depends on it: traffic_calibration.json):
#   stream  wide coalesced reads: 128-byte requests tallied at 64 B -> x 2
#   local   gathers that fall into a few KB per wave (a slice's packed bases, a tile's candidates): neighbouring lanes' requests
#           share 128-byte lines -> between x 1 and x 2; x 2 is reported (upper bound), x 1 beside it
#   gather  random 4..16-byte reads over an array far larger than the caches: one 64-byte request each, tallied at 64 B -> x 1
PATTERN = [("k_hash_bs", "stream"), ("k_bs_select", "local"), ("k_sel_stretch", "local"), ("k_emit", "local"), ("k_gap_fix", "local"), ("k_gap_post", "stream"),
           ("k_pj1_scatter", "stream"), ("k_pj2_bucket", "stream"), ("k_pj_join", "stream"), ("k_flags_pj", "stream"),
           ("k_vertices_pj", "gather"), ("k_adjacency", "gather"), ("k_edge_flags", "stream"), ("k_edges", "gather"), ("k_block_prefix", "stream"),
           ("k_bs_edges", "stream")]
FACTOR = {"stream": 2.0, "local": 2.0, "gather": 1.0}


def pattern_of(kernel):
    for name, pat in PATTERN:
        if name in kernel:
            return pat
    return "stream"


calib = None
cpath = os.path.join(src, RND, "traffic_calibration.json")
if os.path.exists(cpath):
    calib = json.load(open(cpath))
    json.dump(calib, open(os.path.join(out, "traffic_calibration.json"), "w"), indent=1)

for wl in ("configs2", "configs3"):
    sdir, pdir = os.path.join(src, f"{RND}_stats_{wl}"), os.path.join(src, f"{RND}_pmc_{wl}")
    if os.path.exists(os.path.join(sdir, "b_kernel_stats.csv")):
        shutil.copy(os.path.join(sdir, "b_kernel_stats.csv"), os.path.join(out, f"{wl}_kernel_stats.csv"))
        try:
            json.dump(last_json(os.path.join(sdir, "bench.json")), open(os.path.join(out, f"{wl}_bench_under_rocprofv3.json"), "w"), indent=1)
        except Exception as e:
            print(wl, "no bench line under rocprofv3:", e)
        try:
            subprocess.check_call([sys.executable, os.path.join(root, "tools", "step_timeline.py"), os.path.join(sdir, "b_kernel_trace.csv"),
                                   os.path.join(out, f"{wl}_step_timeline.txt")])
        except Exception as e:
            print(wl, "no step timeline:", e)
    if not os.path.exists(os.path.join(pdir, "pmc_by_kernel.json")):
        print(wl, "no PMC passes under", pdir)
        continue
    pmc = json.load(open(os.path.join(pdir, "pmc_by_kernel.json")))
    json.dump(pmc, open(os.path.join(out, f"{wl}_pmc_by_kernel.json"), "w"), indent=1, sort_keys=True)
    pb = last_json(os.path.join(pdir, "bench_pass_4.json"))
    steps = pb["steps"] + pb["warmup"]
    bases_step, mx = pb["config"]["bases_per_step"], pb["config"]["minimizers"]
    alg = 0.25 * bases_step + 70.0 * mx
    rows, total, total_lo = [], 0.0, 0.0
    for k, c in pmc.items():
        if "mxg::" not in k or "k_synth" in k or "k_strip_runs" in k:
            continue
        f, w, n = c.get("FETCH_SIZE", 0.0) * 1024.0, c.get("WRITE_SIZE", 0.0) * 1024.0, c["launches"]
        pat = pattern_of(k)
        per_step = (FACTOR[pat] * f + w) * n / steps
        per_step_lo = (f + w) * n / steps
        total += per_step
        total_lo += per_step_lo if pat != "stream" else per_step
        rows.append({"kernel": k, "launches_per_step": round(n / steps, 2), "read_pattern": pat, "fetch_size_bytes_per_launch": round(f),
                     "fetch_factor": FACTOR[pat], "write_bytes_per_launch": round(w), "bytes_per_step": round(per_step),
                     "bytes_per_step_if_fetch_x1": round(per_step_lo), "avg_us": round(c.get("avg_us", 0.0), 1)})
    rows.sort(key=lambda r: -r["bytes_per_step"])
    hname = next(k for k in pmc if "k_hash_bs" in k)
    hc = pmc[hname]
    bases_launch = (pb["roofline"].get("other_kernel") or pb["roofline"]).get("bases_per_launch") if "k_hash_bs" not in pb["roofline"]["kernel"] \
        else pb["roofline"]["bases_per_launch"]
    hbytes = 2.0 * hc["FETCH_SIZE"] * 1024.0 + hc["WRITE_SIZE"] * 1024.0
    traffic = {
        "workload": wl, "mbp": 3000.0, "config": pb["config"]["workload"],
        "command": f"tools/pmc_round.sh {RND}_pmc_{wl} (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace, separate passes, MXG_ONE_STREAM=1, "
                   "bench.py --steps 3 --warmup 1)",
        "fetch_correction": "per read pattern (traffic_calibration.json: tools/ubench_traffic.hip under the same counters -- a wide coalesced "
                            "streaming read reports FETCH_SIZE = half its bytes (128-byte requests tallied at 64 B), a random 12-byte gather one "
                            "64-byte tally per 128-byte line it touches, writes are tallied as the 32-byte sectors they dirty): stream x 2, "
                            "gather x 1, local (gathers inside a few KB per wave: lanes share lines) x 2 as the upper bound with x 1 beside it",
        "k_hash_kernel": hname, "k_hash_bytes_per_launch": round(hbytes), "k_hash_bytes_per_base": hbytes / bases_launch,
        "k_hash_algorithmic_bytes_per_base": 0.25,
        "step": {"hbm_bytes": round(total), "hbm_bytes_lower_bound": round(total_lo), "algorithmic_bytes": round(alg), "ratio": round(total / alg, 3),
                 "ratio_lower_bound": round(total_lo / alg, 3), "formula": "0.25 B x bases + 70 B x minimizers (SURVEY.md 8d)"},
        "per_kernel": rows, "commit": commit,
        "kernel_sources_digest": pb.get("kernel_sources_digest") or _bench.kernel_sources_digest(),
        "k_hash_min_traffic_bytes_per_base": 0.25 + 0.25 / 32 + 0.125,
    }
    json.dump(traffic, open(os.path.join(out, "hbm_traffic.json" if wl == "configs2" else f"{wl}_hbm_traffic.json"), "w"), indent=1)
    print(wl, json.dumps(traffic["step"]))
    for r in rows[:8]:
        print("   ", r)
    if wl != "configs2":
        continue
    cycles = hc["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
    valu = {
        "kernel": hname, "bases_per_launch": bases_launch, "waves": hc["SQ_WAVES"], "valu_wave_instr": hc["SQ_INSTS_VALU"],
        "valu_wave_instr_per_base": hc["SQ_INSTS_VALU"] / bases_launch, "valu_lane_ops_per_base": hc["SQ_INSTS_VALU"] * 64.0 / bases_launch,
        "gpu_cycles_per_valu_instr_per_simd": cycles * 1024.0 / hc["SQ_INSTS_VALU"], "issue_floor_cycles_per_instr": 2.0,
        "lds_instr": hc["SQ_INSTS_LDS"], "wait_any_frac_of_wave_cycles": hc["SQ_WAIT_ANY"] / hc["SQ_WAVE_CYCLES"], "gpu_cycles": cycles,
        "avg_us": hc.get("avg_us"), "source": f"profiles/{RND}/configs2_pmc_by_kernel.json (tools/pmc_round.sh)", "commit": commit,
    }
    json.dump(valu, open(os.path.join(out, "configs2_hash_kernel_pmc.json"), "w"), indent=1)
    sname = next((k for k in pmc if "k_bs_select" in k), None)
    if sname:
        sc = pmc[sname]
        # slices per launch: the bench line of the pass says how many slices its timed launches took
        spl = (pb["roofline"].get("slices_per_launch") or (pb["roofline"].get("other_kernel") or {}).get("slices_per_launch"))
        scyc = sc["GRBM_GUI_ACTIVE"] / 8.0
        per = lambda name: round(sc.get(name, 0.0) / spl, 1) if spl else None  # noqa: E731
        sel = {
            "kernel": sname, "commit": commit, "kernel_sources_digest": traffic["kernel_sources_digest"], "slices_per_launch": spl,
            "waves": sc["SQ_WAVES"], "waves_per_simd": round(sc["SQ_WAVES"] / 1024.0, 2), "avg_us": sc.get("avg_us"), "gpu_cycles": scyc,
            "per_slice": {"valu": per("SQ_INSTS_VALU"), "salu": per("SQ_INSTS_SALU"), "lds": per("SQ_INSTS_LDS"), "vmem_rd": per("SQ_INSTS_VMEM_RD"),
                          "vmem_wr": per("SQ_INSTS_VMEM_WR"), "flat": per("SQ_INSTS_FLAT"), "smem": per("SQ_INSTS_SMEM"), "branch": per("SQ_INSTS_BRANCH")},
            "gpu_cycles_per_valu_instr_per_simd": scyc * 1024.0 / sc["SQ_INSTS_VALU"],
            "cycles_per_valu_instr": 4.0,
            "cycles_per_valu_instr_note": "model: the kernel's mix of compares, shifts, selects, DPP and SDWA issues at 4 cycles per wave64 instruction "
                                          "(profiles/ubench/README.md: one instruction class at a time); SQ_ACTIVE_INST_VALU counts instructions on this "
                                          "chip, not busy cycles, so the counters neither confirm nor refute it -- gpu_cycles_per_valu_instr_per_simd is "
                                          "what a SIMD really spent per VALU instruction of the launch",
            "lds_bank_conflict_cycles_per_slice": per("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_per_slice": per("SQ_LDS_IDX_ACTIVE"),
            "wait_any_frac": sc["SQ_WAIT_ANY"] / sc["SQ_WAVE_CYCLES"], "wait_inst_lds_frac": sc["SQ_WAIT_INST_LDS"] / sc["SQ_WAVE_CYCLES"],
            "source": f"profiles/{RND}/configs2_pmc_by_kernel.json (tools/pmc_round.sh: MXG_ONE_STREAM=1, the kernel alone on the GPU)",
        }
        json.dump(sel, open(os.path.join(out, "configs2_select_pmc.json"), "w"), indent=1)
        print(json.dumps(sel["per_slice"]), "cycles/VALU/SIMD", round(sel["gpu_cycles_per_valu_instr_per_simd"], 2))

for f in sorted(glob.glob(os.path.join(src, RND, "bench_*.json"))) + sorted(glob.glob(os.path.join(src, RND, "*.txt"))):
    if os.path.getsize(f) > 0 and not f.endswith("calib.txt"):
        shutil.copy(f, os.path.join(out, os.path.basename(f)))
print("written to", out)
