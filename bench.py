#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the hot path on MI355X (contract: the task statement; numbers explained in DESIGN.md 6).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|configs1|configs2|configs3|configs4] [--mbp M]

One STEP = one pass of the whole hot path over one batch of synthetic input already resident in HBM (2-bit packed bases,
produced on the device by the counter-based generator of ntjoin_amd/csrc/synth.hip): sketch every assembly (ntHash +
window arg-min) -> uniqueness -> intersection -> adjacency edges, results handed back through the C-ABI.
metric = Gbp/s = (sum of bases over all assemblies) / time  (device-resident).

Workloads (BASELINE.json configs; `auto` = the config the metric names for this GPU count):
  N=1  configs[2]  24 reference records of 60-220 Mbp totalling 3.0 Gbp (weight 2) + a ~3.0 Gbp target of ~26 k contigs
                   derived from it (cut, half reverse-complemented, 0.5 % substitutions, 20-500 bp dropped, shuffled;
                   weight 1), k=32 w=1000                                    [--workload configs1 / --mbp M: 1 x M Mbp + target]
  N>1  see main(): contig-sharded over the ranks, one RCCL exchange per step (ntjoin_amd/dist.py)
  --workload repeats [--mbp M]: what random genomes lack -- repeat families, satellite arrays, low-complexity runs, N gaps
                   (synth.repeat_rich_records; ASCII route, so N is really there) + a target derived from it: prices the
                   fallbacks (`fallbacks` block: k-mers re-sketched densely, batches redone, assemblies redone whole); compare
                   with `--workload configs2 --mbp M`, the same sizes of i.i.d. sequence

Beside the contract's fields the JSON line carries
  roofline      the kernel that reads every base (k_hash_bs, the bit-sliced ring filter; one launch per assembly) against the
                HBM roof: algorithmic bytes = 0.25 B/bp, time = HIP events recorded by the library on ITS launch stream around
                every launch of the timed region (live)
  valu          the same kernel against what really bounds it, VALU issue: wave64 instructions per chunk counted by the
                generator (csrc/hash_bs_k32.inc) x chunks per launch / 1024 SIMDs x 2 cycles at 2.4 GHz
  step_roofline the whole step's algorithmic bytes (0.25 L + 70 M) over the step time
  kernels       per-kernel GPU time of a step, measured in a second, shorter pass with one HIP-event pair per kernel
  cpu_baseline  the oracle's `indexlr -t nproc` + graph stage in C on the host cores, on a stated sample of the workload
  end_to_end    FASTA files (page cache) -> .tsv + .mx.dot on disk through the CLI a ntJoin user runs (never `value`)
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

K = 32
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
PROFILE_ROUND = "r06"   # profiles/<round>/: the committed rocprofv3 summaries static figures are quoted from
ALG_BYTES_PER_BASE_HASH = 0.25   # hash kernel: one 2-bit packed base read per base
ALG_BYTES_PER_MINIMIZER = 70.0   # whole path: sketch tuple + uniqueness + intersection + edge build


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def n_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs but run the container with a quota of 16: more threads than that only add contention)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                txt = fh.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh2:
                        n = min(n, max(1, q // int(fh2.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def logical_cpus():
    return os.cpu_count() or 1


SIMDS = 256 * 4            # MI355X: 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # peak engine clock (MI355X_MICROARCH.md)
ISSUE_CYCLES = 2.0         # a wave64 VALU instruction occupies its SIMD for 2 cycles at best (measured: profiles/ubench/README.md)
BS_CHUNK = 65536           # base positions per chunk of the bit-sliced filter (csrc/bs_kernels.h)


KERNEL_SOURCES = ("sketch.hip", "sketch_bs.hip", "bs_kernels.h", "hash_bs_k32.inc", "nthash_dev.h", "scan_kernels.h", "graph.hip")


def kernel_sources_digest():
    """sha256 over the kernel sources: profile summaries under profiles/ carry the digest they were captured at, and a static
    figure (PMC traffic) is only quoted while the kernels are still the ones that were profiled"""
    import hashlib
    hsh = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(REPO, "ntjoin_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def bs_route():
    return os.environ.get("MXG_BS", "1") != "0"


def valu_model(bases_per_launch, avg_ms):
    """k_hash_bs against VALU issue, the roof that binds it: the generator counts the wave64 VALU instructions of one chunk
    (HASH_BS_VALU_PER_CHUNK, all of the fast class); a launch's chunks are spread over all SIMDs"""
    try:
        txt = open(os.path.join(REPO, "ntjoin_amd", "csrc", "hash_bs_k32.inc")).read()
        per_chunk = int(txt.split("#define HASH_BS_VALU_PER_CHUNK", 1)[1].split()[0])
    except Exception:
        return None
    chunks = bases_per_launch / BS_CHUNK
    bound_ms = per_chunk * chunks / SIMDS * ISSUE_CYCLES / CLOCK_HZ * 1e3
    return {"kernel": "k_hash_bs", "wave64_valu_instr_per_chunk": per_chunk, "bases_per_chunk": BS_CHUNK,
            "wave64_valu_instr_per_base": round(per_chunk / BS_CHUNK, 5),
            "lane_ops_per_base": round(per_chunk * 64 / BS_CHUNK, 3),
            "valu_issue_bound_ms": round(bound_ms, 4), "avg_launch_ms": round(avg_ms, 4),
            "frac_of_issue_bound": round(bound_ms / avg_ms, 4) if avg_ms > 0 else None,
            "model": f"instructions x chunks / {SIMDS} SIMDs x {ISSUE_CYCLES:g} cycles / {CLOCK_HZ / 1e9:g} GHz; the instruction count is "
                     "static (generated code: ntjoin_amd/csrc/gen/bs_gen.py), the launch time is this run's"}


SELECT_ISSUE_CYCLES = 3.49  # what a wave64 VALU instruction of k_bs_select's mix (47 % slow class: compares, address arithmetic, SDWA) occupies its
                            # SIMD at four waves per SIMD: MEASURED, profiles/r06/select_issue_ubench.txt (profiles/ubench/select_mix.hip; 4.0 assumed until round 5)
                           # for: the instruction classes measured one at a time in profiles/ubench/README.md -- anything outside the
                           # filter's full-rate class keeps the stream at ~4 cycles per instruction
ALG_BYTES_PER_MINIMIZER_TUPLE = 16.0  # the (hash, pos, record) tuple a selected minimizer is written as


def committed_kernel_stats(wl):
    """rocprofv3 --kernel-trace --stats summary of this workload committed under profiles/<round>/ -> [(kernel, total ns, avg ns, calls)]
    of the step's kernels, largest first (the generator k_synth and the runtime's fill / copy kernels are not part of a step)"""
    import csv
    path = os.path.join(REPO, "profiles", PROFILE_ROUND, f"{wl}_kernel_stats.csv")
    rows = []
    try:
        with open(path, newline="") as fh:
            for r in csv.DictReader(fh):
                name = r["Name"].split("(")[0].replace("void ", "").replace("mxg::", "")
                if name.startswith("k_synth") or name.startswith("__amd_rocclr") or name.startswith("k_strip_runs"):
                    continue
                rows.append((name, float(r["TotalDurationNs"]), float(r["AverageNs"]), int(r["Calls"])))
    except (OSError, KeyError, ValueError):
        return None, path
    rows.sort(key=lambda x: -x[1])
    return rows, path


def select_valu_model(launch_ms, slices_per_launch):
    """k_bs_select against VALU issue: wave64 instructions per slice from the committed PMC passes of the kernel at these sources
    (profiles/<round>/configs2_select_pmc.json, made by tools/pmc_sel.sh), this run's launch time"""
    path = os.path.join(REPO, "profiles", PROFILE_ROUND, "configs2_select_pmc.json")
    try:
        pj = json.load(open(path))
    except (OSError, ValueError):
        return None
    if pj.get("kernel_sources_digest") != kernel_sources_digest():
        return {"kernel": "k_bs_select", "stale": f"profiles/{PROFILE_ROUND}/configs2_select_pmc.json was captured at commit {pj.get('commit', '?')} "
                                                  "and the kernel sources have changed since; not quoted"}
    per = pj["per_slice"]
    bound_ms = per["valu"] * slices_per_launch / SIMDS * SELECT_ISSUE_CYCLES / CLOCK_HZ * 1e3
    return {"kernel": "k_bs_select", "wave64_instr_per_slice": per, "slices_per_launch": int(slices_per_launch),
            "cycles_per_valu_instr_measured": SELECT_ISSUE_CYCLES,
            "cycles_per_valu_instr_measured_by": "profiles/r06/select_issue_ubench.txt: the kernel's VALU opcode mix as a register-only loop at four waves per SIMD "
                                                 "(3.49; fast class alone 2.39, slow class alone 4.09)",
            "wave_cycles_per_valu_instr_in_the_pmc_pass": pj.get("cycles_per_valu_instr"), "lds_bank_conflict_cycles_per_slice": pj.get("lds_bank_conflict_cycles_per_slice"),
            "wait_any_frac_of_wave_cycles": pj.get("wait_any_frac"), "waves_per_simd": pj.get("waves_per_simd"),
            "valu_issue_bound_ms": round(bound_ms, 4), "avg_launch_ms": round(launch_ms, 4),
            "frac_of_issue_bound": round(bound_ms / launch_ms, 4) if launch_ms > 0 else None,
            "source": f"profiles/{PROFILE_ROUND}/configs2_select_pmc.json (commit {pj.get('commit', '?')})",
            "gpu_cycles_per_valu_instr_per_simd_in_the_pmc_pass": pj.get("gpu_cycles_per_valu_instr_per_simd"),
            "model": f"VALU instructions per slice x slices / {SIMDS} SIMDs x {SELECT_ISSUE_CYCLES:g} cycles / {CLOCK_HZ / 1e9:g} GHz: the kernel's "
                     "instruction mix (47 % slow class) issues at 3.49 cycles per wave64 instruction, measured (select_issue_ubench.txt), not at the 2.4 "
                     "of the filter's fast-class stream"}


def workload_tables(name, mbp, w, seed=1):
    """segment tables of the assemblies of a workload, references first (the reference's load order, bin/ntjoin.py:178-186):
    list of (assembly name, weight, segs, n_words, sub_per_65536, sub_seed)"""
    from ntjoin_amd import synth
    if name == "configs1":
        cfg = synth.genome_config(int(mbp * 1e6), 1, seed=seed)
        label = f"configs[1]: synthetic 1x{mbp:g} Mbp reference + derived target, k=32 w={w}, weights 2/1"
    elif name in ("configs2", "configs3"):
        cfg = synth.genome_config(int(mbp * 1e6), 24, seed=seed, min_len=3000, max_len=600_000)
        label = (f"configs[2]: synthetic human-scale {mbp / 1000:g} Gbp reference (24 records) + derived ~{mbp / 1000:g} Gbp "
                 f"target, k=32 w={w}, weights 2/1")
    elif name == "configs4":
        cfg = synth.genome_config(int(mbp * 1e6), 12, seed=seed, min_len=1000, max_len=200_000)
        label = (f"configs[4]: synthetic conifer-scale {mbp / 1000:g} Gbp reference (12 records of ~{mbp / 12000:.2f} Gbp, cut between "
                 f"ranks) + derived ~{mbp / 1000:g} Gbp target of {len(cfg['tgt_segs'])} contigs of 1-200 kbp, k=32 w={w}, weights 2/1")
    else:
        raise ValueError(name)
    ss = cfg["sub_seed"]
    if name == "configs3":
        # target + 3 references: the same base genome with independent 0.5 / 1 / 2 % divergence, weights 2/2/2 and 1
        label = (f"configs[3]: synthetic {mbp / 1000:g} Gbp target + 3 references (0.5 / 1 / 2 % divergence), k=32 w={w}, "
                 "weights 2/2/2/1")
        asms = [(f"ref{i}.fa.k{K}.w{w}.tsv", 2.0, cfg["ref_segs"], cfg["ref_words"], rate, ss + 101 * (i + 1))
                for i, rate in enumerate((328, 655, 1311))]
        asms.append((f"tgt.fa.k{K}.w{w}.tsv", 1.0, cfg["tgt_segs"], cfg["tgt_words"], synth.SUB_PER_65536, ss))
    else:
        asms = [(f"ref.fa.k{K}.w{w}.tsv", 2.0, cfg["ref_segs"], cfg["ref_words"], 0, ss),
                (f"tgt.fa.k{K}.w{w}.tsv", 1.0, cfg["tgt_segs"], cfg["tgt_words"], synth.SUB_PER_65536, ss)]
    return cfg, asms, label


def add_rank_share(eng, name, weight, segs, seed, sub_seed, sub, rank, world, device, split=True):
    """this rank's share of one assembly, born in HBM.  split=True: shard `rank` of `world` EQUAL BASE RANGES of the concatenated
    records, whatever the record borders (records cut by a range travel as pieces with a halo of w k-mers, mxg_plan_split);
    split=False: a contiguous range of WHOLE records balanced by base count (mxg_shard_range) -- what the hash-partitioned graph
    stage needs, whose adjacency messages never leave a record's rank.  Every rank registers every record, so record indices
    are global.  world = 1: the whole assembly.  -> (device tensor, rec_start, rec_len) of what the rank holds"""
    from ntjoin_amd import synth
    from ntjoin_amd.dist import shard_range
    lens = np.ascontiguousarray(segs[:, 2])
    if split or world == 1:
        lo, hi, drop = eng.plan_split(lens, rank, world)
    else:
        r0, r1 = shard_range(lens, rank, world)
        lo, hi, drop = np.zeros(len(lens), dtype=np.uint64), np.zeros(len(lens), dtype=np.uint64), np.zeros(len(lens), dtype=np.uint8)
        hi[r0:r1] = lens[r0:r1]
    keep = hi > lo
    b0 = lo & ~np.uint64(15)
    plen = np.where(keep, hi - b0, 0).astype(np.uint64)
    starts_k, n_words = synth.layout(plen[keep])
    rec_start = np.zeros(len(lens), dtype=np.uint64)
    rec_start[keep] = starts_k
    loc = np.zeros((int(keep.sum()), 4), dtype=np.uint64)
    rc = segs[keep, 3]
    loc[:, 0] = starts_k
    loc[:, 1] = np.where(rc == 1, segs[keep, 1] + (lens[keep] - hi[keep]), segs[keep, 1] + b0[keep])
    loc[:, 2] = plen[keep]
    loc[:, 3] = rc
    d = synth.fill_device(loc, n_words, seed, sub_seed, sub, device=device)
    eng.add_packed_device_pieces(name, weight, d.data_ptr(), rec_start, lens, lo, hi, drop)
    return d, rec_start, lens


def cpu_baseline(asms_host, w, budget_s, weights):
    """The oracle (checker / reported baseline only: nothing here is on the product path): the C restatement of
    `indexlr -t nproc` (chunked records, one worker per core) + the C restatement of read_minimizers' uniqueness,
    filter_minimizers and build_graph, on a sample of the workload sized for ~budget_s seconds of wall time."""
    from tests import _oracle
    orc = _oracle.load()
    cores = n_cores()
    # calibrate on ~16 Mbp per core, then size the sample
    words0, starts0, lens0 = asms_host[0]
    cal_len = int(min(int(lens0[0]), 4_000_000 * cores))
    t0 = time.perf_counter()
    orc.sketch_packed_mt(words0, starts0[:1], np.array([cal_len], dtype=np.uint64), K, w, threads=cores, chunk_kmers=1 << 18)
    rate = cal_len / max(time.perf_counter() - t0, 1e-6)  # bases per second, all cores
    total = sum(int(l.sum()) for _, _, l in asms_host)
    frac = min(1.0, rate * budget_s / total)
    sample, bases = [], 0
    for words, starts, lens in asms_host:
        csum = np.cumsum(lens.astype(np.int64))
        n = int(np.searchsorted(csum, frac * csum[-1], side="left")) + 1
        n = min(max(n, 1), len(lens))
        sample.append((words, starts[:n], lens[:n]))
        bases += int(lens[:n].sum())
    t0 = time.perf_counter()
    sk = [orc.sketch_packed_mt(wd, st, ln, K, w, threads=cores, chunk_kmers=1 << 18) for wd, st, ln in sample]
    t_sketch = time.perf_counter() - t0
    t1 = time.perf_counter()
    g = orc.graph([s[0] for s in sk], [s[2] for s in sk], list(weights))
    t_graph = time.perf_counter() - t1
    return {"bases": bases, "seconds": t_sketch + t_graph, "t_sketch": t_sketch, "t_graph": t_graph, "cores": cores,
            "minimizers": int(sum(len(s[0]) for s in sk)), "vertices": g["vertices"], "edges": g["edges"], "frac": frac,
            "records": [len(s[1]) for s in sample]}


E2E_SETTLE_S = 1.0


def end_to_end(asms_host, w, td, threads):
    """FASTA files in the page cache -> .tsv + .mx.dot on disk, through what a user of ntJoin:204-205 runs: the native
    `indexlr` CLI per assembly (cold process, HIP init included), then `python -m ntjoin_amd.run` (TSVs -> .mx.dot)."""
    from ntjoin_amd import capi
    lib = capi.load()
    fas, sizes = [], 0
    for i, (words, starts, lens) in enumerate(asms_host):
        fa = os.path.join(td, ("ref.fa", "tgt.fa")[i] if len(asms_host) == 2 else f"asm{i}.fa")
        rc = lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, np.ascontiguousarray(starts, dtype=np.uint64).ctypes.data,
                                       np.ascontiguousarray(lens, dtype=np.uint64).ctypes.data, len(lens), b"s", 80, max(threads, 1))
        if rc != 0:
            raise RuntimeError(f"mxg_synth_write_fasta failed ({rc})")
        fas.append(fa)
        sizes += os.path.getsize(fa)
    for fa in fas:  # make sure the text is in the page cache (it was just written; read it once anyway)
        with open(fa, "rb") as fh:
            while fh.read(1 << 26):
                pass
    exe = os.path.join(REPO, "ntjoin_amd", "bin", "indexlr")
    tsvs = [f"{fa}.k{K}.w{w}.tsv" for fa in fas]
    # Both routes start from a quiet device: a process that opens the GPU while the driver is still releasing the tens of GB of a
    # process that has just ended waits for that (device + handle 0.23-0.5 s instead of 0.08 s, measured with tools/e2e_ab.py).
    # The pause is outside the timed regions; inside the two-process route its three processes follow one another as they do
    # under make.
    time.sleep(E2E_SETTLE_S)
    t0 = time.perf_counter()
    for fa, tsv in zip(fas, tsvs):
        subprocess.check_call([exe, "--seq", "--long", "--pos", f"-k{K}", f"-w{w}", f"-t{threads}", "-o", tsv, fa])
    t_sketch = time.perf_counter() - t0
    t1 = time.perf_counter()
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.check_call([sys.executable, "-m", "ntjoin_amd.run", "-p", os.path.join(td, "out"), "-n", "1", "-s", tsvs[-1], "-l", "1",
                           "-r", " ".join(["2"] * (len(tsvs) - 1)), "-k", str(K)] + tsvs[:-1], env=env, stdout=subprocess.DEVNULL)
    t_graph = time.perf_counter() - t1
    two = {t: open(t, "rb").read(1 << 16) for t in tsvs}  # (heads of the two-process outputs, compared with the one-process ones)
    dot_size = os.path.getsize(os.path.join(td, "out.mx.dot"))
    for t in tsvs:
        os.remove(t)
    # the same job in ONE process (ntJoin-mx mxgraph's default recipe): FASTA -> TSVs + .mx.dot, one HIP initialisation, the graph
    # stage on the sketches in HBM
    one = os.path.join(REPO, "ntjoin_amd", "bin", "mxgraph")
    # (every cold process is one sample of a noisy quantity -- one run in five or so loses 30-70 ms to the first kernels of the process:
    # each leg runs twice, the line carries both times and quotes the faster one)
    t_one_all, phases, same, phases_all = [], None, True, []
    for _rep in range(2):
        for t in tsvs + [os.path.join(td, "one.mx.dot")]:
            if os.path.exists(t):
                os.remove(t)
        time.sleep(E2E_SETTLE_S)
        t2 = time.perf_counter()
        pr = subprocess.run([one, "-v", f"-k{K}", f"-w{w}", f"-t{threads}", "-p", os.path.join(td, "one"), "-s", fas[-1], "-l", "1",
                             "-r", " ".join(["2"] * (len(fas) - 1))] + fas[:-1], check=True, stderr=subprocess.PIPE, text=True)
        t_one_all.append(time.perf_counter() - t2)
        ph = next((ln.split("mxgraph: ", 1)[1] for ln in pr.stderr.splitlines() if "device + handle" in ln), None)
        phases_all.append(ph)
        if t_one_all[-1] == min(t_one_all):
            phases = ph
        same = same and all(open(t, "rb").read(1 << 16) == two[t] for t in tsvs) and os.path.getsize(os.path.join(td, "one.mx.dot")) == dot_size
    t_one = min(t_one_all)
    # ... and the same once more with the worker attached (MXG_NO_DETACH=1): `mxgraph` does its work in a child and the parent
    # returns when every output is written and closed; the child then still gives ~13 GB of HBM and its pinned memory back to the
    # driver.  t_one stops with the parent (what a user waiting for the files sees); this one stops when the process that held
    # the GPU is gone (what the NEXT GPU job waits for).
    t_att_all = []
    for _rep in range(2):
        for t in tsvs + [os.path.join(td, "one.mx.dot")]:
            os.remove(t)
        time.sleep(E2E_SETTLE_S)
        t3 = time.perf_counter()
        subprocess.run([one, f"-k{K}", f"-w{w}", f"-t{threads}", "-p", os.path.join(td, "one"), "-s", fas[-1], "-l", "1",
                        "-r", " ".join(["2"] * (len(fas) - 1))] + fas[:-1], check=True, stderr=subprocess.DEVNULL, env=dict(os.environ, MXG_NO_DETACH="1"))
        t_att_all.append(time.perf_counter() - t3)
        same = same and all(open(t, "rb").read(1 << 16) == two[t] for t in tsvs) and os.path.getsize(os.path.join(td, "one.mx.dot")) == dot_size
    t_attached = min(t_att_all)
    bases = sum(int(l.sum()) for _, _, l in asms_host)
    return {"bases": bases, "fasta_bytes": sizes, "t_sketch_cli": t_sketch, "t_graph_cli": t_graph, "t_one_process": t_one, "t_attached": t_attached,
            "t_one_all": t_one_all, "t_attached_all": t_att_all, "phases_all": phases_all,
            "one_process_same_outputs": bool(same), "one_process_phases": phases,
            "tsv_bytes": sum(os.path.getsize(t) for t in tsvs), "dot_bytes": os.path.getsize(os.path.join(td, "out.mx.dot"))}


XGMI_LINK_GBS = 153.0 * 0.5   # one xGMI link, one direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, both directions)


def exchange_beside_sketches(sketch_ms, send_ms):
    """when the last of the per-assembly exchanges ends: assembly a is sketched behind assembly a - 1 (sketch_ms[a] each), its exchange
    (send_ms[a]) starts when its sketch has ended AND the exchange of the assembly in front of it has (one communication stream)"""
    t = c = 0.0
    for s_a, x_a in zip(sketch_ms, send_ms):
        t += s_a
        c = max(c, t) + x_a
    return c


def dry_run(args):
    """bench.py --gpus N --dry on one GPU: rank r = 0..N-1 of the N-GPU workload one after the other, each at its real share
    (configs[4] for N = 8: 1/8 of the bases of both assemblies = 5 Gbp per rank), sketch stage timed for real; the exchanges are
    priced from the counts at the link rate (partitioned: 16 B per minimizer to its hash's owner, 8 B verdict back, 16 B per
    adjacency message to each end point's owner, (N-1)/N of it leaves the rank, spread over N-1 links; union: the fixed part of
    12 B per minimizer + 4 B per record to every peer over its own link).  The graph stages are TIMED through the code the real run
    takes: the partitioned route's partitioned_graph() on the rank's own minimizers with its collectives replaced by copies (the
    same number of keys and records as the rank would own, not the same keys), the union route's graph stage on the real union of
    the ranks' sketches.  Both routes as built (an assembly's exchange beside the next assembly's sketch) and with every exchange
    behind the sketches; value = the fastest, as the real run picks by measurement."""
    import torch
    from ntjoin_amd.engine import MxEngine
    N = args.gpus
    if N < 2 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        sys.exit("bench.py --dry: one process, --gpus N >= 2")
    torch.cuda.set_device(0)
    wl = args.workload if args.workload != "auto" else {2: "configs2", 4: "configs3", 8: "configs4"}.get(N, "configs2")
    W = args.w or (500 if wl == "configs3" else 1000)
    mbp = args.mbp or {"configs1": 100.0, "configs2": 3000.0, "configs3": 3000.0, "configs4": 20000.0}[wl]
    cfg, asms, label = workload_tables(wl, mbp, W, seed=1)
    bases_job = sum(int(a[2][:, 2].sum()) for a in asms)
    ranks = []
    parts = [[] for _ in asms]   # per assembly: every rank's sketch (host arrays), for the union route's graph stage
    rec_ids = None
    # the partitioned route's graph stage is timed through its REAL code path (ntjoin_amd/dist.py partitioned_graph: the rank's packing /
    # verdict / message kernels and the owner's kernels, on as many minimizers as the rank owns) with the collectives of ONE rank: copies
    import torch.distributed as tdist
    from ntjoin_amd.dist import partitioned_graph

    class _Done:
        def wait(self):
            return True

    def _copy(out, inp, *a, **k):
        out.view(-1)[:inp.numel()].copy_(inp.reshape(-1))
        return _Done()
    loop = {"get_world_size": lambda group=None: 1, "get_rank": lambda group=None: 0, "all_to_all_single": _copy,
            "all_gather_into_tensor": _copy, "all_reduce": lambda t, *a, **k: _Done()}
    saved = {k_: getattr(tdist, k_) for k_ in loop}
    xs = torch.cuda.Stream()
    for r in range(N):
        eng = MxEngine(k=K, w=W, device=0, timing=True, timing_fine=True, cand_per_window=args.cand, stream=xs.cuda_stream)
        keep = [add_rank_share(eng, name, weight, segs, cfg["seed"], sub_seed, sub, r, N, 0)[0] for name, weight, segs, _, sub, sub_seed in asms]
        eng.global_records = True
        for _ in range(max(args.warmup, 1)):
            eng.sketch(-2)
            eng.build_graph()
        eng.reset_timers()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.sketch(-2)
        torch.cuda.synchronize()
        t_sk = (time.perf_counter() - t0) / args.steps * 1e3
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.build_graph()
        torch.cuda.synchronize()
        t_gr = (time.perf_counter() - t1) / args.steps * 1e3
        st = eng.stats()
        for k_, f_ in loop.items():
            setattr(tdist, k_, f_)
        try:
            owner = None
            for _ in range(3):                      # the exact exchange, then the fixed slots
                owner = partitioned_graph(eng, K, W, 0, owner, stream=xs)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(args.steps):
                owner = partitioned_graph(eng, K, W, 0, owner, stream=xs)
            torch.cuda.synchronize()
            t_dg = (time.perf_counter() - t3) / args.steps * 1e3
            dg_slots = getattr(owner, "_slots", None) is not None
            owner.close()
        finally:
            for k_, f_ in saved.items():
                setattr(tdist, k_, f_)
        # shared minimizers on this rank: the job's vertex count x assemblies / minimizers, from the committed one-GPU run of the
        # same workload when there is one (a rank's own two shares are different stretches of the genome: its own graph says nothing)
        frac = 0.75
        try:
            one = json.loads(open(os.path.join(REPO, "profiles", PROFILE_ROUND, f"bench_{wl}.json")).read().strip().splitlines()[-1])
            frac = one["config"]["vertices"] * len(asms) / max(one["config"]["minimizers"], 1)
        except Exception:
            pass
        m = int(st["minimizers"])
        per_asm = []
        for a in range(len(asms)):
            sk = eng.get_sketch(a)
            per_asm.append(len(sk["out_hash"]))
            parts[a].append((sk["out_hash"].copy(), sk["pos"].copy(), sk["record"].copy()))
            if r == 0:
                rec_ids = (rec_ids or []) + [sk["record_ids"]]
        shared = int(frac * m)
        out_frac = (N - 1) / N
        sent = {"items": int(16 * m * out_frac), "verdicts": int(8 * m * out_frac), "adjacency_messages": int(2 * 16 * shared * out_frac)}  # one message to each end point's owner
        per_link = sum(sent.values()) / (N - 1)
        ranks.append({"rank": r, "bases": int(st["bases"]), "minimizers": m, "minimizers_by_assembly": per_asm,
                      "bases_by_assembly": [int(a_[2][:, 2].sum()) // N for a_ in asms],   # (equal base ranges of every assembly)
                      "sketch_ms": round(t_sk, 3), "graph_stage_on_own_minimizers_ms": round(t_gr, 3),
                      "partitioned_graph_stage_ms": round(t_dg, 3), "partitioned_graph_stage_with_fixed_slots": dg_slots,
                      "kernel_ms_per_step": {"filter": round(st["ms_hash"] / args.steps, 3), "select (or count+reorder)": round(st["ms_reorder"] / args.steps, 3),
                                             "stretches (+resolve)": round(st["ms_resolve_kernel"] / args.steps, 3), "emit": round(st["ms_emit"] / args.steps, 3),
                                             "join": round(st["ms_join"] / args.steps, 3), "vertices": round(st["ms_vertices"] / args.steps, 3),
                                             "edges": round(st["ms_edges"] / args.steps, 3)},
                      "bytes_sent_per_step": sent, "exchange_ms_at_link_rate": round(per_link / (XGMI_LINK_GBS * 1e9) * 1e3, 3)})
        eng.close()
        del keep
        torch.cuda.empty_cache()
    part_ms = max(x["sketch_ms"] + x["partitioned_graph_stage_ms"] + x["exchange_ms_at_link_rate"] for x in ranks) + 6 * 0.03
    # ... and as built since round 6: every assembly's items (16 B per minimizer to its hash's owner) leave in an all-to-all of their own
    # when that assembly's sketch ends, never before the assembly in front of it; verdicts and adjacency messages behind all of them
    part_ovl = 0.0
    for x in ranks:
        tot_b = max(sum(x["bases_by_assembly"]), 1)
        s_a = [x["sketch_ms"] * b / tot_b for b in x["bases_by_assembly"]]
        items_ms = [16 * m_a * (N - 1) / N / (N - 1) / (XGMI_LINK_GBS * 1e9) * 1e3 + 0.03 for m_a in x["minimizers_by_assembly"]]  # (bytes per link)
        c = exchange_beside_sketches(s_a, items_ms)
        rest = (x["bytes_sent_per_step"]["verdicts"] + x["bytes_sent_per_step"]["adjacency_messages"]) / (N - 1) / (XGMI_LINK_GBS * 1e9) * 1e3
        part_ovl = max(part_ovl, c + rest + x["partitioned_graph_stage_ms"] + 5 * 0.03)
    # ---- the union route (ntjoin_amd/dist.py sketch_union_graph): every rank's sketch to every rank, the graph of the union on
    # every rank.  Its graph stage is measured on the real union of the ranks' sketches; an assembly's part is its fixed slot
    # (MXG_XCHG_SLOT_PCT above the largest share), all-gathered over N - 1 links at once (one part per link and direction).
    un = MxEngine(k=K, w=W, device=0, timing=True)
    for a, (name, weight, *_rest) in enumerate(asms):
        un.add_minimizers(name, weight, np.concatenate([p_[0] for p_ in parts[a]]), np.concatenate([p_[1] for p_ in parts[a]]),
                          np.concatenate([p_[2] for p_ in parts[a]]), rec_ids[a])
    for _ in range(max(args.warmup, 1)):
        un.build_graph()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        un.build_graph()
    torch.cuda.synchronize()
    t_union_graph = (time.perf_counter() - t2) / args.steps * 1e3
    ug = un.get_graph()
    union_counts = {"vertices": int(len(ug["vertex_hash"])), "edges": int(len(ug["edge_u"]))}
    un.close()
    del parts
    pct = int(os.environ.get("MXG_XCHG_SLOT_PCT", "110"))
    link = XGMI_LINK_GBS * 1e9
    slot_caps = [(max(x["minimizers_by_assembly"][a] for x in ranks) * pct // 100 + 64 + 7) // 8 * 8 for a in range(len(asms))]
    part_bytes = [64 + 12 * c + 4 * ((len(rec_ids[a]) + 3) // 4 * 4) for a, c in enumerate(slot_caps)]   # (ntjoin_amd/dist.py part_bytes)
    one_slot_bytes = 64 + 16 * sum(slot_caps)   # (the single gather's slot: 16 B per entry, record column included)
    gather_ms = [b / link * 1e3 + 0.03 for b in part_bytes]   # (+ one collective's latency)
    unpack_ms = sum(28.0 * N * (b - 64) / 12 / 3.0e12 * 1e3 for b in part_bytes)  # the unpack kernels move 12 B in + 16 B out per entry at ~3 TB/s
    union_one, union_ovl = 0.0, 0.0
    for x in ranks:
        tot_b = max(sum(x["bases_by_assembly"]), 1)
        s_a = [x["sketch_ms"] * b / tot_b for b in x["bases_by_assembly"]]   # the assemblies one behind the other
        union_ovl = max(union_ovl, exchange_beside_sketches(s_a, gather_ms))  # part a travels as soon as it is packed and the part before it has gone
        union_one = max(union_one, x["sketch_ms"] + one_slot_bytes / link * 1e3 + 0.03)
    union_ovl += unpack_ms + t_union_graph
    union_one += unpack_ms + t_union_graph
    routes = {"partitioned, every exchange behind the sketches (MXG_XCHG_OVERLAP=0)": round(part_ms, 3),
              "partitioned, an assembly's items beside the next assembly's sketch (as built)": round(part_ovl, 3),
              "union, one all-gather behind the sketches (MXG_XCHG_OVERLAP=0)": round(union_one, 3),
              "union, one all-gather per assembly beside the next assembly's sketch (as built)": round(union_ovl, 3)}
    step_ms = min(routes.values())   # bench.py --gpus N tries the routes in its warm-up and times the faster one
    out = {"metric": f"PREDICTION of Gbp/s minimizer-sketch+graph-build (k=32,w={W}) on {N} GPUs, from one GPU playing every rank in turn",
           "value": round(bases_job / (step_ms * 1e-3) / 1e9, 2), "unit": "Gbp/s", "n_gpus": N, "dry": True, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(step_ms, 3), "higher_is_better": True, "data": "synthetic", "dtype": "u64",
           "config": {"workload": label + f", rank shares of 1/{N} of every assembly's bases", "bases_per_step": int(bases_job)},
           "model": "partitioned: step = max over ranks of (sketch stage + exchange bytes / (N-1) links at "
                    f"{XGMI_LINK_GBS:g} GB/s per link and direction + the route's own graph stage -- partitioned_graph() timed on one rank whose collectives are copies: the rank's packing, "
                    "verdict and message kernels and the owner's kernels on as many minimizers as the rank owns; until round 6 the model took the one-GPU graph stage here, 3-4 x less) + 6 collectives x 30 us; "
                    "the five host syncs of the exact partitioned exchange are inside the measured stages' own syncs or not modelled; as built, an assembly's items leave when its sketch ends (sketch stage split by bases), one all-to-all per assembly, verdicts and messages behind them.  "
                    "union: an assembly's part = its fixed slot (12 B per entry + 4 B per record, MXG_XCHG_SLOT_PCT above the largest share) to every peer over its own link; "
                    "per assembly: ready when its sketch ends (sketch stage split by bases), gone one part-time + 30 us later, never before the part in front of it; "
                    "then unpack + the graph stage MEASURED on the union of the ranks' sketches.  value = the fastest route (bench.py --gpus N measures the routes and takes the faster)",
           "routes_ms_per_step": routes, "union_graph_stage_ms": round(t_union_graph, 3), "union_part_bytes": part_bytes, "union_graph": union_counts,
           "ranks": ranks}
    print(json.dumps(out), flush=True)
    return 0


def repeat_rich_side_run(cand):
    """What random genomes lack, beside the headline: `--workload repeats` (1 Gbp + derived target; repeat families, satellite
    arrays, low-complexity runs, N gaps) and i.i.d. sequence of the same size, each in a process of its own -- steady state and
    the first step of the handle, which every CLI run is."""
    def run(extra):
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-end-to-end", "--no-kernels",
               "--no-repeats", "--cand", str(cand)] + extra
        try:
            pr = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            return json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:  # (a side measurement: the headline does not depend on it)
            return {"error": repr(e)[:200]}
    rep, iid = run(["--workload", "repeats"]), run(["--workload", "configs2", "--mbp", "1000"])
    if "error" in rep or "error" in iid:
        return {"error": rep.get("error") or iid.get("error")}
    first = rep["fallbacks"]["first_step_of_the_handle"] or {}
    return {"workload": rep["config"]["workload"], "value": rep["value"], "unit": "Gbp/s", "ms_per_step": rep["ms_per_step"],
            "iid_same_size_value": iid["value"], "iid_same_size_ms_per_step": iid["ms_per_step"],
            "slower_than_iid": round(iid["value"] / rep["value"], 3) if rep["value"] else None,
            "first_step_of_the_handle": first, "first_step_iid_ms": (iid["fallbacks"]["first_step_of_the_handle"] or {}).get("ms"),
            "per_step": {k_: rep["fallbacks"][k_] for k_ in ("dense_kmers_per_step", "batches_redone_per_step", "assemblies_redone_whole_per_step",
                                                            "assemblies_enqueued_twice_per_step", "stretches_sketched_apart_per_step")},
            "what": "bench.py --workload repeats / --workload configs2 --mbp 1000, 10 steps each in processes of their own (never `value`)"}


def other_k_run(w=1000, mbp=1000.0, steps=10, warmup=3):
    """`bench.py --other-k`: the routes the headline does not take, on 1 Gbp + 1 Gbp of the configs[2] genome -- k = 24 (no
    bit-sliced filter: the rolling-hash kernel k_hash_sparse -> k_reorder_w -> k_resolve) and k = 32 with the `min(fwd, rev)`
    variant (the same kernels; reference ntJoin:33,36: k and w are the user's variables).  One JSON object; never `value`."""
    import torch
    from ntjoin_amd.engine import MxEngine
    torch.cuda.set_device(0)
    cfg, asms, _ = workload_tables("configs2", mbp, w, seed=1)
    bases = sum(int(a[2][:, 2].sum()) for a in asms)
    out = {}
    for tag, k, variant in (("k24", 24, None), ("k32_min_variant", 32, "v1")):
        kw = {"variant": variant} if variant else {}
        eng = MxEngine(k=k, w=w, device=0, timing=True, **kw)
        keep = [add_rank_share(eng, name, weight, segs, cfg["seed"], sub_seed, sub, 0, 1, 0)[0] for name, weight, segs, _, sub, sub_seed in asms]
        for _ in range(warmup):
            eng.sketch(-2)
            eng.build_graph()
        eng.reset_timers()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.sketch(-2)
            eng.build_graph()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = eng.stats()
        out[tag] = {"k": k, "variant": "min(fwd, rev) (v1)" if variant else "fwd + rev (v2)", "value": round(bases * steps / dt / 1e9, 2), "unit": "Gbp/s",
                    "ms_per_step": round(dt / steps * 1e3, 4), "minimizers": int(st["minimizers"]), "vertices": int(st["vertices"]), "edges": int(st["edges"]),
                    "bit_sliced_filter_ran": bool(st["bs_filter_bases"]), "hash_kernel_ms_per_step": round(st["ms_hash"] / steps, 4),
                    "hash_kernel_frac_of_hbm_roof": round(ALG_BYTES_PER_BASE_HASH * st["hash_kernel_bases"] / max(st["ms_hash"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                    "batches_redone": int(st["batches_redone"])}
        eng.close()
        del keep
        torch.cuda.empty_cache()
    out["workload"] = f"configs[2] genome at {mbp / 1000:g} Gbp + {mbp / 1000:g} Gbp, w={w}, weights 2/1, {steps} steps each"
    out["what"] = "the rolling-hash route (k_hash_sparse -> k_reorder_w -> k_resolve -> k_emit) + the graph stage, in a process of its own; never `value`"
    print(json.dumps(out), flush=True)
    return 0


def other_k_side_run():
    try:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--other-k"], capture_output=True, text=True, timeout=600)
        return json.loads([ln for ln in pr.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # (a side measurement: the headline does not depend on it)
        return {"error": repr(e)[:200]}


def scaling_fields(wl, W, world, value, bases_total):
    """N > 1: what one GPU does on the SAME workload (the committed line of `bench.py --workload <wl>` under profiles/, with the commit
    it was measured at) and value / (N x that).  The default workload differs with N (BASELINE.json names configs[2] for one and two
    GPUs, configs[3] for four, configs[4] for eight), so value(N) / value(1) of two default lines is not an efficiency; this is."""
    for rnd in sorted((d for d in os.listdir(os.path.join(REPO, "profiles")) if d.startswith("r0")), reverse=True):
        path = os.path.join(REPO, "profiles", rnd, f"bench_{wl}.json")
        if not os.path.exists(path):
            continue
        try:
            one = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])
        except Exception:
            continue
        c = one.get("config", {})
        if one.get("n_gpus") != 1 or c.get("w") != W or abs(c.get("bases_per_step", 0) - bases_total) > 0.05 * bases_total:
            continue  # (another size of the workload)
        v1 = float(one["value"])
        return {"one_gpu_same_workload": {"value": v1, "unit": "Gbp/s", "ms_per_step": one.get("ms_per_step"), "from": f"profiles/{rnd}/bench_{wl}.json",
                                          "kernel_sources_digest": one.get("kernel_sources_digest"), "bases_per_step": c.get("bases_per_step")},
                "efficiency": round(value / (world * v1), 4),
                "efficiency_is": f"value / ({world} x one_gpu_same_workload.value): same workload, same sizes, one GPU (committed line; not measured in this run)"}
    return {"one_gpu_same_workload": None, "efficiency": None,
            "efficiency_is": f"no committed one-GPU line of workload {wl} under profiles/: run `python bench.py --workload {wl}` on one GPU"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here -- one process per GPU, the environment torchrun
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT on 127.0.0.1 with a free port) -- hand rank 0's
    stdout through (its last line is the JSON line), drop the other ranks' stdout, and end with a non-zero status as soon as any
    rank does (the others are then stopped: a rank waiting in a collective for a dead one would hang the run) OR when the run
    outlives its wall-clock budget (MXG_BENCH_BUDGET_S, default 900 s: a rank stuck inside a collective -- a link that never
    comes up, a peer that spins -- keeps every status at "running"): the process groups are then stopped and every rank's last
    stderr lines are printed, so that the one lease a multi-GPU run gets ends with a diagnosis instead of the driver's timeout.
    Every rank's stderr goes through a file of its own (shown in full for rank 0 at the end, as a tail for the others).
    The driver's own `python -m torch.distributed.run ... bench.py --gpus N` takes the other branch (WORLD_SIZE set)."""
    import signal
    import socket
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    budget = float(os.environ.get("MXG_BENCH_BUDGET_S", "900"))
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MXG_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        errs.append(tempfile.TemporaryFile(mode="w+b"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=errs[r], start_new_session=True))
    rc, why = 0, None
    t_start = time.perf_counter()
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if c != 0 and rc == 0:
                        rc = c if c > 0 else 1
                        why = f"rank {r} ended with status {c}; stopping the other ranks"
            if rc == 0 and live and time.perf_counter() - t_start > budget:
                rc = 124
                why = (f"ranks {sorted(live)} still running after {budget:g} s (MXG_BENCH_BUDGET_S): stuck in a collective or before the "
                       "rendezvous; stopping every rank")
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)   # the exact process groups started above
                except ProcessLookupError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
        if why:
            print(f"bench.py: {why}", file=sys.stderr, flush=True)
        for r, f in enumerate(errs):
            f.seek(0)
            text = f.read().decode("utf-8", "replace")
            f.close()
            if not text.strip():
                continue
            lines = text.rstrip("\n").splitlines()
            if rc and (r != 0 or len(lines) > 40):
                lines = ["..."] + lines[-15:] if len(lines) > 15 else lines
            if rc or r == 0:
                print("\n".join(f"[rank {r}] {ln}" for ln in lines), file=sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto", "configs1", "configs2", "configs3", "configs4", "repeats"])
    ap.add_argument("--mbp", type=float, default=0.0, help="reference size in Mbp (default: 100 for configs1, 3000 for configs2)")
    ap.add_argument("--w", type=int, default=0, help="window size (default: 500 for configs3, else 1000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel breakdown pass")
    ap.add_argument("--no-repeats", action="store_true", help="skip the repeat-rich side measurement of the default line")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="wall-time budget of the cpu_baseline sample")
    ap.add_argument("--e2e-mbp", type=float, default=0.0, help="end-to-end sample: Mbp per assembly (0 = the whole workload)")
    ap.add_argument("--cand", type=int, default=0, help="candidates per window for the sparse path (0 = library default)")
    ap.add_argument("--dry", action="store_true",
                    help="with --gpus N on ONE GPU and one process: play every rank's share of the N-GPU workload in turn (real sizes, real "
                         "kernels, no collective) and print the per-rank kernel times, the bytes each rank would send and the step time "
                         "they predict -- a prediction to hold the first real N-GPU run against, never a measurement of it")
    ap.add_argument("--other-k", action="store_true", help="only the side measurement of the routes the headline does not take (k = 24, the min variant)")
    args = ap.parse_args()
    if args.other_k:
        return other_k_run()
    if args.dry:
        return dry_run(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MXG_BENCH_ONE_DEVICE") == "1":  # testing: several ranks share GPU 0 (with MXG_BENCH_BACKEND=gloo)
        local_rank = 0
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if os.environ.get("MXG_BENCH_KILL_RANK") == str(rank) and world > 1:  # testing: a rank that dies before the rendezvous
        sys.exit(f"bench.py: rank {rank} told to fail (MXG_BENCH_KILL_RANK)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("MXG_BENCH_FORCE_DIST") == "1"  # exercise the N>1 path with one rank (testing)
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("MXG_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for the one-GPU test
        # (a collective that does not complete within this ends the rank with an error instead of the default ten minutes)
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("MXG_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    from ntjoin_amd.dist import partitioned_graph, partitioned_totals, sketch_union_graph

    wl = args.workload
    if wl == "auto":  # the configuration BASELINE.json's metric names for this GPU count
        wl = {1: "configs2", 2: "configs2", 4: "configs3", 8: "configs4"}.get(world, "configs2")
    W = args.w or (500 if wl == "configs3" else 1000)
    mbp = args.mbp or {"configs1": 100.0, "configs2": 3000.0, "configs3": 3000.0, "configs4": 20000.0, "repeats": 1000.0}[wl]
    repeats = wl == "repeats"
    if repeats:
        if multi:
            sys.exit("bench.py: --workload repeats is a one-GPU workload")
        args.no_kernels = args.no_cpu_baseline = args.no_end_to_end = True
        n_rec = 24
        ref_recs = synth.repeat_rich_records(1, n_rec, int(mbp * 1e6) // n_rec)
        tgt_recs = synth.derive_target_with_n(ref_recs, 2, min_len=3000, max_len=600_000)
        label = (f"repeat-rich synthetic {mbp / 1000:g} Gbp reference ({n_rec} records: 30 % repeat-family copies, 8 % satellite arrays, "
                 f"5 % low-complexity runs, 2 % N gaps) + derived target of {len(tgt_recs)} contigs, k=32 w={W}, weights 2/1, ASCII route")
        as_segs = lambda recs: np.array([[0, 0, len(r), 0] for r in recs], dtype=np.uint64)  # noqa: E731  (lengths only)
        cfg = {"seed": 1}
        asms = [(f"ref.fa.k{K}.w{W}.tsv", 2.0, as_segs(ref_recs), 0, 0, 0), (f"tgt.fa.k{K}.w{W}.tsv", 1.0, as_segs(tgt_recs), 0, 0, 0)]
    else:
        cfg, asms, label = workload_tables(wl, mbp, W, seed=1)  # the same genome on every rank: each takes its share of it
    keep, host_layout = [], []
    # N > 1: the library works on the stream the collectives are issued on, so pack -> all-gather -> unpack need no host sync
    xstream = torch.cuda.Stream() if multi else None
    if not multi:  # HIP-event pairs around one batch in four: around every batch they cost 2 % of the step (sums scaled below)
        os.environ.setdefault("MXG_TIMING_SAMPLE", "4")
    eng = MxEngine(k=K, w=W, device=local_rank, timing=True, cand_per_window=args.cand,
                   stream=xstream.cuda_stream if xstream is not None else None)
    bases_job = sum(int(a[2][:, 2].sum()) for a in asms)
    # the graph route of N > 1: MXG_BENCH_GRAPH names it; else both are tried in the warm-up (two steps each behind a first one that
    # sets the exchange up) and the faster one -- the same on every rank: the times are max-reduced -- runs the timed steps
    graph_mode = os.environ.get("MXG_BENCH_GRAPH") or ("measure" if multi else "union")
    route_trial = None
    if repeats:
        for (name, weight, *_), recs in zip(asms, (ref_recs, tgt_recs)):
            eng.add_records(name, weight, [(f"r{i}", synth.to_ascii5(c)) for i, c in enumerate(recs)])
        del ref_recs, tgt_recs
    for name, weight, segs, n_words, sub, sub_seed in ([] if repeats else asms):
        d, rec_start, rec_len = add_rank_share(eng, name, weight, segs, cfg["seed"], sub_seed, sub, rank if world > 1 else 0, world,
                                               local_rank, split=os.environ.get("MXG_BENCH_WHOLE_RECORDS") != "1")  # bases born in HBM
        keep.append(d)
        host_layout.append((d, rec_start, rec_len))
    eng.global_records = True
    torch.cuda.synchronize()
    union = None

    n_calls = [0]

    def step(e=eng):
        nonlocal union
        n_calls[0] += 1
        if multi and n_calls[0] == 2:  # testing: a rank that dies / hangs while its peers wait for it inside the step's collectives
            if os.environ.get("MXG_BENCH_DIE_IN_STEP") == str(rank):
                time.sleep(1.0)
                os._exit(3)
            if os.environ.get("MXG_BENCH_HANG_RANK") == str(rank):
                print(f"bench.py: rank {rank} told to hang (MXG_BENCH_HANG_RANK)", file=sys.stderr, flush=True)
                time.sleep(3600.0)
        if not multi and os.environ.get("MXG_BENCH_FUSED", "1") != "0":
            # mxg_sketch_graph: every assembly's sketch and the graph stage in ONE call with one host sync (the graph stage is enqueued
            # behind the sketches with the counts read on the device); round 6: 1.3 % faster than the two calls (2.57-2.60 against
            # 2.60-2.63 ms on one box, interleaved), no faster in round 3.  MXG_BENCH_FUSED=0: mxg_sketch(ALL) + mxg_build_graph
            e.sketch_graph()
            return
        if multi and graph_mode != "partitioned":
            # sketch -> pack -> all-gather -> unpack -> graph of the union, one host sync per step in steady state
            union = sketch_union_graph(e, K, W, local_rank, union, stream=xstream)
            return
        if multi:
            # this rank's part of the graph; the call sketches too: in steady state every assembly's items leave for their owners
            # while the next assembly is still being sketched (ntjoin_amd/dist.py: _partitioned_slots, sketch_inside)
            union = partitioned_graph(e, K, W, local_rank, union, stream=xstream, sketch=True)
        else:
            e.sketch(-2)  # MXG_SKETCH_ALL: every assembly enqueued back to back
            e.build_graph()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if graph_mode == "measure":
        route_trial = {}
        unions = {}
        for mode in ("union", "partitioned"):
            graph_mode, union = mode, None
            step()                      # (the first step of a route: sizes exchanged, slots laid out)
            step()
            fence()
            t_r = time.perf_counter()
            step()
            step()
            fence()
            t_mode = torch.tensor([(time.perf_counter() - t_r) / 2], dtype=torch.float64, device="cuda")
            dist.all_reduce(t_mode, op=dist.ReduceOp.MAX)
            route_trial[mode] = round(float(t_mode[0]) * 1e3, 4)
            unions[mode] = union
        graph_mode = min(route_trial, key=route_trial.get)
        union = unions[graph_mode]
        for m_, u_ in unions.items():
            if m_ != graph_mode and u_ is not None:
                u_.close()
        del unions
    cold = None
    for i in range(args.warmup):
        if i == 0 and not multi:  # the first pass of a fresh handle (what a one-shot CLI run pays): no candidate counts yet
            torch.cuda.synchronize()
            tc = time.perf_counter()
            step()
            torch.cuda.synchronize()
            t_cold = time.perf_counter() - tc   # (before mxg_get_stats: its lazy count of the unique minimizers copies the flags to the
            cs = eng.stats()                    # host -- ~5 ms that rounds 4-6 booked on the first step by reading the clock behind it)
            cold = {"ms": round(t_cold * 1e3, 3), "batches_redone": int(cs["batches_redone"]),
                    "assemblies_redone_whole": int(cs["sync_assemblies"]), "dense_kmers": int(cs["dense_kmers"]),
                    "assemblies_enqueued_twice": int(cs["retried_assemblies"]), "stretches_sketched_apart": int(cs["deferred_stretches"])}
        else:
            step()
    eng.reset_timers()
    fence()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_times.append(time.perf_counter() - ts)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    bases_total = float(bases_job)  # the whole job's bases: every rank sketched 1/world of every assembly
    dist_info = None
    if multi:
        # "did the communicator really hold N ranks on N devices": a sum of ones over the backend's all-reduce, and every rank's
        # device (index + PCI bus id) gathered through the same communicator
        ones = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(ones)
        pr = torch.cuda.get_device_properties(local_rank)
        mine = torch.tensor([local_rank, getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", -1)], dtype=torch.int64, device="cuda")
        devs = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(devs, mine)
        be = dist.get_backend()
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if be == "nccl" else None
        except Exception:
            rccl = None
        dist_info = {"world": world, "backend": be + (" (= RCCL on ROCm)" if be == "nccl" else " (test backend: ranks may share one GPU)"),
                     "rccl_version": rccl, "communicator_ranks": dist.get_world_size(), "allreduce_of_ones": int(ones[0]),
                     "devices_by_rank": [[int(v) for v in d.tolist()] for d in devs],
                     "distinct_devices": len({tuple(int(v) for v in d.tolist()) for d in devs}),
                     "launched_by": "bench.py itself (python bench.py --gpus N)" if os.environ.get("MXG_BENCH_SELF_LAUNCHED") == "1"
                                    else "an external launcher (torch.distributed.run)",
                     "graph_route": graph_mode,
                     "graph_route_chosen_by": ("MXG_BENCH_GRAPH" if os.environ.get("MXG_BENCH_GRAPH") else
                                               {"ms_per_step_in_the_warm_up": route_trial, "rule": "the faster of two warm-up steps per route (max over ranks)"}),
                     "collective_timeout_s": float(os.environ.get("MXG_BENCH_COLLECTIVE_TIMEOUT_S", "120"))}

    st = eng.stats()
    if union is not None and graph_mode == "partitioned":
        gst = dict(union.stats())
        gst.update(partitioned_totals(union))      # global vertex / edge counts (collectives: every rank calls)
    else:
        gst = (union.stats() if union is not None else st)
    result_line = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = bases_total * args.steps / dt / 1e9
        # dominant kernel = the ntHash/candidate kernel; HIP events recorded by the library on ITS launch stream
        launches = max(st["launches_hash"], 1)
        avg_ms = st["ms_hash"] / launches
        # the timed launches cover hash_kernel_bases of the bases hashed in the timed region: the stage sums are scaled by that
        t_scale = max(1.0, bases_total / world * args.steps / max(st["hash_kernel_bases"], 1)) if not multi else 1.0
        bytes_per_launch = ALG_BYTES_PER_BASE_HASH * st["hash_kernel_bases"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        n_mx = int(st["minimizers"]) if not multi else int(gst.get("minimizers", st["minimizers"]))
        step_alg_bytes = ALG_BYTES_PER_BASE_HASH * bases_total + ALG_BYTES_PER_MINIMIZER * float(st["minimizers"]) * world
        step_gbs = step_alg_bytes / (ms_step * 1e-3) / 1e9
        # PMC traffic of the hash kernel: only quoted when the committed counters were collected on THIS workload
        traffic, traffic_src, tj_commit = None, None, None
        tpath = os.path.join(REPO, "profiles", PROFILE_ROUND, "hbm_traffic.json")
        if os.path.exists(tpath) and bs_route():
            try:
                tj = json.load(open(tpath))
                tj_commit = tj.get("commit")
                if tj.get("workload") == wl and abs(tj.get("mbp", 0) - mbp) < 1e-9 and not multi:
                    if tj.get("kernel_sources_digest") == kernel_sources_digest():
                        traffic = round(tj["k_hash_bytes_per_base"] * st["hash_kernel_bases"] / launches)
                        traffic_src = ("static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload at commit "
                                       f"{tj.get('commit', '?')}, profiles/" + PROFILE_ROUND + "/hbm_traffic.json")
                    else:
                        traffic_src = (f"stale: profiles/{PROFILE_ROUND}/hbm_traffic.json was captured at commit {tj.get('commit', '?')} and the "
                                       "kernel sources have changed since; not quoted")
            except Exception:
                traffic = None
        out = {
            "metric": f"Gbp/s minimizer-sketch+graph-build (k=32,w={W})", "value": round(value, 4), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong" if (world > 1 and wl == "configs2") else "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": label + ", bases resident in HBM (2-bit packed, generated on the device)",
                       "k": K, "w": W, "bases_per_step": int(bases_total), "minimizers": int(st["minimizers"]) if world == 1 else None,
                       "minimizers_rank0": int(st["minimizers"]),
                       "vertices": int(gst["vertices"]), "edges": int(gst["edges"]),
                       "records": [int(len(a[2])) for a in asms],
                       "parallelism": ("1 GPU" if world == 1 else f"{world} ranks, " +
                                       ("a contiguous range of whole records of every assembly each (balanced by bases), "
                                        if os.environ.get("MXG_BENCH_WHOLE_RECORDS") == "1" else
                                        f"each 1/{world} of every assembly's bases (records cut at the range borders travel as pieces "
                                        "with a w-k-mer halo), ") +
                                       ("graph stage partitioned by hash range (RCCL all-to-all)" if graph_mode == "partitioned"
                                        else "RCCL all-gather of sketches, graph of the union on every rank")),
                       "exchange": (None if not multi else
                                    ("every assembly's items leave in an all-to-all of their own while the next assembly is sketched"
                                     if graph_mode == "partitioned" and union is not None and getattr(union, "_slots", None) is not None
                                     and getattr(union, "_comm", None) is not None and not getattr(union, "_no_overlap", False)
                                     else "items behind the sketches") if graph_mode == "partitioned" else
                                    ("one all-gather per assembly beside the next assembly's sketch, 12 B per minimizer"
                                     if union is not None and getattr(union, "_slots", None) and "send_parts" in union._slots
                                     else "one all-gather behind the sketches, 16 B per minimizer"))},
            "distributed": dist_info,
            **(scaling_fields(wl, W, world, value, bases_total) if world > 1 else {}),
            "kernel_sources_digest": kernel_sources_digest(),
            "knobs_in_force": eng.knobs(),  # MXG_* environment switches the handle read and found set ("" = library defaults)
            "step_is": ("mxg_sketch_graph(h): every assembly's sketch + the graph stage in one call, one host sync"
                        if (not multi and os.environ.get("MXG_BENCH_FUSED", "1") != "0") else
                        ("mxg_sketch(h, MXG_SKETCH_ALL) + mxg_build_graph(h): two calls, two host syncs" if not multi else "sketch + exchange + graph (see distributed)")),
            "resident_input": "2-bit packed bases (0.25 B/bp) as handed over through mxg_add_assembly_packed_device*; every step reads them as they are",
            "step_ms_min_max": [round(min(step_times) * 1e3, 4), round(max(step_times) * 1e3, 4)],
            "roofline": None,  # (filled below: the dominant kernel of the step)
            "valu": valu_model(st["hash_kernel_bases"] / launches, avg_ms) if bs_route() else None,
            "step_roofline": {"bound": "hbm", "alg_bytes_per_step": int(step_alg_bytes),
                              "formula": "0.25 B x bases + 70 B x minimizers (SURVEY.md 8d)",
                              "achieved": round(step_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_gbs / HBM_PEAK_GBS, 6)},
            "stage_ms_per_step": {"filter": round(st["ms_hash"] * t_scale / args.steps, 4),
                                  "graph": round(gst["ms_graph"] / args.steps, 4)},
            # what the common route (every batch enqueued once, one host sync per step) could not finish, per step
            "fallbacks": {"dense_kmers_per_step": int(st["dense_kmers"] / max(args.steps, 1)),
                          "batches_redone_per_step": round(st["batches_redone"] / max(args.steps, 1), 3),
                          "assemblies_redone_whole_per_step": round(st["sync_assemblies"] / max(args.steps, 1), 3),
                          "assemblies_enqueued_twice_per_step": round(st["retried_assemblies"] / max(args.steps, 1), 3),
                          "stretches_sketched_apart_per_step": round(st["deferred_stretches"] / max(args.steps, 1), 1),
                          "candidates_per_step": int(st["candidates"] / max(args.steps, 1)),
                          "first_step_of_the_handle": cold,
                          "note": "counted over the timed steps (warm: the candidate counts of the warm-up steps size the grids)"},
        }
        # ---- the roofline object: the step's dominant kernel, and the two kernels that together turn bases into minimizer tuples
        filter_obj = {"bound": "hbm",
                      "kernel": ("k_hash_bs (bit-sliced ntHash top rings + candidate filter, one launch per assembly)" if bs_route()
                                 else "k_hash_sparse (ntHash fwd/rc rings + candidate filter)"),
                      "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                      "traffic_commit": tj_commit,
                      "alg_bytes_per_base": ALG_BYTES_PER_BASE_HASH, "alg_bytes_per_launch": int(bytes_per_launch),
                      "avg_launch_ms": round(avg_ms, 4), "launches": int(st["launches_hash"]),
                      "timed": "HIP-event pair around every launch of the kernel inside the timed region, on the launch stream",
                      "streams": ("free-running (MXG_STAGGER=0): this kernel shares the GPU with the other assembly's slice kernel"
                                  if os.environ.get("MXG_STAGGER") == "0" else
                                  "an assembly of 2^31 k-mers or more starts its filter behind the slice kernel of the assembly before it, so "
                                  "the kernel's time is its own (MXG_STAGGER=0: free-running streams, the step ~2 % shorter, kernel times shared)"),
                      "min_traffic_bytes_per_base": 0.25 + 0.25 / 32 + 0.125,
                      "min_traffic_note": "what this formulation must move: 2 bits per base (bit planes) + 1/32 of that (the "
                                          "strips' predecessors) read, 1 bit per position (the candidate bitmap) written",
                      "bases_per_launch": int(st["hash_kernel_bases"] / launches),
                      "share_of_step_time": round(st["ms_hash"] * t_scale / args.steps / ms_step, 4)}
        roof = filter_obj
        sel_ran = bs_route() and st.get("select_slices", 0) > 0 and not multi
        if sel_ran:
            # k_bs_select: one launch per assembly and step (a span of its own in the timed region); SURVEY.md 8(d) prices what it
            # produces at the 16-byte tuple per selected minimizer -- bitmap -> candidates -> window decision has no bytes of its own
            n_asm = len(asms)
            sel_launches = args.steps * n_asm
            sel_ms_step = st["ms_reorder"] / args.steps
            sel_launch_ms = st["ms_reorder"] / max(sel_launches, 1)
            mx_launch = float(st["minimizers"]) / n_asm
            sel_bytes = ALG_BYTES_PER_MINIMIZER_TUPLE * mx_launch
            sel_ach = sel_bytes / (sel_launch_ms * 1e-3) / 1e9 if sel_launch_ms > 0 else 0.0
            sel_traffic = None
            try:
                tj2 = json.load(open(tpath))
                if tj2.get("workload") == wl and tj2.get("kernel_sources_digest") == kernel_sources_digest():
                    row = next(r for r in tj2.get("per_kernel", []) if "k_bs_select" in r["kernel"])
                    sel_traffic = int(row["bytes_per_step"] / row["launches_per_step"])
            except Exception:
                sel_traffic = None
            select_obj = {"bound": "hbm", "kernel": "k_bs_select (the filter's bitmap -> exact hashes -> window decision -> selected minimizers, one launch per assembly)",
                          "achieved": round(sel_ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sel_ach / HBM_PEAK_GBS, 6),
                          "traffic": sel_traffic,
                          "traffic_source": (f"static: profiles/{PROFILE_ROUND}/hbm_traffic.json (counter passes at these kernel sources; the correction "
                                             "of each counter for this kernel's access patterns: profiles/" + PROFILE_ROUND + "/traffic_calibration.json)")
                                            if sel_traffic else None,
                          "alg_bytes_per_minimizer": ALG_BYTES_PER_MINIMIZER_TUPLE, "alg_bytes_per_launch": int(sel_bytes),
                          "alg_note": "SURVEY.md 8(d) defines bytes for what this kernel produces (the 16-byte minimizer tuple), none for what it "
                                      "reads (bitmap, packed bases of the candidates): the fraction says how far an instruction-bound kernel is "
                                      "from the roof of the little it must write; what bounds it is in valu_by_kernel",
                          "avg_launch_ms": round(sel_launch_ms, 4), "launches": int(sel_launches),
                          "timed": "HIP-event pair around every launch of the kernel inside the timed region, on the launch stream",
                          "minimizers_per_launch": int(mx_launch), "slices_per_launch": int(st["select_slices"] / max(sel_launches, 1)),
                          "share_of_step_time": round(sel_ms_step / ms_step, 4)}
            # which of the two is the step's largest consumer: the committed rocprofv3 stats of this workload say (total duration)
            stats, spath = committed_kernel_stats(wl)
            dom, dom_src = None, None
            if stats:
                dom = stats[0][0]
                tot_ns = sum(r[1] for r in stats)
                dom_src = (f"{os.path.relpath(spath, REPO)}: " + ", ".join(f"{r[0]} {100 * r[1] / tot_ns:.1f} %" for r in stats[:3]) +
                           " of the step's kernel time")
            live_dom = "k_bs_select" if sel_ms_step > st["ms_hash"] * t_scale / args.steps else "k_hash_bs"
            if dom is None or not (dom.startswith("k_bs_select") or dom.startswith("k_hash_bs")):
                dom, dom_src = live_dom, "this run's event pairs (no committed rocprofv3 stats of this workload)"
            roof = dict(select_obj if dom.startswith("k_bs_select") else filter_obj)
            roof["dominant_by"] = dom_src
            roof["dominant_in_this_run"] = live_dom
            pair_bytes = ALG_BYTES_PER_BASE_HASH * bases_total + ALG_BYTES_PER_MINIMIZER_TUPLE * float(st["minimizers"])
            pair_ms = st["ms_hash"] * t_scale / args.steps + sel_ms_step
            pair_gbs = pair_bytes / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0
            roof["sketch_pair"] = {"kernels": "k_hash_bs + k_bs_select", "alg_bytes_per_step": int(pair_bytes),
                                   "formula": "0.25 B x bases + 16 B x minimizers: together the two kernels do what SURVEY.md 8(d) prices so",
                                   "ms_per_step": round(pair_ms, 4), "achieved": round(pair_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(pair_gbs / HBM_PEAK_GBS, 6)}
            roof["other_kernel"] = filter_obj if dom.startswith("k_bs_select") else select_obj
        out["roofline"] = roof
        vbk = {}
        if out.get("valu"):
            vbk["k_hash_bs"] = out["valu"]
        if sel_ran:
            sm = select_valu_model(select_obj["avg_launch_ms"], select_obj["slices_per_launch"])
            if sm:
                vbk["k_bs_select"] = sm
        out["valu_by_kernel"] = vbk or None
        if not multi and not args.no_kernels:
            # per-kernel GPU time: a second handle on the same bases with one event pair per kernel (a few steps)
            eng2 = MxEngine(k=K, w=W, device=local_rank, timing_fine=True, cand_per_window=args.cand)
            for (name, weight, segs, _, _, _), d in zip(asms, keep):
                eng2.add_packed_device(name, weight, d.data_ptr(), segs[:, 0], segs[:, 2])
            nk = 3
            step(eng2)
            eng2.reset_timers()
            for _ in range(nk):
                step(eng2)
            s2 = eng2.stats()
            bs = bs_route()
            sel = s2.get("select_slices", 0) > 0   # k_bs_select ran (its time is booked where the other route books count + reorder)
            ker = {"k_hash_bs" if bs else "k_hash_sparse": s2["ms_hash"] / nk,
                   ("k_bs_select" if sel else "k_bs_count+k_bs_reorder_w") if bs else "k_reorder_w": s2["ms_reorder"] / nk,
                   "k_sel_stretch+k_gap_fix+k_gap_post" if sel else "k_resolve+k_gap_fix+k_gap_post": s2["ms_resolve_kernel"] / nk,
                   "k_emit": s2["ms_emit"] / nk, "join (k_pj_* / k_insert+k_flags)": s2["ms_join"] / nk,
                   "k_vertices+k_adjacency": s2["ms_vertices"] / nk, "k_edge_flags+k_edges": s2["ms_edges"] / nk}
            tot = sum(ker.values()) or 1.0
            out["kernels"] = {"ms_per_step": {k_: round(v, 4) for k_, v in ker.items()},
                              "share": {k_: round(v / tot, 4) for k_, v in ker.items()},
                              "note": "HIP-event pair per kernel on a second handle (3 steps after the timed region); "
                                      "spans of the two streams may overlap when the assemblies are pipelined"}
            # the first step of a FRESH handle in a warm process (what a one-shot caller of the library pays: buffers, tables, no
            # hints from earlier steps; the code objects are loaded and the driver has released nothing yet)
            torch.cuda.synchronize()
            eng3 = MxEngine(k=K, w=W, device=local_rank, cand_per_window=args.cand)
            for (name, weight, segs, _, _, _), d in zip(asms, keep):
                eng3.add_packed_device(name, weight, d.data_ptr(), segs[:, 0], segs[:, 2])
            torch.cuda.synchronize()
            t1s = time.perf_counter()
            eng3.sketch(-2)
            torch.cuda.synchronize()
            t1g = time.perf_counter()
            eng3.build_graph()
            torch.cuda.synchronize()
            one_ms = (time.perf_counter() - t1s) * 1e3
            s3 = eng3.stats()
            out["one_shot"] = {"ms": round(one_ms, 3), "value": round(bases_total / (one_ms * 1e-3) / 1e9, 2), "unit": "Gbp/s",
                               "sketch_ms": round((t1g - t1s) * 1e3, 3), "graph_ms": round(one_ms - (t1g - t1s) * 1e3, 3),
                               "graph_join": hex(int(s3.get("graph_join", 0))),
                               "vs_steady_state": round(one_ms / ms_step, 2),
                               "assemblies_enqueued_twice": int(s3["retried_assemblies"]), "batches_redone": int(s3["batches_redone"]),
                               "what": "first sketch + graph step of a fresh handle on the same resident bases, timed after the steady-state "
                                       "region in the same process (code objects loaded); `fallbacks.first_step_of_the_handle` is the very first "
                                       "step of the process"}
            eng3.close()
            eng2.close()
            # every kernel group against the HBM roof: algorithmic bytes where SURVEY.md 8(d) defines them (0.25 B per base for the
            # kernel that reads the bases; per minimizer 16 B tuple written, 18 B uniqueness + intersection, 36 B edge build), the
            # counter traffic of the committed PMC passes when they are of these kernel sources, the time of the pass above
            mx = float(st["minimizers"])
            alg = {"filter": ALG_BYTES_PER_BASE_HASH * bases_total, "emit": 16.0 * mx, "join": 18.0 * mx, "graph": 36.0 * mx}
            groups = [("k_hash_bs" if bs else "k_hash_sparse", "filter", ("k_hash_bs", "k_hash_sparse")),
                      (("k_bs_select" if sel else "k_bs_count+k_bs_reorder_w") if bs else "k_reorder_w", None, ("k_bs_select", "k_bs_count", "k_bs_reorder_w", "k_reorder")),
                      ("k_sel_stretch+k_gap_fix+k_gap_post" if sel else "k_resolve+k_gap_fix+k_gap_post", None, ("k_resolve", "k_sel_stretch", "k_gap_fix", "k_gap_post")),
                      ("k_emit", "emit", ("k_emit",)),
                      ("join (k_pj_* / k_insert+k_flags)", "join", ("k_pj", "k_flags", "k_insert")),
                      ("k_vertices+k_adjacency", "graph", ("k_vertices", "k_adjacency", "k_block_prefix", "k_edge_flags", "k_edges")),
                      ("k_edge_flags+k_edges", None, ())]
            pmc = {}
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", PROFILE_ROUND, "hbm_traffic.json")))
                if tj.get("workload") == wl and tj.get("kernel_sources_digest") == kernel_sources_digest():
                    pmc = {r["kernel"]: r["bytes_per_step"] for r in tj.get("per_kernel", [])}
            except Exception:
                pmc = {}
            rbk = []
            for label_k, alg_key, names in groups:
                ms_k = ker.get(label_k, 0.0)
                if alg_key == "graph":
                    ms_k += ker.get("k_edge_flags+k_edges", 0.0)
                if label_k == "k_edge_flags+k_edges":
                    continue
                pb = sum(v for k_, v in pmc.items() if any(nm in k_ for nm in names)) if pmc else None
                ab = alg.get(alg_key) if alg_key else 0.0
                rbk.append({"kernels": label_k + (" + k_edge_flags+k_edges" if alg_key == "graph" else ""), "ms_per_step": round(ms_k, 4),
                            "alg_bytes_per_step": int(ab),
                            "alg_frac_of_hbm_peak": round(ab / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if ms_k > 0 else None,
                            "pmc_bytes_per_step": int(pb) if pb else None,
                            "pmc_frac_of_hbm_peak": round(pb / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if pb and ms_k > 0 else None})
            out["roofline_by_kernel"] = {"rows": rbk, "peak_gbs": HBM_PEAK_GBS,
                                         "note": "ms: HIP-event pair per kernel group (the `kernels` pass); alg bytes: SURVEY.md 8(d) -- none are "
                                                 "defined for turning the filter's bitmap into selected candidates or for candidate-free "
                                                 "stretches; pmc bytes: FETCH_SIZE x 2 + WRITE_SIZE of profiles/" + PROFILE_ROUND +
                                                 "/hbm_traffic.json when it was captured at these kernel sources, else null"}
            big = max(rbk, key=lambda r: r["ms_per_step"])
            out["roofline"]["largest_time_consumer"] = {"kernels": big["kernels"], "ms_per_step": big["ms_per_step"],
                                                        "see": "roofline_by_kernel"}
        asms_host = None
        if not multi and not (args.no_cpu_baseline and args.no_end_to_end):
            asms_host = [(d.cpu().numpy().view(np.uint32), st_, ln_) for d, st_, ln_ in host_layout]
        if not multi and not args.no_cpu_baseline:
            cb = cpu_baseline(asms_host, W, args.cpu_seconds, [a[1] for a in asms])
            out["cpu_baseline"] = {
                "value": round(cb["bases"] / cb["seconds"] / 1e9, 5), "unit": "Gbp/s", "cores": cb["cores"], "kind": "port",
                "cpu": cpu_model(), "logical_cpus_visible": logical_cpus(),
                "sample": f"the first {' / '.join(str(x) for x in cb['records'])} records of the assemblies "
                          f"({cb['bases'] / 1e6:.0f} Mbp, {100 * cb['frac']:.0f} % of the step's workload): C restatement of "
                          f"`indexlr -t {cb['cores']}` (records chunked, one worker per core: {cb['t_sketch']:.2f} s) + C restatement of "
                          f"read_minimizers/filter_minimizers/build_graph on arrays, 1 thread like the reference ({cb['t_graph']:.2f} s)",
                "seconds": round(cb["seconds"], 3),
                "reference_python_graph_stage": "what users run today behind indexlr: the reference's own read_minimizers + "
                                                "filter_minimizers + build_graph in Python take ~5.7 + 0.57 us per minimizer (SURVEY.md "
                                                "section 6, measured once in the build container on the reference's code: ~75 s for "
                                                "this workload's 12 M minimizers, 0.08 Gbp/s for the graph stage alone)"}
            if cb["frac"] >= 1.0:  # whole workload: same inputs -> same counts (bit-exact parity lives in tests/)
                out["parity_counts_match_cpu"] = bool(cb["minimizers"] == st["minimizers"] and
                                                      cb["vertices"] == st["vertices"] and cb["edges"] == st["edges"])
        if not multi and not args.no_end_to_end:
            td = tempfile.mkdtemp(prefix="mxg_e2e_")
            try:
                sample = asms_host
                if args.e2e_mbp > 0:
                    sample = []
                    for words, starts, lens in asms_host:
                        csum = np.cumsum(lens.astype(np.int64))
                        n = min(int(np.searchsorted(csum, args.e2e_mbp * 1e6, side="left")) + 1, len(lens))
                        sample.append((words, starts[:n], lens[:n]))
                e2e = end_to_end(sample, W, td, int(os.environ.get("MXG_BENCH_E2E_THREADS", min(n_cores(), 8))))
                t_all = e2e["t_sketch_cli"] + e2e["t_graph_cli"]
                out["end_to_end"] = {
                    "value": round(e2e["bases"] / e2e["t_one_process"] / 1e9, 4), "unit": "Gbp/s",
                    "route": "ntjoin_amd/bin/mxgraph (= `ntJoin-mx mxgraph`): FASTA -> TSVs + .mx.dot in one GPU process, cold (HIP init included); "
                             "the command forks: a worker does everything, the parent returns when every output is written and closed",
                    "seconds": round(e2e["t_one_process"], 3), "clock_stops": "when the parent returns (outputs closed; the worker's release of its HBM is not on it)",
                    "seconds_of_every_run": [round(x, 3) for x in e2e["t_one_all"]], "phases_of_every_run": e2e["phases_all"],
                    "samples": "each leg twice, every run a cold process; `seconds` / `seconds_until_worker_exit` = the faster run, every run's time beside it",
                    "seconds_until_worker_exit": round(e2e["t_attached"], 3),
                    "seconds_until_worker_exit_of_every_run": [round(x, 3) for x in e2e["t_attached_all"]],
                    "value_until_worker_exit": round(e2e["bases"] / e2e["t_attached"] / 1e9, 4),
                    "until_worker_exit_is": "the same command with MXG_NO_DETACH=1 (one process, no fork): until the process that held the GPU is gone",
                    "phases": e2e["one_process_phases"], "same_outputs_as_two_process_route": e2e["one_process_same_outputs"],
                    "two_process_value": round(e2e["bases"] / t_all / 1e9, 4),
                    "sketch_only_value": round(e2e["bases"] / e2e["t_sketch_cli"] / 1e9, 4),
                    "boundary": "FASTA text in the page cache -> <asm>.k32.w%d.tsv (--seq --pos) + out.mx.dot on disk; cold processes "
                                "(HIP init included): `indexlr` per assembly, then `python -m ntjoin_amd.run`" % W,
                    "bases": int(e2e["bases"]), "fasta_bytes": int(e2e["fasta_bytes"]), "tsv_bytes": int(e2e["tsv_bytes"]),
                    "dot_bytes": int(e2e["dot_bytes"]), "seconds_indexlr": round(e2e["t_sketch_cli"], 3),
                    "seconds_graph": round(e2e["t_graph_cli"], 3), "threads": int(os.environ.get("MXG_BENCH_E2E_THREADS", min(n_cores(), 8))),
                    "settle_s_before_each_route": E2E_SETTLE_S}
            finally:
                shutil.rmtree(td, ignore_errors=True)
        if not multi and args.workload == "auto" and not args.no_repeats and not args.mbp:
            out["repeat_rich"] = repeat_rich_side_run(args.cand)
            out["other_k"] = other_k_side_run()
        result_line = json.dumps(out)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its banner through C stdio, which is flushed at exit: flush it first so that the JSON line is
    # the LAST line on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    sys.exit(main() or 0)
