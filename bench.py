#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the hot path on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mbp M]

One STEP = one pass of the whole hot path over one batch of synthetic input already resident in HBM:
sketch every assembly (ntHash + window arg-min) -> uniqueness -> intersection -> adjacency edges, results
handed back through the C-ABI.  Workload = BASELINE.json configs[1]: 1 x 100 Mbp reference (weight 2) +
a target derived from it (weight 1), k=32, w=1000.  metric = Gbp/s = (sum of bases over all assemblies) / time.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling -- every rank sketches its own
100 Mbp + 100 Mbp shard of an N-times larger genome; then the graph of the WHOLE genome is built, either
"union": one RCCL all-gather of the sketches and every rank builds the whole graph (N times the graph work on every
rank, but only one collective: cheapest while a rank's share is small, as in configs[1]), or "partitioned": the graph
stage distributed by hash range over RCCL all-to-alls (ntjoin_amd/dist.py, csrc/dgraph.hip), whose work per rank does
not grow with N.  Measured on one GPU (world = 1 over RCCL), M = minimizers per rank in millions (0.4 per 100 Mbp + 100 Mbp
at w = 1000): the union path costs about 0.04 ms for the exchange + 0.14 ms x N x M for unpacking and the graph of the
union, the partitioned path about 0.39 ms + 0.2 ms x M.  Default: partitioned when M x (0.14 N - 0.2) > 0.39, i.e.
above ~1.2 Gbp per rank at N = 2, ~270 Mbp per rank at N = 4 and ~105 Mbp per rank at N = 8 (configs[1] runs the union path
at every N <= 8);
MXG_BENCH_GRAPH=union|partitioned overrides.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

K, W = 32, 1000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_BASE_HASH = 0.25  # hash kernel: one 2-bit packed base read per base


def valu_info(st):
    """integer-VALU view of the hash kernel: instructions per base and issue utilisation from the committed PMC summary
    (profiles/hash_kernel_pmc.json, tools/pmc_bench.sh), rate from this run's HIP-event time"""
    path = os.path.join(REPO, "profiles", "hash_kernel_pmc.json")
    try:
        pmc = json.load(open(path))
    except Exception:
        return None
    per_base = pmc["valu_per_base"]
    return {"wave64_instr_per_base": round(per_base, 2),
            "int_lane_ops_per_s": round(per_base * st["hash_kernel_bases"] / max(st["ms_hash"], 1e-9) * 1e3, 0),
            "peak_lane_ops_per_s": 256 * 4 * 16 * 2.4e9,  # 1024 SIMD16 units: one wave64 VALU instruction per 4 cycles
            "valu_busy_pmc": round(pmc["valu_busy"], 3),
            "source": "rocprofv3 SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / GRBM_GUI_ACTIVE, profiles/hash_kernel_pmc.json"}


def cpu_baseline(ref, tgt):
    """The oracle (scalar C port of indexlr + Python port of the graph stage) on the SAME workload, 1 core.
    Checker/baseline only: nothing here is on the product path."""
    from ntjoin_amd import synth
    from oracle import graph_oracle
    from tests import _oracle
    orc = _oracle.load()
    warm = synth.to_ascii(ref[0][:2_000_000])
    orc.sketch(warm, K, W)  # fault the allocator's pages in once
    t0 = time.perf_counter()
    sketches = []
    for recs in (ref, tgt):
        out = []
        for codes in recs:
            out.append(orc.sketch(synth.to_ascii(codes), K, W))
        sketches.append(out)
    t_sketch = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as td:
        names = [os.path.join(td, "ref.k32.w1000.tsv"), os.path.join(td, "tgt.k32.w1000.tsv")]
        for path, out in zip(names, sketches):
            with open(path, "w", encoding="ascii") as fh:
                for r, mxs in enumerate(out):
                    fh.write(f"{r}\t" + " ".join(f"{h}:{p}:N" for h, p, _, _ in mxs) + "\n")
        t1 = time.perf_counter()
        state = graph_oracle.load_and_build([names[0]], [2.0], names[1], 1.0)
        t_graph = time.perf_counter() - t1
    bases = sum(len(c) for c in ref) + sum(len(c) for c in tgt)
    n_mx = sum(len(m) for out in sketches for m in out)
    return {"seconds": t_sketch + t_graph, "t_sketch": t_sketch, "t_graph": t_graph, "bases": bases,
            "minimizers": n_mx, "vertices": len(state["vertices"]), "edges": len(state["edges"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mbp", type=float, default=100.0, help="reference size per rank in Mbp (configs[1]: 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cand", type=int, default=0, help="candidates per window for the sparse path (0 = library default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MXG_BENCH_ONE_DEVICE") == "1":  # testing: several ranks share GPU 0 (with MXG_BENCH_BACKEND=gloo)
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py: for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N ...")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("MXG_BENCH_FORCE_DIST") == "1"  # exercise the N>1 path with one rank (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("MXG_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for the one-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ntjoin_amd import synth
    from ntjoin_amd.engine import MxEngine
    from ntjoin_amd.dist import partitioned_graph, partitioned_totals, sketch_union_graph

    n_bases = int(args.mbp * 1e6)
    ref, tgt = synth.config2(seed=1 + 100 * rank, n_bases=n_bases)
    bases_rank = sum(len(c) for c in ref) + sum(len(c) for c in tgt)
    keep = []
    # N > 1: the library works on the stream the collectives are issued on, so pack -> all-gather -> unpack need no host sync
    xstream = torch.cuda.Stream() if (world > 1 or force_dist) else None
    eng = MxEngine(k=K, w=W, device=local_rank, timing=True, cand_per_window=args.cand,
                   stream=xstream.cuda_stream if xstream is not None else None)
    for name, weight, recs in (("ref.fa.k32.w1000.tsv", 2.0, ref), ("tgt.fa.k32.w1000.tsv", 1.0, tgt)):
        words, starts, lens = synth.pack_records(recs)
        d = torch.from_numpy(words.view(np.int32)).cuda()  # bases resident in HBM before the timed region
        keep.append(d)
        eng.add_packed_device(name, weight, d.data_ptr(), starts, lens)
    union = None
    m_rank = 4e-3 * args.mbp * 1000.0 / W  # minimizers per rank, millions (both assemblies, density 2/(w+1))
    graph_mode = os.environ.get("MXG_BENCH_GRAPH") or ("partitioned" if m_rank * (0.14 * world - 0.2) > 0.39 else "union")

    def step():
        nonlocal union
        if world == 1 and not force_dist and os.environ.get("MXG_BENCH_FUSED") == "1":
            eng.sketch_graph()  # sketches and graph stage in one call with one host sync (measured: no faster)
            return
        if (world > 1 or force_dist) and graph_mode != "partitioned":
            # sketch -> pack -> all-gather -> unpack -> graph of the union, one host sync per step in steady state
            union = sketch_union_graph(eng, K, W, local_rank, union, stream=xstream)
            return
        eng.sketch(-2)  # MXG_SKETCH_ALL: both assemblies enqueued back to back, one host sync
        if world > 1 or force_dist:
            union = partitioned_graph(eng, K, W, local_rank, union, stream=xstream)   # this rank's part of the graph
        else:
            eng.build_graph()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    eng.reset_timers()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt, float(bases_rank)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, bases_total = float(tmax[0]), float(t[1])
    else:
        bases_total = float(bases_rank)

    st = eng.stats()
    if union is not None and graph_mode == "partitioned":
        gst = dict(union.stats())
        gst.update(partitioned_totals(union))      # global vertex / edge counts (collectives: every rank calls)
    else:
        gst = (union.stats() if union is not None else st)
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = bases_total * args.steps / dt / 1e9
        # dominant kernel = the ntHash/candidate kernel; HIP events recorded by the library on ITS launch stream
        launches = max(st["launches_hash"], 1)
        avg_ms = st["ms_hash"] / launches
        bytes_per_launch = ALG_BYTES_PER_BASE_HASH * st["hash_kernel_bases"] / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                # measured on the 100 Mbp launch of configs[1]; the kernel's traffic is linear in the bases of a launch
                traffic = round(json.load(open(tpath)).get("k_hash_bytes_per_base") * st["hash_kernel_bases"] / launches)
            except Exception:
                traffic = None
        # the committed rocprofv3 kernel-trace summary of this same command, for comparison with the live HIP-event time
        # (the events also wait while the other assembly's kernels hold the machine: two streams)
        rocprof_ms = None
        try:
            import csv
            with open(os.path.join(REPO, "profiles", "r01_bench_kernel_stats.csv"), newline="") as fh:
                for row in csv.DictReader(fh):
                    if "k_hash_sparse" in row["Name"]:
                        rocprof_ms = round(float(row["AverageNs"]) / 1e6, 4)
        except Exception:
            rocprof_ms = None
        out = {
            "metric": "Gbp/s minimizer-sketch+graph-build (k=32,w=1000)", "value": round(value, 4), "unit": "Gbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic 1x{args.mbp:g} Mbp reference + derived target per GPU, "
                                   "k=32 w=1000, weights 2/1, bases resident in HBM (2-bit packed)",
                       "k": K, "w": W, "bases_per_step": int(bases_total), "minimizers": int(st["minimizers"]),
                       "vertices": int(gst["vertices"]), "edges": int(gst["edges"]),
                       "parallelism": ("1 GPU" if world == 1 else f"contig-sharded x{world}, " +
                                       ("graph stage partitioned by hash range (RCCL all-to-all)" if graph_mode == "partitioned"
                                        else "RCCL all-gather of sketches, graph of the union on every rank"))},
            "roofline": {"bound": "hbm", "kernel": "k_hash (ntHash fwd/rc rolling + candidate filter)",
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/hbm_traffic.json)",
                         "alg_bytes_per_base": ALG_BYTES_PER_BASE_HASH, "alg_bytes_per_launch": int(bytes_per_launch),
                         "avg_launch_ms": round(avg_ms, 4),
                         "avg_launch_ms_rocprofv3": rocprof_ms if abs(args.mbp - 100.0) < 1e-9 else None,
                         "launches": int(st["launches_hash"]),
                         "bases_per_launch": int(st["hash_kernel_bases"] / launches)},
            # the binding resource is integer VALU issue, not HBM: reported beside the HBM figure
            "valu": valu_info(st),
            "stage_ms_per_step": {"hash": round(st["ms_hash"] / args.steps, 4),
                                  "graph": round(gst["ms_graph"] / args.steps, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(ref, tgt)
            out["cpu_baseline"] = {
                "value": round(cb["bases"] / cb["seconds"] / 1e9, 5), "unit": "Gbp/s", "cores": 1, "kind": "port",
                "sample": f"the whole step workload ({cb['bases'] / 1e6:.0f} Mbp): scalar C port of indexlr "
                          f"({cb['t_sketch']:.1f} s) + Python port of read/filter/build_graph ({cb['t_graph']:.1f} s)",
            }
            # same inputs -> same counts (full bit-exact parity lives in tests/)
            out["parity_counts_match_cpu"] = bool(cb["minimizers"] == st["minimizers"] and
                                                  cb["vertices"] == st["vertices"] and cb["edges"] == st["edges"])
        result_line = json.dumps(out)
    else:
        result_line = None
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its banner through C stdio, which is flushed at exit: flush it first so that the JSON line is
    # the LAST line on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)



if __name__ == "__main__":
    main()
