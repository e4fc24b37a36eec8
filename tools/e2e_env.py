#!/usr/bin/env python3
"""A/B of the one-process route (ntjoin_amd/bin/mxgraph) under settings of the environment, on the GPU box:
   tools/e2e_env.py MBP REPS label[:VAR=value[,VAR=value...]] ...     e.g.  tools/e2e_env.py 3000 3 default old-pool:MXG_PIN_MALLOC=1
The FASTA files are written once into a temp dir; the outputs are removed between runs (an overwritten 1 GB file costs its
truncation) and every run starts one second after the last (the driver lets go of the previous process's memory)."""
import glob, os, shutil, subprocess, sys, tempfile, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from ntjoin_amd import capi, synth

mbp, reps = float(sys.argv[1]), int(sys.argv[2])
variants = []
for spec in sys.argv[3:] or ["default"]:
    label, _, kv = spec.partition(":")
    variants.append((label, dict(x.split("=", 1) for x in kv.split(",") if x)))
threads = min(bench.n_cores(), 8)
cfg, asms, _ = bench.workload_tables("configs2", mbp, 1000, seed=1)
lib = capi.load()
keep = os.environ.get("MXG_E2E_KEEP")  # a directory: the FASTA files stay there for what the caller runs next
td = keep or tempfile.mkdtemp(prefix="mxg_env_")
os.makedirs(td, exist_ok=True)
try:
    fas, bases = [], 0
    for i, (name, weight, segs, n_words, sub, sub_seed) in enumerate(asms):
        d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
        words = d.cpu().numpy().view(np.uint32)
        fa = os.path.join(td, ("ref.fa", "tgt.fa")[i])
        st, ln = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
        assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, st.ctypes.data, ln.ctypes.data, len(ln), b"s", 80, threads) == 0
        fas.append(fa)
        bases += int(ln.sum())
        del d, words
    import torch
    torch.cuda.empty_cache()
    exe = os.path.join(REPO, "ntjoin_amd", "bin", "mxgraph")
    for rep in range(reps):
        for label, env in variants:
            for f in glob.glob(os.path.join(td, "o.*")) + glob.glob(os.path.join(td, "*.tsv")):
                os.remove(f)
            time.sleep(1.0)
            t0 = time.perf_counter()
            pr = subprocess.run([exe, "-v", "-k32", "-w1000", f"-t{threads}", "-p", os.path.join(td, "o"), "-s", fas[1], "-l", "1", "-r", "2", fas[0]],
                                stderr=subprocess.PIPE, text=True, env=dict(os.environ, MXG_DEBUG_IO="1", **env))
            dt = time.perf_counter() - t0
            ph = next((ln.split("mxgraph: ", 1)[1] for ln in pr.stderr.splitlines() if "device + handle" in ln), pr.stderr[-300:])
            print(f"{label}: total {dt:.3f} s = {bases / dt / 1e9:.2f} Gbp/s | {ph}", flush=True)
            for ln in pr.stderr.splitlines():
                if ln.startswith("[mxg] load_fasta") or ln.startswith("[mxg] write_outputs: graph") or ln.startswith("[mxg] write_tsv_device") or ln.startswith("[mxg] write_dot"):
                    print("    " + ln, flush=True)
finally:
    if not keep:
        shutil.rmtree(td, ignore_errors=True)
