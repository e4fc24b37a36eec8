#!/bin/bash
# the one-process route at -t 8 / 12 / 16, each run from a settled device: tools/e2e_threads.sh [mbp]   (GPU box)
cd "$(dirname "$0")/.."
python - "${1:-3000}" <<'PY'
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from ntjoin_amd import capi, synth
mbp = float(sys.argv[1])
cfg, asms, _ = bench.workload_tables("configs2", mbp, 1000, seed=1)
lib = capi.load()
td = tempfile.mkdtemp(prefix="mxg_thr_")
fas = []
for i, (name, weight, segs, n_words, sub, sub_seed) in enumerate(asms):
    d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
    words = d.cpu().numpy().view(np.uint32)
    fa = os.path.join(td, ("ref.fa", "tgt.fa")[i])
    st, ln = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
    assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, st.ctypes.data, ln.ctypes.data, len(ln), b"s", 80, 8) == 0
    fas.append(fa)
    del d, words
import torch
torch.cuda.empty_cache()
exe = os.path.join(os.getcwd(), "ntjoin_amd", "bin", "mxgraph")
for rep in range(2):
    for t in (8, 12, 16):
        time.sleep(1.5)
        t0 = time.perf_counter()
        pr = subprocess.run([exe, "-v", "-k32", "-w1000", f"-t{t}", "-p", os.path.join(td, "o"), "-s", fas[1], "-l", "1", "-r", "2", fas[0]], stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        ph = next((ln.split("mxgraph: ", 1)[1] for ln in pr.stderr.splitlines() if "device + handle" in ln), pr.stderr[-200:])
        print(f"-t{t}: total {dt:.3f} s | {ph}", flush=True)
shutil.rmtree(td, ignore_errors=True)
PY
