#!/usr/bin/env python3
"""one-process route against -t on the GPU box: tools/e2e_threads.py [mbp] [reps]   (fresh output names, a pause before every run)"""
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from ntjoin_amd import capi, synth

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg, asms, _ = bench.workload_tables("configs2", mbp, 1000, seed=1)
lib = capi.load()
td = tempfile.mkdtemp(prefix="mxg_thr_")
try:
    fas = []
    for i, (name, weight, segs, n_words, sub, sub_seed) in enumerate(asms):
        d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
        words = d.cpu().numpy().view(np.uint32)
        fa = os.path.join(td, ("ref.fa", "tgt.fa")[i])
        st, ln = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
        assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, st.ctypes.data, ln.ctypes.data, len(ln), b"s", 80, 8) == 0
        fas.append(fa)
        del d, words
    exe = os.path.join(REPO, "ntjoin_amd", "bin", "mxgraph")
    run = 0
    for rep in range(reps):
        for t in (8, 12, 16, 24):
            run += 1
            for f in os.listdir(td):
                if f.endswith(".tsv") or f.endswith(".dot"):
                    os.remove(os.path.join(td, f))
            time.sleep(1.0)
            t0 = time.perf_counter()
            pr = subprocess.run([exe, "-v", "-k32", "-w1000", f"-t{t}", "-p", os.path.join(td, f"o{run}"), "-s", fas[1], "-l", "1", "-r", "2", fas[0]],
                                stderr=subprocess.PIPE, text=True)
            dt = time.perf_counter() - t0
            ph = next((ln.split("mxgraph: ", 1)[1] for ln in pr.stderr.splitlines() if "device + handle" in ln), pr.stderr[-200:])
            print(f"t{t}: total {dt:.3f} s | {ph}", flush=True)
finally:
    shutil.rmtree(td, ignore_errors=True)
