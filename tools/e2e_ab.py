#!/usr/bin/env python3
"""A/B of the one-process route's output writers on the GPU box: tools/e2e_ab.py [mbp] [reps]
(FASTA files written once into a temp dir, ntjoin_amd/bin/mxgraph -v run `reps` times per setting of MXG_NO_MMAP_OUT)"""
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from ntjoin_amd import capi, synth

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
threads = min(bench.n_cores(), 8)
cfg, asms, _ = bench.workload_tables("configs2", mbp, 1000, seed=1)
lib = capi.load()
td = tempfile.mkdtemp(prefix="mxg_ab_")
try:
    fas = []
    for i, (name, weight, segs, n_words, sub, sub_seed) in enumerate(asms):
        d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
        words = d.cpu().numpy().view(np.uint32)
        fa = os.path.join(td, ("ref.fa", "tgt.fa")[i])
        st, ln = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
        assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, st.ctypes.data, ln.ctypes.data, len(ln), b"s", 80, threads) == 0
        fas.append(fa)
        del d, words
    exe = os.path.join(REPO, "ntjoin_amd", "bin", "mxgraph")
    os.system("df -h /tmp /dev/shm | cat")
    for rep in range(reps):
        for label, extra, prefix, t in (("tmp t8", [], os.path.join(td, "o"), 8), ("tmp t8 no-tsv", ["--no-tsv"], os.path.join(td, "o"), 8),
                                       ("shm t8 no-tsv", ["--no-tsv"], "/dev/shm/mxg_o", 8), ("tmp t16 no-tsv", ["--no-tsv"], os.path.join(td, "o"), 16),
                                       ("null t8 no-tsv", ["--no-tsv"], "/dev/null", 8)):
            t0 = time.perf_counter()
            w0 = time.time()
            pr = subprocess.run([exe, "-v", "-k32", "-w1000", f"-t{t}", "-p", prefix, "-s", fas[1], "-l", "1", "-r", "2", fas[0]] + extra,
                                stderr=subprocess.PIPE, text=True)
            dt = time.perf_counter() - t0
            w1 = time.time()
            ph = next((ln.split("mxgraph: ", 1)[1] for ln in pr.stderr.splitlines() if "device + handle" in ln), pr.stderr[-200:])
            print(f"{label}: total {dt:.3f} s | {ph}", flush=True)
            for ln in pr.stderr.splitlines():
                if ln.startswith("[mxg] write") or ln.startswith("[mxg] statistics") or ln.startswith("[mxg] load_fasta"):
                    print("    " + ln, flush=True)
                if ln.startswith("[mxg] main() entered"):
                    print(f"    spawn -> main(): {float(ln.split()[4]) - w0:.3f} s", flush=True)
                if ln.startswith("[mxg] leaving main()"):
                    print(f"    end of main() -> parent has the exit status: {w1 - float(ln.split()[4]):.3f} s", flush=True)
    for f in ("/dev/shm/mxg_o.mx.dot",):
        if os.path.exists(f):
            os.remove(f)
finally:
    shutil.rmtree(td, ignore_errors=True)
