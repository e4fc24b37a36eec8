"""FASTA -> packed bases in HBM (mxg_add_assembly_fasta) timed inside one warm process, four times over (the first call pays the
pinned pool and the first allocations): python tools/ingest_rate.py [mbp] [threads].  Beside tools/h2d_roof.py it says how far the
file route's ingest is from what the box can copy.  (Measured in round 4: 36 GB/s warm for a 3 GB file, 8 threads, against 57 GB/s
of a pinned upload; staging with pread instead of copies out of the mapping: 27 GB/s.)"""
import os, subprocess, sys, tempfile, time, shutil
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from ntjoin_amd.engine import MxEngine
    fa, threads = sys.argv[2], int(sys.argv[3])
    for rep in range(4):
        with MxEngine(k=32, w=1000, device=0, threads=threads) as eng:
            t0 = time.perf_counter()
            eng.add_fasta("x", 1.0, fa)
            dt = time.perf_counter() - t0
        print(f"  add_fasta: {dt:.3f} s = {os.path.getsize(fa) / dt / 1e9:.1f} GB/s", flush=True)
    sys.exit(0)
import bench
from ntjoin_amd import capi, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3000.0
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg, asms, _ = bench.workload_tables("configs2", mbp, 1000, seed=1)
lib = capi.load()
td = tempfile.mkdtemp(prefix="mxg_ing_")
try:
    name, weight, segs, n_words, sub, sub_seed = asms[0]
    d = synth.fill_device(segs, n_words, cfg["seed"], sub_seed, sub)
    words = d.cpu().numpy().view(np.uint32)
    fa = os.path.join(td, "ref.fa")
    st, ln = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
    assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, st.ctypes.data, ln.ctypes.data, len(ln), b"s", 80, 8) == 0
    del d, words
    subprocess.run([sys.executable, __file__, "--child", fa, str(threads)])
finally:
    shutil.rmtree(td, ignore_errors=True)
