"""where the file route's time goes at genome scale, and how the CPU baseline scales with threads (GPU box diagnostics):
   python tools/diag_e2e.py [mbp=3000]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ntjoin_amd import capi, synth  # noqa: E402
from tests import _oracle  # noqa: E402

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = synth.genome_config(mbp * 1_000_000, 24, seed=1, target=False)
segs = cfg["ref_segs"]
words = synth.fill_device(segs, cfg["ref_words"], cfg["seed"]).cpu().numpy().view(np.uint32)
rs, rl = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
td = tempfile.mkdtemp(prefix="diag_")
fa = os.path.join(td, "ref.fa")
t0 = time.perf_counter()
assert capi.load().mxg_synth_write_fasta(fa.encode(), words.ctypes.data, rs.ctypes.data, rl.ctypes.data, len(rl), b"s", 80, 32) == 0
print(f"wrote {os.path.getsize(fa) / 1e9:.2f} GB FASTA in {time.perf_counter() - t0:.2f} s", flush=True)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ntjoin_amd", "bin", "indexlr")
for t in (4, 16, 64):
    t0 = time.perf_counter()
    r = subprocess.run([exe, "--seq", "--long", "--pos", "-k32", "-w1000", f"-t{t}", "-v", "-o", os.path.join(td, "o.tsv"), fa],
                       capture_output=True, text=True)
    print(f"-t{t}: wall {time.perf_counter() - t0:.3f} s; {r.stderr.strip().splitlines()[0] if r.stderr.strip() else ''}", flush=True)
orc = _oracle.load()
n = 4
for th in (1, 16, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    m = n if th > 1 else 1
    t0 = time.perf_counter()
    out = orc.sketch_packed_mt(words, rs[:m], rl[:m], 32, 1000, threads=th, chunk_kmers=1 << 18)
    dt = time.perf_counter() - t0
    print(f"oracle -t{th}: {rl[:m].sum() / 1e9 / dt:.3f} Gbp/s ({dt:.2f} s, {len(out[0])} minimizers)", flush=True)
import shutil
shutil.rmtree(td, ignore_errors=True)
