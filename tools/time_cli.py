"""where the end-to-end time of the file route goes (FASTA -> sketch -> TSV): python tools/time_cli.py [mbp]"""
import os
import random
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
from ntjoin_amd.engine import MxEngine

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = tempfile.mkdtemp()
fa = os.path.join(d, "g.fa")
rng = random.Random(1)
with open(fa, "w") as f:
    n = mbp * 1_000_000
    f.write(">chr1\n")
    block = "".join(rng.choices("ACGT", k=1_000_000))
    for i in range(mbp):
        s = block[i % 977:] + block[:i % 977]
        for j in range(0, len(s), 60):
            f.write(s[j:j + 60] + "\n")
size = os.path.getsize(fa)
t0 = time.perf_counter()
eng = MxEngine(k=32, w=1000)
t1 = time.perf_counter()
eng.add_fasta("g", 1.0, fa)
t2 = time.perf_counter()
eng.sketch()
t3 = time.perf_counter()
eng.write_tsv(0, os.path.join(d, "g.tsv"), with_pos=True, with_strand=False, with_seq=True)
t4 = time.perf_counter()
print(f"{mbp} Mbp FASTA ({size / 1e6:.0f} MB): create {1e3 * (t1 - t0):.0f} ms, add_fasta (read+pack+upload) {1e3 * (t2 - t1):.0f} ms "
      f"= {size / 1e6 / (t2 - t1):.0f} MB/s, sketch {1e3 * (t3 - t2):.1f} ms, write_tsv {1e3 * (t4 - t3):.0f} ms "
      f"({os.path.getsize(os.path.join(d, 'g.tsv')) / 1e6:.1f} MB)")
eng.close()
exe = os.path.join("ntjoin_amd", "bin", "indexlr")
t0 = time.perf_counter()
subprocess.check_call([exe, "--seq", "--long", "--pos", "-k32", "-w1000", "-t4", "-o", os.path.join(d, "cli.tsv"), fa])
print(f"indexlr CLI wall {time.perf_counter() - t0:.2f} s")
