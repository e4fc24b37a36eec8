"""load time of a 100 Mbp sketch through the TSV parser vs the binary side-car: python tools/time_sidecar.py [mbp]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ntjoin_amd import synth
from ntjoin_amd.engine import MxEngine

mbp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ref, _ = synth.config2(seed=1, n_bases=mbp * 1_000_000)
d = tempfile.mkdtemp()
with MxEngine(k=32, w=1000) as eng:
    words, starts, lens = synth.pack_records(ref)
    t = torch.from_numpy(words.view(np.int32)).cuda()
    eng.add_packed_device("ref", 1.0, t.data_ptr(), starts, lens, keepalive=t)
    eng.sketch()
    n = eng.sketch_size(0)
    eng.write_tsv(0, os.path.join(d, "ref.tsv"), with_seq=True)
    eng.write_sketch_bin(0, os.path.join(d, "ref.bin"))
for kind in ("tsv", "bin", "tsv", "bin"):
    with MxEngine(k=32, w=1) as eng:
        t0 = time.perf_counter()
        (eng.add_tsv if kind == "tsv" else eng.add_bin)("ref", 1.0, os.path.join(d, "ref." + kind))
        dt = time.perf_counter() - t0
    size = os.path.getsize(os.path.join(d, "ref." + kind))
    print(f"{kind}: {n} minimizers, file {size / 1e6:.1f} MB, load {dt * 1e3:.1f} ms = {dt / n * 1e9:.0f} ns per minimizer")
