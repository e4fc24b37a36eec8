"""GPU tests of the counter-based generator (ntjoin_amd/csrc/synth.hip): the kernel, the library's host mirror and the numpy
mirror (ntjoin_amd/synth.py) produce the same bases, and an assembly born in HBM sketches to what the oracle computes from
the numpy mirror of the same coordinates."""
import numpy as np
import pytest

from ntjoin_amd import synth
from ntjoin_amd.engine import MxEngine
from tests import _oracle

pytestmark = pytest.mark.gpu


def _unpack(words, start, n):
    ww = words[start // 16: start // 16 + (n + 15) // 16]
    return ((ww[:, None] >> (np.arange(16, dtype=np.uint32) * 2)[None, :]) & 3).astype(np.uint8).ravel()[:n]


def test_device_fill_equals_host_and_numpy_mirrors():
    cfg = synth.genome_config(3_000_000, 4, seed=21, min_len=700, max_len=90_000)
    for segs, n_words, sub in ((cfg["ref_segs"], cfg["ref_words"], 0), (cfg["tgt_segs"], cfg["tgt_words"], synth.SUB_PER_65536)):
        dev = synth.fill_device(segs, n_words, cfg["seed"], cfg["sub_seed"], sub).cpu().numpy().view(np.uint32)
        host = synth.fill_host(segs, n_words, cfg["seed"], cfg["sub_seed"], sub, n_threads=4)
        assert np.array_equal(dev, host)
        for seg in segs[:: max(1, len(segs) // 7)]:
            codes = synth.segment_codes(seg, cfg["seed"], cfg["sub_seed"], sub)
            assert np.array_equal(_unpack(dev, int(seg[0]), len(codes)), codes)


@pytest.mark.parametrize("w", [100, 1000])
def test_sketch_of_generated_assembly_equals_oracle(w):
    orc = _oracle.load()
    cfg = synth.genome_config(6_000_000, 3, seed=33, min_len=3000, max_len=300_000)
    with MxEngine(k=32, w=w) as eng:
        keep = []
        for name, segs, n_words, sub in (("ref", cfg["ref_segs"], cfg["ref_words"], 0),
                                         ("tgt", cfg["tgt_segs"], cfg["tgt_words"], synth.SUB_PER_65536)):
            d = synth.fill_device(segs, n_words, cfg["seed"], cfg["sub_seed"], sub)
            keep.append(d)
            eng.add_packed_device(name, 1.0, d.data_ptr(), segs[:, 0], segs[:, 2])
        eng.sketch()
        for a, (segs, sub) in enumerate(((cfg["ref_segs"], 0), (cfg["tgt_segs"], synth.SUB_PER_65536))):
            sk = eng.get_sketch(a)
            first = sk["record_first"]
            for r in list(range(min(len(segs), 6))) + [len(segs) - 1]:
                seq = synth.to_ascii(synth.segment_codes(segs[r], cfg["seed"], cfg["sub_seed"], sub))
                want = orc.sketch(seq, 32, w)
                lo, hi = int(first[r]), int(first[r + 1])
                assert sk["out_hash"][lo:hi].tolist() == [x[0] for x in want]
                assert sk["pos"][lo:hi].tolist() == [x[1] for x in want]
        eng.build_graph()
        st = eng.stats()
        assert st["vertices"] > 0.5 * st["minimizers"] / 2 * 0.5  # the target shares most of the reference's minimizers
