"""GPU test of the per-assembly exchange buffers (mxg_sketch_pack_parts -> mxg_xchg_unpack_graph_parts) in ONE process: a part
carries hash + position per minimizer and the FIRST ENTRY of every record instead of a record column, so the receiver's record
column is rebuilt by bisection -- here on assemblies whose records are long, too short for any minimizer, and empty of
minimizers in runs at the front, in the middle and at the end; "two ranks" = the same part twice with a record shift."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(rng, pattern):
    """pattern: list of lengths (0 < len < k: no k-mer at all)"""
    return [(f"r{i}", "".join("ACGT"[c] for c in rng.integers(0, 4, n))) for i, n in enumerate(pattern)]


@pytest.mark.parametrize("world", [1, 2, 3])
def test_parts_round_trip_with_records_that_have_no_minimizer(world):
    import torch
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(11)
    k, w = 32, 1000      # (the route whose sketches are packed on the device; w = 100 travels as "sizes first")
    big, small = 1_500_000, 20
    pats = [[small, small, big, small, big, big, small, small, small, big, small],   # runs without entries: front, middle, end
            [big] + [small] * 37 + [big, 300_000] + [small] * 5,
            [small] * 3]                                                             # an assembly without any minimizer
    with MxEngine(k=k, w=w) as eng, MxEngine(k=k, w=w) as ref, MxEngine(k=k, w=w) as union:
        for a, pat in enumerate(pats):
            recs = _records(rng, pat)
            eng.add_records(f"a{a}", 1.0 + a, recs)
            ref.add_records(f"a{a}", 1.0 + a, recs)
            ids = [f"q{q}:{rid}" for q in range(world) for rid, _ in recs]
            union.add_minimizers(f"a{a}", 1.0 + a, np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), ids)
        ref.sketch(-2)
        want = [ref.get_sketch(a) for a in range(len(pats))]
        assert len(want[0]["out_hash"]) > 5_000 and len(want[2]["out_hash"]) == 0
        assert len(np.unique(want[0]["record"])) == 4 and len(np.unique(want[1]["record"])) == 3
        caps = [(len(s["out_hash"]) * 11 // 10 + 64 + 7) // 8 * 8 for s in want]
        rcaps = [(len(p) + 3) // 4 * 4 for p in pats]
        nbytes = [64 + 12 * c + 4 * rc for c, rc in zip(caps, rcaps)]
        send = [torch.zeros(n, dtype=torch.uint8, device="cuda") for n in nbytes]
        for _step in range(2):   # (twice: the second time over buffers that hold the first step's tables)
            eng.sketch_pack_parts([t.data_ptr() for t in send], caps, rcaps)
            torch.cuda.synchronize()
            eng.sketch_finish()
            recv = [t.repeat(world) for t in send]           # every "rank" sent the same part
            rec_off = np.concatenate([[q * len(p) for q in range(world)] for p in pats]).astype(np.uint64)
            assert union.xchg_unpack_graph_parts([t.data_ptr() for t in recv], world, caps, rcaps, rec_off)
            for a, s in enumerate(want):
                got = union.get_sketch(a)
                n = len(s["out_hash"])
                assert len(got["out_hash"]) == world * n, a
                for q in range(world):
                    assert np.array_equal(got["out_hash"][q * n:(q + 1) * n], s["out_hash"]), (a, q)
                    assert np.array_equal(got["pos"][q * n:(q + 1) * n], s["pos"]), (a, q)
                    assert np.array_equal(got["record"][q * n:(q + 1) * n], s["record"] + q * len(pats[a])), (a, q)
            for a, s in enumerate(want):                     # the sender's own sketches are complete after sketch_finish
                mine = eng.get_sketch(a)
                assert np.array_equal(mine["out_hash"], s["out_hash"]) and np.array_equal(mine["record"], s["record"])


def test_parts_say_so_when_the_record_table_is_too_small():
    """rcaps below the sender's number of records: the part travels as "does not fit" (-1), the receiver reports it (False)"""
    import torch
    from ntjoin_amd.engine import MxEngine
    rng = np.random.default_rng(5)
    recs = _records(rng, [1_000_000] * 6)
    with MxEngine(k=32, w=1000) as eng, MxEngine(k=32, w=1000) as union:
        eng.add_records("a", 1.0, recs)
        union.add_minimizers("a", 1.0, np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint32), [r for r, _ in recs])
        caps, rcaps = [64_000], [4]
        send = [torch.zeros(64 + 12 * caps[0] + 4 * rcaps[0], dtype=torch.uint8, device="cuda")]
        eng.sketch_pack_parts([send[0].data_ptr()], caps, rcaps)
        torch.cuda.synchronize()
        eng.sketch_finish()
        assert int(send[0][:8].view(torch.int64)[0]) == -1
        assert not union.xchg_unpack_graph_parts([send[0].data_ptr()], 1, caps, rcaps, np.zeros(1, np.uint64))
        assert eng.sketch_size(0) > 5_000   # (the sender's own sketch is complete all the same)
