"""CPU tests of the boundary: the C-ABI library loads, exports every declared symbol, formats text like
python's repr(), and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import random
import re

import pytest

from tests.conftest import REPO


def _lib():
    from ntjoin_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return capi, capi.load()


def test_every_declared_symbol_is_exported():
    capi, lib = _lib()
    header = open(os.path.join(REPO, "include", "ntjoin_mx.h"), encoding="utf-8").read()
    declared = set(re.findall(r"\b(mxg_[a-z_0-9]+)\s*\(", header))
    declared -= {"mxg_create"} if False else set()
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ntjoin_mx.h but not exported"
    assert set(capi.SYMBOLS) == declared
    assert lib.mxg_abi_version() == capi.ABI_VERSION


def test_struct_sizes_match_header():
    capi, _ = _lib()
    # mxg_config: 6*u32 + ptr + u32 + 5*u32 (with natural alignment) ; mxg_stats layout is mirrored field by field
    assert C.sizeof(capi.Config) == 56
    assert C.sizeof(capi.Stats) == 8 + 8 * 8 + 3 * 8 + 2 * 8 + 8 * 8 + 6 * 8


def test_py_repr_double_matches_python():
    _, lib = _lib()
    rng = random.Random(9)
    vals = [0.0, 1.0, 2.0, 3.0, 1.5, 0.1, 0.2, 0.1 + 0.2, 0.30000000000000004, 100000.0, 1e15, 1e16, 1e17, 1.5e16,
            123456789012345680.0, 1e-4, 1e-5, 0.0001234, 5e-324, 1.7976931348623157e308, -2.5, 1e22, 1e21, 2.0 ** 53,
            0.1 + 0.2 + 1.5, 3.3000000000000003, float("inf")]
    vals += [rng.uniform(-10, 10) for _ in range(200)] + [rng.random() * 10 ** rng.randint(-10, 25) for _ in range(200)]
    vals += [float(rng.randint(0, 50)) / 4 for _ in range(50)]
    buf = C.create_string_buffer(64)
    for v in vals:
        n = lib.mxg_py_repr_double(v, buf, 64)
        assert buf.value.decode() == repr(v) and n == len(repr(v)), v


def test_py_repr_str_matches_python():
    _, lib = _lib()
    buf = C.create_string_buffer(256)
    for s in ["test", "1_f", "chr'1", 'chr"1', "a'b\"c", "back\\slash", "tab\there", "", "ctgA0_f", "naïve", "x\x01y"]:
        lib.mxg_py_repr_str(s.encode("utf-8"), buf, 256)
        assert buf.value.decode("utf-8") == repr(s), s


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK), reason="a GPU is visible")
def test_no_cpu_fallback():
    """without a HIP device the engine refuses to start: nothing silently routes to a CPU path"""
    from ntjoin_amd.engine import MxEngine, MxError
    from ntjoin_amd import capi
    with pytest.raises(MxError) as ei:
        MxEngine(k=32, w=100)
    assert ei.value.code == capi.MXG_EDEVICE


def test_product_does_not_import_oracle():
    """the package never references oracle/ (the oracle is the checker, never the shipped path)"""
    pkg = os.path.join(REPO, "ntjoin_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(root, f), encoding="utf-8").read()
                assert "mx_oracle" not in text and "graph_oracle" not in text and "import oracle" not in text, f


def test_shard_range_partitions_records():
    """contig sharding rule (host only): contiguous ranges, every record in exactly one shard, balanced by bases"""
    import numpy as np
    from ntjoin_amd.dist import shard_range
    rng = np.random.default_rng(5)
    for lens in ([10], [5, 5], [0, 0, 7], list(rng.integers(1, 10**6, size=200)),
                 [250_000_000, 50_000_000, 240_000_000, 60_000_000] * 6, []):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(lens, s, world) for s in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == len(lens)
            for (lo, hi), (lo2, _) in zip(ranges, ranges[1:]):
                assert lo <= hi == lo2
            if len(lens) >= 50 * world:
                loads = [sum(int(x) for x in lens[lo:hi]) for lo, hi in ranges]
                assert max(loads) <= sum(loads) / world + max(int(x) for x in lens)


def test_plan_split_pieces_and_continuation_bit():
    """sub-record sharding plan (host only): the shards' own k-mer starts tile every record; piece_drop bit 0 = the piece opens with
    a halo of w k-mers and withholds its first minimizer, bit 1 = the record began on an earlier shard -- also when the cut
    falls inside the record's first w k-mers, where the halo reaches back to base 0 (the FASTA split route's `cont`)"""
    import numpy as np
    from ntjoin_amd import capi
    L = capi.load()
    k, w = 32, 100

    def plan(lens, s, n):
        ln = np.ascontiguousarray(lens, dtype=np.uint64)
        lo, hi = np.zeros(len(ln), dtype=np.uint64), np.zeros(len(ln), dtype=np.uint64)
        dr = np.zeros(len(ln), dtype=np.uint8)
        assert L.mxg_plan_split(ln.ctypes.data, len(ln), s, n, k, w, lo.ctypes.data, hi.ctypes.data, dr.ctypes.data) == 0
        return lo, hi, dr

    # two shards of one record of 1000 bases + a short one: the cut (base 550) is far from both ends
    lo, hi, dr = plan([1000, 100], 1, 2)
    assert (int(lo[0]), int(hi[0]), int(dr[0])) == (550 - w, 1000, 3) and (int(lo[1]), int(hi[1]), int(dr[1])) == (0, 100, 0)
    lo, hi, dr = plan([1000, 100], 0, 2)
    assert (int(lo[0]), int(hi[0]), int(dr[0])) == (0, 550 + k - 1, 0) and int(hi[1]) == 0
    # the cut 60 bases into the second record: fewer than w k-mers before it -> the piece starts at base 0, nothing withheld,
    # but the record did begin on the shard before
    lens = [940, 1060]
    lo, hi, dr = plan(lens, 1, 2)
    assert int(hi[0]) == 0 and (int(lo[1]), int(hi[1]), int(dr[1])) == (0, 1060, 2)
    lo0, hi0, dr0 = plan(lens, 0, 2)
    assert (int(lo0[1]), int(hi0[1]), int(dr0[1])) == (0, 60 + k - 1, 0)
    # random plans: every base range is owned exactly once, bits consistent
    rng = np.random.default_rng(3)
    for _ in range(30):
        lens = rng.integers(1, 5000, size=int(rng.integers(1, 40)))
        n = int(rng.integers(1, 9))
        total = int(lens.sum())
        for s in range(n):
            lo, hi, dr = plan(lens, s, n)
            cut_lo = total * s // n
            starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
            for r in range(len(lens)):
                if hi[r] > lo[r]:
                    began_before = starts[r] < cut_lo
                    assert bool(dr[r] & 2) == bool(began_before)
                    assert not (dr[r] & 1) or (dr[r] & 2)
                    assert not lo[r] > 0 or (dr[r] & 1)  # a piece that does not start at base 0 opens with a halo
                else:
                    assert dr[r] == 0


def test_ntjoin_constructor_needs_w_for_fasta():
    """sketching FASTA with the TSV route's placeholder w=1 would write a huge, wrong checkpoint TSV"""
    import argparse
    import pytest
    from ntjoin_amd.ntjoin import Ntjoin
    with pytest.raises(ValueError):
        Ntjoin(argparse.Namespace(FILES=[], s="t.tsv", l=1, p="o", k=32), fasta={"t.tsv": "t.fa"})
