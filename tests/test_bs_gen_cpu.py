"""The generated bit-sliced ring filter (ntjoin_amd/csrc/gen/bs_gen.py -> csrc/hash_bs_k32.inc) checked on the CPU:

  * the generator's numpy VM executes the instruction list it emits (64 lanes x 32-bit registers, the same operand banks) on
    random packed bases -- from the lane's 64 packed words through the in-register bit transposes to the candidate bitmap -- and must reproduce `reference_bits`, the plain restatement of the ring test, word for word -- including the
    word in front of a chunk, the prefetch of the next chunk's planes and thresholds at both ends of the range;
  * `reference_bits` itself is pinned to the oracle: every k-mer whose canonical hash (oracle: reference ntHash)
    is < tau must pass the test (the filter may let more through, never fewer), and what it lets through beyond them stays
    within the few percent the design states;
  * the committed .inc is what the generator emits now (a stale generated file would ship an unchecked kernel).
"""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "ntjoin_amd", "csrc", "gen"))
import bs_gen as G  # noqa: E402


def _vm_chunk(seed, tt, c, n_chunks=3, **gen_kw):
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, n_chunks * G.CHUNK).astype(np.uint8)
    packed = G.pack_chunks(codes, n_chunks)
    g = G.Gen(32, **gen_kw)
    g.chunk()
    c_next = min(c + 1, n_chunks - 1)
    vm = G.VM(packed, tt, c, c_next)
    out = vm.run(g)[:2048]  # (the VM's last word belongs to the next chunk's first slot)
    ext = np.concatenate([codes, np.zeros(64, dtype=np.uint8)])  # (bases behind the assembly read as A, like k_bs_transpose)
    ref = G.reference_bits(ext[:n_chunks * G.CHUNK + 31], 32, tt)
    # the chunk writes the words [c * 2048 - 1, c * 2048 + 2047) of the position bitmap
    lo = c * 2048 - 1
    want = np.zeros(2048, dtype=np.uint64)
    sh = np.arange(32, dtype=np.uint64)
    for wi in range(2048):
        gi = lo + wi
        if gi >= 0:
            want[wi] = int((ref[gi * 32: gi * 32 + 32].astype(np.uint64) << sh).sum())
    first = 1 if lo < 0 else 0
    assert np.array_equal(np.asarray(out[first:], dtype=np.uint64), want[first:])
    # the words left in the W registers are the next chunk's (requested while this one was computed): lane L holds words 64 L ..,
    # and the two words in front of them are in the running-Q registers
    lanes = np.arange(64)
    for i in range(64):
        assert np.array_equal(vm.vr[f"v{G.W0 + i}"], packed[c_next, 64 * lanes + i]), i
    flat = np.concatenate([np.zeros(2, dtype=np.uint32), packed.reshape(-1)])
    for j in (0, 1):
        assert np.array_equal(vm.vr[f"v{int(g.Qr[0][1:]) + j}"], flat[c_next * 4096 + 64 * lanes + j])
    return float(ref.mean())


@pytest.mark.parametrize("seed,tt,c", [(1, 164, 1), (2, 40, 0), (3, (1 << G.B_PLANES) - 1, 2), (4, 0, 1)])
def test_vm_matches_reference_bits(seed, tt, c):
    dens = _vm_chunk(seed, tt, c)
    if tt == (1 << G.B_PLANES) - 1:
        assert dens == 1.0  # every sum is <= the largest threshold
    else:
        expect = (tt + 3) / float(1 << G.B_PLANES)  # St in {-2, -1, 0 .. tt}
        assert abs(dens - expect) < 0.15 * expect + 2e-4


@pytest.mark.parametrize("kw", [{"perm16": ("in", "out")}, {"perm16": ("out",)}, {"ablate": ("nowarmpairs",)}])
def test_vm_with_other_generator_options(kw):
    """generator options that were measured and not shipped (stage 16 of the transposes as v_perm_b32: 800 instructions fewer
    and slower, profiles/r06/filter_lds16_ab.txt) or that the shipped stream replaced (one warm-up step per pass over the ring)
    still produce the same bitmap"""
    _vm_chunk(7, 164, 1, **kw)


@pytest.mark.parametrize("lds16", [("in",), ("out",), ("in", "out")])
def test_vm_with_the_transposes_stage_16_through_lds(lds16):
    """the generator's variant that exchanges half words through LDS (measured in round 6 and not shipped: profiles/r06/
    filter_lds16_ab.txt) computes the same bitmap"""
    _vm_chunk(5, 131, 1, lds16=lds16)
    g = G.Gen(32, lds16=lds16)
    g.chunk()
    assert sum(1 for i in g.ins if i[0] in ("dsw16", "dsw16hi", "dsr32")) == 6 * 16 * (2 * ("in" in lds16) + ("out" in lds16))


def test_instruction_classes_and_banks():
    """only instructions of the class that issues at full rate on gfx950 (profiles/ubench/README.md), and three-register
    v_bitop3 operands in three different register banks"""
    g = G.Gen(32)
    g.chunk()
    ops = {i[0] for i in g.ins}
    valu = {"xor", "and", "or", "mov", "bitop3", "add", "lshr"}
    # besides them: loads, waits and the marker where the results of the chunk before are stored -- nothing of the slow class
    # (... and the LDS stores / loads of the transposes' stage 16: they are not VALU instructions)
    assert ops - valu <= {"gload4", "gload2", "waitcnt", "comment", "prev_stores", "dsw16", "dsw16hi", "dsr32"}
    assert {i[0] for i in g.store_ins} == {"gstore1", "gstore3", "gstore4"}
    g.check_banks()
    n_valu = sum(1 for i in g.ins if i[0] in valu)
    inc = open(os.path.join(REPO, "ntjoin_amd", "csrc", "hash_bs_k32.inc")).read()
    assert f"#define HASH_BS_VALU_PER_CHUNK {n_valu}\n" in inc
    assert f"#define HASH_BS_VGPR_END {G.VEND}\n" in inc
    assert G.VEND <= 256  # two waves per SIMD need <= 256 VGPRs each


def test_committed_inc_is_current(tmp_path):
    p = tmp_path / "hash_bs_k32.inc"
    G.emit_inc(str(p), 32)
    assert p.read_text() == open(os.path.join(REPO, "ntjoin_amd", "csrc", "hash_bs_k32.inc")).read()


def test_planes_in_registers():
    """the in-register transposes: after planes_in, plane (t, beta) holds bit beta of the base at (32 lane + s) * 32 + t in bit s,
    and QL / QH hold the strip in front of the lane's first one (previous lane's last strip; lane 0: the words in front of the chunk)"""
    rng = np.random.default_rng(9)
    codes = rng.integers(0, 4, 2 * G.CHUNK).astype(np.uint8)
    packed = G.pack_chunks(codes, 2)
    for c in (0, 1):
        g = G.Gen(32)
        g.planes_in()
        vm = G.VM(packed, 0, c, c)
        vm.run(g)
        for _ in range(300):
            lane, s_, t, be = (int(rng.integers(0, n)) for n in (64, 32, 32, 2))
            p = c * G.CHUNK + (32 * lane + s_) * 32 + t
            assert (int(vm.vr[g.W[(t, be)]][lane]) >> s_) & 1 == (int(codes[p]) >> be) & 1
            q = c * G.CHUNK + (32 * lane - 1) * 32 + t
            reg = (g.QL if t < 16 else g.QH)[be]
            want = (int(codes[q]) >> be) & 1 if q >= 0 else 0
            assert (int(vm.vr[reg][lane]) >> (t % 16)) & 1 == want
        for be in (0, 1):  # nothing above bit 15
            assert not (vm.vr[g.QL[be]] >> np.uint32(16)).any() and not (vm.vr[g.QH[be]] >> np.uint32(16)).any()


@pytest.mark.parametrize("cand_per_window,w", [(10, 1000), (10, 200), (18, 500), (2, 500)])
def test_reference_bits_is_a_superset_of_the_oracle(oracle, cand_per_window, w):
    """tau as the library sets it (sparse_plan: tau_hi = even(frac * 2^32), tau = tau_hi << 32; bs_hash: T = tau_hi / 2,
    tt = (T - 1) >> 17): no k-mer with hash < tau may fail the ring test"""
    rng = np.random.default_rng(cand_per_window * 1000 + w)
    n = 400_000
    codes = rng.integers(0, 4, n).astype(np.uint8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
    mh, _, _, ok = oracle.kmer_hashes(seq, 32)
    assert ok.all()
    frac = cand_per_window / w
    tau_hi = max(2, int(min(4294967294.0, frac * 4294967296.0)) & ~1)
    tau = tau_hi << 32
    t31 = tau_hi >> 1
    tt = (t31 - 1) >> (31 - G.B_PLANES)
    bits = G.reference_bits(codes, 32, tt)
    real = mh < np.uint64(tau)
    assert real.sum() > 0
    assert not (real & ~bits).any()
    extra = bits.sum() / real.sum() - 1.0
    # two extra sums (-2, -1) and the rounding of T up to a multiple of 2^17: a few percent at the library's densities
    assert extra < (3.0 + (1 << (31 - G.B_PLANES)) / t31 * (tt + 1) - 0.0) / (tt + 1) + 0.05
