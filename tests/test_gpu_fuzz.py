"""GPU differential fuzz: random (k, w, variant, candidates-per-window, strip length) x random ragged / N-heavy /
low-complexity records, sparse path vs the CPU oracle, bit for bit.  MXG_FUZZ_TRIALS raises the trial count."""
import os
import random

import pytest

from tests import _oracle

pytestmark = pytest.mark.gpu


def _rand_record(rng, n):
    kind = rng.random()
    if kind < 0.5:
        return "".join(rng.choice("ACGT") for _ in range(n))
    if kind < 0.65:
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 9)))
        return (unit * (n // len(unit) + 1))[:n]
    if kind < 0.85:
        s = [rng.choice("ACGT") for _ in range(n)]
        for _ in range(rng.randint(1, 6)):
            p = rng.randrange(max(n, 1))
            ln = rng.choice([1, 1, 3, 40, 700])
            s[p:p + ln] = "N" * len(s[p:p + ln])
        return "".join(s)
    s = [rng.choice("ACGT") for _ in range(n)]
    p = rng.randrange(max(n, 1))
    ln = rng.randint(0, n)
    s[p:p + ln] = [rng.choice("Aa") for _ in s[p:p + ln]]
    return "".join(s)


def test_fuzz_sparse_path_vs_oracle(oracle):
    from ntjoin_amd.engine import MxEngine
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "60"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "2024")))
    saved = os.environ.get("MXG_SPARSE_S")
    try:
        for t in range(trials):
            k = rng.choice([4, 11, 15, 16, 21, 31, 32, 33, 47, 64])
            w = rng.choice([16, 50, 100, 200, 333, 1000, 2000])
            variant = rng.choice(["v2", "v2", "v1"])
            c = rng.choice([1, 2, 4, 8, 16, 16, 24]) if w >= 200 else rng.choice([1, 2])
            os.environ["MXG_SPARSE_S"] = str(rng.choice([16, 32, 64, 128, 256, 512]))
            recs = [(f"r{i}", _rand_record(rng, rng.choice([0, 5, k + w - 2, k + w - 1, 500, 3000, 9000, 40000])))
                    for i in range(rng.randint(1, 8))]
            with MxEngine(k=k, w=w, variant=variant, cand_per_window=c) as eng:
                eng.add_records("x", 1.0, recs)
                eng.sketch()
                sk = eng.get_sketch(0)
            ov = _oracle.V1_MIN if variant == "v1" else _oracle.V2_SUM
            for r, (rid, seq) in enumerate(recs):
                lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
                want = oracle.sketch(seq, k, w, ov)
                got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist(), sk["forward"][lo:hi].tolist()))
                assert got == [(h, p, f) for h, p, f, _ in want], (t, k, w, variant, c, os.environ["MXG_SPARSE_S"], rid)
    finally:
        if saved is None:
            os.environ.pop("MXG_SPARSE_S", None)
        else:
            os.environ["MXG_SPARSE_S"] = saved


def test_fuzz_k32_route_vs_oracle(oracle):
    """the same at k = 32, the route ntJoin's default takes (bit-sliced filter over whole chunks of 65 536 positions, bitmap ->
    ordered candidates, stretches on the device or left to the tile kernel): records around and beyond the chunk size, chunk
    borders inside runs of N and inside low-complexity sequence, several chained batches, both stretch routes"""
    from ntjoin_amd.engine import MxEngine
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "100"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "3232")))
    knobs = ("MXG_SPARSE_S", "MXG_SPARSE_BATCH_KMERS", "MXG_DEV_GAPS")
    saved = {k_: os.environ.get(k_) for k_ in knobs}
    try:
        for t in range(trials):
            w = rng.choice([150, 200, 333, 500, 1000, 2000])
            c = rng.choice([2, 4, 8, 10, 16, 18])
            while c / w > 0.125:
                c //= 2
            os.environ["MXG_SPARSE_S"] = str(rng.choice([64, 128, 320, 512]))
            os.environ["MXG_SPARSE_BATCH_KMERS"] = str(rng.choice([40_000, 150_000, 10**9]))
            os.environ["MXG_DEV_GAPS"] = str(rng.choice([0, 1]))
            recs = [(f"r{i}", _rand_record(rng, rng.choice([31, 32, 33, 32 + w - 2, 32 + w - 1, 5000, 65536 - 31, 65536, 65537, 70000, 140000])))
                    for i in range(rng.randint(1, 6))]
            with MxEngine(k=32, w=w, cand_per_window=c) as eng:
                eng.add_records("x", 1.0, recs)
                eng.sketch()
                sk = eng.get_sketch(0)
                st = eng.stats()
            assert st["bs_filter_bases"] == sum(len(s_) for _, s_ in recs) or st["kmers"] == 0 or not any(len(s_) >= 32 + w - 1 for _, s_ in recs) \
                or st["bs_filter_bases"] == 0
            for r, (rid, seq) in enumerate(recs):
                lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
                want = oracle.sketch(seq, 32, w, _oracle.V2_SUM)
                got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist(), sk["forward"][lo:hi].tolist()))
                assert got == [(h, p, f) for h, p, f, _ in want], (t, w, c, {k_: os.environ[k_] for k_ in knobs}, rid, len(seq))
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v


def _graph_trial(rng, t, A):
    """one random set of A sketches through mxg_add_assembly_tsv against the graph oracle"""
    from ntjoin_amd.engine import MxEngine
    from oracle import graph_oracle as go
    universe = [rng.getrandbits(64) for _ in range(rng.choice([5, 30, 200, 2000]))]
    names, weights = [], []
    for a in range(A):
        name = f"t{t}_asm{a}.k32.w10.tsv"
        with open(name, "w", encoding="ascii") as fh:
            for r in range(rng.randint(1, 12)):
                n = rng.choice([0, 0, 1, 2, 5, 40, 300])
                picks = [rng.choice(universe) for _ in range(n)] if rng.random() < 0.5 else \
                    rng.sample(universe, min(n, len(universe)))
                pos = sorted(rng.sample(range(10 ** 6), len(picks)))
                fh.write(f"ctg{r}\t" + " ".join(f"{h}:{p}:ACGT" for h, p in zip(picks, pos)) + "\n")
        names.append(name)
        weights.append(rng.choice([1, 2, 0.5, 1.5, 0.1, 3]))
    with MxEngine(k=32, w=10) as eng:
        for nm, wt in zip(names, weights):
            eng.add_tsv(nm, wt, nm)
        eng.build_graph()
        eng.write_dot(f"t{t}.mx.dot")
        flags = [eng.get_mx_flags(a) for a in range(A)]
        sks = [eng.get_sketch(a) for a in range(A)]
    state = go.load_and_build(names[:-1], weights[:-1], names[-1], weights[-1])
    got = go.canonical_dot_from_text(open(f"t{t}.mx.dot", encoding="utf-8").read())
    assert got == go.canonical_dot_from_state(state), (t, A)
    for a, nm in enumerate(names):
        uniq = {str(h) for h, f in zip(sks[a]["out_hash"].tolist(), flags[a].tolist()) if f & 1}
        assert uniq == set(state["list_mx_info"][nm].keys()), (t, A, a)


def test_fuzz_graph_stage_vs_oracle(tmp_path):
    """random sketches fed through mxg_add_assembly_tsv: duplicates inside and across assemblies, empty records,
    1-6 assemblies, fractional weights -> flags, filtered lists, edges, weights and canonical .mx.dot vs the oracle"""
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "60"))
    rng = random.Random(77)
    os.chdir(tmp_path)
    for t in range(trials):
        _graph_trial(rng, t, rng.randint(1, 6))


@pytest.mark.parametrize("n_asm", [4, 5, 8, 9, 13])
def test_fuzz_graph_stage_by_number_of_assemblies(tmp_path, n_asm):
    """the edge kernels know three cases -- up to four assemblies (their adjacency entries of a vertex requested together), up to
    eight (an edge's support mask travels in its flag byte), more (the mask is looked up again): each against the oracle"""
    rng = random.Random(1000 + n_asm)
    os.chdir(tmp_path)
    for t in range(int(os.environ.get("MXG_FUZZ_TRIALS", "60")) // 6):
        _graph_trial(rng, t, n_asm)


def _derived_assembly(rng, base, sub):
    """contigs cut out of `base` (some reverse-complemented, substitutions at rate `sub`, a run of N or a duplicated piece now
    and then): shares most minimizers with it, repeats some inside itself"""
    comp = str.maketrans("ACGTacgtN", "TGCAtgcaN")
    recs, p, i = [], 0, 0
    while p < len(base):
        ln = rng.choice([40, 900, 5000, 30000, 70000, 140000])
        seg = list(base[p:p + ln])
        p += ln + rng.choice([0, 25, 300])
        for q in range(len(seg)):
            if rng.random() < sub:
                seg[q] = rng.choice("ACGT")
        if rng.random() < 0.3 and len(seg) > 3000:
            a = rng.randrange(len(seg) - 2000)
            seg[a:a + rng.choice([1, 50, 1500])] = "N" * len(seg[a:a + rng.choice([1, 50, 1500])])
        if rng.random() < 0.3 and len(seg) > 8000:  # the same 3 kbp twice: minimizers that are not unique in the assembly
            a, b = rng.randrange(len(seg) - 3000), rng.randrange(len(seg) - 3000)
            seg[b:b + 3000] = seg[a:a + 3000]
        s = "".join(seg)
        if rng.random() < 0.5:
            s = s.translate(comp)[::-1]
        recs.append((f"c{i}", s))
        i += 1
    rng.shuffle(recs)
    return recs


def test_fuzz_whole_path_k32_vs_oracle(oracle, tmp_path):
    """sketch -> uniqueness -> intersection -> edges on the device, k = 32 route, 2-4 related assemblies with low-complexity
    stretches, N runs and repeated pieces, batches and stretch routes forced small: every record's minimizers and the canonical
    .mx.dot against the oracle (the graph oracle reads the TSVs this engine wrote, whose records are checked first)"""
    from ntjoin_amd.engine import MxEngine
    from oracle import graph_oracle as go
    trials = int(os.environ.get("MXG_FUZZ_TRIALS", "40"))
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "515")))
    knobs = ("MXG_SPARSE_S", "MXG_SPARSE_BATCH_KMERS", "MXG_DEV_GAPS")
    saved = {k_: os.environ.get(k_) for k_ in knobs}
    os.chdir(tmp_path)
    try:
        for t in range(trials):
            w = rng.choice([100, 200, 500, 1000])
            c = rng.choice([4, 8, 10, 18])
            while c / w > 0.125:
                c //= 2
            os.environ["MXG_SPARSE_S"] = str(rng.choice([64, 128, 320, 512]))
            os.environ["MXG_SPARSE_BATCH_KMERS"] = str(rng.choice([60_000, 200_000, 10**9]))
            os.environ["MXG_DEV_GAPS"] = str(rng.choice([0, 1]))
            base = "".join(_rand_record(rng, rng.choice([20000, 70000, 140000])) for _ in range(rng.randint(1, 3)))
            A = rng.randint(2, 4)
            asms = [[("chr", base)]] + [_derived_assembly(rng, base, rng.choice([0.0, 0.002, 0.01])) for _ in range(A - 1)]
            names = [f"t{t}_a{a}.fa.k32.w{w}.tsv" for a in range(A)]
            weights = [rng.choice([1, 2, 0.5, 1.5]) for _ in range(A)]
            fused = rng.random() < 0.5
            with MxEngine(k=32, w=w, cand_per_window=c) as eng:
                for nm, wt, recs in zip(names, weights, asms):
                    eng.add_records(nm, wt, recs)
                if fused:
                    eng.sketch_graph()
                else:
                    eng.sketch()
                for a, recs in enumerate(asms):
                    sk = eng.get_sketch(a)
                    for r, (rid, seq) in enumerate(recs):
                        lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
                        want = oracle.sketch(seq, 32, w, _oracle.V2_SUM)
                        got = list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist()))
                        assert got == [(h, p) for h, p, _, _ in want], (t, w, c, {k_: os.environ[k_] for k_ in knobs}, a, rid, len(seq))
                    eng.write_tsv(a, names[a])
                if not fused:
                    eng.build_graph()
                eng.write_dot(f"t{t}.mx.dot")
            state = go.load_and_build(names[:-1], weights[:-1], names[-1], weights[-1])
            got = go.canonical_dot_from_text(open(f"t{t}.mx.dot", encoding="utf-8").read())
            assert got == go.canonical_dot_from_state(state), (t, w, c, fused, {k_: os.environ[k_] for k_ in knobs})
    finally:
        for k_, v in saved.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
