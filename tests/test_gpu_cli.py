"""GPU tests of the process boundary: the indexlr-compatible binary and the make front-end."""
import filecmp
import os
import shutil
import subprocess

import pytest

from oracle import graph_oracle as go
from tests.conftest import BIN_DIR, GOLDEN, REPO, load_case

pytestmark = pytest.mark.gpu
INDEXLR = os.path.join(BIN_DIR, "indexlr")
FASTA = os.path.join(GOLDEN, "fasta")


def test_indexlr_stdout_matches_golden_tsv(tmp_path):
    """the reference recipe, verbatim: indexlr --seq --long --pos -k K -w W -t T X.fa > X.fa.kK.wW.tsv"""
    case = load_case("synth_w100")["meta"]
    for a in case["refs"] + [case["target"]]:
        out = tmp_path / a["tsv"]
        with open(out, "wb") as fh:
            subprocess.check_call([INDEXLR, "--seq", "--long", "--pos", "-k", "32", "-w", "100", "-t", "4",
                                   os.path.join(FASTA, a["fasta"])], stdout=fh)
        assert filecmp.cmp(str(out), os.path.join(GOLDEN, "cases", "synth_w100", a["tsv"]), shallow=False)


def test_indexlr_glued_options_and_o(tmp_path):
    """run_indexlr()'s spelling (reference bin/ntjoin_utils.py:198): -k32 -w100 -t4 ... -o file"""
    out = tmp_path / "o.tsv"
    subprocess.check_call([INDEXLR, os.path.join(FASTA, "ref.fa"), "--seq", "--long", "--pos", "-k32", "-w100", "-t4",
                           "-o", str(out)])
    assert filecmp.cmp(str(out), os.path.join(GOLDEN, "cases", "f-f_w100_config1", "ref.fa.k32.w100.tsv"), shallow=False)
    # hash-only and strand columns
    r = subprocess.run([INDEXLR, "-k", "32", "-w", "1000", "--variant", "v1", "--pos", os.path.join(FASTA, "ref.fa")],
                       check=True, capture_output=True)
    want = open(os.path.join(GOLDEN, "reference_expected_outputs", "ref.fa.k32.w1000.tsv"), "rb").read()
    assert r.stdout == want  # the reference's own (stale-format) golden file, bit for bit
    r = subprocess.run([INDEXLR, "-k", "32", "-w", "1000", "--pos", "--strand", os.path.join(FASTA, "ref.fa")],
                       check=True, capture_output=True)
    fields = r.stdout.decode().split("\t")[1].split()[0].split(":")
    assert len(fields) == 3 and fields[2] in "+-"


def test_indexlr_o_into_a_fifo_and_dev_stdout(tmp_path):
    """-o names something that is not a regular file (a FIFO a consumer reads, /dev/stdout): the text is written in order at
    the descriptor's own position (no pwrite), and the name is never unlinked -- not even when the run fails"""
    import threading
    want = open(os.path.join(GOLDEN, "cases", "f-f_w100_config1", "ref.fa.k32.w100.tsv"), "rb").read()
    fifo = tmp_path / "out.fifo"
    os.mkfifo(fifo)
    got = {}

    def reader():
        with open(fifo, "rb") as fh:
            got["data"] = fh.read()
    t = threading.Thread(target=reader)
    t.start()
    subprocess.check_call([INDEXLR, os.path.join(FASTA, "ref.fa"), "--seq", "--long", "--pos", "-k32", "-w100", "-o", str(fifo)])
    t.join(timeout=60)
    assert got.get("data") == want
    assert os.path.exists(fifo) and not os.path.isfile(fifo)
    r = subprocess.run([INDEXLR, os.path.join(FASTA, "ref.fa"), "--seq", "--long", "--pos", "-k32", "-w100", "-o", "/dev/stdout"],
                       check=True, capture_output=True)
    assert r.stdout == want
    assert os.path.islink("/dev/stdout")
    # a failing run leaves the FIFO in place (a reader must be there for the open to return)
    t = threading.Thread(target=reader)
    t.start()
    r = subprocess.run([INDEXLR, str(tmp_path / "missing.fa"), "-k32", "-w100", "-o", str(fifo)], capture_output=True)
    if t.is_alive():  # (the run failed before it opened the FIFO: release the reader)
        with open(fifo, "wb"):
            pass
    t.join(timeout=60)
    assert r.returncode != 0 and os.path.exists(fifo)


def test_indexlr_failure_is_loud(tmp_path):
    r = subprocess.run([INDEXLR, "-k", "32", "-w", "100", str(tmp_path / "missing.fa")], capture_output=True)
    assert r.returncode != 0 and r.stdout == b"" and b"cannot open" in r.stderr
    r = subprocess.run([INDEXLR, "-w", "100", os.path.join(FASTA, "ref.fa")], capture_output=True)
    assert r.returncode == 2 and r.stdout == b""


def test_make_front_end_mxgraph(tmp_path):
    """ntJoin's variable surface: target= references= reference_weights= k= w= prefix= ; same file naming"""
    case = load_case("f-f-f_w100_weights")
    meta = case["meta"]
    for a in meta["refs"] + [meta["target"]]:
        shutil.copy(os.path.join(FASTA, a["fasta"]), tmp_path / a["fasta"])
    refs = " ".join(a["fasta"] for a in meta["refs"])
    wts = " ".join(str(a["weight"]) for a in meta["refs"])
    subprocess.check_call(["make", "-f", os.path.join(REPO, "ntJoin-mx"), "mxgraph", f"target={meta['target']['fasta']}",
                           f"target_weight={meta['target']['weight']}", f"references={refs}",
                           f"reference_weights={wts}", "k=32", "w=100", "prefix=out"], cwd=tmp_path)
    for a in meta["refs"] + [meta["target"]]:
        assert filecmp.cmp(str(tmp_path / a["tsv"]), os.path.join(GOLDEN, "cases", meta["name"], a["tsv"]), shallow=False)
    with open(os.path.join(GOLDEN, "cases", meta["name"], "reference.mx.dot"), encoding="utf-8") as fh:
        want = go.canonical_dot_from_text(fh.read())
    assert go.canonical_dot_from_text((tmp_path / "out.mx.dot").read_text(encoding="utf-8")) == want


@pytest.mark.parametrize("case_name", ["f-f-f_w100_weights", "f-f_w1000", "f-f_termN_w1000", "gap-dist_w500"])
def test_one_process_route_is_byte_identical_to_the_two_process_route(tmp_path, case_name):
    """`ntJoin-mx mxgraph` in one process (ntjoin_amd/bin/mxgraph: FASTA -> TSVs + .mx.dot, sketches never leave HBM) against the
    reference's two steps (`indexlr` per assembly, then the graph stage on the parsed TSVs): the same bytes in every TSV and in
    the .mx.dot"""
    meta = load_case(case_name)["meta"]
    outs = {}
    for mode in ("True", "False"):
        d = tmp_path / mode
        d.mkdir()
        for a in meta["refs"] + [meta["target"]]:
            shutil.copy(os.path.join(FASTA, a["fasta"]), d / a["fasta"])
        refs = " ".join(a["fasta"] for a in meta["refs"])
        wts = " ".join(str(a["weight"]) for a in meta["refs"])
        subprocess.check_call(["make", "-f", os.path.join(REPO, "ntJoin-mx"), "mxgraph", f"target={meta['target']['fasta']}",
                               f"target_weight={meta['target']['weight']}", f"references={refs}", f"reference_weights={wts}",
                               f"k={meta['k']}", f"w={meta['w']}", "prefix=out", f"mx_one_process={mode}"], cwd=d)
        outs[mode] = d
    for a in meta["refs"] + [meta["target"]]:
        assert filecmp.cmp(str(outs["True"] / a["tsv"]), str(outs["False"] / a["tsv"]), shallow=False)
    assert filecmp.cmp(str(outs["True"] / "out.mx.dot"), str(outs["False"] / "out.mx.dot"), shallow=False)
    assert os.path.getsize(outs["True"] / "out.mx.dot") > 100


def test_mxgraph_file_route_knobs_give_the_same_bytes(tmp_path):
    """the one-process route's file handling, old and new: the pinned pool registered in pieces (default) or taken in one
    allocation (MXG_PIN_MALLOC=1), the FASTA mappings left to the end of the process (default for the one-shot handle) or taken
    apart at once (MXG_UNMAP_EARLY=1), buffers kept (MXG_KEEP_BUFFERS=1), a clean exit through mxg_destroy (MXG_CLEAN_EXIT=1):
    the same TSVs and the same .mx.dot; a file of several staging buffers (> 4 x 32 MB) goes through the ring of pieces"""
    import numpy as np
    rng = np.random.default_rng(17)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    fa_r, fa_t = tmp_path / "r.fa", tmp_path / "t.fa"
    with open(fa_r, "wb") as fr, open(fa_t, "wb") as ft:
        for i in range(70):   # 70 x 2.3 Mbp = 163 MB of text: five staging buffers and a bit
            seq = lut[rng.integers(0, 4, size=2_300_000, dtype=np.uint8)]
            lines = np.full((23_000, 101), ord("\n"), dtype=np.uint8)
            lines[:, :100] = seq.reshape(23_000, 100)
            fr.write(f">chr{i}\n".encode() + lines.tobytes())
            if i % 2 == 0:   # the target: a few contigs of every other record
                for c in range(3):
                    ft.write(f">ctg{i}_{c}\n".encode() + seq[c * 700_000:c * 700_000 + 650_000].tobytes() + b"\n")
    exe = os.path.join(BIN_DIR, "mxgraph")
    outs = {}
    for tag, env in (("default", {}), ("malloc", {"MXG_PIN_MALLOC": "1"}), ("unmap", {"MXG_UNMAP_EARLY": "1"}),
                     ("keep", {"MXG_KEEP_BUFFERS": "1"}), ("clean", {"MXG_CLEAN_EXIT": "1", "MXG_NO_DETACH": "1"})):
        subprocess.run([exe, "-k32", "-w1000", "-t4", "-p", str(tmp_path / tag), "-s", str(fa_t), "-r", "2", str(fa_r)], check=True,
                       env=dict(os.environ, **env))
        outs[tag] = tuple(open(f, "rb").read() for f in (str(tmp_path / tag) + ".mx.dot", str(fa_r) + ".k32.w1000.tsv", str(fa_t) + ".k32.w1000.tsv"))
        os.remove(str(fa_r) + ".k32.w1000.tsv")
        os.remove(str(fa_t) + ".k32.w1000.tsv")
    assert len(outs["default"][0]) > 10_000 and len(outs["default"][1]) > 1_000_000
    for tag in outs:
        assert outs[tag] == outs["default"], tag


def test_mxgraph_cli_failures_are_loud(tmp_path):
    exe = os.path.join(BIN_DIR, "mxgraph")
    fa = tmp_path / "a.fa"
    fa.write_text(">x\n" + "ACGT" * 100 + "\n")
    r = subprocess.run([exe, "-k", "32", "-w", "10", "-s", str(fa), "-r", "1 2", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "one weight per reference" in r.stdout          # the reference's check and exit status
    r = subprocess.run([exe, "-k", "32", "-w", "10", "-s", str(tmp_path / "missing.fa"), "-r", "1", str(fa)], capture_output=True, text=True)
    assert r.returncode == 1 and "missing.fa" in r.stderr
    assert not (tmp_path / "out.mx.dot").exists()
    r = subprocess.run([exe, "-k", "32"], capture_output=True, text=True)
    assert r.returncode == 2


def test_mxgraph_worker_and_parent(tmp_path):
    """mxgraph does its work in a child and leaves as soon as the child reports its outputs complete (the driver's teardown of the
    child is nobody's wait): same bytes and the same statuses as in one process (MXG_NO_DETACH=1), output pipes released at once, a
    worker that dies without a word reported by its status"""
    import signal
    import time
    from tests.conftest import GOLDEN, load_case
    meta = load_case("f-f_w100_config1")["meta"]
    fasta_dir = os.path.join(GOLDEN, "fasta")
    exe = os.path.join(BIN_DIR, "mxgraph")
    outs = {}
    for mode, env in (("detached", {}), ("one", {"MXG_NO_DETACH": "1"})):
        d = tmp_path / mode
        d.mkdir()
        fas = []
        for a in meta["refs"] + [meta["target"]]:
            shutil.copy(os.path.join(fasta_dir, a["fasta"]), d / a["fasta"])
            fas.append(a["fasta"])
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-v", "-k", str(meta["k"]), "-w", str(meta["w"]), "-p", "out", "-s", fas[-1], "-l", str(meta["target"]["weight"]),
                            "-r", " ".join(str(a["weight"]) for a in meta["refs"])] + fas[:-1], cwd=d, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "vertices" in r.stderr, r.stderr[-500:]     # (-v statistics arrive before the parent leaves)
        outs[mode] = {f: (d / f).read_bytes() for f in ["out.mx.dot"] + [a["tsv"] for a in meta["refs"] + [meta["target"]]]}
        assert time.perf_counter() - t0 < 120
    assert outs["detached"] == outs["one"]
    if os.environ.get("MXG_NO_DETACH"):   # (the sanitizer runs keep mxgraph in one process: no worker to kill)
        return
    # a worker killed before it reports: the parent returns 128 + signal (MXG_TEST_WORKER_SIGNAL: the worker raises it at its start)
    r = subprocess.run([exe, "-k", "32", "-w", "10", "-s", "x.fa", "-r", "1", "y.fa"], cwd=tmp_path,
                       env=dict(os.environ, MXG_TEST_WORKER_SIGNAL=str(int(signal.SIGKILL))), capture_output=True, text=True, timeout=60)
    assert r.returncode == 128 + int(signal.SIGKILL)


def test_python_mirror_functions(tmp_path):
    """read_minimizers / filter_minimizers / build_graph counterparts return what the reference's returned"""
    from ntjoin_amd import ntjoin_utils as nu
    case = load_case("gap-dist_w500")
    meta, ref = case["meta"], case["reference"]
    cdir = os.path.join(GOLDEN, "cases", meta["name"])
    os.chdir(cdir)
    list_mxs, weights = {}, {}
    for a in meta["refs"] + [meta["target"]]:
        info, mxs = nu.read_minimizers(a["tsv"])
        assert {m: list(v) for m, v in info.items()} == ref["mx_info"][a["tsv"]]
        assert mxs == ref["mxs"][a["tsv"]]
        list_mxs[a["tsv"]] = mxs
        weights[a["tsv"]] = float(a["weight"])
    filt = nu.filter_minimizers(list_mxs)
    assert filt == ref["filtered"]
    g = nu.build_graph(filt, weights)
    mine = {frozenset((s, t)): (sup, w) for s, t, sup, w in g.edge_list_named()}
    theirs = {frozenset((s, t)): (sup, w) for s, t, sup, w in ref["edges"]}
    assert mine == theirs and sorted(g.names, key=int) == ref["vertices"]
    # error behaviour: an entry without exactly three fields raises ValueError, as the reference's unpack does
    bad = tmp_path / "bad.tsv"
    bad.write_text("ctg\t123:45\n")
    with pytest.raises(ValueError):
        nu.read_minimizers(str(bad))


def test_many_small_calls_in_one_pass():
    """filter_minimizers_many / build_graph_many: the overlap stage's thousands of calls on two lists of ~15 minimizers
    (reference bin/ntjoin_overlap.py:28,132) taken together -- the same results as the calls one at a time and as the oracle's
    restatement of the reference's functions, including hashes that occur in several items, duplicates inside a list, empty
    lists, and items of more than two assemblies"""
    import random
    from ntjoin_amd import ntjoin_utils as nu
    from oracle import graph_oracle as go
    rng = random.Random(int(os.environ.get("MXG_FUZZ_SEED", "5")))
    pool = [str(rng.getrandbits(64)) for _ in range(400)]
    items = []
    for i in range(max(60, int(os.environ.get("MXG_FUZZ_TRIALS", "60")))):
        a_n = 2 if i % 9 else 3
        base = rng.sample(pool, rng.randint(0, 25))
        item = {}
        for a in range(a_n):
            lst = [m for m in base if rng.random() < 0.8] + rng.sample(pool, rng.randint(0, 4))
            rng.shuffle(lst)
            if lst and rng.random() < 0.15:
                lst.insert(rng.randrange(len(lst) + 1), lst[0])     # a duplicate inside the list
            cut = rng.randint(0, len(lst))
            item[f"asm{a}_{i}"] = [lst[:cut], lst[cut:]] if rng.random() < 0.3 else [lst]
        items.append(item)
    many = nu.filter_minimizers_many(items)
    assert len(many) == len(items)
    for it, got in zip(items, many):
        assert got == go.filter_minimizers(it)
        assert list(got.keys()) == list(it.keys())
    assert many[:5] == [nu.filter_minimizers(it) for it in items[:5]]
    # graphs: on what the filter returned, minus duplicates (build_graph's contract: every minimizer once per assembly)
    clean, weights = [], []
    for got in many:
        dup = {m for lists in got.values() for lst in lists for m in lst if sum(l.count(m) for l in lists) > 1}
        clean.append({a: [[m for m in lst if m not in dup] for lst in lists] for a, lists in got.items()})
        weights.append({a: float(rng.choice([1, 2, 1.5])) for a in got})
    graphs = nu.build_graph_many(clean, weights)
    for it, w_i, g in zip(clean, weights, graphs):
        want_vertices, want_edges = go.build_edges(it, w_i)
        mine = {frozenset((s, t)): (sup, w) for s, t, sup, w in g.edge_list_named()}
        assert mine == {frozenset((s, t)): (sup, w) for s, t, sup, w in want_edges}
        assert set(g.names) == want_vertices
    one = nu.build_graph(clean[3], weights[3])
    assert one.edge_list_named() == graphs[3].edge_list_named() and one.names == graphs[3].names
    same_w = {f"slot{a}": 1.0 for a in range(2)}
    pairs = [{"slot0": it[list(it)[0]], "slot1": it[list(it)[1]]} for it in clean if len(it) == 2][:6]
    assert [g.edge_list_named() for g in nu.build_graph_many(pairs, same_w)] == [nu.build_graph(p, same_w).edge_list_named() for p in pairs]


def test_array_backed_views_equal_reference_on_all_goldens():
    """read_minimizers(views=True) and Ntjoin.make_minimizer_graph(materialize="views") (what a genome-scale run uses instead of
    dicts of Python strings) hold exactly what the reference's own functions returned, for every golden case"""
    import argparse
    from ntjoin_amd import ntjoin_utils as nu
    from ntjoin_amd.ntjoin import Ntjoin
    from tests.conftest import golden_cases
    cwd = os.getcwd()
    for m in golden_cases():
        case = load_case(m["name"])
        meta, ref = case["meta"], case["reference"]
        os.chdir(os.path.join(GOLDEN, "cases", meta["name"]))
        try:
            for a in meta["refs"] + [meta["target"]]:
                info, mxs = nu.read_minimizers(a["tsv"], k=meta["k"], views=True)
                assert {k_: list(v) for k_, v in info.items()} == ref["mx_info"][a["tsv"]], (m["name"], a["tsv"])
                assert mxs.to_lists() == ref["mxs"][a["tsv"]]
            args = argparse.Namespace(FILES=[r["tsv"] for r in meta["refs"]], s=meta["target"]["tsv"], l=meta["target"]["weight"],
                                      p=os.path.join("/tmp", "views_" + meta["name"]), k=meta["k"], n=1)
            nj = Ntjoin(args, variant=meta.get("variant", "v2"))
            nj.weights_list = [float(r["weight"]) for r in meta["refs"]]
            try:
                nj.load_minimizers_scaffold()
                nj.make_minimizer_graph(materialize="views")
                for a in ref["assemblies"]:
                    assert {k_: list(v) for k_, v in nj.list_mx_info[a].items()} == ref["mx_info"][a]
                    assert nj.list_mxs[a].to_lists() == ref["mxs"][a]
                mine = {frozenset((s, t)): (sup, w) for s, t, sup, w in nj.graph.edge_list_named()}
                theirs = {frozenset((s, t)): (sup, w) for s, t, sup, w in ref["edges"]}
                assert mine == theirs and sorted(nj.graph.names, key=int) == ref["vertices"]
            finally:
                nj.close()
        finally:
            os.chdir(cwd)


def test_tsv_parsed_by_several_threads(tmp_path):
    """the TSV parser cuts the file at line ends into one piece per worker: same sketch, ids, order and error lines as one worker"""
    import numpy as np
    from ntjoin_amd.engine import MxEngine, MxError
    rng = np.random.default_rng(3)
    path = tmp_path / "big.tsv"
    want = []
    with open(path, "w") as fh:
        for r in range(60_000):
            n = int(rng.integers(0, 12))
            ents = [(int(rng.integers(0, 2**63)), int(rng.integers(0, 10**8))) for _ in range(n)]
            fh.write(f"ctg{r}\t" + " ".join(f"{h}:{p}:ACGT" for h, p in ents) + "\n")
            if n:
                want.append((f"ctg{r}", ents))
    assert os.path.getsize(path) > 4 << 20
    res = []
    for t in (1, 7):
        with MxEngine(k=32, w=1, threads=t) as eng:
            eng.add_tsv("a", 1.0, str(path))
            sk = eng.get_sketch(0)
            res.append(sk)
            first = sk["record_first"]
            assert sk["record_ids"] == [w_[0] for w_ in want]
            for r in (0, 1, len(want) // 2, len(want) - 1):
                lo, hi = int(first[r]), int(first[r + 1])
                assert list(zip(sk["out_hash"][lo:hi].tolist(), sk["pos"][lo:hi].tolist())) == want[r][1]
    for key in ("out_hash", "pos", "record", "record_first"):
        assert np.array_equal(res[0][key], res[1][key])
    with open(path, "a") as fh:
        fh.write("broken\t12:34\n")
    with MxEngine(k=32, w=1, threads=5) as eng:
        with pytest.raises(MxError) as ei:
            eng.add_tsv("a", 1.0, str(path))
        assert f":{60_001}:" in str(ei.value) and "three" in str(ei.value)


def test_make_j_concurrent_indexlr_instances_and_log_time(tmp_path):
    """B1 convention: make runs one `indexlr` per assembly, concurrently under `make -j` (one GPU context each on the same
    device); with time=True every target gets its `.time` file (reference ntJoin:98-107,205)"""
    import stat
    meta = load_case("synth3_w50")["meta"]
    asms = meta["refs"] + [meta["target"]]
    for a in asms:
        shutil.copy(os.path.join(FASTA, a["fasta"]), tmp_path / a["fasta"])
    bind = tmp_path / "bin"
    bind.mkdir()
    fake_time = bind / "time"   # GNU time is not installed here: a stand-in with its `-v -o FILE cmd...` interface
    fake_time.write_text('#!/bin/bash\nout=""; while [ "$1" = "-v" ] || [ "$1" = "-o" ]; do if [ "$1" = "-o" ]; then out="$2"; shift; fi; shift; done\n'
                         'echo "ran: $*" > "$out"; exec "$@"\n')
    fake_time.chmod(fake_time.stat().st_mode | stat.S_IXUSR)
    env = dict(os.environ, PATH=f"{bind}:{os.environ['PATH']}")
    refs = " ".join(a["fasta"] for a in meta["refs"])
    wts = " ".join(str(a["weight"]) for a in meta["refs"])
    subprocess.check_call(["make", "-j", "8", "-f", os.path.join(REPO, "ntJoin-mx"), "mxgraph", f"target={meta['target']['fasta']}",
                           f"target_weight={meta['target']['weight']}", f"references={refs}", f"reference_weights={wts}",
                           f"k={meta['k']}", f"w={meta['w']}", "prefix=out", "time=True", "mx_one_process=False"], cwd=tmp_path, env=env)
    for a in asms:
        assert filecmp.cmp(str(tmp_path / a["tsv"]), os.path.join(GOLDEN, "cases", meta["name"], a["tsv"]), shallow=False)
        assert (tmp_path / (a["tsv"] + ".time")).read_text().startswith("ran: ")
    assert "ntjoin_amd.run" in (tmp_path / "out.mx.dot.time").read_text()
    with open(os.path.join(GOLDEN, "cases", meta["name"], "reference.mx.dot"), encoding="utf-8") as fh:
        want = go.canonical_dot_from_text(fh.read())
    assert go.canonical_dot_from_text((tmp_path / "out.mx.dot").read_text(encoding="utf-8")) == want
    # the same instances started at the same moment by hand
    procs = [subprocess.Popen([INDEXLR, "--seq", "--long", "--pos", f"-k{meta['k']}", f"-w{meta['w']}", "-t2", "-o", str(tmp_path / f"c{i}.tsv"),
                               str(tmp_path / asms[i % len(asms)]["fasta"])]) for i in range(6)]
    assert [p.wait() for p in procs] == [0] * 6
    for i in range(6):
        assert filecmp.cmp(str(tmp_path / f"c{i}.tsv"), os.path.join(GOLDEN, "cases", meta["name"], asms[i % len(asms)]["tsv"]), shallow=False)


def test_mxgraph_back_to_back_and_a_gpu_job_right_behind_the_parent(tmp_path):
    """`mxgraph` at genome scale (MXG_TEST_E2E_MBP, default 1000 Mbp per assembly: the worker holds ~5 GB of HBM + its pinned buffers
    when the parent returns): two runs back to back and, the instant the first parent is back, a third GPU process (an `indexlr` of
    the same reference) -- while the first run's worker is still handing its memory back to the driver.  No failure, no OOM, and
    every run's outputs are the same bytes; the attached run (MXG_NO_DETACH=1) writes them too."""
    import hashlib
    import time
    import numpy as np
    from ntjoin_amd import capi, synth
    if "clang_rt" in os.environ.get("LD_PRELOAD", "") or os.environ.get("MXG_NO_DETACH"):
        pytest.skip("a sanitizer run (tools/asan_run.sh): torch, which makes this test's genomes, cannot start under the preloaded runtime, "
                    "and mxgraph stays in one process there")
    mbp = float(os.environ.get("MXG_TEST_E2E_MBP", "1000"))
    w = 1000
    lib = capi.load()
    fas = []
    for i, (name, seed) in enumerate((("ref.fa", 5), ("tgt.fa", 6))):
        lens = np.full(12 if i == 0 else 4000, int(mbp * 1e6) // (12 if i == 0 else 4000), dtype=np.uint64)
        segs, n_words, _ = synth.reference_segments(lens)
        words = synth.fill_device(segs, n_words, seed).cpu().numpy().view(np.uint32)
        fa = str(tmp_path / name)
        rs, rl = np.ascontiguousarray(segs[:, 0]), np.ascontiguousarray(segs[:, 2])
        assert lib.mxg_synth_write_fasta(fa.encode(), words.ctypes.data, rs.ctypes.data, rl.ctypes.data, len(rl), b"c", 80, 16) == 0
        del words
        fas.append(fa)
    import torch
    torch.cuda.empty_cache()
    exe, ilr = os.path.join(BIN_DIR, "mxgraph"), os.path.join(BIN_DIR, "indexlr")

    def run(prefix, env=None):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-k32", f"-w{w}", "-t8", "-p", str(tmp_path / prefix), "-s", fas[1], "-l", "1", "-r", "2", fas[0]],
                           env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return time.perf_counter() - t0

    def digest(prefix):
        out = []
        for f in [str(tmp_path / (prefix + ".mx.dot"))] + [f"{fa}.k32.w{w}.tsv" for fa in fas]:
            hsh = hashlib.sha256()
            with open(f, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    hsh.update(blk)
            out.append(hsh.hexdigest())
        return out

    t_first = run("a")
    # the instant the parent is back: another GPU process, and the second run right behind it
    third = subprocess.Popen([ilr, "--long", "--pos", "-k32", f"-w{w}", "-t8", "-o", str(tmp_path / "third.tsv"), fas[0]], stderr=subprocess.PIPE)
    d_a = digest("a")
    t_second = run("b")
    assert third.wait(timeout=900) == 0, third.stderr.read()[-2000:]
    d_b = digest("b")
    assert d_a == d_b
    t_attached = run("c", {"MXG_NO_DETACH": "1"})
    assert digest("c") == d_a
    # the third process's sketch of the reference = the reference's TSV without the sequence column's absence mattering: same line count
    with open(tmp_path / "third.tsv", "rb") as fh:
        n_third = sum(1 for _ in fh)
    with open(f"{fas[0]}.k32.w{w}.tsv", "rb") as fh:
        n_ref = sum(1 for _ in fh)
    assert n_third == n_ref == 12
    print(f"mxgraph at {mbp:g} + {mbp:g} Mbp: detached {t_first:.2f} s, again {t_second:.2f} s, attached {t_attached:.2f} s")
