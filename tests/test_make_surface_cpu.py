"""CPU tests of the make front-end: `ntJoin-mx assemble` hands the job to the reference's own ntJoin with this
repository's `indexlr` FIRST on PATH (so the reference's recipe at ntJoin:204-205 runs on the GPU) and forwards every
variable; `time=True` wraps the recipes in `$(log_time)` exactly as the reference does (ntJoin:98-107)."""
import os
import stat
import subprocess

from tests.conftest import REPO

MK = os.path.join(REPO, "ntJoin-mx")


def _stub(path, body):
    with open(path, "w") as fh:
        fh.write("#!/bin/bash\n" + body)
    os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR)


def test_assemble_delegates_with_our_indexlr_first_on_path(tmp_path):
    stub = tmp_path / "ntJoin"
    _stub(stub, 'echo "indexlr=$(command -v indexlr)" > "$PWD/seen.txt"\nfor a in "$@"; do echo "arg=$a" >> "$PWD/seen.txt"; done\n')
    subprocess.check_call(["make", "-f", MK, "assemble", "target=scaf.fa", "references=r1.fa r2.fa", "reference_weights=2 3",
                           "target_weight=1", "k=24", "w=250", "n=2", "t=7", "prefix=pre", "time=True", f"ntjoin={stub}"],
                          cwd=tmp_path)
    seen = (tmp_path / "seen.txt").read_text().splitlines()
    assert seen[0] == "indexlr=" + os.path.join(REPO, "ntjoin_amd", "bin", "indexlr")
    args = [l[4:] for l in seen[1:]]
    assert args[0] == "assemble"
    for want in ("target=scaf.fa", "references=r1.fa r2.fa", "reference_weights=2 3", "target_weight=1", "k=24", "w=250", "n=2",
                 "t=7", "prefix=pre", "time=True", "reference_config=None"):
        assert want in args, (want, args)


def test_assemble_forwards_the_downstream_variables_by_name(tmp_path):
    """the variables of the stages behind the graph (reference ntJoin:36-77: g, G, m, mkt, agp, no_cut, overlap, overlap_k,
    overlap_w, overlap_g, assemble_t) reach the reference's make explicitly when given, and are left to its defaults when not"""
    stub = tmp_path / "ntJoin"
    _stub(stub, 'for a in "$@"; do echo "arg=$a" >> "$PWD/seen.txt"; done\n')
    given = ["g=30", "G=500", "m=80", "mkt=True", "agp=True", "no_cut=True", "overlap=False", "overlap_k=17", "overlap_w=12",
             "overlap_g=25", "assemble_t=3"]
    subprocess.check_call(["make", "-f", MK, "assemble", "target=scaf.fa", "references=r1.fa", "reference_weights=2", f"ntjoin={stub}"] + given,
                          cwd=tmp_path, env={k: v for k, v in os.environ.items() if k not in ("MAKEFLAGS", "MFLAGS")})
    args = [l[4:] for l in (tmp_path / "seen.txt").read_text().splitlines()]
    for want in given:
        assert want in args, (want, args)
    os.remove(tmp_path / "seen.txt")
    subprocess.check_call(["make", "-f", MK, "assemble", "target=scaf.fa", "references=r1.fa", "reference_weights=2", f"ntjoin={stub}"],
                          cwd=tmp_path)
    args = [l[4:] for l in (tmp_path / "seen.txt").read_text().splitlines()]
    assert not any(a.split("=")[0] in ("g", "G", "m", "mkt", "agp", "no_cut", "overlap", "overlap_k", "overlap_w", "overlap_g", "assemble_t")
                   for a in args), args


def test_assemble_without_reference_path_is_an_error(tmp_path):
    r = subprocess.run(["make", "-f", MK, "assemble", "target=a.fa", "references=b.fa", "reference_weights=2"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "ntjoin=" in r.stderr


def test_log_time_wraps_the_sketch_recipe(tmp_path):
    """time=True: `command time -v -o <target>.time <recipe>` (GNU time is looked up on PATH like in the reference; here a
    stand-in that records what it was asked to run, and a stand-in indexlr through mx_engine=indexlr)"""
    bind = tmp_path / "bin"
    bind.mkdir()
    _stub(bind / "time", 'out=""; while [ "$1" = "-v" ] || [ "$1" = "-o" ]; do if [ "$1" = "-o" ]; then out="$2"; shift; fi; shift; done\n'
                         'echo "ran: $*" > "$out"; exec "$@"\n')
    _stub(bind / "indexlr", 'echo "fake sketch of ${@: -1}"\n')
    (tmp_path / "a.fa").write_text(">x\nACGT\n")
    env = dict(os.environ, PATH=f"{bind}:{os.environ['PATH']}")
    subprocess.check_call(["make", "-f", MK, "a.fa.k32.w100.tsv", "k=32", "w=100", "t=3", "time=True", "mx_engine=indexlr"], cwd=tmp_path, env=env)
    assert (tmp_path / "a.fa.k32.w100.tsv").read_text() == "fake sketch of a.fa\n"
    assert (tmp_path / "a.fa.k32.w100.tsv.time").read_text().strip() == "ran: indexlr --seq --long --pos -k 32 -w 100 -t 3 a.fa"
    # time=False (the default): no wrapper, no .time file
    os.remove(tmp_path / "a.fa.k32.w100.tsv")
    os.remove(tmp_path / "a.fa.k32.w100.tsv.time")
    subprocess.check_call(["make", "-f", MK, "a.fa.k32.w100.tsv", "k=32", "w=100", "mx_engine=indexlr"], cwd=tmp_path, env=env)
    assert not (tmp_path / "a.fa.k32.w100.tsv.time").exists()


def test_readme_lists_every_environment_knob_the_library_reads():
    """every MXG_* name the sources under ntjoin_amd/csrc read from the environment has a line in README.md's knob table, and the
    table names nothing that no source (library, Python package, bench.py, tests, tools) reads any more"""
    import re
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(here, "ntjoin_amd", "csrc")
    read = set()
    for f in os.listdir(csrc):
        if f.endswith((".cpp", ".hip", ".h")):
            read |= set(re.findall(r'"(MXG_[A-Z0-9_]+)"', open(os.path.join(csrc, f), encoding="utf-8").read()))
    readme = open(os.path.join(here, "README.md"), encoding="utf-8").read()
    named = set(re.findall(r"MXG_[A-Z0-9_]+", readme))
    assert not (read - named), f"knobs the library reads that README.md does not name: {sorted(read - named)}"
    elsewhere = set()
    for d, _, files in os.walk(here):
        if any(part in d for part in (".git", "gpurun_out", "profiles", "__pycache__")):
            continue
        for f in files:
            if f.endswith((".py", ".sh", ".h", "ntJoin-mx")) or f == "ntJoin-mx":
                try:
                    elsewhere |= set(re.findall(r"MXG_[A-Z0-9_]+", open(os.path.join(d, f), encoding="utf-8").read()))
                except (OSError, UnicodeDecodeError):
                    pass
    stale = {n for n in named - read - elsewhere if n != "MXG_X"}   # (MXG_X: the placeholder of tools/abenv.sh's usage line)
    assert not stale, f"README.md names knobs nothing reads: {sorted(stale)}"
