"""Probe against a REAL btllib `indexlr` when one is on PATH (SURVEY.md 7 "hard parts"): the only thing that can pin what no file
under the reference tree pins -- hash VALUES of the current canonical variant (fwd+rev), windows spanning N, lower case / U /
IUPAC handling, records shorter than k+w-1, the case of the --seq column.  Skipped (not failed) where no such binary exists --
as in this image and on the GPU boxes; DESIGN.md keeps those points listed as "parity unpinned" until this test has run."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
OURS = os.path.join(REPO, "ntjoin_amd", "bin", "indexlr")


def _real_indexlr():
    ours_dir = os.path.realpath(os.path.dirname(OURS))
    for d in os.environ.get("PATH", "").split(os.pathsep):
        cand = os.path.join(d, "indexlr")
        if d and os.path.isfile(cand) and os.access(cand, os.X_OK) and os.path.realpath(d) != ours_dir:
            try:  # ours prints "GPU (MI355X) minimizer sketcher" in its usage text
                txt = subprocess.run([cand, "--help"], capture_output=True, text=True, timeout=30)
                if "MI355X" not in (txt.stdout + txt.stderr):
                    return cand
            except (OSError, subprocess.SubprocessError):
                continue
    return os.environ.get("MXG_REAL_INDEXLR") or None


REAL = _real_indexlr()


@pytest.mark.skipif(REAL is None, reason="no btllib indexlr on PATH (set MXG_REAL_INDEXLR=/path/to/indexlr to point at one)")
@pytest.mark.parametrize("k,w", [(32, 100), (32, 1000), (15, 10)])
def test_tsv_equals_real_indexlr(tmp_path, k, w):
    fastas = sorted(f for f in os.listdir(os.path.join(GOLDEN, "fasta")) if f.endswith(".fa"))
    # plus the shapes nothing in the reference tree pins: lower case, U, IUPAC, N inside windows, short records
    extra = tmp_path / "unpinned.fa"
    import random
    rng = random.Random(1)
    s = "".join(rng.choice("ACGT") for _ in range(30_000))
    extra.write_text(">lower\n" + s[:9000].lower() + "\n>rna\n" + s[9000:15000].replace("T", "U") + "\n>iupac\n" + s[15000:18000] + "RYN" +
                     s[18000:24000] + "\n>nwin\n" + s[24000:26000] + "N" * 17 + s[26000:29000] + "\n>short\n" + s[:k + w - 2] + "\n>tiny\nACGT\n")
    for fa in [os.path.join(GOLDEN, "fasta", f) for f in fastas] + [str(extra)]:
        theirs = subprocess.run([REAL, "--seq", "--long", "--pos", "-k", str(k), "-w", str(w), "-t", "2", fa], capture_output=True, check=True).stdout
        ours = subprocess.run([OURS, "--seq", "--long", "--pos", "-k", str(k), "-w", str(w), "-t", "2", fa], capture_output=True, check=True).stdout
        assert ours == theirs, os.path.basename(fa)
        theirs = subprocess.run([REAL, "--pos", "--strand", "-k", str(k), "-w", str(w), fa], capture_output=True, check=True).stdout
        ours = subprocess.run([OURS, "--pos", "--strand", "-k", str(k), "-w", str(w), fa], capture_output=True, check=True).stdout
        assert ours == theirs, ("--strand", os.path.basename(fa))
