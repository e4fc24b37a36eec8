"""bench.py without a GPU: the launcher of `python bench.py --gpus N` (ranks started, failures reported, no JSON line from a failed
run) and the helpers that read the committed profiles."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_own_launcher_reports_a_failed_rank():
    """no GPU here: every rank ends with "bench.py needs a GPU"; the launcher must come back with a non-zero status, say which
    rank failed, and print no JSON line"""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=REPO, env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():   # (on a GPU box this test has nothing to say: tests/test_gpu_dist2.py runs the launcher for real)
        return
    assert out.returncode != 0
    assert "ended with status" in out.stderr and "needs a GPU" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_rank_count_must_match():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "1"], cwd=REPO,
                         env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "must agree" in (out.stderr + out.stdout)


def test_committed_profiles_are_readable():
    """what bench.py quotes from profiles/<round>/: the kernel stats name the step's kernels (largest first) and the slice kernel's
    PMC summary has the fields the valu_by_kernel entry is made of"""
    sys.path.insert(0, REPO)
    import bench
    rows, path = bench.committed_kernel_stats("configs2")
    if rows is None:   # (no profile of this round committed yet)
        assert not os.path.exists(path)
        return
    names = [r[0] for r in rows]
    assert any(n.startswith("k_hash_bs") for n in names) and any(n.startswith("k_bs_select") for n in names)
    assert rows == sorted(rows, key=lambda r: -r[1]) and not any(n.startswith("k_synth") for n in names)
    sp = os.path.join(REPO, "profiles", bench.PROFILE_ROUND, "configs2_select_pmc.json")
    if os.path.exists(sp):
        pj = json.load(open(sp))
        assert pj["per_slice"]["valu"] > 0 and pj["per_slice"]["salu"] > 0 and pj["slices_per_launch"] > 0
        m = bench.select_valu_model(0.5, pj["slices_per_launch"])
        assert m is not None and ("stale" in m or 0 < m["frac_of_issue_bound"] < 1.5)


def test_scaling_fields_quote_the_same_workload_on_one_gpu():
    """the N-GPU line's efficiency = value / (N x the committed one-GPU line of the SAME workload and size); another size: no figure"""
    sys.path.insert(0, REPO)
    import bench
    f = bench.scaling_fields("configs2", 1000, 2, 3000.0, 5993199554.0)
    one = f["one_gpu_same_workload"]
    assert one and one["from"].startswith("profiles/r0") and one["from"].endswith("bench_configs2.json") and one["value"] > 1000
    assert abs(f["efficiency"] - 3000.0 / (2 * one["value"])) < 1e-3
    g = bench.scaling_fields("configs2", 1000, 2, 30.0, 1.0e7)   # (a test-sized run: no committed line of that size)
    assert g["efficiency"] is None and g["one_gpu_same_workload"] is None and "python bench.py --workload configs2" in g["efficiency_is"]


def test_own_launcher_stops_a_run_that_outlives_its_budget(tmp_path):
    """the launcher's wall-clock budget, without a GPU: a stand-in rank program that sleeps; rc 124, the ranks named, their stderr shown"""
    sys.path.insert(0, REPO)
    import bench
    prog = tmp_path / "sleeper.py"
    prog.write_text("import os, sys, time\nprint('rank', os.environ['RANK'], 'sleeping', file=sys.stderr, flush=True)\ntime.sleep(600)\n")
    code = ("import sys, os\nsys.path.insert(0, %r)\nimport bench\nbench.__file__ = %r\nsys.argv = ['bench.py']\n"
            "os.environ['MXG_BENCH_BUDGET_S'] = '3'\nsys.exit(bench.launch_ranks(2))\n") % (REPO, str(prog))
    out = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 124, out.stderr[-2000:]
    assert "still running after 3 s" in out.stderr and "[rank 1] rank 1 sleeping" in out.stderr and "[rank 0] rank 0 sleeping" in out.stderr


def test_exchange_beside_sketches_timeline():
    """the dry run's model of an exchange per assembly on one communication stream: an exchange starts when its assembly is sketched
    and the exchange in front of it has ended"""
    import bench
    f = bench.exchange_beside_sketches
    assert f([1.0, 1.0], [0.5, 0.5]) == 2.5                 # each hides under the next sketch; the last one is exposed
    assert f([1.0, 1.0], [3.0, 0.5]) == 4.5                 # the first is still travelling when the second is ready
    assert f([0.5, 0.5, 0.5, 0.5], [0.2] * 4) == 2.2        # four assemblies: only the last exchange shows
    assert f([1.0], [0.7]) == 1.7 and f([], []) == 0.0
    # never better than everything behind the sketches, never worse than that either
    import itertools
    for sk, se in itertools.product([[0.3, 0.9, 0.1], [1.0, 1.0, 1.0]], [[0.5, 0.1, 0.8], [2.0, 2.0, 2.0]]):
        assert max(sum(sk) + se[-1], sk[0] + sum(se)) - 1e-12 <= f(sk, se) <= sum(sk) + sum(se) + 1e-12
