"""CPU tests of the array-backed B2 containers (ntjoin_amd/ntjoin_utils.py: MxInfo, MxLists, MxGraph.from_arrays): they must
behave like the dict / lists / graph of the exact mode they replace at genome scale (reference bin/ntjoin_utils.py:187-193)."""
import pickle

import numpy as np

from ntjoin_amd.ntjoin_utils import MxGraph, MxInfo, MxLists, sketch_views


def _sketch(rng, n_rec=40):
    first = np.concatenate(([0], np.cumsum(rng.integers(0, 9, size=n_rec)))).astype(np.uint64)
    n = int(first[-1])
    sk = {"out_hash": rng.integers(0, 60, size=n).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15),
          "pos": rng.integers(0, 10_000, size=n).astype(np.uint32), "record": np.repeat(np.arange(n_rec), np.diff(first.astype(np.int64))).astype(np.uint32),
          "record_first": first, "record_ids": [f"ctg{r}" for r in range(n_rec)]}
    v, c = np.unique(sk["out_hash"], return_counts=True)
    return sk, np.isin(sk["out_hash"], v[c == 1])


def test_views_equal_the_exact_containers():
    rng = np.random.default_rng(4)
    sk, uniq = _sketch(rng)
    info, lists = sketch_views(sk, uniq)
    # the exact mode, as read_minimizers builds it
    ids = sk["record_ids"]
    want_info = {str(h): (ids[r], int(p)) for h, p, r in zip(sk["out_hash"][uniq].tolist(), sk["pos"][uniq].tolist(), sk["record"][uniq].tolist())}
    want_lists = []
    for r in range(len(ids)):
        lo, hi = int(sk["record_first"][r]), int(sk["record_first"][r + 1])
        if hi > lo:
            want_lists.append([str(x) for x in sk["out_hash"][lo:hi][uniq[lo:hi]].tolist()])
    assert isinstance(info, MxInfo) and isinstance(lists, MxLists)
    assert dict(info.items()) == want_info and info.to_dict() == want_info and len(info) == len(want_info)
    assert lists.to_lists() == want_lists and len(lists) == len(want_lists) and lists[-1] == want_lists[-1]
    some = next(iter(want_info))
    assert some in info and int(some) in info and info[int(some)] == want_info[some]
    assert "12345" not in info and "not a number" not in info
    assert pickle.loads(pickle.dumps(info)).to_dict() == want_info
    assert pickle.loads(pickle.dumps(lists)).to_lists() == want_lists


def test_lazy_graph_equals_the_exact_graph():
    vh = np.array([11, 5, 8, 2], dtype=np.uint64)
    eu, ev = np.array([0, 1, 2], dtype=np.uint32), np.array([1, 2, 3], dtype=np.uint32)
    sup, wt = np.array([3, 1, 2], dtype=np.uint32), np.array([3.0, 2.0, 1.0])
    lazy = MxGraph.from_arrays(vh, eu, ev, sup, wt, ["r.tsv", "t.tsv"])
    exact = MxGraph([str(h) for h in vh.tolist()], zip(eu.tolist(), ev.tolist()),
                    [["r.tsv", "t.tsv"], ["r.tsv"], ["t.tsv"]], wt.tolist())
    assert lazy.vcount() == exact.vcount() == 4 and lazy.ecount() == exact.ecount() == 3
    assert "names" not in lazy.__dict__          # nothing was turned into strings yet
    assert lazy.edge_list_named() == exact.edge_list_named()
    assert lazy.get_eid("5", "11") == exact.get_eid("5", "11") == 0 and lazy.degree() == exact.degree()
    assert lazy.vertex_index("8") == 2
    again = pickle.loads(pickle.dumps(lazy))
    assert again.edge_list_named() == exact.edge_list_named()
