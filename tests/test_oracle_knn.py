"""CPU: pins the kNN oracle against the reference's own known-answer tests (G1),
a literal numpy restatement of the reference lines, and the committed goldens."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import knn_oracle as ko
from claude_semantic_search_amd import synth

GOLD = Path(__file__).resolve().parent / "golden"


def _cases():
    return json.loads((GOLD / "knn_reference_cases.json").read_text())


@pytest.mark.parametrize("case", _cases()["cases"], ids=lambda c: c["name"])
def test_oracle_matches_reference_known_answers(case):
    x = ko.normalize_rows(np.array(case["rows"], dtype=np.float32))      # src/storage.py:347-350
    q = ko.normalize_rows(np.array(case["query"], dtype=np.float32))     # src/storage.py:426
    ix = ko.FlatIndexOracle(4, ko.METRIC_IP)
    ix.add(x)
    k = min(100, ix.ntotal)                                              # src/storage.py:432
    D, I = ix.search(q, k)
    assert I[0].tolist() == case["expected_ids"]
    assert np.allclose(D[0], case["expected_sims"], atol=1e-6)
    a = case["asserts"]  # what the reference's own test asserts
    if "top1_id" in a:
        assert I[0][0] == a["top1_id"]
    if "top1_sim_gt" in a:
        assert D[0][0] > a["top1_sim_gt"]
    if a.get("strictly_descending"):
        assert all(D[0][i] > D[0][i + 1] for i in range(k - 1))
    if "all_sims_gt" in a:
        assert (D[0] > a["all_sims_gt"]).all()


def test_env_case_self_query_returns_itself():
    env = _cases()["env_case"]  # tests/test_environment_setup.py:199-220
    v = np.random.default_rng(env["seed"]).random((env["n"], env["d"])).astype(np.float32)
    ix = ko.FlatIndexOracle(env["d"], ko.METRIC_IP)
    assert ix.d == env["d"] and ix.ntotal == 0
    ix.add(v)
    assert ix.ntotal == 10
    D, I = ix.search(v[0:1], env["k"])
    assert D.shape == (1, 5) and I.shape == (1, 5)
    assert I[0][0] == 0


def test_normalize_matches_numpy_reference_lines():
    x = synth.rows(257, 768, 77)
    a = ko.normalize_rows(x)
    b = ko.normalize_rows_numpy(x)
    assert np.allclose(a, b, atol=2e-7)
    z = np.zeros((2, 8), np.float32)  # all-zero row: 0 / (0 + 1e-8) = 0 (tests/test_incremental_indexing.py:129)
    assert np.array_equal(ko.normalize_rows(z), z)


@pytest.mark.parametrize("metric", [ko.METRIC_IP, ko.METRIC_L2])
def test_oracle_vs_numpy_full_sort(metric):
    x = ko.normalize_rows(synth.rows(3000, 96, 5))
    q = ko.normalize_rows(synth.rows(7, 96, 6))
    ix = ko.FlatIndexOracle(96, metric)
    ix.add(x)
    D, I = ix.search(q, 20)
    x64, q64 = x.astype(np.float64), q.astype(np.float64)
    if metric == ko.METRIC_IP:
        s = q64 @ x64.T
        order = np.argsort(-s, axis=1, kind="stable")[:, :20]
    else:
        s = ((q64[:, None, :] - x64[None, :, :]) ** 2).sum(-1)
        order = np.argsort(s, axis=1, kind="stable")[:, :20]
    ref = np.take_along_axis(s, order, axis=1)
    assert np.allclose(D, ref, atol=2e-6)
    # ids equal wherever the fp64 gap to the neighbours is not a near tie
    gaps = np.abs(np.diff(ref, axis=1))
    safe = np.ones_like(order, bool)
    safe[:, 1:] &= gaps > 1e-6
    safe[:, :-1] &= gaps > 1e-6
    assert (I[safe] == order[safe]).all()
    Db, Ib = ko.search_blas(x, q, 20, metric)
    assert np.allclose(Db, D, atol=5e-6)
    assert (Ib[safe] == I[safe]).all()


def test_padding_and_tie_order():
    ix = ko.FlatIndexOracle(4, ko.METRIC_IP)
    ix.add(np.array([[1, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0]], np.float32))
    D, I = ix.search(np.array([[1, 0, 0, 0]], np.float32), 5)
    assert I[0].tolist() == [0, 1, 2, -1, -1]          # ties -> lower id; -1 padded
    assert D[0][3] == -np.finfo(np.float32).max
    l2 = ko.FlatIndexOracle(4, ko.METRIC_L2)
    l2.add(np.array([[1, 0, 0, 0], [0, 1, 0, 0]], np.float32))
    D, I = l2.search(np.array([[1, 0, 0, 0]], np.float32), 3)
    assert I[0].tolist() == [0, 1, -1] and D[0][0] == 0.0 and D[0][1] == 2.0
    assert D[0][2] == np.finfo(np.float32).max


def test_sharded_merge_equals_whole():
    x = ko.normalize_rows(synth.rows(5000, 64, 11))
    q = ko.normalize_rows(synth.rows(9, 64, 12))
    whole = ko.FlatIndexOracle(64)
    whole.add(x)
    D, I = whole.search(q, 10)
    parts_d, parts_i = [], []
    for lo, hi in [(0, 1250), (1250, 2500), (2500, 3750), (3750, 5000)]:
        s = ko.FlatIndexOracle(64)
        s.add(x[lo:hi])
        d, i = s.search(q, 10)
        parts_d.append(d)
        parts_i.append(i + lo)
    Dm, Im = ko.merge_topk(np.stack(parts_d), np.stack(parts_i))
    assert np.array_equal(Im, I) and np.array_equal(Dm, D)


def test_synth_generator_numpy_equals_c():
    assert np.array_equal(synth.rows(33, 768, 1234, 7), ko.synth_rows(33, 768, 1234, 7))
    r = synth.rows(2000, 768, 99)
    assert abs(float(r.mean())) < 5e-3 and abs(float(r.std()) - 1.0) < 5e-3


def test_committed_synthetic_goldens_reproduce():
    for name, metric, norm in (("knn_synth_ip.npz", ko.METRIC_IP, True), ("knn_synth_l2.npz", ko.METRIC_L2, False)):
        g = np.load(GOLD / name)
        x = synth.rows(int(g["n"]), int(g["d"]), int(g["seed_x"]))
        q = synth.rows(int(g["nq"]), int(g["d"]), int(g["seed_q"]))
        if norm:
            x, q = ko.normalize_rows(x), ko.normalize_rows(q)
        ix = ko.FlatIndexOracle(int(g["d"]), metric)
        ix.add(x)
        D, I = ix.search(q, int(g["k"]))
        assert np.array_equal(I, g["I"].astype(np.int64))
        assert np.allclose(D, g["D"], atol=1e-6)
        assert np.allclose(D, g["D64"], atol=5e-6 if norm else 5e-4)
