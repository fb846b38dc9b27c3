"""Generates tests/golden/knn_*.{json,npz}.  Run from the repo root:

    python tests/golden/make_knn_goldens.py

G1 (knn_reference_cases.json): the fixtures of the reference's own known-answer
tests for the flat index, with expected values computed by a float64 numpy
evaluation of the reference's own lines (src/storage.py:347-350 normalise,
:425-426 query normalise, exact inner product, descending order).  What the
reference's tests assert about them is recorded next to each case
("asserts"), so tests/test_oracle_knn.py can check both.

G2/G3 (knn_synth_ip.npz / knn_synth_l2.npz): N=10000 x 768 rows + 16 queries
from include/css_synth.h (seeds stored), top-100 ids/scores (IP, rows and
queries normalised as add_chunks/search do) and top-10 (squared L2, raw rows)
from the CPU oracle, plus fp64 re-scores.  Inputs are regenerated from the
seeds at test time; only outputs are stored.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import knn_oracle as ko  # noqa: E402

OUT = Path(__file__).resolve().parent


def ref_case(name, source, rows, query, asserts):
    x = np.array(rows, dtype=np.float32)
    xn = (x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-8)).astype(np.float32)  # storage.py:349-350
    q = np.array(query, dtype=np.float32)
    qn = (q / (np.linalg.norm(q) + 1e-8)).astype(np.float32)  # storage.py:426
    s = xn.astype(np.float64) @ qn.astype(np.float64)
    order = np.argsort(-s, kind="stable")
    return {
        "name": name, "source": source, "rows": rows, "query": query,
        "expected_ids": order.tolist(), "expected_sims": s[order].tolist(), "asserts": asserts,
    }


def main():
    cases = [
        ref_case("test_storage_fixture", "tests/test_storage.py:106-156,277-291",
                 [[0.1, 0.2, 0.3, 0.4], [0.5, 0.6, 0.7, 0.8], [0.9, 0.1, 0.2, 0.3]], [0.1, 0.2, 0.3, 0.4],
                 {"top1_id": 0, "top1_sim_gt": 0.8}),
        ref_case("test_storage_and_search_workflow", "tests/test_integration.py:144-204",
                 [[0.1, 0.2, 0.3, 0.4], [0.2, 0.3, 0.4, 0.5], [0.3, 0.4, 0.5, 0.6]], [0.15, 0.25, 0.35, 0.45],
                 {"all_sims_gt": 0.0, "n_results": 3}),
        ref_case("test_search_relevance", "tests/test_integration.py:312-353",
                 [[1.0, 0.9, 0.8, 0.7], [0.8, 0.7, 0.6, 0.5], [0.2, 0.3, 0.4, 0.5]], [1.0, 0.9, 0.8, 0.7],
                 {"top1_id": 0, "strictly_descending": True, "top1_sim_gt": 0.9, "n_results": 3}),
    ]
    # values quoted in SURVEY.md 8(c), computed independently by the surveyor
    survey = {
        "test_storage_fixture": ([0, 1, 2], [0.9999999, 0.9688640, 0.5432198]),
        "test_storage_and_search_workflow": ([1, 0, 2], [0.9988701, 0.9979654, 0.9935983]),
        "test_search_relevance": ([0, 1, 2], [1.0, 0.9992177, 0.9047619]),
    }
    for c in cases:
        ids, sims = survey[c["name"]]
        assert c["expected_ids"] == ids, (c["name"], c["expected_ids"])
        assert np.allclose(c["expected_sims"], sims, atol=2e-7), (c["name"], c["expected_sims"])
    # tests/test_environment_setup.py:199-220: d=128, 10 rows in [0,1), self query -> id 0, shapes (1,5)
    rng = np.random.default_rng(20250725)
    v = rng.random((10, 128)).astype(np.float32)
    # the reference uses unseeded data and asserts I[0][0] == 0; pick a seed where that holds for exact IP
    s = v @ v[0]
    assert int(np.argmax(s)) == 0
    env = {"name": "test_faiss_functionality", "source": "tests/test_environment_setup.py:199-220",
           "seed": 20250725, "d": 128, "n": 10, "k": 5, "asserts": {"top1_id": 0, "shape": [1, 5]}}
    (OUT / "knn_reference_cases.json").write_text(json.dumps({"cases": cases, "env_case": env}, indent=1))

    n, d, nq = 10000, 768, 16
    seed_x, seed_q = 1234, 4321
    x = ko.synth_rows(n, d, seed_x)
    q = ko.synth_rows(nq, d, seed_q)
    ip = ko.FlatIndexOracle(d, ko.METRIC_IP)
    ip.add(ko.normalize_rows(x))
    qn = ko.normalize_rows(q)
    D, I = ip.search(qn, 100)
    D64 = ip.rescore64(qn, I)
    np.savez_compressed(OUT / "knn_synth_ip.npz", n=n, d=d, nq=nq, seed_x=seed_x, seed_q=seed_q, k=100,
                        I=I.astype(np.int32), D=D, D64=D64)
    l2 = ko.FlatIndexOracle(d, ko.METRIC_L2)
    l2.add(x)
    D2, I2 = l2.search(q, 10)
    D264 = l2.rescore64(q, I2)
    np.savez_compressed(OUT / "knn_synth_l2.npz", n=n, d=d, nq=nq, seed_x=seed_x, seed_q=seed_q, k=10,
                        I=I2.astype(np.int32), D=D2, D64=D264)
    print("wrote goldens:", [p.name for p in OUT.glob("knn_*")])


if __name__ == "__main__":
    main()
