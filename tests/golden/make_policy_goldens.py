"""Generates tests/golden/batch_size_table.json and tests/golden/filter_truth_table.json (G8).

G7 restates the reference's formula by hand (src/gpu_utils.py:169-192: clamp(int((free_GB - 1) / (dim * 16 / 2^30)),
8, 256), 64 for "mps", 8 when free_GB <= 1) in exact rational arithmetic -- it does not import the reference or
the build.  G8 holds the inputs and expected truth values of the reference's own test (tests/test_storage.py:617-647)
plus rows that follow from the reference's rules line by line (src/storage.py:508-540): unknown keys are ignored,
range bounds gte / lte / gt / lt, list = membership, project_name = case-insensitive substring, anything else ==.
Run: python tests/golden/make_policy_goldens.py"""
import json
from fractions import Fraction
from pathlib import Path

HERE = Path(__file__).resolve().parent


def batch(free_gb: str, dim: int, backend: str) -> int:
    working = Fraction(free_gb) - 1
    if working <= 0:
        return 8
    per_item = Fraction(dim * 16, 1024**3)
    n = int(working / per_item)
    return max(8, min(n, 64 if backend == "mps" else 256))


rows = []
for free in ("0", "0.5", "1", "1.00005", "1.0005", "1.001", "1.002", "1.0029", "1.003", "2", "8", "24", "192", "288"):
    for dim in (768, 384):
        for backend in ("cuda", "mps"):
            rows.append({"free_gb": float(free), "dim": dim, "backend": backend, "batch": batch(free, dim, backend)})
(HERE / "batch_size_table.json").write_text(json.dumps({"source": "src/gpu_utils.py:169-192", "rows": rows}, indent=1) + "\n")

chunk = {"project_name": "test_project", "word_count": 10, "has_code": True, "chunk_type": "qa_pair"}
truth = [
    # the reference's own assertions (tests/test_storage.py:628-647)
    [{"project_name": "test_project"}, True], [{"project_name": "other_project"}, False],
    [{"word_count": {"gte": 5}}, True], [{"word_count": {"lte": 15}}, True], [{"word_count": {"gt": 10}}, False],
    [{"chunk_type": ["qa_pair", "code_block"]}, True], [{"chunk_type": ["tool_usage"]}, False],
    # rows that follow from src/storage.py:508-540
    [{}, True], [{"no_such_key": "x"}, True], [{"related_to": "chunk_9", "same_session": True}, True],
    [{"word_count": {"gte": 10}}, True], [{"word_count": {"gte": 11}}, False], [{"word_count": {"lte": 10}}, True],
    [{"word_count": {"lte": 9}}, False], [{"word_count": {"gt": 9}}, True], [{"word_count": {"lt": 10}}, False],
    [{"word_count": {"lt": 11}}, True], [{"word_count": {"gte": 5, "lt": 10}}, False], [{"word_count": {"gt": 5, "lte": 10}}, True],
    [{"word_count": {}}, True], [{"word_count": 10}, True], [{"word_count": 11}, False], [{"word_count": [9, 10]}, True],
    [{"project_name": "TEST_PRO"}, True], [{"project_name": "ject"}, True], [{"project_name": ""}, True],
    [{"project_name": "test project"}, False], [{"project_name": ["test_project"]}, True], [{"project_name": ["TEST_PROJECT"]}, False],
    [{"has_code": True}, True], [{"has_code": False}, False], [{"has_code": 1}, True],
    [{"chunk_type": "qa_pair"}, True], [{"chunk_type": "QA_PAIR"}, False], [{"chunk_type": "qa"}, False],
    [{"project_name": "test", "chunk_type": ["qa_pair"], "word_count": {"gte": 10}}, True],
    [{"project_name": "test", "chunk_type": ["qa_pair"], "word_count": {"gt": 10}}, False],
]
(HERE / "filter_truth_table.json").write_text(json.dumps(
    {"source": "tests/test_storage.py:617-647, src/storage.py:508-540", "chunk_data": chunk, "cases": truth}, indent=1) + "\n")
print(len(rows), "batch rows;", len(truth), "filter cases")
