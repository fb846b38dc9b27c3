"""CPU: HybridStorage host logic (SQLite, id maps, filters, persistence format)
with an oracle-backed TEST DOUBLE standing in for the device index.  The double
lives here in tests/ only; the product never falls back to it."""
import numpy as np
import pytest

from oracle import knn_oracle as ko
from claude_semantic_search_amd import flat_index as fi
from storage_cases import StorageCases


class _FakeIndex:
    def __init__(self, d, metric=0, device=0):
        self.d, self.metric_type, self.device = int(d), int(metric), device
        self._o = ko.FlatIndexOracle(d, metric)

    ntotal = property(lambda self: self._o.ntotal)

    def add(self, x, normalize=False):
        x = np.asarray(x, np.float32).reshape(-1, self.d)
        self._o.add(ko.normalize_rows(x) if normalize else x)

    def search(self, q, k, normalize=False, allow=None):
        q = np.asarray(q, np.float32).reshape(-1, self.d)
        q = ko.normalize_rows(q) if normalize else q
        if allow is None:
            return self._o.search(q, k)
        # masked search = the oracle over the allowed rows only, ids mapped back
        sub = np.flatnonzero(np.asarray(allow, dtype=bool))
        o = ko.FlatIndexOracle(self.d, self.metric_type)
        if sub.size:
            o.add(self._o._xb[sub])
        D, I = o.search(q, k)
        return D, np.where(I >= 0, sub[np.clip(I, 0, max(sub.size - 1, 0))] if sub.size else -1, -1)

    def reconstruct_n(self, row0=0, n=None):
        n = self.ntotal - row0 if n is None else n
        return self._o._xb[row0:row0 + n].copy()

    def reserve(self, n):
        pass

    def reset(self):
        self._o.reset()

    def close(self):
        pass


@pytest.fixture(autouse=True)
def fake_device_index(monkeypatch):
    monkeypatch.setattr(fi, "IndexFlat", _FakeIndex)
    monkeypatch.setattr(fi, "IndexFlatIP", lambda d, device=0: _FakeIndex(d, 0, device))
    monkeypatch.setattr(fi, "IndexFlatL2", lambda d, device=0: _FakeIndex(d, 1, device))


class TestStorageHostLogic(StorageCases):
    pass


def test_index_file_layout_roundtrip(tmp_path):
    ix = _FakeIndex(6, 1)
    ix.add(np.arange(30, dtype=np.float32).reshape(5, 6))
    p = tmp_path / "x.faiss"
    fi.write_index(ix, str(p))
    raw = p.read_bytes()
    assert raw[:4] == b"IxF2" and len(raw) == 4 + 4 + 8 + 16 + 1 + 4 + 8 + 30 * 4
    back = fi.read_index(str(p))
    assert back.d == 6 and back.metric_type == 1 and back.ntotal == 5
    assert np.array_equal(back.reconstruct_n(0, 5), ix.reconstruct_n(0, 5))
    p.write_bytes(raw[:50])
    with pytest.raises(RuntimeError):
        fi.read_index(str(p))
