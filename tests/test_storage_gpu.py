"""GPU: the same HybridStorage cases through libcss_hip.so (no test double)."""
import pytest

from storage_cases import StorageCases

pytestmark = pytest.mark.gpu


class TestStorageOnHip(StorageCases):
    def test_native_library_is_what_runs(self):
        from claude_semantic_search_amd import flat_index as fi

        self.storage.initialize()
        assert type(self.storage.faiss_index) is fi.IndexFlatIP
        assert self.storage.faiss_index._h is not None
