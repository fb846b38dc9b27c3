"""Small fixtures shared by the GPU tests."""
import pytest


@pytest.fixture
def oracle_index():
    from oracle import knn_oracle as ko

    return ko.FlatIndexOracle


@pytest.fixture
def hip_index():
    from claude_semantic_search_amd.flat_index import IndexFlat

    return IndexFlat
