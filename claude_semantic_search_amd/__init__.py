"""MI355X-native embed-and-search core of claude-semantic-search.

Public surface = the reference's ``src/__init__.py:10-31`` names for the hot
path (``EmbeddingGenerator``, ``HybridStorage`` and their config/result
dataclasses, ``Chunk``), implemented over ``libcss_hip.so`` (hand-written HIP
for gfx950, C ABI in ``include/css_hip.h``).  Importing the package touches
neither HIP nor torch; the native library is loaded on first use.
"""
from __future__ import annotations

import importlib

__version__ = "0.1.0"

_LAZY = {
    "Chunk": ".chunk",
    "StorageConfig": ".storage",
    "SearchConfig": ".storage",
    "SearchResult": ".storage",
    "HybridStorage": ".storage",
    "EmbeddingConfig": ".embeddings",
    "EmbeddingStats": ".embeddings",
    "EmbeddingGenerator": ".embeddings",
    "EmbeddingBatcher": ".embeddings",
    "IndexFlat": ".flat_index",
    "IndexFlatIP": ".flat_index",
    "IndexFlatL2": ".flat_index",
    "MpnetEncoder": ".mpnet_encoder",
    "ShardedFlatIndex": ".sharded",
    "GPUCapability": ".gpu_utils",
    "assess_gpu_capability": ".gpu_utils",
    "calculate_optimal_batch_size": ".gpu_utils",
}

__all__ = sorted(_LAZY)


def __getattr__(name: str):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
    return getattr(importlib.import_module(mod, __name__), name)
