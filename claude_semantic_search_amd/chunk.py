"""``Chunk``: the unit the hot path consumes (mirror of ``src/chunker.py:16-23``).

Only the dataclass crosses into the embed/search path; the chunking strategies
themselves are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional


@dataclass
class Chunk:
    id: str
    text: str
    metadata: Dict[str, Any] = field(default_factory=dict)
    embedding: Optional[List[float]] = None
