"""``HybridStorage``: flat vector index in MI355X HBM + SQLite metadata.

Drop-in for the reference's ``src/storage.py`` (same public names, argument
meaning and error behaviour) with the faiss calls replaced by
``flat_index`` (libcss_hip.so).  Numeric path, reference file:line:

  * metric choice: ``normalize_embeddings`` -> inner product, else squared L2
    (``src/storage.py:252-258``); unknown ``index_type`` -> ``ValueError``
    (``:267``); "ivf"/"hnsw" are not reachable from the product and are not
    implemented here.
  * ``add_chunks``: float32 cast, ``x / (||x|| + 1e-8)``, ids = ntotal.. (``:343-365``)
    -- the normalisation is fused into the device ingest kernel.
  * ``search``: query normalise, ``k' = min(max_results, ntotal)``, top-k'
    (``:424-436``), then threshold / tombstone skip / SQLite row / filters /
    stop at ``top_k`` in rank order (``:438-492``).

Deliberate differences (all on the host side of the kernel):
  * nothing touches the GPU before ``initialize()`` (fork safety);
  * a lock serialises index/id-map mutation against searches (the reference has none);
  * ``auto_save`` appends the new rows to ``embeddings.faiss`` instead of
    rewriting the whole file after every add (the reference's O(N^2) I/O);
  * there is no CPU index: without a HIP device ``initialize()`` raises;
  * opt-in ``StorageConfig.filter_pushdown``: filters / tombstones become an allow-bitmap consumed by the
    kernel (``css_index_search_masked``) instead of an over-fetch of ``max_results`` hits.
"""
from __future__ import annotations

import json
import logging
import os
import sqlite3
import struct
import threading
from dataclasses import dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np

from . import flat_index as fi
from .chunk import Chunk
from .gpu_utils import GPUCapability, assess_gpu_capability, log_gpu_status

# bytes in front of the row data of faiss' IndexFlat file: fourcc, d, ntotal, 2 dummies, is_trained, metric, n_floats
_INDEX_HEADER_BYTES = 4 + 4 + 8 + 16 + 1 + 4 + 8


@dataclass
class StorageConfig:
    data_dir: str = "~/.claude-semantic-search/data"
    db_name: str = "metadata.db"
    index_name: str = "embeddings.faiss"
    embedding_dim: int = 768
    index_type: str = "flat"  # only "flat" is implemented
    ivf_nlist: int = 100
    hnsw_m: int = 16
    normalize_embeddings: bool = True
    auto_save: bool = True
    backup_enabled: bool = True
    use_gpu: bool = False
    gpu_memory_fraction: float = 0.8
    device: int = 0  # HIP device ordinal holding the index (extension)
    # extension: push filters and tombstones down into the kNN kernel as an allow-bitmap, so a
    # filtered search returns the true filtered top_k instead of whatever survives inside the first
    # ``max_results`` unfiltered hits.  Off by default: the reference's over-fetch semantics are kept bit for bit.
    filter_pushdown: bool = False
    # extension (the reference pins faiss to one device, src/storage.py:283): row-shard the index over the
    # ranks of the default torch.distributed process group, one process per GPU.  SPMD: every rank constructs the same
    # HybridStorage (its own data_dir: SQLite and the index file are replicated per rank) and makes the same calls
    # with the same arguments; search() is then the local masked search of every shard + ONE all-gather + merge.
    # CSS_STORAGE_SHARDED=1 switches it on for an unmodified caller (the reference's CLI under torch.distributed.run).
    sharded: bool = False


@dataclass
class SearchConfig:
    top_k: int = 10
    similarity_threshold: float = 0.0
    include_metadata: bool = True
    include_text: bool = True
    max_results: int = 100


@dataclass
class SearchResult:
    chunk_id: str
    similarity: float
    chunk: Optional[Chunk] = None
    metadata: Optional[Dict[str, Any]] = None
    text: Optional[str] = None


_CHUNK_COLUMNS = (
    "id", "text", "metadata", "faiss_id", "session_id", "project_name", "file_path", "chunk_type",
    "timestamp", "has_code", "has_tools", "message_count", "char_count", "word_count", "updated_at",
)

_SCHEMA = (
    """CREATE TABLE IF NOT EXISTS chunks (
        id TEXT PRIMARY KEY, text TEXT NOT NULL, metadata TEXT, faiss_id INTEGER,
        session_id TEXT, project_name TEXT, file_path TEXT, chunk_type TEXT, timestamp DATETIME,
        has_code BOOLEAN, has_tools BOOLEAN, message_count INTEGER, char_count INTEGER, word_count INTEGER,
        created_at DATETIME DEFAULT CURRENT_TIMESTAMP, updated_at DATETIME DEFAULT CURRENT_TIMESTAMP)""",
    """CREATE TABLE IF NOT EXISTS files (
        path TEXT PRIMARY KEY, last_modified DATETIME, last_indexed DATETIME, chunk_count INTEGER DEFAULT 0)""",
    "CREATE INDEX IF NOT EXISTS idx_chunks_session ON chunks(session_id)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_project ON chunks(project_name)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_timestamp ON chunks(timestamp)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_type ON chunks(chunk_type)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_has_code ON chunks(has_code)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_has_tools ON chunks(has_tools)",
    "CREATE INDEX IF NOT EXISTS idx_chunks_faiss_id ON chunks(faiss_id)",
    # build-owned: journal of the two-file operations (index file + SQLite) that must survive a crash between them
    "CREATE TABLE IF NOT EXISTS storage_meta (key TEXT PRIMARY KEY, value TEXT)",
)


def _fsync_dir(path: Path) -> None:
    """Make a rename inside ``path`` durable (POSIX: fsync of the directory; skipped where a directory cannot be opened)."""
    try:
        fd = os.open(str(path), os.O_RDONLY)
    except OSError:
        return
    try:
        os.fsync(fd)
    except OSError:
        pass
    finally:
        os.close(fd)


def _row_to_chunk(row) -> Chunk:
    meta = json.loads(row["metadata"]) if row["metadata"] else {}
    return Chunk(id=row["id"], text=row["text"], metadata=meta, embedding=None)


class HybridStorage:
    def __init__(self, config: Optional[StorageConfig] = None) -> None:
        self.config: StorageConfig = config or StorageConfig()
        self.logger = logging.getLogger(__name__)
        self.data_dir: Path = Path(self.config.data_dir).expanduser()
        self.data_dir.mkdir(parents=True, exist_ok=True)
        self.db_path: Path = self.data_dir / self.config.db_name
        self.index_path: Path = self.data_dir / self.config.index_name

        self.db: Optional[sqlite3.Connection] = None
        self.faiss_index: Optional[fi.IndexFlat] = None
        self.chunk_id_to_faiss_id: Dict[str, int] = {}
        self.faiss_id_to_chunk_id: Dict[int, str] = {}

        self._gpu_capability: Optional[GPUCapability] = None
        self._gpu_resources: Optional[Any] = None
        self._is_gpu_index: bool = False
        self._lock = threading.RLock()
        self._saved_rows = -1  # rows known to be in index_path (-1: unknown)
        self._allow_cache: Dict[str, Any] = {}  # filter key -> (stamp, allow mask); push-down only
        self._mutations = 0  # bumped by every change of the chunk set (part of the allow-cache stamp)

        self.total_chunks: int = 0
        self.embedding_dim: int = self.config.embedding_dim

    # ------------------------------------------------------------ lifecycle
    def initialize(self) -> None:
        self.logger.info("Initializing hybrid storage...")
        with self._lock:
            self._init_sqlite()
            self._init_faiss()
            self._load_existing_data()
        self.logger.info(f"Storage initialized with {self.total_chunks} chunks")

    def _init_sqlite(self) -> None:
        self.db = sqlite3.connect(str(self.db_path), check_same_thread=False)
        self.db.row_factory = sqlite3.Row
        self._create_tables()

    def _create_tables(self) -> None:
        if not self.db:
            raise RuntimeError("Database not initialized")
        cur = self.db.cursor()
        for stmt in _SCHEMA:
            cur.execute(stmt)
        self.db.commit()

    def _create_cpu_index(self) -> fi.IndexFlat:
        """Name kept from the reference; the index is created in HBM."""
        kind = self.config.index_type
        if kind == "flat":
            if self._sharded():
                from .sharded import ShardedIndexFacade

                metric = fi.METRIC_INNER_PRODUCT if self.config.normalize_embeddings else fi.METRIC_L2
                return ShardedIndexFacade(self.embedding_dim, metric, device=self.config.device)
            cls = fi.IndexFlatIP if self.config.normalize_embeddings else fi.IndexFlatL2
            return cls(self.embedding_dim, device=self.config.device)
        if kind in ("ivf", "hnsw"):
            raise NotImplementedError(
                f"index type {kind!r} is not reachable from the product and is not implemented on MI355X; use 'flat'"
            )
        raise ValueError(f"Unknown index type: {kind}")

    def _sharded(self) -> bool:
        return bool(self.config.sharded) or os.environ.get("CSS_STORAGE_SHARDED") == "1"

    def _read_index(self, path: str):
        if self._sharded():
            from .sharded import read_index_sharded

            return read_index_sharded(path, device=self.config.device)
        return fi.read_index(path, device=self.config.device)

    def _init_faiss(self) -> None:
        if self.config.use_gpu and self._gpu_capability is None:
            # assessed here, not in __init__, so no HIP context exists before a fork
            self._gpu_capability = assess_gpu_capability()
            if not self._gpu_capability.can_use_gpu:
                self.logger.warning(f"GPU requested but not available: {self._gpu_capability.status_message}")
        self.faiss_index = self._create_cpu_index()
        self._is_gpu_index = True
        self._saved_rows = -1
        self.logger.info(f"Initialized HIP flat index: {type(self.faiss_index).__name__} on device {self.config.device}")
        if self._gpu_capability:
            log_gpu_status(self._gpu_capability, self.logger)

    def _convert_to_gpu_index(self, cpu_index):
        return fi.index_cpu_to_gpu(self._gpu_resources, self.config.device, cpu_index)

    def _convert_to_cpu_index(self, gpu_index):
        return fi.index_gpu_to_cpu(gpu_index)

    def _recover_interrupted_compaction(self) -> None:
        """Finish or discard a compaction (``_rebuild_faiss_index``) that a crash interrupted.  Its order is:
        (1) compacted rows -> ``<index>.compact``; (2) ONE SQLite transaction renumbers the ids and sets
        ``storage_meta['pending_compact']``; (3) ``os.replace(<index>.compact, <index>)``; (4) the flag is cleared.
        So: flag set -> SQLite already speaks the new numbering and the compacted file is either still beside the
        index (finish step 3) or already in place; flag absent -> a leftover ``.compact`` was never committed."""
        compact = Path(str(self.index_path) + ".compact")
        row = self.db.execute("SELECT value FROM storage_meta WHERE key = 'pending_compact'").fetchone()
        if row is not None:
            if compact.exists():
                os.replace(str(compact), str(self.index_path))
                _fsync_dir(self.index_path.parent)
                self.logger.warning("Completed an interrupted index compaction (compacted file moved into place)")
            self.db.execute("DELETE FROM storage_meta WHERE key = 'pending_compact'")
            self.db.commit()
        elif compact.exists():
            compact.unlink()
            self.logger.warning("Removed the leftover of an index compaction that never committed")

    def _load_existing_data(self) -> None:
        self._recover_interrupted_compaction()
        if not self.index_path.exists():
            return
        try:
            loaded = self._read_index(str(self.index_path))
            if loaded.d != self.embedding_dim:
                raise RuntimeError(f"index file has d={loaded.d}, expected {self.embedding_dim}")
            self.faiss_index = loaded
            self._saved_rows = loaded.ntotal
            self.logger.info(f"Loaded flat index with {loaded.ntotal} vectors")
            self._rebuild_id_mappings()
            bad = [i for i in self.faiss_id_to_chunk_id if i >= loaded.ntotal]
            if bad:   # e.g. a crash before the last save with auto_save off: those chunks have no vector on file
                self.logger.warning(f"{len(bad)} chunks refer to rows beyond the {loaded.ntotal} rows of the index file; "
                                    "they cannot be returned by searches until they are re-indexed")
        except Exception as e:  # corrupt / foreign file -> fresh index (src/storage.py:314-316)
            self.logger.warning(f"Could not load existing FAISS index: {e}")
            self._init_faiss()

    def _rebuild_id_mappings(self) -> None:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        cur = self.db.cursor()
        cur.execute("SELECT id, faiss_id FROM chunks WHERE faiss_id IS NOT NULL")
        fwd: Dict[str, int] = {}
        rev: Dict[int, str] = {}
        for chunk_id, faiss_id in cur.fetchall():
            fwd[chunk_id] = faiss_id
            rev[faiss_id] = chunk_id
        self.chunk_id_to_faiss_id.update(fwd)
        self.faiss_id_to_chunk_id.update(rev)
        self.total_chunks = len(self.chunk_id_to_faiss_id)
        self.logger.info(f"Rebuilt ID mappings for {self.total_chunks} chunks")

    # ------------------------------------------------------------------ add
    def add_chunks(self, chunks: List[Chunk]) -> None:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        if not chunks:
            return
        with_emb = [c for c in chunks if c.embedding is not None]
        if not with_emb:
            self.logger.warning("No chunks with embeddings to add")
            return
        if all(isinstance(c.embedding, np.ndarray) for c in with_emb):
            x = np.stack([c.embedding for c in with_emb]).astype(np.float32, copy=False)  # EmbeddingConfig.embeddings_as_arrays
        else:
            x = np.array([c.embedding for c in with_emb], dtype=np.float32)
        if not self.faiss_index:
            raise RuntimeError("FAISS index not initialized")
        if not self.db:
            raise RuntimeError("Database not initialized")
        with self._lock:
            first_id = self.faiss_index.ntotal
            # x / (||x|| + 1e-8) happens inside the ingest kernel when requested
            self.faiss_index.add(x, normalize=self.config.normalize_embeddings)
            now = datetime.now().isoformat()
            rows = []
            for off, chunk in enumerate(with_emb):
                fid = first_id + off
                self.chunk_id_to_faiss_id[chunk.id] = fid
                self.faiss_id_to_chunk_id[fid] = chunk.id
                md = chunk.metadata
                rows.append(
                    (chunk.id, chunk.text, json.dumps(md), fid, md.get("session_id"), md.get("project_name"),
                     md.get("file_path"), md.get("chunk_type"), md.get("timestamp"), md.get("has_code", False),
                     md.get("has_tools", False), md.get("message_count", 0), md.get("char_count", 0),
                     md.get("word_count", 0), now)
                )
            marks = ", ".join("?" for _ in _CHUNK_COLUMNS)
            self.db.cursor().executemany(
                f"INSERT OR REPLACE INTO chunks ({', '.join(_CHUNK_COLUMNS)}) VALUES ({marks})", rows
            )
            self.db.commit()
            self.total_chunks += len(with_emb)
            if self.config.auto_save:
                self.save_index()
        self.logger.info(f"Added {len(with_emb)} chunks to storage")

    # --------------------------------------------------------------- search
    def search(self, query_embedding, config: Optional[SearchConfig] = None,
               filters: Optional[Dict[str, Any]] = None) -> List[SearchResult]:
        cfg = config or SearchConfig()
        if not self.faiss_index:
            return []
        with self._lock:
            ntotal = self.faiss_index.ntotal
            if ntotal == 0:
                return []
            k = min(cfg.max_results, ntotal, fi.MAX_K)   # (fi.MAX_K = 2048: beyond 128 the index takes passes of 128)
            if k <= 0:
                return []
            # accepts ndarray or a plain list (tests/test_integration.py:203-204 of the reference)
            q = np.asarray(query_embedding, dtype=np.float32).reshape(1, -1)
            allow = None
            if self.config.filter_pushdown and (filters or len(self.faiss_id_to_chunk_id) < ntotal):
                allow = self._allow_mask(filters or {}, ntotal)
                k = max(1, min(k, cfg.top_k, fi.MAX_K))
            if allow is not None:
                sims, ids = self.faiss_index.search(q, k, normalize=self.config.normalize_embeddings, allow=allow)
            else:
                sims, ids = self.faiss_index.search(q, k, normalize=self.config.normalize_embeddings)
            out: List[SearchResult] = []
            for score, fid in zip(sims[0].tolist(), ids[0].tolist()):
                if score < cfg.similarity_threshold:
                    continue
                chunk_id = self.faiss_id_to_chunk_id.get(fid)
                if not chunk_id:  # tombstone: row deleted from SQLite, vector still in the index
                    continue
                data = self._get_chunk_data(chunk_id)
                if not data:
                    continue
                if filters and not self._matches_filters(data, filters):
                    continue
                res = SearchResult(chunk_id=chunk_id, similarity=float(score))
                meta = None
                if cfg.include_metadata:
                    meta = json.loads(data["metadata"]) if data["metadata"] else {}
                    res.metadata = meta
                if cfg.include_text:
                    res.text = data["text"]
                if cfg.include_metadata and cfg.include_text:
                    res.chunk = Chunk(id=chunk_id, text=data["text"], metadata=dict(meta), embedding=None)
                out.append(res)
                if len(out) >= cfg.top_k:
                    break
            return out

    def _allow_mask(self, filters: Dict[str, Any], ntotal: int) -> np.ndarray:
        """Boolean mask over index rows: live (not tombstoned) and matching ``filters`` under exactly the
        semantics of ``_matches_filters``.  One pass over the chunks table, cached until the id maps change."""
        key = json.dumps(filters, sort_keys=True, default=str)
        hit = self._allow_cache.get(key)
        stamp = (self._mutations, len(self.faiss_id_to_chunk_id), ntotal, self.total_chunks)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        allow = np.zeros(ntotal, dtype=bool)
        if not self.db:
            raise RuntimeError("Database not initialized")
        for row in self.db.cursor().execute("SELECT * FROM chunks"):
            fid = self.chunk_id_to_faiss_id.get(row["id"])
            if fid is None or fid >= ntotal or self.faiss_id_to_chunk_id.get(fid) != row["id"]:
                continue
            if filters and not self._matches_filters({k_: row[k_] for k_ in row.keys()}, filters):
                continue
            allow[fid] = True
        if len(self._allow_cache) >= 8:
            self._allow_cache.clear()
        self._allow_cache[key] = (stamp, allow)
        return allow

    def _get_chunk_data(self, chunk_id: str) -> Optional[Dict[str, Any]]:
        if not self.db:
            raise RuntimeError("Database not initialized")
        row = self.db.cursor().execute("SELECT * FROM chunks WHERE id = ?", (chunk_id,)).fetchone()
        return {key: row[key] for key in row.keys()} if row else None

    _RANGE_OPS = {
        "gte": lambda v, b: v >= b,
        "lte": lambda v, b: v <= b,
        "gt": lambda v, b: v > b,
        "lt": lambda v, b: v < b,
    }

    def _matches_filters(self, chunk_data: Dict[str, Any], filters: Dict[str, Any]) -> bool:
        """Range dict / list-IN / case-insensitive substring for project_name /
        exact otherwise; keys absent from the row are ignored (``src/storage.py:508-543``)."""
        for key, want in filters.items():
            if key not in chunk_data:
                continue
            have = chunk_data[key]
            if isinstance(want, dict):
                for op, bound in want.items():
                    test = self._RANGE_OPS.get(op)
                    if test is not None and not test(have, bound):
                        return False
            elif isinstance(want, list):
                if have not in want:
                    return False
            elif key == "project_name" and isinstance(want, str) and isinstance(have, str):
                if want.lower() not in have.lower():
                    return False
            elif have != want:
                return False
        return True

    # ------------------------------------------------------------- metadata
    def get_chunk_by_id(self, chunk_id: str) -> Optional[Chunk]:
        if not self.db:
            raise RuntimeError("Database not initialized")
        data = self._get_chunk_data(chunk_id)
        if not data:
            return None
        meta = json.loads(data["metadata"]) if data["metadata"] else {}
        return Chunk(id=chunk_id, text=data["text"], metadata=meta, embedding=None)

    def _chunks_where(self, column: str, value: str) -> List[Chunk]:
        if not self.db:
            raise RuntimeError("Database not initialized")
        cur = self.db.cursor()
        cur.execute(f"SELECT * FROM chunks WHERE {column} = ? ORDER BY timestamp", (value,))
        return [_row_to_chunk(r) for r in cur.fetchall()]

    def get_chunks_by_session(self, session_id: str) -> List[Chunk]:
        return self._chunks_where("session_id", session_id)

    def get_chunks_by_project(self, project_name: str) -> List[Chunk]:
        return self._chunks_where("project_name", project_name)

    def delete_chunk(self, chunk_id: str) -> bool:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        with self._lock:
            fid = self.chunk_id_to_faiss_id.get(chunk_id)
            if fid is None:
                return False
            cur = self.db.cursor()
            cur.execute("DELETE FROM chunks WHERE id = ?", (chunk_id,))
            if cur.rowcount == 0:
                return False
            # the vector stays in the index as a tombstone (reference behaviour)
            del self.chunk_id_to_faiss_id[chunk_id]
            del self.faiss_id_to_chunk_id[fid]
            self.db.commit()
            self.total_chunks -= 1
            return True

    def delete_chunks_by_session(self, session_id: str) -> int:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        if not self.db:
            raise RuntimeError("Database not initialized")
        cur = self.db.cursor()
        cur.execute("SELECT id FROM chunks WHERE session_id = ?", (session_id,))
        return sum(1 for (cid,) in cur.fetchall() if self.delete_chunk(cid))

    def get_stats(self) -> Dict[str, Any]:
        if not self.db:
            raise RuntimeError("Database not initialized")
        cur = self.db.cursor()
        one = lambda sql: cur.execute(sql).fetchone()[0]  # noqa: E731
        total_chunks = one("SELECT COUNT(*) FROM chunks")
        total_sessions = one("SELECT COUNT(DISTINCT session_id) FROM chunks")
        total_projects = one("SELECT COUNT(DISTINCT project_name) FROM chunks")
        chunk_types = dict(cur.execute("SELECT chunk_type, COUNT(*) FROM chunks GROUP BY chunk_type").fetchall())
        try:
            projects = self.get_all_projects()
        except Exception as e:
            self.logger.warning(f"Failed to get projects list: {e}")
            projects = []
        index_bytes = self.index_path.stat().st_size if self.index_path.exists() else 0
        db_bytes = self.db_path.stat().st_size if self.db_path.exists() else 0
        stats: Dict[str, Any] = {
            "total_chunks": total_chunks,
            "total_sessions": total_sessions,
            "total_projects": total_projects,
            "projects": projects,
            "chunk_types": chunk_types,
            "faiss_index_size": index_bytes,
            "database_size": db_bytes,
            "total_storage_size": index_bytes + db_bytes,
            "embedding_dimension": self.embedding_dim,
            "index_type": self.config.index_type,
            "use_gpu": self.config.use_gpu,
            "is_gpu_index": self._is_gpu_index,
        }
        cap = self._gpu_capability
        if cap:
            info = {
                "gpu_available": cap.can_use_gpu,
                "gpu_count": cap.gpu_count,
                "gpu_names": cap.gpu_names,
                "status_message": cap.status_message,
            }
            if cap.gpu_memory_total is not None:
                info["gpu_memory_total_gb"] = cap.gpu_memory_total / (1024**3)
            if cap.gpu_memory_free is not None:
                info["gpu_memory_free_gb"] = cap.gpu_memory_free / (1024**3)
            stats["gpu_info"] = info
        return stats

    def get_all_projects(self) -> List[str]:
        if not self.db:
            raise RuntimeError("Database not initialized. Call initialize() first.")
        cur = self.db.cursor()
        cur.execute(
            "SELECT DISTINCT project_name FROM chunks "
            "WHERE project_name IS NOT NULL AND project_name != '' ORDER BY project_name"
        )
        return [r[0] for r in cur.fetchall()]

    # --------------------------------------------------------- file tracking
    def update_file_info(self, file_path: str, chunk_count: int) -> None:
        if not self.db:
            raise RuntimeError("Database not initialized")
        try:
            modified = datetime.fromtimestamp(os.path.getmtime(file_path))
        except OSError:
            modified = datetime.now()
        self.db.cursor().execute(
            "INSERT OR REPLACE INTO files (path, last_modified, last_indexed, chunk_count) VALUES (?, ?, ?, ?)",
            (file_path, modified, datetime.now(), chunk_count),
        )
        self.db.commit()

    def is_file_modified(self, file_path: str) -> bool:
        try:
            current = datetime.fromtimestamp(os.path.getmtime(file_path))
        except OSError:
            return True
        row = self.db.cursor().execute(
            "SELECT last_modified, last_indexed FROM files WHERE path = ?", (file_path,)
        ).fetchone()
        if not row or not row["last_modified"]:
            return True
        return current > datetime.fromisoformat(row["last_modified"])

    def remove_chunks_for_file(self, file_path: str) -> int:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        with self._lock:
            cur = self.db.cursor()
            doomed = cur.execute("SELECT id, faiss_id FROM chunks WHERE file_path = ?", (file_path,)).fetchall()
            if not doomed:
                return 0
            cur.execute("DELETE FROM chunks WHERE file_path = ?", (file_path,))
            self.db.commit()
            for row in doomed:
                self.chunk_id_to_faiss_id.pop(row["id"], None)
                if row["faiss_id"] is not None:
                    self.faiss_id_to_chunk_id.pop(row["faiss_id"], None)
            return len(doomed)

    def clear_all_data(self) -> None:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        with self._lock:
            self.faiss_index = self._create_cpu_index()
            self._saved_rows = -1
            cur = self.db.cursor()
            cur.execute("DELETE FROM chunks")
            cur.execute("DELETE FROM files")
            self.db.commit()
            self.chunk_id_to_faiss_id.clear()
            self.faiss_id_to_chunk_id.clear()
            self.total_chunks = 0
            if self.config.auto_save:
                self.save_index()
        self.logger.info("Cleared all data from storage")

    # ---------------------------------------------------------- persistence
    def save_index(self) -> None:
        """Write ``embeddings.faiss`` (IndexFlat on-disk layout).  When the file
        already holds the first ``_saved_rows`` rows only the new rows are appended
        and the two counters in the header are patched."""
        if not self.faiss_index:
            self.logger.warning("No FAISS index to save")
            return
        with self._lock:
            ix = self.faiss_index
            n = ix.ntotal
            path = str(self.index_path)
            if 0 <= self._saved_rows <= n and self.index_path.exists() and self._saved_rows > 0:
                if n > self._saved_rows:
                    new_rows = ix.reconstruct_n(self._saved_rows, n - self._saved_rows)
                    with open(path, "r+b") as f:
                        # rows first, made durable, THEN the two header counters: a crash in between leaves a
                        # file whose header still describes the old, complete prefix (trailing bytes are ignored
                        # by read_index), never a header that promises rows the file does not hold
                        f.seek(_INDEX_HEADER_BYTES + self._saved_rows * ix.d * 4)
                        f.write(new_rows.tobytes())
                        f.truncate()
                        f.flush()
                        os.fsync(f.fileno())
                        f.seek(8)
                        f.write(struct.pack("<q", n))
                        f.seek(8 + 8 + 16 + 1 + 4)
                        f.write(struct.pack("<Q", n * ix.d))
                        f.flush()
                        os.fsync(f.fileno())
            else:
                self._write_index_atomically(ix, path)
            self._saved_rows = n
        self.logger.info(f"Saved flat index ({n} vectors) to {self.index_path}")

    def _crash_point(self, where: str) -> None:
        """Test seam: tests replace this to simulate a crash between the steps of a journaled operation."""

    @staticmethod
    def _write_index_atomically(ix, path: str) -> None:
        """Whole-file write through a temporary sibling + ``os.replace``: readers (and a crash) see the old file or
        the new one, never a torn one; the directory entry is made durable too."""
        tmp = path + ".tmp"
        fi.write_index(ix, tmp)
        with open(tmp, "rb") as f:
            os.fsync(f.fileno())
        os.replace(tmp, path)
        _fsync_dir(Path(path).parent)

    def backup(self, backup_dir: str) -> None:
        dest = Path(backup_dir)
        dest.mkdir(parents=True, exist_ok=True)
        if self.faiss_index and self.faiss_index.ntotal > 0:
            fi.write_index(self.faiss_index, str(dest / self.config.index_name))
        if self.db_path.exists() and self.db:
            target = sqlite3.connect(str(dest / self.config.db_name))
            self.db.backup(target)
            target.close()
        self.logger.info(f"Backup created in {dest}")

    def restore(self, backup_dir: str) -> None:
        self._mutations += 1  # invalidates cached allow masks (filter push-down)
        src = Path(backup_dir)
        with self._lock:
            ipath = src / self.config.index_name
            if ipath.exists():
                self.faiss_index = self._read_index(str(ipath))
                self._saved_rows = -1
            dpath = src / self.config.db_name
            if dpath.exists():
                self.db.close()
                self.db = sqlite3.connect(str(self.db_path), check_same_thread=False)
                self.db.row_factory = sqlite3.Row
                source = sqlite3.connect(str(dpath))
                source.backup(self.db)
                source.close()
            self._rebuild_id_mappings()
        self.logger.info(f"Restored from backup in {src}")

    def optimize(self) -> None:
        self.logger.info("Optimizing storage...")
        self.db.execute("VACUUM")
        if self.total_chunks != self.faiss_index.ntotal:
            self.logger.info("Rebuilding flat index...")
            self._rebuild_faiss_index()
        self.logger.info("Storage optimization complete")

    def _rebuild_faiss_index(self) -> None:
        """Compact tombstones: keep only rows still referenced by SQLite, in
        faiss_id order, and renumber.  (The reference leaves this as a stub that
        would drop every vector, ``src/storage.py:944-969``; the vectors are
        available here because the index can export its rows.)"""
        with self._lock:
            cur = self.db.cursor()
            live = cur.execute(
                "SELECT id, faiss_id FROM chunks WHERE faiss_id IS NOT NULL ORDER BY faiss_id"
            ).fetchall()
            if not live:
                return
            old = self.faiss_index
            fresh = self._create_cpu_index()
            fresh.reserve(len(live))
            ids = [r["faiss_id"] for r in live]
            step = 1 << 16
            for s in range(0, len(ids), step):
                part = ids[s:s + step]
                lo, hi = part[0], part[-1] + 1
                block = old.reconstruct_n(lo, hi - lo)
                fresh.add(block[np.asarray(part) - lo])  # already normalised
            fwd: Dict[str, int] = {}
            rev: Dict[int, str] = {}
            updates = []
            for new_id, row in enumerate(live):
                fwd[row["id"]] = new_id
                rev[new_id] = row["id"]
                updates.append((new_id, row["id"]))
            # The renumbered ids only make sense with the compacted rows, whatever auto_save says.  Two files cannot
            # change atomically together, so the step is journaled (see _recover_interrupted_compaction): compacted
            # file beside the index, then ONE transaction with the new ids + a flag, then the rename, then the flag
            # is cleared.  A crash at any point leaves a state the next initialize() completes or discards.
            compact = str(self.index_path) + ".compact"
            self._write_index_atomically(fresh, compact)
            cur.executemany("UPDATE chunks SET faiss_id = ? WHERE id = ?", updates)
            cur.execute("INSERT OR REPLACE INTO storage_meta (key, value) VALUES ('pending_compact', ?)", (str(len(live)),))
            self.db.commit()
            self._crash_point("compaction committed, file not yet moved")
            os.replace(compact, str(self.index_path))
            _fsync_dir(self.index_path.parent)
            cur.execute("DELETE FROM storage_meta WHERE key = 'pending_compact'")
            self.db.commit()
            self.chunk_id_to_faiss_id = fwd     # (in-memory maps change only once both files have)
            self.faiss_id_to_chunk_id = rev
            self.faiss_index = fresh
            self._saved_rows = fresh.ntotal
            self.total_chunks = len(live)
            self._mutations += 1
            old.close()
        self.logger.info("Flat index rebuilt")

    def close(self) -> None:
        if self.config.auto_save:
            self.save_index()
        if self.db:
            self.db.close()
        self.logger.info("Storage closed")

    def __enter__(self) -> "HybridStorage":
        self.initialize()
        return self

    def __exit__(self, exc_type: Any, exc_val: Any, exc_tb: Any) -> None:
        self.close()
