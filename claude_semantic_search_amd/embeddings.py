"""``EmbeddingGenerator``: chunk texts -> 768-d sentence embeddings on the MI355X.

Drop-in for the reference's ``src/embeddings.py`` (same public names, argument
meaning and error behaviour) with ``SentenceTransformer`` replaced
by ``MpnetEncoder`` (libcss_hip.so).  Reference lines followed:

  * config / stats dataclasses ............................ src/embeddings.py:28-52
  * lazy ``load_model``: model object, ``.to(device)``,
    ``max_seq_length``, GPU batch-size policy, embedding dim  src/embeddings.py:75-134
  * ``generate_embeddings``: ``[]`` for no chunks; writes
    ``chunk.embedding = row.tolist()`` ...................... src/embeddings.py:159-177
  * ``generate_single_embedding`` ........................... src/embeddings.py:179-190
  * text sanitising (None -> "", non-str -> str(x),
    blank -> "empty") + ``encode`` kwargs .................... src/embeddings.py:192-236
  * numpy cosine helpers .................................... src/embeddings.py:238-275
  * stats / npz save+load / validate / benchmark / info ..... src/embeddings.py:277-507

Differences: the model always runs on a HIP device (no CPU model); without local
weights ``load_model`` raises unless ``EmbeddingConfig.synthetic_weights_seed`` is
set (seeded synthetic weights + hashing tokenizer, used by benchmarks and tests).
"""
from __future__ import annotations

import logging
import os
import time
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, Hashable, List, Optional, Sequence, Tuple

import numpy as np

from .chunk import Chunk
from .gpu_utils import assess_gpu_capability, calculate_optimal_batch_size, log_gpu_status


@dataclass
class EmbeddingConfig:
    model_name: str = "all-mpnet-base-v2"
    batch_size: int = 16
    max_seq_length: int = 384
    device: str = "auto"
    use_gpu: bool = False
    auto_batch_size: bool = True
    normalize_embeddings: bool = True
    show_progress: bool = True
    cache_dir: Optional[str] = None
    # extensions (keyword-only in practice: appended after the reference's fields)
    synthetic_weights_seed: Optional[int] = None
    compute: str = "bf16"  # "bf16" (MFMA product path) or "fp32" (verification mode)
    device_index: int = 0
    # keep ``chunk.embedding`` as a float32 ndarray row instead of the reference's Python list (src/embeddings.py:175):
    # the list round trip (tolist here, np.array in add_chunks, src/storage.py:343) costs as much host time per 10k
    # chunks (~0.5 s) as their whole GPU encode.  Off by default (drop-in behaviour).
    embeddings_as_arrays: bool = False


@dataclass
class EmbeddingStats:
    total_chunks: int = 0
    total_tokens: int = 0
    generation_time: float = 0.0
    average_chunk_length: float = 0.0
    throughput_chunks_per_second: float = 0.0
    model_info: Dict[str, Any] = field(default_factory=dict)


class EmbeddingBatcher:
    """Cross-file batching of the index pipeline.

    The reference encodes one conversation file at a time (``src/cli.py:120-169``: ``generate_embeddings(chunks)``
    per file), so on a GPU whose batch is 256 almost every device batch is ragged and small.  This accumulator takes
    the chunk lists of many files, encodes them in FULL batches (``flush_at`` texts per ``model.encode`` call, a
    multiple of the device batch; the encoder length-sorts inside a call) and hands every file back -- through
    ``on_file_done(key, chunks, rows)`` -- as soon as all of its chunks are embedded, in submission order, so that
    the caller can run ``storage.add_chunks(chunks)`` / ``update_file_info`` exactly as the reference does per file.

    ``chunk.embedding`` is written as ``generate_embeddings`` writes it (a list, or an ndarray row with
    ``EmbeddingConfig.embeddings_as_arrays``).  A file may be larger than ``flush_at``; empty files complete at once.
    """

    def __init__(self, generator: "EmbeddingGenerator", on_file_done: Optional[Callable] = None,
                 flush_at: Optional[int] = None):
        self.gen = generator
        self.on_file_done = on_file_done
        self._flush_at = flush_at
        self._pending: List[Tuple[Hashable, List[Chunk], List[Optional[np.ndarray]], int]] = []  # key, chunks, rows, done
        self._queue: List[Tuple[int, int]] = []     # (index into _pending, chunk index) awaiting an encode
        self._base = 0                               # files completed and dropped from the front of _pending
        self.batches: List[int] = []                 # sizes of the encode calls made (diagnostics / tests)

    @property
    def flush_at(self) -> int:
        if self._flush_at:
            return int(self._flush_at)
        return 16 * max(1, int(self.gen.config.batch_size))

    def add(self, key: Hashable, chunks: List[Chunk]) -> None:
        """Queue one file's chunks; encodes whenever a full ``flush_at`` texts are waiting."""
        slot = self._base + len(self._pending)
        self._pending.append((key, chunks, [None] * len(chunks), 0))
        self._queue.extend((slot, i) for i in range(len(chunks)))
        if not chunks:
            self._complete_ready()
        while len(self._queue) >= self.flush_at:
            self._encode(self.flush_at)

    def flush(self) -> None:
        """Encode whatever is still waiting (the last, possibly ragged batch) and complete every file."""
        while self._queue:
            self._encode(min(len(self._queue), self.flush_at))
        self._complete_ready()

    def _encode(self, n: int) -> None:
        take = self._queue[:n]
        texts = [self._pending[s - self._base][1][i].text for s, i in take]
        rows = self.gen._generate_embeddings_batch(texts)   # (a failure leaves the chunks queued: flush() can be retried)
        self._queue = self._queue[n:]
        self.batches.append(len(texts))
        for (s, i), row in zip(take, rows):
            key, chunks, out, done = self._pending[s - self._base]
            out[i] = row
            chunks[i].embedding = row if self.gen.config.embeddings_as_arrays else row.tolist()
            self._pending[s - self._base] = (key, chunks, out, done + 1)
        self._complete_ready()

    def _complete_ready(self) -> None:
        # files complete strictly in submission order (the queue is FIFO), so only the front can be ready
        while self._pending and self._pending[0][3] == len(self._pending[0][1]):
            key, chunks, out, _ = self._pending.pop(0)
            self._base += 1
            if self.on_file_done is not None:
                rows = np.stack(out) if out else np.zeros((0, self.gen._embedding_dim or 768), dtype=np.float32)
                self.on_file_done(key, chunks, rows)


def SentenceTransformer(model_name_or_path: str, cache_folder: Optional[str] = None, **kwargs):
    """Same call shape as ``sentence_transformers.SentenceTransformer(name, cache_folder=...)``
    (``src/embeddings.py:86-88``); returns the HIP encoder."""
    from .mpnet_encoder import MpnetEncoder

    return MpnetEncoder(model_name_or_path, cache_folder=cache_folder, **kwargs)


class EmbeddingGenerator:
    def __init__(self, config: Optional[EmbeddingConfig] = None):
        self.config = config or EmbeddingConfig()
        self.model = None
        self.logger = logging.getLogger(__name__)
        self._embedding_dim: Optional[int] = None
        self._gpu_capability = None  # assessed in load_model(): nothing touches HIP before a fork

    # ------------------------------------------------------------------ model
    def load_model(self) -> None:
        try:
            self.logger.info(f"Loading model: {self.config.model_name}")
            if self.config.use_gpu and self._gpu_capability is None:
                self._gpu_capability = assess_gpu_capability()
                if not self._gpu_capability.can_use_gpu:
                    self.logger.warning(f"GPU requested but not available: {self._gpu_capability.status_message}")
            cache_dir = self.config.cache_dir
            if cache_dir:
                os.environ["SENTENCE_TRANSFORMERS_HOME"] = cache_dir
            extras: Dict[str, Any] = {}
            if self.config.synthetic_weights_seed is not None:
                extras["synthetic_seed"] = self.config.synthetic_weights_seed
            if self.config.compute != "bf16":
                extras["compute"] = self.config.compute
            if self.config.device_index != 0:
                extras["device"] = self.config.device_index
            self.model = SentenceTransformer(self.config.model_name, cache_folder=cache_dir, **extras)
            target = self._determine_target_device()
            self.model.to(target)
            self.model.max_seq_length = self.config.max_seq_length
            cap = self._gpu_capability
            if self.config.use_gpu and self.config.auto_batch_size and cap and cap.gpu_memory_free:
                self.config.batch_size = calculate_optimal_batch_size(cap.gpu_memory_free / (1024**3), backend="cuda")
                self.logger.info(f"Auto-adjusted batch size for GPU (hip): {self.config.batch_size}")
            self._embedding_dim = self.model.get_sentence_embedding_dimension()
            self.logger.info(f"Model loaded successfully on {self.model.device}. Embedding dimension: {self._embedding_dim}")
            if self.config.use_gpu and cap:
                log_gpu_status(cap, self.logger)
        except Exception as e:
            self.logger.error(f"Failed to load model {self.config.model_name}: {e}")
            raise

    def _determine_target_device(self) -> str:
        if self.config.device != "auto":
            return self.config.device
        return f"cuda:{self.config.device_index}"

    # ---------------------------------------------------------------- encode
    def generate_embeddings(self, chunks: List[Chunk]):
        if not self.model:
            self.load_model()
        if not chunks:
            return []
        embeddings = self._generate_embeddings_batch([c.text for c in chunks])
        if self.config.embeddings_as_arrays:
            for chunk, row in zip(chunks, embeddings):
                chunk.embedding = row
        else:
            for chunk, row in zip(chunks, embeddings):
                chunk.embedding = row.tolist()
        return embeddings

    def generate_embeddings_many(self, chunk_lists: Sequence[List[Chunk]], flush_at: Optional[int] = None) -> List[np.ndarray]:
        """The chunks of MANY files through full device batches (``EmbeddingBatcher``): per file the same result as
        ``generate_embeddings(chunks)`` -- ``chunk.embedding`` set, an ``[n_i, 768]`` array returned -- without the
        ragged per-file batches of the reference's loop (``src/cli.py:120-169``)."""
        if not self.model:
            self.load_model()
        out: List[Optional[np.ndarray]] = [None] * len(chunk_lists)

        def done(key, _chunks, rows):
            out[key] = rows

        b = EmbeddingBatcher(self, on_file_done=done, flush_at=flush_at)
        for i, chunks in enumerate(chunk_lists):
            b.add(i, chunks)
        b.flush()
        return out  # type: ignore[return-value]

    def generate_single_embedding(self, text: str) -> np.ndarray:
        if not self.model:
            self.load_model()
        return self.model.encode(text, normalize_embeddings=self.config.normalize_embeddings, show_progress_bar=False)

    def _generate_embeddings_batch(self, texts: List[str]):
        t0 = time.time()
        clean: List[str] = []
        for i, text in enumerate(texts):
            if text is None:
                self.logger.warning(f"Skipping chunk {i}: text is None")
                clean.append("")
            elif not isinstance(text, str):
                self.logger.warning(f"Skipping chunk {i}: text is not string (type: {type(text)})")
                clean.append(str(text) if text else "")
            elif not text.strip():
                self.logger.warning(f"Skipping chunk {i}: text is empty or whitespace only")
                clean.append("empty")
            else:
                clean.append(text)
        out = self.model.encode(
            clean,
            batch_size=self.config.batch_size,
            normalize_embeddings=self.config.normalize_embeddings,
            show_progress_bar=self.config.show_progress,
            convert_to_numpy=True,
        )
        dt = time.time() - t0
        if self.config.show_progress:
            rate = len(texts) / dt if dt > 0 else 0
            avg = np.mean([len(t) if isinstance(t, str) else 0 for t in texts])
            self.logger.info(f"Generated {len(texts)} embeddings in {dt:.2f}s ({rate:.1f} chunks/s, avg length: {avg:.0f} chars)")
        return out

    # ---------------------------------------------------------------- cosine helpers (numpy, as the reference)
    def compute_similarity(self, embedding1: np.ndarray, embedding2: np.ndarray) -> float:
        return np.dot(embedding1, embedding2) / (np.linalg.norm(embedding1) * np.linalg.norm(embedding2))

    def compute_similarity_matrix(self, embeddings: List[np.ndarray]) -> np.ndarray:
        n = len(embeddings)
        m = np.zeros((n, n))
        for i in range(n):
            for j in range(i, n):
                m[i, j] = m[j, i] = self.compute_similarity(embeddings[i], embeddings[j])
        return m

    def find_similar_chunks(self, query_embedding: np.ndarray, chunk_embeddings: List[np.ndarray], top_k: int = 5) -> List[tuple]:
        sims = [(i, self.compute_similarity(query_embedding, e)) for i, e in enumerate(chunk_embeddings)]
        sims.sort(key=lambda t: t[1], reverse=True)
        return sims[:top_k]

    # ---------------------------------------------------------------- stats / io
    def _model_info_small(self) -> Dict[str, Any]:
        return {
            "model_name": self.config.model_name,
            "embedding_dimension": self._embedding_dim,
            "max_seq_length": self.config.max_seq_length,
            "device": str(self.model.device) if hasattr(self.model, "device") else "unknown",
        }

    def get_embedding_stats(self, chunks: List[Chunk]) -> EmbeddingStats:
        if not chunks:
            return EmbeddingStats()
        return EmbeddingStats(
            total_chunks=len(chunks),
            total_tokens=sum(len(c.text.split()) for c in chunks),
            average_chunk_length=np.mean([len(c.text) for c in chunks]),
            model_info=self._model_info_small() if self.model else {},
        )

    def save_embeddings(self, chunks: List[Chunk], file_path: str) -> None:
        rows = [{"chunk_id": c.id, "embedding": c.embedding, "text": c.text, "metadata": c.metadata}
                for c in chunks if c.embedding]
        np.savez_compressed(file_path, embeddings=rows)
        self.logger.info(f"Saved {len(rows)} embeddings to {file_path}")

    def load_embeddings(self, file_path: str) -> List[Chunk]:
        data = np.load(file_path, allow_pickle=True)
        chunks = [Chunk(id=it["chunk_id"], text=it["text"], metadata=it["metadata"], embedding=it["embedding"])
                  for it in data["embeddings"]]
        self.logger.info(f"Loaded {len(chunks)} embeddings from {file_path}")
        return chunks

    def validate_embeddings(self, chunks: List[Chunk]) -> Dict[str, Any]:
        res: Dict[str, Any] = {"total_chunks": len(chunks), "chunks_with_embeddings": 0, "embedding_dimension": None,
                               "embedding_stats": {}, "issues": []}
        vecs = []
        for c in chunks:
            if not c.embedding:
                res["issues"].append(f"Missing embedding for chunk {c.id}")
                continue
            res["chunks_with_embeddings"] += 1
            vecs.append(np.array(c.embedding))
            if res["embedding_dimension"] is None:
                res["embedding_dimension"] = len(c.embedding)
            elif res["embedding_dimension"] != len(c.embedding):
                res["issues"].append(f"Inconsistent embedding dimension for chunk {c.id}")
        if vecs:
            if len({len(v) for v in vecs}) == 1:
                a = np.array(vecs)
                norms = np.linalg.norm(a, axis=1)
                res["embedding_stats"] = {"mean": a.mean(0).tolist(), "std": a.std(0).tolist(), "min": a.min(0).tolist(),
                                          "max": a.max(0).tolist(), "norm_mean": np.mean(norms), "norm_std": np.std(norms)}
            else:
                norms = [np.linalg.norm(v) for v in vecs]
                res["embedding_stats"] = {"norm_mean": np.mean(norms), "norm_std": np.std(norms),
                                          "note": "Embeddings have different dimensions, limited stats computed"}
        return res

    def benchmark_model(self, test_texts: List[str], warmup_runs: int = 3) -> Dict[str, Any]:
        if not self.model:
            self.load_model()
        for _ in range(warmup_runs):
            self.model.encode(test_texts[: min(5, len(test_texts))], show_progress_bar=False)
        perf = {}
        for bs in (1, 4, 8, 16, 32):
            if bs > len(test_texts):
                continue
            t0 = time.time()
            for i in range(0, len(test_texts), bs):
                self.model.encode(test_texts[i:i + bs], show_progress_bar=False)
            dt = time.time() - t0
            perf[f"batch_size_{bs}"] = {"total_time": dt, "throughput": len(test_texts) / dt,
                                        "avg_time_per_text": dt / len(test_texts)}
        mem = {}
        try:
            from . import _native as nat

            info = nat.device_info(self.config.device_index)
            used = info["hbm_total_bytes"] - info["hbm_free_bytes"]
            mem = {"allocated": used, "reserved": used, "max_allocated": used}
        except Exception:
            pass
        return {
            "model_name": self.config.model_name,
            "device": str(self.model.device) if hasattr(self.model, "device") else "unknown",
            "embedding_dimension": self._embedding_dim,
            "test_texts_count": len(test_texts),
            "performance": perf,
            "memory_info": mem,
        }

    @property
    def embedding_dimension(self) -> Optional[int]:
        return self._embedding_dim

    @property
    def is_model_loaded(self) -> bool:
        return self.model is not None

    def get_model_info(self) -> Dict[str, Any]:
        if not self.model:
            return {}
        cap = self._gpu_capability
        info = dict(self._model_info_small(), batch_size=self.config.batch_size, use_gpu=self.config.use_gpu,
                    gpu_available=cap.can_use_gpu if cap else False)
        problem = getattr(self.model, "tokenizer_problem", None)
        if problem:   # real weights without their vocabulary: text is refused rather than hashed (build addition)
            info["tokenizer_problem"] = problem
        if cap and cap.can_use_gpu:
            info["gpu_info"] = {
                "gpu_count": cap.gpu_count,
                "gpu_names": cap.gpu_names,
                "gpu_memory_total_gb": cap.gpu_memory_total / (1024**3) if cap.gpu_memory_total else 0,
                "gpu_memory_free_gb": cap.gpu_memory_free / (1024**3) if cap.gpu_memory_free else 0,
                "recommended_batch_size": cap.recommended_batch_size,
            }
        return info

    @property
    def is_using_gpu(self) -> bool:
        if not self.model:
            return False
        dev = str(self.model.device) if hasattr(self.model, "device") else ""
        return "cuda" in dev.lower() or self.config.use_gpu
