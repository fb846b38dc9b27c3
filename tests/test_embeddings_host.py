"""CPU: EmbeddingGenerator host logic and call contract, modelled on the
reference's tests/test_embeddings.py (which patches SentenceTransformer the same way)."""
import os
from unittest.mock import Mock, patch

import numpy as np
import pytest

from claude_semantic_search_amd.chunk import Chunk
from claude_semantic_search_amd.embeddings import EmbeddingConfig, EmbeddingGenerator, EmbeddingStats

SEAM = "claude_semantic_search_amd.embeddings.SentenceTransformer"


def _chunks():
    return [Chunk("c1", "first text", {"a": 1}), Chunk("c2", "second text", {}), Chunk("c3", "third", {})]


def _mock_model(dim=3, out=None):
    m = Mock()
    m.get_sentence_embedding_dimension.return_value = dim
    m.device = "cuda:0"
    if out is not None:
        m.encode.return_value = out
    return m


def test_config_defaults_match_reference():
    c = EmbeddingConfig()  # src/embeddings.py:28-40
    assert (c.model_name, c.batch_size, c.max_seq_length, c.device) == ("all-mpnet-base-v2", 16, 384, "auto")
    assert c.use_gpu is False and c.auto_batch_size and c.normalize_embeddings and c.show_progress and c.cache_dir is None
    assert EmbeddingStats().total_chunks == 0
    g = EmbeddingGenerator()
    assert g.model is None and g._embedding_dim is None and not g.is_model_loaded and g.get_model_info() == {}


@patch(SEAM)
def test_load_model_call_contract(st):  # tests/test_embeddings.py:110-139
    st.return_value = _mock_model(384)
    g = EmbeddingGenerator(EmbeddingConfig(model_name="all-MiniLM-L6-v2", batch_size=4, show_progress=False))
    g.load_model()
    st.assert_called_once_with("all-MiniLM-L6-v2", cache_folder=None)
    assert g.is_model_loaded and g.embedding_dimension == 384
    assert g.model.max_seq_length == 384
    st.reset_mock()
    g2 = EmbeddingGenerator(EmbeddingConfig(cache_dir="/tmp/cache"))
    g2.load_model()
    st.assert_called_once_with("all-mpnet-base-v2", cache_folder="/tmp/cache")
    assert os.environ.get("SENTENCE_TRANSFORMERS_HOME") == "/tmp/cache"


@patch(SEAM)
def test_load_model_failure_reraises(st):  # :141-147
    st.side_effect = Exception("Model loading failed")
    with pytest.raises(Exception, match="Model loading failed"):
        EmbeddingGenerator().load_model()


@patch(SEAM)
def test_single_embedding_encode_kwargs(st):  # :149-166
    vec = np.array([0.1, 0.2, 0.3])
    st.return_value = _mock_model(3, vec)
    g = EmbeddingGenerator(EmbeddingConfig(show_progress=False))
    out = g.generate_single_embedding("test text")
    np.testing.assert_array_equal(out, vec)
    g.model.encode.assert_called_once_with("test text", normalize_embeddings=True, show_progress_bar=False)


@patch(SEAM)
def test_batch_encode_kwargs_and_tolist(st):  # :168-204
    rows = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.7, 0.8, 0.9]])
    st.return_value = _mock_model(3, rows)
    g = EmbeddingGenerator(EmbeddingConfig(batch_size=4, show_progress=False))
    chunks = _chunks()
    out = g.generate_embeddings(chunks)
    assert len(out) == 3
    for i, c in enumerate(chunks):
        assert c.embedding == rows[i].tolist()
    g.model.encode.assert_called_once_with([c.text for c in chunks], batch_size=4, normalize_embeddings=True,
                                           show_progress_bar=False, convert_to_numpy=True)
    assert EmbeddingGenerator().generate_embeddings([]) == []


@patch(SEAM)
def test_text_sanitising(st):  # src/embeddings.py:197-213
    st.return_value = _mock_model(3, np.zeros((4, 3)))
    g = EmbeddingGenerator(EmbeddingConfig(show_progress=False))
    g._generate_embeddings_batch.__func__  # exists
    g.load_model()
    g._generate_embeddings_batch([None, 123, "   ", "ok"])
    sent = g.model.encode.call_args[0][0]
    assert sent == ["", "123", "empty", "ok"]


def test_cosine_helpers():  # :206-254
    g = EmbeddingGenerator()
    e1, e2, e3 = np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([1.0, 0, 0])
    assert abs(g.compute_similarity(e1, e2)) < 1e-10 and abs(g.compute_similarity(e1, e3) - 1) < 1e-10
    m = g.compute_similarity_matrix([e1, e2, e3])
    assert m.shape == (3, 3) and abs(m[0, 2] - 1) < 1e-10 and abs(m[0, 1]) < 1e-10
    top = g.find_similar_chunks(e1, [e1, e2, np.array([0.7, 0.7, 0])], top_k=2)
    assert [t[0] for t in top] == [0, 2] and top[0][1] == 1.0 and top[1][1] > 0.5


def test_stats_save_load_validate(tmp_path):  # :256-351
    g = EmbeddingGenerator()
    chunks = _chunks()
    for i, c in enumerate(chunks):
        c.embedding = [0.1 * i, 0.2 * i, 0.3 * i]
    st = g.get_embedding_stats(chunks)
    assert st.total_chunks == 3 and st.total_tokens == 5 and st.model_info == {}
    chunks[0].embedding = [0.5, 0.1, 0.2]
    p = str(tmp_path / "emb.npz")
    g.save_embeddings(chunks, p)
    back = g.load_embeddings(p)
    assert [c.id for c in back] == ["c1", "c2", "c3"] and back[0].embedding == [0.5, 0.1, 0.2]
    v = g.validate_embeddings(chunks)
    assert v["total_chunks"] == 3 and v["chunks_with_embeddings"] == 3 and v["embedding_dimension"] == 3 and not v["issues"]
    chunks[1].embedding = None
    chunks[2].embedding = [1.0, 2.0]
    v = g.validate_embeddings(chunks)
    assert any("Missing" in s for s in v["issues"]) and any("Inconsistent" in s for s in v["issues"])


@patch(SEAM)
def test_benchmark_and_info_keys(st):  # :353-419
    st.return_value = _mock_model(3, np.zeros((1, 3)))
    g = EmbeddingGenerator(EmbeddingConfig(show_progress=False))
    r = g.benchmark_model(["a", "b", "c", "d", "e"], warmup_runs=1)
    for key in ("model_name", "device", "embedding_dimension", "test_texts_count", "performance", "memory_info"):
        assert key in r
    assert set(r["performance"]) == {"batch_size_1", "batch_size_4"}
    info = g.get_model_info()
    for key in ("model_name", "embedding_dimension", "max_seq_length", "device", "batch_size", "use_gpu", "gpu_available"):
        assert key in info
    assert g.is_using_gpu  # device string contains "cuda" (what PyTorch-ROCm calls a HIP device)


# ---- cross-file batching (the reference encodes file by file, src/cli.py:120-169) ----
def _hash_model(dim=5):
    """Deterministic stand-in for the encoder: a text's embedding depends on the text alone."""
    m = _mock_model(dim)

    def encode(texts, **_kw):
        if isinstance(texts, str):
            texts = [texts]
        return np.array([[float((hash(t) >> (8 * j)) % 251) for j in range(dim)] for t in texts], dtype=np.float32)

    m.encode.side_effect = encode
    return m


@patch(SEAM)
def test_batcher_runs_full_batches_and_completes_files_in_order(st):
    from claude_semantic_search_amd.embeddings import EmbeddingBatcher

    st.return_value = _hash_model()
    g = EmbeddingGenerator(EmbeddingConfig(batch_size=4, show_progress=False))
    g.load_model()
    sizes = [3, 0, 10, 1, 7, 25, 4]                       # chunks per file; one empty, one larger than flush_at
    files = [[Chunk(f"f{f}c{i}", f"text {f}/{i}", {}) for i in range(n)] for f, n in enumerate(sizes)]
    done = []
    b = EmbeddingBatcher(g, on_file_done=lambda key, chunks, rows: done.append((key, len(chunks), rows.shape)), flush_at=8)
    for f, chunks in enumerate(files):
        b.add(f, chunks)
    assert all(n == 8 for n in b.batches) and len(b.batches) == sum(sizes) // 8      # only full batches so far
    assert [d[0] for d in done] == list(range(len(done)))                            # submission order
    b.flush()
    assert b.batches[-1] == sum(sizes) % 8 and sum(b.batches) == sum(sizes)
    assert [d[0] for d in done] == list(range(len(sizes))) and [d[1] for d in done] == sizes
    assert all(d[2] == (n, 5) for d, n in zip(done, sizes))
    # every chunk got the embedding of ITS text, as a list (reference behaviour, src/embeddings.py:175)
    for chunks in files:
        for c in chunks:
            assert isinstance(c.embedding, list) and c.embedding == g.model.encode([c.text])[0].tolist()
    # encode kwargs are those of generate_embeddings (src/embeddings.py:216-222)
    kw = g.model.encode.call_args_list[0].kwargs
    assert kw["batch_size"] == 4 and kw["normalize_embeddings"] is True and kw["convert_to_numpy"] is True


@patch(SEAM)
def test_generate_embeddings_many_equals_per_file_calls(st):
    st.return_value = _hash_model()
    g = EmbeddingGenerator(EmbeddingConfig(batch_size=2, show_progress=False, embeddings_as_arrays=True))
    files = [[Chunk(f"a{i}", f"alpha {i}", {}) for i in range(5)], [], [Chunk("b0", None, {}), Chunk("b1", "   ", {})]]
    many = g.generate_embeddings_many(files, flush_at=4)
    assert [m.shape for m in many] == [(5, 5), (0, 5), (2, 5)]
    for chunks, rows in zip(files, many):
        if chunks:
            ref = g.generate_embeddings([Chunk(c.id, c.text, {}) for c in chunks])   # same sanitising per text
            assert np.array_equal(rows, ref)
            assert all(isinstance(c.embedding, np.ndarray) for c in chunks)


def test_batch_size_policy_matches_the_committed_table():
    """calculate_optimal_batch_size against a table computed from the reference's formula in exact
    arithmetic (tests/golden/make_policy_goldens.py; src/gpu_utils.py:169-192), incl. the edges around 1 GB free and
    the cap of 256 (64 for "mps")."""
    import json
    from pathlib import Path

    from claude_semantic_search_amd.gpu_utils import calculate_optimal_batch_size

    g = json.loads((Path(__file__).resolve().parent / "golden" / "batch_size_table.json").read_text())
    assert len(g["rows"]) >= 50
    for r in g["rows"]:
        got = calculate_optimal_batch_size(r["free_gb"], embedding_dim=r["dim"], backend=r["backend"])
        assert got == r["batch"], r
    assert calculate_optimal_batch_size(2.0) == 256 and calculate_optimal_batch_size(0.5) == 8
