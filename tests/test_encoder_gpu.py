"""GPU parity tests: HIP MPNet encoder (through the C ABI) vs the CPU oracle.

Tolerances: fp32 verification mode <= 1e-4 max-abs on unit vectors (kernel-level
parity); bf16 MFMA product mode: per-row cosine >= 1 - 1e-3."""
import numpy as np
import pytest

from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder

pytestmark = pytest.mark.gpu


def _oracle(cfg_layers, lengths, wseed, bseed):
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=cfg_layers)
    batch = mo.synth_batch(cfg, lengths, seed=bseed)
    return cfg, batch, mo.encode(mo.synth_weights(cfg, wseed), cfg, batch)


def test_synthetic_weights_identical_on_device_and_host():
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=1)
    w = mo.synth_weights(cfg, 5)
    enc = MpnetEncoder(synthetic_seed=5, compute="fp32", cfg_overrides={"num_layers": 1})
    for name in ("embeddings.word_embeddings.weight", "embeddings.LayerNorm.weight",
                 "encoder.relative_attention_bias.weight", "encoder.layer.0.attention.attn.k.weight",
                 "encoder.layer.0.attention.attn.v.bias", "encoder.layer.0.output.dense.weight",
                 "encoder.layer.0.output.LayerNorm.bias"):
        got = enc.export_weight(name, tuple(w[name].shape))
        assert np.array_equal(got, w[name].numpy()), name


@pytest.mark.parametrize("lengths", [[1], [2, 7, 31], [128, 5, 64, 33], [383, 384, 2]])
def test_fp32_mode_matches_oracle_2_layers(lengths):
    cfg, batch, ref = _oracle(2, lengths, 7, 11)
    enc = MpnetEncoder(synthetic_seed=7, compute="fp32", cfg_overrides={"num_layers": 2})
    out = enc.encode_ids(batch)
    assert np.abs(out - ref).max() < 1e-4, np.abs(out - ref).max()
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize("lengths", [[1], [2, 7, 31], [128, 5, 64, 33], [383, 384, 2], [130, 129, 127, 65, 63]])
def test_bf16_mode_matches_oracle_2_layers(lengths):
    cfg, batch, ref = _oracle(2, lengths, 7, 11)
    enc = MpnetEncoder(synthetic_seed=7, compute="bf16", cfg_overrides={"num_layers": 2})
    out = enc.encode_ids(batch)
    cos = (out * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    assert np.abs(out - ref).max() < 2e-2


def test_full_12_layer_bf16_and_fp32_vs_oracle():
    cfg, batch, ref = _oracle(12, [8, 40, 100, 384, 17, 250], 3, 4)
    for mode, tol in (("fp32", 3e-4), ("bf16", None)):
        enc = MpnetEncoder(synthetic_seed=3, compute=mode)
        out = enc.encode_ids(batch)
        cos = (out * ref).sum(1)
        assert cos.min() > 1 - 1e-3, (mode, cos)
        if tol:
            assert np.abs(out - ref).max() < tol, (mode, np.abs(out - ref).max())
        enc.close()


def test_unnormalized_pooling_and_batch_independence():
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=2)
    w = mo.synth_weights(cfg, 7)
    batch = mo.synth_batch(cfg, [9, 70, 200], seed=2)
    enc = MpnetEncoder(synthetic_seed=7, compute="fp32", cfg_overrides={"num_layers": 2})
    raw = enc.encode_ids(batch, normalize=False)
    ref = mo.encode(w, cfg, batch, normalize=False)
    assert np.abs(raw - ref).max() < 2e-4
    alone = np.concatenate([enc.encode_ids([s]) for s in batch])
    together = enc.encode_ids(batch)
    assert np.abs(alone - together).max() < 1e-6  # packing must not leak across sequences


def test_sentence_transformer_surface_and_order_restoring():
    enc = MpnetEncoder(synthetic_seed=1, compute="bf16", cfg_overrides={"num_layers": 2})
    assert enc.get_sentence_embedding_dimension() == 768 and enc.to("cuda") is enc and "cuda" in enc.device
    enc.max_seq_length = 384
    texts = ["short", "a much longer sentence about python error handling with try except blocks " * 3,
             "medium length text here", ""]
    single = enc.encode(texts[0], normalize_embeddings=True, show_progress_bar=False)
    assert single.shape == (768,) and single.dtype == np.float32
    many = enc.encode(texts, batch_size=2, normalize_embeddings=True, show_progress_bar=False, convert_to_numpy=True)
    assert many.shape == (4, 768)
    assert np.allclose(many[0], single, atol=1e-6)
    assert np.allclose(np.linalg.norm(many, axis=1), 1.0, atol=1e-4)
    rev = enc.encode(texts[::-1], batch_size=3)
    assert np.allclose(rev[::-1], many, atol=1e-6)


def test_invalid_inputs_fail_loudly():
    enc = MpnetEncoder(synthetic_seed=1, compute="fp32", cfg_overrides={"num_layers": 1})
    with pytest.raises(RuntimeError):
        enc.encode_ids([[0, 5, 1, 2]])          # pad id inside a sequence
    with pytest.raises(RuntimeError):
        enc.encode_ids([[0] + [5] * 400 + [2]])  # longer than max_seq_len
    with pytest.raises(FileNotFoundError):
        MpnetEncoder("all-mpnet-base-v2")        # no weights offline: never silently synthetic


def test_committed_goldens_2layer_and_probes():
    """HIP encoder vs the committed oracle goldens (tests/golden/encoder_2layer.npz): pooled
    embeddings and the layer-0 probes (embedding LN, attention block output, layer output)."""
    from pathlib import Path

    from oracle import mpnet_oracle as mo

    g = np.load(Path(__file__).resolve().parent / "golden" / "encoder_2layer.npz")
    lengths = g["lengths"].tolist()
    cfg = mo.MpnetCfg(num_layers=2)
    batch = mo.synth_batch(cfg, lengths, seed=int(g["bseed"]))
    for mode, tol, ptol in (("fp32", 1e-4, 2e-4), ("bf16", 2e-2, 6e-2)):
        enc = MpnetEncoder(synthetic_seed=int(g["wseed"]), compute=mode, cfg_overrides={"num_layers": 2})
        out = enc.encode_ids(batch)
        assert np.abs(out - g["emb"]).max() < tol and ((out * g["emb"]).sum(1)).min() > 1 - 1e-3
        enc.close()
        enc1 = MpnetEncoder(synthetic_seed=int(g["wseed"]), compute=mode, cfg_overrides={"num_layers": 1})
        enc1.encode_ids(batch)
        T = sum(lengths)
        x = enc1.debug_read("x32", (T, 768))  # layer-0 output = final hidden state of the 1-layer model
        off = np.cumsum([0] + lengths)
        got = np.stack([np.stack([x[off[i], :8], x[off[i + 1] - 1, :8]]) for i in range(len(lengths))])
        assert np.abs(got - g["ffn_out"]).max() < ptol, (mode, np.abs(got - g["ffn_out"]).max())
        enc1.close()


def test_committed_goldens_12layer():
    from pathlib import Path

    from oracle import mpnet_oracle as mo

    g = np.load(Path(__file__).resolve().parent / "golden" / "encoder_12layer.npz")
    cfg = mo.MpnetCfg()
    batch = mo.synth_batch(cfg, g["lengths"].tolist(), seed=int(g["bseed"]))
    enc = MpnetEncoder(synthetic_seed=int(g["wseed"]), compute="bf16")
    out = enc.encode_ids(batch)
    cos = (out * g["emb"]).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    enc.close()


def test_single_query_graph_replay_is_bit_stable():
    """Tiny batches are replayed from a hipGraph from their third use on (eager, capture, replay):
    every call must return the same bits, also after an interleaved call of another shape."""
    enc = MpnetEncoder(synthetic_seed=5, compute="bf16", cfg_overrides={"num_layers": 3})
    a = [0, 17, 923, 4055, 12000, 2]
    b = [0, 99, 2]
    first = enc.encode_ids([a])
    for i in range(6):
        again = enc.encode_ids([a])
        assert np.array_equal(first, again), i
        if i == 2:
            other = enc.encode_ids([b])
            assert np.array_equal(other, enc.encode_ids([b]))
    both = enc.encode_ids([a, b])
    assert np.allclose(both[0], first[0], atol=1e-6) and np.array_equal(both, enc.encode_ids([a, b]))
    big = enc.encode_ids([list(a) * 60])  # 360 tokens: grows the activation buffers, graphs are dropped
    assert np.array_equal(first, enc.encode_ids([a])) and big.shape == (1, 768)
