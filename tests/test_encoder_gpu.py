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


def test_bf16_folded_layernorm_path_matches_oracle_and_is_bit_reproducible():
    """Batches of >= 1024 tokens take the path with LayerNorm folded into the GEMM epilogues (no LayerNorm
    kernels: folded weights + row statistics, css_encoder_kernels.h).  Ragged lengths (row tiles cut sequences,
    the last tile is partial), 2 layers, against the oracle; the row statistics are accumulated with integer
    atomics, so two runs must agree bit for bit; the same sequences through the small-batch path (separate
    LayerNorm kernels) must agree to bf16 rounding noise."""
    lengths = [384, 1, 200, 383, 77, 129, 300, 256, 2, 45]          # 1777 tokens
    cfg, batch, ref = _oracle(2, lengths, 7, 13)
    enc = MpnetEncoder(synthetic_seed=7, compute="bf16", cfg_overrides={"num_layers": 2})
    out = enc.encode_ids(batch)
    cos = (out * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    assert np.abs(out - ref).max() < 2e-2, np.abs(out - ref).max()
    again = enc.encode_ids(batch)
    assert np.array_equal(out, again)
    with pytest.raises(RuntimeError):
        enc.debug_read("x32", (sum(lengths), 768))       # not materialised on this path: loud, not stale
    small = np.concatenate([enc.encode_ids([s]) for s in batch])
    assert ((small * out).sum(1)).min() > 1 - 5e-5 and np.abs(small - out).max() < 5e-3
    raw = enc.encode_ids(batch, normalize=False)
    from oracle import mpnet_oracle as mo

    ref_raw = mo.encode(mo.synth_weights(cfg, 7), cfg, batch, normalize=False)
    assert np.abs(raw - ref_raw).max() < 3e-2 * np.abs(ref_raw).max()
    enc.close()


def test_bf16_folded_path_many_short_sequences():
    """The folded path at the other extreme of its shape range: 700 sequences of 1..4 tokens (1024+ token rows, far
    more sequences than row tiles; pooling partials for 700 sequences; attention blocks of a handful of keys)."""
    rng = np.random.default_rng(5)
    lengths = rng.integers(1, 5, size=700).tolist()
    assert sum(lengths) >= 1024
    cfg, batch, ref = _oracle(2, lengths, 7, 17)
    enc = MpnetEncoder(synthetic_seed=7, compute="bf16", cfg_overrides={"num_layers": 2})
    out = enc.encode_ids(batch)
    cos = (out * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos.min()
    assert np.abs(out - ref).max() < 2e-2
    assert np.array_equal(out, enc.encode_ids(batch))
    enc.close()


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
    """12 layers, lengths up to the truncation limit, against embeddings computed by transformers.MPNetModel
    (tests/golden/make_encoder_goldens.py), in both compute modes."""
    from pathlib import Path

    from oracle import mpnet_oracle as mo

    g = np.load(Path(__file__).resolve().parent / "golden" / "encoder_12layer.npz")
    assert "transformers" in str(g["source"])
    cfg = mo.MpnetCfg()
    batch = mo.synth_batch(cfg, g["lengths"].tolist(), seed=int(g["bseed"]))
    for mode, tol in (("bf16", None), ("fp32", 3e-4)):
        enc = MpnetEncoder(synthetic_seed=int(g["wseed"]), compute=mode)
        out = enc.encode_ids(batch)
        cos = (out * g["emb"]).sum(1)
        assert cos.min() > 1 - 1e-3, (mode, cos)
        if tol:
            assert np.abs(out - g["emb"]).max() < tol, (mode, np.abs(out - g["emb"]).max())
        enc.close()


def test_batch_256x384_product_shape_against_the_oracle():
    """BASELINE configs[2] at full size: 256 sequences of 384 tokens, 12 layers, bf16 -- the inputs bench.py
    encodes (token seed 7, weights seed 1).  The persistent GEMM walks 384 row tiles, attention runs 256 x 3 x 12
    blocks, pooling reduces 98 304 token rows: 16 sequences spread over the batch (first, last, tile boundaries)
    are checked against the CPU oracle, and the same sequences encoded alone must give the same embeddings."""
    import torch

    from claude_semantic_search_amd import synth
    from oracle import mpnet_oracle as mo

    B, L = 256, 384
    cfg = mo.MpnetCfg()
    ids = synth.uint(7, np.arange(B * L, dtype=np.uint64), 4, cfg.vocab).astype(np.int32)
    ids[0::L] = 0
    ids[L - 1::L] = 2
    batch = [ids[i * L:(i + 1) * L].tolist() for i in range(B)]
    enc = MpnetEncoder(synthetic_seed=1, compute="bf16")
    out = enc.encode_ids(batch)
    assert out.shape == (B, 768) and np.isfinite(out).all()
    assert np.abs(np.linalg.norm(out, axis=1) - 1.0).max() < 1e-4
    pick = [0, 1, 2, 17, 63, 64, 100, 127, 128, 170, 191, 192, 230, 253, 254, 255]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref = mo.encode(mo.synth_weights(cfg, 1), cfg, [batch[i] for i in pick])
    cos = (out[pick] * ref).sum(1)
    assert cos.min() > 1 - 1e-3, cos
    assert np.abs(out[pick] - ref).max() < 2e-2
    # batch composition must not matter: the same rows through a 16-sequence batch (other tile walk, same kernels)
    # and one sequence alone (the 128x128-tile GEMM configuration: another summation order, bf16 rounding noise only)
    small = enc.encode_ids([batch[i] for i in pick])
    assert ((small * out[pick]).sum(1)).min() > 1 - 1e-5 and np.abs(small - out[pick]).max() < 5e-3
    alone = enc.encode_ids([batch[255]])
    assert float((alone[0] * out[255]).sum()) > 1 - 1e-5
    # different sequences give different embeddings (a stuck tile would repeat rows)
    assert len({tuple(np.round(r[:8], 4)) for r in out}) == B
    enc.close()


def _write_checkpoint(dirpath, cfg, w, prefix="", with_pooler=True, subdir=False):
    """A local model directory in the layout sentence-transformers / transformers write: config.json +
    model.safetensors with the HF key names (q / k / v unfused), optionally under 0_Transformer/."""
    import json

    from safetensors.numpy import save_file

    root = dirpath / "0_Transformer" if subdir else dirpath
    root.mkdir(parents=True, exist_ok=True)
    (root / "config.json").write_text(json.dumps({
        "model_type": "mpnet", "num_hidden_layers": cfg.num_layers, "hidden_size": cfg.hidden,
        "num_attention_heads": cfg.heads, "intermediate_size": cfg.ffn, "vocab_size": cfg.vocab,
        "max_position_embeddings": cfg.max_pos, "relative_attention_num_buckets": cfg.rel_buckets,
        "pad_token_id": cfg.pad_id, "layer_norm_eps": cfg.ln_eps}))
    sd = {prefix + k: v.numpy() for k, v in w.items()}
    if with_pooler:   # present in real checkpoints, unused by sentence-transformers' Pooling(mean)
        sd[prefix + "pooler.dense.weight"] = np.zeros((cfg.hidden, cfg.hidden), np.float32)
        sd[prefix + "pooler.dense.bias"] = np.zeros((cfg.hidden,), np.float32)
        sd[prefix + "embeddings.position_ids"] = np.arange(cfg.max_pos, dtype=np.float32)[None]
    save_file(sd, str(root / "model.safetensors"))
    return sd


@pytest.mark.parametrize("prefix,subdir", [("", False), ("mpnet.", False), ("0.auto_model.", True)])
def test_checkpoint_directory_loads_like_the_synthetic_init(tmp_path, prefix, subdir):
    """The real-weight path (src/embeddings.py:86-88 of the reference: SentenceTransformer(name, cache_folder)):
    directory discovery, safetensors read, prefix stripping, unfused q/k/v -> fused [2304, 768], pooler skipped.
    The same seeded tensors written as a checkpoint and loaded through MpnetEncoder(path) must give the device
    weights and the embeddings of css_encoder_init_synthetic bit for bit, and match the transformers golden."""
    from pathlib import Path

    from oracle import mpnet_oracle as mo

    g = np.load(Path(__file__).resolve().parent / "golden" / "encoder_2layer.npz")
    cfg = mo.MpnetCfg(num_layers=2)
    w = mo.synth_weights(cfg, int(g["wseed"]))
    _write_checkpoint(tmp_path / "all-mpnet-base-v2", cfg, w, prefix=prefix, subdir=subdir)
    batch = mo.synth_batch(cfg, g["lengths"].tolist(), seed=int(g["bseed"]))
    for mode, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        # by name + cache_folder, as the reference constructs it
        enc = MpnetEncoder("all-mpnet-base-v2", cache_folder=str(tmp_path), compute=mode)
        assert enc.cfg["num_layers"] == 2 and enc.get_sentence_embedding_dimension() == 768
        syn = MpnetEncoder(synthetic_seed=int(g["wseed"]), compute=mode, cfg_overrides={"num_layers": 2})
        for name, shape in (("embeddings.word_embeddings.weight", (cfg.vocab, 768)),
                            ("encoder.layer.1.attention.attn.qkv.weight", (2304, 768)),
                            ("encoder.layer.0.attention.attn.qkv.bias", (2304,)),
                            ("encoder.layer.1.attention.attn.v.weight", (768, 768)),
                            ("encoder.layer.0.output.dense.weight", (768, 3072)),
                            ("encoder.relative_attention_bias.weight", (32, 12))):
            assert np.array_equal(enc.export_weight(name, shape), syn.export_weight(name, shape)), name
        out, ref = enc.encode_ids(batch), syn.encode_ids(batch)
        assert np.array_equal(out, ref)
        assert np.abs(out - g["emb"]).max() < tol and ((out * g["emb"]).sum(1)).min() > 1 - 1e-3
        enc.close()
        syn.close()


def test_real_weights_without_vocabulary_refuse_text_and_root_level_vocab_is_used(tmp_path):
    """a checkpoint directory without vocab.txt used to get the CRC32 HashTokenizer silently --
    plausible-looking garbage embeddings.  Now: ids still encode, text raises, get_model_info() says why; with the
    tokenizer files at the ROOT of a 0_Transformer/ layout (modern sentence-transformers) text works and honours
    tokenizer_config.json's do_lower_case."""
    import json

    from oracle import mpnet_oracle as mo
    from claude_semantic_search_amd.embeddings import EmbeddingConfig, EmbeddingGenerator

    cfg = mo.MpnetCfg(num_layers=1)
    w = mo.synth_weights(cfg, 3)
    root = tmp_path / "all-mpnet-base-v2"
    _write_checkpoint(root, cfg, w, prefix="0.auto_model.", subdir=True)
    enc = MpnetEncoder("all-mpnet-base-v2", cache_folder=str(tmp_path), compute="fp32")
    assert enc.tokenizer is None and "vocab.txt" in enc.tokenizer_problem
    assert enc.encode_ids([[0, 9, 77, 2]]).shape == (1, 768)
    with pytest.raises(RuntimeError, match="vocab.txt"):
        enc.encode("fix the python error")
    enc.close()
    gen = EmbeddingGenerator(EmbeddingConfig(model_name="all-mpnet-base-v2", cache_dir=str(tmp_path), use_gpu=True))
    gen.load_model()
    assert "vocab.txt" in gen.get_model_info()["tokenizer_problem"]
    # tokenizer files at the root, weights in 0_Transformer/
    pieces = ["<s>", "<pad>", "</s>", "<unk>", "fix", "the", "python", "error", "Fix", "##s"]
    (root / "vocab.txt").write_text("\n".join(pieces) + "\n")
    (root / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": False}))
    (root / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 128}))
    enc = MpnetEncoder("all-mpnet-base-v2", cache_folder=str(tmp_path), compute="fp32")
    assert enc.tokenizer_problem is None and enc.max_seq_length == 128
    assert [list(map(int, t)) for t in enc.tokenize(["Fix the python errors", "fix"])] == [[0, 8, 5, 6, 7, 9, 2], [0, 4, 2]]   # case kept
    a = enc.encode(["Fix the python errors"])
    ref = mo.encode(w, cfg, [[0, 8, 5, 6, 7, 9, 2]])
    assert np.abs(a - ref).max() < 1e-4
    enc.close()
    (root / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": True}))
    enc = MpnetEncoder("all-mpnet-base-v2", cache_folder=str(tmp_path), compute="fp32")
    assert [list(map(int, t)) for t in enc.tokenize(["Fix"])] == [[0, 4, 2]]
    enc.close()


def test_incomplete_or_ambiguous_checkpoints_are_rejected(tmp_path):
    from oracle import mpnet_oracle as mo
    from claude_semantic_search_amd._native import CssError

    cfg = mo.MpnetCfg(num_layers=1)
    w = mo.synth_weights(cfg, 2)
    sd = {k: v.numpy() for k, v in w.items()}
    enc = MpnetEncoder(synthetic_seed=2, compute="fp32", cfg_overrides={"num_layers": 1})
    enc.load_state_dict(sd)                                   # complete: accepted
    for drop in ("encoder.layer.0.attention.attn.k.bias", "encoder.layer.0.output.LayerNorm.weight",
                 "embeddings.position_embeddings.weight"):
        part = {k: v for k, v in sd.items() if k != drop}
        with pytest.raises(CssError, match="cover|no tensor"):
            enc.load_state_dict(part)
    dup = dict(sd)
    dup["mpnet.encoder.layer.0.attention.attn.o.bias"] = sd["encoder.layer.0.attention.attn.o.bias"]
    with pytest.raises(CssError, match="more than once"):
        enc.load_state_dict(dup)
    bad = dict(sd)
    bad["encoder.layer.0.intermediate.dense.bias"] = np.zeros(7, np.float32)
    with pytest.raises(CssError, match="elements"):
        enc.load_state_dict(bad)
    with pytest.raises(CssError, match="unknown parameter"):
        enc.load_state_dict({**sd, "encoder.layer.5.output.dense.bias": np.zeros(768, np.float32)})
    enc.load_state_dict(sd)                                   # still usable afterwards
    ref = mo.encode(w, cfg, [[0, 9, 77, 2]])
    assert np.abs(enc.encode_ids([[0, 9, 77, 2]]) - ref).max() < 1e-4
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


@pytest.mark.parametrize("qk_gain", [1.0, 10.0])
def test_attention_reference_free_pass_and_its_guard(qk_gain):
    """k_attention_bf16 forms softmax rows as exp2(score) / sum without a running maximum while every row sum of
    a block stays inside (2^-100, 2^100), and repeats the block with the running-maximum pass otherwise.  Both
    passes must give the same embeddings: (a) ordinary logits -- the default takes the fast pass, range = 0 forces
    the guarded one; (b) q / k weights scaled until logits reach the hundreds, where exp2(score) overflows fp32 and
    the DEFAULT setting has to fall back by itself (no inf / NaN, same result as the forced pass, close to the
    fp32 oracle as far as bf16 logits of that size allow)."""
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=2)
    w = mo.synth_weights(cfg, 21)
    if qk_gain != 1.0:
        for li in range(cfg.num_layers):
            for nm in ("q", "k"):
                w[f"encoder.layer.{li}.attention.attn.{nm}.weight"] = w[f"encoder.layer.{li}.attention.attn.{nm}.weight"] * qk_gain
    lengths = [384, 129, 64, 33, 300, 1, 383, 200]    # 1494 tokens: the folded path; + a small batch below
    batch = mo.synth_batch(cfg, lengths, seed=4)
    ref = mo.encode(w, cfg, batch)
    enc = MpnetEncoder(synthetic_seed=21, compute="bf16", cfg_overrides={"num_layers": 2})
    enc.load_state_dict({k: v.numpy() for k, v in w.items()})
    for sub in (batch, batch[1:4]):
        enc.set_attention_range(2.0 ** 100)
        fast = enc.encode_ids(sub)
        enc.set_attention_range(0.0)
        safe = enc.encode_ids(sub)
        assert np.isfinite(fast).all() and np.isfinite(safe).all()
        assert np.abs(fast - safe).max() < 2e-3, np.abs(fast - safe).max()
        r = ref if len(sub) == len(batch) else ref[1:4]
        cos = (safe * r).sum(1)
        assert cos.min() > (1 - 1e-3 if qk_gain == 1.0 else 0.95), cos
    if qk_gain != 1.0:   # the logits really are outside the fast pass's range
        qkv = enc.debug_read("qkv", (sum(lengths[1:4]), 3 * 768))
        q0, k0 = qkv[:129, :64], qkv[:129, 768:768 + 64]      # sequence of 129 tokens, head 0 (q pre-scaled to log2 units)
        assert np.abs(q0 @ k0.T).max() > 110
    enc.close()


def test_trained_like_outlier_statistics_keep_the_bf16_paths_inside_the_tolerance():
    """every other encoder test draws its weights from N(0, 0.02^2).  A trained MPNet has outlier
    channels (a few hidden dimensions with LayerNorm gains and activations tens of times the others'), peaked softmax
    rows and FFN units deep inside GELU's tails -- the regime the bf16 residual stream with fixed-point row statistics
    (the LayerNorm-folded path of batches >= 1024 tokens) is most exposed to.  ``mpnet_oracle.trained_like_weights``
    builds such weights; 256 chunks of mixed lengths go through the folded path and (in slices) through the small-batch
    bf16 path, and both are held against the fp32 oracle: cosine >= 1 - 1e-3 per row AND pairwise-score drift <= 1e-3
    over the whole 256 x 256 score matrix -- the north-star's tolerance on what a search actually ranks by.
    Measured (tools/enc_drift_probe.py, one MI355X, 12 layers, drift = max over the 65 536 pairs): default gains 4 / 4 /
    1.5 (outlier channels ~33 x the mean activation, logits up to ~30): folded path 5.4e-4, small-batch path 5.6e-4;
    gains 6 / 6 / 1.5 (outliers ~64 x): 1.5e-3 / 2.1e-3 -- there the bf16 MFMA operands themselves exceed 1e-3, the
    fp32-residual small-batch path no less than the bf16-residual folded one; plain N(0, 0.02^2) weights: 2e-5 / 8e-6."""
    from claude_semantic_search_amd import synth
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=12)
    w = mo.trained_like_weights(cfg, 33)
    chars = 100 + synth.uint(5, np.arange(256, dtype=np.uint64), 0, 1901)
    lengths = np.clip(np.round(chars / 10).astype(np.int64) + 2, 2, 384).tolist()      # mean ~ 107 tokens, 27 k tokens in all
    lengths[0], lengths[1], lengths[2] = 384, 1, 383
    batch = mo.synth_batch(cfg, lengths, seed=6)
    ref = mo.encode_batched(w, cfg, batch, batch_size=16)
    # the outlier channels really dominate the oracle's residual stream (otherwise this test is the ordinary one)
    probes = {}
    with torch_no_grad():
        mo.encode_tokens(w, cfg, batch[3], probes=probes)
    a = probes["attn_out"].abs()
    assert float(a[:, [7, 77, 300]].mean()) > 10 * float(a.mean())
    enc = MpnetEncoder(synthetic_seed=33, compute="bf16")
    enc.load_state_dict({k: v.numpy() for k, v in w.items()})
    folded = enc.encode_ids(batch)                                         # one call: the LayerNorm-folded path
    small = np.concatenate([enc.encode_ids(batch[i:i + 4]) for i in range(0, 64, 4)])   # <= 1024-token calls: the other bf16 path
    enc.close()
    for name, got, r in (("folded", folded, ref), ("small batches", small, ref[:64])):
        assert np.isfinite(got).all(), name
        cos = (got * r).sum(1)
        assert cos.min() > 1 - 1e-3, (name, float(cos.min()))
        drift = np.abs(got @ got.T - r @ r.T).max()
        assert drift <= 1e-3, (name, float(drift), float(cos.min()))


def test_huge_lone_logits_do_not_poison_the_running_maximum_pass():
    """Found by the test above with extreme gains: a one-token sequence whose only attention logit is hugely negative
    (-1e4 in the log2 domain) takes the running-maximum pass (exp2 underflows in the fast pass), whose first-tile
    rescale multiplied the still-zero sums by exp2(+1e4) = inf -> NaN for the whole row.  With attention logits of
    +-7000 (gains 30 / 20 / 6) every output must be finite, and rows whose softmax is trivially exact -- one-token
    sequences: the single probability is 1 whatever the logit -- must still match the fp32 oracle."""
    from oracle import mpnet_oracle as mo

    cfg = mo.MpnetCfg(num_layers=3)
    w = mo.trained_like_weights(cfg, 34, gamma_gain=30.0, emb_gain=20.0, logit_gain=6.0)
    lengths = [1] * 40 + [384, 200, 77, 384, 300]             # 1385 tokens: the folded path; the ones alone: the small path
    batch = mo.synth_batch(cfg, lengths, seed=8)
    ref = mo.encode(w, cfg, batch[:40])
    enc = MpnetEncoder(synthetic_seed=34, compute="bf16", cfg_overrides={"num_layers": 3})
    enc.load_state_dict({k: v.numpy() for k, v in w.items()})
    folded = enc.encode_ids(batch)
    small = enc.encode_ids(batch[:40])
    enc.close()
    assert np.isfinite(folded).all() and np.isfinite(small).all()
    assert (folded[:40] * ref).sum(1).min() > 1 - 1e-3 and (small * ref).sum(1).min() > 1 - 1e-3


def torch_no_grad():
    import torch

    return torch.no_grad()


def test_real_checkpoint_directory_matches_transformers_when_one_is_supplied():
    """Opt-in (``CSS_REAL_MODEL_DIR`` = a local all-mpnet-base-v2 directory in sentence-transformers / HF layout; skipped
    when unset: no checkpoint exists offline): the user's own weights and vocabulary through ``MpnetEncoder`` -- loader,
    WordPiece front end, HIP encoder -- against ``transformers.AutoTokenizer`` / ``AutoModel`` + mean pooling + normalise,
    the pipeline ``SentenceTransformer.encode(..., normalize_embeddings=True)`` runs (``src/embeddings.py:86-88, :216-222``).
    Nothing from the reference or from the checkpoint is copied anywhere."""
    import os

    d = os.environ.get("CSS_REAL_MODEL_DIR")
    if not d:
        pytest.skip("CSS_REAL_MODEL_DIR is not set (no all-mpnet-base-v2 checkpoint offline)")
    import torch
    import torch.nn.functional as F
    from transformers import AutoModel, AutoTokenizer

    texts = ["How do I configure the semantic search index?", "def add(a, b):\n    return a + b", "Résumé of the café meeting — naïve façade.",
             "x", "The quick brown fox jumps over the lazy dog. " * 40]
    sub = os.path.join(d, "0_Transformer") if os.path.isdir(os.path.join(d, "0_Transformer")) else d
    tok = AutoTokenizer.from_pretrained(sub)
    model = AutoModel.from_pretrained(sub).eval()
    with torch.no_grad():
        b = tok(texts, padding=True, truncation=True, max_length=384, return_tensors="pt")
        h = model(**b).last_hidden_state
        m = b["attention_mask"][:, :, None].float()
        ref = F.normalize((h * m).sum(1) / m.sum(1).clamp(min=1e-9), p=2, dim=1).numpy()
    for compute, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        enc = MpnetEncoder(d, compute=compute)
        ids = enc.tokenize(texts)
        want = [[t for t, a in zip(row, att) if a] for row, att in zip(b["input_ids"].tolist(), b["attention_mask"].tolist())]
        assert [list(map(int, r)) for r in ids] == want                   # the tokenizer front end, id for id
        got = enc.encode(texts, batch_size=8, normalize_embeddings=True)
        enc.close()
        assert np.abs(got - ref).max() < tol and (got * ref).sum(1).min() > 1 - 1e-3, compute


def test_four_wave_gemm_variant_keeps_the_folded_path_bit_reproducible_and_inside_the_tolerance():
    """k_gemm4w (css_encoder_kernels.h: the LayerNorm-folded GEMMs on four 128 x 128 waves per block, accumulators in
    AGPRs, hand-placed instruction stream) is opt-in -- it measured slower end to end than k_gemm8p -- and
    stays under test: the folded-path parity test once more in a child process with all four GEMMs on it
    (CSS_GEMM_4W is read once per process)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, CSS_GEMM_4W="15")
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_encoder_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", "folded_layernorm_path_matches or trained_like_outlier"],
                       cwd=str(root), env=env, capture_output=True, text=True, timeout=600)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, f"folded-path tests under CSS_GEMM_4W=15 failed:\n{tail}\n{r.stderr[-2000:]}"
    assert " passed" in tail
