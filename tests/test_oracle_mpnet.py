"""CPU: pins the encoder oracle (oracle/mpnet_oracle.py) against the in-container
transformers.MPNetModel (the architecture all-mpnet-base-v2 uses) on seeded
weights, and the host-side relative-position bucket helper of libcss_hip against
transformers' own function."""
import numpy as np
import pytest
import torch

from oracle import mpnet_oracle as mo

transformers = pytest.importorskip("transformers")


def _hf_model(cfg: mo.MpnetCfg, w):
    from transformers import MPNetConfig, MPNetModel

    hf = MPNetModel(MPNetConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers,
                                num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                                max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps,
                                relative_attention_num_buckets=cfg.rel_buckets, hidden_act="gelu",
                                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
                    add_pooling_layer=False).eval()
    missing, unexpected = hf.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    return hf


def test_oracle_matches_transformers_mpnet_padded_batch():
    cfg = mo.MpnetCfg(num_layers=2)
    w = mo.synth_weights(cfg, seed=7)
    hf = _hf_model(cfg, w)
    lengths = [40, 17, 2, 1]
    batch = mo.synth_batch(cfg, lengths, seed=11)
    Lmax = max(lengths)
    ids = torch.full((len(batch), Lmax), cfg.pad_id, dtype=torch.long)
    mask = torch.zeros((len(batch), Lmax), dtype=torch.long)
    for b, s in enumerate(batch):
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    with torch.no_grad():
        hs = hf(input_ids=ids, attention_mask=mask).last_hidden_state
    for b, s in enumerate(batch):
        with torch.no_grad():
            mine = mo.encode_tokens(w, cfg, s)
        assert torch.allclose(mine, hs[b, :len(s)], atol=5e-6), float((mine - hs[b, :len(s)]).abs().max())
    # sentence-transformers Pooling(mean) + Normalize on the HF output
    m = mask[:, :, None].float()
    pooled = (hs * m).sum(1) / m.sum(1).clamp(min=1e-9)
    ref = torch.nn.functional.normalize(pooled, p=2, dim=1).numpy()
    out = mo.encode(w, cfg, batch)
    assert np.allclose(out, ref, atol=1e-6)
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-6)


def test_oracle_matches_transformers_12_layers_full_length():
    """The full 12-layer architecture at the truncation length (384) and one below it: last hidden states and
    pooled embeddings of the oracle against transformers.MPNetModel on a padded batch with its attention mask."""
    torch.set_num_threads(8)
    cfg = mo.MpnetCfg()
    w = mo.synth_weights(cfg, seed=3)
    hf = _hf_model(cfg, w)
    lengths = [384, 383, 129]
    batch = mo.synth_batch(cfg, lengths, seed=21)
    ids = torch.full((3, 384), cfg.pad_id, dtype=torch.long)
    mask = torch.zeros((3, 384), dtype=torch.long)
    for b, s in enumerate(batch):
        ids[b, :len(s)] = torch.tensor(s)
        mask[b, :len(s)] = 1
    with torch.no_grad():
        hs = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        for b, s in enumerate(batch):
            mine = mo.encode_tokens(w, cfg, s)
            assert torch.allclose(mine, hs[b, :len(s)], atol=3e-5), float((mine - hs[b, :len(s)]).abs().max())
    m = mask[:, :, None].float()
    ref = torch.nn.functional.normalize((hs * m).sum(1) / m.sum(1).clamp(min=1e-9), p=2, dim=1).numpy()
    assert np.abs(mo.encode(w, cfg, batch) - ref).max() < 2e-6


def test_batched_padded_oracle_equals_the_per_sequence_oracle():
    cfg = mo.MpnetCfg(num_layers=2)
    w = mo.synth_weights(cfg, seed=9)
    batch = mo.synth_batch(cfg, [33, 5, 64, 1, 17, 40, 2], seed=3)
    assert np.abs(mo.encode_batched(w, cfg, batch, batch_size=3) - mo.encode(w, cfg, batch)).max() < 2e-6


def test_rel_bucket_table_spot_values_and_library_helper():
    from transformers.models.mpnet import modeling_mpnet as mm
    from claude_semantic_search_amd import _native as nat

    rel = torch.arange(-511, 512)
    hf = mm.MPNetEncoder.relative_position_bucket(rel.view(1, -1), num_buckets=32)[0]
    assert torch.equal(mo.relative_position_bucket(rel, 32), hf)
    # SURVEY.md 8(c) G6 spot values
    assert hf[511 - 9:511 + 10].tolist() == [8, 8, 7, 6, 5, 4, 3, 2, 1, 0, 17, 18, 19, 20, 21, 22, 23, 24, 24]
    assert int(hf[511 - 380]) == 15 and int(hf[511 + 380]) == 31
    lib = nat.lib()
    mine = [lib.css_mpnet_rel_bucket(int(r), 32, 128) for r in rel.tolist()]
    assert mine == hf.tolist()


def test_synth_weights_spec():
    cfg = mo.MpnetCfg(num_layers=1)
    w = mo.synth_weights(cfg, seed=3)
    assert w["embeddings.word_embeddings.weight"].shape == (30527, 768)
    assert float(w["embeddings.word_embeddings.weight"][cfg.pad_id].abs().max()) == 0.0
    assert float(w["embeddings.position_embeddings.weight"][cfg.pad_id].abs().max()) == 0.0
    assert abs(float(w["encoder.layer.0.intermediate.dense.weight"].std()) - 0.02) < 5e-4
    assert abs(float(w["encoder.layer.0.attention.LayerNorm.weight"].mean()) - 1.0) < 2e-2
    n = sum(v.numel() for v in w.values())
    assert n == 30527 * 768 + 514 * 768 + 2 * 768 + 32 * 12 + (4 * (768 * 768 + 768) + 2 * 768 * 3072 + 3072 + 768 + 4 * 768)


def test_committed_bucket_table_and_goldens_reproduce():
    from pathlib import Path

    gold = Path(__file__).resolve().parent / "golden"
    tab = np.load(gold / "rel_bucket_table.npy")
    rel = torch.arange(-511, 512)
    assert np.array_equal(tab, mo.relative_position_bucket(rel, 32).numpy().astype(np.int8))
    g = np.load(gold / "encoder_2layer.npz")
    cfg = mo.MpnetCfg(num_layers=2)
    lengths = g["lengths"].tolist()[:3]  # the short ones keep the CPU suite fast
    out = mo.encode(mo.synth_weights(cfg, int(g["wseed"])), cfg, mo.synth_batch(cfg, g["lengths"].tolist(), seed=int(g["bseed"]))[:3])
    # the committed embeddings come from transformers.MPNetModel (make_encoder_goldens.py), not from the oracle
    assert "transformers" in str(g["source"]) and float(g["oracle_maxdiff"]) < 2e-6
    assert np.abs(out - g["emb"][:3]).max() < 2e-6
    g12 = np.load(gold / "encoder_12layer.npz")
    assert "transformers" in str(g12["source"]) and float(g12["oracle_maxdiff"]) < 2e-6
    cfg12 = mo.MpnetCfg()
    b12 = mo.synth_batch(cfg12, g12["lengths"].tolist(), seed=int(g12["bseed"]))
    pick = [0, 4, 6]  # the short ones (8, 17 and 1 tokens) keep the CPU suite fast
    out12 = mo.encode(mo.synth_weights(cfg12, int(g12["wseed"])), cfg12, [b12[i] for i in pick])
    assert np.abs(out12 - g12["emb"][pick]).max() < 2e-6
