"""Generates tests/golden/encoder_*.npz.

    python tests/golden/make_encoder_goldens.py

The expected embeddings (`emb`) come from the in-container ``transformers.MPNetModel`` (5.15.0: the architecture
all-mpnet-base-v2 uses) with the seeded synthetic weights loaded through ``load_state_dict``, a padded batch with its
attention mask, then sentence-transformers' ``Pooling(mean)`` + ``Normalize`` restated on the HF output (SURVEY.md
App. A items 5, 7).  The CPU oracle (oracle/mpnet_oracle.py) is run next to it as a cross-check and must agree to
2e-6; its layer-0 probes are stored for the kernel-level tests.

G4 (encoder_2layer.npz): 2-layer config, B = 6, lengths {2, 7, 31, 128, 383, 384}, weights seed 7,
    token seed 11 -> pooled+normalised [6,768] and layer-0 probes (emb-LN, attention-block output,
    layer output: first 8 dims of the first and last token of every sequence).
G5 (encoder_12layer.npz): full 12-layer config, B = 10 mixed lengths incl. 383 / 384, weights seed 3,
    token seed 4 -> [10,768].
G6 (rel_bucket_table.npy): relative-position bucket for rel in [-511, 511] from transformers.
Inputs are regenerated from the seeds at test time; only outputs are stored.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import mpnet_oracle as mo  # noqa: E402

OUT = Path(__file__).resolve().parent


def hf_model(cfg: mo.MpnetCfg, w):
    from transformers import MPNetConfig, MPNetModel

    hf = MPNetModel(MPNetConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers,
                                num_attention_heads=cfg.heads, intermediate_size=cfg.ffn,
                                max_position_embeddings=cfg.max_pos, layer_norm_eps=cfg.ln_eps,
                                relative_attention_num_buckets=cfg.rel_buckets, hidden_act="gelu",
                                hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
                    add_pooling_layer=False).eval()
    missing, unexpected = hf.load_state_dict(w, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    return hf


def hf_encode(cfg: mo.MpnetCfg, w, batch, chunk=4) -> np.ndarray:
    """transformers forward on padded batches + Pooling(mean) + Normalize."""
    hf = hf_model(cfg, w)
    outs = []
    for c0 in range(0, len(batch), chunk):
        part = batch[c0:c0 + chunk]
        Lmax = max(len(s) for s in part)
        ids = torch.full((len(part), Lmax), cfg.pad_id, dtype=torch.long)
        mask = torch.zeros((len(part), Lmax), dtype=torch.long)
        for b, s in enumerate(part):
            ids[b, :len(s)] = torch.tensor(s)
            mask[b, :len(s)] = 1
        with torch.no_grad():
            hs = hf(input_ids=ids, attention_mask=mask).last_hidden_state
        m = mask[:, :, None].float()
        pooled = (hs * m).sum(1) / m.sum(1).clamp(min=1e-9)
        outs.append(torch.nn.functional.normalize(pooled, p=2, dim=1).numpy())
    return np.concatenate(outs).astype(np.float32)


def main():
    torch.set_num_threads(8)
    cfg = mo.MpnetCfg(num_layers=2)
    lengths = [2, 7, 31, 128, 383, 384]
    w = mo.synth_weights(cfg, 7)
    batch = mo.synth_batch(cfg, lengths, seed=11)
    emb = hf_encode(cfg, w, batch)
    emb_o = mo.encode(w, cfg, batch)
    d2 = float(np.abs(emb - emb_o).max())
    assert d2 < 2e-6, d2
    probes = {"emb_ln": [], "attn_out": [], "ffn_out": []}
    cfg1 = mo.MpnetCfg(num_layers=1)
    for ids in batch:
        pr = {}
        with torch.no_grad():
            mo.encode_tokens(w, cfg1, ids, probes=pr)
        for k in probes:
            t = pr[k]
            probes[k].append(np.stack([t[0, :8].numpy(), t[-1, :8].numpy()]))
    np.savez_compressed(OUT / "encoder_2layer.npz", lengths=np.array(lengths), wseed=7, bseed=11, emb=emb,
                        source="transformers.MPNetModel 5.15.0 + mean pooling + normalize", oracle_maxdiff=d2,
                        **{k: np.stack(v) for k, v in probes.items()})
    cfg12 = mo.MpnetCfg()
    lengths12 = [8, 40, 100, 384, 17, 250, 1, 64, 383, 129]
    w12 = mo.synth_weights(cfg12, 3)
    batch12 = mo.synth_batch(cfg12, lengths12, seed=4)
    emb12 = hf_encode(cfg12, w12, batch12)
    emb12_o = mo.encode(w12, cfg12, batch12)
    d12 = float(np.abs(emb12 - emb12_o).max())
    assert d12 < 2e-6, d12
    np.savez_compressed(OUT / "encoder_12layer.npz", lengths=np.array(lengths12), wseed=3, bseed=4, emb=emb12,
                        source="transformers.MPNetModel 5.15.0 + mean pooling + normalize", oracle_maxdiff=d12)
    from transformers.models.mpnet import modeling_mpnet as mm

    rel = torch.arange(-511, 512)
    tab = mm.MPNetEncoder.relative_position_bucket(rel.view(1, -1), num_buckets=32)[0].numpy().astype(np.int8)
    np.save(OUT / "rel_bucket_table.npy", tab)
    print("wrote", [p.name for p in OUT.glob("encoder_*")] + ["rel_bucket_table.npy"], "oracle vs transformers:", d2, d12)


if __name__ == "__main__":
    main()
