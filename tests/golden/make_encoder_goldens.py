"""Generates tests/golden/encoder_*.npz from the CPU oracle (oracle/mpnet_oracle.py,
itself pinned against transformers.MPNetModel by tests/test_oracle_mpnet.py).

    python tests/golden/make_encoder_goldens.py

G4 (encoder_2layer.npz): 2-layer config, B = 6, lengths {2, 7, 31, 128, 383, 384}, weights seed 7,
    token seed 11 -> pooled+normalised [6,768] and layer-0 probes (emb-LN, attention-block output,
    layer output: first 8 dims of the first and last token of every sequence).
G5 (encoder_12layer.npz): full 12-layer config, B = 8 mixed lengths, weights seed 3, token seed 4
    -> [8,768].
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


def main():
    torch.set_num_threads(8)
    cfg = mo.MpnetCfg(num_layers=2)
    lengths = [2, 7, 31, 128, 383, 384]
    w = mo.synth_weights(cfg, 7)
    batch = mo.synth_batch(cfg, lengths, seed=11)
    emb = mo.encode(w, cfg, batch)
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
                        **{k: np.stack(v) for k, v in probes.items()})
    cfg12 = mo.MpnetCfg()
    lengths12 = [8, 40, 100, 384, 17, 250, 1, 64]
    emb12 = mo.encode(mo.synth_weights(cfg12, 3), cfg12, mo.synth_batch(cfg12, lengths12, seed=4))
    np.savez_compressed(OUT / "encoder_12layer.npz", lengths=np.array(lengths12), wseed=3, bseed=4, emb=emb12)
    from transformers.models.mpnet import modeling_mpnet as mm

    rel = torch.arange(-511, 512)
    tab = mm.MPNetEncoder.relative_position_bucket(rel.view(1, -1), num_buckets=32)[0].numpy().astype(np.int8)
    np.save(OUT / "rel_bucket_table.npy", tab)
    print("wrote", [p.name for p in OUT.glob("encoder_*")] + ["rel_bucket_table.npy"])


if __name__ == "__main__":
    main()
