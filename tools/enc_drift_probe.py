import numpy as np, sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from claude_semantic_search_amd import synth
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder
from oracle import mpnet_oracle as mo
cfg = mo.MpnetCfg(num_layers=12)
for kw in (dict(), dict(gamma_gain=4.0, emb_gain=4.0, logit_gain=1.5), dict(gamma_gain=1.0, emb_gain=1.0, logit_gain=1.0, ffn_bias_gain=1.0)):
    w = mo.trained_like_weights(cfg, 33, **kw)
    chars = 100 + synth.uint(5, np.arange(256, dtype=np.uint64), 0, 1901)
    lengths = np.clip(np.round(chars / 10).astype(np.int64) + 2, 2, 384).tolist()
    lengths[0], lengths[1], lengths[2] = 384, 1, 383
    batch = mo.synth_batch(cfg, lengths, seed=6)
    ref = mo.encode_batched(w, cfg, batch, batch_size=16)
    enc = MpnetEncoder(synthetic_seed=33, compute="bf16")
    enc.load_state_dict({k: v.numpy() for k, v in w.items()})
    folded = enc.encode_ids(batch)
    small = np.concatenate([enc.encode_ids(batch[i:i + 4]) for i in range(0, 256, 4)])
    enc.close()
    for name, got in (("folded", folded), ("small", small)):
        cos = (got * ref).sum(1)
        d = np.abs(got @ got.T - ref @ ref.T)
        print(kw, name, 'min cos %.7f' % cos.min(), 'drift max %.2e mean %.2e p99.9 %.2e' % (d.max(), d.mean(), np.quantile(d, 0.999)), flush=True)
