import sys; sys.path.insert(0,'.')
import numpy as np, torch, math
from oracle import mpnet_oracle as mo
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder
cfg = mo.MpnetCfg(num_layers=1)
w = mo.synth_weights(cfg, 7)
batch = mo.synth_batch(cfg, [2, 7, 31], seed=11)
T = sum(len(b) for b in batch)
# oracle intermediates for layer 0
def oracle_inter(ids):
    pr = {}
    y = mo.encode_tokens(w, cfg, ids, probes=pr)
    x = pr["emb_ln"]
    p = "encoder.layer.0."
    q = (x @ w[p+"attention.attn.q.weight"].T + w[p+"attention.attn.q.bias"]) / 8
    k = x @ w[p+"attention.attn.k.weight"].T + w[p+"attention.attn.k.bias"]
    v = x @ w[p+"attention.attn.v.weight"].T + w[p+"attention.attn.v.bias"]
    L = len(ids)
    qh = q.view(L,12,64).transpose(0,1); kh = k.view(L,12,64).transpose(0,1); vh = v.view(L,12,64).transpose(0,1)
    ctxi = torch.arange(L)[:,None]; mem = torch.arange(L)[None,:]
    bias = w["encoder.relative_attention_bias.weight"][mo.relative_position_bucket(mem-ctxi,32)].permute(2,0,1)
    s = qh @ kh.transpose(1,2) + bias
    c = (torch.softmax(s,-1) @ vh).transpose(0,1).reshape(L,768)
    return torch.cat([q,k,v],1).numpy(), c.numpy(), y.numpy()
oq, oc, oy = zip(*[oracle_inter(b) for b in batch])
oq = np.concatenate(oq); oc = np.concatenate(oc); oy = np.concatenate(oy)
for mode in ("fp32", "bf16"):
    enc = MpnetEncoder(synthetic_seed=7, compute=mode, cfg_overrides={"num_layers": 1})
    out = enc.encode_ids(batch)
    qkv = enc.debug_read("qkv", (T, 2304)); ctx = enc.debug_read("ctx", (T, 768)); x32 = enc.debug_read("x32", (T, 768))
    print(mode, "qkv err", np.abs(qkv-oq).max(), "ctx err", np.abs(ctx-oc).max(), "final err", np.abs(x32-oy).max())
    e = np.abs(ctx-oc); 
    if e.max() > 0.05:
        bad = np.argwhere(e > 0.05); print(" bad ctx entries", len(bad), "rows", sorted(set(bad[:,0].tolist()))[:20], "cols sample", sorted(set(bad[:,1].tolist()))[:40])
