import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import mpnet_oracle as mo
from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder
cfg = mo.MpnetCfg(num_layers=1)
for L in (2, 5, 40):
    batch = mo.synth_batch(cfg, [L], seed=11)
    enc = MpnetEncoder(synthetic_seed=7, compute="bf16", cfg_overrides={"num_layers": 1})
    enc.encode_ids(batch)
    qkv = enc.debug_read("qkv", (L, 2304)); ctx = enc.debug_read("ctx", (L, 768))
    q = torch.from_numpy(qkv[:, :768]).view(L,12,64).transpose(0,1)
    k = torch.from_numpy(qkv[:, 768:1536]).view(L,12,64).transpose(0,1)
    v = torch.from_numpy(qkv[:, 1536:]).view(L,12,64).transpose(0,1)
    relw = torch.from_numpy(enc.export_weight("encoder.relative_attention_bias.weight", (32,12)))
    ci = torch.arange(L)[:,None]; mi = torch.arange(L)[None,:]
    bias = relw[mo.relative_position_bucket(mi-ci,32)].permute(2,0,1)
    s = q @ k.transpose(1,2) + bias
    P = torch.softmax(s,-1)
    ref = (P @ v).transpose(0,1).reshape(L,768).numpy()
    print("L", L, "ctx err vs ref-from-device-qkv", np.abs(ctx-ref).max())
    h = 0
    V = v[h].numpy()  # [L,64]
    for qi in range(min(L,3)):
        c = ctx[qi, h*64:(h+1)*64]
        sol, res, *_ = np.linalg.lstsq(V.T, c, rcond=None)
        print("  q", qi, "kernel weights", np.round(sol[:6],3), "oracle P", np.round(P[h,qi,:6].numpy(),3), "resid", float(np.abs(V.T@sol-c).max()))
