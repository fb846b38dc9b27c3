"""Tiny encoder check used by ``__graft_entry__.smoke()`` (imports the oracle,
so it is only ever called from there)."""
import numpy as np


def run() -> None:
    from oracle import mpnet_oracle as mo
    from .mpnet_encoder import MpnetEncoder

    cfg = mo.MpnetCfg(num_layers=2)
    batch = mo.synth_batch(cfg, [5, 33, 64], seed=3)
    ref = mo.encode(mo.synth_weights(cfg, 9), cfg, batch)
    for mode, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        enc = MpnetEncoder(synthetic_seed=9, compute=mode, cfg_overrides={"num_layers": 2})
        out = enc.encode_ids(batch)
        cos = (out * ref).sum(1)
        assert np.abs(out - ref).max() < tol and cos.min() > 1 - 1e-3, (mode, np.abs(out - ref).max(), cos.min())
        enc.close()
    print("smoke ok: 2-layer MPNet encoder (fp32 and bf16 modes) matches the oracle")
