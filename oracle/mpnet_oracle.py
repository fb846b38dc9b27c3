"""TEST INFRASTRUCTURE, NOT PRODUCT CODE -- CPU oracle for the sentence encoder.

Plain PyTorch fp32 restatement of what the reference reaches through
``SentenceTransformer("all-mpnet-base-v2").encode(..., normalize_embeddings=True)``
(``src/embeddings.py:184-188``, ``:216-222``): MPNet encoder -> masked mean
pooling -> L2 normalise.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.

The arithmetic lives in third-party packages that are not part of the reference
checkout: ``sentence-transformers>=5.0.0`` -> ``transformers>=4.53.1``
(``pyproject.toml:8,13`` of the reference).  sentence-transformers is not
installed here; ``transformers`` 5.15.0 is, and this restatement follows its
``models/mpnet/modeling_mpnet.py`` (line numbers of that version):

  * position ids ``cumsum(mask)*mask + padding_idx`` ............ :873-881
  * ``LayerNorm(word_emb[ids] + pos_emb[pos])`` .................. :58-95
  * relative position bucket / shared bias ...................... :312-348
  * per layer: q,k,v projections, ``q.k^T/sqrt(64) + bias + mask``,
    softmax, ``P.v``, ``LN(o(c) + x)``, ``LN(W2 gelu(W1 a) + a)`` :115-261
  * pooling / normalise: sentence-transformers ``Pooling(mean)`` then
    ``Normalize`` [from knowledge; SURVEY.md App. A item 5]

PARITY PINNING: the reference's own tests mock ``SentenceTransformer`` everywhere
(``tests/test_embeddings.py:110-411``) and hold no numeric fixture for the encoder,
and no model weights / vocabulary exist offline, so encoder parity is pinned by
(a) ``tests/test_oracle_mpnet.py`` checking this restatement against the
in-container ``transformers.MPNetModel`` on seeded weights, and (b) the committed
goldens generated from it (``tests/golden/make_encoder_goldens.py``).  Parity with
the real all-mpnet-base-v2 checkpoint is "unpinned" (no weights offline).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from claude_semantic_search_amd import synth


@dataclass
class MpnetCfg:
    num_layers: int = 12
    hidden: int = 768
    heads: int = 12
    ffn: int = 3072
    vocab: int = 30527
    max_pos: int = 514
    rel_buckets: int = 32
    pad_id: int = 1
    max_seq_len: int = 384
    ln_eps: float = 1e-5


def _t(seed: int, tid: int, shape, mean: float, std: float) -> torch.Tensor:
    n = int(np.prod(shape))
    v = synth.normal(synth.tensor_seed(seed, tid), np.arange(n, dtype=np.uint64))
    out = np.float32(mean) + np.float32(std) * v
    return torch.from_numpy(out.astype(np.float32).reshape(shape))


def synth_weights(cfg: MpnetCfg, seed: int) -> Dict[str, torch.Tensor]:
    """Same tensors, bit for bit, as ``css_encoder_init_synthetic`` (DESIGN.md
    "synthetic weights"): weights N(0, 0.02^2), biases N(0, 0.05^2), LayerNorm
    gamma N(1, 0.1^2) / beta N(0, 0.05^2), relative bias N(0, 0.1^2); pad rows zero."""
    H, Fd = cfg.hidden, cfg.ffn
    w: Dict[str, torch.Tensor] = {}
    w["embeddings.word_embeddings.weight"] = _t(seed, 0, (cfg.vocab, H), 0.0, 0.02)
    w["embeddings.position_embeddings.weight"] = _t(seed, 1, (cfg.max_pos, H), 0.0, 0.02)
    w["embeddings.word_embeddings.weight"][cfg.pad_id] = 0
    w["embeddings.position_embeddings.weight"][cfg.pad_id] = 0
    w["embeddings.LayerNorm.weight"] = _t(seed, 2, (H,), 1.0, 0.1)
    w["embeddings.LayerNorm.bias"] = _t(seed, 3, (H,), 0.0, 0.05)
    w["encoder.relative_attention_bias.weight"] = _t(seed, 4, (cfg.rel_buckets, cfg.heads), 0.0, 0.1)
    for i in range(cfg.num_layers):
        p, b = f"encoder.layer.{i}.", 16 + 16 * i
        qkv_w = _t(seed, b + 0, (3 * H, H), 0.0, 0.02)
        qkv_b = _t(seed, b + 1, (3 * H,), 0.0, 0.05)
        for j, nm in enumerate("qkv"):
            w[p + f"attention.attn.{nm}.weight"] = qkv_w[j * H:(j + 1) * H].clone()
            w[p + f"attention.attn.{nm}.bias"] = qkv_b[j * H:(j + 1) * H].clone()
        w[p + "attention.attn.o.weight"] = _t(seed, b + 6, (H, H), 0.0, 0.02)
        w[p + "attention.attn.o.bias"] = _t(seed, b + 7, (H,), 0.0, 0.05)
        w[p + "attention.LayerNorm.weight"] = _t(seed, b + 8, (H,), 1.0, 0.1)
        w[p + "attention.LayerNorm.bias"] = _t(seed, b + 9, (H,), 0.0, 0.05)
        w[p + "intermediate.dense.weight"] = _t(seed, b + 10, (Fd, H), 0.0, 0.02)
        w[p + "intermediate.dense.bias"] = _t(seed, b + 11, (Fd,), 0.0, 0.05)
        w[p + "output.dense.weight"] = _t(seed, b + 12, (H, Fd), 0.0, 0.02)
        w[p + "output.dense.bias"] = _t(seed, b + 13, (H,), 0.0, 0.05)
        w[p + "output.LayerNorm.weight"] = _t(seed, b + 14, (H,), 1.0, 0.1)
        w[p + "output.LayerNorm.bias"] = _t(seed, b + 15, (H,), 0.0, 0.05)
    return w


def trained_like_weights(cfg: MpnetCfg, seed: int, outlier_dims: Sequence[int] = (7, 77, 300, 511, 640, 767),
                         gamma_gain: float = 4.0, emb_gain: float = 4.0, logit_gain: float = 1.5,
                         ffn_bias_gain: float = 10.0) -> Dict[str, torch.Tensor]:
    """``synth_weights`` reshaped towards the statistics of a TRAINED encoder, which N(0, 0.02^2) weights lack: a handful
    of hidden dimensions carry outlier magnitudes through the whole residual stream (LayerNorm gamma x ``gamma_gain``
    with alternating signs and beta +-3 on ``outlier_dims`` in EVERY LayerNorm, embedding columns x ``emb_gain``), attention
    logits are ``logit_gain`` x larger (q and k weights and biases x sqrt(gain): peaked softmax rows), and one FFN unit in
    twenty has a bias ``ffn_bias_gain`` x the others' (units that sit far inside GELU's linear or zero region).  This is
    the regime in which a bf16 residual stream with fixed-point row statistics is most exposed.
    Defaults (measured on the fp32 oracle, 12 layers): outlier channels ~33 x the mean |activation| behind every
    LayerNorm, attention logits of std 6-8 and |max| ~30 (a trained model's range).  The gains compound: 6 / 6 / 1.5 give
    outliers of ~64 x and logits up to ~70, 30 / 20 / 6 logits of +-7000 -- far outside any trained model, for overflow
    tests only."""
    w = synth_weights(cfg, seed)
    d = torch.tensor(list(outlier_dims), dtype=torch.long)
    sign = torch.tensor([1.0 if i % 2 == 0 else -1.0 for i in range(len(outlier_dims))])
    w["embeddings.word_embeddings.weight"][:, d] *= emb_gain
    w["embeddings.word_embeddings.weight"][cfg.pad_id] = 0
    ln_names = ["embeddings.LayerNorm"]
    for i in range(cfg.num_layers):
        ln_names += [f"encoder.layer.{i}.attention.LayerNorm", f"encoder.layer.{i}.output.LayerNorm"]
    for nm in ln_names:
        w[nm + ".weight"][d] *= gamma_gain * sign
        w[nm + ".bias"][d] += 3.0 * sign
    g = math.sqrt(logit_gain)
    for i in range(cfg.num_layers):
        p = f"encoder.layer.{i}."
        for nm in ("q", "k"):
            w[p + f"attention.attn.{nm}.weight"] *= g
            w[p + f"attention.attn.{nm}.bias"] *= g
        w[p + "intermediate.dense.bias"][::20] *= ffn_bias_gain
    return w


def relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """modeling_mpnet.py:326-348 (rel = key position - query position)."""
    ret = 0
    n = -rel
    num_buckets //= 2
    ret = ret + (n < 0).to(torch.long) * num_buckets
    n = torch.abs(n)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    return ret + torch.where(is_small, n, val_if_large)


def encode_tokens(w: Dict[str, torch.Tensor], cfg: MpnetCfg, ids: Sequence[int], dtype=torch.float32,
                  probes: dict | None = None) -> torch.Tensor:
    """Last hidden state [L, H] of ONE un-padded sequence."""
    H, nh = cfg.hidden, cfg.heads
    hd = H // nh
    x_ids = torch.tensor(list(ids), dtype=torch.long)
    L = x_ids.numel()
    mask = (x_ids != cfg.pad_id).long()
    pos = torch.cumsum(mask, 0) * mask + cfg.pad_id                       # :873-881
    g = lambda k: w[k].to(dtype)  # noqa: E731
    x = g("embeddings.word_embeddings.weight")[x_ids] + g("embeddings.position_embeddings.weight")[pos]
    x = F.layer_norm(x, (H,), g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), cfg.ln_eps)
    if probes is not None:
        probes["emb_ln"] = x.clone()
    ctx = torch.arange(L)[:, None]
    mem = torch.arange(L)[None, :]
    bucket = relative_position_bucket(mem - ctx, cfg.rel_buckets)          # :312-324
    bias = g("encoder.relative_attention_bias.weight")[bucket].permute(2, 0, 1)  # [heads, L, L]
    for i in range(cfg.num_layers):
        p = f"encoder.layer.{i}."
        q = x @ g(p + "attention.attn.q.weight").T + g(p + "attention.attn.q.bias")
        k = x @ g(p + "attention.attn.k.weight").T + g(p + "attention.attn.k.bias")
        v = x @ g(p + "attention.attn.v.weight").T + g(p + "attention.attn.v.bias")
        qh = q.view(L, nh, hd).transpose(0, 1)
        kh = k.view(L, nh, hd).transpose(0, 1)
        vh = v.view(L, nh, hd).transpose(0, 1)
        s = qh @ kh.transpose(1, 2) / math.sqrt(hd) + bias                 # :150-170
        pr = torch.softmax(s.float(), dim=-1).to(dtype)
        c = (pr @ vh).transpose(0, 1).reshape(L, H)
        a = c @ g(p + "attention.attn.o.weight").T + g(p + "attention.attn.o.bias")
        a = F.layer_norm(a + x, (H,), g(p + "attention.LayerNorm.weight"), g(p + "attention.LayerNorm.bias"), cfg.ln_eps)
        if probes is not None and i == 0:
            probes["attn_out"] = a.clone()
        h = F.gelu(a @ g(p + "intermediate.dense.weight").T + g(p + "intermediate.dense.bias"))  # exact erf GELU
        y = h @ g(p + "output.dense.weight").T + g(p + "output.dense.bias")
        x = F.layer_norm(y + a, (H,), g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), cfg.ln_eps)
        if probes is not None and i == 0:
            probes["ffn_out"] = x.clone()
    return x


def encode(w: Dict[str, torch.Tensor], cfg: MpnetCfg, batch: List[Sequence[int]], normalize: bool = True,
           dtype=torch.float32) -> np.ndarray:
    """[B, H] float32: masked mean pooling (all tokens of an un-padded sequence)
    with ``sum / max(count, 1e-9)``, then ``e / max(||e||, 1e-12)``."""
    out = []
    with torch.no_grad():
        for ids in batch:
            y = encode_tokens(w, cfg, ids, dtype).float()
            e = y.sum(0) / max(float(len(ids)), 1e-9)
            if normalize:
                e = F.normalize(e, p=2, dim=0, eps=1e-12)
            out.append(e.numpy())
    return np.stack(out).astype(np.float32)


def encode_batched(w: Dict[str, torch.Tensor], cfg: MpnetCfg, batch: List[Sequence[int]], batch_size: int = 16,
                   normalize: bool = True) -> np.ndarray:
    """The same model evaluated the way sentence-transformers runs it on a CPU: ``batch_size`` sequences at a
    time, padded to the longest of the batch, with the padded-key mask (modeling_mpnet.py: -inf on padded keys),
    batched matmuls on all torch threads.  Numerically the padded form of ``encode`` (tests/test_oracle_mpnet.py
    holds the two together); bench.py times this one as the encoder's CPU baseline."""
    H, nh = cfg.hidden, cfg.heads
    hd = H // nh
    out = []
    with torch.no_grad():
        for b0 in range(0, len(batch), batch_size):
            part = batch[b0:b0 + batch_size]
            B, L = len(part), max(len(s) for s in part)
            ids = torch.full((B, L), cfg.pad_id, dtype=torch.long)
            for i, sq in enumerate(part):
                ids[i, :len(sq)] = torch.tensor(list(sq), dtype=torch.long)
            mask = (ids != cfg.pad_id).long()
            pos = torch.cumsum(mask, 1) * mask + cfg.pad_id
            x = w["embeddings.word_embeddings.weight"][ids] + w["embeddings.position_embeddings.weight"][pos]
            x = F.layer_norm(x, (H,), w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], cfg.ln_eps)
            ar = torch.arange(L)
            bias = w["encoder.relative_attention_bias.weight"][relative_position_bucket(ar[None, :] - ar[:, None],
                                                                                        cfg.rel_buckets)].permute(2, 0, 1)
            keymask = torch.zeros((B, 1, 1, L))
            keymask.masked_fill_(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
            for i in range(cfg.num_layers):
                p = f"encoder.layer.{i}."
                lin = lambda t, nm: t @ w[p + nm + ".weight"].T + w[p + nm + ".bias"]  # noqa: E731
                q = lin(x, "attention.attn.q").view(B, L, nh, hd).transpose(1, 2)
                k = lin(x, "attention.attn.k").view(B, L, nh, hd).transpose(1, 2)
                v = lin(x, "attention.attn.v").view(B, L, nh, hd).transpose(1, 2)
                sc = q @ k.transpose(2, 3) / math.sqrt(hd) + bias[None] + keymask
                c = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, L, H)
                a = F.layer_norm(lin(c, "attention.attn.o") + x, (H,), w[p + "attention.LayerNorm.weight"],
                                 w[p + "attention.LayerNorm.bias"], cfg.ln_eps)
                y = lin(F.gelu(lin(a, "intermediate.dense")), "output.dense")
                x = F.layer_norm(y + a, (H,), w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], cfg.ln_eps)
            m = mask[:, :, None].float()
            e = (x * m).sum(1) / m.sum(1).clamp(min=1e-9)
            if normalize:
                e = F.normalize(e, p=2, dim=1, eps=1e-12)
            out.append(e.numpy())
    return np.concatenate(out).astype(np.float32)


def synth_batch(cfg: MpnetCfg, lengths: Sequence[int], seed: int) -> List[List[int]]:
    """Token ids: <s>=0 first, </s>=2 last, uniform in [4, vocab) between."""
    batch = []
    for b, L in enumerate(lengths):
        body = synth.uint(seed, np.arange(L, dtype=np.uint64) + np.uint64(b) * np.uint64(1 << 20), 4, cfg.vocab)
        ids = body.tolist()
        ids[0] = 0
        if L > 1:
            ids[-1] = 2
        batch.append(ids)
    return batch
