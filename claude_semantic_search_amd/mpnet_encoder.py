"""``MpnetEncoder``: the sentence encoder on the MI355X behind the six
``SentenceTransformer`` members the reference touches:
``.to(dev)``, ``.max_seq_length``, ``.get_sentence_embedding_dimension()``,
``.device`` and ``.encode(str|list, batch_size=, normalize_embeddings=,
show_progress_bar=, convert_to_numpy=)`` (``src/embeddings.py:86-117``,
``:184-188``, ``:216-222``).

``encode`` semantics restated from sentence-transformers [from knowledge,
SURVEY.md App. A item 7]: a str gives a 1-D vector, a list gives ``[n, 768]``;
sentences are sorted by length (descending), processed in batches of
``batch_size``, truncated to ``max_seq_length`` tokens, and returned in input
order as float32.  Pooling(mean) + Normalize() are part of the model, so outputs
are unit norm even before ``normalize_embeddings=True`` re-normalises them.
Batches are packed var-len (no padding) before they cross the C ABI.
"""
from __future__ import annotations

import ctypes
import json
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _native as nat
from .tokenizer import HashTokenizer, WordPieceTokenizer, make_wordpiece  # noqa: F401

DEFAULT_CFG = dict(num_layers=12, hidden=768, heads=12, ffn=3072, vocab=30527, max_pos=514, rel_buckets=32,
                   pad_id=1, max_seq_len=384, ln_eps=1e-5)


def _find_model_dir(name_or_path: str, cache_folder: Optional[str]) -> Optional[Path]:
    cands = [Path(name_or_path)]
    if cache_folder:
        cands += [Path(cache_folder) / name_or_path, Path(cache_folder) / f"sentence-transformers_{name_or_path}"]
    home = os.environ.get("SENTENCE_TRANSFORMERS_HOME")
    if home:
        cands += [Path(home) / name_or_path, Path(home) / f"sentence-transformers_{name_or_path}"]
    for c in cands:
        if c.is_dir() and ((c / "config.json").exists() or (c / "0_Transformer").is_dir()):
            return c
    return None


def _load_state_dict(model_dir: Path) -> Dict[str, np.ndarray]:
    sub = model_dir / "0_Transformer"
    root = sub if sub.is_dir() else model_dir
    st = root / "model.safetensors"
    if st.exists():
        from safetensors.numpy import load_file

        return {k: np.asarray(v, dtype=np.float32) for k, v in load_file(str(st)).items()}
    pt = root / "pytorch_model.bin"
    if pt.exists():
        import torch

        sd = torch.load(str(pt), map_location="cpu", weights_only=True)
        return {k: v.float().numpy() for k, v in sd.items()}
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {root}")


class MpnetEncoder:
    def __init__(self, model_name_or_path: Optional[str] = "all-mpnet-base-v2", cache_folder: Optional[str] = None,
                 device: int = 0, compute: str = "bf16", synthetic_seed: Optional[int] = None,
                 cfg_overrides: Optional[dict] = None):
        cfg = dict(DEFAULT_CFG)
        model_dir = None
        if synthetic_seed is None:
            model_dir = _find_model_dir(model_name_or_path or "", cache_folder)
            if model_dir is None:
                raise FileNotFoundError(
                    f"model '{model_name_or_path}' not found locally (no network): pass a directory in HF layout, "
                    "or synthetic_seed=<int> for seeded synthetic weights"
                )
            root = model_dir / "0_Transformer" if (model_dir / "0_Transformer").is_dir() else model_dir
            hf = json.loads((root / "config.json").read_text())
            cfg.update(num_layers=hf.get("num_hidden_layers", 12), hidden=hf.get("hidden_size", 768),
                       heads=hf.get("num_attention_heads", 12), ffn=hf.get("intermediate_size", 3072),
                       vocab=hf.get("vocab_size", 30527), max_pos=hf.get("max_position_embeddings", 514),
                       rel_buckets=hf.get("relative_attention_num_buckets", 32), pad_id=hf.get("pad_token_id", 1),
                       ln_eps=hf.get("layer_norm_eps", 1e-5))
        if cfg_overrides:
            cfg.update(cfg_overrides)
        self.cfg = cfg
        self._device_index = int(device)
        self.compute = compute
        c = nat.EncoderCfg(cfg["num_layers"], cfg["hidden"], cfg["heads"], cfg["ffn"], cfg["vocab"], cfg["max_pos"],
                           cfg["rel_buckets"], cfg["pad_id"], cfg["max_seq_len"], cfg["ln_eps"],
                           0 if compute == "bf16" else 1)
        h = ctypes.c_void_p()
        nat.check(nat.lib().css_encoder_create(ctypes.byref(c), self._device_index, ctypes.byref(h)))
        self._h = h
        self.max_seq_length = cfg["max_seq_len"]
        if synthetic_seed is not None:
            nat.check(nat.lib().css_encoder_init_synthetic(self._h, ctypes.c_uint64(synthetic_seed)))
            self.tokenizer = HashTokenizer(cfg["vocab"])
        else:
            self.load_state_dict(_load_state_dict(model_dir))
            root = model_dir / "0_Transformer" if (model_dir / "0_Transformer").is_dir() else model_dir
            vocab = root / "vocab.txt"
            self.tokenizer = make_wordpiece(str(vocab)) if vocab.exists() else HashTokenizer(cfg["vocab"])

    # -- lifetime -----------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            nat.lib().css_encoder_free(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- weights ------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, np.ndarray]) -> None:
        names = [k for k in sd if "pooler." not in k and "position_ids" not in k]
        arr = (nat.Tensor * len(names))()
        keep = []
        for i, k in enumerate(names):
            a = np.ascontiguousarray(sd[k], dtype=np.float32)
            keep.append(a)
            arr[i].name = k.encode()
            arr[i].data = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
            arr[i].numel = a.size
        nat.check(nat.lib().css_encoder_load_weights(self._h, arr, len(names)))

    def export_weight(self, name: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        nat.check(nat.lib().css_encoder_export_weight(self._h, name.encode(), out.ctypes.data, out.size))
        return out

    def set_attention_range(self, value: float) -> None:
        """Verification knob of the bf16 attention kernel (``css_encoder_set_attention_range``): 0 forces the
        running-maximum softmax pass, the default 2**100 keeps the reference-free pass while row sums fit fp32."""
        nat.check(nat.lib().css_encoder_set_attention_range(self._h, ctypes.c_float(value)))

    def debug_read(self, what: str, shape) -> np.ndarray:
        out = np.empty(shape, dtype=np.float32)
        nat.check(nat.lib().css_encoder_debug_read(self._h, what.encode(), out.ctypes.data, out.size))
        return out

    # -- SentenceTransformer surface ------------------------------------------
    def to(self, device) -> "MpnetEncoder":
        return self  # the model already lives on its HIP device

    @property
    def device(self) -> str:
        return f"cuda:{self._device_index}"  # the name PyTorch-ROCm gives a HIP device

    def get_sentence_embedding_dimension(self) -> int:
        return int(self.cfg["hidden"])

    def tokenize(self, texts: Sequence[str]) -> List[List[int]]:
        L = min(int(self.max_seq_length), int(self.cfg["max_seq_len"]))
        if hasattr(self.tokenizer, "encode_batch"):
            return self.tokenizer.encode_batch(texts, L)
        return [self.tokenizer.encode(t, L) for t in texts]

    def encode_ids(self, batch: Sequence[Sequence[int]], normalize: bool = True) -> np.ndarray:
        """One packed var-len batch through ``css_encoder_forward``."""
        B = len(batch)
        lens = np.fromiter((len(s) for s in batch), dtype=np.int32, count=B)
        cu = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(lens, out=cu[1:])
        ids = np.concatenate([np.asarray(s, dtype=np.int32) for s in batch]) if B else np.zeros(0, np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.empty((B, self.cfg["hidden"]), dtype=np.float32)
        nat.check(nat.lib().css_encoder_forward(self._h, ids.ctypes.data, cu.ctypes.data, B, 1 if normalize else 0,
                                                out.ctypes.data))
        return out

    def encode(self, sentences: Union[str, Sequence[str]], batch_size: int = 32, normalize_embeddings: bool = False,
               show_progress_bar: bool = False, convert_to_numpy: bool = True, **_ignored) -> np.ndarray:
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        if not texts:
            return np.zeros((0, self.cfg["hidden"]), dtype=np.float32)
        out = np.empty((len(texts), self.cfg["hidden"]), dtype=np.float32)
        bs = max(1, int(batch_size))
        # Super-batches of 16 device batches: the next one is tokenised on a host thread (both the C++ tokenizer and
        # the forward release the GIL) while the GPU encodes the current one; texts are length-sorted inside a
        # super-batch (packed var-len batches have no padding, sorting only keeps a batch's attention tiles even).
        span = max(16 * bs, 1024)
        starts = list(range(0, len(texts), span))
        bar = None
        if show_progress_bar:
            try:
                from tqdm import tqdm

                bar = tqdm(total=(len(texts) + bs - 1) // bs, desc="Batches")
            except Exception:
                bar = None
        def run(base: int, toks) -> None:
            order = sorted(range(len(toks)), key=lambda i: -len(toks[i]))
            for s in range(0, len(order), bs):
                idx = order[s:s + bs]
                # Normalize() is a module of the model: outputs are unit norm regardless of the flag
                out[[base + i for i in idx]] = self.encode_ids([toks[i] for i in idx], normalize=True)
                if bar is not None:
                    bar.update(1)

        if len(starts) == 1:  # (the single-query path: no thread hop)
            run(0, self.tokenize(texts))
        else:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=1) as pool:
                fut = pool.submit(self.tokenize, texts[:span])
                for si, base in enumerate(starts):
                    toks = fut.result()
                    if si + 1 < len(starts):
                        fut = pool.submit(self.tokenize, texts[starts[si + 1]:starts[si + 1] + span])
                    run(base, toks)
        if bar is not None:
            bar.close()
        return out[0] if single else out
