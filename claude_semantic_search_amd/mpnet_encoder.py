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
import logging
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np

from . import _native as nat
from .tokenizer import HashTokenizer, WordPieceTokenizer, make_wordpiece  # noqa: F401

DEFAULT_CFG = dict(num_layers=12, hidden=768, heads=12, ffn=3072, vocab=30527, max_pos=514, rel_buckets=32,
                   pad_id=1, max_seq_len=384, ln_eps=1e-5)


def _is_model_dir(c: Path) -> bool:
    return c.is_dir() and ((c / "config.json").exists() or (c / "0_Transformer").is_dir())


def _hub_snapshots(root: Path, name: str) -> List[Path]:
    """Snapshot directories of the HF-hub cache layout ``root/models--<org>--<name>/snapshots/<rev>/`` -- where
    sentence-transformers >= 3 (the reference pins >= 5, ``pyproject.toml``) leaves an auto-downloaded model when it
    is given ``cache_folder`` (``src/embeddings.py:81-88``).  ``refs/main`` names the current revision; otherwise
    the most recently modified snapshot wins."""
    org, _, base = name.rpartition("/")
    repos = [f"models--{org.replace('/', '--')}--{base}"] if org else [f"models--sentence-transformers--{base}", f"models--{base}"]
    out: List[Path] = []
    for repo in repos:
        snaps = root / repo / "snapshots"
        if not snaps.is_dir():
            continue
        ref = root / repo / "refs" / "main"
        if ref.is_file():
            cand = snaps / ref.read_text().strip()
            if cand.is_dir():
                out.append(cand)
        out += sorted((d for d in snaps.iterdir() if d.is_dir()), key=lambda d: -d.stat().st_mtime)
    return out


def _find_model_dir(name_or_path: str, cache_folder: Optional[str]) -> Optional[Path]:
    """Where ``SentenceTransformer(name, cache_folder=...)`` would find the model without a network: the path itself,
    the flat layouts ``<cache>/<name>`` and ``<cache>/sentence-transformers_<name>`` (what the reference's
    ``scripts/model_setup.py:38-52`` writes), and the HF-hub layout under ``cache_folder``,
    ``$SENTENCE_TRANSFORMERS_HOME``, ``$HF_HUB_CACHE`` / ``$HF_HOME/hub`` and ``~/.cache/huggingface/hub``."""
    cands: List[Path] = [Path(name_or_path)]
    base = name_or_path.rpartition("/")[2]
    roots: List[Path] = []
    if cache_folder:
        roots.append(Path(cache_folder))
    home = os.environ.get("SENTENCE_TRANSFORMERS_HOME")
    if home:
        roots.append(Path(home))
    hub_roots = list(roots)
    if os.environ.get("HF_HUB_CACHE"):
        hub_roots.append(Path(os.environ["HF_HUB_CACHE"]))
    if os.environ.get("HF_HOME"):
        hub_roots.append(Path(os.environ["HF_HOME"]) / "hub")
    hub_roots.append(Path.home() / ".cache" / "huggingface" / "hub")
    for r in roots:
        cands += [r / name_or_path, r / base, r / f"sentence-transformers_{base}"]
    for r in hub_roots:
        cands += _hub_snapshots(r, name_or_path)
    for c in cands:
        if _is_model_dir(c):
            return c
    return None


def _tokenizer_files(model_dir: Path) -> Dict[str, object]:
    """``vocab.txt`` and the lower-casing flag of a checkpoint directory.  Modern sentence-transformers layouts keep the
    tokenizer files at the ROOT beside ``modules.json`` even when the weights sit in ``0_Transformer/``; older ones
    keep everything in ``0_Transformer/``.  ``do_lower_case`` comes from ``tokenizer_config.json`` (all-mpnet-base-v2:
    true -- MPNetTokenizer lower-cases; the ``do_lower_case: false`` of ``sentence_bert_config.json`` only says that
    sentence-transformers does not lower-case a second time)."""
    places = [model_dir, model_dir / "0_Transformer"]
    vocab = next((p / "vocab.txt" for p in places if (p / "vocab.txt").is_file()), None)
    lower = True
    for p in places:
        f = p / "tokenizer_config.json"
        if f.is_file():
            lower = bool(json.loads(f.read_text()).get("do_lower_case", True))
            break
    max_len = None
    for p in places:
        f = p / "sentence_bert_config.json"
        if f.is_file():
            max_len = json.loads(f.read_text()).get("max_seq_length")
            break
    return {"vocab": vocab, "lower": lower, "max_seq_length": max_len}


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
        self.model_dir = model_dir
        #: why text cannot be encoded (None when it can); surfaced by EmbeddingGenerator.get_model_info()
        self.tokenizer_problem: Optional[str] = None
        if synthetic_seed is not None:
            nat.check(nat.lib().css_encoder_init_synthetic(self._h, ctypes.c_uint64(synthetic_seed)))
            self.tokenizer = HashTokenizer(cfg["vocab"])   # synthetic weights: any deterministic text -> ids map will do
        else:
            self.load_state_dict(_load_state_dict(model_dir))
            tk = _tokenizer_files(model_dir)
            if tk["max_seq_length"]:
                self.max_seq_length = min(int(tk["max_seq_length"]), cfg["max_seq_len"])
            if tk["vocab"] is not None:
                self.tokenizer = make_wordpiece(str(tk["vocab"]), lower=bool(tk["lower"]))
            else:
                # REAL weights with word ids from a hash would give plausible-looking garbage: refuse text instead
                self.tokenizer = None
                self.tokenizer_problem = (f"checkpoint {model_dir} has no vocab.txt (looked beside config.json and in "
                                          "0_Transformer/): text cannot be tokenised for these weights")
                logging.getLogger(__name__).warning(self.tokenizer_problem + "; encode_ids() still works")

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
        if self.tokenizer is None:
            raise RuntimeError(self.tokenizer_problem or "no tokenizer")
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
