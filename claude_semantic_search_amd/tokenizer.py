"""Host-side text -> token ids for the encoder (front end of ``encode``).

all-mpnet-base-v2 uses a BERT-style WordPiece tokenizer (lower-casing, accent
stripping, punctuation splitting, greedy longest-match pieces with "##"
continuation) and the special ids ``<s>``=0, ``<pad>``=1, ``</s>``=2,
``<unk>``=3 [from knowledge of the public model card; SURVEY.md App. A].  No
vocabulary file exists offline, so:

  * ``WordPieceTokenizer(vocab_path)`` implements the algorithm against a
    user-supplied ``vocab.txt``; the algorithm is pinned against ``transformers``'
    MPNetTokenizer on a synthetic vocabulary (``tests/test_tokenizer.py``), parity on the
    real vocabulary is unpinned;
  * ``FastWordPieceTokenizer`` runs the same pipeline on the HF ``tokenizers`` library (the
    native implementation the reference's sentence-transformers stack uses) and encodes
    batches in parallel; ``make_wordpiece`` picks it when the library is importable;
  * ``HashTokenizer`` is the stand-in used with synthetic weights: the same basic
    tokenisation, each word hashed into ``[4, vocab)``.

Both produce ``[<s>] + pieces[: max_len - 2] + [</s>]`` (truncation to
``max_seq_length``, ``src/embeddings.py:97``).
"""
from __future__ import annotations

import unicodedata
import zlib
from typing import Dict, List, Optional

BOS, PAD, EOS, UNK = 0, 1, 2, 3


def _is_punct(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    # the CJK ideograph blocks BERT's tokenizer isolates character by character
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


def basic_tokenize(text: str, lower: bool = True) -> List[str]:
    """BERT's basic tokenisation in the order of the HF ``tokenizers`` pipeline MPNet uses
    (BertNormalizer(clean_text, handle_chinese_chars, strip_accents = lowercase, lowercase) +
    BertPreTokenizer): drop NUL / U+FFFD / control characters, every whitespace -> " ", CJK ideographs
    isolated, NFD + combining marks removed, lower-cased, split on whitespace and around each punctuation mark."""
    cleaned: List[str] = []
    for ch in text:
        cp = ord(ch)
        if ch in "\t\n\r":
            cleaned.append(" ")
        elif cp == 0 or cp == 0xFFFD or unicodedata.category(ch).startswith("C"):
            continue
        elif ch == " " or unicodedata.category(ch) == "Zs":
            cleaned.append(" ")
        elif _is_cjk(cp):
            cleaned.extend((" ", ch, " "))
        else:
            cleaned.append(ch)
    text = "".join(cleaned)
    if lower:
        text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
        text = text.lower()
    out: List[str] = []
    word: List[str] = []
    for ch in text:
        if ch.isspace():
            if word:
                out.append("".join(word))
                word = []
        elif _is_punct(ch):
            if word:
                out.append("".join(word))
                word = []
            out.append(ch)
        else:
            word.append(ch)
    if word:
        out.append("".join(word))
    return out


class HashTokenizer:
    def __init__(self, vocab_size: int = 30527, lower: bool = True):
        self.vocab_size = vocab_size
        self.lower = lower

    def encode(self, text: str, max_len: int = 384) -> List[int]:
        ids = [BOS]
        for w in basic_tokenize(text, self.lower)[: max(0, max_len - 2)]:
            ids.append(4 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 5))
        ids.append(EOS)
        return ids[:max_len] if max_len >= 2 else ids[:1]


class WordPieceTokenizer:
    def __init__(self, vocab_path: str, lower: bool = True, max_chars_per_word: int = 100):
        self.vocab: Dict[str, int] = {}
        with open(vocab_path, encoding="utf-8") as f:
            for i, line in enumerate(f):
                self.vocab[line.rstrip("\n")] = i
        self.lower = lower
        self.max_chars = max_chars_per_word
        self.vocab_size = len(self.vocab)
        # MPNet's tokenizer config names "[UNK]" (not "<unk>") as the unknown token [from knowledge of the model card]
        self.unk = self.vocab.get("[UNK]", self.vocab.get("<unk>", UNK))
        self.bos = self.vocab.get("<s>", self.vocab.get("[CLS]", BOS))
        self.eos = self.vocab.get("</s>", self.vocab.get("[SEP]", EOS))

    def _pieces(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk]
        out, start = [], 0
        while start < len(word):
            end, cur = len(word), None
            while start < end:
                sub = word[start:end] if start == 0 else "##" + word[start:end]
                if sub in self.vocab:
                    cur = self.vocab[sub]
                    break
                end -= 1
            if cur is None:
                return [self.unk]
            out.append(cur)
            start = end
        return out

    def encode(self, text: str, max_len: int = 384) -> List[int]:
        ids: List[int] = []
        budget = max(0, max_len - 2)
        for w in basic_tokenize(text, self.lower):
            ids.extend(self._pieces(w))
            if len(ids) >= budget:
                break
        return [self.bos] + ids[:budget] + [self.eos]


class FastWordPieceTokenizer(WordPieceTokenizer):
    """The same tokenisation on the HF ``tokenizers`` library: WordPiece + BertNormalizer + BertPreTokenizer, the
    recipe of transformers' MPNetTokenizer.  ``encode_batch`` runs on all host cores (the pure-Python class needs
    ~2 ms per 1.5 kB chunk, twenty times the GPU time of the encoder forward it feeds)."""

    def __init__(self, vocab_path: str, lower: bool = True, max_chars_per_word: int = 100):
        super().__init__(vocab_path, lower, max_chars_per_word)
        from tokenizers import Tokenizer, normalizers, pre_tokenizers
        from tokenizers.models import WordPiece

        unk_str = next(t for t, i in self.vocab.items() if i == self.unk)
        tk = Tokenizer(WordPiece(dict(self.vocab), unk_token=unk_str, max_input_chars_per_word=max_chars_per_word))
        tk.normalizer = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=True, strip_accents=None,
                                                   lowercase=lower)
        tk.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
        self._tk = tk

    def encode(self, text: str, max_len: int = 384) -> List[int]:
        return self.encode_batch([text], max_len)[0]

    def encode_batch(self, texts, max_len: int = 384) -> List[List[int]]:
        budget = max(0, max_len - 2)
        return [[self.bos] + e.ids[:budget] + [self.eos]
                for e in self._tk.encode_batch(list(texts), add_special_tokens=False)]


class NativeWordPieceTokenizer(WordPieceTokenizer):
    """Batch front end on the C++ tokenizer of libcss_hip.so (``css_tokenizer_*``, all host cores).  The library
    handles ASCII and UTF-8 texts through tables generated from this module's own rules
    (``tools/gen_unicode_tables.py``); what it reports back -- invalid UTF-8, texts with a capital sigma (final-sigma
    rule of ``str.lower``), non-ASCII text with lower-casing off -- is tokenised by this class's Python code, so both
    halves follow one specification.  ``last_fallbacks`` counts the texts of the last batch that took that route."""

    def __init__(self, vocab_path: str, lower: bool = True):
        super().__init__(vocab_path, lower)
        import ctypes

        from . import _native as nat

        self._nat, self._ct = nat, ctypes
        h = ctypes.c_void_p()
        nat.check(nat.lib().css_tokenizer_create(str(vocab_path).encode(), 1 if lower else 0, ctypes.byref(h)))
        self._h = h
        self.last_fallbacks = 0

    def __del__(self):  # pragma: no cover
        try:
            if getattr(self, "_h", None):
                self._nat.lib().css_tokenizer_free(self._h)
                self._h = None
        except Exception:
            pass

    def encode(self, text: str, max_len: int = 384) -> List[int]:
        return [int(v) for v in self.encode_batch([text], max_len)[0]]

    def encode_batch(self, texts, max_len: int = 384):
        """List of int32 numpy arrays (one per text, ``<s> ... </s>`` included)."""
        import numpy as np

        texts = list(texts)
        n = len(texts)
        if n == 0:
            return []
        raw = [t.encode("utf-8", "surrogatepass") for t in texts]
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.fromiter((len(b) for b in raw), dtype=np.int64, count=n), out=offsets[1:])
        blob = b"".join(raw)
        ids = np.empty((n, max_len), dtype=np.int32)
        lens = np.empty(n, dtype=np.int32)
        self._nat.check(self._nat.lib().css_tokenizer_encode_batch(self._h, blob, offsets.ctypes.data, n, int(max_len),
                                                                   ids.ctypes.data, lens.ctypes.data, 0))
        out = [ids[i, : lens[i]] if lens[i] >= 0 else None for i in range(n)]
        rest = [i for i in range(n) if lens[i] < 0]
        self.last_fallbacks = len(rest)
        for i in rest:
            out[i] = np.asarray(WordPieceTokenizer.encode(self, texts[i], max_len), dtype=np.int32)
        return out


def make_wordpiece(vocab_path: str, lower: bool = True) -> WordPieceTokenizer:
    """Fastest available implementation of the one pipeline: the C++ tokenizer of libcss_hip.so when the library is
    built, else ``FastWordPieceTokenizer`` (HF ``tokenizers``), else the pure-Python class."""
    try:
        return NativeWordPieceTokenizer(vocab_path, lower)
    except Exception:
        pass
    try:
        return FastWordPieceTokenizer(vocab_path, lower)
    except ImportError:
        return WordPieceTokenizer(vocab_path, lower)
