"""Host-side text -> token ids for the encoder (front end of ``encode``).

all-mpnet-base-v2 uses a BERT-style WordPiece tokenizer (lower-casing, accent
stripping, punctuation splitting, greedy longest-match pieces with "##"
continuation) and the special ids ``<s>``=0, ``<pad>``=1, ``</s>``=2,
``<unk>``=3 [from knowledge of the public model card; SURVEY.md App. A].  No
vocabulary file exists offline, so:

  * ``WordPieceTokenizer(vocab_path)`` implements the algorithm against a
    user-supplied ``vocab.txt`` (parity with the real tokenizer is unpinned:
    SURVEY.md 8f rank 3);
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


def basic_tokenize(text: str, lower: bool = True) -> List[str]:
    if lower:
        text = text.lower()
        text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
    out: List[str] = []
    word: List[str] = []
    for ch in text:
        if ch.isspace() or ord(ch) == 0 or unicodedata.category(ch) in ("Cc", "Cf"):
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
        self.unk = self.vocab.get("<unk>", self.vocab.get("[UNK]", UNK))
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
