"""CPU: host-side tokenisation front end.  The WordPiece pipeline is pinned against transformers'
MPNetTokenizer (HF ``tokenizers`` backend) on a synthetic vocabulary; parity on the real all-mpnet-base-v2
vocabulary is unpinned because no vocabulary file exists offline."""
import random
import string

import pytest

from claude_semantic_search_amd.tokenizer import (BOS, EOS, FastWordPieceTokenizer, HashTokenizer, WordPieceTokenizer,
                                                  basic_tokenize, make_wordpiece)


def test_basic_tokenize_lowercase_accents_punct():
    assert basic_tokenize("Hello, Wörld! x=1") == ["hello", ",", "world", "!", "x", "=", "1"]
    assert basic_tokenize("  \t\n ") == []
    assert basic_tokenize("don't") == ["don", "'", "t"]
    assert basic_tokenize("Keep CASE", lower=False) == ["Keep", "CASE"]
    # BERT cleaning: control / zero-width characters and U+FFFD vanish, CJK ideographs stand alone
    assert basic_tokenize("x\u200by \x00a\ufffdb") == ["xy", "ab"]
    assert basic_tokenize("a中文b") == ["a", "中", "文", "b"]


def test_hash_tokenizer_shape_truncation_determinism():
    t = HashTokenizer(30527)
    ids = t.encode("python error handling with try except", 384)
    assert ids[0] == BOS and ids[-1] == EOS and len(ids) == 8
    assert all(4 <= i < 30527 - 1 for i in ids[1:-1])
    assert ids == t.encode("Python  ERROR handling with try   except", 384)   # case / spacing insensitive
    long = t.encode("word " * 1000, 384)
    assert len(long) == 384 and long[0] == BOS and long[-1] == EOS              # src/embeddings.py:97 truncation
    assert t.encode("", 384) == [BOS, EOS]


def test_wordpiece_greedy_longest_match(tmp_path):
    vocab = ["<s>", "<pad>", "</s>", "<unk>", "un", "##aff", "##able", "aff", "python", "##s", ",", "the"]
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(vocab) + "\n")
    t = WordPieceTokenizer(str(p))
    v = {w: i for i, w in enumerate(vocab)}
    assert t.encode("unaffable", 16) == [0, v["un"], v["##aff"], v["##able"], 2]
    assert t.encode("Pythons, the xyz", 16) == [0, v["python"], v["##s"], v[","], v["the"], v["<unk>"], 2]
    assert t.encode("the " * 50, 8) == [0] + [v["the"]] * 6 + [2]


def _synthetic_vocab(tmp_path, rng):
    words = ["".join(rng.choice(string.ascii_lowercase) for _ in range(rng.randint(1, 8))) for _ in range(800)]
    pieces = ["<s>", "<pad>", "</s>", "<unk>", "[UNK]", "<mask>"] + sorted(set(words))
    pieces += ["##" + "".join(rng.choice(string.ascii_lowercase) for _ in range(rng.randint(1, 3))) for _ in range(600)]
    pieces += list(string.punctuation) + list(string.digits) + ["##" + d for d in string.digits]
    pieces += ["é", "中", "##中", "ü", "naive", "cafe", "…", "“", "ss", "i"]
    pieces = list(dict.fromkeys(pieces))
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(pieces) + "\n", encoding="utf-8")
    return str(p), words


def _random_text(rng, words):
    parts = []
    for _ in range(rng.randint(1, 40)):
        r = rng.random()
        if r < 0.6:
            parts.append(rng.choice(words))
        elif r < 0.7:
            parts.append(rng.choice(words).upper())
        elif r < 0.8:
            parts.append(rng.choice(words) + rng.choice(words))
        elif r < 0.85:
            parts.append(rng.choice(["café", "naïve", "Ünï", "中文字", "a中b", "…", "“q”", "x\u200by", "tab\there",
                                     "\x00nul", "ＡＢ", "ß", "İ", "\ufffd", "a\u00adb", "x" * 120]))
        elif r < 0.95:
            parts.append(rng.choice(string.punctuation) + rng.choice(words))
        else:
            parts.append(str(rng.randint(0, 99999)))
    return rng.choice([" ", "  ", "\n"]).join(parts)


def test_wordpiece_pipeline_matches_transformers_mpnet_tokenizer(tmp_path):
    mpnet = pytest.importorskip("transformers.models.mpnet.tokenization_mpnet")
    rng = random.Random(1)
    vocab_path, words = _synthetic_vocab(tmp_path, rng)
    mine = WordPieceTokenizer(vocab_path)
    fast = make_wordpiece(vocab_path)
    hf = mpnet.MPNetTokenizer(vocab=dict(mine.vocab))
    texts = [_random_text(rng, words) for _ in range(600)] + ["", " ", "word " * 500]
    want = [hf(t, truncation=True, max_length=64)["input_ids"] for t in texts]
    assert [mine.encode(t, 64) for t in texts] == want
    assert FastWordPieceTokenizer(vocab_path).encode_batch(texts, 64) == want   # HF tokenizers is part of this image
    # the C++ front end of libcss_hip.so (ASCII texts native, the rest through the Unicode-complete path)
    from claude_semantic_search_amd.tokenizer import NativeWordPieceTokenizer

    assert isinstance(fast, NativeWordPieceTokenizer)
    ascii_only = ["".join(ch for ch in t if ord(ch) < 128) for t in texts] + ["a\x0bb\x7fc  D.E\tf\r\ng", "x" * 101, "x" * 100]
    got = fast.encode_batch(texts + ascii_only, 64)
    assert [g.tolist() for g in got] == want + [hf(t, truncation=True, max_length=64)["input_ids"] for t in ascii_only]
    assert fast.encode("Hello, World", 8) == hf("Hello, World", truncation=True, max_length=8)["input_ids"]


def test_native_tokenizer_follows_the_python_rules_for_every_code_point(tmp_path):
    """The C++ tables (tools/gen_unicode_tables.py) against the Python implementation they were generated from:
    every BMP code point and a sample of the astral planes, embedded in a word and standing alone."""
    from claude_semantic_search_amd.tokenizer import NativeWordPieceTokenizer

    rng = random.Random(3)
    vocab_path, words = _synthetic_vocab(tmp_path, rng)
    extra = ["ab", "cd", "##cd", "##ab", "σ", "ς", "##σ", "ᄀ", "##ᅡ", "##ᆨ", "i", "ss", "ω", "a", "##a", "中", "。", "！"]
    with open(vocab_path, "a", encoding="utf-8") as f:
        f.write("\n".join(extra) + "\n")
    py = WordPieceTokenizer(vocab_path)
    nat = NativeWordPieceTokenizer(vocab_path)
    cps = [c for c in list(range(0x80, 0x10000)) + list(range(0x10000, 0x110000, 37)) if not 0xD800 <= c <= 0xDFFF]
    texts = ["ab" + chr(c) + "cd " + chr(c) for c in cps]
    got = nat.encode_batch(texts, 16)
    assert nat.last_fallbacks == 1                      # U+03A3 only (final-sigma rule stays in Python)
    bad = [hex(c) for c, t, g in zip(cps, texts, got) if g.tolist() != py.encode(t, 16)]
    assert not bad, bad[:20]
    # mixed sentences, truncation, long words, invalid UTF-8 (lone surrogate) through the fallback
    sents = [_random_text(rng, words) + rng.choice(["", " 가각 한국어", " Ünïcödé Σίσυφος", " 日本語のテキスト", " a\u0301b e\u0308"]) for _ in range(400)]
    sents += ["\ud800 lone surrogate", "x" * 99 + "é" * 2, "é" * 101, ""]
    got = nat.encode_batch(sents, 48)
    assert [g.tolist() for g in got] == [py.encode(t, 48) for t in sents]
    assert nat.last_fallbacks >= 1
