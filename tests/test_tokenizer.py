"""CPU: host-side tokenisation front end (self-made cases; parity with the real
all-mpnet-base-v2 tokenizer is unpinned because no vocabulary exists offline)."""
from claude_semantic_search_amd.tokenizer import BOS, EOS, HashTokenizer, WordPieceTokenizer, basic_tokenize


def test_basic_tokenize_lowercase_accents_punct():
    assert basic_tokenize("Hello, Wörld! x=1") == ["hello", ",", "world", "!", "x", "=", "1"]
    assert basic_tokenize("  \t\n ") == []
    assert basic_tokenize("don't") == ["don", "'", "t"]
    assert basic_tokenize("Keep CASE", lower=False) == ["Keep", "CASE"]


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
