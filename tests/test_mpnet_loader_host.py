"""CPU: the host half of the real-checkpoint path (directory discovery and tensor files) of MpnetEncoder --
what the reference reaches through SentenceTransformer(name, cache_folder=...) (src/embeddings.py:86-88)."""
import json

import numpy as np
import pytest

from claude_semantic_search_amd import mpnet_encoder as me


def _fake_dir(root, sub=False, fmt="safetensors"):
    d = root / "0_Transformer" if sub else root
    d.mkdir(parents=True, exist_ok=True)
    (d / "config.json").write_text(json.dumps({"num_hidden_layers": 1}))
    sd = {"embeddings.LayerNorm.weight": np.arange(4, dtype=np.float32),
          "encoder.layer.0.attention.attn.q.bias": np.ones(3, dtype=np.float16)}
    if fmt == "safetensors":
        from safetensors.numpy import save_file

        save_file(sd, str(d / "model.safetensors"))
    else:
        import torch

        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, str(d / "pytorch_model.bin"))
    return sd


@pytest.mark.parametrize("sub", [False, True])
@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_state_dict_files_are_read_as_fp32(tmp_path, sub, fmt):
    sd = _fake_dir(tmp_path / "m", sub, fmt)
    got = me._load_state_dict(tmp_path / "m")
    assert set(got) == set(sd)
    for k in sd:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], sd[k].astype(np.float32))


def test_model_directory_discovery(tmp_path, monkeypatch):
    _fake_dir(tmp_path / "direct")
    _fake_dir(tmp_path / "cache" / "all-mpnet-base-v2")
    _fake_dir(tmp_path / "cache2" / "sentence-transformers_all-mpnet-base-v2", sub=True)
    _fake_dir(tmp_path / "home" / "all-mpnet-base-v2")
    monkeypatch.delenv("SENTENCE_TRANSFORMERS_HOME", raising=False)
    assert me._find_model_dir(str(tmp_path / "direct"), None) == tmp_path / "direct"
    assert me._find_model_dir("all-mpnet-base-v2", str(tmp_path / "cache")) == tmp_path / "cache" / "all-mpnet-base-v2"
    assert me._find_model_dir("all-mpnet-base-v2", str(tmp_path / "cache2")) == \
        tmp_path / "cache2" / "sentence-transformers_all-mpnet-base-v2"
    assert me._find_model_dir("all-mpnet-base-v2", None) is None
    # the reference sets SENTENCE_TRANSFORMERS_HOME as a side effect of cache_dir (src/embeddings.py:81-83)
    monkeypatch.setenv("SENTENCE_TRANSFORMERS_HOME", str(tmp_path / "home"))
    assert me._find_model_dir("all-mpnet-base-v2", None) == tmp_path / "home" / "all-mpnet-base-v2"
    assert me._find_model_dir("no-such-model", str(tmp_path / "cache")) is None
    (tmp_path / "empty").mkdir()
    assert me._find_model_dir(str(tmp_path / "empty"), None) is None
    with pytest.raises(FileNotFoundError):
        me._load_state_dict(tmp_path / "empty")


def test_hf_hub_cache_layout_is_found(tmp_path, monkeypatch):
    """sentence-transformers >= 3 (the reference pins >= 5) downloads into the HF-hub layout under cache_folder:
    models--sentence-transformers--<name>/snapshots/<rev>/ (src/embeddings.py:81-88 passes cache_folder and sets
    SENTENCE_TRANSFORMERS_HOME)."""
    monkeypatch.delenv("SENTENCE_TRANSFORMERS_HOME", raising=False)
    monkeypatch.delenv("HF_HOME", raising=False)
    monkeypatch.delenv("HF_HUB_CACHE", raising=False)
    monkeypatch.setenv("HOME", str(tmp_path / "nohome"))
    repo = tmp_path / "cache" / "models--sentence-transformers--all-mpnet-base-v2"
    old, new = repo / "snapshots" / "aaaa", repo / "snapshots" / "bbbb"
    _fake_dir(old)
    _fake_dir(new)
    (repo / "refs").mkdir()
    (repo / "refs" / "main").write_text("bbbb\n")
    for name in ("all-mpnet-base-v2", "sentence-transformers/all-mpnet-base-v2"):
        assert me._find_model_dir(name, str(tmp_path / "cache")) == new
    (repo / "refs" / "main").unlink()                      # no ref file: the newest snapshot
    import os
    os.utime(old, (2_000_000_000, 2_000_000_000))
    assert me._find_model_dir("all-mpnet-base-v2", str(tmp_path / "cache")) == old
    # another organisation: models--<org>--<name>
    _fake_dir(tmp_path / "cache" / "models--acme--tiny-mpnet" / "snapshots" / "r1")
    assert me._find_model_dir("acme/tiny-mpnet", str(tmp_path / "cache")).name == "r1"
    assert me._find_model_dir("acme/other", str(tmp_path / "cache")) is None
    # through the environment the reference sets, and through HF_HOME
    monkeypatch.setenv("SENTENCE_TRANSFORMERS_HOME", str(tmp_path / "cache"))
    assert me._find_model_dir("all-mpnet-base-v2", None) == old
    monkeypatch.delenv("SENTENCE_TRANSFORMERS_HOME")
    hub = tmp_path / "hf" / "hub" / "models--sentence-transformers--all-mpnet-base-v2" / "snapshots" / "zz"
    _fake_dir(hub)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    assert me._find_model_dir("all-mpnet-base-v2", None) == hub


def test_tokenizer_files_root_level_vocab_and_lowercase_flag(tmp_path):
    """Modern sentence-transformers directories keep vocab.txt / tokenizer_config.json at the root while older ones keep
    them with the weights in 0_Transformer/; do_lower_case is the TOKENIZER's flag."""
    d = tmp_path / "m"
    _fake_dir(d, sub=True)
    assert me._tokenizer_files(d) == {"vocab": None, "lower": True, "max_seq_length": None}
    (d / "0_Transformer" / "vocab.txt").write_text("<s>\n<pad>\n</s>\n<unk>\n")
    assert me._tokenizer_files(d)["vocab"] == d / "0_Transformer" / "vocab.txt"
    (d / "vocab.txt").write_text("<s>\n<pad>\n</s>\n<unk>\nroot\n")       # the root wins when both exist
    (d / "tokenizer_config.json").write_text(json.dumps({"do_lower_case": False}))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 256, "do_lower_case": True}))
    got = me._tokenizer_files(d)
    assert got == {"vocab": d / "vocab.txt", "lower": False, "max_seq_length": 256}
