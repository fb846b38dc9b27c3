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
