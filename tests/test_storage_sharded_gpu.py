"""GPU: ``HybridStorage`` over a shard group through libcss_hip.so -- two ranks sharing cuda:0 (gloo moves the packed
top-k records; RCCL refuses two ranks on one device), every case of tests/storage_cases.py on both ranks, plus the
checks of tests/test_storage_sharded_gloo.py on the merged answers (``src/storage.py:408-492`` behind > 1 shard)."""
import socket

import pytest

pytestmark = pytest.mark.gpu


def test_storage_cases_on_two_shards_of_one_gpu(tmp_path):
    import torch.multiprocessing as mp

    from test_storage_sharded_gloo import check_outputs, run_cases

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(run_cases, args=(2, port, str(tmp_path), False), nprocs=2, join=True)
    check_outputs(tmp_path)
