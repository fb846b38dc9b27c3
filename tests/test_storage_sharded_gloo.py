"""CPU, world_size 2 over gloo: ``HybridStorage`` over a shard group (``StorageConfig.sharded`` /
``CSS_STORAGE_SHARDED=1``) behaves like ``HybridStorage`` over one index.

Every behaviour case of tests/storage_cases.py (the reference's own storage / integration / filter / incremental
tests, cited there) runs SPMD on both ranks -- same calls, same arguments, each rank its own data_dir -- with the
device index of every shard replaced by the oracle-backed double of tests/test_storage_host.py.  That covers, through
the shard group: add (routed to the least-full shard), search incl. filters, tombstones and the allow-mask push-down
(``src/storage.py:408-492``), deletes, index files written from / read back into shards, backup / restore, the
journaled compaction and its crash windows.  ``tests/test_storage_sharded_gpu.py`` runs the same on two ranks
sharing one GPU through libcss_hip.so.
"""
import os
import socket
import traceback

import numpy as np
import torch.multiprocessing as mp


def run_cases(rank, world, port, out_dir, use_fakes):
    """Rank body shared with the GPU twin: every StorageCases test, setup / teardown around each, in name order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CSS_STORAGE_SHARDED"] = "1"
    import datetime

    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    failures, ran = [], []
    try:
        from claude_semantic_search_amd import flat_index as fi
        from claude_semantic_search_amd import sharded
        from storage_cases import StorageCases

        if use_fakes:
            from oracle import knn_oracle as ko
            from test_storage_host import _FakeIndex

            class _FakeShard(_FakeIndex):
                base = 0

                def set_id_base(self, b):
                    self.base = int(b)

                def search(self, q, k, normalize=False, allow=None):
                    D, I = super().search(q, k, normalize=normalize, allow=allow)
                    return D, np.where(I >= 0, I + self.base, -1)

            orig_init = sharded.ShardedFlatIndex.__init__

            def init(self, d, metric=0, **kw):        # (the merge double needs the metric of the index it serves)
                def m(Dg, Ig, k):
                    D, I = ko.merge_topk(Dg.numpy(), Ig.numpy(), metric)
                    return torch.from_numpy(D), torch.from_numpy(I)
                kw["merge"] = m
                kw["device_index"] = None
                orig_init(self, d, metric, **kw)

            sharded.ShardedFlatIndex.__init__ = init
            fi.IndexFlat = _FakeShard
        else:
            torch.cuda.set_device(0)
        cases = StorageCases()
        for name in sorted(n for n in dir(StorageCases) if n.startswith("test_")):
            cases.setup_method()
            try:
                getattr(cases, name)()
                ran.append(name)
            except BaseException:
                failures.append((name, traceback.format_exc()))
            finally:
                cases.teardown_method()
            if failures:
                break            # the ranks run in lockstep: stop at the first failure (the peer times out on its next collective)
        # the shards really are shards: a fresh storage with a few adds has rows on both ranks, ids of one index
        import tempfile

        from claude_semantic_search_amd.chunk import Chunk
        from claude_semantic_search_amd.storage import HybridStorage, SearchConfig, StorageConfig

        if not failures:
            rng = np.random.default_rng(3)
            with tempfile.TemporaryDirectory() as tmp:
                s = HybridStorage(StorageConfig(data_dir=tmp, embedding_dim=16, auto_save=True, filter_pushdown=True))
                s.initialize()
                vecs = rng.standard_normal((40, 16)).astype(np.float32)
                for f in range(4):          # four "files" of ten chunks
                    s.add_chunks([Chunk(f"c{f}_{j}", f"text {f} {j}", {"project_name": f"p{f % 2}", "session_id": f"s{f}"},
                                        vecs[f * 10 + j]) for j in range(10)])
                sizes = list(s.faiss_index.sh.shard_sizes)
                s.delete_chunk("c2_3")
                res = s.search(vecs[23], SearchConfig(top_k=5, max_results=300), filters={"project_name": "p0"})
                ids = [r.chunk_id for r in res]
                full = s.search(vecs[7], SearchConfig(top_k=40, max_results=40))
                s.close()
                s2 = HybridStorage(StorageConfig(data_dir=tmp, embedding_dim=16, auto_save=True))
                s2.initialize()                                   # index file -> shards again
                again = [r.chunk_id for r in s2.search(vecs[7], SearchConfig(top_k=40, max_results=40))]
                sizes2 = list(s2.faiss_index.sh.shard_sizes)
                s2.close()
            np.savez(os.path.join(out_dir, f"extra{rank}.npz"), sizes=np.array(sizes), sizes2=np.array(sizes2), ids=np.array(ids),
                     full=np.array([r.chunk_id for r in full]), again=np.array(again),
                     sims=np.array([r.similarity for r in full]), vecs=vecs)
    finally:
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
            f.write(f"ran {len(ran)}\n")
            for name, tb in failures:
                f.write(f"FAILED {name}\n{tb}\n")
        dist.destroy_process_group()


def check_outputs(tmp_path, world=2):
    from storage_cases import StorageCases

    ncases = len([n for n in dir(StorageCases) if n.startswith("test_")])
    for r in range(world):
        txt = (tmp_path / f"rank{r}.txt").read_text()
        assert "FAILED" not in txt, txt
        assert txt.startswith(f"ran {ncases}\n"), txt
    a, b = (np.load(tmp_path / f"extra{r}.npz") for r in range(2))
    assert a["sizes"].tolist() == [20, 20] and a["sizes2"].tolist() == [20, 20]      # adds routed to the least-full shard; file re-split
    for key in ("ids", "full", "again", "sims"):
        assert np.array_equal(a[key], b[key])                                          # every rank holds the merged answer
    # against a plain numpy restatement of src/storage.py:424-492 over the 40 vectors
    v = a["vecs"] / (np.linalg.norm(a["vecs"], axis=1, keepdims=True) + 1e-8)
    sims = v @ v[7]
    order = [i for i in np.argsort(-sims, kind="stable") if i != 23 and sims[i] >= 0.0]   # c2_3 (row 23) was deleted; default threshold 0.0
    names = [f"c{i // 10}_{i % 10}" for i in order]
    assert a["full"].tolist() == names and a["again"].tolist() == names and len(names) > 10
    assert np.allclose(a["sims"], sims[order], atol=1e-5)
    sims = v @ v[23]
    want = [f"c{i // 10}_{i % 10}" for i in np.argsort(-sims, kind="stable") if (i // 10) % 2 == 0 and i != 23 and sims[i] >= 0.0][:5]
    assert a["ids"].tolist() == want and len(want) == 5                                # filtered top-5 with the push-down, over both shards


def test_storage_cases_on_two_shards(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(run_cases, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    check_outputs(tmp_path)
