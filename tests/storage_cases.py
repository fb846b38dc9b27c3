"""HybridStorage behaviour cases, modelled on the reference's tests/test_storage.py,
tests/test_integration.py, tests/test_project_filter.py and
tests/test_incremental_indexing.py (cited per test).  The same cases run
(a) on CPU with an oracle-backed test double standing in for the device index
(host logic only) and (b) on the GPU through libcss_hip.so."""
import os
import shutil
import tempfile
from pathlib import Path

import numpy as np
import pytest

from claude_semantic_search_amd.chunk import Chunk
from claude_semantic_search_amd.storage import HybridStorage, SearchConfig, SearchResult, StorageConfig


def _chunks():
    # fixture of the reference's tests/test_storage.py:106-156
    def md(session, project, ctype, ts, code, tools, msgs, chars, words):
        return {"session_id": session, "project_name": project, "chunk_type": ctype, "timestamp": ts,
                "has_code": code, "has_tools": tools, "message_count": msgs, "char_count": chars, "word_count": words}

    return [
        Chunk("chunk_001", "This is about machine learning and AI.",
              md("session_1", "test_project", "qa_pair", "2024-01-15T10:00:00", False, False, 2, 38, 8), [0.1, 0.2, 0.3, 0.4]),
        Chunk("chunk_002", "Python programming and data science topics.",
              md("session_1", "test_project", "code_block", "2024-01-15T10:01:00", True, True, 1, 43, 6), [0.5, 0.6, 0.7, 0.8]),
        Chunk("chunk_003", "Natural language processing techniques.",
              md("session_2", "other_project", "tool_usage", "2024-01-15T11:00:00", False, True, 3, 37, 4), [0.9, 0.1, 0.2, 0.3]),
    ]


class StorageCases:
    def setup_method(self):
        self.tmp = tempfile.mkdtemp()
        self.config = StorageConfig(data_dir=self.tmp, embedding_dim=4, auto_save=False)
        self.storage = HybridStorage(self.config)
        self.chunks = _chunks()

    def teardown_method(self):
        try:
            self.storage.close()
        except Exception:
            pass
        shutil.rmtree(self.tmp, ignore_errors=True)

    # tests/test_storage.py:169-176
    def test_construction_is_lazy(self):
        assert self.storage.data_dir == Path(self.tmp)
        assert self.storage.db is None and self.storage.faiss_index is None
        assert self.storage.total_chunks == 0 and self.storage.embedding_dim == 4

    # tests/test_storage.py:178-195, :604-615
    def test_initialize_picks_ip_or_l2(self):
        self.storage.initialize()
        assert self.storage.db is not None and self.storage.db_path.exists()
        assert self.storage.faiss_index.metric_type == 0
        tables = [r[0] for r in self.storage.db.cursor().execute("SELECT name FROM sqlite_master WHERE type='table'")]
        assert "chunks" in tables and "files" in tables
        s2 = HybridStorage(StorageConfig(data_dir=self.tmp, embedding_dim=4, normalize_embeddings=False))
        s2.initialize()
        assert s2.faiss_index.metric_type == 1
        s2.close()

    # tests/test_storage.py:225-233
    def test_unknown_index_type(self):
        s = HybridStorage(StorageConfig(data_dir=self.tmp, embedding_dim=4, index_type="invalid"))
        with pytest.raises(ValueError, match="Unknown index type"):
            s.initialize()

    # tests/test_storage.py:235-275
    def test_add_chunks_counts(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        assert self.storage.faiss_index.ntotal == 3 and self.storage.total_chunks == 3
        assert self.storage.db.cursor().execute("SELECT COUNT(*) FROM chunks").fetchone()[0] == 3
        assert len(self.storage.chunk_id_to_faiss_id) == 3 and len(self.storage.faiss_id_to_chunk_id) == 3
        self.storage.add_chunks([Chunk("no_emb", "Test", {})])
        self.storage.add_chunks([])
        assert self.storage.faiss_index.ntotal == 3

    def test_add_before_initialize_raises(self):
        with pytest.raises(RuntimeError, match="FAISS index not initialized"):
            self.storage.add_chunks(self.chunks)

    # tests/test_storage.py:277-308 + golden G1
    def test_search_basic_and_config(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        res = self.storage.search(np.array([0.1, 0.2, 0.3, 0.4]))
        assert all(isinstance(r, SearchResult) for r in res)
        assert [r.chunk_id for r in res] == ["chunk_001", "chunk_002", "chunk_003"]
        assert np.allclose([r.similarity for r in res], [0.9999999, 0.9688640, 0.5432198], atol=1e-6)
        assert res[0].similarity > 0.8
        cfg = SearchConfig(top_k=2, similarity_threshold=0.5, include_metadata=True, include_text=True)
        res = self.storage.search(np.array([0.1, 0.2, 0.3, 0.4]), cfg)
        assert len(res) <= 2 and all(r.similarity >= 0.5 for r in res)
        assert all(r.metadata is not None and r.text is not None and r.chunk is not None for r in res)
        res = self.storage.search(np.array([0.1, 0.2, 0.3, 0.4]), SearchConfig(include_metadata=False, include_text=False))
        assert all(r.metadata is None and r.text is None and r.chunk is None for r in res)

    # tests/test_storage.py:310-345
    def test_search_filters(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        q = np.array([0.1, 0.2, 0.3, 0.4])
        r = self.storage.search(q, filters={"project_name": "test_project"})
        assert len(r) == 2 and all(x.metadata["project_name"] == "test_project" for x in r)
        assert len(self.storage.search(q, filters={"word_count": {"gte": 5}})) == 2
        assert len(self.storage.search(q, filters={"chunk_type": ["qa_pair", "code_block"]})) == 2
        assert len(self.storage.search(q, filters={"project_name": "TEST_PROJ"})) == 2  # case-insensitive substring
        assert len(self.storage.search(q, filters={"unknown_key": 1})) == 3             # unknown keys ignored

    # tests/test_storage.py:617-647
    def test_matches_filters_truth_table(self):
        self.storage.initialize()
        d = {"project_name": "test_project", "word_count": 10, "has_code": True, "chunk_type": "qa_pair"}
        m = self.storage._matches_filters
        assert m(d, {"project_name": "test_project"}) and not m(d, {"project_name": "other_project"})
        assert m(d, {"word_count": {"gte": 5}}) and m(d, {"word_count": {"lte": 15}})
        assert not m(d, {"word_count": {"gt": 10}}) and not m(d, {"word_count": {"lt": 10}})
        assert m(d, {"chunk_type": ["qa_pair", "code_block"]}) and not m(d, {"chunk_type": ["tool_usage"]})
        assert m(d, {"has_code": True}) and not m(d, {"has_code": False})

    # tests/test_storage.py:693-700
    def test_search_empty_and_uninitialised(self):
        assert self.storage.search(np.array([0.1, 0.2, 0.3, 0.4])) == []
        self.storage.initialize()
        assert self.storage.search(np.array([0.1, 0.2, 0.3, 0.4])) == []

    # tests/test_integration.py:141-226 (query passed as a plain list, :203-204)
    def test_integration_workflow_list_query(self):
        def c(i, text, sess, ctype, code, tools, emb):
            return Chunk(i, text, {"session_id": sess, "project_name": "test-project", "chunk_type": ctype,
                                   "has_code": code, "has_tools": tools}, emb)

        chunks = [c("chunk_1", "Python programming basics and variables", "test-session", "qa_pair", False, False, [0.1, 0.2, 0.3, 0.4]),
                  c("chunk_2", "Error handling in Python with try-except blocks", "test-session", "qa_pair", True, False, [0.2, 0.3, 0.4, 0.5]),
                  c("chunk_3", "Database connections and SQL queries", "test-session-2", "context_segment", True, True, [0.3, 0.4, 0.5, 0.6])]
        self.storage.initialize()
        self.storage.add_chunks(chunks)
        st = self.storage.get_stats()
        assert (st["total_chunks"], st["total_sessions"], st["total_projects"], st["embedding_dimension"]) == (3, 2, 1, 4)
        cfg = SearchConfig(top_k=3)
        q = [0.15, 0.25, 0.35, 0.45]
        res = self.storage.search(q, cfg)
        assert [r.chunk_id for r in res] == ["chunk_2", "chunk_1", "chunk_3"]
        assert np.allclose([r.similarity for r in res], [0.9988701, 0.9979654, 0.9935983], atol=1e-6)
        assert all(r.similarity > 0 and r.text and r.metadata for r in res)
        code = self.storage.search(q, cfg, {"has_code": True})
        assert len(code) == 2 and all(r.metadata.get("has_code") for r in code)
        assert len(self.storage.search(q, cfg, {"project_name": "test-project"})) == 3

    # tests/test_integration.py:312-353
    def test_search_relevance_ordering(self):
        chunks = [Chunk("highly_relevant", "a", {"chunk_type": "qa_pair"}, [1.0, 0.9, 0.8, 0.7]),
                  Chunk("somewhat_relevant", "b", {"chunk_type": "qa_pair"}, [0.8, 0.7, 0.6, 0.5]),
                  Chunk("less_relevant", "c", {"chunk_type": "qa_pair"}, [0.2, 0.3, 0.4, 0.5])]
        self.storage.initialize()
        self.storage.add_chunks(chunks)
        res = self.storage.search([1.0, 0.9, 0.8, 0.7], SearchConfig(top_k=3))
        assert [r.chunk_id for r in res] == ["highly_relevant", "somewhat_relevant", "less_relevant"]
        assert res[0].similarity > res[1].similarity > res[2].similarity and res[0].similarity > 0.9
        assert np.allclose([r.similarity for r in res], [1.0, 0.9992177, 0.9047619], atol=1e-6)

    # tests/test_storage.py:541-558 (save/load round trip) and :679-691 (auto save)
    def test_save_load_roundtrip_and_incremental_append(self):
        cfg = StorageConfig(data_dir=self.tmp, embedding_dim=4, auto_save=True)
        s = HybridStorage(cfg)
        s.initialize()
        s.add_chunks(self.chunks[:2])
        assert s.index_path.exists()
        size2 = s.index_path.stat().st_size
        s.add_chunks(self.chunks[2:])           # appended, header patched
        assert s.index_path.stat().st_size == size2 + 4 * 4
        s.close()
        s2 = HybridStorage(cfg)
        s2.initialize()
        assert s2.faiss_index.ntotal == 3 and s2.total_chunks == 3
        res = s2.search(np.array([0.1, 0.2, 0.3, 0.4]))
        assert [r.chunk_id for r in res] == ["chunk_001", "chunk_002", "chunk_003"]
        s2.close()

    def test_corrupt_index_file_gives_fresh_index(self):
        (Path(self.tmp) / "embeddings.faiss").write_bytes(b"garbage")
        self.storage.initialize()  # src/storage.py:314-316: warning + new empty index
        assert self.storage.faiss_index.ntotal == 0

    # tests/test_storage.py:560-593
    def test_backup_and_restore(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        bdir = os.path.join(self.tmp, "backup")
        self.storage.backup(bdir)
        assert (Path(bdir) / self.config.index_name).exists() and (Path(bdir) / self.config.db_name).exists()
        self.storage.close()
        for p in (self.storage.index_path, self.storage.db_path):
            if p.exists():
                p.unlink()
        s = HybridStorage(self.config)
        s.initialize()
        s.restore(bdir)
        assert s.total_chunks == 3 and s.get_chunk_by_id("chunk_001") is not None
        assert s.faiss_index.ntotal == 3
        s.close()

    # tests/test_storage.py:347-428 (delete / sessions), tombstone semantics src/storage.py:449-451
    def test_delete_leaves_tombstone_and_optimize_compacts(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        assert self.storage.delete_chunk("chunk_001") and not self.storage.delete_chunk("chunk_001")
        assert self.storage.total_chunks == 2 and self.storage.faiss_index.ntotal == 3
        res = self.storage.search(np.array([0.1, 0.2, 0.3, 0.4]))
        assert [r.chunk_id for r in res] == ["chunk_002", "chunk_003"]
        self.storage.optimize()
        assert self.storage.faiss_index.ntotal == 2 and self.storage.total_chunks == 2
        res = self.storage.search(np.array([0.1, 0.2, 0.3, 0.4]))
        assert [r.chunk_id for r in res] == ["chunk_002", "chunk_003"]
        assert self.storage.delete_chunks_by_session("session_2") == 1
        assert [c.id for c in self.storage.get_chunks_by_session("session_1")] == ["chunk_002"]
        assert self.storage.get_chunks_by_project("other_project") == []

    # tests/test_storage.py:430-445 (stats keys)
    def test_stats_keys_and_projects(self):
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        st = self.storage.get_stats()
        for key in ("total_chunks", "total_sessions", "total_projects", "projects", "chunk_types", "faiss_index_size",
                    "database_size", "total_storage_size", "embedding_dimension", "index_type", "use_gpu", "is_gpu_index"):
            assert key in st
        assert st["projects"] == ["other_project", "test_project"] and st["chunk_types"]["qa_pair"] == 1
        with pytest.raises(RuntimeError, match="Database not initialized"):
            HybridStorage(self.config).get_all_projects()

    # tests/test_incremental_indexing.py (file tracking + clear), d = 768 constant vectors incl. all-zero row
    def test_incremental_file_tracking_768(self):
        cfg = StorageConfig(data_dir=self.tmp, embedding_dim=768, auto_save=False, db_name="inc.db", index_name="inc.faiss")
        s = HybridStorage(cfg)
        s.initialize()
        f = os.path.join(self.tmp, "conv.jsonl")
        Path(f).write_text("{}")
        assert s.is_file_modified(f)
        chunks = [Chunk(f"c{i}", f"text {i}", {"file_path": f, "project_name": "p"}, [0.1 * i] * 768) for i in range(4)]
        s.add_chunks(chunks)
        s.update_file_info(f, 4)
        assert not s.is_file_modified(f) and s.faiss_index.ntotal == 4
        assert s.remove_chunks_for_file(f) == 4 and s.total_chunks == 4  # reference does not decrement here
        assert s.search(np.ones(768, np.float32)) == []                 # all tombstones
        s.clear_all_data()
        assert s.faiss_index.ntotal == 0 and s.total_chunks == 0 and s.is_file_modified(f)
        s.close()

    # tests/test_project_filter.py:68-132 (768-d, counts only)
    def test_project_filter_768(self):
        cfg = StorageConfig(data_dir=self.tmp, embedding_dim=768, auto_save=False, db_name="pf.db", index_name="pf.faiss")
        s = HybridStorage(cfg)
        s.initialize()
        rng = np.random.default_rng(0)
        chunks = []
        for i in range(9):
            proj = ["daisy-hft-engine", "claude-semantic-search", "other"][i % 3]
            chunks.append(Chunk(f"chunk_{i}", f"text {i}", {"project_name": proj, "session_id": f"s{i}"},
                                rng.random(768).astype(np.float32).tolist()))
        s.add_chunks(chunks)
        q = rng.random(768).astype(np.float32)
        assert len(s.search(q, SearchConfig(top_k=10))) == 9
        assert len(s.search(q, SearchConfig(top_k=10), {"project_name": "daisy"})) == 3
        assert len(s.search(q, SearchConfig(top_k=10), {"project_name": "nope"})) == 0
        assert len(s.search(q, SearchConfig(top_k=2), {"project_name": "SEMANTIC"})) == 2
        s.close()

    # SURVEY 8f rank 2 (extension, opt-in): filters / tombstones pushed down into the kernel as an allow-bitmap.
    # 300 chunks, the 6 chunks of project "rare" are the WORST matches of the query: the reference's over-fetch of
    # max_results=100 unfiltered hits never sees them (``src/storage.py:429-492``), the push-down returns them.
    def test_filter_pushdown_returns_true_filtered_topk(self):
        rng = np.random.default_rng(3)
        q = rng.standard_normal(768).astype(np.float32)
        chunks = []
        for i in range(300):
            v = rng.standard_normal(768).astype(np.float32)
            rare = i % 50 == 7
            v = v + (-3.0 if rare else 1.5) * q        # rare rows point away from the query
            chunks.append(Chunk(f"c{i}", f"text {i}", {"project_name": "rare" if rare else "common", "session_id": f"s{i}",
                                                       "message_count": i}, v.tolist()))
        results = {}
        for pushdown in (False, True):
            cfg = StorageConfig(data_dir=self.tmp, embedding_dim=768, auto_save=False, db_name=f"pd{int(pushdown)}.db",
                                index_name=f"pd{int(pushdown)}.faiss", filter_pushdown=pushdown)
            s = HybridStorage(cfg)
            s.initialize()
            s.add_chunks(chunks)
            sc = SearchConfig(top_k=5, similarity_threshold=-1.0)
            results[pushdown] = [r.chunk_id for r in s.search(q, sc, {"project_name": "rare"})]
            # unfiltered and satisfiable searches are identical in both modes
            assert [r.chunk_id for r in s.search(q, sc)] == [r.chunk_id for r in s.search(q, sc, {"project_name": "common"})]
            if pushdown:
                rng_ids = [r.chunk_id for r in s.search(q, sc, {"message_count": {"gte": 100, "lt": 110}})]
                assert len(rng_ids) == 5 and all(100 <= int(c[1:]) < 110 for c in rng_ids)
                assert s.delete_chunk(results[True][0])            # tombstone: never returned again, slot not wasted
                again = [r.chunk_id for r in s.search(q, sc, {"project_name": "rare"})]
                assert results[True][0] not in again and len(again) == 5
            s.close()
        assert results[False] == []                                # starved inside the first 100 unfiltered hits
        assert len(results[True]) == 5 and all(int(c[1:]) % 50 == 7 for c in results[True])

    def test_compaction_is_on_disk_before_the_new_ids_are_committed(self):
        """optimize() renumbers faiss ids in SQLite; the compacted rows must already be in the index file then, also
        with auto_save=False and without close(): a process that dies right after must find ids and rows that match."""
        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        self.storage.save_index()
        assert self.storage.delete_chunk("chunk_001")
        self.storage.optimize()                                # no save_index(), no close() afterwards: "crash"
        other = HybridStorage(self.config)                     # a second process opening the same data_dir
        other.initialize()
        assert other.faiss_index.ntotal == 2 and other.total_chunks == 2
        res = other.search(np.array([0.9, 0.1, 0.2, 0.3]))
        assert res[0].chunk_id == "chunk_003" and abs(res[0].similarity - 1.0) < 1e-5
        assert [r.chunk_id for r in other.search(np.array([0.5, 0.6, 0.7, 0.8]))][0] == "chunk_002"
        assert not (self.storage.index_path.parent / (self.storage.index_path.name + ".tmp")).exists()
        other.close()

    def test_compaction_survives_a_crash_between_its_steps(self):
        """optimize() changes two files (index rows, SQLite ids).  The step is journaled; a crash (a) after the
        compacted file is written but before the ids are committed, and (b) after the commit but before the file is
        moved into place, must both leave a data_dir in which ids and rows match at the next initialize() (ADVICE r2:
        the old order paired compacted rows with stale sparse ids)."""
        class Crash(Exception):
            pass

        q3, q2 = np.array([0.9, 0.1, 0.2, 0.3]), np.array([0.5, 0.6, 0.7, 0.8])

        def check(n_rows):
            other = HybridStorage(self.config)
            other.initialize()
            assert other.total_chunks == 2 and other.faiss_index.ntotal == n_rows
            r3, r2 = other.search(q3), other.search(q2)
            assert r3[0].chunk_id == "chunk_003" and abs(r3[0].similarity - 1.0) < 1e-5
            assert r2[0].chunk_id == "chunk_002" and abs(r2[0].similarity - 1.0) < 1e-5
            assert "chunk_001" not in [r.chunk_id for r in r3 + r2]
            assert not Path(str(other.index_path) + ".compact").exists()
            assert other.db.execute("SELECT COUNT(*) FROM storage_meta WHERE key = 'pending_compact'").fetchone()[0] == 0
            other.close()

        self.storage.initialize()
        self.storage.add_chunks(self.chunks)
        self.storage.save_index()
        assert self.storage.delete_chunk("chunk_001")            # a tombstone in FRONT: every later id moves
        # (a) crash before the transaction: simulated by making the id update fail after the file was written
        real_write = HybridStorage._write_index_atomically

        def write_then_die(ix, path):
            real_write(ix, path)
            if path.endswith(".compact"):
                raise Crash()

        self.storage._write_index_atomically = write_then_die
        with pytest.raises(Crash):
            self.storage.optimize()
        self.storage.db.rollback()
        assert Path(str(self.storage.index_path) + ".compact").exists()
        check(3)                                                   # old file + old ids; the leftover is removed
        # (b) crash after the commit, before the rename
        del self.storage._write_index_atomically

        def die(where):
            raise Crash(where)

        self.storage._crash_point = die
        with pytest.raises(Crash):
            self.storage.optimize()
        assert Path(str(self.storage.index_path) + ".compact").exists()
        check(2)                                                   # recovery moved the compacted file into place
        check(2)                                                   # and the repaired state is stable

    def test_appended_rows_are_durable_before_the_header_counts_them(self):
        """save_index() appends new rows first and patches the header afterwards: a file cut off right behind the old
        rows + a stale header still loads as the old, complete index (trailing bytes are ignored)."""
        import struct

        self.storage.initialize()
        self.storage.add_chunks(self.chunks[:2])
        self.storage.save_index()
        before = self.storage.index_path.read_bytes()
        self.storage.add_chunks(self.chunks[2:])
        self.storage.save_index()                              # append path
        after = self.storage.index_path.read_bytes()
        assert len(after) == len(before) + 4 * 4 and after[45:len(before)] == before[45:]
        assert struct.unpack("<q", after[8:16])[0] == 3 and struct.unpack("<Q", after[37:45])[0] == 12
        # what a crash between "rows durable" and "header patched" leaves behind: old header + appended rows
        torn = before[:45] + after[45:]
        self.storage.index_path.write_bytes(torn)
        s2 = HybridStorage(self.config)
        s2.initialize()
        assert s2.faiss_index.ntotal == 2
        s2.close()

    def test_allow_mask_cache_is_dropped_when_the_chunk_set_changes(self):
        """Filter push-down caches one allow mask per filter; clearing and re-adding the SAME NUMBER of chunks with other
        metadata must not reuse the old mask."""
        cfg = StorageConfig(data_dir=self.tmp, embedding_dim=4, auto_save=False, db_name="ac.db", index_name="ac.faiss",
                            filter_pushdown=True)
        s = HybridStorage(cfg)
        s.initialize()
        s.add_chunks(self.chunks)
        sc = SearchConfig(top_k=3, similarity_threshold=-1.0)
        q = np.array([0.1, 0.2, 0.3, 0.4])
        assert [r.chunk_id for r in s.search(q, sc, {"project_name": "other_project"})] == ["chunk_003"]
        s.clear_all_data()
        swapped = _chunks()
        swapped[0].metadata["project_name"], swapped[2].metadata["project_name"] = "other_project", "test_project"
        s.add_chunks(swapped)                                  # same ids, same count, other projects
        assert [r.chunk_id for r in s.search(q, sc, {"project_name": "other_project"})] == ["chunk_001"]
        assert s.delete_chunk("chunk_001")
        assert s.search(q, sc, {"project_name": "other_project"}) == []
        s.close()

    def test_context_manager(self):
        with HybridStorage(self.config) as s:
            s.add_chunks(self.chunks)
            assert s.total_chunks == 3
