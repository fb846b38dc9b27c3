#!/usr/bin/env python3
"""bench.py -- headline benchmark of the embed-and-search hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1 is launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``,
one rank per GPU over RCCL).  Rank 0 prints ONE JSON line.

Workload.  N = 1: BASELINE.json configs[3] (the configuration the north-star roofline target is quoted on): a
10M x 768 fp32 flat inner-product index resident in HBM, a batch of 1000 queries, exact top-10.  N > 1:
BASELINE.json configs[4] -- 10 M rows PER GPU (N = 8: the 80M x 768 index), one virtual synthetic index
row-partitioned over the ranks, the same 1000 queries on every rank: weak scaling.  A "step" is one pass of the
search path over the whole query batch through the product class (``ShardedFlatIndex.search_tensors``): local
cascade on every shard -> ONE RCCL all-gather of the packed per-shard top-k (nq*k*12 bytes per rank) -> merge.
``config.strong_10M`` carries the strong-scaling line of the fixed 10 M-row index for the same N.

The JSON line also carries:
  roofline      dominant kernel of the timed region -- the main stage of the search cascade,
                k_scan_qreg_i8<12,true> (k_scan_coarse8<false,true,..> under CSS_KNN_QREG=0) on the int8 shadow rows (int8 MFMA peak; `frac_of_bf16_peak`
                beside it; the bf16 scan where the library takes that) -- measured with HIP events on the
                launch stream inside this run (css_prof_*); `traffic` = L2-miss (fabric-side) bytes per launch
                from the newest profiles/r*_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes).
                Sub-objects (same fields): `nq1_k10` -- the single-query search, the reference's real call
                shape, with the HBM roofline of its main sweep over the bytes that sweep reads (int8 rows + row
                scales where the index keeps them, else bf16 rows; the north-star's 60 % target); `exact_fp32_mode`
                -- every score formed by fp32 fmaf chains; `encode` -- batch-256 x 384 encoder forward
                (BASELINE configs[2]) with chunks/s and its bf16-MFMA roofline (the 40 % target).
  cpu_baseline  the CPU oracle (kind "port") timed on this box's host cores on a bounded sample of the same
                workload, rank 0 / N=1 only: blocked SGEMM + heap on all cores; `encode`: the torch oracle at
                batch 16.
  extra         details: k'=100 single query, masked search, per-kernel encoder times, length-mix and
                text-path (strings in) runs, flagged fraction on clustered rows (1 M and 10 M), shadow-less index,
                BASELINE configs[4]'s 80 M rows on one GPU (`single_gpu_80M`).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

def cascade_growth(k: int) -> int:
    """Growth factor of the batched coarse cascade (css_index.hip, launch_scan_coarse): CSS_KNN_GROWTH, else by k."""
    env = os.environ.get("CSS_KNN_GROWTH", "")
    if env in ("4", "8", "16"):
        return int(env)
    return 8 if k <= 32 else 4


SWEEP_GROWTH = int(os.environ.get("CSS_KNN_GROWTH_SWEEP", "4"))         # ... of the 1..4-query sweep cascade
if SWEEP_GROWTH not in (4, 8, 16):
    SWEEP_GROWTH = 4
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3    # dense fp32-input MFMA peak
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 MFMA peak
INT8_MFMA_PEAK_TF = 5000.0   # dense int8 MFMA peak (MI355X_MICROARCH.md: the cycles of the bf16 form at twice the K)


def batch_scan_is_int8(index, k: int, rows: int, dim: int) -> bool:
    """Mirror of css_index.hip batch_i8_wanted: which shadow rows the batched candidate scan reads."""
    env = os.environ.get("CSS_KNN_SCAN", "")
    if env == "bf16" or not hasattr(index, "shadow_info") or not index.shadow_info().get("int8"):
        return False
    if dim % 256 != 0:
        return False
    return env == "i8" or (rows >= 300_000 if k <= 32 else (k <= 128 and rows >= 2_000_000))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="total index rows (all GPUs)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-heavy-extra", action="store_true", help="skip the extras that build further 10 M-row indexes (counter passes)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline work")
    ap.add_argument("--enc-batch", type=int, default=256, help="encoder batch (sequences)")
    ap.add_argument("--enc-len", type=int, default=384, help="encoder sequence length (tokens)")
    ap.add_argument("--enc-steps", type=int, default=5)
    ap.add_argument("--no-encoder", action="store_true")
    ap.add_argument("--only-encoder", action="store_true", help="development aid: run just the encoder leg")
    ap.add_argument("--enc-fixed-only", action="store_true", help="encoder leg: only the fixed batch x len shape (counter passes)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the strong-scaling line of the 10 M-row index")
    ap.add_argument("--legs", default="all", help="comma list of the extra legs to run: nq1,e2e,masked,exact,clustered,noshadow,"
                                                   "80m,clustered10m (default all; profiling passes isolate one population per kernel)")
    ap.add_argument("--no-check", action="store_true", help="skip the self-check of the timed region's answers")
    ap.add_argument("--allow-debug", action="store_true", help="print a line (marked invalid) although a *_DBG switch is set")
    a = ap.parse_args(argv)
    allowed = ("nq1", "e2e", "masked", "exact", "clustered", "noshadow", "80m", "clustered10m")
    legs = set(allowed) if a.legs == "all" else {x for x in a.legs.split(",") if x and x != "none"}
    if legs - set(allowed):
        ap.error(f"--legs: unknown {sorted(legs - set(allowed))}; choose from {allowed}")
    if a.no_extra:
        legs = set()
    if a.no_heavy_extra:
        legs -= {"noshadow", "80m", "clustered10m"}
    a.legs = legs
    return a


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box's share is far
    below os.cpu_count(); BLAS pools sized by cpu_count oversubscribe it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q_ > 0:
                n = min(n, max(1, -(-q_ // p_)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline_knn(args, log):
    """faiss-cpu's batched path restated (oracle.knn_oracle.search_blas): blocked SGEMM on every host core (the BLAS
    numpy links) + a per-query heap fold in C/OpenMP, on a bounded row sample; time scaled to the full index."""
    import numpy as np
    from oracle import knn_oracle as ko

    ncpu = usable_cpus()
    ko.set_threads(min(ncpu, 64))
    q = ko.normalize_rows(ko.synth_rows(args.nq, args.dim, 5))
    probe_rows = 100_000
    x = ko.normalize_rows(ko.synth_rows(probe_rows, args.dim, 4))
    # BLAS pool size: the fastest of a few candidates on a probe (numpy's default pool is os.cpu_count() threads, which
    # oversubscribes a box whose cgroup grants a fraction of its cores); `cores` reports the size actually used
    blas_threads, t_probe, limiter = ncpu, None, None
    try:
        from threadpoolctl import threadpool_limits
        for cand in sorted({ncpu, min(ncpu, 16), min(ncpu, 32), min(ncpu, 64)}):
            with threadpool_limits(limits=cand):
                ko.search_blas(x[:20000], q, args.k)     # warm this pool size
                t0 = time.perf_counter()
                ko.search_blas(x, q, args.k)
                tc_ = time.perf_counter() - t0
            if t_probe is None or tc_ < t_probe:
                blas_threads, t_probe = cand, tc_
        limiter = threadpool_limits(limits=blas_threads)
    except ImportError:
        ko.search_blas(x[:20000], q, args.k)
        t0 = time.perf_counter()
        ko.search_blas(x, q, args.k)
        t_probe = time.perf_counter() - t0
    rows = int(min(4_000_000, max(probe_rows, probe_rows * args.cpu_seconds / max(t_probe, 1e-3))))
    rows = min(rows, args.rows)
    x = ko.normalize_rows(ko.synth_rows(rows, args.dim, 4))
    t0 = time.perf_counter()
    ko.search_blas(x, q, args.k)
    t = time.perf_counter() - t0
    full = t * (args.rows / rows)
    gflops = 2.0 * rows * args.dim * args.nq / t / 1e9
    log(f"cpu baseline: {rows} rows x {args.nq} queries in {t:.2f}s ({gflops:.0f} GFLOP/s, BLAS threads={blas_threads})")
    # The reference's REAL call shape (src/storage.py:429-436: one query, k' = min(10 * top_k, ntotal) = 100): faiss
    # scans the rows for a single query on one thread.  One thread of the C oracle over a bounded prefix, scaled.
    ko.set_threads(1)
    rows1 = min(rows, 1_000_000)
    o1 = ko.FlatIndexOracle(args.dim, 0)
    o1._xb = x[:rows1]                                    # (no copy; the oracle only reads it)
    o1.search(q[:1], 100)
    reps1 = 3
    t0 = time.perf_counter()
    for i in range(reps1):
        o1.search(q[i:i + 1], 100)
    t1 = (time.perf_counter() - t0) / reps1
    full1 = t1 * (args.rows / rows1)
    ko.set_threads(min(ncpu, 64))
    if limiter is not None:
        limiter.restore_original_limits()
    log(f"cpu baseline, one query on one thread: {rows1} rows in {t1 * 1e3:.0f} ms -> {full1 * 1e3:.0f} ms per query at {args.rows} rows")
    return {
        "value": args.nq / full,
        "unit": "queries/s",
        "cores": int(blas_threads),
        "kind": "port",
        "achieved_GFLOPs": gflops,
        "sample": f"first {rows} of {args.rows} rows x {args.nq} q, numpy SGEMM + C heap fold, time x{args.rows / rows:.1f} (extrapolated)",
        "blas_note": f"numpy BLAS pool sized by a probe ({blas_threads} of {ncpu} usable CPUs, os.cpu_count() = {os.cpu_count()}); oracle.knn_oracle.search_blas",
        # the reference's real call shape: one query, k' = 100, one thread (faiss scans single queries on one thread)
        "nq1_one_thread_latency_ms": full1 * 1e3,
        "nq1_one_thread_scan_GBps": rows1 * args.dim * 4 / t1 / 1e9,
        "nq1_sample": f"1 query top-100, scalar C scan on 1 thread, first {rows1} rows, mean of {reps1}, time x{args.rows / rows1:.1f}",
    }


# what the rocprofv3 FETCH_SIZE / WRITE_SIZE counters behind every "traffic" field measure (MI355X_MICROARCH.md)
TRAFFIC_KIND = "L2-miss bytes (fabric side: HBM + Infinity-Cache hits), rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes"


def bench_query_e2e(args, dev, index, stream, log):
    """The reference's user-visible query (src/cli.py:232-251): generate_single_embedding(text) -> search with
    k' = min(10 * top_k, ntotal) = 100 (src/storage.py:429-436).  Encoder = the tiny-batch hipGraph path, search = the
    single-query cascade over the resident index.  Two chains: device resident (ids on the device, one stream, one
    synchronisation at the end) and the host API the reference's call sites use (numpy in / numpy out per call)."""
    import ctypes

    import numpy as np
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder

    enc = MpnetEncoder(synthetic_seed=1, compute="bf16", device=dev.index or 0)
    ntok, kq = 16, 100                      # a one-line query: <s> + 14 word pieces + </s>
    ids_h = np.concatenate([[0], 4 + (np.arange(ntok - 2) * 7919) % 30000, [2]]).astype(np.int32)
    cu_h = np.array([0, ntok], dtype=np.int32)
    ids, cu = torch.from_numpy(ids_h).to(dev), torch.from_numpy(cu_h).to(dev)
    emb = torch.empty((1, 768), dtype=torch.float32, device=dev)
    D = torch.empty((1, kq), dtype=torch.float32, device=dev)
    I = torch.empty((1, kq), dtype=torch.int64, device=dev)

    def encode_dev():
        nat.check(nat.lib().css_encoder_forward_dev(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()), 1,
                                                    ntok, ntok, 1, ctypes.c_void_p(emb.data_ptr()), ctypes.c_void_p(stream)))

    def search_dev():
        index.search_dev(emb.data_ptr(), 1, kq, D.data_ptr(), I.data_ptr(), stream, normalize=True)

    def timed(fn, reps):
        for _ in range(4):                  # (a tiny-batch shape is captured into a hipGraph on its second use)
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()        # per-query latency: every query waits for its answer
        return (time.perf_counter() - t0) / reps * 1e3

    reps = 30
    enc_ms, search_ms = timed(encode_dev, reps), timed(search_dev, reps)
    both_ms = timed(lambda: (encode_dev(), search_dev()), reps)
    q_h = None

    def host_chain():
        nonlocal q_h
        q_h = enc.encode_ids([ids_h.tolist()])             # numpy [1, 768] back on the host
        return index.search(q_h, kq, normalize=True)

    host_ms = timed(host_chain, 10)
    host_enc_ms = timed(lambda: enc.encode_ids([ids_h.tolist()]), 10)
    enc.close()
    out = {"encode_ms": enc_ms, "search_ms": search_ms, "sum_ms": enc_ms + search_ms, "chained_ms": both_ms,
           "host_api_chain_ms": host_ms, "host_api_encode_ms": host_enc_ms, "host_api_search_ms": host_ms - host_enc_ms,
           "query_tokens": ntok, "k": kq, "rows": int(index.ntotal),
           "note": "encode = hipGraph replay of the tiny-batch encoder path, search = single-query coarse sweep + exact "
                   "rescoring (k' = 100, the reference's call shape for top_k = 10); device-resident numbers synchronise "
                   "once per query; host_api_* are the numpy-in / numpy-out calls of the reference's own call sites "
                   "(PCIe copies and two synchronisations included)"}
    log(f"query e2e: encode {enc_ms:.2f} ms + search {search_ms:.2f} ms (chained {both_ms:.2f} ms; host API {host_ms:.2f} ms)")
    return out


def encoder_flops(lengths, layers=12):
    """Algorithmic flops: per token per layer 14,155,776 (GEMMs) + 4*L*768 (attention)."""
    return float(sum(layers * (L * 14155776 + 4 * L * L * 768) for L in lengths))


def pmc_traffic(kernel_prefix, workload):
    """L2-miss (fabric-side: HBM + Infinity-Cache hits, MI355X_MICROARCH.md) bytes per launch of `kernel_prefix` from the newest committed rocprofv3 PMC summary
    (profiles/r*_pmc_hbm_traffic.json, made by tools/pmc_summarise.py from separate --pmc FETCH_SIZE /
    --pmc WRITE_SIZE passes over this same command); None when no summary matches this workload."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        for name, v in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix):
                return {"bytes_per_launch": v["hbm_bytes_per_launch_corrected"], "source": os.path.relpath(f, ROOT)}
    return None


def pmc_traffic_encoder(workload):
    """L2-miss (fabric-side, includes Infinity-Cache hits) bytes of ONE encoder forward: every kernel's bytes summed over a profiled run of the fixed shape
    (profiles/r*_pmc_hbm_traffic_encoder.json), divided by the number of forwards (= launches of the embedding kernel)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_encoder.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        ks = d.get("kernels", {})
        fw = sum(v["launches"] for k_, v in ks.items() if k_.startswith(("k_embed_ln", "k_embed_pre")))
        if not fw:
            continue
        tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for k_, v in ks.items()
                  if k_.startswith(("k_gemm", "k_attention", "k_layernorm", "k_embed", "k_pool")))
        return {"bytes_per_forward": tot / fw, "source": os.path.relpath(f, ROOT)}
    return None


def bench_encoder(args, dev, log):
    """Batch-256 x 384-token encode (BASELINE.json configs[2]): synthetic ids + seeded weights."""
    import ctypes

    import numpy as np
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.mpnet_encoder import MpnetEncoder

    B, L = args.enc_batch, args.enc_len
    enc = MpnetEncoder(synthetic_seed=1, compute="bf16", device=dev.index or 0)
    lengths = [L] * B
    T = B * L
    ids_h = synth.uint(7, np.arange(T, dtype=np.uint64), 4, enc.cfg["vocab"]).astype(np.int32)
    ids_h[0::L] = 0
    ids_h[L - 1::L] = 2
    cu_h = (np.arange(B + 1, dtype=np.int64) * L).astype(np.int32)
    ids = torch.from_numpy(ids_h).to(dev)
    cu = torch.from_numpy(cu_h).to(dev)
    out = torch.empty((B, 768), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def fwd():
        nat.check(nat.lib().css_encoder_forward_dev(enc._h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(cu.data_ptr()),
                                                    B, T, L, 1, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))

    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    # the reported time: plain forwards back to back (the per-kernel HIP events of the next loop put two markers
    # around each of the 63 launches of a forward, which costs ~0.2-0.3 ms per forward in launch gaps)
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.enc_steps
    nat.prof_reset()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        fwd()
    torch.cuda.synchronize()
    dt_prof = (time.perf_counter() - t0) / args.enc_steps
    nat.prof_enable(False)
    kern = {}
    for name in ("enc_gemm_qkv", "enc_gemm_o", "enc_gemm_ffn1", "enc_gemm_ffn2", "enc_attention", "enc_layernorm",
                 "enc_embed_ln", "enc_pool"):
        ms, n = nat.prof_read(name)
        if n:
            kern[name] = {"ms_per_batch": ms / args.enc_steps, "launches_per_batch": n // args.enc_steps}
    nat.prof_reset()
    fl = encoder_flops(lengths)
    gemm_fl = {"enc_gemm_qkv": 2.0 * T * 768 * 2304 * 12, "enc_gemm_o": 2.0 * T * 768 * 768 * 12,
               "enc_gemm_ffn1": 2.0 * T * 768 * 3072 * 12, "enc_gemm_ffn2": 2.0 * T * 768 * 3072 * 12,
               "enc_attention": 12.0 * B * 4 * L * L * 768}
    for k_, f in gemm_fl.items():
        if k_ in kern:
            kern[k_]["TFLOPs"] = f / (kern[k_]["ms_per_batch"] / 1e3) / 1e12
    norms = out.norm(dim=1)
    if not os.environ.get("CSS_BENCH_NOCHECK"):   # (kernel timing experiments with deliberately wrong results)
        assert bool(torch.isfinite(out).all()) and float((norms - 1).abs().max()) < 1e-3
    res = {
        "chunks_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B, "seq_len": L, "dtype": "bf16 MFMA, fp32 accumulate",
        "algorithmic_TFLOP_per_batch": fl / 1e12,
        "roofline": {"bound": "mfma", "kernel": "whole forward (12 x [QKV, attention, O+LN, FFN1+GELU, FFN2+LN], pooling)",
                     "achieved": fl / dt / 1e12, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": fl / dt / 1e12 / BF16_MFMA_PEAK_TF, "traffic": None},
        "kernels": kern, "ms_per_batch_with_kernel_events": dt_prof * 1e3,
    }
    tr = pmc_traffic_encoder({"enc_batch": B, "enc_len": L})
    if tr:
        res["roofline"]["traffic"] = tr["bytes_per_forward"]
        res["roofline"]["traffic_source"] = tr["source"]
        res["roofline"]["traffic_kind"] = TRAFFIC_KIND
    log(f"encoder: {B}x{L} in {dt * 1e3:.2f} ms -> {B / dt:.0f} chunks/s, {fl / dt / 1e12:.0f} TFLOP/s")
    if not args.no_cpu_baseline:
        from oracle import mpnet_oracle as mo

        cfg = mo.MpnetCfg()
        w = mo.synth_weights(cfg, 1)
        # sentence-transformers' CPU default batch (src/embeddings.py:33: 16), two batches, every host core torch has
        nb = 32
        pick = list(range(0, B, max(1, B // nb)))[:nb]
        batch = [ids_h[i * L:(i + 1) * L].tolist() for i in pick]
        mo.encode_batched(w, cfg, batch[:2], batch_size=2)   # warm-up
        t0 = time.perf_counter()
        ref = mo.encode_batched(w, cfg, batch, batch_size=16)
        tc = time.perf_counter() - t0
        cos = (out[pick].cpu().numpy() * ref).sum(1)
        res["cpu_baseline"] = {"value": len(pick) / tc, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                               "achieved_GFLOPs": encoder_flops([L] * len(pick)) / tc / 1e9,
                               "sample": f"{len(pick)} of {B} seqs (L={L}), batch 16, mpnet_oracle (torch fp32, {torch.get_num_threads()} thr)"}
        res["parity_vs_oracle_min_cos"] = float(cos.min())
        log(f"encoder cpu baseline: {len(pick)} seqs in {tc:.2f}s; min cos vs oracle {cos.min():.6f}")
    if args.enc_fixed_only:
        enc.close()
        return res
    # second shape: the chunk-length mix of config 1 (chars ~ U[100, 2000],
    # tokens = clip(round(chars / 4) + 2, 2, 384)), packed var-len, same batch size
    chars = 100 + synth.uint(11, np.arange(B, dtype=np.uint64), 0, 1901)
    lens = np.clip(np.round(chars / 4.0).astype(np.int64) + 2, 2, 384)
    lens = np.sort(lens)[::-1].copy()  # sentence-transformers batches are length sorted
    Tm = int(lens.sum())
    cu_m = np.zeros(B + 1, dtype=np.int32)
    np.cumsum(lens, out=cu_m[1:])
    ids_m = synth.uint(13, np.arange(Tm, dtype=np.uint64), 4, enc.cfg["vocab"]).astype(np.int32)
    ids_m[cu_m[:-1]] = 0
    ids_m[cu_m[1:] - 1] = 2
    ids2 = torch.from_numpy(ids_m).to(dev)
    cu2 = torch.from_numpy(cu_m).to(dev)

    def fwd2():
        nat.check(nat.lib().css_encoder_forward_dev(enc._h, ctypes.c_void_p(ids2.data_ptr()), ctypes.c_void_p(cu2.data_ptr()),
                                                    B, Tm, int(lens.max()), 1, ctypes.c_void_p(out.data_ptr()),
                                                    ctypes.c_void_p(stream)))

    for _ in range(2):
        fwd2()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        fwd2()
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / args.enc_steps
    fl2 = encoder_flops(lens.tolist())
    res["length_mix"] = {"chunks_per_s": B / dt2, "ms_per_batch": dt2 * 1e3, "mean_tokens": float(lens.mean()),
                         "total_tokens": Tm, "achieved_TFLOPs": fl2 / dt2 / 1e12,
                         "frac_of_bf16_peak": fl2 / dt2 / 1e12 / BF16_MFMA_PEAK_TF}
    log(f"encoder length mix: mean {lens.mean():.0f} tokens -> {B / dt2:.0f} chunks/s, {fl2 / dt2 / 1e12:.0f} TFLOP/s")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # N > 1 (one rank per GPU): the host-thread-heavy extras below (tokenizer on all cores, per-file encode loops)
        # are single-GPU figures; N ranks running them at once would only oversubscribe the host
        enc.close()
        return res

    # ---- text path: strings -> C++ WordPiece tokenizer (host threads) -> encoder, tokenisation of the next
    # super-batch overlapped with the GPU (MpnetEncoder.encode); synthetic vocabulary and texts (no real
    # vocabulary exists offline), chunk lengths ~ U[100, 2000] characters as in the length-mix run
    try:
        import random
        import string
        import tempfile

        from claude_semantic_search_amd.tokenizer import make_wordpiece

        rng = random.Random(5)
        words = ["".join(rng.choice(string.ascii_lowercase) for _ in range(rng.randint(2, 9))) for _ in range(8000)]
        vocab = ["<s>", "<pad>", "</s>", "<unk>", "[UNK]"] + words + list(string.ascii_lowercase) + \
                ["##" + c for c in string.ascii_lowercase] + list(string.punctuation) + list(string.digits)
        vdir = tempfile.mkdtemp()
        with open(os.path.join(vdir, "vocab.txt"), "w") as f:
            f.write("\n".join(dict.fromkeys(vocab)) + "\n")
        enc.tokenizer = make_wordpiece(os.path.join(vdir, "vocab.txt"))
        ntext = 4096
        texts = []
        for _ in range(ntext):
            target = rng.randint(100, 2000)
            parts, n = [], 0
            while n < target:
                w = rng.choice(words)
                parts.append(w)
                n += len(w) + 1
            texts.append(" ".join(parts) + ".")
        enc.encode(texts[:512], batch_size=B)
        t0 = time.perf_counter()
        toks = enc.tokenize(texts)
        t_tok = time.perf_counter() - t0
        t0 = time.perf_counter()
        emb = enc.encode(texts, batch_size=B)
        t_all = time.perf_counter() - t0
        assert emb.shape == (ntext, 768)
        res["text_path"] = {"chunks_per_s": ntext / t_all, "texts": ntext, "mean_chars": sum(map(len, texts)) / ntext,
                            "mean_tokens": float(sum(len(t) for t in toks) / ntext),
                            "tokenizer": type(enc.tokenizer).__name__, "tokenize_only_ms_per_256": t_tok / ntext * 256 * 1e3,
                            "host_threads": os.cpu_count(), "includes": "utf-8 encode, tokenise, H2D ids, forward, D2H"}
        log(f"encoder text path: {ntext / t_all:.0f} chunks/s end to end ({type(enc.tokenizer).__name__}, "
            f"{t_tok / ntext * 256 * 1e3:.1f} ms tokenisation per 256 texts)")
        # ---- file-size mix: the reference embeds one conversation file at a time
        # (src/cli.py:120-169), i.e. ragged batches; EmbeddingGenerator.generate_embeddings_many / EmbeddingBatcher
        # carry the chunks of many files through full device batches.  Same texts, same generator, both ways.
        from claude_semantic_search_amd.chunk import Chunk
        from claude_semantic_search_amd.embeddings import EmbeddingConfig, EmbeddingGenerator

        enc.close()
        enc = None
        gen = EmbeddingGenerator(EmbeddingConfig(synthetic_weights_seed=1, batch_size=B, show_progress=False,
                                                 embeddings_as_arrays=True, use_gpu=True, auto_batch_size=False))
        gen.load_model()
        gen.model.tokenizer = make_wordpiece(os.path.join(vdir, "vocab.txt"))
        files, i = [], 0
        while i < ntext:
            r = rng.random()
            n = rng.randint(1, 6) if r < 0.55 else (rng.randint(8, 40) if r < 0.9 else rng.randint(60, 300))
            n = min(n, ntext - i)
            files.append([Chunk(f"c{i + j}", texts[i + j], {}) for j in range(n)])
            i += n
        gen.generate_embeddings(files[-1])          # warm-up of both shapes
        gen.generate_embeddings_many(files[:8])
        t0 = time.perf_counter()
        for f in files:
            gen.generate_embeddings(f)
        t_files = time.perf_counter() - t0
        t0 = time.perf_counter()
        gen.generate_embeddings_many(files)
        t_many = time.perf_counter() - t0
        sizes = sorted(len(f) for f in files)
        res["file_mix"] = {"files": len(files), "chunks": ntext, "chunks_per_file_median": sizes[len(sizes) // 2],
                           "chunks_per_file_max": sizes[-1],
                           "per_file_loop_chunks_per_s": ntext / t_files, "cross_file_batches_chunks_per_s": ntext / t_many,
                           "speedup": t_files / t_many,
                           "note": "per_file_loop = the reference's src/cli.py:120-169 call pattern (generate_embeddings per "
                                   "file); cross_file = EmbeddingGenerator.generate_embeddings_many (EmbeddingBatcher)"}
        log(f"file mix ({len(files)} files, median {sizes[len(sizes) // 2]} chunks): per-file loop {ntext / t_files:.0f} chunks/s, "
            f"cross-file batches {ntext / t_many:.0f} chunks/s")
        if hasattr(gen.model, "close"):
            gen.model.close()
    except Exception as ex:  # the text path is an extra: never fail the bench line over it
        res.setdefault("text_path", {"error": repr(ex)})
        res.setdefault("file_mix", {"error": repr(ex)})
    if enc is not None:
        enc.close()
    return res


def env_overrides():
    """Every CSS_* switch set in this process's environment ("NAME=value"): the shipped library honours them (DESIGN.md
    6, "Environment switches"), so a bench line states which were in force."""
    return sorted(f"{k_}={v}" for k_, v in os.environ.items() if k_.startswith("CSS_"))


def self_launch(n: int, script: str, argv) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N <script>
    <same arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1), pass its stdout / stderr
    through and return its exit status (non-zero when any rank failed)."""
    import socket
    import subprocess

    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def check_answers(index, q_host, D, I, id_base, k, log, nq_check=256, n_rescore=16, sample_rows=100_000, require_local=True):
    """Self-check of the answers the timed region produced, on this rank's shard (size-independent properties, the
    ones tests/test_fullsize_gpu.py holds): fp64 re-score of `n_rescore` returned rows read back from the index, and
    no row of a `sample_rows`-row sample of the shard beats a query's k-th returned score.  numpy on the host only."""
    import numpy as np

    Dh, Ih = D.cpu().numpy(), I.cpu().numpy()
    nq = Dh.shape[0]
    qn = q_host.astype(np.float64)
    qn = qn / (np.linalg.norm(qn, axis=1, keepdims=True) + 1e-8)
    n_local = index.ntotal
    local = (Ih >= id_base) & (Ih < id_base + n_local)
    worst, checked = 0.0, 0
    for r in np.linspace(0, nq - 1, num=min(nq, 64)).astype(int):
        for j in np.nonzero(local[r])[0]:
            if checked >= n_rescore:
                break
            row = index.reconstruct(int(Ih[r, j] - id_base)).astype(np.float64)
            worst = max(worst, abs(float(row @ qn[r]) - float(Dh[r, j])))
            checked += 1
    assert worst < 1e-3 and (checked > 0 or not require_local), f"fp64 re-score of returned rows differs by {worst} (checked {checked})"
    m = min(sample_rows, n_local)
    row0 = max(0, (n_local * 3 // 10) - m // 2)
    sample = index.reconstruct_n(row0, m)
    nqc = min(nq, nq_check)
    try:   # (a bounded BLAS pool: the box's CPU share is far below os.cpu_count())
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=16):
            best = (qn[:nqc].astype(np.float32) @ sample.T).max(axis=1)
    except ImportError:
        best = (qn[:nqc].astype(np.float32) @ sample.T).max(axis=1)
    in_sample = ((Ih[:nqc] >= id_base + row0) & (Ih[:nqc] < id_base + row0 + m)).any(axis=1)
    viol = int((best[~in_sample] > Dh[:nqc][~in_sample, k - 1] + 1e-5).sum())
    assert viol == 0, f"{viol} queries: a sampled row beats the k-th returned score"
    log(f"self-check: fp64 re-score of {checked} returned rows within {worst:.1e}; {m}-row sample never beats the k-th "
        f"score of {nqc} queries")
    return {"rescored_rows": checked, "max_abs_diff_vs_fp64": worst, "sample_rows": int(m), "sample_queries": int(nqc),
            "sample_violations": viol}


def timed_steps(step, fence, nsteps):
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    fence()
    return time.perf_counter() - t0


class HipPlatform:
    """What main() needs from the machine: the device, the process group, the fences and the sharded index.  The
    product platform is one HIP device per rank with RCCL ("nccl") between them.  tests/bench_rehearsal.py
    substitutes a CPU platform (gloo + oracle-backed doubles that live under tests/) to drive THIS main() through
    its N > 1 control flow without a GPU; bench.py itself never imports those doubles."""
    name = "hip"

    def __init__(self, local_rank: int):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
        # CSS_BENCH_ONE_GPU=1 (rehearsal of the N > 1 path on a one-GPU box): every rank uses cuda:0 over gloo
        self.one_gpu = os.environ.get("CSS_BENCH_ONE_GPU") == "1"
        self.local_rank = 0 if self.one_gpu else local_rank
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.reduce_dev = torch.device("cpu") if self.one_gpu else self.dev   # where small all-reduce tensors live

    def init_group(self):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=self.dev)

    def barrier(self):
        import torch.distributed as dist

        if self.one_gpu:
            dist.barrier()
        else:
            dist.barrier(device_ids=[self.local_rank])   # (RCCL: name the device, or the barrier guesses it from the rank)

    def sync(self):
        import torch

        torch.cuda.synchronize()

    def stream(self) -> int:
        import torch

        return torch.cuda.current_stream().cuda_stream

    def make_sharded(self, dim: int):
        from claude_semantic_search_amd.sharded import ShardedFlatIndex

        return ShardedFlatIndex(dim, 0, device_index=self.local_rank)


def main(argv=None, platform_factory=None):
    """``platform_factory(local_rank)``: see HipPlatform (tests only pass another one)."""
    rank = int(os.environ.get("RANK", "0"))
    try:
        _main(argv, platform_factory)
    except SystemExit:
        raise
    except BaseException as ex:   # a failure on ANY rank must fail the job, and say which rank it was
        import traceback

        traceback.print_exc()
        print(f"bench.py: rank {rank} failed: {ex!r}", file=sys.stderr, flush=True)
        sys.exit(1)


def _main(argv, platform_factory):
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, BEFORE anything in
        # this process touches the GPU (no HIP call, no torch.cuda.* above this line); relay rank 0's JSON line and
        # the ranks' exit status.  Under torch.distributed.run (WORLD_SIZE set) this branch is never taken.
        sys.exit(self_launch(args.gpus, sys.argv[0], list(sys.argv[1:] if argv is None else argv)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
        raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    overrides = env_overrides()
    debug_switches = [o for o in overrides if o.split("=")[0].endswith("_DBG") or o.startswith("CSS_BENCH_NOCHECK")]
    if debug_switches and not args.allow_debug:
        print(f"bench.py: {debug_switches} make the library return wrong results by design (timing ablations): no headline "
              "is printed with them set (--allow-debug prints a line marked invalid)", file=sys.stderr)
        sys.exit(2)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth

    plat = (platform_factory or HipPlatform)(local_rank)
    dev = plat.dev
    hip = plat.name == "hip"
    if args.only_encoder:
        print(json.dumps({"encode": bench_encoder(args, dev, log)}), flush=True)
        return
    if world > 1:
        plat.init_group()

    def fence():
        if world > 1:
            plat.barrier()
        plat.sync()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=plat.reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- build this rank's shard in HBM (rows generated on the device) -------
    # weak scaling: args.rows rows PER GPU of one virtual index of args.rows * world rows (N = 8: 80 M x 768)
    rows_total = args.rows * world
    stream = plat.stream()
    sh = plat.make_sharded(args.dim)
    t0 = time.perf_counter()
    sh.add_synthetic_global(rows_total, seed=4, normalize=True, stream=stream)
    index = sh.local
    shard = index.ntotal
    plat.sync()
    log(f"rank0 shard: {shard} of {rows_total} rows x {args.dim} ({shard * args.dim * 4 / 1e9:.2f} GB fp32) generated in "
        f"{time.perf_counter() - t0:.2f}s")

    q_host = synth.rows(args.nq, args.dim, 5)
    q = torch.from_numpy(q_host).to(dev)
    result = {}

    def step():
        result["DI"] = sh.search_tensors(q, args.k, normalize=True)

    for _ in range(args.warmup):
        step()
    fence()
    nat.prof_reset()
    nat.prof_enable(True)
    elapsed = timed_steps(step, fence, args.steps)
    nat.prof_enable(False)
    elapsed = max_over_ranks(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    qps = args.nq * args.steps / elapsed
    D, I = result["DI"]
    assert D.shape == (args.nq, args.k) and bool((I >= 0).all()) and bool((I < rows_total).all())
    assert bool((D[:, 1:] <= D[:, :-1]).all())
    if world > 1:   # every shard contributes to the merged answer
        owners = torch.bincount((I.flatten() // args.rows).clamp_(max=world - 1), minlength=world)
        assert int((owners > 0).sum()) == world, owners.tolist()
    D_timed, I_timed = D.clone(), I.clone()   # (checked after the latency legs below: see check_answers)

    # ---- roofline of the dominant kernel (HIP events on the launch stream) ----
    kernels = {}
    for name in ("knn_scan_coarse_main", "knn_coarse_cascade", "knn_scan_mfma", "knn_scan_split", "knn_scan_small",
                 "knn_fix_scan", "knn_merge", "knn_merge_parts"):
        ms, n = nat.prof_read(name)
        if n:
            kernels[name] = (ms, n)
    cand = {k_: v for k_, v in kernels.items() if k_ != "knn_coarse_cascade"}
    dom = max(cand, key=lambda k_: cand[k_][0]) if cand else None
    roofline = None
    wl = {"rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k}
    if dom:
        ms, n = kernels[dom]
        avg_s = ms / n / 1e3
        sweep_bytes = shard * args.dim * 4          # algorithmic bytes of one fp32 sweep of this rank's shard
        if dom == "knn_scan_coarse_main":
            # last stage of the cascade (k_scan_coarse8<false,true,..>): the row tiles t with t % g != 0 (g = 8 at
            # k <= 32), i.e. 7/8 of the shard, one bf16 MFMA product per (row, query, k); see css_knn_coarse.h
            ntiles = -(-shard // 256)
            scan_i8 = batch_scan_is_int8(sh.local, args.k, shard, args.dim)
            # int8 scan: growth 4 with the last step taken as two steps of 2 -- the main stage is the odd row tiles
            g_ = 2 if scan_i8 else cascade_growth(args.k)
            main_tiles = (ntiles - 1) - (ntiles - 1) // g_
            main_rows = min(main_tiles * 256, shard)
            flops = 2.0 * main_rows * args.dim * args.nq          # ALGORITHMIC flops (multiply-adds x 2) of that launch
            sweep_bytes = main_rows * (args.dim + 4 if scan_i8 else args.dim * 2)   # shadow rows (int8 + scale / bf16) read once
            peak_ = INT8_MFMA_PEAK_TF if scan_i8 else BF16_MFMA_PEAK_TF
            # int8 rows of 256 / 512 / 768 columns: the stages after the first run with the queries resident in registers
            qreg = scan_i8 and args.dim in (256, 512, 768) and os.environ.get("CSS_KNN_QREG", "1")[:1] != "0"
            roofline = {"bound": "mfma",
                        "kernel": (f"k_scan_qreg_i8<{args.dim // 64},true> (cascade main stage, int8 rows, queries in registers)" if qreg else
                                   "k_scan_coarse8<false,true,false,4096,true> (cascade main stage, int8 rows)" if scan_i8
                                   else "k_scan_coarse8<false,true,false,4096,false> (cascade main stage, bf16 rows)"),
                        "achieved": flops / avg_s / 1e12, "peak": peak_, "unit": "TFLOP/s",
                        "frac": flops / avg_s / 1e12 / peak_, "traffic": None,
                        "launches": n, "avg_ms": ms / n, "rows_per_launch": main_rows, "cascade_growth": g_,
                        "hbm_GBps": sweep_bytes / avg_s / 1e9,
                        "arithmetic": ("int8 x int8 -> exact int32 (v_mfma_i32_16x16x64_i8); band rows rescored in fp32" if scan_i8 else
                                       "bf16 x bf16 -> fp32 (v_mfma_f32_16x16x32_bf16); band rows rescored in fp32"),
                        "frac_of_bf16_peak": flops / avg_s / 1e12 / BF16_MFMA_PEAK_TF,
                        "executed_mfma_TFLOPs": flops * (-(-args.nq // 256) * 256 / args.nq) / avg_s / 1e12}
            if scan_i8:
                roofline["note"] = ("MFMA-paced loop at the clock the chip holds under it; CSS_KNN_QREG=0 measures k_scan_coarse8, "
                                    "CSS_KNN_SCAN=bf16 the bf16 scan (DESIGN.md 3.1)") if qreg else \
                    "LDS-traffic / power bound loop; CSS_KNN_SCAN=bf16 measures the bf16 scan (DESIGN.md 3.1)"
            if "knn_coarse_cascade" in kernels:
                cms, cn = kernels["knn_coarse_cascade"]
                roofline["cascade_ms"] = cms / cn           # all stages + selects + rescoring of one search
                roofline["cascade_algorithmic_TFLOPs"] = 2.0 * shard * args.dim * args.nq / (cms / cn / 1e3) / 1e12
        else:
            roofline = {"bound": "hbm", "kernel": dom, "achieved": sweep_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": sweep_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "launches": n, "avg_ms": ms / n}
        roofline["timed_scopes_ms"] = {k_: v[0] / v[1] for k_, v in kernels.items()}
        tr = None
        main_names = (f"k_scan_qreg_i8<{args.dim // 64}, true>",) if (dom == "knn_scan_coarse_main" and scan_i8 and qreg) else \
            ("k_scan_coarse8<false, true, false, 4096, true>",) if (dom == "knn_scan_coarse_main" and scan_i8) else \
            ("k_scan_coarse8<false, true, false, 4096, false>", "k_scan_coarse8<false, true, false, 4096>", "k_scan_coarse<false, true")
        for pref in ({"knn_scan_coarse_main": main_names}.get(dom, ("k_scan_small",))):
            tr = tr or pmc_traffic(pref, wl)
        if tr:
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
            roofline["traffic_kind"] = TRAFFIC_KIND
            roofline["algorithmic_bytes_per_launch"] = sweep_bytes
    nat.prof_reset()

    # ---- the reference's real call shape: one query per call (k = 10 and k' = 100) -------------
    extra = {}
    D = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
    I = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
    if args.legs and hip:
        D1 = torch.empty((1, 100), dtype=torch.float32, device=dev)
        I1 = torch.empty((1, 100), dtype=torch.int64, device=dev)
        for kq, Dq, Iq in (((10, D[:1], I[:1]), (100, D1, I1)) if "nq1" in args.legs else ()):
            for _ in range(3):
                index.search_dev(q.data_ptr(), 1, kq, Dq.data_ptr(), Iq.data_ptr(), stream, normalize=True)
            fence()
            nat.prof_reset()
            nat.prof_enable(True)
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                index.search_dev(q.data_ptr(), 1, kq, Dq.data_ptr(), Iq.data_ptr(), stream, normalize=True)
            fence()
            dt = (time.perf_counter() - t0) / reps
            nat.prof_enable(False)
            info = index.shadow_info() if hasattr(index, "shadow_info") else {"int8": False}
            i8 = bool(info.get("int8")) and os.environ.get("CSS_KNN_SWEEP", "") != "bf16"
            row_b = args.dim * (1 if i8 else 2) + (4 if i8 else 0)   # int8 rows + their fp32 scale, or bf16 rows
            ms, n = nat.prof_read("knn_sweep_fused")
            fused = bool(n)
            if not fused:
                ms, n = nat.prof_read("knn_sweep_coarse_main")
            if fused:   # the whole cascade (every stage and the selects between them) is ONE launch over all shadow rows
                kbytes = shard * row_b
                kname = "k_sweep_cascade<1, 3, true>" if i8 else "k_sweep_cascade<1, 6, false>"
            elif n:   # coarse sweep over the int8 (or bf16) shadow rows: main stage = the row tiles t with t % g != 0 (g = 4)
                ntiles = -(-shard // 256)
                main_rows = min(((ntiles - 1) - (ntiles - 1) // SWEEP_GROWTH) * 256, shard)
                kbytes = main_rows * row_b
                kname = "k_sweep_coarse_i8<1, 3, true>" if i8 else "k_sweep_coarse<1, 6, true>"
            else:   # fp32 sweep (no shadow rows)
                ms, n = nat.prof_read("knn_scan_small")
                kbytes = shard * args.dim * 4
                kname = "k_scan_small<1,"
            gbs = kbytes / (ms / n / 1e3) / 1e9 if n else None
            cms, cn = nat.prof_read("knn_sweep_cascade")
            tr = pmc_traffic(kname, wl)
            rec = {"bound": "hbm", "kernel": kname + (" (the single-query cascade in one launch)" if fused else " (main stage of the single-query cascade)"), "achieved": gbs,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                   "traffic": tr["bytes_per_launch"] if tr else None, "algorithmic_bytes_per_launch": kbytes,
                   "latency_ms": dt * 1e3, "cascade_ms": cms / cn if cn else None, "scan_kernel_ms": ms / n if n else None,
                   # the whole call against the bytes it has to read (all shadow rows once): the end-to-end HBM fraction
                   "whole_call_GBps": shard * row_b / dt / 1e9,
                   "whole_call_frac_of_peak": shard * row_b / dt / 1e9 / HBM_PEAK_GBS,
                   "effective_fp32_index_GBps": shard * args.dim * 4 / dt / 1e9,
                   "rows_read_as": "int8 + per-row scale (bf16 rows would be twice the bytes)" if i8 else "bf16"}
            if roofline is not None:   # flat scalars: the driver's parser keeps no nested objects (details: extra.nq1_k*)
                roofline[f"nq1_k{kq}_latency_ms"] = rec["latency_ms"]
                roofline[f"nq1_k{kq}_main_sweep_hbm_frac"] = rec["frac"]
                roofline[f"nq1_k{kq}_main_sweep_GBps"] = rec["achieved"]
                roofline[f"nq1_k{kq}_whole_call_hbm_frac"] = rec["whole_call_frac_of_peak"]
                roofline[f"nq1_k{kq}_traffic_bytes"] = rec["traffic"]
            extra[f"nq1_k{kq}"] = rec
            nat.prof_reset()
        if "nq1" in args.legs and hip:   # a few queries per call (2..4: the sweep cascade, or the int8 scan; 16: the int8 scan)
            few = {}
            for nqf in (2, 4, 16):
                Df = torch.empty((nqf, args.k), dtype=torch.float32, device=dev)
                If = torch.empty((nqf, args.k), dtype=torch.int64, device=dev)
                for _ in range(3):
                    index.search_dev(q.data_ptr(), nqf, args.k, Df.data_ptr(), If.data_ptr(), stream, normalize=True)
                fence()
                t0 = time.perf_counter()
                for _ in range(10):
                    index.search_dev(q.data_ptr(), nqf, args.k, Df.data_ptr(), If.data_ptr(), stream, normalize=True)
                fence()
                few[f"nq{nqf}_k{args.k}_latency_ms"] = (time.perf_counter() - t0) / 10 * 1e3
            extra["few_queries"] = few
            if roofline is not None:
                roofline.update(few)

        # ---- the reference's user-visible query: encode one query + search it (k' = 100) ----
        if world == 1 and not args.no_encoder and "e2e" in args.legs:
            try:
                extra["query_e2e"] = bench_query_e2e(args, dev, index, stream, log)
                if roofline is not None:
                    roofline["query_e2e_ms"] = extra["query_e2e"]["chained_ms"]
                    roofline["query_e2e_encode_ms"] = extra["query_e2e"]["encode_ms"]
                    roofline["query_e2e_search_ms"] = extra["query_e2e"]["search_ms"]
                    roofline["query_e2e_host_api_ms"] = extra["query_e2e"]["host_api_chain_ms"]
            except Exception as ex_:   # an extra: never fail the bench line over it
                extra["query_e2e"] = {"error": repr(ex_)}

        # ---- masked search (filter / tombstone push-down), half of the rows allowed ----
        words = (shard + 31) // 32
        mbits = torch.full((words,), 0x55555555, dtype=torch.int32, device=dev)   # every other row
        mk = {}
        for nqm, reps in (((1, 10), (args.nq, 3)) if "masked" in args.legs else ()):
            for _ in range(2):
                index.search_dev(q.data_ptr(), nqm, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True,
                                 allow_bits_ptr=mbits.data_ptr())
            fence()
            t0 = time.perf_counter()
            for _ in range(reps):
                index.search_dev(q.data_ptr(), nqm, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True,
                                 allow_bits_ptr=mbits.data_ptr())
            fence()
            dt = (time.perf_counter() - t0) / reps
            assert bool(((I[:nqm] - index_id_base(sh)) % 2 == 0).all())
            mk[f"nq{nqm}_k{args.k}"] = {"ms": dt * 1e3, "queries_per_s": nqm / dt}
        if mk:
            extra["masked_half_rows"] = mk
        del mbits

        # ---- the parity mode: every score formed by fp32 fmaf chains inside the scan kernels ----
        if "exact" in args.legs:
            index.set_search_mode("exact_fp32")
            try:
                ex = {}
                for _ in range(2):
                    index.search_dev(q.data_ptr(), 1, 10, D.data_ptr(), I.data_ptr(), stream, normalize=True)
                fence()
                nat.prof_reset()
                nat.prof_enable(True)
                t0 = time.perf_counter()
                for _ in range(10):
                    index.search_dev(q.data_ptr(), 1, 10, D.data_ptr(), I.data_ptr(), stream, normalize=True)
                fence()
                dt = (time.perf_counter() - t0) / 10
                nat.prof_enable(False)
                ms, n = nat.prof_read("knn_scan_small")
                gbs = shard * args.dim * 4 / (ms / n / 1e3) / 1e9 if n else None
                tr = pmc_traffic("k_scan_small<1,", wl)
                ex["nq1_k10"] = {"bound": "hbm", "kernel": "k_scan_small<1,12,IP> (fp32 VALU sweep)", "achieved": gbs,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                                 "traffic": tr["bytes_per_launch"] if tr else None,
                                 "algorithmic_bytes_per_launch": shard * args.dim * 4,
                                 "latency_ms": dt * 1e3, "scan_kernel_ms": ms / n if n else None}
                nat.prof_reset()
                nqe = min(args.nq, 256)   # the fp32-input MFMA scan is 16x slower per flop than bf16: a quarter batch
                index.search_dev(q.data_ptr(), nqe, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True)
                fence()
                nat.prof_enable(True)
                t0 = time.perf_counter()
                index.search_dev(q.data_ptr(), nqe, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True)
                fence()
                dt = time.perf_counter() - t0
                nat.prof_enable(False)
                ms, n = nat.prof_read("knn_scan_mfma")
                fl = 2.0 * shard * args.dim * nqe
                tr = pmc_traffic("k_scan_mfma<", wl)
                ex[f"nq{nqe}_k{args.k}"] = {
                    "bound": "mfma", "kernel": "k_scan_mfma<IP> (v_mfma_f32_32x32x2_f32, exact fp32 fmaf chains)",
                    "achieved": fl / (ms / n / 1e3) / 1e12 if n else None, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": fl / (ms / n / 1e3) / 1e12 / FP32_MFMA_PEAK_TF if n else None,
                    "traffic": tr["bytes_per_launch"] if tr else None, "algorithmic_bytes_per_launch": shard * args.dim * 4,
                    "queries_per_s": nqe / dt, "ms_per_batch": dt * 1e3, "scan_kernel_ms": ms / n if n else None}
                nat.prof_reset()
                extra["exact_fp32_mode"] = ex
                if roofline is not None:
                    roofline["exact_fp32_nq1_hbm_frac"] = ex["nq1_k10"]["frac"]
                    roofline["exact_fp32_nq1_GBps"] = ex["nq1_k10"]["achieved"]
                    roofline["exact_fp32_nq1_latency_ms"] = ex["nq1_k10"]["latency_ms"]
                    roofline["exact_fp32_nq1_traffic_bytes"] = ex["nq1_k10"]["traffic"]
                    roofline[f"exact_fp32_nq{nqe}_mfma_frac"] = ex[f"nq{nqe}_k{args.k}"]["frac"]
                    roofline[f"exact_fp32_nq{nqe}_ms"] = ex[f"nq{nqe}_k{args.k}"]["ms_per_batch"]
            finally:
                index.set_search_mode("auto")

    # ---- self-check of the timed region's answers on this rank's shard (not timed).  It runs AFTER the single-query
    # latency legs: its host-side SGEMM leaves numpy's BLAS threads spinning on the box's few CPUs for tens of ms, which
    # the next leg would measure as launch latency (seen: 4.7 ms instead of 1.5 ms per single query)
    self_check = None
    if hip and not args.no_check:
        self_check = check_answers(index, q_host, D_timed, I_timed, index_id_base(sh), args.k, log, require_local=world == 1)
    del D_timed, I_timed

    # ---- N > 1: strong scaling of the fixed args.rows-row index on the same ranks ----
    strong = None
    if world > 1 and not args.no_strong:
        sh.local.close()
        sh2 = plat.make_sharded(args.dim)
        sh2.add_synthetic_global(args.rows, seed=4, normalize=True, stream=stream)
        for _ in range(max(2, args.warmup)):
            sh2.search_tensors(q, args.k, normalize=True)
        fence()
        ns = max(args.steps, 5)
        el = max_over_ranks(timed_steps(lambda: sh2.search_tensors(q, args.k, normalize=True), fence, ns))
        strong = {"rows_total": args.rows, "rows_per_gpu": sh2.local.ntotal, "ms_per_step": el / ns * 1e3,
                  "queries_per_s": args.nq * ns / el, "scaling": "strong"}
        sh2.local.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_knn(args, log)

    # ---- clustered rows: how much of the throughput survives dense candidate bands (flagged fraction) ----
    if world == 1 and args.legs and hip:
        if "clustered" in args.legs:
            try:
                extra["clustered_1M"] = bench_clustered(args, dev, stream, log)
            except Exception as ex_:   # an extra: never fail the bench line over it
                extra["clustered_1M"] = {"error": repr(ex_)}
        if args.rows >= 10_000_000 and args.legs & {"noshadow", "80m", "clustered10m"}:
            index.close()   # (46 GB back before three more 10 M-row indexes are built, one at a time)
            if "noshadow" in args.legs:
                try:   # an index without bf16 shadow rows (what a > 38 M-row shard gets): ranges of on-the-fly bf16 rows vs the split-operand scan
                    extra["no_shadow_10M"] = bench_no_shadow(args, dev, stream, log)
                except Exception as ex_:
                    extra["no_shadow_10M"] = {"error": repr(ex_)}
            if "80m" in args.legs:
                try:   # SURVEY 8(d) config 5 on ONE GPU: 80 M x 768 = 245.8 GB of fp32 rows, no room for any shadow copy
                    extra["single_gpu_80M"] = bench_80m_one_gpu(args, dev, stream, log)
                except Exception as ex_:
                    extra["single_gpu_80M"] = {"error": repr(ex_)}
            for nc_ in ((20000, 2000) if "clustered10m" in args.legs else ()):   # 500 and 5000 rows per cluster: a band fits the cascade's buffers / only the second pass's
                try:
                    extra[f"clustered_10M_{nc_}_clusters"] = bench_clustered(args, dev, stream, log, n=10_000_000, nc=nc_,
                                                                             with_uniform=False)
                except Exception as ex_:
                    extra[f"clustered_10M_{nc_}_clusters"] = {"error": repr(ex_)}

    if not args.no_encoder:
        index.close()  # free the shard before the encoder leg
        if world == 1:
            enc = bench_encoder(args, dev, log)
        else:
            # encoder: replicas only (weights replicated, one batch per rank, no collective in the path)
            args.no_cpu_baseline = True
            enc = bench_encoder(args, dev, (lambda m: None) if rank else log)
            t = torch.tensor([enc["chunks_per_s"], enc["ms_per_batch"]], dtype=torch.float64, device=plat.reduce_dev)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            enc["chunks_per_s_all_ranks"] = float(tsum[0].item())
            enc["ms_per_batch_max_over_ranks"] = float(t[1].item())
            enc["parallelism"] = f"{world} replicas, no collective"
        extra["encode"] = enc
        if roofline is not None:   # "chunks embedded/sec" of BASELINE's metric, as flat scalars (details: extra.encode)
            roofline["encode_chunks_per_s"] = enc.get("chunks_per_s_all_ranks", enc["chunks_per_s"])
            roofline["encode_ms_per_batch"] = enc["ms_per_batch"]
            roofline["encode_frac"] = enc["roofline"]["frac"]
            roofline["encode_TFLOPs"] = enc["roofline"]["achieved"]
            roofline["encode_peak_TFLOPs"] = enc["roofline"]["peak"]
            roofline["encode_batch"] = enc["batch"]
            roofline["encode_seq_len"] = enc["seq_len"]
            roofline["encode_traffic_bytes"] = enc["roofline"].get("traffic")
            roofline["encode_length_mix_chunks_per_s"] = enc.get("length_mix", {}).get("chunks_per_s")
            roofline["encode_min_cos_vs_oracle"] = enc.get("parity_vs_oracle_min_cos")
            for kn_, kv_ in enc.get("kernels", {}).items():
                if "TFLOPs" in kv_:
                    roofline[f"{kn_}_ms"] = kv_["ms_per_batch"]
                    roofline[f"{kn_}_frac"] = kv_["TFLOPs"] / BF16_MFMA_PEAK_TF
        if cpu is not None and "cpu_baseline" in enc:   # flat scalars (details: extra.encode.cpu_baseline)
            cpu["encode_chunks_per_s"] = enc["cpu_baseline"]["value"]
            cpu["encode_cores"] = enc["cpu_baseline"]["cores"]
            cpu["encode_sample"] = enc["cpu_baseline"]["sample"]

    if world > 1:
        plat.barrier()
    if rank == 0:
        cfg_name = ("BASELINE configs[3]" if world == 1 and args.rows == 10_000_000 else "BASELINE configs[4] shape")
        out = {
            "metric": f"queries/sec@top-{args.k} over Nx{args.dim} index (+ chunks embedded/sec: roofline.encode_chunks_per_s)",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (int8/bf16 MFMA candidate scan in a measured error band + exact fp32 rescoring)",
            "data": "synthetic" if hip else f"synthetic, on the {plat.name} platform (control-flow rehearsal: NOT a measurement)",
            "config": {"workload": f"nq={args.nq} k={args.k} {rows_total}x{args.dim} fp32 flat IP index, {world} GPU "
                                   f"({args.rows} rows/GPU): {cfg_name}",
                       "rows_total": rows_total, "rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k,
                       "parallelism": f"row-shard x{world} + one packed all-gather(top-k) + merge",
                       "strong_10M": strong},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "self_check": self_check,
            "env_overrides": overrides,
            "extra": extra,
        }
        if debug_switches:
            out["invalid"] = f"debug switches set: {debug_switches}"
            out["metric"] = "INVALID (debug switches) " + out["metric"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def index_id_base(sh):
    """Global id of this shard's local row 0 (one segment: the synthetic bench index)."""
    return sh.segments[0][1] - sh.segments[0][0] if sh.segments else 0


def bench_80m_one_gpu(args, dev, stream, log, n=80_000_000):
    """BASELINE configs[4]'s index on one GPU (the G = 1 end of its strong-scaling line): the fp32
    rows alone fill 245.8 of the 288 GB, so batches convert them range by range into int8 scratch rows per search and
    single queries take the exact fp32 sweep.  Skipped (with the reason) when the rows do not fit this device."""
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info()
    need = n * args.dim * 4
    if args.dim != 768 or free_b < need + (24 << 30):
        return {"skipped": f"{free_b / 1e9:.0f} GB free, {need / 1e9:.0f} GB of rows + 24 GB of scratch needed"}
    ix = IndexFlatIP(args.dim, device=dev.index or 0)
    out = {"rows": n, "nq": args.nq, "k": args.k}
    try:
        ix.reserve(n)
        ix.add_synthetic(n, seed=4, first_row=0, normalize=True, stream=stream)
        qd = torch.from_numpy(synth.rows(args.nq, args.dim, 5)).to(dev)
        Dd = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
        Id = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
        for nq_, reps in ((args.nq, 3), (1, 3)):
            for _ in range(2):
                ix.search_dev(qd.data_ptr(), nq_, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
            torch.cuda.synchronize()
            nat.prof_reset()
            nat.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                ix.search_dev(qd.data_ptr(), nq_, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            nat.prof_enable(False)
            rec = {"ms": dt * 1e3, "queries_per_s": nq_ / dt}
            for scope in ("knn_rows_to_i8", "knn_rows_to_bf16", "knn_coarse_cascade", "knn_scan_small"):
                ms_, cn_ = nat.prof_read(scope)
                if cn_:
                    rec[scope + "_ms"] = ms_ / reps
            if nq_ == 1:
                rec["fp32_sweep_GBps"] = n * args.dim * 4 / dt / 1e9
            out[f"nq{nq_}"] = rec
            nat.prof_reset()
        # parity of this configuration (rows beyond 2^26 of one add, several int8 scratch ranges, k_merge_parts with ids
        # beyond 2^26): queries that ARE rows around 2^24 / 2^26 / 6e7 / the end must find themselves first with score 1,
        # and the batched ranges path must agree with the exact fp32 kernels on a mixed batch.  Fails the extra if not.
        import numpy as np

        probe = np.array([(1 << 24) + 3, (1 << 26) - 1, (1 << 26) + 12_345, 60_000_001, n - 5], dtype=np.int64)
        qrows = np.stack([ix.reconstruct(int(i)) for i in probe])
        qmix = np.concatenate([qrows, synth.rows(19, args.dim, 5)]).astype(np.float32)       # 24 queries: the ranges path
        qm = torch.from_numpy(qmix).to(dev)
        nm = qmix.shape[0]

        def run(nq_):
            ix.search_dev(qm.data_ptr(), nq_, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
            torch.cuda.synchronize()
            return Dd[:nq_].cpu().numpy().copy(), Id[:nq_].cpu().numpy().copy()

        Db, Ib = run(nm)
        ix.set_search_mode("exact_fp32")
        try:
            De, Ie = run(nm)
        finally:
            ix.set_search_mode("auto")
        D1, I1 = run(1)
        assert (Ib[:len(probe), 0] == probe).all() and np.abs(Db[:len(probe), 0] - 1).max() < 1e-5, (Ib[:len(probe), 0], probe)
        assert int(I1[0, 0]) == int(probe[0]) and abs(float(D1[0, 0]) - 1) < 1e-5
        assert np.abs(Db - De).max() < 1e-5 and (Ib == Ie).mean() > 0.97, (np.abs(Db - De).max(), (Ib == Ie).mean())
        assert (np.diff(Db, axis=1) <= 0).all() and all(len(set(r.tolist())) == args.k for r in Ib)
        out["check"] = {"probe_rows_found_first": True, "max_score_diff_vs_exact_fp32": float(np.abs(Db - De).max()),
                        "id_agreement_with_exact_fp32": float((Ib == Ie).mean()), "queries": int(nm)}
        log(f"80 M rows on one GPU: {out[f'nq{args.nq}']['ms']:.0f} ms per {args.nq} queries, {out['nq1']['ms']:.1f} ms per single "
            f"query; probe rows beyond 2^26 found, ranges path == exact fp32 kernels on {nm} queries")
    finally:
        ix.close()
        torch.cuda.empty_cache()
    return out


def bench_no_shadow(args, dev, stream, log, n=10_000_000):
    """The benchmark's index (seed 4) WITHOUT bf16 shadow rows, the state of a shard beyond ~38 M rows per GPU: a
    batch rounds the rows to bf16 range by range into scratch memory and runs the cascade of shadowed indexes per
    range ("ranges", the default); "split" = the split-operand candidate scan it replaced (CSS_SEARCH_SPLIT)."""
    import numpy as np
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    ix = IndexFlatIP(args.dim, device=dev.index or 0)
    ix.set_shadow(False)
    ix.reserve(n)
    ix.add_synthetic(n, seed=4, first_row=0, normalize=True, stream=stream)
    qd = torch.from_numpy(synth.rows(args.nq, args.dim, 5)).to(dev)
    Dd = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
    Id = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
    out = {"rows": n, "nq": args.nq, "k": args.k}
    ref = None
    for mode in ("auto", "split"):
        ix.set_search_mode(mode)
        for _ in range(2):
            ix.search_dev(qd.data_ptr(), args.nq, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
        torch.cuda.synchronize()
        nat.prof_reset()
        nat.prof_enable(True)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_dev(qd.data_ptr(), args.nq, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nat.prof_enable(False)
        cms, cn = nat.prof_read("knn_rows_to_bf16")
        ims, icn = nat.prof_read("knn_rows_to_i8")
        nat.prof_reset()
        rec = {"ms_per_batch": dt * 1e3, "queries_per_s": args.nq / dt}
        if cn:
            rec["rows_to_bf16_ms"] = cms / reps
        if icn:   # the range's scratch rows are int8 where the int8 scan pays (38 GB of conversion traffic instead of 46)
            rec["rows_to_i8_ms"] = ims / reps
        out["ranges" if mode == "auto" else "split"] = rec
        if ref is None:
            ref = (Dd.clone(), Id.clone())
        else:   # both paths return exact fp32 scores (ids may swap inside fp32 near-ties: another summation order in the fix-up)
            out["id_mismatch_fraction_between_paths"] = float((ref[1] != Id).float().mean())
            out["max_score_difference_between_paths"] = float((ref[0] - Dd).abs().max())
            assert out["max_score_difference_between_paths"] < 1e-4 and out["id_mismatch_fraction_between_paths"] < 1e-2
    ix.close()
    log(f"no shadow rows, {n} rows: ranges {out['ranges']['ms_per_batch']:.1f} ms, split-operand scan {out['split']['ms_per_batch']:.1f} ms per batch")
    return out


def bench_clustered(args, dev, stream, log, n=1_000_000, nc=2000, with_uniform=True):
    """1 M rows in 2000 tight clusters (the generator of tests/test_fullsize_gpu.py): queries near cluster centres
    have hundreds of rows inside the bf16 error band.  Reports throughput of the product path, how many queries
    overflowed band or buffer (and were re-run exactly by the device-side fix-up), next to uniform rows of the same size."""
    import numpy as np
    import torch

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    d = args.dim
    cent = synth.rows(nc, d, 71)
    out = {"rows": n, "clusters": nc, "rows_per_cluster": n // nc}
    for name in (("clustered", "uniform") if with_uniform else ("clustered",)):
        ix = IndexFlatIP(d, device=dev.index or 0)
        ix.reserve(n)
        if name == "uniform":
            ix.add_synthetic(n, seed=4, first_row=0, normalize=True, stream=stream)
            qh = synth.rows(args.nq, d, 5)
        else:
            # rows generated on the device (synth.rows_torch is bit-identical to synth.rows): row r = centre r % nc +
            # 0.05 * noise(seed 72 + r // 250000, row r % 250000) -- the generator of tests/test_fullsize_gpu.py
            cent_d = torch.from_numpy(cent).to(dev)
            for c0 in range(0, n, 250_000):
                m = min(250_000, n - c0)
                ids = torch.arange(c0, c0 + m, device=dev) % nc
                rows = cent_d[ids] + 0.05 * synth.rows_torch(m, d, 72 + c0 // 250_000, device=dev)
                ix.add_dev(rows.data_ptr(), m, normalize=True, stream=stream)
                torch.cuda.synchronize()     # (rows is freed at the end of the iteration)
            del cent_d
            near = cent[np.arange(args.nq * 2 // 3) % nc] + 0.02 * synth.rows(args.nq * 2 // 3, d, 90)
            qh = np.concatenate([near, synth.rows(args.nq - near.shape[0], d, 91)])
        qd = torch.from_numpy(np.ascontiguousarray(qh, dtype=np.float32)).to(dev)
        Dd = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
        Id = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
        for _ in range(2):
            ix.search_dev(qd.data_ptr(), args.nq, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
        torch.cuda.synchronize()
        nat.prof_reset()
        nat.prof_enable(True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_dev(qd.data_ptr(), args.nq, args.k, Dd.data_ptr(), Id.data_ptr(), stream, normalize=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nat.prof_enable(False)
        fms, fn = nat.prof_read("knn_fix_scan")
        cms, cn = nat.prof_read("knn_coarse_cascade")
        p2ms, p2n = nat.prof_read("knn_coarse_pass2")
        nat.prof_reset()
        flagged = ix.last_flagged()
        swept = ix.last_swept() if hasattr(ix, "last_swept") else None
        out[name] = {"queries_per_s": args.nq / dt, "ms_per_batch": dt * 1e3, "cascade_ms": cms / cn if cn else None,
                     "flagged_queries": flagged, "flagged_fraction": flagged / args.nq, "swept_exactly": swept,
                     "second_pass_ms": p2ms / p2n if p2n else None, "fixup_scan_ms": fms / fn if fn else None}
        ix.close()
    out["note"] = ("flagged queries (band or buffer overflow) are settled by a second coarse pass against the threshold their "
                   "exactly rescored candidates give (second_pass_ms); swept_exactly = those that overflowed its 32768-slot "
                   "buffers too and were re-run by the exact fp32 sweep (fixup_scan_ms; about 2 us when there are none)")
    log(f"clustered {n} rows / {nc} clusters: {out['clustered']['queries_per_s']:.0f} q/s ({out['clustered']['flagged_queries']} flagged)"
        + (f" vs uniform {out['uniform']['queries_per_s']:.0f} q/s" if with_uniform else ""))
    return out


if __name__ == "__main__":
    main()
