#!/usr/bin/env python3
"""bench.py -- headline benchmark of the embed-and-search hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N > 1 is launched by
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...``,
one rank per GPU over RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[3], the configuration the north-star roofline
target is quoted on): a 10M x 768 fp32 flat inner-product index resident in HBM,
a batch of 1000 queries, exact top-10.  A "step" is one pass of the search path
over the whole query batch.  With N GPUs the SAME 10M-row index is
row-partitioned over the ranks (strong scaling); each step ends with one RCCL
all-gather of the per-shard top-k and a merge.

The JSON line also carries:
  roofline      dominant kernel of the timed region -- the main stage of the search cascade,
                k_scan_coarse<false,true,..> (bf16 MFMA bound) -- measured with HIP events on the
                launch stream inside this run (css_prof_*); `traffic` = HBM-side bytes per launch
                from the newest profiles/r*_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes).
  cpu_baseline  the CPU oracle (kind "port") timed on this box's host cores on a
                bounded sample of the same workload, rank 0 / N=1 only.
  extra         nq1_k10 / nq1_k100: single-query search (the reference's real call shape) with the
                HBM roofline of its main sweep stage; masked_half_rows: the same searches with an
                allow-bitmap; exact_fp32_mode: the parity mode (every score formed in fp32 by the
                scan kernels) on the same index; encode: batch-256 x 384 encoder forward with its
                MFMA roofline, per-kernel times, length-mix and text-path (strings in) runs, CPU
                baseline and parity against the oracle.
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

CASCADE_GROWTH = 8 if os.environ.get("CSS_KNN_GROWTH") == "8" else 4   # growth factor of the coarse cascade (css_index.hip)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3    # dense fp32-input MFMA peak
BF16_MFMA_PEAK_TF = 2500.0   # dense bf16 MFMA peak


def parse_args():
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline work")
    ap.add_argument("--enc-batch", type=int, default=256, help="encoder batch (sequences)")
    ap.add_argument("--enc-len", type=int, default=384, help="encoder sequence length (tokens)")
    ap.add_argument("--enc-steps", type=int, default=5)
    ap.add_argument("--no-encoder", action="store_true")
    ap.add_argument("--only-encoder", action="store_true", help="development aid: run just the encoder leg")
    return ap.parse_args()


def cpu_baseline_knn(args, log):
    """Oracle (numpy/BLAS port of faiss-cpu's batched path) on a bounded row sample."""
    import numpy as np
    from oracle import knn_oracle as ko

    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    q = ko.normalize_rows(ko.synth_rows(args.nq, args.dim, 5))
    probe_rows = 50_000
    x = ko.normalize_rows(ko.synth_rows(probe_rows, args.dim, 4))
    t0 = time.perf_counter()
    ko.search_blas(x, q, args.k)
    t_probe = time.perf_counter() - t0
    rows = int(min(2_000_000, max(probe_rows, probe_rows * args.cpu_seconds / max(t_probe, 1e-3))))
    rows = min(rows, args.rows)
    x = ko.normalize_rows(ko.synth_rows(rows, args.dim, 4))
    t0 = time.perf_counter()
    ko.search_blas(x, q, args.k)
    t = time.perf_counter() - t0
    full = t * (args.rows / rows)
    log(f"cpu baseline: {rows} rows x {args.nq} queries in {t:.2f}s (threads={blas_threads})")
    return {
        "value": args.nq / full,
        "unit": "queries/s",
        "cores": int(blas_threads),
        "kind": "port",
        "sample": f"first {rows} of {args.rows} rows x {args.nq} queries, numpy SGEMM + top-{args.k} "
                  f"(oracle.knn_oracle.search_blas), time scaled x{args.rows / rows:.1f} (extrapolated)",
    }


def encoder_flops(lengths, layers=12):
    """Algorithmic flops: per token per layer 14,155,776 (GEMMs) + 4*L*768 (attention)."""
    return float(sum(layers * (L * 14155776 + 4 * L * L * 768) for L in lengths))


def pmc_traffic(kernel_prefix, workload):
    """HBM bytes per launch of `kernel_prefix` from the newest committed rocprofv3 PMC summary
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
    nat.prof_reset()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.enc_steps):
        fwd()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.enc_steps
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
    assert bool(torch.isfinite(out).all()) and float((norms - 1).abs().max()) < 1e-3
    res = {
        "chunks_per_s": B / dt, "ms_per_batch": dt * 1e3, "batch": B, "seq_len": L, "dtype": "bf16 MFMA, fp32 accumulate",
        "algorithmic_TFLOP_per_batch": fl / 1e12,
        "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                     "frac": fl / dt / 1e12 / BF16_MFMA_PEAK_TF, "traffic": None},
        "kernels": kern,
    }
    log(f"encoder: {B}x{L} in {dt * 1e3:.2f} ms -> {B / dt:.0f} chunks/s, {fl / dt / 1e12:.0f} TFLOP/s")
    if not args.no_cpu_baseline:
        from oracle import mpnet_oracle as mo

        cfg = mo.MpnetCfg()
        w = mo.synth_weights(cfg, 1)
        nb = 2
        batch = [ids_h[i * L:(i + 1) * L].tolist() for i in range(nb)]
        # one sequence at a time is small-matrix work: more threads than ~16 only thrash
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        ref = mo.encode(w, cfg, batch)
        tc = time.perf_counter() - t0
        cos = (out[:nb].cpu().numpy() * ref).sum(1)
        res["cpu_baseline"] = {"value": nb / tc, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{nb} of the {B} sequences (L={L}) through oracle.mpnet_oracle (torch fp32)"}
        res["parity_vs_oracle_min_cos"] = float(cos.min())
        log(f"encoder cpu baseline: {nb} seqs in {tc:.2f}s; min cos vs oracle {cos.min():.6f}")
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
    except Exception as ex:  # the text path is an extra: never fail the bench line over it
        res["text_path"] = {"error": repr(ex)}
    enc.close()
    return res


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    import numpy as np
    import torch
    import torch.distributed as dist

    from claude_semantic_search_amd import _native as nat
    from claude_semantic_search_amd import synth
    from claude_semantic_search_amd.flat_index import IndexFlatIP

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device (no CPU fallback)")
    # CSS_BENCH_ONE_GPU=1 (rehearsal of the N > 1 path on a one-GPU box): every rank uses cuda:0 over gloo
    one_gpu = os.environ.get("CSS_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.only_encoder:
        print(json.dumps({"encode": bench_encoder(args, dev, log)}), flush=True)
        return
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- build this rank's shard in HBM (rows generated on the device) -------
    lo = rank * args.rows // world
    hi = (rank + 1) * args.rows // world
    shard = hi - lo
    stream = torch.cuda.current_stream().cuda_stream
    index = IndexFlatIP(args.dim, device=local_rank)
    index.reserve(shard)
    t0 = time.perf_counter()
    index.add_synthetic(shard, seed=4, first_row=lo, normalize=True, stream=stream)
    index.set_id_base(lo)
    torch.cuda.synchronize()
    log(f"rank0 shard: {shard} rows x {args.dim} ({shard * args.dim * 4 / 1e9:.2f} GB) generated in "
        f"{time.perf_counter() - t0:.2f}s")

    q_host = synth.rows(args.nq, args.dim, 5)
    q = torch.from_numpy(q_host).to(dev)
    D = torch.empty((args.nq, args.k), dtype=torch.float32, device=dev)
    I = torch.empty((args.nq, args.k), dtype=torch.int64, device=dev)
    if world > 1:
        Dg = torch.empty((world, args.nq, args.k), dtype=torch.float32, device=dev)
        Ig = torch.empty((world, args.nq, args.k), dtype=torch.int64, device=dev)
        Dm = torch.empty_like(D)
        Im = torch.empty_like(I)

    def step():
        index.search_dev(q.data_ptr(), args.nq, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True)
        if world > 1:
            if one_gpu:  # gloo has no device all_gather_into_tensor: stage through the host (rehearsal only)
                dl = [torch.empty_like(D, device="cpu") for _ in range(world)]
                il = [torch.empty_like(I, device="cpu") for _ in range(world)]
                dist.all_gather(dl, D.cpu())
                dist.all_gather(il, I.cpu())
                Dg.copy_(torch.stack(dl))
                Ig.copy_(torch.stack(il))
            else:
                dist.all_gather_into_tensor(Dg.view(world * args.nq, args.k), D)
                dist.all_gather_into_tensor(Ig.view(world * args.nq, args.k), I)
            nat.check(nat.lib().css_merge_topk_dev(ctypes.c_void_p(Dg.data_ptr()), ctypes.c_void_p(Ig.data_ptr()),
                                                   world, args.nq, args.k, 0, ctypes.c_void_p(Dm.data_ptr()),
                                                   ctypes.c_void_p(Im.data_ptr()), local_rank,
                                                   ctypes.c_void_p(stream)))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    nat.prof_reset()
    nat.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    nat.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    qps = args.nq * args.steps / elapsed

    # ---- roofline of the dominant kernel (HIP events on the launch stream) ----
    kernels = {}
    for name in ("knn_scan_coarse_main", "knn_coarse_cascade", "knn_scan_mfma", "knn_scan_small", "knn_merge",
                 "knn_merge_parts"):
        ms, n = nat.prof_read(name)
        if n:
            kernels[name] = (ms, n)
    cand = {k_: v for k_, v in kernels.items() if k_ != "knn_coarse_cascade"}
    dom = max(cand, key=lambda k_: cand[k_][0]) if cand else None
    roofline = None
    if dom:
        ms, n = kernels[dom]
        avg_s = ms / n / 1e3
        sweep_bytes = shard * args.dim * 4          # algorithmic bytes of one fp32 sweep of this rank's shard
        if dom == "knn_scan_coarse_main":
            # last stage of the cascade (k_scan_coarse<false,true,..>): the row tiles t with t % g != 0 (g = 4), i.e.
            # 3/4 of the shard, one bf16 MFMA product per (row, query, k); see css_knn_coarse.h
            ntiles = -(-shard // 256)
            main_tiles = (ntiles - 1) - (ntiles - 1) // CASCADE_GROWTH
            main_rows = min(main_tiles * 256, shard)
            flops = 2.0 * main_rows * args.dim * args.nq          # ALGORITHMIC flops of that launch
            sweep_bytes = main_rows * args.dim * 2                 # bf16 shadow rows read once
            roofline = {"bound": "mfma", "kernel": "k_scan_coarse<false,true,false,16> (main stage of the cascade)",
                        "achieved": flops / avg_s / 1e12, "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                        "frac": flops / avg_s / 1e12 / BF16_MFMA_PEAK_TF, "traffic": None,
                        "launches": n, "avg_ms": ms / n, "rows_per_launch": main_rows,
                        "hbm_GBps": sweep_bytes / avg_s / 1e9,
                        "arithmetic": "bf16 operands (shadow rows), fp32 accumulate, v_mfma_f32_16x16x32_bf16; "
                                      "candidates rescored in fp32",
                        "executed_mfma_TFLOPs": flops * (-(-args.nq // 256) * 256 / args.nq) / avg_s / 1e12}
            if "knn_coarse_cascade" in kernels:
                cms, cn = kernels["knn_coarse_cascade"]
                roofline["cascade_ms"] = cms / cn           # all stages + selects + rescoring of one search
                roofline["cascade_algorithmic_TFLOPs"] = 2.0 * shard * args.dim * args.nq / (cms / cn / 1e3) / 1e12
        elif dom == "knn_scan_mfma":
            nq_launch = args.nq * args.steps / n     # queries served per launch
            flops = 2.0 * shard * args.dim * nq_launch   # ALGORITHMIC (fp32 dot-product) flops
            split = os.environ.get("CSS_KNN_BATCH", "split") != "fp32"
            # split mode: every fp32-grade product costs 3 bf16 MFMA products (h.h + h.l + l.h), so the
            # roof of the emulation is the dense bf16 peak / 3; fp32 mode: the fp32-input MFMA peak.
            peak = BF16_MFMA_PEAK_TF / 3.0 if split else FP32_MFMA_PEAK_TF
            roofline = {"bound": "mfma", "kernel": dom, "achieved": flops / avg_s / 1e12, "peak": peak,
                        "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / peak, "traffic": None,
                        "launches": n, "avg_ms": ms / n, "hbm_GBps": sweep_bytes / avg_s / 1e9,
                        "arithmetic": ("bf16x3 split-operand MFMA, fp32 accumulate (peak = 2500/3)" if split
                                       else "fp32-input MFMA (exact fp32)"),
                        "executed_mfma_TFLOPs": (3.0 if split else 1.0) * flops * (-(-args.nq // 128) * 128 / args.nq) / avg_s / 1e12}
        else:
            roofline = {"bound": "hbm", "kernel": dom, "achieved": sweep_bytes / avg_s / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": sweep_bytes / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "launches": n, "avg_ms": ms / n}
    if roofline:
        roofline["timed_scopes_ms"] = {k_: v[0] / v[1] for k_, v in kernels.items()}
        wl = {"rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k}
        tr = pmc_traffic({"knn_scan_mfma": "k_scan_mfma", "knn_scan_coarse_main": "k_scan_coarse<false, true"}.get(dom, "k_scan_small"), wl)
        if tr:
            roofline["traffic"] = tr["bytes_per_launch"]
            roofline["traffic_source"] = tr["source"]
            roofline["algorithmic_bytes_per_launch"] = sweep_bytes
    nat.prof_reset()

    # ---- extra: the reference's real call shape (one query, k'=100) -------------
    extra = {}
    if not args.no_extra:
        D1 = torch.empty((1, 100), dtype=torch.float32, device=dev)
        I1 = torch.empty((1, 100), dtype=torch.int64, device=dev)
        for kq, Dq, Iq in ((10, D[:1], I[:1]), (100, D1, I1)):
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
            ms, n = nat.prof_read("knn_sweep_coarse_main")
            if n:   # coarse sweep over the bf16 shadow rows: main stage = the row tiles t with t % 4 != 0
                ntiles = -(-shard // 256)
                main_rows = min(((ntiles - 1) - (ntiles - 1) // CASCADE_GROWTH) * 256, shard)
                kbytes = main_rows * args.dim * 2
                kname = "k_sweep_coarse<1, 6, true>"
            else:   # fp32 sweep (no shadow rows)
                ms, n = nat.prof_read("knn_scan_small")
                kbytes = shard * args.dim * 4
                kname = "k_scan_small<1,"
            gbs = kbytes / (ms / n / 1e3) / 1e9 if n else None
            cms, cn = nat.prof_read("knn_sweep_cascade")
            tr = pmc_traffic(kname, {"rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k})
            extra[f"nq1_k{kq}"] = {"latency_ms": dt * 1e3, "cascade_ms": cms / cn if cn else None,
                                   "scan_kernel_ms": ms / n if n else None,
                                   "roofline": {"bound": "hbm", "kernel": kname, "achieved": gbs, "peak": HBM_PEAK_GBS,
                                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                                                "traffic": tr["bytes_per_launch"] if tr else None,
                                                "algorithmic_bytes_per_launch": kbytes},
                                   "effective_fp32_index_GBps": shard * args.dim * 4 / dt / 1e9}
            nat.prof_reset()

        # ---- extra: masked search (filter / tombstone push-down), half of the rows allowed ----
        words = (shard + 31) // 32
        mbits = torch.full((words,), 0x55555555, dtype=torch.int32, device=dev)   # every other row
        mk = {}
        for nqm, reps in ((1, 10), (args.nq, 3)):
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
            assert bool((I[:nqm] % 2 == 0).all())
            mk[f"nq{nqm}_k{args.k}"] = {"ms": dt * 1e3, "queries_per_s": nqm / dt}
        extra["masked_half_rows"] = mk

        # ---- extra: the parity mode (every score formed in fp32 by the scan kernels) on the same index ----
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
            tr = pmc_traffic("k_scan_small<1,", {"rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k})
            ex["nq1_k10"] = {"latency_ms": dt * 1e3, "scan_kernel_ms": ms / n if n else None,
                             "roofline": {"bound": "hbm", "kernel": "k_scan_small<1,12,IP>", "achieved": gbs,
                                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS if gbs else None,
                                          "traffic": tr["bytes_per_launch"] if tr else None,
                                          "algorithmic_bytes_per_launch": shard * args.dim * 4}}
            nat.prof_reset()
            index.search_dev(q.data_ptr(), args.nq, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True)
            fence()
            nat.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(2):
                index.search_dev(q.data_ptr(), args.nq, args.k, D.data_ptr(), I.data_ptr(), stream, normalize=True)
            fence()
            dt = (time.perf_counter() - t0) / 2
            nat.prof_enable(False)
            ms, n = nat.prof_read("knn_scan_mfma")
            fl = 2.0 * shard * args.dim * args.nq
            ex[f"nq{args.nq}_k{args.k}"] = {
                "queries_per_s": args.nq / dt, "ms_per_batch": dt * 1e3, "scan_kernel_ms": ms / n if n else None,
                "roofline": {"bound": "mfma", "kernel": "k_scan_mfma_split<IP,8,8>",
                             "achieved": fl / (ms / n / 1e3) / 1e12 if n else None, "peak": BF16_MFMA_PEAK_TF / 3.0,
                             "unit": "TFLOP/s", "frac": fl / (ms / n / 1e3) / 1e12 / (BF16_MFMA_PEAK_TF / 3.0) if n else None,
                             "arithmetic": "fp32 operands split into bf16 pairs, 3 MFMA products per fp32-grade product"}}
            nat.prof_reset()
            extra["exact_fp32_mode"] = ex
        finally:
            index.set_search_mode("auto")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_knn(args, log)
    if not args.no_encoder:
        index.close()  # free the shard before the encoder leg
        if world == 1:
            extra["encode"] = bench_encoder(args, dev, log)
        else:
            # encoder: replicas only (weights replicated, one batch per rank, no collective in the path)
            args.no_cpu_baseline = True
            enc = bench_encoder(args, dev, (lambda m: None) if rank else log)
            t = torch.tensor([enc["chunks_per_s"], enc["ms_per_batch"]], dtype=torch.float64, device="cpu" if one_gpu else dev)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            enc["chunks_per_s_all_ranks"] = float(tsum[0].item())
            enc["ms_per_batch_max_over_ranks"] = float(t[1].item())
            enc["parallelism"] = f"{world} replicas, no collective"
            extra["encode"] = enc

    if world > 1:
        dist.barrier()
    if rank == 0:
        out = {
            "metric": "queries/sec@top-10 over 10Mx768 flat index (1k-query batch)",
            "value": qps,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (index, queries and returned scores fp32; candidate selection by a bf16 MFMA scan with a "
                     "rigorous error band, candidates rescored in fp32)",
            "data": "synthetic",
            "config": {"workload": f"{args.rows}x{args.dim} fp32 flat inner-product index, nq={args.nq}, "
                                   f"top-{args.k}, index row-partitioned over {world} GPU(s)",
                       "rows_total": args.rows, "rows_per_gpu": shard, "dim": args.dim, "nq": args.nq, "k": args.k,
                       "parallelism": f"row-shard x{world} + all-gather(top-k)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
