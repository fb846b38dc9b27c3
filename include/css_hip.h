/*
 * css_hip.h -- C ABI of libcss_hip.so, the MI355X (gfx950) implementation of the
 * embed-and-search hot path of pauloportella/claude-semantic-search.
 *
 * The reference is pure Python; its arithmetic is reached through two third
 * party seams.  Each entry point below names the reference
 * call site it replaces (paths relative to the reference checkout):
 *
 *   faiss.IndexFlatIP(d) / IndexFlatL2(d)    src/storage.py:252-258  -> css_index_create
 *   faiss_index.add(x)                       src/storage.py:359      -> css_index_add
 *   x / (norm + 1e-8) row-normalise          src/storage.py:347-350  -> css_index_add(normalize=1)
 *   faiss_index.ntotal                       src/storage.py:358,421  -> css_index_ntotal
 *   q / (norm + 1e-8), reshape(1,-1)         src/storage.py:424-429  -> css_index_search(normalize_q=1)
 *   faiss_index.search(q, k) -> (D, I)       src/storage.py:436      -> css_index_search
 *   faiss.write_index / read_index payload   src/storage.py:306,879  -> css_index_export / css_index_add
 *   faiss.index_cpu_to_gpu / get_num_gpus    src/storage.py:283, src/gpu_utils.py:117-118
 *                                                                    -> css_device_count / css_device_info
 *   SentenceTransformer(name).encode(...)    src/embeddings.py:184-188, :216-222
 *                                                                    -> css_encoder_forward
 *   model.get_sentence_embedding_dimension() src/embeddings.py:117   -> css_encoder_cfg.hidden
 *
 * Conventions: extern "C", opaque handles, plain pointers and sizes.  Every
 * function returns 0 on success or a negative css_status; the message for the
 * calling thread is available from css_last_error().  Host pointers are caller
 * owned and are consumed before return.  "_dev" twins take device pointers and
 * a hipStream_t (as void*) and enqueue asynchronously on that stream; they are
 * what a PyTorch-ROCm host passes tensor.data_ptr() / current_stream to.
 * There is NO CPU fallback in this library: with no HIP device every compute
 * entry point fails with CSS_ERR_NO_DEVICE.
 */
#ifndef CSS_HIP_H
#define CSS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum css_status {
    CSS_OK = 0,
    CSS_ERR_INVALID = -1,   /* bad argument */
    CSS_ERR_NO_DEVICE = -2, /* no usable HIP device */
    CSS_ERR_HIP = -3,       /* HIP runtime error (message has the hipError string) */
    CSS_ERR_OOM = -4,       /* device allocation failed */
    CSS_ERR_STATE = -5      /* object not in a state that allows the call */
} css_status;

enum { CSS_METRIC_IP = 0, CSS_METRIC_L2 = 1 };

/* Largest k accepted by css_index_search.  The reference asks for
 * k' = min(SearchConfig.max_results, ntotal) (src/storage.py:432; max_results
 * defaults to 100, :69, and may be set to anything).  Up to 128 a search is one
 * pass of the scan kernels; beyond, every query takes ceil(k / 128) passes over
 * the rows the earlier passes did not return (still exact, still enqueue-only). */
#define CSS_MAX_K 2048

typedef struct css_devinfo {
    char name[128];
    char gcn_arch[64];
    int compute_units;
    int wavefront_size;
    int64_t hbm_total_bytes;
    int64_t hbm_free_bytes;
    int lds_bytes_per_cu;
    int clock_mhz;
} css_devinfo;

typedef struct css_index css_index;
typedef struct css_encoder css_encoder;

const char* css_version(void);
const char* css_last_error(void);

int css_device_count(int* n);
int css_device_info(int device, css_devinfo* out);

/* ---- flat exact index (IndexFlatIP / IndexFlatL2 semantics) ---- */
int css_index_create(int dim, int metric, int device, css_index** out);
int css_index_free(css_index* ix);
int css_index_reset(css_index* ix);                 /* ntotal := 0, keeps capacity */
int css_index_reserve(css_index* ix, int64_t n);    /* capacity >= n rows, no copy later */
int css_index_ntotal(const css_index* ix, int64_t* n);
int css_index_dim(const css_index* ix, int* dim);
int css_index_metric(const css_index* ix, int* metric);
int css_index_device(const css_index* ix, int* device);
/* Diagnostics: how many queries of the LAST candidate-path search (its last chunk of up to 4096 queries) overflowed
 * their candidate buffer or band (and were settled by the second coarse pass or the exact sweep).  Waits for the device. */
int css_index_last_flagged(css_index* ix, int64_t* n);
/* ... and how many of those the second coarse pass could not settle either (candidate buffers of 32768 rows
 * overflowed, or more than 1024 flagged queries in a chunk): these were re-run by the exact fp32 sweep. */
int css_index_last_swept(css_index* ix, int64_t* n);
/* Reduced-precision copies of the rows, the operands of the candidate scans (results never depend on them: candidates
 * come from a scan inside an error band measured at ingest and are rescored in fp32):
 *   bf16 rows (+50 % HBM next to the fp32 rows) and INT8 rows (signed byte = round(x / s), s = max|x| / 127 per row;
 *   +25 %; rows of at most 1024 elements).  Searches of 1..4 queries sweep the int8 rows (one query: the whole stage
 *   cascade in one persistent kernel launch, css_knn_coarse.h: k_sweep_cascade), 3..32 inner-product queries sweep them on
 *   the int8 MFMA with the queries as the register operand (k_sweep_mfma_i8); batches scan them with int8
 *   MFMA (rows of 256 / 512 / 768 elements: the queries resident in registers, k_scan_qreg_i8) where that pays (inner product, rows a multiple of 256 elements: k <= 32 from 300 k rows, k <= 128 from 2 M rows; an index whose int8 searches flag more than 5 % of their queries falls back to the bf16 rows
 *   for the next 16 searches), otherwise the bf16 rows.
 * policy: -1 = automatic -- both copies while 7 bytes per element fit in 80 % of the HBM, bf16 only at 6 bytes, INT8 ONLY
 * at 5 bytes (inner product, rows a multiple of 256 elements: ~38-46 M rows of 768 floats on a 288 GB GPU), nothing
 * beyond (batches then round row ranges into scratch memory per search); 0 = never any copy; 1 = always bf16 (+ int8
 * while it fits); 2 = int8 rows only.  Only on an empty index. */
int css_index_set_shadow(css_index* ix, int policy);
/* Diagnostics: which reduced-precision copies of the rows the index currently holds (0 / 1 each). */
int css_index_shadow_info(css_index* ix, int* has_bf16, int* has_int8);
/* Global id of local row 0 (shards of a row-partitioned index). */
int css_index_set_id_base(css_index* ix, int64_t base);

/* Append n rows (row-major [n, dim] fp32).  normalize != 0 applies the
 * reference's x / (||x||_2 + 1e-8) per row on the device while copying in. */
int css_index_add(css_index* ix, const float* x_host, int64_t n, int normalize);
int css_index_add_dev(css_index* ix, const float* x_dev, int64_t n, int normalize, void* stream);
/* Append n rows generated on the device from include/css_synth.h:
 * row r, column c = css_synth_normal(seed, (first_row + r) * dim + c). */
int css_index_add_synthetic(css_index* ix, int64_t n, uint64_t seed, int64_t first_row,
                            int normalize, void* stream);
/* Copy rows [row0, row0 + n) back to the host as [n, dim] fp32. */
int css_index_export(const css_index* ix, int64_t row0, int64_t n, float* x_out_host);

/* Exact top-k.  IP: D descending inner products.  L2: D ascending squared
 * distances.  Ties: lower id first.  Fewer than k rows: I = -1 and
 * D = -FLT_MAX (IP) / +FLT_MAX (L2).  1 <= k <= CSS_MAX_K.
 * normalize_q != 0 applies q / (||q||_2 + 1e-8) first (src/storage.py:426).
 * Indexes keep a bf16 shadow copy of the rows while it fits in HBM; searches
 * of large indexes then select candidates with a bf16 scan inside a rigorous
 * error band and return exact fp32 scores of the rescored candidates (same
 * results as the fp32 kernels).  Without shadow rows, batches round the rows to
 * bf16 one row range at a time into scratch memory and run the same scan per range.  The _dev form only enqueues on `stream` and never
 * waits for the device (one exception: the FIRST batched search of an index without shadow rows allocates that scratch
 * -- up to half of the free HBM -- and a growing workspace is reallocated; hipMalloc / hipFree synchronise the device.
 * Later searches of the same shapes allocate nothing): queries whose candidate band overflows are re-run exactly by
 * one launch that follows every cascade and returns at once when there are none.
 * Rows appended by css_index_add_dev / css_index_add_synthetic on another stream are
 * ordered before the search by an event (no caller-side synchronisation needed); successive
 * asynchronous adds on different streams are chained the same way.
 * All searches of one index share one set of device workspaces: a search enqueued on a
 * different stream than the previous one first waits (on the device, by an event) for that
 * search to finish, so searches of ONE index execute one after the other whatever streams
 * they are given -- use one index per concurrent stream (or shard) for overlap.  D_dev /
 * I_dev belong to the caller: read them after synchronising with `stream` as usual. */
/* Search path: CSS_SEARCH_AUTO (default) selects candidates with a reduced-precision scan
 * inside a rigorous error band and rescores them in fp32 where the multi-launch cascade
 * pays (5 or more queries, or k > 32: always; 1..4 queries with k <= 32: from 100 k rows;
 * rows are kept as fp32 + bf16 + int8 copies where the HBM allows); CSS_SEARCH_EXACT_FP32 forms every score
 * with fp32 fmaf chains inside the scan kernels (VALU sweeps up to 16 queries, fp32-input
 * MFMA beyond; the parity mode of the tests, and what small indexes use for few queries). */
#define CSS_SEARCH_AUTO 0
#define CSS_SEARCH_EXACT_FP32 1
#define CSS_SEARCH_COARSE 2 /* the candidate path whatever the index size (AUTO uses it only where it pays) */
#define CSS_SEARCH_SPLIT 3  /* batches of > 16 queries: candidates from split-operand (bf16 pair) products of the fp32 rows +
                               fp32 rescoring -- what an index without shadow rows falls back to when no HBM is left for
                               the scratch rows of its bf16 ranges; selectable for verification */
int css_index_set_search_mode(css_index* ix, int mode);
/* Indexes without shadow rows: rows per bf16 scratch range of a batched search (0 = automatic: half of the free HBM,
 * at most 2^24 rows).  A tuning / verification knob: results do not depend on it. */
int css_index_set_range_rows(css_index* ix, int64_t rows);
int css_index_search(css_index* ix, const float* q_host, int64_t nq, int k, int normalize_q,
                     float* D_host, int64_t* I_host);
int css_index_search_dev(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q,
                         float* D_dev, int64_t* I_dev, void* stream);

/* Masked search (filter / tombstone push-down; the reference
 * instead over-fetches 100 hits and filters them afterwards, src/storage.py:438-492):
 * only rows whose bit is set in allow_bits -- bit (r & 31) of word r >> 5, local row
 * numbering, ceil(ntotal / 32) words -- can be returned; NULL = all rows.  Fewer than
 * k allowed rows: padded like css_index_search. */
int css_index_search_masked(css_index* ix, const float* q_host, int64_t nq, int k, int normalize_q,
                            const uint32_t* allow_bits_host, float* D_host, int64_t* I_host);
int css_index_search_masked_dev(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q,
                                const uint32_t* allow_bits_dev, float* D_dev, int64_t* I_dev,
                                void* stream);

/* Merge `nparts` per-shard results ([nparts, nq, k] each) into the global
 * top-k by (score, id); used after the RCCL all-gather of per-shard top-k. */
int css_merge_topk_dev(const float* D_parts_dev, const int64_t* I_parts_dev, int nparts,
                       int64_t nq, int k, int metric, float* D_out_dev, int64_t* I_out_dev,
                       int device, void* stream);

/* The same merge straight from the exchange buffer of the sharded search: `nparts` records of `record_bytes`
 * bytes (a multiple of 8, >= 12 * nq * k), each [nq * k int64 ids][nq * k float scores] -- the layout one
 * RCCL all-gather of nq * k * 12 bytes per rank produces (a single exchange step). */
int css_merge_topk_packed_dev(const void* packed_dev, int nparts, int64_t record_bytes, int64_t nq, int k,
                              int metric, float* D_out_dev, int64_t* I_out_dev, int device, void* stream);

/* ---- MPNet sentence encoder (all-mpnet-base-v2 architecture) ---- */
typedef struct css_encoder_cfg {
    int num_layers;       /* 12 */
    int hidden;           /* 768 */
    int heads;            /* 12 (head_dim = hidden / heads must be 64) */
    int ffn;              /* 3072 */
    int vocab;            /* 30527 */
    int max_pos;          /* 514 */
    int rel_buckets;      /* 32 */
    int pad_id;           /* 1 */
    int max_seq_len;      /* 384 (kernel limit 512) */
    float ln_eps;         /* 1e-5 */
    int compute;          /* 0 = bf16 MFMA (product), 1 = fp32 verification mode */
} css_encoder_cfg;

typedef struct css_tensor {
    const char* name;     /* HF key, e.g. "encoder.layer.0.attention.attn.q.weight" */
    const float* data;    /* host fp32, row-major */
    int64_t numel;
} css_tensor;

int css_encoder_create(const css_encoder_cfg* cfg, int device, css_encoder** out);
int css_encoder_free(css_encoder* enc);
int css_encoder_load_weights(css_encoder* enc, const css_tensor* tensors, int n);
/* Seeded synthetic weights generated on the device ("synthetic weights"). */
int css_encoder_init_synthetic(css_encoder* enc, uint64_t seed);
/* Copy one named parameter (fp32 master copy) back to the host. */
int css_encoder_export_weight(const css_encoder* enc, const char* name, float* out_host, int64_t numel);
/* Packed var-len batch: input_ids[cu_seqlens[B]] tokens, sequence b occupies
 * [cu_seqlens[b], cu_seqlens[b+1]); every length in [1, max_seq_len].
 * out: [B, hidden] fp32 = masked mean-pool (+ L2 normalise when normalize != 0).
 * The same batch gives the same bits on every run (row statistics of the folded
 * LayerNorm are accumulated with integer atomics); another batch composition may take
 * another kernel path (GEMM tile walk, folded / separate LayerNorm): bf16 rounding noise. */
int css_encoder_forward(css_encoder* enc, const int32_t* input_ids_host, const int32_t* cu_seqlens_host,
                        int B, int normalize, float* out_host);
int css_encoder_forward_dev(css_encoder* enc, const int32_t* input_ids_dev, const int32_t* cu_seqlens_dev,
                            int B, int total_tokens, int max_len, int normalize, float* out_dev, void* stream);

/* bf16 attention computes softmax rows as exp2(score) / sum WITHOUT a running maximum while every row sum of a
 * block stays in (1 / range, range) and repeats the block with the running-maximum (online) softmax otherwise --
 * same result, the guard only protects the fp32 range.  Default 2^100 (|logit| < 69); range = 0 always takes the
 * running-maximum pass (verification; env default CSS_ATT_RANGE). */
int css_encoder_set_attention_range(css_encoder* enc, float range);

/* Test/diagnostic hook: copy an activation buffer of the LAST forward back to the
 * host as fp32 ("x32" [T,H] final hidden states, "qkv" [T,3H], "ctx" [T,H],
 * "ffn" [T,F], "pre32" [T,H]; with num_layers = 1 these are the layer-0 probes).
 * bf16 batches of >= 1024 tokens run with LayerNorm folded into the GEMM epilogues and
 * never materialise "x32" / "pre32" as such: "x32" is then an error (CSS_ENC_FUSE_LN=0
 * keeps the separate LayerNorm kernels), "pre32" / "ctx" / "ffn" hold that path's buffers. */
int css_encoder_debug_read(css_encoder* enc, const char* what, float* out_host, int64_t numel);

/* Host-only helper (no device needed): bucket of a relative position
 * rel = key - query, as transformers' MPNetEncoder.relative_position_bucket
 * (the encoder builds its per-head Toeplitz bias table from it). */
int css_mpnet_rel_bucket(int rel, int num_buckets, int max_distance);

/* ---- WordPiece front end of encode() (host code; the tokenizer half of
 * SentenceTransformer.encode, src/embeddings.py:184-188, :216-222) ----
 * vocab.txt in HF layout (one piece per line, id = line number).  encode_batch:
 * `bytes` holds the n UTF-8 texts back to back, text i = [offsets[i], offsets[i+1]);
 * ids_out is [n, max_len] (padded with <pad>), lens_out[i] the token count incl.
 * <s> and </s>, or -1 for a text the tables cannot express (invalid UTF-8, a capital
 * sigma, non-ASCII text with lower-casing off), which the caller tokenises with the
 * Python implementation of the same pipeline.  nthreads <= 0: all cores. */
typedef struct css_tokenizer css_tokenizer;
int css_tokenizer_create(const char* vocab_path, int lowercase, css_tokenizer** out);
int css_tokenizer_free(css_tokenizer* t);
int css_tokenizer_vocab_size(const css_tokenizer* t, int* n);
int css_tokenizer_encode_batch(const css_tokenizer* t, const char* bytes, const int64_t* offsets,
                               int64_t n, int max_len, int32_t* ids_out, int32_t* lens_out,
                               int nthreads);

/* ---- in-library kernel timing (HIP events on the launch stream) ---- */
/* When enabled, each launch of a named dominant kernel is bracketed by HIP
 * events on the stream it is launched on; css_prof_read drains and sums them. */
int css_prof_enable(int on);
int css_prof_reset(void);
int css_prof_read(const char* kernel, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* CSS_HIP_H */
