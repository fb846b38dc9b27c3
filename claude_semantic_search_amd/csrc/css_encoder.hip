// css_encoder.hip -- MPNet sentence encoder (all-mpnet-base-v2 architecture) on gfx950.
//
// Replaces, behind include/css_hip.h, what the reference reaches through
// SentenceTransformer.encode() (src/embeddings.py:184-188, :216-222): MPNet
// encoder forward -> masked mean pooling -> L2 normalise.
// Tokenisation stays on the host (claude_semantic_search_amd/tokenizer.py); the
// boundary here is (packed input_ids, cu_seqlens).
//
// Pipeline per forward (one stream, no host syncs between kernels):
//   k_embed_ln -> 12 x [ k_gemm<QKV> -> attention -> k_gemm<RESID> -> k_layernorm
//                        -> k_gemm<GELU> -> k_gemm<RESID> -> k_layernorm ] -> k_pool_norm
// Residual stream / LayerNorm / softmax / pooling are fp32; GEMM and attention
// operands are bf16 (MFMA) in the product mode, fp32 in the verification mode.
#include "css_common.h"
#include "css_encoder_kernels.h"
#include "../../include/css_synth.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace css;

namespace {

struct Param {
    float* p = nullptr;
    int64_t numel = 0;
    uint32_t synth_id = 0;
    float mean = 0.f, std = 0.02f;
};

struct LayerW {
    float *wqkv, *bqkv, *wo, *bo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
    bf16_t *wqkv_h, *wo_h, *w1_h, *w2_h;
    // LayerNorm-folded operands (k_fold_ln): the QKV weight carries gamma of the LayerNorm BEFORE this layer
    // (embedding LN / previous layer's output LN), the FFN1 weight gamma of this layer's attention LN
    bf16_t *wqkv_f = nullptr, *w1_f = nullptr;
    bf16_t *w2_p = nullptr, *wo_p = nullptr;   // W2 / Wo in bf16 with the K order of their blocked A operands (k_f32_to_bf16_kperm)
    float *dqkv = nullptr, *d1 = nullptr;   // d rows of the two folded GEMMs
    float *bo_f = nullptr, *b2_f = nullptr; // bias + beta of the LayerNorm whose output is the residual (EPI_RES)
};

// Process-wide switches, read from the environment ONCE (thread-safe static init); all are A/B-experiment knobs,
// the defaults are the product configuration.
struct EncEnv {
    int resid = 2;        // CSS_ENC_RESID: residual stream storage in bf16 mode (see forward_typed)
    int gemm4w = 0;       // CSS_GEMM_4W: bit mask of the folded GEMMs that run on k_gemm4w (1 QKV, 2 O, 4 FFN1, 8 FFN2)
    int dbg = 0;          // CSS_GEMM_DBG bit0: skip epilogue, bit1: skip MFMA, bit2: skip loads (timing experiments)
    bool big_tiles = true;   // CSS_GEMM_TILE=128: the 128x128 tiles everywhere
    bool mfma16 = true;      // CSS_GEMM_MFMA=32: 32x32x16 MFMA kernel (k_gemm) instead of k_gemm16
    bool loop8 = true;       // CSS_GEMM_LOOP=old: the round-1 main loop (k_gemm16) instead of k_gemm8p
    bool fuse_ln = true;     // CSS_ENC_FUSE_LN=0: separate LayerNorm kernels also for large batches
    float att_range = -1.f;  // CSS_ATT_RANGE: default of css_encoder_set_attention_range (0 = always the safe softmax pass)
    int cg_qkv = 0, cg_ffn1 = 0, cg_o = 0, cg_ffn2 = 0;   // CSS_GEMM_CG="qkv,ffn1,o,ffn2": column-group tile walk of k_gemm8p (0 = N-fastest)
    int grid_qkv = 0, grid_ffn1 = 0, grid_o = 0, grid_ffn2 = 0;   // CSS_GEMM_GRID="qkv,ffn1,o,ffn2": blocks of k_gemm8p (0 = automatic)
    EncEnv() {
        if (const char* t = getenv("CSS_GEMM_TILE")) big_tiles = atoi(t) != 128;
        if (const char* t = getenv("CSS_GEMM_DBG")) dbg = atoi(t);
        if (const char* t = getenv("CSS_ENC_RESID")) resid = atoi(t);
        if (const char* t = getenv("CSS_GEMM_4W")) gemm4w = atoi(t);
        if (const char* t = getenv("CSS_GEMM_MFMA")) mfma16 = atoi(t) != 32;
        if (const char* t = getenv("CSS_GEMM_LOOP")) loop8 = std::string(t) != "old";
        if (const char* t = getenv("CSS_ENC_FUSE_LN")) fuse_ln = atoi(t) != 0;
        if (const char* t = getenv("CSS_ATT_RANGE")) att_range = (float)atof(t);
        if (const char* t = getenv("CSS_GEMM_CG")) sscanf(t, "%d,%d,%d,%d", &cg_qkv, &cg_ffn1, &cg_o, &cg_ffn2);
        if (const char* t = getenv("CSS_GEMM_GRID")) sscanf(t, "%d,%d,%d,%d", &grid_qkv, &grid_ffn1, &grid_o, &grid_ffn2);
    }
};
const EncEnv& enc_env() {
    static const EncEnv env;
    return env;
}

__global__ void k_synth_fill(float* p, size_t n, uint64_t seed, float mean, float std) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = mean + std * css_synth_normal(seed, (uint64_t)i);
}

__global__ void k_build_bias_tab(const float* __restrict__ relw, const int* __restrict__ bucket, int heads, int maxL,
                                 float scale, float* __restrict__ tab) {
    const int n = 2 * maxL - 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * heads) return;
    const int h = i / n, r = i - h * n;
    tab[i] = scale * relw[bucket[r] * heads + h];  // bf16 path: log2(e) (scores in the log2 domain)
}

__global__ void k_zero_row(float* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

}  // namespace

// Bucket of a relative position (rel = key - query), transformers'
// MPNetEncoder.relative_position_bucket (modeling_mpnet.py:312-348 in 5.15.0).
// Evaluated in double: the only integer-valued cases are |rel| = 16, 32, 64 where
// log(n/8)/log(16)*8 is exact in double as well (tests/test_encoder_host.py pins
// all 1023 values against the transformers implementation).
extern "C" int css_mpnet_rel_bucket(int rel, int num_buckets, int max_distance) {
    int n = -rel;
    const int half = num_buckets / 2;
    int ret = n < 0 ? half : 0;
    n = n < 0 ? -n : n;
    const int max_exact = half / 2;
    if (n < max_exact) return ret + n;
    const double v = std::log((double)n / max_exact) / std::log((double)max_distance / max_exact) * (half - max_exact);
    int large = max_exact + (int)v;
    if (large > half - 1) large = half - 1;
    return ret + large;
}

struct css_encoder {
    css_encoder_cfg cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, Param> params;
    float *wemb = nullptr, *pemb = nullptr, *embg = nullptr, *embb = nullptr, *relw = nullptr;
    std::vector<LayerW> layers;
    float* bias_tab = nullptr;  // [heads][2*maxL-1]
    int* bucket_dev = nullptr;
    bool weights_ready = false;
    // activations (capacity in tokens / sequences)
    int cap_tokens = 0, cap_seqs = 0, num_cus = 256;
    float *x32 = nullptr, *pre32 = nullptr;          // [T, H] fp32
    void *x16 = nullptr, *qkv = nullptr, *ctx = nullptr, *ffn = nullptr;  // operand-typed
    long long* stats[2] = {nullptr, nullptr};        // [T][2] fixed-point row sums of the two pre tensors (folded-LN path)
    bool x32_valid = true;                           // false after a folded-LN forward (x32 is not materialised there)
    float att_range = 1.2676506e30f;                 // 2^100: row-sum range of the reference-free softmax pass (attn_pass); 0 = always the SAFE pass
    int32_t *ids_dev = nullptr, *cu_dev = nullptr;
    float* out_dev = nullptr;
    // hipGraph cache for the launch-bound tiny-batch path (generate_single_embedding): key =
    // (B, T, max_len, normalize); a key is run eagerly once, captured on its second use, replayed after
    struct GraphEntry {
        int uses = 0;
        hipGraphExec_t exec = nullptr;
    };
    std::map<uint64_t, GraphEntry> graphs;
    bool graphs_ok = true;
    std::mutex mu;
};

namespace {

int alloc_param(css_encoder* e, const std::string& name, int64_t numel, uint32_t sid, float mean, float std,
                float** out) {
    float* p = nullptr;
    hipError_t err = hipMalloc((void**)&p, (size_t)numel * sizeof(float));
    if (err != hipSuccess) return css::hip_fail(err, "hipMalloc(weights)", __FILE__, __LINE__);
    Param prm;
    prm.p = p;
    prm.numel = numel;
    prm.synth_id = sid;
    prm.mean = mean;
    prm.std = std;
    e->params[name] = prm;
    *out = p;
    return CSS_OK;
}

// HF key names.  q/k/v are views into the fused
// [3H, H] weight / [3H] bias, in that order.
int build_params(css_encoder* e) {
    const css_encoder_cfg& c = e->cfg;
    const int64_t H = c.hidden, F = c.ffn;
    int rc;
#define AP(name, n, sid, mean, std, ptr) \
    if ((rc = alloc_param(e, name, n, sid, mean, std, ptr)) != CSS_OK) return rc
    AP("embeddings.word_embeddings.weight", (int64_t)c.vocab * H, 0, 0.f, 0.02f, &e->wemb);
    AP("embeddings.position_embeddings.weight", (int64_t)c.max_pos * H, 1, 0.f, 0.02f, &e->pemb);
    AP("embeddings.LayerNorm.weight", H, 2, 1.f, 0.1f, &e->embg);
    AP("embeddings.LayerNorm.bias", H, 3, 0.f, 0.05f, &e->embb);
    AP("encoder.relative_attention_bias.weight", (int64_t)c.rel_buckets * c.heads, 4, 0.f, 0.1f, &e->relw);
    e->layers.resize(c.num_layers);
    for (int i = 0; i < c.num_layers; ++i) {
        LayerW& L = e->layers[i];
        const std::string pre = "encoder.layer." + std::to_string(i) + ".";
        const uint32_t base = 16 + 16 * i;
        AP(pre + "attention.attn.qkv.weight", 3 * H * H, base + 0, 0.f, 0.02f, &L.wqkv);
        AP(pre + "attention.attn.qkv.bias", 3 * H, base + 1, 0.f, 0.05f, &L.bqkv);
        AP(pre + "attention.attn.o.weight", H * H, base + 6, 0.f, 0.02f, &L.wo);
        AP(pre + "attention.attn.o.bias", H, base + 7, 0.f, 0.05f, &L.bo);
        AP(pre + "attention.LayerNorm.weight", H, base + 8, 1.f, 0.1f, &L.ln1g);
        AP(pre + "attention.LayerNorm.bias", H, base + 9, 0.f, 0.05f, &L.ln1b);
        AP(pre + "intermediate.dense.weight", F * H, base + 10, 0.f, 0.02f, &L.w1);
        AP(pre + "intermediate.dense.bias", F, base + 11, 0.f, 0.05f, &L.b1);
        AP(pre + "output.dense.weight", H * F, base + 12, 0.f, 0.02f, &L.w2);
        AP(pre + "output.dense.bias", H, base + 13, 0.f, 0.05f, &L.b2);
        AP(pre + "output.LayerNorm.weight", H, base + 14, 1.f, 0.1f, &L.ln2g);
        AP(pre + "output.LayerNorm.bias", H, base + 15, 0.f, 0.05f, &L.ln2b);
        // q/k/v views (weights: rows [0,H) [H,2H) [2H,3H) of the fused matrix)
        const char* nm[3] = {"q", "k", "v"};
        for (int j = 0; j < 3; ++j) {
            Param w;
            w.p = L.wqkv + (size_t)j * H * H;
            w.numel = H * H;
            w.synth_id = base + 2 * j;  // informational; synthetic init fills the fused tensors
            e->params[pre + "attention.attn." + nm[j] + ".weight"] = w;
            Param b;
            b.p = L.bqkv + (size_t)j * H;
            b.numel = H;
            e->params[pre + "attention.attn." + nm[j] + ".bias"] = b;
        }
        CSS_HIP_TRY(hipMalloc((void**)&L.wqkv_h, (size_t)3 * H * H * 2));
        CSS_HIP_TRY(hipMalloc((void**)&L.wo_h, (size_t)H * H * 2));
        CSS_HIP_TRY(hipMalloc((void**)&L.w1_h, (size_t)F * H * 2));
        CSS_HIP_TRY(hipMalloc((void**)&L.w2_h, (size_t)H * F * 2));
        if (c.compute == 0) {
            CSS_HIP_TRY(hipMalloc((void**)&L.wqkv_f, (size_t)3 * H * H * 2));
            CSS_HIP_TRY(hipMalloc((void**)&L.w1_f, (size_t)F * H * 2));
            CSS_HIP_TRY(hipMalloc((void**)&L.w2_p, (size_t)H * F * 2));
            CSS_HIP_TRY(hipMalloc((void**)&L.wo_p, (size_t)H * H * 2));
            CSS_HIP_TRY(hipMalloc((void**)&L.dqkv, (size_t)3 * H * 4));
            CSS_HIP_TRY(hipMalloc((void**)&L.d1, (size_t)F * 4));
            CSS_HIP_TRY(hipMalloc((void**)&L.bo_f, (size_t)H * 4));
            CSS_HIP_TRY(hipMalloc((void**)&L.b2_f, (size_t)H * 4));
        }
    }
#undef AP
    const int maxL = c.max_seq_len;
    CSS_HIP_TRY(hipMalloc((void**)&e->bias_tab, (size_t)c.heads * (2 * maxL - 1) * sizeof(float)));
    CSS_HIP_TRY(hipMalloc((void**)&e->bucket_dev, (size_t)(2 * maxL - 1) * sizeof(int)));
    std::vector<int> bucket(2 * maxL - 1);
    for (int r = 0; r < 2 * maxL - 1; ++r) bucket[r] = css_mpnet_rel_bucket(r - (maxL - 1), c.rel_buckets, 128);
    CSS_HIP_TRY(hipMemcpy(e->bucket_dev, bucket.data(), bucket.size() * sizeof(int), hipMemcpyHostToDevice));
    return CSS_OK;
}

// After the fp32 masters change: bf16 operand copies + the per-head Toeplitz bias table.
int finalize_weights(css_encoder* e) {
    const css_encoder_cfg& c = e->cfg;
    const size_t H = c.hidden, F = c.ffn;
    hipStream_t st = e->stream;
    for (size_t li = 0; li < e->layers.size(); ++li) {
        LayerW& L = e->layers[li];
        if (L.wqkv_f) {
            const float* g_in = li == 0 ? e->embg : e->layers[li - 1].ln2g;
            const float* b_in = li == 0 ? e->embb : e->layers[li - 1].ln2b;
            hipLaunchKernelGGL(k_fold_ln, dim3((3 * H + 3) / 4), dim3(256), 0, st, L.wqkv, g_in, b_in, L.bqkv, (int)(3 * H),
                               (int)H, L.wqkv_f, L.dqkv, 1);
            hipLaunchKernelGGL(k_fold_ln, dim3((F + 3) / 4), dim3(256), 0, st, L.w1, L.ln1g, L.ln1b, L.b1, (int)F, (int)H,
                               L.w1_f, L.d1, 1);
            hipLaunchKernelGGL(k_add_vec, dim3((H + 255) / 256), dim3(256), 0, st, L.bo, b_in, L.bo_f, (int)H);
            hipLaunchKernelGGL(k_add_vec, dim3((H + 255) / 256), dim3(256), 0, st, L.b2, L.ln1b, L.b2_f, (int)H);
            hipLaunchKernelGGL(k_f32_to_bf16_kperm, dim3(1024), dim3(256), 0, st, L.w2, L.w2_p, H * F, (int)F);
            hipLaunchKernelGGL(k_f32_to_bf16_kperm, dim3(1024), dim3(256), 0, st, L.wo, L.wo_p, H * H, (int)H);
        }
    }
    for (auto& L : e->layers) {
        hipLaunchKernelGGL(k_f32_to_bf16, dim3(1024), dim3(256), 0, st, L.wqkv, L.wqkv_h, 3 * H * H);
        hipLaunchKernelGGL(k_f32_to_bf16, dim3(1024), dim3(256), 0, st, L.wo, L.wo_h, H * H);
        hipLaunchKernelGGL(k_f32_to_bf16, dim3(1024), dim3(256), 0, st, L.w1, L.w1_h, F * H);
        hipLaunchKernelGGL(k_f32_to_bf16, dim3(1024), dim3(256), 0, st, L.w2, L.w2_h, H * F);
    }
    const int n = c.heads * (2 * c.max_seq_len - 1);
    hipLaunchKernelGGL(k_build_bias_tab, dim3((n + 255) / 256), dim3(256), 0, st, e->relw, e->bucket_dev, c.heads,
                       c.max_seq_len, c.compute == 0 ? 1.44269504088896341f : 1.0f, e->bias_tab);
    CSS_LAUNCH_CHECK();
    CSS_HIP_TRY(hipStreamSynchronize(st));
    e->weights_ready = true;
    return CSS_OK;
}

int ensure_acts(css_encoder* e, int T, int B) {
    const css_encoder_cfg& c = e->cfg;
    const size_t H = c.hidden, F = c.ffn;
    const size_t es = c.compute == 0 ? 2 : 4;
    T = std::max(T, 8 * B);  // pre32 also holds the [B][8][H] pooling partials
    if (T > e->cap_tokens || B > e->cap_seqs) {
        for (auto& kv : e->graphs)
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        e->graphs.clear();  // captured pointers are about to change
    }
    if (T > e->cap_tokens) {
        void* ptrs[] = {e->x32, e->pre32, e->x16, e->qkv, e->ctx, e->ffn, e->ids_dev, e->stats[0], e->stats[1]};
        for (void* p : ptrs)
            if (p) CSS_HIP_TRY(hipFree(p));
        e->cap_tokens = 0;
        // GEMM tiles store whole 256-row tiles unconditionally: keep >= 256 slack rows
        const size_t cap = ((size_t)T + 255) / 256 * 256 + 256;
        CSS_HIP_TRY(hipMalloc((void**)&e->x32, cap * H * 4));
        CSS_HIP_TRY(hipMalloc((void**)&e->pre32, cap * H * 4));
        // (+ cap * 64 bytes: the padding of the blocked layout the LayerNorm-folded path keeps in x16 / pre32 / ffn)
        CSS_HIP_TRY(hipMalloc((void**)&e->x16, cap * H * es + cap * 64));
        CSS_HIP_TRY(hipMalloc((void**)&e->qkv, cap * 3 * H * es));
        CSS_HIP_TRY(hipMalloc((void**)&e->ctx, cap * H * es));
        CSS_HIP_TRY(hipMalloc((void**)&e->ffn, cap * F * es + cap * 64));
        CSS_HIP_TRY(hipMalloc((void**)&e->ids_dev, cap * sizeof(int32_t)));
        e->stats[0] = e->stats[1] = nullptr;
        if (c.compute == 0) {
            for (int i = 0; i < 2; ++i) {
                CSS_HIP_TRY(hipMalloc((void**)&e->stats[i], cap * 16));
                CSS_HIP_TRY(hipMemset(e->stats[i], 0, cap * 16));  // slack rows: finite values
            }
            // the memsets run on the null stream, forwards on non-blocking streams: order them here, once per growth
            CSS_HIP_TRY(hipDeviceSynchronize());
        }
        e->cap_tokens = (int)cap;
    }
    if (B > e->cap_seqs) {
        if (e->cu_dev) CSS_HIP_TRY(hipFree(e->cu_dev));
        if (e->out_dev) CSS_HIP_TRY(hipFree(e->out_dev));
        e->cap_seqs = 0;
        const size_t cap = (size_t)B + 64;
        CSS_HIP_TRY(hipMalloc((void**)&e->cu_dev, (cap + 1) * sizeof(int32_t)));
        CSS_HIP_TRY(hipMalloc((void**)&e->out_dev, cap * H * 4));
        e->cap_seqs = (int)cap;
    }
    return CSS_OK;
}

// Tile shapes: 2x2 waves x (2x2) MFMA tiles = 128x128, or 2x4 waves x (4x2) tiles = 256x256
// (8 waves, 128 KiB ring).  Persistent: one block per CU (grid a multiple of 8).
template <typename TIn, int EPI, int WM, int WN, int TM, int TN, int NST, int RB, int SPS>
int launch_gemm_t(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int qscale_cols,
                  float qscale, int num_cus, hipStream_t st, const char* prof) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    CSS_REQUIRE(N % BN == 0 && K % 64 == 0 && K / 64 >= 3, "gemm: N=%d must be a multiple of %d and K=%d of 64 (>= 192)", N, BN, K);
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    auto kern = k_gemm<TIn, EPI, WM, WN, TM, TN, NST, RB, SPS>;
    const size_t lds = (size_t)NST * (BM + BN) * RB;  // ring of NST stages, RB bytes of K per row
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int rc_ = css::ensure_dynamic_lds((const void*)kern, lds, dev_);   // (mutex-protected, once per kernel and device)
    if (rc_ != CSS_OK) return rc_;
    const int blocks_per_cu = lds <= 80 * 1024 ? 2 : 1;
    int grid = std::min(ntn * ntm, num_cus * blocks_per_cu);
    grid = std::max(8, grid / 8 * 8);
    ProfScope ps(prof, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, st, (const TIn*)A, (const TIn*)W, bias, C, M, N, K,
                       qscale_cols, qscale, enc_env().dbg);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

// k_gemm8p launch (256x256 tiles, persistent, one block per CU); `side` carries the LayerNorm-folding operands
template <int EPI, bool ABLK = false, int TAG = 0>
int launch_gemm8p(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int qscale_cols,
                  float qscale, const G8Side& side, int num_cus, hipStream_t st, const char* prof) {
    CSS_REQUIRE(N % 256 == 0 && K % 128 == 0 && (size_t)M * K * 2 < ((size_t)1 << 32), "gemm8p: bad shape %d x %d x %d", M, N, K);
    CSS_REQUIRE(EPI != EPI_RES || K >= 256, "gemm8p: EPI_RES needs K >= 256 (statistics are flushed in a tile's third K step)");
    auto kern = k_gemm8p<EPI, ABLK, TAG>;
    constexpr size_t lds = 2 * 4 * G8_HT;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int rc_ = css::ensure_dynamic_lds((const void*)kern, lds, dev_);
    if (rc_ != CSS_OK) return rc_;
    const int ntiles = (N / 256) * ((M + 255) / 256);
    int grid = std::min(ntiles, side.grid > 0 ? std::min(side.grid, num_cus) : num_cus);
    grid = std::max(8, grid / 8 * 8);
    ProfScope ps(prof, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K,
                       qscale_cols, qscale, side);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}
// k_gemm4w launch (the LayerNorm-folded GEMMs on four 128 x 128 waves per block; CSS_GEMM_4W=0 keeps k_gemm8p)
template <int EPI, int TAG = 0>
int launch_gemm4w(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int qscale_cols,
                  float qscale, const G8Side& side, int num_cus, hipStream_t st, const char* prof) {
    CSS_REQUIRE(N % 256 == 0 && K % 64 == 0 && K >= 256 && (size_t)M * K * 2 < ((size_t)1 << 32), "gemm4w: bad shape %d x %d x %d", M, N, K);
    auto kern = k_gemm4w<EPI, TAG>;
    constexpr size_t lds = 2 * 65536;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int rc_ = css::ensure_dynamic_lds((const void*)kern, lds, dev_);
    if (rc_ != CSS_OK) return rc_;
    const int ntiles = (N / 256) * ((M + 255) / 256);
    int grid = std::min(ntiles, side.grid > 0 ? std::min(side.grid, num_cus) : num_cus);
    grid = std::max(8, grid / 8 * 8);
    ProfScope ps(prof, st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, (const bf16_t*)A, (const bf16_t*)W, bias, (bf16_t*)C, M, N, K,
                       qscale_cols, qscale, side);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}
// CSS_GEMM_TILE=128 selects the 128x128 variant (A/B experiments)

template <typename TIn, int EPI>
int launch_gemm(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int qscale_cols,
                float qscale, int num_cus, hipStream_t st, const char* prof) {
    if (M <= 64 && N % 32 == 0 && K % 768 == 0) {  // single-query / tiny batches: weight-streaming kernel (K/4 = 12n fragment steps)
        ProfScope ps(prof, st);
        if (M <= 32)
            hipLaunchKernelGGL((k_gemm_skinny<TIn, EPI, 1>), dim3(N / 32), dim3(256), 0, st, (const TIn*)A, (const TIn*)W,
                               bias, C, M, N, K, qscale_cols, qscale);
        else
            hipLaunchKernelGGL((k_gemm_skinny<TIn, EPI, 2>), dim3(N / 32), dim3(256), 0, st, (const TIn*)A, (const TIn*)W,
                               bias, C, M, N, K, qscale_cols, qscale);
        CSS_LAUNCH_CHECK();
        return CSS_OK;
    }
    const EncEnv& env = enc_env();
    if (env.big_tiles && M >= 1024 && N % 256 == 0) {
        // ring: 2 stages x 128 B rows.  Rings of 3 / 4 x 64 B rows and 5 x 64 B with two stages per step were
        // measured slower (more barriers, same LDS fill rate) and are not kept.
        if constexpr (sizeof(TIn) == 2 && (EPI == EPI_QKV || EPI == EPI_GELU)) {
            if (env.loop8 && K % 128 == 0 && (size_t)M * K * 2 < ((size_t)1 << 32))
                return launch_gemm8p<EPI>(A, W, bias, C, M, N, K, qscale_cols, qscale, G8Side{}, num_cus, st, prof);
        }
        if constexpr (sizeof(TIn) == 2) {
            if (env.mfma16) {  // product mode: the 16x16x32 MFMA variant (CSS_GEMM_MFMA=32 selects k_gemm for A/B runs)
                CSS_REQUIRE(K % 64 == 0 && K / 64 >= 3, "gemm: K=%d must be a multiple of 64 (>= 192)", K);
                auto kern = k_gemm16<EPI>;
                constexpr size_t lds = 2 * 512 * 128;
                int dev_ = 0;
                (void)hipGetDevice(&dev_);
                int rc_ = css::ensure_dynamic_lds((const void*)kern, lds, dev_);
                if (rc_ != CSS_OK) return rc_;
                const int ntiles = (N / 256) * ((M + 255) / 256);
                int grid = std::min(ntiles, num_cus);
                grid = std::max(8, grid / 8 * 8);
                ProfScope ps(prof, st);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, (const bf16_t*)A, (const bf16_t*)W, bias, C, M, N, K,
                                   qscale_cols, qscale);
                CSS_LAUNCH_CHECK();
                return CSS_OK;
            }
        }
        return launch_gemm_t<TIn, EPI, 2, 4, 4, 2, 2, 128, 1>(A, W, bias, C, M, N, K, qscale_cols, qscale, num_cus, st, prof);
    }
    return launch_gemm_t<TIn, EPI, 2, 2, 2, 2, 4, 64, 1>(A, W, bias, C, M, N, K, qscale_cols, qscale, num_cus, st, prof);
}

template <typename TIn>
int forward_typed(css_encoder* e, const int32_t* ids, const int32_t* cu, int B, int T, int max_len, int normalize,
                  float* out, hipStream_t st) {
    const css_encoder_cfg& c = e->cfg;
    const int H = c.hidden, F = c.ffn;
    constexpr bool BF = sizeof(TIn) == 2;
    int rc;
    {
        ProfScope ps("enc_embed_ln", st);
        hipLaunchKernelGGL(k_embed_ln<768>, dim3((T + 3) / 4), dim3(256), 0, st, ids, cu, B, e->wemb, e->pemb,
                           e->embg, e->embb, c.ln_eps, c.vocab, c.max_pos, e->x32, BF ? (bf16_t*)e->x16 : nullptr, T);
        CSS_LAUNCH_CHECK();
    }
    const int maxL = c.max_seq_len;
    // Residual stream storage (bf16 mode): CSS_ENC_RESID 0 = fp32 rows (x32) + fp32 branch outputs,
    // 1 = bf16 residual (the row the next GEMM reads anyway) + fp32 branch outputs, 2 = both bf16.
    const int rmode = BF ? enc_env().resid : 0;
    // attention-output / FFN2 projection into `pre32` (fp32, or bf16 rows when rmode == 2)
    auto launch_branch_out = [&](const void* a, const void* w, const float* bias, int K, const char* prof) -> int {
        if (rmode == 2) return launch_gemm<TIn, EPI_QKV>(a, w, bias, e->pre32, T, H, K, 0, 1.0f, e->num_cus, st, prof);
        return launch_gemm<TIn, EPI_RESID>(a, w, bias, e->pre32, T, H, K, 0, 1.0f, e->num_cus, st, prof);
    };
    // x = LN(pre + x); `last`: the fp32 row is needed by the pooling
    auto launch_ln = [&](const float* g, const float* b, bool last) -> int {
        ProfScope ps("enc_layernorm", st);
        const dim3 grid((T + 3) / 4), blk(256);
        bf16_t* o16 = BF ? (bf16_t*)e->x16 : nullptr;
        if (rmode == 0)
            hipLaunchKernelGGL((k_layernorm<768, float, float>), grid, blk, 0, st, (const float*)e->pre32, (const float*)e->x32, g, b,
                               c.ln_eps, e->x32, o16, T);
        else if (rmode == 1)
            hipLaunchKernelGGL((k_layernorm<768, float, bf16_t>), grid, blk, 0, st, (const float*)e->pre32, (const bf16_t*)e->x16, g, b,
                               c.ln_eps, last ? e->x32 : (float*)nullptr, o16, T);
        else
            hipLaunchKernelGGL((k_layernorm<768, bf16_t, bf16_t>), grid, blk, 0, st, (const bf16_t*)e->pre32, (const bf16_t*)e->x16, g, b,
                               c.ln_eps, last ? e->x32 : (float*)nullptr, o16, T);
        CSS_LAUNCH_CHECK();
        return CSS_OK;
    };
    for (int li = 0; li < c.num_layers; ++li) {
        const LayerW& L = e->layers[li];
        const void* xin = BF ? e->x16 : (const void*)e->x32;
        const void* wqkv = BF ? (const void*)L.wqkv_h : (const void*)L.wqkv;
        if ((rc = launch_gemm<TIn, EPI_QKV>(xin, wqkv, L.bqkv, e->qkv, T, 3 * H, H, H, BF ? 0.125f * 1.44269504088896341f : 0.125f, e->num_cus, st, "enc_gemm_qkv")) != CSS_OK)
            return rc;
        {
            ProfScope ps("enc_attention", st);
            if constexpr (BF) {
                const size_t lds = 4 * 8192 + (size_t)(2 * maxL - 1 + 64) * 4;
                const int nqb = (max_len + 127) / 128;
                hipLaunchKernelGGL((k_attention_bf16<64, false>), dim3(B * nqb * c.heads), dim3(256), lds, st,
                                   (const bf16_t*)e->qkv, cu, e->bias_tab, maxL, H, (bf16_t*)e->ctx, nqb, c.heads, e->att_range);
            } else {
                hipLaunchKernelGGL(k_attention_f32, dim3(B, max_len, c.heads), dim3(64), 0, st, (const float*)e->qkv, cu,
                                   e->bias_tab, maxL, H, (float*)e->ctx);
            }
            CSS_LAUNCH_CHECK();
        }
        const void* wo = BF ? (const void*)L.wo_h : (const void*)L.wo;
        if ((rc = launch_branch_out(e->ctx, wo, L.bo, H, "enc_gemm_o")) != CSS_OK) return rc;
        if ((rc = launch_ln(L.ln1g, L.ln1b, false)) != CSS_OK) return rc;
        xin = BF ? e->x16 : (const void*)e->x32;
        const void* w1 = BF ? (const void*)L.w1_h : (const void*)L.w1;
        if ((rc = launch_gemm<TIn, EPI_GELU>(xin, w1, L.b1, e->ffn, T, F, H, 0, 1.0f, e->num_cus, st, "enc_gemm_ffn1")) != CSS_OK)
            return rc;
        const void* w2 = BF ? (const void*)L.w2_h : (const void*)L.w2;
        if ((rc = launch_branch_out(e->ffn, w2, L.b2, F, "enc_gemm_ffn2")) != CSS_OK) return rc;
        if ((rc = launch_ln(L.ln2g, L.ln2b, li == c.num_layers - 1)) != CSS_OK) return rc;
    }
    {
        ProfScope ps("enc_pool", st);
        constexpr int kPoolSlices = 8;
        // partial sums live in pre32 (free after the last LayerNorm): [B][8][H] floats
        hipLaunchKernelGGL(k_pool_partial<768>, dim3(B, kPoolSlices), dim3(256), 0, st, e->x32, cu, kPoolSlices, e->pre32);
        hipLaunchKernelGGL(k_pool_final<768>, dim3(B), dim3(256), 0, st, e->pre32, cu, kPoolSlices, normalize, out);
        CSS_LAUNCH_CHECK();
    }
    return CSS_OK;
}

// bf16 product path for batches of >= 1024 tokens: LayerNorm folded into the GEMM epilogues (css_encoder_kernels.h,
// "LayerNorm without LayerNorm kernels").  Launches per layer: QKV GEMM, attention, O GEMM, FFN1 GEMM, FFN2 GEMM.
// The two pre tensors alternate between x16 and pre32 (both hold bf16 rows here).
int forward_folded_bf16(css_encoder* e, const int32_t* ids, const int32_t* cu, int B, int T, int max_len, int normalize,
                        float* out, hipStream_t st) {
    const css_encoder_cfg& c = e->cfg;
    const int H = c.hidden, F = c.ffn, maxL = c.max_seq_len;
    int rc;
    bf16_t* pre[2] = {(bf16_t*)e->x16, (bf16_t*)e->pre32};
    {
        ProfScope ps("enc_embed_ln", st);
        hipLaunchKernelGGL(k_embed_pre<768>, dim3((T + 15) / 16), dim3(256), 0, st, ids, cu, B, e->wemb, e->pemb, c.vocab,
                           c.max_pos, pre[0], e->stats[0], T, kStatScale1, kStatScale2);
        CSS_LAUNCH_CHECK();
    }
    const float *g_in = e->embg, *b_in = e->embb;  // the LayerNorm that turns pre[0] into the layer input
    G8Side side{};
    side.inv_h = 1.0f / H;
    side.eps = c.ln_eps;
    for (int li = 0; li < c.num_layers; ++li) {
        const LayerW& L = e->layers[li];
        // x = LN(pre[0]) -> qkv; zeroes stats[1]
        side.stats_in = e->stats[0];
        side.stats_out = e->stats[1];
        side.cgroup = enc_env().cg_qkv;
        side.grid = enc_env().grid_qkv;
        const int g4 = enc_env().gemm4w;
        if ((rc = (g4 & 1) ? launch_gemm4w<EPI_AFF_QKV>(pre[0], L.wqkv_f, L.dqkv, e->qkv, T, 3 * H, H, H, 0.125f * 1.44269504088896341f,
                                                          side, e->num_cus, st, "enc_gemm_qkv")
                           : launch_gemm8p<EPI_AFF_QKV, true>(pre[0], L.wqkv_f, L.dqkv, e->qkv, T, 3 * H, H, H, 0.125f * 1.44269504088896341f,
                                                                side, e->num_cus, st, "enc_gemm_qkv")) != CSS_OK)
            return rc;
        {
            ProfScope ps("enc_attention", st);
            const size_t lds = 4 * 8192 + (size_t)(2 * maxL - 1 + 64) * 4;
            const int nqb = (max_len + 127) / 128;
            hipLaunchKernelGGL((k_attention_bf16<64, true>), dim3(B * nqb * c.heads), dim3(256), lds, st, (const bf16_t*)e->qkv, cu,
                               e->bias_tab, maxL, H, (bf16_t*)e->ctx, nqb, c.heads, e->att_range);
            CSS_LAUNCH_CHECK();
        }
        // pre[1] = ctx Wo^T + (bo + beta) + gamma (pre[0] - mu) rs; stats[1] += row sums
        side.pprev = pre[0];
        side.cvec = g_in;
        side.cgroup = enc_env().cg_o;
        side.grid = enc_env().grid_o;
        if ((rc = (g4 & 2) ? launch_gemm4w<EPI_RES, 1>(e->ctx, L.wo_p, L.bo_f, pre[1], T, H, H, 0, 1.0f, side, e->num_cus, st, "enc_gemm_o")
                           : launch_gemm8p<EPI_RES, true, 1>(e->ctx, L.wo_p, L.bo_f, pre[1], T, H, H, 0, 1.0f, side, e->num_cus, st, "enc_gemm_o")) != CSS_OK)
            return rc;
        // x1 = LN1(pre[1]) -> ffn = gelu(x1 W1^T + b1); zeroes stats[0]
        side.stats_in = e->stats[1];
        side.stats_out = e->stats[0];
        side.cgroup = enc_env().cg_ffn1;
        side.grid = enc_env().grid_ffn1;
        if ((rc = (g4 & 4) ? launch_gemm4w<EPI_AFF_GELU>(pre[1], L.w1_f, L.d1, e->ffn, T, F, H, 0, 1.0f, side, e->num_cus, st, "enc_gemm_ffn1")
                           : launch_gemm8p<EPI_AFF_GELU, true>(pre[1], L.w1_f, L.d1, e->ffn, T, F, H, 0, 1.0f, side, e->num_cus, st, "enc_gemm_ffn1")) != CSS_OK)
            return rc;
        // pre[0] = ffn W2^T + (b2 + beta1) + gamma1 (pre[1] - mu) rs; stats[0] += row sums
        side.pprev = pre[1];
        side.cvec = L.ln1g;
        side.cgroup = enc_env().cg_ffn2;
        side.grid = enc_env().grid_ffn2;
        if ((rc = (g4 & 8) ? launch_gemm4w<EPI_RES>(e->ffn, L.w2_p, L.b2_f, pre[0], T, H, F, 0, 1.0f, side, e->num_cus, st, "enc_gemm_ffn2")
                           : launch_gemm8p<EPI_RES, true>(e->ffn, L.w2_p, L.b2_f, pre[0], T, H, F, 0, 1.0f, side, e->num_cus, st, "enc_gemm_ffn2")) != CSS_OK)
            return rc;
        g_in = L.ln2g;
        b_in = L.ln2b;
    }
    {
        ProfScope ps("enc_pool", st);
        constexpr int kPoolSlices = 8;
        // partial sums live in ffn (free after the last FFN2; capacity >= 8 B rows of 8 H bytes): [B][8][H] floats
        float* part = (float*)e->ffn;
        hipLaunchKernelGGL(k_pool_partial_ln<768>, dim3(B, kPoolSlices), dim3(256), 0, st, (const bf16_t*)pre[0], e->stats[0], g_in,
                           b_in, 1.0f / H, c.ln_eps, cu, kPoolSlices, part);
        hipLaunchKernelGGL(k_pool_final<768>, dim3(B), dim3(256), 0, st, part, cu, kPoolSlices, normalize, out);
        CSS_LAUNCH_CHECK();
    }
    e->x32_valid = false;
    return CSS_OK;
}

int forward_any(css_encoder* e, const int32_t* ids, const int32_t* cu, int B, int T, int max_len, int normalize,
                float* out, hipStream_t st) {
    if (!e->weights_ready) {
        css::set_error("css_encoder_forward: weights not loaded (call css_encoder_load_weights or css_encoder_init_synthetic)");
        return CSS_ERR_STATE;
    }
    CSS_REQUIRE(B >= 1 && T >= B, "css_encoder_forward: bad batch (B=%d, tokens=%d)", B, T);
    CSS_REQUIRE(max_len >= 1 && max_len <= e->cfg.max_seq_len, "css_encoder_forward: max_len=%d outside [1, %d]", max_len,
                e->cfg.max_seq_len);
    const EncEnv& env = enc_env();
    if (e->cfg.compute == 0 && env.fuse_ln && env.big_tiles && env.loop8 && T >= 1024 && e->cfg.ffn % 256 == 0 &&
        (size_t)T * e->cfg.ffn * 2 < ((size_t)1 << 32))
        return forward_folded_bf16(e, ids, cu, B, T, max_len, normalize, out, st);
    e->x32_valid = true;
    return e->cfg.compute == 0 ? forward_typed<bf16_t>(e, ids, cu, B, T, max_len, normalize, out, st)
                               : forward_typed<float>(e, ids, cu, B, T, max_len, normalize, out, st);
}

}  // namespace

extern "C" {

int css_encoder_create(const css_encoder_cfg* cfg, int device, css_encoder** out) {
    CSS_REQUIRE(cfg && out, "css_encoder_create: NULL argument");
    CSS_REQUIRE(cfg->hidden == 768 && cfg->heads * 64 == cfg->hidden,
                "css_encoder_create: kernels are built for hidden=768, head_dim=64 (got hidden=%d heads=%d)", cfg->hidden,
                cfg->heads);
    CSS_REQUIRE(cfg->ffn % 128 == 0 && cfg->ffn >= 128, "css_encoder_create: ffn must be a multiple of 128");
    CSS_REQUIRE(cfg->num_layers >= 1 && cfg->num_layers <= 64, "css_encoder_create: num_layers out of range");
    CSS_REQUIRE(cfg->max_seq_len >= 1 && cfg->max_seq_len <= 512, "css_encoder_create: max_seq_len outside [1, 512]");
    CSS_REQUIRE(cfg->max_pos >= cfg->max_seq_len + 2, "css_encoder_create: max_pos must be >= max_seq_len + 2");
    CSS_REQUIRE(cfg->vocab >= 4 && cfg->rel_buckets >= 4 && cfg->rel_buckets % 4 == 0, "css_encoder_create: bad vocab / rel_buckets");
    CSS_REQUIRE(cfg->compute == 0 || cfg->compute == 1, "css_encoder_create: compute must be 0 (bf16) or 1 (fp32)");
    int rc = css::check_device(device);
    if (rc != CSS_OK) return rc;
    DeviceGuard g(device);
    (void)enc_env();
    css_encoder* e = new css_encoder();
    e->cfg = *cfg;
    e->device = device;
    if (enc_env().att_range >= 0.f) e->att_range = enc_env().att_range;
    {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, device) == hipSuccess) e->num_cus = p.multiProcessorCount;
    }
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err != hipSuccess) {
        delete e;
        return css::hip_fail(err, "hipStreamCreate", __FILE__, __LINE__);
    }
    rc = build_params(e);
    if (rc != CSS_OK) {
        css_encoder_free(e);
        return rc;
    }
    *out = e;
    return CSS_OK;
}

int css_encoder_free(css_encoder* e) {
    if (!e) return CSS_OK;
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& kv : e->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    for (auto& kv : e->params) {
        // q/k/v views alias the fused tensors: free only owning entries
        const std::string& n = kv.first;
        const bool view = n.find(".attn.q.") != std::string::npos || n.find(".attn.k.") != std::string::npos ||
                          n.find(".attn.v.") != std::string::npos;
        if (!view && kv.second.p) (void)hipFree(kv.second.p);
    }
    for (auto& L : e->layers) {
        if (L.wqkv_h) (void)hipFree(L.wqkv_h);
        if (L.wo_h) (void)hipFree(L.wo_h);
        if (L.w1_h) (void)hipFree(L.w1_h);
        if (L.w2_h) (void)hipFree(L.w2_h);
        void* fp[] = {L.wqkv_f, L.w1_f, L.w2_p, L.wo_p, L.dqkv, L.d1, L.bo_f, L.b2_f};
        for (void* p : fp)
            if (p) (void)hipFree(p);
    }
    void* ptrs[] = {e->bias_tab, e->bucket_dev, e->x32, e->pre32, e->x16, e->qkv, e->ctx, e->ffn, e->ids_dev,
                    e->cu_dev, e->out_dev, e->stats[0], e->stats[1]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return CSS_OK;
}

int css_encoder_load_weights(css_encoder* e, const css_tensor* tensors, int n) {
    CSS_REQUIRE(e && tensors && n >= 0, "css_encoder_load_weights: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    // A checkpoint must cover every parameter exactly once: a missing tensor would leave uninitialised device
    // memory in the model, a duplicate (the same key under two prefixes) makes the result depend on the order.
    std::map<std::string, int> seen;
    for (int i = 0; i < n; ++i) {
        CSS_REQUIRE(tensors[i].name && tensors[i].data, "css_encoder_load_weights: tensor %d has NULL name/data", i);
        std::string name = tensors[i].name;
        const std::string pfx = "0.auto_model.";  // sentence-transformers module prefix
        if (name.compare(0, pfx.size(), pfx) == 0) name = name.substr(pfx.size());
        if (name.compare(0, 6, "mpnet.") == 0) name = name.substr(6);
        if (name.compare(0, 7, "pooler.") == 0 || name == "embeddings.position_ids") continue;  // unused
        auto it = e->params.find(name);
        CSS_REQUIRE(it != e->params.end(), "css_encoder_load_weights: unknown parameter '%s'", name.c_str());
        CSS_REQUIRE(it->second.numel == tensors[i].numel, "css_encoder_load_weights: '%s' has %lld elements, expected %lld",
                    name.c_str(), (long long)tensors[i].numel, (long long)it->second.numel);
        CSS_REQUIRE(seen[name]++ == 0, "css_encoder_load_weights: parameter '%s' appears more than once", name.c_str());
        CSS_HIP_TRY(hipMemcpy(it->second.p, tensors[i].data, (size_t)tensors[i].numel * 4, hipMemcpyHostToDevice));
    }
    for (const auto& kv : e->params) {
        const std::string& nm = kv.first;
        const bool view = nm.find(".attn.q.") != std::string::npos || nm.find(".attn.k.") != std::string::npos ||
                          nm.find(".attn.v.") != std::string::npos;
        const size_t fq = nm.find(".attn.qkv.");
        if (fq != std::string::npos) {
            // fused [3H, H] weight / [3H] bias: given as one tensor, or as its three HF views q, k, v -- not both
            int parts = 0;
            for (const char* v : {"q", "k", "v"}) {
                std::string vn = nm;
                vn.replace(fq, 10, std::string(".attn.") + v + ".");
                parts += seen.count(vn) ? 1 : 0;
            }
            const bool whole = seen.count(nm) != 0;
            CSS_REQUIRE(!(whole && parts > 0), "css_encoder_load_weights: '%s' given both fused and as q/k/v", nm.c_str());
            CSS_REQUIRE(whole || parts == 3, "css_encoder_load_weights: checkpoint does not cover '%s' (q/k/v: %d of 3)",
                        nm.c_str(), parts);
        } else if (!view) {
            CSS_REQUIRE(seen.count(nm) != 0, "css_encoder_load_weights: checkpoint has no tensor for '%s'", nm.c_str());
        }
    }
    return finalize_weights(e);
}

int css_encoder_init_synthetic(css_encoder* e, uint64_t seed) {
    CSS_REQUIRE(e, "css_encoder_init_synthetic: NULL encoder");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    for (auto& kv : e->params) {
        const std::string& n = kv.first;
        const bool view = n.find(".attn.q.") != std::string::npos || n.find(".attn.k.") != std::string::npos ||
                          n.find(".attn.v.") != std::string::npos;
        if (view) continue;
        const Param& p = kv.second;
        hipLaunchKernelGGL(k_synth_fill, dim3(1024), dim3(256), 0, e->stream, p.p, (size_t)p.numel,
                           css_synth_tensor_seed(seed, p.synth_id), p.mean, p.std);
    }
    // padding_idx rows are zero at init (word_embeddings[pad], position_embeddings[pad])
    const int H = e->cfg.hidden;
    hipLaunchKernelGGL(k_zero_row, dim3((H + 255) / 256), dim3(256), 0, e->stream, e->wemb + (size_t)e->cfg.pad_id * H, H);
    hipLaunchKernelGGL(k_zero_row, dim3((H + 255) / 256), dim3(256), 0, e->stream, e->pemb + (size_t)e->cfg.pad_id * H, H);
    CSS_LAUNCH_CHECK();
    return finalize_weights(e);
}

int css_encoder_export_weight(const css_encoder* ce, const char* name, float* out_host, int64_t numel) {
    css_encoder* e = const_cast<css_encoder*>(ce);
    CSS_REQUIRE(e && name && out_host, "css_encoder_export_weight: NULL argument");
    auto it = e->params.find(name);
    CSS_REQUIRE(it != e->params.end(), "css_encoder_export_weight: unknown parameter '%s'", name);
    CSS_REQUIRE(it->second.numel == numel, "css_encoder_export_weight: '%s' has %lld elements, not %lld", name,
                (long long)it->second.numel, (long long)numel);
    DeviceGuard g(e->device);
    CSS_HIP_TRY(hipMemcpy(out_host, it->second.p, (size_t)numel * 4, hipMemcpyDeviceToHost));
    return CSS_OK;
}

int css_encoder_set_attention_range(css_encoder* e, float range) {
    CSS_REQUIRE(e, "css_encoder_set_attention_range: NULL encoder");
    CSS_REQUIRE(range == 0.f || (range >= 1.f && range <= 1.2676506e30f), "css_encoder_set_attention_range: range must be 0 or in [1, 2^100]");
    std::lock_guard<std::mutex> lk(e->mu);
    e->att_range = range;
    // captured tiny-batch graphs hold the old kernel argument
    for (auto& kv : e->graphs)
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    e->graphs.clear();
    return CSS_OK;
}

int css_encoder_debug_read(css_encoder* e, const char* what, float* out_host, int64_t numel) {
    CSS_REQUIRE(e && what && out_host && numel > 0, "css_encoder_debug_read: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    const std::string w = what;
    const bool bf = e->cfg.compute == 0;
    const void* src = nullptr;
    bool typed = true;  // operand-typed (bf16 in product mode) vs always fp32
    if (w == "x32") {
        CSS_REQUIRE(e->x32_valid, "css_encoder_debug_read: the last forward ran the LayerNorm-folded path, which does not "
                                  "materialise x32 (set CSS_ENC_FUSE_LN=0)");
        src = e->x32;
        typed = false;
    }
    else if (w == "pre32") { src = e->pre32; typed = false; }
    else if (w == "qkv") src = e->qkv;
    else if (w == "ctx") src = e->ctx;
    else if (w == "ffn") src = e->ffn;
    CSS_REQUIRE(src != nullptr, "css_encoder_debug_read: unknown or unallocated buffer '%s'", what);
    CSS_HIP_TRY(hipDeviceSynchronize());
    if (typed && bf) {
        std::vector<uint16_t> tmp((size_t)numel);
        CSS_HIP_TRY(hipMemcpy(tmp.data(), src, (size_t)numel * 2, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < numel; ++i) {
            const uint32_t u = (uint32_t)tmp[i] << 16;
            memcpy(&out_host[i], &u, 4);
        }
    } else {
        CSS_HIP_TRY(hipMemcpy(out_host, src, (size_t)numel * 4, hipMemcpyDeviceToHost));
    }
    return CSS_OK;
}

int css_encoder_forward(css_encoder* e, const int32_t* ids, const int32_t* cu, int B, int normalize, float* out) {
    CSS_REQUIRE(e && ids && cu && out, "css_encoder_forward: NULL argument");
    CSS_REQUIRE(B >= 1, "css_encoder_forward: B < 1");
    CSS_REQUIRE(cu[0] == 0, "css_encoder_forward: cu_seqlens[0] must be 0");
    int max_len = 0;
    for (int b = 0; b < B; ++b) {
        const int len = cu[b + 1] - cu[b];
        CSS_REQUIRE(len >= 1 && len <= e->cfg.max_seq_len, "css_encoder_forward: sequence %d has length %d outside [1, %d]", b,
                    len, e->cfg.max_seq_len);
        max_len = len > max_len ? len : max_len;
    }
    const int T = cu[B];
    for (int t = 0; t < T; ++t)
        CSS_REQUIRE(ids[t] >= 0 && ids[t] < e->cfg.vocab && ids[t] != e->cfg.pad_id,
                    "css_encoder_forward: token %d has id %d (outside the vocabulary or the pad id)", t, ids[t]);
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int rc = ensure_acts(e, T, B);
    if (rc != CSS_OK) return rc;
    CSS_HIP_TRY(hipMemcpyAsync(e->ids_dev, ids, (size_t)T * 4, hipMemcpyHostToDevice, e->stream));
    CSS_HIP_TRY(hipMemcpyAsync(e->cu_dev, cu, (size_t)(B + 1) * 4, hipMemcpyHostToDevice, e->stream));
    bool done = false;
    if (e->graphs_ok && T <= 64 && B <= 8 && !css::prof_enabled()) {
        // ~90 launches of a few microseconds each: replay them as one hipGraph
        const uint64_t key = ((uint64_t)B << 40) | ((uint64_t)T << 20) | ((uint64_t)max_len << 1) | (normalize ? 1u : 0u);
        auto& ge = e->graphs[key];
        ++ge.uses;
        if (ge.exec == nullptr && ge.uses == 2) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int frc = forward_any(e, e->ids_dev, e->cu_dev, B, T, max_len, normalize, e->out_dev, e->stream);
                const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
                if (frc == CSS_OK && ce == hipSuccess && graph &&
                    hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    // instantiated
                } else {
                    ge.exec = nullptr;
                    e->graphs_ok = false;  // fall back to eager launches for good
                    (void)hipGetLastError();
                }
                if (graph) (void)hipGraphDestroy(graph);
            } else {
                e->graphs_ok = false;
                (void)hipGetLastError();
            }
        }
        if (ge.exec) {
            CSS_HIP_TRY(hipGraphLaunch(ge.exec, e->stream));
            done = true;
        }
    }
    if (!done && (rc = forward_any(e, e->ids_dev, e->cu_dev, B, T, max_len, normalize, e->out_dev, e->stream)) != CSS_OK)
        return rc;
    CSS_HIP_TRY(hipMemcpyAsync(out, e->out_dev, (size_t)B * e->cfg.hidden * 4, hipMemcpyDeviceToHost, e->stream));
    CSS_HIP_TRY(hipStreamSynchronize(e->stream));
    return CSS_OK;
}

int css_encoder_forward_dev(css_encoder* e, const int32_t* ids_dev, const int32_t* cu_dev, int B, int total_tokens,
                            int max_len, int normalize, float* out_dev, void* stream) {
    CSS_REQUIRE(e && ids_dev && cu_dev && out_dev, "css_encoder_forward_dev: NULL argument");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int rc = ensure_acts(e, total_tokens, B);
    if (rc != CSS_OK) return rc;
    return forward_any(e, ids_dev, cu_dev, B, total_tokens, max_len, normalize, out_dev, (hipStream_t)stream);
}

}  // extern "C"
