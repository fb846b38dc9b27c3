// css_index.hip -- exact flat index (IndexFlatIP / IndexFlatL2 semantics) for gfx950.
//
// Replaces, behind include/css_hip.h, the faiss calls of the reference's
// HybridStorage: index creation (src/storage.py:252-258), add with the fused
// row normalisation (src/storage.py:343-359) and the brute-force search
// (src/storage.py:424-436).
//
// HBM layout: xb[cap][dpad] fp32 row-major, dpad = dim rounded up to 64 floats
// (768 -> 768, i.e. 3072 B rows), zero padded, xnorm2[cap] (squared norms) and,
// for inner-product indexes while it fits, a bf16 shadow copy xh[cap][dpad] of
// the rows (coarse scans read it; see css_knn_coarse.h).  Capacity grows
// geometrically or is reserved up front (css_index_reserve) so a 10M..80M row
// shard is allocated once.
//
// Kernels (DESIGN.md has the rooflines):
//   k_ingest_rows      one wave per row: optional synthetic generation, fused
//                      x/(||x||+1e-8), zero pad, squared norm, bf16 shadow.  HBM bound
//   k_scan_coarse,     default search path: coarse bf16 scores (MFMA scan for
//   k_sweep_coarse,    batches, HBM-bound sweep for 1..4 queries) inside a rigorous
//   k_coarse_select    error band, then exact fp32 rescoring of the band (css_knn_coarse.h)
//   k_scan_small       exact fp32 sweep for 1..16 queries (L2 metric, no shadow rows,
//                      flagged queries): 16 lanes per row, VALU FMAs, DPP row reduction,
//                      block-shared sorted top-k lists in LDS, grid-wide threshold.   HBM bound
//   k_scan_mfma_split  exact batched scan on split bf16 operands (3 MFMAs per product);
//   k_scan_mfma        the same on fp32-input MFMA (CSS_KNN_BATCH=fp32, verification)
//   k_merge_final      per query: merge the per-block lists into the final top-k.
#include "css_common.h"
#include "css_knn_kernels.h"
#include "../../include/css_synth.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <string>
#include <cfloat>
#include <cmath>
#include <mutex>
#include <shared_mutex>
#include <type_traits>
#include <vector>

using namespace css;

struct css_index {
    int dim = 0, dpad = 0, metric = 0, device = 0;
    int64_t ntotal = 0, cap = 0, id_base = 0;
    // what css_index_ntotal reports: searches of shadow-less indexes temporarily narrow xb / xnorm2 / ntotal / id_base
    // to a row range (RowView, under ws_mu), and a concurrent reader must not see that
    std::atomic<int64_t> ntotal_pub{0};
    float* xb = nullptr;
    float* xnorm2 = nullptr;
    unsigned short* xh = nullptr;  // bf16 shadow rows [cap][dpad] for the coarse scan (nullptr: not kept)
    int shadow = -1;               // -1 undecided, 0 off, 1 on (CSS_KNN_SHADOW, HBM headroom)
    int shadow_policy = -1;        // css_index_set_shadow: -1 automatic, 0 never, 1 always
    int search_mode = CSS_SEARCH_AUTO;
    const uint32_t* cur_mask = nullptr;  // allow-bitmap of the search in progress (set under ws_mu)
    uint32_t* mask_ws = nullptr;   size_t mask_ws_cap = 0;   // device copy of a host bitmap
    uint32_t* excl_ws = nullptr;   size_t excl_ws_cap = 0;   // k > 128: allow-bitmap minus the rows earlier passes returned
    int* maxn2 = nullptr;          // device, 3 words: bits of max ||row||^2, max ||row - bf16(row)||^2, max ||row - int8(row)||^2 (cz_eps)
    // int8 shadow rows (kept next to the bf16 ones when there is room): signed byte = round(x / s), s = max|x| / 127 per
    // row; read by the 1..4-query sweep and by the int8 MFMA scan of batches
    unsigned char* x8 = nullptr;
    float* x8s = nullptr;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // reusable workspaces (grown on demand, guarded by ws_mu)
    float* q_raw = nullptr;   size_t q_raw_cap = 0;     // floats
    float* qpad = nullptr;    size_t qpad_cap = 0;      // floats
    float* qnorm2 = nullptr;  size_t qnorm2_cap = 0;    // floats
    float* qerr2 = nullptr;   size_t qerr2_cap = 0;     // per query: ||q - bf16(q)||^2 (cz_eps)
    float* qerr2_i8 = nullptr; size_t qerr2_i8_cap = 0; // per query: ||q - int8(q)||^2 (int8 MFMA scan)
    float* qscale = nullptr;  size_t qscale_cap = 0;    // per query: scale of its int8 row
    // int8 policy feedback: the flagged count of the last search that read the int8 rows travels to pinned host memory
    // behind the search (no synchronisation); the next search of that kind looks at it when it has landed
    struct I8Feedback {
        int* h_nflag = nullptr;
        hipEvent_t ev = nullptr;
        bool pending = false;
        int nq = 0;        // queries of the search the pending count belongs to
        int backoff = 0;   // searches left on the bf16 rows after an int8 search that flagged too many queries
    };
    I8Feedback fb_batch, fb_sweep;
    unsigned short* qsplit = nullptr; size_t qsplit_cap = 0;  // bf16 (h,l) pairs
    int* gthr = nullptr;      size_t gthr_cap = 0;      // ints
    float* part_s = nullptr;  uint32_t* part_i = nullptr; size_t part_cap = 0;  // entries
    int64_t* out_i = nullptr; size_t out_cap = 0;      // entries (12 bytes each); a call's rows: [nq * k ids | nq * k scores]
    // pinned staging of the host API for the reference's call shape (one query, k' = 100: 3 KB in, 1.2 KB out): pageable
    // copies of that size cost a staging pass and a wait each
    char* h_stage = nullptr;
    static constexpr size_t kHostStage = 64 * 1024;   // bytes, each way
    float* stage = nullptr;   size_t stage_cap = 0;     // floats
    // coarse + rescore path (css_knn_coarse.h)
    unsigned short* qh = nullptr; size_t qh_cap = 0;    // bf16 queries
    float* cthr = nullptr;    size_t cthr_cap = 0;
    int* cand_n = nullptr;    size_t cand_n_cap = 0;
    int* cflags = nullptr;    size_t cflags_cap = 0;    // [nq_pad] flags | [nq_pad] flagged list | [1] count | [1] fix-up blocks done
    float* cand_s = nullptr;  size_t cand_s_cap = 0;
    uint32_t* cand_i = nullptr; size_t cand_i_cap = 0;
    int* cpace = nullptr;     size_t cpace_cap = 0;     // sibling pacing counters [stage][group]
    int* fs_state = nullptr;  size_t fs_state_cap = 0;  // k_sweep_cascade: ticket / stage counters / threshold key words
    // device-side exact fix-up of flagged queries (k_scan_small<FIX>): one global list + lock per query
    float* fix_s = nullptr;   uint32_t* fix_i = nullptr; size_t fix_cap = 0;    // entries [nq_pad][k]
    int* fix_lock = nullptr;  size_t fix_lock_cap = 0;
    // second coarse pass over flagged queries (launch_scan_coarse): up to kF2Max slots with CZ_CAP2 candidates each
    unsigned short* qh2 = nullptr; size_t qh2_cap = 0;   // bf16 rows of the flagged queries
    float* thr2 = nullptr;    size_t thr2_cap = 0;
    int* rs_work = nullptr;   size_t rs_work_cap = 0;   // [count | (query, part) items] of the band rescoring
    int* cand_n2 = nullptr;   size_t cand_n2_cap = 0;
    float* cand_s2 = nullptr; size_t cand_s2_cap = 0;
    uint32_t* cand_i2 = nullptr; size_t cand_i2_cap = 0;
    int* flagB = nullptr;     size_t flagB_cap = 0;      // [nq_pad] list | [1] count | [1] fix-up blocks done: queries left to the exact sweep
    // shadow-less indexes: bf16 rows of one row range at a time + the per-range top-k lists (search_noshadow_ranges)
    unsigned short* xh_tmp = nullptr; size_t xh_tmp_cap = 0;
    float* x8s_tmp = nullptr; size_t x8s_tmp_cap = 0;   // row scales when the scratch rows are int8
    int64_t range_rows = 0;   // css_index_set_range_rows: rows per range (0: from the free HBM, at most 2^24)
    float* rng_d = nullptr;   int64_t* rng_i = nullptr;  size_t rng_cap = 0;
    const int* last_nswept = nullptr;                    // device counter behind css_index_last_swept
    // rows written by css_index_add_dev / _add_synthetic on the CALLER's stream: searches, reallocation and
    // export wait for this event before touching rows, norms or maxn2
    hipEvent_t ingest_ev = nullptr;
    bool ingest_pending = false;
    // every search shares ONE set of workspaces (qpad, gthr, cthr, cand_*, cflags, fix_*, part_*): ws_mu serialises the
    // host-side enqueue only, so the last search's stream is remembered and a search on ANOTHER stream first waits
    // for this event (recorded at the end of each search) before it overwrites them
    hipEvent_t ws_ev = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_pending = false;
    const int* last_nflag = nullptr;   // device counter of the last candidate-path search (css_index_last_flagged)
    std::shared_mutex mu;  // search: shared; add/reset/reserve: exclusive
    std::mutex ws_mu;      // workspaces + own stream are single-user
};

namespace {

constexpr int kWaves = 4;  // waves per block in the scan kernels

template <typename T>
int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap) return CSS_OK;
    size_t ncap = std::max(need, *cap * 2);
    if (*p) CSS_HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    hipError_t e = hipMalloc((void**)p, ncap * sizeof(T));
    if (e != hipSuccess) return css::hip_fail(e, "hipMalloc(workspace)", __FILE__, __LINE__);
    *cap = ncap;
    return CSS_OK;
}

// ------------------------------------------------------------------ ingest
// One wave per row.  SYNTH: value = css_synth_normal(seed, (first_row+row)*dim + c).
template <bool SYNTH>
__device__ __forceinline__ void ingest_row(int64_t row, int lane, const float* __restrict__ src, float* __restrict__ dst,
                                           float* __restrict__ norm2, int dim, int dpad, int normalize, uint64_t seed,
                                           int64_t first_row, unsigned short* __restrict__ dsth, int* __restrict__ maxn2,
                                           float* __restrict__ err2_out, unsigned char* __restrict__ dst8,
                                           float* __restrict__ dst8s) {
    const float* s = SYNTH ? nullptr : src + row * (int64_t)dim;
    const uint64_t base = (uint64_t)((first_row + row) * (int64_t)dim);
    float ss = 0.f, amax = 0.f;
    for (int c = lane; c < dim; c += 64) {
        float v = SYNTH ? css_synth_normal(seed, base + (uint64_t)c) : s[c];
        ss = fmaf(v, v, ss);
        amax = fmaxf(amax, fabsf(v));
    }
    ss = wave_allsum(ss);
    amax = wave_allmax(amax);
    // reference: x / (||x||_2 + 1e-8)  (src/storage.py:349-350, :426)
    const float nrm = sqrtf(ss) + 1e-8f;
    float* d = dst + row * (int64_t)dpad;
    // int8 shadow row: byte = rint(v / s8), s8 = max|v| / 127 (of the values as stored, i.e. after normalisation)
    if (normalize) amax = amax / nrm;
    const float s8 = amax > 0.f ? amax / 127.f : 1.f, inv8 = amax > 0.f ? 127.f / amax : 0.f;
    float e8 = 0.f;
    float s2 = 0.f, e2 = 0.f;
    for (int c = lane; c < dpad; c += 64) {
        float v = 0.f;
        if (c < dim) {
            v = SYNTH ? css_synth_normal(seed, base + (uint64_t)c) : s[c];
            if (normalize) v = v / nrm;
        }
        d[c] = v;
        const __bf16 h = (__bf16)v;   // the rounding every bf16 copy of this row uses (shadow rows, k_rows_to_bf16*)
        if (dsth) dsth[row * (int64_t)dpad + c] = __builtin_bit_cast(unsigned short, h);
        s2 = fmaf(v, v, s2);
        const float dv = v - (float)h;   // exact in fp32
        e2 = fmaf(dv, dv, e2);
        {   // (measured whether or not the int8 row is kept: shadow-less indexes quantise the same way per search)
            const float k8 = fminf(fmaxf(rintf(v * inv8), -127.f), 127.f);
            if (dst8) dst8[row * (int64_t)dpad + c] = (unsigned char)((int)k8 & 0xFF);   // signed int8 (what the int8 MFMA takes)
            const float d8 = fmaf(-s8, k8, v);   // v - s8 * k8 with one rounding
            e8 = fmaf(d8, d8, e8);
        }
    }
    s2 = wave_allsum(s2);
    e2 = wave_allsum(e2);
    e8 = wave_allsum(e8);
    if (lane == 0) {
        if (dst8) dst8s[row] = s8;
        if (maxn2 && e8 > __int_as_float(__hip_atomic_load(maxn2 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            atomicMax(maxn2 + 2, __float_as_int(e8));
    }
    if (lane == 0 && norm2) norm2[row] = s2;
    // ||row - bf16(row)||^2: what the rounding actually cost (cz_eps: the measured error band of the candidate scans)
    if (lane == 0 && err2_out) err2_out[row] = e2;
    if (lane == 0 && maxn2 && e2 > __int_as_float(__hip_atomic_load(maxn2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(maxn2 + 1, __float_as_int(e2));
    // running max of ||row||^2 (non-negative floats order like their bit patterns); rows are ~unit
    // norm in the product, so after the first few rows almost no atomic is issued
    if (lane == 0 && maxn2 && s2 > __int_as_float(__hip_atomic_load(maxn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(maxn2, __float_as_int(s2));
}
template <bool SYNTH>
__global__ __launch_bounds__(256) void k_ingest_rows(const float* __restrict__ src, float* __restrict__ dst,
                                                     float* __restrict__ norm2, int64_t n, int dim, int dpad,
                                                     int normalize, uint64_t seed, int64_t first_row,
                                                     unsigned short* __restrict__ dsth, int* __restrict__ maxn2,
                                                     float* __restrict__ err2_out, unsigned char* __restrict__ dst8,
                                                     float* __restrict__ dst8s) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    ingest_row<SYNTH>(row, threadIdx.x & 63, src, dst, norm2, dim, dpad, normalize, seed, first_row, dsth, maxn2, err2_out, dst8,
                      dst8s);
}

// ------------------------------------------------------------------ scan (small nq)
// Block = 4 waves.  A wave instruction covers 4 rows: lane = 16*r + sub reads the
// float4 at column 64*t + 4*sub of row r, so 16 lanes fetch 256 contiguous bytes.
// Scores are "larger is better": IP -> dot, L2 -> -(sum (x-q)^2).
// LDS: qs[NQ][dpad] | ls[NQ][k] | li[NQ][k] | lock[NQ] | qid[NQ] | (FIX) per-wave merge scratch
//
// FIX = true is the device-side exact fix-up of the candidate path (css_knn_coarse.h): the launch follows
// every cascade unconditionally, reads the number of flagged queries from device memory and returns at once
// when it is zero (the usual case) -- no host round trip.  Otherwise the grid walks the flagged queries NQ at
// a time; a block's lists are merged into one global list per query under an agent-scope lock (release /
// acquire fences around plain loads and stores, MI355X_MICROARCH.md "Valid forms"); the lists were reset by
// the kernel that flagged the query; the block that finishes last turns them into D / I rows.
// (rows are read once per sweep: non-temporal, like the shadow-row sweeps -- css_knn_coarse.h, cz_row_load; CSS_SCAN_NT=0 builds for A/B runs)
#ifndef CSS_SCAN_NT
#define CSS_SCAN_NT 1
#endif
__device__ __forceinline__ float4 scan_row_load(const float4* p) {
#if CSS_SCAN_NT
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
template <int NQ, int TT, int METRIC, bool FIX = false>
__global__ __launch_bounds__(256, 4) void k_scan_small(const float4* __restrict__ xb, const float* __restrict__ qpad,
                                                    int64_t ntotal, int T_rt, int k, int64_t groups_per_block,
                                                    int* __restrict__ gthr, float* __restrict__ part_s,
                                                    uint32_t* __restrict__ part_i, int nq_real_arg,
                                                    const uint32_t* __restrict__ mask,
                                                    const int* __restrict__ flag_list, const int* __restrict__ nflag_p,
                                                    float* fix_s, uint32_t* fix_i, int* fix_lock, int* fix_done,
                                                    int64_t id_base, float* D, int64_t* I) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = TT > 0 ? TT : T_rt;  // float4 steps of 16 lanes: dpad = 64*T
    const int dpad = T * 64;
    float* qs = reinterpret_cast<float*>(smem);
    float* ls = qs + NQ * dpad;
    uint32_t* li = reinterpret_cast<uint32_t*>(ls + NQ * k);
    int* lock = reinterpret_cast<int*>(li + NQ * k);
    int* qid = lock + NQ;                                   // query served by list j
    float* ms = reinterpret_cast<float*>(qid + NQ);         // FIX: [4 waves][k] merge scratch
    uint32_t* mi = reinterpret_cast<uint32_t*>(ms + 4 * k);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, rsub = lane >> 4;
    const int nfl = FIX ? __hip_atomic_load(nflag_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
    if (FIX && nfl == 0) return;

  for (int c0 = 0; c0 < (FIX ? nfl : 1); c0 += NQ) {
    const int nq_real = FIX ? min(NQ, nfl - c0) : nq_real_arg;
    if (tid < NQ) {
        lock[tid] = 0;
        qid[tid] = FIX ? (tid < nq_real ? flag_list[c0 + tid] : 0) : tid;
    }
    __syncthreads();
    for (int i = tid; i < NQ * dpad; i += 256) {
        const int j = i / dpad;
        qs[i] = j < nq_real ? qpad[(size_t)qid[j] * dpad + (i - j * dpad)] : 0.f;
    }
    for (int i = tid; i < NQ * k; i += 256) {
        ls[i] = -INFINITY;
        li[i] = kInvalidRow;
    }
    __syncthreads();

    const float4* qs4 = reinterpret_cast<const float4*>(qs);
    const int64_t ngroups = (ntotal + 3) >> 2;
    const int64_t g_begin = (int64_t)blockIdx.x * groups_per_block;
    const int64_t g_end = min(g_begin + groups_per_block, ngroups);

    float gcache[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) gcache[j] = -INFINITY;
    int iter = 0;

    for (int64_t g = g_begin + wave; g < g_end; g += kWaves, ++iter) {
        // refresh the grid-wide thresholds now and then; kept at the top of the
        // iteration so the FMA block below and its consumers stay in one basic block (stale values only prune less)
        if ((iter & 15) == 0) {
#pragma unroll
            for (int j = 0; j < NQ; ++j)
                gcache[j] = key2f(__hip_atomic_load(&gthr[qid[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }

        const int64_t row = g * 4 + rsub;
        const bool in_range = row < ntotal;
        const int64_t rowc = in_range ? row : ntotal - 1;
        // masked search: rows whose bit is clear can never be candidates (filter / tombstone push-down)
        const bool valid = in_range && (mask == nullptr || ((mask[rowc >> 5] >> (rowc & 31)) & 1u));
        const float4* xr = xb + rowc * (int64_t)(T * 16) + sub;

        float acc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[j] = 0.f;
        // NQ > 1: keep the query fragments in LDS (re-read per row group) instead of
        // letting LICM pin 12*NQ float4 in VGPRs, which would cost all the occupancy.
        if constexpr (NQ > 1) asm volatile("" ::: "memory");

        if constexpr (TT > 0) {
            float4 xv[TT > 0 ? TT : 1];
#pragma unroll
            for (int t = 0; t < TT; ++t) xv[t] = scan_row_load(xr + t * 16);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const float4 q = qs4[j * (TT * 16) + t * 16 + sub];
                    if constexpr (METRIC == CSS_METRIC_IP) {
                        acc[j] = fmaf(xv[t].x, q.x, acc[j]);
                        acc[j] = fmaf(xv[t].y, q.y, acc[j]);
                        acc[j] = fmaf(xv[t].z, q.z, acc[j]);
                        acc[j] = fmaf(xv[t].w, q.w, acc[j]);
                    } else {
                        float dx = xv[t].x - q.x, dy = xv[t].y - q.y, dz = xv[t].z - q.z, dw = xv[t].w - q.w;
                        acc[j] = fmaf(dx, dx, acc[j]);
                        acc[j] = fmaf(dy, dy, acc[j]);
                        acc[j] = fmaf(dz, dz, acc[j]);
                        acc[j] = fmaf(dw, dw, acc[j]);
                    }
                }
                // fence the scheduler per column step: otherwise all 12*NQ LDS reads are
                // clustered up front and the kernel spills
                if constexpr (NQ > 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int t = 0; t < T; ++t) {
                const float4 x = scan_row_load(xr + t * 16);
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const float4 q = qs4[j * (T * 16) + t * 16 + sub];
                    if constexpr (METRIC == CSS_METRIC_IP) {
                        acc[j] = fmaf(x.x, q.x, acc[j]);
                        acc[j] = fmaf(x.y, q.y, acc[j]);
                        acc[j] = fmaf(x.z, q.z, acc[j]);
                        acc[j] = fmaf(x.w, q.w, acc[j]);
                    } else {
                        float dx = x.x - q.x, dy = x.y - q.y, dz = x.z - q.z, dw = x.w - q.w;
                        acc[j] = fmaf(dx, dx, acc[j]);
                        acc[j] = fmaf(dy, dy, acc[j]);
                        acc[j] = fmaf(dz, dz, acc[j]);
                        acc[j] = fmaf(dw, dw, acc[j]);
                    }
                }
            }
        }

        // All scores and pass flags are formed in this basic block (one combined
        // ballot), so the FMA chains above cannot be sunk behind the rare slow path.
        float sc[NQ];
        bool anyp = false;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float s = row16_allsum(acc[j]);
            if constexpr (METRIC == CSS_METRIC_L2) s = -s;
            sc[j] = s;
            const float lthr = ls[j * k + (k - 1)];
            // non-strict: an equal score with a lower row id must still reach the comparator
            anyp |= (s >= lthr) & (s >= gcache[j]) & (j < nq_real);  // '&': no short-circuit branches
        }
        if (__ballot(anyp && valid && sub == 0) == 0ull) continue;

        for (int j = 0; j < nq_real; ++j) {
            float s = sc[0];
#pragma unroll
            for (int u = 1; u < NQ; ++u) s = j == u ? sc[u] : s;
            const float lthr = ls[j * k + (k - 1)];
            const float gj = key2f(__hip_atomic_load(&gthr[qid[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const bool pass = valid && sub == 0 && s >= lthr && s >= gj;
            unsigned long long m = __ballot(pass);
            if (m == 0ull) continue;
            // slow path: serialise on the block-shared list of query j
            if (lane == 0) {
                while (atomicCAS(&lock[j], 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            bool changed = false;
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float cs = __shfl(s, l);
                const uint32_t cid = (uint32_t)(g * 4 + (l >> 4));
                changed |= wave_insert<uint32_t>(ls + j * k, li + j * k, k, cs, cid, lane);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const float kth = ls[j * k + (k - 1)];
            if (lane == 0) {
                atomicExch(&lock[j], 0);
                if (changed && kth > gj) atomicMax(&gthr[qid[j]], f2key(kth));
            }
        }
    }
    __syncthreads();
    if constexpr (FIX) {
        // merge this block's lists into the global list of each query (rare path: clarity over speed)
        float* ws = ms + wave * k;
        uint32_t* wi = mi + wave * k;
        for (int j = wave; j < nq_real; j += kWaves) {
            const int q = qid[j];
            float* gs = fix_s + (size_t)q * k;
            uint32_t* gi = fix_i + (size_t)q * k;
            if (li[j * k] == kInvalidRow) continue;  // nothing found in this block's rows (wave uniform)
            // a lower bound of the global k-th best: lists only improve, so a stale value only merges more
            const float gk = key2f(__hip_atomic_load(&gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (ls[j * k] < gk) continue;
            // All blocks finish their rows at about the same time and queue here with the threshold they saw at the
            // start of the queue.  Every merge raises gthr to the global k-th best, so a waiting block keeps looking
            // at it and leaves the queue as soon as its own best row can no longer enter the list: ~k ln(blocks)
            // merges per query instead of one per block (measured on 1 M clustered rows: 12 ms -> under 1 ms per chunk
            // of 8 flagged queries).
            int give_up = 0;
            if (lane == 0) {
                int expect = 0;
                while (!__hip_atomic_compare_exchange_strong(&fix_lock[q], &expect, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)) {
                    expect = 0;
                    __builtin_amdgcn_s_sleep(8);
                    if (ls[j * k] < key2f(__hip_atomic_load(&gthr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        give_up = 1;
                        break;
                    }
                }
            }
            if (__shfl(give_up, 0)) continue;   // (strictly below the k-th best: ties still merge, lowest ids win)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int i = lane; i < k; i += 64) {
                ws[i] = gs[i];
                wi[i] = gi[i];
            }
            bool changed = false;
            for (int p = 0; p < k; ++p) {
                const uint32_t id = li[j * k + p];
                if (id == kInvalidRow) break;
                if (!wave_insert<uint32_t>(ws, wi, k, ls[j * k + p], id, lane)) break;  // sorted: the rest is worse
                changed = true;
            }
            if (changed) {
                for (int i = lane; i < k; i += 64) {
                    gs[i] = ws[i];
                    gi[i] = wi[i];
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float kth = ws[k - 1];
            if (lane == 0) {
                if (changed && wi[k - 1] != kInvalidRow) atomicMax(&gthr[q], f2key(kth));
                __hip_atomic_store(&fix_lock[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();  // the next chunk re-initialises the LDS lists
    } else {
        // part layout: [q][block][k]
        const int G = gridDim.x;
        for (int i = tid; i < nq_real * k; i += 256) {
            const int j = i / k, p = i - j * k;
            const size_t o = ((size_t)j * G + blockIdx.x) * k + p;
            part_s[o] = ls[i];
            part_i[o] = li[i];
        }
    }
  }
    if constexpr (FIX) {
        // The block that finishes last turns the global lists into the D / I rows of the flagged queries (a launch of
        // its own until round 4: 9 us of every search for nothing, flagged queries being rare).  Every merge above
        // ended with an agent-scope release under the list's lock; the counter add follows this block's last one.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(fix_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (int)gridDim.x - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lock[0] = last;
        }
        __syncthreads();
        if (lock[0]) {
            for (int f = 0; f < nfl; ++f) {
                const int q = flag_list[f];
                for (int i = tid; i < k; i += 256) {
                    const uint32_t id = __hip_atomic_load(&fix_i[(size_t)q * k + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float s = __hip_atomic_load(&fix_s[(size_t)q * k + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // IP: dot; L2: -(squared distance)
                    const bool ok = id != kInvalidRow;
                    D[(size_t)q * k + i] = METRIC == CSS_METRIC_IP ? (ok ? s : -FLT_MAX) : (ok ? -s : FLT_MAX);
                    I[(size_t)q * k + i] = ok ? id_base + (int64_t)id : (int64_t)-1;
                }
            }
        }
    }
}

// Empty index: every slot padded (cannot be reached through the reference: src/storage.py:421-422).
__global__ void k_fill_pad(float* __restrict__ D, int64_t* __restrict__ I, int64_t n, float pad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        D[i] = pad;
        I[i] = -1;
    }
}


// ------------------------------------------------------------------ scan (query batches, MFMA)
// C[128 rows x 128 queries] = X[128 x 768] * Q^T on v_mfma_f32_32x32x2_f32 (exact
// fp32 fmaf chains, 157 TFLOP/s dense peak).  Block = 4 waves; wave w owns query
// columns [32w, 32w+32) x all 128 rows = 4 accumulator tiles (64 AGPR/VGPR).
// Per K-step (BK = 32 floats): X tile and Q tile are staged global -> registers
// -> LDS (double buffered, one barrier per step, loads for step t+1 in flight
// during the MFMAs of step t).  LDS rows are 128 B; the 16-B chunk index is
// XOR-swizzled with (row>>1)&7 so every ds_read_b128 lane group hits 16
// different slots of the 256-B bank row.
// After the last K-step of a row tile the epilogue compares the 64 scores a lane
// holds (one query column per lane) with that query's current k-th best; only
// when something passes is the tile spilled to an LDS scratch and inserted into
// the wave-owned sorted lists.  Thresholds are also exchanged grid-wide (gthr).
constexpr int MF_BM = 128, MF_BN = 128, MF_BK = 32;
#ifndef CSS_MF_LAG
#define CSS_MF_LAG 12
#endif
#ifndef CSS_MF_POLL
#define CSS_MF_POLL 8   // (a power of two)
#endif
constexpr int MF_LAG = CSS_MF_LAG;   // K-steps a block may run ahead of its slowest sibling (k_scan_mfma pacing)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));  // plain clang vector: stays in VGPRs

__device__ __forceinline__ int mf_swz(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <int METRIC>
__global__ __launch_bounds__(256, 1) void k_scan_mfma(const float* __restrict__ xb, const float* __restrict__ xnorm2,
                                                      const float* __restrict__ qpad, int nq_real, int64_t ntotal,
                                                      int dpad, int k, int nstrips, int nqtiles,
                                                      int64_t tiles_per_strip, int* __restrict__ gthr,
                                                      float* __restrict__ part_s, uint32_t* __restrict__ part_i,
                                                      const uint32_t* __restrict__ mask, int* __restrict__ pace) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [2][128][32]
    float* Bs = As + 2 * MF_BM * MF_BK;                         // [2][128][32]
    float* scr = Bs + 2 * MF_BN * MF_BK;                        // [4 waves][32][33]
    float* xn2s = scr + 4 * 32 * 33;                            // [128]
    float* ls = xn2s + MF_BM;                                   // [128][k]
    uint32_t* li = reinterpret_cast<uint32_t*>(ls + MF_BN * k);  // [128][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware decode: blocks l, l+8, l+16, ... (same XCD under round-robin dispatch)
    // walk the query tiles of ONE strip, so the strip's rows are fetched once per XCD L2.
    const int l = blockIdx.x;
    const int strip = (l / (8 * nqtiles)) * 8 + (l & 7);
    const int qtile = (l >> 3) % nqtiles;
    const int64_t ntiles = (ntotal + MF_BM - 1) / MF_BM;
    const int64_t t_begin = (int64_t)strip * tiles_per_strip;
    const int64_t t_end = min(t_begin + tiles_per_strip, ntiles);
    const int q_base = qtile * MF_BN;

    for (int i = tid; i < MF_BN * k; i += 256) {
        ls[i] = -INFINITY;
        li[i] = kInvalidRow;
    }
    __syncthreads();
    if (t_begin >= t_end) {
        for (int i = tid; i < MF_BN * k; i += 256) {
            const int j = i / k, p = i - j * k;
            if (q_base + j < nq_real) {
                const size_t o = ((size_t)(q_base + j) * nstrips + strip) * k + p;
                part_s[o] = -INFINITY;
                part_i[o] = kInvalidRow;
            }
        }
        return;
    }

    const int KT = dpad / MF_BK;
    const int64_t n_it = (t_end - t_begin) * KT;
    // staging map: thread -> (row = (tid>>3) + 32*i, 16-B chunk = tid&7), i = 0..3
    const int srow = tid >> 3, schunk = tid & 7;
    v4f ra[4], rb[4];

// (macros, not lambdas: by-reference lambda captures left ra/rb in scratch memory)
#define MF_GLOAD(RT, KT)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        int64_t row_ = (RT) * MF_BM + srow + 32 * i;                                                         \
        row_ = row_ < ntotal ? row_ : ntotal - 1;                                                            \
        ra[i] = *reinterpret_cast<const v4f*>(xb + row_ * (int64_t)dpad + (KT) * MF_BK + schunk * 4);     \
        rb[i] = *reinterpret_cast<const v4f*>(qpad + (int64_t)(q_base + srow + 32 * i) * dpad +           \
                                                 (KT) * MF_BK + schunk * 4);                                 \
    }
#define MF_SSTORE(BUF)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        *reinterpret_cast<v4f*>(As + (BUF) * MF_BM * MF_BK + mf_swz(srow + 32 * i, schunk)) = ra[i];      \
        *reinterpret_cast<v4f*>(Bs + (BUF) * MF_BN * MF_BK + mf_swz(srow + 32 * i, schunk)) = rb[i];      \
    }

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int jq = wave * 32 + fr;  // this lane's query column inside the tile
    float thr_g = -INFINITY;

    MF_GLOAD(t_begin, 0)
    MF_SSTORE(0)
    __syncthreads();
    int cur = 0;
    int64_t rt = t_begin;  // row tile / K-step of the tile being computed
    int kt = 0;
    for (int64_t it = 0; it < n_it; ++it) {
        // Sibling pacing: the nqtiles blocks of a strip sit on one XCD and read the same rows.  Every CSS_MF_POLL-th
        // K-step a block publishes its step count and looks at its siblings'; it does not run more than MF_LAG steps
        // ahead of the slowest, so a K-step of rows (16 KB) is still in that XCD's L2 when the others ask for it.
        // Unpaced, the siblings drift apart with their insert work and each fetches the strip from HBM again.
        // Measured at 10 M rows x 256 queries (rocprofv3 --pmc FETCH_SIZE, 30.72 GB algorithmic): unpaced 52.0 GB
        // (1.69 x) in 49.0 ms; poll 8 / lag 12: 34.0 GB (1.11 x) in 50.6 ms; poll 4 / lag 8: 33.2 GB, 51.9 ms; poll
        // 16 / lag 16: 43.0 GB, 49.8 ms.  The kernel is bound by the fp32 MFMA pipe, not by HBM, so the saved traffic
        // buys no time here (it frees HBM for whatever else runs on the chip); CSS_KNN_PACE=0 turns it off.
        // The look is issued here and used after the MFMAs of the step; the spin is bounded, so a sibling that is
        // not resident only costs a wait.
        int sib_lo = 1 << 30;
        const bool pace_now = pace != nullptr && tid == 0 && (it & (CSS_MF_POLL - 1)) == 0;
        if (pace_now) {
            __hip_atomic_store(pace + (size_t)strip * nqtiles + qtile, (int)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int j = 0; j < nqtiles; ++j)
                sib_lo = min(sib_lo, __hip_atomic_load(pace + (size_t)strip * nqtiles + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        if (it + 1 < n_it) {
            const int64_t nrt = kt + 1 < KT ? rt : rt + 1;
            const int nkt = kt + 1 < KT ? kt + 1 : 0;
            MF_GLOAD(nrt, nkt)
        }
        const float* A = As + cur * MF_BM * MF_BK;
        const float* B = Bs + cur * MF_BN * MF_BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // 8 k-values per chunk pair: lane half fh takes chunk 2c+fh
            const v4f b = *reinterpret_cast<const v4f*>(B + mf_swz(jq, 2 * c + fh));
            v4f a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const v4f*>(A + mf_swz(32 * m + fr, 2 * c + fh));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].x, b.x, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].y, b.y, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].z, b.z, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m].w, b.w, acc[m], 0, 0, 0);
            }
        }
        if (kt == KT - 1) {
            // ---------------- epilogue of row tile rt ----------------
            const int64_t row0 = rt * MF_BM;
            if constexpr (METRIC == CSS_METRIC_L2) {
                // s = 2 x.q - ||x||^2  (||q||^2 is added in the final merge)
                if (tid < MF_BM) xn2s[tid] = row0 + tid < ntotal ? xnorm2[row0 + tid] : 0.f;
                __syncthreads();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[m][r] = 2.f * acc[m][r] - xn2s[32 * m + (r & 3) + 8 * (r >> 2) + 4 * fh];
            }
            thr_g = key2f(__hip_atomic_load(&gthr[q_base + jq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            float thr_l = ls[jq * k + (k - 1)];
            const float thr = fmaxf(thr_l, thr_g);
            const bool full_tile = row0 + MF_BM <= ntotal;
            bool anyp = false;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) anyp |= acc[m][r] >= thr;
            anyp &= jq + q_base < nq_real;
            if (__ballot(anyp) != 0ull) {
                float* S = scr + wave * (32 * 33);
                bool changed = false;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * fh) * 33 + fr] = acc[m][r];
                    for (int rr = 0; rr < 32; ++rr) {
                        const int64_t row = row0 + 32 * m + rr;
                        const float sv = S[rr * 33 + fr];
                        const bool pass = fh == 0 && (full_tile || row < ntotal) && q_base + jq < nq_real &&
                                          (mask == nullptr || ((mask[row >> 5] >> (row & 31)) & 1u)) &&
                                          sv >= thr_l && sv >= thr_g;
                        unsigned long long mk = __ballot(pass);
                        if (mk == 0ull) continue;
                        while (mk) {
                            const int src = __ffsll((long long)mk) - 1;
                            mk &= mk - 1;
                            const float cs = __shfl(sv, src);
                            const int cj = wave * 32 + src;
                            const bool ins = wave_insert<uint32_t>(ls + cj * k, li + cj * k, k, cs, (uint32_t)row, lane);
                            changed |= ins && (fr == src);
                        }
                        thr_l = ls[jq * k + (k - 1)];
                    }
                }
                if (changed && fh == 0 && thr_l > thr_g) atomicMax(&gthr[q_base + jq], f2key(thr_l));
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
        if (pace_now && sib_lo + MF_LAG < (int)it) {
            int spin = 0;
            for (; spin < 2048 && sib_lo + MF_LAG < (int)it; ++spin) {
                __builtin_amdgcn_s_sleep(4);
                sib_lo = 1 << 30;
                for (int j = 0; j < nqtiles; ++j)
                    sib_lo = min(sib_lo, __hip_atomic_load(pace + (size_t)strip * nqtiles + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            if (spin == 2048) pace = nullptr;   // a sibling is not running: stop waiting for it (thread 0's copy is the one used)
        }
        if (it + 1 < n_it) {
            MF_SSTORE(cur ^ 1)
        }
        __syncthreads();
        cur ^= 1;
        if (++kt == KT) {
            kt = 0;
            ++rt;
        }
    }
#undef MF_GLOAD
#undef MF_SSTORE
    // (a block that is done must not hold its siblings back)
    if (pace != nullptr && tid == 0)
        __hip_atomic_store(pace + (size_t)strip * nqtiles + qtile, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // part layout: [q][strip][k]
    for (int i = tid; i < MF_BN * k; i += 256) {
        const int j = i / k, p = i - j * k;
        if (q_base + j < nq_real) {
            const size_t o = ((size_t)(q_base + j) * nstrips + strip) * k + p;
            part_s[o] = ls[i];
            part_i[o] = li[i];
        }
    }
}

// Split-bf16 variant of k_scan_mfma: every fp32 operand is split on the fly into a bf16 pair
// (h, l), x ~= h + l to 16 significant bits, and the four products (h+l).(h+l) run on
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate, 4 MFMAs instead of 8x2): ~4x the
// throughput at fp32-grade error (operand truncation 2^-17, random sign over 768 terms:
// ~1e-7 on unit vectors, same order as fp32 accumulation order effects).  The index stays
// fp32 in HBM; queries are pre-split once per search (k_split_queries).
// NW waves x 32 queries = BN query columns per block; every wave holds MT 32-row tiles (BM = 32*MT rows).
// (NW, MT) = (4, 4): 128x128 tile, <= 80 KiB LDS, two blocks per CU; (8, 8): 256x256 tile, one block of
// 8 waves per CU, half the operand bytes per MFMA (the CU's L2->LDS path is the scarce resource).
// The l_x.l_q product is <= 2^-18 |x.q| per term -- the size of the split's own truncation error -- and is
// not formed (three MFMAs per fp32-grade product instead of four).
constexpr bool kSplitLowLow = false;
template <int METRIC, int NW, int MT>
__global__ __launch_bounds__(64 * NW) void k_scan_mfma_split(const float* __restrict__ xb, const float* __restrict__ xnorm2,
                                                      const unsigned short* __restrict__ qsplit, int nq_real, int64_t ntotal,
                                                      int dpad, int k, int nstrips, int nqtiles,
                                                      int64_t tiles_per_strip, int* __restrict__ gthr,
                                                      float* __restrict__ part_s, uint32_t* __restrict__ part_i,
                                                      const uint32_t* __restrict__ mask) {
    constexpr int NT = 64 * NW, BM = 32 * MT, BN = 32 * NW;
    constexpr int APASS = BM * 4 / NT;   // A staging passes (rows per pass = NT/4)
    static_assert(NW <= MT && APASS >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [2][BM][32]
    float* Bs = As + 2 * BM * MF_BK;                         // [2][BM][32]
    float* xn2s = Bs + 2 * BN * MF_BK;                       // [BM]
    int* eflag = reinterpret_cast<int*>(xn2s + BM);          // [2] slow-path votes (alternating)
    float* ls = xn2s + BM + 4;                               // [BM][k]
    uint32_t* li = reinterpret_cast<uint32_t*>(ls + BN * k);  // [BM][k]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware decode: blocks l, l+8, l+16, ... (same XCD under round-robin dispatch)
    // walk the query tiles of ONE strip, so the strip's rows are fetched once per XCD L2.
    const int l = blockIdx.x;
    const int strip = (l / (8 * nqtiles)) * 8 + (l & 7);
    const int qtile = (l >> 3) % nqtiles;
    const int64_t ntiles = (ntotal + BM - 1) / BM;
    const int64_t t_begin = (int64_t)strip * tiles_per_strip;
    const int64_t t_end = min(t_begin + tiles_per_strip, ntiles);
    const int q_base = qtile * BN;

    for (int i = tid; i < BN * k; i += NT) {
        ls[i] = -INFINITY;
        li[i] = kInvalidRow;
    }
    if (tid < 2) eflag[tid] = 0;
    __syncthreads();
    if (t_begin >= t_end) {
        for (int i = tid; i < BN * k; i += NT) {
            const int j = i / k, p = i - j * k;
            if (q_base + j < nq_real) {
                const size_t o = ((size_t)(q_base + j) * nstrips + strip) * k + p;
                part_s[o] = -INFINITY;
                part_i[o] = kInvalidRow;
            }
        }
        return;
    }

    const int KT = dpad / MF_BK;
    const int64_t n_it = (t_end - t_begin) * KT;
    // staging map: thread -> (row = (tid>>3) + 32*i, 16-B chunk = tid&7), i = 0..3
    const int srow = tid >> 3, schunk = tid & 7;
    const int arow = tid >> 2, akg = tid & 3;
    // global loads run two K-steps ahead in two register sets when one wave per SIMD must hide the
    // whole L2/HBM latency (NW = 4); with two waves per SIMD (NW = 8) one set is enough and the
    // freed registers let the compiler read LDS fragments ahead of the MFMAs
    constexpr bool PF2 = NW < 8;
    v4f ra0[2 * APASS], rb0[4], ra1[PF2 ? 2 * APASS : 1], rb1[PF2 ? 4 : 1];

// (macros, not lambdas: by-reference lambda captures left ra/rb in scratch memory)
// A: thread -> row (tid>>2) + 64*i, 8 consecutive k = (tid&3)*8 (two float4), converted here to
// the bf16 pair (h, l) with x ~= h + l (16 significant bits); LDS row = [h k0..31 | l k0..31].
// B: the pre-split queries are copied 16 B at a time (thread -> row (tid>>3) + 32*i, chunk tid&7).
#define MF_GLOAD(RT, KT, RA, RB)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < APASS; ++i) {                                                      \
        int64_t row_ = (RT) * BM + arow + (NT / 4) * i;                                                         \
        row_ = row_ < ntotal ? row_ : ntotal - 1;                                                            \
        const float* p_ = xb + row_ * (int64_t)dpad + (KT) * MF_BK + akg * 8;                                \
        RA[2 * i] = *reinterpret_cast<const v4f*>(p_);                                                       \
        RA[2 * i + 1] = *reinterpret_cast<const v4f*>(p_ + 4);                                               \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        RB[i] = *reinterpret_cast<const v4f*>(qsplit + ((int64_t)(q_base + srow + (NT / 8) * i) * (dpad / MF_BK) + (KT)) * 64 + \
                                              schunk * 8);                                                   \
    }
#define MF_SSTORE(BUF, RA, RB)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < APASS; ++i) {                                                      \
        v8bf h_, l_;                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                      \
            const float x_ = j < 4 ? RA[2 * i][j] : RA[2 * i + 1][j - 4];                                    \
            const __bf16 hb_ = (__bf16)x_;                                                                   \
            h_[j] = hb_;                                                                                     \
            l_[j] = (__bf16)(x_ - (float)hb_);                                                               \
        }                                                                                                    \
        char* A_ = reinterpret_cast<char*>(As + (BUF) * BM * MF_BK);                                      \
        *reinterpret_cast<v8bf*>(A_ + mf_swz(arow + (NT / 4) * i, akg) * 4) = h_;                                  \
        *reinterpret_cast<v8bf*>(A_ + mf_swz(arow + (NT / 4) * i, 4 + akg) * 4) = l_;                              \
    }                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
        *reinterpret_cast<v4f*>(Bs + (BUF) * BN * MF_BK + mf_swz(srow + (NT / 8) * i, schunk)) = RB[i];         \
    }

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    const int jq = wave * 32 + fr;  // this lane's query column inside the tile
    float thr_g = -INFINITY;

    // software pipeline: global loads run TWO K-steps ahead in two register sets (one K-step of
    // MFMAs is shorter than the L2/HBM latency), LDS is double buffered one step ahead.
    int64_t rt = t_begin;  // row tile / K-step of the tile being computed
    int kt = 0;
    int64_t lrt = t_begin;  // (row tile, K-step) of the next global load to issue
    int lkt = 0;
#define MF_ADVANCE_LOAD()   \
    if (++lkt == KT) {      \
        lkt = 0;            \
        ++lrt;              \
    }
    MF_GLOAD(lrt, lkt, ra0, rb0)
    MF_ADVANCE_LOAD()
    MF_SSTORE(0, ra0, rb0)
    if (n_it > 1) {
        MF_GLOAD(lrt, lkt, ra0, rb0)  // tile 1 (PF2: set 0 again, tile 2 goes to set 1 below)
        MF_ADVANCE_LOAD()
    }
    if constexpr (PF2) {
        if (n_it > 2) {
            MF_GLOAD(lrt, lkt, ra1, rb1)
            MF_ADVANCE_LOAD()
        }
    }
    __syncthreads();
    int cur = 0;
// PF2 step: compute tile it | store tile it+1 (set RA/RB) to LDS | reload that set with tile it+3
#define MF_STEP2(RA, RB)                                                                   \
    {                                                                                      \
        MF_COMPUTE_AND_EPILOGUE()                                                          \
        if (it + 1 < n_it) {                                                               \
            MF_SSTORE(cur ^ 1, RA, RB)                                                     \
        }                                                                                  \
        if (it + 3 < n_it) {                                                               \
            MF_GLOAD(lrt, lkt, RA, RB)                                                     \
            MF_ADVANCE_LOAD()                                                              \
        }                                                                                  \
        __syncthreads();                                                                   \
        cur ^= 1;                                                                          \
        if (++kt == KT) {                                                                  \
            kt = 0;                                                                        \
            ++rt;                                                                          \
        }                                                                                  \
        ++it;                                                                              \
    }
// single-set step: compute tile it | store tile it+1 | load tile it+2 into the same set
#define MF_STEP1()                                                                         \
    {                                                                                      \
        MF_COMPUTE_AND_EPILOGUE()                                                          \
        if (it + 1 < n_it) {                                                               \
            MF_SSTORE(cur ^ 1, ra0, rb0)                                                   \
        }                                                                                  \
        if (it + 2 < n_it) {                                                               \
            MF_GLOAD(lrt, lkt, ra0, rb0)                                                   \
            MF_ADVANCE_LOAD()                                                              \
        }                                                                                  \
        __syncthreads();                                                                   \
        cur ^= 1;                                                                          \
        if (++kt == KT) {                                                                  \
            kt = 0;                                                                        \
            ++rt;                                                                          \
        }                                                                                  \
        ++it;                                                                              \
    }
    auto compute_and_epilogue = [&]() {
        const float* A = As + cur * BM * MF_BK;
        const float* B = Bs + cur * BN * MF_BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // two 16-wide k-steps per 32-k tile; lane half fh takes chunk 2ks+fh
            const v8bf bh = *reinterpret_cast<const v8bf*>(B + mf_swz(jq, 2 * ks + fh));
            const v8bf bl = *reinterpret_cast<const v8bf*>(B + mf_swz(jq, 4 + 2 * ks + fh));
            // A fragments are read one 32-row tile ahead of the MFMAs that consume them, so the
            // LDS latency hides behind the previous tile's four MFMAs.
            v8bf ah_n = *reinterpret_cast<const v8bf*>(A + mf_swz(fr, 2 * ks + fh));
            v8bf al_n = *reinterpret_cast<const v8bf*>(A + mf_swz(fr, 4 + 2 * ks + fh));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const v8bf ah = ah_n, al = al_n;
                if (m + 1 < MT) {
                    ah_n = *reinterpret_cast<const v8bf*>(A + mf_swz(32 * (m + 1) + fr, 2 * ks + fh));
                    al_n = *reinterpret_cast<const v8bf*>(A + mf_swz(32 * (m + 1) + fr, 4 + 2 * ks + fh));
                }
                // (h_x + l_x).(h_q + l_q): small terms first
                if constexpr (kSplitLowLow) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
            }
        }
        if (kt == KT - 1) {
            // ---------------- epilogue of row tile rt ----------------
            const int64_t row0 = rt * BM;
            if constexpr (METRIC == CSS_METRIC_L2) {
                // s = 2 x.q - ||x||^2  (||q||^2 is added in the final merge)
                if (tid < BM) xn2s[tid] = row0 + tid < ntotal ? xnorm2[row0 + tid] : 0.f;
                __syncthreads();
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[m][r] = 2.f * acc[m][r] - xn2s[32 * m + (r & 3) + 8 * (r >> 2) + 4 * fh];
            }
            thr_g = key2f(__hip_atomic_load(&gthr[q_base + jq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            float thr_l = ls[jq * k + (k - 1)];
            const float thr = fmaxf(thr_l, thr_g);
            const bool full_tile = row0 + BM <= ntotal;
            const bool colok = jq + q_base < nq_real;
            unsigned hitm = 0;  // bit m: this lane's query has a candidate in 32-row tile m
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bool h = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) h |= acc[m][r] >= thr;
                hitm |= (h && colok) ? (1u << m) : 0u;
            }
            const bool anyp = hitm != 0;
            // Block-uniform vote (one extra barrier per row tile): the slow path borrows the
            // just-consumed A staging buffer as its scratch, which keeps the block under 80 KiB
            // of LDS (two blocks per CU: one block's MFMAs overlap the other's staging).
            int* fl = eflag + (int)(rt & 1);
            if (__ballot(anyp) != 0ull && lane == 0) *fl = 1;
            __syncthreads();
            if (*fl) {
                if (tid == 0) eflag[(int)((rt + 1) & 1)] = 0;  // re-arm the other flag for the next row tile
                float* S = const_cast<float*>(As + cur * BM * MF_BK) + wave * (32 * 32);
                bool changed = false;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    if (__ballot((hitm >> m) & 1u) == 0ull) continue;  // wave-uniform: no candidate in this 32-row tile
#pragma unroll
                    for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * fh) * 32 + fr] = acc[m][r];
                    for (int rr = 0; rr < 32; ++rr) {
                        const int64_t row = row0 + 32 * m + rr;
                        const float sv = S[rr * 32 + fr];
                        const bool pass = fh == 0 && (full_tile || row < ntotal) && q_base + jq < nq_real &&
                                          (mask == nullptr || ((mask[row >> 5] >> (row & 31)) & 1u)) &&
                                          sv >= thr_l && sv >= thr_g;
                        unsigned long long mk = __ballot(pass);
                        if (mk == 0ull) continue;
                        while (mk) {
                            const int src = __ffsll((long long)mk) - 1;
                            mk &= mk - 1;
                            const float cs = __shfl(sv, src);
                            const int cj = wave * 32 + src;
                            const bool ins = wave_insert<uint32_t>(ls + cj * k, li + cj * k, k, cs, (uint32_t)row, lane);
                            changed |= ins && (fr == src);
                        }
                        thr_l = ls[jq * k + (k - 1)];
                    }
                }
                if (changed && fh == 0 && thr_l > thr_g) atomicMax(&gthr[q_base + jq], f2key(thr_l));
            } else if (tid == 0) {
                eflag[(int)((rt + 1) & 1)] = 0;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }

    };
#define MF_COMPUTE_AND_EPILOGUE() compute_and_epilogue();
    // Invariant at the top of step `it`: LDS buf[cur] holds tile it; register set S(it+1) holds tile
    // it+1 (already loaded); PF2: set S(it+2) holds tile it+2.  S alternates 0,1 for PF2, is always 0 otherwise.
    if constexpr (PF2) {
        for (int64_t it = 0; it < n_it;) {
            // tile it+1 is in set 0 when `it` is even; tile it+3 is loaded into that set after it is stored
            MF_STEP2(ra0, rb0)
            if (it >= n_it) break;
            MF_STEP2(ra1, rb1)
        }
    } else {
        for (int64_t it = 0; it < n_it;) {
            MF_STEP1()
        }
    }
#undef MF_STEP1
#undef MF_STEP2
#undef MF_ADVANCE_LOAD
#undef MF_COMPUTE_AND_EPILOGUE
#undef MF_GLOAD
#undef MF_SSTORE
    // part layout: [q][strip][k]
    for (int i = tid; i < BN * k; i += NT) {
        const int j = i / k, p = i - j * k;
        if (q_base + j < nq_real) {
            const size_t o = ((size_t)(q_base + j) * nstrips + strip) * k + p;
            part_s[o] = ls[i];
            part_i[o] = li[i];
        }
    }
}



#include "css_knn_coarse.h"

// qpad [nq_pad][dpad] fp32 -> qsplit [nq_pad][dpad/32][h(32) | l(32)] bf16
__global__ void k_split_queries(const float* __restrict__ qpad, unsigned short* __restrict__ qsplit, int64_t n, int dpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // element index
    if (i >= n * dpad) return;
    const int64_t row = i / dpad;
    const int c = (int)(i - row * dpad), kt = c >> 5, kk = c & 31;
    const float x = qpad[i];
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    unsigned short* o = qsplit + (row * (dpad >> 5) + kt) * 64;
    o[kk] = __builtin_bit_cast(unsigned short, h);
    o[32 + kk] = __builtin_bit_cast(unsigned short, l);
}

// ------------------------------------------------------------------ final merge
// One block per query merges the G per-block lists (each sorted best-first) into
// the final top-k, RANK-MAJOR: round r looks at rank r of every list that is still
// alive.  A list whose rank-r entry cannot enter the current top-k is dead for
// good (its later ranks are worse and the k-th best only improves), so after the
// first round only a handful of lists stay alive.  Round 0 (all G heads) is a
// block-wide bitonic sort in LDS; later rounds append the few survivors to an LDS
// buffer and wave 0 inserts them.  G > 2048 is handled in chunks of 2048 lists.
constexpr int kMergeCap = 2048;
constexpr int kMergePerThread = kMergeCap / 256;
template <int METRIC>
__global__ __launch_bounds__(256) void k_merge_final(const float* __restrict__ part_s,
                                                     const uint32_t* __restrict__ part_i, int G, int k,
                                                     const int* __restrict__ gthr, const float* __restrict__ qnorm2,
                                                     int64_t id_base, float* __restrict__ D, int64_t* __restrict__ I,
                                                     int l2_expanded, float* __restrict__ cs_out,
                                                     uint32_t* __restrict__ ci_out, int* __restrict__ cn_out) {
    __shared__ float fs[CSS_KERNEL_MAX_K];
    __shared__ uint32_t fi[CSS_KERNEL_MAX_K];
    __shared__ float cs[kMergeCap];
    __shared__ uint32_t ci[kMergeCap];
    __shared__ int cnt;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)q * G * k;
    const float gfloor = key2f(gthr[q]);  // valid lower bound of the global k-th best
    for (int i = tid; i < k; i += 256) {
        fs[i] = -INFINITY;
        fi[i] = kInvalidRow;
    }
    __syncthreads();
    for (int c0 = 0; c0 < G; c0 += kMergeCap) {
        const int nb = min(kMergeCap, G - c0);  // lists in this chunk
        bool alive[kMergePerThread];
#pragma unroll
        for (int j = 0; j < kMergePerThread; ++j) alive[j] = tid + 256 * j < nb;
        int r0 = 0;
        if (c0 == 0) {
            // round 0: bitonic sort (best first) of the nb heads, padded to a power of two
            int n2 = 1;
            while (n2 < nb) n2 <<= 1;
            n2 = max(n2, 256);
            for (int i = tid; i < n2; i += 256) {
                float sv = -INFINITY;
                uint32_t iv = kInvalidRow;
                if (i < nb) {
                    const size_t o = base + (size_t)(c0 + i) * k;
                    const uint32_t id = part_i[o];
                    const float v = part_s[o];
                    if (id != kInvalidRow && v >= gfloor) {
                        sv = v;
                        iv = id;
                    }
                }
                cs[i] = sv;
                ci[i] = iv;
            }
            __syncthreads();
            for (int sz = 2; sz <= n2; sz <<= 1)
                for (int st = sz >> 1; st > 0; st >>= 1) {
                    for (int t = tid; t < n2 / 2; t += 256) {
                        const int lo = 2 * t - (t & (st - 1));  // index with bit `st` clear
                        const int hi = lo + st;
                        const bool desc = (lo & sz) == 0;      // best-first in the first half of each block
                        const float a = cs[lo], b2 = cs[hi];
                        const uint32_t ia = ci[lo], ib = ci[hi];
                        const bool a_first = better<uint32_t>(a, ia, b2, ib);
                        if (a_first != desc) {
                            cs[lo] = b2; ci[lo] = ib;
                            cs[hi] = a;  ci[hi] = ia;
                        }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < k; i += 256) {
                fs[i] = cs[i];   // n2 >= 256 >= k
                fi[i] = ci[i];
            }
            __syncthreads();
            // a head that did not make the top-k kills its list
            const float kth = fs[k - 1];
            const uint32_t kid = fi[k - 1];
#pragma unroll
            for (int j = 0; j < kMergePerThread; ++j) {
                if (!alive[j]) continue;
                const size_t o = base + (size_t)(c0 + tid + 256 * j) * k;
                const uint32_t id = part_i[o];
                const float v = part_s[o];
                // alive iff the head is in the list, i.e. not worse than the k-th entry
                alive[j] = id != kInvalidRow && v >= gfloor && !better<uint32_t>(kth, kid, v, id);
            }
            r0 = 1;
            __syncthreads();
        }
        for (int r = r0; r < k; ++r) {
            if (tid == 0) cnt = 0;
            __syncthreads();
            const float kth = fs[k - 1];
            const uint32_t kid = fi[k - 1];
#pragma unroll
            for (int j = 0; j < kMergePerThread; ++j) {
                if (!alive[j]) continue;
                const size_t o = base + (size_t)(c0 + tid + 256 * j) * k + r;
                const uint32_t id = part_i[o];
                const float v = part_s[o];
                if (id != kInvalidRow && v >= gfloor && better<uint32_t>(v, id, kth, kid)) {
                    const int p = atomicAdd(&cnt, 1);
                    cs[p] = v;
                    ci[p] = id;
                } else {
                    alive[j] = false;
                }
            }
            __syncthreads();
            const int n = cnt;
            if (n == 0) break;  // block-uniform
            if (wave == 0)
                for (int c = 0; c < n; ++c) wave_insert<uint32_t>(fs, fi, k, cs[c], ci[c], lane);
            __syncthreads();
        }
        __syncthreads();
    }
    if (cs_out != nullptr) {
        // candidate-path caller (launch_scan_split_rescore): the k best scan scores, in the scan's own form, become
        // the query's candidate buffer for k_coarse_select<FINAL>
        for (int i = tid; i < k; i += 256) {
            const uint32_t id = fi[i];
            cs_out[(size_t)q * CZ_CAP + i] = id == kInvalidRow ? -INFINITY : fs[i];
            ci_out[(size_t)q * CZ_CAP + i] = id;
        }
        if (tid == 0) cn_out[(size_t)q * CZ_NS] = k;
        return;
    }
    for (int i = tid; i < k; i += 256) {
        const uint32_t id = fi[i];
        float s = fs[i];
        float d;
        if (METRIC == CSS_METRIC_IP) {
            d = id == kInvalidRow ? -FLT_MAX : s;
        } else {
            // scan kernels produce s = -(squared distance) (direct form) or, in the
            // expanded MFMA form, s = 2 x.q - ||x||^2 (||q||^2 added here)
            if (id == kInvalidRow) d = FLT_MAX;
            else if (l2_expanded) d = fmaxf(0.f, qnorm2[q] - s);
            else d = -s;
        }
        D[(size_t)q * k + i] = d;
        I[(size_t)q * k + i] = id == kInvalidRow ? (int64_t)-1 : id_base + (int64_t)id;
    }
}

// Merge of per-shard final results with global int64 ids (multi-GPU exchange).
// Part p holds its [nq, k] scores at Dp + p * stride_d and its ids at Ip + p * stride_i (elements): dense
// [nparts, nq, k] arrays, or the packed exchange records of the sharded search ([ids | scores] per rank).
template <int METRIC>
__global__ __launch_bounds__(64) void k_merge_parts(const float* __restrict__ Dp, const int64_t* __restrict__ Ip,
                                                    int nparts, int64_t stride_d, int64_t stride_i, int64_t nq, int k,
                                                    float* __restrict__ D, int64_t* __restrict__ I) {
    __shared__ float fs[CSS_KERNEL_MAX_K];
    __shared__ int64_t fi[CSS_KERNEL_MAX_K];
    const int64_t q = blockIdx.x;
    const int lane = threadIdx.x;
    for (int i = lane; i < k; i += 64) {
        fs[i] = -INFINITY;
        fi[i] = INT64_MAX;
    }
    for (int p = 0; p < nparts; ++p)
        for (int j = 0; j < k; ++j) {
            const size_t o = (size_t)q * k + j;
            const int64_t id = Ip[(size_t)p * stride_i + o];
            if (id < 0) continue;  // wave uniform
            float s = Dp[(size_t)p * stride_d + o];
            if (METRIC == CSS_METRIC_L2) s = -s;
            wave_insert<int64_t>(fs, fi, k, s, id, lane);
        }
    for (int i = lane; i < k; i += 64) {
        const int64_t id = fi[i];
        const bool empty = id == INT64_MAX;
        float s = fs[i];
        D[q * k + i] = METRIC == CSS_METRIC_IP ? (empty ? -FLT_MAX : s) : (empty ? FLT_MAX : -s);
        I[q * k + i] = empty ? -1 : id;
    }
}

// ---- k beyond CSS_KERNEL_MAX_K (the reference passes k' = min(max_results, ntotal) for any max_results,
// src/storage.py:432): passes of up to CSS_KERNEL_MAX_K results per query, every pass over the rows the earlier
// passes did not return (an exclusion bitmap ANDed with the caller's allow-bitmap), then one sort of the k results.
__global__ void k_mask_init(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int64_t words) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words) dst[i] = src ? src[i] : 0xFFFFFFFFu;
}
__global__ void k_mask_clear(uint32_t* __restrict__ mask, const int64_t* __restrict__ I, int n, int64_t id_base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = I[i] - id_base;
    if (I[i] >= 0 && r >= 0) atomicAnd(&mask[r >> 5], ~(1u << (r & 31)));
}
// one block per query: bitonic sort of its k (score, id) pairs by (score better first, lower id first; pads last)
template <int METRIC>
__global__ __launch_bounds__(1024) void k_sort_rows(float* __restrict__ D, int64_t* __restrict__ I, int k) {
    __shared__ float ss[CSS_MAX_K];
    __shared__ int64_t si[CSS_MAX_K];
    float* Dq = D + (size_t)blockIdx.x * k;
    int64_t* Iq = I + (size_t)blockIdx.x * k;
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        const bool real = i < k && Iq[i] >= 0;
        const float d = real ? Dq[i] : 0.f;
        ss[i] = real ? (METRIC == CSS_METRIC_IP ? d : -d) : -INFINITY;   // larger = better
        si[i] = real ? Iq[i] : INT64_MAX;
    }
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n2 / 2; t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;   // this sub-sequence sorts best-first
                const float a = ss[lo], b = ss[hi];
                const int64_t ia = si[lo], ib = si[hi];
                const bool b_better = b > a || (b == a && ib < ia);
                if (b_better == up) {
                    ss[lo] = b; ss[hi] = a;
                    si[lo] = ib; si[hi] = ia;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const bool empty = si[i] == INT64_MAX;
        Dq[i] = empty ? (METRIC == CSS_METRIC_IP ? -FLT_MAX : FLT_MAX) : (METRIC == CSS_METRIC_IP ? ss[i] : -ss[i]);
        Iq[i] = empty ? -1 : si[i];
    }
}

// The same merge for k beyond CSS_KERNEL_MAX_K (wave_insert holds two list slots per lane): every part is a list sorted
// by the total order (score better first, then lower id), ids are unique over the parts, so the final rank of entry j
// of part p is j + the number of entries of the other parts that precede it -- one binary search per other part.
template <int METRIC>
__global__ __launch_bounds__(256) void k_merge_parts_rank(const float* __restrict__ Dp, const int64_t* __restrict__ Ip,
                                                         int nparts, int64_t stride_d, int64_t stride_i, int64_t nq, int k,
                                                         float* __restrict__ D, int64_t* __restrict__ I) {
    const int64_t q = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        D[q * k + i] = METRIC == CSS_METRIC_IP ? -FLT_MAX : FLT_MAX;
        I[q * k + i] = -1;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nparts * k; e += blockDim.x) {
        const int p = e / k, j = e - p * k;
        const size_t o = (size_t)q * k + j;
        const int64_t id = Ip[(size_t)p * stride_i + o];
        if (id < 0) continue;
        const float d = Dp[(size_t)p * stride_d + o];
        const float sc = METRIC == CSS_METRIC_IP ? d : -d;
        int rank = j;
        for (int p2 = 0; p2 < nparts && rank < k; ++p2) {
            if (p2 == p) continue;
            const float* D2 = Dp + (size_t)p2 * stride_d + (size_t)q * k;
            const int64_t* I2 = Ip + (size_t)p2 * stride_i + (size_t)q * k;
            int lo = 0, hi = k;   // first position of part p2 that does NOT precede (sc, id); pads precede nothing
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t id2 = I2[mid];
                const float d2 = D2[mid];
                const float s2 = METRIC == CSS_METRIC_IP ? d2 : -d2;
                const bool precedes = id2 >= 0 && better<int64_t>(s2, id2, sc, id);
                if (precedes) lo = mid + 1;
                else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) {
            D[q * k + rank] = d;
            I[q * k + rank] = id;
        }
    }
}

__global__ void k_fill_int(int* p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------- host side
// bf16 shadow rows for the coarse scan: kept when the metric is inner product, rows are a whole number
// of 64-element K stages and fp32 + bf16 rows fit in 80 % of the HBM (CSS_KNN_SHADOW=0/1 overrides).
bool want_shadow(css_index* ix, int64_t ncap) {
    if (ix->shadow == 0) return false;
    if (ix->dpad % 64 != 0) return false;
    static const int env_policy = [] {
        const char* e = getenv("CSS_KNN_SHADOW");
        return (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : -1);
    }();
    const int policy = ix->shadow_policy >= 0 ? ix->shadow_policy : env_policy;
    if (policy == 0 || policy == 2) return false;   // (2: int8 rows only, want_i8_only)
    if (policy == 1) return true;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    return (double)ncap * ix->dpad * 6.0 <= 0.8 * (double)tot;
}

// int8 rows for the 1..4-query sweep (k_sweep_coarse_i8): only next to bf16 shadow rows, rows of at most 1024
// elements (the fp32 accumulation term of the sweep's error bound, cz_eps) and 7 bytes per element within 80 % of the HBM
// (CSS_KNN_I8=0 switches them off).
bool want_i8(css_index* ix, int64_t ncap) {
    static const bool env_on = [] {
        const char* e = getenv("CSS_KNN_I8");
        return !(e && e[0] == '0');
    }();
    if (!env_on || ix->dpad % 64 != 0 || ix->dpad > 1024) return false;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    return (double)ncap * ix->dpad * 7.0 <= 0.8 * (double)tot;
}

// int8 rows WITHOUT bf16 rows (5 bytes per element): shards of ~38-46 M rows of 768 floats on a 288 GB GPU, where fp32 +
// bf16 rows no longer fit in 80 % of the HBM but fp32 + int8 rows do -- such an index otherwise re-converts its rows into
// scratch memory on every batched search (6.7 ms per 10 M rows).  Only where the int8 scan and sweep apply (inner
// product, rows a whole number of 256-element K-step pairs, at most 1024 elements); css_index_set_shadow(ix, 2) forces
// it (tests), policy 0 forbids every shadow copy.
bool want_i8_only(css_index* ix, int64_t ncap) {
    static const bool env_on = [] {
        const char* e = getenv("CSS_KNN_I8");
        return !(e && e[0] == '0');
    }();
    if (!env_on || ix->metric != CSS_METRIC_IP || ix->dpad % 256 != 0 || ix->dpad > 1024) return false;
    if (ix->shadow_policy == 0) return false;
    if (ix->shadow_policy == 2) return true;
    if (ix->shadow_policy == 1) return false;   // "always bf16" that did not fit: nothing
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    return (double)ncap * ix->dpad * 5.0 <= 0.8 * (double)tot;
}

// (Re)allocate the row storage for exactly ncap rows, carrying the ntotal existing rows over.
int reallocate_rows(css_index* ix, int64_t ncap) {
    float* nxb = nullptr;
    float* nn2 = nullptr;
    unsigned short* nxh = nullptr;
    hipError_t e = hipMalloc((void**)&nxb, (size_t)ncap * ix->dpad * sizeof(float));
    if (e != hipSuccess) return css::hip_fail(e, "hipMalloc(index rows)", __FILE__, __LINE__);
    e = hipMalloc((void**)&nn2, ((size_t)ncap + 256) * sizeof(float));  // +256: the coarse scan reads whole tiles of norms
    if (e != hipSuccess) {
        (void)hipFree(nxb);
        return css::hip_fail(e, "hipMalloc(index norms)", __FILE__, __LINE__);
    }
    // the shadow can only be carried over (or started on an empty index), never rebuilt here
    const bool can_shadow = ix->xh != nullptr || ix->ntotal == 0;
    if (can_shadow && want_shadow(ix, ncap)) {
        // (+256 rows: k_scan_coarse8 reads whole 256-row tiles; scores of rows >= ntotal are masked)
        if (hipMalloc((void**)&nxh, ((size_t)ncap + 256) * ix->dpad * sizeof(unsigned short)) != hipSuccess) {
            (void)hipGetLastError();  // no room: batched search falls back to the split-operand kernel
            nxh = nullptr;
        }
    }
    // the int8 rows of the few-query sweep ride along with the bf16 ones when 7 bytes per element still fit
    unsigned char* nx8 = nullptr;
    float* nx8s = nullptr;
    const bool i8_only = !nxh && ((ix->x8 != nullptr && ix->xh == nullptr) || ix->ntotal == 0) && want_i8_only(ix, ncap);
    if ((nxh && (ix->x8 != nullptr || ix->ntotal == 0) && want_i8(ix, ncap)) || i8_only) {
        if (hipMalloc((void**)&nx8, ((size_t)ncap + 256) * ix->dpad) != hipSuccess ||
            hipMalloc((void**)&nx8s, ((size_t)ncap + 256) * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            if (nx8) (void)hipFree(nx8);
            nx8 = nullptr;
            nx8s = nullptr;
        }
    }
    if (ix->ntotal > 0) {
        if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent(ix->stream, ix->ingest_ev, 0));
        if (nx8) {
            CSS_HIP_TRY(hipMemcpyAsync(nx8, ix->x8, (size_t)ix->ntotal * ix->dpad, hipMemcpyDeviceToDevice, ix->stream));
            CSS_HIP_TRY(hipMemcpyAsync(nx8s, ix->x8s, (size_t)ix->ntotal * sizeof(float), hipMemcpyDeviceToDevice, ix->stream));
        }
        CSS_HIP_TRY(hipMemcpyAsync(nxb, ix->xb, (size_t)ix->ntotal * ix->dpad * sizeof(float),
                                   hipMemcpyDeviceToDevice, ix->stream));
        CSS_HIP_TRY(hipMemcpyAsync(nn2, ix->xnorm2, (size_t)ix->ntotal * sizeof(float), hipMemcpyDeviceToDevice,
                                   ix->stream));
        if (nxh)
            CSS_HIP_TRY(hipMemcpyAsync(nxh, ix->xh, (size_t)ix->ntotal * ix->dpad * sizeof(unsigned short),
                                       hipMemcpyDeviceToDevice, ix->stream));
        CSS_HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    if (ix->xb) CSS_HIP_TRY(hipFree(ix->xb));
    if (ix->xnorm2) CSS_HIP_TRY(hipFree(ix->xnorm2));
    if (ix->xh) CSS_HIP_TRY(hipFree(ix->xh));
    if (ix->x8) CSS_HIP_TRY(hipFree(ix->x8));
    if (ix->x8s) CSS_HIP_TRY(hipFree(ix->x8s));
    ix->xb = nxb;
    ix->xnorm2 = nn2;
    ix->xh = nxh;
    ix->x8 = nx8;
    ix->x8s = nx8s;
    ix->shadow = nxh ? 1 : 0;
    ix->cap = ncap;
    return CSS_OK;
}

int ensure_capacity(css_index* ix, int64_t need) {
    if (need > ix->cap) {
        int64_t ncap = std::max<int64_t>(need, ix->cap + ix->cap / 2);
        return reallocate_rows(ix, std::max<int64_t>(ncap, 1024));
    }
    if (ix->ntotal == 0 && !ix->xh && ix->shadow < 0 && ix->cap > 0 && want_shadow(ix, ix->cap)) {
        // emptied index (css_index_reset): start a shadow again if there is room now
        if (hipMalloc((void**)&ix->xh, ((size_t)ix->cap + 256) * ix->dpad * sizeof(unsigned short)) != hipSuccess) {
            (void)hipGetLastError();
            ix->xh = nullptr;
        }
        ix->shadow = ix->xh ? 1 : 0;
        if (ix->xh && !ix->x8 && want_i8(ix, ix->cap)) {
            if (hipMalloc((void**)&ix->x8, ((size_t)ix->cap + 256) * ix->dpad) != hipSuccess ||
                hipMalloc((void**)&ix->x8s, ((size_t)ix->cap + 256) * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError();
                if (ix->x8) (void)hipFree(ix->x8);
                ix->x8 = nullptr;
                ix->x8s = nullptr;
            }
        }
    }
    return CSS_OK;
}

int ingest(css_index* ix, const float* x_dev, int64_t n, int normalize, bool synth, uint64_t seed,
           int64_t first_row, hipStream_t st) {
    // One wave per row: a dispatch carries at most 2^32 work-items, i.e. 2^26 rows -- beyond that the rows were silently
    // not written (found with an 80 M-row index on one GPU: every search ended in the exact sweep over garbage rows).
    // 2^24 rows (2^30 work-items) per launch.
    constexpr int64_t kRowsPerLaunch = 1ll << 24;
    for (int64_t c0 = 0; c0 < n; c0 += kRowsPerLaunch) {
        const int64_t nc = std::min<int64_t>(kRowsPerLaunch, n - c0);
        const int64_t r0 = ix->ntotal + c0;
        const unsigned blocks = (unsigned)((nc + 3) / 4);
        float* dst = ix->xb + (size_t)r0 * ix->dpad;
        float* n2 = ix->xnorm2 + r0;
        unsigned short* dh = ix->xh ? ix->xh + (size_t)r0 * ix->dpad : nullptr;
        unsigned char* d8 = ix->x8 ? ix->x8 + (size_t)r0 * ix->dpad : nullptr;
        float* d8s = ix->x8 ? ix->x8s + r0 : nullptr;
        if (synth)
            hipLaunchKernelGGL(k_ingest_rows<true>, dim3(blocks), dim3(256), 0, st, nullptr, dst, n2, nc, ix->dim, ix->dpad,
                               normalize, seed, first_row + c0, dh, ix->maxn2, (float*)nullptr, d8, d8s);
        else
            hipLaunchKernelGGL(k_ingest_rows<false>, dim3(blocks), dim3(256), 0, st, x_dev + (size_t)c0 * ix->dim, dst, n2, nc,
                               ix->dim, ix->dpad, normalize, 0ull, 0ll, dh, ix->maxn2, (float*)nullptr, d8, d8s);
        CSS_LAUNCH_CHECK();
    }
    return CSS_OK;
}

// ---- environment switches (experiments and verification): read once, never written afterwards
struct KnnEnv {
    int batch = 0;        // CSS_KNN_BATCH: "split" = 1 (split-operand candidate scan for every batch), "fp32" = 2 (fp32-MFMA scan)
    bool exact_k = true;        // CSS_KNN_EXACTK=0: two-eps thresholds from coarse scores only in the int8 scan's selects
    int batch_i8 = 1;           // CSS_KNN_SCAN=bf16 / i8: batches always scan the bf16 / the int8 shadow rows (1 = where it pays)
    bool sweep_i8 = true;       // CSS_KNN_SWEEP=bf16: 1..4 queries sweep the bf16 shadow rows even where int8 rows exist (A/B runs)
    bool eps_measured = true;   // CSS_KNN_EPS=apriori: unit-roundoff error band instead of the measured one (cz_eps)
    int growth = 0;       // CSS_KNN_GROWTH=4|8|16: growth factor of the nested row sample (batched MFMA cascade); 0: by k
    int growth_sweep = 0;   // CSS_KNN_GROWTH_SWEEP=4|8|16: the same for the few-query sweep cascades (0 = by shape: launch_scan_coarse)
    int sweep_fused = 1;    // CSS_KNN_SWEEP_FUSED=0 / 2: the 1..4-query cascade never / always as ONE launch (k_sweep_cascade); 1 = where it pays
    int qreg = 1;           // CSS_KNN_QREG=0: the int8 batch scan's later stages on k_scan_coarse8 instead of k_scan_qreg_i8 (A/B runs)
    int qreg_min = 1024;    // CSS_KNN_QREG_MIN=<tile tasks>: stages with fewer (row tile, query tile) pairs stay on k_scan_coarse8 (two per block: measured, launch_scan_coarse)
    int sweep_mfma = 1;     // CSS_KNN_SWEEP_MFMA=0: 3..32 queries never take the int8-MFMA sweep (A/B runs); 2: at every index size (tests)
    int sweep_maxq = -1;    // CSS_KNN_SWEEP_MAXQ=n: searches of up to n (0..4) queries take the sweep cascade (A/B runs); -1 = by size
    int fs_spins = CZ_FS_SPINS;   // CSS_KNN_FS_SPINS=n: polls before a waiting wave of k_sweep_cascade gives up (tests: 0 = at once)
    int fs_blocks = 0;      // CSS_KNN_FS_BLOCKS=n: at most n blocks of k_sweep_cascade per CU (A/B runs); 0 = what fits
    int mfma_shape = 16;  // CSS_KNN_MFMA=32: 32x32x16 MFMA in k_scan_coarse (A/B runs)
    int pacing = 1;       // CSS_KNN_PACE=0: no sibling pacing in k_scan_coarse (A/B runs)
    int dbg = 0;          // CSS_KNN_DBG: timing ablations of k_scan_coarse (results are wrong when set)
    int loop8 = 1;        // CSS_KNN_LOOP=old: the round-1 main loop (k_scan_coarse) instead of k_scan_coarse8 (A/B runs)
    int pass2 = 1;        // CSS_KNN_PASS2=0: flagged queries go straight to the exact fp32 sweep (A/B runs)
    int noshadow_ranges = 1;   // CSS_KNN_NOSHADOW=split: shadow-less batches through the split-operand scan (A/B runs)
};
const KnnEnv& knn_env() {
    static const KnnEnv env = [] {
        KnnEnv e;
        if (const char* m = getenv("CSS_KNN_BATCH")) e.batch = std::string(m) == "split" ? 1 : (std::string(m) == "fp32" ? 2 : 0);
        if (const char* m = getenv("CSS_KNN_EPS")) e.eps_measured = strcmp(m, "apriori") != 0;
        if (const char* m = getenv("CSS_KNN_SWEEP")) e.sweep_i8 = strcmp(m, "bf16") != 0;
        if (const char* m = getenv("CSS_KNN_EXACTK")) e.exact_k = m[0] != '0';
        if (const char* m = getenv("CSS_KNN_SCAN")) e.batch_i8 = strcmp(m, "bf16") == 0 ? 0 : (strcmp(m, "i8") == 0 ? 2 : 1);
        if (const char* m = getenv("CSS_KNN_GROWTH")) {
            const int v = atoi(m);
            e.growth = (v == 4 || v == 8 || v == 16) ? v : 0;
        }
        if (const char* m = getenv("CSS_KNN_GROWTH_SWEEP")) {
            const int v = atoi(m);
            e.growth_sweep = (v == 4 || v == 8 || v == 16) ? v : 0;
        }
        if (const char* m = getenv("CSS_KNN_SWEEP_FUSED")) e.sweep_fused = m[0] == '0' ? 0 : (m[0] == '2' ? 2 : 1);
        if (const char* m = getenv("CSS_KNN_FS_BLOCKS")) e.fs_blocks = std::max(0, atoi(m));
        if (const char* m = getenv("CSS_KNN_FS_SPINS")) e.fs_spins = std::max(0, atoi(m));
        if (const char* m = getenv("CSS_KNN_QREG")) e.qreg = m[0] == '0' ? 0 : 1;
        if (const char* m = getenv("CSS_KNN_QREG_MIN")) e.qreg_min = std::max(0, atoi(m));
        if (const char* m = getenv("CSS_KNN_SWEEP_MFMA")) e.sweep_mfma = m[0] == '0' ? 0 : (m[0] == '2' ? 2 : 1);
        if (const char* m = getenv("CSS_KNN_SWEEP_MAXQ")) e.sweep_maxq = std::min(4, std::max(0, atoi(m)));
        if (const char* m = getenv("CSS_KNN_MFMA")) e.mfma_shape = atoi(m) == 32 ? 32 : 16;
        if (const char* m = getenv("CSS_KNN_PACE")) e.pacing = m[0] == '0' ? 0 : 1;
        if (const char* m = getenv("CSS_KNN_DBG")) e.dbg = atoi(m);
        if (const char* m = getenv("CSS_KNN_LOOP")) e.loop8 = std::string(m) == "old" ? 0 : 1;
        if (const char* m = getenv("CSS_KNN_PASS2")) e.pass2 = m[0] == '0' ? 0 : 1;
        if (const char* m = getenv("CSS_KNN_NOSHADOW")) e.noshadow_ranges = std::string(m) == "split" ? 0 : 1;
        return e;
    }();
    return env;
}

// geometry of the exact fp32 sweep (k_scan_small) for this index and k
struct SweepGeom {
    int nq_sweep;   // queries per sweep that fit the kernel's LDS budget (1..16)
    int G;          // blocks
    int64_t gpb;    // row groups (of 4) per block
};

template <int NQ, int TT, int METRIC, bool FIX>
int launch_scan_small_t(css_index* ix, const float* qpad, int nq_real, int k, int* gthr, const SweepGeom& sg,
                        hipStream_t st, const int* flag_list, const int* nflag, float* fix_s, uint32_t* fix_i,
                        int* fix_lock, float* D_dev, int64_t* I_dev) {
    const int T = ix->dpad / 64;
    const size_t lds = (size_t)NQ * ix->dpad * 4 + (size_t)NQ * k * 8 + NQ * 8 + (FIX ? (size_t)4 * k * 8 : 0);
    auto kern = k_scan_small<NQ, TT, METRIC, FIX>;
    int rc;
    if (lds > 48 * 1024 && (rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    ProfScope ps(FIX ? "knn_fix_scan" : "knn_scan_small", st);
    hipLaunchKernelGGL(kern, dim3(sg.G), dim3(256), lds, st, (const float4*)ix->xb, qpad, ix->ntotal, T, k, sg.gpb, gthr,
                       ix->part_s, ix->part_i, nq_real, ix->cur_mask, flag_list, nflag, fix_s, fix_i, fix_lock,
                       FIX ? const_cast<int*>(nflag) + 1 : (int*)nullptr,   // the blocks-done word sits behind the flagged count
                       ix->id_base, D_dev, I_dev);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

template <int NQ, bool FIX>
int launch_scan_small_nq(css_index* ix, const float* qpad, int nq_real, int k, int* gthr, const SweepGeom& sg,
                         hipStream_t st, const int* flag_list = nullptr, const int* nflag = nullptr,
                         float* fix_s = nullptr, uint32_t* fix_i = nullptr, int* fix_lock = nullptr, float* D_dev = nullptr,
                         int64_t* I_dev = nullptr) {
    const bool ip = ix->metric == CSS_METRIC_IP;
    if (ix->dpad == 768) {
        return ip ? launch_scan_small_t<NQ, 12, CSS_METRIC_IP, FIX>(ix, qpad, nq_real, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev)
                  : launch_scan_small_t<NQ, 12, CSS_METRIC_L2, FIX>(ix, qpad, nq_real, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev);
    }
    return ip ? launch_scan_small_t<NQ, 0, CSS_METRIC_IP, FIX>(ix, qpad, nq_real, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev)
              : launch_scan_small_t<NQ, 0, CSS_METRIC_L2, FIX>(ix, qpad, nq_real, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev);
}

int grow_part(css_index* ix, size_t entries) {
    if (entries <= ix->part_cap) return CSS_OK;
    size_t ncap = std::max(entries, ix->part_cap * 2);
    if (ix->part_s) CSS_HIP_TRY(hipFree(ix->part_s));
    if (ix->part_i) CSS_HIP_TRY(hipFree(ix->part_i));
    ix->part_s = nullptr;
    ix->part_i = nullptr;
    ix->part_cap = 0;
    hipError_t e = hipMalloc((void**)&ix->part_s, ncap * sizeof(float));
    if (e != hipSuccess) return css::hip_fail(e, "hipMalloc(part_s)", __FILE__, __LINE__);
    e = hipMalloc((void**)&ix->part_i, ncap * sizeof(uint32_t));
    if (e != hipSuccess) return css::hip_fail(e, "hipMalloc(part_i)", __FILE__, __LINE__);
    ix->part_cap = ncap;
    return CSS_OK;
}

inline int host_f2key(float f) {
    int i;
    memcpy(&i, &f, 4);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}

// Queries [q0, q0+nqc) (nqc <= 16) against the whole index, results to D/I rows q0...
int search_chunk_small(css_index* ix, int q0, int nqc, int k, const SweepGeom& sg, float* D_dev, int64_t* I_dev,
                       hipStream_t st) {
    int* gthr = ix->gthr + q0;
    hipLaunchKernelGGL(k_fill_int, dim3(1), dim3(64), 0, st, gthr, nqc, host_f2key(-INFINITY));
    CSS_LAUNCH_CHECK();
    const float* qp = ix->qpad + (size_t)q0 * ix->dpad;
    int rc;
    if (nqc <= 1) rc = launch_scan_small_nq<1, false>(ix, qp, nqc, k, gthr, sg, st);
    else if (nqc <= 2) rc = launch_scan_small_nq<2, false>(ix, qp, nqc, k, gthr, sg, st);
    else if (nqc <= 8)  // (an NQ=4 instantiation spills under hipcc 7.2; 3..8 share NQ=8)
         rc = launch_scan_small_nq<8, false>(ix, qp, nqc, k, gthr, sg, st);
    else rc = launch_scan_small_nq<16, false>(ix, qp, nqc, k, gthr, sg, st);
    if (rc != CSS_OK) return rc;
    {
        ProfScope ps("knn_merge", st);
        if (ix->metric == CSS_METRIC_IP)
            hipLaunchKernelGGL(k_merge_final<CSS_METRIC_IP>, dim3(nqc), dim3(256), 0, st, ix->part_s, ix->part_i, sg.G,
                               k, gthr, ix->qnorm2 + q0, ix->id_base, D_dev + (size_t)q0 * k,
                               I_dev + (size_t)q0 * k, 0, (float*)nullptr, (uint32_t*)nullptr, (int*)nullptr);
        else
            hipLaunchKernelGGL(k_merge_final<CSS_METRIC_L2>, dim3(nqc), dim3(256), 0, st, ix->part_s, ix->part_i, sg.G,
                               k, gthr, ix->qnorm2 + q0, ix->id_base, D_dev + (size_t)q0 * k,
                               I_dev + (size_t)q0 * k, 0, (float*)nullptr, (uint32_t*)nullptr, (int*)nullptr);
        CSS_LAUNCH_CHECK();
    }
    return CSS_OK;
}

// Device-side exact fix-up of the queries a candidate path flagged (overflowing buffer or band, band not closed):
// one launch that returns at once when nothing is flagged (nflag[1]: its blocks-done counter, zeroed with the count).  All pointers are already offset to the chunk's
// first query; the flagging kernel has reset gthr / fix lists / locks of every flagged query.
int launch_fixup(css_index* ix, const float* qpad, int nq, int k, int* gthr, const int* flag_list, const int* nflag,
                 float* fix_s, uint32_t* fix_i, int* fix_lock, float* D_dev, int64_t* I_dev, const SweepGeom& sg,
                 hipStream_t st) {
    int rc;
    if (sg.nq_sweep >= 8) rc = launch_scan_small_nq<8, true>(ix, qpad, 8, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev);
    else if (sg.nq_sweep >= 2) rc = launch_scan_small_nq<2, true>(ix, qpad, 2, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev);
    else rc = launch_scan_small_nq<1, true>(ix, qpad, 1, k, gthr, sg, st, flag_list, nflag, fix_s, fix_i, fix_lock, D_dev, I_dev);
    return rc;
}

// workspaces shared by the candidate paths: thresholds, candidate buffers, flags, fix-up lists for nq_pad queries
int grow_candidate_ws(css_index* ix, size_t nq_pad, int k) {
    int rc;
    if ((rc = grow(&ix->cthr, &ix->cthr_cap, nq_pad)) != CSS_OK) return rc;
    if ((rc = grow(&ix->cand_n, &ix->cand_n_cap, (size_t)nq_pad * CZ_NS)) != CSS_OK) return rc;
    if ((rc = grow(&ix->rs_work, &ix->rs_work_cap, 1 + (size_t)nq_pad * CZ_PARTS)) != CSS_OK) return rc;
    ix->last_nflag = nullptr;
    ix->last_nswept = nullptr;
    if ((rc = grow(&ix->cflags, &ix->cflags_cap, 2 * nq_pad + 2)) != CSS_OK) return rc;
    if ((rc = grow(&ix->cand_s, &ix->cand_s_cap, nq_pad * CZ_CAP)) != CSS_OK) return rc;
    if ((rc = grow(&ix->cand_i, &ix->cand_i_cap, nq_pad * CZ_CAP)) != CSS_OK) return rc;
    if ((rc = grow(&ix->fix_lock, &ix->fix_lock_cap, nq_pad)) != CSS_OK) return rc;
    const size_t need = nq_pad * (size_t)k;
    if (need > ix->fix_cap) {
        if (ix->fix_s) CSS_HIP_TRY(hipFree(ix->fix_s));
        if (ix->fix_i) CSS_HIP_TRY(hipFree(ix->fix_i));
        ix->fix_s = nullptr;
        ix->fix_i = nullptr;
        ix->fix_cap = 0;
        CSS_HIP_TRY(hipMalloc((void**)&ix->fix_s, need * sizeof(float)));
        CSS_HIP_TRY(hipMalloc((void**)&ix->fix_i, need * sizeof(uint32_t)));
        ix->fix_cap = need;
    }
    return CSS_OK;
}

// second coarse pass over flagged queries: slots (query rows) and candidates per slot
constexpr int kF2Max = 1024;
constexpr int CZ_CAP2 = 32768;
// largest k the MFMA kernels' LDS lists hold next to their staging buffers
constexpr int kMfmaMaxK = 64;
// extra ranks the split-operand candidate scan keeps beyond k (the band must close inside them, else the
// query is flagged and re-run exactly)
constexpr int kSplitExtra = 4;
// |split score - x.q| <= kSplitEps ||q|| max||x||: operand residuals 2 x 2^-16, the dropped l.l term 2^-16,
// 144 fp32 accumulation steps 2^-16.8 -- together < 3.6 x 2^-16; 2^-14 leaves a margin
constexpr float kSplitEps = 6.103515625e-05f;

// Exact fp32 batched scan (CSS_SEARCH_EXACT_FP32 with more than 16 queries, CSS_KNN_BATCH=fp32):
// v_mfma_f32_32x32x2_f32, bit-exact fmaf chains, scores written as they are.
template <int METRIC>
int launch_scan_fp32mfma(css_index* ix, int nq, int k, float* D_dev, int64_t* I_dev, hipStream_t st) {
    const int nq_pad = (nq + MF_BN - 1) / MF_BN * MF_BN;  // <= nq + 127 (qpad / gthr have 256 rows of slack)
    const int nqtiles = nq_pad / MF_BN;
    const int64_t ntiles = (ix->ntotal + MF_BM - 1) / MF_BM;
    const size_t lds = (size_t)(2 * MF_BM * MF_BK + 2 * MF_BN * MF_BK + 4 * 32 * 33 + MF_BM) * 4 + (size_t)MF_BN * k * 8;
    // strips in multiples of 8 for the XCD-aware block decode
    int nstrips = std::max(8, ix->num_cus / nqtiles / 8 * 8);
    nstrips = (int)std::min<int64_t>(nstrips, (ntiles + 7) / 8 * 8);
    const int64_t tps = (ntiles + nstrips - 1) / nstrips;
    int rc;
    if ((rc = grow_part(ix, (size_t)nq * nstrips * k)) != CSS_OK) return rc;
    if (nq_pad > nq)
        CSS_HIP_TRY(hipMemsetAsync(ix->qpad + (size_t)nq * ix->dpad, 0, (size_t)(nq_pad - nq) * ix->dpad * 4, st));
    hipLaunchKernelGGL(k_fill_int, dim3((nq_pad + 255) / 256), dim3(256), 0, st, ix->gthr, nq_pad, host_f2key(-INFINITY));
    CSS_LAUNCH_CHECK();
    auto kern = k_scan_mfma<METRIC>;
    if ((rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    // sibling pacing (see the kernel) where a strip has siblings; the counters share the cascade's pacing words
    int* pace = nullptr;
    if (knn_env().pacing && nqtiles > 1 && nstrips * nqtiles <= ix->num_cus) {   // (all blocks resident: one per CU)
        if ((rc = grow(&ix->cpace, &ix->cpace_cap, (size_t)nstrips * nqtiles)) != CSS_OK) return rc;
        pace = ix->cpace;
        CSS_HIP_TRY(hipMemsetAsync(pace, 0, (size_t)nstrips * nqtiles * sizeof(int), st));
    }
    {
        ProfScope ps("knn_scan_mfma", st);
        hipLaunchKernelGGL(kern, dim3(nstrips * nqtiles), dim3(256), lds, st, ix->xb, ix->xnorm2, ix->qpad, nq,
                           ix->ntotal, ix->dpad, k, nstrips, nqtiles, tps, ix->gthr, ix->part_s, ix->part_i, ix->cur_mask, pace);
        CSS_LAUNCH_CHECK();
    }
    {
        ProfScope ps("knn_merge", st);
        hipLaunchKernelGGL(k_merge_final<METRIC>, dim3(nq), dim3(256), 0, st, ix->part_s, ix->part_i, nstrips, k,
                           ix->gthr, ix->qnorm2, ix->id_base, D_dev, I_dev, METRIC == CSS_METRIC_L2 ? 1 : 0,
                           (float*)nullptr, (uint32_t*)nullptr, (int*)nullptr);
        CSS_LAUNCH_CHECK();
    }
    return CSS_OK;
}

// Batched candidate path of an index WITHOUT bf16 shadow rows (shards beyond ~38 M rows of 768 floats, or
// css_index_set_shadow(0)): the coarse scores are split-operand products formed from the fp32 rows
// (k_scan_mfma_split: h.h + h.l + l.h, error <= kSplitEps ||q|| max||x||), the scan keeps k + kSplitExtra
// ranks per query, and k_coarse_select<FINAL> rescores the band in fp32 exactly as the bf16 cascade does.  A
// query whose band does not close inside the kept ranks is flagged and re-run by the device-side fix-up.
// After the last stage of a candidate scan: band cut + flagging (one block per query), exact rescoring of the bands
// (CZ_PARTS work items per query over a fixed grid), sort by exact score + output (one block per query).
constexpr int kRescoreGrid = 4096;
struct EpsSet {   // what cz_eps needs to know about the operands a scan read
    float eps_rel;        // a-priori bound relative to ||q|| max||x||
    const float* qerr2;   // per query ||q - q^||^2 (null: fp32 queries)
    int measured;         // 0: a-priori only; else the word of maxn2 with the rows' measured error (1 bf16, 2 int8)
};
int launch_final_select(css_index* ix, int nq, int k, EpsSet e1, EpsSet e2, bool exact_k, int l2, int closed_n,
                        const float* qpad, const float* qnorm2, int* gthr, int* flags, int* nflag, int* flag_list, float* D_dev, int64_t* I_dev, float* thr2,
                        unsigned short* qh2, int f2, hipStream_t st) {
    // e1: the scan that filled the buffers; e2: the second pass over flagged queries (always bf16 rows and queries)
    const float eps_rel = e1.eps_rel;
    const float* qerr2 = e1.qerr2;
    const int measured = e1.measured;
    hipLaunchKernelGGL(k_coarse_select<true>, dim3(nq), dim3(256), 0, st, ix->cand_s, ix->cand_i, ix->cand_n, ix->cthr, flags,
                       nflag, flag_list, qnorm2, ix->maxn2, eps_rel, l2, k, closed_n, gthr, qerr2, measured, ix->fix_s, ix->fix_i, ix->fix_lock,
                       exact_k ? qpad : (const float*)nullptr, exact_k ? (const float*)ix->xb : (const float*)nullptr, ix->dpad);
    // (a work list of the live parts pays from a few dozen queries on; a handful of queries launch all their parts)
    const bool plan = nq > 16;
    if (plan) hipLaunchKernelGGL(k_rescore_plan, dim3(1), dim3(1024), 0, st, (const int*)ix->cand_n, nq, ix->rs_work, ix->rs_work + 1);
    hipLaunchKernelGGL(k_rescore_parts<false>, dim3(std::min(kRescoreGrid, nq * CZ_PARTS)), dim3(256), 0, st, ix->cand_s,
                       ix->cand_i, ix->cand_n, CZ_CAP, nq, (const int*)nullptr, (const int*)nullptr, (const float*)nullptr, l2, qpad,
                       ix->xb, ix->dpad, plan ? (const int*)ix->rs_work : (const int*)nullptr,
                       plan ? (const int*)(ix->rs_work + 1) : (const int*)nullptr);
    hipLaunchKernelGGL(k_coarse_final, dim3(nq), dim3(256), 0, st, ix->cand_s, ix->cand_i, ix->cand_n, flags, qnorm2, ix->maxn2,
                       e2.eps_rel, l2, k, qpad, ix->dpad, ix->id_base, D_dev, I_dev, thr2, qh2, f2, e2.qerr2, e2.measured);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

template <int METRIC>
int launch_scan_split_rescore(css_index* ix, int q0, int nq, int k, float* D_dev, int64_t* I_dev, const SweepGeom& sg,
                              hipStream_t st) {
    // queries [q0, q0 + nq) of the padded query rows; nq <= 4096 and q0 a multiple of 256 (the caller chunks: the
    // candidate buffers are 32 KiB per query, and only the last chunk has padding rows to clear)
    const int kp = k + kSplitExtra;
    float* const qpad = ix->qpad + (size_t)q0 * ix->dpad;
    float* const qnorm2 = ix->qnorm2 + q0;
    int* const gthr = ix->gthr + q0;
    D_dev += (size_t)q0 * k;
    I_dev += (size_t)q0 * k;
    const bool big = nq > 128 && kp <= 14;  // 256x256 tiles (8 waves) for real batches
    const int BMs = big ? 256 : MF_BM, BNs = big ? 256 : MF_BN;
    const int nq_pad = (nq + BNs - 1) / BNs * BNs;  // <= nq + 255 (qpad / gthr have 256 rows of slack)
    const int nqtiles = nq_pad / BNs;
    const int64_t ntiles = (ix->ntotal + BMs - 1) / BMs;
    // staging + lists (the slow-path scratch borrows a staging buffer); <= 80 KiB means two blocks share a CU
    const size_t lds = (size_t)(2 * BMs * MF_BK + 2 * BNs * MF_BK + BMs + 4) * 4 + (size_t)BNs * kp * 8;
    const int bpc = (!big && lds <= 80 * 1024) ? 2 : 1;
    int nstrips = std::max(8, bpc * ix->num_cus / nqtiles / 8 * 8);
    nstrips = (int)std::min<int64_t>(nstrips, (ntiles + 7) / 8 * 8);
    const int64_t tps = (ntiles + nstrips - 1) / nstrips;
    int rc;
    if ((rc = grow_part(ix, (size_t)nq * nstrips * kp)) != CSS_OK) return rc;
    if ((rc = grow_candidate_ws(ix, (size_t)nq_pad, k)) != CSS_OK) return rc;
    if ((rc = grow(&ix->qsplit, &ix->qsplit_cap, (size_t)nq_pad * ix->dpad * 2)) != CSS_OK) return rc;
    int* flags = ix->cflags;
    int* flag_list = ix->cflags + nq_pad;
    int* nflag = ix->cflags + 2 * nq_pad;
    ix->last_nflag = nflag;
    ix->last_nswept = nflag;
    if (nq_pad > nq)
        CSS_HIP_TRY(hipMemsetAsync(qpad + (size_t)nq * ix->dpad, 0, (size_t)(nq_pad - nq) * ix->dpad * 4, st));
    hipLaunchKernelGGL(k_fill_int, dim3((nq_pad + 255) / 256), dim3(256), 0, st, gthr, nq_pad, host_f2key(-INFINITY));
    hipLaunchKernelGGL(k_coarse_init, dim3((nq_pad + 255) / 256), dim3(256), 0, st, ix->cthr, ix->cand_n, flags, nflag,
                       nq, nq_pad, 0, (int*)nullptr, 0, (int*)nullptr, (float*)nullptr, (int*)nullptr, 0, (int*)nullptr,
                       (const float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 0, 0, 0);
    const int64_t ne = (int64_t)nq_pad * ix->dpad;
    hipLaunchKernelGGL(k_split_queries, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, qpad, ix->qsplit,
                       (int64_t)nq_pad, ix->dpad);
    CSS_LAUNCH_CHECK();
    ProfScope all("knn_split_cascade", st);
    {
        ProfScope ps("knn_scan_split", st);
        if (big) {
            auto kern = k_scan_mfma_split<METRIC, 8, 8>;
            if ((rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(nstrips * nqtiles), dim3(512), lds, st, ix->xb, ix->xnorm2, ix->qsplit, nq,
                               ix->ntotal, ix->dpad, kp, nstrips, nqtiles, tps, gthr, ix->part_s, ix->part_i, ix->cur_mask);
        } else {
            auto kern = k_scan_mfma_split<METRIC, 4, 4>;
            if ((rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
            hipLaunchKernelGGL(kern, dim3(nstrips * nqtiles), dim3(256), lds, st, ix->xb, ix->xnorm2, ix->qsplit, nq,
                               ix->ntotal, ix->dpad, kp, nstrips, nqtiles, tps, gthr, ix->part_s, ix->part_i, ix->cur_mask);
        }
        CSS_LAUNCH_CHECK();
    }
    // the kp best split scores of every query -> its candidate buffer (scores stay in the scan's form: IP dot
    // products, L2 2 x.q - ||x||^2, the form k_coarse_select expects of coarse scores)
    hipLaunchKernelGGL(k_merge_final<METRIC>, dim3(nq), dim3(256), 0, st, ix->part_s, ix->part_i, nstrips, kp, gthr,
                       qnorm2, ix->id_base, D_dev, I_dev, METRIC == CSS_METRIC_L2 ? 1 : 0, ix->cand_s, ix->cand_i,
                       ix->cand_n);
    if ((rc = launch_final_select(ix, nq, k, EpsSet{kSplitEps, nullptr, 0}, EpsSet{kSplitEps, nullptr, 0}, false, METRIC == CSS_METRIC_L2 ? 1 : 0, kp, qpad, qnorm2, gthr, flags, nflag,
                                  flag_list, D_dev, I_dev, nullptr, nullptr, 0, st)) != CSS_OK)
        return rc;
    return launch_fixup(ix, qpad, nq, k, gthr, flag_list, nflag, ix->fix_s, ix->fix_i, ix->fix_lock, D_dev, I_dev,
                        sg, st);
}

// Coarse bf16 scan + exact rescoring (css_knn_coarse.h) for queries [q0, q0 + nq) of ix->qpad; nq <= 4096.
template <int NQ, int TT, bool MAIN>
int launch_sweep_coarse_t(css_index* ix, const float* qpad, int nq, int64_t count, int64_t stride, int gm1, bool stage0,
                          hipStream_t st) {
    const size_t lds = (size_t)NQ * ix->dpad * sizeof(float);
    const int grid = (int)std::min<int64_t>((int64_t)ix->num_cus * 8, count);
    auto kern = k_sweep_coarse<NQ, TT, MAIN>;
    int rc;
    if (lds > 48 * 1024 && (rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, ix->xh, qpad, ix->cthr, ix->cand_s, ix->cand_i, ix->cand_n,
                       ix->ntotal, ix->dpad, nq, count, stride, gm1, stage0 ? 1 : 0, ix->cur_mask,
                       ix->metric == CSS_METRIC_L2 ? ix->xnorm2 : nullptr);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

template <int NQ, int TT, bool MAIN>
int launch_sweep_coarse_i8_t(css_index* ix, const float* qpad, int nq, int64_t count, int64_t stride, int gm1, bool stage0,
                             hipStream_t st) {
    const int steps = TT > 0 ? TT : (ix->dpad / 16 + 15) / 16;
    const size_t lds = ((size_t)NQ * 256 * steps + NQ) * sizeof(float);
    const int grid = (int)std::min<int64_t>((int64_t)ix->num_cus * 8, count);
    auto kern = k_sweep_coarse_i8<NQ, TT, MAIN>;
    int rc;
    if (lds > 48 * 1024 && (rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, ix->x8, ix->x8s, qpad, ix->cthr, ix->cand_s, ix->cand_i,
                       ix->cand_n, ix->ntotal, ix->dpad, nq, count, stride, gm1, stage0 ? 1 : 0, ix->cur_mask,
                       ix->metric == CSS_METRIC_L2 ? ix->xnorm2 : nullptr);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

// (the int8 rows exist only next to bf16 shadow rows; a RowView of a shadow-less index never gets here)
// Rows whose int8 copy is poor (a few dominant elements set the row scale and the rest rounds to nothing) make the
// measured band so wide that the buffers overflow and the queries end in the fix-up: results stay exact, the search
// gets slow.  So the use of the int8 rows adapts per index: `permille` = flagged share of the last int8 search above
// which the next 16 searches of that kind read the bf16 rows, then int8 is tried again.  (Caller holds ws_mu.)
inline bool i8_feedback_allows(css_index::I8Feedback& f, int permille) {
    if (f.pending && hipEventQuery(f.ev) == hipSuccess) {
        f.pending = false;
        if ((int64_t)*f.h_nflag * 1000 > (int64_t)f.nq * permille) f.backoff = 16;
    }
    if (f.backoff > 0) {
        --f.backoff;
        return false;
    }
    return true;
}
inline int i8_feedback_record(css_index::I8Feedback& f, const int* nflag_dev, int nq, hipStream_t st) {
    if (f.h_nflag == nullptr) {
        CSS_HIP_TRY(hipHostMalloc((void**)&f.h_nflag, sizeof(int), hipHostMallocDefault));
        CSS_HIP_TRY(hipEventCreateWithFlags(&f.ev, hipEventDisableTiming));
    }
    if (!f.pending) {
        CSS_HIP_TRY(hipMemcpyAsync(f.h_nflag, nflag_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        CSS_HIP_TRY(hipEventRecord(f.ev, st));
        f.pending = true;
        f.nq = nq;
    }
    return CSS_OK;
}
// 1..4 queries: any flagged query sends the next searches back to the bf16 sweep
inline bool sweep_uses_i8(css_index* ix) {
    if (ix->x8 == nullptr || !knn_env().sweep_i8) return false;
    return i8_feedback_allows(ix->fb_sweep, 0);
}
// batches: the int8 MFMA scan -- inner product, rows a whole (even) number of 128-B K steps, the 8-phase loop's shape --
// where it pays: its candidate band is wider than the bf16 scan's (with the one-eps thresholds of k_coarse_select ~150
// instead of ~25 band rows per query at 10 M rows, all rescored exactly, and the k best candidates are scored exactly
// in every select), costs per query that do not shrink with the index, while the saving is half of the scan.
// Measured (1000 queries, ms int8 / bf16, one session): k = 10: 0.1 M rows 0.40 / 0.32, 0.2 M 0.57 / 0.49, 0.3 M
// 0.64 / 0.68, 0.6 M 0.93 / 1.06, 1 M 1.25 / 1.61, 1.25 M 1.42 / 1.95, 2.5 M 2.35 / 3.5, 10 M 7.6-7.8 / 12.3-13.1;
// 10 M rows: k = 16 7.8 / 12.6, k = 32 8.4 / 12.8, k = 64 9.7 / 13.3, k = 100 10.8 / 13.7, k = 128 11.6 / 14.0; but
// 1 M rows, k = 100: 3.0 / 2.2, and 64 queries, k = 100, 10 M rows: 3.8 / 3.6 (the selects score k rows per query
// exactly); 4096 queries, k = 10: 29.0 / 52.3.  CSS_KNN_SCAN=i8 / bf16 force one or the other.
// Its wider band also flags more queries on clustered rows (10 M rows in 20 000 clusters: 23 % of the queries, 14.8 ms
// against the bf16 scan's 13.3 with none flagged), so the choice adapts per index: when an int8 batch flagged more than
// 5 % of its queries the next 16 batches read the bf16 rows, then int8 is tried again.  (Caller holds ws_mu.)
// (`rows`: the rows one cascade covers -- the index, or one range of a shadow-less index)
inline bool batch_i8_wanted(css_index* ix, int k, int64_t rows, int64_t nq) {
    const KnnEnv& e = knn_env();
    if (e.batch_i8 == 0 || ix->metric != CSS_METRIC_IP || ix->dpad % 256 != 0 || ix->dpad > 1024 || !e.loop8 || e.mfma_shape != 16)
        return false;
    if (e.batch_i8 == 2) return true;
    // (round 4, later stages on k_scan_qreg_i8, ms int8 / bf16: k = 100, 1000 queries: 4 M rows 4.1 / 6.1, 2 M 2.7 / 3.3, 1 M 2.0 /
    // 2.0, 300 k 1.3 / 0.9; 256 queries: 2 M 1.0 / 1.4, 1 M 0.87 / 0.75.  k = 10: 1 M 1.04 / 1.57, 300 k 0.53 / 0.64 (256 queries
    // 0.31 / 0.31), 100 k 0.35 / 0.31)
    // (fewer than 256 queries, k = 100, ms int8 / bf16, 33 / 64 / 128 / 255 queries: 10 M rows 1.89 / 3.19, 2.08 / 3.33, 2.19 /
    // 3.45, 2.50 / 3.87; 2 M rows 0.73 / 0.82 .. 0.99 / 1.08: the query count is no condition any more)
    (void)nq;
    const bool pays = k <= 32 ? rows >= 300000 : (k * 2 <= CZ_EXK && rows >= 2000000);
    if (!pays) return false;
    // (with <= 256 flagged queries the second pass is one query tile of bf16 scan -- a quarter of a 1000-query bf16 step --
    // and the int8 search still wins: 10 M rows in 20 000 clusters, 33 flagged: 12.8 ms against 13.3 on bf16 rows)
    return i8_feedback_allows(ix->fb_batch, 50);
}

template <int NQ>
int launch_sweep_coarse_nq(css_index* ix, const float* qpad, int nq, int64_t count, int64_t stride, int gm1, bool stage0,
                           hipStream_t st, bool i8) {
    const bool main_stage = stride == 1 && !stage0;
    if (i8) {
        if (ix->dpad == 768)
            return main_stage ? launch_sweep_coarse_i8_t<NQ, 3, true>(ix, qpad, nq, count, stride, gm1, stage0, st)
                              : launch_sweep_coarse_i8_t<NQ, 3, false>(ix, qpad, nq, count, stride, gm1, stage0, st);
        return launch_sweep_coarse_i8_t<NQ, 0, false>(ix, qpad, nq, count, stride, gm1, stage0, st);
    }
    if (ix->dpad == 768)
        return main_stage ? launch_sweep_coarse_t<NQ, 6, true>(ix, qpad, nq, count, stride, gm1, stage0, st)
                          : launch_sweep_coarse_t<NQ, 6, false>(ix, qpad, nq, count, stride, gm1, stage0, st);
    return launch_sweep_coarse_t<NQ, 0, false>(ix, qpad, nq, count, stride, gm1, stage0, st);
}

// 3..32 queries, inner product, int8 rows in view: does the sweep on the int8 MFMA (k_sweep_mfma_i8) answer sooner than what
// it replaces -- the VALU sweep (3, 4 queries) or the 256-query tiles of the batch scan?  tools/knn_fewq_probe.py, one
// session, ms with / without, 3 .. 16 queries: 10 M rows k = 10 1.65 / 1.92-2.07, k = 100 1.83-1.85 / 2.75-3.26; 1 M rows
// 0.32-0.33 / 0.34-0.41 and 0.46-0.48 / 0.45-0.58; 100 k rows 0.13-0.14 / 0.15-0.16 but 0.22 / 0.18-0.20 at k = 100 (a
// select with 100 exactly scored rows behind every stage); 20 k rows 0.11-0.12 / 0.11 and 0.17-0.19 / 0.13-0.15.
inline bool mfma_sweep_applies(const css_index* ix, int64_t nq, int k) {
    const int mode = knn_env().sweep_mfma;
    if (mode == 0 || ix->x8 == nullptr || ix->metric != CSS_METRIC_IP || ix->dpad > 1024 || nq < 3 || nq > 32) return false;
    if (mode == 2) return true;
    // 17..32 queries (two fragment sets per lane), ms with / without: k = 10: 10 M rows 1.78 / 1.66 (the batch scan with the
    // queries in registers is ahead there), 1 M rows 0.33 / 0.38, 100 k rows 0.13 / 0.155; k = 100: 10 M rows 2.0 / 3.2-3.3
    // (the bf16 scan: fewer than 256 queries), 1 M rows 0.56-0.59 / 0.53
    // (k = 100 at 10 M rows was measured against the bf16 scan; the int8 batch scan, which such a search takes since, is at
    // 1.89 ms for 33 queries: the sweep stays with k <= 32)
    if (nq > 16) return k <= 32 && ix->ntotal >= 50000 && ix->ntotal < 4000000;
    return ix->ntotal >= (k <= 32 ? 50000 : 1000000);
}

// one cascade stage of the int8 MFMA sweep (3..32 queries: k_sweep_mfma_i8); the int8 queries sit in ix->qh
int launch_sweep_mfma(css_index* ix, int nq, int64_t count, int64_t stride, int gm1, bool stage0, hipStream_t st) {
    const int grid = (int)std::min<int64_t>((int64_t)ix->num_cus * 8, count);
    const bool main_stage = stride == 1 && !stage0;
    const signed char* q8 = reinterpret_cast<const signed char*>(ix->qh);
#define CSS_LAUNCH_SWEEP_MFMA(KS_, MAIN_, NG_)                                                                         \
    hipLaunchKernelGGL((k_sweep_mfma_i8<KS_, MAIN_, NG_>), dim3(grid), dim3(256), 0, st, ix->x8, ix->x8s, q8, ix->qscale, ix->cthr,  \
                       ix->cand_s, ix->cand_i, ix->cand_n, ix->ntotal, ix->dpad, nq, count, stride, gm1, stage0 ? 1 : 0, \
                       ix->cur_mask)
    if (nq <= 16) {
        if (ix->dpad == 768) {
            if (main_stage) CSS_LAUNCH_SWEEP_MFMA(12, true, 1);
            else CSS_LAUNCH_SWEEP_MFMA(12, false, 1);
        } else {
            CSS_LAUNCH_SWEEP_MFMA(0, false, 1);
        }
    } else {   // 17..32 queries: two fragment sets per lane
        if (ix->dpad == 768) {
            if (main_stage) CSS_LAUNCH_SWEEP_MFMA(12, true, 2);
            else CSS_LAUNCH_SWEEP_MFMA(12, false, 2);
        } else {
            CSS_LAUNCH_SWEEP_MFMA(0, false, 2);
        }
    }
#undef CSS_LAUNCH_SWEEP_MFMA
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

// one later stage of the int8 batch scan with the queries in registers (k_scan_qreg_i8): two 4-wave blocks per CU
inline int qreg_grid(const css_index* ix) { return std::max(8, ix->num_cus * 2 / 8 * 8); }
template <int KS>
int launch_scan_qreg_t(css_index* ix, int nqt, int64_t count, int64_t stride, int gm1, bool main_stage, hipStream_t st) {
    const size_t lds = qr_lds_bytes<KS>();
    auto kern = main_stage ? k_scan_qreg_i8<KS, true> : k_scan_qreg_i8<KS, false>;
    int rc;
    if ((rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(qreg_grid(ix)), dim3(256), lds, st, (const unsigned char*)ix->x8,
                       reinterpret_cast<const signed char*>(ix->qh), (const float*)ix->cthr, ix->cand_s, ix->cand_i, ix->cand_n,
                       ix->ntotal, nqt, count, stride, gm1, ix->cur_mask, (const float*)ix->x8s, (const float*)ix->qscale);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}
inline bool qreg_applies(const css_index* ix, int nqt) {
    return knn_env().qreg && (ix->dpad == 256 || ix->dpad == 512 || ix->dpad == 768) && (qreg_grid(ix) / 8) / nqt >= 1;
}
int launch_scan_qreg(css_index* ix, int nqt, int64_t count, int64_t stride, int gm1, bool main_stage, hipStream_t st) {
    if (ix->dpad == 768) return launch_scan_qreg_t<12>(ix, nqt, count, stride, gm1, main_stage, st);
    if (ix->dpad == 512) return launch_scan_qreg_t<8>(ix, nqt, count, stride, gm1, main_stage, st);
    return launch_scan_qreg_t<4>(ix, nqt, count, stride, gm1, main_stage, st);
}

// the whole sweep cascade in one launch (k_sweep_cascade); sc: the schedule as tickets
template <int NQ, int TT, bool I8>
int launch_sweep_cascade_t(css_index* ix, const float* qpad, int nq, const FsSched& sc, int* flags, const float* qnorm2,
                           float eps_rel, int l2, int k, int measured, hipStream_t st) {
    const int steps = I8 ? (TT > 0 ? TT : (ix->dpad / 16 + 15) / 16) : 0;
    const size_t lds = (size_t)cz_fs_lds_floats(NQ, I8 ? 256 * steps : ix->dpad) * sizeof(float);
    auto kern = k_sweep_cascade<NQ, TT, I8>;
    int rc;
    if (lds > 48 * 1024 && (rc = css::ensure_dynamic_lds((const void*)kern, lds, ix->device)) != CSS_OK) return rc;
    // as many blocks as the chip holds at once (a wave leaves only when the tickets run out: more blocks would start at
    // the very end, load the queries and find nothing to do); nothing depends on the blocks being co-resident
    static int per_cu_of[64];   // (asked once per device and instantiation: the answer depends on nothing else here)
    int& per_cu_cached = per_cu_of[ix->device & 63];
    if (per_cu_cached == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_cached, kern, 256, lds) != hipSuccess || per_cu_cached < 1))
        per_cu_cached = 1;
    int per_cu = per_cu_cached;
    per_cu = std::min(per_cu, 3);   // (12 waves per CU already draw the whole HBM rate: 2 / 3 / 4 blocks 1.305 / 1.302 / 1.316 ms at 10 M rows)
    if (const int cap = knn_env().fs_blocks) per_cu = std::min(per_cu, cap);
    const int grid = (int)std::min<int64_t>((int64_t)ix->num_cus * std::min(per_cu, 8), (sc.first[sc.nstage] + 3) / 4);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, I8 ? (const void*)ix->x8 : (const void*)ix->xh,
                       I8 ? (const float*)ix->x8s : (const float*)nullptr, qpad, ix->cand_s, ix->cand_i, ix->cand_n, ix->cthr,
                       flags, ix->ntotal, ix->dpad, nq, sc, ix->fs_state, ix->cur_mask,
                       ix->metric == CSS_METRIC_L2 ? ix->xnorm2 : nullptr, qnorm2, (const int*)ix->maxn2, eps_rel, l2, k, measured,
                       knn_env().fs_spins);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}
template <int NQ>
int launch_sweep_cascade_nq(css_index* ix, const float* qpad, int nq, const FsSched& sc, int* flags, const float* qnorm2,
                            float eps_rel, int l2, int k, int measured, hipStream_t st, bool i8) {
    if (i8)
        return ix->dpad == 768 ? launch_sweep_cascade_t<NQ, 3, true>(ix, qpad, nq, sc, flags, qnorm2, eps_rel, l2, k, measured, st)
                               : launch_sweep_cascade_t<NQ, 0, true>(ix, qpad, nq, sc, flags, qnorm2, eps_rel, l2, k, measured, st);
    return ix->dpad == 768 ? launch_sweep_cascade_t<NQ, 6, false>(ix, qpad, nq, sc, flags, qnorm2, eps_rel, l2, k, measured, st)
                           : launch_sweep_cascade_t<NQ, 0, false>(ix, qpad, nq, sc, flags, qnorm2, eps_rel, l2, k, measured, st);
}

// grid of the persistent k_scan_coarse launches and the largest query chunk it can serve: the nqt blocks
// that share a row tile must fit the grid / 8 blocks of one XCD group (a CPX partition has few CUs)
inline int coarse_grid(const css_index* ix) { return std::max(8, ix->num_cus / 8 * 8); }
inline int coarse_max_chunk(const css_index* ix) { return std::min(4096, coarse_grid(ix) / 8 * CZ_T); }

// sweep = true: 1..4 queries through the HBM-bound bf16 sweep (k_sweep_coarse) instead of the MFMA scan.
// Everything is enqueued on `st`; nothing waits for the device (flagged queries are fixed up on the device).
// use_i8: which shadow rows this SEARCH reads, decided once by the caller (search_dev_enqueue / search_noshadow_ranges:
// the per-index feedback is consulted once per search, not per chunk); record_fb: this is the search's last chunk, whose
// flagged count is what the feedback sees.
// q_raw != null (sweep only): the raw query rows, still to be prepared (normalize_q as in search_dev_enqueue) -- the init
// launch of the cascade does it.
int launch_scan_coarse(css_index* ix, int q0, int nq, int k, float* D_dev, int64_t* I_dev, hipStream_t st,
                       const SweepGeom& sg, bool sweep, bool use_i8, bool record_fb, const float* q_raw = nullptr,
                       int normalize_q = 0) {
    const KnnEnv& env = knn_env();
    const float* qpad = ix->qpad + (size_t)q0 * ix->dpad;
    const float* qnorm2 = ix->qnorm2 + q0;
    D_dev += (size_t)q0 * k;
    I_dev += (size_t)q0 * k;
    // 3..32 queries on int8 rows (inner product): the sweep on the int8 MFMA (k_sweep_mfma_i8): int8 queries too
    const bool sweep_mfma = sweep && use_i8 && mfma_sweep_applies(ix, nq, k);
    const int nq_pad = sweep ? (sweep_mfma ? (nq <= 16 ? 16 : 32) : nq) : (nq + CZ_T - 1) / CZ_T * CZ_T;
    const int nqt = sweep ? 1 : nq_pad / CZ_T;
    // error of one coarse score relative to ||q|| max||x||: both operands bf16 (MFMA scan) or rows only (sweep)
    // int8 rows: a-priori |x^ - x| <= (s / 2) sqrt(d), s = max|x_i| / 127 <= ||x|| / 127 (the same for int8 queries)
    const bool i8 = use_i8;
    int rc_early = CSS_OK;
    const float i8_rel = sqrtf((float)ix->dpad) / 254.f;
    const float eps_rel = i8 ? ((sweep && !sweep_mfma) ? i8_rel : 2.f * i8_rel + i8_rel * i8_rel) + 0.00048828125f
                             : (sweep ? 0.00390625f + 0.00048828125f : 0.0078125f + 0.00048828125f);
    // ... tightened by the rounding errors actually measured at ingest / query prep (cz_eps); CSS_KNN_EPS=apriori for A/B.
    // measured = the word of maxn2 that holds the rows' error: 1 = bf16 rows, 2 = int8 rows
    const int measured = env.eps_measured ? (i8 ? 2 : 1) : 0;
    // (the int8 queries' error norms: sized HERE, before the pointer below is taken -- until round 4 the buffer grew further
    // down, so the selects of the first int8 search with more queries than any before read the freed, shorter one: zeros on
    // a fresh device, i.e. a band without the query term; stale bytes otherwise, i.e. everything flagged -- 744 of 1000
    // queries and 33 ms in the third index of one process, tools/seq_probe.py)
    if (i8 && (!sweep || sweep_mfma) && (rc_early = grow(&ix->qerr2_i8, &ix->qerr2_i8_cap, (size_t)q0 + nq_pad)) != CSS_OK) return rc_early;
    const float* qerr2 = (sweep && !sweep_mfma) ? nullptr : (i8 ? ix->qerr2_i8 + q0 : ix->qerr2 + q0);
    // the second pass over flagged queries reads the bf16 rows with bf16 queries
    const EpsSet eps_p2{0.0078125f + 0.00048828125f, ix->qerr2 + q0, env.eps_measured ? 1 : 0};
    const int l2 = ix->metric == CSS_METRIC_L2 ? 1 : 0;
    const float* xn2 = l2 ? ix->xnorm2 : nullptr;  // L2: coarse score = 2 x.q - ||x||^2
    int rc;
    if (!sweep && (rc = grow(&ix->qh, &ix->qh_cap, (size_t)nq_pad * ix->dpad)) != CSS_OK) return rc;
    if ((rc = grow_candidate_ws(ix, (size_t)nq_pad, k)) != CSS_OK) return rc;
    int* flags = ix->cflags;
    int* flag_list = ix->cflags + nq_pad;
    int* nflag = ix->cflags + 2 * nq_pad;
    ix->last_nflag = nflag;
    // Second pass (batches on indexes whose rows can overflow a 4096-slot buffer at all): flagged queries -- band or
    // buffer overflow: dense clusters, duplicate floods -- are scanned once more, together, against the threshold the
    // exact scores of their buffered candidates give (k_coarse_select<true>), into CZ_CAP2-slot buffers; only what
    // overflows those too goes on to the exact fp32 sweep.  Round 2 sent every flagged query to that sweep (8 queries
    // per pass over the fp32 rows): 22 flagged of 1000 queries cost 2.9 ms of a 7 ms batch at 1 M clustered rows.
    // (the second pass reads the bf16 rows: a shadow-less range scanned from int8 scratch rows sends its flagged queries
    // straight to the exact sweep -- and, through the feedback of batch_uses_i8, the next searches to bf16 ranges)
    const bool pass2 = !sweep && env.pass2 && env.loop8 && env.mfma_shape == 16 && ix->dpad % 128 == 0 && ix->ntotal > CZ_CAP &&
                       ix->xh != nullptr;
    const int f2 = pass2 ? std::min(nq_pad, kF2Max) : 0;   // (a multiple of CZ_T)
    int* flag_listB = nullptr;
    int* nflagB = nullptr;
    ix->last_nswept = nflag;
    if (pass2) {
        const size_t qh2_before = ix->qh2_cap;
        if ((rc = grow(&ix->qh2, &ix->qh2_cap, (size_t)f2 * ix->dpad)) != CSS_OK) return rc;
        // slots beyond the flagged count are scanned with a +inf threshold; their rows must still be finite numbers
        if (ix->qh2_cap != qh2_before) CSS_HIP_TRY(hipMemsetAsync(ix->qh2, 0, ix->qh2_cap * sizeof(unsigned short), st));
        if ((rc = grow(&ix->thr2, &ix->thr2_cap, (size_t)f2)) != CSS_OK) return rc;
        if ((rc = grow(&ix->cand_n2, &ix->cand_n2_cap, (size_t)f2 * CZ_NS)) != CSS_OK) return rc;
        if ((rc = grow(&ix->cand_s2, &ix->cand_s2_cap, (size_t)f2 * CZ_CAP2)) != CSS_OK) return rc;
        if ((rc = grow(&ix->cand_i2, &ix->cand_i2_cap, (size_t)f2 * CZ_CAP2)) != CSS_OK) return rc;
        if ((rc = grow(&ix->flagB, &ix->flagB_cap, (size_t)nq_pad + 2)) != CSS_OK) return rc;
        flag_listB = ix->flagB;
        nflagB = ix->flagB + nq_pad;
        ix->last_nswept = nflagB;
    }

    // cascade schedule: a nested, uniformly strided sample of row tiles.  Stage 0 reads every s-th tile (at most 15
    // tiles: it keeps every score, 3840 of the 4096 slots), every later stage the tiles at a stride `ratio` times
    // smaller that were not read before; the last stage has stride 1.
    // Growth factor g: every stage reads g-1 times the tiles read before it.  Batches (MFMA scan): g = 8 for k <= 32
    // (4 stages at 1.25 M rows instead of 6; the last stage is 7/8 of the rows and appends ~7k + band candidates per
    // query, cheap since the tile epilogue walks hits per lane: 10 M x 1000, k = 10: 12.25 ms vs 12.47 at g = 4, 100 k
    // rows 0.32 vs 0.34 ms; g = 16: 12.4 ms), g = 4 above (k = 100: 13.55 vs 13.79 ms; k = 32..64 equal).  Round 2's
    // epilogue made g = 8 2 % slower.  1..4 queries (sweep): g = 4.  Fewer, larger stages do not help
    // there -- measured at 10 M rows, k = 10: g = 4 / 8 / 16 (7 / 5 / 4 stages) all take 2.71-2.72 ms, the call is the
    // 15.36 GB of shadow rows at the sweep's bandwidth plus ~0.25 ms -- and with k = 100 (the reference's call shape)
    // g = 16 overflows the 4096-slot buffers (~k g candidates per stage) and lands in the exact fix-up: 10 ms.
    const int64_t ntiles = (ix->ntotal + CZ_T - 1) / CZ_T;
    // (int8 scan: its band is ~4 x wider, growth 8 would append ~2500 rows per query in the main stage)
    // (3..32 queries on the int8 MFMA: one launch per stage, so below ~4 M rows fewer, larger stages win -- ms at growth 4 / 8,
    // k = 10: 1 M rows 0.33 / 0.27, 100 k rows 0.13 / 0.10, 10 M rows 1.59 / 1.55-1.67; k = 100 overflows the buffers at
    // growth 8 and 10 M rows, as the VALU sweep did)
    const int g_sweep = env.growth_sweep ? env.growth_sweep : ((sweep_mfma && k <= 32 && ix->ntotal < 4000000) ? 8 : 4);
    const int g = sweep ? g_sweep : (env.growth ? env.growth : ((k <= 32 && !i8) ? 8 : 4));
    struct Stage {
        int64_t stride;
        int ratio;   // stride of the previous stage / this stride (stage 0: unused)
    };
    std::vector<Stage> sched;
    {
        int64_t s0 = 1;
        while (ntiles / (s0 * g) >= 2) s0 *= g;
#ifndef CZ_S0_FILL
#define CZ_S0_FILL 1
#endif
        // (2 .. 2g-1 tiles so far; stage 0 holds 15: where the next finer stride still fits, the cascade is one stage shorter --
        // batches 152.5 k -> 153.7 k queries/s, 4 / 16 queries on the int8 MFMA 1.67 / 1.72 -> 1.59 ms at 10 M rows; not for the
        // one-launch cascade of 1..2 queries, whose first select is one wave's work: 1.29 -> 1.32 ms)
        if (CZ_S0_FILL && (!sweep || sweep_mfma) && s0 >= g && (ntiles + s0 / g - 1) / (s0 / g) <= 15) s0 /= g;
        const int64_t w0 = (ntiles + s0 - 1) / s0;          // tiles at stride s0
        const int g0 = (int)((w0 + 14) / 15);                // > 1: one coarser first stage in front
        if (g0 > 1) sched.push_back({s0 * g0, 0});
        sched.push_back({s0, g0});
        for (int64_t s = s0 / g; s >= 1; s /= g) sched.push_back({s, g});
        // int8 scan (band ~4 x wider): the last step of 4 is taken as two steps of 2 -- the main stage is then half of
        // the rows under the threshold of the other half (~3 x fewer appends than 3/4 of the rows under a quarter's)
        if (i8 && (!sweep || sweep_mfma) && g == 4 && sched.size() >= 2 && sched.back().stride == 1 && ntiles >= 64) {
            sched.back() = {2, 2};
            sched.push_back({1, 2});
        }
    }
    const int64_t n0 = (ntiles + sched[0].stride - 1) / sched[0].stride;
    // 1..4 queries: the stages and the selects between them as ONE launch (k_sweep_cascade), quarter tiles as tickets in stage order
    // (measured, ms one launch / one per stage: 10 M rows, 1 query k = 10 1.35 / 1.41, k = 100 1.40 / 1.56, 2 queries 1.40 /
    // 1.47, 4 queries 2.68 / 2.69; 100 k rows: 0.085 / 0.095, 0.118 / 0.125, but 2 queries 0.117 / 0.111, 4: 0.186 / 0.155)
    bool fused = sweep && !sweep_mfma && env.sweep_fused && (env.sweep_fused == 2 || nq == 1 || ix->ntotal >= 1000000) &&
                 (int)sched.size() <= CZ_FS_MAXST && ntiles < (int64_t)1 << 28;
    FsSched fsched{};
    if (fused) {
        fsched.nstage = (int)sched.size();
        int64_t first = 0;
        for (size_t si = 0; si < sched.size(); ++si) {
            const int64_t W = (ntiles + sched[si].stride - 1) / sched[si].stride;
            const int gr = si == 0 ? g : sched[si].ratio;
            const int64_t count = si == 0 ? W : (W - 1) - (W - 1) / gr;
            if (count <= 0) fused = false;   // (a stage without tiles would leave nobody to run its select; not seen with these schedules)
            fsched.first[si] = (int)first;
            fsched.stride[si] = (int)sched[si].stride;
            fsched.gm1[si] = std::max(1, gr - 1);
            first += 4 * count;   // tickets are quarter tiles
        }
        fsched.first[sched.size()] = (int)first;
        if (fused && (rc = grow(&ix->fs_state, &ix->fs_state_cap, (size_t)CZ_FS_WORDS)) != CSS_OK) return rc;
    }
    constexpr int kPaceGroups = 512, kPaceStages = 20;
    const bool use_pace = !sweep && env.pacing;
    if (use_pace && (rc = grow(&ix->cpace, &ix->cpace_cap, (size_t)kPaceGroups * kPaceStages)) != CSS_OK) return rc;

    {
        if (!sweep && i8) {   // int8 query rows (into the same buffer: half its bytes), their scales and error norms
            if ((rc = grow(&ix->qscale, &ix->qscale_cap, (size_t)nq_pad)) != CSS_OK) return rc;
            if ((rc = grow(&ix->qerr2_i8, &ix->qerr2_i8_cap, (size_t)q0 + nq_pad)) != CSS_OK) return rc;
            hipLaunchKernelGGL(k_rows_to_i8, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, qpad,
                               reinterpret_cast<signed char*>(ix->qh), ix->qscale, ix->qerr2_i8 + q0, nq, nq_pad, ix->dpad);
            CSS_LAUNCH_CHECK();
        } else if (!sweep) {
            const int64_t ne = (int64_t)nq_pad * ix->dpad;
            hipLaunchKernelGGL(k_rows_to_bf16, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, qpad, ix->qh,
                               (int64_t)nq, (int64_t)nq_pad, ix->dpad);
            CSS_LAUNCH_CHECK();
        }
        const int npace = use_pace ? kPaceGroups * kPaceStages : 0;
        const int ninit = std::max(std::max(std::max(std::max(nq_pad, npace), f2), fused ? CZ_FS_KEYWORDS : 0),
                                   q_raw != nullptr ? nq * 64 : 0);   // (query prep: one wave per query)
        hipLaunchKernelGGL(k_coarse_init, dim3((ninit + 255) / 256), dim3(256), 0, st, ix->cthr, ix->cand_n, flags,
                           nflag, nq, nq_pad, (int)(n0 * CZ_T), use_pace ? ix->cpace : (int*)nullptr, npace,
                           pass2 ? ix->cand_n2 : (int*)nullptr, pass2 ? ix->thr2 : (float*)nullptr, nflagB, f2,
                           fused ? ix->fs_state : (int*)nullptr, q_raw, ix->qpad + (size_t)q0 * ix->dpad, ix->qnorm2 + q0,
                           ix->qerr2 + q0, ix->dim, ix->dpad, normalize_q);
        CSS_LAUNCH_CHECK();
        if (sweep_mfma) {   // int8 query rows (16, zero padded), their scales and error norms -- behind the init launch, which may have prepared qpad
            if ((rc = grow(&ix->qh, &ix->qh_cap, (size_t)nq_pad * ix->dpad)) != CSS_OK) return rc;
            if ((rc = grow(&ix->qscale, &ix->qscale_cap, (size_t)nq_pad)) != CSS_OK) return rc;
            if ((rc = grow(&ix->qerr2_i8, &ix->qerr2_i8_cap, (size_t)q0 + nq_pad)) != CSS_OK) return rc;
            hipLaunchKernelGGL(k_rows_to_i8, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, qpad,
                               reinterpret_cast<signed char*>(ix->qh), ix->qscale, ix->qerr2_i8 + q0, nq, nq_pad, ix->dpad);
            CSS_LAUNCH_CHECK();
        }
    }
    const size_t lds = (size_t)CZ_NST * CZ_STAGE;
    typedef void (*scan_fn)(const unsigned short*, const unsigned short*, const float*, float*, uint32_t*, int*, int64_t, int,
                            int, int64_t, int64_t, int, int*, const uint32_t*, const float*, int, const int*, const float*,
                            const float*);
    // (the DBG instantiations honour CSS_KNN_DBG; the product kernels carry no timing switches)
    // v_mfma_f32_16x16x32_bf16 by default: same cycles per flop and LDS traffic as 32x32x16, but the chip holds a
    // higher clock under it (measured in one session: main stage 11.1 ms vs 12.1 ms); CSS_KNN_MFMA=32 for A/B runs
    const bool m16 = env.mfma_shape == 16;
    // the 8-phase ping-pong loop (k_scan_coarse8) wherever its shape constraints hold; CSS_KNN_LOOP=old for A/B runs
    const bool loop8 = env.loop8 && m16 && ix->dpad % 128 == 0;
    const bool i8b = i8 && !sweep;   // (batch_uses_i8 implies the 8-phase loop)
    // one-eps thresholds from exactly scored top-k candidates (k_coarse_select): the int8 scan, whose band is wide
    // (CSS_KNN_EXACTK=0 for A/B runs; the scores are inner products: batch_uses_i8)
    const bool exact_k = (i8b || sweep_mfma) && env.exact_k && k * 2 <= CZ_EXK;   // (both operands int8: the widest band)
    const scan_fn f_stage0 = i8b ? k_scan_coarse8<true, false, false, CZ_CAP, true>
                                 : (loop8 ? k_scan_coarse8<true, false>
                                          : (m16 ? k_scan_coarse<true, false, false, 16> : k_scan_coarse<true, false>));
    const scan_fn f_mid = i8b ? (env.dbg ? k_scan_coarse8<false, false, true, CZ_CAP, true> : k_scan_coarse8<false, false, false, CZ_CAP, true>)
                              : (loop8 ? (env.dbg ? k_scan_coarse8<false, false, true> : k_scan_coarse8<false, false>)
                                       : (env.dbg ? k_scan_coarse<false, false, true>
                                                  : (m16 ? k_scan_coarse<false, false, false, 16> : k_scan_coarse<false, false>)));
    const scan_fn f_main = i8b ? (env.dbg ? k_scan_coarse8<false, true, true, CZ_CAP, true> : k_scan_coarse8<false, true, false, CZ_CAP, true>)
                               : (loop8 ? (env.dbg ? k_scan_coarse8<false, true, true> : k_scan_coarse8<false, true>)
                                        : (env.dbg ? k_scan_coarse<false, true, true>
                                                   : (m16 ? k_scan_coarse<false, true, false, 16> : k_scan_coarse<false, true>)));
    const unsigned short* scan_rows = i8b ? reinterpret_cast<const unsigned short*>(ix->x8) : ix->xh;
    const float* scan_xsc = i8b ? ix->x8s : nullptr;
    const float* scan_qsc = i8b ? ix->qscale : nullptr;
    if (!sweep)
        for (scan_fn f : {f_stage0, f_mid, f_main})
            if ((rc = css::ensure_dynamic_lds((const void*)f, lds, ix->device)) != CSS_OK) return rc;
    const int grid = coarse_grid(ix);
    CSS_REQUIRE(sweep || (grid / 8) / nqt >= 1, "css_index_search: %d query tiles do not fit a grid of %d blocks", nqt, grid);

    int stage_idx = 0;
    if (fused) {
        ProfScope all("knn_sweep_cascade", st);
        {
            ProfScope ps("knn_sweep_fused", st);
            if (nq <= 1) rc = launch_sweep_cascade_nq<1>(ix, qpad, nq, fsched, flags, qnorm2, eps_rel, l2, k, measured, st, i8);
            else if (nq <= 2) rc = launch_sweep_cascade_nq<2>(ix, qpad, nq, fsched, flags, qnorm2, eps_rel, l2, k, measured, st, i8);
            else rc = launch_sweep_cascade_nq<4>(ix, qpad, nq, fsched, flags, qnorm2, eps_rel, l2, k, measured, st, i8);
            if (rc != CSS_OK) return rc;
        }
        if ((rc = launch_final_select(ix, nq, k, EpsSet{eps_rel, qerr2, measured}, eps_p2, exact_k, l2, 0, qpad, qnorm2, ix->gthr + q0, flags, nflag, flag_list, D_dev,
                                      I_dev, pass2 ? ix->thr2 : nullptr, pass2 ? ix->qh2 : nullptr, f2, st)) != CSS_OK)
            return rc;
    } else {
    ProfScope all(sweep ? "knn_sweep_cascade" : "knn_coarse_cascade", st);
    for (size_t si = 0; si < sched.size(); ++si) {
        const int64_t s = sched[si].stride;
        const int gr = si == 0 ? g : sched[si].ratio;   // (stage 0 does not use the ratio)
        const bool stage0 = si == 0;
        const int64_t W = (ntiles + s - 1) / s;
        const int64_t count = stage0 ? W : (W - 1) - (W - 1) / gr;
        if (count > 0 && sweep_mfma) {
            ProfScope ps(s == 1 && !stage0 ? "knn_sweep_mfma_main" : "knn_sweep_mfma_stage", st);
            if ((rc = launch_sweep_mfma(ix, nq, count, s, std::max(1, gr - 1), stage0, st)) != CSS_OK) return rc;
        } else if (count > 0 && sweep) {
            ProfScope ps(s == 1 && !stage0 ? "knn_sweep_coarse_main" : "knn_sweep_coarse_stage", st);
            if (nq <= 1) rc = launch_sweep_coarse_nq<1>(ix, qpad, nq, count, s, gr - 1, stage0, st, i8);
            else if (nq <= 2) rc = launch_sweep_coarse_nq<2>(ix, qpad, nq, count, s, gr - 1, stage0, st, i8);
            else rc = launch_sweep_coarse_nq<4>(ix, qpad, nq, count, s, gr - 1, stage0, st, i8);
            if (rc != CSS_OK) return rc;
        } else if (count > 0) {
            const scan_fn f = stage0 ? f_stage0 : (s == 1 ? f_main : f_mid);
            ProfScope ps(s == 1 && !stage0 ? "knn_scan_coarse_main" : "knn_scan_coarse_stage", st);
            // int8 rows, later stages: the queries stay in registers (k_scan_qreg_i8) -- from two (row tile, query tile) pairs
            // per block on: a block first loads its 256 queries.  ms per search, k = 10, 64 / 256 / 1000 queries, minimum
            // 0 / 1024 / 4096 / never: 300 k rows 0.31, 0.35, 0.56 / 0.26, 0.31, 0.53 / 0.26, 0.31, 0.56 / 0.25, 0.31, 0.56;
            // 3 M rows (1024 / 4096 / never) 0.70, 0.79, 2.25 / 0.73, 0.84, 2.28 / 0.78, 0.90, 2.56; 10 M rows 1.88, 1.99,
            // 6.45 / 1.88, 2.01, 6.52 / 2.05, 2.27, 7.36.
            if (i8b && !stage0 && !env.dbg && qreg_applies(ix, nqt) && count * nqt >= env.qreg_min) {
                if ((rc = launch_scan_qreg(ix, nqt, count, s, gr - 1, s == 1, st)) != CSS_OK) return rc;
            } else {
                // (int8 rows: no sibling pacing -- a row tile fetched by every query-tile block on its own is still only
                // ~3.5 TB/s worst case at this scan's speed, and the coupling costs more than the HBM traffic it saves:
                // 9.65 vs 9.02 ms per batch; the bf16 scan reads twice the bytes per row and needs it)
                int* pace = (env.pacing && !i8b && grid / 8 <= kPaceGroups / 8 && stage_idx < kPaceStages) ? ix->cpace + (size_t)stage_idx * kPaceGroups : nullptr;
                hipLaunchKernelGGL(f, dim3(grid), dim3(512), lds, st, scan_rows, ix->qh, ix->cthr, ix->cand_s, ix->cand_i,
                                   ix->cand_n, ix->ntotal, ix->dpad, nqt, count, s, gr - 1, pace, ix->cur_mask, xn2, env.dbg,
                                   (const int*)nullptr, scan_xsc, scan_qsc);
                CSS_LAUNCH_CHECK();
            }
        }
        ++stage_idx;
        if (s == 1) {
            if ((rc = launch_final_select(ix, nq, k, EpsSet{eps_rel, qerr2, measured}, eps_p2, exact_k, l2, 0, qpad, qnorm2, ix->gthr + q0, flags, nflag, flag_list, D_dev,
                                          I_dev, pass2 ? ix->thr2 : nullptr, pass2 ? ix->qh2 : nullptr, f2, st)) != CSS_OK)
                return rc;
            break;
        }
        hipLaunchKernelGGL(k_coarse_select<false>, dim3(nq), dim3(256), 0, st, ix->cand_s, ix->cand_i, ix->cand_n,
                           ix->cthr, flags, nflag, flag_list, qnorm2, ix->maxn2, eps_rel, l2, k, 0, ix->gthr + q0, qerr2, measured, ix->fix_s,
                           ix->fix_i, ix->fix_lock, exact_k ? qpad : (const float*)nullptr,
                           exact_k ? (const float*)ix->xb : (const float*)nullptr, ix->dpad);
        CSS_LAUNCH_CHECK();
    }
    }
    // feedback for batch_i8_wanted / sweep_uses_i8: the last chunk of a search speaks for it
    if (record_fb && i8b && env.batch_i8 == 1 && (rc = i8_feedback_record(ix->fb_batch, nflag, nq, st)) != CSS_OK) return rc;
    if (record_fb && i8 && sweep && (rc = i8_feedback_record(ix->fb_sweep, nflag, nq, st)) != CSS_OK) return rc;
    if (pass2) {
        // every launch below reads the flagged count from device memory and returns at once when there is nothing to do
        ProfScope ps("knn_coarse_pass2", st);
        const scan_fn f_all = k_scan_coarse8<false, true, false, CZ_CAP2>;   // every row tile against thr2 (the gate also makes tile = ordinal)
        if ((rc = css::ensure_dynamic_lds((const void*)f_all, lds, ix->device)) != CSS_OK) return rc;
        const int nqt2 = f2 / CZ_T;
        {
            int* pace = (env.pacing && grid / 8 <= kPaceGroups / 8 && stage_idx < kPaceStages) ? ix->cpace + (size_t)stage_idx * kPaceGroups : nullptr;
            hipLaunchKernelGGL(f_all, dim3(grid), dim3(512), lds, st, ix->xh, ix->qh2, ix->thr2, ix->cand_s2, ix->cand_i2,
                               ix->cand_n2, ix->ntotal, ix->dpad, nqt2, ntiles, (int64_t)1, 1 << 30, pace, ix->cur_mask, xn2, 0,
                               (const int*)nflag, (const float*)nullptr, (const float*)nullptr);
        }
        hipLaunchKernelGGL(k_rescore_parts<true>, dim3(kRescoreGrid), dim3(256), 0, st, ix->cand_s2, ix->cand_i2, ix->cand_n2,
                           CZ_CAP2, f2, nflag, flag_list, ix->thr2, l2, qpad, ix->xb, ix->dpad, (const int*)nullptr, (const int*)nullptr);
        hipLaunchKernelGGL(k_coarse_select2<CZ_CAP2>, dim3(f2), dim3(256), 0, st, ix->cand_s2, ix->cand_i2, ix->cand_n2, nflag,
                           flag_list, f2, nflagB, flag_listB, l2, k, ix->id_base, D_dev, I_dev);
        CSS_LAUNCH_CHECK();
        // what overflowed the second pass too (tens of thousands of rows inside one band): exact fp32 sweep
        return launch_fixup(ix, qpad, nq, k, ix->gthr + q0, flag_listB, nflagB, ix->fix_s, ix->fix_i, ix->fix_lock, D_dev, I_dev,
                            sg, st);
    }
    // queries whose candidate buffer or band overflowed (thousands of duplicate rows, a zero query): exact
    // fp32 sweep on the device, two launches that return at once when the flag count is zero
    return launch_fixup(ix, qpad, nq, k, ix->gthr + q0, flag_list, nflag, ix->fix_s, ix->fix_i, ix->fix_lock, D_dev, I_dev,
                        sg, st);
}

// sweep geometry of the small-batch kernel (also the exact fix-up of the candidate paths) for the rows in view
int make_sweep_geom(const css_index* ix, int k, SweepGeom* sg) {
    sg->nq_sweep = (int)std::min<int64_t>(16, (64 * 1024) / ((int64_t)ix->dpad * 4 + (int64_t)k * 8 + 8));
    CSS_REQUIRE(sg->nq_sweep >= 1, "css_index_search: dim=%d too large for the scan kernel", ix->dim);
    // enough blocks to fill the chip (8 per CU) but at least ~64 row groups of work each
    const int64_t ngroups = (ix->ntotal + 3) / 4;
    int64_t G = std::max<int64_t>(1, std::min<int64_t>((int64_t)ix->num_cus * 8, (ngroups + 63) / 64));
    sg->gpb = (ngroups + G - 1) / G;
    sg->G = (int)((ngroups + sg->gpb - 1) / sg->gpb);
    return CSS_OK;
}

// RAII: the index narrowed to rows [row0, row0 + n) with `xh` as their bf16 shadow rows -- every launcher below reads
// rows, norms, count, id base and the allow-bitmap through the css_index fields, so a row range is searched exactly
// like an index of its own.  Caller holds ws_mu (export and css_index_ntotal do not look at these fields unguarded).
struct RowView {
    css_index* ix;
    float* xb;
    float* xnorm2;
    unsigned short* xh;
    unsigned char* x8;
    float* x8s;
    int64_t ntotal, id_base;
    const uint32_t* mask;
    // xh_rows / x8_rows + x8_scales: the range's bf16 OR int8 scratch rows (the other kind null)
    RowView(css_index* i, int64_t row0, int64_t n, unsigned short* xh_rows, unsigned char* x8_rows = nullptr,
            float* x8_scales = nullptr)
        : ix(i), xb(i->xb), xnorm2(i->xnorm2), xh(i->xh), x8(i->x8), x8s(i->x8s), ntotal(i->ntotal), id_base(i->id_base),
          mask(i->cur_mask) {
        ix->xb = xb + (size_t)row0 * ix->dpad;
        ix->xnorm2 = xnorm2 + row0;
        ix->xh = xh_rows;
        ix->x8 = x8_rows;
        ix->x8s = x8_scales;
        ix->ntotal = n;
        ix->id_base = id_base + row0;
        if (mask) ix->cur_mask = mask + row0 / 32;   // (row0 is a multiple of 256)
    }
    ~RowView() {
        ix->xb = xb;
        ix->xnorm2 = xnorm2;
        ix->xh = xh;
        ix->x8 = x8;
        ix->x8s = x8s;
        ix->ntotal = ntotal;
        ix->id_base = id_base;
        ix->cur_mask = mask;
    }
};

int merge_parts(const float* Dp, const int64_t* Ip, int nparts, int64_t stride_d, int64_t stride_i, int64_t nq, int k,
                int metric, float* D, int64_t* I, int device, void* stream, const char* who) {
    CSS_REQUIRE(Dp && Ip && D && I, "%s: NULL buffer", who);
    CSS_REQUIRE(nparts >= 1 && nq >= 0 && k >= 1 && k <= CSS_MAX_K, "%s: bad sizes", who);
    CSS_REQUIRE(metric == CSS_METRIC_IP || metric == CSS_METRIC_L2, "%s: unknown metric", who);
    int rc = css::check_device(device);
    if (rc != CSS_OK) return rc;
    if (nq == 0) return CSS_OK;
    DeviceGuard g(device);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps("knn_merge_parts", st);
    if (k > CSS_KERNEL_MAX_K) {
        if (metric == CSS_METRIC_IP)
            hipLaunchKernelGGL(k_merge_parts_rank<CSS_METRIC_IP>, dim3((unsigned)nq), dim3(256), 0, st, Dp, Ip, nparts, stride_d,
                               stride_i, nq, k, D, I);
        else
            hipLaunchKernelGGL(k_merge_parts_rank<CSS_METRIC_L2>, dim3((unsigned)nq), dim3(256), 0, st, Dp, Ip, nparts, stride_d,
                               stride_i, nq, k, D, I);
        CSS_LAUNCH_CHECK();
        return CSS_OK;
    }
    if (metric == CSS_METRIC_IP)
        hipLaunchKernelGGL(k_merge_parts<CSS_METRIC_IP>, dim3((unsigned)nq), dim3(64), 0, st, Dp, Ip, nparts, stride_d,
                           stride_i, nq, k, D, I);
    else
        hipLaunchKernelGGL(k_merge_parts<CSS_METRIC_L2>, dim3((unsigned)nq), dim3(64), 0, st, Dp, Ip, nparts, stride_d,
                           stride_i, nq, k, D, I);
    CSS_LAUNCH_CHECK();
    return CSS_OK;
}

// Batches on an index WITHOUT bf16 shadow rows (more than ~38 M rows of 768 floats on one 288 GB GPU, or
// css_index_set_shadow(ix, 0)): the rows are rounded to bf16 one range at a time into a scratch buffer sized from
// the free HBM, every range is searched by the SAME cascade as a shadowed index (same error band: the scratch
// rows are exactly the shadow rows it would have had), and the per-range top-k lists are merged.  Per 10 M rows:
// 46 GB of conversion traffic + the 13.6 ms cascade, against 45-48 ms for the split-operand scan it replaces (three
// MFMA products per score, 4.4 x its algorithmic bytes), and the cost of the conversion is shared by up to 4096
// queries.  Returns kNoRangeScratch without touching the outputs (nothing enqueued) when no scratch of at least 2^20 rows can be had.
// (internal, never crosses the C ABI: "no scratch memory for the row ranges -- take the fallback"; distinct from every
// css_status so that an error of an inner launch can never be mistaken for it)
constexpr int kNoRangeScratch = 1;
int search_noshadow_ranges(css_index* ix, int64_t nq, int k, float* D_dev, int64_t* I_dev, hipStream_t st, bool allow_i8) {
    int rc;
    const int64_t ntotal = ix->ntotal;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return kNoRangeScratch;
    const size_t row_b = (size_t)ix->dpad * 2;
    int64_t rows_fit = (int64_t)((free_b + ix->xh_tmp_cap * 2) / 2 / row_b);          // half of what is free (incl. our own scratch)
    rows_fit = std::min<int64_t>(rows_fit, 16ll << 20) / CZ_T * CZ_T;
    int64_t S = std::min<int64_t>((ntotal + CZ_T - 1) / CZ_T * CZ_T, rows_fit);
    if (S < std::min<int64_t>(ntotal, 1ll << 20)) return kNoRangeScratch;
    if (ix->range_rows > 0) S = std::min<int64_t>(S, (ix->range_rows + CZ_T - 1) / CZ_T * CZ_T);
    // int8 scratch rows where the int8 scan pays (38 GB of conversion traffic per 10 M rows instead of 46, half the scan):
    // same quantiser as k_ingest_rows, whose running maximum of the int8 error norms covers every row of the index
    // (allow_i8 = false: an index with int8 rows of its own whose int8 choice was already declined for this search)
    const bool use_i8 = allow_i8 && batch_i8_wanted(ix, k, std::min<int64_t>(S, ntotal), nq);
    if (use_i8 && (rc = grow(&ix->x8s_tmp, &ix->x8s_tmp_cap, (size_t)S + 256)) != CSS_OK) return rc;
    if ((size_t)S * ix->dpad > ix->xh_tmp_cap) {   // (exact size: grow() would double a multi-GB buffer)
        if (ix->xh_tmp) CSS_HIP_TRY(hipFree(ix->xh_tmp));
        ix->xh_tmp = nullptr;
        ix->xh_tmp_cap = 0;
        if (hipMalloc((void**)&ix->xh_tmp, (size_t)S * row_b) != hipSuccess) {
            (void)hipGetLastError();
            return kNoRangeScratch;
        }
        ix->xh_tmp_cap = (size_t)S * ix->dpad;
    }
    const int nranges = (int)((ntotal + S - 1) / S);
    float* Dp = D_dev;
    int64_t* Ip = I_dev;
    if (nranges > 1) {
        const size_t need = (size_t)nranges * nq * k;
        if (need > ix->rng_cap) {
            if (ix->rng_d) CSS_HIP_TRY(hipFree(ix->rng_d));
            if (ix->rng_i) CSS_HIP_TRY(hipFree(ix->rng_i));
            ix->rng_d = nullptr;
            ix->rng_i = nullptr;
            ix->rng_cap = 0;
            if (hipMalloc((void**)&ix->rng_d, need * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&ix->rng_i, need * sizeof(int64_t)) != hipSuccess) {   // (nothing enqueued yet: the fallback is safe)
                (void)hipGetLastError();
                if (ix->rng_d) (void)hipFree(ix->rng_d);
                ix->rng_d = nullptr;
                ix->rng_i = nullptr;
                return kNoRangeScratch;
            }
            ix->rng_cap = need;
        }
    }
    ProfScope all("knn_noshadow_ranges", st);
    for (int r = 0; r < nranges; ++r) {
        const int64_t row0 = (int64_t)r * S, n = std::min<int64_t>(S, ntotal - row0);
        if (use_i8) {
            ProfScope ps("knn_rows_to_i8", st);
            const float* src_ = ix->xb + (size_t)row0 * ix->dpad;
            signed char* dst_ = reinterpret_cast<signed char*>(ix->xh_tmp);
            const dim3 grid_((unsigned)((n + 3) / 4));
            switch (ix->dpad) {   // (dpad % 256 == 0 and <= 1024: batch_i8_wanted)
                case 256: hipLaunchKernelGGL(k_rows_to_i8_wide<4>, grid_, dim3(256), 0, st, src_, dst_, ix->x8s_tmp, n); break;
                case 512: hipLaunchKernelGGL(k_rows_to_i8_wide<8>, grid_, dim3(256), 0, st, src_, dst_, ix->x8s_tmp, n); break;
                case 768: hipLaunchKernelGGL(k_rows_to_i8_wide<12>, grid_, dim3(256), 0, st, src_, dst_, ix->x8s_tmp, n); break;
                default: hipLaunchKernelGGL(k_rows_to_i8_wide<16>, grid_, dim3(256), 0, st, src_, dst_, ix->x8s_tmp, n); break;
            }
            CSS_LAUNCH_CHECK();
        } else {
            ProfScope ps("knn_rows_to_bf16", st);
            const int64_t n8 = n * ix->dpad / 8;   // (dpad is a multiple of 64)
            const unsigned blocks = (unsigned)std::min<int64_t>((n8 + 255) / 256, (int64_t)ix->num_cus * 64);
            hipLaunchKernelGGL(k_rows_to_bf16_x8, dim3(blocks), dim3(256), 0, st, ix->xb + (size_t)row0 * ix->dpad, ix->xh_tmp, n8);
            CSS_LAUNCH_CHECK();
        }
        if (nranges > 1) {
            Dp = ix->rng_d + (size_t)r * nq * k;
            Ip = ix->rng_i + (size_t)r * nq * k;
        }
        RowView view(ix, row0, n, use_i8 ? nullptr : ix->xh_tmp, use_i8 ? reinterpret_cast<unsigned char*>(ix->xh_tmp) : nullptr,
                     use_i8 ? ix->x8s_tmp : nullptr);
        SweepGeom sg;
        if ((rc = make_sweep_geom(ix, k, &sg)) != CSS_OK) return rc;
        const int chunk = coarse_max_chunk(ix);
        for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
            const int nqc = (int)std::min<int64_t>(chunk, nq - q0);
            if ((rc = launch_scan_coarse(ix, (int)q0, nqc, k, Dp, Ip, st, sg, false, use_i8,
                                         r == nranges - 1 && q0 + nqc == nq)) != CSS_OK) return rc;
        }
    }
    if (nranges > 1)
        return merge_parts(ix->rng_d, ix->rng_i, nranges, nq * k, nq * k, nq, k, ix->metric, D_dev, I_dev, ix->device, st,
                           "css_index_search");
    return CSS_OK;
}

// q_dev: raw [nq, dim] device queries.  Caller holds ws_mu and a shared lock on mu.
int search_dev_enqueue(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q, float* D_dev,
                       int64_t* I_dev, hipStream_t st);
int search_dev_locked(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q, float* D_dev,
                      int64_t* I_dev, hipStream_t st) {
    // the previous search may still be running on another stream and owns the shared workspaces until its event
    if (ix->ws_pending && ix->ws_stream != st) CSS_HIP_TRY(hipStreamWaitEvent(st, ix->ws_ev, 0));
    const int rc = search_dev_enqueue(ix, q_dev, nq, k, normalize_q, D_dev, I_dev, st);
    // (also after a failed enqueue: whatever was launched before the error still uses the workspaces)
    if (hipEventRecord(ix->ws_ev, st) == hipSuccess) {
        ix->ws_stream = st;
        ix->ws_pending = true;
    } else {
        (void)hipDeviceSynchronize();
        ix->ws_pending = false;
    }
    return rc;
}
int search_dev_enqueue(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q, float* D_dev,
                       int64_t* I_dev, hipStream_t st) {
    CSS_REQUIRE(k >= 1 && k <= CSS_MAX_K, "css_index_search: k=%d outside [1, %d]", k, CSS_MAX_K);
    CSS_REQUIRE(nq >= 0 && nq < (1 << 24), "css_index_search: nq=%lld out of range", (long long)nq);
    if (nq == 0) return CSS_OK;
    const KnnEnv& env = knn_env();
    int rc;
    // rows appended on another stream (css_index_add_dev / _add_synthetic) must have landed
    if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent(st, ix->ingest_ev, 0));
    if ((rc = grow(&ix->qpad, &ix->qpad_cap, (size_t)(nq + 256) * ix->dpad)) != CSS_OK) return rc;
    if ((rc = grow(&ix->qnorm2, &ix->qnorm2_cap, (size_t)nq + 256)) != CSS_OK) return rc;
    if ((rc = grow(&ix->qerr2, &ix->qerr2_cap, (size_t)nq + 256)) != CSS_OK) return rc;
    // (int8 rows: the int8 queries' error norms of every chunk of this search -- never reallocated between two chunks)
    if (ix->x8 != nullptr && (rc = grow(&ix->qerr2_i8, &ix->qerr2_i8_cap, (size_t)nq + 256)) != CSS_OK) return rc;
    if ((rc = grow(&ix->gthr, &ix->gthr_cap, (size_t)nq + 256)) != CSS_OK) return rc;
    // 1..4 queries through the sweep cascade: its init launch prepares the query rows as well (one launch less in front
    // of a 1.4 ms search).  Which shadow rows a search reads is decided once (the per-index int8 feedback counts searches).
    // 3 or 4 queries are VALU-bound in that sweep (10 M rows: 2.65 ms at k = 10): they, and up to 16 queries, sweep on the
    // int8 MFMA where that applies (mfma_sweep_applies); where not, 3 or 4 queries are sooner through the int8 scan of
    // batches from 3 M rows on (1.98 ms; 1 M rows: sweep 0.34 vs scan 0.39 ms; k = 100 goes through the bf16 scan, 3.2 ms
    // against the sweep's 2.8-2.9); two queries always sweep (1.36 vs 1.93 ms).  tools/knn_fewq_probe.py, one session.
    const bool i8_scan_ok = ix->x8 != nullptr && ix->metric == CSS_METRIC_IP && ix->dpad % 256 == 0 && ix->dpad <= 1024 &&
                            env.batch_i8 != 0 && env.loop8 && env.mfma_shape == 16;
    const bool mfma_sweep_ok = mfma_sweep_applies(ix, nq, k);
    const int sweep_max = env.sweep_maxq >= 0 ? env.sweep_maxq : ((k <= 32 && ix->ntotal >= 3000000 && i8_scan_ok) ? 2 : 4);   // VALU sweep
    const bool sweep_base = ix->ntotal > 0 && k <= CSS_KERNEL_MAX_K && env.batch == 0 && (ix->xh != nullptr || ix->x8 != nullptr) &&
                            (ix->search_mode == CSS_SEARCH_COARSE ||
                             (ix->search_mode == CSS_SEARCH_AUTO && (nq > 4 || k > 32 || ix->ntotal >= 100000)));
    const bool sweep_i8 = sweep_base && ix->x8 != nullptr && (nq <= sweep_max || mfma_sweep_ok) && sweep_uses_i8(ix);
    const bool sweep_path = sweep_base && ((mfma_sweep_ok && sweep_i8) || (nq <= sweep_max && (sweep_i8 || ix->xh != nullptr)));
    // query prep: same row kernel as ingest (normalise, zero pad, squared norm)
    if (!sweep_path) {
        const int64_t blocks = (nq + 3) / 4;
        hipLaunchKernelGGL(k_ingest_rows<false>, dim3((unsigned)blocks), dim3(256), 0, st, q_dev, ix->qpad,
                           ix->qnorm2, nq, ix->dim, ix->dpad, normalize_q, 0ull, 0ll, (unsigned short*)nullptr,
                           (int*)nullptr, ix->qerr2, (unsigned char*)nullptr, (float*)nullptr);
        CSS_LAUNCH_CHECK();
    }
    if (ix->ntotal == 0) {
        const int64_t n = nq * k;
        hipLaunchKernelGGL(k_fill_pad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, D_dev, I_dev, n,
                           ix->metric == CSS_METRIC_IP ? -FLT_MAX : FLT_MAX);
        CSS_LAUNCH_CHECK();
        return CSS_OK;
    }
    CSS_REQUIRE(k <= CSS_KERNEL_MAX_K, "css_index_search: internal: k=%d reached the scan kernels (limit %d)", k, CSS_KERNEL_MAX_K);
    SweepGeom sg;
    if ((rc = make_sweep_geom(ix, k, &sg)) != CSS_OK) return rc;

    const int mode = ix->search_mode;
    const bool batch_ok = nq > 16 && k <= kMfmaMaxK && ix->dpad % MF_BK == 0;  // the MFMA scan kernels apply
    // Where the candidate path answers sooner than the exact fp32 kernels (tools/knn_crossover.py on MI355X, 768-d, ms
    // candidate / exact): 5..16 queries at every size (2 k rows 0.07 / 0.15, 100 k 0.16 / 0.28, 1 M 0.44 / 0.93); 1..4
    // queries with the reference's k' = 100 at every size too (2 k 0.065 / 0.134, 100 k 0.13 / 0.17, 1 M 0.32 / 0.66:
    // the exact kernels keep k-entry lists per block), with k = 10 from ~100 k rows on (50 k 0.093 / 0.074, 100 k
    // 0.099 / 0.119, 1 M 0.25 / 0.64).  Round 2 switched at 1.2 M / 0.4 M rows: the cascade has since lost most of its
    // fixed cost and the few-query sweep reads int8 rows.
    const bool coarse_pays = nq > 4 || k > 32 || ix->ntotal >= 100000;
    const bool want_split = mode == CSS_SEARCH_SPLIT || env.batch == 1;   // split-operand candidate scan from the fp32 rows
    const bool want_candidates = env.batch == 0 && (mode == CSS_SEARCH_COARSE || (mode == CSS_SEARCH_AUTO && coarse_pays));
    // Which shadow rows this search reads is decided HERE, once (the per-index int8 feedback counts searches, not chunks).
    // An index with int8 rows only takes the candidate path where the int8 rows are chosen; otherwise it goes on like an
    // index without shadow rows (bf16 scratch ranges for batches, the exact kernels for a few queries).
    if (want_candidates && (ix->xh != nullptr || ix->x8 != nullptr)) {
        const bool sweep = sweep_path;
        const bool use_i8 = ix->x8 != nullptr && (sweep ? sweep_i8 : batch_i8_wanted(ix, k, ix->ntotal, nq));
        if (use_i8 || ix->xh != nullptr) {
            if (sweep) return launch_scan_coarse(ix, 0, (int)nq, k, D_dev, I_dev, st, sg, true, use_i8, true, q_dev, normalize_q);
            const int chunk = coarse_max_chunk(ix);
            for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
                const int nqc = (int)std::min<int64_t>(chunk, nq - q0);
                if ((rc = launch_scan_coarse(ix, (int)q0, nqc, k, D_dev, I_dev, st, sg, false, use_i8, q0 + nqc == nq)) != CSS_OK)
                    return rc;
            }
            return CSS_OK;
        }
    }
    // no shadow rows: batches take the same cascade over bf16 rows rounded on the fly, one row range at a time
    // (CSS_KNN_NOSHADOW=split, a k beyond the MFMA kernels, or no HBM left for the scratch rows: the split-operand scan)
    if (want_candidates && ix->xh == nullptr && nq > 16 && env.noshadow_ranges && ix->dpad % 128 == 0) {
        rc = search_noshadow_ranges(ix, nq, k, D_dev, I_dev, st, ix->x8 == nullptr);
        if (rc != kNoRangeScratch) return rc;
    }
    if (batch_ok && k + kSplitExtra <= kMfmaMaxK && (want_split || (want_candidates && ix->xh == nullptr))) {
        for (int64_t q0 = 0; q0 < nq; q0 += 4096) {   // (chunked like the bf16 cascade: candidate buffers are per query)
            const int nqc = (int)std::min<int64_t>(4096, nq - q0);
            rc = ix->metric == CSS_METRIC_IP ? launch_scan_split_rescore<CSS_METRIC_IP>(ix, (int)q0, nqc, k, D_dev, I_dev, sg, st)
                                             : launch_scan_split_rescore<CSS_METRIC_L2>(ix, (int)q0, nqc, k, D_dev, I_dev, sg, st);
            if (rc != CSS_OK) return rc;
        }
        return CSS_OK;
    }
    // shadow-less batches whose k leaves no room for the split scan's extra ranks (k = 61 .. 64): the fp32-input MFMA
    // scan, not 16-query VALU sweeps
    if (batch_ok && (want_split || (want_candidates && ix->xh == nullptr))) {
        return ix->metric == CSS_METRIC_IP ? launch_scan_fp32mfma<CSS_METRIC_IP>(ix, (int)nq, k, D_dev, I_dev, st)
                                           : launch_scan_fp32mfma<CSS_METRIC_L2>(ix, (int)nq, k, D_dev, I_dev, st);
    }
    // exact fp32 arithmetic inside the scan: fp32-input MFMA for batches, VALU sweeps for up to 16 queries
    if (batch_ok && (mode == CSS_SEARCH_EXACT_FP32 || env.batch == 2)) {
        return ix->metric == CSS_METRIC_IP ? launch_scan_fp32mfma<CSS_METRIC_IP>(ix, (int)nq, k, D_dev, I_dev, st)
                                           : launch_scan_fp32mfma<CSS_METRIC_L2>(ix, (int)nq, k, D_dev, I_dev, st);
    }
    if ((rc = grow_part(ix, (size_t)sg.nq_sweep * sg.G * k)) != CSS_OK) return rc;
    for (int64_t q0 = 0; q0 < nq; q0 += sg.nq_sweep) {
        const int nqc = (int)std::min<int64_t>(sg.nq_sweep, nq - q0);
        if ((rc = search_chunk_small(ix, (int)q0, nqc, k, sg, D_dev, I_dev, st)) != CSS_OK) return rc;
    }
    return CSS_OK;
}


}  // namespace

extern "C" {

int css_index_create(int dim, int metric, int device, css_index** out) {
    CSS_REQUIRE(out != nullptr, "css_index_create: out is NULL");
    CSS_REQUIRE(dim >= 1 && dim <= 8192, "css_index_create: dim=%d outside [1, 8192]", dim);
    CSS_REQUIRE(metric == CSS_METRIC_IP || metric == CSS_METRIC_L2, "css_index_create: unknown metric %d", metric);
    int rc = css::check_device(device);
    if (rc != CSS_OK) return rc;
    DeviceGuard g(device);
    css_index* ix = new css_index();
    ix->dim = dim;
    ix->dpad = (dim + 63) / 64 * 64;
    ix->metric = metric;
    ix->device = device;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) == hipSuccess) ix->num_cus = p.multiProcessorCount;
    hipError_t e = hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ix;
        return css::hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
    e = hipEventCreateWithFlags(&ix->ws_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ix->stream);
        delete ix;
        return css::hip_fail(e, "hipEventCreate", __FILE__, __LINE__);
    }
    e = hipEventCreateWithFlags(&ix->ingest_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&ix->maxn2, 3 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(ix->maxn2, 0, 3 * sizeof(int));
    // (null-stream memset vs the non-blocking streams every later launch uses: order it here, once)
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (ix->ingest_ev) (void)hipEventDestroy(ix->ingest_ev);
        if (ix->ws_ev) (void)hipEventDestroy(ix->ws_ev);
        (void)hipStreamDestroy(ix->stream);
        delete ix;
        return css::hip_fail(e, "hipMalloc(maxn2)", __FILE__, __LINE__);
    }
    *out = ix;
    return CSS_OK;
}

int css_index_free(css_index* ix) {
    if (!ix) return CSS_OK;
    DeviceGuard g(ix->device);
    (void)hipStreamSynchronize(ix->stream);
    if (ix->ingest_pending) (void)hipEventSynchronize(ix->ingest_ev);
    for (css_index::I8Feedback* f : {&ix->fb_batch, &ix->fb_sweep})
        if (f->h_nflag) {   // (the count of the last int8 search may still be on its way)
            (void)hipDeviceSynchronize();
            (void)hipEventDestroy(f->ev);
            (void)hipHostFree(f->h_nflag);
        }
    void* ptrs[] = {ix->xb, ix->xnorm2, ix->xh, ix->x8, ix->x8s, ix->maxn2, ix->q_raw, ix->qpad, ix->qnorm2, ix->qerr2, ix->qerr2_i8, ix->qscale, ix->gthr, ix->qsplit,
                    ix->part_s, ix->part_i, ix->out_i, ix->stage, ix->qh, ix->cthr, ix->cand_n,
                    ix->cflags, ix->cand_s, ix->cand_i, ix->cpace, ix->fs_state, ix->mask_ws, ix->excl_ws, ix->fix_s, ix->fix_i, ix->fix_lock,
                    ix->qh2, ix->thr2, ix->rs_work, ix->cand_n2, ix->cand_s2, ix->cand_i2, ix->flagB, ix->xh_tmp, ix->x8s_tmp, ix->rng_d, ix->rng_i};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);  // (hipFree waits for the device: nothing enqueued by a _dev call still runs)
    if (ix->h_stage) (void)hipHostFree(ix->h_stage);
    if (ix->ingest_ev) (void)hipEventDestroy(ix->ingest_ev);
    if (ix->ws_ev) (void)hipEventDestroy(ix->ws_ev);
    (void)hipStreamDestroy(ix->stream);
    delete ix;
    return CSS_OK;
}

int css_index_reset(css_index* ix) {
    CSS_REQUIRE(ix, "css_index_reset: NULL index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    ix->ntotal = 0;
    ix->ntotal_pub.store(0);
    if (!ix->xh) ix->shadow = -1;
    DeviceGuard g(ix->device);
    if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent(ix->stream, ix->ingest_ev, 0));
    if (ix->ws_pending) CSS_HIP_TRY(hipStreamWaitEvent(ix->stream, ix->ws_ev, 0));   // a search enqueued on another stream still reads maxn2
    CSS_HIP_TRY(hipMemsetAsync(ix->maxn2, 0, 3 * sizeof(int), ix->stream));
    CSS_HIP_TRY(hipStreamSynchronize(ix->stream));
    return CSS_OK;
}

int css_index_reserve(css_index* ix, int64_t n) {
    CSS_REQUIRE(ix && n >= 0, "css_index_reserve: bad argument");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (n <= ix->cap) return CSS_OK;
    // exact-size allocation (no 1.5x growth): a 245 GB shard must not over-allocate
    return reallocate_rows(ix, n);
}

int css_index_ntotal(const css_index* ix, int64_t* n) {
    CSS_REQUIRE(ix && n, "css_index_ntotal: NULL argument");
    *n = ix->ntotal_pub.load();
    return CSS_OK;
}

int css_index_dim(const css_index* ix, int* dim) {
    CSS_REQUIRE(ix && dim, "css_index_dim: NULL argument");
    *dim = ix->dim;
    return CSS_OK;
}

int css_index_metric(const css_index* ix, int* metric) {
    CSS_REQUIRE(ix && metric, "css_index_metric: NULL argument");
    *metric = ix->metric;
    return CSS_OK;
}

int css_index_device(const css_index* ix, int* device) {
    CSS_REQUIRE(ix && device, "css_index_device: NULL argument");
    *device = ix->device;
    return CSS_OK;
}

int css_index_set_search_mode(css_index* ix, int mode) {
    CSS_REQUIRE(ix, "css_index_set_search_mode: NULL index");
    CSS_REQUIRE(mode == CSS_SEARCH_AUTO || mode == CSS_SEARCH_EXACT_FP32 || mode == CSS_SEARCH_COARSE || mode == CSS_SEARCH_SPLIT,
                "css_index_set_search_mode: unknown mode %d", mode);
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    ix->search_mode = mode;
    return CSS_OK;
}

int css_index_last_flagged(css_index* ix, int64_t* n) {
    CSS_REQUIRE(ix && n, "css_index_last_flagged: NULL argument");
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);
    *n = 0;
    if (ix->last_nflag == nullptr) return CSS_OK;
    DeviceGuard g(ix->device);
    CSS_HIP_TRY(hipDeviceSynchronize());  // diagnostics: whichever stream the search ran on
    int v = 0;
    CSS_HIP_TRY(hipMemcpy(&v, ix->last_nflag, sizeof(int), hipMemcpyDeviceToHost));
    *n = v;
    return CSS_OK;
}

int css_index_set_range_rows(css_index* ix, int64_t rows) {
    CSS_REQUIRE(ix, "css_index_set_range_rows: NULL index");
    CSS_REQUIRE(rows >= 0, "css_index_set_range_rows: rows < 0");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    ix->range_rows = rows;
    return CSS_OK;
}

int css_index_last_swept(css_index* ix, int64_t* n) {
    CSS_REQUIRE(ix && n, "css_index_last_swept: NULL argument");
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);
    *n = 0;
    if (ix->last_nswept == nullptr) return CSS_OK;
    DeviceGuard g(ix->device);
    CSS_HIP_TRY(hipDeviceSynchronize());
    int v = 0;
    CSS_HIP_TRY(hipMemcpy(&v, ix->last_nswept, sizeof(int), hipMemcpyDeviceToHost));
    *n = v;
    return CSS_OK;
}

int css_index_shadow_info(css_index* ix, int* has_bf16, int* has_int8) {
    CSS_REQUIRE(ix && has_bf16 && has_int8, "css_index_shadow_info: NULL argument");
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);   // (a RowView of a running enqueue swaps xh)
    *has_bf16 = ix->xh != nullptr ? 1 : 0;
    *has_int8 = ix->x8 != nullptr ? 1 : 0;
    return CSS_OK;
}

int css_index_set_shadow(css_index* ix, int policy) {
    CSS_REQUIRE(ix, "css_index_set_shadow: NULL index");
    CSS_REQUIRE(policy >= -1 && policy <= 2, "css_index_set_shadow: policy %d outside {-1, 0, 1, 2}", policy);
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    if (ix->ntotal != 0) {
        css::set_error("css_index_set_shadow: the index already holds %lld rows (set the policy on an empty index)",
                       (long long)ix->ntotal);
        return CSS_ERR_STATE;
    }
    DeviceGuard g(ix->device);
    ix->shadow_policy = policy;
    if (ix->xh) {  // start over: the next add decides again
        CSS_HIP_TRY(hipFree(ix->xh));
        ix->xh = nullptr;
    }
    if (ix->x8) {
        CSS_HIP_TRY(hipFree(ix->x8));
        CSS_HIP_TRY(hipFree(ix->x8s));
        ix->x8 = nullptr;
        ix->x8s = nullptr;
    }
    ix->shadow = -1;
    return CSS_OK;
}

int css_index_set_id_base(css_index* ix, int64_t base) {
    CSS_REQUIRE(ix, "css_index_set_id_base: NULL index");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    ix->id_base = base;
    return CSS_OK;
}

int css_index_add(css_index* ix, const float* x_host, int64_t n, int normalize) {
    CSS_REQUIRE(ix, "css_index_add: NULL index");
    CSS_REQUIRE(n >= 0, "css_index_add: n < 0");
    if (n == 0) return CSS_OK;
    CSS_REQUIRE(x_host, "css_index_add: x is NULL");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);
    DeviceGuard g(ix->device);
    if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent(ix->stream, ix->ingest_ev, 0));
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc != CSS_OK) return rc;
    // stage through a bounded device buffer so huge adds do not double the footprint
    const int64_t chunk = std::max<int64_t>(1, (64ll << 20) / ((int64_t)ix->dim * 4));
    if ((rc = grow(&ix->stage, &ix->stage_cap, (size_t)std::min(n, chunk) * ix->dim)) != CSS_OK) return rc;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int64_t m = std::min(chunk, n - r0);
        CSS_HIP_TRY(hipMemcpyAsync(ix->stage, x_host + (size_t)r0 * ix->dim, (size_t)m * ix->dim * 4,
                                   hipMemcpyHostToDevice, ix->stream));
        if ((rc = ingest(ix, ix->stage, m, normalize, false, 0, 0, ix->stream)) != CSS_OK) return rc;
        CSS_HIP_TRY(hipStreamSynchronize(ix->stream));
        ix->ntotal += m;
        ix->ntotal_pub.store(ix->ntotal);
    }
    return CSS_OK;
}

int css_index_add_dev(css_index* ix, const float* x_dev, int64_t n, int normalize, void* stream) {
    CSS_REQUIRE(ix, "css_index_add_dev: NULL index");
    CSS_REQUIRE(n >= 0, "css_index_add_dev: n < 0");
    if (n == 0) return CSS_OK;
    CSS_REQUIRE(x_dev, "css_index_add_dev: x is NULL");
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc != CSS_OK) return rc;
    // an earlier asynchronous add may have used another stream: chain it, so that the ONE event below covers it too
    if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ix->ingest_ev, 0));
    for (int64_t r0 = 0; r0 < n; r0 += (1ll << 30)) {
        const int64_t m = std::min<int64_t>(1ll << 30, n - r0);
        if ((rc = ingest(ix, x_dev + (size_t)r0 * ix->dim, m, normalize, false, 0, 0, (hipStream_t)stream)) != CSS_OK)
            return rc;
        ix->ntotal += m;
        ix->ntotal_pub.store(ix->ntotal);
    }
    // the rows are written asynchronously on the caller's stream: later searches / reallocations / exports wait for this
    CSS_HIP_TRY(hipEventRecord(ix->ingest_ev, (hipStream_t)stream));
    ix->ingest_pending = true;
    return CSS_OK;
}

int css_index_add_synthetic(css_index* ix, int64_t n, uint64_t seed, int64_t first_row, int normalize,
                            void* stream) {
    CSS_REQUIRE(ix, "css_index_add_synthetic: NULL index");
    CSS_REQUIRE(n >= 0, "css_index_add_synthetic: n < 0");
    if (n == 0) return CSS_OK;
    std::unique_lock<std::shared_mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    int rc = ensure_capacity(ix, ix->ntotal + n);
    if (rc != CSS_OK) return rc;
    if (ix->ingest_pending) CSS_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, ix->ingest_ev, 0));
    for (int64_t r0 = 0; r0 < n; r0 += (1ll << 30)) {
        const int64_t m = std::min<int64_t>(1ll << 30, n - r0);
        if ((rc = ingest(ix, nullptr, m, normalize, true, seed, first_row + r0, (hipStream_t)stream)) != CSS_OK)
            return rc;
        ix->ntotal += m;
        ix->ntotal_pub.store(ix->ntotal);
    }
    CSS_HIP_TRY(hipEventRecord(ix->ingest_ev, (hipStream_t)stream));
    ix->ingest_pending = true;
    return CSS_OK;
}

int css_index_export(const css_index* cix, int64_t row0, int64_t n, float* x_out_host) {
    css_index* ix = const_cast<css_index*>(cix);
    CSS_REQUIRE(ix && x_out_host, "css_index_export: NULL argument");
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);   // (a search may have the index narrowed to a row range while it enqueues)
    CSS_REQUIRE(row0 >= 0 && n >= 0 && row0 + n <= ix->ntotal, "css_index_export: rows [%lld, %lld) outside [0, %lld)",
                (long long)row0, (long long)(row0 + n), (long long)ix->ntotal);
    if (n == 0) return CSS_OK;
    DeviceGuard g(ix->device);
    if (ix->ingest_pending) CSS_HIP_TRY(hipEventSynchronize(ix->ingest_ev));
    CSS_HIP_TRY(hipMemcpy2D(x_out_host, (size_t)ix->dim * 4, ix->xb + (size_t)row0 * ix->dpad, (size_t)ix->dpad * 4,
                            (size_t)ix->dim * 4, (size_t)n, hipMemcpyDeviceToHost));
    return CSS_OK;
}

// RAII: the allow-bitmap of the search in progress (read by the kernel launchers); caller holds ws_mu
namespace {
struct MaskScope {
    css_index* ix;
    MaskScope(css_index* i, const uint32_t* m) : ix(i) { ix->cur_mask = m; }
    ~MaskScope() { ix->cur_mask = nullptr; }
};

// Any k in [1, CSS_MAX_K].  Up to CSS_KERNEL_MAX_K: one search.  Beyond: per query, ceil(k / CSS_KERNEL_MAX_K) passes of
// the SAME search paths, pass p over the allowed rows the passes before it did not return (exclusion bitmap), its
// results written straight into columns [p * 128, ..) of the query's output row; the comparator is total (score, then
// id), so the concatenation is the exact top-k, and one final sort puts entries whose scores came from different
// summation orders (fix-up sweep vs rescoring: <= 1e-6 apart) in order.  Everything is enqueued on `st`; nothing
// waits for the device.  Caller holds ws_mu and a shared lock on mu.
int search_any_k(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q, const uint32_t* allow_dev,
                 float* D_dev, int64_t* I_dev, hipStream_t st) {
    CSS_REQUIRE(k >= 1 && k <= CSS_MAX_K, "css_index_search: k=%d outside [1, %d]", k, CSS_MAX_K);
    if (k <= CSS_KERNEL_MAX_K || ix->ntotal == 0 || nq == 0) {
        MaskScope ms(ix, allow_dev);
        return search_dev_locked(ix, q_dev, nq, k, normalize_q, D_dev, I_dev, st);
    }
    CSS_REQUIRE(nq < (1 << 24), "css_index_search: nq=%lld out of range", (long long)nq);
    int rc;
    const int64_t words = (ix->ntotal + 31) / 32;
    if (ix->ws_pending && ix->ws_stream != st) CSS_HIP_TRY(hipStreamWaitEvent(st, ix->ws_ev, 0));   // excl_ws is a shared workspace
    if ((rc = grow(&ix->excl_ws, &ix->excl_ws_cap, (size_t)words)) != CSS_OK) return rc;
    for (int64_t q = 0; q < nq; ++q) {
        hipLaunchKernelGGL(k_mask_init, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, ix->excl_ws, allow_dev, words);
        CSS_LAUNCH_CHECK();
        for (int p = 0; p < k; p += CSS_KERNEL_MAX_K) {
            const int kk = std::min(CSS_KERNEL_MAX_K, k - p);
            float* Dq = D_dev + (size_t)q * k + p;
            int64_t* Iq = I_dev + (size_t)q * k + p;
            {
                MaskScope ms(ix, ix->excl_ws);
                if ((rc = search_dev_locked(ix, q_dev + (size_t)q * ix->dim, 1, kk, normalize_q, Dq, Iq, st)) != CSS_OK) return rc;
            }
            if (p + kk < k) {
                hipLaunchKernelGGL(k_mask_clear, dim3(1), dim3(CSS_KERNEL_MAX_K), 0, st, ix->excl_ws, (const int64_t*)Iq, kk, ix->id_base);
                CSS_LAUNCH_CHECK();
            }
        }
    }
    if (ix->metric == CSS_METRIC_IP) hipLaunchKernelGGL(k_sort_rows<CSS_METRIC_IP>, dim3((unsigned)nq), dim3(1024), 0, st, D_dev, I_dev, k);
    else hipLaunchKernelGGL(k_sort_rows<CSS_METRIC_L2>, dim3((unsigned)nq), dim3(1024), 0, st, D_dev, I_dev, k);
    CSS_LAUNCH_CHECK();
    if (hipEventRecord(ix->ws_ev, st) == hipSuccess) {   // (the exclusion bitmap is in use until here)
        ix->ws_stream = st;
        ix->ws_pending = true;
    }
    return CSS_OK;
}
}  // namespace

int css_index_search_masked_dev(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q,
                                const uint32_t* allow_bits_dev, float* D_dev, int64_t* I_dev, void* stream) {
    CSS_REQUIRE(ix, "css_index_search_dev: NULL index");
    CSS_REQUIRE(nq == 0 || (q_dev && D_dev && I_dev), "css_index_search_dev: NULL buffer");
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);
    DeviceGuard g(ix->device);
    return search_any_k(ix, q_dev, nq, k, normalize_q, allow_bits_dev, D_dev, I_dev, (hipStream_t)stream);
}

int css_index_search_dev(css_index* ix, const float* q_dev, int64_t nq, int k, int normalize_q, float* D_dev,
                         int64_t* I_dev, void* stream) {
    return css_index_search_masked_dev(ix, q_dev, nq, k, normalize_q, nullptr, D_dev, I_dev, stream);
}

int css_index_search_masked(css_index* ix, const float* q_host, int64_t nq, int k, int normalize_q,
                            const uint32_t* allow_bits_host, float* D_host, int64_t* I_host) {
    CSS_REQUIRE(ix, "css_index_search: NULL index");
    CSS_REQUIRE(nq >= 0, "css_index_search: nq < 0");
    if (nq == 0) return CSS_OK;
    CSS_REQUIRE(q_host && D_host && I_host, "css_index_search: NULL buffer");
    CSS_REQUIRE(k >= 1 && k <= CSS_MAX_K, "css_index_search: k=%d outside [1, %d]", k, CSS_MAX_K);
    std::shared_lock<std::shared_mutex> lk(ix->mu);
    std::lock_guard<std::mutex> wl(ix->ws_mu);
    DeviceGuard g(ix->device);
    int rc;
    if ((rc = grow(&ix->q_raw, &ix->q_raw_cap, (size_t)nq * ix->dim)) != CSS_OK) return rc;
    {
        size_t need = (size_t)nq * k;
        if (need > ix->out_cap) {
            if (ix->out_i) CSS_HIP_TRY(hipFree(ix->out_i));
            ix->out_i = nullptr;
            ix->out_cap = 0;
            CSS_HIP_TRY(hipMalloc((void**)&ix->out_i, need * (sizeof(int64_t) + sizeof(float))));
            ix->out_cap = need;
        }
    }
    const size_t q_bytes = (size_t)nq * ix->dim * 4, out_bytes = (size_t)nq * k * 12;
    const bool staged = q_bytes <= css_index::kHostStage && out_bytes <= css_index::kHostStage;
    if (staged && ix->h_stage == nullptr)
        CSS_HIP_TRY(hipHostMalloc((void**)&ix->h_stage, 2 * css_index::kHostStage, hipHostMallocDefault));
    const uint32_t* mask_dev = nullptr;
    if (allow_bits_host && ix->ntotal > 0) {
        const size_t words = (size_t)((ix->ntotal + 31) / 32);
        if ((rc = grow(&ix->mask_ws, &ix->mask_ws_cap, words)) != CSS_OK) return rc;
        CSS_HIP_TRY(hipMemcpyAsync(ix->mask_ws, allow_bits_host, words * sizeof(uint32_t), hipMemcpyHostToDevice, ix->stream));
        mask_dev = ix->mask_ws;
    }
    // (the output rows of THIS call sit at the front of the allocation: ids of nq * k entries, then their scores)
    float* const d_out = reinterpret_cast<float*>(ix->out_i + (size_t)nq * k);
    if (staged) {
        memcpy(ix->h_stage, q_host, q_bytes);
        CSS_HIP_TRY(hipMemcpyAsync(ix->q_raw, ix->h_stage, q_bytes, hipMemcpyHostToDevice, ix->stream));
    } else {
        CSS_HIP_TRY(hipMemcpyAsync(ix->q_raw, q_host, q_bytes, hipMemcpyHostToDevice, ix->stream));
    }
    if ((rc = search_any_k(ix, ix->q_raw, nq, k, normalize_q, mask_dev, d_out, ix->out_i, ix->stream)) != CSS_OK) return rc;
    if (staged) {   // one copy into pinned memory, one wait
        char* back = ix->h_stage + css_index::kHostStage;
        CSS_HIP_TRY(hipMemcpyAsync(back, ix->out_i, out_bytes, hipMemcpyDeviceToHost, ix->stream));
        CSS_HIP_TRY(hipStreamSynchronize(ix->stream));
        memcpy(I_host, back, (size_t)nq * k * 8);
        memcpy(D_host, back + (size_t)nq * k * 8, (size_t)nq * k * 4);
        return CSS_OK;
    }
    CSS_HIP_TRY(hipMemcpyAsync(D_host, d_out, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ix->stream));
    CSS_HIP_TRY(hipMemcpyAsync(I_host, ix->out_i, (size_t)nq * k * 8, hipMemcpyDeviceToHost, ix->stream));
    CSS_HIP_TRY(hipStreamSynchronize(ix->stream));
    return CSS_OK;
}

int css_index_search(css_index* ix, const float* q_host, int64_t nq, int k, int normalize_q, float* D_host,
                     int64_t* I_host) {
    return css_index_search_masked(ix, q_host, nq, k, normalize_q, nullptr, D_host, I_host);
}


int css_merge_topk_dev(const float* Dp, const int64_t* Ip, int nparts, int64_t nq, int k, int metric, float* D,
                       int64_t* I, int device, void* stream) {
    return merge_parts(Dp, Ip, nparts, nq * k, nq * k, nq, k, metric, D, I, device, stream, "css_merge_topk_dev");
}

int css_merge_topk_packed_dev(const void* packed, int nparts, int64_t record_bytes, int64_t nq, int k, int metric,
                              float* D, int64_t* I, int device, void* stream) {
    CSS_REQUIRE(packed != nullptr, "css_merge_topk_packed_dev: NULL buffer");
    CSS_REQUIRE(nq >= 0 && k >= 1 && record_bytes >= 12 * nq * k && record_bytes % 8 == 0,
                "css_merge_topk_packed_dev: record of %lld bytes cannot hold [%lld x %d] ids and scores (multiple of 8)",
                (long long)record_bytes, (long long)nq, k);
    const int64_t* Ip = reinterpret_cast<const int64_t*>(packed);
    const float* Dp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(packed) + 8 * nq * k);
    return merge_parts(Dp, Ip, nparts, record_bytes / 4, record_bytes / 8, nq, k, metric, D, I, device, stream,
                       "css_merge_topk_packed_dev");
}

}  // extern "C"
