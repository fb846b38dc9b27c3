// css_knn_coarse.h -- batched exact search as "coarse bf16 scan + exact fp32 rescoring".
// Included by css_index.hip (inside its anonymous namespace, after the MFMA vector types).
//
// Batched flat search (src/storage.py:424-436 with many queries) is a [rows x 768] . [768 x nq] product
// followed by a top-k.  Computing every product to fp32 grade costs 3 bf16 MFMAs (split operands) and is
// MFMA bound; but only ~k of the 10^7 scores per query matter.  So:
//
//   coarse   c(x, q) = bf16(x) . bf16(q), fp32 accumulate, ONE bf16 MFMA product per (row, query, k),
//            from a bf16 shadow copy of the index rows (xh, +50 % HBM, kept next to the fp32 rows).
//            Round-to-nearest bf16 has relative error <= 2^-8 per operand, so for every row
//                |c(x,q) - x.q|  <=  eps_q = (2^-7 + 2^-11) ||q|| max_row ||x||      (Cauchy-Schwarz;
//            the 2^-11 covers the fp32 accumulation of either side).
//   select   Let Tc be the k-th largest coarse score of a query.  Every row of the exact top-k has
//            c >= Tc - 2 eps_q  (the k rows with the largest c have exact scores >= Tc - eps, so the exact
//            k-th best is >= Tc - eps, and a row that good has c >= Tc - 2 eps).  The candidate band
//            {c >= Tc - 2 eps} holds k + a few dozen rows.
//   rescore  exact fp32 dot products of the band rows against the fp32 index, sorted by (score desc, id asc).
//
// Finding the band without sorted lists in the hot kernel: the scan runs as a cascade over a nested,
// uniformly strided sample of row tiles (stage 0: every s0-th tile, then strides s0/4, s0/16, ... 1, each
// stage only the tiles not seen before).  Before each stage the k-th best coarse score of everything seen
// so far gives a valid lower bound of the final Tc, so the stage appends only rows with c >= thr = Tc' - 2 eps
// to a per-query candidate buffer (lane-local atomicAdd + two stores; ~3k + band rows per stage).  The hot
// kernel is then a plain persistent bf16 GEMM (the encoder's LDS-DMA ring) whose epilogue is 128 compares.
// A query whose buffer or band overflows is flagged and re-run on the exact path by the host.
//
// Measured on MI355X (10M x 768, 1000 queries, 32x32x16 MFMA): main stage 11.6 ms; LDS-DMA only 7.4 ms (~49 GB/s per CU into
// LDS, the same with one or two stages in flight: a bandwidth, not a latency limit), MFMA + LDS reads only
// 7.4 ms, epilogue ~0.8 ms.  A variant with separate 3-slot query / 2-slot row rings (DMA issue spread between
// the MFMAs and staggered between the two waves of a SIMD) was 5 % slower and is not kept.  On v_mfma_f32_16x16x32_bf16
// (the default) the main stage takes 11.1 ms: same cycles and LDS traffic, higher sustained clock.
#pragma once

constexpr int CZ_T = 256;            // rows per tile and queries per tile
constexpr int CZ_RB = 128;           // bytes of K per LDS stage row (64 bf16)
constexpr int CZ_STAGE = 2 * CZ_T * CZ_RB;   // 64 KiB: [256 query rows | 256 index rows] x 128 B
constexpr int CZ_NST = 2;            // ring stages (measured fastest for the encoder GEMM: fewest barriers)
constexpr int CZ_CAP = 4096;         // candidate slots per query
constexpr int CZ_RMAX = CZ_CAP;      // largest band that is rescored in place: the whole buffer (1024 until round 2: on
                                     // clustered rows 19 % of the queries then took the 15x slower exact fix-up)

__device__ __forceinline__ int cz_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// fp32 rows -> bf16 rows (queries; rows >= n_real are zero filled)
__global__ void k_rows_to_bf16(const float* __restrict__ in, unsigned short* __restrict__ out, int64_t n_real,
                               int64_t n_pad, int dpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad * dpad) return;
    const float x = i < n_real * dpad ? in[i] : 0.f;
    out[i] = __builtin_bit_cast(unsigned short, (__bf16)x);
}

// fp32 rows -> bf16 rows, 8 elements per thread (16-B stores), grid-stride: the bf16 ranges of shadow-less indexes
// (a dispatch carries at most 2^32 work-items, a 10 M-row range has 7.7e9 elements)
__global__ __launch_bounds__(256) void k_rows_to_bf16_x8(const float* __restrict__ in, unsigned short* __restrict__ out, int64_t n8) {
    typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(in)[2 * i], b = reinterpret_cast<const float4*>(in)[2 * i + 1];
        v8bf_t h;
        h[0] = (__bf16)a.x; h[1] = (__bf16)a.y; h[2] = (__bf16)a.z; h[3] = (__bf16)a.w;
        h[4] = (__bf16)b.x; h[5] = (__bf16)b.y; h[6] = (__bf16)b.z; h[7] = (__bf16)b.w;
        reinterpret_cast<v8bf_t*>(out)[i] = h;
    }
}

// fp32 rows -> int8 rows for the int8 MFMA scan: one wave per row; signed byte = round(x / s), s = max|x| / 127,
// err2 = ||x - s k||^2.  Queries (err2 = the query side of the measured error band, cz_eps; rows nq .. nq_pad-1:
// zeros, scale 1) and the scratch rows of shadow-less indexes (err2 null: the quantiser is k_ingest_rows', whose
// running maximum already covers these rows).
__global__ __launch_bounds__(256) void k_rows_to_i8(const float* __restrict__ in, signed char* __restrict__ out,
                                                    float* __restrict__ scale, float* __restrict__ err2, int nq, int nq_pad,
                                                    int dpad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq_pad) return;
    const float* src = in + (size_t)row * dpad;
    float amax = 0.f;
    if (row < nq)
        for (int c = lane; c < dpad; c += 64) amax = fmaxf(amax, fabsf(src[c]));
    amax = wave_allmax(amax);
    const float s8 = amax > 0.f ? amax / 127.f : 1.f, inv8 = amax > 0.f ? 127.f / amax : 0.f;
    float e2 = 0.f;
    for (int c = lane; c < dpad; c += 64) {
        const float v = row < nq ? src[c] : 0.f;
        const float k8 = fminf(fmaxf(rintf(v * inv8), -127.f), 127.f);
        out[(size_t)row * dpad + c] = (signed char)(int)k8;
        const float d8 = fmaf(-s8, k8, v);
        e2 = fmaf(d8, d8, e2);
    }
    e2 = wave_allsum(e2);
    if (lane == 0) {
        scale[row] = s8;
        if (err2) err2[row] = e2;
    }
}

// The same for whole row ranges (the int8 scratch rows of shadow-less indexes): one wave per row, a lane holds EPL =
// dpad / 64 consecutive elements (EPL a multiple of 4) in registers -- one pass, 16-byte loads, 4-byte stores of packed
// bytes -- instead of two strided passes with byte stores (12.9 ms per 10 M rows; this form: the conversion's 38 GB at
// the streaming rate).  Same arithmetic as k_rows_to_i8 / k_ingest_rows element for element.
template <int EPL>
__global__ __launch_bounds__(256) void k_rows_to_i8_wide(const float* __restrict__ in, signed char* __restrict__ out,
                                                         float* __restrict__ scale, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float4* src = reinterpret_cast<const float4*>(in + (size_t)row * (64 * EPL) + (size_t)lane * EPL);
    float v[EPL];
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
        const float4 t = src[j];
        v[4 * j] = t.x;
        v[4 * j + 1] = t.y;
        v[4 * j + 2] = t.z;
        v[4 * j + 3] = t.w;
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < EPL; ++j) amax = fmaxf(amax, fabsf(v[j]));
    amax = wave_allmax(amax);
    const float s8 = amax > 0.f ? amax / 127.f : 1.f, inv8 = amax > 0.f ? 127.f / amax : 0.f;
    unsigned* dst = reinterpret_cast<unsigned*>(out + (size_t)row * (64 * EPL) + (size_t)lane * EPL);
#pragma unroll
    for (int j = 0; j < EPL / 4; ++j) {
        unsigned w = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float k8 = fminf(fmaxf(rintf(v[4 * j + e] * inv8), -127.f), 127.f);
            w |= ((unsigned)(int)k8 & 0xFFu) << (8 * e);
        }
        dst[j] = w;
    }
    if (lane == 0) scale[row] = s8;
}

// k_coarse_select: at most this many of the best coarse candidates are scored exactly for the one-eps threshold
constexpr int CZ_EXK = 256;
// Band rescoring (k_rescore_parts): a band is split over at most CZ_PARTS blocks
constexpr int CZ_PARTS = 16;
// Spacing of the per-query candidate counters, in ints: a 128-byte line each.  The batch scan, which adds once per
// (row tile, query) with a count, measured no difference to packed counters; the int8 MFMA sweep of 3..32 queries adds
// once per hit and lane, and with all sixteen counters on one line its appends queued up behind each other (1 M rows,
// 16 queries, k = 100: 0.89 ms packed, 0.54 ms a line apart; 4 KB apart: the same).
#ifndef CZ_NS_STRIDE
#define CZ_NS_STRIDE 32
#endif
constexpr int CZ_NS = CZ_NS_STRIDE;
// k_sweep_cascade (the 1..4-query cascade in one launch, below): its limits and its state words
constexpr int CZ_FS_MAXST = 16;            // stages (growth 4: 4^15 tiles)
constexpr int CZ_FS_RING = 16;             // pending quarter tiles per wave
constexpr int CZ_FS_SENT = 0x7FFFFFFF;     // key word "not published yet" (a NaN's key: a published threshold never is one)
// state words (k_coarse_init).  Agent-scope atomics AND sc1 loads of one 128-byte line complete at only ~30 per microsecond
// chip-wide (they are served behind the L2s), so every hot word has a line -- and a 4-KiB stride: a channel -- of its own:
// [0] next ticket | [LINE] abort | [LINE (2 + s)] quarters of stage s appended | from CZ_FS_KEY: CZ_FS_COPIES copies (one per
// blockIdx % CZ_FS_COPIES, 256 bytes each) of the key words, [4 s + q] = threshold that stage s applies to query q
constexpr int CZ_FS_LINE = 1024, CZ_FS_COPIES = 64;
constexpr int CZ_FS_ABORT = CZ_FS_LINE, CZ_FS_DONE = 2 * CZ_FS_LINE, CZ_FS_KEY = (2 + CZ_FS_MAXST) * CZ_FS_LINE;
constexpr int CZ_FS_KEYWORDS = 4 * CZ_FS_MAXST * CZ_FS_COPIES, CZ_FS_WORDS = CZ_FS_KEY + CZ_FS_KEYWORDS;

// thr = -inf (real queries) / +inf (padding), counters and flags cleared; with them (one launch instead of three)
// the sibling-pacing counters of the scan stages and the counters / thresholds of the second pass: slots that no
// flagged query claims take part in that scan with thr2 = +inf, i.e. without ever appending
// q_raw != null (the 1..4-query sweep): this launch also prepares the query rows -- normalise, zero pad, squared norm:
// ingest_row, one wave per query, the arithmetic of the separate k_ingest_rows launch it replaces
__global__ void k_coarse_init(float* thr, int* cand_n, int* flags, int* nflag, int nq, int nq_pad, int n0rows,
                              int* __restrict__ pace, int npace, int* __restrict__ cand_n2, float* __restrict__ thr2,
                              int* __restrict__ nflagB, int f2max, int* __restrict__ fs, const float* __restrict__ q_raw,
                              float* __restrict__ q_out, float* __restrict__ qnorm2_out, float* __restrict__ qerr2_out,
                              int dim, int dpad, int normalize_q) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (q_raw != nullptr && (i >> 6) < nq)
        ingest_row<false>(i >> 6, i & 63, q_raw, q_out, qnorm2_out, dim, dpad, normalize_q, 0ull, 0ll, (unsigned short*)nullptr,
                          (int*)nullptr, qerr2_out, (unsigned char*)nullptr, (float*)nullptr);
    // k_sweep_cascade: counters zero, stage 0 open (threshold -inf), every later stage unpublished
    if (fs != nullptr && i < CZ_FS_KEYWORDS) {
        fs[CZ_FS_KEY + i] = (i & (4 * CZ_FS_MAXST - 1)) < 4 ? f2key(-INFINITY) : CZ_FS_SENT;
        if (i < 2 + CZ_FS_MAXST) fs[i * CZ_FS_LINE] = 0;
    }
    if (i == 0) {   // (word 1 behind a count: the blocks-done counter of the exact fix-up, k_scan_small<FIX>)
        nflag[0] = nflag[1] = 0;
        if (nflagB) nflagB[0] = nflagB[1] = 0;
    }
    if (i < npace) pace[i] = 0;
    if (i < f2max) {
        cand_n2[(size_t)i * CZ_NS] = 0;
        thr2[i] = INFINITY;
    }
    if (i >= nq_pad) return;
    thr[i] = i < nq ? -INFINITY : INFINITY;
    cand_n[(size_t)(i) * CZ_NS] = i < nq ? n0rows : 0;   // stage 0 writes its rows to fixed slots
    flags[i] = 0;
}

// allow-bitmap test of a masked search (filter / tombstone push-down): bit r&31 of word r>>5
#define CZ_ALLOWED(MASK_, ROW_) ((MASK_) == nullptr || (((MASK_)[(ROW_) >> 5] >> ((ROW_) & 31)) & 1u))

// Epilogue of one 256x256 tile of k_scan_coarse (uses the kernel's locals).  Accumulator geometry by MFMA shape MS:
//   32x32x16: lane (lq = lane & 31, lg = lane >> 5): query 32 m + lq, register r = row (r&3) + 8 (r>>2) + 4 lg of row tile n
//   16x16x32: lane (lq = lane & 15, lg = lane >> 4): query 16 m + lq, register r = row 4 lg + r of row tile n
#define CZ_QOFF(M_) (MS * (M_) + lq)
#define CZ_ROFF(N_, R_) (MS == 32 ? 32 * (N_) + ((R_) & 3) + 8 * ((R_) >> 2) + 4 * lg : 16 * (N_) + 4 * lg + (R_))
// Wave-local hit list of k_scan_coarse8 (CZ_STAGED kernels): [score | row | query] arrays of CZ_WCAP entries per wave.
// CZ_FLUSH: every lane takes entries lane, lane + 64, ...: one returning atomic per entry, all of a pass in flight
// together.  (Inline-asm LDS access: compiler-visible LDS traffic next to the pending LDS-DMA of the ring costs a
// vmcnt(0), css_encoder_kernels.h.)
constexpr int CZ_WCAP = 256;
// OR of v over the 64 lanes (DPP: four shifts inside the 16-lane rows, then the row totals travel to the last lane)
__device__ __forceinline__ unsigned cz_wave_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
#define CZ_FLUSH()                                                                                                     \
    {                                                                                                                  \
        for (int i_ = lane; i_ < wcount; i_ += 64) {                                                                   \
            float fs_;                                                                                                 \
            unsigned fr_, fq_;                                                                                         \
            const unsigned fa_ = wl_base + (unsigned)i_ * 4u;                                                          \
            asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:%4\n\tds_read_b32 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)" \
                         : "=&v"(fs_), "=&v"(fr_), "=&v"(fq_) : "v"(fa_), "n"(CZ_WCAP * 4), "n"(CZ_WCAP * 8) : "memory"); \
            if constexpr (I8) fs_ *= qsc[fq_];   /* the list holds scores in units of the query scale */             \
            const int slot_ = atomicAdd(&cand_n[(size_t)(fq_) * CZ_NS], 1);                                                              \
            if (slot_ < KCAP) {                                                                                      \
                cand_s[(size_t)fq_ * KCAP + slot_] = fs_;                                                            \
                cand_i[(size_t)fq_ * KCAP + slot_] = fr_;                                                            \
            }                                                                                                          \
        }                                                                                                              \
        wcount = 0;                                                                                                    \
    }
#define CZ_EPILOGUE()                                                                                        \
            const int64_t tile = CZ_TILE_OF(ct_tile);                                                                  \
            const int64_t row0 = tile * CZ_T + wc * 64;                                                                \
            if (xn2 != nullptr) { /* L2: score = 2 x.q - ||x||^2 (||q||^2 is the same for every row of a query) */ \
_Pragma("unroll")                                                                                                      \
                for (int n = 0; n < TN; ++n)                                                                           \
_Pragma("unroll")                                                                                                      \
                    for (int rg = 0; rg < NR / 4; ++rg) {                                                              \
                        const float4 xv = *reinterpret_cast<const float4*>(xn2 + row0 + CZ_ROFF(n, 4 * rg));           \
                        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};                                                  \
_Pragma("unroll")                                                                                                      \
                        for (int m = 0; m < TM; ++m)                                                                   \
_Pragma("unroll")                                                                                                      \
                            for (int e = 0; e < 4; ++e) acc[m][n][4 * rg + e] = fmaf(2.f, acc[m][n][4 * rg + e], -xs[e]); \
                    }                                                                                                  \
            }                                                                                                          \
            if constexpr (I8) { /* int32 accumulators -> scores in units of the query scale: acc * s_row */        \
_Pragma("unroll")                                                                                                      \
                for (int n = 0; n < TN; ++n)                                                                           \
_Pragma("unroll")                                                                                                      \
                    for (int rg = 0; rg < NR / 4; ++rg) {                                                              \
                        /* the row scales of this wave's 64 rows came in by LDS-DMA during the K loop (C8_SCALE_ISSUE) */ \
                        float4 sv4_;                                                                                   \
                        {                                                                                              \
                            const unsigned sa_ = sxs_base + (unsigned)CZ_ROFF(n, 4 * rg) * 4u;                         \
                            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(sv4_) : "v"(sa_) : "memory"); \
                        }                                                                                              \
                        const float svs_[4] = {sv4_.x, sv4_.y, sv4_.z, sv4_.w};                                        \
_Pragma("unroll")                                                                                                      \
                        for (int m = 0; m < TM; ++m)                                                                   \
_Pragma("unroll")                                                                                                      \
                            for (int e = 0; e < 4; ++e)                                                                \
                                acc[m][n][4 * rg + e] = (float)__float_as_int(acc[m][n][4 * rg + e]) * svs_[e];        \
                    }                                                                                                  \
            }                                                                                                          \
            if (dbg & 1) {                                                                                             \
            } else if constexpr (STAGE0) {                                                                             \
                const int64_t u = u0 + (int64_t)ct_tile * ustep;                                                       \
_Pragma("unroll")                                                                                                      \
                for (int m = 0; m < TM; ++m) {                                                                         \
                    const size_t qb = (size_t)(qtile * CZ_T + wr * 128 + CZ_QOFF(m)) * KCAP + (size_t)u * CZ_T + wc * 64; \
                    float sqm_ = 1.f;                                                                                  \
                    if constexpr (I8) sqm_ = ssq[wr * 128 + CZ_QOFF(m)];                                               \
_Pragma("unroll")                                                                                                      \
                    for (int n = 0; n < TN; ++n)                                                                       \
_Pragma("unroll")                                                                                                      \
                        for (int rg = 0; rg < NR / 4; ++rg) { /* registers 4 rg .. 4 rg + 3 are 4 consecutive rows: 16-B stores */ \
                            const int ro = CZ_ROFF(n, 4 * rg);                                                         \
                            float4 sv_;                                                                                \
                            uint4 iv_;                                                                                 \
                            float* svp_ = &sv_.x;                                                                      \
                            unsigned* ivp_ = &iv_.x;                                                                   \
_Pragma("unroll")                                                                                                      \
                            for (int e = 0; e < 4; ++e) {                                                              \
                                const bool ok = row0 + ro + e < ntotal && CZ_ALLOWED(mask, row0 + ro + e);             \
                                svp_[e] = ok ? (I8 ? acc[m][n][4 * rg + e] * sqm_ : acc[m][n][4 * rg + e]) : -INFINITY; \
                                ivp_[e] = ok ? (uint32_t)(row0 + ro + e) : kInvalidRow;                                \
                            }                                                                                          \
                            *reinterpret_cast<float4*>(cand_s + qb + ro) = sv_;                                        \
                            *reinterpret_cast<uint4*>(cand_i + qb + ro) = iv_;                                         \
                        }                                                                                              \
                }                                                                                                      \
            } else {                                                                                                   \
                /* One wave-wide vote per query tile m keeps the usual case -- no hit -- at a compare per score.  A tile m  \
                   with hits builds each lane's bit mask of its TN * NR scores that reach the threshold (and lie inside    \
                   the index / the allow bitmap) and hands the hits on.  (Until round 2 every hit made its own returning   \
                   atomic, each waited for in turn: ~0.7 us per hit and wave, 100 us per tile in the early stages of the    \
                   cascade, whose thresholds are loose, and 2 us per tile in the main stage.) */                            \
                unsigned anym = 0u;                                                                                    \
                float thrv_[TM];   /* (kept for the hit path: a second LDS read there sits on the tile's critical path) */ \
_Pragma("unroll")                                                                                                      \
                for (int m = 0; m < TM; ++m) {                                                                         \
                    const float thr_q = sthr[wr * 128 + CZ_QOFF(m)];                                                   \
                    thrv_[m] = thr_q;                                                                                  \
                    float mx_ = acc[m][0][0];   /* (v_max3_f32 chain: 8 instructions for the 16 scores of a lane) */   \
_Pragma("unroll")                                                                                                      \
                    for (int e = 1; e < TN * NR; e += 2) {                                                             \
                        const float a_ = acc[m][e / NR][e % NR];                                                       \
                        const float b_ = e + 1 < TN * NR ? acc[m][(e + 1) / NR][(e + 1) % NR] : a_;                    \
                        mx_ = fmaxf(fmaxf(mx_, a_), b_);                                                               \
                    }                                                                                                  \
                    anym |= __ballot(mx_ >= thr_q) != 0ull ? 1u << m : 0u;                                             \
                }                                                                                                      \
                if (anym != 0u && !(dbg & 2)) {   /* dbg bit1 (timing experiments): votes only, no appends */           \
                    const bool edge = row0 + 64 > ntotal || mask != nullptr;   /* wave uniform */                      \
                    if constexpr (CZ_STAGED) {                                                                         \
                        /* k_scan_coarse8 (16x16 tiles: TM = 8, TN = 4, NR = 4).  The 8 waves of a block and the blocks     \
                           that share its row tiles move in lockstep, so a tile costs what its SLOWEST wave spends here:    \
                           the work per hit has to be small, not only the work per tile.  A query tile m with a hit builds   \
                           each lane's 16-bit mask and every lane walks its OWN hits (a per-lane bit picks the score out   \
                           of the 16 accumulator registers through a select tree), so a pass of the loop takes one hit     \
                           from every lane that has one: the loop runs max-hits-per-lane times (1..3), not once per score  \
                           position with a hit somewhere in the wave (~10 of 16 in the early stages, whose thresholds are  \
                           loose: 65 us of an 87 us stage at 1.25 M rows).  Hits go to this wave's LDS list (slot = wave   \
                           count + rank of the lane among the pass's hits), which is written to the candidate              \
                           buffers when it is nearly full and at the end of the kernel (CZ_FLUSH): no atomic round trip in \
                           the tile loop. */                                                                                  \
_Pragma("unroll")                                                                                                      \
                        for (int m = 0; m < TM; ++m) {                                                                 \
                            if (((anym >> m) & 1u) == 0u) continue;   /* wave uniform */                               \
                            const float thr_q = thrv_[m];                                                              \
                            unsigned h = 0u;                                                                           \
_Pragma("unroll")                                                                                                      \
                            for (int n = 0; n < TN; ++n)                                                               \
_Pragma("unroll")                                                                                                      \
                                for (int r = 0; r < NR; ++r) h |= (acc[m][n][r] >= thr_q ? 1u : 0u) << (n * NR + r);   \
                            const unsigned qv = (unsigned)(qtile * CZ_T + wr * 128 + MS * m + lq);                     \
_Pragma("nounroll")                                                                                                    \
                            while (true) {                                                                             \
                                const bool has = h != 0u;                                                              \
                                if (__ballot(has) == 0ull) break;                                                      \
                                if (wcount + 64 > CZ_WCAP) {                                                           \
                                    CZ_FLUSH();                                                                        \
                                }                                                                                      \
                                const int bit = has ? __builtin_ctz(h) : 0;                                            \
                                h &= h - 1u;                                                                           \
                                /* acc[m][bit >> 2][bit & 3] of a per-lane bit: a 4-level select tree (15 v_cndmask) */\
                                const bool s0_ = bit & 1, s1_ = bit & 2, s2_ = bit & 4, s3_ = bit & 8;                 \
                                const float t0_ = s0_ ? acc[m][0][1] : acc[m][0][0], t1_ = s0_ ? acc[m][0][3] : acc[m][0][2];\
                                const float t2_ = s0_ ? acc[m][1][1] : acc[m][1][0], t3_ = s0_ ? acc[m][1][3] : acc[m][1][2];\
                                const float t4_ = s0_ ? acc[m][2][1] : acc[m][2][0], t5_ = s0_ ? acc[m][2][3] : acc[m][2][2];\
                                const float t6_ = s0_ ? acc[m][3][1] : acc[m][3][0], t7_ = s0_ ? acc[m][3][3] : acc[m][3][2];\
                                const float u0_ = s1_ ? t1_ : t0_, u1_ = s1_ ? t3_ : t2_, u2_ = s1_ ? t5_ : t4_, u3_ = s1_ ? t7_ : t6_;\
                                const float w0_ = s2_ ? u1_ : u0_, w1_ = s2_ ? u3_ : u2_;                              \
                                const float v_ = s3_ ? w1_ : w0_;                                                      \
                                const int64_t row = row0 + 16 * (bit >> 2) + 4 * lg + (bit & 3);                       \
                                bool hit = has;                                                                        \
                                if (edge) hit = hit && row < ntotal && CZ_ALLOWED(mask, row);                          \
                                const unsigned long long b = __ballot(hit);                                            \
                                if (hit) {                                                                             \
                                    const unsigned sl = (unsigned)wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));\
                                    const unsigned ad = wl_base + sl * 4u;                                             \
                                    asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%4\n\tds_write_b32 %0, %3 offset:%5"\
                                                 : : "v"(ad), "v"(v_), "v"((unsigned)row), "v"(qv), "n"(CZ_WCAP * 4), "n"(CZ_WCAP * 8) : "memory");\
                                }                                                                                      \
                                wcount += __popcll(b);                                                                 \
                            }                                                                                          \
                        }                                                                                              \
                    } else {                                                                                           \
_Pragma("unroll")                                                                                                      \
                    for (int m = 0; m < TM; ++m) {                                                                     \
                        if (((anym >> m) & 1u) == 0u) continue;   /* wave uniform */                                   \
                        const float thr_q = sthr[wr * 128 + CZ_QOFF(m)];                                               \
                        unsigned h = 0u;                                                                               \
_Pragma("unroll")                                                                                                      \
                        for (int n = 0; n < TN; ++n)                                                                   \
_Pragma("unroll")                                                                                                      \
                            for (int r = 0; r < NR; ++r) h |= (acc[m][n][r] >= thr_q ? 1u : 0u) << (n * NR + r);       \
                        if (edge && h != 0u) {                                                                         \
_Pragma("unroll")                                                                                                      \
                            for (int n = 0; n < TN; ++n)                                                               \
_Pragma("unroll")                                                                                                      \
                                for (int r = 0; r < NR; ++r) {                                                         \
                                    const int64_t row = row0 + CZ_ROFF(n, r);                                          \
                                    if (((h >> (n * NR + r)) & 1u) && !(row < ntotal && CZ_ALLOWED(mask, row)))        \
                                        h &= ~(1u << (n * NR + r));                                                    \
                                }                                                                                      \
                        }                                                                                              \
                        if (h != 0u) {                                                                                 \
                            /* one returning atomic per lane and query tile reserves the slots of all its hits */      \
                            const unsigned qv = (unsigned)(qtile * CZ_T + wr * 128 + CZ_QOFF(m));                      \
                            int slot = atomicAdd(&cand_n[(size_t)(qv) * CZ_NS], __popc(h));                                              \
                            const size_t qb = (size_t)qv * KCAP;                                                     \
_Pragma("unroll")                                                                                                      \
                            for (int n = 0; n < TN; ++n)                                                               \
_Pragma("unroll")                                                                                                      \
                                for (int r = 0; r < NR; ++r) {                                                         \
                                    if ((h >> (n * NR + r)) & 1u) {                                                    \
                                        if (slot < KCAP) {                                                           \
                                            cand_s[qb + slot] = acc[m][n][r];                                          \
                                            cand_i[qb + slot] = (uint32_t)(row0 + CZ_ROFF(n, r));                      \
                                        }                                                                              \
                                        ++slot;                                                                        \
                                    }                                                                                  \
                                }                                                                                      \
                        }                                                                                              \
                    }                                                                                                  \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
_Pragma("unroll")                                                                                                      \
            for (int m = 0; m < TM; ++m)                                                                               \
_Pragma("unroll")                                                                                                      \
                for (int n = 0; n < TN; ++n)                                                                           \
_Pragma("unroll")                                                                                                      \
                    for (int r = 0; r < NR; ++r) acc[m][n][r] = 0.f;                                                   \
    do {} while (0)

// One stage of the cascade.  Grid = one block of 8 waves per CU (persistent).  Block -> (query tile, stream of
// row tiles): the nqt blocks that share a row tile sit on one XCD (blockIdx % 8) and walk side by side, so the
// tile's rows are fetched from HBM once per XCD L2.
// Wave grid 2 (query halves) x 4 (row quarters); a wave's 128 queries x 64 rows are 8 x 4 accumulator tiles of
// 16x16 (MS = 16, default: a lane owns 8 query columns x 16 rows) or 4 x 2 tiles of 32x32 (MS = 32).
// STAGE0: every score is written to slot (tile ordinal * 256 + row in tile); otherwise scores >= thr are
// appended.  MAIN only gives the last (stride 1, 3/4 of the rows) stage its own name in profiles.
template <bool STAGE0, bool MAIN, bool DBG = false, int MS = 32>
__global__ __launch_bounds__(512) void k_scan_coarse(const unsigned short* __restrict__ xh,
                                                     const unsigned short* __restrict__ qh,
                                                     const float* __restrict__ thr, float* __restrict__ cand_s,
                                                     uint32_t* __restrict__ cand_i, int* __restrict__ cand_n,
                                                     int64_t ntotal, int K, int nqt, int64_t count, int64_t stride,
                                                     int gm1, int* __restrict__ pace_cnt, const uint32_t* __restrict__ mask,
                                                     const float* __restrict__ xn2, int dbg_arg, const int* __restrict__ gate,
                                                     const float* __restrict__ xsc, const float* __restrict__ qsc) {
    constexpr int KCAP = CZ_CAP;   // candidate slots per query (the shared epilogue macros)
    constexpr bool I8 = false;     // (the int8 operands exist in k_scan_coarse8 only; xsc / qsc are null here)
    const float* const ssq = nullptr;
    const unsigned sxs_base = 0u;
    (void)xsc;
    (void)qsc;
    (void)ssq;
    (void)sxs_base;
    (void)gate;
    // dbg (CSS_KNN_DBG, timing experiments only, honoured by the DBG instantiation alone so that the product
    // kernel carries no such branches): bit0 skip the epilogue, bit1 skip MFMA + LDS reads, bit2 skip the
    // LDS-DMA loads, bit3 LDS reads without MFMAs, bit4 MFMAs without LDS reads, bit5 half of the LDS reads
    const int dbg = DBG ? dbg_arg : 0;
    static_assert(MS == 32 || MS == 16, "MFMA shape: 32x32x16 or 16x16x32");
    constexpr int NW = 8, WN = 4, TM = 128 / MS, TN = 64 / MS, NR = MS == 32 ? 16 : 4;
    typedef typename std::conditional<MS == 32, f32x16, v4f>::type acc_t;
    constexpr int A_BYTES = CZ_T * CZ_RB;
    constexpr int PPW = (2 * CZ_T / 8) / NW;  // 1-KiB LDS-DMA pieces (8 rows) per wave per stage = 8
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lq = MS == 32 ? (lane & 31) : (lane & 15), lg = MS == 32 ? (lane >> 5) : (lane >> 4);

    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int slots = per_x / nqt;
    if (jx >= slots * nqt) return;
    const int qtile = jx % nqt;
    const int64_t u0 = xcd + 8 * (jx / nqt), ustep = 8 * slots;
    const int my_ntiles = u0 < count ? (int)((count - u0 + ustep - 1) / ustep) : 0;
    const int KT = K / 64;
    const int total = my_ntiles * KT;
    if (total == 0) return;

    // the block's 256 thresholds wait in LDS (a separate object from the DMA ring) and are read in the epilogue:
    // keeping them in registers through the main loop pushed the DMA source pointers into scratch
    __shared__ float sthr[CZ_T];
    if (tid < CZ_T) sthr[tid] = STAGE0 ? -INFINITY : thr[qtile * CZ_T + tid];
    __syncthreads();
    constexpr bool CZ_STAGED = false;   // hits are appended straight from the epilogue (see k_scan_coarse8)
    int wcount = 0;
    const unsigned wl_base = 0u;
    (void)wcount;
    (void)wl_base;

    // Sibling pacing (speed only, never needed for correctness): the nqt blocks that walk the same row tiles
    // drift apart (appends, DMA jitter); once they are more than ~2 K-steps apart the tile's rows have left
    // the XCD's 4 MiB L2 (2 MiB stream through it per step) and every sibling fetches them again (measured
    // 2.6x the row bytes from HBM/MALL).  Each block announces "3 steps from the end of tile i" on a
    // per-group counter, reads it one step later (asynchronously) and, before issuing the first stage of tile
    // i+1, waits for its siblings -- a bounded spin, after which the block stops pacing for good.
    int* my_cnt = pace_cnt ? pace_cnt + xcd * slots + jx / nqt : nullptr;
    bool pace = my_cnt != nullptr && nqt > 1 && KT >= 4 && wave == 0;
    int seen = 0;

    // LDS-DMA: piece p (1 KiB = 8 rows x 128 B) of a stage; pieces 0..31 = query rows, 32..63 = index rows.
    // The swizzle is applied on the per-lane SOURCE address; the LDS destination is linear.
    const int prow = lane >> 3, pchunk = lane & 7;
    const char* src[PPW];
    int dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const bool isA = piece < 32;
        const int trow = (isA ? piece : piece - 32) * 8 + prow;
        dst[i] = (isA ? 0 : A_BYTES) + (isA ? piece : piece - 32) * 1024;
        if (isA) src[i] = reinterpret_cast<const char*>(qh + (size_t)(qtile * CZ_T + trow) * K) + ((pchunk ^ ((trow >> 1) & 7)) << 4);
    }
// (macros, not lambdas: a by-reference capture of src[] leaves the array in scratch once register pressure rises)
#define CZ_TILE_OF(TI_) ((STAGE0 ? (u0 + (int64_t)(TI_) * ustep) : (u0 + (int64_t)(TI_) * ustep) + (u0 + (int64_t)(TI_) * ustep) / gm1 + 1) * stride)
#define CZ_SET_SRC(TI_)                                                                                              \
    {                                                                                                                \
        const int64_t r0_ = CZ_TILE_OF(TI_) * CZ_T; /* multiples of the growth factor belong to earlier stages */     \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                            \
            const int piece = wave + NW * i;                                                                         \
            if (piece >= 32) {                                                                                       \
                const int trow = (piece - 32) * 8 + prow;                                                            \
                int64_t grow = r0_ + trow;                                                                           \
                grow = grow < ntotal ? grow : ntotal - 1;                                                            \
                src[i] = reinterpret_cast<const char*>(xh + (size_t)grow * K) + ((pchunk ^ ((trow >> 1) & 7)) << 4);  \
            }                                                                                                        \
        }                                                                                                            \
    }
#define CZ_ISSUE(KT_, SLOT_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                                \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(KT_) * CZ_RB), \
                                         (__attribute__((address_space(3))) void*)(smem + (SLOT_) * CZ_STAGE + dst[i]), 16, 0, 0); \
    }

    acc_t acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[m][n][r] = 0.f;

    int it_tile = 0, it_kt = 0, gi = 0;
    CZ_SET_SRC(0)
    if (!(dbg & 4)) {
        CZ_ISSUE(0, 0)
    }
    gi = 1;
    if (++it_kt == KT) {
        it_kt = 0;
        if (++it_tile < my_ntiles) CZ_SET_SRC(it_tile)
    }
    int ct_tile = 0, kt = 0;
    for (int g = 0; g < total; ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pace && ct_tile + 1 < my_ntiles) {
            if (kt == KT - 3) {
                if (lane == 0) __hip_atomic_fetch_add(my_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (kt == KT - 2) {
                if (lane == 0) seen = __hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (kt == KT - 1) {
                const int target = nqt * (ct_tile + 1);
                int spins = 0;
                int v = __builtin_amdgcn_readfirstlane(seen);
                while (v < target && spins < 64) {
                    __builtin_amdgcn_s_sleep(8);
                    v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    ++spins;
                }
                if (v < target) pace = false;  // siblings not co-resident (or far behind): stop waiting for them
            }
        }
        __builtin_amdgcn_s_barrier();  // stage g landed for every wave; the slot of stage g-1 is free
        const char* Ab = smem + (g & 1) * CZ_STAGE;
        const char* Bb = Ab + A_BYTES;
// the DMA of stage g+1 (into the other slot); ISSUE_FIRST: before this step's fragment reads, else after the first ones
#define CZ_ISSUE_NEXT()                                                  \
        if (gi < total) {                                                \
            if (!(dbg & 4)) {                                            \
                CZ_ISSUE(it_kt, gi & 1)                                  \
            }                                                            \
            ++gi;                                                        \
            if (++it_kt == KT) {                                         \
                it_kt = 0;                                               \
                if (++it_tile < my_ntiles) CZ_SET_SRC(it_tile)           \
            }                                                            \
        }
        if constexpr (DBG) {
            CZ_ISSUE_NEXT()
        }
        // C_: 16-wide k step 0..3 (MS = 32: chunk 2 C_ + lg) or 32-wide k step 0..1 (MS = 16: chunk 4 C_ + lg)
#define CZ_READ(A_, B_, C_)                                                                                            \
    _Pragma("unroll") for (int m = 0; m < TM; ++m)                                                                    \
        A_[m] = *reinterpret_cast<const v4f*>(Ab + cz_swz(wr * 128 + MS * m + lq, (MS == 32 ? 2 : 4) * (C_) + lg));   \
    _Pragma("unroll") for (int n = 0; n < TN; ++n)                                                                    \
        B_[n] = *reinterpret_cast<const v4f*>(Bb + cz_swz(wc * 64 + MS * n + lq, (MS == 32 ? 2 : 4) * (C_) + lg));
#define CZ_MFMA(A_, B_)                                                                                                \
    _Pragma("unroll") for (int m = 0; m < TM; ++m)                                                                    \
        _Pragma("unroll") for (int n = 0; n < TN; ++n) {                                                              \
            if constexpr (MS == 32)                                                                                    \
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, B_[n]),                  \
                                                                    __builtin_bit_cast(v8bf, A_[m]), acc[m][n], 0, 0, 0); \
            else                                                                                                       \
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, B_[n]),                  \
                                                                    __builtin_bit_cast(v8bf, A_[m]), acc[m][n], 0, 0, 0); \
        }
        // MFMA rows <- index rows, MFMA columns <- queries: lane (fr, fh) holds query fr and index rows
        // (r&3) + 8(r>>2) + 4 fh of the 32-row tile in register r
        if constexpr (!DBG) {
            // (explicitly double-buffered fragment reads pinned with sched_barrier measured 2 % slower than
            // hipcc's own read / wait / 4-MFMA groups: LDS latency is not what bounds this loop)
            // the first fragment reads go out right behind the barrier, the 8 DMA instructions of the next stage
            // (slow to issue) follow while those reads are in flight
            {   // (first k-step in two halves of the query tiles: fewer fragments live across the DMA issue)
                constexpr int CM = (MS == 32 ? 2 : 4);  // chunk index multiplier of this MFMA shape
                v4f b[TN], a[TM / 2];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const v4f*>(Bb + cz_swz(wc * 64 + MS * n + lq, lg));
#pragma unroll
                for (int m = 0; m < TM / 2; ++m) a[m] = *reinterpret_cast<const v4f*>(Ab + cz_swz(wr * 128 + MS * m + lq, lg));
                __builtin_amdgcn_sched_barrier(0);
                CZ_ISSUE_NEXT()
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1) {
#pragma unroll
                        for (int m = 0; m < TM / 2; ++m)
                            a[m] = *reinterpret_cast<const v4f*>(Ab + cz_swz(wr * 128 + MS * (TM / 2 + m) + lq, lg));
                    }
#pragma unroll
                    for (int m = 0; m < TM / 2; ++m)
#pragma unroll
                        for (int n = 0; n < TN; ++n) {
                            if constexpr (MS == 32)
                                acc[h * (TM / 2) + m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                    __builtin_bit_cast(v8bf, b[n]), __builtin_bit_cast(v8bf, a[m]), acc[h * (TM / 2) + m][n], 0, 0, 0);
                            else
                                acc[h * (TM / 2) + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(v8bf, b[n]), __builtin_bit_cast(v8bf, a[m]), acc[h * (TM / 2) + m][n], 0, 0, 0);
                        }
                }
                (void)CM;
            }
#pragma unroll
            for (int c = 1; c < (MS == 32 ? 4 : 2); ++c) {
                v4f a[TM], b[TN];
                CZ_READ(a, b, c)
                CZ_MFMA(a, b)
            }
        } else if (!(dbg & 2)) {
            v4f a[TM], b[TN];
#pragma unroll
            for (int c = 0; c < (MS == 32 ? 4 : 2); ++c) {
                if ((dbg & 32) && (c & 1)) {
                    // bit5: odd k-steps reuse the previous fragments (half the LDS read traffic; wrong results)
                } else if (!(dbg & 16)) {
                    CZ_READ(a, b, c)
                } else {
#pragma unroll
                    for (int m = 0; m < TM; ++m) a[m] = v4f{1.f, 2.f, 3.f, 4.f};
#pragma unroll
                    for (int n = 0; n < TN; ++n) b[n] = v4f{1.f, 2.f, 3.f, 4.f};
                }
                if (dbg & 8) {
#pragma unroll
                    for (int m = 0; m < TM; ++m) asm volatile("" ::"v"(a[m]));
#pragma unroll
                    for (int n = 0; n < TN; ++n) asm volatile("" ::"v"(b[n]));
                } else {
                    CZ_MFMA(a, b)
                }
            }
        }
#undef CZ_READ
#undef CZ_MFMA
#undef CZ_ISSUE_NEXT
        if (++kt == KT) {
            CZ_EPILOGUE();
            kt = 0;
            ++ct_tile;
        }
    }
#undef CZ_ISSUE
#undef CZ_SET_SRC
#undef CZ_TILE_OF
}

// ---------------------------------------------------------------------------------------------------------------
// k_scan_coarse8: the same stage with the 8-phase ping-pong main loop of the encoder's k_gemm8p (css_encoder_kernels.h
// has the full description): a 64-deep K step is four phases of 16 MFMAs, each [fragment reads + 2 LDS-DMA
// instructions] s_barrier [MFMAs] s_barrier; wave row 1 runs one barrier behind wave row 0, so one wave of every SIMD
// issues MFMAs while its partner reads and issues DMA; half-tile slots (A = queries, B = index rows) are refilled
// three ahead behind a counted vmcnt(6).  Accumulator geometry, thresholds, candidate appends and sibling pacing
// are those of k_scan_coarse<.., 16> (the epilogue macro is shared).  Needs K % 128 == 0 and 256 slack rows behind
// the shadow rows (the last tile reads them; their scores are masked by row < ntotal).
// Measured (10 M x 768, 1000 queries, main stage): see DESIGN.md section 3.
// one MFMA of the 8-phase loop: bf16 16x16x32, or int8 16x16x64 with the int32 accumulators kept in the v4f registers
template <bool I8>
__device__ __forceinline__ v4f c8_mfma(v4f b, v4f a, v4f c) {
    if constexpr (I8) {
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(v4f, __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(v4i_t, b), __builtin_bit_cast(v4i_t, a),
                                                                              __builtin_bit_cast(v4i_t, c), 0, 0, 0));
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, b), __builtin_bit_cast(v8bf, a), c, 0, 0, 0);
    }
}
constexpr int C8_HT = 16384;
constexpr int C8_A0 = 0, C8_B0 = 1, C8_B1 = 2, C8_A1 = 3;

// KCAP_T: candidate slots per query (CZ_CAP in the cascade; the second pass over flagged queries has larger buffers).
// gate (second pass only, else null): number of queries that take part -- the launch is enqueued before that
// number is known, so blocks whose query tile lies beyond it (all of them when it is 0) return at once.
// I8: the operands are the int8 shadow rows (css_index: signed byte = round(x / s), per-row scale s in xsc) and int8
// queries (k_rows_to_i8, per-query scale in qsc) on v_mfma_i32_16x16x64_i8 -- the same bytes per K step as bf16 carry
// twice the multiply-adds, a 768-element row is 6 K steps instead of 12.  The integer accumulators (exact; |.| <
// 2^24 for rows of up to 1024 elements) are converted once per tile, score = acc * s_row * s_query: the row scale is
// applied in the epilogue, the query scale is folded into the thresholds (thr / s_query) and into the scores that
// leave the kernel.  Inner product only (the L2 form 2 x.q - ||x||^2 does not factor).
template <bool STAGE0, bool MAIN, bool DBG = false, int KCAP_T = CZ_CAP, bool I8 = false>
__global__ __launch_bounds__(512) void k_scan_coarse8(const unsigned short* __restrict__ xh,
                                                      const unsigned short* __restrict__ qh,
                                                      const float* __restrict__ thr, float* __restrict__ cand_s,
                                                      uint32_t* __restrict__ cand_i, int* __restrict__ cand_n,
                                                      int64_t ntotal, int K, int nqt_arg, int64_t count, int64_t stride,
                                                      int gm1, int* __restrict__ pace_cnt, const uint32_t* __restrict__ mask,
                                                      const float* __restrict__ xn2, int dbg_arg, const int* __restrict__ gate,
                                                      const float* __restrict__ xsc, const float* __restrict__ qsc) {
    constexpr int MS = 16, TM = 8, TN = 4, NR = 4;
    constexpr int KCAP = KCAP_T;
    const int dbg = DBG ? dbg_arg : 0;   // CSS_KNN_DBG (timing experiments): bit0 skips the epilogue
    (void)MAIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][4][C8_HT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lq = lane & 15, lg = lane >> 4;

    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    // (second pass: the launch is sized for nqt_arg query tiles, the flagged queries fill the first ceil(*gate / 256))
    int nqt = nqt_arg;
    if (gate != nullptr) {
        const int g = *gate;
        if (g <= 0) return;
        nqt = min(nqt_arg, (g + CZ_T - 1) / CZ_T);
    }
    const int slots = per_x / nqt;
    if (jx >= slots * nqt) return;
    const int qtile = jx % nqt;
    const int64_t u0 = xcd + 8 * (jx / nqt), ustep = 8 * slots;
    const int my_ntiles = u0 < count ? (int)((count - u0 + ustep - 1) / ustep) : 0;
    const unsigned ROWB = I8 ? (unsigned)K : (unsigned)K * 2u;   // bytes per operand row
    const int KT = (int)(ROWB / 128u);     // K steps of 128 B; even (host check)
    const int total = my_ntiles * KT;
    if (total == 0) return;

    __shared__ float sthr[CZ_T];
    __shared__ float ssq[I8 ? CZ_T : 1];   // I8: the query scales of this query tile
    __shared__ __attribute__((aligned(16))) float sxs[I8 ? 8 * 64 : 4];   // I8: per wave, the scales of its 64 rows of the current tile
    const unsigned sxs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&sxs[I8 ? wave * 64 : 0];
    (void)sxs_base;
    __shared__ int space[4];   // landing word of the sibling-pacing counter read
    __shared__ float wl[8][3][CZ_WCAP];   // per-wave hit lists: [score | row | query] (CZ_FLUSH)
    if (tid < CZ_T) {
        float t_ = STAGE0 ? -INFINITY : thr[qtile * CZ_T + tid];
        if constexpr (I8) {
            const float sq_ = qsc[qtile * CZ_T + tid];
            ssq[tid] = sq_;
            t_ = t_ / sq_;   // (+-inf stay; scales are > 0)
        }
        sthr[tid] = t_;
    }
    __syncthreads();
    constexpr bool CZ_STAGED = !STAGE0;
    int wcount = 0;                       // entries in this wave's list (wave uniform)
    const unsigned wl_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&wl[wave][0][0];

    int* my_cnt = pace_cnt ? pace_cnt + xcd * slots + jx / nqt : nullptr;
    bool pace = my_cnt != nullptr && nqt > 1 && KT >= 4 && wave == 0;
    int seen = 0;

    // tile of ordinal u: STAGE0: u * stride; else (u + u / gm1 + 1) * stride (multiples of the growth factor belong to
    // earlier stages).  u advances by ustep per tile; quotient and remainder by gm1 are carried along, so the loop
    // holds no division.  cur_tile: the tile being computed (what the shared epilogue macro asks for).
    // gate != null (second pass): ALL tiles in order, tile = u (the "+ 1" that skips the earlier stages' tiles is undone)
    const int64_t tbias = gate != nullptr ? -1 : 0;
    const int64_t tile0 = (STAGE0 ? u0 : u0 + u0 / gm1 + 1) * stride + (STAGE0 ? 0 : tbias);
    int64_t cur_tile = tile0;
    int ucmp = (int)u0, uqc = (int)(u0 / gm1), urc = (int)(u0 % gm1);
#define CZ_TILE_OF(TI_) (cur_tile)
    // ---- DMA bookkeeping.  Queries (A kinds): fixed per-lane 32-bit offsets, only the K step advances.  Index rows
    // (B kinds): fixed per-lane offset inside a tile + a 64-bit uniform tile base that advances with the tile.
    const int prow = lane >> 3, pchunk = lane & 7;
    unsigned lofs[4];       // per-lane byte offset of the first piece
    size_t tbase[4];        // uniform byte offset of the kind's current row tile (B kinds)
    const int sq = (int)(ustep / gm1), sr = (int)(ustep % gm1);
    int ucur[4], uq[4], ur[4];
    int it_tile[4], it_kt[4];
    int dsto[4];
    dsto[C8_A0] = dsto[C8_A1] = (wr * 64 + wc * 16) * 128;
    dsto[C8_B0] = dsto[C8_B1] = (16 * wave) * 128;
    const unsigned row8 = 8u * ROWB;
    {
        const int rrA = wc * 16 + prow, srowA = wr * 64 + rrA;
        const unsigned swA = (unsigned)((pchunk ^ ((srowA >> 1) & 7)) << 4);
        lofs[C8_A0] = (unsigned)(qtile * CZ_T + wr * 128 + rrA) * ROWB + swA;
        lofs[C8_A1] = (unsigned)(qtile * CZ_T + wr * 128 + 64 + rrA) * ROWB + swA;
        const int srowB = 16 * wave + prow;
        const unsigned swB = (unsigned)((pchunk ^ ((srowB >> 1) & 7)) << 4);
        lofs[C8_B0] = (unsigned)((srowB >> 5) * 64 + (srowB & 31)) * ROWB + swB;
        lofs[C8_B1] = (unsigned)((srowB >> 5) * 64 + 32 + (srowB & 31)) * ROWB + swB;
    }
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
        it_tile[kd] = 0;
        it_kt[kd] = 0;
        ucur[kd] = (int)u0;
        uq[kd] = (int)(u0 / gm1);
        ur[kd] = (int)(u0 % gm1);
        tbase[kd] = (size_t)tile0 * CZ_T * (size_t)ROWB;
    }
#ifndef C8_ROW_AUX
#define C8_ROW_AUX 0   // cache policy of the row-tile LDS-DMA (2 = non-temporal: measured)
#endif
#define C8_ISSUE(KIND_, DB_)                                                                                  \
    {                                                                                                         \
        const bool isA_ = (KIND_) == C8_A0 || (KIND_) == C8_A1;                                               \
        const char* base_ = isA_ ? reinterpret_cast<const char*>(qh) : reinterpret_cast<const char*>(xh) + tbase[KIND_]; \
        const unsigned o0_ = lofs[KIND_] + (unsigned)it_kt[KIND_] * 128u;                                     \
        const unsigned o1_ = (o0_ + row8) ^ 64u;                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_ + o0_),        \
            (__attribute__((address_space(3))) void*)(smem + ((DB_) * 4 + (KIND_)) * C8_HT + dsto[KIND_]), 16, 0, \
            ((KIND_) == C8_A0 || (KIND_) == C8_A1) ? 0 : C8_ROW_AUX);                                          \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_ + o1_),        \
            (__attribute__((address_space(3))) void*)(smem + ((DB_) * 4 + (KIND_)) * C8_HT + dsto[KIND_] + 1024), 16, 0, \
            ((KIND_) == C8_A0 || (KIND_) == C8_A1) ? 0 : C8_ROW_AUX);                                          \
        if (++it_kt[KIND_] == KT) {                                                                           \
            it_kt[KIND_] = 0;                                                                                 \
            if (!isA_ && it_tile[KIND_] + 1 < my_ntiles) {                                                    \
                ++it_tile[KIND_];                                                                             \
                ucur[KIND_] += (int)ustep;                                                                    \
                uq[KIND_] += sq;                                                                              \
                ur[KIND_] += sr;                                                                              \
                if (ur[KIND_] >= gm1) {                                                                       \
                    ur[KIND_] -= gm1;                                                                         \
                    ++uq[KIND_];                                                                              \
                }                                                                                             \
                const int64_t t_ = STAGE0 ? (int64_t)ucur[KIND_] * stride : ((int64_t)ucur[KIND_] + uq[KIND_] + 1) * stride + tbias; \
                tbase[KIND_] = (size_t)t_ * CZ_T * (size_t)ROWB;                                              \
            }                                                                                                 \
        }                                                                                                     \
    }
    typedef v4f acc_t;
    acc_t acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};

// I8: the scales of this wave's 64 rows of tile TILE_ travel into sxs by one 4-byte-per-lane LDS-DMA, issued before the
// tile's K loop: older than every ring DMA of the loop, so the loop's counted vmcnt waits cover it long before the
// epilogue reads it (fetched in the epilogue itself the loads cost ~3 us per tile of exposed latency: 9.8 vs 8.5 ms
// per 10 M-row batch).  (Rows beyond ntotal: the slack rows of the scale array.)
#define C8_SCALE_ISSUE(TILE_)                                                                                 \
    if constexpr (I8) {                                                                                       \
        __builtin_amdgcn_global_load_lds(                                                                     \
            (const __attribute__((address_space(1))) void*)(xsc + (size_t)(TILE_) * CZ_T + wc * 64 + lane),   \
            (__attribute__((address_space(3))) void*)&sxs[wave * 64], 4, 0, 0);                               \
    }
    C8_SCALE_ISSUE(tile0)
    C8_ISSUE(C8_A0, 0)
    C8_ISSUE(C8_B0, 0)
    C8_ISSUE(C8_B1, 0)
    C8_ISSUE(C8_A1, 0)
    C8_ISSUE(C8_A0, 1)
    C8_ISSUE(C8_B0, 1)
    C8_ISSUE(C8_B1, 1)
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0

    const int a_o0 = cz_swz(wr * 64 + lq, lg), a_o1 = a_o0 ^ 64;
    const int b_o0 = cz_swz(wc * 32 + lq, lg), b_o1 = b_o0 ^ 64;
    v4f a[4][2], b0[2][2], b1[2][2];
#define C8_READ_A(S_, D_)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
        a[j][0] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? C8_A1 : C8_A0)) * C8_HT + j * 2048 + a_o0); \
        a[j][1] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? C8_A1 : C8_A0)) * C8_HT + j * 2048 + a_o1); \
    }
#define C8_READ_B(S_, D_, B_)                                                                                 \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                           \
        B_[n][0] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? C8_B1 : C8_B0)) * C8_HT + n * 2048 + b_o0); \
        B_[n][1] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? C8_B1 : C8_B0)) * C8_HT + n * 2048 + b_o1); \
    }
// MFMA rows <- index rows, MFMA columns <- queries: lane (lq, lg) holds query 16 m + lq and rows 16 n + 4 lg + r
#define C8_MFMA(MH_, NH_, B_)                                                                                 \
    {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                 \
                    acc[4 * (MH_) + j][2 * (NH_) + n] = c8_mfma<I8>(B_[n][c], a[j][c], acc[4 * (MH_) + j][2 * (NH_) + n]); \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    }
#define C8_SYNC_A()                                   \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
#define C8_SYNC_B()                    \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0);
    int ct_tile = 0, kt = 0;
// sibling pacing (k_scan_coarse has the rationale): announce three K steps before the tile ends, read the counter
// one step later, wait (bounded) for the siblings before the tile's last K step
#define C8_PACE()                                                                                             \
    if (pace && ct_tile + 1 < my_ntiles) {                                                                    \
        if (kt == KT - 3) {                                                                                   \
            if (lane == 0) __hip_atomic_fetch_add(my_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     \
        } else if (kt == KT - 2) {                                                                            \
            /* the counter travels by a 4-byte LDS-DMA (sc1: served by L2) into space[]: no VGPR destination, so neither \
               the compiler (vmcnt(0) before a visible load's use) nor a late register write can hurt; it has landed   \
               behind the counted vmcnt(6) of this K step's P4 (eight younger DMA instructions follow it) */            \
            if (lane == 0)                                                                                    \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)my_cnt,       \
                                                 (__attribute__((address_space(3))) void*)&space[0], 4, 0, 16);  \
        } else if (kt == KT - 1) {                                                                            \
            {                                                                                                 \
                const unsigned so_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)&space[0];    \
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(so_) : "memory");   \
            }                                                                            \
            const int target = nqt * (ct_tile + 1);                                                           \
            int spins = 0;                                                                                    \
            int v = __builtin_amdgcn_readfirstlane(seen);                                                     \
            while (v < target && spins < 64) {                                                                \
                __builtin_amdgcn_s_sleep(8);                                                                  \
                v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); \
                ++spins;                                                                                      \
            }                                                                                                 \
            if (v < target) pace = false;                                                                     \
        }                                                                                                     \
    }
#define C8_KSTEP(D_)                                                                                          \
    {                                                                                                         \
        C8_PACE()                                                                                             \
        /* P1 */                                                                                              \
        C8_READ_B(0, D_, b0)                                                                                  \
        C8_READ_A(0, D_)                                                                                      \
        C8_ISSUE(C8_A1, (D_) ^ 1)                                                                             \
        C8_SYNC_A()                                                                                           \
        C8_MFMA(0, 0, b0)                                                                                     \
        C8_SYNC_B()                                                                                           \
        /* P2 */                                                                                              \
        C8_READ_B(1, D_, b1)                                                                                  \
        C8_ISSUE(C8_A0, D_)                                                                                   \
        C8_SYNC_A()                                                                                           \
        C8_MFMA(0, 1, b1)                                                                                     \
        C8_SYNC_B()                                                                                           \
        /* P3 */                                                                                              \
        C8_READ_A(1, D_)                                                                                      \
        C8_ISSUE(C8_B0, D_)                                                                                   \
        C8_SYNC_A()                                                                                           \
        C8_MFMA(1, 1, b1)                                                                                     \
        C8_SYNC_B()                                                                                           \
        /* P4 */                                                                                              \
        C8_ISSUE(C8_B1, D_)                                                                                   \
        if (g + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        C8_SYNC_A()                                                                                           \
        C8_MFMA(1, 0, b0)                                                                                     \
        C8_SYNC_B()                                                                                           \
        ++g;                                                                                                  \
        ++kt;                                                                                                 \
    }
    for (int g = 0; g < total;) {
        C8_KSTEP(0)
        C8_KSTEP(1)
        if (kt == KT) {
            CZ_EPILOGUE();
            kt = 0;
            ++ct_tile;
            ucmp += (int)ustep;
            uqc += sq;
            urc += sr;
            if (urc >= gm1) {
                urc -= gm1;
                ++uqc;
            }
            cur_tile = STAGE0 ? (int64_t)ucmp * stride : ((int64_t)ucmp + uqc + 1) * stride + tbias;
            if (ct_tile < my_ntiles) C8_SCALE_ISSUE(cur_tile)
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger barrier of wave row 1
    if constexpr (CZ_STAGED) {
        if (wcount > 0) CZ_FLUSH();
    }
#undef C8_KSTEP
#undef C8_SCALE_ISSUE
#undef C8_PACE
#undef C8_SYNC_A
#undef C8_SYNC_B
#undef C8_MFMA
#undef C8_READ_A
#undef C8_READ_B
#undef C8_ISSUE
#undef CZ_TILE_OF
}

// ------------------------------------------------------------------ k_scan_qreg_i8: the int8 scan with the queries in registers
// k_scan_coarse8 streams BOTH operands of every 256 x 256 tile through LDS: 24 fragment reads of 1 KiB per 64 MFMAs and
// wave, ~96 of the CU's 128 B/clk at the full MFMA rate, and runs at ~45 % of the int8 peak.  Here the queries never
// move: a wave keeps 64 of them -- all K columns, 4 KS fragments of 16 B per lane: 192 registers at K = 768 -- for its whole
// life (the first 32 fragments in AGPRs, which v_mfma reads as its second operand directly; hipcc splits the 256
// registers of a wave 128 / 128 at two waves per SIMD, so the rest sit in VGPRs), and only the index rows pass through
// LDS: a ring of QR_RING groups of 16 rows (LDS-DMA, the XOR-swizzled 128-byte-chunk layout of the other scan kernels), each
// read by the 4 waves of the block -- KS fragment reads per 4 KS MFMAs and wave, a sixth of the LDS traffic.  Two 4-wave
// blocks per CU (a block = 256 queries x a stream of row tiles, mapped to XCDs like k_scan_coarse8); one block barrier per
// group; no operand ever waits in a register for a compiler-inserted vmcnt(0): query loads, DMA and LDS reads of the loop
// are inline asm.  tools/scan_lab.hip is the loop alone (5 M rows x 1024 queries: 2.75-2.85 ms = 2.8 POPS where
// k_scan_coarse8's main stage takes 3.4 ms for 1000 queries); ring depth (3..6 groups) and groups per barrier (1, 2) measured
// the same there -- the loop runs at the rate the chip's clock under a dense int8 MFMA load allows.
// Stages after the first only (stage 0 keeps every score: k_scan_coarse8<true, ..>); inner product; K = 256, 512 or 768.
// Epilogue per group: score = acc * row scale (units of the query scale, as in k_scan_coarse8) against thr / query scale; a
// wave-wide vote per 16 queries keeps the usual case at a convert, a multiply and a compare per score; hits go through
// the wave's LDS list and CZ_FLUSH.  (The MFMAs being asm, the hazard recogniser does not see them: explicit s_nop before
// the first VALU read of an accumulator.)
constexpr int QR_RING = 4;
#ifndef QR_SETPRIO
#define QR_SETPRIO 1
#endif
template <int KS>
constexpr size_t qr_lds_bytes() { return (size_t)QR_RING * (KS / 2) * 2048; }
template <int KS, bool MAIN>
__global__ __launch_bounds__(256, 2) void k_scan_qreg_i8(const unsigned char* __restrict__ x8, const signed char* __restrict__ q8,
                                                         const float* __restrict__ thr, float* __restrict__ cand_s,
                                                         uint32_t* __restrict__ cand_i, int* __restrict__ cand_n,
                                                         int64_t ntotal, int nqt, int64_t count, int64_t stride, int gm1,
                                                         const uint32_t* __restrict__ mask, const float* __restrict__ xsc,
                                                         const float* __restrict__ qsc) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    static_assert(KS == 4 || KS == 8 || KS == 12, "K = 256, 512 or 768 int8 columns");
    constexpr bool I8 = true;          // (CZ_FLUSH: list scores are in units of the query scale)
    constexpr int KCAP = CZ_CAP;
    constexpr int ROWB = 64 * KS;      // bytes per row
    constexpr int GSLOT = (KS / 2) * 2048;   // a group in LDS: [KS / 2 chunks of 128 B][16 rows][128 B]
    constexpr int NPW = KS / 4;        // DMA pieces (1 KiB: 8 rows x 128 B) per wave and group
    constexpr int NA = 4 * KS < 32 ? 4 * KS : 32;   // query fragments held in AGPRs
    (void)MAIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ring
    __shared__ __attribute__((aligned(16))) float sxs[2][CZ_T];   // row scales of the tile being computed / the next one
    __shared__ float wl[4][3][CZ_WCAP];                            // per-wave hit lists: [score | row | query] (CZ_FLUSH)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int slots = per_x / nqt;
    if (jx >= slots * nqt) return;
    const int qtile = jx % nqt;
    const int64_t u0 = xcd + 8 * (jx / nqt), ustep = 8 * slots;
    const int my_ntiles = u0 < count ? (int)((count - u0 + ustep - 1) / ustep) : 0;
    const int nsteps = my_ntiles * 16;
    if (nsteps == 0) return;
    const int qbase = qtile * CZ_T + wave * 64;

    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned sxs_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&sxs[0][0];
    const unsigned wl_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&wl[wave][0][0];
    int wcount = 0;   // entries in this wave's hit list (wave uniform)

    // DMA: piece p of a group = chunk p >> 1, rows 8 (p & 1) .. + 7; this wave issues pieces NPW wave .. + NPW - 1
    const int prow = lane >> 3, pchunk = lane & 7;
    unsigned lofs[NPW], ldst[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = NPW * wave + i, kc = p >> 1, srow = 8 * (p & 1) + prow;
        lofs[i] = (unsigned)srow * ROWB + (unsigned)kc * 128u + (unsigned)((pchunk ^ ((srow >> 1) & 7)) << 4);
        ldst[i] = (unsigned)(kc * 2048 + (p & 1) * 1024);
    }
    const unsigned sc_lofs = (unsigned)(wave * 64 + lane) * 4u;   // scale DMA: 4 bytes per lane, this wave's 64 rows of a tile
    // tile of ordinal u: (u + u / gm1 + 1) * stride (multiples of the growth factor belong to earlier stages); u advances by
    // ustep per tile, quotient and remainder by gm1 are carried along
    const int sq = (int)(ustep / gm1), sr = (int)(ustep % gm1);
    int iu = (int)u0, iuq = (int)(u0 / gm1), iur = (int)(u0 % gm1), itile_n = 0;   // issue side
    const char* ibase = reinterpret_cast<const char*>(x8) + (size_t)(((int64_t)iu + iuq + 1) * stride) * CZ_T * ROWB;
    int is_step = 0;
// (s_mov of a compiler-computed operand into m0 inside the statement: see G4_DMA in css_encoder_kernels.h)
#define QR_DMA16(SBASE_, VOFF_, LDS_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF_), "s"(SBASE_), "s"(LDS_) : "memory")
#define QR_DMA4(SBASE_, VOFF_, LDS_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(VOFF_), "s"(SBASE_), "s"(LDS_) : "memory")
// group is_step -> ring slot is_step % QR_RING; past the end the last group once more, into a free slot (the counted
// vmcnt below stays uniform).  A tile's row scales travel in front of its first group (one more, older, operation: the
// counted waits only get stricter).
#define QR_ISSUE()                                                                                                     \
    {                                                                                                                  \
        if (is_step < nsteps) {                                                                                        \
            const int g_ = is_step & 15;                                                                               \
            if (g_ == 0) {                                                                                             \
                if (is_step > 0) {                                                                                     \
                    iu += (int)ustep;                                                                                  \
                    iuq += sq;                                                                                         \
                    iur += sr;                                                                                         \
                    if (iur >= gm1) {                                                                                  \
                        iur -= gm1;                                                                                    \
                        ++iuq;                                                                                         \
                    }                                                                                                  \
                    ++itile_n;                                                                                         \
                }                                                                                                      \
                const int64_t t_ = ((int64_t)iu + iuq + 1) * stride;                                                   \
                ibase = reinterpret_cast<const char*>(x8) + (size_t)t_ * CZ_T * ROWB;                                  \
                QR_DMA4(xsc + (size_t)t_ * CZ_T, sc_lofs, sxs_base + (unsigned)(itile_n & 1) * (CZ_T * 4u) + (unsigned)wave * 256u); \
            } else {                                                                                                   \
                ibase += 16 * ROWB;                                                                                    \
            }                                                                                                          \
        }                                                                                                              \
        const unsigned dst_ = smem_base + (unsigned)(is_step % QR_RING) * GSLOT;                                       \
        _Pragma("unroll") for (int i_ = 0; i_ < NPW; ++i_) QR_DMA16(ibase, lofs[i_], dst_ + ldst[i_]);                 \
        ++is_step;                                                                                                     \
    }
#pragma unroll 1
    for (int i = 0; i < QR_RING - 1; ++i) QR_ISSUE()

    // resident query fragments: query qbase + 16 j + lq, bytes 64 t + 16 lg .. + 15 (asm loads: hipcc would wait for its own
    // loads at their first use INSIDE the loop, with a vmcnt that drains the DMA ring).  Behind the ring's first DMAs: one
    // memory latency in front of the loop instead of two (the later stages of a 1 M-row index are a few tiles per block).
    v4i_t qf[4][KS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const signed char* src = q8 + (size_t)(qbase + 16 * j + lq) * ROWB + 64 * t + 16 * lg;
            if (j * KS + t < NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[j][t]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[j][t]) : "v"(src) : "memory");
        }
    float thr_q[4];   // thresholds in units of the query scale (+-inf stay; scales are > 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) thr_q[j] = thr[qbase + 16 * j + lq] / qsc[qbase + 16 * j + lq];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const unsigned a_o0 = smem_base + (unsigned)cz_swz(lq, lg), a_o1 = a_o0 ^ 64u;
    int cu = (int)u0, cuq = (int)(u0 / gm1), cur_ = (int)(u0 % gm1), ctile_n = 0;   // compute side
    int64_t cur_tile = ((int64_t)cu + cuq + 1) * stride;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int grp = s & 15;
        if (grp == 0 && s > 0) {
            cu += (int)ustep;
            cuq += sq;
            cur_ += sr;
            if (cur_ >= gm1) {
                cur_ -= gm1;
                ++cuq;
            }
            ++ctile_n;
            cur_tile = ((int64_t)cu + cuq + 1) * stride;
        }
        // own pieces of group s have landed when at most the NPW (QR_RING - 2) youngest DMA instructions are outstanding
        static_assert(QR_RING == 4, "counted waits below");
        if constexpr (NPW == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (NPW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everybody's pieces of group s are in LDS; everybody is done with group s - 1
        QR_ISSUE()                      // group s + QR_RING - 1 -> the slot of group s - 1
        const unsigned slot = (unsigned)(s % QR_RING) * GSLOT;
        v4i_t acc[4];
        v4i_t a[4];   // fragment reads run three K steps ahead of the MFMAs
#define QR_LD(D_, T_) asm volatile("ds_read_b128 %0, %1" : "=v"(D_) : "v"(((T_) & 1 ? a_o1 : a_o0) + slot + (unsigned)((T_) >> 1) * 2048u) : "memory")
#define QR_MFMA0(J_, T_)                                                                                             \
    if ((J_) * KS + (T_) < NA) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc[J_]) : "v"(a[(T_) & 3]), "a"(qf[J_][T_])); \
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc[J_]) : "v"(a[(T_) & 3]), "v"(qf[J_][T_]));
#define QR_MFMA(J_, T_)                                                                                              \
    if ((J_) * KS + (T_) < NA) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[J_]) : "v"(a[(T_) & 3]), "a"(qf[J_][T_])); \
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[J_]) : "v"(a[(T_) & 3]), "v"(qf[J_][T_]));
        QR_LD(a[0], 0);
        QR_LD(a[1], 1);
        QR_LD(a[2], 2);
#if QR_SETPRIO
        // MFMAs of this wave go in front of the partner wave's epilogue VALU work: 145.5 k -> 151.8 k queries/s (10 M x 1000).
        // (Raising it already in front of the DMA issue: the same; different priorities for the two blocks of a CU: -1 %.)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            if (t + 3 < KS) {
                QR_LD(a[(t + 3) & 3], t + 3);
                asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            } else if (t + 2 < KS) {
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            } else if (t + 1 < KS) {
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if (t == 0) {
                QR_MFMA0(0, 0) QR_MFMA0(1, 0) QR_MFMA0(2, 0) QR_MFMA0(3, 0)
            } else {
                QR_MFMA(0, t) QR_MFMA(1, t) QR_MFMA(2, t) QR_MFMA(3, t)
            }
        }
#undef QR_LD
#undef QR_MFMA0
#undef QR_MFMA
#if QR_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- epilogue of the group: lane (lq, lg) holds queries qbase + 16 j + lq, rows 4 lg + r
        float4 sv4;
        {
            const unsigned sa_ = sxs_base + (unsigned)((ctile_n & 1) * CZ_T + grp * 16 + 4 * lg) * 4u;
            // (the s_nop: results of the asm MFMAs are read by VALU instructions below -- the accumulators pass THROUGH the
            // statement, or the compiler is free to place their conversions in front of it: seen at K = 256 / 512)
            asm volatile("ds_read_b128 %0, %5\n\ts_nop 15\n\ts_nop 3\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(sv4), "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(sa_) : "memory");
        }
        const float svs[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
        float v[4][4];
        unsigned anym = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j][r] = (float)acc[j][r] * svs[r];
            const float mx = fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3]));
            anym |= __ballot(mx >= thr_q[j]) != 0ull ? 1u << j : 0u;
        }
        if (anym != 0u) {
            const int64_t row0 = cur_tile * CZ_T + grp * 16;
            const bool edge = row0 + 16 > ntotal || mask != nullptr;   // wave uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (((anym >> j) & 1u) == 0u) continue;   // wave uniform
                const unsigned qv = (unsigned)(qbase + 16 * j + lq);
                // one pass per row position r (static register indices: a per-lane pick out of v[j][.] becomes a scratch
                // access, whose pending load would put a vmcnt(0) -- the whole DMA ring -- in front of the loop's LDS reads)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + 4 * lg + r;
                    bool hit = v[j][r] >= thr_q[j];
                    if (edge) hit = hit && row < ntotal && CZ_ALLOWED(mask, row);
                    const unsigned long long b = __ballot(hit);
                    if (b == 0ull) continue;   // wave uniform
                    if (wcount + 64 > CZ_WCAP) {
                        CZ_FLUSH();
                    }
                    if (hit) {
                        const unsigned sl = (unsigned)wcount + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
                        const unsigned ad = wl_base + sl * 4u;
                        asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%4\n\tds_write_b32 %0, %3 offset:%5"
                                     : : "v"(ad), "v"(v[j][r]), "v"((unsigned)row), "v"(qv), "n"(CZ_WCAP * 4), "n"(CZ_WCAP * 8) : "memory");
                    }
                    wcount += __popcll(b);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wcount > 0) CZ_FLUSH();
#undef QR_ISSUE
#undef QR_DMA16
#undef QR_DMA4
}

// Row loads of the 1..4-query sweeps.  Every row is read ONCE per search, so the loads are non-temporal: the 7.7 GB of a
// 10 M-row sweep no longer push what IS reused (the encoder's weights between two queries, the candidate buffers) out of
// L2 / the Infinity Cache, and the sweep itself got faster -- one query, 10 M rows: cascade kernel 1.30 -> 1.24 ms
// (6.2 TB/s), encode + search chained 1.99 -> 1.81 ms (CZ_SWEEP_NT=0 builds for A/B runs).
#ifndef CZ_SWEEP_NT
#define CZ_SWEEP_NT 1
#endif
__device__ __forceinline__ uint4 cz_row_load(const uint4* p) {
#if CZ_SWEEP_NT
    typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
    const nt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
// One step of the bf16 sweep: rows ra_ / rb_ (clamped to the index) against the NQ queries in LDS; lane `sub` == j of
// a row's 16 lanes gets query j's scores (ma: row ra_, mb: row rb_).  xn2 != null: L2 form 2 x.q - ||x||^2.
template <int NQ, int TT>
__device__ __forceinline__ void cz_sweep_pair_bf16(const unsigned short* __restrict__ xh, const float* qs, int64_t ra_, int64_t rb_,
                                                   int dpad, int steps, int sub, const float* __restrict__ xn2, float& ma, float& mb) {
    const uint4* pa = reinterpret_cast<const uint4*>(xh + (size_t)ra_ * dpad) + sub;
    const uint4* pb = reinterpret_cast<const uint4*>(xh + (size_t)rb_ * dpad) + sub;
    float sa[NQ], sb[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) sa[j] = sb[j] = 0.f;
    if constexpr (TT > 0) {
        uint4 va[TT], vb[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            va[t] = cz_row_load(pa + 16 * t);
            vb[t] = cz_row_load(pb + 16 * t);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const unsigned wa[4] = {va[t].x, va[t].y, va[t].z, va[t].w};
            const unsigned wb[4] = {vb[t].x, vb[t].y, vb[t].z, vb[t].w};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float4 q0 = *reinterpret_cast<const float4*>(qs + j * dpad + 128 * t + 8 * sub);
                const float4 q1 = *reinterpret_cast<const float4*>(qs + j * dpad + 128 * t + 8 * sub + 4);
                const float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    sa[j] = fmaf(__uint_as_float(wa[w] << 16), qv[2 * w], sa[j]);
                    sa[j] = fmaf(__uint_as_float(wa[w] & 0xFFFF0000u), qv[2 * w + 1], sa[j]);
                    sb[j] = fmaf(__uint_as_float(wb[w] << 16), qv[2 * w], sb[j]);
                    sb[j] = fmaf(__uint_as_float(wb[w] & 0xFFFF0000u), qv[2 * w + 1], sb[j]);
                }
            }
        }
    } else {
        for (int t = 0; t < steps; ++t) {
            if (128 * t + 8 * sub >= dpad) break;  // dpad is a multiple of 64: whole 16-B chunks
            const uint4 xa = cz_row_load(pa + 16 * t), xb4 = cz_row_load(pb + 16 * t);
            const unsigned wa[4] = {xa.x, xa.y, xa.z, xa.w};
            const unsigned wb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const float* qv = qs + j * dpad + 128 * t + 8 * sub;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    sa[j] = fmaf(__uint_as_float(wa[w] << 16), qv[2 * w], sa[j]);
                    sa[j] = fmaf(__uint_as_float(wa[w] & 0xFFFF0000u), qv[2 * w + 1], sa[j]);
                    sb[j] = fmaf(__uint_as_float(wb[w] << 16), qv[2 * w], sb[j]);
                    sb[j] = fmaf(__uint_as_float(wb[w] & 0xFFFF0000u), qv[2 * w + 1], sb[j]);
                }
            }
        }
    }
    float xa = 0.f, xb2 = 0.f;  // L2: ||row||^2 (score = 2 x.q - ||x||^2)
    if (xn2 != nullptr) {
        xa = xn2[ra_];
        xb2 = xn2[rb_];
    }
    ma = mb = 0.f;  // score of query `sub` for this lane's row
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float ra = row16_allsum(sa[j]), rb = row16_allsum(sb[j]);
        if (sub == j) {
            ma = xn2 != nullptr ? fmaf(2.f, ra, -xa) : ra;
            mb = xn2 != nullptr ? fmaf(2.f, rb, -xb2) : rb;
        }
    }
}

// One step of the int8 sweep (layout and arithmetic: k_sweep_coarse_i8 below): rows ra_ / rb_ against the NQ permuted
// queries in LDS (qlen floats each; my_off = 128 * sum of the query's elements for lane `sub`'s query).
// QREG (one query, compile-time steps): the lane's 4 * TT float4 of the query come in registers (qr) instead of LDS.
template <int NQ, int TT, bool QREG = false>
__device__ __forceinline__ void cz_sweep_pair_i8(const unsigned char* __restrict__ x8, const float* __restrict__ x8s, const float* qs,
                                                 int qlen, float my_off, int64_t ra_, int64_t rb_, int dpad, int chunks, int steps,
                                                 int sub, const float* __restrict__ xn2, float& ma, float& mb,
                                                 const float4* qr = nullptr) {
    const uint4* pa = reinterpret_cast<const uint4*>(x8 + (size_t)ra_ * dpad) + sub;
    const uint4* pb = reinterpret_cast<const uint4*>(x8 + (size_t)rb_ * dpad) + sub;
    const float scA = x8s[ra_], scB = x8s[rb_];
    float sa[NQ], sb[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) sa[j] = sb[j] = 0.f;
#define CZ_I8_STEP(T_, VA_, VB_)                                                                              \
    {                                                                                                  \
        /* signed bytes -> offset binary (byte + 128) for v_cvt_f32_ubyte: one XOR per four elements */  \
        const unsigned wa[4] = {(VA_).x ^ 0x80808080u, (VA_).y ^ 0x80808080u, (VA_).z ^ 0x80808080u, (VA_).w ^ 0x80808080u}; \
        const unsigned wb[4] = {(VB_).x ^ 0x80808080u, (VB_).y ^ 0x80808080u, (VB_).z ^ 0x80808080u, (VB_).w ^ 0x80808080u}; \
        _Pragma("unroll") for (int j = 0; j < NQ; ++j) {                                               \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
                const float4 qv = QREG ? qr[4 * (T_) + r]                                                \
                                       : *reinterpret_cast<const float4*>(qs + j * qlen + 256 * (T_) + 64 * r + 4 * sub); \
                sa[j] = fmaf((float)(wa[r] & 0xFFu), qv.x, sa[j]);                                     \
                sa[j] = fmaf((float)((wa[r] >> 8) & 0xFFu), qv.y, sa[j]);                              \
                sa[j] = fmaf((float)((wa[r] >> 16) & 0xFFu), qv.z, sa[j]);                             \
                sa[j] = fmaf((float)(wa[r] >> 24), qv.w, sa[j]);                                       \
                sb[j] = fmaf((float)(wb[r] & 0xFFu), qv.x, sb[j]);                                     \
                sb[j] = fmaf((float)((wb[r] >> 8) & 0xFFu), qv.y, sb[j]);                              \
                sb[j] = fmaf((float)((wb[r] >> 16) & 0xFFu), qv.z, sb[j]);                             \
                sb[j] = fmaf((float)(wb[r] >> 24), qv.w, sb[j]);                                       \
            }                                                                                          \
        }                                                                                              \
    }
    if constexpr (TT > 0) {
        uint4 va[TT], vb[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            va[t] = cz_row_load(pa + 16 * t);
            vb[t] = cz_row_load(pb + 16 * t);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) CZ_I8_STEP(t, va[t], vb[t])
    } else {
        for (int t = 0; t < steps; ++t) {
            if (16 * t + sub >= chunks) break;
            const uint4 xa = cz_row_load(pa + 16 * t), xb4 = cz_row_load(pb + 16 * t);
            CZ_I8_STEP(t, xa, xb4)
        }
    }
#undef CZ_I8_STEP
    float xa = 0.f, xb2 = 0.f;  // L2: ||row||^2 (score = 2 x.q - ||x||^2)
    if (xn2 != nullptr) {
        xa = xn2[ra_];
        xb2 = xn2[rb_];
    }
    ma = mb = 0.f;  // score of query `sub` for this lane's row
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const float ra = row16_allsum(sa[j]), rb = row16_allsum(sb[j]);
        if (sub == j) {
            const float ca = scA * (ra - my_off), cb = scB * (rb - my_off);
            ma = xn2 != nullptr ? fmaf(2.f, ca, -xa) : ca;
            mb = xn2 != nullptr ? fmaf(2.f, cb, -xb2) : cb;
        }
    }
}

// The same cascade stage for 1..4 queries: an HBM-bound sweep over the bf16 shadow rows (half the bytes of the
// fp32 sweep of k_scan_small).  16 lanes per row read 16 B each (8 bf16 = 256 B contiguous per row and column
// step), the bf16 are widened to fp32 by a shift, multiplied with the fp32 query from LDS and reduced over the
// 16 lanes with 4 DPP adds; lane j of the row group then tests query j's score against its threshold.
// Two groups of 4 rows per iteration keep 2*TT 16-B loads in flight per lane.  TT = dpad/128 (0: run-time).
template <int NQ, int TT, bool MAIN>  // MAIN: the last (stride 1) stage gets its own name in profiles
__global__ __launch_bounds__(256) void k_sweep_coarse(const unsigned short* __restrict__ xh,
                                                      const float* __restrict__ qpad, const float* __restrict__ thr,
                                                      float* __restrict__ cand_s, uint32_t* __restrict__ cand_i,
                                                      int* __restrict__ cand_n, int64_t ntotal, int dpad, int nq,
                                                      int64_t count, int64_t stride, int gm1, int stage0,
                                                      const uint32_t* __restrict__ mask, const float* __restrict__ xn2) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [NQ][dpad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, rg = lane >> 4;
    for (int i = tid; i < NQ * dpad; i += 256) qs[i] = (i / dpad) < nq ? qpad[i] : 0.f;
    float my_thr = INFINITY;  // lane `sub` looks after query `sub`
    if (sub < nq && !stage0) my_thr = thr[sub];
    __syncthreads();
    const int steps = TT > 0 ? TT : (dpad + 127) / 128;
    for (int64_t u = blockIdx.x; u < count; u += gridDim.x) {
        const int64_t tile = (stage0 ? u : u + u / gm1 + 1) * stride;
        const int64_t row_base = tile * CZ_T + wave * 64;
#pragma unroll 1
        for (int it = 0; it < 16; it += 2) {
            // NQ > 1: keep the query reads inside the loop (hoisted, they take 48 NQ registers and the
            // occupancy with them); one query stays in registers
            if constexpr (NQ > 1) asm volatile("" ::: "memory");
            const int64_t rowA = row_base + it * 4 + rg, rowB = rowA + 4;
            float ma, mb;
            cz_sweep_pair_bf16<NQ, TT>(xh, qs, rowA < ntotal ? rowA : ntotal - 1, rowB < ntotal ? rowB : ntotal - 1, dpad, steps, sub,
                                       xn2, ma, mb);
            if (sub < nq) {
                const bool okA = rowA < ntotal && CZ_ALLOWED(mask, rowA), okB = rowB < ntotal && CZ_ALLOWED(mask, rowB);
                if (stage0) {
                    const size_t o = (size_t)sub * CZ_CAP + (size_t)u * CZ_T + wave * 64 + it * 4 + rg;
                    cand_s[o] = okA ? ma : -INFINITY;
                    cand_i[o] = okA ? (uint32_t)rowA : kInvalidRow;
                    cand_s[o + 4] = okB ? mb : -INFINITY;
                    cand_i[o + 4] = okB ? (uint32_t)rowB : kInvalidRow;
                } else {
                    if (ma >= my_thr && okA) {
                        const int slot = atomicAdd(&cand_n[(size_t)(sub) * CZ_NS], 1);
                        if (slot < CZ_CAP) {
                            cand_s[(size_t)sub * CZ_CAP + slot] = ma;
                            cand_i[(size_t)sub * CZ_CAP + slot] = (uint32_t)rowA;
                        }
                    }
                    if (mb >= my_thr && okB) {
                        const int slot = atomicAdd(&cand_n[(size_t)(sub) * CZ_NS], 1);
                        if (slot < CZ_CAP) {
                            cand_s[(size_t)sub * CZ_CAP + slot] = mb;
                            cand_i[(size_t)sub * CZ_CAP + slot] = (uint32_t)rowB;
                        }
                    }
                }
            }
        }
    }
}

// The same sweep over the INT8 shadow rows (css_index: signed byte = round(x / s), s = max|x| / 127 per row): half the
// bytes of the bf16 sweep.  With u = byte + 128 (one XOR per dword): score = s * (sum_i u_i q_i - 128 sum_i q_i); a
// lane converts its 16 bytes of a 16-B chunk with v_cvt_f32_ubyte0..3 and accumulates 16 * steps products in order, the 16 lanes of a row add up in a
// 4-step tree (the fp32 accumulation term of the error bound, cz_eps, counts on that depth).  The query sits in LDS
// permuted -- element 256 t + 16 sub + 4 r + e at 256 t + 64 r + 4 sub + e -- so that the 16 lanes of a row read 256
// contiguous bytes per ds_read_b128.  TT = dpad / 256 when that is exact (768: 3), else 0 = run-time steps.
template <int NQ, int TT, bool MAIN>
__global__ __launch_bounds__(256) void k_sweep_coarse_i8(const unsigned char* __restrict__ x8, const float* __restrict__ x8s,
                                                         const float* __restrict__ qpad, const float* __restrict__ thr,
                                                         float* __restrict__ cand_s, uint32_t* __restrict__ cand_i,
                                                         int* __restrict__ cand_n, int64_t ntotal, int dpad, int nq,
                                                         int64_t count, int64_t stride, int gm1, int stage0,
                                                         const uint32_t* __restrict__ mask, const float* __restrict__ xn2) {
    extern __shared__ __attribute__((aligned(16))) float qs[];  // [NQ][256 * steps] permuted, then [NQ] offsets
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, rg = lane >> 4;
    const int chunks = dpad >> 4;                       // 16-B chunks per row (dpad is a multiple of 64)
    const int steps = TT > 0 ? TT : (chunks + 15) / 16;
    const int qlen = 256 * steps;
    float* qoff = qs + NQ * qlen;
    for (int i = tid; i < NQ * qlen; i += 256) {
        const int j = i / qlen, pos = i - j * qlen;
        const int t = pos >> 8, r = (pos >> 6) & 3, sb = (pos >> 2) & 15, e = pos & 3;
        const int col = 256 * t + 16 * sb + 4 * r + e;
        qs[i] = (j < nq && col < dpad) ? qpad[(size_t)j * dpad + col] : 0.f;
    }
    __syncthreads();
    if (wave == 0) {   // 128 * sum of the query's elements (the offset of the unsigned bytes)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float a = 0.f;
            for (int i = lane; i < qlen; i += 64) a += qs[j * qlen + i];
            a = wave_allsum(a);
            if (lane == 0) qoff[j] = 128.f * a;
        }
    }
    float my_thr = INFINITY;  // lane `sub` looks after query `sub`
    if (sub < nq && !stage0) my_thr = thr[sub];
    __syncthreads();
    float my_off = sub < NQ ? qoff[sub < NQ ? sub : 0] : 0.f;
    for (int64_t u = blockIdx.x; u < count; u += gridDim.x) {
        const int64_t tile = (stage0 ? u : u + u / gm1 + 1) * stride;
        const int64_t row_base = tile * CZ_T + wave * 64;
#pragma unroll 1
        for (int it = 0; it < 16; it += 2) {
            if constexpr (NQ > 1) asm volatile("" ::: "memory");
            const int64_t rowA = row_base + it * 4 + rg, rowB = rowA + 4;
            float ma, mb;
            cz_sweep_pair_i8<NQ, TT>(x8, x8s, qs, qlen, my_off, rowA < ntotal ? rowA : ntotal - 1, rowB < ntotal ? rowB : ntotal - 1,
                                     dpad, chunks, steps, sub, xn2, ma, mb);
            if (sub < nq) {
                const bool okA = rowA < ntotal && CZ_ALLOWED(mask, rowA), okB = rowB < ntotal && CZ_ALLOWED(mask, rowB);
                if (stage0) {
                    const size_t o = (size_t)sub * CZ_CAP + (size_t)u * CZ_T + wave * 64 + it * 4 + rg;
                    cand_s[o] = okA ? ma : -INFINITY;
                    cand_i[o] = okA ? (uint32_t)rowA : kInvalidRow;
                    cand_s[o + 4] = okB ? mb : -INFINITY;
                    cand_i[o + 4] = okB ? (uint32_t)rowB : kInvalidRow;
                } else {
                    if (ma >= my_thr && okA) {
                        const int slot = atomicAdd(&cand_n[(size_t)(sub) * CZ_NS], 1);
                        if (slot < CZ_CAP) {
                            cand_s[(size_t)sub * CZ_CAP + slot] = ma;
                            cand_i[(size_t)sub * CZ_CAP + slot] = (uint32_t)rowA;
                        }
                    }
                    if (mb >= my_thr && okB) {
                        const int slot = atomicAdd(&cand_n[(size_t)(sub) * CZ_NS], 1);
                        if (slot < CZ_CAP) {
                            cand_s[(size_t)sub * CZ_CAP + slot] = mb;
                            cand_i[(size_t)sub * CZ_CAP + slot] = (uint32_t)rowB;
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 3..32 queries: the sweep on the int8 MFMA
// The VALU sweep above costs one fp32 multiply-add per element and QUERY: four queries are VALU-bound (2.65 ms at
// 10 M rows), and up to 32 queries went through the 256-query tiles of the batch scan, whose LDS-DMA ring re-fetches the
// query tile for every row tile (1.95-2.0 ms).  Here the (up to 16) int8 queries sit in registers as the B operand
// of v_mfma_i32_16x16x64_i8 -- lane l: query l & 15, bytes 16 (l >> 4) .. + 15 of every 64-byte K step -- and the int8
// rows stream from HBM straight into the A operand (lane l: row l & 15 of a 16-row group, the same bytes), so the
// sweep is HBM-bound again: a 16 x 16 score tile costs 12 loads of 16 bytes and 12 MFMAs per lane at 768 columns.
// Result lane l: query l & 15, rows 4 (l >> 4) .. + 3 of the group; score = acc * row scale * query scale (exact int32
// accumulation; error band: both operands int8, as in the batch scan).  One launch per cascade stage; thresholds,
// selects, rescoring and fix-up are the candidate path's.
// NG = 1: up to 16 queries; NG = 2: up to 32 (two fragment sets, twice the MFMAs per row group: still far from MFMA-bound).
template <int KS, bool MAIN, int NG = 1>   // KS = dpad / 64 when known at compile time (768: 12), else 0 = run-time steps; MAIN names the stride-1 stage
__global__ __launch_bounds__(256) void k_sweep_mfma_i8(const unsigned char* __restrict__ x8, const float* __restrict__ x8s,
                                                       const signed char* __restrict__ q8, const float* __restrict__ qsc,
                                                       const float* __restrict__ thr, float* __restrict__ cand_s,
                                                       uint32_t* __restrict__ cand_i, int* __restrict__ cand_n, int64_t ntotal,
                                                       int dpad, int nq, int64_t count, int64_t stride, int gm1, int stage0,
                                                       const uint32_t* __restrict__ mask) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane & 15, lg = lane >> 4;
    const int ks = KS > 0 ? KS : dpad >> 6;
    constexpr int KSMAX = KS > 0 ? KS : 16;   // (rows of at most 1024 elements carry int8 copies)
    // this lane's query fragments (query 16 g + lq): rows nq .. 16 NG - 1 of the int8 query block are zeros (k_rows_to_i8)
    v4i_t qf[NG][KSMAX];
    float my_qs[NG], my_thr[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int t = 0; t < KSMAX; ++t)
            qf[g][t] = t < ks ? *reinterpret_cast<const v4i_t*>(q8 + (size_t)(16 * g + lq) * dpad + 64 * t + 16 * lg) : v4i_t{0, 0, 0, 0};
        my_qs[g] = 16 * g + lq < nq ? qsc[16 * g + lq] : 0.f;
        my_thr[g] = (16 * g + lq < nq && !stage0) ? thr[16 * g + lq] : INFINITY;
    }
    for (int64_t u = blockIdx.x; u < count; u += gridDim.x) {
        const int64_t tile = (stage0 ? u : u + u / gm1 + 1) * stride;
        const int64_t row_base = tile * CZ_T + wave * 64;
#pragma unroll 1
        for (int grp = 0; grp < 4; grp += 2) {   // two 16-row groups per pass: 2 * ks loads of 16 bytes in flight per lane
            v4i_t a0[KSMAX], a1[KSMAX];
            const int64_t rA = row_base + 16 * grp + lq, rB = rA + 16;
            const uint4* pa = reinterpret_cast<const uint4*>(x8 + (size_t)(rA < ntotal ? rA : ntotal - 1) * dpad) + lg;
            const uint4* pb = reinterpret_cast<const uint4*>(x8 + (size_t)(rB < ntotal ? rB : ntotal - 1) * dpad) + lg;
#pragma unroll
            for (int t = 0; t < KSMAX; ++t) {
                if (t < ks) {
                    // (default cache policy: a 128-byte line is touched by two consecutive steps, 64 bytes each)
                    const uint4 va = pa[4 * t], vb = pb[4 * t];
                    a0[t] = v4i_t{(int)va.x, (int)va.y, (int)va.z, (int)va.w};
                    a1[t] = v4i_t{(int)vb.x, (int)vb.y, (int)vb.z, (int)vb.w};
                }
            }
            v4i_t c0[NG], c1[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) c0[g] = c1[g] = v4i_t{0, 0, 0, 0};
#pragma unroll
            for (int t = 0; t < KSMAX; ++t) {
                if (t < ks) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        c0[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[t], qf[g][t], c0[g], 0, 0, 0);
                        c1[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[t], qf[g][t], c1[g], 0, 0, 0);
                    }
                }
            }
            // lane: queries 16 g + lq, rows R0 .. R0 + 3 (group A) and + 16 (group B)
            const int64_t R0 = row_base + 16 * grp + 4 * lg;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int64_t Rh = R0 + 16 * half;
                float xs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) xs[r] = x8s[Rh + r < ntotal ? Rh + r : ntotal - 1];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int qi = 16 * g + lq;
                    if (qi >= nq) continue;
                    const v4i_t c = half ? c1[g] : c0[g];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = Rh + r;
                        const float sc = (float)c[r] * xs[r] * my_qs[g];
                        const bool ok = row < ntotal && CZ_ALLOWED(mask, row);
                        if (stage0) {
                            const size_t o = (size_t)qi * CZ_CAP + (size_t)u * CZ_T + (size_t)(row - tile * CZ_T);
                            cand_s[o] = ok ? sc : -INFINITY;
                            cand_i[o] = ok ? (uint32_t)row : kInvalidRow;
                        } else if (sc >= my_thr[g] && ok) {
                            const int slot = atomicAdd(&cand_n[(size_t)qi * CZ_NS], 1);
                            if (slot < CZ_CAP) {
                                cand_s[(size_t)qi * CZ_CAP + slot] = sc;
                                cand_i[(size_t)qi * CZ_CAP + slot] = (uint32_t)row;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Block-wide bitonic sort of P (power of two, <= CZ_CAP) LDS entries, best first: score desc, id asc.
__device__ __forceinline__ void cz_bitonic(float* s, uint32_t* id, int P, int tid) {
    for (int size = 2; size <= P; size <<= 1)
        for (int st = size >> 1; st > 0; st >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += 256) {
                const int lo = ((t & ~(st - 1)) << 1) | (t & (st - 1));
                const int hi = lo | st;
                const bool desc = (lo & size) == 0;
                const float sl = s[lo], sh = s[hi];
                const uint32_t il = id[lo], ih = id[hi];
                const bool hi_better = better<uint32_t>(sh, ih, sl, il);
                if (hi_better == desc) {
                    s[lo] = sh; s[hi] = sl;
                    id[lo] = ih; id[hi] = il;
                }
            }
        }
    __syncthreads();
}

// Exact fp32 scores of candidate rows id[first .. first + R) -> s[first ..]: one wave per row, fixed summation order
// (lane-strided float4 columns, four fmaf chains, one wave reduction), FOUR rows per wave in flight -- one row at a
// time made the rescoring of a dense band a chain of dependent HBM latencies (4096 rows: 1.7 ms per query block).
// Entries whose id is kInvalidRow, or whose current (coarse) score in s[] is below `cmin`, get -inf.
// IP: x.q; L2: -(||x - q||^2), formed directly as the exact kernels do.  (s / id may be global or LDS arrays.)
__device__ __forceinline__ void cz_rescore_rows(float* s, const uint32_t* id, int first, int R, const float4* qv,
                                                const float* __restrict__ xb, int dpad, int l2, int wave, int lane,
                                                float cmin) {
    const int nj = dpad >> 2;
    for (int c0 = first + wave * 4; c0 < first + R; c0 += 16) {
        const float4* xv[4];
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = c0 + r < first + R ? c0 + r : c0;   // (clamped: the duplicate is not stored)
            const uint32_t row = id[ci];
            live[r] = c0 + r < first + R && row != kInvalidRow && s[ci] >= cmin;
            xv[r] = reinterpret_cast<const float4*>(xb + (size_t)(live[r] ? row : 0u) * dpad);
        }
        float a[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[r][e] = 0.f;
#define CZ_RESCORE_STEP(J_)                                                                        \
        {                                                                                              \
            const float4 y = qv[J_];                                                                   \
            float4 x[4];                                                                               \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) x[r] = xv[r][J_];                            \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
                if (l2) {                                                                              \
                    const float dx = x[r].x - y.x, dy = x[r].y - y.y, dz = x[r].z - y.z, dw = x[r].w - y.w; \
                    a[r][0] = fmaf(dx, dx, a[r][0]);                                                   \
                    a[r][1] = fmaf(dy, dy, a[r][1]);                                                   \
                    a[r][2] = fmaf(dz, dz, a[r][2]);                                                   \
                    a[r][3] = fmaf(dw, dw, a[r][3]);                                                   \
                } else {                                                                               \
                    a[r][0] = fmaf(x[r].x, y.x, a[r][0]);                                              \
                    a[r][1] = fmaf(x[r].y, y.y, a[r][1]);                                              \
                    a[r][2] = fmaf(x[r].z, y.z, a[r][2]);                                              \
                    a[r][3] = fmaf(x[r].w, y.w, a[r][3]);                                              \
                }                                                                                      \
            }                                                                                          \
        }
        if (nj == 192) {   // dim 768 (the reference's model): fixed trip count, all 12 row loads of a pass in flight together
#pragma unroll
            for (int it = 0; it < 3; ++it) CZ_RESCORE_STEP(lane + 64 * it)
        } else {
            for (int j = lane; j < nj; j += 64) CZ_RESCORE_STEP(j)
        }
#undef CZ_RESCORE_STEP
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = wave_allsum((a[r][0] + a[r][1]) + (a[r][2] + a[r][3]));
            if (lane == 0 && c0 + r < first + R) s[c0 + r] = live[r] ? (l2 ? -e : e) : -INFINITY;
        }
    }
}

// Error bound of a coarse score c of row x for query q.
//   a priori   : |c - x.q| <= eps_rel ||q|| max||x|| (unit roundoff of the rounded operands, Cauchy-Schwarz).
//   measured   : c - x.q = (x^ - x).q^ + x.(q^ - q) exactly (x^, q^ the bf16 operands), so
//                |c - x.q| <= max||x^ - x|| (||q|| + ||q^ - q||) + max||x|| ||q^ - q||  +  2^-11 ||q|| max||x||
//                (last term: fp32 accumulation, ~10 x its worst case at dim 768).  The error norms are what the
//                conversions actually produced: max over the index rows (k_ingest_rows, second word of maxn2) and
//                per query (qerr2; 0 for the sweep, whose queries stay fp32); `measured` = which copy of the rows
//                the scores came from (maxn2 word 1: bf16 rows, 2: the int8 rows of the sweep).  ex2 < 0: a priori bound only (the
//                split-operand scan, whose operands are not the bf16 roundings).  Random rows: ~0.41 of a unit
//                roundoff each, the band is ~2.3 x narrower than the a-priori one and holds ~3 x fewer rows.
// L2 (score 2 x.q - ||x||^2 vs the directly computed -(||x-q||^2) + ||q||^2): twice that, plus the fp32 cancellation
// of the expanded form.
#ifndef CZ_EPS_TEST_SCALE
#define CZ_EPS_TEST_SCALE 1.f   // (mutation builds: a band drawn too narrow must make tests/test_knn_gpu.py fail)
#endif
__device__ __forceinline__ float cz_eps(float eps_rel, float qn2, float mx2, int l2, float ex2, float qe2) {
    const float qn = sqrtf(qn2), mx = sqrtf(mx2);
    float eps = eps_rel * qn * mx;
    if (ex2 >= 0.f) {
        const float qe = sqrtf(qe2);
        // (1.001: rounding of the fp32 sums behind the error norms, ~dim * 2^-24 relative)
        eps = fminf(eps, CZ_EPS_TEST_SCALE * 1.001f * (sqrtf(ex2) * (qn + qe) + mx * qe) + 0.00048828125f * qn * mx);
    }
    eps += 1e-30f;
    if (l2) eps = 2.f * eps + 9.5367431640625e-07f * (mx2 + qn2);
    return eps;
}
#define CZ_EPS_OF(Q_) cz_eps(eps_rel, qnorm2[Q_], __int_as_float(maxn2_bits[0]), l2,                      \
                             measured ? __int_as_float(maxn2_bits[measured]) : -1.f, (measured && qerr2) ? qerr2[Q_] : 0.f)

// One block per query.  Between the stages of the cascade (FINAL = false): sort the candidate buffer, take the k-th
// best coarse score Tc, publish thr = Tc - 2 eps and keep only the entries >= thr.  After the last stage (FINAL =
// true) the same cut leaves the BAND in the buffer (cand_n = its size), to be rescored exactly by k_rescore_parts
// and turned into the top-k by k_coarse_final: three launches, because the rescoring of a dense band (hundreds to
// thousands of rows of one cluster) inside this one-block-per-query kernel was a serial chain of HBM latencies --
// 1.7 ms per batch on clustered rows in round 2 -- and now spreads over CZ_PARTS blocks per query.
// A query whose band or buffer overflowed is FLAGGED: flags[q] = its slot in flag_list + 1.  Only the 512 best
// buffered candidates of such a query are rescored: all that is wanted from them is a lower bound of the exact k-th
// best score for the second pass, and any subset of rows gives one.
// Rows per part of a band of n rows split over at most CZ_PARTS blocks (a multiple of the 16 rows a block has in flight)
__device__ __forceinline__ int cz_part_rows(int n) { return (((n + CZ_PARTS - 1) / CZ_PARTS) + 15) & ~15; }
constexpr int CZ_FLAGGED_RESCORE = 512;
// k-th largest of s[0 .. n) (n >= k) by a most-significant-digit radix selection over order-preserving 32-bit keys: four
// passes of an 8-bit LDS histogram + one wave's suffix scan over the 256 bins.  The stage selects need the k-th best
// coarse score and the entries above a threshold derived from it, not an ordering: the full bitonic sort they used
// until round 3 (78 barrier-separated rounds at 4096 entries) was 15-35 us of every select and, with one block per
// query, 0.35 ms of the 2.8 ms single-query search (7 selects in a row).  Block of 256 threads; hist: 256 words of LDS.
// (one digit of that selection: hist holds the counts of the 256 digit values among the keys still in play; wave 0 finds the
// bin of the kk-th largest and the number of keys in the bins above it -- sel[0], sel[1]; ends with a block barrier)
__device__ __forceinline__ unsigned long long cz_radix_lane(const unsigned* hist, int kk, int lane, int& bsel, int& above) {
    // lane l owns bins 255 - 4 l .. 252 - 4 l; cum = keys in its bins and all higher ones.  Returns the lanes whose cum
    // reaches kk: the lowest of them holds the kk-th largest's bin (bsel) and the number of keys in the bins above it
    const int b0 = 255 - 4 * lane;
    const unsigned c0 = hist[b0], c1 = hist[b0 - 1], c2 = hist[b0 - 2], c3 = hist[b0 - 3];
    const unsigned tot = (c0 + c1) + (c2 + c3);
    unsigned cum = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)cum, o);
        if (lane >= o) cum += v;
    }
    unsigned a = cum - tot;
    bsel = b0;
    if (a + c0 < (unsigned)kk) {
        a += c0;
        bsel = b0 - 1;
        if (a + c1 < (unsigned)kk) {
            a += c1;
            bsel = b0 - 2;
            if (a + c2 < (unsigned)kk) {
                a += c2;
                bsel = b0 - 3;
            }
        }
    }
    above = (int)a;
    return __ballot(cum >= (unsigned)kk);   // (never empty: the candidates of this pass number >= kk)
}
__device__ __forceinline__ void cz_radix_pick(const unsigned* hist, int kk, int* sel, int tid) {
    if (tid < 64) {
        int bsel, above;
        const unsigned long long hit = cz_radix_lane(hist, kk, tid, bsel, above);
        if (tid == __ffsll((long long)hit) - 1) {
            sel[0] = bsel;
            sel[1] = above;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ float cz_kth_largest(const float* s, int n, int k, unsigned* hist, int* sel, int tid) {
    unsigned prefix = 0, mask = 0;
    int kk = k;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const unsigned u = (unsigned)f2key(s[i]) ^ 0x80000000u;   // unsigned order = float order
            if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
        }
        __syncthreads();
        cz_radix_pick(hist, kk, sel, tid);
        prefix |= (unsigned)sel[0] << shift;
        mask |= 0xFFu << shift;
        kk -= sel[1];
        __syncthreads();
    }
    return key2f((int)(prefix ^ 0x80000000u));
}
template <bool FINAL>
__global__ __launch_bounds__(256) void k_coarse_select(float* __restrict__ cand_s, uint32_t* __restrict__ cand_i,
                                                       int* __restrict__ cand_n, float* __restrict__ thr,
                                                       int* __restrict__ flags, int* __restrict__ nflag,
                                                       int* __restrict__ flag_list, const float* __restrict__ qnorm2,
                                                       const int* __restrict__ maxn2_bits, float eps_rel, int l2, int k,
                                                       int closed_n, int* __restrict__ gthr,
                                                       const float* __restrict__ qerr2, int measured,
                                                       float* __restrict__ fix_s, uint32_t* __restrict__ fix_i,
                                                       int* __restrict__ fix_lock, const float* __restrict__ ex_q,
                                                       const float* __restrict__ ex_x, int ex_dpad) {
    __shared__ float s[CZ_CAP];
    __shared__ uint32_t id[CZ_CAP];
    __shared__ unsigned hist[256];
    __shared__ int sel[2];
    __shared__ int cnt;
    __shared__ uint32_t ex_id[CZ_EXK];
    __shared__ float ex_s[CZ_EXK];
    __shared__ int ex_n;
    __shared__ float ex_thr;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n_raw = cand_n[(size_t)(q) * CZ_NS];
    const bool overflow = n_raw > CZ_CAP;
    const int n = min(n_raw, CZ_CAP);
    for (int i = tid; i < n; i += 256) {
        s[i] = cand_s[(size_t)q * CZ_CAP + i];
        id[i] = cand_i[(size_t)q * CZ_CAP + i];
    }
    if (tid == 0) cnt = 0;
    __syncthreads();
    const float Tc = n >= k ? cz_kth_largest(s, n, k, hist, sel, tid) : -INFINITY;
    const float eps = CZ_EPS_OF(q);
    float thr_new = Tc - 2.f * eps;  // -inf stays -inf
    // Sharper, with ONE eps (ex_x != null: the int8 scan of batches, inner product, k <= CZ_EXK / 2): the rows that hold
    // the k best coarse scores are scored exactly here; s_k = their k-th best exact score is a lower bound of the final
    // k-th best exact score (a k-th best over a subset), so every row of the final top-k has a coarse score >= s_k - eps.
    // Both bounds are valid, the larger one is used; s_k sits within the (small) actual error of Tc, so the threshold
    // rises by about one eps -- on the int8 rows, whose band is 0.87 sigma of the score distribution wide, that is ~4 x
    // fewer appends in the stage that follows and ~4 x fewer band rows at the end.
    if (ex_x != nullptr && n >= k && k * 2 <= CZ_EXK) {
        if (tid == 0) {
            ex_n = 0;
            ex_thr = -INFINITY;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256)
            if (s[i] >= Tc && id[i] != kInvalidRow) {
                const int p_ = atomicAdd(&ex_n, 1);
                if (p_ < CZ_EXK) {
                    ex_id[p_] = id[i];
                    ex_s[p_] = 0.f;   // (cz_rescore_rows looks at the score it replaces)
                }
            }
        __syncthreads();
        const int T = min(ex_n, CZ_EXK);
        // (the rescoring of the final select: four rows in flight per wave, the same summation order)
        cz_rescore_rows(ex_s, ex_id, 0, T, reinterpret_cast<const float4*>(ex_q + (size_t)q * ex_dpad), ex_x, ex_dpad, 0,
                        tid >> 6, tid & 63, -INFINITY);
        __syncthreads();
        if (T >= k && tid < T) {   // rank by counting (T <= CZ_EXK entries)
            const float mine = ex_s[tid];
            int rank = 0;
            for (int j = 0; j < T; ++j) rank += (ex_s[j] > mine || (ex_s[j] == mine && j < tid)) ? 1 : 0;
            if (rank == k - 1) ex_thr = mine - eps;
        }
        __syncthreads();
        thr_new = fmaxf(thr_new, ex_thr);
    }
    bool bad = false;
    if constexpr (FINAL) {
        // closed_n > 0 (split-operand scan, which keeps only its closed_n best scores): a band that reaches the last
        // kept rank may continue beyond it -- decided on the count, below; buffer overflows are known already
        bad = overflow || flags[q] != 0;
    }
    if (!bad) {
        // the band, compacted in place of the buffer (in arrival order of the LDS counter: nothing downstream depends
        // on the order -- the next select takes a k-th largest again, the final sort is by exact score and id)
        for (int i = tid; i < n; i += 256) {
            if (s[i] >= thr_new && id[i] != kInvalidRow) {
                const int pos = atomicAdd(&cnt, 1);
                cand_s[(size_t)q * CZ_CAP + pos] = s[i];
                cand_i[(size_t)q * CZ_CAP + pos] = id[i];
            }
        }
        __syncthreads();
    }
    const int m = cnt;
    if constexpr (!FINAL) {
        if (tid == 0) {
            cand_n[(size_t)(q) * CZ_NS] = m;
            thr[q] = thr_new;
            if (overflow) flags[q] = 1;
        }
    } else {
        if (!bad && closed_n > 0 && m >= closed_n) bad = true;   // (m > CZ_RMAX cannot happen: the band is part of the buffer)
        int R = m;
        if (bad) {
            // the exact fix-up (k_scan_small<FIX>), should the query get that far, starts its global list, threshold
            // and lock from scratch
            for (int i = tid; i < k; i += 256) {
                fix_s[(size_t)q * k + i] = -INFINITY;
                fix_i[(size_t)q * k + i] = kInvalidRow;
            }
            // a flagged query hands its CZ_FLAGGED_RESCORE best buffered candidates to the rescoring (the tighter the
            // bound they give, the fewer rows the second pass collects): the one case that still sorts
            int P = 2;
            while (P < n) P <<= 1;
            for (int i = n + tid; i < P; i += 256) {
                s[i] = -INFINITY;
                id[i] = kInvalidRow;
            }
            cz_bitonic(s, id, P, tid);   // (starts and ends with a block barrier)
            if (tid == 0) cnt = 0;
            __syncthreads();
            int c = 0;
            for (int i = tid; i < min(n, CZ_FLAGGED_RESCORE); i += 256) {
                if (s[i] >= thr_new && id[i] != kInvalidRow) {   // (the band is a prefix of the sorted buffer)
                    cand_s[(size_t)q * CZ_CAP + i] = s[i];
                    cand_i[(size_t)q * CZ_CAP + i] = id[i];
                    ++c;
                }
            }
            if (c) atomicAdd(&cnt, c);
            __syncthreads();
            R = cnt;
        }
        if (tid == 0) {
            cand_n[(size_t)(q) * CZ_NS] = R;
            int fl = 0;
            if (bad) {
                gthr[q] = f2key(-INFINITY);
                fix_lock[q] = 0;
                const int slot = atomicAdd(nflag, 1);
                flag_list[slot] = q;
                fl = slot + 1;
            }
            flags[q] = fl;
        }
    }
}

// ------------------------------------------------------------------ the 1..4-query cascade in ONE launch
// k_sweep_cascade runs every stage of the sweep cascade (the schedule of launch_scan_coarse) and the selects between
// them in one persistent launch: 8 sweeps + 7 one-block selects at 10 M rows cost ~0.25 ms of launch gaps, ramps and
// tails on top of the bytes.  The unit of work is a QUARTER tile (64 rows: what one wave covers of a tile in the stage
// kernels); quarters are TICKETS in stage order (stage 0 first, the stride-1 stage last).
// Tickets are dealt in two levels, because agent-scope atomics on ONE address complete at only ~30 per microsecond on this
// chip (measured: a counter add per tile made the launch atomic-bound, 1.5-3.6 ms): a block takes a CHUNK of
// consecutive tickets from the global counter (4 in the short early stages, up to 64 in the long ones, shrinking
// again over the last stretch), its four waves draw single tickets from the chunk with one 64-bit LDS add.
// Every WAVE is an agent of its own -- no block barrier after the prologue: it scores its quarter into its LDS ring of
// CZ_FS_RING pending quarters and tests a pending quarter against its stage's thresholds once those are published.  A
// wave counts the quarters it has appended per stage and adds them to the stage's counter when it moves on to another
// stage; the wave whose add completes stage s runs select(s) -- the arithmetic of k_coarse_select<false>, by one
// wave over the buffer in L2 -- and publishes the thresholds of stage s + 1 as key words.
// No wait can deadlock, whatever part of the grid is resident: chunks are claimed in ticket order and a block's waves
// draw a chunk in order, so the smallest ticket not yet appended is always either being scored or the OLDEST pending
// quarter of its wave, and the thresholds it waits for depend on smaller tickets only (stage 0 needs none).  Spins are
// bounded all the same: a wave that gives up sets the abort word and flags every query, which sends them to the
// exact fix-up (k_scan_small<FIX>) -- slow, never wrong.
// Visibility across CUs / XCDs (cdna_hip_programming.md Guideline 16): every shared word is an agent-scope atomic;
// candidate entries are write-through (sc1) stores, drained by the storing wave before its counter add; the selecting
// wave takes one agent-scope acquire before it loads them (with sc1 loads).
constexpr int CZ_FS_SPINS = 1 << 18;       // polls (~0.5 us apart) before a wave gives up: ~0.1 s (kernel argument: tests pass 0)
// dynamic LDS of k_sweep_cascade in floats: queries | 4 offsets | 4 (the block's chunk word) | 4 waves x (ring | 256 bins | 2 ring arrays)
__host__ __device__ constexpr int cz_fs_lds_floats(int nq_t, int qlen) {
    return nq_t * qlen + 4 + 4 + 4 * (CZ_FS_RING * nq_t * 64 + 256 + 2 * CZ_FS_RING);
}
struct FsSched {
    int nstage;
    int first[CZ_FS_MAXST + 1];   // first ticket (quarter tile) of stage s; first[nstage] = number of tickets
    int stride[CZ_FS_MAXST];      // tile stride of stage s
    int gm1[CZ_FS_MAXST];         // stage s > 0 reads the tiles of its stride that stage s - 1 did not: u + u / gm1 + 1
};
#define CZ_AT_LD(P_) __hip_atomic_load((P_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CZ_AT_ST(P_, V_) __hip_atomic_store((P_), (V_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// LDS traffic between the lanes of ONE wave: its DS operations execute in order, what has to be stopped is the compiler
#define CZ_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// k-th largest of the n floats at s (global memory, other CUs' write-through stores: sc1 loads) by ONE wave: the
// radix selection of cz_kth_largest with the wave's own 256 bins
__device__ __forceinline__ float cz_kth_largest_wave(const float* s, int n, int k, unsigned* hist, int lane) {
    unsigned prefix = 0, mask = 0;
    int kk = k;
#pragma unroll 1
    for (int shift = 24; shift >= 0; shift -= 8) {
        hist[lane] = hist[lane + 64] = hist[lane + 128] = hist[lane + 192] = 0;
        CZ_WAVE_LDS_SYNC();
        for (int c = 0; c < n; c += 64 * 8) {   // 8 loads in flight per lane (each is an L2 round trip)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = c + 64 * j + lane;
                v[j] = i < n ? CZ_AT_LD(&s[i]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned u = (unsigned)f2key(v[j]) ^ 0x80000000u;   // unsigned order = float order
                if (c + 64 * j + lane < n && (u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
            }
        }
        CZ_WAVE_LDS_SYNC();
        int bsel, above;
        const unsigned long long hit = cz_radix_lane(hist, kk, lane, bsel, above);
        const int src = __ffsll((long long)hit) - 1;
        prefix |= (unsigned)__shfl(bsel, src) << shift;
        mask |= 0xFFu << shift;
        kk -= __shfl(above, src);
        CZ_WAVE_LDS_SYNC();
    }
    return key2f((int)(prefix ^ 0x80000000u));
}

template <int NQ, int TT, bool I8>
__global__ __launch_bounds__(256) void k_sweep_cascade(const void* __restrict__ rows, const float* __restrict__ x8s,
                                                       const float* __restrict__ qpad, float* cand_s, uint32_t* cand_i,
                                                       int* cand_n, float* thr_out, int* flags, int64_t ntotal, int dpad,
                                                       int nq, FsSched sc, int* fs, const uint32_t* __restrict__ mask,
                                                       const float* __restrict__ xn2, const float* __restrict__ qnorm2,
                                                       const int* __restrict__ maxn2_bits, float eps_rel, int l2, int k,
                                                       int measured, int spin_limit) {
    // queries (layout of the stage kernels) | [4] offsets | per wave: ring, bins, ring bookkeeping (all of it in the
    // dynamic region: statics in front of it would shift its 16-byte alignment)
    extern __shared__ __attribute__((aligned(16))) float qs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15, rg = lane >> 4;
    const int chunks = dpad >> 4;
    const int steps = I8 ? (TT > 0 ? TT : (chunks + 15) / 16) : (TT > 0 ? TT : (dpad + 127) / 128);
    const int qlen = I8 ? 256 * steps : dpad;
    float* qoff = qs + NQ * qlen;
    // the block's chunk of tickets: first ticket << 32 | size << 16 | tickets drawn (first = -1: none yet, = total: the end)
    unsigned long long* chunk = reinterpret_cast<unsigned long long*>(qoff + 4);
    constexpr int kWaveWords = CZ_FS_RING * NQ * 64 + 256 + 2 * CZ_FS_RING;
    float* ring = qoff + 8 + wave * kWaveWords;                          // [CZ_FS_RING][NQ][64] scores of the pending quarters
    unsigned* hist = reinterpret_cast<unsigned*>(ring + CZ_FS_RING * NQ * 64);   // [256]
    int* ring_stage = reinterpret_cast<int*>(hist + 256);                // [CZ_FS_RING] stage of a pending quarter ...
    int* ring_rel = ring_stage + CZ_FS_RING;                             // ... and its ticket relative to the stage's first
    if constexpr (I8) {
        for (int i = tid; i < NQ * qlen; i += 256) {
            const int j = i / qlen, pos = i - j * qlen;
            const int t = pos >> 8, r = (pos >> 6) & 3, sb = (pos >> 2) & 15, e = pos & 3;
            const int col = 256 * t + 16 * sb + 4 * r + e;
            qs[i] = (j < nq && col < dpad) ? qpad[(size_t)j * dpad + col] : 0.f;
        }
    } else {
        for (int i = tid; i < NQ * dpad; i += 256) qs[i] = (i / dpad) < nq ? qpad[i] : 0.f;
    }
    if (tid == 0) *chunk = 0xFFFFFFFFull << 32;
    __syncthreads();
    if (I8 && wave == 0) {   // 128 * sum of the query's elements (the offset of the unsigned bytes)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            float a = 0.f;
            for (int i = lane; i < qlen; i += 64) a += qs[j * qlen + i];
            a = wave_allsum(a);
            if (lane == 0) qoff[j] = 128.f * a;
        }
    }
    __syncthreads();   // the last block barrier: from here on every wave runs on its own
    const float my_off = (I8 && sub < NQ) ? qoff[sub < NQ ? sub : 0] : 0.f;
#ifndef CZ_FS_QREG
#define CZ_FS_QREG 1
#endif
    // one int8 query: its 48 floats per lane stay in registers (as the stage kernels' compiler-hoisted copy did)
    constexpr bool kQReg = CZ_FS_QREG && I8 && NQ == 1 && TT > 0;
    float4 qr[kQReg ? 4 * TT : 1];
    if constexpr (kQReg) {
#pragma unroll
        for (int i = 0; i < 4 * TT; ++i) qr[i] = *reinterpret_cast<const float4*>(qs + 256 * (i >> 2) + 64 * (i & 3) + 4 * sub);
    }
    const int total = sc.first[sc.nstage];
    const int long_from = sc.first[sc.nstage > 2 ? sc.nstage - 2 : 0];   // the two long stages start here
    const int* keys = fs + CZ_FS_KEY + ((int)blockIdx.x % CZ_FS_COPIES) * 4 * CZ_FS_MAXST;   // this block's copy of the key words
    int head = 0, npend = 0, known = 0, cur = 0;
    int acc_stage = 0, acc_n = 0;   // quarters of stage acc_stage this wave has appended and not yet added to the stage's counter
    bool acc_stored = false, aborted = false;
    float thr[NQ];   // thresholds of stage `known`
#pragma unroll
    for (int q = 0; q < NQ; ++q) thr[q] = -INFINITY;

    // first row of ticket `rel` of stage s (a quarter of the tile the stage kernels would give their block)
    auto quarter_rows = [&](int s, int rel) -> int64_t {
        const int u = rel >> 2;
        const int64_t tile = (int64_t)(s == 0 ? u : u + u / sc.gm1[s] + 1) * sc.stride[s];
        return tile * CZ_T + (rel & 3) * 64;
    };

    // select(s), by this wave: k-th best coarse score of everything seen, thresholds of stage s + 1, buffer cut to them
    auto select_stage = [&](int s) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int thr_key[4] = {0, 0, 0, 0};
        for (int q = 0; q < nq; ++q) {
            float* cs = cand_s + (size_t)q * CZ_CAP;
            uint32_t* ci = cand_i + (size_t)q * CZ_CAP;
            const int n_raw = CZ_AT_LD(&cand_n[(size_t)q * CZ_NS]);
            const int n = min(n_raw, CZ_CAP);
            const float Tc = n >= k ? cz_kth_largest_wave(cs, n, k, hist, lane) : -INFINITY;
            const float eps = cz_eps(eps_rel, qnorm2[q], __int_as_float(maxn2_bits[0]), l2,
                                     measured ? __int_as_float(maxn2_bits[measured]) : -1.f, 0.f);
            float thr_new = Tc - 2.f * eps;  // -inf stays -inf
            if (!(thr_new == thr_new)) thr_new = -INFINITY;   // NaN scores: keep everything (the query ends up flagged)
            // the band, compacted in place 64 entries at a time (a chunk is in registers before anything is written,
            // and what is written lies in front of the chunks still to be read)
            int kept = 0;
            for (int c = 0; c < n; c += 64 * 4) {   // (4 chunks loaded before the first is written)
                float v[4];
                uint32_t id[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = c + 64 * j + lane;
                    v[j] = -INFINITY;
                    id[j] = kInvalidRow;
                    if (i < n) {
                        v[j] = CZ_AT_LD(&cs[i]);
                        id[j] = CZ_AT_LD(&ci[i]);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool keep = c + 64 * j + lane < n && v[j] >= thr_new && id[j] != kInvalidRow;
                    const unsigned long long m = __ballot(keep);
                    if (keep) {
                        const int pos = kept + __popcll(m & ((1ull << lane) - 1ull));
                        CZ_AT_ST(&cs[pos], v[j]);
                        CZ_AT_ST(&ci[pos], id[j]);
                    }
                    kept += __popcll(m);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                CZ_AT_ST(&cand_n[(size_t)q * CZ_NS], kept);
                thr_out[q] = thr_new;
                if (n_raw > CZ_CAP) CZ_AT_ST(&flags[q], 1);
            }
            thr_key[q] = f2key(thr_new);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the counts are in place before anybody may append
        static_assert(CZ_FS_COPIES == 64, "one copy of the key words per lane");
        for (int q = 0; q < nq; ++q) CZ_AT_ST(&fs[CZ_FS_KEY + lane * 4 * CZ_FS_MAXST + 4 * (s + 1) + q], thr_key[q]);
    };

    // adds the quarters this wave has appended to their stage's counter; the add that completes the stage runs its select
    auto publish = [&]() {
        if (acc_n == 0) return;
        if (acc_stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's entries are out before it counts them
        int last = 0;
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(&fs[CZ_FS_DONE + acc_stage * CZ_FS_LINE], acc_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = old + acc_n == sc.first[acc_stage + 1] - sc.first[acc_stage];
        }
        const int s = acc_stage;
        acc_n = 0;
        acc_stored = false;
        if (__builtin_amdgcn_readfirstlane(last)) select_stage(s);
    };

    // tests the pending quarters, oldest first, against their stage's thresholds; force: wait for them
    auto drain = [&](bool force) {
        int spins = 0;
        while (npend > 0) {
            const int slot = (head + CZ_FS_RING - npend) % CZ_FS_RING;
            const int s = __builtin_amdgcn_readfirstlane(ring_stage[slot]);
            if (s != acc_stage) publish();   // (this wave has moved on: what it appended of the earlier stage counts now)
            if (s > known) {   // (every lane polls the same words -- this block's copy: one request per load)
                int ok = (force && (spins & 63) == 63 && CZ_AT_LD(&fs[CZ_FS_ABORT]) != 0) ? -1 : 1;
                float tq[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    tq[q] = -INFINITY;
                    if (q < nq) {
                        const int key = CZ_AT_LD(&keys[4 * s + q]);
                        if (key == CZ_FS_SENT) ok = ok > 0 ? 0 : ok;
                        tq[q] = key2f(key);
                    }
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                if (ok == 0 && force && ++spins > spin_limit) {   // (never seen outside the tests: see the kernel's header)
                    if (lane == 0) {
                        CZ_AT_ST(&fs[CZ_FS_ABORT], 1);
                        for (int q = 0; q < nq; ++q) {
                            CZ_AT_ST(&flags[q], 1);                        // -> the exact fix-up answers the query
                            CZ_AT_ST(&cand_n[(size_t)q * CZ_NS], 0);       // (stage-0 slots may never be written now)
                        }
                    }
                    ok = -1;
                }
                if (ok < 0) {
                    aborted = true;
                    npend = 0;
                    return;
                }
                if (ok == 0) {
                    if (!force) return;
                    __builtin_amdgcn_s_sleep(16);
                    continue;
                }
                known = s;
#pragma unroll
                for (int q = 0; q < NQ; ++q) thr[q] = tq[q];
            }
            const int rel = __builtin_amdgcn_readfirstlane(ring_rel[slot]);
            const int64_t row = quarter_rows(s, rel) + lane;
            const bool row_ok = row < ntotal && CZ_ALLOWED(mask, row);
            const float* rs = ring + slot * NQ * 64;
            bool stored = false;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (q >= nq) break;
                const float v = rs[q * 64 + lane];
                if (s == 0) {   // stage 0 keeps every score, at fixed slots (cand_n starts at their number)
                    const size_t o = (size_t)q * CZ_CAP + (size_t)rel * 64 + lane;
                    CZ_AT_ST(&cand_s[o], row_ok ? v : -INFINITY);
                    CZ_AT_ST(&cand_i[o], row_ok ? (uint32_t)row : kInvalidRow);
                    stored = true;
                } else if (row_ok && v >= thr[q]) {
                    const int at = atomicAdd(&cand_n[(size_t)q * CZ_NS], 1);
                    if (at < CZ_CAP) {
                        CZ_AT_ST(&cand_s[(size_t)q * CZ_CAP + at], v);
                        CZ_AT_ST(&cand_i[(size_t)q * CZ_CAP + at], (uint32_t)row);
                    }
                    stored = true;
                }
            }
            --npend;
            if (s + 1 >= sc.nstage) continue;   // the last stage: nothing in this launch waits for it
            acc_stage = s;
            ++acc_n;
            acc_stored |= __ballot(stored) != 0ull;
        }
    };

    // one ticket for this wave, or -1 when they are used up
    auto draw = [&]() -> int {
        int spins = 0;
        for (;;) {
            unsigned long long old = 0;
            if (lane == 0) old = atomicAdd(chunk, 1ull);
            const int lo = __builtin_amdgcn_readfirstlane((int)(unsigned)old), first = __builtin_amdgcn_readfirstlane((int)(unsigned)(old >> 32));
            const int drawn = lo & 0xFFFF, size = (lo >> 16) & 0xFFFF;
            if (first == total) return -1;
            if (drawn < size) return first + drawn;
            if (drawn == size) {   // the draw that found the chunk used up fetches the block's next one
                // 4 tickets (one per wave) in the short early stages, whose selects wait for the slowest quarter; up to 64
                // in the long ones, shrinking over the last stretch so that the blocks finish together
                const int pos = first < 0 ? 0 : first + size;
                int want = 4;
                if (pos >= long_from) want = min(64, max(4, ((total - pos) / (2 * (int)gridDim.x)) & ~3));
                int g = 0;
                if (lane == 0) g = __hip_atomic_fetch_add(&fs[0], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g = __builtin_amdgcn_readfirstlane(g);
                const unsigned long long word = g >= total ? (unsigned long long)(unsigned)total << 32
                                                           : ((unsigned long long)(unsigned)g << 32) | ((unsigned long long)min(want, total - g) << 16);
                if (lane == 0) __hip_atomic_store(chunk, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                CZ_WAVE_LDS_SYNC();
                continue;
            }
            // another wave of the block is fetching: wait for the chunk to change
            for (;;) {
                __builtin_amdgcn_s_sleep(4);
                unsigned long long now = 0;
                if (lane == 0) now = __hip_atomic_load(chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane((int)(unsigned)(now >> 32)) != first) break;
                if (++spins > spin_limit) return -1;   // (never seen; the tickets of the missing chunk leave their stage open: the waiters give up as well)
            }
        }
    };

    for (;;) {
        const int t = aborted ? -1 : draw();
        if (t < 0) break;
        while (t >= sc.first[cur + 1]) ++cur;
        if (cur != acc_stage) publish();
        const int rel = t - sc.first[cur];
        const int64_t row_base = quarter_rows(cur, rel);
        float* rs = ring + head * NQ * 64;
#pragma unroll 1
        for (int it = 0; it < 16; it += 2) {
            if constexpr (NQ > 1) asm volatile("" ::: "memory");
            const int64_t rowA = row_base + it * 4 + rg, rowB = rowA + 4;
            const int64_t ra_ = rowA < ntotal ? rowA : ntotal - 1, rb_ = rowB < ntotal ? rowB : ntotal - 1;
            float ma, mb;
            if constexpr (I8)
                cz_sweep_pair_i8<NQ, TT, kQReg>(static_cast<const unsigned char*>(rows), x8s, qs, qlen, my_off, ra_, rb_, dpad,
                                                chunks, steps, sub, xn2, ma, mb, qr);
            else
                cz_sweep_pair_bf16<NQ, TT>(static_cast<const unsigned short*>(rows), qs, ra_, rb_, dpad, steps, sub, xn2, ma, mb);
            if (sub < NQ) {
                rs[sub * 64 + it * 4 + rg] = ma;
                rs[sub * 64 + it * 4 + rg + 4] = mb;
            }
        }
        if (lane == 0) {
            ring_stage[head] = cur;
            ring_rel[head] = rel;
        }
        CZ_WAVE_LDS_SYNC();
        head = (head + 1) % CZ_FS_RING;
        ++npend;
        drain(npend == CZ_FS_RING);
    }
    drain(true);
    publish();
}

// Work list of the band rescoring: the (query, part) items that exist, in query order -- one block, an exclusive
// scan over the per-query part counts.  (Built inside k_coarse_select<true> with returning atomics it cost that
// kernel 11-25 us; launching all CZ_PARTS parts of every query cost the rescoring 65 us of block dispatch.)
__global__ __launch_bounds__(1024) void k_rescore_plan(const int* __restrict__ cand_n, int nq, int* __restrict__ work_n,
                                                       int* __restrict__ work) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int q = q0 + tid;
        int np = 0;
        if (q < nq) {
            const int n = min(cand_n[(size_t)q * CZ_NS], CZ_CAP);
            const int per = cz_part_rows(n);
            np = (n + per - 1) / per;
        }
        int incl = np;   // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int pre = base;
        for (int w = 0; w < wave; ++w) pre += wsum[w];
        const int pos = pre + incl - np;
        for (int p = 0; p < np; ++p) work[pos + p] = q * CZ_PARTS + p;
        __syncthreads();
        if (tid == 1023) base = pre + incl;
        __syncthreads();
    }
    if (tid == 0) *work_n = base;
}

// Exact scores for candidate buffers, CZ_PARTS work items per buffer, grid-stride over the items.
//   PASS2 = false: buffer of query q = slot (cap CZ_CAP, cand_n[q] band rows left by k_coarse_select<true>);
//   PASS2 = true:  buffer of slot b < min(*nflag, f2max) belongs to query flag_list[b] (cap entries per slot);
//                  a slot that overflowed is left to k_coarse_select2.
template <bool PASS2>
__global__ __launch_bounds__(256) void k_rescore_parts(float* __restrict__ cand_s, const uint32_t* __restrict__ cand_i,
                                                       const int* __restrict__ cand_n, int cap, int nslots_arg,
                                                       const int* __restrict__ nflag, const int* __restrict__ flag_list,
                                                       const float* __restrict__ thr2, int l2,
                                                       const float* __restrict__ qpad, const float* __restrict__ xb, int dpad,
                                                       const int* __restrict__ work_n, const int* __restrict__ work) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nslots = PASS2 ? min(*nflag, nslots_arg) : nslots_arg;
    // work != null: the list of (slot, part) items that exist, left by k_rescore_plan (a batch of 1000 queries
    // has ~2 live parts per query: launching all 16 was 16000 blocks and 88 us, most of it block dispatch)
    const int nitems = work ? *work_n : nslots * CZ_PARTS;
    for (int w = blockIdx.x; w < nitems; w += gridDim.x) {
        const int item = work ? work[w] : w;
        const int slot = item / CZ_PARTS, part = item % CZ_PARTS;
        const int n = cand_n[(size_t)(slot) * CZ_NS];
        if (n > cap) continue;
        const int per = cz_part_rows(n);
        const int lo = part * per, hi = min(n, lo + per);
        if (lo >= hi) continue;
        const int q = PASS2 ? flag_list[slot] : slot;
        const float4* qv = reinterpret_cast<const float4*>(qpad + (size_t)q * dpad);
        cz_rescore_rows(cand_s + (size_t)slot * cap, cand_i + (size_t)slot * cap, lo, hi - lo, qv, xb, dpad, l2, wave, lane,
                        PASS2 ? thr2[slot] : -INFINITY);
    }
}

// One block per query: the band, now with exact scores, is sorted by (score desc, id asc) and its k best rows are
// the answer.  thr2 / qh2 / f2max (cascade only, else null / 0): a flagged query that holds one of the first f2max
// slots of flag_list also leaves what the SECOND coarse pass needs (launch_scan_coarse): its bf16 row and the
// threshold s_k - eps, s_k = exact k-th best score among the rows rescored -- every row of the exact top-k has an
// exact score >= s_k, hence a coarse score >= s_k - eps (one eps, from exact scores; the cascade's own thresholds
// are k-th best COARSE scores minus two eps).
__global__ __launch_bounds__(256) void k_coarse_final(const float* __restrict__ cand_s, const uint32_t* __restrict__ cand_i,
                                                      const int* __restrict__ cand_n, const int* __restrict__ flags,
                                                      const float* __restrict__ qnorm2, const int* __restrict__ maxn2_bits,
                                                      float eps_rel, int l2, int k, const float* __restrict__ qpad, int dpad,
                                                      int64_t id_base, float* __restrict__ D, int64_t* __restrict__ I,
                                                      float* __restrict__ thr2, unsigned short* __restrict__ qh2, int f2max,
                                                      const float* __restrict__ qerr2, int measured) {
    __shared__ float s[CZ_CAP];
    __shared__ uint32_t id[CZ_CAP];
    __shared__ unsigned hist[256];
    __shared__ int sel[2], cnt;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int R = min(cand_n[(size_t)(q) * CZ_NS], CZ_CAP);
    int P = 2;
    while (P < R) P <<= 1;
    for (int i = tid; i < P; i += 256) {
        s[i] = i < R ? cand_s[(size_t)q * CZ_CAP + i] : -INFINITY;
        id[i] = i < R ? cand_i[(size_t)q * CZ_CAP + i] : kInvalidRow;
        if (!(s[i] > -INFINITY)) {   // dropped by the rescoring (or a NaN score: the ranks below need a total order)
            s[i] = -INFINITY;
            id[i] = kInvalidRow;
        }
    }
    // Only the k best matter.  The usual band (k + a few dozen to several hundred rows): the k-th best exact score by
    // radix selection, the entries that reach it (k, more on ties) move to the upper half of the arrays, each of them
    // counts the ones in front of it and lands at its rank -- ~20 barriers instead of the 36-78 rounds of a bitonic sort
    // of the whole band.
    bool sorted = false;
    if (R <= CZ_CAP / 2) {
        float* s2 = s + CZ_CAP / 2;
        uint32_t* id2 = id + CZ_CAP / 2;
        __syncthreads();
        const int kk = min(k, R);
        const float Tk = kk > 0 ? cz_kth_largest(s, R, kk, hist, sel, tid) : -INFINITY;
        if (tid == 0) cnt = 0;
        __syncthreads();
        for (int i = tid; i < R; i += 256)
            if (s[i] >= Tk) {
                const int pos = atomicAdd(&cnt, 1);
                s2[pos] = s[i];
                id2[pos] = id[i];
            }
        __syncthreads();
        const int m = cnt;   // >= kk
        if (m <= 512) {
            for (int i = tid; i < m; i += 256) {
                const float si = s2[i];
                const uint32_t ii = id2[i];
                int rank = 0;
                for (int j = 0; j < m; ++j) {
                    const float sj = s2[j];
                    const uint32_t ij = id2[j];
                    rank += (sj > si || (sj == si && (ij < ii || (ij == ii && j < i)))) ? 1 : 0;
                }
                s[rank] = si;     // (ranks < m <= R; the band's own entries there are no longer needed:
                id[rank] = ii;    // every thread has passed the barrier behind the selection's last read of them)
            }
            __syncthreads();
            sorted = true;
        }
    }
    if (!sorted) {
        for (int i = tid; i < P; i += 256) {   // (the selection leaves s / id as they were unless it sorted)
            s[i] = i < R ? cand_s[(size_t)q * CZ_CAP + i] : -INFINITY;
            id[i] = i < R ? cand_i[(size_t)q * CZ_CAP + i] : kInvalidRow;
            if (!(s[i] > -INFINITY)) {
                s[i] = -INFINITY;
                id[i] = kInvalidRow;
            }
        }
        cz_bitonic(s, id, P, tid);
    }
    for (int i = tid; i < k; i += 256) {
        const bool ok = i < R && id[i] != kInvalidRow;
        D[(size_t)q * k + i] = l2 ? (ok ? -s[i] : FLT_MAX) : (ok ? s[i] : -FLT_MAX);
        I[(size_t)q * k + i] = ok ? id_base + (int64_t)id[i] : (int64_t)-1;
    }
    const int fslot = flags[q] - 1;
    if (fslot >= 0 && thr2 != nullptr && fslot < f2max) {
        // second coarse pass: coarse scores are x.q (IP) or 2 x.q - ||x||^2 = -(||x - q||^2) + ||q||^2 (L2)
        if (tid == 0) {
            // (eps_rel / qerr2 / measured here describe the SECOND pass's operands -- the bf16 rows and queries --
            // whatever the first pass read)
            const float eps = CZ_EPS_OF(q);
            thr2[fslot] = (R >= k && id[k - 1] != kInvalidRow) ? s[k - 1] + (l2 ? qnorm2[q] : 0.f) - eps : -INFINITY;
        }
        for (int i = tid; i < dpad; i += 256)
            qh2[(size_t)fslot * dpad + i] = __builtin_bit_cast(unsigned short, (__bf16)qpad[(size_t)q * dpad + i]);
    }
}

// Second pass over flagged queries: slot b (< min(*nflag, f2max)) holds query flag_list[b]; its candidates -- every
// row whose coarse score reached thr2[b] -- carry exact scores (k_rescore_parts<true>) and are folded into a running
// top-k, CZ_CAP - 128 at a time; from the second chunk on only entries that reach the current k-th best are sorted.
// A slot whose buffer overflowed again, and every flagged query beyond the f2max slots, goes on to list B, which the
// exact fp32 sweep (launch_fixup) works off.
template <int CAP2>
__global__ __launch_bounds__(256) void k_coarse_select2(const float* __restrict__ cand_s, const uint32_t* __restrict__ cand_i,
                                                        const int* __restrict__ cand_n, const int* __restrict__ nflag,
                                                        const int* __restrict__ flag_list, int f2max,
                                                        int* __restrict__ nflagB, int* __restrict__ flag_listB, int l2, int k,
                                                        int64_t id_base, float* __restrict__ D, int64_t* __restrict__ I) {
    __shared__ float s[CZ_CAP];
    __shared__ uint32_t id[CZ_CAP];
    __shared__ int cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nf = *nflag;
    if (b == 0)   // flagged queries without a slot
        for (int i = f2max + tid; i < nf; i += 256) flag_listB[atomicAdd(nflagB, 1)] = flag_list[i];
    if (b >= min(nf, f2max)) return;
    const int q = flag_list[b];
    const int n = cand_n[(size_t)(b) * CZ_NS];
    if (n > CAP2) {
        if (tid == 0) flag_listB[atomicAdd(nflagB, 1)] = q;
        return;
    }
    constexpr int CH = CZ_CAP - 128;   // k <= 128 entries carried in front
    for (int i = tid; i < k; i += 256) {
        s[i] = -INFINITY;
        id[i] = kInvalidRow;
    }
    float tk = -INFINITY;   // k-th best exact score so far
    for (int base = 0; base < n; base += CH) {
        const int cn = min(CH, n - base);
        if (tid == 0) cnt = k;
        __syncthreads();
        for (int i = tid; i < cn; i += 256) {
            const float e = cand_s[(size_t)b * CAP2 + base + i];
            if (e != -INFINITY && e >= tk) {
                const int pos = atomicAdd(&cnt, 1);
                s[pos] = e;
                id[pos] = cand_i[(size_t)b * CAP2 + base + i];
            }
        }
        __syncthreads();
        const int tot = cnt;
        int P = 2;
        while (P < tot) P <<= 1;
        for (int i = tot + tid; i < P; i += 256) {
            s[i] = -INFINITY;
            id[i] = kInvalidRow;
        }
        cz_bitonic(s, id, P, tid);   // (starts and ends with a block barrier)
        tk = s[k - 1];
        __syncthreads();
    }
    for (int i = tid; i < k; i += 256) {
        const bool ok = id[i] != kInvalidRow;
        D[(size_t)q * k + i] = l2 ? (ok ? -s[i] : FLT_MAX) : (ok ? s[i] : -FLT_MAX);
        I[(size_t)q * k + i] = ok ? id_base + (int64_t)id[i] : (int64_t)-1;
    }
}

