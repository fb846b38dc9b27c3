// gemm_lab.hip -- development bench for the persistent bf16 GEMM main loop shared by the encoder GEMMs
// (k_gemm16) and the kNN coarse scan (k_scan_coarse).  Standalone: hipcc only, no library.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/gemm_lab.hip -o tools/gemm_lab
//   tools/gemm_lab [M N K [variant-mask [reps]]]
//
// C[M,N] (bf16) = A[M,K] . W[N,K]^T + bias, both operands K-contiguous bf16, fp32 accumulate.
// Variants run interleaved in one process on the same random data (cdna_hip_programming.md rule 24):
//   v0  the round-1 loop: 2 x 64 KiB LDS-DMA stages, vmcnt(0) + one barrier per 64-deep K step
//   v1  8-phase ping-pong: four 16-MFMA phases per K step, half-tile DMA slots three ahead with a counted
//       vmcnt, two barriers per phase, the two wave rows staggered by one barrier so that one wave of every
//       SIMD runs MFMAs while its partner reads fragments and issues DMA
//   v3  (mask 8; v4 = mask 16 with sched_group_barrier hints) round 3 prototype: FOUR waves of 128 x 128 (one per SIMD,
//       accumulators meant for AGPRs) on v0's loop -- a third less LDS read traffic per flop.  As written hipcc keeps
//       part of the accumulators in VGPRs (48-66 spills) and does not interleave the fragment reads with the MFMAs:
//       500 TFLOP/s against v1's 1095 on the FFN1 shape.  A contender only with inline-asm AGPR MFMAs and a
//       hand-placed read schedule, as v1 needed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

typedef unsigned short bf16_t;
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
// -DLAB_F16: the same loops on fp16 operands (v_mfma_f32_16x16x32_f16): clock / power comparison with bf16
#if defined(LAB_I8)
// -DLAB_I8 (throughput only, results are not checked): the same loops, the same bytes, on v_mfma_i32_16x16x64_i8 --
// every 16-B fragment is 16 int8 instead of 8 bf16, so a launch does TWICE the printed flops (a K of 768 bf16
// columns is 1536 int8 columns).  What an int8 coarse scan would gain from the MFMA rate alone.
typedef int v8in __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f lab_mfma_i8(v8in a, v8in b, v4f c) {
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(v4f, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, __builtin_bit_cast(v4i_t, c), 0, 0, 0));
}
#define LAB_MFMA(A_, B_, C_, X_, Y_, Z_) lab_mfma_i8(A_, B_, C_)
#elif defined(LAB_F16)
typedef _Float16 v8in __attribute__((ext_vector_type(8)));
#define LAB_MFMA __builtin_amdgcn_mfma_f32_16x16x32_f16
#else
typedef __bf16 v8in __attribute__((ext_vector_type(8)));
#define LAB_MFMA __builtin_amdgcn_mfma_f32_16x16x32_bf16
#endif

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ int swz_byte(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ------------------------------------------------------------------------------------------------ v0
template <int DUMMY>
__global__ __launch_bounds__(512) void k_gemm_v0(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ Cout, int M, int N,
                                                 int K, int dbg = 0) {
    constexpr int NW = 8, WN = 4, TM = 8, TN = 4, BM = 256, BN = 256, RB = 128;
    constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB, PPW = 8;
    constexpr int E = 2 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float sbias[BN];
    constexpr int EPI_ROW = 144;
    __shared__ __attribute__((aligned(16))) char sepi[NW][16 * EPI_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lq = lane & 15, lg = lane >> 4;
    float4 bias_regs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int q = nwg / 8, r = nwg % 8;
    const int xfirst = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcount = q + (xcd < r ? 1 : 0);
    const int my_ntiles = jx < xcount ? (xcount - jx + per_x - 1) / per_x : 0;
    const int KT = K / 64;
    const int total = my_ntiles * KT;
    if (total == 0) return;
    const int prow = lane >> 3, pchunk = lane & 7;
    const char* src[PPW];
    int dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        dst[i] = (piece < 32 ? 0 : A_BYTES) + (piece & 31) * 1024;
    }
    auto set_src = [&](int tile_idx) {
        const int tile = xfirst + jx + tile_idx * per_x;
        const int r0 = (tile / ntn) * BM, c0 = (tile % ntn) * BN;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            const bool isA = piece < 32;
            const int trow = (piece & 31) * 8 + prow;
            int grow = (isA ? r0 : c0) + trow;
            const int lim = isA ? M : N;
            grow = grow < lim ? grow : lim - 1;
            src[i] = reinterpret_cast<const char*>((isA ? A : W) + (size_t)grow * K) + ((pchunk ^ ((trow >> 1) & 7)) << 4);
        }
    };
#define V0_ISSUE(KT_, SLOT_)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                        \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)(KT_) * RB), \
                                         (__attribute__((address_space(3))) void*)(smem + (SLOT_) * STAGE + dst[i]), 16, 0, 0); \
    }
    v4f acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
    int it_tile = 0, it_kt = 0, gi = 0;
    set_src(0);
    V0_ISSUE(0, 0)
    gi = 1;
    if (++it_kt == KT) {
        it_kt = 0;
        if (++it_tile < my_ntiles) set_src(it_tile);
    }
    int ct_tile = 0, kt = 0;
    for (int g = 0; g < total; ++g) {
        if (ct_tile > 0 && kt == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 0 && kt == 2) {
            *reinterpret_cast<float4*>(&sbias[4 * lane]) = bias_regs;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && kt == 1) {
            const int tile_b = xfirst + jx + ct_tile * per_x;
            bias_regs = *reinterpret_cast<const float4*>(bias + (tile_b % ntn) * BN + 4 * lane);
        }
        if (gi < total) {
            V0_ISSUE(it_kt, gi & 1)
            ++gi;
            if (++it_kt == KT) {
                it_kt = 0;
                if (++it_tile < my_ntiles) set_src(it_tile);
            }
        }
        const char* Ab = smem + (g & 1) * STAGE;
        const char* Bb = Ab + A_BYTES;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            v4f a[TM], b[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) a[m] = *reinterpret_cast<const v4f*>(Ab + swz_byte(wr * 128 + 16 * m + lq, 4 * c + lg));
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const v4f*>(Bb + swz_byte(wc * 64 + 16 * n + lq, 4 * c + lg));
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = LAB_MFMA(__builtin_bit_cast(v8in, b[n]),
                                                                        __builtin_bit_cast(v8in, a[m]), acc[m][n], 0, 0, 0);
        }
        if (++kt == KT) {
            const int tile = xfirst + jx + ct_tile * per_x;
            const int row0 = (tile / ntn) * BM + wr * 128 + lq, col0 = (tile % ntn) * BN + wc * 64;
            float4 bv[TN];
#pragma unroll
            for (int n = 0; n < TN; ++n) bv[n] = *reinterpret_cast<const float4*>(&sbias[wc * 64 + 16 * n + 4 * lg]);
            if (!(dbg & 1))
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                char* mine = sepi[wave];
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[m][n][0] + bv[n].x) | ((unsigned)f2bf(acc[m][n][1] + bv[n].y) << 16);
                    pk.y = (unsigned)f2bf(acc[m][n][2] + bv[n].z) | ((unsigned)f2bf(acc[m][n][3] + bv[n].w) << 16);
                    *reinterpret_cast<uint2*>(mine + lq * EPI_ROW + (16 * n + 4 * lg) * 2) = pk;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rr = 8 * t + (lane >> 3);
                    const uint4 o = *reinterpret_cast<const uint4*>(mine + rr * EPI_ROW + (lane & 7) * 16);
                    const size_t gidx = (size_t)((tile / ntn) * BM + wr * 128 + 16 * m + rr) * N + col0 + (lane & 7) * 8;
                    *reinterpret_cast<uint4*>(Cout + gidx) = o;
                }
            }
            (void)row0;
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            ++ct_tile;
        }
    }
#undef V0_ISSUE
}

// ------------------------------------------------------------------------------------------------ v1
// LDS: two K-step buffers x four 16-KiB half-tile slots, in issue order A0, B0, B1, A1:
//   A_s  rows {wr*128 + s*64 + [0,64)}, slot row = wr*64 + r     (every wave row reads only its own 64 rows)
//   B_s  rows {wc*64 + s*32 + [0,32)},  slot row = wc*32 + r
// Phases of K step u (buffer D = u & 1); "issue" = 2 LDS-DMA instructions per wave = one half tile per block:
//   P1  read A_0 (8) + B_0 (4)   issue A_1(u+1) -> D^1    MFMA (A0,B0)
//   P2  read B_1 (4)             issue A_0(u+2) -> D      MFMA (A0,B1)
//   P3  read A_1 (8)             issue B_0(u+2) -> D      MFMA (A1,B1)
//   P4  (B_0 stays in registers) issue B_1(u+2) -> D, vmcnt(6)   MFMA (A1,B0)
// Every phase is  [reads, issue] s_barrier [16 MFMAs] s_barrier ; wave row 1 runs one barrier behind wave row 0.
// A slot is rewritten at the earliest one phase (own rows of A) or two phases (B) after its last read, which covers
// the stagger.
constexpr int HT = 16384;
constexpr int SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;

template <int EPI, bool DBG, int SCHED = 0>
__global__ __launch_bounds__(512) void k_gemm_v1(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ Cout, int M, int N,
                                                 int K, int dbg_arg) {
    const int dbg = DBG ? dbg_arg : 32;   // product form: no timing switches, nontemporal stores
    constexpr int NW = 8, BM = 256, BN = 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][4][HT]
    __shared__ __attribute__((aligned(16))) float sbias[2][BN];
    constexpr int EPI_ROW = 144;
    __shared__ __attribute__((aligned(16))) char sepi[NW][16 * EPI_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lq = lane & 15, lg = lane >> 4;

    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int q = nwg / 8, r = nwg % 8;
    const int xfirst = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcount = q + (xcd < r ? 1 : 0);
    const int my_ntiles = jx < xcount ? (xcount - jx + per_x - 1) / per_x : 0;
    const int KT = K / 64;                 // even (host check)
    const int total = my_ntiles * KT;      // K steps of this block
    if (total == 0) return;
    if (dbg >> 8) {   // experiment: desynchronise the CUs' tile boundaries by a start delay of (jx & 3) quarter tiles
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        const long long wait = (long long)(jx & 3) * (dbg >> 8) * 25;   // memrealtime ticks at 100 MHz: 25 ticks = 0.25 us
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(4);
    }

    // ---- DMA bookkeeping: one 32-bit source offset per half-tile kind (its first piece; the second piece is 8
    // rows further: + 8 K elements rows, and its swizzled chunk differs by XOR 4), advanced independently.
    // Rows beyond M read the slack rows of the activation buffer (their outputs land in slack rows too).
    const int prow = lane >> 3, pchunk = lane & 7;
    // slot rows this wave fills: A kinds: wr*64 + wc*16 + 8*i + prow ; B kinds: 16*wave + 8*i + prow
    unsigned srco[4];
    int it_tile[4], it_kt[4];
    int dsto[4];  // byte offset of this wave's first piece inside a slot
    dsto[SLOT_A0] = dsto[SLOT_A1] = (wr * 64 + wc * 16) * 128;
    dsto[SLOT_B0] = dsto[SLOT_B1] = (16 * wave) * 128;
    const unsigned row8 = 8u * (unsigned)K * 2u;  // bytes between the two pieces' source rows
#define V1_SET_SRC(KIND_)                                                                                     \
    {                                                                                                         \
        const int tile_ = xfirst + jx + it_tile[KIND_] * per_x;                                               \
        const int r0_ = (tile_ / ntn) * BM, c0_ = (tile_ % ntn) * BN;                                         \
        int srow_, grow_;                                                                                     \
        if ((KIND_) == SLOT_A0 || (KIND_) == SLOT_A1) {                                                       \
            const int s_ = (KIND_) == SLOT_A1 ? 1 : 0;                                                        \
            const int rr_ = wc * 16 + prow;                /* row inside the wave row's 64-row sub half */     \
            srow_ = wr * 64 + rr_;                                                                            \
            grow_ = r0_ + wr * 128 + s_ * 64 + rr_;                                                           \
        } else {                                                                                              \
            const int s_ = (KIND_) == SLOT_B1 ? 1 : 0;                                                        \
            srow_ = 16 * wave + prow;                      /* = wc' * 32 + r, r < 16 + 8 */                   \
            grow_ = c0_ + (srow_ >> 5) * 64 + s_ * 32 + (srow_ & 31);                                         \
        }                                                                                                     \
        srco[KIND_] = (unsigned)grow_ * (unsigned)K * 2u + ((pchunk ^ ((srow_ >> 1) & 7)) << 4);              \
    }
// issue kind KIND_ of its next K step into buffer DB_, then advance that kind's cursor
#define V1_ISSUE(KIND_, DB_)                                                                                  \
    {                                                                                                         \
        if (!(dbg & 4)) {                                                                                     \
            const char* base_ = reinterpret_cast<const char*>(((KIND_) == SLOT_A0 || (KIND_) == SLOT_A1) ? A : W); \
            const unsigned o0_ = srco[KIND_] + (unsigned)it_kt[KIND_] * 128u;                                 \
            const unsigned o1_ = (o0_ + row8) ^ 64u;                                                          \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_ + o0_),    \
                (__attribute__((address_space(3))) void*)(smem + ((DB_) * 4 + (KIND_)) * HT + dsto[KIND_]), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_ + o1_),    \
                (__attribute__((address_space(3))) void*)(smem + ((DB_) * 4 + (KIND_)) * HT + dsto[KIND_] + 1024), 16, 0, 0); \
        }                                                                                                     \
        /* past the block's last K step the cursor keeps re-reading the last tile: harmless (the slot it lands in  \
           is never read again) and it keeps the issue unconditional */                                       \
        if (++it_kt[KIND_] == KT) {                                                                           \
            it_kt[KIND_] = 0;                                                                                 \
            if (it_tile[KIND_] + 1 < my_ntiles) {                                                             \
                ++it_tile[KIND_];                                                                             \
                V1_SET_SRC(KIND_)                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    }
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) {
        it_tile[kd] = 0;
        it_kt[kd] = 0;
    }
    V1_SET_SRC(SLOT_A0)
    V1_SET_SRC(SLOT_B0)
    V1_SET_SRC(SLOT_B1)
    V1_SET_SRC(SLOT_A1)

    v4f acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};

    // bias row of the block's first tile -> sbias[0] (1 KiB = one DMA instruction of wave 0)
    // (later tiles: issued by wave 0 in P2 of the tile's first K step into sbias[tile parity])
#define V1_BIAS(TI_)                                                                                          \
    if (wave == 0) {                                                                                          \
        const int tile_b = xfirst + jx + (TI_) * per_x;                                                       \
        __builtin_amdgcn_global_load_lds(                                                                     \
            (const __attribute__((address_space(1))) void*)(bias + (tile_b % ntn) * BN + 4 * lane),           \
            (__attribute__((address_space(3))) void*)(&sbias[(TI_) & 1][0]), 16, 0, 0);                        \
    }

    // ---- prologue: K step 0 entirely, K step 1 without its A_1
    V1_BIAS(0)
    V1_ISSUE(SLOT_A0, 0)
    V1_ISSUE(SLOT_B0, 0)
    V1_ISSUE(SLOT_B1, 0)
    V1_ISSUE(SLOT_A1, 0)
    if (total > 1) {
        V1_ISSUE(SLOT_A0, 1)
        V1_ISSUE(SLOT_B0, 1)
        V1_ISSUE(SLOT_B1, 1)
        if constexpr (SCHED == 1) {
            V1_ISSUE(SLOT_A1, 1)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    {
        v4f bv[4];
        const unsigned sboff = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&sbias[0][wc * 64 + 4 * lg];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                     "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]) : "v"(sboff) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = bv[n];
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0

    // fragment read offsets (bytes inside a slot): row r + 16 j keeps (r >> 1) & 7, so the rows of the j-th
    // 16-row tile are 2048 j bytes further; the second 32-wide k step is chunk ^ 4 = byte offset ^ 64
    const int a_o0 = swz_byte(wr * 64 + lq, lg), a_o1 = a_o0 ^ 64;
    const int b_o0 = swz_byte(wc * 32 + lq, lg), b_o1 = b_o0 ^ 64;
    v4f a[4][2], b0[2][2], b1[2][2];
#define V1_READ_A(S_, D_)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
        a[j][0] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? SLOT_A1 : SLOT_A0)) * HT + j * 2048 + a_o0); \
        a[j][1] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? SLOT_A1 : SLOT_A0)) * HT + j * 2048 + a_o1); \
    }
#define V1_READ_B(S_, D_, B_)                                                                                 \
    _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                           \
        B_[n][0] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? SLOT_B1 : SLOT_B0)) * HT + n * 2048 + b_o0); \
        B_[n][1] = *reinterpret_cast<const v4f*>(smem + ((D_) * 4 + ((S_) ? SLOT_B1 : SLOT_B0)) * HT + n * 2048 + b_o1); \
    }
#define V1_MFMA(MH_, NH_, B_)                                                                                 \
    if (!(dbg & 2)) {                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
                _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                 \
                    acc[4 * (MH_) + j][2 * (NH_) + n] = LAB_MFMA(              \
                        __builtin_bit_cast(v8in, B_[n][c]), __builtin_bit_cast(v8in, a[j][c]),                \
                        acc[4 * (MH_) + j][2 * (NH_) + n], 0, 0, 0);                                          \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    }
#define V1_SYNC_A()                                   \
    __builtin_amdgcn_sched_barrier(0);                \
    __builtin_amdgcn_s_barrier();                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);
#define V1_SYNC_B()                    \
    __builtin_amdgcn_sched_barrier(0); \
    __builtin_amdgcn_s_barrier();      \
    __builtin_amdgcn_sched_barrier(0);

    int ct_tile = 0, kt = 0;
// one K step on buffer D_ (compile-time 0 / 1)
// MFMA cluster with the phase's DMA issue inside it (SCHED 3): 8 MFMAs, the two DMA instructions, 8 MFMAs
#define V1_MFMA_I(MH_, NH_, B_, KIND_, DB_)                                                                   \
    {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                     \
                acc[4 * (MH_) + j][2 * (NH_) + n] = LAB_MFMA(                  \
                    __builtin_bit_cast(v8in, B_[n][0]), __builtin_bit_cast(v8in, a[j][0]),                    \
                    acc[4 * (MH_) + j][2 * (NH_) + n], 0, 0, 0);                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        V1_ISSUE(KIND_, DB_)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                     \
                acc[4 * (MH_) + j][2 * (NH_) + n] = LAB_MFMA(                  \
                    __builtin_bit_cast(v8in, B_[n][1]), __builtin_bit_cast(v8in, a[j][1]),                    \
                    acc[4 * (MH_) + j][2 * (NH_) + n], 0, 0, 0);                                              \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    }
#define V1_KSTEP(D_)                                                                                          \
    if constexpr (SCHED == 3) {                                                                               \
        /* P1 */                                                                                              \
        V1_READ_B(0, D_, b0)                                                                                  \
        V1_READ_A(0, D_)                                                                                      \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* B_1 of this K step has landed (read in P2) */    \
        V1_SYNC_A()                                                                                           \
        V1_MFMA_I(0, 0, b0, SLOT_A1, (D_) ^ 1)                                                                \
        V1_SYNC_B()                                                                                           \
        /* P2 */                                                                                              \
        V1_READ_B(1, D_, b1)                                                                                  \
        if (kt == 0 && ct_tile + 1 < my_ntiles) V1_BIAS(ct_tile + 1)                                          \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* A_1 of this K step has landed (read in P3) */    \
        V1_SYNC_A()                                                                                           \
        V1_MFMA_I(0, 1, b1, SLOT_A0, D_)                                                                      \
        V1_SYNC_B()                                                                                           \
        /* P3 */                                                                                              \
        V1_READ_A(1, D_)                                                                                      \
        V1_SYNC_A()                                                                                           \
        V1_MFMA_I(1, 1, b1, SLOT_B0, D_)                                                                      \
        V1_SYNC_B()                                                                                           \
        /* P4 */                                                                                              \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   /* A_0, B_0 of the next K step have landed */       \
        V1_SYNC_A()                                                                                           \
        V1_MFMA_I(1, 0, b0, SLOT_B1, D_)                                                                      \
        V1_SYNC_B()                                                                                           \
        ++g;                                                                                                  \
        ++kt;                                                                                                 \
    } else {                                                                                                  \
        /* P1 */                                                                                              \
        V1_READ_B(0, D_, b0)                                                                                  \
        V1_READ_A(0, D_)                                                                                      \
        if constexpr (SCHED != 1) V1_ISSUE(SLOT_A1, (D_) ^ 1)                                                 \
        if constexpr (SCHED == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                           \
        V1_SYNC_A()                                                                                           \
        V1_MFMA(0, 0, b0)                                                                                     \
        V1_SYNC_B()                                                                                           \
        /* P2 */                                                                                              \
        V1_READ_B(1, D_, b1)                                                                                  \
        V1_ISSUE(SLOT_A0, D_)                                                                                 \
        if (kt == 0 && ct_tile + 1 < my_ntiles) V1_BIAS(ct_tile + 1)                                          \
        if constexpr (SCHED == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                           \
        V1_SYNC_A()                                                                                           \
        V1_MFMA(0, 1, b1)                                                                                     \
        V1_SYNC_B()                                                                                           \
        /* P3 */                                                                                              \
        V1_READ_A(1, D_)                                                                                      \
        V1_ISSUE(SLOT_B0, D_)                                                                                 \
        V1_SYNC_A()                                                                                           \
        V1_MFMA(1, 1, b1)                                                                                     \
        V1_SYNC_B()                                                                                           \
        /* P4 */                                                                                              \
        V1_ISSUE(SLOT_B1, D_)                                                                                 \
        if constexpr (SCHED == 1) {                                                                           \
            V1_ISSUE(SLOT_A1, D_)                                                                             \
            if (g + 2 < total) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        } else if constexpr (SCHED == 2) {                                                                    \
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                                 \
        } else {                                                                                              \
            if (g + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                               \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             \
        }                                                                                                     \
        V1_SYNC_A()                                                                                           \
        V1_MFMA(1, 0, b0)                                                                                     \
        V1_SYNC_B()                                                                                           \
        ++g;                                                                                                  \
        ++kt;                                                                                                 \
    }

    for (int g = 0; g < total;) {
        V1_KSTEP(0)
        V1_KSTEP(1)
        if (kt == KT) {
            // ---- epilogue of output tile ct_tile (no block barrier inside: the stagger carries over)
            const int tile = xfirst + jx + ct_tile * per_x;
            const int col0 = (tile % ntn) * BN + wc * 64;
            if constexpr (EPI == 0) {
            if (!(dbg & 1)) {
                // 16 x 64 bf16 per-wave transpose through LDS with inline-asm ds ops: a compiler-visible LDS access here
                // makes hipcc wait vmcnt(0) (pending LDS-DMA might alias), i.e. for the previous stores, 8 times per tile
                const unsigned wbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(sepi[wave] + lq * EPI_ROW + 8 * lg);
                const unsigned rbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(sepi[wave] + (lane >> 3) * EPI_ROW + (lane & 7) * 16);
                // dbg bit 4 (16): every tile of this block stores to the block's first tile (stays in L2: is the epilogue bound
                // by the CU's store path or by HBM write bandwidth?)
                const int tile_st = (dbg & 16) ? xfirst + jx : tile;
                bf16_t* cbase = Cout + (size_t)((tile_st / ntn) * BM + wr * 128 + (lane >> 3)) * N + (tile_st % ntn) * BN + wc * 64 + (lane & 7) * 8;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    uint2 pk[4];
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        pk[n].x = (unsigned)f2bf(acc[m][n][0]) | ((unsigned)f2bf(acc[m][n][1]) << 16);
                        pk[n].y = (unsigned)f2bf(acc[m][n][2]) | ((unsigned)f2bf(acc[m][n][3]) << 16);
                    }
                    uint4 o0, o1;
                    asm volatile("ds_write_b64 %6, %2\n\tds_write_b64 %6, %3 offset:32\n\tds_write_b64 %6, %4 offset:64\n\t"
                                 "ds_write_b64 %6, %5 offset:96\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 "ds_read_b128 %0, %7\n\tds_read_b128 %1, %7 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(o0), "=&v"(o1)
                                 : "v"(pk[0]), "v"(pk[1]), "v"(pk[2]), "v"(pk[3]), "v"(wbase), "v"(rbase), "n"(8 * EPI_ROW)
                                 : "memory");
                    if (dbg & 32) {
                        typedef unsigned v4u __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(v4u{o0.x, o0.y, o0.z, o0.w}, reinterpret_cast<v4u*>(cbase + (size_t)(16 * m) * N));
                        __builtin_nontemporal_store(v4u{o1.x, o1.y, o1.z, o1.w}, reinterpret_cast<v4u*>(cbase + (size_t)(16 * m + 8) * N));
                    } else if (!(dbg & 8)) {
                        *reinterpret_cast<uint4*>(cbase + (size_t)(16 * m) * N) = o0;
                        *reinterpret_cast<uint4*>(cbase + (size_t)(16 * m + 8) * N) = o1;
                    } else {
                        asm volatile("" ::"v"(o0.x), "v"(o1.x));
                    }
                }
            }
            } else {
                // accumulators started from the bias row (below): pack, regroup 8 consecutive columns per lane with
                // v_permlane16_swap (lane groups 16 apart trade their halves of two neighbouring 16-column tiles),
                // store 16 B per lane.  No LDS access the compiler can see: no vmcnt(0) of its own in here.
                if (!(dbg & 1)) {
                    bf16_t* crow = Cout + (size_t)((tile / ntn) * BM + wr * 128 + lq) * N + col0 + 16 * (lg & 1) + 8 * (lg >> 1);
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
#pragma unroll
                        for (int np = 0; np < 2; ++np) {
                            const v4f va = acc[m][2 * np], vb = acc[m][2 * np + 1];
                            const unsigned ax = (unsigned)f2bf(va[0]) | ((unsigned)f2bf(va[1]) << 16);
                            const unsigned ay = (unsigned)f2bf(va[2]) | ((unsigned)f2bf(va[3]) << 16);
                            const unsigned bx = (unsigned)f2bf(vb[0]) | ((unsigned)f2bf(vb[1]) << 16);
                            const unsigned by = (unsigned)f2bf(vb[2]) | ((unsigned)f2bf(vb[3]) << 16);
                            const auto rx = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
                            const auto ry = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
                            *reinterpret_cast<uint4*>(crow + (size_t)(16 * m) * N + 32 * np) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                        }
                    }
                }
            }
            {
                // bias row of the NEXT tile (in LDS since P2 of this tile's first K step) -> accumulator start values;
                // read with inline asm: a compiler-visible LDS read here would make hipcc drain the DMA queue (vmcnt(0))
                {
                    v4f bv[4];
                    const unsigned sboff = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&sbias[(ct_tile + 1) & 1][wc * 64 + 4 * lg];
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\t"
                                 "ds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]) : "v"(sboff) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < 8; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[m][n] = bv[n];
                }
            }
            kt = 0;
            ++ct_tile;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may still be landing in LDS when the block ends
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger barrier of wave row 1
}

// ------------------------------------------------------------------------------------------------ v3
// Four waves (one per SIMD), each a 128 x 128 output tile: 256 accumulator registers per lane (the unified 512-register
// file of a single-wave SIMD), 16 fragment reads per 64 MFMAs instead of 12 per 32 -- 64 KB of LDS reads per 32-deep K
// step and CU instead of 96 KB.  Loop of v0 (two 64-KiB stages, one barrier per K step); the reads of the next half
// step are left to the compiler to place among the MFMAs of the current one.
template <int SCHED>
__global__ __launch_bounds__(256) void k_gemm_v3(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ Cout, int M, int N,
                                                 int K, int dbg = 0) {
    constexpr int NW = 4, WN = 2, TM = 8, TN = 8, BM = 256, BN = 256, RB = 128;
    constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB, PPW = 16;
    constexpr int E = 4 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float sbias[BN];
    constexpr int EPI_ROW = 272;
    __shared__ __attribute__((aligned(16))) char sepi[NW][16 * EPI_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lq = lane & 15, lg = lane >> 4;
    float4 bias_regs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int q = nwg / 8, r = nwg % 8;
    const int xfirst = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcount = q + (xcd < r ? 1 : 0);
    const int my_ntiles = jx < xcount ? (xcount - jx + per_x - 1) / per_x : 0;
    const int KT = K / 64;
    const int total = my_ntiles * KT;
    if (total == 0) return;
    const int prow = lane >> 3, pchunk = lane & 7;
    // pieces wave + 4 i: i < 8 rows of A, i >= 8 rows of W, 32 rows apart; the swizzle term is the same for every i
    const char* srcA;
    const char* srcB;
    const size_t step32 = (size_t)32 * K * 2;
    const int dst0 = wave * 1024;
    auto set_src = [&](int tile_idx) {
        const int tile = xfirst + jx + tile_idx * per_x;
        const int r0 = (tile / ntn) * BM, c0 = (tile % ntn) * BN;
        const int trow = wave * 8 + prow;
        const int sw = (pchunk ^ ((trow >> 1) & 7)) << 4;
        srcA = reinterpret_cast<const char*>(A + (size_t)(r0 + trow) * K) + sw;   // (rows beyond M: the buffer's slack rows)
        srcB = reinterpret_cast<const char*>(W + (size_t)(c0 + trow) * K) + sw;
    };
#define V3_ISSUE(KT_, SLOT_)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA + i * step32 + (size_t)(KT_) * RB), \
                                         (__attribute__((address_space(3))) void*)(smem + (SLOT_) * STAGE + dst0 + i * 4096), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB + i * step32 + (size_t)(KT_) * RB), \
                                         (__attribute__((address_space(3))) void*)(smem + (SLOT_) * STAGE + A_BYTES + dst0 + i * 4096), 16, 0, 0); \
    }
    // fragment (16 m + lq, k chunk 4 c + lg): 16 rows further = +2048 B, the other half step = XOR 64 (rows 16 apart share
    // the swizzle term)
    const int aoff = swz_byte(wr * 128 + lq, lg), boff = swz_byte(wc * 128 + lq, lg);
    v4f acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
    int it_tile = 0, it_kt = 0, gi = 0;
    set_src(0);
    V3_ISSUE(0, 0)
    gi = 1;
    if (++it_kt == KT) {
        it_kt = 0;
        if (++it_tile < my_ntiles) set_src(it_tile);
    }
    int ct_tile = 0, kt = 0;
    for (int g = 0; g < total; ++g) {
        if (ct_tile > 0 && kt == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(E) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 0 && kt == 2) {
            *reinterpret_cast<float4*>(&sbias[4 * lane]) = bias_regs;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wave == 0 && kt == 1) {
            const int tile_b = xfirst + jx + ct_tile * per_x;
            bias_regs = *reinterpret_cast<const float4*>(bias + (tile_b % ntn) * BN + 4 * lane);
        }
        if (gi < total) {
            V3_ISSUE(it_kt, gi & 1)
            ++gi;
            if (++it_kt == KT) {
                it_kt = 0;
                if (++it_tile < my_ntiles) set_src(it_tile);
            }
        }
        const char* Ab = smem + (g & 1) * STAGE;
        const char* Bb = Ab + A_BYTES;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            v4f a[TM], b[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) a[m] = *reinterpret_cast<const v4f*>(Ab + (aoff ^ (c * 64)) + m * 2048);
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const v4f*>(Bb + (boff ^ (c * 64)) + n * 2048);
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = LAB_MFMA(__builtin_bit_cast(v8in, b[n]), __builtin_bit_cast(v8in, a[m]), acc[m][n], 0, 0, 0);
            if (SCHED == 1) {   // 16 reads spread over the 64 MFMAs of the half step before
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
            }
        }
        if (++kt == KT) {
            const int tile = xfirst + jx + ct_tile * per_x;
            const int col0 = (tile % ntn) * BN + wc * 128;
            float4 bv[TN];
#pragma unroll
            for (int n = 0; n < TN; ++n) bv[n] = *reinterpret_cast<const float4*>(&sbias[wc * 128 + 16 * n + 4 * lg]);
            if (!(dbg & 1))
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                char* mine = sepi[wave];
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[m][n][0] + bv[n].x) | ((unsigned)f2bf(acc[m][n][1] + bv[n].y) << 16);
                    pk.y = (unsigned)f2bf(acc[m][n][2] + bv[n].z) | ((unsigned)f2bf(acc[m][n][3] + bv[n].w) << 16);
                    *reinterpret_cast<uint2*>(mine + lq * EPI_ROW + (16 * n + 4 * lg) * 2) = pk;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = 4 * t + (lane >> 4);
                    const uint4 o = *reinterpret_cast<const uint4*>(mine + rr * EPI_ROW + (lane & 15) * 16);
                    const size_t gidx = (size_t)((tile / ntn) * BM + wr * 128 + 16 * m + rr) * N + col0 + (lane & 15) * 8;
                    *reinterpret_cast<uint4*>(Cout + gidx) = o;
                }
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
            kt = 0;
            ++ct_tile;
        }
    }
#undef V3_ISSUE
}

// ------------------------------------------------------------------------------------------------ v5
// Round 4: v3's geometry (FOUR waves of 128 x 128, one per SIMD) with what v3 lacked --
//  * accumulators pinned to AGPRs: every MFMA is inline asm with a "+a" accumulator operand (256 AGPRs), operands in VGPRs;
//  * a hand-placed stream: the 64 MFMAs of a half step (c = 0 / 1) carry the 16 fragment reads of the NEXT half step (into
//    the other fragment set) one per 4 MFMAs, and the c = 1 block also the 16 LDS-DMA pieces of the stage after next;
//  * ONE barrier per K step, in its middle: behind it stage g + 1 has landed for every wave (its reads start in block
//    (g, 1)) and every wave has finished reading stage g - 1's slot (the DMA of stage g + 2 into it starts in block (g, 1)).
// Every asm statement clobbers "memory", so asm and DMA builtins keep their written order; the compiler only allocates.
#define V5_MFMA(ACC_, BF_, AF_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC_) : "v"(BF_), "v"(AF_) : "memory")
#define V5_READ(DST_, BASE_, OFF_) asm volatile("ds_read_b128 %0, %1 offset:" #OFF_ : "=v"(DST_) : "v"(BASE_) : "memory")
template <int DUMMY>
__global__ __launch_bounds__(256) void k_gemm_v5(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ Cout, int M, int N,
                                                 int K, int dbg = 0) {
    constexpr int NW = 4, WN = 2, TM = 8, TN = 8, BM = 256, BN = 256, RB = 128;
    constexpr int A_BYTES = BM * RB, STAGE = (BM + BN) * RB;
    constexpr int E = 4 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ __attribute__((aligned(16))) float sbias[BN];
    constexpr int EPI_ROW = 272;
    __shared__ __attribute__((aligned(16))) char sepi[NW][16 * EPI_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int lq = lane & 15, lg = lane >> 4;
    float4 bias_regs = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ntn = N / BN, ntm = (M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int q = nwg / 8, r = nwg % 8;
    const int xfirst = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int xcount = q + (xcd < r ? 1 : 0);
    const int my_ntiles = jx < xcount ? (xcount - jx + per_x - 1) / per_x : 0;
    const int KT = K / 64;
    const int total = my_ntiles * KT;
    if (total == 0) return;
    const int prow = lane >> 3, pchunk = lane & 7;
    // LDS-DMA pieces as inline asm (saddr form: 64-bit scalar base = operand + K-step, 32-bit per-lane offset), so that they
    // stay where the stream puts them: piece I (0..7: rows of A, 8..15: rows of W) of stage gi goes to slot gi & 1.
    // voff[I]: byte offset of this lane's 16 bytes of piece I inside the operand at K step 0 (all < 2^32), per tile.
    unsigned voffA = 0, voffB = 0;
    const size_t step32 = (size_t)32 * K * 2;
    const unsigned lds_dst0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;
    auto set_src = [&](int tile_idx) {
        const int tile = xfirst + jx + tile_idx * per_x;
        const int r0 = (tile / ntn) * BM, c0 = (tile % ntn) * BN;
        const int trow = wave * 8 + prow;
        const unsigned sw = (unsigned)((pchunk ^ ((trow >> 1) & 7)) << 4);
        voffA = (unsigned)(r0 + trow) * (unsigned)K * 2u + sw;   // (rows beyond M: the buffer's slack rows)
        voffB = (unsigned)(c0 + trow) * (unsigned)K * 2u + sw;
    };
    int it_tile = 0, it_kt = 0, gi = 0;   // DMA side: next stage to issue
    const char* kA = reinterpret_cast<const char*>(A);   // operand + K step of stage gi (uniform)
    const char* kB = reinterpret_cast<const char*>(W);
    unsigned lds_dst = lds_dst0;                          // slot of stage gi (uniform)
// (s_mov, not s_add: an s_add inside the asm would clobber SCC between an s_add_u32 / s_addc_u32 pair of the compiler's)
#define V5_DMA(VOFF_, SBASE_, IMM_)                                                                          \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF_), "s"(SBASE_), "s"(lds_dst + (IMM_)) : "memory")
#define V5_PIECE(I_)                                                                                         \
    switch (I_) {                                                                                            \
        case 0: V5_DMA(voffA, kA, 0); break;                                                                 \
        case 1: V5_DMA(voffA, kA + step32, 4096); break;                                                     \
        case 2: V5_DMA(voffA, kA + 2 * step32, 8192); break;                                                 \
        case 3: V5_DMA(voffA, kA + 3 * step32, 12288); break;                                                \
        case 4: V5_DMA(voffA, kA + 4 * step32, 16384); break;                                                \
        case 5: V5_DMA(voffA, kA + 5 * step32, 20480); break;                                                \
        case 6: V5_DMA(voffA, kA + 6 * step32, 24576); break;                                                \
        case 7: V5_DMA(voffA, kA + 7 * step32, 28672); break;                                                \
        case 8: V5_DMA(voffB, kB, 32768); break;                                                             \
        case 9: V5_DMA(voffB, kB + step32, 36864); break;                                                    \
        case 10: V5_DMA(voffB, kB + 2 * step32, 40960); break;                                               \
        case 11: V5_DMA(voffB, kB + 3 * step32, 45056); break;                                               \
        case 12: V5_DMA(voffB, kB + 4 * step32, 49152); break;                                               \
        case 13: V5_DMA(voffB, kB + 5 * step32, 53248); break;                                               \
        case 14: V5_DMA(voffB, kB + 6 * step32, 57344); break;                                               \
        default: V5_DMA(voffB, kB + 7 * step32, 61440); break;                                               \
    }
    /* (stages beyond the last one repeat a K step of the last tile into a slot nobody reads any more: no branches in the stream) */
#define V5_STAGE_DONE()                                                                                      \
    {                                                                                                        \
        ++gi;                                                                                                \
        lds_dst = lds_dst0 + (gi & 1) * STAGE;                                                               \
        if (++it_kt == KT) {                                                                                 \
            it_kt = 0;                                                                                       \
            if (++it_tile < my_ntiles) set_src(it_tile);                                                     \
        }                                                                                                    \
        kA = reinterpret_cast<const char*>(A) + (size_t)it_kt * RB;                                          \
        kB = reinterpret_cast<const char*>(W) + (size_t)it_kt * RB;                                          \
    }
#define V5_ISSUE_ALL()                                                                                       \
    V5_PIECE(0) V5_PIECE(1) V5_PIECE(2) V5_PIECE(3) V5_PIECE(4) V5_PIECE(5) V5_PIECE(6) V5_PIECE(7)         \
    V5_PIECE(8) V5_PIECE(9) V5_PIECE(10) V5_PIECE(11) V5_PIECE(12) V5_PIECE(13) V5_PIECE(14) V5_PIECE(15)   \
    V5_STAGE_DONE()
    // fragment bases (LDS byte addresses): [slot][c] for A and B; fragment m / n at + m * 2048 (B: + A_BYTES)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int aoff = swz_byte(wr * 128 + lq, lg), boff = swz_byte(wc * 128 + lq, lg);
    // [0]: the slot of the stage being multiplied, [1]: the other slot; swapped after every K step
    unsigned ab[2][2], bb[2][2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            ab[sl][c] = lds0 + sl * STAGE + (aoff ^ (c * 64));
            bb[sl][c] = lds0 + sl * STAGE + A_BYTES + (boff ^ (c * 64));
        }
    v4f acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
    // fragments: two sets of A (set c feeds half step c), ONE set of B: B fragment n is dead behind the 8 MFMAs that use
    // it and is reloaded for the next half step right there (256 VGPRs hold no second B set next to the DMA addressing)
    v4f fa[2][TM], fb[TN];
    // side action J_ of a block that multiplies half C_ (next half: slot SL_, half 1 - C_): J_ = 2 n, 2 n + 1 follow the
    // MFMAs of B fragment n.  0..7: next A fragments, two per group; 8..14: next B fragments 0..6 (their groups are done);
    // 15: nothing -- B fragment 7 follows the block (V5_FRAG_LAST) and is needed last in the next one
#define V5_FRAG(SL_, C_, J_)                                                                                 \
    if ((J_) < 8) { V5_READ_SW(fa[1 - (C_)][(J_) & 7], ab[SL_][1 - (C_)], (J_) & 7) }                         \
    else if ((J_) < 15) { V5_READ_SW(fb[(J_) - 8], bb[SL_][1 - (C_)], (J_) - 8) }
#define V5_FRAG_LAST(SL_, C_) V5_READ_SW(fb[7], bb[SL_][1 - (C_)], 7)
#define V5_READ_SW(DST_, BASE_, IDX_)                                                                        \
    switch (IDX_) {                                                                                          \
        case 0: V5_READ(DST_, BASE_, 0); break;                                                              \
        case 1: V5_READ(DST_, BASE_, 2048); break;                                                           \
        case 2: V5_READ(DST_, BASE_, 4096); break;                                                           \
        case 3: V5_READ(DST_, BASE_, 6144); break;                                                           \
        case 4: V5_READ(DST_, BASE_, 8192); break;                                                           \
        case 5: V5_READ(DST_, BASE_, 10240); break;                                                          \
        case 6: V5_READ(DST_, BASE_, 12288); break;                                                          \
        default: V5_READ(DST_, BASE_, 14336); break;                                                         \
    }
    // 4 MFMAs of half step C_ (B fragment J_ / 2 against A fragments 4 (J_ % 2) .. + 3), then the J_-th side action
#define V5_QUAD(C_, J_)                                                                                      \
    V5_MFMA(acc[4 * ((J_) % 2) + 0][(J_) / 2], fb[(J_) / 2], fa[C_][4 * ((J_) % 2) + 0]);                    \
    V5_MFMA(acc[4 * ((J_) % 2) + 1][(J_) / 2], fb[(J_) / 2], fa[C_][4 * ((J_) % 2) + 1]);                    \
    V5_MFMA(acc[4 * ((J_) % 2) + 2][(J_) / 2], fb[(J_) / 2], fa[C_][4 * ((J_) % 2) + 2]);                    \
    V5_MFMA(acc[4 * ((J_) % 2) + 3][(J_) / 2], fb[(J_) / 2], fa[C_][4 * ((J_) % 2) + 3]);

    // prologue: stages 0 and 1 on their way, stage 0 landed for everybody, its c = 0 fragments read
    set_src(0);
    V5_ISSUE_ALL()
    V5_ISSUE_ALL()
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // half 0 of stage 0: A set 0 and the B fragments
        V5_READ_SW(fa[0][j], ab[0][0], j)
        V5_READ_SW(fb[j], bb[0][0], j)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // tiles outside, K steps inside: the accumulators are loop-carried through the inner loop only and zeroed in straight-line
    // code between two tiles (one flat loop with the epilogue under a condition made the allocator spill accumulator tiles)
    for (int ct_tile = 0; ct_tile < my_ntiles; ++ct_tile) {
#pragma unroll 1
        for (int kt = 0; kt < KT; ++kt) {
            // ---- block (g, 0): MFMAs on A set 0, the c = 1 fragments of this stage behind them
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                // (B fragment 7 was read behind the previous block, 14 reads ago, and is multiplied from here on)
                if (j == 14) asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
                V5_QUAD(0, j)
                V5_FRAG(0, 0, j)
            }
            V5_FRAG_LAST(0, 0)
            // ---- middle of the step: stage g + 1 has landed (this wave's pieces; everybody's behind the barrier)
            if (wave == 0 && kt == 2) *reinterpret_cast<float4*>(&sbias[4 * lane]) = bias_regs;
            if (ct_tile > 0 && kt == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(E) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wave == 0 && kt == 1) {
                const int tile_b = xfirst + jx + ct_tile * per_x;
                bias_regs = *reinterpret_cast<const float4*>(bias + (tile_b % ntn) * BN + 4 * lane);
            }
            // ---- block (g, 1): MFMAs on A set 1, the c = 0 fragments of stage g + 1 behind them, the DMA of stage g + 2
            // (behind the last stage the reads fetch fragments nobody multiplies)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                V5_QUAD(1, j)
                V5_FRAG(1, 1, j)
                V5_PIECE(j)
            }
            V5_FRAG_LAST(1, 1)
            V5_STAGE_DONE()
            asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");   // everything but B fragment 7 (used last) is there
#pragma unroll
            for (int c = 0; c < 2; ++c) {   // the other slot becomes the current one
                const unsigned ta = ab[0][c], tb = bb[0][c];
                ab[0][c] = ab[1][c];
                ab[1][c] = ta;
                bb[0][c] = bb[1][c];
                bb[1][c] = tb;
            }
        }
        {
            const int tile = xfirst + jx + ct_tile * per_x;
            const int col0 = (tile % ntn) * BN + wc * 128;
            float4 bv[TN];
#pragma unroll
            for (int n = 0; n < TN; ++n) bv[n] = *reinterpret_cast<const float4*>(&sbias[wc * 128 + 16 * n + 4 * lg]);
            if (!(dbg & 1))
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                char* mine = sepi[wave];
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    uint2 pk;
                    pk.x = (unsigned)f2bf(acc[m][n][0] + bv[n].x) | ((unsigned)f2bf(acc[m][n][1] + bv[n].y) << 16);
                    pk.y = (unsigned)f2bf(acc[m][n][2] + bv[n].z) | ((unsigned)f2bf(acc[m][n][3] + bv[n].w) << 16);
                    *reinterpret_cast<uint2*>(mine + lq * EPI_ROW + (16 * n + 4 * lg) * 2) = pk;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = 4 * t + (lane >> 4);
                    const uint4 o = *reinterpret_cast<const uint4*>(mine + rr * EPI_ROW + (lane & 15) * 16);
                    const size_t gidx = (size_t)((tile / ntn) * BM + wr * 128 + 16 * m + rr) * N + col0 + (lane & 15) * 8;
                    *reinterpret_cast<uint4*>(Cout + gidx) = o;
                }
            }
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[m][n] = v4f{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // (accumulator writes -> the MFMAs inside asm: no hazard pass sees them)
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the surplus DMA pieces and fragment reads)
#undef V5_PIECE
#undef V5_STAGE_DONE
#undef V5_ISSUE_ALL
#undef V5_FRAG
#undef V5_FRAG_LAST
#undef V5_READ_SW
#undef V5_QUAD
}

// ------------------------------------------------------------------------------------------------ host
static inline float bf2f_host(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline bf16_t f2bf_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
    return (bf16_t)u;
}

#ifdef LAB_F16
static inline bf16_t f2in_host(float f) { _Float16 h = (_Float16)f; bf16_t u; memcpy(&u, &h, 2); return u; }
static inline float in2f_host(bf16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
#else
#define f2in_host f2bf_host
#define in2f_host bf2f_host
#endif
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 98304, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    int mask = argc > 4 ? atoi(argv[4]) : 7, reps = argc > 5 ? atoi(argv[5]) : 20;
    int dbg = argc > 6 ? atoi(argv[6]) : 0;
    if (N % 256 || K % 128) {
        fprintf(stderr, "N must be a multiple of 256, K of 128\n");
        return 1;
    }
    printf("M=%d N=%d K=%d  (%.2f GFLOP)\n", M, N, K, 2.0 * M * N * K / 1e9);
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K);
    std::vector<float> hb(N);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0);
    };
    for (auto& v : hA) v = f2in_host(rnd());
    for (auto& v : hW) v = f2in_host(rnd() * 0.05f);
    for (auto& v : hb) v = rnd();
    bf16_t *dA, *dW, *dC;
    float* db;
    const size_t Mpad = ((size_t)M + 255) / 256 * 256 + 256;
    HIP_OK(hipMalloc(&dA, Mpad * K * 2));
    HIP_OK(hipMalloc(&dW, (size_t)N * K * 2));
    HIP_OK(hipMalloc(&dC, Mpad * N * 2));
    HIP_OK(hipMalloc(&db, N * 4));
    HIP_OK(hipMemset(dA, 0, Mpad * K * 2));
    HIP_OK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int ntiles = (N / 256) * ((M + 255) / 256);
    int grid = std::min(ntiles, ncu);
    grid = std::max(8, grid / 8 * 8);
    const size_t lds = 131072;
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v0<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v1<0, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v1<0, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v3<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v3<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_OK(hipFuncSetAttribute((const void*)k_gemm_v5<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto launch = [&](int v) {
        if (v == 5) hipLaunchKernelGGL(k_gemm_v5<0>, dim3(grid), dim3(256), lds, 0, dA, dW, db, dC, M, N, K, dbg);
        else if (v == 3) hipLaunchKernelGGL(k_gemm_v3<0>, dim3(grid), dim3(256), lds, 0, dA, dW, db, dC, M, N, K, dbg);
        else if (v == 4) hipLaunchKernelGGL(k_gemm_v3<1>, dim3(grid), dim3(256), lds, 0, dA, dW, db, dC, M, N, K, dbg);
        else if (v == 0) hipLaunchKernelGGL(k_gemm_v0<0>, dim3(grid), dim3(512), lds, 0, dA, dW, db, dC, M, N, K, dbg);
        else if (v == 1) hipLaunchKernelGGL((k_gemm_v1<0, false, 3>), dim3(grid), dim3(512), lds, 0, dA, dW, db, dC, M, N, K, dbg);
        else hipLaunchKernelGGL((k_gemm_v1<0, false, 0>), dim3(grid), dim3(512), lds, 0, dA, dW, db, dC, M, N, K, dbg);
    };
    // ---- correctness: sampled elements against an fp64 host reference
    std::vector<bf16_t> hC((size_t)M * N);
    for (int v = 0; v < 6; ++v) {
        if (!(mask & (1 << v))) continue;
        HIP_OK(hipMemset(dC, 0xFF, Mpad * N * 2));
        launch(v);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        int bad = 0;
        uint64_t t = 12345;
        const int samples = 20000;
        for (int i = 0; i < samples + 4 * 256; ++i) {
            int rr, cc;
            if (i < samples) {
                t = t * 6364136223846793005ull + 1442695040888963407ull;
                rr = (int)((t >> 33) % (uint64_t)M);
                cc = (int)((t >> 13) % (uint64_t)N);
            } else {  // whole rows at the corners of the first / last tiles
                const int j = i - samples;
                rr = (j / 256) == 0 ? 0 : ((j / 256) == 1 ? 255 : ((j / 256) == 2 ? M - 1 : M / 2 + 77));
                cc = (j % 256) * (N / 256);
            }
            double ref = hb[cc];
            for (int k2 = 0; k2 < K; ++k2) ref += (double)in2f_host(hA[(size_t)rr * K + k2]) * in2f_host(hW[(size_t)cc * K + k2]);
            const double got = bf2f_host(hC[(size_t)rr * N + cc]);
            const double err = fabs(got - ref);
            if (err > maxerr) maxerr = err;
            if (err > 0.02 + 0.01 * fabs(ref)) {
                if (bad < 5) printf("  v%d mismatch at (%d,%d): got %f ref %f\n", v, rr, cc, got, ref);
                ++bad;
            }
        }
        printf("v%d check: max |err| = %.4g, %d bad of %d samples\n", v, maxerr, bad, samples + 1024);
    }
    // ---- timing: interleaved rounds
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    double best[6] = {1e30, 1e30, 1e30, 1e30, 1e30, 1e30}, sum[6] = {0, 0, 0, 0, 0, 0};
    const int rounds = 5;
    for (int r = 0; r < rounds; ++r)
        for (int v = 0; v < 6; ++v) {
            if (!(mask & (1 << v))) continue;
            for (int i = 0; i < 3; ++i) launch(v);
            HIP_OK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) launch(v);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipEventSynchronize(e1));
            float ms;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            if (ms < best[v]) best[v] = ms;
            sum[v] += ms;
        }
    for (int v = 0; v < 6; ++v)
        if (mask & (1 << v))
            printf("v%d: best %.4f ms (%.0f TFLOP/s), mean %.4f ms (%.0f TFLOP/s)\n", v, best[v], 2.0 * M * N * K / best[v] / 1e9,
                   sum[v] / rounds, 2.0 * M * N * K / (sum[v] / rounds) / 1e9);
    return 0;
}
