// scan_lab.hip -- development bench for a QUERY-STATIONARY int8 scan loop (kNN batch scan, main stage shape).
// Standalone: hipcc only, no library.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/scan_lab.hip -o tools/scan_lab
//   tools/scan_lab [rows [queries [reps [check]]]]
//
// The product's k_scan_coarse8 streams BOTH operands of every 256 x 256 tile through LDS: 24 fragment reads of 1 KiB
// per 64 MFMAs and wave, ~96 B/clk of the CU's 128 B/clk LDS port at the full MFMA rate -- it runs at ~45 % of the int8
// peak.  Here a wave keeps its 64 queries (768 int8 columns: 48 fragments = 192 registers, meant for AGPRs) in
// registers for its whole life; only the index rows go through LDS (a ring of 16-row groups, 12 KiB each, LDS-DMA), and
// every group is read by the 4 waves of the block: 12 fragment reads per 48 MFMAs and wave.  Two 4-wave blocks per CU.
// Variants: -DLAB_PURE (the MFMA stream alone: the chip's ceiling, 3.57 POPS), -DLAB_PRIO (s_setprio around the MFMAs),
// -DLAB_PP (explicit ping-pong of two wave groups in one 8-wave block: 2.63-2.67 ms vs 2.76), -DLAB_KS=4|8, -DLAB_G, -DLAB_RING.
// The lab's "epilogue" keeps a running maximum per (lane, query group) so that nothing is dead code; LAB_EPI=1 adds the
// scale-multiply-compare work of the product's epilogue.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define HIP_OK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
#ifndef LAB_KS
#define LAB_KS 12
#endif
constexpr int KS = LAB_KS;       // 64-byte K steps of an int8 row (12: 768 columns)
constexpr int NPW = KS / 4;      // DMA pieces per wave and group
constexpr int ROWB = 64 * KS;    // bytes per row
constexpr int GROUP = 16;        // rows per ring slot
#ifndef LAB_G
#define LAB_G 1
#endif
constexpr int G = LAB_G;                 // 16-row groups per ring slot = per barrier (1, 2 or 4: a divisor of a tile's 16)
constexpr int GSLOT = GROUP * ROWB;      // 12 KiB at KS = 12: [KS / 2 chunks of 128 B][16 rows][128 B], XOR-swizzled 16-byte columns
constexpr int SLOT = G * GSLOT;
#ifndef LAB_RING
#define LAB_RING 4
#endif
constexpr int RING = LAB_RING;
#ifndef LAB_EPI
#define LAB_EPI 1
#endif

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// src0 = 16 index rows (VGPRs, from LDS), src1 = 16 queries (resident: the first LAB_NA fragments of a wave in AGPRs, the
// rest in VGPRs -- with two waves per SIMD hipcc splits the 256 registers of a wave 128 / 128), accumulators in VGPRs
#ifndef LAB_NA
#define LAB_NA (4 * LAB_KS < 32 ? 4 * LAB_KS : 32)
#endif
#ifdef LAB_SWAP
#define LAB_F(J_, T_) ((3 - (J_)) * KS + (T_))
#else
#define LAB_F(J_, T_) ((J_) * KS + (T_))
#endif
#define LAB_MFMA_A(ACC_, A_, Q_) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC_) : "v"(A_), "a"(Q_))
#define LAB_MFMA_V(ACC_, A_, Q_) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(ACC_) : "v"(A_), "v"(Q_))
#define LAB_MFMA(ACC_, A_, J_, T_)                                       \
    if (LAB_F(J_, T_) < LAB_NA) LAB_MFMA_A(ACC_, A_, qf[J_][T_]); \
    else LAB_MFMA_V(ACC_, A_, qf[J_][T_]);

// grid: 8 * per_x blocks; block b: XCD b & 7, query tile (b >> 3) % nqt, row-tile stream (b & 7) + 8 * ((b >> 3) / nqt)
__global__ __launch_bounds__(256, 2) void k_scan_qreg(const signed char* __restrict__ x8, const signed char* __restrict__ q8,
                                                      const float* __restrict__ xs, const float* __restrict__ thr,
                                                      int* __restrict__ out_max, int* __restrict__ out_hits,
                                                      int64_t ntiles, int nqt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // RING slots
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 15, lg = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int slots = per_x / nqt;
    if (jx >= slots * nqt) return;
    const int qtile = jx % nqt;
    const int64_t u0 = xcd + 8 * (jx / nqt), ustep = 8 * slots;
    const int my_ntiles = u0 < ntiles ? (int)((ntiles - u0 + ustep - 1) / ustep) : 0;
    const int nsteps = my_ntiles * (256 / GROUP / G);
    if (nsteps == 0) return;

    // resident query fragments: query 256 qtile + 64 wave + 16 j + lq, bytes 64 t + 16 lg .. + 15
    v4i qf[4][KS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const signed char* src = q8 + (size_t)(qtile * 256 + wave * 64 + 16 * j + lq) * ROWB + 64 * t + 16 * lg;
            if (LAB_F(j, t) < LAB_NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[j][t]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[j][t]) : "v"(src) : "memory");   // (asm too: hipcc would wait for its own loads INSIDE the loop, draining the DMA ring)
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float my_thr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) my_thr[j] = thr[qtile * 256 + wave * 64 + 16 * j + lq];

    // DMA: a group = 12 pieces of 1 KiB (chunk kc = p >> 1 of 128 B, rows 8 (p & 1) .. + 7); wave w issues pieces 3 w .. 3 w + 2
    const int prow = lane >> 3, pchunk = lane & 7;
    unsigned lofs[NPW];
    unsigned ldst[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int p = NPW * wave + i, kc = p >> 1, srow = 8 * (p & 1) + prow;
        lofs[i] = (unsigned)srow * ROWB + (unsigned)kc * 128u + (unsigned)((pchunk ^ ((srow >> 1) & 7)) << 4);
        ldst[i] = (unsigned)(kc * 2048 + (p & 1) * 1024);
    }
    int is_step = 0;   // next group to issue
// (s_mov of a compiler-computed operand into m0 inside the statement: see G4_DMA in css_encoder_kernels.h)
#define LAB_DMA(SBASE_, VOFF_, LDS_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF_), "s"(SBASE_), "s"(LDS_) : "memory")
#define LAB_ISSUE()                                                                                                    \
    {                                                                                                                  \
        /* past the end: the last group once more, into a free slot (keeps the vmcnt arithmetic uniform) */           \
        const int src_ = min(is_step, nsteps - 1) * G;                                                                 \
        const int64_t tile_ = u0 + (int64_t)(src_ >> 4) * ustep;                                                       \
        const char* base_ = reinterpret_cast<const char*>(x8) + ((size_t)tile_ * 256 + (size_t)(src_ & 15) * GROUP) * ROWB; \
        const unsigned dst_ = smem_base + (unsigned)(is_step % RING) * SLOT;                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                                                             \
            _Pragma("unroll") for (int i_ = 0; i_ < NPW; ++i_)                                                         \
                LAB_DMA(base_ + g_ * GSLOT, lofs[i_], dst_ + g_ * GSLOT + ldst[i_]);                                   \
        }                                                                                                              \
        ++is_step;                                                                                                     \
    }
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll 1
    for (int i = 0; i < RING - 1; ++i) LAB_ISSUE()

    const int a_o0 = swz(lq, lg), a_o1 = a_o0 ^ 64;
    int run_max[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
    int hits = 0;
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        // own pieces of group s have landed when at most 3 (RING - 2) younger DMA instructions are outstanding
        constexpr int kWait = NPW * G * (RING - 2);
        if constexpr (kWait == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (kWait == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (kWait == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (kWait == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (kWait == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (kWait == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if constexpr (kWait == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (kWait == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        static_assert(kWait == 0 || kWait == 2 || kWait == 4 || kWait == 3 || kWait == 6 || kWait == 9 || kWait == 12 || kWait == 18 || kWait == 24, "wait count");
#ifndef LAB_PURE
        __builtin_amdgcn_s_barrier();   // everybody's pieces of group s are in LDS; everybody is done reading group s - 1
        LAB_ISSUE()                     // group s + RING - 1 -> the slot of group s - 1
#endif
#pragma unroll 1
      for (int g = 0; g < G; ++g) {
        const char* slot = smem + (s % RING) * SLOT + g * GSLOT;
        v4i acc[4] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
#define LAB_LD(T_) (*reinterpret_cast<const v4i*>(slot + ((T_) >> 1) * 2048 + (((T_) & 1) ? a_o1 : a_o0)))
        v4i a[4];   // fragment reads run three K steps ahead of the MFMAs
        a[0] = LAB_LD(0);
        a[1] = LAB_LD(1);
        a[2] = LAB_LD(2);
#ifdef LAB_PURE
        a[3] = LAB_LD(3);   // -DLAB_PURE: the MFMA stream alone (operands from these four reads; no barrier, DMA or reads in the loop): the chip's ceiling
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef LAB_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int t = 0; t < KS; ++t) {
#ifdef LAB_NOP
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#endif
#ifndef LAB_PURE
            if (t + 3 < KS) a[(t + 3) & 3] = LAB_LD(t + 3);
            if (t + 3 < KS) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (t + 2 < KS) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (t + 1 < KS) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            LAB_MFMA(acc[0], a[t & 3], 0, t)
            LAB_MFMA(acc[1], a[t & 3], 1, t)
            LAB_MFMA(acc[2], a[t & 3], 2, t)
            LAB_MFMA(acc[3], a[t & 3], 3, t)
        }
#undef LAB_LD
        // (the MFMAs are inline asm: the compiler's hazard recogniser does not pad a VALU read of their result)
#ifdef LAB_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");   // (the accumulators pass through: nothing reads them earlier)
#if LAB_EPI
        // lane: queries 16 j + lq, rows 4 lg + r of the group: score = acc * row scale against the query's (scaled) threshold
        {
            const int64_t tile_ = u0 + (int64_t)((s * G) >> 4) * ustep;
            const float4 sc4 = *reinterpret_cast<const float4*>(xs + (size_t)tile_ * 256 + (size_t)((s * G + g) & 15) * GROUP + 4 * lg);
            const float scl[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (float)acc[j][r] * scl[r];
                    hits += v >= my_thr[j] ? 1 : 0;
                    run_max[j] = max(run_max[j], acc[j][r]);
                }
        }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) run_max[j] = max(run_max[j], acc[j][r]);
#endif
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = run_max[j];
        m = max(m, __shfl_xor(m, 16));
        m = max(m, __shfl_xor(m, 32));
        if (lg == 0) atomicMax(&out_max[qtile * 256 + wave * 64 + 16 * j + lq], m);
    }
    if (hits) atomicAdd(out_hits, hits);
#undef LAB_ISSUE
#undef LAB_DMA
}


// -DLAB_PP: explicit ping-pong.  ONE block of 8 waves per CU, 512 queries; waves 0-3 (group A) and 4-7 (group B) share the
// SIMDs pairwise and alternate between an M phase (the 4 KS MFMAs of one 16-row group) and an E phase (its epilogue and,
// every other group, this wave group's DMA duty), B half a step behind A, one block barrier between half-steps: every SIMD
// has exactly one wave in its M phase at any time.  Even groups are fetched by A's waves, odd ones by B's; a wave group
// issues group j + PP_RING - 1 (A) / j + PP_RING (B) in the E phase of an odd group j and waits there for group j + 1 (A)
// / j + 2 (B): both land one barrier before A's M phase needs them.
#ifndef LAB_PP_RING
#define LAB_PP_RING 6
#endif
constexpr int PP_RING = LAB_PP_RING;   // even
__global__ __launch_bounds__(512, 1) void k_scan_pp(const signed char* __restrict__ x8, const signed char* __restrict__ q8,
                                                    int* __restrict__ out_max, int64_t ntiles, int nqt) {
    static_assert(PP_RING % 2 == 0 && KS == 12, "ring of an even number of groups; 3 pieces per issuing wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = wave >> 2, ws = wave & 3;
    const int lq = lane & 15, lg = lane >> 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, per_x = gridDim.x >> 3;
    const int slots = per_x / nqt;
    if (jx >= slots * nqt) return;
    const int qtile = jx % nqt;
    const int64_t u0 = xcd + 8 * (jx / nqt), ustep = 8 * slots;
    const int my_ntiles = u0 < ntiles ? (int)((ntiles - u0 + ustep - 1) / ustep) : 0;
    const int nsteps = my_ntiles * 16;
    if (nsteps == 0) return;
    const int qbase = qtile * 512 + wave * 64;
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int prow = lane >> 3, pchunk = lane & 7;
    unsigned lofs[3], ldst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = 3 * ws + i, kc = p >> 1, srow = 8 * (p & 1) + prow;
        lofs[i] = (unsigned)srow * ROWB + (unsigned)kc * 128u + (unsigned)((pchunk ^ ((srow >> 1) & 7)) << 4);
        ldst[i] = (unsigned)(kc * 2048 + (p & 1) * 1024);
    }
#define PP_DMA(SBASE_, VOFF_, LDS_) \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF_), "s"(SBASE_), "s"(LDS_) : "memory")
#define PP_ISSUE(GRP_)                                                                                                 \
    {                                                                                                                  \
        const int src_ = min((GRP_), nsteps - 1);   /* past the end: the last group once more, into a free slot */      \
        const int64_t tile_ = u0 + (int64_t)(src_ >> 4) * ustep;                                                       \
        const char* base_ = reinterpret_cast<const char*>(x8) + ((size_t)tile_ * 256 + (size_t)(src_ & 15) * GROUP) * ROWB; \
        const unsigned dst_ = smem_base + (unsigned)((GRP_) % PP_RING) * GSLOT;                                        \
        PP_DMA(base_, lofs[0], dst_ + ldst[0]);                                                                        \
        PP_DMA(base_, lofs[1], dst_ + ldst[1]);                                                                        \
        PP_DMA(base_, lofs[2], dst_ + ldst[2]);                                                                        \
    }
#define PP_WAIT()                                                                                                      \
    {                                                                                                                  \
        constexpr int kW = 3 * (PP_RING - 2) / 2;                                                                      \
        static_assert(kW == 3 || kW == 6 || kW == 9, "wait count");                                                    \
        if constexpr (kW == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                        \
        else if constexpr (kW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                   \
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                                                          \
    }
    // prologue: A fetches the even groups 0 .. PP_RING - 2, B the odd groups 1 .. PP_RING - 1
#pragma unroll 1
    for (int g = wg; g < PP_RING; g += 2) PP_ISSUE(g)
    v4i qf[4][KS];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const signed char* src = q8 + (size_t)(qbase + 16 * j + lq) * ROWB + 64 * t + 16 * lg;
            if (LAB_F(j, t) < LAB_NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[j][t]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[j][t]) : "v"(src) : "memory");
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (everything, the prologue's groups included)
    __builtin_amdgcn_s_barrier();                       // ... of every wave
    const int a_o0 = swz(lq, lg), a_o1 = a_o0 ^ 64;
    int run_max[4] = {INT32_MIN, INT32_MIN, INT32_MIN, INT32_MIN};
#define PP_LD(SLOT_, T_) (*reinterpret_cast<const v4i*>((SLOT_) + ((T_) >> 1) * 2048 + (((T_) & 1) ? a_o1 : a_o0)))
    v4i a[4];   // the first three fragments of a group are read in the E phase in front of its M phase
    a[0] = PP_LD(smem, 0);
    a[1] = PP_LD(smem, 1);
    a[2] = PP_LD(smem, 2);
    if (wg == 1) __builtin_amdgcn_s_barrier();   // B runs half a step behind A
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        // ---- M phase
        const char* slot = smem + (s % PP_RING) * GSLOT;
        v4i acc[4] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            if (t + 3 < KS) a[(t + 3) & 3] = PP_LD(slot, t + 3);
            if (t + 3 < KS) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else if (t + 2 < KS) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else if (t + 1 < KS) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            LAB_MFMA(acc[0], a[t & 3], 0, t)
            LAB_MFMA(acc[1], a[t & 3], 1, t)
            LAB_MFMA(acc[2], a[t & 3], 2, t)
            LAB_MFMA(acc[3], a[t & 3], 3, t)
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- E phase (the partner group is in its M phase)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) run_max[j] = max(run_max[j], acc[j][r]);
        if (wg == 0) {
            if (s & 1) {
                PP_ISSUE(s + PP_RING - 1)
            } else {   // group s + 2 (fetched by this wave group three steps ago) has landed
                static_assert(PP_RING == 6 || PP_RING == 8, "A's wait count");
                if constexpr (PP_RING == 6) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            }
        } else if (s & 1) {
            PP_ISSUE(s + PP_RING)
            PP_WAIT()   // group s + 2 has landed
        }
        {   // group s + 1 is in LDS and visible (its fetchers waited at least one barrier ago)
            const char* nslot = smem + ((s + 1) % PP_RING) * GSLOT;
            a[0] = PP_LD(nslot, 0);
            a[1] = PP_LD(nslot, 1);
            a[2] = PP_LD(nslot, 2);
        }
        __builtin_amdgcn_s_barrier();
    }
#undef PP_LD
    if (wg == 0) __builtin_amdgcn_s_barrier();   // balance B's extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = run_max[j];
        m = max(m, __shfl_xor(m, 16));
        m = max(m, __shfl_xor(m, 32));
        if (lg == 0) atomicMax(&out_max[qbase + 16 * j + lq], m);
    }
#undef PP_ISSUE
#undef PP_DMA
#undef PP_WAIT
}

// reference: max over rows of the int32 dot product, one block per query
__global__ void k_ref_max(const signed char* x8, const signed char* q8, int* out, int64_t nrows) {
    const int q = blockIdx.x;
    int best = INT32_MIN;
    for (int64_t r = threadIdx.x; r < nrows; r += blockDim.x) {
        int acc = 0;
        for (int c = 0; c < ROWB; ++c) acc += (int)x8[r * ROWB + c] * (int)q8[(size_t)q * ROWB + c];
        best = max(best, acc);
    }
    atomicMax(&out[q], best);
}

__global__ void k_fill(signed char* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // roughly bell-shaped int8 like quantised unit rows: sum of four uniform bytes
        const int v = (int)(h & 63) + (int)((h >> 8) & 63) + (int)((h >> 16) & 63) + (int)((h >> 24) & 63) - 126;
        p[i] = (signed char)max(-127, min(127, v));
    }
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 5000192;
    const int nq = argc > 2 ? atoi(argv[2]) : 1024;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    const int check = argc > 4 ? atoi(argv[4]) : 0;
    const int64_t ntiles = rows / 256;
#ifdef LAB_PP
    const int nqt = nq / 512;
#else
    const int nqt = nq / 256;
#endif
    signed char *x8, *q8;
    float *xs, *thr;
    int *omax, *oref, *ohits;
    HIP_OK(hipMalloc(&x8, (size_t)(ntiles * 256 + 256) * ROWB));
    HIP_OK(hipMalloc(&q8, (size_t)nq * ROWB));
    HIP_OK(hipMalloc(&xs, (size_t)(ntiles * 256 + 256) * 4));
    HIP_OK(hipMalloc(&thr, (size_t)nq * 4));
    HIP_OK(hipMalloc(&omax, (size_t)nq * 4));
    HIP_OK(hipMalloc(&oref, (size_t)nq * 4));
    HIP_OK(hipMalloc(&ohits, 4));
    k_fill<<<4096, 256>>>(x8, (size_t)(ntiles * 256 + 256) * ROWB, 17u);
    k_fill<<<256, 256>>>(q8, (size_t)nq * ROWB, 99u);
    std::vector<float> hs((size_t)ntiles * 256 + 256, 1.0f), ht(nq, 1e30f);
    HIP_OK(hipMemcpy(xs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(thr, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> init(nq, INT32_MIN);
    HIP_OK(hipMemcpy(omax, init.data(), nq * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(oref, init.data(), nq * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(ohits, 0, 4));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
#ifdef LAB_PP
    const int grid = cus / 8 * 8;
    const size_t lds = (size_t)PP_RING * GSLOT;
    HIP_OK(hipFuncSetAttribute((const void*)k_scan_pp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_scan_pp, 512, lds));
    hipFuncAttributes fa;
    HIP_OK(hipFuncGetAttributes(&fa, (const void*)k_scan_pp));
#define LAB_LAUNCH() hipLaunchKernelGGL(k_scan_pp, dim3(grid), dim3(512), lds, 0, x8, q8, omax, ntiles, nqt)
#else
    const int grid = cus / 8 * 8 * 2;
    const size_t lds = (size_t)RING * SLOT;
    HIP_OK(hipFuncSetAttribute((const void*)k_scan_qreg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_scan_qreg, 256, lds));
    hipFuncAttributes fa;
    HIP_OK(hipFuncGetAttributes(&fa, (const void*)k_scan_qreg));
#define LAB_LAUNCH() hipLaunchKernelGGL(k_scan_qreg, dim3(grid), dim3(256), lds, 0, x8, q8, xs, thr, omax, ohits, ntiles, nqt)
#endif
    printf("rows=%lld nq=%d grid=%d lds=%zu occupancy=%d blocks/CU regs=%d scratch=%zu\n", (long long)rows, nq, grid, lds, occ,
           fa.numRegs, (size_t)fa.localSizeBytes);
    LAB_LAUNCH();
    HIP_OK(hipDeviceSynchronize());
    if (check) {
        hipLaunchKernelGGL(k_ref_max, dim3(nq), dim3(256), 0, 0, x8, q8, oref, ntiles * 256);
        HIP_OK(hipDeviceSynchronize());
        std::vector<int> a(nq), b(nq);
        HIP_OK(hipMemcpy(a.data(), omax, nq * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(b.data(), oref, nq * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < nq; ++i) bad += a[i] != b[i];
        printf("check: %d of %d query maxima differ (first: %d vs %d)\n", bad, nq, a[0], b[0]);
        int shown = 0;
        for (int i = 0; i < nq && shown < 24; ++i)
            if (a[i] != b[i]) {
                printf("  q=%d (tile %d wave %d j %d lq %d): %d vs %d\n", i, i >> 8, (i >> 6) & 3, (i >> 4) & 3, i & 15, a[i], b[i]);
                ++shown;
            }
    }
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round) {
        HIP_OK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i)
            LAB_LAUNCH();
        HIP_OK(hipEventRecord(e1));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double ops = 2.0 * (double)(ntiles * 256) * ROWB * nq;
        printf("k_scan_qreg: %.3f ms  %.2f POPS (%.1f %% of 5)\n", ms, ops / ms / 1e12, ops / ms / 1e12 / 5 * 100);
    }
    return 0;
}
