// css_knn_kernels.h -- device helpers shared by the flat-index kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace css {

constexpr uint32_t kInvalidRow = 0xFFFFFFFFu;

// Order preserving float <-> int key (for atomicMax on thresholds).
__device__ __forceinline__ int f2key(float f) {
    int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

// DPP move within a 16-lane row (ctrl is an immediate).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// All-reduce (sum) over each group of 16 consecutive lanes: 4 DPP adds.
// quad xor1 (0xB1), quad xor2 (0x4E), row_half_mirror (0x141), row_mirror (0x140).
__device__ __forceinline__ float row16_allsum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    return v;
}

__device__ __forceinline__ float wave_allsum(float v) {
    v = row16_allsum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ float wave_allmax(float v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}

// "a is a better hit than b": larger score first, then lower id.
template <typename IdT>
__device__ __forceinline__ bool better(float sa, IdT ia, float sb, IdT ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// Wave-cooperative insertion into a best-first sorted list of k <= 128 entries
// living in LDS (or global).  Empty slots hold (-inf, max id).  All 64 lanes
// of the wave must call this with identical (s, id).  Returns true if inserted.
template <typename IdT>
__device__ __forceinline__ bool wave_insert(float* S, IdT* I, int k, float s, IdT id, int lane) {
    const int p0 = lane, p1 = lane + 64;
    float s0 = -INFINITY, s1 = -INFINITY;
    IdT i0 = (IdT)~(IdT)0, i1 = (IdT)~(IdT)0;
    bool v0 = p0 < k, v1 = p1 < k;
    if (v0) {
        s0 = S[p0];
        i0 = I[p0];
    }
    if (v1) {
        s1 = S[p1];
        i1 = I[p1];
    }
    const bool b0 = v0 && better<IdT>(s0, i0, s, id);
    const bool b1 = v1 && better<IdT>(s1, i1, s, id);
    const int pos = __popcll(__ballot(b0)) + __popcll(__ballot(b1));
    if (pos >= k) return false;
    // shift [pos, k-2] -> [pos+1, k-1]; all reads above precede the writes below
    if (v0 && p0 >= pos && p0 + 1 < k) {
        S[p0 + 1] = s0;
        I[p0 + 1] = i0;
    }
    if (v1 && p1 >= pos && p1 + 1 < k) {
        S[p1 + 1] = s1;
        I[p1 + 1] = i1;
    }
    if (lane == 0) {
        S[pos] = s;
        I[pos] = id;
    }
    return true;
}

}  // namespace css
