// Probe: semantics of ds_read_b64_tr_b16 and of the bf16 32x32x16 MFMA operand maps on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_tr(short* out) {
    __shared__ short lds[64 * 64];  // [row][64 cols], value = row*100 + col
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)((i / 64) * 100 + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x;
    const int g16 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3, fh = lane >> 5;
    const int row = 4 * fh + q4, col = 16 * g16 + 4 * p4;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + row * 64 + col));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = r[j];
}

// C = A(32x16) * B(16x32), A[i][k] = i*16+k (as bf16-exact small ints), B[k][j] = (k==j%16) ? 1 : 0 ...
__global__ void k_mfma(float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v8bf a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        a[j] = (__bf16)(float)(r * 2 + (k == 3 ? 1 : 0));      // A[row r][k]: asymmetric: extra 1 at k==3
        b[j] = (__bf16)(float)((k == 3) ? (r + 1) : 0);         // B[k][col r]: only k==3 row non zero = col+1
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

int main() {
    short* d; float* f;
    hipMalloc(&d, 64 * 4 * 2); hipMalloc(&f, 64 * 16 * 4);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, f);
    std::vector<short> h(256); std::vector<float> g(1024);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost); hipMemcpy(g.data(), f, 4096, hipMemcpyDeviceToHost);
    printf("tr-read: lane -> 4 elements (value = row*100+col); lane addr row=4*fh+q4, col=16*g16+4*p4\n");
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    // expected C[i][j] = A[i][3]*B[3][j] = (2i+1)*(j+1); check C/D map row=(reg&3)+8*(reg>>2)+4*(lane>>5), col=lane&31
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int rg = 0; rg < 16; ++rg) {
        int row = (rg & 3) + 8 * (rg >> 2) + 4 * (l >> 5), col = l & 31;
        float e = (2 * row + 1) * (col + 1);
        if (g[l * 16 + rg] != e) { if (bad < 5) printf("mfma mismatch lane %d reg %d got %g exp %g\n", l, rg, g[l*16+rg], e); ++bad; }
    }
    printf("mfma C/D + A/B map check: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return 0;
}
