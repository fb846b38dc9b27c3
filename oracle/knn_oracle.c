/*
 * knn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32 with an fp64 shadow) of the flat exact kNN the
 * reference reaches through faiss-cpu.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product path
 * (claude_semantic_search_amd/) never does and fails loudly without the HIP
 * library.
 *
 * faiss-cpu (>=1.11.0, pyproject.toml:9 of the reference; not vendored and not
 * installed here) is restated from its published semantics, anchored on the
 * reference's own call sites:
 *   - IndexFlatIP / IndexFlatL2 choice ............ src/storage.py:252-258
 *   - row normalise x / (||x||_2 + 1e-8) ........... src/storage.py:347-350
 *   - ids = insertion order from ntotal ............ src/storage.py:358-365
 *   - query normalise + reshape(1,-1).astype(f32) .. src/storage.py:424-429
 *   - search(q, k) -> (D desc IP | asc squared L2) . src/storage.py:436
 * Pinned by the reference's own known-answer tests (tests/test_storage.py:277-308,
 * tests/test_integration.py:141-226, :312-353, tests/test_environment_setup.py:199-220)
 * through tests/test_oracle_knn.py and tests/golden/knn_reference_cases.json.
 *
 * Tie order is implementation defined in faiss; this oracle (and the HIP
 * kernels) fix (score better first, then lower id first).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_METRIC_IP 0
#define ORACLE_METRIC_L2 1

/* x / (||x||_2 + 1e-8), norm = sqrt(sum x^2) in fp32 -- src/storage.py:349-350
 * (np.linalg.norm on a float32 array reduces in float32). */
void knn_oracle_normalize_rows(float* x, int64_t n, int d) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float* row = x + r * (int64_t)d;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = 0; j < d; ++j) acc[j & 7] += row[j] * row[j];
        float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        /* numpy divides element-wise; keep the division to follow the reference. */
        float nrm = sqrtf(s) + 1e-8f;
        for (int j = 0; j < d; ++j) row[j] = row[j] / nrm;
    }
}

static inline float dot8(const float* a, const float* b, int d) {
    /* eight partial sums = one 256-bit SIMD accumulator, the shape of
     * faiss' fvec_inner_product; association order is unspecified upstream. */
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int u = 0; u < 8; ++u) acc[u] += a[j + u] * b[j + u];
    for (; j < d; ++j) acc[j & 7] += a[j] * b[j];
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

static inline float l2sqr8(const float* a, const float* b, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int j = 0;
    for (; j + 8 <= d; j += 8)
        for (int u = 0; u < 8; ++u) {
            float t = a[j + u] - b[j + u];
            acc[u] += t * t;
        }
    for (; j < d; ++j) {
        float t = a[j] - b[j];
        acc[j & 7] += t * t;
    }
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

/* "a is a better hit than b": IP larger score first, L2 smaller distance
 * first; ties -> lower id. */
static inline int better(int metric, float sa, int64_t ia, float sb, int64_t ib) {
    if (sa != sb) return metric == ORACLE_METRIC_IP ? (sa > sb) : (sa < sb);
    return ia < ib;
}

/* Exact top-k of xb[n,d] for every query.  D[nq,k], I[nq,k]; missing slots
 * I=-1, D=-FLT_MAX (IP) / +FLT_MAX (L2) as faiss pads. */
void knn_oracle_search(const float* xb, int64_t n, int d, const float* q, int64_t nq, int k,
                       int metric, float* D, int64_t* I) {
    const float pad = metric == ORACLE_METRIC_IP ? -FLT_MAX : FLT_MAX;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* qv = q + qi * (int64_t)d;
        float* Dq = D + qi * (int64_t)k;
        int64_t* Iq = I + qi * (int64_t)k;
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            Dq[j] = pad;
            Iq[j] = -1;
        }
        for (int64_t r = 0; r < n; ++r) {
            const float* row = xb + r * (int64_t)d;
            float s = metric == ORACLE_METRIC_IP ? dot8(row, qv, d) : l2sqr8(row, qv, d);
            if (cnt == k && !better(metric, s, r, Dq[k - 1], Iq[k - 1])) continue;
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && better(metric, s, r, Dq[pos - 1], Iq[pos - 1])) {
                Dq[pos] = Dq[pos - 1];
                Iq[pos] = Iq[pos - 1];
                --pos;
            }
            Dq[pos] = s;
            Iq[pos] = r;
            if (cnt < k) ++cnt;
        }
    }
}

/* fp64 re-score of given ids (shadow precision for near-tie analysis). */
void knn_oracle_rescore64(const float* xb, int d, const float* q, int64_t nq, int k, int metric,
                          const int64_t* I, double* D64) {
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* qv = q + qi * (int64_t)d;
        for (int j = 0; j < k; ++j) {
            int64_t id = I[qi * (int64_t)k + j];
            double s = 0.0;
            if (id < 0) {
                D64[qi * (int64_t)k + j] = metric == ORACLE_METRIC_IP ? -DBL_MAX : DBL_MAX;
                continue;
            }
            const float* row = xb + id * (int64_t)d;
            if (metric == ORACLE_METRIC_IP)
                for (int c = 0; c < d; ++c) s += (double)row[c] * (double)qv[c];
            else
                for (int c = 0; c < d; ++c) {
                    double t = (double)row[c] - (double)qv[c];
                    s += t * t;
                }
            D64[qi * (int64_t)k + j] = s;
        }
    }
}

/* Merge per-shard top-k lists ([nparts, nq, k]) -> global top-k, same
 * comparator (mirrors the multi-GPU exchange of SURVEY.md 8e). */
void knn_oracle_merge(const float* Dp, const int64_t* Ip, int nparts, int64_t nq, int k, int metric,
                      float* D, int64_t* I) {
    const float pad = metric == ORACLE_METRIC_IP ? -FLT_MAX : FLT_MAX;
    for (int64_t qi = 0; qi < nq; ++qi) {
        float* Dq = D + qi * (int64_t)k;
        int64_t* Iq = I + qi * (int64_t)k;
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            Dq[j] = pad;
            Iq[j] = -1;
        }
        for (int p = 0; p < nparts; ++p)
            for (int j = 0; j < k; ++j) {
                int64_t off = ((int64_t)p * nq + qi) * k + j;
                float s = Dp[off];
                int64_t id = Ip[off];
                if (id < 0) continue;
                if (cnt == k && !better(metric, s, id, Dq[k - 1], Iq[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && better(metric, s, id, Dq[pos - 1], Iq[pos - 1])) {
                    Dq[pos] = Dq[pos - 1];
                    Iq[pos] = Iq[pos - 1];
                    --pos;
                }
                Dq[pos] = s;
                Iq[pos] = id;
                if (cnt < k) ++cnt;
            }
    }
}

/* Heap phase of faiss-cpu's batched search (nq >= 20: blocked SGEMM, then every score of the block is offered to
 * the query's result heap): S[nq, nb] holds the scores of rows [r0, r0 + nb) -- inner products, or squared
 * distances for L2 -- and is folded into the running lists D/I[nq, k] (sorted, best first; the caller starts them
 * at pad / -1).  One thread per query, the common case (score no better than the current k-th) is one compare. */
void knn_oracle_fold_block(const float* S, int64_t nq, int64_t nb, int64_t r0, int k, int metric, float* D,
                           int64_t* I) {
#pragma omp parallel for schedule(static)
    for (int64_t qi = 0; qi < nq; ++qi) {
        const float* s = S + qi * nb;
        float* Dq = D + qi * (int64_t)k;
        int64_t* Iq = I + qi * (int64_t)k;
        int cnt = 0;
        while (cnt < k && Iq[cnt] >= 0) ++cnt;
        for (int64_t j = 0; j < nb; ++j) {
            const float v = s[j];
            if (cnt == k) {
                const float kth = Dq[k - 1];
                if (metric == ORACLE_METRIC_IP ? (v < kth) : (v > kth)) continue;
                if (!better(metric, v, r0 + j, kth, Iq[k - 1])) continue;
            }
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && better(metric, v, r0 + j, Dq[pos - 1], Iq[pos - 1])) {
                Dq[pos] = Dq[pos - 1];
                Iq[pos] = Iq[pos - 1];
                --pos;
            }
            Dq[pos] = v;
            Iq[pos] = r0 + j;
            if (cnt < k) ++cnt;
        }
    }
}

int knn_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void knn_oracle_set_threads(int t) {
#ifdef _OPENMP
    omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* Synthetic rows from include/css_synth.h (same generator as the device). */
#include "../include/css_synth.h"
void knn_oracle_synth_rows(float* x, int64_t n, int d, uint64_t seed, int64_t first_row) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int c = 0; c < d; ++c)
            x[r * (int64_t)d + c] = css_synth_normal(seed, (uint64_t)((first_row + r) * (int64_t)d + c));
}
