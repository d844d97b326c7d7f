/*
 * CPU oracle for the implicit-ALS half-sweep, in C.  TEST INFRASTRUCTURE ONLY.
 *
 * A restatement of the reference's per-row exact solve, used (a) by the tests to
 * check the HIP path at sizes where the Python-loop oracle is too slow and (b) by
 * bench.py's cpu_baseline leg.  The product library never links or loads this file.
 *
 * Follows reclibwh/core/ALS.py:67-88 (ALSRef._run_single_step):
 *     Y2 = Y^T Y                                               (ALS.py:70)
 *     for each row u: p_idx = stored non-zero columns          (ALS.py:79)
 *         A  = Y2 + Yp^T (alpha * rp o Yp) + lamb I            (ALS.py:82)
 *         L  = inv(A)                                          (ALS.py:82)
 *         x  = L (Yp^T (cup o pp)),  cup = 1 + alpha rp, pp=1  (ALS.py:83-86)
 * and reclibwh/core/ALS.py:160-176 (als_run_regression) for the float32 variant,
 * which differs only in precision and in p = (r > 0) (ALS.py:169).
 *
 * inv() is a Gauss-Jordan elimination with partial pivoting (numpy's inv is LAPACK
 * getrf/getri; both are pivoted LU to rounding).  Pinned against golden vectors minted
 * from ALSRef itself: tests/test_oracle.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DEFINE_ORACLE(T, SUFFIX)                                                                  \
    /* out[K*K] = Y^T Y over n rows, accumulated in row order (ALS.py:70 / ALS.py:243) */         \
    void als_ref_gram_##SUFFIX(const T *Y, int64_t n, int K, T *out) {                            \
        for (int i = 0; i < K * K; ++i) out[i] = 0;                                               \
        for (int64_t r = 0; r < n; ++r) {                                                         \
            const T *y = Y + r * K;                                                               \
            for (int a = 0; a < K; ++a)                                                           \
                for (int b = 0; b < K; ++b) out[a * K + b] += y[a] * y[b];                        \
        }                                                                                         \
    }                                                                                             \
    /* in-place inverse of the K x K matrix A (row major) using W as K x K scratch; 0 = ok */     \
    static int invert_##SUFFIX(T *A, T *W, int K) {                                               \
        for (int i = 0; i < K; ++i)                                                               \
            for (int j = 0; j < K; ++j) W[i * K + j] = (i == j) ? (T)1 : (T)0;                    \
        for (int c = 0; c < K; ++c) {                                                             \
            int piv = c;                                                                          \
            T best = (T)fabs((double)A[c * K + c]);                                               \
            for (int r = c + 1; r < K; ++r) {                                                     \
                T v = (T)fabs((double)A[r * K + c]);                                              \
                if (v > best) { best = v; piv = r; }                                              \
            }                                                                                     \
            if (best == (T)0) return 1;                                                           \
            if (piv != c)                                                                         \
                for (int j = 0; j < K; ++j) {                                                     \
                    T t = A[c * K + j]; A[c * K + j] = A[piv * K + j]; A[piv * K + j] = t;        \
                    t = W[c * K + j]; W[c * K + j] = W[piv * K + j]; W[piv * K + j] = t;          \
                }                                                                                 \
            T d = (T)1 / A[c * K + c];                                                            \
            for (int j = 0; j < K; ++j) { A[c * K + j] *= d; W[c * K + j] *= d; }                 \
            for (int r = 0; r < K; ++r) {                                                         \
                if (r == c) continue;                                                             \
                T m = A[r * K + c];                                                               \
                if (m == (T)0) continue;                                                          \
                for (int j = 0; j < K; ++j) {                                                     \
                    A[r * K + j] -= m * A[c * K + j];                                             \
                    W[r * K + j] -= m * W[c * K + j];                                             \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
        memcpy(A, W, sizeof(T) * (size_t)K * K);                                                  \
        return 0;                                                                                 \
    }                                                                                             \
    /* Half-sweep over rows [row_begin, row_end).  Xout is the full [n_rows, K] matrix; only    */\
    /* the visited rows are written.  Y2 may be NULL (then it is computed from Yfix).           */\
    /* Returns the number of rows whose system could not be inverted.                           */\
    int als_ref_half_sweep_##SUFFIX(const int64_t *indptr, const int32_t *indices, const T *data, \
                                    int64_t row_begin, int64_t row_end, const T *Yfix,            \
                                    int64_t n_opp, int K, const T *Y2in, double lamb,             \
                                    double alpha, T *Xout, int nthreads) {                        \
        T *Y2 = (T *)malloc(sizeof(T) * (size_t)K * K);                                           \
        if (Y2in) memcpy(Y2, Y2in, sizeof(T) * (size_t)K * K);                                    \
        else als_ref_gram_##SUFFIX(Yfix, n_opp, K, Y2);                                           \
        int bad = 0;                                                                              \
        const T al = (T)alpha, lm = (T)lamb;                                                      \
        _Pragma("omp parallel num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : bad)")       \
        {                                                                                         \
            T *A = (T *)malloc(sizeof(T) * (size_t)K * K);                                        \
            T *W = (T *)malloc(sizeof(T) * (size_t)K * K);                                        \
            T *b = (T *)malloc(sizeof(T) * (size_t)K);                                            \
            _Pragma("omp for schedule(dynamic, 16)")                                              \
            for (int64_t u = row_begin; u < row_end; ++u) {                                       \
                memcpy(A, Y2, sizeof(T) * (size_t)K * K);                                         \
                for (int i = 0; i < K; ++i) b[i] = 0;                                             \
                for (int64_t e = indptr[u]; e < indptr[u + 1]; ++e) {                             \
                    const T r = data[e];                                                          \
                    if (r == (T)0) continue;               /* sps.find drops stored zeros */      \
                    const T *y = Yfix + (int64_t)indices[e] * K;                                  \
                    const T w = al * r, c = (T)1 + al * r;                                        \
                    for (int i = 0; i < K; ++i) {                                                 \
                        const T wy = w * y[i];                                                    \
                        for (int j = 0; j < K; ++j) A[i * K + j] += wy * y[j];                    \
                        b[i] += c * y[i];                                                         \
                    }                                                                             \
                }                                                                                 \
                for (int i = 0; i < K; ++i) A[i * K + i] += lm;                                   \
                if (invert_##SUFFIX(A, W, K)) { bad += 1; continue; }                             \
                T *x = Xout + u * K;                                                              \
                for (int i = 0; i < K; ++i) {                                                     \
                    T s = 0;                                                                      \
                    for (int j = 0; j < K; ++j) s += A[i * K + j] * b[j];                         \
                    x[i] = s;                                                                     \
                }                                                                                 \
            }                                                                                     \
            free(A); free(W); free(b);                                                            \
        }                                                                                         \
        free(Y2);                                                                                 \
        return bad;                                                                               \
    }

DEFINE_ORACLE(double, f64)
DEFINE_ORACLE(float, f32)

int als_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
