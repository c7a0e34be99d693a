/* CPU restatement of the sparse x dense product behind gcn/layers.py:31
 * (tf.sparse_tensor_dense_matmul).  TEST INFRASTRUCTURE ONLY: linked by tests,
 * smoke() and bench.py's cpu_baseline leg, never by the product library.
 *
 * The upstream TensorFlow CPU kernel is not part of the reference tree; its
 * published algorithm is: out = 0; for each nonzero (m, k, a) in feed order:
 * out[m,:] += a * b[k,:], float32, one multiply and one add per element.  Both
 * functions below keep exactly that per-row order; the file is compiled with
 * -ffp-contract=off so no multiply-add is fused.
 */
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void row_product(int64_t i, int64_t f, const int64_t* indptr, const int32_t* indices,
                        const float* vals, const float* b, int64_t ldb, float* c, int64_t ldc, int relu) {
    float* out = c + i * ldc;
    for (int64_t j = 0; j < f; ++j) out[j] = 0.0f;
    for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {
        const float a = vals[k];
        const float* br = b + (int64_t)indices[k] * ldb;
        for (int64_t j = 0; j < f; ++j) {
            float prod = a * br[j];
            out[j] = out[j] + prod;
        }
    }
    if (relu)
        for (int64_t j = 0; j < f; ++j) out[j] = out[j] > 0.0f ? out[j] : 0.0f;
}

/* CSR form: rows may be split over threads, the order inside a row never changes. */
void oracle_spmm_csr_f32(int64_t n, int64_t f, const int64_t* indptr, const int32_t* indices, const float* vals,
                         const float* b, int64_t ldb, float* c, int64_t ldc, int relu, int threads) {
    if (threads <= 1) {
        for (int64_t i = 0; i < n; ++i) row_product(i, f, indptr, indices, vals, b, ldb, c, ldc, relu);
        return;
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
#endif
    for (int64_t i = 0; i < n; ++i) row_product(i, f, indptr, indices, vals, b, ldb, c, ldc, relu);
}

/* COO form: the literal single-threaded nnz loop over a (row, col, value) triple in feed order. */
void oracle_spmm_coo_f32(int64_t n, int64_t nnz, int64_t f, const int32_t* rows, const int32_t* cols,
                         const float* vals, const float* b, int64_t ldb, float* c, int64_t ldc) {
    for (int64_t i = 0; i < n; ++i) memset(c + i * ldc, 0, (size_t)f * sizeof(float));
    for (int64_t k = 0; k < nnz; ++k) {
        const float a = vals[k];
        float* out = c + (int64_t)rows[k] * ldc;
        const float* br = b + (int64_t)cols[k] * ldb;
        for (int64_t j = 0; j < f; ++j) {
            float prod = a * br[j];
            out[j] = out[j] + prod;
        }
    }
}
