// Dense pieces of the GCN layer: tf.nn.dropout (gcn/layers.py:185) and the feature transform
// H.W (gcn/layers.py:32-33) on the float32 MFMA pipe of gfx950.
#include "common.h"

namespace gcn {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// dense dropout (+ optional ReLU gate for the backward pass)
// ------------------------------------------------------------------------------------------------
__global__ void dropout_kernel(int64_t n_rows, int64_t f, const float* __restrict__ x, int64_t ldx,
                               const float* __restrict__ gate, int64_t ldg, float* __restrict__ y, int64_t ldy,
                               float keep_prob, uint64_t key) {
    const int64_t total = n_rows * f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / f, c = i - r * f;
        float v = x[r * ldx + c];
        // TF 1.x: div(x, keep_prob) * floor(keep_prob + uniform)
        v = __fdiv_rn(v, keep_prob);
        v = keep_entry(key, (uint64_t)i, keep_prob) ? v : 0.0f;
        if (gate && !(gate[r * ldg + c] > 0.0f)) v = 0.0f;
        y[r * ldy + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// C[m x n] = A[m x k] . op(B) with v_mfma_f32_32x32x2_f32 (exact float32, k-ordered fma chain)
// block tile 128 x (32*NT), 4 waves, each wave 32 rows x (32*NT) columns; K staged 16 at a time in LDS
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;
constexpr int kBK = 16;
constexpr int kPad = 4;

template <int NT>
__global__ __launch_bounds__(kBlock) void gemm_mfma_kernel(int64_t m, int64_t n, int64_t k,
                                                           const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ b, int64_t ldb, int trans_b,
                                                           float* __restrict__ c, int64_t ldc, int relu) {
    constexpr int kBN = 32 * NT;
    __shared__ float as[kBK][kBM + kPad];  // k-major: lanes of one MFMA operand read consecutive floats
    __shared__ float bs[kBK][kBN + kPad];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * kBM;
    const int64_t n0 = (int64_t)blockIdx.y * kBN;

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    for (int64_t k0 = 0; k0 < k; k0 += kBK) {
        // A tile: 128 rows x 16 k; thread -> (row = tid / 2, 8 consecutive k)
        {
            const int row = tid >> 1;
            const int kq = (tid & 1) * 8;
            const int64_t gr = m0 + row;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t gk = k0 + kq + i;
                as[kq + i][row] = (gr < m && gk < k) ? a[gr * lda + gk] : 0.0f;
            }
        }
        // B tile: 16 k x kBN columns
        for (int idx = tid; idx < kBK * kBN; idx += kBlock) {
            int kk, col;
            if (trans_b) {
                col = idx / kBK;
                kk = idx - col * kBK;
            } else {
                kk = idx / kBN;
                col = idx - kk * kBN;
            }
            const int64_t gk = k0 + kk, gc = n0 + col;
            float v = 0.0f;
            if (gk < k && gc < n) v = trans_b ? b[gc * ldb + gk] : b[gk * ldb + gc];
            bs[kk][col] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const int ks = kk + (lane >> 5);
            const float av = as[ks][wave * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float bv = bs[ks][j * 32 + (lane & 31)];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int64_t gc = n0 + j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t gr = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (gr < m && gc < n) {
                float v = acc[j][e];
                if (relu) v = v > 0.0f ? v : 0.0f;
                c[gr * ldc + gc] = v;
            }
        }
    }
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_dropout_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, const float* gate, int64_t ldg, float* y,
                    int64_t ldy, float keep_prob, uint64_t seed, uint64_t stream_id, void* stream) {
    if (n_rows < 0 || f < 0 || ldx < f || ldy < f || (gate && ldg < f)) return fail(GCN_ERR_BAD_ARG, "bad shape");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    const int64_t total = n_rows * f;
    if (total == 0) return GCN_OK;
    if (!x || !y) return fail(GCN_ERR_BAD_ARG, "null pointer");
    int64_t blocks = ceil_div(total, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dropout_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, f, x, ldx, gate, ldg, y, ldy,
                                                                        keep_prob, stream_key(seed, stream_id));
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_gemm_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                 int trans_b, float* c, int64_t ldc, int flags, void* stream) {
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    if (lda < k || ldc < n || ldb < (trans_b ? k : n)) return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (flags & ~GCN_RELU) return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if (m == 0 || n == 0) return GCN_OK;
    if (!c || (k > 0 && (!a || !b))) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int relu = (flags & GCN_RELU) ? 1 : 0;
    if (n <= 32) {
        dim3 grid((unsigned)ceil_div(m, kBM), (unsigned)ceil_div(n, 32));
        gemm_mfma_kernel<1><<<grid, kBlock, 0, st>>>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu);
    } else {
        dim3 grid((unsigned)ceil_div(m, kBM), (unsigned)ceil_div(n, 64));
        gemm_mfma_kernel<2><<<grid, kBlock, 0, st>>>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu);
    }
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

}  // extern "C"
