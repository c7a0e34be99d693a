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
                               float keep_prob, uint64_t key, uint64_t index_offset) {
    const int64_t total = n_rows * f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / f, c = i - r * f;
        float v = x[r * ldx + c];
        // TF 1.x: div(x, keep_prob) * floor(keep_prob + uniform)
        v = __fdiv_rn(v, keep_prob);
        v = keep_entry(key, index_offset + (uint64_t)i, keep_prob) ? v : 0.0f;
        if (gate && !(gate[r * ldg + c] > 0.0f)) v = 0.0f;
        y[r * ldy + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// C[m x n] = A[m x k] . op(B) with v_mfma_f32_32x32x2_f32 (exact float32: a k-ordered fma chain).
// Block tile 128 x BN (BN = 32, 64, 128), 4 waves, K staged 16 deep through LDS in two buffers:
// the global loads of tile t+1 are issued into registers before the MFMAs of tile t and written to
// the other LDS buffer after them, one barrier per tile.  Operand tiles sit k-major in LDS so the 32
// lanes of one MFMA operand read consecutive floats.  Tiles are numbered so that the blocks sharing
// an XCD (ids equal modulo 8) walk the column tiles of one row panel one after the other.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;
constexpr int kBK = 16;
constexpr int kPad = 4;

template <int BN, bool TRANS_B>
__global__ __launch_bounds__(kBlock) void gemm_mfma_kernel(int64_t m, int64_t n, int64_t k,
                                                           const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ b, int64_t ldb,
                                                           float* __restrict__ c, int64_t ldc, int relu, int a_vec,
                                                           int b_vec, int64_t tiles_m, int64_t tiles_n) {
    constexpr int kWavesN = (BN >= 64) ? 2 : 1;
    constexpr int kWavesM = 4 / kWavesN;
    constexpr int MT = kBM / (kWavesM * 32);
    constexpr int NT = BN / (kWavesN * 32);
    constexpr int kBUnits = kBK * BN / 4;                      // float4 units of a B tile
    constexpr int kBPer = (kBUnits + kBlock - 1) / kBlock;     // per thread
    __shared__ float as[2][kBK][kBM + kPad];
    __shared__ float bs[2][kBK][BN + kPad];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / kWavesN;
    const int wn = wave % kWavesN;

    // XCD-aware tile numbering (bijective for any tile count)
    const int64_t n_tiles = tiles_m * tiles_n;
    const int64_t orig = blockIdx.x;
    const int64_t q = n_tiles / 8, r = n_tiles % 8, xcd = orig % 8;
    const int64_t tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
    const int64_t m0 = (tile / tiles_n) * kBM;
    const int64_t n0 = (tile % tiles_n) * BN;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging registers: A 2 x float4 per thread (row = tid / 2, 8 consecutive k), B kBPer x float4
    float ra[8];
    float rb[kBPer][4];
    const int a_row = tid >> 1;
    const int a_kq = (tid & 1) * 8;

    auto load_tile = [&](int64_t k0) {
        const int64_t gr = m0 + a_row;
        const float* ap = a + gr * lda + k0 + a_kq;
        if (gr < m && a_vec && k0 + a_kq + 8 <= k) {
            const float4 v0 = *reinterpret_cast<const float4*>(ap);
            const float4 v1 = *reinterpret_cast<const float4*>(ap + 4);
            ra[0] = v0.x; ra[1] = v0.y; ra[2] = v0.z; ra[3] = v0.w;
            ra[4] = v1.x; ra[5] = v1.y; ra[6] = v1.z; ra[7] = v1.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) ra[i] = (gr < m && k0 + a_kq + i < k) ? ap[i] : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < kBPer; ++p) {
            const int u = tid + p * kBlock;
            if (u < kBUnits) {
                if (TRANS_B) {  // B stored [n x k]: unit = (column, 4 consecutive k)
                    const int col = u / (kBK / 4);
                    const int kq = (u % (kBK / 4)) * 4;
                    const int64_t gc = n0 + col;
                    const float* bp = b + gc * ldb + k0 + kq;
                    if (gc < n && b_vec && k0 + kq + 4 <= k) {
                        const float4 v = *reinterpret_cast<const float4*>(bp);
                        rb[p][0] = v.x; rb[p][1] = v.y; rb[p][2] = v.z; rb[p][3] = v.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rb[p][i] = (gc < n && k0 + kq + i < k) ? bp[i] : 0.0f;
                    }
                } else {        // B stored [k x n]: unit = (k row, 4 consecutive columns)
                    const int kk = u / (BN / 4);
                    const int cq = (u % (BN / 4)) * 4;
                    const int64_t gk = k0 + kk, gc = n0 + cq;
                    const float* bp = b + gk * ldb + gc;
                    if (gk < k && b_vec && gc + 4 <= n) {
                        const float4 v = *reinterpret_cast<const float4*>(bp);
                        rb[p][0] = v.x; rb[p][1] = v.y; rb[p][2] = v.z; rb[p][3] = v.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) rb[p][i] = (gk < k && gc + i < n) ? bp[i] : 0.0f;
                    }
                }
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) as[buf][a_kq + i][a_row] = ra[i];
#pragma unroll
        for (int p = 0; p < kBPer; ++p) {
            const int u = tid + p * kBlock;
            if (u < kBUnits) {
                if (TRANS_B) {
                    const int col = u / (kBK / 4);
                    const int kq = (u % (kBK / 4)) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bs[buf][kq + i][col] = rb[p][i];
                } else {
                    const int kk = u / (BN / 4);
                    const int cq = (u % (BN / 4)) * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) bs[buf][kk][cq + i] = rb[p][i];
                }
            }
        }
    };

    const int64_t n_ktiles = (k + kBK - 1) / kBK;
    if (n_ktiles > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int64_t t = 0; t < n_ktiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < n_ktiles) load_tile((t + 1) * kBK);
#pragma unroll
        for (int kk = 0; kk < kBK; kk += 2) {
            const int ks = kk + (lane >> 5);
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = as[buf][ks][(wm * MT + i) * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = bs[buf][ks][(wn * NT + j) * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < n_ktiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t gc = n0 + (wn * NT + j) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = m0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (gr < m && gc < n) {
                    float v = acc[i][j][e];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    c[gr * ldc + gc] = v;
                }
            }
        }
}

template <int BN>
static int launch_gemm(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       int trans_b, float* c, int64_t ldc, int relu, hipStream_t st) {
    const int64_t tiles_m = ceil_div(m, kBM), tiles_n = ceil_div(n, BN);
    const int64_t n_tiles = tiles_m * tiles_n;
    if (n_tiles > INT32_MAX) return fail(GCN_ERR_BAD_ARG, "matrix too large for one launch");
    const int a_vec = (lda % 4 == 0) && (((uintptr_t)a & 15) == 0);
    const int b_vec = (ldb % 4 == 0) && (((uintptr_t)b & 15) == 0);
    if (trans_b)
        gemm_mfma_kernel<BN, true><<<(unsigned)n_tiles, kBlock, 0, st>>>(m, n, k, a, lda, b, ldb, c, ldc, relu, a_vec,
                                                                         b_vec, tiles_m, tiles_n);
    else
        gemm_mfma_kernel<BN, false><<<(unsigned)n_tiles, kBlock, 0, st>>>(m, n, k, a, lda, b, ldb, c, ldc, relu,
                                                                          a_vec, b_vec, tiles_m, tiles_n);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_dropout_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, const float* gate, int64_t ldg, float* y,
                    int64_t ldy, float keep_prob, uint64_t seed, uint64_t stream_id, uint64_t index_offset,
                    void* stream) {
    if (n_rows < 0 || f < 0 || ldx < f || ldy < f || (gate && ldg < f)) return fail(GCN_ERR_BAD_ARG, "bad shape");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    const int64_t total = n_rows * f;
    if (total == 0) return GCN_OK;
    if (!x || !y) return fail(GCN_ERR_BAD_ARG, "null pointer");
    int64_t blocks = ceil_div(total, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dropout_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, f, x, ldx, gate, ldg, y, ldy,
                                                                        keep_prob, stream_key(seed, stream_id), index_offset);
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
    if (n <= 32) return launch_gemm<32>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
    if (n <= 64) return launch_gemm<64>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
    return launch_gemm<128>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
}

}  // extern "C"
