// Dense pieces of the GCN layer: tf.nn.dropout (gcn/layers.py:185) and the feature transform
// H.W (gcn/layers.py:32-33) on the float32 MFMA pipe of gfx950.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace gcn {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// dense dropout (+ optional ReLU gate for the backward pass)
// ------------------------------------------------------------------------------------------------
__global__ void dropout_kernel(int64_t n_rows, int64_t f, const float* __restrict__ x, int64_t ldx,
                               const float* __restrict__ gate, int64_t ldg, float* __restrict__ y, int64_t ldy,
                               float keep_prob, uint32_t threshold, uint64_t key, uint64_t index_offset,
                               const uint64_t* __restrict__ stream_offset, uint64_t seed, uint64_t stream_id) {
    if (stream_offset) key = stream_key(seed, stream_id + *stream_offset);  // per replay of a recorded launch
    const int64_t total = n_rows * f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / f, c = i - r * f;
        float v = x[r * ldx + c];
        // TF 1.x: div(x, keep_prob) * floor(keep_prob + uniform)
        v = __fdiv_rn(v, keep_prob);
        const uint64_t id = index_offset + (uint64_t)i;
        v = keep_field(decision_hash(key, id), id, threshold) ? v : 0.0f;
        if (gate && !(gate[r * ldg + c] > 0.0f)) v = 0.0f;
        y[r * ldy + c] = v;
    }
}

// The same on contiguous matrices (row stride = width, 16-byte aligned, element count a multiple of 4):
// 128-bit accesses, no division per element, and no hash at all when keep_prob is 1 (the pure
// ReLU-gate pass of the backward).
template <bool RNG>
__global__ void dropout_vec_kernel(int64_t n_vec, const float4* __restrict__ x, const float4* __restrict__ gate,
                                   float4* __restrict__ y, float keep_prob, uint32_t threshold, uint64_t key,
                                   uint64_t index_offset, const uint64_t* __restrict__ stream_offset, uint64_t seed,
                                   uint64_t stream_id) {
    if (RNG && stream_offset) key = stream_key(seed, stream_id + *stream_offset);
    const bool one_hash = (index_offset & 3) == 0;     // a vector's 4 ids then start at a multiple of 4: one hash
    // a block takes contiguous tiles of kDepth x 256 vectors (16 KB), kDepth independent 128-bit loads in
    // flight per thread; consecutive blocks take consecutive tiles, so the accesses in flight at any
    // time cover one contiguous stretch of memory (6.2 TB/s; a grid-stride loop over single vectors
    // with the grid capped at 8192 blocks ran at 4.8 TB/s)
    constexpr int kDepth = 4;
    const int64_t stride = blockDim.x;
    const int64_t tile = (int64_t)kDepth * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * tile + threadIdx.x; i0 < n_vec; i0 += (int64_t)gridDim.x * tile) {
        float4 v[kDepth], g[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < n_vec) {
                v[u] = x[i];
                if (gate) g[u] = gate[i];
            }
        }
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n_vec) break;
            float o[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (RNG) {
                const uint64_t id = index_offset + 4 * (uint64_t)i;
                const uint64_t h4 = one_hash ? decision_hash(key, id) : 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint64_t h = one_hash ? h4 : decision_hash(key, id + e);
                    o[e] = __fdiv_rn(o[e], keep_prob);
                    o[e] = keep_field(h, id + e, threshold) ? o[e] : 0.0f;
                }
            }
            if (gate) {
                if (!(g[u].x > 0.0f)) o[0] = 0.0f;
                if (!(g[u].y > 0.0f)) o[1] = 0.0f;
                if (!(g[u].z > 0.0f)) o[2] = 0.0f;
                if (!(g[u].w > 0.0f)) o[3] = 0.0f;
            }
            y[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// The decisions of tf.nn.dropout on a matrix [n_rows x f], one bit per element: bit c % 32 of word c / 32 of
// row r is 1 iff element (r, c) is kept -- decision index_offset + r * pitch + c of the stream, the one
// dropout_kernel draws.  One thread per word (32 hashes); columns past f give 0 bits.
__global__ void dropout_bits_kernel(int64_t n_rows, int64_t f, int64_t words_per_row, uint32_t threshold, uint64_t key,
                                    uint64_t index_offset, int64_t pitch, uint32_t* __restrict__ bits, int64_t ld_words,
                                    const uint64_t* __restrict__ stream_offset, uint64_t seed, uint64_t stream_id) {
    if (stream_offset) key = stream_key(seed, stream_id + *stream_offset);
    const int64_t total = n_rows * words_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / words_per_row, w = i - r * words_per_row;
        const int64_t c0 = w * 32;
        const uint64_t id0 = index_offset + (uint64_t)(r * pitch + c0);
        uint32_t word = 0;
        if ((id0 & 3) == 0) {                 // the word's 32 ids are 8 whole groups of 4: 8 hashes
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint64_t h = decision_hash(key, id0 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (keep_field(h, (uint64_t)e, threshold)) word |= 1u << (4 * q + e);
            }
            if (c0 + 32 > f) word &= (f - c0 >= 32) ? 0xffffffffu : ((1u << (int)(f - c0)) - 1u);
        } else {
#pragma unroll 8
            for (int e = 0; e < 32; ++e) {
                const uint64_t id = id0 + (uint64_t)e;
                if (c0 + e < f && keep_field(decision_hash(key, id), id, threshold)) word |= 1u << e;
            }
        }
        bits[r * ld_words + w] = word;
    }
}

// A masked, scaled element of the left operand: tf.nn.dropout's `div(x, keep_prob) * floor(keep_prob + u)`
// with the decision read from a mask bit.  A keep probability that is a power of two divides exactly as a
// product with its inverse (one instruction instead of a correctly rounded division).
struct InMask {
    const uint32_t* bits;   // null: no input dropout
    int64_t ld_words;
    float keep_prob, inv_keep;
    int exact;              // keep_prob is a power of two
};
__device__ __forceinline__ float masked_scaled(float x, uint32_t word, int bit, const InMask& im) {
    const float q = im.exact ? x * im.inv_keep : __fdiv_rn(x, im.keep_prob);
    return ((word >> bit) & 1u) ? q : 0.0f;
}
__device__ __forceinline__ float4 masked_scaled4(float4 v, uint32_t word, int shift, const InMask& im) {
    return make_float4(masked_scaled(v.x, word, shift, im), masked_scaled(v.y, word, shift + 1, im),
                       masked_scaled(v.z, word, shift + 2, im), masked_scaled(v.w, word, shift + 3, im));
}

// dst[i, :] = src[rows[i], :] for a column range of a row-major matrix: the send buffer of the
// multi-GPU exchange (the rows of the local block that other ranks reference).  One thread per
// float4 (or per float when the layout does not allow 128-bit accesses).
template <int VEC>
__global__ void gather_rows_kernel(int64_t n_out, int64_t f, const float* __restrict__ src, int64_t lds_,
                                   const int64_t* __restrict__ rows, float* __restrict__ dst, int64_t ldd) {
    const int64_t per_row = (f + VEC - 1) / VEC;
    const int64_t total = n_out * per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / per_row, c = (i - r * per_row) * VEC;
        const float* s = src + rows[r] * lds_ + c;
        float* d = dst + r * ldd + c;
        if (VEC == 4) {
            *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
        } else {
            d[0] = s[0];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C[m x n] = A[m x k] . op(B) with v_mfma_f32_32x32x2_f32 (exact float32 products and sums).
// Block tile 128 x BN (BN = 32, 64, 128), 4 waves, K staged 32 deep through one LDS buffer (35 KB).
// The kernel is bounded to 168 registers (64 of them the accumulators) so that 3 blocks share a CU:
// it is the other blocks that cover a block's barriers and its global-load latency (at 2 blocks per
// CU: 124 instead of 130 TFLOP/s on 2^22 x 512 x 512; the few spills this costs sit outside the k
// loop; 4 blocks would need 128 registers and spill inside it).  A second LDS buffer at lower
// occupancy measured the same or slower, 256 x 128 and 128 x 256 tiles 15 % slower, 128 x 64 tiles at 4
// blocks per CU 2 % slower, and persistent blocks that run the k stages of successive tiles as one
// pipeline 3-6 % slower (the hardware's block dispatch balances better than a static tile list).  The loads of tile t+1 go to registers before the MFMAs of tile t and to LDS after them.
//  - A sits row-major in LDS as it does in memory (pitch 36 floats: conflict-free 128-bit reads and
//    writes).  The MFMA wants, per lane, A[row = lane % 32][k = 2 s + lane / 32] at step s; instead
//    lanes 0-31 take k = 0..15 of the stage and lanes 32-63 k = 16..31, so a lane reads 4
//    consecutive k with one ds_read_b128.  B uses the same k mapping, which makes every step an
//    exact rank-2 update; only the order in which the 32 products of a stage are added differs
//    from ascending k.
//  - Interior blocks (whole tile in range, 16-byte aligned rows) run a main loop without any bounds
//    test: a guarded load is a branch per element, and branches between the loads and the MFMAs
//    keep the compiler from overlapping them (85 -> 121 TFLOP/s on 2^22 x 512 x 512).  The last
//    partial k tile and edge blocks take the guarded loader.
//  - Tiles are numbered so that the blocks sharing an XCD (ids equal modulo 8) walk the column
//    tiles of one row panel one after the other.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// kBK = 32, or 16 for products whose whole K fits 16 (hidden -> classes)
template <int kBM, int BN, bool TRANS_B, int kBK>
__global__ __launch_bounds__(kBlock, 3) void gemm_mfma_kernel(int64_t m, int64_t n, int64_t k,
                                                           const float* __restrict__ a, int64_t lda,
                                                           const float* __restrict__ b, int64_t ldb,
                                                           float* __restrict__ c, int64_t ldc, int relu, int a_vec,
                                                           int b_vec, int64_t tiles_m, int64_t tiles_n,
                                                           const float* __restrict__ gate, int64_t ldg,
                                                           float divisor, OutDrop od, InMask im) {
    constexpr int kWavesN = (BN >= 64) ? 2 : 1;
    constexpr int kWavesM = 4 / kWavesN;
    constexpr int MT = kBM / (kWavesM * 32);
    constexpr int NT = BN / (kWavesN * 32);
    constexpr int kAPitch = kBK + 4;
    constexpr int kBPitch = BN + 4;
    constexpr int kKQ = kBK / 4;                            // float4 units along k
    constexpr int kARows = kBlock / kKQ;                    // A rows loaded per pass
    constexpr int kAPer = kBM / kARows;                     // float4 units of an A tile per thread
    constexpr int kBUnits = kBK * BN / 4;                   // float4 units of a B tile
    constexpr int kBPer = (kBUnits + kBlock - 1) / kBlock;  // per thread
    constexpr bool kBAll = kBUnits % kBlock == 0;           // every thread has kBPer units
    __shared__ __attribute__((aligned(16))) float as[kBM * kAPitch];
    // B tile: k-major [kBK][BN + 4] when B is stored [k x n]; column-major [BN][kBK + 4] (the layout of
    // the A tile) when B is stored [n x k], so that it too is written and read 128 bits at a time
    __shared__ __attribute__((aligned(16))) float bs[TRANS_B ? BN * kAPitch : kBK * kBPitch];
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

    // staging registers.  A: kAPer x float4 per thread (rows tid / kKQ + kARows i, k quad tid % kKQ).
    // B [k x n]: unit u = tid + 256 p -> k row u / (BN / 4), column quad u % (BN / 4);
    // B [n x k] (TRANS_B): column u / kKQ, k quad u % kKQ.
    float4 ra[kAPer], rb[kBPer];
    const int a_r = tid / kKQ, a_kq = (tid % kKQ) * 4;

    // loaders: A and B separately, so that a narrow output (n < BN: every tile is an edge tile in n) still
    // streams A -- nearly all of the bytes -- without a bounds test per element
    // input dropout on A (im.bits): a float4 holds 4 consecutive k of one row, all inside one mask word
    auto mask_a = [&](int i, int64_t gr, int64_t k0) {
        if (im.bits) {
            const int64_t kk = k0 + a_kq;
            const uint32_t word = (gr < m && kk < k) ? im.bits[gr * im.ld_words + (kk >> 5)] : 0u;
            ra[i] = masked_scaled4(ra[i], word, (int)(kk & 31), im);
        }
    };
    auto load_a_fast = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < kAPer; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(a + (m0 + a_r + kARows * i) * lda + k0 + a_kq);
            mask_a(i, m0 + a_r + kARows * i, k0);
        }
    };
    auto load_b_fast = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < kBPer; ++p) {
            const int u = tid + kBlock * p;
            if (!kBAll && u >= kBUnits) continue;
            if (TRANS_B)
                rb[p] = *reinterpret_cast<const float4*>(b + (n0 + u / kKQ) * ldb + k0 + (u % kKQ) * 4);
            else
                rb[p] = *reinterpret_cast<const float4*>(b + (k0 + u / (BN / 4)) * ldb + n0 + (u % (BN / 4)) * 4);
        }
    };
    auto load_a_guarded = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < kAPer; ++i) {
            const int64_t gr = m0 + a_r + kARows * i;
            const float* ap = a + gr * lda + k0 + a_kq;
            if (gr < m && a_vec && k0 + a_kq + 4 <= k) {
                ra[i] = *reinterpret_cast<const float4*>(ap);
            } else {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (gr < m && k0 + a_kq + e < k) ? ap[e] : 0.0f;
                ra[i] = make_float4(t[0], t[1], t[2], t[3]);
            }
            mask_a(i, gr, k0);
        }
    };
    auto load_b_guarded = [&](int64_t k0) {
#pragma unroll
        for (int p = 0; p < kBPer; ++p) {
            const int u = tid + kBlock * p;
            if (!kBAll && u >= kBUnits) continue;
            float t[4];
            if (TRANS_B) {
                const int64_t gc = n0 + u / kKQ;
                const int kq = (u % kKQ) * 4;
                const float* bp = b + gc * ldb + k0 + kq;
                if (gc < n && b_vec && k0 + kq + 4 <= k) {
                    rb[p] = *reinterpret_cast<const float4*>(bp);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (gc < n && k0 + kq + e < k) ? bp[e] : 0.0f;
            } else {
                const int64_t gk = k0 + u / (BN / 4), gc = n0 + (u % (BN / 4)) * 4;
                const float* bp = b + gk * ldb + gc;
                if (gk < k && b_vec && gc + 4 <= n) {
                    rb[p] = *reinterpret_cast<const float4*>(bp);
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (gk < k && gc + e < n) ? bp[e] : 0.0f;
            }
            rb[p] = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < kAPer; ++i) *reinterpret_cast<float4*>(as + (a_r + kARows * i) * kAPitch + a_kq) = ra[i];
#pragma unroll
        for (int p = 0; p < kBPer; ++p) {
            const int u = tid + kBlock * p;
            if (!kBAll && u >= kBUnits) continue;
            if (TRANS_B) {
                const int col = u / kKQ, kq = (u % kKQ) * 4;
                *reinterpret_cast<float4*>(bs + col * kAPitch + kq) = rb[p];
            } else {
                *reinterpret_cast<float4*>(bs + (u / (BN / 4)) * kBPitch + (u % (BN / 4)) * 4) = rb[p];
            }
        }
    };
    const int kh = (lane >> 5) * (kBK / 2);  // this lane's half of the stage's k range
    const int l31 = lane & 31;
    // fragments of 4 consecutive k steps: A as one 128-bit read per 32-row tile, B as 4 scalar reads per
    // 32-column tile.  The fragments of group g+1 are requested before the 4 x MT x NT MFMAs of group g.
    auto read_frags = [&](int kq, float4 (&av)[MT], float (&bv)[NT][4]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
            av[i] = *reinterpret_cast<const float4*>(as + ((wm * MT + i) * 32 + l31) * kAPitch + kh + kq);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            if (TRANS_B) {
                const float4 t = *reinterpret_cast<const float4*>(bs + ((wn * NT + j) * 32 + l31) * kAPitch + kh + kq);
                bv[j][0] = t.x;
                bv[j][1] = t.y;
                bv[j][2] = t.z;
                bv[j][3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = bs[(kh + kq + e) * kBPitch + (wn * NT + j) * 32 + l31];
            }
    };
    auto mfma_group = [&](const float4 (&av)[MT], const float (&bv)[NT][4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float ax = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[j][e], acc[i][j], 0, 0, 0);
            }
    };
    auto compute = [&]() {
        constexpr int kGroups = kBK / 8;
        float4 av[2][MT];
        float bv[2][NT][4];
        read_frags(0, av[0], bv[0]);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            if (g + 1 < kGroups) read_frags(4 * (g + 1), av[(g + 1) & 1], bv[(g + 1) & 1]);
            mfma_group(av[g & 1], bv[g & 1]);
        }
    };

    const int64_t n_ktiles = (k + kBK - 1) / kBK;
    const bool a_inside = (m0 + kBM <= m) && a_vec;      // the A tile needs no guards (full k tiles)
    const bool b_inside = (n0 + BN <= n) && b_vec;
    const bool interior = (m0 + kBM <= m) && (n0 + BN <= n);
    const int64_t n_full = a_inside ? k / kBK : 0;       // leading k tiles whose A loads need no guards
    if (n_ktiles > 0) {
        if (n_full > 0)
            load_a_fast(0);
        else
            load_a_guarded(0);
        if (n_full > 0 && b_inside)
            load_b_fast(0);
        else
            load_b_guarded(0);
        store_tile();
    }
    __syncthreads();
    int64_t t = 0;
    if (b_inside) {
        for (; t + 1 < n_full; ++t) {  // the next tile needs no guards: a branch-free body
            load_a_fast((t + 1) * kBK);
            load_b_fast((t + 1) * kBK);
            compute();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
    } else {
        for (; t + 1 < n_full; ++t) {  // edge tile in n (a narrow output): only the small B tile is guarded
            load_a_fast((t + 1) * kBK);
            load_b_guarded((t + 1) * kBK);
            compute();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
    }
    for (; t < n_ktiles; ++t) {
        if (t + 1 < n_ktiles) {
            load_a_guarded((t + 1) * kBK);
            load_b_guarded((t + 1) * kBK);
        }
        compute();
        __syncthreads();
        if (t + 1 < n_ktiles) store_tile();
        __syncthreads();
    }
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (gate) {
        // C = gate > 0 ? (A.B) / divisor : 0 -- the backward of dropout(relu(.)) folded into the product
        // that computes the gradient of its output (gate = that output: positive where the unit was
        // active and kept)
        if (interior) {
            // the gate values of a 32x32 tile are all requested before the first store
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int64_t r0 = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
                    const int64_t c0 = n0 + (wn * NT + j) * 32 + l31;
                    const float* gp = gate + r0 * ldg + c0;
                    float gv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        gv[e] = gp[0];
                        gp += ((e & 3) == 3) ? 5 * ldg : ldg;
                    }
                    float* cp = c + r0 * ldc + c0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        cp[0] = gv[e] > 0.0f ? __fdiv_rn(acc[i][j][e], divisor) : 0.0f;
                        cp += ((e & 3) == 3) ? 5 * ldc : ldc;
                    }
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t gc = n0 + (wn * NT + j) * 32 + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t gr = m0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (gr < m && gc < n)
                        c[gr * ldc + gc] = gate[gr * ldg + gc] > 0.0f ? __fdiv_rn(acc[i][j][e], divisor) : 0.0f;
                }
            }
        return;
    }
    if (od.keep_prob < 1.0f) {
        // C = dropout(act(A.B)): the next layer's input dropout in the store (decision id = index_offset +
        // row * pitch + column, as a separate pass over C would draw it)
        const uint64_t key = od.stream_offset ? stream_key(od.seed, od.stream_id + *od.stream_offset) : od.key;
        if (interior && !od.row_ids) {     // running pointer and running decision id, no bounds tests
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int64_t r0 = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);
                    const int64_t c0 = n0 + (wn * NT + j) * 32 + l31;
                    float* cp = c + r0 * ldc + c0;
                    uint64_t id = (uint64_t)(od.index_offset + r0 * od.pitch + c0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[i][j][e];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        cp[0] = keep_field(decision_hash(key, id), id, od.threshold) ? __fdiv_rn(v, od.keep_prob) : 0.0f;
                        const int step = ((e & 3) == 3) ? 5 : 1;
                        cp += step * ldc;
                        id += (uint64_t)(step * od.pitch);
                    }
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t gc = n0 + (wn * NT + j) * 32 + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t gr = m0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (gr < m && gc < n) {
                        float v = acc[i][j][e];
                        if (relu) v = v > 0.0f ? v : 0.0f;
                        const int64_t drawn_row = od.row_ids ? od.row_ids[gr] : gr;
                        const uint64_t id = (uint64_t)(od.index_offset + drawn_row * od.pitch + gc);
                        c[gr * ldc + gc] =
                            keep_field(decision_hash(key, id), id, od.threshold) ? __fdiv_rn(v, od.keep_prob) : 0.0f;
                    }
                }
            }
        return;
    }
    if (interior) {
        // no bounds tests and one running pointer per 32x32 tile: the guarded form below spends ~15
        // VALU instructions per stored element, as much issue time as a tenth of a K = 512 tile's MFMAs
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float* cp = c + (m0 + (wm * MT + i) * 32 + 4 * (lane >> 5)) * ldc + n0 + (wn * NT + j) * 32 + l31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    cp[0] = v;
                    cp += ((e & 3) == 3) ? 5 * ldc : ldc;  // rows 0-3, 8-11, 16-19, 24-27 of this half-wave
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t gc = n0 + (wn * NT + j) * 32 + l31;
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

// ------------------------------------------------------------------------------------------------
// The same product, same tile and same order of additions, for the part of a problem that needs no guard
// anywhere: whole 128-row panels, n a multiple of 128, k a multiple of 32, 16-byte aligned rows.
// What it removes is vector arithmetic outside the MFMAs.  With three blocks on a CU the MFMA pipe of a
// SIMD is saturated by the other blocks' waves, and every ordinary vector instruction of a wave waits for
// a gap between their MFMAs -- about one MFMA time (64 cycles) each.  Measured with per-block
// timestamps on 2^22 x 512 x 512: of the 97 us a tile holds its slot, 12.7 us went to the ~250 vector
// instructions before the first stage was in LDS and 8.5 us to the 64 stores of a lane at 5 vector
// instructions each, against 73 us in the k loop.  So here
//  - the tile comes from the block index by shifts (grid x = XCD + 8 * column tile, grid y = group of 8
//    row panels: the blocks of an XCD still walk the column tiles of one panel one after the other),
//  - every global access is a buffer instruction: a per-lane byte offset computed once per tile, and a
//    wave-uniform offset (k stage, row of the output tile) kept and advanced in scalar registers,
//  - ReLU is one v_max_f32 against the constant 0 per element.  It writes the bits of the guarded kernel's
//    `v > 0 ? v : 0` for everything an accumulator can hold: the hardware maximum returns +0 for -0 and
//    the other operand (+0) for a quiet NaN -- the only kind arithmetic produces --, and an infinity
//    passes as itself (tested on rows that hold NaN, both infinities and negative underflow).
// ------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
enum { kEpiPlain = 0, kEpiGate = 1, kEpiDrop = 2 };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_range_buffer(const void* base) {
    // raw buffer (stride 0), no range clamp below 4 GiB, 32-bit data format
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
}

// A_MASK: input dropout on A from mask bits (InMask), applied between the global load of a stage and its LDS
// store: one more dword load per row and stage (the mask word of the stage's 32 k) and ~3 vector instructions per
// element.  EPI == kEpiDrop: the store applies the NEXT layer's input dropout to the (optionally rectified)
// output, one hash per element (OutDrop; row_ids optional).
// the value lane `src` (0..3) of every quad holds, in all four lanes of the quad (a DPP quad_perm move)
__device__ __forceinline__ uint32_t quad_read(uint32_t x, int src) {
    switch (src) {
        case 0: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x00, 0xf, 0xf, true);
        case 1: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x55, 0xf, 0xf, true);
        case 2: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xaa, 0xf, 0xf, true);
        default: return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xff, 0xf, 0xf, true);
    }
}

template <bool TRANS_B, int EPI, bool A_MASK = false>
__global__ __launch_bounds__(kBlock, 3) void gemm_panels_kernel(int tiles_m, int n_ktiles,
                                                               const float* __restrict__ a, int64_t lda,
                                                               const float* __restrict__ b, int64_t ldb,
                                                               float* __restrict__ c, int64_t ldc, int relu,
                                                               const float* __restrict__ gate, int64_t ldg,
                                                               float divisor, InMask im, OutDrop od, int raster) {
    constexpr int kBM = 128, BN = 128, kBK = 32, MT = 2, NT = 2;
    constexpr int kAPitch = kBK + 4, kBPitch = BN + 4;
    __shared__ __attribute__((aligned(16))) float as[kBM * kAPitch];
    __shared__ __attribute__((aligned(16))) float bs[TRANS_B ? BN * kAPitch : kBK * kBPitch];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // The blocks of an XCD (block ids equal modulo 8) are dispatched in the order of (y, x / 8).  Plain order:
    // they walk the column tiles of one row panel, then the next panel's.  With many column tiles (`raster`)
    // that makes every panel stream the whole of B past the XCD's L2, so the order becomes: down a group of 8
    // row panels for one column tile, then the next column tile -- the ~128 blocks an XCD holds at a time
    // then share 8 panels of A and 16 column tiles of B.  All of it scalar arithmetic.
    int py = (int)blockIdx.y, cx = (int)(blockIdx.x >> 3);
    if (raster) {
        const int nct = (int)(gridDim.x >> 3);
        const int g0 = py & ~7;
        const int gh = min(8, (int)gridDim.y - g0);
        const int r = (py - g0) * nct + cx;
        if (gh == 8) {
            cx = r >> 3;
            py = g0 + (r & 7);
        } else {
            cx = r / gh;
            py = g0 + r % gh;
        }
    }
    const int panel = py * 8 + (int)(blockIdx.x & 7);
    if (panel >= tiles_m) return;
    const int64_t m0 = (int64_t)panel * kBM;
    const int64_t n0 = (int64_t)cx * BN;

    const __amdgpu_buffer_rsrc_t res_a = whole_range_buffer(a + m0 * lda);
    const __amdgpu_buffer_rsrc_t res_b = whole_range_buffer(TRANS_B ? b + n0 * ldb : b + n0);
    // A: rows tid / 8 + 32 i, k quad tid % 8.  B [k x n]: k row tid / 32 + 8 p, column quad tid % 32.
    // B [n x k]: column tid / 8 + 32 p, k quad tid % 8.
    const int a_voff = ((tid >> 3) * (int)lda + (tid & 7) * 4) * 4;
    const int b_voff = TRANS_B ? ((tid >> 3) * (int)ldb + (tid & 7) * 4) * 4 : ((tid >> 5) * (int)ldb + (tid & 31) * 4) * 4;
    const int a_step = 32 * (int)lda * 4;                                   // unit i -> i + 1
    const int b_step = TRANS_B ? 32 * (int)ldb * 4 : 8 * (int)ldb * 4;       // unit p -> p + 1
    const int b_kstep = TRANS_B ? kBK * 4 : kBK * (int)ldb * 4;              // one k stage
    float4 ra[4], rb[4];
    // mask words of this thread's 4 rows for the stage being loaded (A_MASK): word = stage number
    uint32_t rm[4];
    const __amdgpu_buffer_rsrc_t res_m = whole_range_buffer(A_MASK ? im.bits + m0 * im.ld_words : nullptr);
    const int m_voff = A_MASK ? (tid >> 3) * (int)im.ld_words * 4 : 0;
    const int m_step = A_MASK ? 32 * (int)im.ld_words * 4 : 0;
    const int m_shift = (tid & 7) * 4;
    auto load_stage = [&](int a_k, int b_k) {      // scalar byte offsets of the stage in A and in B
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4v v = __builtin_bit_cast(
                f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_a, a_voff, a_k + i * a_step, 0));
            ra[i] = make_float4(v.x, v.y, v.z, v.w);
            if (A_MASK) rm[i] = __builtin_amdgcn_raw_buffer_load_b32(res_m, m_voff, (a_k >> 5) + i * m_step, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x4v v = __builtin_bit_cast(
                f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_b, b_voff, b_k + p * b_step, 0));
            rb[p] = make_float4(v.x, v.y, v.z, v.w);
        }
    };
    float* const as_w = as + (tid >> 3) * kAPitch + (tid & 7) * 4;
    float* const bs_w = TRANS_B ? bs + (tid >> 3) * kAPitch + (tid & 7) * 4 : bs + (tid >> 5) * kBPitch + (tid & 31) * 4;
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (A_MASK) ra[i] = masked_scaled4(ra[i], rm[i], m_shift, im);
            *reinterpret_cast<float4*>(as_w + 32 * i * kAPitch) = ra[i];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(bs_w + (TRANS_B ? 32 * p * kAPitch : 8 * p * kBPitch)) = rb[p];
    };

    f32x16 acc[MT][NT];
    const int kh = (lane >> 5) * (kBK / 2);
    const int l31 = lane & 31;
    const float* const as_r = as + (wm * MT * 32 + l31) * kAPitch + kh;
    const float* const bs_r = TRANS_B ? bs + (wn * NT * 32 + l31) * kAPitch + kh : bs + kh * kBPitch + wn * NT * 32 + l31;
    auto read_frags = [&](int kq, float4 (&av)[MT], float (&bv)[NT][4]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = *reinterpret_cast<const float4*>(as_r + i * 32 * kAPitch + kq);
#pragma unroll
        for (int j = 0; j < NT; ++j)
            if (TRANS_B) {
                const float4 t = *reinterpret_cast<const float4*>(bs_r + j * 32 * kAPitch + kq);
                bv[j][0] = t.x;
                bv[j][1] = t.y;
                bv[j][2] = t.z;
                bv[j][3] = t.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][e] = bs_r[(kq + e) * kBPitch + j * 32];
            }
    };
    // `first`: the accumulators start as the zero C operand of their first MFMA (no 64 moves per lane)
    auto mfma_group = [&](const float4 (&av)[MT], const float (&bv)[NT][4], auto first) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const float ax = e == 0 ? av[i].x : e == 1 ? av[i].y : e == 2 ? av[i].z : av[i].w;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (decltype(first)::value && e == 0) {
                        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                                             0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[j][e], zero, 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[j][e], acc[i][j], 0, 0, 0);
                    }
                }
            }
    };
    auto compute = [&](auto first) {
        constexpr int kGroups = kBK / 8;
        float4 av[2][MT];
        float bv[2][NT][4];
        read_frags(0, av[0], bv[0]);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            if (g + 1 < kGroups) read_frags(4 * (g + 1), av[(g + 1) & 1], bv[(g + 1) & 1]);
            if (g == 0)
                mfma_group(av[g & 1], bv[g & 1], first);
            else
                mfma_group(av[g & 1], bv[g & 1], std::false_type{});
        }
    };

    load_stage(0, 0);
    store_tile();
    __syncthreads();
    if (n_ktiles == 1) {
        compute(std::true_type{});
    } else {
        int a_k = kBK * 4, b_k = b_kstep;
        load_stage(a_k, b_k);
        compute(std::true_type{});
        __syncthreads();
        store_tile();
        __syncthreads();
        for (int t = 2; t < n_ktiles; ++t) {
            a_k += kBK * 4;
            b_k += b_kstep;
            load_stage(a_k, b_k);
            compute(std::false_type{});
            __syncthreads();
            store_tile();
            __syncthreads();
        }
        compute(std::false_type{});
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const __amdgpu_buffer_rsrc_t res_c = whole_range_buffer(c + m0 * ldc + n0);
    const int c_voff = (4 * (lane >> 5) * (int)ldc + l31) * 4;
    if (EPI == kEpiGate) {
        // C = gate > 0 ? (A.B) / divisor : 0 (the backward of dropout(relu(.)) folded into the product).
        // For a power-of-two divisor (keep_prob = 0.5) the quotient is the product with the inverse, bit for
        // bit: one multiplication instead of the dozen instructions of a correctly rounded division.
        const __amdgpu_buffer_rsrc_t res_g = whole_range_buffer(gate + m0 * ldg + n0);
        const int g_voff = (4 * (lane >> 5) * (int)ldg + l31) * 4;
        const bool scale_exact = (__float_as_uint(divisor) & 0x007fffffu) == 0;
        const float inv_divisor = 1.0f / divisor;
        auto gated_store = [&](auto exact) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float gv[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                        const unsigned int bits = __builtin_amdgcn_raw_buffer_load_b32(
                            res_g, g_voff, (row * (int)ldg + (wn * NT + j) * 32) * 4, 0);
                        gv[e] = __uint_as_float(bits);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                        const float q = decltype(exact)::value ? acc[i][j][e] * inv_divisor
                                                               : __fdiv_rn(acc[i][j][e], divisor);
                        const float v = gv[e] > 0.0f ? q : 0.0f;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), res_c, c_voff,
                                                              (row * (int)ldc + (wn * NT + j) * 32) * 4, 0);
                    }
                }
        };
        if (scale_exact)
            gated_store(std::true_type{});
        else
            gated_store(std::false_type{});
        return;
    }
    if (EPI == kEpiDrop) {
        // C = dropout(act(A.B)): decision id = index_offset + drawn row * pitch + column.  The four lanes of a
        // quad hold four consecutive columns of the same rows, i.e. (when pitch and index_offset are multiples
        // of 4) the four ids of one hash: lane s of the quad hashes for the rows 8 t + s of a 32 x 32 tile and the
        // quad passes the hashes round with DPP moves -- a quarter of the 64-bit multiplications.
        const uint64_t key = od.stream_offset ? stream_key(od.seed, od.stream_id + *od.stream_offset) : od.key;
        const bool scale_exact = (__float_as_uint(od.keep_prob) & 0x007fffffu) == 0;
        const float inv_keep = 1.0f / od.keep_prob;
        const bool quads = ((od.pitch | od.index_offset) & 3) == 0;
        const int s4 = lane & 3;
        const int field_shift = 16 * (1 - (s4 & 1));
        auto finish = [&](float v, bool kept) {
            if (relu) asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(v));
            const float q = scale_exact ? v * inv_keep : __fdiv_rn(v, od.keep_prob);
            return kept ? q : 0.0f;
        };
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t row0 = m0 + (wm * MT + i) * 32 + 4 * (lane >> 5);    // row of element e: row0 + (e & 3) + 8 (e >> 2)
                const int64_t col = n0 + (wn * NT + j) * 32 + l31;
                if (quads) {
                    uint32_t h_lo[4], h_hi[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int64_t gr = row0 + s4 + 8 * t;
                        const int64_t drawn = od.row_ids ? od.row_ids[gr] : gr;
                        const uint64_t h = decision_hash(key, (uint64_t)(od.index_offset + drawn * od.pitch + (col - s4)));
                        h_lo[t] = (uint32_t)h;
                        h_hi[t] = (uint32_t)(h >> 32);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                        const uint32_t lo = quad_read(h_lo[e >> 2], e & 3);
                        const uint32_t hi = quad_read(h_hi[e >> 2], e & 3);
                        const uint32_t field = ((s4 < 2 ? hi : lo) >> field_shift) & 0xffffu;   // field s4 of the hash
                        const float v = finish(acc[i][j][e], field >= od.threshold);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), res_c, c_voff,
                                                              (row * (int)ldc + (wn * NT + j) * 32) * 4, 0);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                        const int64_t gr = row0 + (e & 3) + 8 * (e >> 2);
                        const int64_t drawn = od.row_ids ? od.row_ids[gr] : gr;
                        const uint64_t id = (uint64_t)(od.index_offset + drawn * od.pitch + col);
                        const float v = finish(acc[i][j][e], keep_field(decision_hash(key, id), id, od.threshold));
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), res_c, c_voff,
                                                              (row * (int)ldc + (wn * NT + j) * 32) * 4, 0);
                    }
                }
            }
        return;
    }
    if (relu) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                    float v;   // one instruction: the compiler's fmaxf is a canonicalising max followed by the max
                    asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(acc[i][j][e]));
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), res_c, c_voff,
                                                          (row * (int)ldc + (wn * NT + j) * 32) * 4, 0);
                }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2);
                const float v = acc[i][j][e];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), res_c, c_voff,
                                                      (row * (int)ldc + (wn * NT + j) * 32) * 4, 0);
            }
}

// true when gemm_panels_kernel can take the leading whole panels of this problem
static bool panels_form_fits(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                             int trans_b, int64_t ldc, const float* gate, int64_t ldg) {
    const int64_t limit = (int64_t)1 << 29;          // every byte offset inside a tile's buffers stays below 2^31
    if (m < 128 || n % 128 != 0 || k % 32 != 0 || k < 32) return false;
    if (lda % 4 || ldb % 4 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15)) return false;
    if (128 * lda + k >= limit || 128 * ldc + n >= limit || (gate && 128 * ldg + n >= limit)) return false;
    if (trans_b ? 128 * ldb + k >= limit : (k + 32) * ldb + 128 >= limit) return false;
    return m / 128 <= 8 * 65535;
}

static const InMask kNoInMask = {nullptr, 0, 1.0f, 1.0f, 1};

template <int kBM, int BN>
static int launch_gemm(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       int trans_b, float* c, int64_t ldc, int relu, hipStream_t st, const float* gate = nullptr,
                       int64_t ldg = 0, float divisor = 1.0f, const OutDrop* out_drop = nullptr,
                       const InMask* in_mask = nullptr) {
    const InMask im = in_mask ? *in_mask : kNoInMask;
    const OutDrop no_drop = {0, 1.0f, 0, 0, nullptr, 0, 0, nullptr, 0};
    // the byte offset of a mask word inside a panel's buffer stays below 2^31 like every other offset
    const bool mask_fits = !im.bits || (!trans_b && !gate && !out_drop && 128 * im.ld_words + k / 32 < ((int64_t)1 << 29));
    if (kBM == 128 && BN == 128 && mask_fits && panels_form_fits(m, n, k, a, lda, b, ldb, trans_b, ldc, gate, ldg)) {
        // whole 128-row panels through the kernel without guards, the rows left over through the general one
        const int64_t panels = m / 128;
        const dim3 grid((unsigned)(8 * (n / 128)), (unsigned)ceil_div(panels, 8));
        // 16 column tiles or more: groups of 8 panels per column tile (8192^3: 148.1 -> 149.7 TF/s, 16384 x 4096 x
        // 4096: 148.1 -> 150.6, 2^20 x 512 x 2048: 146.6 -> 147.8; no difference at 2 or 4 column tiles)
        const int raster = n / 128 >= 16 ? 1 : 0;
#define GCN_PANELS_GO(TB, EPI, AM)                                                                                \
    gemm_panels_kernel<TB, EPI, AM><<<grid, kBlock, 0, st>>>((int)panels, (int)(k / 32), a, lda, b, ldb, c, ldc, relu, gate, \
                                                             ldg, divisor, im, out_drop ? out_drop_ready(*out_drop) : no_drop, raster)
        if (im.bits) {
            GCN_PANELS_GO(false, kEpiPlain, true);
        } else if (out_drop) {
            if (trans_b)
                GCN_PANELS_GO(true, kEpiDrop, false);
            else
                GCN_PANELS_GO(false, kEpiDrop, false);
        } else if (gate) {
            if (trans_b)
                GCN_PANELS_GO(true, kEpiGate, false);
            else
                GCN_PANELS_GO(false, kEpiGate, false);
        } else {
            if (trans_b)
                GCN_PANELS_GO(true, kEpiPlain, false);
            else
                GCN_PANELS_GO(false, kEpiPlain, false);
        }
#undef GCN_PANELS_GO
        GCN_LAUNCH_CHECK();
        const int64_t done = panels * 128;
        if (done == m) return GCN_OK;
        // the rows left over: their decisions and mask words are those of rows `done` onwards
        OutDrop rest_drop = out_drop ? *out_drop : no_drop;
        if (out_drop) {
            if (rest_drop.row_ids)
                rest_drop.row_ids += done;
            else
                rest_drop.index_offset += done * rest_drop.pitch;
        }
        InMask rest_mask = im;
        if (im.bits) rest_mask.bits += done * im.ld_words;
        return launch_gemm<kBM, BN>(m - done, n, k, a + done * lda, lda, b, ldb, trans_b, c + done * ldc, ldc, relu, st,
                                    gate ? gate + done * ldg : nullptr, ldg, divisor, out_drop ? &rest_drop : nullptr,
                                    im.bits ? &rest_mask : nullptr);
    }
    const OutDrop od = out_drop ? out_drop_ready(*out_drop) : OutDrop{0, 1.0f, 0, 0, nullptr, 0, 0};
    const int64_t tiles_m = ceil_div(m, kBM), tiles_n = ceil_div(n, BN);
    const int64_t n_tiles = tiles_m * tiles_n;
    if (n_tiles > INT32_MAX) return fail(GCN_ERR_BAD_ARG, "matrix too large for one launch");
    const int a_vec = (lda % 4 == 0) && (((uintptr_t)a & 15) == 0);
    const int b_vec = (ldb % 4 == 0) && (((uintptr_t)b & 15) == 0);
#define GCN_GEMM_GO(TB, BK)                                                                                  \
    gemm_mfma_kernel<kBM, BN, TB, BK><<<(unsigned)n_tiles, kBlock, 0, st>>>(m, n, k, a, lda, b, ldb, c, ldc, relu, a_vec, \
                                                                       b_vec, tiles_m, tiles_n, gate, ldg, divisor, od, im)
    if (k <= 16) {
        if (trans_b)
            GCN_GEMM_GO(true, 16);
        else
            GCN_GEMM_GO(false, 16);
    } else {
        if (trans_b)
            GCN_GEMM_GO(true, 32);
        else
            GCN_GEMM_GO(false, 32);
    }
#undef GCN_GEMM_GO
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// C[m x n] = act(A[m x k] . B[k x n]) for n <= 32 (hidden -> classes, gcn/layers.py:32-33 on the last layer):
// A is read once, C is n / k of it, so the product is bound by HBM -- as long as nothing else is in the way.
// The tiled kernel above pads n = 16 to a 32-column MFMA tile and its loads, LDS stores, barriers and MFMAs take
// turns.  Here
//  - the MFMA is v_mfma_f32_16x16x4_f32: a 16-column tile, no padding at n = 16;
//  - A goes from global memory straight into the MFMA operand registers: lane (i = lane % 16, g = lane / 16)
//    loads the four consecutive k  k0 + 4 g .. k0 + 4 g + 3  of its row as one 128-bit load per 16-k step and
//    feeds element e to the e-th MFMA of the step, whose four lane groups thus cover k0 + e, k0 + 4 + e,
//    k0 + 8 + e, k0 + 12 + e; B (at most 64 KB) sits in LDS in that order.  No LDS traffic for A, no barrier in
//    the loop, the next step's loads in flight during the MFMAs of this one;
//  - a wave owns 16 rows at a time and requests 1 KiB of each of them (sixteen 16-k steps) before it uses the
//    first; the tiles are walked with a grid stride.
// 2^24 x 256 x 16: 3.65 ms against 3.72-3.85 ms of the tiled kernel (5.0 TB/s; see the note at the stores).
// Order of additions of an output element: steps in ascending k, inside a step e = 0..3, inside an MFMA the
// hardware's fixed order over its four k.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(kBlock) void gemm_skinny_kernel(int64_t m, int n, int k, const float* __restrict__ a,
                                                            int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                            float* __restrict__ c, int64_t ldc, int relu,
                                                            int64_t n_tiles, int c_vec) {
    constexpr int kCols = 16 * NT;
    constexpr int kTrPitch = 20;           // of a wave's 16 x 16 output tile in LDS: conflict-free both ways
    constexpr int kPitch = kCols + 4;      // rows 4 apart (the four lane groups of a fragment read) 16 banks apart
    constexpr int kSteps = 16;             // 16-k steps requested together: 1 KiB of every row
    extern __shared__ __attribute__((aligned(16))) float skinny_b[];
    for (int idx = threadIdx.x; idx < k * kCols; idx += kBlock) {
        const int kk = idx / kCols, col = idx % kCols;
        skinny_b[kk * kPitch + col] = col < n ? b[(int64_t)kk * ldb + col] : 0.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    const int steps = k / 16;
    for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
        const int64_t row0 = tile * 16;
        const int64_t row = min(row0 + i, m - 1);                    // rows past the end repeat the last one
        const float* src = a + row * lda + 4 * g;
        f32x4v acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        for (int s0 = 0; s0 < steps; s0 += kSteps) {
            // every line of the rows' next 1 KiB is requested before the first is used: the 16 rows of a tile
            // are 16 runs of consecutive lines for the memory system, not 16 x 8 separate visits
            f32x4v av[kSteps];
#pragma unroll
            for (int u = 0; u < kSteps; ++u)
                if (s0 + u < steps) av[u] = *reinterpret_cast<const f32x4v*>(src + (s0 + u) * 16);
#pragma unroll
            for (int u = 0; u < kSteps; ++u) {
                if (s0 + u < steps) {
                    const float* frag = skinny_b + ((s0 + u) * 16 + 4 * g) * kPitch + i;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][e], frag[e * kPitch + 16 * t], acc[t], 0, 0, 0);
                }
            }
        }
        // Lane (j = lane % 16, g) holds rows 4 g .. 4 g + 3 of column j of every 16 x 16 tile: stored as it stands,
        // a wave instruction writes four 64-byte pieces.  The tile goes through 1.25 KiB of LDS private to the
        // wave instead, and lane l writes the four consecutive columns 4 (l % 4) .. of row l / 4: one 128-bit store
        // per lane, 1 KiB contiguous per instruction when C's rows are 16 floats.  (What the output costs is not
        // its shape: on 2^24 x 256 x 16 the kernel takes 2.85 ms with the stores removed or confined to the L2 --
        // 6.1 TB/s, the rate of a pure read sweep -- and 3.65 ms with the 1 GB of C going to HBM, in 64-byte
        // pieces, in whole lines or non-temporal alike; a bare sweep that reads 16 GiB and writes 1 GiB beside it
        // takes 3.55-3.69 ms too: that is the rate of this traffic mix.)
        float* tr = skinny_b + k * kPitch + (threadIdx.x >> 6) * (16 * kTrPitch);
        const int orow_l = lane >> 2, ocol_l = 4 * (lane & 3);
        const int64_t orow = row0 + orow_l;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][r];
                if (relu) v = v > 0.0f ? v : 0.0f;
                tr[(4 * g + r) * kTrPitch + i] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4v v4 = *reinterpret_cast<const f32x4v*>(tr + orow_l * kTrPitch + ocol_l);
            const int col = 16 * t + ocol_l;
            if (orow < m) {
                float* dst = c + orow * ldc + col;
                if (c_vec && col + 3 < n) {
                    *reinterpret_cast<f32x4v*>(dst) = v4;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < n) dst[e] = v4[e];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// true when gemm_skinny_kernel takes the product: few output columns, k in whole 16-k steps, B small enough for
// LDS, 128-bit rows of A, and rows enough for the streaming form to matter
static bool skinny_form_fits(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, int trans_b) {
    const int64_t lds = (k * (n <= 16 ? 20 : 36) + (kBlock / 64) * 16 * 20) * (int64_t)sizeof(float);
    return !trans_b && n <= 32 && k >= 16 && k % 16 == 0 && lds <= 64 * 1024 && m >= 4096 && lda % 4 == 0 &&
           ((uintptr_t)a & 15) == 0;
}

static int launch_skinny(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                         float* c, int64_t ldc, int relu, hipStream_t st) {
    const int64_t n_groups = ceil_div(m, 16);          // 16-row tiles, one wave each at a time
    const int nt = n <= 16 ? 1 : 2;
    const size_t lds = ((size_t)k * (size_t)(16 * nt + 4) + (kBlock / 64) * 16 * 20) * sizeof(float);
    const int c_vec = (ldc % 4 == 0 && ((uintptr_t)c & 15) == 0) ? 1 : 0;
    // every block stages B once: few blocks per CU's worth of waves, each walking many row groups
    const int64_t blocks = std::min<int64_t>(ceil_div(n_groups, kBlock / 64), 256 * 8);
    if (nt == 1)
        gemm_skinny_kernel<1><<<(unsigned)blocks, kBlock, lds, st>>>(m, (int)n, (int)k, a, lda, b, ldb, c, ldc, relu, n_groups, c_vec);
    else
        gemm_skinny_kernel<2><<<(unsigned)blocks, kBlock, lds, st>>>(m, (int)n, (int)k, a, lda, b, ldb, c, ldc, relu, n_groups, c_vec);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

// C[k x n] = A^T . G for A [m x k], G [m x n]: the weight gradient dW = H^T . dP of a layer
// (the gradient of tf.matmul / tf.sparse_tensor_dense_matmul w.r.t. the weights, gcn/models.py:86).
// The reduction runs over the m nodes, so both operands are read in their storage order: a stage is 32
// rows of A (TM columns) and of G (BN columns), copied to LDS as they are, and lane (column l % 32,
// half l / 32) feeds row 16 * half + s of the stage to MFMA step s.  The rows are cut into `splits`
// ranges, one block per (tile, range); with more than one range the partial tiles go to the workspace
// and reduce_splits_kernel adds them in range order, so the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------------
// BUF (128 x 128 tiles only): the unguarded stages are read with buffer instructions -- per-lane byte
// offsets computed once, the stage offset in a scalar register -- so that the loop holds no vector
// arithmetic and 16 registers fewer (see gemm_panels_kernel).
template <int TM, int BN, bool BUF>
__global__ __launch_bounds__(64 * ((TM / 32) * (BN / 32) > 4 ? 4 : (TM / 32) * (BN / 32)), 3)
void gemm_tn_kernel(int64_t m, int64_t n, int64_t k, const float* __restrict__ a, int64_t lda,
                    const float* __restrict__ g, int64_t ldg, float* __restrict__ out, int64_t ldo,
                    int64_t split_stride, int64_t rows_per_split, int vec_ok, int tiles_n, int tiles, int splits,
                    InMask im) {
    constexpr int kTiles = (TM / 32) * (BN / 32);
    constexpr int kWaves = kTiles > 4 ? 4 : kTiles;
    constexpr int kThreads = 64 * kWaves;
    constexpr int kWavesN = (BN / 32 >= 2 && kWaves >= 2) ? (TM == 32 ? kWaves : 2) : 1;
    constexpr int kWavesM = kWaves / kWavesN;
    constexpr int MT = TM / (32 * kWavesM), NT = BN / (32 * kWavesN);
    constexpr int kRows = 32;
    constexpr int kXPitch = TM + 4, kGPitch = BN + 4;
    constexpr int kXPer = kRows * TM / 4 / kThreads, kGPer = kRows * BN / 4 / kThreads;
    __shared__ __attribute__((aligned(16))) float xs[kRows * kXPitch];
    __shared__ __attribute__((aligned(16))) float gs[kRows * kGPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / kWavesN, wn = wave % kWavesN;
    // Block numbering: the tiles of one row range read the same rows of A and G, so they run on one XCD
    // (block ids equal modulo 8), one after the other in dispatch order, and share its L2: XCD x takes
    // the row ranges x, x + 8, ...
    const unsigned per_xcd = blockIdx.x >> 3;
    const int tile = (int)(per_xcd % (unsigned)tiles);
    const int split = (int)(per_xcd / (unsigned)tiles) * 8 + (int)(blockIdx.x & 7);
    if (split >= splits) return;
    const int64_t k0 = (int64_t)(tile / tiles_n) * TM;  // first column of A = first row of C
    const int64_t n0 = (int64_t)(tile % tiles_n) * BN;
    const int64_t r_begin = (int64_t)split * rows_per_split;
    const int64_t r_end = (r_begin + rows_per_split < m) ? r_begin + rows_per_split : m;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    float4 rx[kXPer], rg[kGPer];
    // input dropout on A (im.bits): the mask of the forward pass, read again -- a float4 of A holds 4
    // consecutive columns of one row, all inside one mask word
    auto mask_x = [&](int p, int64_t gr, int64_t gc) {
        if (im.bits) {
            const uint32_t word = (gr < r_end && gc < k) ? im.bits[gr * im.ld_words + (gc >> 5)] : 0u;
            rx[p] = masked_scaled4(rx[p], word, (int)(gc & 31), im);
        }
    };
    auto load_fast = [&](int64_t r0) {
#pragma unroll
        for (int p = 0; p < kXPer; ++p) {
            const int u = tid + kThreads * p;
            rx[p] = *reinterpret_cast<const float4*>(a + (r0 + u / (TM / 4)) * lda + k0 + (u % (TM / 4)) * 4);
            mask_x(p, r0 + u / (TM / 4), k0 + (u % (TM / 4)) * 4);
        }
#pragma unroll
        for (int p = 0; p < kGPer; ++p) {
            const int u = tid + kThreads * p;
            rg[p] = *reinterpret_cast<const float4*>(g + (r0 + u / (BN / 4)) * ldg + n0 + (u % (BN / 4)) * 4);
        }
    };
    // edge tiles and the last rows: whole float4 units where they are in range and aligned (a narrow
    // operand -- 16 hidden units, 7 classes -- makes every tile an edge tile), single elements elsewhere
    auto guarded_unit = [&](const float* base, int64_t ld, int64_t gr, int64_t gc, int64_t limit) -> float4 {
        if (gr >= r_end || gc >= limit) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float* src = base + gr * ld + gc;
        if (vec_ok && gc + 4 <= limit) return *reinterpret_cast<const float4*>(src);
        float t[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = (gc + e < limit) ? src[e] : 0.0f;
        return make_float4(t[0], t[1], t[2], t[3]);
    };
    auto load_guarded = [&](int64_t r0) {
#pragma unroll
        for (int p = 0; p < kXPer; ++p) {
            const int u = tid + kThreads * p;
            rx[p] = guarded_unit(a, lda, r0 + u / (TM / 4), k0 + (u % (TM / 4)) * 4, k);
            mask_x(p, r0 + u / (TM / 4), k0 + (u % (TM / 4)) * 4);
        }
#pragma unroll
        for (int p = 0; p < kGPer; ++p) {
            const int u = tid + kThreads * p;
            rg[p] = guarded_unit(g, ldg, r0 + u / (BN / 4), n0 + (u % (BN / 4)) * 4, n);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int p = 0; p < kXPer; ++p) {
            const int u = tid + kThreads * p;
            *reinterpret_cast<float4*>(xs + (u / (TM / 4)) * kXPitch + (u % (TM / 4)) * 4) = rx[p];
        }
#pragma unroll
        for (int p = 0; p < kGPer; ++p) {
            const int u = tid + kThreads * p;
            *reinterpret_cast<float4*>(gs + (u / (BN / 4)) * kGPitch + (u % (BN / 4)) * 4) = rg[p];
        }
    };
    const int half = (lane >> 5) * (kRows / 2);
    const int l31 = lane & 31;
    auto compute = [&]() {
#pragma unroll
        for (int s = 0; s < kRows / 2; ++s) {
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = xs[(half + s) * kXPitch + (wm * MT + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = gs[(half + s) * kGPitch + (wn * NT + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    };

    const bool cols_inside = (k0 + TM <= k) && (n0 + BN <= n) && vec_ok;
    const int64_t n_stages = (r_end > r_begin) ? (r_end - r_begin + kRows - 1) / kRows : 0;
    const int64_t n_full = cols_inside ? (r_end - r_begin) / kRows : 0;
    if (n_stages > 0) {
        if (n_full > 0)
            load_fast(r_begin);
        else
            load_guarded(r_begin);
        store_tile();
    }
    __syncthreads();
    int64_t t = 0;
    if (BUF) {
        const __amdgpu_buffer_rsrc_t res_x = whole_range_buffer(a + r_begin * lda + k0);
        const __amdgpu_buffer_rsrc_t res_g = whole_range_buffer(g + r_begin * ldg + n0);
        // unit p of a thread: row tid / 32 + 8 p of the stage, column quad tid % 32
        const int x_voff = ((tid >> 5) * (int)lda + (tid & 31) * 4) * 4;
        const int g_voff = ((tid >> 5) * (int)ldg + (tid & 31) * 4) * 4;
        const int x_unit = 8 * (int)lda * 4, g_unit = 8 * (int)ldg * 4;
        // mask words (input dropout on A): row tid / 32 + 8 p of the stage, word (k0 + 4 (tid % 32)) / 32
        const bool masked = im.bits != nullptr;
        const __amdgpu_buffer_rsrc_t res_m =
            whole_range_buffer(masked ? im.bits + r_begin * im.ld_words + (k0 >> 5) : nullptr);
        const int m_voff = masked ? ((tid >> 5) * (int)im.ld_words + ((tid & 31) >> 3)) * 4 : 0;
        const int m_unit = masked ? 8 * (int)im.ld_words * 4 : 0;
        const int m_shift = (tid & 7) * 4;
        int x_stage = 0, g_stage = 0, m_stage = 0;
        for (; t + 1 < n_full; ++t) {
            x_stage += kRows * (int)lda * 4;
            g_stage += kRows * (int)ldg * 4;
            m_stage += masked ? kRows * (int)im.ld_words * 4 : 0;
            uint32_t rm[kXPer];
#pragma unroll
            for (int p = 0; p < kXPer; ++p) {
                const f32x4v v = __builtin_bit_cast(
                    f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_x, x_voff, x_stage + p * x_unit, 0));
                rx[p] = make_float4(v.x, v.y, v.z, v.w);
                if (masked) rm[p] = __builtin_amdgcn_raw_buffer_load_b32(res_m, m_voff, m_stage + p * m_unit, 0);
            }
#pragma unroll
            for (int p = 0; p < kGPer; ++p) {
                const f32x4v v = __builtin_bit_cast(
                    f32x4v, __builtin_amdgcn_raw_buffer_load_b128(res_g, g_voff, g_stage + p * g_unit, 0));
                rg[p] = make_float4(v.x, v.y, v.z, v.w);
            }
            compute();
            __syncthreads();
            if (masked) {
#pragma unroll
                for (int p = 0; p < kXPer; ++p) rx[p] = masked_scaled4(rx[p], rm[p], m_shift, im);
            }
            store_tile();
            __syncthreads();
        }
    } else {
        for (; t + 1 < n_full; ++t) {
            load_fast(r_begin + (t + 1) * kRows);
            compute();
            __syncthreads();
            store_tile();
            __syncthreads();
        }
    }
    for (; t < n_stages; ++t) {
        if (t + 1 < n_stages) load_guarded(r_begin + (t + 1) * kRows);
        compute();
        __syncthreads();
        if (t + 1 < n_stages) store_tile();
        __syncthreads();
    }
    float* dst = out + (int64_t)split * split_stride;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t gc = n0 + (wn * NT + j) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t gr = k0 + (wm * MT + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (gr < k && gc < n) dst[gr * ldo + gc] = acc[i][j][e];
            }
        }
}

// C[k x n] = sum over the row ranges, in range order, of the partial products [splits][k x n]
__global__ void reduce_splits_kernel(int64_t k, int64_t n, int64_t splits, const float* __restrict__ part,
                                     float* __restrict__ c, int64_t ldc) {
    const int64_t total = k * n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.0f;
        int64_t p = 0;
        for (; p + 8 <= splits; p += 8) {  // 8 independent loads in flight, added in range order
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = part[(p + u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = __fadd_rn(s, x[u]);
        }
        for (; p < splits; ++p) s = __fadd_rn(s, part[p * total + i]);
        c[(i / n) * ldc + (i % n)] = s;
    }
}

struct TnShape {
    int64_t tiles, tiles_n, splits, rows_per_split;
    int tm, bn;
};

static TnShape tn_shape(int64_t m, int64_t n, int64_t k) {
    TnShape s;
    s.tm = k <= 32 ? 32 : 128;
    s.bn = n <= 32 ? 32 : 128;
    s.tiles_n = ceil_div(n, s.bn);
    s.tiles = ceil_div(k, s.tm) * s.tiles_n;
    // enough blocks to fill the chip several times over, at least 64 rows each (on a small graph the
    // stages of a block are a latency chain: better many short blocks), at most 256 MiB of partials
    int64_t splits = ceil_div(4096, std::max<int64_t>(s.tiles, 1));
    splits = std::min(splits, std::max<int64_t>(1, m / 64));
    splits = std::min(splits, std::max<int64_t>(1, (256ll << 20) / std::max<int64_t>(k * n * 4, 1)));
    splits = std::min<int64_t>(splits, 65535);
    s.rows_per_split = align_up(ceil_div(std::max<int64_t>(m, 1), splits), 32);
    s.splits = ceil_div(std::max<int64_t>(m, 1), s.rows_per_split);
    return s;
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_dropout_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, const float* gate, int64_t ldg, float* y,
                    int64_t ldy, float keep_prob, uint64_t seed, uint64_t stream_id, const uint64_t* stream_offset,
                    uint64_t index_offset, void* stream) {
    if (n_rows < 0 || f < 0 || ldx < f || ldy < f || (gate && ldg < f)) return fail(GCN_ERR_BAD_ARG, "bad shape");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    const int64_t total = n_rows * f;
    if (total == 0) return GCN_OK;
    if (!x || !y) return fail(GCN_ERR_BAD_ARG, "null pointer");
    const bool contiguous = ldx == f && ldy == f && (!gate || ldg == f) && total % 4 == 0 &&
                            (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gate) & 15) == 0;
    if (contiguous) {
        const int64_t n_vec = total / 4;
        int64_t vblocks = ceil_div(n_vec, (int64_t)kBlock * 4);   // tiles of 4 vectors per thread
        if (vblocks > (1 << 22)) vblocks = 1 << 22;
        if (keep_prob < 1.0f)
            dropout_vec_kernel<true><<<(unsigned)vblocks, kBlock, 0, (hipStream_t)stream>>>(
                n_vec, (const float4*)x, (const float4*)gate, (float4*)y, keep_prob, keep_threshold(keep_prob),
                stream_key(seed, stream_id), index_offset, stream_offset, seed, stream_id);
        else
            dropout_vec_kernel<false><<<(unsigned)vblocks, kBlock, 0, (hipStream_t)stream>>>(
                n_vec, (const float4*)x, (const float4*)gate, (float4*)y, keep_prob, 0, 0, index_offset, nullptr, 0, 0);
        GCN_LAUNCH_CHECK();
        return GCN_OK;
    }
    int64_t blocks = ceil_div(total, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dropout_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, f, x, ldx, gate, ldg, y, ldy,
                                                                        keep_prob, keep_threshold(keep_prob),
                                                                        stream_key(seed, stream_id), index_offset,
                                                                        stream_offset, seed, stream_id);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_gather_rows_f32(int64_t n_out, int64_t f, int64_t n_src_rows, const float* src, int64_t ld_src,
                        const int64_t* rows, float* dst, int64_t ld_dst, void* stream) {
    if (n_out < 0 || f < 0 || n_src_rows < 0 || ld_src < f || ld_dst < f) return fail(GCN_ERR_BAD_ARG, "bad shape");
    if (n_out == 0 || f == 0) return GCN_OK;
    if (!src || !rows || !dst) return fail(GCN_ERR_BAD_ARG, "null pointer");
    const bool vec = (f % 4 == 0) && (ld_src % 4 == 0) && (ld_dst % 4 == 0) && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
    const int64_t work = n_out * (vec ? f / 4 : f);
    const int64_t blocks = std::min<int64_t>(ceil_div(work, kBlock), 1 << 16);
    if (vec)
        gather_rows_kernel<4><<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_out, f, src, ld_src, rows, dst, ld_dst);
    else
        gather_rows_kernel<1><<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_out, f, src, ld_src, rows, dst, ld_dst);
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
    if (skinny_form_fits(m, n, k, a, lda, trans_b)) return launch_skinny(m, n, k, a, lda, b, ldb, c, ldc, relu, st);
    if (n <= 32) return launch_gemm<128, 32>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
    if (n <= 64) return launch_gemm<128, 64>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
    return launch_gemm<128, 128>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st);
}

int gcn_gemm_gated_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       int trans_b, const float* gate, int64_t ldg, float divisor, float* c, int64_t ldc,
                       void* stream) {
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    if (lda < k || ldc < n || ldg < n || ldb < (trans_b ? k : n))
        return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (!(divisor > 0.0f)) return fail(GCN_ERR_BAD_ARG, "divisor must be positive");
    if (m == 0 || n == 0) return GCN_OK;
    if (!c || !gate || (k > 0 && (!a || !b))) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (n <= 32) return launch_gemm<128, 32>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, 0, st, gate, ldg, divisor);
    if (n <= 64) return launch_gemm<128, 64>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, 0, st, gate, ldg, divisor);
    return launch_gemm<128, 128>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, 0, st, gate, ldg, divisor);
}

int gcn_gemm_out_dropout_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                             int trans_b, float* c, int64_t ldc, int flags, float keep_prob, uint64_t seed,
                             uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                             const int64_t* row_ids, void* stream) {
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    if (lda < k || ldc < n || ldb < (trans_b ? k : n)) return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (flags & ~GCN_RELU) return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    if (index_offset < 0 || pitch < n)
        return fail(GCN_ERR_BAD_ARG, "bad decision layout: index_offset=%lld pitch=%lld for %lld columns",
                    (long long)index_offset, (long long)pitch, (long long)n);
    if (m == 0 || n == 0) return GCN_OK;
    if (!c || (k > 0 && (!a || !b))) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int relu = (flags & GCN_RELU) ? 1 : 0;
    const OutDrop od = {stream_key(seed, stream_id), keep_prob, index_offset, pitch, stream_offset, seed, stream_id,
                        row_ids};
    if (n <= 32) return launch_gemm<128, 32>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, &od);
    if (n <= 64) return launch_gemm<128, 64>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, &od);
    return launch_gemm<128, 128>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, &od);
}

int gcn_gemm_tn_workspace_bytes(int64_t m, int64_t n, int64_t k, size_t* bytes) {
    if (!bytes) return fail(GCN_ERR_BAD_ARG, "bytes is null");
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    const TnShape s = tn_shape(m, n, k);
    *bytes = s.splits > 1 ? (size_t)s.splits * (size_t)k * (size_t)n * sizeof(float) : 0;
    return GCN_OK;
}

static int gemm_tn_entry(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* g, int64_t ldg,
                         float* c, int64_t ldc, void* workspace, size_t workspace_bytes, void* stream, const InMask& im) {
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    if (lda < k || ldg < n || ldc < n) return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (k == 0 || n == 0) return GCN_OK;
    if (!c || (m > 0 && (!a || !g))) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const TnShape s = tn_shape(m, n, k);
    float* out = c;
    int64_t ldo = ldc, split_stride = 0;
    if (s.splits > 1) {
        const size_t need = (size_t)s.splits * (size_t)k * (size_t)n * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        out = (float*)workspace;
        ldo = n;
        split_stride = k * n;
    }
    const int vec_ok = (lda % 4 == 0) && (ldg % 4 == 0) && (((uintptr_t)a & 15) == 0) && (((uintptr_t)g & 15) == 0);
    const int64_t n_blocks = s.tiles * align_up(s.splits, 8);
    if (n_blocks > INT32_MAX) return fail(GCN_ERR_BAD_ARG, "matrix too large for one launch");
    const dim3 grid((unsigned)n_blocks);
#define GCN_TN_GO(TM, BN, BUF)                                                                                     \
    gemm_tn_kernel<TM, BN, BUF><<<grid, 64 * ((TM / 32) * (BN / 32) > 4 ? 4 : (TM / 32) * (BN / 32)), 0, st>>>(      \
        m, n, k, a, lda, g, ldg, out, ldo, split_stride, s.rows_per_split, vec_ok, (int)s.tiles_n, (int)s.tiles,    \
        (int)s.splits, im)
    // byte offsets inside one row range of a block stay below 2^31
    const bool buf_ok = vec_ok && (s.rows_per_split + 32) * std::max(std::max(lda, ldg), im.bits ? im.ld_words : 0) <
                                      ((int64_t)1 << 29);
    if (s.tm == 128 && s.bn == 128 && buf_ok)
        GCN_TN_GO(128, 128, true);
    else if (s.tm == 128 && s.bn == 128)
        GCN_TN_GO(128, 128, false);
    else if (s.tm == 128)
        GCN_TN_GO(128, 32, false);
    else if (s.bn == 128)
        GCN_TN_GO(32, 128, false);
    else
        GCN_TN_GO(32, 32, false);
#undef GCN_TN_GO
    GCN_LAUNCH_CHECK();
    if (s.splits > 1) {
        const int64_t blocks = std::min<int64_t>(ceil_div(k * n, kBlock), 4096);
        reduce_splits_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(k, n, s.splits, out, c, ldc);
        GCN_LAUNCH_CHECK();
    }
    return GCN_OK;
}

static int make_in_mask(const uint32_t* bits, int64_t ld_words, int64_t width, float keep_prob, InMask* im) {
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    if (!bits) return fail(GCN_ERR_BAD_ARG, "mask is null");
    if (ld_words < (width + 31) / 32)
        return fail(GCN_ERR_BAD_ARG, "mask rows of %lld words cannot hold %lld bits", (long long)ld_words, (long long)width);
    im->bits = bits;
    im->ld_words = ld_words;
    im->keep_prob = keep_prob;
    im->inv_keep = 1.0f / keep_prob;
    uint32_t u;
    memcpy(&u, &keep_prob, 4);
    im->exact = (u & 0x007fffffu) == 0;
    return GCN_OK;
}

int gcn_gemm_tn_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* g, int64_t ldg,
                    float* c, int64_t ldc, void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_tn_entry(m, n, k, a, lda, g, ldg, c, ldc, workspace, workspace_bytes, stream, kNoInMask);
}

int gcn_gemm_tn_masked_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const uint32_t* mask,
                           int64_t ld_words, float keep_prob, const float* g, int64_t ldg, float* c, int64_t ldc,
                           void* workspace, size_t workspace_bytes, void* stream) {
    InMask im;
    const int rc = make_in_mask(mask, ld_words, k, keep_prob, &im);
    if (rc != GCN_OK) return rc;
    return gemm_tn_entry(m, n, k, a, lda, g, ldg, c, ldc, workspace, workspace_bytes, stream, im);
}

int gcn_gemm_masked_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const uint32_t* mask,
                        int64_t ld_words, float keep_prob, const float* b, int64_t ldb, int trans_b, float* c,
                        int64_t ldc, int flags, void* stream) {
    if (m < 0 || n < 0 || k < 0) return fail(GCN_ERR_BAD_ARG, "negative dimension");
    if (lda < k || ldc < n || ldb < (trans_b ? k : n)) return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (flags & ~GCN_RELU) return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    InMask im;
    const int rc = make_in_mask(mask, ld_words, k, keep_prob, &im);
    if (rc != GCN_OK) return rc;
    if (m == 0 || n == 0) return GCN_OK;
    if (!c || (k > 0 && (!a || !b))) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int relu = (flags & GCN_RELU) ? 1 : 0;
    if (n <= 32) return launch_gemm<128, 32>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, nullptr, &im);
    if (n <= 64) return launch_gemm<128, 64>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, nullptr, &im);
    return launch_gemm<128, 128>(m, n, k, a, lda, b, ldb, trans_b, c, ldc, relu, st, nullptr, 0, 1.0f, nullptr, &im);
}

int gcn_dropout_bits(int64_t n_rows, int64_t f, float keep_prob, uint64_t seed, uint64_t stream_id,
                     const uint64_t* stream_offset, uint64_t index_offset, int64_t pitch, uint32_t* bits_out,
                     int64_t ld_words, void* stream) {
    if (n_rows < 0 || f < 0 || pitch < f || ld_words < (f + 31) / 32) return fail(GCN_ERR_BAD_ARG, "bad shape");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    const int64_t words = (f + 31) / 32;
    if (n_rows == 0 || words == 0) return GCN_OK;
    if (!bits_out) return fail(GCN_ERR_BAD_ARG, "null pointer");
    const int64_t blocks = std::min<int64_t>(ceil_div(n_rows * words, kBlock), 1 << 20);
    dropout_bits_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(
        n_rows, f, words, keep_threshold(keep_prob), stream_key(seed, stream_id), index_offset, pitch, bits_out, ld_words,
        stream_offset, seed, stream_id);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

}  // extern "C"
