// The elementwise tail of a training step as two kernels: masked softmax cross-entropy with its
// gradient (gcn/metrics.py:85-101, gcn/models.py:92-99) and TensorFlow's Adam update
// (gcn/models.py:35-38, 86).  On the citation graphs these replace ~35 launches of a few
// microseconds each; on a 2^24-node graph they read the logits once instead of a dozen times.
#include <algorithm>

#include "common.h"

namespace gcn {

constexpr int kBlock = 256;
constexpr int kXentTile = 32;   // wave-sized row groups per block of the loss kernel on a large matrix

// row_stats[i] = (CE_i * w_i, correct_i * w_i) with w_i = mask_i / mean(mask);
// dlogits[i][c] = d(mean_i CE_i w_i) / d logits[i][c] = (softmax_ic * sum_c labels_ic - labels_ic) * w_i / n.
// An all-zero label row (CiteSeer's padded nodes) has CE 0 and argmax 0, as TensorFlow gives.
//
// G lanes per node (the power of two that holds the classes), one class per lane: a wave reads and
// writes 64 consecutive floats of the row-major matrices per instruction, and max / argmax / sums
// are butterflies inside the lane group.  (One thread per node reads its row at a 4 c-byte stride
// several times over: 7.8 ms on 2^24 x 16 logits against 0.7 ms for this form.)
template <int G>
__global__ void softmax_xent_group_kernel(int64_t n, int c, const float* __restrict__ logits, int64_t ldx,
                                          const int32_t* __restrict__ labels, int64_t ldl,
                                          const int32_t* __restrict__ mask, float mask_mean,
                                          float* __restrict__ row_stats, float* __restrict__ dlogits, int64_t ldd,
                                          int tile) {
    constexpr int kRows = 64 / G;                       // nodes per wave
    const float inv_n = 1.0f / (float)n;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G;
    // a block walks a contiguous stretch of `tile` wave-sized row groups, its waves interleaved:
    // consecutive blocks read consecutive memory
    const int64_t n_groups = (n + kRows - 1) / kRows;
    const int64_t g_begin = (int64_t)blockIdx.x * tile;
    const int64_t g_end = g_begin + tile < n_groups ? g_begin + tile : n_groups;
    for (int64_t g = g_begin + (threadIdx.x >> 6); g < g_end; g += kBlock / 64) {   // uniform per wave: every lane takes part in the shuffles
        const int64_t i = g * kRows + lane / G;
        const bool live = i < n && sub < c;
        float x = live ? logits[i * ldx + sub] : -INFINITY;
        int32_t lab = live ? labels[i * ldl + sub] : INT32_MIN;
        float mx = x;
        int arg_x = sub, arg_l = sub;
        int32_t best_l = lab;
        float lab_sum = live ? (float)lab : 0.0f;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) {
            const float ov = __shfl_xor(mx, d);
            const int oa = __shfl_xor(arg_x, d);
            if (ov > mx || (ov == mx && oa < arg_x)) {      // first maximum wins, as argmax does
                mx = ov;
                arg_x = oa;
            }
            const int32_t ol = __shfl_xor(best_l, d);
            const int ola = __shfl_xor(arg_l, d);
            if (ol > best_l || (ol == best_l && ola < arg_l)) {
                best_l = ol;
                arg_l = ola;
            }
            lab_sum += __shfl_xor(lab_sum, d);
        }
        float s = live ? expf(x - mx) : 0.0f;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) s += __shfl_xor(s, d);
        const float lse = mx + logf(s);
        float ce = live ? (float)lab * (lse - x) : 0.0f;
#pragma unroll
        for (int d = 1; d < G; d <<= 1) ce += __shfl_xor(ce, d);
        if (i < n) {
            const float w = (float)mask[i] / mask_mean;
            if (sub == 0) {
                row_stats[2 * i] = ce * w;
                row_stats[2 * i + 1] = (arg_x == arg_l) ? w : 0.0f;
            }
            if (dlogits && sub < c) dlogits[i * ldd + sub] = (expf(x - lse) * lab_sum - (float)lab) * (w * inv_n);
        }
    }
}

// More than 64 classes: one thread per node.
__global__ void softmax_xent_kernel(int64_t n, int c, const float* __restrict__ logits, int64_t ldx,
                                    const int32_t* __restrict__ labels, int64_t ldl,
                                    const int32_t* __restrict__ mask, float mask_mean,
                                    float* __restrict__ row_stats, float* __restrict__ dlogits, int64_t ldd) {
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float* x = logits + i * ldx;
        const int32_t* lab = labels + i * ldl;
        float mx = x[0];
        int arg_x = 0, arg_l = 0;
        int32_t best_l = lab[0];
        float lab_sum = (float)lab[0];
        for (int j = 1; j < c; ++j) {
            const float v = x[j];
            if (v > mx) {
                mx = v;
                arg_x = j;
            }
            const int32_t l = lab[j];
            if (l > best_l) {
                best_l = l;
                arg_l = j;
            }
            lab_sum += (float)l;
        }
        float s = 0.0f;
        for (int j = 0; j < c; ++j) s += expf(x[j] - mx);
        const float lse = mx + logf(s);
        float ce = 0.0f;
        for (int j = 0; j < c; ++j) ce += (float)lab[j] * (lse - x[j]);
        const float w = (float)mask[i] / mask_mean;
        row_stats[2 * i] = ce * w;
        row_stats[2 * i + 1] = (arg_x == arg_l) ? w : 0.0f;
        if (dlogits) {
            const float scale = w * inv_n;
            float* d = dlogits + i * ldd;
            for (int j = 0; j < c; ++j) d[j] = (expf(x[j] - lse) * lab_sum - (float)lab[j]) * scale;
        }
    }
}

// loss and accuracy of a step from the per-node rows of the loss kernel and the weights:
//   out[0] = mean_i stats[i][0] + weight_decay * sum_j w_j^2 / 2     (gcn/models.py:92-99)
//   out[1] = mean_i stats[i][1]                                       (gcn/models.py:106-108)
// Sums are taken in float64 in a fixed order (thread-strided, then a tree inside the block, then the
// blocks in order), so the result does not depend on scheduling.  One block writes the result itself;
// more blocks leave three partial sums each in `partial` for loss_summary_final_kernel.
constexpr int kSummaryBlock = 1024;

__device__ __forceinline__ double block_sum(double v, double* lds) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += lds[w];
    return t;
}

__global__ __launch_bounds__(kSummaryBlock) void loss_summary_kernel(int64_t n, const float* __restrict__ stats,
                                                                    int64_t n_w, const float* __restrict__ w,
                                                                    float weight_decay, double* __restrict__ partial,
                                                                    float* __restrict__ out) {
    __shared__ double lds[kSummaryBlock / 64];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s0 = 0.0, s1 = 0.0, sw = 0.0;
    const float2* rows = reinterpret_cast<const float2*>(stats);
    for (int64_t i = first; i < n; i += stride) {
        const float2 r = rows[i];
        s0 += (double)r.x;
        s1 += (double)r.y;
    }
    for (int64_t j = first; j < n_w; j += stride) sw += (double)w[j] * (double)w[j];
    s0 = block_sum(s0, lds);
    s1 = block_sum(s1, lds);
    sw = block_sum(sw, lds);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1) {
            out[0] = (float)(s0 / (double)n + 0.5 * (double)weight_decay * sw);
            out[1] = (float)(s1 / (double)n);
        } else {
            partial[3 * blockIdx.x] = s0;
            partial[3 * blockIdx.x + 1] = s1;
            partial[3 * blockIdx.x + 2] = sw;
        }
    }
}

__global__ void loss_summary_final_kernel(int64_t n, int blocks, const double* __restrict__ partial, float weight_decay,
                                          float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s0 = 0.0, s1 = 0.0, sw = 0.0;
    for (int b = 0; b < blocks; ++b) {
        s0 += partial[3 * b];
        s1 += partial[3 * b + 1];
        sw += partial[3 * b + 2];
    }
    out[0] = (float)(s0 / (double)n + 0.5 * (double)weight_decay * sw);
    out[1] = (float)(s1 / (double)n);
}

static inline int summary_blocks(int64_t n, int64_t n_w) {
    const int64_t work = std::max(n, n_w);
    return (int)std::min<int64_t>(256, std::max<int64_t>(1, work / (kSummaryBlock * 16)));
}

template <int G>
static void launch_xent_group(int64_t n, int c, const float* logits, int64_t ldx, const int32_t* labels, int64_t ldl,
                              const int32_t* mask, float mask_mean, float* row_stats, float* dlogits, int64_t ldd,
                              hipStream_t stream) {
    const int64_t n_groups = ceil_div(n, (int64_t)(64 / G));
    // one group per wave until that gives more than ~8 blocks per CU (a citation graph: 85 blocks), then longer tiles
    const int tile = (int)std::min<int64_t>(kXentTile, std::max<int64_t>(kBlock / 64, ceil_div(n_groups, 2048)));
    const int64_t blocks = ceil_div(n_groups, (int64_t)tile);
    softmax_xent_group_kernel<G><<<(unsigned)blocks, kBlock, 0, stream>>>(n, c, logits, ldx, labels, ldl, mask, mask_mean,
                                                                         row_stats, dlogits, ldd, tile);
}

// g' = g + wd * p;  m += (1 - b1)(g' - m);  v += (1 - b2)(g'^2 - v);  p -= lr_t * m / (sqrt(v) + eps).
// lr_t comes from device memory when lr_dev is set (a launch recorded in a HIP graph).
__global__ void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr_t, const float* __restrict__ lr_dev, float one_minus_b1,
                            float one_minus_b2, float eps, float wd) {
    if (lr_dev) lr_t = *lr_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float w = p[i];
        const float gi = __fadd_rn(g[i], __fmul_rn(wd, w));
        const float mi = __fadd_rn(m[i], __fmul_rn(one_minus_b1, __fsub_rn(gi, m[i])));
        const float vi = __fadd_rn(v[i], __fmul_rn(one_minus_b2, __fsub_rn(__fmul_rn(gi, gi), v[i])));
        m[i] = mi;
        v[i] = vi;
        p[i] = __fsub_rn(w, __fdiv_rn(__fmul_rn(lr_t, mi), __fadd_rn(__fsqrt_rn(vi), eps)));
    }
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_softmax_xent_f32(int64_t n, int64_t c, const float* logits, int64_t ldx, const int32_t* labels, int64_t ldl,
                         const int32_t* mask, float mask_mean, float* row_stats, float* dlogits, int64_t ldd,
                         void* stream) {
    if (n < 0 || c < 1 || c > INT32_MAX || ldx < c || ldl < c || (dlogits && ldd < c))
        return fail(GCN_ERR_BAD_ARG, "bad shape: n=%lld c=%lld", (long long)n, (long long)c);
    if (!(mask_mean > 0.0f)) return fail(GCN_ERR_BAD_ARG, "mask_mean must be positive (an empty mask has no loss)");
    if (n == 0) return GCN_OK;
    if (n > ((int64_t)1 << 35)) return fail(GCN_ERR_BAD_ARG, "more than 2^35 rows in one launch");
    if (!logits || !labels || !mask || !row_stats) return fail(GCN_ERR_BAD_ARG, "null pointer");
    hipStream_t st = (hipStream_t)stream;
#define GCN_XENT_GROUP(G) \
    launch_xent_group<G>(n, (int)c, logits, ldx, labels, ldl, mask, mask_mean, row_stats, dlogits, ldd, st)
    if (c <= 1) GCN_XENT_GROUP(1);
    else if (c <= 2) GCN_XENT_GROUP(2);
    else if (c <= 4) GCN_XENT_GROUP(4);
    else if (c <= 8) GCN_XENT_GROUP(8);
    else if (c <= 16) GCN_XENT_GROUP(16);
    else if (c <= 32) GCN_XENT_GROUP(32);
    else if (c <= 64) GCN_XENT_GROUP(64);
    else {
        const int64_t blocks = std::min<int64_t>(ceil_div(n, kBlock), 1 << 16);
        softmax_xent_kernel<<<(unsigned)blocks, kBlock, 0, st>>>(n, (int)c, logits, ldx, labels, ldl, mask, mask_mean,
                                                                row_stats, dlogits, ldd);
    }
#undef GCN_XENT_GROUP
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_loss_summary_workspace_bytes(int64_t n, int64_t n_weights, size_t* bytes) {
    if (!bytes || n < 0 || n_weights < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    const int blocks = summary_blocks(n, n_weights);
    *bytes = blocks > 1 ? (size_t)blocks * 3 * sizeof(double) : 0;
    return GCN_OK;
}

int gcn_loss_summary_f32(int64_t n, const float* row_stats, int64_t n_weights, const float* weights, float weight_decay,
                         void* workspace, size_t workspace_bytes, float* out2, void* stream) {
    if (n < 1 || n_weights < 0) return fail(GCN_ERR_BAD_ARG, "bad size: n=%lld n_weights=%lld", (long long)n,
                                            (long long)n_weights);
    if (!row_stats || !out2 || (n_weights > 0 && !weights)) return fail(GCN_ERR_BAD_ARG, "null pointer");
    if (((uintptr_t)row_stats & 7) != 0) return fail(GCN_ERR_BAD_ARG, "row_stats must be 8-byte aligned");
    const int blocks = summary_blocks(n, n_weights);
    hipStream_t st = (hipStream_t)stream;
    if (blocks > 1) {
        const size_t need = (size_t)blocks * 3 * sizeof(double);
        if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 7) != 0)
            return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed (8-byte aligned)", workspace_bytes,
                        need);
    }
    loss_summary_kernel<<<blocks, kSummaryBlock, 0, st>>>(n, row_stats, n_weights, weights, weight_decay,
                                                         (double*)workspace, out2);
    GCN_LAUNCH_CHECK();
    if (blocks > 1) {
        loss_summary_final_kernel<<<1, 64, 0, st>>>(n, blocks, (const double*)workspace, weight_decay, out2);
        GCN_LAUNCH_CHECK();
    }
    return GCN_OK;
}

int gcn_adam_f32(int64_t n, float* param, const float* grad, float* m, float* v, float lr_t, const float* lr_t_device,
                 float beta1, float beta2, float epsilon, float weight_decay, void* stream) {
    if (n < 0) return fail(GCN_ERR_BAD_ARG, "negative size");
    if (n == 0) return GCN_OK;
    if (!param || !grad || !m || !v) return fail(GCN_ERR_BAD_ARG, "null pointer");
    const int64_t blocks = std::min<int64_t>(ceil_div(n, kBlock), 1 << 16);
    adam_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n, param, grad, m, v, lr_t, lr_t_device,
                                                                     1.0f - beta1, 1.0f - beta2, epsilon, weight_decay);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

}  // extern "C"
