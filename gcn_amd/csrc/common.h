// Shared host/device helpers of libgcn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/gcn_hip.h"

namespace gcn {

// ---- per-thread error text behind gcn_last_error_string() --------------------------------
void set_error(const char* fmt, ...);
int fail(gcn_status code, const char* fmt, ...);

#define GCN_HIP_TRY(expr)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return ::gcn::fail(GCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                               __FILE__, __LINE__);                                                \
    } while (0)

#define GCN_LAUNCH_CHECK()                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = hipGetLastError();                                                                 \
        if (e__ != hipSuccess)                                                                              \
            return ::gcn::fail(GCN_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, \
                               __LINE__);                                                                   \
    } while (0)

// ---- counter-based random numbers: identical integer arithmetic on host, device and in the CPU oracle ----
__host__ __device__ __forceinline__ uint64_t sm64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// key of the random stream (seed, stream_id); computed once on the host per call
__host__ __device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t stream_id) {
    return sm64(sm64(seed) ^ stream_id);
}
// Dropout decisions: one 64-bit hash serves FOUR consecutive decision ids -- h = sm64(key ^ (idx >> 2)), and
// decision idx reads the 16-bit field (idx & 3) of h, most significant field first -- because the two 64-bit
// multiplications of a hash are what a dropout costs (integer multiplies run at a quarter of the vector
// rate): a kernel whose lane holds 4 consecutive ids hashes once for them.
//   u = field * 2^-16 in [0,1), exactly representable in float32
__host__ __device__ __forceinline__ uint64_t decision_hash(uint64_t key, uint64_t idx) { return sm64(key ^ (idx >> 2)); }
__host__ __device__ __forceinline__ uint32_t decision_field(uint64_t h, uint64_t idx) {
    return (uint32_t)(h >> (48 - 16 * (int)(idx & 3))) & 0xffffu;
}
__device__ __forceinline__ float uniform16(uint64_t key, uint64_t idx) {
    return (float)decision_field(decision_hash(key, idx), idx) * 1.52587890625e-05f;  // 2^-16
}
// dropout predicate of gcn/layers.py:21-23: floor(keep_prob + u) == 1, evaluated in float32
__device__ __forceinline__ bool keep_entry(uint64_t key, uint64_t idx, float keep_prob) {
    return floorf(__fadd_rn(keep_prob, uniform16(key, idx))) >= 1.0f;
}
// The same predicate as an integer comparison: field >= keep_threshold(keep_prob), the smallest field value
// whose u passes (the float32 sum keep_prob + u is monotone in u, so the two forms agree on every field value;
// 65536 when none passes).
__host__ __device__ __forceinline__ bool passes_f32(float keep_prob, uint32_t field) {
    const float sum = keep_prob + (float)field * 1.52587890625e-05f;   // one rounding: no contraction possible
    return sum >= 1.0f;                                                // floor(sum) >= 1 for a non-negative sum
}
static inline uint32_t keep_threshold(float keep_prob) {
    uint32_t lo = 0, hi = 65536;                // smallest t in [0, 65536] with passes(t); passes(65536) taken as true
    while (lo < hi) {
        const uint32_t mid = (lo + hi) / 2;
        if (passes_f32(keep_prob, mid))
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}
__device__ __forceinline__ bool keep_field(uint64_t h, uint64_t idx, uint32_t threshold) {
    return decision_field(h, idx) >= threshold;
}

// Dropout applied to the OUTPUT of a product -- sparse x dense or dense x dense -- (the next layer's
// tf.nn.dropout on its input, gcn/layers.py:185, fused into the store): element (r, col) of C draws decision
// index_offset + r * pitch + col of stream `key` -- the decision a separate pass over a matrix
// `pitch` columns wide would draw, of which C may be a row block and a column range.
struct OutDrop {
    uint64_t key;
    float keep_prob;     // 1: off (the launchers fill `threshold` from it: out_drop_ready)
    int64_t index_offset;
    int64_t pitch;
    const uint64_t* stream_offset;  // device value added to stream_id when the kernel runs (HIP-graph replays); may be null
    uint64_t seed, stream_id;
    // dense products only, may be null: row r of C is row row_ids[r] of the matrix the decisions are drawn
    // for (C holds a subset of its rows)
    const int64_t* row_ids;
    uint32_t threshold;  // keep_threshold(keep_prob)
};
static inline OutDrop out_drop_ready(OutDrop od) {
    od.threshold = keep_threshold(od.keep_prob);
    return od;
}

// Fixed bijection on [0, 2^bits): odd multiplications and xor-shifts, each invertible modulo 2^bits.
// The R-MAT generator relabels vertices with it so that hub vertices are spread over the id range.
__host__ __device__ __forceinline__ uint64_t scramble_id(uint64_t v, int bits, uint64_t key) {
    const uint64_t mask = (bits >= 64) ? ~0ull : ((1ull << bits) - 1);
    const int sh = (bits + 1) / 2;
    uint64_t x = (v + key) & mask;
    x = (x * 0x9E3779B97F4A7C15ull) & mask;
    x ^= x >> sh;
    x = (x * 0xBF58476D1CE4E5B9ull) & mask;
    x ^= x >> sh;
    return x;
}

// long-row tables of a product plan as plan creation builds them on the host (spmm.hip)
struct LongTableHost {
    std::vector<int32_t> long_row;    // rows longer than the split threshold, ascending
    std::vector<int64_t> long_chunk;  // [n_long + 1] first chunk of each such row
    std::vector<int32_t> chunk_long;  // [n_chunks] slot in long_row of each chunk
};
void build_long_table(const std::vector<int32_t>& rows, const std::vector<int64_t>& starts,
                      const std::vector<int64_t>& ends, int64_t split, int64_t chunk_len, LongTableHost* out);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace gcn
