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
// 24 random bits -> u in [0,1), exactly representable in float32
__device__ __forceinline__ float uniform24(uint64_t key, uint64_t idx) {
    return (float)(uint32_t)(sm64(key ^ idx) >> 40) * 5.9604644775390625e-08f;  // 2^-24
}
// dropout predicate of gcn/layers.py:21-23: floor(keep_prob + u) == 1, evaluated in float32
__device__ __forceinline__ bool keep_entry(uint64_t key, uint64_t idx, float keep_prob) {
    return floorf(__fadd_rn(keep_prob, uniform24(key, idx))) >= 1.0f;
}

// Dropout applied to the OUTPUT of a product -- sparse x dense or dense x dense -- (the next layer's
// tf.nn.dropout on its input, gcn/layers.py:185, fused into the store): element (r, col) of C draws decision
// index_offset + r * pitch + col of stream `key` -- the decision a separate pass over a matrix
// `pitch` columns wide would draw, of which C may be a row block and a column range.
struct OutDrop {
    uint64_t key;
    float keep_prob;     // 1: off
    int64_t index_offset;
    int64_t pitch;
    const uint64_t* stream_offset;  // device value added to stream_id when the kernel runs (HIP-graph replays); may be null
    uint64_t seed, stream_id;
    // dense products only, may be null: row r of C is row row_ids[r] of the matrix the decisions are drawn
    // for (C holds a subset of its rows)
    const int64_t* row_ids;
};

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
