// CSR x dense products of the GCN layer: A_hat.(XW) (gcn/layers.py:202) and X'.W0 with fused
// sparse dropout (gcn/layers.py:19-25, 198-199).  HBM-bound gather kernels for gfx950.
//
// Work decomposition
//   strips : the work axis -- entries plus kRowCost per row, so that a run of very short rows does
//            not count as no work -- is cut every strip_nnz units (256; fewer on a matrix too small
//            to fill the chip with such strips); the rows that START inside a strip belong to
//            it, so strips carry equal work however skewed the degrees are.  A lane group of LPR lanes
//            owns one row at a time; a lane holds VPL vectors of VEC floats of that row.  Wide rows
//            (LPR = 64): one wave per strip, rows one after the other.  Narrow rows (LPR < 64): a
//            wave takes 64/LPR consecutive strips and its lane groups pull the next row of that
//            range whenever they finish one, so no group idles behind a longer neighbour.
//   chunks : rows longer than the split threshold (256 entries for narrow operands, 1024 for wide
//            ones) are left out of the strips, cut in chunks of that many entries, each chunk summed by one lane group into a partial row, and the partial rows
//            of a row are added in chunk order by a second kernel.
// Every row of at most 256 entries is therefore accumulated sequentially in storage order with
// a separate multiply and add (this file is compiled with -ffp-contract=off): bit-identical to the
// serial CPU loop.
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "common.h"

struct gcn_spmm_plan {
    int64_t n_rows, n_cols, nnz;
    int64_t n_strips;
    int strip_nnz;       // entries per strip
    int chunk_len;       // entries per chunk of a long row; 0: the split threshold itself
    int split[2];        // rows longer than this are summed chunk-wise (narrow / wide dense operands)
    int device;
    int32_t* strip_row;  // [n_strips + 1] first row starting in each strip
    // rows longer than the split threshold, cut in chunks of that many entries; table 0 serves narrow
    // dense operands (threshold kSplitNarrow), table 1 wide ones (kSplitWide)
    struct LongTable {
        int64_t n_long, n_chunks;
        int32_t* long_row;    // [n_long] ascending
        int64_t* long_chunk;  // [n_long + 1] first chunk of each long row
        int32_t* chunk_long;  // [n_chunks] long-row slot of each chunk
    } table[2];
    // columns >= hot_cols are loaded with the non-temporal hint by the wave-per-row instances (0: off).
    // After a degree-descending relabel the first columns are the hub rows of B: the rows worth caching.
    int64_t hot_cols;
    // narrow products stage the first lds_hub_rows rows of B in LDS, once per workgroup, and entries whose
    // column is below that number read them from there (0: off).  For a degree-ordered matrix those are the
    // hub rows: the "LDS-staged row-block" form of the product.
    int64_t lds_hub_rows;
    int timing;           // record events around the kernels of every product
    hipEvent_t ev[3];     // start, after the row kernel, after the long-row reduction
};

// Scratch of gcn_spmm_csr_f32_noalloc: the partial rows of long-row chunks, which the other entry
// points take from a caller workspace.  One copy per GPU the library runs on, held until the process
// ends.  In host code the array's name is only a handle for hipGetSymbolAddress: kernels and memory
// calls get the per-device address that call returns, never the name.
constexpr size_t kScratchBytes = (size_t)32 << 20;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace gcn {

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each GPU, resolved once
static std::mutex g_scratch_mutex;

// The address of `scratch` on the CURRENT device (each GPU holds its own copy at its own address).
// Resolved at set-up -- plan creation calls this on the plan's device -- and cached.
static int scratch_on_current_device(void** out) {
    int dev = -1;
    GCN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(GCN_ERR_BAD_ARG, "device ordinal %d is not supported", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        GCN_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(GCN_ERR_HIP, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return GCN_OK;
}

constexpr int kStripNnz = 256;       // entries per strip on a matrix large enough to fill the chip with such strips
constexpr int kStripNnzMin = 16;
constexpr int kChunkSmall = 64;
constexpr int kRowCost = 4;          // a row's fixed cost (pointer loads, store) in units of one entry
constexpr int64_t kStripsWanted = 8192;  // 8 waves on each of the 1024 SIMDs

// A strip is one wave's sequential work (wide operands).  With 256-entry strips a matrix of 10^5
// entries gives a few hundred waves, each walking 64 dependent batches of loads: the product is
// then bound by that latency chain, not by bandwidth.  Shorter strips trade it for parallelism.
static inline int strip_nnz_for(int64_t work) {
    int s = kStripNnz;
    while (s > kStripNnzMin && work / s < kStripsWanted) s >>= 1;
    return s;
}
constexpr int kSplitNarrow = 256;   // lane groups narrower than a wave: every work item <= 256 entries
constexpr int kSplitWide = 1024;    // one wave per row
constexpr int kBlock = 256;
constexpr int kUnroll = 4;       // entries in flight per lane group
// A matrix too small to fill the chip finishes when its longest row does: that row's entries are a
// chain of dependent batches walked by one lane group.  Narrow operands then take 16 entries per
// batch (same order of additions, same bits; the extra loads of short rows hit the clamped, cached
// address).  Cora, F = 16: 29 -> 12 us per product.
constexpr int kUnrollSmall = 16;
// LDS a workgroup of a narrow product may fill with hub rows of B: 32 KiB leaves room for five
// workgroups per CU, the number the kernel's registers allow
constexpr int64_t kHubLdsBytes = 32 * 1024;

// ------------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------------
__global__ void strip_rows_kernel(int64_t n_rows, const int64_t* __restrict__ indptr, int64_t n_strips,
                                  int strip_nnz, int32_t* __restrict__ strip_row) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > n_strips) return;
    if (w == n_strips) {
        strip_row[w] = (int32_t)n_rows;  // trailing empty rows belong to the last strip
        return;
    }
    const int64_t target = w * strip_nnz;
    int64_t lo = 0, hi = n_rows;  // smallest r in [0, n_rows] with work before row r >= target
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] + mid * kRowCost >= target)
            hi = mid;
        else
            lo = mid + 1;
    }
    strip_row[w] = (int32_t)lo;
}

__global__ void collect_long_rows_kernel(int64_t n_rows, const int64_t* __restrict__ indptr, int split, int64_t capacity,
                                         int32_t* __restrict__ rows, int64_t* __restrict__ starts,
                                         int64_t* __restrict__ ends, unsigned long long* __restrict__ counter) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t k0 = indptr[r], k1 = indptr[r + 1];
    if (k1 - k0 > split) {
        const unsigned long long slot = atomicAdd(counter, 1ull);
        if ((int64_t)slot < capacity) {
            rows[slot] = (int32_t)r;
            starts[slot] = k0;
            ends[slot] = k1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// product kernels
// ------------------------------------------------------------------------------------------------
template <int VEC>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<4> {
    using type = float4;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&out)[VEC]);
template <>
__device__ __forceinline__ void load_vec<1>(const float* p, float (&out)[1]) {
    out[0] = *p;
}
template <>
__device__ __forceinline__ void load_vec<4>(const float* p, float (&out)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
}
// the same loads with the non-temporal hint (`nt`: the line is marked for early eviction)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float* p, float (&out)[VEC]);
template <>
__device__ __forceinline__ void load_vec_nt<1>(const float* p, float (&out)[1]) {
    out[0] = __builtin_nontemporal_load(p);
}
template <>
__device__ __forceinline__ void load_vec_nt<4>(const float* p, float (&out)[4]) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
    out[3] = v.w;
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]);
template <>
__device__ __forceinline__ void store_vec<1>(float* p, const float (&v)[1]) {
    *p = v[0];
}
template <>
__device__ __forceinline__ void store_vec<4>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

struct DropArgs {
    const int64_t* perm;
    uint64_t key;
    float keep_prob;
    float inv_keep;
    // when stream_offset is set the kernel derives the key itself from (seed, stream_id + *stream_offset):
    // a launch recorded in a HIP graph then draws a new mask on every replay
    const uint64_t* stream_offset;
    uint64_t seed, stream_id;
    uint32_t threshold;   // keep_threshold(keep_prob): an entry is kept iff its 16-bit field is >= this
};

// sparse_dropout decisions of one batch of U entries starting at kb, for a lane group of LPR lanes.
// Every lane of a group needs all U decisions, but a decision costs a 64-bit hash: lane `sub` of the
// group hashes only entries sub, sub + LPR, ... and a wave ballot hands the bits round.
// entry_ids: this lane's entries' ids (perm[k] or k), fetched with the batch.
template <int LPR, int U>
struct DropIds {
    static constexpr int kRounds = (U + LPR - 1) / LPR;
    uint64_t v[kRounds];
};

template <int LPR, int U>
__device__ __forceinline__ void fetch_drop_ids(const DropArgs& dr, int64_t kb, int64_t k_last, int sub,
                                               DropIds<LPR, U>& ids) {
#pragma unroll
    for (int j = 0; j < DropIds<LPR, U>::kRounds; ++j) {
        const int64_t kj = kb + j * LPR + sub;
        const int64_t ku = kj < k_last ? kj : k_last;
        ids.v[j] = dr.perm ? (uint64_t)dr.perm[ku] : (uint64_t)ku;
    }
}

template <int LPR, int U>
__device__ __forceinline__ void drop_decisions(const DropArgs& dr, const DropIds<LPR, U>& ids, int group_base,
                                               bool (&kept)[U]) {
#pragma unroll
    for (int j = 0; j < DropIds<LPR, U>::kRounds; ++j) {
        const uint64_t bits = __ballot(keep_field(decision_hash(dr.key, ids.v[j]), ids.v[j], dr.threshold));
#pragma unroll
        for (int i = 0; i < LPR; ++i)
            if (j * LPR + i < U) kept[j * LPR + i] = (bits >> (group_base + i)) & 1ull;
    }
}

// A dense operand stored in two parts: its rows [0, n_first) at `first` (row pitch ld_first), the others in the
// matrix the kernel is given as b -- whose pointer the host has moved back by n_first rows, so that row c of the
// operand is b + c * ldb there too.  A row-sharded product reads its own block of rows where it lies and only the
// rows received from other ranks from the exchange buffer: the own rows are never copied.
struct TwoPart {
    const float* first;
    uint32_t ld_first;
    uint32_t n_first;
};
template <bool SEG>
__device__ __forceinline__ const float* operand_row(const float* b, uint32_t ldb, const TwoPart& tp, uint32_t c) {
    if (SEG) {
        const bool in_first = c < tp.n_first;
        const float* base = in_first ? tp.first : b;
        const uint32_t ld = in_first ? tp.ld_first : ldb;
        return base + (uint64_t)c * (uint64_t)ld;
    }
    return b + (uint64_t)c * (uint64_t)ldb;
}

// acc += sum over entries k0..k1-1 (in order) of a_k * B[col_k, this lane's columns].
// Entries are taken kUnroll at a time: the B rows of one batch and the (index, value) pairs of the
// next batch are all in flight together.  Addresses past the end of the range are clamped to its
// last entry (a valid, cached address) so that no load sits behind a branch; what such a slot
// loads is never added.
// NT (wave-per-row instances only: the column of an entry is the same in every lane): rows of B whose
// column is >= hot are loaded with the non-temporal hint, so that they do not push the hub rows out of
// the caches.
template <int LPR, int VEC, int VPL, bool DROPOUT, bool NT = false, bool SEG = false>
__device__ __forceinline__ void accumulate_range(int64_t k0, int64_t k1, const int32_t* __restrict__ indices,
                                                 const float* __restrict__ vals, const float* __restrict__ b,
                                                 uint32_t ldb, const int (&lcol)[VPL], const DropArgs& dr,
                                                 float (&acc)[VPL][VEC], uint32_t hot = 0,
                                                 const TwoPart& tp = TwoPart{nullptr, 0, 0}) {
    if (k0 >= k1) return;
    const int64_t k_last = k1 - 1;
    const int sub = (threadIdx.x & 63) % LPR, group_base = (threadIdx.x & 63) - sub;
    uint32_t c[kUnroll];
    float a[kUnroll];
    DropIds<LPR, kUnroll> ids;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
        const int64_t ku = (k0 + u < k_last) ? k0 + u : k_last;
        c[u] = (uint32_t)indices[ku];
        a[u] = vals[ku];
    }
    if (DROPOUT) fetch_drop_ids<LPR, kUnroll>(dr, k0, k_last, sub, ids);
    for (int64_t k = k0;; k += kUnroll) {
        float bv[kUnroll][VPL][VEC];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const float* brow = operand_row<SEG>(b, ldb, tp, c[u]);
            if (NT && (uint32_t)__builtin_amdgcn_readfirstlane((int)c[u]) >= hot) {
#pragma unroll
                for (int v = 0; v < VPL; ++v) load_vec_nt<VEC>(brow + lcol[v], bv[u][v]);
            } else {
#pragma unroll
                for (int v = 0; v < VPL; ++v) load_vec<VEC>(brow + lcol[v], bv[u][v]);
            }
        }
        float a_now[kUnroll];
        bool kept[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            a_now[u] = DROPOUT ? __fmul_rn(a[u], dr.inv_keep) : a[u];
            kept[u] = true;
        }
        if (DROPOUT) drop_decisions<LPR, kUnroll>(dr, ids, group_base, kept);
        // (index, value) pairs of the next batch
        const int64_t kn = k + kUnroll;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t ku = (kn + u < k_last) ? kn + u : k_last;
            c[u] = (uint32_t)indices[ku];
            a[u] = vals[ku];
        }
        if (DROPOUT) fetch_drop_ids<LPR, kUnroll>(dr, kn, k_last, sub, ids);
        const int64_t rem = k1 - k;
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (u < rem && kept[u]) {
#pragma unroll
                for (int v = 0; v < VPL; ++v)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[v][e] = __fadd_rn(acc[v][e], __fmul_rn(a_now[u], bv[u][v][e]));
            }
        }
        if (rem <= kUnroll) break;
    }
}

// The same sums with two batches of B rows in flight (more registers, lower occupancy): for matrices too
// small to hide latency with occupancy, where the longest row's chain of dependent batches ends the kernel.
template <int LPR, int VEC, int VPL, bool DROPOUT>
__device__ __forceinline__ void accumulate_range_piped(int64_t k0, int64_t k1, const int32_t* __restrict__ indices,
                                                 const float* __restrict__ vals, const float* __restrict__ b,
                                                 uint32_t ldb, const int (&lcol)[VPL], const DropArgs& dr,
                                                 float (&acc)[VPL][VEC]) {
    if (k0 >= k1) return;
    const int64_t k_last = k1 - 1;
    const int sub = (threadIdx.x & 63) % LPR, group_base = (threadIdx.x & 63) - sub;
    uint32_t c[kUnroll];
    float a[kUnroll];
    DropIds<LPR, kUnroll> ids;
    auto fetch_entries = [&](int64_t kb) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t ku = (kb + u < k_last) ? kb + u : k_last;
            c[u] = (uint32_t)indices[ku];
            a[u] = vals[ku];
        }
        if (DROPOUT) fetch_drop_ids<LPR, kUnroll>(dr, kb, k_last, sub, ids);
    };
    // two batches of B rows in flight: while batch n is added, the rows of batch n+1 are on their way and
    // the entries of batch n+2 are being fetched -- a long row is a chain of dependent batches, and this
    // halves the number of memory latencies it pays
    float bv[2][kUnroll][VPL][VEC];
    float a_b[2][kUnroll];
    bool kept[2][kUnroll];
    auto issue_rows = [&](int s) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const float* brow = b + (uint64_t)c[u] * (uint64_t)ldb;
#pragma unroll
            for (int v = 0; v < VPL; ++v) load_vec<VEC>(brow + lcol[v], bv[s][u][v]);
            a_b[s][u] = DROPOUT ? __fmul_rn(a[u], dr.inv_keep) : a[u];
            kept[s][u] = true;
        }
        if (DROPOUT) drop_decisions<LPR, kUnroll>(dr, ids, group_base, kept[s]);
    };
    auto add_batch = [&](int s, int64_t rem) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (u < rem && kept[s][u]) {
#pragma unroll
                for (int v = 0; v < VPL; ++v)
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        acc[v][e] = __fadd_rn(acc[v][e], __fmul_rn(a_b[s][u], bv[s][u][v][e]));
            }
        }
    };
    fetch_entries(k0);
    issue_rows(0);
    fetch_entries(k0 + kUnroll);
    for (int64_t k = k0;;) {
        issue_rows(1);
        fetch_entries(k + 2 * kUnroll);
        add_batch(0, k1 - k);
        k += kUnroll;
        if (k >= k1) break;
        issue_rows(0);
        fetch_entries(k + 2 * kUnroll);
        add_batch(1, k1 - k);
        k += kUnroll;
        if (k >= k1) break;
    }
}

// out = act(acc + addend): the addend row (may be null) is the "+ X" of a smoothing step
template <int VEC, int VPL>
__device__ __forceinline__ void store_row(float* dst, const float* addend, const int (&col)[VPL],
                                          const bool (&on)[VPL], float (&acc)[VPL][VEC], bool relu,
                                          const OutDrop& od, int64_t row) {
#pragma unroll
    for (int v = 0; v < VPL; ++v)
        if (on[v]) {
            if (addend) {
                float d[VEC];
                load_vec<VEC>(addend + col[v], d);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[v][e] = __fadd_rn(acc[v][e], d[e]);
            }
            if (relu) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[v][e] = acc[v][e] > 0.0f ? acc[v][e] : 0.0f;
            }
            if (od.keep_prob < 1.0f) {  // TF 1.x: div(x, keep_prob) * floor(keep_prob + uniform)
                const uint64_t id = (uint64_t)(od.index_offset + row * od.pitch + col[v]);
                // four consecutive ids starting at a multiple of 4 share one hash
                const bool one_hash = VEC == 4 && (id & 3) == 0;
                const uint64_t h4 = one_hash ? decision_hash(od.key, id) : 0;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const uint64_t h = one_hash ? h4 : decision_hash(od.key, id + e);
                    acc[v][e] = keep_field(h, id + e, od.threshold) ? __fdiv_rn(acc[v][e], od.keep_prob) : 0.0f;
                }
            }
            store_vec<VEC>(dst + col[v], acc[v]);
        }
}

// rows of ordinary length; one wave per strip
template <int LPR, int VEC, int VPL, bool DROPOUT, bool PIPED, bool NT = false, bool SEG = false>
__device__ __forceinline__ void strips_block(int64_t block, int64_t n_strips, const int32_t* __restrict__ strip_row,
                                             const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                             const float* __restrict__ vals, int64_t f, const float* __restrict__ b,
                                             uint32_t ldb, float* __restrict__ c, int64_t ldc, int relu,
                                             const DropArgs& dr, int split, const float* __restrict__ add,
                                             int64_t ldadd, const OutDrop& od, uint32_t hot = 0,
                                             const TwoPart& tp = TwoPart{nullptr, 0, 0}) {
    constexpr int kGroups = 64 / LPR;
    constexpr int kTile = LPR * VEC * VPL;  // columns covered per pass
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t strip = block * (kBlock / 64) + (threadIdx.x >> 6);
    if (strip >= n_strips) return;
    const int r_lo = strip_row[strip];
    const int r_hi = strip_row[strip + 1];
    for (int64_t col0 = 0; col0 < f; col0 += kTile) {
        int col[VPL];   // first column of each of this lane's vectors
        int lcol[VPL];  // the column it loads from: clamped into the row so that idle lanes load valid data
        bool on[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            col[v] = (int)col0 + (v * LPR + sub) * VEC;
            on[v] = col[v] < f;
            lcol[v] = on[v] ? col[v] : (int)f - VEC;
        }
        for (int r = r_lo + grp; r < r_hi; r += kGroups) {
            const int64_t k0 = indptr[r];
            const int64_t k1 = indptr[r + 1];
            if (k1 - k0 > split) continue;  // summed chunk-wise
            float acc[VPL][VEC];
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[v][e] = 0.0f;
            if (PIPED)
                accumulate_range_piped<LPR, VEC, VPL, DROPOUT>(k0, k1, indices, vals, b, ldb, lcol, dr, acc);
            else
                accumulate_range<LPR, VEC, VPL, DROPOUT, NT, SEG>(k0, k1, indices, vals, b, ldb, lcol, dr, acc, hot, tp);
            store_row<VEC, VPL>(c + (int64_t)r * ldc, add ? add + (int64_t)r * ldadd : nullptr, col, on, acc,
                                relu != 0, od, r);
        }
    }
}

// rows of ordinary length, narrow dense operand: a wave owns 64/LPR consecutive strips; each lane
// group takes the next unclaimed row of that range when it is about to finish its own.
//  - The claim is made one batch EARLY (while the last batch of the current row is in flight) and the
//    first (column, value) batch of the claimed row is fetched then, so that a group switching rows has
//    its entries in registers already: waits are wave-wide, and a wave of 16 groups over 50-entry rows
//    has some group switching in almost every turn -- fetching at the switch made every turn pay two
//    dependent memory latencies instead of one.
//  - Row pointers come from a register window (indptr of 64 consecutive rows, one per lane, read with
//    a cross-lane shuffle), not from memory at claim time.
//  - HUBS: the first `hub_rows` rows of B (f columns each, row-major) sit in LDS at `hub_lds`; an entry whose
//    column is below hub_rows takes its row from there instead of from memory.
template <int LPR, int VEC, int VPL, bool DROPOUT, int kUnroll, bool HUBS = false, bool SEG = false>
__device__ __forceinline__ void slots_block(int64_t block, int64_t n_strips, const int32_t* __restrict__ strip_row,
                                            const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                            const float* __restrict__ vals, int64_t f, const float* __restrict__ b,
                                            uint32_t ldb, float* __restrict__ c, int64_t ldc, int relu,
                                            const DropArgs& dr, int split, const float* __restrict__ add,
                                            int64_t ldadd, int64_t nnz, const OutDrop& od,
                                            const float* hub_lds = nullptr, uint32_t hub_rows = 0,
                                            const TwoPart& tp = TwoPart{nullptr, 0, 0}) {
    constexpr int kGroups = 64 / LPR;
    constexpr int kTile = LPR * VEC * VPL;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave = block * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t s0 = wave * kGroups;
    if (s0 >= n_strips) return;
    const int64_t s1 = (s0 + kGroups < n_strips) ? s0 + kGroups : n_strips;
    const int r_lo = strip_row[s0];
    const int r_hi = strip_row[s1];
    // A batch is read as whole 16-byte vectors starting at its first entry (4-byte aligned: the
    // hardware takes it), entries past the row's end included -- they are valid entries of later rows
    // and are never added.  Only a wave whose rows end within one batch of the end of the arrays
    // reads entry by entry with clamped addresses instead (wave-uniform choice).
    const bool whole_batches = __builtin_amdgcn_readfirstlane((int)(indptr[r_hi] + kUnroll <= nnz)) != 0;
    auto fetch_batch = [&](int64_t kb, int64_t k_last, uint32_t (&ci)[kUnroll], float (&av)[kUnroll],
                           DropIds<LPR, kUnroll>& idv) {
        if (whole_batches) {
#pragma unroll
            for (int q = 0; q < kUnroll / 4; ++q) {
                int4 iv;
                float4 fv;
                __builtin_memcpy(&iv, indices + kb + 4 * q, 16);
                __builtin_memcpy(&fv, vals + kb + 4 * q, 16);
                ci[4 * q + 0] = (uint32_t)iv.x;
                ci[4 * q + 1] = (uint32_t)iv.y;
                ci[4 * q + 2] = (uint32_t)iv.z;
                ci[4 * q + 3] = (uint32_t)iv.w;
                av[4 * q + 0] = fv.x;
                av[4 * q + 1] = fv.y;
                av[4 * q + 2] = fv.z;
                av[4 * q + 3] = fv.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int64_t ku = (kb + u < k_last) ? kb + u : k_last;
                ci[u] = (uint32_t)indices[ku];
                av[u] = vals[ku];
            }
        }
        if (DROPOUT) fetch_drop_ids<LPR, kUnroll>(dr, kb, k_last, sub, idv);
    };
    uint64_t leaders = 0;  // lane 0 of every group
#pragma unroll
    for (int g = 0; g < kGroups; ++g) leaders |= 1ull << (g * LPR);
    const uint64_t below = leaders & ((1ull << (grp * LPR)) - 1ull);  // leaders of the groups before mine
    for (int64_t col0 = 0; col0 < f; col0 += kTile) {
        int col[VPL];
        int lcol[VPL];
        bool on[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            col[v] = (int)col0 + (v * LPR + sub) * VEC;
            on[v] = col[v] < f;
            lcol[v] = on[v] ? col[v] : (int)f - VEC;
        }
        int r_next = r_lo;  // first unclaimed row (same in every lane)
        bool live = true;   // the range may still hold a row for this group
        // current row: r (-1: none), entries [k, k1), the batch at k in (cidx, a, id)
        int r = -1;
        int64_t k = 0, k1 = 0;
        uint32_t cidx[kUnroll];
        float a[kUnroll];
        DropIds<LPR, kUnroll> id;
        // claimed row: its first batch is on its way
        bool have_next = false;
        int rn_row = -1;
        int64_t nk = 0, nk1 = 0;
        uint32_t ncidx[kUnroll];
        float na[kUnroll];
        DropIds<LPR, kUnroll> nid;
        float acc[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[v][e] = 0.0f;
        // register window of row pointers: lane l holds indptr[w_base + l] and indptr[w_base + l + 1]
        int w_base = r_lo;
        int64_t ip_lo = indptr[(w_base + lane < r_hi) ? w_base + lane : r_hi];
        int64_t ip_hi = indptr[(w_base + lane + 1 < r_hi) ? w_base + lane + 1 : r_hi];
        for (;;) {
            // ---- claims: groups without a claimed row whose current row ends with this batch (or is none)
            if (r_next - w_base > 64 - kGroups) {  // wave-uniform: keep the rows claimable this turn inside the window
                w_base = r_next;
                ip_lo = indptr[(w_base + lane < r_hi) ? w_base + lane : r_hi];
                ip_hi = indptr[(w_base + lane + 1 < r_hi) ? w_base + lane + 1 : r_hi];
            }
            const bool want = live && !have_next && (k + kUnroll >= k1);
            const uint64_t m = __ballot(want) & leaders;
            const int rn = r_next + __popcll(m & below);
            r_next += __popcll(m);
            // every lane takes part in the shuffles; only claimers use what they return
            const int src = (want && rn < r_hi) ? rn - w_base : 0;
            const int64_t a0 = __shfl(ip_lo, src);
            const int64_t a1 = __shfl(ip_hi, src);
            if (want) {
                if (rn < r_hi) {
                    if (a1 - a0 <= split) {  // longer rows are summed chunk-wise: nothing to claim
                        have_next = true;
                        rn_row = rn;
                        nk = a0;
                        nk1 = a1;
                        if (a1 > a0) fetch_batch(a0, a1 - 1, ncidx, na, nid);
                    }
                } else {
                    live = false;
                }
            }
            // ---- one batch of the current row
            if (k < k1) {
                const int64_t k_last = k1 - 1;
                const int64_t rem = k1 - k;
                float bv[kUnroll][VPL][VEC];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    // a slot past the row's end holds an entry of a later row (batches are read whole): point
                    // it at this batch's first row of B, which is being fetched anyway, instead of gathering
                    // a row nobody adds -- the path behind the L2 is what a narrow product runs out of
                    const uint32_t cu = (u == 0 || u < rem) ? cidx[u] : cidx[0];
                    const float* brow = operand_row<SEG>(b, ldb, tp, cu);
                    if (HUBS && cu < hub_rows) {
                        const float* hrow = hub_lds + cu * (uint32_t)f;
#pragma unroll
                        for (int v = 0; v < VPL; ++v) load_vec<VEC>(hrow + lcol[v], bv[u][v]);
                    } else {
#pragma unroll
                        for (int v = 0; v < VPL; ++v) load_vec<VEC>(brow + lcol[v], bv[u][v]);
                    }
                }
                float a_now[kUnroll];
                bool kept[kUnroll];
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    a_now[u] = DROPOUT ? __fmul_rn(a[u], dr.inv_keep) : a[u];
                    kept[u] = true;
                }
                if (DROPOUT) drop_decisions<LPR, kUnroll>(dr, id, lane - sub, kept);
                const int64_t kn = k + kUnroll;
                if (kn < k1) fetch_batch(kn, k_last, cidx, a, id);  // the same row goes on: its next batch
#pragma unroll
                for (int u = 0; u < kUnroll; ++u) {
                    if (u < rem && kept[u]) {
#pragma unroll
                        for (int v = 0; v < VPL; ++v)
#pragma unroll
                            for (int e = 0; e < VEC; ++e)
                                acc[v][e] = __fadd_rn(acc[v][e], __fmul_rn(a_now[u], bv[u][v][e]));
                    }
                }
                k = kn;
            }
            // ---- row finished (or none): store it, move to the claimed one
            if (k >= k1) {
                if (r >= 0) {
                    store_row<VEC, VPL>(c + (int64_t)r * ldc, add ? add + (int64_t)r * ldadd : nullptr, col, on, acc,
                                        relu != 0, od, r);
#pragma unroll
                    for (int v = 0; v < VPL; ++v)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) acc[v][e] = 0.0f;
                }
                r = -1;
                k = 0;
                k1 = 0;
                if (have_next) {
                    r = rn_row;
                    k = nk;
                    k1 = nk1;
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        cidx[u] = ncidx[u];
                        a[u] = na[u];
                    }
                    if (DROPOUT) id = nid;
                    have_next = false;
                }
            }
            if (__ballot(live || r >= 0) == 0) break;
        }
    }
}

// one lane group per chunk of a long row -> partial row in the workspace
template <int LPR, int VEC, int VPL, bool DROPOUT, bool NT = false, bool SEG = false>
__device__ __forceinline__ void chunks_block(int64_t block, int64_t n_chunks, const int32_t* __restrict__ chunk_long,
                                             const int32_t* __restrict__ long_row,
                                             const int64_t* __restrict__ long_chunk,
                                             const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                             const float* __restrict__ vals, int64_t f, const float* __restrict__ b,
                                             uint32_t ldb, float* __restrict__ part, int64_t ldp, const DropArgs& dr,
                                             int chunk_len, uint32_t hot = 0, const TwoPart& tp = TwoPart{nullptr, 0, 0}) {
    constexpr int kGroups = 64 / LPR;
    constexpr int kTile = LPR * VEC * VPL;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const int grp = lane / LPR;
    const int64_t wave = block * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t chunk = wave * kGroups + grp;
    if (chunk >= n_chunks) return;
    const int li = chunk_long[chunk];
    const int r = long_row[li];
    const int64_t k0 = indptr[r] + (chunk - long_chunk[li]) * (int64_t)chunk_len;
    const int64_t row_end = indptr[r + 1];
    const int64_t k1 = (k0 + chunk_len < row_end) ? k0 + chunk_len : row_end;
    for (int64_t col0 = 0; col0 < f; col0 += kTile) {
        int col[VPL];
        int lcol[VPL];
        bool on[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            col[v] = (int)col0 + (v * LPR + sub) * VEC;
            on[v] = col[v] < f;
            lcol[v] = on[v] ? col[v] : (int)f - VEC;
        }
        float acc[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[v][e] = 0.0f;
        accumulate_range<LPR, VEC, VPL, DROPOUT, NT, SEG>(k0, k1, indices, vals, b, ldb, lcol, dr, acc, hot, tp);
        const OutDrop none = {0, 1.0f, 0, 0, nullptr, 0, 0};   // partial rows are stored as they are
        store_row<VEC, VPL>(part + chunk * ldp, nullptr, col, on, acc, false, none, 0);
    }
}

struct PlanView {
    int64_t n_strips, n_chunks, chunk_blocks, nnz;
    int split;
    int chunk_len;
    uint32_t hot_cols;
    uint32_t lds_hub_rows;
    const int32_t* strip_row;
    const int32_t* chunk_long;
    const int32_t* long_row;
    const int64_t* long_chunk;
};

// The whole product in one launch: the blocks that sum chunks of long rows come first (they are
// the longest work items), the strip blocks after them.
template <int LPR, int VEC, int VPL, bool DROPOUT, int U, bool NT = false, bool HUBS = false, bool SEG = false>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(PlanView pv, const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ vals, int64_t f,
                                                           const float* __restrict__ b, uint32_t ldb,
                                                           float* __restrict__ c, int64_t ldc,
                                                           float* __restrict__ part, int64_t ldp, int relu,
                                                           DropArgs dr, const float* __restrict__ add,
                                                           int64_t ldadd, OutDrop od, TwoPart tp) {
    if (DROPOUT && dr.stream_offset) dr.key = stream_key(dr.seed, dr.stream_id + *dr.stream_offset);
    if (od.keep_prob < 1.0f && od.stream_offset) od.key = stream_key(od.seed, od.stream_id + *od.stream_offset);
    const int64_t block = blockIdx.x;
    if (block < pv.chunk_blocks)
        chunks_block<LPR, VEC, VPL, DROPOUT, NT, SEG>(block, pv.n_chunks, pv.chunk_long, pv.long_row, pv.long_chunk,
                                                      indptr, indices, vals, f, b, ldb, part, ldp, dr, pv.chunk_len,
                                                      pv.hot_cols, tp);
    else if (LPR >= 32)
        strips_block<LPR, VEC, VPL, DROPOUT, (U > kUnroll), NT, SEG>(block - pv.chunk_blocks, pv.n_strips, pv.strip_row,
                                                                     indptr, indices, vals, f, b, ldb, c, ldc, relu, dr,
                                                                     pv.split, add, ldadd, od, pv.hot_cols, tp);
    else if (HUBS) {
        // stage rows 0 .. lds_hub_rows-1 of B: f is a multiple of 4 and the rows are 16-byte aligned (VEC == 4)
        extern __shared__ float4 hub_stage[];
        const uint32_t quads_per_row = (uint32_t)f / 4;
        const uint32_t quads = pv.lds_hub_rows * quads_per_row;
        for (uint32_t q = threadIdx.x; q < quads; q += kBlock) {
            const uint32_t row = q / quads_per_row, quad = q - row * quads_per_row;
            hub_stage[q] = *reinterpret_cast<const float4*>(b + (uint64_t)row * ldb + 4 * quad);
        }
        __syncthreads();
        slots_block<LPR, VEC, VPL, DROPOUT, U, true>(block - pv.chunk_blocks, pv.n_strips, pv.strip_row, indptr,
                                                     indices, vals, f, b, ldb, c, ldc, relu, dr, pv.split, add, ldadd,
                                                     pv.nnz, od, reinterpret_cast<const float*>(hub_stage),
                                                     pv.lds_hub_rows);
    } else
        slots_block<LPR, VEC, VPL, DROPOUT, U, false, SEG>(block - pv.chunk_blocks, pv.n_strips, pv.strip_row, indptr,
                                                           indices, vals, f, b, ldb, c, ldc, relu, dr, pv.split, add,
                                                           ldadd, pv.nnz, od, nullptr, 0, tp);
}

// partial rows of each long row, added in chunk order; one wave per long row.  Partial rows are
// 16-byte aligned with a leading dimension that is a multiple of 4, so they are always read as
// float4, eight chunks in flight; C is written as float4 only when `vec_c` says its layout allows it.
__global__ __launch_bounds__(kBlock) void spmm_long_reduce_kernel(int64_t n_long, const int32_t* __restrict__ long_row,
                                                                  const int64_t* __restrict__ long_chunk, int64_t f,
                                                                  const float* __restrict__ part, int64_t ldp,
                                                                  float* __restrict__ c, int64_t ldc, int relu,
                                                                  int vec_c, const float* __restrict__ add,
                                                                  int64_t ldadd, OutDrop od) {
    if (od.keep_prob < 1.0f && od.stream_offset) od.key = stream_key(od.seed, od.stream_id + *od.stream_offset);
    constexpr int kDepth = 8;
    const int lane = threadIdx.x & 63;
    const int64_t li = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (li >= n_long) return;
    const int64_t c0 = long_chunk[li], c1 = long_chunk[li + 1];
    float* dst = c + (int64_t)long_row[li] * ldc;
    const float* addrow = add ? add + (int64_t)long_row[li] * ldadd : nullptr;
    for (int64_t j = (int64_t)lane * 4; j < f; j += 256) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float* src = part + j;
        int64_t ch = c0;
        for (; ch + kDepth <= c1; ch += kDepth) {
            float x[kDepth][4];
#pragma unroll
            for (int u = 0; u < kDepth; ++u) load_vec<4>(src + (ch + u) * ldp, x[u]);
#pragma unroll
            for (int u = 0; u < kDepth; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], x[u][e]);
        }
        for (; ch < c1; ++ch) {
            float x[4];
            load_vec<4>(src + ch * ldp, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], x[e]);
        }
        if (addrow) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < f) s[e] = __fadd_rn(s[e], addrow[j + e]);
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.0f ? s[e] : 0.0f;
        }
        if (od.keep_prob < 1.0f) {
            const uint64_t id = (uint64_t)(od.index_offset + (int64_t)long_row[li] * od.pitch + j);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s[e] = keep_field(decision_hash(od.key, id + e), id + e, od.threshold) ? __fdiv_rn(s[e], od.keep_prob)
                                                                                       : 0.0f;
        }
        if (vec_c) {
            store_vec<4>(dst + j, s);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < f) dst[j + e] = s[e];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Packed dense operand: rows that are mostly zeros, stored without them.
//
// The wide product gathers one row of B per stored entry of A, and the number of 128-byte lines it pulls
// through the path behind the L2 is what it costs.  The operands a GCN training step gathers are sparse
// themselves: dropout(X) keeps a fraction keep_prob of the elements (gcn/layers.py:185), and the gradient that
// the backward product propagates is zero wherever the unit was dropped or inactive (ReLU and dropout of
// gcn/layers.py:185, 208).  A packed row of f <= 256 columns is
//     8 x {mask word w: bit c % 32 set iff column 32 w + c is kept; number of kept columns in the words before w}
//     then the kept values in column order
// at a row pitch of pack_pitch(f) floats (a multiple of 128 bytes), so a row with k kept values occupies
// 64 + 4 k bytes: 5 lines instead of 8 at keep_prob 0.5, 3 instead of 8 at a quarter kept.  Lane l of the wave that
// multiplies a row owns columns 4 l .. 4 l + 3: it reads the pair of its mask word (one 8-byte load, 64 bytes
// per wave), counts the kept columns before its own (prefix of the pair + popcount under a lane mask), reads
// 16 bytes of values from that position and hands them to its columns according to its four mask bits.  The
// sums are the dense kernel's: the same entries in the same order, and a skipped zero would have added a zero.
// ------------------------------------------------------------------------------------------------
constexpr int kPackHeader = 16;   // floats: the eight {mask, prefix} pairs
constexpr int kPackMaxF = 256;
constexpr int kPackUnroll = 4;    // entries in flight per wave: 47 registers, 8 waves per SIMD (6 or 8 entries: slower)

static inline int64_t pack_pitch(int64_t f) { return (kPackHeader + f + 31) / 32 * 32; }

// One wave per row.  DROP: the row is dropout(x) -- x / keep_prob where the decision of element
// index_offset + row * pitch + col says keep, as dropout_kernel computes it -- and what is packed are its
// nonzeros; otherwise the nonzeros of x itself.
// GATED: the row is where(x > 0, (a . w^T) / divisor, 0) for a [n_rows x k] and w [f x k], k <= kGatedMaxK -- the
// gradient with respect to a hidden layer's pre-activation, (dP . W^T) gated by the dropped activations x
// (gcn_gemm_gated_f32 with trans_b) -- computed row by row with the lane's four rows of w in registers and the
// row of a in scalar registers, and packed without the dense matrix ever being written.
constexpr int kGatedMaxK = 16;
struct GatedSource {
    const float* a;
    int64_t lda;
    const float* w;
    int64_t ldw;
    int k;
    float divisor;
    float inverse;   // 1 / divisor when the divisor is a power of two (the product with it is then the quotient), else 0
    // may be null: the rows to form (ascending).  Item i of the launch then works on row rows[i] of a, of the gate and
    // of the packed output (n_rows of the launch counts items); a row number outside [0, m) is skipped.
    const int64_t* rows;
    int64_t m;
};

enum { kPackPlain = 0, kPackDrop = 1, kPackGated = 2 };

template <int MODE>
__global__ __launch_bounds__(kBlock) void pack_rows_kernel(int64_t n_rows, int f, const float* __restrict__ x,
                                                          int64_t ldx, int x_vec, float* __restrict__ packed,
                                                          int64_t ldpk, OutDrop od, GatedSource gs) {
    constexpr bool DROP = MODE == kPackDrop;
    if (DROP && od.stream_offset) od.key = stream_key(od.seed, od.stream_id + *od.stream_offset);
    const int lane = threadIdx.x & 63;
    const int w = lane >> 3, sh = 4 * (lane & 7), col = 4 * lane;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // rows as scalars
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + wave_in_block;
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    __shared__ __attribute__((aligned(16))) float stage_all[(kBlock / 64) * 256];
    float* stage = stage_all + wave_in_block * 256;
    float wr[4][MODE == kPackGated ? kGatedMaxK : 1];
    if (MODE == kPackGated) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kk = 0; kk < kGatedMaxK; ++kk)
                wr[e][kk] = gs.w[(int64_t)min(col + e, f - 1) * gs.ldw + min(kk, gs.k - 1)] *
                            ((col + e < f && kk < gs.k) ? 1.0f : 0.0f);      // no branch between the 64 loads
    }
    // a row's elements of x (and, GATED, of a) are requested one row ahead of their use
    auto fetch = [&](int64_t item, float (&v)[4], float& mine, int64_t& drawn) {
        // DROP: the row of the matrix the dropout decisions are drawn for; GATED with a row list: the row this
        // item works on (a scalar load; the row's own loads wait for it)
        drawn = (DROP && od.row_ids) ? od.row_ids[item] : item;
        int64_t row = item;
        if (MODE == kPackGated && gs.rows) {
            drawn = gs.rows[item];
            row = (drawn >= 0 && drawn < gs.m) ? drawn : 0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.0f;
        const float* src = x + row * ldx + col;
        if (x_vec && col + 3 < f) {
            load_vec<4>(src, v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < f) v[e] = src[e];
        }
        if (MODE == kPackGated) mine = gs.a[row * gs.lda + min(lane, gs.k - 1)];
    };
    float v_next[4], mine_next = 0.0f;
    int64_t drawn_next = 0;
    if (wave < n_rows) fetch(wave, v_next, mine_next, drawn_next);
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v_next[e];
        const float mine = mine_next;
        const int64_t drawn = drawn_next;
        fetch(min(row + n_waves, n_rows - 1), v_next, mine_next, drawn_next);
        int64_t out_row = row;
        if (MODE == kPackGated && gs.rows) {
            if (drawn < 0 || drawn >= gs.m) continue;          // not a row of the matrix: nothing is written
            out_row = drawn;
        }
        if (MODE == kPackGated) {
            // The row of a (at most 64 bytes, new for every row) is read by one vector load, element kk by lane kk,
            // and handed to all lanes through scalar registers.  (As 16 scalar loads it cost 16 misses of the scalar
            // cache on the same line, one after the other: 28.7 ms for 16.7 M rows.)  Positions past k repeat the
            // last element and meet zeros in wr.
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kk = 0; kk < kGatedMaxK; ++kk) {
                const float ak = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mine), kk));
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __fmaf_rn(ak, wr[e][kk], acc[e]);
            }
            // a power of two divides exactly as a product with its inverse (keep_prob 0.5: one multiply instead
            // of the eleven instructions of an IEEE division)
            if (gs.inverse != 0.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (v[e] > 0.0f) ? __fmul_rn(acc[e], gs.inverse) : 0.0f;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float q = __fdiv_rn(acc[e], gs.divisor);
                    v[e] = (v[e] > 0.0f) ? q : 0.0f;
                }
            }
        }
        if (DROP) {
            const uint64_t id = (uint64_t)(od.index_offset + drawn * od.pitch + col);
            const bool one_hash = (id & 3) == 0;
            const uint64_t h4 = one_hash ? decision_hash(od.key, id) : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint64_t h = one_hash ? h4 : decision_hash(od.key, id + e);
                const float scaled = __fdiv_rn(v[e], od.keep_prob);
                v[e] = (col + e < f && keep_field(h, id + e, od.threshold)) ? scaled : 0.0f;
            }
        }
        uint32_t nib = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) nib |= (v[e] != 0.0f ? 1u : 0u) << e;
        uint32_t word = nib << sh;                        // OR over the 8 lanes that share a mask word
        word |= (uint32_t)__shfl_xor((int)word, 1);
        word |= (uint32_t)__shfl_xor((int)word, 2);
        word |= (uint32_t)__shfl_xor((int)word, 4);
        const uint32_t count = (uint32_t)__popc(word);
        uint32_t before = 0;                              // kept columns in the words before this lane's
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const uint32_t t = (uint32_t)__shfl((int)count, 8 * q);
            if (q < w) before += t;
        }
        float* dst = packed + out_row * ldpk;
        if ((lane & 7) == 0) reinterpret_cast<uint2*>(dst)[w] = make_uint2(word, before);
        // The kept values go to their packed positions in 1 KiB of LDS private to the wave, and leave it as whole
        // 16-byte vectors, lane l the values 4 l .. 4 l + 3: contiguous stores instead of up to four scattered
        // 4-byte stores per lane.
        float* out = stage + before + __popc(word & ((1u << sh) - 1u));
        int at = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if ((nib >> e) & 1u) out[at++] = v[e];
        const uint32_t total = (uint32_t)__shfl((int)(before + count), 63);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if ((uint32_t)(4 * lane) < total) {
            float o[4];
            load_vec<4>(stage + 4 * lane, o);
            store_vec<4>(dst + kPackHeader + 4 * lane, o);     // what lies past the last kept value is never used
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// every row of a packed matrix made empty: 64 zero bytes per row (its values are then never read)
__global__ __launch_bounds__(kBlock) void clear_packed_rows_kernel(int64_t n_rows, float* __restrict__ packed, int64_t ldpk) {
    // four lanes per row, 16 bytes each
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < 4 * n_rows; i += stride) {
        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        store_vec<4>(packed + (i >> 2) * ldpk + 4 * (i & 3), zero);
    }
}

// dense row-major copy of a packed matrix (tests and callers that need the expanded form)
__global__ __launch_bounds__(kBlock) void unpack_rows_kernel(int64_t n_rows, int f, const float* __restrict__ packed,
                                                            int64_t ldpk, float* __restrict__ y, int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int w = lane >> 3, sh = 4 * (lane & 7), col = 4 * lane;
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
    for (int64_t row = wave; row < n_rows; row += n_waves) {
        const float* src = packed + row * ldpk;
        const uint2 h = reinterpret_cast<const uint2*>(src)[w];
        const float* in = src + kPackHeader + h.y + __popc(h.x & ((1u << sh) - 1u));
        const uint32_t nib = (h.x >> sh) & 15u;
        int at = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = ((nib >> e) & 1u) ? in[at++] : 0.0f;
            if (col + e < f) y[row * ldy + col + e] = v;
        }
    }
}

// acc += sum over entries k0..k1-1 (in order) of a_k * (row col_k of the packed operand, this lane's four columns).
// Per entry a lane reads the pair of its mask word (8 bytes; 64 bytes per wave), counts the kept columns before its
// own, reads 16 bytes of values from that (4-byte aligned) position and hands them to its columns by its four mask
// bits: column e takes value number (kept columns among 0..e-1) if its bit is set, zero otherwise -- selects by bit
// masks, no branch (a branch per entry would also let the compiler sink each load next to its use: one exposed
// latency per entry).  Three stages in flight: the (index, value) pairs two batches ahead, the mask pairs of the next
// batch, and the values of this one.  Slots past the end of the range (clamped repeats of the last entry) get an
// empty mask and add zeros.
// Measured on the scale-24 graph, f = 256: 46.6 ms at half the columns kept and 35.2 ms at a quarter, against 74 ms for
// the dense operand; a packed row costs 5 and 3 lines of 128 bytes there, which bounds the product at 46 and 28 ms.
// The steps that got it there from 61 / 57 ms, where the vector ALU was 80 % busy (53 instructions per entry) and the
// address unit 87 %: the hand-out as a queue with one-instruction bit selects (26 instructions per entry), 32-bit loop
// control, the header read by 16 lanes instead of 64 (41 -> 24 cache look-ups per entry), then the scalar unit (one
// per CU): whole batches of entries as multi-dword scalar loads, the row base computed once per entry (50.6 / 42.6 ->
// 46.9 / 36.3), and the rows of a strip walked as one stream (stream_packed below).
// Aligned reads by the first kept/4 lanes straight into LDS and a read-back per column chain two more latencies per
// batch and are bound by the LDS-DMA rate: 64-85 ms.
// (mask & yes) | (~mask & no) in one instruction; written as an expression the compiler turns it into a compare
// and a conditional move
__device__ __forceinline__ uint32_t bit_select(uint32_t mask, uint32_t yes, uint32_t no) {
    uint32_t out;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(out) : "v"(mask), "v"(yes), "v"(no));
    return out;
}

// acc += a * (the lane's four columns of a packed row): `values` are the 16 bytes read at the lane's position in the
// row's kept values, bits 0..3 of `nib` say which of the lane's columns are kept (higher bits are ignored).  The
// values are a queue; column e takes its head if the column's bit is set (zero otherwise), and the queue then moves
// up by one: 4 masks, 4 ANDs, 6 bit selects.
__device__ __forceinline__ void add_packed_entry(uint32_t nib, const float4& values, float a, float (&acc)[4]) {
    const int32_t m = (int32_t)nib;
    const uint32_t m0 = (uint32_t)((m << 31) >> 31), m1 = (uint32_t)((m << 30) >> 31),
                   m2 = (uint32_t)((m << 29) >> 31), m3 = (uint32_t)((m << 28) >> 31);
    uint32_t q0 = __float_as_uint(values.x), q1 = __float_as_uint(values.y), q2 = __float_as_uint(values.z);
    const uint32_t q3 = __float_as_uint(values.w);
    const float x0 = __uint_as_float(q0 & m0);
    q0 = bit_select(m0, q1, q0);
    q1 = bit_select(m0, q2, q1);
    q2 = bit_select(m0, q3, q2);
    const float x1 = __uint_as_float(q0 & m1);
    q0 = bit_select(m1, q1, q0);
    q1 = bit_select(m1, q2, q1);
    const float x2 = __uint_as_float(q0 & m2);
    q0 = bit_select(m2, q1, q0);
    const float x3 = __uint_as_float(q0 & m3);
    acc[0] = __fadd_rn(acc[0], __fmul_rn(a, x0));
    acc[1] = __fadd_rn(acc[1], __fmul_rn(a, x1));
    acc[2] = __fadd_rn(acc[2], __fmul_rn(a, x2));
    acc[3] = __fadd_rn(acc[3], __fmul_rn(a, x3));
}

template <int U>
__device__ __forceinline__ void accumulate_packed(int64_t k0, int64_t k1, const int32_t* __restrict__ indices,
                                                  const float* __restrict__ vals, const float* __restrict__ pk,
                                                  uint32_t ldpk, int w, int sh, float (&acc)[4]) {
    if (k0 >= k1) return;
    // a range is a row or a chunk of at most the split threshold: 32-bit offsets from k0, so that the loop control
    // and the clamps are scalar compares (there is no ordered 64-bit compare on the scalar unit)
    const int n = (int)(k1 - k0);
    const int32_t* idx = indices + k0;
    const float* val = vals + k0;
    const uint32_t below = (1u << sh) - 1u;
    const uint32_t pitch_bytes = ldpk * 4u;
    const char* pk_bytes = reinterpret_cast<const char*>(pk);
    uint32_t c0[U], c1[U];
    float a0[U], a1[U];
    const char* base0[U];   // row bases of the batch whose headers are in h: computed once, used for header and values
    uint32_t h[U];          // lane l < 16: dword l of the row's sixteen header dwords
    const int lane4 = 4 * (int)(threadIdx.x & 63);
    // The scalar unit of a CU serves all four SIMDs, one instruction a cycle: what a wave spends there per entry
    // counts like vector work.  A batch that lies wholly inside the range comes as one multi-dword scalar load per
    // array (no clamp, no address per entry); the last, partial batch clamps to the last entry; a batch past the end
    // loads nothing and keeps the previous (valid, masked-out) entries.
    auto entries = [&](int jb, uint32_t (&c)[U], float (&a)[U]) {
        if (jb + U <= n) {
            __builtin_memcpy(&c[0], idx + jb, 4 * U);
            __builtin_memcpy(&a[0], val + jb, 4 * U);
        } else if (jb < n) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ju = min(jb + u, n - 1);
                c[u] = (uint32_t)idx[ju];
                a[u] = val[ju];
            }
        }
    };
    // The 64 header bytes of a row are read once, a dword by each of the first 16 lanes: a buffer of 64 bytes, in
    // which the other lanes' offsets are out of range and cost no access.  (Every lane reading its own pair --
    // eight distinct pairs -- kept all 16 lane quads of the load busy: 16 of the ~41 cache look-ups per entry.)
    auto headers = [&](const uint32_t (&c)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            base0[u] = pk_bytes + (uint64_t)c[u] * (uint64_t)pitch_bytes;
            const __amdgpu_buffer_rsrc_t res =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0[u]), 0, 4 * kPackHeader, 0x00020000);
            h[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(res, lane4, 0, 0);
        }
    };
    // One batch: positions from the headers, the value loads, then (while those are in flight) the next batch's
    // headers and the entries two batches on, then the hand-out.  The last batch of a range may be partial -- its
    // slots past the end repeat the last entry under an empty mask -- and has nothing to request ahead.
    auto batch = [&](int j, auto is_last) {
        constexpr bool kLast = decltype(is_last)::value;
        float4 v[U];
        uint32_t nib[U];
        float a_now[U];
        const char* from[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // this lane's pair comes from lanes 2 w and 2 w + 1 through the cross-lane path of the LDS unit
            const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(8 * w, (int)h[u]);
            const uint32_t before = (uint32_t)__builtin_amdgcn_ds_bpermute(8 * w + 4, (int)h[u]);
            const uint32_t at = before + (uint32_t)__popc(word & below);
            // bits 0..3 are this lane's columns (the bits above them are never looked at)
            nib[u] = kLast ? ((word >> sh) & ((j + u < n) ? 15u : 0u)) : (word >> sh);
            from[u] = base0[u] + (uint32_t)((at + kPackHeader) * 4u);
            a_now[u] = a0[u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_memcpy(&v[u], from[u], 16);   // 4-byte aligned: the hardware takes it
        __builtin_amdgcn_sched_barrier(0);
        if (!kLast) {
            headers(c1);
#pragma unroll
            for (int u = 0; u < U; ++u) a0[u] = a1[u];
            entries(j + 2 * U, c1, a1);
        }
        // everything above is in flight before anything below waits for it
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            add_packed_entry(nib[u], v[u], a_now[u], acc);
        }
    };
    entries(0, c0, a0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        c1[u] = c0[u];
        a1[u] = a0[u];
    }
    entries(U, c1, a1);
    headers(c0);
    int j = 0;
    for (; n - j > U; j += U) batch(j, std::false_type{});
    batch(j, std::true_type{});
}

// Rows r .. r_hi - 1 of a strip as ONE stream of entries: their entries are contiguous in the CSR arrays, so the
// three stages above keep running across row boundaries instead of draining and refilling once per row (a row of
// this graph has 32 entries on average = 8 batches, and a refill is three dependent memory latencies).  A batch that
// lies inside the current row adds its entries without a test; a batch in which a row ends takes the copy of the
// code that tests before every entry, stores the finished row and moves on.  The stream ends after row r_hi - 1 or
// in front of a row longer than the split threshold (those are summed chunk-wise); the row it stopped at is
// returned.  Row r itself must not be longer than the threshold.  Same entries, same order of additions per row as
// accumulate_packed.
template <int U, typename Store>
__device__ __forceinline__ int stream_packed(int r, const int r_hi, const int64_t* __restrict__ indptr,
                                             const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                             const float* __restrict__ pk, uint32_t ldpk, int split, int w, int sh,
                                             Store&& store) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t k0 = indptr[r];
    int64_t end_abs = indptr[r + 1];                      // end of the current row
    int64_t next_abs = indptr[min(r + 2, r_hi)];          // end of the row after it, requested a row early
    // entries the requests may touch: up to the end of the strip's rows (a long row among them only bounds the
    // look-ahead further out; what is added is decided by the row ends below)
    const int n = (int)min(indptr[r_hi] - k0, (int64_t)(1 << 30));
    if (n == 0) {                                          // nothing but empty rows
        for (; r < r_hi; ++r) store(r, acc);
        return r_hi;
    }
    int row_end = (int)(end_abs - k0);
    const int32_t* idx = indices + k0;
    const float* val = vals + k0;
    const uint32_t below = (1u << sh) - 1u;
    const uint32_t pitch_bytes = ldpk * 4u;
    const char* pk_bytes = reinterpret_cast<const char*>(pk);
    uint32_t c0[U], c1[U];
    float a0[U], a1[U];
    const char* base0[U];
    uint32_t h[U];
    const int lane4 = 4 * (int)(threadIdx.x & 63);
    auto entries = [&](int jb, uint32_t (&c)[U], float (&a)[U]) {
        if (jb + U <= n) {
            __builtin_memcpy(&c[0], idx + jb, 4 * U);
            __builtin_memcpy(&a[0], val + jb, 4 * U);
        } else if (jb < n) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ju = min(jb + u, n - 1);
                c[u] = (uint32_t)idx[ju];
                a[u] = val[ju];
            }
        }
    };
    auto headers = [&](const uint32_t (&c)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            base0[u] = pk_bytes + (uint64_t)c[u] * (uint64_t)pitch_bytes;
            const __amdgpu_buffer_rsrc_t res =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0[u]), 0, 4 * kPackHeader, 0x00020000);
            h[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(res, lane4, 0, 0);
        }
    };
    // the current row is complete: store it and make the next one current; true when the stream ends here
    auto next_row = [&]() -> bool {
        store(r, acc);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = 0.0f;
        ++r;
        if (r == r_hi) return true;
        const int64_t len = next_abs - end_abs;
        if (len > (int64_t)split) return true;
        row_end += (int)len;
        end_abs = next_abs;
        next_abs = indptr[min(r + 2, r_hi)];
        return false;
    };
    entries(0, c0, a0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        c1[u] = c0[u];
        a1[u] = a0[u];
    }
    entries(U, c1, a1);
    headers(c0);
    for (int j = 0;; j += U) {
        float4 v[U];
        uint32_t nib[U];
        float a_now[U];
        const char* from[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t word = (uint32_t)__builtin_amdgcn_ds_bpermute(8 * w, (int)h[u]);
            const uint32_t before = (uint32_t)__builtin_amdgcn_ds_bpermute(8 * w + 4, (int)h[u]);
            const uint32_t at = before + (uint32_t)__popc(word & below);
            nib[u] = word >> sh;
            from[u] = base0[u] + (uint32_t)((at + kPackHeader) * 4u);
            a_now[u] = a0[u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_memcpy(&v[u], from[u], 16);
        __builtin_amdgcn_sched_barrier(0);
        if (j + U < n) {                                   // there is a next batch
            headers(c1);
#pragma unroll
            for (int u = 0; u < U; ++u) a0[u] = a1[u];
            entries(j + 2 * U, c1, a1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + U <= row_end) {
#pragma unroll
            for (int u = 0; u < U; ++u) add_packed_entry(nib[u], v[u], a_now[u], acc);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                while (j + u == row_end)                   // `while`: rows without entries end where they start
                    if (next_row()) return r;
                add_packed_entry(nib[u], v[u], a_now[u], acc);
            }
        }
    }
}

// The product with a packed operand, one launch: chunk blocks first, strip blocks after them, a wave per row
// or chunk (f <= 256: four columns per lane), the split threshold of the wide dense product.
template <int U>
__global__ __launch_bounds__(kBlock, 8) void spmm_packed_kernel(PlanView pv, const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, int f,
                                                            const float* __restrict__ pk, uint32_t ldpk,
                                                            float* __restrict__ c, int64_t ldc, int c_vec,
                                                            float* __restrict__ part, int64_t ldp, int relu) {
    const int lane = threadIdx.x & 63;
    const int w = lane >> 3, sh = 4 * (lane & 7), col = 4 * lane;
    const int64_t block = blockIdx.x;
    // the wave's number as a scalar: its rows, their entry ranges and the entries themselves (the same in every
    // lane) then live in scalar registers and come through scalar loads
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto store = [&](float* dst, bool vec, const float (&acc)[4], bool act) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (act && !(acc[e] > 0.0f)) ? 0.0f : acc[e];
        if (vec && col + 3 < f) {
            store_vec<4>(dst + col, o);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < f) dst[col + e] = o[e];
        }
    };
    if (block < pv.chunk_blocks) {
        const int64_t chunk = block * (kBlock / 64) + wave_in_block;
        if (chunk >= pv.n_chunks) return;
        const int li = pv.chunk_long[chunk];
        const int r = pv.long_row[li];
        const int64_t k0 = indptr[r] + (chunk - pv.long_chunk[li]) * (int64_t)pv.chunk_len;
        const int64_t row_end = indptr[r + 1];
        const int64_t k1 = (k0 + pv.chunk_len < row_end) ? k0 + pv.chunk_len : row_end;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        accumulate_packed<U>(k0, k1, indices, vals, pk, ldpk, w, sh, acc);
        store(part + chunk * ldp, true, acc, false);      // partial rows: 16-byte aligned, pitch a multiple of 4
        return;
    }
    const int64_t strip = (block - pv.chunk_blocks) * (kBlock / 64) + wave_in_block;
    if (strip >= pv.n_strips) return;
    const int r_lo = pv.strip_row[strip], r_hi = pv.strip_row[strip + 1];
    int r = r_lo;
    while (r < r_hi) {
        if (indptr[r + 1] - indptr[r] > pv.split) {       // summed chunk-wise
            ++r;
            continue;
        }
        r = stream_packed<U>(r, r_hi, indptr, indices, vals, pk, ldpk, pv.split, w, sh,
                             [&](int row, const float (&acc)[4]) {
                                 store(c + (int64_t)row * ldc, c_vec != 0, acc, relu != 0);
                             });
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
struct SpmmCall {
    const gcn_spmm_plan* plan;
    const int64_t* indptr;
    const int32_t* indices;
    const float* vals;
    int64_t f;
    const float* b;
    uint32_t ldb;
    float* c;
    int64_t ldc;
    int relu;
    int wide_split;
    float* part;
    int64_t ldp;
    DropArgs dr;
    const float* add;
    int64_t ldadd;
    OutDrop od;
    TwoPart tp;       // n_first == 0: the operand is one matrix
    hipStream_t st;
};

template <int LPR, int VEC, int VPL, bool DROPOUT, int U = kUnroll>
static int launch_variant(const SpmmCall& a) {
    const gcn_spmm_plan* p = a.plan;
    constexpr int kWaves = kBlock / 64;
    constexpr int kGroups = 64 / LPR;
    const int ti = (LPR >= 32 || a.wide_split) ? 1 : 0;
    const gcn_spmm_plan::LongTable& lt = p->table[ti];
    PlanView pv;
    pv.n_strips = p->n_strips;
    pv.nnz = p->nnz;
    pv.n_chunks = lt.n_chunks;
    pv.chunk_blocks = ceil_div(ceil_div(lt.n_chunks, kGroups), kWaves);
    pv.split = p->split[ti];
    pv.chunk_len = p->chunk_len ? p->chunk_len : pv.split;
    pv.strip_row = p->strip_row;
    pv.chunk_long = lt.chunk_long;
    pv.long_row = lt.long_row;
    pv.long_chunk = lt.long_chunk;
    const int64_t strip_waves = (LPR >= 32) ? p->n_strips : ceil_div(p->n_strips, kGroups);
    const int64_t grid = pv.chunk_blocks + ceil_div(strip_waves, kWaves);
    if (grid > INT32_MAX) return fail(GCN_ERR_BAD_ARG, "matrix too large for one launch");
    pv.hot_cols = 0;
    if (p->timing) GCN_HIP_TRY(hipEventRecord(p->ev[0], a.st));
    bool launched = false;
    if constexpr (LPR == 64 && !DROPOUT && U == kUnroll) {
        if (p->hot_cols > 0 && p->hot_cols < p->n_cols && a.tp.n_first == 0) {
            pv.hot_cols = (uint32_t)p->hot_cols;
            spmm_rows_kernel<LPR, VEC, VPL, DROPOUT, U, true><<<(unsigned)grid, kBlock, 0, a.st>>>(
                pv, a.indptr, a.indices, a.vals, a.f, a.b, a.ldb, a.c, a.ldc, a.part, a.ldp, a.relu, a.dr, a.add,
                a.ldadd, a.od, a.tp);
            launched = true;
        }
    }
    pv.lds_hub_rows = 0;
    if constexpr (LPR <= 16 && VEC == 4 && VPL == 1 && !DROPOUT && U == kUnroll) {
        // rows of B that fit kHubLdsBytes of LDS per workgroup, and exist, and are covered by one pass
        const int64_t fit = kHubLdsBytes / (a.f * (int64_t)sizeof(float));
        const int64_t hubs = std::min<int64_t>(std::min<int64_t>(p->lds_hub_rows, fit), p->n_cols);
        // an empty matrix reads no row of B (b may be null): nothing to stage
        if (hubs > 0 && p->nnz > 0 && a.b != nullptr && a.tp.n_first == 0 && a.f % 4 == 0 && a.f <= LPR * VEC) {
            pv.lds_hub_rows = (uint32_t)hubs;
            const size_t lds_bytes = (size_t)hubs * (size_t)a.f * sizeof(float);
            spmm_rows_kernel<LPR, VEC, VPL, DROPOUT, U, false, true><<<(unsigned)grid, kBlock, lds_bytes, a.st>>>(
                pv, a.indptr, a.indices, a.vals, a.f, a.b, a.ldb, a.c, a.ldc, a.part, a.ldp, a.relu, a.dr, a.add,
                a.ldadd, a.od, a.tp);
            launched = true;
        }
    }
    if constexpr (VEC == 4 && !DROPOUT && U == kUnroll) {
        if (!launched && a.tp.n_first > 0) {     // two-part operand (wide and narrow instances of large matrices)
            spmm_rows_kernel<LPR, VEC, VPL, DROPOUT, U, false, false, true><<<(unsigned)grid, kBlock, 0, a.st>>>(
                pv, a.indptr, a.indices, a.vals, a.f, a.b, a.ldb, a.c, a.ldc, a.part, a.ldp, a.relu, a.dr, a.add,
                a.ldadd, a.od, a.tp);
            launched = true;
        }
    }
    if (!launched && a.tp.n_first > 0)
        return fail(GCN_ERR_BAD_ARG, "this product has no two-part form (it needs 16-byte aligned rows, a width that is "
                                     "a multiple of 4 and a matrix of at least 2M entries): pass the operand as one matrix");
    if (!launched)
        spmm_rows_kernel<LPR, VEC, VPL, DROPOUT, U><<<(unsigned)grid, kBlock, 0, a.st>>>(
            pv, a.indptr, a.indices, a.vals, a.f, a.b, a.ldb, a.c, a.ldc, a.part, a.ldp, a.relu, a.dr, a.add,
            a.ldadd, a.od, a.tp);
    GCN_LAUNCH_CHECK();
    if (p->timing) GCN_HIP_TRY(hipEventRecord(p->ev[1], a.st));
    if (lt.n_chunks > 0) {
        const unsigned grid2 = (unsigned)ceil_div(lt.n_long, kWaves);
        spmm_long_reduce_kernel<<<grid2, kBlock, 0, a.st>>>(lt.n_long, lt.long_row, lt.long_chunk, a.f, a.part,
                                                            a.ldp, a.c, a.ldc, a.relu, VEC == 4 ? 1 : 0, a.add,
                                                            a.ldadd, a.od);
        GCN_LAUNCH_CHECK();
    }
    if (p->timing) GCN_HIP_TRY(hipEventRecord(p->ev[2], a.st));
    return GCN_OK;
}

template <int VEC, bool DROPOUT>
static int launch_by_width(const SpmmCall& a) {
    const int64_t nvec = ceil_div(a.f, VEC);
    if (a.plan->strip_nnz < kStripNnz && nvec <= 16) {  // small matrix, narrow operand
        if (nvec <= 1) return launch_variant<1, VEC, 1, DROPOUT, kUnrollSmall>(a);
        if (nvec <= 2) return launch_variant<2, VEC, 1, DROPOUT, kUnrollSmall>(a);
        if (nvec <= 4) return launch_variant<4, VEC, 1, DROPOUT, kUnrollSmall>(a);
        if (nvec <= 8) return launch_variant<8, VEC, 1, DROPOUT, kUnrollSmall>(a);
        return launch_variant<16, VEC, 1, DROPOUT, kUnrollSmall>(a);
    }
    if (nvec <= 1) return launch_variant<1, VEC, 1, DROPOUT>(a);
    if (nvec <= 2) return launch_variant<2, VEC, 1, DROPOUT>(a);
    if (nvec <= 4) return launch_variant<4, VEC, 1, DROPOUT>(a);
    if (nvec <= 8) return launch_variant<8, VEC, 1, DROPOUT>(a);
    if (nvec <= 16) return launch_variant<16, VEC, 1, DROPOUT>(a);
    if (a.plan->strip_nnz < kStripNnz) {  // small matrix, wide operand: two batches of B rows in flight per row
        if (nvec <= 32) return launch_variant<32, VEC, 1, DROPOUT, kUnrollSmall>(a);
        if (nvec <= 64) return launch_variant<64, VEC, 1, DROPOUT, kUnrollSmall>(a);
        return launch_variant<64, VEC, 2, DROPOUT, kUnrollSmall>(a);
    }
    if (nvec <= 32) return launch_variant<32, VEC, 1, DROPOUT>(a);
    if (nvec <= 64) return launch_variant<64, VEC, 1, DROPOUT>(a);
    return launch_variant<64, VEC, 2, DROPOUT>(a);  // wider rows: passes of 128 vectors
}

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static inline int64_t partial_ld(int64_t f) { return (f + 3) / 4 * 4; }

template <bool DROPOUT>
static int spmm_entry(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                      int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc, int flags, void* workspace,
                      size_t workspace_bytes, DropArgs dr, void* stream, const float* add = nullptr,
                      int64_t ldadd = 0, const OutDrop* od = nullptr, const TwoPart* first_part = nullptr) {
    if (!plan) return fail(GCN_ERR_BAD_ARG, "plan is null");
    if (add && ldadd < f) return fail(GCN_ERR_BAD_ARG, "bad addend leading dimension %lld", (long long)ldadd);
    if (f < 0 || ldb < f || ldc < f || ldb > INT32_MAX || f > INT32_MAX - 1024)
        return fail(GCN_ERR_BAD_ARG, "bad width: f=%lld ldb=%lld ldc=%lld", (long long)f, (long long)ldb, (long long)ldc);
    if (!indptr || (plan->nnz > 0 && (!indices || !vals))) return fail(GCN_ERR_BAD_ARG, "null CSR pointer");
    if (plan->n_rows == 0 || f == 0) return GCN_OK;
    if (!c || (plan->nnz > 0 && !b)) return fail(GCN_ERR_BAD_ARG, "null dense pointer");
    if (flags & ~(GCN_RELU | GCN_SPLIT_WIDE)) return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    SpmmCall a;
    a.plan = plan;
    a.indptr = indptr;
    a.indices = indices;
    a.vals = vals;
    a.f = f;
    a.b = b;
    a.ldb = (uint32_t)ldb;
    a.c = c;
    a.ldc = ldc;
    a.relu = (flags & GCN_RELU) ? 1 : 0;
    a.wide_split = (flags & GCN_SPLIT_WIDE) ? 1 : 0;
    a.part = nullptr;
    a.ldp = partial_ld(f);
    a.dr = dr;
    a.add = add;
    a.ldadd = ldadd;
    a.od = od ? out_drop_ready(*od) : OutDrop{0, 1.0f, 0, 0, nullptr, 0, 0};
    a.st = (hipStream_t)stream;
    a.tp = TwoPart{nullptr, 0, 0};
    if (first_part && first_part->n_first > 0) {
        // rows [0, n_first) at first_part->first; the caller's b holds the rows from n_first on: move it back so
        // that both parts are addressed by the operand's own row numbers
        a.tp = *first_part;
        a.b = (const float*)((uintptr_t)b - (uintptr_t)first_part->n_first * (uintptr_t)ldb * sizeof(float));
    }
    const int64_t max_chunks = std::max(plan->table[0].n_chunks, plan->table[1].n_chunks);
    if (max_chunks > 0) {
        const size_t need = (size_t)max_chunks * (size_t)a.ldp * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        if (!aligned16(workspace)) return fail(GCN_ERR_BAD_ARG, "workspace must be 16-byte aligned");
        a.part = (float*)workspace;
    }
    const bool vec4 = (f % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(b) && aligned16(c) &&
                      (!add || ((ldadd % 4 == 0) && aligned16(add))) &&
                      (a.tp.n_first == 0 || ((a.tp.ld_first % 4 == 0) && aligned16(a.tp.first)));
    return vec4 ? launch_by_width<4, DROPOUT>(a) : launch_by_width<1, DROPOUT>(a);
}

// Host half of plan creation: from the rows longer than the smaller threshold (any order, as the collecting
// kernel's atomics left them) the table of the rows longer than `split` in ascending row order, each cut in
// chunks of chunk_len entries.  long_chunk has one more entry than long_row (the total closes it).
void build_long_table(const std::vector<int32_t>& rows, const std::vector<int64_t>& starts,
                      const std::vector<int64_t>& ends, int64_t split, int64_t chunk_len, LongTableHost* out) {
    const size_t count = rows.size();
    std::vector<int64_t> order(count);
    for (size_t i = 0; i < count; ++i) order[i] = (int64_t)i;
    std::sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return rows[x] < rows[y]; });
    out->long_row.clear();
    out->long_chunk.clear();
    out->chunk_long.clear();
    int64_t total = 0;
    for (size_t i = 0; i < count; ++i) {
        const int64_t o = order[i];
        const int64_t len = ends[o] - starts[o];
        if (len <= split) continue;
        const int32_t slot = (int32_t)out->long_row.size();
        out->long_row.push_back(rows[o]);
        out->long_chunk.push_back(total);
        const int64_t nch = ceil_div(len, chunk_len);
        out->chunk_long.insert(out->chunk_long.end(), (size_t)nch, slot);
        total += nch;
    }
    out->long_chunk.push_back(total);
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_spmm_plan_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, gcn_spmm_plan** plan_out,
                         void* stream) {
    return gcn_spmm_plan_create_split(n_rows, n_cols, nnz, indptr, 0, plan_out, stream);
}

int gcn_spmm_plan_create_split(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, int64_t split_entries,
                               gcn_spmm_plan** plan_out, void* stream) {
    if (!plan_out) return fail(GCN_ERR_BAD_ARG, "plan_out is null");
    *plan_out = nullptr;
    if (split_entries < 0 || (split_entries > 0 && split_entries < 16) || split_entries > kSplitWide)
        return fail(GCN_ERR_BAD_ARG, "split_entries must be 0 (default) or in [16, %d]", kSplitWide);
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_rows > INT32_MAX - 1 || n_cols > INT32_MAX)
        return fail(GCN_ERR_BAD_ARG, "bad shape %lld x %lld, nnz %lld", (long long)n_rows, (long long)n_cols,
                    (long long)nnz);
    if (!indptr) return fail(GCN_ERR_BAD_ARG, "indptr is null");
    hipStream_t st = (hipStream_t)stream;
    gcn_spmm_plan* p = new (std::nothrow) gcn_spmm_plan();
    if (!p) return fail(GCN_ERR_HIP, "out of host memory");
    p->n_rows = n_rows;
    p->n_cols = n_cols;
    p->nnz = nnz;
    const int64_t work = nnz + n_rows * kRowCost;
    p->strip_nnz = strip_nnz_for(work);
    // on a small matrix a chunk is a chain of dependent batches like any row: keep it short (the rows it
    // applies to are summed chunk-wise either way)
    p->chunk_len = p->strip_nnz < kStripNnz ? kChunkSmall : 0;
    p->split[0] = split_entries ? (int)std::min<int64_t>(split_entries, kSplitNarrow) : kSplitNarrow;
    p->split[1] = split_entries ? (int)split_entries : kSplitWide;
    if (p->chunk_len > p->split[0]) p->chunk_len = p->split[0];
    p->n_strips = std::max<int64_t>(1, ceil_div(work, p->strip_nnz));
    p->strip_row = nullptr;
    for (int t = 0; t < 2; ++t) {
        p->table[t].n_long = p->table[t].n_chunks = 0;
        p->table[t].long_row = nullptr;
        p->table[t].long_chunk = nullptr;
        p->table[t].chunk_long = nullptr;
    }
    p->timing = 0;
    p->hot_cols = 0;
    p->lds_hub_rows = 0;
    p->ev[0] = p->ev[1] = p->ev[2] = nullptr;
    int32_t* d_rows = nullptr;
    int64_t *d_starts = nullptr, *d_ends = nullptr;
    unsigned long long* d_count = nullptr;
    int rc = GCN_OK;
    auto cleanup_tmp = [&]() {
        (void)hipFree(d_rows);
        (void)hipFree(d_starts);
        (void)hipFree(d_ends);
        (void)hipFree(d_count);
    };
#define PLAN_TRY(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e__ = (expr);                                                                             \
        if (e__ != hipSuccess) {                                                                             \
            rc = fail(GCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            cleanup_tmp();                                                                                   \
            gcn_spmm_plan_destroy(p);                                                                        \
            return rc;                                                                                       \
        }                                                                                                    \
    } while (0)
    PLAN_TRY(hipGetDevice(&p->device));
    {   // Set-up of the no-allocation entry point on this device, best effort: no other product depends on
        // it, so a failure here (a device ordinal beyond the table, a symbol lookup error) is left for
        // gcn_spmm_csr_f32_noalloc to report -- it resolves the address again at its first call.
        void* unused = nullptr;
        (void)scratch_on_current_device(&unused);
    }
    PLAN_TRY(hipMalloc(&p->strip_row, (size_t)(p->n_strips + 1) * sizeof(int32_t)));
    strip_rows_kernel<<<(unsigned)ceil_div(p->n_strips + 1, kBlock), kBlock, 0, st>>>(
        n_rows, indptr, p->n_strips, p->strip_nnz, p->strip_row);
    PLAN_TRY(hipGetLastError());
    // rows longer than the smaller split threshold (at most nnz / (that threshold + 1) of them); the table
    // for the larger threshold is the subset of the same list
    const int64_t cap = nnz / (p->split[0] + 1);
    if (cap > 0 && n_rows > 0) {
        PLAN_TRY(hipMalloc(&d_rows, (size_t)cap * sizeof(int32_t)));
        PLAN_TRY(hipMalloc(&d_starts, (size_t)cap * sizeof(int64_t)));
        PLAN_TRY(hipMalloc(&d_ends, (size_t)cap * sizeof(int64_t)));
        PLAN_TRY(hipMalloc(&d_count, sizeof(unsigned long long)));
        PLAN_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), st));
        collect_long_rows_kernel<<<(unsigned)ceil_div(n_rows, kBlock), kBlock, 0, st>>>(
            n_rows, indptr, p->split[0], cap, d_rows, d_starts, d_ends, d_count);
        PLAN_TRY(hipGetLastError());
        unsigned long long count = 0;
        PLAN_TRY(hipMemcpyAsync(&count, d_count, sizeof(count), hipMemcpyDeviceToHost, st));
        PLAN_TRY(hipStreamSynchronize(st));
        if ((int64_t)count > cap) {
            cleanup_tmp();
            gcn_spmm_plan_destroy(p);
            return fail(GCN_ERR_BAD_ARG, "indptr is inconsistent with nnz=%lld", (long long)nnz);
        }
        if (count > 0) {
            std::vector<int32_t> rows(count);
            std::vector<int64_t> starts(count), ends(count);
            PLAN_TRY(hipMemcpy(rows.data(), d_rows, count * sizeof(int32_t), hipMemcpyDeviceToHost));
            PLAN_TRY(hipMemcpy(starts.data(), d_starts, count * sizeof(int64_t), hipMemcpyDeviceToHost));
            PLAN_TRY(hipMemcpy(ends.data(), d_ends, count * sizeof(int64_t), hipMemcpyDeviceToHost));
            for (int t = 0; t < 2; ++t) {
                LongTableHost host;
                build_long_table(rows, starts, ends, p->split[t], p->chunk_len ? (int64_t)p->chunk_len : p->split[t], &host);
                const std::vector<int32_t>& long_row = host.long_row;
                const std::vector<int64_t>& long_chunk = host.long_chunk;
                const std::vector<int32_t>& chunk_long = host.chunk_long;
                const int64_t total = long_chunk.back();
                const size_t nl = long_row.size();
                if (nl == 0) continue;
                gcn_spmm_plan::LongTable& lt = p->table[t];
                lt.n_long = (int64_t)nl;
                lt.n_chunks = total;
                PLAN_TRY(hipMalloc(&lt.long_row, nl * sizeof(int32_t)));
                PLAN_TRY(hipMalloc(&lt.long_chunk, (nl + 1) * sizeof(int64_t)));
                PLAN_TRY(hipMalloc(&lt.chunk_long, (size_t)total * sizeof(int32_t)));
                PLAN_TRY(hipMemcpy(lt.long_row, long_row.data(), nl * sizeof(int32_t), hipMemcpyHostToDevice));
                PLAN_TRY(hipMemcpy(lt.long_chunk, long_chunk.data(), (nl + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
                PLAN_TRY(hipMemcpy(lt.chunk_long, chunk_long.data(), (size_t)total * sizeof(int32_t),
                                   hipMemcpyHostToDevice));
            }
        }
    }
    PLAN_TRY(hipStreamSynchronize(st));
#undef PLAN_TRY
    cleanup_tmp();
    *plan_out = p;
    return GCN_OK;
}

int gcn_spmm_plan_destroy(gcn_spmm_plan* plan) {
    if (!plan) return GCN_OK;
    (void)hipFree(plan->strip_row);
    for (int t = 0; t < 2; ++t) {
        (void)hipFree(plan->table[t].long_row);
        (void)hipFree(plan->table[t].long_chunk);
        (void)hipFree(plan->table[t].chunk_long);
    }
    for (int i = 0; i < 3; ++i)
        if (plan->ev[i]) (void)hipEventDestroy(plan->ev[i]);
    delete plan;
    return GCN_OK;
}

int gcn_spmm_plan_set_timing(gcn_spmm_plan* plan, int enable) {
    if (!plan) return fail(GCN_ERR_BAD_ARG, "plan is null");
    if (enable && !plan->ev[0])
        for (int i = 0; i < 3; ++i) GCN_HIP_TRY(hipEventCreate(&plan->ev[i]));
    plan->timing = enable ? 1 : 0;
    return GCN_OK;
}

int gcn_spmm_plan_last_timing(const gcn_spmm_plan* plan, float* ms2) {
    if (!plan || !ms2) return fail(GCN_ERR_BAD_ARG, "null argument");
    if (!plan->timing || !plan->ev[0]) return fail(GCN_ERR_BAD_ARG, "timing is not enabled on this plan");
    GCN_HIP_TRY(hipEventSynchronize(plan->ev[2]));
    GCN_HIP_TRY(hipEventElapsedTime(&ms2[0], plan->ev[0], plan->ev[1]));
    GCN_HIP_TRY(hipEventElapsedTime(&ms2[1], plan->ev[1], plan->ev[2]));
    return GCN_OK;
}

int gcn_spmm_plan_set_hot_columns(gcn_spmm_plan* plan, int64_t hot_cols) {
    if (!plan || hot_cols < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    plan->hot_cols = hot_cols;
    return GCN_OK;
}

int gcn_spmm_plan_set_lds_hub_rows(gcn_spmm_plan* plan, int64_t hub_rows) {
    if (!plan || hub_rows < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    plan->lds_hub_rows = hub_rows;
    return GCN_OK;
}

int gcn_spmm_plan_info(const gcn_spmm_plan* plan, int64_t* what4) {
    if (!plan || !what4) return fail(GCN_ERR_BAD_ARG, "null argument");
    what4[0] = plan->n_strips;
    what4[1] = plan->table[0].n_long;
    what4[2] = plan->table[0].n_chunks;
    what4[3] = plan->split[0];
    return GCN_OK;
}

int gcn_spmm_plan_workspace_bytes(const gcn_spmm_plan* plan, int64_t f, size_t* workspace_bytes) {
    if (!plan || !workspace_bytes || f < 0) return fail(GCN_ERR_BAD_ARG, "bad argument");
    const int64_t max_chunks = std::max(plan->table[0].n_chunks, plan->table[1].n_chunks);
    *workspace_bytes = (size_t)max_chunks * (size_t)partial_ld(f) * sizeof(float);
    return GCN_OK;
}

int gcn_spmm_csr_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                     int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc, int flags, void* workspace,
                     size_t workspace_bytes, void* stream) {
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    return spmm_entry<false>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, workspace, workspace_bytes, dr,
                             stream);
}

int gcn_spmm_csr_f32_noalloc(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                             const float* vals, int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc,
                             int flags, void* stream) {
    if (!plan) return fail(GCN_ERR_BAD_ARG, "plan is null");
    if (f < 0) return fail(GCN_ERR_BAD_ARG, "bad width: f=%lld", (long long)f);
    int dev = -1;
    GCN_HIP_TRY(hipGetDevice(&dev));
    if (dev != plan->device)
        return fail(GCN_ERR_BAD_ARG, "the plan was made on device %d, the current device is %d", plan->device, dev);
    const int64_t max_chunks = std::max(plan->table[0].n_chunks, plan->table[1].n_chunks);
    const size_t need = (size_t)max_chunks * (size_t)partial_ld(f) * sizeof(float);
    if (need > kScratchBytes)
        return fail(GCN_ERR_WORKSPACE,
                    "this product needs %zu bytes for the partial rows of its long rows; the no-allocation entry "
                    "point holds %zu: call gcn_spmm_csr_f32 with a workspace of the caller's",
                    need, (size_t)kScratchBytes);
    void* ptr = nullptr;
    if (need > 0) {
        const int rc = scratch_on_current_device(&ptr);   // cached by plan creation (a table lookup here); resolved now if that failed
        if (rc != GCN_OK) return rc;
    }
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    return spmm_entry<false>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, ptr, need ? kScratchBytes : 0, dr,
                             stream);
}

int gcn_spmm_csr_two_part_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                              const float* vals, int64_t f, const float* b_first, int64_t ld_first, int64_t n_first,
                              const float* b_rest, int64_t ld_rest, float* c, int64_t ldc, int flags, float keep_prob,
                              uint64_t seed, uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset,
                              int64_t pitch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!plan) return fail(GCN_ERR_BAD_ARG, "plan is null");
    if (n_first < 0 || n_first > plan->n_cols || ld_first < f || ld_first > INT32_MAX)
        return fail(GCN_ERR_BAD_ARG, "bad first part: %lld rows of pitch %lld for a %lld-column matrix, width %lld",
                    (long long)n_first, (long long)ld_first, (long long)plan->n_cols, (long long)f);
    if (n_first > 0 && !b_first && plan->nnz > 0) return fail(GCN_ERR_BAD_ARG, "null dense pointer");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    OutDrop od = {0, 1.0f, 0, 0, nullptr, 0, 0};
    if (keep_prob < 1.0f) {
        if (index_offset < 0 || pitch < f)
            return fail(GCN_ERR_BAD_ARG, "bad decision layout: index_offset=%lld pitch=%lld for %lld columns",
                        (long long)index_offset, (long long)pitch, (long long)f);
        od = OutDrop{stream_key(seed, stream_id), keep_prob, index_offset, pitch, stream_offset, seed, stream_id};
    }
    const TwoPart tp = {b_first, (uint32_t)ld_first, (uint32_t)n_first};
    // a matrix whose every column lies in the first part never reads b_rest: any valid pointer stands in for it
    const float* rest = (n_first == plan->n_cols && !b_rest) ? b_first : b_rest;
    return spmm_entry<false>(plan, indptr, indices, vals, f, rest, n_first == plan->n_cols && !b_rest ? ld_first : ld_rest,
                             c, ldc, flags, workspace, workspace_bytes, dr, stream, nullptr, 0,
                             keep_prob < 1.0f ? &od : nullptr, &tp);
}

int gcn_spmm_csr_add_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                         int64_t f, const float* b, int64_t ldb, const float* add, int64_t ldadd, float* c,
                         int64_t ldc, int flags, void* workspace, size_t workspace_bytes, void* stream) {
    if (!add) return fail(GCN_ERR_BAD_ARG, "addend is null");
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    return spmm_entry<false>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, workspace, workspace_bytes, dr,
                             stream, add, ldadd);
}

int gcn_spmm_csr_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                             const float* vals, const int64_t* perm, int64_t f, const float* b, int64_t ldb, float* c,
                             int64_t ldc, int flags, float keep_prob, uint64_t seed, uint64_t stream_id,
                             const uint64_t* stream_offset,
                             void* workspace, size_t workspace_bytes, void* stream) {
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    DropArgs dr = {perm, stream_key(seed, stream_id), keep_prob, 1.0f / keep_prob, stream_offset, seed, stream_id,
                   keep_threshold(keep_prob)};
    return spmm_entry<true>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, workspace, workspace_bytes, dr,
                            stream);
}

int gcn_spmm_csr_out_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                                 const float* vals, int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc,
                                 int flags, float keep_prob, uint64_t seed, uint64_t stream_id,
                                 const uint64_t* stream_offset, int64_t index_offset, int64_t pitch, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    if (index_offset < 0 || pitch < f)
        return fail(GCN_ERR_BAD_ARG, "bad decision layout: index_offset=%lld pitch=%lld for %lld columns",
                    (long long)index_offset, (long long)pitch, (long long)f);
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    OutDrop od = {stream_key(seed, stream_id), keep_prob, index_offset, pitch, stream_offset, seed, stream_id};
    return spmm_entry<false>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, workspace, workspace_bytes, dr,
                             stream, nullptr, 0, &od);
}

int gcn_spmm_csr_add_out_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                                     const float* vals, int64_t f, const float* b, int64_t ldb, const float* add,
                                     int64_t ldadd, float* c, int64_t ldc, int flags, float keep_prob, uint64_t seed,
                                     uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset,
                                     int64_t pitch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!add) return fail(GCN_ERR_BAD_ARG, "addend is null");
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    if (index_offset < 0 || pitch < f)
        return fail(GCN_ERR_BAD_ARG, "bad decision layout: index_offset=%lld pitch=%lld for %lld columns",
                    (long long)index_offset, (long long)pitch, (long long)f);
    DropArgs dr = {nullptr, 0, 1.0f, 1.0f};
    OutDrop od = {stream_key(seed, stream_id), keep_prob, index_offset, pitch, stream_offset, seed, stream_id};
    return spmm_entry<false>(plan, indptr, indices, vals, f, b, ldb, c, ldc, flags, workspace, workspace_bytes, dr,
                             stream, add, ldadd, keep_prob < 1.0f ? &od : nullptr);
}

int gcn_pack_pitch(int64_t f, int64_t* pitch_floats) {
    if (!pitch_floats || f < 1 || f > kPackMaxF) return fail(GCN_ERR_BAD_ARG, "a packed row holds 1 to %d columns", kPackMaxF);
    *pitch_floats = pack_pitch(f);
    return GCN_OK;
}

static int pack_entry(bool drop, int64_t n_rows, int64_t f, const float* x, int64_t ldx, float* packed, int64_t ldpk,
                      const OutDrop& od, void* stream, const GatedSource* gated = nullptr) {
    if (n_rows < 0 || f < 1 || f > kPackMaxF) return fail(GCN_ERR_BAD_ARG, "a packed row holds 1 to %d columns", kPackMaxF);
    if (ldx < f || ldpk < pack_pitch(f) || ldpk % 32 != 0 || ldpk > INT32_MAX)
        return fail(GCN_ERR_BAD_ARG, "bad pitch: ldx=%lld, packed pitch %lld (needs a multiple of 32 floats, at least %lld)",
                    (long long)ldx, (long long)ldpk, (long long)pack_pitch(f));
    if (n_rows == 0) return GCN_OK;
    if (!x || !packed) return fail(GCN_ERR_BAD_ARG, "null pointer");
    if ((uintptr_t)packed & 127) return fail(GCN_ERR_BAD_ARG, "the packed matrix must be 128-byte aligned");
    const int x_vec = (ldx % 4 == 0 && aligned16(x)) ? 1 : 0;
    const int64_t blocks = std::min<int64_t>(ceil_div(n_rows, kBlock / 64), (int64_t)1 << 20);
    const GatedSource none = {nullptr, 0, nullptr, 0, 0, 1.0f, 0.0f, nullptr, 0};
    if (gated) {
        // every wave first loads its lanes' rows of w (64 values per lane): few waves, each walking many rows
        const int64_t few = std::min<int64_t>(blocks, 256 * 5);
        pack_rows_kernel<kPackGated><<<(unsigned)few, kBlock, 0, (hipStream_t)stream>>>(n_rows, (int)f, x, ldx, x_vec,
                                                                                       packed, ldpk, od, *gated);
    }
    else if (drop)
        pack_rows_kernel<kPackDrop><<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, (int)f, x, ldx, x_vec,
                                                                                         packed, ldpk, od, none);
    else
        pack_rows_kernel<kPackPlain><<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, (int)f, x, ldx, x_vec,
                                                                                          packed, ldpk, od, none);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_pack_rows_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float* packed, int64_t ldpk, void* stream) {
    const OutDrop none = {0, 1.0f, 0, 0, nullptr, 0, 0};
    return pack_entry(false, n_rows, f, x, ldx, packed, ldpk, none, stream);
}

int gcn_dropout_pack_rows_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float keep_prob, uint64_t seed,
                              uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                              const int64_t* row_ids, float* packed, int64_t ldpk, void* stream) {
    if (!(keep_prob > 0.0f) || keep_prob > 1.0f) return fail(GCN_ERR_BAD_ARG, "keep_prob must be in (0, 1]");
    if (index_offset < 0 || pitch < f)
        return fail(GCN_ERR_BAD_ARG, "bad decision layout: index_offset=%lld pitch=%lld for %lld columns",
                    (long long)index_offset, (long long)pitch, (long long)f);
    if (keep_prob == 1.0f) return gcn_pack_rows_f32(n_rows, f, x, ldx, packed, ldpk, stream);
    OutDrop od = out_drop_ready(OutDrop{stream_key(seed, stream_id), keep_prob, index_offset, pitch, stream_offset,
                                        seed, stream_id});
    od.row_ids = row_ids;
    return pack_entry(true, n_rows, f, x, ldx, packed, ldpk, od, stream);
}

int gcn_dropout_pack_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float keep_prob, uint64_t seed,
                         uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                         float* packed, int64_t ldpk, void* stream) {
    return gcn_dropout_pack_rows_f32(n_rows, f, x, ldx, keep_prob, seed, stream_id, stream_offset, index_offset, pitch,
                                     nullptr, packed, ldpk, stream);
}

int gcn_gemm_gated_pack_rows_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b,
                                 int64_t ldb, const float* gate, int64_t ldg, float divisor, const int64_t* rows,
                                 int64_t n_listed, float* packed, int64_t ldpk, void* stream) {
    if (m < 0 || n < 1 || n > kPackMaxF || k < 1 || k > kGatedMaxK)
        return fail(GCN_ERR_BAD_ARG, "the packed gated product takes 1 to %d columns and an inner dimension of 1 to %d",
                    kPackMaxF, kGatedMaxK);
    if (lda < k || ldb < k) return fail(GCN_ERR_BAD_ARG, "leading dimension too small");
    if (!(divisor > 0.0f)) return fail(GCN_ERR_BAD_ARG, "divisor must be positive");
    if (m > 0 && (!a || !b)) return fail(GCN_ERR_BAD_ARG, "null pointer");
    if (rows ? (n_listed < 0 || n_listed > m) : n_listed != 0)
        return fail(GCN_ERR_BAD_ARG, "bad row list: %lld rows listed of %lld", (long long)n_listed, (long long)m);
    int exponent = 0;
    const bool power_of_two = std::frexp(divisor, &exponent) == 0.5f && exponent > -100 && exponent < 100;
    const GatedSource gs = {a, lda, b, ldb, (int)k, divisor, power_of_two ? 1.0f / divisor : 0.0f, rows, m};
    const OutDrop none = {0, 1.0f, 0, 0, nullptr, 0, 0};
    if (!rows) return pack_entry(false, m, n, gate, ldg, packed, ldpk, none, stream, &gs);
    // with a row list: every row empty first (64 bytes each), then the listed rows formed
    if (m == 0) return GCN_OK;
    if (!gate || !packed) return fail(GCN_ERR_BAD_ARG, "null pointer");
    if (ldg < n || ldpk < pack_pitch(n) || ldpk % 32 != 0 || ldpk > INT32_MAX || ((uintptr_t)packed & 127))
        return fail(GCN_ERR_BAD_ARG, "bad pitch or alignment: ldg=%lld, packed pitch %lld", (long long)ldg, (long long)ldpk);
    const int64_t clear_blocks = std::min<int64_t>(ceil_div(4 * m, kBlock), 256 * 16);
    clear_packed_rows_kernel<<<(unsigned)clear_blocks, kBlock, 0, (hipStream_t)stream>>>(m, packed, ldpk);
    GCN_LAUNCH_CHECK();
    if (n_listed == 0) return GCN_OK;
    return pack_entry(false, n_listed, n, gate, ldg, packed, ldpk, none, stream, &gs);
}

int gcn_gemm_gated_pack_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                            const float* gate, int64_t ldg, float divisor, float* packed, int64_t ldpk, void* stream) {
    return gcn_gemm_gated_pack_rows_f32(m, n, k, a, lda, b, ldb, gate, ldg, divisor, nullptr, 0, packed, ldpk, stream);
}

int gcn_unpack_rows_f32(int64_t n_rows, int64_t f, const float* packed, int64_t ldpk, float* y, int64_t ldy, void* stream) {
    if (n_rows < 0 || f < 1 || f > kPackMaxF) return fail(GCN_ERR_BAD_ARG, "a packed row holds 1 to %d columns", kPackMaxF);
    if (ldy < f || ldpk < pack_pitch(f) || ldpk % 32 != 0) return fail(GCN_ERR_BAD_ARG, "bad pitch");
    if (n_rows == 0) return GCN_OK;
    if (!packed || !y) return fail(GCN_ERR_BAD_ARG, "null pointer");
    if ((uintptr_t)packed & 127) return fail(GCN_ERR_BAD_ARG, "the packed matrix must be 128-byte aligned");
    const int64_t blocks = std::min<int64_t>(ceil_div(n_rows, kBlock / 64), (int64_t)1 << 20);
    unpack_rows_kernel<<<(unsigned)blocks, kBlock, 0, (hipStream_t)stream>>>(n_rows, (int)f, packed, ldpk, y, ldy);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_spmm_csr_packed_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                            int64_t f, const float* packed_b, int64_t ldpk, float* c, int64_t ldc, int flags,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (!plan) return fail(GCN_ERR_BAD_ARG, "plan is null");
    if (f < 1 || f > kPackMaxF) return fail(GCN_ERR_BAD_ARG, "a packed row holds 1 to %d columns", kPackMaxF);
    if (ldc < f || ldpk < pack_pitch(f) || ldpk % 32 != 0 || ldpk > INT32_MAX)
        return fail(GCN_ERR_BAD_ARG, "bad pitch: ldc=%lld, packed pitch %lld", (long long)ldc, (long long)ldpk);
    if (!indptr || (plan->nnz > 0 && (!indices || !vals))) return fail(GCN_ERR_BAD_ARG, "null CSR pointer");
    if (flags & ~GCN_RELU) return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if (plan->n_rows == 0) return GCN_OK;
    if (!c || (plan->nnz > 0 && !packed_b)) return fail(GCN_ERR_BAD_ARG, "null dense pointer");
    if ((uintptr_t)packed_b & 127) return fail(GCN_ERR_BAD_ARG, "the packed matrix must be 128-byte aligned");
    const gcn_spmm_plan::LongTable& lt = plan->table[1];          // the cut points of the wide dense product
    constexpr int kWaves = kBlock / 64;
    PlanView pv;
    pv.n_strips = plan->n_strips;
    pv.nnz = plan->nnz;
    pv.n_chunks = lt.n_chunks;
    pv.chunk_blocks = ceil_div(lt.n_chunks, kWaves);
    pv.split = plan->split[1];
    pv.chunk_len = plan->chunk_len ? plan->chunk_len : pv.split;
    pv.hot_cols = 0;
    pv.lds_hub_rows = 0;
    pv.strip_row = plan->strip_row;
    pv.chunk_long = lt.chunk_long;
    pv.long_row = lt.long_row;
    pv.long_chunk = lt.long_chunk;
    float* part = nullptr;
    const int64_t ldp = partial_ld(f);
    if (lt.n_chunks > 0) {
        const size_t need = (size_t)lt.n_chunks * (size_t)ldp * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
        if (!aligned16(workspace)) return fail(GCN_ERR_BAD_ARG, "workspace must be 16-byte aligned");
        part = (float*)workspace;
    }
    const int64_t grid = pv.chunk_blocks + ceil_div(plan->n_strips, kWaves);
    if (grid > INT32_MAX) return fail(GCN_ERR_BAD_ARG, "matrix too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    const int relu = (flags & GCN_RELU) ? 1 : 0;
    const int c_vec = (ldc % 4 == 0 && aligned16(c)) ? 1 : 0;
    if (plan->timing) GCN_HIP_TRY(hipEventRecord(plan->ev[0], st));
    spmm_packed_kernel<kPackUnroll><<<(unsigned)grid, kBlock, 0, st>>>(pv, indptr, indices, vals, (int)f, packed_b,
                                                                      (uint32_t)ldpk, c, ldc, c_vec, part, ldp, relu);
    GCN_LAUNCH_CHECK();
    if (plan->timing) GCN_HIP_TRY(hipEventRecord(plan->ev[1], st));
    if (lt.n_chunks > 0) {
        const OutDrop none = {0, 1.0f, 0, 0, nullptr, 0, 0};
        spmm_long_reduce_kernel<<<(unsigned)ceil_div(lt.n_long, kWaves), kBlock, 0, st>>>(
            lt.n_long, lt.long_row, lt.long_chunk, f, part, ldp, c, ldc, relu, c_vec, nullptr, 0, none);
        GCN_LAUNCH_CHECK();
    }
    if (plan->timing) GCN_HIP_TRY(hipEventRecord(plan->ev[2], st));
    return GCN_OK;
}

}  // extern "C"
