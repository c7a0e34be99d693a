// Graph preprocessing for the GCN propagation path: COO -> CSR with self-loops, row sums,
// D^-1/2 scaling, stable transpose, R-MAT generation.  Integer kernels; the float64 scaling
// follows gcn/utils.py:297-323 operation by operation.
#include <atomic>
#include <cstring>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace gcn {

constexpr uint64_t kInvalidKey = ~0ull;
constexpr int kBlock = 256;

static inline unsigned bits_for(uint64_t v) {  // number of bits needed to hold values in [0, v]
    unsigned b = 0;
    while (v) {
        ++b;
        v >>= 1;
    }
    return b ? b : 1;
}

// Grids are capped (gridDim.x * blockDim.x must stay below 2^32); kernels walk their range with a grid stride.
constexpr int64_t kMaxBlocks = 1 << 20;
static inline dim3 grid_for(int64_t n) {
    const int64_t blocks = ceil_div(n > 0 ? n : 1, kBlock);
    return dim3((unsigned)(blocks < kMaxBlocks ? blocks : kMaxBlocks));
}
#define GCN_GRID_STRIDE(i, n)                                                              \
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); \
         i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------------------------
// sorted keys (row << 32 | col) -> CSR, merging equal keys
// ------------------------------------------------------------------------------------------------
__global__ void head_flags_kernel(int64_t m, const uint64_t* __restrict__ keys, int64_t* __restrict__ pos) {
    GCN_GRID_STRIDE(i, m) {
        const uint64_t k = keys[i];
        pos[i] = (k != kInvalidKey && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
    }
}

// pos holds the EXCLUSIVE scan of the head flags.  One thread per sorted element; heads write.
// mode 0: value = sum of vals over the run; mode 1: value = run length; mode 2: no value (binary)
__global__ void scatter_unique_kernel(int64_t m, int64_t n_rows, const uint64_t* __restrict__ keys,
                                      const double* __restrict__ vals, const int64_t* __restrict__ pos, int mode,
                                      int64_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                      double* __restrict__ vals_out, int64_t* __restrict__ nnz_out) {
    GCN_GRID_STRIDE(i, m) {
        const uint64_t k = keys[i];
        if (k == kInvalidKey) {
            if (i == 0) {  // no valid entry at all
                for (int64_t q = 0; q <= n_rows; ++q) indptr[q] = 0;
                *nnz_out = 0;
            }
            continue;
        }
        const int64_t row = (int64_t)(k >> 32);
        const bool head = (i == 0) || (keys[i - 1] != k);
        const int64_t j = pos[i] - (head ? 0 : 1);  // index of this element's unique entry
        if (head) {
            indices[j] = (int32_t)(uint32_t)k;
            if (mode != 2) {
                double s = (mode == 0) ? vals[i] : 1.0;
                for (int64_t t = i + 1; t < m && keys[t] == k; ++t) s += (mode == 0) ? vals[t] : 1.0;
                vals_out[j] = s;
            }
            const int64_t prev_row = (i == 0) ? -1 : (int64_t)(keys[i - 1] >> 32);
            for (int64_t q = prev_row + 1; q <= row; ++q) indptr[q] = j;  // rows that start here (and empty ones before)
        }
        const bool last = (i == m - 1) || (keys[i + 1] == kInvalidKey);
        if (last) {
            const int64_t total = j + 1;
            for (int64_t q = row + 1; q <= n_rows; ++q) indptr[q] = total;
            *nnz_out = total;
        }
    }
}

struct SortScratch {
    uint64_t* keys_a;
    uint64_t* keys_b;
    double* vals_a;
    double* vals_b;
    int64_t* pos;
    void* prim;
    size_t prim_bytes;
};

static size_t prim_temp_bytes(int64_t m, bool valued) {
    size_t a = 0, b = 0;
    rocprim::double_buffer<uint64_t> kd(nullptr, nullptr);
    if (valued) {
        rocprim::double_buffer<double> vd(nullptr, nullptr);
        (void)rocprim::radix_sort_pairs(nullptr, a, kd, vd, (size_t)m, 0u, 64u, (hipStream_t)0);
    } else {
        (void)rocprim::radix_sort_keys(nullptr, a, kd, (size_t)m, 0u, 64u, (hipStream_t)0);
    }
    (void)rocprim::exclusive_scan(nullptr, b, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)m,
                                  rocprim::plus<int64_t>(), (hipStream_t)0);
    return align_up(a > b ? a : b, 256) + 256;
}

static size_t sort_scratch_bytes(int64_t m, bool valued) {
    const size_t seg = align_up((size_t)(m > 0 ? m : 1) * 8, 256);
    return seg * (valued ? 5 : 3) + prim_temp_bytes(m, valued);
}

static int carve_scratch(void* workspace, size_t workspace_bytes, int64_t m, bool valued, SortScratch* s) {
    const size_t need = sort_scratch_bytes(m, valued);
    if (workspace == nullptr || workspace_bytes < need)
        return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) != 0) return fail(GCN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    const size_t seg = align_up((size_t)(m > 0 ? m : 1) * 8, 256);
    char* p = (char*)workspace;
    s->keys_a = (uint64_t*)p;
    p += seg;
    s->keys_b = (uint64_t*)p;
    p += seg;
    s->pos = (int64_t*)p;
    p += seg;
    s->vals_a = s->vals_b = nullptr;
    if (valued) {
        s->vals_a = (double*)p;
        p += seg;
        s->vals_b = (double*)p;
        p += seg;
    }
    s->prim = p;
    s->prim_bytes = prim_temp_bytes(m, valued);
    return GCN_OK;
}

// keys (and values) sit in scratch->keys_a / vals_a; sorts, merges, writes the CSR.
static int sorted_merge_to_csr(SortScratch* s, int64_t m, bool valued, int mode, unsigned key_bits, int64_t n_rows,
                               int64_t* indptr, int32_t* indices, double* vals_out, int64_t* nnz_out,
                               hipStream_t st) {
    if (m == 0) {
        GCN_HIP_TRY(hipMemsetAsync(indptr, 0, (size_t)(n_rows + 1) * 8, st));
        GCN_HIP_TRY(hipMemsetAsync(nnz_out, 0, 8, st));
        return GCN_OK;
    }
    rocprim::double_buffer<uint64_t> kd(s->keys_a, s->keys_b);
    size_t tb = s->prim_bytes;
    const double* sorted_vals = nullptr;
    if (valued) {
        rocprim::double_buffer<double> vd(s->vals_a, s->vals_b);
        GCN_HIP_TRY(rocprim::radix_sort_pairs(s->prim, tb, kd, vd, (size_t)m, 0u, key_bits, st));
        sorted_vals = vd.current();
    } else {
        // invalid keys are all ones: they need the full 64 bits to sort to the end
        GCN_HIP_TRY(rocprim::radix_sort_keys(s->prim, tb, kd, (size_t)m, 0u, key_bits, st));
    }
    const uint64_t* sorted = kd.current();
    head_flags_kernel<<<grid_for(m), kBlock, 0, st>>>(m, sorted, s->pos);
    GCN_LAUNCH_CHECK();
    tb = s->prim_bytes;
    GCN_HIP_TRY(rocprim::exclusive_scan(s->prim, tb, s->pos, s->pos, (int64_t)0, (size_t)m, rocprim::plus<int64_t>(),
                                        st));
    scatter_unique_kernel<<<grid_for(m), kBlock, 0, st>>>(m, n_rows, sorted, sorted_vals, s->pos, mode, indptr,
                                                          indices, vals_out, nnz_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

__global__ void pack_coo_keys_kernel(int64_t nnz_in, int64_t n_loops, const int32_t* __restrict__ rows,
                                     const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                     uint64_t* __restrict__ keys, double* __restrict__ kvals) {
    GCN_GRID_STRIDE(i, nnz_in + n_loops) {
        if (i < nnz_in) {
            keys[i] = ((uint64_t)(uint32_t)rows[i] << 32) | (uint32_t)cols[i];
            if (kvals) kvals[i] = vals ? vals[i] : 1.0;
        } else {
            const uint64_t d = (uint64_t)(i - nnz_in);
            keys[i] = (d << 32) | d;  // the sp.eye(n) entry
            if (kvals) kvals[i] = 1.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// row-grouped triple -> indptr
// ------------------------------------------------------------------------------------------------
__global__ void rows_to_indptr_kernel(int64_t n_rows, int64_t nnz, const int32_t* __restrict__ rows,
                                      int64_t* __restrict__ indptr) {
    GCN_GRID_STRIDE(k, nnz) {
        const int64_t r = rows[k];
        const int64_t rp = (k == 0) ? -1 : (int64_t)rows[k - 1];
        for (int64_t q = rp + 1; q <= r; ++q) indptr[q] = k;
        if (k == nnz - 1)
            for (int64_t q = r + 1; q <= n_rows; ++q) indptr[q] = nnz;
    }
}

// ------------------------------------------------------------------------------------------------
// row sums, degree scaling, normalisation
// ------------------------------------------------------------------------------------------------
__global__ void row_sums_kernel(int64_t n_rows, const int64_t* __restrict__ indptr, const double* __restrict__ vals,
                                double* __restrict__ out) {
    GCN_GRID_STRIDE(r, n_rows) {
        const int64_t k0 = indptr[r], k1 = indptr[r + 1];
        double s = 0.0;
        if (vals) {
            for (int64_t k = k0; k < k1; ++k) s += vals[k];  // storage order, like the reference's sum(1)
        } else {
            s = (double)(k1 - k0);
        }
        out[r] = s;
    }
}

// x^-0.5 to (all but provably) correct rounding: 1/sqrt(x) refined by one Newton step whose
// residual is evaluated with an exact product split, so the result agrees with a correctly
// rounded pow(x, -0.5).
__device__ __forceinline__ double inv_sqrt_rn(double x) {
    const double y0 = 1.0 / sqrt(x);
    const double t_hi = x * y0;
    const double t_lo = fma(x, y0, -t_hi);
    const double e = fma(-t_hi, y0, 1.0) - t_lo * y0;  // 1 - x*y0^2
    return fma(0.5 * y0, e, y0);
}

__global__ void degree_scale_kernel(int64_t n, const double* __restrict__ rowsum, int type, double* __restrict__ d) {
    GCN_GRID_STRIDE(r, n) {
        const double x = rowsum[r];
        double y;
        if (type == GCN_NORM_SYM) {
            y = (x > 0.0) ? inv_sqrt_rn(x) : ((x == 0.0) ? 0.0 : __longlong_as_double(0x7ff8000000000000ll));
        } else {
            y = (x != 0.0) ? 1.0 / x : 0.0;
        }
        if (isinf(y)) y = 0.0;
        d[r] = y;
    }
}

// bag-of-words row scaling: one thread per row, sum and scale in storage order
template <typename T>
__global__ void row_normalize_kernel(int64_t n_rows, const int64_t* __restrict__ indptr, const T* __restrict__ vals,
                                     T* __restrict__ out) {
    GCN_GRID_STRIDE(r, n_rows) {
        const int64_t k0 = indptr[r], k1 = indptr[r + 1];
        T s = (T)0;
        for (int64_t k = k0; k < k1; ++k) s += vals[k];
        T inv = (T)1 / s;  // np.power(rowsum, -1)
        if (isinf(inv)) inv = (T)0;
        for (int64_t k = k0; k < k1; ++k) out[k] = inv * vals[k];
    }
}

__device__ __forceinline__ int64_t row_of_entry(const int64_t* __restrict__ indptr, int64_t n_rows, int64_t k) {
    // largest r with indptr[r] <= k
    int64_t lo = 0, hi = n_rows;  // invariant: indptr[lo] <= k < indptr[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= k)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ void normalize_kernel(int64_t n_rows, int64_t nnz, const int64_t* __restrict__ indptr,
                                 const int32_t* __restrict__ indices, const double* __restrict__ vals,
                                 const double* __restrict__ d, int type, double* __restrict__ out64,
                                 float* __restrict__ out32) {
    GCN_GRID_STRIDE(k, nnz) {
        const int64_t r = row_of_entry(indptr, n_rows, k);
        const double a = vals ? vals[k] : 1.0;
        double w;
        if (type == GCN_NORM_SYM) {
            // adj.dot(D).transpose().dot(D): the stored (r, c) entry becomes result[c, r] = (a * d[c]) * d[r]
            const double t = __dmul_rn(a, d[indices[k]]);
            w = __dmul_rn(t, d[r]);
        } else {
            w = __dmul_rn(d[r], a);
        }
        if (out64) out64[k] = w;
        if (out32) out32[k] = (float)w;
    }
}

// ------------------------------------------------------------------------------------------------
// stable transpose
// ------------------------------------------------------------------------------------------------
__global__ void transpose_keys_kernel(int64_t nnz, const int32_t* __restrict__ indices, uint32_t* __restrict__ keys,
                                      int64_t* __restrict__ src) {
    GCN_GRID_STRIDE(k, nnz) {
        keys[k] = (uint32_t)indices[k];
        src[k] = k;
    }
}

__global__ void transpose_scatter_kernel(int64_t n_rows, int64_t n_cols, int64_t nnz,
                                         const int64_t* __restrict__ indptr, const uint32_t* __restrict__ keys,
                                         const int64_t* __restrict__ src, const void* __restrict__ vals, int val_bytes,
                                         int64_t* __restrict__ t_indptr, int32_t* __restrict__ t_indices,
                                         void* __restrict__ t_vals, int64_t* __restrict__ perm) {
    GCN_GRID_STRIDE(j, nnz) {
        const int64_t k = src[j];
        t_indices[j] = (int32_t)row_of_entry(indptr, n_rows, k);
        if (val_bytes == 4) ((float*)t_vals)[j] = ((const float*)vals)[k];
        if (val_bytes == 8) ((double*)t_vals)[j] = ((const double*)vals)[k];
        if (perm) perm[j] = k;
        const int64_t c = keys[j];
        const int64_t cp = (j == 0) ? -1 : (int64_t)keys[j - 1];
        for (int64_t q = cp + 1; q <= c; ++q) t_indptr[q] = j;
        if (j == nnz - 1)
            for (int64_t q = c + 1; q <= n_cols; ++q) t_indptr[q] = nnz;
    }
}

static size_t transpose_prim_bytes(int64_t nnz) {
    size_t a = 0;
    rocprim::double_buffer<uint32_t> kd(nullptr, nullptr);
    rocprim::double_buffer<int64_t> vd(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, a, kd, vd, (size_t)nnz, 0u, 32u, (hipStream_t)0);
    return align_up(a, 256) + 256;
}

// ------------------------------------------------------------------------------------------------
// vertex reordering: degree-descending order and relabelling of a CSR structure
// ------------------------------------------------------------------------------------------------
// key = (2^32 - 1 - degree) << 32 | vertex: ascending keys = descending degree, ties by ascending id
__global__ void degree_keys_kernel(int64_t n, const int64_t* __restrict__ indptr, uint64_t* __restrict__ keys) {
    GCN_GRID_STRIDE(v, n) {
        const uint64_t deg = (uint64_t)(indptr[v + 1] - indptr[v]);
        keys[v] = ((0xFFFFFFFFull - deg) << 32) | (uint64_t)v;
    }
}

__global__ void order_from_keys_kernel(int64_t n, const uint64_t* __restrict__ sorted, int32_t* __restrict__ new_id,
                                       int32_t* __restrict__ old_id) {
    GCN_GRID_STRIDE(i, n) {
        const uint32_t v = (uint32_t)sorted[i];
        if (old_id) old_id[i] = (int32_t)v;
        if (new_id) new_id[v] = (int32_t)i;
    }
}

// Hub-first order.  sorted: the degree keys in ascending order (descending degree).  Position i < n_hubs: the
// vertex is a hub and takes the new id i; every other vertex is flagged 1 in rest_flag (hubs 0).
__global__ void hub_marks_kernel(int64_t n, int64_t n_hubs, const uint64_t* __restrict__ sorted,
                                 int32_t* __restrict__ new_id, int64_t* __restrict__ rest_flag) {
    GCN_GRID_STRIDE(i, n) {
        const uint32_t v = (uint32_t)sorted[i];
        if (i < n_hubs) {
            new_id[v] = (int32_t)i;
            rest_flag[v] = 0;
        } else {
            rest_flag[v] = 1;
        }
    }
}

// rest_rank: the exclusive scan of rest_flag = how many non-hub vertices precede v in the old order
__global__ void hub_finish_kernel(int64_t n, int64_t n_hubs, const int64_t* __restrict__ rest_flag,
                                  const int64_t* __restrict__ rest_rank, int32_t* __restrict__ new_id,
                                  int32_t* __restrict__ old_id) {
    GCN_GRID_STRIDE(v, n) {
        int32_t id = new_id[v];
        if (rest_flag[v]) {
            id = (int32_t)(n_hubs + rest_rank[v]);
            new_id[v] = id;
        }
        if (old_id) old_id[id] = (int32_t)v;
    }
}

__global__ void relabel_keys_kernel(int64_t n_rows, int64_t nnz, const int64_t* __restrict__ indptr,
                                    const int32_t* __restrict__ indices, const int32_t* __restrict__ new_row,
                                    const int32_t* __restrict__ new_col, const double* __restrict__ vals,
                                    uint64_t* __restrict__ keys, double* __restrict__ kvals) {
    GCN_GRID_STRIDE(k, nnz) {
        const int64_t r = row_of_entry(indptr, n_rows, k);
        const uint32_t c = (uint32_t)indices[k];
        const uint32_t nr = new_row ? (uint32_t)new_row[r] : (uint32_t)r;
        const uint32_t nc = new_col ? (uint32_t)new_col[c] : c;
        keys[k] = ((uint64_t)nr << 32) | nc;
        if (kvals) kvals[k] = vals[k];
    }
}

// ------------------------------------------------------------------------------------------------
// R-MAT and synthetic features
// ------------------------------------------------------------------------------------------------
__global__ void rmat_keys_kernel(int scale, int64_t n_pairs, int64_t n_loops, uint64_t key, uint64_t label_key,
                                 int scramble, uint32_t ta, uint32_t tb, uint32_t tc, uint64_t* __restrict__ keys) {
    GCN_GRID_STRIDE(e, n_pairs + n_loops) {
        if (e >= n_pairs) {  // the diagonal of A + I
            const uint64_t d = (uint64_t)(e - n_pairs);
            keys[2 * n_pairs + d] = (d << 32) | d;
            continue;
        }
        uint64_t u = 0, v = 0;
        for (int level = 0; level < scale; ++level) {
            const uint32_t r = (uint32_t)(sm64(key ^ ((uint64_t)e * (uint64_t)scale + (uint64_t)level)) >> 32);
            const uint32_t q = (r >= ta) + (r >= tb) + (r >= tc);
            u = (u << 1) | (q >> 1);
            v = (v << 1) | (q & 1);
        }
        const bool loop = (u == v);
        if (scramble) {
            u = scramble_id(u, scale, label_key);
            v = scramble_id(v, scale, label_key);
        }
        keys[2 * e] = loop ? kInvalidKey : ((u << 32) | v);
        keys[2 * e + 1] = loop ? kInvalidKey : ((v << 32) | u);
    }
}

__global__ void fill_synthetic_kernel(int64_t count, uint64_t key, uint64_t first, float* __restrict__ out) {
    GCN_GRID_STRIDE(i, count) {
        const int32_t hi = (int32_t)(uint32_t)(sm64(key ^ (first + (uint64_t)i)) >> 32);
        out[i] = (float)(hi >> 8) * 1.1920928955078125e-07f;  // 2^-23
    }
}

// The generated pair e of the R-MAT stream, ids scrambled or not: the arithmetic of rmat_keys_kernel.
__device__ __forceinline__ void rmat_pair(int scale, uint64_t key, uint64_t label_key, int scramble, uint32_t ta,
                                          uint32_t tb, uint32_t tc, int64_t e, uint64_t* u_out, uint64_t* v_out,
                                          bool* loop_out) {
    uint64_t u = 0, v = 0;
    for (int level = 0; level < scale; ++level) {
        const uint32_t r = (uint32_t)(sm64(key ^ ((uint64_t)e * (uint64_t)scale + (uint64_t)level)) >> 32);
        const uint32_t q = (r >= ta) + (r >= tb) + (r >= tc);
        u = (u << 1) | (q >> 1);
        v = (v << 1) | (q & 1);
    }
    *loop_out = (u == v);
    if (scramble) {
        u = scramble_id(u, scale, label_key);
        v = scramble_id(v, scale, label_key);
    }
    *u_out = u;
    *v_out = v;
}

// Keys of the rows [row_begin, row_end) only.  keys == nullptr: count them.  Otherwise every key takes the
// next free slot (the order of the slots does not matter: the keys are sorted afterwards).
__global__ void rmat_rows_keys_kernel(int scale, int64_t n_pairs, uint64_t key, uint64_t label_key, int scramble,
                                      uint32_t ta, uint32_t tb, uint32_t tc, uint64_t row_begin, uint64_t row_end,
                                      unsigned long long* __restrict__ counter, int64_t capacity,
                                      uint64_t* __restrict__ keys) {
    // every wave runs the same number of rounds, so that all 64 lanes meet at the ballots below: one
    // atomic per wave and round instead of one per key
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t wave_first = first - (threadIdx.x & 63);
    for (int64_t base = wave_first; base < n_pairs; base += stride) {
        const int64_t e = base + (threadIdx.x & 63);
        uint64_t u = 0, v = 0;
        bool mine_u = false, mine_v = false;
        if (e < n_pairs) {
            bool loop;
            rmat_pair(scale, key, label_key, scramble, ta, tb, tc, e, &u, &v, &loop);
            mine_u = !loop && u >= row_begin && u < row_end;
            mine_v = !loop && v >= row_begin && v < row_end;
        }
        const uint64_t bu = __ballot(mine_u), bv = __ballot(mine_v);
        const int total = __popcll(bu) + __popcll(bv);
        if (total == 0) continue;                                   // the same in every lane of the wave
        const int lane = threadIdx.x & 63;
        unsigned long long wave_slot = 0;
        if (lane == 0) wave_slot = atomicAdd(counter, (unsigned long long)total);
        wave_slot = __shfl(wave_slot, 0);
        if (!keys) continue;
        const uint64_t below = (1ull << lane) - 1ull;
        int64_t at = (int64_t)wave_slot + __popcll(bu & below) + __popcll(bv & below);
        if (mine_u && at < capacity) keys[at++] = ((u - row_begin) << 32) | v;
        if (mine_v && at < capacity) keys[at] = ((v - row_begin) << 32) | u;
    }
}

__global__ void diagonal_keys_kernel(int64_t n_local, uint64_t row_begin, uint64_t* __restrict__ keys) {
    GCN_GRID_STRIDE(i, n_local) keys[i] = ((uint64_t)i << 32) | (row_begin + (uint64_t)i);
}

// One range check: the n values at p must lie in [0, bound).  A null p checks nothing.
struct RangeCheck {
    const int32_t* p;
    int64_t n;
    int64_t bound;
};

// Up to three checks in one pass; bit i of *bad is set when check i found a value outside its range.
__global__ void check_ranges_kernel(RangeCheck c0, RangeCheck c1, RangeCheck c2, int64_t n_max,
                                    int* __restrict__ bad) {
    int mine = 0;
    GCN_GRID_STRIDE(k, n_max) {
        if (c0.p && k < c0.n) {
            const int64_t v = c0.p[k];
            if (v < 0 || v >= c0.bound) mine |= 1;
        }
        if (c1.p && k < c1.n) {
            const int64_t v = c1.p[k];
            if (v < 0 || v >= c1.bound) mine |= 2;
        }
        if (c2.p && k < c2.n) {
            const int64_t v = c2.p[k];
            if (v < 0 || v >= c2.bound) mine |= 4;
        }
    }
    if (mine) atomicOr(bad, mine);
}

// Flag words of the range checks: a small array inside the library (one copy per GPU), so that a check
// makes no hipMalloc / hipFree -- a hipFree waits for the whole device.  A call takes the next of
// kFlagSlots words in turn; a check holds its word only until it returns (it waits for its stream), so
// the words of up to kFlagSlots host threads checking at once never collide.
constexpr int kFlagSlots = 256;
constexpr int kFlagStride = 16;   // ints: one 64-byte line per slot
constexpr int kMaxDevices = 64;
static __device__ int range_flags[kFlagSlots * kFlagStride];
static int* g_flags_ptr[kMaxDevices];
static std::mutex g_flags_mutex;
static std::atomic<unsigned> g_flag_turn{0};

static int next_flag_word(int** out) {
    int dev = -1;
    GCN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(GCN_ERR_BAD_ARG, "device ordinal %d is not supported", dev);
    {
        std::lock_guard<std::mutex> lock(g_flags_mutex);
        if (!g_flags_ptr[dev]) {
            void* p = nullptr;
            GCN_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(range_flags)));
            if (!p) return fail(GCN_ERR_HIP, "hipGetSymbolAddress returned a null address for the flag array");
            g_flags_ptr[dev] = (int*)p;
        }
    }
    *out = g_flags_ptr[dev] + (size_t)(g_flag_turn.fetch_add(1) % kFlagSlots) * kFlagStride;
    return GCN_OK;
}

// Range checks before any kernel uses an index as an address: the reference raises ValueError (scipy's
// coo_matrix) or InvalidArgumentError (TensorFlow's per-nonzero check) for the same inputs.  One kernel
// for all the checks of a call, no allocation; waits for the stream.  *bits_out: bit i = check i failed.
static int check_ranges(const RangeCheck (&c)[3], hipStream_t st, int* bits_out) {
    *bits_out = 0;
    int64_t n_max = 0;
    for (const RangeCheck& r : c)
        if (r.p && r.n > n_max) n_max = r.n;
    if (n_max == 0) return GCN_OK;
    int* flag = nullptr;
    const int rc = next_flag_word(&flag);
    if (rc != GCN_OK) return rc;
    GCN_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), st));
    check_ranges_kernel<<<grid_for(n_max), kBlock, 0, st>>>(c[0], c[1], c[2], n_max, flag);
    GCN_LAUNCH_CHECK();
    int host = 0;
    GCN_HIP_TRY(hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, st));
    GCN_HIP_TRY(hipStreamSynchronize(st));
    *bits_out = host;
    return GCN_OK;
}

// rows against n_rows and cols against n_cols (either may be null), nnz of each
static int check_coo_range(int64_t nnz, const int32_t* rows, int64_t n_rows, const int32_t* cols, int64_t n_cols,
                           hipStream_t st) {
    const RangeCheck c[3] = {{rows, nnz, n_rows}, {cols, nnz, n_cols}, {nullptr, 0, 0}};
    int bits = 0;
    const int rc = check_ranges(c, st, &bits);
    if (rc != GCN_OK) return rc;
    if (bits & 1) return fail(GCN_ERR_INDEX_RANGE, "a row index is outside [0, %lld)", (long long)n_rows);
    if (bits & 2) return fail(GCN_ERR_INDEX_RANGE, "a column index is outside [0, %lld)", (long long)n_cols);
    return GCN_OK;
}

}  // namespace gcn

using namespace gcn;

extern "C" {

int gcn_coo_to_csr_bound(int64_t n_rows, int64_t nnz_in, int flags, int64_t* nnz_capacity, size_t* workspace_bytes) {
    if (n_rows < 0 || nnz_in < 0 || !nnz_capacity || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    const int64_t m = nnz_in + ((flags & GCN_ADD_SELF_LOOPS) ? n_rows : 0);
    *nnz_capacity = m > 0 ? m : 1;
    *workspace_bytes = sort_scratch_bytes(m, !(flags & GCN_BINARY));
    return GCN_OK;
}

int gcn_coo_to_csr(int64_t n_rows, int64_t n_cols, int64_t nnz_in, const int32_t* rows, const int32_t* cols,
                   const double* vals, int flags, int64_t* indptr_out, int32_t* indices_out, double* vals_out,
                   int64_t* nnz_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz_in < 0 || n_rows > INT32_MAX || n_cols > INT32_MAX)
        return fail(GCN_ERR_BAD_ARG, "bad shape %lld x %lld, nnz %lld", (long long)n_rows, (long long)n_cols,
                    (long long)nnz_in);
    if ((nnz_in > 0 && (!rows || !cols)) || !indptr_out || !indices_out || !nnz_out)
        return fail(GCN_ERR_BAD_ARG, "null pointer");
    const bool binary = (flags & GCN_BINARY) != 0;
    if (!binary && !vals_out) return fail(GCN_ERR_BAD_ARG, "vals_out is required unless GCN_BINARY is set");
    if ((flags & GCN_ADD_SELF_LOOPS) && n_rows != n_cols)
        return fail(GCN_ERR_BAD_ARG, "self-loops need a square matrix");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_loops = (flags & GCN_ADD_SELF_LOOPS) ? n_rows : 0;
    const int64_t m = nnz_in + n_loops;
    const bool valued = !binary;
    SortScratch s;
    int rc = carve_scratch(workspace, workspace_bytes, m, valued, &s);
    if (rc != GCN_OK) return rc;
    // the keys pack each index into 32 bits and the scatter walks indptr by row: nothing below is safe
    // for an index outside the shape
    rc = check_coo_range(nnz_in, rows, n_rows, cols, n_cols, st);
    if (rc != GCN_OK) return rc;
    if (m > 0) {
        pack_coo_keys_kernel<<<grid_for(m), kBlock, 0, st>>>(nnz_in, n_loops, rows, cols, vals, s.keys_a,
                                                             valued ? s.vals_a : nullptr);
        GCN_LAUNCH_CHECK();
    }
    const unsigned key_bits = 32 + bits_for((uint64_t)(n_rows > 0 ? n_rows - 1 : 0));
    return sorted_merge_to_csr(&s, m, valued, binary ? 2 : 0, key_bits, n_rows, indptr_out, indices_out, vals_out,
                               nnz_out, st);
}

int gcn_rows_to_indptr(int64_t n_rows, int64_t nnz, const int32_t* rows, int64_t* indptr_out, void* stream) {
    if (n_rows < 0 || nnz < 0 || !indptr_out || (nnz > 0 && !rows)) return fail(GCN_ERR_BAD_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (nnz == 0) {
        GCN_HIP_TRY(hipMemsetAsync(indptr_out, 0, (size_t)(n_rows + 1) * 8, st));
        return GCN_OK;
    }
    const int rc = check_coo_range(nnz, rows, n_rows, nullptr, 0, st);
    if (rc != GCN_OK) return rc;
    rows_to_indptr_kernel<<<grid_for(nnz), kBlock, 0, st>>>(n_rows, nnz, rows, indptr_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_row_sums(int64_t n_rows, const int64_t* indptr, const double* vals, double* rowsum_out, void* stream) {
    if (n_rows < 0 || !indptr || !rowsum_out) return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (n_rows == 0) return GCN_OK;
    row_sums_kernel<<<grid_for(n_rows), kBlock, 0, (hipStream_t)stream>>>(n_rows, indptr, vals, rowsum_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_row_normalize(int64_t n_rows, int64_t nnz, const int64_t* indptr, const void* vals, int val_bytes,
                      void* vals_out, void* stream) {
    if (n_rows < 0 || nnz < 0 || !indptr || (nnz > 0 && (!vals || !vals_out)))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (val_bytes != 4 && val_bytes != 8) return fail(GCN_ERR_BAD_ARG, "val_bytes must be 4 or 8");
    if (n_rows == 0 || nnz == 0) return GCN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (val_bytes == 4)
        row_normalize_kernel<float><<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, indptr, (const float*)vals,
                                                                         (float*)vals_out);
    else
        row_normalize_kernel<double><<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, indptr, (const double*)vals,
                                                                          (double*)vals_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_degree_scale(int64_t n_rows, const double* rowsum, int type, double* d_out, void* stream) {
    if (n_rows < 0 || !rowsum || !d_out || (type != GCN_NORM_SYM && type != GCN_NORM_RW))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (n_rows == 0) return GCN_OK;
    degree_scale_kernel<<<grid_for(n_rows), kBlock, 0, (hipStream_t)stream>>>(n_rows, rowsum, type, d_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_normalize_adj(int64_t n_rows, int64_t nnz, const int64_t* indptr, const int32_t* indices, const double* vals,
                      const double* d, int type, double* vals64_out, float* vals32_out, void* stream) {
    if (n_rows < 0 || nnz < 0 || !indptr || !d || (nnz > 0 && !indices) ||
        (type != GCN_NORM_SYM && type != GCN_NORM_RW))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (nnz == 0) return GCN_OK;
    normalize_kernel<<<grid_for(nnz), kBlock, 0, (hipStream_t)stream>>>(n_rows, nnz, indptr, indices, vals, d, type,
                                                                        vals64_out, vals32_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_csr_transpose_bound(int64_t n_rows, int64_t n_cols, int64_t nnz, size_t* workspace_bytes) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    const size_t n = (size_t)(nnz > 0 ? nnz : 1);
    *workspace_bytes = 2 * align_up(n * 4, 256) + 2 * align_up(n * 8, 256) + transpose_prim_bytes(nnz);
    return GCN_OK;
}

int gcn_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                      const void* vals, int val_bytes, int64_t* t_indptr, int32_t* t_indices, void* t_vals,
                      int64_t* perm_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_rows > INT32_MAX || n_cols > INT32_MAX || !indptr || !t_indptr ||
        (nnz > 0 && (!indices || !t_indices)))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (val_bytes != 0 && val_bytes != 4 && val_bytes != 8) return fail(GCN_ERR_BAD_ARG, "val_bytes must be 0, 4 or 8");
    if (val_bytes != 0 && nnz > 0 && (!vals || !t_vals)) return fail(GCN_ERR_BAD_ARG, "null value pointer");
    hipStream_t st = (hipStream_t)stream;
    if (nnz == 0) {
        GCN_HIP_TRY(hipMemsetAsync(t_indptr, 0, (size_t)(n_cols + 1) * 8, st));
        return GCN_OK;
    }
    size_t need = 0;
    gcn_csr_transpose_bound(n_rows, n_cols, nnz, &need);
    if (!workspace || workspace_bytes < need)
        return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) != 0) return fail(GCN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    char* p = (char*)workspace;
    uint32_t* ka = (uint32_t*)p;
    p += align_up((size_t)nnz * 4, 256);
    uint32_t* kb = (uint32_t*)p;
    p += align_up((size_t)nnz * 4, 256);
    int64_t* va = (int64_t*)p;
    p += align_up((size_t)nnz * 8, 256);
    int64_t* vb = (int64_t*)p;
    p += align_up((size_t)nnz * 8, 256);
    size_t tb = transpose_prim_bytes(nnz);
    transpose_keys_kernel<<<grid_for(nnz), kBlock, 0, st>>>(nnz, indices, ka, va);
    GCN_LAUNCH_CHECK();
    rocprim::double_buffer<uint32_t> kd(ka, kb);
    rocprim::double_buffer<int64_t> vd(va, vb);
    const unsigned bits = bits_for((uint64_t)(n_cols > 0 ? n_cols - 1 : 0));
    GCN_HIP_TRY(rocprim::radix_sort_pairs(p, tb, kd, vd, (size_t)nnz, 0u, bits, st));
    transpose_scatter_kernel<<<grid_for(nnz), kBlock, 0, st>>>(n_rows, n_cols, nnz, indptr, kd.current(),
                                                               vd.current(), vals, val_bytes, t_indptr, t_indices,
                                                               t_vals, perm_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_degree_order_bound(int64_t n_rows, size_t* workspace_bytes) {
    if (n_rows < 0 || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    *workspace_bytes = sort_scratch_bytes(n_rows, false);
    return GCN_OK;
}

int gcn_degree_order(int64_t n_rows, const int64_t* indptr, int32_t* new_id_out, int32_t* old_id_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (n_rows < 0 || n_rows > INT32_MAX || !indptr || (!new_id_out && !old_id_out))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (n_rows == 0) return GCN_OK;
    hipStream_t st = (hipStream_t)stream;
    SortScratch s;
    int rc = carve_scratch(workspace, workspace_bytes, n_rows, false, &s);
    if (rc != GCN_OK) return rc;
    degree_keys_kernel<<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, indptr, s.keys_a);
    GCN_LAUNCH_CHECK();
    rocprim::double_buffer<uint64_t> kd(s.keys_a, s.keys_b);
    size_t tb = s.prim_bytes;
    GCN_HIP_TRY(rocprim::radix_sort_keys(s.prim, tb, kd, (size_t)n_rows, 0u, 64u, st));
    order_from_keys_kernel<<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, kd.current(), new_id_out, old_id_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_hub_order_bound(int64_t n_rows, size_t* workspace_bytes) {
    if (n_rows < 0 || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    // the sort's scratch (its `pos` segment holds the flags) and one more segment for their scan
    *workspace_bytes = sort_scratch_bytes(n_rows, false) + align_up((size_t)(n_rows > 0 ? n_rows : 1) * 8, 256);
    return GCN_OK;
}

int gcn_hub_order(int64_t n_rows, const int64_t* indptr, int64_t n_hubs, int32_t* new_id_out, int32_t* old_id_out,
                  void* workspace, size_t workspace_bytes, void* stream) {
    if (n_rows < 0 || n_rows > INT32_MAX || n_hubs < 0 || !indptr || !new_id_out)
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (n_rows == 0) return GCN_OK;
    if (n_hubs > n_rows) n_hubs = n_rows;
    hipStream_t st = (hipStream_t)stream;
    size_t need = 0;
    gcn_hub_order_bound(n_rows, &need);
    if (!workspace || workspace_bytes < need)
        return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    SortScratch s;
    const size_t sort_bytes = sort_scratch_bytes(n_rows, false);
    int rc = carve_scratch(workspace, sort_bytes, n_rows, false, &s);
    if (rc != GCN_OK) return rc;
    int64_t* rest_rank = (int64_t*)((char*)workspace + sort_bytes);
    degree_keys_kernel<<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, indptr, s.keys_a);
    GCN_LAUNCH_CHECK();
    rocprim::double_buffer<uint64_t> kd(s.keys_a, s.keys_b);
    size_t tb = s.prim_bytes;
    GCN_HIP_TRY(rocprim::radix_sort_keys(s.prim, tb, kd, (size_t)n_rows, 0u, 64u, st));
    hub_marks_kernel<<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, n_hubs, kd.current(), new_id_out, s.pos);
    GCN_LAUNCH_CHECK();
    tb = s.prim_bytes;
    GCN_HIP_TRY(rocprim::exclusive_scan(s.prim, tb, s.pos, rest_rank, (int64_t)0, (size_t)n_rows,
                                        rocprim::plus<int64_t>(), st));
    hub_finish_kernel<<<grid_for(n_rows), kBlock, 0, st>>>(n_rows, n_hubs, s.pos, rest_rank, new_id_out, old_id_out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

int gcn_csr_relabel_bound(int64_t n_rows, int64_t nnz, int valued, size_t* workspace_bytes) {
    if (n_rows < 0 || nnz < 0 || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    *workspace_bytes = sort_scratch_bytes(nnz, valued != 0);
    return GCN_OK;
}

int gcn_csr_relabel(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                    const double* vals, const int32_t* new_row, const int32_t* new_col, int64_t* indptr_out,
                    int32_t* indices_out, double* vals_out, int64_t* nnz_out, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_rows > INT32_MAX || n_cols > INT32_MAX || !indptr || !indptr_out ||
        !nnz_out || (nnz > 0 && (!indices || !indices_out)))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if ((vals == nullptr) != (vals_out == nullptr)) return fail(GCN_ERR_BAD_ARG, "vals and vals_out go together");
    hipStream_t st = (hipStream_t)stream;
    const bool valued = vals != nullptr;
    SortScratch s;
    int rc = carve_scratch(workspace, workspace_bytes, nnz, valued, &s);
    if (rc != GCN_OK) return rc;
    // the new labels become row numbers and column indices, the old column indices address new_col:
    // all three must lie inside the shape (one pass checks them)
    {
        const RangeCheck c[3] = {{new_row, n_rows, n_rows}, {new_col, n_cols, n_cols}, {indices, nnz, n_cols}};
        int bits = 0;
        rc = check_ranges(c, st, &bits);
        if (rc != GCN_OK) return rc;
        if (bits & 1) return fail(GCN_ERR_INDEX_RANGE, "a row index is outside [0, %lld)", (long long)n_rows);
        if (bits & 6) return fail(GCN_ERR_INDEX_RANGE, "a column index is outside [0, %lld)", (long long)n_cols);
    }
    if (nnz > 0) {
        relabel_keys_kernel<<<grid_for(nnz), kBlock, 0, st>>>(n_rows, nnz, indptr, indices, new_row, new_col, vals,
                                                             s.keys_a, valued ? s.vals_a : nullptr);
        GCN_LAUNCH_CHECK();
    }
    const unsigned key_bits = 32 + bits_for((uint64_t)(n_rows > 0 ? n_rows - 1 : 0));
    return sorted_merge_to_csr(&s, nnz, valued, valued ? 0 : 2, key_bits, n_rows, indptr_out, indices_out, vals_out,
                               nnz_out, st);
}

int gcn_check_indices(int64_t nnz, const int32_t* indices, int64_t n_cols, void* stream) {
    if (nnz < 0 || (nnz > 0 && !indices)) return fail(GCN_ERR_BAD_ARG, "bad argument");
    return check_coo_range(nnz, nullptr, 0, indices, n_cols, (hipStream_t)stream);
}

int gcn_rmat_csr_bound(int scale, int edge_factor, int flags, int64_t* nnz_capacity, size_t* workspace_bytes) {
    if (scale < 1 || scale > 30 || edge_factor < 1 || !nnz_capacity || !workspace_bytes)
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    const int64_t m = 2 * ((int64_t)edge_factor << scale) + ((flags & GCN_ADD_SELF_LOOPS) ? ((int64_t)1 << scale) : 0);
    *nnz_capacity = m;
    *workspace_bytes = sort_scratch_bytes(m, false);
    return GCN_OK;
}

int gcn_rmat_csr(int scale, int edge_factor, uint64_t seed, int flags, int64_t* indptr_out, int32_t* indices_out,
                 int64_t* nnz_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (scale < 1 || scale > 30 || edge_factor < 1 || !indptr_out || !indices_out || !nnz_out)
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (flags & ~(GCN_ADD_SELF_LOOPS | GCN_RMAT_NATURAL_ORDER))
        return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_pairs = (int64_t)edge_factor << scale;
    const int64_t n_loops = (flags & GCN_ADD_SELF_LOOPS) ? ((int64_t)1 << scale) : 0;
    const int64_t m = 2 * n_pairs + n_loops;
    SortScratch s;
    int rc = carve_scratch(workspace, workspace_bytes, m, false, &s);
    if (rc != GCN_OK) return rc;
    const uint32_t ta = (uint32_t)(0.57 * 4294967296.0);
    const uint32_t tb = (uint32_t)((0.57 + 0.19) * 4294967296.0);
    const uint32_t tc = (uint32_t)((0.57 + 0.19 + 0.19) * 4294967296.0);
    rmat_keys_kernel<<<grid_for(n_pairs + n_loops), kBlock, 0, st>>>(scale, n_pairs, n_loops, stream_key(seed, 0),
                                                                     stream_key(seed, 1),
                                                                     (flags & GCN_RMAT_NATURAL_ORDER) ? 0 : 1, ta, tb,
                                                                     tc, s.keys_a);
    GCN_LAUNCH_CHECK();
    // invalid keys are all ones and must sort last: sort on all 64 bits
    return sorted_merge_to_csr(&s, m, false, 2, 64u, (int64_t)1 << scale, indptr_out, indices_out, nullptr, nnz_out,
                               st);
}

int gcn_fill_synthetic_f32(int64_t count, uint64_t seed, float* out, void* stream) {
    return gcn_fill_synthetic_range_f32(0, count, seed, out, stream);
}

int gcn_fill_synthetic_range_f32(int64_t first, int64_t count, uint64_t seed, float* out, void* stream) {
    if (first < 0 || count < 0 || (count > 0 && !out)) return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (count == 0) return GCN_OK;
    fill_synthetic_kernel<<<grid_for(count), kBlock, 0, (hipStream_t)stream>>>(count, stream_key(seed, 0),
                                                                                (uint64_t)first, out);
    GCN_LAUNCH_CHECK();
    return GCN_OK;
}

static int rmat_rows_args(int scale, int edge_factor, int flags, int64_t row_begin, int64_t row_end) {
    if (scale < 1 || scale > 30 || edge_factor < 1) return fail(GCN_ERR_BAD_ARG, "bad argument");
    if (flags & ~(GCN_ADD_SELF_LOOPS | GCN_RMAT_NATURAL_ORDER))
        return fail(GCN_ERR_BAD_ARG, "unknown flag bits 0x%x", flags);
    if (row_begin < 0 || row_end < row_begin || row_end > ((int64_t)1 << scale))
        return fail(GCN_ERR_BAD_ARG, "bad row range [%lld, %lld)", (long long)row_begin, (long long)row_end);
    return GCN_OK;
}

static void rmat_thresholds(uint32_t* ta, uint32_t* tb, uint32_t* tc) {
    *ta = (uint32_t)(0.57 * 4294967296.0);
    *tb = (uint32_t)((0.57 + 0.19) * 4294967296.0);
    *tc = (uint32_t)((0.57 + 0.19 + 0.19) * 4294967296.0);
}

int gcn_rmat_rows_count(int scale, int edge_factor, uint64_t seed, int flags, int64_t row_begin, int64_t row_end,
                        int64_t* n_keys_host, void* stream) {
    int rc = rmat_rows_args(scale, edge_factor, flags, row_begin, row_end);
    if (rc != GCN_OK) return rc;
    if (!n_keys_host) return fail(GCN_ERR_BAD_ARG, "n_keys_host is null");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_pairs = (int64_t)edge_factor << scale;
    uint32_t ta, tb, tc;
    rmat_thresholds(&ta, &tb, &tc);
    unsigned long long* counter = nullptr;
    GCN_HIP_TRY(hipMalloc(&counter, sizeof(unsigned long long)));
    unsigned long long host = 0;
    hipError_t e = hipMemsetAsync(counter, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        rmat_rows_keys_kernel<<<grid_for(n_pairs), kBlock, 0, st>>>(
            scale, n_pairs, stream_key(seed, 0), stream_key(seed, 1), (flags & GCN_RMAT_NATURAL_ORDER) ? 0 : 1, ta, tb,
            tc, (uint64_t)row_begin, (uint64_t)row_end, counter, 0, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, counter, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(counter);
    if (e != hipSuccess) return fail(GCN_ERR_HIP, "counting the keys failed: %s", hipGetErrorString(e));
    *n_keys_host = (int64_t)host + ((flags & GCN_ADD_SELF_LOOPS) ? (row_end - row_begin) : 0);
    return GCN_OK;
}

int gcn_rmat_csr_rows_bound(int64_t n_keys, size_t* workspace_bytes) {
    if (n_keys < 0 || !workspace_bytes) return fail(GCN_ERR_BAD_ARG, "bad argument");
    *workspace_bytes = sort_scratch_bytes(n_keys, false) + 256;
    return GCN_OK;
}

int gcn_rmat_csr_rows(int scale, int edge_factor, uint64_t seed, int flags, int64_t row_begin, int64_t row_end,
                      int64_t n_keys, int64_t* indptr_out, int32_t* indices_out, int64_t* nnz_out, void* workspace,
                      size_t workspace_bytes, void* stream) {
    int rc = rmat_rows_args(scale, edge_factor, flags, row_begin, row_end);
    if (rc != GCN_OK) return rc;
    if (n_keys < 0 || !indptr_out || !nnz_out || (n_keys > 0 && !indices_out))
        return fail(GCN_ERR_BAD_ARG, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_local = row_end - row_begin;
    const int64_t n_loops = (flags & GCN_ADD_SELF_LOOPS) ? n_local : 0;
    if (n_keys < n_loops) return fail(GCN_ERR_BAD_ARG, "n_keys is smaller than the number of diagonal entries");
    size_t need = 0;
    gcn_rmat_csr_rows_bound(n_keys, &need);
    if (!workspace || workspace_bytes < need)
        return fail(GCN_ERR_WORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) != 0) return fail(GCN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    unsigned long long* counter = (unsigned long long*)workspace;       // the first 256 bytes
    SortScratch s;
    rc = carve_scratch((char*)workspace + 256, workspace_bytes - 256, n_keys, false, &s);
    if (rc != GCN_OK) return rc;
    GCN_HIP_TRY(hipMemsetAsync(counter, 0, 256, st));
    if (n_loops > 0) {
        diagonal_keys_kernel<<<grid_for(n_loops), kBlock, 0, st>>>(n_loops, (uint64_t)row_begin, s.keys_a);
        GCN_LAUNCH_CHECK();
    }
    const int64_t n_pairs = (int64_t)edge_factor << scale;
    const int64_t capacity = n_keys - n_loops;
    uint32_t ta, tb, tc;
    rmat_thresholds(&ta, &tb, &tc);
    rmat_rows_keys_kernel<<<grid_for(n_pairs), kBlock, 0, st>>>(
        scale, n_pairs, stream_key(seed, 0), stream_key(seed, 1), (flags & GCN_RMAT_NATURAL_ORDER) ? 0 : 1, ta, tb, tc,
        (uint64_t)row_begin, (uint64_t)row_end, counter, capacity, s.keys_a + n_loops);
    GCN_LAUNCH_CHECK();
    // n_keys must be the count gcn_rmat_rows_count returned for these arguments: every slot is then filled
    unsigned long long emitted = 0;
    GCN_HIP_TRY(hipMemcpyAsync(&emitted, counter, sizeof(emitted), hipMemcpyDeviceToHost, st));
    GCN_HIP_TRY(hipStreamSynchronize(st));
    if ((int64_t)emitted != capacity)
        return fail(GCN_ERR_BAD_ARG, "n_keys=%lld does not match the %lld keys of this row range", (long long)n_keys,
                    (long long)emitted + (long long)n_loops);
    const unsigned key_bits = 32 + bits_for((uint64_t)(n_local > 0 ? n_local - 1 : 0));
    return sorted_merge_to_csr(&s, n_keys, false, 2, key_bits, n_local, indptr_out, indices_out, nullptr, nnz_out, st);
}

}  // extern "C"
