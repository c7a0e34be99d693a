/* gcn_hip.h -- C ABI of libgcn_hip.so: the GCN propagation path A_hat . (X . W)
 * as HIP kernels for gfx950 (MI355X).
 *
 * The reference (hareeshks/gcn) has no native interface: its path is Python
 * calling TensorFlow / scipy.  Each entry point below names the reference call
 * site(s) whose arithmetic it replaces; the Python package gcn_amd binds them
 * with ctypes under the reference's own function names.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host";
 *  - the caller owns every buffer; the library allocates only inside
 *    gcn_spmm_plan_create (plan metadata, freed by gcn_spmm_plan_destroy);
 *  - `stream` is a hipStream_t passed as void*; every launch is asynchronous
 *    on it; only the functions marked [sync] wait for the stream;
 *  - CSR: indptr int64 [n_rows+1], indices int32, values float32 (float64 where
 *    stated); dense matrices are row-major float32 with a leading dimension in
 *    elements;
 *  - every function returns a gcn_status; gcn_last_error_string() describes the
 *    last failure on the calling thread.  No C++ exception crosses this ABI.
 */
#ifndef GCN_HIP_H_
#define GCN_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GCN_OK = 0,
    GCN_ERR_BAD_ARG = 1,      /* Python: ValueError   (reference: ValueError at train.py:141, utils.py:293) */
    GCN_ERR_INDEX_RANGE = 2,  /* Python: ValueError   (TensorFlow: InvalidArgumentError for out-of-range sparse indices) */
    GCN_ERR_HIP = 3,          /* Python: RuntimeError */
    GCN_ERR_WORKSPACE = 4,    /* Python: ValueError   (workspace too small) */
    GCN_ERR_RCCL = 5          /* Python: RuntimeError (RCCL missing or a collective failed) */
} gcn_status;

/* flags of gcn_coo_to_csr */
#define GCN_ADD_SELF_LOOPS 1  /* adj + sp.eye(n): ADD 1.0 on the diagonal            (gcn/utils.py:321) */
#define GCN_BINARY 2          /* duplicates collapse to one entry of value 1 (simple graph, no values out) */
#define GCN_RMAT_NATURAL_ORDER 8 /* gcn_rmat_csr: keep the generator's own vertex ids (hubs at the ids with few one-bits) */
/* flags of the SpMM / GEMM entry points */
#define GCN_RELU 1            /* act = relu fused into the epilogue                  (gcn/layers.py:214, models.py:17) */
#define GCN_SPLIT_WIDE 4      /* spmm: cut long rows every 1024 entries whatever the width, so that a product
                                 computed over ranges of feature columns equals the full-width one bit for bit */
/* type of gcn_normalize_adj */
#define GCN_NORM_SYM 0        /* D^-1/2 A D^-1/2                                     (gcn/utils.py:299-308) */
#define GCN_NORM_RW 1         /* D^-1 A                                              (gcn/utils.py:309-315) */

int gcn_version(void);
const char* gcn_last_error_string(void);
/* sha256 (hex) of the concatenation, in byte order of the file names, of gcn_amd/csrc/{*.cpp,*.h,*.hip}
 * followed by include/gcn_hip.h, taken when the library was built: tells a stale prebuilt library from
 * one built from the sources beside it (gcn_amd._lib.source_hash() computes the same over the files on
 * disk).  No counterpart in the reference, which has nothing to build. */
const char* gcn_build_hash(void);

/* ------------------------------------------------------------------------- *
 * Graph preprocessing -- integer work, bit-exact.                            *
 * ------------------------------------------------------------------------- */

/* Capacity and scratch needed by gcn_coo_to_csr.  Host outputs. */
int gcn_coo_to_csr_bound(int64_t n_rows, int64_t nnz_in, int flags, int64_t* nnz_capacity, size_t* workspace_bytes);

/* COO (any order, duplicates allowed) -> CSR with ascending columns and
 * duplicates summed; with GCN_ADD_SELF_LOOPS one extra (i,i,1.0) per row is
 * merged in.  Replaces the scipy conversions behind `adj + sp.eye(adj.shape[0])`
 * and `sp.coo_matrix(adj)` at gcn/utils.py:300,321.
 * vals may be NULL (all ones).  vals_out (float64) may be NULL with GCN_BINARY.
 * indices_out / vals_out must hold nnz_capacity entries; *nnz_out (device
 * int64) receives the merged count.  A row or column index outside the shape is
 * GCN_ERR_INDEX_RANGE (scipy's coo_matrix raises ValueError for it) and nothing
 * is written.  [sync] */
int gcn_coo_to_csr(int64_t n_rows, int64_t n_cols, int64_t nnz_in, const int32_t* rows, const int32_t* cols,
                   const double* vals, int flags, int64_t* indptr_out, int32_t* indices_out, double* vals_out,
                   int64_t* nnz_out, void* workspace, size_t workspace_bytes, void* stream);

/* indptr from a non-decreasing row array: CSR view of a row-grouped triple
 * that keeps the triple's own order inside every row (the feature triple of
 * sparse_to_tuple, gcn/utils.py:258-275 as fed at train.py:118).  A row index
 * outside [0, n_rows) is GCN_ERR_INDEX_RANGE.  [sync] */
int gcn_rows_to_indptr(int64_t n_rows, int64_t nnz, const int32_t* rows, int64_t* indptr_out, void* stream);

/* rowsum = adj.sum(1) (gcn/utils.py:301, 310).  vals NULL = all ones. float64 out. */
int gcn_row_sums(int64_t n_rows, const int64_t* indptr, const double* vals, double* rowsum_out, void* stream);

/* d = rowsum^-0.5 (GCN_NORM_SYM, gcn/utils.py:305-306) or rowsum^-1
 * (GCN_NORM_RW, utils.py:311-312), float64, inf -> 0. */
int gcn_degree_scale(int64_t n_rows, const double* rowsum, int type, double* d_out, void* stream);

/* normalize_adj (gcn/utils.py:297-315) on CSR storage, d from gcn_degree_scale.
 * For entry k stored at (r, c):  SYM: w[k] = (vals[k] * d[c]) * d[r], which is
 * the value of result[c, r] -- the element order of the triple the reference
 * returns (column-major); for a symmetric adj it is also the CSR of the result.
 * RW: w[k] = d[r] * vals[k].  float64 arithmetic; vals NULL = all ones;
 * vals64_out and/or vals32_out (the float32 cast made by the float32
 * placeholder at train.py:145) may be NULL. */
int gcn_normalize_adj(int64_t n_rows, int64_t nnz, const int64_t* indptr, const int32_t* indices, const double* vals,
                      const double* d, int type, double* vals64_out, float* vals32_out, void* stream);

int gcn_csr_transpose_bound(int64_t n_rows, int64_t n_cols, int64_t nnz, size_t* workspace_bytes);

/* Stable CSR transpose (source rows ascend inside every output row): the
 * adjoint operand of the backward products A_hat^T.dZ and X'^T.dP that
 * `minimize` builds at gcn/models.py:86.  vals: val_bytes 0 (none), 4 or 8.
 * perm_out (int64, may be NULL): source position of every output entry. */
int gcn_csr_transpose(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                      const void* vals, int val_bytes, int64_t* t_indptr, int32_t* t_indices, void* t_vals,
                      int64_t* perm_out, void* workspace, size_t workspace_bytes, void* stream);

/* Vertex order by descending degree (row length), ties by ascending id: new_id_out[v] = position of
 * vertex v in that order, old_id_out[i] = the vertex at position i (either may be NULL).  An integer
 * preprocessing step in front of preprocess_adj (gcn/utils.py:318-323); the reference keeps the
 * dataset's order.  After it the hub rows of the dense operand are its first rows -- one contiguous,
 * cache-resident block -- and rows of similar length sit next to each other. */
int gcn_degree_order_bound(int64_t n_rows, size_t* workspace_bytes);
int gcn_degree_order(int64_t n_rows, const int64_t* indptr, int32_t* new_id_out, int32_t* old_id_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Hub-first vertex order: the n_hubs vertices of highest degree take the ids 0 .. n_hubs-1 in the order of
 * gcn_degree_order, every other vertex follows in its old order.  The hub rows of the dense operand then
 * form one contiguous block (narrow rows of hubs share cache lines) while the rest of the graph keeps the
 * mix of row lengths it had -- measured on the 2^24-node R-MAT graph with 65,536 hubs: products 3-6 %
 * faster at every width from 16 to 256, where the full degree order gains at narrow widths and loses
 * at wide ones.  new_id_out is required, old_id_out may be NULL.  Same role as gcn_degree_order: an
 * integer preprocessing step in front of preprocess_adj (gcn/utils.py:318-323), which keeps the dataset's
 * order. */
int gcn_hub_order_bound(int64_t n_rows, size_t* workspace_bytes);
int gcn_hub_order(int64_t n_rows, const int64_t* indptr, int64_t n_hubs, int32_t* new_id_out, int32_t* old_id_out,
                  void* workspace, size_t workspace_bytes, void* stream);

/* CSR of P_r . A . P_c^T: entry (r, c) moves to (new_row[r], new_col[c]); columns ascend inside every
 * output row.  new_row / new_col NULL = identity; labels outside the shape are GCN_ERR_INDEX_RANGE.
 * vals (float64, with vals_out) may be NULL for a structure-only matrix.  With new_row == new_col of a
 * symmetric adjacency this is the relabelled graph of gcn_degree_order; products on it equal the
 * original ones with the rows of B and C permuted the same way (gcn_gather_rows_f32), up to the order
 * of the additions inside a row.  [sync] */
int gcn_csr_relabel_bound(int64_t n_rows, int64_t nnz, int valued, size_t* workspace_bytes);
int gcn_csr_relabel(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, const int32_t* indices,
                    const double* vals, const int32_t* new_row, const int32_t* new_col, int64_t* indptr_out,
                    int32_t* indices_out, double* vals_out, int64_t* nnz_out, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Range check of CSR column indices: GCN_ERR_INDEX_RANGE if any is outside
 * [0, n_cols) -- the check TensorFlow's kernel makes per nonzero.  [sync] */
int gcn_check_indices(int64_t nnz, const int32_t* indices, int64_t n_cols, void* stream);

/* ------------------------------------------------------------------------- *
 * Synthetic inputs for measurement (the reference has no generator).         *
 * ------------------------------------------------------------------------- */

int gcn_rmat_csr_bound(int scale, int edge_factor, int flags, int64_t* nnz_capacity, size_t* workspace_bytes);

/* R-MAT (a,b,c,d)=(0.57,0.19,0.19,0.05), edge_factor*2^scale generated pairs,
 * vertex ids scrambled by a fixed bijection (so that contiguous row blocks carry
 * equal work; GCN_RMAT_NATURAL_ORDER leaves them as generated), self-loops dropped, symmetrised, deduplicated -> CSR structure of
 * a simple undirected graph (the kind nx.from_dict_of_lists builds at
 * gcn/utils.py:123).  flags: GCN_ADD_SELF_LOOPS also inserts every (i, i), giving
 * the structure of A + I (all values 1).  Integer-only, counter-based: identical
 * to the CPU oracle bit for bit. */
int gcn_rmat_csr(int scale, int edge_factor, uint64_t seed, int flags, int64_t* indptr_out, int32_t* indices_out,
                 int64_t* nnz_out, void* workspace, size_t workspace_bytes, void* stream);

/* The rows [row_begin, row_end) of the same graph without building the rest: what one rank of a
 * row-partitioned run holds (column indices stay global; indptr_out has row_end - row_begin + 1
 * entries and starts at 0).  Equal, bit for bit, to that slice of gcn_rmat_csr's output.
 *   1. gcn_rmat_rows_count   -> *n_keys_host (host): generated keys of the range, before deduplication [sync]
 *   2. gcn_rmat_csr_rows_bound(n_keys) -> workspace bytes; indices_out must hold n_keys entries
 *   3. gcn_rmat_csr_rows(..., n_keys, ...)  [sync] */
int gcn_rmat_rows_count(int scale, int edge_factor, uint64_t seed, int flags, int64_t row_begin, int64_t row_end,
                        int64_t* n_keys_host, void* stream);
int gcn_rmat_csr_rows_bound(int64_t n_keys, size_t* workspace_bytes);
int gcn_rmat_csr_rows(int scale, int edge_factor, uint64_t seed, int flags, int64_t row_begin, int64_t row_end,
                      int64_t n_keys, int64_t* indptr_out, int32_t* indices_out, int64_t* nnz_out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* out[i] = (int32(h >> 32) >> 8) * 2^-23 in [-1,1), h = splitmix(key(seed) ^ i). */
int gcn_fill_synthetic_f32(int64_t count, uint64_t seed, float* out, void* stream);
/* The elements first .. first + count - 1 of that sequence (a row block of a larger matrix). */
int gcn_fill_synthetic_range_f32(int64_t first, int64_t count, uint64_t seed, float* out, void* stream);

/* ------------------------------------------------------------------------- *
 * Sparse x dense products -- the hot path.                                   *
 * ------------------------------------------------------------------------- */

typedef struct gcn_spmm_plan gcn_spmm_plan;

/* Row-length analysis of one CSR structure: work-balanced strips for ordinary
 * rows, chunk tables for rows longer than the split thresholds.  [sync] */
int gcn_spmm_plan_create(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, gcn_spmm_plan** plan_out,
                         void* stream);

/* The same with the split threshold chosen by the caller: rows longer than split_entries (16..1024;
 * 0 = the defaults, 256 / 1024 entries for narrow / wide dense operands) are summed in chunks.  For
 * the operands of the gradient products that optimizer.minimize derives (gcn/models.py:86), e.g.
 * X^T of a citation graph, whose rows are long and few: a lower threshold trades the serial order of
 * the additions in rows between the two thresholds (results still within 1e-5) for parallelism.
 * [sync] */
int gcn_spmm_plan_create_split(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* indptr, int64_t split_entries,
                               gcn_spmm_plan** plan_out, void* stream);
int gcn_spmm_plan_destroy(gcn_spmm_plan* plan);
/* Cache hint for products with this plan (wide dense operands, one wave per row): rows of B whose
 * column index is >= hot_cols are loaded with the non-temporal hint, the first hot_cols rows with the
 * default policy.  Meant for a matrix relabelled by gcn_degree_order, whose hub columns come first; 0
 * (default) turns the hint off.  Results do not change.  No counterpart in the reference (its kernel,
 * gcn/layers.py:31, runs on a CPU). */
int gcn_spmm_plan_set_hot_columns(gcn_spmm_plan* plan, int64_t hot_cols);
/* LDS-staged hub rows for narrow products (f <= 64, f a multiple of 4) with this plan: every workgroup
 * copies the first min(hub_rows, 32 KiB / (4 f)) rows of B into LDS and entries whose column index is below
 * that number read their row from LDS instead of memory.  Meant for a matrix relabelled by
 * gcn_degree_order, whose first columns are the hubs (the rows selected by degree); 0 (default): off.
 * Same additions in the same order: results do not change.  No counterpart in the reference. */
int gcn_spmm_plan_set_lds_hub_rows(gcn_spmm_plan* plan, int64_t hub_rows);
/* host-side facts about a plan: what[0]=n_strips, [1]=rows longer than the split threshold, [2]=their
 * chunks, [3]=the split threshold (the smallest one: rows up to it are summed serially for every width) */
int gcn_spmm_plan_info(const gcn_spmm_plan* plan, int64_t* what4);
int gcn_spmm_plan_workspace_bytes(const gcn_spmm_plan* plan, int64_t f, size_t* workspace_bytes);
/* Measurement hook (the reference times its step with time.time(), train.py:203-205): when enabled,
 * every product on this plan records HIP events on its stream around its kernels;
 * gcn_spmm_plan_last_timing waits for the last product and returns ms2[0] = the row kernel,
 * ms2[1] = the long-row reduction.  [sync] */
int gcn_spmm_plan_set_timing(gcn_spmm_plan* plan, int enable);
int gcn_spmm_plan_last_timing(const gcn_spmm_plan* plan, float* ms2);

/* C[n_rows x f] = A . B, A in CSR, B [n_cols x f]: `dot(x, y, sparse=True)`
 * = tf.sparse_tensor_dense_matmul at gcn/layers.py:28-31, called at
 * layers.py:202 (A_hat . (XW)) and layers.py:198-199 (X . W0).  Every output row
 * up to the split threshold (256 nonzeros when f <= 64, 1024 for wider operands or with
 * GCN_SPLIT_WIDE) is accumulated sequentially in storage order with a separate float32
 * multiply and add -- the order and the bits of the serial CPU loop; a longer row is summed
 * in chunks of that many entries whose partial rows are then added in chunk order. */
int gcn_spmm_csr_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                     int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc, int flags, void* workspace,
                     size_t workspace_bytes, void* stream);

/* gcn_spmm_csr_f32 for callers that cannot allocate or pass a workspace.  It has no counterpart in the
 * reference; it is for code under HIP graph capture, tight serving loops and harnesses that forbid
 * hipMalloc.  The partial rows of long rows go to a 32 MiB array inside the library (one per GPU, held
 * until the process ends) instead of a caller workspace; the call makes no hipMalloc / hipFree and no
 * allocating library call.  Same kernels, same bits as gcn_spmm_csr_f32.
 *  - a product whose gcn_spmm_plan_workspace_bytes(plan, f) exceeds 32 MiB is GCN_ERR_WORKSPACE;
 *  - the plan must have been created on the current device (plan creation resolves the array's
 *    address there);
 *  - every call on a GPU shares that GPU's array: two calls that overlap on one GPU (different streams
 *    or threads) race on it -- order them.  Nothing in it needs to start at zero: a call writes every
 *    partial row before it reads it. */
int gcn_spmm_csr_f32_noalloc(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                             const float* vals, int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc,
                             int flags, void* stream);

/* gcn_spmm_csr_f32 (optionally with the store dropout of gcn_spmm_csr_out_dropout_f32: keep_prob < 1) for a
 * dense operand stored in TWO parts: its rows [0, n_first) at b_first (row pitch ld_first), the rows from n_first
 * on at b_rest (row pitch ld_rest; b_rest[0] is row n_first).  The rank of a row-partitioned product reads its own
 * block of the per-layer matrix (`pre_sup`, gcn/layers.py:198-202) where it lies and only the rows received from
 * the other ranks from the exchange buffer, so the own rows are never packed or copied; the single-process reference
 * has no counterpart.  Same additions in the same order as the product on the stacked operand: the same bits.
 * Needs 16-byte aligned rows in both parts, f a multiple of 4 and a plan whose matrix has at least ~2M entries
 * (the instances for small matrices have no two-part form): otherwise GCN_ERR_BAD_ARG, and the caller stacks the
 * operand. */
int gcn_spmm_csr_two_part_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                              const float* vals, int64_t f, const float* b_first, int64_t ld_first, int64_t n_first,
                              const float* b_rest, int64_t ld_rest, float* c, int64_t ldc, int flags, float keep_prob,
                              uint64_t seed, uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset,
                              int64_t pitch, void* workspace, size_t workspace_bytes, void* stream);

/* C = A . B + D: one smoothing / propagation step with the residual added in the epilogue
 * (`new_feature = adj * new_feature + features`, gcn/utils.py:568-569; float32 here).  D is a
 * dense [n_rows x f] matrix; GCN_RELU applies after the addition.  C may be D itself (every element is
 * read and then written by the same lane). */
int gcn_spmm_csr_add_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                         int64_t f, const float* b, int64_t ldb, const float* add, int64_t ldadd, float* c,
                         int64_t ldc, int flags, void* workspace, size_t workspace_bytes, void* stream);

/* Same product with sparse_dropout (gcn/layers.py:19-25) fused on A's entries:
 * entry k is kept iff floor(keep_prob + u_k) == 1 in float32 and then scaled by
 * float32(1/keep_prob); u_k is the counter-based uniform of entry id
 * (perm ? perm[k] : k) in stream (seed, stream_id).  perm lets a transposed
 * copy of X reproduce the mask of the original feed order.  stream_offset (device pointer, may be
 * NULL): the 64-bit value it points to is added to stream_id when the kernel runs, so that a launch
 * recorded in a HIP graph draws a new mask on every replay (the caller updates the value between
 * replays, as sess.run draws new noise on every call). */
int gcn_spmm_csr_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                             const float* vals, const int64_t* perm, int64_t f, const float* b, int64_t ldb, float* c,
                             int64_t ldc, int flags, float keep_prob, uint64_t seed, uint64_t stream_id,
                             const uint64_t* stream_offset, void* workspace, size_t workspace_bytes, void* stream);

/* C = dropout(act(A . B)): the product of one layer (gcn/layers.py:202-208) with the NEXT layer's
 * tf.nn.dropout on its input (gcn/layers.py:185, reached through gcn/models.py:56-66) applied in the
 * store, so the activations are written once, already dropped.  Element (r, c) draws the decision
 * of id index_offset + r * pitch + c in stream (seed, stream_id): what gcn_dropout_f32 draws for a
 * matrix `pitch` columns wide.  C may be a row block of that matrix (index_offset = first row *
 * pitch) and a column range of it (index_offset += first column; pitch >= f).  The dropped output
 * is also the gate of the backward pass: it is positive exactly where the unit was active and
 * kept (see gcn_gemm_gated_f32).  stream_offset: as for gcn_spmm_csr_dropout_f32. */
int gcn_spmm_csr_out_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                                 const float* vals, int64_t f, const float* b, int64_t ldb, float* c, int64_t ldc,
                                 int flags, float keep_prob, uint64_t seed, uint64_t stream_id,
                                 const uint64_t* stream_offset, int64_t index_offset, int64_t pitch, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* C = dropout(act(A . B + D)): gcn_spmm_csr_add_f32 with the store dropout of gcn_spmm_csr_out_dropout_f32
 * (same decision ids).  A row-partitioned layer product (gcn/layers.py:202-208) that multiplies the entries
 * reading the rank's own rows while the other ranks' rows are still on the links finishes with this call:
 * D is that first partial product, A the entries that read received rows, and the activation and the next
 * layer's dropout (gcn/layers.py:185) are applied once, in this store.  C may be D itself: every element is read
 * and then written by the same lane.  The single-process reference has no counterpart. */
int gcn_spmm_csr_add_out_dropout_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices,
                                     const float* vals, int64_t f, const float* b, int64_t ldb, const float* add,
                                     int64_t ldadd, float* c, int64_t ldc, int flags, float keep_prob, uint64_t seed,
                                     uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset,
                                     int64_t pitch, void* workspace, size_t workspace_bytes, void* stream);

/* Packed dense operand.  The matrices a GCN training step multiplies a support with are sparse themselves:
 * dropout(X) keeps keep_prob of its elements (gcn/layers.py:185 feeding 198-202), and the gradient the backward
 * product propagates (the adjoint of gcn/layers.py:202, derived by gcn/models.py:86) is zero wherever the unit was
 * dropped or inactive.  A packed row of f <= 256 columns is eight {mask word, kept columns in the words before it}
 * pairs followed by the kept values in column order, at a pitch of gcn_pack_pitch(f) floats; a product that gathers
 * such rows moves 64 + 4 k bytes per row with k kept values instead of 4 f.  The reference has no counterpart (it
 * multiplies the dense matrices).
 *
 * gcn_pack_pitch: the row pitch, in floats, of the packed form of an f-column matrix (a multiple of 32). */
int gcn_pack_pitch(int64_t f, int64_t* pitch_floats);

/* packed <- the nonzero elements of x [n_rows x f] (ldx >= f).  `packed` holds n_rows * ldpk floats, ldpk a multiple of
 * 32 not below gcn_pack_pitch(f), 128-byte aligned. */
int gcn_pack_rows_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float* packed, int64_t ldpk, void* stream);

/* packed <- tf.nn.dropout(x, keep_prob) (gcn/layers.py:185) without ever writing the dense result: element (r, c) is
 * kept iff decision index_offset + r * pitch + c of stream (seed, stream_id) says so and is then x / keep_prob -- the
 * values and decisions of gcn_dropout_f32.  stream_offset: as for gcn_spmm_csr_dropout_f32. */
int gcn_dropout_pack_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float keep_prob, uint64_t seed,
                         uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                         float* packed, int64_t ldpk, void* stream);

/* gcn_dropout_pack_f32 for rows that are not consecutive rows of the matrix the decisions are drawn for: row r of x is
 * row row_ids[r] of that matrix (decision index_offset + row_ids[r] * pitch + c); row_ids null: r itself.  A rank of a
 * row-partitioned run holds rows of the constant feature matrix (gcn/layers.py:185 drops them anew every step) that
 * belong to other ranks and drops them as their owner would.  No counterpart in the reference (one process). */
int gcn_dropout_pack_rows_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, float keep_prob, uint64_t seed,
                              uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                              const int64_t* row_ids, float* packed, int64_t ldpk, void* stream);

/* packed <- where(gate > 0, (A . B^T) / divisor, 0) for A [m x k], B [n x k] (n <= 256, k <= 16), gate [m x n]: the
 * gradient of a hidden layer's pre-activation -- gcn_gemm_gated_f32 with trans_b, the backward of tf.matmul
 * (gcn/layers.py:32-33) with the backward of tf.nn.dropout (gcn/layers.py:185) and of the ReLU (gcn/layers.py:208) --
 * written in packed form for the backward product (gcn_spmm_csr_packed_f32); the dense gradient is never stored.
 * Each element is summed over k in ascending order with fused multiply-adds. */
int gcn_gemm_gated_pack_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                            const float* gate, int64_t ldg, float divisor, float* packed, int64_t ldpk, void* stream);

/* gcn_gemm_gated_pack_f32 for a gradient known to be zero outside some rows: `rows` (ascending, n_listed of them) are
 * the rows of A that can hold a nonzero; every row of `packed` is first made empty (64 bytes), then the listed rows are
 * formed from their rows of A and of the gate.  The gradient that reaches a hidden layer through the backward product
 * of a masked loss (gcn/metrics.py:85-92) is non-zero at the neighbours of labelled nodes only.  A row number outside
 * [0, m) is skipped.  rows null (n_listed 0): every row, as gcn_gemm_gated_pack_f32.  No counterpart in the
 * reference. */
int gcn_gemm_gated_pack_rows_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b,
                                 int64_t ldb, const float* gate, int64_t ldg, float divisor, const int64_t* rows,
                                 int64_t n_listed, float* packed, int64_t ldpk, void* stream);

/* y [n_rows x f] <- the dense form of a packed matrix (zeros where nothing was kept). */
int gcn_unpack_rows_f32(int64_t n_rows, int64_t f, const float* packed, int64_t ldpk, float* y, int64_t ldy, void* stream);

/* C[n_rows x f] = A . B for B in packed form [n_cols rows, pitch ldpk]: gcn_spmm_csr_f32 (gcn/layers.py:202) on an
 * operand that is mostly zeros.  Same entries in the same order as the dense product, long rows cut where the wide
 * dense product cuts them: the same values (a skipped zero would have added a zero).  flags: GCN_RELU.  Workspace
 * as for gcn_spmm_csr_f32. */
int gcn_spmm_csr_packed_f32(const gcn_spmm_plan* plan, const int64_t* indptr, const int32_t* indices, const float* vals,
                            int64_t f, const float* packed_b, int64_t ldpk, float* c, int64_t ldc, int flags,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * Dense pieces of the layer.                                                 *
 * ------------------------------------------------------------------------- */

/* rowsum^-1 scaling of a CSR matrix in place of `r_mat_inv.dot(features)` -- the bag-of-words
 * normalisation of gcn/utils.py:279-285: vals_out[k] = r_inv[row(k)] * vals[k], r_inv = 1 / rowsum with
 * inf -> 0, in the value type of the matrix (val_bytes 4 or 8; the row sum itself is taken in
 * float64 as `features.sum(1)` over a float64 LIL matrix does). */
int gcn_row_normalize(int64_t n_rows, int64_t nnz, const int64_t* indptr, const void* vals, int val_bytes,
                      void* vals_out, void* stream);

/* y = (x / keep_prob) * floor(keep_prob + u): tf.nn.dropout at gcn/layers.py:185.
 * gate (may be NULL): additionally y = 0 where gate <= 0 (ReLU backward).
 * Element (r, c) draws the uniform of id index_offset + r * f + c: a rank that holds the rows
 * [r0, r1) of a larger matrix passes r0 * f and gets the mask of the unsplit matrix.
 * stream_offset: as for gcn_spmm_csr_dropout_f32. */
int gcn_dropout_f32(int64_t n_rows, int64_t f, const float* x, int64_t ldx, const float* gate, int64_t ldg, float* y,
                    int64_t ldy, float keep_prob, uint64_t seed, uint64_t stream_id, const uint64_t* stream_offset,
                    uint64_t index_offset, void* stream);

/* C[m x n] = A . op(B) in float32 on the f32 MFMA pipe: tf.matmul at
 * gcn/layers.py:32-33 (H . W).  A is [m x k]; trans_b = 0: B is [k x n];
 * trans_b = 1: B is stored [n x k] (dH = dP . W^T in the backward pass). */
int gcn_gemm_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                 int trans_b, float* c, int64_t ldc, int flags, void* stream);

/* C = gate > 0 ? (A . op(B)) / divisor : 0.  The gradient of tf.matmul w.r.t. its left operand
 * (dH = dP . W^T, gcn/layers.py:32-33 under gcn/models.py:86) with the backward of the
 * dropout(relu(.)) that produced that operand folded into the store: gate is the dropped
 * activation matrix [m x n] itself and divisor the keep probability.  Same bits as gcn_gemm_f32
 * followed by gcn_dropout_f32 with the ReLU gate. */
int gcn_gemm_gated_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       int trans_b, const float* gate, int64_t ldg, float divisor, float* c, int64_t ldc,
                       void* stream);

/* C = dropout(act(A . op(B))): tf.matmul (gcn/layers.py:32-33) with the activation and the NEXT layer's
 * tf.nn.dropout on its input (gcn/layers.py:185) in the store -- the dense counterpart of
 * gcn_spmm_csr_out_dropout_f32, with the same decision ids (index_offset + r * pitch + c) and the same
 * bits as gcn_gemm_f32 followed by gcn_dropout_f32.  row_ids (may be NULL): C holds a subset of the rows
 * of the matrix the decisions are drawn for, and row r of C draws the decisions of row row_ids[r]. */
int gcn_gemm_out_dropout_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* b, int64_t ldb,
                             int trans_b, float* c, int64_t ldc, int flags, float keep_prob, uint64_t seed,
                             uint64_t stream_id, const uint64_t* stream_offset, int64_t index_offset, int64_t pitch,
                             const int64_t* row_ids, void* stream);

/* The decisions of tf.nn.dropout (gcn/layers.py:185) on a matrix [n_rows x f] as one bit per element: bit
 * c % 32 of word bits_out[r * ld_words + c / 32] is 1 iff element (r, c) is kept.  Element (r, c) draws the
 * decision of id index_offset + r * pitch + c -- the one gcn_dropout_f32 draws (there pitch = f).  With the
 * two products below the dropped matrix is never written: the layer that consumes it reads x and the bits
 * (1/32 of the bytes), forward and backward. */
int gcn_dropout_bits(int64_t n_rows, int64_t f, float keep_prob, uint64_t seed, uint64_t stream_id,
                     const uint64_t* stream_offset, uint64_t index_offset, int64_t pitch, uint32_t* bits_out,
                     int64_t ld_words, void* stream);

/* C = dropout(A) . op(B): tf.nn.dropout on a dense layer input (gcn/layers.py:185) followed by tf.matmul
 * (gcn/layers.py:198-199, 32-33), the dropout applied to A's elements as they are loaded: element (r, kk) of
 * A becomes A[r, kk] / keep_prob if bit (r, kk) of `mask` (gcn_dropout_bits, rows of ld_words words) is set,
 * else 0.  Same bits as gcn_dropout_f32 followed by gcn_gemm_f32. */
int gcn_gemm_masked_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const uint32_t* mask,
                        int64_t ld_words, float keep_prob, const float* b, int64_t ldb, int trans_b, float* c,
                        int64_t ldc, int flags, void* stream);

/* C[k x n] = dropout(A)^T . G with the same mask: the weight gradient of that layer (gcn/models.py:86).
 * Workspace as for gcn_gemm_tn_f32; same bits as gcn_dropout_f32 followed by gcn_gemm_tn_f32. */
int gcn_gemm_tn_masked_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const uint32_t* mask,
                           int64_t ld_words, float keep_prob, const float* g, int64_t ldg, float* c, int64_t ldc,
                           void* workspace, size_t workspace_bytes, void* stream);

/* C[k x n] = A^T . G for A [m x k] and G [m x n], both row-major: the weight gradient
 * dW = H^T . dP that tf.train.AdamOptimizer.minimize derives for tf.matmul and for
 * tf.sparse_tensor_dense_matmul with a dense left operand (gcn/models.py:86, gcn/layers.py:28-34).
 * The sum over the m rows is cut into ranges whose partial results go to the workspace and are added
 * in range order (deterministic); query the size first (0 when one range is enough). */
int gcn_gemm_tn_workspace_bytes(int64_t m, int64_t n, int64_t k, size_t* bytes);
int gcn_gemm_tn_f32(int64_t m, int64_t n, int64_t k, const float* a, int64_t lda, const float* g, int64_t ldg,
                    float* c, int64_t ldc, void* workspace, size_t workspace_bytes, void* stream);

/* dst[i, 0:f] = src[rows[i], 0:f]: packs the rows of a rank's block of a dense matrix that other ranks
 * reference into the send buffer of the multi-GPU exchange (no counterpart in the single-process
 * reference; the dense operand is the `pre_sup` of gcn/layers.py:198-202).  rows: int64 [n_out], every
 * value in [0, n_src_rows) -- not checked. */
int gcn_gather_rows_f32(int64_t n_out, int64_t f, int64_t n_src_rows, const float* src, int64_t ld_src,
                        const int64_t* rows, float* dst, int64_t ld_dst, void* stream);

/* ------------------------------------------------------------------------- *
 * Exchange step of the row-partitioned product (one process per GPU).        *
 * The reference is single-process (its only parallelism knob is               *
 * intra_op_parallelism_threads, train.py:276-277), so these have no           *
 * counterpart there; the dense operand they move is the `pre_sup` of          *
 * gcn/layers.py:198-202.  RCCL is opened at the first call.                   *
 * ------------------------------------------------------------------------- */

/* A 128-byte communicator id (host buffer) made by one rank and handed to the others by the caller. */
int gcn_comm_unique_id(void* id128_host);
/* This process's rank of an n_ranks communicator on the CURRENT device (collective over the ranks). */
int gcn_comm_init_rank(void** comm_out, int n_ranks, const void* id128_host, int rank);
int gcn_comm_destroy(void* comm);

/* recv [n_ranks * rows_per_rank x f] = the send blocks [rows_per_rank x f] of all ranks in rank order:
 * the all-gather of the per-layer dense matrix; both buffers contiguous row-major. */
int gcn_allgather_rows_f32(void* comm, const float* send, float* recv, int64_t rows_per_rank, int64_t f, void* stream);
/* Exchange of the referenced rows only: send holds, rank after rank, send_rows_host[p] rows of f floats
 * for rank p; recv receives recv_rows_host[p] rows from rank p, rank after rank (host arrays of n_ranks
 * counts).  One grouped send/receive pair per peer, so every xGMI link carries its own pair at once.
 * If a send or receive cannot be posted the communicator is aborted (so that the peers' matching calls
 * end with an error instead of waiting for ever) and GCN_ERR_RCCL is returned: after that status the
 * communicator must not be used again.  gcn_comm_destroy after an abort: do not call it. */
int gcn_exchange_rows_f32(void* comm, const float* send, const int64_t* send_rows_host, float* recv,
                          const int64_t* recv_rows_host, int64_t f, void* stream);

/* ------------------------------------------------------------------------- *
 * Loss and optimiser of the training step.                                   *
 * ------------------------------------------------------------------------- */

/* masked_softmax_cross_entropy and masked_accuracy (gcn/metrics.py:85-101) in one pass over the
 * logits, plus the gradient of the loss w.r.t. the logits that optimizer.minimize (gcn/models.py:86)
 * would derive.  labels: int32 one-hot [n x c] (all-zero rows allowed); mask: int32 [n];
 * mask_mean = mean(mask) > 0.  row_stats [n x 2]: (CE_i * w_i, correct_i * w_i) with
 * w_i = mask_i / mask_mean -- the caller's mean over n of column 0 is the data term of the loss, of
 * column 1 the accuracy.  dlogits (may be NULL) [n x c] = (softmax_i * sum(labels_i) - labels_i) * w_i / n. */
int gcn_softmax_xent_f32(int64_t n, int64_t c, const float* logits, int64_t ldx, const int32_t* labels, int64_t ldl,
                         const int32_t* mask, float mask_mean, float* row_stats, float* dlogits, int64_t ldd,
                         void* stream);

/* Loss and accuracy of a step from the rows gcn_softmax_xent_f32 wrote and the model's weights:
 *   out2[0] = mean_i row_stats[i][0] + weight_decay * sum_j weights[j]^2 / 2
 *             (`loss += weight_decay * tf.nn.l2_loss(var)` for every variable, then the masked cross
 *             entropy: gcn/models.py:92-99; the variables lie in one buffer of n_weights floats)
 *   out2[1] = mean_i row_stats[i][1]   (gcn/models.py:106-108)
 * float64 sums in a fixed order.  workspace: gcn_loss_summary_workspace_bytes (0 for small inputs). */
int gcn_loss_summary_workspace_bytes(int64_t n, int64_t n_weights, size_t* bytes);
int gcn_loss_summary_f32(int64_t n, const float* row_stats, int64_t n_weights, const float* weights, float weight_decay,
                         void* workspace, size_t workspace_bytes, float* out2, void* stream);

/* One tf.train.AdamOptimizer step on one variable (gcn/models.py:35-38) with the weight-decay term of
 * the loss (gcn/models.py:94-96) folded into the gradient: g' = g + weight_decay * p;
 * m += (1 - beta1)(g' - m); v += (1 - beta2)(g'^2 - v); p -= lr_t * m / (sqrt(v) + epsilon), where
 * lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) is computed by the caller.  lr_t_device (may be NULL):
 * read lr_t from device memory instead (a launch recorded in a HIP graph). */
int gcn_adam_f32(int64_t n, float* param, const float* grad, float* m, float* v, float lr_t, const float* lr_t_device,
                 float beta1, float beta2, float epsilon, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCN_HIP_H_ */
