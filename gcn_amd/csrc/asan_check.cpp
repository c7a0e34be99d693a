// Host-side sanitizer run of libgcn_hip's sources (`make -C gcn_amd/csrc asan`): this file and the library's
// sources are compiled into ONE executable with AddressSanitizer and UndefinedBehaviorSanitizer on the host
// code (device code is built as usual and never runs here).  It needs no GPU: it walks the host-only parts of
// the C ABI -- the size queries, the argument validation of every entry point, which returns before any HIP
// call, the error text -- and the host half of plan creation (the long-row tables) on random row lengths.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "common.h"

static int g_failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); \
            ++g_failures;                                                     \
        }                                                                     \
    } while (0)

static void size_queries() {
    int64_t cap = 0;
    size_t bytes = 0;
    for (int64_t n : {0ll, 1ll, 1000ll, 1ll << 20}) {
        EXPECT(gcn_coo_to_csr_bound(n, 3 * n, GCN_ADD_SELF_LOOPS, &cap, &bytes) == GCN_OK && cap >= 4 * n && bytes > 0);
        EXPECT(gcn_coo_to_csr_bound(n, 3 * n, GCN_BINARY, &cap, &bytes) == GCN_OK);
        EXPECT(gcn_csr_transpose_bound(n, n, 5 * n, &bytes) == GCN_OK && bytes > 0);
        EXPECT(gcn_degree_order_bound(n, &bytes) == GCN_OK);
        EXPECT(gcn_csr_relabel_bound(n, 5 * n, 1, &bytes) == GCN_OK);
        EXPECT(gcn_hub_order_bound(n, &bytes) == GCN_OK && bytes > 0);
        EXPECT(gcn_rmat_csr_rows_bound(7 * n, &bytes) == GCN_OK);
        EXPECT(gcn_gemm_tn_workspace_bytes(n, 256, 256, &bytes) == GCN_OK);
        EXPECT(gcn_gemm_tn_workspace_bytes(n, 7, 16, &bytes) == GCN_OK);
        EXPECT(gcn_loss_summary_workspace_bytes(n, 1433 * 16 + 16 * 7, &bytes) == GCN_OK);
    }
    EXPECT(gcn_rmat_csr_bound(20, 16, GCN_ADD_SELF_LOOPS, &cap, &bytes) == GCN_OK && cap == (33ll << 20));
    EXPECT(gcn_rmat_csr_bound(0, 16, 0, &cap, &bytes) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_coo_to_csr_bound(-1, 0, 0, &cap, &bytes) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_coo_to_csr_bound(1, 1, 0, nullptr, &bytes) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_version() >= 100 && strlen(gcn_build_hash()) == 64);
}

static void argument_validation() {
    gcn_spmm_plan* plan = nullptr;
    float x[16] = {0};
    int64_t ip[2] = {0, 0};
    EXPECT(gcn_spmm_plan_create(-1, 1, 0, ip, &plan, nullptr) == GCN_ERR_BAD_ARG && plan == nullptr);
    EXPECT(strstr(gcn_last_error_string(), "bad shape") != nullptr);
    EXPECT(gcn_spmm_plan_create(1, 1, 0, nullptr, &plan, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_create(1, 1, 0, ip, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_create_split(1, 1, 0, ip, 5, &plan, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_create_split(1, 1, 0, ip, 5000, &plan, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_destroy(nullptr) == GCN_OK);
    EXPECT(gcn_spmm_plan_set_timing(nullptr, 1) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_set_lds_hub_rows(nullptr, 1) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_set_hot_columns(nullptr, 1) == GCN_ERR_BAD_ARG);
    int64_t info[4];
    size_t bytes = 0;
    EXPECT(gcn_spmm_plan_info(nullptr, info) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_plan_workspace_bytes(nullptr, 4, &bytes) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, x, 4, 0, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_f32_noalloc(nullptr, ip, nullptr, nullptr, 4, x, 4, x, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_add_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, nullptr, 4, x, 4, 0, nullptr, 0, nullptr) ==
           GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_add_out_dropout_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, nullptr, 4, x, 4, 0, 0.5f, 1, 1,
                                            nullptr, 0, 4, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_add_out_dropout_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, x, 4, x, 4, 0, 0.5f, 1, 1, nullptr,
                                            0, 2, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    {   // packed operand: pitch, shapes, alignment
        int64_t pitch = 0;
        EXPECT(gcn_pack_pitch(256, &pitch) == GCN_OK && pitch == 288);
        EXPECT(gcn_pack_pitch(1, &pitch) == GCN_OK && pitch == 32);
        EXPECT(gcn_pack_pitch(257, &pitch) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_pack_pitch(16, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_pack_rows_f32(4, 300, x, 300, x, 320, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_pack_rows_f32(4, 16, x, 16, x, 16, nullptr) == GCN_ERR_BAD_ARG);        // pitch below 32
        EXPECT(gcn_pack_rows_f32(4, 16, nullptr, 16, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_pack_rows_f32(0, 16, nullptr, 16, nullptr, 32, nullptr) == GCN_OK);
        EXPECT(gcn_dropout_pack_f32(4, 16, x, 16, 0.0f, 1, 1, nullptr, 0, 16, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_dropout_pack_f32(4, 16, x, 16, 0.5f, 1, 1, nullptr, 0, 8, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_dropout_pack_rows_f32(4, 16, x, 16, 1.5f, 1, 1, nullptr, 0, 16, nullptr, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_dropout_pack_rows_f32(4, 16, x, 16, 0.5f, 1, 1, nullptr, -1, 16, nullptr, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_unpack_rows_f32(4, 16, x, 32, x, 8, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_gemm_gated_pack_f32(4, 16, 17, x, 17, x, 17, x, 16, 0.5f, x, 32, nullptr) == GCN_ERR_BAD_ARG);   // k > 16
        EXPECT(gcn_gemm_gated_pack_f32(4, 300, 8, x, 8, x, 8, x, 300, 0.5f, x, 320, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_gemm_gated_pack_f32(4, 16, 8, x, 8, x, 8, x, 16, 0.0f, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_gemm_gated_pack_f32(4, 16, 8, nullptr, 8, x, 8, x, 16, 0.5f, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_gemm_gated_pack_rows_f32(4, 16, 8, x, 8, x, 8, x, 16, 0.5f, nullptr, 2, x, 32, nullptr) == GCN_ERR_BAD_ARG);
        EXPECT(gcn_gemm_gated_pack_rows_f32(4, 16, 8, x, 8, x, 8, x, 16, 0.5f, (const int64_t*)x, 5, x, 32, nullptr) ==
               GCN_ERR_BAD_ARG);                                                              // more rows listed than exist
        EXPECT(gcn_spmm_csr_packed_f32(nullptr, ip, nullptr, nullptr, 16, x, 32, x, 16, 0, nullptr, 0, nullptr) ==
               GCN_ERR_BAD_ARG);
    }
    EXPECT(gcn_spmm_csr_dropout_f32(nullptr, ip, nullptr, nullptr, nullptr, 4, x, 4, x, 4, 0, 0.0f, 1, 1, nullptr,
                                    nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_out_dropout_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, x, 4, 0, 0.5f, 1, 1, nullptr, 0, 2,
                                        nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_spmm_csr_two_part_f32(nullptr, ip, nullptr, nullptr, 4, x, 4, 1, x, 4, x, 4, 0, 1.0f, 0, 0, nullptr, 0, 4,
                                     nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    uint32_t bits[4] = {0};
    EXPECT(gcn_dropout_bits(4, 40, 0.5f, 1, 1, nullptr, 0, 40, bits, 1, nullptr) == GCN_ERR_BAD_ARG);   // 2 words needed
    EXPECT(gcn_dropout_bits(4, 40, 0.0f, 1, 1, nullptr, 0, 40, bits, 2, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_dropout_bits(0, 40, 0.5f, 1, 1, nullptr, 0, 40, bits, 2, nullptr) == GCN_OK);
    EXPECT(gcn_gemm_masked_f32(4, 4, 40, x, 40, bits, 1, 0.5f, x, 4, 0, x, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_masked_f32(4, 4, 4, x, 4, nullptr, 1, 0.5f, x, 4, 0, x, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_masked_f32(4, 4, 4, x, 4, bits, 1, 1.5f, x, 4, 0, x, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_tn_masked_f32(4, 4, 4, x, 4, nullptr, 1, 0.5f, x, 4, x, 4, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_tn_masked_f32(4, 4, 4, x, 2, bits, 1, 0.5f, x, 4, x, 4, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_f32(4, 4, 4, nullptr, 2, nullptr, 4, 0, nullptr, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_f32(4, 4, 4, nullptr, 4, nullptr, 4, 0, nullptr, 4, 64, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_f32(4, 4, 4, nullptr, 4, nullptr, 4, 0, nullptr, 4, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_f32(0, 4, 4, nullptr, 4, nullptr, 4, 0, nullptr, 4, 0, nullptr) == GCN_OK);
    EXPECT(gcn_gemm_gated_f32(4, 4, 4, x, 4, x, 4, 0, x, 4, 0.0f, x, 4, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_out_dropout_f32(4, 4, 4, x, 4, x, 4, 0, x, 4, 0, 1.5f, 1, 1, nullptr, 0, 4, nullptr, nullptr) ==
           GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_out_dropout_f32(4, 4, 4, x, 4, x, 4, 0, x, 4, 0, 0.5f, 1, 1, nullptr, 0, 2, nullptr, nullptr) ==
           GCN_ERR_BAD_ARG);
    EXPECT(gcn_gemm_tn_f32(4, 4, 4, x, 2, x, 4, x, 4, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_dropout_f32(4, 4, nullptr, 4, nullptr, 0, nullptr, 4, 0.0f, 1, 1, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_dropout_f32(4, 4, nullptr, 4, nullptr, 0, nullptr, 4, 0.5f, 1, 1, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_gather_rows_f32(4, 4, 4, nullptr, 2, nullptr, nullptr, 4, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_coo_to_csr(-1, 1, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                          nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_rows_to_indptr(-1, 0, nullptr, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_row_sums(-1, nullptr, nullptr, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_degree_scale(4, nullptr, 0, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_normalize_adj(4, 4, nullptr, nullptr, nullptr, nullptr, 7, nullptr, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_row_normalize(4, 4, ip, x, 3, x, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_csr_transpose(4, 4, 4, nullptr, nullptr, nullptr, 4, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_degree_order(4, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_hub_order(4, nullptr, 1, nullptr, nullptr, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_csr_relabel(-1, 4, 4, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_check_indices(-1, nullptr, 4, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_check_indices(0, nullptr, 4, nullptr) == GCN_OK);
    EXPECT(gcn_rmat_csr(0, 16, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    int64_t keys = 0;
    EXPECT(gcn_rmat_rows_count(10, 16, 1, 64, 0, 8, &keys, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_rmat_rows_count(10, 16, 1, 0, 8, 4, &keys, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_rmat_csr_rows(10, 16, 1, 0, 0, 8, -1, nullptr, nullptr, nullptr, nullptr, 0, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_fill_synthetic_f32(-1, 1, nullptr, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_fill_synthetic_range_f32(0, 0, 1, nullptr, nullptr) == GCN_OK);
    EXPECT(gcn_softmax_xent_f32(-1, 4, nullptr, 4, nullptr, 4, nullptr, 1.0f, nullptr, nullptr, 4, nullptr) != GCN_OK);
    EXPECT(gcn_adam_f32(-1, nullptr, nullptr, nullptr, nullptr, 0.1f, nullptr, 0.9f, 0.999f, 1e-8f, 0.0f, nullptr) != GCN_OK);
    EXPECT(gcn_loss_summary_f32(-1, nullptr, 0, nullptr, 0.0f, nullptr, 0, nullptr, nullptr) != GCN_OK);
    // the exchange entry points: every argument check comes before RCCL is opened
    void* comm = nullptr;
    char id[128] = {0};
    int64_t counts[1] = {0};
    EXPECT(gcn_comm_unique_id(nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_comm_init_rank(nullptr, 1, id, 0) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_comm_init_rank(&comm, 0, id, 0) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_comm_init_rank(&comm, 2, id, 2) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_comm_init_rank(&comm, 1, nullptr, 0) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_comm_destroy(nullptr) == GCN_OK);
    EXPECT(gcn_allgather_rows_f32(nullptr, x, x, 1, 4, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(gcn_exchange_rows_f32(nullptr, x, counts, x, counts, 4, nullptr) == GCN_ERR_BAD_ARG);
    EXPECT(strlen(gcn_last_error_string()) > 0);
}

static void dropout_thresholds() {
    // the integer form of the dropout predicate agrees with the float32 form on every field value
    for (float keep : {0.5f, 0.8f, 0.9f, 0.25f, 1.0f, 0.999999f, 1e-6f, 0.3333333f}) {
        const uint32_t t = gcn::keep_threshold(keep);
        EXPECT(t <= 65536);
        for (uint32_t field = 0; field < 65536; ++field) {
            const bool by_float = gcn::passes_f32(keep, field);
            if (by_float != (field >= t)) {
                EXPECT(by_float == (field >= t));
                break;
            }
        }
    }
    EXPECT(gcn::keep_threshold(1.0f) == 0 && gcn::keep_threshold(0.5f) == 32768);
    // four consecutive ids share a hash and read its fields from the most significant one down
    const uint64_t key = gcn::stream_key(12, 3);
    for (uint64_t id = 0; id < 64; id += 4) {
        const uint64_t h = gcn::decision_hash(key, id);
        for (uint64_t e = 0; e < 4; ++e) {
            EXPECT(gcn::decision_hash(key, id + e) == h);
            EXPECT(gcn::decision_field(h, id + e) == ((h >> (48 - 16 * e)) & 0xffff));
        }
    }
}

static void long_row_tables() {
    std::mt19937_64 rng(7);
    for (int trial = 0; trial < 200; ++trial) {
        const size_t count = (size_t)(rng() % 300);
        const int64_t split0 = 16 + (int64_t)(rng() % 300);
        std::vector<int32_t> rows(count);
        std::vector<int64_t> starts(count), ends(count);
        // distinct row numbers in a shuffled order (the collecting kernel's atomics give any order)
        std::vector<int32_t> ids(count);
        for (size_t i = 0; i < count; ++i) ids[i] = (int32_t)(i * 3 + (rng() % 3));
        for (size_t i = count; i > 1; --i) std::swap(ids[i - 1], ids[rng() % i]);
        for (size_t i = 0; i < count; ++i) {
            rows[i] = ids[i];
            starts[i] = (int64_t)(rng() % (1ll << 40));
            ends[i] = starts[i] + split0 + 1 + (int64_t)(rng() % 5000);
        }
        for (int64_t split : {split0, split0 * 4}) {
            const int64_t chunk_len = (trial & 1) ? split : std::max<int64_t>(16, split / 4);
            gcn::LongTableHost t;
            gcn::build_long_table(rows, starts, ends, split, chunk_len, &t);
            EXPECT(t.long_chunk.size() == t.long_row.size() + 1);
            EXPECT(t.long_chunk.empty() ? false : (int64_t)t.chunk_long.size() == t.long_chunk.back());
            size_t expected = 0;
            for (size_t i = 0; i < count; ++i) expected += (ends[i] - starts[i] > split) ? 1 : 0;
            EXPECT(t.long_row.size() == expected);
            for (size_t i = 0; i + 1 < t.long_row.size(); ++i) EXPECT(t.long_row[i] < t.long_row[i + 1]);
            for (size_t i = 0; i < t.long_row.size(); ++i) {
                size_t src = 0;
                while (rows[src] != t.long_row[i]) ++src;
                const int64_t len = ends[src] - starts[src];
                const int64_t n_chunks = t.long_chunk[i + 1] - t.long_chunk[i];
                EXPECT(n_chunks == (len + chunk_len - 1) / chunk_len);
                for (int64_t c = t.long_chunk[i]; c < t.long_chunk[i + 1]; ++c) EXPECT(t.chunk_long[(size_t)c] == (int32_t)i);
            }
        }
    }
}

int main() {
    size_queries();
    argument_validation();
    dropout_thresholds();
    long_row_tables();
    if (g_failures) {
        fprintf(stderr, "asan_check: %d check(s) failed\n", g_failures);
        return 1;
    }
    printf("asan_check ok\n");
    return 0;
}
