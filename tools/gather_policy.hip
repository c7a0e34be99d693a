// What a random gather of short rows costs behind the L2, per cache policy of the load and per row size.
//
// A table of `rows` rows of ROW_BYTES bytes (64, 128 or 256) is gathered at hashed positions by groups of
// ROW_BYTES / 16 lanes, each lane one 16-byte buffer load whose cache-policy bits (sc0 = 1, nt = 2, sc1 = 16)
// are the template argument; every group sums `per_group` rows and writes one float4.  Prints the time per
// launch and the rows per second for each policy; under rocprofv3 --pmc TCC_EA0_RDREQ_{64B,128B}_sum the
// counters tell which policies fetch half lines.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_policy.hip -o gpurun_out/gather_policy && gpurun_out/gather_policy
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t sm64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <int AUX, int ROW_BYTES>
__global__ __launch_bounds__(256) void gather_kernel(const float* table, uint32_t row_mask, int per_group,
                                                    float* out, uint32_t n_groups) {
    constexpr int kLanes = ROW_BYTES / 16;
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t group = tid / kLanes, lane = tid % kLanes;
    if (group >= n_groups) return;
    __amdgpu_buffer_rsrc_t res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(table), 0, 0xffffffff, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    uint64_t h = sm64(group);
    for (int i = 0; i < per_group; i += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {          // four independent rows in flight per lane, as the product kernel has
            h = h * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t row = (uint32_t)(h >> 33) & row_mask;
            // the table is at most 4 GiB: byte offsets fit 32 bits
            v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                res, row * (uint32_t)ROW_BYTES + lane * 16u, 0, AUX));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    reinterpret_cast<f32x4*>(out)[tid] = acc;
}

template <int AUX, int ROW_BYTES>
static void run(const char* name, const float* table, uint32_t rows, float* out, size_t total_rows) {
    constexpr int kLanes = ROW_BYTES / 16;
    const int per_group = 64;
    const uint32_t n_groups = (uint32_t)(total_rows / per_group);
    const uint32_t blocks = (uint32_t)(((size_t)n_groups * kLanes + 255) / 256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        gather_kernel<AUX, ROW_BYTES><<<blocks, 256>>>(table, rows - 1, per_group, out, n_groups);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double n = (double)n_groups * per_group;
    printf("row %3d B  policy %-10s  %8.3f ms  %6.1f G rows/s  %6.2f TB/s of rows\n", ROW_BYTES, name, best,
           n / best * 1e-6, n * ROW_BYTES / best * 1e-9);
    fflush(stdout);
}

template <int ROW_BYTES>
static void all_policies(const float* table, size_t table_bytes, float* out, size_t total_rows) {
    const uint32_t rows = (uint32_t)(table_bytes / ROW_BYTES);
    run<0, ROW_BYTES>("default", table, rows, out, total_rows);
    run<1, ROW_BYTES>("sc0", table, rows, out, total_rows);
    run<2, ROW_BYTES>("nt", table, rows, out, total_rows);
    run<3, ROW_BYTES>("sc0 nt", table, rows, out, total_rows);
    run<16, ROW_BYTES>("sc1", table, rows, out, total_rows);
    run<17, ROW_BYTES>("sc0 sc1", table, rows, out, total_rows);
    run<18, ROW_BYTES>("sc1 nt", table, rows, out, total_rows);
    run<19, ROW_BYTES>("sc0 sc1 nt", table, rows, out, total_rows);
}

int main(int argc, char** argv) {
    const size_t table_bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048) << 20;   // MiB, a power of two, <= 4096
    const size_t total_rows = (argc > 2 ? (size_t)atoll(argv[2]) : 512) << 20;     // Mi rows gathered per launch
    if (table_bytes > (4096ull << 20) || (table_bytes & (table_bytes - 1))) { fprintf(stderr, "table size\n"); return 2; }
    float *table, *out;
    CHECK(hipMalloc(&table, table_bytes));
    CHECK(hipMemset(table, 0, table_bytes));
    CHECK(hipMalloc(&out, (total_rows / 64) * 16 * 16));     // groups x up to 16 lanes x 16 bytes
    printf("table %zu MiB, %zu Mi rows gathered per launch\n", table_bytes >> 20, total_rows >> 20);
    all_policies<64>(table, table_bytes, out, total_rows);
    all_policies<128>(table, table_bytes, out, total_rows);
    all_policies<256>(table, table_bytes, out, total_rows);
    CHECK(hipFree(table)); CHECK(hipFree(out));
    return 0;
}
