// Read bandwidth of one MI355X for a 16 GiB table, per access shape of a wave instruction:
//   rows:   every wave instruction reads 1 KiB contiguous (64 lanes x 16 B), U instructions in flight per wave
//   halves: lane (i = lane % 16, g = lane / 16) reads 16 B at row i, byte offset 64 s + 16 g of a 1 KiB row:
//           16 rows x 64 B per instruction, the sixteen instructions s = 0..15 of a 16-row tile issued together
// Prints GB/s for each.   hipcc -O3 --offload-arch=gfx950 tools/stream_read.hip -o tools/stream_read.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void rows_kernel(const f32x4* __restrict__ t, int64_t n_vec, float* out) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t base = wave * 64 * U; base + 64 * U <= n_vec; base += n_waves * 64 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = t[base + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

__global__ __launch_bounds__(256) void halves_kernel(const float* __restrict__ t, int64_t n_tiles, float* out) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t tile = wave; tile < n_tiles; tile += n_waves) {
        const float* src = t + (tile * 16 + i) * 256 + 4 * g;
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + 16 * u);
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// A read sweep with a thin write stream beside it: a wave reads GROUP consecutive 16 KiB tiles (16 rows of
// 1 KiB) and then writes GROUP KiB contiguous (64 bytes per row read), the traffic of a product with 256 input and
// 16 output columns.
template <int GROUP>
__global__ __launch_bounds__(256) void read_write_kernel(const float* __restrict__ t, int64_t n_tiles, float* __restrict__ w) {
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    for (int64_t grp = wave; grp * GROUP < n_tiles; grp += n_waves) {
        f32x4 acc[GROUP];
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const f32x4* src = reinterpret_cast<const f32x4*>(t + (grp * GROUP + q) * 4096) + lane;
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[u * 64];
            acc[q] = v[0];
#pragma unroll
            for (int u = 1; u < 16; ++u) acc[q] += v[u];
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q)
            reinterpret_cast<f32x4*>(w + (grp * GROUP + q) * 256)[lane] = acc[q];
    }
}

template <typename F>
static void timed(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, bytes / best * 1e-6);
    fflush(stdout);
}

int main() {
    const size_t bytes = 16ull << 30;
    float *t, *out;
    CHECK(hipMalloc(&t, bytes));
    CHECK(hipMemset(t, 0, bytes));
    CHECK(hipMalloc(&out, 256));
    const int64_t n_vec = bytes / 16;
    for (int blocks : {256 * 2, 256 * 4, 256 * 8, 256 * 16}) {
        char name[96];
        snprintf(name, sizeof name, "rows, 4 in flight, %d blocks", blocks);
        timed(name, bytes, [&] { rows_kernel<4><<<blocks, 256>>>((const f32x4*)t, n_vec, out); });
        snprintf(name, sizeof name, "rows, 8 in flight, %d blocks", blocks);
        timed(name, bytes, [&] { rows_kernel<8><<<blocks, 256>>>((const f32x4*)t, n_vec, out); });
        snprintf(name, sizeof name, "rows, 16 in flight, %d blocks", blocks);
        timed(name, bytes, [&] { rows_kernel<16><<<blocks, 256>>>((const f32x4*)t, n_vec, out); });
        snprintf(name, sizeof name, "16 rows x 64 B per instruction, %d blocks", blocks);
        timed(name, bytes, [&] { halves_kernel<<<blocks, 256>>>(t, (int64_t)(bytes / 16384), out); });
    }
    float* w;
    CHECK(hipMalloc(&w, bytes / 16));
    const int64_t n_tiles = bytes / 16384;
    for (int blocks : {256 * 4, 256 * 8}) {
        char name[96];
        snprintf(name, sizeof name, "read 16 KiB, write 1 KiB per wave, %d blocks", blocks);
        timed(name, bytes + bytes / 16, [&] { read_write_kernel<1><<<blocks, 256>>>(t, n_tiles, w); });
        snprintf(name, sizeof name, "read 64 KiB, write 4 KiB per wave, %d blocks", blocks);
        timed(name, bytes + bytes / 16, [&] { read_write_kernel<4><<<blocks, 256>>>(t, n_tiles, w); });
        snprintf(name, sizeof name, "read 256 KiB, write 16 KiB per wave, %d blocks", blocks);
        timed(name, bytes + bytes / 16, [&] { read_write_kernel<16><<<blocks, 256>>>(t, n_tiles, w); });
    }
    return 0;
}
