// tools/ubench_hbm.hip — what a read-only streaming kernel reaches on this chip, by load width, loads in flight per lane and grid
// shape (the ceiling the K4 / K5 / KL streaming kernels are measured against).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench_hbm tools/ubench_hbm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

template <class V, int UN, bool NT>
__global__ __launch_bounds__(256) void k_sum(const V* __restrict__ x, int64_t n, double* __restrict__ out)
{
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += UN * stride) {
        V v[UN];
#pragma unroll
        for (int w = 0; w < UN; ++w) {
            const int64_t j = i + w * stride;
            if (j < n) v[w] = NT ? __builtin_nontemporal_load(x + j) : x[j];
            else v[w] = V{};
        }
#pragma unroll
        for (int w = 0; w < UN; ++w) {
            if constexpr (sizeof(V) == 8) acc += ((const double*)&v[w])[0];
            else for (unsigned q = 0; q < sizeof(V) / 8; ++q) acc += ((const double*)&v[w])[q];
        }
    }
    if (acc == 123.456) out[0] = acc;      // keep the loads
}

// rows x columns: blockIdx.y = row (the [dates][paths] shape of the exposure matrix), 16-byte loads
template <int UN>
__global__ __launch_bounds__(256) void k_rows(const d2* __restrict__ x, int64_t n2, int64_t ld2, double* __restrict__ out)
{
    const d2* row = x + (int64_t)blockIdx.y * ld2;
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += UN * stride) {
        d2 v[UN];
#pragma unroll
        for (int w = 0; w < UN; ++w) { const int64_t j = i + w * stride; v[w] = j < n2 ? row[j] : d2{0.0, 0.0}; }
#pragma unroll
        for (int w = 0; w < UN; ++w) acc += v[w].x + v[w].y;
    }
    if (acc == 123.456) out[0] = acc;
}

template <class F>
static double time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int64_t rows = 121, cols = 1 << 21;                       // BASELINE config 5's exposure matrix: 1.94 GB
    const int64_t n = rows * cols;
    double *x, *out;
    hipMalloc(&x, n * 8); hipMalloc(&out, 8);
    hipMemset(x, 0, n * 8);
    const double gb = n * 8 / 1e9;
    int cus = 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
    printf("device %s, %d CUs, array %.2f GB\n", p.name, cus, gb);
    for (int bpc : {4, 8, 16, 32}) {
        const int grid = bpc * cus;
#define RUN(V, UN, NT, name) { double ms = time_ms([&] { hipLaunchKernelGGL((k_sum<V, UN, NT>), dim3(grid), dim3(256), 0, 0, (const V*)x, n * 8 / (int64_t)sizeof(V), out); }, 5); \
        printf("flat  %-14s blocks/CU %2d  %7.3f ms  %6.2f TB/s\n", name, bpc, ms, gb / ms); }
        RUN(double, 4, false, "8B x4");
        RUN(d2, 4, false, "16B x4");
        RUN(d2, 8, false, "16B x8");
        RUN(d4, 4, false, "32B x4");
        RUN(d2, 4, true, "16B x4 nt");
    }
    for (int gx : {8, 17, 34, 68}) {
        double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<4>), dim3(gx, rows), dim3(256), 0, 0, (const d2*)x, cols / 2, cols / 2, out); }, 5);
        printf("rows  16B x4  grid (%3d, %lld)  %7.3f ms  %6.2f TB/s\n", gx, (long long)rows, ms, gb / ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_rows<8>), dim3(gx, rows), dim3(256), 0, 0, (const d2*)x, cols / 2, cols / 2, out); }, 5);
        printf("rows  16B x8  grid (%3d, %lld)  %7.3f ms  %6.2f TB/s\n", gx, (long long)rows, ms, gb / ms);
    }
    return 0;
}
