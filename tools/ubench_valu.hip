// Micro-benchmark: sustained issue cost (ns per wave64 instruction per SIMD, every CU busy, 8 waves per SIMD) of the VALU ops the
// path kernels are made of.  NCH independent chains per lane; ONE launch per figure, >= 10 ms long (ITER is sized for it), timed
// by HIP events around that launch alone.  The shader clock the chip holds during these launches comes from the same binary run
// under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` (tools/measure_clock.sh: busy cycles / kernel duration / 8 XCDs).
// (Round 2 also printed s_memtime deltas: s_memtime runs off a constant reference clock, not the shader clock, and a delta taken
//  inside a 38 us launch includes the wave's ramp-in — the column disagreed with the wall clock by 2.5x and is gone.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define NCH 8
template <int OP>
__global__ __launch_bounds__(256) void bench(double* out, double seed, int iters)
{
    double a[NCH]; uint32_t u[NCH]; uint64_t w[NCH];
    for (int c = 0; c < NCH; ++c) { a[c] = seed + c + threadIdx.x * 1e-3; u[c] = (uint32_t)(threadIdx.x * 2654435761u + c); w[c] = u[c]; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (OP == 0) a[c] = __builtin_fma(a[c], 1.0000001, 0.5);
            if (OP == 1) a[c] = a[c] + 1.25;
            if (OP == 2) a[c] = a[c] * 1.0000001;
            if (OP == 3) w[c] = (uint64_t)(uint32_t)w[c] * 0xD2511F53u + (w[c] >> 32);
            if (OP == 4) u[c] = __umulhi(u[c], 0xD2511F53u) ^ (uint32_t)c;
            if (OP == 5) u[c] = u[c] * 0xCD9E8D57u + 1u;
            if (OP == 6) a[c] = __builtin_amdgcn_rcp(a[c]);
            if (OP == 7) a[c] = __builtin_amdgcn_rsq(a[c]);
            if (OP == 8) a[c] = __builtin_amdgcn_ldexp(a[c], 1);
            if (OP == 9) a[c] = (double)u[c] + a[c] * 0.0, u[c] += 3;
            if (OP == 10) u[c] = (u[c] ^ 0x9E3779B9u) ^ (u[c] >> 3);
            if (OP == 11) a[c] = __builtin_amdgcn_sqrt(a[c]);
            if (OP == 12) { float f = (float)a[c]; f = __builtin_amdgcn_logf(f); a[c] = f; }
            if (OP == 13) a[c] = fmax(a[c], 0.5) ;
            if (OP == 14) a[c] = __builtin_amdgcn_fract(a[c]) + 1.0;
        }
    }
    double s = 0; for (int c = 0; c < NCH; ++c) s += a[c] + u[c] + (double)w[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char* name, int ninstr_per_chain)
{
    double* out; const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipMalloc(&out, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // size the launch: a short probe, then ITER for ~20 ms
    int iters = 4096;
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * (20.0 / (ms > 1e-3 ? ms : 1e-3)));
    if (iters < 100000) iters = 100000;
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iters);      // clocks up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iters);      // the ONE timed launch
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    // each SIMD issues 8 waves x iters*NCH*ninstr wave-instructions
    const double ns_per = ms * 1e6 / (8.0 * (double)iters * NCH * ninstr_per_chain);
    printf("%-36s launch %7.2f ms (ITER %8d)  %.3f ns per wave-instr per SIMD (8 waves resident)\n", name, ms, iters, ns_per);
    hipFree(out);
}
int main()
{
    run<0>("v_fma_f64", 1); run<1>("v_add_f64", 1); run<2>("v_mul_f64", 1);
    run<3>("v_mad_u64_u32 (+shift)", 1); run<4>("v_mul_hi_u32 + xor", 2); run<5>("v_mul_lo_u32 + add (mad_u32_u24?)", 1);
    run<6>("v_rcp_f64", 1); run<7>("v_rsq_f64", 1); run<8>("v_ldexp_f64", 1); run<9>("v_cvt_f64_u32+fma+add", 3);
    run<10>("xor/shift int (3 ops)", 3); run<11>("v_sqrt_f64", 1); run<12>("cvt+v_log_f32+cvt", 3); run<13>("v_max_f64", 1); run<14>("v_fract_f64+add", 2);
    return 0;
}
