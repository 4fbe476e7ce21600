// Micro-benchmark: issue throughput (cycles per wave-instruction per SIMD) of the VALU ops the path kernel is made of.
// Many independent chains per lane, 8 waves/SIMD resident, s_memtime around an unrolled loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 256
#define NCH 8
template <int OP>
__global__ __launch_bounds__(256) void bench(double* out, uint64_t* cyc, double seed)
{
    double a[NCH]; uint32_t u[NCH]; uint64_t w[NCH];
    for (int c = 0; c < NCH; ++c) { a[c] = seed + c + threadIdx.x * 1e-3; u[c] = (uint32_t)(threadIdx.x * 2654435761u + c); w[c] = u[c]; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
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
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int c = 0; c < NCH; ++c) s += a[c] + u[c] + (double)w[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int ninstr_per_chain)
{
    double* out; uint64_t* cyc; const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    uint64_t* h = new uint64_t[blocks]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int b = 0; b < blocks; ++b) avg += h[b]; avg /= blocks;
    // each wave issues ITER*NCH*ninstr instr; 8 waves share a SIMD -> cycles per wave-instruction at full occupancy:
    double per = avg / (ITER * NCH * ninstr_per_chain) / 8.0;
    // wall clock: 20 launches, each SIMD issues 8 waves x ITER*NCH*ninstr wave-instructions (+ ~5 % prologue/epilogue)
    double ns_per = ms * 1e6 / 20.0 / (8.0 * ITER * NCH * ninstr_per_chain);
    printf("%-28s %8.0f ticks/wave-loop  => %.2f s_memtime-ticks, %.3f ns wall per wave-instr per SIMD (8 waves resident) = %.2f clk at 2.4 GHz\n",
           name, avg, per, ns_per, ns_per * 2.4);
    hipFree(out); hipFree(cyc); delete[] h;
}
int main()
{
    run<0>("v_fma_f64", 1); run<1>("v_add_f64", 1); run<2>("v_mul_f64", 1);
    run<3>("v_mad_u64_u32 (+shift)", 1); run<4>("v_mul_hi_u32 + xor", 2); run<5>("v_mul_lo_u32 + add (mad_u32_u24?)", 1);
    run<6>("v_rcp_f64", 1); run<7>("v_rsq_f64", 1); run<8>("v_ldexp_f64", 1); run<9>("v_cvt_f64_u32+fma+add", 3);
    run<10>("xor/shift int (3 ops)", 3); run<11>("v_sqrt_f64", 1); run<12>("cvt+v_log_f32+cvt", 3); run<13>("v_max_f64", 1); run<14>("v_fract_f64+add", 2);
    return 0;
}
