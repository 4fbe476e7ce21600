// mcx_math.h — lean float64 exp / log / sincos(2*pi*u) for the path kernels (gfx950).
//
// Why not ocml's exp/log/sincospi: a v_fma_f64 can take only ONE scalar/literal operand, so the backend materialises every
// polynomial coefficient of an inlined libm routine in a VGPR pair and hoists it out of the sub-step loop — log + sincospi
// + exp pin ~80-120 VGPRs for the whole kernel (measured: the fused kernel sat at 160-200 VGPRs = 2-3 waves/SIMD with a
// tiny live state).  Here the coefficients come from a constant-address-space table: scalar loads -> SGPRs -> the single
// scalar operand of each Horner FMA.  Accuracy: <= ~2 ulp on the ranges used (u in (0,1), |x| < 700), checked against the
// CPU oracle (glibc libm) in tests at 1e-11 relative on whole paths.
#pragma once
// (included from mcx_internal.h right after ldk())

__device__ const double MCX_EXP_C[14] __attribute__((aligned(64))) = {1.00000000000000000e+00, 1.00000000000000000e+00, 5.00000000000000000e-01, 1.66666666666666657e-01, 4.16666666666666644e-02, 8.33333333333333322e-03, 1.38888888888888894e-03, 1.98412698412698413e-04, 2.48015873015873016e-05, 2.75573192239858925e-06, 2.75573192239858883e-07, 2.50521083854417202e-08, 2.08767569878681002e-09, 1.60590438368216133e-10};
__device__ const double MCX_SIN_C[10] = {3.14159265358979312e+00, -5.16771278004997026e+00, 2.55016403987734552e+00, -5.99264529320792105e-01, 8.21458866111282326e-02, -7.37043094571435044e-03, 4.66302805767612554e-04, -2.19153534478302173e-05, 7.95205400147551261e-07, -2.29484289972698730e-08};   // sin(pi r) = sum_k S_k r^(2k+1)
__device__ const double MCX_COS_C[10] = {1.00000000000000000e+00, -4.93480220054467900e+00, 4.05871212641676848e+00, -1.33526276885458950e+00, 2.35330630358893206e-01, -2.58068913900140612e-02, 1.92957430940392314e-03, -1.04638104924845705e-04, 4.30306958703294729e-06, -1.38789524622137714e-07};   // cos(pi r) = sum_k C_k r^(2k)
__device__ const double MCX_LOG_C[7] = {6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01, 2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01, 1.479819860511658591e-01};    // fdlibm e_log.c Lg1..Lg7

// "Region zero": an SGPR that holds 0 but is opaque to the optimiser (volatile asm).  A coefficient-table read whose index
// carries a region zero cannot be hoisted above the asm, so the table's SGPRs are live only inside the code region that
// executed it (the date block, one run of the sub-step loop) instead of for the whole kernel: without it the backend keeps
// ~60 loop-invariant coefficient SGPRs alive everywhere and spills them to VGPR lanes (v_writelane / v_readlane are VALU
// instructions — the pipe these kernels are bound by).  Re-reading a table costs a couple of s_load_dwordx16 per region.
__device__ __forceinline__ int mcx_region_zero()
{
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    return z;
}
__device__ __forceinline__ uint32_t mcx_region_copy(uint32_t v)      // an SGPR copy the optimiser cannot trace back to v
{
    uint32_t o;
    asm volatile("s_mov_b32 %0, %1" : "=s"(o) : "s"(v));
    return o;
}

// exp(x), |x| <~ 708.  x = k ln2 + r, |r| <= ln2/2; Taylor degree 13 in r; scale by 2^k.
struct mcx_exp_coef { double c[14]; };
// the coefficient table as 28 SGPRs (two wide scalar loads); z: region zero (or literal 0)
__device__ __forceinline__ mcx_exp_coef mcx_exp_load(int z = 0)
{
    return ldk_struct((const mcx_exp_coef*)(MCX_EXP_C + z));
}
__device__ __forceinline__ double mcx_exp(double x, const mcx_exp_coef& C)
{
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);       // ln2_hi
    r = fma(k, -1.90821492927058770002e-10, r);              // ln2_lo
    double p = C.c[13];
#pragma unroll
    for (int j = 12; j >= 0; --j) p = fma(p, r, C.c[j]);
    return ldexp(p, (int)k);
}
__device__ __forceinline__ double mcx_exp(double x)
{
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);       // ln2_hi
    r = fma(k, -1.90821492927058770002e-10, r);              // ln2_lo
    double p = ldk(MCX_EXP_C + 13);
#pragma unroll
    for (int j = 12; j >= 0; --j) p = fma(p, r, ldk(MCX_EXP_C + j));
    return ldexp(p, (int)k);
}

// log(x) for normal x > 0 (fdlibm __ieee754_log without the special cases)
__device__ __forceinline__ double mcx_log(double x)
{
    int e;
    double m = frexp(x, &e);                                  // m in [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                                       // m in [sqrt(1/2), sqrt(2))
    e -= lo ? 1 : 0;
    const double f = m - 1.0;
    // f / (2 + f) with 2 + f in [1.7, 2.42]: v_rcp_f64 + two Newton steps + one residual correction (no IEEE fix-up paths)
    const double dd = 2.0 + f;
    double rr = __builtin_amdgcn_rcp(dd);
    rr = fma(fma(-dd, rr, 1.0), rr, rr);
    rr = fma(fma(-dd, rr, 1.0), rr, rr);
    double s = f * rr;
    s = fma(fma(-dd, s, f), rr, s);
    const double z = s * s;
    double R = ldk(MCX_LOG_C + 6);
#pragma unroll
    for (int j = 5; j >= 0; --j) R = fma(R, z, ldk(MCX_LOG_C + j));
    R *= z;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// (sin, cos)(2 pi u) for u in [0, 1): x = 2u in units of pi; n = nearest multiple of 1/2, r = x - n/2 in [-1/4, 1/4]
__device__ __forceinline__ void mcx_sincos2pi(double u, double& s, double& c)
{
    const double x = u + u;
    const double n = rint(x + x);
    const double r = fma(n, -0.5, x);
    const double t = r * r;
    double ps = ldk(MCX_SIN_C + 9), pc = ldk(MCX_COS_C + 9);
#pragma unroll
    for (int j = 8; j >= 0; --j) { ps = fma(ps, t, ldk(MCX_SIN_C + j)); pc = fma(pc, t, ldk(MCX_COS_C + j)); }
    ps *= r;
    const int q = (int)n;                                     // quadrant: angle = q*pi/2 + pi*r
    const bool swap = q & 1;
    const double a = swap ? pc : ps, b = swap ? ps : pc;      // sin(theta), cos(theta) up to signs
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// 1 / d for a normal d (numeraires, annuities): v_rcp_f64 seed + two Newton steps, no IEEE fix-up paths (a full f64 division
// compiles to ~14 VALU instructions: div_scale x2, rcp, five fma, div_fmas, div_fixup).  <= 1 ulp
__device__ __forceinline__ double mcx_rcp(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// sqrt(a) for a >= 0 in the normal range (Box-Muller radius^2, CIR state): v_rsq_f64 seed (~2^-26) + one coupled
// Goldschmidt step (~2^-50) + one residual correction, without the scaling / special-case code of the IEEE-complete library
// routine.  <= 1 ulp (correctly rounded in ~99.9 % of cases; a second residual step changed no path beyond 1e-15).
__device__ __forceinline__ double mcx_sqrt(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, a);
    g = fma(d, h, g);
    return a > 0.0 ? g : 0.0;
}

// sqrt(a) together with h = 1 / (2 sqrt(a)) — the second Goldschmidt iterate, accurate to ~2^-50: what the derivative of a root needs
// (mcx_dual.h dsqrt paid an IEEE division 0.5 / sqrt(a) for it).  a <= 0: (0, +inf) like 0.5 / 0.
__device__ __forceinline__ double mcx_sqrt_h(double a, double& h_out)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, a);
    g = fma(d, h, g);
    h_out = a <= 0.0 ? __builtin_huge_val() : h;               // (a NaN argument stays NaN in both, like sqrt)
    return a <= 0.0 ? 0.0 : g;
}

// the same without the residual correction: v_rsq_f64 seed + one coupled Goldschmidt step, error ~1.5 eps_seed^2 (1-2 ulp).  Used
// where the result feeds a Monte-Carlo increment (Box-Muller radius, CIR diffusion): 3 VALU fewer per root on a kernel that is
// bound by VALU issue
__device__ __forceinline__ double mcx_sqrt_g(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    double g = a * y;
    const double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    return a <= 0.0 ? 0.0 : g;                                 // (a NaN argument stays NaN, like sqrt)
}

// the same for a > 0 (the squared Box-Muller radius -2 log u, u < 1): no zero test
__device__ __forceinline__ double mcx_sqrt_gp(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    const double g = a * y;
    const double h = 0.5 * y;
    return fma(g, fma(-h, g, 0.5), g);
}

// ---- table-driven log / sincos for the Box-Muller transform ----------------------------------------------------------
// The path kernel is bound by f64 VALU issue (4 clk per wave64 instruction), and ~half of a sub-step was the polynomial
// log + sincos above (35 + 35 instructions).  A 128-entry table per function (2 x 2 KiB in LDS, one ds_read_b128 per lane —
// the LDS pipe is otherwise idle in these kernels) shrinks the argument range by 2^7, so a degree-7 log1p and degree-6/7
// sin/cos corrections are enough: ~16 + ~19 VALU instructions.  Absolute accuracy ~1e-16 (a few ulp of the RESULT away from
// the table nodes); log(u) keeps full relative accuracy as u -> 1 because the last node is exactly c = 1.
// BMB (template parameter of the table-driven functions): log2 of the table size.  7 (4 KiB of LDS per block) everywhere but in
// the one-launch kernel of the two-factor rates/credit configuration (kf_lean.hip), which runs four blocks per CU and spends 10
// (32 KiB) to drop two terms of the log1p polynomial and one of each trigonometric correction: |t| <= 2^-11, |d| <= pi/1024.
#include "mcx_tables.h"
#define MCX_BM_LDS_DOUBLES_B(BMB) (4 << (BMB))
#define MCX_BM_LDS_DOUBLES MCX_BM_LDS_DOUBLES_B(7)
template <int BMB> struct mcx_bm_shape;
template <> struct mcx_bm_shape<7> {
    static constexpr int LOG_TERMS = 6, TRIG_TERMS = 3;
    static __device__ __forceinline__ double log_src(int q) { return MCX_LOG_TAB[q]; }
    static __device__ __forceinline__ double trig_src(int q) { return MCX_TRIG_TAB[q]; }
};
template <> struct mcx_bm_shape<10> {
    static constexpr int LOG_TERMS = 4, TRIG_TERMS = 2;
    static __device__ __forceinline__ double log_src(int q) { return MCX_LOG_TAB_L[q]; }
    static __device__ __forceinline__ double trig_src(int q) { return MCX_TRIG_TAB_L[q]; }
};

// [0..5] -2 log1p(t) = s + s^2 (c0 + c1 s + ...), s = -2 t: the coefficients of log1p scaled by exact powers of two
// (c_k' = -c_k (-1/2)^k / 2), so the result is bit for bit -2 x (t + t^2 (-1/2 + t/3 - ...));
// [6..8] sin d = d + d^3 (...);  [9..11] cos d = 1 + d^2 (...)
__device__ const double MCX_BM_C[12] __attribute__((aligned(64))) = {0.25, (1.0 / 3.0) * 0.25, 0.03125, 0.2 * 0.0625, (1.0 / 6.0) * 0.03125, (1.0 / 7.0) * 0.015625,
                                                                     -1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, -0.5, 1.0 / 24.0, -1.0 / 720.0};
struct mcx_bm_coef { double c[12]; };
__device__ __forceinline__ mcx_bm_coef mcx_bm_coef_load(int z = 0)      // 24 SGPRs; z: region zero (or literal 0)
{
    return ldk_struct((const mcx_bm_coef*)(MCX_BM_C + z));
}

// Loop-resident copies of the constants that would otherwise be rebuilt inside a hot loop.  A VOP3 instruction reads at most one
// scalar operand, so fma(c_a, x, c_b) with both coefficients in SGPRs costs a v_mov of one of them per use, and v_fmac with a
// literal multiplier needs its constant addend copied into the destination first.  A kernel that can spare ten VGPRs builds this
// once outside its sub-step loop: the additive constants as VGPR values, the multipliers of the uniform conversion as SGPR
// values the optimiser cannot fold back into literals.
struct mcx_bm_vconst {
    double c54;            // 2^-54 (VGPR)
    double s53, s32;       // 2^-53, 2^-32 (SGPR)
    double log_head;       // leading coefficient of the log1p polynomial (VGPR)
    double sin_head, cos_head;   // leading coefficients of the sin / cos corrections (VGPR)
    double trig_off;       // -pi / N (VGPR)
};
__device__ __forceinline__ double mcx_opaque_v(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double mcx_opaque_s(double x) { asm volatile("" : "+s"(x)); return x; }
template <int BMB = 7>
__device__ __forceinline__ mcx_bm_vconst mcx_bm_vconst_make(const mcx_bm_coef& C)
{
    using SH = mcx_bm_shape<BMB>;
    mcx_bm_vconst v;
    v.c54 = mcx_opaque_v(0x1.0p-54); v.s53 = mcx_opaque_s(0x1.0p-53); v.s32 = mcx_opaque_s(0x1.0p-32);
    v.log_head = mcx_opaque_v(C.c[SH::LOG_TERMS - 1]);
    v.sin_head = mcx_opaque_v(C.c[5 + SH::TRIG_TERMS]);
    v.cos_head = mcx_opaque_v(C.c[8 + SH::TRIG_TERMS]);
    v.trig_off = mcx_opaque_v(-3.14159265358979323846 / (double)(1 << BMB));
    return v;
}

// cooperative copy of both tables into the block's LDS area (MCX_BM_LDS_DOUBLES_B(BMB) doubles); includes the barrier
template <int BMB = 7>
__device__ __forceinline__ void mcx_bm_load(double* __restrict__ tab)
{
    constexpr int N = 1 << BMB;
    for (int q = threadIdx.x; q < 2 * N; q += blockDim.x) {
        tab[q] = -2.0 * mcx_bm_shape<BMB>::log_src(q);        // (-2 / c, -2 log c): the radius needs -2 log u (exact scaling)
        tab[2 * N + q] = mcx_bm_shape<BMB>::trig_src(q);
    }
    __syncthreads();
}

typedef double mcx_d2 __attribute__((ext_vector_type(2)));

// -2 log(x), x a normal double in (0, 1]: the squared Box-Muller radius.  Every constant of log(x) = k ln2 + log c + log1p(m/c - 1)
// carries the factor -2 (table, ln2 split, polynomial: all exact power-of-two scalings), so the product -2 * log costs nothing
template <int BMB = 7>
__device__ __forceinline__ double mcx_m2log_tab(double x, const double* __restrict__ tab, const mcx_bm_coef& C, const mcx_bm_vconst* vc = nullptr)
{
    constexpr int N = 1 << BMB, LOG_TERMS = mcx_bm_shape<BMB>::LOG_TERMS;
    const double m = __builtin_amdgcn_frexp_mant(x);                  // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    const int j = (__double2hiint(m) >> (20 - BMB)) & (N - 1);        // top mantissa bits
    const mcx_d2 tc = ((const mcx_d2*)tab)[j];                        // (-2/c, -2 log c)
    const double s = fma(m, tc.x, 2.0);                                // -2 t, |t| <= 2^-(BMB+1)
    double q = vc ? vc->log_head : C.c[LOG_TERMS - 1];
#pragma unroll
    for (int k = LOG_TERMS - 2; k >= 0; --k) q = fma(q, s, C.c[k]);
    const double p = fma(s * s, q, s);                                 // -2 log1p(t)
    // -2 (e ln2 + log c) + p.  One constant for ln2: the rounding error of e * ln2 is <= 2^-54 of the RESULT (>= 2 |e| ln2 / 2), the
    // accuracy the squared radius needs (a hi / lo split would only matter for log itself near x = 1, where e = 0 anyway)
    return fma((double)e, -2.0 * 6.93147180559945309417e-01, tc.y) + p;
}

// (sin, cos)(2 pi u) from the cell index j = floor(u N) and the remainder ur = u - j / N in (0, 1/N) — both read off the integer
// image of the 53-bit uniform by the caller (draw_pair: a shift and a mask instead of scaling, rounding and converting u).
// The table nodes sit at the cell centres, angle 2 pi (j + 1/2) / N: |d| <= pi / N.
template <int BMB = 7>
__device__ __forceinline__ void mcx_sincos2pi_tab(double ur, int j, const double* __restrict__ tab, double& s, double& c, const mcx_bm_coef& C,
                                                  const mcx_bm_vconst* vc = nullptr)
{
    constexpr int N = 1 << BMB;
    const double d = fma(ur, 6.28318530717958647692, vc ? vc->trig_off : -3.14159265358979323846 / (double)N);
    const mcx_d2 sc = ((const mcx_d2*)(tab + 2 * N))[j];
    const double d2 = d * d;
    double qs, qc;
    if constexpr (mcx_bm_shape<BMB>::TRIG_TERMS == 3) {
        qs = fma(fma(vc ? vc->sin_head : C.c[8], d2, C.c[7]), d2, C.c[6]);
        qc = fma(fma(vc ? vc->cos_head : C.c[11], d2, C.c[10]), d2, C.c[9]);
    } else {
        qs = fma(vc ? vc->sin_head : C.c[7], d2, C.c[6]);
        qc = fma(vc ? vc->cos_head : C.c[10], d2, C.c[9]);
    }
    const double ps = fma(d * d2, qs, d);                              // sin d
    const double pc = fma(d2, qc, 1.0);                                // cos d
    s = fma(sc.x, pc, sc.y * ps);
    c = fma(-sc.x, ps, sc.y * pc);
}

// ---- table-driven exp for the date programs of kf_lean.hip -----------------------------------------------------------------
// A Bermudan swaption's exercise value is ~35 zero-bond prices per path and date, each one exponential: with the degree-13
// polynomial above that is 19 f64 VALU per exponential.  A 128-entry table of 2^(j/128) (1 KiB in LDS, one ds_read_b64 per lane
// on the otherwise idle LDS pipe) shrinks the reduced argument to |r| <= ln2/256, where a degree-5 polynomial is exact to 5e-19:
// 12 f64 + 3 integer VALU.  exp(x) = 2^e T[j] (1 + q(r)), x = (128 e + j) ln2/128 + r.  <= ~2 ulp.
#define MCX_EXP_LDS_DOUBLES 128
__device__ const double MCX_EXPQ_C[4] __attribute__((aligned(32))) = {0.5, 1.0 / 6.0, 1.0 / 24.0, 1.0 / 120.0};
struct mcx_expq_coef { double c[4]; };
__device__ __forceinline__ mcx_expq_coef mcx_expq_load(int z = 0) { return ldk_struct((const mcx_expq_coef*)(MCX_EXPQ_C + z)); }

// cooperative copy of the table into the block's LDS area (MCX_EXP_LDS_DOUBLES doubles); the caller synchronises
__device__ __forceinline__ void mcx_exp_tab_load(double* __restrict__ tab)
{
    for (int q = threadIdx.x; q < MCX_EXP_LDS_DOUBLES; q += blockDim.x) tab[q] = MCX_EXP2_TAB[q];
}

__device__ __forceinline__ double mcx_exp_tab(double x, const double* __restrict__ tab, const mcx_expq_coef& C)
{
    const double k = rint(x * MCX_128_LN2);
    double r = fma(k, -MCX_LN2_128_HI, x);
    r = fma(k, -MCX_LN2_128_LO, r);
    const int ki = (int)k;
    const double T = tab[ki & 127];
    double q = fma(C.c[3], r, C.c[2]);
    q = fma(q, r, C.c[1]);
    q = fma(q, r, C.c[0]);
    q = fma(q, r, 1.0);
    q *= r;                                                  // q = e^r - 1
    return ldexp(fma(T, q, T), ki >> 7);
}

// NX exponentials staged for latency: every table read goes out first, the remainder polynomials (which need no table value) cover
// the LDS latency, then the results are assembled.  The same arithmetic as mcx_exp_tab per argument (bit-identical).
template <int NX>
__device__ __forceinline__ void mcx_exp_tab_n(const double (&x)[NX], double (&out)[NX], const double* __restrict__ tab, const mcx_expq_coef& C)
{
    double r[NX], T[NX], q[NX];
    int ki[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const double k = rint(x[j] * MCX_128_LN2);
        r[j] = fma(k, -MCX_LN2_128_HI, x[j]);
        r[j] = fma(k, -MCX_LN2_128_LO, r[j]);
        ki[j] = (int)k;
        T[j] = tab[ki[j] & 127];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        double p = fma(C.c[3], r[j], C.c[2]);
        p = fma(p, r[j], C.c[1]);
        p = fma(p, r[j], C.c[0]);
        p = fma(p, r[j], 1.0);
        q[j] = p * r[j];                                       // e^r - 1
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; ++j) out[j] = ldexp(fma(T[j], q[j], T[j]), ki[j] >> 7);
}
