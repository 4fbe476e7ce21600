// kf_common.h — records and device helpers shared by the fused main-pass kernels (kf_fused.hip: event interpreter and
// evaluation-from-paths pass; kf_lean.hip: the straight-line one-launch kernel).  gfx950 only.
#pragma once
#include <cstddef>
#include "mcx_device.h"



struct FAtom {             // value = a + d*x + b*exp(c0 + c1*x), x = register `reg` of the lane (reg < 0: x = 0)
    int32_t reg, pad;
    double a, d, b, c0, c1;
};
struct FTerm { double w; FAtom atom; };
struct FEvent {
    int32_t kind, flags;
    int32_t term_begin, term_end;
    int32_t coeff_off, row;
    int32_t ns, sidx;          // netting-set slot (0..3), stateful-product slot (0..3) or -1
    int32_t init_state, pad;
    double strike, sign;
    double aux[4];
    FAtom num, x;
};
struct FMetricOp {             // one (netting set, metric date) pair, executed after the date's events
    int32_t ns, m;
    int32_t rec_profile;       // record index of relu(u) (rec+1 = -relu(-u)), -1: no profiles
    int32_t has_cva;           // 1: m < n_dates-1 and CVA wanted
    double threshold;
    FAtom surv, cond;
};

struct ChunkHeader { int32_t n_ev, n_mop, n_terms, bytes; };

// Straight-line record of a date whose program is the common linear-book shape (one netting set; cashflows that are an
// affine term + <= 4 exponential terms over a pure-exponential or constant numeraire; stateless polynomial exposures; an
// optional threshold / EPE-ENE record / CVA increment).  Such a date runs ~130 instructions of branch-light code with every
// control field in SGPRs instead of ~500 instructions of event interpretation; any other date uses the interpreter.
struct FastDate {
    // ---- hot head (FastDateHot below is a view of these 144 bytes: ONE batch of scalar loads at the top of a date) ----
    int32_t valid, flags;            // valid: 0 interpreted, 1 straight-line in every fused kernel, 2 straight-line in kf_lean only
                                     // (exercise event / state-dependent exposure: flags 128 / 256)
                                     // flags: 1 cash, 2 expo, 4 cva, 8 profile, 16 constant numeraire, 32 metric op present,
                                     // 64: the CVA increment may use the merged discount x survival factor (m_*): no threshold,
                                     //     no EPE / ENE record on this date
    int32_t ni_reg, lin_reg, n_exp, x_reg, coeff_off0, coeff_off1, rec_profile, s_reg, c_reg, pad;
    int32_t t_reg[4];
    double x_a, x_d;                 // explanatory x = x_a + x_d * reg[x_reg]
    double c_a, c_b, c_c0, c_c1;     // S(t,t+) cond = c_a + c_b exp(c_c0 + c_c1 reg[c_reg])
    double m_b, m_c0, m_n1, m_s1;    // S(0,t) / numeraire = m_b exp(m_c0 + m_n1 reg[ni_reg] + m_s1 reg[s_reg])   (flag 64)
    // ---- the rest: read at the point of use ----
    double ni_c0, ni_c1;             // 1/numeraire = exp(ni_c0 + ni_c1 x)   (flag 16: = ni_c0)
    double k0, k1;                   // cash affine part k0 + k1 * reg[lin_reg]
    double t_w[4], t_c0[4], t_c1[4];
    double thr;
    double s_b, s_c0, s_c1;          // S(0,t)       = s_b exp(s_c0 + s_c1 reg[s_reg])
    // flag 128: exercise event of the book's ONE two-state exercise product (bermudan_option.py:93-131): immediate value
    // max(ex_sign (ex_k0 + ex_k1 reg[ex_lin_reg] + sum_j w_j exp(c0_j + c1_j reg[r_j]) - ex_strike), 0) over the LeanTerm range
    // [ex_term_off, ex_term_off + ex_n); continuation = polynomial (row of state 1 at ex_coeff_off + n_basis) in ex_x_a + ex_x_d reg[ex_x_reg]
    // flag 256: the exposure polynomial's coefficient row is indexed by the lane's exercise state (coeff_off0 + state * n_basis)
    int32_t ex_n, ex_term_off, ex_coeff_off, ex_x_reg, ex_lin_reg, ex_pad;
    double ex_k0, ex_k1, ex_strike, ex_sign, ex_x_a, ex_x_d;
    // flag 512: the date's cash value is a plain option payoff (european_option.py:45-68): max(op_sign (value - op_strike), 0)
    double op_strike, op_sign;
    // flag 1024: the exercise value (affine part, constants and every exponential term) is also available as ONE verified polynomial of
    // reg[ex_p_reg] (mcx_vpoly.hip): p(t), t = fma(x, ex_p_ih, ex_p_ms), ex_p_blk blocks of 4 coefficients at vcoef + ex_p_off, valid
    // for ex_p_lo <= x <= ex_p_hi; a wave that holds a path outside runs the LeanTerm loop
    double ex_p_lo, ex_p_hi, ex_p_ms, ex_p_ih;
    int32_t ex_p_off, ex_p_blk, ex_p_reg, ex_p_pad;
};
// the head of a FastDate as one record (what a CVA-only date reads: kf_lean.hip lean_date)
struct FastDateHot {
    int32_t valid, flags;
    int32_t ni_reg, lin_reg, n_exp, x_reg, coeff_off0, coeff_off1, rec_profile, s_reg, c_reg, pad;
    int32_t t_reg[4];
    double x_a, x_d;
    double c_a, c_b, c_c0, c_c1;
    double m_b, m_c0, m_n1, m_s1;
};
static_assert(sizeof(FastDateHot) == 144 && offsetof(FastDate, ni_c0) == sizeof(FastDateHot), "FastDateHot must mirror the head of FastDate");
struct LeanTerm { double w, c0, c1; int32_t reg, pad; };     // w exp(c0 + c1 reg[reg]), read through scalar loads

struct FusedArgs {
    K1Args k1;
    const FastDate* __restrict__ fast;        // [n_dates]
    const LeanTerm* __restrict__ lterms;      // exponential terms of the exercise values (FastDate::ex_term_off)
    const double* __restrict__ vcoef;         // coefficients of the exercise-value polynomials (FastDate::ex_p_off), one spare block at the end
    const unsigned char* __restrict__ prog;   // per-date program chunks (header | events | terms | metric ops)
    const int32_t* __restrict__ date_off;     // [n_dates+1] byte offset of each date's chunk (16-byte aligned)
    const int32_t* __restrict__ date_row;     // [n_dates] exposure row of this timeline date or -1
    const double* __restrict__ coeffs;
    double* __restrict__ cfs;                 // nullable [NS][ld_out]
    double* __restrict__ expo;                // nullable [NS][n_expo_rows][ld_out]
    double* __restrict__ partials;            // [gridDim.x][n_rec][4]
    int64_t ld_out;
    int32_t n_dates, n_basis, n_ns, n_rec, n_expo_rows, n_stateful, chunk_cap, pad;
    int32_t rec_pv[MCX_FUSED_MAX_NS];         // record index of the PV record of ns slot k, or -1
    int32_t rec_cva[MCX_FUSED_MAX_NS];
    double lgd[MCX_FUSED_MAX_NS];
    int32_t init_state[MCX_FUSED_MAX_STATEFUL];
};

// merge of two (count, mean, centred second moment) triples (Chan, Golub, LeVeque pairwise update)
__device__ __forceinline__ void chan_merge(double& N, double& mean, double& M2, double n, double m, double q)
{
    if (n <= 0.0) return;
    if (N == 0.0) { N = n; mean = m; M2 = q; return; }
    const double delta = m - mean, tot = N + n;
    mean += delta * n / tot;
    M2 += q + delta * delta * N * n / tot;
    N = tot;
}

#define RFL(x) __builtin_amdgcn_readfirstlane(x)

template <int NREG>
__device__ __forceinline__ double f_atom(const FAtom& a, const double (&reg)[NREG])
{
    // the program lives in LDS, so its fields arrive in VGPRs; the CONTROL fields are made wave-uniform SGPRs
    // (v_readfirstlane) so that selects / branches are scalar instead of exec-masked divergent code
    const int r = RFL(a.reg), fl = RFL(a.pad);          // pad: bit0 = has exp term, bit1 = has affine term
    const double x = r >= 0 ? reg[r] : 0.0;             // uniform dynamic index -> M0-relative VGPR read (v_movrels)
    double v = (fl & 2) ? fma(a.d, x, a.a) : 0.0;
    if (fl & 1) v = fma(a.b, mcx_exp(fma(a.c1, x, a.c0)), v);
    return v;
}

// coefficients of a STATELESS product are the same for every lane: wave-uniform offset -> scalar loads (s_load through the
// scalar cache) instead of a dependent per-lane global load with L2 latency on every exposure date
__device__ __forceinline__ double f_poly_uniform(const double* __restrict__ coeffs, int off_vgpr, int K, double x)
{
    const double* __restrict__ c = coeffs + __builtin_amdgcn_readfirstlane(off_vgpr);
    double v = 0.0, xp = 1.0;
#pragma unroll 1
    for (int k = 0; k < K; ++k) { v = fma(ldk(c + k), xp, v); xp *= x; }
    return v;
}

__device__ __forceinline__ double f_poly(const double* __restrict__ c, int K, double x)
{
    double v = 0.0, xp = 1.0;
#pragma unroll 1
    for (int k = 0; k < K; ++k) { v = fma(c[k], xp, v); xp *= x; }
    return v;
}

// LDS record area: shift[n_rec] | acc[4 waves][n_rec][2]
__device__ __forceinline__ void f_record(double v, bool live, int rec, int n_rec, bool first_tile, double* __restrict__ lds)
{
    if (first_tile) {                      // block-uniform: the first path the block sees fixes the record's shift
        __syncthreads();
        if (threadIdx.x == 0) lds[rec] = v;
        __syncthreads();
    }
    const double c = lds[rec];
    const double d = live ? v - c : 0.0;
    const double s1 = wave_sum(d), s2 = wave_sum(d * d);
    if ((threadIdx.x & 63) == 0) {
        double* acc = lds + n_rec + ((threadIdx.x >> 6) * n_rec + rec) * 2;
        acc[0] += s1;
        acc[1] += s2;
    }
}

template <int NREG>
__device__ __forceinline__ double f_regsel(int r, const double (&reg)[NREG])      // r is wave-uniform (SGPR)
{
    return r >= 0 ? reg[r] : 0.0;       // uniform dynamic index -> v_movrels (M0-relative VGPR read), no select chain
}


// kf_lean.hip: launches the straight-line one-launch kernel for a book whose every date has a FastDate record; returns the
// grid size (= number of per-block partial records written to a.partials), or -1 when (slots, z) has no instantiation;
// simulate = false: the date programs run on the paths tensor a.k1.paths (the evaluation pass of mcx_fused_eval_paths)
int mcx_launch_kf_lean(const FusedArgs& a, const mcx_sim_desc& sd, int n_cu, bool inject, bool simulate, hipStream_t s);
