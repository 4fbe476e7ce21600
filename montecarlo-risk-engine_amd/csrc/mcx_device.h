// mcx_device.h — device functions shared by the path kernel (k1_paths.hip) and the fused kernel (kf_fused.hip):
// Philox4x32-10, Box-Muller, the per-model sub-step maps.  gfx950 only.
#pragma once
#include "mcx_internal.h"

struct K1Args {
    mcx_slot slots[MCX_MAX_SLOTS];
    double init_state[MCX_MAX_STATE];
    const mcx_step* __restrict__ steps;
    const double* __restrict__ chol;
    const double* __restrict__ aux;
    double* __restrict__ paths;
    const double* __restrict__ inject_z;
    const double* __restrict__ inject_u;
    const double* __restrict__ init_paths;     // nullable: per-path initial state [n_state][ld] (mcx_generate_paths_from_state)
    int64_t n, ld;
    uint64_t seed, path_offset;
    int32_t scheme, n_steps, n_state, n_initial_store, flags, n_uniform;
};

static inline void mcx_fill_k1_args(const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths, int64_t ld,
                                    double* d_paths, const double* d_inject_z, const double* d_inject_u, K1Args* a)
{
    const mcx_sim_desc& d = sim->desc;
    memset(a, 0, sizeof(*a));
    for (int s = 0; s < d.n_slots; ++s) a->slots[s] = d.slots[s];
    for (int c = 0; c < d.n_state; ++c) a->init_state[c] = d.init_state[c];
    a->steps = sim->d_steps; a->chol = sim->d_chol; a->aux = sim->d_aux;
    a->paths = d_paths; a->inject_z = d_inject_z; a->inject_u = d_inject_u;
    a->n = n_paths; a->ld = ld; a->seed = seed; a->path_offset = path_offset;
    a->scheme = d.scheme; a->n_steps = d.n_steps; a->n_state = d.n_state; a->n_initial_store = d.n_initial_store;
    a->flags = d.flags; a->n_uniform = d.n_uniform;
}

// ---- Philox4x32-10 ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;     // v_mad_u64_u32
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);   // xor3 in one VALU op
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi, const mcx_bm_vconst* vc = nullptr)
{
    // ((x >> 11) + 0.5) * 2^-53 with x = hi:lo, evaluated as hi*2^-32 + ((lo >> 11)*2^-53 + 2^-54): the inner sum is exact and the
    // outer fma rounds once, exactly like the oracle's (double)(x >> 11) + 0.5.  vc: loop-resident constants (mcx_math.h)
    if (vc) return fma((double)hi, vc->s32, fma((double)(lo >> 11), vc->s53, vc->c54));
    return fma((double)hi, 0x1.0p-32, fma((double)(lo >> 11), 0x1.0p-53, 0x1.0p-54));
}

// ---- barrier options (barrier_option.py:60-223): MCX_EV_OPTION modes 4 (discrete monitoring) and 5 (+ Brownian bridge) ------
// RNG state of the bridge draws, one per book (mcx_book_set_bridge_rng): production draws are Philox4x32-10 with key = seed and
// counter = (global path id, interval, 0x80000000 | (2 * draw_id + barrier)); a parity run injects the reference's uniforms.
struct DevBridge {
    uint64_t seed, path_offset;
    const double* const* inject;     // [n_products] device pointers to [2 * n_intervals][ld] uniforms, or nullptr
    int64_t ld;
};

__device__ __forceinline__ double dev_barrier_event(const DevEvent& e, const DevTerm* __restrict__ terms, const double* __restrict__ coeffs,
                                                    const DevBridge* __restrict__ bridge, const double* __restrict__ paths, int64_t D,
                                                    int64_t ld, int64_t i, double num)
{
    const int types = (int)e.aux[3];
    const int t1 = types & 7, t2 = types >> 3;
    const bool with_bridge = e.aux[0] == 5.0;
    double mx = -1.0e300, mn = 1.0e300, keep1 = 1.0, keep2 = 1.0, prev1 = 0.0, prev2 = 0.0;
    double c = 0.0;
    int draw_id = 0;
    DevBridge br = {0, 0, nullptr, 0};
    const double* __restrict__ inj = nullptr;
    if (with_bridge) {
        c = ldk(coeffs + e.coeff_off);                       // -2 / (sigma^2 * maturity / n_observations)   (:146)
        draw_id = (int)ldk(coeffs + e.coeff_off + 1);
        br = ldk_struct(bridge);
        if (br.inject) inj = br.inject[draw_id];
    }
    AtomCache bc = {-1, -1, 0.0};
    for (int j = e.term_begin; j < e.term_end; ++j) {
        const double s = dev_atom_cached(ldk_struct(&terms[j]).atom, paths, D, ld, i, bc);
        mx = fmax(mx, s); mn = fmin(mn, s);
        if (with_bridge) {
            const double cur1 = mcx_log(s / e.aux[1]), cur2 = t2 ? mcx_log(s / e.aux[2]) : 0.0;
            if (j > e.term_begin) {
                const int k = j - e.term_begin - 1;          // interval index
                for (int bi = 0; bi < (t2 ? 2 : 1); ++bi) {
                    const double p = mcx_exp(c * (bi ? prev2 * cur2 : prev1 * cur1));      // crossing probability of the bridge
                    double u;
                    if (inj) u = inj[(int64_t)(2 * k + bi) * br.ld + i];
                    else {
                        uint32_t w0, w1, w2, w3;
                        const uint64_t path = br.path_offset + (uint64_t)i;
                        philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), (uint32_t)k, 0x80000000u | (uint32_t)(2 * draw_id + bi),
                                      (uint32_t)br.seed, (uint32_t)(br.seed >> 32), w0, w1, w2, w3);
                        u = u53(w0, w1);
                    }
                    const double hit = fmin(fmax((p - u + 0.05) / 0.1, 0.0), 1.0);         // compute_degree_of_truth(p - u, True)
                    if (bi) keep2 *= 1.0 - hit; else keep1 *= 1.0 - hit;
                }
            }
            prev1 = cur1; prev2 = cur2;
        }
    }
    double pay = fmax(e.sign * (dev_atom(e.x, paths, D, ld, i) - e.strike), 0.0) * dev_barrier_ind(t1, e.aux[1], mx, mn);
    if (with_bridge) pay *= (t1 <= 2) ? keep1 : 1.0 - keep1;          // out: never hit; in: hit at least once
    if (t2) {
        pay *= dev_barrier_ind(t2, e.aux[2], mx, mn);
        if (with_bridge) pay *= (t2 <= 2) ? keep2 : 1.0 - keep2;
    }
    return pay / num;
}

// one draw = two uniforms in (0,1) and their Box-Muller pair (include/mcx.h "RNG contract")
// TAB: `tab` is the block's LDS copy of the Box-Muller tables (mcx_bm_load); otherwise polynomial log / sincos
// the four words of one Philox block -> first uniform + Box-Muller pair (what mcx_box_muller exposes for tests)
// GUARD = false (kf_lean.hip): the root is taken unguarded and the function returns whether this lane's draw is one of the 2^-32
// whose high word is all ones — the only ones that can round to u = 1, where the unguarded root may be NaN.  The caller then repeats
// such a draw with GUARD = true behind ONE rare branch placed after the draws of all its paths, so that the common path is a single
// basic block in which the scheduler can overlap the table reads of one path with the arithmetic of the other.
template <bool TAB = false, int BMB = 7, bool GUARD = true>
__device__ __forceinline__ bool pair_from_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, double& ua, double& z0, double& z1,
                                                const double* __restrict__ tab, const mcx_bm_coef& bc, const mcx_bm_vconst* vc = nullptr)
{
    ua = u53(w0, w1, vc);
    double s, c, r;
    if (TAB) {
        // The squared radius -2 log u is > 0 for every u < 1, and the root then needs no zero test.  One draw in 2^53 rounds to
        // u = 1 exactly ((2^53 - 1/2) 2^-53, a tie): its radius is 0 (as on the CPU path), where the table evaluation leaves a
        // rounding residual of either sign and the unguarded root may return NaN.  u = 1 needs the high word all ones: a wave that
        // holds such a lane (one in 10^8 wave-steps) takes the guarded path; the common path pays one integer compare.
        const double r2 = mcx_m2log_tab<BMB>(ua, tab, bc, vc);
        if (!GUARD) r = mcx_sqrt_gp(r2);
        else if (__builtin_expect(__any(w1 == 0xffffffffu), 0)) r = ua < 1.0 ? mcx_sqrt_g(r2) : 0.0;
        else r = mcx_sqrt_gp(r2);
        // second uniform u = ((x >> 11) + 0.5) 2^-53, x = w3:w2: its top BMB bits are the table cell, the rest is u - j/N
        // (the same conversion on the masked word, exact)
        const int j = (int)(w3 >> (32 - BMB));
        const double ur = u53(w2, w3 & ((1u << (32 - BMB)) - 1u), vc);
        mcx_sincos2pi_tab<BMB>(ur, j, tab, s, c, bc, vc);
    } else {
        r = mcx_sqrt(-2.0 * mcx_log(ua));
        mcx_sincos2pi(u53(w2, w3), s, c);
    }
    z0 = r * c;
    z1 = r * s;
    return w1 == 0xffffffffu;
}

template <bool TAB = false, int BMB = 7, bool GUARD = true>
__device__ __forceinline__ bool draw_pair(uint64_t seed, uint64_t path, uint32_t step, uint32_t draw, double& ua, double& z0, double& z1,
                                          const double* __restrict__ tab, const mcx_bm_coef& bc, const mcx_bm_vconst* vc = nullptr)
{
    uint32_t w0, w1, w2, w3;
    philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), step, draw, (uint32_t)seed, (uint32_t)(seed >> 32), w0, w1, w2, w3);
    return pair_from_words<TAB, BMB, GUARD>(w0, w1, w2, w3, ua, z0, z1, tab, bc, vc);
}

template <bool TAB = false>
__device__ __forceinline__ void draw_pair(uint64_t seed, uint64_t path, uint32_t step, uint32_t draw, double& ua, double& z0, double& z1,
                                          const double* __restrict__ tab = nullptr)
{
    const mcx_bm_coef bc = mcx_bm_coef_load();
    draw_pair<TAB>(seed, path, step, draw, ua, z0, z1, tab, bc);
}

__device__ __forceinline__ double degree_of_truth(double x, bool fuzzy, double eps)
{
    if (!fuzzy) return x > 0.0 ? 1.0 : 0.0;
    const double v = (x + eps) / (2.0 * eps);
    return fmin(fmax(v, 0.0), 1.0);
}

// the aux row of one (sub-step, slot) as the step maps read it: through scalar loads at the point of use (AuxPtr), or from values a
// kernel loaded AHEAD of the draws of the sub-step (AuxRegs: kf_lean.hip issues the scalar loads of a sub-step before its Philox /
// Box-Muller stage, whose ~100 VALU instructions then cover their latency — with one or two waves per SIMD nothing else does)
struct AuxPtr {
    const double* __restrict__ p;
    __device__ __forceinline__ double operator[](int i) const { return ldk(p + i); }
    __device__ __forceinline__ mcx_aux_drv drv() const { return ldk_struct((const mcx_aux_drv*)(p + MCX_AUX_C0)); }
};
struct AuxRegs {
    double v[MCX_AUX];
    __device__ __forceinline__ double operator[](int i) const { return v[i]; }
    __device__ __forceinline__ mcx_aux_drv drv() const { mcx_aux_drv d; d.c0 = v[MCX_AUX_C0]; d.c1 = v[MCX_AUX_C1]; d.c2 = v[MCX_AUX_C2]; return d; }
};

// one sub-step of one sub-model (reference formulas, see oracle/mcx_oracle.c for the line-by-line citations)
// KIND / SCHEME >= 0 are compile-time constants (specialised kernels: the switch folds away and only the parameters the
// model really uses stay live in SGPRs); -1 = wave-uniform run-time dispatch (generic kernels).
// SL / KA below: `mcx_slot` / `K1Args`, or their constant-address-space views (kf_lean.hip reads the kernel arguments through
// a region-local pointer into the kernarg segment, so that every field is a scalar load at its point of use)
// POS: the caller guarantees a positive CIR++ state at entry (the model's y0 > 0, cirpp.py:40, and the 1e-12 floor after every step):
// the root of the diffusion then needs no zero test
template <int KIND, int SCHEME, bool POS = false, class SL, class AUX>
__device__ __forceinline__ void step_slot(const SL& sl, int scheme_rt, int flags, double dt, double sq,
                                          const AUX& aux, double& s0, double& s1, double zc0, double zc1, double u)
{
    const auto* p = sl.p;
    const int scheme = SCHEME >= 0 ? SCHEME : scheme_rt;
    switch (KIND >= 0 ? KIND : sl.kind) {
    case MCX_MODEL_BS:
        if (scheme == MCX_SCHEME_ANALYTICAL) {
            s0 = s0 * mcx_exp(aux[0] + (zc0 - aux[1]));                       // black_scholes.py:61-67
        } else {
            // black_scholes.py:79-85, S + (r S dt + sigma S sqrt(dt) z) = S + S (r dt + sigma sqrt(dt) z); the step constants
            // r dt, sigma sqrt(dt) come from the derived entries of the step table (mcx_sim_create)
            const mcx_aux_drv dc = aux.drv();
            s0 = fma(s0, fma(dc.c2, zc0, dc.c0), s0);
        }
        break;
    case MCX_MODEL_VASICEK: {
        const double r = s0;
        s1 = s1 + r * dt;                                                 // left-endpoint integral, vasicek.py:80/107
        if (scheme == MCX_SCHEME_ANALYTICAL) s0 = (p[2] + (r - p[2]) * aux[0]) + zc0;
        else {
            // r - (a dt) r + sigma sqrt(dt) z + a theta dt as three accumulations INTO the state register (one scalar operand each):
            // the state array is a register tuple indexed by the date programs, a value computed elsewhere would be copied into it
            const mcx_aux_drv dc = aux.drv();
            s0 = fma(dc.c1, r, s0);
            s0 = fma(dc.c2, zc0, s0);
            s0 = s0 + dc.c0;
        }
        break;
    }
    case MCX_MODEL_HW: {
        const double r = s0;
        s1 = s1 + r * dt;
        if (scheme == MCX_SCHEME_ANALYTICAL) s0 = r * aux[0] + aux[1] + zc0;
        else s0 = r + (aux[0] - p[3] * r) * dt + p[1] * sq * zc0;
        break;
    }
    case MCX_MODEL_CIRPP: {                                               // cirpp.py:188-198
        const double y = s0;
        const double sy = POS ? mcx_sqrt_gp(y) : mcx_sqrt_g(y);   // = sqrt(clamp(y, 0)) of cirpp.py:194: mcx_sqrt_g returns 0 for y <= 0
        // y - (kappa dt) y + (sigma sqrt(dt)) sqrt(y) z + kappa theta dt, accumulated into the state register (see VASICEK)
        const mcx_aux_drv dc = aux.drv();
        s1 = fma(y + aux[0], dt, s1);
        s0 = fma(dc.c1, y, s0);
        s0 = fma(dc.c2 * sy, zc0, s0);
        s0 = s0 + dc.c0;
        s0 = fmax(s0, 1e-12);
        break;
    }
    case MCX_MODEL_CIRPP_DET:                                             // cirpp.py:155-172
        s1 = s1 + aux[0] * dt;
        s0 = aux[1];
        break;
    case MCX_MODEL_S2F: {                                                 // schwartz_two_factor.py:147-196; registers (x, y)
        const double x = s0, y = s1;
        if (scheme == MCX_SCHEME_ANALYTICAL) {
            s0 = x * aux[0] + zc0;                                  // short factor: mean reversion + w_x
            s1 = y + p[3] * dt + zc1;                                     // long factor: drift + w_y
        } else {
            s0 = x - p[1] * x * dt + p[2] * sq * zc0;
            s1 = y + p[3] * dt + p[4] * sq * zc1;
        }
        break;
    }
    case MCX_MODEL_HESTON: {
        const double logS = s0, v = s1;
        const double sigma = p[1], rate = p[2], kappa = p[4], theta = p[5];
        // (roots and reciprocals: v_rsq_f64 / v_rcp_f64 seeds + Newton steps, <= 1 ulp, instead of the IEEE-complete library
        //  sqrt and division sequences — five roots and six divisions per QE step were ~170 of its ~270 f64 instructions)
        if (scheme == MCX_SCHEME_EULER) {                                 // heston.py:109-121
            const double sv = mcx_sqrt(fmax(v, 0.0));
            s0 = logS + (rate - 0.5 * v) * dt + sv * sq * zc0;
            s1 = fmax(v + kappa * (theta - v) * dt + sigma * sv * sq * zc1, 0.0);
        } else {                                                          // heston.py:161-253 (Andersen QE)
            const double eps = 1e-12;
            const bool fuzzy = (flags & MCX_FLAG_SMOOTHING) != 0;
            // The seven reciprocals of the step in three groups of simultaneously known denominators, each group from ONE v_rcp_f64 of
            // the product (1/x = (1/(x y z)) y z: the seed + Newton sequence costs ~7.5 multiply-adds, the extra products one each),
            // and sqrt(2/psi) sqrt(t) as one root of the product: ~30 of the step's ~290 VALU instructions.  Every denominator keeps
            // the reference's eps where the reference has it (heston.py:185-239); the products stay far inside the double range
            // (>= 1e-36, <= ~1e24 with the eps floors).
            const double m = theta + (v - theta) * aux[0];
            const double s2 = v * aux[6] + aux[7];
            const double omu = fmax(1.0 - u, eps);
            const double d1 = m * m + eps, d5 = m + eps;
            const double d15 = d1 * d5;
            const double ra = mcx_rcp(d15 * omu);
            const double r_omu = ra * d15, ra6 = ra * omu;
            const double r1 = ra6 * d5, r5 = ra6 * d1;                           // 1/(m^2 + eps), 1/(m + eps)
            const double psi = s2 * r1;
            const double d2 = psi + eps, d4 = psi + 1.0;
            const double rb = mcx_rcp(d2 * d4);
            const double invpsi = rb * d4, r4 = rb * d2;                        // 1/(psi + eps), 1/(psi + 1)
            const double t = fmax(2.0 * invpsi - 1.0, 0.0);
            // (roots that feed the Monte-Carlo increment: seed + one coupled Goldschmidt step, 1-2 ulp, as for the CIR diffusion)
            // The smoothed scheme can take the variance — and with it s2 and psi — below zero; the reference then
            // takes torch.sqrt of the negative 2/psi (heston.py:203) and the path is NaN from there on (torch.clamp keeps a NaN).
            // Reproduced: the merged root below is 0 there (t = 0), so the NaN is put in by hand, and b2's clamp keeps it.
            double root = mcx_sqrt_g(2.0 * invpsi * t);
            root = invpsi < 0.0 ? __builtin_nan("") : root;
            const double b2u = 2.0 * invpsi - 1.0 + root;
            const double b2 = b2u < 0.0 ? 0.0 : b2u;
            const double b = mcx_sqrt_g(b2);
            const double pp = fmin(fmax((psi - 1.0) * r4, 0.0), 1.0 - 1e-6);
            const double beta = (1.0 - pp) * r5;
            const double d3 = 1.0 + b2, d7 = beta + eps;
            const double rc = mcx_rcp(d3 * d7);
            const double a = m * (rc * d7);
            const double v1 = a * (b + zc1) * (b + zc1);
            const double omp = fmax(1.0 - pp, eps);
            const double v_tail = mcx_log(omp * r_omu) * (rc * d3);
            const double v2 = degree_of_truth(u - pp, fuzzy, 0.3) * v_tail;
            const double w = degree_of_truth(psi - 1.5, fuzzy, 0.5);
            const double vn = (1.0 - w) * v1 + w * v2;
            const double var_int = fmax(aux[4] * v + aux[5] * vn, 0.0);
            const double vol = mcx_sqrt_gp(fmax(var_int, eps));
            s0 = logS + rate * dt + aux[1] + aux[2] * v + aux[3] * vn + vol * zc0;
            s1 = vn;
        }
        break;
    }
    default: break;
    }
}


// ---- model signatures: compile-time (kinds, scheme) of the hot configurations ---------------------------------------
enum { SIG_GENERIC = 0, SIG_VAS_CIR_E = 1, SIG_BS_A = 2, SIG_BS_E = 3, SIG_HESTON_QE = 4, SIG_HESTON_E = 5,
       SIG_VAS_E = 6, SIG_VAS_A = 7, SIG_BS_VAS_CIRDET_E = 8 };

__host__ __device__ constexpr int sig_kind(int sig, int slot)
{
    return sig == SIG_VAS_CIR_E ? (slot == 0 ? MCX_MODEL_VASICEK : MCX_MODEL_CIRPP)
         : (sig == SIG_BS_A || sig == SIG_BS_E) ? MCX_MODEL_BS
         : (sig == SIG_HESTON_QE || sig == SIG_HESTON_E) ? MCX_MODEL_HESTON
         : (sig == SIG_VAS_E || sig == SIG_VAS_A) ? MCX_MODEL_VASICEK
         : sig == SIG_BS_VAS_CIRDET_E ? (slot == 0 ? MCX_MODEL_BS : slot == 1 ? MCX_MODEL_VASICEK : MCX_MODEL_CIRPP_DET)
         : -1;
}
__host__ __device__ constexpr int sig_scheme(int sig)
{
    return (sig == SIG_VAS_CIR_E || sig == SIG_BS_E || sig == SIG_HESTON_E || sig == SIG_VAS_E || sig == SIG_BS_VAS_CIRDET_E) ? MCX_SCHEME_EULER
         : (sig == SIG_BS_A || sig == SIG_VAS_A) ? MCX_SCHEME_ANALYTICAL
         : sig == SIG_HESTON_QE ? MCX_SCHEME_QE : -1;
}
__host__ __device__ constexpr bool sig_is_bs(int sig, int slot) { return sig_kind(sig, slot) == MCX_MODEL_BS; }

// signature of a simulation descriptor (host)
static inline int mcx_sim_signature(const mcx_sim_desc& d)
{
    auto k = [&](int s) { return d.slots[s].kind; };
    if (d.n_slots == 2 && k(0) == MCX_MODEL_VASICEK && k(1) == MCX_MODEL_CIRPP && d.scheme == MCX_SCHEME_EULER) return SIG_VAS_CIR_E;
    if (d.n_slots == 3 && k(0) == MCX_MODEL_BS && k(1) == MCX_MODEL_VASICEK && k(2) == MCX_MODEL_CIRPP_DET && d.scheme == MCX_SCHEME_EULER)
        return SIG_BS_VAS_CIRDET_E;
    if (d.n_slots == 1) {
        if (k(0) == MCX_MODEL_BS) return d.scheme == MCX_SCHEME_ANALYTICAL ? SIG_BS_A : d.scheme == MCX_SCHEME_EULER ? SIG_BS_E : SIG_GENERIC;
        if (k(0) == MCX_MODEL_HESTON) return d.scheme == MCX_SCHEME_QE ? SIG_HESTON_QE : d.scheme == MCX_SCHEME_EULER ? SIG_HESTON_E : SIG_GENERIC;
        if (k(0) == MCX_MODEL_VASICEK) return d.scheme == MCX_SCHEME_EULER ? SIG_VAS_E : d.scheme == MCX_SCHEME_ANALYTICAL ? SIG_VAS_A : SIG_GENERIC;
    }
    return SIG_GENERIC;
}

// compile-time recursion over the slots (the slot index must be a constant expression for the signature lookup)
__device__ __forceinline__ AuxPtr aux_of_slot(const double* __restrict__ ax, int s) { return AuxPtr{ax + s * MCX_AUX}; }
template <int NSLOT>
__device__ __forceinline__ const AuxRegs& aux_of_slot(const AuxRegs (&ax)[NSLOT], int s) { return ax[s]; }

template <int NSLOT, int NZ, int SIG, int S, bool POS = false, class KA, class AX>
__device__ __forceinline__ void step_slots(const KA& k, const mcx_step& sp, const AX& ax,
                                           double (&reg)[2 * NSLOT], const double (&zc)[NZ], double u)
{
    if constexpr (S < NSLOT) {
        // ModelConfig only hosts sub-models with simulation_dim == 1 (model_config.py:106-107); Heston runs alone
        const double zc0 = (NSLOT == 1) ? zc[0] : zc[S < NZ ? S : 0];
        const double zc1 = (NSLOT == 1 && NZ > 1) ? zc[NZ > 1 ? 1 : 0] : 0.0;
        step_slot<sig_kind(SIG, S), sig_scheme(SIG), POS>(k.slots[S], k.scheme, k.flags | k.slots[S].flags, sp.dt, sp.sqrt_dt,
                                                      aux_of_slot(ax, S), reg[2 * S], reg[2 * S + 1], zc0, zc1, u);
        step_slots<NSLOT, NZ, SIG, S + 1, POS>(k, sp, ax, reg, zc, u);
    }
}

// the draws of one sub-step of a lane's path: NZ standard normals (+ the uniform of the Heston QE step).  They depend on the counter
// (path, step) only, never on the state: a kernel may draw several sub-steps ahead (kf_lean.hip, small path counts)
// seed / bc: the Philox key and the Box-Muller coefficients (SGPRs) of the calling code region (mcx_math.h "region zero")
// GUARD = false: see pair_from_words; returns true when the lane's draws must be repeated with GUARD = true
template <int NZ, bool INJECT, int SIG, int BMB = 7, bool GUARD = true, class KA>
__device__ __forceinline__ bool sim_draw(const KA& k, int step, uint64_t path, int64_t i, double (&z)[NZ], double& u,
                                         const double* __restrict__ tab, uint64_t seed, const mcx_bm_coef& bc, const mcx_bm_vconst* vc = nullptr)
{
    u = 0.0;
    bool rare = false;
    if (INJECT) {
#pragma unroll
        for (int j = 0; j < NZ; ++j) z[j] = k.inject_z[((int64_t)step * NZ + j) * k.ld + i];
        if (k.n_uniform) u = k.inject_u[(int64_t)step * k.ld + i];
    } else {
        double ua;
#pragma unroll
        for (int q = 0; q < (NZ + 1) / 2; ++q) {
            double z0, z1;
            rare |= draw_pair<true, BMB, GUARD>(seed, path, (uint32_t)step, (uint32_t)q, ua, z0, z1, tab, bc, vc);
            z[2 * q] = z0;
            if (2 * q + 1 < NZ) z[2 * q + 1] = z1;
        }
        if (sig_scheme(SIG) == MCX_SCHEME_QE || (sig_scheme(SIG) < 0 && k.n_uniform)) {
            double z0, z1;
            draw_pair<false>(seed, path, (uint32_t)step, (uint32_t)((NZ + 1) / 2), u, z0, z1, nullptr, bc);
        }
    }
    return rare;
}

// ---- the draws of PPL paths, staged for latency (kf_lean.hip) -------------------------------------------------------------------
// The same arithmetic as draw_pair<true, BMB, false> per path, in an order the backend is made to keep (sched_barrier): (A) the
// Philox blocks of all paths, round by round; (B) the table cells of both Box-Muller functions of all paths and their LDS reads;
// (C) everything that needs no table value — the second uniform's remainder, the angle polynomials, the exponent; (D) the rest.
// Left to itself the backend emits path after path, each `ds_read` followed at once by the wait for it: with one or two waves per
// SIMD (a GPU's share of a strong-scaled run) nothing covers the LDS latency; here stage C does.
// (a "chain" is one (path, sub-step) counter: the PPL paths of a lane at one sub-step, or one path at several sub-steps ahead)
template <int PPL>
__device__ __forceinline__ void philox4x32_10_n(const uint64_t (&path)[PPL], const uint32_t (&step)[PPL], uint32_t draw, uint64_t seed,
                                                uint32_t (&o0)[PPL], uint32_t (&o1)[PPL], uint32_t (&o2)[PPL], uint32_t (&o3)[PPL])
{
    uint32_t c0[PPL], c1[PPL], c2[PPL], c3[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) { c0[q] = (uint32_t)path[q]; c1[q] = (uint32_t)(path[q] >> 32); c2[q] = step[q]; c3[q] = draw; }
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0[q];
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2[q];
            const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1[q], k0, 0x96);
            const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3[q], k1, 0x96);
            c1[q] = (uint32_t)p1; c3[q] = (uint32_t)p0; c0[q] = n0; c2[q] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int q = 0; q < PPL; ++q) { o0[q] = c0[q]; o1[q] = c1[q]; o2[q] = c2[q]; o3[q] = c3[q]; }
}

// returns whether some path's first uniform may have rounded to 1 (high word all ones): the caller repeats the draws guarded
template <int PPL, int BMB>
__device__ __forceinline__ bool draw_pairs_staged(uint64_t seed, const uint64_t (&path)[PPL], const uint32_t (&step)[PPL], uint32_t draw,
                                                  double (&z0)[PPL], double (&z1)[PPL], const double* __restrict__ tab,
                                                  const mcx_bm_coef& C, const mcx_bm_vconst& vc)
{
    constexpr int N = 1 << BMB, LOG_TERMS = mcx_bm_shape<BMB>::LOG_TERMS;
    uint32_t w0[PPL], w1[PPL], w2[PPL], w3[PPL];
    philox4x32_10_n<PPL>(path, step, draw, seed, w0, w1, w2, w3);
    double m[PPL], ed[PPL], ps[PPL], pc[PPL];
    int e[PPL];
    mcx_d2 tc[PPL], sc[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {                                     // (B) cells and table reads: mcx_m2log_tab, mcx_sincos2pi_tab
        const double ua = u53(w0[q], w1[q], &vc);
        m[q] = __builtin_amdgcn_frexp_mant(ua);
        e[q] = __builtin_amdgcn_frexp_exp(ua);
        const int jl = (__double2hiint(m[q]) >> (20 - BMB)) & (N - 1);
        tc[q] = ((const mcx_d2*)tab)[jl];
        sc[q] = ((const mcx_d2*)(tab + 2 * N))[(int)(w3[q] >> (32 - BMB))];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {                                     // (C) no table value needed
        const double ur = u53(w2[q], w3[q] & ((1u << (32 - BMB)) - 1u), &vc);
        const double d = fma(ur, 6.28318530717958647692, vc.trig_off);
        const double d2 = d * d;
        double qs, qc;
        if constexpr (mcx_bm_shape<BMB>::TRIG_TERMS == 3) {
            qs = fma(fma(vc.sin_head, d2, C.c[7]), d2, C.c[6]);
            qc = fma(fma(vc.cos_head, d2, C.c[10]), d2, C.c[9]);
        } else {
            qs = fma(vc.sin_head, d2, C.c[6]);
            qc = fma(vc.cos_head, d2, C.c[9]);
        }
        ps[q] = fma(d * d2, qs, d);
        pc[q] = fma(d2, qc, 1.0);
        ed[q] = (double)e[q];
    }
    __builtin_amdgcn_sched_barrier(0);
    bool rare = false;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {                                     // (D) radius and rotation
        const double s = fma(m[q], tc[q].x, 2.0);
        double p = vc.log_head;
#pragma unroll
        for (int k = LOG_TERMS - 2; k >= 0; --k) p = fma(p, s, C.c[k]);
        const double r2 = fma(ed[q], -2.0 * 6.93147180559945309417e-01, tc[q].y) + fma(s * s, p, s);
        const double r = mcx_sqrt_gp(r2);
        const double sn = fma(sc[q].x, pc[q], sc[q].y * ps[q]);
        const double cs = fma(-sc[q].x, ps[q], sc[q].y * pc[q]);
        z0[q] = r * cs;
        z1[q] = r * sn;
        rare |= w1[q] == 0xffffffffu;
    }
    return rare;
}

// the draws of PPL chains (sim_draw per chain, the normals through draw_pairs_staged)
template <int PPL, int NZ, int SIG, int BMB, class KA>
__device__ __forceinline__ bool sim_draw_n(const KA& k, const uint32_t (&step)[PPL], const uint64_t (&path)[PPL], double (&z)[PPL][NZ], double (&u)[PPL],
                                           const double* __restrict__ tab, uint64_t seed, const mcx_bm_coef& bc, const mcx_bm_vconst& vc)
{
    bool rare = false;
#pragma unroll
    for (int dq = 0; dq < (NZ + 1) / 2; ++dq) {
        double a[PPL], b[PPL];
        rare |= draw_pairs_staged<PPL, BMB>(seed, path, step, (uint32_t)dq, a, b, tab, bc, vc);
#pragma unroll
        for (int q = 0; q < PPL; ++q) { z[q][2 * dq] = a[q]; if (2 * dq + 1 < NZ) z[q][2 * dq + 1] = b[q]; }
    }
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        u[q] = 0.0;
        if (sig_scheme(SIG) == MCX_SCHEME_QE || (sig_scheme(SIG) < 0 && k.n_uniform)) {
            double t0, t1;
            draw_pair<false>(seed, path[q], step[q], (uint32_t)((NZ + 1) / 2), u[q], t0, t1, nullptr, bc);
        }
    }
    return rare;
}

// the state update of one sub-step from its draws: Cholesky, per-slot maps.  reg[2s], reg[2s+1] = state of slot s.
// Returns the timeline date stored after this sub-step (mcx_step::store_idx, -1: none) — from the SAME record load as dt, so that
// the loop condition of the caller does not wait for a second scalar load at the end of every sub-step.
template <int NSLOT, int NZ, int SIG, bool POS = false, class KA>
__device__ __forceinline__ int sim_apply(const KA& k, int step, double (&reg)[2 * NSLOT], const double (&z)[NZ], double u)
{
    const mcx_step sp = ldk_struct(&k.steps[step]);    // wave-uniform -> scalar loads
    double zc[NZ];
    // EULER / QE correlate with the factor of a CORRELATION matrix (model.py:66-73): its first entry is sqrt(1) = 1 exactly, and
    // there is ONE factor for the whole run (mcx_sim_create checks chol_idx == 0): its address does not wait for the step record
    constexpr bool UNIT_L00 = sig_scheme(SIG) == MCX_SCHEME_EULER || sig_scheme(SIG) == MCX_SCHEME_QE;
    const double* __restrict__ L = UNIT_L00 ? k.chol : k.chol + (int64_t)sp.chol_idx * NZ * NZ;     // model.py:48  z @ chol.T
#pragma unroll
    for (int r = 0; r < NZ; ++r) {
        double acc = (UNIT_L00 && r == 0) ? z[0] : ldk(L + r * NZ) * z[0];
#pragma unroll
        for (int c = 1; c <= r; ++c) acc = fma(ldk(L + r * NZ + c), z[c], acc);
        zc[r] = acc;
    }
    const double* __restrict__ ax = k.aux + (int64_t)step * NSLOT * MCX_AUX;
    step_slots<NSLOT, NZ, SIG, 0, POS>(k, sp, ax, reg, zc, u);
    return sp.store_idx;
}

// The wave-uniform data of one sub-step, loaded AHEAD of its draws (scalar loads issued before the Philox / Box-Muller stage):
// the step record, the Cholesky factor and the aux rows.  Entries the compile-time model signature does not read are dead loads
// and disappear.  Under ANALYTICAL the factor depends on the step record (one factor per distinct dt): it stays a late load.
template <int NSLOT, int NZ>
struct StepData {
    mcx_step sp;
    double L[NZ * NZ];
    AuxRegs ax[NSLOT];
};
template <int NSLOT, int NZ, int SIG, class KA>
__device__ __forceinline__ StepData<NSLOT, NZ> sim_step_load(const KA& k, int step)
{
    StepData<NSLOT, NZ> d;
    d.sp = ldk_struct(&k.steps[step]);
    constexpr bool UNIT_L00 = sig_scheme(SIG) == MCX_SCHEME_EULER || sig_scheme(SIG) == MCX_SCHEME_QE;
    const double* __restrict__ L = UNIT_L00 ? k.chol : k.chol + (int64_t)d.sp.chol_idx * NZ * NZ;
#pragma unroll
    for (int q = 0; q < NZ * NZ; ++q) d.L[q] = ldk(L + q);
    // a slot's row as three records — entries (0,1), (2,3) and the derived step constants (4..7) — so that a model reads its
    // constants through one or two wide scalar loads (seven separate loads with their address arithmetic were a quarter of the
    // SALU work of a sub-step); records nobody reads are dead loads
    struct P2 { double v[2]; };
    struct P4 { double v[4]; };
    static_assert(MCX_AUX == 8, "aux row layout");
    const double* __restrict__ ax = k.aux + (int64_t)step * NSLOT * MCX_AUX;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const P2 a = ldk_struct((const P2*)(ax + s * MCX_AUX)), b = ldk_struct((const P2*)(ax + s * MCX_AUX + 2));
        const P4 c = ldk_struct((const P4*)(ax + s * MCX_AUX + 4));
        d.ax[s].v[0] = a.v[0]; d.ax[s].v[1] = a.v[1]; d.ax[s].v[2] = b.v[0]; d.ax[s].v[3] = b.v[1];
#pragma unroll
        for (int q = 0; q < 4; ++q) d.ax[s].v[4 + q] = c.v[q];
    }
    return d;
}
template <int NSLOT, int NZ, int SIG, bool POS = false, class KA>
__device__ __forceinline__ int sim_apply_loaded(const KA& k, const StepData<NSLOT, NZ>& d, double (&reg)[2 * NSLOT], const double (&z)[NZ], double u)
{
    double zc[NZ];
    constexpr bool UNIT_L00 = sig_scheme(SIG) == MCX_SCHEME_EULER || sig_scheme(SIG) == MCX_SCHEME_QE;
#pragma unroll
    for (int r = 0; r < NZ; ++r) {
        double acc = (UNIT_L00 && r == 0) ? z[0] : d.L[r * NZ] * z[0];
#pragma unroll
        for (int c = 1; c <= r; ++c) acc = fma(d.L[r * NZ + c], z[c], acc);
        zc[r] = acc;
    }
    step_slots<NSLOT, NZ, SIG, 0, POS>(k, d.sp, d.ax, reg, zc, u);
    return d.sp.store_idx;
}

// one sub-step of the whole model for a lane: draws, Cholesky, per-slot maps
template <int NSLOT, int NZ, bool INJECT, int SIG, int BMB = 7, bool POS = false, class KA>
__device__ __forceinline__ void sim_substep(const KA& k, int step, uint64_t path, int64_t i, double (&reg)[2 * NSLOT],
                                            const double* __restrict__ tab, uint64_t seed, const mcx_bm_coef& bc, const mcx_bm_vconst* vc = nullptr)
{
    double z[NZ], u;
    sim_draw<NZ, INJECT, SIG, BMB>(k, step, path, i, z, u, tab, seed, bc, vc);
    sim_apply<NSLOT, NZ, SIG, POS>(k, step, reg, z, u);
}

template <int NSLOT, int NZ, bool INJECT, int SIG>
__device__ __forceinline__ void sim_substep(const K1Args& k, int step, uint64_t path, int64_t i, double (&reg)[2 * NSLOT],
                                            const double* __restrict__ tab)
{
    const mcx_bm_coef bc = mcx_bm_coef_load();
    sim_substep<NSLOT, NZ, INJECT, SIG>(k, step, path, i, reg, tab, k.seed, bc);
}

// state columns of a slot kind: Black-Scholes 1, Schwartz two-factor 3 (log S is derived: log F0(t) + x + y), the others 2
__host__ __device__ constexpr int mcx_kind_state_dim(int kind) { return kind == MCX_MODEL_BS ? 1 : kind == MCX_MODEL_S2F ? 3 : 2; }

// i: the lane's path (used only when the run starts from a per-path state tensor)
template <int NSLOT, int SIG, class KA>
__device__ __forceinline__ void sim_init_state(const KA& k, double (&reg)[2 * NSLOT], int64_t i = 0)
{
    const double* __restrict__ ip = k.init_paths;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int kind = sig_kind(SIG, s) >= 0 ? sig_kind(SIG, s) : k.slots[s].kind;
        const int c = k.slots[s].state_off + (kind == MCX_MODEL_S2F ? 1 : 0);      // S2F registers are (x, y) = columns 1, 2
        if (ip) {
            reg[2 * s] = ip[(int64_t)c * k.ld + i];
            reg[2 * s + 1] = kind == MCX_MODEL_BS ? 0.0 : ip[(int64_t)(c + 1) * k.ld + i];
        } else {
            reg[2 * s] = k.init_state[c];
            reg[2 * s + 1] = kind == MCX_MODEL_BS ? 0.0 : k.init_state[c + 1];
        }
    }
}

// ax: aux row of the sub-step that reached date t (nullptr for the dates that hold the initial state)
template <int NSLOT, int SIG, class KA>
__device__ __forceinline__ void sim_store_state(const KA& k, int t, int64_t i, const double (&reg)[2 * NSLOT],
                                                const double* __restrict__ ax = nullptr)
{
    const int D = k.n_state;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int kind = sig_kind(SIG, s) >= 0 ? sig_kind(SIG, s) : k.slots[s].kind;
        const int c = k.slots[s].state_off;
        if (kind == MCX_MODEL_S2F) {
            // log S(t) = log F0(t) + x + y (schwartz_two_factor.py:170-171, 194-195); with a per-path start the caller's log S
            const double logF = ax ? ldk(ax + s * MCX_AUX + 1) : k.slots[s].p[6];
            double logS = (logF + reg[2 * s]) + reg[2 * s + 1];
            if (!ax && k.init_paths) logS = k.init_paths[(int64_t)c * k.ld + i];
            k.paths[((int64_t)t * D + c) * k.ld + i] = logS;
            k.paths[((int64_t)t * D + c + 1) * k.ld + i] = reg[2 * s];
            k.paths[((int64_t)t * D + c + 2) * k.ld + i] = reg[2 * s + 1];
            continue;
        }
        k.paths[((int64_t)t * D + c) * k.ld + i] = reg[2 * s];
        if (kind != MCX_MODEL_BS) k.paths[((int64_t)t * D + c + 1) * k.ld + i] = reg[2 * s + 1];
    }
}
