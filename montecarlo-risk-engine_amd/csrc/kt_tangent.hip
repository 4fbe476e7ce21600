// kt_tangent.hip — forward-mode (tangent) path kernel for PV sensitivities of European options (BASELINE configs 2, 4).
//
// The reference tapes the WHOLE simulation (every sub-step of every path) and calls torch.autograd.grad per metric value
// (controller/controller.py:609-627): memory grows with N x steps (4 M x 500 does not fit on the CPU reference) and the
// Heston QE tape makes path generation 18x slower.  Here each lane propagates dual numbers (value + d/d theta_j for the
// P model parameters) through the same step maps in registers: no tape, memory independent of the number of steps.
//
// Subgradient conventions follow torch exactly (they decide the reference's numbers):
//   torch.clamp(x, min, max): gradient passes iff min <= x <= max;  torch.maximum(a, b): a > b -> a, tie -> 1/2 each;
//   (x > 0).float(): zero gradient;  degree_of_truth (fuzzy): clamp((x+eps)/(2eps), 0, 1).
#include "mcx_dual.h"

namespace {

// the parameter-only factors of one QE step (heston.py:161-253): functions of (kappa, theta, sigma, rho, rate, dt), the same for
// every path and — with an evenly spaced timeline — for every step.  Built on the host as dual numbers (value + 7 tangents),
// one record per distinct dt; the kernel reads them with scalar loads instead of redoing ~20 dual operations per path and step.
// Tangent order [sigma, kappa, theta, v0, rho]; the variance-chain factors carry the first four (kt_heston).
struct QEConst {
    Dual<4> E;        // exp(-kappa dt)
    Dual<4> A1;       // sigma^2 E (1 - E) / kappa                  (s2 = v A1 + A2)
    Dual<4> A2;       // theta sigma^2 (1 - E)^2 / (2 kappa)
    Dual<5> base;     // rate dt + K0,  K0 = -(rho kappa theta / sigma) dt      (the rate tangent is the elapsed time: kt_heston)
    Dual<5> K1;       // (kappa rho / sigma - 1/2) dt - rho / sigma      (gamma1 = 1, gamma2 = 0)
    Dual<5> K2;       // rho / sigma
    Dual<5> K3;       // (1 - rho^2) dt
    double pad[3];    // 40 doubles: dword-aligned records of a power-of-two friendly size
};

struct KTArgs {
    K1Args k1;
    const QEConst* __restrict__ qe;             // [n distinct dt]
    const int32_t* __restrict__ qe_idx;         // [n_steps] -> record of the step's dt
    const mcx_tangent_option* __restrict__ opts;
    double* __restrict__ cfs;
    double* __restrict__ dcfs;
    int64_t ld_out;
    int32_t n_opts, n_ns;
};

// Black-Scholes: params [spot, sigma, rate], state S
// NNS: netting sets with accumulators (1 for a single-netting-set book: the BASELINE configs; MCX_FUSED_MAX_NS otherwise).  The
// payoff block is force-inlined at its two call sites: left as an out-of-line lambda it took the accumulator arrays by address
// and the backend kept them in scratch memory (48 B per lane here, 240 B in kt_heston).
template <bool INJECT, int NNS>
__global__ __launch_bounds__(MCX_BLOCK) void kt_bs(const KTArgs a)
{
    constexpr int P = 3;
    const K1Args& k = a.k1;
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    __shared__ double bm_lds[INJECT ? 2 : MCX_BM_LDS_DOUBLES];        // table-driven Box-Muller (mcx_math.h)
    const double* tab = nullptr;
    if (!INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    if (i >= k.n) return;
    const double* p = k.slots[0].p;
    const Dual<P> sigma = dseed<P>(p[1], 1), rate = dseed<P>(p[2], 2);
    Dual<P> S = dseed<P>(p[0], 0);
    double acc[NNS], dacc[NNS][P];
#pragma unroll
    for (int n = 0; n < NNS; ++n) { acc[n] = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) dacc[n][j] = 0.0; }
    auto on_date = [&](int t) __attribute__((always_inline)) {
        for (int q = 0; q < a.n_opts; ++q) {
            const mcx_tangent_option o = ldk_struct(&a.opts[q]);
            if (o.t_idx != t) continue;
            // payoff = torch.maximum(sign*(S-K), 0) / numeraire   (european_option.py:45-68)
            const double x = o.sign * (S.v - o.strike);
            const double w = x > 0.0 ? 1.0 : (x == 0.0 ? 0.5 : 0.0);
            const double pay = fmax(x, 0.0);
            const double inv = 1.0 / o.numeraire;
#pragma unroll
            for (int n = 0; n < NNS; ++n) {
                const double sel = (NNS == 1 || n == o.netting_set) ? inv : 0.0;        // (a select, not a branch: static register indices)
                acc[n] += pay * sel;
#pragma unroll
                for (int j = 0; j < P; ++j) dacc[n][j] += w * o.sign * S.d[j] * sel;
                dacc[n][2] -= pay * sel * inv * o.dnum_drate;
            }
        }
    };
    for (int t = 0; t < k.n_initial_store; ++t) on_date(t);
    const uint64_t path = k.path_offset + (uint64_t)i;
    for (int step = 0; step < k.n_steps; ++step) {
        const mcx_step sp = ldk_struct(&k.steps[step]);
        double z;
        if (INJECT) z = k.inject_z[(int64_t)step * k.ld + i];
        else { double ua, z1; draw_pair<true>(k.seed, path, (uint32_t)step, 0u, ua, z, z1, tab); }
        if (k.scheme == MCX_SCHEME_ANALYTICAL) {
            // w = chol(sigma^2 dt) z = sigma * (L / sigma) * z : the Cholesky factor carries the graph to sigma (model.py:56-64)
            const double sq_chol = ldk(k.chol + sp.chol_idx) / p[1];
            const Dual<P> arg = rate * sp.dt + (sigma * (sq_chol * z) - sigma * sigma * (0.5 * sp.dt));
            S = S * dexp(arg);                                                    // black_scholes.py:61-67
        } else {
            S = S + (rate * S * sp.dt + sigma * S * (sp.sqrt_dt * z));            // black_scholes.py:79-85
        }
        if (sp.store_idx >= 0) on_date(sp.store_idx);
    }
#pragma unroll
    for (int n = 0; n < NNS; ++n) {          // compile-time indices: the accumulators stay in registers
        if (n < a.n_ns) {
            a.cfs[(int64_t)n * a.ld_out + i] = acc[n];
#pragma unroll
            for (int j = 0; j < P; ++j) a.dcfs[((int64_t)n * P + j) * a.ld_out + i] = dacc[n][j];
        }
    }
}

// Heston: params [spot, sigma_v, rate, rho, kappa, theta, v0], state (log S, v).
// The tangent sets are sparse and the kernel carries only what can be non-zero:
//   * d logS / d spot = 1 / spot and d logS / d rate = elapsed time are constants of the path (rate and spot enter the log-price
//     additively and the variance not at all): applied at the payoff, never propagated;
//   * the variance depends on (sigma, kappa, theta, v0) only under QE — and on rho as well under Euler, where the variance
//     draw is the correlated one (heston.py:109-121): PV = 4 / 5 tangents through the whole variance chain;
//   * log S carries those and rho: PL = 5.
// Local tangent order [sigma, kappa, theta, v0, rho].
template <int PA, int PB>
__device__ __forceinline__ Dual<PB> dext(const Dual<PA>& a)
{
    Dual<PB> r;
    r.v = a.v;
#pragma unroll
    for (int j = 0; j < PB; ++j) r.d[j] = j < PA ? a.d[j] : 0.0;
    return r;
}

template <bool INJECT, bool QE, int NNS>
__global__ __launch_bounds__(MCX_BLOCK) void kt_heston(const KTArgs a)
{
    constexpr int P = 7, PV = QE ? 4 : 5, PL = 5;
    const K1Args& k = a.k1;
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    __shared__ double bm_lds[INJECT ? 2 : MCX_BM_LDS_DOUBLES];        // table-driven Box-Muller (mcx_math.h)
    const double* tab = nullptr;
    if (!INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    if (i >= k.n) return;
    const double* p = k.slots[0].p;
    const double inv_spot = 1.0 / p[0];
    const Dual<PV> sigma = dseed<PV>(p[1], 0), kappa = dseed<PV>(p[4], 1), theta = dseed<PV>(p[5], 2);
    const Dual<PL> rho = dseed<PL>(p[3], 4);
    const double rate = p[2];
    Dual<PL> logS = dconst<PL>(mcx_log(p[0]));
    Dual<PV> v = dseed<PV>(p[6], 3);
    double elapsed = 0.0;                                              // d logS / d rate
    const bool fuzzy = ((k.flags | k.slots[0].flags) & MCX_FLAG_SMOOTHING) != 0;
    double acc[NNS], dacc[NNS][P];
#pragma unroll
    for (int n = 0; n < NNS; ++n) { acc[n] = 0.0;
#pragma unroll
        for (int j = 0; j < P; ++j) dacc[n][j] = 0.0; }
    auto on_date = [&](int t) __attribute__((always_inline)) {
        for (int q = 0; q < a.n_opts; ++q) {
            const mcx_tangent_option o = ldk_struct(&a.opts[q]);
            if (o.t_idx != t) continue;
            const double Sv = mcx_exp(logS.v);                                     // heston.py:258-260
            // d S / d param in the reference's order [spot, sigma_v, rate, rho, kappa, theta, v0]
            const double dS[P] = {Sv * inv_spot, Sv * logS.d[0], Sv * elapsed, Sv * logS.d[4], Sv * logS.d[1], Sv * logS.d[2], Sv * logS.d[3]};
            const double x = o.sign * (Sv - o.strike);
            const double w = x > 0.0 ? 1.0 : (x == 0.0 ? 0.5 : 0.0);
            const double pay = fmax(x, 0.0);
            const double inv = 1.0 / o.numeraire;
#pragma unroll
            for (int n = 0; n < NNS; ++n) {
                const double sel = (NNS == 1 || n == o.netting_set) ? inv : 0.0;        // (a select, not a branch: static register indices)
                acc[n] += pay * sel;
#pragma unroll
                for (int j = 0; j < P; ++j) dacc[n][j] += w * o.sign * dS[j] * sel;
                dacc[n][2] -= pay * sel * inv * o.dnum_drate;
            }
        }
    };
    for (int t = 0; t < k.n_initial_store; ++t) on_date(t);
    const uint64_t path = k.path_offset + (uint64_t)i;
    for (int step = 0; step < k.n_steps; ++step) {
        const mcx_step sp = ldk_struct(&k.steps[step]);
        const double dt = sp.dt;
        double z0, z1, u = 0.0;
        if (INJECT) {
            z0 = k.inject_z[((int64_t)step * 2 + 0) * k.ld + i];
            z1 = k.inject_z[((int64_t)step * 2 + 1) * k.ld + i];
            if (k.n_uniform) u = k.inject_u[(int64_t)step * k.ld + i];
        } else {
            double ua;
            draw_pair<true>(k.seed, path, (uint32_t)step, 0u, ua, z0, z1, tab);
            if (k.n_uniform) { double t0, t1; draw_pair(k.seed, path, (uint32_t)step, 1u, u, t0, t1); }
        }
        if constexpr (!QE) {
            // corr = chol([[1,rho],[rho,1]]): zc0 = z0, zc1 = rho z0 + sqrt(1-rho^2) z1      heston.py:109-121
            const Dual<PL> zc1 = rho * z0 + dsqrt(1.0 - rho * rho) * z1;
            const Dual<PV> sv = dsqrt(dclamp_min(v, 0.0));
            logS = logS + (rate - v * 0.5) * dt + sv * (sp.sqrt_dt * z0);
            const Dual<PV> v_n = v + kappa * (theta - v) * dt + sigma * sv * zc1 * sp.sqrt_dt;
            v = dclamp_min(v_n, 0.0);
        } else {
            // Andersen QE, heston.py:161-253 (same operation order as the oracle); the parameter-only factors come from QEConst
            const double eps = 1e-12;
            const QEConst* qc = a.qe + ldk(a.qe_idx + step);
            const Dual<PV> E = ldk_struct(&qc->E);
            const Dual<PV> m = theta + (v - theta) * E;
            const Dual<PV> s2 = v * ldk_struct(&qc->A1) + ldk_struct(&qc->A2);
            // From here to the next variance the step is a function of TWO numbers, vn = G(m, s2) (+ the draws): it runs in dual
            // numbers with two directions (d/dm, d/ds2) and the four parameter tangents follow by the chain rule,
            // d vn = G_m d m + G_s2 d s2 — the ~30 operations of G carry 2 tangent components instead of 4.
            // (the seven reciprocals in three groups from one v_rcp_f64 each, the two roots of b2 as one: as in the primal step,
            //  mcx_device.h; every denominator keeps the reference's eps)
            Dual<2> M, S2;
            M.v = m.v; M.d[0] = 1.0; M.d[1] = 0.0;
            S2.v = s2.v; S2.d[0] = 0.0; S2.d[1] = 1.0;
            const double omu = fmax(1.0 - u, eps);
            const Dual<2> d1 = M * M + eps, d5 = M + eps;
            const double d15 = d1.v * d5.v;
            const double ra = mcx_rcp(d15 * omu);
            const double r_omu = ra * d15, ra6 = ra * omu;
            const Dual<2> psi = ddiv_r(S2, d1, ra6 * d5.v);
            const Dual<2> d2 = psi + eps, d4 = psi + 1.0;
            const double rb = mcx_rcp(d2.v * d4.v);
            const Dual<2> invpsi = drcp_r(d2, rb * d4.v);
            const Dual<2> t = dclamp_min(invpsi * 2.0 - 1.0, 0.0);
            Dual<2> root = dsqrt(invpsi * 2.0 * t);
            if (invpsi.v < 0.0) root.v = __builtin_nan("");          // torch.sqrt(2 / psi) of a negative psi (mcx_device.h, QE step)
            const Dual<2> b2 = dclamp_min(invpsi * 2.0 - 1.0 + root, 0.0);
            const Dual<2> b = dsqrt(b2);
            const Dual<2> pp = dclamp(ddiv_r(psi - 1.0, d4, rb * d2.v), 0.0, 1.0 - 1e-6);
            const Dual<2> beta = ddiv_r(1.0 - pp, d5, ra6 * d1.v);
            const Dual<2> d3 = 1.0 + b2, d7 = beta + eps;
            const double rc = mcx_rcp(d3.v * d7.v);
            const Dual<2> aa = ddiv_r(M, d3, rc * d7.v);
            const Dual<2> bz = b + z1;
            const Dual<2> v1 = aa * bz * bz;
            const Dual<2> omp = dclamp_min(1.0 - pp, eps);
            const Dual<2> v_tail = ddiv_r(dlog(omp * r_omu), d7, rc * d3.v);
            const Dual<2> w_mass = ddegree(u - pp, fuzzy, 0.3);
            const Dual<2> v2 = w_mass * v_tail;
            const Dual<2> w = ddegree(psi - 1.5, fuzzy, 0.5);
            const Dual<2> vn2 = (1.0 - w) * v1 + w * v2;
            Dual<PV> vn;
            vn.v = vn2.v;
#pragma unroll
            for (int j = 0; j < PV; ++j) vn.d[j] = fma(vn2.d[0], m.d[j], vn2.d[1] * s2.d[j]);
            const Dual<PL> vL = dext<PV, PL>(v), vnL = dext<PV, PL>(vn);
            const Dual<PL> var_int = dclamp_min(ldk_struct(&qc->K3) * vL, 0.0);               // K4 = 0
            const Dual<PL> vol = dsqrt(dclamp_min(var_int, eps));
            logS = logS + ldk_struct(&qc->base) + ldk_struct(&qc->K1) * vL + ldk_struct(&qc->K2) * vnL + vol * z0;
            v = vn;
        }
        elapsed += dt;
        if (sp.store_idx >= 0) on_date(sp.store_idx);
    }
#pragma unroll
    for (int n = 0; n < NNS; ++n) {          // compile-time indices: the accumulators stay in registers
        if (n < a.n_ns) {
            a.cfs[(int64_t)n * a.ld_out + i] = acc[n];
#pragma unroll
            for (int j = 0; j < P; ++j) a.dcfs[((int64_t)n * P + j) * a.ld_out + i] = dacc[n][j];
        }
    }
}

// host-side dual numbers for the QE records (same formulas as the step they replace)
struct HD { double v, d[5]; };        // [sigma, kappa, theta, v0, rho]
static HD hd_c(double c) { HD r; r.v = c; for (double& x : r.d) x = 0.0; return r; }
static HD hd_seed(double c, int k) { HD r = hd_c(c); r.d[k] = 1.0; return r; }
static HD operator+(const HD& a, const HD& b) { HD r; r.v = a.v + b.v; for (int j = 0; j < 5; ++j) r.d[j] = a.d[j] + b.d[j]; return r; }
static HD operator-(const HD& a, const HD& b) { HD r; r.v = a.v - b.v; for (int j = 0; j < 5; ++j) r.d[j] = a.d[j] - b.d[j]; return r; }
static HD operator*(const HD& a, const HD& b) { HD r; r.v = a.v * b.v; for (int j = 0; j < 5; ++j) r.d[j] = a.d[j] * b.v + a.v * b.d[j]; return r; }
static HD operator/(const HD& a, const HD& b) { HD r; const double ib = 1.0 / b.v; r.v = a.v * ib; for (int j = 0; j < 5; ++j) r.d[j] = (a.d[j] - r.v * b.d[j]) * ib; return r; }
static HD operator*(const HD& a, double c) { return a * hd_c(c); }
static HD hd_exp(const HD& a) { HD r; r.v = exp(a.v); for (int j = 0; j < 5; ++j) r.d[j] = r.v * a.d[j]; return r; }
template <int PD> static void hd_store(const HD& a, Dual<PD>* out) { out->v = a.v; for (int j = 0; j < PD; ++j) out->d[j] = a.d[j]; }

static QEConst qe_const(const double* p, double dt)
{
    // params [spot, sigma_v, rate, rho, kappa, theta, v0]
    const HD sigma = hd_seed(p[1], 0), kappa = hd_seed(p[4], 1), theta = hd_seed(p[5], 2), rho = hd_seed(p[3], 4), rate = hd_c(p[2]);
    const HD E = hd_exp(kappa * (-dt));
    const HD om = hd_c(1.0) - E;
    const HD ros = rho / sigma;
    QEConst c;
    memset(&c, 0, sizeof(c));
    hd_store(E, &c.E);
    hd_store(sigma * sigma * E * om / kappa, &c.A1);
    hd_store(theta * sigma * sigma * om * om / (kappa * 2.0), &c.A2);
    hd_store(rate * dt + (rho * kappa * theta / sigma) * (-dt), &c.base);
    hd_store((kappa * ros - hd_c(0.5)) * dt - ros, &c.K1);
    hd_store(ros, &c.K2);
    hd_store((hd_c(1.0) - rho * rho) * dt, &c.K3);
    return c;
}

}  // namespace

extern "C" int mcx_tangent_european(mcx_handle* h, const mcx_sim* sim, const mcx_tangent_option* h_opts, int32_t n_opts,
                                    int32_t n_netting_sets, uint64_t seed, uint64_t path_offset, int64_t n_paths, int64_t ld,
                                    double* d_cfs, double* d_dcfs, int64_t ld_out, const double* d_inject_z,
                                    const double* d_inject_u, void* stream)
{
    if (!h || !sim || !h_opts || !d_cfs || !d_dcfs) return -1;
    const mcx_sim_desc& d = sim->desc;
    if (d.n_slots != 1 || (d.slots[0].kind != MCX_MODEL_BS && d.slots[0].kind != MCX_MODEL_HESTON))
        MCX_FAIL(h, -2, "mcx_tangent_european: single Black-Scholes or Heston model required");
    if (n_netting_sets < 1 || n_netting_sets > MCX_FUSED_MAX_NS) MCX_FAIL(h, -2, "mcx_tangent_european: 1..%d netting sets", MCX_FUSED_MAX_NS);
    if (n_opts < 1 || n_opts > 4096) MCX_FAIL(h, -2, "mcx_tangent_european: option count out of range");
    for (int q = 0; q < n_opts; ++q)
        if (h_opts[q].t_idx < 0 || h_opts[q].t_idx >= d.n_dates || h_opts[q].netting_set < 0 || h_opts[q].netting_set >= n_netting_sets)
            MCX_FAIL(h, -2, "mcx_tangent_european: option %d out of range", q);
    if (n_paths <= 0) return 0;
    if (ld_out < n_paths || (d_inject_z && ld < n_paths)) MCX_FAIL(h, -2, "mcx_tangent_european: leading dimension < n_paths");
    if (d.n_uniform && d_inject_z && !d_inject_u) MCX_FAIL(h, -3, "mcx_tangent_european: inject_u required with inject_z under QE");
    hipStream_t s = (hipStream_t)stream;
    const mcx_tangent_option* d_opts = (const mcx_tangent_option*)mcx_stage_small(h, h_opts, sizeof(mcx_tangent_option) * (size_t)n_opts, s);
    if (!d_opts) return -100;
    KTArgs a;
    memset(&a, 0, sizeof(a));
    mcx_fill_k1_args(sim, seed, path_offset, n_paths, ld > 0 ? ld : n_paths, nullptr, d_inject_z, d_inject_u, &a.k1);
    a.opts = d_opts; a.cfs = d_cfs; a.dcfs = d_dcfs; a.ld_out = ld_out; a.n_opts = n_opts; a.n_ns = n_netting_sets;
    if (d.slots[0].kind == MCX_MODEL_HESTON && d.scheme == MCX_SCHEME_QE) {
        // one record of parameter-only dual factors per distinct dt of the sub-step table
        std::vector<double> dts;
        std::vector<int32_t> idx(sim->h_steps.size());
        for (size_t k = 0; k < sim->h_steps.size(); ++k) {
            size_t q = 0;
            while (q < dts.size() && dts[q] != sim->h_steps[k].dt) ++q;
            if (q == dts.size()) {
                if (dts.size() >= 4096) MCX_FAIL(h, -2, "mcx_tangent_european: more than 4096 distinct step sizes");
                dts.push_back(sim->h_steps[k].dt);
            }
            idx[k] = (int32_t)q;
        }
        std::vector<QEConst> recs(dts.size());
        for (size_t q = 0; q < dts.size(); ++q) recs[q] = qe_const(d.slots[0].p, dts[q]);
        a.qe = (const QEConst*)mcx_stage_small(h, recs.data(), sizeof(QEConst) * recs.size(), s);
        a.qe_idx = (const int32_t*)mcx_stage_small(h, idx.data(), sizeof(int32_t) * idx.size(), s);
        if (!a.qe || !a.qe_idx) return -100;
    }
    const int grid = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    const bool inj = d_inject_z != nullptr;
    const bool one = n_netting_sets == 1;
#define MCX_KT(kern_one, kern_all) do { if (one) hipLaunchKernelGGL(kern_one, dim3(grid), dim3(MCX_BLOCK), 0, s, a); \
                                        else hipLaunchKernelGGL(kern_all, dim3(grid), dim3(MCX_BLOCK), 0, s, a); } while (0)
    if (d.slots[0].kind == MCX_MODEL_BS) {
        if (inj) MCX_KT((kt_bs<true, 1>), (kt_bs<true, MCX_FUSED_MAX_NS>));
        else MCX_KT((kt_bs<false, 1>), (kt_bs<false, MCX_FUSED_MAX_NS>));
    } else {
        const bool qe = d.scheme == MCX_SCHEME_QE;
        if (qe && inj) MCX_KT((kt_heston<true, true, 1>), (kt_heston<true, true, MCX_FUSED_MAX_NS>));
        else if (qe) MCX_KT((kt_heston<false, true, 1>), (kt_heston<false, true, MCX_FUSED_MAX_NS>));
        else if (inj) MCX_KT((kt_heston<true, false, 1>), (kt_heston<true, false, MCX_FUSED_MAX_NS>));
        else MCX_KT((kt_heston<false, false, 1>), (kt_heston<false, false, MCX_FUSED_MAX_NS>));
    }
#undef MCX_KT
    MCX_HIP(h, hipGetLastError());
    return 0;          // stream-ordered (descriptors travel through the handle's staging ring)
}
