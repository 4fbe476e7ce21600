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

struct KTArgs {
    K1Args k1;
    const mcx_tangent_option* __restrict__ opts;
    double* __restrict__ cfs;
    double* __restrict__ dcfs;
    int64_t ld_out;
    int32_t n_opts, n_ns;
};

// Black-Scholes: params [spot, sigma, rate], state S
template <bool INJECT>
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
    double acc[MCX_FUSED_MAX_NS] = {0, 0, 0, 0}, dacc[MCX_FUSED_MAX_NS][P] = {};
    auto on_date = [&](int t) {
        for (int q = 0; q < a.n_opts; ++q) {
            const mcx_tangent_option o = ldk_struct(&a.opts[q]);
            if (o.t_idx != t) continue;
            // payoff = torch.maximum(sign*(S-K), 0) / numeraire   (european_option.py:45-68)
            const double x = o.sign * (S.v - o.strike);
            const double w = x > 0.0 ? 1.0 : (x == 0.0 ? 0.5 : 0.0);
            const double pay = fmax(x, 0.0);
            const double inv = 1.0 / o.numeraire;
            for (int n = 0; n < MCX_FUSED_MAX_NS; ++n) {
                if (n != o.netting_set) continue;
                acc[n] += pay * inv;
                for (int j = 0; j < P; ++j) dacc[n][j] += w * o.sign * S.d[j] * inv;
                dacc[n][2] -= pay * inv * inv * o.dnum_drate;
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
    for (int n = 0; n < a.n_ns; ++n) {
        a.cfs[(int64_t)n * a.ld_out + i] = acc[n];
        for (int j = 0; j < P; ++j) a.dcfs[((int64_t)n * P + j) * a.ld_out + i] = dacc[n][j];
    }
}

// Heston: params [spot, sigma_v, rate, rho, kappa, theta, v0], state (log S, v)
template <bool INJECT>
__global__ __launch_bounds__(MCX_BLOCK) void kt_heston(const KTArgs a)
{
    constexpr int P = 7;
    const K1Args& k = a.k1;
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    __shared__ double bm_lds[INJECT ? 2 : MCX_BM_LDS_DOUBLES];        // table-driven Box-Muller (mcx_math.h)
    const double* tab = nullptr;
    if (!INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    if (i >= k.n) return;
    const double* p = k.slots[0].p;
    const Dual<P> spot = dseed<P>(p[0], 0), sigma = dseed<P>(p[1], 1), rate = dseed<P>(p[2], 2), rho = dseed<P>(p[3], 3),
                  kappa = dseed<P>(p[4], 4), theta = dseed<P>(p[5], 5), v0 = dseed<P>(p[6], 6);
    Dual<P> logS = dlog(spot), v = v0;
    const bool fuzzy = ((k.flags | k.slots[0].flags) & MCX_FLAG_SMOOTHING) != 0;
    double acc[MCX_FUSED_MAX_NS] = {0, 0, 0, 0}, dacc[MCX_FUSED_MAX_NS][P] = {};
    auto on_date = [&](int t) {
        for (int q = 0; q < a.n_opts; ++q) {
            const mcx_tangent_option o = ldk_struct(&a.opts[q]);
            if (o.t_idx != t) continue;
            const Dual<P> S = dexp(logS);                                          // heston.py:258-260
            const double x = o.sign * (S.v - o.strike);
            const double w = x > 0.0 ? 1.0 : (x == 0.0 ? 0.5 : 0.0);
            const double pay = fmax(x, 0.0);
            const double inv = 1.0 / o.numeraire;
            for (int n = 0; n < MCX_FUSED_MAX_NS; ++n) {
                if (n != o.netting_set) continue;
                acc[n] += pay * inv;
                for (int j = 0; j < P; ++j) dacc[n][j] += w * o.sign * S.d[j] * inv;
                dacc[n][2] -= pay * inv * inv * o.dnum_drate;
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
        if (k.scheme == MCX_SCHEME_EULER) {
            // corr = chol([[1,rho],[rho,1]]): zc0 = z0, zc1 = rho z0 + sqrt(1-rho^2) z1      heston.py:109-121
            const Dual<P> zc1 = rho * z0 + dsqrt(1.0 - rho * rho) * z1;
            const Dual<P> sv = dsqrt(dclamp_min(v, 0.0));
            const Dual<P> logS_n = logS + (rate - v * 0.5) * dt + sv * (sp.sqrt_dt * z0);
            const Dual<P> v_n = v + kappa * (theta - v) * dt + sigma * sv * zc1 * sp.sqrt_dt;
            logS = logS_n;
            v = dclamp_min(v_n, 0.0);
        } else {
            // Andersen QE, heston.py:161-253 (same operation order as the oracle)
            const double eps = 1e-12;
            const Dual<P> E = dexp(kappa * (-dt));
            const Dual<P> m = theta + (v - theta) * E;
            const Dual<P> om = 1.0 - E;
            const Dual<P> s2 = v * sigma * sigma * E * om / kappa + theta * sigma * sigma * om * om / (kappa * 2.0);
            const Dual<P> psi = s2 / (m * m + eps);
            const Dual<P> invpsi = 1.0 / (psi + eps);
            const Dual<P> t = dclamp_min(invpsi * 2.0 - 1.0, 0.0);
            const Dual<P> b2 = dclamp_min(invpsi * 2.0 - 1.0 + dsqrt(invpsi * 2.0) * dsqrt(t), 0.0);
            const Dual<P> b = dsqrt(b2);
            const Dual<P> aa = m / (1.0 + b2);
            const Dual<P> bz = b + z1;
            const Dual<P> v1 = aa * bz * bz;
            const Dual<P> pp = dclamp((psi - 1.0) / (psi + 1.0), 0.0, 1.0 - 1e-6);
            const Dual<P> beta = (1.0 - pp) / (m + eps);
            const double omu = fmax(1.0 - u, eps);
            const Dual<P> omp = dclamp_min(1.0 - pp, eps);
            const Dual<P> v_tail = dlog(omp * (1.0 / omu)) / (beta + eps);
            const Dual<P> w_mass = ddegree(u - pp, fuzzy, 0.3);
            const Dual<P> v2 = w_mass * v_tail;
            const Dual<P> w = ddegree(psi - 1.5, fuzzy, 0.5);
            const Dual<P> vn = (1.0 - w) * v1 + w * v2;
            const Dual<P> ros = rho / sigma;
            const Dual<P> K0 = (rho * kappa * theta / sigma) * (-dt);
            const Dual<P> K1 = (kappa * ros - 0.5) * dt - ros;
            const Dual<P> K2 = ros;                                         // gamma2 = 0
            const Dual<P> K3 = (1.0 - rho * rho) * dt;
            const Dual<P> var_int = dclamp_min(K3 * v, 0.0);                // K4 = 0
            const Dual<P> vol = dsqrt(dclamp_min(var_int, eps));
            logS = logS + rate * dt + K0 + K1 * v + K2 * vn + vol * z0;
            v = vn;
        }
        if (sp.store_idx >= 0) on_date(sp.store_idx);
    }
    for (int n = 0; n < a.n_ns; ++n) {
        a.cfs[(int64_t)n * a.ld_out + i] = acc[n];
        for (int j = 0; j < P; ++j) a.dcfs[((int64_t)n * P + j) * a.ld_out + i] = dacc[n][j];
    }
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
    mcx_tangent_option* d_opts = nullptr;
    MCX_HIP(h, hipMalloc(&d_opts, sizeof(mcx_tangent_option) * (size_t)n_opts));
    MCX_HIP(h, hipMemcpyAsync(d_opts, h_opts, sizeof(mcx_tangent_option) * (size_t)n_opts, hipMemcpyHostToDevice, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    KTArgs a;
    memset(&a, 0, sizeof(a));
    mcx_fill_k1_args(sim, seed, path_offset, n_paths, ld > 0 ? ld : n_paths, nullptr, d_inject_z, d_inject_u, &a.k1);
    a.opts = d_opts; a.cfs = d_cfs; a.dcfs = d_dcfs; a.ld_out = ld_out; a.n_opts = n_opts; a.n_ns = n_netting_sets;
    const int grid = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    const bool inj = d_inject_z != nullptr;
    if (d.slots[0].kind == MCX_MODEL_BS) {
        if (inj) hipLaunchKernelGGL(kt_bs<true>, dim3(grid), dim3(MCX_BLOCK), 0, s, a);
        else hipLaunchKernelGGL(kt_bs<false>, dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    } else {
        if (inj) hipLaunchKernelGGL(kt_heston<true>, dim3(grid), dim3(MCX_BLOCK), 0, s, a);
        else hipLaunchKernelGGL(kt_heston<false>, dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    }
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    MCX_HIP(h, hipFree(d_opts));
    return 0;
}
