// k1_paths.hip — K1: fused Philox4x32-10 + Box-Muller + Cholesky + SDE step kernel for gfx950 (CDNA4, wave64).
//
// Replaces the reference's Python time loop (engine/engine.py:35-123): `torch.randn(N, sim_dim) @ chol.T`
// (models/model.py:46-48) followed by ~10-60 ATen launches per sub-step (models/<model>.py simulate_time_step_*).
//
// Mapping to the hardware
//   * one lane owns one path for ALL sub-steps: the state lives in VGPRs, nothing but the stored timeline dates ever
//     touches HBM;
//   * layout [date][state][path]: the 64 lanes of a wavefront store 64 consecutive doubles = 512 contiguous bytes;
//   * every per-sub-step table (dt, sqrt(dt), Cholesky factor, psi(t), exp(-a dt), QE constants) is indexed by the
//     wave-uniform loop counter, so it is fetched with scalar loads (s_load_dwordx*) through the scalar cache into
//     SGPRs — one fetch per wave instead of 64 LDS reads; the model parameters travel in the kernel argument segment
//     (also SGPRs).  (The north-star sketch stages these in LDS; on CDNA4 a wave-uniform table is cheaper in SGPRs.)
//   * the model-kind switch is wave-uniform (s_cbranch), there is no lane divergence except the QE branch blend, which
//     is computed branch-free exactly like the reference's fuzzy blend.
//   * counter-based RNG: counter = (global path id, sub-step, draw), key = seed  => results are independent of the
//     launch geometry and of how paths are sharded over GPUs.
#include "mcx_device.h"

namespace {


#ifndef MCX_K1_PPL
#define MCX_K1_PPL 1      // paths per lane: 2 interleaves two independent RNG/SDE dependency chains per lane (ILP)
#endif
template <int NSLOT, int NZ, bool INJECT, int SIG>
__global__ __launch_bounds__(MCX_BLOCK) void k1_paths(const K1Args a)
{
    constexpr int PPL = INJECT ? 1 : MCX_K1_PPL;
    __shared__ double bm_lds[INJECT ? 2 : MCX_BM_LDS_DOUBLES];
    const double* tab = nullptr;
    if (!INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    const int64_t tid = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    const int64_t half = (int64_t)gridDim.x * MCX_BLOCK;          // lane handles paths tid + q*half
    if (tid >= a.n) return;
    double reg[PPL][2 * NSLOT];
    int64_t idx[PPL];
    bool live[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        idx[q] = tid + q * half;
        live[q] = idx[q] < a.n;
        if (!live[q]) idx[q] = a.n - 1;
        sim_init_state<NSLOT, SIG>(a, reg[q], idx[q]);
    }
    for (int t = 0; t < a.n_initial_store; ++t)
#pragma unroll
        for (int q = 0; q < PPL; ++q) if (live[q]) sim_store_state<NSLOT, SIG>(a, t, idx[q], reg[q]);
    const mcx_bm_coef bc = mcx_bm_coef_load();
    const mcx_bm_vconst vc = mcx_bm_vconst_make(bc);          // Box-Muller constants kept in registers across the step loop
    for (int k = 0; k < a.n_steps; ++k) {
        int st;
        if constexpr (INJECT || SIG == SIG_GENERIC) {      // (run-time model dispatch: every aux entry of every slot would be live)
#pragma unroll
            for (int q = 0; q < PPL; ++q)
                sim_substep<NSLOT, NZ, INJECT, SIG>(a, k, a.path_offset + (uint64_t)idx[q], idx[q], reg[q], tab, a.seed, bc, &vc);
            st = ldk(&a.steps[k].store_idx);
        } else {
            // as in the one-launch kernel (kf_lean.hip): the scalar loads of the sub-step go out ahead of its draws, the draws of the
            // lane's paths are staged (table reads first, the arithmetic that needs no table value covers their latency) with the
            // unguarded root, one rare branch repeats the 2^-32 draws that may round to u = 1
            const StepData<NSLOT, NZ> sdat = sim_step_load<NSLOT, NZ, SIG>(a, k);
            __builtin_amdgcn_sched_barrier(0);
            uint64_t path[PPL];
            uint32_t st_q[PPL];
            double zz[PPL][NZ], uu[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) { path[q] = a.path_offset + (uint64_t)idx[q]; st_q[q] = (uint32_t)k; }
            const bool rare = sim_draw_n<PPL, NZ, SIG, 7>(a, st_q, path, zz, uu, tab, a.seed, bc, vc);
            if (__builtin_expect(__any(rare), 0)) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) sim_draw<NZ, false, SIG, 7, true>(a, k, path[q], idx[q], zz[q], uu[q], tab, a.seed, bc, &vc);
            }
            st = -1;
#pragma unroll
            for (int q = 0; q < PPL; ++q) st = sim_apply_loaded<NSLOT, NZ, SIG>(a, sdat, reg[q], zz[q], uu[q]);
        }
        if (st >= 0)
#pragma unroll
            for (int q = 0; q < PPL; ++q) if (live[q]) sim_store_state<NSLOT, SIG>(a, st, idx[q], reg[q], a.aux + (int64_t)k * NSLOT * MCX_AUX);
    }
}

template <int NSLOT, int NZ, int SIG>
int launch_k1(const K1Args& a, bool inject, hipStream_t s)
{
    const int ppl = inject ? 1 : MCX_K1_PPL;
    const int grid = (int)((a.n + (int64_t)MCX_BLOCK * ppl - 1) / ((int64_t)MCX_BLOCK * ppl));
    if (inject) hipLaunchKernelGGL((k1_paths<NSLOT, NZ, true, SIG>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k1_paths<NSLOT, NZ, false, SIG>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    return 0;
}

// the draw (path0 + i, step, draw) exactly as sim_substep consumes it: Philox words, the two uniforms, the Box-Muller pair
__global__ __launch_bounds__(MCX_BLOCK) void k1_rng_draws(uint64_t seed, uint64_t path0, int64_t n, uint32_t step, uint32_t draw,
                                                          uint32_t* __restrict__ words, double* __restrict__ u, double* __restrict__ z)
{
    __shared__ double bm_lds[MCX_BM_LDS_DOUBLES];
    mcx_bm_load(bm_lds);
    const mcx_bm_coef bc = mcx_bm_coef_load();
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const uint64_t path = path0 + (uint64_t)i;
        if (words) {
            uint32_t w0, w1, w2, w3;
            philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), step, draw, (uint32_t)seed, (uint32_t)(seed >> 32), w0, w1, w2, w3);
            words[i] = w0; words[n + i] = w1; words[2 * n + i] = w2; words[3 * n + i] = w3;
            if (u) { u[i] = u53(w0, w1); u[n + i] = u53(w2, w3); }
        }
        if (z) {
            double ua, z0, z1;
            draw_pair<true>(seed, path, step, draw, ua, z0, z1, bm_lds, bc);
            z[i] = z0; z[n + i] = z1;
        }
    }
}

// words [4][n] of Philox blocks -> uniforms [2][n] and Box-Muller pairs [2][n], with the BMB-bit tables
template <int BMB>
__global__ __launch_bounds__(MCX_BLOCK) void k1_box_muller(const uint32_t* __restrict__ words, int64_t n, double* __restrict__ u,
                                                          double* __restrict__ z)
{
    __shared__ double bm_lds[MCX_BM_LDS_DOUBLES_B(BMB)];
    mcx_bm_load<BMB>(bm_lds);
    const mcx_bm_coef bc = mcx_bm_coef_load();
    const mcx_bm_vconst vc = mcx_bm_vconst_make<BMB>(bc);
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const uint32_t w0 = words[i], w1 = words[n + i], w2 = words[2 * n + i], w3 = words[3 * n + i];
        double ua, z0, z1;
        pair_from_words<true, BMB>(w0, w1, w2, w3, ua, z0, z1, bm_lds, bc, &vc);
        if (u) { u[i] = ua; u[n + i] = u53(w2, w3); }
        z[i] = z0; z[n + i] = z1;
    }
}

}  // namespace

extern "C" int mcx_box_muller(mcx_handle* h, const uint32_t* d_words, int64_t n, int32_t table_bits, double* d_u, double* d_z, void* stream)
{
    if (!h || !d_words || !d_z) return -1;
    if (table_bits != 7 && table_bits != 10) MCX_FAIL(h, -2, "mcx_box_muller: table_bits must be 7 or 10");
    if (n <= 0) return 0;
    const int grid = mcx_grid_for(n, MCX_BLOCK, 2048);
    if (table_bits == 7) hipLaunchKernelGGL(k1_box_muller<7>, dim3(grid), dim3(MCX_BLOCK), 0, (hipStream_t)stream, d_words, n, d_u, d_z);
    else hipLaunchKernelGGL(k1_box_muller<10>, dim3(grid), dim3(MCX_BLOCK), 0, (hipStream_t)stream, d_words, n, d_u, d_z);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_rng_draws(mcx_handle* h, uint64_t seed, uint64_t path0, int64_t n, uint32_t step, uint32_t draw,
                             uint32_t* d_words, double* d_u, double* d_z, void* stream)
{
    if (!h) return -1;
    if (n <= 0) return 0;
    if (d_u && !d_words) MCX_FAIL(h, -2, "mcx_rng_draws: d_u needs d_words");
    const int grid = mcx_grid_for(n, MCX_BLOCK, 2048);
    hipLaunchKernelGGL(k1_rng_draws, dim3(grid), dim3(MCX_BLOCK), 0, (hipStream_t)stream, seed, path0, n, step, draw, d_words, d_u, d_z);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_sim_create(mcx_handle* h, const mcx_sim_desc* d, mcx_sim** out)
{
    if (!h || !d || !out) return -1;
    if (d->n_slots < 1 || d->n_slots > MCX_MAX_SLOTS || d->n_z > MCX_MAX_Z || d->n_state > MCX_MAX_STATE)
        MCX_FAIL(h, -2, "mcx_sim_create: dimensions out of range (slots %d, z %d, state %d)", d->n_slots, d->n_z, d->n_state);
    if (!(d->n_slots == 1 || d->n_z == d->n_slots))
        MCX_FAIL(h, -3, "mcx_sim_create: a multi-model configuration needs simulation_dim == 1 per sub-model");
    int so = 0;
    for (int s = 0; s < d->n_slots; ++s) {
        const int sd = mcx_kind_state_dim(d->slots[s].kind);
        if (d->slots[s].kind == MCX_MODEL_S2F && d->n_slots != 1) MCX_FAIL(h, -3, "mcx_sim_create: the Schwartz two-factor model runs alone");
        if (d->slots[s].state_off != so) MCX_FAIL(h, -4, "mcx_sim_create: state offsets must be packed");
        so += sd;
        // the step maps that exist per model (the reference's simulate_time_step_* overrides; model.py:102-141 leaves the others
        // unimplemented): the kernels read scheme-specific derived step constants, an unsupported pair would run on zeros
        const int kind = d->slots[s].kind, sc = d->scheme;
        bool ok;
        switch (kind) {
        case MCX_MODEL_HESTON: ok = sc == MCX_SCHEME_EULER || sc == MCX_SCHEME_QE; break;
        case MCX_MODEL_CIRPP: ok = sc == MCX_SCHEME_EULER; break;
        case MCX_MODEL_CIRPP_DET: ok = sc >= MCX_SCHEME_EULER && sc <= MCX_SCHEME_QE; break;
        case MCX_MODEL_BS: case MCX_MODEL_VASICEK: case MCX_MODEL_HW: case MCX_MODEL_S2F:
            ok = sc == MCX_SCHEME_EULER || sc == MCX_SCHEME_ANALYTICAL; break;
        default: ok = false; break;
        }
        if (!ok) MCX_FAIL(h, -6, "mcx_sim_create: slot %d (model kind %d) has no step map for scheme %d", s, kind, sc);
    }
    if (so != d->n_state) MCX_FAIL(h, -4, "mcx_sim_create: n_state does not match the slots");
    for (int k = 0; k < d->n_steps; ++k) {
        if (d->steps[k].store_idx >= d->n_dates || d->steps[k].chol_idx < 0 || d->steps[k].chol_idx >= d->n_chol)
            MCX_FAIL(h, -5, "mcx_sim_create: step %d references date/cholesky out of range", k);
        // EULER / QE correlate every sub-step with the factor of the one correlation matrix (model.py:66-73): the kernels read
        // factor 0 without waiting for the step record
        if ((d->scheme == MCX_SCHEME_EULER || d->scheme == MCX_SCHEME_QE) && d->steps[k].chol_idx != 0)
            MCX_FAIL(h, -5, "mcx_sim_create: step %d: the EULER / QE schemes use Cholesky factor 0 for every sub-step", k);
    }
    MCX_HIP(h, hipSetDevice(h->device));
    mcx_sim* sim = new mcx_sim();
    sim->desc = *d;
    sim->d_steps = nullptr; sim->d_chol = nullptr; sim->d_aux = nullptr;
    // init_state travels in the kernel-argument segment; keep a private host copy
    sim->n_state_total = d->n_state;
    double* init = new double[MCX_MAX_STATE]();
    for (int c = 0; c < d->n_state; ++c) init[c] = d->init_state[c];
    sim->desc.init_state = init;
    sim->h_steps.assign(d->steps, d->steps + (d->n_steps > 0 ? d->n_steps : 0));
    sim->desc.steps = nullptr; sim->desc.chol = nullptr; sim->desc.aux = nullptr;
    const size_t nb_steps = sizeof(mcx_step) * (size_t)(d->n_steps > 0 ? d->n_steps : 1);
    const size_t nb_chol = sizeof(double) * (size_t)(d->n_chol > 0 ? d->n_chol : 1) * d->n_z * d->n_z;
    const size_t nb_aux = sizeof(double) * (size_t)(d->n_steps > 0 ? d->n_steps : 1) * d->n_slots * MCX_AUX;
    hipError_t e = hipMalloc(&sim->d_steps, nb_steps);
    if (e == hipSuccess) e = hipMalloc(&sim->d_chol, nb_chol);
    if (e == hipSuccess) e = hipMalloc(&sim->d_aux, nb_aux);
    if (e == hipSuccess && d->n_steps > 0) e = hipMemcpy(sim->d_steps, d->steps, sizeof(mcx_step) * d->n_steps, hipMemcpyHostToDevice);
    if (e == hipSuccess && d->n_steps > 0) {
        std::vector<double> aux(d->aux, d->aux + (size_t)d->n_steps * d->n_slots * MCX_AUX);
        if (d->scheme == MCX_SCHEME_EULER) {
            for (int k = 0; k < d->n_steps; ++k)
                for (int q = 0; q < d->n_slots; ++q) {
                    double* a = aux.data() + ((size_t)k * d->n_slots + q) * MCX_AUX;
                    const double* p = d->slots[q].p;
                    const double dt = d->steps[k].dt, sq = d->steps[k].sqrt_dt;
                    switch (d->slots[q].kind) {
                    case MCX_MODEL_BS: a[MCX_AUX_C0] = p[2] * dt; a[MCX_AUX_C2] = p[1] * sq; break;
                    case MCX_MODEL_VASICEK: a[MCX_AUX_C0] = p[3] * p[2] * dt; a[MCX_AUX_C1] = -(p[3] * dt); a[MCX_AUX_C2] = p[1] * sq; break;
                    case MCX_MODEL_CIRPP: a[MCX_AUX_C0] = p[0] * p[1] * dt; a[MCX_AUX_C1] = -(p[0] * dt); a[MCX_AUX_C2] = p[2] * sq; break;
                    default: break;
                    }
                }
        }
        e = hipMemcpy(sim->d_aux, aux.data(), sizeof(double) * aux.size(), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && d->n_chol > 0)
        e = hipMemcpy(sim->d_chol, d->chol, sizeof(double) * (size_t)d->n_chol * d->n_z * d->n_z, hipMemcpyHostToDevice);
    if (e != hipSuccess) {                      // nothing of a failed create survives
        h->err = std::string("mcx_sim_create: ") + hipGetErrorString(e);
        mcx_sim_destroy(sim);
        return -100;
    }
    *out = sim;
    return 0;
}

extern "C" void mcx_sim_destroy(mcx_sim* sim)
{
    if (!sim) return;
    hipFree(sim->d_steps); hipFree(sim->d_chol); hipFree(sim->d_aux);
    delete[] sim->desc.init_state;
    delete sim;
}

static int generate_paths_impl(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths, int64_t ld,
                               const double* d_init_state, double* d_paths, const double* d_inject_z, const double* d_inject_u, void* stream);

extern "C" int mcx_generate_paths(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                                  int64_t ld, double* d_paths, const double* d_inject_z, const double* d_inject_u, void* stream)
{
    return generate_paths_impl(h, sim, seed, path_offset, n_paths, ld, nullptr, d_paths, d_inject_z, d_inject_u, stream);
}

extern "C" int mcx_generate_paths_from_state(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                                             int64_t ld, const double* d_init_state, double* d_paths,
                                             const double* d_inject_z, const double* d_inject_u, void* stream)
{
    if (!d_init_state) return -1;
    return generate_paths_impl(h, sim, seed, path_offset, n_paths, ld, d_init_state, d_paths, d_inject_z, d_inject_u, stream);
}

static int generate_paths_impl(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths, int64_t ld,
                               const double* d_init_state, double* d_paths, const double* d_inject_z, const double* d_inject_u, void* stream)
{
    if (!h || !sim || !d_paths) return -1;
    if (n_paths <= 0) return 0;
    if (ld < n_paths) MCX_FAIL(h, -2, "mcx_generate_paths: ld < n_paths");
    const mcx_sim_desc& d = sim->desc;
    if (d.n_uniform && d_inject_z && !d_inject_u) MCX_FAIL(h, -3, "mcx_generate_paths: inject_u required with inject_z under QE");
    K1Args a;
    mcx_fill_k1_args(sim, seed, path_offset, n_paths, ld, d_paths, d_inject_z, d_inject_u, &a);
    a.init_paths = d_init_state;
    hipStream_t s = (hipStream_t)stream;
    const bool inj = d_inject_z != nullptr;
    switch (mcx_sim_signature(d)) {       // specialised (compile-time model kinds) instantiations of the hot configurations
    case SIG_VAS_CIR_E: launch_k1<2, 2, SIG_VAS_CIR_E>(a, inj, s); break;
    case SIG_BS_A: launch_k1<1, 1, SIG_BS_A>(a, inj, s); break;
    case SIG_BS_E: launch_k1<1, 1, SIG_BS_E>(a, inj, s); break;
    case SIG_HESTON_QE: launch_k1<1, 2, SIG_HESTON_QE>(a, inj, s); break;
    case SIG_HESTON_E: launch_k1<1, 2, SIG_HESTON_E>(a, inj, s); break;
    case SIG_VAS_E: launch_k1<1, 1, SIG_VAS_E>(a, inj, s); break;
    case SIG_VAS_A: launch_k1<1, 1, SIG_VAS_A>(a, inj, s); break;
    case SIG_BS_VAS_CIRDET_E: launch_k1<3, 3, SIG_BS_VAS_CIRDET_E>(a, inj, s); break;
    default:
        switch (d.n_slots * 16 + d.n_z) {
        case 1 * 16 + 1: launch_k1<1, 1, SIG_GENERIC>(a, inj, s); break;
        case 1 * 16 + 2: launch_k1<1, 2, SIG_GENERIC>(a, inj, s); break;
        case 2 * 16 + 2: launch_k1<2, 2, SIG_GENERIC>(a, inj, s); break;
        case 3 * 16 + 3: launch_k1<3, 3, SIG_GENERIC>(a, inj, s); break;
        case 4 * 16 + 4: launch_k1<4, 4, SIG_GENERIC>(a, inj, s); break;
        case 5 * 16 + 5: launch_k1<5, 5, SIG_GENERIC>(a, inj, s); break;
        case 6 * 16 + 6: launch_k1<6, 6, SIG_GENERIC>(a, inj, s); break;
        case 7 * 16 + 7: launch_k1<7, 7, SIG_GENERIC>(a, inj, s); break;
        case 8 * 16 + 8: launch_k1<8, 8, SIG_GENERIC>(a, inj, s); break;
        default: MCX_FAIL(h, -4, "mcx_generate_paths: unsupported (slots=%d, z=%d)", d.n_slots, d.n_z);
        }
    }
    MCX_HIP(h, hipGetLastError());
    return 0;
}
