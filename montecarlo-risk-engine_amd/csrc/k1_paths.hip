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
#include "mcx_internal.h"

namespace {

struct K1Args {
    mcx_slot slots[MCX_MAX_SLOTS];
    double init_state[MCX_MAX_STATE];
    const mcx_step* __restrict__ steps;
    const double* __restrict__ chol;
    const double* __restrict__ aux;
    double* __restrict__ paths;
    const double* __restrict__ inject_z;
    const double* __restrict__ inject_u;
    int64_t n, ld;
    uint64_t seed, path_offset;
    int32_t scheme, n_steps, n_state, n_initial_store, flags, n_uniform;
};

// ---- Philox4x32-10 ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;     // v_mad_u64_u32
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

__device__ __forceinline__ double u53(uint32_t lo, uint32_t hi)
{
    const uint64_t x = ((uint64_t)hi << 32) | lo;
    return ((double)(x >> 11) + 0.5) * 0x1.0p-53;
}

// one draw = two uniforms in (0,1) and their Box-Muller pair (include/mcx.h "RNG contract")
__device__ __forceinline__ void draw_pair(uint64_t seed, uint64_t path, uint32_t step, uint32_t draw, double& ua, double& z0, double& z1)
{
    uint32_t w0, w1, w2, w3;
    philox4x32_10((uint32_t)path, (uint32_t)(path >> 32), step, draw, (uint32_t)seed, (uint32_t)(seed >> 32), w0, w1, w2, w3);
    ua = u53(w0, w1);
    const double ub = u53(w2, w3);
    const double r = sqrt(-2.0 * log(ua));
    double s, c;
    sincospi(2.0 * ub, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

__device__ __forceinline__ double degree_of_truth(double x, bool fuzzy, double eps)
{
    if (!fuzzy) return x > 0.0 ? 1.0 : 0.0;
    const double v = (x + eps) / (2.0 * eps);
    return fmin(fmax(v, 0.0), 1.0);
}

// one sub-step of one sub-model (reference formulas, see oracle/mcx_oracle.c for the line-by-line citations)
__device__ __forceinline__ void step_slot(const mcx_slot& sl, int scheme, int flags, double dt, double sq,
                                          const double* __restrict__ aux, double& s0, double& s1, double zc0, double zc1, double u)
{
    const double* p = sl.p;
    switch (sl.kind) {
    case MCX_MODEL_BS:
        if (scheme == MCX_SCHEME_ANALYTICAL) {
            s0 = s0 * exp(aux[0] + (zc0 - aux[1]));                       // black_scholes.py:61-67
        } else {
            s0 = s0 + (p[2] * s0 * dt + p[1] * s0 * sq * zc0);           // black_scholes.py:79-85
        }
        break;
    case MCX_MODEL_VASICEK: {
        const double r = s0;
        s1 = s1 + r * dt;                                                 // left-endpoint integral, vasicek.py:80/107
        if (scheme == MCX_SCHEME_ANALYTICAL) s0 = (p[2] + (r - p[2]) * aux[0]) + zc0;
        else s0 = r + p[3] * (p[2] - r) * dt + p[1] * sq * zc0;
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
        const double sy = sqrt(fmax(y, 0.0));
        const double yn = y + p[0] * (p[1] - y) * dt + p[2] * sy * sq * zc0;
        s1 = s1 + (y + aux[0]) * dt;
        s0 = fmax(yn, 1e-12);
        break;
    }
    case MCX_MODEL_CIRPP_DET:                                             // cirpp.py:155-172
        s1 = s1 + aux[0] * dt;
        s0 = aux[1];
        break;
    case MCX_MODEL_HESTON: {
        const double logS = s0, v = s1;
        const double sigma = p[1], rate = p[2], kappa = p[4], theta = p[5];
        if (scheme == MCX_SCHEME_EULER) {                                 // heston.py:109-121
            const double sv = sqrt(fmax(v, 0.0));
            s0 = logS + (rate - 0.5 * v) * dt + sv * sq * zc0;
            s1 = fmax(v + kappa * (theta - v) * dt + sigma * sv * sq * zc1, 0.0);
        } else {                                                          // heston.py:161-253 (Andersen QE)
            const double eps = 1e-12;
            const bool fuzzy = (flags & MCX_FLAG_SMOOTHING) != 0;
            const double m = theta + (v - theta) * aux[0];
            const double s2 = v * aux[6] + aux[7];
            const double psi = s2 / (m * m + eps);
            const double invpsi = 1.0 / (psi + eps);
            const double t = fmax(2.0 * invpsi - 1.0, 0.0);
            const double b2 = fmax(2.0 * invpsi - 1.0 + sqrt(2.0 * invpsi) * sqrt(t), 0.0);
            const double b = sqrt(b2);
            const double a = m / (1.0 + b2);
            const double v1 = a * (b + zc1) * (b + zc1);
            const double pp = fmin(fmax((psi - 1.0) / (psi + 1.0), 0.0), 1.0 - 1e-6);
            const double beta = (1.0 - pp) / (m + eps);
            const double omu = fmax(1.0 - u, eps);
            const double omp = fmax(1.0 - pp, eps);
            const double v_tail = log(omp / omu) / (beta + eps);
            const double v2 = degree_of_truth(u - pp, fuzzy, 0.3) * v_tail;
            const double w = degree_of_truth(psi - 1.5, fuzzy, 0.5);
            const double vn = (1.0 - w) * v1 + w * v2;
            const double var_int = fmax(aux[4] * v + aux[5] * vn, 0.0);
            const double vol = sqrt(fmax(var_int, eps));
            s0 = logS + rate * dt + aux[1] + aux[2] * v + aux[3] * vn + vol * zc0;
            s1 = vn;
        }
        break;
    }
    default: break;
    }
}

template <int NSLOT, int NZ, bool INJECT>
__global__ __launch_bounds__(MCX_BLOCK) void k1_paths(const K1Args a)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const int64_t ld = a.ld;
    const int D = a.n_state;
    double* __restrict__ out = a.paths + i;

    double st[NSLOT][2];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        st[s][0] = a.init_state[a.slots[s].state_off];
        st[s][1] = (a.slots[s].kind == MCX_MODEL_BS) ? 0.0 : a.init_state[a.slots[s].state_off + 1];
    }
    auto store = [&](int t) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int c = a.slots[s].state_off;
            out[((int64_t)t * D + c) * ld] = st[s][0];
            if (a.slots[s].kind != MCX_MODEL_BS) out[((int64_t)t * D + c + 1) * ld] = st[s][1];
        }
    };
    for (int t = 0; t < a.n_initial_store; ++t) store(t);

    const uint64_t path = a.path_offset + (uint64_t)i;
    for (int k = 0; k < a.n_steps; ++k) {
        const mcx_step sp = a.steps[k];                    // wave-uniform -> scalar loads
        double z[NZ], zc[NZ], u = 0.0;
        if (INJECT) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) z[j] = a.inject_z[((int64_t)k * NZ + j) * ld + i];
            if (a.n_uniform) u = a.inject_u[(int64_t)k * ld + i];
        } else {
            double ua;
#pragma unroll
            for (int q = 0; q < (NZ + 1) / 2; ++q) {
                double z0, z1;
                draw_pair(a.seed, path, (uint32_t)k, (uint32_t)q, ua, z0, z1);
                z[2 * q] = z0;
                if (2 * q + 1 < NZ) z[2 * q + 1] = z1;
            }
            if (a.n_uniform) {
                double z0, z1;
                draw_pair(a.seed, path, (uint32_t)k, (uint32_t)((NZ + 1) / 2), u, z0, z1);
            }
        }
        const double* __restrict__ L = a.chol + (int64_t)sp.chol_idx * NZ * NZ;     // model.py:48  z @ chol.T
#pragma unroll
        for (int r = 0; r < NZ; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= r; ++c) acc += L[r * NZ + c] * z[c];
            zc[r] = acc;
        }
        const double* __restrict__ ax = a.aux + (int64_t)k * NSLOT * MCX_AUX;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            // ModelConfig only hosts sub-models with simulation_dim == 1 (model_config.py:106-107); Heston runs alone
            const double zc0 = (NSLOT == 1) ? zc[0] : zc[s < NZ ? s : 0];
            const double zc1 = (NSLOT == 1 && NZ > 1) ? zc[NZ > 1 ? 1 : 0] : 0.0;
            step_slot(a.slots[s], a.scheme, a.flags | a.slots[s].flags, sp.dt, sp.sqrt_dt, ax + s * MCX_AUX,
                      st[s][0], st[s][1], zc0, zc1, u);
        }
        if (sp.store_idx >= 0) store(sp.store_idx);
    }
}

template <int NSLOT, int NZ>
int launch_k1(const K1Args& a, bool inject, hipStream_t s)
{
    const int grid = (int)((a.n + MCX_BLOCK - 1) / MCX_BLOCK);
    if (inject) hipLaunchKernelGGL((k1_paths<NSLOT, NZ, true>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k1_paths<NSLOT, NZ, false>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    return 0;
}

}  // namespace

extern "C" int mcx_sim_create(mcx_handle* h, const mcx_sim_desc* d, mcx_sim** out)
{
    if (!h || !d || !out) return -1;
    if (d->n_slots < 1 || d->n_slots > MCX_MAX_SLOTS || d->n_z > MCX_MAX_Z || d->n_state > MCX_MAX_STATE)
        MCX_FAIL(h, -2, "mcx_sim_create: dimensions out of range (slots %d, z %d, state %d)", d->n_slots, d->n_z, d->n_state);
    if (!(d->n_slots == 1 || d->n_z == d->n_slots))
        MCX_FAIL(h, -3, "mcx_sim_create: a multi-model configuration needs simulation_dim == 1 per sub-model");
    int so = 0;
    for (int s = 0; s < d->n_slots; ++s) {
        const int sd = d->slots[s].kind == MCX_MODEL_BS ? 1 : 2;
        if (d->slots[s].state_off != so) MCX_FAIL(h, -4, "mcx_sim_create: state offsets must be packed");
        so += sd;
    }
    if (so != d->n_state) MCX_FAIL(h, -4, "mcx_sim_create: n_state does not match the slots");
    for (int k = 0; k < d->n_steps; ++k) {
        if (d->steps[k].store_idx >= d->n_dates || d->steps[k].chol_idx < 0 || d->steps[k].chol_idx >= d->n_chol)
            MCX_FAIL(h, -5, "mcx_sim_create: step %d references date/cholesky out of range", k);
    }
    MCX_HIP(h, hipSetDevice(h->device));
    mcx_sim* sim = new mcx_sim();
    sim->desc = *d;
    sim->d_steps = nullptr; sim->d_chol = nullptr; sim->d_aux = nullptr;
    const size_t nb_steps = sizeof(mcx_step) * (size_t)(d->n_steps > 0 ? d->n_steps : 1);
    const size_t nb_chol = sizeof(double) * (size_t)(d->n_chol > 0 ? d->n_chol : 1) * d->n_z * d->n_z;
    const size_t nb_aux = sizeof(double) * (size_t)(d->n_steps > 0 ? d->n_steps : 1) * d->n_slots * MCX_AUX;
    MCX_HIP(h, hipMalloc(&sim->d_steps, nb_steps));
    MCX_HIP(h, hipMalloc(&sim->d_chol, nb_chol));
    MCX_HIP(h, hipMalloc(&sim->d_aux, nb_aux));
    if (d->n_steps > 0) {
        MCX_HIP(h, hipMemcpy(sim->d_steps, d->steps, sizeof(mcx_step) * d->n_steps, hipMemcpyHostToDevice));
        MCX_HIP(h, hipMemcpy(sim->d_aux, d->aux, sizeof(double) * (size_t)d->n_steps * d->n_slots * MCX_AUX, hipMemcpyHostToDevice));
    }
    if (d->n_chol > 0)
        MCX_HIP(h, hipMemcpy(sim->d_chol, d->chol, sizeof(double) * (size_t)d->n_chol * d->n_z * d->n_z, hipMemcpyHostToDevice));
    // init_state travels in the kernel-argument segment; keep a private host copy
    sim->n_state_total = d->n_state;
    double* init = new double[MCX_MAX_STATE]();
    for (int c = 0; c < d->n_state; ++c) init[c] = d->init_state[c];
    sim->desc.init_state = init;
    sim->desc.steps = nullptr; sim->desc.chol = nullptr; sim->desc.aux = nullptr;
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

extern "C" int mcx_generate_paths(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                                  int64_t ld, double* d_paths, const double* d_inject_z, const double* d_inject_u, void* stream)
{
    if (!h || !sim || !d_paths) return -1;
    if (n_paths <= 0) return 0;
    if (ld < n_paths) MCX_FAIL(h, -2, "mcx_generate_paths: ld < n_paths");
    const mcx_sim_desc& d = sim->desc;
    if (d.n_uniform && d_inject_z && !d_inject_u) MCX_FAIL(h, -3, "mcx_generate_paths: inject_u required with inject_z under QE");
    K1Args a;
    memset(&a, 0, sizeof(a));
    for (int s = 0; s < d.n_slots; ++s) a.slots[s] = d.slots[s];
    for (int c = 0; c < d.n_state; ++c) a.init_state[c] = d.init_state[c];
    a.steps = sim->d_steps; a.chol = sim->d_chol; a.aux = sim->d_aux;
    a.paths = d_paths; a.inject_z = d_inject_z; a.inject_u = d_inject_u;
    a.n = n_paths; a.ld = ld; a.seed = seed; a.path_offset = path_offset;
    a.scheme = d.scheme; a.n_steps = d.n_steps; a.n_state = d.n_state; a.n_initial_store = d.n_initial_store;
    a.flags = d.flags; a.n_uniform = d.n_uniform;
    hipStream_t s = (hipStream_t)stream;
    const bool inj = d_inject_z != nullptr;
    const int key = d.n_slots * 16 + d.n_z;
    switch (key) {
    case 1 * 16 + 1: launch_k1<1, 1>(a, inj, s); break;
    case 1 * 16 + 2: launch_k1<1, 2>(a, inj, s); break;
    case 2 * 16 + 2: launch_k1<2, 2>(a, inj, s); break;
    case 3 * 16 + 3: launch_k1<3, 3>(a, inj, s); break;
    case 4 * 16 + 4: launch_k1<4, 4>(a, inj, s); break;
    case 5 * 16 + 5: launch_k1<5, 5>(a, inj, s); break;
    case 6 * 16 + 6: launch_k1<6, 6>(a, inj, s); break;
    case 7 * 16 + 7: launch_k1<7, 7>(a, inj, s); break;
    case 8 * 16 + 8: launch_k1<8, 8>(a, inj, s); break;
    default: MCX_FAIL(h, -4, "mcx_generate_paths: unsupported (slots=%d, z=%d)", d.n_slots, d.n_z);
    }
    MCX_HIP(h, hipGetLastError());
    return 0;
}
