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

