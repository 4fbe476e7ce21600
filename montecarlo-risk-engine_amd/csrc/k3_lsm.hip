// k3_lsm.hip — K3: Longstaff-Schwartz pre-simulation step.
//
// Replaces one iteration of the reference's backward loop (controller/controller.py:316-383): roll the cached future
// cashflows one window back along the exercise policy (cf_cache, :325-352), multiply by the numeraire at the regression
// date (:368) and form the normal equations of the monomial regression that torch.linalg.lstsq solves (:370-374).
// The K x K solve (K <= 6): on the host after the cross-GPU all-reduce of the moments (mcx_lsm_step), or on the device with the
// whole backward induction of a product enqueued back to back (mcx_lsm_run: no host round trip per date).
//
// Two moment kernels:
//   * VALU + wave64 shuffle reduction (default): each lane keeps the 2K-1 + S*K running sums of its paths in VGPRs;
//   * MFMA (MCX_LSM_MFMA): the Gram matrix of the feature rows f = [1, z, .., z^(K-1), Y_0 .. Y_(S-1)] is accumulated with
//     v_mfma_f64_16x16x4_f64 — A = F^T (16 features x 4 paths), B = F (4 paths x 16 features) per instruction; the
//     features of 64 paths are staged through LDS in a bank-conflict-free [feature][path] image (row stride 66 doubles).
// Both read each path's state once (HBM-bound: 8*(1+S+atoms) B/path).
#include "mcx_device.h"

namespace {

struct K3Args {
    const DevTerm* __restrict__ terms;
    const DevEvent* __restrict__ events;   // already offset to the product's cf_begin
    const DevAtom* __restrict__ atoms;
    const double* __restrict__ coeffs;
    const double* __restrict__ paths;
    double* __restrict__ W;
    double* __restrict__ partials;         // [gridDim.x][NM]
    DevAtom num, x;
    double shift, scale;
    int64_t n, ld, ld_w;
    int32_t roll_begin, roll_end, n_basis, n_state, f32_cache;
    const DevBridge* __restrict__ bridge;
    uint8_t* __restrict__ ex_bits;         // exercise-decision record / replay (mcx_book_set_exercise_replay)
    int64_t ex_ld;
    int32_t ex_mode, ev_base;              // ev_base: book index of events[0]
    const double* etab;                    // the block's LDS copy of the 2^(j/128) table (set by the kernel, mcx_exp_tab)
    const DevVPoly* __restrict__ vpoly;    // value polynomials of the book's events (DevEvent::pad, mcx_vpoly.hip) or nullptr
    const double* __restrict__ vcoef;
};

__device__ __forceinline__ double k3_poly(const double* __restrict__ c, int K, double x)
{
    double v = 0.0, xp = 1.0;
    for (int k = 0; k < K; ++k) { v = fma(c[k], xp, v); xp *= x; }
    return v;
}

// value of one cash event for path i.  Every kind but EXERCISE is independent of the product's exercise state; for an EXERCISE
// event the function returns the state-independent pieces (immediate value `imm`, numeraire `num`, explanatory variable `x`) and
// sets `exercise`: the decision per hypothetical state is k3_exercise() — the 64-term swap value of a Bermudan swaption is then
// evaluated ONCE per path and date, not once per state.
__device__ __forceinline__ double k3_cash_event(const DevEvent& e, const K3Args& a, int64_t D, int64_t i, bool& exercise, double& imm,
                                                double& num, double& x)
{
    exercise = false;
    num = dev_atom(e.num, a.paths, D, a.ld, i);
    double common = 0.0, own = 0.0, glog = 0.0;
    if (e.kind == MCX_EV_OPTION && (e.aux[0] == 4.0 || e.aux[0] == 5.0))      // barrier options (barrier_option.py:60-223)
        return dev_barrier_event(e, a.terms, a.coeffs, a.bridge, a.paths, D, a.ld, i, num);
    if (e.kind == MCX_EV_OPTION && e.aux[0] == 3.0) {                     // binary payoff (binary_option.py:38-43): fuzzy indicator
        double val = 0.0;
        AtomCache bc = {-1, -1, 0.0};
        for (int j = e.term_begin; j < e.term_end; ++j) {
            const DevTerm tm = ldk_struct(&a.terms[j]);
            val = fma(tm.w, dev_atom_cached(tm.atom, a.paths, D, a.ld, i, bc), val);
        }
        const double dot = fmin(fmax((val - e.strike + e.aux[2]) / (2.0 * e.aux[2]), 0.0), 1.0);
        return e.aux[1] * (e.sign > 0.0 ? dot : 1.0 - dot) / num;
    }
    const bool basket = e.kind == MCX_EV_OPTION && e.aux[0] != 0.0;       // geometric aggregate needed (basket_option.py:56-82)
    AtomCache ac = {-1, -1, 0.0};
    // the event's value as ONE verified polynomial of its state variable (a Bermudan swaption's ~35-64 zero-bond prices of the short
    // rate): ~20 multiply-adds instead of ~15 instructions per term; a wave that holds a path outside the verified range runs the terms
    bool collapsed = false;
    if (e.pad > 0 && a.vpoly) {
        const DevVPoly vp = ldk_struct(&a.vpoly[e.pad - 1]);
        const double xv = a.paths[((int64_t)vp.t_idx * D + vp.col) * a.ld + i];
        if (__all(xv >= vp.lo && xv <= vp.hi)) { common = dev_vpoly(vp, a.vcoef, xv); collapsed = true; }
    }
    if (!collapsed && e.term_end > e.term_begin) {
        // the term records are read one iteration ahead (a Bermudan swaption's exercise value has up to 64 of them: the scalar
        // load of term j+1 is in flight while term j's exponential is evaluated)
        const mcx_expq_coef ec = mcx_expq_load();
        DevTerm tm = ldk_struct(&a.terms[e.term_begin]);
        for (int j = e.term_begin; j < e.term_end; ++j) {
            const DevTerm nx = ldk_struct(&a.terms[j + 1 < e.term_end ? j + 1 : j]);
            const double av = dev_atom_cached_tab(tm.atom, a.paths, D, a.ld, i, ac, a.etab, ec);
            const double v = tm.w * av;
            if (basket) glog = fma(tm.w, mcx_log(av + 1e-10), glog);
            if (tm.den < 0) common += v;
            else own += v / dev_atom(ldk_struct(&a.atoms[tm.den]), a.paths, D, a.ld, i);
            tm = nx;
        }
    }
    if (e.kind == MCX_EV_CASHFLOW) return common / num + own;
    imm = fmax(e.sign * (common - e.strike), 0.0);
    if (basket) {
        const double geo = fmax(e.sign * (mcx_exp(glog) - e.strike), 0.0);
        return (e.aux[0] == 1.0 ? geo : imm - geo + e.aux[1]) / num;
    }
    if (e.kind == MCX_EV_OPTION) return imm / num;
    exercise = true;
    x = e.coeff_off >= 0 ? dev_atom(e.x, a.paths, D, a.ld, i) : 0.0;
    return 0.0;
}

// exercise decision of one hypothetical state s (bermudan_option.py:93-131, flexicall.py:118-133 for aux[0] = 1)
__device__ __forceinline__ double k3_exercise(const DevEvent& e, const K3Args& a, double imm, double num, double x, int& s,
                                              uint8_t* __restrict__ cell, int bit)
{
    double cont = 0.0, cont_ex = 0.0;
    if (e.coeff_off >= 0) {
        cont = k3_poly(a.coeffs + e.coeff_off + s * a.n_basis, a.n_basis, x);
        if (e.aux[0] == 1.0 && s > 0) cont_ex = k3_poly(a.coeffs + e.coeff_off + (s - 1) * a.n_basis, a.n_basis, x);
    }
    const bool ex = dev_exercise_decision((imm + cont_ex > cont) && (s > 0), s, a.ex_mode, cell, bit);
    if (ex) s -= 1;
    return ex ? imm / num : 0.0;
}

// roll the window for path i; returns Y_s = numeraire * W[s] in y[] and z
template <int S>
__device__ __forceinline__ void k3_roll(const K3Args& a, int64_t i, double (&y)[S], double& z)
{
    const int64_t D = a.n_state;
    double w[S];
#pragma unroll
    for (int s = 0; s < S; ++s) w[s] = a.W[(int64_t)s * a.ld_w + i];
    if (a.roll_end > a.roll_begin) {
        int st[S];
        double sv[S];
#pragma unroll
        for (int s0 = 0; s0 < S; ++s0) { st[s0] = s0; sv[s0] = 0.0; }
        for (int q = a.roll_begin; q < a.roll_end; ++q) {                   // controller.py:333-341
            const DevEvent e = ldk_struct(&a.events[q]);
            bool exercise;
            double imm = 0.0, num = 1.0, x = 0.0;
            const double v = k3_cash_event(e, a, D, i, exercise, imm, num, x);
            uint8_t* cell = a.ex_mode ? a.ex_bits + (int64_t)(a.ev_base + q) * a.ex_ld + i : nullptr;
#pragma unroll
            for (int s0 = 0; s0 < S; ++s0) {
                sv[s0] += exercise ? k3_exercise(e, a, imm, num, x, st[s0], cell, s0) : v;
                if (a.f32_cache) sv[s0] = (double)(float)sv[s0];             // float32 cf_cache quirk (controller.py:312-330)
            }
        }
        double wn[S];
#pragma unroll
        for (int s0 = 0; s0 < S; ++s0) {
            double tail = w[0];
#pragma unroll
            for (int q = 1; q < S; ++q) tail = (st[s0] == q) ? w[q] : tail;  // lookup_state_values (product.py:150-155)
            const double total = sv[s0] + tail;
            wn[s0] = a.f32_cache ? (double)(float)total : total;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) { w[s] = wn[s]; a.W[(int64_t)s * a.ld_w + i] = wn[s]; }
    }
    const double num = dev_atom(a.num, a.paths, D, a.ld, i);
    const double x = dev_atom(a.x, a.paths, D, a.ld, i);
    z = (x - a.shift) * a.scale;
#pragma unroll
    for (int s = 0; s < S; ++s) y[s] = num * w[s];
}

// block reduction of NM per-lane accumulators with a single barrier: wave64 shuffles, one LDS row per wave, thread q sums the
// four rows (the per-accumulator block_sum paid two barriers for each of the NM values)
template <int NM>
__device__ __forceinline__ void k3_block_reduce(const double (&acc)[NM], double* __restrict__ dst)
{
    __shared__ double rows[4][NM];
    const int lane = threadIdx.x & (MCX_WAVE - 1), wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NM; ++q) {
        const double r = wave_sum(acc[q]);
        if (lane == 0) rows[wv][q] = r;
    }
    __syncthreads();
    if (threadIdx.x < NM) dst[threadIdx.x] = (rows[0][threadIdx.x] + rows[1][threadIdx.x]) + (rows[2][threadIdx.x] + rows[3][threadIdx.x]);
}

template <int K, int S>
__global__ __launch_bounds__(MCX_BLOCK) void k3_step_valu(const K3Args a_in)
{
    constexpr int NM = (2 * K - 1) + S * K;
    __shared__ double etab[MCX_EXP_LDS_DOUBLES];
    mcx_exp_tab_load(etab);
    __syncthreads();
    K3Args a = a_in;
    a.etab = etab;
    double acc[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) acc[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        double y[S], z;
        k3_roll<S>(a, i, y, z);
        double zp = 1.0;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            acc[k] += zp;
            if (k < K) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[(2 * K - 1) + s * K + k] = fma(zp, y[s], acc[(2 * K - 1) + s * K + k]);
            }
            zp *= z;
        }
    }
    k3_block_reduce<NM>(acc, a.partials + (int64_t)blockIdx.x * NM);
}

// ---- MFMA variant ----------------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));
#define K3_FROW 66   // LDS row stride (doubles) of the [feature][path] image: 2*feature mod 32 spreads the 16 rows over banks

template <int K, int S>
__global__ __launch_bounds__(MCX_BLOCK) void k3_step_mfma(const K3Args a_in)
{
    constexpr int NF = K + S;                  // features per path (<= 16)
    constexpr int NM = (2 * K - 1) + S * K;
    __shared__ double etab[MCX_EXP_LDS_DOUBLES];
    mcx_exp_tab_load(etab);
    __syncthreads();
    K3Args a = a_in;
    a.etab = etab;
    __shared__ double feat[4][16 * K3_FROW];   // one image per wave
    __shared__ double gram[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* img = feat[wv];
    // zero the padding rows once (features NF..15 stay 0 for the whole kernel)
    for (int r = NF; r < 16; ++r) img[r * K3_FROW + lane] = 0.0;
    double4_t accm = {0.0, 0.0, 0.0, 0.0};
    const int fi = lane & 15, pk = lane >> 4;
    const int64_t stride = (int64_t)gridDim.x * MCX_BLOCK;
    // every wave runs the same number of iterations (EXEC must be all ones around the MFMA)
    const int64_t iters = (a.n + stride - 1) / stride;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t i = it * stride + (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
        double y[S], z = 0.0;
        const bool live = i < a.n;
#pragma unroll
        for (int s = 0; s < S; ++s) y[s] = 0.0;
        if (live) k3_roll<S>(a, i, y, z);
        double zp = live ? 1.0 : 0.0;           // dead lanes contribute an all-zero feature row
#pragma unroll
        for (int k = 0; k < K; ++k) { img[k * K3_FROW + lane] = zp; zp *= z; }
#pragma unroll
        for (int s = 0; s < S; ++s) img[(K + s) * K3_FROW + lane] = y[s];
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): this wave's LDS writes have landed (same-wave RAW)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const double v = img[fi * K3_FROW + 4 * g + pk];   // A[i=fi][k=pk] and B[k=pk][j=fi] are the same number
            accm = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, accm, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // accumulator layout of v_mfma_f64_16x16x4_f64 (probed on gfx950 with tools/mfma_probe.hip):
    //   D[row = 4*r + lane/16][col = lane%16] = accm[r]
#pragma unroll
    for (int r = 0; r < 4; ++r) gram[wv][(4 * r + pk) * 16 + fi] = accm[r];
    __syncthreads();
    if (threadIdx.x < NM) {
        const int q = threadIdx.x;
        int row, col;
        if (q < 2 * K - 1) { row = q < K ? 0 : q - (K - 1); col = q < K ? q : K - 1; }     // sum z^q = G[row][col], row+col = q
        else { const int s = (q - (2 * K - 1)) / K, k = (q - (2 * K - 1)) % K; row = k; col = K + s; }
        double r = 0.0;
        for (int w = 0; w < 4; ++w) r += gram[w][row * 16 + col];
        a.partials[(int64_t)blockIdx.x * NM + q] = r;
    }
}

// ---- product-batched step (books of thousands of products) --------------------------------------------
// One launch runs the LSM step of MANY products at once: blockIdx.y = job, every job carries its own roll window, atoms,
// shift/scale and cashflow-cache block.  The per-(product, date) launch + host round trip of mcx_lsm_step (~40 us each,
// 800 k of them for the reference's 5,000-product book) becomes one launch per backward step of the whole book.
struct K3Job {
    int32_t ev_off, roll_begin, roll_end, pad;
    int64_t w_off;                          // offset (doubles) of this product's [S][ld_w] cashflow cache in W
    double shift, scale;
    DevAtom num, x;
};

template <int K, int S>
__global__ __launch_bounds__(MCX_BLOCK) void k3_step_batch(K3Args a, const K3Job* __restrict__ jobs, int blocks_per_job)
{
    constexpr int NM = (2 * K - 1) + S * K;
    __shared__ double etab[MCX_EXP_LDS_DOUBLES];
    mcx_exp_tab_load(etab);
    __syncthreads();
    a.etab = etab;
    const K3Job jb = ldk_struct(&jobs[blockIdx.y]);
    a.events += jb.ev_off; a.ev_base = jb.ev_off; a.roll_begin = jb.roll_begin; a.roll_end = jb.roll_end; a.W += jb.w_off;
    a.shift = jb.shift; a.scale = jb.scale; a.num = jb.num; a.x = jb.x;
    double acc[NM];
#pragma unroll
    for (int q = 0; q < NM; ++q) acc[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        double y[S], z;
        k3_roll<S>(a, i, y, z);
        double zp = 1.0;
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            acc[k] += zp;
            if (k < K) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[(2 * K - 1) + s * K + k] = fma(zp, y[s], acc[(2 * K - 1) + s * K + k]);
            }
            zp *= z;
        }
    }
    k3_block_reduce<NM>(acc, a.partials + ((int64_t)blockIdx.y * blocks_per_job + blockIdx.x) * NM);
}

// out[job][q] = sum over the job's blocks (fixed order: deterministic)
__global__ void k3_finish_batch(const double* __restrict__ partials, int nm, int blocks_per_job, double* __restrict__ out, int ld_out)
{
    const int job = blockIdx.x, q = threadIdx.x;
    if (q >= nm) return;
    double s = 0.0;
    for (int b = 0; b < blocks_per_job; ++b) s += partials[((int64_t)job * blocks_per_job + b) * nm + q];
    out[(int64_t)job * ld_out + q] = s;
}

__global__ void k3_scatter_coeffs(const int64_t* __restrict__ offsets, const double* __restrict__ values, int len, double* __restrict__ coeffs)
{
    const int j = blockIdx.x;
    for (int q = threadIdx.x; q < len; q += blockDim.x) coeffs[offsets[j] + q] = values[(int64_t)j * len + q];
}

__global__ void k3_finish(const double* __restrict__ partials, int nm, int n_blocks, double* __restrict__ out)
{
    const int q = blockIdx.x;
    __shared__ double lds[4];
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) s += partials[(int64_t)b * nm + q];
    s = block_sum(s, lds);
    if (threadIdx.x == 0) out[q] = s;
}

__global__ __launch_bounds__(MCX_BLOCK) void k3_minmax(const DevAtom* __restrict__ atoms, const int32_t* __restrict__ ids,
                                                       const double* __restrict__ paths, int64_t D, int64_t n, int64_t ld,
                                                       double* __restrict__ partials)
{
    const int q = blockIdx.y;
    const DevAtom at = ldk_struct(&atoms[ldk(ids + q)]);
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const double x = dev_atom(at, paths, D, ld, i);
        lo = fmin(lo, x); hi = fmax(hi, x);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    __shared__ double l[4], u[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { l[w] = lo; u[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { lo = fmin(lo, l[k]); hi = fmax(hi, u[k]); }
        partials[((int64_t)q * gridDim.x + blockIdx.x) * 2 + 0] = lo;
        partials[((int64_t)q * gridDim.x + blockIdx.x) * 2 + 1] = hi;
    }
}

// the same over raw rows of a [rows][ld] tensor (state columns of the pre-simulation paths: ranges of the value polynomials)
__global__ __launch_bounds__(MCX_BLOCK) void k3_row_minmax(const double* __restrict__ x, int64_t n, int64_t ld, double* __restrict__ partials)
{
    const int q = blockIdx.y;
    const double* __restrict__ row = x + (int64_t)q * ld;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const double v = row[i];
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    __shared__ double l[4], u[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { l[w] = lo; u[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { lo = fmin(lo, l[k]); hi = fmax(hi, u[k]); }
        partials[((int64_t)q * gridDim.x + blockIdx.x) * 2 + 0] = lo;
        partials[((int64_t)q * gridDim.x + blockIdx.x) * 2 + 1] = hi;
    }
}

__global__ void k3_minmax_finish(const double* __restrict__ partials, int n_blocks, double* __restrict__ out)
{
    const int q = blockIdx.x;
    double lo = INFINITY, hi = -INFINITY;
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        lo = fmin(lo, partials[((int64_t)q * n_blocks + b) * 2 + 0]);
        hi = fmax(hi, partials[((int64_t)q * n_blocks + b) * 2 + 1]);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (threadIdx.x == 0) { out[2 * q] = lo; out[2 * q + 1] = hi; }
}

// ---- device-side normal-equation solve (mcx_lsm_run) ---------------------------------------------------------------------
// One date of the backward induction: moments of the shifted / scaled basis -> least-squares coefficients in the RAW monomial
// basis, written straight into the book's coefficient array (what the next date's roll reads) and into the result table.
// Same algorithm as the host solver (mcx/plan.py solve_normal_equations): LU with partial pivoting of the K x K Gram matrix,
// back-transformation z^k = scale^k (x - shift)^k, and the minimum-norm solution of the exactly rank-1 system of a date on
// which every path shares x = x0 (the calibration date).  K <= 6, S <= 8: one lane.
struct K3Solve {
    double shift, scale, x0;
    int64_t off0, off1;          // coefficient offsets in the book (-1: none)
    int32_t degenerate, K, S, date;
};

// KK / SS > 0: basis size / state count known at compile time — every loop unrolls and the K x K system lives in registers (with
// run-time bounds the local arrays sit in scratch memory: ~11 us for a 3 x 3 solve by one thread, mostly scratch latency).
// Pivoting by conditional row swaps with static indices: after the r-loop row c holds the largest |entry| of column c, as with
// LAPACK's single swap; the remaining rows may be ordered differently, the solution is the same up to rounding.
template <int KK, int SS>
__device__ __forceinline__ void k3_solve_t(const double* __restrict__ m, const K3Solve& q, double* __restrict__ coeffs, double* __restrict__ table,
                                           int32_t* __restrict__ status)
{
    const int K = KK ? KK : q.K, S = SS ? SS : q.S;
    constexpr int KA = KK ? KK : MCX_MAX_BASIS, SA = SS ? SS : MCX_MAX_STATES;
    double out[SA][KA];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < K; ++k) out[s][k] = 0.0;
    const double n = m[0];
    int st = 0;
    if (n > 0.0 && q.degenerate) {
        double v[KA], vv = 0.0, xp = 1.0;
#pragma unroll
        for (int k = 0; k < K; ++k) { v[k] = xp; vv += xp * xp; xp *= q.x0; }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double mean_y = m[(2 * K - 1) + s * K] / n;
#pragma unroll
            for (int k = 0; k < K; ++k) out[s][k] = v[k] * (mean_y / vv);
        }
    } else if (n > 0.0) {
        double G[KA][KA], B[KA][SA];
        double gmax = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int k = 0; k < K; ++k) { G[j][k] = m[j + k]; gmax = fmax(gmax, fabs(G[j][k])); }
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int s = 0; s < S; ++s) B[k][s] = m[(2 * K - 1) + s * K + k];
#pragma unroll
        for (int c = 0; c < K; ++c) {                                  // LU, partial pivoting
#pragma unroll
            for (int r = c + 1; r < K; ++r) {
                const bool sw = fabs(G[r][c]) > fabs(G[c][c]);
#pragma unroll
                for (int k = 0; k < K; ++k) { const double x = G[c][k], y = G[r][k]; G[c][k] = sw ? y : x; G[r][k] = sw ? x : y; }
#pragma unroll
                for (int s = 0; s < S; ++s) { const double x = B[c][s], y = B[r][s]; B[c][s] = sw ? y : x; B[r][s] = sw ? x : y; }
            }
            if (!(fabs(G[c][c]) > 1e-14 * gmax)) st = 1;               // numerically singular: the caller re-solves on the host
            const double piv = st ? 1.0 : G[c][c];
#pragma unroll
            for (int r = c + 1; r < K; ++r) {
                const double f = G[r][c] / piv;
#pragma unroll
                for (int k = c + 1; k < K; ++k) G[r][k] -= f * G[c][k];
#pragma unroll
                for (int s = 0; s < S; ++s) B[r][s] -= f * B[c][s];
            }
        }
        if (st == 0) {
#pragma unroll
            for (int c = K - 1; c >= 0; --c)
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    double acc = B[c][s];
#pragma unroll
                    for (int k = c + 1; k < K; ++k) acc -= G[c][k] * B[k][s];
                    B[c][s] = acc / G[c][c];
                }
            // T[j][k] = coefficient of x^j in z^k = scale^k C(k, j) (-shift)^(k-j)
            double sp = 1.0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double binom = 1.0;
#pragma unroll
                for (int j = 0; j <= k; ++j) {
                    double ms = 1.0;
                    for (int e = 0; e < k - j; ++e) ms *= -q.shift;
                    const double T = sp * binom * ms;
#pragma unroll
                    for (int s = 0; s < S; ++s) out[s][j] += T * B[k][s];
                    binom = binom * (double)(k - j) / (double)(j + 1);
                }
                sp *= q.scale;
            }
        }
    }
    status[q.date] = st;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double c = out[s][k];
            if (table) table[((int64_t)q.date * S + s) * K + k] = c;
            if (st == 0) {
                if (q.off0 >= 0) coeffs[q.off0 + s * K + k] = c;
                if (q.off1 >= 0) coeffs[q.off1 + s * K + k] = c;
            }
        }
}

__device__ void k3_solve_body(const double* __restrict__ m, const K3Solve& q, double* __restrict__ coeffs, double* __restrict__ table,
                              int32_t* __restrict__ status)
{
    const int key = q.K * 16 + q.S;
    switch (key) {
    case 2 * 16 + 1: k3_solve_t<2, 1>(m, q, coeffs, table, status); break;
    case 2 * 16 + 2: k3_solve_t<2, 2>(m, q, coeffs, table, status); break;
    case 3 * 16 + 1: k3_solve_t<3, 1>(m, q, coeffs, table, status); break;
    case 3 * 16 + 2: k3_solve_t<3, 2>(m, q, coeffs, table, status); break;
    case 3 * 16 + 3: k3_solve_t<3, 3>(m, q, coeffs, table, status); break;
    case 4 * 16 + 1: k3_solve_t<4, 1>(m, q, coeffs, table, status); break;
    case 4 * 16 + 2: k3_solve_t<4, 2>(m, q, coeffs, table, status); break;
    default: k3_solve_t<0, 0>(m, q, coeffs, table, status); break;
    }
}

__global__ void k3_solve(const double* __restrict__ m, const K3Solve q, double* __restrict__ coeffs, double* __restrict__ table,
                         int32_t* __restrict__ status)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    k3_solve_body(m, q, coeffs, table, status);
}

// finish + solve in ONE launch (mcx_lsm_run on one GPU: the moments need no all-reduce between the two): wave q of a 1024-thread
// block sums moment q over the step kernel's per-block partials (fixed order: deterministic), thread 0 solves.  Two launches per
// regression date instead of three — 121 dates of a Bermudan are latency: ~5 us + a kernel boundary each.
__global__ __launch_bounds__(1024) void k3_finish_solve(const double* __restrict__ partials, int nm, int n_blocks, double* __restrict__ mom_out,
                                                        const K3Solve q, double* __restrict__ coeffs, double* __restrict__ table,
                                                        int32_t* __restrict__ status)
{
    __shared__ double m[(2 * MCX_MAX_BASIS - 1) + MCX_MAX_STATES * MCX_MAX_BASIS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = wv; j < nm; j += 16) {
        double s = 0.0;
        for (int b = lane; b < n_blocks; b += MCX_WAVE) s += partials[(int64_t)b * nm + j];
        s = wave_sum(s);
        if (lane == 0) { m[j] = s; mom_out[j] = s; }
    }
    __syncthreads();
    if (threadIdx.x == 0) k3_solve_body(m, q, coeffs, table, status);
}

template <int K, int S>
void launch_k3(const K3Args& a, int grid, bool mfma, hipStream_t s)
{
    if (mfma) hipLaunchKernelGGL((k3_step_mfma<K, S>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k3_step_valu<K, S>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
}

template <int S>
int dispatch_k3(int K, const K3Args& a, int grid, bool mfma, hipStream_t s)
{
    switch (K) {
    case 1: launch_k3<1, S>(a, grid, mfma, s); return 0;
    case 2: launch_k3<2, S>(a, grid, mfma, s); return 0;
    case 3: launch_k3<3, S>(a, grid, mfma, s); return 0;
    case 4: launch_k3<4, S>(a, grid, mfma, s); return 0;
    case 5: launch_k3<5, S>(a, grid, mfma, s); return 0;
    case 6: launch_k3<6, S>(a, grid, mfma, s); return 0;
    default: return -1;
    }
}

template <int S>
int dispatch_k3_batch(int K, const K3Args& a, const K3Job* jobs, dim3 grid, hipStream_t s)
{
    switch (K) {
    case 1: hipLaunchKernelGGL((k3_step_batch<1, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    case 2: hipLaunchKernelGGL((k3_step_batch<2, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    case 3: hipLaunchKernelGGL((k3_step_batch<3, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    case 4: hipLaunchKernelGGL((k3_step_batch<4, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    case 5: hipLaunchKernelGGL((k3_step_batch<5, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    case 6: hipLaunchKernelGGL((k3_step_batch<6, S>), grid, dim3(MCX_BLOCK), 0, s, a, jobs, (int)grid.x); return 0;
    default: return -1;
    }
}

}  // namespace

extern "C" int mcx_lsm_stats(mcx_handle* h, const mcx_book* b, const int32_t* h_atom_ids, int32_t n_ids, const double* d_paths,
                             int64_t n_paths, int64_t ld, double* h_out, void* stream)
{
    if (!h || !b || !h_atom_ids || !d_paths || !h_out) return -1;
    if (n_ids <= 0) return 0;
    for (int q = 0; q < n_ids; ++q)
        if (h_atom_ids[q] < 0 || h_atom_ids[q] >= b->n_atoms) MCX_FAIL(h, -2, "mcx_lsm_stats: atom id out of range");
    if (n_paths <= 0) { for (int q = 0; q < n_ids; ++q) { h_out[2 * q] = INFINITY; h_out[2 * q + 1] = -INFINITY; } return 0; }
    hipStream_t s = (hipStream_t)stream;
    const int grid = mcx_grid_for(n_paths, MCX_BLOCK, 256);
    if ((size_t)n_ids * grid * 2 * sizeof(double) + (size_t)n_ids * 2 * sizeof(double) > h->ws_bytes || n_ids > 65535 ||
        (size_t)n_ids * 2 * sizeof(double) > h->pinned_bytes)
        MCX_FAIL(h, -2, "mcx_lsm_stats: too many atoms in one call");
    const int32_t* d_ids = (const int32_t*)mcx_stage_small(h, h_atom_ids, sizeof(int32_t) * (size_t)n_ids, s);
    if (!d_ids) return -100;
    double* part = h->d_ws;
    double* d_out = h->d_ws + (size_t)n_ids * grid * 2;
    hipLaunchKernelGGL(k3_minmax, dim3(grid, n_ids), dim3(MCX_BLOCK), 0, s, b->d_atoms, d_ids, d_paths, (int64_t)b->n_state, n_paths, ld, part);
    hipLaunchKernelGGL(k3_minmax_finish, dim3(n_ids), dim3(MCX_WAVE), 0, s, part, grid, d_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h->h_pinned, d_out, sizeof(double) * 2 * (size_t)n_ids, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_out, h->h_pinned, sizeof(double) * 2 * (size_t)n_ids);
    return 0;
}

extern "C" int mcx_rows_minmax(mcx_handle* h, const double* d_x, int32_t n_rows, int64_t n, int64_t ld, double* h_out, void* stream)
{
    if (!h || !d_x || !h_out) return -1;
    if (n_rows <= 0) return 0;
    if (ld < n) MCX_FAIL(h, -2, "mcx_rows_minmax: ld < n");
    if (n <= 0) { for (int q = 0; q < n_rows; ++q) { h_out[2 * q] = INFINITY; h_out[2 * q + 1] = -INFINITY; } return 0; }
    hipStream_t s = (hipStream_t)stream;
    // rows are short reductions: enough blocks per row to fill the chip across all rows, at most 64
    int grid = mcx_grid_for(n, MCX_BLOCK * 8, 64);
    if (n_rows > 65535 || (size_t)n_rows * 2 * sizeof(double) > h->pinned_bytes) MCX_FAIL(h, -2, "mcx_rows_minmax: too many rows in one call");
    double* part = (double*)mcx_scratch(h, 2, sizeof(double) * 2 * (size_t)n_rows * (grid + 1));
    if (!part) return -100;
    double* d_out = part + (size_t)n_rows * grid * 2;
    hipLaunchKernelGGL(k3_row_minmax, dim3(grid, n_rows), dim3(MCX_BLOCK), 0, s, d_x, n, ld, part);
    hipLaunchKernelGGL(k3_minmax_finish, dim3(n_rows), dim3(MCX_WAVE), 0, s, part, grid, d_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h->h_pinned, d_out, sizeof(double) * 2 * (size_t)n_rows, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_out, h->h_pinned, sizeof(double) * 2 * (size_t)n_rows);
    return 0;
}

// roll + moments of one (product, date) on the stream: d_moments[NM] (device)
static int lsm_step_launch(mcx_handle* h, const mcx_book* b, int32_t product, int32_t roll_begin, int32_t roll_end, int32_t num_atom,
                           int32_t x_atom, double shift, double scale, const double* d_paths, int64_t n_paths, int64_t ld,
                           double* d_W, int64_t ld_w, double* d_moments, int32_t flags, hipStream_t s, const char* who,
                           int* grid_out = nullptr)          // grid_out: the caller sums the per-block partials (h->d_ws) itself
{
    if (product < 0 || product >= b->n_products) MCX_FAIL(h, -2, "%s: product out of range", who);
    const DevProduct& pr = b->h_products[product];
    const int n_cf = pr.cf_end - pr.cf_begin;
    if (roll_begin < 0 || roll_end < roll_begin || roll_end > n_cf) MCX_FAIL(h, -2, "%s: roll window out of range", who);
    if (num_atom < 0 || num_atom >= b->n_atoms || x_atom < 0 || x_atom >= b->n_atoms) MCX_FAIL(h, -2, "%s: atom out of range", who);
    if (ld < n_paths || ld_w < n_paths) MCX_FAIL(h, -2, "%s: leading dimension < n_paths", who);
    const int K = b->n_basis, S = pr.n_states;
    const int NM = (2 * K - 1) + S * K;
    if (grid_out) *grid_out = 0;
    if (n_paths <= 0) { MCX_HIP(h, hipMemsetAsync(d_moments, 0, sizeof(double) * NM, s)); return 0; }
    const int grid = mcx_grid_for(n_paths, MCX_BLOCK, 4 * h->n_cu);
    if ((size_t)grid * NM * sizeof(double) > h->ws_bytes) MCX_FAIL(h, -2, "%s: workspace too small", who);
    auto flat = [&](int id) { DevAtom o; const mcx_atom& q = b->h_atoms[id]; o.t_idx = q.t_idx; o.col = q.col; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1; return o; };
    K3Args a;
    a.etab = nullptr;
    a.terms = b->d_terms; a.events = b->d_events + pr.cf_begin; a.atoms = b->d_atoms; a.coeffs = b->d_coeffs; a.paths = d_paths;
    a.W = d_W; a.partials = h->d_ws; a.num = flat(num_atom); a.x = flat(x_atom); a.shift = shift; a.scale = scale;
    a.n = n_paths; a.ld = ld; a.ld_w = ld_w; a.roll_begin = roll_begin; a.roll_end = roll_end; a.n_basis = K; a.n_state = b->n_state;
    a.f32_cache = (flags & MCX_LSM_F32_CACHE) ? 1 : 0; a.bridge = b->d_bridge;
    a.ex_mode = b->ex_mode; a.ex_bits = b->d_ex_bits; a.ex_ld = b->ex_ld; a.ev_base = pr.cf_begin;
    a.vpoly = b->d_vpoly; a.vcoef = b->d_vcoef;
    if (a.ex_mode && a.ex_ld < n_paths) MCX_FAIL(h, -2, "%s: exercise replay buffer narrower than the path count", who);
    const bool mfma = (flags & MCX_LSM_MFMA) != 0;
    int rc = -1;
    switch (S) {
    case 1: rc = dispatch_k3<1>(K, a, grid, mfma, s); break;
    case 2: rc = dispatch_k3<2>(K, a, grid, mfma, s); break;
    case 3: rc = dispatch_k3<3>(K, a, grid, mfma, s); break;
    case 4: rc = dispatch_k3<4>(K, a, grid, mfma, s); break;
    case 5: rc = dispatch_k3<5>(K, a, grid, mfma, s); break;
    case 6: rc = dispatch_k3<6>(K, a, grid, mfma, s); break;
    case 7: rc = dispatch_k3<7>(K, a, grid, mfma, s); break;
    case 8: rc = dispatch_k3<8>(K, a, grid, mfma, s); break;
    default: break;
    }
    if (rc != 0) MCX_FAIL(h, -3, "%s: unsupported (basis=%d, states=%d)", who, K, S);
    MCX_HIP(h, hipGetLastError());
    if (grid_out) { *grid_out = grid; return 0; }
    hipLaunchKernelGGL(k3_finish, dim3(NM), dim3(MCX_BLOCK), 0, s, h->d_ws, NM, grid, d_moments);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_lsm_step(mcx_handle* h, const mcx_book* b, int32_t product, int32_t roll_begin, int32_t roll_end, int32_t num_atom,
                            int32_t x_atom, double shift, double scale, const double* d_paths, int64_t n_paths, int64_t ld,
                            double* d_W, int64_t ld_w, double* d_moments, int32_t flags, void* stream)
{
    if (!h || !b || !d_paths || !d_W || !d_moments) return -1;
    return lsm_step_launch(h, b, product, roll_begin, roll_end, num_atom, x_atom, shift, scale, d_paths, n_paths, ld, d_W, ld_w,
                           d_moments, flags, (hipStream_t)stream, "mcx_lsm_step");
}

extern "C" int mcx_lsm_solve(mcx_handle* h, mcx_book* b, int32_t product, const double* d_moments, const mcx_lsm_date* date,
                             int32_t date_index, double* d_table, int32_t* d_status, void* stream)
{
    if (!h || !b || !d_moments || !date || !d_table || !d_status) return -1;
    if (product < 0 || product >= b->n_products || date_index < 0) MCX_FAIL(h, -2, "mcx_lsm_solve: product / date index out of range");
    const int K = b->n_basis, S = b->h_products[product].n_states;
    for (int w = 0; w < 2; ++w)
        if (date->coeff_off[w] >= 0 && date->coeff_off[w] + (int64_t)S * K > b->n_coeffs) MCX_FAIL(h, -2, "mcx_lsm_solve: coefficient offset out of range");
    K3Solve sv;
    sv.shift = date->shift; sv.scale = date->scale; sv.x0 = date->x0; sv.off0 = date->coeff_off[0]; sv.off1 = date->coeff_off[1];
    sv.degenerate = date->degenerate; sv.K = K; sv.S = S; sv.date = date_index;
    hipLaunchKernelGGL(k3_solve, dim3(1), dim3(64), 0, (hipStream_t)stream, d_moments, sv, b->d_coeffs, d_table, d_status);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_lsm_run(mcx_handle* h, mcx_book* b, int32_t product, const mcx_lsm_date* h_dates, int32_t n_dates,
                           const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w,
                           double* h_coeffs, int32_t* h_status, int32_t flags, void* stream)
{
    if (!h || !b || !h_dates || !d_paths || !d_W || !h_coeffs || !h_status) return -1;
    if (n_dates <= 0) return 0;
    if (product < 0 || product >= b->n_products) MCX_FAIL(h, -2, "mcx_lsm_run: product out of range");
    const int K = b->n_basis, S = b->h_products[product].n_states, NM = (2 * K - 1) + S * K;
    const size_t tab_bytes = sizeof(double) * (size_t)n_dates * S * K, st_bytes = sizeof(int32_t) * (size_t)n_dates;
    for (int d = 0; d < n_dates; ++d)
        for (int w = 0; w < 2; ++w)
            if (h_dates[d].coeff_off[w] >= 0 && h_dates[d].coeff_off[w] + (int64_t)S * K > b->n_coeffs)
                MCX_FAIL(h, -2, "mcx_lsm_run: date %d coefficient offset out of range", d);
    hipStream_t s = (hipStream_t)stream;
    // workspace of the run: [moments NM | table n_dates*S*K | status n_dates], a scratch buffer of the handle
    double* d_ws = (double*)mcx_scratch(h, 1, sizeof(double) * NM + tab_bytes + st_bytes + 64);
    if (!d_ws) return -100;
    double* d_mom = d_ws;
    double* d_tab = d_ws + NM;
    int32_t* d_st = (int32_t*)(d_tab + (size_t)n_dates * S * K);
    int rc = 0;
    for (int d = 0; d < n_dates && rc == 0; ++d) {
        const mcx_lsm_date& q = h_dates[d];
        const bool multi = h->comm && h->comm_ranks > 1;
        int grid = 0;
        rc = lsm_step_launch(h, b, product, q.roll_begin, q.roll_end, q.num_atom, q.x_atom, q.shift, q.scale, d_paths, n_paths, ld, d_W, ld_w,
                             d_mom, flags, s, "mcx_lsm_run", multi ? nullptr : &grid);
        if (rc != 0) break;
        if (multi) { rc = mcx_allreduce_f64(h, d_mom, NM, stream); if (rc != 0) break; }   // stream-ordered
        K3Solve sv;
        sv.shift = q.shift; sv.scale = q.scale; sv.x0 = q.x0; sv.off0 = q.coeff_off[0]; sv.off1 = q.coeff_off[1];
        sv.degenerate = q.degenerate; sv.K = K; sv.S = S; sv.date = d;
        if (!multi && grid > 0) hipLaunchKernelGGL(k3_finish_solve, dim3(1), dim3(1024), 0, s, h->d_ws, NM, grid, d_mom, sv, b->d_coeffs, d_tab, d_st);
        else hipLaunchKernelGGL(k3_solve, dim3(1), dim3(64), 0, s, d_mom, sv, b->d_coeffs, d_tab, d_st);
        if (hipGetLastError() != hipSuccess) { h->err = "mcx_lsm_run: launch failed"; rc = -100; }
    }
    if (rc == 0 && (hipMemcpyAsync(h_coeffs, d_tab, tab_bytes, hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipMemcpyAsync(h_status, d_st, st_bytes, hipMemcpyDeviceToHost, s) != hipSuccess)) { h->err = "mcx_lsm_run: copy failed"; rc = -100; }
    if (hipStreamSynchronize(s) != hipSuccess && rc == 0) { h->err = "mcx_lsm_run: synchronise failed"; rc = -100; }
    return rc;
}

// h_moments: host output (synchronises) or NULL; d_moments: device output [n_jobs][NM] (stream-ordered) or NULL
static int lsm_step_batch_impl(mcx_handle* h, const mcx_book* b, const mcx_lsm_job* h_jobs, int32_t n_jobs, int32_t n_states,
                               const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                               double* h_moments, double* d_moments, int32_t flags, void* stream)
{
    if (!h || !b || !h_jobs || !d_paths || !d_W || (!h_moments && !d_moments)) return -1;
    if (n_jobs <= 0) return 0;
    const int K = b->n_basis, S = n_states;
    if (S < 1 || S > MCX_MAX_STATES) MCX_FAIL(h, -2, "mcx_lsm_step_batch: n_states out of range");
    const int NM = (2 * K - 1) + S * K;
    if (ld < n_paths || ld_w < n_paths) MCX_FAIL(h, -2, "mcx_lsm_step_batch: leading dimension < n_paths");
    hipStream_t s = (hipStream_t)stream;
    if (n_paths <= 0) {
        if (h_moments) memset(h_moments, 0, sizeof(double) * (size_t)n_jobs * NM);
        if (d_moments) MCX_HIP(h, hipMemsetAsync(d_moments, 0, sizeof(double) * (size_t)n_jobs * NM, s));
        return 0;
    }
    auto flat = [&](int id) { DevAtom o; const mcx_atom& q = b->h_atoms[id]; o.t_idx = q.t_idx; o.col = q.col; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1; return o; };
    std::vector<K3Job> jobs((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        const mcx_lsm_job& q = h_jobs[j];
        if (q.product < 0 || q.product >= b->n_products) MCX_FAIL(h, -2, "mcx_lsm_step_batch: job %d product out of range", j);
        const DevProduct& pr = b->h_products[q.product];
        if (pr.n_states != S) MCX_FAIL(h, -2, "mcx_lsm_step_batch: job %d has %d states, the batch %d", j, pr.n_states, S);
        if (q.roll_begin < 0 || q.roll_end < q.roll_begin || q.roll_end > pr.cf_end - pr.cf_begin) MCX_FAIL(h, -2, "mcx_lsm_step_batch: job %d roll window", j);
        if (q.num_atom < 0 || q.num_atom >= b->n_atoms || q.x_atom < 0 || q.x_atom >= b->n_atoms) MCX_FAIL(h, -2, "mcx_lsm_step_batch: job %d atoms", j);
        if (q.w_offset < 0 || q.w_offset + (int64_t)S * ld_w > w_len) MCX_FAIL(h, -2, "mcx_lsm_step_batch: job %d cache block outside d_W", j);
        K3Job& o = jobs[j];
        o.ev_off = pr.cf_begin; o.roll_begin = q.roll_begin; o.roll_end = q.roll_end; o.pad = 0; o.w_off = q.w_offset;
        o.shift = q.shift; o.scale = q.scale; o.num = flat(q.num_atom); o.x = flat(q.x_atom);
    }
    // few paths per product are the norm for big books: one block per 256 paths, capped
    int bpj = mcx_grid_for(n_paths, MCX_BLOCK, 64);
    const int max_jobs = 32768;
    const int chunk = n_jobs < max_jobs ? n_jobs : max_jobs;
    K3Job* d_jobs = (K3Job*)mcx_scratch(h, 1, sizeof(K3Job) * (size_t)chunk);
    double* d_part = (double*)mcx_scratch(h, 2, sizeof(double) * (size_t)chunk * bpj * NM);
    double* d_out = (double*)mcx_scratch(h, 3, sizeof(double) * (size_t)chunk * NM);
    if (!d_jobs || !d_part || !d_out) return -100;
    K3Args a;
    a.etab = nullptr;
    memset(&a, 0, sizeof(a));
    a.terms = b->d_terms; a.events = b->d_events; a.atoms = b->d_atoms; a.coeffs = b->d_coeffs; a.paths = d_paths;
    a.W = d_W; a.partials = d_part; a.n = n_paths; a.ld = ld; a.ld_w = ld_w; a.n_basis = K; a.n_state = b->n_state;
    a.f32_cache = (flags & MCX_LSM_F32_CACHE) ? 1 : 0; a.bridge = b->d_bridge;
    a.ex_mode = b->ex_mode; a.ex_bits = b->d_ex_bits; a.ex_ld = b->ex_ld; a.ev_base = 0;
    a.vpoly = b->d_vpoly; a.vcoef = b->d_vcoef;
    if (a.ex_mode && a.ex_ld < n_paths) MCX_FAIL(h, -2, "mcx_lsm_step_batch: exercise replay buffer narrower than the path count");
    int rc = 0;
    for (int j0 = 0; j0 < n_jobs && rc == 0; j0 += chunk) {
        const int nj = n_jobs - j0 < chunk ? n_jobs - j0 : chunk;
        const K3Job* d_jobs_now = (const K3Job*)mcx_upload_call_data(h, jobs.data() + j0, sizeof(K3Job) * (size_t)nj, d_jobs, s);
        if (!d_jobs_now) return -100;
        const dim3 grid(bpj, nj);
        switch (S) {
        case 1: rc = dispatch_k3_batch<1>(K, a, d_jobs_now, grid, s); break;
        case 2: rc = dispatch_k3_batch<2>(K, a, d_jobs_now, grid, s); break;
        case 3: rc = dispatch_k3_batch<3>(K, a, d_jobs_now, grid, s); break;
        case 4: rc = dispatch_k3_batch<4>(K, a, d_jobs_now, grid, s); break;
        case 5: rc = dispatch_k3_batch<5>(K, a, d_jobs_now, grid, s); break;
        case 6: rc = dispatch_k3_batch<6>(K, a, d_jobs_now, grid, s); break;
        case 7: rc = dispatch_k3_batch<7>(K, a, d_jobs_now, grid, s); break;
        case 8: rc = dispatch_k3_batch<8>(K, a, d_jobs_now, grid, s); break;
        default: rc = -1; break;
        }
        if (rc != 0) break;
        double* dst = d_moments ? d_moments + (size_t)j0 * NM : d_out;
        hipLaunchKernelGGL(k3_finish_batch, dim3(nj), dim3(64), 0, s, d_part, NM, bpj, dst, NM);
        if (h_moments) {
            if (hipMemcpyAsync(h_moments + (size_t)j0 * NM, dst, sizeof(double) * (size_t)nj * NM, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = -100; break; }
            if (hipStreamSynchronize(s) != hipSuccess) { rc = -100; break; }
        }
    }
    if (rc == -1) MCX_FAIL(h, -3, "mcx_lsm_step_batch: unsupported (basis=%d, states=%d)", K, S);
    if (rc != 0) MCX_FAIL(h, -100, "mcx_lsm_step_batch: HIP error: %s", hipGetErrorString(hipGetLastError()));
    return 0;
}

extern "C" int mcx_lsm_step_batch(mcx_handle* h, const mcx_book* b, const mcx_lsm_job* h_jobs, int32_t n_jobs, int32_t n_states,
                                  const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                                  double* h_moments, int32_t flags, void* stream)
{
    if (!h_moments) return -1;
    return lsm_step_batch_impl(h, b, h_jobs, n_jobs, n_states, d_paths, n_paths, ld, d_W, ld_w, w_len, h_moments, nullptr, flags, stream);
}

extern "C" int mcx_lsm_step_batch_dev(mcx_handle* h, const mcx_book* b, const mcx_lsm_job* h_jobs, int32_t n_jobs, int32_t n_states,
                                      const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                                      double* d_moments, int32_t flags, void* stream)
{
    if (!d_moments) return -1;
    return lsm_step_batch_impl(h, b, h_jobs, n_jobs, n_states, d_paths, n_paths, ld, d_W, ld_w, w_len, nullptr, d_moments, flags, stream);
}

namespace {
struct K3SolveJob { double shift, scale, x0; int64_t off0, off1; int32_t degenerate, pad; };
// one thread per (product, date) system of a batched step: solve + coefficient scatter on the device; a numerically singular
// system raises the flag (its coefficients are not written: the caller repeats the induction with the host solver)
__global__ __launch_bounds__(64) void k3_solve_batch(const double* __restrict__ moments, const K3SolveJob* __restrict__ jobs, int n_jobs,
                                                     int K, int S, double* __restrict__ coeffs, int32_t* __restrict__ flag)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    const K3SolveJob q = jobs[j];
    K3Solve sv;
    sv.shift = q.shift; sv.scale = q.scale; sv.x0 = q.x0; sv.off0 = q.off0; sv.off1 = q.off1;
    sv.degenerate = q.degenerate; sv.K = K; sv.S = S; sv.date = 0;
    int32_t st = 0;
    k3_solve_body(moments + (int64_t)j * ((2 * K - 1) + S * K), sv, coeffs, nullptr, &st);
    if (st) atomicOr(flag, 1);
}
// finish + solve of a batched step in ONE launch (mcx_lsm_run_batch on one GPU: no all-reduce between the two): thread j sums the
// partial moments of job j over its blocks in the order k3_finish_batch does (bit-identical moments) and solves
#define K3_NM_MAX ((2 * MCX_MAX_BASIS - 1) + MCX_MAX_STATES * MCX_MAX_BASIS)
__global__ __launch_bounds__(64) void k3_finish_solve_batch(const double* __restrict__ partials, int blocks_per_job, const K3SolveJob* __restrict__ jobs,
                                                            int n_jobs, int K, int S, double* __restrict__ coeffs, int32_t* __restrict__ flag)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= n_jobs) return;
    const int nm = (2 * K - 1) + S * K;
    double m[K3_NM_MAX];
    for (int q = 0; q < nm; ++q) {
        double s = 0.0;
        for (int b = 0; b < blocks_per_job; ++b) s += partials[((int64_t)j * blocks_per_job + b) * nm + q];
        m[q] = s;
    }
    const K3SolveJob q = jobs[j];
    K3Solve sv;
    sv.shift = q.shift; sv.scale = q.scale; sv.x0 = q.x0; sv.off0 = q.off0; sv.off1 = q.off1;
    sv.degenerate = q.degenerate; sv.K = K; sv.S = S; sv.date = 0;
    int32_t st = 0;
    k3_solve_body(m, sv, coeffs, nullptr, &st);
    if (st) atomicOr(flag, 1);
}
}  // namespace

extern "C" int mcx_lsm_solve_batch(mcx_handle* h, mcx_book* b, const mcx_lsm_solve_job* h_jobs, int32_t n_jobs, int32_t n_states,
                                   const double* d_moments, int32_t* d_flag, void* stream)
{
    if (!h || !b || !h_jobs || !d_moments || !d_flag) return -1;
    if (n_jobs <= 0) return 0;
    const int K = b->n_basis, S = n_states;
    if (S < 1 || S > MCX_MAX_STATES) MCX_FAIL(h, -2, "mcx_lsm_solve_batch: n_states out of range");
    std::vector<K3SolveJob> jobs((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        const mcx_lsm_solve_job& q = h_jobs[j];
        for (int w = 0; w < 2; ++w)
            if (q.coeff_off[w] >= 0 && q.coeff_off[w] + (int64_t)S * K > b->n_coeffs) MCX_FAIL(h, -2, "mcx_lsm_solve_batch: job %d coefficient offset out of range", j);
        K3SolveJob& o = jobs[j];
        o.shift = q.shift; o.scale = q.scale; o.x0 = q.x0; o.off0 = q.coeff_off[0]; o.off1 = q.coeff_off[1]; o.degenerate = q.degenerate; o.pad = 0;
    }
    hipStream_t s = (hipStream_t)stream;
    K3SolveJob* d_jobs = (K3SolveJob*)mcx_scratch(h, 1, sizeof(K3SolveJob) * (size_t)n_jobs);      // (slot 1: the step's job table is consumed by then, stream order)
    if (!d_jobs) return -100;
    const K3SolveJob* d_jobs_now = (const K3SolveJob*)mcx_upload_call_data(h, jobs.data(), sizeof(K3SolveJob) * (size_t)n_jobs, d_jobs, s);
    if (!d_jobs_now) return -100;
    hipLaunchKernelGGL(k3_solve_batch, dim3((n_jobs + 63) / 64), dim3(64), 0, s, d_moments, d_jobs_now, n_jobs, K, S, b->d_coeffs, d_flag);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

// The whole product-batched backward induction in ONE call: the (step, [all-reduce], solve) pairs of mcx_lsm_step_batch_dev /
// mcx_lsm_solve_batch for every step of the schedule, job tables uploaded once, nothing but launches in the loop (the Python loop
// around the two entry points cost ~240 us of host time per step against ~25 us of kernels: 633 steps for the 5,000-product book).
extern "C" int mcx_lsm_run_batch(mcx_handle* h, mcx_book* b, const mcx_lsm_job* h_jobs, const mcx_lsm_solve_job* h_solve,
                                 const int32_t* h_step_begin, const int32_t* h_step_states, int32_t n_steps,
                                 const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                                 int32_t* h_flag, int32_t flags, void* stream)
{
    if (!h || !b || !h_jobs || !h_solve || !h_step_begin || !h_step_states || !d_paths || !d_W || !h_flag) return -1;
    *h_flag = 0;
    if (n_steps <= 0) return 0;
    const int K = b->n_basis;
    if (K < 1 || K > 6) MCX_FAIL(h, -3, "mcx_lsm_run_batch: unsupported basis size %d", K);
    if (ld < n_paths || ld_w < n_paths) MCX_FAIL(h, -2, "mcx_lsm_run_batch: leading dimension < n_paths");
    if (h_step_begin[0] != 0) MCX_FAIL(h, -2, "mcx_lsm_run_batch: step table must start at job 0");
    const int64_t n_jobs = h_step_begin[n_steps];
    if (n_jobs <= 0) return 0;
    const int max_jobs = 32768;
    int nm_max = 0, widest = 0;
    for (int t = 0; t < n_steps; ++t) {
        const int S = h_step_states[t], nj = h_step_begin[t + 1] - h_step_begin[t];
        if (nj < 0) MCX_FAIL(h, -2, "mcx_lsm_run_batch: step table not ascending at step %d", t);
        if (S < 1 || S > MCX_MAX_STATES) MCX_FAIL(h, -2, "mcx_lsm_run_batch: step %d: n_states out of range", t);
        const int NM = (2 * K - 1) + S * K, chunk = nj < max_jobs ? nj : max_jobs;
        if (NM > nm_max) nm_max = NM;
        if (chunk > widest) widest = chunk;
    }
    auto flat = [&](int id) { DevAtom o; const mcx_atom& q = b->h_atoms[id]; o.t_idx = q.t_idx; o.col = q.col; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1; return o; };
    std::vector<K3Job> jobs((size_t)n_jobs);
    std::vector<K3SolveJob> solves((size_t)n_jobs);
    for (int t = 0; t < n_steps; ++t) {
        const int S = h_step_states[t];
        for (int j = h_step_begin[t]; j < h_step_begin[t + 1]; ++j) {
            const mcx_lsm_job& q = h_jobs[j];
            if (q.product < 0 || q.product >= b->n_products) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d product out of range", j);
            const DevProduct& pr = b->h_products[q.product];
            if (pr.n_states != S) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d has %d states, its step %d", j, pr.n_states, S);
            if (q.roll_begin < 0 || q.roll_end < q.roll_begin || q.roll_end > pr.cf_end - pr.cf_begin) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d roll window", j);
            if (q.num_atom < 0 || q.num_atom >= b->n_atoms || q.x_atom < 0 || q.x_atom >= b->n_atoms) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d atoms", j);
            if (q.w_offset < 0 || q.w_offset + (int64_t)S * ld_w > w_len) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d cache block outside d_W", j);
            K3Job& o = jobs[j];
            o.ev_off = pr.cf_begin; o.roll_begin = q.roll_begin; o.roll_end = q.roll_end; o.pad = 0; o.w_off = q.w_offset;
            o.shift = q.shift; o.scale = q.scale; o.num = flat(q.num_atom); o.x = flat(q.x_atom);
            const mcx_lsm_solve_job& v = h_solve[j];
            for (int w = 0; w < 2; ++w)
                if (v.coeff_off[w] >= 0 && v.coeff_off[w] + (int64_t)S * K > b->n_coeffs) MCX_FAIL(h, -2, "mcx_lsm_run_batch: job %d coefficient offset out of range", j);
            K3SolveJob& u = solves[j];
            u.shift = v.shift; u.scale = v.scale; u.x0 = v.x0; u.off0 = v.coeff_off[0]; u.off1 = v.coeff_off[1]; u.degenerate = v.degenerate; u.pad = 0;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const int bpj = n_paths > 0 ? mcx_grid_for(n_paths, MCX_BLOCK, 64) : 1;
    const size_t job_bytes = sizeof(K3Job) * (size_t)n_jobs, solve_bytes = sizeof(K3SolveJob) * (size_t)n_jobs;
    unsigned char* d_tab = (unsigned char*)mcx_scratch(h, 1, job_bytes + solve_bytes);
    double* d_part = (double*)mcx_scratch(h, 2, sizeof(double) * (size_t)widest * bpj * nm_max);
    double* d_mom = (double*)mcx_scratch(h, 3, sizeof(double) * ((size_t)widest * nm_max + 1));
    if (!d_tab || !d_part || !d_mom) return -100;
    int32_t* d_flag = (int32_t*)(d_mom + (size_t)widest * nm_max);
    // both tables in one device buffer, copied before the first launch (the host vectors die with this call)
    MCX_HIP(h, hipMemcpyAsync(d_tab, jobs.data(), job_bytes, hipMemcpyHostToDevice, s));
    MCX_HIP(h, hipMemcpyAsync(d_tab + job_bytes, solves.data(), solve_bytes, hipMemcpyHostToDevice, s));
    MCX_HIP(h, hipMemsetAsync(d_flag, 0, sizeof(int32_t), s));
    const K3Job* d_jobs = (const K3Job*)d_tab;
    const K3SolveJob* d_solves = (const K3SolveJob*)(d_tab + job_bytes);
    K3Args a;
    memset(&a, 0, sizeof(a));
    a.terms = b->d_terms; a.events = b->d_events; a.atoms = b->d_atoms; a.coeffs = b->d_coeffs; a.paths = d_paths;
    a.W = d_W; a.partials = d_part; a.n = n_paths; a.ld = ld; a.ld_w = ld_w; a.n_basis = K; a.n_state = b->n_state;
    a.f32_cache = (flags & MCX_LSM_F32_CACHE) ? 1 : 0; a.bridge = b->d_bridge;
    a.ex_mode = b->ex_mode; a.ex_bits = b->d_ex_bits; a.ex_ld = b->ex_ld; a.ev_base = 0;
    a.vpoly = b->d_vpoly; a.vcoef = b->d_vcoef;
    if (a.ex_mode && a.ex_ld < n_paths) MCX_FAIL(h, -2, "mcx_lsm_run_batch: exercise replay buffer narrower than the path count");
    const bool multi = h->comm && h->comm_ranks > 1;
    int rc = 0;
    for (int t = 0; t < n_steps && rc == 0; ++t) {
        const int S = h_step_states[t], NM = (2 * K - 1) + S * K;
        for (int j0 = h_step_begin[t]; j0 < h_step_begin[t + 1] && rc == 0; j0 += max_jobs) {
            const int nj = h_step_begin[t + 1] - j0 < max_jobs ? h_step_begin[t + 1] - j0 : max_jobs;
            const dim3 grid(bpj, nj);
            if (n_paths <= 0) {                            // a rank without paths still takes part in the all-reduce
                if (hipMemsetAsync(d_mom, 0, sizeof(double) * (size_t)nj * NM, s) != hipSuccess) { rc = -100; break; }
            } else switch (S) {
            case 1: rc = dispatch_k3_batch<1>(K, a, d_jobs + j0, grid, s); break;
            case 2: rc = dispatch_k3_batch<2>(K, a, d_jobs + j0, grid, s); break;
            case 3: rc = dispatch_k3_batch<3>(K, a, d_jobs + j0, grid, s); break;
            case 4: rc = dispatch_k3_batch<4>(K, a, d_jobs + j0, grid, s); break;
            case 5: rc = dispatch_k3_batch<5>(K, a, d_jobs + j0, grid, s); break;
            case 6: rc = dispatch_k3_batch<6>(K, a, d_jobs + j0, grid, s); break;
            case 7: rc = dispatch_k3_batch<7>(K, a, d_jobs + j0, grid, s); break;
            case 8: rc = dispatch_k3_batch<8>(K, a, d_jobs + j0, grid, s); break;
            default: rc = -1; break;
            }
            if (rc != 0) break;
            if (!multi && n_paths > 0) {
                hipLaunchKernelGGL(k3_finish_solve_batch, dim3((nj + 63) / 64), dim3(64), 0, s, d_part, bpj, d_solves + j0, nj, K, S, b->d_coeffs, d_flag);
                continue;
            }
            if (n_paths > 0) hipLaunchKernelGGL(k3_finish_batch, dim3(nj), dim3(64), 0, s, d_part, NM, bpj, d_mom, NM);
            if (multi) { rc = mcx_allreduce_f64(h, d_mom, (int64_t)nj * NM, stream); if (rc != 0) break; }    // stream-ordered
            hipLaunchKernelGGL(k3_solve_batch, dim3((nj + 63) / 64), dim3(64), 0, s, d_mom, d_solves + j0, nj, K, S, b->d_coeffs, d_flag);
        }
    }
    if (rc == -1) MCX_FAIL(h, -3, "mcx_lsm_run_batch: unsupported (basis=%d)", K);
    if (rc == 0 && hipGetLastError() != hipSuccess) rc = -100;
    if (rc == 0 && hipMemcpyAsync(h_flag, d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) rc = -100;
    if (hipStreamSynchronize(s) != hipSuccess && rc == 0) rc = -100;
    if (rc != 0) { if (rc == -100) h->err = "mcx_lsm_run_batch: HIP error"; return rc; }
    return 0;
}

extern "C" int mcx_book_get_coeffs(mcx_handle* h, const mcx_book* b, int64_t offset, int64_t count, double* h_out, void* stream)
{
    if (!h || !b || !h_out) return -1;
    if (offset < 0 || count < 0 || offset + count > b->n_coeffs) MCX_FAIL(h, -2, "mcx_book_get_coeffs: range out of bounds");
    if (count == 0) return 0;
    // through the handle's pinned buffer, chunk by chunk: a device-to-pageable copy of the 13 MB coefficient array of a 5,000-product
    // book ran at 0.7 GB/s (18 ms)
    hipStream_t s = (hipStream_t)stream;
    const int64_t chunk = (int64_t)(h->pinned_bytes / sizeof(double));
    for (int64_t q = 0; q < count; q += chunk) {
        const int64_t nq = count - q < chunk ? count - q : chunk;
        MCX_HIP(h, hipMemcpyAsync(h->h_pinned, b->d_coeffs + offset + q, sizeof(double) * (size_t)nq, hipMemcpyDeviceToHost, s));
        MCX_HIP(h, hipStreamSynchronize(s));
        memcpy(h_out + q, h->h_pinned, sizeof(double) * (size_t)nq);
    }
    return 0;
}

extern "C" int mcx_book_set_coeffs_batch(mcx_handle* h, mcx_book* b, const int64_t* h_offsets, int32_t n, int32_t len,
                                         const double* h_values, void* stream)
{
    if (!h || !b || !h_offsets || !h_values) return -1;
    if (n <= 0 || len <= 0) return 0;
    for (int j = 0; j < n; ++j)
        if (h_offsets[j] < 0 || h_offsets[j] + len > b->n_coeffs) MCX_FAIL(h, -2, "mcx_book_set_coeffs_batch: range %d out of bounds", j);
    hipStream_t s = (hipStream_t)stream;
    int64_t* d_off = (int64_t*)mcx_scratch(h, 1, sizeof(int64_t) * (size_t)n);
    double* d_val = (double*)mcx_scratch(h, 2, sizeof(double) * (size_t)n * len);
    if (!d_off || !d_val) return -100;
    MCX_HIP(h, hipMemcpyAsync(d_off, h_offsets, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, s));
    MCX_HIP(h, hipMemcpyAsync(d_val, h_values, sizeof(double) * (size_t)n * len, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k3_scatter_coeffs, dim3(n), dim3(64), 0, s, d_off, d_val, len, b->d_coeffs);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}
