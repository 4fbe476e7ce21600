// kt_book.hip — forward-mode (tangent) pass through the WHOLE exposure path: paths, LSM regression, book evaluation, CVA.
//
// The reference obtains d CVA / d theta by taping pre-simulation + lstsq + main simulation and calling torch.autograd.grad
// (controller/controller.py:609-627, regression coefficients carry their graph: :370-383).  Here the derivative travels
// forward in dual numbers, MCX_TANGENT_NP parameters per pass:
//   kt_paths : K1 with dual state           -> paths [T][D][N] and d paths / d theta [NP][T][D][N]
//   kt_lsm   : normal equations of one (product, regression date) with dual moments (the host differentiates the solve:
//              G c = r  =>  dc = G^-1 (dr - dG c))
//   kt_eval  : the book's cashflow / polynomial-exposure events with dual atoms and dual coefficients
//   kt_cva   : sum_m relu(thr(E_m)) S(0,t_m) (1 - S(t_m,t_m+1)) (1-R) per path with tangents (cva_metric.py:62-100)
// The derivatives of every host-computed descriptor number (model parameters per slot, psi(t) tables, initial state, the
// closed-form coefficients of each atom) arrive as arrays next to the primal descriptors (mcx/aad.py builds them).
//   kt_lsm_step : the same for products with exercise rights (Bermudan / American / FlexiCall): the cashflow cache rolled back
//              along the FROZEN exercise policy in dual numbers, moments per hypothetical state
// Scope: EULER scheme; Black-Scholes / Vasicek / CIR++ (stochastic and deterministic) slots; cashflow, plain option and exercise
// events, polynomial (also state-indexed) and analytic Black-Scholes exposures; thresholds and MPoR collateral in the metric
// kernels.  Anything else keeps the common-random-number bump path.
#include "mcx_dual.h"

namespace {

constexpr int NP = MCX_TANGENT_NP;
typedef Dual<NP> DN;

__device__ __forceinline__ DN ld_dual(double v, const double* __restrict__ d)      // wave-uniform derivative row -> scalar loads
{
    DN r;
    r.v = v;
#pragma unroll
    for (int q = 0; q < NP; ++q) r.d[q] = ldk(d + q);
    return r;
}

// ---- paths -------------------------------------------------------------------------------------------------------------
struct KTPArgs {
    K1Args k1;
    const double* __restrict__ dslot;   // [n_slots][MCX_SLOT_NPARAM][NP]
    const double* __restrict__ dinit;   // [n_state][NP]
    const double* __restrict__ daux;    // [n_steps][n_slots][MCX_AUX][NP]
    double* __restrict__ dpaths;        // [NP][T][D][ld]
    int64_t pstride;                    // T * D * ld
    int32_t n_slots, pad;
};

template <int NSLOT>
__device__ __forceinline__ void ktp_store(const KTPArgs& a, int t, int64_t i, const DN (&reg)[2 * NSLOT])
{
    const K1Args& k = a.k1;
    const int D = k.n_state;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int c = k.slots[s].state_off;
        const int nd = k.slots[s].kind == MCX_MODEL_BS ? 1 : 2;
        for (int e = 0; e < nd; ++e) {
            const int64_t off = ((int64_t)t * D + c + e) * k.ld + i;
            k.paths[off] = reg[2 * s + e].v;
#pragma unroll
            for (int q = 0; q < NP; ++q) a.dpaths[q * a.pstride + off] = reg[2 * s + e].d[q];
        }
    }
}

template <int NSLOT, int NZ, bool INJECT>
__global__ __launch_bounds__(MCX_BLOCK) void kt_paths(const KTPArgs a)
{
    const K1Args& k = a.k1;
    __shared__ double bm_lds[INJECT ? 2 : MCX_BM_LDS_DOUBLES];
    const double* tab = nullptr;
    if (!INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= k.n) return;
    DN reg[2 * NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const int c = k.slots[s].state_off;
        const bool bs = k.slots[s].kind == MCX_MODEL_BS;
        reg[2 * s] = ld_dual(k.init_state[c], a.dinit + (int64_t)c * NP);
        reg[2 * s + 1] = bs ? dconst<NP>(0.0) : ld_dual(k.init_state[c + 1], a.dinit + (int64_t)(c + 1) * NP);
    }
    for (int t = 0; t < k.n_initial_store; ++t) ktp_store<NSLOT>(a, t, i, reg);
    const uint64_t path = k.path_offset + (uint64_t)i;
#pragma unroll 1
    for (int step = 0; step < k.n_steps; ++step) {
        const mcx_step sp = ldk_struct(&k.steps[step]);
        double z[NZ], zc[NZ];
        if (INJECT) {
#pragma unroll
            for (int j = 0; j < NZ; ++j) z[j] = k.inject_z[((int64_t)step * NZ + j) * k.ld + i];
        } else {
#pragma unroll
            for (int q = 0; q < (NZ + 1) / 2; ++q) {
                double ua, z0, z1;
                draw_pair<true>(k.seed, path, (uint32_t)step, (uint32_t)q, ua, z0, z1, tab);
                z[2 * q] = z0;
                if (2 * q + 1 < NZ) z[2 * q + 1] = z1;
            }
        }
        const double* __restrict__ L = k.chol + (int64_t)sp.chol_idx * NZ * NZ;
#pragma unroll
        for (int r = 0; r < NZ; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c <= r; ++c) acc += ldk(L + r * NZ + c) * z[c];
            zc[r] = acc;
        }
        const double dt = sp.dt, sq = sp.sqrt_dt;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const double* __restrict__ p = k.slots[s].p;
            const double* __restrict__ dp = a.dslot + (int64_t)s * MCX_SLOT_NPARAM * NP;
            const double* __restrict__ ax = k.aux + ((int64_t)step * NSLOT + s) * MCX_AUX;
            const double* __restrict__ dax = a.daux + ((int64_t)step * NSLOT + s) * MCX_AUX * NP;
            const double zs = zc[s < NZ ? s : 0];
            DN& s0 = reg[2 * s];
            DN& s1 = reg[2 * s + 1];
            switch (k.slots[s].kind) {
            case MCX_MODEL_BS: {                                          // black_scholes.py:79-85
                const DN sigma = ld_dual(p[1], dp + 1 * NP), rate = ld_dual(p[2], dp + 2 * NP);
                s0 = s0 + (rate * s0 * dt + sigma * s0 * (sq * zs));
                break;
            }
            case MCX_MODEL_VASICEK: {                                     // vasicek.py:88-112
                const DN sigma = ld_dual(p[1], dp + 1 * NP), mean = ld_dual(p[2], dp + 2 * NP), speed = ld_dual(p[3], dp + 3 * NP);
                const DN r = s0;
                s1 = s1 + r * dt;
                s0 = r + speed * (mean - r) * dt + sigma * (sq * zs);
                break;
            }
            case MCX_MODEL_CIRPP: {                                       // cirpp.py:188-198
                const DN kappa = ld_dual(p[0], dp + 0 * NP), theta = ld_dual(p[1], dp + 1 * NP), sigma = ld_dual(p[2], dp + 2 * NP);
                const DN psi = ld_dual(ldk(ax + 0), dax + 0 * NP);
                const DN y = s0;
                const DN sy = dsqrt(dclamp_min(y, 0.0));
                const DN yn = y + kappa * (theta - y) * dt + sigma * sy * (sq * zs);
                s1 = s1 + (y + psi) * dt;
                s0 = dclamp_min(yn, 1e-12);
                break;
            }
            case MCX_MODEL_CIRPP_DET: {                                   // cirpp.py:155-172
                s1 = s1 + ld_dual(ldk(ax + 0), dax + 0 * NP) * dt;
                s0 = ld_dual(ldk(ax + 1), dax + 1 * NP);
                break;
            }
            default: break;
            }
        }
        const int st = sp.store_idx;
        if (st >= 0) ktp_store<NSLOT>(a, st, i, reg);
    }
}

// ---- dual atoms ---------------------------------------------------------------------------------------------------------
struct KTBook {
    const DevTerm* __restrict__ terms;
    const DevEvent* __restrict__ events;
    const DevAtom* __restrict__ atoms;
    const double* __restrict__ datoms;   // [n_atoms][5][NP]: d(a, d, b, c0, c1) / d theta
    const double* __restrict__ paths;
    const double* __restrict__ dpaths;
    int64_t n, ld, pstride;
    int32_t n_state, n_basis;
};

__device__ __forceinline__ DN kt_atom(const KTBook& b, const DevAtom& a, int atom_id, int64_t i)
{
    const double* __restrict__ da = b.datoms + (int64_t)atom_id * 5 * NP;
    DN x = dconst<NP>(0.0);
    if (a.col >= 0) {
        const int64_t off = ((int64_t)a.t_idx * b.n_state + a.col) * b.ld + i;
        x.v = b.paths[off];
#pragma unroll
        for (int q = 0; q < NP; ++q) x.d[q] = b.dpaths[q * b.pstride + off];
    }
    DN v = ld_dual(a.a, da) + ld_dual(a.d, da + NP) * x;
    if (a.b != 0.0) v = v + ld_dual(a.b, da + 2 * NP) * dexp(ld_dual(a.c0, da + 3 * NP) + ld_dual(a.c1, da + 4 * NP) * x);
    return v;
}

// DevEvent carries COPIES of its numeraire / explanatory atoms; the tangent kernels need their ids to find the derivative rows
struct KTEventIds { int32_t num, x; };

// normalised dual cashflow of one stateless cash event (CASHFLOW or plain OPTION)
__device__ __forceinline__ DN kt_cash_event(const KTBook& b, const DevEvent& e, const KTEventIds& id, const int32_t* __restrict__ term_atom, int64_t i)
{
    const DN num = kt_atom(b, e.num, id.num, i);
    DN val = dconst<NP>(0.0), own = dconst<NP>(0.0);
    for (int j = e.term_begin; j < e.term_end; ++j) {
        const DevTerm tm = ldk_struct(&b.terms[j]);
        const DN v = kt_atom(b, tm.atom, ldk(term_atom + j), i) * tm.w;
        if (tm.den < 0) val = val + v;
        else own = own + v / kt_atom(b, ldk_struct(&b.atoms[tm.den]), tm.den, i);      // the unequal-tenor swap quirk (mcx_term.den)
    }
    if (e.kind == MCX_EV_CASHFLOW) return val / num + own;
    const DN x = (val - e.strike) * e.sign;                         // torch.maximum(x, 0): gradient 1 for x > 0, 1/2 at the tie
    const double w = x.v > 0.0 ? 1.0 : (x.v == 0.0 ? 0.5 : 0.0);
    DN pay;
    pay.v = fmax(x.v, 0.0);
#pragma unroll
    for (int q = 0; q < NP; ++q) pay.d[q] = w * x.d[q];
    return pay / num;
}

// ---- exercise events in dual numbers -------------------------------------------------------------------------------------------
// The reference's tape puts NO gradient through the boolean `should_exercise` (bermudan_option.py:122-128; flexicall.py:118-133):
// the decision is taken from the primal values — immediate value against the regression continuation value of the path's state —
// and the tangent flows through the branch that was taken only (the immediate value, where the path exercises).
// State-independent pieces of one MCX_EV_EXERCISE event for path i: the dual immediate value already divided by the numeraire and
// the primal explanatory variable of the continuation polynomial.
struct KTExercise { DN pay; double imm, x; };
__device__ __forceinline__ KTExercise kt_exercise_value(const KTBook& b, const DevEvent& e, const KTEventIds& id, const int32_t* __restrict__ term_atom,
                                                         int64_t i)
{
    KTExercise r;
    const DN num = kt_atom(b, e.num, id.num, i);
    DN val = dconst<NP>(0.0);
    for (int j = e.term_begin; j < e.term_end; ++j) {
        const DevTerm tm = ldk_struct(&b.terms[j]);
        val = val + kt_atom(b, tm.atom, ldk(term_atom + j), i) * tm.w;
    }
    const DN xs = (val - e.strike) * e.sign;                         // torch.maximum(x, 0): gradient 1 for x > 0, 1/2 at the tie
    const double w = xs.v > 0.0 ? 1.0 : (xs.v == 0.0 ? 0.5 : 0.0);
    DN imm;
    imm.v = fmax(xs.v, 0.0);
#pragma unroll
    for (int q = 0; q < NP; ++q) imm.d[q] = w * xs.d[q];
    r.pay = imm / num;
    r.imm = imm.v;
    r.x = e.coeff_off >= 0 ? kt_atom(b, e.x, id.x, i).v : 0.0;
    return r;
}
__device__ __forceinline__ double kt_poly(const double* __restrict__ c, int K, double x)
{
    double v = 0.0, xp = 1.0;
    for (int k = 0; k < K; ++k) { v = fma(c[k], xp, v); xp *= x; }
    return v;
}
// decision of a path in state s (the PRIMAL coefficients of the base run: the same decisions as its primal pass); s is decremented
__device__ __forceinline__ bool kt_exercises(const DevEvent& e, const double* __restrict__ coeffs, int K, const KTExercise& ev, int& s)
{
    double cont = 0.0, cont_ex = 0.0;
    if (e.coeff_off >= 0) {
        cont = kt_poly(coeffs + e.coeff_off + s * K, K, ev.x);
        if (e.aux[0] == 1.0 && s > 0) cont_ex = kt_poly(coeffs + e.coeff_off + (s - 1) * K, K, ev.x);      // flexicall.py:118-133
    }
    const bool ex = (ev.imm + cont_ex > cont) && (s > 0);
    if (ex) s -= 1;
    return ex;
}

// ---- LSM step of an exercise product with tangents: the cashflow cache is rolled one window back along the frozen policy ----------
// (controller.py:316-383; K3's k3_roll in dual numbers).  W [S][ld_w] and dW [NP][S][ld_w] hold, per hypothetical state, the
// discounted cashflows after the previous regression date and their tangents; moments [1+NP][(2K-1) + S K].
struct KTSArgs {
    KTBook b;
    const KTEventIds* __restrict__ ev_ids;
    const int32_t* __restrict__ term_atom;
    const double* __restrict__ coeffs;         // primal coefficients of the base run (decisions)
    double* __restrict__ W;
    double* __restrict__ dW;
    DevAtom num, x;
    double shift, scale;
    double* __restrict__ partials;             // [gridDim.x][1+NP][NM]
    int64_t ld_w, w_stride;                    // w_stride = S * ld_w (between tangents)
    int32_t roll_begin, roll_end, num_id, x_id;
};

template <int K, int S>
__global__ __launch_bounds__(MCX_BLOCK) void kt_lsm_step(const KTSArgs a)
{
    constexpr int NM = (2 * K - 1) + S * K;
    double acc[1 + NP][NM];
#pragma unroll
    for (int q = 0; q <= NP; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[q][m] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.b.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        DN w[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            w[s].v = a.W[(int64_t)s * a.ld_w + i];
#pragma unroll
            for (int q = 0; q < NP; ++q) w[s].d[q] = a.dW[q * a.w_stride + (int64_t)s * a.ld_w + i];
        }
        if (a.roll_end > a.roll_begin) {
            int st[S];
            DN sv[S];
#pragma unroll
            for (int s0 = 0; s0 < S; ++s0) { st[s0] = s0; sv[s0] = dconst<NP>(0.0); }
            for (int q = a.roll_begin; q < a.roll_end; ++q) {                   // controller.py:333-341
                const DevEvent e = ldk_struct(&a.b.events[q]);
                const KTEventIds id = ldk_struct(&a.ev_ids[q]);
                if (e.kind == MCX_EV_EXERCISE) {
                    const KTExercise ev = kt_exercise_value(a.b, e, id, a.term_atom, i);      // once per path and date, not per state
#pragma unroll
                    for (int s0 = 0; s0 < S; ++s0)
                        if (kt_exercises(e, a.coeffs, K, ev, st[s0])) sv[s0] = sv[s0] + ev.pay;
                } else {
                    const DN v = kt_cash_event(a.b, e, id, a.term_atom, i);
#pragma unroll
                    for (int s0 = 0; s0 < S; ++s0) sv[s0] = sv[s0] + v;
                }
            }
            DN wn[S];
#pragma unroll
            for (int s0 = 0; s0 < S; ++s0) {
                DN tail = w[0];
#pragma unroll
                for (int q = 1; q < S; ++q) if (st[s0] == q) tail = w[q];        // lookup_state_values (product.py:150-155)
                wn[s0] = sv[s0] + tail;
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                w[s] = wn[s];
                a.W[(int64_t)s * a.ld_w + i] = wn[s].v;
#pragma unroll
                for (int q = 0; q < NP; ++q) a.dW[q * a.w_stride + (int64_t)s * a.ld_w + i] = wn[s].d[q];
            }
        }
        const DN num = kt_atom(a.b, a.num, a.num_id, i);                                   // :368
        const DN z = (kt_atom(a.b, a.x, a.x_id, i) - a.shift) * a.scale;
        DN zp = dconst<NP>(1.0);
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            acc[0][k] += zp.v;
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[1 + q][k] += zp.d[q];
            if (k < K) {
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const DN zy = zp * (num * w[s]);
                    acc[0][(2 * K - 1) + s * K + k] += zy.v;
#pragma unroll
                    for (int q = 0; q < NP; ++q) acc[1 + q][(2 * K - 1) + s * K + k] += zy.d[q];
                }
            }
            zp = zp * z;
        }
    }
    __shared__ double lds[4];
#pragma unroll
    for (int q = 0; q <= NP; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double r = block_sum(acc[q][m], lds);
            if (threadIdx.x == 0) a.partials[((int64_t)blockIdx.x * (1 + NP) + q) * NM + m] = r;
        }
}

// ---- LSM moments with tangents ------------------------------------------------------------------------------------------
struct KTLArgs {
    KTBook b;
    const KTEventIds* __restrict__ ev_ids;     // [n_events]
    const int32_t* __restrict__ term_atom;     // [n_terms] atom id of every term
    DevAtom num, x;
    double shift, scale;
    double* __restrict__ partials;             // [gridDim.x][(1+NP)][NM]
    int32_t ev_first, ev_end, num_id, x_id;
};

template <int K>
__global__ __launch_bounds__(MCX_BLOCK) void kt_lsm(const KTLArgs a)
{
    constexpr int NM = (2 * K - 1) + K;
    double acc[1 + NP][NM];
#pragma unroll
    for (int q = 0; q <= NP; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[q][m] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.b.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        DN total = dconst<NP>(0.0);
        for (int q = a.ev_first; q < a.ev_end; ++q)                                       // controller.py:333-352 without the cache
            total = total + kt_cash_event(a.b, ldk_struct(&a.b.events[q]), ldk_struct(&a.ev_ids[q]), a.term_atom, i);
        const DN y = kt_atom(a.b, a.num, a.num_id, i) * total;                             // :368
        const DN z = (kt_atom(a.b, a.x, a.x_id, i) - a.shift) * a.scale;
        DN zp = dconst<NP>(1.0);
#pragma unroll
        for (int k = 0; k < 2 * K - 1; ++k) {
            acc[0][k] += zp.v;
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[1 + q][k] += zp.d[q];
            if (k < K) {
                const DN zy = zp * y;
                acc[0][(2 * K - 1) + k] += zy.v;
#pragma unroll
                for (int q = 0; q < NP; ++q) acc[1 + q][(2 * K - 1) + k] += zy.d[q];
            }
            zp = zp * z;
        }
    }
    __shared__ double lds[4];
#pragma unroll
    for (int q = 0; q <= NP; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const double r = block_sum(acc[q][m], lds);
            if (threadIdx.x == 0) a.partials[((int64_t)blockIdx.x * (1 + NP) + q) * NM + m] = r;
        }
}

__global__ void kt_sum_partials(const double* __restrict__ partials, int count, int n_blocks, double* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    double s = 0.0;
    for (int b = 0; b < n_blocks; ++b) s += partials[(int64_t)b * count + j];
    out[j] = s;
}

// ---- book evaluation with tangents ----------------------------------------------------------------------------------------
struct KTEArgs {
    KTBook b;
    const KTEventIds* __restrict__ ev_ids;
    const int32_t* __restrict__ term_atom;
    const DevProduct* __restrict__ products;
    const double* __restrict__ coeffs;         // [n_coeffs] regression coefficients of the tangent pass
    const double* __restrict__ dcoeffs;        // [n_coeffs][NP]
    double* __restrict__ cfs;                  // [1+NP][n_ns][ld]   (zero-initialised)
    double* __restrict__ expo;                 // [1+NP][n_ns][n_rows][ld] (zero-initialised)
    const int32_t* __restrict__ ev_param;      // [n_events][2] tangent slot (0..NP-1 or -1) of an EXPO_BS event's sigma / rate; nullable
    int32_t n_products, n_ns, n_rows, pad;
};

__global__ __launch_bounds__(MCX_BLOCK) void kt_eval(const KTEArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= a.b.n) return;
    const int K = a.b.n_basis;
    const int64_t cf_stride = (int64_t)a.n_ns * a.b.ld, ex_stride = (int64_t)a.n_ns * a.n_rows * a.b.ld;
    for (int p = 0; p < a.n_products; ++p) {
        const DevProduct pr = ldk_struct(&a.products[p]);
        DN acc = dconst<NP>(0.0);
        int state = pr.init_state;                                                        // rights left (exercise products)
        for (int q = pr.ev_begin; q < pr.ev_end; ++q) {
            const DevEvent e = ldk_struct(&a.b.events[q]);
            const KTEventIds id = ldk_struct(&a.ev_ids[q]);
            if (e.kind <= MCX_EV_OPTION) {
                acc = acc + kt_cash_event(a.b, e, id, a.term_atom, i);
            } else if (e.kind == MCX_EV_EXERCISE) {
                // decision from the primal values, tangent through the taken branch only (bermudan_option.py:93-131)
                const KTExercise ev = kt_exercise_value(a.b, e, id, a.term_atom, i);
                if (kt_exercises(e, a.coeffs, K, ev, state)) acc = acc + ev.pay;
            } else {                                                                      // exposures (controller.py:430-447)
                DN v = dconst<NP>(0.0);
                if (e.kind == MCX_EV_EXPO_BS) {
                    // analytic Black-Scholes exposure (european_option.py:123-145) in dual numbers: sigma and rate are model
                    // parameters (their tangent slots come from the host), the spot is the path state
                    if (e.aux[2] > 0.0) {
                        DN sig = dconst<NP>(e.aux[0]), rate = dconst<NP>(e.aux[1]);
                        const int s_sig = a.ev_param ? ldk(a.ev_param + 2 * q) : -1, s_rate = a.ev_param ? ldk(a.ev_param + 2 * q + 1) : -1;
#pragma unroll
                        for (int r = 0; r < NP; ++r) { sig.d[r] = (r == s_sig) ? 1.0 : 0.0; rate.d[r] = (r == s_rate) ? 1.0 : 0.0; }
                        const double tau = e.aux[2], sq = sqrt(tau);
                        const DN spot = kt_atom(a.b, e.x, id.x, i);
                        const DN d1 = (dlog(spot * (1.0 / e.strike)) + (rate + sig * sig * 0.5) * tau) / (sig * sq);
                        const DN d2 = d1 - sig * sq;
                        const DN df = dexp(rate * (-tau));
                        auto ncdf = [](const DN& x) {                      // Phi(x), d Phi = phi(x) dx
                            DN r;
                            r.v = 0.5 * (1.0 + erf(x.v * 0.70710678118654752440));
                            const double pdf = 0.39894228040143267794 * exp(-0.5 * x.v * x.v);
#pragma unroll
                            for (int q2 = 0; q2 < NP; ++q2) r.d[q2] = pdf * x.d[q2];
                            return r;
                        };
                        const DN price = e.sign > 0.0 ? spot * ncdf(d1) - df * ncdf(d2) * e.strike
                                                      : df * ncdf(d2 * -1.0) * e.strike - spot * ncdf(d1 * -1.0);
                        v = price / kt_atom(a.b, e.num, id.num, i);
                    }
                } else if (e.coeff_off >= 0) {
                    const DN x = kt_atom(a.b, e.x, id.x, i);
                    DN xp = dconst<NP>(1.0);
                    // the coefficient row of the path's exercise state (product.py:150-184); stateless products: row 0
                    const int row0 = e.coeff_off + (pr.n_states > 1 ? state * K : 0);
                    for (int k = 0; k < K; ++k) {
                        DN ck;
                        ck.v = a.coeffs[row0 + k];
#pragma unroll
                        for (int r = 0; r < NP; ++r) ck.d[r] = a.dcoeffs[(int64_t)(row0 + k) * NP + r];
                        v = v + ck * xp;
                        xp = xp * x;
                    }
                    v = v / kt_atom(a.b, e.num, id.num, i);
                }
                const int64_t off = ((int64_t)pr.netting_set * a.n_rows + e.row) * a.b.ld + i;
                a.expo[off] += v.v;
#pragma unroll
                for (int r = 0; r < NP; ++r) a.expo[(1 + r) * ex_stride + off] += v.d[r];
            }
        }
        const int64_t off = (int64_t)pr.netting_set * a.b.ld + i;
        a.cfs[off] += acc.v;
#pragma unroll
        for (int r = 0; r < NP; ++r) a.cfs[(1 + r) * cf_stride + off] += acc.d[r];
    }
}

// ---- CVA with tangents ------------------------------------------------------------------------------------------------------
// unsecured exposure of one netting set at metric date m with tangents (netting_set.py:48-72, 156-184; mcx_unsecured_desc):
//   not collateralised: thr(E[row]);   collateralised: E[row] - thr(E[delayed]) (delayed < 0: no collateral yet)
__device__ __forceinline__ DN kt_thr(const DN& e, double h)
{
    if (h == 0.0) return e;
    DN r = e;
    if (e.v > h) r.v = e.v - h;
    else if (e.v < -h) r.v = e.v + h;
    else r = dconst<NP>(0.0);
    return r;
}
__device__ __forceinline__ DN kt_load_expo(const double* __restrict__ expo, int64_t ex_stride, int64_t off)
{
    DN e;
    e.v = expo[off];
#pragma unroll
    for (int q = 0; q < NP; ++q) e.d[q] = expo[(1 + q) * ex_stride + off];
    return e;
}
__device__ __forceinline__ DN kt_unsecured(const double* __restrict__ expo, int64_t ex_stride, const int32_t* __restrict__ rows,
                                           const int32_t* __restrict__ delayed, int collateralized, double h, int64_t ld, int m, int64_t i)
{
    const DN e = kt_load_expo(expo, ex_stride, (int64_t)ldk(rows + m) * ld + i);
    if (!collateralized) return kt_thr(e, h);
    if (!delayed) return e;
    const int dm = ldk(delayed + m);
    if (dm < 0) return e;
    return e - kt_thr(kt_load_expo(expo, ex_stride, (int64_t)dm * ld + i), h);
}

struct KTCArgs {
    KTBook b;
    const double* __restrict__ expo;           // [1+NP][n_rows][ld] of ONE netting set (stride ex_stride between tangents)
    const int32_t* __restrict__ rows;          // [n_dates] exposure row of every metric date
    const int32_t* __restrict__ delayed;       // [n_dates] delayed (t - MPoR) row or -1; nullptr when not collateralised
    const int32_t* __restrict__ surv;          // [n_dates-1] atom ids
    const int32_t* __restrict__ cond;
    double* __restrict__ out;                  // [1+NP][ld]
    int64_t ex_stride;
    double threshold, lgd;
    int32_t n_dates, collateralized;
};

__global__ __launch_bounds__(MCX_BLOCK) void kt_cva(const KTCArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= a.b.n) return;
    DN cva = dconst<NP>(0.0);
    for (int m = 0; m < a.n_dates - 1; ++m) {                                             // cva_metric.py:78-96
        const DN pos = kt_unsecured(a.expo, a.ex_stride, a.rows, a.delayed, a.collateralized, a.threshold, a.b.ld, m, i);
        if (!(pos.v > 0.0)) continue;                    // relu: gradient passes where the unsecured exposure is positive
        const int sa = ldk(a.surv + m), ca = ldk(a.cond + m);
        const DN sp = kt_atom(a.b, ldk_struct(&a.b.atoms[sa]), sa, i);
        const DN cs = kt_atom(a.b, ldk_struct(&a.b.atoms[ca]), ca, i);
        cva = cva + pos * sp * (1.0 - cs);
    }
    a.out[i] = cva.v * a.lgd;
#pragma unroll
    for (int q = 0; q < NP; ++q) a.out[(1 + q) * a.b.ld + i] = cva.d[q] * a.lgd;
}

// ---- EPE / ENE profile tangents: sum_i 1[u > 0] du and sum_i 1[u < 0] du per metric date (epe_metric.py, ene_metric.py) -----
struct KTFArgs {
    const double* __restrict__ expo;           // [1+NP][n_rows][ld] of one netting set
    const int32_t* __restrict__ rows;
    const int32_t* __restrict__ delayed;
    double* __restrict__ partials;             // [n_dates][gridDim.x][2*NP]
    int64_t ex_stride, n, ld;
    double threshold;
    int32_t collateralized, pad;
};

__global__ __launch_bounds__(MCX_BLOCK) void kt_profiles(const KTFArgs a)
{
    const int m = blockIdx.y;
    double acc[2 * NP];
#pragma unroll
    for (int q = 0; q < 2 * NP; ++q) acc[q] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const DN u = kt_unsecured(a.expo, a.ex_stride, a.rows, a.delayed, a.collateralized, a.threshold, a.ld, m, i);
        const double wp = u.v > 0.0 ? 1.0 : 0.0, wn = u.v < 0.0 ? 1.0 : 0.0;      // torch.relu: zero gradient at 0
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            acc[q] = fma(wp, u.d[q], acc[q]);
            acc[NP + q] = fma(wn, u.d[q], acc[NP + q]);
        }
    }
    __shared__ double lds[4];
#pragma unroll
    for (int q = 0; q < 2 * NP; ++q) {
        const double r = block_sum(acc[q], lds);
        if (threadIdx.x == 0) a.partials[((int64_t)m * gridDim.x + blockIdx.x) * 2 * NP + q] = r;
    }
}

// ---- PFE tangent: d x_(r) / d theta = the tangent of the path that realises the order statistic (pfe_metric.py:61-66: the
// reference differentiates through torch.sort, i.e. through the selected element) -------------------------------------------
struct KTQArgs {
    const double* __restrict__ expo;
    const int32_t* __restrict__ rows;
    const int32_t* __restrict__ delayed;
    const double* __restrict__ targets;        // [n_dates] exact order statistic of the unsecured exposure (K5 radix select)
    unsigned long long* __restrict__ first;    // [n_dates] smallest local path index with u == target (init: ~0)
    double* __restrict__ out;                  // [n_dates][1+NP]: index (or -1), tangent
    int64_t ex_stride, n, ld;
    double threshold;
    int32_t collateralized, pad;
};

__global__ __launch_bounds__(MCX_BLOCK) void kt_pick_find(const KTQArgs a)
{
    const int m = blockIdx.y;
    const double target = ldk(a.targets + m);
    unsigned long long best = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const DN u = kt_unsecured(a.expo, a.ex_stride, a.rows, a.delayed, a.collateralized, a.threshold, a.ld, m, i);
        if (u.v == target && (unsigned long long)i < best) best = (unsigned long long)i;
    }
    if (best != ~0ull) atomicMin(a.first + m, best);
}

__global__ void kt_pick_read(const KTQArgs a, int n_dates)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_dates) return;
    const unsigned long long i = a.first[m];
    double* o = a.out + (int64_t)m * (1 + NP);
    if (i == ~0ull) {
        o[0] = -1.0;
        for (int q = 0; q < NP; ++q) o[1 + q] = 0.0;
        return;
    }
    const DN u = kt_unsecured(a.expo, a.ex_stride, a.rows, a.delayed, a.collateralized, a.threshold, a.ld, m, (int64_t)i);
    o[0] = (double)i;
    for (int q = 0; q < NP; ++q) o[1 + q] = u.d[q];
}

template <int NSLOT, int NZ>
void launch_ktp(const KTPArgs& a, int grid, bool inject, hipStream_t s)
{
    if (inject) hipLaunchKernelGGL((kt_paths<NSLOT, NZ, true>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((kt_paths<NSLOT, NZ, false>), dim3(grid), dim3(MCX_BLOCK), 0, s, a);
}

struct DevBuf {            // scoped device allocation
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t upload(const void* src, size_t bytes, hipStream_t s)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, s) : hipSuccess;
    }
};

int fill_book(mcx_handle* h, const mcx_book* b, const double* d_datoms, const double* d_paths, const double* d_dpaths, int64_t n_paths,
              int64_t ld, int32_t n_dates, KTBook* out)
{
    out->terms = b->d_terms; out->events = b->d_events; out->atoms = b->d_atoms; out->datoms = d_datoms; out->paths = d_paths;
    out->dpaths = d_dpaths; out->n = n_paths; out->ld = ld; out->pstride = (int64_t)n_dates * b->n_state * ld;
    out->n_state = b->n_state; out->n_basis = b->n_basis;
    return 0;
}

// ids of the atoms behind every event / term (the flattened device records hold copies, the derivative rows are per atom id)
int upload_ids(mcx_handle* h, const mcx_book* b, DevBuf& ev_ids, DevBuf& term_atom, hipStream_t s)
{
    std::vector<KTEventIds> ids((size_t)b->n_events);
    for (int q = 0; q < b->n_events; ++q) { ids[q].num = b->h_event_num_atom[q]; ids[q].x = b->h_event_x_atom[q]; }
    std::vector<int32_t> ta((size_t)b->n_terms);
    for (int j = 0; j < b->n_terms; ++j) ta[j] = b->h_term_atom[j];
    MCX_HIP(h, ev_ids.upload(ids.data(), sizeof(KTEventIds) * ids.size(), s));
    MCX_HIP(h, term_atom.upload(ta.data(), sizeof(int32_t) * ta.size(), s));
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}

DevAtom flat_atom(const mcx_book* b, int id)
{
    DevAtom o; const mcx_atom& q = b->h_atoms[id];
    o.t_idx = q.t_idx; o.col = q.col; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1;
    return o;
}

}  // namespace

extern "C" int mcx_tangent_paths(mcx_handle* h, const mcx_sim* sim, const double* h_dslot, const double* h_dinit, const double* h_daux,
                                 uint64_t seed, uint64_t path_offset, int64_t n_paths, double* d_paths, double* d_dpaths, int64_t ld,
                                 const double* d_inject_z, void* stream)
{
    if (!h || !sim || !h_dslot || !h_dinit || !h_daux || !d_paths || !d_dpaths) return -1;
    if (n_paths <= 0) return 0;
    if (ld < n_paths) MCX_FAIL(h, -2, "mcx_tangent_paths: ld < n_paths");
    const mcx_sim_desc& sd = sim->desc;
    if (sd.scheme != MCX_SCHEME_EULER) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_paths: EULER scheme only");
    for (int s = 0; s < sd.n_slots; ++s) {
        const int kd = sd.slots[s].kind;
        if (kd != MCX_MODEL_BS && kd != MCX_MODEL_VASICEK && kd != MCX_MODEL_CIRPP && kd != MCX_MODEL_CIRPP_DET)
            MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_paths: slot %d: model kind %d has no tangent step", s, kd);
    }
    if (sd.n_z != sd.n_slots) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_paths: one normal per slot expected");
    hipStream_t s = (hipStream_t)stream;
    DevBuf dslot, dinit, daux;
    MCX_HIP(h, dslot.upload(h_dslot, sizeof(double) * (size_t)sd.n_slots * MCX_SLOT_NPARAM * NP, s));
    MCX_HIP(h, dinit.upload(h_dinit, sizeof(double) * (size_t)sd.n_state * NP, s));
    MCX_HIP(h, daux.upload(h_daux, sizeof(double) * (size_t)sd.n_steps * sd.n_slots * MCX_AUX * NP, s));
    KTPArgs a;
    memset(&a, 0, sizeof(a));
    mcx_fill_k1_args(sim, seed, path_offset, n_paths, ld, d_paths, d_inject_z, nullptr, &a.k1);
    a.dslot = (const double*)dslot.p; a.dinit = (const double*)dinit.p; a.daux = (const double*)daux.p; a.dpaths = d_dpaths;
    a.pstride = (int64_t)sd.n_dates * sd.n_state * ld; a.n_slots = sd.n_slots;
    const int grid = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    const bool inj = d_inject_z != nullptr;
    switch (sd.n_slots) {
    case 1: launch_ktp<1, 1>(a, grid, inj, s); break;
    case 2: launch_ktp<2, 2>(a, grid, inj, s); break;
    case 3: launch_ktp<3, 3>(a, grid, inj, s); break;
    case 4: launch_ktp<4, 4>(a, grid, inj, s); break;
    default: MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_paths: %d slots have no instantiation", sd.n_slots);
    }
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}

extern "C" int mcx_tangent_lsm(mcx_handle* h, const mcx_book* b, int32_t product, int32_t first_event, int32_t num_atom, int32_t x_atom,
                               double shift, double scale, const double* d_datoms, const double* d_paths, const double* d_dpaths,
                               int64_t n_paths, int64_t ld, int32_t n_dates, double* h_moments, void* stream)
{
    if (!h || !b || !d_datoms || !d_paths || !d_dpaths || !h_moments) return -1;
    if (product < 0 || product >= b->n_products) MCX_FAIL(h, -2, "mcx_tangent_lsm: product out of range");
    const DevProduct& pr = b->h_products[product];
    if (pr.n_states != 1) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm: stateless products only");
    const int n_cf = pr.cf_end - pr.cf_begin;
    if (first_event < 0 || first_event > n_cf) MCX_FAIL(h, -2, "mcx_tangent_lsm: first_event out of range");
    for (int q = pr.cf_begin; q < pr.cf_end; ++q) {
        const DevEvent& e = b->h_events[q];
        if (!(e.kind == MCX_EV_CASHFLOW || (e.kind == MCX_EV_OPTION && e.aux[0] == 0.0)))
            MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm: event %d (kind %d, mode %g) has no tangent form", q, e.kind, e.aux[0]);
        if (e.kind == MCX_EV_OPTION)
            for (int j = e.term_begin; j < e.term_end; ++j)
                if (b->h_terms[j].den >= 0) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm: option over per-term denominators");
    }
    if (num_atom < 0 || num_atom >= b->n_atoms || x_atom < 0 || x_atom >= b->n_atoms) MCX_FAIL(h, -2, "mcx_tangent_lsm: atom out of range");
    const int K = b->n_basis, NM = (2 * K - 1) + K;
    if (n_paths <= 0) { memset(h_moments, 0, sizeof(double) * (size_t)(1 + NP) * NM); return 0; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf ev_ids, term_atom;
    if (int rc = upload_ids(h, b, ev_ids, term_atom, s)) return rc;
    const int grid = mcx_grid_for(n_paths, MCX_BLOCK, 2 * h->n_cu);
    const int count = (1 + NP) * NM;
    if ((size_t)(grid + 1) * count * sizeof(double) > h->ws_bytes || (size_t)count * sizeof(double) > h->pinned_bytes)
        MCX_FAIL(h, -2, "mcx_tangent_lsm: workspace too small");
    KTLArgs a;
    memset(&a, 0, sizeof(a));
    fill_book(h, b, d_datoms, d_paths, d_dpaths, n_paths, ld, n_dates, &a.b);
    a.ev_ids = (const KTEventIds*)ev_ids.p; a.term_atom = (const int32_t*)term_atom.p;
    a.num = flat_atom(b, num_atom); a.x = flat_atom(b, x_atom); a.num_id = num_atom; a.x_id = x_atom; a.shift = shift; a.scale = scale;
    a.partials = h->d_ws; a.ev_first = pr.cf_begin + first_event; a.ev_end = pr.cf_end;
    switch (K) {
    case 1: hipLaunchKernelGGL((kt_lsm<1>), dim3(grid), dim3(MCX_BLOCK), 0, s, a); break;
    case 2: hipLaunchKernelGGL((kt_lsm<2>), dim3(grid), dim3(MCX_BLOCK), 0, s, a); break;
    case 3: hipLaunchKernelGGL((kt_lsm<3>), dim3(grid), dim3(MCX_BLOCK), 0, s, a); break;
    case 4: hipLaunchKernelGGL((kt_lsm<4>), dim3(grid), dim3(MCX_BLOCK), 0, s, a); break;
    default: MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm: basis size %d has no instantiation", K);
    }
    MCX_HIP(h, hipGetLastError());
    double* d_out = h->d_ws + (size_t)grid * count;
    hipLaunchKernelGGL(kt_sum_partials, dim3((count + 63) / 64), dim3(64), 0, s, h->d_ws, count, grid, d_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h->h_pinned, d_out, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_moments, h->h_pinned, sizeof(double) * (size_t)count);
    return 0;
}

extern "C" int mcx_tangent_lsm_step(mcx_handle* h, const mcx_book* b, int32_t product, int32_t roll_begin, int32_t roll_end, int32_t num_atom,
                                    int32_t x_atom, double shift, double scale, const double* d_datoms, const double* d_paths,
                                    const double* d_dpaths, int64_t n_paths, int64_t ld, int32_t n_dates, double* d_W, double* d_dW,
                                    int64_t ld_w, double* h_moments, void* stream)
{
    if (!h || !b || !d_datoms || !d_paths || !d_dpaths || !d_W || !d_dW || !h_moments) return -1;
    if (product < 0 || product >= b->n_products) MCX_FAIL(h, -2, "mcx_tangent_lsm_step: product out of range");
    const DevProduct& pr = b->h_products[product];
    const int n_cf = pr.cf_end - pr.cf_begin;
    if (roll_begin < 0 || roll_end < roll_begin || roll_end > n_cf) MCX_FAIL(h, -2, "mcx_tangent_lsm_step: roll window out of range");
    if (ld < n_paths || ld_w < n_paths) MCX_FAIL(h, -2, "mcx_tangent_lsm_step: leading dimension < n_paths");
    for (int q = pr.cf_begin + roll_begin; q < pr.cf_begin + roll_end; ++q) {
        const DevEvent& e = b->h_events[q];
        const bool ok = e.kind == MCX_EV_CASHFLOW || (e.kind == MCX_EV_OPTION && e.aux[0] == 0.0) ||
                        (e.kind == MCX_EV_EXERCISE && (e.aux[0] == 0.0 || e.aux[0] == 1.0));
        if (!ok) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm_step: event %d (kind %d, mode %g) has no tangent form", q, e.kind, e.aux[0]);
        if (e.kind != MCX_EV_CASHFLOW)
            for (int j = e.term_begin; j < e.term_end; ++j)
                if (b->h_terms[j].den >= 0) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm_step: option over per-term denominators");
    }
    if (num_atom < 0 || num_atom >= b->n_atoms || x_atom < 0 || x_atom >= b->n_atoms) MCX_FAIL(h, -2, "mcx_tangent_lsm_step: atom out of range");
    const int K = b->n_basis, S = pr.n_states, NM = (2 * K - 1) + S * K;
    if (n_paths <= 0) { memset(h_moments, 0, sizeof(double) * (size_t)(1 + NP) * NM); return 0; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf ev_ids, term_atom;
    if (int rc = upload_ids(h, b, ev_ids, term_atom, s)) return rc;
    const int grid = mcx_grid_for(n_paths, MCX_BLOCK, 2 * h->n_cu);
    const int count = (1 + NP) * NM;
    if ((size_t)(grid + 1) * count * sizeof(double) > h->ws_bytes || (size_t)count * sizeof(double) > h->pinned_bytes)
        MCX_FAIL(h, -2, "mcx_tangent_lsm_step: workspace too small");
    KTSArgs a;
    memset(&a, 0, sizeof(a));
    fill_book(h, b, d_datoms, d_paths, d_dpaths, n_paths, ld, n_dates, &a.b);
    a.ev_ids = (const KTEventIds*)ev_ids.p; a.term_atom = (const int32_t*)term_atom.p; a.coeffs = b->d_coeffs;
    a.W = d_W; a.dW = d_dW; a.ld_w = ld_w; a.w_stride = (int64_t)S * ld_w;
    a.num = flat_atom(b, num_atom); a.x = flat_atom(b, x_atom); a.num_id = num_atom; a.x_id = x_atom; a.shift = shift; a.scale = scale;
    a.partials = h->d_ws; a.roll_begin = pr.cf_begin + roll_begin; a.roll_end = pr.cf_begin + roll_end;
    bool launched = true;
#define MCX_KTS(KK, SS) hipLaunchKernelGGL((kt_lsm_step<KK, SS>), dim3(grid), dim3(MCX_BLOCK), 0, s, a)
    switch (K * 16 + S) {
    case 2 * 16 + 1: MCX_KTS(2, 1); break;  case 2 * 16 + 2: MCX_KTS(2, 2); break;  case 2 * 16 + 3: MCX_KTS(2, 3); break;
    case 3 * 16 + 1: MCX_KTS(3, 1); break;  case 3 * 16 + 2: MCX_KTS(3, 2); break;  case 3 * 16 + 3: MCX_KTS(3, 3); break;
    case 3 * 16 + 4: MCX_KTS(3, 4); break;  case 4 * 16 + 1: MCX_KTS(4, 1); break;  case 4 * 16 + 2: MCX_KTS(4, 2); break;
    default: launched = false; break;
    }
#undef MCX_KTS
    if (!launched) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_lsm_step: (basis=%d, states=%d) has no instantiation", K, S);
    MCX_HIP(h, hipGetLastError());
    double* d_out = h->d_ws + (size_t)grid * count;
    hipLaunchKernelGGL(kt_sum_partials, dim3((count + 63) / 64), dim3(64), 0, s, h->d_ws, count, grid, d_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h->h_pinned, d_out, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_moments, h->h_pinned, sizeof(double) * (size_t)count);
    return 0;
}

extern "C" int mcx_tangent_eval(mcx_handle* h, const mcx_book* b, const double* d_datoms, const double* d_coeffs, const double* d_dcoeffs,
                                const double* d_paths, const double* d_dpaths, int64_t n_paths, int64_t ld, int32_t n_dates,
                                double* d_cfs, double* d_expo, const int32_t* h_ev_param, void* stream)
{
    if (!h || !b || !d_datoms || !d_coeffs || !d_dcoeffs || !d_paths || !d_dpaths || !d_cfs || !d_expo) return -1;
    if (n_paths <= 0) return 0;
    for (int p = 0; p < b->n_products; ++p) {
        const DevProduct& pr = b->h_products[p];
        for (int q = pr.ev_begin; q < pr.ev_end; ++q) {
            const DevEvent& e = b->h_events[q];
            const bool ok = e.kind == MCX_EV_CASHFLOW || (e.kind == MCX_EV_OPTION && e.aux[0] == 0.0) || e.kind == MCX_EV_EXPO_POLY ||
                            (e.kind == MCX_EV_EXPO_BS && h_ev_param != nullptr) ||
                            (e.kind == MCX_EV_EXERCISE && pr.n_states > 1 && (e.aux[0] == 0.0 || e.aux[0] == 1.0));
            if (!ok) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_eval: event %d (kind %d) has no tangent form", q, e.kind);
            if (e.kind == MCX_EV_OPTION || e.kind == MCX_EV_EXERCISE)
                for (int j = e.term_begin; j < e.term_end; ++j)
                    if (b->h_terms[j].den >= 0) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "mcx_tangent_eval: option over per-term denominators");
        }
    }
    hipStream_t s = (hipStream_t)stream;
    DevBuf ev_ids, term_atom, ev_param;
    if (int rc = upload_ids(h, b, ev_ids, term_atom, s)) return rc;
    if (h_ev_param) MCX_HIP(h, ev_param.upload(h_ev_param, sizeof(int32_t) * 2 * (size_t)b->n_events, s));
    const int n_rows = b->n_expo_rows > 0 ? b->n_expo_rows : 1;
    MCX_HIP(h, hipMemsetAsync(d_cfs, 0, sizeof(double) * (size_t)(1 + NP) * b->n_netting_sets * ld, s));
    MCX_HIP(h, hipMemsetAsync(d_expo, 0, sizeof(double) * (size_t)(1 + NP) * b->n_netting_sets * n_rows * ld, s));
    KTEArgs a;
    memset(&a, 0, sizeof(a));
    fill_book(h, b, d_datoms, d_paths, d_dpaths, n_paths, ld, n_dates, &a.b);
    a.ev_ids = (const KTEventIds*)ev_ids.p; a.term_atom = (const int32_t*)term_atom.p; a.products = b->d_products;
    a.coeffs = d_coeffs; a.dcoeffs = d_dcoeffs; a.cfs = d_cfs; a.expo = d_expo; a.n_products = b->n_products;
    a.n_ns = b->n_netting_sets; a.n_rows = n_rows; a.ev_param = h_ev_param ? (const int32_t*)ev_param.p : nullptr;
    hipLaunchKernelGGL(kt_eval, dim3((unsigned)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK)), dim3(MCX_BLOCK), 0, s, a);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}

extern "C" int mcx_tangent_cva(mcx_handle* h, const mcx_book* b, const double* d_datoms, const int32_t* h_rows, const int32_t* h_surv,
                               const int32_t* h_cond, const int32_t* h_delayed, int32_t collateralized, int32_t n_dates_metric,
                               double threshold, double recovery, const double* d_expo_ns,
                               int64_t expo_tangent_stride, const double* d_paths, const double* d_dpaths, int64_t n_paths, int64_t ld,
                               int32_t n_dates, double* d_out, void* stream)
{
    if (!h || !b || !d_datoms || !h_rows || !h_surv || !h_cond || !d_expo_ns || !d_paths || !d_dpaths || !d_out) return -1;
    if (n_paths <= 0 || n_dates_metric < 1) return 0;
    for (int m = 0; m < n_dates_metric - 1; ++m)
        if (h_surv[m] < 0 || h_surv[m] >= b->n_atoms || h_cond[m] < 0 || h_cond[m] >= b->n_atoms) MCX_FAIL(h, -2, "mcx_tangent_cva: atom out of range");
    hipStream_t s = (hipStream_t)stream;
    DevBuf rows, surv, cond, delayed;
    MCX_HIP(h, rows.upload(h_rows, sizeof(int32_t) * (size_t)n_dates_metric, s));
    if (h_delayed) MCX_HIP(h, delayed.upload(h_delayed, sizeof(int32_t) * (size_t)n_dates_metric, s));
    MCX_HIP(h, surv.upload(h_surv, sizeof(int32_t) * (size_t)(n_dates_metric - 1), s));
    MCX_HIP(h, cond.upload(h_cond, sizeof(int32_t) * (size_t)(n_dates_metric - 1), s));
    KTCArgs a;
    memset(&a, 0, sizeof(a));
    fill_book(h, b, d_datoms, d_paths, d_dpaths, n_paths, ld, n_dates, &a.b);
    a.expo = d_expo_ns; a.rows = (const int32_t*)rows.p; a.surv = (const int32_t*)surv.p; a.cond = (const int32_t*)cond.p; a.out = d_out;
    a.ex_stride = expo_tangent_stride; a.threshold = threshold; a.lgd = 1.0 - recovery; a.n_dates = n_dates_metric;
    a.delayed = h_delayed ? (const int32_t*)delayed.p : nullptr; a.collateralized = collateralized;
    hipLaunchKernelGGL(kt_cva, dim3((unsigned)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK)), dim3(MCX_BLOCK), 0, s, a);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}

extern "C" int mcx_tangent_profiles(mcx_handle* h, const int32_t* h_rows, const int32_t* h_delayed, int32_t collateralized,
                                    int32_t n_dates_metric, double threshold, const double* d_expo_ns,
                                    int64_t expo_tangent_stride, int64_t n_paths, int64_t ld, double* h_out, void* stream)
{
    if (!h || !h_rows || !d_expo_ns || !h_out) return -1;
    if (n_dates_metric <= 0) return 0;
    memset(h_out, 0, sizeof(double) * (size_t)n_dates_metric * 2 * NP);
    if (n_paths <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int grid = mcx_grid_for(n_paths, MCX_BLOCK, 64);
    std::vector<double> part((size_t)n_dates_metric * grid * 2 * NP);
    DevBuf rows, d_part, delayed;
    MCX_HIP(h, rows.upload(h_rows, sizeof(int32_t) * (size_t)n_dates_metric, s));
    if (h_delayed) MCX_HIP(h, delayed.upload(h_delayed, sizeof(int32_t) * (size_t)n_dates_metric, s));
    MCX_HIP(h, hipMalloc(&d_part.p, sizeof(double) * part.size()));
    KTFArgs a;
    a.expo = d_expo_ns; a.rows = (const int32_t*)rows.p; a.delayed = h_delayed ? (const int32_t*)delayed.p : nullptr;
    a.collateralized = collateralized; a.pad = 0; a.partials = (double*)d_part.p; a.ex_stride = expo_tangent_stride;
    a.n = n_paths; a.ld = ld; a.threshold = threshold;
    hipLaunchKernelGGL(kt_profiles, dim3(grid, n_dates_metric), dim3(MCX_BLOCK), 0, s, a);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(part.data(), d_part.p, sizeof(double) * part.size(), hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    for (int m = 0; m < n_dates_metric; ++m)
        for (int b = 0; b < grid; ++b)
            for (int q = 0; q < 2 * NP; ++q) h_out[(size_t)m * 2 * NP + q] += part[((size_t)m * grid + b) * 2 * NP + q];
    return 0;
}

extern "C" int mcx_tangent_pick(mcx_handle* h, const int32_t* h_rows, const int32_t* h_delayed, int32_t collateralized,
                                int32_t n_dates_metric, double threshold, const double* h_targets, const double* d_expo_ns,
                                int64_t expo_tangent_stride, int64_t n_paths, int64_t ld, double* h_out, void* stream)
{
    if (!h || !h_rows || !h_targets || !d_expo_ns || !h_out) return -1;
    if (n_dates_metric <= 0) return 0;
    for (int m = 0; m < n_dates_metric; ++m) { h_out[(size_t)m * (1 + NP)] = -1.0; for (int q = 0; q < NP; ++q) h_out[(size_t)m * (1 + NP) + 1 + q] = 0.0; }
    if (n_paths <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DevBuf rows, delayed, targets, first, out;
    MCX_HIP(h, rows.upload(h_rows, sizeof(int32_t) * (size_t)n_dates_metric, s));
    if (h_delayed) MCX_HIP(h, delayed.upload(h_delayed, sizeof(int32_t) * (size_t)n_dates_metric, s));
    MCX_HIP(h, targets.upload(h_targets, sizeof(double) * (size_t)n_dates_metric, s));
    MCX_HIP(h, hipMalloc(&first.p, sizeof(unsigned long long) * (size_t)n_dates_metric));
    MCX_HIP(h, hipMemsetAsync(first.p, 0xFF, sizeof(unsigned long long) * (size_t)n_dates_metric, s));
    MCX_HIP(h, hipMalloc(&out.p, sizeof(double) * (size_t)n_dates_metric * (1 + NP)));
    KTQArgs a;
    a.expo = d_expo_ns; a.rows = (const int32_t*)rows.p; a.delayed = h_delayed ? (const int32_t*)delayed.p : nullptr;
    a.targets = (const double*)targets.p; a.first = (unsigned long long*)first.p; a.out = (double*)out.p;
    a.ex_stride = expo_tangent_stride; a.n = n_paths; a.ld = ld; a.threshold = threshold; a.collateralized = collateralized; a.pad = 0;
    hipLaunchKernelGGL(kt_pick_find, dim3(mcx_grid_for(n_paths, MCX_BLOCK, 256), n_dates_metric), dim3(MCX_BLOCK), 0, s, a);
    hipLaunchKernelGGL(kt_pick_read, dim3((n_dates_metric + 63) / 64), dim3(64), 0, s, a, n_dates_metric);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h_out, out.p, sizeof(double) * (size_t)n_dates_metric * (1 + NP), hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}
