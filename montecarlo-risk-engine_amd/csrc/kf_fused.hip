// kf_fused.hip — the fused main-simulation pass: K1 (Philox + Box-Muller + Cholesky + SDE step) + K2 (requests,
// cashflows, exposures) + K4 (PV / EPE / ENE / CVA reductions) in ONE launch, no intermediate tensor in HBM.
//
// Reference dataflow replaced (controller/controller.py:677-694): generate_paths -> paths[N,T,D] (1.6 GB at config 3)
// -> resolve_requests (~2.4 GB) -> evaluate_products -> exposures[E,N] (0.4 GB) -> metric reductions.  Here a lane keeps
// its path's state in VGPRs, runs the book's events of a timeline date right after the sub-steps that reach it (every
// atom of such an event reads the state of THAT date), and folds the result into per-path accumulators (cashflows, CVA
// integrand) and per-date block accumulators (EPE / ENE) that live in LDS across the block's path tiles.
//
// Block-level reduction design: per record r the block keeps a shift c_r (value of the first path it sees) and, per wave,
// the pair (sum (x-c), sum (x-c)^2) in LDS; every date costs one wave64 shuffle reduction per record.  Blocks write one
// (n, c, s1, s2) record each; a tiny second kernel merges blocks with Chan's update (deterministic, no float atomics).
#include "kf_common.h"

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdlib>

namespace {

// straight-line evaluation of a FastDate (see the struct); returns false when the date must be interpreted
template <int NSLOT, int SIG, int NNS, bool STORE, bool ALL_FAST>
__device__ __forceinline__ bool kf_fast_date(const FusedArgs& a, int t, int64_t i, bool live, bool first_tile, double* __restrict__ lds,
                                             const double (&reg)[2 * NSLOT], double (&cfs)[NNS], double (&cva)[NNS])
{
    constexpr int NREG = 2 * NSLOT;
    if (NNS != 1) return false;
    if (!ALL_FAST && ldk(&a.fast[t].valid) != 1) return false;
    const FastDate* __restrict__ fp = a.fast + t;                // every field is a scalar load at its point of use
#define FD(x) ldk(&fp->x)
    const K1Args& k = a.k1;
    if (STORE && k.paths && live) sim_store_state<NSLOT, SIG>(k, t, i, reg);
    const int flags = FD(flags);
    const double inv = (flags & 16) ? FD(ni_c0) : mcx_exp(fma(FD(ni_c1), f_regsel<NREG>(FD(ni_reg), reg), FD(ni_c0)));
    if (flags & 1) {
        double val = fma(FD(k1), f_regsel<NREG>(FD(lin_reg), reg), FD(k0));
        const int n_exp = FD(n_exp);
#pragma unroll 1
        for (int j = 0; j < n_exp; ++j)
            val = fma(FD(t_w[j]), mcx_exp(fma(FD(t_c1[j]), f_regsel<NREG>(FD(t_reg[j]), reg), FD(t_c0[j]))), val);
        cfs[0] = fma(val, inv, cfs[0]);
    }
    double e = 0.0;
    if (flags & 2) {
        const double x = fma(FD(x_d), f_regsel<NREG>(FD(x_reg), reg), FD(x_a));
        double p = 0.0;
        const int off0 = FD(coeff_off0), off1 = FD(coeff_off1);
        if (off0 >= 0) {
            const double* __restrict__ c = a.coeffs + off0;
            double xp = 1.0;
#pragma unroll 1
            for (int q = 0; q < a.n_basis; ++q) { p = fma(ldk(c + q), xp, p); xp *= x; }
        }
        if (off1 >= 0) {
            const double* __restrict__ c = a.coeffs + off1;
            double xp = 1.0;
#pragma unroll 1
            for (int q = 0; q < a.n_basis; ++q) { p = fma(ldk(c + q), xp, p); xp *= x; }
        }
        e = p * inv;
    }
    const int row = ldk(a.date_row + t);
    if (a.expo && row >= 0 && live) a.expo[(int64_t)row * a.ld_out + i] = e;
    if (flags & 32) {
        const double u = dev_thr(e, FD(thr));
        if (flags & 8) {
            const int rp = FD(rec_profile);
            f_record(fmax(u, 0.0), live, rp, a.n_rec, first_tile, lds);
            f_record(fmin(u, 0.0), live, rp + 1, a.n_rec, first_tile, lds);
        }
        if (flags & 4) {
            const double sp = FD(s_b) * mcx_exp(fma(FD(s_c1), f_regsel<NREG>(FD(s_reg), reg), FD(s_c0)));
            double cs = FD(c_a);
            const double cb = FD(c_b);
            if (cb != 0.0) cs = fma(cb, mcx_exp(fma(FD(c_c1), f_regsel<NREG>(FD(c_reg), reg), FD(c_c0))), cs);
            cva[0] = fma(fmax(u, 0.0), sp * (1.0 - cs), cva[0]);
        }
    }
#undef FD
    return true;
}

// The book's events + metric operations of ONE timeline date for one lane.  The date's program chunk sits in the wave's
// private LDS slot (`chunk`): all 64 lanes read the same addresses (LDS broadcast, no bank conflicts), so walking the
// program costs ds_read latency (~64-128 clk) instead of a chain of dependent L2 round trips per record.
template <int NSLOT, int SIG, int NNS, int NSTA, bool STORE>
__device__ __forceinline__ void kf_on_date(const FusedArgs& a, int t, int64_t i, bool live, bool first_tile, double* __restrict__ lds,
                                           const unsigned char* __restrict__ chunk, const double (&reg)[2 * NSLOT],
                                           double (&cfs)[NNS], double (&cva)[NNS], int (&est)[NSTA])
{
    constexpr int NREG = 2 * NSLOT;
    const K1Args& k = a.k1;
    const int n_rec = a.n_rec;
    if (STORE && k.paths && live) sim_store_state<NSLOT, SIG>(k, t, i, reg);
    const ChunkHeader* hdp = (const ChunkHeader*)chunk;
    ChunkHeader hd;
    hd.n_ev = RFL(hdp->n_ev); hd.n_mop = RFL(hdp->n_mop); hd.n_terms = RFL(hdp->n_terms); hd.bytes = 0;
    const FEvent* __restrict__ evs = (const FEvent*)(chunk + sizeof(ChunkHeader));
    const FTerm* __restrict__ terms = (const FTerm*)(evs + hd.n_ev);
    const FMetricOp* __restrict__ mops = (const FMetricOp*)(terms + hd.n_terms);
    double e_ns[NNS];
#pragma unroll
    for (int q = 0; q < NNS; ++q) e_ns[q] = 0.0;
    double inv_num = 0.0;
#pragma unroll 1
    for (int q = 0; q < hd.n_ev; ++q) {
        const FEvent& ev = evs[q];
        struct { int kind, flags, term_begin, term_end, coeff_off, ns, sidx, init_state; double strike, sign; const FAtom &num, &x; } e =
            {RFL(ev.kind), RFL(ev.flags), RFL(ev.term_begin), RFL(ev.term_end), RFL(ev.coeff_off), RFL(ev.ns), RFL(ev.sidx),
             RFL(ev.init_state), ev.strike, ev.sign, ev.num, ev.x};
        if (!(e.flags & 2)) inv_num = 1.0 / f_atom<NREG>(e.num, reg);       // flag bit1: same numeraire as the previous event
        double v = 0.0;
        if (e.kind <= MCX_EV_EXERCISE) {
            double val = 0.0, glog = 0.0;
            const int basket = e.kind == MCX_EV_OPTION ? RFL((int)ev.aux[0]) : 0;      // basket aggregation mode (basket_option.py)
#pragma unroll 1
            for (int j = e.term_begin; j < e.term_end; ++j) {
                const double av = f_atom<NREG>(terms[j].atom, reg);
                val = fma(terms[j].w, av, val);
                if (basket == 1 || basket == 2) glog = fma(terms[j].w, mcx_log(av + 1e-10), glog);
            }
            if (e.kind == MCX_EV_CASHFLOW) {
                v = val * inv_num;
            } else {
                const double imm = fmax(e.sign * (val - e.strike), 0.0);
                if (e.kind == MCX_EV_OPTION) {
                    v = imm * inv_num;
                    if (basket == 3) {                                   // binary payoff: fuzzy indicator (binary_option.py:38-43)
                        const double dot = fmin(fmax((val - e.strike + ev.aux[2]) / (2.0 * ev.aux[2]), 0.0), 1.0);
                        v = ev.aux[1] * (e.sign > 0.0 ? dot : 1.0 - dot) * inv_num;
                    } else if (basket) {
                        const double geo = fmax(e.sign * (mcx_exp(glog) - e.strike), 0.0);
                        v = (basket == 1 ? geo : imm - geo + ev.aux[1]) * inv_num;
                    }
                } else {
                    int s = 0;
#pragma unroll
                    for (int w = 0; w < NSTA; ++w) s = (e.sidx == w) ? est[w] : s;
                    double cont = 0.0, cont_ex = 0.0;
                    if (e.coeff_off >= 0) {
                        const double x = f_atom<NREG>(e.x, reg);
                        cont = f_poly(a.coeffs + e.coeff_off + s * a.n_basis, a.n_basis, x);
                        if (RFL((int)ev.aux[0]) == 1)            // FlexiCall: continuation after exercising (flexicall.py:118-133)
                            cont_ex = s > 0 ? f_poly(a.coeffs + e.coeff_off + (s - 1) * a.n_basis, a.n_basis, x) : 0.0;
                    }
                    const bool ex = (imm + cont_ex > cont) && (s > 0);
                    v = ex ? imm * inv_num : 0.0;
#pragma unroll
                    for (int w = 0; w < NSTA; ++w) est[w] = (e.sidx == w && ex) ? s - 1 : est[w];
                }
            }
#pragma unroll
            for (int w = 0; w < NNS; ++w) cfs[w] += (NNS == 1 || e.ns == w) ? v : 0.0;
        } else {
            if (e.kind == MCX_EV_EXPO_POLY) {
                int s = e.init_state;
#pragma unroll
                for (int w = 0; w < NSTA; ++w) s = (e.sidx == w) ? est[w] : s;
                const double x = f_atom<NREG>(e.x, reg);
                if (e.coeff_off >= 0)
                    v = (e.sidx < 0 ? f_poly_uniform(a.coeffs, e.coeff_off + e.init_state * a.n_basis, a.n_basis, x)
                                    : f_poly(a.coeffs + e.coeff_off + s * a.n_basis, a.n_basis, x)) * inv_num;
            }
#pragma unroll
            for (int w = 0; w < NNS; ++w) e_ns[w] += (NNS == 1 || e.ns == w) ? v : 0.0;
        }
    }
    const int row = ldk(a.date_row + t);
    if (a.expo && row >= 0 && live) {
        for (int w = 0; w < a.n_ns; ++w) {
            double ev = e_ns[0];
#pragma unroll
            for (int q = 1; q < NNS; ++q) ev = (w == q) ? e_ns[q] : ev;
            a.expo[((int64_t)w * a.n_expo_rows + row) * a.ld_out + i] = ev;
        }
    }
#pragma unroll 1
    for (int q = 0; q < hd.n_mop; ++q) {
        const FMetricOp& mv = mops[q];
        struct { int ns, rec_profile, has_cva; double threshold; const FAtom &surv, &cond; } mo =
            {RFL(mv.ns), RFL(mv.rec_profile), RFL(mv.has_cva), mv.threshold, mv.surv, mv.cond};
        double e = e_ns[0];
#pragma unroll
        for (int w = 1; w < NNS; ++w) e = (mo.ns == w) ? e_ns[w] : e;
        const double u = dev_thr(e, mo.threshold);
        if (mo.rec_profile >= 0) {
            f_record(fmax(u, 0.0), live, mo.rec_profile, n_rec, first_tile, lds);
            f_record(fmin(u, 0.0), live, mo.rec_profile + 1, n_rec, first_tile, lds);
        }
        if (mo.has_cva) {
            const double sp = f_atom<NREG>(mo.surv, reg);
            const double cs = f_atom<NREG>(mo.cond, reg);
            const double inc = fmax(u, 0.0) * (sp * (1.0 - cs));
#pragma unroll
            for (int w = 0; w < NNS; ++w) cva[w] += (NNS == 1 || mo.ns == w) ? inc : 0.0;
        }
    }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// NNS = compile-time bound on netting sets (1 or MCX_FUSED_MAX_NS), NST = bound on exercise products (0 -> none).
// NPF = 1 KiB pieces of the NEXT date's program chunk each wave prefetches into VGPRs (one coalesced global_load_dwordx4
// per piece, issued before the sub-steps that lead to that date, written to the wave's LDS slot just before use: the HBM/L2
// latency hides under ~5 sub-steps of RNG + SDE work).  NPF = 0: chunks larger than 2 KiB are copied at use.
// SIMULATE = false: the same event/metric program runs on a paths tensor produced earlier by K1 (`k.paths` is then the
// INPUT [date][state][path]): one pass over the paths replaces K2 + K4 and writes no exposure matrix unless asked to.
template <int NSLOT, int NZ, bool INJECT, int SIG, int NNS, int NST, int NPF, bool SIMULATE>
__device__ __forceinline__ void kf_body(const FusedArgs& a)
{
    constexpr int NREG = 2 * NSLOT;
    constexpr int NSTA = NST > 0 ? NST : 1;
    constexpr int NPFA = NPF > 0 ? NPF : 1;
    constexpr bool FAST_ONLY = NPF < 0;          // every date has a FastDate record: no interpreter, no program slots
    extern __shared__ double lds[];
    const K1Args& k = a.k1;
    const int n_rec = a.n_rec;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rec_area = (9 * n_rec + 1) & ~1;                       // doubles; keeps the program slots 16-byte aligned
    unsigned char* slot = (unsigned char*)(lds + rec_area) + (size_t)wv * a.chunk_cap;       // wave-private program slot
    for (int q = threadIdx.x; q < 9 * n_rec; q += MCX_BLOCK) lds[q] = 0.0;
    __shared__ double bm_lds[(SIMULATE && !INJECT) ? MCX_BM_LDS_DOUBLES : 2];      // Box-Muller lookup tables
    const double* tab = nullptr;
    if (SIMULATE && !INJECT) { mcx_bm_load(bm_lds); tab = bm_lds; }
    __syncthreads();
    const int64_t tiles = (k.n + MCX_BLOCK - 1) / MCX_BLOCK;
    double n_block = 0.0;

    u32x4 pf[NPFA];
    auto prefetch = [&](int t) {             // issue the loads of date t's chunk (NPF > 0)
        if (NPF > 0 && t < a.n_dates) {
            const int off = ldk(a.date_off + t), end = ldk(a.date_off + t + 1);
#pragma unroll
            for (int p = 0; p < NPFA; ++p) {
                const int b = off + (p * 64 + lane) * 16;
                pf[p] = b < end ? *(const u32x4*)(a.prog + b) : u32x4{0, 0, 0, 0};
            }
        }
    };
    auto stage = [&](int t) {                // make date t's chunk visible in the wave's LDS slot
        if (NPF > 0) {
#pragma unroll
            for (int p = 0; p < NPFA; ++p) ((u32x4*)slot)[p * 64 + lane] = pf[p];
        } else {
            const int off = ldk(a.date_off + t), end = ldk(a.date_off + t + 1);
            for (int b = lane * 16; b < end - off; b += 64 * 16) *(u32x4*)(slot + b) = *(const u32x4*)(a.prog + off + b);
        }
    };

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const bool first_tile = tile == (int64_t)blockIdx.x;
        const int64_t i_raw = tile * MCX_BLOCK + threadIdx.x;
        const bool live = i_raw < k.n;
        const int64_t i = live ? i_raw : k.n - 1;          // dead lanes shadow the last path (no stores, no contribution)
        { const int64_t rest = k.n - tile * MCX_BLOCK; n_block += (double)(rest < MCX_BLOCK ? rest : MCX_BLOCK); }

        double reg[NREG];                                  // reg[2s], reg[2s+1] = state of slot s
        sim_init_state<NSLOT, SIG>(k, reg);
        double cfs[NNS], cva[NNS];
#pragma unroll
        for (int q = 0; q < NNS; ++q) { cfs[q] = 0.0; cva[q] = 0.0; }
        int est[NSTA];
#pragma unroll
        for (int q = 0; q < NSTA; ++q) est[q] = a.init_state[q];

        auto on_date = [&](int t, auto store) {
            constexpr bool ST = decltype(store)::value;
            if (FAST_ONLY) {
                kf_fast_date<NSLOT, SIG, NNS, ST, true>(a, t, i, live, first_tile, lds, reg, cfs, cva);
            } else {
                const bool fast = kf_fast_date<NSLOT, SIG, NNS, ST, false>(a, t, i, live, first_tile, lds, reg, cfs, cva);
                if (!fast) stage(t);
                prefetch(t + 1);
                if (!fast) kf_on_date<NSLOT, SIG, NNS, NSTA, ST>(a, t, i, live, first_tile, lds, slot, reg, cfs, cva, est);
            }
        };
        int next_t = 0;                                    // timeline dates are visited in increasing order, each once
        if (!FAST_ONLY) prefetch(0);
        if (SIMULATE) {
            for (int t = 0; t < k.n_initial_store; ++t) {
                on_date(t, std::integral_constant<bool, true>());
                next_t = t + 1;
            }
            const uint64_t path = k.path_offset + (uint64_t)i;
            // two nested loops: the inner one is the bare sub-step recursion up to the next timeline date (register
            // allocation then treats it as the hot loop: the date program's scalars are not kept alive / spilled through it)
            int step = 0;
#pragma unroll 1
            while (step < k.n_steps) {
                int st;
#pragma unroll 1
                do {
                    sim_substep<NSLOT, NZ, INJECT, SIG>(k, step, path, i, reg, tab);
                    st = ldk(&k.steps[step].store_idx);
                    ++step;
                } while (st < 0 && step < k.n_steps);
                if (st >= 0) {
                    on_date(st, std::integral_constant<bool, true>());
                    next_t = st + 1;
                }
            }
        } else {
            const int D = k.n_state;
            double nxt[NREG];
            auto load_row = [&](int t, double (&dst)[NREG]) {
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const bool bs = sig_kind(SIG, s) >= 0 ? sig_is_bs(SIG, s) : (k.slots[s].kind == MCX_MODEL_BS);
                    const int c = k.slots[s].state_off;
                    dst[2 * s] = k.paths[((int64_t)t * D + c) * k.ld + i];
                    dst[2 * s + 1] = bs ? 0.0 : k.paths[((int64_t)t * D + c + 1) * k.ld + i];
                }
            };
            load_row(0, nxt);
#pragma unroll 1
            for (int t = 0; t < a.n_dates; ++t) {
#pragma unroll
                for (int q = 0; q < NREG; ++q) reg[q] = nxt[q];
                if (t + 1 < a.n_dates) load_row(t + 1, nxt);      // next date's state streams in while this date's ops run
                on_date(t, std::integral_constant<bool, false>());
                next_t = t + 1;
            }
        }
        (void)next_t;
        // per-path quantities
        for (int w = 0; w < a.n_ns; ++w) {
            double cv = cfs[0], cc = cva[0];
#pragma unroll
            for (int q = 1; q < NNS; ++q) { cv = (w == q) ? cfs[q] : cv; cc = (w == q) ? cva[q] : cc; }
            if (a.cfs && live) a.cfs[(int64_t)w * a.ld_out + i] = cv;
            if (a.rec_pv[w] >= 0) f_record(cv, live, a.rec_pv[w], n_rec, first_tile, lds);
            if (a.rec_cva[w] >= 0) f_record(cc * a.lgd[w], live, a.rec_cva[w], n_rec, first_tile, lds);
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n_rec; r += MCX_BLOCK) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < 4; ++w) { s1 += lds[n_rec + (w * n_rec + r) * 2]; s2 += lds[n_rec + (w * n_rec + r) * 2 + 1]; }
        double* dst = a.partials + ((int64_t)blockIdx.x * n_rec + r) * 4;
        dst[0] = n_block; dst[1] = lds[r]; dst[2] = s1; dst[3] = s2;
    }
}

template <int NSLOT, int NZ, bool INJECT, int SIG, int NNS, int NST, int NPF, bool SIMULATE>
__global__ __launch_bounds__(MCX_BLOCK) void kf_fused(const FusedArgs a)
{
    kf_body<NSLOT, NZ, INJECT, SIG, NNS, NST, NPF, SIMULATE>(a);
}

// merge per-block records (Chan, Golub, LeVeque pairwise update) -> out[r] = (n, mean, 0, M2).
// One 256-thread block per record: every thread folds a strided subset of the block records (independent loads in
// flight), the 256 partial triples are then combined through LDS.  (A one-thread serial merge over 2048 dependent loads
// took 0.8 ms — a quarter of the whole pass.)
__global__ __launch_bounds__(MCX_BLOCK) void kf_merge(const double* __restrict__ partials, int n_rec, int n_blocks, mcx_acc* __restrict__ out)
{
    const int r = blockIdx.x;
    __shared__ double sN[MCX_BLOCK], sM[MCX_BLOCK], sQ[MCX_BLOCK];
    double N = 0.0, mean = 0.0, M2 = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += MCX_BLOCK) {
        const double* p = partials + ((int64_t)b * n_rec + r) * 4;
        const double n = p[0];
        if (n > 0.0) chan_merge(N, mean, M2, n, p[1] + p[2] / n, fmax(p[3] - p[2] * p[2] / n, 0.0));
    }
    sN[threadIdx.x] = N; sM[threadIdx.x] = mean; sQ[threadIdx.x] = M2;
    __syncthreads();
    for (int stride = MCX_BLOCK / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride) {
            double a = sN[threadIdx.x], b = sM[threadIdx.x], c = sQ[threadIdx.x];
            chan_merge(a, b, c, sN[threadIdx.x + stride], sM[threadIdx.x + stride], sQ[threadIdx.x + stride]);
            sN[threadIdx.x] = a; sM[threadIdx.x] = b; sQ[threadIdx.x] = c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[r].n = sN[0]; out[r].shift = sM[0]; out[r].s1 = 0.0; out[r].s2 = sQ[0]; }
}

template <int NSLOT, int NZ, int SIG>
void launch_kf(const FusedArgs& a, int grid, size_t lds, int npf, bool inject, bool simulate, hipStream_t s)
{
    const bool one_ns = a.n_ns == 1, no_state = a.n_stateful == 0;
#define MCX_KF(INJ, NNS, NST, NPF) do { if (simulate) hipLaunchKernelGGL((kf_fused<NSLOT, NZ, INJ, SIG, NNS, NST, NPF, true>), dim3(grid), dim3(MCX_BLOCK), lds, s, a); \
        else if (!INJ) hipLaunchKernelGGL((kf_fused<NSLOT, NZ, false, SIG, NNS, NST, NPF, false>), dim3(grid), dim3(MCX_BLOCK), lds, s, a); } while (0)
#define MCX_KF_NPF(INJ, NNS, NST) do { if (npf == 1) MCX_KF(INJ, NNS, NST, 1); else if (npf == 2) MCX_KF(INJ, NNS, NST, 2); else MCX_KF(INJ, NNS, NST, 0); } while (0)
    if (inject) {
        if (one_ns && no_state) MCX_KF_NPF(true, 1, 0);
        else MCX_KF(true, MCX_FUSED_MAX_NS, MCX_FUSED_MAX_STATEFUL, 0);
    } else {
        if (one_ns && no_state) MCX_KF_NPF(false, 1, 0);
        else if (one_ns) MCX_KF_NPF(false, 1, MCX_FUSED_MAX_STATEFUL);
        else MCX_KF(false, MCX_FUSED_MAX_NS, MCX_FUSED_MAX_STATEFUL, 0);
    }
#undef MCX_KF_NPF
#undef MCX_KF
}

}  // namespace

struct mcx_fused {
    const mcx_sim* sim;
    const mcx_book* book;
    int n_rec, n_ns, n_dates, n_expo_rows, n_stateful, want_pv;
    unsigned char* d_prog;
    FastDate* d_fast;
    int32_t* d_date_off;
    int32_t* d_date_row;
    int chunk_cap, npf;
    int lean;                  // every date has a FastDate record kf_lean.hip can run (valid != 0)
    LeanTerm* d_lterms;
    double* d_vcoef;           // this object's copy of the exercise-value polynomial coefficients (the book may rebuild its own)
    int32_t rec_pv[MCX_FUSED_MAX_NS], rec_cva[MCX_FUSED_MAX_NS];
    double lgd[MCX_FUSED_MAX_NS];
    int32_t init_state[MCX_FUSED_MAX_STATEFUL];
    double* d_partials;
    size_t partial_bytes;
    mcx_acc* d_out;
    // optional timing of the main kernel alone (mcx_fused_set_timing): event pairs around its launch, on the launch stream
    mutable hipEvent_t tev[2 * MCX_FUSED_TIMING_RING];
    mutable int timing, t_count, t_launch;      // timing: 0 off, k > 0: every k-th launch is timed
};

static void fused_timing_off(mcx_fused* f)
{
    if (f->timing) for (int q = 0; q < 2 * MCX_FUSED_TIMING_RING; ++q) hipEventDestroy(f->tev[q]);
    f->timing = 0; f->t_count = 0; f->t_launch = 0;
}

extern "C" int mcx_fused_set_timing(mcx_fused* f, int32_t enable)
{
    if (!f) return -1;
    fused_timing_off(f);
    if (enable) {
        for (int q = 0; q < 2 * MCX_FUSED_TIMING_RING; ++q)
            if (hipEventCreate(&f->tev[q]) != hipSuccess) { for (int r = 0; r < q; ++r) hipEventDestroy(f->tev[r]); return -100; }
        f->timing = enable > 0 ? enable : 1;
    }
    return 0;
}

extern "C" int mcx_fused_kernel_times(mcx_fused* f, float* h_ms, int32_t capacity, int32_t* n_out)
{
    if (!f || !h_ms || !n_out) return -1;
    *n_out = 0;
    if (!f->timing) return 0;
    const int n = f->t_count < capacity ? f->t_count : capacity;
    for (int q = 0; q < n; ++q) {
        if (hipEventSynchronize(f->tev[2 * q + 1]) != hipSuccess) return -100;
        if (hipEventElapsedTime(&h_ms[q], f->tev[2 * q], f->tev[2 * q + 1]) != hipSuccess) return -100;
    }
    *n_out = n;
    f->t_count = 0; f->t_launch = 0;
    return 0;
}

extern "C" void mcx_fused_destroy(mcx_fused* f)
{
    if (!f) return;
    fused_timing_off(f);
    hipFree(f->d_lterms); hipFree(f->d_vcoef); hipFree(f->d_prog); hipFree(f->d_fast); hipFree(f->d_date_off); hipFree(f->d_date_row); hipFree(f->d_partials); hipFree(f->d_out);
    delete f;
}

extern "C" int mcx_fused_num_records(const mcx_fused* f) { return f ? f->n_rec : -1; }

extern "C" int mcx_fused_create(mcx_handle* h, const mcx_sim* sim, const mcx_book* book, const mcx_fused_desc* d, mcx_fused** out)
{
    if (!h || !sim || !book || !d || !out) return -1;
    const mcx_sim_desc& sd = sim->desc;
    if (book->n_state != sd.n_state) MCX_FAIL(h, -2, "mcx_fused_create: book and sim disagree on the state dimension");
    if (d->n_netting_sets < 1 || d->n_netting_sets > MCX_FUSED_MAX_NS || d->n_netting_sets != book->n_netting_sets)
        MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: %d netting sets (max %d)", book->n_netting_sets, MCX_FUSED_MAX_NS);
    if (d->n_expo_rows != book->n_expo_rows) MCX_FAIL(h, -2, "mcx_fused_create: n_expo_rows mismatch");
    const int T = sd.n_dates;
    for (int s = 0; s < sd.n_slots; ++s)
        if (sd.slots[s].kind == MCX_MODEL_S2F) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: the Schwartz two-factor spot is a derived state column");
    // state column -> lane register
    int col_reg[MCX_MAX_STATE];
    for (int s = 0; s < sd.n_slots; ++s) {
        col_reg[sd.slots[s].state_off] = 2 * s;
        if (sd.slots[s].kind != MCX_MODEL_BS) col_reg[sd.slots[s].state_off + 1] = 2 * s + 1;
    }
    bool ok = true;
    std::string why;
    auto fatom = [&](const mcx_atom& q, int t_event) {
        FAtom o; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1; o.reg = -1;
        o.pad = (q.b != 0.0 ? 1 : 0) | ((q.a != 0.0 || q.d != 0.0) ? 2 : 0);
        if (q.col >= 0) {
            if (q.t_idx != t_event) { ok = false; why = "an event reads the state of another date (unequal swap tenors)"; }
            o.reg = col_reg[q.col];
        }
        return o;
    };
    const DevEventVec& hev = book->h_events;
    const std::vector<DevTerm>& hte = book->h_terms;
    auto devatom_to_mcx = [](const DevAtom& q) { mcx_atom o; o.t_idx = q.t_idx; o.col = q.col; o.a = q.a; o.d = q.d; o.b = q.b; o.c0 = q.c0; o.c1 = q.c1; return o; };

    // events bucketed by timeline date, product order preserved
    std::vector<std::vector<FEvent>> by_date(T);
    std::vector<std::vector<int>> by_date_q(T);              // book index of each bucketed event
    std::vector<std::vector<FTerm>> terms_by_date(T);
    int n_stateful = 0;
    int32_t init_state[MCX_FUSED_MAX_STATEFUL] = {0, 0, 0, 0};
    if ((int)book->h_event_t_idx.size() != book->n_events) MCX_FAIL(h, -2, "mcx_fused_create: internal (event dates missing)");
    for (int p = 0; p < book->n_products && ok; ++p) {
        const DevProduct& pr = book->h_products[p];
        if (pr.ev_end == pr.ev_begin) continue;
        int sidx = -1;
        if (pr.n_states > 1) {
            if (n_stateful >= MCX_FUSED_MAX_STATEFUL) { ok = false; why = "too many exercise products"; break; }
            sidx = n_stateful;
            init_state[n_stateful++] = pr.init_state;
        }
        for (int q = pr.ev_begin; q < pr.ev_end && ok; ++q) {
            const DevEvent& e = hev[q];
            const int t = book->h_event_t_idx[q];
            if (t < 0 || t >= T) { ok = false; why = "event date out of range"; break; }
            FEvent fe;
            memset(&fe, 0, sizeof(fe));
            fe.kind = e.kind; fe.flags = e.flags; fe.coeff_off = e.coeff_off; fe.row = e.row; fe.ns = pr.netting_set; fe.sidx = sidx;
            fe.init_state = pr.init_state; fe.pad = pr.n_states; fe.strike = e.strike; fe.sign = e.sign;
            for (int w = 0; w < 4; ++w) fe.aux[w] = e.aux[w];
            if (e.kind == MCX_EV_OPTION && (e.aux[0] == 4.0 || e.aux[0] == 5.0)) { ok = false; why = "barrier monitoring is evaluated by the book kernel (K2)"; }
            fe.num = fatom(devatom_to_mcx(e.num), t);
            fe.x = fatom(devatom_to_mcx(e.x), t);
            std::vector<FTerm>& terms = terms_by_date[t];
            fe.term_begin = (int)terms.size();
            for (int j = e.term_begin; j < e.term_end; ++j) {
                if (hte[j].den >= 0) { ok = false; why = "a cashflow term carries its own numeraire (unequal swap tenors)"; break; }
                FTerm ft; ft.w = hte[j].w; ft.atom = fatom(devatom_to_mcx(hte[j].atom), t);
                terms.push_back(ft);
            }
            fe.term_end = (int)terms.size();
            // numeraire identical to the previous event of this date: reuse 1/numeraire (flag bit1)
            if (!by_date[t].empty() && memcmp(&by_date[t].back().num, &fe.num, sizeof(FAtom)) == 0) fe.flags |= 2;
            else fe.flags &= ~2;
            by_date[t].push_back(fe);
            by_date_q[t].push_back(q);
        }
    }
    // metric ops
    std::vector<std::vector<FMetricOp>> mop_by_date(T);
    std::vector<int32_t> date_row(T, -1);
    for (int r = 0; r < d->n_expo_rows; ++r) {
        const int t = d->row_t_idx[r];
        if (t < 0 || t >= T) MCX_FAIL(h, -2, "mcx_fused_create: exposure row %d has no timeline date", r);
        date_row[t] = r;
    }
    mcx_fused* f = new mcx_fused();
    memset(f, 0, sizeof(*f));
    int n_rec = 0;
    for (int k = 0; k < MCX_FUSED_MAX_NS; ++k) { f->rec_pv[k] = -1; f->rec_cva[k] = -1; f->lgd[k] = 0.0; }
    for (int k = 0; k < d->n_netting_sets && ok; ++k) {
        const mcx_fused_ns_desc& nd = d->ns[k];
        if (nd.netting_set != k) { ok = false; why = "netting-set descriptors must be in order"; break; }
        if (d->want_pv) f->rec_pv[k] = n_rec++;
        int rec_prof = -1;
        if (nd.want_profiles) { rec_prof = n_rec; n_rec += 2 * nd.n_dates; }
        if (nd.want_cva) { f->rec_cva[k] = n_rec++; f->lgd[k] = 1.0 - nd.recovery; }
        for (int m = 0; m < nd.n_dates; ++m) {
            const int row = nd.row[m];
            if (row < 0 || row >= d->n_expo_rows) { ok = false; why = "metric row out of range"; break; }
            const int t = d->row_t_idx[row];
            FMetricOp mo;
            memset(&mo, 0, sizeof(mo));
            mo.ns = k; mo.m = m; mo.threshold = nd.threshold;
            mo.rec_profile = nd.want_profiles ? rec_prof + 2 * m : -1;
            mo.has_cva = (nd.want_cva && m < nd.n_dates - 1) ? 1 : 0;
            mo.surv.reg = -1; mo.cond.reg = -1; mo.surv.pad = 0; mo.cond.pad = 0;
            if (mo.has_cva) {
                const int sa = nd.surv_atoms[m], ca = nd.cond_atoms[m];
                if (sa < 0 || sa >= book->n_atoms || ca < 0 || ca >= book->n_atoms) { ok = false; why = "CVA atom out of range"; break; }
                mo.surv = fatom(book->h_atoms[sa], t);
                mo.cond = fatom(book->h_atoms[ca], t);
            }
            if (mo.rec_profile >= 0 || mo.has_cva) mop_by_date[t].push_back(mo);
        }
    }
    if (!ok) { delete f; MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: %s", why.c_str()); }
    if (n_rec < 1) { delete f; MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: no reducible metric requested"); }
    if ((size_t)9 * n_rec * sizeof(double) > 48 * 1024) { delete f; MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: %d records exceed the LDS budget", n_rec); }

    // straight-line records for dates of the linear-book shape (FastDate); such dates need no interpreted chunk
    std::vector<FastDate> fast(T);
    std::vector<LeanTerm> lterms;
    std::vector<double> vcoef;
    const bool fast_dates_enabled = d->n_netting_sets == 1;
    for (int t = 0; t < T; ++t) {
        FastDate fd;
        memset(&fd, 0, sizeof(fd));
        fd.ni_reg = fd.lin_reg = fd.x_reg = fd.s_reg = fd.c_reg = -1;
        fd.coeff_off0 = fd.coeff_off1 = -1; fd.rec_profile = -1;
        for (int j = 0; j < 4; ++j) fd.t_reg[j] = -1;
        bool okf = fast_dates_enabled, lean_only = false;
        const FAtom* num = nullptr;
        int n_expo = 0;
        for (size_t ei = 0; ei < by_date[t].size(); ++ei) {
            const FEvent& e = by_date[t][ei];
            if (!okf) break;
            const bool stateful = e.sidx >= 0;
            if (stateful && (e.sidx != 0 || e.pad != 2 || d->n_netting_sets != 1)) { okf = false; break; }      // one two-state product
            const bool plain_option = e.kind == MCX_EV_OPTION && e.aux[0] == 0.0 && !stateful;
            if (e.kind != MCX_EV_CASHFLOW && e.kind != MCX_EV_EXPO_POLY && !plain_option &&
                !(e.kind == MCX_EV_EXERCISE && stateful && e.aux[0] == 0.0)) { okf = false; break; }
            if (stateful && e.kind == MCX_EV_CASHFLOW) { okf = false; break; }
            if ((plain_option || e.kind == MCX_EV_CASHFLOW) && (fd.flags & 512)) { okf = false; break; }      // the payoff's max() stands alone
            if (plain_option) {
                if (fd.flags & 1) { okf = false; break; }
                fd.flags |= 512; lean_only = true;
                fd.op_strike = e.strike; fd.op_sign = e.sign;
            }
            if (num && memcmp(num, &e.num, sizeof(FAtom)) != 0) { okf = false; break; }     // one numeraire per date
            num = &e.num;
            if (e.kind == MCX_EV_EXERCISE) {
                if ((fd.flags & (128 | 2)) || e.x.b != 0.0) { okf = false; break; }            // one exercise event, before the exposure
                fd.flags |= 128; lean_only = true;
                fd.ex_term_off = (int32_t)lterms.size(); fd.ex_coeff_off = e.coeff_off; fd.ex_strike = e.strike; fd.ex_sign = e.sign;
                fd.ex_x_reg = e.x.reg; fd.ex_x_a = e.x.a; fd.ex_x_d = e.x.d; fd.ex_lin_reg = -1;
                for (int j = e.term_begin; j < e.term_end && okf; ++j) {
                    const FTerm& tm = terms_by_date[t][j];
                    fd.ex_k0 += tm.w * tm.atom.a;
                    if (tm.atom.d != 0.0) {
                        if (fd.ex_lin_reg >= 0 && fd.ex_lin_reg != tm.atom.reg) okf = false;
                        fd.ex_lin_reg = tm.atom.reg; fd.ex_k1 += tm.w * tm.atom.d;
                    }
                    if (tm.atom.b != 0.0) {
                        if (tm.atom.reg < 0 || tm.atom.c1 == 0.0) fd.ex_k0 += tm.w * tm.atom.b * exp(tm.atom.c0);      // state-independent term
                        else { LeanTerm lt; lt.w = tm.w * tm.atom.b; lt.c0 = tm.atom.c0; lt.c1 = tm.atom.c1; lt.reg = tm.atom.reg; lt.pad = 0; lterms.push_back(lt); }
                    }
                }
                fd.ex_n = (int32_t)lterms.size() - fd.ex_term_off;
                // the book's verified value polynomial of this event (mcx_book_collapse_values), copied into this object
                const int vq = book->h_event_vpoly.empty() ? -1 : book->h_event_vpoly[by_date_q[t][ei]];
                if (okf && vq >= 0) {
                    const DevVPoly& vp = book->h_vpoly[vq];
                    fd.flags |= 1024;
                    fd.ex_p_lo = vp.lo; fd.ex_p_hi = vp.hi; fd.ex_p_ms = vp.ms; fd.ex_p_ih = vp.ih;
                    fd.ex_p_off = (int32_t)vcoef.size(); fd.ex_p_blk = vp.n_blk; fd.ex_p_reg = col_reg[vp.col];
                    vcoef.insert(vcoef.end(), book->h_vcoef.begin() + vp.coef_off, book->h_vcoef.begin() + vp.coef_off + (size_t)vp.n_blk * MCX_VPOLY_BLK);
                }
                continue;
            }
            if (e.kind == MCX_EV_CASHFLOW || plain_option) {
                fd.flags |= 1;
                for (int j = e.term_begin; j < e.term_end && okf; ++j) {
                    const FTerm& tm = terms_by_date[t][j];
                    fd.k0 += tm.w * tm.atom.a;
                    if (tm.atom.d != 0.0) {
                        if (fd.lin_reg >= 0 && fd.lin_reg != tm.atom.reg) okf = false;
                        fd.lin_reg = tm.atom.reg; fd.k1 += tm.w * tm.atom.d;
                    }
                    if (tm.atom.b != 0.0) {
                        if (fd.n_exp >= 4) { okf = false; break; }
                        fd.t_reg[fd.n_exp] = tm.atom.reg; fd.t_w[fd.n_exp] = tm.w * tm.atom.b;
                        fd.t_c0[fd.n_exp] = tm.atom.c0; fd.t_c1[fd.n_exp] = tm.atom.c1; fd.n_exp++;
                    }
                }
            } else {
                if (e.x.b != 0.0 || n_expo >= 2) { okf = false; break; }
                if ((fd.flags & 2) && (fd.x_reg != e.x.reg || fd.x_a != e.x.a || fd.x_d != e.x.d)) { okf = false; break; }
                if (stateful && n_expo > 0) { okf = false; break; }                            // the state-indexed exposure stands alone
                if ((fd.flags & 256)) { okf = false; break; }
                fd.flags |= 2; fd.x_reg = e.x.reg; fd.x_a = e.x.a; fd.x_d = e.x.d;
                int off = e.coeff_off >= 0 ? e.coeff_off + e.init_state * book->n_basis : -1;
                if (stateful) { fd.flags |= 256; lean_only = true; off = e.coeff_off; if (off < 0) { okf = false; break; } }
                (n_expo == 0 ? fd.coeff_off0 : fd.coeff_off1) = off;
                n_expo++;
            }
        }
        if (okf && num) {
            if (num->a == 0.0 && num->d == 0.0 && num->b > 0.0) { fd.ni_reg = num->reg; fd.ni_c0 = -num->c0 - log(num->b); fd.ni_c1 = -num->c1; }
            else if (num->b == 0.0 && num->d == 0.0 && num->a != 0.0) { fd.flags |= 16; fd.ni_c0 = 1.0 / num->a; }
            else okf = false;
        }
        if (okf && mop_by_date[t].size() > 1) okf = false;
        if (okf && mop_by_date[t].size() == 1) {
            const FMetricOp& mo = mop_by_date[t][0];
            fd.flags |= 32; fd.thr = mo.threshold;
            if (mo.rec_profile >= 0) { fd.flags |= 8; fd.rec_profile = mo.rec_profile; }
            if (mo.has_cva) {
                if (mo.surv.a != 0.0 || mo.surv.d != 0.0 || mo.cond.d != 0.0) okf = false;
                fd.flags |= 4;
                fd.s_reg = mo.surv.reg; fd.s_b = mo.surv.b; fd.s_c0 = mo.surv.c0; fd.s_c1 = mo.surv.c1;
                fd.c_reg = mo.cond.reg; fd.c_a = mo.cond.a; fd.c_b = mo.cond.b; fd.c_c0 = mo.cond.c0; fd.c_c1 = mo.cond.c1;
            }
        }
        if (okf && !num && !mop_by_date[t].empty()) { fd.flags |= 16; fd.ni_c0 = 1.0; }      // metric op on an empty date
        // a state reference that is absent (reg < 0) becomes register 0 with a zero coefficient: the kernels may then index the
        // lane's state registers without a range test
        auto bind = [](int32_t& r, double& coef) { if (r < 0) { r = 0; coef = 0.0; } };
        bind(fd.ni_reg, fd.ni_c1); bind(fd.lin_reg, fd.k1); bind(fd.x_reg, fd.x_d); bind(fd.s_reg, fd.s_c1); bind(fd.c_reg, fd.c_c1);
        bind(fd.ex_x_reg, fd.ex_x_d); bind(fd.ex_lin_reg, fd.ex_k1);
        for (int j = 0; j < 4; ++j) bind(fd.t_reg[j], fd.t_c1[j]);
        // merged discount x survival factor of the CVA increment: relu(p / N) S (1 - Sc) = relu(p) (S / N) (1 - Sc) when the date
        // has no threshold and records no EPE / ENE profile: one exponential instead of two
        if (okf && (fd.flags & 4) && !(fd.flags & 8) && fd.thr == 0.0) {
            fd.flags |= 64;
            if (fd.flags & 16) { fd.m_b = fd.s_b * fd.ni_c0; fd.m_c0 = fd.s_c0; fd.m_n1 = 0.0; }
            else { fd.m_b = fd.s_b; fd.m_c0 = fd.s_c0 + fd.ni_c0; fd.m_n1 = fd.ni_c1; }
            fd.m_s1 = fd.s_c1;
        }
        if (!okf) lterms.resize(fd.ex_term_off <= (int32_t)lterms.size() && (fd.flags & 128) ? fd.ex_term_off : lterms.size());
        fd.valid = okf ? (lean_only ? 2 : 1) : 0;
        fast[t] = fd;
    }
    // per-date program chunks: header | events | terms | metric ops, 16-byte aligned
    std::vector<unsigned char> prog;
    std::vector<int32_t> date_off(T + 1, 0);
    int max_chunk = 0;
    bool bs_exposure = false;
    for (int t = 0; t < T; ++t) for (const FEvent& e : by_date[t]) if (e.kind == MCX_EV_EXPO_BS) bs_exposure = true;
    for (int t = 0; t < T; ++t) {
        date_off[t] = (int32_t)prog.size();
        if (fast[t].valid == 1) continue;              // evaluated from its FastDate record in every kernel: no interpreted chunk
        ChunkHeader hd;
        hd.n_ev = (int)by_date[t].size(); hd.n_mop = (int)mop_by_date[t].size(); hd.n_terms = (int)terms_by_date[t].size();
        const size_t bytes = sizeof(hd) + sizeof(FEvent) * by_date[t].size() + sizeof(FTerm) * terms_by_date[t].size() +
                             sizeof(FMetricOp) * mop_by_date[t].size();
        const size_t padded = (bytes + 15) & ~(size_t)15;
        hd.bytes = (int32_t)padded;
        const size_t base = prog.size();
        prog.resize(base + padded, 0);
        unsigned char* p = prog.data() + base;
        memcpy(p, &hd, sizeof(hd)); p += sizeof(hd);
        if (hd.n_ev) { memcpy(p, by_date[t].data(), sizeof(FEvent) * by_date[t].size()); p += sizeof(FEvent) * by_date[t].size(); }
        if (hd.n_terms) { memcpy(p, terms_by_date[t].data(), sizeof(FTerm) * terms_by_date[t].size()); p += sizeof(FTerm) * terms_by_date[t].size(); }
        if (hd.n_mop) memcpy(p, mop_by_date[t].data(), sizeof(FMetricOp) * mop_by_date[t].size());
        max_chunk = std::max(max_chunk, (int)padded);
    }
    date_off[T] = (int32_t)prog.size();
    if (bs_exposure) { delete f; MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: analytic Black-Scholes exposures are evaluated by the book kernel (K2)"); }
    // LDS budget: 4 wave slots + the record area must fit comfortably (several blocks per CU)
    bool all_fast = d->n_netting_sets == 1 && n_stateful == 0, lean = d->n_netting_sets == 1 && n_stateful <= 1;
    for (int t = 0; t < T; ++t) { all_fast = all_fast && fast[t].valid == 1; lean = lean && fast[t].valid != 0; }
    f->npf = max_chunk <= 1024 ? 1 : (max_chunk <= 2048 ? 2 : 0);
    f->chunk_cap = f->npf > 0 ? f->npf * 1024 : ((max_chunk + 255) & ~255);
    if (all_fast) { f->npf = -1; f->chunk_cap = 0; }      // no interpreted date at all: the straight-line instantiations
    // the straight-line kernel skips the zero test under the CIR++ diffusion root: it needs a positive initial intensity (the
    // reference asserts y0 > 0, cirpp.py:40; the state is floored at 1e-12 after every step); other starts run the interpreter
    for (int q = 0; q < sd.n_slots; ++q)
        if (sd.slots[q].kind == MCX_MODEL_CIRPP && !(sd.init_state[sd.slots[q].state_off] > 0.0)) lean = false;
    f->lean = lean ? 1 : 0;
    if ((size_t)4 * f->chunk_cap + sizeof(double) * 9 * (size_t)n_rec > 60 * 1024) {
        delete f;
        MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: a date's event program (%d B) exceeds the per-wave LDS slot budget", max_chunk);
    }

    f->sim = sim; f->book = book; f->n_rec = n_rec; f->n_ns = d->n_netting_sets; f->n_dates = T; f->n_expo_rows = d->n_expo_rows;
    f->n_stateful = n_stateful; f->want_pv = d->want_pv;
    for (int q = 0; q < MCX_FUSED_MAX_STATEFUL; ++q) f->init_state[q] = init_state[q];
    auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    prog.resize(prog.size() + 4096, 0);       // slack so the fixed-size prefetch of the last chunk stays in bounds
    lterms.resize(lterms.size() + 1);          // never empty
    vcoef.resize(vcoef.size() + MCX_VPOLY_BLK, 0.0);      // the spare block the kernels prefetch
    f->partial_bytes = sizeof(double) * 4 * (size_t)n_rec * 2048;
    hipError_t e = up((void**)&f->d_prog, prog.data(), prog.size());
    if (e == hipSuccess) e = up((void**)&f->d_fast, fast.data(), sizeof(FastDate) * fast.size());
    if (e == hipSuccess) e = up((void**)&f->d_lterms, lterms.data(), sizeof(LeanTerm) * lterms.size());
    if (e == hipSuccess) e = up((void**)&f->d_vcoef, vcoef.data(), sizeof(double) * vcoef.size());
    if (e == hipSuccess) e = up((void**)&f->d_date_off, date_off.data(), sizeof(int32_t) * date_off.size());
    if (e == hipSuccess) e = up((void**)&f->d_date_row, date_row.data(), sizeof(int32_t) * date_row.size());
    if (e == hipSuccess) e = hipMalloc(&f->d_partials, f->partial_bytes);
    if (e == hipSuccess) e = hipMalloc(&f->d_out, sizeof(mcx_acc) * (size_t)n_rec);
    if (e != hipSuccess) {                                   // nothing leaks: the partially built object is destroyed
        mcx_fused_destroy(f);
        MCX_FAIL(h, -100 - (int)e, "mcx_fused_create: %s", hipGetErrorString(e));
    }
    *out = f;
    return 0;
}

static int fused_run_impl(mcx_handle* h, const mcx_fused* f, bool simulate, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                          double* d_paths, int64_t ld, double* d_cfs, double* d_expo, int64_t ld_out,
                          const double* d_inject_z, const double* d_inject_u, mcx_acc* h_out, mcx_acc* d_out, void* stream)
{
    // exactly one of h_out (host records, the call synchronises) / d_out (device records, stream-ordered, no synchronisation)
    if (!h || !f || (h_out == nullptr) == (d_out == nullptr)) return -1;
    if (n_paths <= 0) {
        if (h_out) memset(h_out, 0, sizeof(mcx_acc) * (size_t)f->n_rec);
        else MCX_HIP(h, hipMemsetAsync(d_out, 0, sizeof(mcx_acc) * (size_t)f->n_rec, (hipStream_t)stream));
        return 0;
    }
    if ((d_paths || d_inject_z) && ld < n_paths) MCX_FAIL(h, -2, "mcx_fused_run: ld < n_paths");
    if ((d_cfs || d_expo) && ld_out < n_paths) MCX_FAIL(h, -2, "mcx_fused_run: ld_out < n_paths");
    if ((size_t)f->n_rec * sizeof(mcx_acc) > h->pinned_bytes) MCX_FAIL(h, -2, "mcx_fused_run: too many records");
    const mcx_sim_desc& sd = f->sim->desc;
    if (sd.n_uniform && d_inject_z && !d_inject_u) MCX_FAIL(h, -3, "mcx_fused_run: inject_u required with inject_z under QE");
    if (!simulate && (!d_paths || ld < n_paths)) MCX_FAIL(h, -2, "mcx_fused_eval_paths: a paths tensor with ld >= n_paths is required");
    FusedArgs a;
    memset(&a, 0, sizeof(a));
    mcx_fill_k1_args(f->sim, seed, path_offset, n_paths, ld > 0 ? ld : n_paths, d_paths, d_inject_z, d_inject_u, &a.k1);
    a.prog = f->d_prog; a.fast = f->d_fast; a.lterms = f->d_lterms; a.vcoef = f->d_vcoef; a.date_off = f->d_date_off; a.chunk_cap = f->chunk_cap;
    a.date_row = f->d_date_row; a.coeffs = f->book->d_coeffs; a.cfs = d_cfs; a.expo = d_expo; a.partials = f->d_partials;
    a.ld_out = ld_out; a.n_dates = f->n_dates; a.n_basis = f->book->n_basis; a.n_ns = f->n_ns; a.n_rec = f->n_rec;
    a.n_expo_rows = f->n_expo_rows; a.n_stateful = f->n_stateful;
    for (int k = 0; k < MCX_FUSED_MAX_NS; ++k) { a.rec_pv[k] = f->rec_pv[k]; a.rec_cva[k] = f->rec_cva[k]; a.lgd[k] = f->lgd[k]; }
    for (int k = 0; k < MCX_FUSED_MAX_STATEFUL; ++k) a.init_state[k] = f->init_state[k];
    hipStream_t s = (hipStream_t)stream;
    const bool inj = d_inject_z != nullptr;
    int grid;
    const bool timed = f->timing && (f->t_launch++ % f->timing) == 0 && f->t_count < MCX_FUSED_TIMING_RING;
    if (timed) MCX_HIP(h, hipEventRecord(f->tev[2 * f->t_count], s));
    // host records: the merge writes its few hundred bytes straight into the handle's pinned host buffer (mapped into the
    // device address space): no copy kernel and no extra dispatch on the stream between the pass and its result
    const bool direct = !d_out && sizeof(mcx_acc) * (size_t)f->n_rec <= h->pinned_bytes && h->d_pinned_alias;
    mcx_acc* dst = d_out ? d_out : (direct ? (mcx_acc*)h->d_pinned_alias : f->d_out);
    if (f->lean && (simulate || !inj)) {
        // every date is a straight-line record: the two-paths-per-lane kernel of kf_lean.hip (simulating, or streaming a paths tensor)
        // (merging the per-block records in the kernel's last block — arrival ticket + device-scope fences — was measured: the
        //  fences cost ~25 us per launch, three times the kernel boundary they save)
        grid = mcx_launch_kf_lean(a, sd, h->n_cu, inj, simulate, s);
        if (grid < 0) MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: (slots=%d, z=%d) has no fused instantiation", sd.n_slots, sd.n_z);
    } else {
        const int64_t tiles = (n_paths + MCX_BLOCK - 1) / MCX_BLOCK;
        grid = (int)std::min<int64_t>(tiles, 2048);
        // keep the tiles-per-block count integral when possible (equal work per block)
        if (tiles > 2048) { int per = (int)((tiles + 2047) / 2048); grid = (int)((tiles + per - 1) / per); }
        const size_t lds = sizeof(double) * (size_t)((9 * f->n_rec + 1) & ~1) + (size_t)4 * f->chunk_cap;
        switch (mcx_sim_signature(sd)) {
        case SIG_VAS_CIR_E: launch_kf<2, 2, SIG_VAS_CIR_E>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_BS_A: launch_kf<1, 1, SIG_BS_A>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_BS_E: launch_kf<1, 1, SIG_BS_E>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_HESTON_QE: launch_kf<1, 2, SIG_HESTON_QE>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_HESTON_E: launch_kf<1, 2, SIG_HESTON_E>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_VAS_E: launch_kf<1, 1, SIG_VAS_E>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_VAS_A: launch_kf<1, 1, SIG_VAS_A>(a, grid, lds, f->npf, inj, simulate, s); break;
        case SIG_BS_VAS_CIRDET_E: launch_kf<3, 3, SIG_BS_VAS_CIRDET_E>(a, grid, lds, f->npf, inj, simulate, s); break;
        default:
            switch (sd.n_slots * 16 + sd.n_z) {
            case 1 * 16 + 1: launch_kf<1, 1, SIG_GENERIC>(a, grid, lds, f->npf, inj, simulate, s); break;
            case 1 * 16 + 2: launch_kf<1, 2, SIG_GENERIC>(a, grid, lds, f->npf, inj, simulate, s); break;
            case 2 * 16 + 2: launch_kf<2, 2, SIG_GENERIC>(a, grid, lds, f->npf, inj, simulate, s); break;
            case 3 * 16 + 3: launch_kf<3, 3, SIG_GENERIC>(a, grid, lds, f->npf, inj, simulate, s); break;
            case 4 * 16 + 4: launch_kf<4, 4, SIG_GENERIC>(a, grid, lds, f->npf, inj, simulate, s); break;
            default: MCX_FAIL(h, MCX_E_NOT_FUSABLE, "not fusable: (slots=%d, z=%d) has no fused instantiation", sd.n_slots, sd.n_z);
            }
        }
    }
    MCX_HIP(h, hipGetLastError());
    if (timed) { MCX_HIP(h, hipEventRecord(f->tev[2 * f->t_count + 1], s)); ++f->t_count; }
    hipLaunchKernelGGL(kf_merge, dim3(f->n_rec), dim3(MCX_BLOCK), 0, s, f->d_partials, f->n_rec, grid, dst);
    MCX_HIP(h, hipGetLastError());
    if (d_out) return 0;
    if (!direct) MCX_HIP(h, hipMemcpyAsync(h->h_pinned, f->d_out, sizeof(mcx_acc) * (size_t)f->n_rec, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_out, h->h_pinned, sizeof(mcx_acc) * (size_t)f->n_rec);
    return 0;
}

extern "C" int mcx_fused_is_straight_line(const mcx_fused* f)
{
    return (f && f->lean) ? 1 : 0;
}

extern "C" int mcx_fused_run(mcx_handle* h, const mcx_fused* f, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                             double* d_paths, int64_t ld, double* d_cfs, double* d_expo, int64_t ld_out,
                             const double* d_inject_z, const double* d_inject_u, mcx_acc* h_out, void* stream)
{
    return fused_run_impl(h, f, true, seed, path_offset, n_paths, d_paths, ld, d_cfs, d_expo, ld_out, d_inject_z, d_inject_u, h_out, nullptr, stream);
}

extern "C" int mcx_fused_run_device(mcx_handle* h, const mcx_fused* f, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                                    double* d_paths, int64_t ld, double* d_cfs, double* d_expo, int64_t ld_out,
                                    const double* d_inject_z, const double* d_inject_u, mcx_acc* d_out, void* stream)
{
    return fused_run_impl(h, f, true, seed, path_offset, n_paths, d_paths, ld, d_cfs, d_expo, ld_out, d_inject_z, d_inject_u, nullptr, d_out, stream);
}

extern "C" int mcx_fused_eval_paths(mcx_handle* h, const mcx_fused* f, const double* d_paths, int64_t n_paths, int64_t ld,
                                    double* d_cfs, double* d_expo, int64_t ld_out, mcx_acc* h_out, void* stream)
{
    return fused_run_impl(h, f, false, 0, 0, n_paths, const_cast<double*>(d_paths), ld, d_cfs, d_expo, ld_out, nullptr, nullptr, h_out, nullptr, stream);
}

extern "C" int mcx_fused_eval_paths_device(mcx_handle* h, const mcx_fused* f, const double* d_paths, int64_t n_paths, int64_t ld,
                                           double* d_cfs, double* d_expo, int64_t ld_out, mcx_acc* d_out, void* stream)
{
    return fused_run_impl(h, f, false, 0, 0, n_paths, const_cast<double*>(d_paths), ld, d_cfs, d_expo, ld_out, nullptr, nullptr, nullptr, d_out, stream);
}
