// kf_lean.hip — the one-launch main pass for books whose every timeline date compiled to a straight-line FastDate record
// (kf_common.h): Philox + Box-Muller + Cholesky + SDE sub-steps, the date's cashflows / regression exposure / threshold /
// EPE-ENE records / CVA increment, and the block accumulators — nothing materialised in HBM.
//
// Reference dataflow replaced: controller/controller.py:677-694 (generate_paths -> resolve_requests -> evaluate_products ->
// metric reductions) for the linear-book shape (bond.py:115-214, swap.py:129-172 cashflows; controller.py:385-471 exposures;
// cva_metric.py:23-100; epe/ene_metric.py).
//
// Mapping to the hardware (MI355X: 256 CUs x 4 SIMDs, f64 VALU issue is the bound of this kernel — MI355X_MICROARCH.md):
//   * TWO paths per lane (PPL = 2): a lane carries two independent RNG / SDE / payoff dependency chains, so 4 resident
//     waves per SIMD give the issue parallelism of 8 and every wave-uniform cost (scalar table loads, SALU control, the
//     date record) is paid once per 128 paths instead of once per 64;
//   * the launch is sized to the residency the code object guarantees: __launch_bounds__(256, 4) => <= 128 VGPRs, 4 blocks
//     per CU for any SGPR count (800 / (ceil(sgpr/16)*16 + 16) >= 4), so grid = 4 x CUs blocks are all co-resident and the
//     path tiles (512 paths per block-tile) are dealt round-robin: 2^20 paths = 2048 tiles = exactly two rounds, no
//     partially filled last round (the round-1 kernel ran 8 waves per SIMD slot on a 6-7 wave residency: one round in
//     eight at a fraction of the issue rate);
//   * scalar state is REGION-LOCAL: every code region (tile prologue, one run of sub-steps, a date block, tile epilogue)
//     reads the kernel arguments, the FastDate record and the polynomial coefficient tables through a pointer that carries
//     a "region zero" (mcx_math.h), i.e. as scalar loads at the point of use.  Left alone, the backend loads all ~100
//     argument dwords in the prologue, keeps them live through every loop and spills them to VGPR lanes; each reload is a
//     v_readlane — a VALU instruction, the pipe this kernel is bound by (the round-1 kernel carried 82 such spills).
#include <cstdlib>
#include <map>
#include <utility>

#include "kf_common.h"

namespace {

#define FD(x) ldk(&fp->x)

// the kernel arguments as seen from one code region: the kernarg segment (explicit arguments start at offset 0) behind a
// region zero
typedef const MCX_KONST FusedArgs KArgs;
__device__ __forceinline__ KArgs& kargs_region(int z)
{
    return *(KArgs*)((const MCX_KONST char*)__builtin_amdgcn_kernarg_segment_ptr() + z);
}

// polynomial in the raw explanatory variable with wave-uniform coefficients (scalar loads); K == 3 is the default
// PolyomialRegression(degree=2) of the reference (controller.py:35)
template <int PPL>
__device__ __forceinline__ void lean_poly_add(const double* __restrict__ c, int K, const double (&x)[PPL], double (&p)[PPL])
{
    if (K == 3) {
        const double c0 = ldk(c), c1 = ldk(c + 1), c2 = ldk(c + 2);
#pragma unroll
        for (int q = 0; q < PPL; ++q) p[q] += fma(fma(c2, x[q], c1), x[q], c0);
    } else {
        double v[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) v[q] = 0.0;
#pragma unroll 1
        for (int k = K - 1; k >= 0; --k) {
            const double ck = ldk(c + k);
#pragma unroll
            for (int q = 0; q < PPL; ++q) v[q] = fma(v[q], x[q], ck);
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) p[q] += v[q];
    }
}

// LDS record area: shift[n_rec] | acc[4 waves][n_rec][2]   (layout of f_record, kf_common.h)
template <int PPL>
__device__ __forceinline__ void lean_record(const double (&v)[PPL], const bool (&live)[PPL], int rec, int n_rec, bool first_tile,
                                            double* __restrict__ lds)
{
    if (first_tile) {                      // block-uniform: the first path the block sees fixes the record's shift
        __syncthreads();
        if (threadIdx.x == 0) lds[rec] = v[0];
        __syncthreads();
    }
    const double c = lds[rec];
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const double d = live[q] ? v[q] - c : 0.0;
        d1 += d;
        d2 = fma(d, d, d2);
    }
    const double s1 = wave_sum(d1), s2 = wave_sum(d2);
    if ((threadIdx.x & 63) == 0) {
        double* acc = lds + n_rec + ((threadIdx.x >> 6) * n_rec + rec) * 2;
        acc[0] += s1;
        acc[1] += s2;
    }
}

template <int NSLOT, int SIG, int PPL, bool STORE>
__device__ __forceinline__ void lean_date(int t, const int64_t (&i)[PPL], const bool (&live)[PPL], bool first_tile,
                                          double* __restrict__ lds, const double (&reg)[PPL][2 * NSLOT], double (&cfs)[PPL], double (&cva)[PPL],
                                          int (&est)[PPL], const double* __restrict__ etab)
{
    // (state registers are indexed by wave-uniform record fields: M0-relative VGPR reads; mcx_fused_create binds an absent
    // reference to register 0 with a zero coefficient, so no range test is needed)
    const int zd = mcx_region_zero();                  // arguments, date record and exp coefficients: live in this block only
    KArgs& a = kargs_region(zd);
    const FastDate* __restrict__ fp = a.fast + t;
    const auto& k = a.k1;
    const mcx_expq_coef ec = mcx_expq_load(zd);       // exponentials: table of 2^(j/128) in LDS + degree-5 remainder
    // The CVA-only date (the config-3 shape: regression exposure, merged discount x survival factor, conditional default
    // probability — nothing stored, no cashflow consumer) in TWO rounds of scalar loads instead of one per branch of the general
    // program below: (1) the head of the record + the arguments, (2) the coefficient rows; and one round of LDS reads for its
    // two exponentials.  The waits of ~12 dependent scalar loads per date were a third of a wave's cycles at one or two waves
    // per SIMD.  Same arithmetic as the general path.
    // (the simulating kernel with two paths per lane at four waves per SIMD: measured slower — 52 SGPR spills around the date — than
    //  the general program below; the streaming kernel has no generator state to keep and is bound by the scalar unit, which the
    //  four SIMDs of a CU share: ~500 scalar instructions per date and wave of the general program cap it at ~4.3 TB/s)
    if ((PPL == 1 || !STORE) && !(STORE && k.paths)) {
        const FastDateHot hot = ldk_struct((const FastDateHot*)fp);
        const bool pure_cva = (hot.flags & (64 | 128 | 256 | 2)) == (64 | 2) && !((hot.flags & 1) && (a.cfs != nullptr || a.rec_pv[0] >= 0)) &&
                              a.expo == nullptr && a.n_basis == 3 && hot.c_b != 0.0;
        if (pure_cva) {
            const double* __restrict__ cf = a.coeffs;
            double c0[3] = {0.0, 0.0, 0.0}, c1[3] = {0.0, 0.0, 0.0};
            if (hot.coeff_off0 >= 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) c0[j] = ldk(cf + hot.coeff_off0 + j);
            }
            if (hot.coeff_off1 >= 0) {
#pragma unroll
                for (int j = 0; j < 3; ++j) c1[j] = ldk(cf + hot.coeff_off1 + j);
            }
            double xe[2 * PPL], ev[2 * PPL], p[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                xe[2 * q] = fma(hot.m_s1, reg[q][hot.s_reg], fma(hot.m_n1, reg[q][hot.ni_reg], hot.m_c0));
                xe[2 * q + 1] = fma(hot.c_c1, reg[q][hot.c_reg], hot.c_c0);
            }
            mcx_exp_tab_n<2 * PPL>(xe, ev, etab, ec);
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const double x = fma(hot.x_d, reg[q][hot.x_reg], hot.x_a);
                p[q] = 0.0;
                if (hot.coeff_off0 >= 0) p[q] += fma(fma(c0[2], x, c0[1]), x, c0[0]);
                if (hot.coeff_off1 >= 0) p[q] += fma(fma(c1[2], x, c1[1]), x, c1[0]);
                const double w = hot.m_b * ev[2 * q];
                const double cs = fma(hot.c_b, ev[2 * q + 1], hot.c_a);
                cva[q] = fma(fmax(p[q], 0.0), w * (1.0 - cs), cva[q]);
            }
            return;
        }
    }
    if (STORE && k.paths) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) if (live[q]) sim_store_state<NSLOT, SIG>(k, t, i[q], reg[q]);
    }
    const int flags = FD(flags);
    // cashflows feed the PV record / the cashflow output only; a CVA / exposure-profile run skips them
    const bool want_cash = (flags & 1) && (a.cfs != nullptr || a.rec_pv[0] >= 0);
    // CVA-only date without threshold: relu(p / N) S (1 - Sc) = relu(p) (S / N) (1 - Sc), one exponential for S / N
    const bool merged = (flags & 64) && !(flags & 128) && !want_cash && a.expo == nullptr;
    double inv[PPL];
    if (!merged) {
        if (flags & 16) {
            const double c = FD(ni_c0);
#pragma unroll
            for (int q = 0; q < PPL; ++q) inv[q] = c;
        } else {
            const double c0 = FD(ni_c0), c1 = FD(ni_c1);
            const int r = FD(ni_reg);
#pragma unroll
            for (int q = 0; q < PPL; ++q) inv[q] = mcx_exp_tab(fma(c1, reg[q][r], c0), etab, ec);
        }
    }
    if (want_cash) {
        const double k0 = FD(k0), k1 = FD(k1);
        const int lr = FD(lin_reg), n_exp = FD(n_exp);
        double val[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) val[q] = fma(k1, reg[q][lr], k0);
#pragma unroll 1
        for (int j = 0; j < n_exp; ++j) {
            const double w = FD(t_w[j]), c0 = FD(t_c0[j]), c1 = FD(t_c1[j]);
            const int r = FD(t_reg[j]);
#pragma unroll
            for (int q = 0; q < PPL; ++q) val[q] = fma(w, mcx_exp_tab(fma(c1, reg[q][r], c0), etab, ec), val[q]);
        }
        if (flags & 512) {                                 // plain option payoff (european_option.py:45-68)
            const double strike = FD(op_strike), sign = FD(op_sign);
#pragma unroll
            for (int q = 0; q < PPL; ++q) val[q] = fmax(sign * (val[q] - strike), 0.0);
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) cfs[q] = fma(val[q], inv[q], cfs[q]);
    }
    if (flags & 128) {
        // exercise event of the two-state product (bermudan_option.py:93-131): exercise iff the immediate value exceeds the
        // regression continuation value and a right is left; the (up to ~64) exponential terms of the immediate value come
        // through scalar loads, one 32-byte record per term, and serve both paths of the lane
        const double strike = FD(ex_strike), sign = FD(ex_sign);
        double val[PPL];
        // the whole value as ONE verified polynomial of the state variable (mcx_vpoly.hip: ~20 multiply-adds for both paths of the
        // lane instead of ~15 VALU per term and path); a wave that holds a path outside the verified range runs the terms
        bool collapsed = false;
        if (flags & 1024) {
            const double lo = FD(ex_p_lo), hi = FD(ex_p_hi);
            const int pr = FD(ex_p_reg);
            bool in = true;
#pragma unroll
            for (int q = 0; q < PPL; ++q) in = in && reg[q][pr] >= lo && reg[q][pr] <= hi;
            if (__all(in)) {
                const double ms = FD(ex_p_ms), ih = FD(ex_p_ih);
                const int nb = FD(ex_p_blk);
                const VPolyBlk* __restrict__ blk = (const VPolyBlk*)(a.vcoef + FD(ex_p_off));
                double tt[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) { tt[q] = fma(reg[q][pr], ih, ms); val[q] = 0.0; }
                VPolyBlk c = ldk_struct(blk);
#pragma unroll 1
                for (int kb = 0; kb < nb; ++kb) {
                    const VPolyBlk nx = ldk_struct(blk + kb + 1);
#pragma unroll
                    for (int j = 0; j < MCX_VPOLY_BLK; ++j)
#pragma unroll
                        for (int q = 0; q < PPL; ++q) val[q] = fma(val[q], tt[q], c.c[j]);
                    c = nx;
                }
                collapsed = true;
            }
        }
        if (!collapsed) {
            const double k0 = FD(ex_k0), k1 = FD(ex_k1);
            const int lr = FD(ex_lin_reg), n_t = FD(ex_n);
            const LeanTerm* __restrict__ lt = a.lterms + FD(ex_term_off);
#pragma unroll
            for (int q = 0; q < PPL; ++q) val[q] = fma(k1, reg[q][lr], k0);
            // the next term's record is in flight while this term's two exponentials run (the table has one spare entry at its end)
            LeanTerm tm = ldk_struct(lt);
#pragma unroll 1
            for (int j = 0; j < n_t; ++j) {
                const LeanTerm nx = ldk_struct(lt + j + 1);
#pragma unroll
                for (int q = 0; q < PPL; ++q) val[q] = fma(tm.w, mcx_exp_tab(fma(tm.c1, reg[q][tm.reg], tm.c0), etab, ec), val[q]);
                tm = nx;
            }
        }
        const double xa = FD(ex_x_a), xd = FD(ex_x_d);
        const int xr = FD(ex_x_reg), co = FD(ex_coeff_off);
        double x[PPL], cont[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) { x[q] = fma(xd, reg[q][xr], xa); cont[q] = 0.0; }
        if (co >= 0) lean_poly_add<PPL>(a.coeffs + co + a.n_basis, a.n_basis, x, cont);      // row of state 1 (a right is left)
        const bool want_ex_cash = a.cfs != nullptr || a.rec_pv[0] >= 0;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const double imm = fmax(sign * (val[q] - strike), 0.0);
            const bool ex = (imm > cont[q]) && (est[q] > 0);
            if (want_ex_cash) cfs[q] = ex ? fma(imm, inv[q], cfs[q]) : cfs[q];
            est[q] = ex ? est[q] - 1 : est[q];
        }
    }
    double p[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) p[q] = 0.0;
    if (flags & 2) {
        const double xa = FD(x_a), xd = FD(x_d);
        const int xr = FD(x_reg), off0 = FD(coeff_off0), off1 = FD(coeff_off1);
        double x[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) x[q] = fma(xd, reg[q][xr], xa);
        if (flags & 256) {
            // exposure of the exercise product: the coefficient row of the lane's state (product.py:150-184)
            double p0[PPL], p1[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) { p0[q] = 0.0; p1[q] = 0.0; }
            lean_poly_add<PPL>(a.coeffs + off0, a.n_basis, x, p0);
            lean_poly_add<PPL>(a.coeffs + off0 + a.n_basis, a.n_basis, x, p1);
#pragma unroll
            for (int q = 0; q < PPL; ++q) p[q] = est[q] > 0 ? p1[q] : p0[q];
        } else {
            if (off0 >= 0) lean_poly_add<PPL>(a.coeffs + off0, a.n_basis, x, p);
            if (off1 >= 0) lean_poly_add<PPL>(a.coeffs + off1, a.n_basis, x, p);
        }
    }
    // survival probability over the next interval, conditional on the credit state (cva_metric.py:66-89)
    auto cond_surv = [&](double (&cs)[PPL]) {
        const double ca = FD(c_a), cb = FD(c_b);
#pragma unroll
        for (int q = 0; q < PPL; ++q) cs[q] = ca;
        if (cb != 0.0) {
            const double cc0 = FD(c_c0), cc1 = FD(c_c1);
            const int cr = FD(c_reg);
#pragma unroll
            for (int q = 0; q < PPL; ++q) cs[q] = fma(cb, mcx_exp_tab(fma(cc1, reg[q][cr], cc0), etab, ec), ca);
        }
    };
    if (merged) {
        const double mb = FD(m_b), mc0 = FD(m_c0), mn1 = FD(m_n1), ms1 = FD(m_s1);
        const int nr = FD(ni_reg), sr = FD(s_reg);
        double w[PPL], cs[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) w[q] = mb * mcx_exp_tab(fma(ms1, reg[q][sr], fma(mn1, reg[q][nr], mc0)), etab, ec);
        cond_surv(cs);
#pragma unroll
        for (int q = 0; q < PPL; ++q) cva[q] = fma(fmax(p[q], 0.0), w[q] * (1.0 - cs[q]), cva[q]);
        return;
    }
    double e[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) e[q] = p[q] * inv[q];
    if (a.expo) {
        const int row = ldk(a.date_row + t);
        if (row >= 0) {
#pragma unroll
            for (int q = 0; q < PPL; ++q) if (live[q]) a.expo[(int64_t)row * a.ld_out + i[q]] = e[q];
        }
    }
    if (flags & 32) {
        const double thr = FD(thr);
        double u[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) u[q] = dev_thr(e[q], thr);
        if (flags & 8) {
            const int rp = FD(rec_profile);
            double up[PPL], un[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) { up[q] = fmax(u[q], 0.0); un[q] = fmin(u[q], 0.0); }
            lean_record<PPL>(up, live, rp, a.n_rec, first_tile, lds);
            lean_record<PPL>(un, live, rp + 1, a.n_rec, first_tile, lds);
        }
        if (flags & 4) {
            const double sb = FD(s_b), sc0 = FD(s_c0), sc1 = FD(s_c1);
            const int sr = FD(s_reg);
            double sp[PPL], cs[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) sp[q] = sb * mcx_exp_tab(fma(sc1, reg[q][sr], sc0), etab, ec);
            cond_surv(cs);
#pragma unroll
            for (int q = 0; q < PPL; ++q) cva[q] = fma(fmax(u[q], 0.0), sp[q] * (1.0 - cs[q]), cva[q]);
        }
    }
}

#define MCX_LEAN_WAVES 4
// SIMULATE = false: the same date programs on a paths tensor produced earlier by K1 (k1.paths is then the INPUT
// [date][state][path]): one streaming pass, the next date's state columns in flight while this date's program runs
// (Measured and dropped: drawing the normals of 2-5 sub-steps AHEAD as independent staged chains, for small path counts — the draws
// depend on the counter (path, step) only.  No gain at one or two waves per SIMD: what a thin launch waits for is scalar work, see
// launch_lean.)
template <int NSLOT, int NZ, bool INJECT, int SIG, int PPL, bool SIMULATE>
__global__ __launch_bounds__(MCX_BLOCK, MCX_LEAN_WAVES) void kf_lean(const FusedArgs)      // read through kargs_region(), never by name
{
    constexpr int NREG = 2 * NSLOT;
    constexpr int TILE = MCX_BLOCK * PPL;
    extern __shared__ double lds[];
    KArgs& a0 = kargs_region(0);
    const int n_rec = a0.n_rec;
    const int64_t n = a0.k1.n;
    for (int q = threadIdx.x; q < 9 * n_rec; q += MCX_BLOCK) lds[q] = 0.0;
    // 1024-entry Box-Muller tables (32 KiB of LDS) where the sub-steps dominate and four blocks per CU still fit: the two-factor
    // rates / credit signature; 128 entries elsewhere (books with hundreds of per-date records need the LDS for those)
    constexpr int BMB = SIG == SIG_VAS_CIR_E ? 10 : 7;
    constexpr int BM = (INJECT || !SIMULATE) ? 0 : MCX_BM_LDS_DOUBLES_B(BMB);
    __shared__ double tab_lds[BM + MCX_EXP_LDS_DOUBLES];            // Box-Muller lookup tables | exp table
    const double* tab = nullptr;
    const double* etab = tab_lds + BM;
    mcx_exp_tab_load(tab_lds + BM);
    if (BM) { mcx_bm_load<BMB>(tab_lds); tab = tab_lds; }
    __syncthreads();
    const int64_t tiles = (n + TILE - 1) / TILE;
    double n_block = 0.0;

    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const bool first_tile = tile == (int64_t)blockIdx.x;
        int64_t i[PPL];
        bool live[PPL];
        uint64_t path[PPL];
        double reg[PPL][NREG];                         // reg[q][2s], reg[q][2s+1] = state of slot s of the lane's q-th path
        double cfs[PPL], cva[PPL];
        int est[PPL];                                  // rights left of the book's (at most one) exercise product
        int n_init, n_steps;
        {
            KArgs& a = kargs_region(mcx_region_zero());          // tile prologue
            const auto& k = a.k1;
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                const int64_t i_raw = tile * TILE + q * MCX_BLOCK + threadIdx.x;
                live[q] = i_raw < n;
                i[q] = live[q] ? i_raw : n - 1;        // dead lanes shadow the last path (no stores, no contribution)
                path[q] = k.path_offset + (uint64_t)i[q];
                sim_init_state<NSLOT, SIG>(k, reg[q]);
                cfs[q] = 0.0; cva[q] = 0.0;
                est[q] = a.init_state[0];
            }
            const int64_t rest = n - tile * TILE;
            n_block += (double)(rest < TILE ? rest : TILE);
            n_init = k.n_initial_store; n_steps = k.n_steps;
        }
        if (SIMULATE) {
        // ONE date call site: the dates that hold the initial state come first (their sub-step run is empty), then
        // alternately a run of sub-steps up to the next timeline date and that date's program
        int step = 0, t_init = 0;
#pragma unroll 1
        while (true) {
            const bool init = t_init < n_init;
            if (!init && step >= n_steps) break;
            int st = init ? t_init : -1;
            t_init += init ? 1 : 0;
            {
                const int zr0 = mcx_region_zero();      // arguments, Philox key schedule, Box-Muller coefficients of this run
                const auto& k = kargs_region(zr0).k1;
                const uint64_t seed = k.seed;
                const mcx_bm_coef bc = mcx_bm_coef_load(zr0);
                const mcx_bm_vconst vc = mcx_bm_vconst_make<BMB>(bc);  // constants kept in registers across the run of sub-steps
#pragma unroll 1
                while (st < 0 && step < n_steps) {
                    if constexpr (SIG == SIG_GENERIC) {
                        // run-time model dispatch (every aux entry of every slot would have to be loaded ahead): path after path
#pragma unroll
                        for (int q = 0; q < PPL; ++q) sim_substep<NSLOT, NZ, INJECT, SIG, BMB, true>(k, step, path[q], i[q], reg[q], tab, seed, bc, &vc);   // POS: mcx_fused_create
                        st = ldk(&k.steps[step].store_idx);
                    } else {
                        // draws of all the lane's paths in ONE basic block (unguarded root, pair_from_words), one rare branch for the
                        // 2^-32 draws that may round to u = 1, then the state updates
                        double zz[PPL][NZ], uu[PPL];
                        bool rare = false;
                        // the scalar loads of this sub-step go out first: the draws below cover their latency
                        const StepData<NSLOT, NZ> sdat = sim_step_load<NSLOT, NZ, SIG>(k, step);
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (INJECT) {
#pragma unroll
                            for (int q = 0; q < PPL; ++q) sim_draw<NZ, true, SIG, BMB>(k, step, path[q], i[q], zz[q], uu[q], tab, seed, bc, &vc);
                        } else {
                            uint32_t st_q[PPL];
#pragma unroll
                            for (int q = 0; q < PPL; ++q) st_q[q] = (uint32_t)step;
                            rare = sim_draw_n<PPL, NZ, SIG, BMB>(k, st_q, path, zz, uu, tab, seed, bc, vc);
                        }
                        if (!INJECT && __builtin_expect(__any(rare), 0)) {
#pragma unroll
                            for (int q = 0; q < PPL; ++q) sim_draw<NZ, false, SIG, BMB, true>(k, step, path[q], i[q], zz[q], uu[q], tab, seed, bc, &vc);
                        }
#pragma unroll
                        for (int q = 0; q < PPL; ++q) st = sim_apply_loaded<NSLOT, NZ, SIG, true>(k, sdat, reg[q], zz[q], uu[q]);   // POS: mcx_fused_create
                    }
                    ++step;
                }
            }
            if (st >= 0) lean_date<NSLOT, SIG, PPL, true>(st, i, live, first_tile, lds, reg, cfs, cva, est, etab);
        }
        } else {
            auto load_row = [&](int t, double (&dst)[PPL][NREG]) {
                const auto& k = kargs_region(0).k1;
                const int D = k.n_state;
#pragma unroll
                for (int q = 0; q < PPL; ++q)
#pragma unroll
                    for (int sl = 0; sl < NSLOT; ++sl) {
                        const bool bs = sig_kind(SIG, sl) >= 0 ? sig_is_bs(SIG, sl) : (k.slots[sl].kind == MCX_MODEL_BS);
                        const int c = k.slots[sl].state_off;
                        dst[q][2 * sl] = k.paths[((int64_t)t * D + c) * k.ld + i[q]];
                        dst[q][2 * sl + 1] = bs ? 0.0 : k.paths[((int64_t)t * D + c + 1) * k.ld + i[q]];
                    }
            };
            const int n_dates = kargs_region(0).n_dates;
            // the pass is a pure stream: two state buffers, the date loop unrolled by two so that each buffer is consumed where its
            // loads landed (a single call site with a rotating third buffer cost 32 register copies per path and date — a fifth of the
            // VALU work of the pass); the loads of date t+2 go out as soon as date t is done and have the program of date t+1 to land
            double bufb[PPL][NREG];
            load_row(0, reg);
            load_row(n_dates > 1 ? 1 : 0, bufb);
#pragma unroll 1
            for (int t = 0; t < n_dates; t += 2) {
                lean_date<NSLOT, SIG, PPL, false>(t, i, live, first_tile, lds, reg, cfs, cva, est, etab);
                if (t + 2 < n_dates) load_row(t + 2, reg);
                if (t + 1 < n_dates) {
                    lean_date<NSLOT, SIG, PPL, false>(t + 1, i, live, first_tile, lds, bufb, cfs, cva, est, etab);
                    if (t + 3 < n_dates) load_row(t + 3, bufb);
                }
            }
        }
        {
            KArgs& a = kargs_region(mcx_region_zero());          // tile epilogue: per-path quantities
#pragma unroll
            for (int q = 0; q < PPL; ++q) if (a.cfs && live[q]) a.cfs[i[q]] = cfs[q];
            if (a.rec_pv[0] >= 0) lean_record<PPL>(cfs, live, a.rec_pv[0], n_rec, first_tile, lds);
            if (a.rec_cva[0] >= 0) {
                const double lgd = a.lgd[0];
                double cc[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) cc[q] = cva[q] * lgd;
                lean_record<PPL>(cc, live, a.rec_cva[0], n_rec, first_tile, lds);
            }
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n_rec; r += MCX_BLOCK) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < 4; ++w) { s1 += lds[n_rec + (w * n_rec + r) * 2]; s2 += lds[n_rec + (w * n_rec + r) * 2 + 1]; }
        double* dst = a0.partials + ((int64_t)blockIdx.x * n_rec + r) * 4;
        dst[0] = n_block; dst[1] = lds[r]; dst[2] = s1; dst[3] = s2;
    }
}

#undef FD

// launch of one shape (paths per lane) of the kernel; returns the grid
template <int NSLOT, int NZ, int SIG, int PPL>
int launch_lean_shape(const FusedArgs& a, int n_cu, bool inject, bool simulate, hipStream_t s)
{
    const int64_t tiles = (a.k1.n + MCX_BLOCK * PPL - 1) / (MCX_BLOCK * PPL);
    const size_t lds = sizeof(double) * (size_t)((9 * a.n_rec + 1) & ~1);
    // blocks that are really co-resident: __launch_bounds__(256, 4) guarantees the registers of 4 blocks per CU, but a book with
    // hundreds of records (one per metric date) adds dynamic LDS to the 34 KiB of tables and may leave room for 3 only; a grid
    // sized for 4 would then run its last quarter as a tail at a third of the occupancy
    auto residency = [&](auto kernel) {
        thread_local std::map<std::pair<const void*, size_t>, int> cache;      // (the query costs microseconds: once per kernel and size)
        const auto key = std::make_pair((const void*)kernel, lds);
        auto it = cache.find(key);
        if (it == cache.end()) {
            int per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, MCX_BLOCK, lds) != hipSuccess || per_cu < 1) per_cu = 1;
            it = cache.emplace(key, per_cu < MCX_LEAN_WAVES ? per_cu : MCX_LEAN_WAVES).first;
        }
        return (int64_t)it->second * n_cu;
    };
    auto sized = [&](int64_t resident) {
        if (tiles <= resident) return (int)tiles;
        const int64_t per = (tiles + resident - 1) / resident;  // equal number of tiles per block whenever the count divides
        return (int)((tiles + per - 1) / per);
    };
    int grid;
    if (!simulate) {
        auto kern = kf_lean<NSLOT, NZ, false, SIG, PPL, false>;
        grid = sized(residency(kern));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MCX_BLOCK), lds, s, a);
    } else if (inject) {
        auto kern = kf_lean<NSLOT, NZ, true, SIG, PPL, true>;
        grid = sized(residency(kern));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MCX_BLOCK), lds, s, a);
    } else {
        auto kern = kf_lean<NSLOT, NZ, false, SIG, PPL, true>;
        grid = sized(residency(kern));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MCX_BLOCK), lds, s, a);
    }
    return grid;
}

template <int NSLOT, int NZ, int SIG>
void launch_lean(const FusedArgs& a, int n_cu, bool inject, bool simulate, hipStream_t s, int* grid_out)
{
    // Full shape: two paths per lane where both fit the 128-VGPR budget of 4 waves per SIMD (the generic run-time-dispatch kernels
    // of several sub-models carry every model's step code and would spill: one path per lane).  512 paths per block-tile, 4 blocks
    // per CU: 2^19 paths fill the chip once.
    constexpr int PPL = (SIG == SIG_GENERIC && NSLOT >= 2) ? 1 : 2;
#ifdef MCX_LEAN_AB
    // tools/build_variants.sh only (never defined in the product build): the shape picked by the environment, 10 * PPL + 1
    if (simulate && !inject) {
        const char* sh = getenv("MCX_LEAN_SHAPE");
        const int code = sh ? atoi(sh) : 0;
        if (code == 11) { *grid_out = launch_lean_shape<NSLOT, NZ, SIG, 1>(a, n_cu, inject, simulate, s); return; }
        if (code == 21) { *grid_out = launch_lean_shape<NSLOT, NZ, SIG, 2>(a, n_cu, inject, simulate, s); return; }
    }
#endif
    // Small path counts (one GPU's share of a strong-scaled run: 2^20 paths over 8 GPUs = 2^17 each): the full shape would put one
    // wave on a SIMD.  A wave of this kernel spends a third of its cycles on wave-uniform work — scalar loads of the step and date
    // records and their waits, SALU control (SQ counters, profiles/README.md) — which a second wave on the SIMD overlaps with its
    // own VALU work and a single wave cannot: below half a chip-filling launch one path per lane (twice the waves) is faster,
    // 0.176 against 0.195 ms at 131,072 paths, equal at 262,144, slower from there on (the scalar work per path doubles).
    const int64_t full_tiles = (a.k1.n + MCX_BLOCK * PPL - 1) / (MCX_BLOCK * PPL);
    if (PPL == 2 && simulate && !inject && full_tiles < (int64_t)2 * n_cu) {
        *grid_out = launch_lean_shape<NSLOT, NZ, SIG, 1>(a, n_cu, inject, simulate, s);
        return;
    }
    *grid_out = launch_lean_shape<NSLOT, NZ, SIG, PPL>(a, n_cu, inject, simulate, s);
}

}  // namespace

// host entry used by kf_fused.hip (fused_run_impl); returns the grid size (number of per-block partial records), or -1 when
// the (slots, z) shape has no instantiation
int mcx_launch_kf_lean(const FusedArgs& a, const mcx_sim_desc& sd, int n_cu, bool inject, bool simulate, hipStream_t s)
{
    int grid = -1;
    switch (mcx_sim_signature(sd)) {
    case SIG_VAS_CIR_E: launch_lean<2, 2, SIG_VAS_CIR_E>(a, n_cu, inject, simulate, s, &grid); break;
#ifndef MCX_LEAN_ONE_SIG
    case SIG_BS_A: launch_lean<1, 1, SIG_BS_A>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_BS_E: launch_lean<1, 1, SIG_BS_E>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_HESTON_QE: launch_lean<1, 2, SIG_HESTON_QE>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_HESTON_E: launch_lean<1, 2, SIG_HESTON_E>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_VAS_E: launch_lean<1, 1, SIG_VAS_E>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_VAS_A: launch_lean<1, 1, SIG_VAS_A>(a, n_cu, inject, simulate, s, &grid); break;
    case SIG_BS_VAS_CIRDET_E: launch_lean<3, 3, SIG_BS_VAS_CIRDET_E>(a, n_cu, inject, simulate, s, &grid); break;
    default:
        switch (sd.n_slots * 16 + sd.n_z) {
        case 1 * 16 + 1: launch_lean<1, 1, SIG_GENERIC>(a, n_cu, inject, simulate, s, &grid); break;
        case 1 * 16 + 2: launch_lean<1, 2, SIG_GENERIC>(a, n_cu, inject, simulate, s, &grid); break;
        case 2 * 16 + 2: launch_lean<2, 2, SIG_GENERIC>(a, n_cu, inject, simulate, s, &grid); break;
        case 3 * 16 + 3: launch_lean<3, 3, SIG_GENERIC>(a, n_cu, inject, simulate, s, &grid); break;
        case 4 * 16 + 4: launch_lean<4, 4, SIG_GENERIC>(a, n_cu, inject, simulate, s, &grid); break;
        default: break;
        }
#else
    default: break;
#endif
    }
    return grid;
}
