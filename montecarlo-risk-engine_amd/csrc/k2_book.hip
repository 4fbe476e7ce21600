// k2_book.hip — K2: fused request resolution + cashflows + exposures (one pass over the paths tensor).
//
// Replaces, per path, RequestInterface.resolve_requests (request_interface.py:115-130: ~2.4 GB of materialised [N]
// vectors at 1 M paths), Product.compute_normalized_cashflows / get_value (bond.py, swap.py, european_option.py,
// bermudan_option.py) and the exposure evaluation `A @ coeffs / numeraire` of controller.py:385-471, summed per netting
// set (controller.py:584-591).  Market quantities are "atoms" evaluated in registers from the simulated state; nothing
// is materialised except the outputs the Metrics API consumes: cfs [NS][N] and exposures [NS][E][N].
//
// One lane = one path; the event program (products -> events -> terms) is wave-uniform, so it is walked with scalar
// loads and scalar branches; all HBM accesses are 512-byte coalesced rows of the [date][state][path] layout.
#include <algorithm>

#include "mcx_device.h"

namespace {

struct K2Args {
    const DevTerm* __restrict__ terms;
    const DevEvent* __restrict__ events;
    const DevProduct* __restrict__ products;
    const DevAtom* __restrict__ atoms;
    const double* __restrict__ coeffs;
    const double* __restrict__ paths;
    double* __restrict__ cfs;
    double* __restrict__ expo;
    int64_t n, ld, ld_out;
    int32_t n_products, n_basis, n_state, n_expo_rows, want_cfs, want_expo;
    // product-chunked mode (big books on few paths): blockIdx.y = chunk of `chunk_products` consecutive products, each chunk
    // ACCUMULATES into its own zero-initialised [n_ns][*][ld_out] image (cfs / expo then point at the chunk images)
    int32_t chunk_products, n_netting_sets;
    const DevBridge* __restrict__ bridge;       // RNG state of Brownian-bridge barrier events (device struct of the book)
    uint8_t* __restrict__ ex_bits;              // exercise-decision record / replay (mcx_book_set_exercise_replay)
    int64_t ex_ld;
    int32_t ex_mode;
    const DevVPoly* __restrict__ vpoly;         // value polynomials of the book's events (DevEvent::pad, mcx_vpoly.hip) or nullptr
    const double* __restrict__ vcoef;
};

__device__ __forceinline__ double dev_poly(const double* __restrict__ c, int K, double x)
{
    double v = 0.0, xp = 1.0;
    for (int k = 0; k < K; ++k) { v = fma(c[k], xp, v); xp *= x; }
    return v;
}

__device__ __forceinline__ double dev_norm_cdf(double x) { return 0.5 * (1.0 + erf(x * 0.70710678118654752440)); }

// normalised cashflow of one product-date event for a path in exercise state s (s may be decremented)
__device__ __forceinline__ double dev_cash_event(const DevEvent& e, const DevTerm* __restrict__ terms, const DevAtom* __restrict__ atoms,
                                                 const double* __restrict__ coeffs, int K, const double* __restrict__ paths,
                                                 int64_t D, int64_t ld, int64_t i, int& s, const DevBridge* __restrict__ bridge,
                                                 int ex_mode = 0, uint8_t* __restrict__ ex_cell = nullptr,
                                                 const DevVPoly* __restrict__ vpoly = nullptr, const double* __restrict__ vcoef = nullptr)
{
    const double num = dev_atom(e.num, paths, D, ld, i);
    double common = 0.0, own = 0.0, glog = 0.0;
    if (e.kind == MCX_EV_OPTION && (e.aux[0] == 4.0 || e.aux[0] == 5.0))      // barrier options (barrier_option.py:60-223)
        return dev_barrier_event(e, terms, coeffs, bridge, paths, D, ld, i, num);
    if (e.kind == MCX_EV_OPTION && e.aux[0] == 3.0) {                     // binary payoff (binary_option.py:38-43): fuzzy indicator
        double val = 0.0;
        AtomCache bc = {-1, -1, 0.0};
        for (int j = e.term_begin; j < e.term_end; ++j) {
            const DevTerm tm = ldk_struct(&terms[j]);
            val = fma(tm.w, dev_atom_cached(tm.atom, paths, D, ld, i, bc), val);
        }
        const double dot = fmin(fmax((val - e.strike + e.aux[2]) / (2.0 * e.aux[2]), 0.0), 1.0);
        return e.aux[1] * (e.sign > 0.0 ? dot : 1.0 - dot) / num;
    }
    const bool basket = e.kind == MCX_EV_OPTION && e.aux[0] != 0.0;       // geometric aggregate needed (basket_option.py:56-82)
    AtomCache ac = {-1, -1, 0.0};
    int j_begin = e.term_begin;
    if (e.pad > 0 && vpoly) {                                 // the event's value as one verified polynomial (mcx_vpoly.hip)
        const DevVPoly vp = ldk_struct(&vpoly[e.pad - 1]);
        const double xv = paths[((int64_t)vp.t_idx * D + vp.col) * ld + i];
        if (__all(xv >= vp.lo && xv <= vp.hi)) { common = dev_vpoly(vp, vcoef, xv); j_begin = e.term_end; }
    }
    for (int j = j_begin; j < e.term_end; ++j) {
        const DevTerm tm = ldk_struct(&terms[j]);
        const double av = dev_atom_cached(tm.atom, paths, D, ld, i, ac);
        const double v = tm.w * av;
        if (basket) glog = fma(tm.w, mcx_log(av + 1e-10), glog);
        if (tm.den < 0) common += v;
        else own += v / dev_atom(ldk_struct(&atoms[tm.den]), paths, D, ld, i);
    }
    if (e.kind == MCX_EV_CASHFLOW) return common / num + own;
    const double imm = fmax(e.sign * (common - e.strike), 0.0);
    if (basket) {
        const double geo = fmax(e.sign * (mcx_exp(glog) - e.strike), 0.0);
        return (e.aux[0] == 1.0 ? geo : imm - geo + e.aux[1]) / num;
    }
    if (e.kind == MCX_EV_OPTION) return imm / num;
    double cont = 0.0;                                        // MCX_EV_EXERCISE (bermudan_option.py:93-131)
    double cont_ex = 0.0;                                     // continuation AFTER exercising (flexicall.py:118-133), aux[0] = 1
    if (e.coeff_off >= 0) {
        const double x = dev_atom(e.x, paths, D, ld, i);
        cont = dev_poly(coeffs + e.coeff_off + s * K, K, x);
        if (e.aux[0] == 1.0 && s > 0) cont_ex = dev_poly(coeffs + e.coeff_off + (s - 1) * K, K, x);
    }
    const bool ex = dev_exercise_decision((imm + cont_ex > cont) && (s > 0), s, ex_mode, ex_cell, 0);
    if (ex) s -= 1;
    return ex ? imm / num : 0.0;
}

__global__ __launch_bounds__(MCX_BLOCK) void k2_eval_book(const K2Args a)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const int64_t D = a.n_state, ld = a.ld;
    const int K = a.n_basis;
    int cur_ns = -1;
    double acc_ns = 0.0;
    const bool chunked = a.chunk_products > 0;
    const int p_begin = chunked ? (int)blockIdx.y * a.chunk_products : 0;
    const int p_end = chunked ? min(p_begin + a.chunk_products, a.n_products) : a.n_products;
    double* __restrict__ cfs_out = a.cfs;
    double* __restrict__ expo_out = a.expo;
    if (chunked) {
        if (cfs_out) cfs_out += (int64_t)blockIdx.y * a.n_netting_sets * a.ld_out;
        if (expo_out) expo_out += (int64_t)blockIdx.y * a.n_netting_sets * a.n_expo_rows * a.ld_out;
    }
    for (int p = p_begin; p < p_end; ++p) {
        const DevProduct pr = ldk_struct(&a.products[p]);
        if (pr.ev_end == pr.ev_begin) continue;              // analytically valued product: no Monte-Carlo events
        if (pr.netting_set != cur_ns) {
            if (cur_ns >= 0 && a.want_cfs) cfs_out[(int64_t)cur_ns * a.ld_out + i] = acc_ns;
            cur_ns = pr.netting_set;
            acc_ns = 0.0;
        }
        int s = pr.init_state;
        double acc = 0.0;
        for (int q = pr.ev_begin; q < pr.ev_end; ++q) {
            const DevEvent e = ldk_struct(&a.events[q]);
            if (e.kind <= MCX_EV_EXERCISE) {
                // a cashflow / payoff of a product without exercise states feeds the cashflow output alone: nothing to do when no
                // metric asks for discounted cashflows (a CVA / exposure run: the payments of a swap were 150 of its 151 exponentials)
                if (!a.want_cfs && pr.n_states == 1 && e.kind != MCX_EV_EXERCISE) continue;
                acc += dev_cash_event(e, a.terms, a.atoms, a.coeffs, K, a.paths, D, ld, i, s, a.bridge, a.ex_mode,
                                      a.ex_mode ? a.ex_bits + (int64_t)q * a.ex_ld + i : nullptr, a.vpoly, a.vcoef);
            } else {
                double v = 0.0;
                if (e.kind == MCX_EV_EXPO_POLY) {
                    const double num = dev_atom(e.num, a.paths, D, ld, i);
                    const double x = dev_atom(e.x, a.paths, D, ld, i);
                    const double cont = e.coeff_off >= 0 ? dev_poly(a.coeffs + e.coeff_off + s * K, K, x) : 0.0;
                    v = cont / num;
                } else if (e.aux[2] > 0.0) {                  // MCX_EV_EXPO_BS (european_option.py:88-145)
                    const double num = dev_atom(e.num, a.paths, D, ld, i);
                    const double spot = dev_atom(e.x, a.paths, D, ld, i);
                    const double sig = e.aux[0], rate = e.aux[1], tau = e.aux[2], Kx = e.strike;
                    const double sq = sqrt(tau);
                    const double d1 = (log(spot / Kx) + (rate + 0.5 * sig * sig) * tau) / (sig * sq);
                    const double d2 = d1 - sig * sq;
                    const double df = exp(-rate * tau);
                    const double price = e.sign > 0.0 ? spot * dev_norm_cdf(d1) - Kx * df * dev_norm_cdf(d2)
                                                      : Kx * df * dev_norm_cdf(-d2) - spot * dev_norm_cdf(-d1);
                    v = price / num;
                }
                double* dst = expo_out + ((int64_t)pr.netting_set * a.n_expo_rows + e.row) * a.ld_out + i;
                if ((e.flags & 1) || chunked) *dst += v;
                else *dst = v;
            }
        }
        acc_ns += acc;
    }
    if (cur_ns >= 0 && a.want_cfs) cfs_out[(int64_t)cur_ns * a.ld_out + i] = acc_ns;
}

// ---- PPL paths per lane ---------------------------------------------------------------------------------------------------
// The one-path-per-lane kernel above is latency-bound at large path counts: a lane walks the event program through a chain of
// scalar record loads and dependent 8-byte global loads with 4 resident waves per SIMD.  Here a lane carries PPL paths through
// the same (wave-uniform) program: PPL independent loads in flight per event, every record load and branch paid once per
// PPL x 64 paths.  Books with barrier events (per-path bridge RNG) and the product-chunked mode stay on the kernel above.
// FEAT: the event families the book contains (host-side scan); everything else compiles out, which is what keeps the common
// books (cashflows, plain options, polynomial exposures) at 8 resident waves per SIMD.
enum { K2F_DEN = 1, K2F_EXOTIC = 2, K2F_EXERCISE = 4, K2F_BS_EXPO = 8, K2F_ALL = 15 };
template <int PPL, int FEAT>
__global__ __launch_bounds__(MCX_BLOCK) void k2_eval_book_v(const K2Args a)
{
    int64_t i[PPL];
    bool live[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int64_t i_raw = ((int64_t)blockIdx.x * PPL + q) * MCX_BLOCK + threadIdx.x;
        live[q] = i_raw < a.n;
        i[q] = live[q] ? i_raw : a.n - 1;
    }
    __shared__ double etab[MCX_EXP_LDS_DOUBLES];         // 2^(j/128): the exponentials of the atoms (mcx_exp_tab)
    mcx_exp_tab_load(etab);
    __syncthreads();
    if (!live[0]) return;
    const mcx_expq_coef ec = mcx_expq_load();
    const int64_t D = a.n_state, ld = a.ld;
    const int K = a.n_basis;
    int cur_ns = -1;
    double acc_ns[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) acc_ns[q] = 0.0;
    for (int p = 0; p < a.n_products; ++p) {
        const DevProduct pr = ldk_struct(&a.products[p]);
        if (pr.ev_end == pr.ev_begin) continue;
        if (pr.netting_set != cur_ns) {
            if (cur_ns >= 0 && a.want_cfs) {
#pragma unroll
                for (int q = 0; q < PPL; ++q) if (live[q]) a.cfs[(int64_t)cur_ns * a.ld_out + i[q]] = acc_ns[q];
            }
            cur_ns = pr.netting_set;
#pragma unroll
            for (int q = 0; q < PPL; ++q) acc_ns[q] = 0.0;
        }
        int s[PPL];
        double acc[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) { s[q] = pr.init_state; acc[q] = 0.0; }
        double num[PPL];                                // 1 / numeraire of the event: the divisions below become multiplications
#pragma unroll
        for (int q = 0; q < PPL; ++q) num[q] = 0.0;
        for (int e_i = pr.ev_begin; e_i < pr.ev_end; ++e_i) {
            const DevEvent e = ldk_struct(&a.events[e_i]);
            // flags bit 1: same numeraire atom as the previous event of the product (cashflow and exposure of one date): kept
            if (!(e.flags & 2) && (e.kind != MCX_EV_EXPO_BS || e.aux[2] > 0.0)) {
                if (e.num.a == 0.0 && e.num.d == 0.0 && e.num.b == 1.0 && e.num.col >= 0) {
                    // a pure exponential (the money-market account exp(log B)): its reciprocal is the exponential of the negated
                    // argument — no reciprocal sequence
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const double x = a.paths[((int64_t)e.num.t_idx * D + e.num.col) * ld + i[q]];
                        num[q] = mcx_exp_tab(-fma(e.num.c1, x, e.num.c0), etab, ec);
                    }
                } else {
                    dev_atoms<PPL>(e.num, a.paths, D, ld, i, num, etab, ec);
#pragma unroll
                    for (int q = 0; q < PPL; ++q) num[q] = mcx_rcp(num[q]);
                }
            }
            if (e.kind <= MCX_EV_EXERCISE) {
                // (as in the one-path kernel: a stateless product's cash event is dead without a cashflow output; its numeraire
                //  stays evaluated above — the exposure event of the same date re-uses it, flags bit 1)
                if (!a.want_cfs && pr.n_states == 1 && e.kind != MCX_EV_EXERCISE) continue;
                double common[PPL], own[PPL], glog[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) { common[q] = 0.0; own[q] = 0.0; glog[q] = 0.0; }
                const bool binary = (FEAT & K2F_EXOTIC) && e.kind == MCX_EV_OPTION && e.aux[0] == 3.0;
                const bool basket = (FEAT & K2F_EXOTIC) && e.kind == MCX_EV_OPTION && e.aux[0] != 0.0 && !binary;
                int j_begin = e.term_begin;
                if (e.pad > 0 && a.vpoly) {                 // the event's value as one verified polynomial (mcx_vpoly.hip)
                    const DevVPoly vp = ldk_struct(&a.vpoly[e.pad - 1]);
                    double xv[PPL];
                    bool in = true;
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        xv[q] = a.paths[((int64_t)vp.t_idx * D + vp.col) * ld + i[q]];
                        in = in && xv[q] >= vp.lo && xv[q] <= vp.hi;
                    }
                    if (__all(in)) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) common[q] = dev_vpoly(vp, a.vcoef, xv[q]);
                        j_begin = e.term_end;
                    }
                }
                for (int j = j_begin; j < e.term_end; ++j) {
                    const DevTerm tm = ldk_struct(&a.terms[j]);
                    double av[PPL];
                    dev_atoms<PPL>(tm.atom, a.paths, D, ld, i, av, etab, ec);
                    if (!(FEAT & K2F_DEN) || tm.den < 0) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) common[q] = binary ? fma(tm.w, av[q], common[q]) : common[q] + tm.w * av[q];
                    } else {
                        double dn[PPL];
                        dev_atoms<PPL>(ldk_struct(&a.atoms[tm.den]), a.paths, D, ld, i, dn, etab, ec);
#pragma unroll
                        for (int q = 0; q < PPL; ++q) own[q] += tm.w * av[q] / dn[q];
                    }
                    if (basket) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) glog[q] = fma(tm.w, mcx_log(av[q] + 1e-10), glog[q]);
                    }
                }
                double v[PPL];
                if (e.kind == MCX_EV_CASHFLOW) {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) v[q] = common[q] * num[q] + own[q];
                } else if (binary) {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const double dot = fmin(fmax((common[q] - e.strike + e.aux[2]) / (2.0 * e.aux[2]), 0.0), 1.0);
                        v[q] = e.aux[1] * (e.sign > 0.0 ? dot : 1.0 - dot) * num[q];
                    }
                } else {
                    double imm[PPL];
#pragma unroll
                    for (int q = 0; q < PPL; ++q) imm[q] = fmax(e.sign * (common[q] - e.strike), 0.0);
                    if (basket) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            const double geo = fmax(e.sign * (mcx_exp(glog[q]) - e.strike), 0.0);
                            v[q] = (e.aux[0] == 1.0 ? geo : imm[q] - geo + e.aux[1]) * num[q];
                        }
                    } else if (!(FEAT & K2F_EXERCISE) || e.kind == MCX_EV_OPTION) {
#pragma unroll
                        for (int q = 0; q < PPL; ++q) v[q] = imm[q] * num[q];
                    } else {                                   // MCX_EV_EXERCISE (bermudan_option.py:93-131, flexicall.py:118-133)
                        double x[PPL];
                        if (e.coeff_off >= 0) dev_atoms<PPL>(e.x, a.paths, D, ld, i, x, etab, ec);
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            double cont = 0.0, cont_ex = 0.0;
                            if (e.coeff_off >= 0) {
                                cont = dev_poly(a.coeffs + e.coeff_off + s[q] * K, K, x[q]);
                                if (e.aux[0] == 1.0 && s[q] > 0) cont_ex = dev_poly(a.coeffs + e.coeff_off + (s[q] - 1) * K, K, x[q]);
                            }
                            const bool ex = dev_exercise_decision((imm[q] + cont_ex > cont) && (s[q] > 0), s[q], a.ex_mode,
                                                                  a.ex_mode ? a.ex_bits + (int64_t)e_i * a.ex_ld + i[q] : nullptr, 0);
                            if (ex) s[q] -= 1;
                            v[q] = ex ? imm[q] * num[q] : 0.0;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < PPL; ++q) acc[q] += v[q];
            } else {
                double v[PPL];
#pragma unroll
                for (int q = 0; q < PPL; ++q) v[q] = 0.0;
                if (e.kind == MCX_EV_EXPO_POLY) {
                    double x[PPL];
                    dev_atoms<PPL>(e.x, a.paths, D, ld, i, x, etab, ec);
#pragma unroll
                    for (int q = 0; q < PPL; ++q) v[q] = 0.0;
                    if (e.coeff_off >= 0) {
                        if (!(FEAT & K2F_EXERCISE) || pr.n_states == 1) {                // stateless product: one coefficient row for every lane -> scalar loads
                            const double* __restrict__ c = a.coeffs + e.coeff_off + pr.init_state * K;
                            double xp[PPL];
#pragma unroll
                            for (int q = 0; q < PPL; ++q) xp[q] = 1.0;
                            for (int k = 0; k < K; ++k) {
                                const double ck = ldk(c + k);
#pragma unroll
                                for (int q = 0; q < PPL; ++q) { v[q] = fma(ck, xp[q], v[q]); xp[q] *= x[q]; }
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < PPL; ++q) v[q] = dev_poly(a.coeffs + e.coeff_off + s[q] * K, K, x[q]);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < PPL; ++q) v[q] *= num[q];
                } else if ((FEAT & K2F_BS_EXPO) && e.aux[2] > 0.0) {                  // MCX_EV_EXPO_BS (european_option.py:88-145)
                    double spot[PPL];
                    dev_atoms<PPL>(e.x, a.paths, D, ld, i, spot, etab, ec);
                    const double sig = e.aux[0], rate = e.aux[1], tau = e.aux[2], Kx = e.strike;
                    const double sq = sqrt(tau), df = exp(-rate * tau);
#pragma unroll
                    for (int q = 0; q < PPL; ++q) {
                        const double d1 = (log(spot[q] / Kx) + (rate + 0.5 * sig * sig) * tau) / (sig * sq);
                        const double d2 = d1 - sig * sq;
                        const double price = e.sign > 0.0 ? spot[q] * dev_norm_cdf(d1) - Kx * df * dev_norm_cdf(d2)
                                                          : Kx * df * dev_norm_cdf(-d2) - spot[q] * dev_norm_cdf(-d1);
                        v[q] = price * num[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < PPL; ++q) {
                    if (!live[q]) continue;
                    double* dst = a.expo + ((int64_t)pr.netting_set * a.n_expo_rows + e.row) * a.ld_out + i[q];
                    if (e.flags & 1) *dst += v[q];
                    else *dst = v[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) acc_ns[q] += acc[q];
    }
    if (cur_ns >= 0 && a.want_cfs) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) if (live[q]) a.cfs[(int64_t)cur_ns * a.ld_out + i[q]] = acc_ns[q];
    }
}

// out[j] = sum over chunks of part[c][j] in chunk order (deterministic)
__global__ __launch_bounds__(MCX_BLOCK) void k2_sum_chunks(const double* __restrict__ part, int n_chunks, int64_t count, double* __restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (j >= count) return;
    double s = 0.0;
    for (int c = 0; c < n_chunks; ++c) s += part[(int64_t)c * count + j];
    out[j] = s;
}

__global__ __launch_bounds__(MCX_BLOCK) void k2_resolve(const DevAtom* __restrict__ atoms, const int32_t* __restrict__ ids, int n_ids,
                                                        const double* __restrict__ paths, int64_t D, int64_t n, int64_t ld,
                                                        double* __restrict__ out, int64_t ld_out)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int q = blockIdx.y;
    out[(int64_t)q * ld_out + i] = dev_atom(ldk_struct(&atoms[ldk(ids + q)]), paths, D, ld, i);
}

}  // namespace

extern "C" int mcx_eval_book(mcx_handle* h, const mcx_book* b, const double* d_paths, int64_t n_paths, int64_t ld,
                             double* d_cfs, double* d_expo, int64_t ld_out, void* stream)
{
    if (!h || !b || !d_paths) return -1;
    if (n_paths <= 0) return 0;
    if (ld < n_paths || ld_out < n_paths) MCX_FAIL(h, -2, "mcx_eval_book: leading dimension < n_paths");
    if (b->want_cfs && !d_cfs) MCX_FAIL(h, -2, "mcx_eval_book: d_cfs is NULL but cashflows are required");
    if (b->want_expo && !d_expo) MCX_FAIL(h, -2, "mcx_eval_book: d_expo is NULL but exposures are required");
    hipStream_t s = (hipStream_t)stream;
    // netting sets without a Monte-Carlo product keep zero cashflows; exposure rows nobody writes are zeroed
    if (b->want_cfs) MCX_HIP(h, hipMemsetAsync(d_cfs, 0, sizeof(double) * (size_t)b->n_netting_sets * ld_out, s));
    if (b->want_expo && b->expo_needs_memset)
        MCX_HIP(h, hipMemsetAsync(d_expo, 0, sizeof(double) * (size_t)b->n_netting_sets * b->n_expo_rows * ld_out, s));
    K2Args a;
    a.terms = b->d_terms; a.events = b->d_events; a.products = b->d_products; a.atoms = b->d_atoms; a.coeffs = b->d_coeffs;
    a.paths = d_paths; a.cfs = d_cfs; a.expo = d_expo; a.n = n_paths; a.ld = ld; a.ld_out = ld_out;
    a.n_products = b->n_products; a.n_basis = b->n_basis; a.n_state = b->n_state; a.n_expo_rows = b->n_expo_rows;
    a.want_cfs = b->want_cfs; a.want_expo = b->want_expo;
    a.chunk_products = 0; a.n_netting_sets = b->n_netting_sets; a.bridge = b->d_bridge;
    a.ex_mode = b->ex_mode; a.ex_bits = b->d_ex_bits; a.ex_ld = b->ex_ld;
    a.vpoly = b->d_vpoly; a.vcoef = b->d_vcoef;
    if (a.ex_mode && a.ex_ld < n_paths) MCX_FAIL(h, -2, "mcx_eval_book: exercise replay buffer narrower than the path count");
    const int grid = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    // Few paths x many products (the reference's 5,000-product books run on ~1,000 paths): the path grid alone leaves the
    // chip empty (4 workgroups) while every lane walks ~10^6 events.  Split the PRODUCT list over blockIdx.y instead; each
    // chunk accumulates into its own image, a second kernel adds the images in chunk order.
    const size_t img = (size_t)b->n_netting_sets * ((b->want_expo ? b->n_expo_rows : 0) + (b->want_cfs ? 1 : 0)) * (size_t)ld_out * sizeof(double);
    int n_chunks = 1;
    if (grid < 64 && b->n_products >= 64 && ld_out == n_paths && img > 0) {
        n_chunks = std::min(std::min(1024 / grid, b->n_products / 8), (int)std::max<size_t>(1, ((size_t)512 << 20) / img));
        if (n_chunks < 2) n_chunks = 1;
    }
    if (n_chunks == 1) {
        const bool barrier = b->has_barrier;
#ifndef MCX_K2_PPL
#define MCX_K2_PPL 2
#endif
        constexpr int PPL = MCX_K2_PPL;
        if (!barrier && grid >= 8 * h->n_cu) {          // enough paths to fill the chip at PPL paths per lane
            const int gv = (int)((n_paths + (int64_t)MCX_BLOCK * PPL - 1) / ((int64_t)MCX_BLOCK * PPL));
            const int feat = (b->has_exotic ? K2F_EXOTIC : 0) | (b->has_exercise ? K2F_EXERCISE : 0) | (b->has_bs_expo ? K2F_BS_EXPO : 0) |
                             (b->has_den ? K2F_DEN : 0);          // (found once, at mcx_book_create)
#ifndef MCX_K2_PPL_LIGHT
#define MCX_K2_PPL_LIGHT 4
#endif
            // cashflow / plain-option / polynomial-exposure books are light on registers: more paths per lane, so that every
            // wave of a million-path run is resident at once and the event chain is walked once per 256 paths
            constexpr int PL = MCX_K2_PPL_LIGHT;
            const bool wide = grid >= 4 * PL * h->n_cu;
            const int gl = (int)((n_paths + (int64_t)MCX_BLOCK * PL - 1) / ((int64_t)MCX_BLOCK * PL));
            if (feat == 0 && wide) hipLaunchKernelGGL((k2_eval_book_v<PL, 0>), dim3(gl), dim3(MCX_BLOCK), 0, s, a);
            else if (feat == K2F_DEN && wide) hipLaunchKernelGGL((k2_eval_book_v<PL, K2F_DEN>), dim3(gl), dim3(MCX_BLOCK), 0, s, a);
            else if (feat == 0) hipLaunchKernelGGL((k2_eval_book_v<PPL, 0>), dim3(gv), dim3(MCX_BLOCK), 0, s, a);
            else if (feat == K2F_DEN) hipLaunchKernelGGL((k2_eval_book_v<PPL, K2F_DEN>), dim3(gv), dim3(MCX_BLOCK), 0, s, a);
            else if (!(feat & (K2F_EXOTIC | K2F_BS_EXPO))) hipLaunchKernelGGL((k2_eval_book_v<PPL, K2F_DEN | K2F_EXERCISE>), dim3(gv), dim3(MCX_BLOCK), 0, s, a);
            else if (!(feat & K2F_BS_EXPO)) hipLaunchKernelGGL((k2_eval_book_v<PPL, K2F_DEN | K2F_EXERCISE | K2F_EXOTIC>), dim3(gv), dim3(MCX_BLOCK), 0, s, a);
            else hipLaunchKernelGGL((k2_eval_book_v<PPL, K2F_ALL>), dim3(gv), dim3(MCX_BLOCK), 0, s, a);
            MCX_HIP(h, hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL(k2_eval_book, dim3(grid), dim3(MCX_BLOCK), 0, s, a);
        MCX_HIP(h, hipGetLastError());
        return 0;
    }
    a.chunk_products = (b->n_products + n_chunks - 1) / n_chunks;
    n_chunks = (b->n_products + a.chunk_products - 1) / a.chunk_products;
    const int64_t n_cfs = b->want_cfs ? (int64_t)b->n_netting_sets * ld_out : 0;
    const int64_t n_expo = b->want_expo ? (int64_t)b->n_netting_sets * b->n_expo_rows * ld_out : 0;
    double* d_part = (double*)mcx_scratch(h, 2, sizeof(double) * (size_t)n_chunks * (size_t)(n_cfs + n_expo));
    if (!d_part) return -100;
    MCX_HIP(h, hipMemsetAsync(d_part, 0, sizeof(double) * (size_t)n_chunks * (size_t)(n_cfs + n_expo), s));
    a.cfs = n_cfs ? d_part : nullptr;
    a.expo = n_expo ? d_part + (size_t)n_chunks * n_cfs : nullptr;
    hipLaunchKernelGGL(k2_eval_book, dim3(grid, n_chunks), dim3(MCX_BLOCK), 0, s, a);
    if (n_cfs) hipLaunchKernelGGL(k2_sum_chunks, dim3((unsigned)((n_cfs + MCX_BLOCK - 1) / MCX_BLOCK)), dim3(MCX_BLOCK), 0, s, a.cfs, n_chunks, n_cfs, d_cfs);
    if (n_expo) hipLaunchKernelGGL(k2_sum_chunks, dim3((unsigned)((n_expo + MCX_BLOCK - 1) / MCX_BLOCK)), dim3(MCX_BLOCK), 0, s, a.expo, n_chunks, n_expo, d_expo);
    MCX_HIP(h, hipGetLastError());
    return 0;          // stream-ordered (the chunk images live in a scratch buffer of the handle)
}

extern "C" int mcx_resolve_atoms(mcx_handle* h, const mcx_book* b, const int32_t* h_atom_ids, int32_t n_ids, const double* d_paths,
                                 int64_t n_paths, int64_t ld, double* d_out, int64_t ld_out, void* stream)
{
    if (!h || !b || !h_atom_ids || !d_paths || !d_out) return -1;
    if (n_ids <= 0 || n_paths <= 0) return 0;
    if (n_ids > 65535) MCX_FAIL(h, -2, "mcx_resolve_atoms: too many atoms in one call");
    for (int q = 0; q < n_ids; ++q)
        if (h_atom_ids[q] < 0 || h_atom_ids[q] >= b->n_atoms) MCX_FAIL(h, -2, "mcx_resolve_atoms: atom id out of range");
    hipStream_t s = (hipStream_t)stream;
    const int32_t* d_ids = (const int32_t*)mcx_stage_small(h, h_atom_ids, sizeof(int32_t) * (size_t)n_ids, s);
    if (!d_ids) return -100;
    const int gx = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    hipLaunchKernelGGL(k2_resolve, dim3(gx, n_ids), dim3(MCX_BLOCK), 0, s, b->d_atoms, d_ids, n_ids, d_paths, (int64_t)b->n_state,
                       n_paths, ld, d_out, ld_out);
    MCX_HIP(h, hipGetLastError());
    return 0;          // stream-ordered
}
