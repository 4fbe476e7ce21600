// mcx_vpoly.hip — value polynomials: the cash value of an event that sums many atoms of ONE state variable, collapsed on the host
// into a verified polynomial of that variable.
//
// Reference dataflow replaced: a Bermudan swaption's exercise value is the underlying swap priced from ~35-64 zero-bond requests
// per exercise date (products/bermudan_option.py:40-43, 93-131; products/bond.py:42-68, 115-163), each one exponential of the short
// rate; the reference evaluates them as separate tensors, the kernels of round 2 as a per-path loop of ~15 VALU instructions per
// term (~500 per path and date: the whole cost of config 5's main kernel and of its LSM roll).  All of them read the SAME state
// variable x of the SAME date, so the sum f(x) = sum_j w_j (a_j + d_j x + b_j exp(c0_j + c1_j x)) is one smooth function on the
// range the paths visit: ~15-20 fused multiply-adds.
//
// Contract (checked HERE, on the host, before a kernel may use a polynomial):
//   |p(x) - f(x)| <= rel_tol * sum_j |w_j| (|a_j| + |d_j x| + |b_j| exp(c0_j + c1_j x))   for every x of a dense grid on [lo, hi]
// with f evaluated in long double and p evaluated in double with exactly the fused multiply-add chain the kernels run
// (t = fma(x, 1/half, -mid/half); Horner from the highest power).  The bound is relative to the sum of the absolute terms — the
// scale that also bounds the rounding error of the exact term loop (n eps of it).  An event whose fit misses the bound keeps the
// term loop; so do lanes whose x lies outside [lo, hi] (the kernels test the range per wave).
#include <cmath>
#include <thread>
#include <vector>

#include "mcx_internal.h"

namespace {

// value and scale (sum of the absolute terms) of the event at x: double exp per term (<= 1 ulp), long double accumulation — the
// reference value is good to ~1.2e-16 of the scale, two orders below the tolerances in use (expl per term made the fit of a
// 120-date Bermudan cost a second of host time)
inline void vp_eval(const mcx_value_term* tm, int n, double x, long double& val, long double& scale)
{
    long double s = 0.0L, sc = 0.0L;
    for (int j = 0; j < n; ++j) {
        const double lin = fma(tm[j].d, x, tm[j].a);
        const double ex = tm[j].b != 0.0 ? tm[j].b * exp(fma(tm[j].c1, x, tm[j].c0)) : 0.0;
        s += (long double)tm[j].w * ((long double)lin + (long double)ex);
        sc += (long double)fabs(tm[j].w) * ((long double)(fabs(tm[j].a) + fabs(tm[j].d * x)) + (long double)fabs(ex));
    }
    val = s; scale = sc;
}

// p(t) with the kernels' arithmetic: coefficients ascending in c[0..deg], evaluated from the highest power by fma
inline double vp_horner(const double* c, int deg, double t)
{
    double p = 0.0;
    for (int k = deg; k >= 0; --k) p = fma(p, t, c[k]);
    return p;
}

}  // namespace

extern "C" int mcx_value_poly_fit(const mcx_value_term* terms, int32_t n_terms, double lo, double hi, double rel_tol, int32_t max_degree,
                                  double* coef, int32_t* degree, double* mid_out, double* half_out, double* max_rel_err)
{
    if (!terms || n_terms < 1 || !coef || !degree || !(hi > lo) || !(rel_tol > 0.0) || max_degree < 1 || max_degree > MCX_VPOLY_MAX_DEGREE)
        return -1;
    if (!std::isfinite(lo) || !std::isfinite(hi)) return -1;
    *degree = -1;
    const long double PI = 3.14159265358979323846264338327950288L;
    const double mid = 0.5 * (lo + hi), half = 0.5 * (hi - lo);
    const double ih = 1.0 / half, ms = -mid * ih;
    constexpr int M = 64;                                  // Chebyshev nodes of the first kind (>= 2 x the highest degree kept)
    static const struct CosTab {                           // cos(pi n (k + 1/2) / M)
        long double c[M / 2][M];
        CosTab() { const long double pi = 3.14159265358979323846264338327950288L;
                   for (int n = 0; n < M / 2; ++n) for (int k = 0; k < M; ++k) c[n][k] = cosl(pi * n * (k + 0.5L) / M); }
    } ct;
    long double fk[M], scale_min = INFINITY;
    for (int k = 0; k < M; ++k) {
        const double x = (double)((long double)mid + (long double)half * ct.c[1][k]);
        long double sc;
        vp_eval(terms, n_terms, x, fk[k], sc);
        if (!std::isfinite((double)fk[k]) || !std::isfinite((double)sc)) return 0;
        if (sc < scale_min) scale_min = sc;
    }
    if (!(scale_min > 0.0L)) {                             // identically zero value: the zero polynomial is exact
        coef[0] = 0.0; *degree = 0;
        if (mid_out) *mid_out = mid;
        if (half_out) *half_out = half;
        if (max_rel_err) *max_rel_err = 0.0;
        return 0;
    }
    // (the nodes were rounded to double: the interpolation conditions hold within 1 ulp of the Chebyshev nodes — a perturbation of
    // eps |f'| half, far below the bound; the verification below measures the polynomial that is actually used)
    const int NC = max_degree + 1 < M / 2 ? max_degree + 1 : M / 2;
    long double cheb[M / 2];
    for (int n = 0; n < NC; ++n) {
        long double s = 0.0L;
        for (int k = 0; k < M; ++k) s += fk[k] * ct.c[n][k];
        cheb[n] = s * (n == 0 ? 1.0L : 2.0L) / M;
    }
    // first guess of the degree: the Chebyshev tail below a tenth of the budget
    int deg = NC - 1;
    {
        long double tail = 0.0L;
        for (; deg > 1; --deg) {
            tail += fabsl(cheb[deg]);
            if (tail > 0.1L * (long double)rel_tol * scale_min) break;
        }
    }
    for (; deg <= max_degree && deg < NC; ++deg) {
        // Chebyshev -> monomial in t (long double), T_0 = 1, T_1 = t, T_{n+1} = 2 t T_n - T_{n-1}
        long double mono[MCX_VPOLY_MAX_DEGREE + 1] = {0}, Tp[MCX_VPOLY_MAX_DEGREE + 1] = {0}, Tc[MCX_VPOLY_MAX_DEGREE + 1] = {0}, Tn[MCX_VPOLY_MAX_DEGREE + 1];
        Tp[0] = 1.0L;
        mono[0] += cheb[0];
        if (deg >= 1) { Tc[1] = 1.0L; mono[1] += cheb[1]; }
        for (int n = 2; n <= deg; ++n) {
            for (int k = 0; k <= deg; ++k) Tn[k] = (k > 0 ? 2.0L * Tc[k - 1] : 0.0L) - Tp[k];
            for (int k = 0; k <= deg; ++k) { mono[k] += cheb[n] * Tn[k]; Tp[k] = Tc[k]; Tc[k] = Tn[k]; }
        }
        double c[MCX_VPOLY_MAX_DEGREE + 1];
        for (int k = 0; k <= deg; ++k) c[k] = (double)mono[k];
        // verification on a dense grid: uniform points and the Chebyshev extremes (where an interpolant's error peaks), endpoints included
        const int G = 4 * (deg + 2);
        long double worst = 0.0L;
        for (int g = 0; g <= 2 * G; ++g) {
            long double xl;
            if (g <= G) xl = (long double)lo + ((long double)hi - (long double)lo) * g / G;
            else xl = (long double)mid + (long double)half * cosl(PI * (g - G) / G);
            double x = (double)xl;
            if (x < lo) x = lo;
            if (x > hi) x = hi;
            const double t = fma(x, ih, ms);
            long double fx, sc;
            vp_eval(terms, n_terms, x, fx, sc);
            const long double rel = fabsl((long double)vp_horner(c, deg, t) - fx) / sc;
            if (rel > worst) worst = rel;
        }
        if (worst <= (long double)rel_tol) {
            for (int k = 0; k <= deg; ++k) coef[k] = c[k];
            *degree = deg;
            if (mid_out) *mid_out = mid;
            if (half_out) *half_out = half;
            if (max_rel_err) *max_rel_err = (double)worst;
            return 0;
        }
    }
    return 0;                                               // *degree = -1: no polynomial within the bound
}

// drop every polynomial of the book (host and device state)
static void vpoly_clear(mcx_book* b)
{
    hipFree(b->d_vpoly); hipFree(b->d_vcoef);
    b->d_vpoly = nullptr; b->d_vcoef = nullptr;
    b->h_vpoly.clear(); b->h_vcoef.clear();
    b->h_event_vpoly.assign((size_t)b->n_events, -1);
    for (auto& e : b->h_events) e.pad = 0;
}

extern "C" int mcx_book_collapse_values(mcx_handle* h, mcx_book* b, const double* h_lo, const double* h_hi, int32_t n_dates, int32_t n_state,
                                        double pad, double rel_tol, int32_t min_terms, int32_t* n_collapsed, void* stream)
{
    if (!h || !b) return -1;
    if (n_collapsed) *n_collapsed = 0;
    hipStream_t s = (hipStream_t)stream;
    const bool off = !h_lo || !h_hi || n_dates <= 0;
    struct Cand { int q, t_idx, col; double lo, hi; std::vector<mcx_value_term> vt; int32_t deg; double mid, half; double coef[MCX_VPOLY_MAX_DEGREE + 1]; };
    std::vector<Cand> cand;
    if (!off) {
        if (n_state != b->n_state) MCX_FAIL(h, -2, "mcx_book_collapse_values: n_state %d != the book's %d", n_state, b->n_state);
        if (!(pad > -0.5) || !(rel_tol > 0.0) || min_terms < 2) MCX_FAIL(h, -2, "mcx_book_collapse_values: bad pad / tolerance / min_terms");
        for (int q = 0; q < b->n_events; ++q) {
            const DevEvent& e = b->h_events[q];
            const int nt = e.term_end - e.term_begin;
            if (nt < min_terms) continue;
            // plain sums only: cashflows, exercise values and plain option payoffs (baskets need the per-term values, binaries and
            // barriers their own state)
            if (!(e.kind == MCX_EV_CASHFLOW || e.kind == MCX_EV_EXERCISE || (e.kind == MCX_EV_OPTION && e.aux[0] == 0.0))) continue;
            Cand c;
            c.q = q; c.t_idx = -1; c.col = -1; c.deg = -1;
            bool ok = true;
            for (int j = e.term_begin; j < e.term_end && ok; ++j) {
                const DevTerm& tm = b->h_terms[j];
                if (tm.den >= 0) { ok = false; break; }
                mcx_value_term v = {tm.w, tm.atom.a, tm.atom.d, tm.atom.b, tm.atom.c0, tm.atom.c1};
                if (tm.atom.col < 0) {                       // state-independent atom: a constant
                    v.a = tm.atom.a + (tm.atom.b != 0.0 ? tm.atom.b * exp(tm.atom.c0) : 0.0); v.d = 0.0; v.b = 0.0;
                } else if (c.col < 0) { c.t_idx = tm.atom.t_idx; c.col = tm.atom.col; }
                else if (tm.atom.t_idx != c.t_idx || tm.atom.col != c.col) ok = false;
                c.vt.push_back(v);
            }
            if (!ok || c.col < 0 || c.t_idx >= n_dates) continue;
            const double lo0 = h_lo[(size_t)c.t_idx * n_state + c.col], hi0 = h_hi[(size_t)c.t_idx * n_state + c.col];
            if (!(hi0 > lo0) || !std::isfinite(lo0) || !std::isfinite(hi0)) continue;      // a date on which every path shares x
            c.lo = lo0 - pad * (hi0 - lo0); c.hi = hi0 + pad * (hi0 - lo0);
            cand.push_back(std::move(c));
        }
    }
    // the same book over the same ranges (a controller re-run on the same seeds): the verified polynomials stand
    {
        std::vector<double> key;
        key.push_back(rel_tol);
        for (const Cand& c : cand) { key.push_back((double)c.q); key.push_back(c.lo); key.push_back(c.hi); }
        if (key == b->vpoly_key) {
            if (n_collapsed) *n_collapsed = (int32_t)b->h_vpoly.size();
            return 0;
        }
        b->vpoly_key = key;
    }
    // one fit per candidate, spread over the host cores (a 120-date Bermudan: ~0.15 ms per date single-threaded)
    {
        unsigned nthr = std::thread::hardware_concurrency();
        if (nthr > 16) nthr = 16;
        if (nthr < 1) nthr = 1;
        if (nthr > cand.size()) nthr = (unsigned)cand.size();
        auto work = [&](unsigned id) {
            for (size_t k = id; k < cand.size(); k += nthr) {
                Cand& c = cand[k];
                int32_t deg = -1;
                double err = 0.0;
                if (mcx_value_poly_fit(c.vt.data(), (int32_t)c.vt.size(), c.lo, c.hi, rel_tol, MCX_VPOLY_MAX_DEGREE, c.coef, &deg, &c.mid, &c.half, &err) != 0) deg = -1;
                // a polynomial must also be cheaper than the loop it replaces (~15 VALU per exponential term)
                if (deg + 1 > 8 * (int)c.vt.size()) deg = -1;
                c.deg = deg;
            }
        };
        if (nthr <= 1) { if (!cand.empty()) work(0); }
        else {
            std::vector<std::thread> pool;
            for (unsigned id = 0; id < nthr; ++id) pool.emplace_back(work, id);
            for (auto& t : pool) t.join();
        }
    }
    MCX_HIP(h, hipStreamSynchronize(s));                    // kernels of earlier calls may still read the tables replaced below
    const bool had = !b->h_vpoly.empty();
    vpoly_clear(b);
    for (const Cand& c : cand) {
        if (c.deg < 0) continue;
        DevVPoly vp;
        vp.lo = c.lo; vp.hi = c.hi; vp.ih = 1.0 / c.half; vp.ms = -c.mid * vp.ih;
        vp.n_blk = (c.deg + 1 + MCX_VPOLY_BLK - 1) / MCX_VPOLY_BLK;
        vp.coef_off = (int32_t)b->h_vcoef.size(); vp.t_idx = c.t_idx; vp.col = c.col;
        // highest power first, zero-padded in FRONT to whole blocks (Horner starts from p = 0)
        const int len = vp.n_blk * MCX_VPOLY_BLK;
        for (int k = len - 1; k >= 0; --k) b->h_vcoef.push_back(k <= c.deg ? c.coef[k] : 0.0);
        b->h_event_vpoly[c.q] = (int32_t)b->h_vpoly.size();
        b->h_vpoly.push_back(vp);
        b->h_events[c.q].pad = (int32_t)b->h_vpoly.size();  // index + 1
    }
    if (n_collapsed) *n_collapsed = (int32_t)b->h_vpoly.size();
    if (b->h_vpoly.empty() && !had) return 0;
    if (!b->h_vpoly.empty()) {
        // (one spare block at the end: the kernels prefetch the next block while they consume the current one)
        for (int k = 0; k < MCX_VPOLY_BLK; ++k) b->h_vcoef.push_back(0.0);
        hipError_t e = hipMalloc((void**)&b->d_vpoly, sizeof(DevVPoly) * b->h_vpoly.size());
        if (e == hipSuccess) e = hipMalloc((void**)&b->d_vcoef, sizeof(double) * b->h_vcoef.size());
        if (e == hipSuccess) e = hipMemcpy(b->d_vpoly, b->h_vpoly.data(), sizeof(DevVPoly) * b->h_vpoly.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(b->d_vcoef, b->h_vcoef.data(), sizeof(double) * b->h_vcoef.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            vpoly_clear(b);
            b->vpoly_key.clear();
            hipMemcpy(b->d_events, b->h_events.data(), sizeof(DevEvent) * b->h_events.size(), hipMemcpyHostToDevice);
            MCX_FAIL(h, -100 - (int)e, "mcx_book_collapse_values: %s", hipGetErrorString(e));
        }
    }
    if (b->n_events > 0) MCX_HIP(h, hipMemcpy(b->d_events, b->h_events.data(), sizeof(DevEvent) * b->h_events.size(), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int mcx_book_value_poly_info(const mcx_book* b, int32_t event, int32_t* degree_blocks, double* lo, double* hi)
{
    if (!b || event < 0 || event >= b->n_events) return -1;
    const int v = b->h_event_vpoly.empty() ? -1 : b->h_event_vpoly[event];
    if (v < 0) return 0;
    if (degree_blocks) *degree_blocks = b->h_vpoly[v].n_blk;
    if (lo) *lo = b->h_vpoly[v].lo;
    if (hi) *hi = b->h_vpoly[v].hi;
    return 1;
}
