/*
 * mcx.h — C ABI of the MI355X-native Monte-Carlo path & exposure engine (libmcx_hip.so).
 *
 * The reference (konstantineder/montecarlo-risk-engine) is pure Python on torch-CPU and has NO FFI; its seams are
 * Python call signatures.  This header is the boundary a maintainer would bind (ctypes) to replace, for ONE hot path:
 *
 *   reference seam (file:line)                                   entry point here
 *   ------------------------------------------------------------ -------------------------------------------
 *   MonteCarloEngine.generate_paths        engine/engine.py:27-123        mcx_generate_paths          (K1)
 *   Model.generate_correlated_randn        models/model.py:38-48          (inside K1: Philox4x32-10 + Box-Muller + L·z)
 *   *.simulate_time_step_{analytically,euler,qe}  models/<model>.py           (inside K1)
 *   RequestInterface.resolve_requests      request_interface.py:115-130   mcx_resolve_atoms / fused in K2
 *   SimulationController._evaluate_product controller/controller.py:385-471   mcx_eval_book           (K2)
 *   SimulationController._perform_regression_for_product  controller.py:294-383  mcx_lsm_stats, mcx_lsm_step (K3)
 *   NettingSet.compute_unsecured_exposure_profiles  products/netting_set.py:156-184   (prologue of K4/K5)
 *   Metric._compute_mc_mean_and_error + PV/CE/EPE/ENE/CVA  metrics/<metric>.py   mcx_reduce_vector, mcx_reduce_profiles, mcx_reduce_cva (K4)
 *   PFEMetric.evaluate_numerically (torch.sort)     metrics/pfe_metric.py:49-73       mcx_select_hist  (K5)
 *
 * Conventions
 *   - every d_* pointer is DEVICE memory owned by the caller (torch.empty(device="cuda").data_ptr()); the library never
 *     frees caller memory and keeps no reference to it after the call returns.
 *   - every h_* / descriptor pointer is HOST memory, copied during the call.
 *   - all floating point is IEEE binary64.  Layout of paths: [T][D][ld] with the path index contiguous (ld >= n_paths).
 *   - calls enqueue on the hipStream_t passed as `stream` (NULL = default stream) and return without synchronising,
 *     except where an h_* output is written (then the call synchronises that stream).
 *   - return 0 on success, <0 on error; mcx_last_error() describes the last failure on that handle.
 *   - one handle per GPU; a handle is not thread-safe; distinct handles may be used from distinct threads.
 *
 * The same descriptors are consumed by the CPU oracle (oracle/mcx_oracle.c, prefix orc_) which is TEST infrastructure.
 */
#ifndef MCX_H
#define MCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCX_ABI_VERSION 6   /* 6: mcx_lsm_run_batch; 5: value polynomials (mcx_book_collapse_values), mcx_book_set_exercise_replay n_rows, mcx_sim_create rejects (model, scheme) pairs without a step map; 4: mcx_unsecured_desc.n_rows, mcx_lsm_step_batch w_len (host-side bounds of every row / offset a kernel reads); 3: mcx_rng_draws, mcx_comm_* (RCCL), interpolated collateral; 2: batched LSM, tangent-book kernels, bridge RNG */

#define MCX_MAX_SLOTS   8    /* sub-models in one ModelConfig                                  */
#define MCX_MAX_Z       8    /* total simulation dimension (correlated normals per sub-step)    */
#define MCX_MAX_STATE   16   /* total state dimension D                                         */
#define MCX_SLOT_NPARAM 8
#define MCX_AUX         8    /* host-precomputed per-(sub-step, slot) constants                 */
#define MCX_MAX_BASIS   6    /* regression basis functions (polynomial degree + 1)              */
#define MCX_MAX_STATES  8    /* product exercise states (Bermudan: 2, FlexiCall: rights + 1)    */

/* SimulationScheme values mirror common/enums.py:4-9 */
enum { MCX_SCHEME_EULER = 0, MCX_SCHEME_MILSTEIN = 1, MCX_SCHEME_ANALYTICAL = 2, MCX_SCHEME_QE = 3 };

/* sub-model kinds */
enum {
    MCX_MODEL_BS        = 1, /* p = [spot, sigma, rate]                       state [S]        black_scholes.py:50-85      */
    MCX_MODEL_HESTON    = 2, /* p = [spot, sigma_v, rate, rho, kappa, theta, v0]  state [logS, v]  heston.py:99-253        */
    MCX_MODEL_VASICEK   = 3, /* p = [r0, sigma, theta, a]                     state [r, logB]  vasicek.py:61-112           */
    MCX_MODEL_CIRPP     = 4, /* p = [kappa, theta, sigma, y0]                 state [y, Lambda] cirpp.py:174-198           */
    MCX_MODEL_CIRPP_DET = 5, /* deterministic hazard                          state [lam, Lambda] cirpp.py:155-172         */
    MCX_MODEL_HW        = 6, /* Hull-White 1F: p = [r0, sigma, -, a], theta(t) in aux   hull_white.py:58-88 (spec only)    */
    MCX_MODEL_S2F       = 7  /* Schwartz two-factor: p = [rate, kappa, sigma_s, mu, sigma_l, rho, log F0(t0)]
                                state [logS, x, y] (3 columns), 2 normals              schwartz_two_factor.py:147-196     */
};

/* per-(sub-step, slot) aux[] semantics (filled by the host from the model parameters and dt):
 *   BS        ANALYTICAL: aux0 = rate*dt, aux1 = 0.5*dt*sigma^2          EULER: unused
 *   VASICEK   ANALYTICAL: aux0 = exp(-a*dt)                               EULER: unused
 *   HW        ANALYTICAL: aux0 = exp(-a*dt), aux1 = mean shift alpha(t2)-alpha(t1)*exp(-a dt); EULER: aux0 = theta(t1)
 *   CIRPP     EULER:      aux0 = psi(t1)
 *   CIRPP_DET any:        aux0 = lambda_mkt(t1), aux1 = lambda_mkt(t2)
 *   S2F       ANALYTICAL: aux0 = exp(-kappa dt) (1 when kappa ~ 0), aux1 = log F0(t2)        EULER: aux1 = log F0(t2)
 *   HESTON    QE:         aux0 = E = exp(-kappa dt), aux1..aux5 = K0..K4, aux6 = sigma^2 E (1-E)/kappa, aux7 = theta sigma^2 (1-E)^2/(2 kappa)
 * Under EULER, aux4..aux6 of BS / VASICEK / CIRPP slots are reserved: mcx_sim_create overwrites them in its device copy with
 * step constants derived from the slot parameters and dt (drift and diffusion factors); the caller's array is not modified.
 */

#define MCX_FLAG_SMOOTHING 1  /* Model.perform_smoothing (fuzzy QE branches), model.py:83-90 */

typedef struct {
    int32_t kind;
    int32_t state_off;   /* first state column of this sub-model        */
    int32_t z_off;       /* first correlated normal consumed            */
    int32_t flags;
    double  p[MCX_SLOT_NPARAM];
} mcx_slot;

typedef struct {
    double  dt;          /* sub-step length (engine.py:49-50)                                        */
    double  sqrt_dt;
    double  t1;          /* start time of the sub-step: accumulated t_prev (engine.py:60)            */
    int32_t store_idx;   /* timeline index stored after this sub-step, -1 = none                      */
    int32_t chol_idx;    /* Cholesky factor used (one per distinct dt for ANALYTICAL, model.py:56-64) */
} mcx_step;

typedef struct {
    int32_t scheme;           /* MCX_SCHEME_*                                                         */
    int32_t n_slots;
    int32_t n_state;          /* D                                                                    */
    int32_t n_z;              /* normals per sub-step                                                 */
    int32_t n_uniform;        /* extra uniforms per sub-step (Heston QE: 1, heston.py:192)            */
    int32_t n_steps;          /* total sub-steps S                                                    */
    int32_t n_dates;          /* stored timeline dates T                                              */
    int32_t n_chol;
    int32_t n_initial_store;  /* leading timeline dates stored before any step (t == calibration date)*/
    int32_t flags;
    mcx_slot slots[MCX_MAX_SLOTS];
    const mcx_step* steps;    /* [n_steps]                                                            */
    const double*   chol;     /* [n_chol][n_z][n_z] row-major, lower triangular                       */
    const double*   aux;      /* [n_steps][n_slots][MCX_AUX]                                          */
    const double*   init_state; /* [n_state]                                                          */
} mcx_sim_desc;

/* ---- RNG contract (bit-exact between oracle and HIP) ----------------------------------------------------------------
 * Philox4x32-10 (Salmon et al. 2011, Random123 constants M0=0xD2511F53 M1=0xCD9E8D57 W0=0x9E3779B9 W1=0xBB67AE85).
 *   key     = (seed_lo, seed_hi)
 *   counter = (path_lo, path_hi, sub_step, draw)        path = path_offset + local path index
 *   out (w0,w1,w2,w3):  ua = ((w0 | w1<<32) >> 11 + 0.5) * 2^-53 ,  ub likewise from (w2,w3)   (both in (0,1))
 *   Box-Muller: r = sqrt(-2 ln ua); z[2*draw] = r cos(2 pi ub); z[2*draw+1] = r sin(2 pi ub)
 *   extra uniform j (Heston QE) = ua of draw index ceil(n_z/2)+j
 * Inject mode (parity against recorded reference draws): d_inject_z [n_steps][n_z][ld], d_inject_u [n_steps][ld].
 */

/* ---- book program (K2) ------------------------------------------------------------------------------------------- */

/* atom: a per-path market quantity at one timeline date (the reference's AtomicRequest resolved by a model):
 *   value = a + d*x + b*exp(c0 + c1*x),   x = paths[t_idx][col][n]   (col < 0: x = 0)
 * covers SPOT, NUMERAIRE, DISCOUNT_FACTOR, FORWARD_RATE (ZCB), LIBOR_RATE, SURVIVAL and CONDITIONAL_SURVIVAL for every
 * supported model (vasicek.py:114-156, cirpp.py:246-317, black_scholes.py:87-111, heston.py:255-280). */
typedef struct {
    int32_t t_idx;
    int32_t col;
    double  a, d, b, c0, c1;
} mcx_atom;

typedef struct {
    double  w;
    int32_t atom;
    int32_t den;    /* -1: divided by the event's numeraire; >= 0: this term is divided by atom `den` instead.  Needed for a
                       reference quirk: a swap leg's payment-index labels are looked up in the swap's MERGED timeline
                       (request_interface.py:61-66 with swap.py:86-100), so with tenor_fixed != tenor_float a leg's
                       numeraire / LIBOR are read from the state at another date than the payment date. */
} mcx_term;

enum {
    MCX_EV_CASHFLOW    = 1, /* cfs += sum_j w_j atom_j / (den_j or numeraire)                      bond.py:171-214, swap.py:142-172              */
    MCX_EV_OPTION      = 2, /* cfs += max(sign*(value-strike),0)/numeraire  european_option.py:45-68                      */
    MCX_EV_EXERCISE    = 3, /* Bermudan exercise step                       bermudan_option.py:93-131                     */
    MCX_EV_EXPO_POLY   = 4, /* expo[row] += poly(x; coeffs[state])/numeraire  controller.py:439-447, product.py:157-184   */
    MCX_EV_EXPO_BS     = 5  /* analytic Black-Scholes exposure              european_option.py:123-145                    */
};

typedef struct {
    int32_t kind;
    int32_t t_idx;
    int32_t num_atom;      /* numeraire atom                                                                     */
    int32_t x_atom;        /* explanatory atom (EXERCISE / EXPO_*), -1 = none                                    */
    int32_t term_begin;    /* value = sum_{j in [term_begin, term_end)} terms[j].w * atom(terms[j].atom)          */
    int32_t term_end;
    int32_t coeff_off;     /* offset in coeffs[] of this date's [n_states][n_basis] block; -1 = continuation 0   */
    int32_t expo_row;      /* row of the exposure matrix (EXPO_*)                                                */
    double  strike;
    double  sign;          /* +1 call / -1 put                                                                   */
    double  aux[4];        /* EXPO_BS: sigma, rate, remaining maturity.  EXERCISE: aux[0] = 1: FlexiCall rule, exercise iff
                              immediate + continuation(state-1) > continuation(state) (flexicall.py:118-133).  OPTION: aux[0] = basket aggregation mode
                              (0: value = sum w_j atom_j; 1: geometric exp(sum w_j log(atom_j + 1e-10)); 2: arithmetic payoff
                              - geometric payoff + aux[1], the control variate of basket_option.py:75-82;
                              3: binary payoff aux[1] * ind(sign * (value - strike)) with the fuzzy indicator
                              clamp((x + aux[2]) / (2 aux[2]), 0, 1) of maths.py:3-9, binary_option.py:38-43;
                              4: discretely monitored barrier option (barrier_option.py:60-125): the terms are the
                              monitored spots, x_atom the spot at maturity, aux[1] / aux[2] the barrier levels,
                              aux[3] = type1 + 8 * type2 (1 up-out, 2 down-out, 3 up-in, 4 down-in; type2 0 = none);
                              5: mode 4 with the Brownian-bridge crossing correction (barrier_option.py:126-223):
                              coeffs[coeff_off] = -2 / (sigma^2 maturity / n_obs), coeffs[coeff_off + 1] = draw id;
                              uniforms per mcx_book_set_bridge_rng)                                              */
} mcx_event;

typedef struct {
    int32_t ev_begin, ev_end;   /* main-simulation stream: cashflow + exposure events in evaluation order (controller.py:414-461) */
    int32_t cf_begin, cf_end;   /* product-date events only, product-timeline order (LSM roll, controller.py:333-341)              */
    int32_t netting_set;
    int32_t init_state;         /* product.get_initial_state()                                                                     */
    int32_t n_states;           /* product.get_num_states()                                                                        */
    int32_t flags;
} mcx_product;

typedef struct {
    int32_t n_atoms, n_terms, n_events, n_products;
    int32_t n_netting_sets, n_expo_rows, n_basis, n_coeffs;
    int32_t want_cfs, want_expo;
    int32_t n_state, n_dates;    /* D and T of the paths tensor [T][D][ld] the atoms index into (every t_idx is checked against T) */
    const mcx_atom*    atoms;
    const mcx_term*    terms;
    const mcx_event*   events;
    const mcx_product* products;
    const double*      coeffs;     /* [n_coeffs] initial regression coefficients (usually zeros) */
} mcx_book_desc;

/* ---- metrics (K4/K5) ---------------------------------------------------------------------------------------------- */

/* unsecured exposure of one netting set at metric date m (netting_set.py:156-184):
 *   not collateralised: thr(E[row[m]])            collateralised: E[row[m]] - (delayed[m] >= 0 ? thr(E[delayed[m]]) : 0)
 *   thr(x) = x > h ? x-h : (x < -h ? x+h : 0)  (h == 0: identity)                                netting_set.py:48-72 */
typedef struct {
    int32_t n_dates;            /* metric exposure dates                                     */
    int32_t collateralized;
    double  threshold;
    const int32_t* row;         /* [n_dates] row of the netting set's exposure block          */
    const int32_t* delayed;     /* [n_dates] delayed row or -1 (NULL if not collateralised)   */
    int32_t n_rows;             /* rows of the exposure block handed to the call ([n_rows][ld]): every row / delayed index is
                                   checked against it on the host                              */
    int32_t reserved;
} mcx_unsecured_desc;

/* accumulator record written by every reduction: sums of (x - shift) over the local paths */
typedef struct {
    double n;       /* number of paths reduced      */
    double shift;   /* x of the first local path    */
    double s1;      /* sum (x - shift)              */
    double s2;      /* sum (x - shift)^2            */
} mcx_acc;

/* ---- fused main-simulation pass (K1 + K2 + K4 in one launch) --------------------------------------------------------
 * The reference materialises paths [N,T,D], ~5 resolved request vectors per date and exposures [E,N] between its phases
 * (controller.py:677-694).  For the library's own metrics nothing of that needs to exist in HBM: the fused kernel keeps a
 * path's state in registers, evaluates the book's events of each stored date right after the sub-steps that reach it and
 * reduces PV / EPE / ENE / CVA on the fly.  Paths / exposures / cashflows are still written when the caller passes
 * buffers (PFE select, pluggable metrics, inspection).  Not every book is fusable (collateral look-back, the unequal-tenor
 * swap quirk): mcx_fused_create then returns MCX_E_NOT_FUSABLE and the caller uses K1/K2/K4. */
#define MCX_E_NOT_FUSABLE (-10)
#define MCX_FUSED_MAX_NS 4
#define MCX_FUSED_MAX_STATEFUL 4

typedef struct {
    int32_t netting_set;
    int32_t n_dates;            /* metric exposure dates of this netting set                              */
    int32_t want_profiles;      /* 2*n_dates records {relu(u_m), -relu(-u_m)}  (EPE / ENE / CE)           */
    int32_t want_cva;           /* 1 record: (1-R) sum_m relu(u_m) S(0,t_m) (1 - S(t_m,t_{m+1}))           */
    double  threshold;          /* symmetric threshold (netting_set.py:48-72); collateral is NOT fusable   */
    double  recovery;
    const int32_t* row;         /* [n_dates] exposure row of each metric date                              */
    const int32_t* surv_atoms;  /* [n_dates-1] (want_cva)                                                  */
    const int32_t* cond_atoms;  /* [n_dates-1] (want_cva)                                                  */
} mcx_fused_ns_desc;

typedef struct {
    int32_t n_netting_sets;     /* entries of ns[] (<= MCX_FUSED_MAX_NS)                                   */
    int32_t want_pv;            /* 1 PV record per netting set (pv_metric.py:17-18)                        */
    int32_t n_expo_rows;
    int32_t reserved;
    const int32_t* row_t_idx;   /* [n_expo_rows] timeline index of every exposure row                      */
    const mcx_fused_ns_desc* ns;
} mcx_fused_desc;

typedef struct mcx_handle mcx_handle;
typedef struct mcx_sim    mcx_sim;
typedef struct mcx_book   mcx_book;
typedef struct mcx_fused  mcx_fused;

int         mcx_abi_version(void);
int         mcx_create(mcx_handle** out, int device_id);
void        mcx_destroy(mcx_handle* h);
const char* mcx_last_error(mcx_handle* h);
int         mcx_device_info(mcx_handle* h, int32_t* n_cu, int64_t* hbm_bytes, char* name, int32_t name_len);

/* K1 — replaces MonteCarloEngine(...).generate_paths() (engine/engine.py:10-33). */
int  mcx_sim_create(mcx_handle* h, const mcx_sim_desc* desc, mcx_sim** out);
void mcx_sim_destroy(mcx_sim* sim);
int  mcx_generate_paths(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                        int64_t ld, double* d_paths /* [n_dates][n_state][ld] */,
                        const double* d_inject_z, const double* d_inject_u, void* stream);

/* The same time loop started from a PER-PATH state d_init_state [n_state][ld] instead of the descriptor's initial state — the
 * reference's Model.simulate_time_step_{analytically,euler,qe}(time1, time2, state, corr_randn) (models/vasicek.py:61-112,
 * cirpp.py:174-198, ...) is this call on a one-step descriptor whose Cholesky factor is the identity, with the caller's
 * correlated normals injected; also the restart of a simulation from stored states (nested simulation). */
int  mcx_generate_paths_from_state(mcx_handle* h, const mcx_sim* sim, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                                   int64_t ld, const double* d_init_state, double* d_paths,
                                   const double* d_inject_z, const double* d_inject_u, void* stream);

/* Probe of the RNG contract above (what the path kernels consume): for i < n the draw (path = path0 + i, step, draw) as
 * d_words [4][n] Philox4x32-10 output words (bit-exact against Random123 / the oracle), d_u [2][n] = (ua, ub) and
 * d_z [2][n] = the Box-Muller pair computed exactly as in the kernels (table-driven log / sincos).  Each output is nullable. */
int  mcx_rng_draws(mcx_handle* h, uint64_t seed, uint64_t path0, int64_t n, uint32_t step, uint32_t draw,
                   uint32_t* d_words, double* d_u, double* d_z, void* stream);
/* The map from the four words of a Philox block to the two uniforms and the Box-Muller pair, exactly as the path kernels apply
 * it (d_words [4][n] -> d_u [2][n] (nullable), d_z [2][n]); table_bits = 7: the 128-entry tables of the path kernels, 10: the
 * 1024-entry tables of the one-launch kernel of the two-factor configuration.  For tests: edge words (all ones: u rounds to 1 and
 * the pair is (0, 0)), table-cell boundaries, random words against a libm evaluation. */
int  mcx_box_muller(mcx_handle* h, const uint32_t* d_words, int64_t n, int32_t table_bits, double* d_u, double* d_z, void* stream);

/* K2 — replaces RequestInterface.resolve_requests + SimulationController._evaluate_product summed per netting set
 * (controller/controller.py:385-471, 584-591). d_cfs [n_netting_sets][ld_out], d_expo [n_netting_sets][n_expo_rows][ld_out]. */
int  mcx_book_create(mcx_handle* h, const mcx_book_desc* desc, mcx_book** out);
void mcx_book_destroy(mcx_book* book);
int  mcx_book_set_coeffs(mcx_handle* h, mcx_book* book, int64_t offset, int64_t count, const double* h_coeffs, void* stream);
int  mcx_eval_book(mcx_handle* h, const mcx_book* book, const double* d_paths, int64_t n_paths, int64_t ld,
                   double* d_cfs, double* d_expo, int64_t ld_out, void* stream);
/* Value polynomials.  An event whose cash value sums >= min_terms atoms of ONE state variable x of one date — a Bermudan swaption's
 * exercise value is the underlying swap priced from ~35-64 zero-bond requests per exercise date (products/bermudan_option.py:40-43,
 * 93-131; products/bond.py:42-68, 115-163) — is a smooth function f(x) = sum_j w_j (a_j + d_j x + b_j exp(c0_j + c1_j x)).
 * mcx_book_collapse_values fits one polynomial per such event on [lo - pad w, hi + pad w], w = hi - lo, from the column ranges
 * h_lo / h_hi [n_dates][n_state] (e.g. the extremes of the pre-simulation paths) and VERIFIES on the host
 *     |p(x) - f(x)| <= rel_tol * sum_j |w_j| (|a_j| + |d_j x| + |b_j| exp(c0_j + c1_j x))
 * on a dense grid (f in long double, p with the kernels' FMA chain).  The LSM roll (mcx_lsm_*), the book kernel and the one-launch
 * kernel (a mcx_fused created AFTERWARDS) then evaluate ~20 multiply-adds instead of ~15 instructions per term; events whose fit
 * misses the bound keep the term loop, and so does every wave that holds a path outside the verified range.  h_lo == NULL removes
 * the polynomials; pad > -1/2 (a negative pad verifies a range NARROWER than the data: tests of the fallback).  Synchronises the
 * stream.  mcx_value_poly_fit is the fit alone (host only, no GPU): coef[0..*degree] ascending in
 * t = (x - mid) / half, *degree = -1 when no polynomial of degree <= max_degree (<= MCX_VPOLY_MAX_DEGREE) meets rel_tol. */
#define MCX_VPOLY_MAX_DEGREE 31
typedef struct { double w, a, d, b, c0, c1; } mcx_value_term;
int  mcx_value_poly_fit(const mcx_value_term* terms, int32_t n_terms, double lo, double hi, double rel_tol, int32_t max_degree,
                        double* coef, int32_t* degree, double* mid, double* half, double* max_rel_err);
int  mcx_book_collapse_values(mcx_handle* h, mcx_book* book, const double* h_lo, const double* h_hi, int32_t n_dates, int32_t n_state,
                              double pad, double rel_tol, int32_t min_terms, int32_t* n_collapsed, void* stream);
/* h_out[2 q] = min, h_out[2 q + 1] = max of row q of the [n_rows][ld] tensor d_x over its first n entries (the state columns of the
 * pre-simulation paths [n_dates * n_state][ld]: the ranges mcx_book_collapse_values wants).  Synchronises the stream. */
int  mcx_rows_minmax(mcx_handle* h, const double* d_x, int32_t n_rows, int64_t n, int64_t ld, double* h_out, void* stream);
/* 1 when `event` has a polynomial (its block count of 4 coefficients and verified range are returned), 0 when not, < 0 on error */
int  mcx_book_value_poly_info(const mcx_book* book, int32_t event, int32_t* n_blocks, double* lo, double* hi);
/* materialise resolved requests for pluggable Metric subclasses (request_interface.py:115-130): d_out [n_ids][ld_out] */
int  mcx_resolve_atoms(mcx_handle* h, const mcx_book* book, const int32_t* h_atom_ids, int32_t n_ids,
                       const double* d_paths, int64_t n_paths, int64_t ld, double* d_out, int64_t ld_out, void* stream);

/* fused pass. h_out receives, for every entry of desc.ns in order: [PV record if want_pv][2*n_dates profile records if
 * want_profiles][CVA record if want_cva].  d_paths / d_cfs / d_expo may each be NULL (not materialised). */
int  mcx_fused_create(mcx_handle* h, const mcx_sim* sim, const mcx_book* book, const mcx_fused_desc* desc, mcx_fused** out);
void mcx_fused_destroy(mcx_fused* f);
int  mcx_fused_num_records(const mcx_fused* f);
/* 1 when every timeline date of the program compiled to a straight-line record (the lean kernel: the one-launch plan is then
 * the fastest), 0 when some dates run the generic interpreter (K1 + mcx_fused_eval_paths is usually faster) */
int  mcx_fused_is_straight_line(const mcx_fused* f);
int  mcx_fused_run(mcx_handle* h, const mcx_fused* f, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                   double* d_paths, int64_t ld, double* d_cfs, double* d_expo, int64_t ld_out,
                   const double* d_inject_z, const double* d_inject_u, mcx_acc* h_out, void* stream);

/* the same pass with the records left in DEVICE memory (d_out [mcx_fused_num_records], caller-owned): stream-ordered, the call
 * does not synchronise — for a multi-GPU caller that gathers the records of all ranks on the device (one RCCL all-gather,
 * mcx_allgather_f64 or torch.distributed) before the single device-to-host copy of a pass */
int  mcx_fused_run_device(mcx_handle* h, const mcx_fused* f, uint64_t seed, uint64_t path_offset, int64_t n_paths,
                          double* d_paths, int64_t ld, double* d_cfs, double* d_expo, int64_t ld_out,
                          const double* d_inject_z, const double* d_inject_u, mcx_acc* d_out, void* stream);
/* Measurement: time the MAIN kernel of a fused pass alone (not the few-microsecond record merge that follows it) with event pairs
 * recorded around its launch on the launch stream.  mcx_fused_set_timing(f, k) arms a ring of MCX_FUSED_TIMING_RING pairs and
 * times every k-th launch (k = 1: all; a pair of event markers costs ~8 us of stream time on this stack, so a benchmark samples);
 * k = 0 disarms.  mcx_fused_kernel_times synchronises the recorded pairs, returns their durations in launch order (ms) and
 * re-arms the ring. */
#define MCX_FUSED_TIMING_RING 64
int  mcx_fused_set_timing(mcx_fused* f, int32_t enable);
int  mcx_fused_kernel_times(mcx_fused* f, float* h_ms, int32_t capacity, int32_t* n_out);

/* same event/metric program on a paths tensor produced earlier by mcx_generate_paths: ONE pass over the paths replaces
 * mcx_eval_book + mcx_reduce_* (no exposure matrix is written unless d_expo is given). Record layout as mcx_fused_run. */
int  mcx_fused_eval_paths(mcx_handle* h, const mcx_fused* f, const double* d_paths, int64_t n_paths, int64_t ld,
                          double* d_cfs, double* d_expo, int64_t ld_out, mcx_acc* h_out, void* stream);

int  mcx_fused_eval_paths_device(mcx_handle* h, const mcx_fused* f, const double* d_paths, int64_t n_paths, int64_t ld,
                                 double* d_cfs, double* d_expo, int64_t ld_out, mcx_acc* d_out, void* stream);

/* Tangent (forward-mode) pass — replaces `torch.autograd.grad(value, model.get_model_params())` (controller.py:609-627) for
 * PV metrics of European options under a single Black-Scholes or Heston model (BASELINE configs 2 and 4): the path kernel
 * carries d state / d theta_j for every model parameter in dual numbers (same smoothing / subgradient conventions as
 * torch: clamp passes the gradient on [min, max], torch.maximum splits ties).  Outputs per path:
 *   d_cfs  [n_netting_sets][ld_out]                 normalised cashflows (primal, with smoothing on)
 *   d_dcfs [n_netting_sets][n_params][ld_out]       d cfs / d theta_j     (parameter order = model.get_model_params()) */
typedef struct {
    int32_t t_idx;            /* timeline index of the exercise date                     */
    int32_t netting_set;
    double  strike, sign;     /* +1 call / -1 put                                        */
    double  numeraire;        /* exp(rate * (T - t0))                                    */
    double  dnum_drate;       /* d numeraire / d rate = (T - t0) * numeraire             */
} mcx_tangent_option;
int  mcx_tangent_european(mcx_handle* h, const mcx_sim* sim, const mcx_tangent_option* h_opts, int32_t n_opts,
                          int32_t n_netting_sets, uint64_t seed, uint64_t path_offset, int64_t n_paths, int64_t ld,
                          double* d_cfs, double* d_dcfs, int64_t ld_out, const double* d_inject_z, const double* d_inject_u,
                          void* stream);

/* ---- forward-mode pass through the exposure path (csrc/kt_book.hip) -------------------------------------------------------
 * d metric / d theta for MCX_TANGENT_NP model parameters per pass, replacing torch.autograd.grad through pre-simulation, lstsq
 * and main simulation (controller.py:609-627, :370-383).  The caller supplies the derivative of every host-computed descriptor
 * number next to the primal descriptor:
 *   h_dslot [n_slots][MCX_SLOT_NPARAM][NP], h_dinit [n_state][NP], h_daux [n_steps][n_slots][MCX_AUX][NP]   (simulation)
 *   d_datoms [n_atoms][5][NP] = d(a, d, b, c0, c1)                                                          (book, device)
 * Tangent tensors carry the parameter index outermost: d_dpaths [NP][T][D][ld], d_cfs [1+NP][n_ns][ld] (index 0 = value),
 * d_expo [1+NP][n_ns][n_rows][ld].  Scope: EULER; BS / Vasicek / CIR++ slots; cashflow, plain option, exercise and polynomial-
 * exposure events; otherwise MCX_E_NOT_FUSABLE (the caller falls back to bump-and-revalue). */
#define MCX_TANGENT_NP 4
int  mcx_tangent_paths(mcx_handle* h, const mcx_sim* sim, const double* h_dslot, const double* h_dinit, const double* h_daux,
                       uint64_t seed, uint64_t path_offset, int64_t n_paths, double* d_paths, double* d_dpaths, int64_t ld,
                       const double* d_inject_z, void* stream);
/* dual normal equations of (product, regression date): Y = numeraire * sum of the product's cash events with index >=
 * first_event; h_moments [1+NP][(2K-1)+K]: sums of z^k (k < 2K-1) then of z^k Y (k < K), z = (x - shift) * scale */
int  mcx_tangent_lsm(mcx_handle* h, const mcx_book* book, int32_t product, int32_t first_event, int32_t num_atom, int32_t x_atom,
                     double shift, double scale, const double* d_datoms, const double* d_paths, const double* d_dpaths,
                     int64_t n_paths, int64_t ld, int32_t n_dates, double* h_moments, void* stream);
/* the same for a product with exercise rights (Bermudan / American / FlexiCall, bermudan_option.py:93-131, flexicall.py:118-133):
 * one step of the backward induction in dual numbers.  d_W [S][ld_w] / d_dW [NP][S][ld_w] are the cashflow cache per hypothetical
 * state and its tangents (zero before the first step); the step rolls them over the product's cash events
 * [roll_begin, roll_end) (relative to its first cash event) along the FROZEN exercise policy — decisions from the primal values and
 * the book's current coefficients, tangents through the taken branch only, as on the reference's tape — and returns
 * h_moments [1+NP][(2K-1) + S K]: sums of z^k, then of z^k numeraire W_s per state s. */
int  mcx_tangent_lsm_step(mcx_handle* h, const mcx_book* book, int32_t product, int32_t roll_begin, int32_t roll_end, int32_t num_atom,
                          int32_t x_atom, double shift, double scale, const double* d_datoms, const double* d_paths,
                          const double* d_dpaths, int64_t n_paths, int64_t ld, int32_t n_dates, double* d_W, double* d_dW,
                          int64_t ld_w, double* h_moments, void* stream);
int  mcx_tangent_eval(mcx_handle* h, const mcx_book* book, const double* d_datoms, const double* d_coeffs, const double* d_dcoeffs,
                      const double* d_paths, const double* d_dpaths, int64_t n_paths, int64_t ld, int32_t n_dates,
                      double* d_cfs, double* d_expo,
                      const int32_t* h_ev_param /* nullable [n_events][2]: tangent slot (or -1) of sigma / rate of EXPO_BS events */,
                      void* stream);
/* per-path CVA of one netting set with tangents: d_out [1+NP][ld]; d_expo_ns = the netting set's block of d_expo,
 * expo_tangent_stride = doubles between consecutive tangent images (n_ns * n_rows * ld) */
/* EPE / ENE profile tangents of one netting set: h_out [n_dates_metric][2][NP] = sum over the local paths of 1[u > 0] du and of
 * 1[u < 0] du, u = the thresholded exposure (the caller divides by the global path count) */
int  mcx_tangent_profiles(mcx_handle* h, const int32_t* h_rows, const int32_t* h_delayed /* nullable */, int32_t collateralized,
                          int32_t n_dates_metric, double threshold, const double* d_expo_ns,
                          int64_t expo_tangent_stride, int64_t n_paths, int64_t ld, double* h_out, void* stream);
/* PFE tangent: for every metric date the local path with the smallest index whose unsecured exposure equals h_targets[m] (the
 * exact order statistic from the radix select): h_out [n_dates_metric][1+NP] = (index or -1, its tangent) — the reference
 * differentiates through torch.sort, i.e. through the selected element (pfe_metric.py:61-66) */
int  mcx_tangent_pick(mcx_handle* h, const int32_t* h_rows, const int32_t* h_delayed /* nullable */, int32_t collateralized,
                      int32_t n_dates_metric, double threshold, const double* h_targets, const double* d_expo_ns,
                      int64_t expo_tangent_stride, int64_t n_paths, int64_t ld, double* h_out, void* stream);
int  mcx_tangent_cva(mcx_handle* h, const mcx_book* book, const double* d_datoms, const int32_t* h_rows, const int32_t* h_surv,
                     const int32_t* h_cond, const int32_t* h_delayed /* nullable: rows at t - MPoR */, int32_t collateralized,
                     int32_t n_dates_metric, double threshold, double recovery, const double* d_expo_ns,
                     int64_t expo_tangent_stride, const double* d_paths, const double* d_dpaths, int64_t n_paths, int64_t ld,
                     int32_t n_dates, double* d_out, void* stream);

/* K3 — Longstaff-Schwartz normal equations (controller/controller.py:316-374).
 * mcx_lsm_stats: h_out[2*i+0] = min x_i, h_out[2*i+1] = max x_i over local paths for each explanatory atom (basis centring /
 *                scaling z = (x-shift)*scale, and detection of the exactly rank-1 regression at the calibration date).
 * mcx_lsm_step : for every path and hypothetical start state s0, accumulate the cashflows of product-date events
 *                roll_begin .. roll_end-1 forward along the exercise policy and add the cached tail W[final state]
 *                (d_W [n_states][ld_w] in/out, the reference's cf_cache), then with z = (x - shift)*scale and
 *                Y_s = numeraire * W[s] accumulate   d_moments[k] = sum z^k (k < 2K-1),  d_moments[2K-1 + s*K + k] = sum z^k Y_s.
 *                d_moments is device memory (so it can be all-reduced over RCCL before the host solve).
 *                flags: MCX_LSM_MFMA selects the v_mfma_f64_16x16x4_f64 Gram kernel (default: VALU + wave reduction);
 *                MCX_LSM_F32_CACHE reproduces the reference's float32 cf_cache (torch.zeros without dtype,
 *                controller.py:312-317,330): the running cashflow sum is rounded to float32 after every product date. */
#define MCX_LSM_MFMA      1
#define MCX_LSM_F32_CACHE 2
int  mcx_lsm_stats(mcx_handle* h, const mcx_book* book, const int32_t* h_atom_ids, int32_t n_ids,
                   const double* d_paths, int64_t n_paths, int64_t ld, double* h_out, void* stream);
int  mcx_lsm_step(mcx_handle* h, const mcx_book* book, int32_t product, int32_t roll_begin, int32_t roll_end,
                  int32_t num_atom, int32_t x_atom, double shift, double scale,
                  const double* d_paths, int64_t n_paths, int64_t ld,
                  double* d_W, int64_t ld_w, double* d_moments, int32_t flags, void* stream);

/* The whole backward induction of ONE product on the device: for every date of h_dates, in order, the roll + moments of
 * mcx_lsm_step, the (rank-aware) K x K solve and the scatter of the [n_states][K] coefficient block into the book's coefficient
 * array (where the next date's exercise decisions read it) are enqueued back to back on the stream — no host round trip per date
 * (controller.py:294-383 is a Python loop with one lstsq per date).  With a communicator on the handle (mcx_comm_init) the moments
 * are all-reduced over the ranks between roll and solve, also stream-ordered.  h_coeffs [n_dates][n_states][K] receives the
 * coefficients; h_status[d] != 0 marks a numerically singular system (its coefficients were NOT written: the caller re-solves
 * that product with mcx_lsm_step and its own solver).  degenerate / x0: every path shares x = x0 (minimum-norm rank-1 solution). */
typedef struct {
    int32_t roll_begin, roll_end, num_atom, x_atom;
    int32_t degenerate, reserved;
    int64_t coeff_off[2];   /* offsets of the date's coefficient block in the book's coefficient array, -1 = none        */
    double  shift, scale, x0;
} mcx_lsm_date;
int  mcx_lsm_run(mcx_handle* h, mcx_book* book, int32_t product, const mcx_lsm_date* h_dates, int32_t n_dates,
                 const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w,
                 double* h_coeffs, int32_t* h_status, int32_t flags, void* stream);
/* One date of that chain for callers that own the exchange between roll and solve (several GPUs under torch.distributed: mcx_lsm_step
 * leaves the moments on the device, the caller all-reduces them stream-ordered, mcx_lsm_solve runs the K x K solve and the
 * coefficient scatter of `date` on the device: d_table [n_dates][n_states][K] and d_status [n_dates] as in mcx_lsm_run, entry
 * date_index).  No host round trip per date. */
int  mcx_lsm_solve(mcx_handle* h, mcx_book* book, int32_t product, const double* d_moments, const mcx_lsm_date* date,
                   int32_t date_index, double* d_table, int32_t* d_status, void* stream);

/* Product-batched LSM step (books of thousands of products: tests/exposure_tests/cva_perfprmance_large_netting_set.py): the
 * step of mcx_lsm_step for n_jobs products with the SAME number of exercise states in one launch.  Job j uses the cashflow
 * cache block d_W + w_offset (layout [n_states][ld_w]) and returns its moments in h_moments[j * NM .. ), NM = (2K-1) + S*K
 * (host memory; the call synchronises).  Replaces the per-product loop of controller.py:289-291. */
typedef struct {
    int32_t product, roll_begin, roll_end, num_atom, x_atom, reserved;
    int64_t w_offset;
    double  shift, scale;
} mcx_lsm_job;
int  mcx_lsm_step_batch(mcx_handle* h, const mcx_book* book, const mcx_lsm_job* h_jobs, int32_t n_jobs, int32_t n_states,
                        const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                        double* h_moments, int32_t flags, void* stream);     /* w_len: doubles in d_W (bounds of w_offset) */
/* The same step with the moments left on the device (d_moments [n_jobs][NM], stream-ordered), and the K x K solves + coefficient
 * scatter of all its systems on the device: a big book's backward induction is then (step, [all-reduce], solve) pairs enqueued
 * back to back — ~600 of them for the reference's 5,000-product book — instead of a host round trip with a batched numpy solve per
 * step.  d_flag: one int32 the caller zeroes once; set to 1 if any system was numerically singular (its coefficients are not
 * written: repeat with mcx_lsm_step_batch and a host solver).  mcx_book_get_coeffs reads coefficients back (synchronises). */
typedef struct {
    double  shift, scale, x0;     /* basis map z = (x - shift) * scale; x0: the common x of a degenerate (rank-1) system */
    int64_t coeff_off[2];         /* where the [n_states][K] block goes in the book's coefficient array (-1: nowhere)     */
    int32_t degenerate, reserved;
} mcx_lsm_solve_job;
int  mcx_lsm_step_batch_dev(mcx_handle* h, const mcx_book* book, const mcx_lsm_job* h_jobs, int32_t n_jobs, int32_t n_states,
                            const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                            double* d_moments, int32_t flags, void* stream);
int  mcx_lsm_solve_batch(mcx_handle* h, mcx_book* book, const mcx_lsm_solve_job* h_jobs, int32_t n_jobs, int32_t n_states,
                         const double* d_moments, int32_t* d_flag, void* stream);
/* The whole product-batched backward induction in ONE call (replaces the Python loop of controller.py:289-383 over products and
 * dates for books of thousands of products): step t runs the jobs [h_step_begin[t], h_step_begin[t+1]) of h_jobs / h_solve — all of
 * h_step_states[t] exercise states — as mcx_lsm_step_batch_dev + mcx_lsm_solve_batch would (same kernels, same arithmetic per
 * (product, date)); the job tables are uploaded once and the steps enqueued back to back.  With a communicator on the handle
 * (mcx_comm_init) the moments of a step are all-reduced between roll and solve, stream-ordered; a rank without paths takes part with
 * zero moments.  *h_flag != 0: some system was numerically singular (its coefficients were not written: repeat with
 * mcx_lsm_step_batch and a host solver).  The call synchronises the stream. */
int  mcx_lsm_run_batch(mcx_handle* h, mcx_book* book, const mcx_lsm_job* h_jobs, const mcx_lsm_solve_job* h_solve,
                       const int32_t* h_step_begin, const int32_t* h_step_states, int32_t n_steps,
                       const double* d_paths, int64_t n_paths, int64_t ld, double* d_W, int64_t ld_w, int64_t w_len,
                       int32_t* h_flag, int32_t flags, void* stream);
int  mcx_book_get_coeffs(mcx_handle* h, const mcx_book* book, int64_t offset, int64_t count, double* h_out, void* stream);
/* coeffs[h_offsets[j] + q] = h_values[j * len + q], q < len, for n blocks in one call (the batched form of mcx_book_set_coeffs) */
int  mcx_book_set_coeffs_batch(mcx_handle* h, mcx_book* book, const int64_t* h_offsets, int32_t n, int32_t len,
                               const double* h_values, void* stream);

/* Exercise decisions of MCX_EV_EXERCISE events recorded (mode 1) or replayed (mode 2) by mcx_eval_book and mcx_lsm_step* / mcx_lsm_run
 * (mode 0: off): d_bits [n_events][ld] bytes, one per (event of the book, path); bit k = the decision of the LSM roll that started
 * in exercise state k (bit 0 in the main simulation).  The reference's tape carries no gradient through the boolean
 * `should_exercise` (bermudan_option.py:122-128), i.e. its sensitivities hold the exercise policy fixed: bump-and-revalue runs of
 * exercise products replay the base run's decisions so that no path flips its policy under the bump.  The fused kernels do not
 * record or replay (callers run the K1 / K2 / K4 plan while a mode is set).  ld >= the paths of the next call; n_rows = rows of d_bits,
 * >= the number of events of the book (checked here: every (event, path) byte a kernel touches is inside the buffer). */
int  mcx_book_set_exercise_replay(mcx_handle* h, mcx_book* book, int32_t mode, uint8_t* d_bits, int64_t n_rows, int64_t ld);

/* RNG of the Brownian-bridge barrier events (OPTION mode 5), set before mcx_eval_book / mcx_lsm_step*: one uniform per monitored
 * interval k and barrier b from Philox4x32-10 with key = seed and counter = (global path id = path_offset + i, k,
 * 0x80000000 | (2 * draw_id + b)).  The reference draws them from a numpy Generator on the host (barrier_option.py:52-53, 163,
 * 187); a parity run injects those numbers: h_inject[product] = device pointer to [2 * n_intervals][ld] uniforms (row 2k + b),
 * NULL entries / NULL table = Philox. */
int  mcx_book_set_bridge_rng(mcx_handle* h, mcx_book* book, uint64_t seed, uint64_t path_offset, const double* const* h_inject,
                             int64_t ld, void* stream);

/* K4 — reductions. Every output record is an mcx_acc (host memory, valid on return).
 * mcx_reduce_vector  : PVMetric on cfs (pv_metric.py:17-18)                                   h_out[1]
 * mcx_reduce_profiles: EPE / ENE per metric date (epe_metric.py:11-16, ene_metric.py:11-16)   h_out[2*n_dates] = {pos_m, neg_m}
 * mcx_reduce_cva     : sum_{m<n_dates-1} relu(u_m) * S(0,t_m) * (1 - S(t_m,t_{m+1})) * (1-R)   (cva_metric.py:62-100)  h_out[1]
 *                      surv_atoms / cond_atoms: [n_dates-1] atom ids in `book`. */
int  mcx_reduce_vector(mcx_handle* h, const double* d_x, int64_t n_paths, mcx_acc* h_out, void* stream);
int  mcx_reduce_profiles(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                         mcx_acc* h_out, void* stream);
int  mcx_reduce_cva(mcx_handle* h, const mcx_book* book, const mcx_unsecured_desc* u,
                    const int32_t* h_surv_atoms, const int32_t* h_cond_atoms, double recovery,
                    const double* d_expo_ns, const double* d_paths, int64_t n_paths, int64_t ld_expo, int64_t ld_paths,
                    mcx_acc* h_out, void* stream);
/* materialise unsecured exposures for pluggable metrics: d_out [n_dates][ld_out] */
int  mcx_unsecured(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                   double* d_out, int64_t ld_out, void* stream);

/* K5 — one radix-select pass for exact order statistics (pfe_metric.py:49-73 without sorting).
 * key(x) = order-preserving uint64 image of the double. For each metric date m and each of n_sel selections j the pass
 * histograms the `bits`-wide digit at `shift` of every unsecured exposure whose key matches prefix[m*n_sel+j] on the bits
 * above (shift+bits).  d_hist [n_dates][n_sel][1<<bits] uint64 counters (device, zeroed by the call, all-reducible). */
int  mcx_select_hist(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                     int32_t n_sel, const uint64_t* h_prefix, int32_t shift, int32_t bits,
                     uint64_t* d_hist, void* stream);
/* The same digit pass with the prefixes already on the device, and the step between two passes on the device: from the
 * (all-reduced) histogram pick, per (date, selection), the bin b holding the remaining rank (d_rem [n_dates][n_sel], starts as the
 * 0-based global ranks), d_prefix |= b << shift, d_rem -= count below b.  A select is then six (hist, [all-reduce], narrow) pairs
 * enqueued back to back and ONE copy of the final prefixes (the order statistics as order-preserving uint64 keys): no host round
 * trip per pass (pfe_metric.py:49-73 sorts N values per date on the host). */
int  mcx_select_hist_dev(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                         int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, void* stream);
int  mcx_select_narrow(mcx_handle* h, int32_t n_dates, int32_t n_sel, const uint64_t* d_hist, int32_t shift, int32_t bits,
                       uint64_t* d_prefix, int64_t* d_rem, void* stream);
/* Bracket pass of the select.  The six digit passes above read the exposure matrix six times; the order statistic of a date lies
 * almost surely between two order statistics of a SAMPLE of its paths (the caller selects those on a prefix of the paths with the
 * functions above).  mcx_select_bracket reads the matrix ONCE: d_below[m] = number of paths with exposure < d_lo[m],
 * d_count[m] = number inside [d_lo[m], d_hi[m]], whose values are gathered into d_cand[m][0 .. min(count, cap)) in no particular
 * order.  mcx_select_hist_rows is the digit pass over such a plain [n_rows][ld] tensor whose row m holds d_row_n[m] values
 * (clipped to ld); the caller checks below <= rank < below + count over all GPUs and d_count <= cap, and falls back to the digit
 * passes over the matrix for dates that fail — exactness never rests on the bracket.  A date whose candidates could not all be
 * staged (far more paths inside the bracket than a sample suggests: every path at one value) has MCX_SELECT_LOST added to its
 * count: count & (MCX_SELECT_LOST - 1) stays the exact number inside, the candidates are incomplete.  (metrics/pfe_metric.py:49-73) */
#define MCX_SELECT_LOST (1ull << 44)
int  mcx_select_bracket(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                        const double* d_lo, const double* d_hi, uint64_t* d_below, uint64_t* d_count, double* d_cand, int64_t cap,
                        void* stream);
int  mcx_select_hist_rows(mcx_handle* h, const double* d_rows, int32_t n_rows, int64_t ld, const uint64_t* d_row_n,
                          int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, void* stream);

/* Multi-GPU exchange (paths shard over the GPUs of a node, one process per GPU; the only data that crosses
 * GPUs are accumulator records, LSM moments and select histograms).  RCCL over xGMI, loaded at run time (librccl.so.1 — the
 * library has no link-time dependency on it).  A caller without torch.distributed creates the id on rank 0
 * (mcx_comm_unique_id), ships the 128 bytes to the other ranks by any means, and every rank calls mcx_comm_init.
 * mcx_allreduce_f64 sums d_buf[0..n) in place over the ranks on `stream`; mcx_allgather_f64 gathers n doubles per rank into
 * d_out [n_ranks][n].  One communicator per handle. */
#define MCX_COMM_ID_BYTES 128
int  mcx_comm_unique_id(mcx_handle* h, void* out_id /* MCX_COMM_ID_BYTES */);
int  mcx_comm_init(mcx_handle* h, int32_t n_ranks, int32_t rank, const void* id /* MCX_COMM_ID_BYTES */);
int  mcx_comm_destroy(mcx_handle* h);
int  mcx_allreduce_f64(mcx_handle* h, double* d_buf, int64_t n, void* stream);
int  mcx_allgather_f64(mcx_handle* h, const double* d_in, double* d_out, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCX_H */
