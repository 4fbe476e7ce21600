// mcx_internal.h — private layout of libmcx_hip.so (gfx950 / CDNA4 only).
// Device-side tables are flattened copies of the descriptors of include/mcx.h, built once at *_create time.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mcx.h"

#define MCX_WAVE 64           // CDNA4 wavefront
#define MCX_BLOCK 256         // 4 waves: one per SIMD of a CU
#define MCX_MAX_PARTIAL_BLOCKS 1024

struct mcx_handle {
    int device;
    int n_cu;
    std::string err;
    double* d_ws;          // reduction workspace (partials)
    size_t ws_bytes;
    void* h_pinned;        // pinned staging for small device->host results
    void* d_pinned_alias;  // the same buffer as the device addresses it (hipHostGetDevicePointer), nullptr if not mapped
    size_t pinned_bytes;
    hipDeviceProp_t prop;
    // small-descriptor staging ring: host arrays of a call (row tables, prefixes, atom ids) travel pinned -> device without a
    // hipMalloc / hipFree per call (mcx_stage_small)
    unsigned char* d_small;
    unsigned char* h_small;
    size_t small_bytes, small_cursor;
    // the ring is used as two halves: moving into the other half drains the stream first, so what an API call staged stays
    // valid until another half-ring of descriptors has been staged after it (also for kernels the call has not launched yet)
    hipStream_t small_stream;      // the stream of the last staging call (a change of stream drains the previous one)
    bool small_stream_valid;
    void* d_acc;           // device image of the accumulator records of a reduction call (pinned_bytes large)
    void* scratch[4];      // device scratch buffers of the entry points, grown on demand and freed with the handle
    size_t scratch_bytes[4];
    void* comm;            // ncclComm_t of mcx_comm_init (mcx_comm.hip) or nullptr
    int comm_ranks, comm_rank;
};

#define MCX_FAIL(h, code, ...)                                   \
    do {                                                         \
        char _b[512];                                            \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
        (h)->err = _b;                                           \
        return (code);                                           \
    } while (0)

#define MCX_HIP(h, expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) MCX_FAIL(h, -100 - (int)_e, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ---- flattened device records ---------------------------------------------------------------------------------------
struct DevAtom {            // value = a + d*x + b*exp(c0 + c1*x), x = paths[(t_idx*D + col)*ld + i]
    int32_t t_idx, col;
    double a, d, b, c0, c1;
};

struct DevTerm {
    double w;
    DevAtom atom;
    int32_t den;            // -1 or index into DevBook::atoms
    int32_t pad;
};

struct DevEvent {
    int32_t kind, flags;    // flags bit0: exposure row is accumulated (+=) instead of stored
    int32_t term_begin, term_end;
    int32_t coeff_off, row;
    int32_t netting_set, pad;       // pad: 1 + index of the event's value polynomial (DevVPoly), 0 = none
    double strike, sign;
    double aux[4];
    DevAtom num;
    DevAtom x;
};

// std::vector whose resize() does not zero its elements (4 x 10^5 events of a 5,000-product book are 64 MB: the zeroing pass and
// the page faults of the first touch belong to the threads that fill the chunks)
template <class T>
struct mcx_noinit_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = mcx_noinit_alloc<U>; };
    template <class U, class... A> void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<DevEvent, mcx_noinit_alloc<DevEvent>> DevEventVec;

// value polynomial of an event (mcx_vpoly.hip): sum of the event's terms = p(t), t = fma(x, ih, ms), x = paths[t_idx][col], valid
// for lo <= x <= hi; n_blk blocks of MCX_VPOLY_BLK coefficients at coef_off, highest power first (zero-padded in front)
#define MCX_VPOLY_BLK 4
struct DevVPoly {
    double lo, hi, ms, ih;
    int32_t coef_off, n_blk, t_idx, col;
};
struct VPolyBlk { double c[MCX_VPOLY_BLK]; };

struct DevProduct {
    int32_t ev_begin, ev_end, cf_begin, cf_end;
    int32_t netting_set, init_state, n_states, flags;
};

struct mcx_book {
    int n_atoms, n_terms, n_events, n_products, n_netting_sets, n_expo_rows, n_basis, n_coeffs, want_cfs, want_expo;
    int n_state;                   // inferred: max col + 1 is NOT used; D comes from the sim (passed via paths layout)
    DevAtom* d_atoms;
    DevTerm* d_terms;
    DevEvent* d_events;
    DevProduct* d_products;
    double* d_coeffs;
    struct DevBridge* d_bridge;      // RNG state of Brownian-bridge barrier events (mcx_device.h), one device struct per book
    const double** d_bridge_inject;  // [n_products] device table of injected-uniform pointers (nullptr entries allowed)
    std::vector<mcx_atom> h_atoms;
    DevEventVec h_events;          // (default-initialising allocator: mcx_book_create fills it from several threads, first touch included)
    std::vector<DevTerm> h_terms;
    std::vector<int32_t> h_event_t_idx;
    std::vector<int32_t> h_event_num_atom, h_event_x_atom, h_term_atom;
    std::vector<DevProduct> h_products;
    // exercise decisions recorded (1) / replayed (2) by K2 and K3 (mcx_book_set_exercise_replay): [n_events][ex_ld] bytes
    int ex_mode;
    uint8_t* d_ex_bits;
    int64_t ex_ld;
    // value polynomials (mcx_book_collapse_values)
    DevVPoly* d_vpoly;
    double* d_vcoef;
    std::vector<DevVPoly> h_vpoly;
    std::vector<double> h_vcoef;
    std::vector<int32_t> h_event_vpoly;   // [n_events] index into h_vpoly or -1
    std::vector<double> vpoly_key;        // (tolerance, candidate events and ranges) of the last mcx_book_collapse_values
    std::vector<uint8_t> ns_has_writer;   // [n_netting_sets * n_expo_rows]
    bool expo_needs_memset;
    // event families the book contains, found once at mcx_book_create (mcx_eval_book picks its kernel specialisation from them: a
    // scan of the host events per call walked 64 MB for a 5,000-product book)
    bool has_barrier, has_exotic, has_exercise, has_bs_expo, has_den;
};

// derived per-step constants of the Euler maps, written by mcx_sim_create into otherwise unused entries of the DEVICE copy of
// the step table (aux [step][slot][MCX_AUX]): the step is then a few fused multiply-adds per state instead of a chain of
// wave-uniform products re-evaluated by every lane.  Black-Scholes: C0 = r dt, C2 = sigma sqrt(dt); Vasicek: C0 = a theta dt,
// C1 = -a dt, C2 = sigma sqrt(dt); CIR++: C0 = kappa theta dt, C1 = -kappa dt, C2 = sigma sqrt(dt).
#define MCX_AUX_C0 4
#define MCX_AUX_C1 5
#define MCX_AUX_C2 6
struct mcx_aux_drv { double c0, c1, c2; };      // aux[MCX_AUX_C0 .. MCX_AUX_C2] as one scalar load

struct mcx_sim {
    mcx_sim_desc desc;             // host copy (pointer members unused after create)
    mcx_step* d_steps;
    double* d_chol;
    double* d_aux;
    std::vector<mcx_step> h_steps; // host copy of the sub-step table (per-step records derived from it: kt_tangent.hip)
    int n_state_total;
    int state_dim[MCX_MAX_SLOTS];
};

// ---- device helpers -------------------------------------------------------------------------------------------------
// Read-only, wave-uniform table loads.  Kernel tables (sub-step table, Cholesky factors, event program, coefficients) are
// indexed by wave-uniform counters; reading them through the CONSTANT address space lets the backend issue scalar loads
// (s_load_dwordx*, scalar cache -> SGPRs) instead of 64 identical vector loads that occupy VGPRs.  (`__restrict__` on
// struct members does not give the alias information the backend needs to do this on its own.)
#define MCX_KONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ T ldk(const T* p)
{
    return *(const MCX_KONST T*)(uintptr_t)p;
}
template <class T>
__device__ __forceinline__ T ldk_struct(const T* p)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized records only");
    T out;
    uint32_t* o = (uint32_t*)&out;
    const MCX_KONST uint32_t* s = (const MCX_KONST uint32_t*)(uintptr_t)p;
#pragma unroll
    for (unsigned q = 0; q < sizeof(T) / 4; ++q) o[q] = s[q];
    return out;
}

#include "mcx_math.h"

__device__ __forceinline__ double dev_atom(const DevAtom& a, const double* __restrict__ paths, int64_t D, int64_t ld, int64_t i)
{
    double x = 0.0;
    if (a.col >= 0) x = paths[((int64_t)a.t_idx * D + a.col) * ld + i];
    double v = fma(a.d, x, a.a);
    if (a.b != 0.0) v = fma(a.b, mcx_exp(fma(a.c1, x, a.c0)), v);
    return v;
}

// PPL paths of a lane through one (wave-uniform) atom; exponentials from the block's LDS copy of the 2^(j/128) table
template <int PPL>
__device__ __forceinline__ void dev_atoms(const DevAtom& a, const double* __restrict__ paths, int64_t D, int64_t ld, const int64_t (&i)[PPL],
                                          double (&v)[PPL], const double* __restrict__ etab, const mcx_expq_coef& ec)
{
    double x[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) x[q] = a.col >= 0 ? paths[((int64_t)a.t_idx * D + a.col) * ld + i[q]] : 0.0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        v[q] = fma(a.d, x[q], a.a);
        if (a.b != 0.0) v[q] = fma(a.b, mcx_exp_tab(fma(a.c1, x[q], a.c0), etab, ec), v[q]);
    }
}
// the terms of one event mostly read the SAME state column of the SAME date (e.g. 64 zero-bond prices of one short rate):
// keep the last (date, column) -> value in registers instead of re-issuing the global load for every term
struct AtomCache { int t_idx, col; double x; };
__device__ __forceinline__ double dev_atom_cached(const DevAtom& a, const double* __restrict__ paths, int64_t D, int64_t ld, int64_t i,
                                                  AtomCache& c)
{
    double x = 0.0;
    if (a.col >= 0) {
        if (a.t_idx != c.t_idx || a.col != c.col) {           // wave-uniform test
            c.x = paths[((int64_t)a.t_idx * D + a.col) * ld + i];
            c.t_idx = a.t_idx; c.col = a.col;
        }
        x = c.x;
    }
    double v = fma(a.d, x, a.a);
    if (a.b != 0.0) v = fma(a.b, mcx_exp(fma(a.c1, x, a.c0)), v);
    return v;
}

// dev_atom_cached with the exponential from the block's LDS table (mcx_exp_tab)
__device__ __forceinline__ double dev_atom_cached_tab(const DevAtom& a, const double* __restrict__ paths, int64_t D, int64_t ld, int64_t i,
                                                      AtomCache& c, const double* __restrict__ etab, const mcx_expq_coef& ec)
{
    double x = 0.0;
    if (a.col >= 0) {
        if (a.t_idx != c.t_idx || a.col != c.col) {           // wave-uniform test
            c.x = paths[((int64_t)a.t_idx * D + a.col) * ld + i];
            c.t_idx = a.t_idx; c.col = a.col;
        }
        x = c.x;
    }
    double v = fma(a.d, x, a.a);
    if (a.b != 0.0) v = fma(a.b, mcx_exp_tab(fma(a.c1, x, a.c0), etab, ec), v);
    return v;
}


// value polynomial of an event at x (mcx_vpoly.hip): wave-uniform record and coefficients -> scalar loads, the next block of
// coefficients in flight while the current one is consumed (the pool ends with one spare block).  Caller: x inside [vp.lo, vp.hi].
__device__ __forceinline__ double dev_vpoly(const DevVPoly& vp, const double* __restrict__ vcoef, double x)
{
    const double t = fma(x, vp.ih, vp.ms);
    const VPolyBlk* __restrict__ blk = (const VPolyBlk*)(vcoef + vp.coef_off);
    VPolyBlk c = ldk_struct(blk);
    double p = 0.0;
    for (int k = 0; k < vp.n_blk; ++k) {
        const VPolyBlk nx = ldk_struct(blk + k + 1);
#pragma unroll
        for (int j = 0; j < MCX_VPOLY_BLK; ++j) p = fma(p, t, c.c[j]);
        c = nx;
    }
    return p;
}

// exercise decision with optional record / replay (mcx_book_set_exercise_replay): `cell` = the (event, path) byte, `bit` = the
// hypothetical start state of the LSM roll (0 in the main simulation)
__device__ __forceinline__ bool dev_exercise_decision(bool decided, int s, int mode, uint8_t* __restrict__ cell, int bit)
{
    if (mode == 2) return ((*cell >> bit) & 1) && s > 0;
    if (mode == 1) *cell = (uint8_t)((*cell & ~(1u << bit)) | ((unsigned)decided << bit));
    return decided;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, MCX_WAVE);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, MCX_WAVE));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, MCX_WAVE));
    return v;
}

// sum over a 256-thread block; result valid in thread 0. `lds` must hold >= 4 doubles; includes the barriers it needs.
__device__ __forceinline__ double block_sum(double v, double* lds)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (MCX_WAVE - 1), w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + MCX_WAVE - 1) >> 6;
        for (int q = 0; q < nw; ++q) r += lds[q];
    }
    return r;
}

static inline int mcx_grid_for(int64_t n, int block, int cap)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// unsecured-exposure descriptor copied by value into kernel arguments (K4/K5)
#define MCX_MAX_METRIC_DATES 512
struct DevUnsec {
    int32_t n_dates, collateralized;
    double threshold;
    const int32_t* row;       // device
    const int32_t* delayed;   // device or nullptr
};

__device__ __forceinline__ double dev_thr(double x, double h)
{
    if (h == 0.0) return x;
    return x > h ? x - h : (x < -h ? x + h : 0.0);
}

__device__ __forceinline__ double dev_unsec(const DevUnsec& u, const double* __restrict__ expo, int64_t ld, int m, int64_t i)
{
    double e = expo[(int64_t)ldk(u.row + m) * ld + i];
    if (!u.collateralized) return dev_thr(e, u.threshold);
    double coll = 0.0;
    if (u.delayed) {
        int dm = ldk(u.delayed + m);
        if (dm >= 0) coll = dev_thr(expo[(int64_t)dm * ld + i], u.threshold);
    }
    return e - coll;
}

// barrier indicator of one barrier level (barrier_option.py:77-125): fuzzy "max below" / "min above" with the default
// eps = 0.05 of maths.py:8; type 1 up-and-out, 2 down-and-out, 3 up-and-in, 4 down-and-in
__device__ __forceinline__ double dev_barrier_ind(int type, double b, double mx, double mn)
{
    const double below = fmin(fmax((b - mx + 0.05) / 0.1, 0.0), 1.0), above = fmin(fmax((mn - b + 0.05) / 0.1, 0.0), 1.0);
    return type == 1 ? below : type == 2 ? above : type == 3 ? 1.0 - below : 1.0 - above;
}

// host-side helpers implemented in mcx_api.hip
// device scratch buffer `slot` (0..3) of at least `bytes` bytes, owned by the handle: no hipMalloc / hipFree per call (a buffer
// is re-allocated only when a call needs more than any call before it); nullptr + handle error on failure
void* mcx_scratch(mcx_handle* h, int slot, size_t bytes);
static inline double* mcx_path_scratch(mcx_handle* h, size_t n) { return (double*)mcx_scratch(h, 0, sizeof(double) * n); }
// copy `bytes` of host data to device memory that stays valid for the kernels enqueued on `s` by the current API call (a ring:
// wrapping synchronises the stream first); returns nullptr and sets the handle's error on failure
void* mcx_stage_small(mcx_handle* h, const void* src, size_t bytes, hipStream_t s);
// the same for tables that may exceed the ring (job tables of batched steps): through the ring when they fit, otherwise copied into
// `fallback` (a device buffer of >= bytes) and completed before returning — `src` may be freed as soon as the call returns either way
const void* mcx_upload_call_data(mcx_handle* h, const void* src, size_t bytes, void* fallback, hipStream_t s);
int mcx_upload_unsec(mcx_handle* h, const mcx_unsecured_desc* u, DevUnsec* out, int32_t** d_tmp, hipStream_t s);
int mcx_finish_acc(mcx_handle* h, const double* d_partials, int n_records, int n_blocks, double n_paths,
                   const double* d_shifts, mcx_acc* h_out, hipStream_t s);
