// k5_select.hip — K5: one digit pass of an exact radix select (PFE without sorting).
//
// The reference sorts all N exposures per date (`torch.sort`, metrics/pfe_metric.py:61-66) to read x_(ceil(qN)-1) and its
// two neighbours.  Here each pass histograms one digit of the order-preserving uint64 image of the doubles; the host
// walks the digits (6 passes of 11/11/11/11/11/9 bits), narrowing up to three rank prefixes per date at once.  Histograms
// are integer counters: exact, order-independent and all-reducible across GPUs (no path data leaves a GPU).
//
// Per block: an LDS histogram [n_sel][2^bits] of u32 (<= 24 KB), wave-coalesced 512-B row reads, LDS integer atomics,
// then one global u64 atomic per non-empty bin.
#include "mcx_internal.h"

namespace {

__device__ __forceinline__ uint64_t dev_key(double x)
{
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}

#define K5_MAX_SEL 4
struct K5Prefix { uint64_t p[K5_MAX_SEL]; };

// PLAIN: `expo` is a plain [rows][ld] tensor whose row m holds row_n[m] values (the candidates gathered by k5_bracket)
template <bool PLAIN>
__global__ __launch_bounds__(MCX_BLOCK) void k5_hist(const DevUnsec u, const double* __restrict__ expo, int64_t n_in, int64_t ld,
                                                     int n_sel, const uint64_t* __restrict__ prefix, int shift, int bits,
                                                     unsigned long long* __restrict__ hist, const unsigned long long* __restrict__ row_n)
{
    extern __shared__ uint32_t lh[];                  // [n_sel][1<<bits]
    const int m = blockIdx.y;
    const int nb = 1 << bits;
    int64_t n = n_in;
    if (PLAIN) { const int64_t rn = (int64_t)row_n[m]; n = rn < n ? rn : n; }
    for (int q = threadIdx.x; q < n_sel * nb; q += MCX_BLOCK) lh[q] = 0;
    K5Prefix pf;
    for (int j = 0; j < K5_MAX_SEL; ++j) pf.p[j] = j < n_sel ? prefix[m * n_sel + j] : 0;
    __syncthreads();
    const int hi = shift + bits;
    const int lane = threadIdx.x & 63;
    // four elements per lane and iteration: the four loads are in flight together (one 8-byte load per lane and iteration left
    // the pass latency-bound at ~1.4 TB/s)
    constexpr int UN = 4;
    const int64_t stride = (int64_t)gridDim.x * MCX_BLOCK;
    for (int64_t i0 = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i0 < n; i0 += UN * stride) {
        uint64_t kk[UN];
        bool in[UN];
#pragma unroll
        for (int w = 0; w < UN; ++w) {
            const int64_t i = i0 + w * stride;
            in[w] = i < n;
            const int64_t ii = in[w] ? i : (n > 0 ? n - 1 : 0);
            kk[w] = dev_key(PLAIN ? expo[(int64_t)m * ld + ii] : dev_unsec(u, expo, ld, m, ii));
        }
#pragma unroll
        for (int w = 0; w < UN; ++w) {
            const uint64_t k = kk[w];
            const uint32_t digit = (uint32_t)((k >> shift) & (uint64_t)(nb - 1));
#pragma unroll                                      // (compile-time j: a run-time index would put the prefixes into scratch memory)
            for (int j = 0; j < K5_MAX_SEL; ++j) {
                if (j >= n_sel) break;
                // neighbouring ranks (q-1, q, q+1) share their prefix until the very last digits: count such a selection once
                if (j > 0 && (hi >= 64 || (pf.p[j] >> hi) == (pf.p[0] >> hi))) continue;
                const bool match = in[w] && (hi >= 64 || (k >> hi) == (pf.p[j] >> hi));
                // Exposures of one date share their leading bits, so in the high-order passes most lanes of a wave hit the SAME
                // bin: aggregate per distinct digit with ballots (one LDS atomic per distinct digit per wave) instead of 64
                // serialised same-address atomics.
                unsigned long long todo = __ballot(match);
#pragma unroll 1
                for (int round = 0; round < 2 && todo; ++round) {          // the (at most two) dominant digits of the wave
                    const int leader = __ffsll((long long)todo) - 1;
                    const uint32_t d = (uint32_t)__shfl((int)digit, leader, MCX_WAVE);
                    const unsigned long long same = __ballot(match && digit == d) & todo;
                    if (lane == leader) atomicAdd(&lh[j * nb + d], (uint32_t)__popcll(same));
                    todo &= ~same;
                }
                if ((todo >> lane) & 1ull) atomicAdd(&lh[j * nb + digit], 1u);   // scattered remainder: distinct bins, no conflict
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_sel * nb; q += MCX_BLOCK) {
        const int j = q / nb;
        uint64_t pj = pf.p[0];
#pragma unroll
        for (int w = 1; w < K5_MAX_SEL; ++w) pj = (j == w) ? pf.p[w] : pj;
        const bool shared = j > 0 && (hi >= 64 || (pj >> hi) == (pf.p[0] >> hi));
        const uint32_t c = lh[shared ? q - j * nb : q];
        if (c) atomicAdd(&hist[(int64_t)m * n_sel * nb + q], (unsigned long long)c);
    }
}

// ---- bracket pass ------------------------------------------------------------------------------------------------------------
// Six digit passes read the exposure matrix six times (11.6 GB for the 1.94 GB of BASELINE config 5).  The order statistic of a
// date lies, with overwhelming probability, between two order statistics of a SAMPLE of its paths (the caller selects those on
// the first 65,536 paths — paths are exchangeable): ONE pass over the matrix counts the values below that bracket and gathers
// the ~1-2 % inside it; the exact radix select then runs on the gathered candidates only.  Exactness does not rest on the
// bracket: the caller checks below <= rank < below + inside over all GPUs and falls back to the digit passes for a date that
// fails (or whose candidates overflowed `cap`, e.g. a date on which every path has the same exposure).
#define K5_STAGE 512                                   // doubles staged per wave
#define K5_FLUSH_EVERY 8                              // iterations between two looks at the wave's stage
// VEC: a plain (uncollateralised) exposure row with an even leading dimension — 16-byte loads, two paths per lane and load
template <bool VEC>
__global__ __launch_bounds__(MCX_BLOCK) void k5_bracket(const DevUnsec u, const double* __restrict__ expo, int64_t n, int64_t ld,
                                                        const double* __restrict__ lo, const double* __restrict__ hi,
                                                        unsigned long long* __restrict__ below, unsigned long long* __restrict__ count,
                                                        double* __restrict__ cand, int64_t cap)
{
    // The scalar unit is shared by the four SIMDs of a CU: a ballot + popcount per path (the first version of this kernel, ~80 SALU
    // instructions per 512 paths) capped the pass at 3 TB/s.  Here the comparisons stay in the vector unit — a lane keeps its own
    // 32-bit `below` counter and a bit mask of its paths inside the bracket; only a lane with a hit (~2 % of the paths) takes a slot
    // in its wave's LDS stage with an LDS atomic.  Every K5_FLUSH_EVERY iterations the wave moves its stage to the date's candidate
    // row: one global atomic reserves the space, the copy out is coalesced.  A stage that overflowed (> K5_STAGE hits in 8
    // iterations: a date whose paths sit at one value) adds MCX_SELECT_LOST to the date's count, which sends the date to the digit passes.
    __shared__ double stage_all[4 * K5_STAGE];
    __shared__ unsigned stage_n[4];
    __shared__ unsigned long long red[4];
    const int m = blockIdx.y;
    const double l = ldk(lo + m), h = ldk(hi + m);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double* __restrict__ stage = stage_all + wv * K5_STAGE;
    if (lane == 0) stage_n[wv] = 0;
    unsigned n_below = 0;                                // per lane: < 2^32 paths per lane
    constexpr int UN = 4, W = VEC ? 2 : 1;               // loads in flight per lane, paths per load
    const int64_t stride = (int64_t)gridDim.x * MCX_BLOCK * W;
    auto flush = [&]() {                                 // wave-uniform
        const unsigned ns = stage_n[wv];                 // (LDS operations of one wave complete in order)
        if (ns == 0) return;
        const unsigned kept = ns < K5_STAGE ? ns : K5_STAGE;
        unsigned long long b0 = 0;
        // an overflowed stage lost candidates: the count stays exact, MCX_SELECT_LOST is added on top (the caller then redoes the date
        // by digit passes)
        if (lane == 0) b0 = atomicAdd(&count[m], (unsigned long long)ns + (ns > K5_STAGE ? MCX_SELECT_LOST : 0ull));
        b0 = (unsigned long long)__shfl((long long)b0, 0, MCX_WAVE) & (MCX_SELECT_LOST - 1ull);
        for (unsigned q = (unsigned)lane; q < kept; q += MCX_WAVE)
            if ((int64_t)(b0 + q) < cap) cand[(int64_t)m * cap + (int64_t)(b0 + q)] = stage[q];
        if (lane == 0) stage_n[wv] = 0;
    };
    const double thr = u.threshold;
    const double* __restrict__ row = VEC ? expo + (int64_t)ldk(u.row + m) * ld : nullptr;
    // the loads of the next group of paths are in flight while this group is compared and staged
    auto load = [&](int64_t i0, double (&x)[UN * W], unsigned& live) {
        live = 0;
#pragma unroll
        for (int w = 0; w < UN; ++w) {
            const int64_t i = i0 + (int64_t)lane * W + w * stride;
            if constexpr (VEC) {
                const bool a = i < n, b = i + 1 < n;
                const mcx_d2 v = b ? *(const mcx_d2*)(row + i) : mcx_d2{a ? row[i] : 0.0, 0.0};
                x[2 * w] = dev_thr(v.x, thr); x[2 * w + 1] = dev_thr(v.y, thr);
                live |= (a ? 1u : 0u) << (2 * w) | (b ? 1u : 0u) << (2 * w + 1);
            } else {
                const bool a = i < n;
                x[w] = dev_unsec(u, expo, ld, m, a ? i : n - 1);
                live |= (a ? 1u : 0u) << w;
            }
        }
    };
    const int64_t first = ((int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x - lane) * W;      // wave-uniform
    double xn[UN * W];
    unsigned live_n = 0;
    if (first < n) load(first, xn, live_n);
    int since = 0;
    for (int64_t i0 = first; i0 < n; i0 += UN * stride) {
        double x[UN * W];
#pragma unroll
        for (int w = 0; w < UN * W; ++w) x[w] = xn[w];
        const unsigned live = live_n;
        if (i0 + UN * stride < n) load(i0 + UN * stride, xn, live_n);
        unsigned hits = 0;
#pragma unroll
        for (int w = 0; w < UN * W; ++w) {
            const bool lv = (live >> w) & 1u;
            const bool lt = lv && x[w] < l;
            n_below += lt ? 1u : 0u;
            hits |= (lv && !lt && x[w] <= h) ? (1u << w) : 0u;
        }
        if (hits) {                                      // ~2 % of the paths; ONE LDS atomic per lane with hits (one round trip per iteration)
            unsigned slot = atomicAdd(&stage_n[wv], (unsigned)__popc(hits));
#pragma unroll
            for (int w = 0; w < UN * W; ++w)
                if ((hits >> w) & 1u) {
                    if (slot < K5_STAGE) stage[slot] = x[w];
                    ++slot;
                }
        }
        // (mid-run: only a stage that is filling up is moved; global atomics on the date's counter serialise — 62,000 of them, one
        //  per wave at its end, were a third of the pass — so the usual case is ONE reservation per block, below)
        if (++since == K5_FLUSH_EVERY) { since = 0; if (stage_n[wv] > K5_STAGE / 2) flush(); }
    }
    // end of the block: one reservation for the four waves' stages
    __shared__ unsigned long long blk_base;
    __syncthreads();
    {
        unsigned cnt[4], lost = 0, tot = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const unsigned c = stage_n[q]; lost |= c > K5_STAGE ? 1u : 0u; cnt[q] = c; tot += c; }
        if (tot) {
            if (threadIdx.x == 0) blk_base = atomicAdd(&count[m], (unsigned long long)tot + (lost ? MCX_SELECT_LOST : 0ull)) & (MCX_SELECT_LOST - 1ull);
            __syncthreads();
            unsigned long long b0 = blk_base;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (q < wv) b0 += cnt[q];            // (true counts: a lost stage leaves a gap, the date is redone anyway)
            const unsigned kept = cnt[wv] < K5_STAGE ? cnt[wv] : K5_STAGE;
            for (unsigned q = (unsigned)lane; q < kept; q += MCX_WAVE)
                if ((int64_t)(b0 + q) < cap) cand[(int64_t)m * cap + (int64_t)(b0 + q)] = stage[q];
        }
    }
    // paths below the bracket: wave shuffle + one global atomic per block
    unsigned long long nb = n_below;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) nb += __shfl_down(nb, off, MCX_WAVE);
    if (lane == 0) red[wv] = nb;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long tot = red[0] + red[1] + red[2] + red[3];
        if (tot) atomicAdd(&below[m], tot);
    }
}

// After a digit pass (and the all-reduce of its histograms over the ranks): per (date, selection) the bin b that holds the
// remaining rank r — the number of bins whose inclusive cumulative count is <= r — then prefix |= b << shift and
// r -= (count below bin b).  One block per (date, selection); integer arithmetic only.  Keeps the six passes of a select on the
// device: the host reads the final prefixes once instead of a histogram per pass.
__global__ __launch_bounds__(MCX_BLOCK) void k5_narrow(const unsigned long long* __restrict__ hist, int bits, int shift,
                                                       unsigned long long* __restrict__ prefix, long long* __restrict__ rem)
{
    const int nb = 1 << bits, per = (nb + MCX_BLOCK - 1) / MCX_BLOCK;
    const unsigned long long* h = hist + (int64_t)blockIdx.x * nb;
    __shared__ long long part[MCX_BLOCK];
    __shared__ long long red_cnt[MCX_BLOCK], red_below[MCX_BLOCK];
    const int t = threadIdx.x, b0 = t * per;
    long long local = 0;
    for (int q = 0; q < per; ++q) if (b0 + q < nb) local += (long long)h[b0 + q];
    part[t] = local;
    __syncthreads();
    for (int off = 1; off < MCX_BLOCK; off <<= 1) {            // inclusive scan of the 256 partial sums
        const long long add = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    long long cum = part[t] - local;                           // count below this thread's first bin
    const long long r = rem[blockIdx.x];
    long long cnt = 0, below = 0;
    for (int q = 0; q < per; ++q) {
        if (b0 + q >= nb) break;
        cum += (long long)h[b0 + q];
        if (cum <= r) { ++cnt; below = cum; }                  // cum is non-decreasing: the last hit is the largest
    }
    red_cnt[t] = cnt; red_below[t] = below;
    __syncthreads();
    for (int off = MCX_BLOCK / 2; off > 0; off >>= 1) {
        if (t < off) {
            red_cnt[t] += red_cnt[t + off];
            red_below[t] = red_below[t] > red_below[t + off] ? red_below[t] : red_below[t + off];
        }
        __syncthreads();
    }
    if (t == 0) {
        prefix[blockIdx.x] |= (unsigned long long)red_cnt[0] << shift;
        rem[blockIdx.x] = r - red_below[0];
    }
}

}  // namespace

static int select_hist_impl(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                            int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, hipStream_t s);

extern "C" int mcx_select_hist_dev(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                                   int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, void* stream)
{
    if (!h || !u || !d_expo_ns || !d_prefix || !d_hist) return -1;
    return select_hist_impl(h, u, d_expo_ns, n_paths, ld, n_sel, d_prefix, shift, bits, d_hist, (hipStream_t)stream);
}

extern "C" int mcx_select_narrow(mcx_handle* h, int32_t n_dates, int32_t n_sel, const uint64_t* d_hist, int32_t shift, int32_t bits,
                                 uint64_t* d_prefix, int64_t* d_rem, void* stream)
{
    if (!h || !d_hist || !d_prefix || !d_rem) return -1;
    if (n_dates < 1 || n_sel < 1 || n_sel > K5_MAX_SEL || bits < 1 || bits > 11 || shift < 0 || shift + bits > 64)
        MCX_FAIL(h, -2, "mcx_select_narrow: bad selection geometry (n_dates=%d n_sel=%d shift=%d bits=%d)", n_dates, n_sel, shift, bits);
    hipLaunchKernelGGL(k5_narrow, dim3(n_dates * n_sel), dim3(MCX_BLOCK), 0, (hipStream_t)stream, (const unsigned long long*)d_hist, (int)bits,
                       (int)shift, (unsigned long long*)d_prefix, (long long*)d_rem);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_select_hist(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                               int32_t n_sel, const uint64_t* h_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, void* stream)
{
    if (!h || !u || !d_expo_ns || !h_prefix || !d_hist) return -1;
    if (n_sel < 1 || n_sel > K5_MAX_SEL || u->n_dates < 1) MCX_FAIL(h, -2, "mcx_select_hist: bad selection geometry (n_sel=%d)", n_sel);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t* d_prefix = (const uint64_t*)mcx_stage_small(h, h_prefix, sizeof(uint64_t) * (size_t)u->n_dates * n_sel, s);
    if (!d_prefix) return -100;
    return select_hist_impl(h, u, d_expo_ns, n_paths, ld, n_sel, d_prefix, shift, bits, d_hist, s);
}

static int select_hist_impl(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                            int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, hipStream_t s)
{
    if (n_sel < 1 || n_sel > K5_MAX_SEL || bits < 1 || bits > 11 || shift < 0 || shift + bits > 64)
        MCX_FAIL(h, -2, "mcx_select_hist: bad selection geometry (n_sel=%d shift=%d bits=%d)", n_sel, shift, bits);
    const size_t nh = (size_t)u->n_dates * n_sel * ((size_t)1 << bits);
    MCX_HIP(h, hipMemsetAsync(d_hist, 0, nh * sizeof(uint64_t), s));
    if (n_paths <= 0) return 0;
    DevUnsec du; int32_t* tmp = nullptr;
    int rc = mcx_upload_unsec(h, u, &du, &tmp, s);
    if (rc) return rc;
    // a block pays a fixed cost (zeroing and scanning its n_sel x 2^bits LDS bins): give it >= 32 elements per thread, and no
    // more blocks than ~16 per CU over all dates
    int gx = mcx_grid_for(n_paths, MCX_BLOCK * 32, (16 * h->n_cu + u->n_dates - 1) / u->n_dates);
    const size_t lds = sizeof(uint32_t) * (size_t)n_sel * ((size_t)1 << bits);
    hipLaunchKernelGGL(k5_hist<false>, dim3(gx, u->n_dates), dim3(MCX_BLOCK), lds, s, du, d_expo_ns, n_paths, ld, (int)n_sel, d_prefix,
                       (int)shift, (int)bits, (unsigned long long*)d_hist, (const unsigned long long*)nullptr);
    MCX_HIP(h, hipGetLastError());
    return 0;          // stream-ordered: the caller reads d_hist through the stream (a collective or a copy)
}

extern "C" int mcx_select_bracket(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                                  const double* d_lo, const double* d_hi, uint64_t* d_below, uint64_t* d_count, double* d_cand, int64_t cap,
                                  void* stream)
{
    if (!h || !u || !d_expo_ns || !d_lo || !d_hi || !d_below || !d_count || !d_cand) return -1;
    if (u->n_dates < 1 || cap < 1) MCX_FAIL(h, -2, "mcx_select_bracket: bad geometry (n_dates=%d cap=%lld)", u->n_dates, (long long)cap);
    if (ld < n_paths) MCX_FAIL(h, -2, "mcx_select_bracket: ld < n_paths");
    hipStream_t s = (hipStream_t)stream;
    MCX_HIP(h, hipMemsetAsync(d_below, 0, sizeof(uint64_t) * (size_t)u->n_dates, s));
    MCX_HIP(h, hipMemsetAsync(d_count, 0, sizeof(uint64_t) * (size_t)u->n_dates, s));
    if (n_paths <= 0) return 0;
    DevUnsec du; int32_t* tmp = nullptr;
    int rc = mcx_upload_unsec(h, u, &du, &tmp, s);
    if (rc) return rc;
    // 64 paths per thread (8 iterations of 8): many short blocks — with the 33 KiB stage of the first version 4 blocks fitted a CU and
    // a grid of 4.02 rounds of them ran for 5 (the pass took 0.67 ms at 3 TB/s for that reason alone)
    const int gx = mcx_grid_for(n_paths, MCX_BLOCK * 64, 4096);
    const bool vec = !u->collateralized && (ld % 2 == 0) && ((uintptr_t)d_expo_ns % 16 == 0);
    if (vec) hipLaunchKernelGGL(k5_bracket<true>, dim3(gx, u->n_dates), dim3(MCX_BLOCK), 0, s, du, d_expo_ns, n_paths, ld, d_lo, d_hi,
                                (unsigned long long*)d_below, (unsigned long long*)d_count, d_cand, cap);
    else hipLaunchKernelGGL(k5_bracket<false>, dim3(gx, u->n_dates), dim3(MCX_BLOCK), 0, s, du, d_expo_ns, n_paths, ld, d_lo, d_hi,
                            (unsigned long long*)d_below, (unsigned long long*)d_count, d_cand, cap);
    MCX_HIP(h, hipGetLastError());
    return 0;
}

extern "C" int mcx_select_hist_rows(mcx_handle* h, const double* d_rows, int32_t n_rows, int64_t ld, const uint64_t* d_row_n,
                                    int32_t n_sel, const uint64_t* d_prefix, int32_t shift, int32_t bits, uint64_t* d_hist, void* stream)
{
    if (!h || !d_rows || !d_row_n || !d_prefix || !d_hist) return -1;
    if (n_rows < 1 || ld < 1 || n_sel < 1 || n_sel > K5_MAX_SEL || bits < 1 || bits > 11 || shift < 0 || shift + bits > 64)
        MCX_FAIL(h, -2, "mcx_select_hist_rows: bad selection geometry (rows=%d n_sel=%d shift=%d bits=%d)", n_rows, n_sel, shift, bits);
    hipStream_t s = (hipStream_t)stream;
    const size_t nh = (size_t)n_rows * n_sel * ((size_t)1 << bits);
    MCX_HIP(h, hipMemsetAsync(d_hist, 0, nh * sizeof(uint64_t), s));
    DevUnsec du;
    memset(&du, 0, sizeof(du));
    du.n_dates = n_rows;
    const int gx = mcx_grid_for(ld, MCX_BLOCK * 32, (16 * h->n_cu + n_rows - 1) / n_rows);
    const size_t lds = sizeof(uint32_t) * (size_t)n_sel * ((size_t)1 << bits);
    hipLaunchKernelGGL(k5_hist<true>, dim3(gx, n_rows), dim3(MCX_BLOCK), lds, s, du, d_rows, ld, ld, (int)n_sel, d_prefix,
                       (int)shift, (int)bits, (unsigned long long*)d_hist, (const unsigned long long*)d_row_n);
    MCX_HIP(h, hipGetLastError());
    return 0;
}
