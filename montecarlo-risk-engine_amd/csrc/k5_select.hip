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

__global__ __launch_bounds__(MCX_BLOCK) void k5_hist(const DevUnsec u, const double* __restrict__ expo, int64_t n, int64_t ld,
                                                     int n_sel, const uint64_t* __restrict__ prefix, int shift, int bits,
                                                     unsigned long long* __restrict__ hist)
{
    extern __shared__ uint32_t lh[];                  // [n_sel][1<<bits]
    const int m = blockIdx.y;
    const int nb = 1 << bits;
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
            kk[w] = dev_key(dev_unsec(u, expo, ld, m, in[w] ? i : n - 1));
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
    hipLaunchKernelGGL(k5_hist, dim3(gx, u->n_dates), dim3(MCX_BLOCK), lds, s, du, d_expo_ns, n_paths, ld, (int)n_sel, d_prefix,
                       (int)shift, (int)bits, (unsigned long long*)d_hist);
    MCX_HIP(h, hipGetLastError());
    return 0;          // stream-ordered: the caller reads d_hist through the stream (a collective or a copy)
}
