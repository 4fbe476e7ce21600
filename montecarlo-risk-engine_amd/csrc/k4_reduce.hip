// k4_reduce.hip — K4: netting-set post-processing + Monte-Carlo reductions (PV / CE / EPE / ENE / CVA).
//
// Replaces NettingSet.compute_unsecured_exposure_profiles (products/netting_set.py:156-184, fused as a prologue) and the
// `values.mean()`, `values.std(unbiased=True)` pairs of metrics/metric.py:26-35 behind PVMetric, CEMetric, EPEMetric,
// ENEMetric and CVAMetric (metrics/cva_metric.py:62-100).
//
// HBM-bound streaming kernels: rows of the [date][path] exposure matrix are read once with 512-byte coalesced wave
// loads; partial sums stay in VGPRs, then wave64 shuffle (DPP) -> LDS across the 4 waves -> one partial per block ->
// a deterministic second stage (no float atomics: results are bitwise reproducible).  Every record is the shifted pair
// (sum (x-c), sum (x-c)^2) with c = value of the first local path, so the variance does not cancel catastrophically and a
// deterministic profile gives an exact zero Monte-Carlo error.
#include "mcx_internal.h"

namespace {

__global__ __launch_bounds__(MCX_BLOCK) void k4_vector(const double* __restrict__ x, int64_t n, double* __restrict__ partials,
                                                       double* __restrict__ shifts)
{
    const double c = x[0];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const double d = x[i] - c;
        s1 += d; s2 = fma(d, d, s2);
    }
    __shared__ double lds[8];
    s1 = block_sum(s1, lds);
    s2 = block_sum(s2, lds + 4);
    if (threadIdx.x == 0) {
        partials[(int64_t)blockIdx.x * 2 + 0] = s1;
        partials[(int64_t)blockIdx.x * 2 + 1] = s2;
        if (blockIdx.x == 0) shifts[0] = c;
    }
}

// grid (chunks, n_dates): block (b, m) reduces date m over its path chunk; records 2m (positive part) and 2m+1 (negative part)
__global__ __launch_bounds__(MCX_BLOCK) void k4_profiles(const DevUnsec u, const double* __restrict__ expo, int64_t n, int64_t ld,
                                                         double* __restrict__ partials, double* __restrict__ shifts)
{
    const int m = blockIdx.y;
    const double x0 = dev_unsec(u, expo, ld, m, 0);
    const double cp = fmax(x0, 0.0), cn = fmin(x0, 0.0);
    double p1 = 0.0, p2 = 0.0, n1 = 0.0, n2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * MCX_BLOCK) {
        const double x = dev_unsec(u, expo, ld, m, i);
        const double dp = fmax(x, 0.0) - cp;         // relu(x)          epe_metric.py:14
        const double dn = fmin(x, 0.0) - cn;         // -relu(-x)        ene_metric.py:14
        p1 += dp; p2 = fma(dp, dp, p2);
        n1 += dn; n2 = fma(dn, dn, n2);
    }
    __shared__ double lds[16];
    p1 = block_sum(p1, lds); p2 = block_sum(p2, lds + 4); n1 = block_sum(n1, lds + 8); n2 = block_sum(n2, lds + 12);
    if (threadIdx.x == 0) {
        const int R = 2 * gridDim.y;
        double* dst = partials + (int64_t)blockIdx.x * R * 2;
        dst[(2 * m) * 2 + 0] = p1; dst[(2 * m) * 2 + 1] = p2;
        dst[(2 * m + 1) * 2 + 0] = n1; dst[(2 * m + 1) * 2 + 1] = n2;
        if (blockIdx.x == 0) { shifts[2 * m] = cp; shifts[2 * m + 1] = cn; }
    }
}

// per-path CVA integrand (cva_metric.py:88-99): one lane = one path, loop over the exposure dates
__global__ __launch_bounds__(MCX_BLOCK) void k4_cva_paths(const DevUnsec u, const DevAtom* __restrict__ atoms,
                                                          const int32_t* __restrict__ surv, const int32_t* __restrict__ cond,
                                                          double lgd, const double* __restrict__ expo, const double* __restrict__ paths,
                                                          int64_t D, int64_t n, int64_t ld_expo, int64_t ld_paths, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= n) return;
    double acc = 0.0;
    for (int m = 0; m < u.n_dates - 1; ++m) {
        const double e = fmax(dev_unsec(u, expo, ld_expo, m, i), 0.0);
        const double sp = dev_atom(ldk_struct(&atoms[ldk(surv + m)]), paths, D, ld_paths, i);
        const double cs = dev_atom(ldk_struct(&atoms[ldk(cond + m)]), paths, D, ld_paths, i);
        acc = fma(e, sp * (1.0 - cs), acc);
    }
    out[i] = acc * lgd;
}

// the same with PPL paths per lane and table exponentials (large path counts: independent loads in flight, the atom records
// and the date loop paid once per PPL x 64 paths)
template <int PPL>
__global__ __launch_bounds__(MCX_BLOCK) void k4_cva_paths_v(const DevUnsec u, const DevAtom* __restrict__ atoms,
                                                            const int32_t* __restrict__ surv, const int32_t* __restrict__ cond,
                                                            double lgd, const double* __restrict__ expo, const double* __restrict__ paths,
                                                            int64_t D, int64_t n, int64_t ld_expo, int64_t ld_paths, double* __restrict__ out)
{
    __shared__ double etab[MCX_EXP_LDS_DOUBLES];
    mcx_exp_tab_load(etab);
    __syncthreads();
    int64_t i[PPL];
    bool live[PPL];
    double acc[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int64_t i_raw = ((int64_t)blockIdx.x * PPL + q) * MCX_BLOCK + threadIdx.x;
        live[q] = i_raw < n;
        i[q] = live[q] ? i_raw : n - 1;
        acc[q] = 0.0;
    }
    if (!live[0]) return;
    const mcx_expq_coef ec = mcx_expq_load();
    for (int m = 0; m < u.n_dates - 1; ++m) {
        double e[PPL], sp[PPL], cs[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) e[q] = fmax(dev_unsec(u, expo, ld_expo, m, i[q]), 0.0);
        dev_atoms<PPL>(ldk_struct(&atoms[ldk(surv + m)]), paths, D, ld_paths, i, sp, etab, ec);
        dev_atoms<PPL>(ldk_struct(&atoms[ldk(cond + m)]), paths, D, ld_paths, i, cs, etab, ec);
#pragma unroll
        for (int q = 0; q < PPL; ++q) acc[q] = fma(e[q], sp[q] * (1.0 - cs[q]), acc[q]);
    }
#pragma unroll
    for (int q = 0; q < PPL; ++q) if (live[q]) out[i[q]] = acc[q] * lgd;
}

__global__ __launch_bounds__(MCX_BLOCK) void k4_unsecured(const DevUnsec u, const double* __restrict__ expo, int64_t n, int64_t ld,
                                                          double* __restrict__ out, int64_t ld_out)
{
    const int64_t i = (int64_t)blockIdx.x * MCX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int m = blockIdx.y;
    out[(int64_t)m * ld_out + i] = dev_unsec(u, expo, ld, m, i);
}

int reduce_vector_dev(mcx_handle* h, const double* d_x, int64_t n, mcx_acc* h_out, hipStream_t s)
{
    const int grid = mcx_grid_for(n, MCX_BLOCK, MCX_MAX_PARTIAL_BLOCKS);
    double* part = h->d_ws;
    double* shifts = h->d_ws + (size_t)grid * 2;
    hipLaunchKernelGGL(k4_vector, dim3(grid), dim3(MCX_BLOCK), 0, s, d_x, n, part, shifts);
    MCX_HIP(h, hipGetLastError());
    return mcx_finish_acc(h, part, 1, grid, (double)n, shifts, h_out, s);
}

}  // namespace

extern "C" int mcx_reduce_vector(mcx_handle* h, const double* d_x, int64_t n_paths, mcx_acc* h_out, void* stream)
{
    if (!h || !d_x || !h_out) return -1;
    if (n_paths <= 0) { memset(h_out, 0, sizeof(mcx_acc)); return 0; }
    return reduce_vector_dev(h, d_x, n_paths, h_out, (hipStream_t)stream);
}

extern "C" int mcx_reduce_profiles(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                                   mcx_acc* h_out, void* stream)
{
    if (!h || !u || !d_expo_ns || !h_out) return -1;
    if (n_paths <= 0) { memset(h_out, 0, sizeof(mcx_acc) * 2 * (size_t)u->n_dates); return 0; }
    if (ld < n_paths) MCX_FAIL(h, -2, "mcx_reduce_profiles: ld < n_paths");
    hipStream_t s = (hipStream_t)stream;
    DevUnsec du; int32_t* tmp = nullptr;
    int rc = mcx_upload_unsec(h, u, &du, &tmp, s);
    if (rc) return rc;
    const int R = 2 * u->n_dates;
    int gx = mcx_grid_for(n_paths, MCX_BLOCK * 4, 8 * h->n_cu / (u->n_dates > 8 ? 8 : u->n_dates) + 1);
    while ((size_t)gx * R * 2 * sizeof(double) + (size_t)R * sizeof(double) > h->ws_bytes && gx > 1) gx /= 2;
    double* part = h->d_ws;
    double* shifts = h->d_ws + (size_t)gx * R * 2;
    hipLaunchKernelGGL(k4_profiles, dim3(gx, u->n_dates), dim3(MCX_BLOCK), 0, s, du, d_expo_ns, n_paths, ld, part, shifts);
    MCX_HIP(h, hipGetLastError());
    return mcx_finish_acc(h, part, R, gx, (double)n_paths, shifts, h_out, s);
}

extern "C" int mcx_reduce_cva(mcx_handle* h, const mcx_book* b, const mcx_unsecured_desc* u, const int32_t* h_surv_atoms,
                              const int32_t* h_cond_atoms, double recovery, const double* d_expo_ns, const double* d_paths,
                              int64_t n_paths, int64_t ld_expo, int64_t ld_paths, mcx_acc* h_out, void* stream)
{
    if (!h || !b || !u || !d_expo_ns || !d_paths || !h_out) return -1;
    if (n_paths <= 0) { memset(h_out, 0, sizeof(mcx_acc)); return 0; }
    if (ld_expo < n_paths || ld_paths < n_paths) MCX_FAIL(h, -2, "mcx_reduce_cva: leading dimension < n_paths");
    const int nd = u->n_dates - 1;
    for (int m = 0; m < nd; ++m)
        if (h_surv_atoms[m] < 0 || h_surv_atoms[m] >= b->n_atoms || h_cond_atoms[m] < 0 || h_cond_atoms[m] >= b->n_atoms)
            MCX_FAIL(h, -2, "mcx_reduce_cva: atom id out of range");
    hipStream_t s = (hipStream_t)stream;
    DevUnsec du; int32_t* tmp = nullptr;
    int rc = mcx_upload_unsec(h, u, &du, &tmp, s);
    if (rc) return rc;
    for (int m = 0; m < u->n_dates; ++m)          // the rows index the netting set's exposure block of the book
        if (u->row[m] < 0 || u->row[m] >= b->n_expo_rows || (u->delayed && u->delayed[m] >= b->n_expo_rows))
            MCX_FAIL(h, -2, "mcx_reduce_cva: exposure row of metric date %d out of range", m);
    double* d_v = mcx_path_scratch(h, (size_t)n_paths);
    if (!d_v) return -100;
    const int32_t* d_ids = nullptr;
    if (nd > 0) {
        std::vector<int32_t> ids(h_surv_atoms, h_surv_atoms + nd);
        ids.insert(ids.end(), h_cond_atoms, h_cond_atoms + nd);
        d_ids = (const int32_t*)mcx_stage_small(h, ids.data(), sizeof(int32_t) * ids.size(), s);
        if (!d_ids) return -100;
    }
    const int grid = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    if (grid >= 16 * h->n_cu)
        hipLaunchKernelGGL((k4_cva_paths_v<2>), dim3((grid + 1) / 2), dim3(MCX_BLOCK), 0, s, du, b->d_atoms, d_ids, d_ids + nd, 1.0 - recovery,
                           d_expo_ns, d_paths, (int64_t)b->n_state, n_paths, ld_expo, ld_paths, d_v);
    else
        hipLaunchKernelGGL(k4_cva_paths, dim3(grid), dim3(MCX_BLOCK), 0, s, du, b->d_atoms, d_ids, d_ids + nd, 1.0 - recovery, d_expo_ns,
                           d_paths, (int64_t)b->n_state, n_paths, ld_expo, ld_paths, d_v);
    MCX_HIP(h, hipGetLastError());
    return reduce_vector_dev(h, d_v, n_paths, h_out, s);
}

extern "C" int mcx_unsecured(mcx_handle* h, const mcx_unsecured_desc* u, const double* d_expo_ns, int64_t n_paths, int64_t ld,
                             double* d_out, int64_t ld_out, void* stream)
{
    if (!h || !u || !d_expo_ns || !d_out) return -1;
    if (n_paths <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    DevUnsec du; int32_t* tmp = nullptr;
    int rc = mcx_upload_unsec(h, u, &du, &tmp, s);
    if (rc) return rc;
    const int gx = (int)((n_paths + MCX_BLOCK - 1) / MCX_BLOCK);
    hipLaunchKernelGGL(k4_unsecured, dim3(gx, u->n_dates), dim3(MCX_BLOCK), 0, s, du, d_expo_ns, n_paths, ld, d_out, ld_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipStreamSynchronize(s));
    return 0;
}
