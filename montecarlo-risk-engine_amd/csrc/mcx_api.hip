// mcx_api.hip — handle, book (descriptor flattening + upload) and small shared host helpers of libmcx_hip.so.
#include <algorithm>
#include <atomic>
#include <thread>

#include "mcx_device.h"

extern "C" int mcx_abi_version(void) { return MCX_ABI_VERSION; }

extern "C" int mcx_create(mcx_handle** out, int device_id)
{
    if (!out) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return -2;
    if (hipSetDevice(device_id) != hipSuccess) return -3;
    mcx_handle* h = new mcx_handle();
    h->device = device_id;
    if (hipGetDeviceProperties(&h->prop, device_id) != hipSuccess) { delete h; return -4; }
    h->n_cu = h->prop.multiProcessorCount;
    h->ws_bytes = 8u << 20;
    h->pinned_bytes = 1u << 20;
    h->small_bytes = 4u << 20; h->small_cursor = 0;
    h->small_stream = nullptr; h->small_stream_valid = false;
    h->d_ws = nullptr; h->h_pinned = nullptr; h->d_small = nullptr; h->h_small = nullptr; h->d_acc = nullptr; h->d_pinned_alias = nullptr;
    for (int q = 0; q < 4; ++q) { h->scratch[q] = nullptr; h->scratch_bytes[q] = 0; }
    h->comm = nullptr; h->comm_ranks = 1; h->comm_rank = 0;
    const bool ok = hipMalloc(&h->d_ws, h->ws_bytes) == hipSuccess
                 && hipHostMalloc(&h->h_pinned, h->pinned_bytes, hipHostMallocMapped) == hipSuccess
                 && hipMalloc(&h->d_acc, h->pinned_bytes) == hipSuccess
                 && hipMalloc((void**)&h->d_small, h->small_bytes) == hipSuccess
                 && hipHostMalloc((void**)&h->h_small, h->small_bytes, hipHostMallocDefault) == hipSuccess;
    if (!ok) { mcx_destroy(h); return -5; }
    if (hipHostGetDevicePointer(&h->d_pinned_alias, h->h_pinned, 0) != hipSuccess) h->d_pinned_alias = nullptr;
    *out = h;
    return 0;
}

extern "C" void mcx_destroy(mcx_handle* h)
{
    if (!h) return;
    mcx_comm_destroy(h);
    hipFree(h->d_ws);
    hipHostFree(h->h_pinned);
    hipFree(h->d_small);
    hipHostFree(h->h_small);
    for (int q = 0; q < 4; ++q) hipFree(h->scratch[q]);
    hipFree(h->d_acc);
    delete h;
}

extern "C" const char* mcx_last_error(mcx_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int mcx_device_info(mcx_handle* h, int32_t* n_cu, int64_t* hbm_bytes, char* name, int32_t name_len)
{
    if (!h) return -1;
    if (n_cu) *n_cu = h->n_cu;
    if (hbm_bytes) *hbm_bytes = (int64_t)h->prop.totalGlobalMem;
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", h->prop.name, h->prop.gcnArchName);
    return 0;
}

// ---- book ------------------------------------------------------------------------------------------------------------
static DevAtom flat_atom(const mcx_atom& a)
{
    DevAtom o;
    o.t_idx = a.t_idx; o.col = a.col; o.a = a.a; o.d = a.d; o.b = a.b; o.c0 = a.c0; o.c1 = a.c1;
    return o;
}

extern "C" int mcx_book_create(mcx_handle* h, const mcx_book_desc* d, mcx_book** out)
{
    if (!h || !d || !out) return -1;
    if (d->n_basis < 1 || d->n_basis > MCX_MAX_BASIS) MCX_FAIL(h, -2, "mcx_book_create: n_basis %d out of range", d->n_basis);
    if (d->n_state < 1 || d->n_state > MCX_MAX_STATE) MCX_FAIL(h, -2, "mcx_book_create: n_state %d out of range", d->n_state);
    // validate every index the kernels will dereference (a wild index on the GPU can take the whole node down)
    if (d->n_dates < 1) MCX_FAIL(h, -2, "mcx_book_create: n_dates %d out of range", d->n_dates);
    for (int i = 0; i < d->n_atoms; ++i)
        if (d->atoms[i].col >= d->n_state || d->atoms[i].t_idx < 0 || d->atoms[i].t_idx >= d->n_dates)
            MCX_FAIL(h, -3, "mcx_book_create: atom %d out of range", i);
    for (int i = 0; i < d->n_terms; ++i)
        if (d->terms[i].atom < 0 || d->terms[i].atom >= d->n_atoms || d->terms[i].den >= d->n_atoms)
            MCX_FAIL(h, -3, "mcx_book_create: term %d references a bad atom", i);
    for (int i = 0; i < d->n_events; ++i) {
        const mcx_event& e = d->events[i];
        if (e.kind < MCX_EV_CASHFLOW || e.kind > MCX_EV_EXPO_BS) MCX_FAIL(h, -3, "mcx_book_create: event %d has a bad kind", i);
        if (e.t_idx < 0 || e.t_idx >= d->n_dates) MCX_FAIL(h, -3, "mcx_book_create: event %d date out of range", i);
        if (e.num_atom < 0 || e.num_atom >= d->n_atoms || e.x_atom >= d->n_atoms) MCX_FAIL(h, -3, "mcx_book_create: event %d atoms", i);
        if (e.term_begin < 0 || e.term_end < e.term_begin || e.term_end > d->n_terms) MCX_FAIL(h, -3, "mcx_book_create: event %d terms", i);
        if (e.coeff_off >= d->n_coeffs) MCX_FAIL(h, -3, "mcx_book_create: event %d coefficient offset", i);   // (block extent: per product below)
        if ((e.kind == MCX_EV_EXERCISE && e.coeff_off >= 0 && e.x_atom < 0) || (e.kind >= MCX_EV_EXPO_POLY && e.x_atom < 0))
            MCX_FAIL(h, -3, "mcx_book_create: event %d needs an explanatory atom", i);
        if (e.kind == MCX_EV_OPTION && e.aux[0] == 5.0 && (e.coeff_off < 0 || e.coeff_off + 2 > d->n_coeffs))
            MCX_FAIL(h, -3, "mcx_book_create: bridge barrier event %d needs two parameters at coeff_off", i);
        if (e.kind == MCX_EV_OPTION && e.aux[0] == 5.0 && d->coeffs &&
            !(d->coeffs[e.coeff_off + 1] >= 0.0 && d->coeffs[e.coeff_off + 1] < (double)d->n_products))
            MCX_FAIL(h, -3, "mcx_book_create: bridge barrier event %d: draw id outside [0, n_products)", i);
        if (e.kind == MCX_EV_OPTION && (e.aux[0] == 4.0 || e.aux[0] == 5.0) && (e.x_atom < 0 || ((int)e.aux[3] & 7) < 1 || ((int)e.aux[3] & 7) > 4 || ((int)e.aux[3] >> 3) > 4))
            MCX_FAIL(h, -3, "mcx_book_create: barrier event %d needs the maturity spot in x_atom and barrier types in 1..4", i);
        if (e.kind == MCX_EV_OPTION && e.aux[0] == 3.0 && !(e.aux[2] > 0.0)) MCX_FAIL(h, -3, "mcx_book_create: binary event %d needs eps > 0", i);
        if (e.kind >= MCX_EV_EXPO_POLY && (e.expo_row < 0 || e.expo_row >= d->n_expo_rows)) MCX_FAIL(h, -3, "mcx_book_create: event %d row", i);
    }
    for (int p = 0; p < d->n_products; ++p) {
        const mcx_product& pr = d->products[p];
        if (pr.ev_begin < 0 || pr.ev_end < pr.ev_begin || pr.ev_end > d->n_events || pr.cf_begin < 0 || pr.cf_end < pr.cf_begin ||
            pr.cf_end > d->n_events || pr.netting_set < 0 || pr.netting_set >= d->n_netting_sets || pr.n_states < 1 ||
            pr.n_states > MCX_MAX_STATES || pr.init_state < 0 || pr.init_state >= pr.n_states)
            MCX_FAIL(h, -3, "mcx_book_create: product %d out of range", p);
        for (int q = pr.ev_begin; q < pr.ev_end; ++q) {
            const mcx_event& e = d->events[q];
            const bool bridge = e.kind == MCX_EV_OPTION && e.aux[0] == 5.0;      // coeff_off = two parked parameters, checked above
            if (!bridge && e.coeff_off >= 0 && e.coeff_off + pr.n_states * d->n_basis > d->n_coeffs)
                MCX_FAIL(h, -3, "mcx_book_create: product %d event %d coefficients out of range", p, q);
        }
        for (int q = pr.cf_begin; q < pr.cf_end; ++q) {
            const mcx_event& e = d->events[q];
            if (e.kind > MCX_EV_EXERCISE) MCX_FAIL(h, -3, "mcx_book_create: product %d cf event %d is not a cash event", p, q);
            const bool bridge = e.kind == MCX_EV_OPTION && e.aux[0] == 5.0;
            if (!bridge && e.coeff_off >= 0 && e.coeff_off + pr.n_states * d->n_basis > d->n_coeffs)
                MCX_FAIL(h, -3, "mcx_book_create: product %d cf event %d coefficients out of range", p, q);
        }
    }
    MCX_HIP(h, hipSetDevice(h->device));
    mcx_book* b = new mcx_book();
    b->n_atoms = d->n_atoms; b->n_terms = d->n_terms; b->n_events = d->n_events; b->n_products = d->n_products;
    b->n_netting_sets = d->n_netting_sets; b->n_expo_rows = d->n_expo_rows; b->n_basis = d->n_basis; b->n_coeffs = d->n_coeffs;
    b->want_cfs = d->want_cfs; b->want_expo = d->want_expo; b->n_state = d->n_state;
    b->h_atoms.assign(d->atoms, d->atoms + d->n_atoms);

    std::vector<DevAtom> atoms(d->n_atoms > 0 ? d->n_atoms : 1);
    for (int i = 0; i < d->n_atoms; ++i) atoms[i] = flat_atom(d->atoms[i]);
    std::vector<DevTerm> terms(d->n_terms > 0 ? d->n_terms : 1);
    for (int i = 0; i < d->n_terms; ++i) {
        terms[i].w = d->terms[i].w;
        terms[i].atom = flat_atom(d->atoms[d->terms[i].atom]);
        terms[i].den = d->terms[i].den;
        terms[i].pad = 0;
    }
    // the flattened events are built in place in the book's host image (4 x 10^5 events of a 5,000-product book are 64 MB: a
    // second vector and its copy were a third of this call), by a few threads for big books: each zeroes and fills its own chunk
    DevEventVec& events = b->h_events;
    events.resize(d->n_events > 0 ? d->n_events : 1);
    DevAtom none; memset(&none, 0, sizeof(none)); none.col = -1;
    // (the per-event side tables and the event-family flags are filled in the same pass: every extra walk over the 32 MB of
    //  descriptors of a 5,000-product book costs ~3 ms)
    b->h_event_t_idx.resize(d->n_events);
    b->h_event_num_atom.resize(d->n_events); b->h_event_x_atom.resize(d->n_events);
    std::atomic<int> fam{0};                       // bit 0 barrier, 1 exotic, 2 exercise, 3 closed-form exposure
    auto convert = [&](int i0, int i1) {
        if (i1 > i0) memset((void*)&events[i0], 0, sizeof(DevEvent) * (size_t)(i1 - i0));
        int mine = 0;
        for (int i = i0; i < i1; ++i) {
            const mcx_event& e = d->events[i];
            b->h_event_t_idx[i] = e.t_idx; b->h_event_num_atom[i] = e.num_atom; b->h_event_x_atom[i] = e.x_atom;
            if (e.kind == MCX_EV_OPTION && (e.aux[0] == 4.0 || e.aux[0] == 5.0)) mine |= 1;
            if (e.kind == MCX_EV_OPTION && e.aux[0] != 0.0) mine |= 2;
            if (e.kind == MCX_EV_EXERCISE) mine |= 4;
            if (e.kind == MCX_EV_EXPO_BS) mine |= 8;
            DevEvent& o = events[i];
            o.kind = e.kind; o.term_begin = e.term_begin; o.term_end = e.term_end; o.coeff_off = e.coeff_off; o.row = e.expo_row;
            o.strike = e.strike; o.sign = e.sign;
            for (int q = 0; q < 4; ++q) o.aux[q] = e.aux[q];
            o.num = flat_atom(d->atoms[e.num_atom]);
            o.x = e.x_atom >= 0 ? flat_atom(d->atoms[e.x_atom]) : none;
        }
        fam.fetch_or(mine);
    };
    if (d->n_events <= 0) memset((void*)&events[0], 0, sizeof(DevEvent));
    const int n_thr = d->n_events >= (1 << 16) ? 8 : 1;
    if (n_thr == 1) convert(0, d->n_events);
    else {
        std::vector<std::thread> pool;
        const int per = (d->n_events + n_thr - 1) / n_thr;
        for (int t = 0; t < n_thr; ++t) pool.emplace_back(convert, std::min(t * per, d->n_events), std::min((t + 1) * per, d->n_events));
        for (auto& th : pool) th.join();
    }
    // exposure rows: the first writer of a (netting set, row) stores, later ones accumulate; rows nobody writes need a memset
    b->ns_has_writer.assign((size_t)d->n_netting_sets * (d->n_expo_rows > 0 ? d->n_expo_rows : 1), 0);
    b->h_products.resize(d->n_products);
    for (int p = 0; p < d->n_products; ++p) {
        const mcx_product& pr = d->products[p];
        DevProduct& o = b->h_products[p];
        o.ev_begin = pr.ev_begin; o.ev_end = pr.ev_end; o.cf_begin = pr.cf_begin; o.cf_end = pr.cf_end;
        o.netting_set = pr.netting_set; o.init_state = pr.init_state; o.n_states = pr.n_states; o.flags = pr.flags;
        bool have_num = false;
        for (int q = pr.ev_begin; q < pr.ev_end; ++q) {
            DevEvent& e = events[q];
            e.netting_set = pr.netting_set;
            if (e.kind >= MCX_EV_EXPO_POLY) {
                uint8_t& seen = b->ns_has_writer[(size_t)pr.netting_set * d->n_expo_rows + e.row];
                e.flags = seen ? 1 : 0;
                seen = 1;
            }
            // bit 1: the numeraire atom equals the one the previous event of this product evaluated (the cashflow and the exposure
            // of one date share it): the multi-path book kernel keeps the reciprocal in registers
            const bool evaluates = e.kind != MCX_EV_EXPO_BS || e.aux[2] > 0.0;
            if (have_num && evaluates && memcmp(&e.num, &events[q - 1].num, sizeof(DevAtom)) == 0) e.flags |= 2;
            have_num = evaluates || (have_num && (e.flags & 2));
        }
    }
    b->h_terms = terms; b->h_terms.resize(d->n_terms);
    // atom ids behind the flattened copies (the tangent kernels index per-atom derivative rows, kt_book.hip): events above, terms here
    b->h_term_atom.resize(d->n_terms);
    for (int i = 0; i < d->n_terms; ++i) b->h_term_atom[i] = d->terms[i].atom;
    const int fam_all = fam.load();
    b->has_barrier = (fam_all & 1) != 0; b->has_exotic = (fam_all & 2) != 0; b->has_exercise = (fam_all & 4) != 0; b->has_bs_expo = (fam_all & 8) != 0;
    b->has_den = false;
    for (int p = 0; p < d->n_products; ++p) if (d->products[p].n_states != 1) b->has_exercise = true;
    for (int i = 0; i < d->n_terms; ++i) if (d->terms[i].den >= 0) b->has_den = true;
    b->expo_needs_memset = false;
    if (d->want_expo)
        for (size_t q = 0; q < (size_t)d->n_netting_sets * d->n_expo_rows; ++q)
            if (!b->ns_has_writer[q]) b->expo_needs_memset = true;

    b->d_atoms = nullptr; b->d_terms = nullptr; b->d_events = nullptr; b->d_products = nullptr; b->d_coeffs = nullptr;
    b->d_bridge = nullptr; b->d_bridge_inject = nullptr;
    b->ex_mode = 0; b->d_ex_bits = nullptr; b->ex_ld = 0;
    b->d_vpoly = nullptr; b->d_vcoef = nullptr;
    b->h_event_vpoly.assign((size_t)d->n_events, -1);
    auto upload = [&](void** dst, const void* src, size_t bytes) -> bool {        // false: the error is in h->err; the caller frees
        hipError_t e = hipMalloc(dst, bytes ? bytes : 8);
        if (e == hipSuccess && src && bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { h->err = std::string("mcx_book_create: ") + hipGetErrorString(e); return false; }
        return true;
    };
    bool ok = upload((void**)&b->d_atoms, atoms.data(), sizeof(DevAtom) * atoms.size())
           && upload((void**)&b->d_terms, terms.data(), sizeof(DevTerm) * terms.size())
           && upload((void**)&b->d_events, events.data(), sizeof(DevEvent) * events.size())
           && upload((void**)&b->d_products, b->h_products.data(), sizeof(DevProduct) * b->h_products.size())
           && upload((void**)&b->d_coeffs, d->n_coeffs > 0 ? d->coeffs : nullptr, sizeof(double) * (size_t)(d->n_coeffs > 0 ? d->n_coeffs : 1))
           && upload((void**)&b->d_bridge, nullptr, sizeof(DevBridge))
           && upload((void**)&b->d_bridge_inject, nullptr, sizeof(double*) * (size_t)(d->n_products > 0 ? d->n_products : 1));
    if (ok && hipMemset(b->d_bridge, 0, sizeof(DevBridge)) != hipSuccess) { h->err = "mcx_book_create: hipMemset failed"; ok = false; }
    if (!ok) { mcx_book_destroy(b); return -100; }
    b->h_events.resize(d->n_events);
    *out = b;
    return 0;
}

extern "C" void mcx_book_destroy(mcx_book* b)
{
    if (!b) return;
    hipFree(b->d_atoms); hipFree(b->d_terms); hipFree(b->d_events); hipFree(b->d_products); hipFree(b->d_coeffs);
    hipFree(b->d_bridge); hipFree(b->d_bridge_inject);
    hipFree(b->d_vpoly); hipFree(b->d_vcoef);
    delete b;
}

extern "C" int mcx_book_set_coeffs(mcx_handle* h, mcx_book* b, int64_t offset, int64_t count, const double* h_coeffs, void* stream)
{
    if (!h || !b || !h_coeffs) return -1;
    if (offset < 0 || count < 0 || offset + count > b->n_coeffs) MCX_FAIL(h, -2, "mcx_book_set_coeffs: range out of bounds");
    if (count == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // stage through pinned memory so the copy is stream-ordered with the kernels that read the coefficients
    if ((size_t)count * sizeof(double) <= h->pinned_bytes) {
        MCX_HIP(h, hipStreamSynchronize(s));
        memcpy(h->h_pinned, h_coeffs, (size_t)count * sizeof(double));
        MCX_HIP(h, hipMemcpyAsync(b->d_coeffs + offset, h->h_pinned, (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
        MCX_HIP(h, hipStreamSynchronize(s));
    } else {
        MCX_HIP(h, hipStreamSynchronize(s));
        MCX_HIP(h, hipMemcpy(b->d_coeffs + offset, h_coeffs, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int mcx_book_set_bridge_rng(mcx_handle* h, mcx_book* b, uint64_t seed, uint64_t path_offset, const double* const* h_inject,
                                       int64_t ld, void* stream)
{
    if (!h || !b) return -1;
    hipStream_t s = (hipStream_t)stream;
    MCX_HIP(h, hipStreamSynchronize(s));
    DevBridge br;
    br.seed = seed; br.path_offset = path_offset; br.inject = nullptr; br.ld = ld;
    if (h_inject) {
        MCX_HIP(h, hipMemcpy(b->d_bridge_inject, h_inject, sizeof(double*) * (size_t)b->n_products, hipMemcpyHostToDevice));
        br.inject = b->d_bridge_inject;
    }
    MCX_HIP(h, hipMemcpy(b->d_bridge, &br, sizeof(br), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int mcx_book_set_exercise_replay(mcx_handle* h, mcx_book* b, int32_t mode, uint8_t* d_bits, int64_t n_rows, int64_t ld)
{
    if (!h || !b) return -1;
    if (mode < 0 || mode > 2 || (mode != 0 && (!d_bits || ld <= 0))) MCX_FAIL(h, -2, "mcx_book_set_exercise_replay: bad arguments");
    if (mode != 0 && n_rows < b->n_events)
        MCX_FAIL(h, -3, "mcx_book_set_exercise_replay: %lld rows for a book of %d events", (long long)n_rows, b->n_events);
    b->ex_mode = mode; b->d_ex_bits = mode ? d_bits : nullptr; b->ex_ld = mode ? ld : 0;
    return 0;
}

// ---- shared helpers ---------------------------------------------------------------------------------------------------
void* mcx_scratch(mcx_handle* h, int slot, size_t bytes)
{
    if (bytes <= h->scratch_bytes[slot] && h->scratch[slot]) return h->scratch[slot];
    hipFree(h->scratch[slot]);               // (synchronises: only when a call needs more than any call before it)
    h->scratch[slot] = nullptr; h->scratch_bytes[slot] = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes;
    if (hipMalloc(&h->scratch[slot], want) != hipSuccess) { h->err = "hipMalloc of a scratch buffer failed"; return nullptr; }
    h->scratch_bytes[slot] = want;
    return h->scratch[slot];
}

void* mcx_stage_small(mcx_handle* h, const void* src, size_t bytes, hipStream_t s)
{
    const size_t need = (bytes + 255) & ~(size_t)255;
    const size_t half = h->small_bytes / 2;
    if (need > half) { h->err = "descriptor too large for the staging ring"; return nullptr; }
    // one stream at a time: work staged for another stream is drained before this stream's data may displace it
    if (h->small_stream_valid && h->small_stream != s && hipStreamSynchronize(h->small_stream) != hipSuccess) {
        h->err = "hipStreamSynchronize failed"; return nullptr;
    }
    h->small_stream = s; h->small_stream_valid = true;
    const int cur = h->small_cursor >= half ? 1 : 0;
    if (h->small_cursor + need > (size_t)(cur + 1) * half) {
        // this half is full: drain the stream (every kernel that reads the OTHER half has then finished) and move there.  What the
        // current call staged in this half for kernels it has not launched yet stays intact: it is displaced only after another
        // half-ring (2 MiB) of descriptors
        if (hipStreamSynchronize(s) != hipSuccess) { h->err = "hipStreamSynchronize failed"; return nullptr; }
        h->small_cursor = (size_t)(cur ^ 1) * half;
    }
    unsigned char* hp = h->h_small + h->small_cursor;
    unsigned char* dp = h->d_small + h->small_cursor;
    memcpy(hp, src, bytes);
    if (hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, s) != hipSuccess) { h->err = "hipMemcpyAsync failed"; return nullptr; }
    h->small_cursor += need;
    return dp;
}

const void* mcx_upload_call_data(mcx_handle* h, const void* src, size_t bytes, void* fallback, hipStream_t s)
{
    if (bytes <= h->small_bytes / 4) return mcx_stage_small(h, src, bytes, s);
    // too large for the ring: a copy from pageable memory into the caller's device buffer, completed before `src` may go away
    if (hipMemcpyAsync(fallback, src, bytes, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        h->err = "host-to-device copy of a job table failed";
        return nullptr;
    }
    return fallback;
}

int mcx_upload_unsec(mcx_handle* h, const mcx_unsecured_desc* u, DevUnsec* out, int32_t** d_tmp, hipStream_t s)
{
    if (u->n_dates < 1 || u->n_dates > MCX_MAX_METRIC_DATES) MCX_FAIL(h, -2, "unsecured desc: n_dates %d out of range", u->n_dates);
    if (u->n_rows < 1 || !u->row) MCX_FAIL(h, -2, "unsecured desc: n_rows %d / row table missing", u->n_rows);
    for (int m = 0; m < u->n_dates; ++m) {                          // the kernels index the exposure block with these
        if (u->row[m] < 0 || u->row[m] >= u->n_rows) MCX_FAIL(h, -2, "unsecured desc: row[%d] = %d outside the %d exposure rows", m, u->row[m], u->n_rows);
        if (u->delayed && (u->delayed[m] < -1 || u->delayed[m] >= u->n_rows))
            MCX_FAIL(h, -2, "unsecured desc: delayed[%d] = %d outside the %d exposure rows", m, u->delayed[m], u->n_rows);
    }
    *d_tmp = nullptr;                                               // (nothing to free: the tables live in the staging ring)
    const int32_t* d_row = (const int32_t*)mcx_stage_small(h, u->row, sizeof(int32_t) * u->n_dates, s);
    if (!d_row) return -100;
    const int32_t* d_del = nullptr;
    if (u->delayed) {
        d_del = (const int32_t*)mcx_stage_small(h, u->delayed, sizeof(int32_t) * u->n_dates, s);
        if (!d_del) return -100;
    }
    out->n_dates = u->n_dates; out->collateralized = u->collateralized; out->threshold = u->threshold;
    out->row = d_row; out->delayed = d_del;
    return 0;
}

namespace {
// sums per-block partials: partials [n_blocks][n_records][2] -> out [n_records] as mcx_acc
__global__ void k_finish_acc(const double* __restrict__ partials, int n_records, int n_blocks, double n_paths,
                             const double* __restrict__ shifts, mcx_acc* __restrict__ out)
{
    const int r = blockIdx.x;
    __shared__ double lds[8];
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
        s1 += partials[((int64_t)b * n_records + r) * 2 + 0];
        s2 += partials[((int64_t)b * n_records + r) * 2 + 1];
    }
    s1 = block_sum(s1, lds);
    s2 = block_sum(s2, lds + 4);
    if (threadIdx.x == 0) {
        out[r].n = n_paths; out[r].shift = shifts[r]; out[r].s1 = s1; out[r].s2 = s2;
    }
}
}  // namespace

int mcx_finish_acc(mcx_handle* h, const double* d_partials, int n_records, int n_blocks, double n_paths,
                   const double* d_shifts, mcx_acc* h_out, hipStream_t s)
{
    if ((size_t)n_records * sizeof(mcx_acc) > h->pinned_bytes) MCX_FAIL(h, -2, "too many accumulator records");
    mcx_acc* d_out = (mcx_acc*)h->d_acc;
    hipLaunchKernelGGL(k_finish_acc, dim3(n_records), dim3(MCX_BLOCK), 0, s, d_partials, n_records, n_blocks, n_paths, d_shifts, d_out);
    MCX_HIP(h, hipGetLastError());
    MCX_HIP(h, hipMemcpyAsync(h->h_pinned, d_out, sizeof(mcx_acc) * (size_t)n_records, hipMemcpyDeviceToHost, s));
    MCX_HIP(h, hipStreamSynchronize(s));
    memcpy(h_out, h->h_pinned, sizeof(mcx_acc) * (size_t)n_records);
    return 0;
}
