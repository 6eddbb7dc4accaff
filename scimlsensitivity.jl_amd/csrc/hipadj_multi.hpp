// hipadj_multi.hpp — ONE handle over several devices (hipadj_config.ndevices / device_ids, ABI 108).
//
// The reference shards an ensemble over workers (EnsembleDistributed, docs/src/tutorials/data_parallel.md:77-136, test/Core4/distributed.jl:15-41): every worker solves its
// own trajectories, the outer loss sums them.  One process per GPU plus an RCCL all-reduce (hipadj_comm_*) is that pattern; THIS file is the other half of SURVEY.md 8(b)'s
// `device_ids[G]`: a host that calls `solve` ONCE — the case "drops into DifferentialEquations.jl unchanged" — reaches every GPU of its node through one handle.
//
// A multi handle owns G ordinary handles ("shards"), shard g holding the contiguous trajectory range [g N / G, (g + 1) N / G) on device_ids[g] with its own
// stream, workspaces and kernels; nothing about a shard's kernels knows that it is a shard.  Trajectories never interact, so the only cross-shard arithmetic is the sum of
// dL/dp over the shards when p is shared — taken in shard order (fixed), on the host for the host-pointer calls and by one small kernel on the first device for the
// device-pointer calls.  du0 / out stay sliced.
//   host-pointer calls   every shard's uploads and kernels are enqueued on ITS stream before any stream is waited for: the devices run concurrently; the downloads follow per shard
//   device-pointer calls buffers live on device_ids[0] ("the primary"): shards on that device read and write the caller's slices in place, shards elsewhere go through their
//                        staging buffers with hipMemcpyPeerAsync; the primary stream (hipadj_set_stream) is ordered before and after the shard streams by events
// The same ordinal may repeat in device_ids: "virtual shards" on one device — how a 1-GPU box tests this path and how tests/c/julia_seam.c drives it.
#pragma once
#include "hipadj_host.hpp"

static int forward_host_enqueue(hipadj_handle* h, const double* u0, const double* p, double* out);
static int forward_host_finish(hipadj_handle* h, double* out);
static int adjoint_host_run(hipadj_handle* h, const double* dLdu);
static int adjoint_host_download(hipadj_handle* h, double* du0, double* dp);

static __global__ void k_sum_rows(int G, int np, const double* __restrict__ parts, double* __restrict__ dp) {   // dp[j] = sum_g parts[g][j], shard order
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    double s = 0.0;
    for (int g = 0; g < G; ++g) s += parts[(long)g * np + j];
    dp[j] = s;
}

static inline long multi_lo(long N, int G, int g) { return (N * g) / G; }

static void multi_free(hipadj_handle* h) {
    for (hipadj_handle* c : h->shards) if (c) (void)hipadj_destroy(c);
    h->shards.clear();
    (void)hipSetDevice(h->cfg.device);
    if (h->d_dp_parts) (void)hipFree(h->d_dp_parts);
    for (hipEvent_t e : h->shard_ev) if (e) (void)hipEventDestroy(e);
    if (h->multi_in) (void)hipEventDestroy(h->multi_in);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
}

static int multi_create(const hipadj_config* cfg, hipadj_handle** out, std::string& cerr) {
    const int G = cfg->ndevices;
    if (G > 64) { cerr = "hipadj_config.ndevices: at most 64 shards per handle"; return HIPADJ_ERR_INVALID_ARG; }
    if (cfg->ntraj < G) { cerr = "hipadj_config.ndevices exceeds the number of trajectories (every shard needs at least one)"; return HIPADJ_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { cerr = hipadj_status_string(HIPADJ_ERR_NO_DEVICE); return HIPADJ_ERR_NO_DEVICE; }
    std::vector<int> ids(G);
    for (int g = 0; g < G; ++g) {
        ids[g] = cfg->device_ids ? cfg->device_ids[g] : g;
        if (ids[g] < 0 || ids[g] >= ndev) { cerr = "hipadj_config.device_ids: ordinal out of range (the box has " + std::to_string(ndev) + " device(s))"; return HIPADJ_ERR_INVALID_ARG; }
    }
    auto* h = new hipadj_handle();
    h->multi = true; h->cfg = *cfg; h->cfg.save_times = nullptr; h->cfg.checkpoints = nullptr; h->cfg.device_ids = nullptr;
    h->cfg.device = ids[0]; h->dev_ids = ids; h->N = cfg->ntraj;
    auto fail = [&](int code, const std::string& msg) { cerr = msg; multi_free(h); delete h; return code; };
    for (int g = 0; g < G; ++g) {
        hipadj_config c = *cfg;
        c.ndevices = 0; c.device_ids = nullptr; c.device = ids[g];
        const long lo = multi_lo(cfg->ntraj, G, g), hi = multi_lo(cfg->ntraj, G, g + 1);
        c.ntraj = hi - lo;
        hipadj_handle* ch = nullptr;
        const int rc = hipadj_create(&c, &ch);
        if (rc != HIPADJ_OK) return fail(rc, "shard " + std::to_string(g) + " (device " + std::to_string(ids[g]) + "): " + hipadj_last_error(nullptr));
        h->shards.push_back(ch); h->shard_off.push_back(lo);
    }
    h->shard_off.push_back(cfg->ntraj);
    const hipadj_handle* c0 = h->shards[0];
    h->n = c0->n; h->np = c0->np; h->M = c0->M; h->S = c0->S;
    if (hipSetDevice(ids[0]) != hipSuccess) return fail(HIPADJ_ERR_HIP, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(HIPADJ_ERR_HIP, "hipStreamCreate failed");
    h->stream = h->own_stream;
    if (hipEventCreateWithFlags(&h->multi_in, hipEventDisableTiming) != hipSuccess) return fail(HIPADJ_ERR_HIP, "hipEventCreate failed");
    if (hipMalloc((void**)&h->d_dp_parts, sizeof(double) * ((size_t)G * (size_t)std::max(h->np, 1) + (size_t)G)) != hipSuccess) return fail(HIPADJ_ERR_HIP, "hipMalloc failed");
    h->shard_ev.assign(G, nullptr);
    for (int g = 0; g < G; ++g) {
        if (hipSetDevice(ids[g]) != hipSuccess || hipEventCreateWithFlags(&h->shard_ev[g], hipEventDisableTiming) != hipSuccess) return fail(HIPADJ_ERR_HIP, "hipEventCreate failed");
    }
    h->dp_host.assign((size_t)G * (size_t)std::max(h->np, 1), 0.0);
    h->st.struct_size = sizeof(hipadj_stats);
    *out = h;
    return HIPADJ_OK;
}

// first failing shard's status and message become the handle's
static int multi_fail(hipadj_handle* h, int g, int rc) {
    h->err = "shard " + std::to_string(g) + " (device " + std::to_string(h->dev_ids[g]) + "): " + hipadj_last_error(h->shards[g]);
    return rc;
}

static int multi_synchronize(hipadj_handle* h) {
    int first = HIPADJ_OK, who = -1;
    for (size_t g = 0; g < h->shards.size(); ++g) { const int rc = hipadj_synchronize(h->shards[g]); if (rc != HIPADJ_OK && first == HIPADJ_OK) { first = rc; who = (int)g; } }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return first == HIPADJ_OK ? HIPADJ_OK : multi_fail(h, who, first);
}

static int multi_forward(hipadj_handle* h, const double* u0, const double* p, double* out) {
    const int G = (int)h->shards.size(); const size_t n = h->n, np = h->np, M = h->M;
    for (int g = 0; g < G; ++g) {
        const size_t lo = (size_t)h->shard_off[g];
        const int rc = forward_host_enqueue(h->shards[g], u0 + lo * n, h->cfg.p_shared ? p : p + lo * np, out ? out + lo * M * n : nullptr);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
    }
    for (int g = 0; g < G; ++g) {      // every shard is running: now drain them one by one and hand the outputs over
        const int rc = forward_host_finish(h->shards[g], out ? out + (size_t)h->shard_off[g] * M * n : nullptr);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
    }
    h->have_forward = true;
    return multi_synchronize(h);
}

static int multi_adjoint(hipadj_handle* h, const double* dLdu, double* du0, double* dp) {
    const int G = (int)h->shards.size(); const size_t n = h->n, np = h->np, M = h->M;
    for (int g = 0; g < G; ++g) {      // phase 1 on every shard: uploads and reverse passes enqueued, nothing waits — the devices run concurrently
        const size_t lo = (size_t)h->shard_off[g];
        const int rc = adjoint_host_run(h->shards[g], dLdu ? dLdu + lo * M * n : nullptr);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
    }
    for (int g = 0; g < G; ++g) {      // phase 2: each shard's downloads once ITS stream is drained
        const size_t lo = (size_t)h->shard_off[g];
        const int rc = adjoint_host_download(h->shards[g], du0 + lo * n, h->cfg.p_shared ? h->dp_host.data() + (size_t)g * np : dp + lo * np);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
    }
    TRY(multi_synchronize(h));
    if (h->cfg.p_shared)      // dp = sum over the shards, in shard order (compare across shard counts at rtol 1e-12, not bitwise: include/hipadj.h)
        for (size_t j = 0; j < np; ++j) { double s = 0.0; for (int g = 0; g < G; ++g) s += h->dp_host[(size_t)g * np + j]; dp[j] = s; }
    return HIPADJ_OK;
}

// ---- device-pointer calls: buffers of the primary device -----------------------------------------------------------------------------------------------------
// every shard stream starts behind the primary stream (the caller's inputs are ready) ...
static int multi_fan_out(hipadj_handle* h) {
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipEventRecord(h->multi_in, h->stream));
    for (size_t g = 0; g < h->shards.size(); ++g) { HIP_TRY(h, hipSetDevice(h->dev_ids[g])); HIP_TRY(h, hipStreamWaitEvent(h->shards[g]->stream, h->multi_in, 0)); }
    return HIPADJ_OK;
}
// ... and the primary stream continues behind all of them
static int multi_fan_in(hipadj_handle* h) {
    for (size_t g = 0; g < h->shards.size(); ++g) { HIP_TRY(h, hipSetDevice(h->dev_ids[g])); HIP_TRY(h, hipEventRecord(h->shard_ev[g], h->shards[g]->stream)); }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (size_t g = 0; g < h->shards.size(); ++g) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->shard_ev[g], 0));
    return HIPADJ_OK;
}
static inline bool multi_local(const hipadj_handle* h, int g) { return h->dev_ids[g] == h->cfg.device; }

static int multi_forward_dev(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const int G = (int)h->shards.size(); const size_t n = h->n, np = h->np, M = h->M; const int d0 = h->cfg.device;
    TRY(multi_fan_out(h));
    for (int g = 0; g < G; ++g) {
        hipadj_handle* c = h->shards[g]; const size_t lo = (size_t)h->shard_off[g], Ng = (size_t)c->N;
        const double* pu = d_u0 + lo * n; const double* pp = h->cfg.p_shared ? d_p : d_p + lo * np; double* po = d_out ? d_out + lo * M * n : nullptr;
        if (!multi_local(h, g)) {   // a shard elsewhere: its own copies of the inputs, its own block for the output
            HIP_TRY(h, hipSetDevice(c->cfg.device));
            HIP_TRY(h, hipMemcpyPeerAsync(c->d_u0, c->cfg.device, pu, d0, sizeof(double) * Ng * n, c->stream));
            HIP_TRY(h, hipMemcpyPeerAsync(c->d_p, c->cfg.device, pp, d0, sizeof(double) * (h->cfg.p_shared ? np : Ng * np), c->stream));
            pu = c->d_u0; pp = c->d_p; if (po) po = c->d_io_a;
        }
        const int rc = hipadj_forward_dev(c, pu, pp, po);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
        if (!multi_local(h, g) && d_out && M > 0) HIP_TRY(h, hipMemcpyPeerAsync(d_out + lo * M * n, d0, c->d_io_a, c->cfg.device, sizeof(double) * Ng * M * n, c->stream));
    }
    h->have_forward = true;
    return multi_fan_in(h);
}

static int multi_adjoint_dev(hipadj_handle* h, const double* d_dLdu, double* d_du0, double* d_dp) {
    const int G = (int)h->shards.size(); const size_t n = h->n, np = h->np, M = h->M; const int d0 = h->cfg.device;
    TRY(multi_fan_out(h));
    for (int g = 0; g < G; ++g) {
        hipadj_handle* c = h->shards[g]; const size_t lo = (size_t)h->shard_off[g], Ng = (size_t)c->N;
        const double* pc = d_dLdu ? d_dLdu + lo * M * n : nullptr; double* pdu = d_du0 + lo * n;
        double* pdp = h->cfg.p_shared ? h->d_dp_parts + (size_t)g * np : d_dp + lo * np;      // shared parameters: the shard's partial, summed below
        const bool local = multi_local(h, g);
        if (!local) {
            HIP_TRY(h, hipSetDevice(c->cfg.device));
            if (pc && M > 0) { HIP_TRY(h, hipMemcpyPeerAsync(c->d_io_a, c->cfg.device, pc, d0, sizeof(double) * Ng * M * n, c->stream)); pc = c->d_io_a; }
            pdu = c->d_du0; pdp = c->d_dp;
        }
        const int rc = hipadj_adjoint_dev(c, pc, pdu, pdp);
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
        if (!local) {
            HIP_TRY(h, hipMemcpyPeerAsync(d_du0 + lo * n, d0, c->d_du0, c->cfg.device, sizeof(double) * Ng * n, c->stream));
            HIP_TRY(h, hipMemcpyPeerAsync(h->cfg.p_shared ? h->d_dp_parts + (size_t)g * np : d_dp + lo * np, d0, c->d_dp, c->cfg.device, sizeof(double) * (h->cfg.p_shared ? np : Ng * np), c->stream));
        }
    }
    TRY(multi_fan_in(h));
    if (h->cfg.p_shared) {
        hipLaunchKernelGGL(k_sum_rows, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, h->stream, G, (int)np, (const double*)h->d_dp_parts, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    return HIPADJ_OK;
}

static int multi_set_loss_data(hipadj_handle* h, const double* data, bool dev) {
    const size_t n = h->n, M = h->M; const int d0 = h->cfg.device;
    if (dev) TRY(multi_fan_out(h));
    for (size_t g = 0; g < h->shards.size(); ++g) {
        hipadj_handle* c = h->shards[g]; const size_t lo = (size_t)h->shard_off[g], Ng = (size_t)c->N;
        int rc;
        if (!dev) rc = hipadj_set_loss_data(c, data + lo * M * n);
        else if (multi_local(h, (int)g)) rc = hipadj_set_loss_data_dev(c, data + lo * M * n);
        else {   // through the shard's staging block, then the ordinary device entry point (which copies it into the shard's own data block)
            HIP_TRY(h, hipSetDevice(c->cfg.device));
            HIP_TRY(h, hipMemcpyPeerAsync(c->d_io_a, c->cfg.device, data + lo * M * n, d0, sizeof(double) * Ng * M * n, c->stream));
            rc = hipadj_set_loss_data_dev(c, c->d_io_a);
        }
        if (rc != HIPADJ_OK) return multi_fail(h, (int)g, rc);
    }
    h->have_ldata = true;
    return dev ? multi_fan_in(h) : HIPADJ_OK;
}

static int multi_loss_value(hipadj_handle* h, const double* out, double* loss, bool dev) {
    const int G = (int)h->shards.size(); const size_t n = h->n, M = h->M; const int d0 = h->cfg.device;
    if (!dev) {
        double s = 0.0;
        for (int g = 0; g < G; ++g) { double v = 0.0; const int rc = hipadj_loss_value(h->shards[g], out + (size_t)h->shard_off[g] * M * n, &v); if (rc != HIPADJ_OK) return multi_fail(h, g, rc); s += v; }
        *loss = s;
        return HIPADJ_OK;
    }
    double* parts = h->d_dp_parts + (size_t)G * (size_t)std::max(h->np, 1);     // G doubles behind the dp partials (multi_create)
    TRY(multi_fan_out(h));
    for (int g = 0; g < G; ++g) {
        hipadj_handle* c = h->shards[g]; const size_t lo = (size_t)h->shard_off[g], Ng = (size_t)c->N;
        const double* po = out + lo * M * n;
        int rc;
        if (multi_local(h, g)) rc = hipadj_loss_value_dev(c, po, parts + g);
        else {
            HIP_TRY(h, hipSetDevice(c->cfg.device));
            HIP_TRY(h, hipMemcpyPeerAsync(c->d_io_a, c->cfg.device, po, d0, sizeof(double) * Ng * M * n, c->stream));
            if (!c->d_lval && hipMalloc((void**)&c->d_lval, sizeof(double)) != hipSuccess) { (void)hipGetLastError(); HIPADJ_FAIL(h, HIPADJ_ERR_HIP, "hipMalloc failed"); }
            rc = hipadj_loss_value_dev(c, c->d_io_a, c->d_lval);
            if (rc == HIPADJ_OK) HIP_TRY(h, hipMemcpyPeerAsync(parts + g, d0, c->d_lval, c->cfg.device, sizeof(double), c->stream));
        }
        if (rc != HIPADJ_OK) return multi_fail(h, g, rc);
    }
    TRY(multi_fan_in(h));
    hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(256), 0, h->stream, G, 1, (const double*)parts, loss);   // the shards' values in shard order
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}

static int multi_get_stats(hipadj_handle* h, hipadj_stats* st) {
    hipadj_stats a{}; a.struct_size = sizeof(hipadj_stats);
    for (size_t g = 0; g < h->shards.size(); ++g) {
        hipadj_stats s{}; s.struct_size = sizeof(hipadj_stats);
        const int rc = hipadj_get_stats(h->shards[g], &s);
        if (rc != HIPADJ_OK) return multi_fail(h, (int)g, rc);
        if (g == 0) a = s;
        else {
            a.ntraj += s.ntraj; a.workspace_bytes += s.workspace_bytes; a.adjoint_algorithmic_bytes += s.adjoint_algorithmic_bytes; a.vjp_steps += s.vjp_steps;
            // the shards run concurrently: the pass takes as long as the slowest one
            a.forward_ms_last = std::max(a.forward_ms_last, s.forward_ms_last); a.adjoint_ms_last = std::max(a.adjoint_ms_last, s.adjoint_ms_last);
            a.forward_ms_total = std::max(a.forward_ms_total, s.forward_ms_total); a.adjoint_ms_total = std::max(a.adjoint_ms_total, s.adjoint_ms_total);
            a.adjoint_main_kernel_ms_last = std::max(a.adjoint_main_kernel_ms_last, s.adjoint_main_kernel_ms_last);
            a.adjoint_main_kernel_ms_total = std::max(a.adjoint_main_kernel_ms_total, s.adjoint_main_kernel_ms_total);
        }
    }
    *st = a;
    return HIPADJ_OK;
}
