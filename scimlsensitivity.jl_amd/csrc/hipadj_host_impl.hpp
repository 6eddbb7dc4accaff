// hipadj_host_impl.hpp — the launch sequences (host code) of every kernel family.  Included by the hipadj_tu_*.hip units only:
// each unit explicitly instantiates the sequences of one compiled-in model / family, so that the device code of the families
// compiles in parallel.
#pragma once
#include "hipadj_host.hpp"
#include "hipadj_quad_ts5.hpp"

template <class Mo> int forward_impl(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    dbl2* knots = h->d_knots;
    double* ck = h->d_ckpt;
    if (h->fwd_ev && h->d_fev_knot && h->quad_fwd && QuadForm<Mo>::value)     // four lanes per trajectory (hipadj_quad.hpp)
        hipLaunchKernelGGL((k_forward_quad<Mo>), dim3((unsigned)((h->N + WAVE / 4 - 1) / (WAVE / 4))), dim3(WAVE), 0, h->stream, h->g, d_u0, d_p,
                           FwdEvents{h->d_fev_knot, h->d_fev_save, h->d_fev_ckpt, h->nfev}, knots, ck,
                           (d_out && h->M > 0 && !h->offgrid) ? h->d_outT : (double*)nullptr, h->d_yT);
    else if (h->fwd_ev && h->d_fev_knot)
        hipLaunchKernelGGL((k_forward_ev<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, d_u0, d_p,
                           FwdEvents{h->d_fev_knot, h->d_fev_save, h->d_fev_ckpt, h->nfev}, knots, ck,
                           (d_out && h->M > 0 && !h->offgrid) ? h->d_outT : (double*)nullptr, h->d_yT);
    else
    hipLaunchKernelGGL((k_forward<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, d_u0, d_p, knots, ck,
                       h->d_ckpt_of_knot, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, h->d_save_of_knot, h->d_yT);
    HIP_TRY(h, hipGetLastError());
    if (h->offgrid && d_out && h->M > 0) {   // out = sol(ts) by interpolation: the save times are not knots
        hipLaunchKernelGGL((k_out_offgrid<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, (const dbl2*)knots, (const double*)h->d_save_t, h->M, h->d_outT);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->offgrid && h->d_ckpt) {           // Backsolve: the checkpoint states at the (off-grid) checkpoint times
        hipLaunchKernelGGL((k_out_offgrid<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, (const dbl2*)knots, (const double*)h->d_ck_t, h->nck, h->d_ckpt);
        HIP_TRY(h, hipGetLastError());
    }
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}

template <class Mo, int LOSS> int adjoint_impl_l(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    // software-prefetch depth, chosen so that each kernel keeps 2 waves per SIMD (<= 256 VGPRs): the cotangent ring
    // and the Gauss-node state cost registers
    constexpr int PF = (LOSS & 1) == 1 ? 8 : 6, PFG = 4;   // LOSS here = MODE = discrete-loss kind | (continuous cost << 1)
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    const unsigned cblocks = (unsigned)((h->N + FIN / 4 - 1) / (FIN / 4));   // composition: 4 lanes per trajectory
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;   // per-trajectory dp rows [N][np]
    double* dp_sum = (h->cfg.p_shared && h->fused_final) ? d_dp : (double*)nullptr;   // in-launch last-arriver reduction (optional)
    // composition workgroups: 64 trajectories each, or 16 each while that still leaves the chip short of workgroups
    // (10^4 trajectories: 625 instead of 157 workgroups, -1.7 us per reverse pass; profiles/README.md)
    const bool small_blocks = h->cbs == 64 || (h->cbs == 0 && cblocks < 1024);
    unsigned compose_blocks = small_blocks ? (unsigned)((h->N + 15) / 16) : cblocks;
    static const bool compose16 = []() { const char* e = std::getenv("HIPADJ_COMPOSE16"); return !(e && e[0] == '0'); }();   // A/B switch of the 16-lane composition
    // lane-per-trajectory form (coalesced rows, maps spread over the waves of a workgroup) while a wave folds at most three maps: up to 24 lower
    // maps, i.e. >= ~5000 trajectories at 1000 steps (10^4: tail 20.8 -> 16.0 us, 5000: 21 -> 18 us); with more maps per trajectory (small shards,
    // 50-62 segments) the 16-lane form's shorter chains win (profiles/r2_compose_w_segments.log).  HIPADJ_COMPOSE_W = 0 / 4 / 8 forces a form.
    static const int compose_w = []() { const char* e = std::getenv("HIPADJ_COMPOSE_W"); return e ? std::atoi(e) : -1; }();
    const unsigned wblocks = (unsigned)((h->N + WAVE - 1) / WAVE);
    const int lower_maps = h->nseg - 1;
    const bool wave_form = h->nseg > 1 && (compose_w > 0 || (compose_w < 0 && lower_maps <= 24));
    const int wave_w = compose_w > 0 ? compose_w : (lower_maps <= 12 ? 4 : 8);
    if (wave_form) compose_blocks = wblocks;
    auto launch_compose = [&]() {
        if (wave_form && wave_w >= 8)
            hipLaunchKernelGGL((k_compose_finish_w<Mo, 8>), dim3(wblocks), dim3(WAVE * 8), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        else if (wave_form)
            hipLaunchKernelGGL((k_compose_finish_w<Mo, 4>), dim3(wblocks), dim3(WAVE * 4), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        else if (small_blocks && compose16)      // 16 lanes per trajectory, FIN / 16 = 16 trajectories per workgroup: the same number of workgroups (and partials)
            hipLaunchKernelGGL((k_compose_finish16<Mo>), dim3(compose_blocks), dim3(FIN), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        else if (small_blocks)
            hipLaunchKernelGGL((k_compose_finish<Mo, 64>), dim3(compose_blocks), dim3(64), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        else
            hipLaunchKernelGGL((k_compose_finish<Mo>), dim3(compose_blocks), dim3(FIN), 0, h->stream, h->g, h->nseg, (const double*)h->d_segbuf,
                               d_du0, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
    };
    // the column the sweep streams at the loss times: the caller's block already in the streaming layout (hipadj_adjoint_dev_soa), or the handle's own buffer — cotangents
    // transposed here per pass, or the data block of a device-resident loss transposed ONCE by hipadj_set_loss_data
    const double* cotT = h->cot_soa ? h->cot_soa : h->d_cotT;
    // cotangents in the pullback's layout [N][M][n]: the one-launch Interpolating sweep reads them in place (template variant HIPADJ_MODE_COT_INPLACE, hipadj_lane.hpp
    // load_cot), every other sequence keeps the transposition launch.  HIPADJ_COT_INPLACE=0: always the launch (A/B)
    static const bool insweep_on = []() { const char* e = std::getenv("HIPADJ_COT_INPLACE"); return !(e && e[0] == '0'); }();
    const bool one_launch = LOSS == 0 && h->fused && h->d_tbuf && !h->offgrid && !h->ip_ckpt && h->cfg.alg == HIPADJ_ALG_INTERPOLATING && !h->wpb4;   // = the k_interp_fused branch below
    const bool cot_aos_in = h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0 && !h->cot_soa;
    const bool insweep = cot_aos_in && one_launch && insweep_on && (double)h->N * h->M * h->n * 8.0 < 2147483648.0;
    if (cot_aos_in && !insweep) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);                   // ring full: only now wait for the oldest call
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    hipEvent_t k0 = es.k0, k1 = es.k1;          // dominant-kernel bracket
    bool dispatch_events = false;               // k0/k1 ride on the kernel's dispatch packet instead (k_interp, below)
    const bool fused_main = h->fused && h->d_tbuf && !h->offgrid && !h->ip_ckpt;   // the sweep kernel carries k0 / k1 on its own dispatch packet
    if (h->timing >= 1 && !fused_main && (h->offgrid || !(h->cfg.alg == HIPADJ_ALG_INTERPOLATING && !h->ip_ckpt))) HIP_TRY(h, hipEventRecord(k0, h->stream));
    if (h->offgrid && h->nseg > 1) {   // off-grid Interpolating / Gauss, time-segmented over the reverse step list (k_offgrid_seg + composition)
        RevSteps R{h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->nrs, h->rs_save_at_start, h->cfg.t1};
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->cfg.alg == HIPADJ_ALG_GAUSS)
            hipLaunchKernelGGL((k_offgrid_seg<Mo, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, R, sp, p, (const dbl2*)h->d_knots, cotT, h->d_segbuf);
        else
            hipLaunchKernelGGL((k_offgrid_seg<Mo, LOSS, false>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, R, sp, p, (const dbl2*)h->d_knots, cotT, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        if (h->cfg.p_shared && !h->fused_final) {
            hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)compose_blocks, h->np, (const double*)h->d_partial, d_dp);
            HIP_TRY(h, hipGetLastError());
        }
        if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
        es.pending = h->timing >= 1; es.full = h->timing >= 2;
        return HIPADJ_OK;
    }
    if (h->offgrid && h->cfg.alg == HIPADJ_ALG_QUADRATURE) {   // dense lambda over the reverse step list, then adaptive GK15 per (trajectory, loss interval)
        RevSteps R{h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->nrs, h->rs_save_at_start, h->cfg.t1};
        hipLaunchKernelGGL((k_quad_adj_offgrid<Mo, LOSS>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, p, (const dbl2*)h->d_knots, cotT, h->d_adj, d_du0, (double*)nullptr);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_quad_gk_offgrid<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nq), dim3(WAVE), 0, h->stream, h->g, R, p, (const dbl2*)h->d_knots,
                           (const dbl2*)h->d_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                           (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        HIP_TRY(h, hipGetLastError());
        if (h->cfg.p_shared && !h->fused_final) {
            hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)fblocks, h->np, (const double*)h->d_partial, d_dp);
            HIP_TRY(h, hipGetLastError());
        }
        if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
        es.pending = h->timing >= 1; es.full = h->timing >= 2;
        return HIPADJ_OK;
    }
    if (h->offgrid) {   // loss times off the step grid, one column, sequential in time (Backsolve; runtime models; forced time_segments = 1)
        RevSteps R{h->d_rs_t, h->d_rs_h, h->d_rs_te, h->d_rs_save, h->d_rs_ck, h->nrs, h->rs_save_at_start, h->cfg.t1};
        if (h->og_ck) {   // checkpointing = true: per-interval re-solve into the lane's knot tile (offgrid_ckpt_lane)
            const OgIntervals I{h->d_og_i, h->d_og_i + h->og_nint, h->d_og_i + 2 * h->og_nint, h->d_og_h, h->d_ck_t, h->og_nint};
            if (h->cfg.alg == HIPADJ_ALG_INTERPOLATING)
                hipLaunchKernelGGL((k_offgrid_ckpt<Mo, LOSS, 0>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, I, p, (const double*)h->d_ckpt, h->d_og_tile, cotT, d_du0, h->d_dp_traj);
            else if (h->cfg.alg == HIPADJ_ALG_GAUSS)
                hipLaunchKernelGGL((k_offgrid_ckpt<Mo, LOSS, 2>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, I, p, (const double*)h->d_ckpt, h->d_og_tile, cotT, d_du0, h->d_dp_traj);
            else
                hipLaunchKernelGGL((k_offgrid_ckpt<Mo, LOSS, 4>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, I, p, (const double*)h->d_ckpt, h->d_og_tile, cotT, d_du0, h->d_dp_traj);
        } else
        if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE)
            hipLaunchKernelGGL((k_backsolve_offgrid<Mo, (LOSS >> 1)>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, p, (const double*)h->d_yT, (const double*)h->d_ckpt, cotT, d_du0, h->d_dp_traj);
        else if (h->cfg.alg == HIPADJ_ALG_GAUSS) {
            hipLaunchKernelGGL((k_gauss_offgrid<Mo, LOSS>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, p, (const dbl2*)h->d_knots, cotT, d_du0, h->d_dp_traj);
        } else if (h->cfg.alg == HIPADJ_ALG_GAUSS_KRONROD) {
            hipLaunchKernelGGL((k_gauss_offgrid<Mo, LOSS, true>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, p, (const dbl2*)h->d_knots, cotT, d_du0, h->d_dp_traj);
        } else
        hipLaunchKernelGGL((k_interp_offgrid<Mo, LOSS>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, R, p, (const dbl2*)h->d_knots, cotT, d_du0, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                           (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        HIP_TRY(h, hipGetLastError());
        if (h->cfg.p_shared && !h->fused_final) {
            hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)fblocks, h->np, (const double*)h->d_partial, d_dp);
            HIP_TRY(h, hipGetLastError());
        }
        if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
        es.pending = h->timing >= 1; es.full = h->timing >= 2;
        return HIPADJ_OK;
    }
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->ip_ckpt && h->ck_long)
            hipLaunchKernelGGL((k_interp_ckpt<Mo, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, cotT, (const int*)h->d_save_rev, h->d_segbuf, h->d_gtile, h->gtile_stride);
        else if (h->ip_ckpt)
            hipLaunchKernelGGL((k_interp_ckpt<Mo, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, cotT, (const int*)h->d_save_rev, h->d_segbuf, h->d_gtile, h->gtile_stride);
        else if (h->fused && h->d_tbuf && !h->wpb4) {
            // ONE launch per reverse pass (hipadj_fused.hpp): the sweep's waves compose their segment maps as a tree through HBM, the root wave of
            // each block writes du0 and its partial of dp, the last block sums the partials.  No k_compose_finish*, no k_reduce_final.
            TreePlan tp = h->tp; tp.tbuf = h->d_tbuf; tp.cnt = h->d_tcnt; tp.partial = h->d_partial; tp.ticket = h->d_ticket;
#ifdef HIPADJ_WAVE_TRACE
            tp.trace = g_hipadj_wave_trace; h->g.trace = g_hipadj_wave_trace;      // development builds: per-wave time stamps of this launch (hipadj_debug_set_trace)
#endif
            const hipEvent_t e0 = h->timing >= 1 ? k0 : (hipEvent_t) nullptr, e1 = h->timing >= 1 ? k1 : (hipEvent_t) nullptr;
            double* dps = h->cfg.p_shared ? d_dp : (double*)nullptr;
            bool launched = false;
            if constexpr (model_has_ops<Mo>::value && (LOSS >> 1) == 0) {
                if (h->fgroup) {      // (the planner set it only for the configuration this branch serves: shared parameters, several segments, no cost — plan_group_choice)
                    // G waves per workgroup, first composition level in LDS (k_interp_fused_g); the loss forms of the plain one-launch pass: fused LSQ_SHIFT, a streamed
                    // column (cotangents in the streaming layout, the data block of LSQ_DATA), cotangents read in place from the pullback's layout
                    const unsigned ng = (unsigned)((h->nseg + h->fgroup - 1) / h->fgroup);
                    auto go = [&](auto mode, auto gsz, const double* col) {
                        constexpr int MODE = decltype(mode)::value, GS = decltype(gsz)::value;
                        hipExtLaunchKernelGGL((k_interp_fused_g<Mo, 4, MODE, GS>), dim3(waves, ng), dim3(WAVE * GS), 0, h->stream, e0, e1, 0, h->g, sp, tp, p,
                                              (const dbl2*)h->d_knots, col, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag);
                    };
                    using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
                    bool inplace = false;
                    if constexpr (LOSS == 0) inplace = insweep;
                    if (inplace) { if (h->fgroup == 8) go(std::integral_constant<int, HIPADJ_MODE_COT_INPLACE>{}, I8{}, d_cot); else go(std::integral_constant<int, HIPADJ_MODE_COT_INPLACE>{}, I4{}, d_cot); }
                    else { if (h->fgroup == 8) go(std::integral_constant<int, LOSS>{}, I8{}, cotT); else go(std::integral_constant<int, LOSS>{}, I4{}, cotT); }
                    launched = true;
                }
                if (!launched && h->cfg.p_shared && h->nseg > 1 && !h->no_ops) {
                    if constexpr (LOSS == 0) {
                        if (insweep) {
                            hipExtLaunchKernelGGL((k_interp_fused<Mo, 4, HIPADJ_MODE_COT_INPLACE, true, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, tp, p,
                                                  (const dbl2*)h->d_knots, d_cot, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag);
                            launched = true;
                        }
                    }
                    if (!launched)
                    hipExtLaunchKernelGGL((k_interp_fused<Mo, 4, LOSS, true, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, tp, p,
                                          (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag);
                    launched = true;
                }
            }
            if constexpr (LOSS == 0) {
                if (!launched && insweep) {
                    hipExtLaunchKernelGGL((k_interp_fused<Mo, PF, HIPADJ_MODE_COT_INPLACE>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, tp, p,
                                          (const dbl2*)h->d_knots, d_cot, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag);
                    launched = true;
                }
            }
            if (!launched)
                hipExtLaunchKernelGGL((k_interp_fused<Mo, PF, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, tp, p,
                                      (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, d_du0, dp_rows, dps, h->d_flag);
            HIP_TRY(h, hipGetLastError());
            if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
            es.pending = h->timing >= 1; es.full = h->timing >= 2;
            return HIPADJ_OK;
        }
        else if (h->wpb4) {
            // 256-thread workgroups, four (wave block, segment) items each: one wave per SIMD by construction (hipadj_kernels.hpp)
            const unsigned items = waves * (unsigned)h->nseg;
            hipExtLaunchKernelGGL((k_interp<Mo, PF, LOSS, true, 4>), dim3((items + 3) / 4), dim3(4 * WAVE), 0, h->stream, h->timing >= 1 ? k0 : (hipEvent_t) nullptr,
                                  h->timing >= 1 ? k1 : (hipEvent_t) nullptr, 0, h->g, sp, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, h->d_segbuf);
            dispatch_events = true;
        } else {
            // the dominant kernel's own begin/end timestamps (events attached to the dispatch packet): what rocprofv3 reports
            // as the kernel's duration.  A hipEventRecord pair around the launch also counts the two marker packets and the
            // dispatch latency (+8-10 us on a 0.12 ms kernel).
            const hipEvent_t e0 = h->timing >= 1 ? k0 : (hipEvent_t) nullptr, e1 = h->timing >= 1 ? k1 : (hipEvent_t) nullptr;
            bool launched = false;
            if constexpr (model_has_ops<Mo>::value && (LOSS >> 1) == 0) {
                // shared parameters + several time segments: the stage-operator form of the multi-column step (adj_rk4_step_ops)
                if (h->cfg.p_shared && h->nseg > 1 && !h->no_ops) {
                    // prefetch depth 4: the multi-segment launch is bound by FP64 issue, not by HBM latency (PF 8 / 6 / 4 / 3 measure within 2 % of
                    // each other, 4 marginally best and half the code of 8: profiles/r2_kbench_visit9_prefetch_depth.log)
                    hipExtLaunchKernelGGL((k_interp<Mo, 4, LOSS, true, 1, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, p,
                                          (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, h->d_segbuf);
                    launched = true;
                }
            }
            if (!launched)
                hipExtLaunchKernelGGL((k_interp<Mo, PF, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, e0, e1, 0, h->g, sp, p,
                                      (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev, h->d_segbuf);
            dispatch_events = h->timing >= 1;
        }
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1 && !dispatch_events) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_BACKSOLVE: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->fused && h->d_tbuf) {   // one launch per reverse pass (hipadj_fused.hpp)
            TreePlan tp = h->tp; tp.tbuf = h->d_tbuf; tp.cnt = h->d_tcnt; tp.partial = h->d_partial; tp.ticket = h->d_ticket;
            hipExtLaunchKernelGGL((k_backsolve_fused<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->timing >= 1 ? k0 : (hipEvent_t) nullptr,
                                  h->timing >= 1 ? k1 : (hipEvent_t) nullptr, 0, h->g, sp, tp, p, (const double*)h->d_yT, (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot,
                                  cotT, (const int*)h->d_save_rev, d_du0, dp_rows, h->cfg.p_shared ? d_dp : (double*)nullptr, h->d_flag);
            HIP_TRY(h, hipGetLastError());
            if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
            es.pending = h->timing >= 1; es.full = h->timing >= 2;
            return HIPADJ_OK;
        }
        hipLaunchKernelGGL((k_backsolve<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_yT,
                           (const double*)h->d_ckpt, (const int*)h->d_ckpt_of_knot, cotT,
                           (const int*)h->d_save_rev, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_GAUSS: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->ip_ckpt && h->ck_long)
            hipLaunchKernelGGL((k_gauss_ckpt<Mo, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, cotT, (const int*)h->d_save_rev, h->d_segbuf, h->d_gtile, h->gtile_stride);
        else if (h->ip_ckpt)
            hipLaunchKernelGGL((k_gauss_ckpt<Mo, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const double*)h->d_ckpt,
                               (const int*)h->d_ckpt_of_knot, (const int*)h->d_prev_ck, cotT, (const int*)h->d_save_rev, h->d_segbuf, h->d_gtile, h->gtile_stride);
        else if (h->fused && h->d_tbuf) {   // one launch per reverse pass (hipadj_fused.hpp)
            TreePlan tp = h->tp; tp.tbuf = h->d_tbuf; tp.cnt = h->d_tcnt; tp.partial = h->d_partial; tp.ticket = h->d_ticket;
            hipExtLaunchKernelGGL((k_gauss_fused<Mo, PFG, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->timing >= 1 ? k0 : (hipEvent_t) nullptr,
                                  h->timing >= 1 ? k1 : (hipEvent_t) nullptr, 0, h->g, sp, tp, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev,
                                  d_du0, dp_rows, h->cfg.p_shared ? d_dp : (double*)nullptr, h->d_flag);
            HIP_TRY(h, hipGetLastError());
            if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
            es.pending = h->timing >= 1; es.full = h->timing >= 2;
            return HIPADJ_OK;
        }
        else
            hipLaunchKernelGGL((k_gauss<Mo, PFG, LOSS>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const dbl2*)h->d_knots,
                               cotT, (const int*)h->d_save_rev, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_GAUSS_KRONROD: {
        SegPlan sp{h->nseg, h->d_seg_bounds};
        if (h->fused && h->d_tbuf) {   // one launch per reverse pass (hipadj_fused.hpp)
            TreePlan tp = h->tp; tp.tbuf = h->d_tbuf; tp.cnt = h->d_tcnt; tp.partial = h->d_partial; tp.ticket = h->d_ticket;
            hipExtLaunchKernelGGL((k_gauss_fused<Mo, PFG, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->timing >= 1 ? k0 : (hipEvent_t) nullptr,
                                  h->timing >= 1 ? k1 : (hipEvent_t) nullptr, 0, h->g, sp, tp, p, (const dbl2*)h->d_knots, cotT, (const int*)h->d_save_rev,
                                  d_du0, dp_rows, h->cfg.p_shared ? d_dp : (double*)nullptr, h->d_flag);
            HIP_TRY(h, hipGetLastError());
            if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
            es.pending = h->timing >= 1; es.full = h->timing >= 2;
            return HIPADJ_OK;
        }
        hipLaunchKernelGGL((k_gauss<Mo, PFG, LOSS, true>), dim3(waves, (unsigned)h->nseg), dim3(WAVE), 0, h->stream, h->g, sp, p, (const dbl2*)h->d_knots,
                           cotT, (const int*)h->d_save_rev, h->d_segbuf);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        launch_compose();
        HIP_TRY(h, hipGetLastError());
        break; }
    case HIPADJ_ALG_QUADRATURE: {
        hipLaunchKernelGGL((k_quad_adj<Mo, PF, LOSS>), dim3(waves), dim3(WAVE), 0, h->stream, h->g, p, (const dbl2*)h->d_knots,
                           cotT, (const int*)h->d_save_rev, h->d_adj, d_du0, (double*)nullptr);
        HIP_TRY(h, hipGetLastError());
        if (h->timing >= 1) HIP_TRY(h, hipEventRecord(k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_quad_gk<Mo, (LOSS >> 1)>), dim3(waves, (unsigned)h->nq), dim3(WAVE), 0, h->stream, h->g, p, (const dbl2*)h->d_knots,
                           (const dbl2*)h->d_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        break; }
    }
    // finishing stage: NaN/Inf scan + per-workgroup partial sums of mu (Interpolating fused it with the composition)
    if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                           (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, dp_sum);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->cfg.p_shared && !h->fused_final) {   // dp = sum over workgroup partials, fixed order
        const unsigned nb = h->cfg.alg == HIPADJ_ALG_QUADRATURE ? fblocks : compose_blocks;
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)nb, h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;   // timings are harvested lazily at the next synchronize / call
}

template <class Mo> int adjoint_impl(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    // no loss times => no cotangent buffer exists: run the LSQ specialisation (its jump select is never taken)
    const int mode = (loss_streams(h) ? 0 : 1) | (h->cfg.cont_cost << 1);   // MODE = loss | cost << 1; loss bit 0: a column is streamed (cotangents, or the data of HIPADJ_LOSS_LSQ_DATA)
    switch (mode) {
    case 0: return adjoint_impl_l<Mo, 0>(h, d_cot, d_du0, d_dp);
    case 1: return adjoint_impl_l<Mo, 1>(h, d_cot, d_du0, d_dp);
    case 2: return adjoint_impl_l<Mo, 2>(h, d_cot, d_du0, d_dp);
    case 3: return adjoint_impl_l<Mo, 3>(h, d_cot, d_du0, d_dp);
    case 4: return adjoint_impl_l<Mo, 4>(h, d_cot, d_du0, d_dp);
    case 5: return adjoint_impl_l<Mo, 5>(h, d_cot, d_du0, d_dp);
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "cont_cost %d is not available for compiled-in models", h->cfg.cont_cost);
    }
}

// ---- workgroup-per-trajectory family (Brusselator) -----------------------------------------------------
template <int G> int field_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    if (h->cfg.stepper == HIPADJ_STEPPER_ETDRK4_FIXED) {   // the exponential stepper (hipadj_field_etd.hpp)
        hipLaunchKernelGGL((k_bruss_forward_etd<G>), dim3((unsigned)h->N), dim3(Bruss<G>::T), 0, h->stream, h->fg, d_u0, d_p, h->d_fknots,
                           (d_out && h->M > 0) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot);
        HIP_TRY(h, hipGetLastError());
        return HIPADJ_OK;
    }
    hipLaunchKernelGGL((k_bruss_forward<G>), dim3((unsigned)h->N), dim3(Bruss<G>::T), 0, h->stream, h->fg, d_u0, d_p, h->d_fknots,
                       (d_out && h->M > 0) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
template <int G> int field_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const double* p = h->p_dev_last;
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    const dim3 grid((unsigned)h->N), blk(Bruss<G>::T);
    const bool etd = h->cfg.stepper == HIPADJ_STEPPER_ETDRK4_FIXED;
    const int cc = h->cfg.cont_cost;      // continuous cost: a compile-time variant of the RK4 kernels (a run-time branch cost the cost-free kernels 16-24 registers and spills)
    switch (h->cfg.alg) {
    case HIPADJ_ALG_INTERPOLATING:
        if (etd)
            hipLaunchKernelGGL((k_bruss_adjoint_etd<G, 0>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                               (const int*)h->d_save_rev, (double*)nullptr, d_du0, h->d_dp_traj, h->d_flag);
        else if (cc == 1)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 0, 1>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else if (cc == 2)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 0, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else
        hipLaunchKernelGGL((k_bruss_adjoint<G, 0>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        break;
    case HIPADJ_ALG_GAUSS:
        if (etd)
            hipLaunchKernelGGL((k_bruss_adjoint_etd<G, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                               (const int*)h->d_save_rev, (double*)nullptr, d_du0, h->d_dp_traj, h->d_flag);
        else if (cc == 1)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 2, 1>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else if (cc == 2)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 2, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else
        hipLaunchKernelGGL((k_bruss_adjoint<G, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        break;
    case HIPADJ_ALG_GAUSS_KRONROD:      // the adaptive (7,15) rule per step in the Gauss sweep (RK4 only: the planner refuses the exponential stepper)
        if (cc == 1)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 4, 1>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else if (cc == 2)
            hipLaunchKernelGGL((k_bruss_adjoint<G, 4, 2>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        else
            hipLaunchKernelGGL((k_bruss_adjoint<G, 4>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, d_du0, h->d_dp_traj, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        break;
    case HIPADJ_ALG_QUADRATURE: {
        if (etd)     // pass 1 with the exponential stepper; pass 2 (k_bruss_quad_gk) integrates over the same knots and the same dense record
            hipLaunchKernelGGL((k_bruss_adjoint_etd<G, 3>), grid, blk, 0, h->stream, h->fg, h->Npad, p, (const double*)h->d_fknots, d_cot,
                               (const int*)h->d_save_rev, h->d_fadj, d_du0, (double*)nullptr, h->d_flag);
        else if (cc == 1)
            hipLaunchKernelGGL((k_bruss_quad_adj<G, 1>), grid, blk, 0, h->stream, h->fg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, h->d_fadj, d_du0, h->d_flag);
        else if (cc == 2)
            hipLaunchKernelGGL((k_bruss_quad_adj<G, 2>), grid, blk, 0, h->stream, h->fg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, h->d_fadj, d_du0, h->d_flag);
        else
        hipLaunchKernelGGL((k_bruss_quad_adj<G>), grid, blk, 0, h->stream, h->fg, p, (const double*)h->d_fknots, d_cot,
                           (const int*)h->d_save_rev, h->d_fadj, d_du0, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        if (cc == 2)      // only this cost has a g_p for the integrand
            hipLaunchKernelGGL((k_bruss_quad_gk<G, 128, 2>), dim3((unsigned)h->N, (unsigned)h->nq), blk, 0, h->stream, h->fg, h->Npad, p,
                               (const double*)h->d_fknots, (const double*)h->d_fadj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        else
        hipLaunchKernelGGL((k_bruss_quad_gk<G, 128>), dim3((unsigned)h->N, (unsigned)h->nq), blk, 0, h->stream, h->fg, h->Npad, p,
                           (const double*)h->d_fknots, (const double*)h->d_fadj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        const unsigned waves = (unsigned)(h->Npad / WAVE);
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
        break; }
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg not available for the PDE family");
    }
    hipLaunchKernelGGL((k_finish<0, 3>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                       (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, h->cfg.p_shared ? d_dp : (double*)nullptr);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = true;
    return HIPADJ_OK;
}

// ---- FP64-MFMA family (MLP neural ODE) -----------------------------------------------------------------
template <int H> int mlp_forward_launch(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    hipLaunchKernelGGL((k_mlp_forward<H>), dim3((unsigned)(h->mg.B / 16), (unsigned)h->N), dim3(Mlp<H>::NT), 0, h->stream, h->mg, d_u0, d_p,
                       h->d_fknots, (d_out && h->M > 0) ? d_out : (double*)nullptr, (const int*)h->d_save_of_knot);
    HIP_TRY(h, hipGetLastError());
    return HIPADJ_OK;
}
// QuadratureAdjoint pass 2 for the FP64-MFMA family: quadgk(integrand, t_{i-1}, t_i; atol, rtol) for every trajectory and loss interval
// (src/quadrature_adjoint.jl:573-616), integrand = f_p^T lam summed over the whole batch.  The error norm of a panel is a norm over all
// parameters AND all columns, so the adaptive decisions cannot be taken inside a workgroup: the panels are evaluated on the device
// (k_mlp_quad_panel: 15-point Kronrod and 7-point Gauss sums as two entries, per-workgroup partials, fixed-order reduction into a pool of
// panel vectors), the norms come back to the host, and the host runs QuadGK's loop — global error heap per interval, bisect the worst
// segment, stop at E <= max(atol, rtol |I|), final re-sum — exactly as the oracle's quadgk_vec does, one bisection per unfinished interval
// and round (the intervals are independent, so batching them changes nothing).  Smooth problems finish in round 0.
template <int H> int mlp_quadrature(hipadj_handle* h, const double* p, double* d_dp) {
    constexpr int NPAR = Mlp<H>::NPAR;
    const int nWG = h->mg.B / 16, nq = h->nq; const long N = h->N;
    const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
    struct Seg { double a, b, E; int idK; };
    struct Item { std::vector<Seg> segs; double E = 0.0, Inorm = 0.0; long nev = 0; bool done = false; };
    std::vector<Item> items((size_t)N * nq);
    int next_id = 0;
    auto grow_pool = [&](long need) -> int {
        if (need <= h->mq_pool_cap) return HIPADJ_OK;
        long cap = h->mq_pool_cap; while (cap < need) cap *= 2;
        double* np_ = nullptr;
        HIP_TRY(h, hipMalloc((void**)&np_, sizeof(double) * (size_t)cap * NPAR));
        HIP_TRY(h, hipMemcpyAsync(np_, h->d_mq_pool, sizeof(double) * (size_t)next_id * NPAR, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        (void)hipFree(h->d_mq_pool);
        h->ws_bytes += (double)((cap - h->mq_pool_cap) * (long)NPAR * 8);
        h->d_mq_pool = np_; h->mq_pool_cap = cap;
        return HIPADJ_OK;
    };
    // evaluate panels [a, b] of trajectory tr: returns per request (||K||^2, ||K - G||^2) and the pool id of K
    struct Req { int tr; double a, b; int idK; double nK2, nE2; };
    auto eval = [&](std::vector<Req>& R) -> int {
        TRY(grow_pool((long)next_id + 2 * (long)R.size()));
        for (auto& r : R) { r.idK = next_id; next_id += 2; }
        const size_t per = (size_t)h->mq_chunk / 2;          // requests per launch (two entries each)
        std::vector<MlpPanel> pan; std::vector<int> ids; std::vector<double> nrm;
        for (size_t b0 = 0; b0 < R.size(); b0 += per) {
            const size_t cnt = std::min(per, R.size() - b0);
            pan.clear(); ids.clear();
            for (size_t q = 0; q < cnt; ++q) { const Req& r = R[b0 + q]; pan.push_back(MlpPanel{r.tr, 0, r.a, r.b}); pan.push_back(MlpPanel{r.tr, 1, r.a, r.b}); }
            for (size_t q = 0; q < cnt; ++q) ids.push_back(R[b0 + q].idK);
            for (size_t q = 0; q < cnt; ++q) ids.push_back(R[b0 + q].idK + 1);
            HIP_TRY(h, hipMemcpyAsync(h->d_mq_panels, pan.data(), sizeof(MlpPanel) * pan.size(), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->d_mq_ids, ids.data(), sizeof(int) * ids.size(), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL((k_mlp_quad_panel<H>), dim3((unsigned)nWG, (unsigned)pan.size()), dim3(MlpG<H>::NT), 0, h->stream, h->mg, p, (const double*)h->d_fknots,
                               (const double*)h->d_fadj, (const MlpPanel*)h->d_mq_panels, h->d_c1);
            HIP_TRY(h, hipGetLastError());
            // the entries of this launch occupy consecutive pool slots idK(first request) ..: (K, G) pairs
            hipLaunchKernelGGL(k_mlp_grad_reduce, dim3((NPAR + 255) / 256, (unsigned)pan.size()), dim3(256), 0, h->stream, (int)NPAR, (long)nWG, (const double*)h->d_c1,
                               h->d_mq_pool + (size_t)R[b0].idK * NPAR);
            HIP_TRY(h, hipGetLastError());
            hipLaunchKernelGGL(k_mlp_quad_norm, dim3((unsigned)cnt), dim3(256), 0, h->stream, (int)NPAR, (const double*)h->d_mq_pool, (const int*)h->d_mq_ids, (const int*)h->d_mq_ids + cnt, h->d_mq_norm);
            HIP_TRY(h, hipGetLastError());
            nrm.resize(2 * cnt);
            HIP_TRY(h, hipMemcpyAsync(nrm.data(), h->d_mq_norm, sizeof(double) * 2 * cnt, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));    // also orders the reuse of d_mq_panels / d_mq_ids / d_c1 by the next launch
            for (size_t q = 0; q < cnt; ++q) { R[b0 + q].nK2 = nrm[2 * q]; R[b0 + q].nE2 = nrm[2 * q + 1]; }
        }
        return HIPADJ_OK;
    };
    // || sum of the K vectors of the listed items ||
    auto total_norms = [&](const std::vector<size_t>& which) -> int {
        const size_t per = 1024;
        for (size_t b0 = 0; b0 < which.size(); b0 += per) {
            const size_t cnt = std::min(per, which.size() - b0);
            std::vector<int> ids, start(1, 0);
            for (size_t q = 0; q < cnt; ++q) { for (const Seg& sg : items[which[b0 + q]].segs) ids.push_back(sg.idK); start.push_back((int)ids.size()); }
            if (ids.size() + start.size() > ((size_t)1 << 16)) { HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "QuadratureAdjoint (MLP family): the segment lists outgrew the id buffer; loosen the quadrature tolerances"); }
            std::vector<int> both(ids); both.insert(both.end(), start.begin(), start.end());
            HIP_TRY(h, hipMemcpyAsync(h->d_mq_ids, both.data(), sizeof(int) * both.size(), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_mlp_quad_sum, dim3((unsigned)cnt), dim3(256), 0, h->stream, (int)NPAR, (const double*)h->d_mq_pool, (const int*)h->d_mq_ids, (const int*)h->d_mq_ids + ids.size(),
                               (double*)nullptr, h->d_mq_norm);
            HIP_TRY(h, hipGetLastError());
            std::vector<double> n2(cnt);
            HIP_TRY(h, hipMemcpyAsync(n2.data(), h->d_mq_norm, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            for (size_t q = 0; q < cnt; ++q) items[which[b0 + q]].Inorm = std::sqrt(n2[q]);
        }
        return HIPADJ_OK;
    };
    // round 0: the whole interval
    {
        std::vector<Req> R;
        for (long tr = 0; tr < N; ++tr) for (int qi = 0; qi < nq; ++qi) R.push_back(Req{(int)tr, h->qa_host[qi], h->qb_host[qi], 0, 0.0, 0.0});
        TRY(eval(R));
        for (size_t q = 0; q < R.size(); ++q) { Item& it = items[q]; it.segs.push_back(Seg{R[q].a, R[q].b, std::sqrt(R[q].nE2), R[q].idK}); it.E = it.segs[0].E; it.Inorm = std::sqrt(R[q].nK2); it.nev = 15; }
    }
    const long maxevals = 10000000L;
    for (int round = 0; round < 4096; ++round) {
        std::vector<size_t> act; std::vector<int> worst; std::vector<Req> R;
        for (size_t q = 0; q < items.size(); ++q) {
            Item& it = items[q];
            if (it.done) continue;
            if (it.E <= std::fmax(atol, rtol * it.Inorm) || it.nev >= maxevals) { it.done = true; continue; }
            int wi = 0; for (int sidx = 1; sidx < (int)it.segs.size(); ++sidx) if (it.segs[sidx].E > it.segs[wi].E) wi = sidx;
            const Seg sw = it.segs[wi]; const double mid = 0.5 * (sw.a + sw.b);
            if (!(mid > std::fmin(sw.a, sw.b) && mid < std::fmax(sw.a, sw.b))) { it.done = true; continue; }      // cannot be split further
            act.push_back(q); worst.push_back(wi);
            const int tr = (int)(q / (size_t)nq);
            R.push_back(Req{tr, sw.a, mid, 0, 0.0, 0.0}); R.push_back(Req{tr, mid, sw.b, 0, 0.0, 0.0});
        }
        if (act.empty()) break;
        TRY(eval(R));
        for (size_t q = 0; q < act.size(); ++q) {
            Item& it = items[act[q]]; const Seg sw = it.segs[worst[q]];
            const Seg s1{R[2 * q].a, R[2 * q].b, std::sqrt(R[2 * q].nE2), R[2 * q].idK}, s2{R[2 * q + 1].a, R[2 * q + 1].b, std::sqrt(R[2 * q + 1].nE2), R[2 * q + 1].idK};
            it.E += s1.E + s2.E - sw.E; it.nev += 30;
            it.segs[worst[q]] = s1; it.segs.push_back(s2);
        }
        TRY(total_norms(act));
    }
    // dp = sum of the K vectors of all accepted segments (QuadGK's final re-sum), per parameter group
    {
        const long groups = h->cfg.p_shared ? 1 : N;
        std::vector<int> ids, start(1, 0);
        for (long gidx = 0; gidx < groups; ++gidx) {
            const long t0 = h->cfg.p_shared ? 0 : gidx, t1 = h->cfg.p_shared ? N : gidx + 1;
            for (long tr = t0; tr < t1; ++tr) for (int qi = 0; qi < nq; ++qi) for (const Seg& sg : items[(size_t)tr * nq + qi].segs) ids.push_back(sg.idK);
            start.push_back((int)ids.size());
        }
        if (ids.size() + start.size() > ((size_t)1 << 16)) { HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "QuadratureAdjoint (MLP family): the segment lists outgrew the id buffer; loosen the quadrature tolerances"); }
        std::vector<int> both(ids); both.insert(both.end(), start.begin(), start.end());
        HIP_TRY(h, hipMemcpyAsync(h->d_mq_ids, both.data(), sizeof(int) * both.size(), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_mlp_quad_sum, dim3((unsigned)groups, (unsigned)((NPAR + 255) / 256)), dim3(256), 0, h->stream, (int)NPAR, (const double*)h->d_mq_pool, (const int*)h->d_mq_ids, (const int*)h->d_mq_ids + ids.size(), d_dp, (double*)nullptr);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));        // `both` must outlive the copy
    }
    if (std::getenv("HIPADJ_MQ_DEBUG")) {                   // debugging hook: how much adaptivity a run needed
        long segs = 0; for (const Item& it : items) segs += (long)it.segs.size();
        std::fprintf(stderr, "hipadj: MLP quadrature: %zu intervals, %ld accepted segments, %d panel vectors evaluated\n", items.size(), segs, next_id);
    }
    return HIPADJ_OK;
}

template <int H> int mlp_adjoint_launch(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const double* p = h->p_dev_last;
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    // in-register parameter gradient: the sweep leaves one partial gradient per workgroup, a fixed-order sum finishes dp
    const dim3 grid((unsigned)(h->mg.B / 16), (unsigned)h->N), gblk(MlpG<H>::NT);
    const int* ck = nullptr; double* noadj = nullptr;
    if (h->cfg.alg == HIPADJ_ALG_GAUSS)
        hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 2>), grid, gblk, 0, h->stream, h->mg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, ck, h->d_c1, noadj, d_du0, h->d_flag);
    else if (h->cfg.alg == HIPADJ_ALG_BACKSOLVE)
        hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 1>), grid, gblk, 0, h->stream, h->mg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev,
                           h->cfg.checkpointing ? (const int*)h->d_ckpt_of_knot : ck, h->d_c1, noadj, d_du0, h->d_flag);
    else if (h->cfg.alg == HIPADJ_ALG_QUADRATURE) {
        hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 3>), grid, gblk, 0, h->stream, h->mg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, ck, (double*)nullptr, h->d_fadj, d_du0, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(es.k1, h->stream));
        TRY(mlp_quadrature<H>(h, p, d_dp));
        HIP_TRY(h, hipEventRecord(es.a1, h->stream));
        es.pending = true;
        return HIPADJ_OK;
    }
    else
        hipLaunchKernelGGL((k_mlp_adjoint_grad<H, 0>), grid, gblk, 0, h->stream, h->mg, p, (const double*)h->d_fknots, d_cot, (const int*)h->d_save_rev, ck, h->d_c1, noadj, d_du0, h->d_flag);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    const long groups = h->cfg.p_shared ? 1 : h->N;
    const long per_group = (h->N * (long)(h->mg.B / 16)) / groups;
    hipLaunchKernelGGL(k_mlp_grad_reduce, dim3((Mlp<H>::NPAR + 255) / 256, (unsigned)groups), dim3(256), 0, h->stream, (int)Mlp<H>::NPAR, per_group, (const double*)h->d_c1, d_dp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = true;
    return HIPADJ_OK;
}

template <class Mo> int adaptive_forward(hipadj_handle* h, const double* d_u0, const double* d_p, double* d_out) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const bool sized = h->auto_steps && h->cfg.alg != HIPADJ_ALG_BACKSOLVE;
    if (sized && h->ip_ckpt) h->ag.SmaxI = (int)h->rec_cap;
    for (int pass = 0; pass < 2; ++pass) {
        if (h->cfg.stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE)      // the stiff stepper: same records, ros23_integrate (hipadj_adaptive.hpp)
            hipLaunchKernelGGL((k_forward_tsit5<Mo, 1>), dim3(waves), dim3(WAVE), 0, h->stream, h->ag, d_u0, d_p, h->d_rec, h->d_nsteps,
                               (const double*)h->d_save_t, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag);
        else if (h->quad_fwd && QuadForm<Mo>::value)      // four lanes per trajectory (hipadj_quad_ts5.hpp): same records, a third of the instructions per wave
            hipLaunchKernelGGL((k_forward_tsit5_quad<Mo>), dim3((unsigned)((h->N + 15) / 16)), dim3(WAVE), 0, h->stream, h->ag, d_u0, d_p, h->ip_ckpt ? (double*)nullptr : h->d_rec, h->d_nsteps,
                               (const double*)h->d_save_t, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag);
        else
        hipLaunchKernelGGL((k_forward_tsit5<Mo>), dim3(waves), dim3(WAVE), 0, h->stream, h->ag, d_u0, d_p, h->ip_ckpt ? (double*)nullptr : h->d_rec, h->d_nsteps,
                           (const double*)h->d_save_t, (d_out && h->M > 0) ? h->d_outT : (double*)nullptr, (const double*)h->d_ck_t, h->d_ckpt, h->d_yT, h->d_flag);
        HIP_TRY(h, hipGetLastError());
        if (!sized) break;
        const int again = adaptive_autosize(h);
        if (again < 0) return again;
        if (again == 0) break;
    }
    if (d_out && h->M > 0) TRY(launch_transpose_to_aos(h, h->d_outT, d_out, h->M * h->n));
    return HIPADJ_OK;
}
template <class Mo, int ALG, int CC, bool CK = false, int STEP = 0> int adaptive_adjoint_l(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    const unsigned waves = (unsigned)(h->Npad / WAVE);
    const unsigned fblocks = (unsigned)((h->N + FIN - 1) / FIN);
    double* dp_rows = h->cfg.p_shared ? (double*)nullptr : d_dp;
    // the column the sweep streams at the loss times: the caller's block already in the streaming layout (hipadj_adjoint_dev_soa), or the handle's own buffer — cotangents
    // transposed here per pass, or the data block of a device-resident loss transposed ONCE by hipadj_set_loss_data
    const double* cotT = h->cot_soa ? h->cot_soa : h->d_cotT;
    if (h->cfg.loss_kind == HIPADJ_LOSS_COTANGENT && h->M > 0 && !h->cot_soa) TRY(launch_transpose_to_soa(h, d_cot, h->d_cotT, h->M * h->n));
    hipadj_handle::EvSet& es = h->evs[h->ev_next];
    h->ev_next = (h->ev_next + 1) % hipadj_handle::NSET;
    harvest_set(h, es, true);
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a0, h->stream));
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k0, h->stream));
    for (int pass = 0; pass < 2; ++pass) {
        constexpr bool QUAD_OK = QuadAdj<Mo>::value && CC == 0 && !CK && STEP == 0 && (ALG == 0 || ALG == 1 || ALG == 2);
        if (QUAD_OK && h->quad_adj) {
            if constexpr (QUAD_OK)
                hipLaunchKernelGGL((k_adjoint_tsit5_quad<Mo, ALG>), dim3((unsigned)((h->N + 15) / 16)), dim3(WAVE), 0, h->stream, h->ag, h->p_dev_last, (const double*)h->d_rec,
                                   (const int*)h->d_nsteps, (const double*)h->d_yT, (const double*)h->d_ckpt, (const double*)h->d_ck_t, (const double*)h->d_save_t,
                                   (const double*)h->d_tstops, h->ntstops, cotT, d_du0, h->d_dp_traj, h->d_flag);
        } else
        hipLaunchKernelGGL((k_adjoint_tsit5<Mo, ALG, CC, CK, STEP>), dim3(waves), dim3(WAVE), 0, h->stream, h->ag, h->p_dev_last, (const double*)h->d_rec,
                           (const int*)h->d_nsteps, (const double*)h->d_yT, (const double*)h->d_ckpt, (const double*)h->d_ck_t, (const double*)h->d_save_t,
                           (const double*)h->d_tstops, h->ntstops, cotT, d_du0, h->d_dp_traj, h->d_flag,
                           h->d_arec, h->d_nsteps_adj, h->SmaxA);
        HIP_TRY(h, hipGetLastError());
        if (!(ALG == 3 && h->auto_steps)) break;
        const int again = adaptive_adjoint_autosize(h);   // dense adjoint record too small for some trajectory: regrown, sweep repeated once
        if (again < 0) return again;
        if (again == 0) break;
    }
    if (h->timing >= 1) HIP_TRY(h, hipEventRecord(es.k1, h->stream));
    if constexpr (ALG == 3) {
        const double atol = h->cfg.quad_abstol > 0 ? h->cfg.quad_abstol : 1e-6, rtol = h->cfg.quad_reltol > 0 ? h->cfg.quad_reltol : 1e-3;
        hipLaunchKernelGGL((k_quad_gk_tsit5<Mo, CC>), dim3(waves, (unsigned)h->nq), dim3(WAVE), 0, h->stream, h->ag, h->p_dev_last, (const double*)h->d_rec,
                           (const int*)h->d_nsteps, (const double*)h->d_arec, (const int*)h->d_nsteps_adj, (const double*)h->d_qa, (const double*)h->d_qb, atol, rtol, h->d_qres);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(k_quad_sum, dim3(waves), dim3(WAVE), 0, h->stream, h->N, h->Npad, h->np, h->nq, (const double*)h->d_qres, h->d_dp_traj);
        HIP_TRY(h, hipGetLastError());
    }
    hipLaunchKernelGGL((k_finish<Mo::N, Mo::NP>), dim3(fblocks), dim3(FIN), 0, h->stream, h->N, h->Npad, (const double*)d_du0,
                       (const double*)h->d_dp_traj, dp_rows, h->d_partial, h->d_flag, h->d_ticket, (double*)nullptr);
    HIP_TRY(h, hipGetLastError());
    if (h->cfg.p_shared) {
        hipLaunchKernelGGL(k_reduce_final, dim3((unsigned)h->np), dim3(FIN), 0, h->stream, (int)fblocks, h->np, (const double*)h->d_partial, d_dp);
        HIP_TRY(h, hipGetLastError());
    }
    if (h->timing >= 2) HIP_TRY(h, hipEventRecord(es.a1, h->stream));
    es.pending = h->timing >= 1; es.full = h->timing >= 2;
    return HIPADJ_OK;
}
template <class Mo> int adaptive_adjoint(hipadj_handle* h, const double* d_cot, double* d_du0, double* d_dp) {
    if (h->cfg.stepper == HIPADJ_STEPPER_ROSENBROCK23_ADAPTIVE) {   // the stiff stepper: the same table with STEP = 1 (a DAE model: the planner admitted no cost and no Backsolve)
#define HIPADJ_ROS_CASE(A, C, K) case A * 4 + C: if constexpr (!model_dae<Mo>::value || (C == 0 && A != 1)) return adaptive_adjoint_l<Mo, A, C, K, 1>(h, d_cot, d_du0, d_dp); else break;
        if (h->ip_ckpt) switch (h->cfg.alg * 4 + h->cfg.cont_cost) {      // checkpointing = true: the intervals re-solved with Rosenbrock23 inside the sweep
        HIPADJ_ROS_CASE(0, 0, true) HIPADJ_ROS_CASE(0, 1, true) HIPADJ_ROS_CASE(0, 2, true)
        HIPADJ_ROS_CASE(2, 0, true) HIPADJ_ROS_CASE(2, 1, true) HIPADJ_ROS_CASE(2, 2, true)
        HIPADJ_ROS_CASE(4, 0, true) HIPADJ_ROS_CASE(4, 1, true) HIPADJ_ROS_CASE(4, 2, true)
        default: break;
        }
        else switch (h->cfg.alg * 4 + h->cfg.cont_cost) {
        HIPADJ_ROS_CASE(0, 0, false) HIPADJ_ROS_CASE(0, 1, false) HIPADJ_ROS_CASE(0, 2, false)
        HIPADJ_ROS_CASE(1, 0, false) HIPADJ_ROS_CASE(1, 1, false) HIPADJ_ROS_CASE(1, 2, false)
        HIPADJ_ROS_CASE(2, 0, false) HIPADJ_ROS_CASE(2, 1, false) HIPADJ_ROS_CASE(2, 2, false)
        HIPADJ_ROS_CASE(3, 0, false) HIPADJ_ROS_CASE(3, 1, false) HIPADJ_ROS_CASE(3, 2, false)
        HIPADJ_ROS_CASE(4, 0, false) HIPADJ_ROS_CASE(4, 1, false) HIPADJ_ROS_CASE(4, 2, false)
        default: break;
        }
#undef HIPADJ_ROS_CASE
        HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "Rosenbrock23: sensealg %d / cont_cost %d%s has no device kernel", h->cfg.alg, h->cfg.cont_cost, h->ip_ckpt ? " (checkpointed)" : "");
    }
    if (h->ip_ckpt) {   // checkpointing=true for Interpolating / Gauss: per-interval re-solve inside the sweep
        switch (h->cfg.alg * 4 + h->cfg.cont_cost) {
        case HIPADJ_ALG_INTERPOLATING * 4 + 0: return adaptive_adjoint_l<Mo, 0, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_INTERPOLATING * 4 + 1: return adaptive_adjoint_l<Mo, 0, 1, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_INTERPOLATING * 4 + 2: return adaptive_adjoint_l<Mo, 0, 2, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS * 4 + 0: return adaptive_adjoint_l<Mo, 2, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS * 4 + 1: return adaptive_adjoint_l<Mo, 2, 1, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS * 4 + 2: return adaptive_adjoint_l<Mo, 2, 2, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return adaptive_adjoint_l<Mo, 4, 0, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return adaptive_adjoint_l<Mo, 4, 1, true>(h, d_cot, d_du0, d_dp);
        case HIPADJ_ALG_GAUSS_KRONROD * 4 + 2: return adaptive_adjoint_l<Mo, 4, 2, true>(h, d_cot, d_du0, d_dp);
        default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg %d / cont_cost %d has no checkpointed adaptive device kernel", h->cfg.alg, h->cfg.cont_cost);
        }
    }
    switch (h->cfg.alg * 4 + h->cfg.cont_cost) {
    case HIPADJ_ALG_INTERPOLATING * 4 + 0: return adaptive_adjoint_l<Mo, 0, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_INTERPOLATING * 4 + 1: return adaptive_adjoint_l<Mo, 0, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_INTERPOLATING * 4 + 2: return adaptive_adjoint_l<Mo, 0, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 0: return adaptive_adjoint_l<Mo, 1, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 1: return adaptive_adjoint_l<Mo, 1, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_BACKSOLVE * 4 + 2: return adaptive_adjoint_l<Mo, 1, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS * 4 + 0: return adaptive_adjoint_l<Mo, 2, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS * 4 + 1: return adaptive_adjoint_l<Mo, 2, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS * 4 + 2: return adaptive_adjoint_l<Mo, 2, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 0: return adaptive_adjoint_l<Mo, 3, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 1: return adaptive_adjoint_l<Mo, 3, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_QUADRATURE * 4 + 2: return adaptive_adjoint_l<Mo, 3, 2>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 0: return adaptive_adjoint_l<Mo, 4, 0>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 1: return adaptive_adjoint_l<Mo, 4, 1>(h, d_cot, d_du0, d_dp);
    case HIPADJ_ALG_GAUSS_KRONROD * 4 + 2: return adaptive_adjoint_l<Mo, 4, 2>(h, d_cot, d_du0, d_dp);
    default: HIPADJ_FAIL(h, HIPADJ_ERR_UNSUPPORTED, "sensealg %d / cont_cost %d has no adaptive device kernel", h->cfg.alg, h->cfg.cont_cost);
    }
}
