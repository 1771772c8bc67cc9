// nasr_pass.hip — one step on the handle's streams: forward pass (dense stages, hoisted input GEMMs, recurrence, projection),
// CTC, backward pass (projection, BPTT, input / weight gradients, gradient buckets).  The graph of create_network
// (networks/bilstm_ctc_net.py:10-52, networks/lstm_ctc_net.py:10-47, networks/deepspeech.py:43-121) and its gradients
// (networks/tfnetwork.py:115-140) as launches; see DESIGN.md §4.
#include "nasr_ctx.h"

using namespace nasr;
using namespace nasr_impl;

namespace nasr_impl {

// ---- the per-timestep loops over steps [s0, s1), optionally replayed from a hipGraph -----------
int run_steps(nasr_ctx* h, int l, bool bwd, int s0, int s1, hipStream_t st) {
  const LstmDims dm{h->T, h->B, h->Bp, h->H, h->Hp, h->D};
  if (!bwd && h->wide && s0 == 0 && s1 == h->T && wide_supported(h->Hp, h->Bp)) {
    for (int d = 0; d < h->D; ++d) {
      const size_t k = (size_t)l * h->D + d;
      launch_lstm_wide_fwd(dm, d, h->Uw + k * wide_image_bytes(h->Hp), h->Ucinv + k * h->N4, h->gates[l].as<float>(),
                           h->cbuf[l].as<float>(), h->outb[l].as<float>(), h->seq_p, h->whx, h->wpart, h->wctl, h->perr,
                           h->Gbase, h->cfg.forget_bias, st);
    }
    h->persist_used = true;
    HIPCHK(h, hipGetLastError());
    return NASR_OK;
  }
  if (bwd && h->wide && s0 == 0 && s1 == h->T && wide_supported(h->Hp, h->Bp)) {
    launch_wide_row_scales(dm, dout_of(h, l), h->seq_p, h->wsrow, st);
    for (int d = 0; d < h->D; ++d) {
      const size_t k = (size_t)l * h->D + d;
      launch_lstm_wide_bwd(dm, d, h->Uwb + k * wide_image_bytes(h->Hp), h->Urinv + k * h->Hp, h->wsrow,
                           h->gates[l].as<float>(), dg_of(h, l), h->cbuf[l].as<float>(), dout_of(h, l), h->seq_p, h->wpart,
                           h->wpx, h->wctl, h->perr, h->Gbase, st);
    }
    h->persist_used = true;
    HIPCHK(h, hipGetLastError());
    return NASR_OK;
  }
  if (h->persist && s0 == 0 && s1 == h->T) {
    const size_t k = (size_t)l * h->D;
    if (!bwd)
      launch_lstm_persist_fwd(dm, h->Upf + k * h->imf, h->rec_f16 ? h->Ucinv + k * h->N4 : nullptr,
                              h->gates[l].as<float>(), h->cbuf[l].as<float>(),
                              h->outb[l].as<float>(), h->seq_p, h->xchf + (size_t)l * (persist_hx_bytes(h->Hp) / 4),
                              h->pctl + 1 + l, h->perr, h->Gbase, h->cfg.forget_bias, st, true);
    else
    {
      launch_lstm_persist_bwd(dm, h->Upb + k * h->imb, h->gates[l].as<float>(), dg_of(h, l), h->cbuf[l].as<float>(),
                              dout_of(h, l), h->seq_p, h->xchb + (size_t)l * (persist_px_bytes() / 4), h->pctl + 1 + h->L + l,
                              h->perr, h->Gbase, st, true, h->dgmax.as<float>(),
                              h->dgmax.as<float>() + (size_t)h->D * 32 * h->T * h->Bp);
      h->dgmax_layer = l;
    }
    h->persist_used = true;
    HIPCHK(h, hipGetLastError());
    return NASR_OK;
  }
  const size_t sU = (size_t)l * h->D * h->Hp * h->N4;
  const size_t hs = (size_t)h->D * h->Bp * h->Hp;   // one h-state image
  const size_t ps = (size_t)h->D * lstm_bwd_partials(h->Hp) * h->Bp * h->Hp;   // one partial-sum image
  float* hst = h->hstate.as<float>();
  float* par = h->partial.as<float>();
  float* dcs = h->dcstate.as<float>();
  auto body = [&]() {
    if (!bwd) {
      if (s0 == 0) (void)hipMemsetAsync(hst, 0, hs * 4, st);
      for (int s = s0; s < s1; ++s)
        launch_lstm_fwd_step(dm, s, h->Uf + sU, hst + (s & 1) * hs, hst + ((s + 1) & 1) * hs,
                             h->gates[l].as<float>(), h->cbuf[l].as<float>(), h->outb[l].as<float>(),
                             h->seq_p, h->cfg.forget_bias, st);
    } else {
      if (s1 == h->T) {
        (void)hipMemsetAsync(par, 0, ps * 4, st);
        (void)hipMemsetAsync(dcs, 0, hs * 4, st);
      }
      for (int s = s1 - 1; s >= s0; --s) {
        const int k = h->T - 1 - s;
        launch_lstm_bwd_step(dm, s, h->Ub + sU, par + (k & 1) * ps, par + ((k + 1) & 1) * ps,
                             h->gates[l].as<float>(), dg_of(h, l), h->cbuf[l].as<float>(), dout_of(h, l),
                             dcs + (k & 1) * hs, dcs + ((k + 1) & 1) * hs, h->seq_p, st);
      }
    }
  };
  if (!h->graph_mode) {
    body();
    HIPCHK(h, hipGetLastError());
    return NASR_OK;
  }
  const GraphKey key{h->T, l, bwd ? 1 : 0, s0 * 4096 + (s1 - s0)};
  auto it = h->graphs.find(key);
  if (it == h->graphs.end()) {
    if (h->graphs.size() > 256) drop_graphs(h);
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    body();
    HIPCHK(h, hipStreamEndCapture(st, &g));
    hipGraphExec_t ex = nullptr;
    HIPCHK(h, hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = h->graphs.emplace(key, ex).first;
  }
  HIPCHK(h, hipGraphLaunch(it->second, st));
  return NASR_OK;
}

float* ensure_slabs(nasr_ctx* h, int split, int M, int N) {
  if (split <= 1) return nullptr;
  bool grew = false;
  if (!h->slabs.ensure((size_t)split * M * N * 4, &grew)) return nullptr;
  return h->slabs.as<float>();
}

// operand scales of layer l's dG (rows = frames: sc_gr, optional; columns = gates: sc_gc): from the maxima the persistent
// BPTT kernel took while it stored dG, or by a pass over dG
void dg_scales(nasr_ctx* h, int l, int R, bool rows, hipStream_t st) {
  const int DN = h->D * h->N4;
  if (h->dgmax_layer == l) {
    const float* rp = h->dgmax.as<float>();
    launch_tph_scales_from_parts(rp, h->D * 32, R, rows ? h->sc_gr.sp() : nullptr, rows ? h->sc_gr.ip() : nullptr,
                                 rp + (size_t)h->D * 32 * R, 8 / h->D, DN, gc_of(h, l).sp(), gc_of(h, l).ip(), st);
  } else {
    pl_scales(h, dg_of(h, l), R, DN, DN, rows ? &h->sc_gr : nullptr, &gc_of(h, l), st);
  }
}

// gates_l = X_l * Wx_l + bias_l over all R rows - or, for a ragged batch (h->cmp_rows, nasr_ctx.h), over its frames only:
// the split pass gathers them, the epilogue scatters the result rows back (padding rows of gates_l keep what they had; the
// recurrence kernels select by t < seq_len[b] and never use them)
void gemm_xproj(nasr_ctx* h, int l, int R, hipStream_t st) {
  const int D = h->D, N4 = h->N4, Ip = h->Ip[l];
  const int rows = h->cmp_rows ? h->cmp_rows : R;
  const int* vm = h->cmp_rows ? h->vrow_p : nullptr;
  ActScale as = lstm_in_scale(h, l);
  if (vm && l == 0) { as.rs = h->sc_cx.sp(); as.rinv = h->sc_cx.ip(); }   // the feature rows' scales in the compacted order
  // a training step wants the layer below's output a second time, with the frame index as contraction index (its own
  // recurrent weight gradient and this layer's input weight gradient): both plane sets in this one pass over it
  const bool both = l > 0 && h->cur && h->cur->has_labels;
  launch_tph_split2(lstm_input(h, l), h->XTP.as<unsigned char>(), both ? h->OTT[l - 1].as<unsigned char>() : nullptr, rows, Ip, Ip,
                    as.rs, 1.f, both ? as.cs : nullptr, 1.f, nullptr, st, vm);
  if (l > 0) h->ott_valid[l - 1] = both;
  GemmTPHDesc g{};
  g.A = h->XTP.as<unsigned char>(); g.B = h->WfTP + h->off_wftp[l]; g.C = h->gates[l].as<float>();
  g.M = rows; g.N = D * N4; g.K = Ip; g.nkbA = (Ip + 15) / 16; g.nkbB = g.nkbA; g.ldc = D * N4;
  g.bias = h->P + h->off_bias[l]; g.split_k = 1;
  g.c_map = vm;
  pl_gemm(g, as.rinv, h->sc_wc[l].ip(), st);
}

// dOut_{l-1} = dG_l * Wx_l^T : the gradient wrt layer l's input = the layer below's output.  weight_grads(l) follows:
// dG is split ONCE into both plane sets (frame-row scales for this product, gate-column scales for the weight gradients)
// and its 64-row partial column sums (the bias gradient).  (Ragged batch: over the compacted rows, as gemm_xproj.)
void gemm_dx(nasr_ctx* h, int l, int R, hipStream_t st) {
  const int D = h->D, N4 = h->N4;
  const int rows = h->cmp_rows ? h->cmp_rows : R;
  const int* vm = h->cmp_rows ? h->vrow_p : nullptr;
  const float* A = dg_of(h, l);
  float* C = l > 0 ? dout_of(h, l - 1) : h->dYbuf[h->npre - 1].as<float>();
  dg_scales(h, l, R, true, st);
  const float *rs = h->sc_gr.sp(), *rinv = h->sc_gr.ip();
  if (vm) {
    launch_gather_rows(h->sc_cr.sp(), h->sc_gr.sp(), vm, h->cmp_rows_p, 1.f, st);
    launch_gather_rows(h->sc_cr.ip(), h->sc_gr.ip(), vm, h->cmp_rows_p, 1.f, st);
    rs = h->sc_cr.sp(); rinv = h->sc_cr.ip();
  }
  launch_tph_split2(A, h->GTP.as<unsigned char>(), gttp_of(h, l), rows, D * N4, D * N4, rs, 1.f, gc_of(h, l).sp(), 1.f,
                    csws_of(h, l), st, vm);
  h->gttp_layer = l;   // weight_grads(l): transposed planes and column-sum partials of dG are there
  GemmTPHDesc g{};
  g.A = h->GTP.as<unsigned char>(); g.B = h->WbTP + h->off_wbtp[l]; g.C = C;
  g.M = rows; g.N = h->Ip[l]; g.K = D * N4; g.nkbA = (D * N4 + 15) / 16; g.nkbB = g.nkbA; g.ldc = h->Ip[l];
  g.split_k = gemm_tph_pick_split(g.M, g.N, g.K);
  g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
  if (g.split_k > 1 && !g.slabs) g.split_k = 1;
  g.c_map = vm;
  pl_gemm(g, rinv, h->sc_wr[l].ip(), st);
}

// ---- dense stages (networks/deepspeech.py:43-68,106-113) -----------------------------------------------------
// Y_i = dropout(min(relu(X W_i + b_i), clip)): one tiled-plane GEMM + the in-place epilogue of dense.hip
int dense_forward(nasr_ctx* h, int i, const float* X) {
  const int R = h->T * h->Bp, Ip = h->dIp[i], Wp = h->dWp[i];
  const ActScale as = dense_in_scale(h, i);
  pl_split(X, h->XTP.as<unsigned char>(), nullptr, R, Ip, Ip, as.rs, nullptr, nullptr, h->st);
  GemmTPHDesc g{};
  g.A = h->XTP.as<unsigned char>(); g.B = h->DfTP + h->off_dftp[i]; g.C = h->Ybuf[i].as<float>();
  g.M = R; g.N = Wp; g.K = Ip; g.nkbA = (Ip + 15) / 16; g.nkbB = g.nkbA; g.ldc = Wp;
  g.bias = h->P + h->off_db[i]; g.split_k = 1;
  pl_gemm(g, as.rinv, h->sc_dc[i].ip(), h->st);
  launch_dense_act(h->Ybuf[i].as<float>(), R, h->Bp, h->B, h->dWid[i], Wp, h->cfg.relu_clip, h->cfg.dropout[i],
                   h->drop_seed, h->drop_counter, i, h->st);
  // the stage's output feeds the next GEMM (rows = frames) and, transposed, its weight gradient (rows = features)
  pl_scales(h, h->Ybuf[i].as<float>(), R, Wp, Wp, &h->sc_yr[i], &h->sc_yc[i], h->st);
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

// dY_i (in dYbuf[i]) -> dW_i, db_i and, when dX is given, the gradient wrt the stage's input [R][dIp]
int dense_backward(nasr_ctx* h, int i, const float* X, float* dX) {
  const int R = h->T * h->Bp, Ip = h->dIp[i], Wp = h->dWp[i];
  const int nkb = (R + 15) / 16;
  float* dZ = h->dYbuf[i].as<float>();
  launch_dense_act_bwd(dZ, h->Ybuf[i].as<float>(), (int64_t)R * Wp, h->cfg.relu_clip, h->cfg.dropout[i], h->st);
  const ActScale as = dense_in_scale(h, i);
  pl_scales(h, dZ, R, Wp, Wp, dX ? &h->sc_gr : nullptr, &h->sc_gc, h->st);
  // both forms of dZ (the first only when an input gradient follows) + column-sum partials in one pass
  pl_split(dZ, dX ? h->GTP.as<unsigned char>() : nullptr, h->GTTP.as<unsigned char>(), R, Wp, Wp, h->sc_gr.sp(),
           h->sc_gc.sp(), h->csws.as<float>(), h->st);
  pl_split(X, nullptr, h->DTP.as<unsigned char>(), R, Ip, Ip, nullptr, as.cs, nullptr, h->st);
  {  // dW = X^T dZ
    GemmTPHDesc g{};
    g.A = h->DTP.as<unsigned char>(); g.B = h->GTTP.as<unsigned char>(); g.C = h->G + h->off_dw[i];
    g.M = Ip; g.N = Wp; g.K = R; g.nkbA = nkb; g.nkbB = nkb; g.ldc = Wp;
    g.split_k = gemm_tph_pick_split(g.M, g.N, g.K);
    g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) return h->fail(NASR_ERR_HIP, "slab workspace allocation failed");
    pl_gemm(g, as.cinv, h->sc_gc.ip(), h->st);
  }
  launch_colsum_parts(h->csws.as<float>(), tp_split2_parts(R), Wp, h->G + h->off_db[i], h->st);
  if (dX) {  // dX = dZ W^T
    GemmTPHDesc g{};
    g.A = h->GTP.as<unsigned char>(); g.B = h->DbTP + h->off_dbtp[i]; g.C = dX;
    g.M = R; g.N = Ip; g.K = Wp; g.nkbA = (Wp + 15) / 16; g.nkbB = g.nkbA; g.ldc = Ip;
    g.split_k = gemm_tph_pick_split(g.M, g.N, g.K);
    g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) g.split_k = 1;
    pl_gemm(g, h->sc_gr.ip(), h->sc_dr[i].ip(), h->st);
  }
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

int forward(nasr_ctx* h) {
  if (!h->resident) return h->fail(NASR_ERR_STATE, "no resident batch: call nasr_upload_batch first");
  const int Bp = h->Bp, T = h->T, D = h->D, Hp = h->Hp;
  const int R = T * Bp;
  h->n_fwd_launch = 0;
  std::fill(h->ott_valid.begin(), h->ott_valid.end(), 0);
  // the fault word of the pass that starts here (a training step or a forward-only call); what an unread earlier word
  // said is gone with it
  HIPCHK(h, hipMemsetAsync(h->Gbase, 0, GRAD_HEAD * 4, h->st));
  // the control blocks of this pass's persistent launches, cleared in one go (one per layer: run_steps)
  if (h->persist) {
    HIPCHK(h, hipMemsetAsync(h->pctl + 1, 0, (size_t)h->L * sizeof(PersistCtl), h->st));
    HIPCHK(h, hipMemsetAsync(h->xchf, 0, (size_t)h->L * persist_hx_bytes(h->Hp), h->st));   // epoch 0 everywhere (lstm_persist.hip)
  }
  for (int i = 0; i < h->npre; ++i) {
    PhaseScope ps(h, PH_XPROJ);
    int rc = dense_forward(h, i, i == 0 ? h->X0.as<float>() : h->Ybuf[i - 1].as<float>());
    if (rc) return rc;
  }
  for (int l = 0; l < h->L; ++l) {
    {
      PhaseScope ps(h, PH_XPROJ);
      gemm_xproj(h, l, R, h->st);
      HIPCHK(h, hipGetLastError());
    }
    PhaseScope ps(h, PH_RECF);
    int rc = run_steps(h, l, false, 0, T, h->st);
    if (rc) return rc;
    h->n_fwd_launch += h->persist ? 1 : (h->wide && wide_supported(h->Hp, h->Bp)) ? D : T;
  }
  if (h->has_post) {
    PhaseScope ps(h, PH_XPROJ);
    int rc = dense_forward(h, h->npre, h->outb[h->L - 1].as<float>());
    if (rc) return rc;
  }
  if (h->ndense) h->drop_counter += 1;   // one counter value per forward pass
  {
    PhaseScope ps(h, PH_PROJCTC);
    const bool sr = h->cfg.merge == NASR_MERGE_STACK_RESHAPE && D == 2;
    GemmDesc g{};
    g.A = h->has_post ? h->Ybuf[h->npre].as<float>() : h->outb[h->L - 1].as<float>();
    g.B = h->P + h->off_w;
    g.C = h->logits.as<float>();
    g.M = h->Tp * Bp; g.N = h->Cp; g.K = h->Pinp;
    g.lda = sr ? Hp : h->Pinp; g.ldb = h->Cp; g.ldc = h->Cp;
    g.a_map = sr ? h->rowmap_p : nullptr;
    g.a_rows = sr ? 2 * R : R;
    g.bias = h->P + h->off_b;
    // N = Cp (32 for the 29 classes) gives the 128-row tiles of gemm.hip one block column: split K to fill the chip
    g.split_k = gemm_pick_split(g.M, g.N, g.K);
    g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) g.split_k = 1;
    launch_gemm(g, h->st);
    HIPCHK(h, hipGetLastError());
  }
  h->have_fwd = true;
  return NASR_OK;
}

CtcDims ctc_dims(nasr_ctx* h) {
  CtcDims d;
  d.Tp = h->Tp; d.B = h->B; d.Bp = h->Bp; d.C = h->C; d.Cp = h->Cp; d.Lmax = std::max(h->Lmax, 1);
  d.KS = h->KS; d.Tws = h->T + 8;
  d.lprobs = h->ctcprobs.as<float>(); d.goff = h->ctckexp.as<double>();
  return d;
}

int ctc_forward(nasr_ctx* h) {
  PhaseScope ps(h, PH_PROJCTC);
  const CtcDims d = ctc_dims(h);
  launch_ctc_logz(d, h->logits.as<float>(), h->seq_p, h->logz.as<float>(), h->st);
  launch_ctc_alpha_beta(d, h->logits.as<float>(), h->logz.as<float>(), h->labels_p, h->lablen_p,
                        h->seq_p, h->alpha.as<float>(), h->beta.as<float>(), h->aoff.as<double>(),
                        h->boff.as<double>(), h->nll.as<float>(), h->logp.as<double>(), h->st);
  launch_mean(h->nll.as<float>(), h->B, h->loss.as<float>(), h->st);
  if (h->step_decode) {
    if (h->step_greedy)
      launch_greedy(d, h->logits.as<float>(), h->seq_p, h->amax.as<int>(), h->ids.as<int>(), h->lens.as<int>(), h->st);
    h->have_decoded = h->step_greedy;
    // what Network.train returns is known HERE, before the backward pass: copy it out now (nasr_get_step_results)
    h->res_cur ^= 1;
    nasr_ctx::StepRes& r = h->res[h->res_cur];
    const size_t ids_bytes = 8 + (size_t)h->Bp * 4 + (size_t)h->B * h->Tp * 4;
    const size_t lg_off = (ids_bytes + 255) / 256 * 256, lg_bytes = h->step_logits ? (size_t)h->Tp * h->Bp * h->Cp * 4 : 0;
    const size_t bytes = lg_off + lg_bytes;
    if (!pinned_ensure(&r.host, &r.cap, bytes)) return h->fail(NASR_ERR_HIP, "hipHostMalloc of the step results failed");
    char* hp = static_cast<char*>(r.host);
    // the step's logits too (before the CTC gradient overwrites them in place): what tf.nn.ctc_beam_search_decoder reads in
    // the reference's train step (tfnetwork.py:61-64,188-189) - the host decodes them while the device runs on
    // They leave from a snapshot on a stream of their own: the compute stream pays a 1 MB device copy, not the PCIe transfer.
    if (lg_bytes) {
      bool grew = false;
      if (!h->logits_snap.ensure(lg_bytes, &grew)) return h->fail(NASR_ERR_HIP, "hipMalloc of the logits snapshot failed");
      if (!r.ev_lg) HIPCHK(h, hipEventCreateWithFlags(&r.ev_lg, hipEventDisableTiming));
      nasr_ctx::StepRes& prev = h->res[h->res_cur ^ 1];
      if (prev.logits && prev.ev_lg) HIPCHK(h, hipStreamWaitEvent(h->st, prev.ev_lg, 0));      // the snapshot's last reader (long done)
      HIPCHK(h, hipMemcpyAsync(h->logits_snap.p, h->logits.p, lg_bytes, hipMemcpyDeviceToDevice, h->st));
      HIPCHK(h, hipEventRecord(h->ev_snap, h->st));
      HIPCHK(h, hipStreamWaitEvent(h->d2h, h->ev_snap, 0));
      HIPCHK(h, hipMemcpyAsync(hp + lg_off, h->logits_snap.p, lg_bytes, hipMemcpyDeviceToHost, h->d2h));
      HIPCHK(h, hipEventRecord(r.ev_lg, h->d2h));
    }
    r.logits = lg_bytes != 0;
    r.seq = ++h->stamp_seq;
    // (enabled = 2: loss, fault word and logits only - the host runs its own decoder and has no use for the greedy one's)
    launch_publish_results(h->loss.as<float>(), h->Gbase, h->lens.as<int>(), h->step_greedy ? h->Bp : 0, h->ids.as<int>(),
                           h->step_greedy ? h->B * h->Tp : 0, r.host, r.stamp, r.seq, h->st);
    r.greedy = h->step_greedy;
    r.valid = true; r.B = h->B; r.Bp = h->Bp; r.Tp = h->Tp;
  }
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

// weight / bias gradients of layer l from its complete dG, on stream ws: the main stream, or (side = true) the side stream
// with the 3-wave GEMM instantiation that shares the CUs with the persistent BPTT launch of the layer below
int weight_grads(nasr_ctx* h, int l, hipStream_t ws, bool side) {
  const int Bp = h->Bp, T = h->T, D = h->D, Hp = h->Hp, N4 = h->N4;
  const int R = T * Bp;
  float* dG = dg_of(h, l);
  unsigned char* GT = gttp_of(h, l);
  float* cs_part = csws_of(h, l);
  nasr_ctx::SV& gc = gc_of(h, l);
  DevBuf& slab_buf = side ? h->slabs2 : h->slabs;
  auto slabs_for = [&](int split, int M, int N) -> float* {
    if (split <= 1) return nullptr;
    bool grew = false;
    return slab_buf.ensure((size_t)split * M * N * 4, &grew) ? slab_buf.as<float>() : nullptr;
  };
  // a ragged batch contracts over its frames only (h->cmp_rows, nasr_ctx.h): every operand below is then gathered by vrow
  const int rows = h->cmp_rows ? h->cmp_rows : R;
  const int* vm = h->cmp_rows ? h->vrow_p : nullptr;
  const int nkb = (rows + 15) / 16;
  // The side instantiation splits K exactly as the main one would: every output element then sums the same k-blocks in
  // the same order whatever the tile shape - the gradients are bitwise those of the serial order.  (NASR_SIDE_SPLIT=own:
  // the split its own cost model picks, for the A/B logs.)
  static const int split_mode = [] { const char* e = getenv("NASR_SIDE_SPLIT"); return !e ? 0 : e[0] == 'o' ? 1 : e[0] == '1' ? 2 : 0; }();
  const bool side_split = side && split_mode == 1;
  const bool side_one = side && split_mode == 2;       // (A/B logs: no K split at all on the side stream)
  // one pass over dG: its transposed planes + 64-row partial column sums (already there when gemm_dx(l) ran)
  if (h->gttp_layer != l) {
    dg_scales(h, l, R, false, ws);
    launch_tph_split2(dG, nullptr, GT, rows, D * N4, D * N4, nullptr, 1.f, gc.sp(), 1.f, cs_part, ws, vm);
  }
  h->gttp_layer = -1;
  const ActScale ao = act_out(h), ai = lstm_in_scale(h, l);
  // out[l-1] (input weight gradient) and out[l] (recurrent weight gradient) with the frame index as contraction index, where
  // the forward pass has not left them (gemm_xproj).  Compacted rows: out[l] is wanted SHIFTED by one frame, which is no
  // constant row offset there - its planes are built here from the rows vprev (forward direction's columns) / vnext
  // (backward direction's), and the plain transposed planes of out[l] are not needed at all.
  for (int m = std::max(l - 1, 0); m <= (vm ? l - 1 : l); ++m)
    if (!h->ott_valid[m]) {
      launch_tph_split2(h->outb[m].as<float>(), nullptr, h->OTT[m].as<unsigned char>(), rows, D * Hp, D * Hp, nullptr, 1.f, ao.cs, 1.f,
                        nullptr, ws, vm);
      h->ott_valid[m] = 1;
    }
  unsigned char* OS = h->OTS.as<unsigned char>() + (wg_alt(h, l) ? tph_bytes(D * Hp, R) : 0);
  if (vm)
    launch_tph_split2(h->outb[l].as<float>(), nullptr, OS, rows, D * Hp, D * Hp, nullptr, 1.f, ao.cs, 1.f, nullptr, ws, h->vprev_p,
                      D == 2 ? h->vnext_p : nullptr, Hp);
  if (l == 0 && h->npre)
    pl_split(lstm_input(h, l), nullptr, h->X0TTP.as<unsigned char>(), R, h->Ip[0], h->Ip[0], nullptr, ai.cs, nullptr, ws);
  {  // dWx = X^T dG
    GemmTPHDesc g{};
    g.A = l == 0 ? h->X0TTP.as<unsigned char>() : h->OTT[l - 1].as<unsigned char>();
    g.B = GT; g.C = h->G + h->off_wx[l];
    g.M = h->Ip[l]; g.N = D * N4; g.K = rows; g.nkbA = nkb; g.nkbB = nkb; g.ldc = D * N4;
    g.side = side;
    g.split_k = side_one ? 1 : gemm_tph_pick_split(g.M, g.N, g.K, 1, side_split);
    g.slabs = slabs_for(g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) return h->fail(NASR_ERR_HIP, "slab workspace allocation failed");
    pl_gemm(g, ai.cinv, gc.ip(), ws);
  }
  launch_colsum_parts(cs_part, tp_split2_parts(rows), D * N4, h->G + h->off_bias[l], ws);
  {  // dU = shift(H)^T dG : h_prev of frame t is out[t-1] (fw) / out[t+1] (bw); both directions in one launch
    GemmTPHDesc g{};
    g.A = vm ? OS : h->OTT[l].as<unsigned char>(); g.B = GT; g.C = h->G + h->off_u[(size_t)l * D];
    g.M = Hp; g.N = N4; g.K = rows; g.nkbA = nkb; g.nkbB = nkb; g.ldc = N4;
    g.a_kshift = vm ? 0 : -Bp;
    g.nbatch = D;
    g.a_bstride = (size_t)(Hp / 32) * pl_rb_bytes(nkb); g.b_bstride = (size_t)(N4 / 32) * pl_rb_bytes(nkb);
    g.c_bstride = (int64_t)Hp * N4;            // off_u[l*D + 1] - off_u[l*D] (build_layout)
    g.a_kshift1 = vm ? 0 : Bp;
    g.side = side;
    g.split_k = side_one ? 1 : gemm_tph_pick_split(g.M, g.N, g.K, D, side_split);
    g.slabs = slabs_for(g.split_k * D, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) return h->fail(NASR_ERR_HIP, "slab workspace allocation failed");
    pl_gemm(g, ao.cinv, gc.ip(), ws, Hp, N4);
  }
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

// the main stream waits for layer l's side-stream weight gradients (no-op when there are none outstanding)
int wg_join(nasr_ctx* h, int l) {
  if (l >= 0 && l < h->L && h->wg_pending[l]) {
    HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_wg[l], 0));
    h->wg_pending[l] = 0;
  }
  return NASR_OK;
}

int backward(nasr_ctx* h) {
  const int Bp = h->Bp, T = h->T, D = h->D, Hp = h->Hp;
  const int R = T * Bp, Rp = h->Tp * Bp;
  const bool sr = h->cfg.merge == NASR_MERGE_STACK_RESHAPE && D == 2;
  if (h->persist) {
    HIPCHK(h, hipMemsetAsync(h->pctl + 1 + h->L, 0, (size_t)h->L * sizeof(PersistCtl), h->st));
    HIPCHK(h, hipMemsetAsync(h->xchb, 0, (size_t)h->L * persist_px_bytes(), h->st));   // epoch 0 everywhere (lstm_persist.hip)
  }
  {
    PhaseScope ps(h, PH_PROJCTC);
    const CtcDims d = ctc_dims(h);
    launch_ctc_grad(d, h->logits.as<float>(), h->logz.as<float>(), h->lablen_p, h->seq_p,
                    h->cstart_p, h->cpos_p, h->alpha.as<float>(), h->beta.as<float>(),
                    h->aoff.as<double>(), h->boff.as<double>(), h->logp.as<double>(), 1.f / (float)h->B, h->st);
    HIPCHK(h, hipGetLastError());
  }
  {
    PhaseScope ps(h, PH_PROJB);
    // dW = gather(out)^T dlogits
    GemmDesc g{};
    g.A = h->has_post ? h->Ybuf[h->npre].as<float>() : h->outb[h->L - 1].as<float>();
    g.B = h->logits.as<float>();
    g.C = h->G + h->off_w;
    g.M = h->Pinp; g.N = h->Cp; g.K = Rp;
    g.lda = sr ? Hp : h->Pinp; g.ldb = h->Cp; g.ldc = h->Cp;
    g.a_col = true; g.a_map = sr ? h->rowmap_p : nullptr; g.a_rows = sr ? 2 * R : R;
    g.split_k = gemm_pick_split(g.M, g.N, g.K);
    g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) return h->fail(NASR_ERR_HIP, "slab workspace allocation failed");
    launch_gemm(g, h->st);
    launch_colsum(h->logits.as<float>(), Rp, h->Cp, h->Cp, h->G + h->off_b, h->csws.as<float>(), h->st);
    // dOut_last = scatter(dlogits W^T)
    GemmDesc x{};
    x.A = h->logits.as<float>();
    x.B = h->P + h->off_w;
    x.C = h->has_post ? h->dYbuf[h->npre].as<float>() : dout_of(h, h->L - 1);
    x.M = Rp; x.N = h->Pinp; x.K = h->Cp;
    x.lda = h->Cp; x.ldb = h->Cp; x.ldc = sr ? Hp : h->Pinp;
    x.b_col = true; x.a_rows = Rp; x.c_map = sr ? h->rowmap_p : nullptr; x.split_k = 1;
    launch_gemm(x, h->st);
    HIPCHK(h, hipGetLastError());
  }
  if (h->has_post) {
    PhaseScope ps(h, PH_WGRAD);
    int rc = dense_backward(h, h->npre, h->outb[h->L - 1].as<float>(), dout_of(h, h->L - 1));
    if (rc) return rc;
  }
  h->n_bwd_launch = 0;
  h->dgmax_layer = -1;
  for (int l = h->L - 1; l >= 0; --l) {
    const bool defer = (h->persist || h->wide) && h->bucket_defer;
    {
      PhaseScope ps(h, PH_RECB);
      int rc = run_steps(h, l, true, 0, T, h->st);
      if (rc) return rc;
      h->n_bwd_launch += h->persist ? 1 : (h->wide && wide_supported(h->Hp, h->Bp)) ? D : T;
    }
    if (defer && l + 1 < h->L && h->bucket_of_layer[l + 1] >= 0) {   // the layer above's bucket, held back over this launch
      if (int rc = wg_join(h, l + 1)) return rc;
      HIPCHK(h, hipEventRecord(h->ev_bucket[h->bucket_of_layer[l + 1]], h->st));
    }
    PhaseScope ps(h, PH_WGRAD);
    // layer l+2's side-stream weight gradients read the plane buffers this layer's passes are about to rewrite (they
    // finished a BPTT launch ago: the wait costs nothing, it only makes the order formal)
    if (int rc = wg_join(h, l + 2)) return rc;
    if (l > 0 || h->npre > 0) gemm_dx(h, l, R, h->st);   // critical path first
    // layer l's weight gradients feed nothing before Adam: with the overlap on they leave the main stream here and run
    // beside the persistent BPTT launch of layer l-1 (tfnetwork.py:120-128: the gradients are a set, nothing orders them)
    const bool side = h->wg_overlap && h->persist && l > 0 && h->gttp_layer == l;
    if (side) {
      HIPCHK(h, hipEventRecord(h->ev_dx, h->st));
      HIPCHK(h, hipStreamWaitEvent(h->wst, h->ev_dx, 0));
      int rc = weight_grads(h, l, h->wst, true);
      if (rc) return rc;
      HIPCHK(h, hipEventRecord(h->ev_wg[l], h->wst));
      h->wg_pending[l] = 1;
    } else {
      int rc = weight_grads(h, l, h->st, false);
      if (rc) return rc;
    }
    if (h->bucket_of_layer[l] >= 0 && !(defer && l > 0)) {
      if (int rc = wg_join(h, l)) return rc;
      HIPCHK(h, hipEventRecord(h->ev_bucket[h->bucket_of_layer[l]], h->st));
    }
  }
  for (int l = 0; l < h->L; ++l)
    if (int rc = wg_join(h, l)) return rc;     // whatever is still out: before the last bucket / Adam
  for (int i = h->npre - 1; i >= 0; --i) {
    PhaseScope ps(h, PH_WGRAD);
    int rc = dense_backward(h, i, i == 0 ? h->X0.as<float>() : h->Ybuf[i - 1].as<float>(),
                            i > 0 ? h->dYbuf[i - 1].as<float>() : nullptr);
    if (rc) return rc;
  }
  HIPCHK(h, hipEventRecord(h->ev_bucket.back(), h->st));   // the bucket with the fault word: nothing of the step is left
  h->have_grads = true;
  return NASR_OK;
}

int fetch_logits(nasr_ctx* h, float* logits_out) {
  const size_t n = (size_t)h->Tp * h->Bp * h->Cp;
  std::vector<float> host(n);
  HIPCHK(h, hipMemcpyAsync(host.data(), h->logits.p, n * 4, hipMemcpyDeviceToHost, h->st));
  if (int rc = sync_checked(h)) return rc;
  for (int t = 0; t < h->Tp; ++t)
    for (int b = 0; b < h->B; ++b)
      memcpy(logits_out + ((size_t)t * h->B + b) * h->C, host.data() + ((size_t)t * h->Bp + b) * h->Cp,
             (size_t)h->C * 4);
  return NASR_OK;
}

}  // namespace nasr_impl
