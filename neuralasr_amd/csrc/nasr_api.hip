// nasr_api.hip — the C ABI of include/nasr.h: every entry point names the reference interface it replaces there.  The work
// behind them lives in nasr_layout.hip (parameters, operand images), nasr_batch.hip (batches), nasr_pass.hip (the step) and
// nasr_comm.hip (RCCL); the shared handle is nasr_ctx.h.
#include "nasr_ctx.h"

using namespace nasr;
using namespace nasr_impl;

namespace {
int settle_end(nasr_ctx* h, nasr_ctx::StepEnd& e, int* void_out) {
  // the end of THAT step only
  if (!wait_stamp(e.stamp, e.seq, 60.0)) return h->fail(NASR_ERR_HIP, "nasr_settle_step: the step did not end within 60 s");
  if (*e.host != 0.f) {
    *void_out = 1;
    (void)persist_check(h);   // a local abort: this handle continues on the per-step kernels (message in last_error)
  }
  return NASR_OK;
}
}  // namespace

// =============================================================================== C ABI
extern "C" {

int nasr_create(const nasr_model_cfg* cfg, int device_id, void* stream, nasr_handle* out) {
  if (!cfg || !out) {
    g_create_error = "nasr_create: null argument";
    return NASR_ERR_ARG;
  }
  *out = nullptr;
  if (cfg->feature_size < 1 || cfg->hidden < 1 || cfg->num_layers < 1 || cfg->num_classes < 2) {
    g_create_error = "nasr_create: feature_size, hidden, num_layers must be >= 1 and num_classes >= 2";
    return NASR_ERR_ARG;
  }
  if (cfg->bidirectional && cfg->merge != NASR_MERGE_STACK_RESHAPE && cfg->merge != NASR_MERGE_CONCAT) {
    g_create_error = "nasr_create: bidirectional nets need merge = STACK_RESHAPE or CONCAT";
    return NASR_ERR_ARG;
  }
  if (cfg->num_pre < 0 || cfg->num_pre > 3 || cfg->post_width < 0) {
    g_create_error = "nasr_create: num_pre must be in [0,3] and post_width >= 0";
    return NASR_ERR_ARG;
  }
  for (int i = 0; i < cfg->num_pre; ++i)
    if (cfg->pre_width[i] < 1) {
      g_create_error = "nasr_create: pre_width[i] must be >= 1 for i < num_pre";
      return NASR_ERR_ARG;
    }
  for (int i = 0; i < 4; ++i)
    if (!(cfg->dropout[i] >= 0.f && cfg->dropout[i] < 1.f)) {
      g_create_error = "nasr_create: dropout probabilities must be in [0,1)";
      return NASR_ERR_ARG;
    }
  if ((cfg->num_pre > 0 || cfg->post_width > 0) && !(cfg->relu_clip > 0.f)) {
    g_create_error = "nasr_create: relu_clip must be > 0 when dense stages are present";
    return NASR_ERR_ARG;
  }
  if ((cfg->num_pre > 0 || cfg->post_width > 0) && cfg->bidirectional && cfg->merge != NASR_MERGE_CONCAT) {
    g_create_error = "nasr_create: the DeepSpeech family concatenates the directions (merge = CONCAT)";
    return NASR_ERR_ARG;
  }
  if (cfg->bidirectional && cfg->merge == NASR_MERGE_STACK_RESHAPE && cfg->num_layers != 1) {
    g_create_error = "nasr_create: STACK_RESHAPE is the literal 1-layer BiLstmCTCNet; use CONCAT for stacks";
    return NASR_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_error = "nasr_create: no HIP device visible (libnasr has no CPU fallback)";
    return NASR_ERR_HIP;
  }
  if (device_id < 0 || device_id >= ndev) {
    g_create_error = "nasr_create: device_id out of range";
    return NASR_ERR_ARG;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
    g_create_error = "nasr_create: hipGetDeviceProperties failed";
    return NASR_ERR_HIP;
  }
  if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
    g_create_error = std::string("nasr_create: device is ") + prop.gcnArchName + ", libnasr is built for gfx950 only";
    return NASR_ERR_HIP;
  }
  nasr_ctx* h = new nasr_ctx();
  h->cfg = *cfg;
  if (!cfg->bidirectional) h->cfg.merge = NASR_MERGE_NONE;
  h->device = device_id;
  h->lr = cfg->learning_rate;
  auto bail = [&](int code, const std::string& m) {
    g_create_error = m;
    nasr_destroy(h);
    return code;
  };
  if (hipSetDevice(device_id) != hipSuccess) return bail(NASR_ERR_HIP, "hipSetDevice failed");
  if (stream) {
    h->st = reinterpret_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess)
      return bail(NASR_ERR_HIP, "hipStreamCreate failed");
    h->own_stream = true;
  }
  if (build_layout(h) != NASR_OK) return bail(NASR_ERR_ARG, t_err);
  {
    const char* ec = getenv("NASR_COMPACT");
    h->compactable = h->ndense == 0 && !(ec && ec[0] == '0');
  }
  {
    h->sc_wr.resize(h->L); h->sc_wc.resize(h->L);
    h->sc_dr.resize(h->ndense); h->sc_dc.resize(h->ndense); h->sc_yr.resize(h->ndense); h->sc_yc.resize(h->ndense);
    {
      bool ok = gemm_tph_prepare() == hipSuccess, g2 = false;
      int rmax = 1, cmax = 1;
      for (int l = 0; l < h->L; ++l) {
        ok = ok && h->sc_wr[l].ensure((size_t)h->Ip[l]) && h->sc_wc[l].ensure((size_t)h->D * h->N4);
        rmax = std::max(rmax, h->Ip[l]); cmax = std::max(cmax, h->D * h->N4);
      }
      for (int i = 0; i < h->ndense; ++i) {
        ok = ok && h->sc_dr[i].ensure((size_t)h->dIp[i]) && h->sc_dc[i].ensure((size_t)h->dWp[i]);
        rmax = std::max(rmax, h->dIp[i]); cmax = std::max(cmax, h->dWp[i]);
      }
      size_t wsf = 0;     // launch_tph_scales_batch works on all weight matrices at once
      for (int l = 0; l < h->L; ++l) wsf += tph_scale_ws_floats(h->Ip[l], h->D * h->N4);
      for (int i = 0; i < h->ndense; ++i) wsf += tph_scale_ws_floats(h->dIp[i], h->dWp[i]);
      ok = ok && h->scws.ensure(std::max(wsf, tph_scale_ws_floats(rmax, cmax)) * 4, &g2);
      if (!ok) return bail(NASR_ERR_HIP, "set-up of the fp16-plane GEMMs failed");
    }
    {
      size_t of = 0, ob = 0;
      h->off_wftp.resize(h->L); h->off_wbtp.resize(h->L);
      for (int l = 0; l < h->L; ++l) {
        h->off_wftp[l] = of; of += tph_bytes(h->D * h->N4, h->Ip[l]);
        h->off_wbtp[l] = ob; if (l > 0 || h->npre > 0) ob += tph_bytes(h->Ip[l], h->D * h->N4);
      }
      size_t df = 0, db = 0;
      h->off_dftp.assign(h->ndense, 0); h->off_dbtp.assign(h->ndense, 0);
      for (int i = 0; i < h->ndense; ++i) {
        h->off_dftp[i] = df; df += tph_bytes(h->dWp[i], h->dIp[i]);
        h->off_dbtp[i] = db; if (i > 0 || h->npre == 0) db += tph_bytes(h->dIp[i], h->dWp[i]);
      }
      if (hipMalloc(&h->WfTP, of) != hipSuccess ||
          hipMalloc(&h->WbTP, std::max<size_t>(ob, 1024)) != hipSuccess ||
          hipMalloc(&h->DfTP, std::max<size_t>(df, 1024)) != hipSuccess ||
          hipMalloc(&h->DbTP, std::max<size_t>(db, 1024)) != hipSuccess)
        return bail(NASR_ERR_HIP, "hipMalloc of the tiled weight planes failed");
    }
  }
  const size_t nb = (size_t)h->np_int * 4;
  const size_t gb = nb + GRAD_HEAD * 4;   // the gradient buffer starts with the fault word (+ padding): see nasr_grad_device_count
  const size_t ub = (size_t)h->L * h->D * h->Hp * h->N4 * 4;
  if (hipMalloc(&h->P, nb) != hipSuccess || hipMalloc(&h->M, nb) != hipSuccess || hipMalloc(&h->V, nb) != hipSuccess ||
      hipMalloc(&h->Gbase, gb) != hipSuccess || hipMalloc(&h->Uf, ub) != hipSuccess || hipMalloc(&h->Ub, ub) != hipSuccess)
    return bail(NASR_ERR_HIP, "hipMalloc of parameter buffers failed");
  if (hipMalloc(&h->adam_dev, sizeof(AdamDev)) != hipSuccess) return bail(NASR_ERR_HIP, "hipMalloc of the Adam state failed");
  (void)hipMemsetAsync(h->adam_dev, 0, sizeof(AdamDev), h->st);
  (void)hipMemsetAsync(h->P, 0, nb, h->st);
  (void)hipMemsetAsync(h->M, 0, nb, h->st);
  (void)hipMemsetAsync(h->V, 0, nb, h->st);
  h->G = h->Gbase + GRAD_HEAD;
  (void)hipMemsetAsync(h->Gbase, 0, gb, h->st);
  {
    // Buckets for an all-reduce that overlaps the rest of the backward pass (nasr_grad_bucket*): the internal layout
    // is [head | dense stages | layer 0 | ... | layer L-1 | W | b] and backward() finishes W, b first, then the layers
    // from the top down, then the dense stages in front of the stack.  Bucket 0 = layer L-1 + W + b, then one bucket
    // per layer down to layer 1, and a last one with everything in front of layer 1 INCLUDING the fault word, which
    // any launch of the step may still raise.  A one-layer net has a single bucket, unless dense stages precede it.
    h->bucket_of_layer.assign(h->L, -1);
    if (h->L > 1 && h->L <= MAX_BUCKETS) {
      for (int l = h->L - 1; l >= 1; --l) {
        const int64_t lo = h->off_wx[l], hi = l == h->L - 1 ? h->np_int : h->off_wx[l + 1];
        h->bucket_of_layer[l] = (int)h->buckets.size();
        h->buckets.push_back({GRAD_HEAD + lo, hi - lo});
      }
      h->buckets.push_back({0, GRAD_HEAD + h->off_wx[1]});
    } else if (h->L == 1 && h->npre > 0) {
      // a DeepSpeech-shaped net: the (Bi)LSTM's gradients (4/5 of the parameters at the reference's widths) + W + b are
      // complete before the backward pass of the dense stages in front of it, which then hides their all-reduce
      h->bucket_of_layer[0] = 0;
      h->buckets.push_back({GRAD_HEAD + h->off_wx[0], h->np_int - h->off_wx[0]});
      h->buckets.push_back({0, GRAD_HEAD + h->off_wx[0]});
    } else {
      h->buckets.push_back({0, GRAD_HEAD + h->np_int});
    }
    h->ev_bucket.resize(h->buckets.size());
    for (auto& e2 : h->ev_bucket)
      if (hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return bail(NASR_ERR_HIP, "hipEventCreate failed");
  }
  (void)hipMemsetAsync(h->Uf, 0, ub, h->st);
  (void)hipMemsetAsync(h->Ub, 0, ub, h->st);
  {
    const char* e = getenv("NASR_PERSIST");
    h->persist = !(e && e[0] == '0') && persist_supported(h->Hp) && prop.multiProcessorCount == 256;
    if (h->persist) {
      h->imf = persist_image_floats(h->Hp, false);
      h->imb = persist_image_floats(h->Hp, true);
      const size_t nk = (size_t)h->L * h->D;
      if (persist_prepare() != hipSuccess || hipMalloc(&h->Upf, nk * h->imf * 4) != hipSuccess ||
          hipMalloc(&h->Upb, nk * h->imb * 4) != hipSuccess ||
          hipMalloc(&h->xchf, (size_t)h->L * persist_hx_bytes(h->Hp)) != hipSuccess ||
          hipMalloc(&h->xchb, (size_t)h->L * persist_px_bytes()) != hipSuccess ||
          hipMalloc(&h->pctl, (size_t)(1 + 2 * h->L) * sizeof(PersistCtl)) != hipSuccess ||   // [0] census, then one per layer pass
          hipHostMalloc(&h->perr, 64, hipHostMallocMapped) != hipSuccess)
        return bail(NASR_ERR_HIP, "allocation of the persistent-recurrence buffers failed");
      *h->perr = 0;
      const char* er = getenv("NASR_REC");
      h->rec_f16 = !(er && std::string(er) == "f32");
      if (h->rec_f16) {
        bool g2 = false;
        size_t wsf = 0;
        for (size_t k = 0; k < nk; ++k) wsf += tph_scale_ws_floats(h->Hp, h->N4);
        if (hipMalloc(&h->Ucs, nk * h->N4 * 4) != hipSuccess || hipMalloc(&h->Ucinv, nk * h->N4 * 4) != hipSuccess ||
            !h->scws.ensure(wsf * 4, &g2))
          return bail(NASR_ERR_HIP, "allocation of the recurrent-weight scales failed");
        (void)hipMemsetAsync(h->Ucs, 0, nk * h->N4 * 4, h->st);
        (void)hipMemsetAsync(h->Ucinv, 0, nk * h->N4 * 4, h->st);
      }
      (void)hipMemsetAsync(h->Upf, 0, nk * h->imf * 4, h->st);
      (void)hipMemsetAsync(h->Upb, 0, nk * h->imb * 4, h->st);
    }
  }
  {
    const char* e = getenv("NASR_PERSIST");
    const char* ew = getenv("NASR_WIDE");
    h->wide = !h->persist && !(e && e[0] == '0') && !(ew && ew[0] == '0') && wide_supported(h->Hp, 16) &&
              prop.multiProcessorCount == 256;
    if (h->wide) {
      const size_t nk = (size_t)h->L * h->D;
      bool g2 = false;
      size_t wsf = 0;
      for (size_t k = 0; k < nk; ++k) wsf += tph_scale_ws_floats(h->Hp, h->N4);
      if (wide_prepare() != hipSuccess || hipMalloc(&h->Uw, nk * wide_image_bytes(h->Hp)) != hipSuccess ||
          hipMalloc(&h->whx, wide_hx_bytes(64)) != hipSuccess || hipMalloc(&h->wpart, wide_part_bytes(64)) != hipSuccess ||
          hipMalloc(&h->wctl, sizeof(WideCtl)) != hipSuccess || hipMalloc(&h->Uwb, nk * wide_image_bytes(h->Hp)) != hipSuccess ||
          hipMalloc(&h->wpx, wide_px_bytes(64)) != hipSuccess || hipMalloc(&h->Urs, nk * h->Hp * 4) != hipSuccess ||
          hipMalloc(&h->Urinv, nk * h->Hp * 4) != hipSuccess || hipMalloc(&h->wsrow, 2 * 64 * 4) != hipSuccess ||
          hipMalloc(&h->Ucs, nk * h->N4 * 4) != hipSuccess ||
          hipMalloc(&h->Ucinv, nk * h->N4 * 4) != hipSuccess || !h->scws.ensure(wsf * 4, &g2) ||
          (!h->perr && hipHostMalloc(&h->perr, 64, hipHostMallocMapped) != hipSuccess))
        return bail(NASR_ERR_HIP, "allocation of the wide persistent-recurrence buffers failed");
      *h->perr = 0;
      (void)hipMemsetAsync(h->whx, 0, wide_hx_bytes(64), h->st);
      (void)hipMemsetAsync(h->wpx, 0, wide_px_bytes(64), h->st);
      h->wide_wanted = true;
    }
  }
  h->gates.resize(h->L);
  h->OTT.resize(h->L);
  h->ott_valid.assign(h->L, 0);
  h->outb.resize(h->L);
  h->cbuf.resize(h->L);
  h->Ybuf.resize(h->ndense);
  h->dYbuf.resize(h->ndense);
  if (hipStreamCreateWithFlags(&h->cst, hipStreamNonBlocking) != hipSuccess) return bail(NASR_ERR_HIP, "hipStreamCreate (copy stream) failed");
  if (hipStreamCreateWithFlags(&h->d2h, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_snap, hipEventDisableTiming) != hipSuccess)
    return bail(NASR_ERR_HIP, "hipStreamCreate (results stream) failed");
  {
    const char* eo = getenv("NASR_WGRAD_OVERLAP");
    const bool eligible = h->persist && h->Hp == 512 && h->L > 1;
    h->wg_overlap = eligible && !(eo && eo[0] == '0');          // on unless NASR_WGRAD_OVERLAP=0 (nasr_set_wgrad_overlap)
    h->ev_wg.assign(h->L, nullptr);
    h->wg_pending.assign(h->L, 0);
    if (eligible) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = lowest priority (largest number)
      if (hipStreamCreateWithPriority(&h->wst, hipStreamNonBlocking, lo) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_dx, hipEventDisableTiming) != hipSuccess)
        return bail(NASR_ERR_HIP, "set-up of the weight-gradient side stream failed");
      for (auto& e : h->ev_wg)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(NASR_ERR_HIP, "hipEventCreate failed");
      persist_set_bwd_lean(true);
    }
  }
  for (BatchSlot& bs : h->slots)
    if (hipEventCreateWithFlags(&bs.ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&bs.ev_released, hipEventDisableTiming) != hipSuccess)
      return bail(NASR_ERR_HIP, "hipEventCreate failed");
  for (auto& r : h->res) {
    if (hipHostMalloc(reinterpret_cast<void**>(&r.stamp), 64, hipHostMallocMapped) != hipSuccess)
      return bail(NASR_ERR_HIP, "set-up of the step-result stamps failed");
    *r.stamp = 0;
  }
  for (auto& e : h->endw) {
    if (hipHostMalloc(reinterpret_cast<void**>(&e.host), 64, hipHostMallocMapped) != hipSuccess)
      return bail(NASR_ERR_HIP, "set-up of the step-end words failed");
    e.stamp = reinterpret_cast<uint32_t*>(e.host) + 8;
    *e.host = 0.f;
    *e.stamp = 0;
  }
  (void)hipEventCreate(&h->ev_total_a);
  (void)hipEventCreate(&h->ev_total_b);
  memset(&h->last_times, 0, sizeof(h->last_times));
  if (hipStreamSynchronize(h->st) != hipSuccess) return bail(NASR_ERR_HIP, "stream synchronize failed in create");
  if (h->persist) {
    if (!persist_census(h)) h->persist = false;
    h->persist_ok = h->persist;
    h->persist_wanted = h->persist;
    const char* er = getenv("NASR_PERSIST_REARM");
    h->rearm_after = er && *er ? std::max<long long>(0, atoll(er)) : 200;
    const char* eb = getenv("NASR_BUCKET_DEFER");
    h->bucket_defer = !(eb && eb[0] == '0');
  }
  if (h->wide) {
    const char* er = getenv("NASR_PERSIST_REARM");
    h->rearm_after = er && *er ? std::max<long long>(0, atoll(er)) : 200;
  }
  *out = h;
  return NASR_OK;
}

int nasr_destroy(nasr_handle h) {
  if (!h) return NASR_OK;
  (void)hipSetDevice(h->device);
  if (h->st) (void)hipStreamSynchronize(h->st);
  for (hipEvent_t e : h->ev_bucket) (void)hipEventDestroy(e);
  drop_graphs(h);
  for (float* p : {h->P, h->M, h->V, h->Gbase, h->Uf, h->Ub, h->Upf, h->Upb, h->xchf, h->xchb, h->Ucs, h->Ucinv})
    if (p) (void)hipFree(p);
  if (h->WfTP) (void)hipFree(h->WfTP);
  if (h->WbTP) (void)hipFree(h->WbTP);
  if (h->DfTP) (void)hipFree(h->DfTP);
  if (h->DbTP) (void)hipFree(h->DbTP);
  h->DTP.release();
  for (auto& b : h->Ybuf) b.release();
  for (auto& b : h->dYbuf) b.release();
  if (h->pctl) (void)hipFree(h->pctl);
  if (h->Uw) (void)hipFree(h->Uw);
  if (h->Uwb) (void)hipFree(h->Uwb);
  if (h->wpx) (void)hipFree(h->wpx);
  for (float* p : {h->Urs, h->Urinv, h->wsrow})
    if (p) (void)hipFree(p);
  if (h->whx) (void)hipFree(h->whx);
  if (h->wpart) (void)hipFree(h->wpart);
  if (h->wctl) (void)hipFree(h->wctl);
  if (h->perr) (void)hipHostFree(h->perr);
  for (DevBuf* b : {&h->XTP, &h->X0TTP, &h->GTP, &h->GTTP, &h->scws, &h->GTTP2, &h->csws2, &h->slabs2}) b->release();
  h->sc_gc2.release();
  h->sc_cr.release(); h->sc_cx.release(); h->OTS.release();
  if (h->wst) { (void)hipStreamSynchronize(h->wst); (void)hipStreamDestroy(h->wst); }
  if (h->ev_dx) (void)hipEventDestroy(h->ev_dx);
  for (hipEvent_t e : h->ev_wg) if (e) (void)hipEventDestroy(e);
  for (auto& b : h->OTT) b.release();
  for (nasr_ctx::SV* v : {&h->sc15, &h->sc_x0r, &h->sc_x0c, &h->sc_gr, &h->sc_gc}) v->release();
  for (auto* vec : {&h->sc_yr, &h->sc_yc, &h->sc_wr, &h->sc_wc, &h->sc_dr, &h->sc_dc})
    for (auto& v : *vec) v.release();
  (void)nasr_comm_destroy(h);
  if (h->adam_dev) (void)hipFree(h->adam_dev);
  if (h->d2h) {
    (void)hipStreamSynchronize(h->d2h);
    (void)hipStreamDestroy(h->d2h);
  }
  if (h->ev_snap) (void)hipEventDestroy(h->ev_snap);
  h->logits_snap.release();
  for (auto& r : h->res) {
    if (r.host) (void)hipHostFree(r.host);
    if (r.stamp) (void)hipHostFree(r.stamp);
    if (r.ev_lg) (void)hipEventDestroy(r.ev_lg);
  }
  for (auto& e : h->endw)
    if (e.host) (void)hipHostFree(e.host);
  if (h->cst) {
    (void)hipStreamSynchronize(h->cst);
    (void)hipStreamDestroy(h->cst);
  }
  for (BatchSlot& bs : h->slots) {
    bs.dfeats.release();
    bs.dmeta.release();
    if (bs.hfeats) (void)hipHostFree(bs.hfeats);
    if (bs.hmeta) (void)hipHostFree(bs.hmeta);
    if (bs.ev_copy) (void)hipEventDestroy(bs.ev_copy);
    if (bs.ev_released) (void)hipEventDestroy(bs.ev_released);
  }
  for (DevBuf* b : {&h->seqbuf, &h->X0, &h->dout, &h->hstate, &h->partial, &h->dcstate, &h->dgbuf, &h->logits, &h->logz,
                    &h->alpha, &h->beta, &h->aoff, &h->boff, &h->logp, &h->nll, &h->loss, &h->slabs, &h->ctcprobs, &h->ctckexp,
                    &h->csws, &h->amax, &h->ids, &h->lens, &h->stage, &h->dgmax})
    b->release();
  for (auto& b : h->gates) b.release();
  for (auto& b : h->outb) b.release();
  for (auto& b : h->cbuf) b.release();
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->ev_total_a) (void)hipEventDestroy(h->ev_total_a);
  if (h->ev_total_b) (void)hipEventDestroy(h->ev_total_b);
  if (h->own_stream && h->st) (void)hipStreamDestroy(h->st);
  delete h;
  return NASR_OK;
}

const char* nasr_last_error(nasr_handle h) {
  if (!h) return g_create_error.c_str();
  return t_err_handle == h ? t_err.c_str() : "";
}
const char* nasr_backend(nasr_handle) { return "hip-gfx950"; }

int nasr_synchronize(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  return sync_checked(h);
}

int64_t nasr_param_count(nasr_handle h) { return h ? h->np_tf : -1; }
int nasr_num_tensors(nasr_handle h) { return h ? (int)h->tensors.size() : -1; }

int nasr_tensor_info(nasr_handle h, int idx, char name[64], int64_t* offset, int64_t* rows, int64_t* cols) {
  if (!h) return NASR_ERR_ARG;
  if (idx < 0 || idx >= (int)h->tensors.size()) return h->fail(NASR_ERR_ARG, "tensor index out of range");
  const TensorInfo& t = h->tensors[idx];
  if (name) {
    strncpy(name, t.name.c_str(), 63);
    name[63] = 0;
  }
  if (offset) *offset = t.offset;
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return NASR_OK;
}

int nasr_set_params(nasr_handle h, const float* flat, int64_t n) {
  if (!h || !flat) return NASR_ERR_ARG;
  if (n != h->np_tf) return h->fail(NASR_ERR_ARG, "nasr_set_params: expected " + std::to_string(h->np_tf) + " floats");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = scatter_to_device(h, flat, h->P);
  if (rc) return rc;
  rc = repack(h);
  if (rc) return rc;
  HIPCHK(h, hipStreamSynchronize(h->st));
  return NASR_OK;
}

int nasr_get_params(nasr_handle h, float* flat, int64_t n) {
  if (!h || !flat) return NASR_ERR_ARG;
  if (n != h->np_tf) return h->fail(NASR_ERR_ARG, "nasr_get_params: wrong length");
  HIPCHK(h, hipSetDevice(h->device));
  return gather_from_device(h, h->P, flat);
}

int nasr_set_adam_state(nasr_handle h, const float* m, const float* v, int64_t n, int64_t step) {
  if (!h || !m || !v) return NASR_ERR_ARG;
  if (n != h->np_tf || step < 0) return h->fail(NASR_ERR_ARG, "nasr_set_adam_state: wrong length or negative step");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = scatter_to_device(h, m, h->M);
  if (rc) return rc;
  rc = scatter_to_device(h, v, h->V);
  if (rc) return rc;
  const AdamDev init{(long long)step, 0.f, 0};
  HIPCHK(h, hipMemcpyAsync(h->adam_dev, &init, sizeof(init), hipMemcpyHostToDevice, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  return NASR_OK;
}

int nasr_get_adam_state(nasr_handle h, float* m, float* v, int64_t n, int64_t* step) {
  if (!h) return NASR_ERR_ARG;
  if (n != h->np_tf) return h->fail(NASR_ERR_ARG, "nasr_get_adam_state: wrong length");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = NASR_OK;
  if (m) rc = gather_from_device(h, h->M, m);
  if (!rc && v) rc = gather_from_device(h, h->V, v);
  if (step) {
    AdamDev d{};
    HIPCHK(h, hipMemcpyAsync(&d, h->adam_dev, sizeof(d), hipMemcpyDeviceToHost, h->st));
    if (int rc2 = sync_checked(h)) return rc2;
    *step = d.step;
  }
  return rc;
}

int nasr_set_learning_rate(nasr_handle h, float lr) {
  if (!h) return NASR_ERR_ARG;
  h->lr = lr;
  return NASR_OK;
}

int nasr_logit_frames(nasr_handle h, int T) {
  if (!h) return NASR_ERR_ARG;
  return (h->cfg.bidirectional && h->cfg.merge == NASR_MERGE_STACK_RESHAPE) ? 2 * T : T;
}

int nasr_upload_batch(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                      const int32_t* label_len, int B, int T, int Lmax) {
  if (!h) return NASR_ERR_ARG;
  return upload(h, feats, seq_len, labels, label_len, B, T, Lmax);
}

int nasr_upload_batch_context(nasr_handle h, const float* centre, const float* pad_value, int numcontext, int numcep,
                              const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B, int T,
                              int Lmax) {
  if (!h) return NASR_ERR_ARG;
  if (!centre) return h->fail(NASR_ERR_ARG, "null input buffer");
  return upload(h, nullptr, seq_len, labels, label_len, B, T, Lmax, centre, pad_value, numcontext, numcep);
}

int nasr_stage_batch(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                     const int32_t* label_len, int B, int T, int Lmax, int* ticket) {
  if (!h) return NASR_ERR_ARG;
  return stage(h, feats, seq_len, labels, label_len, B, T, Lmax, nullptr, nullptr, 0, 0, ticket);
}

int nasr_stage_batch_context(nasr_handle h, const float* centre, const float* pad_value, int numcontext, int numcep,
                             const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B, int T,
                             int Lmax, int* ticket) {
  if (!h) return NASR_ERR_ARG;
  if (!centre) return h->fail(NASR_ERR_ARG, "null input buffer");
  return stage(h, nullptr, seq_len, labels, label_len, B, T, Lmax, centre, pad_value, numcontext, numcep, ticket);
}

int nasr_commit_batch(nasr_handle h, int ticket) {
  if (!h) return NASR_ERR_ARG;
  BatchSlot* s = slot_of_ticket(h, ticket);
  if (!s) return h->fail(NASR_ERR_STATE, "nasr_commit_batch: no staged batch behind this ticket");
  const int rc = slot_commit(h, s);
  if (rc && h->cur != s) slot_set_state(h, s, SLOT_FREE);
  return rc;
}

int nasr_discard_batch(nasr_handle h, int ticket) {
  if (!h) return NASR_ERR_ARG;
  BatchSlot* s = slot_of_ticket(h, ticket);
  if (!s) return h->fail(NASR_ERR_STATE, "nasr_discard_batch: no staged batch behind this ticket");
  slot_set_state(h, s, SLOT_FREE);
  return NASR_OK;
}

int nasr_compute_grads(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  if (!h->resident) return h->fail(NASR_ERR_STATE, "no resident batch");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->profiling && !h->window_open) {  // a fresh timing window per step when the batch stays resident
    h->ev_used = 0;
    h->spans.clear();
    (void)hipEventRecord(h->ev_total_a, h->st);
    h->window_open = true;
    h->total_valid = false;
  }
  persist_rearm(h);
  int rc = forward(h);   // clears the step's fault word
  if (rc) return rc;
  rc = ctc_forward(h);
  if (rc) return rc;
  return backward(h);
}

void* nasr_grad_device_ptr(nasr_handle h) { return h ? h->Gbase : nullptr; }
int64_t nasr_grad_device_count(nasr_handle h) { return h ? h->np_int + GRAD_HEAD : -1; }

int nasr_grad_bucket_count(nasr_handle h) { return h ? (int)h->buckets.size() : NASR_ERR_ARG; }

int nasr_grad_bucket(nasr_handle h, int i, int64_t* offset, int64_t* count) {
  if (!h) return NASR_ERR_ARG;
  if (i < 0 || i >= (int)h->buckets.size() || !offset || !count) return h->fail(NASR_ERR_ARG, "nasr_grad_bucket: bad index");
  *offset = h->buckets[i].first;
  *count = h->buckets[i].second;
  return NASR_OK;
}

int nasr_grad_bucket_wait(nasr_handle h, int i, void* hip_stream) {
  if (!h) return NASR_ERR_ARG;
  if (i < 0 || i >= (int)h->buckets.size()) return h->fail(NASR_ERR_ARG, "nasr_grad_bucket_wait: bad index");
  HIPCHK(h, hipStreamWaitEvent((hipStream_t)hip_stream, h->ev_bucket[i], 0));
  return NASR_OK;
}

int nasr_diag_bucket_traffic(nasr_handle h, int i, void* hip_stream, int nblocks, int passes) {
  if (!h) return NASR_ERR_ARG;
  if (i < 0 || i >= (int)h->buckets.size() || nblocks < 1 || nblocks > 1024 || passes < 1)
    return h->fail(NASR_ERR_ARG, "nasr_diag_bucket_traffic: bad bucket index, nblocks (1..1024) or passes");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamWaitEvent((hipStream_t)hip_stream, h->ev_bucket[i], 0));
  launch_ring_standin(h->Gbase + h->buckets[i].first, h->buckets[i].second, nblocks, passes, (hipStream_t)hip_stream);
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

int nasr_apply_adam(nasr_handle h, float grad_scale) {
  if (!h) return NASR_ERR_ARG;
  if (!h->have_grads) return h->fail(NASR_ERR_STATE, "nasr_apply_adam without gradients");
  HIPCHK(h, hipSetDevice(h->device));
  {
    PhaseScope ps(h, PH_ADAM);
    launch_adam(h->P, h->M, h->V, h->G, h->np_int, h->adam_dev, h->lr, h->cfg.beta1, h->cfg.beta2, h->cfg.epsilon, grad_scale,
                h->Gbase, h->st);
    int rc = repack(h);
    if (rc) return rc;
    // the step's fault word as it stands now (all-reduced with the gradients): read later, without a stream sync
    h->end_cur = (h->end_cur + 1) % nasr_ctx::NEND;
    nasr_ctx::StepEnd& e = h->endw[h->end_cur];
    e.seq = ++h->stamp_seq;
    e.token = ++h->step_token;
    launch_stamp(e.stamp, e.seq, e.host, (const float*)h->Gbase, h->st);
    e.valid = true;
  }
  if (h->profiling && h->window_open) {
    (void)hipEventRecord(h->ev_total_b, h->st);
    h->window_open = false;
    h->total_valid = true;
  }
  h->have_grads = false;
  return NASR_OK;
}

int nasr_get_grads(nasr_handle h, float* flat, int64_t n) {
  if (!h || !flat) return NASR_ERR_ARG;
  if (n != h->np_tf) return h->fail(NASR_ERR_ARG, "nasr_get_grads: wrong length");
  if (!h->have_grads) return h->fail(NASR_ERR_STATE, "nasr_get_grads without gradients");
  HIPCHK(h, hipSetDevice(h->device));
  return gather_from_device(h, h->G, flat);
}

int nasr_set_grads(nasr_handle h, const float* flat, int64_t n) {
  if (!h || !flat) return NASR_ERR_ARG;
  if (n != h->np_tf) return h->fail(NASR_ERR_ARG, "nasr_set_grads: wrong length");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = scatter_to_device(h, flat, h->G);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(h->Gbase, 0, GRAD_HEAD * 4, h->st));
  h->have_grads = true;
  return NASR_OK;
}

int nasr_label_error_rate(const int32_t* hyp_ids, const int32_t* hyp_lens, int hyp_stride, const int32_t* labels,
                          const int32_t* label_len, int Lmax, int B, float* ler_out) {
  if (!hyp_ids || !hyp_lens || !labels || !label_len || !ler_out || B < 1) return NASR_ERR_ARG;
  double acc = 0.0;
  std::vector<int> row;
  for (int b = 0; b < B; ++b) {
    const int n = hyp_lens[b], m = label_len[b];
    const int32_t* hy = hyp_ids + (size_t)b * hyp_stride;
    const int32_t* tr = labels + (size_t)b * Lmax;
    if (m == 0) {
      acc += n > 0 ? INFINITY : 0.0;
      continue;
    }
    row.resize((size_t)m + 1);
    for (int j = 0; j <= m; ++j) row[j] = j;
    for (int i = 1; i <= n; ++i) {
      int prev = row[0];
      row[0] = i;
      for (int j = 1; j <= m; ++j) {
        const int cur = row[j];
        const int sub = prev + (hy[i - 1] != tr[j - 1] ? 1 : 0);
        row[j] = std::min(std::min(row[j] + 1, row[j - 1] + 1), sub);
        prev = cur;
      }
    }
    acc += (double)row[m] / (double)m;
  }
  *ler_out = (float)(acc / B);
  return NASR_OK;
}

int nasr_get_loss(nasr_handle h, float* loss_out) {
  if (!h || !loss_out) return NASR_ERR_ARG;
  // the step's fault word travels with the gradients through the all-reduce: non-zero = some rank's persistent
  // recurrence gave up, every rank's Adam launch of that step was a no-op (optim.hip) and the step is void everywhere
  float fault = 0.f;
  HIPCHK(h, hipMemcpyAsync(loss_out, h->loss.p, 4, hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(&fault, h->Gbase, 4, hipMemcpyDeviceToHost, h->st));
  const int rc = sync_checked(h);
  if (fault != 0.f) {
    if (rc) return rc;
    return h->fail(NASR_ERR_HIP, "this training step is void: the persistent recurrence of another rank aborted; no "
                                 "parameters were changed on any rank");
  }
  return rc;
}

int nasr_step_void(nasr_handle h, int* void_out) {
  if (!h || !void_out) return NASR_ERR_ARG;
  float fault = 0.f;
  HIPCHK(h, hipMemcpyAsync(&fault, h->Gbase, 4, hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  (void)persist_check(h);   // a local abort: switch this handle to the per-step kernels (the message stays in last_error)
  *void_out = fault != 0.f ? 1 : 0;
  return NASR_OK;
}

int nasr_get_step_results(nasr_handle h, float* loss_out, int* fault_out, int32_t* ids_out, int32_t* lens_out) {
  if (!h) return NASR_ERR_ARG;
  nasr_ctx::StepRes& r = h->res[h->res_cur];
  if (!r.valid) return h->fail(NASR_ERR_STATE, "nasr_get_step_results: no step with nasr_set_step_decode(1) has been enqueued");
  // the forward pass + CTC of the step; its backward pass may still run
  if (!wait_stamp(r.stamp, r.seq, 60.0)) return h->fail(NASR_ERR_HIP, "nasr_get_step_results: the step's results did not arrive within 60 s");
  const char* hp = static_cast<const char*>(r.host);
  float fault;
  memcpy(&fault, hp + 4, 4);
  if (loss_out) memcpy(loss_out, hp, 4);
  if (fault_out) *fault_out = fault != 0.f ? 1 : 0;
  if (!r.greedy) {                      // nasr_set_step_decode(h, 2): no greedy decode was run - empty hypotheses
    if (lens_out) memset(lens_out, 0, (size_t)r.B * 4);
    if (ids_out) memset(ids_out, 0, (size_t)r.B * r.Tp * 4);
    return NASR_OK;
  }
  if (lens_out) memcpy(lens_out, hp + 8, (size_t)r.B * 4);
  if (ids_out) memcpy(ids_out, hp + 8 + (size_t)r.Bp * 4, (size_t)r.B * r.Tp * 4);
  return NASR_OK;
}


int nasr_settle_step(nasr_handle h, int previous, int* void_out) {
  if (!h || !void_out) return NASR_ERR_ARG;
  *void_out = 0;
  nasr_ctx::StepEnd& e = h->endw[previous ? (h->end_cur + nasr_ctx::NEND - 1) % nasr_ctx::NEND : h->end_cur];
  if (!e.valid) return NASR_OK;
  return settle_end(h, e, void_out);
}

int64_t nasr_step_token(nasr_handle h) { return h ? h->step_token : -1; }

int nasr_settle_token(nasr_handle h, int64_t token, int* void_out) {
  if (!h || !void_out) return NASR_ERR_ARG;
  *void_out = 0;
  if (token <= 0 || token > h->step_token) return h->fail(NASR_ERR_ARG, "nasr_settle_token: no such step");
  for (auto& e : h->endw)
    if (e.valid && e.token == token) return settle_end(h, e, void_out);
  return h->fail(NASR_ERR_STATE, "nasr_settle_token: the library remembers the last " + std::to_string(nasr_ctx::NEND) +
                                     " optimiser steps; this token is older");
}

int nasr_resident_frames(nasr_handle h, int64_t* frames) {
  if (!h || !frames) return NASR_ERR_ARG;
  *frames = h->resident ? h->frames : 0;
  return NASR_OK;
}

int nasr_set_row_compaction(nasr_handle h, int enabled) {
  if (!h) return NASR_ERR_ARG;
  // what the resident batch's plane buffers hold depends on it: takes effect with the next uploaded / committed batch
  h->compactable = enabled && h->ndense == 0;
  return NASR_OK;
}

int nasr_resident_rows(nasr_handle h, int64_t* rows) {
  if (!h || !rows) return NASR_ERR_ARG;
  *rows = !h->resident ? 0 : h->cmp_rows ? h->cmp_rows : (int64_t)h->T * h->Bp;
  return NASR_OK;
}

int nasr_train_step(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                    const int32_t* label_len, int B, int T, int Lmax, float* loss_out) {
  if (!h) return NASR_ERR_ARG;
  if (!labels) return h->fail(NASR_ERR_ARG, "nasr_train_step needs labels");
  int rc = upload(h, feats, seq_len, labels, label_len, B, T, Lmax);
  if (rc) return rc;
  rc = nasr_compute_grads(h);
  if (rc) return rc;
  rc = nasr_apply_adam(h, 1.f);
  if (rc) return rc;
  if (loss_out) return nasr_get_loss(h, loss_out);
  return NASR_OK;
}

int nasr_forward(nasr_handle h, const float* feats, const int32_t* seq_len, int B, int T, float* logits_out) {
  if (!h) return NASR_ERR_ARG;
  int rc = upload(h, feats, seq_len, nullptr, nullptr, B, T, 0);
  if (rc) return rc;
  rc = forward(h);
  if (rc) return rc;
  if (logits_out) return fetch_logits(h, logits_out);
  return nasr_synchronize(h);
}

int nasr_loss(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
              const int32_t* label_len, int B, int T, int Lmax, float* loss_out, float* nll_out) {
  if (!h) return NASR_ERR_ARG;
  if (!labels) return h->fail(NASR_ERR_ARG, "nasr_loss needs labels");
  int rc = upload(h, feats, seq_len, labels, label_len, B, T, Lmax);
  if (rc) return rc;
  rc = forward(h);
  if (rc) return rc;
  rc = ctc_forward(h);
  if (rc) return rc;
  if (nll_out) HIPCHK(h, hipMemcpyAsync(nll_out, h->nll.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->st));
  if (loss_out) return nasr_get_loss(h, loss_out);
  return nasr_synchronize(h);
}

int nasr_loss_and_grads(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                        const int32_t* label_len, int B, int T, int Lmax, float* loss_out, float* nll_out,
                        float* flat_grads_out) {
  if (!h) return NASR_ERR_ARG;
  if (!labels) return h->fail(NASR_ERR_ARG, "nasr_loss_and_grads needs labels");
  int rc = upload(h, feats, seq_len, labels, label_len, B, T, Lmax);
  if (rc) return rc;
  rc = nasr_compute_grads(h);
  if (rc) return rc;
  if (nll_out) HIPCHK(h, hipMemcpyAsync(nll_out, h->nll.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->st));
  if (loss_out) {
    rc = nasr_get_loss(h, loss_out);
    if (rc) return rc;
  }
  if (flat_grads_out) return gather_from_device(h, h->G, flat_grads_out);
  return nasr_synchronize(h);
}

int nasr_greedy_decode(nasr_handle h, const float* feats, const int32_t* seq_len, int B, int T, int32_t* ids_out,
                       int32_t* lens_out) {
  if (!h || !ids_out || !lens_out) return NASR_ERR_ARG;
  int rc = upload(h, feats, seq_len, nullptr, nullptr, B, T, 0);
  if (rc) return rc;
  rc = forward(h);
  if (rc) return rc;
  const CtcDims d = ctc_dims(h);
  launch_greedy(d, h->logits.as<float>(), h->seq_p, h->amax.as<int>(), h->ids.as<int>(), h->lens.as<int>(),
                h->st);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(lens_out, h->lens.p, (size_t)B * 4, hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(ids_out, h->ids.p, (size_t)B * h->Tp * 4, hipMemcpyDeviceToHost, h->st));
  return sync_checked(h);
}

int nasr_set_step_decode(nasr_handle h, int enabled) {
  if (!h) return NASR_ERR_ARG;
  h->step_decode = enabled != 0;
  h->step_greedy = (enabled & 1) != 0;
  h->step_logits = (enabled & 2) != 0;
  return NASR_OK;
}

int nasr_get_step_logits(nasr_handle h, float* logits_out) {
  if (!h || !logits_out) return NASR_ERR_ARG;
  nasr_ctx::StepRes& r = h->res[h->res_cur];
  if (!r.valid || !r.logits)
    return h->fail(NASR_ERR_STATE, "nasr_get_step_logits: no step with nasr_set_step_decode(h, 3) has been enqueued");
  if (!wait_stamp(r.stamp, r.seq, 60.0)) return h->fail(NASR_ERR_HIP, "nasr_get_step_logits: the step's results did not arrive within 60 s");
  HIPCHK(h, hipEventSynchronize(r.ev_lg));          // the logits travel on a stream of their own (ctc_forward)
  const size_t ids_bytes = 8 + (size_t)r.Bp * 4 + (size_t)r.B * r.Tp * 4;
  const float* src = reinterpret_cast<const float*>(static_cast<const char*>(r.host) + (ids_bytes + 255) / 256 * 256);
  for (int t = 0; t < r.Tp; ++t)
    for (int b = 0; b < r.B; ++b)
      memcpy(logits_out + ((size_t)t * r.B + b) * h->C, src + ((size_t)t * r.Bp + b) * h->Cp, (size_t)h->C * 4);
  return NASR_OK;
}

int nasr_get_decoded(nasr_handle h, int32_t* ids_out, int32_t* lens_out) {
  if (!h || !ids_out || !lens_out) return NASR_ERR_ARG;
  if (!h->have_decoded) return h->fail(NASR_ERR_STATE, "nasr_get_decoded: no decoded step (enable nasr_set_step_decode)");
  HIPCHK(h, hipMemcpyAsync(lens_out, h->lens.p, (size_t)h->B * 4, hipMemcpyDeviceToHost, h->st));
  HIPCHK(h, hipMemcpyAsync(ids_out, h->ids.p, (size_t)h->B * h->Tp * 4, hipMemcpyDeviceToHost, h->st));
  return sync_checked(h);
}

int nasr_set_profiling(nasr_handle h, int enabled) {
  if (!h) return NASR_ERR_ARG;
  h->profiling = enabled != 0;
  h->ev_used = 0;
  h->spans.clear();
  h->window_open = false;
  h->total_valid = false;
  return NASR_OK;
}

int nasr_get_phase_times(nasr_handle h, nasr_phase_times* out) {
  if (!h || !out) return NASR_ERR_ARG;
  HIPCHK(h, hipStreamSynchronize(h->st));
  float acc[PH_COUNT] = {0};
  for (const auto& s : h->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) acc[s.ph] += ms;
  }
  nasr_phase_times t;
  memset(&t, 0, sizeof(t));
  t.pack_ms = acc[PH_PACK]; t.xproj_ms = acc[PH_XPROJ]; t.rec_fwd_ms = acc[PH_RECF]; t.proj_ctc_ms = acc[PH_PROJCTC];
  t.proj_bwd_ms = acc[PH_PROJB]; t.rec_bwd_ms = acc[PH_RECB]; t.wgrad_ms = acc[PH_WGRAD]; t.adam_ms = acc[PH_ADAM];
  float tot = 0.f;
  if (h->total_valid && hipEventElapsedTime(&tot, h->ev_total_a, h->ev_total_b) == hipSuccess) t.total_ms = tot;
  (void)hipGetLastError();
  t.rec_fwd_launches = h->n_fwd_launch;
  t.rec_bwd_launches = h->n_bwd_launch;
  *out = t;
  return NASR_OK;
}

int nasr_set_graph_mode(nasr_handle h, int enabled) {
  if (!h) return NASR_ERR_ARG;
  h->graph_mode = enabled != 0;
  if (!h->graph_mode) drop_graphs(h);
  return NASR_OK;
}

int nasr_set_dropout_state(nasr_handle h, uint32_t seed, uint32_t counter) {
  if (!h) return NASR_ERR_ARG;
  h->drop_seed = seed;
  h->drop_counter = counter;
  return NASR_OK;
}

int nasr_get_dropout_state(nasr_handle h, uint32_t* seed, uint32_t* counter) {
  if (!h) return NASR_ERR_ARG;
  if (seed) *seed = h->drop_seed;
  if (counter) *counter = h->drop_counter;
  return NASR_OK;
}

int nasr_get_recurrence_mode(nasr_handle h) { return !h ? 0 : h->persist ? 1 : h->wide ? 2 : 0; }

int nasr_set_recurrence_mode(nasr_handle h, int persistent) {
  if (!h) return NASR_ERR_ARG;
  if (h->Uw) {   // a wide layer: the wide forward kernel on / off
    HIPCHK(h, hipStreamSynchronize(h->st));
    h->wide = persistent != 0;
    h->wide_wanted = h->wide;
    return repack(h);
  }
  if (persistent && !h->persist_ok)
    return h->fail(NASR_ERR_STATE, "the persistent recurrence is not available on this device / hidden size");
  HIPCHK(h, hipStreamSynchronize(h->st));
  h->persist = persistent != 0;
  h->persist_wanted = h->persist;
  return repack(h);
}

int nasr_set_wgrad_overlap(nasr_handle h, int enabled) {
  if (!h) return NASR_ERR_ARG;
  if (enabled && !h->wst) return h->fail(NASR_ERR_STATE, "the weight-gradient side stream was not set up for this handle "
                                                         "(needs the persistent recurrence at Hp = 512 and more than one layer)");
  HIPCHK(h, hipStreamSynchronize(h->st));
  h->wg_overlap = enabled != 0;
  if (h->resident) return ensure_shape(h, h->B, h->T, h->Lmax);     // the side stream's own copies of the dG planes
  return NASR_OK;
}

int nasr_get_wgrad_overlap(nasr_handle h) { return h && h->wg_overlap ? 1 : 0; }

int nasr_set_bucket_defer(nasr_handle h, int defer) {
  if (!h) return NASR_ERR_ARG;
  h->bucket_defer = defer != 0;
  return NASR_OK;
}

int nasr_get_persist_stats(nasr_handle h, int* aborts, int* rearms) {
  if (!h) return NASR_ERR_ARG;
  if (aborts) *aborts = h->persist_aborts;
  if (rearms) *rearms = h->persist_rearms;
  return NASR_OK;
}

}  // extern "C"
