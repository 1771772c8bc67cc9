// nasr_batch.hip — the batch side of a handle: HBM buffers sized to the batch shape, input validation, and the batch slots
// (synchronous upload, staging through pinned memory on the copy stream, commit).  Replaces the feed_dict of
// TensorFlowNetwork.train (networks/tfnetwork.py:183-190) and DataSet.get_next_batch's hand-over (dataset.py:33-40).
#include "nasr_ctx.h"

using namespace nasr;
using namespace nasr_impl;

namespace nasr_impl {

int ensure_shape(nasr_ctx* h, int B, int T, int Lmax) {
  const int Bp = rup(B, 16);
  const int Tp = nasr_logit_frames(h, T);
  const size_t R = (size_t)T * Bp;
  const int D = h->D, Hp = h->Hp, N4 = h->N4;
  const int KS = std::max(1, (2 * std::max(Lmax, 0) + 1 + 63) / 64);
  if (KS > 16) return h->fail(NASR_ERR_ARG, "label length > 511 not supported by the CTC lattice kernel");
  bool grew = false;
  bool ok = true;
  ok &= h->X0.ensure(R * h->Fp * 4, &grew);
  ok &= h->seqbuf.ensure((size_t)Bp * 4, &grew);
  ok &= h->dout.ensure(R * D * Hp * 4, &grew);
  ok &= h->hstate.ensure((size_t)2 * D * Bp * Hp * 4, &grew);
  ok &= h->partial.ensure((size_t)2 * D * (Hp / 32) * Bp * Hp * 4, &grew);
  ok &= h->dcstate.ensure((size_t)2 * D * Bp * Hp * 4, &grew);
  ok &= h->dgbuf.ensure(R * D * N4 * 4, &grew);
  if (h->Upf) ok &= h->dgmax.ensure(persist_dgmax_floats(T, Bp, Hp, D) * 4, &grew);
  {
    int ipmax = h->Fp, wmax = D * N4;
    for (int l = 0; l < h->L; ++l) ipmax = std::max(ipmax, h->Ip[l]);
    for (int i = 0; i < h->ndense; ++i) { ipmax = std::max(ipmax, h->dIp[i]); wmax = std::max(wmax, h->dWp[i]); }
    ok &= h->XTP.ensure(tph_bytes((int)R, ipmax), &grew);
    ok &= h->X0TTP.ensure(tph_bytes(h->Ip[0], (int)R), &grew);
    for (int l = 0; l < h->L; ++l) ok &= h->OTT[l].ensure(tph_bytes(D * Hp, (int)R), &grew);
    ok &= h->GTP.ensure(tph_bytes((int)R, wmax), &grew);
    ok &= h->GTTP.ensure(tph_bytes(wmax, (int)R), &grew);
    if (h->wg_overlap) {
      ok &= h->GTTP2.ensure(tph_bytes(wmax, (int)R), &grew);
      ok &= h->sc_gc2.ensure((size_t)wmax);
    }
    if (h->ndense) ok &= h->DTP.ensure(tph_bytes(ipmax, (int)R), &grew);
    {
      const size_t n15 = std::max<size_t>(R, (size_t)std::max(ipmax, wmax));
      if (n15 > h->sc15_n) {
        ok &= h->sc15.ensure(n15);
        if (ok) {
          launch_fill(h->sc15.sp(), 32768.f, (int)n15, h->st);
          launch_fill(h->sc15.ip(), 1.f / 32768.f, (int)n15, h->st);
          h->sc15_n = n15;
        }
      }
      ok &= h->sc_x0r.ensure(R) && h->sc_x0c.ensure((size_t)h->Fp);
      ok &= h->sc_gr.ensure(R) && h->sc_gc.ensure((size_t)wmax);
      if (h->compactable) {
        ok &= h->sc_cr.ensure(R + 64) && h->sc_cx.ensure(R + 64);
        ok &= h->OTS.ensure((h->wg_overlap ? 2 : 1) * tph_bytes(D * Hp, (int)R), &grew);
      }
      for (int i = 0; i < h->ndense; ++i) ok &= h->sc_yr[i].ensure(R) && h->sc_yc[i].ensure((size_t)h->dWp[i]);
      bool g2 = false;
      ok &= h->scws.ensure(tph_scale_ws_floats((int)R, std::max(ipmax, wmax)) * 4, &g2);
    }
  }
  ok &= h->logits.ensure((size_t)Tp * Bp * h->Cp * 4, &grew);
  ok &= h->logz.ensure((size_t)Tp * Bp * 4, &grew);
  const int KSa = KS <= 1 ? 2 : KS <= 8 ? KS : (KS <= 12 ? 12 : 16);   // kernel instantiations (two states per lane at least: ctc.hip (2b))
  ok &= h->alpha.ensure((size_t)B * (T + 8) * KSa * 64 * 4, &grew);
  ok &= h->beta.ensure((size_t)B * (T + 8) * KSa * 64 * 4, &grew);
  ok &= h->aoff.ensure((size_t)B * (T + 8) * 8, &grew);
  ok &= h->boff.ensure((size_t)B * (T + 8) * 8, &grew);
  ok &= h->logp.ensure((size_t)Bp * 8, &grew);
  ok &= h->ctcprobs.ensure((size_t)Tp * Bp * h->Cp * 4, &grew);              // emission rows of the CTC lattice (ctc.hip (2b))
  ok &= h->ctckexp.ensure((size_t)B * 2 * ((T + 8) / 4 + 3) * 8, &grew);     // its column offsets per group of frames
  ok &= h->nll.ensure((size_t)Bp * 4, &grew);
  ok &= h->loss.ensure(16, &grew);
  int csw = std::max(D * N4, h->Cp);
  for (int i = 0; i < h->ndense; ++i) csw = std::max(csw, h->dWp[i]);
  // column-sum partials: 32 rows of launch_colsum, or the 64-row partials of the split pass (tp_split2_parts)
  ok &= h->csws.ensure((size_t)std::max(32, tp_split2_parts((int)R)) * csw * 4, &grew);
  if (h->wg_overlap) ok &= h->csws2.ensure((size_t)std::max(32, tp_split2_parts((int)R)) * csw * 4, &grew);
  ok &= h->amax.ensure((size_t)Tp * Bp * 4, &grew);
  ok &= h->ids.ensure((size_t)B * Tp * 4, &grew);
  ok &= h->lens.ensure((size_t)Bp * 4, &grew);
  for (int i = 0; i < h->ndense; ++i) {
    ok &= h->Ybuf[i].ensure(R * h->dWp[i] * 4, &grew);
    ok &= h->dYbuf[i].ensure(R * h->dWp[i] * 4, &grew);
  }
  for (int l = 0; l < h->L; ++l) {
    ok &= h->gates[l].ensure(R * D * N4 * 4, &grew);
    ok &= h->outb[l].ensure(R * D * Hp * 4, &grew);
    ok &= h->cbuf[l].ensure(R * D * Hp * 4, &grew);
  }
  if (!ok) return h->fail(NASR_ERR_HIP, "hipMalloc failed while sizing batch buffers");
  if (grew || Bp != h->Bp) drop_graphs(h);
  h->B = B; h->Bp = Bp; h->T = T; h->Lmax = Lmax; h->Tp = Tp; h->KS = KSa;
  return NASR_OK;
}

int validate_batch(nasr_ctx* h, const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B, int T,
                   int Lmax) {
  if (B < 1 || B > 64) return h->fail(NASR_ERR_ARG, "per-GPU batch must be in [1,64]");
  if (T < 1) return h->fail(NASR_ERR_ARG, "T must be >= 1");
  for (int b = 0; b < B; ++b) {
    if (seq_len[b] < 1 || seq_len[b] > T)
      return h->fail(NASR_ERR_ARG, "seq_len[" + std::to_string(b) + "] out of [1,T]");
    if (!labels) continue;
    const int L = label_len[b];
    if (L < 0 || L > Lmax) return h->fail(NASR_ERR_ARG, "label_len[" + std::to_string(b) + "] out of [0,Lmax]");
    int rep = 0;
    for (int i = 0; i < L; ++i) {
      const int v = labels[(size_t)b * Lmax + i];
      if (v < 0 || v >= h->C - 1)
        return h->fail(NASR_ERR_ARG, "label id out of [0, num_classes-2] (blank = num_classes-1 is not a label)");
      if (i > 0 && v == labels[(size_t)b * Lmax + i - 1]) ++rep;
    }
    if (L + rep > seq_len[b])
      return h->fail(NASR_ERR_INFEASIBLE, "Not enough time for target transition sequence (required: " +
                                              std::to_string(L + rep) + ", available: " + std::to_string(seq_len[b]) +
                                              ") in sequence " + std::to_string(b));
  }
  return NASR_OK;
}

bool pinned_ensure(void** p, size_t* cap, size_t bytes) {
  if (bytes <= *cap) return true;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  const size_t want = bytes + bytes / 8;
  if (hipHostMalloc(p, want, hipHostMallocMapped) != hipSuccess) return false;   // (kernels write step results into it)
  *cap = want;
  return true;
}

// Takes a free slot (round robin), marks it FILLING.  NULL when every slot holds a staged or the resident batch.
BatchSlot* slot_acquire(nasr_ctx* h, bool for_stage) {
  std::lock_guard<std::mutex> lk(h->slot_mu);
  if (for_stage) {   // staged batches never take the slot a synchronous upload (validate, decode, ...) needs
    int ahead = 0;
    for (const BatchSlot& s : h->slots) ahead += s.state == SLOT_STAGED || s.state == SLOT_FILLING;
    if (ahead >= NSTAGE) return nullptr;
  }
  for (int k = 0; k < NSLOT; ++k) {
    BatchSlot& s = h->slots[(h->slot_rr + k) % NSLOT];
    if (s.state == SLOT_FREE) {
      h->slot_rr = (h->slot_rr + k + 1) % NSLOT;
      s.state = SLOT_FILLING;
      s.gen += 1;
      return &s;
    }
  }
  return nullptr;
}

void slot_set_state(nasr_ctx* h, BatchSlot* s, int st) {
  std::lock_guard<std::mutex> lk(h->slot_mu);
  s->state = st;
}

// Copies one batch into slot s: the integer arrays through the slot's pinned meta buffer, the features from the caller's
// memory (`pinned_feats` false: hipMemcpyAsync from pageable memory, which returns when the source may be reused) or
// through the slot's pinned feature buffer (true: the H2D is a plain DMA that overlaps whatever the compute stream runs).
// All device copies go to stream cs and end with the slot's ev_copy.
int slot_fill(nasr_ctx* h, BatchSlot* s, const float* feats, const int32_t* seq_len, const int32_t* labels,
              const int32_t* label_len, int B, int T, int Lmax, const float* centre, const float* pad_value, int ctx,
              int ncep, hipStream_t cs, bool pinned_feats) {
  if ((!feats && !centre) || !seq_len) return h->fail(NASR_ERR_ARG, "null input buffer");
  if (centre && (!pad_value || ctx < 0 || ncep < 1 || (2 * ctx + 1) * ncep != h->F))
    return h->fail(NASR_ERR_ARG, "context upload: feature_size must equal (2*numcontext+1)*numcep");
  if (labels && !label_len) return h->fail(NASR_ERR_ARG, "labels without label_len");
  int rc = validate_batch(h, seq_len, labels, label_len, B, T, Lmax);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  const int Bp = rup(B, 16), Tp = nasr_logit_frames(h, T), C = h->C, Lm = std::max(labels ? Lmax : 0, 1);
  const bool sr = h->cfg.merge == NASR_MERGE_STACK_RESHAPE && h->D == 2;
  // meta layout (int32): seq [Bp] | lablen [Bp] | labels [B*Lm] | cstart [B*(C+1)] | cpos [B*Lm] | rowmap [Tp*Bp]
  s->o_seq = 0;
  s->o_lablen = s->o_seq + Bp;
  s->o_labels = s->o_lablen + Bp;
  s->o_cstart = s->o_labels + (size_t)B * Lm;
  s->o_cpos = s->o_cstart + (labels ? (size_t)B * (C + 1) : 0);
  s->o_rowmap = s->o_cpos + (size_t)B * Lm;
  // compacted rows: worth it when at least 15 % of the T x Bp rows are padding (profiles/r03_row_compaction_ab.log: even at
  // 10 %), and only for training batches
  int64_t rv = 0;
  for (int b = 0; b < B; ++b) rv += seq_len[b];
  s->cmp = h->compactable && labels && rv * 20 <= (int64_t)T * Bp * 17;
  s->Rv = s->cmp ? (int)rv : 0;
  s->Rvp = s->cmp ? rup((int)rv, 64) : 0;
  s->o_vrow = s->o_rowmap + (sr ? (size_t)Tp * Bp : 0);
  s->o_vprev = s->o_vrow + s->Rvp;
  s->o_vnext = s->o_vprev + s->Rvp;
  const size_t nmeta = s->o_vnext + s->Rvp;
  const size_t nfeat = centre ? (size_t)B * T * ncep + B : (size_t)B * T * h->F;
  bool grew = false;
  if (!s->dmeta.ensure(nmeta * 4, &grew) || !s->dfeats.ensure(nfeat * 4, &grew) ||
      !pinned_ensure(&s->hmeta, &s->hmeta_cap, nmeta * 4) ||
      (pinned_feats && !pinned_ensure(&s->hfeats, &s->hfeats_cap, nfeat * 4)))
    return h->fail(NASR_ERR_HIP, "allocation of a batch slot failed");
  if (s->copy_valid) HIPCHK(h, hipEventSynchronize(s->ev_copy));          // the pinned mirrors are free to overwrite
  if (s->released_valid && cs != h->st) HIPCHK(h, hipStreamWaitEvent(cs, s->ev_released, 0));   // and the device side unread
  int32_t* m = static_cast<int32_t*>(s->hmeta);
  memset(m, 0, nmeta * 4);
  s->frames = 0;
  for (int b = 0; b < B; ++b) {
    m[s->o_seq + b] = seq_len[b];
    s->frames += seq_len[b];
  }
  if (labels) {
    for (int b = 0; b < B; ++b) m[s->o_lablen + b] = label_len[b];
    if (Lmax > 0) memcpy(m + s->o_labels, labels, (size_t)B * Lmax * 4);
    // the label positions of every utterance sorted by class (counting sort): the fixed summation order of ctc_grad
    std::vector<int32_t> fill((size_t)C);
    for (int b = 0; b < B; ++b) {
      int32_t* c0 = m + s->o_cstart + (size_t)b * (C + 1);
      const int32_t* lb = labels + (size_t)b * Lmax;
      for (int i = 0; i < label_len[b]; ++i) c0[lb[i] + 1] += 1;
      for (int c = 0; c < C; ++c) c0[c + 1] += c0[c];
      std::copy(c0, c0 + C, fill.begin());
      for (int i = 0; i < label_len[b]; ++i) m[s->o_cpos + (size_t)b * Lm + fill[lb[i]]++] = i;
    }
  }
  if (sr) {
    // logits[t',b'] <- flat row q = b'*2T + t' of O = stack(fw,bw) [2,B,T,H];
    // physical row index in the [(t*Bp+b)*2 + d][Hp] view of the last layer's output.
    int32_t* map = m + s->o_rowmap;
    for (size_t i = 0; i < (size_t)Tp * Bp; ++i) map[i] = -1;
    for (int tp = 0; tp < Tp; ++tp)
      for (int bq = 0; bq < B; ++bq) {
        const int64_t q = (int64_t)bq * 2 * T + tp;
        const int d = (int)(q / ((int64_t)B * T));
        const int64_t rem = q % ((int64_t)B * T);
        const int b = (int)(rem / T), t = (int)(rem % T);
        map[(size_t)tp * Bp + bq] = (t * Bp + b) * 2 + d;
      }
  }
  if (s->cmp) {
    int32_t *vr = m + s->o_vrow, *vp = m + s->o_vprev, *vn = m + s->o_vnext;
    int i = 0;
    for (int t = 0; t < T; ++t)
      for (int b = 0; b < B; ++b)
        if (seq_len[b] > t) {
          vr[i] = t * Bp + b;
          vp[i] = t > 0 ? (t - 1) * Bp + b : -1;
          vn[i] = t + 1 < seq_len[b] ? (t + 1) * Bp + b : -1;
          ++i;
        }
    for (; i < s->Rvp; ++i) vr[i] = vp[i] = vn[i] = -1;
  }
  if (centre) {
    const size_t nc = (size_t)B * T * ncep;
    if (pinned_feats) {
      memcpy(s->hfeats, centre, nc * 4);
      memcpy(static_cast<float*>(s->hfeats) + nc, pad_value, (size_t)B * 4);
      HIPCHK(h, hipMemcpyAsync(s->dfeats.p, s->hfeats, (nc + B) * 4, hipMemcpyHostToDevice, cs));
    } else {
      HIPCHK(h, hipMemcpyAsync(s->dfeats.p, centre, nc * 4, hipMemcpyHostToDevice, cs));
      HIPCHK(h, hipMemcpyAsync(s->dfeats.as<float>() + nc, pad_value, (size_t)B * 4, hipMemcpyHostToDevice, cs));
    }
  } else if (pinned_feats) {
    memcpy(s->hfeats, feats, nfeat * 4);
    HIPCHK(h, hipMemcpyAsync(s->dfeats.p, s->hfeats, nfeat * 4, hipMemcpyHostToDevice, cs));
  } else {
    HIPCHK(h, hipMemcpyAsync(s->dfeats.p, feats, nfeat * 4, hipMemcpyHostToDevice, cs));
  }
  HIPCHK(h, hipMemcpyAsync(s->dmeta.p, s->hmeta, nmeta * 4, hipMemcpyHostToDevice, cs));
  HIPCHK(h, hipEventRecord(s->ev_copy, cs));
  s->copy_valid = true;
  s->B = B; s->T = T; s->Lmax = labels ? Lmax : 0; s->Bp = Bp; s->Tp = Tp; s->ctx = ctx; s->ncep = ncep;
  s->has_labels = labels != nullptr;
  s->centre = centre != nullptr;
  return NASR_OK;
}

// Makes the filled slot the resident batch: the compute stream waits for its copies, the previous resident slot is
// released, and the features are laid out for the step (time-major rows, context windows, operand scales).
int slot_commit(nasr_ctx* h, BatchSlot* s) {
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_shape(h, s->B, s->T, s->Lmax);
  if (rc) return rc;
  const int B = s->B, T = s->T, Bp = h->Bp;
  {
    std::lock_guard<std::mutex> lk(h->slot_mu);
    if (h->cur && h->cur != s) {
      // every kernel that reads the old batch's arrays is already on the compute stream: an event here releases them
      (void)hipEventRecord(h->cur->ev_released, h->st);
      h->cur->released_valid = true;
      h->cur->state = SLOT_FREE;
    }
    s->state = SLOT_RESIDENT;
    h->cur = s;
  }
  HIPCHK(h, hipStreamWaitEvent(h->st, s->ev_copy, 0));
  int32_t* md = s->meta_d();
  // seq_len lives at a FIXED address: the hipGraphs of the per-step recurrence captured it
  HIPCHK(h, hipMemcpyAsync(h->seqbuf.p, md + s->o_seq, (size_t)Bp * 4, hipMemcpyDeviceToDevice, h->st));
  h->seq_p = h->seqbuf.as<int32_t>(); h->lablen_p = md + s->o_lablen; h->labels_p = md + s->o_labels;
  h->cstart_p = md + s->o_cstart; h->cpos_p = md + s->o_cpos; h->rowmap_p = md + s->o_rowmap;
  const bool cmp = s->cmp && h->compactable;      // (staged with it on, switched off since: the slot's row lists go unused)
  h->cmp_rows = cmp ? s->Rv : 0;
  h->cmp_rows_p = cmp ? s->Rvp : 0;
  h->vrow_p = md + s->o_vrow; h->vprev_p = md + s->o_vprev; h->vnext_p = md + s->o_vnext;
  h->ev_used = 0;
  h->spans.clear();
  if (h->profiling) {
    (void)hipEventRecord(h->ev_total_a, h->st);
    h->window_open = true;
    h->total_valid = false;
  }
  h->h_seq.assign((size_t)Bp, 0);
  const int32_t* hm = static_cast<const int32_t*>(s->hmeta);
  for (int b = 0; b < B; ++b) h->h_seq[b] = hm[s->o_seq + b];
  h->frames = s->frames;
  {
    PhaseScope ps(h, PH_PACK);
    if (s->centre)
      launch_expand_context(s->dfeats.as<float>(), s->dfeats.as<float>() + (size_t)B * T * s->ncep, h->seq_p,
                            h->X0.as<float>(), B, Bp, T, s->ctx, s->ncep, h->Fp, h->st);
    else
      launch_pack_feats(s->dfeats.as<float>(), h->X0.as<float>(), B, Bp, T, h->F, h->Fp, h->st);
    pl_scales(h, h->X0.as<float>(), T * Bp, h->Fp, h->Fp, &h->sc_x0r, &h->sc_x0c, h->st);
    if (h->cmp_rows)    // the feature rows' scales in the compacted order (layer 0's input GEMM)
      launch_gather_rows(h->sc_cx.sp(), h->sc_x0r.sp(), h->vrow_p, h->cmp_rows_p, 1.f, h->st),
      launch_gather_rows(h->sc_cx.ip(), h->sc_x0r.ip(), h->vrow_p, h->cmp_rows_p, 1.f, h->st);
    if (s->has_labels && h->npre == 0)   // layer-0 input with the frame index as contraction index, for dWx = X^T dG
      launch_tph_split2(h->X0.as<float>(), nullptr, h->X0TTP.as<unsigned char>(), h->cmp_rows ? h->cmp_rows : T * Bp, h->Fp, h->Fp,
                        nullptr, 1.f, h->sc_x0c.sp(), 1.f, nullptr, h->st, h->cmp_rows ? h->vrow_p : nullptr);
    HIPCHK(h, hipGetLastError());
  }
  h->resident = true;
  h->have_grads = false;
  h->have_fwd = false;
  h->have_decoded = false;
  return NASR_OK;
}

// the synchronous upload of nasr_upload_batch / nasr_train_step / ...: fill on the compute stream, commit
int upload(nasr_ctx* h, const float* feats, const int32_t* seq_len, const int32_t* labels, const int32_t* label_len,
           int B, int T, int Lmax, const float* centre, const float* pad_value, int ctx, int ncep) {
  BatchSlot* s = slot_acquire(h, false);
  if (!s) return h->fail(NASR_ERR_STATE, "every batch slot holds a staged batch: commit or discard one first");
  int rc = slot_fill(h, s, feats, seq_len, labels, label_len, B, T, Lmax, centre, pad_value, ctx, ncep, h->st, false);
  if (!rc) rc = slot_commit(h, s);
  if (rc && h->cur != s) slot_set_state(h, s, SLOT_FREE);
  return rc;
}

BatchSlot* slot_of_ticket(nasr_ctx* h, int ticket) {
  if (ticket < 0 || (ticket & 255) >= NSLOT) return nullptr;
  BatchSlot* s = &h->slots[ticket & 255];
  std::lock_guard<std::mutex> lk(h->slot_mu);
  return (s->state == SLOT_STAGED && (int)(s->gen & 0x7fffff) == (ticket >> 8)) ? s : nullptr;
}

int stage(nasr_ctx* h, const float* feats, const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B,
          int T, int Lmax, const float* centre, const float* pad_value, int ctx, int ncep, int* ticket) {
  if (!ticket) return h->fail(NASR_ERR_ARG, "null ticket");
  *ticket = -1;
  BatchSlot* s = slot_acquire(h, true);
  if (!s) return h->fail(NASR_ERR_STATE, "no free batch slot: commit or discard a staged batch first");
  const int rc = slot_fill(h, s, feats, seq_len, labels, label_len, B, T, Lmax, centre, pad_value, ctx, ncep, h->cst, true);
  if (rc) {
    slot_set_state(h, s, SLOT_FREE);
    return rc;
  }
  slot_set_state(h, s, SLOT_STAGED);
  *ticket = (int)(s - h->slots) | (int)((s->gen & 0x7fffff) << 8);
  return NASR_OK;
}
}  // namespace nasr_impl
