// nasr_layout.hip — parameter layout of a handle (TF variable order <-> the padded internal buffers), the operand images the
// kernels keep of the weights (repack), host <-> device parameter copies, and the management of the persistent recurrence
// mode (abort handling, census, re-arming).  See include/nasr.h and DESIGN.md §3.
#include "nasr_ctx.h"

using namespace nasr;
using namespace nasr_impl;

namespace nasr_impl {

std::string g_create_error;
thread_local std::string t_err;
thread_local const void* t_err_handle = nullptr;

// scales of src [rows][K]: per row into `row`, per column into `col` (either may be NULL)
void pl_scales(nasr_ctx* h, const float* src, int rows, int K, int ld, nasr_ctx::SV* row, nasr_ctx::SV* col, hipStream_t st) {
  launch_tph_scales(src, rows, K, ld, row ? row->sp() : nullptr, row ? row->ip() : nullptr, col ? col->sp() : nullptr,
                    col ? col->ip() : nullptr, h->scws.as<float>(), st);
}
// planes of src [rows][K] (tpN, scaled per row by rs[]) and / or of its transpose (tpT, scaled per src column by cs[]);
// colpart: 64-row partial column sums for launch_colsum_parts
void pl_split(const float* src, unsigned char* tpN, unsigned char* tpT, int rows, int K, int ld, const float* rs,
              const float* cs, float* colpart, hipStream_t st) {
  launch_tph_split2(src, tpN, tpT, rows, K, ld, rs, 1.f, cs, 1.f, colpart, st);
}
// a_inv / b_inv: inverse scales of A's / B's rows; the strides apply to batch 1 of a two-batch launch
void pl_gemm(GemmTPHDesc g, const float* a_inv, const float* b_inv, hipStream_t st, int64_t ainv_bstride, int64_t binv_bstride) {
  g.a_inv = a_inv; g.b_inv = b_inv; g.ainv_bstride = ainv_bstride; g.binv_bstride = binv_bstride;
  launch_gemm_tph(g, st);
}
int persist_check(nasr_ctx* h) {
  if (!h->persist_used) return NASR_OK;
  h->persist_used = false;
  const unsigned code = *reinterpret_cast<volatile unsigned*>(h->perr);
  if (!code) return NASR_OK;
  *reinterpret_cast<volatile unsigned*>(h->perr) = 0;
  h->persist = false;
  h->persist_ok = false;
  h->wide = false;
  h->persist_aborts += 1;
  h->clean_steps = 0;
  h->rearm_wait = h->persist_aborts <= 1 ? h->rearm_after : std::min<int64_t>(h->rearm_wait * 2, (int64_t)1 << 20);
  (void)repack(h);   // operand images of the per-step kernels
  return h->fail(NASR_ERR_HIP, "persistent recurrence aborted (code " + std::to_string(code) +
                                   ": 1 = hand-off timeout, 2 = workgroup placement, 4 = dG beyond its fp16 planes); the results of this step are "
                                   "invalid, later steps use the per-step kernels" +
                                   (h->rearm_wait > 0 ? " (the persistent kernels are tried again after " +
                                                            std::to_string(h->rearm_wait) + " clean steps)"
                                                      : ""));
}

// Census: two steps of both persistent kernels on a zero layer.  A chip that does not place 32 workgroups on each of
// its 8 XCDs (partition modes, masked CUs, a co-tenant) is detected here and served by the per-step kernels.
// Synchronises the stream.
bool persist_census(nasr_ctx* h) {
  const int Bp = 16, T = 2;
  const size_t R = (size_t)T * Bp;
  DevBuf g, c, o, dg, sq;
  bool grew = false;
  bool ok = g.ensure(R * h->D * h->N4 * 4, &grew) && c.ensure(R * h->D * h->Hp * 4, &grew) &&
            o.ensure(R * h->D * h->Hp * 4, &grew) && dg.ensure(R * h->D * h->N4 * 4, &grew) && sq.ensure(Bp * 4, &grew);
  if (ok) {
    (void)hipMemsetAsync(g.p, 0, R * h->D * h->N4 * 4, h->st);
    (void)hipMemsetAsync(o.p, 0, R * h->D * h->Hp * 4, h->st);
    std::vector<int32_t> two((size_t)Bp, T);
    (void)hipMemcpyAsync(sq.p, two.data(), Bp * 4, hipMemcpyHostToDevice, h->st);
    const LstmDims dm{T, Bp, Bp, h->H, h->Hp, h->D};
    launch_lstm_persist_fwd(dm, h->Upf, h->rec_f16 ? h->Ucinv : nullptr, g.as<float>(), c.as<float>(), o.as<float>(),
                            sq.as<int>(), h->xchf, h->pctl, h->perr, nullptr, 1.f, h->st);
    launch_lstm_persist_bwd(dm, h->Upb, g.as<float>(), dg.as<float>(), c.as<float>(), o.as<float>(), sq.as<int>(),
                            h->xchb, h->pctl, h->perr, nullptr, h->st);
    ok = hipStreamSynchronize(h->st) == hipSuccess && hipGetLastError() == hipSuccess && *h->perr == 0;
  }
  for (DevBuf* b : {&g, &c, &o, &dg, &sq}) b->release();
  *h->perr = 0;
  return ok;
}

// After `rearm_wait` clean steps on the per-step kernels: run the census again and go back to the persistent kernels
// (called at the start of a step, before anything of it is enqueued).
void persist_rearm(nasr_ctx* h) {
  if (h->wide_wanted && !h->wide && h->persist_aborts > 0 && h->rearm_wait > 0) {
    // the wide forward kernel has no census launch of its own: its next launch is the census (a second abort voids that
    // step, which the caller repeats on the per-step kernels, and doubles the wait)
    if (++h->clean_steps <= h->rearm_wait) return;
    h->clean_steps = 0;
    if (hipStreamSynchronize(h->st) != hipSuccess) return;
    h->wide = true;
    if (repack(h) != NASR_OK) { h->wide = false; return; }
    h->persist_rearms += 1;
    return;
  }
  if (h->persist || !h->persist_wanted || h->persist_aborts == 0 || h->rearm_wait <= 0 || !h->Upf) return;
  if (++h->clean_steps <= h->rearm_wait) return;   // `rearm_wait` whole steps ran on the per-step kernels since the abort
  h->clean_steps = 0;
  if (hipStreamSynchronize(h->st) != hipSuccess) return;
  // the operand images of the persistent kernels are stale (repack() only maintains the mode in use): rebuild first
  h->persist = true;
  if (repack(h) != NASR_OK || !persist_census(h)) {
    h->persist = false;
    h->rearm_wait = std::min<int64_t>(h->rearm_wait * 2, (int64_t)1 << 20);
    (void)repack(h);
    return;
  }
  h->persist_ok = true;
  h->persist_rearms += 1;
  drop_graphs(h);
}
// A word in host-mapped pinned memory, written by a one-thread kernel in stream order (system-scope store): the host
// learns that everything enqueued before it has happened by READING MEMORY - no runtime call, no event.  (Waiting on a HIP
// event recorded a whole step earlier cost 0.4-0.75 ms per call here although the event had long fired.)
__global__ void stamp_kernel(unsigned* dst, unsigned value, float* f0_dst, const float* f0_src) {
  if (f0_dst) __hip_atomic_store(f0_dst, *f0_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(dst, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_stamp(unsigned* dst, unsigned value, float* f0_dst, const float* f0_src, hipStream_t st) {
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, st, dst, value, f0_dst, f0_src);
}

// The values a step returns to the host (loss, fault word, greedy decode), written by ONE kernel straight into the host's
// pinned result buffer and stamped behind them - instead of four small device-to-host copies and a stamp launch per step.
// host: [loss f32][fault f32][lens i32 x Bp][ids i32 x n_ids]
__global__ __launch_bounds__(256) void publish_results_kernel(const float* __restrict__ loss, const float* __restrict__ fault,
                                                              const int* __restrict__ lens, int Bp, const int* __restrict__ ids,
                                                              int n_ids, unsigned* host, unsigned* stamp, unsigned value) {
  if (threadIdx.x == 0) {
    host[0] = __float_as_uint(*loss);
    host[1] = __float_as_uint(*fault);
  }
  for (int i = threadIdx.x; i < Bp; i += 256) host[2 + i] = (unsigned)lens[i];
  for (int i = threadIdx.x; i < n_ids; i += 256) host[2 + Bp + i] = (unsigned)ids[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(stamp, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_publish_results(const float* loss, const float* fault, const int* lens, int Bp, const int* ids, int n_ids, void* host,
                            unsigned* stamp, unsigned value, hipStream_t st) {
  hipLaunchKernelGGL(publish_results_kernel, dim3(1), dim3(256), 0, st, loss, fault, lens, Bp, ids, n_ids,
                     static_cast<unsigned*>(host), stamp, value);
}

bool wait_stamp(const uint32_t* w, uint32_t want, double timeout_s) {
  const volatile uint32_t* v = w;
  for (int i = 0; i < 4000; ++i)
    if (*v == want) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 0;; ++n) {
    if (*v == want) return true;
    if ((n & 63) == 63) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
      usleep(20);
    } else {
      sched_yield();
    }
  }
}

int sync_checked(nasr_ctx* h) {
  HIPCHK(h, hipStreamSynchronize(h->st));
  return persist_check(h);
}

void drop_graphs(nasr_ctx* h) {
  for (auto& kv : h->graphs) (void)hipGraphExecDestroy(kv.second);
  h->graphs.clear();
}

hipEvent_t next_event(nasr_ctx* h) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    h->ev_pool.push_back(e);
  }
  return h->ev_pool[h->ev_used++];
}
// ---- model layout ---------------------------------------------------------------------------
int build_layout(nasr_ctx* h) {
  const nasr_model_cfg& c = h->cfg;
  h->F = c.feature_size;
  h->H = c.hidden;
  h->L = c.num_layers;
  h->D = c.bidirectional ? 2 : 1;
  h->C = c.num_classes;
  h->Fp = rup(h->F, 32);
  h->Hp = rup(h->H, 64);
  h->N4 = 4 * h->Hp;
  h->Cp = rup(h->C, 32);
  const bool concat = c.bidirectional && c.merge == NASR_MERGE_CONCAT;
  const int D = h->D, Hp = h->Hp, N4 = h->N4, H = h->H;
  const int lstm_out = concat ? 2 * H : H, lstm_outp = concat ? 2 * Hp : Hp;

  // dense stages
  h->npre = c.num_pre;
  h->has_post = c.post_width > 0;
  h->ndense = h->npre + (h->has_post ? 1 : 0);
  h->dWid.assign(h->ndense, 0); h->dWp.assign(h->ndense, 0); h->dIn.assign(h->ndense, 0); h->dIp.assign(h->ndense, 0);
  for (int i = 0; i < h->npre; ++i) {
    h->dWid[i] = c.pre_width[i]; h->dWp[i] = rup(c.pre_width[i], 64);
    h->dIn[i] = i == 0 ? h->F : h->dWid[i - 1];
    h->dIp[i] = i == 0 ? h->Fp : h->dWp[i - 1];
  }
  if (h->has_post) {
    const int i = h->npre;
    h->dWid[i] = c.post_width; h->dWp[i] = rup(c.post_width, 64);
    h->dIn[i] = lstm_out; h->dIp[i] = lstm_outp;
  }
  h->F0 = h->npre ? h->dWid[h->npre - 1] : h->F;
  h->Pin = h->has_post ? c.post_width : lstm_out;
  h->Pinp = h->has_post ? h->dWp[h->npre] : lstm_outp;

  int64_t off = 0;
  h->off_dw.assign(h->ndense, 0); h->off_db.assign(h->ndense, 0);
  for (int i = 0; i < h->ndense; ++i) {
    h->off_dw[i] = off; off += (int64_t)h->dIp[i] * h->dWp[i];
    h->off_db[i] = off; off += h->dWp[i];
  }
  h->Ip.resize(h->L);
  h->off_wx.resize(h->L);
  h->off_bias.resize(h->L);
  h->off_u.resize((size_t)h->L * D);
  for (int l = 0; l < h->L; ++l) {
    h->Ip[l] = l == 0 ? (h->npre ? h->dWp[h->npre - 1] : h->Fp) : D * Hp;
    h->off_wx[l] = off;
    off += (int64_t)h->Ip[l] * D * N4;
    h->off_bias[l] = off;
    off += (int64_t)D * N4;
    for (int d = 0; d < D; ++d) {
      h->off_u[(size_t)l * D + d] = off;
      off += (int64_t)Hp * N4;
    }
  }
  h->off_w = off;
  off += (int64_t)h->Pinp * h->Cp;
  h->off_b = off;
  off += h->Cp;
  h->np_int = off;  // every term is a multiple of 32
  if (off >= (int64_t)1 << 31) return h->fail(NASR_ERR_ARG, "model too large for 32-bit parameter indexing");

  // TF variable order + element map.  Plain (Bi)LstmCTCNet: cells, W, b.  DeepSpeech family (creation order of
  // networks/deepspeech.py): b1,h1,b2,h2,b3,h3, cells, b5,h5, b6,h6.
  const bool ds = h->ndense > 0;
  h->tensors.clear();
  int64_t tfo = 0;
  auto add = [&](const std::string& n, int64_t r, int64_t cc) {
    h->tensors.push_back({n, tfo, r, cc});
    tfo += r * cc;
  };
  for (int i = 0; i < h->npre; ++i) {
    add("b" + std::to_string(i + 1), h->dWid[i], 1);
    add("h" + std::to_string(i + 1), h->dIn[i], h->dWid[i]);
  }
  for (int l = 0; l < h->L; ++l) {
    const int I = l == 0 ? h->F0 : D * H;
    for (int d = 0; d < D; ++d) {
      std::string pre = "l" + std::to_string(l) + "/";
      if (D == 2) pre += d == 0 ? "fw/" : "bw/";
      add(pre + "kernel", I + H, 4 * H);
      add(pre + "bias", 4 * H, 1);
    }
  }
  if (ds) {
    if (h->has_post) {
      add("b5", h->dWid[h->npre], 1);
      add("h5", h->dIn[h->npre], h->dWid[h->npre]);
    }
    add("b6", h->C, 1);
    add("h6", h->Pin, h->C);
  } else {
    add("W", h->Pin, h->C);
    add("b", h->C, 1);
  }
  h->np_tf = tfo;
  h->tf2int.assign((size_t)tfo, 0);
  size_t ti = 0;
  // rows of a matrix fed by the concatenated (fw, bw) outputs: the bw half starts at the padded width
  auto cat_row = [&](int r) { return (D == 2 && concat && r >= H) ? Hp + (r - H) : r; };
  auto map_dense = [&](int i, bool from_lstm) {
    const TensorInfo& tb = h->tensors[ti++];
    for (int cc = 0; cc < h->dWid[i]; ++cc) h->tf2int[(size_t)(tb.offset + cc)] = (int32_t)(h->off_db[i] + cc);
    const TensorInfo& tw = h->tensors[ti++];
    for (int r = 0; r < h->dIn[i]; ++r) {
      const int ir = from_lstm ? cat_row(r) : r;
      for (int cc = 0; cc < h->dWid[i]; ++cc)
        h->tf2int[(size_t)(tw.offset + (int64_t)r * h->dWid[i] + cc)] = (int32_t)(h->off_dw[i] + (int64_t)ir * h->dWp[i] + cc);
    }
  };
  for (int i = 0; i < h->npre; ++i) map_dense(i, false);
  for (int l = 0; l < h->L; ++l) {
    const int I = l == 0 ? h->F0 : D * H;
    for (int d = 0; d < D; ++d) {
      const TensorInfo& tk = h->tensors[ti++];
      for (int r = 0; r < I + H; ++r) {
        for (int cc = 0; cc < 4 * H; ++cc) {
          const int g = cc / H, j = cc % H;
          int64_t dst;
          if (r < I) {
            int ir = r;
            if (l > 0 && D == 2 && r >= H) ir = Hp + (r - H);
            dst = h->off_wx[l] + (int64_t)ir * D * N4 + d * N4 + 4 * j + g;
          } else {
            dst = h->off_u[(size_t)l * D + d] + (int64_t)(r - I) * N4 + 4 * j + g;
          }
          h->tf2int[(size_t)(tk.offset + (int64_t)r * 4 * H + cc)] = (int32_t)dst;
        }
      }
      const TensorInfo& tb = h->tensors[ti++];
      for (int cc = 0; cc < 4 * H; ++cc) {
        const int g = cc / H, j = cc % H;
        h->tf2int[(size_t)(tb.offset + cc)] = (int32_t)(h->off_bias[l] + d * N4 + 4 * j + g);
      }
    }
  }
  if (h->has_post) map_dense(h->npre, true);
  auto map_w = [&]() {
    const TensorInfo& tw = h->tensors[ti++];
    for (int r = 0; r < h->Pin; ++r) {
      const int ir = h->has_post ? r : cat_row(r);
      for (int cc = 0; cc < h->C; ++cc)
        h->tf2int[(size_t)(tw.offset + (int64_t)r * h->C + cc)] = (int32_t)(h->off_w + (int64_t)ir * h->Cp + cc);
    }
  };
  auto map_b = [&]() {
    const TensorInfo& tb = h->tensors[ti++];
    for (int cc = 0; cc < h->C; ++cc) h->tf2int[(size_t)(tb.offset + cc)] = (int32_t)(h->off_b + cc);
  };
  if (ds) { map_b(); map_w(); } else { map_w(); map_b(); }
  return NASR_OK;
}

int repack(nasr_ctx* h) {
  // only the operand images of the kernels in use (a mode switch calls repack again)
  // scales of every matrix that needs them - recurrent matrices of the persistent / wide kernels, input and dense
  // weights of the plane GEMMs - in ONE batch (two launches), then the images
  std::vector<TphScaleJob> jobs;
  if (h->persist && h->rec_f16)
    for (size_t k = 0; k < h->off_u.size(); ++k)
      jobs.push_back({h->P + h->off_u[k], h->Hp, h->N4, h->N4, nullptr, nullptr, h->Ucs + k * h->N4, h->Ucinv + k * h->N4});
  if (!h->persist && h->wide)
    for (size_t k = 0; k < h->off_u.size(); ++k)
      jobs.push_back({h->P + h->off_u[k], h->Hp, h->N4, h->N4, h->Urs + k * h->Hp, h->Urinv + k * h->Hp,
                      h->Ucs + k * h->N4, h->Ucinv + k * h->N4});
  for (int l = 0; l < h->L; ++l) {
    const bool back = l > 0 || h->npre > 0;
    jobs.push_back({h->P + h->off_wx[l], h->Ip[l], h->D * h->N4, h->D * h->N4, back ? h->sc_wr[l].sp() : nullptr,
                    back ? h->sc_wr[l].ip() : nullptr, h->sc_wc[l].sp(), h->sc_wc[l].ip()});
  }
  for (int i = 0; i < h->ndense; ++i) {
    const bool back = i > 0 || h->npre == 0;
    jobs.push_back({h->P + h->off_dw[i], h->dIp[i], h->dWp[i], h->dWp[i], back ? h->sc_dr[i].sp() : nullptr,
                    back ? h->sc_dr[i].ip() : nullptr, h->sc_dc[i].sp(), h->sc_dc[i].ip()});
  }
  {
    bool g2 = false;
    if (!h->scws.ensure(tph_scale_batch_ws_floats(jobs.data(), (int)jobs.size()) * 4, &g2))
      return h->fail(NASR_ERR_HIP, "allocation of the scale workspace failed");
  }
  launch_tph_scales_batch(jobs.data(), (int)jobs.size(), h->scws.as<float>(), h->st);
  if (h->persist) {
    launch_repack_persist(h->P, h->off_u.data(), (int)h->off_u.size(), h->Upf, h->Upb, h->Hp,
                          h->rec_f16 ? h->Ucs : nullptr, h->st);
  } else {
    if (!h->wide)   // (a fall-back from the wide kernels calls repack again: persist_check)
      for (int l = 0; l < h->L; ++l)
        for (int d = 0; d < h->D; ++d) {
          const size_t k = (size_t)l * h->D + d;
          const size_t o = k * (size_t)h->Hp * h->N4;
          launch_repack_u(h->P + h->off_u[k], h->Uf + o, h->Ub + o, h->Hp, h->st);
        }
    if (h->wide)    // the fp16-plane images of the wide kernels
      for (size_t k = 0; k < h->off_u.size(); ++k) {
        launch_repack_wide(h->P + h->off_u[k], h->Ucs + k * h->N4, h->Uw + k * wide_image_bytes(h->Hp), h->Hp, h->st);
        launch_repack_wide_bwd(h->P + h->off_u[k], h->Urs + k * h->Hp, h->Uwb + k * wide_image_bytes(h->Hp), h->Hp, h->st);
      }
  }
  for (int l = 0; l < h->L; ++l) {
    // forward operand = planes of Wx^T, input-gradient operand = planes of Wx: one pass where both are needed
    const bool back = l > 0 || h->npre > 0;
    const float* W = h->P + h->off_wx[l];
    pl_split(W, back ? h->WbTP + h->off_wbtp[l] : nullptr, h->WfTP + h->off_wftp[l], h->Ip[l], h->D * h->N4,
             h->D * h->N4, back ? h->sc_wr[l].sp() : nullptr, h->sc_wc[l].sp(), nullptr, h->st);
  }
  for (int i = 0; i < h->ndense; ++i) {
    const bool back = i > 0 || h->npre == 0;   // the first pre stage reads the features: no gradient wrt its input
    const float* W = h->P + h->off_dw[i];
    pl_split(W, back ? h->DbTP + h->off_dbtp[i] : nullptr, h->DfTP + h->off_dftp[i], h->dIp[i], h->dWp[i], h->dWp[i],
             back ? h->sc_dr[i].sp() : nullptr, h->sc_dc[i].sp(), nullptr, h->st);
  }
  HIPCHK(h, hipGetLastError());
  return NASR_OK;
}

int scatter_to_device(nasr_ctx* h, const float* tf_flat, float* dev) {
  std::vector<float> host((size_t)h->np_int, 0.f);
  for (int64_t i = 0; i < h->np_tf; ++i) host[(size_t)h->tf2int[(size_t)i]] = tf_flat[i];
  HIPCHK(h, hipMemcpyAsync(dev, host.data(), (size_t)h->np_int * 4, hipMemcpyHostToDevice, h->st));
  HIPCHK(h, hipStreamSynchronize(h->st));
  return NASR_OK;
}

int gather_from_device(nasr_ctx* h, const float* dev, float* tf_flat) {
  std::vector<float> host((size_t)h->np_int);
  HIPCHK(h, hipMemcpyAsync(host.data(), dev, (size_t)h->np_int * 4, hipMemcpyDeviceToHost, h->st));
  if (int rc = sync_checked(h)) return rc;
  for (int64_t i = 0; i < h->np_tf; ++i) tf_flat[i] = host[(size_t)h->tf2int[(size_t)i]];
  return NASR_OK;
}

// ---- batch buffers --------------------------------------------------------------------------
}  // namespace nasr_impl
