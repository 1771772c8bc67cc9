// nasr_api.hip — the C ABI of include/nasr.h: context, HBM layout, TF<->internal parameter maps and
// the orchestration of one training step on one GPU.  See include/nasr.h for the reference
// interfaces each entry point replaces and DESIGN.md for the layout.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sched.h>
#include <unistd.h>

#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/nasr.h"
#include "kernels.h"

using namespace nasr;

namespace {

std::string g_create_error;
// nasr_last_error: the message of the calling thread's last failed call (nasr_stage_batch* may fail on a loader thread
// while the training thread is inside another call of the same handle: neither sees nor overwrites the other's text)
thread_local std::string t_err;
thread_local const void* t_err_handle = nullptr;

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t bytes, bool* grew) {
    if (bytes <= cap) return true;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8;  // head room: fewer re-allocations for ragged T
    if (hipMalloc(&p, want) != hipSuccess) {
      if (hipMalloc(&p, bytes) != hipSuccess) return false;
      want = bytes;
    }
    cap = want;
    if (grew) *grew = true;
    return true;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

struct TensorInfo {
  std::string name;
  int64_t offset, rows, cols;
};

enum Phase { PH_PACK = 0, PH_XPROJ, PH_RECF, PH_PROJCTC, PH_PROJB, PH_RECB, PH_WGRAD, PH_ADAM, PH_COUNT };

constexpr int GRAD_HEAD = 32;   // floats in front of the gradients (h->G = h->Gbase + GRAD_HEAD); [0] = fault word of the step
constexpr int MAX_BUCKETS = 16;

struct GraphKey {
  int T, l, bwd, s0;
  bool operator<(const GraphKey& o) const {
    if (T != o.T) return T < o.T;
    if (l != o.l) return l < o.l;
    if (bwd != o.bwd) return bwd < o.bwd;
    return s0 < o.s0;
  }
};

// One uploaded batch: the caller's arrays in HBM (features as given, or their centre slice + pad values) and the small
// integer arrays of the step packed into one "meta" buffer, with pinned host mirrors.  One slot is the
// resident batch, others take the NEXT batches while the step runs (nasr_stage_batch: copies on the handle's copy
// stream from pinned memory), so the upload of dataset.py:33-40's next batch leaves the timed step.
constexpr int NSLOT = 4;           // the resident batch + up to NSTAGE staged ahead + one always free for a synchronous upload
constexpr int NSTAGE = 2;
enum SlotState { SLOT_FREE = 0, SLOT_FILLING, SLOT_STAGED, SLOT_RESIDENT };
struct BatchSlot {
  DevBuf dfeats, dmeta;
  void *hfeats = nullptr, *hmeta = nullptr;      // hipHostMalloc
  size_t hfeats_cap = 0, hmeta_cap = 0;
  hipEvent_t ev_copy = nullptr, ev_released = nullptr;
  bool copy_valid = false, released_valid = false;
  int state = SLOT_FREE;
  unsigned gen = 0;                              // ticket = slot index | gen << 8
  // shape and layout of what is in it
  int B = 0, T = 0, Lmax = 0, Bp = 0, Tp = 0, ctx = 0, ncep = 0;
  bool has_labels = false, centre = false;
  int64_t frames = 0;
  size_t o_seq = 0, o_lablen = 0, o_labels = 0, o_cstart = 0, o_cpos = 0, o_rowmap = 0;   // int offsets into meta
  int32_t* meta_d() const { return dmeta.as<int32_t>(); }
};

}  // namespace

struct nasr_ctx {
  nasr_model_cfg cfg;
  int device = 0;
  hipStream_t st = nullptr;
  bool own_stream = false;
  // Bulk GEMMs (input projections, input / weight gradients, dense stages): fp32 products from two fp16 planes per
  // operand and three MFMA products (gemm_tph.hip); the planes are tiled copies made once per operand.
  unsigned char* WfTP = nullptr;       // per layer planes of Wx^T [D*N4][Ip]: B operand of the input GEMM
  unsigned char* WbTP = nullptr;       // per layer (l >= 1) planes of Wx [Ip][D*N4]: B operand of the input-gradient GEMM
  std::vector<size_t> off_wftp, off_wbtp;
  // Persistent recurrence (lstm_persist.hip): one launch per layer pass instead of T step launches.  Needs the full
  // 8 XCD x 32 CU chip and Hp <= 512; NASR_PERSIST=0 keeps the per-step kernels.
  bool persist = false;
  bool persist_ok = false;             // the device passed the census at create time
  bool persist_used = false;           // a persistent launch is in flight since the last check of *perr
  // re-arming the persistent recurrence after an abort (persist_check): the per-step kernels serve `rearm_after` clean
  // steps, then the census of nasr_create runs again and, if it passes, the persistent kernels come back; every further
  // abort doubles the wait.  NASR_PERSIST_REARM sets the first wait (0 = never re-arm).
  bool persist_wanted = false;         // the persistent mode is what this handle should run when the device allows it
  int64_t rearm_after = 0, rearm_wait = 0, clean_steps = 0;
  int persist_aborts = 0, persist_rearms = 0;
  float *Upf = nullptr, *Upb = nullptr;   // [L][D] operand images
  // forward recurrence on fp16 planes of U (v_mfma_f32_4x4x4_16B_f16, lstm_persist.hip): column scales / inverse scales of
  // every (layer, direction) matrix, [L*D][N4] each, measured after every optimiser step.  NASR_REC=f32 keeps fp32 MFMAs.
  bool rec_f16 = false;
  float *Ucs = nullptr, *Ucinv = nullptr;
  size_t imf = 0, imb = 0;             // floats per (layer, direction) image
  // the hand-offs validate themselves by epoch bits (lstm_persist.hip) and start from cleared buffers: one buffer per
  // layer pass, all of a pass cleared in one go
  float* xchf = nullptr;               // [L] h exchange buffers of the forward launches (persist_hx_bytes each)
  float* xchb = nullptr;               // [L] partial-sum exchange buffers of the BPTT launches (persist_px_bytes each)
  PersistCtl* pctl = nullptr;
  // Wide persistent FORWARD recurrence (lstm_wide.hip): Hp = 2048 (DeepSpeech's cell count), one launch per direction
  // with U resident in the registers of all 256 CUs; the BPTT of such a layer stays on the per-step kernels.  NASR_WIDE=0
  // (or NASR_PERSIST=0) keeps the per-step forward kernels.  Shares the abort / re-arm bookkeeping above.
  bool wide = false, wide_wanted = false;
  unsigned char* Uw = nullptr;         // [L][D] forward operand images (wide_image_bytes each)
  unsigned char* Uwb = nullptr;        // [L][D] BPTT operand images (U^T fragments under per-row scales)
  float *Urs = nullptr, *Urinv = nullptr;   // [L*D][Hp] row scales of every recurrent matrix and their inverses
  float* wsrow = nullptr;              // [D][64] dG scale per (direction, utterance) of the running BPTT pass
  void* whx = nullptr;                 // h exchange
  float* wpart = nullptr;              // cross-XCD inboxes: partial sums (forward) / dG planes (BPTT)
  void* wpx = nullptr;                 // BPTT: partial dh through the XCD's L2
  WideCtl* wctl = nullptr;
  unsigned* perr = nullptr;            // host-mapped sticky error word
  // in-library gradient exchange (nasr_comm_*): one RCCL rank per handle, collectives on a side stream
  void* comm = nullptr;                  // ncclComm_t
  // nasr_comm_mean's own communicator (ncclCommSplit of `comm`, same ranks) and stream: the two host floats of a step do
  // not queue up behind the step's gradient buckets.  NULL (old librccl): the mean shares `comm` and waits for them.
  void* comm2 = nullptr;
  hipStream_t comm_st2 = nullptr;
  int comm_rank = 0, comm_n = 1;
  hipStream_t comm_st = nullptr;
  hipEvent_t ev_comm = nullptr;
  float* comm_scratch = nullptr;         // 64 floats for nasr_comm_mean

  // model dims
  int F, Fp, H, Hp, N4, D, L, C, Cp, Pin, Pinp;
  std::vector<int> Ip;                  // padded input width per layer
  std::vector<int64_t> off_wx, off_bias;  // per layer
  std::vector<int64_t> off_u;           // per (layer, dir)
  int64_t off_w = 0, off_b = 0, np_int = 0;
  std::vector<TensorInfo> tensors;
  int64_t np_tf = 0;
  std::vector<int32_t> tf2int;          // TF flat index -> internal flat index

  // dense stages of the DeepSpeech family (networks/deepspeech.py): stage i < npre feeds the LSTM stack, stage npre
  // (when has_post) sits between the stack and the logits.  W_i [dIp][dWp] row-major, b_i [dWp].
  int npre = 0, ndense = 0;
  bool has_post = false;
  int F0 = 0;                            // unpadded input width of LSTM layer 0 (F, or the last pre stage's width)
  std::vector<int> dWid, dWp, dIn, dIp;
  std::vector<int64_t> off_dw, off_db;
  std::vector<size_t> off_dftp, off_dbtp;
  unsigned char *DfTP = nullptr, *DbTP = nullptr;   // TP of W_i^T [dWp][dIp] and of W_i [dIp][dWp]
  std::vector<DevBuf> Ybuf, dYbuf;       // stage outputs and their gradients [R][dWp]
  DevBuf DTP;                            // scratch: TP of a stage input with the frame index as contraction index
  uint32_t drop_seed = 4567u, drop_counter = 0;   // random_seed of networks/deepspeech.py:26

  float *P = nullptr, *M = nullptr, *V = nullptr, *G = nullptr, *Uf = nullptr, *Ub = nullptr;
  // Every operand row of a plane GEMM carries a power-of-two scale (device floats, scale and 1/scale), measured per step
  // for everything whose range is not known in advance.
  struct SV {
    DevBuf s, inv;
    bool ensure(size_t n) { bool g = false; return s.ensure(n * 4, &g) && inv.ensure(n * 4, &g); }
    void release() { s.release(); inv.release(); }
    float* sp() const { return s.as<float>(); }
    float* ip() const { return inv.as<float>(); }
  };
  SV sc15;                                   // constants 2^15 / 2^-15: LSTM outputs (|h| < 1), rows and columns
  size_t sc15_n = 0;
  SV sc_x0r, sc_x0c;                         // features: per frame row / per feature column
  std::vector<SV> sc_yr, sc_yc;              // dense stage outputs
  SV sc_gr, sc_gc;                           // the gate / dense pre-activation gradient being worked on
  std::vector<SV> sc_wr, sc_wc;              // Wx[l]: per input row / per gate column
  std::vector<SV> sc_dr, sc_dc;              // dense W[i]
  DevBuf scws;                               // partial maxima (launch_tph_scales)
  int gttp_layer = -1;                       // layer whose transposed dG planes gemm_dx has just written (fused split)
  int dgmax_layer = -1;                      // layer whose |dG| maxima the persistent BPTT kernel has left in `dgmax`
  DevBuf dgmax;                              // [D*32][R] row parts | [8/D][D*N4] column parts (persist_dgmax_floats)
  float* Gbase = nullptr;                    // allocation behind G: [GRAD_HEAD floats, [0] = fault word][np_int gradients]
  // gradient buckets: (offset, count) in floats from Gbase, in the order backward() completes them; one event each
  std::vector<std::pair<int64_t, int64_t>> buckets;
  std::vector<hipEvent_t> ev_bucket;
  std::vector<int> bucket_of_layer;          // LSTM layer -> bucket whose last gradients are that layer's (-1: none)
  // Persistent mode: bucket(l)'s event is recorded AFTER the persistent BPTT launch of layer l-1 instead of right after
  // weight_grads(l), so that a collective released by it co-runs with the GEMM phase of layer l-1, not with the launch
  // that wants every CU's memory queue to itself (nasr_set_bucket_defer; NASR_BUCKET_DEFER=0 at create).
  bool bucket_defer = true;
  // Adam's step count t lives ON THE DEVICE (AdamDev, optim.hip): the launch that finds the step's fault word set leaves
  // it alone, so a void step never enters the bias correction - whenever the host learns about it.
  AdamDev* adam_dev = nullptr;
  float lr;
  // Results of a step without waiting for its end (nasr_get_step_results): loss, the fault word as it stands after the
  // forward pass, and the greedy decode are copied to pinned memory right behind the CTC forward kernels; the fault
  // word at the END of a step is copied behind its Adam launch (nasr_settle_step).  Two slots each: the host may be
  // one step ahead of the device.
  struct StepRes { void* host = nullptr; size_t cap = 0; uint32_t* stamp = nullptr; uint32_t seq = 0; bool valid = false; int B = 0, Bp = 0, Tp = 0; };
  StepRes res[2];
  int res_cur = 0;
  struct StepEnd { float* host = nullptr; uint32_t* stamp = nullptr; uint32_t seq = 0; bool valid = false; int64_t token = 0; };
  static constexpr int NEND = 4;             // steps whose end the host may still ask about (nasr_settle_token)
  StepEnd endw[NEND];
  int end_cur = 0;
  int64_t step_token = 0;                    // sequence number of the optimiser step enqueued last
  uint32_t stamp_seq = 0;

  // resident batch
  bool resident = false, have_grads = false, have_fwd = false;
  int B = 0, Bp = 0, T = 0, Lmax = 0, Tp = 0, KS = 1;
  int64_t frames = 0;
  std::vector<int32_t> h_seq;
  BatchSlot slots[NSLOT];
  BatchSlot* cur = nullptr;                  // the resident batch
  hipStream_t cst = nullptr;                 // copy stream of nasr_stage_batch
  std::mutex slot_mu;                        // slot states (nasr_stage_batch may run on a loader thread)
  int slot_rr = 0;
  // device arrays of the resident batch (inside cur->dmeta / cur->dfeats)
  int32_t *seq_p = nullptr, *lablen_p = nullptr, *labels_p = nullptr, *cstart_p = nullptr, *cpos_p = nullptr,
          *rowmap_p = nullptr;

  // Weight gradients under the BPTT of the layer below (persistent mode, Hp = 512, L > 1; NASR_WGRAD_OVERLAP=0 turns it off): weight_grads(l)
  // runs on a low-priority side stream in the 3-wave GEMM instantiation that fits on a CU beside a persistent workgroup,
  // from its own copies of everything the main stream rewrites meanwhile (dG^T planes, column scales, partial column sums,
  // slabs: index l & 1), and is joined before layer l's gradients are released / Adam.
  bool wg_overlap = false;
  hipStream_t wst = nullptr;
  hipEvent_t ev_dx = nullptr;
  std::vector<hipEvent_t> ev_wg;             // per layer: its weight gradients are complete
  std::vector<char> wg_pending;              // ... and the main stream has not waited for that yet
  DevBuf GTTP2, csws2, slabs2;
  SV sc_gc2;
  DevBuf XTP, X0TTP, GTP, GTTP;   // tiled-plane copies of activations / dG
  std::vector<DevBuf> OTT;        // per layer: planes of out[l] with the frame index as contraction index (weight gradients)
  std::vector<char> ott_valid;    // ... written by the forward pass of this step already (together with the planes of layer l+1's input)
  DevBuf seqbuf, X0, logits, logz, alpha, beta, aoff, boff, logp, nll, loss, slabs, csws, amax, ids, lens,
      stage;
  std::vector<DevBuf> gates, outb, cbuf;
  DevBuf dout, hstate, partial, dcstate, dgbuf;   // shared by the layers (a layer's backward pass is over before the next starts)

  // graphs
  bool graph_mode = true;
  bool step_decode = false, have_decoded = false;
  std::map<GraphKey, hipGraphExec_t> graphs;

  // profiling
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { int ph; hipEvent_t a, b; };
  std::vector<Span> spans;
  hipEvent_t ev_total_a = nullptr, ev_total_b = nullptr;
  bool window_open = false, total_valid = false;   // timing window [upload|compute_grads .. apply_adam]
  int n_fwd_launch = 0, n_bwd_launch = 0;
  nasr_phase_times last_times;

  int fail(int code, const std::string& m) {
    t_err = m;
    t_err_handle = this;
    return code;
  }
};

namespace {

#define HIPCHK(h, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return (h)->fail(NASR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

// A persistent launch that gave up (bounded spin, unexpected placement) leaves its outputs undefined: surface it at
// the next host sync and use the per-step kernels from then on.
// ---- tiled fp16 planes (gemm_tph.hip) ---------------------------------------------------------------------------
inline size_t pl_rb_bytes(int nkb) { return (size_t)nkb * 2 * 1024; }   // one 32-row block: nkb k-blocks x 2 parts x 1 KiB
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
void pl_gemm(GemmTPHDesc g, const float* a_inv, const float* b_inv, hipStream_t st, int64_t ainv_bstride = 0,
             int64_t binv_bstride = 0) {
  g.a_inv = a_inv; g.b_inv = b_inv; g.ainv_bstride = ainv_bstride; g.binv_bstride = binv_bstride;
  launch_gemm_tph(g, st);
}
// scale vectors of an activation tensor: the features, a dense stage's output (index i), or an LSTM layer's output
struct ActScale { const float *rs, *rinv, *cs, *cinv; };
inline ActScale act_x0(const nasr_ctx* h) { return {h->sc_x0r.sp(), h->sc_x0r.ip(), h->sc_x0c.sp(), h->sc_x0c.ip()}; }
inline ActScale act_y(const nasr_ctx* h, int i) { return {h->sc_yr[i].sp(), h->sc_yr[i].ip(), h->sc_yc[i].sp(), h->sc_yc[i].ip()}; }
inline ActScale act_out(const nasr_ctx* h) { return {h->sc15.sp(), h->sc15.ip(), h->sc15.sp(), h->sc15.ip()}; }
inline ActScale lstm_in_scale(const nasr_ctx* h, int l) {
  if (l > 0) return act_out(h);
  return h->npre ? act_y(h, h->npre - 1) : act_x0(h);
}
inline ActScale dense_in_scale(const nasr_ctx* h, int i) {
  if (i == 0 && h->npre > 0) return act_x0(h);
  if (i < h->npre) return act_y(h, i - 1);
  return act_out(h);                               // the post stage reads the top LSTM layer
}

int repack(nasr_ctx* h);
void drop_graphs(nasr_ctx* h);

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

struct PhaseScope {
  nasr_ctx* h;
  int ph;
  hipEvent_t a = nullptr;
  PhaseScope(nasr_ctx* h_, int ph_) : h(h_), ph(ph_) {
    if (h->profiling) {
      a = next_event(h);
      (void)hipEventRecord(a, h->st);
    }
  }
  ~PhaseScope() {
    if (h->profiling) {
      hipEvent_t b = next_event(h);
      (void)hipEventRecord(b, h->st);
      h->spans.push_back({ph, a, b});
    }
  }
};

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
      for (int i = 0; i < h->ndense; ++i) ok &= h->sc_yr[i].ensure(R) && h->sc_yc[i].ensure((size_t)h->dWp[i]);
      bool g2 = false;
      ok &= h->scws.ensure(tph_scale_ws_floats((int)R, std::max(ipmax, wmax)) * 4, &g2);
    }
  }
  ok &= h->logits.ensure((size_t)Tp * Bp * h->Cp * 4, &grew);
  ok &= h->logz.ensure((size_t)Tp * Bp * 4, &grew);
  const int KSa = KS <= 8 ? KS : (KS <= 12 ? 12 : 16);   // kernel instantiations
  ok &= h->alpha.ensure((size_t)B * (T + 8) * KSa * 64 * 4, &grew);
  ok &= h->beta.ensure((size_t)B * (T + 8) * KSa * 64 * 4, &grew);
  ok &= h->aoff.ensure((size_t)B * (T + 8) * 8, &grew);
  ok &= h->boff.ensure((size_t)B * (T + 8) * 8, &grew);
  ok &= h->logp.ensure((size_t)Bp * 8, &grew);
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
  if (hipHostMalloc(p, want, hipHostMallocDefault) != hipSuccess) return false;
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
  const size_t nmeta = s->o_rowmap + (sr ? (size_t)Tp * Bp : 0);
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
    if (s->has_labels && h->npre == 0)   // layer-0 input with the frame index as contraction index, for dWx = X^T dG
      pl_split(h->X0.as<float>(), nullptr, h->X0TTP.as<unsigned char>(), T * Bp, h->Fp, h->Fp, nullptr, h->sc_x0c.sp(),
               nullptr, h->st);
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
           int B, int T, int Lmax, const float* centre = nullptr, const float* pad_value = nullptr, int ctx = 0,
           int ncep = 0) {
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

inline float* dout_of(nasr_ctx* h, int) { return h->dout.as<float>(); }
inline float* dg_of(nasr_ctx* h, int) { return h->dgbuf.as<float>(); }
// what weight_grads(l) reads of layer l's dG: with the overlap on, odd layers have copies of their own (the main stream
// is rewriting the others for layer l-1 while the side stream still reads these)
inline bool wg_alt(const nasr_ctx* h, int l) { return h->wg_overlap && (l & 1); }
inline unsigned char* gttp_of(nasr_ctx* h, int l) { return (wg_alt(h, l) ? h->GTTP2 : h->GTTP).as<unsigned char>(); }
inline float* csws_of(nasr_ctx* h, int l) { return (wg_alt(h, l) ? h->csws2 : h->csws).as<float>(); }
inline nasr_ctx::SV& gc_of(nasr_ctx* h, int l) { return wg_alt(h, l) ? h->sc_gc2 : h->sc_gc; }

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

// input of LSTM layer l: the features, the last pre-dense stage's output, or the layer below
inline const float* lstm_input(nasr_ctx* h, int l) {
  if (l > 0) return h->outb[l - 1].as<float>();
  return h->npre ? h->Ybuf[h->npre - 1].as<float>() : h->X0.as<float>();
}

// gates_l = X_l * Wx_l + bias_l over all R rows
void gemm_xproj(nasr_ctx* h, int l, int R, hipStream_t st) {
  const int D = h->D, N4 = h->N4, Ip = h->Ip[l];
  const ActScale as = lstm_in_scale(h, l);
  // a training step wants the layer below's output a second time, with the frame index as contraction index (its own
  // recurrent weight gradient and this layer's input weight gradient): both plane sets in this one pass over it
  const bool both = l > 0 && h->cur && h->cur->has_labels;
  pl_split(lstm_input(h, l), h->XTP.as<unsigned char>(), both ? h->OTT[l - 1].as<unsigned char>() : nullptr, R, Ip, Ip, as.rs,
           both ? as.cs : nullptr, nullptr, st);
  if (l > 0) h->ott_valid[l - 1] = both;
  GemmTPHDesc g{};
  g.A = h->XTP.as<unsigned char>(); g.B = h->WfTP + h->off_wftp[l]; g.C = h->gates[l].as<float>();
  g.M = R; g.N = D * N4; g.K = Ip; g.nkbA = (Ip + 15) / 16; g.nkbB = g.nkbA; g.ldc = D * N4;
  g.bias = h->P + h->off_bias[l]; g.split_k = 1;
  pl_gemm(g, as.rinv, h->sc_wc[l].ip(), st);
}

// dOut_{l-1} = dG_l * Wx_l^T : the gradient wrt layer l's input = the layer below's output.  weight_grads(l) follows:
// dG is split ONCE into both plane sets (frame-row scales for this product, gate-column scales for the weight gradients)
// and its 64-row partial column sums (the bias gradient).
void gemm_dx(nasr_ctx* h, int l, int R, hipStream_t st) {
  const int D = h->D, N4 = h->N4;
  const float* A = dg_of(h, l);
  float* C = l > 0 ? dout_of(h, l - 1) : h->dYbuf[h->npre - 1].as<float>();
  dg_scales(h, l, R, true, st);
  pl_split(A, h->GTP.as<unsigned char>(), gttp_of(h, l), R, D * N4, D * N4, h->sc_gr.sp(), gc_of(h, l).sp(), csws_of(h, l), st);
  h->gttp_layer = l;   // weight_grads(l): transposed planes and column-sum partials of dG are there
  GemmTPHDesc g{};
  g.A = h->GTP.as<unsigned char>(); g.B = h->WbTP + h->off_wbtp[l]; g.C = C;
  g.M = R; g.N = h->Ip[l]; g.K = D * N4; g.nkbA = (D * N4 + 15) / 16; g.nkbB = g.nkbA; g.ldc = h->Ip[l];
  g.split_k = gemm_tph_pick_split(g.M, g.N, g.K);
  g.slabs = ensure_slabs(h, g.split_k, g.M, g.N);
  if (g.split_k > 1 && !g.slabs) g.split_k = 1;
  pl_gemm(g, h->sc_gr.ip(), h->sc_wr[l].ip(), st);
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
    launch_greedy(d, h->logits.as<float>(), h->seq_p, h->amax.as<int>(), h->ids.as<int>(), h->lens.as<int>(),
                  h->st);
    h->have_decoded = true;
    // what Network.train returns is known HERE, before the backward pass: copy it out now (nasr_get_step_results)
    h->res_cur ^= 1;
    nasr_ctx::StepRes& r = h->res[h->res_cur];
    const size_t bytes = 8 + (size_t)h->Bp * 4 + (size_t)h->B * h->Tp * 4;
    if (!pinned_ensure(&r.host, &r.cap, bytes)) return h->fail(NASR_ERR_HIP, "hipHostMalloc of the step results failed");
    char* hp = static_cast<char*>(r.host);
    HIPCHK(h, hipMemcpyAsync(hp, h->loss.p, 4, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipMemcpyAsync(hp + 4, h->Gbase, 4, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipMemcpyAsync(hp + 8, h->lens.p, (size_t)h->Bp * 4, hipMemcpyDeviceToHost, h->st));
    HIPCHK(h, hipMemcpyAsync(hp + 8 + (size_t)h->Bp * 4, h->ids.p, (size_t)h->B * h->Tp * 4, hipMemcpyDeviceToHost, h->st));
    r.seq = ++h->stamp_seq;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, h->st, r.stamp, r.seq, (float*)nullptr, (const float*)nullptr);
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
  const int nkb = (R + 15) / 16;
  // The side instantiation splits K exactly as the main one would: every output element then sums the same k-blocks in
  // the same order whatever the tile shape - the gradients are bitwise those of the serial order.  (NASR_SIDE_SPLIT=own:
  // the split its own cost model picks, for the A/B logs.)
  static const int split_mode = [] { const char* e = getenv("NASR_SIDE_SPLIT"); return !e ? 0 : e[0] == 'o' ? 1 : e[0] == '1' ? 2 : 0; }();
  const bool side_split = side && split_mode == 1;
  const bool side_one = side && split_mode == 2;       // (A/B logs: no K split at all on the side stream)
  // one pass over dG: its transposed planes + 64-row partial column sums (already there when gemm_dx(l) ran)
  if (h->gttp_layer != l) {
    dg_scales(h, l, R, false, ws);
    pl_split(dG, nullptr, GT, R, D * N4, D * N4, nullptr, gc.sp(), cs_part, ws);
  }
  h->gttp_layer = -1;
  const ActScale ao = act_out(h), ai = lstm_in_scale(h, l);
  for (int m = std::max(l - 1, 0); m <= l; ++m)     // out[l] (recurrent weight gradient), out[l-1] (input weight gradient)
    if (!h->ott_valid[m]) {
      pl_split(h->outb[m].as<float>(), nullptr, h->OTT[m].as<unsigned char>(), R, D * Hp, D * Hp, nullptr, ao.cs, nullptr, ws);
      h->ott_valid[m] = 1;
    }
  if (l == 0 && h->npre)
    pl_split(lstm_input(h, l), nullptr, h->X0TTP.as<unsigned char>(), R, h->Ip[0], h->Ip[0], nullptr, ai.cs, nullptr, ws);
  {  // dWx = X^T dG
    GemmTPHDesc g{};
    g.A = l == 0 ? h->X0TTP.as<unsigned char>() : h->OTT[l - 1].as<unsigned char>();
    g.B = GT; g.C = h->G + h->off_wx[l];
    g.M = h->Ip[l]; g.N = D * N4; g.K = R; g.nkbA = nkb; g.nkbB = nkb; g.ldc = D * N4;
    g.side = side;
    g.split_k = side_one ? 1 : gemm_tph_pick_split(g.M, g.N, g.K, 1, side_split);
    g.slabs = slabs_for(g.split_k, g.M, g.N);
    if (g.split_k > 1 && !g.slabs) return h->fail(NASR_ERR_HIP, "slab workspace allocation failed");
    pl_gemm(g, ai.cinv, gc.ip(), ws);
  }
  launch_colsum_parts(cs_part, tp_split2_parts(R), D * N4, h->G + h->off_bias[l], ws);
  {  // dU = shift(H)^T dG : h_prev of frame t is out[t-1] (fw) / out[t+1] (bw); both directions in one launch
    GemmTPHDesc g{};
    g.A = h->OTT[l].as<unsigned char>(); g.B = GT; g.C = h->G + h->off_u[(size_t)l * D];
    g.M = Hp; g.N = N4; g.K = R; g.nkbA = nkb; g.nkbB = nkb; g.ldc = N4;
    g.a_kshift = -Bp;
    g.nbatch = D;
    g.a_bstride = (size_t)(Hp / 32) * pl_rb_bytes(nkb); g.b_bstride = (size_t)(N4 / 32) * pl_rb_bytes(nkb);
    g.c_bstride = (int64_t)Hp * N4;            // off_u[l*D + 1] - off_u[l*D] (build_layout)
    g.a_kshift1 = Bp;
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
  {
    const char* eo = getenv("NASR_WGRAD_OVERLAP");
    h->wg_overlap = !(eo && eo[0] == '0') && h->persist && h->Hp == 512 && h->L > 1;   // on unless NASR_WGRAD_OVERLAP=0
    h->ev_wg.assign(h->L, nullptr);
    h->wg_pending.assign(h->L, 0);
    if (h->wg_overlap) {
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
  if (h->wst) { (void)hipStreamSynchronize(h->wst); (void)hipStreamDestroy(h->wst); }
  if (h->ev_dx) (void)hipEventDestroy(h->ev_dx);
  for (hipEvent_t e : h->ev_wg) if (e) (void)hipEventDestroy(e);
  for (auto& b : h->OTT) b.release();
  for (nasr_ctx::SV* v : {&h->sc15, &h->sc_x0r, &h->sc_x0c, &h->sc_gr, &h->sc_gc}) v->release();
  for (auto* vec : {&h->sc_yr, &h->sc_yc, &h->sc_wr, &h->sc_wc, &h->sc_dr, &h->sc_dc})
    for (auto& v : *vec) v.release();
  (void)nasr_comm_destroy(h);
  if (h->adam_dev) (void)hipFree(h->adam_dev);
  for (auto& r : h->res) {
    if (r.host) (void)hipHostFree(r.host);
    if (r.stamp) (void)hipHostFree(r.stamp);
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
                    &h->alpha, &h->beta, &h->aoff, &h->boff, &h->logp, &h->nll, &h->loss, &h->slabs,
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
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, h->st, e.stamp, e.seq, e.host, (const float*)h->Gbase);
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
  if (lens_out) memcpy(lens_out, hp + 8, (size_t)r.B * 4);
  if (ids_out) memcpy(ids_out, hp + 8 + (size_t)r.Bp * 4, (size_t)r.B * r.Tp * 4);
  return NASR_OK;
}

}  // extern "C"
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
extern "C" {

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

// ---- in-library gradient exchange over RCCL ----------------------------------------------------------------------
// librccl is bound at run time (dlopen), only when a host asks for it: a single-GPU host never loads it.
extern "C++" {
namespace {
struct RcclApi {
  struct Uid { char internal[128]; };
  int (*GetUniqueId)(Uid*) = nullptr;
  int (*CommInitRank)(void**, int, Uid, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommSplit)(void*, int, int, void**, void*) = nullptr;     // optional (RCCL >= 2.18): a second communicator of the same ranks
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string why;
};
RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) {
    api.why = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?");
    return api;
  }
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  api.CommSplit = reinterpret_cast<decltype(api.CommSplit)>(dlsym(lib, "ncclCommSplit"));
  api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllReduce;
  if (!api.ok) api.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
  return api;
}
constexpr int kNcclFloat = 7, kNcclSum = 0;      // rccl.h: ncclFloat32, ncclSum
int rccl_fail(nasr_ctx* h, const char* what, int rc) {
  const RcclApi& r = rccl();
  return h->fail(NASR_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" +
                                   std::to_string(rc) + ")");
}
}  // namespace
}  // extern "C++"

int nasr_comm_unique_id(void* id128) {
  if (!id128) return NASR_ERR_ARG;
  RcclApi& r = rccl();
  if (!r.ok) {
    g_create_error = r.why;
    return NASR_ERR_HIP;
  }
  RcclApi::Uid u;
  const int rc = r.GetUniqueId(&u);
  if (rc) {
    g_create_error = std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return NASR_ERR_HIP;
  }
  memcpy(id128, u.internal, 128);
  return NASR_OK;
}

int nasr_comm_init(nasr_handle h, const void* id128, int rank, int nranks) {
  if (!h || !id128) return NASR_ERR_ARG;
  if (nranks < 1 || rank < 0 || rank >= nranks) return h->fail(NASR_ERR_ARG, "nasr_comm_init: bad rank / nranks");
  if (h->comm) return h->fail(NASR_ERR_STATE, "nasr_comm_init: this handle already has a communicator");
  RcclApi& r = rccl();
  if (!r.ok) return h->fail(NASR_ERR_HIP, r.why);
  HIPCHK(h, hipSetDevice(h->device));
  RcclApi::Uid u;
  memcpy(u.internal, id128, 128);
  void* c = nullptr;
  const int rc = r.CommInitRank(&c, nranks, u, rank);       // blocks until every rank has joined
  if (rc) return rccl_fail(h, "ncclCommInitRank", rc);
  h->comm = c;
  h->comm_rank = rank;
  h->comm_n = nranks;
  HIPCHK(h, hipStreamCreateWithFlags(&h->comm_st, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
  HIPCHK(h, hipMalloc(&h->comm_scratch, 64 * sizeof(float)));
  if (r.CommSplit) {                      // collective over all ranks of `comm`: every rank gets here (same library everywhere)
    void* c2 = nullptr;
    if (r.CommSplit(c, 0, rank, &c2, nullptr) == 0 && c2) {
      h->comm2 = c2;
      HIPCHK(h, hipStreamCreateWithFlags(&h->comm_st2, hipStreamNonBlocking));
    }
  }
  return NASR_OK;
}

int nasr_comm_size(nasr_handle h) { return h ? (h->comm ? h->comm_n : 1) : NASR_ERR_ARG; }

int nasr_comm_allreduce_grads(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  if (!h->comm) return h->fail(NASR_ERR_STATE, "nasr_comm_allreduce_grads: call nasr_comm_init first");
  if (!h->have_grads) return h->fail(NASR_ERR_STATE, "nasr_comm_allreduce_grads without gradients");
  RcclApi& r = rccl();
  HIPCHK(h, hipSetDevice(h->device));
  // bucket i crosses xGMI as soon as the backward pass has finished it (its event, held back over the next persistent
  // BPTT launch when bucket_defer is on), under the layers below; the handle's stream then waits for the last collective
  for (size_t i = 0; i < h->buckets.size(); ++i) {
    HIPCHK(h, hipStreamWaitEvent(h->comm_st, h->ev_bucket[i], 0));
    float* p = h->Gbase + h->buckets[i].first;
    const int rc = r.AllReduce(p, p, (size_t)h->buckets[i].second, kNcclFloat, kNcclSum, h->comm, h->comm_st);
    if (rc) return rccl_fail(h, "ncclAllReduce", rc);
  }
  HIPCHK(h, hipEventRecord(h->ev_comm, h->comm_st));
  HIPCHK(h, hipStreamWaitEvent(h->st, h->ev_comm, 0));
  return NASR_OK;
}

int nasr_comm_mean(nasr_handle h, float* vals, int n) {
  if (!h || !vals) return NASR_ERR_ARG;
  if (n < 1 || n > 64) return h->fail(NASR_ERR_ARG, "nasr_comm_mean: 1..64 values");
  if (!h->comm) return NASR_OK;                   // one rank: the mean is the value
  RcclApi& r = rccl();
  HIPCHK(h, hipSetDevice(h->device));
  // On its own communicator and stream when the library offers ncclCommSplit: the collective of a few floats neither waits
  // for the gradient buckets of the step in flight nor for the compute stream.  Otherwise (one communicator executes its
  // collectives in issue order) it goes behind them on the compute stream, as documented in include/nasr.h.
  void* c = h->comm2 ? h->comm2 : h->comm;
  hipStream_t st = h->comm2 ? h->comm_st2 : h->st;
  HIPCHK(h, hipMemcpyAsync(h->comm_scratch, vals, (size_t)n * 4, hipMemcpyHostToDevice, st));
  const int rc = r.AllReduce(h->comm_scratch, h->comm_scratch, (size_t)n, kNcclFloat, kNcclSum, c, st);
  if (rc) return rccl_fail(h, "ncclAllReduce", rc);
  HIPCHK(h, hipMemcpyAsync(vals, h->comm_scratch, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int i = 0; i < n; ++i) vals[i] /= (float)h->comm_n;
  return NASR_OK;
}

int nasr_comm_destroy(nasr_handle h) {
  if (!h) return NASR_ERR_ARG;
  if (!h->comm) return NASR_OK;
  (void)hipSetDevice(h->device);
  if (h->comm_st) (void)hipStreamSynchronize(h->comm_st);
  if (h->comm_st2) (void)hipStreamSynchronize(h->comm_st2);
  if (h->comm2) (void)rccl().CommDestroy(h->comm2);
  h->comm2 = nullptr;
  if (h->comm_st2) (void)hipStreamDestroy(h->comm_st2);
  h->comm_st2 = nullptr;
  (void)rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  if (h->comm_st) (void)hipStreamDestroy(h->comm_st);
  if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
  if (h->comm_scratch) (void)hipFree(h->comm_scratch);
  h->comm_st = nullptr; h->ev_comm = nullptr; h->comm_scratch = nullptr;
  h->comm_n = 1; h->comm_rank = 0;
  return NASR_OK;
}

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
