// nasr_ctx.h — what the translation units behind include/nasr.h share: the handle (struct nasr_ctx: model layout, HBM buffers,
// batch slots, streams / events, recurrence mode), small helpers, and the functions they call in each other (internal, C++).
//   nasr_layout.hip  parameter layout, TF <-> internal maps, operand images (repack), persistent-mode management
//   nasr_batch.hip   batch buffers and slots: upload, stage / commit
//   nasr_pass.hip    forward, CTC, backward: the orchestration of one step on the handle's streams
//   nasr_api.hip     the C ABI entry points
//   nasr_comm.hip    RCCL bound with dlopen: nasr_comm_*
#pragma once
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

namespace nasr_impl {
using namespace nasr;

extern std::string g_create_error;
// nasr_last_error: the message of the calling thread's last failed call (nasr_stage_batch* may fail on a loader thread
// while the training thread is inside another call of the same handle: neither sees nor overwrites the other's text)
extern thread_local std::string t_err;
extern thread_local const void* t_err_handle;

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
  size_t o_vrow = 0, o_vprev = 0, o_vnext = 0;   // compacted rows of a ragged batch (Rv of them, padded to Rvp with -1), or unused
  int Rv = 0, Rvp = 0;
  bool cmp = false;
  int32_t* meta_d() const { return dmeta.as<int32_t>(); }
};

}  // namespace nasr_impl

// (internal header: the translation units behind the ABI use both namespaces unqualified)
using namespace nasr;
using namespace nasr_impl;

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
  struct StepRes { void* host = nullptr; size_t cap = 0; uint32_t* stamp = nullptr; uint32_t seq = 0; bool valid = false; int B = 0, Bp = 0, Tp = 0; bool logits = false, greedy = false;
                   hipEvent_t ev_lg = nullptr; };   // ev_lg: the step's logits have landed in host memory (stream d2h)
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
  hipStream_t d2h = nullptr;                 // the step's logits leave on this one, from a snapshot (ctc_forward)
  hipEvent_t ev_snap = nullptr;
  DevBuf logits_snap;
  std::mutex slot_mu;                        // slot states (nasr_stage_batch may run on a loader thread)
  int slot_rr = 0;
  // device arrays of the resident batch (inside cur->dmeta / cur->dfeats)
  int32_t *seq_p = nullptr, *lablen_p = nullptr, *labels_p = nullptr, *cstart_p = nullptr, *cpos_p = nullptr,
          *rowmap_p = nullptr;
  // Ragged batches (dataset.py:75-77 pads every utterance to the batch maximum): when at least 15 % of the T x Bp frame rows
  // are padding, the plane passes and GEMMs of a plain (Bi)LSTM stack work on the COMPACTED rows - only the frames t < seq_len[b],
  // time-major - and scatter their results back (split passes gather by vrow, GEMM epilogues scatter by it; vprev / vnext =
  // the row of the frame before / after each compacted row, -1 at an utterance's first / last frame: the h_{t-1} operand of
  // the recurrent weight gradient).  cmp_rows = Rv (0: no compaction for the resident batch).  NASR_COMPACT=0 turns it off.
  bool compactable = false;
  int cmp_rows = 0, cmp_rows_p = 0;
  int32_t *vrow_p = nullptr, *vprev_p = nullptr, *vnext_p = nullptr;
  SV sc_cr, sc_cx;                           // row scales gathered to the compacted order: dG rows / feature rows
  DevBuf OTS;                                // planes of shift(out[l])^T over the compacted rows (recurrent weight gradient)

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
  DevBuf ctcprobs, ctckexp;                  // emission rows and column offsets of the CTC lattice (ctc.hip (2b))
  DevBuf seqbuf, X0, logits, logz, alpha, beta, aoff, boff, logp, nll, loss, slabs, csws, amax, ids, lens,
      stage;
  std::vector<DevBuf> gates, outb, cbuf;
  DevBuf dout, hstate, partial, dcstate, dgbuf;   // shared by the layers (a layer's backward pass is over before the next starts)

  // graphs
  bool graph_mode = true;
  bool step_decode = false, step_logits = false, have_decoded = false;   // nasr_set_step_decode: any bit / bit 1
  bool step_greedy = false;                                              // ... bit 0
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

namespace nasr_impl {

#define HIPCHK(h, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return (h)->fail(NASR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)
// ---- tiled fp16 planes (gemm_tph.hip) ---------------------------------------------------------------------------
inline size_t pl_rb_bytes(int nkb) { return (size_t)nkb * 2 * 1024; }   // one 32-row block: nkb k-blocks x 2 parts x 1 KiB
// scales of src [rows][K]: per row into `row`, per column into `col` (either may be NULL)
void pl_scales(nasr_ctx* h, const float* src, int rows, int K, int ld, nasr_ctx::SV* row, nasr_ctx::SV* col, hipStream_t st);
// planes of src [rows][K] (tpN, scaled per row by rs[]) and / or of its transpose (tpT, scaled per src column by cs[]);
// colpart: 64-row partial column sums for launch_colsum_parts
void pl_split(const float* src, unsigned char* tpN, unsigned char* tpT, int rows, int K, int ld, const float* rs,
              const float* cs, float* colpart, hipStream_t st);
// a_inv / b_inv: inverse scales of A's / B's rows; the strides apply to batch 1 of a two-batch launch
void pl_gemm(GemmTPHDesc g, const float* a_inv, const float* b_inv, hipStream_t st, int64_t ainv_bstride = 0,
             int64_t binv_bstride = 0);
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

// ---- nasr_layout.hip
int build_layout(nasr_ctx* h);
int repack(nasr_ctx* h);
int scatter_to_device(nasr_ctx* h, const float* tf_flat, float* dev);
int gather_from_device(nasr_ctx* h, const float* dev, float* tf_flat);
int persist_check(nasr_ctx* h);
bool persist_census(nasr_ctx* h);
void persist_rearm(nasr_ctx* h);
// a word in host-mapped pinned memory written in stream order (and, with f0_dst, a device float copied beside it)
void launch_stamp(unsigned* dst, unsigned value, float* f0_dst, const float* f0_src, hipStream_t st);
void launch_publish_results(const float* loss, const float* fault, const int* lens, int Bp, const int* ids, int n_ids, void* host,
                            unsigned* stamp, unsigned value, hipStream_t st);
bool wait_stamp(const uint32_t* w, uint32_t want, double timeout_s);
int sync_checked(nasr_ctx* h);
void drop_graphs(nasr_ctx* h);
hipEvent_t next_event(nasr_ctx* h);
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

// ---- nasr_batch.hip
int ensure_shape(nasr_ctx* h, int B, int T, int Lmax);
bool pinned_ensure(void** p, size_t* cap, size_t bytes);
void slot_set_state(nasr_ctx* h, BatchSlot* s, int st);
int slot_commit(nasr_ctx* h, BatchSlot* s);
int upload(nasr_ctx* h, const float* feats, const int32_t* seq_len, const int32_t* labels, const int32_t* label_len,
           int B, int T, int Lmax, const float* centre = nullptr, const float* pad_value = nullptr, int ctx = 0,
           int ncep = 0);
BatchSlot* slot_of_ticket(nasr_ctx* h, int ticket);
int stage(nasr_ctx* h, const float* feats, const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B,
          int T, int Lmax, const float* centre, const float* pad_value, int ctx, int ncep, int* ticket);

inline float* dout_of(nasr_ctx* h, int) { return h->dout.as<float>(); }
inline float* dg_of(nasr_ctx* h, int) { return h->dgbuf.as<float>(); }
// what weight_grads(l) reads of layer l's dG: with the overlap on, odd layers have copies of their own (the main stream
// is rewriting the others for layer l-1 while the side stream still reads these)
inline bool wg_alt(const nasr_ctx* h, int l) { return h->wg_overlap && (l & 1); }
inline unsigned char* gttp_of(nasr_ctx* h, int l) { return (wg_alt(h, l) ? h->GTTP2 : h->GTTP).as<unsigned char>(); }
inline float* csws_of(nasr_ctx* h, int l) { return (wg_alt(h, l) ? h->csws2 : h->csws).as<float>(); }
inline nasr_ctx::SV& gc_of(nasr_ctx* h, int l) { return wg_alt(h, l) ? h->sc_gc2 : h->sc_gc; }
// input of LSTM layer l: the features, the last pre-dense stage's output, or the layer below
inline const float* lstm_input(nasr_ctx* h, int l) {
  if (l > 0) return h->outb[l - 1].as<float>();
  return h->npre ? h->Ybuf[h->npre - 1].as<float>() : h->X0.as<float>();
}

// ---- nasr_pass.hip
int forward(nasr_ctx* h);
CtcDims ctc_dims(nasr_ctx* h);
int ctc_forward(nasr_ctx* h);
int backward(nasr_ctx* h);
int fetch_logits(nasr_ctx* h, float* logits_out);

}  // namespace nasr_impl
