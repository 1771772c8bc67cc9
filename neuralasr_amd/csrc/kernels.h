// kernels.h — launch wrappers of the gfx950 kernels behind libnasr.so (internal, C++).
// Data layout in HBM (all fp32, time-major, padded):
//   rows   r = t*Bp + b           Bp = round_up(B,16)
//   X0     [R][Fp]                Fp = round_up(F,32)
//   gates  [R][D*N4]              N4 = 4*Hp, Hp = round_up(H,64); column d*N4 + 4*j + g (gate-interleaved)
//   out,c  [R][D*Hp]
//   logits [T'*Bp][Cp]            Cp = round_up(C,32)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nasr {

// Fault-injection hooks of the tests (NASR_PERSIST_FAULT, NASR_WIDE_FAULT, ...): honoured only in a process that had
// NASR_TEST_HOOKS=1 in its environment when the library first asked - a stray variable in a production run does nothing,
// and a production launch costs one cached bool instead of getenv calls.
const char* test_hook(const char* name);

// ---- GEMM (gemm.hip): C[M,N] = opA[M,K] * opB[K,N] (+bias[n]) on v_mfma_f32_32x32x2_f32 ----
struct GemmDesc {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;          // K % 4 == 0, M % 4 == 0, N % 4 == 0
  int lda, ldb, ldc;    // element strides (multiples of 4)
  bool a_col;           // false: A(m,k) = A[row(m)*lda + k]      true: A(m,k) = A[row(k)*lda + m]
  bool b_col;           // false: B(k,n) = B[k*ldb + n]           true: B(k,n) = B[n*ldb + k]
  const int* a_map;     // logical->physical row of A (-1 = zero row); NULL: physical = logical + a_shift
  int a_shift;          // row shift applied when a_map == NULL
  int a_rows;           // physical rows of A (rows outside [0,a_rows) read as zero)
  const int* c_map;     // logical->physical row of C (-1 = skip); NULL: identity
  const float* bias;    // per-n bias or NULL
  int split_k;          // >=1; >1 writes partial slabs to `slabs` and reduces them into C
  float* slabs;         // workspace of split_k*M*N floats (only when split_k > 1)
};
void launch_gemm(const GemmDesc& g, hipStream_t st);
int gemm_pick_split(int M, int N, int K);

// ---- fp32-accurate GEMM from pre-split, pre-tiled fp16 operands (gemm_tph.hip): C[M,N] = A[M,K] * B[N,K]^T (+bias) ----
// A "tiled planes" operand holds the two fp16 parts of a scaled fp32 matrix [rows][K] as 1 KiB tiles
// TPH[row/32][k/16][part][32 rows x 16 k]; tph_bytes() sizes it, launch_tph_split2() fills it from fp32.
int tp_split2_parts(int rows);           // 64-row partial column sums the split pass can emit
int gemm_tp_tile_rows(int M);            // 256, or 192 where 256-row tiles would leave > 10 % of their rows empty
// Every operand row carries a power-of-two scale (constant along the contraction) that brings its largest magnitude into
// [2^14, 2^15); the epilogue multiplies by the inverse scales of the output's row and column.
size_t tph_bytes(int rows, int K);
size_t tph_scale_ws_floats(int rows, int K);
void launch_tph_scales(const float* src, int rows, int K, int ld, float* row_scale, float* row_inv, float* col_scale,
                       float* col_inv, float* ws, hipStream_t st);
// several matrices in two launches (column scales always, row scales when row_scale != NULL); ws:
// tph_scale_batch_ws_floats(jobs, n) floats
constexpr int TPH_MAX_JOBS = 16;
struct TphScaleJob { const float* src; int rows, K, ld; float *row_scale, *row_inv, *col_scale, *col_inv; };
size_t tph_scale_batch_ws_floats(const TphScaleJob* jobs, int n);
void launch_tph_scales_batch(const TphScaleJob* jobs, int n, float* ws, hipStream_t st);
// scales from partial maxima somebody else took: rowpart [nrp][rows], colpart [ncp][K] (either pair may be NULL), one launch
void launch_tph_scales_from_parts(const float* rowpart, int nrp, int rows, float* row_scale, float* row_inv, const float* colpart,
                                  int ncp, int K, float* col_scale, float* col_inv, hipStream_t st);
void launch_fill(float* p, float v, int n, hipStream_t st);
// one pass over src [rows][K]: tpN = planes of src (scale per src row: row_scale[] or the constant rs), tpT = planes of its
// transpose (scale per src column: col_scale[] or cs); either may be NULL; colpart (or NULL): tp_split2_parts(rows) x K
// partial column sums of src, finished by launch_colsum_parts
// rowmap (or NULL): logical row i = physical row rowmap[i] of src (-1: zero row); rowmap2: the same for the columns from col2 on
void launch_tph_split2(const float* src, unsigned char* tpN, unsigned char* tpT, int rows, int K, int ld,
                       const float* row_scale, float rs, const float* col_scale, float cs, float* colpart, hipStream_t st,
                       const int* rowmap = nullptr, const int* rowmap2 = nullptr, int col2 = 0);
void launch_gather_rows(float* dst, const float* src, const int* map, int n, float fill, hipStream_t st);
struct GemmTPHDesc {
  const unsigned char* A;   // TPH of [>= M rows][K_A], starting at the first row block used
  const unsigned char* B;   // TPH of [>= N rows][K_B]
  float* C;
  int M, N, K;              // M, N multiples of 4 (row blocks past M / N read as zero), K = contraction length
  int nkbA, nkbB;           // k-blocks per row block in A / B: ceil(K_A/16), ceil(K_B/16)
  int ldc;
  int a_kshift;             // multiple of 16: A is read at k + a_kshift, zero outside [0, K_A)
  const float* bias;        // per-n, only with split_k == 1
  const float* a_inv;       // [M] inverse scales of A's rows
  const float* b_inv;       // [N] inverse scales of B's rows
  int split_k;              // > 1: partial slabs + reduction; needs ldc == N and no bias
  float* slabs;             // split_k * M * N floats
  int tile_rows;            // 0: gemm_tp_tile_rows(M); 192 / 256 forces the block tile (tools/gemmbench.hip)
  // nbatch == 2: a second product of the same shape in the same launch (the two directions' recurrent weight gradients):
  // its operands start a_bstride / b_bstride BYTES after A / B, its result c_bstride floats after C, its A shift is
  // a_kshift1.  With split_k > 1: c_bstride must be M * N (one reduction covers both) and slabs hold 2 * split_k * M * N.
  int nbatch;
  size_t a_bstride, b_bstride;
  int64_t c_bstride, ainv_bstride, binv_bstride;
  int a_kshift1;
  bool side;                // the 3-wave 128 x 192 instantiation that fits beside a persistent-recurrence workgroup
  const int* c_map;         // (or NULL) output row m goes to row c_map[m] of C (-1: dropped); nbatch == 1 only
};
hipError_t gemm_tph_prepare();
int gemm_tph_pick_split(int M, int N, int K, int nbatch = 1, bool side = false);
void launch_gemm_tph(const GemmTPHDesc& g, hipStream_t st);

// ---- LSTM recurrence (lstm.hip) ----
struct LstmDims {
  int T, B, Bp, H, Hp, D;   // D directions
};
void launch_pack_feats(const float* feats_bm, float* X0, int B, int Bp, int T, int F, int Fp, hipStream_t st);
void launch_expand_context(const float* centre, const float* pad, const int* seq_len, float* X0, int B, int Bp, int T,
                           int ctx, int ncep, int Fp, hipStream_t st);
// repack canonical U [Hp][N4] of every (layer,dir) into the forward / backward MFMA B-operand images
void launch_repack_u(const float* U, float* Uf, float* Ub, int Hp, hipStream_t st);
void launch_lstm_fwd_step(const LstmDims& dm, int s, const float* Uf, const float* hin, float* hout, float* gates,
                          float* cbuf, float* out, const int* seq_len, float forget_bias, hipStream_t st);
// BPTT step: pin/pout = [D][lstm_bwd_partials(Hp)][Bp][Hp] partial sums handed launch to launch, dcin/dcout = [D][Bp][Hp]
int lstm_bwd_partials(int Hp);
void launch_lstm_bwd_step(const LstmDims& dm, int s, const float* Ub, const float* pin, float* pout,
                          const float* gates, float* dgbuf, const float* cbuf, const float* dout, const float* dcin,
                          float* dcout, const int* seq_len, hipStream_t st);

// ---- persistent recurrence (lstm_persist.hip): one launch per layer pass, one XCD per (direction, utterance slice) ----
struct PersistCtl {            // device words, zeroed before every launch
  unsigned xcc_count[8];       // ticket per XCD: member index of a workgroup inside its group
  unsigned error;              // bit 0: a bounded spin gave up, bit 1: placement is not 32 workgroups on each of 8 XCDs
  unsigned pad[23];
  unsigned flags[8 * 128];     // per group: forward 32 words (one per member), BPTT 128 (member*4 + wave)
#if defined(NASR_PSTAMP) && NASR_PSTAMP
  unsigned stamps[256][12];    // diagnostic build only (tools/persistbench): wave 0's phase cycles of EVERY workgroup
#endif
};
bool persist_supported(int Hp);
size_t persist_image_floats(int Hp, bool bwd);   // floats of one direction's operand image
size_t persist_xch_floats(int Hp);               // floats of the exchange buffer (shared by forward and BPTT)
size_t persist_hx_bytes(int Hp);                 // its head, the forward kernel's h buffers: cleared before every forward launch
size_t persist_px_bytes();                       // the BPTT kernel's partial-sum buffers: cleared before every BPTT launch
hipError_t persist_prepare();                    // once per process: raise the kernels' dynamic-LDS limit
// all n (layer, direction) matrices in one launch: matrix k is P + offs[k], its images Upf + k*image_floats(fwd) / Upb + k*...(bwd)
constexpr int PERSIST_MAX_MATS = 16;
// col_scale (or NULL): [n][4*Hp] power-of-two scales of every matrix's columns -> the forward image holds two fp16 planes
void launch_repack_persist(const float* P, const int64_t* offs, int n, float* Upf, float* Upb, int Hp, const float* col_scale,
                           hipStream_t st);
// Upf/Upb: [D] images of this layer; xch: persist_xch_floats(Hp) floats; ctl: one PersistCtl (zeroed by the launcher);
// sticky: host-mapped word that receives the error code of an aborted launch (or NULL); fault: device float set to 1
// by an aborted launch (or NULL) - the engine keeps it behind the gradients so that it is all-reduced with them
// cinv (or NULL): [D][4*Hp] inverse column scales of this layer's matrices = the fp16 form of the forward image
void launch_lstm_persist_fwd(const LstmDims& dm, const float* Upf, const float* cinv, float* gates, float* cbuf,
                             float* out, const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky, float* fault,
                             float forget_bias, hipStream_t st, bool ctl_zeroed = false);   // ctl_zeroed: the caller cleared *ctl AND the first persist_hx_bytes of xch
void launch_lstm_persist_bwd(const LstmDims& dm, const float* Upb, const float* gates, float* dgbuf, const float* cbuf,
                             const float* dout, const int* seq_len, float* xch, PersistCtl* ctl, unsigned* sticky,
                             float* fault, hipStream_t st, bool ctl_zeroed = false, float* rowpart = nullptr,
                             float* colpart = nullptr);   // ctl_zeroed: the caller cleared *ctl AND persist_px_bytes of xch
// rowpart [D*32][T*Bp], colpart [8/D][D*4Hp] (or NULL): partial maxima of |dG| per frame row / per gate column, written by the
// kernel's memory wave; launch_tph_scales_from_parts turns them into operand scales (persist_dgmax_floats sizes both)
size_t persist_dgmax_floats(int T, int Bp, int Hp, int D);
void persist_set_bwd_lean(bool lean);            // BPTT launches declare 24 KB of LDS instead of 96 KB (process-wide)

// ---- wide persistent forward recurrence (lstm_wide.hip): Hp = 2048, one launch per direction over all 256 CUs ----
struct WideCtl {               // device words, zeroed before every launch
  unsigned xcc_count[8];
  unsigned error;              // bit 0: a bounded spin gave up, bit 1: placement is not 32 workgroups on each of 8 XCDs,
                               // bit 2 (BPTT): dG left the fp16 range of its planes
  unsigned pad[23];
  unsigned hflag[8 * 32];      // [XCD x][member]: timesteps whose h this workgroup has published
  unsigned stamps[160];        // diagnostic build (NASR_WSTAMP, tools/widebench): phase cycles of two workgroups' waves
};
struct WideGeom {
  int T, Bp, Hp, D, d;
  int inject;                  // test hook (NASR_WIDE_FAULT=s): workgroup (0,0) treats the poll of step s as timed out
  int scale_shift;             // test hook (NASR_WIDE_SCALE_SHIFT): added to the exponent of the BPTT's dG scales
  float* fault;
};
bool wide_supported(int Hp, int Bp);
size_t wide_image_bytes(int Hp);     // one direction's operand image
size_t wide_hx_bytes(int Bp);        // h exchange buffer
size_t wide_part_bytes(int Bp);      // partial-sum exchange buffer
hipError_t wide_prepare();           // once per process: raise the kernels' dynamic-LDS limit
// U [Hp][4Hp] canonical, cs [4Hp] power-of-two column scales -> Uw (wide_image_bytes)
void launch_repack_wide(const float* U, const float* cs, void* Uw, int Hp, hipStream_t st);
// direction d of one layer: Uw / cinv ([4Hp], 1 / cs) of that direction; gates/cbuf/out frame-indexed as everywhere
void launch_lstm_wide_fwd(const LstmDims& dm, int d, const void* Uw, const float* cinv, float* gates, float* cbuf,
                          float* out, const int* seq_len, void* hx, float* part, WideCtl* ctl, unsigned* sticky,
                          float* fault, float forget_bias, hipStream_t st);

// BPTT of the same layer: Uwb = the backward image (U^T fragments under per-row scales rs [Hp]; rinv = 1 / rs), srow [D][Bp]
// = largest |dOut| per (direction, utterance) from launch_wide_row_scales (the kernel derives its power-of-two dG scale); inbox = the forward kernel's partial-sum
// buffer (wide_part_bytes), px = wide_px_bytes
size_t wide_px_bytes(int Bp);
void launch_repack_wide_bwd(const float* U, const float* rs, void* Uwb, int Hp, hipStream_t st);
void launch_wide_row_scales(const LstmDims& dm, const float* dout, const int* seq_len, float* srow, hipStream_t st);
void launch_lstm_wide_bwd(const LstmDims& dm, int d, const void* Uwb, const float* rinv, const float* srow,
                          const float* gates, float* dgbuf, const float* cbuf, const float* dout, const int* seq_len,
                          float* inbox, void* px, WideCtl* ctl, unsigned* sticky, float* fault, hipStream_t st);

// ---- DeepSpeech dense stages (dense.hip): clipped ReLU + hash-defined dropout, in place ----
void launch_dense_act(float* z, int R, int Bp, int B, int W, int ld, float clip, float p, uint32_t seed, uint32_t counter,
                      int stage, hipStream_t st);
void launch_dense_act_bwd(float* dy, const float* y, int64_t n, float clip, float p, hipStream_t st);

// ---- CTC (ctc.hip) ----
struct CtcDims {
  int Tp;      // logit frames T'
  int B, Bp, C, Cp, Lmax;
  int KS;      // states per lane: ceil((2*Lmax+1)/64)
  int Tws;     // rows of the alpha/beta workspace per utterance (T + 8: a group of up to 8 frames may run past the last one)
  // workspaces of the engineered lattice (ctc.hip (2b)), or NULL (the plain one only):
  float* lprobs = nullptr;  // [T'][Bp][Cp] emission rows log2 y(t,k) (launch_ctc_logz writes them)
  double* goff = nullptr;   // [B][2][Tws/4 + 3] column offsets per walk and group of 4 frames
};
void launch_ctc_logz(const CtcDims& d, const float* logits, const int* seq_len, float* logz, hipStream_t st);
void launch_ctc_alpha_beta(const CtcDims& d, const float* logits, const float* logz, const int* labels,
                           const int* label_len, const int* seq_len, float* alpha, float* beta, double* aoff,
                           double* boff, float* nll, double* logp, hipStream_t st);
// in place: logits -> d(mean nll)/dlogits
// cstart [B][C+1], cpos [B][Lmax]: the label positions of every utterance sorted by class (ascending inside a class)
void launch_ctc_grad(const CtcDims& d, float* logits, const float* logz, const int* label_len, const int* seq_len,
                     const int* cstart, const int* cpos, const float* alpha, const float* beta, const double* aoff,
                     const double* boff, const double* logp, float scale, hipStream_t st);
void launch_mean(const float* v, int n, float* out, hipStream_t st);
void launch_greedy(const CtcDims& d, const float* logits, const int* seq_len, int* argmax_ws, int* ids, int* lens,
                   hipStream_t st);

// ---- optimiser / reductions (optim.hip) ----
// Adam's step count and the step size derived from it, on the device: a launch whose `fault` word (device float, or NULL)
// is non-zero is a no-op AND leaves the count alone, so a void step never enters the bias correction.
struct AdamDev { long long step; float lr_t; int applied; };
void launch_adam(float* p, float* m, float* v, const float* g, int64_t n, AdamDev* state, float lr, float beta1, float beta2,
                 float eps, float gscale, const float* fault, hipStream_t st);
// out[n] = sum_r M[r*ld + n], deterministic two-stage; ws holds 32*N floats
void launch_colsum(const float* M, int R, int N, int ld, float* out, float* ws, hipStream_t st);
void launch_colsum_parts(const float* part, int nparts, int N, float* out, hipStream_t st);   // out[n] = sum_k part[k][n]
void launch_reduce_slabs(const float* slabs, int S, int64_t n, float* out, hipStream_t st);
void launch_reduce_slabs_rows(const float* slabs, int S, int M, int N, int ldc, const int* map, float* out, hipStream_t st);
// diagnostics: nblocks x 256 threads sweep buf[0..n) `passes` times with 16-byte loads / stores, data unchanged
void launch_ring_standin(float* buf, int64_t n, int nblocks, int passes, hipStream_t st);

}  // namespace nasr
