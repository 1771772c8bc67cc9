/* nasr_cref.c — plain C (OpenMP) restatement of NeuralASR's CTC training step, fp32 with an fp64 CTC lattice.
 *
 * TEST INFRASTRUCTURE ONLY: a second CPU checker beside oracle/nasr_oracle.py and the timed `cpu_baseline`
 * of bench.py.  Nothing under neuralasr_amd/ links or loads it.  PARITY STATUS: unpinned against TensorFlow
 * (not in this image; the reference ships no fixtures for this path); pinned against the NumPy oracle, which is
 * pinned by alignment enumeration, finite differences and torch (tests/test_oracle_pins.py, tests/test_cref.py).
 *
 * Reference call sites followed (relative to /root/reference):
 *   networks/bilstm_ctc_net.py:17-48   BasicLSTMCell x2 in bidirectional_dynamic_rnn, stack-reshape, W/b, [2T,B,C]
 *   networks/lstm_ctc_net.py:17-43     MultiRNNCell of LSTMCell, dynamic_rnn
 *   networks/deepspeech.py:35-127      clipped-ReLU dense stages with dropout around the (Bi)LSTM (num_pre / post_width)
 *   networks/tfnetwork.py:58-59        tf.nn.ctc_loss + reduce_mean
 * TF op semantics: SURVEY.md Appendix A.1-A.4 (gate order i,j,f,o; forget_bias inside the sigmoid; zero output and
 * carried state past seq_len; bw direction over reverse_sequence; blank = C-1; beta excludes the emission at t).
 *
 * Parameter / gradient vectors use TF variable order: per layer (fw kernel [I+H,4H], fw bias [4H], bw kernel,
 * bw bias) or (kernel, bias); then W [Hin,C], b [C].  DeepSpeech family (creation order of networks/deepspeech.py):
 * b1,h1,b2,h2,b3,h3, the cells, b5,h5, b6,h6.  Its dropout masks are the counter hash of oracle/nasr_oracle.py
 * (dropout_mask): element idx = row*W + j of stage `layer` is kept iff lowbias32(idx ^ key) >> 8 >= floor(p * 2^24).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int32_t feature_size, hidden, num_layers, bidirectional, merge; /* merge: 0 none, 1 stack_reshape, 2 concat */
  int32_t num_classes;
  float forget_bias;
  /* DeepSpeech family; all zero for the plain (Bi)LSTM-CTC nets */
  int32_t num_pre, pre_width[3], post_width;
  float relu_clip, dropout[4];
  uint32_t drop_seed, drop_counter;
  int32_t use_dropout;
} cref_spec;

static inline int drop_keep(uint32_t idx, uint32_t key, uint32_t thr) {
  uint32_t x = idx ^ key;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return (x >> 8) >= thr;
}
static inline uint32_t drop_key(const cref_spec* sp, int layer) {
  return sp->drop_seed + 0x9E3779B9u * (uint32_t)(layer + 1) + 0x85EBCA6Bu * sp->drop_counter;
}
/* Y = dropout(min(relu(Z), clip)) in place on a copy: Z [R][W] -> Y [R][W] */
static void dense_act(const cref_spec* sp, int layer, const float* Z, float* Y, size_t R, int W) {
  const float p = sp->use_dropout ? sp->dropout[layer] : 0.f, clip = sp->relu_clip;
  const uint32_t key = drop_key(sp, layer), thr = (uint32_t)floor((double)p * 16777216.0);
  const float inv = 1.f / (1.f - p);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < R; ++r)
    for (int j = 0; j < W; ++j) {
      float a = Z[r * W + j];
      a = a > 0.f ? (a < clip ? a : clip) : 0.f;
      if (p > 0.f) a = drop_keep((uint32_t)(r * W + j), key, thr) ? a * inv : 0.f;
      Y[r * W + j] = a;
    }
}
/* dZ = dY * keep/(1-p) * [0 < Z < clip], in place on dY */
static void dense_act_bwd(const cref_spec* sp, int layer, const float* Z, float* dY, size_t R, int W) {
  const float p = sp->use_dropout ? sp->dropout[layer] : 0.f, clip = sp->relu_clip;
  const uint32_t key = drop_key(sp, layer), thr = (uint32_t)floor((double)p * 16777216.0);
  const float inv = 1.f / (1.f - p);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < R; ++r)
    for (int j = 0; j < W; ++j) {
      const float z = Z[r * W + j];
      float g = (z > 0.f && z < clip) ? dY[r * W + j] : 0.f;
      if (p > 0.f) g = drop_keep((uint32_t)(r * W + j), key, thr) ? g * inv : 0.f;
      dY[r * W + j] = g;
    }
}
static void colsum(const float* M, size_t R, int W, float* out) {
#pragma omp parallel for schedule(static)
  for (int j = 0; j < W; ++j) { double s = 0; for (size_t r = 0; r < R; ++r) s += M[r * W + j]; out[j] = (float)s; }
}

static inline float sigm(float x) { return 1.f / (1.f + expf(-x)); }

/* C[M,N] = A[M,K] * B[K,N] (+ bias[n]) */
static void gemm_nn(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    const float* bias) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    float* c = C + (size_t)i * ldc;
    for (int j = 0; j < N; ++j) c[j] = bias ? bias[j] : 0.f;
    for (int k = 0; k < K; ++k) {
      const float a = A[(size_t)i * lda + k];
      if (a == 0.f) continue;
      const float* b = B + (size_t)k * ldb;
      for (int j = 0; j < N; ++j) c[j] += a * b[j];
    }
  }
}

/* C[M,N] = A[R,M]^T * B[R,N]   (contraction over rows) */
static void gemm_tn(const float* A, const float* B, float* C, int M, int N, int R, int lda, int ldb, int ldc) {
#pragma omp parallel for schedule(static)
  for (int i0 = 0; i0 < M; i0 += 4) {
    const int i1 = i0 + 4 < M ? i0 + 4 : M;
    for (int i = i0; i < i1; ++i) memset(C + (size_t)i * ldc, 0, (size_t)N * sizeof(float));
    for (int r = 0; r < R; ++r) {
      const float* b = B + (size_t)r * ldb;
      for (int i = i0; i < i1; ++i) {
        const float a = A[(size_t)r * lda + i];
        if (a == 0.f) continue;
        float* c = C + (size_t)i * ldc;
        for (int j = 0; j < N; ++j) c[j] += a * b[j];
      }
    }
  }
}

/* C[M,N] (+)= A[M,K] * B[N,K]^T */
static void gemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    int accumulate) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < M; ++i) {
    const float* a = A + (size_t)i * lda;
    for (int j = 0; j < N; ++j) {
      const float* b = B + (size_t)j * ldb;
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += a[k] * b[k];
      if (accumulate) C[(size_t)i * ldc + j] += s; else C[(size_t)i * ldc + j] = s;
    }
  }
}

typedef struct {
  int I, H;
  const float* kernel; /* [I+H][4H] */
  const float* bias;
  float *xp, *act, *c, *out; /* [T*B][4H], [T*B][4H] (si,tj,sf,so), [T*B][H], [T*B][H] */
} dir_t;

/* one direction of (bidirectional_)dynamic_rnn; X time-major [T*B][I] */
static void lstm_dir_forward(dir_t* d, const float* X, const int32_t* len, int B, int T, int reverse, float fb) {
  const int H = d->H, I = d->I, N4 = 4 * H;
  gemm_nn(X, d->kernel, d->xp, T * B, N4, I, I, N4, N4, d->bias);
  const float* U = d->kernel + (size_t)I * N4;
  float* hcur = (float*)calloc((size_t)B * H, sizeof(float));
  float* hnew = (float*)calloc((size_t)B * H, sizeof(float));
  float* ccur = (float*)calloc((size_t)B * H, sizeof(float));
  memset(d->out, 0, (size_t)T * B * H * sizeof(float));
  memset(d->c, 0, (size_t)T * B * H * sizeof(float));
  memset(d->act, 0, (size_t)T * B * N4 * sizeof(float));
#pragma omp parallel
  for (int s = 0; s < T; ++s) {
#pragma omp for schedule(static)
    for (int j0 = 0; j0 < H; j0 += 8) {
      const int j1 = j0 + 8 < H ? j0 + 8 : H;
      for (int b = 0; b < B; ++b) {
        if (s >= len[b]) {
          for (int j = j0; j < j1; ++j) hnew[(size_t)b * H + j] = hcur[(size_t)b * H + j];
          continue;
        }
        const int tb = reverse ? len[b] - 1 - s : s;
        const size_t r = (size_t)tb * B + b;
        float g[4][8];
        for (int q = 0; q < 4; ++q)
          for (int j = j0; j < j1; ++j) g[q][j - j0] = d->xp[r * N4 + q * H + j];
        const float* h = hcur + (size_t)b * H;
        for (int k = 0; k < H; ++k) {
          const float hv = h[k];
          const float* u = U + (size_t)k * N4;
          for (int q = 0; q < 4; ++q)
            for (int j = j0; j < j1; ++j) g[q][j - j0] += hv * u[q * H + j];
        }
        for (int j = j0; j < j1; ++j) {
          const float si = sigm(g[0][j - j0]), tj = tanhf(g[1][j - j0]);
          const float sf = sigm(g[2][j - j0] + fb), so = sigm(g[3][j - j0]);
          const float cn = ccur[(size_t)b * H + j] * sf + si * tj;
          const float hn = tanhf(cn) * so;
          float* a = d->act + r * N4;
          a[j] = si; a[H + j] = tj; a[2 * H + j] = sf; a[3 * H + j] = so;
          d->c[r * H + j] = cn;
          d->out[r * H + j] = hn;
          ccur[(size_t)b * H + j] = cn;
          hnew[(size_t)b * H + j] = hn;
        }
      }
    }
#pragma omp single
    { float* t = hcur; hcur = hnew; hnew = t; }
  }
  free(hcur); free(hnew); free(ccur);
}

/* BPTT of one direction: dout [T*B][H] -> dG [T*B][4H] (frame indexed, zero at masked frames) */
static void lstm_dir_backward(const dir_t* d, const float* dout, float* dG, const int32_t* len, int B, int T,
                              int reverse) {
  const int H = d->H, I = d->I, N4 = 4 * H;
  const float* U = d->kernel + (size_t)I * N4;
  float* dh = (float*)calloc((size_t)B * H, sizeof(float));
  float* dhn = (float*)calloc((size_t)B * H, sizeof(float));
  float* dc = (float*)calloc((size_t)B * H, sizeof(float));
  memset(dG, 0, (size_t)T * B * N4 * sizeof(float));
#pragma omp parallel
  for (int s = T - 1; s >= 0; --s) {
    /* pointwise: dG of this step from dh (recurrent) + dout */
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      if (s >= len[b]) continue;
      const int tb = reverse ? len[b] - 1 - s : s;
      const size_t r = (size_t)tb * B + b;
      const size_t rp = reverse ? r + B : r - B;
      const float* a = d->act + r * N4;
      float* g = dG + r * N4;
      for (int j = 0; j < H; ++j) {
        const float dht = dh[(size_t)b * H + j] + dout[r * H + j];
        const float si = a[j], tj = a[H + j], sf = a[2 * H + j], so = a[3 * H + j];
        const float tc = tanhf(d->c[r * H + j]);
        const float cp = s > 0 ? d->c[rp * H + j] : 0.f;
        const float dct = dc[(size_t)b * H + j] + dht * so * (1.f - tc * tc);
        g[j] = dct * tj * si * (1.f - si);
        g[H + j] = dct * si * (1.f - tj * tj);
        g[2 * H + j] = dct * cp * sf * (1.f - sf);
        g[3 * H + j] = dht * tc * so * (1.f - so);
        dc[(size_t)b * H + j] = dct * sf;
      }
    }
    /* dh_prev[b][k] = sum_n dG[b][n] U[k][n] */
#pragma omp for schedule(static)
    for (int k = 0; k < H; ++k) {
      const float* u = U + (size_t)k * N4;
      for (int b = 0; b < B; ++b) {
        if (s >= len[b]) { dhn[(size_t)b * H + k] = dh[(size_t)b * H + k]; continue; }
        const int tb = reverse ? len[b] - 1 - s : s;
        const float* g = dG + ((size_t)tb * B + b) * N4;
        float acc = 0.f;
        for (int n = 0; n < N4; ++n) acc += g[n] * u[n];
        dhn[(size_t)b * H + k] = acc;
      }
    }
#pragma omp single
    { float* t = dh; dh = dhn; dhn = t; }
  }
  free(dh); free(dhn); free(dc);
}

static double lse2(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}

/* tf.nn.ctc_loss for one utterance; logits rows stride `ls`; grad written in place of a separate buffer */
static double ctc_one(const float* lg, size_t ls, int Tb, int C, const int32_t* lab, int L, float* grad, float scale) {
  const int S = 2 * L + 1, blank = C - 1;
  double* lp = (double*)malloc((size_t)Tb * C * sizeof(double));
  double* al = (double*)malloc((size_t)Tb * S * sizeof(double));
  double* be = (double*)malloc((size_t)Tb * S * sizeof(double));
  for (int t = 0; t < Tb; ++t) {
    const float* x = lg + (size_t)t * ls;
    double m = x[0];
    for (int c = 1; c < C; ++c) if (x[c] > m) m = x[c];
    double z = 0;
    for (int c = 0; c < C; ++c) z += exp(x[c] - m);
    z = m + log(z);
    for (int c = 0; c < C; ++c) lp[(size_t)t * C + c] = x[c] - z;
  }
#define EXT(s) (((s) & 1) ? lab[(s) >> 1] : blank)
  for (int i = 0; i < Tb * S; ++i) al[i] = be[i] = -INFINITY;
  al[0] = lp[blank];
  if (S > 1) al[1] = lp[EXT(1)];
  for (int t = 1; t < Tb; ++t)
    for (int s = 0; s < S; ++s) {
      double v = al[(size_t)(t - 1) * S + s];
      if (s >= 1) v = lse2(v, al[(size_t)(t - 1) * S + s - 1]);
      if (s >= 2 && EXT(s) != blank && EXT(s) != EXT(s - 2)) v = lse2(v, al[(size_t)(t - 1) * S + s - 2]);
      al[(size_t)t * S + s] = v + lp[(size_t)t * C + EXT(s)];
    }
  be[(size_t)(Tb - 1) * S + S - 1] = 0;
  if (S > 1) be[(size_t)(Tb - 1) * S + S - 2] = 0;
  for (int t = Tb - 2; t >= 0; --t)
    for (int s = 0; s < S; ++s) {
      double v = be[(size_t)(t + 1) * S + s] + lp[(size_t)(t + 1) * C + EXT(s)];
      if (s + 1 < S) v = lse2(v, be[(size_t)(t + 1) * S + s + 1] + lp[(size_t)(t + 1) * C + EXT(s + 1)]);
      if (s + 2 < S && EXT(s + 2) != blank && EXT(s + 2) != EXT(s))
        v = lse2(v, be[(size_t)(t + 1) * S + s + 2] + lp[(size_t)(t + 1) * C + EXT(s + 2)]);
      be[(size_t)t * S + s] = v;
    }
  double logp = -INFINITY;
  for (int s = 0; s < S; ++s) logp = lse2(logp, al[s] + be[s]);
  double* post = (double*)malloc((size_t)C * sizeof(double));
  for (int t = 0; t < Tb; ++t) {
    for (int c = 0; c < C; ++c) post[c] = 0;
    for (int s = 0; s < S; ++s) post[EXT(s)] += exp(al[(size_t)t * S + s] + be[(size_t)t * S + s] - logp);
    for (int c = 0; c < C; ++c) grad[(size_t)t * ls + c] = (float)((exp(lp[(size_t)t * C + c]) - post[c]) * scale);
  }
#undef EXT
  free(post); free(lp); free(al); free(be);
  return -logp;
}

int64_t cref_param_count(const cref_spec* sp) {
  const int D = sp->bidirectional ? 2 : 1, H = sp->hidden;
  int64_t n = 0;
  const int F0c = sp->num_pre ? sp->pre_width[sp->num_pre - 1] : sp->feature_size;
  for (int l = 0; l < sp->num_layers; ++l) {
    const int I = l == 0 ? F0c : D * H;
    n += (int64_t)D * ((int64_t)(I + H) * 4 * H + 4 * H);
  }
  const int Pin = (sp->bidirectional && sp->merge == 2) ? 2 * H : H;
  for (int i = 0; i < sp->num_pre; ++i)
    n += (int64_t)(i == 0 ? sp->feature_size : sp->pre_width[i - 1]) * sp->pre_width[i] + sp->pre_width[i];
  if (sp->post_width) n += (int64_t)Pin * sp->post_width + sp->post_width;
  const int Pp = sp->post_width ? sp->post_width : Pin;
  return n + (int64_t)Pp * sp->num_classes + sp->num_classes;
}

void cref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int cref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* loss = mean CTC nll and d loss / d params (TF order).  feats [B,T,F] batch-major.  Returns 0, or -3 when a label
 * needs more frames than seq_len ("Not enough time for target transition sequence"). grads/logits may be NULL. */
int cref_loss_and_grads(const cref_spec* sp, const float* params, const float* feats, const int32_t* seq_len,
                        const int32_t* labels, const int32_t* label_len, int B, int T, int Lmax, float* loss_out,
                        float* nll_out, float* grads, float* logits_out) {
  const int D = sp->bidirectional ? 2 : 1, H = sp->hidden, C = sp->num_classes, L = sp->num_layers, F = sp->feature_size;
  const int sr = sp->bidirectional && sp->merge == 1;
  const int Pin = (sp->bidirectional && sp->merge == 2) ? 2 * H : H;
  const int Tp = sr ? 2 * T : T;
  const size_t R = (size_t)T * B;
  for (int b = 0; b < B; ++b) {
    int rep = 0;
    for (int i = 1; i < label_len[b]; ++i) rep += labels[(size_t)b * Lmax + i] == labels[(size_t)b * Lmax + i - 1];
    if (label_len[b] + rep > seq_len[b]) return -3;
  }
  /* ---- parameter views */
  const int npre = sp->num_pre, post = sp->post_width;
  const int F0 = npre ? sp->pre_width[npre - 1] : F;   /* width the first LSTM layer reads */
  dir_t* dirs = (dir_t*)calloc((size_t)L * D, sizeof(dir_t));
  const float* p = params;
  const float *pb[3] = {0, 0, 0}, *ph[3] = {0, 0, 0};
  int pin_w[3] = {0, 0, 0};
  for (int i = 0; i < npre; ++i) {
    pin_w[i] = i == 0 ? F : sp->pre_width[i - 1];
    pb[i] = p; p += sp->pre_width[i];
    ph[i] = p; p += (size_t)pin_w[i] * sp->pre_width[i];
  }
  for (int l = 0; l < L; ++l)
    for (int d = 0; d < D; ++d) {
      dir_t* q = &dirs[l * D + d];
      q->I = l == 0 ? F0 : D * H; q->H = H;
      q->kernel = p; p += (size_t)(q->I + H) * 4 * H;
      q->bias = p; p += 4 * H;
      q->xp = (float*)malloc(R * 4 * H * sizeof(float));
      q->act = (float*)malloc(R * 4 * H * sizeof(float));
      q->c = (float*)malloc(R * H * sizeof(float));
      q->out = (float*)malloc(R * H * sizeof(float));
    }
  const float *b5 = 0, *h5 = 0;
  if (post) { b5 = p; p += post; h5 = p; p += (size_t)Pin * post; }
  const int Pp = post ? post : Pin;                    /* width the projection reads */
  const float *W, *bproj;
  if (npre || post) { bproj = p; p += C; W = p; }      /* b6, h6 */
  else { W = p; p += (size_t)Pin * C; bproj = p; }
  /* ---- forward */
  float** X = (float**)calloc((size_t)L + 1, sizeof(float*));
  float* feat_tm = (float*)malloc(R * F * sizeof(float));
#pragma omp parallel for
  for (int t = 0; t < T; ++t)
    for (int b = 0; b < B; ++b) memcpy(feat_tm + ((size_t)t * B + b) * F, feats + ((size_t)b * T + t) * F, (size_t)F * sizeof(float));
  /* dense stages in front of the stack: Zpre[i] pre-activations, Ypre[i] outputs */
  float *Zpre[3] = {0, 0, 0}, *Ypre[3] = {0, 0, 0};
  for (int i = 0; i < npre; ++i) {
    const int Wd = sp->pre_width[i];
    Zpre[i] = (float*)malloc(R * Wd * sizeof(float));
    Ypre[i] = (float*)malloc(R * Wd * sizeof(float));
    gemm_nn(i == 0 ? feat_tm : Ypre[i - 1], ph[i], Zpre[i], (int)R, Wd, pin_w[i], pin_w[i], Wd, Wd, pb[i]);
    dense_act(sp, i, Zpre[i], Ypre[i], R, Wd);
  }
  X[0] = npre ? Ypre[npre - 1] : feat_tm;
  for (int l = 0; l < L; ++l) {
    for (int d = 0; d < D; ++d) lstm_dir_forward(&dirs[l * D + d], X[l], seq_len, B, T, d == 1, sp->forget_bias);
    X[l + 1] = (float*)malloc(R * D * H * sizeof(float));
#pragma omp parallel for
    for (size_t r = 0; r < R; ++r)
      for (int d = 0; d < D; ++d) memcpy(X[l + 1] + (r * D + d) * H, dirs[l * D + d].out + r * H, (size_t)H * sizeof(float));
  }
  /* projection input rows: flat[q] for logits row (t', b') */
  const size_t Rp = (size_t)Tp * B;
  float* flat = (float*)malloc(Rp * Pin * sizeof(float));
#pragma omp parallel for
  for (size_t rr = 0; rr < Rp; ++rr) {
    const int tp = (int)(rr / B), bq = (int)(rr % B);
    if (sr) { /* row b'*2T + t' of stack(fw,bw) [2,B,T,H] */
      const int64_t qq = (int64_t)bq * 2 * T + tp;
      const int d = (int)(qq / ((int64_t)B * T));
      const int64_t rem = qq % ((int64_t)B * T);
      const int b = (int)(rem / T), t = (int)(rem % T);
      memcpy(flat + rr * Pin, dirs[(L - 1) * D + d].out + ((size_t)t * B + b) * H, (size_t)H * sizeof(float));
    } else {
      memcpy(flat + rr * Pin, X[L] + rr * Pin, (size_t)Pin * sizeof(float));
    }
  }
  float *Zpost = 0, *Ypost = 0;
  if (post) {
    Zpost = (float*)malloc(Rp * post * sizeof(float));
    Ypost = (float*)malloc(Rp * post * sizeof(float));
    gemm_nn(flat, h5, Zpost, (int)Rp, post, Pin, Pin, post, post, b5);
    dense_act(sp, npre, Zpost, Ypost, Rp, post);
  }
  const float* proj_in = post ? Ypost : flat;
  float* logits = (float*)malloc(Rp * C * sizeof(float));
  gemm_nn(proj_in, W, logits, (int)Rp, C, Pp, Pp, C, C, bproj);
  if (logits_out) memcpy(logits_out, logits, Rp * C * sizeof(float));
  /* ---- CTC */
  float* dlog = (float*)calloc(Rp * C, sizeof(float));
  double total = 0;
  double* nlls = (double*)malloc((size_t)B * sizeof(double));
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b)
    nlls[b] = ctc_one(logits + (size_t)b * C, (size_t)B * C, seq_len[b], C, labels + (size_t)b * Lmax, label_len[b],
                      dlog + (size_t)b * C, 1.f / (float)B);
  for (int b = 0; b < B; ++b) { total += nlls[b]; if (nll_out) nll_out[b] = (float)nlls[b]; }
  if (loss_out) *loss_out = (float)(total / B);
  if (grads) {
    float* g = grads;
    float *gpb[3] = {0, 0, 0}, *gph[3] = {0, 0, 0};
    for (int i = 0; i < npre; ++i) { gpb[i] = g; g += sp->pre_width[i]; gph[i] = g; g += (size_t)pin_w[i] * sp->pre_width[i]; }
    float** gk = (float**)calloc((size_t)L * D, sizeof(float*));
    float** gb = (float**)calloc((size_t)L * D, sizeof(float*));
    for (int l = 0; l < L; ++l)
      for (int d = 0; d < D; ++d) {
        gk[l * D + d] = g; g += (size_t)(dirs[l * D + d].I + H) * 4 * H;
        gb[l * D + d] = g; g += 4 * H;
      }
    float *gb5 = 0, *gh5 = 0;
    if (post) { gb5 = g; g += post; gh5 = g; g += (size_t)Pin * post; }
    float *gW, *gbp;
    if (npre || post) { gbp = g; g += C; gW = g; }
    else { gW = g; g += (size_t)Pin * C; gbp = g; }
    gemm_tn(proj_in, dlog, gW, Pp, C, (int)Rp, Pp, C, C);
    colsum(dlog, Rp, C, gbp);
    float* dflat = (float*)malloc(Rp * Pin * sizeof(float));
    if (post) {
      float* dYp = (float*)malloc(Rp * post * sizeof(float));
      gemm_nt(dlog, W, dYp, (int)Rp, post, C, C, C, post, 0);
      dense_act_bwd(sp, npre, Zpost, dYp, Rp, post);            /* dYp is dZ5 now */
      gemm_tn(flat, dYp, gh5, Pin, post, (int)Rp, Pin, post, post);
      colsum(dYp, Rp, post, gb5);
      gemm_nt(dYp, h5, dflat, (int)Rp, Pin, post, post, post, Pin, 0);
      free(dYp);
    } else {
      gemm_nt(dlog, W, dflat, (int)Rp, Pin, C, C, C, Pin, 0);
    }
    /* gradient wrt the last layer's outputs, per direction [T*B][H] */
    float** dout = (float**)calloc((size_t)D, sizeof(float*));
    for (int d = 0; d < D; ++d) dout[d] = (float*)calloc(R * H, sizeof(float));
#pragma omp parallel for
    for (size_t rr = 0; rr < Rp; ++rr) {
      const int tp = (int)(rr / B), bq = (int)(rr % B);
      if (sr) {
        const int64_t qq = (int64_t)bq * 2 * T + tp;
        const int d = (int)(qq / ((int64_t)B * T));
        const int64_t rem = qq % ((int64_t)B * T);
        const int b = (int)(rem / T), t = (int)(rem % T);
        memcpy(dout[d] + ((size_t)t * B + b) * H, dflat + rr * Pin, (size_t)H * sizeof(float));
      } else {
        for (int d = 0; d < D; ++d) memcpy(dout[d] + rr * H, dflat + rr * Pin + (size_t)d * H, (size_t)H * sizeof(float));
      }
    }
    float* dG = (float*)malloc(R * 4 * H * sizeof(float));
    float* hprev = (float*)malloc(R * H * sizeof(float));
    for (int l = L - 1; l >= 0; --l) {
      const int I = dirs[l * D].I;
      float* dX = (l > 0 || npre) ? (float*)calloc(R * I, sizeof(float)) : NULL;
      for (int d = 0; d < D; ++d) {
        dir_t* q = &dirs[l * D + d];
        lstm_dir_backward(q, dout[d], dG, seq_len, B, T, d == 1);
        gemm_tn(X[l], dG, gk[l * D + d], I, 4 * H, (int)R, I, 4 * H, 4 * H);
        /* h_prev of frame t: out[t-1] (fw) / out[t+1] (bw), zero at the ends */
        memset(hprev, 0, R * H * sizeof(float));
        if (d == 0) memcpy(hprev + (size_t)B * H, q->out, (R - B) * H * sizeof(float));
        else memcpy(hprev, q->out + (size_t)B * H, (R - B) * H * sizeof(float));
        gemm_tn(hprev, dG, gk[l * D + d] + (size_t)I * 4 * H, H, 4 * H, (int)R, H, 4 * H, 4 * H);
        for (int n = 0; n < 4 * H; ++n) { double s = 0; for (size_t r = 0; r < R; ++r) s += dG[r * 4 * H + n]; gb[l * D + d][n] = (float)s; }
        if (dX) gemm_nt(dG, q->kernel, dX, (int)R, I, 4 * H, 4 * H, 4 * H, I, 1);   /* rows 0..I-1 of kernel = Wx */
      }
      if (l > 0) {
        for (int d = 0; d < D; ++d)
#pragma omp parallel for
          for (size_t r = 0; r < R; ++r) memcpy(dout[d] + r * H, dX + r * I + (size_t)d * H, (size_t)H * sizeof(float));
        free(dX);
      } else if (npre) {
        /* dense stages in front of the stack, last to first: dX is d loss / d Ypre[npre-1] */
        float* dY = dX;
        for (int i = npre - 1; i >= 0; --i) {
          const int Wd = sp->pre_width[i];
          dense_act_bwd(sp, i, Zpre[i], dY, R, Wd);
          gemm_tn(i == 0 ? feat_tm : Ypre[i - 1], dY, gph[i], pin_w[i], Wd, (int)R, pin_w[i], Wd, Wd);
          colsum(dY, R, Wd, gpb[i]);
          float* dIn = NULL;
          if (i > 0) {
            dIn = (float*)malloc(R * pin_w[i] * sizeof(float));
            gemm_nt(dY, ph[i], dIn, (int)R, pin_w[i], Wd, Wd, Wd, pin_w[i], 0);
          }
          free(dY);
          dY = dIn;
        }
      }
    }
    free(dG); free(hprev); free(dflat);
    for (int d = 0; d < D; ++d) free(dout[d]);
    free(dout); free(gk); free(gb);
  }
  free(nlls); free(dlog); free(logits); free(flat);
  free(Zpost); free(Ypost);
  for (int l = 1; l <= L; ++l) free(X[l]);            /* X[0] is feat_tm or the last dense output */
  free(X);
  for (int i = 0; i < npre; ++i) { free(Zpre[i]); free(Ypre[i]); }
  free(feat_tm);
  for (int i = 0; i < L * D; ++i) { free(dirs[i].xp); free(dirs[i].act); free(dirs[i].c); free(dirs[i].out); }
  free(dirs);
  return 0;
}
