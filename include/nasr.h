/* nasr.h — C ABI of libnasr.so: the MI355X (gfx950) implementation of NeuralASR's CTC training
 * hot path ((Bi)LSTM stack -> affine projection -> CTC loss/gradient -> Adam), one handle per
 * GPU / process.
 *
 * The reference has no FFI: its hot path is a TensorFlow-1 graph driven from Python
 * (/root/reference/networks/tfnetwork.py).  Each entry point below names the reference
 * interface it replaces; the Python `Network` subclass in neuralasr_amd/networks binds them
 * with ctypes (INTEGRATION.md shows the stub a NeuralASR maintainer would add).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; nasr_last_error() gives the message
 *     (NASR_ERR_INFEASIBLE mirrors TF's "Not enough time for target transition sequence").
 *   - the caller owns every host buffer; the library owns all device memory.
 *   - one host thread per handle; a handle is not re-entrant (tfnetwork.py: one Session).
 *   - features are batch-major float32 [B,T,F] C-contiguous, zero past seq_len (dataset.py:75-77);
 *     labels int32 [B,Lmax] padded with 0; label ids in [0, C-2]; blank = C-1 (A.4).
 *   - flat parameter / gradient order is TF variable order: per layer (fw kernel [I+H,4H] rows
 *     [input;h], gate columns i,j,f,o; fw bias [4H]; bw kernel; bw bias) or (kernel, bias);
 *     then W [Hin,C], b [C]  (SURVEY.md §8b, Appendix A.1).
 *   - there is NO CPU fallback: nasr_create fails when no gfx950 device is usable.
 */
#ifndef NASR_H
#define NASR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NASR_OK 0
#define NASR_ERR_ARG (-1)        /* bad argument / shape */
#define NASR_ERR_HIP (-2)        /* HIP runtime error */
#define NASR_ERR_INFEASIBLE (-3) /* CTC: label needs more frames than seq_len (TF InvalidArgument) */
#define NASR_ERR_STATE (-4)      /* call order (e.g. backward without a resident batch) */

#define NASR_MERGE_NONE 0          /* unidirectional (networks/lstm_ctc_net.py:17-23) */
#define NASR_MERGE_STACK_RESHAPE 1 /* literal BiLstmCTCNet: tf.reshape on the (fw,bw) tuple
                                      (networks/bilstm_ctc_net.py:33,45) */
#define NASR_MERGE_CONCAT 2        /* tf.concat(outputs, 2) (networks/deepspeech.py:103) */

typedef struct nasr_ctx* nasr_handle;

/* Model + optimiser shape.  Replaces the hard-coded locals of create_network
 * (networks/bilstm_ctc_net.py:14, networks/lstm_ctc_net.py:14-15) and the AdamOptimizer
 * defaults of setup_training_network (networks/tfnetwork.py:116-117). */
typedef struct {
  int32_t feature_size;  /* F = (2*numcontext+1)*numcep  (config.py:25) */
  int32_t hidden;        /* H: LSTM units per direction */
  int32_t num_layers;
  int32_t bidirectional; /* 0 | 1 */
  int32_t merge;         /* NASR_MERGE_* */
  int32_t num_classes;   /* C = symbols.counter (networks/tfnetwork.py:18); blank = C-1 */
  float forget_bias;     /* 1.0 (BasicLSTMCell / LSTMCell default) */
  float learning_rate;   /* config.learningrate */
  float beta1, beta2, epsilon; /* 0.9, 0.999, 1e-8 */
  /* DeepSpeech family (networks/deepspeech.py:10-132): clipped-ReLU dense stages with dropout in front of the LSTM
   * stack (layers 1-3: widths n_hidden, n_hidden, 2*n_cell_dim) and one between the stack and the logits (layer 5).
   * All zero for the (Bi)LstmCTCNet models.  With any of them set the variable order is the creation order of
   * deepspeech.py: b1,h1,b2,h2,b3,h3, fw kernel, fw bias, bw kernel, bw bias, b5,h5, b6,h6 (bias BEFORE weight). */
  int32_t num_pre;       /* 0..3 dense stages before the stack */
  int32_t pre_width[3];
  int32_t post_width;    /* 0 = none */
  float relu_clip;       /* 20.0 */
  float dropout[4];      /* drop probability of the pre stages, then of the post stage ([0.05,0.05,0.05] and 0.05) */
} nasr_model_cfg;

/* Phase timings of the last nasr_compute_grads / nasr_apply_adam (HIP events on the handle's
 * stream), milliseconds.  Used by bench.py's roofline object. */
typedef struct {
  float pack_ms;      /* feature transpose+pad into time-major HBM layout */
  float xproj_ms;     /* input-to-hidden GEMMs (all layers) */
  float rec_fwd_ms;   /* forward recurrence: all layers' per-timestep kernels */
  float proj_ctc_ms;  /* projection GEMM + CTC (logZ, alpha/beta, gradient) */
  float proj_bwd_ms;  /* projection backward GEMMs */
  float rec_bwd_ms;   /* BPTT: all layers' per-timestep kernels */
  float wgrad_ms;     /* weight-gradient / input-gradient GEMMs + bias column sums */
  float adam_ms;      /* fused Adam + recurrent-weight repack */
  float total_ms;
  int32_t rec_fwd_launches; /* kernel launches of the recurrence in rec_fwd_ms (T per layer, or 1 when persistent) */
  int32_t rec_bwd_launches;
} nasr_phase_times;

/* ---- lifetime ---------------------------------------------------------------------------
 * nasr_create replaces TensorFlowNetwork.__init__ graph construction + tf.Session
 * (networks/tfnetwork.py:14-43).  `stream` is a hipStream_t to launch on (NULL: the library
 * creates its own); pass torch.cuda.current_stream().cuda_stream to order the handle's work
 * with torch.distributed collectives.  Parameters start at zero: call nasr_set_params. */
int nasr_create(const nasr_model_cfg* cfg, int device_id, void* stream, nasr_handle* out);
int nasr_destroy(nasr_handle h);
/* message of the CALLING THREAD's last failed call on h (h NULL: of a failed nasr_create); valid until that thread's next
 * failing call.  Per thread, so that nasr_stage_batch* on a loader thread and the training thread never read or overwrite
 * each other's text. */
const char* nasr_last_error(nasr_handle h);
const char* nasr_backend(nasr_handle h);    /* "hip-gfx950" */
int nasr_synchronize(nasr_handle h);        /* hipStreamSynchronize on the handle's stream */

/* ---- parameters / optimiser state (replaces tf.train.Saver's view of the variables,
 * networks/tfnetwork.py:40,142-164; TF variable order) ------------------------------------ */
int64_t nasr_param_count(nasr_handle h);
int nasr_num_tensors(nasr_handle h);
/* name (<=63 chars), offset into the flat vector, rows, cols (cols = 1 for vectors) */
int nasr_tensor_info(nasr_handle h, int idx, char name[64], int64_t* offset, int64_t* rows, int64_t* cols);
int nasr_set_params(nasr_handle h, const float* flat, int64_t n);
int nasr_get_params(nasr_handle h, float* flat, int64_t n);
int nasr_set_adam_state(nasr_handle h, const float* m, const float* v, int64_t n, int64_t step);
int nasr_get_adam_state(nasr_handle h, float* m, float* v, int64_t n, int64_t* step);
int nasr_set_learning_rate(nasr_handle h, float lr);

/* ---- one-call entry points over host buffers -------------------------------------------
 * nasr_train_step replaces sess.run([optimizer, loss]) of TensorFlowNetwork.train
 * (networks/tfnetwork.py:183-190) for one tower: forward, CTC, backward, Adam.  loss_out =
 * reduce_mean of the per-utterance CTC NLL (networks/tfnetwork.py:59). */
int nasr_train_step(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                    const int32_t* label_len, int B, int T, int Lmax, float* loss_out);
/* forward only (inference graph, networks/tfnetwork.py:33-37): logits_out is time-major
 * [T',B,C] with T' = nasr_logit_frames(h,T) (2T for STACK_RESHAPE). May be NULL. */
int nasr_forward(nasr_handle h, const float* feats, const int32_t* seq_len, int B, int T, float* logits_out);
int nasr_logit_frames(nasr_handle h, int T);
/* loss of validate()/evaluate() (networks/tfnetwork.py:166-177): forward + CTC, no update.
 * nll_out [B] per-utterance (may be NULL). */
int nasr_loss(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
              const int32_t* label_len, int B, int T, int Lmax, float* loss_out, float* nll_out);
/* parity hook: loss + d loss/d every variable, TF order, no update.  flat_grads_out [param_count]. */
int nasr_loss_and_grads(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                        const int32_t* label_len, int B, int T, int Lmax, float* loss_out, float* nll_out,
                        float* flat_grads_out);
/* tf.nn.ctc_greedy_decoder(merge_repeated=True), the decoder named at networks/tfnetwork.py:62-63:
 * ids_out [B, T'] (row b holds lens_out[b] ids), lens_out [B]. */
int nasr_greedy_decode(nasr_handle h, const float* feats, const int32_t* seq_len, int B, int T,
                       int32_t* ids_out, int32_t* lens_out);

/* ---- data-parallel building blocks (one shard per GPU; replaces make_parallel +
 * average_gradients, networks/tfnetwork.py:72-140).  Typical step on every rank:
 *   nasr_upload_batch(shard) ; nasr_compute_grads ; all-reduce(sum) nasr_grad_device_ptr over
 *   RCCL ; nasr_apply_adam(1/world) ; nasr_get_loss
 * The gradient buffer is one flat fp32 device array of nasr_grad_device_count elements in the
 * library's padded internal layout (identical on every rank; padding elements are always 0).  Its first 32
 * floats are not gradients: the first of them is the step's FAULT word (0, or 1 when this rank's persistent
 * recurrence gave up).  Reduce the whole array: a non-zero sum makes nasr_apply_adam a no-op on every rank and
 * nasr_get_loss return NASR_ERR_HIP ("step void"), so the replicas never diverge.
 *
 * Overlapping the exchange with the backward pass (the reference's towers cannot: average_gradients waits for
 * every tower's full gradient list, tfnetwork.py:72-86): the array is cut into nasr_grad_bucket_count()
 * contiguous buckets in the order nasr_compute_grads completes them (top LSTM layer + W + b first, then one per
 * layer going down, the bottom layer + the fault word last; a one-layer net has one bucket).
 * nasr_grad_bucket(i) gives bucket i's [offset, offset + count) in floats from nasr_grad_device_ptr();
 * nasr_grad_bucket_wait(i, s) makes HIP stream s wait until the nasr_compute_grads call issued before it has
 * finished bucket i (no host sync).  All-reduce bucket i on s after that wait, for i = 0 .. count-1, then make the
 * handle's stream wait for s before nasr_apply_adam.  The buckets cover the whole array exactly once. */
int nasr_upload_batch(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                      const int32_t* label_len, int B, int T, int Lmax);
/* Same, but the context stacking of utils.py:8-21 (include_context) happens on the device: `centre` is the
 * un-stacked [B,T,numcep] slice (for features made by preprocess_mfcc.py: columns [numcontext*numcep,
 * (numcontext+1)*numcep) of the stacked array), pad_value[b] the value the stacked array holds in the
 * out-of-utterance context frames of utterance b (its element [b,0,0]).  Needs feature_size ==
 * (2*numcontext+1)*numcep.  Moves 1/(2*numcontext+1) of the bytes over PCIe. */
int nasr_upload_batch_context(nasr_handle h, const float* centre, const float* pad_value, int numcontext, int numcep,
                              const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B, int T,
                              int Lmax);
/* The input pipeline's half of the step (dataset.py:33-40 loads and train.py:23-26 times the NEXT batch inside the
 * step): nasr_stage_batch copies a batch into one of the handle's staging slots - host side
 * through pinned memory (hipHostMalloc), device side with hipMemcpyAsync on the handle's COPY stream - while the
 * compute stream is busy with the current step, and returns a ticket; nasr_commit_batch(ticket) makes that batch the
 * resident one (the compute stream waits for the slot's copy event; no host sync).  At most two batches
 * staged ahead (NASR_ERR_STATE beyond that); a synchronous upload always finds a slot.  nasr_stage_batch* may be called from another host thread than the rest
 * of the handle's calls (a loader thread); everything else stays one thread per handle.  The synchronous
 * nasr_upload_batch* = stage on the compute stream + commit. */
int nasr_stage_batch(nasr_handle h, const float* feats, const int32_t* seq_len, const int32_t* labels,
                     const int32_t* label_len, int B, int T, int Lmax, int* ticket);
int nasr_stage_batch_context(nasr_handle h, const float* centre, const float* pad_value, int numcontext, int numcep,
                             const int32_t* seq_len, const int32_t* labels, const int32_t* label_len, int B, int T,
                             int Lmax, int* ticket);
int nasr_commit_batch(nasr_handle h, int ticket);
int nasr_discard_batch(nasr_handle h, int ticket);   /* give a staged batch's slot back unused */
int nasr_compute_grads(nasr_handle h);          /* forward+CTC+backward on the resident batch (async) */
/* ---- in-library gradient exchange: average_gradients (tfnetwork.py:72-86) for hosts without torch.distributed ------
 * One RCCL rank per handle (one process per GPU, or one host thread per handle).  librccl.so is bound with dlopen when
 * the first of these calls is made.  Rank 0 calls nasr_comm_unique_id and hands the 128 bytes to the other ranks by
 * the host's own means (MPI_Bcast, a file, a socket); every rank then calls nasr_comm_init (it blocks until all have
 * joined).  Per step:  nasr_compute_grads(h); nasr_comm_allreduce_grads(h); nasr_apply_adam(h, 1.f / nranks);
 * nasr_comm_allreduce_grads enqueues one sum all-reduce per gradient bucket on the handle's communication stream, each
 * behind its bucket's completion event (so the upper layers' gradients cross xGMI under the backward pass of the layers
 * below), and makes the handle's stream wait for the last one: no host synchronisation.  The step's fault word travels
 * in the last bucket, so a void step (nasr_step_void) is void on every rank.  nasr_comm_mean averages a few host floats
 * over the ranks (the reduce_mean of loss / LER at tfnetwork.py:135-136); without a communicator it leaves them as
 * they are.  It runs on a communicator and a stream of its own (ncclCommSplit of the handle's communicator at
 * nasr_comm_init), so it neither waits for the gradient buckets of a step in flight nor for the compute stream; with a
 * librccl that has no ncclCommSplit it shares the gradient communicator and is executed behind the buckets issued
 * before it (one communicator runs its collectives in issue order). */
int nasr_comm_unique_id(void* id128);
int nasr_comm_init(nasr_handle h, const void* id128, int rank, int nranks);
int nasr_comm_size(nasr_handle h);
int nasr_comm_allreduce_grads(nasr_handle h);
int nasr_comm_mean(nasr_handle h, float* vals, int n);
int nasr_comm_destroy(nasr_handle h);
/* In persistent mode bucket i's event is held back over the NEXT persistent BPTT launch (the layer below's), so that
 * a collective released by nasr_grad_bucket_wait co-runs with that layer's GEMM phase rather than with a launch whose
 * hand-offs want every CU's memory queue to themselves (default on, NASR_BUCKET_DEFER=0 at create); 0 records every
 * bucket's event as soon as its gradients are complete. */
int nasr_set_bucket_defer(nasr_handle h, int defer);
void* nasr_grad_device_ptr(nasr_handle h);
int64_t nasr_grad_device_count(nasr_handle h);
int nasr_grad_bucket_count(nasr_handle h);
int nasr_grad_bucket(nasr_handle h, int i, int64_t* offset, int64_t* count);
int nasr_grad_bucket_wait(nasr_handle h, int i, void* hip_stream);
int nasr_apply_adam(nasr_handle h, float grad_scale);
/* Diagnostics - what ONE GPU can show of a collective that co-runs with the step (average_gradients moved under the
 * backward pass, tfnetwork.py:72-86): waits on `hip_stream` for bucket i like nasr_grad_bucket_wait, then launches there a
 * kernel shaped like a ring all-reduce step over that bucket - nblocks workgroups of 256 threads, each sweeping its slice
 * `passes` times with 16-byte loads and stores, the data unchanged.  tools/rccl_standin.py, tests/test_gpu_persist.py. */
int nasr_diag_bucket_traffic(nasr_handle h, int i, void* hip_stream, int nblocks, int passes);
/* The gradients of tfnetwork.py:120-128 are a set: nothing orders dW(l) before the backward pass of layer l-1.  With the
 * persistent recurrence, 500-wide layers and more than one layer, layer l's weight gradients run on a side stream beside the
 * persistent BPTT launch of layer l-1 (bitwise the gradients of the serial order).  On by default
 * (NASR_WGRAD_OVERLAP=0 at nasr_create turns it off); nasr_set_wgrad_overlap switches it for A/B measurements. */
int nasr_set_wgrad_overlap(nasr_handle h, int enabled);
int nasr_get_wgrad_overlap(nasr_handle h); /* g*grad_scale, TF Adam, step += 1 (async) */
/* copy the gradients out (TF order) / load externally reduced gradients (TF order) for nasr_apply_adam:
 * the single-process form of average_gradients (several towers time-sliced on one GPU). */
int nasr_get_grads(nasr_handle h, float* flat, int64_t n);
int nasr_set_grads(nasr_handle h, const float* flat, int64_t n);
int nasr_get_loss(nasr_handle h, float* loss_out);    /* synchronises; loss of last compute_grads */
/* synchronises; *void_out = 1 when the (all-reduced) fault word of the last step is set: nasr_apply_adam was a no-op on
 * every rank and the caller should run the step again (a rank whose persistent recurrence aborted has switched to the
 * per-step kernels by then).  Call it after nasr_apply_adam on every rank: all ranks get the same answer. */
int nasr_step_void(nasr_handle h, int* void_out);
/* The values Network.train returns (tfnetwork.py:183-190: loss, and the decode the LER is computed from) are known after
 * the forward pass and the CTC kernels; the backward pass, the gradient exchange and Adam need not be waited for.  With
 * nasr_set_step_decode(1) every nasr_compute_grads copies the loss, the fault word as it stands after the forward pass and
 * the greedy decode (ids [B][T'] row-major, lens [B]) to pinned host memory right behind the CTC kernels;
 * nasr_get_step_results waits for THAT copy only.  A host that returns from train() at this point enqueues the next
 * step while the device still runs the backward pass of this one: the device never waits for the host.  fault_out = 1:
 * this rank's forward recurrence aborted, the values are meaningless (then use nasr_step_void and repeat the step).
 * nasr_settle_step(h, previous, &v) waits for the END of the latest (previous = 0) or the one-before-latest (1) step
 * that reached nasr_apply_adam and says whether it was void (on every rank: the fault word is all-reduced with the
 * gradients); a void step's Adam launch was a no-op and is taken out of the step count.
 * A host that runs more than one step ahead names the step instead: nasr_step_token(h) = the sequence number of the
 * optimiser step nasr_apply_adam enqueued last (> 0; 0 = none yet), nasr_settle_token(h, token, &v) waits for the end of
 * exactly that step.  The library remembers the last 4 steps; an older token is NASR_ERR_STATE.  (What the reference
 * gets from sess.run returning, tfnetwork.py:188-190: the step is over and its update applied - here per step, without a
 * stream synchronisation.) */
int nasr_get_step_results(nasr_handle h, float* loss_out, int* fault_out, int32_t* ids_out, int32_t* lens_out);
int nasr_settle_step(nasr_handle h, int previous, int* void_out);
int64_t nasr_step_token(nasr_handle h);
int nasr_settle_token(nasr_handle h, int64_t token, int* void_out);
int nasr_resident_frames(nasr_handle h, int64_t* frames); /* sum(seq_len) of the resident batch */
/* Ragged batches.  DataSet.get_next_batch (dataset.py:75-77) pads every utterance to the longest of its batch and
 * tf.nn.(bidirectional_)dynamic_rnn (networks/bilstm_ctc_net.py:40-53) masks by sequence_length.  Here, when at least
 * 15 % of a training batch's T x B frame rows are such padding, the operand passes and GEMMs of a plain (Bi)LSTM stack work
 * on the frames t < seq_len[b] only (gathered on the way in, scattered on the way out); the recurrences still run T steps.
 * Results are the uncompacted ones up to summation order.  On by default (NASR_COMPACT=0 in the environment: off);
 * a change takes effect with the next batch uploaded or committed.  nasr_resident_rows: the rows those passes cover for
 * the resident batch - sum(seq_len) when compacted, T x (B rounded up to 16) otherwise. */
int nasr_set_row_compaction(nasr_handle h, int enabled);
int nasr_resident_rows(nasr_handle h, int64_t* rows);

/* TensorFlowNetwork.train fetches mean_ler with every step (networks/tfnetwork.py:188-189): with
 * step-decode enabled nasr_compute_grads / nasr_loss also run the greedy decoder on the step's logits
 * (before the CTC gradient overwrites them); nasr_get_decoded returns that result (synchronises). */
int nasr_set_step_decode(nasr_handle h, int enabled);
/* enabled = 3: the step's logits are copied out as well (pinned memory, behind the CTC forward kernels, before the CTC
 * gradient overwrites them): nasr_get_step_logits waits for that copy only and returns them time-major [T',B,C], as
 * nasr_forward does.  The host can then run the reference's own decoder (nasr_ctc_beam_search) for the step's mean_ler
 * while the device runs the backward pass and the steps behind it (tfnetwork.py:61-70,188-189).  enabled = 2: the same
 * without the greedy decoder (a host with its own decoder has no use for it): nasr_get_step_results then returns loss and
 * fault word with empty hypotheses. */
int nasr_get_step_logits(nasr_handle h, float* logits_out);
int nasr_get_decoded(nasr_handle h, int32_t* ids_out /*[B,T']*/, int32_t* lens_out /*[B]*/);

/* create_model (networks/tfnetwork.py:61-64): tf.nn.ctc_beam_search_decoder on host logits, time-major
 * [T',B,C] (as nasr_forward returns them); TF defaults are beam_width 100, merge_repeated 1, top path only.
 * ids_out [B,T'] (row b holds lens_out[b] ids), logp_out [B] = log-probability of the best beam (may be NULL).
 * Host code (one thread per utterance), no GPU work. */
int nasr_ctc_beam_search(const float* logits, const int32_t* seq_len, int B, int Tp, int C, int beam_width,
                         int merge_repeated, int32_t* ids_out, int32_t* lens_out, float* logp_out);

/* create_metric (networks/tfnetwork.py:66-70): mean over the batch of Levenshtein(hyp, truth)/len(truth)
 * (tf.edit_distance normalize=True, Appendix A.7).  Host code, no GPU work.  hyp_ids [B,hyp_stride],
 * labels [B,Lmax].  An empty truth gives inf for a non-empty hypothesis and 0 otherwise, as TF does. */
int nasr_label_error_rate(const int32_t* hyp_ids, const int32_t* hyp_lens, int hyp_stride, const int32_t* labels,
                          const int32_t* label_len, int Lmax, int B, float* ler_out);

/* tf.nn.dropout of the dense stages (networks/deepspeech.py:50,59,68,113; applied in every graph, training or not).
 * TensorFlow's random stream is not reproducible, so the keep-mask of forward pass number `counter` is a pure function
 * of (seed, counter, stage, frame, utterance, unit): see neuralasr_amd/csrc/dense.hip.  Every forward pass uses the
 * current counter and then increments it; nasr_set_dropout_state pins both (tests, resuming). */
int nasr_set_dropout_state(nasr_handle h, uint32_t seed, uint32_t counter);
int nasr_get_dropout_state(nasr_handle h, uint32_t* seed, uint32_t* counter);

/* ---- measurement ------------------------------------------------------------------------ */
int nasr_set_profiling(nasr_handle h, int enabled); /* record HIP events around the phases */
int nasr_get_phase_times(nasr_handle h, nasr_phase_times* out); /* synchronises */
int nasr_set_graph_mode(nasr_handle h, int enabled); /* capture the per-timestep loops in hipGraphs */
/* How the recurrence of tf.nn.(bidirectional_)dynamic_rnn (networks/bilstm_ctc_net.py:24-28,
 * networks/lstm_ctc_net.py:22-23) runs: 1 = one persistent launch per layer pass (one XCD per direction and
 * utterance slice, recurrent matrix resident in registers), 2 = a layer of 2048 cells (networks/deepspeech.py:70-103):
 * one persistent launch per DIRECTION and pass (the matrix resident in the registers of all 256 CUs as fp16 planes),
 * 0 = one launch per timestep.  nasr_set_recurrence_mode(0)
 * forces the per-step kernels; (1) asks for the persistent ones again and returns NASR_ERR_STATE where the device or
 * the hidden size does not support them.  A persistent launch wants every CU of the device for itself: if another
 * process or handle keeps CUs busy for longer than its bounded spins (~0.5 s), the launch gives up, the step is void
 * (nasr_step_void) and this handle continues on the per-step kernels. */
int nasr_get_recurrence_mode(nasr_handle h);
int nasr_set_recurrence_mode(nasr_handle h, int persistent);
/* After an abort the handle serves NASR_PERSIST_REARM (default 200; 0 = never) clean steps on the per-step kernels,
 * then repeats the placement census of nasr_create at the start of a step and returns to the persistent kernels if it
 * passes; every further abort doubles the wait.  aborts / rearms: how often each has happened on this handle (a rank
 * that sits on the per-step kernels slows every rank of a data-parallel job: bench.py reports these per rank). */
int nasr_get_persist_stats(nasr_handle h, int* aborts, int* rearms);

#ifdef __cplusplus
}
#endif
#endif /* NASR_H */
