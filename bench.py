#!/usr/bin/env python3
"""bench.py — audio-frames/sec of the CTC training step (forward + CTC + backward + gradient
all-reduce + Adam) on N MI355X, one process per GPU.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): bilstm_ctc_net 3x500 bidirectional (concat merge),
16 kHz / 26 MFCC / numcontext 10 => F = 546, C = 29, batch 16 per GPU, T = 500 frames, synthetic features
resident in HBM, random-init weights.  `--workload literal` runs the reference's literal 1x500 BiLstmCTCNet
(stack-reshape merge) instead; both are reported in DESIGN.md.  A "step" is one pass of the hot path over one
resident batch; decode/LER are not on the timed path.  Weak scaling: every rank owns its own
shard of 16 utterances; the only exchange is one RCCL all-reduce (sum) of the flat fp32 gradient buffer."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


class Workload:
    """Shape bookkeeping of one network (the fields nasr_model_cfg takes) - bench.py builds its inputs itself; only
    the cpu_baseline leg touches oracle/."""

    def __init__(self, feature_size, hidden, num_layers, bidirectional, merge, num_classes, pre=(), post=0,
                 relu_clip=20.0, dropout=()):
        self.feature_size, self.hidden, self.num_layers = feature_size, hidden, num_layers
        self.bidirectional, self.merge, self.num_classes = bidirectional, merge, num_classes
        self.pre, self.post, self.relu_clip, self.dropout = tuple(pre), post, relu_clip, tuple(dropout)

    @property
    def dirs(self):
        return 2 if self.bidirectional else 1

    @property
    def deepspeech(self):
        return bool(self.pre) or self.post > 0

    @property
    def proj_in(self):
        return 2 * self.hidden if (self.bidirectional and self.merge == 'concat') else self.hidden

    def layer_input(self, l):
        if l == 0:
            return self.pre[-1] if self.pre else self.feature_size
        return self.hidden * self.dirs

    def param_count(self):
        n, w_in = 0, self.feature_size
        for w in self.pre:
            n += w + w_in * w
            w_in = w
        for l in range(self.num_layers):
            n += self.dirs * ((self.layer_input(l) + self.hidden) * 4 * self.hidden + 4 * self.hidden)
        last = self.proj_in
        if self.post:
            n += self.post + last * self.post
            last = self.post
        return n + last * self.num_classes + self.num_classes

    def oracle_spec(self):
        from oracle.nasr_oracle import ModelSpec      # cpu_baseline leg only
        return ModelSpec(self.feature_size, self.hidden, self.num_layers, self.bidirectional, self.merge,
                         self.num_classes, pre=self.pre, post=self.post, relu_clip=self.relu_clip, dropout=self.dropout)


def workload_spec(name):
    if name == 'literal':
        return Workload(546, 500, 1, True, 'stack_reshape', 29), 'bilstm_ctc_net literal 1x500 bi stack_reshape'
    if name == 'lstm3':
        return Workload(546, 500, 3, False, 'none', 29), 'lstm_ctc_net 3x500 uni'
    if name == 'deepspeech':     # BASELINE.json configs[3]: networks/deepspeech.py at its own sizes, batch 32 per GPU
        return (Workload(546, 2048, 1, True, 'concat', 29, pre=(2048, 2048, 4096), post=2048, relu_clip=20.0,
                         dropout=(0.05, 0.05, 0.05, 0.05)), 'deepspeech (3 dense + BiLSTM 2048 + dense), dropout 0.05')
    return Workload(546, 500, 3, True, 'concat', 29), 'bilstm_ctc_net 3x500 bi concat'


def synth_batch(spec, B, T, seed=1234, var_len=False):
    """BASELINE.md §4 synthetic batch: features N(0,1) zeroed past seq_len, labels U{1..C-2}, L ~ U{40..80} (scaled
    with T), seq_len = T or U{T/2..T} sorted ascending (the size-sorted scp of preprocess_mfcc.py:33)."""
    rs = np.random.RandomState(seed)
    if var_len:
        seq_len = np.sort(rs.randint(max(T // 2, 1), T + 1, size=B)).astype(np.int32)
        seq_len[-1] = T
    else:
        seq_len = np.full(B, T, np.int32)
    feats = rs.randn(B, T, spec.feature_size).astype(np.float32)
    for b in range(B):
        feats[b, seq_len[b]:] = 0
    lmax = max(1, min(80, T * 80 // 500))
    label_len = rs.randint(max(1, lmax // 2), lmax + 1, size=B).astype(np.int32)
    label_len = np.minimum(label_len, np.maximum(seq_len // 2, 1)).astype(np.int32)
    labels = np.zeros((B, int(label_len.max())), np.int32)
    for b in range(B):
        labels[b, :label_len[b]] = rs.randint(1, max(spec.num_classes - 1, 2), size=label_len[b])
    return feats, seq_len, labels, label_len


def init_params(tensors, seed=1):
    """Random-init weights of the architecture in TF variable order (Engine.tensors()): glorot-uniform for every matrix,
    zero biases - the defaults the reference's graph relies on."""
    rs = np.random.RandomState(seed)
    chunks = []
    for _, _, rows, cols in tensors:
        if cols == 1:
            chunks.append(np.zeros(rows, np.float32))
        else:
            lim = np.sqrt(6.0 / (rows + cols))
            chunks.append(rs.uniform(-lim, lim, size=rows * cols).astype(np.float32))
    return np.concatenate(chunks)


def algorithmic_bytes(spec, B, T):
    """A (activations) + W (params + Adam) + R (recurrent weights re-streamed every
    timestep, forward U and backward U^T).  Returns (A, W, R) in bytes per training step."""
    N = B * T
    D, H, C = spec.dirs, spec.hidden, spec.num_classes
    A = 0
    for l in range(spec.num_layers):
        I = spec.layer_input(l)
        A += 4 * N * (I + 6 * D * H + 15 * D * H + I + (I if l > 0 else 0))
    rows = 2 if (spec.bidirectional and spec.merge == 'stack_reshape') else 1
    A += 16 * N * C * rows
    # dense stages: forward reads the input and writes the output; backward reads dY, Y, writes dZ, re-reads input and dZ
    # for the weight gradient and writes the input gradient
    widths = [(spec.feature_size if i == 0 else spec.pre[i - 1], w) for i, w in enumerate(spec.pre)]
    if spec.post:
        widths.append((spec.proj_in, spec.post))
    for i_w, o_w in widths:
        A += 4 * N * (i_w + o_w + 3 * o_w + i_w + o_w + i_w)
    W = 40 * spec.param_count()
    R = 2 * T * sum(D * 4 * H * H * 4 for _ in range(spec.num_layers))
    return A, W, R


def algorithmic_flops(spec, B, T):
    """SURVEY.md §8d "ALGORITHMIC FLOPs per frame": forward = input + recurrent contractions of every layer, the dense
    stages and the projection; backward = 2x forward minus the input-gradient GEMM of whatever reads the features.
    Returns (dense FLOPs per step: everything hoisted out of the time loop, recurrent FLOPs per step)."""
    N = B * T
    D, H, C = spec.dirs, spec.hidden, spec.num_classes
    rows = 2 if (spec.bidirectional and spec.merge == 'stack_reshape') else 1
    hoisted, rec, first = 0, 0, None
    widths = [(spec.feature_size if i == 0 else spec.pre[i - 1], w) for i, w in enumerate(spec.pre)]
    for i_w, o_w in widths:
        hoisted += 2 * i_w * o_w
        first = first if first is not None else 2 * i_w * o_w
    for l in range(spec.num_layers):
        I = spec.layer_input(l)
        hoisted += 2 * I * 4 * H * D
        first = first if first is not None else 2 * I * 4 * H * D
        rec += 2 * H * 4 * H * D
    if spec.post:
        hoisted += 2 * spec.proj_in * spec.post
    hoisted += 2 * (spec.post or spec.proj_in) * C * rows
    # backward: weight gradient + input gradient for every contraction (2x), no input gradient for the first one; the
    # recurrent weight gradient (one more recurrent-sized product) is a hoisted GEMM over all frames
    dense = N * (hoisted + 2 * hoisted - first + rec)
    recurrent = N * (rec + rec)
    return dense, recurrent


def step_kernel_bytes(spec, B):
    """Algorithmic bytes of ONE launch of the per-timestep recurrence kernel (one timestep, both
    directions): the recurrent matrix U (or U^T) once + the per-frame activations it touches
    (forward: 4 gate pre-activations in, 4 activations out, c, c_prev, h out, h state in/out;
    BPTT: dG state in/out, 4 activations in, 4 dG out, c, c_prev, dOut, dc in/out)."""
    D, H = spec.dirs, spec.hidden
    u = D * 4 * H * H * 4
    fwd = u + B * D * H * 4 * (4 + 4 + 1 + 1 + 1 + 2)
    bwd = u + B * D * H * 4 * (8 + 4 + 4 + 1 + 1 + 1 + 2)
    return fwd, bwd


def cpu_baseline(spec, B, seed):
    """The CPU restatement timed on the GPU box's host cores (kind "port": TensorFlow itself is not in this image,
    SURVEY.md D8): oracle/cref/nasr_cref.c (plain C + OpenMP, fp32, the same forward + CTC + backward, Adam
    excluded) on a bounded sample of the same workload — same net and batch size, T sized by a short probe so the
    timed call takes ~10-20 s.  Falls back to the fp64 NumPy oracle if the C library cannot be built."""
    from oracle import nasr_oracle as O
    spec = spec.oracle_spec()
    try:
        from oracle import cref
        cref.set_threads(cref.usable_cpus())           # honour the container's CPU quota
        threads = cref.num_threads()
        params = O.flatten(O.init_params(spec, seed=1)).astype(np.float32)

        def run(T):
            feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=seed)
            t0 = time.time()
            cref.loss_and_grads(spec, params, feats, seq_len, labels, label_len,
                                drop=(4567, 0) if spec.deepspeech else None)
            return time.time() - t0, int(seq_len.sum())
        tp, tmin = (6, 8) if spec.deepspeech else (24, 32)   # 0.6 GFLOP per frame at DeepSpeech's sizes: probe small
        run(tp // 3)                                   # warm the thread pool / page in
        dt, fr = run(tp)
        Tc = int(max(tmin, min(500, tp * 14.0 / max(dt, 1e-3))))
        dt, fr = run(Tc)
        reps = 1
        while dt < 10.0 and reps < 8:                  # a fast host: repeat the step until ~10 s are on the clock
            d2, f2 = run(Tc)
            dt, fr, reps = dt + d2, fr + f2, reps + 1
        return {'value': fr / dt, 'unit': 'frames/s', 'cores': int(threads), 'kind': 'port',
                'sample': f'oracle/cref/nasr_cref.c (C + OpenMP, fp32), {reps} fwd+CTC+bwd step(s) of the same net at '
                          f'B={B}, T={Tc} ({fr} frames, {dt:.1f} s); Adam excluded'}
    except Exception as exc:                           # noqa: BLE001 - baseline must not take the bench down
        note = f'C restatement unavailable ({type(exc).__name__}); '
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        thr = os.cpu_count() or 1
    try:                                               # BLAS may size its pool by the host; the cgroup quota is what runs
        from oracle.cref import usable_cpus
        thr = min(thr, usable_cpus())
    except Exception:
        pass
    Tc = 24 if spec.deepspeech else (200 if spec.num_layers > 1 else 400)
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, Tc, seed=seed)
    params = O.init_params(spec, seed=1)
    t0 = time.time()
    O.network_loss_and_grads(spec, params, feats, seq_len, labels, label_len)
    dt = time.time() - t0
    return {'value': float(seq_len.sum() / dt), 'unit': 'frames/s', 'cores': int(thr), 'kind': 'port',
            'sample': note + f'fp64 NumPy oracle, 1 fwd+bwd step of the same net at B={B}, T={Tc} '
                             f'({int(seq_len.sum())} frames, {dt:.1f} s); Adam excluded'}


def kernel_roofline(spec, B, T, phases, mode):
    """The `roofline` object of a workload in compact form (the secondary entries of the line): the dominant recurrence
    kernel's algorithmic bytes per launch (SURVEY.md §8d per-timestep figure x the timesteps one launch runs) over its live
    launch duration against the HBM peak - or, for the wide kernels that keep a 2048-cell matrix resident, its three-product
    fp16 MFMA work against the dense fp16 matrix peak (as the main line does for --workload deepspeech)."""
    fb, bb = step_kernel_bytes(spec, B)
    persistent, wide = mode == 'persistent', mode == 'wide-persistent'
    spl = T if (persistent or wide) else 1
    dpl = spec.dirs if wide else 1
    fwd_us = phases['rec_fwd_ms'] * 1e3 / max(phases['rec_fwd_launches'], 1)
    bwd_us = phases['rec_bwd_ms'] * 1e3 / max(phases['rec_bwd_launches'], 1)
    dom_bwd = phases['rec_bwd_ms'] >= phases['rec_fwd_ms']
    k_bytes, k_us = (bb * spl / dpl, bwd_us) if dom_bwd else (fb * spl / dpl, fwd_us)
    fam = 'lstm_persist' if persistent else 'lstm_wide' if wide else 'lstm'
    kname = f"{fam}_{'bwd' if dom_bwd else 'fwd'}{'' if (persistent or wide) else '_step'}_kernel"
    achieved = k_bytes / (k_us * 1e-6) / 1e9
    rl = {'bound': 'hbm', 'kernel': kname, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
          'frac': achieved / HBM_PEAK_GBS, 'traffic': None, 'bytes_per_launch': k_bytes, 'us_per_launch': k_us,
          'timesteps_per_launch': spl, 'fwd_step_us': fwd_us / spl, 'bwd_step_us': bwd_us / spl}
    if wide:
        bp = (B + 15) // 16 * 16
        tf = 2.0 * bp * 2048 * 8192 * 3 * T / 1e12
        rl.update({'bound': 'mfma', 'hbm_algorithmic_frac': rl['frac'], 'achieved': tf / (k_us * 1e-6), 'peak': 2500.0,
                   'unit': 'TFLOP/s', 'frac': tf / (k_us * 1e-6) / 2500.0, 'mfma_products_per_fp32_product': 3})
    return rl


def measure_secondary(name, stream, device, steps=5, warmup=2):
    """One more workload under the same clock, in the same process (N = 1): `steps` timed optimisation steps of the named
    workload on a resident synthetic batch, bracketed by synchronisations like the main measurement."""
    import torch
    from neuralasr_amd.engine import Engine
    spec, wname = workload_spec(name)
    B, T = (32 if name == 'deepspeech' else 16), 500
    eng = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                 learning_rate=1e-4, device_id=device, stream=stream, pre=spec.pre, post=spec.post,
                 relu_clip=spec.relu_clip, dropout=spec.dropout)
    try:
        eng.set_params(init_params(eng.tensors(), seed=1))
        feats, seq_len, labels, label_len = synth_batch(spec, B, T, seed=1234)
        eng.upload_batch(feats, seq_len, labels, label_len)
        frames = eng.resident_frames()

        def step():
            eng.compute_grads()
            eng.apply_adam(1.0)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loss = eng.get_loss()
        eng.set_profiling(True)
        step()
        phases = eng.phase_times()
        eng.set_profiling(False)
        aborts, _ = eng.persist_stats()
        return {'config': {'workload': f'{wname}, F={spec.feature_size}, C={spec.num_classes}, batch {B} per GPU, T={T} frames'},
                'steps': steps, 'warmup': warmup, 'ms_per_step': dt / steps * 1e3, 'value': frames * steps / dt,
                'unit': 'frames/s', 'dtype': 'f32', 'loss': loss, 'recurrence': eng.recurrence_mode, 'persist_aborts': aborts,
                'roofline': kernel_roofline(spec, B, T, phases, eng.recurrence_mode),
                'phases_ms': {k: round(v, 4) for k, v in phases.items() if k.endswith('_ms')}}
    finally:
        eng.close()


PROFILE_FILES = {'pmc': 'pmc_traffic.json', 'stamps': 'persist_stamps.json'}


def load_profile(name):
    path = os.path.join(ROOT, 'profiles', PROFILE_FILES[name])
    try:
        return json.load(open(path)), os.path.join('profiles', PROFILE_FILES[name])
    except Exception:      # noqa: BLE001 - a missing profile only blanks the fields that come from it
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='bilstm3x500', choices=['bilstm3x500', 'literal', 'lstm3', 'deepspeech'])
    ap.add_argument('--batch', type=int, default=None, help='utterances per GPU (16; 32 for deepspeech)')
    ap.add_argument('--frames', type=int, default=500)
    ap.add_argument('--var-len', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary workloads of the default run (literal, deepspeech)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--per-step', action='store_true', help='per-timestep launches (lstm.hip) instead of the persistent recurrence')
    ap.add_argument('--allreduce', default='auto', choices=['auto', 'bucketed', 'bucketed-eager', 'single', 'lib'],
                    help='N > 1: gradient exchange as one all-reduce after the backward pass (single), or per-layer buckets '
                         'on a side stream under the rest of the backward pass, each released after the next persistent '
                         'BPTT launch (bucketed) or as soon as it is complete (bucketed-eager); auto times these three during '
                         'warm-up; lib = the same buckets through the library\'s own RCCL communicator (nasr_comm_*), opt-in')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from neuralasr_amd.engine import Engine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    if os.environ.get('NASR_BENCH_BACKEND', 'nccl') != 'nccl':
        local = 0                                   # rehearsal: every rank on the one GPU
    torch.cuda.set_device(local)
    use_dist = world > 1 or 'RANK' in os.environ        # under torch.distributed.run even at N = 1
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # NASR_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path with several ranks on ONE GPU (RCCL refuses that);
        # never a measurement
        backend = os.environ.get('NASR_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
    on_device = not use_dist or dist.get_backend() == 'nccl'

    spec, wname = workload_spec(args.workload)
    B, T = args.batch or (32 if args.workload == 'deepspeech' else 16), args.frames
    # One explicit (non-default) torch stream carries everything: the engine launches on it and torch.distributed
    # orders its RCCL work against the CURRENT stream, so kernels -> all-reduce -> Adam need no host sync.
    # (The legacy default stream cannot be used: it is not capturable and an engine-owned stream would not be
    # ordered with the collective.)
    tstream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    eng = Engine(spec.feature_size, spec.hidden, spec.num_layers, spec.bidirectional, spec.merge, spec.num_classes,
                 learning_rate=1e-4, device_id=local, stream=stream, pre=spec.pre, post=spec.post,
                 relu_clip=spec.relu_clip, dropout=spec.dropout)
    eng.set_graph_mode(not args.no_graph)
    if args.per_step:
        eng.set_recurrence_mode(False)
    eng.set_params(init_params(eng.tensors(), seed=1))   # same weights on every rank
    feats, seq_len, labels, label_len = synth_batch(spec, B, T, seed=1234 + rank, var_len=args.var_len)
    eng.upload_batch(feats, seq_len, labels, label_len)
    frames = eng.resident_frames()
    gt = eng.grad_tensor() if use_dist else None

    reducer = None
    if use_dist and world > 1 and args.allreduce == 'lib':
        # the library's own communicator: rank 0's ncclUniqueId goes round by torch.distributed (a C host would use
        # MPI_Bcast or a file), the collectives themselves are issued inside libnasr
        box = [eng.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(box[0], rank, world)
    elif use_dist and world > 1 and args.allreduce != 'single':
        try:                                  # the same code on every rank: a failure here is a failure everywhere
            from neuralasr_amd.parallel import BucketedAllReduce
            reducer = BucketedAllReduce(eng, dist, gt)
        except Exception as exc:              # noqa: BLE001 - fall back to the one-collective exchange
            print(f'bucketed all-reduce unavailable ({type(exc).__name__}: {exc}); using one all-reduce', file=sys.stderr)
            reducer = None
    can_bucket = reducer is not None and len(reducer.views) > 1
    ar_mode = (args.allreduce if args.allreduce != 'auto' else 'bucketed') if can_bucket else 'single'
    if args.allreduce == 'lib' and use_dist and world > 1:
        ar_mode = 'lib'

    def set_mode(mode):
        eng.set_bucket_defer(mode != 'bucketed-eager')

    def step():
        eng.compute_grads()
        if use_dist:
            if ar_mode == 'lib':
                eng.comm_allreduce_grads()
            elif ar_mode != 'single':
                reducer.all_reduce()
            else:
                dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        eng.apply_adam(1.0 / world)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_scalar(x, op):
        if not use_dist:
            return float(x)
        t = torch.tensor([float(x)], dtype=torch.float64, device='cuda' if on_device else 'cpu')
        dist.all_reduce(t, op=op)
        return float(t.item())

    ar_probe = None
    if use_dist and world > 1 and args.allreduce == 'auto' and can_bucket:
        # measure, don't guess: a few untimed steps each way (part of the warm-up), slowest rank decides for everybody
        probe = {}
        for mode in ('single', 'bucketed', 'bucketed-eager'):
            ar_mode = mode
            set_mode(mode)
            step(); step()
            fence()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            fence()
            probe[mode] = reduce_scalar((time.perf_counter() - t0) / 4, dist.ReduceOp.MAX) * 1e3
        ar_mode = min(probe, key=probe.get)
        ar_probe = probe
    set_mode(ar_mode)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_own = time.perf_counter() - t0
    dt = reduce_scalar(dt_own, dist.ReduceOp.MAX) if use_dist else dt_own
    total_frames = reduce_scalar(frames, dist.ReduceOp.SUM) if use_dist else float(frames)
    # also surfaces a void step / an aborted persistent launch on this rank: reported in the line, never fatal here
    loss, rank_error = float('nan'), None
    try:
        loss = eng.get_loss()
    except Exception as exc:              # noqa: BLE001
        rank_error = str(exc)[:200]
    # what every rank ran: a rank that fell back to the per-step kernels (or saw void steps) slows the whole job and
    # would be invisible in rank 0's line otherwise
    aborts, rearms = eng.persist_stats()
    mine = {'rank': rank, 'ms_per_step': dt_own / args.steps * 1e3, 'recurrence': eng.recurrence_mode,
            'persist_aborts': aborts, 'persist_rearms': rearms, 'frames': int(frames), 'loss': loss, 'error': rank_error}
    ranks = [mine]
    if use_dist:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    # ---- the same step fed from host buffers (features cross PCIe every step): never `value`, reported beside it
    NH = 5
    fence()
    t1 = time.perf_counter()
    for _ in range(NH):
        eng.upload_batch(feats, seq_len, labels, label_len)
        step()
    fence()
    dt_h2d = (time.perf_counter() - t1) / NH
    # ---- ... and with the NEXT batch staged through pinned memory on the copy stream while the step runs
    # (nasr_stage_batch / nasr_commit_batch: what train_model's loader thread does)
    def staged_loop(stage):
        t = stage()
        eng.commit_batch(t)
        t = stage()
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(NH):
            eng.commit_batch(t)
            step()
            t = stage()
        fence()
        d = (time.perf_counter() - t1) / NH
        eng.discard_batch(t)
        return d
    dt_h2d_staged = staged_loop(lambda: eng.stage_batch(feats, seq_len, labels, label_len))
    # ---- and fed the way HipNetwork.train feeds it: features with the include_context structure of preprocess_mfcc.py
    # (utils.py:8-21) go over PCIe as their centre [B,T,numcep] slice and are re-stacked on the device
    dt_ctx = dt_ctx_staged = None
    if spec.feature_size == 21 * 26:
        from neuralasr_amd.utils import include_context
        rs = np.random.RandomState(77 + rank)
        fctx = np.zeros((B, T, spec.feature_size), np.float32)
        for b in range(B):
            fctx[b, :seq_len[b]] = include_context(rs.randn(int(seq_len[b]), 26).astype(np.float32), 10, 26)
        if eng.upload_batch_context(fctx, seq_len, labels, label_len, 10, 26):
            step()
            fence()
            t1 = time.perf_counter()
            for _ in range(NH):
                eng.upload_batch_context(fctx, seq_len, labels, label_len, 10, 26)
                step()
            fence()
            dt_ctx = (time.perf_counter() - t1) / NH
            dt_ctx_staged = staged_loop(lambda: eng.stage_batch(fctx, seq_len, labels, label_len, 10, 26))
    eng.upload_batch(feats, seq_len, labels, label_len)      # back to the bench batch for the phase timings
    # ---- the gradient all-reduce alone (time per step and bus bandwidth), N > 1 only
    ar_ms = None
    if use_dist and world > 1:
        fence()
        t2 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        fence()
        ar_ms = reduce_scalar((time.perf_counter() - t2) / 10, dist.ReduceOp.MAX) * 1e3
        gt.zero_()            # the summed buffer is not a gradient any more; the next compute_grads overwrites it

    # ---- per-phase / per-launch timing with HIP events on the engine's stream (a few extra steps)
    eng.set_profiling(True)
    acc = None
    NP = 3
    for _ in range(NP):
        step()
        pt = eng.phase_times()
        acc = pt if acc is None else {k: (acc[k] + v if k.endswith('_ms') else v) for k, v in pt.items()}
    eng.set_profiling(False)
    phases = {k: (v / NP if k.endswith('_ms') else v) for k, v in acc.items()}
    # With the weight gradients of the upper layers co-running (the default where it applies), the persistent BPTT launches
    # share their CUs: a few more profiled steps WITHOUT the overlap give the dominant kernel's duration on its own.
    alone = None
    if getattr(eng, 'wgrad_overlap', False):
        eng.set_wgrad_overlap(False)
        step()
        eng.set_profiling(True)
        acc2 = None
        for _ in range(NP):
            step()
            pt = eng.phase_times()
            acc2 = pt if acc2 is None else {k: (acc2[k] + v if k.endswith('_ms') else v) for k, v in pt.items()}
        eng.set_profiling(False)
        eng.set_wgrad_overlap(True)
        alone = {k: (v / NP if k.endswith('_ms') else v) for k, v in acc2.items()}

    if rank == 0:
        ms = dt / args.steps * 1e3
        A, W, R = algorithmic_bytes(spec, B, T)
        fb, bb = step_kernel_bytes(spec, B)
        # persistent recurrence: ONE launch runs all T timesteps of a layer, so a launch's algorithmic bytes are
        # T x the per-timestep figure of SURVEY.md §8d (which prices the recurrent matrix once per timestep; the
        # persistent kernels keep it in registers, so the PMC traffic sits far below this figure)
        persistent = eng.recurrence_mode == 'persistent'
        wide = eng.recurrence_mode == 'wide-persistent'     # one launch per DIRECTION and pass (lstm_wide.hip)
        spl = T if (persistent or wide) else 1
        dpl = spec.dirs if wide else 1                      # launches that share one timestep's algorithmic bytes
        fwd_us = phases['rec_fwd_ms'] * 1e3 / max(phases['rec_fwd_launches'], 1)
        bwd_us = phases['rec_bwd_ms'] * 1e3 / max(phases['rec_bwd_launches'], 1)
        dom_bwd = phases['rec_bwd_ms'] >= phases['rec_fwd_ms']
        k_bytes, k_us = (bb * spl / dpl, bwd_us) if dom_bwd else (fb * spl / dpl, fwd_us)
        achieved = k_bytes / (k_us * 1e-6) / 1e9
        kname = ('lstm_persist_bwd_kernel' if dom_bwd else 'lstm_persist_fwd_kernel') if persistent else \
                ('lstm_wide_bwd_kernel' if dom_bwd else 'lstm_wide_fwd_kernel') if wide else \
                ('lstm_bwd_step_kernel' if dom_bwd else 'lstm_fwd_step_kernel')
        # PMC figures of the same command (separate --pmc passes, tools/pmc_summary.py); the files are named in the line
        pj, pmc_file = load_profile('pmc')
        wkey = args.workload + ('_varlen' if args.var_len else '')
        pmc = (pj or {}).get(wkey, {}).get(kname)
        mfma_busy = ((pj or {}).get('mfma', {}).get(wkey, {}).get(kname) or {}).get('mfma_util')
        sj, stamp_file = load_profile('stamps')
        # the chip-wide MFMA floor of one timestep: D*B*H*4H MACs on 1024 SIMDs; forward as 3 fp16 4x4x4 products
        # (8 cycles per 2048-FLOP MFMA), BPTT as fp32 4x4x1 (8 cycles per 512-FLOP MFMA), at the 2.4 GHz peak clock
        macs = spec.dirs * 16 * 512 * 2048 if spec.hidden <= 512 else None      # padded Hp = 512, 16 utterance rows
        floor_fwd = macs * 3 / 1024 / (1024 * 2.4e3) * 8 if macs else None        # us
        floor_bwd = macs / 256 / (1024 * 2.4e3) * 8 if macs else None
        if wide:
            # one DIRECTION's timestep: Bp x 2048 x 8192 MACs as 3 fp16 products on 16x16x32 MFMAs (8192 MACs, 16 cycles),
            # forward and BPTT alike
            bp = (B + 15) // 16 * 16
            floor_fwd = floor_bwd = bp * 2048 * 8192 * 3 / 8192 / 1024 * 16 / 2.4e3
        out = {
            'metric': f'audio-frames/sec (fwd+bwd+CTC+Adam) at batch {B} per GPU',
            'value': total_frames * args.steps / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{wname}, 16 kHz / 26 MFCC / numcontext 10 (F={spec.feature_size}), '
                                   f'C={spec.num_classes}, batch {B} per GPU, T={T} frames'
                                   + (', ragged lengths' if args.var_len else ''),
                       'batch_per_gpu': B, 'frames': T, 'var_len': bool(args.var_len),
                       'parallelism': f'dp{world}', 'hipgraph': not args.no_graph,
                       'recurrence': eng.recurrence_mode,
                       'wgrad_overlap': bool(getattr(eng, 'wgrad_overlap', False)),
                       'recurrence_forward_mfma': ('fp32 products from 2 fp16 planes of U and 2 fp16 parts of h (4x4x4 f16 MFMA)'
                                                   if os.environ.get('NASR_REC', 'f16') != 'f32' else 'fp32 4x4x1 MFMA'),
                       'gemm': 'fp32 products from 2 fp16 planes x 3 MFMA products, fp32 accumulation, power-of-two '
                               'row scales (gemm_tph.hip); projection on fp32 MFMA (gemm.hip)',
                       'allreduce': (ar_mode if world > 1 else None)},
            'loss': loss,
            'ranks': ranks,
            'roofline': {'bound': 'hbm', 'kernel': kname,
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc, 'bytes_per_launch': k_bytes, 'us_per_launch': k_us,
                         'timesteps_per_launch': spl, 'fwd_step_us': fwd_us / spl, 'bwd_step_us': bwd_us / spl,
                         # what the HBM actually moved for this kernel (PMC bytes / live launch time / peak): the kernel
                         # keeps the recurrent matrix in registers, so this is far below `frac` by design
                         'hbm_measured_frac': (pmc / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if pmc else None,
                         'mfma_busy': mfma_busy,
                         # what does bound it: the dependent chain of one timestep (hand-off + MFMA + cell update)
                         'latency': {'step_us': (bwd_us if dom_bwd else fwd_us) / spl,
                                     'mfma_floor_us': floor_bwd if dom_bwd else floor_fwd,
                                     'fwd_step_us': fwd_us / spl, 'fwd_mfma_floor_us': floor_fwd,
                                     'bwd_step_us': bwd_us / spl, 'bwd_mfma_floor_us': floor_bwd,
                                     'phase_cycles': (sj or {}).get(wkey if (wkey in (sj or {}) or wide) else 'bilstm3x500'),
                                     'note': ('bound = per-timestep dependent chain across the chip (partial sums / dG '
                                              'planes across the XCDs, h / partial dh through the XCD L2, MFMA phase, cell '
                                              'update), not HBM; step_us is per DIRECTION-step (one launch per direction)'
                                              if wide else
                                              'bound = per-timestep dependent chain inside one XCD (flag/payload hand-off '
                                              'through L2, MFMA phase, cell update), not HBM; phase_cycles = in-kernel '
                                              's_memtime stamps of wave 0 (tools/persistbench, NASR_PSTAMP build)')},
                         'sources': {'traffic_mfma_busy': pmc_file, 'phase_cycles': stamp_file,
                                     'us_per_launch': 'live HIP events on the engine stream (nasr_get_phase_times)'}},
            'roofline_step': {'bound': 'hbm', 'bytes_alg': A + W + R, 'bytes_compulsory': A + W,
                              'achieved': (A + W + R) / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': (A + W + R) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              'frac_compulsory': (A + W) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            'roofline_mfma': None,
            'phases_ms': {k: round(v, 4) for k, v in phases.items() if k.endswith('_ms')},
            'incl_h2d': {'ms_per_step': dt_h2d * 1e3, 'value': float(frames) * world / dt_h2d, 'unit': 'frames/s',
                         'note': 'features uploaded from host memory every step (PCIe-inclusive); rank-0 clock',
                         'staged': {'ms_per_step': dt_h2d_staged * 1e3, 'value': float(frames) * world / dt_h2d_staged,
                                    'note': 'the next batch staged through pinned memory on the copy stream while the '
                                            'step runs (nasr_stage_batch + nasr_commit_batch)'}},
        }
        if alone is not None:
            # `roofline` prices the dominant kernel as it runs in the timed steps - sharing its CUs with the side-stream weight-
            # gradient GEMMs (the step is faster for it, the launch slower).  The same kernel on its own:
            a_bwd_us = alone['rec_bwd_ms'] * 1e3 / max(alone['rec_bwd_launches'], 1)
            a_fwd_us = alone['rec_fwd_ms'] * 1e3 / max(alone['rec_fwd_launches'], 1)
            a_us, a_bytes = (a_bwd_us, bb * spl / dpl) if dom_bwd else (a_fwd_us, fb * spl / dpl)
            out['roofline']['alone'] = {'us_per_launch': a_us, 'achieved': a_bytes / (a_us * 1e-6) / 1e9,
                                        'frac': a_bytes / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 'step_us': a_us / spl,
                                        'phases_ms': {k: round(v, 4) for k, v in alone.items() if k.endswith('_ms')},
                                        'note': 'the same launches with nasr_set_wgrad_overlap(0): nothing shares their CUs '
                                                '(3 profiled steps after the timed ones); the timed steps run WITH the overlap'}
        if wide:
            # The wide kernels keep the recurrent matrix resident for the whole launch, so SURVEY §8d's per-timestep bytes
            # (which re-price it every step) exceed what any memory system could deliver in the launch's time: quoted as
            # `hbm_algorithmic_frac`, not as the bound.  What the launch is priced against is its MFMA work: one direction's
            # T timesteps of Bp x Hp x 4Hp MACs as three fp16 products, against the dense fp16 matrix peak.
            rl = out['roofline']
            bp = (B + 15) // 16 * 16
            tf = 2.0 * bp * 2048 * 8192 * 3 * T / 1e12
            rl.update({'bound': 'mfma', 'hbm_algorithmic_frac': rl['frac'], 'achieved': tf / (k_us * 1e-6), 'peak': 2500.0,
                       'unit': 'TFLOP/s', 'frac': tf / (k_us * 1e-6) / 2500.0, 'mfma_products_per_fp32_product': 3,
                       'note': 'one direction per launch; the dependent chain of a timestep (see latency) bounds it, the '
                               'MFMA floor is 1.28 us of it'})
        # (ii) of SURVEY.md §8d: the dense contractions (everything hoisted out of the time loop).  They run as THREE fp16
        # MFMA products per fp32 product, so the instruction stream is priced against the dense fp16 matrix peak with the
        # product count in; the fp32-equivalent rate is given beside it.  The phases include the plane / scale passes.
        gf_dense, gf_rec = algorithmic_flops(spec, B, T)
        t_dense = (phases['xproj_ms'] + phases['wgrad_ms'] + phases['proj_bwd_ms']) * 1e-3
        gemm_busy = ((pj or {}).get('mfma', {}).get(wkey, {}).get('gemm_tph_kernel') or {}).get('mfma_util')
        out['roofline_mfma'] = {'bound': 'mfma', 'flops_alg': gf_dense, 'mfma_products_per_fp32_product': 3,
                                'achieved': 3 * gf_dense / t_dense / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s',
                                'frac': 3 * gf_dense / t_dense / 1e12 / 2500.0,
                                'fp32_equivalent_tflops': gf_dense / t_dense / 1e12,
                                'mfma_busy_gemm_tph_kernel': gemm_busy,
                                'note': 'hoisted GEMMs (input projections, input / weight / recurrent-weight gradients, dense '
                                        'stages, projection backward) over the xproj + wgrad + proj_bwd phases; achieved = 3 x '
                                        'algorithmic FLOP (the fp16 MFMA products issued per fp32 product) / phase time; peak = '
                                        'dense fp16 MFMA (MI355X_MICROARCH.md); mfma_busy from the PMC pass',
                                'recurrent_flops': gf_rec,
                                'sources': {'mfma_busy': pmc_file}}
        if dt_ctx is not None:
            out['incl_h2d']['context_upload'] = {
                'ms_per_step': dt_ctx * 1e3, 'value': float(frames) * world / dt_ctx,
                'note': 'context-stacked features uploaded as their centre slice, stacking rebuilt on the device '
                        '(nasr_upload_batch_context)',
                'staged': {'ms_per_step': dt_ctx_staged * 1e3, 'value': float(frames) * world / dt_ctx_staged,
                           'vs_resident': dt_ctx_staged * 1e3 / ms,
                           'note': 'the same through nasr_stage_batch_context + nasr_commit_batch (what '
                                   'HipNetwork.train does under train_model)'}}
        if ar_ms is not None:
            gbytes = gt.numel() * 4 / 1e9
            out['allreduce'] = {'ms': ar_ms, 'bytes': gt.numel() * 4,
                                'bus_GBps': 2.0 * (world - 1) / world * gbytes / (ar_ms * 1e-3),
                                'xgmi_peak_GBps': 7 * 153.0, 'mode': ar_mode, 'buckets': len(eng.grad_buckets()),
                                'probe_ms_per_step': ar_probe,
                                'note': 'ms / bus_GBps: ONE all-reduce of the whole buffer, timed alone; mode = how the '
                                        'timed steps exchange gradients (bucketed: per-layer buckets on a side stream '
                                        'under the rest of the backward pass, each released after the next persistent '
                                        'BPTT launch; bucketed-eager: released at once)'}
        # BASELINE.json's other single-GPU shapes under the same clock (default run at N = 1 only): the literal 1x500
        # BiLstmCTCNet (the parity shape, configs[1]'s class as the reference ships it) and configs[3]'s per-GPU workload
        if (world == 1 and args.workload == 'bilstm3x500' and not args.var_len and not args.per_step and not args.no_secondary
                and args.batch is None and args.frames == 500):
            sec = {}
            for name in ('literal', 'deepspeech'):
                try:
                    sec[name] = measure_secondary(name, stream, local)
                except Exception as exc:      # noqa: BLE001 - a secondary workload never takes the line down
                    sec[name] = {'error': f'{type(exc).__name__}: {exc}'[:300]}
            out['secondary'] = sec
        # What TensorFlowNetwork.train also runs in every step (tfnetwork.py:61-70,188-189): the width-100 beam search + LER on the
        # step's logits.  Host code here (nasr_ctc_beam_search, one thread per utterance) and NOT part of `value`: HipNetwork.train
        # reports the greedy LER by default because this costs many steps' worth of time on the GPU box's cores.
        if world == 1 and not args.no_secondary:
            try:
                lg = eng.forward(feats, seq_len)
                t_b = time.perf_counter()
                eng.beam_search(lg, seq_len, 100)
                out['beam_ms_per_step'] = {'value': (time.perf_counter() - t_b) * 1e3, 'beam_width': 100, 'logit_frames': int(lg.shape[0]),
                                           'utterances': B, 'host_threads': min(B, os.cpu_count() or 1),
                                           'vs_step': (time.perf_counter() - t_b) * 1e3 / ms,
                                           'note': 'tf.nn.ctc_beam_search_decoder defaults on the logits of the timed weights (random init + '
                                                   'the timed Adam steps: near-flat posteriors, the decoder\'s slow case); see '
                                                   'tools/beamtime.py for trained-looking posteriors'}
                eng.upload_batch(feats, seq_len, labels, label_len)
            except Exception as exc:      # noqa: BLE001
                out['beam_ms_per_step'] = {'error': f'{type(exc).__name__}: {exc}'[:200]}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(spec, B, 1234)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
