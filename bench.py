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


def workload_spec(name):
    from oracle.nasr_oracle import ModelSpec      # shape bookkeeping + synthetic batch only
    if name == 'literal':
        return ModelSpec(546, 500, 1, True, 'stack_reshape', 29), 'bilstm_ctc_net literal 1x500 bi stack_reshape'
    if name == 'lstm3':
        return ModelSpec(546, 500, 3, False, 'none', 29), 'lstm_ctc_net 3x500 uni'
    if name == 'deepspeech':     # BASELINE.json configs[3]: networks/deepspeech.py at its own sizes, batch 32 per GPU
        return (ModelSpec(546, 2048, 1, True, 'concat', 29, pre=(2048, 2048, 4096), post=2048, relu_clip=20.0,
                          dropout=(0.05, 0.05, 0.05, 0.05)), 'deepspeech (3 dense + BiLSTM 2048 + dense), dropout 0.05')
    return ModelSpec(546, 500, 3, True, 'concat', 29), 'bilstm_ctc_net 3x500 bi concat'


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
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--per-step', action='store_true', help='per-timestep launches (lstm.hip) instead of the persistent recurrence')
    ap.add_argument('--allreduce', default='auto', choices=['auto', 'bucketed', 'single'],
                    help='N > 1: gradient exchange as one all-reduce after the backward pass, or per-layer buckets on a '
                         'side stream under the rest of the backward pass; auto times both during warm-up')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from neuralasr_amd.engine import Engine
    from oracle import nasr_oracle as O

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
    eng.set_params(O.flatten(O.init_params(spec, seed=1)).astype(np.float32))   # same weights on every rank
    feats, seq_len, labels, label_len = O.synth_batch(spec, B, T, seed=1234 + rank, var_len=args.var_len)
    eng.upload_batch(feats, seq_len, labels, label_len)
    frames = eng.resident_frames()
    gt = eng.grad_tensor() if use_dist else None

    reducer = None
    if use_dist and world > 1 and args.allreduce != 'single':
        try:                                  # the same code on every rank: a failure here is a failure everywhere
            from neuralasr_amd.parallel import BucketedAllReduce
            reducer = BucketedAllReduce(eng, dist, gt)
        except Exception as exc:              # noqa: BLE001 - fall back to the one-collective exchange
            print(f'bucketed all-reduce unavailable ({type(exc).__name__}: {exc}); using one all-reduce', file=sys.stderr)
            reducer = None
    ar_mode = 'bucketed' if (reducer is not None and len(reducer.views) > 1) else 'single'

    def step():
        eng.compute_grads()
        if use_dist:
            if ar_mode == 'bucketed':
                reducer.all_reduce()
            else:
                dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        eng.apply_adam(1.0 / world)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ar_probe = None
    if use_dist and world > 1 and args.allreduce == 'auto' and ar_mode == 'bucketed':
        # measure, don't guess: a few untimed steps each way (part of the warm-up), slowest rank decides for everybody
        probe = {}
        for mode in ('single', 'bucketed'):
            ar_mode = mode
            step(); step()
            fence()
            t0 = time.perf_counter()
            for _ in range(4):
                step()
            fence()
            tt = torch.tensor([(time.perf_counter() - t0) / 4], device='cuda', dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            probe[mode] = float(tt.item()) * 1e3
        ar_mode = 'bucketed' if probe['bucketed'] <= probe['single'] else 'single'
        ar_probe = probe
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ft = torch.tensor([frames], device='cuda', dtype=torch.float64)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        total_frames = float(ft.item())
    else:
        total_frames = float(frames)
    loss = eng.get_loss()

    # ---- the same step fed from host buffers (features cross PCIe every step): never `value`, reported beside it
    NH = 5
    fence()
    t1 = time.perf_counter()
    for _ in range(NH):
        eng.upload_batch(feats, seq_len, labels, label_len)
        step()
    fence()
    dt_h2d = (time.perf_counter() - t1) / NH
    # ---- and fed the way HipNetwork.train feeds it: features with the include_context structure of preprocess_mfcc.py
    # (utils.py:8-21) go over PCIe as their centre [B,T,numcep] slice and are re-stacked on the device
    dt_ctx = None
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
            eng.upload_batch(feats, seq_len, labels, label_len)      # back to the bench batch for the phase timings
    # ---- the gradient all-reduce alone (time per step and bus bandwidth), N > 1 only
    ar_ms = None
    if use_dist and world > 1:
        fence()
        t2 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        fence()
        ar = torch.tensor([(time.perf_counter() - t2) / 10], device='cuda', dtype=torch.float64)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_ms = float(ar.item()) * 1e3
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

    if rank == 0:
        ms = dt / args.steps * 1e3
        A, W, R = algorithmic_bytes(spec, B, T)
        fb, bb = step_kernel_bytes(spec, B)
        # persistent recurrence: ONE launch runs all T timesteps of a layer, so a launch's algorithmic bytes are
        # T x the per-timestep figure of SURVEY.md §8d (which prices the recurrent matrix once per timestep; the
        # persistent kernels keep it in registers, so the PMC traffic sits far below this figure)
        persistent = eng.recurrence_mode == 'persistent'
        spl = T if persistent else 1
        fwd_us = phases['rec_fwd_ms'] * 1e3 / max(phases['rec_fwd_launches'], 1)
        bwd_us = phases['rec_bwd_ms'] * 1e3 / max(phases['rec_bwd_launches'], 1)
        dom_bwd = phases['rec_bwd_ms'] >= phases['rec_fwd_ms']
        k_bytes, k_us = (bb * spl, bwd_us) if dom_bwd else (fb * spl, fwd_us)
        achieved = k_bytes / (k_us * 1e-6) / 1e9
        kname = ('lstm_persist_bwd_kernel' if dom_bwd else 'lstm_persist_fwd_kernel') if persistent else \
                ('lstm_bwd_step_kernel' if dom_bwd else 'lstm_fwd_step_kernel')
        pmc = None
        pmc_path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(pmc_path):
            try:
                pj = json.load(open(pmc_path))
                pmc = pj.get(args.workload, {}).get(kname)
            except Exception:
                pmc = None
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
                       'recurrence_forward_mfma': ('fp32 products from 2 fp16 planes of U and 2 fp16 parts of h (4x4x4 f16 MFMA)'
                                                   if os.environ.get('NASR_REC', 'f16') != 'f32' else 'fp32 4x4x1 MFMA'),
                       'gemm': {'tp': 'fp32 products from 2 fp16 planes x 3 MFMA products, fp32 accumulation, power-of-two '
                                      'row scales (gemm_tph.hip)',
                                'tp3': 'fp32 products from 3 bf16 planes x 6 MFMA products, fp32 accumulation (gemm_tp.hip)',
                                'bf16': 'as tp3, operands split inside the GEMM (gemm_bf16.hip)',
                                'f32': 'fp32 MFMA (gemm.hip)'}.get(os.environ.get('NASR_GEMM', 'tp'), 'tp'),
                       'allreduce': (ar_mode if world > 1 else None)},
            'loss': loss,
            'roofline': {'bound': 'hbm', 'kernel': kname,
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc, 'bytes_per_launch': k_bytes, 'us_per_launch': k_us,
                         'timesteps_per_launch': spl, 'fwd_step_us': fwd_us / spl, 'bwd_step_us': bwd_us / spl},
            'roofline_step': {'bound': 'hbm', 'bytes_alg': A + W + R, 'bytes_compulsory': A + W,
                              'achieved': (A + W + R) / (ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': (A + W + R) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            'roofline_mfma': None,
            'phases_ms': {k: round(v, 4) for k, v in phases.items() if k.endswith('_ms')},
            'incl_h2d': {'ms_per_step': dt_h2d * 1e3, 'value': float(frames) * world / dt_h2d, 'unit': 'frames/s',
                         'note': 'features uploaded from host memory every step (PCIe-inclusive); rank-0 clock'},
        }
        # (ii) of SURVEY.md §8d: the dense contractions (everything hoisted out of the time loop) against the fp32 matrix
        # peak the reference's arithmetic type would be held to; the phases include the plane / scale passes of the GEMMs
        gf_dense, gf_rec = algorithmic_flops(spec, B, T)
        t_dense = (phases['xproj_ms'] + phases['wgrad_ms'] + phases['proj_bwd_ms']) * 1e-3
        out['roofline_mfma'] = {'bound': 'mfma', 'flops_alg': gf_dense, 'achieved': gf_dense / t_dense / 1e12,
                                'peak': 157.3, 'unit': 'TFLOP/s', 'frac': gf_dense / t_dense / 1e12 / 157.3,
                                'note': 'hoisted GEMMs (input projections, input / weight / recurrent-weight gradients, dense '
                                        'stages, projection backward) over the xproj + wgrad + proj_bwd phases; peak = fp32 '
                                        'MFMA (v_mfma_f32_32x32x2_f32); the GEMMs run fp32-accurate products on the 16-bit '
                                        'matrix cores, which is how frac can exceed 1',
                                'recurrent_flops': gf_rec,
                                'recurrent_step_us': {'fwd': fwd_us / spl, 'bwd': bwd_us / spl,
                                                      'dependent_launch_floor_us': 1.55}}
        if dt_ctx is not None:
            out['incl_h2d']['context_upload'] = {
                'ms_per_step': dt_ctx * 1e3, 'value': float(frames) * world / dt_ctx,
                'note': 'context-stacked features uploaded as their centre slice, stacking rebuilt on the device '
                        '(nasr_upload_batch_context, what HipNetwork.train does)'}
        if ar_ms is not None:
            gbytes = gt.numel() * 4 / 1e9
            out['allreduce'] = {'ms': ar_ms, 'bytes': gt.numel() * 4,
                                'bus_GBps': 2.0 * (world - 1) / world * gbytes / (ar_ms * 1e-3),
                                'xgmi_peak_GBps': 7 * 153.0, 'mode': ar_mode, 'buckets': len(eng.grad_buckets()),
                                'probe_ms_per_step': ar_probe,
                                'note': 'ms / bus_GBps: ONE all-reduce of the whole buffer, timed alone; mode = how the '
                                        'timed steps exchange gradients (bucketed: per-layer buckets on a side stream '
                                        'under the rest of the backward pass)'}
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
