#!/usr/bin/env python3
"""bench.py -- images/sec of the full G+D training step on synthetic FFHQ-shaped batches.

One process per GPU.  ``python bench.py`` runs N=1; for N>1 the driver launches it with
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (RCCL).

A "step" is one iteration of the reference loop (generator_trainer.py:351-353):
discriminator_update + generator_update with d_every=1, g_reg_every=4, d_reg_every=16,
path_batch_shrink=2 (ffhq.json:74-81), vanilla semantics, lazy regularisers included at their
cadence (the iteration counter keeps running through warm-up so i % 4 / i % 16 line up).

Workload (BASELINE.json): FFHQ 1024x1024, 4 images per GPU -- config[2]'s "batch=32 DDP on 8 GPUs"
at N=8, the same per-GPU work at N=1/2/4 (weak scaling).  ``--size 512 --batch-per-gpu 16`` gives
config[1].  Everything inside the timed region is real: G and D forward/backward through the HIP
kernels, R1 / path-length double-backward, Adam steps, EMA, gradient all-reduce.

Rank 0 prints ONE JSON line with the contract fields plus
  "roofline":      the dominant kernel's achieved TFLOP/s (algorithmic flops / HIP-event time on the
                   launch stream, measured inside the timed region) against the fp32 MFMA peak,
  "cpu_baseline":  the oracle (CPU restatement of the reference FUSED=False path) timed on the
                   host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (the split-bf16 path issues 3 bf16 MFMAs per product)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=32)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--batch-per-gpu', type=int, default=4)
    ap.add_argument('--precision', default=os.environ.get('GANCONTROL_CONV_PRECISION', 'bf16x3'), choices=['f32', 'bf16x3', 'bf16'],
                    help='conv arithmetic: bf16x3 = split-bf16 MFMA (fp32 storage, ~5e-6 relative error per layer), f32 = exact fp32 MFMA')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fp32-leg', action='store_true', help='skip the second timed region in exact fp32 arithmetic ("fp32_exact" in the JSON line)')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--timer', default='roofline', choices=['roofline', 'all'], help='which launches get HIP-event brackets in the timed region')
    ap.add_argument('--cpu-baseline-size', type=int, default=None, help='resolution of the CPU sample (default: --size)')
    return ap.parse_args()


def cpu_baseline(size):
    """Time the oracle (CPU restatement of the reference FUSED=False path, kind = "port") on the host cores with the SAME step
    definition as the GPU line: each of the four phases of an iteration is run once at batch 1 -- D step, R1 step, G step,
    path-length step (its batch is max(1, 1 // path_batch_shrink) = 1) -- and the lazy regularisers enter at their cadence:
    seconds per iteration = t_D + t_G + t_R1 / 16 + t_PL / 4."""
    from gan_control_amd.models.gan_model import Generator, Discriminator
    from oracle.step import OracleStep
    import warnings
    warnings.filterwarnings('ignore')
    host_cores = os.cpu_count() or 1
    avail = host_cores
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(avail, 32)       # MKL-DNN grouped convs stop scaling (and thrash) far below a 256-thread host
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    d = Discriminator(size, channel_multiplier=2)
    o = OracleStep(g.state_dict(), d.state_dict(), size, 1)
    gen = torch.Generator().manual_seed(0)
    real = torch.rand(1, 3, size, size, generator=gen) * 2 - 1
    phases = {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        fn(*a)
        phases[name] = time.perf_counter() - t0

    timed('d_step', o.d_step, real, torch.randn(1, 512, generator=gen))
    timed('r1_step', o.d_reg, real)
    timed('g_step', o.g_step, torch.randn(1, 512, generator=gen))
    timed('pl_step', o.g_reg, torch.randn(1, 512, generator=gen))
    per_iter = phases['d_step'] + phases['g_step'] + phases['r1_step'] / 16 + phases['pl_step'] / 4
    plain = phases['d_step'] + phases['g_step']
    return {'value': 1.0 / per_iter, 'unit': 'images/sec', 'cores': cores, 'host_cores': host_cores, 'kind': 'port',
            'phase_seconds': {k: round(v, 2) for k, v in phases.items()},
            'value_without_regularisers': 1.0 / plain,
            'sample': f'one call of each phase (D step, R1, G step, path length) at {size}x{size}, batch 1, fp32, oracle/step.py OracleStep on '
                      f'{cores} of {host_cores} host threads, {sum(phases.values()):.1f} s; images/sec = 1 / (t_D + t_G + t_R1/16 + t_PL/4), '
                      f'the cadence of the GPU step (the lazy regularisers are {100 * (per_iter / plain - 1):.0f} % of the CPU iteration)'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)')
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.set_num_threads(4)          # N ranks share the host: do not let each spawn one intra-op thread per core
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('GANCONTROL_FORCE_DDP') == '1')
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from gan_control_amd import _lib
    _lib.load()                                   # fail loudly without the HIP library
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    from gan_control_amd.utils.profiling import KernelTimer

    if args.precision:
        _backend.get().conv_mode = args.precision
    precision = _backend.get().conv_mode
    cfg = default_config(args.size, args.batch_per_gpu * world)
    trainer = GeneratorTrainer(cfg, device=f'cuda:{local_rank}', seed=0)
    real = trainer.synthetic_batch()              # resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def roofline_of(timer_summary, elapsed_s):
        convs = {k: v for k, v in timer_summary.items() if k.startswith('conv_')}
        if not convs:
            return None
        name, dom = max(convs.items(), key=lambda kv: kv[1]['total_ms'])
        ach = dom['work'] / (dom['total_ms'] * 1e-3) / 1e12
        split = name.startswith('conv_bf16x3')
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        return {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                'launches': dom['launches'], 'avg_launch_us': dom['avg_us'], 'gpu_time_share': dom['total_ms'] / (1e3 * elapsed_s)}

    it = 0
    # Discovery during the (untimed) warm-up: every convolution launch is bracketed to find the variant that takes the most
    # GPU time; in the timed region only that variant (and the FIR tile kernel, for the HBM line) gets HIP events, so the
    # measurement costs < 1 % instead of ~3 % of the step.
    discover = None
    if rank == 0 and not args.no_kernel_timer and args.timer == 'roofline':
        discover = KernelTimer(only=('conv',))
        _backend.get().timer = discover
    for _ in range(args.warmup):
        trainer.train_iteration(it, real)
        it += 1
    timer = None
    if rank == 0 and not args.no_kernel_timer:
        if discover is not None:
            torch.cuda.synchronize()
            found = discover.summary()
            names = {'fir44_tile_kernel'}
            if found:
                names.add(max(found.items(), key=lambda kv: kv[1]['total_ms'])[0])
            timer = KernelTimer(only=('conv', 'fir44'), names=names)
        else:
            timer = KernelTimer(only=None)
        _backend.get().timer = timer
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_iteration(it, real)
        it += 1
    barrier()
    elapsed = time.perf_counter() - t0
    _backend.get().timer = None

    t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    stats = trainer.reduced_stats()

    # The same workload once more in exact fp32 arithmetic (the reference's precision): the headline `value` above is the
    # split-bf16 mode (fp32 in HBM, ~16 mantissa bits in the products), so the line carries the fp32 figure next to it.
    fp32_exact = None
    if precision != 'f32' and not args.no_fp32_leg:
        _backend.get().conv_mode = 'f32'
        disc32 = KernelTimer(only=('conv',)) if (rank == 0 and not args.no_kernel_timer) else None
        _backend.get().timer = disc32
        for _ in range(args.warmup):
            trainer.train_iteration(it, real)
            it += 1
        timer32 = None
        if disc32 is not None:
            torch.cuda.synchronize()
            found = disc32.summary()
            timer32 = KernelTimer(only=('conv',), names={max(found.items(), key=lambda kv: kv[1]['total_ms'])[0]} if found else set())
        _backend.get().timer = timer32
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_iteration(it, real)
            it += 1
        barrier()
        el32 = time.perf_counter() - t0
        _backend.get().timer = None
        _backend.get().conv_mode = precision
        t = torch.tensor([el32], dtype=torch.float64, device='cuda')
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el32 = float(t.item())
        fp32_exact = {'value': args.steps * args.batch_per_gpu * world / el32, 'unit': 'images/sec', 'ms_per_step': 1e3 * el32 / args.steps,
                      'steps': args.steps, 'warmup': args.warmup, 'dtype': 'f32'}
        if timer32 is not None:
            fp32_exact['roofline'] = roofline_of(timer32.summary(), el32)

    if rank == 0:
        images = args.steps * args.batch_per_gpu * world
        out = {
            'metric': 'images/sec G+D step FFHQ-%d' % args.size, 'value': images / elapsed, 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': precision, 'data': 'synthetic',
            'config': {'workload': 'FFHQ %dx%d full G+D train step (D step, G step, lazy R1 every 16 and path-length every 4, '
                                   'Adam, EMA), %d images/GPU, fp32 storage, conv arithmetic %s' % (args.size, args.size, args.batch_per_gpu, precision),
                       'global_batch': args.batch_per_gpu * world, 'parallelism': 'dp%d' % world},
            'losses': {k: round(v, 5) for k, v in stats.items()},
        }
        if timer is not None:
            summ = timer.summary()
            convs = {k: v for k, v in summ.items() if k.startswith('conv_')}
            kernels = {}
            for k, v in summ.items():
                unit_tf = k.startswith('conv_') or k.startswith('wgrad')
                rate = v['work'] / (v['total_ms'] * 1e-3) / (1e12 if unit_tf else 1e9)
                kernels[k] = {'launches': v['launches'], 'avg_us': round(v['avg_us'], 2), 'total_ms': round(v['total_ms'], 2),
                              'achieved': round(rate, 2), 'unit': 'TFLOP/s' if unit_tf else 'GB/s'}
            out['kernels'] = kernels
            if convs:
                out['roofline'] = roofline_of(summ, elapsed)
                name = out['roofline']['kernel']
                ach, peak, split = out['roofline']['achieved'], out['roofline']['peak'], name.startswith('conv_bf16x3')
                # separate rocprofv3 --pmc passes (tools/pmc_mix.py), launch-weighted over this kernel's shape mix; newest round first
                for traffic_file in sorted((f for f in os.listdir(os.path.join(REPO, 'profiles')) if f.startswith('pmc_r') and f.endswith('_traffic.json')), reverse=True):
                    pmc = json.load(open(os.path.join(REPO, 'profiles', traffic_file)))
                    if pmc.get('kernel') == name:
                        out['roofline']['traffic'] = pmc['traffic_bytes_per_launch']
                        out['roofline']['traffic_unit'] = 'bytes/launch'
                        out['roofline']['traffic_source'] = pmc['source'] + ' (profiles/%s)' % traffic_file
                        break
                if split:
                    out['roofline']['note'] = ('achieved = ALGORITHMIC flops/s; the split-bf16 kernel issues 3 bf16 MFMAs per product, '
                                               'so the matrix pipes do 3x this (frac of bf16 peak spent = %.3f) and frac <= 1/3 by construction; '
                                               'per-shape PMC traffic: profiles/pmc_r*_traffic.json, profiles/pmc_r01.md' % (3 * ach / peak))
            fir = summ.get('fir44_tile_kernel')
            if fir:
                ach = fir['work'] / (fir['total_ms'] * 1e-3) / 1e9
                out['roofline_hbm'] = {'bound': 'hbm', 'kernel': 'fir44_tile_kernel', 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                       'frac': ach / PEAK_HBM_GBS, 'traffic': None, 'launches': fir['launches'], 'avg_launch_us': fir['avg_us']}
        if fp32_exact is not None:
            out['fp32_exact'] = fp32_exact
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.cpu_baseline_size or args.size)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
