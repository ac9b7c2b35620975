#!/usr/bin/env python3
"""bench.py -- images/sec of the full G+D training step on synthetic FFHQ-shaped batches.

One process per GPU.  ``python bench.py`` runs N=1.  For N>1 either the driver launches it with
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (RCCL), or -- when WORLD_SIZE is
not set -- ``python bench.py --gpus N`` launches exactly that command itself as a CHILD process before anything here
touches the GPU, relays the child's JSON line and exits with its code (the reference's train_generator.py:12-19 is one
command too).

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
                   host cores on a bounded sample of the same workload,
  "families":      per kernel family (stride-1 conv, transposed conv, stride-2 conv, weight gradients, FIR, activation /
                   plane reductions, RGB-side pointwise kernels, weight re-layouts, ATen) ms and launches per step with the
                   achieved rate against its roofline -- from one UNTIMED pass of 16 iterations under torch.profiler
                   after the timed region,
  "step_roofline": algorithmic convolution TFLOP per step (tallied launch by launch over that pass) / ms_per_step,
  "phases_fired":  how many lazy-regulariser passes fell inside the timed window,
  "comm":          (N > 1) gradient bytes handed to RCCL per step and the compute-stream stall they caused.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'gan-control_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (the split-bf16 path issues 3 bf16 MFMAs per product)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak


def parse(argv=None):
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
    ap.add_argument('--no-host-issue', action='store_true', help='skip the untimed host-issue-time pass ("host_issue_ms_per_step")')
    ap.add_argument('--no-families', action='store_true', help='skip the untimed profiler pass behind "families" / "step_roofline"')
    # Multi-GPU knobs: they only set the environment of the rank processes this command launches (self_launch), so the first session on an
    # 8-GPU node is a sweep of flags, not an edit.  Under the driver's own torch.distributed.run command set the variables directly.
    ap.add_argument('--rccl-algo', default=None, help='NCCL_ALGO for the ranks (Ring | Tree | ...; RCCL default when unset)')
    ap.add_argument('--rccl-proto', default=None, help='NCCL_PROTO for the ranks (Simple | LL | LL128)')
    ap.add_argument('--bucket-mb', type=float, default=None, help='gradient bucket size in MiB (GANCONTROL_BUCKET_MB, default 32)')
    ap.add_argument('--last-bucket-mb', type=float, default=None, help='size of the bucket of the LAST gradients of a backward (GANCONTROL_LAST_BUCKET_MB, default = bucket size)')
    ap.add_argument('--force-ddp', action='store_true',
                    help='with --gpus 1: run the one rank through torch.distributed.run with the RCCL gradient reducer on (GANCONTROL_FORCE_DDP=1): '
                         'the single-GPU cost of the data-parallel path (flat-buffer copies, hooks, one-rank collectives) against the plain run')
    return ap.parse_args(argv)


RANK_ENV_FLAGS = (('rccl_algo', 'NCCL_ALGO'), ('rccl_proto', 'NCCL_PROTO'), ('bucket_mb', 'GANCONTROL_BUCKET_MB'), ('last_bucket_mb', 'GANCONTROL_LAST_BUCKET_MB'))


def visible_gpu_count():
    """GPUs this process would see, WITHOUT touching the HIP runtime in this process (the launcher parent must stay free to start children, and on
    some ROCm builds torch.cuda.device_count() falls through to hipGetDeviceCount): KFD topology nodes with SIMDs, cut down by the usual visibility
    variables; if sysfs is not readable, a short-lived child process asks torch."""
    n = None
    try:
        root = '/sys/class/kfd/kfd/topology/nodes'
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get('simd_count', '0')) > 0
        # a container is usually handed a subset of the host's GPUs: only the render nodes it can open count
        import glob
        nodes = glob.glob('/dev/dri/renderD*')
        if nodes:
            n = min(n, sum(os.access(d, os.R_OK | os.W_OK) for d in nodes))
        if not os.access('/dev/kfd', os.R_OK | os.W_OK):
            n = 0
    except (OSError, ValueError):
        n = None
    if n is None:
        r = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'], capture_output=True, text=True)
        try:
            return int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            return 0
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(',') if t.strip() != '']))
    return n


def self_launch(args, argv, entry):
    """``bench.py --gpus N`` outside torch.distributed.run: start the N-rank job as a child and relay it.

    Runs before any GPU call or torch.cuda query of this process (re-exec'ing a process that has initialised the GPU takes
    the machine down on this pool; a child process is safe)."""
    if not _TEST_CPU['enabled']:
        have = visible_gpu_count()
        if have < args.gpus:
            print('bench.py: --gpus %d asked for, %d GPU(s) visible on this machine: not starting the %d-rank job (the ranks beyond the visible '
                  'devices would fail at set_device and leave the others waiting at the rendezvous)' % (args.gpus, have, args.gpus), file=sys.stderr)
            return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), entry] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault('NCCL_DEBUG', 'VERSION')                 # RCCL prints its version line once: evidence of which library ran
    for flag, var in RANK_ENV_FLAGS:
        if getattr(args, flag) is not None:
            env[var] = str(getattr(args, flag))
    if args.force_ddp:
        env['GANCONTROL_FORCE_DDP'] = '1'
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print('bench.py: the %d-rank child printed no JSON line' % args.gpus, file=sys.stderr)
        return 1
    return proc.returncode


# kernel name -> family.  Stride-1 convolutions include the input gradients of stride-1 layers (the same kernels); the input gradient of
# a transposed convolution is a stride-2 convolution and vice versa.  Two spellings arrive here: the profiler's demangled names
# (`conv_bf16x3_kernel<1, 4, 2, 1, 1, 2, 3>(...)`: template arguments ..., UP, DOWN, KS) and the backend's tally names
# (`conv_bf16x3_kernel<1,4,2,1>|up1,down2,k3`: geometry after the bar).
_GEOM_TAG = re.compile(r'\|up(\d+),down(\d+)')
_TEMPLATE = re.compile(r'(conv_bf16x3_kernel|conv_mfma_kernel)<([^>]*)>')
_SIMPLE = (
    ('rccl', re.compile(r'rccl|nccl', re.I)),
    ('wgrad', re.compile(r'wgrad_')),
    ('pointwise', re.compile(r'pw_(narrow|widen|wgrad)')),
    ('convt', re.compile(r'convt_')),
    ('conv_s2', re.compile(r'conv_s2ws_')),          # the wave-specialised stride-2 kernel (conv_s2ws.hip): its template arguments carry no geometry
    ('fir', re.compile(r'fir44|firK|generic_kernel|affine_warp|reflect_pad')),
    ('bias_act', re.compile(r'bias_act|plane_dot|channel_sum|rows_sum_div')),
    ('weights', re.compile(r'weight_layout|pack_weights')),
    ('style', re.compile(r'outer_kernel|style_')),
)


def family_of(name):
    for fam, rx in _SIMPLE:
        if rx.search(name):
            return fam
    if 'conv_' not in name and 'splitk_finish' not in name:
        return 'aten'
    up = down = 1
    m = _GEOM_TAG.search(name)
    if m:
        up, down = int(m.group(1)), int(m.group(2))
    else:
        m = _TEMPLATE.search(name)
        if m:
            t = [int(v) for v in m.group(2).replace(' ', '').split(',')]
            if len(t) >= 7:
                up, down = t[-3], t[-2]
        elif 'conv_f32_small_kernel<' in name:
            t = [int(v) for v in name.split('conv_f32_small_kernel<')[1].split('>')[0].replace(' ', '').split(',')]
            down = t[3] if len(t) >= 4 else 1
    return 'convt' if up > 1 else ('conv_s2' if down > 1 else 'conv_s1')


def family_table(trainer, it, real, backend, precision, steps, KernelTimer, world, use_dist):
    """One untimed pass of ``steps`` iterations (a whole cadence cycle of the lazy regularisers) under torch.profiler: device time
    and launches per kernel family from the profiler's kernel records, algorithmic work per family from the backend's launch
    tallies (flops for the convolutions, bytes for the FIR / activation kernels that report them)."""
    import torch
    from torch.profiler import profile, ProfilerActivity
    tally = KernelTimer(only=None)
    # Only the profiler's own start / stop / export may fail softly (another tracer attached, ...): an exception out of train_iteration -- a
    # kernel error, a collective timeout -- is a failed bench and propagates.  Rank 0 alone carries the profiler's overhead in this pass; the
    # other ranks wait for it in their collectives, which is why the pass is untimed and `comm` is tallied over the timed region only.
    prof = None
    try:
        prof = profile(activities=[ProfilerActivity.CUDA])
        prof.__enter__()
    except Exception as e:       # noqa: BLE001
        print('bench.py: profiler did not start (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
        prof = None
    backend.timer = tally
    # Algorithmic bytes of the memory-bound families the backend's own tallies do not cover (activation passes, plane reductions, the RGB-side
    # kernels) and of the ATen launches: every CUDA tensor a call reads counted once, every tensor it returns counted once.  For ATen that is an
    # ESTIMATE of what its kernels move (a dispatch-mode tally over the ops that launch kernels; views and allocations excluded).
    extra = {'bias_act': 0.0, 'pointwise': 0.0, 'aten': 0.0}
    HBM_METHODS = {'bias_act_bwd': 'bias_act', 'bias_act_bwd_reduce': 'bias_act', 'bias_act_bwd_reduce_adjoint': 'bias_act', 'plane_dot': 'bias_act',
                   'channel_sum': 'bias_act', 'rows_sum_div': 'bias_act', 'pw_act_wgrad': 'pointwise', 'pw_act_dgrad': 'pointwise'}

    def cuda_tensors(obj):
        if torch.is_tensor(obj):
            if obj.is_cuda:
                yield obj
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                yield from cuda_tensors(o)

    def nbytes(obj):
        return float(sum(t.numel() * t.element_size() for t in cuda_tensors(obj)))

    saved = {}
    for meth, famname in HBM_METHODS.items():
        fn = getattr(backend, meth, None)
        if fn is None:
            continue
        saved[meth] = fn

        def wrapped(*a, _fn=fn, _fam=famname, **kw):
            res = _fn(*a, **kw)
            extra[_fam] += nbytes(list(a) + list(kw.values())) + nbytes(res)
            return res
        setattr(backend, meth, wrapped)
    # the thin 1x1 convolutions (ToRGB / FromRGB on the vector ALUs) are HBM-bound: their forward / input-gradient launches in bytes
    from gan_control_amd.utils.profiling import conv_variant as _variant
    conv_fn = backend.conv2d
    saved['conv2d'] = conv_fn

    def conv_wrapped(x, w_t, in_scale, out_scale, geom, *a, **kw):
        res = conv_fn(x, w_t, in_scale, out_scale, geom, *a, **kw)
        if _variant(geom, w_t.shape[3], x.shape[0], x.shape[1], backend.conv_mode, in_hw=(x.shape[2], x.shape[3])).startswith('pw_'):
            extra['pointwise'] += nbytes([x, res]) + nbytes(list(a) + list(kw.values()))
        return res
    backend.conv2d = conv_wrapped
    from torch.utils._python_dispatch import TorchDispatchMode
    NO_KERNEL = ('view', 'reshape', 'empty', 'as_strided', 'expand', 'permute', 'transpose', 'detach', 'alias', 'slice', 'select', 'unsqueeze', 'squeeze',
                 'unbind', 'split', 'chunk', 't.', 'size', 'stride', 'is_', 'lift', 'new_empty', '_unsafe_view', 'diagonal', 'unfold', 'narrow', '_local_scalar_dense', 'set_', 'resize_')

    class AtenBytes(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            res = func(*args, **(kwargs or {}))
            name = func.__name__
            if not any(name.startswith(pfx) for pfx in NO_KERNEL):
                extra['aten'] += nbytes(list(args) + list((kwargs or {}).values())) + nbytes(res)
            return res
    try:
        with AtenBytes():
            for _ in range(steps):
                trainer.train_iteration(it, real)
                it += 1
        torch.cuda.synchronize()
    finally:
        backend.timer = None
        for meth, fn in saved.items():
            if meth in backend.__dict__:
                delattr(backend, meth)
    if prof is None:
        return None, None, it
    try:
        prof.__exit__(None, None, None)
        path = os.path.join(tempfile.mkdtemp(prefix='gc_bench_'), 'trace.json')
        prof.export_chrome_trace(path)
        events = json.load(open(path))['traceEvents']
        os.remove(path)
    except Exception as e:       # noqa: BLE001
        print('bench.py: family pass skipped (%s: %s)' % (type(e).__name__, e), file=sys.stderr)
        return None, None, it
    fam = {}
    for e in events:
        if e.get('cat') == 'kernel' and 'dur' in e:
            d = fam.setdefault(family_of(e['name']), {'us': 0.0, 'launches': 0})
            d['us'] += e['dur']
            d['launches'] += 1
    work = {}
    for name, _, _, w in tally.records:
        f = 'wgrad' if name.startswith('wgrad') else ('bias_act' if name.startswith('bias_act') else family_of(name))
        if f == 'pointwise':
            continue                 # tallied in flops by the backend; its rate is stated in bytes below (thin 1x1 convolutions: HBM-bound)
        work[f] = work.get(f, 0.0) + w
    flops_pw = sum(w for name, _, _, w in tally.records if family_of(name) == 'pointwise' and not name.startswith('wgrad'))
    for f, b in extra.items():
        work[f] = work.get(f, 0.0) + b
    mfma_peak = PEAK_FP32_MFMA_TFLOPS if precision == 'f32' else PEAK_BF16_MFMA_TFLOPS
    out = {}
    for f, d in sorted(fam.items(), key=lambda kv: -kv[1]['us']):
        row = {'ms_per_step': round(d['us'] / 1e3 / steps, 3), 'launches_per_step': round(d['launches'] / steps, 1), 'achieved': None, 'peak': None,
               'unit': None, 'frac': None}
        w = work.get(f)
        if w and d['us'] > 0:
            if f in ('conv_s1', 'conv_s2', 'convt', 'wgrad'):
                row.update(achieved=round(w / (d['us'] * 1e-6) / 1e12, 1), peak=mfma_peak, unit='TFLOP/s')
            elif f in ('fir', 'bias_act', 'pointwise', 'aten'):
                row.update(achieved=round(w / (d['us'] * 1e-6) / 1e9, 1), peak=PEAK_HBM_GBS, unit='GB/s')
                if f == 'aten':
                    row['note'] = 'bytes estimated from the tensors each ATen op reads and returns (dispatch-mode tally); many of these launches are latency-bound scalars'
            if row['achieved'] is not None:
                row['frac'] = round(row['achieved'] / row['peak'], 4)
        out[f] = row
    conv_flops = sum(work.get(f, 0.0) for f in ('conv_s1', 'conv_s2', 'convt', 'wgrad')) + flops_pw
    return out, conv_flops / steps, it


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
    # How many threads?  Measured, not asserted, AT THE SAMPLE'S OWN RESOLUTION: one discriminator forward of the oracle (the convolution-heavy
    # half of every phase) at `size`, batch 1, at each candidate count; the fastest runs the sample.  The sweep goes into the JSON line
    # ("thread_sweep").  Candidates stop at 64: profiles/cpu_threads_r03.json holds the full sweep of this host class (128 threads 2.9x, all
    # 256 threads 62x slower than the best -- oversubscribed MKL-DNN on per-sample convolutions), which would also cost minutes of bench time.
    from oracle import networks as onet
    sweep = {}
    torch.manual_seed(0)
    g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    d = Discriminator(size, channel_multiplier=2)
    g_sd, d_sd = g.state_dict(), d.state_dict()
    gen = torch.Generator().manual_seed(0)
    real = torch.rand(1, 3, size, size, generator=gen) * 2 - 1
    with torch.no_grad():
        for n in sorted({c for c in (8, 16, 32, 64) if 1 <= c <= avail} or {avail}):
            torch.set_num_threads(n)
            onet.discriminator_forward(d_sd, real)          # first call at this count: thread pool start-up, primitive caches
            t0 = time.perf_counter()
            onet.discriminator_forward(d_sd, real)
            sweep[n] = time.perf_counter() - t0
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)

    def run_phases(batch, which):
        o = OracleStep(g_sd, d_sd, size, batch)
        x = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
        out = {}
        for name in which:
            t0 = time.perf_counter()
            if name == 'd_step':
                o.d_step(x, torch.randn(batch, 512, generator=gen))
            elif name == 'r1_step':
                o.d_reg(x)
            elif name == 'g_step':
                o.g_step(torch.randn(batch, 512, generator=gen))
            else:
                o.g_reg(torch.randn(max(1, batch // 2), 512, generator=gen))
            out[name] = time.perf_counter() - t0
        return out

    phases = run_phases(1, ('d_step', 'r1_step', 'g_step', 'pl_step'))
    per_iter = phases['d_step'] + phases['g_step'] + phases['r1_step'] / 16 + phases['pl_step'] / 4
    plain = phases['d_step'] + phases['g_step']
    out = {'value': 1.0 / per_iter, 'unit': 'images/sec', 'cores': cores, 'host_cores': host_cores, 'kind': 'port',
           'thread_sweep': {'workload': 'oracle discriminator forward at %dx%d, batch 1, seconds' % (size, size), 'seconds': {str(k): round(v, 3) for k, v in sweep.items()}},
           'phase_seconds': {k: round(v, 2) for k, v in phases.items()},
           'value_without_regularisers': 1.0 / plain,
           'sample': f'one call of each phase (D step, R1, G step, path length) at {size}x{size}, batch 1, fp32, oracle/step.py OracleStep on '
                     f'{cores} of {host_cores} host threads (the fastest of the thread sweep in this line), {sum(phases.values()):.1f} s; images/sec = 1 / (t_D + t_G + t_R1/16 + t_PL/4), '
                     f'the cadence of the GPU step (the lazy regularisers are {100 * (per_iter / plain - 1):.0f} % of the CPU iteration)'}
    # The GPU step works on 4 images: the plain iteration (D step + G step) once more at batch 4, where the host's cores have four samples to
    # spread over -- skipped altogether when the batch-1 sample was slow, so that the default bench stays within a few minutes.
    if plain <= 45.0:
        four = run_phases(4, ('d_step', 'g_step'))
        t4 = four['d_step'] + four['g_step']
        out['batch4'] = {'value_without_regularisers': 4.0 / t4, 'phase_seconds': {k: round(v, 2) for k, v in four.items()},
                         'note': 'D step and G step at batch 4 on the same threads, both measured; with the batch-1 share of the lazy regularisers: '
                                 '%.4f images/sec' % (4.0 / (t4 * per_iter / plain))}
        out['value_batch4'] = 4.0 / (t4 * per_iter / plain)
    return out


# Tests only (tests/bench_emulated_entry.py): the launcher / rank plumbing of this file exercised on CPUs over gloo with the emulated
# C ABI installed by the test entry.  Never set by bench.py itself: without it the bench refuses to run without a GPU.
_TEST_CPU = {'enabled': False}


def _sync():
    if not _TEST_CPU['enabled']:
        torch.cuda.synchronize()


def main(argv=None, entry=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.force_ddp):
        # nothing above this line has touched the GPU (importing torch does not)
        raise SystemExit(self_launch(args, argv, entry or os.path.abspath(__file__)))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    cpu_test = _TEST_CPU['enabled']
    if not cpu_test:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
        if local_rank >= torch.cuda.device_count():
            raise SystemExit('bench.py: rank %d wants cuda:%d but %d GPU(s) are visible' % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    if world > 1:
        # N ranks share the host: do not let each spawn one intra-op thread per core (an explicit OMP_NUM_THREADS below 4 is respected)
        torch.set_num_threads(max(1, min(4, int(os.environ.get('OMP_NUM_THREADS', '4') or 4))))
    use_dist = world > 1 or ('RANK' in os.environ and os.environ.get('GANCONTROL_FORCE_DDP') == '1')
    rccl = None
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if cpu_test:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        rccl = {'ranks': dist.get_world_size(), 'backend': dist.get_backend()}
        if not cpu_test:
            try:
                rccl['version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())     # RCCL reports through the nccl interface
            except Exception as e:       # noqa: BLE001
                rccl['version'] = 'unavailable (%s)' % type(e).__name__

    from gan_control_amd import _lib
    _lib.load()                                   # fail loudly without the HIP library
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    from gan_control_amd.utils.profiling import KernelTimer

    if cpu_test:
        args.no_kernel_timer = args.no_families = args.no_fp32_leg = args.no_host_issue = True
    elif args.precision:
        _backend.get().conv_mode = args.precision
    precision = getattr(_backend.get(), 'conv_mode', 'emulated')
    cfg = default_config(args.size, args.batch_per_gpu * world)
    dev = 'cpu' if cpu_test else f'cuda:{local_rank}'
    trainer = GeneratorTrainer(cfg, device=dev, seed=0)
    real = trainer.synthetic_batch()              # resident in HBM before the timed region
    for red in (trainer.g_reducer, trainer.d_reducer):
        red.measure = use_dist

    def barrier():
        _sync()
        if use_dist:
            dist.barrier()
        _sync()

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
            _sync()
            found = discover.summary()
            names = {'fir44_tile_kernel'}
            if found:
                names.add(max(found.items(), key=lambda kv: kv[1]['total_ms'])[0])
            timer = KernelTimer(only=('conv', 'fir44'), names=names)
        else:
            timer = KernelTimer(only=None)
        _backend.get().timer = timer
    barrier()
    for red in (trainer.g_reducer, trainer.d_reducer):
        red.comm_summary(reset=True)              # drop what warm-up tallied
    first_timed = it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_iteration(it, real)
        it += 1
    barrier()
    elapsed = time.perf_counter() - t0
    _backend.get().timer = None
    tc = cfg['training_config']
    timed = range(first_timed, first_timed + args.steps)
    phases_fired = {'d': sum(1 for i in timed if i % tc['d_every'] == 0), 'g': args.steps,
                    'r1': sum(1 for i in timed if i % tc['d_reg_every'] == 0), 'pl': sum(1 for i in timed if i % tc['g_reg_every'] == 0),
                    'cadence': {'r1_every': tc['d_reg_every'], 'pl_every': tc['g_reg_every']}}
    comm = None
    if use_dist:
        parts = [red.comm_summary() for red in (trainer.g_reducer, trainer.d_reducer)]
        comm = {'bytes_per_step': sum(c['bytes'] for c in parts) / args.steps, 'exposed_ms_per_step': sum(c['exposed_ms'] for c in parts) / args.steps,
                'note': 'gradient payload handed to the all-reduce per rank and step; exposed = compute-stream stall at GradientReducer.finish() '
                        '(HIP events either side of the waits), i.e. communication that backward did not hide'}

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
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
        while it % 16 != first_timed % 16 or it < first_timed + args.steps + args.warmup:      # same phase alignment as the first timed region
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
        t = torch.tensor([el32], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el32 = float(t.item())
        fp32_exact = {'value': args.steps * args.batch_per_gpu * world / el32, 'unit': 'images/sec', 'ms_per_step': 1e3 * el32 / args.steps,
                      'steps': args.steps, 'warmup': args.warmup, 'dtype': 'f32'}
        if timer32 is not None:
            fp32_exact['roofline'] = roofline_of(timer32.summary(), el32)

    # Untimed: the whole picture per kernel family (one cadence cycle of 16 iterations under torch.profiler on rank 0; the other ranks
    # run the same iterations so the collectives line up).
    families = step_flops = None
    if not args.no_families and not args.no_kernel_timer:
        cycle = 16
        if rank == 0:
            families, step_flops, it = family_table(trainer, it, real, _backend.get(), precision, cycle, KernelTimer, world, use_dist)
        else:
            for _ in range(cycle):
                trainer.train_iteration(it, real)
                it += 1
        barrier()

    if families is None:
        step_flops = None
    # Untimed: HOST issue time per step -- one cadence cycle of 16 iterations, each issued into a drained device and timed on the host clock
    # up to the return of train_iteration (no synchronisation inside): what Python / ctypes / autograd cost per step when nothing blocks.
    # The step is device-bound as long as this stays below ms_per_step; it is the floor a faster kernel set would hit.
    host_issue = None
    if not args.no_host_issue:
        cycle, acc_ms = 16, 0.0
        worst_ms = 0.0
        for _ in range(cycle):
            _sync()
            t0 = time.perf_counter()
            trainer.train_iteration(it, real)
            dt = (time.perf_counter() - t0) * 1e3
            acc_ms += dt
            worst_ms = max(worst_ms, dt)
            it += 1
        barrier()
        host_issue = {'ms_per_step': acc_ms / cycle, 'max_ms': worst_ms, 'iterations': cycle,
                      'note': 'host wall time of train_iteration() issued into a drained device, mean over one cadence cycle (16 iterations: 1 R1 + 4 path-length passes)'}
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
            'phases_fired': phases_fired,
        }
        if host_issue is not None:
            out['host_issue_ms_per_step'] = round(host_issue['ms_per_step'], 3)
            out['host_issue'] = host_issue
        if rccl is not None:
            rccl['knobs'] = {var: os.environ.get(var) for _, var in RANK_ENV_FLAGS}
            rccl['buckets'] = {'generator': len(trainer.g_reducer.buckets), 'discriminator': len(trainer.d_reducer.buckets),
                               'bucket_bytes': trainer.g_reducer.bucket_bytes, 'last_bucket_bytes': trainer.g_reducer.last_bucket_bytes}
            out['rccl'] = rccl
        if comm is not None:
            out['comm'] = comm
        if families is not None:
            out['families'] = families
            peak = PEAK_FP32_MFMA_TFLOPS if precision == 'f32' else PEAK_BF16_MFMA_TFLOPS
            ach = step_flops / (elapsed / args.steps) / 1e12
            out['step_roofline'] = {'tflop_per_step': round(step_flops / 1e12, 3), 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                                    'note': 'algorithmic convolution + weight-gradient flops of one iteration (launch tallies over a 16-iteration cadence cycle, '
                                            'per GPU) / ms_per_step of the timed region; families: untimed profiler pass, ms are device time per step'}
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
                # Counters of this kernel from separate rocprofv3 --pmc passes: HBM-side traffic (tools/pmc_mix.py, launch-weighted over its shape mix) and
                # MFMA utilisation (tools/pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x launch cycles)).  The counters were collected on SOME build of the
                # kernels: only a file stamped with the hash of the sources THIS library was built from (gc_source_hash) may be quoted.  Among the
                # files of profiles/ the one with the matching hash is taken; if none matches, the newest is named as dropped.
                prof_dir, my_hash = os.path.join(REPO, 'profiles'), _lib.source_hash()

                def counter_files(suffix_test):
                    cands = []
                    for f in sorted((f for f in os.listdir(prof_dir) if f.startswith('pmc_r') and f.endswith('.json') and suffix_test(f)), reverse=True):
                        try:
                            cands.append((f, json.load(open(os.path.join(prof_dir, f)))))
                        except (OSError, ValueError):
                            pass
                    return cands
                traffic = [(f, d) for f, d in counter_files(lambda f: f.endswith('_traffic.json')) if d.get('kernel') == name]
                hit = next(((f, d) for f, d in traffic if d.get('source_hash') == my_hash), None)
                if hit is not None:
                    out['roofline']['traffic'] = hit[1]['traffic_bytes_per_launch']
                    out['roofline']['traffic_unit'] = 'bytes/launch'
                    out['roofline']['traffic_source'] = hit[1]['source'] + ' (profiles/%s)' % hit[0]
                elif traffic:
                    out['roofline']['traffic_source'] = ('dropped: profiles/%s was collected on other kernel sources (%s, now %s); re-run tools/pmc_mix.py'
                                                         % (traffic[0][0], traffic[0][1].get('source_hash'), my_hash))
                sub = name.split('|')[0].replace(',', ', ').rstrip('>')
                mfma = [(f, d, next((k for k in d.get('kernels', []) if sub in k.get('name_substring', '')), None)) for f, d in counter_files(lambda f: '_mfma' in f)]
                mfma = [(f, d, row) for f, d, row in mfma if row is not None]
                hit = next(((f, d, row) for f, d, row in mfma if d.get('source_hash') == my_hash), None)
                if hit is not None:
                    f, d, row = hit
                    out['roofline']['mfma_busy'] = round(row['mfma_busy'], 4)
                    out['roofline']['mfma_busy_source'] = ('share of the launch in which a SIMD\'s matrix pipe executes an MFMA, %s (profiles/%s; wave time parked %.0f %%, issue-stalled %.0f %%, '
                                                           '%.2f vector instructions per MFMA)' % (row['kernel'], f, 100 * (row.get('wait_any_frac') or 0), 100 * (row.get('wait_inst_any_frac') or 0), row.get('valu_per_mfma') or 0))
                elif mfma:
                    out['roofline']['mfma_busy_source'] = ('dropped: profiles/%s was collected on other kernel sources (%s, now %s); re-run tools/pmc_mfma.sh'
                                                           % (mfma[0][0], mfma[0][1].get('source_hash'), my_hash))
                if name.startswith('conv_bf16x3_ws_kernel'):
                    out['roofline']['rocprof_names'] = ('rocprofv3 lists this kernel once per epilogue variant -- %s, <KS, WOC, CB, 0 | 1 | 2, residual> (full / scale-or-residual / none): '
                                                        'compare avg_launch_us with their launch-weighted average' % name.split('|')[0].replace('>', ', EPK, RES>'))
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
