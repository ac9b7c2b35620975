"""world_size-2 (gloo, CPU) coverage of the data-parallel path.

1. GradientReducer: bucketed, hook-driven mean all-reduce equals the full-batch gradient.
2. Trainer: one full iteration (D step, R1, G step, path-length, EMA) on 2 ranks x 4 images equals
   the same iteration on 1 rank x 8 images (the reference's single-process semantics), with the
   per-rank shards chosen to keep the minibatch-stddev groups identical (members are strided).
"""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, EmulatedBackend

SIZE, GLOBAL_B = 16, 8


def _setup(rank, world, port):
    for p in (REPO, os.path.join(REPO, 'gan-control_amd'), os.path.join(REPO, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _reducer_worker(rank, world, port, out):
    _setup(rank, world, port)
    from gan_control_amd.trainers.ddp import GradientReducer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(10, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7), torch.nn.Linear(7, 1))
    unused = torch.nn.Parameter(torch.zeros(3))
    net.register_parameter('unused', unused)
    red = GradientReducer(net, bucket_bytes=4096)
    assert len(red.buckets) > 2
    x = torch.randn(8, 10, generator=torch.Generator().manual_seed(1))
    # two micro-batches with accumulation, reduce only on the last; the first optimiser step of a phase learns which
    # parameters the pass produces (reduced from finish()), the second launches every bucket from a gradient hook
    xs = x[rank::world]
    for it in range(2):
        net.zero_grad(set_to_none=True)
        red.begin(sync=False, phase='step'); net(xs[:2]).mean().mul(0.5).backward()
        red.begin(sync=True, phase='step'); net(xs[2:]).mean().mul(0.5).backward()
        if it == 1:
            assert all(b.work is not None for b in red.buckets if any(p is not unused for p in b.params)), 'a bucket did not launch from its hook'
        red.finish()
        assert red.report['step']['late'] == 0
        assert (red.report['step']['finish'] == 0) == (it == 1), red.report
        for b in red.buckets:                       # gradients live in the bucket buffer afterwards: no copy back
            for p in b.params:
                assert p.grad is None or p.grad.data_ptr() == b.view(p).data_ptr()
    if rank == 0:
        torch.save({n: p.grad for n, p in net.named_parameters()}, out)
    dist.destroy_process_group()


def test_gradient_reducer_matches_full_batch():
    port = 29500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'g.pt')
        mp.spawn(_reducer_worker, args=(2, port, out), nprocs=2, join=True)
        got = torch.load(out)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(10, 300), torch.nn.ReLU(), torch.nn.Linear(300, 7), torch.nn.Linear(7, 1))
    x = torch.randn(8, 10, generator=torch.Generator().manual_seed(1))
    # mean over ranks of (mean of first halves + mean of second halves)/2 == mean over all 8
    net(x).mean().backward()
    assert got.pop('unused') is None          # the unused parameter keeps grad None
    for n, p in net.named_parameters():
        assert torch.allclose(got[n], p.grad, atol=1e-6), n


def _changing_set_worker(rank, world, port, out):
    """A phase whose gradient set changes between iterations, a backward OUTSIDE begin() .. finish(), and a gap large enough to split a
    bucket's span: every reduced gradient must still equal the full-batch one and nothing stale may be reduced."""
    _setup(rank, world, port)
    from gan_control_amd.trainers.ddp import GradientReducer
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(6, 40), torch.nn.Linear(6, 40), torch.nn.Linear(6, 40)
    net = torch.nn.ModuleList([a, b, c])
    red = GradientReducer(net, bucket_bytes=1 << 20)          # one bucket: a | b | c in one flat buffer
    red.max_gap = 8                                           # elements: b's weight (240) is a "large" gap here
    assert len(red.buckets) == 1
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))[rank::world]
    results = {}
    uses = [(a, b, c), (a, c), (a, c), (a, b, c), (c,)]      # the set shrinks, stays, grows, shrinks again
    for it, used in enumerate(uses):
        net.zero_grad(set_to_none=True)
        if it == 2:
            # a stray backward between two bracketed passes (an evaluation-time gradient, say): its hooks must not count
            b(x).sum().backward()
            assert b.weight.grad is not None
            net.zero_grad(set_to_none=True)
            assert not red._fired, 'hooks fired outside begin() .. finish() were recorded'
        red.begin(sync=True, phase='p')
        sum(m(x).square().mean() for m in used).backward()
        red.finish()
        assert red._expected['p'] == {p for m in used for p in m.parameters()}, it
        assert not red._armed and not red._fired
        if it == 1:
            # a and c reduced, b (in the middle of the buffer) not: two spans, and b's slot still holds what iteration 0 left there
            assert red.report['p']['late'] == 0
        results[it] = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
    flat = red.buckets[0].flat
    assert torch.isfinite(flat).all()
    if rank == 0:
        torch.save(results, out)
    dist.destroy_process_group()


def test_reducer_with_a_changing_gradient_set():
    port = 27500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'c.pt')
        mp.spawn(_changing_set_worker, args=(2, port, out), nprocs=2, join=True)
        got = torch.load(out)
    torch.manual_seed(0)
    a, b, c = torch.nn.Linear(6, 40), torch.nn.Linear(6, 40), torch.nn.Linear(6, 40)
    net = torch.nn.ModuleList([a, b, c])
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(5))
    uses = [(a, b, c), (a, c), (a, c), (a, b, c), (c,)]
    for it, used in enumerate(uses):
        net.zero_grad(set_to_none=True)
        # mean over ranks of per-rank means == full-batch mean (equal shard sizes)
        sum(m(x).square().mean() for m in used).backward()
        for n, p in net.named_parameters():
            if p.grad is None:
                assert got[it][n] is None, (it, n)
            else:
                assert torch.allclose(got[it][n], p.grad, atol=1e-6), (it, n)


def _inputs():
    import op_checks as oc
    gen = torch.Generator().manual_seed(77)
    real = torch.rand(GLOBAL_B, 3, SIZE, SIZE, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(GLOBAL_B, 512, generator=gen), torch.randn(GLOBAL_B, 512, generator=gen)
    z_pl = torch.randn(GLOBAL_B // 2, 512, generator=gen)
    pl_noise = torch.randn(GLOBAL_B // 2, 3, SIZE, SIZE, generator=gen)
    return dict(real=real, z_d=z_d, z_g=z_g, z_pl=z_pl, pl_noise=pl_noise, n_d=oc.seeded_noise(SIZE, GLOBAL_B, 1),
                n_g=oc.seeded_noise(SIZE, GLOBAL_B, 2), n_pl=oc.seeded_noise(SIZE, GLOBAL_B // 2, 3))


def _run_iteration(rank, world, device='cpu'):
    import step_checks
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.utils import requires_grad, accumulate
    if device == 'cpu':
        _backend._install_for_tests(EmulatedBackend())
    tr = step_checks.make_trainer(device, size=SIZE, batch=GLOBAL_B)
    assert tr.local_batch == GLOBAL_B // world
    inp = _inputs()
    sh = lambda t: t[rank::world].contiguous().to(device)            # strided shard keeps the stddev groups of the 1-rank run
    shl = lambda maps: [sh(m) for m in maps]
    requires_grad(tr.generator, False); requires_grad(tr.discriminator, True)
    tr.discriminator_step([[sh(inp['z_d'])]], [sh(inp['real'])], noise=shl(inp['n_d']))
    tr.discriminator_regularize_step([sh(inp['real'])])
    requires_grad(tr.generator, True); requires_grad(tr.discriminator, False)
    tr.generator_step([[sh(inp['z_g'])]], noise=shl(inp['n_g']))
    tr.generator_regularize_step(noise=shl(inp['n_pl']), pl_noise=sh(inp['pl_noise']), z=[sh(inp['z_pl'])])
    accumulate(tr.g_ema, tr.generator, tr.accum)
    state = {'g': {k: v.clone() for k, v in tr.generator.named_parameters()},
             'd': {k: v.clone() for k, v in tr.discriminator.named_parameters()},
             'mean_path_length': float(tr.mean_path_length), 'd_loss': tr.reduced_stats()['d_loss']}
    return state


def _trainer_worker(rank, world, port, out, device='cpu'):
    _setup(rank, world, port)
    state = _run_iteration(rank, world, device)
    # every rank draws its own noise from the global generators (NoiseInjection, path-length noise, ADA transforms) ...
    draw = torch.randn(16, device=device).cpu()
    both = [torch.empty(16) for _ in range(world)]
    dist.all_gather(both, draw)
    assert not torch.equal(both[0], both[1]), 'ranks share the global RNG stream: the global batch would see identical injected noise'
    # ... but replicas must stay bit-identical across ranks
    for k, v in state['g'].items():
        ref = v.detach().clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, v.detach()), k
    if rank == 0:
        torch.save({'g': {k: v.detach().cpu() for k, v in state['g'].items()}, 'd': {k: v.detach().cpu() for k, v in state['d'].items()},
                    'mean_path_length': state['mean_path_length'], 'd_loss': state['d_loss']}, out)
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    port = 31500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 's.pt')
        mp.spawn(_trainer_worker, args=(2, port, out), nprocs=2, join=True)
        two = torch.load(out)
    one = _run_iteration(0, 1)
    from gan_control_amd.models.op import _backend
    _backend._install_for_tests(_backend.HipBackend())
    assert abs(two['mean_path_length'] - one['mean_path_length']) < 1e-5 * max(1, abs(one['mean_path_length']))
    assert abs(two['d_loss'] - one['d_loss']) < 1e-5
    worst, big = 0.0, 0
    n = 0
    for tag in ('g', 'd'):
        for k, v in one[tag].items():
            diff = (two[tag][k] - v.detach()).abs()
            worst = max(worst, float(diff.max()))
            big += int((diff > 1e-3).sum()); n += diff.numel()
    # Adam's first steps are sign-like: identical up to fp32 summation order except where a gradient is ~0
    assert big <= n * 1e-3, (big, n, worst)


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_equal_one_rank():
    """The same 2 x 4 == 1 x 8 iteration with BOTH ranks on the HIP kernels of one MI355X (two processes sharing cuda:0; gloo moves the
    gradients, since RCCL refuses two ranks on one device): the bucketed reducer launching from autograd hooks next to real kernels, the
    flat-buffer gradient views under a device optimiser, per-rank device RNG, bit-identical replicas.  What is left for the 8-GPU node is
    RCCL itself."""
    from gan_control_amd.models.op import _backend
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    port = 33500 + os.getpid() % 2000
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    hip = _backend.get()
    assert hip.name == 'hip'
    prev, hip.conv_mode = hip.conv_mode, 'f32'
    try:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, 's.pt')
            mp.spawn(_trainer_worker, args=(2, port, out, 'cuda'), nprocs=2, join=True)
            two = torch.load(out)
        one = _run_iteration(0, 1, 'cuda')
    finally:
        hip.conv_mode = prev
    assert abs(two['mean_path_length'] - one['mean_path_length']) < 1e-4 * max(1, abs(one['mean_path_length']))
    assert abs(two['d_loss'] - one['d_loss']) < 1e-5
    big, n = 0, 0
    for tag in ('g', 'd'):
        for k, v in one[tag].items():
            diff = (two[tag][k] - v.detach().cpu()).abs()
            big += int((diff > 1e-3).sum()); n += diff.numel()
    assert big <= n * 1e-3, (big, n)
