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


def _accumulation_worker(rank, world, port, out):
    """Two micro-batches that touch DIFFERENT parameter subsets, finish() after each (the non-synchronising one first): the gradient the
    first micro-batch alone produced must still be part of the reduction the second one launches."""
    _setup(rank, world, port)
    from gan_control_amd.trainers.ddp import GradientReducer
    torch.manual_seed(0)
    a, b = torch.nn.Linear(6, 20), torch.nn.Linear(6, 20)
    net = torch.nn.ModuleList([a, b])
    red = GradientReducer(net, bucket_bytes=256)
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(9))[rank::world]
    results = []
    for it in range(3):
        net.zero_grad(set_to_none=True)
        red.begin(sync=False, phase='acc'); a(x[:2]).square().mean().backward(); red.finish()      # only a
        assert red._fired == set(a.parameters())
        red.begin(sync=True, phase='acc'); b(x[2:]).square().mean().backward(); red.finish()       # only b
        assert not red._fired
        results.append({n: p.grad.clone() for n, p in net.named_parameters()})
    if rank == 0:
        torch.save(results, out)
    dist.destroy_process_group()


def test_accumulation_over_different_parameter_subsets():
    port = 25500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'a.pt')
        mp.spawn(_accumulation_worker, args=(2, port, out), nprocs=2, join=True)
        got = torch.load(out)
    torch.manual_seed(0)
    a, b = torch.nn.Linear(6, 20), torch.nn.Linear(6, 20)
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(9))
    # rank r holds x[r::2] = 4 rows: its first micro-batch is rows 0, 1 of the shard, the second rows 2, 3
    sh = [x[r::2] for r in range(2)]
    (sum(a(s[:2]).square().mean() for s in sh) / 2).backward()
    (sum(b(s[2:]).square().mean() for s in sh) / 2).backward()
    want = {'0.weight': a.weight.grad, '0.bias': a.bias.grad, '1.weight': b.weight.grad, '1.bias': b.bias.grad}
    for it in range(3):
        for n, g in want.items():
            assert torch.allclose(got[it][n], g, atol=1e-6), (it, n)


class _Interleaved(torch.nn.Module):
    """Registration order unlike the order of use (the generator registers conv1, convs, to_rgbs, ... and uses them interleaved)."""

    def __init__(self, n=8, width=48):
        super().__init__()
        self.heads = torch.nn.ModuleList([torch.nn.Linear(width, 3) for _ in range(n)])        # registered first, used all along
        self.body = torch.nn.ModuleList([torch.nn.Linear(width, width) for _ in range(n)])

    def forward(self, x):
        out = 0
        for body, head in zip(self.body, self.heads):
            x = torch.tanh(body(x))
            out = out + head(x)
        return out


def _arrival_worker(rank, world, port, out):
    _setup(rank, world, port)
    from gan_control_amd.trainers.ddp import GradientReducer
    torch.manual_seed(0)
    net = _Interleaved()
    red = GradientReducer(net, bucket_bytes=2 * 48 * 48 * 4)
    before = [list(b.params) for b in red.buckets]
    x = torch.randn(8, 48, generator=torch.Generator().manual_seed(3))[rank::world]
    reports = []
    for it in range(3):
        net.zero_grad(set_to_none=True)
        red.begin(sync=True, phase='p'); net(x).square().mean().backward(); red.finish()
        reports.append(dict(red.report['p']))
    index = {id(p): i for i, p in enumerate(net.parameters())}
    flat_order = [index[id(p)] for b in red.buckets for p in b.params]
    assert flat_order == red.arrival_order                       # buckets hold the gradients in the order backward produced them
    assert [[id(p) for p in b.params] for b in red.buckets] != [[id(p) for p in ps] for ps in before]       # ... which is not reverse registration order for this network
    if rank == 0:
        torch.save({'reports': reports, 'order': flat_order, 'grads': {n: p.grad.clone() for n, p in net.named_parameters()},
                    'n_buckets': len(red.buckets)}, out)
    dist.destroy_process_group()


def test_buckets_follow_the_observed_arrival_order():
    """After the first pass the buckets are re-laid in arrival order; from the second pass on every bucket launches from a hook, bucket k as
    soon as the gradients of buckets 0..k have arrived -- all but the last before the final 10 % of backward."""
    port = 23500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'o.pt')
        mp.spawn(_arrival_worker, args=(2, port, out), nprocs=2, join=True)
        got = torch.load(out)
    assert got['n_buckets'] >= 4
    for rep in got['reports'][1:]:
        assert rep['finish'] == 0 and rep['late'] == 0 and rep['hook'] == got['n_buckets'], rep
        at, total = rep['launched_at'], rep['gradients']
        assert at == sorted(at) and at[-1] == total
        assert all(a <= 0.9 * total for a in at[:-1]), (at, total)
        # evenly spread: a bucket never waits for a gradient that arrives after the next bucket's members
        per = total / len(at)
        assert all(abs(a - per * (i + 1)) <= per for i, a in enumerate(at)), (at, total)
    torch.manual_seed(0)
    net = _Interleaved()
    x = torch.randn(8, 48, generator=torch.Generator().manual_seed(3))
    net(x).square().mean().backward()
    # (the mean over ranks of the per-rank means of equal shards is the full-batch mean)
    for n, p in net.named_parameters():
        assert torch.allclose(got['grads'][n], p.grad, atol=1e-6), n


def _tail_worker(rank, world, port, out):
    _setup(rank, world, port)
    from gan_control_amd.trainers.ddp import GradientReducer
    torch.manual_seed(0)
    net = _Interleaved()
    big = 2 * 48 * 48 * 4
    os.environ['GANCONTROL_BUCKET_MB'] = str(big / (1 << 20))
    os.environ['GANCONTROL_LAST_BUCKET_MB'] = str(9500 / (1 << 20))         # body.0 (48 x 48 weights + 48 biases = 9408 bytes), the layer backward reaches last, alone
    try:
        red = GradientReducer(net)
    finally:
        del os.environ['GANCONTROL_BUCKET_MB'], os.environ['GANCONTROL_LAST_BUCKET_MB']
    assert red.bucket_bytes == big and red.last_bucket_bytes == 9500
    x = torch.randn(8, 48, generator=torch.Generator().manual_seed(3))[rank::world]
    for it in range(3):
        net.zero_grad(set_to_none=True)
        red.begin(sync=True, phase='p'); net(x).square().mean().backward(); red.finish()
    rep = dict(red.report['p'])
    sizes = [sum(p.numel() * 4 for p in b.params) for b in red.buckets]
    if rank == 0:
        torch.save({'report': rep, 'sizes': sizes, 'grads': {n: p.grad.clone() for n, p in net.named_parameters()},
                    'last': [n for n, p in net.named_parameters() if any(p is q for q in red.buckets[-1].params)]}, out)
    dist.destroy_process_group()


def test_small_last_bucket():
    """GANCONTROL_LAST_BUCKET_MB (bench.py --last-bucket-mb): the gradients that arrive last get a bucket of their own, so the one collective
    backward cannot hide is short; every bucket still launches from a hook and the reduced gradients are the full-batch ones."""
    port = 25500 + os.getpid() % 2000
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'o.pt')
        mp.spawn(_tail_worker, args=(2, port, out), nprocs=2, join=True)
        got = torch.load(out)
    assert got['sizes'][-1] == 9408 < min(got['sizes'][:-1]), got['sizes']
    assert sorted(got['last']) == ['body.0.bias', 'body.0.weight'], got['last']
    rep = got['report']
    assert rep['finish'] == 0 and rep['late'] == 0 and rep['hook'] == len(got['sizes'])
    assert rep['launched_at'][-1] == rep['gradients']
    torch.manual_seed(0)
    net = _Interleaved()
    x = torch.randn(8, 48, generator=torch.Generator().manual_seed(3))
    net(x).square().mean().backward()
    for n, p in net.named_parameters():
        assert torch.allclose(got['grads'][n], p.grad, atol=1e-6), n


def _inputs(size=SIZE, global_b=GLOBAL_B):
    import op_checks as oc
    gen = torch.Generator().manual_seed(77)
    real = torch.rand(global_b, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(global_b, 512, generator=gen), torch.randn(global_b, 512, generator=gen)
    z_pl = torch.randn(global_b // 2, 512, generator=gen)
    pl_noise = torch.randn(global_b // 2, 3, size, size, generator=gen)
    return dict(real=real, z_d=z_d, z_g=z_g, z_pl=z_pl, pl_noise=pl_noise, n_d=oc.seeded_noise(size, global_b, 1),
                n_g=oc.seeded_noise(size, global_b, 2), n_pl=oc.seeded_noise(size, global_b // 2, 3))


def _run_iteration(rank, world, device='cpu', size=SIZE, global_b=GLOBAL_B):
    import step_checks
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.utils import requires_grad, accumulate
    if device == 'cpu':
        _backend._install_for_tests(EmulatedBackend())
    tr = step_checks.make_trainer(device, size=size, batch=global_b)
    assert tr.local_batch == global_b // world
    inp = _inputs(size, global_b)
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
    if tr.g_reducer is not None:
        state['reports'] = {'g': dict(tr.g_reducer.report), 'd': dict(tr.d_reducer.report),
                            'buckets': (len(tr.g_reducer.buckets), len(tr.d_reducer.buckets))}
    return state


def _trainer_worker(rank, world, port, out, device='cpu', size=SIZE, global_b=GLOBAL_B):
    _setup(rank, world, port)
    if world > 4:
        torch.set_num_threads(1)
    state = _run_iteration(rank, world, device, size, global_b)
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
                    'mean_path_length': state['mean_path_length'], 'd_loss': state['d_loss'], 'reports': state.get('reports')}, out)
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


@pytest.mark.parametrize('world,size,global_b', [(4, 16, 16), (8, 16, 32)])
def test_four_and_eight_ranks_equal_one_rank(world, size, global_b):
    """The bucket / phase logic at the world sizes of BASELINE configs 3 and 4: 4 ranks x 4 images, and 8 ranks x 4 images = global batch
    32 (config 3's literal partitioning, at 16 x 16 so that eight CPU processes finish in half a minute), against the one-rank iteration over
    the whole batch.  Every rank's shard is one complete minibatch-stddev group (members are `global_b / 4` apart)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    port = 35500 + os.getpid() % 2000 + world
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 's.pt')
        mp.spawn(_trainer_worker, args=(world, port, out, 'cpu', size, global_b), nprocs=world, join=True)
        many = torch.load(out)
    one = _run_iteration(0, 1, 'cpu', size, global_b)
    from gan_control_amd.models.op import _backend
    _backend._install_for_tests(_backend.HipBackend())
    assert abs(many['mean_path_length'] - one['mean_path_length']) < 1e-5 * max(1, abs(one['mean_path_length']))
    assert abs(many['d_loss'] - one['d_loss']) < 1e-5
    big = n = 0
    for tag in ('g', 'd'):
        for k, v in one[tag].items():
            diff = (many[tag][k] - v.detach()).abs()
            big += int((diff > 1e-3).sum()); n += diff.numel()
    assert big <= n * 1e-3, (big, n)
    # all four phases reduced something, nothing arrived late
    rep = many['reports']
    assert set(rep['d']) >= {'d', 'r1'} and set(rep['g']) >= {'g', 'pl'}, rep
    assert all(r['late'] == 0 for net in ('g', 'd') for r in rep[net].values())


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
