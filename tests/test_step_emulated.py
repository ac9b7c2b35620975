"""Host logic of the trainer on the CPU (emulated C ABI) against the golden iteration."""
import step_checks


def test_step_matches_reference(emu_backend):
    step_checks.check_step('cpu')


def test_fused_discriminator_pair_equals_separate_calls(emu_backend):
    """D over the interleaved fake/real batch == two separate D calls (stddev groups are strided)."""
    import torch
    for batch in (4, 8):
        tr = step_checks.make_trainer('cpu', size=16, batch=batch)
        gen = torch.Generator().manual_seed(3)
        fake = torch.randn(batch, 3, 16, 16, generator=gen)
        real = torch.randn(batch, 3, 16, 16, generator=gen)
        with torch.no_grad():
            tr.fuse_d_pair = True
            f1, r1 = tr.discriminate_pair(fake, real)
            tr.fuse_d_pair = False
            f0, r0 = tr.discriminate_pair(fake, real)
        assert torch.allclose(f1, f0, rtol=1e-5, atol=1e-6) and torch.allclose(r1, r0, rtol=1e-5, atol=1e-6)


def test_iteration_with_ada_and_style_mixing(emu_backend):
    """training_config.augment.enabled (metfaces.json:29-34) + mixing > 0: one full iteration runs, losses finite,
    and the ADA controller sees the real predictions."""
    import random
    import torch
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    cfg = default_config(16, 4)
    cfg['training_config']['augment'] = {'enabled': True, 'ada_target': 0.6, 'ada_length': 500000, 'p': 0.5}
    cfg['training_config']['mixing'] = 0.9
    random.seed(0); torch.manual_seed(0)
    tr = GeneratorTrainer(cfg, device='cpu', seed=0, fused_adam=False)
    real = tr.synthetic_batch()
    tr.train_iteration(0, real)
    stats = tr.reduced_stats()
    for k in ('d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss'):
        assert stats[k] == stats[k] and abs(stats[k]) < 1e6, (k, stats[k])
    assert float(tr.ada.accum[1]) == 4.0 and tr.ada.p == 0.5


def test_train_loop_writes_the_reference_checkpoint_layout_and_resumes(emu_backend, tmp_path):
    """GeneratorTrainer.train (generator_trainer.py:329-355) + save_nets (:852-865): files ``checkpoint/<i:06d>.pt`` at the
    ``save_nets_interval`` cadence with the reference's five keys; a trainer restored from one continues bit-for-bit like the
    uninterrupted run."""
    import os
    import random
    import torch
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

    def fresh():
        cfg = default_config(16, 4)
        cfg['training_config']['iter'] = 5
        cfg['training_config']['save_nets_interval'] = 2
        random.seed(0); torch.manual_seed(0)
        return GeneratorTrainer(cfg, device='cpu', seed=0, fused_adam=False)

    seen = []
    a = fresh()
    real = a.synthetic_batch()
    end = a.train(data=iter(lambda: real, None), save_dir=str(tmp_path), on_iteration=lambda i, t: seen.append(i))
    assert seen == [0, 1, 2, 3, 4] and end == 5
    assert sorted(os.listdir(tmp_path / 'checkpoint')) == ['000000.pt', '000002.pt', '000004.pt']
    ckpt = torch.load(tmp_path / 'checkpoint' / '000002.pt')
    assert {'g', 'd', 'g_ema', 'g_optim', 'd_optim'} <= set(ckpt)
    assert set(ckpt['g']) == set(a.generator.state_dict()) and set(ckpt['d']) == set(a.discriminator.state_dict())
    # resume after iteration 2 with the random streams of the uninterrupted run replayed up to that point
    b = fresh()
    assert torch.equal(b.synthetic_batch(), real)                        # (the same draw from the trainer's own generator as in `a`)
    b.train(data=iter(lambda: real, None), iters=3)                      # iterations 0..2 (draws the same random numbers)
    b.load_state_dict(ckpt)                                              # ... and every tensor of the checkpoint on top
    # the reference's stop test is `i > iter` (:346), so a resumed run ends AFTER iteration `iter`: 3, 4 with iter = 4
    assert b.train(data=iter(lambda: real, None), start_iter=3, iters=4) == 5
    for (n, p), (_, q) in zip(a.generator.named_parameters(), b.generator.named_parameters()):
        assert torch.equal(p, q), n
    for (n, p), (_, q) in zip(a.g_ema.named_parameters(), b.g_ema.named_parameters()):
        assert torch.equal(p, q), n
    assert float(a.mean_path_length) == float(b.mean_path_length)
    # debug runs never write checkpoints (:728)
    c = fresh()
    c.training_config['debug'] = True
    c.train(iters=1, save_dir=str(tmp_path / 'dbg'))
    assert not os.path.exists(tmp_path / 'dbg')


def test_direct_forward_equals_function_apply(emu_backend):
    """op/_backend.py::call -- a Function's forward called directly when autograd would record nothing (grad mode off inside a plain backward,
    or no argument requires a gradient) -- must not change a bit: two iterations (both lazy regularisers) with the short-cut on and off."""
    import random
    import torch
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config

    def run(direct):
        keep, _backend._DIRECT = _backend._DIRECT, direct
        try:
            random.seed(0); torch.manual_seed(0)
            tr = GeneratorTrainer(default_config(16, 4), device='cpu', seed=0, fused_adam=False)
            real = tr.synthetic_batch()
            for i in range(2):
                tr.train_iteration(i * 16, real)
            return tr
        finally:
            _backend._DIRECT = keep

    a, b = run(True), run(False)
    for net in ('generator', 'discriminator', 'g_ema'):
        for (n, p), (_, q) in zip(getattr(a, net).named_parameters(), getattr(b, net).named_parameters()):
            assert torch.equal(p, q), (net, n)
    assert float(a.mean_path_length) == float(b.mean_path_length)


def test_direct_forward_records_no_graph(emu_backend):
    """the direct route runs the forward with grad mode OFF, as Function.apply does -- an ATen op inside a forward that touches a
    tensor requiring a gradient which is NOT an argument must not hand back an output with a graph attached."""
    import torch
    from gan_control_amd.models.op import _backend

    hidden = torch.ones(3, requires_grad=True)

    class Leaky(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            assert not torch.is_grad_enabled()
            return x * hidden

        @staticmethod
        def backward(ctx, g):
            return g

    x = torch.ones(3)
    assert _backend._DIRECT
    y = _backend.call(Leaky, x)
    assert not y.requires_grad and y.grad_fn is None
    assert torch.is_grad_enabled()
    with torch.no_grad():
        y = _backend.call(Leaky, x)
        assert not torch.is_grad_enabled()
    assert not y.requires_grad
    y = _backend.call(Leaky, x.clone().requires_grad_(True))
    assert y.requires_grad and y.grad_fn is not None


def test_named_params_cache_notices_a_replaced_parameter():
    """trainers/utils.py::named_params -- a parameter swapped in the MIDDLE of the list is found by the periodic complete
    re-listing, load_state_dict(assign=True) by its hook, and an entry does not keep its network alive."""
    import gc
    import weakref
    import torch
    from torch import nn
    from gan_control_amd.trainers import utils as tu

    net = nn.Sequential(nn.Linear(2, 2), nn.Linear(2, 2), nn.Linear(2, 2))
    first = tu.named_params(net)
    assert tu.named_params(net) is first
    net[1].weight = nn.Parameter(torch.zeros(2, 2))                 # neither the first nor the last parameter
    for _ in range(tu._FULL_CHECK_EVERY):
        lst = tu.named_params(net)
    assert lst is not first and dict(lst)['1.weight'] is net[1].weight
    keep = tu.named_params(net)
    net.load_state_dict({k: v.clone() for k, v in net.state_dict().items()}, assign=True)
    assert tu.named_params(net) is not keep and dict(tu.named_params(net))['1.bias'] is net[1].bias
    net[2].bias = nn.Parameter(torch.ones(2))                       # the last one: the O(1) probe
    assert dict(tu.named_params(net))['2.bias'] is net[2].bias
    ref = weakref.ref(net)
    del net, first, lst, keep
    gc.collect()
    assert ref() is None


def test_noise_mode_zeros_leaves_unused_strengths_without_adam_state(emu_backend):
    """with g_noise_mode 'zeros' the reference never gives ``*.noise.weight`` of the affected layers a gradient (the fixture's
    None set of a plain backward through the reference generator), so its Adam holds no state for them; the trainer's zero-fill of missing
    gradients must leave them alone."""
    import torch
    import op_checks as oc
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    cfg = default_config(32, 4)
    cfg['model_config']['g_noise_mode'] = 'zeros'
    tr = GeneratorTrainer(cfg, device='cpu', seed=0, fused_adam=False)
    gold = oc.load_golden('noise_modes')
    assert sorted(tr.unused_g) == [str(n) for n in gold['zeros/none_grad']]
    real = tr.synthetic_batch()
    tr.train_iteration(0, real)                  # D step, R1, G step, path length
    named = dict(tr.generator.named_parameters())
    for n, p in named.items():
        if n in tr.unused_g:
            assert p.grad is None and p not in tr.g_optim.state, n
            assert float(p.detach().abs().max()) == 0.0
        else:
            assert p in tr.g_optim.state, n
    # the default mode keeps the zero-not-None behaviour for every parameter
    tr = GeneratorTrainer(default_config(32, 4), device='cpu', seed=0, fused_adam=False)
    assert tr.unused_g == set()
