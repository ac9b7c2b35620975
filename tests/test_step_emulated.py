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
