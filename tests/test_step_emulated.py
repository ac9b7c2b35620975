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
