"""The BASELINE.json configurations that earlier rounds only ran at reduced size, at their own size ("configs_untested"):

* config 4 -- MetFaces 1024 x 1024, 4 images per GPU, ADA forced on (p = 0.6) and style mixing 0.9: one full iteration on the HIP
  kernels, and ``augment()`` of a [4, 3, 1024, 1024] batch against oracle/augment.py evaluated in float64 on the same matrices;
* config 5 -- AFHQ 512 x 512 (one full iteration from afhq.json's fields) and its controller at batch 128
  (controller_configs/afhq/default_w_latent_controller.json) against three Adam steps of the reference's own FcStack
  (tests/golden/controller_afhq.npz, oracle/make_golden.py::golden_controller_afhq);
* config 1's literal arithmetic -- ``--precision bf16`` -- at 512 x 512: network forward and the step_512 iteration with tolerances
  derived from measurement (stated next to each assert) instead of the 64 x 64 / 5e-2 / 10 % check of round 2.
"""
import copy
import zlib

import numpy as np
import pytest
import torch
from torch import autograd

from conftest import load_golden, rel_err
import op_checks as oc

DEV = 'cuda'


def _finite(stats, keys):
    for k in keys:
        v = float(stats[k])
        assert v == v and abs(v) < 1e6, (k, v)


# ------------------------------------------------------------------------------------------------ config 4: MetFaces 1024, ADA + mixing
@pytest.mark.gpu
@pytest.mark.parametrize('batch', [4, 16])
def test_metfaces_1024_iteration_with_ada_and_mixing(batch, bf16x3_mode):
    """metfaces.json's hot-path fields at 1024 x 1024 on the GPU -- 4 images (the per-GPU share of batch 16 on 4 GPUs) and the configuration's OWN
    batch of 16 on one GPU (configs/metfaces.json:22-23; 288 GB of HBM hold it without recomputation) --, augmentation
    probability pinned at 0.6 (ADA's target value; `p > 0` fixes it, generator_trainer.py:333-339) and mixing 0.9: iterations 0 (both
    regularisers fire) and 1 run on the HIP kernels -- the 12 x 12 FIR at 2 x (1024 + pad), the bilinear warp and the reflect padding
    included -- with finite losses, both style codes in use and an ADA statistic that moved."""
    import random
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer
    ref = oc.load_configs()['metfaces']
    cfg = copy.deepcopy({'model_config': ref['model_config'], 'training_config': ref['training_config']})
    cfg['model_config']['size'] = 1024
    tc = cfg['training_config']
    tc['batch'] = tc['mini_batch'] = batch
    tc['augment'] = dict(tc['augment'], enabled=True, p=0.6)
    tc['mixing'] = 0.9
    random.seed(3)
    tr = GeneratorTrainer(cfg, device=DEV, seed=0)
    assert tr.ada.p == 0.6 and tc['augment']['enabled']
    two_codes = 0
    sample_z = tr.sample_z
    def counting(batch):
        nonlocal two_codes
        z = sample_z(batch)
        two_codes += len(z) == 2
        return z
    tr.sample_z = counting
    real = tr.synthetic_batch()
    for i in range(2):
        tr.train_iteration(i, real)
        torch.cuda.synchronize()
        _finite(tr.reduced_stats(), ('d_loss', 'g_adv_loss') + (('d_r1_loss', 'g_path_loss') if i == 0 else ()))
    assert two_codes >= 2, 'style mixing never drew two codes'
    assert float(tr.ada.accum[1]) == 2 * batch and tr.stats['ada_aug_p'] == 0.6
    for n, p in list(tr.generator.named_parameters()) + list(tr.discriminator.named_parameters()):
        assert torch.isfinite(p).all(), n
    del tr
    torch.cuda.empty_cache()


def _legal_matrices(nl, p, batch, size, first_seed):
    """The first seed whose sampled transforms need no more reflect padding than the image has (the reference loops for ever on the
    others, non_leaking.py:288-313; the product redraws)."""
    for seed in range(first_seed, first_seed + 50):
        torch.manual_seed(seed)
        G = nl.sample_affine(p, batch, size, size)
        C = nl.sample_color(p, batch)
        px1, px2, py1, py2 = nl.get_padding(torch.inverse(G), size, size)
        if max(px1, px2, py1, py2) + 6 < size:
            return G, C, (px1, px2, py1, py2)
    raise AssertionError('no legal transform in 50 seeds')


@pytest.mark.gpu
def test_augment_1024_against_the_oracle_in_float64():
    """augment() of a [4, 3, 1024, 1024] batch through firK_tile_kernel<12, up 2>, affine_warp_kernel, firK_tile_kernel<12, down 2>,
    reflect_pad_kernel and their adjoints, against oracle/augment.py run in float64 on the SAME matrices (p = 0.6).
    Measured on the MI355X: forward 2.4e-4, input gradient 1.5e-4 of the output range, asserted at 2 x that.  The fixture-size test
    (48 - 64 px) holds 2e-5; the error grows with the image because the bilinear sampling coordinates -- in the reference as here --
    are float32 values of magnitude ~4000 px at this size (2 x (1024 + 2 x 460 px of reflect padding)): 4000 x 2^-24 = 2.4e-4 px, times
    the O(1) difference between neighbouring pixels of a random image.  The float64 oracle does not have that rounding; the reference's
    own float32 path does."""
    from gan_control_amd.trainers import non_leaking as nl
    from oracle import augment as oaug
    size, batch = 1024, 4
    G, C, pads = _legal_matrices(nl, 0.6, batch, size, first_seed=11)
    assert max(pads) > 0, 'a transform that needs no padding does not exercise the reflect path'
    gen = torch.Generator().manual_seed(2)
    img = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    go = torch.randn(batch, 3, size, size, generator=gen)
    x = img.to(DEV).requires_grad_(True)
    out, _ = nl.augment(x, 0.6, (G, C))
    gi, = autograd.grad(out, x, go.to(DEV))
    xr = img.double().requires_grad_(True)
    ref = oaug.augment(xr, G.double(), C.double(), nl.SYM6)
    gr, = autograd.grad(ref, xr, go.double())
    assert out.shape == ref.shape
    e_out, e_gi = rel_err(out, ref), rel_err(gi, gr)
    print('augment 1024: forward %.2e, input gradient %.2e (paddings %s)' % (e_out, e_gi, pads))
    assert e_out < 5e-4 and e_gi < 3e-4, (e_out, e_gi)


# ------------------------------------------------------------------------------------------------ config 5: AFHQ 512 + controller at batch 128
@pytest.mark.gpu
def test_afhq_512_iteration(bf16x3_mode):
    """afhq.json's hot-path fields at their own resolution (512 x 512, split mapping network over dog_id / orientation / other, ADA
    enabled with p starting at 0), 8 images: one full iteration with both regularisers on the HIP kernels."""
    tr = oc.check_config_ingestion(DEV, 'afhq', size=512, batch=8)
    torch.cuda.synchronize()
    assert tr.model_config['size'] == 512 and tr.generator.size == 512


@pytest.mark.gpu
def test_afhq_512_batch16_iteration_in_both_arithmetics():
    """afhq.json at its own resolution AND its own batch (configs/afhq.json:22-23: 512 x 512, 16 images; split mapping network over dog_id /
    orientation / other, ADA enabled): iteration 0 (both regularisers fire) and iteration 1 on the HIP kernels, once in exact fp32 and once in
    split-bf16 from the same seeds.  No reference fixture exists for this configuration (config 2's `step_512_b16` pins the same resolution and
    batch with the regular mapping network): the two arithmetics must agree on every loss of both iterations to 1e-3 -- fp32 being the mode
    that IS pinned against the reference at this size -- and leave finite weights."""
    import copy
    import random
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer
    ref = oc.load_configs()['afhq']
    hip = _backend.get()
    prev = hip.conv_mode
    runs = {}
    try:
        for mode in ('f32', 'bf16x3'):
            hip.conv_mode = mode
            cfg = copy.deepcopy({'model_config': ref['model_config'], 'training_config': ref['training_config']})
            cfg['model_config']['size'] = 512
            assert cfg['training_config']['batch'] == 16 and cfg['training_config']['augment']['enabled']
            random.seed(5); torch.manual_seed(5)
            tr = GeneratorTrainer(cfg, device=DEV, seed=0)
            real = tr.synthetic_batch()
            rec = []
            for i in range(2):
                tr.train_iteration(i, real)
                torch.cuda.synchronize()
                st = tr.reduced_stats()
                _finite(st, ('d_loss', 'g_adv_loss') + (('d_r1_loss', 'g_path_loss') if i == 0 else ()))
                rec.append({k: float(st[k]) for k in ('d_loss', 'g_adv_loss') + (('d_r1_loss', 'g_path_loss') if i == 0 else ())})
            for n, p in list(tr.generator.named_parameters()) + list(tr.discriminator.named_parameters()):
                assert torch.isfinite(p).all(), (mode, n)
            runs[mode] = rec
            del tr
            torch.cuda.empty_cache()
    finally:
        hip.conv_mode = prev
    # iteration 0 starts from identical weights and inputs: the losses differ by the arithmetic only.  Iteration 1 follows four sign-like Adam
    # steps (an element whose gradient is ~0 may step the other way): its losses are held to 2e-2.
    for i, tol in ((0, 1e-3), (1, 2e-2)):
        for k, v in runs['f32'][i].items():
            assert abs(runs['bf16x3'][i][k] - v) <= tol * max(1e-2 if k == 'd_r1_loss' else 1.0, abs(v)), (i, k, runs['bf16x3'][i][k], v)


def controller_procedural_fill_(state_dict):
    """The fill of oracle/make_golden.py::controller_procedural_fill_ (weights from CRC32(key) seeds, stored / lr_mul)."""
    for key in sorted(state_dict):
        gen = torch.Generator().manual_seed(zlib.crc32(('controller/' + key).encode()) & 0x7FFFFFFF)
        v = torch.randn(state_dict[key].shape, generator=gen, dtype=torch.float32)
        v = v * 100.0 if key.endswith('.weight') else v * 0.1
        with torch.no_grad():
            state_dict[key].copy_(v.to(state_dict[key].dtype))
    return state_dict


def _check_afhq_controller(device):
    from gan_control_amd.trainers.controller_trainer import ControllerTrainer
    ref = oc.load_configs()['afhq_controller']
    gold = load_golden('controller_afhq')
    batch, in_dim, mid, n_mlp, out_dim, lo, hi = (int(v) for v in gold['cfg'])
    mc, tc = ref['model_config'], ref['training_config']
    assert (tc['batch'], mc['in_dim'], mc['mid_dim'], mc['n_mlp']) == (batch, in_dim, mid, n_mlp) == (128, 3, 512, 4)
    assert ref['group_chunk'] == [lo, hi] and ref['working_group'] == 'orientation'
    assert tc['losses'] == ['latent_rec', 'latent_adv_', 'attribute_rec_']          # only the first is a loss name the step knows
    tr = ControllerTrainer({'model_config': mc, 'training_config': tc}, ref['group_chunk'], device=device, seed=0)
    tr.fc_controller.load_state_dict(controller_procedural_fill_(tr.fc_controller.state_dict()))
    init = {k: v.detach().clone() for k, v in tr.fc_controller.state_dict().items()}
    gen = torch.Generator().manual_seed(int(gold['input_seed'][0]))
    controls = torch.rand(batch, in_dim, generator=gen) * 2 - 1
    w_latent = torch.randn(batch, mc['latent_size'], generator=gen)
    assert torch.equal(controls, torch.from_numpy(gold['controls'])) and torch.equal(w_latent[:2], torch.from_numpy(gold['w_head']))
    with torch.no_grad():
        y = tr.fc_controller(controls.to(device))
    assert y.shape == (128, hi - lo)
    assert rel_err(y, torch.from_numpy(gold['forward'])) < 1e-4            # four fp32 GEMMs against the float64 reference
    # gradient of the first step (per-parameter norms), then three Adam steps
    tr.fc_controller.zero_grad()
    tr.calc_latent_rec_loss(w_latent.to(device), tr.fc_controller(controls.to(device))).backward()
    names = [str(n) for n in gold['after3/names']]
    grads = dict(tr.fc_controller.named_parameters())
    for n, g_ref in zip(names, gold['grad0/norms']):
        assert abs(float(grads[n].grad.norm()) - g_ref) <= 1e-3 * g_ref, (n, float(grads[n].grad.norm()), g_ref)
    losses = [tr.controller_update((controls, w_latent)) for _ in range(3)]
    assert np.allclose(losses, gold['losses'], rtol=1e-5), (losses, gold['losses'])
    # Parameters: weights are stored / lr_mul (magnitude ~100) and move by ~lr per step, so compare the UPDATE, not the value.
    # Adam's first steps are sign-like: an element whose gradient is within rounding of zero may step the other way.  The L1 gradient is
    # a sum of 128 x 192 signed activations / (128 * 192): elements with |g| below fp32 rounding of that sum are < 1e-3 of a tensor, so at
    # most 2 of 512 samples per tensor may differ (counted below), and the norm of a tensor's total movement agrees to 1 %.
    lr = tc['lr'] * tc['reg_every'] / (tc['reg_every'] + 1)
    after = tr.fc_controller.state_dict()
    for n, moved in zip(names, gold['after3/moved_norm']):
        got = float((after[n] - init[n]).double().norm())
        assert abs(got - moved) <= 1e-2 * moved, (n, got, moved)
        idx = torch.from_numpy(gold['after3/idx/' + n])
        delta_ref = torch.from_numpy(gold['after3/val/' + n]) - init[n].cpu().double().reshape(-1)[idx]
        delta = (after[n] - init[n]).cpu().double().reshape(-1)[idx]
        # values of magnitude 100 carry an fp32 ulp of 7.6e-6: half a step (lr / 2 = 8e-4) separates "same direction" from "flipped"
        flipped = int(((delta - delta_ref).abs() > 0.5 * lr).sum())
        assert flipped <= 2, (n, flipped)
    assert tr.evaluation_dict['latent_rec_loss'] == losses[-1]


def test_afhq_controller_batch_128_emulated(emu_backend):
    _check_afhq_controller('cpu')


@pytest.mark.gpu
def test_afhq_controller_batch_128_hip():
    _check_afhq_controller(DEV)


# ------------------------------------------------------------------------------------------------ config 1's literal arithmetic: --precision bf16 at 512 x 512
# Plain bf16 products (one MFMA per product on bf16-rounded operands, fp32 accumulate / storage / master weights) are NOT a parity mode:
# they carry ~3e-3 relative error per layer by construction.  What is asserted is 2 x the error MEASURED on the MI355X against the
# reference's 512 x 512 fixtures (tools/bf16_error_probe.py -> profiles/bf16_errors_r03.json holds the measured values), with a floor of
# 1e-4 (1e-5 for the mapping network's output, which no convolution touches) under quantities whose measured error is smaller still.
# The "seq ..." statistics are those of the SEQUENTIAL iteration, i.e. after one to three sign-like Adam steps: in this arithmetic a
# one-ulp change anywhere upstream re-rolls thousands of bf16 roundings and with them the signs of near-zero gradients, so these
# statistics scatter from build to build (g_path_loss: 4e-4 and 7e-3 measured on two builds); they are bounded at 3e-2.  The isolated
# passes ("iso ...", "gradnorm/d"), which start from the fixture's weights, are held to 2 x their measured error.
# measured (profiles/bf16_errors_r03.json, 'bf16'): {"thumb": 0.00685, "pixels": 0.00459, "mean": 0.000181, "std": 0.000526, "logits": 0.00909, "w0": 6.96e-07}
BF16_NETWORK_512 = {"thumb": 0.014, "pixels": 0.0092, "mean": 0.00037, "std": 0.0011, "logits": 0.019, "w0": 1e-05}
# measured: {"gradnorm/d:global": 0.00524, "gradnorm/d:param": 0.0137, "iso/r1:global": 4.19e-05, "iso/r1:param": 0.287, "iso d_r1_loss": 7.33e-05, "iso/g:global": 0.0376, "iso/g:param": 0.134, "iso g_adv_loss": 0.00252, "iso/pl:global": 0.00993, "iso/pl:param": 0.0261, "iso g_path_loss": 0.00896, "iso path_lengths": 0.00567, "seq d_loss": 0.00304, "seq d_r1_loss": 1.45e-06, "seq g_adv_loss": 0.000336, "seq g_path_loss": 0.000438, "seq g_mean_path_length": 3.61e-05, "seq path_lengths": 0.00505}
# round 5 (transposed layers split over K, exact-fp32 weight gradients on the <= 8 x 8 planes, edge kernel: other summation orders): gradnorm/d:global 5.46e-03,
# iso/r1:global 3.47e-04 (a norm over ~1e-2 random per-parameter errors: it moved from 4.2e-5, bound 2 x the new value), iso/g:global 4.08e-02, iso/pl:global 2.12e-03, iso g_path_loss 4.49e-03
BF16_STEP_512 = {"gradnorm/d:global": 0.011, "gradnorm/d:param": 0.028, "iso/r1:global": 0.0007, "iso/r1:param": 0.58, "iso d_r1_loss": 0.00015, "iso/g:global": 0.076, "iso/g:param": 0.27, "iso g_adv_loss": 0.0051, "iso/pl:global": 0.02, "iso/pl:param": 0.053, "iso g_path_loss": 0.018, "iso path_lengths": 0.012, "seq d_loss": 0.03, "seq d_r1_loss": 0.001, "seq g_adv_loss": 0.03, "seq g_path_loss": 0.03, "seq g_mean_path_length": 0.03, "seq path_lengths": 0.03}
BF16_STEP_512_OUTLIERS = 12         # sampled parameters that land elsewhere after the four Adam steps: 3 and 4 of 582 measured


@pytest.mark.gpu
def test_bf16_mode_network_512(bf16_mode):
    errs = oc.network_errors(512, DEV)
    print('bf16 network errors at 512:', {k: '%.2e' % v for k, v in errs.items()})
    for k, bound in BF16_NETWORK_512.items():
        assert errs[k] <= bound, (k, errs[k], bound)


@pytest.mark.gpu
def test_bf16_mode_step_512(bf16_mode):
    import step_checks
    m = step_checks._Measure()
    step_checks.check_step(DEV, name='step_512', measure=m)
    print('bf16 step_512 errors:', {k: ('%.2e' % v if isinstance(v, float) else v) for k, v in m.items() if k != '_bounds'})
    for k, bound in BF16_STEP_512.items():
        assert m[k] <= bound, (k, m[k], bound)
    assert m['param outliers'] <= BF16_STEP_512_OUTLIERS, m['param outliers']


# ------------------------------------------------------------------------------------------------ config 2 at its own batch
@pytest.mark.gpu
def test_config2_512_batch16_iterations_in_both_arithmetics():
    """BASELINE config 2 -- FFHQ 512 x 512, batch 16 on ONE GPU -- at its own batch (the reference fixtures of this size stop at batch 4:
    `step_512.npz`): the trainer as `bench.py --size 512 --batch-per-gpu 16` builds it runs iterations 0 (both lazy regularisers) and 1 in
    split-bf16 and in exact fp32 arithmetic from the same seeds; every statistic is finite and the two arithmetics agree on the losses of the
    FIRST iteration (identical weights and inputs there; later ones inherit sign-like Adam steps) to the tolerance of the 512 x 512 step fixture."""
    import random
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    hip = _backend.get()
    prev = hip.conv_mode
    stats = {}
    try:
        for mode in ('f32', 'bf16x3'):
            hip.conv_mode = mode
            random.seed(0); torch.manual_seed(0)
            tr = GeneratorTrainer(default_config(512, 16), device=DEV, seed=0)
            assert tr.local_batch == 16
            real = tr.synthetic_batch()
            tr.train_iteration(0, real)
            first = {k: float(v) for k, v in tr.reduced_stats().items()}
            tr.train_iteration(1, real)
            _finite(tr.reduced_stats(), ('d_loss', 'g_adv_loss', 'd_r1_loss', 'g_path_loss', 'g_mean_path_length'))
            assert all(bool(torch.isfinite(p).all()) for p in tr.generator.parameters())
            stats[mode] = first
            del tr
            torch.cuda.empty_cache()
    finally:
        hip.conv_mode = prev
    for k in ('d_loss', 'g_adv_loss', 'd_r1_loss', 'g_path_loss', 'g_mean_path_length'):
        a, b = stats['bf16x3'][k], stats['f32'][k]
        assert abs(a - b) <= 6e-3 * max(1.0, abs(b)), (k, a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['f32', 'bf16x3'])
def test_config2_512_batch16_against_the_reference(mode):
    """BASELINE config 2 at its OWN batch against the reference (VERDICT r4 "missing" 4): tests/golden/step_512_b16.npz holds the four backward
    passes of an iteration at 512 x 512, 16 images, from the reference's own modules and trainer maths (oracle/make_golden.py::
    golden_step_isolated_chunked -- accumulated over complete minibatch-stddev groups because the joint graph does not fit the build container;
    that procedure is held against the joint passes in float64 at 32 x 32 every time the fixture is made: 5e-14).  The trainer is built as
    `bench.py --size 512 --batch-per-gpu 16` builds it; checks and tolerances are those of the headline test (losses, path lengths, None-gradient
    sets, global and per-parameter gradient norms, 64 sampled gradient elements per tensor)."""
    import step_checks
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    hip = _backend.get()
    prev, hip.conv_mode = hip.conv_mode, mode
    try:
        step_checks.check_isolated(DEV, name='step_512_b16', tol=2e-3, param_tol=None if mode == 'f32' else 5e-3,
                                   trainer=lambda size, batch: GeneratorTrainer(default_config(size, batch), device=DEV, seed=0),
                                   ratchet=step_checks.load_measured().get('%s/' % 'step_512_b16' + mode))
    finally:
        hip.conv_mode = prev
        torch.cuda.empty_cache()
