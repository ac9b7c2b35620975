"""Pin the oracle against the real reference and write the golden vectors.

Run ONLY in the build container (``python -m oracle.make_golden`` from the repo
root): it imports the reference's Python from /root/reference/src (which does not
exist on the GPU box), asserts that every oracle function reproduces the reference,
and writes small ``.npz`` fixtures under tests/golden/.  Nothing from the reference
is copied: the fixtures are inputs and expected outputs only.
"""
import math
import os
import sys
import types
from unittest import mock

import numpy as np
import torch
import torch.nn.functional as F
from torch import autograd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference/src')

from oracle import ops, networks, step  # noqa: E402

import gan_control.models.gan_model as ref_gm  # noqa: E402
from gan_control.trainers import utils as ref_tu  # noqa: E402


def _import_ref_trainer():
    """generator_trainer.py needs tensorboard/torchvision/etc.; stub those so its maths can be called."""
    stubs = {}
    for name in ['torch.utils.tensorboard', 'torchvision', 'torchvision.transforms', 'torchvision.utils',
                 'gan_control.fid_utils.calc_inception', 'gan_control.losses.loss_model',
                 'gan_control.evaluation.tracker', 'gan_control.utils.mini_batch_random_multi_split_utils']:
        if name not in sys.modules:
            stubs[name] = mock.MagicMock()
    op = types.ModuleType('gan_control.models.op')
    op.upfirdn2d = ref_gm.upfirdn2d
    stubs['gan_control.models.op'] = op
    with mock.patch.dict(sys.modules, stubs):
        import gan_control.trainers.generator_trainer as gt
        import gan_control.trainers.non_leaking as nl
    return gt, nl


REF_GT, REF_NL = _import_ref_trainer()
RT = REF_GT.GeneratorTrainer


def close(a, b, tol, what):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
    assert err <= tol, f'{what}: oracle differs from reference, rel err {err:.3e} > {tol}'
    return err


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def grads3(fn, inputs, gen):
    """Forward, first-order grads for a fixed cotangent, and second-order grads of <g_in0, v>."""
    inputs = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(*inputs)
    go = torch.randn(out.shape, generator=gen).requires_grad_(True)
    gi = autograd.grad(out, inputs, go, create_graph=True, allow_unused=True)
    v = torch.randn(gi[0].shape, generator=gen)
    gg = autograd.grad((gi[0] * v).sum(), [go] + inputs, allow_unused=True)
    rec = {'out': out, 'go': go, 'v': v}
    for i, g in enumerate(gi):
        if g is not None:
            rec[f'gi{i}'] = g
    for i, g in enumerate(gg):
        if g is not None:
            rec[f'gg{i}'] = g       # gg0 = d/d(go); gg{i+1} = d/d(inputs[i])
    return rec


# --------------------------------------------------------------------------------------
def golden_upfirdn2d():
    gen = torch.Generator().manual_seed(1234)
    k4 = ref_gm.make_kernel([1, 3, 3, 1])
    sym6 = torch.tensor(REF_NL.SYM6, dtype=torch.float32)
    k12 = ref_gm.make_kernel(sym6)
    krect = torch.randn(3, 5, generator=gen)
    cases = {
        # name: (shape, kernel, up, down, pad)   -- SURVEY 8a-1 argument sets (i)-(v) + ADA pair + edge cases
        'i_torgb_up': ((2, 3, 8, 8), k4 * 4, 2, 1, (2, 1)),
        'ii_g_blur': ((2, 5, 17, 17), k4 * 4, 1, 1, (1, 1)),
        'iii_d_blur3': ((2, 4, 16, 16), k4, 1, 1, (2, 2)),
        'iv_d_blur1': ((2, 4, 16, 16), k4, 1, 1, (1, 1)),
        'v_down': ((2, 3, 16, 16), k4, 1, 2, (1, 1)),
        'ada_up12': ((1, 3, 11, 9), k12 * 4, 2, 1, (2 + 5, 1 + 5)),
        'ada_down12': ((1, 3, 40, 36), k12, 1, 2, (-5 + 2, -5 + 2)),
        'odd_rect': ((1, 2, 7, 13), krect, 1, 1, (2, 1)),
        'crop_neg': ((1, 2, 12, 10), k4, 1, 1, (-1, 0)),
        'up3_down2': ((1, 1, 5, 6), k4, 3, 2, (3, 2)),
        'wide_row': ((1, 1, 3, 300), k4, 1, 1, (2, 1)),
        'one_pixel': ((1, 1, 1, 1), k4 * 4, 2, 1, (2, 1)),
    }
    out = {}
    for name, (shape, k, up, down, pad) in cases.items():
        x = torch.randn(*shape, generator=gen)
        ref = ref_gm.upfirdn2d(x, k, up=up, down=down, pad=pad)
        close(ops.upfirdn2d(x, k, up, down, pad), ref, 1e-6, f'upfirdn2d/{name}')
        rec = grads3(lambda t: ref_gm.upfirdn2d(t, k, up=up, down=down, pad=pad), [x], gen)
        rec_o = grads3(lambda t: ops.upfirdn2d(t, k, up, down, pad), [x], torch.Generator().manual_seed(0))
        assert rec_o['out'].shape == rec['out'].shape
        out.update({f'{name}/x': x, f'{name}/k': k, f'{name}/args': np.array([up, down, pad[0], pad[1]])})
        out.update({f'{name}/{kk}': vv for kk, vv in rec.items()})
    np.savez_compressed(os.path.join(GOLD, 'upfirdn2d.npz'), **to_np(out))
    print('upfirdn2d: %d cases' % len(cases))


def golden_bias_act():
    gen = torch.Generator().manual_seed(99)
    out = {}
    for name, shape in {'4d': (2, 6, 5, 7), '2d': (3, 10), '4d_big': (1, 3, 33, 65)}.items():
        x = torch.randn(*shape, generator=gen)
        x.view(-1)[::7] = 0.0            # exercise the x + b == 0 corner
        b = torch.randn(shape[1], generator=gen)
        b[0] = 0.0
        ref = ref_gm.fused_leaky_relu(x, b)
        close(ops.fused_leaky_relu(x, b), ref, 1e-7, f'fused_leaky_relu/{name}')
        if len(shape) == 4:
            m = ref_gm.FusedLeakyReLU(shape[1])
            with torch.no_grad():
                m.bias.copy_(b)
            close(m(x), ref, 1e-7, 'FusedLeakyReLU module')
        rec = grads3(lambda t, bb: ref_gm.fused_leaky_relu(t, bb), [x, b], gen)
        out.update({f'{name}/x': x, f'{name}/b': b})
        out.update({f'{name}/{kk}': vv for kk, vv in rec.items()})
    np.savez_compressed(os.path.join(GOLD, 'bias_act.npz'), **to_np(out))
    print('bias_act ok')


def golden_convs():
    gen = torch.Generator().manual_seed(4321)
    out = {}
    # --- ModulatedConv2d: plain / upsample / 1x1 no-demod (ToRGB) ------------------------------
    sdim = 16
    for name, (b, ic, oc, k, h, demod, up) in {
        'mod_plain': (3, 8, 6, 3, 7, True, False),
        'mod_up': (2, 8, 6, 3, 5, True, True),
        'mod_rgb': (2, 8, 3, 1, 6, False, False),
        'mod_plain_wide': (2, 16, 40, 3, 12, True, False),
        'mod_up_wide': (1, 40, 16, 3, 9, True, True),
    }.items():
        m = ref_gm.ModulatedConv2d(ic, oc, k, sdim, demodulate=demod, upsample=up, conv_transpose=True)
        with torch.no_grad():
            m.weight.copy_(torch.randn(m.weight.shape, generator=gen))
            m.modulation.weight.copy_(torch.randn(m.modulation.weight.shape, generator=gen))
            m.modulation.bias.copy_(1 + 0.1 * torch.randn(ic, generator=gen))
        x = torch.randn(b, ic, h, h, generator=gen)
        st = torch.randn(b, sdim, generator=gen)

        def ref_fn(x_, st_, w_, mw_, mb_):
            return torch.func.functional_call(m, {'weight': w_, 'modulation.weight': mw_, 'modulation.bias': mb_}, (x_, st_))

        def ora_fn(x_, st_, w_, mw_, mb_):
            return ops.modulated_conv2d(x_, st_, w_, mw_, mb_, demodulate=demod, upsample=up)

        ins = [x, st, m.weight.detach(), m.modulation.weight.detach(), m.modulation.bias.detach()]
        close(ora_fn(*ins), ref_fn(*ins), 2e-6, f'modconv/{name}')
        rec = grads3(ref_fn, ins, gen)
        rec_o = grads3(ora_fn, ins, torch.Generator().manual_seed(0))
        assert set(rec) == set(rec_o)
        for i, nm in enumerate(['x', 'style', 'weight', 'mod_weight', 'mod_bias']):
            out[f'{name}/{nm}'] = ins[i]
        out[f'{name}/cfg'] = np.array([int(demod), int(up), k])
        out.update({f'{name}/{kk}': vv for kk, vv in rec.items()})
    # --- EqualConv2d: s1 p1 / s2 p0 / 1x1 s2 / 1x1 s1 with bias ------------------------------------
    for name, (b, ic, oc, k, h, stride, padding, bias) in {
        'conv_s1p1': (2, 5, 7, 3, 9, 1, 1, False),
        'conv_s2p0': (2, 5, 7, 3, 11, 2, 0, False),
        'conv_1x1s2': (2, 6, 4, 1, 9, 2, 0, False),
        'conv_1x1_bias': (2, 3, 8, 1, 10, 1, 0, True),
        'conv_s1p1_wide': (1, 40, 36, 3, 20, 1, 1, False),
        'conv_s2p0_wide': (1, 36, 40, 3, 21, 2, 0, False),
    }.items():
        m = ref_gm.EqualConv2d(ic, oc, k, stride=stride, padding=padding, bias=bias)
        w = torch.randn(oc, ic, k, k, generator=gen)
        bb = torch.randn(oc, generator=gen) if bias else None
        x = torch.randn(b, ic, h, h, generator=gen)

        def ref_fn(x_, w_, *rest):
            p = {'weight': w_}
            if bias:
                p['bias'] = rest[0]
            return torch.func.functional_call(m, p, (x_,))

        def ora_fn(x_, w_, *rest):
            return ops.equal_conv2d(x_, w_, rest[0] if bias else None, stride=stride, padding=padding)

        ins = [x, w] + ([bb] if bias else [])
        close(ora_fn(*ins), ref_fn(*ins), 2e-6, f'conv/{name}')
        rec = grads3(ref_fn, ins, gen)
        for i, nm in enumerate(['x', 'weight', 'bias'][:len(ins)]):
            out[f'{name}/{nm}'] = ins[i]
        out[f'{name}/cfg'] = np.array([stride, padding, k])
        out.update({f'{name}/{kk}': vv for kk, vv in rec.items()})
    np.savez_compressed(os.path.join(GOLD, 'convs.npz'), **to_np(out))
    print('convs ok')


def golden_misc():
    gen = torch.Generator().manual_seed(777)
    out = {}
    # EqualLinear with lr_mul = 0.01 (+ fused lrelu) and plain
    for name, (n, i, o, lr_mul, act) in {'lin_map': (4, 32, 24, 0.01, 'fused_lrelu'), 'lin_plain': (3, 20, 1, 1.0, None)}.items():
        m = ref_gm.EqualLinear(i, o, lr_mul=lr_mul, activation=act)
        w = torch.randn(o, i, generator=gen) / lr_mul
        b = torch.randn(o, generator=gen)
        x = torch.randn(n, i, generator=gen)
        ref = torch.func.functional_call(m, {'weight': w, 'bias': b}, (x,))
        close(ops.equal_linear(x, w, b, lr_mul=lr_mul, activation=bool(act)), ref, 1e-6, name)
        out.update({f'{name}/x': x, f'{name}/w': w, f'{name}/b': b, f'{name}/out': ref, f'{name}/lr_mul': lr_mul})
    # PixelNorm
    x = torch.randn(3, 16, generator=gen)
    close(ops.pixel_norm(x), ref_gm.PixelNorm()(x), 1e-7, 'pixel_norm')
    out.update({'pixel_norm/x': x, 'pixel_norm/out': ref_gm.PixelNorm()(x)})
    # minibatch stddev, via the reference's _forward_split with identity tail
    d = ref_gm.Discriminator(8)
    for name, b in {'mbstd8': 8, 'mbstd2': 2, 'mbstd4': 4}.items():
        x = torch.randn(b, 16, 4, 4, generator=gen)
        grab = {}
        d._forward_split(x, lambda t: grab.setdefault('cat', t), lambda t: t)
        close(ops.minibatch_stddev(x), grab['cat'], 1e-6, name)
        out.update({f'{name}/x': x, f'{name}/out': grab['cat']})
    # losses, checked against the reference trainer's own methods
    rp, fp = torch.randn(6, 1, generator=gen), torch.randn(6, 1, generator=gen)
    close(step.d_logistic_loss(rp, fp), RT.d_logistic_loss(rp, fp), 1e-7, 'd_logistic_loss')
    close(step.g_nonsaturating_loss(fp), RT.g_nonsaturating_loss(fp), 1e-7, 'g_nonsaturating_loss')
    g = torch.randn(4, 10, 16, generator=gen)
    a = step.path_lengths_and_penalty(g, 0)
    r = RT.g_path_regularize_grad(g, 0)
    for u, v in zip(a, r):
        close(u, v, 1e-7, 'path penalty')
    a = step.path_lengths_and_penalty(g, torch.tensor(0.37))
    r = RT.g_path_regularize_grad(g, torch.tensor(0.37))
    for u, v in zip(a, r):
        close(u, v, 1e-7, 'path penalty (warm)')
    out.update({'loss/real_pred': rp, 'loss/fake_pred': fp, 'loss/d_logistic': RT.d_logistic_loss(rp, fp),
                'loss/g_nonsat': RT.g_nonsaturating_loss(fp), 'loss/path_grad': g,
                'loss/path_penalty': r[0], 'loss/path_mean': r[1], 'loss/path_lengths': r[2]})
    # Adam hyper-parameters (generator_trainer.py:161-173)
    hp = step.adam_hparams(0.002, 4)
    assert abs(hp['lr'] - 0.0016) < 1e-12 and hp['betas'][0] == 0.0 and abs(hp['betas'][1] - 0.99 ** 0.8) < 1e-15
    np.savez_compressed(os.path.join(GOLD, 'misc.npz'), **to_np(out))
    print('misc ok')


# --------------------------------------------------------------------------------------
def seeded_noise(size, batch, seed):
    """Per-layer noise maps for explicit-noise parity runs (shapes as Generator.make_noise gan_model.py:683-696)."""
    gen = torch.Generator().manual_seed(seed)
    log_size = int(math.log2(size))
    maps = [torch.randn(batch, 1, 4, 4, generator=gen)]
    for i in range(3, log_size + 1):
        maps += [torch.randn(batch, 1, 2 ** i, 2 ** i, generator=gen) for _ in range(2)]
    return maps


def thumb(img, n=16):
    return F.adaptive_avg_pool2d(img, min(n, img.shape[-1]))


def sample_pixels(img, count=64, seed=5):
    gen = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, img.numel(), (count,), generator=gen)
    return idx, img.reshape(-1)[idx]


def build_ref(size, fc_config=None):
    g = ref_gm.Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True,
                         split_fc=fc_config is not None, fc_config=fc_config)
    d = ref_gm.Discriminator(size, channel_multiplier=2)
    g_sd = networks.procedural_fill_(g.state_dict())
    d_sd = networks.procedural_fill_(d.state_dict())
    g.load_state_dict(g_sd)
    d.load_state_dict(d_sd)
    # state_dict() aliases the live parameters: hand out snapshots
    return g, d, {k: v.clone() for k, v in g_sd.items()}, {k: v.clone() for k, v in d_sd.items()}


def golden_networks():
    out = {}
    for size, batch in [(32, 4), (64, 2), (256, 2), (512, 2), (1024, 1)]:
        torch.manual_seed(0)
        g, d, g_sd, d_sd = build_ref(size)
        gen = torch.Generator().manual_seed(1000 + size)
        z = torch.randn(batch, 512, generator=gen)
        noise = seeded_noise(size, batch, 2000 + size)
        with torch.no_grad():
            img_ref, lat_ref = g([z], noise=noise, return_latents=True)
            logit_ref, _ = d(img_ref)
            img_o, lat_o = networks.generator_forward(g_sd, [z], size, noise=noise)
            logit_o = networks.discriminator_forward(d_sd, img_ref)
        e1 = close(img_o, img_ref, 1e-4, f'G({size})')
        e2 = close(logit_o, logit_ref, 1e-4, f'D({size})')
        close(lat_o, lat_ref, 1e-5, f'latent({size})')
        idx, px = sample_pixels(img_ref)
        tag = f's{size}'
        out.update({f'{tag}/z': z, f'{tag}/batch': batch, f'{tag}/noise_seed': 2000 + size,
                    f'{tag}/img_mean': img_ref.mean(), f'{tag}/img_std': img_ref.std(),
                    f'{tag}/img_absmax': img_ref.abs().max(),
                    f'{tag}/thumb': thumb(img_ref), f'{tag}/px_idx': idx, f'{tag}/px_val': px,
                    f'{tag}/logits': logit_ref, f'{tag}/w0': lat_ref[:, 0, :16]})
        if size <= 64:
            out[f'{tag}/img'] = img_ref
        print(f'network {size}: oracle-vs-reference rel err G {e1:.2e} D {e2:.2e}; img std {img_ref.std():.3f}')
    # split_fc mapping network (ffhq.json groups: 128 + 6 x 64)
    from gan_control.utils.mini_batch_multi_split_utils import MiniBatchUtils
    names = ['id', 'expression', 'orientation', 'gamma', 'age', 'hair', 'other']
    bounds = [0, 128, 192, 256, 320, 384, 448, 512]
    groups = {n: {'count_in_mini_bach': [2, 12], 'place_in_mini_batch': [2 * i if i else 0, 2 * i + 2 if i else 2],
                  'place_in_latent': [bounds[i], bounds[i + 1]]} for i, n in enumerate(names)}
    mb = MiniBatchUtils(14, groups, total_batch=14)
    fc = mb.get_fc_config()
    g, d, g_sd, d_sd = build_ref(32, fc_config=fc)
    gen = torch.Generator().manual_seed(31337)
    z = torch.randn(2, 512, generator=gen)
    noise = seeded_noise(32, 2, 4242)
    with torch.no_grad():
        img_ref, _ = g([z], noise=noise)
        fc_groups = [(n, tuple(fc.groups[n]['latent_place'])) for n in fc.in_order_group_names]
        img_o, _ = networks.generator_forward(g_sd, [z], 32, noise=noise, fc_groups=fc_groups)
    close(img_o, img_ref, 1e-4, 'G(32, split_fc)')
    out.update({'split32/z': z, 'split32/noise_seed': 4242, 'split32/img': img_ref,
                'split32/group_names': np.array([n for n, _ in fc_groups]),
                'split32/group_bounds': np.array([b for _, b in fc_groups])})
    # style mixing + truncation at 32
    g, d, g_sd, d_sd = build_ref(32)
    z2 = torch.randn(2, 512, generator=gen)
    with torch.no_grad():
        mean_w = g.style(torch.randn(64, 512, generator=gen)).mean(0, keepdim=True)
        img_ref, _ = g([z, z2], noise=noise, inject_index=3, truncation=0.7, truncation_latent=mean_w)
        img_o, _ = networks.generator_forward(g_sd, [z, z2], 32, noise=noise, inject_index=3, truncation=0.7,
                                              truncation_latent=mean_w)
    close(img_o, img_ref, 1e-4, 'G(32, mixing+truncation)')
    out.update({'mix32/z': z, 'mix32/z2': z2, 'mix32/mean_w': mean_w, 'mix32/noise_seed': 4242, 'mix32/img': img_ref})
    np.savez_compressed(os.path.join(GOLD, 'networks.npz'), **to_np(out))
    print('networks ok')


def golden_noise_modes():
    """StyledConv's noise_mode 'zeros' / 'id_zeros' (gan_model.py:391-399, ModulatedNoiseInjection :1019-1035) at 32 x 32: the image, the
    gradient of sum(image * probe) with respect to a few named parameters, and the parameters that receive no gradient at all."""
    out = {}
    for mode in ('zeros', 'id_zeros'):
        g = ref_gm.Generator(32, 512, 8, channel_multiplier=2, conv_transpose=True, noise_mode=mode)
        g_sd = networks.procedural_fill_(g.state_dict())
        g.load_state_dict(g_sd)
        g_sd = {k: v.clone() for k, v in g_sd.items()}
        gen = torch.Generator().manual_seed(515)
        z = torch.randn(2, 512, generator=gen)
        probe = torch.randn(2, 3, 32, 32, generator=gen)
        noise = seeded_noise(32, 2, 616)
        img_ref, _ = g([z], noise=noise)
        with torch.no_grad():
            img_o, _ = networks.generator_forward(g_sd, [z], 32, noise=noise, noise_mode=mode)
        close(img_o, img_ref, 1e-4, f'G(32, noise_mode={mode})')
        (img_ref * probe).sum().backward()
        none = sorted(n for n, p in g.named_parameters() if p.grad is None)
        names = ['convs.0.noise.weight', 'convs.1.noise.weight', 'convs.0.conv.weight', 'convs.3.activate.bias', 'conv1.conv.modulation.weight']
        names = [n for n in names if n not in none]
        out.update({f'{mode}/z': z, f'{mode}/probe': probe, f'{mode}/noise_seed': 616, f'{mode}/img': img_ref.detach(),
                    f'{mode}/none_grad': np.array(none), f'{mode}/grad_names': np.array(names),
                    f'{mode}/grad_norms': np.array([float(dict(g.named_parameters())[n].grad.norm()) for n in names]),
                    f'{mode}/noise_grads': np.array([float(p.grad) if p.grad is not None else np.nan
                                                     for n, p in g.named_parameters() if n.endswith('noise.weight')])})
    np.savez_compressed(os.path.join(GOLD, 'noise_modes.npz'), **to_np(out))
    print('noise modes ok')


def golden_step(size=32, batch=4, name='step'):
    """One full iteration (i = 0: D step, R1, G step, path-length, EMA) at 32x32, batch 4 by default; main() also
    runs it at the BASELINE resolutions (512x512 batch 4, 1024x1024 batch 2: what fits this container's 64 GiB).

    The reference side is driven with the reference's own modules, helper functions
    (trainers/utils.py) and trainer maths (GeneratorTrainer static methods), following
    generator_trainer.py:301-369, 407-436, 568-599, 645-711 with batch == mini_batch.
    """
    g, d, g_sd, d_sd = build_ref(size)
    g_ema = ref_gm.Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
    g_ema.load_state_dict(g_sd)
    ref_tu.accumulate(g_ema, g, 0)
    gen = torch.Generator().manual_seed(2024)
    real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    z_pl = torch.randn(batch // 2, 512, generator=gen)
    pl_noise = torch.randn(batch // 2, 3, size, size, generator=gen)
    noise_d, noise_g = seeded_noise(size, batch, 11), seeded_noise(size, batch, 12)
    noise_pl = seeded_noise(size, batch // 2, 13)
    cfg = dict(r1=1, g_reg_every=4, d_reg_every=16, lr=0.002, path_regularize=2, g_moving_average=10000)

    # --- dry_run (generator_trainer.py:301-327): which parameters end up with grad None ------------
    fake, latent = g([torch.randn(1, 512, generator=gen)], return_latents=True)
    RT.g_path_regularize(fake, latent, 0)[0].backward()
    none_g = sorted(n for n, p in g.named_parameters() if p.grad is None)
    g.zero_grad()
    t_in = torch.randn(1, 3, size, size, generator=gen).requires_grad_(True)
    pred, _ = d(t_in)
    RT.d_r1_loss(None, pred, t_in).backward()
    none_d = sorted(n for n, p in d.named_parameters() if p.grad is None)
    zero_d = sorted(n for n, p in d.named_parameters() if p.grad is not None and float(p.grad.abs().max()) == 0.0)
    d.zero_grad()

    # The parameter samples stored at the end (2 elements per tensor; same generator, same order as below), drawn up front so that every
    # backward pass can record how large the gradient of each SAMPLED ELEMENT is next to its tensor's RMS gradient.  Adam's first steps
    # are sign-like: a sampled element can only differ between two correct implementations if one of its gradients is within the
    # arithmetic's error of zero -- tests/step_checks.py derives its outlier budget from these ratios instead of a flat percentage.
    gen2 = torch.Generator().manual_seed(8)
    sample_idx = {tag: [(n, torch.randint(0, p.numel(), (2,), generator=gen2).tolist()) for n, p in mod.named_parameters()]
                  for tag, mod in (('g', g), ('d', d), ('g_ema', g_ema))}
    small = {tag: {} for tag in sample_idx}

    def record_ratios(mod, tags):
        grads = {n: p.grad for n, p in mod.named_parameters()}
        for tag in tags:
            for n, idx in sample_idx[tag]:
                gten = grads[n]
                for j in idx:
                    key = f'{n}#{j}'
                    if gten is None:
                        continue                      # no Adam update of this tensor in this pass
                    rms = float(gten.pow(2).mean().sqrt())
                    ratio = abs(float(gten.reshape(-1)[j])) / rms if rms > 0 else 0.0
                    small[tag][key] = min(small[tag].get(key, float('inf')), ratio)

    gr, dr = cfg['g_reg_every'] / (cfg['g_reg_every'] + 1), cfg['d_reg_every'] / (cfg['d_reg_every'] + 1)
    g_optim = torch.optim.Adam(g.parameters(), lr=cfg['lr'] * gr, betas=(0 ** gr, 0.99 ** gr))
    d_optim = torch.optim.Adam(d.parameters(), lr=cfg['lr'] * dr, betas=(0 ** dr, 0.99 ** dr))
    stats = {}
    # D step
    ref_tu.requires_grad(g, False); ref_tu.requires_grad(d, True)
    d.zero_grad()
    fake, _ = g([z_d], noise=noise_d)
    fake_pred, _ = d(fake)
    real_pred, _ = d(real)
    d_loss = RT.d_logistic_loss(real_pred, fake_pred)
    d_loss.div_(len(real))
    d_loss.backward(retain_graph=True)
    stats['d_grad_norm'] = torch.stack([p.grad.norm() for p in d.parameters()]).norm()
    grad_samples = {'d': sample_grads(d)}
    record_ratios(d, ('d',))
    d_optim.step()
    stats['d_loss'] = d_loss.detach()
    stats['real_pred_d'] = real_pred.detach()
    # R1
    d.zero_grad()
    real_r = real.clone().requires_grad_(True)
    real_pred, _ = d(real_r)
    r1 = RT.d_r1_loss(None, real_pred, real_r)
    (cfg['r1'] / 2 * r1 * cfg['d_reg_every'] + 0 * real_pred[0]).backward()
    ref_tu.set_grad_none(d, none_d)
    stats['r1_grad_norm'] = torch.stack([p.grad.norm() for p in d.parameters() if p.grad is not None]).norm()
    grad_samples['r1'] = sample_grads(d)
    record_ratios(d, ('d',))
    d_optim.step()
    stats['d_r1_loss'] = r1.detach()
    # G step
    ref_tu.requires_grad(g, True); ref_tu.requires_grad(d, False)
    g.zero_grad()
    fake, _ = g([z_g], noise=noise_g)
    fake_pred, _ = d(fake)
    g_loss = RT.g_nonsaturating_loss(fake_pred)
    g_loss.backward()
    stats['g_grad_norm'] = torch.stack([p.grad.norm() for p in g.parameters()]).norm()
    grad_samples['g'] = sample_grads(g)
    record_ratios(g, ('g', 'g_ema'))
    g_optim.step()
    stats['g_adv_loss'] = g_loss.detach()
    # path-length regulariser with an injected pl_noise (Generator.g_path_regularize_grad gan_model.py:803-811)
    g.zero_grad()
    fake, latent = g([z_pl], noise=noise_pl, return_latents=True)
    with mock.patch.object(torch, 'randn_like', lambda t: pl_noise):
        grad = ref_gm.Generator.g_path_regularize_grad(fake, latent)
    path_loss, mean_path, lengths = RT.g_path_regularize_grad(grad, 0)
    (cfg['path_regularize'] * cfg['g_reg_every'] * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    ref_tu.set_grad_none(g, none_g)
    stats['pl_grad_norm'] = torch.stack([p.grad.norm() for p in g.parameters() if p.grad is not None]).norm()
    grad_samples['pl'] = sample_grads(g)
    record_ratios(g, ('g', 'g_ema'))
    g_optim.step()
    stats.update(g_path_loss=path_loss.detach(), g_mean_path_length=mean_path, path_lengths=lengths.detach())
    ref_tu.accumulate(g_ema, g, 0.5 ** (batch / cfg['g_moving_average']))

    # --- every backward pass once more IN ISOLATION: from the un-updated procedural weights, no optimiser step in between.
    # The first Adam steps are sign-like (+-lr whatever the gradient's size), so in the sequential iteration above an element
    # whose gradient is ~0 may step either way and every later pass inherits that noise; the isolated passes pin each
    # backward's gradients (per-parameter norms) without it.  'd' needs no re-run: it is the first pass of the iteration.
    del fake, fake_pred, real_pred, g_loss, d_loss, grad, path_loss, lengths, r1, real_r
    gi, di, _, _ = build_ref(size)
    iso = {}
    ref_tu.requires_grad(gi, False); ref_tu.requires_grad(di, True)
    real_r = real.clone().requires_grad_(True)
    real_pred, _ = di(real_r)
    r1 = RT.d_r1_loss(None, real_pred, real_r)
    (cfg['r1'] / 2 * r1 * cfg['d_reg_every'] + 0 * real_pred[0]).backward()
    ref_tu.set_grad_none(di, none_d)
    iso['r1'] = sample_grads(di)
    iso_stats = {'d_r1_loss': r1.detach()}
    del real_r, real_pred, r1
    di.zero_grad()
    ref_tu.requires_grad(gi, True); ref_tu.requires_grad(di, False)
    fake, _ = gi([z_g], noise=noise_g)
    fake_pred, _ = di(fake)
    g_loss = RT.g_nonsaturating_loss(fake_pred)
    g_loss.backward()
    iso['g'] = sample_grads(gi)
    iso_stats['g_adv_loss'] = g_loss.detach()
    del fake, fake_pred, g_loss
    gi.zero_grad()
    fake, latent = gi([z_pl], noise=noise_pl, return_latents=True)
    with mock.patch.object(torch, 'randn_like', lambda t: pl_noise):
        grad = ref_gm.Generator.g_path_regularize_grad(fake, latent)
    path_loss, _, lengths = RT.g_path_regularize_grad(grad, 0)
    (cfg['path_regularize'] * cfg['g_reg_every'] * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    ref_tu.set_grad_none(gi, none_g)
    iso['pl'] = sample_grads(gi)
    iso_stats.update(g_path_loss=path_loss.detach(), path_lengths=lengths.detach())
    del fake, latent, grad, path_loss, lengths, gi, di

    # --- the oracle's own step must reproduce all of it -------------------------------------------
    o = step.OracleStep(g_sd, d_sd, size, batch, none_g=none_g, none_d=none_d)
    o.iteration(0, real, z_d, z_g, z_pl=z_pl, noise_d=noise_d, noise_g=noise_g, noise_pl=noise_pl, pl_noise=pl_noise)
    for k in ['d_loss', 'd_r1_loss', 'g_adv_loss', 'g_path_loss', 'g_mean_path_length']:
        close(torch.tensor(o.stats[k]), stats[k], 2e-4, f'step/{k}')
    close(o.stats['path_lengths'], stats['path_lengths'], 2e-4, 'step/path_lengths')
    worst, moved, total = 0.0, 0, 0
    for params, mod in ((o.g_params, g), (o.d_params, d)):
        for n, p in mod.named_parameters():
            diff = (params[n] - p).abs()
            worst = max(worst, diff.max().item())
            moved += int((diff > 5e-4).sum())
            total += diff.numel()
    print('%s: max |param_oracle - param_ref| after the iteration = %.3e; %d of %d elements differ by more than 5e-4' % (name, worst, moved, total))
    # Adam's first steps are +-lr whatever the gradient's scale: an element whose gradient is at rounding-noise level (R1 on
    # smooth procedural weights) may step the other way.  32x32: none; at the BASELINE sizes a handful of ~60 M elements.
    assert moved <= 1e-5 * total and worst < 1e-2

    out = {'real': real, 'z_d': z_d, 'z_g': z_g, 'z_pl': z_pl, 'pl_noise': pl_noise,
           'noise_seeds': np.array([11, 12, 13]), 'none_g': np.array(none_g), 'none_d': np.array(none_d),
           'zero_d_r1': np.array(zero_d), 'input_seed': np.array(2024)}
    if size > 64:
        # the image-sized inputs (25 MB at 1024x1024) are not stored: tests redraw them from `input_seed` in the order used
        # above (real, z_d, z_g, z_pl, pl_noise; tests/step_checks.py::step_inputs) and check the redraw against the stored latents
        del out['real'], out['pl_noise']
    out.update({f'stat/{k}': v for k, v in stats.items()})
    out['cfg'] = np.array([size, batch])
    for phase, (names, norms, elems) in grad_samples.items():   # per-parameter gradient norms (+ sampled elements) of each of the four backward passes
        out[f'gradnorm/{phase}/names'] = np.array(names)
        out[f'gradnorm/{phase}/vals'] = torch.stack(norms)
        out[f'gradnorm/{phase}/elems'] = torch.stack(elems)
    for phase, (names, norms, elems) in iso.items():             # ... and of the passes run in isolation
        out[f'iso/{phase}/names'] = np.array(names)
        out[f'iso/{phase}/vals'] = torch.stack(norms)
        out[f'iso/{phase}/elems'] = torch.stack(elems)
    out.update({f'iso/stat/{k}': v for k, v in iso_stats.items()})
    for tag, mod in (('g', g), ('d', d), ('g_ema', g_ema)):
        names, vals = [], []
        params = dict(mod.named_parameters())
        for n, idx in sample_idx[tag]:
            for j in idx:
                names.append(f'{n}#{j}')
                vals.append(params[n].detach().reshape(-1)[j])
        out[f'param/{tag}/names'] = np.array(names)
        out[f'param/{tag}/vals'] = torch.stack(vals)
        # min over the passes that stepped the element of |gradient element| / RMS(gradient tensor); inf = never stepped
        out[f'param/{tag}/grad_ratio'] = np.asarray([small[tag].get(k, float('inf')) for k in names], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **to_np(out))
    print(name, 'ok; none_g =', none_g, '; none_d =', none_d, '; zero-grad-under-R1 D params:', len(zero_d))


ELEMS = 64


def elem_index(name, numel):
    """The ELEMS sampled positions of parameter ``name`` (seeded by the name: tests/step_checks.py redraws them)."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(('elem:' + name).encode()) & 0x7FFFFFFF)
    return torch.randint(0, numel, (ELEMS,), generator=gen)


def sample_grads(module):
    """(names, norms, elems): the gradient norm of every parameter that has a gradient -- a per-layer pin of each backward pass -- and ELEMS
    sampled ELEMENTS of it (positions from elem_index): norms alone would pass a permuted, transposed or sign-flipped gradient block."""
    names, norms, elems = [], [], []
    for n, p in module.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(p.grad.detach().norm())
            elems.append(p.grad.detach().reshape(-1)[elem_index(n, p.numel())].clone())
    return names, norms, elems


def golden_step_isolated(size=1024, batch=4, name='step_1024_b4'):
    """The four backward passes of an iteration EACH IN ISOLATION from the un-updated procedural weights, at the workload bench.py times
    (1024 x 1024, 4 images per GPU: one full minibatch-stddev group of 4, gan_model.py:1003-1012; step maths generator_trainer.py:645-719).
    The sequential four-pass iteration of golden_step does not fit this container's 64 GiB at this size; one pass at a time does (the D
    step's two discriminator calls -- fake and real batches are separate calls in the reference too -- are back-propagated one after the
    other into the same .grad, which is what autograd's accumulation does in the joint backward).  Stored per pass: losses / predictions,
    per-parameter gradient norms and ELEMS sampled gradient elements per parameter tensor."""
    import gc
    cfg = dict(r1=1, g_reg_every=4, d_reg_every=16, path_regularize=2)
    gen = torch.Generator().manual_seed(4096)
    real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    z_pl = torch.randn(batch // 2, 512, generator=gen)
    pl_noise = torch.randn(batch // 2, 3, size, size, generator=gen)
    noise_d, noise_g = seeded_noise(size, batch, 21), seeded_noise(size, batch, 22)
    noise_pl = seeded_noise(size, batch // 2, 23)
    g, d, g_sd, d_sd = build_ref(size)
    # None-gradient name sets of the dry run (generator_trainer.py:301-327), at batch 1
    fake, latent = g([torch.randn(1, 512, generator=gen)], return_latents=True)
    RT.g_path_regularize(fake, latent, 0)[0].backward()
    none_g = sorted(n for n, p in g.named_parameters() if p.grad is None)
    g.zero_grad()
    t_in = torch.randn(1, 3, size, size, generator=gen).requires_grad_(True)
    pred, _ = d(t_in)
    RT.d_r1_loss(None, pred, t_in).backward()
    none_d = sorted(n for n, p in d.named_parameters() if p.grad is None)
    d.zero_grad()
    del fake, latent, pred, t_in
    gc.collect()
    out = {'cfg': np.array([size, batch]), 'input_seed': np.array(4096), 'noise_seeds': np.array([21, 22, 23]), 'z_d': z_d, 'z_g': z_g, 'z_pl': z_pl,
           'none_g': np.array(none_g), 'none_d': np.array(none_d)}

    def keep(phase, mod, **stats):
        stats = {k: torch.as_tensor(v) for k, v in stats.items()}
        names, norms, elems = sample_grads(mod)
        out[f'iso/{phase}/names'] = np.array(names)
        out[f'iso/{phase}/vals'] = torch.stack(norms)
        out[f'iso/{phase}/elems'] = torch.stack(elems)
        out.update({f'iso/stat/{k}': v for k, v in stats.items()})
        print(name, phase, 'done:', {k: (float(v) if v.numel() == 1 else tuple(v.shape)) for k, v in stats.items()}, flush=True)

    # D step (generator_trainer.py:645-667): d_logistic_loss / len(real), fake and real halves back-propagated one after the other
    ref_tu.requires_grad(g, False); ref_tu.requires_grad(d, True)
    with torch.no_grad():
        fake, _ = g([z_d], noise=noise_d)
    fake_pred, _ = d(fake)
    (F.softplus(fake_pred).mean() / len(real)).backward()
    fake_pred = fake_pred.detach()
    gc.collect()
    real_pred, _ = d(real)
    (F.softplus(-real_pred).mean() / len(real)).backward()
    real_pred = real_pred.detach()
    d_loss = RT.d_logistic_loss(real_pred, fake_pred) / len(real)
    keep('d', d, d_loss=d_loss, real_pred_d=real_pred, fake_pred_d=fake_pred)
    del fake
    d.zero_grad(); gc.collect()
    # R1 (generator_trainer.py:690-719)
    real_r = real.clone().requires_grad_(True)
    real_pred, _ = d(real_r)
    r1 = RT.d_r1_loss(None, real_pred, real_r)
    (cfg['r1'] / 2 * r1 * cfg['d_reg_every'] + 0 * real_pred[0]).backward()
    ref_tu.set_grad_none(d, none_d)
    keep('r1', d, d_r1_loss=r1.detach())
    del real_r, real_pred, r1
    d.zero_grad(); gc.collect()
    # G step (generator_trainer.py:407-436, vanilla)
    ref_tu.requires_grad(g, True); ref_tu.requires_grad(d, False)
    fake, _ = g([z_g], noise=noise_g)
    fake_pred, _ = d(fake)
    g_loss = RT.g_nonsaturating_loss(fake_pred)
    g_loss.backward()
    keep('g', g, g_adv_loss=g_loss.detach(), fake_pred_g=fake_pred.detach())
    del fake, fake_pred, g_loss
    g.zero_grad(); gc.collect()
    # path length (generator_trainer.py:563-599, gan_model.py:803-811) with an injected pl_noise
    fake, latent = g([z_pl], noise=noise_pl, return_latents=True)
    with mock.patch.object(torch, 'randn_like', lambda t: pl_noise):
        grad = ref_gm.Generator.g_path_regularize_grad(fake, latent)
    path_loss, mean_path, lengths = RT.g_path_regularize_grad(grad, 0)
    (cfg['path_regularize'] * cfg['g_reg_every'] * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    ref_tu.set_grad_none(g, none_g)
    keep('pl', g, g_path_loss=path_loss.detach(), path_lengths=lengths.detach(), g_mean_path_length=mean_path)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **to_np(out))
    print(name, 'ok')


def _chunked_passes(g, d, size, batch, inputs, none_g, none_d, cfg, sub):
    """The four backward passes of golden_step_isolated with every network call cut into sub-batches of ``sub`` images, each chunk's
    gradient accumulated into ``.grad`` -- for batches whose joint graph does not fit the build container (512 x 512 x 16 images carries
    twice the activations of 1024 x 1024 x 4).  Exact by construction, not an approximation:

    * the generator has no cross-sample operation (gan_model.py:709-801), and the discriminator's only one is the minibatch standard deviation,
      whose ``view(group, -1, ...)`` (gan_model.py:1003-1012) puts samples {m, m + B/4, m + 2B/4, m + 3B/4} into one group: chunk m = ``[m::B/4]``
      is one complete group, so D(chunk) equals the rows of D(batch);
    * the loss over the whole batch is the reference's own function (d_logistic_loss / g_nonsaturating_loss / g_path_regularize_grad,
      generator_trainer.py:563-566, 618-624, 683-688) applied to the re-assembled network outputs as a leaf: its backward gives the cotangent of every
      output element (incl. the path-length mean's own gradient, :622-623), which each chunk's recomputed graph is then back-propagated with --
      the chain rule, cut at the network output.  R1 (:712-719) is a mean of per-sample penalties and is accumulated directly.

    golden_step_isolated_chunked asserts chunked == joint in float64 at a small size before using it.
    Yields (phase, module, stats) after each pass; the caller records and zeroes the gradients."""
    real, z_d, z_g, z_pl, pl_noise, noise_d, noise_g, noise_pl = inputs
    n_sub = batch // sub
    groups = [torch.arange(m, batch, n_sub) for m in range(n_sub)]          # complete minibatch-stddev groups
    pick = lambda maps, idx: [t[idx] for t in maps]
    # D step (generator_trainer.py:645-667)
    ref_tu.requires_grad(g, False); ref_tu.requires_grad(d, True)
    with torch.no_grad():
        fake = torch.cat([g([z_d[i:i + sub]], noise=[t[i:i + sub] for t in noise_d])[0] for i in range(0, batch, sub)])
        fake_pred = torch.empty(batch, 1, dtype=real.dtype)
        real_pred = torch.empty(batch, 1, dtype=real.dtype)
        for idx in groups:
            fake_pred[idx] = d(fake[idx])[0]
            real_pred[idx] = d(real[idx])[0]
    fp, rp = fake_pred.clone().requires_grad_(True), real_pred.clone().requires_grad_(True)
    d_loss = RT.d_logistic_loss(rp, fp) / len(real)
    d_loss.backward()
    for idx in groups:
        d(fake[idx])[0].backward(fp.grad[idx])
        d(real[idx])[0].backward(rp.grad[idx])
    yield 'd', d, dict(d_loss=d_loss.detach(), real_pred_d=real_pred, fake_pred_d=fake_pred)
    del fake
    # R1 (generator_trainer.py:690-719): the mean over the batch of per-sample penalties
    r1_total = 0.0
    for idx in groups:
        real_r = real[idx].clone().requires_grad_(True)
        pred, _ = d(real_r)
        r1 = RT.d_r1_loss(None, pred, real_r) * (len(idx) / batch)
        (cfg['r1'] / 2 * r1 * cfg['d_reg_every'] + 0 * pred[0]).backward()
        r1_total = r1_total + r1.detach()
        del real_r, pred, r1
    ref_tu.set_grad_none(d, none_d)
    yield 'r1', d, dict(d_r1_loss=r1_total)
    # G step (generator_trainer.py:407-436, vanilla)
    ref_tu.requires_grad(g, True); ref_tu.requires_grad(d, False)
    fake_pred = torch.empty(batch, 1, dtype=real.dtype)
    with torch.no_grad():
        for idx in groups:
            fake_pred[idx] = d(g([z_g[idx]], noise=pick(noise_g, idx))[0])[0]
    fp = fake_pred.clone().requires_grad_(True)
    g_loss = RT.g_nonsaturating_loss(fp)
    g_loss.backward()
    for idx in groups:
        d(g([z_g[idx]], noise=pick(noise_g, idx))[0])[0].backward(fp.grad[idx])
    yield 'g', g, dict(g_adv_loss=g_loss.detach(), fake_pred_g=fake_pred)
    # path length (generator_trainer.py:563-599, gan_model.py:803-811): two sweeps -- the per-sample latent gradients first, then the
    # reference's penalty on all of them as a leaf, then each chunk's second-order graph back-propagated with its slice of the cotangent
    pb = batch // 2
    spans = [slice(i, min(i + sub, pb)) for i in range(0, pb, sub)]

    def latent_grad(sl, create_graph):
        fake, latent = g([z_pl[sl]], noise=[t[sl] for t in noise_pl], return_latents=True)
        with mock.patch.object(torch, 'randn_like', lambda t: pl_noise[sl]):
            grad = ref_gm.Generator.g_path_regularize_grad(fake, latent)
        return (grad, fake) if create_graph else grad.detach()

    grads = torch.cat([latent_grad(sl, False) for sl in spans]).requires_grad_(True)
    path_loss, mean_path, lengths = RT.g_path_regularize_grad(grads, 0)
    weight = cfg['path_regularize'] * cfg['g_reg_every']
    (weight * path_loss).backward()                 # reaches the leaf ``grads`` only
    for sl in spans:
        grad, fake = latent_grad(sl, True)
        autograd.backward([grad, 0 * fake[0, 0, 0, 0]], [grads.grad[sl], None])
    ref_tu.set_grad_none(g, none_g)
    yield 'pl', g, dict(g_path_loss=path_loss.detach(), path_lengths=lengths.detach(), g_mean_path_length=mean_path)


def _joint_passes(g, d, size, batch, inputs, none_g, none_d, cfg):
    """The same four passes on the whole batch at once (the body of golden_step_isolated, as a generator): the yardstick of the chunked form."""
    real, z_d, z_g, z_pl, pl_noise, noise_d, noise_g, noise_pl = inputs
    ref_tu.requires_grad(g, False); ref_tu.requires_grad(d, True)
    with torch.no_grad():
        fake, _ = g([z_d], noise=noise_d)
    fake_pred, _ = d(fake)
    real_pred, _ = d(real)
    d_loss = RT.d_logistic_loss(real_pred, fake_pred) / len(real)
    d_loss.backward()
    yield 'd', d, dict(d_loss=d_loss.detach(), real_pred_d=real_pred.detach(), fake_pred_d=fake_pred.detach())
    real_r = real.clone().requires_grad_(True)
    real_pred, _ = d(real_r)
    r1 = RT.d_r1_loss(None, real_pred, real_r)
    (cfg['r1'] / 2 * r1 * cfg['d_reg_every'] + 0 * real_pred[0]).backward()
    ref_tu.set_grad_none(d, none_d)
    yield 'r1', d, dict(d_r1_loss=r1.detach())
    ref_tu.requires_grad(g, True); ref_tu.requires_grad(d, False)
    fake, _ = g([z_g], noise=noise_g)
    fake_pred, _ = d(fake)
    g_loss = RT.g_nonsaturating_loss(fake_pred)
    g_loss.backward()
    yield 'g', g, dict(g_adv_loss=g_loss.detach(), fake_pred_g=fake_pred.detach())
    fake, latent = g([z_pl], noise=noise_pl, return_latents=True)
    with mock.patch.object(torch, 'randn_like', lambda t: pl_noise):
        grad = ref_gm.Generator.g_path_regularize_grad(fake, latent)
    path_loss, mean_path, lengths = RT.g_path_regularize_grad(grad, 0)
    (cfg['path_regularize'] * cfg['g_reg_every'] * path_loss + 0 * fake[0, 0, 0, 0]).backward()
    ref_tu.set_grad_none(g, none_g)
    yield 'pl', g, dict(g_path_loss=path_loss.detach(), path_lengths=lengths.detach(), g_mean_path_length=mean_path)


def _isolated_inputs(size, batch, seed, noise_seeds, dtype=torch.float32):
    """The draw order tests/step_checks.py::check_isolated repeats."""
    gen = torch.Generator().manual_seed(seed)
    real = torch.rand(batch, 3, size, size, generator=gen) * 2 - 1
    z_d, z_g = torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen)
    z_pl = torch.randn(batch // 2, 512, generator=gen)
    pl_noise = torch.randn(batch // 2, 3, size, size, generator=gen)
    maps = [seeded_noise(size, b, s) for b, s in zip((batch, batch, batch // 2), noise_seeds)]
    cast = lambda t: t.to(dtype)
    return gen, [cast(t) for t in (real, z_d, z_g, z_pl, pl_noise)] + [[cast(t) for t in m] for m in maps]


def _none_grad_sets(g, d, size, gen, dtype=torch.float32):
    """None-gradient name sets of the reference's dry run (generator_trainer.py:301-327), at batch 1."""
    fake, latent = g([torch.randn(1, 512, generator=gen).to(dtype)], return_latents=True)
    RT.g_path_regularize(fake, latent, 0)[0].backward()
    none_g = sorted(n for n, p in g.named_parameters() if p.grad is None)
    g.zero_grad()
    t_in = torch.randn(1, 3, size, size, generator=gen).to(dtype).requires_grad_(True)
    pred, _ = d(t_in)
    RT.d_r1_loss(None, pred, t_in).backward()
    none_d = sorted(n for n, p in d.named_parameters() if p.grad is None)
    d.zero_grad()
    return none_g, none_d


def golden_step_isolated_chunked(size=512, batch=16, name='step_512_b16', sub=4, check_size=32):
    """BASELINE config 2 at its own batch (configs/ffhq.json:11,22,23: 512 x 512, batch 16; same key layout as golden_step_isolated, so
    tests/step_checks.py::check_isolated reads it unchanged).  The joint graph of 16 images does not fit this container, so the passes run
    as _chunked_passes; that procedure is first held against the joint passes of the same reference modules in FLOAT64 at ``check_size``
    (every parameter gradient, every statistic: 1e-9), then in float32 (bounded by summation order), and only then used at ``size``."""
    import gc
    cfg = dict(r1=1, g_reg_every=4, d_reg_every=16, path_regularize=2)
    seeds = (21, 22, 23)
    for dtype, bound in ((torch.float64, 1e-9), (torch.float32, 2e-4)):     # float32: summation order only
        g, d, _, _ = build_ref(check_size)
        g, d = g.to(dtype), d.to(dtype)
        gen, inputs = _isolated_inputs(check_size, batch, 4096, seeds, dtype)
        none_g, none_d = _none_grad_sets(g, d, check_size, gen, dtype)
        worst = 0.0
        for (ph, mod, st), (ph2, mod2, st2) in zip(_joint_passes(g, d, check_size, batch, inputs, none_g, none_d, cfg),
                                                   _snapshot(_chunked_passes, g, d, check_size, batch, inputs, none_g, none_d, cfg, sub)):
            assert ph == ph2
            joint = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
            assert sorted(joint) == sorted(mod2), (ph, set(joint) ^ set(mod2))
            total = float(torch.stack([v.double().pow(2).sum() for v in joint.values()]).sum().sqrt())
            scal = [(float(v), float(mod2[n])) for n, v in joint.items() if v.numel() == 1]
            for n, v in joint.items():
                if v.numel() == 1 and dtype == torch.float32:
                    continue        # a noise strength: ONE scalar = a random-sign sum over batch x channels x pixels; compared together below
                err = float((v - mod2[n]).double().norm()) / max(float(v.double().norm()), 1e-3 * total)
                worst = max(worst, err)
                assert err <= bound, (name, str(dtype), ph, n, err)
            if scal and dtype == torch.float32:
                a, b = torch.tensor(scal, dtype=torch.float64).unbind(1)
                err = float((a - b).norm() / a.norm().clamp_min(1e-3 * total))
                print(name, ph, 'noise strengths as one vector, float32 chunked vs joint: %.2e' % err, flush=True)
                assert err <= 20 * bound, (name, ph, 'scalar parameters', err)
            for k, v in st.items():
                a, b = torch.as_tensor(v).double().reshape(-1), torch.as_tensor(st2[k]).double().reshape(-1)
                err = float((a - b).abs().max() / a.abs().max().clamp_min(1e-6))
                assert err <= bound, (name, str(dtype), ph, k, err)
            mod.zero_grad()
        print(name, 'chunked == joint at %d x %d, batch %d, %s: worst per-parameter gradient difference %.2e' % (check_size, check_size, batch, dtype, worst), flush=True)
        del g, d
    gen, inputs = _isolated_inputs(size, batch, 4096, seeds)
    g, d, _, _ = build_ref(size)
    none_g, none_d = _none_grad_sets(g, d, size, gen)
    gc.collect()
    out = {'cfg': np.array([size, batch]), 'input_seed': np.array(4096), 'noise_seeds': np.array(seeds), 'z_d': inputs[1], 'z_g': inputs[2], 'z_pl': inputs[3],
           'none_g': np.array(none_g), 'none_d': np.array(none_d), 'chunk': np.array(sub)}
    for phase, mod, stats in _chunked_passes(g, d, size, batch, inputs, none_g, none_d, cfg, sub):
        stats = {k: torch.as_tensor(v) for k, v in stats.items()}
        names, norms, elems = sample_grads(mod)
        out[f'iso/{phase}/names'] = np.array(names)
        out[f'iso/{phase}/vals'] = torch.stack(norms)
        out[f'iso/{phase}/elems'] = torch.stack(elems)
        out.update({f'iso/stat/{k}': v for k, v in stats.items()})
        print(name, phase, 'done:', {k: (float(v) if v.numel() == 1 else tuple(v.shape)) for k, v in stats.items()}, flush=True)
        mod.zero_grad(); gc.collect()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **to_np(out))
    print(name, 'ok')


def _snapshot(passes, g, d, *args):
    """Run a pass generator on DEEP COPIES of the networks and yield (phase, {name: gradient}, stats), zeroing in between, so that the joint and
    the chunked passes can be walked side by side."""
    import copy
    g2, d2 = copy.deepcopy(g), copy.deepcopy(d)
    for phase, mod, stats in passes(g2, d2, *args):
        grads = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
        yield phase, grads, stats
        mod.zero_grad()


def golden_augment():
    """ADA transforms: sampled matrices under a fixed torch seed (pins the RNG call order) and the image-space
    result + input gradient of the reference's augment() for those matrices (non_leaking.py:394-398)."""
    from oracle import augment as oaug
    out = {}
    for tag, (b, h, w, p_aug, seed) in {'a': (3, 48, 40, 0.8, 5), 'b': (2, 64, 64, 1.0, 6)}.items():
        # the reference loops forever when a GIVEN G needs more reflect padding than the image has
        # (non_leaking.py:288-313): advance the seed until the padding fits
        while True:
            torch.manual_seed(seed)
            G = REF_NL.sample_affine(p_aug, b, h, w)
            C = REF_NL.sample_color(p_aug, b)
            pads = REF_NL.get_padding(torch.inverse(G), h, w)
            if max(pads[0], pads[1]) + 6 < w and max(pads[2], pads[3]) + 6 < h:
                break
            seed += 1
        gen = torch.Generator().manual_seed(seed + 100)
        img = torch.randn(b, 3, h, w, generator=gen)
        x = img.clone().requires_grad_(True)
        ref, _ = REF_NL.augment(x, p_aug, (G, C))
        go = torch.randn(ref.shape, generator=gen)
        gi, = autograd.grad(ref, x, go)
        close(oaug.augment(img, G, C, REF_NL.SYM6), ref, 1e-5, f'augment/{tag}')
        out.update({f'{tag}/cfg': np.array([b, h, w, seed]), f'{tag}/p': p_aug, f'{tag}/G': G, f'{tag}/C': C, f'{tag}/img': img,
                    f'{tag}/out': ref, f'{tag}/go': go, f'{tag}/gi': gi})
    np.savez_compressed(os.path.join(GOLD, 'augment.npz'), **to_np(out))
    print('augment ok')


def golden_fid():
    """Frechet distance fixtures from the reference's calc_fid (fid_utils/fid.py:43-66) and its batch plan (:22-27)."""
    stubs = {n: mock.MagicMock() for n in ['gan_control.fid_utils.calc_inception', 'tqdm'] if n not in sys.modules}
    with mock.patch.dict(sys.modules, stubs):
        import importlib
        ref_fid = importlib.import_module('gan_control.fid_utils.fid')
    from oracle import fid as ofid
    rng = np.random.default_rng(7)
    out = {}
    for i, (dim, n1, n2, shift) in enumerate([(16, 200, 300, 0.0), (48, 500, 400, 0.3), (64, 90, 2000, 1.0), (32, 1000, 1000, 0.05)]):
        mix1, mix2 = rng.normal(size=(dim, dim)), rng.normal(size=(dim, dim))
        f1 = rng.normal(size=(n1, dim)) @ mix1
        f2 = rng.normal(size=(n2, dim)) @ mix2 + shift
        m1, c1, m2, c2 = f1.mean(0), np.cov(f1, rowvar=False), f2.mean(0), np.cov(f2, rowvar=False)
        ref = float(ref_fid.calc_fid(m1, c1, m2, c2))
        ora = ofid.frechet_distance(m1, c1, m2, c2)
        assert abs(ref - ora) <= 1e-6 * max(1.0, abs(ref)), (i, ref, ora)
        out.update({f'case{i}/m1': m1, f'case{i}/c1': c1, f'case{i}/m2': m2, f'case{i}/c2': c2, f'case{i}/fid': np.float64(ref)})
    # identical statistics: distance 0 up to the square-root round-off of the reference implementation
    out['same/fid'] = np.float64(ref_fid.calc_fid(m1, c1, m1, c1))
    for n, b in [(50000, 20), (101, 20), (7, 8), (40, 20)]:
        n_batch = n // b
        resid = n - n_batch * b
        plan = [b] * n_batch + ([resid] if resid else [])
        assert plan == ofid.batch_plan(n, b)
        out[f'plan/{n}_{b}'] = np.asarray(plan, dtype=np.int64)
    np.savez_compressed(os.path.join(GOLD, 'fid.npz'), **out)
    print('fid ok')


def golden_controller():
    """FcStack forward / gradients and three optimisation steps from the reference's own classes."""
    import gan_control.models.controller_model as ref_cm
    from oracle import controller as octl
    torch.manual_seed(11)
    lr_mlp, n_mlp, in_dim, mid, out_dim, chunk = 0.01, 4, 3, 32, 24, (40, 64)
    ref = ref_cm.FcStack(lr_mlp, n_mlp, in_dim, mid, out_dim).double()
    for prm in ref.parameters():
        prm.data.add_(torch.randn_like(prm) * 0.3)
    controls = torch.randn(6, in_dim, dtype=torch.float64)
    w_latent = torch.randn(6, 96, dtype=torch.float64)
    out = {'controls': controls, 'w_latent': w_latent, 'chunk': np.asarray(chunk), 'hyper': np.asarray([lr_mlp, n_mlp, in_dim, mid, out_dim])}
    for k, v in ref.state_dict().items():
        out['init/' + k] = v.clone()
    y = ref(controls)
    ws = [ref.fc_stack[i].weight.detach().clone() for i in range(n_mlp)]
    bs = [ref.fc_stack[i].bias.detach().clone() for i in range(n_mlp)]
    close(octl.fc_stack_forward(controls, ws, bs, lr_mlp), y, 1e-12, 'controller forward')
    out['forward'] = y.detach()
    # the reference's step: L1 on the group slice, Adam(lr * ratio, betas = (0 ** ratio, 0.99 ** ratio)), ratio = 4 / 5
    ratio = 4 / 5
    opt = torch.optim.Adam(ref.parameters(), lr=0.002 * ratio, betas=(0 ** ratio, 0.99 ** ratio))
    rec = torch.nn.L1Loss()
    losses = []
    for _ in range(3):
        ref.zero_grad()
        loss = rec(ref(controls), w_latent[:, chunk[0]:chunk[1]])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    leaf_w = [w.requires_grad_(True) for w in ws]
    leaf_b = [b.requires_grad_(True) for b in bs]
    ora = octl.controller_step(leaf_w, leaf_b, lr_mlp, controls, w_latent, chunk, steps=3)
    assert np.allclose(ora, losses, rtol=1e-12), (ora, losses)
    for i in range(n_mlp):
        close(leaf_w[i], ref.fc_stack[i].weight, 1e-10, 'controller weight %d after 3 steps' % i)
    out['losses'] = np.asarray(losses)
    for k, v in ref.state_dict().items():
        out['after3/' + k] = v.clone()
    np.savez_compressed(os.path.join(GOLD, 'controller.npz'), **to_np(out))
    print('controller ok')


def controller_procedural_fill_(state_dict):
    """FcStack weights from CRC32(key) seeds (weights are stored / lr_mul, as EqualLinear initialises them): the batch-128 fixture pins
    a 0.9 M-parameter controller without committing its weights.  The test applies the same fill (it is re-stated there)."""
    import zlib
    for key in sorted(state_dict):
        gen = torch.Generator().manual_seed(zlib.crc32(('controller/' + key).encode()) & 0x7FFFFFFF)
        v = torch.randn(state_dict[key].shape, generator=gen, dtype=torch.float32)
        v = v * 100.0 if key.endswith('.weight') else v * 0.1
        with torch.no_grad():
            state_dict[key].copy_(v.to(state_dict[key].dtype))
    return state_dict


def golden_controller_afhq():
    """BASELINE config 5's controller leg at its own size: configs/controller_configs/afhq/default_w_latent_controller.json
    (in_dim 3 -> 512 -> 512 -> 512 -> the orientation group's 192 w dimensions, batch 128, L1 latent reconstruction; `latent_adv_` and
    `attribute_rec_` carry a trailing underscore in the shipped file and are therefore off, controller_trainer.py:210-213).  Three steps
    of the reference's own FcStack under the reference's optimiser set-up; hot-path config fields go to configs.json['afhq_controller']."""
    import json
    import gan_control.models.controller_model as ref_cm
    from gan_control.utils.mini_batch_multi_split_utils import MiniBatchUtils
    from oracle import controller as octl
    cfg = json.load(open('/root/reference/src/gan_control/configs/controller_configs/afhq/default_w_latent_controller.json'))
    gan = json.load(open('/root/reference/src/gan_control/configs/afhq.json'))
    mc, tc, gtc = cfg['model_config'], cfg['training_config'], gan['training_config']
    group = gtc[mc['loss']]['same_group_name']                                   # controller_trainer.py:97-100
    mb = MiniBatchUtils(gtc['mini_batch'], gtc['sub_groups_dict'], total_batch=gtc['batch'])
    chunk = [int(v) for v in mb.place_in_latent_dict[group]]
    fields = {'model_config': {k: mc[k] for k in ('latent_size', 'size', 'lr_mlp', 'n_mlp', 'in_dim', 'mid_dim', 'loss')},
              'training_config': {k: tc[k] for k in ('rec_loss', 'batch', 'reg_every', 'lr', 'losses', 'attribute_rec_w', 'controller_type', 'generate_controls')},
              'working_group': group, 'group_chunk': chunk}
    path = os.path.join(GOLD, 'configs.json')
    allcfg = json.load(open(path))
    allcfg['afhq_controller'] = fields
    with open(path, 'w') as f:
        json.dump(allcfg, f, indent=1, sort_keys=True)
    batch, out_dim = tc['batch'], chunk[1] - chunk[0]
    ref = ref_cm.FcStack(mc['lr_mlp'], mc['n_mlp'], mc['in_dim'], mc['mid_dim'], out_dim).double()
    ref.load_state_dict(controller_procedural_fill_(ref.state_dict()))
    gen = torch.Generator().manual_seed(128)
    controls = (torch.rand(batch, mc['in_dim'], generator=gen) * 2 - 1).double()          # orientation controls are angles scaled to [-1, 1]
    w_latent = torch.randn(batch, mc['latent_size'], generator=gen).double()
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    y = ref(controls)
    n = mc['n_mlp']
    ws = [ref.fc_stack[i].weight.detach().clone() for i in range(n)]
    bs = [ref.fc_stack[i].bias.detach().clone() for i in range(n)]
    close(octl.fc_stack_forward(controls, ws, bs, mc['lr_mlp']), y, 1e-12, 'afhq controller forward')
    ratio = tc['reg_every'] / (tc['reg_every'] + 1)
    opt = torch.optim.Adam(ref.parameters(), lr=tc['lr'] * ratio, betas=(0 ** ratio, 0.99 ** ratio))
    rec = torch.nn.L1Loss() if tc['rec_loss'] == 'l1' else torch.nn.MSELoss()
    losses, grad_norms = [], None
    for it in range(3):
        ref.zero_grad()
        loss = rec(ref(controls), w_latent[:, chunk[0]:chunk[1]])
        loss.backward()
        if it == 0:
            grad_norms = {k: float(p.grad.norm()) for k, p in ref.named_parameters()}
        opt.step()
        losses.append(float(loss.detach()))
    leaf_w = [w.requires_grad_(True) for w in ws]
    leaf_b = [b.requires_grad_(True) for b in bs]
    ora = octl.controller_step(leaf_w, leaf_b, mc['lr_mlp'], controls, w_latent, chunk, lr=tc['lr'], reg_every=tc['reg_every'], steps=3, loss=tc['rec_loss'])
    assert np.allclose(ora, losses, rtol=1e-12), (ora, losses)
    for i in range(n):
        close(leaf_w[i], ref.fc_stack[i].weight, 1e-10, 'afhq controller weight %d after 3 steps' % i)
    out = {'input_seed': np.asarray([128]), 'cfg': np.asarray([batch, mc['in_dim'], mc['mid_dim'], n, out_dim, chunk[0], chunk[1]]),
           'controls': controls.float(), 'w_head': w_latent[:2].float(), 'forward': y.detach().float(), 'losses': np.asarray(losses)}
    names, idx, vals, moved = [], [], [], []
    pick = torch.Generator().manual_seed(5)
    for k, v in ref.state_dict().items():
        flat = v.reshape(-1)
        sel = torch.randperm(flat.numel(), generator=pick)[:512]
        names.append(k)
        idx.append(sel.numpy())
        vals.append(flat[sel].numpy())
        moved.append(float((v - init[k]).norm()))
    out['after3/names'] = np.asarray(names)
    out['after3/moved_norm'] = np.asarray(moved)
    out['grad0/norms'] = np.asarray([grad_norms[k] for k in names])
    for k, i_, v_ in zip(names, idx, vals):
        out['after3/idx/' + k], out['after3/val/' + k] = i_, v_
    np.savez_compressed(os.path.join(GOLD, 'controller_afhq.npz'), **to_np(out))
    print('afhq controller ok: losses', losses)


def golden_losses():
    """Same / not-same hinge losses and the mini-batch re-arrangement, from the reference's own LossModelClass (no_model=True:
    the loss on given features, loss_model.py:18-38, 121-199) and MiniBatchUtils (mini_batch_multi_split_utils.py:55-87)."""
    import json
    from oracle import losses as olosses
    from gan_control.utils.mini_batch_multi_split_utils import MiniBatchUtils
    stubs = {n: mock.MagicMock() for n in ['gan_control.evaluation.orientation'] if n not in sys.modules}
    with mock.patch.dict(sys.modules, stubs):
        from gan_control.losses.loss_model import LossModelClass
    cfg = json.load(open('/root/reference/src/gan_control/configs/ffhq.json'))['training_config']
    groups, mini = cfg['sub_groups_dict'], cfg['mini_batch']
    mb = MiniBatchUtils(mini, groups, total_batch=cfg['batch'])
    gen = torch.Generator().manual_seed(404)
    out = {}
    # re_arrange_z: one style code and two (mixing)
    for tag, count in (('z1', 1), ('z2', 2)):
        z = [torch.randn(mini, 512, generator=gen) for _ in range(count)]
        ref = mb.re_arrange_z([t.clone() for t in z], 0)
        ora = olosses.re_arrange_z(z, groups)
        for a, b in zip(ora, ref):
            assert torch.equal(a, b), 're_arrange_z'
        for i in range(count):
            out[f'{tag}/in{i}'], out[f'{tag}/out{i}'] = z[i], ref[i]
    # the hinge itself: three criteria, intermediate levels switched on for one of them
    cases = {'embedding_loss': [(mini, 8, 6, 6), (mini, 4, 3, 3), (mini, 24)], 'expression_loss': [(mini, 6, 5, 5), (mini, 3, 10)], 'age_loss': [(mini, 101)]}
    for name, shapes in cases.items():
        lc = dict(cfg[name])
        lc['intermediate_layers_weights'] = [0.5, 0.0, 0.25, 1.0][:len(shapes) - 1]
        lc['lower_thres'], lc['upper_thres'] = list(lc['lower_thres'])[:len(shapes) - 1], list(lc['upper_thres'])[:len(shapes) - 1]
        lc['focus_on_list'] = (['not_same_as_last_layer', 'same_as_last_layer', 'not_same_as_last_layer'][:len(shapes) - 1]) + ['same_as_last_layer']
        scale = {'embedding_loss': 0.25, 'expression_loss': 1.5, 'age_loss': 2.0}[name]
        feats = [(torch.randn(*sh, generator=gen) * scale).requires_grad_(True) for sh in shapes]
        loss_class = LossModelClass(lc, loss_name=name, mini_batch_size=mini, no_model=True)
        same, other = mb.extract_same_not_same_from_list(feats, lc['same_group_name'])
        ref = loss_class.calc_mini_batch_loss(last_layer_same_features=same, last_layer_not_same_features=other)
        grads = autograd.grad(ref, feats, allow_unused=True)
        so, oo = olosses.split_same_not_same([f.detach() for f in feats], groups, lc['same_group_name'], mini)
        close(olosses.hinge_pair_loss(so, oo, lc, name), ref, 1e-6, f'hinge/{name}')
        out[f'{name}/cfg'] = np.array(json.dumps(lc))
        out[f'{name}/loss'] = ref.detach()
        for i, (f, g) in enumerate(zip(feats, grads)):
            out[f'{name}/f{i}'] = f.detach()
            out[f'{name}/g{i}'] = torch.zeros_like(f) if g is None else g
    # the age predictor head and the controller criterion of the phase-2 attribute_rec objective (deep_age_criterion.py:24-40)
    from gan_control.losses.deep_expectation_age.deep_age_criterion import DeepAgeCriterion
    crit = DeepAgeCriterion()
    logits = torch.randn(5, 101, generator=gen) * 3
    target = torch.rand(5, generator=gen) * 80
    out['age_head/logits'], out['age_head/target'] = logits, target
    out['age_head/pred'] = crit.predict(logits)
    out['age_head/criterion'] = crit.controller_criterion(crit.predict(logits), target)
    out['sub_groups'] = np.array(json.dumps(groups))
    out['mini_batch'] = np.array(mini)
    np.savez_compressed(os.path.join(GOLD, 'losses.npz'), **to_np(out))
    print('losses ok')


def golden_inception():
    """The FID feature network from the reference's own classes (fid_utils/inception.py + overwrite_inception.py) on procedurally
    generated weights.  torchvision is absent here: ``torchvision.models.inception`` is stood in for by the reference's own copy of that
    file (overwrite_inception.py defines the same Inception{A..E}), the weight download is replaced by the procedural state dict, and the
    undefined global ``path`` of fid_inception_v3 (inception.py:181-182, a NameError as shipped) is supplied."""
    import importlib, types
    ow = importlib.import_module('gan_control.fid_utils.overwrite_inception') if 'torchvision' in sys.modules else None
    tv = types.ModuleType('torchvision'); tvm = types.ModuleType('torchvision.models'); tvu = types.ModuleType('torchvision.models.utils')
    with mock.patch.dict(sys.modules, {'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.utils': tvu}):
        tvu.load_state_dict_from_url = lambda *a, **k: None
        tv.models = tvm
        ow = importlib.import_module('gan_control.fid_utils.overwrite_inception')
        tvm.inception = ow
        tvm.inception_v3 = ow.inception_v3
        ref_inc = importlib.import_module('gan_control.fid_utils.inception')
    from oracle import inception as oinc
    holder = {}

    def fake_download(*a, **k):
        # the state dict fid_inception_v3 loads into the un-wrapped Inception3: procedural values under ITS key names
        model = ow.inception_v3(num_classes=1008, aux_logits=False, pretrained=False)
        model.Mixed_5b = ref_inc.FIDInceptionA(192, pool_features=32)
        model.Mixed_5c = ref_inc.FIDInceptionA(256, pool_features=64)
        model.Mixed_5d = ref_inc.FIDInceptionA(288, pool_features=64)
        model.Mixed_6b = ref_inc.FIDInceptionC(768, channels_7x7=128)
        model.Mixed_6c = ref_inc.FIDInceptionC(768, channels_7x7=160)
        model.Mixed_6d = ref_inc.FIDInceptionC(768, channels_7x7=160)
        model.Mixed_6e = ref_inc.FIDInceptionC(768, channels_7x7=192)
        model.Mixed_7b = ref_inc.FIDInceptionE_1(1280)
        model.Mixed_7c = ref_inc.FIDInceptionE_2(2048)
        holder['raw'] = oinc.procedural_inception_fill_(model.state_dict())
        return holder['raw']

    ref_inc.load_state_dict_from_url = fake_download
    ref_inc.path = 'unused'
    torch.manual_seed(5)
    net = ref_inc.InceptionV3(output_blocks=[0, 1, 2, 3]).eval()
    net.load_state_dict(oinc.procedural_inception_fill_(net.state_dict()))      # keyed on the WRAPPER's names (blocks.<b>.<i>...), as the tests fill the product
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    out = {}
    gen = torch.Generator().manual_seed(21)
    for tag, shape in (('a', (2, 3, 64, 48)), ('b', (1, 3, 299, 299))):
        x = torch.rand(shape, generator=gen)
        with torch.no_grad():
            ref = net(x)
            ora = oinc.inception_features(sd, x, output_blocks=(0, 1, 2, 3))
        for i, (r, o) in enumerate(zip(ref, ora)):
            close(o, r, 1e-5, f'inception/{tag}/block{i}')
        out[f'{tag}/shape'] = np.asarray(shape)          # the input is redrawn in the tests: torch.rand(shape, generator seeded 21), cases in order a, b
        out[f'{tag}/x_check'] = x.flatten()[:64].clone()
        out[f'{tag}/pool3'] = ref[3].reshape(shape[0], -1)
        # the lower blocks are large: a strided sample and the per-channel means pin them
        for i in range(3):
            out[f'{tag}/block{i}_mean'] = ref[i].mean((2, 3))
            out[f'{tag}/block{i}_sample'] = ref[i][:, ::7, ::5, ::3].contiguous()
    # the wrapper's key names against the un-wrapped checkpoint's (what load_fid_weights of the product maps)
    wrapped = sorted(k for k in sd if not k.endswith('num_batches_tracked'))
    out['n_keys'] = np.asarray([len(wrapped)])
    with torch.no_grad():
        x = torch.rand(2, 3, 64, 48, generator=torch.Generator().manual_seed(33))
        no_resize = ref_inc.InceptionV3(output_blocks=[3], resize_input=False, normalize_input=False).eval()
        no_resize.load_state_dict(oinc.procedural_inception_fill_(no_resize.state_dict()))
        out['c/x'] = x
        x_big = torch.nn.functional.interpolate(x, size=(96, 80), mode='bilinear', align_corners=False)
        out['c/pool3'] = no_resize(x_big)[0].reshape(2, -1)
        close(oinc.inception_features({k: v for k, v in no_resize.state_dict().items()}, x_big, (3,), False, False)[0], no_resize(x_big)[0], 1e-5, 'inception/c')
    np.savez_compressed(os.path.join(GOLD, 'inception.npz'), **to_np(out))
    print('inception ok')


def golden_configs():
    """The hot-path fields of the three shipped training configurations (configs/ffhq.json:5-84, metfaces.json, afhq.json:
    numbers and switches, no code) plus what the reference's MiniBatchUtils.get_fc_config (mini_batch_multi_split_utils.py:
    103-115) derives from each ``sub_groups_dict``: the fixture the product's config ingestion is checked against."""
    import json
    from gan_control.utils.mini_batch_multi_split_utils import MiniBatchUtils
    keep = ['parallel_grad_regularize_step', 'iter', 'start_iter', 'batch', 'mini_batch', 'mini_batch_mode', 'transfer_learning_model',
            'augment', 'sub_groups_dict', 'r1', 'd_every', 'g_reg_every', 'd_reg_every', 'lr_g', 'lr_d', 'g_moving_average',
            'path_regularize', 'path_batch_shrink', 'mixing', 'parallel']
    out = {}
    for name in ('ffhq', 'metfaces', 'afhq'):
        cfg = json.load(open(f'/root/reference/src/gan_control/configs/{name}.json'))
        tc = cfg['training_config']
        mb = MiniBatchUtils(tc['mini_batch'], tc['sub_groups_dict'], total_batch=tc['batch'])
        fc = mb.get_fc_config()
        out[name] = {'model_config': cfg['model_config'], 'training_config': {k: tc[k] for k in keep},
                     'fc_config': {'in_order_group_names': list(fc.in_order_group_names),
                                   'groups': {n: {'latent_place': list(fc.groups[n]['latent_place']), 'latent_size': int(fc.groups[n]['latent_size'])}
                                              for n in fc.in_order_group_names}}}
    path = os.path.join(GOLD, 'configs.json')
    if os.path.exists(path):            # keep what other jobs add (golden_controller_afhq: 'afhq_controller')
        for k, v in json.load(open(path)).items():
            out.setdefault(k, v)
    with open(path, 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('configs ok')


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(int(os.environ.get('GOLDEN_THREADS', '8')))
    jobs = {'upfirdn2d': golden_upfirdn2d, 'bias_act': golden_bias_act, 'convs': golden_convs, 'misc': golden_misc,
            'networks': golden_networks, 'noise_modes': golden_noise_modes, 'step': golden_step,
            # the BASELINE resolutions: ~15 min and ~40 GiB on 8 cores (1024x1024 at batch 4 does not fit this container's 64 GiB)
            'step_512': lambda: golden_step(512, 4, 'step_512'), 'step_1024': lambda: golden_step(1024, 2, 'step_1024'),
            # the bench workload itself (1024 x 1024, 4 images), one backward pass at a time: ~25 min, < 60 GiB
            'step_1024_b4': golden_step_isolated,
            # BASELINE config 2 at its own batch (512 x 512, 16 images), chunked by minibatch-stddev group: ~40 min, < 30 GiB
            'step_512_b16': golden_step_isolated_chunked,
            'augment': golden_augment, 'fid': golden_fid, 'controller': golden_controller, 'configs': golden_configs, 'controller_afhq': golden_controller_afhq, 'losses': golden_losses, 'inception': golden_inception}
    for name in (sys.argv[1:] or list(jobs)):
        jobs[name]()
    total = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print('fixtures written to %s (%.1f KiB)' % (GOLD, total / 1024))


if __name__ == '__main__':
    main()
