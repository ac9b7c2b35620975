"""Does the step TRAIN a GAN?  (VERDICT r4 "Next" 3; the reachable evidence for the FID half of BASELINE.json's metric.)

All parity evidence of this repository is one or two iterations long, and trajectories decorrelate within ~9 iterations (sign-like Adam,
beta1 = 0), so element-wise multi-iteration parity is impossible; what CAN be held is a statistical end-to-end statement: trained on a
multi-modal image distribution, the generator's samples move towards held-out real images, the game does not collapse, and the split-bf16
arithmetic ends where exact fp32 ends, within the seed-to-seed spread.  The loop is the reference's (generator_trainer.py:329-355:
discriminator_update / generator_update with lazy R1 every 16 and path length every 4 iterations, Adam, EMA); the distance is the
reference's calc_fid (fid_utils/fid.py:43-66) between feature statistics -- of the package's HIP InceptionV3 with PROCEDURAL weights (the
pretrained FID checkpoint is an external download), pool3 features projected onto PROJ_DIM fixed random directions so that the covariance
of a few thousand samples has full rank.  A fixed random-feature Frechet distance, not an FID.

No files: the image set is procedural (``modes``: seeded mixtures of oriented colour gratings with a blob on top, random phase / position /
contrast per image).  This module is test infrastructure (it imports oracle/ for the procedural feature-network fill and for the CPU
oracle's own training run); ``tools/train_sanity.py`` runs it and writes ``profiles/train_sanity_r05.json``;
``tests/test_train_sanity.py`` holds a short version under ``-m gpu``.
"""
import math
import time

import numpy as np
import torch

PROJ_DIM = 128
N_MODES = 8


def procedural_images(n, size, seed, device='cpu'):
    """[n, 3, size, size] in [-1, 1]: image i belongs to mode i % N_MODES (orientation, spatial frequency and colour pair of a grating fixed
    per mode), with its own phase, contrast, blob position / radius / colour and a little pixel noise."""
    g = torch.Generator().manual_seed(seed)
    mg = torch.Generator().manual_seed(1234)                       # the modes themselves are the same for every seed
    theta = torch.rand(N_MODES, generator=mg) * math.pi
    freq = 1.5 + 3.5 * torch.rand(N_MODES, generator=mg)            # cycles per image
    col_a = torch.rand(N_MODES, 3, generator=mg) * 2 - 1
    col_b = torch.rand(N_MODES, 3, generator=mg) * 2 - 1
    mode = torch.arange(n) % N_MODES
    phase = torch.rand(n, generator=g) * 2 * math.pi
    contrast = 0.6 + 0.4 * torch.rand(n, generator=g)
    centre = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    radius = 0.08 + 0.12 * torch.rand(n, generator=g)
    blob_col = torch.rand(n, 3, generator=g) * 2 - 1
    ys, xs = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing='ij')
    t = theta[mode]
    u = xs[None] * torch.cos(t)[:, None, None] + ys[None] * torch.sin(t)[:, None, None]
    wave = 0.5 + 0.5 * torch.sin(2 * math.pi * freq[mode][:, None, None] * u + phase[:, None, None])          # [n, H, W] in [0, 1]
    wave = 0.5 + (wave - 0.5) * contrast[:, None, None]
    img = col_a[mode][:, :, None, None] * wave[:, None] + col_b[mode][:, :, None, None] * (1 - wave[:, None])
    d2 = (xs[None] - centre[:, 0, None, None]) ** 2 + (ys[None] - centre[:, 1, None, None]) ** 2
    blob = torch.exp(-d2 / (2 * radius[:, None, None] ** 2))[:, None]
    img = img * (1 - blob) + blob_col[:, :, None, None] * blob
    img = img + 0.02 * torch.randn(img.shape, generator=g)
    return img.clamp_(-1, 1).to(device)


class FeatureSpace:
    """HIP InceptionV3 (pool3, 299 x 299 resize as in evaluation/tracker.py:322-329) with procedural weights + a fixed random projection."""

    def __init__(self, device):
        from gan_control_amd.fid_utils.inception import InceptionV3
        from oracle.inception import procedural_inception_fill_
        net = InceptionV3(output_blocks=[3], normalize_input=False)
        net.load_state_dict(procedural_inception_fill_(net.state_dict()))
        self.net = net.to(device).eval()
        self.device = device
        q, _ = torch.linalg.qr(torch.randn(2048, PROJ_DIM, generator=torch.Generator().manual_seed(99)))
        self.proj = q.to(device)

    @torch.no_grad()
    def stats(self, images, batch=100):
        from gan_control_amd.fid_utils.fid import feature_statistics
        feats = []
        for i in range(0, images.shape[0], batch):
            f = self.net(images[i:i + batch].to(self.device))[0].reshape(-1, 2048)
            feats.append((f @ self.proj).cpu())
        return feature_statistics(torch.cat(feats).double().numpy())

    def distance(self, images, real_stats):
        from gan_control_amd.fid_utils.fid import calc_fid
        m, c = self.stats(images)
        return float(calc_fid(m, c, real_stats[0], real_stats[1]))


def pixel_features(images):
    """[n, 96]: 4 x 4 average-pooled RGB (48) and 4 x 4 average-pooled gradient magnitude per channel (48) -- colour layout and amount of
    edge energy per region: what a GAN learns first, and a space in which an untrained generator (near-constant images) is far from any
    image set whatever its seed."""
    import torch.nn.functional as F
    x = images.float()
    gx = (x[..., :, 1:] - x[..., :, :-1]).abs()
    gy = (x[..., 1:, :] - x[..., :-1, :]).abs()
    g = F.pad(gx, (0, 1)) + F.pad(gy, (0, 0, 0, 1))
    return torch.cat([F.adaptive_avg_pool2d(x, 4).flatten(1), F.adaptive_avg_pool2d(g, 4).flatten(1)], 1)


def pixel_stats(images):
    from gan_control_amd.fid_utils.fid import feature_statistics
    return feature_statistics(pixel_features(images).double().cpu().numpy())


def pixel_distance(images, real_stats):
    from gan_control_amd.fid_utils.fid import calc_fid
    m, c = pixel_stats(images)
    return float(calc_fid(m, c, real_stats[0], real_stats[1]))


def windowed(values, n_windows=6):
    """Means over n_windows equal stretches of a curve."""
    v = np.asarray(values, dtype=np.float64)
    edges = np.linspace(0, len(v), n_windows + 1).astype(int)
    return [float(v[a:b].mean()) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _distances(space, real, gen_fn, n_eval, seed=4242, device=None):
    """{'inception': ..., 'pixel': ...} of n_eval samples of gen_fn(z) against the held-out statistics ``real``."""
    dev = device if device is not None else space.device
    zg = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        imgs = torch.cat([gen_fn(torch.randn(100, 512, device=dev, generator=zg)).clamp(-1, 1) for _ in range(n_eval // 100)])
    return {'inception': space.distance(imgs, real['inception']), 'pixel': pixel_distance(imgs, real['pixel'])}


def _curves(curve):
    """{iteration: {'ema': {...}, 'raw': {...}}} -> {'ema/pixel': {iteration: value}, ...} rounded for the report."""
    out = {}
    for it, d in curve.items():
        for net, m in d.items():
            for k, v in m.items():
                out.setdefault('%s/%s' % (net, k), {})[str(it)] = round(v, 4)
    return out


def run_hip(space, real, train_set, size, batch, iters, seed, mode, n_eval, eval_at=()):
    """One training run of the product trainer (bench.py's construction: every default knob) on shuffled batches of ``train_set``.  Distances
    are taken for the EMA generator (what the reference evaluates, tracker.py:322-341; it starts as a copy of the initial weights and lags
    ~600 iterations behind at this batch) and for the raw generator."""
    import random
    from gan_control_amd.models.op import _backend
    from gan_control_amd.trainers.generator_trainer import GeneratorTrainer, default_config
    if mode is not None:
        _backend.get().conv_mode = mode
    random.seed(seed)
    torch.manual_seed(seed)
    tr = GeneratorTrainer(default_config(size, batch), device=space.device, seed=seed)
    order = torch.Generator().manual_seed(1000 + seed)

    def both():
        return {'ema': _distances(space, real, lambda z: tr.g_ema([z])[0], n_eval), 'raw': _distances(space, real, lambda z: tr.generator([z])[0], n_eval)}

    curve = {0: both()}
    d_logistic, g_adv, r1, pl = [], [], [], []
    perm, at = torch.randperm(train_set.shape[0], generator=order), 0
    t0 = time.time()
    for i in range(iters):
        if at + batch > perm.numel():
            perm, at = torch.randperm(train_set.shape[0], generator=order), 0
        real_batch = train_set[perm[at:at + batch].to(train_set.device)]
        at += batch
        tr.train_iteration(i, real_batch)
        if i % 4 == 0:
            s = tr.stats
            d_logistic.append(float(s['d_loss']) * batch)           # the logged value is divided by the images of the mini-batch (:658)
            g_adv.append(float(s['g_adv_loss']))
            r1.append(float(s.get('d_r1_loss', 0.0)))
            pl.append(float(s.get('g_mean_path_length', 0.0)))
        if (i + 1) in eval_at:
            curve[i + 1] = both()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    seconds = time.time() - t0
    curve[iters] = both()
    finite = all(bool(torch.isfinite(p).all()) for p in tr.generator.parameters())
    c = _curves(curve)
    return {'mode': mode, 'seed': seed, 'iterations': iters, 'seconds': round(seconds, 1), 'distance': c,
            'fall': {k: round(v['0'] / min(list(v.values())[1:]), 2) for k, v in c.items()},
            'fall_at_end': {k: round(v['0'] / v[str(iters)], 2) for k, v in c.items()},
            'd_logistic_windows': [round(v, 4) for v in windowed(d_logistic)],
            'g_adv_windows': [round(v, 4) for v in windowed(g_adv)], 'r1_windows': [round(v, 5) for v in windowed(r1)],
            'mean_path_length_windows': [round(v, 4) for v in windowed(pl)], 'd_logistic_min_max': [round(min(d_logistic), 4), round(max(d_logistic), 4)],
            'first_iterations': {'d_logistic': [round(v, 4) for v in d_logistic[:64]], 'g_adv': [round(v, 4) for v in g_adv[:64]]},
            'finite': finite}


def run_oracle(space, real, train_set, size, batch, seconds_budget, seed, n_eval, threads=4):
    """The CPU oracle's own loop (oracle/step.py: the restatement of the reference step) on the same data for as long as the budget allows:
    its loss curve over those iterations is what the HIP runs' first iterations are compared with (seed-to-seed spread of the HIP runs as the
    yardstick).  ``threads``: these are tiny tensors; more host threads make the loop slower."""
    from gan_control_amd.models.gan_model import Generator, Discriminator
    from oracle import networks
    from oracle.step import OracleStep
    keep = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(seed)
        g = Generator(size, 512, 8, channel_multiplier=2, conv_transpose=True)
        d = Discriminator(size, channel_multiplier=2)
        o = OracleStep(g.state_dict(), d.state_dict(), size, batch)
        gen = torch.Generator().manual_seed(seed)
        data = train_set.cpu()

        def both():
            return {'ema': _distances(space, real, lambda z: networks.generator_forward(o.g_ema, [z], size)[0].to(space.device), n_eval, device='cpu'),
                    'raw': _distances(space, real, lambda z: networks.generator_forward(o.g, [z], size)[0].to(space.device), n_eval, device='cpu')}

        curve = {0: both()}
        d_logistic, g_adv = [], []
        t0, i = time.time(), 0
        while time.time() - t0 < seconds_budget:
            idx = torch.randint(0, data.shape[0], (batch,), generator=gen)
            o.iteration(i, data[idx], torch.randn(batch, 512, generator=gen), torch.randn(batch, 512, generator=gen))
            if i % 4 == 0:
                d_logistic.append(o.stats['d_loss'] * batch)
                g_adv.append(o.stats['g_adv_loss'])
            i += 1
        curve[i] = both()
    finally:
        torch.set_num_threads(keep)
    return {'mode': 'cpu oracle (oracle/step.py)', 'seed': seed, 'iterations': i, 'seconds': round(time.time() - t0, 1), 'threads': threads,
            'distance': _curves(curve), 'd_logistic': [round(v, 4) for v in d_logistic], 'g_adv': [round(v, 4) for v in g_adv]}


def main(size=32, batch=16, iters=2500, seeds=(0, 1, 2), modes=('f32', 'bf16x3'), n_train=4096, n_eval=2000, oracle_seconds=240, out=None, device='cuda:0'):
    import json
    space = FeatureSpace(device)
    train_set = procedural_images(n_train, size, seed=7, device=device)
    held_out = procedural_images(n_eval, size, seed=8, device=device)
    real = {'inception': space.stats(held_out), 'pixel': pixel_stats(held_out)}
    floor = {'inception': space.distance(train_set[:n_eval], real['inception']), 'pixel': pixel_distance(train_set[:n_eval], real['pixel'])}   # real vs real
    report = {'workload': 'StyleGAN2 G+D training loop at %dx%d, batch %d, %d iterations per run, procedural %d-mode image set (%d training / %d held-out images)'
                          % (size, size, batch, iters, N_MODES, n_train, n_eval),
              'distance': 'Frechet distance (fid_utils.fid.calc_fid) between %d generated and the held-out images in two fixed feature spaces: "inception" = HIP '
                          'InceptionV3 pool3 features with procedural weights projected on %d fixed random directions; "pixel" = 4x4 pooled RGB + 4x4 pooled '
                          'gradient magnitude (96 numbers); for the EMA generator (what the reference evaluates) and the raw one' % (n_eval, PROJ_DIM),
              'real_vs_real': {k: round(v, 5) for k, v in floor.items()}, 'runs': []}
    every = max(1, iters // 5)
    for mode in modes:
        for seed in seeds:
            r = run_hip(space, real, train_set, size, batch, iters, seed, mode, n_eval, eval_at=tuple(range(every, iters, every)))
            print(json.dumps(r), flush=True)
            report['runs'].append(r)
    if oracle_seconds:
        r = run_oracle(space, real, train_set, size, batch, oracle_seconds, 0, min(n_eval, 500))
        print(json.dumps(r), flush=True)
        report['oracle_run'] = r
        # the oracle's loss curve over its iterations against the HIP runs' over the same stretch (every 4th iteration is sampled on both sides)
        n = len(r['d_logistic'])
        if n >= 4:
            hip = np.array([[np.mean(x['first_iterations'][k][:n]) for x in report['runs']] for k in ('d_logistic', 'g_adv')])
            report['oracle_vs_hip_first_iterations'] = {
                'iterations': r['iterations'], 'oracle': {'d_logistic': float(np.mean(r['d_logistic'])), 'g_adv': float(np.mean(r['g_adv']))},
                'hip_mean': {'d_logistic': float(hip[0].mean()), 'g_adv': float(hip[1].mean())},
                'hip_seed_sigma': {'d_logistic': float(hip[0].std(ddof=1)) if hip.shape[1] > 1 else None, 'g_adv': float(hip[1].std(ddof=1)) if hip.shape[1] > 1 else None}}
    key = 'raw/pixel'
    finals = {m: [r['distance'][key][str(r['iterations'])] for r in report['runs'] if r['mode'] == m] for m in modes}
    report['final_distance'] = {'metric': key, **{m: {'values': v, 'mean': float(np.mean(v)), 'std': float(np.std(v, ddof=1)) if len(v) > 1 else None} for m, v in finals.items()}}
    if all(len(finals.get(m, ())) > 1 for m in ('f32', 'bf16x3')):
        a, b = np.array(finals['f32']), np.array(finals['bf16x3'])
        sigma = float(np.std(a, ddof=1))
        report['bf16x3_vs_f32'] = {'mean_difference': float(abs(a.mean() - b.mean())), 'f32_seed_sigma': sigma, 'within_2_sigma': bool(abs(a.mean() - b.mean()) < 2 * sigma)}
    report['criteria'] = {'raw_pixel_distance_falls_5x_every_run': all(r['fall_at_end'][key] >= 5 for r in report['runs']),
                          'd_logistic_windows_in_0.1_2': all(0.1 < w < 2 for r in report['runs'] for w in r['d_logistic_windows']),
                          'all_finite': all(r['finite'] for r in report['runs'])}
    if out:
        with open(out, 'w') as f:
            json.dump(report, f, indent=1)
    return report
