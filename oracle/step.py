"""Step-level oracle: losses, lazy regularisers and one full G+D iteration on the CPU.

Restates generator_trainer.py's vanilla step (no predictor losses) as plain functions
over the functional networks of oracle/networks.py.  Also serves as the
``cpu_baseline`` ("port") leg of bench.py.  Test infrastructure only.
"""
import math

import torch
import torch.nn.functional as F
from torch import autograd

from . import networks


def d_logistic_loss(real_pred, fake_pred):
    """Reference: generator_trainer.py:690-695."""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred):
    """Reference: generator_trainer.py:563-566."""
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """Reference: generator_trainer.py:713-719."""
    grad, = autograd.grad(real_pred.sum(), real_img, create_graph=True)
    return grad.pow(2).reshape(grad.shape[0], -1).sum(1).mean()


def path_lengths_and_penalty(grad, mean_path_length, decay=0.01):
    """Reference: g_path_regularize_grad generator_trainer.py:617-624."""
    lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (lengths.mean() - mean_path_length)
    penalty = (lengths - path_mean).pow(2).mean()
    return penalty, path_mean.detach(), lengths


def path_grad(fake_img, latent, pl_noise=None):
    """Reference: Generator.g_path_regularize_grad gan_model.py:803-811."""
    if pl_noise is None:
        pl_noise = torch.randn_like(fake_img)
    pl_noise = pl_noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    grad, = autograd.grad((fake_img * pl_noise).sum(), latent, create_graph=True)
    return grad


def adam_hparams(lr, reg_every):
    """Reference: generator_trainer.py:161-173 (lazy-regularisation rescaling)."""
    c = reg_every / (reg_every + 1)
    return dict(lr=lr * c, betas=(0 ** c, 0.99 ** c))


class OracleStep:
    """One-process CPU restatement of discriminator_update + generator_update.

    Reference: generator_trainer.py:329-369 (loop), :626-711 (D), :407-436 and :568-599 (G),
    trainers/utils.py:8-12 (EMA), with batch == mini_batch and vanilla semantics.
    Explicit ``noise`` / ``pl_noise`` arguments make every step reproducible.
    """

    def __init__(self, g_sd, d_sd, size, batch, lr_g=0.002, lr_d=0.002, r1=1.0, path_regularize=2.0,
                 g_reg_every=4, d_reg_every=16, path_batch_shrink=2, g_moving_average=10000,
                 none_g=None, none_d=None):
        self.size, self.batch = size, batch
        self.g = {k: (v.clone().requires_grad_(True) if not self._is_buffer(k) else v.clone()) for k, v in g_sd.items()}
        self.d = {k: (v.clone().requires_grad_(True) if not self._is_buffer(k) else v.clone()) for k, v in d_sd.items()}
        self.g_ema = {k: v.detach().clone() for k, v in self.g.items()}
        self.g_params = {k: v for k, v in self.g.items() if v.requires_grad}
        self.d_params = {k: v for k, v in self.d.items() if v.requires_grad}
        self.g_optim = torch.optim.Adam(list(self.g_params.values()), **adam_hparams(lr_g, g_reg_every))
        self.d_optim = torch.optim.Adam(list(self.d_params.values()), **adam_hparams(lr_d, d_reg_every))
        self.r1, self.path_regularize = r1, path_regularize
        self.g_reg_every, self.d_reg_every = g_reg_every, d_reg_every
        self.path_batch_shrink = path_batch_shrink
        self.accum = 0.5 ** (batch / g_moving_average)
        self.mean_path_length = 0
        # name sets the reference's dry_run discovers (generator_trainer.py:301-327)
        self.none_g = set(none_g) if none_g is not None else {k for k in self.g_params if k.startswith('to_rgb') and k.endswith('.bias') and 'conv' not in k}
        self.none_d = set(none_d) if none_d is not None else {'final_linear.1.bias'}
        self.stats = {}

    @staticmethod
    def _is_buffer(key):
        return key.endswith('.kernel') or key.startswith('noises.')

    @staticmethod
    def _set_req(params, flag):
        for p in params.values():
            p.requires_grad_(flag)

    @staticmethod
    def _zero(params):
        for p in params.values():
            p.grad = None

    def G(self, z, noise=None):
        return networks.generator_forward(self.g, [z], self.size, noise=noise)

    def D(self, img):
        return networks.discriminator_forward(self.d, img)

    def d_step(self, real, z, noise=None):
        self._set_req(self.g_params, False); self._set_req(self.d_params, True)
        self._zero(self.d_params)
        fake, _ = self.G(z, noise)
        loss = d_logistic_loss(self.D(real), self.D(fake)) / real.shape[0]   # generator_trainer.py:658
        loss.backward()
        self.d_optim.step()
        self.stats['d_loss'] = float(loss)

    def d_reg(self, real):
        self._set_req(self.g_params, False); self._set_req(self.d_params, True)
        self._zero(self.d_params)
        real = real.detach().requires_grad_(True)
        pred = self.D(real)
        r1 = d_r1_loss(pred, real)
        (self.r1 / 2 * r1 * self.d_reg_every + 0 * pred[0]).backward()          # generator_trainer.py:706
        for k in self.none_d:
            self.d_params[k].grad = None
        self.d_optim.step()
        self.stats['d_r1_loss'] = float(r1)

    def g_step(self, z, noise=None):
        self._set_req(self.g_params, True); self._set_req(self.d_params, False)
        self._zero(self.g_params)
        fake, _ = self.G(z, noise)
        loss = g_nonsaturating_loss(self.D(fake))
        loss.backward()
        self.g_optim.step()
        self.stats['g_adv_loss'] = float(loss)

    def g_reg(self, z, noise=None, pl_noise=None):
        self._set_req(self.g_params, True); self._set_req(self.d_params, False)
        self._zero(self.g_params)
        fake, latent = self.G(z, noise)
        grad = path_grad(fake, latent, pl_noise)
        penalty, self.mean_path_length, lengths = path_lengths_and_penalty(grad, self.mean_path_length)
        (self.path_regularize * self.g_reg_every * penalty + 0 * fake[0, 0, 0, 0]).backward()
        for k in self.none_g:
            self.g_params[k].grad = None
        self.g_optim.step()
        self.stats.update(g_path_loss=float(penalty), g_path_length=float(lengths.mean()),
                          g_mean_path_length=float(self.mean_path_length))
        self.stats['path_lengths'] = lengths.detach().clone()

    def ema(self):
        with torch.no_grad():
            for k, v in self.g_params.items():
                self.g_ema[k].mul_(self.accum).add_(v.detach(), alpha=1 - self.accum)

    def iteration(self, i, real, z_d, z_g, z_pl=None, noise_d=None, noise_g=None, noise_pl=None, pl_noise=None):
        """Order per iteration: D step -> [R1] -> G step -> [path] -> EMA (generator_trainer.py:351-369)."""
        self.d_step(real, z_d, noise_d)
        if i % self.d_reg_every == 0:
            self.d_reg(real)
        self.g_step(z_g, noise_g)
        if i % self.g_reg_every == 0:
            pb = max(1, self.batch // self.path_batch_shrink)
            if z_pl is None:
                z_pl = torch.randn(pb, z_g.shape[1])
            self.g_reg(z_pl[:pb], noise_pl, pl_noise)
        self.ema()
