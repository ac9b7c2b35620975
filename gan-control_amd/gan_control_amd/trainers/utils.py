"""Small trainer helpers with the reference's names and semantics (src/gan_control/trainers/utils.py)."""
import random

import torch


def accumulate(model1, model2, decay=0.999):
    """EMA over named parameters only -- buffers are not averaged (utils.py:8-12)."""
    p1 = [p.data for _, p in sorted(model1.named_parameters())]
    p2 = [p.data for _, p in sorted(model2.named_parameters())]
    torch._foreach_mul_(p1, decay)
    torch._foreach_add_(p1, p2, alpha=1 - decay)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def make_noise(batch, latent_dim, n_noise, device, generator=None):
    if n_noise == 1:
        return torch.randn(batch, latent_dim, device=device, generator=generator)
    return torch.randn(n_noise, batch, latent_dim, device=device, generator=generator).unbind(0)


def mixing_noise(batch, latent_dim, prob, device, generator=None):
    """Style-mixing latents with probability ``prob`` (utils.py:19-23)."""
    if prob > 0 and random.random() < prob:
        return make_noise(batch, latent_dim, 2, device, generator)
    return [make_noise(batch, latent_dim, 1, device, generator)]


def make_mini_batch_from_noise(noise, batch, mini_batch):
    """[n_noise][batch, D] -> [n_chunks][n_noise][mini_batch, D] (utils.py:33-42)."""
    chunks = [n.chunk(batch // mini_batch) for n in noise]
    return [[c[i] for c in chunks] for i in range(len(chunks[0]))]


def set_grad_none(model, targets):
    for n, p in model.named_parameters():
        if n in targets:
            p.grad = None
