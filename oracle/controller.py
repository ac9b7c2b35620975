"""ORACLE (test infrastructure only): the controller MLP and its training step restated with plain ATen ops.

Reference: models/controller_model.py:13-53 (FcStack of EqualLinear + fused leaky-ReLU, gan_model.py:171-202 / :25-41) and
trainers/controller_trainer.py:202-229 (L1 between FcStack(controls) and the group's w slice, Adam with the lazy-reg ratio).
Pinned by tests/golden/controller.npz: outputs of the reference's own FcStack and of torch.optim.Adam driven as the
reference drives it (oracle/make_golden.py::golden_controller).
"""
import math

import torch
import torch.nn.functional as F


def fc_stack_forward(x, weights, biases, lr_mul):
    for w, b in zip(weights, biases):
        scale = (1 / math.sqrt(w.shape[1])) * lr_mul
        x = F.leaky_relu(F.linear(x, w * scale) + b * lr_mul, 0.2) * math.sqrt(2)
    return x


def controller_step(weights, biases, lr_mul, controls, w_latent, chunk, lr=0.002, reg_every=4, steps=1, loss='l1'):
    """`steps` Adam updates in place on leaf tensors; returns the list of loss values."""
    ratio = reg_every / (reg_every + 1)
    params = list(weights) + list(biases)
    opt = torch.optim.Adam(params, lr=lr * ratio, betas=(0 ** ratio, 0.99 ** ratio))
    out = []
    for _ in range(steps):
        opt.zero_grad()
        pred = fc_stack_forward(controls, weights, biases, lr_mul)
        target = w_latent[:, chunk[0]:chunk[1]]
        val = (pred - target).abs().mean() if loss == 'l1' else (pred - target).pow(2).mean()
        val.backward()
        opt.step()
        out.append(float(val.detach()))
    return out
