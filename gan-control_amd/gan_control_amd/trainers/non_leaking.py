"""Adaptive discriminator augmentation (ADA): random affine + colour transforms of image batches.

Mirrors the call surface of the reference's src/gan_control/trainers/non_leaking.py
(``augment(img, p, transform_matrix=(None, None))`` :394-398, ``random_apply_affine`` :316-371,
``random_apply_color`` :385-391, ``sample_affine`` :151-207, ``sample_color`` :210-241), which the trainer
calls when ``training_config.augment.enabled`` (generator_trainer.py:421-424, 651-653).  The reference file
imports ``upfirdn2d`` from a package it does not ship (non_leaking.py:6); here the two 12x12 sym6
anti-aliasing passes (x2 up-sampling before, x2 down-sampling after the warp) run on the HIP upfirdn2d
kernel (generic path), with first- and second-order gradients.

Transform matrices are sampled on the host with the SAME order of random draws as the reference, so a
given torch seed yields the same augmentation (pinned by tests/golden/augment.npz).  The bilinear warp
itself is still ATen's grid_sample (plumbing for this round).
"""
import math

import torch
from torch.nn import functional as F

from ..models.op import upfirdn2d

# Daubechies least-asymmetric ("symlet") 6 low-pass filter, 12 taps -- the ADA paper's anti-aliasing kernel
SYM6 = (
    0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
    0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
    -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148,
)


# ---- homogeneous 2-D / 3-D matrices, batched on dim 0 ------------------------------------------------------
def _eye(n, dim):
    return torch.eye(dim).unsqueeze(0).repeat(n, 1, 1)


def translate_mat(t_x, t_y):
    m = _eye(t_x.shape[0], 3)
    m[:, 0, 2], m[:, 1, 2] = t_x, t_y
    return m


def rotate_mat(theta):
    m = _eye(theta.shape[0], 3)
    c, s = torch.cos(theta), torch.sin(theta)
    m[:, 0, 0], m[:, 0, 1], m[:, 1, 0], m[:, 1, 1] = c, -s, s, c
    return m


def scale_mat(s_x, s_y):
    m = _eye(s_x.shape[0], 3)
    m[:, 0, 0], m[:, 1, 1] = s_x, s_y
    return m


def translate3d_mat(t_x, t_y, t_z):
    m = _eye(t_x.shape[0], 4)
    m[:, 0, 3], m[:, 1, 3], m[:, 2, 3] = t_x, t_y, t_z
    return m


def scale3d_mat(s_x, s_y, s_z):
    m = _eye(s_x.shape[0], 4)
    m[:, 0, 0], m[:, 1, 1], m[:, 2, 2] = s_x, s_y, s_z
    return m


def rotate3d_mat(axis, theta):
    """Rodrigues rotation about ``axis`` by ``theta`` (batched)."""
    ux, uy, uz = (float(a) for a in axis)
    u = torch.tensor((ux, uy, uz), dtype=torch.float32)
    cross = torch.tensor([[0.0, -uz, uy], [uz, 0.0, -ux], [-uy, ux, 0.0]], dtype=torch.float32)
    c, s = torch.cos(theta).view(-1, 1, 1), torch.sin(theta).view(-1, 1, 1)
    rot = c * torch.eye(3) + s * cross + (1 - c) * torch.outer(u, u)
    m = _eye(theta.shape[0], 4)
    m[:, :3, :3] = rot
    return m


def luma_flip_mat(axis, i):
    v = torch.tensor(tuple(axis) + (0,), dtype=torch.float32)
    return _eye(i.shape[0], 4) - 2 * torch.outer(v, v) * i.view(-1, 1, 1)


def saturation_mat(axis, i):
    v = torch.tensor(tuple(axis) + (0,), dtype=torch.float32)
    proj = torch.outer(v, v)
    return proj + (_eye(i.shape[0], 4) - proj) * i.view(-1, 1, 1)


# ---- samplers (one torch RNG call each, in the reference's order) ------------------------------------------------
def lognormal_sample(size, mean=0, std=1):
    return torch.empty(size).log_normal_(mean=mean, std=std)


def category_sample(size, categories):
    return torch.tensor(categories)[torch.randint(high=len(categories), size=(size,))]


def uniform_sample(size, low, high):
    return torch.empty(size).uniform_(low, high)


def normal_sample(size, mean=0, std=1):
    return torch.empty(size).normal_(mean, std)


def bernoulli_sample(size, p):
    return torch.empty(size).bernoulli_(p)


def random_mat_apply(p, transform, prev, eye):
    """With probability p per sample, left-multiply ``prev`` by ``transform``."""
    keep = bernoulli_sample(transform.shape[0], p).view(-1, 1, 1)
    return (keep * transform + (1 - keep) * eye) @ prev


def sample_affine(p, size, height, width):
    """Geometric pipeline of the ADA paper (x-flip, 90-degree rotations, integer translation, isotropic scale,
    rotation split around an anisotropic scale, fractional translation); reference :151-207."""
    eye = _eye(size, 3)
    G = eye
    flip = category_sample(size, (0, 1))
    G = random_mat_apply(p, scale_mat(1 - 2.0 * flip, torch.ones(size)), G, eye)
    quarter = category_sample(size, (0, 3))
    G = random_mat_apply(p, rotate_mat(-math.pi / 2 * quarter), G, eye)
    shift = uniform_sample(size, -0.125, 0.125)
    G = random_mat_apply(p, translate_mat(torch.round(shift * width) / width, torch.round(shift * height) / height), G, eye)
    iso = lognormal_sample(size, std=0.2 * math.log(2))
    G = random_mat_apply(p, scale_mat(iso, iso), G, eye)
    p_rot = 1 - math.sqrt(1 - p)
    G = random_mat_apply(p_rot, rotate_mat(-uniform_sample(size, -math.pi, math.pi)), G, eye)
    aniso = lognormal_sample(size, std=0.2 * math.log(2))
    G = random_mat_apply(p, scale_mat(aniso, 1 / aniso), G, eye)
    G = random_mat_apply(p_rot, rotate_mat(-uniform_sample(size, -math.pi, math.pi)), G, eye)
    frac = normal_sample(size, std=0.125)
    G = random_mat_apply(p, translate_mat(frac, frac), G, eye)
    return G


def sample_color(p, size):
    """Colour pipeline (brightness, contrast, luma flip, hue rotation, saturation); reference :210-241."""
    eye = _eye(size, 4)
    C = eye
    axis = (1 / math.sqrt(3),) * 3
    b = normal_sample(size, std=0.2)
    C = random_mat_apply(p, translate3d_mat(b, b, b), C, eye)
    c = lognormal_sample(size, std=0.5 * math.log(2))
    C = random_mat_apply(p, scale3d_mat(c, c, c), C, eye)
    C = random_mat_apply(p, luma_flip_mat(axis, category_sample(size, (0, 1))), C, eye)
    C = random_mat_apply(p, rotate3d_mat(axis, uniform_sample(size, -math.pi, math.pi)), C, eye)
    C = random_mat_apply(p, saturation_mat(axis, lognormal_sample(size, std=1 * math.log(2))), C, eye)
    return C


# ---- image-space application ---------------------------------------------------------------------------------
def make_grid(shape, x0, x1, y0, y1, device):
    n, _, h, w = shape
    xs = torch.linspace(x0, x1, w, device=device).view(1, 1, w).expand(n, h, w)
    ys = torch.linspace(y0, y1, h, device=device).view(1, h, 1).expand(n, h, w)
    return torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)


def affine_grid(grid, mat):
    n, h, w, _ = grid.shape
    return (grid.reshape(n, h * w, 3) @ mat.transpose(1, 2)).reshape(n, h, w, 2)


def get_padding(G, height, width):
    """Reflect-padding (in pixels) needed so that the warped image never samples outside; reference :266-285."""
    corners = torch.tensor([(-1.0, -1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, 1)]).t()
    ext = G[:, :2, :] @ corners
    size = torch.tensor((width, height))
    low = ((ext.min(-1).values + 1) * size).clamp(max=0).abs().ceil().max(0).values.to(torch.int64).tolist()
    high = (ext.max(-1).values * size - size).clamp(min=0).ceil().max(0).values.to(torch.int64).tolist()
    return low[0], high[0], low[1], high[1]


def try_sample_affine_and_pad(img, p, pad_k, G=None):
    batch, _, height, width = img.shape
    while True:
        G_try = sample_affine(p, batch, height, width) if G is None else G
        pad_x1, pad_x2, pad_y1, pad_y2 = get_padding(torch.inverse(G_try), height, width)
        try:
            img_pad = F.pad(img, (pad_x1 + pad_k, pad_x2 + pad_k, pad_y1 + pad_k, pad_y2 + pad_k), mode='reflect')
        except RuntimeError:
            if G is not None:
                raise
            continue            # padding larger than the image: draw again (reference :288-313)
        return img_pad, G_try, (pad_x1, pad_x2, pad_y1, pad_y2)


def random_apply_affine(img, p, G=None, antialiasing_kernel=SYM6):
    """Reflect-pad -> x2 up-sample (sym6) -> bilinear warp by inverse(G) -> x2 down-sample (sym6) -> crop."""
    taps = torch.as_tensor(antialiasing_kernel, dtype=torch.float32)
    len_k = taps.numel()
    pad_k = (len_k + 1) // 2
    kernel = torch.outer(taps, taps).to(img)
    kernel_flip = torch.flip(kernel, (0, 1))
    img_pad, G, (pad_x1, pad_x2, pad_y1, pad_y2) = try_sample_affine_and_pad(img, p, pad_k, G)
    h_o, w_o = img.shape[2], img.shape[3]
    h_p, w_p = img_pad.shape[2] - len_k + 1, img_pad.shape[3] - len_k + 1

    img_2x = upfirdn2d(img_pad, kernel_flip, up=2)
    grid = make_grid(img_2x.shape, -2 * pad_x1 / w_o - 1, 2 * (w_p - pad_x1) / w_o - 1,
                     -2 * pad_y1 / h_o - 1, 2 * (h_p - pad_y1) / h_o - 1, device=img_2x.device).to(img_2x)
    grid = affine_grid(grid, torch.inverse(G)[:, :2, :].to(img_2x))
    grid = grid * torch.tensor([w_o / w_p, h_o / h_p], device=grid.device) + \
        torch.tensor([(w_o + 2 * pad_x1) / w_p - 1, (h_o + 2 * pad_y1) / h_p - 1], device=grid.device)
    img_affine = F.grid_sample(img_2x, grid, mode='bilinear', align_corners=False, padding_mode='zeros')
    img_down = upfirdn2d(img_affine, kernel, down=2)
    end_y = img_down.shape[2] if pad_y2 + 1 == 0 else -pad_y2 - 1
    end_x = img_down.shape[3] if pad_x2 + 1 == 0 else -pad_x2 - 1
    return img_down[:, :, pad_y1:end_y, pad_x1:end_x], G


def apply_color(img, mat):
    """Per-sample 3x4 colour matrix applied to the channel axis."""
    lin = mat[:, :3, :3].to(img)
    off = mat[:, :3, 3].to(img)
    return torch.einsum('bij,bjhw->bihw', lin, img) + off[:, :, None, None]


def random_apply_color(img, p, C=None):
    if C is None:
        C = sample_color(p, img.shape[0])
    return apply_color(img, C.to(img)), C


def augment(img, p, transform_matrix=(None, None)):
    img, G = random_apply_affine(img, p, transform_matrix[0])
    img, C = random_apply_color(img, p, transform_matrix[1])
    return img, (G, C)


class AdaptiveAugmentState:
    """ADA probability controller of the trainer (generator_trainer.py:333-339, 669-688)."""

    def __init__(self, augment_config, device):
        self.cfg = augment_config
        self.p = augment_config['p'] if augment_config['p'] > 0 else 0.0
        self.step = augment_config['ada_target'] / augment_config['ada_length']
        self.accum = torch.zeros(2, device=device)       # [sum sign(real_pred), count]
        self.r_t = 0.0

    def update(self, real_pred, reduce_sum=None):
        """Accumulate sign statistics of D's real predictions; every > 255 predictions move p towards the target."""
        self.accum += torch.stack([torch.sign(real_pred).sum(), torch.tensor(float(real_pred.shape[0]), device=real_pred.device)])
        stats = self.accum.clone()
        if reduce_sum is not None:
            stats = reduce_sum(stats)
        if float(stats[1]) > 255:
            signs, count = stats.tolist()
            self.r_t = signs / count
            if self.cfg['enabled'] and self.cfg['p'] == 0:
                sign = 1 if self.r_t > self.cfg['ada_target'] else -1
                self.p = min(1.0, max(0.0, self.p + sign * self.step * count))
            self.accum.zero_()
        return self.p
