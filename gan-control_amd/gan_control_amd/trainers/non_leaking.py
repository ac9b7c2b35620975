"""Adaptive discriminator augmentation (ADA): random affine + colour transforms of image batches.

Mirrors the call surface of the reference's src/gan_control/trainers/non_leaking.py
(``augment(img, p, transform_matrix=(None, None))`` :394-398, ``random_apply_affine`` :316-371,
``random_apply_color`` :385-391, ``sample_affine`` :151-207, ``sample_color`` :210-241), which the trainer
calls when ``training_config.augment.enabled`` (generator_trainer.py:421-424, 651-653).  The reference file
imports ``upfirdn2d`` from a package it does not ship (non_leaking.py:6); here the two 12x12 sym6
anti-aliasing passes (x2 up-sampling before, x2 down-sampling after the warp) run on the HIP upfirdn2d
kernel (LDS-tiled 12 x 12 variant), with first- and second-order gradients.

Transform matrices are sampled on the host with the SAME order of random draws as the reference, so a
given torch seed yields the same augmentation (pinned by tests/golden/augment.npz).  The reflect padding and the
bilinear warp run on HIP kernels too (models/op/warp.py): the reference's sampling grid is an affine function of the
output pixel index, folded on the host into one 2 x 3 matrix per sample (``warp_matrix``).
"""
import math

import torch
from torch.nn import functional as F

from ..models.op import upfirdn2d, affine_warp_bilinear, reflect_pad

# Daubechies least-asymmetric ("symlet") 6 low-pass filter, 12 taps -- the ADA paper's anti-aliasing kernel
SYM6 = (
    0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633,
    0.4910559419267466, 0.787641141030194, 0.3379294217276218, -0.07263752278646252,
    -0.021060292512300564, 0.04472490177066578, 0.0017677118642428036, -0.007800708325034148,
)


# ---- batched homogeneous matrices ------------------------------------------------------------------------------
def _identity(n, dim):
    return torch.eye(dim).unsqueeze(0).repeat(n, 1, 1)


def _with(n, dim, entries):
    """Identity [n, dim, dim] with the given {(row, col): values[n]} entries overwritten."""
    m = _identity(n, dim)
    for (r, c), v in entries.items():
        m[:, r, c] = v
    return m


def _shift2(tx, ty):
    return _with(tx.shape[0], 3, {(0, 2): tx, (1, 2): ty})


def _rot2(theta):
    c, s = torch.cos(theta), torch.sin(theta)
    return _with(theta.shape[0], 3, {(0, 0): c, (0, 1): -s, (1, 0): s, (1, 1): c})


def _scale2(sx, sy):
    return _with(sx.shape[0], 3, {(0, 0): sx, (1, 1): sy})


def _shift3(t):
    return _with(t.shape[0], 4, {(0, 3): t, (1, 3): t, (2, 3): t})


def _scale3(k):
    return _with(k.shape[0], 4, {(0, 0): k, (1, 1): k, (2, 2): k})


_GREY = (1 / math.sqrt(3),) * 3          # luma axis of the colour transforms


def _rot3_about_grey(theta):
    """Rodrigues rotation about the grey axis (hue rotation)."""
    u = torch.tensor(_GREY, dtype=torch.float32)
    ux, uy, uz = _GREY
    cross = torch.tensor([[0.0, -uz, uy], [uz, 0.0, -ux], [-uy, ux, 0.0]], dtype=torch.float32)
    c, s = torch.cos(theta).view(-1, 1, 1), torch.sin(theta).view(-1, 1, 1)
    m = _identity(theta.shape[0], 4)
    m[:, :3, :3] = c * torch.eye(3) + s * cross + (1 - c) * torch.outer(u, u)
    return m


def _grey_projector():
    v = torch.tensor(_GREY + (0,), dtype=torch.float32)
    return torch.outer(v, v)


def _luma_flip(i):
    return _identity(i.shape[0], 4) - 2 * _grey_projector() * i.view(-1, 1, 1)


def _saturation(k):
    proj = _grey_projector()
    return proj + (_identity(k.shape[0], 4) - proj) * k.view(-1, 1, 1)


# ---- samplers: exactly one torch RNG call each, so the draw order below reproduces the reference's stream -----
def _lognormal(n, std):
    return torch.empty(n).log_normal_(mean=0, std=std)


def _choice(n, values):
    return torch.tensor(values)[torch.randint(high=len(values), size=(n,))]


def _uniform(n, lo, hi):
    return torch.empty(n).uniform_(lo, hi)


def _normal(n, std):
    return torch.empty(n).normal_(0, std)


def _maybe(p, transform, prev):
    """Per sample, with probability p: prev <- transform @ prev (one bernoulli draw of size n)."""
    n = transform.shape[0]
    on = torch.empty(n).bernoulli_(p).view(n, 1, 1)
    return (on * transform + (1 - on) * _identity(n, transform.shape[1])) @ prev


def sample_affine(p, size, height, width):
    """Geometric ADA pipeline: x-flip, 90-degree rotation, integer translation, isotropic scale, rotation split
    around an anisotropic scale, fractional translation (reference sample_affine :151-207).  Every stage draws
    its parameter first and its on/off mask second -- the reference's RNG order."""
    ln2 = math.log(2)
    p_rot = 1 - math.sqrt(1 - p)
    G = _identity(size, 3)
    G = _maybe(p, _scale2(1 - 2.0 * _choice(size, (0, 1)), torch.ones(size)), G)
    G = _maybe(p, _rot2(-math.pi / 2 * _choice(size, (0, 3))), G)
    t = _uniform(size, -0.125, 0.125)
    G = _maybe(p, _shift2(torch.round(t * width) / width, torch.round(t * height) / height), G)
    k = _lognormal(size, 0.2 * ln2)
    G = _maybe(p, _scale2(k, k), G)
    G = _maybe(p_rot, _rot2(-_uniform(size, -math.pi, math.pi)), G)
    k = _lognormal(size, 0.2 * ln2)
    G = _maybe(p, _scale2(k, 1 / k), G)
    G = _maybe(p_rot, _rot2(-_uniform(size, -math.pi, math.pi)), G)
    t = _normal(size, 0.125)
    return _maybe(p, _shift2(t, t), G)


def sample_color(p, size):
    """Colour ADA pipeline: brightness, contrast, luma flip, hue rotation, saturation (reference :210-241)."""
    ln2 = math.log(2)
    C = _identity(size, 4)
    C = _maybe(p, _shift3(_normal(size, 0.2)), C)
    C = _maybe(p, _scale3(_lognormal(size, 0.5 * ln2)), C)
    C = _maybe(p, _luma_flip(_choice(size, (0, 1))), C)
    C = _maybe(p, _rot3_about_grey(_uniform(size, -math.pi, math.pi)), C)
    return _maybe(p, _saturation(_lognormal(size, 1 * ln2)), C)


# ---- image-space application ---------------------------------------------------------------------------------
def make_grid(shape, x0, x1, y0, y1, device):
    n, _, h, w = shape
    xs = torch.linspace(x0, x1, w, device=device).view(1, 1, w).expand(n, h, w)
    ys = torch.linspace(y0, y1, h, device=device).view(1, h, 1).expand(n, h, w)
    return torch.stack([xs, ys, torch.ones_like(xs)], dim=-1)


def affine_grid(grid, mat):
    n, h, w, _ = grid.shape
    return (grid.reshape(n, h * w, 3) @ mat.transpose(1, 2)).reshape(n, h, w, 2)


def warp_matrix(g_inv, h2, w2, x_range, y_range, scale, shift):
    """The reference's sampling grid (make_grid -> affine_grid -> scale / shift, then grid_sample's un-normalisation with
    align_corners=False; non_leaking.py:244-263, 338-357) is an affine function of the output pixel index (ox, oy).  Returns it as
    [B, 6] rows (m0, m1, m2, m3, m4, m5): sx = m0*ox + m1*oy + m2, sy = m3*ox + m4*oy + m5 in input pixel units, in float64 on the host."""
    a = g_inv[:, :2, :].double()
    (x0, x1), (y0, y1) = x_range, y_range
    dx, dy = (x1 - x0) / (w2 - 1), (y1 - y0) / (h2 - 1)
    half = torch.tensor([w2 / 2.0, h2 / 2.0], dtype=torch.float64)
    k = torch.tensor(scale, dtype=torch.float64) * half                       # normalised grid -> pixels, per axis
    off = (torch.tensor(shift, dtype=torch.float64)) * half + torch.tensor([(w2 - 1) / 2.0, (h2 - 1) / 2.0], dtype=torch.float64)
    m = torch.empty(a.shape[0], 6, dtype=torch.float64)
    for r in range(2):                                                         # r = 0: sx from row 0 of inverse(G); r = 1: sy
        m[:, 3 * r + 0] = k[r] * a[:, r, 0] * dx
        m[:, 3 * r + 1] = k[r] * a[:, r, 1] * dy
        m[:, 3 * r + 2] = k[r] * (a[:, r, 0] * x0 + a[:, r, 1] * y0 + a[:, r, 2]) + off[r]
    return m.float()


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
            img_pad = reflect_pad(img, (pad_x1 + pad_k, pad_x2 + pad_k, pad_y1 + pad_k, pad_y2 + pad_k))
        except ValueError as e:
            if G is not None:
                raise RuntimeError(str(e)) from e
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
    mat = warp_matrix(torch.inverse(G), img_2x.shape[2], img_2x.shape[3], (-2 * pad_x1 / w_o - 1, 2 * (w_p - pad_x1) / w_o - 1),
                      (-2 * pad_y1 / h_o - 1, 2 * (h_p - pad_y1) / h_o - 1), (w_o / w_p, h_o / h_p),
                      ((w_o + 2 * pad_x1) / w_p - 1, (h_o + 2 * pad_y1) / h_p - 1))
    img_affine = affine_warp_bilinear(img_2x, mat.to(img_2x), img_2x.shape[2], img_2x.shape[3])
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
        self.accum = torch.zeros(2, device=device)       # [sum sign(real_pred), count] of THIS rank since the last decision
        self.count = 0                                   # the same count summed over ranks, kept on the host
        self.r_t = 0.0

    def update(self, real_pred, reduce_sum=None, world=1):
        """Accumulate sign statistics of D's real predictions; every > 255 predictions (over all ranks) move p towards the
        target.  The prediction count is known on the host, so the device is only read (one scalar, and one all-reduce when
        ``reduce_sum`` is given) in the iterations that cross the threshold -- every 256 / global-batch steps -- instead of
        a host synchronisation per D step (generator_trainer.py:669-671 does ``.item()`` every step)."""
        n = int(real_pred.shape[0])
        self.accum[0] += torch.sign(real_pred).sum()
        self.accum[1] += n
        self.count += n * world
        if self.count > 255:
            stats = self.accum.clone()
            if reduce_sum is not None:
                stats = reduce_sum(stats)
            signs, count = stats.tolist()
            self.r_t = signs / count
            if self.cfg['enabled'] and self.cfg['p'] == 0:
                sign = 1 if self.r_t > self.cfg['ada_target'] else -1
                self.p = min(1.0, max(0.0, self.p + sign * self.step * count))
            self.accum.zero_()
            self.count = 0
        return self.p
