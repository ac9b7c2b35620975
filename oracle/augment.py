"""Oracle for the ADA image-space transforms (test infrastructure only -- see oracle/__init__.py).

CPU restatement of random_apply_affine / apply_color of the reference's trainers/non_leaking.py
(:316-371, :374-391) for GIVEN transform matrices, on the oracle's own upfirdn2d.
"""
import torch
import torch.nn.functional as F

from . import ops


def apply_affine(img, G, taps):
    """Reference: random_apply_affine non_leaking.py:316-371 with an explicit G (no sampling)."""
    n, c, h_o, w_o = img.shape
    taps = torch.as_tensor(taps, dtype=img.dtype)
    k = torch.outer(taps, taps)
    len_k = taps.numel()
    pad_k = (len_k + 1) // 2
    # get_padding, non_leaking.py:266-285
    Ginv = torch.inverse(G)
    ext = Ginv[:, :2, :] @ torch.tensor([(-1.0, -1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, 1)], dtype=Ginv.dtype).t()
    size = torch.tensor((w_o, h_o))
    lo = ((ext.min(-1).values + 1) * size).clamp(max=0).abs().ceil().max(0).values.to(torch.int64).tolist()
    hi = (ext.max(-1).values * size - size).clamp(min=0).ceil().max(0).values.to(torch.int64).tolist()
    px1, px2, py1, py2 = lo[0], hi[0], lo[1], hi[1]
    x = F.pad(img, (px1 + pad_k, px2 + pad_k, py1 + pad_k, py2 + pad_k), mode='reflect')
    w_p, h_p = x.shape[3] - len_k + 1, x.shape[2] - len_k + 1
    x2 = ops.upfirdn2d(x, torch.flip(k, (0, 1)), up=2)
    n2, _, h2, w2 = x2.shape
    gx = torch.linspace(-2 * px1 / w_o - 1, 2 * (w_p - px1) / w_o - 1, w2).view(1, 1, w2).expand(n2, h2, w2)
    gy = torch.linspace(-2 * py1 / h_o - 1, 2 * (h_p - py1) / h_o - 1, h2).view(1, h2, 1).expand(n2, h2, w2)
    grid = torch.stack([gx, gy, torch.ones_like(gx)], -1).to(img.dtype)
    grid = (grid.reshape(n2, h2 * w2, 3) @ Ginv[:, :2, :].to(img.dtype).transpose(1, 2)).reshape(n2, h2, w2, 2)
    grid = grid * torch.tensor([w_o / w_p, h_o / h_p], dtype=img.dtype) + \
        torch.tensor([(w_o + 2 * px1) / w_p - 1, (h_o + 2 * py1) / h_p - 1], dtype=img.dtype)
    warped = F.grid_sample(x2, grid, mode='bilinear', align_corners=False, padding_mode='zeros')
    down = ops.upfirdn2d(warped, k, down=2)
    return down[:, :, py1:down.shape[2] - py2 - 1, px1:down.shape[3] - px2 - 1]


def apply_color(img, C):
    """Reference: apply_color non_leaking.py:374-382."""
    x = img.permute(0, 2, 3, 1)
    x = x @ C[:, :3, :3].transpose(1, 2).reshape(img.shape[0], 1, 3, 3).to(img.dtype) + C[:, :3, 3].reshape(img.shape[0], 1, 1, 3).to(img.dtype)
    return x.permute(0, 3, 1, 2)


def augment(img, G, C, taps):
    return apply_color(apply_affine(img, G, taps), C)
