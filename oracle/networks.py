"""Network-level oracle: Generator / Discriminator forward as pure functions of a state_dict.

The state_dict uses the reference's key names, so the same
dictionary drives the reference modules (make_golden.py), this oracle and the HIP
product.  Test infrastructure only -- see oracle/__init__.py.
"""
import math
import zlib

import torch

from . import ops

BLUR = (1, 3, 3, 1)


def n_latent(size):
    """Reference: Generator.__init__ gan_model.py:616."""
    return int(math.log2(size)) * 2 - 2


def procedural_fill_(state_dict, salt=0):
    """Overwrite every floating entry with values derived from its key name.

    Network weights are too large to commit (>= 7 M parameters even at size 8), so
    fixtures pin networks whose weights are generated from CRC32(key) seeds.  Applied
    identically to the reference modules (in make_golden.py) and to the product.
    Blur kernels (``*.kernel``) are buffers fixed by construction and are left alone.
    Returns the dict for convenience.
    """
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if not torch.is_floating_point(t) or key.endswith('.kernel'):
            continue
        gen = torch.Generator(device='cpu')
        gen.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
        v = torch.randn(t.shape, generator=gen, dtype=torch.float32)
        if key.endswith('noise.weight'):          # NoiseInjection strength, zero at init
            v = v * 0.1
        elif key.endswith('.bias') and 'modulation' in key:
            v = 1.0 + 0.1 * v                      # modulation bias is initialised to 1
        elif key.endswith('.bias') or key.endswith('activate.bias'):
            v = v * 0.1
        elif key.startswith('style.') and key.endswith('.weight'):
            v = v * 100.0                          # mapping weights are stored / lr_mul (0.01)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict


def _mapping(sd, z, lr_mlp=0.01, fc_groups=None):
    """Reference: create_regular_fc_stack gan_model.py:633-642; MultiFcStack gan_model.py:489-502."""
    def stack(prefix, v):
        v = ops.pixel_norm(v)
        i = 1
        while f'{prefix}{i}.weight' in sd:
            v = ops.equal_linear(v, sd[f'{prefix}{i}.weight'], sd[f'{prefix}{i}.bias'], lr_mul=lr_mlp, activation=True)
            i += 1
        return v

    if fc_groups is None:
        return stack('style.', z)
    outs = []
    for name, (lo, hi) in fc_groups:
        outs.append(stack(f'style.{name}.', z[:, lo:hi]))
    return torch.cat(outs, dim=1)


def _styled_conv(sd, prefix, x, w, noise, upsample, noise_mode='normal'):
    """Reference: StyledConv.forward gan_model.py:402-408, NoiseInjection gan_model.py:340-345; noise_mode 'zeros' / 'id_zeros' =
    ModulatedNoiseInjection gan_model.py:1019-1035 (no noise at all / noise on the first half of the channels only)."""
    y = ops.modulated_conv2d(x, w, sd[f'{prefix}.conv.weight'], sd[f'{prefix}.conv.modulation.weight'],
                             sd[f'{prefix}.conv.modulation.bias'], demodulate=True, upsample=upsample)
    if noise_mode != 'zeros':
        if noise is None:
            noise = torch.randn(y.shape[0], 1, y.shape[2], y.shape[3], dtype=y.dtype, device=y.device)
        if noise_mode == 'id_zeros':
            pose, ident = torch.chunk(y, 2, dim=1)
            y = torch.cat([pose + sd[f'{prefix}.noise.weight'] * noise, ident], dim=1)
        else:
            y = y + sd[f'{prefix}.noise.weight'] * noise
    return ops.fused_leaky_relu(y, sd[f'{prefix}.activate.bias'])


def _to_rgb(sd, prefix, x, w, skip):
    """Reference: ToRGB.forward gan_model.py:424-435; Upsample gan_model.py:71-89 (pad (2, 1))."""
    y = ops.modulated_conv2d(x, w, sd[f'{prefix}.conv.weight'], sd[f'{prefix}.conv.modulation.weight'],
                             sd[f'{prefix}.conv.modulation.bias'], demodulate=False)
    y = y + sd[f'{prefix}.bias']
    if skip is not None:
        y = y + ops.upfirdn2d(skip, ops.fir_kernel(BLUR, 4.0).to(skip), up=2, pad=(2, 1))
    return y


def generator_forward(sd, styles, size, noise=None, input_is_latent=False, inject_index=None,
                      truncation=1.0, truncation_latent=None, fc_groups=None, noise_mode='normal'):
    """Reference: Generator.forward gan_model.py:709-801.  Returns (image, latent [B, n_latent, D]).

    ``styles`` is a list of z (or w if input_is_latent) tensors; ``noise`` a list of
    per-layer maps (None entries are drawn fresh, as randomize_noise=True does).
    """
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    nl = n_latent(size)
    if not input_is_latent:
        styles = [_mapping(sd, s, fc_groups=fc_groups) for s in styles]
    if noise is None:
        noise = [None] * num_layers
    if truncation < 1:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, nl, 1) if styles[0].ndim < 3 else styles[0]
    else:
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, nl - inject_index, 1)], 1)
    b = latent.shape[0]
    x = sd['input.input'].repeat(b, 1, 1, 1)
    x = _styled_conv(sd, 'conv1', x, latent[:, 0], noise[0], upsample=False, noise_mode=noise_mode)
    skip = _to_rgb(sd, 'to_rgb1', x, latent[:, 1], None)
    li = 1
    for blk in range(log_size - 2):
        x = _styled_conv(sd, f'convs.{2 * blk}', x, latent[:, li], noise[1 + 2 * blk], upsample=True, noise_mode=noise_mode)
        x = _styled_conv(sd, f'convs.{2 * blk + 1}', x, latent[:, li + 1], noise[2 + 2 * blk], upsample=False)      # the second convolution of a block is always built with the default mode (gan_model.py:606-610)
        skip = _to_rgb(sd, f'to_rgbs.{blk}', x, latent[:, li + 2], skip)
        li += 2
    return skip, latent


def _conv_layer(sd, prefix, x, downsample, activate, ksize):
    """Reference: ConvLayer gan_model.py:844-890 (Blur -> EqualConv2d -> FusedLeakyReLU)."""
    idx = 0
    if downsample:
        p = (len(BLUR) - 2) + (ksize - 1)
        x = ops.upfirdn2d(x, ops.fir_kernel(BLUR).to(x), pad=((p + 1) // 2, p // 2))
        idx = 1
    wkey = f'{prefix}.{idx}.weight'
    bkey = f'{prefix}.{idx}.bias'
    x = ops.equal_conv2d(x, sd[wkey], sd.get(bkey) if not activate else None,
                         stride=2 if downsample else 1, padding=0 if downsample else ksize // 2)
    if activate:
        x = ops.fused_leaky_relu(x, sd[f'{prefix}.{idx + 1}.bias'])
    return x


def discriminator_forward(sd, img):
    """Reference: Discriminator.forward gan_model.py:990-1016, ResBlock gan_model.py:893-922."""
    x = _conv_layer(sd, 'convs.0', img, False, True, 1)
    i = 1
    while f'convs.{i}.conv1.0.weight' in sd:
        y = _conv_layer(sd, f'convs.{i}.conv1', x, False, True, 3)
        y = _conv_layer(sd, f'convs.{i}.conv2', y, True, True, 3)
        s = _conv_layer(sd, f'convs.{i}.skip', x, True, False, 1)
        x = (y + s) / math.sqrt(2)
        i += 1
    x = ops.minibatch_stddev(x, 4)
    x = _conv_layer(sd, 'final_conv', x, False, True, 3)
    x = x.reshape(x.shape[0], -1)
    x = ops.equal_linear(x, sd['final_linear.0.weight'], sd['final_linear.0.bias'], activation=True)
    return ops.equal_linear(x, sd['final_linear.1.weight'], sd['final_linear.1.bias'])
