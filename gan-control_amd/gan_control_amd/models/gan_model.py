"""StyleGAN2 Generator / Discriminator of gan-control on the gfx950 operator socket.

Mirrors the module API of the reference's src/gan_control/models/gan_model.py for everything the
trainer can reach: same constructor arguments, forward signatures, return tuples and
state_dict keys, so reference checkpoints load unchanged.  Every tensor
operation on the hot path goes through ``gan_control_amd.models.op`` (HIP kernels); there is no
ATen convolution, no materialised per-sample weight and no CPU fallback in here.

Unreachable reference features (VAE mapping, model_mode='896', verification head, Downsample
branch of ModulatedConv2d, ScaledLeakyReLU) are not built.
"""
import math
import random

import torch
from torch import nn, autograd
from torch.nn import functional as F

import os
_FUSE_EPILOGUE = os.environ.get('GANCONTROL_FUSE_EPILOGUE', '1') != '0'   # debugging knob: 0 = convolution and activation as two launches
_FUSE_BLUR_ADJ = os.environ.get('GANCONTROL_FUSE_BLUR_ADJOINT', '1') != '0'   # ResBlock: Blur adjoint + conv1's activation backward in one pass over the gradient
_FORK_TORGB = os.environ.get('GANCONTROL_FORK_TORGB', '1') == '1'         # round 2: -0.3 % (the in_scale gradient needed gx - gfork, a pass of its own); on since that gradient comes from the per-sample weight gradient (modulated_conv._samples_route): four image-sized adds less per generator backward
from .op import _backend
from .op import (FusedLeakyReLU, fused_leaky_relu, upfirdn2d, upfirdn2d_bias_act, conv2d_gradfix, modulated_conv2d,
                 modulated_conv2d_act)
from .op.upfirdn2d import blur_of_activation
from .op.modulated_conv import demod_coefficients, weight_sq_all, _eps_vector
from .op.linear import scaled_mm, equal_linear
from .op import style as style_op
# The style path (26 modulations, 18 demodulation sums per forward pass) as grouped launches over layer-major flat tensors (op/style.py).
# GANCONTROL_FUSED_STYLE=0: one GEMM call per layer, as in round 2.
_FUSED_STYLE = os.environ.get('GANCONTROL_FUSED_STYLE', '1') != '0'
import weakref
_STYLE_PLANS = weakref.WeakKeyDictionary()      # Generator -> _StylePlan (kept out of the module: holds ctypes tables, not state)


class _StylePlan:
    """Which modulated convolution reads which latent, in which group of the grouped launches.  Groups are ordered demodulated layers
    first (so that the demodulation step reads a contiguous prefix of the modulation output), ToRGB layers last."""

    def __init__(self, layers, style_dim):
        order = [j for j, (m, _) in enumerate(layers) if m.demodulate] + [j for j, (m, _) in enumerate(layers) if not m.demodulate]
        self.layers = [layers[j][0] for j in order]                    # group order
        self.latent_index = [layers[j][1] for j in order]
        self.group_of_exec = {j: g for g, j in enumerate(order)}
        self.n_demod = sum(1 for m in self.layers if m.demodulate)
        self.s_cols = [m.in_channel for m in self.layers]
        self.d_cols = [m.out_channel for m in self.layers[:self.n_demod]]
        self.mod_plan = style_op.Plan([style_op.GroupSpec(m.in_channel, style_dim, m.modulation.scale, m.modulation.lr_mul, g * style_dim)
                                       for g, m in enumerate(self.layers)])
        specs, at = [], 0
        for m in self.layers[:self.n_demod]:
            specs.append(style_op.GroupSpec(m.out_channel, m.in_channel, m.scale * m.scale, 1.0, at))
            at += m.in_channel
        self.demod_in_cols = at
        self.demod_plan = style_op.Plan(specs, in_cols=at)
        self.ok = style_op._supported(self.mod_plan) and style_op._supported(self.demod_plan)
        self._idx = {}

    def selector(self, device, n_latent):
        """[groups, n_latent] one-hot rows: ``selector @ latents`` is the per-layer gather as ONE GEMM call.  Exact (every product is 1 * x or
        0 * x) and -- unlike index_select, whose backward scatters with atomics -- deterministic in both directions: the gradient of a
        latent read by several layers is summed in the GEMM's fixed order.  Two caveats: the product must be exact -- _gather_latents
        runs it in float64, forward and backward, when the process-wide matmul precision is not the default (set_float32_matmul_precision('high')
        / allow_tf32 would otherwise round the latents on this route only) -- and a non-finite latent entry
        reaches EVERY layer's style as NaN through 0 * inf, where a gather would confine it to the layers that read that latent (the image is
        non-finite either way; GANCONTROL_FUSED_STYLE=0 selects the per-layer route)."""
        t = self._idx.get((device, n_latent))
        if t is None:
            t = torch.zeros(len(self.latent_index), n_latent, dtype=torch.float32)
            t[torch.arange(len(self.latent_index)), torch.tensor(self.latent_index)] = 1.0
            t = self._idx[(device, n_latent)] = t.to(device)
        return t
CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512}


def channel_table(channel_multiplier):
    """Reference: gan_model.py:552-563 and :931-941."""
    t = dict(CHANNELS)
    for res, base in ((64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)):
        t[res] = int(base * channel_multiplier)
    return t


def make_kernel(k):
    """Reference: gan_model.py:60-68."""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def _gather_latents(plan, latent):
    """latent [B, n_latent, D] -> [groups, B * D]: row g = the latent layer g reads (StyleGroupPlan.selector: one exact GEMM).  The product
    has to be exact whatever the process-wide matmul precision says, so that the grouped and the per-layer style paths see the same bits:
    under the default ('highest', no TF32) it is one fp32 GEMM; under any other setting it runs in float64 (every product is 1 * x or 0 * x,
    [groups, n_latent] @ [n_latent, B * D] is tiny) -- no process-wide state is touched (writing the setting back through the legacy
    ``allow_tf32`` attribute poisons ``get_float32_matmul_precision`` on current torch)."""
    flat = latent.transpose(0, 1).reshape(latent.shape[1], -1)
    sel = plan.selector(latent.device, latent.shape[1])
    if torch.get_float32_matmul_precision() == 'highest' and not torch.backends.cuda.matmul.allow_tf32:
        return sel @ flat
    return (sel.double() @ flat.double()).to(flat.dtype)


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(input.pow(2).mean(dim=1, keepdim=True) + 1e-8)


class _Fir(nn.Module):
    """Common base of Upsample / Downsample / Blur: owns the ``kernel`` buffer (gan_model.py:77,98,122)."""

    def __init__(self, kernel, gain, up, down, pad):
        super().__init__()
        self.register_buffer('kernel', make_kernel(kernel) * gain)
        self.up, self.down, self.pad = up, down, pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.up, down=self.down, pad=self.pad)


class Upsample(_Fir):
    def __init__(self, kernel, factor=2):
        p = len(kernel) - factor
        super().__init__(kernel, factor ** 2, factor, 1, ((p + 1) // 2 + factor - 1, p // 2))
        self.factor = factor


class Downsample(_Fir):
    def __init__(self, kernel, factor=2):
        p = len(kernel) - factor
        super().__init__(kernel, 1, 1, factor, ((p + 1) // 2, p // 2))
        self.factor = factor


class Blur(_Fir):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__(kernel, upsample_factor ** 2 if upsample_factor > 1 else 1, 1, 1, pad)


class EqualConv2d(nn.Module):
    """Reference: gan_model.py:132-168.  Runs on conv2d_gradfix (MFMA implicit GEMM)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return conv2d_gradfix.conv2d(input, self.weight, bias=self.bias, stride=self.stride, padding=self.padding, weight_scale=self.scale)

    def __repr__(self):
        oc, ic, k, _ = self.weight.shape
        return f'{self.__class__.__name__}({ic}, {oc}, {k}, stride={self.stride}, padding={self.padding})'


class EqualLinear(nn.Module):
    """Reference: gan_model.py:171-202.  The GEMM is a plain library GEMM (rocBLAS via F.linear)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init))) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, input):
        # lr_mul * bias + scale * (x @ W^T) as ONE GEMM call (alpha / beta) instead of two elementwise passes + GEMM:
        # the mapping network and the 26 style modulations are ~400 tiny launches per iteration otherwise
        x = input.reshape(-1, input.shape[-1])
        if self.bias is None:
            out = scaled_mm(x, self.weight.t(), self.scale)
        else:
            out = equal_linear(x, self.weight, self.bias, self.scale, self.lr_mul)
        out = out.reshape(*input.shape[:-1], self.weight.shape[0])
        if self.activation:
            if self.bias is None:
                raise NotImplementedError('EqualLinear: activation without bias is not built')
            return fused_leaky_relu(out, None)
        return out

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ModulatedConv2d(nn.Module):
    """Reference: gan_model.py:217-331 (plain and conv_transpose up-sampling branches)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1], conv_transpose=False, overwrite_padding=None):
        super().__init__()
        if not conv_transpose:
            raise ValueError('conv_transpose is %s' % str(conv_transpose))
        if downsample or overwrite_padding is not None:
            raise NotImplementedError('ModulatedConv2d: downsample / overwrite_padding are unreachable from the trainer and not built')
        self.eps = 1e-8
        self.kernel_size, self.in_channel, self.out_channel = kernel_size, in_channel, out_channel
        self.upsample, self.downsample, self.conv_transpose = upsample, downsample, conv_transpose
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    def styles(self, style):
        """(s, d): the modulation factors and the demodulation coefficients of this layer for the latent ``style``."""
        s = self.modulation(style)
        d = demod_coefficients(self.weight, s, self.scale) if self.demodulate else None
        return s, d

    def forward(self, input, style):
        s = self.modulation(style)
        if self.upsample:
            return modulated_conv2d(input, self.weight, s, demodulate=self.demodulate, upsample=True,
                                    blur_kernel=self.blur.kernel, blur_pad=self.blur.pad)
        return modulated_conv2d(input, self.weight, s, demodulate=self.demodulate, padding=self.padding)


class NoiseInjection(nn.Module):
    """Owns the noise strength (gan_model.py:334-345); the add itself is fused into the activation kernel."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    """Reference: gan_model.py:361-408.  conv -> (noise + bias + leaky-ReLU) in one fused pass."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, conv_transpose=False, overwrite_padding=None, noise_mode='normal'):
        super().__init__()
        if noise_mode not in ('normal', 'same_for_same_id', 'zeros', 'id_zeros'):
            raise ValueError(f'unknown noise_mode {noise_mode!r}')
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate, conv_transpose=conv_transpose,
                                    overwrite_padding=overwrite_padding)
        self.noise_mode = noise_mode
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, mod=None):
        """mod: (s, d) computed ahead of time by ``conv.styles(style)`` (Generator.forward's style path)."""
        conv = self.conv
        s_pre, d_pre = mod if mod is not None else (None, None)
        if self.noise_mode in ('zeros', 'id_zeros'):
            # ModulatedNoiseInjection (gan_model.py:1019-1035): 'zeros' injects nothing (the strength parameter stays without a gradient),
            # 'id_zeros' adds the noise to the first half of the channels (torch.chunk) only.  Not used by the shipped configs: the
            # convolution and the activation are the HIP ops, the half-tensor add is left to ATen.
            s = s_pre if mod is not None else conv.modulation(style)
            if conv.upsample:
                out = modulated_conv2d(input, conv.weight, s, demodulate=conv.demodulate, upsample=True, blur_kernel=conv.blur.kernel,
                                       blur_pad=conv.blur.pad, demod=d_pre)
            else:
                out = modulated_conv2d(input, conv.weight, s, demodulate=conv.demodulate, padding=conv.padding, demod=d_pre)
            if self.noise_mode == 'id_zeros':
                if noise is None:
                    b, _, h, w = out.shape
                    noise = out.new_empty(b, 1, h, w).normal_()
                pose, ident = torch.chunk(out, 2, dim=1)
                out = torch.cat([pose + self.noise.weight * noise, ident], dim=1)
            return self.activate(out)
        if noise is not None and noise.shape[0] == 1 and input.shape[0] > 1:
            # the registered [1, 1, h, w] buffers of randomize_noise=False broadcast over the batch (gan_model.py:343-345)
            noise = noise.expand(input.shape[0], -1, -1, -1)
        if _FUSE_EPILOGUE and not conv.upsample:
            # conv -> noise -> bias + leaky-ReLU in one launch (the activation runs in the convolution's epilogue)
            if noise is None:
                b, _, h, w = input.shape
                pad = conv.padding
                noise = input.new_empty(b, 1, h + 2 * pad - conv.kernel_size + 1, w + 2 * pad - conv.kernel_size + 1).normal_()
            act = self.activate
            return modulated_conv2d_act(input, conv.weight, s_pre if mod is not None else conv.modulation(style), act.bias, noise, self.noise.weight,
                                        demodulate=conv.demodulate, padding=conv.padding,
                                        negative_slope=act.negative_slope, act_scale=act.scale, demod=d_pre)
        if _FUSE_EPILOGUE and conv.upsample:
            # transposed conv, then Blur -> noise -> bias + leaky-ReLU in one launch (the activation runs in the FIR's epilogue)
            out = modulated_conv2d(input, conv.weight, s_pre if mod is not None else conv.modulation(style), demodulate=conv.demodulate, upsample=True,
                                   apply_blur=False, demod=d_pre)
            if noise is None:
                b, _, h, w = out.shape
                p0, p1 = conv.blur.pad
                kh, kw = conv.blur.kernel.shape
                noise = out.new_empty(b, 1, h + p0 + p1 - kh + 1, w + p0 + p1 - kw + 1).normal_()
            act = self.activate
            return upfirdn2d_bias_act(out, conv.blur.kernel, conv.blur.pad, act.bias, noise, self.noise.weight, act.negative_slope, act.scale)
        out = conv(input, style)
        if noise is None:
            b, _, h, w = out.shape
            noise = out.new_empty(b, 1, h, w).normal_()
        return self.activate(out, noise, self.noise.weight)


class ToRGB(nn.Module):
    """Reference: gan_model.py:411-435."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1], out_channels=3,
                 conv_transpose=False, overwrite_negative_padding=None):
        super().__init__()
        if overwrite_negative_padding is not None:
            raise NotImplementedError('ToRGB: overwrite_negative_padding (model_mode 896) is not built')
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, out_channels, 1, style_dim, demodulate=False, conv_transpose=conv_transpose)
        self.bias = nn.Parameter(torch.zeros(1, out_channels, 1, 1))

    def forward(self, input, style, skip=None, fork=False, mod=None):
        """fork=True returns (rgb, input') where input' is `input` for its second consumer (the next up-sampling layer): that
        consumer's gradient is then added inside this layer's input-gradient kernel instead of by a separate pass."""
        conv = self.conv
        if not _FUSE_EPILOGUE:
            out = conv(input, style) + self.bias
            if skip is not None:
                out = out + self.upsample(skip)
            return (out, input) if fork else out
        # 1x1 modulated conv + bias + up-sampled skip in ONE launch: the two adds run in the convolution's epilogue
        up = self.upsample(skip) if skip is not None else None
        return modulated_conv2d(input, conv.weight, mod[0] if mod is not None else conv.modulation(style), demodulate=conv.demodulate, padding=conv.padding,
                                bias=self.bias, residual=up, fork=fork)


class MultiFcStack(nn.Module):
    """Per-group mapping networks (gan_model.py:489-502); groups come from FcConfig."""

    def __init__(self, fc_dict, fc_config):
        super().__init__()
        self.fc_config = fc_config
        for name in fc_config.in_order_group_names:
            setattr(self, name, fc_dict[name])

    def forward(self, x):
        parts = []
        for name in self.fc_config.in_order_group_names:
            lo, hi = self.fc_config.groups[name]['latent_place']
            parts.append(getattr(self, name)(x[:, lo:hi]))
        return torch.cat(parts, dim=1)


class Generator(nn.Module):
    """Reference: gan_model.py:505-811."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 out_channels=3, vae=False, bottleneck_size=256, split_fc=False, marge_fc=False, fc_config=None,
                 conv_transpose=False, model_mode='normal', noise_mode='normal'):
        super().__init__()
        if vae or model_mode != 'normal':
            raise NotImplementedError('Generator: vae / model_mode != "normal" are unreachable from the trainer and not built')
        self.noise_mode, self.model_mode, self.size, self.vae = noise_mode, model_mode, size, vae
        self.out_channels, self.fc_config, self.style_dim = out_channels, fc_config, style_dim

        if split_fc:
            self.style = self.make_fc_stacks_using_fc_config(fc_config, lr_mlp, n_mlp)
        elif marge_fc:
            self.style = nn.Sequential(self.make_fc_stacks_using_fc_config(fc_config, lr_mlp, int(math.ceil(n_mlp / 2))),
                                       self.create_regular_fc_stack(lr_mlp, int(math.floor(n_mlp / 2)), style_dim))
        else:
            self.style = self.create_regular_fc_stack(lr_mlp, n_mlp, style_dim)

        self.channels = channel_table(channel_multiplier)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        ch4 = self.channels[4]
        self.input = ConstantInput(ch4)
        self.conv1 = StyledConv(ch4, ch4, 3, style_dim, blur_kernel=blur_kernel, conv_transpose=conv_transpose, noise_mode=noise_mode)
        self.to_rgb1 = ToRGB(ch4, style_dim, upsample=False, out_channels=out_channels, conv_transpose=conv_transpose)
        self.convs, self.upsamples, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        in_ch = ch4
        for i in range(3, self.log_size + 1):
            out_ch = self.channels[2 ** i]
            self.convs.append(StyledConv(in_ch, out_ch, 3, style_dim, upsample=True, blur_kernel=blur_kernel,
                                         conv_transpose=conv_transpose, noise_mode=noise_mode))
            self.convs.append(StyledConv(out_ch, out_ch, 3, style_dim, blur_kernel=blur_kernel, conv_transpose=conv_transpose))
            self.to_rgbs.append(ToRGB(out_ch, style_dim, out_channels=out_channels, conv_transpose=conv_transpose))
            in_ch = out_ch

    # -- mapping networks ---------------------------------------------------------------------
    @staticmethod
    def create_fc_stack(lr_mlp, n_mlp, style_dim, mid_dim=None):
        dims = [style_dim] + [mid_dim if mid_dim is not None else style_dim] * (n_mlp - 1) + [style_dim]
        return nn.Sequential(PixelNorm(), *[EqualLinear(dims[i], dims[i + 1], lr_mul=lr_mlp, activation='fused_lrelu')
                                            for i in range(n_mlp)])

    def create_regular_fc_stack(self, lr_mlp, n_mlp, style_dim):
        return self.create_fc_stack(lr_mlp, n_mlp, style_dim)

    def make_fc_stacks_using_fc_config(self, fc_config, lr_mlp, n_mlp):
        stacks = {name: self.create_fc_stack(lr_mlp, n_mlp, fc_config.groups[name]['latent_size'], mid_dim=256)
                  for name in fc_config.in_order_group_names}
        return MultiFcStack(stacks, fc_config)

    def load_transfer_learning_model(self, transfer_learning_model, load_only_main=True):
        """Reference: gan_model.py:645-656 -- only the mapping network may differ."""
        missing, unexpected = self.load_state_dict(transfer_learning_model.state_dict(), strict=False)
        if (missing or unexpected) and not load_only_main:
            self.load_state_dict(transfer_learning_model.state_dict())
        for key in list(missing) + list(unexpected):
            if key.split('.')[0] != 'style':
                raise ValueError('key %s is part of the main network' % key)

    # -- helpers --------------------------------------------------------------------------------
    def make_noise(self, batch_size=1, device=None):
        device = device if device is not None else self.input.input.device
        noises = [torch.randn(batch_size, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(batch_size, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    # -- forward --------------------------------------------------------------------------------
    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, return_grad=False):
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
        if noise is None:
            noise = self._draw_noise(styles[0].shape[0], styles[0].device) if randomize_noise else \
                [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            latent = styles[0].unsqueeze(1).repeat(1, self.n_latent, 1) if styles[0].ndim < 3 else styles[0]
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

        out = self.input(latent)
        # one unbind instead of 26 slices: its backward is a single stack, a slice's backward is a zero-fill + add of the whole latent
        lat = latent.unbind(1)
        mods = self._style_path_grouped(latent) if _FUSED_STYLE else None
        if mods is None:
            mods = self._style_path(lat)
        out = self.conv1(out, lat[0], noise=noise[0], mod=mods(0))
        # every StyledConv output feeds ToRGB and the next up-sampling layer: ToRGB forks it (see ToRGB.forward)
        rgb = (lambda m, x, w, sk, md: m(x, w, sk, fork=True, mod=md)) if _FORK_TORGB else (lambda m, x, w, sk, md: (m(x, w, sk, mod=md), x))
        skip, out = rgb(self.to_rgb1, out, lat[1], None, mods(1))
        i, j = 1, 2
        for up_conv, conv, n1, n2, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2], self.to_rgbs):
            out = up_conv(out, lat[i], noise=n1, mod=mods(j))
            out = conv(out, lat[i + 1], noise=n2, mod=mods(j + 1))
            skip, out = rgb(to_rgb, out, lat[i + 2], skip, mods(j + 2))
            i += 2
            j += 3
        image = skip
        if return_grad:
            return image, self.g_path_regularize_grad(image, latent)
        return image, (latent if return_latents else None)

    def _draw_noise(self, batch, device):
        """The per-layer noise maps of one forward pass (NoiseInjection draws `image.new_empty(b, 1, h, w).normal_()` per layer,
        gan_model.py:340-345) as ONE normal_() over a flat buffer and 17 views of it: one launch instead of one per layer."""
        sizes = [4 * 4] + [(2 ** res) ** 2 for res in range(3, self.log_size + 1) for _ in range(2)]
        flat = torch.empty(batch * sum(sizes), device=device, dtype=self.input.input.dtype).normal_()
        out, o = [], 0
        for n in sizes:
            side = int(math.isqrt(n))
            out.append(flat[o:o + batch * n].view(batch, 1, side, side))
            o += batch * n
        return out

    def _style_layers(self):
        """[(ModulatedConv2d, index of the latent it reads)] in execution order: conv1, to_rgb1, then (up-sampling conv, conv, to_rgb) per resolution."""
        layers, idx = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)], 1
        for up_conv, conv, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            layers += [(up_conv.conv, idx), (conv.conv, idx + 1), (to_rgb.conv, idx + 2)]
            idx += 2
        return layers

    def _style_path_grouped(self, latent):
        """(s, d) of every modulated convolution from FIVE launches: gather the latents per layer (a one-hot GEMM), all modulations
        (grouped_linear), square, all demodulation sums (grouped_linear), rsqrt.  Returns ``mods(j)`` as _style_path does, or None when
        the shapes are not the grouped kernels' (style_dim or a channel count not a multiple of 4)."""
        plan = _STYLE_PLANS.get(self)
        if plan is None:
            plan = _STYLE_PLANS[self] = _StylePlan(self._style_layers(), self.style_dim)
        if not plan.ok or latent.dim() != 3 or latent.dtype != torch.float32:
            return None
        b = latent.shape[0]
        if latent.shape[1] <= max(plan.latent_index):
            return None
        x = _gather_latents(plan, latent)      # [groups, B * style_dim]
        s_flat = style_op.grouped_linear(x.reshape(-1), b, plan.mod_plan, [m.modulation.weight for m in plan.layers],
                                         [m.modulation.bias for m in plan.layers])
        s_blocks = style_op.blocks(s_flat, b, plan.s_cols)
        d_blocks = []
        if plan.n_demod:
            u = s_flat[:b * plan.demod_in_cols].square()
            wsq = weight_sq_all([m.weight for m in plan.layers[:plan.n_demod]])
            q = style_op.grouped_linear(u, b, plan.demod_plan, wsq, [_eps_vector(latent, m.out_channel, m.eps) for m in plan.layers[:plan.n_demod]])
            d_blocks = style_op.blocks(q.rsqrt(), b, plan.d_cols)

        def mods(j):
            g = plan.group_of_exec[j]
            return s_blocks[g], (d_blocks[g] if g < plan.n_demod else None)
        return mods

    def _style_path(self, lat):
        """(s, d) of every modulated convolution in execution order, one GEMM call per layer (GANCONTROL_FUSED_STYLE=0, and shapes the grouped
        kernels do not take).  Returns ``mods(j)``: the pair of layer j.  (Round 2 also ran this on a side stream; measured no gain
        -- DESIGN.md section 6 -- and removed in round 3 together with its stream-safety caveats.)"""
        pairs = [m.styles(lat[i]) for m, i in self._style_layers()]
        return lambda j: pairs[j]

    @staticmethod
    def g_path_regularize_grad(fake_img, latents, dim_1_shape=1, pl_noise=None):
        """Reference: gan_model.py:803-811.  ``pl_noise`` (optional) makes the draw reproducible."""
        if pl_noise is None:
            pl_noise = torch.randn_like(fake_img)
        pl_noise = pl_noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3] * dim_1_shape)
        with _backend.activation_grads_only():          # only d/d(latents) is wanted: no weight / bias gradients in this backward
            grad, = autograd.grad(outputs=(fake_img * pl_noise).sum(), inputs=latents, create_graph=True)
        return grad


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU]  (gan_model.py:844-890); child indices fix the state_dict keys."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True, activate=True):
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
        self.padding = 0 if downsample else kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding,
                                  stride=2 if downsample else 1, bias=bias and not activate))
        if activate:
            if not bias:
                raise NotImplementedError('ConvLayer: activation without bias (ScaledLeakyReLU) is not built')
            layers.append(FusedLeakyReLU(out_channel))
        super().__init__(*layers)
        # Blur followed by a stride-2 1x1 conv only ever reads the even blur outputs: decimate INSIDE the FIR
        # (upfirdn2d down=2, the same taps at the same positions) and run the 1x1 conv at stride 1 on a quarter
        # of the pixels.  Same values as gan_model.py:844-890, a quarter of the FIR output / conv input traffic.
        self._decimating_fir = downsample and kernel_size == 1
        self._has_blur, self._activate = downsample, activate

    def forward(self, input, out_gain=1.0, residual=None, fork=False, grad_premasked=False, input_act=None):
        """grad_premasked / input_act: the two ends of ResBlock's conv1 -> Blur pairing (op/upfirdn2d.py::_BlurOfActivation): conv1 is
        called with grad_premasked=True, conv2 with input_act=(negative_slope, gain) of conv1's activation.
        out_gain multiplies the layer's output; it is folded into the activation gain / the weight scale (no extra pass).
        residual (activation-free layers) is added in the convolution's epilogue.  fork=True returns (out, input') where
        input' is the input for its second consumer: that consumer's gradient is then added inside this layer's
        input-gradient convolution instead of by a separate elementwise pass."""
        out, idx = input, 0
        if self._has_blur:
            blur, idx = self[0], 1
            # the (H + 1)-wide Blur output goes straight into this layer's stride-2 convolution, which reads a row pitch
            if self._decimating_fir:
                out = upfirdn2d(out, blur.kernel, down=2, pad=blur.pad)
            elif input_act is not None:
                with _backend.pitched_outputs(True):
                    out = blur_of_activation(out, blur.kernel, blur.pad, *input_act)
            else:
                out = upfirdn2d(out, blur.kernel, pad=blur.pad, _internal=True)
        conv = self[idx]
        stride = 1 if self._decimating_fir else conv.stride
        if fork and not (self._activate and _FUSE_EPILOGUE and not self._has_blur):
            raise NotImplementedError('ConvLayer: fork is built for the fused conv + activation layer without Blur only')
        if self._activate and _FUSE_EPILOGUE:
            if residual is not None:
                raise NotImplementedError('ConvLayer: residual after an activation is not built')
            # EqualConv2d -> FusedLeakyReLU in one launch: bias + leaky-ReLU run in the convolution's epilogue
            act = self[idx + 1]
            return conv2d_gradfix.conv2d_bias_act(out, conv.weight, act.bias, stride=stride, padding=conv.padding, weight_scale=conv.scale,
                                                  negative_slope=act.negative_slope, scale=act.scale * out_gain, fork=fork, grad_premasked=grad_premasked)
        if self._activate:
            out = conv2d_gradfix.conv2d(out, conv.weight, bias=conv.bias, stride=stride, padding=conv.padding, weight_scale=conv.scale)
            act = self[idx + 1]
            out = fused_leaky_relu(out, act.bias, act.negative_slope, act.scale * out_gain)
            return out if residual is None else out + residual
        if conv.bias is not None and (out_gain != 1.0 or residual is not None):
            raise NotImplementedError('ConvLayer: out_gain / residual with a plain bias is not built')
        return conv2d_gradfix.conv2d(out, conv.weight, bias=conv.bias, stride=stride, padding=conv.padding, weight_scale=conv.scale * out_gain,
                                     residual=residual)


class ResBlock(nn.Module):
    """Reference: gan_model.py:893-922."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1], overwrite_padding=None):
        super().__init__()
        if overwrite_padding is not None:
            raise NotImplementedError('ResBlock: overwrite_padding (model_mode 896) is not built')
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        # (conv2(conv1(x)) + skip(x)) / sqrt(2) with the 1/sqrt(2) folded into conv2's activation gain and the skip conv's
        # weight scale: one elementwise pass (the add) instead of two
        rs = 1.0 / math.sqrt(2)
        if not _FUSE_EPILOGUE:
            return self.conv2(self.conv1(input), out_gain=rs) + self.skip(input, out_gain=rs)
        # Neither sum of this block is a pass of its own: `out + skip` is the residual epilogue of the skip convolution, and
        # the two gradients of `input` (conv1 path, skip path) meet in the epilogue of conv1's input-gradient convolution
        # (conv1 hands `input` on to the skip branch: fork).
        if _FUSE_BLUR_ADJ:
            # conv1's output feeds conv2's Blur and nothing else: the Blur's adjoint applies conv1's activation mask on its way out
            act1 = self.conv1[1]
            out, forked = self.conv1(input, fork=True, grad_premasked=True)
            return self.skip(forked, out_gain=rs, residual=self.conv2(out, out_gain=rs, input_act=(act1.negative_slope, act1.scale)))
        out, forked = self.conv1(input, fork=True)
        return self.skip(forked, out_gain=rs, residual=self.conv2(out, out_gain=rs))


def minibatch_stddev(x, group_size=4, feat=1):
    """Reference: gan_model.py:1003-1012 (group members are strided by B / group)."""
    b, c, h, w = x.shape
    g = min(b, group_size)
    s = x.view(g, -1, feat, c // feat, h, w)
    s = torch.sqrt(s.var(0, unbiased=False) + 1e-8)
    s = s.mean([2, 3, 4], keepdims=True).squeeze(2)
    return torch.cat([x, s.repeat(g, 1, h, w)], 1)


class Discriminator(nn.Module):
    """Reference: gan_model.py:925-1016 (adversarial head only)."""

    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], in_channels=3,
                 verification=False, verification_res_split=None, model_mode=None):
        super().__init__()
        if verification or model_mode == '896':
            raise NotImplementedError('Discriminator: verification head / model_mode 896 are never enabled by the trainer and not built')
        self.model_mode, self.verification = model_mode, verification
        channels = channel_table(channel_multiplier)
        log_size = int(math.log(size, 2))
        blocks = [ConvLayer(in_channels, channels[size], 1)]
        in_ch = channels[size]
        for i in range(log_size, 2, -1):
            out_ch = channels[2 ** (i - 1)]
            blocks.append(ResBlock(in_ch, out_ch, blur_kernel))
            in_ch = out_ch
        self.convs = nn.Sequential(*blocks)
        self.convs_adv = nn.Sequential()
        self.convs_verification = nn.Sequential()
        self.stddev_group, self.stddev_feat = 4, 1
        self.final_conv = ConvLayer(in_ch + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
                                          EqualLinear(channels[4], 1))

    def forward(self, input):
        out = self.convs(input)
        out = minibatch_stddev(out, self.stddev_group, self.stddev_feat)
        out = self.final_conv(out)
        return self.final_linear(out.view(out.shape[0], -1)), None
