"""The FID feature network (pytorch-fid's InceptionV3) on the HIP kernels -- forward only.

Mirrors the reference's wrapper src/gan_control/fid_utils/inception.py:17-165 (constructor arguments, ``forward`` contract: a list of
the requested blocks' feature maps, ``blocks`` ModuleList and therefore its state_dict keys) over the architecture of
src/gan_control/fid_utils/overwrite_inception.py (BasicConv2d :424-434, InceptionA..E :202-391) with the FID patches of
inception.py:190-311.  Every convolution + BatchNorm + ReLU is one launch of ``gc_conv2d_bn_relu_f32`` writing its slice of the
block's concatenated output (no ``torch.cat``), pooling is ``gc_pool2d_f32`` / ``gc_global_avgpool_f32``, the 299 x 299 bilinear resize
and the ``2 x - 1`` normalisation are ``gc_resize_bilinear_f32``.

The pretrained weights are a download the reference makes at construction (FID_WEIGHTS_URL, inception.py:14); this module never
touches the network: ``load_fid_weights`` takes the state dict of that checkpoint (key names of the un-wrapped Inception3).
Inference only: BatchNorm uses its running statistics, and the module refuses training mode.
"""
import os

import torch
from torch import nn

from ..models.op import _backend
from ..models.op._backend import ConvGeom

# Round 3: the 1 x 1 and 3 x 3 layers (64 % of the network's multiply-adds) run on the MFMA convolution kernels of the training path
# (gc_conv2d_fused_*: split-bf16 products, fp32 accumulate, ~5e-6 per layer) with the folded BatchNorm scale in the weights and
# shift + ReLU in the fused epilogue (slope 0, gain 1); 5 x 5, 1 x 7 / 7 x 1 and 1 x 3 / 3 x 1 stay on gc_conv2d_bn_relu_f32.
# GANCONTROL_INCEPTION_MFMA=0: every layer on the direct kernel, as in round 2.
_MFMA = os.environ.get('GANCONTROL_INCEPTION_MFMA', '1') != '0'

FID_WEIGHTS_URL = 'https://github.com/mseitzer/pytorch-fid/releases/download/fid_weights/pt_inception-2015-12-05-6726825d.pth'


class BasicConv2d(nn.Module):
    """conv (no bias) -> BatchNorm2d(eps=0.001) -> ReLU (overwrite_inception.py:424-434), as one kernel launch.  ``conv`` and ``bn``
    only hold the parameters / running statistics under the reference's names."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)
        self._folded = None
        self._mfma = None

    def _scale_shift(self):
        bn = self.bn
        key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, bn.weight.data_ptr(), bn.weight.device)
        if self._folded is None or self._folded[0] != key:
            scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
            self._folded = (key, scale.contiguous(), (bn.bias.detach() - bn.running_mean * scale).contiguous())
        return self._folded[1], self._folded[2]

    def _mfma_ok(self, x):
        conv = self.conv
        kh, kw = conv.kernel_size
        be = _backend.get()
        return (_MFMA and x.is_cuda and getattr(be, 'name', '') == 'hip' and kh == kw and kh in (1, 3) and conv.stride[0] == conv.stride[1]
                and conv.stride[0] in (1, 2) and conv.padding[0] == conv.padding[1] and conv.in_channels >= 16 and conv.out_channels >= 16)

    def _mfma_weight(self, scale):
        """[kh, kw, K, N] kernel-layout weights with the BatchNorm scale folded in (a frozen Parameter registered with the cache of derived
        weight forms, so the hi / lo packs are made once); rebuilt when the convolution weight or the statistics change."""
        from ..models.op import weight_cache
        w = self.conv.weight
        key = (self._folded[0], w._version, w.data_ptr())
        if self._mfma is None or self._mfma[0] != key:
            w_t = (w.detach() * scale.reshape(-1, 1, 1, 1)).permute(2, 3, 1, 0).contiguous()
            w_t = nn.Parameter(w_t, requires_grad=False)
            weight_cache.register(w_t)
            self._mfma = (key, w_t)
        return self._mfma[1]

    def forward(self, x, out=None, chan_off=0):
        if self.training:
            raise NotImplementedError('BasicConv2d: inference only (BatchNorm running statistics); call .eval()')
        scale, shift = self._scale_shift()
        conv = self.conv
        if self._mfma_ok(x):
            be = _backend.get()
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            oh, ow = (x.shape[2] + 2 * p - k) // s + 1, (x.shape[3] + 2 * p - k) // s + 1
            prev, be.conv_mode = be.conv_mode, 'bf16x3'
            try:
                y = be.conv2d(x.contiguous(), self._mfma_weight(scale), None, None, ConvGeom(k, k, 1, s, p, p, oh, ow), epilogue=(shift, None, None, 0.0, 1.0, True))
            finally:
                be.conv_mode = prev
            if out is None:
                return y
            out[:, chan_off:chan_off + y.shape[1]].copy_(y)
            return out
        return _backend.get().conv2d_bn_relu(x, conv.weight.detach(), scale, shift, conv.stride[0], conv.padding[0], conv.padding[1], True, out, chan_off)


def _pool(x, k, stride, pad, mode, out=None, chan_off=0):
    return _backend.get().pool2d(x, k, stride, pad, mode, out, chan_off)


def _concat_buffer(x, channels, stride=1, k=1, pad=0):
    oh, ow = (x.shape[2] + 2 * pad - k) // stride + 1, (x.shape[3] + 2 * pad - k) // stride + 1
    return x.new_empty((x.shape[0], channels, oh, ow))


class FIDInceptionA(nn.Module):
    """inception.py:190-215 (average pooling without the padded zeros) over overwrite_inception.py:202-237."""

    def __init__(self, in_channels, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 64, 1)
        self.branch5x5_1 = BasicConv2d(in_channels, 48, 1)
        self.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
        self.branch_pool = BasicConv2d(in_channels, pool_features, 1)
        self.out_channels = 64 + 64 + 96 + pool_features

    def forward(self, x):
        out = _concat_buffer(x, self.out_channels)
        self.branch1x1(x, out, 0)
        self.branch5x5_2(self.branch5x5_1(x), out, 64)
        self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), out, 128)
        self.branch_pool(_pool(x, 3, 1, 1, 'avg'), out, 224)
        return out


class InceptionB(nn.Module):
    """overwrite_inception.py:240-266."""

    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3 = BasicConv2d(in_channels, 384, 3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)
        self.in_channels = in_channels

    def forward(self, x):
        out = _concat_buffer(x, 384 + 96 + self.in_channels, stride=2, k=3)
        self.branch3x3(x, out, 0)
        self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)), out, 384)
        _pool(x, 3, 2, 0, 'max', out, 480)
        return out


class FIDInceptionC(nn.Module):
    """inception.py:218-247 over overwrite_inception.py:269-311."""

    def __init__(self, in_channels, channels_7x7):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)

    def forward(self, x):
        out = _concat_buffer(x, 768)
        self.branch1x1(x, out, 0)
        self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)), out, 192)
        self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))), out, 384)
        self.branch_pool(_pool(x, 3, 1, 1, 'avg'), out, 576)
        return out


class InceptionD(nn.Module):
    """overwrite_inception.py:314-343."""

    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)
        self.in_channels = in_channels

    def forward(self, x):
        out = _concat_buffer(x, 320 + 192 + self.in_channels, stride=2, k=3)
        self.branch3x3_2(self.branch3x3_1(x), out, 0)
        self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))), out, 320)
        _pool(x, 3, 2, 0, 'max', out, 512)
        return out


class FIDInceptionE(nn.Module):
    """inception.py:250-311 over overwrite_inception.py:346-391; ``pool`` = 'avg' (E_1: without the padded zeros) or 'max' (E_2)."""

    def __init__(self, in_channels, pool):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 320, 1)
        self.branch3x3_1 = BasicConv2d(in_channels, 384, 1)
        self.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 448, 1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)
        self.pool = pool

    def forward(self, x):
        out = _concat_buffer(x, 2048)
        self.branch1x1(x, out, 0)
        b3 = self.branch3x3_1(x)
        self.branch3x3_2a(b3, out, 320)
        self.branch3x3_2b(b3, out, 704)
        bd = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        self.branch3x3dbl_3a(bd, out, 1088)
        self.branch3x3dbl_3b(bd, out, 1472)
        self.branch_pool(_pool(x, 3, 1, 1, self.pool), out, 1856)
        return out


class _MaxPool(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2) of the wrapper (inception.py:96, 106)."""

    def forward(self, x):
        return _pool(x, 3, 2, 0, 'max')


class _GlobalAvgPool(nn.Module):
    """nn.AdaptiveAvgPool2d(output_size=(1, 1)) (inception.py:129)."""

    def forward(self, x):
        return _backend.get().global_avgpool(x)


class InceptionV3(nn.Module):
    """Reference: inception.py:17-163 -- same constructor arguments, block indices, forward contract and ``blocks.*`` key names."""

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=[DEFAULT_BLOCK_INDEX], resize_input=True, normalize_input=True, requires_grad=False, use_fid_inception=True):
        super().__init__()
        if not use_fid_inception:
            raise NotImplementedError('InceptionV3: only the FID Inception structure is built (use_fid_inception=True)')
        if requires_grad:
            raise NotImplementedError('InceptionV3: forward only (the HIP kernels of this network have no backward)')
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, 'Last possible output block index is 3'
        self.blocks = nn.ModuleList()
        self.blocks.append(nn.Sequential(BasicConv2d(3, 32, 3, stride=2), BasicConv2d(32, 32, 3), BasicConv2d(32, 64, 3, padding=1), _MaxPool()))
        if self.last_needed_block >= 1:
            self.blocks.append(nn.Sequential(BasicConv2d(64, 80, 1), BasicConv2d(80, 192, 3), _MaxPool()))
        if self.last_needed_block >= 2:
            self.blocks.append(nn.Sequential(FIDInceptionA(192, 32), FIDInceptionA(256, 64), FIDInceptionA(288, 64), InceptionB(288),
                                             FIDInceptionC(768, 128), FIDInceptionC(768, 160), FIDInceptionC(768, 160), FIDInceptionC(768, 192)))
        if self.last_needed_block >= 3:
            self.blocks.append(nn.Sequential(InceptionD(768), FIDInceptionE(1280, 'avg'), FIDInceptionE(2048, 'max'), _GlobalAvgPool()))
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    # names of the un-wrapped Inception3 (the FID checkpoint) -> the wrapper's ``blocks.<b>.<i>`` prefixes
    _LAYOUT = (('Conv2d_1a_3x3', 'Conv2d_2a_3x3', 'Conv2d_2b_3x3'), ('Conv2d_3b_1x1', 'Conv2d_4a_3x3'),
               ('Mixed_5b', 'Mixed_5c', 'Mixed_5d', 'Mixed_6a', 'Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'), ('Mixed_7a', 'Mixed_7b', 'Mixed_7c'))

    def load_fid_weights(self, state_dict):
        """Load the pytorch-fid checkpoint (``pt_inception-2015-12-05``: keys of the un-wrapped Inception3, as ``fid_inception_v3`` loads it
        before wrapping, inception.py:166-186); entries of layers this wrapper does not hold (``fc``, unused blocks) are ignored."""
        mapped = {}
        for b, names in enumerate(self._LAYOUT[:self.last_needed_block + 1]):
            for i, name in enumerate(names):
                for k, v in state_dict.items():
                    if k.startswith(name + '.'):
                        mapped['blocks.%d.%d.%s' % (b, i, k[len(name) + 1:])] = v
        return self.load_state_dict(mapped, strict=True)

    def forward(self, inp):
        if self.training:
            raise NotImplementedError('InceptionV3: inference only; call .eval()')
        x = inp
        if self.resize_input or self.normalize_input:
            oh, ow = (299, 299) if self.resize_input else (x.shape[2], x.shape[3])
            mul, add = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
            x = _backend.get().resize_bilinear(x, oh, ow, mul, add)        # F.interpolate(..., align_corners=False) and 2 x - 1 in one pass
        outp = []
        for idx, block in enumerate(self.blocks):
            x = block(x)
            if idx in self.output_blocks:
                outp.append(x)
            if idx == self.last_needed_block:
                break
        return outp
