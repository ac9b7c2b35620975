"""ORACLE (test infrastructure only; never imported by the product path): the FID InceptionV3 feature network, restated.

Reference: src/gan_control/fid_utils/inception.py:17-165 (the ``InceptionV3`` wrapper: resize to 299x299, 2x - 1, four blocks) over
src/gan_control/fid_utils/overwrite_inception.py (the repository's own copy of torchvision's Inception3: ``BasicConv2d`` :424-434,
``InceptionA`` :202-237, ``InceptionB`` :240-266, ``InceptionC`` :269-311, ``InceptionD`` :314-343, ``InceptionE`` :346-391) with the FID
patches of inception.py:190-311 (average pooling without the padded zeros in A / C / E_1, max pooling in E_2).  Functional over a
state_dict with the wrapper's key names (``blocks.<b>.<i>.<branch>.conv.weight`` / ``.bn.*``), inference mode (running statistics).
Pinned by tests/golden/inception.npz: outputs of the reference's own classes on procedurally generated weights
(oracle/make_golden.py::golden_inception; the weights of the FID network are a download the build container cannot make).
"""
import torch
import torch.nn.functional as F


def basic_conv(sd, p, x, stride=1, padding=0):
    """conv (no bias) -> BatchNorm(eps 1e-3, running statistics) -> ReLU   (overwrite_inception.py:424-434)."""
    x = F.conv2d(x, sd[p + '.conv.weight'], None, stride, padding)
    x = F.batch_norm(x, sd[p + '.bn.running_mean'], sd[p + '.bn.running_var'], sd[p + '.bn.weight'], sd[p + '.bn.bias'], False, 0.0, 0.001)
    return F.relu(x)


def _avg(x):        # the FID patch: TensorFlow's average pool leaves the padded zeros out of the count
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)


def inception_a(sd, p, x):
    """inception.py:190-215 over overwrite_inception.py:204-217."""
    b1 = basic_conv(sd, p + '.branch1x1', x)
    b5 = basic_conv(sd, p + '.branch5x5_2', basic_conv(sd, p + '.branch5x5_1', x), padding=2)
    b3 = basic_conv(sd, p + '.branch3x3dbl_1', x)
    b3 = basic_conv(sd, p + '.branch3x3dbl_2', b3, padding=1)
    b3 = basic_conv(sd, p + '.branch3x3dbl_3', b3, padding=1)
    bp = basic_conv(sd, p + '.branch_pool', _avg(x))
    return torch.cat([b1, b5, b3, bp], 1)


def inception_b(sd, p, x):
    """overwrite_inception.py:240-266."""
    b3 = basic_conv(sd, p + '.branch3x3', x, stride=2)
    bd = basic_conv(sd, p + '.branch3x3dbl_1', x)
    bd = basic_conv(sd, p + '.branch3x3dbl_2', bd, padding=1)
    bd = basic_conv(sd, p + '.branch3x3dbl_3', bd, stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_c(sd, p, x):
    """inception.py:218-247 over overwrite_inception.py:271-288."""
    b1 = basic_conv(sd, p + '.branch1x1', x)
    b7 = basic_conv(sd, p + '.branch7x7_1', x)
    b7 = basic_conv(sd, p + '.branch7x7_2', b7, padding=(0, 3))
    b7 = basic_conv(sd, p + '.branch7x7_3', b7, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_1', x)
    bd = basic_conv(sd, p + '.branch7x7dbl_2', bd, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_3', bd, padding=(0, 3))
    bd = basic_conv(sd, p + '.branch7x7dbl_4', bd, padding=(3, 0))
    bd = basic_conv(sd, p + '.branch7x7dbl_5', bd, padding=(0, 3))
    bp = basic_conv(sd, p + '.branch_pool', _avg(x))
    return torch.cat([b1, b7, bd, bp], 1)


def inception_d(sd, p, x):
    """overwrite_inception.py:314-343."""
    b3 = basic_conv(sd, p + '.branch3x3_2', basic_conv(sd, p + '.branch3x3_1', x), stride=2)
    b7 = basic_conv(sd, p + '.branch7x7x3_1', x)
    b7 = basic_conv(sd, p + '.branch7x7x3_2', b7, padding=(0, 3))
    b7 = basic_conv(sd, p + '.branch7x7x3_3', b7, padding=(3, 0))
    b7 = basic_conv(sd, p + '.branch7x7x3_4', b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def inception_e(sd, p, x, pool):
    """inception.py:250-311 (E_1: average pooling without the padded zeros; E_2: max pooling) over overwrite_inception.py:348-363."""
    b1 = basic_conv(sd, p + '.branch1x1', x)
    b3 = basic_conv(sd, p + '.branch3x3_1', x)
    b3 = torch.cat([basic_conv(sd, p + '.branch3x3_2a', b3, padding=(0, 1)), basic_conv(sd, p + '.branch3x3_2b', b3, padding=(1, 0))], 1)
    bd = basic_conv(sd, p + '.branch3x3dbl_2', basic_conv(sd, p + '.branch3x3dbl_1', x), padding=1)
    bd = torch.cat([basic_conv(sd, p + '.branch3x3dbl_3a', bd, padding=(0, 1)), basic_conv(sd, p + '.branch3x3dbl_3b', bd, padding=(1, 0))], 1)
    bp = _avg(x) if pool == 'avg' else F.max_pool2d(x, kernel_size=3, stride=1, padding=1)
    return torch.cat([b1, b3, bd, basic_conv(sd, p + '.branch_pool', bp)], 1)


def inception_features(sd, inp, output_blocks=(3,), resize_input=True, normalize_input=True):
    """InceptionV3.forward (inception.py:130-163): the feature maps of the requested blocks, ascending."""
    x = inp
    if resize_input:
        x = F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=False)
    if normalize_input:
        x = 2 * x - 1
    out, last = [], max(output_blocks)
    x = basic_conv(sd, 'blocks.0.0', x, stride=2)
    x = basic_conv(sd, 'blocks.0.1', x)
    x = basic_conv(sd, 'blocks.0.2', x, padding=1)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    if 0 in output_blocks:
        out.append(x)
    if last >= 1:
        x = basic_conv(sd, 'blocks.1.0', x)
        x = basic_conv(sd, 'blocks.1.1', x)
        x = F.max_pool2d(x, kernel_size=3, stride=2)
        if 1 in output_blocks:
            out.append(x)
    if last >= 2:
        for i in range(3):
            x = inception_a(sd, 'blocks.2.%d' % i, x)
        x = inception_b(sd, 'blocks.2.3', x)
        for i in range(4, 8):
            x = inception_c(sd, 'blocks.2.%d' % i, x)
        if 2 in output_blocks:
            out.append(x)
    if last >= 3:
        x = inception_d(sd, 'blocks.3.0', x)
        x = inception_e(sd, 'blocks.3.1', x, 'avg')
        x = inception_e(sd, 'blocks.3.2', x, 'max')
        x = F.adaptive_avg_pool2d(x, (1, 1))
        if 3 in output_blocks:
            out.append(x)
    return out


def procedural_inception_fill_(state_dict, salt=0):
    """Weights for the fixtures: values derived from CRC32(key) (the FID weights are an external download).  Convolutions get a
    fan-in scaling so that activations stay O(1) through the 94 layers; BatchNorm statistics are kept positive."""
    import zlib
    for key in sorted(state_dict.keys()):
        t = state_dict[key]
        if not torch.is_floating_point(t):
            continue
        gen = torch.Generator(device='cpu')
        gen.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
        v = torch.randn(t.shape, generator=gen, dtype=torch.float32)
        if key.endswith('conv.weight'):
            v = v * (2.0 / (t.shape[1] * t.shape[2] * t.shape[3])) ** 0.5
        elif key.endswith('bn.weight'):
            v = 1.0 + 0.1 * v
        elif key.endswith('bn.bias') or key.endswith('bn.running_mean'):
            v = 0.1 * v
        elif key.endswith('bn.running_var'):
            v = 0.75 + 0.5 * torch.rand(t.shape, generator=gen)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict
