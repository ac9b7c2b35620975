"""Phase-2 controller network.

Reference: models/controller_model.py:13-53 -- ``FcStack``: ``n_mlp`` EqualLinear layers with fused leaky-ReLU,
in_dim -> mid_dim -> ... -> out_dim.  Same constructor arguments and state_dict keys (``fc_stack.<i>.weight / bias``);
every layer runs as one GEMM call + the HIP bias/activation kernel (models/gan_model.py::EqualLinear).
"""
from torch import nn

from .gan_model import EqualLinear


class FcStack(nn.Module):
    def __init__(self, lr_mlp, n_mlp, in_dim, mid_dim, out_dim):
        super().__init__()
        if n_mlp < 1:
            raise ValueError('FcStack needs at least one layer')
        self.lr_mlp, self.n_mlp, self.in_dim, self.mid_dim, self.out_dim = lr_mlp, n_mlp, in_dim, mid_dim, out_dim
        widths = [in_dim] + [mid_dim] * (n_mlp - 1) + [out_dim]
        if n_mlp == 1:
            widths = [in_dim, mid_dim]        # the reference's single-layer stack ends at mid_dim (controller_model.py:30-36)
        self.fc_stack = nn.Sequential(*[EqualLinear(a, b, lr_mul=lr_mlp, activation='fused_lrelu') for a, b in zip(widths[:-1], widths[1:])])

    def describe(self):
        return 'FcStack: input dim %d, middle dim %d, output dim %d, %d layers, lr_mlp %g' % (
            self.in_dim, self.mid_dim, self.out_dim, self.n_mlp, self.lr_mlp)

    def forward(self, x):
        return self.fc_stack(x)
