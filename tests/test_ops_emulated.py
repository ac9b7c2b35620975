"""Host-side logic of the operator socket, validated on the CPU over the emulated C ABI.

These tests exercise the PRODUCT's autograd layer (gan_control_amd.models.op and the modules
in gan_control_amd.models.gan_model): gradient closure to second order, geometry, None-vs-zero
gradient structure.  The kernels themselves are covered by the ``-m gpu`` tests.
"""
import pytest
import torch

import op_checks as oc


@pytest.mark.parametrize('case', oc.names(oc.UPF))
def test_upfirdn2d(case, emu_backend):
    oc.check_upfirdn2d(case, 'cpu')


@pytest.mark.parametrize('case', oc.names(oc.BA))
def test_bias_act(case, emu_backend):
    oc.check_bias_act(case, 'cpu')


def test_noise_bias_act(emu_backend):
    oc.check_noise_bias_act('cpu')


@pytest.mark.parametrize('case', oc.names(oc.CV, 'conv_'))
def test_equal_conv(case, emu_backend):
    oc.check_equal_conv(case, 'cpu')


@pytest.mark.parametrize('case', oc.names(oc.CV, 'mod_'))
def test_modulated_conv(case, emu_backend):
    oc.check_modulated_conv(case, 'cpu')


def test_conv_functional(emu_backend):
    oc.conv_functional_checks('cpu')


@pytest.mark.parametrize('size', [32])
def test_network(size, emu_backend):
    oc.check_network(size, 'cpu')


def test_no_cpu_fallback():
    """Without the emulation the product refuses CPU tensors instead of silently falling back."""
    from gan_control_amd.models.op import upfirdn2d, fused_leaky_relu
    with pytest.raises(RuntimeError):
        upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(2, 2))
    with pytest.raises(RuntimeError):
        fused_leaky_relu(torch.zeros(1, 2, 4, 4), torch.zeros(2))


def test_split_fc(emu_backend):
    oc.check_split_fc('cpu')


def test_mixing_truncation_and_stored_noise(emu_backend):
    oc.check_mixing_truncation('cpu')


def test_transfer_learning_load(emu_backend):
    oc.check_transfer_learning('cpu')


def test_misc_modules(emu_backend):
    oc.check_misc('cpu')


@pytest.mark.parametrize('name', ['ffhq', 'metfaces', 'afhq'])
def test_trainer_from_shipped_config(name, emu_backend):
    oc.check_config_ingestion('cpu', name, size=16, batch=4)
