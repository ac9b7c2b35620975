"""Phase-2 controller: FcStack and the latent-reconstruction step against vectors produced by the
reference's own FcStack + Adam set-up (tests/golden/controller.npz), the oracle against the same vectors, the
DataFrame dataset, and latent splicing."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden, rel_err

from gan_control_amd.datasets import DataFrameDataSet, get_dataframe_data_loader
from gan_control_amd.models.controller_model import FcStack
from gan_control_amd.trainers.controller_trainer import ControllerTrainer, default_controller_config
from oracle import controller as octl

GOLD = load_golden('controller')
LR_MLP, N_MLP, IN_DIM, MID, OUT = float(GOLD['hyper'][0]), *(int(v) for v in GOLD['hyper'][1:])
CHUNK = tuple(int(v) for v in GOLD['chunk'])


def _state(prefix, dtype):
    return {k.split('/', 1)[1]: torch.from_numpy(v).to(dtype) for k, v in GOLD.items() if k.startswith(prefix + '/')}


def _trainer(device, dtype=torch.float32):
    cfg = default_controller_config(IN_DIM, MID, N_MLP, batch=6)
    tr = ControllerTrainer(cfg, CHUNK, device=device, seed=0)
    tr.fc_controller.load_state_dict(_state('init', dtype))
    return tr


def test_oracle_matches_reference_vectors():
    init = _state('init', torch.float64)
    ws = [init[f'fc_stack.{i}.weight'].clone().requires_grad_(True) for i in range(N_MLP)]
    bs = [init[f'fc_stack.{i}.bias'].clone().requires_grad_(True) for i in range(N_MLP)]
    x, w = torch.from_numpy(GOLD['controls']), torch.from_numpy(GOLD['w_latent'])
    assert rel_err(octl.fc_stack_forward(x, ws, bs, LR_MLP), torch.from_numpy(GOLD['forward'])) < 1e-12
    losses = octl.controller_step(ws, bs, LR_MLP, x, w, CHUNK, steps=3)
    assert np.allclose(losses, GOLD['losses'], rtol=1e-12)
    after = _state('after3', torch.float64)
    for i in range(N_MLP):
        assert rel_err(ws[i], after[f'fc_stack.{i}.weight']) < 1e-10
        assert rel_err(bs[i], after[f'fc_stack.{i}.bias']) < 1e-10


def test_fc_stack_layout_follows_reference():
    net = FcStack(LR_MLP, N_MLP, IN_DIM, MID, OUT)
    want = _state('init', torch.float32)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in want.items()}
    assert [tuple(m.weight.shape) for m in FcStack(0.01, 1, 3, 16, 8).fc_stack] == [(16, 3)]
    assert [tuple(m.weight.shape) for m in FcStack(0.01, 2, 3, 16, 8).fc_stack] == [(16, 3), (8, 16)]
    with pytest.raises(ValueError):
        FcStack(0.01, 0, 3, 16, 8)


def _check_step(tr, tol):
    x, w = torch.from_numpy(GOLD['controls']).float(), torch.from_numpy(GOLD['w_latent']).float()
    with torch.no_grad():
        y = tr.fc_controller(x.to(tr.device))
    assert rel_err(y.cpu(), torch.from_numpy(GOLD['forward'])) < tol
    losses = [tr.controller_update((x, w)) for _ in range(3)]
    assert np.allclose(losses, GOLD['losses'], rtol=tol)
    after = _state('after3', torch.float64)
    for k, v in tr.fc_controller.state_dict().items():
        assert rel_err(v.cpu(), after[k]) < tol, k
    assert tr.evaluation_dict['latent_rec_loss'] == losses[-1]


def test_controller_update_emulated(emu_backend):
    _check_step(_trainer('cpu'), 1e-3)          # Adam's first steps are sign-like: fp32 vs the f64 vectors


@pytest.mark.gpu
def test_controller_update_hip():
    _check_step(_trainer('cuda'), 1e-3)


def test_re_arrange_latent_and_loss_choices(emu_backend):
    tr = _trainer('cpu')
    org = torch.randn(5, 96)
    grp = torch.randn(5, CHUNK[1] - CHUNK[0])
    out = tr.re_arrange_latent(org, grp)
    assert torch.equal(out[:, CHUNK[0]:CHUNK[1]], grp) and torch.equal(out[:, :CHUNK[0]], org[:, :CHUNK[0]]) and torch.equal(out[:, CHUNK[1]:], org[:, CHUNK[1]:])
    assert out.data_ptr() != org.data_ptr()
    cfg = default_controller_config(IN_DIM, MID, N_MLP)
    cfg['training_config']['rec_loss'] = 'mse'
    mse = ControllerTrainer(cfg, CHUNK, device='cpu', seed=0)
    assert isinstance(mse.rec_loss, torch.nn.MSELoss)
    cfg['training_config']['losses'] = ['latent_rec', 'attribute_rec']
    with pytest.raises(NotImplementedError):
        ControllerTrainer(cfg, CHUNK, device='cpu')
    with pytest.raises(RuntimeError):
        tr.generate(org, torch.randn(5, IN_DIM))


def test_optimizer_follows_lazy_reg_ratio(emu_backend):
    tr = _trainer('cpu')
    grp = tr.fc_optim.param_groups[0]
    assert grp['lr'] == pytest.approx(0.002 * 0.8) and grp['betas'] == (0.0, pytest.approx(0.99 ** 0.8))


def _frame(n=40):
    rng = np.random.default_rng(0)
    return pd.DataFrame({'latents_w': [rng.standard_normal(16).astype(np.float32) for _ in range(n)],
                         'age': rng.uniform(10, 80, n).astype(np.float32),
                         'orientation': [rng.standard_normal(3).astype(np.float32) for _ in range(n)],
                         'expression_q': rng.integers(0, 8, n)})


def test_dataframe_dataset_split_and_attribute_shapes(tmp_path):
    frame = _frame()
    path = tmp_path / 'frame.pkl'
    frame.to_pickle(path)
    train, evals = DataFrameDataSet(str(path), 'age', train=True), DataFrameDataSet(str(path), 'age', train=False)
    assert (len(train), len(evals)) == (36, 4)
    a, w = train[3]
    assert a.shape == (1,) and float(a) == pytest.approx(float(frame.age[3])) and torch.equal(w, torch.from_numpy(frame.latents_w[3]))
    a, w = evals[0]
    assert float(a) == pytest.approx(float(frame.age[36]))
    a, _ = DataFrameDataSet(frame, 'orientation')[0]
    assert a.shape == (3,)
    a, _ = DataFrameDataSet(frame, 'expression_q')[5]
    assert a.shape == (8,) and int(a.argmax()) == int(frame.expression_q[5]) and int(a.sum()) == 1
    loader = get_dataframe_data_loader(frame, 'orientation', batch_size=8, workers=0)
    ctl, lat = next(iter(loader))
    assert ctl.shape == (8, 3) and lat.shape == (8, 16) and len(loader) == 4        # drop_last


def test_loader_feeds_the_step(emu_backend):
    frame = _frame(64)
    mix = np.random.default_rng(1).standard_normal((3, 16)).astype(np.float32)
    frame['latents_w'] = [np.maximum(o @ mix, 0.0) for o in frame.orientation]          # a learnable attribute -> latent relation
    loader = get_dataframe_data_loader(frame, 'orientation', batch_size=16, workers=0, shuffle=False)
    cfg = default_controller_config(3, 32, 3, batch=16)
    cfg['training_config']['lr'] = 0.2          # parameters live at 1 / lr_mlp scale: the reference's 0.002 needs thousands of steps
    tr = ControllerTrainer(cfg, (4, 12), device='cpu', seed=1)
    epochs = [float(np.mean([tr.controller_update(batch) for batch in loader])) for _ in range(40)]
    assert epochs[-1] < 0.8 * epochs[0]
