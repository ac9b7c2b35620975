"""Phase-2 controller training step.

Reference: trainers/controller_trainer.py -- model / optimiser set-up :89-125, ``controller_update`` :202-220,
``calc_latent_rec_adv_loss`` :222-229, ``re_arrange_latent`` :248-252.  A controller maps an attribute vector to the
w sub-latent of one attribute group; it is trained with L1 (or MSE) against the w latents of generated samples.
Built: the ``latent_rec`` objective, the ``attribute_rec`` objective (:231-239: splice the predicted sub-latent into w, run the
FROZEN generator on the HIP kernels, read the attribute off a predictor and compare with the controls -- the gradient reaches the
controller through the generator's input gradients), the Adam set-up with the reference's lazy-regularisation ratio and latent
splicing.  The predictor is a ``losses.LossModelClass`` whose pretrained network is supplied by the caller (external weights,
SURVEY 8f-4).  Not built: ``latent_adv`` (dead code in the reference: :222-229 never sets it), tensorboard / image dumps.
"""
import torch
from torch import nn, optim

from ..models.controller_model import FcStack


def default_controller_config(in_dim=3, mid_dim=256, n_mlp=8, batch=32):
    """Fields of configs/controller_configs/*.json that the step uses."""
    return {'model_config': {'lr_mlp': 0.01, 'n_mlp': n_mlp, 'in_dim': in_dim, 'mid_dim': mid_dim},
            'training_config': {'batch': batch, 'reg_every': 4, 'lr': 0.002, 'rec_loss': 'l1', 'losses': ['latent_rec'],
                                'attribute_rec_w': 1.0}}


class ControllerTrainer:
    def __init__(self, config, group_chunk, device='cuda', generator=None, seed=None, loss_class=None):
        """group_chunk = (begin, end) of the attribute group inside the w latent (batch_utils.place_in_latent_dict[group]);
        loss_class = the attribute's losses.LossModelClass (predict / controller_criterion), needed by ``attribute_rec``."""
        self.config, self.device = config, torch.device(device)
        self.model_config, self.training_config = config['model_config'], config['training_config']
        self.group_chunk = (int(group_chunk[0]), int(group_chunk[1]))
        if 'latent_adv' in self.training_config['losses']:
            raise NotImplementedError('ControllerTrainer: latent_adv is never computed by the reference either (controller_trainer.py:222-229)')
        if 'attribute_rec' in self.training_config['losses'] and (generator is None or loss_class is None):
            raise ValueError('ControllerTrainer: attribute_rec needs a generator and a loss_class (the attribute predictor)')
        self.loss_class = loss_class
        if seed is not None:
            torch.manual_seed(seed)
        mc, tc = self.model_config, self.training_config
        self.fc_controller = FcStack(mc['lr_mlp'], mc['n_mlp'], mc['in_dim'], mc['mid_dim'], self.group_chunk[1] - self.group_chunk[0]).to(self.device)
        ratio = tc['reg_every'] / (tc['reg_every'] + 1)                      # controller_trainer.py:107-113
        self.fc_optim = optim.Adam(self.fc_controller.parameters(), lr=tc['lr'] * ratio, betas=(0 ** ratio, 0.99 ** ratio))
        self.rec_loss = nn.L1Loss() if tc.get('rec_loss', 'l1') == 'l1' else nn.MSELoss()
        self.generator = generator.eval().to(self.device) if generator is not None else None
        if self.generator is not None:
            for prm in self.generator.parameters():
                prm.requires_grad_(False)
        self.evaluation_dict = {}

    def calc_latent_rec_loss(self, org_latent, pred_latent):
        lo, hi = self.group_chunk
        return self.rec_loss(pred_latent, org_latent[:, lo:hi])

    def controller_update(self, batch):
        """One optimisation step on (controls, w latents); returns the loss value (also in evaluation_dict)."""
        controls = batch[0].to(self.device).float()
        org_latent = batch[1].to(self.device)
        self.fc_controller.train()
        self.fc_controller.zero_grad()
        pred_latent = self.fc_controller(controls.detach())
        tc = self.training_config
        loss = 0.
        if 'latent_rec' in tc['losses']:
            rec = self.calc_latent_rec_loss(org_latent, pred_latent)
            self.evaluation_dict['latent_rec_loss'] = rec.item()
            loss = loss + rec
        if 'attribute_rec' in tc['losses']:
            att = self.calc_attribute_rec_controller_loss(org_latent, pred_latent, controls)
            self.evaluation_dict['attribute_loss'] = att.item()
            loss = loss + att * tc['attribute_rec_w']
        self.evaluation_dict['loss'] = loss.item()
        loss.backward()
        self.fc_optim.step()
        return self.evaluation_dict['loss']

    def calc_attribute_rec_controller_loss(self, org_latent, pred_latent, controls):
        """controller_trainer.py:231-239."""
        latent = self.re_arrange_latent(org_latent, pred_latent)
        fake_img, _ = self.generator([latent], input_is_latent=True)
        pred = self.loss_class.predict(fake_img)
        return self.loss_class.controller_criterion(pred, controls)

    def re_arrange_latent(self, org_latent, group_latent):
        """w latent with the group's slice replaced by the controller output (controller_trainer.py:248-252)."""
        lo, hi = self.group_chunk
        latent = org_latent.clone()
        latent[:, lo:hi] = group_latent
        return latent

    @torch.no_grad()
    def generate(self, org_latent, controls):
        """Images of the frozen generator for latents whose group slice comes from the controller."""
        if self.generator is None:
            raise RuntimeError('ControllerTrainer.generate needs a generator')
        self.fc_controller.eval()
        latent = self.re_arrange_latent(org_latent.to(self.device), self.fc_controller(controls.to(self.device).float()))
        return self.generator([latent], input_is_latent=True)[0]
