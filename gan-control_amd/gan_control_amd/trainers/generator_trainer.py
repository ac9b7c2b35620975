"""The G+D training step of gan-control, data-parallel over the GPUs of one node.

Reproduces the step maths of the reference's GeneratorTrainer
(src/gan_control/trainers/generator_trainer.py) with ``vanilla`` semantics (no predictor
losses -- those need pretrained networks that are not part of the hot path):

  train loop          :329-355      discriminator_update :626-643   generator_update :357-369
  discriminator_step  :645-688      discriminator_regularize_step :697-711   d_r1_loss :713-719
  generator_step      :407-436      generator_regularize_step :568-599
  g_path_regularize_grad :617-624   Adam set-up :161-173            dry_run :301-327

including the quirks listed in SURVEY.md Appendix C (D loss divided by the mini-batch image
count, regularisers firing at i = 0, ``+ 0 * pred[0]`` terms, name sets for set_grad_none).
Multi-GPU is one process per GPU with RCCL gradient all-reduce (trainers/ddp.py) instead of
nn.DataParallel; the config dictionary keeps the reference's JSON schema (configs/ffhq.json).
"""
import copy
import math
import os

import torch
from torch import autograd, optim
from torch.nn import functional as F

from ..models.gan_model import Generator, Discriminator
from ..models.op import _backend
from ..utils.fc_config import fc_config_from_sub_groups
from . import ddp
from .utils import accumulate, requires_grad, mixing_noise, make_mini_batch_from_noise, set_grad_none, named_params, zero_grad_none
from .non_leaking import augment, AdaptiveAugmentState


def default_config(size=512, batch=16):
    """The hot-path fields of configs/ffhq.json:5-84 (vanilla, regular mapping network)."""
    return {
        'model_config': {'vanilla': True, 'img_channels': 3, 'split_fc': False, 'marge_fc': False, 'latent_size': 512,
                         'size': size, 'n_mlp': 8, 'channel_multiplier': 2.0, 'conv_transpose': True, 'g_noise_mode': 'normal'},
        'training_config': {'parallel_grad_regularize_step': True, 'iter': 800000, 'start_iter': 0, 'batch': batch,
                            'mini_batch': batch, 'augment': {'enabled': False, 'ada_target': 0.6, 'ada_length': 500000, 'p': 0},
                            'r1': 1, 'd_every': 1, 'g_reg_every': 4, 'd_reg_every': 16, 'lr_g': 0.002, 'lr_d': 0.002,
                            'g_moving_average': 10000, 'path_regularize': 2, 'path_batch_shrink': 2, 'mixing': 0,
                            'parallel': True, 'sub_groups_dict': None},
    }


def none_grad_names(generator, discriminator, latent_size=512, img_channels=3, size=None):
    """The reference's dry_run (:301-327), run on the product's own graph: one latent through the path-length
    regulariser, one random image through R1, then collect the parameters whose ``grad is None``.

    The reference's ATen graph reaches a parameter "through a mask" (leaky-ReLU's backward reads its input to pick
    the slope) with an exactly-zero gradient, while the fused ops here cut such paths (their mask comes from the
    output).  ``strict_zero_grads`` makes the activation backward hand on zero tensors for those paths during the dry
    run, so ``grad is None`` means what it means in the reference: the parameter is not part of the double-backward
    graph at all (every ToRGB bias of G, the last linear bias of D: SURVEY.md Appendix C #5; pinned against the
    reference's own dry run by tests/golden/step*.npz).
    """
    dev = next(generator.parameters()).device
    size = size if size is not None else generator.size
    saved = [(p, p.grad, p.requires_grad) for m in (generator, discriminator) for p in m.parameters()]
    rng = torch.get_rng_state()
    cuda_rng = torch.cuda.get_rng_state(dev) if dev.type == 'cuda' else None
    try:
        for p, _, _ in saved:
            p.grad = None
            p.requires_grad_(True)
        with _backend.strict_zero_grads():
            test_in = torch.randn(1, latent_size, device=dev)
            fake, latent = generator([test_in], return_latents=True)
            noise = torch.randn_like(fake) / math.sqrt(fake.shape[2] * fake.shape[3])
            grad, = autograd.grad((fake * noise).sum(), latent, create_graph=True)             # g_path_regularize :601-615
            GeneratorTrainer.g_path_regularize_grad(grad, 0)[0].backward()
            none_g = {n for n, p in generator.named_parameters() if p.grad is None}
            test_img = torch.randn(1, img_channels, size, size, device=dev, requires_grad=True)
            pred, _ = discriminator(test_img)
            grad_real, = autograd.grad(pred.sum(), test_img, create_graph=True)                # d_r1_loss :713-719
            grad_real.pow(2).reshape(1, -1).sum(1).mean().backward()
            none_d = {n for n, p in discriminator.named_parameters() if p.grad is None}
    finally:
        for p, g, r in saved:
            p.grad = g
            p.requires_grad_(r)
        torch.set_rng_state(rng)
        if cuda_rng is not None:
            torch.cuda.set_rng_state(cuda_rng, dev)
    return none_g, none_d


def unused_parameter_names(generator, latent_size=512):
    """Parameters the generator's forward does not read at all in its configuration -- the ``*.noise.weight`` strengths of the layers built
    with noise_mode 'zeros' / 'id_zeros' (gan_model.py:391-399): a plain backward leaves them ``None`` in the reference, so Adam never creates
    state for them.  Found the way the dry run finds its sets: one latent, one plain backward."""
    dev = next(generator.parameters()).device
    saved = [(p, p.grad, p.requires_grad) for p in generator.parameters()]
    rng = torch.get_rng_state()
    cuda_rng = torch.cuda.get_rng_state(dev) if dev.type == 'cuda' else None
    try:
        for p, _, _ in saved:
            p.grad = None
            p.requires_grad_(True)
        with _backend.strict_zero_grads():
            fake, _ = generator([torch.randn(1, latent_size, device=dev)])
            fake.sum().backward()
        return {n for n, p in generator.named_parameters() if p.grad is None}
    finally:
        for p, g, r in saved:
            p.grad = g
            p.requires_grad_(r)
        torch.set_rng_state(rng)
        if cuda_rng is not None:
            torch.cuda.set_rng_state(cuda_rng, dev)


class GeneratorTrainer:
    def __init__(self, config, device='cuda', seed=0, fused_adam=None, fuse_d_pair=True, loss_models=None):
        """loss_models: {config name ('embedding_loss', 'orientation_loss', ...): losses.LossModelClass} -- the attribute /
        contrastive losses of the controllable generator step (generator_trainer.py:407-547), used when ``model_config.vanilla``
        is false.  Their pretrained predictors are external; without loss_models the step is the adversarial one."""
        self.config = config
        self.fuse_d_pair = fuse_d_pair
        self.loss_models = dict(loss_models or {})
        self.model_config = config['model_config']
        self.training_config = config['training_config']
        self.device = torch.device(device)
        self.world = ddp.world_size()
        self.rank = ddp.rank()
        tc = self.training_config
        if tc['batch'] % self.world != 0:
            raise ValueError('global batch %d is not divisible by the world size %d' % (tc['batch'], self.world))
        self.local_batch = tc['batch'] // self.world
        # the reference chunks the global batch into mini-batches on ONE process; here each rank
        # takes its shard of every mini-batch
        if tc['mini_batch'] % self.world != 0 and not self.loss_models:
            raise ValueError('mini_batch %d is not divisible by the world size %d' % (tc['mini_batch'], self.world))
        self.local_mini_batch = tc['mini_batch'] // self.world if not self.loss_models else min(tc['mini_batch'], self.local_batch)
        if self.local_mini_batch >= 4 and self.local_mini_batch % 4 != 0:
            raise ValueError('per-GPU mini-batch must be a multiple of 4 (minibatch-stddev groups, gan_model.py:1005-1011)')
        self.batch_utils = None
        if self.loss_models and not self.model_config.get('vanilla', True):
            # the same / not-same pairs are positions inside ONE mini-batch (mini_batch_multi_split_utils.py:64-69): every rank must
            # run whole mini-batches (last row)
            if self.local_mini_batch != tc['mini_batch']:
                raise ValueError('attribute losses need whole mini-batches per rank: batch / world must be a multiple of mini_batch')
            from ..utils.mini_batch_utils import MiniBatchUtils
            self.batch_utils = MiniBatchUtils(tc['mini_batch'], tc['sub_groups_dict'], total_batch=tc['batch'], latent_size=self.model_config['latent_size'])
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed * 1000 + self.rank)
        torch.manual_seed(seed)          # identical initial weights on every rank
        self.init_models_and_optim(fused_adam)
        self.dry_run()
        # Every rank must draw its OWN injected noise, path-length noise and ADA transforms (the single-process reference
        # draws one independent sample per image of the global batch): re-seed the global CPU / device generators per rank
        # now that the replicas are identical.  Python's ``random`` (style-mixing coin, inject index) stays in lock-step:
        # those are one draw per step for the whole global batch in the reference too.
        torch.manual_seed(seed * 1000 + self.rank + 1)
        self.mean_path_length = torch.zeros((), device=self.device)       # a persistent device scalar, updated in place
        self.ada = AdaptiveAugmentState(tc['augment'], self.device)        # generator_trainer.py:333-339
        self.accum = 0.5 ** (tc['batch'] / tc['g_moving_average'])
        self.stats = {}

    # -- set-up ---------------------------------------------------------------------------------
    def init_models_and_optim(self, fused_adam=None):
        mc, tc = self.model_config, self.training_config
        fc = None
        if mc['split_fc'] or mc['marge_fc']:
            fc = fc_config_from_sub_groups(tc['sub_groups_dict'], mc['latent_size'])
        kw = dict(channel_multiplier=mc['channel_multiplier'], out_channels=mc['img_channels'], split_fc=mc['split_fc'],
                  marge_fc=mc['marge_fc'], fc_config=fc, conv_transpose=mc['conv_transpose'], noise_mode=mc['g_noise_mode'])
        self.generator = Generator(mc['size'], mc['latent_size'], mc['n_mlp'], **kw).to(self.device)
        self.g_ema = Generator(mc['size'], mc['latent_size'], mc['n_mlp'], **kw).to(self.device)
        self.discriminator = Discriminator(mc['size'], channel_multiplier=mc['channel_multiplier'], in_channels=mc['img_channels']).to(self.device)
        self.g_ema.eval()
        from ..models.op import weight_cache
        for m in (self.generator, self.g_ema, self.discriminator):
            ddp.broadcast_module(m)
            # derived weight forms are cached for these three: the optimisers' post-step hook invalidates G and D, accumulate() the EMA
            weight_cache.register(m)
        accumulate(self.g_ema, self.generator, 0)
        g_ratio = tc['g_reg_every'] / (tc['g_reg_every'] + 1)
        d_ratio = tc['d_reg_every'] / (tc['d_reg_every'] + 1)
        if fused_adam is None:
            fused_adam = self.device.type == 'cuda'
        extra = {'fused': True} if fused_adam else {}
        self.g_optim = optim.Adam(self.generator.parameters(), lr=tc['lr_g'] * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio), **extra)
        self.d_optim = optim.Adam(self.discriminator.parameters(), lr=tc['lr_d'] * d_ratio, betas=(0 ** d_ratio, 0.99 ** d_ratio), **extra)
        self.g_module, self.d_module, self.g_ema_module = self.generator, self.discriminator, self.g_ema
        self.g_reducer = ddp.GradientReducer(self.generator)
        self.d_reducer = ddp.GradientReducer(self.discriminator)

    def dry_run(self):
        """generator_trainer.py:301-327.  ``training_config['parallel']`` is NOT honoured here on purpose: with
        nn.DataParallel the reference's names carry a 'module.' prefix and never match in set_grad_none (the ToRGB
        biases then keep a zero gradient and Adam steps them on momentum); this trainer implements the
        single-process semantics the reference intends."""
        mc = self.model_config
        self.none_g_grads, self.none_d_grads = none_grad_names(self.generator, self.discriminator, mc['latent_size'], mc['img_channels'], mc['size'])
        # only a non-default noise mode leaves parameters out of the forward pass (no second dry pass in the default configuration)
        self.unused_g = unused_parameter_names(self.generator, mc['latent_size']) if mc.get('g_noise_mode', 'normal') != 'normal' else set()

    def state_dict(self):
        """Checkpoint layout of save_nets (:852-865) plus the state the reference forgets (Appendix C #4)."""
        return {'g': self.generator.state_dict(), 'd': self.discriminator.state_dict(), 'g_ema': self.g_ema.state_dict(),
                'g_optim': self.g_optim.state_dict(), 'd_optim': self.d_optim.state_dict(),
                'mean_path_length': float(self.mean_path_length)}

    def load_state_dict(self, ckpt):
        self.generator.load_state_dict(ckpt['g'])
        self.discriminator.load_state_dict(ckpt['d'])
        self.g_ema.load_state_dict(ckpt['g_ema'])
        self.g_optim.load_state_dict(ckpt['g_optim'])
        self.d_optim.load_state_dict(ckpt['d_optim'])
        if 'mean_path_length' in ckpt:
            self._mean_path_length_tensor().fill_(float(ckpt['mean_path_length']))

    # -- losses (names and maths of the reference's static methods) ------------------------------
    @staticmethod
    def d_logistic_loss(real_pred, fake_pred):
        return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()

    @staticmethod
    def g_nonsaturating_loss(fake_pred):
        return F.softplus(-fake_pred).mean()

    @staticmethod
    def d_r1_loss(real_pred, real_img):
        with _backend.activation_grads_only():          # only d/d(image) is wanted: no parameter gradients in this backward
            grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
        return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()

    @staticmethod
    def g_path_regularize_grad(grad, mean_path_length, decay=0.01, reduce_mean=None):
        """Path-length penalty (:617-624).  With ``reduce_mean`` (an in-place mean all-reduce) the running
        mean uses the GLOBAL batch mean and the returned penalty carries exactly the gradient the
        single-process formula has on the global batch: the (small) term that flows through
        ``path_lengths.mean()`` is re-expressed with the all-reduced mean as a constant."""
        path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
        if reduce_mean is None:
            path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
            path_penalty = (path_lengths - path_mean).pow(2).mean()
            return path_penalty, path_mean.detach(), path_lengths
        global_mean = reduce_mean(path_lengths.detach().mean())
        path_mean = mean_path_length + decay * (global_mean - mean_path_length)
        path_penalty = (path_lengths - path_mean).pow(2).mean()
        through_mean = 2 * decay * (global_mean - path_mean) * path_lengths.mean()
        path_penalty = path_penalty - (through_mean - through_mean.detach())
        return path_penalty, path_mean.detach(), path_lengths

    # -- helpers --------------------------------------------------------------------------------
    def sample_z(self, batch):
        tc = self.training_config
        return mixing_noise(batch, self.model_config['latent_size'], tc['mixing'], self.device, self.gen)

    @staticmethod
    def _fill_missing_grads(module, none_names):
        """Reference autograd yields ZERO (not None) gradients for parameters that only reach a
        regulariser through an activation mask (Appendix C #6), so Adam still steps them on
        momentum.  The fused ops cut those dead paths; restore the zeros here."""
        for n, p in named_params(module):
            if p.requires_grad and p.grad is None and n not in none_names:
                p.grad = torch.zeros_like(p)

    # -- discriminator ----------------------------------------------------------------------------
    def discriminator_step(self, mini_noise_inputs, mini_real_inputs, noise=None):
        self._discriminator_step(mini_noise_inputs, mini_real_inputs, noise=noise)
        self._ada_update()

    def _discriminator_step(self, mini_noise_inputs, mini_real_inputs, noise=None):
        self.stats['d_loss'] = 0
        zero_grad_none(self.discriminator)
        n = len(mini_real_inputs)
        for k, (real, z) in enumerate(zip(mini_real_inputs, mini_noise_inputs)):
            self.d_reducer.begin(sync=(k == n - 1), phase='d')
            fake_img, _ = self.generator(z, noise=noise)
            if self.training_config['augment']['enabled']:                       # generator_trainer.py:651-653
                real, _ = augment(real, self.ada.p)
                fake_img, _ = augment(fake_img, self.ada.p)
            fake_pred, real_pred = self.discriminate_pair(fake_img, real)
            d_loss = self.d_logistic_loss(real_pred, fake_pred)
            # reference divides by the number of IMAGES in the (global) mini-batch (:658)
            d_loss = d_loss / (len(real) * self.world)
            self.stats['d_loss'] = self.stats['d_loss'] + d_loss.detach()
            d_loss.backward()
        self.d_reducer.finish()
        self._fill_missing_grads(self.discriminator, ())          # zeros are the same on every rank: nothing to reduce
        self.d_optim.step()
        self.last_real_pred = real_pred.detach()

    def _ada_update(self):
        # ADA statistic on the last mini-batch's real predictions (generator_trainer.py:669-688), summed over ranks; tracked
        # on every D step like the reference, p only moves when augmentation is enabled
        reduce_sum = (lambda t: ddp.all_reduce_mean_(t).mul_(ddp.world_size())) if ddp.is_dist() else None
        self.stats['ada_aug_p'] = self.ada.update(self.last_real_pred, reduce_sum, self.world)
        self.stats['r_t_stat'] = self.ada.r_t

    def discriminate_pair(self, fake_img, real_img):
        """D(fake), D(real) as the reference computes them (:655-656), in ONE pass over the interleaved batch
        [f0, r0, f1, r1, ...] when that provably changes nothing: the minibatch-stddev groups of D are strided
        (members i, i + B/4, ...; gan_model.py:1005-1011), so with the per-call batch a multiple of 4 the interleaved
        batch of 2B puts exactly the fake samples of one original group, or the real ones, in each group.  Every
        other layer treats samples independently.  Halves the launches and doubles the work per launch."""
        b = fake_img.shape[0]
        if self.fuse_d_pair and b == real_img.shape[0] and b % 4 == 0:
            both = torch.stack([fake_img, real_img], dim=1).reshape(2 * b, *fake_img.shape[1:])
            pred, _ = self.discriminator(both)
            return pred[0::2], pred[1::2]
        fake_pred, _ = self.discriminator(fake_img)
        real_pred, _ = self.discriminator(real_img)
        return fake_pred, real_pred

    def discriminator_regularize_step(self, mini_real_inputs):
        tc = self.training_config
        self.stats['d_r1_loss'] = 0
        zero_grad_none(self.discriminator)
        n = len(mini_real_inputs)
        for k, real in enumerate(mini_real_inputs):
            self.d_reducer.begin(sync=(k == n - 1), phase='r1')
            real = real.detach().requires_grad_(True)
            real_pred, _ = self.discriminator(real)
            r1_loss = self.d_r1_loss(real_pred, real) / n
            self.stats['d_r1_loss'] = self.stats['d_r1_loss'] + r1_loss.detach()
            (tc['r1'] / 2 * r1_loss * tc['d_reg_every'] + 0 * real_pred[0]).backward()
            set_grad_none(self.discriminator, self.none_d_grads)
        self.d_reducer.finish()
        self._fill_missing_grads(self.discriminator, self.none_d_grads)
        self.d_optim.step()

    def discriminator_update(self, i, real_img, noise=None):
        tc = self.training_config
        mini_real = real_img.chunk(max(1, self.local_batch // self.local_mini_batch))
        requires_grad(self.generator, False)
        requires_grad(self.discriminator, True)
        mini_z = make_mini_batch_from_noise(self.sample_z(self.local_batch), self.local_batch, self.local_mini_batch)
        if i % tc['d_every'] == 0:
            self.discriminator_step(mini_z, mini_real, noise=noise)
        if i % tc['d_reg_every'] == 0:
            self.discriminator_regularize_step(mini_real)

    # -- generator ----------------------------------------------------------------------------------
    def generator_step(self, mini_noise_inputs, noise=None):
        self.stats['g_adv_loss'] = 0
        zero_grad_none(self.generator)
        n = len(mini_noise_inputs)
        for name in (self.loss_models if self.batch_utils is not None else ()):
            self.stats['g_' + name] = 0
        for k, z in enumerate(mini_noise_inputs):
            self.g_reducer.begin(sync=(k == n - 1), phase='g')
            if self.batch_utils is not None:
                z = self.batch_utils.re_arrange_z([t.clone() for t in z], k)           # generator_trainer.py:412-415
            fake_img, _ = self.generator(z, noise=noise)
            if self.training_config['augment']['enabled']:                       # generator_trainer.py:421-424
                fake_for_d, _ = augment(fake_img, self.ada.p)
            else:
                fake_for_d = fake_img
            fake_pred, _ = self.discriminator(fake_for_d)
            g_loss = self.g_nonsaturating_loss(fake_pred) / n
            self.stats['g_adv_loss'] = self.stats['g_adv_loss'] + g_loss.detach()
            if self.batch_utils is not None:
                g_loss = g_loss + self.attribute_losses(fake_img, n)                    # calc_id_losses + calc_pose_losses :432-435
            g_loss.backward()
        self.g_reducer.finish()
        self._fill_missing_grads(self.generator, self.unused_g)   # parameters outside the forward keep None, as in the reference (no Adam state)
        self.g_optim.step()

    def attribute_losses(self, fake_img, n_mini_batches):
        """Sum over the enabled predictor losses of weight * (pull the same-attribute pairs together + push the rest apart)
        (calc_id_losses / calc_pose_losses, generator_trainer.py:438-547: the same three lines per loss)."""
        total = 0
        for name, model in self.loss_models.items():
            cfg = self.training_config[name]
            if not cfg.get('enabled', True):
                continue
            feats = model.calc_features(fake_img)
            same, other = self.batch_utils.extract_same_not_same_from_list(feats, cfg['same_group_name'])
            loss = model.calc_mini_batch_loss(last_layer_same_features=same, last_layer_not_same_features=other) / n_mini_batches
            self.stats['g_' + name] = self.stats['g_' + name] + loss.detach()
            total = total + loss
        return total

    def generator_regularize_step(self, noise=None, pl_noise=None, z=None):
        tc = self.training_config
        path_batch = max(1, self.local_batch // tc['path_batch_shrink'])
        if z is None:
            z = self.sample_z(path_batch)
        # the reference chunks the path batch with the full-batch chunk count (:574-575)
        mini_z = make_mini_batch_from_noise(z, self.local_batch, self.local_mini_batch)
        zero_grad_none(self.generator)
        n = len(mini_z)
        reduce_mean = ddp.all_reduce_mean_ if ddp.is_dist() else None
        self.stats['g_path_loss'] = self.stats['g_path_length'] = self.stats['g_mean_path_length'] = 0
        for k, zk in enumerate(mini_z):
            self.g_reducer.begin(sync=(k == n - 1), phase='pl')
            fake_img, latent = self.generator(zk, noise=noise, return_latents=True)
            grad = Generator.g_path_regularize_grad(fake_img, latent, pl_noise=pl_noise)
            mpl = self._mean_path_length_tensor()
            path_loss, new_mean, path_lengths = self.g_path_regularize_grad(grad, mpl, reduce_mean=reduce_mean)
            mpl.copy_(new_mean)
            path_loss = path_loss / n
            weighted = tc['path_regularize'] * tc['g_reg_every'] * path_loss
            if tc['path_batch_shrink']:
                weighted = weighted + 0 * fake_img[0, 0, 0, 0]
            weighted.backward()
            set_grad_none(self.generator, self.none_g_grads)
            # accumulated over the mini-batches like the reference's tracker (:592-596); path_loss already carries the 1 / n
            self.stats['g_path_loss'] = self.stats['g_path_loss'] + path_loss.detach()
            self.stats['g_path_length'] = self.stats['g_path_length'] + path_lengths.detach().mean() / n
            self.stats['g_mean_path_length'] = self.stats['g_mean_path_length'] + self.mean_path_length / n
            self.stats['path_lengths'] = path_lengths.detach()
        self.g_reducer.finish()
        self._fill_missing_grads(self.generator, self.none_g_grads)
        self.g_optim.step()

    def generator_update(self, i, noise=None):
        tc = self.training_config
        requires_grad(self.generator, True)
        requires_grad(self.discriminator, False)
        mini_z = make_mini_batch_from_noise(self.sample_z(self.local_batch), self.local_batch, self.local_mini_batch)
        self.generator_step(mini_z, noise=noise)
        if i % tc['g_reg_every'] == 0:
            self.generator_regularize_step(noise=noise)
        accumulate(self.g_ema_module, self.g_module, self.accum)

    def _mean_path_length_tensor(self):
        if not torch.is_tensor(self.mean_path_length):          # callers may reset it with a plain number
            self.mean_path_length = torch.full((), float(self.mean_path_length), device=self.device)
        return self.mean_path_length

    # -- one iteration ------------------------------------------------------------------------------
    def train_iteration(self, i, real_img):
        """discriminator_update -> generator_update (EMA inside), as the loop at :351-353."""
        self.discriminator_update(i, real_img)
        self.generator_update(i)

    def save_nets(self, i, save_dir, best_fid=False):
        """The reference's checkpoint file (save_nets, generator_trainer.py:852-865): ``<save_dir>/checkpoint/<i:06d>.pt`` (or ``best_fid.pt``)
        holding 'g', 'd', 'g_ema', 'g_optim', 'd_optim' -- plus 'mean_path_length', which the reference forgets (an
        extra key its loader ignores).  Written by rank 0 only: the replicas are identical."""
        path = os.path.join(save_dir, 'checkpoint', 'best_fid.pt' if best_fid else '%s.pt' % str(i).zfill(6))
        if self.rank == 0:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            torch.save(self.state_dict(), path)
        return path

    def train(self, data=None, save_dir=None, iters=None, start_iter=None, save_every=None, on_iteration=None):
        """The loop of the reference's ``train()`` (generator_trainer.py:329-355): for i in start_iter .. iter: discriminator_update(i),
        generator_update(i), then the checkpoint cadence of end_iter_update (:728-731: ``i % save_nets_interval == 0``).  ``data`` is an iterator
        of real batches already on the device (this rank's shard; None = a resident synthetic batch, what bench.py times); evaluation, image
        grids, tensorboard and the CSV monitor of end_iter_update are outside the hot path and left to ``on_iteration(i, trainer)``.
        Returns the iteration counter after the last step."""
        tc = self.training_config
        iters = tc.get('iter', 0) if iters is None else iters
        start = tc.get('start_iter', 0) if start_iter is None else start_iter
        every = tc.get('save_nets_interval', 0) if save_every is None else save_every
        real = None if data is not None else self.synthetic_batch()
        nxt = start
        for idx in range(iters):
            i = idx + start
            if i > iters:                # the reference's own stop test (:346-349): a resumed run ends at `iter`, not `iter` steps later
                break
            batch = next(data) if data is not None else real
            self.discriminator_update(i, batch)
            self.generator_update(i)
            if save_dir is not None and every and i % every == 0 and not tc.get('debug', False):
                self.save_nets(i, save_dir)
            if on_iteration is not None:
                on_iteration(i, self)
            nxt = i + 1
        return nxt

    def reduced_stats(self):
        """Scalar statistics averaged over ranks (what the reference logs from its single process)."""
        out = {}
        for k, v in self.stats.items():
            if torch.is_tensor(v) and v.numel() == 1:
                out[k] = float(ddp.all_reduce_mean_(v.detach().clone().float()))
            elif not torch.is_tensor(v):
                out[k] = float(v)
        return out

    def synthetic_batch(self):
        """FFHQ-shaped stand-in for the data loader: float32 NCHW uniform in [-1, 1] (ffhq_dataset.py:62-63)."""
        mc = self.model_config
        return torch.rand(self.local_batch, mc['img_channels'], mc['size'], mc['size'], device=self.device, generator=self.gen) * 2 - 1
