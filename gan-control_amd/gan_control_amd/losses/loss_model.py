"""Same / not-same hinge losses on predictor features: what makes gan-control "controllable".

Mirrors the part of the reference's ``LossModelClass`` (src/gan_control/losses/loss_model.py:18-38, 107-199) that the
generator step calls when ``model_config.vanilla`` is false (generator_trainer.py:407-547): ``calc_features(img)`` runs a
FROZEN predictor network and returns a list of feature tensors (intermediate layers first, the embedding last);
``calc_mini_batch_loss(same, not_same)`` turns the pairwise distances inside the mini-batch into

    weight * ( mean(relu(d_same - lower)) + mean(relu(upper - d_not_same)) )         per feature level,

where the pairs are read off the mini-batch layout: rows (2i, 2i + 1) of the "same" block share the attribute's
sub-latent (MiniBatchUtils.re_arrange_z), every other pair below the diagonal must differ.

The predictor networks themselves (ArcFace IR-SE50, Hopenet, ESR-9, DEX age, hair segmentation, 3DMM; stock CNNs with
pretrained weights the reference downloads, README.md:97-108) are NOT part of this repository: pass any module / callable
with the ``calc_features`` contract as ``skeleton_model``.  The distance functions of the reference's criteria are here
(``CRITERIA``); they are a handful of elementwise / reduction ops on ``[mini_batch, ...]`` tensors.
"""
import torch


def _pairwise(fn):
    def dist(signatures, queries):
        return fn(signatures.unsqueeze(1) - queries.unsqueeze(0))
    return dist


CRITERIA = {
    # arc_face_criterion.py:15-21 (squared L2 between embeddings); dogfacenet uses the same form
    'embedding_loss': _pairwise(lambda d: d.pow(2).sum(-1)),
    'dog_id_loss': _pairwise(lambda d: d.pow(2).sum(-1)),
    # hopenet_criterion.py:32-37, esr9_criterion.py:15-20 (mean absolute difference over the last two axes)
    'orientation_loss': _pairwise(lambda d: d.abs().mean((-2, -1))),
    'expression_loss': _pairwise(lambda d: d.abs().mean((-2, -1))),
    # deep_age_criterion.py:17-22 (mean absolute difference over the last axis)
    'age_loss': _pairwise(lambda d: d.abs().mean(-1)),
}


def _expected_bin(logits):
    """softmax expectation over the class index: DEX age head (deep_age_criterion.py:24-33)."""
    prob = torch.softmax(logits, dim=-1)
    return (prob * torch.arange(logits.shape[-1], device=logits.device, dtype=prob.dtype)).sum(-1)


# predict(features) and controller_criterion(pred, target) of the reference's criteria, used by the phase-2 controller's
# attribute_rec objective (controller_trainer.py:231-239)
# REPRODUCED REFERENCE QUIRK (not in SURVEY Appendix C): the age criterion's predict() returns [B] (deep_age_criterion.py:25-32) while the
# controller's target is the controls tensor -- [B] in the reference's own call, but [B, 1] when a caller keeps the column axis, and
# `self.mse(pred, target)` (:37-38) then BROADCASTS [B] against [B, 1] to [B, B] (torch only warns).  The criterion below is the same
# unguarded mse_loss on purpose: tests/test_controller.py pins the [B] case against the reference; do not "fix" the shapes here.
PREDICTORS = {
    'age_loss': (_expected_bin, lambda pred, target: torch.nn.functional.mse_loss(pred, target)),
    # hopenet_criterion.py:6-19, 38-43: three 66-bin heads -> degrees; L1 against the controls
    'orientation_loss': (lambda f: _expected_bin(f) * 3 - 99, lambda pred, target: (pred - target).abs().mean()),
}


def _l1_expand(features):
    """Intermediate feature maps [n, c, h, w]: mean absolute difference of every pair (loss_model.py:140-143)."""
    return (features.unsqueeze(1) - features.unsqueeze(0)).abs().mean((2, 3, 4))


class LossModelClass:
    """config keys (configs/ffhq.json:86-160): lower_thres / upper_thres / intermediate_layers_weights per intermediate level,
    last_lower_thres / last_upper_thres / last_layer_weight, focus_on_list (one entry per level incl. the last)."""

    def __init__(self, config, loss_name='embedding_loss', mini_batch_size=4, device='cuda', no_model=False, parallel=True,
                 skeleton_model=None, criterion=None, predict_fn=None, controller_criterion_fn=None):
        self.config, self.loss_name = config, loss_name
        if skeleton_model is None and not no_model:
            raise RuntimeError('LossModelClass(%s): the pretrained predictor is not part of this repository; pass skeleton_model=<frozen '
                               'network returning the list of features>, or no_model=True to use the loss on precomputed features' % loss_name)
        self.skeleton_model = skeleton_model
        if criterion is None:
            if loss_name not in CRITERIA:
                raise ValueError('self.loss_name = %s (not valid)' % loss_name)
            criterion = CRITERIA[loss_name]
        self.last_layer_criterion = criterion
        self.lower_thres, self.upper_thres = config['lower_thres'], config['upper_thres']
        self.last_lower_thres, self.last_upper_thres = config['last_lower_thres'], config['last_upper_thres']
        self.weights = list(config['intermediate_layers_weights']) + [config['last_layer_weight']]
        self.focus_on_list = config['focus_on_list']
        self.mini_batch_size = mini_batch_size
        heads = PREDICTORS.get(loss_name, (None, None))
        self._predict_fn = predict_fn or heads[0]
        self._controller_criterion_fn = controller_criterion_fn or heads[1]

    def calc_features(self, batch):
        return self.skeleton_model(batch)

    def predict(self, generator_output_image, features=None):
        """Attribute values read off the predictor's last features (loss_model.py:107-110)."""
        if self._predict_fn is None:
            raise NotImplementedError('LossModelClass(%s).predict: pass predict_fn' % self.loss_name)
        if features is None:
            features = self.calc_features(generator_output_image)[-1]
        return self._predict_fn(features)

    def controller_criterion(self, pred, target):
        if self._controller_criterion_fn is None:
            raise NotImplementedError('LossModelClass(%s).controller_criterion: pass controller_criterion_fn' % self.loss_name)
        return self._controller_criterion_fn(pred, target)

    # -- pair masks (loss_model.py:181-199): strictly-lower-triangular [row, col] pairs --------------------------------
    @staticmethod
    def pair_masks(n_same, n_not_same, device=None):
        """(valid, same_pairs, not_same_pairs) boolean [n, n] masks, n = n_same + n_not_same: consecutive rows (2i, 2i+1) of the
        first block are the pairs that share the attribute; consecutive rows of the second block are the 'not-same block'
        pairs (they share some OTHER attribute)."""
        n = n_same + n_not_same
        idx = torch.arange(n, device=device)
        valid = idx[:, None] > idx[None, :]
        consecutive = (idx[:, None] == idx[None, :] + 1) & (idx[None, :] % 2 == 0)
        same = consecutive & (idx[None, :] < 2 * (n_same // 2))
        not_same = consecutive & (idx[None, :] >= 2 * (n_same // 2)) & (idx[None, :] < 2 * (n_same // 2) + 2 * (n_not_same // 2))
        return valid, same & valid, not_same & valid

    def _level_loss(self, dist, masks, focus, lower, upper):
        valid, same_mask, not_same_mask = masks
        if focus == 'same_as_last_layer':
            same_d, not_same_d = dist[same_mask], dist[~same_mask & valid]
        elif focus == 'not_same_as_last_layer':
            same_d, not_same_d = dist[not_same_mask], dist[~not_same_mask & valid]
        else:
            raise ValueError('focus_on_list entry %r' % (focus,))
        return torch.clamp(same_d - lower, min=0.).mean() + torch.clamp(upper - not_same_d, min=0.).mean()

    def calc_mini_batch_loss(self, last_layer_same_features=None, last_layer_not_same_features=None):
        if last_layer_same_features is None:
            raise ValueError('last_layer_same_features is None')
        if last_layer_not_same_features is None:
            raise ValueError('last_layer_not_same_features is None')
        same, other = last_layer_same_features, last_layer_not_same_features
        masks = self.pair_masks(same[0].shape[0], other[0].shape[0], device=same[-1].device)
        as_last = bool(self.config.get('intermediate_criterion_as_last_layer'))
        loss = 0
        for level in range(len(same) - 1):
            if self.weights[level] == 0:
                continue
            feats = torch.cat([same[level], other[level]], dim=0)
            dist = self.last_layer_criterion(feats, feats) if as_last else _l1_expand(feats)
            loss = loss + self.weights[level] * self._level_loss(dist, masks, self.focus_on_list[level], self.lower_thres[level], self.upper_thres[level])
        emb = torch.cat([same[-1], other[-1]], dim=0)
        dist = self.last_layer_criterion(emb, emb)
        return loss + self.weights[-1] * self._level_loss(dist, masks, self.focus_on_list[-1], self.last_lower_thres, self.last_upper_thres)
