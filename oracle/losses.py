"""Oracle for the same / not-same hinge losses (test infrastructure only; see oracle/__init__.py).

Plain-loop restatement of LossModelClass.calc_mini_batch_loss (src/gan_control/losses/loss_model.py:121-199) and of
MiniBatchUtils.re_arrange_z / extract_same_not_same_from_list (utils/mini_batch_multi_split_utils.py:55-87), pinned against
the reference's own classes by oracle/make_golden.py::golden_losses.
"""
import torch

DISTANCES = {
    # arc_face_criterion.py:15-21
    'embedding_loss': lambda a, b: (a - b).pow(2).sum(),
    # esr9_criterion.py:15-20 / hopenet_criterion.py:32-37
    'expression_loss': lambda a, b: (a - b).abs().mean(),
    'orientation_loss': lambda a, b: (a - b).abs().mean(),
    # deep_age_criterion.py:17-22
    'age_loss': lambda a, b: (a - b).abs().mean(),
}


def hinge_pair_loss(same, other, cfg, loss_name):
    """same / other: lists of feature tensors (intermediate levels first); returns the scalar loss (loss_model.py:121-179)."""
    n_same, n_other = same[0].shape[0], other[0].shape[0]
    n = n_same + n_other
    weights = list(cfg['intermediate_layers_weights']) + [cfg['last_layer_weight']]
    lowers = list(cfg['lower_thres']) + [cfg['last_lower_thres']]
    uppers = list(cfg['upper_thres']) + [cfg['last_upper_thres']]
    same_pairs = {(2 * i + 1, 2 * i) for i in range(n_same // 2)}                                  # make_same_last_layer_mask :181-186
    other_pairs = {(2 * i + 1, 2 * i) for i in range(n_same // 2, n_same // 2 + n_other // 2)}     # make_not_same_last_layer_mask :188-193
    total = 0
    for level in range(len(same)):
        last = level == len(same) - 1
        if weights[level] == 0 and not last:
            continue
        feats = torch.cat([same[level], other[level]], dim=0)
        if last or cfg.get('intermediate_criterion_as_last_layer'):
            dist = DISTANCES[loss_name]
        else:
            dist = lambda a, b: (a - b).abs().mean()                                                # L1 over (c, h, w) :140-143
        focus = cfg['focus_on_list'][level] if not last else cfg['focus_on_list'][-1]
        pull = same_pairs if focus == 'same_as_last_layer' else other_pairs
        near, far = [], []
        for r in range(n):
            for c in range(r):                                                                      # valid_mask = strictly lower triangle :36
                d = dist(feats[r], feats[c])
                (near if (r, c) in pull else far).append(d)
        near_loss = torch.stack([torch.clamp(d - lowers[level], min=0.) for d in near]).mean()
        far_loss = torch.stack([torch.clamp(uppers[level] - d, min=0.) for d in far]).mean()
        total = total + weights[level] * (near_loss + far_loss)
    return total


def re_arrange_z(z_list, sub_groups):
    """Returns new tensors: within every group's rows, odd rows copy the preceding even row's sub-latent (mb.py:64-78)."""
    out = [z.clone() for z in z_list]
    order = sorted(sub_groups, key=lambda n: sub_groups[n]['place_in_latent'][0])
    for name in order:
        rows = sub_groups[name]['place_in_mini_batch']
        if rows is None:
            continue
        lo, hi = sub_groups[name]['place_in_latent']
        for i in range(rows[0], rows[1], 2):
            out[0][i + 1, lo:hi] = out[0][i, lo:hi]
    if len(out) > 1:
        rows = sub_groups['other']['place_in_mini_batch'] if 'other' in sub_groups else None
        for j in range(1, len(out)):
            if rows is not None:
                out[j][:rows[0]] = out[0][:rows[0]]
                out[j][rows[1]:] = out[0][rows[1]:]
            else:
                out[j] = out[0]
    return out


def split_same_not_same(features, sub_groups, name, mini_batch):
    lo, hi = sub_groups[name]['place_in_mini_batch']
    keep = list(range(lo)) + list(range(hi, mini_batch))
    return [f[lo:hi] for f in features], [f[keep] for f in features]
