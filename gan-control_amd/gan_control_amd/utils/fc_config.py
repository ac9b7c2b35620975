"""Latent sub-group description that shapes the split mapping network.

Reference: FcConfig and MiniBatchUtils.get_fc_config / get_groups / get_ordered_group_names,
src/gan_control/utils/mini_batch_multi_split_utils.py:13-16, 45-54, 103-115.  Only the part the
Generator consumes is built; the same-attribute pairing logic feeds the (out-of-scope) predictor
losses.
"""
from typing import Dict, List


class FcConfig:
    def __init__(self, in_order_group_names: List[str], groups: Dict[str, dict]):
        self.in_order_group_names = in_order_group_names
        self.groups = groups


def fc_config_from_sub_groups(sub_groups_dict, latent_size=512):
    """training_config['sub_groups_dict'] (e.g. ffhq.json:35-71) -> FcConfig, groups ordered by latent offset."""
    names = sorted(sub_groups_dict.keys(), key=lambda n: sub_groups_dict[n]['place_in_latent'][0])
    groups, total = {}, 0
    for n in names:
        lo, hi = sub_groups_dict[n]['place_in_latent']
        groups[n] = {'latent_place': [lo, hi], 'latent_size': hi - lo}
        total += hi - lo
    if total != latent_size:
        raise ValueError('%d != latent_count_size %d' % (latent_size, total))
    return FcConfig(names, groups)
