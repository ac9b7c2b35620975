"""Mini-batch layout of the controllable generator step: which rows share which sub-latent.

Mirrors the reference's ``MiniBatchUtils`` (src/gan_control/utils/mini_batch_multi_split_utils.py:18-115): every attribute
group owns a slice of the latent vector (``place_in_latent``) and a block of rows of the mini-batch (``place_in_mini_batch``);
inside its block consecutive rows (2i, 2i + 1) are made to share the group's sub-latent, so the predictor of that attribute
must see "same" for exactly those pairs (losses/loss_model.py).
"""
from .fc_config import FcConfig, fc_config_from_sub_groups


class MiniBatchUtils:
    def __init__(self, mini_batch, sub_groups_dict, total_batch=8, debug=False, latent_size=512):
        self.mini_batch, self.total_batch, self.sub_groups_dict, self.debug = mini_batch, total_batch, sub_groups_dict, debug
        self.place_in_mini_batch_dict = {n: g['place_in_mini_batch'] for n, g in sub_groups_dict.items()}
        self.place_in_latent_dict = {n: g['place_in_latent'] for n, g in sub_groups_dict.items()}
        self.sub_group_names = sorted(sub_groups_dict, key=lambda n: sub_groups_dict[n]['place_in_latent'][0])
        self.num_of_sub_groups = len(sub_groups_dict)
        self.num_of_mini_batchs = total_batch // mini_batch
        rows = sum(p[1] - p[0] for p in self.place_in_mini_batch_dict.values() if p is not None)
        if rows != mini_batch:
            raise ValueError('self.mini_batch %d != mini_batch_count %d' % (mini_batch, rows))
        width = sum(p[1] - p[0] for p in self.place_in_latent_dict.values())
        if width != latent_size:
            raise ValueError('%d != latent_count_size %d' % (latent_size, width))

    def get_ordered_group_names(self):
        return list(self.sub_group_names)

    def get_fc_config(self) -> FcConfig:
        return fc_config_from_sub_groups(self.sub_groups_dict, sum(p[1] - p[0] for p in self.place_in_latent_dict.values()))

    def get_sub_group(self, batch, sub_group_name='id'):
        lo, hi = self.place_in_mini_batch_dict[sub_group_name]
        return batch[lo:hi]

    def get_not_sub_group(self, batch, sub_group_name='id'):
        lo, hi = self.place_in_mini_batch_dict[sub_group_name]
        return batch[list(range(lo)) + list(range(hi, self.mini_batch))]

    def extract_same_not_same_from_list(self, feature_list, same_group_name):
        return ([self.get_sub_group(f, same_group_name) for f in feature_list],
                [self.get_not_sub_group(f, same_group_name) for f in feature_list])

    def re_arrange_z(self, z_batch, batch_num=0):
        """In place: row 2i + 1 of every group's block takes row 2i's sub-latent of that group; further style codes (mixing)
        follow the first one outside the 'other' block (mini_batch_multi_split_utils.py:64-78)."""
        z0 = z_batch[0]
        for name in self.sub_group_names:
            rows = self.place_in_mini_batch_dict[name]
            if rows is None:
                continue
            lo, hi = self.place_in_latent_dict[name]
            z0[rows[0] + 1:rows[1]:2, lo:hi] = z0[rows[0]:rows[1] - 1:2, lo:hi].detach()
        other = self.place_in_mini_batch_dict.get('other') if 'other' in self.sub_group_names else None
        for i in range(1, len(z_batch)):
            if other is not None:
                z_batch[i][:other[0]] = z0[:other[0]]
                z_batch[i][other[1]:] = z0[other[1]:]
            else:
                z_batch[i] = z0
        return z_batch

    def re_arrange_inject_noise(self, noises, group_name='id'):
        rows = self.place_in_mini_batch_dict[group_name]
        for n in noises:
            n[rows[0] + 1:rows[1]:2] = n[rows[0]:rows[1] - 1:2].detach()
        return noises
