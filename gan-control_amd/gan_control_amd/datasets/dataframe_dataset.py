"""(attribute, w-latent) pairs for the controller.

Reference: datasets/dataframe_dataset.py:18-56 -- a pickled pandas DataFrame with a ``latents_w`` column and one column per
attribute; first 90 % train, last 10 % eval; ``age`` gets a trailing unit axis, ``expression_q`` becomes a one-hot of 8.
"""
import pandas as pd
import torch
from torch.utils import data


class DataFrameDataSet(data.Dataset):
    def __init__(self, dataframe_path, attribute=None, train=True):
        frame = dataframe_path if isinstance(dataframe_path, pd.DataFrame) else pd.read_pickle(dataframe_path)
        cut = int(len(frame.latents_w) * 0.9)
        frame = frame.iloc[:cut] if train else frame.iloc[cut:]
        self.train, self.attribute = train, attribute
        self.attributes_df = frame[['latents_w', attribute]] if attribute is not None else frame

    def __len__(self):
        return len(self.attributes_df.latents_w)

    def __getitem__(self, index):
        row = self.attributes_df.iloc[index]
        value = torch.tensor(row[self.attribute])
        if self.attribute == 'age':
            value = value.unsqueeze(0)
        elif self.attribute == 'expression_q':
            value = torch.nn.functional.one_hot(value, num_classes=8)
        return value, torch.tensor(row['latents_w'])


def get_dataframe_data_loader(dataframe_path, attribute, batch_size=32, shuffle=True, drop_last=True, workers=32, train=True):
    return data.DataLoader(DataFrameDataSet(dataframe_path, attribute=attribute, train=train), batch_size=batch_size,
                           shuffle=shuffle, drop_last=drop_last, num_workers=workers)
