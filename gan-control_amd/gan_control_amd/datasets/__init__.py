from .dataframe_dataset import DataFrameDataSet, get_dataframe_data_loader

__all__ = ['DataFrameDataSet', 'get_dataframe_data_loader']
