from .mnist import DataSet, Datasets, DeviceDataset, read_data_sets, synthetic_mnist  # noqa: F401
