from jnerf_amd.dataset import NerfDataset, SyntheticNerfDataset  # noqa: F401
