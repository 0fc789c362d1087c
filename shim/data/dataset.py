"""Reference module path `data.dataset` (src/data/dataset.py:5-49) -> gaviko_amd.data."""
from gaviko_amd.data import CustomDataset, CustomDatasetPrediction  # noqa: F401
