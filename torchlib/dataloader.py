"""`torchlib.dataloader` names that have a device-side implementation (see torchlib/__init__.py)."""
from primia_amd.datapipe import calc_mean_std  # noqa: F401
