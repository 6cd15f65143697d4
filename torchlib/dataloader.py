"""`torchlib.dataloader` names that have a device-side implementation (see torchlib/__init__.py).  The PIL / DICOM file
loaders of the reference (CombinedLoader, PathDataset, RemoteTensorDataset, AlbumentationsTorchTransform) are host-side
data preparation outside the path: image folders are read by primia_amd.imagefolder, whose
`decode`, `scan`, `client_loader` and `validation_loader` stand where those classes stood."""
from primia_amd.augment import create_albu_transform  # noqa: F401
from primia_amd.datapipe import calc_mean_std, random_split  # noqa: F401
from primia_amd.imagefolder import client_loader, decode, scan, validation_loader  # noqa: F401
