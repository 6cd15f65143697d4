"""`torchlib.models` of the hot path: `resnet18(...)` builds the HIP engine (see torchlib/__init__.py).  vgg16 and
conv_at_resolution are other model families, outside the path this repository accelerates (BASELINE.json north_star)."""
from primia_amd.torchlib_compat import conv_at_resolution, resnet18, vgg16  # noqa: F401
