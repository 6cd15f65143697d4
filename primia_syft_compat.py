"""`import primia_syft_compat as sy` — the PySyft worker-facing objects PriMIA's federated set-up uses,
implemented in primia_amd/syft_compat.py."""
from primia_amd.syft_compat import *  # noqa: F401,F403
from primia_amd.syft_compat import __all__  # noqa: F401
