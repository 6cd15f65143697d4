"""`import primia_syft_compat as sy` — the PySyft worker-facing objects PriMIA's federated set-up uses,
implemented in primia_amd/syft_compat.py."""
from primia_amd import syft_compat as _impl
from primia_amd.syft_compat import *  # noqa: F401,F403
from primia_amd.syft_compat import __all__  # noqa: F401

hook = None
local_worker = None


def __getattr__(name):        # anything not re-exported above (module globals PySyft sets later) comes from the implementation
    return getattr(_impl, name)
