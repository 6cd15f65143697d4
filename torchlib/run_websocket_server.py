"""`read_websocket_config` of torchlib/run_websocket_server.py (see torchlib/__init__.py)."""
from primia_amd.torchlib_compat import read_websocket_config  # noqa: F401
