"""`nlsh.metrics` of the reference -> `nlsh_amd.metrics` (see nlsh/__init__.py)."""
from nlsh_amd.metrics import *  # noqa: F401,F403
from nlsh_amd import metrics as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
