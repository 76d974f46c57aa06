"""`nlsh.data` of the reference -> `nlsh_amd.data` (see nlsh/__init__.py)."""
from nlsh_amd.data import *  # noqa: F401,F403
from nlsh_amd import data as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
