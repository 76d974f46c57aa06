"""`nlsh.hashings` of the reference -> `nlsh_amd.hashings` (see nlsh/__init__.py)."""
from nlsh_amd.hashings import *  # noqa: F401,F403
from nlsh_amd import hashings as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
