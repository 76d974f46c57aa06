"""`nlsh.utils` of the reference -> `nlsh_amd.utils` (see nlsh/__init__.py)."""
from nlsh_amd.utils import *  # noqa: F401,F403
from nlsh_amd import utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
