"""`nlsh.indexer` of the reference -> `nlsh_amd.indexer` (see nlsh/__init__.py)."""
from nlsh_amd.indexer import *  # noqa: F401,F403
from nlsh_amd import indexer as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
