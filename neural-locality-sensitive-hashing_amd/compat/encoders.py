"""Top-level `encoders` module of the reference (encoders.py:8-55) -> `nlsh_amd.encoders`."""
import nlsh  # noqa: F401  (puts the package directory on sys.path)
from nlsh_amd.encoders import *  # noqa: F401,F403
from nlsh_amd import encoders as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
