"""Import shim: the reference's module paths (`nlsh.indexer`, `nlsh.hashings`, `nlsh.utils`, `nlsh.metrics`,
`nlsh.data`, top-level `encoders`) resolving to the MI355X implementation in `nlsh_amd`.

Put THIS directory on sys.path instead of the reference checkout and the callers of the hot path
(`nlsh/trainers/base.py:82-111`, `nlsh/trainers/proposed.py:101-104`, `eval.py:11-13`) import the HIP path
without an edit.  Only the hot-path modules exist here; trainers, loggers and CLIs are out of scope.
"""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
