"""nlsh_amd: MI355X-native query-time hot path of Neural LSH.

Mirrors the reference's Python surface (`Indexer`, `build_index`, `MultivariateBernoulli`,
`hash_codes`, `calculate_recall`, `SIFT.distance` / `Glove.distance`) on top of a C-ABI HIP
library (`include/nlsh_hip.h`).  Submodules are imported lazily so that `nlsh_amd.synth`
and `nlsh_amd.metrics` stay usable where the HIP library has not been built.
"""
__version__ = "0.1.0"
