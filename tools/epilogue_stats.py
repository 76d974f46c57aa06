import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")): sys.path.insert(0, p)
import numpy as np, torch
from nlsh_amd import _capi, io, synth
from nlsh_amd.data import SIFT
from nlsh_amd.indexer import Indexer
N, d, Q = 1_000_000, 128, 10_000
corpus_h, mean, std = synth.standardise(synth.sift_manifold(N, d, seed=synth.SEED_DATA))
queries_h, _, _ = synth.standardise(synth.sift_manifold(Q, d, seed=synth.SEED_QUERY), mean, std)
Ws, bs = io.load_hasher_weights(os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "sift1m_manifold_h16.npz"))
ix = Indexer(io.hashing_from_weights(Ws, bs, compat=True), torch.from_numpy(corpus_h).cuda(), SIFT.distance, algo="tiled")
q = torch.from_numpy(queries_h).cuda()
keys, nkeys = ix.hash_device(q, hash_times=10, seed=7)
runs = 5
for _ in range(runs): ix.scan_tensors(q, keys, nkeys, k=10)
torch.cuda.synchronize()
buf = np.zeros(((1 << 16), 8), dtype=np.float32)
assert _capi.lib().nlsh_debug_scan_trace(buf.ctypes.data_as(ctypes.c_void_p), int(buf.size)) == 0
c = buf[-1] / runs
print(f"(task, query) lists per launch {c[0]:.0f}: with a published bound {c[1] / c[0]:.3f}, no survivor {c[2] / c[0]:.3f}, 1..k-1 survivors {c[3] / c[0]:.3f}; candidates per list {c[4] / c[0]:.1f}, survivors per list {c[5] / c[0]:.1f}")
