import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "neural-locality-sensitive-hashing_amd")):
    sys.path.insert(0, p)
import torch
from nlsh_amd import io, synth
out = {}
for d, H in ((96, 32), (100, 24), (128, 16)):
    Ws, bs = synth.make_weights([d, 256, 256, H], seed=3)
    hashing = io.hashing_from_weights(Ws, bs, compat=False)
    x = torch.randn((4_000_000, d), device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hashing.hash_device(x, n=1); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): hashing.hash_device(x, n=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = 2.0 * (d * 256 + 256 * 256 + 256 * H)
    out[f"{d}->256->256->{H}"] = {"ms_per_4M_rows": round(ms, 3), "mfma_util": round(flops * 4e6 / (ms * 1e-3) / 157.3e12, 4)}
print(json.dumps({"lib": os.path.basename(os.environ.get("NLSH_HIP_LIB", "default")), **out}))
