"""N2 (minimal trainer) host-side formulas against fixtures produced by the unmodified reference
(tests/golden/make_golden.py::g8: nlsh/trainers/triplet.py:16-26 on nlsh/learning/distances.py:245-254)."""
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_triplet_loss_and_code_distance_match_reference_fixture():
    from nlsh_amd.training import code_l2_rowwise, triplet_loss
    g = np.load(os.path.join(G, "g8_triplet.npz"))
    for name in ("a", "b", "c"):
        ta, tp, tn = (torch.from_numpy(g[f"{name}_{w}"]).requires_grad_(True) for w in ("anchor", "pos", "neg"))
        margin = float(g[f"{name}_margin"])
        assert np.array_equal(code_l2_rowwise(ta, tp).detach().numpy(), g[f"{name}_d_pos"])
        loss = triplet_loss(ta, tp, tn, margin=margin)
        assert np.array_equal(loss.detach().numpy(), g[f"{name}_loss"])
        loss.backward()
        assert np.allclose(ta.grad.numpy(), g[f"{name}_grad_anchor"], rtol=0, atol=1e-7)
        assert np.allclose(tn.grad.numpy(), g[f"{name}_grad_neg"], rtol=0, atol=1e-7)


def test_triplet_batches_follow_the_reference_sampling_rule():
    """triplet.py:101-131 (method "random"): shuffled anchors once per epoch, positive = one of the first
    `positive_k` neighbours of the anchor, negative = any row."""
    from nlsh_amd.training import triplet_batches
    n, K = 1000, 10
    knn = torch.arange(n)[:, None] * 100 + torch.arange(K)[None, :]        # neighbour j of row i is encoded as 100*i + j
    gen = torch.Generator()
    gen.manual_seed(3)
    seen = []
    for a, p, ng in triplet_batches(n, knn, positive_k=4, batch_size=128, generator=gen):
        assert a.shape == p.shape == ng.shape == (128,)
        assert bool((p // 100 == a).all()) and bool((p % 100 < 4).all())
        assert bool(((ng >= 0) & (ng < n)).all())
        seen.append(a)
    seen = torch.cat(seen)
    assert seen.numel() == (n // 128) * 128 and seen.unique().numel() == seen.numel()   # no anchor twice per epoch
    band = list(triplet_batches(n, knn, positive_k=4, batch_size=128, generator=gen, negative_band=(6, 9)))
    for a, p, ng in band:
        assert bool((ng // 100 == a).all()) and bool(((ng % 100 >= 6) & (ng % 100 < 9)).all())
