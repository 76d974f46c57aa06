"""BASELINE.json configs[4] at its STATED size on one GPU: Deep100M-shaped, 100,000,000 x 96-d fp32 (38.4 GB, generated on
the device and HBM-resident, plus the bucket-sorted copy), 32-bit learned hash with full-width keys (the committed
checkpoints/deep100m_manifold_h32.npz, trained by tools/scale_deep100m.py --save-hash on a 1 M-row sample of the same
generator), 100,000 queries, k = 10, hash_times = 10.  (The 8-rank form of this config shards these rows; one GPU holds
them all, so every kernel runs at the full problem size here.)

Parity =
* the size-independent properties of `helpers.check_scan_properties` on ALL 100 k queries (candidate counts recomputed
  independently, membership of every returned id in a probed bucket, ascending order, no duplicates, distances vs stock
  torch ops on the returned rows);
* the scalar oracle on a 64-query slice, fed with ONLY the rows of the buckets those queries probe (gathered on the device, a
  ~1 GB host copy instead of 38 GB), arranged so that its (distance, row id) tie order is the global one: ids and candidate
  counts exact, L2 distances bit-identical (tiled schedule: same k-ascending fmaf chain);
* the oracle's AVX2/OpenMP scan (pinned bit-identical to the scalar form, tests/test_oracle_golden.py) on 4,096 queries spread
  over the whole batch, against a host copy of the WHOLE corpus (38.4 GB) and the index's own bucket lists after the grouping has
  been verified in full on the device: ~2 * 10^8 (query, candidate) pairs, ids / counts / distance bits exact;
* the hard keys of a corpus slice against the oracle's forward + full-width pack."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import check_scan_properties
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CKPT = os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "checkpoints", "deep100m_manifold_h32.npz")


def test_deep100m_full_size_one_gpu():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scale_deep100m as gen
    from nlsh_amd import _capi, io
    from nlsh_amd.data import SIFT
    from nlsh_amd.indexer import Indexer
    N, Q, d, H, k, P = 100_000_000, 100_000, 96, 32, 10, 10
    device = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(device)
    assert free > 120e9, f"configs[4] needs ~90 GB of HBM on one GPU, {free / 1e9:.0f} GB free"
    params = gen._manifold_params(device, d)
    cg = gen.deep_manifold_device(0, N, d, 1234, device, params)
    qg = gen.deep_manifold_device(0, Q, d, 4321, device, params)
    Ws, bs = io.load_hasher_weights(CKPT)
    assert [w.shape for w in Ws] == [(256, 96), (256, 256), (32, 256)]
    hashing = io.hashing_from_weights(Ws, bs, compat=False)
    ix = Indexer(hashing, cg, SIFT.distance, compat=False)
    try:
        assert ix._hashing.key_mode == _capi.KEY_FULL and ix.corpus_sorted.shape == (N, 96)
        ckeys = ix.corpus_keys
        assert int((ckeys < 0).sum()) > 0 and int((ckeys >= 0).sum()) > 0      # 32-bit codes on both sides of 2^31
        assert ix.n_buckets > 20_000
        assert bool((ix.uniq_keys[1:] > ix.uniq_keys[:-1]).all())               # CSR in ascending signed order
        assert int(ix.offsets[-1]) == N
        # the index is a permutation grouped by key (sampled: a full check would sort 100 M ids again)
        probe = torch.randint(0, N, (1 << 20,), device=device)
        assert torch.equal(ckeys[ix.perm[probe].long()], ix.uniq_keys[torch.searchsorted(ix.offsets, probe.int(), right=True) - 1])
        assert torch.equal(ix.corpus_sorted[probe], cg[ix.perm[probe].long()])

        keys, nkeys = ix.hash_device(qg, hash_times=P, seed=9)
        assert int(nkeys.min()) >= 1 and int(nkeys.max()) > 1
        dist, idx, nc, _ = ix.scan_tensors(qg, keys, nkeys, k=k)
        assert ix.last_algo == _capi.SCAN_BUCKET_TILED
        assert float(nc.float().mean()) > 10_000                                 # tens of thousands of candidates per query
        check_scan_properties(ix, qg, cg, keys, nkeys, dist, idx, nc, k, "l2")
        again = ix.scan_tensors(qg, keys, nkeys, k=k)                            # idempotent (workspace counters handed back)
        assert torch.equal(again[0], dist) and torch.equal(again[1], idx) and torch.equal(again[2], nc)

        # ---- oracle on a 64-query slice, on the rows of the probed buckets only
        S = 64
        ks, nks = keys[:S], nkeys[:S]
        valid = torch.arange(P, device=device)[None, :] < nks[:, None]
        pos = torch.searchsorted(ix.uniq_keys, ks.clamp(min=int(ix.uniq_keys.min()), max=int(ix.uniq_keys.max()))).clamp(max=ix.n_buckets - 1)
        hit = (ix.uniq_keys[pos] == ks) & valid
        bidx = torch.unique(pos[hit])                                            # probed buckets, ascending CSR order
        lo, hi = ix.offsets[bidx].long(), ix.offsets[bidx + 1].long()
        sizes = hi - lo
        tot = int(sizes.sum())
        assert tot < 8_000_000, tot
        starts = torch.cumsum(sizes, 0) - sizes
        srows = (torch.arange(tot, device=device) - torch.repeat_interleave(starts, sizes)) + torch.repeat_interleave(lo, sizes)
        gids = ix.gid[srows].long()                                              # global row ids, bucket by bucket
        order = torch.argsort(gids)                                              # compact corpus in GLOBAL id order: same tie order
        rank = torch.empty_like(order)
        rank[order] = torch.arange(tot, device=device)
        sub = cg[gids[order]].cpu().numpy()
        perm_c = rank.int().cpu().numpy()                                        # bucket-grouped list of compact row ids
        uniq_c = (ix.uniq_keys[bidx].cpu().numpy().astype(np.int64)) & 0xFFFFFFFF
        offs_c = np.concatenate([starts.cpu().numpy(), [tot]]).astype(np.int64)
        o = np.argsort(uniq_c, kind="stable")                                    # the oracle looks keys up in ascending UNSIGNED order
        offs_sorted = np.concatenate([[0], np.cumsum((offs_c[1:] - offs_c[:-1])[o])]).astype(np.int64)
        perm_sorted = np.concatenate([perm_c[offs_c[b]:offs_c[b + 1]] for b in o]).astype(np.int32)
        od, oi, onc = oracle.query_batch(sub, perm_sorted, uniq_c[o], offs_sorted, qg[:S].cpu().numpy(),
                                         ks.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, nks.cpu().numpy(), k, "l2")
        assert np.array_equal(nc[:S].cpu().numpy(), onc)
        back = gids[order].cpu().numpy()
        oi_global = np.where(oi >= 0, back[np.clip(oi, 0, tot - 1)], -1)
        assert np.array_equal(idx[:S].cpu().numpy(), oi_global)
        assert np.array_equal(dist[:S].cpu().numpy().view(np.uint32), od.view(np.uint32))   # bit-identical L2 distances

        # ---- oracle (SIMD form) on 4,096 queries over the whole batch, on the whole corpus
        # the index's grouping verified IN FULL first, so the oracle may walk the device's bucket lists: a permutation of the rows,
        # buckets in ascending signed key order, rows ascending inside a bucket = what a stable sort by key yields (nlsh/indexer.py:6-24)
        permL = ix.perm.long()
        assert torch.equal(torch.sort(ix.perm).values, torch.arange(N, device=device, dtype=torch.int32))
        sk = ckeys[permL]
        assert bool((sk[1:] >= sk[:-1]).all())
        same = sk[1:] == sk[:-1]
        assert bool((ix.perm[1:][same] > ix.perm[:-1][same]).all())
        heads = torch.nonzero(~same).view(-1) + 1
        assert torch.equal(ix.offsets[1:-1].long(), heads) and torch.equal(ix.uniq_keys, sk[ix.offsets[:-1].long()])
        assert ix.row_ids is None and torch.equal(ix.gid, ix.perm)               # ids the scan reports = the rows' global ids
        del permL, sk, same, heads
        S2 = 4096
        pick = torch.arange(S2, device=device) * (Q // S2)                       # every 24th query of the batch
        corpus_h = cg.cpu().numpy()                                              # 38.4 GB of host memory for the length of this block
        od, oi, onc = oracle.query_batch(corpus_h, ix.perm.cpu().numpy(), ix.uniq_keys.cpu().numpy().astype(np.int64),
                                         ix.offsets.cpu().numpy().astype(np.int64), qg[pick].cpu().numpy(),
                                         keys[pick].cpu().numpy().astype(np.int64), nkeys[pick].cpu().numpy(), k, "l2", simd=True)
        del corpus_h
        assert np.array_equal(nc[pick].cpu().numpy(), onc)
        assert np.array_equal(idx[pick].cpu().numpy(), oi)
        assert np.array_equal(dist[pick].cpu().numpy().view(np.uint32), od.view(np.uint32))
        print(f"[oracle, Deep100M] {S2} queries x {onc.mean():.0f} candidates bit-identical ({onc.sum() / 1e6:.0f} M pairs)")

        # ---- the opt-in folded L2 form on ALL 100 k queries: differences from the exact form are ties at the stated tolerance only
        from helpers import l2_forms_differ_only_at_ties
        ix.l2_form = "folded"
        try:
            d_f, i_f, n_f, _ = ix.scan_tensors(qg, keys, nkeys, k=k)
        finally:
            ix.l2_form = "exact"
        assert torch.equal(n_f, nc)
        n_diff = l2_forms_differ_only_at_ties(qg, cg, dist, idx, d_f, i_f)
        assert n_diff <= Q // 100, n_diff
        print(f"[l2 forms, Deep100M] id lists differing: {n_diff} of {Q}")

        # ---- hard keys of a corpus slice: oracle forward + full-width pack
        z = oracle.mlp_forward(cg[:4096].cpu().numpy(), Ws, bs)
        _, p01 = oracle.head_probs(z)
        ko, _ = oracle.row_keys(p01, 1, "full")
        assert np.array_equal(ko[:, 0].astype(np.int64) & 0xFFFFFFFF, ckeys[:4096].cpu().numpy().astype(np.int64) & 0xFFFFFFFF)
    finally:
        del ix, cg
        torch.cuda.empty_cache()
