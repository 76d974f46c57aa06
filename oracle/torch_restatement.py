"""torch-CPU restatement of the reference's query loop -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

SURVEY.md §8(d) asks for the CPU baseline as "the build's CPU restatement of the reference algorithm (per-query
gather -> distance -> topk using torch-CPU ops)".  This is nlsh/indexer.py:56-96 restated on host tensors: a Python
loop over the queries, `index2row` as a dict of LongTensors (indexer.py:6-24), a dense scratch buffer filled bucket by
bucket (:60-83), `F.pairwise_distance` / `1 - F.cosine_similarity` (nlsh/data.py:99-109,191-201), `topk(k,
largest=False)` (:88-90), the `<k` fallback (:91-93).  Only bench.py's `cpu_baseline` leg and tests import it; the
product never does.  Pinned by tests/test_oracle_golden.py against the C oracle (itself pinned by the reference's
golden vectors): identical candidate counts and, up to torch.topk's tie order, identical id lists.
"""
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def build_index2row(perm: np.ndarray, uniq_keys: np.ndarray, offsets: np.ndarray) -> Dict[int, torch.Tensor]:
    """CSR of the oracle -> the reference's dict key -> LongTensor of ascending rows (indexer.py:6-24)."""
    rows = torch.from_numpy(np.ascontiguousarray(perm).astype(np.int64))
    return {int(k): rows[int(offsets[i]):int(offsets[i + 1])] for i, k in enumerate(uniq_keys)}


def query(corpus: torch.Tensor, index2row: Dict[int, torch.Tensor], queries: torch.Tensor,
          key_lists: Sequence[Sequence[int]], k: int = 10, metric: str = "l2") -> Tuple[List[List[int]], List[int]]:
    """indexer.py:56-96 with the key sets given (hashing is timed separately by the caller)."""
    result, n_candidates = [], []
    buf = torch.empty_like(corpus)                                   # indexer.py:60: scratch the size of the corpus
    empty = torch.zeros((0,), dtype=torch.int64)
    for qi, keys in enumerate(key_lists):
        start, rows_list = 0, []
        rows = empty
        for key in keys:                                             # indexer.py:66-76
            rows = index2row.get(int(key), empty)
            n = rows.shape[0]
            if n:
                torch.index_select(corpus, 0, rows, out=buf[start:start + n])   # no temporary (indexer.py:75-82)
                rows_list.append(rows)
                start += n
        n_candidates.append(start)
        cand = buf[:start]
        q = queries[qi]
        if metric == "l2":
            dist = F.pairwise_distance(q, cand)                      # nlsh/data.py:201
        else:
            dist = 1 - F.cosine_similarity(q, cand, dim=-1)          # nlsh/data.py:109
        if start >= k:
            top = dist.topk(k, largest=False)[1]                     # indexer.py:88-90
            result.append(torch.cat(rows_list)[top].tolist())
        else:                                                        # indexer.py:91-93 (F7): rows of the last key
            result.append(rows.tolist())
    return result, n_candidates
