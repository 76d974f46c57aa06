"""Recall metric, restating reference nlsh/metrics.py:4-25 (exact rational per query)."""
from typing import Callable, List, Optional, Sequence, Union


def _recall(y_true: Sequence[int], y_pred: Sequence[int]) -> float:
    truth = {int(t) for t in y_true}
    return len(truth.intersection(int(p) for p in y_pred)) / len(y_true)


def calculate_recall(y_true, y_pred, reduce_func: Optional[Callable] = None) -> Union[List[float], float]:
    if len(y_true) != len(y_pred):
        raise AssertionError("y_true and y_pred differ in length")
    per_query = [_recall(t, p) for t, p in zip(y_true, y_pred)]
    return per_query if reduce_func is None else reduce_func(per_query)
