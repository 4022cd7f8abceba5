"""rewards.multi_score aggregation on the host (RW:1043-1093): scorers of the plugin surface may return a tensor, an
ndarray or a list of floats, in any order; the reference adds them element by element (RW:1084-1092)."""
import itertools

import numpy as np
import pytest
import torch

from adv_grpo_amd import rewards


def _reference_sum(parts):
    """RW:1084-1092 restated: weighted lists added element by element."""
    total = []
    for w, scores in parts:
        weighted = [w * s for s in scores]
        total = weighted if not total else [a + b for a, b in zip(total, weighted)]
    return [float(t) for t in total]


@pytest.mark.parametrize("order", list(itertools.permutations(["la", "tb", "nc"])))
def test_multi_score_mixes_list_tensor_and_ndarray_scorers_in_any_order(order):
    vals = {"la": [0.25, -1.5, 3.0], "tb": torch.tensor([1.0, 2.0, -0.5]), "nc": np.array([0.5, 0.125, 4.0], dtype=np.float32)}
    weights = {"la": 0.7, "tb": 0.3, "nc": 0.5}
    saved = dict(rewards.score_functions)
    try:
        for k in vals:
            rewards.register_scorer(k, (lambda kk: (lambda: (lambda images, prompts, metadata: (vals[kk], {}))))(k))
        det, meta = rewards.multi_score("cpu", {k: weights[k] for k in order})(None, ["p"] * 3, [{}] * 3)
    finally:
        rewards.score_functions.clear()
        rewards.score_functions.update(saved)
    assert meta == {}
    want = _reference_sum([(weights[k], [float(x) for x in vals[k]]) for k in order])
    got = [float(x) for x in det["avg"]]
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=1e-7)       # f32 accumulation where a tensor is involved
    for k in order:
        assert det[k] is vals[k]


def test_multi_score_ocr_then_pickscore_shape():
    """The multi-reward preset of BASELINE config 4 ({"pickscore": .5, "ocr": .5}, config/grpo.py:379-427) in BOTH key orders."""
    saved = dict(rewards.score_functions)
    try:
        rewards.register_scorer("ocr", lambda: (lambda images, prompts, metadata: ([1.0, 0.0], {})))
        rewards.register_scorer("pickscore", lambda: (lambda images, prompts, metadata: (torch.tensor([0.8, 0.9]), {})))
        a = rewards.multi_score("cpu", {"ocr": 0.5, "pickscore": 0.5})(None, ["x", "y"], [{}] * 2)[0]["avg"]
        b = rewards.multi_score("cpu", {"pickscore": 0.5, "ocr": 0.5})(None, ["x", "y"], [{}] * 2)[0]["avg"]
    finally:
        rewards.score_functions.clear()
        rewards.score_functions.update(saved)
    assert torch.allclose(torch.as_tensor(a), torch.tensor([0.9, 0.45])) and torch.allclose(torch.as_tensor(b), torch.as_tensor(a))
