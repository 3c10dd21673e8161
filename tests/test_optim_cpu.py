"""Host logic of the fused optimizers that needs no GPU: the device-side step counters follow the
parameters when the trainable set changes (vit_torch_amd/optim.py `_StepCounters`)."""
import torch

from vit_torch_amd.optim import _StepCounters


def test_step_counter_pieces_are_stable_after_an_unfreeze():
    """once a run has been cut into pieces with different step counts (parameters unfrozen
    under Adam / Adagrad / AdaBelief), later steps over the SAME runs must hand back the SAME tick
    tensors without a read-back — a captured HIP graph keeps advancing exactly those tensors."""
    c = _StepCounters()
    dev = torch.device("cpu")
    first = c.spans([(64, 128)], dev)                     # only the head trains
    assert [(s, e) for s, e, _ in first] == [(64, 128)]
    first[0][2].add_(3.0)                                 # three steps taken (the kernels advance the tick)
    merged = c.spans([(0, 128)], dev)                     # backbone unfrozen: one run, two histories
    assert [(s, e, float(t)) for s, e, t in merged] == [(0, 64, 0.0), (64, 128, 3.0)]
    ids = [id(t) for _, _, t in merged]
    for _ in range(3):                                    # unchanged runs: same pieces, same tensors
        again = c.spans([(0, 128)], dev)
        assert [id(t) for _, _, t in again] == ids
        assert [(s, e) for s, e, _ in again] == [(0, 64), (64, 128)]
    for _, _, t in merged:
        t.add_(1.0)
    refrozen = c.spans([(64, 128)], dev)                  # the set shrinks again: counts carried over
    assert [(s, e, float(t)) for s, e, t in refrozen] == [(64, 128, 4.0)]
