"""Helpers of the query-path attack drivers (contract of the reference's robustness_eval/_utils.py:104-136; callers
black_box_attack.py:192,249,418 and _NES.py:49).

resolve_loss(...) -> (per-example loss module, sign of the gradient step): only the speech-commands task ('SCR') has a
loss — unreduced cross-entropy — and a targeted attack descends (sign -1) where an untargeted one ascends (+1).  The
remaining positional parameters are accepted because the callers pass them; they do not change the result.
resolve_prediction(decisions) -> the majority label of every row of EOT decisions (first-seen label wins a tie)."""
import numpy as np
import torch.nn as nn

__all__ = ['resolve_loss', 'resolve_prediction']

_LOSS_NAMES = ('Entropy', 'Margin')
_TASKS = ('SCR', 'SV')          # speech-commands recognition / speaker verification


def resolve_loss(loss_name='Entropy', targeted=False, confidence=0., task='CSI', threshold=None, clip_max=True):
    assert loss_name in _LOSS_NAMES, loss_name
    assert task in _TASKS, task
    if task != 'SCR':
        raise NotImplementedError('no loss for task %r: only SCR (speech commands) is supported' % (task,))
    return nn.CrossEntropyLoss(reduction='none'), (-1 if targeted else 1)


def resolve_prediction(decisions):
    majority = []
    for row in decisions:
        tally = {}
        for label in row:                      # dicts keep insertion order: the first-seen label wins a tie, as Counter does
            tally[label] = tally.get(label, 0) + 1
        majority.append(max(tally, key=tally.get))
    return np.array(majority)
