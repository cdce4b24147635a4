"""Query-path helpers of the black-box attack drivers (reference robustness_eval/_utils.py:104-136): the per-example
loss of the speech-commands task and the majority decision over EOT repeats."""
from collections import Counter

import numpy as np
import torch.nn as nn

__all__ = ['resolve_loss', 'resolve_prediction']


def resolve_loss(loss_name='Entropy', targeted=False, confidence=0., task='CSI', threshold=None, clip_max=True):
    assert loss_name in ['Entropy', 'Margin']
    assert task in ['SCR', 'SV']   # speech commands recognition / speaker verification
    if task == 'SCR':
        loss = nn.CrossEntropyLoss(reduction='none')
    else:
        raise NotImplementedError(f'unsupported task yet: {task}!')
    grad_sign = -1 if targeted else 1
    return loss, grad_sign


def resolve_prediction(decisions):
    predict = []
    for d in decisions:
        counts = Counter(d)
        predict.append(counts.most_common(1)[0][0])
    return np.array(predict)
