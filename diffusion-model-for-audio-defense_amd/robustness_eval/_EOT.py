"""Expectation-over-transformation wrapper of the attack drivers (reference robustness_eval/_EOT.py:4-69), query
side: `EOT_size` stochastic evaluations of `model` (an AcousticSystem whose defender draws fresh diffusion noise on
every call) per input, `EOT_batch_size` of them per model call as one repeated batch; returns the mean scores, the
mean per-example loss, the gradient (None here) and every repeat's decision.

The HIP purifier is inference-only, so `use_grad=True` (white-box attacks: backward through the purifier) raises;
the black-box drivers (`black_box_attack.py:199`, NES, SirenAttack) construct the wrapper with `use_grad=False`."""
import torch
import torch.nn as nn


class EOT(nn.Module):

    def __init__(self, model, loss, EOT_size=1, EOT_batch_size=1, use_grad=True):
        super().__init__()
        self.model = model
        self.loss = loss
        self.EOT_size = EOT_size
        self.EOT_batch_size = EOT_batch_size
        self.EOT_num_batches = self.EOT_size // self.EOT_batch_size
        self.use_grad = use_grad

    def forward(self, x_batch, y_batch, EOT_size=None, EOT_batch_size=None, use_grad=None):
        EOT_size = EOT_size if EOT_size else self.EOT_size
        EOT_batch_size = EOT_batch_size if EOT_batch_size else self.EOT_batch_size
        EOT_num_batches = EOT_size // EOT_batch_size
        use_grad = use_grad if use_grad else self.use_grad
        if use_grad:
            raise NotImplementedError('EOT gradients need autograd through the purifier; the HIP engine is inference-only '
                                      '(construct the wrapper with use_grad=False, as the black-box drivers do)')
        n_audios = x_batch.size(0)
        scores = None
        loss = 0
        decisions = [[] for _ in range(n_audios)]
        with torch.no_grad():
            for EOT_index in range(EOT_num_batches):
                x_batch_repeat = x_batch.repeat(EOT_batch_size, 1, 1)
                y_batch_repeat = y_batch.repeat(EOT_batch_size)
                scores_EOT = self.model(x_batch_repeat)                   # (EOT_batch_size * n_audios, n_classes)
                decisions_EOT = scores_EOT.max(1, keepdim=True)[1]
                loss_EOT = self.loss(scores_EOT, y_batch_repeat)
                if EOT_index == 0:
                    scores = scores_EOT.view(EOT_batch_size, -1, scores_EOT.shape[1]).mean(0)
                    loss = loss_EOT.view(EOT_batch_size, -1).mean(0)
                else:
                    scores += scores_EOT.view(EOT_batch_size, -1, scores.shape[1]).mean(0)
                    loss += loss_EOT.view(EOT_batch_size, -1).mean(0)
                decisions_EOT = decisions_EOT.view(EOT_batch_size, -1).cpu().numpy()
                for ii in range(n_audios):
                    decisions[ii] += list(decisions_EOT[:, ii])
        scores = scores / EOT_num_batches
        loss = loss / EOT_num_batches
        return scores, loss, None, decisions
