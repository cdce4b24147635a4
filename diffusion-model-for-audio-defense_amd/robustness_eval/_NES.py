"""NES gradient estimate for the query-only attack drivers (behaviour of the reference's robustness_eval/_NES.py:15-55;
caller black_box_attack.py:186-190).

For every clip: `samples_per_draw` antithetic Gaussian probes x +- sigma * u (the first draw batch also carries the
unperturbed clip in slot 0), each scored through the EOT wrapper; the estimate is  grad = E[loss * u] / sigma.
Returns (mean probe loss [n], grad [n,1,L], loss of the unperturbed clip [n], its scores [n,C], its majority decision [n]).
One quirk of the reference is part of the contract and kept: the EOT wrapper already returns means over its model
calls, and NES divides them by the number of EOT calls once more (ref l.33-35)."""
import torch
import torch.nn as nn

from ._utils import resolve_prediction


class NES(nn.Module):

    def __init__(self, samples_per_draw, samples_per_draw_batch, sigma, EOT_wrapper):
        super().__init__()
        self.samples_per_draw = samples_per_draw
        self.samples_per_draw_batch_size = samples_per_draw_batch
        self.sigma = sigma
        self.EOT_wrapper = EOT_wrapper

    def _probe(self, x, y, with_origin):
        """One draw batch -> (u [n, P, 1, L] probe directions, loss [n, P(+1)], scores [n, P(+1), C], decisions)."""
        n, ch, L = x.shape
        half = torch.randn([n, self.samples_per_draw_batch_size // 2, ch, L], device=x.device)
        u = torch.cat((half, -half), 1)
        probes = torch.cat((torch.zeros_like(x).unsqueeze(1), u), 1) if with_origin else u
        per_clip = probes.shape[1]
        queries = (probes * self.sigma + x.unsqueeze(1)).view(-1, ch, L)
        labels = torch.as_tensor(y, device=x.device).long().repeat_interleave(per_clip)
        scores, loss, _, decisions = self.EOT_wrapper(queries, labels)
        again = int(self.EOT_wrapper.EOT_size // self.EOT_wrapper.EOT_batch_size)
        return u, (loss / again).view(n, per_clip), (scores / again).view(n, per_clip, -1), decisions

    def forward(self, x, y):
        n = x.shape[0]
        draws = self.samples_per_draw // self.samples_per_draw_batch_size
        u, loss, scores, decisions = self._probe(x, y, with_origin=True)
        adver_loss, adver_score = loss[:, 0], scores[:, 0, :]
        predict = resolve_prediction(decisions).reshape(n, -1)[:, 0]
        loss = loss[:, 1:]
        grad = (loss[:, :, None, None] * u).mean(1)
        mean_loss = loss.mean(1)
        for _ in range(1, draws):
            u, loss, _, _ = self._probe(x, y, with_origin=False)
            grad += (loss[:, :, None, None] * u).mean(1)
            mean_loss += loss.mean(1)
        return mean_loss / draws, grad / self.sigma / draws, adver_loss, adver_score, predict
