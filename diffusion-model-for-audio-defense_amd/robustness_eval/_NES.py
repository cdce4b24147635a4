"""Natural-evolution-strategies gradient estimate of the black-box drivers (reference robustness_eval/_NES.py:5-55):
antithetic Gaussian probes around x, each scored through the EOT wrapper (queries only, no autograd), gradient =
mean(loss * noise) / sigma.  Arithmetic and return values follow the reference, including its extra division of
the EOT means by the number of EOT batches (l.33-35)."""
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

    def forward(self, x, y):
        n_audios, n_channels, N = x.shape
        num_batches = self.samples_per_draw // self.samples_per_draw_batch_size
        for i in range(num_batches):
            noise = torch.randn([n_audios, self.samples_per_draw_batch_size // 2, n_channels, N], device=x.device)
            noise = torch.cat((noise, -noise), 1)
            if i == 0:
                noise = torch.cat((torch.zeros_like(x, device=x.device).unsqueeze(1), noise), 1)
            eval_input = noise * self.sigma + x.unsqueeze(1)
            eval_input = eval_input.view(-1, n_channels, N)
            per = self.samples_per_draw_batch_size + 1 if i == 0 else self.samples_per_draw_batch_size
            eval_y = torch.cat([torch.full((per,), int(y_), dtype=torch.long, device=x.device) for y_ in y])
            scores, loss, _, decisions = self.EOT_wrapper(eval_input, eval_y)
            EOT_num_batches = int(self.EOT_wrapper.EOT_size // self.EOT_wrapper.EOT_batch_size)
            loss = loss / EOT_num_batches
            scores = scores / EOT_num_batches
            loss = loss.view(n_audios, -1)
            scores = scores.view(n_audios, -1, scores.shape[1])
            if i == 0:
                adver_loss = loss[..., 0]
                loss = loss[..., 1:]
                adver_score = scores[:, 0, :]
                noise = noise[:, 1:, :, :]
                grad = torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
                mean_loss = loss.mean(1)
                predicts = resolve_prediction(decisions).reshape(n_audios, -1)
                predict = predicts[:, 0]
            else:
                grad += torch.mean(loss.unsqueeze(2).unsqueeze(3) * noise, 1)
                mean_loss += loss.mean(1)
        grad = grad / self.sigma / num_batches
        mean_loss = mean_loss / num_batches
        return mean_loss, grad, adver_loss, adver_score, predict
