"""Expectation over transformation for the query-only attack drivers (behaviour of the reference's
robustness_eval/_EOT.py:19-69 with use_grad=False; callers black_box_attack.py:199, _NES.py:36).

Contract kept: EOT(model, loss, EOT_size, EOT_batch_size, use_grad)(x [n,1,L], y [n]) evaluates every clip
(EOT_size // EOT_batch_size) * EOT_batch_size times under the model's own randomness and returns
    scores    [n, C]  the per-model-call means (EOT_batch_size repeats each), averaged over the calls,
    loss      [n]     the same average of the per-example loss,
    grad      None    (no autograd through the HIP purifier: use_grad=True raises),
    decisions n lists of the arg-max of every single evaluation, in evaluation order.
Built differently: all repeats are ONE batched query (AcousticSystem.query -> dmad_query_logits) instead of one model
call per EOT batch; only the reduction order of the reference is reproduced on the returned [repeats, n, C] block."""
import torch
import torch.nn as nn


class EOT(nn.Module):

    def __init__(self, model, loss, EOT_size=1, EOT_batch_size=1, use_grad=True):
        super().__init__()
        self.model, self.loss = model, loss
        self.EOT_size, self.EOT_batch_size = EOT_size, EOT_batch_size
        self.EOT_num_batches = EOT_size // EOT_batch_size
        self.use_grad = use_grad

    @torch.no_grad()
    def _all_repeats(self, x, per_call, calls):
        """logits [calls * per_call, n, C] of the model under fresh randomness per repeat."""
        if hasattr(self.model, 'query'):
            return self.model.query(x, calls * per_call, per_call=per_call)[0]
        blocks = [self.model(x.repeat(per_call, 1, 1)).view(per_call, x.shape[0], -1) for _ in range(calls)]
        return torch.cat(blocks, 0)

    def forward(self, x_batch, y_batch, EOT_size=None, EOT_batch_size=None, use_grad=None):
        size = EOT_size or self.EOT_size
        per_call = EOT_batch_size or self.EOT_batch_size
        calls = size // per_call
        if use_grad or self.use_grad:
            raise NotImplementedError('EOT gradients need autograd through the purifier; the HIP engine is inference-only '
                                      '(construct the wrapper with use_grad=False, as the black-box drivers do)')
        n = x_batch.shape[0]
        logits = self._all_repeats(x_batch, per_call, calls)                     # [R, n, C]
        R, C = logits.shape[0], logits.shape[-1]
        per_eval_loss = self.loss(logits.reshape(R * n, C), y_batch.repeat(R)).view(calls, per_call, n)
        call_scores = logits.view(calls, per_call, n, C).mean(1)                 # mean inside each model call ...
        call_loss = per_eval_loss.mean(1)
        scores, loss = call_scores[0].clone(), call_loss[0].clone()
        for c in range(1, calls):                                                # ... summed call by call, then averaged
            scores += call_scores[c]
            loss += call_loss[c]
        picks = logits.argmax(-1).cpu().numpy()                                  # [R, n]
        decisions = [list(picks[:, i]) for i in range(n)]
        return scores / calls, loss / calls, None, decisions
