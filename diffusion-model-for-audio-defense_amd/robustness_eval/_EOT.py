"""Expectation over transformation for the query-only attack drivers (behaviour of the reference's
robustness_eval/_EOT.py:19-69 with use_grad=False; callers black_box_attack.py:199, _NES.py:36).

Contract kept: EOT(model, loss, EOT_size, EOT_batch_size, use_grad)(x [n,1,L], y [n]) evaluates every clip
(EOT_size // EOT_batch_size) * EOT_batch_size times under the model's own randomness and returns
    scores    [n, C]  the per-model-call means (EOT_batch_size repeats each), averaged over the calls,
    loss      [n]     the same average of the per-example loss,
    grad      None with use_grad=False; with use_grad=True the gradient of the loss w.r.t. x, averaged like the scores
              (the model is then called once per EOT batch on a leaf that requires grad, which sends the HIP-backed stages
              down their differentiation branch, dmad_hip/autograd.py),
    decisions n lists of the arg-max of every single evaluation, in evaluation order.
Built differently: without gradients all repeats are ONE batched query (AcousticSystem.query -> dmad_query_logits) instead
of one model call per EOT batch; only the reduction order of the reference is reproduced on the returned [repeats, n, C] block."""
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

    def _forward_with_grad(self, x_batch, y_batch, per_call, calls):
        """reference l.36-66 with use_grad: one model call per EOT batch on a leaf that requires grad."""
        n, ch, L = x_batch.shape
        scores = loss = grad = None
        decisions = [[] for _ in range(n)]
        with torch.enable_grad():
            for _ in range(calls):
                x_rep = x_batch.detach().repeat(per_call, 1, 1).requires_grad_(True)
                out = self.model(x_rep)
                l_rep = self.loss(out, y_batch.repeat(per_call))
                l_rep.backward(torch.ones_like(l_rep))
                s_c = out.detach().view(per_call, n, -1).mean(0)
                l_c = l_rep.detach().view(per_call, n).mean(0)
                g_c = x_rep.grad.view(per_call, n, ch, L).mean(0)
                scores, loss, grad = (s_c, l_c, g_c) if scores is None else (scores + s_c, loss + l_c, grad + g_c)
                picks = out.detach().argmax(1).view(per_call, n).cpu().numpy()
                for i in range(n):
                    decisions[i] += list(picks[:, i])
        return scores / calls, loss / calls, grad / calls, decisions

    def forward(self, x_batch, y_batch, EOT_size=None, EOT_batch_size=None, use_grad=None):
        size = EOT_size or self.EOT_size
        per_call = EOT_batch_size or self.EOT_batch_size
        calls = size // per_call
        n = x_batch.shape[0]
        if use_grad or self.use_grad:
            return self._forward_with_grad(x_batch, y_batch, per_call, calls)
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
