"""Multi-scale loss container (interface of reference loss/base.py:12-33): a loss module is
called with the list of predictions (main output first, then the side outputs) and the list
of matching targets and answers with one `(loss_sum, n_elements)` pair per scale."""
import torch


class LossBase(torch.nn.Module):
    def _compute_loss(self, input_, target):
        """one scale -> (loss, number of loss elements)"""
        raise NotImplementedError(f'{type(self).__name__} must define _compute_loss')

    def forward(self, input_tensors, target_tensors):
        pairs = []
        for prediction, target in zip(input_tensors, target_tensors):
            pairs.append(self._compute_loss(prediction, target))
        return tuple(pairs)
