"""Multi-scale loss container (interface of reference loss/base.py:12-33): a loss module is
called with the list of predictions (main output first, then the side outputs) and the list
of matching targets and answers with one `(loss_sum, n_elements)` pair per scale."""
import torch


class LossBase(torch.nn.Module):
    def _speculative_single(self, kind, input_, target, mask=None, weights=None, param=0.0):
        """one item through the multi-loss call with THIS instance's learned expectation of the
        upstream gradient (loss/_multi.py): forward sum + gradient in one pass, confirmed or
        recomputed in backward -> (loss sum, element count, aux)"""
        from . import _multi
        spec = self.__dict__.get('_spec')
        if spec is None:
            spec = self.__dict__['_spec'] = _multi.SpecState(1)
        item = {'kind': kind, 'pred': input_, 'target': target, 'mask': mask, 'weights': weights,
                'param': param, 'total': 0}
        res = _multi.multi_loss([item], 1, spec)
        return res.sums[0], res.counts[0], res.aux[0]

    def _can_speculate(self, input_) -> bool:
        """forward-written gradient for THIS call?  Not for a list of several scales: their sums
        are divided by the SUMMED counts afterwards (base.py:161-182), an upstream factor this
        instance's single record cannot predict per scale — those calls take the two-kernel path
        at once instead of paying for mispredicted gradients until the record switches itself
        off (the task helpers put all scales into one multi-loss call with one total instead)."""
        from . import _functional as F_
        if self.__dict__.get('_several_scales', False):
            return False
        return input_.is_cuda and input_.numel() > 0 and F_.speculation_enabled() and F_.wants_gradient(input_)

    def _compute_loss(self, input_, target):
        """one scale -> (loss, number of loss elements)"""
        raise NotImplementedError(f'{type(self).__name__} must define _compute_loss')

    def forward(self, input_tensors, target_tensors, expected_scales=None):
        """`expected_scales` (extension, optional): per scale the gradient the caller's
        reduction is going to send back to that scale's loss sum (a device scalar, see
        loss/_functional.py `expected_scale`); losses that can write their gradient in the
        forward pass do so, the others ignore it"""
        pairs = []
        # (the flag lives for the duration of this call only — also when a scale raises — and is
        # restored, not cleared: a loss called from inside another loss's forward keeps its own)
        before = self.__dict__.get('_several_scales', False)
        self.__dict__['_several_scales'] = len(input_tensors) > 1
        try:
            for i, (prediction, target) in enumerate(zip(input_tensors, target_tensors)):
                if expected_scales is None or expected_scales[i] is None:
                    pairs.append(self._compute_loss(prediction, target))
                else:
                    pairs.append(self._compute_loss(prediction, target,
                                                    expected_scale=expected_scales[i]))
        finally:
            self.__dict__['_several_scales'] = before
        return tuple(pairs)
