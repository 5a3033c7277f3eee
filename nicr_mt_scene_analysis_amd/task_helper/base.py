"""Common machinery of the task helpers (interface of reference task_helper/base.py:17-210).

A task helper owns the losses and metrics of one task: `initialize(device)`, then per batch
`training_step` / `validation_step` -> (loss dict, log dict) and `validation_epoch_end` ->
(artifacts, examples, logs).  The `collect_*` helpers pair the main output and the side
outputs of a decoder with the targets of the matching resolution.

Element counts may be DEVICE scalars here (the HIP losses return them without a host sync);
`accumulate_losses` then stays on the device as well.
"""
import functools
import time
import warnings

import torch

from ..data.preprocessing.multiscale_supervision import get_downscale

TOTAL_LOSS_SUFFIX = '_total_loss'


def get_total_loss_key(key):
    return key + TOTAL_LOSS_SUFFIX


def _around_step(after):
    """decorator factory: run the step, then `after(result, seconds)` may amend the result"""
    def decorate(step):
        @functools.wraps(step)
        def run(*args, **kwargs):
            started = time.perf_counter()
            result = step(*args, **kwargs)
            after(result, time.perf_counter() - started)
            return result
        return run
    return decorate


class PackedLosses(dict):
    """loss dict whose values are elements of ONE device vector (the output of the multi-loss
    call): `packed` is that vector, `index[name]` the element — lets the log copy be one clone"""
    packed = None
    index = None


def append_detached_losses_to_logs(disabled=False):
    """(losses, logs) steps: copy every loss, detached, into the logs"""
    def after(result, _seconds):
        losses, logs = result
        if disabled:
            return
        if isinstance(losses, PackedLosses) and losses.packed is not None and \
                losses.keys() == losses.index.keys():
            copy = losses.packed.detach().clone().unbind(0)
            for name, i in losses.index.items():
                logs[name] = copy[i]
            return
        for name, value in losses.items():
            logs[name] = value.detach().clone()
    return _around_step(after)


def append_profile_to_logs(key, disabled=False):
    """store the wall time of the step under `key` in the trailing log dict of its result"""
    def after(result, seconds):
        if disabled:
            return
        logs = result[-1]
        assert isinstance(logs, dict)
        logs[key] = seconds
    return _around_step(after)


def _last_dim(output):
    """width of a decoder output (instance decoders hand over a tuple of tensors)"""
    if isinstance(output, tuple):
        output = output[0]
    if not isinstance(output, torch.Tensor):
        raise Exception("Error while determining downscale")
    return output.shape[-1]


class TaskHelperBase(torch.nn.Module):
    # Extension: the factor the trainer multiplies this task's total loss with before calling
    # backward — only the STARTING value of what the helper learns on the device from the upstream
    # gradients it sees (loss/_multi.py): the HIP losses write their gradient in the forward pass
    # for `w / sum_scales(n)` and re-do it in backward only when the real upstream gradient
    # differs, so a wrong value costs one recomputation, never correctness.  A dict gives one
    # factor per total ('instance_center', 'instance_offset', 'instance_orientation', 'semantic').
    backward_scale = 1.0

    def initialize(self, device):
        """create losses / metrics on `device`"""

    def spec_state(self, names):
        """the learned upstream factors of this helper's totals (loss/_multi.py `SpecState`),
        one per name, starting from `backward_scale`"""
        from ..loss import _multi
        st = self.__dict__.get('_spec_states')
        if st is None:
            st = self.__dict__['_spec_states'] = {}
        key = tuple(names)
        if key not in st:
            w = self.backward_scale
            init = [float(w.get(n, 1.0)) if isinstance(w, dict) else float(w) for n in names]
            st[key] = _multi.SpecState(len(names), init)
        return st[key]

    def multi_losses(self, items, item_names, total_names):
        """`items` (loss/_multi.py) through ONE forward call -> PackedLosses {item name: loss sum /
        count, total key: sum_scales loss sum / sum_scales count} — the reductions of reference
        base.py:161-182 done by the call itself (csrc k_multi_totals), no tensor op per scale"""
        from ..loss import _multi
        res = _multi.multi_loss(items, len(total_names), self.spec_state(total_names))
        n = len(items)
        per_item, totals = res.item_losses.unbind(0), res.total_losses.unbind(0)
        out = PackedLosses()
        out.packed, out.index = res.packed, {}
        for i, name in enumerate(item_names):
            out[name] = per_item[i]
            out.index[name] = n + i
        for t, name in enumerate(total_names):
            key = self.mark_as_total(name)
            out[key] = totals[t]
            out.index[key] = 2 * n + t
        return out

    # ---- pairing predictions and targets over the supervision scales ---------------------
    def collect_predictions_for_loss(self, predictions_post, predictions_post_key,
                                     side_outputs_key=None):
        """-> (tensors, keys, downscales): the main output as 'main', every side output that is
        not None as 'down_<factor>' (factor = main width // side width)"""
        main = predictions_post[predictions_post_key]
        tensors, keys, downscales = [main], ['main'], []
        sides = predictions_post[side_outputs_key] if side_outputs_key is not None else None
        for side in sides or ():
            if side is None:                # evaluation mode / decoder without side outputs
                continue
            factor = _last_dim(main) // _last_dim(side)
            tensors.append(side)
            keys.append(f'down_{factor}')
            downscales.append(factor)
        return tensors, keys, downscales

    def collect_targets_for_loss(self, batch, batch_key, downscales=None):
        """targets of the main resolution and of every downscale present in the batch"""
        targets = [batch[batch_key]]
        for factor in downscales or ():
            scaled = get_downscale(batch, factor)
            if scaled is not None:          # None: multiscale supervision disabled
                targets.append(scaled[batch_key])
        return targets

    def collect_predictions_and_targets_for_loss(self, batch, batch_key, predictions_post,
                                                 predictions_post_key, side_outputs_key=None):
        tensors, keys, downscales = self.collect_predictions_for_loss(
            predictions_post, predictions_post_key, side_outputs_key)
        return tensors, self.collect_targets_for_loss(batch, batch_key, downscales), keys

    # ---- reductions --------------------------------------------------------------------------
    def accumulate_losses(self, losses, n_elements):
        """sum(losses) / sum(n_elements) over the main and the side outputs (base.py:161-182);
        an empty selection returns the (zero) loss sum instead of dividing by zero."""
        loss_sum = torch.stack(list(losses)).sum()
        count = sum(n_elements)
        if isinstance(count, torch.Tensor):                 # device count: no host sync
            return torch.where(count == 0, loss_sum,
                               loss_sum / count.clamp(min=1).to(loss_sum.dtype))
        if count == 0:
            warnings.warn("Total number of loss elements is 0. Returning 0 as  "
                          "loss to avoid division by zero.")
            return loss_sum
        return loss_sum / count

    def mark_as_total(self, key):
        return get_total_loss_key(key)

    # ---- to be provided by the task ------------------------------------------------------------
    def training_step(self, batch, batch_idx, predictions_post):
        raise NotImplementedError

    def validation_step(self, batch, batch_idx, predictions_post):
        raise NotImplementedError

    def validation_epoch_end(self):
        raise NotImplementedError
