"""Learning-rate schedules of ``thor.lr`` (src/thor/lr.py); ``lr_kwargs.func_name`` (train.py:189-193) can point here."""
import numpy as np


def edm2_learning_rate_schedule(cur_ndata, batch_size, ref_lr, ref_batches, rampup_Mdata):
    """src/thor/lr.py:6-14 (inverse-sqrt decay after ref_batches, linear ramp-up over rampup_Mdata million items)."""
    lr = ref_lr
    if ref_batches > 0:
        lr /= np.sqrt(max(cur_ndata / (ref_batches * batch_size), 1))
    if rampup_Mdata > 0:
        lr *= min(cur_ndata / (rampup_Mdata * 1e6), 1)
    return lr


def linear_learning_rate_schedule(cur_ndata, total_ndata, ref_lr):
    """src/thor/lr.py:17-19: ref_lr * (1 - cur/total)."""
    return ref_lr * (1 - cur_ndata / total_ndata)
