"""Learning-rate schedule of ``thor.lr`` that the training recipe uses (src/thor/lr.py:17-19; selected through
``lr_kwargs.func_name``, train.py:189-193).  The reference's other schedule has no caller and is not provided."""


def linear_learning_rate_schedule(cur_ndata, total_ndata, ref_lr):
    """Linear decay to zero over the run: ref_lr at cur_ndata = 0, 0 at cur_ndata = total_ndata."""
    return ref_lr * (1 - cur_ndata / total_ndata)
