"""TEST STAND-IN for the caller's own train_utils.py (see configs.py next to this file).  The overlay executes the caller's module and
then replaces the hot-path functions by the device implementations of the same names; the two helpers below stand for "everything
else the caller's module defines", which must survive the overlay unchanged."""


def tree_len(tree):
    return len(tree)


def compute_data_loss(batch, renderings, config):           # must be REPLACED by the overlay (a call would fail loudly)
    raise AssertionError("the overlay did not replace compute_data_loss")
