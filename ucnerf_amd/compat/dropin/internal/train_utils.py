"""`internal.train_utils` of the drop-in overlay: the reference's own module (loaded from the caller's `internal/`, so
every helper train.py / datasets.py / eval.py use -- tree_len, GradientScaler, img_warping with host tensors, the
orientation / normal losses ... -- is there unchanged) with the functions on the training step's hot path replaced by
this repo's device implementations of the same name and signature (train.py:102, 173-216, 221):

    compute_data_loss       one set of launches for all levels, stats fetched lazily (no host sync inside the step)
    anti_interlevel_loss    ucn_interlevel_loss
    distortion_loss         ucn_distortion_loss, O(S) per ray
    hash_decay_loss         reads ray_history[...]['loss_hash_decay'] (ucn_hash_decay inside the model)
    sky_loss, transformIdentityLoss
    clip_gradients          one launch for every gradient's nan_to_num
    create_optimizer        FusedAdam (a torch.optim.Adam subclass with torch's state layout)
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

from . import UPSTREAM as _UPSTREAM

_spec = _ilu.spec_from_file_location("internal._upstream_train_utils", _os.path.join(_UPSTREAM, "train_utils.py"))
_up = _ilu.module_from_spec(_spec)
_sys.modules["internal._upstream_train_utils"] = _up
_spec.loader.exec_module(_up)
globals().update({k: v for k, v in vars(_up).items() if not k.startswith("__")})
upstream = _up

from ucnerf_amd.internal.train_utils import (FusedAdam, LazyStats, anti_interlevel_loss, clip_gradients,  # noqa: E402,F401
                                             compute_data_loss, create_optimizer, distortion_loss, hash_decay_loss,
                                             sky_loss, transformIdentityLoss)
