"""Checkpoint compatibility with the reference's Trainer.

`Trainer.save_checkpoint` (flow3d/trainer.py:126-140) writes
    {"model": SceneModel.state_dict(), "optimizers": {...}, "schedulers": {...}, "global_step", "epoch",
     "move_model": MoveModel.state_dict()}
and `Trainer.init_from_checkpoint` (trainer.py:142-170) rebuilds the scene from `ckpt["model"]`
(`SceneModel.init_from_state_dict`, scene_model.py:145-160: only the `fg.params.*`, `bg.params.*`,
`motion_bases.params.*`, `Ks`, `w2cs` entries are read - the `move_model.*` entries inside it are ignored because the
constructor makes a fresh MoveModel), then loads `ckpt["move_model"]` with `strict=False` after dropping its
`time_params` whenever `time_params.shape[0] != num_fg` - the tensor is [1, 8], so it is dropped for every scene with
more than one foreground Gaussian and the exposure half-widths restart from 0.5 (reference quirk, reproduced).
"""
from __future__ import annotations

import torch

from .move_model import MoveModel
from .scene_model import SceneModel


def load_reference_checkpoint(path_or_dict, device=None):
    """-> (SceneModel, meta) with meta = {"global_step", "epoch", "optimizers", "schedulers"} (the last two as saved:
    per-parameter optimizer state dicts keyed like the reference's, for the caller's own optimizers)."""
    ckpt = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, "read") \
        else path_or_dict
    if "model" not in ckpt:
        raise KeyError("not a reference Trainer checkpoint: no 'model' entry (flow3d/trainer.py:126-140)")
    model = SceneModel.init_from_state_dict(ckpt["model"])
    if device is not None:
        model = model.to(device)
    if "move_model" in ckpt:  # trainer.py:153-161
        num_fg = model.num_fg_gaussians
        mm = MoveModel(num_fg, camera_mode="linear")
        sd = dict(ckpt["move_model"])
        if sd["time_params"].shape[0] != num_fg:
            sd.pop("time_params")
        mm.load_state_dict(sd, strict=False)
        model.move_model = mm.to(model.Ks.device)
    meta = {"global_step": ckpt.get("global_step", 0), "epoch": ckpt.get("epoch", 0),
            "optimizers": ckpt.get("optimizers"), "schedulers": ckpt.get("schedulers")}
    return model, meta


def reference_checkpoint_dict(model: SceneModel, optimizers: dict | None = None, schedulers: dict | None = None,
                              global_step: int = 0, epoch: int = 0) -> dict:
    """The dict `Trainer.save_checkpoint` would `torch.save` for this model (trainer.py:126-140)."""
    return {"model": model.state_dict(),
            "optimizers": {k: v.state_dict() for k, v in (optimizers or {}).items()},
            "schedulers": {k: v.state_dict() for k, v in (schedulers or {}).items()},
            "global_step": global_step, "epoch": epoch, "move_model": model.move_model.state_dict()}
