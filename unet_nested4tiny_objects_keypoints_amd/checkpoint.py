"""Checkpoint compatibility with the reference trainer.

The module tree and parameter names of :class:`UNet_Nested` equal the reference's, so a reference-trained
``state_dict`` loads directly.  This file mirrors the two on-disk formats the reference trainer reads and writes
(/root/reference/trainer/trainer.py):

* ``*.pth`` -- a bare ``state_dict`` (``torch.save(model_state_dic, ...)``, trainer.py:240-249, loaded at :415-419);
  the trainer names it ``best_epoch_{epoch}_heatmaploss_{h}_landmarkloss_{l}.pth``;
* ``*.tar`` -- ``{'model_state_dict', 'optimizer_state_dict', 'epoch'}`` (loaded at trainer.py:403-413).

Files saved from a ``nn.DataParallel`` wrapper without unwrapping carry a ``module.`` prefix on every key; it is
stripped on load (the reference unwraps with ``model.module`` when ``device_count > 1``, trainer.py:240).
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model


def _strip_module_prefix(state):
    if state and all(k.startswith("module.") for k in state):
        return {k[len("module."):]: v for k, v in state.items()}
    return state


def best_model_name(epoch, heatmap_loss, landmark_loss) -> str:
    return "best_epoch_{}_heatmaploss_{}_landmarkloss_{}.pth".format(epoch, heatmap_loss, landmark_loss)


def save_best(model, save_dir: str, epoch, heatmap_loss, landmark_loss) -> str:
    """trainer.py:240-249: the unwrapped state_dict as a .pth named after the validation losses."""
    path = os.path.join(save_dir, best_model_name(epoch, heatmap_loss, landmark_loss))
    torch.save(_unwrap(model).state_dict(), path)
    return path


def save_checkpoint(model, optimizer, epoch: int, path: str) -> str:
    """The .tar layout trainer.py:403-413 resumes from."""
    torch.save({"model_state_dict": _unwrap(model).state_dict(),
                "optimizer_state_dict": None if optimizer is None else optimizer.state_dict(),
                "epoch": int(epoch)}, path)
    return path


def resume(model, path: str, optimizer=None, resume_opt: bool = False, map_location: Optional[str] = "cpu") -> int:
    """trainer.py:399-419.  Loads `path` (suffix .tar or .pth) into `model` (and the optimizer when `resume_opt`);
    returns the epoch to start from (0 unless a .tar is resumed together with its optimizer state)."""
    suf = path.rsplit(".", 1)[-1]
    start_epoch = 0
    target = _unwrap(model)
    if suf == "tar":
        ckpt = torch.load(path, map_location=map_location)
        target.load_state_dict(_strip_module_prefix(ckpt["model_state_dict"]))
        if resume_opt:
            if optimizer is None:
                raise ValueError("resume_opt needs the optimizer")
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            start_epoch = int(ckpt["epoch"]) + 1
    elif suf == "pth":
        target.load_state_dict(_strip_module_prefix(torch.load(path, map_location=map_location)))
    else:
        raise ValueError("unknown checkpoint suffix %r (the reference trainer reads .tar and .pth)" % suf)
    return start_epoch
