"""Oracle: literal restatement of the accumulation loop of Trainer._prepare_control_step
(flow3d/trainer.py:967-989) for ONE rendered view.  Test infrastructure (see oracle/__init__.py).
`trainer.py` itself cannot be imported here (nerfview / pytorch_msssim / tensorboard are absent), so these 20
lines are restated; they are plain indexing arithmetic with no third-party call."""
import torch


@torch.no_grad()
def prepare_control_step(running_stats, current_xys_grads, current_radii, img_wh, batch_size):
    """current_xys_grads / current_radii: lists (one per sub-sample) of [1,N,2] / [1,N] tensors."""
    for ii in range(len(current_xys_grads)):
        sel = current_radii[ii] > 0
        gidcs = torch.where(sel)[1]
        xys_grad = current_xys_grads[ii].clone()
        xys_grad[..., 0] *= img_wh[0] / 2.0 * batch_size * len(current_xys_grads)
        xys_grad[..., 1] *= img_wh[1] / 2.0 * batch_size * len(current_xys_grads)
        running_stats["xys_grad_norm_acc"].index_add_(0, gidcs, xys_grad[sel].norm(dim=-1))
        running_stats["vis_count"].index_add_(0, gidcs, torch.ones_like(gidcs, dtype=torch.int64))
        max_radii = torch.maximum(running_stats["max_radii"].index_select(0, gidcs),
                                  current_radii[ii][sel] / max(img_wh))
        running_stats["max_radii"].index_put((gidcs,), max_radii)
    return running_stats
