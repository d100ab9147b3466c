import numpy as np
import torch


def rel_err(a, b):
    """max|a-b| / max|b|  (the parity norm of SURVEY.md section 7 'Hard parts')."""
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / den)


def frac_bad(a, b, tol):
    """fraction of elements whose error exceeds tol*max|b| (isolated discrete-decision flips)."""
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30)
    return float((np.abs(a - b) > tol * den).mean())


def static_inputs(N, W, H, seed, dtype=torch.float32, scale_mul=3.0, D=3):
    from deblur4dgs_amd.synth import make_scene

    sc = make_scene(N, 0, 1, 1, W, H, seed, dtype=torch.float64, D=D)
    out = dict(means=sc["means"], quats=sc["quats"], scales=torch.exp(sc["scales"]) * scale_mul,
               opac=torch.sigmoid(sc["opacities"]), colors=torch.sigmoid(sc["colors"]), V=sc["viewmat"], K=sc["K"])
    return {k: v.to(dtype) for k, v in out.items()}
