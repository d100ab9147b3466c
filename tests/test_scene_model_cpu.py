"""Host-side mirror (seam S2) without a GPU: state_dict key compatibility with the reference's checkpoints
(flow3d/scene_model.py:145-160, flow3d/trainer.py:126-170), and the no-fallback rule."""
import pytest
import torch

from deblur4dgs_amd.scene_model import GaussianParams, MotionBases, SceneModel
from deblur4dgs_amd.synth import make_scene


def _model():
    sc = make_scene(50, 20, 3, 1, 32, 32, seed=1, T=6)
    keys = ("means", "quats", "scales", "colors", "opacities")
    fg = GaussianParams(*[sc[k][:20].clone() for k in keys], motion_coefs=sc["motion_coefs"].clone())
    bg = GaussianParams(*[sc[k][20:].clone() for k in keys])
    return SceneModel(sc["K"][None], sc["viewmat"][None], fg, MotionBases(sc["rots"], sc["transls"]), bg), sc


def test_state_dict_keys_match_reference_checkpoint_layout():
    m, _ = _model()
    sd = m.state_dict()
    expected = {f"fg.params.{k}" for k in ("means", "quats", "scales", "colors", "opacities", "motion_coefs")}
    expected |= {f"bg.params.{k}" for k in ("means", "quats", "scales", "colors", "opacities")}
    expected |= {"motion_bases.params.rots", "motion_bases.params.transls", "Ks", "w2cs", "bg_scene_scale",
                 "fg.scene_center", "fg.scene_scale", "bg.scene_center", "bg.scene_scale", "move_model.time_params"}
    assert expected <= set(sd), expected - set(sd)
    for k in ("move_model.RT_main.0.weight", "move_model.RT_main.8.bias", "move_model.RT_head0.2.weight",
              "move_model.RT_head1.0.bias"):
        assert k in sd
    assert sd["move_model.RT_main.0.weight"].shape == (64, 66) and sd["move_model.time_params"].shape == (1, 8)
    assert sum(p.numel() for p in m.move_model.parameters()) == 30036  # SURVEY 2.1 [PROBED]


def test_init_from_state_dict_roundtrip():
    m, _ = _model()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = SceneModel.init_from_state_dict(sd)
    assert m2.num_fg_gaussians == 20 and m2.num_bg_gaussians == 30 and m2.num_motion_bases == 3 and m2.num_frames == 6
    for k in ("fg.params.means", "bg.params.colors", "motion_bases.params.rots", "Ks"):
        assert torch.equal(m2.state_dict()[k], sd[k])
    # a foreground-only checkpoint (no bg.* keys) loads with bg = None
    sd_fg = {k: v for k, v in sd.items() if not k.startswith("bg.")}
    assert SceneModel.init_from_state_dict(sd_fg).bg is None


def test_render_on_cpu_fails_loudly():
    m, sc = _model()
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X"):
        m.render(2, sc["viewmat"][None], sc["K"][None], (32, 32))


def test_reference_trainer_checkpoint_loads_with_its_quirks(tmp_path):
    """flow3d/trainer.py:126-170: scene from ckpt["model"], MoveModel weights from ckpt["move_model"], `time_params`
    dropped (shape[0] == 1 != num_fg) and therefore back at its 0.5 initialisation; global_step / epoch returned."""
    from deblur4dgs_amd.checkpoint import load_reference_checkpoint, reference_checkpoint_dict

    m, _ = _model()
    with torch.no_grad():
        for p in m.move_model.parameters():
            p.add_(0.1 * torch.randn_like(p))
        m.move_model.time_params.copy_(torch.tensor([[0.5, 0.3, 0.45, 0.6, 0.2, 0.5, 0.7, 0.5]]))
    opt = {"fg.params.means": torch.optim.Adam([m.fg.params["means"]], lr=1e-3)}
    ck = reference_checkpoint_dict(m, optimizers=opt, global_step=1234, epoch=7)
    assert set(ck) == {"model", "optimizers", "schedulers", "global_step", "epoch", "move_model"}
    assert "move_model.RT_main.0.weight" in ck["model"]  # the submodule's entries ride in the model dict too (ignored on load)
    path = tmp_path / "last.ckpt"
    torch.save(ck, path)
    m2, meta = load_reference_checkpoint(str(path))
    assert meta["global_step"] == 1234 and meta["epoch"] == 7 and "fg.params.means" in meta["optimizers"]
    for k in ("fg.params.means", "fg.params.motion_coefs", "bg.params.scales", "motion_bases.params.rots", "Ks", "w2cs"):
        assert torch.equal(m2.state_dict()[k], m.state_dict()[k]), k
    for k, v in m.move_model.state_dict().items():
        if k == "time_params":
            assert torch.equal(m2.move_model.time_params, torch.full((1, 8), 0.5))  # the reset quirk (trainer.py:156-157)
        else:
            assert torch.equal(m2.move_model.state_dict()[k], v), k
    # a checkpoint without the separate "move_model" entry keeps a freshly initialised MoveModel (zero heads)
    m3, _ = load_reference_checkpoint({"model": ck["model"]})
    assert float(m3.move_model.RT_head0[-1].weight.abs().sum()) == 0.0
    with pytest.raises(KeyError):
        load_reference_checkpoint({"state_dict": {}})
