"""CPU oracle for the Deblur4DGS exposure-rasterizer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (`deblur4dgs_amd/`) imports this
directory; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
use it, and only as the checker / the reported CPU baseline.

What is pinned and what is not
------------------------------
* `oracle.deform`  (reference `flow3d/params.py:39-43,142-180`, `flow3d/transforms.py:41-53`,
  `flow3d/scene_model.py:58-120,352-353`) and the pure-torch half of `oracle.camera`
  (`flow3d/models/utils/spline_utils.py:12-54,177-215`, `flow3d/models/move_model.py:112-135`)
  are PINNED: `tests/golden/*.npz` hold outputs of the reference's own Python, generated in the
  build container by `tests/golden/gen_golden.py` (which imports `/root/reference`), and
  `tests/test_oracle_golden.py` checks this restatement against them.
* `oracle.raster` restates `gsplat==1.1.1` `rasterization(packed=False)` (reference call site
  `flow3d/scene_model.py:360-373`; pinned dependency `requirements.txt:137`).  gsplat's source is
  not vendored in the reference, is CUDA-only and cannot be installed here, and the reference has
  no tests or golden images: **parity unpinned** for the rasterizer.  It is anchored instead by
  analytic known-answer tests, fp64 `gradcheck`, and agreement with the independent scalar C
  restatement in `oracle/raster_ref.c`.
* `oracle.camera`'s pypose restatement (`pypose==0.6.8`, `requirements.txt:354`; call sites
  `move_model.py:145-146`, `spline_utils.py:386-408`) and the roma restatement in `oracle.deform`
  (`roma==1.5.0`, `requirements.txt:385`; call sites `scene_model.py:94-101`) are likewise
  **parity unpinned** (libraries absent); they are checked through group identities only.
"""
