#!/bin/bash
# round-5 batch F: 8 / 16 instances per lane in count / emit on big tile grids (D4GS_CHUNK16=0 = the 4-per-lane chunks), saturation statistics
mkdir -p gpurun_out
{
for c in "--config cfg3" "--config cfg5 --steps 10" "--config refdefault720 --steps 10" "--config cfg5 --steps 10 --spatial-order"; do
  for v in 0 1; do
    D4GS_CHUNK16=$v python bench.py $c --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('[$c] CHUNK16=$v', '%.3f ms' % d['ms_per_step'], d.get('n_isect_per_step'), d['config'].get('lazy_sort'), {n: round(1e3*t) for n,t in list(k.items())[:10]})"
  done
done
} 2>&1 | tee gpurun_out/r5f_ab.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5f_saturation.txt
# how many pixels of the benched cfg2 scene saturate (T <= 1e-4 stops the composite) and where in their tile's list: the data behind the
# "depth-split forward" question (a split needs a re-walk of the segment a pixel stops in)
import torch, bench
from deblur4dgs_amd.exposure import render_exposure
dev = torch.device("cuda:0")
sc, d, leaves, wimg, wacc = bench.make_inputs("cfg2", dev, channels=3)
with torch.no_grad():
    res = render_exposure(leaves["means"], leaves["quats"], leaves["scales"], leaves["opacities"], leaves["colors"], 3, leaves["motion_coefs"],
                          leaves["rots"], leaves["transls"], leaves["times"], leaves["RTs"], leaves["viewmat"], d["K"], 512, 288,
                          background=torch.ones(3, device=dev), return_depth=True, blend=False)
st = res["state"]
T = st.raster["final_T"]; last = st.raster["last_ids"].long()
S, H, W = T.shape
tw, th = st.cfg.tiles
offs = st.proj_out["tile_offsets"].long()
n_list = (offs[1:] - offs[:-1]).view(S, th, tw)
start = offs[:-1].view(S, th, tw)
ys, xs = torch.meshgrid(torch.arange(H, device=dev) // 16, torch.arange(W, device=dev) // 16, indexing="ij")
pl = n_list[:, ys, xs]; ps = start[:, ys, xs]
pos = (last - ps + 1).clamp(min=0).float() / pl.clamp(min=1).float()   # fraction of the tile's list a pixel consumed
sat = T <= 2e-4
print("pixels saturated (final T <= 2e-4): %.3f" % sat.float().mean().item())
print("list fraction consumed, all pixels: mean %.3f; quartiles %s" % (pos.mean().item(), [round(x, 3) for x in torch.quantile(pos.flatten()[::7], torch.tensor([.25, .5, .75], device=dev)).tolist()]))
seg = (pos * 4).clamp(max=3.999).long()   # which quarter of its list a pixel stops in
blk = seg.view(S, H // 4, 4, W // 4, 4).permute(0, 1, 3, 2, 4).reshape(S, H // 4, W // 4, 16)
distinct = torch.stack([(blk == k).any(-1) for k in range(4)], -1).sum(-1).float()
print("distinct list quarters the 16 pixels of a 4x4 block stop in: mean %.2f" % distinct.mean().item())
PY
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r5f_pytest_gpu.txt
