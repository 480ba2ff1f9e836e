"""Dev probe: where do the training-path gradients of the HIP build and the oracle differ (grid mode)?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from oracle import renderer as orr, train_render as otr
import test_gpu_train_render as tt
from npcd.models.pointnerf import train_path as tp
from npcd.hip import render as hr
B, Tn, N, F_, res = 2, 2, 512, 32, 32
coords, feats = orr.synthetic_cloud(N, F_, B, seed=4); coords[1] = coords[1].flip(-1) * 1.2
extr = torch.stack([orr.look_at_pose(20 + 80 * i, 15 - 10 * i) for i in range(Tn)])[None].expand(B, -1, -1, -1).contiguous()
K = orr.srn_intrinsics().clone(); K[0, 0] = K[1, 1] = 131.25 * res / 128; K[0, 2] = K[1, 2] = res / 2
intr = K[None, None].expand(B, Tn, 3, 3).contiguous()
p = orr.init_field_params(F_, seed=2)
net = tt._model(F_, N, p); pn = net.pointnerf.train(); ren, agg = pn.renderer, pn.field.aggregator; ren.depth_resolution = 64
g = torch.Generator().manual_seed(3)
rng = {"ray_perm": torch.randperm(res * res, generator=g), "jitter": torch.rand(B * Tn, ren.ray_subsamples, 64, 1, generator=g)}
# positions on GPU (product formulas) vs CPU (oracle formulas)
o, d, _, _ = hr.ray_gen(extr.flatten(0, 1).cuda(), intr.flatten(0, 1).cuda(), res, 1.0)
o, d = o.view(B, Tn, -1, 3), d.view(B, Tn, -1, 3)
ids = rng["ray_perm"][:ren.ray_subsamples].cuda()
o, d = o[:, :, ids], d[:, :, ids]
s, e = tp.box_limits(o, d, 1.0)
dep = tp.jittered_depths(s, e, 64, rng["jitter"].cuda().reshape(B, Tn, -1, 64))
xg = (o[..., None, :] + dep[..., None] * d[..., None, :]).cpu()
oc, dc = orr.camera_rays(extr.flatten(0, 1), intr.flatten(0, 1), res)
oc, dc = oc.reshape(B, Tn, -1, 3), dc.reshape(B, Tn, -1, 3)
oc, dc, _ = otr.subsample_rays(oc, dc, rng["ray_perm"], ren.ray_subsamples)
Rs = oc.shape[2]
sc, ec = orr.ray_box_limits(oc.reshape(B, Tn * Rs, 3), dc.reshape(B, Tn * Rs, 3), 1.0)
depc = otr.jittered_depths(sc.reshape(B, Tn, Rs, 1), ec.reshape(B, Tn, Rs, 1), 64, rng["jitter"].reshape(B, Tn, Rs, 64))
xc = oc[..., None, :] + depc[..., None] * dc[..., None, :]
print("positions: max abs diff", float((xg - xc).abs().max()), "bitwise equal frac", float((xg == xc).float().mean()))
print("limits diff", float((s.cpu() - sc.reshape(B, Tn, Rs, 1)).abs().max()), float((e.cpu() - ec.reshape(B, Tn, Rs, 1)).abs().max()))
# neighbour lists for the GPU positions from both implementations
pn.voxel_grid.set_pointset(coords.cuda(), torch.full((B,), N, dtype=torch.int, device="cuda"))
idx_g, loc_g, _, _ = agg.voxel_grid.query_dense(agg.k, agg.r, agg.max_shading_pts, x=xg.reshape(B, Tn * Rs, 64, 3).cuda().contiguous(), mode=0, points=coords.cuda())
from oracle.voxel_grid import VoxelGridOracle
grid = VoxelGridOracle(**orr.DEFAULT_GRID); grid.set_pointset(coords.numpy(), np.full((B,), N, dtype=np.int32))
idx_c, loc_c, _, _ = grid.query_dense(xc.reshape(B, Tn * Rs, 64, 3).numpy(), agg.k, agg.r, agg.max_shading_pts)
idx_c2, _, _, _ = grid.query_dense(xg.reshape(B, Tn * Rs, 64, 3).numpy(), agg.k, agg.r, agg.max_shading_pts)
ig = idx_g.cpu().numpy().reshape(idx_c.shape)
print("lists differ (gpu pos on gpu vs cpu pos on oracle):", int((ig != idx_c).any(-1).sum()), "slots of", ig.shape[0] * ig.shape[1] * ig.shape[2])
print("lists differ (same gpu positions, gpu vs oracle):", int((ig != idx_c2).any(-1).sum()))
# ---- gradients
out0 = tt._oracle_slots(p, coords, extr, intr, res, 64, agg, ren, rng)
rng["valid_perm"] = torch.randperm(out0, generator=g)
fd = feats.cuda().requires_grad_(True)
out = ren(coords.cuda(), fd, extr.cuda(), intr.cuda(), res, True, rng=rng)
po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
fo = feats.clone().requires_grad_(True)
ref = otr.render_train(po, coords, fo, extr, intr, res, 64, agg.max_shading_pts, agg.k, agg.r, "grid", ren.ray_subsamples,
                       agg.ray_subsamples, rng["ray_perm"], rng["jitter"], rng["valid_perm"], stable_regroup=True)
gch = torch.randn(ref["channels"].shape, generator=g)
for key in ("mask", "channels", "depth"):
    print(key, "fwd max err", float((out[key].cpu() - ref[key]).abs().max()))
(out["channels"] * gch.cuda()).sum().backward()
(ref["channels"] * gch).sum().backward()
err = (fd.grad.cpu() - fo.grad).abs()
print("grad max", float(fo.grad.abs().max()), "err max", float(err.max()), "per object", err.flatten(1).max(1).values.tolist())
rows = err.max(-1).values
print("rows with err > 1e-7:", int((rows > 1e-7).sum()), "of", rows.numel(), "rows with nonzero grad", int((fo.grad.abs().max(-1).values > 0).sum()))
rel = (fd.grad.cpu() - fo.grad).norm() / fo.grad.norm()
print("rel L2", float(rel))
# double precision replay of the oracle
po64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
