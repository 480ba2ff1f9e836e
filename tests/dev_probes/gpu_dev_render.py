"""Developer probe (not a pytest file): renderer kernels vs the oracle, with error printouts."""
import sys, os, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import numpy as np, torch
from oracle import renderer as orr, voxel_grid as ovg
from npcd.hip import render as hr
from npcd.models.pointnerf import PointNeRF

torch.manual_seed(0)
res = int(os.environ.get("RES", 32)); N = 512; F_ = 32
coords, feats = orr.synthetic_cloud(N, F_, 1, seed=0)
extr = torch.stack([orr.look_at_pose(30, 20), orr.look_at_pose(200, -10)])[None]   # [1,2,4,4]
K = orr.srn_intrinsics().clone(); K[0, 0] = K[1, 1] = 131.25 * res / 128; K[0, 2] = K[1, 2] = res / 2
intr = K[None, None].expand(1, 2, 3, 3).contiguous()

# 1. rays
o_ref, d_ref = orr.camera_rays(extr[0], intr[0], res)
s_ref, e_ref = orr.ray_box_limits(o_ref, d_ref)
o, d, t0, t1 = hr.ray_gen(extr[0].cuda(), intr[0].cuda(), res)
print("rays  o %.2e d %.2e t0 %.2e t1 %.2e" % ((o.cpu()-o_ref).abs().max(), (d.cpu()-d_ref).abs().max(),
      (t0.cpu()-s_ref[...,0]).abs().max(), (t1.cpu()-e_ref[...,0]).abs().max()))

# 2. grid query, bit exact, ray form (feed ORACLE rays to both)
S, M, k = 128, 50, 8
V, R = o_ref.shape[:2]
ro, rd = o_ref.reshape(1, V*R, 3), d_ref.reshape(1, V*R, 3)
rs, re = s_ref.reshape(1, V*R), e_ref.reshape(1, V*R)
dep = orr.depth_samples(rs[..., None], re[..., None], S)
x = (ro[:, :, None, :] + dep[..., None] * rd[:, :, None, :]).numpy()
g = ovg.VoxelGridOracle(); g.set_pointset(coords.numpy(), np.array([N], dtype=np.int32))
t = time.time(); ridx, rloc, rnsel, rss = g.query_dense(x, k, 2.0, M); print("oracle grid query %.1fs" % (time.time()-t))
hg = hr.HipVoxelGrid(**orr.DEFAULT_GRID); hg.set_pointset(coords.cuda(), torch.full((1,), N, dtype=torch.int32, device="cuda"))
idx, loc, ss, nsel = hg.query_dense(k, 2.0, M, rays=(ro.cuda(), rd.cuda(), rs.cuda(), re.cuda()), S=S)
print("grid(ray form): nsel eq", bool((nsel.cpu().numpy() == rnsel).all()), " slot_sample eq", bool((ss.cpu().numpy() == rss).all()),
      " idx eq", bool((idx.cpu().numpy() == ridx).all()), " loc eq", bool((loc.cpu().numpy() == rloc).all()),
      " mismatches", int((idx.cpu().numpy() != ridx).sum()), "of", ridx.size, " P", int((ridx[...,0]>=0).sum()), "Q", int((ridx>=0).sum()))
idx2, loc2, ss2, nsel2 = hg.query_dense(k, 2.0, M, x=torch.from_numpy(x).cuda())
print("grid(x form):   idx eq", bool((idx2.cpu().numpy() == ridx).all()), " nsel eq", bool((nsel2.cpu().numpy() == rnsel).all()))
bidx, bloc, bn = ovg.brute_force_query(x, coords.numpy(), k, 0.08, M)
idx3, loc3, ss3, nsel3 = hg.query_dense(k, 0.08, M, x=torch.from_numpy(x).cuda(), mode=1)
print("brute:          idx eq", bool((idx3.cpu().numpy() == bidx).all()), " nvalid eq", bool((nsel3.cpu().numpy() == bn).all()),
      " loc eq", bool((loc3.cpu().numpy() == bloc).all()))

# 3. shading on the oracle's neighbour lists
p = orr.init_field_params(F_, seed=0)
valid = torch.from_numpy(ridx[..., 0] >= 0).reshape(-1, M)
nb = torch.from_numpy(ridx.astype(np.int64)).reshape(-1, M, k)[valid]
pts = torch.from_numpy(rloc).reshape(-1, M, 3)[valid]
t = time.time(); sig_ref, rgb_ref, _ = orr.shade_points(p, nb, pts, coords, feats); print("oracle shade %.1fs P=%d" % (time.time()-t, nb.shape[0]))
model = PointNeRF(1, F_, N, False)
model.field.load_state_dict(p); model = model.cuda().eval()
sig, rgb = model.field.shade(nb.int().cuda(), pts.cuda(), coords.cuda(), feats.cuda())
print("shade: sigma max-abs %.2e (ref max %.2f)  rgb max-abs %.2e" % ((sig.cpu()-sig_ref[:,0]).abs().max(), sig_ref.max(), (rgb.cpu()-rgb_ref).abs().max()))

# 4. end to end
with torch.no_grad():
    out = model.render(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
ref = orr.render(p, coords, feats, extr, intr, res=res, S=S, M=M, k=k, r=2.0, mode="grid")
for kk in ("mask", "depth", "channels"):
    print("render", kk, "max-abs %.2e" % (out[kk].cpu()-ref[kk]).abs().max())
print("PSNR(hip, oracle) = %.1f dB ; P=%d Q=%d" % (orr.psnr(out["channels"].cpu(), ref["channels"]), out["num_shading_points"], out["num_pairs"]))
with torch.no_grad():
    outb = model.renderer(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res, False, knn_mode=1)
refb = orr.render(p, coords, feats, extr, intr, res=res, S=S, M=M, k=k, r=0.08, mode="brute")
print("brute PSNR = %.1f dB, max-abs %.2e" % (orr.psnr(outb["channels"].cpu(), refb["channels"]), (outb["channels"].cpu()-refb["channels"]).abs().max()))

# 5. timing at 128^2, one view
res = 128
K = orr.srn_intrinsics(); intr1 = K[None, None].cuda(); extr1 = extr[:, :1].cuda()
c, f = coords.cuda(), feats.cuda()
with torch.no_grad():
    for _ in range(3): out = model.render(c, f, extr1, intr1, res)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): out = model.render(c, f, extr1, intr1, res)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
print("128^2 view: %.2f ms  %.2f Mrays/s  P=%d Q=%d" % (dt*1e3, 16384/dt/1e6, out["num_shading_points"], out["num_pairs"]))
