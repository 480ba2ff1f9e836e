"""Synthetic inputs for benchmarks / smoke runs: no dataset or checkpoint is
available offline, so benchmarks run on a car-sized ellipsoid cloud, look-at poses on the radius-1.3
camera sphere of data/srncars_test_poses.npy and the SRN intrinsics."""
import math

import torch


def ellipsoid_cloud(n_points: int = 512, feats_dim: int = 32, batch: int = 1, seed: int = 0):
    """points on an ellipsoid with semi-axes (0.45, 0.20, 0.15); feats ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(batch, n_points, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    return u * torch.tensor([0.45, 0.20, 0.15]), torch.randn(batch, n_points, feats_dim, generator=g)


def look_at_pose(azim_deg: float, elev_deg: float, radius: float = 1.3) -> torch.Tensor:
    """world2cam matrix of a camera on the sphere looking at the origin (+z forward)."""
    az, el = math.radians(azim_deg), math.radians(elev_deg)
    c = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el)])
    fwd = -c / c.norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    rot = torch.stack((right, down, fwd))
    m = torch.eye(4)
    m[:3, :3] = rot
    m[:3, 3] = -rot @ c
    return m


def srn_intrinsics(res: int = 128) -> torch.Tensor:
    """fx = fy = 131.25, cx = cy = 64 at 128x128 (K[2,2] = 0 as in data/srncars_test_intrinsics.npy)."""
    s = res / 128.0
    return torch.tensor([[131.25 * s, 0.0, 64.0 * s], [0.0, 131.25 * s, 64.0 * s], [0.0, 0.0, 0.0]])
