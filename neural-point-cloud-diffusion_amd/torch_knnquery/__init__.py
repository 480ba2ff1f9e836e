"""Drop-in for the `torch_knnquery` package the reference imports
(npcd/models/pointnerf/pointnerf.py:5,20; fields/aggregators/aggregator.py:7,63):

    VoxelGrid(voxel_size, voxel_scale, kernel_size, max_points_per_voxel, max_occ_voxels_per_example, ranges)
        .set_pointset(points [B,N,3] f32, counts [B] int32)
        .query(x [B,R,S,3] f32, k, r, max_shading_pts) -> (sample_idx [Rv,M,k], sample_loc [Rv,M,3], ray_mask [B,R])
        .vsize_tup

backed by the gfx950 HIP kernels in libnpcd_hip.so (deterministic semantics: DESIGN.md, "VoxelGrid spec")."""
from npcd.hip.render import HipVoxelGrid as VoxelGrid

__all__ = ["VoxelGrid"]
