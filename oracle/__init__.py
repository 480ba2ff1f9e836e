"""CPU oracle for the NPCD hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32, CPU) / numpy restatement of the reference
algorithms on the hot path named by BASELINE.json `north_star`:

  * the transformer denoiser + DDPM training loss  (oracle.denoiser, oracle.diffusion)
  * the PointNeRF renderer                          (oracle.renderer, oracle.voxel_grid)

Every function cites the reference file:line it restates (paths relative to the
upstream repo lmb-freiburg/neural-point-cloud-diffusion @ 2024_10_08).

Parity status
-------------
* Denoiser, attention, diffusion loss, normalisers, ray generation, ray limits,
  depth sampling, brute-force neighbour query, aggregator MLP, density / colour
  heads, depth-from-points and ray-march are PINNED: `tests/golden/*.npz` were
  produced by importing the reference itself (tests/golden/make_golden.py, run in
  the build container) and `tests/test_oracle_golden.py` checks this oracle
  against them.
* `torch_knnquery.VoxelGrid` (third-party CUDA extension, unpinned git HEAD, source
  not available) is "PARITY UNPINNED": oracle.voxel_grid implements the
  deterministic specification written in DESIGN.md (derived from the reference's
  call sites); nothing from upstream pins it.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package -- and only as the checker / the timed CPU baseline, never
as part of the product path.  The product (`neural-point-cloud-diffusion_amd/`) never
imports it and fails loudly when the HIP library is missing.
"""
