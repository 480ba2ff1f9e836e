for t in default l128d1 l128d2 l128d3; do
  if [ $t = default ]; then unset NPCD_HIP_LIB; else export NPCD_HIP_LIB=$GRAFT_REPO_ROOT/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$t.so; fi
  echo "== $t"; python3 tools/probes/gpu_dev_lin128.py 4104 | grep "mlp.c_proj\|attn.c_proj"
done
