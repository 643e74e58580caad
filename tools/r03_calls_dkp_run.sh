for v in "" dkp1 dkp2 dkp4 dkp8 dkp15; do
  if [ -n "$v" ]; then export SDPA_HIP_LIB=$PWD/mpi-parallelized-scaled-dot-product-attention-with-avx-512-optimization_amd/lib/variants/libsdpa_hip_$v.so; else unset SDPA_HIP_LIB; fi
  echo "variant ${v:-shipped}"
  SDPA_DKSPLIT_PIPE=1 python tools/gpu_f32_dims.py 512 2>&1 | grep tflops
done
