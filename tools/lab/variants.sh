for v in "" _d24 _d32 _d40; do
  export GFSHIP_LIB=$GRAFT_REPO_ROOT/gerris-fft-particles_amd/lib/libgfship$v.so
  echo "== variant $v"
  timeout -k 10 300 python -m pytest tests/test_gpu_poisson.py -x -q -m gpu -k "fused" 2>&1 | tail -1
  GFSHIP_SKEW_STATS=1 timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "tile \( 0, 0\)|nrelax 4 \(fused" | tail -2 | cut -c1-100
  for l in 7 6 5; do timeout -k 10 100 python tools/relax_only.py $l 2>&1 | grep -E "nrelax 4 \(fused" | tail -1; done
done
