timeout -k 10 300 python -m pytest tests/test_gpu_poisson.py -x -q -m gpu -k "fused" 2>&1 | tail -1
for v in "" _s0 _s1 _s8; do
  export GFSHIP_LIB=$GRAFT_REPO_ROOT/gerris-fft-particles_amd/lib/libgfship$v.so
  echo "== variant $v"
  timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "nrelax 4 \(fused" | tail -1
  timeout -k 10 100 python tools/relax_only.py 6 2>&1 | grep -E "nrelax 4 \(fused" | tail -1
done
