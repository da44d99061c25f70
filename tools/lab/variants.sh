timeout -k 10 300 python -m pytest tests/test_gpu_poisson.py -x -q -m gpu -k "fused or skew" 2>&1 | tail -2
GFSHIP_SKEW_STATS=1 timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "tile \( 0, 0\)|tile \(15,15\)|nrelax 4|per sweep" | tail -5
