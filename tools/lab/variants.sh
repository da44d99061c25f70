timeout -k 10 600 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_timestep.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "ms per sweep|nrelax 4" | tail -3
GFSHIP_SKEW_OLD=1 timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "ms per sweep" | tail -1
