#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a ONE-GPU box: N ranks share the card, the halos are staged
# through the host and sent over gloo (RCCL refuses several ranks on one device).  The numbers
# mean nothing; what is checked is that the multi-box step runs and the JSON line comes out.
N=${1:-2}; LEVEL=${2:-7}
GFSHIP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 2 --warmup 1 --level $LEVEL
