"""The library's RCCL transport between two real ranks on two GPUs (skipped on a box with one): the
first multi-GPU machine that runs the suite tests gfship_domain_comm_init, the unique-id broadcast, the
per-rank device choice, comm_exchange with two distinct peers over xGMI and the all-gather reductions
-- instead of bench.py --gpus 2 being their first user.  2 x 1 x 1 boxes of 32^3 holding their parts of
a field with one period over the lattice, against oracle boxes of the same lattice, bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gfship
from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_two_ranks_over_real_rccl_equal_oracle_boxes(tmp_path):
    if gfship.lib().gfship_device_count() < 2:
        pytest.skip("needs two GPUs")
    from test_multibox_cpu import run_lattice_flow_threads
    level, nsteps = 5, 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the ranks are fresh children: nothing in them has touched a GPU when they start
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "two_rank_worker.py"), str(tmp_path), str(level), str(nsteps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ora = run_lattice_flow_threads(2, level, nsteps, 0)
    for rank in range(2):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        o = ora[rank]
        for name in ("u", "v", "w", "p", "pmac"):
            assert np.array_equal(d[name], o[name]), (rank, name)
        assert float(d["dt"]) == o["dt"] and float(d["t"]) == o["t"]
        assert tuple(int(x) for x in d["niter"]) == tuple(o["niter"])
        assert float(d["res"]) == o["res"]
        assert int(d["msgs"]) > 50
