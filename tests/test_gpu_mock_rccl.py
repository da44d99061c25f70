"""N > 1 ranks of the library's OWN RCCL transport (csrc/transport.hip) on the one GPU of the test box:
the ranks are threads of one process and the RCCL the library opens is the in-process stand-in of
tests/mock_rccl (GFSHIP_RCCL_LIBRARY), which keeps NCCL's matching rule -- the k-th send from A to B
meets the k-th receive from A on B, counts must agree -- and reports mismatches and waits that can
never be served.  With a self-communicator (tests/test_gpu_multibox.py) every peer is rank 0; here the
peers are distinct: a wrong peer[], a wrong rank_of, a left / right mix-up or a posting-order mismatch
between two peers gives wrong bits, a count mismatch or a time-out.  Device boxes against oracle boxes
of the same lattice, bit for bit.  (One process per case: the library opens its RCCL once.)
The real multi-GPU run is tests/test_gpu_two_ranks.py (skipped with fewer than two devices)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mock_rccl_worker.py")] +
                       [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("MOCKRCCL ")][-1]
    return json.loads(line[len("MOCKRCCL "):])


@pytest.mark.parametrize("nboxes,level,overlap,fast", [(2, 5, 0, True), (4, 4, 0, True), (8, 4, 0, True),
                                                       (8, 4, 0, False), (2, 4, 1, True), (8, 4, 1, True)])
def test_lattice_flow_over_the_library_transport_with_distinct_peers(nboxes, level, overlap, fast):
    out = _run("flow", nboxes, level, 2, overlap, "fast" if fast else "nofast")
    assert out["bad"] == [], out
    assert out["differ"]                       # the boxes hold different parts of the field
    assert out["mismatches"] == 0 and out["timeouts"] == 0
    assert out["messages"] > 100 and min(out["sent"]) > 50
    if fast and not overlap:
        # the coarse end of every V-cycle after one all-gather, the tiled Godunov kernels with one
        # message of face states per side
        assert min(out["lattice_cycles"]) >= 4
        if level >= 5:       # the tiles of the Godunov kernels are 32 cells long
            assert min(out["fused_mpi"]) >= 4
    else:
        assert max(out["lattice_cycles"]) == 0


def test_particle_migration_over_the_library_transport_with_distinct_peers():
    out = _run("particles")
    assert out["bad"] == [], out
    assert out["moved"] > 100 and out["foreign"] > 50
    assert out["mismatches"] == 0 and out["timeouts"] == 0
