#!/usr/bin/env python3
"""bench.py -- Mcell-steps/s of the projection + advection time step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the simulation_run loop (src/simulation.c:479-548: predictor, MAC
projection, velocity advection, approximate projection, CFL) on a 3-D 256^3 triply periodic
Taylor-Green box with default parameters, one box per GPU.  Inputs are
resident in HBM before the timed region.  Prints ONE JSON line on rank 0, with
  roofline      the exact-order relax loop (4 pipelined sweeps) of the 256^3 level (dominant
                kernel), timed with HIP events on the stream the kernels run on: 24 B per cell
                and sweep algorithmic traffic
  cpu_baseline  the repo's CPU oracle (a port of the reference algorithm; the reference itself
                cannot be built here) on a bounded sample, 1 core.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gerris-fft-particles_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
RELAX_BYTES_PER_CELL = 24.0    # read u, read rhs, write u
PMC_SUMMARY = "r03_pmc_relax_loop_256.json"


def taylor_green(n):
    c = -0.5 + (np.arange(1, n + 1) - 0.5) / n
    x, y, z = c[None, None, :], c[None, :, None], c[:, None, None]
    u = np.sin(2. * np.pi * x) * np.cos(2. * np.pi * y) * np.cos(2. * np.pi * z)
    v = -np.cos(2. * np.pi * x) * np.sin(2. * np.pi * y) * np.cos(2. * np.pi * z)
    w = np.zeros((n, n, n))
    return u, v, w


def with_ghosts(a):
    n = a.shape[0]
    b = np.zeros((n + 2,) * 3)
    b[1:-1, 1:-1, 1:-1] = a
    return b


def cpu_baseline(level=7, steps=3):
    """The oracle on a bounded sample of the same workload (2^level cells per side, same
    parameters), 1 core."""
    from oracle import oracle as O
    n = 1 << level
    s = O.Sim(3, level, [O.SIDE_PERIODIC] * 6)
    for c, a in enumerate(taylor_green(n)):
        s.u[c].interior()[...] = a
    s.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
    dt = time.perf_counter() - t0
    out = {"value": n ** 3 * steps / dt / 1e6, "unit": "Mcell-steps/s", "cores": 1,
           "kind": "port",
           "sample": "%d^3 Taylor-Green, %d steps, CPU oracle (reference algorithm, "
                     "reference traversal order)" % (n, steps)}
    out["relax_sweep"] = cpu_relax_layouts(level)
    return out


def cpu_relax_layouts(level=7, sweeps=4):
    """Where the CPU time goes: the same relax sweeps on the oracle's flat arrays and on
    a reference-like tree of FttCell / FttOct records with ftt_cell_neighbor lookups and one callback
    per cell (oracle/go_aos.c; both give the same bits, tests/test_oracle_aos_baseline.py)."""
    import ctypes as C
    from oracle import oracle as O
    L = O.lib()
    vp, pd, i = C.c_void_p, C.POINTER(C.c_double), C.c_int
    L.go_aos_new.restype, L.go_aos_new.argtypes = vp, [i]
    L.go_aos_destroy.restype, L.go_aos_destroy.argtypes = None, [vp]
    L.go_aos_load.restype, L.go_aos_load.argtypes = None, [vp, vp, pd, pd, pd]
    L.go_aos_relax.restype, L.go_aos_relax.argtypes = None, [vp, i]
    L.go_aos_bytes_per_cell.restype = C.c_size_t
    n = 1 << level
    dom = O.Domain(3, level, [O.SIDE_PERIODIC] * 6)
    L.go_poisson_coefficients(dom.ptr)
    u, rhs, dia = dom.field(), dom.field(), dom.field()
    rng = np.random.default_rng(0)
    u.interior()[...] = rng.standard_normal(u.interior().shape)
    rhs.interior()[...] = rng.standard_normal(u.interior().shape)
    lev = lambda f: C.cast(L.go_field_level(f.ptr, level), pd)   # noqa: E731
    tree = L.go_aos_new(level)
    L.go_aos_load(tree, dom.ptr, lev(u), lev(rhs), lev(dia))
    t0 = time.perf_counter()
    L.go_aos_relax(tree, sweeps)
    t_aos = time.perf_counter() - t0
    L.go_aos_destroy(tree)
    t0 = time.perf_counter()
    for _ in range(sweeps):
        L.go_homogeneous_bc(u.ptr, u.ptr, level)
        L.go_relax(dom.ptr, 3, level, 1., u.ptr, rhs.ptr, dia.ptr)
    t_soa = time.perf_counter() - t0
    return {"unit": "Mcell-sweeps/s", "cores": 1, "sample": "%d sweeps at %d^3" % (sweeps, n),
            "flat_arrays": n ** 3 * sweeps / t_soa / 1e6,
            "pointer_tree_aos": n ** 3 * sweeps / t_aos / 1e6,
            "aos_bytes_per_cell": int(L.go_aos_bytes_per_cell())}


def measure_vcycle(dom, u, rhs, dia, n):
    """one whole gfs_poisson_cycle of the box as the projections run it (every level: restrictions,
    relax loops with their copies, prolongations, the correction, the new residual), wall clock over
    10 cycles enqueued back to back"""
    res = dom.variable()
    dia.fill(0.)
    dom.bc(u)
    dom.residual(u, rhs, dia, res)
    par = dom.params()
    par.depth = dom.depth
    dom.poisson_cycle(par, u, rhs, dia, res)
    dom.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        dom.poisson_cycle(par, u, rhs, dia, res)
    dom.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    # algorithmic bytes of a cycle: per level 4 sweeps x 24 B + residual 32 B + restriction / prolongation
    # / correction 8 + 8 + 24 B per cell; the levels below add 1/7 of the leaf level
    cells = n ** 3 * 8. / 7.
    return {"ms": ms, "algorithmic_GBps": cells * (4 * 24 + 32 + 40) / (ms * 1e-3) / 1e9,
            "note": "gfship_poisson_cycle, all levels, 10 back to back"}


def measure_roofline(dom, args, n):
    u, rhs, dia = dom.variable(), dom.variable(), dom.variable()
    rng = np.random.default_rng(0)
    u.upload(rng.standard_normal((n + 2,) * 3))
    rhs.upload(rng.standard_normal((n + 2,) * 3))
    dom.poisson_coefficients()
    nrelax = 4
    # medians of 20 single measurements (HIP events on the library's stream around the kernel)
    ms_sweep = float(np.median([dom.time_relax(u, rhs, dia, reps=1) for _ in range(20)]))
    roofline = None
    if args.mode == "exact":
        runs = [dom.time_relax_loop_inclusive(u, rhs, dia, nrelax=nrelax, reps=1) for _ in range(20)]
        ms_loop, fused = float(np.median([r[0] for r in runs])), runs[0][1]
        ms_incl = float(np.median([r[2] for r in runs]))
        bytes_loop = RELAX_BYTES_PER_CELL * n ** 3 * nrelax
        achieved = bytes_loop / (ms_loop * 1e-3) / 1e9
        # HBM bytes per launch from rocprofv3 PMC passes of the same kernel at the same size
        # (FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes): counters cannot
        # be read from inside this process, so the committed summary is reported
        traffic = None
        pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
        if args.level == 8 and fused and os.path.exists(pmc):
            with open(pmc) as f:
                kernels = json.load(f)["kernels"]
                traffic = next(iter(kernels.values()))["hbm_bytes_per_launch_guide_corrected"]
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": None if traffic is None else
                    "profiles/%s: rocprofv3 --pmc passes of this kernel at this size, committed; "
                    "not collected in this run" % PMC_SUMMARY,
                    "algorithmic_bytes_per_launch": bytes_loop,
                    "kernel": "relax loop (%d sweeps%s), level %d (%d^3), mode exact"
                              % (nrelax, ", one pipelined launch" if fused else
                                 ", one launch per sweep", args.level, n),
                    "ms_per_launch": ms_loop if fused else ms_loop / nrelax,
                    # the same loop with everything poisson_cycle runs on this level between the loop
                    # of the level below and the corrected solution: prolongation straight into the
                    # layout of the loop + BC kernel + the sweeps + ghost planes + the way out of the
                    # layout with the correction u += dp in it (the granules are armed on a side stream
                    # beside the coarser levels of the cycle, the rhs arrives with the restriction)
                    "inclusive": {"ms_per_loop": ms_incl,
                                  "what": "get_from_above into the layout + BC + %d sweeps + ghost "
                                          "planes + correct out of the layout" % nrelax,
                                  "achieved": bytes_loop / (ms_incl * 1e-3) / 1e9,
                                  "frac": bytes_loop / (ms_incl * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "ms_per_sweep_in_loop": ms_loop / nrelax,
                    "ms_single_sweep_launch": ms_sweep,
                    "vcycle": measure_vcycle(dom, u, rhs, dia, n)}
    else:
        achieved = RELAX_BYTES_PER_CELL * n ** 3 / (ms_sweep * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "algorithmic_bytes_per_launch": RELAX_BYTES_PER_CELL * n ** 3,
                    "kernel": "relax sweep, level %d (%d^3), mode %s" % (args.level, n, args.mode),
                    "ms_per_launch": ms_sweep}

    return roofline


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--level", type=int, default=8, help="Refine level (8 = 256^3)")
    ap.add_argument("--mode", default="exact", choices=["exact", "redblack"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--particles", type=int, default=2000000,
                    help="tracers of the config D line (0 = skip)")
    ap.add_argument("--self-mpi", action="store_true",
                    help="lab: one rank whose six sides are GfsBoundaryMpi sides facing the box itself "
                         "(RCCL send / recv to self): the code path of a box in an N > 1 run, on one GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand as `python bench.py --gpus N`: start the N ranks as a fresh child under
        # torch.distributed.run BEFORE anything here touches the GPU (this process never does), and
        # relay its output and exit code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        r = subprocess.run(cmd, env=env)
        sys.exit(r.returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE = %d\n" % (args.gpus, world))
        sys.exit(2)
    dist = None
    backend = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the launcher's channel only (rendezvous, the 128 bytes of the RCCL
        # unique id, the barriers around the timed region): a CPU process group.  The halos and
        # reductions of the time step go through the library's own RCCL communicator.
        # GFSHIP_DIST_BACKEND=gloo-staged: rehearsal of the N > 1 path with several ranks on one GPU
        # (RCCL refuses that): halos staged through the host by the hook transport
        # (gfship/distributed.py); never used for reported numbers
        backend = os.environ.get("GFSHIP_DIST_BACKEND", "rccl")
        dist.init_process_group("gloo")
        if backend != "rccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())

    import gfship
    n = 1 << args.level
    hooks = None
    rccl_world = 0
    if world > 1:
        # one 256^3 GfsBox per GPU on a periodic lattice of boxes (2x1x1, 2x2x1, 2x2x2): the sides
        # with a neighbour box are GfsBoundaryMpi sides
        from gfship import distributed as D
        grid = D.BoxGrid(world, 3)
        dom = gfship.Domain(3, args.level, grid.sides(rank), device=local_rank)
        if backend == "rccl":
            import torch
            # two phases, so that no rank can be left alone inside the collective ncclCommInitRank:
            # (1) everything that can fail locally -- the library loaded, RCCL opened, the device
            # selected (all done by the Domain above and comm_unique_id), the unique id received --
            # is agreed on over gloo first; (2) only then every rank enters comm_init.  A failure
            # inside (2) ends the run with a non-zero exit code on that rank (the launcher then ends
            # the others): no asymmetric fallback is attempted
            ok = torch.ones(1, dtype=torch.int32)
            uid = torch.zeros(gfship.UNIQUE_ID_BYTES, dtype=torch.uint8)
            try:
                if rank == 0:
                    uid = torch.frombuffer(bytearray(gfship.comm_unique_id()), dtype=torch.uint8).clone()
                else:
                    gfship.comm_available()
            except Exception as e:      # reported, never silent: see "parallelism" in the line
                sys.stderr.write("bench.py: rank %d: RCCL not usable: %s\n" % (rank, e))
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                dist.broadcast(uid, 0)
                try:
                    dom.comm_init(bytes(uid.numpy().tobytes()), rank, world, grid.b)
                    rccl_world = dom.comm_size()
                except Exception as e:
                    sys.stderr.write("bench.py: rank %d: ncclCommInitRank failed: %s\n" % (rank, e))
                    sys.stderr.flush()
                    os._exit(3)
            else:
                # every rank takes the same decision, before anybody entered a collective of RCCL:
                # host-staged hooks, and the line says so
                backend = "gloo-staged (RCCL not usable on some rank)"
        if backend != "rccl":
            import torch
            torch.cuda.set_device(local_rank)
            hooks = D.DeviceHooks(dom, D.Transport(grid, rank, torch.device("cuda", local_rank)))
    elif args.self_mpi:
        dom = gfship.Domain(3, args.level, [gfship.SIDE_EXTERNAL] * 6, device=local_rank)
        dom.comm_init(gfship.comm_unique_id(), 0, 1, (1, 1, 1))
        rccl_world = dom.comm_size()
    else:
        dom = gfship.Domain(3, args.level, [gfship.SIDE_PERIODIC] * 6, device=local_rank)
    if args.mode == "redblack":
        dom.set_relax_mode(gfship.RELAX_REDBLACK)
    sim = gfship.Simulation(dom)
    for c, a in enumerate(taylor_green(n)):
        sim.u[c].upload(with_ghosts(a))
    sim.start()
    for _ in range(args.warmup):
        sim.step()
    dom.synchronize()

    def barrier():
        dom.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize(local_rank)
            dist.barrier()

    barrier()
    comm0 = dom.comm_stats() if rccl_world else (0, 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
    barrier()
    elapsed = time.perf_counter() - t0
    comm1 = dom.comm_stats() if rccl_world else (0, 0)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline of the dominant kernel: the exact-order relax loop of the 256^3 level (nrelax = 4
    # sweeps with the homogeneous BC between them, src/poisson.c:1070-1089) -- on a periodic box one
    # launch of relax_skew_loop_kernel with the sweeps pipelined behind each other; timed with HIP
    # events on the library's stream.  Algorithmic traffic: 24 B per cell and sweep.
    # (with several boxes the relax loop of a box has the halo exchange between its sweeps: the kernel
    # entry is then measured by rank 0 on a periodic box of its own, after the timed region)
    roofline = None
    rdom = dom if world == 1 and not args.self_mpi else (
        gfship.Domain(3, args.level, [gfship.SIDE_PERIODIC] * 6, device=local_rank) if rank == 0
        else None)
    if rdom is not None:
        roofline = measure_roofline(rdom, args, n)

    # config D: the same box with 2e6 GfsParticle tracers (positions from the
    # fixed-seed LCG, ids 1..Np): the particle event alone, and the step with the event in it
    particles = None
    if world == 1 and args.particles > 0:
        from particle_cases import lcg_positions_fast
        pos, ids = lcg_positions_fast(args.particles)
        pl = gfship.ParticleList(sim, pos, ids)
        pl.event()
        dom.synchronize()
        nev = 32
        t0 = time.perf_counter()
        for _ in range(nev):
            pl.event()
        dom.synchronize()
        t_event = (time.perf_counter() - t0) / nev
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pl.event()          # simulation_run: events first, then the flow step
            sim.step()
        dom.synchronize()
        t_both = (time.perf_counter() - t0) / args.steps
        particles = {"workload": "config D: %d tracers (RK2 midpoint, corner interpolation, "
                                 "periodic wrap), slots re-sorted by cell every 16 events"
                                 % args.particles,
                     "n": args.particles, "events": nev,
                     "value": args.particles / t_event / 1e6, "unit": "Mparticle-steps/s",
                     "ms_per_event": t_event * 1e3,
                     "combined_mcell_steps_per_s": n ** 3 / t_both / 1e6,
                     "alive": pl.count()}
        pl.destroy()
        # the same particles as GfsParticulate objects with the list's forces (inertial, added mass,
        # lift, drag -- inactive without viscosity --, buoyancy): the event alone
        rng = np.random.default_rng(1)
        vol = 1e-6 * (0.5 + rng.random(args.particles))
        pp = gfship.ParticleList(sim, pos, ids)
        pp.set_particulate(np.zeros((args.particles, 3)), 2. * vol, vol)
        pp.set_forces([gfship.FORCE_INERTIAL, gfship.FORCE_ADDEDMASS, gfship.FORCE_LIFT,
                       gfship.FORCE_DRAG, gfship.FORCE_BUOY], (0., 1., 0.))
        pp.event()
        dom.synchronize()
        t0 = time.perf_counter()
        for _ in range(nev):
            pp.event()
        dom.synchronize()
        t_part = (time.perf_counter() - t0) / nev
        particles["particulates"] = {"forces": "inertial, added mass, lift, drag, buoyancy",
                                     "value": args.particles / t_part / 1e6,
                                     "unit": "Mparticle-steps/s", "ms_per_event": t_part * 1e3,
                                     "alive": pp.count()}
        pp.destroy()

    if rank == 0:
        value = world * n ** 3 * args.steps / elapsed / 1e6
        out = {
            "metric": "Mcell-steps/s (projection+advection), 3D %d^3 uniform" % n,
            "value": value, "unit": "Mcell-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D periodic Taylor-Green, %d^3 per GPU, default "
                                   "projection/advection parameters (SURVEY 8d config C)" % n,
                       "relax_mode": args.mode,
                       "parallelism": ("1 box" if not args.self_mpi else
                                       "1 box, all sides GfsBoundaryMpi to itself (lab)") if world == 1 else
                                      "%d boxes of %d^3, lattice %s, one per GPU, halo exchange "
                                      "%s (reference semantics: overlap = 0)"
                                      % (world, n, "x".join(map(str, grid.b)),
                                         "by ncclSend/ncclRecv inside libgfship" if backend == "rccl"
                                         else "staged through the host: " + backend),
                       "rccl_world_size": rccl_world,
                       "poisson_niter": [int(sim.projection_params.niter),
                                         int(sim.approx_projection_params.niter)]},
            "roofline": roofline,
        }
        if rccl_world:
            # what rank 0 sent through the library's communicator during the timed steps
            # (gfship_domain_comm_stats: domain->mpi_messages / mpi_size of the reference)
            out["comm"] = {"messages_per_step": (comm1[0] - comm0[0]) / args.steps,
                           "bytes_per_step": (comm1[1] - comm0[1]) / args.steps,
                           "overlap": 0,
                           "per": "rank 0"}
        if particles is not None:
            out["particles"] = particles
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
