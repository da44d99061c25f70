"""ctypes binding of the CPU oracle (oracle/libgfsoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAXLEVEL = 12
SIDE_PERIODIC, SIDE_BOUNDARY, SIDE_EXTERNAL = 0, 1, 2
BC_SYMMETRY, BC_DIRICHLET, BC_NEUMANN = 0, 1, 2


class Norm(C.Structure):
    _fields_ = [("bias", C.c_double), ("first", C.c_double), ("second", C.c_double),
                ("infty", C.c_double), ("w", C.c_double)]

    def as_tuple(self):
        return (self.bias, self.first, self.second, self.infty, self.w)


class MultilevelParams(C.Structure):
    """GfsMultilevelParams (poisson.h:39-52)."""
    _fields_ = [("tolerance", C.c_double), ("nrelax", C.c_uint), ("erelax", C.c_uint),
                ("minlevel", C.c_uint), ("nitermax", C.c_uint), ("nitermin", C.c_uint),
                ("dimension", C.c_uint), ("niter", C.c_uint), ("depth", C.c_uint),
                ("weighted", C.c_int), ("function", C.c_int),
                ("beta", C.c_double), ("omega", C.c_double),
                ("residual_before", Norm), ("residual", Norm)]


def build(force=False):
    so = os.path.join(_HERE, "libgfsoracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or \
            any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i, d, u = C.c_void_p, C.c_int, C.c_double, C.c_uint
        pd = C.POINTER(C.c_double)
        sig = {
            "go_domain_new": (vp, [i, i, C.POINTER(C.c_int)]),
            "go_domain_destroy": (None, [vp]),
            "go_level_size": (C.c_size_t, [vp, i]),
            "go_order": (C.POINTER(C.c_int), [vp, i]),
            "go_field_new": (vp, [vp, i]),
            "go_field_destroy": (None, [vp]),
            "go_field_level": (pd, [vp, i]),
            "go_field_set_bc": (None, [vp, i, i, pd]),
            "go_bc": (None, [vp, vp, i]),
            "go_homogeneous_bc": (None, [vp, vp, i]),
            "go_multilevel_params_init": (None, [C.POINTER(MultilevelParams), i]),
            "go_poisson_coefficients": (None, [vp]),
            "go_relax": (None, [vp, u, i, d, vp, vp, vp]),
            "go_relax_lexicographic": (None, [vp, u, i, d, vp, vp, vp]),
            "go_residual": (None, [vp, u, i, vp, vp, vp, vp]),
            "go_norm_residual": (Norm, [vp, d, vp]),
            "go_norm_variable": (Norm, [vp, vp]),
            "go_poisson_cycle": (None, [vp, C.POINTER(MultilevelParams), vp, vp, vp, vp]),
            "go_poisson_solve": (None, [vp, C.POINTER(MultilevelParams), vp, vp, vp, vp, d]),
        }
        for name, (res, args) in sig.items():
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB


class Field:
    def __init__(self, dom, component=-1):
        self.dom = dom
        self.ptr = lib().go_field_new(dom.ptr, component)

    def level(self, l):
        """numpy view (with ghosts) of level l, shape (n+2,)*dim, indexed [k, j, i] / [j, i]."""
        n = (1 << l) + 2
        shape = (n,) * self.dom.dim
        p = lib().go_field_level(self.ptr, l)
        return np.ctypeslib.as_array(p, shape=shape)

    def leaf(self):
        return self.level(self.dom.depth)

    def interior(self, l=None):
        l = self.dom.depth if l is None else l
        a = self.level(l)
        return a[(slice(1, -1),) * self.dom.dim]

    def set_bc(self, d, kind, val=None):
        if val is not None:
            val = np.ascontiguousarray(val, dtype=np.float64).ravel()
            lib().go_field_set_bc(self.ptr, d, kind, val.ctypes.data_as(C.POINTER(C.c_double)))
        else:
            lib().go_field_set_bc(self.ptr, d, kind, None)

    def __del__(self):
        try:
            lib().go_field_destroy(self.ptr)
        except Exception:
            pass


class Domain:
    def __init__(self, dim, depth, side=None):
        self.dim, self.depth = dim, depth
        s = (C.c_int * 6)(*(side if side is not None else [SIDE_BOUNDARY] * 6))
        self.side = list(s)
        self.ptr = lib().go_domain_new(dim, depth, s)

    def field(self, component=-1):
        return Field(self, component)

    def params(self):
        p = MultilevelParams()
        lib().go_multilevel_params_init(C.byref(p), self.dim)
        return p

    def snapshot_tree(self, fields):
        """ftt_cell_write_binary + gfs_cell_write_binary of the box (bytes)"""
        L = lib()
        L.go_snapshot_tree_bytes.restype, L.go_snapshot_tree_bytes.argtypes = C.c_size_t, [C.c_void_p, C.c_int]
        L.go_snapshot_tree_write.restype = C.c_size_t
        L.go_snapshot_tree_write.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
        n = len(fields)
        size = L.go_snapshot_tree_bytes(self.ptr, n)
        buf = C.create_string_buffer(size)
        arr = (C.c_void_p * n)(*[f.ptr for f in fields])
        assert L.go_snapshot_tree_write(self.ptr, n, arr, buf) == size
        return buf.raw

    def poisson_coefficients_alpha(self, alpha):
        """gfs_poisson_coefficients with a GfsFunction alpha given as face values: alpha = dim Fields"""
        f = lib().go_poisson_coefficients_alpha
        f.restype, f.argtypes = None, [C.c_void_p, C.POINTER(C.POINTER(C.c_double))]
        arr = (C.POINTER(C.c_double) * 3)()
        for c, a in enumerate(alpha):
            arr[c] = lib().go_field_level(a.ptr, self.depth)
        f(self.ptr, arr)

    def weight(self, d, l=None):
        """f[d].v of the cells of level l (array with ghosts)"""
        l = self.depth if l is None else l
        f = lib().go_domain_weight
        f.restype, f.argtypes = C.POINTER(C.c_double), [C.c_void_p, C.c_int, C.c_int]
        n = (1 << l) + 2
        return np.ctypeslib.as_array(f(self.ptr, d, l), shape=(n,) * self.dim)

    def set_overlap(self, overlap):
        """the domain parameter `overlap' of a parallel run (src/domain.c:225,682)"""
        f = lib().go_domain_set_overlap
        f.restype, f.argtypes = None, [C.c_void_p, C.c_int]
        f(self.ptr, int(overlap))

    def centres(self, l=None):
        """cell-centre coordinate arrays (x, y[, z]) of the interior of level l, broadcastable
        against interior() views."""
        l = self.depth if l is None else l
        n = 1 << l
        c = -0.5 + (np.arange(1, n + 1) - 0.5) / n
        if self.dim == 2:
            return c[None, :], c[:, None]
        return c[None, None, :], c[None, :, None], c[:, None, None]

    def order(self, l):
        n = 1 << l
        return np.ctypeslib.as_array(lib().go_order(self.ptr, l), shape=(n ** self.dim,)).copy()

    def __del__(self):
        try:
            lib().go_domain_destroy(self.ptr)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# GfsSimulation (oracle/go_timestep.c)
# ---------------------------------------------------------------------------------------------

class AdvectionParams(C.Structure):
    _fields_ = [("cfl", C.c_double), ("dt", C.c_double), ("gradient", C.c_int), ("gc", C.c_int)]


def _sim_sigs(L):
    vp, i, d, u = C.c_void_p, C.c_int, C.c_double, C.c_uint
    pd = C.POINTER(C.c_double)
    sig = {
        "go_sim_new": (vp, [i, i, C.POINTER(C.c_int)]),
        "go_sim_destroy": (None, [vp]),
        "go_sim_field": (vp, [vp, i, i]),
        "go_sim_domain": (vp, [vp]),
        "go_sim_projection_params": (C.POINTER(MultilevelParams), [vp]),
        "go_sim_approx_projection_params": (C.POINTER(MultilevelParams), [vp]),
        "go_sim_advection_params": (C.POINTER(AdvectionParams), [vp]),
        "go_sim_un": (pd, [vp, i]),
        "go_sim_fv": (pd, [vp, i]),
        "go_sim_time": (d, [vp]),
        "go_sim_iter": (u, [vp]),
        "go_sim_set_time": (None, [vp, d, d]),
        "go_sim_add_tracer": (i, [vp]),
        "go_predicted_face_velocities": (None, [vp]),
        "go_domain_cfl": (d, [vp]),
        "go_set_timestep": (None, [vp]),
        "go_advection_step": (None, [vp]),
        "go_sim_set_tracer_gradient": (None, [vp, i, i]),
        "go_coarse_init": (None, [vp]),
        "go_sim_start": (None, [vp]),
        "go_sim_step": (None, [vp]),
        "go_divergence": (None, [vp, vp]),
        "go_tracer_advection": (None, [vp, vp, d]),
        "go_sim_set_viscosity": (None, [vp, i, d]),
        "go_sim_set_source": (None, [vp, i, d]),
        "go_sim_set_alpha": (None, [vp, C.POINTER(vp)]),
        "go_variable_diffusion": (None, [vp, vp, vp, d, d, C.POINTER(MultilevelParams)]),
        "go_sim_diffusion_params": (C.POINTER(MultilevelParams), [vp, i]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args


class _FieldView(Field):
    """A field owned by a GoSim (not freed from python)."""

    def __init__(self, dom, ptr):
        self.dom, self.ptr = dom, ptr

    def __del__(self):
        pass


class _DomView(Domain):
    def __init__(self, dim, depth, ptr, side):
        self.dim, self.depth, self.ptr, self.side = dim, depth, ptr, side

    def __del__(self):
        pass


class Sim:
    P, PMAC, U, G, GMAC, TRACER = 0, 1, 2, 3, 4, 5

    def __init__(self, dim, depth, side=None):
        L = lib()
        if not getattr(L, "_sim_ready", False):
            _sim_sigs(L)
            L._sim_ready = True
        s = (C.c_int * 6)(*(side if side is not None else [SIDE_BOUNDARY] * 6))
        self.dim, self.depth = dim, depth
        self.end = 1.7976931348623157e308
        self.ptr = L.go_sim_new(dim, depth, s)
        self.dom = _DomView(dim, depth, L.go_sim_domain(self.ptr), list(s))
        self.p = self.field(self.P)
        self.pmac = self.field(self.PMAC)
        self.u = [self.field(self.U, c) for c in range(dim)]
        self.g = [self.field(self.G, c) for c in range(dim)]
        self.gmac = [self.field(self.GMAC, c) for c in range(dim)]
        self.projection_params = L.go_sim_projection_params(self.ptr).contents
        self.approx_projection_params = L.go_sim_approx_projection_params(self.ptr).contents
        self.advection_params = L.go_sim_advection_params(self.ptr).contents

    def field(self, which, c=0):
        return _FieldView(self.dom, lib().go_sim_field(self.ptr, which, c))

    def add_tracer(self, gradient=None):
        """GfsVariableTracer; gradient 0 = gfs_center_gradient, 1 = van Leer (the default)"""
        t = lib().go_sim_add_tracer(self.ptr)
        if gradient is not None:
            lib().go_sim_set_tracer_gradient(self.ptr, t, gradient)
        return self.field(self.TRACER, t)

    def un(self, d):
        n = (1 << self.depth) + 2
        return np.ctypeslib.as_array(lib().go_sim_un(self.ptr, d), shape=(n,) * self.dim)

    def stream_function(self, psi):
        """GfsVariableStreamFunction (2-D, src/variable.c:931-944,1041-1086): MAC velocities of every
        leaf from psi(x, y) at its four corners, centred velocities = means of the two faces, BC"""
        assert self.dim == 2
        n = 1 << self.depth
        h = 1. / n
        xv = -0.5 + np.arange(n + 1) * h
        # corner positions exactly as init_streamfunction computes them: cell centre -+ h/2
        xc = -0.5 + (np.arange(n) + 0.5) * h
        xm, xp = xc - h / 2., xc + h / 2.
        Y0, X0 = np.meshgrid(xm, xm, indexing="ij")
        Y1, X1 = np.meshgrid(xm, xp, indexing="ij")
        Y2, X2 = np.meshgrid(xp, xp, indexing="ij")
        Y3, X3 = np.meshgrid(xp, xm, indexing="ij")
        p0, p1, p2, p3 = psi(X0, Y0), psi(X1, Y1), psi(X2, Y2), psi(X3, Y3)
        hh = 2. * (h / 2.)
        un = [(p2 - p1) * 1. / hh, (p3 - p0) * 1. / hh, (p3 - p2) * 1. / hh, (p0 - p1) * 1. / hh]
        for d in range(4):
            self.un(d)[1:-1, 1:-1] = un[d]
        self.u[0].interior()[...] = (un[0] + un[1]) / 2.
        self.u[1].interior()[...] = (un[2] + un[3]) / 2.
        for c in range(2):
            lib().go_bc(self.u[c].ptr, self.u[c].ptr, self.depth)

    def advection_step(self):
        lib().go_advection_step(self.ptr)

    def fv(self, d):
        n = (1 << self.depth) + 2
        return np.ctypeslib.as_array(lib().go_sim_fv(self.ptr, d), shape=(n,) * self.dim)

    def set_source(self, c, g):
        """GfsSource {} U/V/W g: constant intensity (a body force per unit mass)"""
        lib().go_sim_set_source(self.ptr, c, g)

    def set_alpha(self, alpha):
        """GfsPhysicalParams { alpha }: dim Fields of face values (None: alpha = NULL)"""
        if alpha is None:
            lib().go_sim_set_alpha(self.ptr, None)
            return
        arr = (C.c_void_p * 3)(*([a.ptr for a in alpha] + [None] * (3 - len(alpha))))
        self._alpha = list(alpha)
        lib().go_sim_set_alpha(self.ptr, arr)

    def set_viscosity(self, c, nu):
        """SourceDiffusion {} U|V|W nu (implicit, Crank-Nicholson beta = 1 by default)"""
        lib().go_sim_set_viscosity(self.ptr, c, nu)

    def variable_diffusion(self, v, rhs, D, dt, par):
        """variable_diffusion (src/timestep.c:923-949): implicit solve of v with source rhs"""
        lib().go_variable_diffusion(self.ptr, v.ptr, rhs.ptr, D, dt, C.byref(par))

    def diffusion_params(self, c):
        return lib().go_sim_diffusion_params(self.ptr, c).contents

    def set_time(self, end=1.7976931348623157e308, dtmax=1.7976931348623157e308):
        self.end = end
        lib().go_sim_set_time(self.ptr, end, dtmax)

    @property
    def t(self):
        return lib().go_sim_time(self.ptr)

    @property
    def i(self):
        return lib().go_sim_iter(self.ptr)

    @property
    def dt(self):
        return self.advection_params.dt

    def restart(self, t, i):
        f = lib().go_sim_restart
        f.restype, f.argtypes = None, [C.c_void_p, C.c_double, C.c_uint]
        f(self.ptr, t, i)

    def start(self):
        lib().go_sim_start(self.ptr)

    def step(self):
        lib().go_sim_step(self.ptr)

    def output_location(self, points, fields):
        """GfsOutputLocation (src/output.c:1153-1203): locate each point, then gfs_interpolate of
        every field there; rows of points outside the domain are dropped."""
        L = lib()
        pd = C.POINTER(C.c_double)
        L.go_locate.restype, L.go_locate.argtypes = C.c_int, [C.c_void_p, pd, C.POINTER(C.c_int)]
        L.go_interpolate.restype = C.c_double
        L.go_interpolate.argtypes = [C.c_void_p, pd, C.POINTER(C.c_int), pd]
        rows = []
        for p in np.asarray(points, dtype=np.float64).reshape(-1, 3):
            p = np.ascontiguousarray(p)
            ijk = (C.c_int * 3)()
            if not L.go_locate(self.dom.ptr, p.ctypes.data_as(pd), ijk):
                continue
            rows.append(list(p) + [L.go_interpolate(self.dom.ptr,
                                                    L.go_field_level(f.ptr, self.depth), ijk,
                                                    p.ctypes.data_as(pd)) for f in fields])
        return np.array(rows)

    def set_bc(self, d, kind):
        """condition of P on side d (BC_DIRICHLET / BC_NEUMANN / BC_SYMMETRY)"""
        self.L.gt_set_bc(self.ptr, d, kind)

    def bc_values(self, l):
        """the values of the GfsFunction of the conditions at the face centres, one per ghost cell"""
        r = (1 << l) + 2
        return np.ctypeslib.as_array(self.L.gt_bc_values(self.ptr, l), shape=(r,) * self.dim)

    def poisson_run(self):
        """one iteration of poisson_run: Div in the PMAC slot, the guess in P"""
        self.L.gt_poisson_run(self.ptr, None)

    def divergence_norm(self):
        e = self.dom.field()
        lib().go_divergence(self.ptr, e.ptr)
        return lib().go_norm_variable(self.dom.ptr, e.ptr)

    def __del__(self):
        try:
            lib().go_sim_destroy(self.ptr)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# GfsParticleList (oracle/go_particles.c)
# ---------------------------------------------------------------------------------------------

FORCE_INERTIAL, FORCE_ADDEDMASS, FORCE_LIFT, FORCE_DRAG, FORCE_BUOY = 1, 2, 3, 4, 5


class Particles:
    def __init__(self, sim, pos, ids):
        L = lib()
        if not getattr(L, "_part_ready", False):
            vp, i = C.c_void_p, C.c_int
            pd, pu = C.POINTER(C.c_double), C.POINTER(C.c_uint)
            for name, (res, args) in {
                "go_particles_new": (vp, [i, pd, pu]),
                "go_particles_destroy": (None, [vp]),
                "go_particles_count": (i, [vp]),
                "go_particles_pos": (pd, [vp]),
                "go_particles_id": (pu, [vp]),
                "go_particle_list_event": (None, [vp, vp]),
                "go_locate": (i, [vp, pd, C.POINTER(C.c_int)]),
                "go_particles_outbox": (i, [vp, i, C.POINTER(pd)]),
                "go_particles_record_size": (i, [vp]),
                "go_particles_clear_outbox": (None, [vp]),
                "go_particles_append": (None, [vp, i, pd]),
                "go_particles_set_particulate": (None, [vp, pd, pd, pd]),
                "go_particles_set_forces": (None, [vp, vp, i, C.POINTER(i)]),
                "go_particles_set_gravity": (None, [vp, pd]),
                "go_particles_set_coefficient": (None, [vp, i, vp]),
                "go_particles_vel": (pd, [vp]),
                "go_particles_mass": (pd, [vp]),
                "go_particles_force": (pd, [vp]),
            }.items():
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
            L._part_ready = True
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self.sim = sim
        self.ptr = L.go_particles_new(len(ids), pos.ctypes.data_as(C.POINTER(C.c_double)),
                                      ids.ctypes.data_as(C.POINTER(C.c_uint)))

    def event(self):
        lib().go_particle_list_event(self.sim.ptr, self.ptr)

    def count(self):
        return lib().go_particles_count(self.ptr)

    def outbox(self, d):
        """records (n x 7: pos, pos_old, id; n x 15 for particulates: + vel, mass, volume, force) of
        the particles sent through MPI side d"""
        rs = lib().go_particles_record_size(self.ptr)
        rec = C.POINTER(C.c_double)()
        n = lib().go_particles_outbox(self.ptr, d, C.byref(rec))
        if n == 0:
            return np.empty((0, rs))
        return np.ctypeslib.as_array(rec, shape=(n, rs)).copy()

    def clear_outbox(self):
        lib().go_particles_clear_outbox(self.ptr)

    def append(self, rec):
        rec = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, lib().go_particles_record_size(self.ptr))
        if len(rec):
            lib().go_particles_append(self.ptr, len(rec), rec.ctypes.data_as(C.POINTER(C.c_double)))

    # GfsParticulate: velocity, mass, volume and the list's forces (FORCE_* in application order)
    def set_particulate(self, vel, mass, volume):
        pd = C.POINTER(C.c_double)
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        mass = np.ascontiguousarray(mass, dtype=np.float64)
        volume = np.ascontiguousarray(volume, dtype=np.float64)
        assert len(vel) == len(mass) == len(volume) == self.count()
        lib().go_particles_set_particulate(self.ptr, vel.ctypes.data_as(pd), mass.ctypes.data_as(pd),
                                           volume.ctypes.data_as(pd))

    def set_forces(self, kinds, gravity=(0., 0., 0.)):
        k = (C.c_int * len(kinds))(*kinds)
        g = (C.c_double * 3)(*gravity)
        lib().go_particles_set_gravity(self.ptr, g)
        lib().go_particles_set_forces(self.ptr, self.sim.ptr, len(kinds), k)

    COEFF_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)

    def set_coefficient(self, force, fn):
        """the GfsFunction of force number `force' of the list: fn (Rep, Urelp, Vrelp, Wrelp, Pdia)"""
        cb = Particles.COEFF_FN(fn)
        self._keep = getattr(self, "_keep", []) + [cb]
        lib().go_particles_set_coefficient(self.ptr, force, C.cast(cb, C.c_void_p))

    def particulate_state(self):
        """(vel, mass, force) of the particles on the list, in list order"""
        n = self.count()
        L = lib()
        vel = np.ctypeslib.as_array(L.go_particles_vel(self.ptr), shape=(max(n, 1), 3))[:n].copy()
        mass = np.ctypeslib.as_array(L.go_particles_mass(self.ptr), shape=(max(n, 1),))[:n].copy()
        force = np.ctypeslib.as_array(L.go_particles_force(self.ptr), shape=(max(n, 1), 3))[:n].copy()
        return vel, mass, force

    def state(self):
        n = self.count()
        pos = np.ctypeslib.as_array(lib().go_particles_pos(self.ptr), shape=(max(n, 1), 3))[:n].copy()
        ids = np.ctypeslib.as_array(lib().go_particles_id(self.ptr), shape=(max(n, 1),))[:n].copy()
        return pos, ids

    def locate(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        ijk = (C.c_int * 3)()
        ok = lib().go_locate(self.sim.dom.ptr, p.ctypes.data_as(C.POINTER(C.c_double)), ijk)
        return tuple(ijk) if ok else None

    def __del__(self):
        try:
            lib().go_particles_destroy(self.ptr)
        except Exception:
            pass


REFINE_FUNC = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p)


def _tree_sigs(L):
    vp, i, d, u = C.c_void_p, C.c_int, C.c_double, C.c_uint
    pd = C.POINTER(C.c_double)
    sig = {
        "gt_new": (vp, [i, REFINE_FUNC, vp]),
        "gt_new_with_sides": (vp, [i, REFINE_FUNC, vp, C.POINTER(C.c_int)]),
        "gt_set_bc": (None, [vp, i, i]),
        "gt_bc_values": (pd, [vp, i]),
        "gt_poisson_run": (None, [vp, vp]),
        "gt_dim": (i, [vp]),
        "gt_periodic_new": (vp, [i, i]),
        "gt_destroy": (None, [vp]),
        "gt_set_time": (None, [vp, d, d]),
        "gt_start": (None, [vp]),
        "gt_step": (None, [vp]),
        "gt_run": (u, [vp]),
        "gt_time": (d, [vp]),
        "gt_end": (d, [vp]),
        "gt_dt": (d, [vp]),
        "gt_iter": (u, [vp]),
        "gt_depth": (i, [vp]),
        "gt_projection_params": (C.POINTER(MultilevelParams), [vp, i]),
        "gt_flags": (C.POINTER(C.c_ubyte), [vp, i]),
        "gt_values": (pd, [vp, i, i]),
        "gt_error_norm": (None, [vp, pd, pd, pd]),
        "gt_divergence_norm": (None, [vp, pd, pd, pd, pd]),
        "gt_divergence_level": (None, [vp, i, pd]),
        "gt_add_tracer": (i, [vp, i]),
        "gt_set_bc_u": (None, [vp, i, i, i]),
        "gt_bc_values_u": (pd, [vp, i, i]),
        "gt_set_viscosity": (None, [vp, i, d]),
        "gt_set_source": (None, [vp, i, d]),
        "gt_diffusion_params": (C.POINTER(MultilevelParams), [vp, i]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args


class Tree:
    """go_tree.c: a GfsSimulation on one periodic 2-D / 3-D box with a statically refined tree
    (coarse-fine stencils).  Levels are dense (n+2)^dim arrays [(k,) j, i] with a flag per cell:
    0 absent, 1 leaf, 2 non-leaf.  refine (x, y) or refine (x, y, z) -> level wanted there."""
    U, V, P, PMAC, GX, GY, GMACX, GMACY, UN0, UN1, UN2, UN3, W, GZ, GMACZ, UN4, UN5 = range(17)

    def __init__(self, refine=None, periodic=None, dim=2, sides=None):
        L = lib()
        if not hasattr(L, "_tree_ready"):
            _tree_sigs(L)
            L._tree_ready = True
        self.L = L
        if periodic is not None:
            self.ptr = L.gt_periodic_new(*periodic)
        else:
            if dim == 2:
                self._cb = REFINE_FUNC(lambda x, y, z, ctx: float(refine(x, y)))
            else:
                self._cb = REFINE_FUNC(lambda x, y, z, ctx: float(refine(x, y, z)))
            if sides is None:
                self.ptr = L.gt_new(dim, self._cb, None)
            else:
                self.ptr = L.gt_new_with_sides(dim, self._cb, None, (C.c_int * 6)(*(list(sides) + [0] * 6)[:6]))
        self.dim = L.gt_dim(self.ptr)
        self.depth = L.gt_depth(self.ptr)
        self.projection_params = L.gt_projection_params(self.ptr, 0).contents
        self.approx_projection_params = L.gt_projection_params(self.ptr, 1).contents

    def flags(self, l):
        r = (1 << l) + 2
        return np.ctypeslib.as_array(self.L.gt_flags(self.ptr, l), shape=(r,) * self.dim)

    def values(self, which, l):
        r = (1 << l) + 2
        return np.ctypeslib.as_array(self.L.gt_values(self.ptr, which, l), shape=(r,) * self.dim)

    def centres(self, l):
        """x, y (, z) of the cells of level l (ghosts included), indexed [(k,) j, i]"""
        n = 1 << l
        c = -0.5 + (np.arange(n + 2) - 0.5) / n
        if self.dim == 2:
            return np.meshgrid(c, c, indexing="xy")
        z, y, x = np.meshgrid(c, c, c, indexing="ij")
        return x, y, z

    def set_time(self, end, cfl):
        self.L.gt_set_time(self.ptr, end, cfl)

    def set_bc_u(self, c, d, kind, value=0.):
        """condition of velocity component c on side d (BC_SYMMETRY / BC_DIRICHLET / BC_NEUMANN) with a
        constant value, or a function value (x, y[, z]) of the face centres"""
        self.L.gt_set_bc_u(self.ptr, c, d, kind)
        for l in range(self.depth + 1):
            r = (1 << l) + 2
            a = np.ctypeslib.as_array(self.L.gt_bc_values_u(self.ptr, c, l), shape=(r,) * self.dim)
            sl = [slice(None)] * self.dim
            ax = self.dim - 1 - d // 2
            sl[ax] = 0 if d & 1 else r - 1
            if callable(value):
                cs = list(self.centres(l))
                h = 1. / (1 << l)
                cs[d // 2] = cs[d // 2] + (0.5 * h if d & 1 else - 0.5 * h)      # ghost centre -> face centre
                a[tuple(sl)] = (value(*cs) + 0. * cs[0])[tuple(sl)]
            else:
                a[tuple(sl)] = value

    def set_viscosity(self, c, nu):
        self.L.gt_set_viscosity(self.ptr, c, nu)

    def set_source(self, c, g):
        """GfsSource {} U|V|W g: constant intensity"""
        self.L.gt_set_source(self.ptr, c, g)

    def diffusion_params(self, c):
        return self.L.gt_diffusion_params(self.ptr, c).contents

    def add_tracer(self, gradient=1):
        """GfsVariableTracer [{ gradient = }] (0 centred, 1 van Leer): the index for values ()"""
        k = self.L.gt_add_tracer(self.ptr, gradient)
        assert k >= 0
        return k

    def start(self):
        self.L.gt_start(self.ptr)

    def step(self):
        self.L.gt_step(self.ptr)

    def run(self):
        return self.L.gt_run(self.ptr)

    t = property(lambda self: self.L.gt_time(self.ptr))
    end = property(lambda self: self.L.gt_end(self.ptr))
    dt = property(lambda self: self.L.gt_dt(self.ptr))
    i = property(lambda self: self.L.gt_iter(self.ptr))

    def error_norm(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.L.gt_error_norm(self.ptr, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def set_bc(self, d, kind):
        """condition of P on side d (BC_DIRICHLET / BC_NEUMANN / BC_SYMMETRY)"""
        self.L.gt_set_bc(self.ptr, d, kind)

    def bc_values(self, l):
        """the values of the GfsFunction of the conditions at the face centres, one per ghost cell"""
        r = (1 << l) + 2
        return np.ctypeslib.as_array(self.L.gt_bc_values(self.ptr, l), shape=(r,) * self.dim)

    def poisson_run(self):
        """one iteration of poisson_run: Div in the PMAC slot, the guess in P"""
        self.L.gt_poisson_run(self.ptr, None)

    def divergence_norm(self):
        """(first, second, infty) of the derived variable Divergence, and the sum of Velocity2"""
        a, b, c, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        self.L.gt_divergence_norm(self.ptr, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value

    def divergence(self, l):
        r = (1 << l) + 2
        out = np.zeros((r,) * self.dim)
        self.L.gt_divergence_level(self.ptr, l, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def destroy(self):
        if self.ptr:
            self.L.gt_destroy(self.ptr)
            self.ptr = None
