"""gfship -- host-side mirror of the libgfship C ABI (include/gfship.h).

Thin ctypes layer used by tests/, bench.py and __graft_entry__.py: object names follow the
reference (domain, variables, GfsMultilevelParams).  There is no CPU fallback: loading fails
loudly when the HIP library is missing, and creating a domain fails without a device.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("GFSHIP_LIB", os.path.join(_PKG, "lib", "libgfship.so"))

SIDE_PERIODIC, SIDE_BOUNDARY, SIDE_EXTERNAL = 0, 1, 2
BC_SYMMETRY, BC_DIRICHLET, BC_NEUMANN = 0, 1, 2
RELAX_EXACT, RELAX_REDBLACK, RELAX_EXACT_HYPERPLANE, RELAX_EXACT_PER_SWEEP = 0, 1, 2, 3


class GfshipError(RuntimeError):
    pass


class Norm(C.Structure):
    _fields_ = [("bias", C.c_double), ("first", C.c_double), ("second", C.c_double),
                ("infty", C.c_double), ("w", C.c_double)]


class MultilevelParams(C.Structure):
    """GfsMultilevelParams (src/poisson.h:39-52)."""
    _fields_ = [("tolerance", C.c_double), ("nrelax", C.c_uint), ("erelax", C.c_uint),
                ("minlevel", C.c_uint), ("nitermax", C.c_uint), ("nitermin", C.c_uint),
                ("dimension", C.c_uint), ("niter", C.c_uint), ("depth", C.c_uint),
                ("weighted", C.c_int), ("function", C.c_int),
                ("beta", C.c_double), ("omega", C.c_double),
                ("residual_before", Norm), ("residual", Norm)]


class AdvectionParams(C.Structure):
    """The fields of GfsAdvectionParams used on this path (src/advection.h:50-69)."""
    _fields_ = [("cfl", C.c_double), ("dt", C.c_double), ("gradient", C.c_int), ("gc", C.c_int)]


_lib = None

# every symbol include/gfship.h declares: name -> (restype, argtypes)
_vp, _i, _d, _u = C.c_void_p, C.c_int, C.c_double, C.c_uint
_pd = C.POINTER(C.c_double)
_pi = C.POINTER(C.c_int)
SIGNATURES = {
    "gfship_last_error": (C.c_char_p, []),
    "gfship_version": (_i, []),
    "gfship_device_count": (_i, []),
    "gfship_domain_create": (_i, [C.POINTER(_vp), _i, _i, _pi, _i]),
    "gfship_domain_destroy": (None, [_vp]),
    "gfship_domain_set_relax_mode": (_i, [_vp, _i]),
    "gfship_domain_synchronize": (_i, [_vp]),
    "gfship_domain_stream": (_vp, [_vp]),
    "gfship_field_alloc": (_i, [_vp, _i]),
    "gfship_field_free": (_i, [_vp, _i]),
    "gfship_field_set_bc": (_i, [_vp, _i, _i, _i, _pd]),
    "gfship_field_upload": (_i, [_vp, _i, _i, _pd]),
    "gfship_field_download": (_i, [_vp, _i, _i, _pd]),
    "gfship_field_fill": (_i, [_vp, _i, _i, _d]),
    "gfship_field_device_ptr": (_vp, [_vp, _i, _i, _pi, _pi]),
    "gfship_bc": (_i, [_vp, _i, _i, _i]),
    "gfship_homogeneous_bc": (_i, [_vp, _i, _i, _i]),
    "gfship_multilevel_params_init": (None, [C.POINTER(MultilevelParams), _i]),
    "gfship_poisson_coefficients": (_i, [_vp]),
    "gfship_poisson_coefficients_alpha": (_i, [_vp, _pi]),
    "gfship_poisson_weights": (_i, [_vp, _i, _pi]),
    "gfship_relax": (_i, [_vp, _u, _i, _d, _i, _i, _i]),
    "gfship_residual": (_i, [_vp, _u, _i, _i, _i, _i, _i]),
    "gfship_norm_residual": (_i, [_vp, _d, _i, C.POINTER(Norm)]),
    "gfship_norm_variable": (_i, [_vp, _i, C.POINTER(Norm)]),
    "gfship_poisson_cycle": (_i, [_vp, C.POINTER(MultilevelParams), _i, _i, _i, _i]),
    "gfship_poisson_solve": (_i, [_vp, C.POINTER(MultilevelParams), _i, _i, _i, _i, _d]),
    "gfship_diffusion_coefficients": (_i, [_vp, _d, _d, _i, _d]),
    "gfship_diffusion_rhs": (_i, [_vp, _i, _i, _i, _d]),
    "gfship_diffusion_residual": (_i, [_vp, _i, _i, _i, _i]),
    "gfship_diffusion_cycle": (_i, [_vp, _u, _u, _u, _i, _i, _i, _i]),
    "gfship_diffusion": (_i, [_vp, C.POINTER(MultilevelParams), _i, _i, _i]),
    "gfship_time_relax": (_i, [_vp, _u, _i, _i, _i, _i, _i, _pd]),
    "gfship_time_relax_loop": (_i, [_vp, _i, _i, _i, _i, _u, _i, _pd, _pi]),
    "gfship_time_relax_loop_inclusive": (_i, [_vp, _i, _i, _i, _i, _u, _i, _pd, _pi, _pd]),
    "gfship_sim_create": (_i, [C.POINTER(_vp), _vp]),
    "gfship_sim_destroy": (None, [_vp]),
    "gfship_sim_variable": (_i, [_vp, _i, _i]),
    "gfship_sim_projection_params": (C.POINTER(MultilevelParams), [_vp]),
    "gfship_sim_approx_projection_params": (C.POINTER(MultilevelParams), [_vp]),
    "gfship_sim_advection_params": (C.POINTER(AdvectionParams), [_vp]),
    "gfship_sim_set_time": (_i, [_vp, _d, _d]),
    "gfship_sim_set_next_event": (_i, [_vp, _vp, _vp]),
    "gfship_sim_time": (_d, [_vp]),
    "gfship_sim_iter": (_u, [_vp]),
    "gfship_sim_add_tracer": (_i, [_vp]),
    "gfship_sim_set_viscosity": (_i, [_vp, _i, _d]),
    "gfship_sim_set_alpha": (_i, [_vp, _pi]),
    "gfship_sim_set_source": (_i, [_vp, _i, _d]),
    "gfship_sim_diffusion_params": (C.POINTER(MultilevelParams), [_vp, _i]),
    "gfship_sim_start": (_i, [_vp]),
    "gfship_sim_step": (_i, [_vp]),
    "gfship_sim_advection_step": (_i, [_vp]),
    "gfship_sim_set_tracer_gradient": (_i, [_vp, _i, _i]),
    "gfship_predicted_face_velocities": (_i, [_vp]),
    "gfship_mac_projection": (_i, [_vp, C.POINTER(MultilevelParams), _d, _i, _pi]),
    "gfship_approximate_projection": (_i, [_vp, C.POINTER(MultilevelParams), _d, _i, _pi]),
    "gfship_centered_velocity_advection": (_i, [_vp, _pi, _pi]),
    "gfship_tracer_advection": (_i, [_vp, _i, _d]),
    "gfship_domain_cfl": (_i, [_vp, _pd]),
    "gfship_set_timestep": (_i, [_vp]),
    "gfship_coarse_init": (_i, [_vp]),
    "gfship_divergence_norm": (_i, [_vp, C.POINTER(Norm)]),
    "gfship_sim_download_un": (_i, [_vp, _i, _pd]),
    "gfship_domain_set_overlap": (_i, [_vp, _i]),
    "gfship_domain_set_exchange": (_i, [_vp, _vp, _vp]),
    "gfship_domain_set_reduce": (_i, [_vp, _vp, _vp]),
    "gfship_domain_set_gather": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(C.c_int)]),
    "gfship_domain_path_counts": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "gfship_sim_restart": (_i, [_vp, _d, _u]),
    "gfship_snapshot_tree_bytes": (C.c_size_t, [_vp, _i]),
    "gfship_snapshot_tree_write": (_i, [_vp, _i, _pi, _vp, C.c_size_t]),
    "gfship_snapshot_tree_read": (_i, [_vp, _i, _pi, _vp, C.c_size_t]),
    "gfship_comm_available": (_i, []),
    "gfship_comm_unique_id": (_i, [_vp]),
    "gfship_domain_comm_init": (_i, [_vp, _vp, _i, _i, _pi]),
    "gfship_domain_comm_size": (_i, [_vp]),
    "gfship_domain_comm_stats": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "gfship_domain_comm_destroy": (_i, [_vp]),
    "gfship_halo_pack": (_i, [_vp, _vp, _i, _i, _vp]),
    "gfship_halo_unpack": (_i, [_vp, _vp, _i, _i, _vp]),
    "gfship_halo_pack_sides": (_i, [_vp, _vp, _i, _i, _pi, C.POINTER(_vp)]),
    "gfship_halo_unpack_sides": (_i, [_vp, _vp, _i, _i, _pi, C.POINTER(_vp)]),
    "gfship_field_interpolate": (_i, [_vp, _i, _i, _pd, _pd, C.POINTER(C.c_ubyte)]),
    "gfship_particles_create": (_i, [C.POINTER(_vp), _vp, _i, _pd, C.POINTER(C.c_uint)]),
    "gfship_particles_destroy": (None, [_vp]),
    "gfship_particle_list_event": (_i, [_vp]),
    "gfship_particles_count": (_i, [_vp]),
    "gfship_particles_set_migrate": (_i, [_vp, _vp, _vp]),
    "gfship_particles_slots": (_i, [_vp]),
    "gfship_particles_record_size": (_i, [_vp]),
    "gfship_particles_sort": (_i, [_vp]),
    "gfship_particles_set_sort_interval": (_i, [_vp, _i]),
    "gfship_particles_download": (_i, [_vp, _pd, C.POINTER(C.c_uint)]),
    "gfship_energy_spectra_bins": (_i, [_vp]),
    "gfship_energy_spectra": (_i, [_vp, _i, C.POINTER(_i), _pd, C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "gfship_init_spectra": (_i, [_vp, _vp, C.POINTER(_i)]),
    "gfship_output_spectra_side": (_i, [_vp]),
    "gfship_output_spectra": (_i, [_vp, _i, _pd, C.POINTER(C.c_double)]),
    "gfship_output_spectra_plane": (_i, [_vp, _i, _i, _d, _pd, _pd]),
    "gfship_turbulent_viscosity": (_i, [_vp, C.POINTER(_i), C.c_double, _i, _i]),
    "gfship_particles_set_particulate": (_i, [_vp, _pd, _pd, _pd]),
    "gfship_particles_set_forces": (_i, [_vp, _i, C.POINTER(_i), _pd]),
    "gfship_particles_set_force_coefficient": (_i, [_vp, _i, C.c_char_p]),
    "gfship_particles_download_particulate": (_i, [_vp, _pd, _pd, _pd]),
    "gfship_tree_create": (_i, [C.POINTER(_vp), _i, C.c_void_p, _vp, _i]),
    "gfship_tree_create_sides": (_i, [C.POINTER(_vp), _i, C.c_void_p, _vp, _pi, _i]),
    "gfship_tree_set_bc": (_i, [_vp, _i, _i]),
    "gfship_tree_poisson_solve": (_i, [_vp, C.POINTER(MultilevelParams), _d]),
    "gfship_tree_destroy": (None, [_vp]),
    "gfship_tree_depth": (_i, [_vp]),
    "gfship_tree_dim": (_i, [_vp]),
    "gfship_tree_flags": (_i, [_vp, _i, C.POINTER(C.c_ubyte)]),
    "gfship_tree_upload": (_i, [_vp, _i, _i, _pd]),
    "gfship_tree_download": (_i, [_vp, _i, _i, _pd]),
    "gfship_tree_projection_params": (C.POINTER(MultilevelParams), [_vp, _i]),
    "gfship_tree_set_time": (_i, [_vp, _d, _d]),
    "gfship_tree_set_next_event": (_i, [_vp, _vp, _vp]),
    "gfship_tree_time": (_d, [_vp]),
    "gfship_tree_dt": (_d, [_vp]),
    "gfship_tree_iter": (_u, [_vp]),
    "gfship_tree_add_tracer": (_i, [_vp, _i]),
    "gfship_tree_set_bc_u": (_i, [_vp, _i, _i, _i]),
    "gfship_tree_set_viscosity": (_i, [_vp, _i, _d]),
    "gfship_tree_set_source": (_i, [_vp, _i, _d]),
    "gfship_tree_diffusion_params": (C.POINTER(MultilevelParams), [_vp, _i]),
    "gfship_tree_start": (_i, [_vp]),
    "gfship_tree_step": (_i, [_vp]),
    "gfship_tree_sweep_levels": (_i, [_vp, _i, _pi, _pi]),
    "gfship_tree_divergence": (_i, [_vp]),
    "gfship_tree_host_check": (_i, [_i, C.c_void_p, _vp, _pi, _u, C.POINTER(C.c_longlong)]),
}


UNIQUE_ID_BYTES = 128


def comm_available():
    """RCCL can be opened on this rank (gfship_comm_available); raises otherwise"""
    _check(lib().gfship_comm_available())


def comm_unique_id():
    """ncclGetUniqueId through the library (gfship_comm_unique_id): bytes for Domain.comm_init"""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib().gfship_comm_unique_id(buf))
    return buf.raw


def lib():
    """Load libgfship.so; raises if it was not built (no fallback of any kind)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GfshipError("%s not found: run __graft_entry__.build() "
                              "(gerris-fft-particles_amd/csrc/build.sh)" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64
        # and loads them by unversioned names, so if libgfship pulled in /opt/rocm's copy first a
        # later `import torch` would start a second HSA runtime that finds no GPU.  Loading torch
        # first makes libgfship's NEEDED libamdhip64.so.7 resolve to the already loaded runtime.
        # (A C host that never loads torch uses /opt/rocm's runtime directly.)
        if "torch" not in sys.modules and os.environ.get("GFSHIP_NO_TORCH") is None:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)   # AttributeError if the library lacks a declared symbol
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise GfshipError("gfship error %d: %s" % (rc, lib().gfship_last_error().decode()))
    return rc


class Variable:
    """A GfsVariable living on the device (all levels)."""

    def __init__(self, dom, component=-1):
        self.dom = dom
        self.h = _check(lib().gfship_field_alloc(dom.ptr, component))

    def _shape(self, level):
        return ((1 << level) + 2,) * self.dom.dim

    def upload(self, a, level=None):
        level = self.dom.depth if level is None else level
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == self._shape(level), (a.shape, self._shape(level))
        _check(lib().gfship_field_upload(self.dom.ptr, self.h, level, a.ctypes.data_as(_pd)))

    def download(self, level=None):
        level = self.dom.depth if level is None else level
        a = np.empty(self._shape(level), dtype=np.float64)
        _check(lib().gfship_field_download(self.dom.ptr, self.h, level, a.ctypes.data_as(_pd)))
        return a

    def fill(self, value, level=None):
        level = self.dom.depth if level is None else level
        _check(lib().gfship_field_fill(self.dom.ptr, self.h, level, value))

    def set_bc(self, d, kind, val=None):
        if val is not None:
            val = np.ascontiguousarray(val, dtype=np.float64).ravel()
            p = val.ctypes.data_as(_pd)
        else:
            p = None
        _check(lib().gfship_field_set_bc(self.dom.ptr, self.h, d, kind, p))

    def free(self):
        if self.h is not None and self.dom.ptr:
            lib().gfship_field_free(self.dom.ptr, self.h)
            self.h = None


class Domain:
    """One uniform GfsBox on one GPU."""

    def __init__(self, dim, depth, side=None, device=0):
        self.dim, self.depth = dim, depth
        s = (C.c_int * 6)(*(side if side is not None else [SIDE_BOUNDARY] * 6))
        p = _vp()
        _check(lib().gfship_domain_create(C.byref(p), dim, depth, s, device))
        self.ptr = p

    def variable(self, component=-1):
        return Variable(self, component)

    def params(self):
        par = MultilevelParams()
        lib().gfship_multilevel_params_init(C.byref(par), self.dim)
        return par

    def comm_init(self, unique_id, rank, nranks, lattice):
        """in-library RCCL transport: this box is rank `rank` of a periodic lattice of boxes, one
        per GPU (gfship_domain_comm_init); unique_id = comm_unique_id() of one rank"""
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _check(lib().gfship_domain_comm_init(self.ptr, buf, rank, nranks, (C.c_int * 3)(*lattice)))

    def comm_size(self):
        return _check(lib().gfship_domain_comm_size(self.ptr))

    def comm_stats(self):
        m, b = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().gfship_domain_comm_stats(self.ptr, C.byref(m), C.byref(b)))
        return int(m.value), int(b.value)

    def path_counts(self):
        """(coarse ends of V-cycles computed for the whole lattice, tiled Godunov launches with MPI sides)"""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().gfship_domain_path_counts(self.ptr, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def snapshot_tree(self, variables):
        """the binary cell data of a GfsBox (gfship_snapshot_tree_write) as bytes"""
        n = len(variables)
        h = (C.c_int * n)(*[v.h for v in variables])
        size = lib().gfship_snapshot_tree_bytes(self.ptr, n)
        buf = C.create_string_buffer(size)
        _check(lib().gfship_snapshot_tree_write(self.ptr, n, h, buf, size))
        return buf.raw

    def snapshot_tree_read(self, variables, data):
        n = len(variables)
        h = (C.c_int * n)(*[v.h for v in variables])
        buf = C.create_string_buffer(bytes(data), len(data))
        _check(lib().gfship_snapshot_tree_read(self.ptr, n, h, buf, len(data)))

    def set_overlap(self, overlap):
        _check(lib().gfship_domain_set_overlap(self.ptr, int(overlap)))

    def set_relax_mode(self, mode):
        _check(lib().gfship_domain_set_relax_mode(self.ptr, mode))

    def synchronize(self):
        _check(lib().gfship_domain_synchronize(self.ptr))

    def bc(self, v, v1=None, level=None):
        level = self.depth if level is None else level
        _check(lib().gfship_bc(self.ptr, v.h, (v1 or v).h, level))

    def homogeneous_bc(self, ov, v, level=None):
        level = self.depth if level is None else level
        _check(lib().gfship_homogeneous_bc(self.ptr, ov.h, v.h, level))

    def poisson_coefficients(self):
        _check(lib().gfship_poisson_coefficients(self.ptr))

    def poisson_coefficients_alpha(self, alpha):
        """alpha: dim Variables holding the face values of the GfsFunction alpha"""
        h = (C.c_int * 3)(*([v.h for v in alpha] + [-1] * (3 - len(alpha))))
        _check(lib().gfship_poisson_coefficients_alpha(self.ptr, h))

    def poisson_weight(self, d, level=None):
        """f[d].v of the cells of a level (host array with ghosts)"""
        w = C.c_int()
        _check(lib().gfship_poisson_weights(self.ptr, d, C.byref(w)))
        v = Variable.__new__(Variable)
        v.dom, v.h = self, w.value
        a = v.download(level)
        v.h = None
        return a

    def relax(self, u, rhs, dia, level=None, omega=1., d=None):
        level = self.depth if level is None else level
        _check(lib().gfship_relax(self.ptr, d or self.dim, level, omega, u.h, rhs.h, dia.h))

    def residual(self, u, rhs, dia, res, level=None, d=None):
        level = self.depth if level is None else level
        _check(lib().gfship_residual(self.ptr, d or self.dim, level, u.h, rhs.h, dia.h, res.h))

    def norm_residual(self, res, dt=1.):
        n = Norm()
        _check(lib().gfship_norm_residual(self.ptr, dt, res.h, C.byref(n)))
        return n

    def norm_variable(self, v):
        n = Norm()
        _check(lib().gfship_norm_variable(self.ptr, v.h, C.byref(n)))
        return n

    def poisson_cycle(self, par, u, rhs, dia, res):
        _check(lib().gfship_poisson_cycle(self.ptr, C.byref(par), u.h, rhs.h, dia.h, res.h))

    def poisson_solve(self, par, lhs, rhs, res, dia, dt=1.):
        _check(lib().gfship_poisson_solve(self.ptr, C.byref(par), lhs.h, rhs.h, res.h, dia.h, dt))

    def diffusion_coefficients(self, D, dt, rhoc, beta=1.):
        _check(lib().gfship_diffusion_coefficients(self.ptr, D, dt, rhoc.h, beta))

    def diffusion_rhs(self, v, rhs, rhoc, beta=1.):
        _check(lib().gfship_diffusion_rhs(self.ptr, v.h, rhs.h, rhoc.h, beta))

    def diffusion_residual(self, u, rhs, rhoc, res):
        _check(lib().gfship_diffusion_residual(self.ptr, u.h, rhs.h, rhoc.h, res.h))

    def diffusion_cycle(self, levelmin, nrelax, u, rhs, rhoc, res):
        _check(lib().gfship_diffusion_cycle(self.ptr, levelmin, self.depth, nrelax,
                                            u.h, rhs.h, rhoc.h, res.h))

    def diffusion(self, par, v, rhs, rhoc):
        _check(lib().gfship_diffusion(self.ptr, C.byref(par), v.h, rhs.h, rhoc.h))

    def time_relax_loop(self, u, rhs, dia, nrelax=4, level=None, reps=5):
        """(ms per relax loop of the sweep kernels alone, fused?)"""
        level = self.depth if level is None else level
        ms, fused = C.c_double(), C.c_int()
        _check(lib().gfship_time_relax_loop(self.ptr, level, u.h, rhs.h, dia.h, nrelax, reps,
                                            C.byref(ms), C.byref(fused)))
        return ms.value, bool(fused.value)

    def time_relax_loop_inclusive(self, u, rhs, dia, nrelax=4, level=None, reps=5):
        """(ms of the sweep kernels alone, fused?, ms of the whole loop as a V-cycle pays for it)"""
        level = self.depth if level is None else level
        ms, fused, incl = C.c_double(), C.c_int(), C.c_double()
        _check(lib().gfship_time_relax_loop_inclusive(self.ptr, level, u.h, rhs.h, dia.h, nrelax, reps,
                                                      C.byref(ms), C.byref(fused), C.byref(incl)))
        return ms.value, bool(fused.value), incl.value

    def energy_spectra(self, comps):
        """GfsOutputEnergySpectra of the variables comps (U, V[, W]): (k, Ek, Etot) as the reference
        prints them (modules/fft.c:1340-1348)"""
        nk = _check(lib().gfship_energy_spectra_bins(self.ptr))
        Ek = np.empty(nk)
        etot, dk = C.c_double(), C.c_double()
        h = (_i * len(comps))(*[c.h for c in comps])
        _check(lib().gfship_energy_spectra(self.ptr, len(comps), h, Ek.ctypes.data_as(_pd),
                                           C.byref(etot), C.byref(dk)))
        i = np.arange(1, nk)
        return dk.value * np.sqrt(i.astype(float)), Ek[1:].copy(), etot.value

    def output_spectra(self, v):
        """GfsOutputSpectra of variable v on the whole 3-D domain: (F, kstep) with F the complex
        array [ix][iy][iz <= N/2] of the r2c DFT of (v - <v>)/ntot (modules/fft.c:1101-1160)"""
        N = _check(lib().gfship_output_spectra_side(self.ptr))
        out = np.empty((N, N, N // 2 + 1), dtype=np.complex128)
        ks = C.c_double()
        _check(lib().gfship_output_spectra(self.ptr, v.h, out.ctypes.data_as(_pd), C.byref(ks)))
        return out, ks.value

    def output_spectra_plane(self, v, normal, pos):
        """GfsOutputSpectra of a plane of the 3-D box (realdim == 2): (F, kstep), F[ia][ib <= N/2] the 2-D r2c
        DFT of the cell values on the plane minus their mean over their number"""
        N = _check(lib().gfship_output_spectra_side(self.ptr))
        out = np.empty((N, N // 2 + 1), dtype=np.complex128)
        ks = C.c_double()
        _check(lib().gfship_output_spectra_plane(self.ptr, v.h, int(normal), float(pos),
                                                 out.ctypes.data_as(_pd), C.byref(ks)))
        return out, ks.value

    def turbulent_viscosity(self, u, Cs, out, model=1):
        """GfsVariableTurbulentViscosity (modules/turbulence.c:953-1105): model 1 Smagorinsky, 0 sigma"""
        h = (_i * 3)(*([c.h for c in u] + [u[-1].h] * (3 - len(u))))
        _check(lib().gfship_turbulent_viscosity(self.ptr, h, float(Cs), int(model), out.h))

    def init_spectra(self, par, fields):
        """GfsInitSpectra (modules/turbulence.c): par = dict with the keywords of the .gfs object
        (x0 y0 z0 L E alpha epsilon c1 c2 c3 ReL kmax seed level)"""
        p = InitSpectraParams()
        for k, v in par.items():
            setattr(p, k, v)
        h = (_i * 3)(*[f.h for f in fields])
        _check(lib().gfship_init_spectra(self.ptr, C.byref(p), h))

    def interpolate(self, v, points):
        """GfsOutputLocation sampling: (values, inside) of variable v at points (np x 3)"""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty(len(pts))
        inside = np.empty(len(pts), dtype=np.uint8)
        _check(lib().gfship_field_interpolate(self.ptr, v.h, len(pts), pts.ctypes.data_as(_pd),
                                              out.ctypes.data_as(_pd),
                                              inside.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return out, inside.astype(bool)

    def time_relax(self, u, rhs, dia, level=None, reps=10, d=None):
        level = self.depth if level is None else level
        ms = C.c_double()
        _check(lib().gfship_time_relax(self.ptr, d or self.dim, level, u.h, rhs.h, dia.h, reps,
                                       C.byref(ms)))
        return ms.value

    def destroy(self):
        if self.ptr:
            lib().gfship_domain_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class _SimVariable(Variable):
    """A variable owned by a Simulation (P, Pmac, U, ...)."""

    def __init__(self, dom, handle):
        self.dom, self.h = dom, handle

    def free(self):
        pass


class Simulation:
    """GfsSimulation on one box: the simulation_run loop (src/simulation.c:432-557)."""
    VAR_P, VAR_PMAC, VAR_U, VAR_G, VAR_GMAC, VAR_TRACER, VAR_UN = range(7)

    def __init__(self, dom):
        self.dom = dom
        p = _vp()
        _check(lib().gfship_sim_create(C.byref(p), dom.ptr))
        self.ptr = p
        self.end = 1.7976931348623157e308
        L = lib()
        dim = dom.dim
        self.p = self._var(self.VAR_P)
        self.pmac = self._var(self.VAR_PMAC)
        self.u = [self._var(self.VAR_U, c) for c in range(dim)]
        self.g = [self._var(self.VAR_G, c) for c in range(dim)]
        self.gmac = [self._var(self.VAR_GMAC, c) for c in range(dim)]
        self.projection_params = L.gfship_sim_projection_params(self.ptr).contents
        self.approx_projection_params = L.gfship_sim_approx_projection_params(self.ptr).contents
        self.advection_params = L.gfship_sim_advection_params(self.ptr).contents

    def _var(self, which, c=0):
        return _SimVariable(self.dom, _check(lib().gfship_sim_variable(self.ptr, which, c)))

    def set_source(self, c, g):
        """GfsSource {} U/V/W g: constant intensity"""
        _check(lib().gfship_sim_set_source(self.ptr, c, float(g)))

    def set_alpha(self, alpha):
        """GfsPhysicalParams { alpha }: dim Variables of face values (None: alpha = NULL)"""
        if alpha is None:
            _check(lib().gfship_sim_set_alpha(self.ptr, None))
            return
        h = (C.c_int * 3)(*([v.h for v in alpha] + [-1] * (3 - len(alpha))))
        _check(lib().gfship_sim_set_alpha(self.ptr, h))

    def set_viscosity(self, c, nu):
        """SourceDiffusion {} U|V|W nu"""
        _check(lib().gfship_sim_set_viscosity(self.ptr, c, nu))

    def diffusion_params(self, c):
        return lib().gfship_sim_diffusion_params(self.ptr, c).contents

    def add_tracer(self, gradient=None):
        """GfsVariableTracer; gradient 0 = gfs_center_gradient, 1 = van Leer (the default)"""
        t = _check(lib().gfship_sim_add_tracer(self.ptr))
        if gradient is not None:
            _check(lib().gfship_sim_set_tracer_gradient(self.ptr, t, gradient))
        return self._var(self.VAR_TRACER, t)

    def mac_velocity(self, c):
        """MAC velocity of the + face of every cell along c (GFSHIP_VAR_UN)"""
        return self._var(self.VAR_UN, c)

    def advection_step(self):
        """loop body of advection_run (GfsAdvection) with the MAC velocities as they are"""
        _check(lib().gfship_sim_advection_step(self.ptr))

    def set_time(self, end=1.7976931348623157e308, dtmax=1.7976931348623157e308):
        self.end = end
        _check(lib().gfship_sim_set_time(self.ptr, end, dtmax))

    @property
    def t(self):
        return lib().gfship_sim_time(self.ptr)

    @property
    def i(self):
        return lib().gfship_sim_iter(self.ptr)

    @property
    def dt(self):
        return self.advection_params.dt

    def restart(self, t, i):
        _check(lib().gfship_sim_restart(self.ptr, t, i))

    def start(self):
        _check(lib().gfship_sim_start(self.ptr))

    def step(self):
        _check(lib().gfship_sim_step(self.ptr))

    def predicted_face_velocities(self):
        _check(lib().gfship_predicted_face_velocities(self.ptr))

    def cfl(self):
        v = C.c_double()
        _check(lib().gfship_domain_cfl(self.ptr, C.byref(v)))
        return v.value

    def divergence_norm(self):
        n = Norm()
        _check(lib().gfship_divergence_norm(self.ptr, C.byref(n)))
        return n

    def tracer_advection(self, t, dt):
        _check(lib().gfship_tracer_advection(self.ptr, t.h, dt))

    def un(self, c):
        n = (1 << self.dom.depth) + 2
        a = np.empty((n,) * self.dom.dim)
        _check(lib().gfship_sim_download_un(self.ptr, c, a.ctypes.data_as(_pd)))
        return a

    def destroy(self):
        if self.ptr:
            lib().gfship_sim_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            if self.dom.ptr:
                self.destroy()
        except Exception:
            pass


FORCE_INERTIAL, FORCE_ADDEDMASS, FORCE_LIFT, FORCE_DRAG, FORCE_BUOY = 1, 2, 3, 4, 5


class InitSpectraParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("x0", "y0", "z0", "L", "E", "alpha", "epsilon", "c1", "c2",
                                           "c3", "ReL", "kmax", "seed")] + [("level", C.c_int)]


class ParticleList:
    """GfsParticleList of GfsParticle tracers on the device."""

    def __init__(self, sim, pos, ids):
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert len(pos) == len(ids)
        self.sim, self.n0 = sim, len(ids)
        p = _vp()
        _check(lib().gfship_particles_create(C.byref(p), sim.ptr, len(ids), pos.ctypes.data_as(_pd),
                                             ids.ctypes.data_as(C.POINTER(C.c_uint))))
        self.ptr = p

    def event(self):
        _check(lib().gfship_particle_list_event(self.ptr))

    def sort(self):
        _check(lib().gfship_particles_sort(self.ptr))

    def set_sort_interval(self, every):
        _check(lib().gfship_particles_set_sort_interval(self.ptr, every))

    def count(self):
        return _check(lib().gfship_particles_count(self.ptr))

    def download(self):
        m = max(_check(lib().gfship_particles_slots(self.ptr)), 1)
        pos = np.empty((m, 3))
        ids = np.empty(m, dtype=np.uint32)
        k = _check(lib().gfship_particles_download(self.ptr, pos.ctypes.data_as(_pd),
                                                   ids.ctypes.data_as(C.POINTER(C.c_uint))))
        return pos[:k].copy(), ids[:k].copy()

    # GfsParticulate: velocity, mass, volume and the list's forces (FORCE_* in application order)
    def set_particulate(self, vel, mass, volume):
        vel = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
        mass = np.ascontiguousarray(mass, dtype=np.float64)
        volume = np.ascontiguousarray(volume, dtype=np.float64)
        _check(lib().gfship_particles_set_particulate(self.ptr, vel.ctypes.data_as(_pd),
                                                      mass.ctypes.data_as(_pd),
                                                      volume.ctypes.data_as(_pd)))

    def set_forces(self, kinds, gravity=(0., 0., 0.)):
        k = (_i * len(kinds))(*kinds)
        g = (C.c_double * 3)(*gravity)
        _check(lib().gfship_particles_set_forces(self.ptr, len(kinds), k, g))

    def set_force_coefficient(self, force, function):
        """the GfsFunction (C text of Rep, Urelp, Vrelp, Wrelp, Pdia, t) of force number `force' of the
        list: compiled for the device with hipRTC"""
        _check(lib().gfship_particles_set_force_coefficient(self.ptr, int(force), function.encode()))

    def particulate_state(self):
        """(vel, mass, force) of the particles on the list, in list order"""
        m = max(_check(lib().gfship_particles_slots(self.ptr)), 1)
        vel, mass, force = np.empty((m, 3)), np.empty(m), np.empty((m, 3))
        k = _check(lib().gfship_particles_download_particulate(
            self.ptr, vel.ctypes.data_as(_pd), mass.ctypes.data_as(_pd), force.ctypes.data_as(_pd)))
        return vel[:k].copy(), mass[:k].copy(), force[:k].copy()

    def destroy(self):
        if self.ptr:
            lib().gfship_particles_destroy(self.ptr)
            self.ptr = None


REFINE_FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p)


def tree_host_check(refine, dim=2, sides=None, nrelax=4):
    """gfship_tree_host_check: the plans of a tree validated on the host (no device needed);
    returns (cell updates, levels sweep after sweep, levels of the loop plans, differing values)"""
    if dim == 2:
        cb = REFINE_FN(lambda x, y, z, ctx: float(refine(x, y)))
    else:
        cb = REFINE_FN(lambda x, y, z, ctx: float(refine(x, y, z)))
    arr = (C.c_int * 6)(*(list(sides) + [0] * 6)[:6]) if sides is not None else None
    stats = (C.c_longlong * 4)()
    _check(lib().gfship_tree_host_check(dim, C.cast(cb, C.c_void_p), None, arr, nrelax, stats))
    return tuple(stats)


class Tree:
    """gfship_tree: a GfsSimulation on one periodic box refined by a GfsRefine function (coarse-fine
    stencils; quadtree or octree).  refine (x, y) or refine (x, y, z) -> level wanted there."""
    P, PMAC, U, V, GX, GY, GMACX, GMACY, UN0, UN1, UN2, UN3, W, GZ, GMACZ, UN4, UN5, DIV, BCVAL, RES, T0, T1, BCU, BCV, BCW = range(25)

    def __init__(self, refine, dim=2, device=0, sides=None):
        self.dim = dim
        if dim == 2:
            self._cb = REFINE_FN(lambda x, y, z, ctx: float(refine(x, y)))
        else:
            self._cb = REFINE_FN(lambda x, y, z, ctx: float(refine(x, y, z)))
        p = _vp()
        if sides is None:
            _check(lib().gfship_tree_create(C.byref(p), dim, C.cast(self._cb, C.c_void_p), None, device))
        else:
            arr = (C.c_int * 6)(*(list(sides) + [0] * 6)[:6])
            _check(lib().gfship_tree_create_sides(C.byref(p), dim, C.cast(self._cb, C.c_void_p), None, arr,
                                                  device))
        self.ptr = p
        self.depth = lib().gfship_tree_depth(p)
        self.projection_params = lib().gfship_tree_projection_params(p, 0).contents
        self.approx_projection_params = lib().gfship_tree_projection_params(p, 1).contents

    def flags(self, level):
        r = (1 << level) + 2
        a = np.zeros((r,) * self.dim, dtype=np.uint8)
        _check(lib().gfship_tree_flags(self.ptr, level, a.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return a

    def centres(self, level):
        n = 1 << level
        c = -0.5 + (np.arange(n + 2) - 0.5) / n
        if self.dim == 2:
            return np.meshgrid(c, c, indexing="xy")
        z, y, x = np.meshgrid(c, c, c, indexing="ij")
        return x, y, z

    def upload(self, var, level, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == ((1 << level) + 2,) * self.dim
        _check(lib().gfship_tree_upload(self.ptr, var, level, a.ctypes.data_as(_pd)))

    def download(self, var, level):
        r = (1 << level) + 2
        a = np.empty((r,) * self.dim)
        _check(lib().gfship_tree_download(self.ptr, var, level, a.ctypes.data_as(_pd)))
        return a

    def set_time(self, end, cfl):
        _check(lib().gfship_tree_set_time(self.ptr, end, cfl))

    def set_bc_u(self, c, d, kind, values=None):
        """condition of velocity component c on side d; values: per level, the arrays of the values at the
        ghost cells (the oracle's gt_bc_values_u layout), uploaded to BCU + c"""
        _check(lib().gfship_tree_set_bc_u(self.ptr, c, d, kind))
        if values is not None:
            for l, a in enumerate(values):
                self.upload(Tree.BCU + c, l, a)

    def set_viscosity(self, c, nu):
        _check(lib().gfship_tree_set_viscosity(self.ptr, c, nu))

    def set_source(self, c, g):
        _check(lib().gfship_tree_set_source(self.ptr, c, g))

    def diffusion_params(self, c):
        return lib().gfship_tree_diffusion_params(self.ptr, c).contents

    def add_tracer(self, gradient=1):
        """GfsVariableTracer [{ gradient = }] (0 centred, 1 van Leer): the variable index (T0, T1)"""
        return _check(lib().gfship_tree_add_tracer(self.ptr, gradient))

    def start(self):
        _check(lib().gfship_tree_start(self.ptr))

    def step(self):
        _check(lib().gfship_tree_step(self.ptr))

    def set_bc(self, d, kind):
        _check(lib().gfship_tree_set_bc(self.ptr, d, kind))

    def poisson_solve(self, par, dt=1.):
        """gfs_poisson_solve: guess in P, right-hand side in DIV, residual left in RES"""
        _check(lib().gfship_tree_poisson_solve(self.ptr, C.byref(par), dt))

    def divergence(self, level):
        """the derived variable Divergence of the leaves of a level (gfs_divergence)"""
        _check(lib().gfship_tree_divergence(self.ptr))
        return self.download(self.DIV, level)

    def sweep_levels(self, level):
        a, b = C.c_int(), C.c_int()
        _check(lib().gfship_tree_sweep_levels(self.ptr, level, C.byref(a), C.byref(b)))
        return a.value, b.value

    t = property(lambda self: lib().gfship_tree_time(self.ptr))
    dt = property(lambda self: lib().gfship_tree_dt(self.ptr))
    i = property(lambda self: lib().gfship_tree_iter(self.ptr))

    def destroy(self):
        if self.ptr:
            lib().gfship_tree_destroy(self.ptr)
            self.ptr = None
