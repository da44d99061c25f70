// domain.hip -- gfship_domain: per-level SoA device storage, variables, transfers.
// Replaces the pointer-linked FttCell/FttOct tree (src/ftt.h:134-159) and the per-cell
// g_malloc0'd GfsStateVector (src/fluid.h:39-52, src/domain.c:2932-2954).
#include "gfship_internal.hpp"

namespace gfship {

static thread_local char g_error[1024] = "";

void set_error (const char * fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (g_error, sizeof (g_error), fmt, ap);
  va_end (ap);
}

int hip_fail (hipError_t e, const char * what, const char * file, int line)
{
  set_error ("HIP error %d (%s) in %s at %s:%d", (int) e, hipGetErrorString (e), what, file, line);
  return e == hipErrorOutOfMemory ? GFSHIP_ENOMEM : GFSHIP_EHIP;
}

Field * get_field (gfship_domain * dom, gfship_field f)
{
  if (!dom || f < 0 || (size_t) f >= dom->fields.size () || !dom->fields[f].used) {
    set_error ("invalid field handle %d", f);
    return nullptr;
  }
  return &dom->fields[f];
}

} // namespace gfship

using namespace gfship;

extern "C" {

const char * gfship_last_error (void) { return g_error; }
int gfship_version (void) { return GFSHIP_VERSION; }

int gfship_device_count (void)
{
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess)
    return 0;
  return n;
}

int gfship_domain_create (gfship_domain ** out, int dim, int depth, const int side[6], int device)
{
  GFSHIP_CHECK (out != nullptr, GFSHIP_EINVAL, "null output pointer");
  *out = nullptr;
  GFSHIP_CHECK (dim == 2 || dim == 3, GFSHIP_EINVAL, "dim must be 2 or 3 (got %d)", dim);
  GFSHIP_CHECK (depth >= 0 && depth <= GFSHIP_MAXLEVEL, GFSHIP_EINVAL,
		"depth must be in [0,%d] (got %d)", GFSHIP_MAXLEVEL, depth);
  int ndev = gfship_device_count ();
  GFSHIP_CHECK (ndev > 0, GFSHIP_ENODEVICE,
		"no HIP device: libgfship has no CPU fallback");
  GFSHIP_CHECK (device >= 0 && device < ndev, GFSHIP_EINVAL, "device %d out of range (%d devices)",
		device, ndev);
  GFSHIP_HIP (hipSetDevice (device));
  gfship_domain * dom = new gfship_domain;
  dom->dim = dim;
  dom->depth = depth;
  dom->device = device;
  for (int d = 0; d < 6; d++) {
    dom->side[d] = side ? side[d] : GFSHIP_SIDE_BOUNDARY;
    if (d < 2*dim && (dom->side[d] < 0 || dom->side[d] > GFSHIP_SIDE_EXTERNAL)) {
      delete dom;
      set_error ("invalid side kind %d for direction %d", side[d], d);
      return GFSHIP_EINVAL;
    }
  }
  for (int d = 0; d < 2*dim; d += 2)
    if ((dom->side[d] == GFSHIP_SIDE_PERIODIC) != (dom->side[d + 1] == GFSHIP_SIDE_PERIODIC)) {
      delete dom;
      set_error ("periodic sides must come in pairs (directions %d,%d)", d, d + 1);
      return GFSHIP_EINVAL;
    }
  for (int d = 0; d < 2*dim; d++)
    if (dom->side[d] == GFSHIP_SIDE_EXTERNAL)
      dom->has_external = true;
  for (int l = 0; l <= depth; l++) {
    Layout & L = dom->lay[l];
    L.n = 1 << l;
    L.dim = dim;
    L.rows = L.n + 2;
    L.xo = 1;
    L.px = L.n + 4;
    L.sy = L.px;
    L.sz = dim == 3 ? (long) L.px*L.rows : 0;
    L.total = (size_t) L.px*L.rows*(dim == 3 ? L.rows : 1);
  }
  hipError_t e = hipStreamCreateWithFlags (&dom->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate (&dom->ev0);
  if (e == hipSuccess) e = hipEventCreate (&dom->ev1);
  if (e == hipSuccess) e = hipStreamCreateWithFlags (&dom->side_stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags (&dom->side_fork, hipEventDisableTiming);
  dom->scratch_doubles = 5*8192 + 64;
  if (e == hipSuccess) e = hipMalloc ((void **) &dom->d_scratch, dom->scratch_doubles*sizeof (double));
  if (e == hipSuccess) e = hipHostMalloc ((void **) &dom->h_pinned, 64*sizeof (double), hipHostMallocDefault);
  if (e == hipSuccess) memset (dom->h_pinned, 0, 64*sizeof (double));
  if (e != hipSuccess) {
    int r = hip_fail (e, "domain resources", __FILE__, __LINE__);
    gfship_domain_destroy (dom);
    return r;
  }
  dom->skew_old = getenv ("GFSHIP_SKEW_OLD") != nullptr;
  dom->patch = getenv ("GFSHIP_SKEW_LINES") == nullptr && !dom->skew_old;
  { const char * w = getenv ("GFSHIP_PATCH_MIN_N"); if (w) dom->patch_min_n = atoi (w); }
  dom->no_fused_godunov3 = getenv ("GFSHIP_NO_ADVECT3") != nullptr;
  dom->no_lattice_cycle = getenv ("GFSHIP_NO_LATTICE_CYCLE") != nullptr;
  dom->no_fused_mpi = getenv ("GFSHIP_NO_FUSED_MPI") != nullptr;
  dom->no_fused_prolongation = getenv ("GFSHIP_NO_FUSED_PROLONGATION") != nullptr;
  dom->no_kernel_arming = getenv ("GFSHIP_KERNEL_ARMING") == nullptr;
  dom->no_xcd_scope = getenv ("GFSHIP_XCD_SCOPE") == nullptr;
  dom->no_fused_restriction = getenv ("GFSHIP_NO_FUSED_RESTRICTION") != nullptr;
  dom->no_arm_ahead = getenv ("GFSHIP_NO_ARM_AHEAD") != nullptr;
  { const char * w = getenv ("GFSHIP_XCD_PLACE"); dom->xcd_place = w && w[0] == '1'; }
  { const char * w = getenv ("GFSHIP_WAVE_LOOP"); dom->wave_loop = w && w[0] == '1'; }
  *out = dom;
  return GFSHIP_OK;
}

void gfship_domain_destroy (gfship_domain * dom)
{
  if (!dom) return;
  (void) hipSetDevice (dom->device);
  if (dom->stream) (void) hipStreamSynchronize (dom->stream);
  if (dom->side_stream) (void) hipStreamSynchronize (dom->side_stream);
  for (size_t f = 0; f < dom->fields.size (); f++)
    if (dom->fields[f].used)
      gfship_field_free (dom, (gfship_field) f);
  comm_free (dom);
  mpi_order_free (dom);
  skew_free (dom);
  if (dom->d_scratch) (void) hipFree (dom->d_scratch);
  if (dom->cfl_partial) (void) hipFree (dom->cfl_partial);
  for (int d = 0; d < 6; d++) {
    if (dom->gfv_send[d]) (void) hipFree (dom->gfv_send[d]);
    if (dom->gfv_recv[d]) (void) hipFree (dom->gfv_recv[d]);
  }
  if (dom->lat_res) (void) hipFree (dom->lat_res);
  if (dom->lat_xch) (void) hipFree (dom->lat_xch);
  if (dom->lat_bar) (void) hipFree (dom->lat_bar);
  if (dom->h_pinned) (void) hipHostFree (dom->h_pinned);
  if (dom->ev0) (void) hipEventDestroy (dom->ev0);
  if (dom->ev1) (void) hipEventDestroy (dom->ev1);
  if (dom->side_fork) (void) hipEventDestroy (dom->side_fork);
  if (dom->side_stream) (void) hipStreamDestroy (dom->side_stream);
  if (dom->stream) (void) hipStreamDestroy (dom->stream);
  delete dom;
}

int gfship_domain_set_relax_mode (gfship_domain * dom, int mode)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  GFSHIP_CHECK (mode == GFSHIP_RELAX_EXACT || mode == GFSHIP_RELAX_REDBLACK ||
		mode == GFSHIP_RELAX_EXACT_HYPERPLANE || mode == GFSHIP_RELAX_EXACT_PER_SWEEP,
		GFSHIP_EINVAL, "unknown relax mode %d", mode);
  dom->force_hyperplane = (mode == GFSHIP_RELAX_EXACT_HYPERPLANE);
  dom->no_fused_loop = (mode == GFSHIP_RELAX_EXACT_PER_SWEEP);
  dom->no_fused_godunov = (mode == GFSHIP_RELAX_EXACT_PER_SWEEP);   /* the unfused reference paths */
  dom->relax_mode = mode == GFSHIP_RELAX_REDBLACK ? GFSHIP_RELAX_REDBLACK : GFSHIP_RELAX_EXACT;
  return GFSHIP_OK;
}

int gfship_domain_set_overlap (gfship_domain * dom, int overlap)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  dom->overlap = overlap != 0;
  return GFSHIP_OK;
}

int gfship_domain_set_exchange (gfship_domain * dom, gfship_exchange_fn fn, void * ctx)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  dom->exchange = fn;
  dom->exchange_ctx = ctx;
  return GFSHIP_OK;
}

int gfship_domain_set_reduce (gfship_domain * dom, gfship_reduce_fn fn, void * ctx)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  dom->reduce = fn;
  dom->reduce_ctx = ctx;
  return GFSHIP_OK;
}

int gfship_domain_set_gather (gfship_domain * dom, gfship_gather_fn fn, void * ctx, int rank,
			      int nboxes, const int lattice[3])
{
  GFSHIP_CHECK (dom && lattice, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (nboxes >= 1 && rank >= 0 && rank < nboxes &&
		(long) lattice[0]*lattice[1]*lattice[2] == nboxes, GFSHIP_EINVAL,
		"box %d of %d on a lattice of %d x %d x %d", rank, nboxes, lattice[0], lattice[1], lattice[2]);
  for (int d = 0; d < 2*dom->dim; d++)
    GFSHIP_CHECK (lattice[d/2] == 1 || dom->side[d] == GFSHIP_SIDE_EXTERNAL, GFSHIP_EINVAL,
		  "side %d faces another box of the lattice: it must be GFSHIP_SIDE_EXTERNAL", d);
  dom->gather = fn;
  dom->gather_ctx = ctx;
  dom->lat_rank = rank; dom->lat_n = nboxes;
  for (int c = 0; c < 3; c++) dom->lat_b[c] = lattice[c];
  return GFSHIP_OK;
}

int gfship_domain_path_counts (gfship_domain * dom, unsigned long long * lattice_cycles,
			       unsigned long long * fused_mpi_launches)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  if (lattice_cycles) *lattice_cycles = dom->n_lattice_cycles;
  if (fused_mpi_launches) *fused_mpi_launches = dom->n_fused_mpi;
  return GFSHIP_OK;
}

int gfship_domain_synchronize (gfship_domain * dom)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  return GFSHIP_OK;
}

void * gfship_domain_stream (gfship_domain * dom)
{
  return dom ? (void *) dom->stream : nullptr;
}

gfship_field gfship_field_alloc (gfship_domain * dom, int component)
{
  GFSHIP_CHECK (dom != nullptr, GFSHIP_EINVAL, "null domain");
  GFSHIP_CHECK (component >= -1 && component < dom->dim, GFSHIP_EINVAL,
		"component %d out of range", component);
  size_t slot = dom->fields.size ();
  for (size_t f = 0; f < dom->fields.size (); f++)
    if (!dom->fields[f].used) { slot = f; break; }
  if (slot == dom->fields.size ())
    dom->fields.emplace_back ();
  Field nf;
  nf.used = true;
  nf.component = component;
  // one allocation for all levels (coarse levels add 1/7 in 3-D), zero-filled like g_malloc0
  size_t total = 0;
  for (int l = 0; l <= dom->depth; l++)
    total += (dom->lay[l].total + 31) & ~(size_t) 31;
  double * base = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &base, total*sizeof (double)));
  GFSHIP_HIP (hipMemsetAsync (base, 0, total*sizeof (double), dom->stream));
  size_t o = 0;
  for (int l = dom->depth; l >= 0; l--) { // finest level first: it gets the aligned base
    nf.lev[l] = base + o;
    nf.zero[l] = true;
    o += (dom->lay[l].total + 31) & ~(size_t) 31;
  }
  dom->fields[slot] = nf;
  return (gfship_field) slot;
}

int gfship_field_free (gfship_domain * dom, gfship_field f)
{
  Field * F = get_field (dom, f);
  if (!F) return GFSHIP_EINVAL;
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  GFSHIP_HIP (hipFree (F->lev[dom->depth]));
  for (int d = 0; d < 6; d++)
    if (F->bcval[d]) GFSHIP_HIP (hipFree (F->bcval[d]));
  *F = Field ();
  return GFSHIP_OK;
}

int gfship_field_set_bc (gfship_domain * dom, gfship_field f, int d, int type, const double * val)
{
  Field * F = get_field (dom, f);
  if (!F) return GFSHIP_EINVAL;
  GFSHIP_CHECK (d >= 0 && d < 2*dom->dim, GFSHIP_EINVAL, "direction %d out of range", d);
  GFSHIP_CHECK (type >= GFSHIP_BC_SYMMETRY && type <= GFSHIP_BC_NEUMANN, GFSHIP_EINVAL,
		"unknown bc type %d", type);
  F->bc[d] = type;
  if (F->bcval[d]) {
    GFSHIP_HIP (hipStreamSynchronize (dom->stream));
    GFSHIP_HIP (hipFree (F->bcval[d]));
    F->bcval[d] = nullptr;
  }
  if (val) {
    int n = dom->lay[dom->depth].n;
    size_t m = dom->dim == 3 ? (size_t) n*n : (size_t) n;
    GFSHIP_HIP (hipMalloc ((void **) &F->bcval[d], m*sizeof (double)));
    GFSHIP_HIP (hipMemcpy (F->bcval[d], val, m*sizeof (double), hipMemcpyHostToDevice));
  }
  return GFSHIP_OK;
}

static int check_level (gfship_domain * dom, int level)
{
  GFSHIP_CHECK (level >= 0 && level <= dom->depth, GFSHIP_EINVAL, "level %d out of [0,%d]", level,
		dom->depth);
  return GFSHIP_OK;
}

} // extern "C"

namespace gfship {
// bring the non-leaf levels of F up to date before level `level` (< depth) is read or written
int coarse_flush (gfship_domain * dom, Field * F, int level)
{
  if (!F->coarse_stale || level >= dom->depth) return GFSHIP_OK;
  F->coarse_stale = false;
  Field * one[1] = { F };
  return launch_coarse_init (dom, one, 1);
}
}

extern "C" {

int gfship_field_upload (gfship_domain * dom, gfship_field f, int level, const double * host)
{
  Field * F = get_field (dom, f);
  if (!F) return GFSHIP_EINVAL;
  GFSHIP_CHECK (host != nullptr, GFSHIP_EINVAL, "null host pointer");
  int r = check_level (dom, level);
  if (r) return r;
  if (dom->before_write && (r = dom->before_write (dom->before_write_ctx))) return r;
  if ((r = coarse_flush (dom, F, level))) return r;
  const Layout & L = dom->lay[level];
  size_t rows = (size_t) L.rows*(dom->dim == 3 ? L.rows : 1);
  GFSHIP_HIP (hipMemcpy2DAsync (F->lev[level] + L.xo, L.px*sizeof (double),
				host, L.rows*sizeof (double),
				L.rows*sizeof (double), rows, hipMemcpyHostToDevice, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  F->zero[level] = false;
  if (level < dom->depth) F->coarse_valid = true;
  return GFSHIP_OK;
}

int gfship_field_download (gfship_domain * dom, gfship_field f, int level, double * host)
{
  Field * F = get_field (dom, f);
  if (!F) return GFSHIP_EINVAL;
  GFSHIP_CHECK (host != nullptr, GFSHIP_EINVAL, "null host pointer");
  int r = check_level (dom, level);
  if (r) return r;
  if ((r = coarse_flush (dom, F, level))) return r;
  const Layout & L = dom->lay[level];
  size_t rows = (size_t) L.rows*(dom->dim == 3 ? L.rows : 1);
  GFSHIP_HIP (hipMemcpy2DAsync (host, L.rows*sizeof (double),
				F->lev[level] + L.xo, L.px*sizeof (double),
				L.rows*sizeof (double), rows, hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  return GFSHIP_OK;
}

int gfship_field_fill (gfship_domain * dom, gfship_field f, int level, double value)
{
  Field * F = get_field (dom, f);
  if (!F) return GFSHIP_EINVAL;
  int r = check_level (dom, level);
  if (r) return r;
  if ((r = coarse_flush (dom, F, level))) return r;
  r = launch_fill (dom, level, F->lev[level], value);
  F->zero[level] = (r == GFSHIP_OK && value == 0.);
  return r;
}

void * gfship_field_device_ptr (gfship_domain * dom, gfship_field f, int level, int * px, int * xo)
{
  Field * F = get_field (dom, f);
  if (!F || level < 0 || level > dom->depth) return nullptr;
  if (px) *px = dom->lay[level].px;
  if (xo) *xo = dom->lay[level].xo;
  if (coarse_flush (dom, F, level) != GFSHIP_OK) return nullptr;
  F->zero[level] = false; /* the caller may write through the raw pointer */
  return F->lev[level];
}

int gfship_bc (gfship_domain * dom, gfship_field v, gfship_field v1, int level)
{
  Field * V = get_field (dom, v), * V1 = get_field (dom, v1);
  if (!V || !V1) return GFSHIP_EINVAL;
  int r = check_level (dom, level);
  if (r) return r;
  if ((r = coarse_flush (dom, V1, level))) return r;
  V1->zero[level] = false;
  return launch_bc (dom, V, V1, level, 0);
}

int gfship_homogeneous_bc (gfship_domain * dom, gfship_field ov, gfship_field v, int level)
{
  Field * OV = get_field (dom, ov), * V = get_field (dom, v);
  if (!OV || !V) return GFSHIP_EINVAL;
  int r = check_level (dom, level);
  if (r) return r;
  if ((r = coarse_flush (dom, OV, level))) return r;
  OV->zero[level] = false;
  return launch_bc (dom, V, OV, level, 1);
}

} // extern "C"
