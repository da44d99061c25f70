// particles.hip -- Lagrangian tracers on the device: GfsParticle in a GfsParticleList
// (src/particle.c:31-44, modules/particulatecommon.c:955-1015,3058-3395).
//
// One thread per particle, SoA storage (x, y, z, old position, id, alive).  Each step:
//   remove_particles_not_in_domain -> gfs_particle_event (RK2 midpoint through
//   gfs_domain_advect_point, src/domain.c:2764-2788) -> gfs_particle_bc (ray march to the exit
//   face, periodic wrap).
// locate = ftt_cell_locate's descent with its strict '>' comparisons (src/ftt.c:1535-1574);
// interpolate = gfs_interpolate: corner values are inverse-distance weighted means of the cells
// sharing the corner (src/fluid.c:2983-3101), then the reference's 8-corner polynomial
// (:2640-2683).  All in the reference's operand order: cell indices and positions are
// bit-identical to the CPU algorithm.  The reference has no test for this path (parity pinned
// on the oracle and on analytic properties only).
//
// Sort by cell: the gather of a particle reads the 27 cells around it for three components and
// two RK stages, so particles of one wavefront should sit in neighbouring cells.  Every
// `sort_every` events the slots are reordered by the linear index of the containing leaf cell
// (stable LSD radix sort of (cell, slot) pairs, rocPRIM through hipCUB), which changes where a
// particle is stored, never its arithmetic: `orig` remembers the slot a particle was created in
// and gfship_particles_download returns the survivors in that order (the reference's list
// order).  Tracers move less than one cell per step (CFL < 1), so the order stays good between
// sorts.
#include "gfship_internal.hpp"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <numeric>

namespace gfship {
struct RtcKernel;
int  rtc_compile_coefficient (gfship_domain * dom, const char * text, RtcKernel ** out);
int  rtc_launch_coefficient (RtcKernel * k, hipStream_t stream, int n, const unsigned char * alive,
			     const double * rep, const double * const rel[3], const double * pdia,
			     double t, double * out);
void rtc_free (RtcKernel * k);
}

struct gfship_particles {
  gfship_sim * sim = nullptr;
  gfship_domain * dom = nullptr;
  int n = 0;                   // slots in use (alive or not)
  int cap = 0;                 // slots allocated
  double * pos[3] = {}, * old[3] = {};
  unsigned * id = nullptr;
  unsigned char * alive = nullptr;
  unsigned * d_count = nullptr;
  // sort by cell
  unsigned * orig = nullptr;          // creation slot of the particle stored in each slot
  double * pos2[3] = {}, * old2[3] = {};   // gather targets (swapped with pos/old after a sort)
  unsigned * id2 = nullptr, * orig2 = nullptr;
  unsigned char * alive2 = nullptr;
  unsigned * key = nullptr, * key2 = nullptr, * slot = nullptr, * slot2 = nullptr;
  void * sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int sort_every = 16, events_since_sort = -1;   // -1: never sorted yet
  // migration through GfsBoundaryMpi sides
  gfship_particle_migrate_fn migrate = nullptr; void * migrate_ctx = nullptr;
  double * outbox = nullptr;          // device, 6 x out_cap records of 7 doubles
  unsigned * out_count = nullptr;     // device, 6 counters
  int out_cap = 0;
  // GfsParticulate (modules/particulatecommon.h:35-48), indexed by the creation slot (`orig`):
  // velocity, force, mass, volume, diameter; the list's forces in application order
  bool particulate = false;
  int np0 = 0;                        // particles at creation
  double * vel[3] = {}, * force[3] = {}, * mass = nullptr, * volume = nullptr, * dia = nullptr;
  int nforces = 0, forces[8] = {};
  double gravity[3] = {};
  gfship_field uold[3] = { -1, -1, -1 };   // Un, Vn, Wn of GfsForceCoeff
  // GfsFunction coefficients of the GfsForceCoeff objects, compiled for the device (rtc.hip): the
  // variables Rep, Urelp, Vrelp, Wrelp, Pdia of every particle (slot order), and the values
  gfship::RtcKernel * coef_fn[8] = {};
  double * coef[8] = {}, * cin[6] = {};
  int coef_cap = 0;
};

namespace gfship {

struct PartArgs {
  Layout L;
  int side[6];
  int n;
  double * pos[3];
  double * old[3];
  unsigned char * alive;
  const double * u[3];
  double dt;
  unsigned * count;
  int migrate;                 // particles leaving through an external side are kept for the hook
};

// ftt_cell_locate on the unit box centred on the origin, leaf level
template <int DIM>
__device__ __forceinline__ bool locate (int depth, const double target[3], int ijk[3])
{
  double pos[3] = { 0., 0., 0. };
  double size = 1./2.;
#pragma unroll
  for (int c = 0; c < DIM; c++)
    if (target[c] > pos[c] + size || target[c] < pos[c] - size)
      return false;
  int q[3] = { 0, 0, 0 };
  for (int l = 0; l < depth; l++) {
    size /= 2.;
#pragma unroll
    for (int c = 0; c < DIM; c++) {
      bool up = target[c] > pos[c];
      q[c] = 2*q[c] + (up ? 1 : 0);
      pos[c] += (up ? 1. : -1.)*size;
    }
  }
  ijk[0] = q[0] + 1; ijk[1] = q[1] + 1; ijk[2] = DIM == 3 ? q[2] + 1 : 0;
  return true;
}

// gfs_interpolate (src/fluid.c:2697-2710) of one variable at p inside cell (i,j,k)
template <int DIM>
__device__ double interpolate (const Layout & L, const double * __restrict__ v, const int cell[3],
			       const double p_[3])
{
  const int n = L.n;
  const double h = 1./n;
  constexpr int NC = 4*(DIM - 1);
  // the 3^DIM neighbourhood (edge/corner ghosts are allocated, never used)
  double nb[DIM == 3 ? 27 : 9];
  bool outm[3], outp[3];    // is cell - 1 / cell + 1 a ghost along each axis?
#pragma unroll
  for (int a = 0; a < DIM; a++) {
    outm[a] = cell[a] - 1 < 1;
    outp[a] = cell[a] + 1 > n;
  }
  const long base = L.idx (cell[0], cell[1], cell[2]);
#pragma unroll
  for (int dz = (DIM == 3 ? -1 : 0); dz <= (DIM == 3 ? 1 : 0); dz++)
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
      for (int dx = -1; dx <= 1; dx++)
	nb[(dx + 1) + 3*(dy + 1) + 9*(DIM == 3 ? dz + 1 : 0)] = v[base + dx + dy*L.sy + dz*L.sz];

  const double a_ = 1./(h*(DIM == 2 ? 0.707106781185 : 0.866025403785) + 1e-12);
  // corner directions, src/fluid.c:2588-2605
  const int c2[4][3] = { {-1,-1,0}, {1,-1,0}, {1,1,0}, {-1,1,0} };
  const int c3[8][3] = { {-1,-1,1}, {1,-1,1}, {1,1,1}, {-1,1,1},
			 {-1,-1,-1}, {1,-1,-1}, {1,1,-1}, {-1,1,-1} };
  double f[NC + 1];
#pragma unroll
  for (int ic = 0; ic < NC; ic++) {
    int sg[3];
#pragma unroll
    for (int a = 0; a < 3; a++) sg[a] = DIM == 2 ? c2[ic][a] : c3[ic][a];
    // gfs_cell_corner_interpolator (src/fluid.c:2983-3068): cells n[m], m = 0..2^DIM-1, offset by
    // (m&1) along x, (m&2) along y, (m&4) along z towards the corner
    double wsum = 0., w0 = 0.;
    int cnt = 0, boundaries = 0;
    bool ex[1 << DIM];
#pragma unroll
    for (int m = 0; m < (1 << DIM); m++) {
      int out = 0;
#pragma unroll
      for (int a = 0; a < DIM; a++)
	if (m & (1 << a))
	  out += (sg[a] < 0 ? outm[a] : outp[a]) ? 1 : 0;
      ex[m] = out <= 1;
      if (ex[m]) {
	if (cnt == 0) w0 = a_;
	cnt++;
	wsum += a_;
	if (out > 0) boundaries++;
      }
    }
    bool skip0 = (cnt == DIM + 1 && boundaries == DIM);   // domain corner: drop the central cell
    if (skip0)
      wsum -= w0;
    const double scale = 1./wsum;
    double r = 0.;
#pragma unroll
    for (int m = 0; m < (1 << DIM); m++) {
      int ox = (m & 1) ? sg[0] : 0, oy = (m & 2) ? sg[1] : 0, oz = (DIM == 3 && (m & 4)) ? sg[2] : 0;
      double val = nb[(ox + 1) + 3*(oy + 1) + 9*(DIM == 3 ? oz + 1 : 0)];
      if (ex[m] && !(skip0 && m == 0)) {
	double wm = a_*scale;
	r += wm*val;
      }
    }
    f[ic] = r;
  }
  f[NC] = nb[1 + 3 + (DIM == 3 ? 9 : 0)];

  // gfs_interpolate_from_corners, src/fluid.c:2640-2683
  double o[3] = { -0.5 + (cell[0] - 0.5)*h, -0.5 + (cell[1] - 0.5)*h,
		  DIM == 3 ? -0.5 + (cell[2] - 0.5)*h : 0. };
  const double size = h/2.;
  double p[3];
  p[0] = (p_[0] - o[0])/size;
  p[1] = (p_[1] - o[1])/size;
  if (DIM == 2) {
    double x = (p[0] + p[1])/2., y = (p[1] - p[0])/2., val = f[4];
    if (x > 0.)
      val += x*(f[2] - f[4]);
    else
      val -= x*(f[0] - f[4]);
    if (y > 0.)
      val += y*(f[3] - f[4]);
    else
      val -= y*(f[1] - f[4]);
    return val;
  }
  p[2] = (p_[2] - o[2])/size;
  double c[8];
  c[0] = - f[0] + f[1] + f[2] - f[3] - f[4] + f[5] + f[6] - f[7];
  c[1] = - f[0] - f[1] + f[2] + f[3] - f[4] - f[5] + f[6] + f[7];
  c[2] =   f[0] + f[1] + f[2] + f[3] - f[4] - f[5] - f[6] - f[7];
  c[3] =   f[0] - f[1] + f[2] - f[3] + f[4] - f[5] + f[6] - f[7];
  c[4] = - f[0] + f[1] + f[2] - f[3] + f[4] - f[5] - f[6] + f[7];
  c[5] = - f[0] - f[1] + f[2] + f[3] + f[4] + f[5] - f[6] - f[7];
  c[6] =   f[0] - f[1] + f[2] - f[3] - f[4] + f[5] - f[6] + f[7];
  c[7] =   f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
  return (c[0]*p[0] + c[1]*p[1] + c[2]*p[2] +
	  c[3]*p[0]*p[1] + c[4]*p[0]*p[2] + c[5]*p[1]*p[2] +
	  c[6]*p[0]*p[1]*p[2] +
	  c[7])/8.;
}

// check_intersetion, modules/particulatecommon.c:3058-3146
template <int DIM>
__device__ bool check_intersection (const double cellpos[3], const double p0[3], const double p1[3],
				    int * dstore, double size)
{
  for (int d = 0; d < 2*DIM; d++) {
    double normal = (double) (d ^ 1) - (double) d;
    int c = d/2;
    if ((p1[c] - p0[c]) != 0 && normal*(p1[c] - p0[c]) > 0) {
      double t = (cellpos[c] + normal*size*0.5 - p0[c])/(p1[c] - p0[c]);
      bool inside = true;
      for (int a = 0; a < DIM; a++)
	if (a != c) {
	  double pa = p0[a] + t*(p1[a] - p0[a]);
	  if (!((pa - cellpos[a] + size*0.5)*(pa - cellpos[a] - size*0.5) <= 0))
	    inside = false;
	}
      if (inside && t*(t - 1) <= 0) {
	*dstore = d;
	return true;
      }
    }
  }
  return false;
}

// gfs_particle_bc, modules/particulatecommon.c:3326-3395: a particle that left the box is taken
// off the list (or marked for the box across a GfsBoundaryMpi side) unless a periodic side puts it
// back; stores the position
template <int DIM>
__device__ void particle_bc_and_store (const PartArgs & A, int q, int depth, const int cell[3],
				       double p[3], double po[3])
{
  const int n = A.L.n;
  const double h = 1./n;
  int cn[3];
  bool keep = true;
  int mig = 0;
  if (!locate<DIM> (depth, p, cn)) {
    // boundarycell (:3149-3186): march from the cell of pos_old to the box side
    int d = 0;
    int cc[3] = { cell[0], cell[1], cell[2] };
    for (int guard = 0; guard < 4*n; guard++) {
      double cellpos[3] = { -0.5 + (cc[0] - 0.5)*h, -0.5 + (cc[1] - 0.5)*h,
			    DIM == 3 ? -0.5 + (cc[2] - 0.5)*h : 0. };
      check_intersection<DIM> (cellpos, po, p, &d, h);
      int c = d/2, next = cc[c] + ((d & 1) ? -1 : 1);
      if (next < 1 || next > n)
	break;
      cc[c] = next;
    }
    if (A.side[d] == GFSHIP_SIDE_EXTERNAL && A.migrate)
      mig = 2 + d;                  /* GfsBoundaryMpi: sent to the box across the side */
    else if (A.side[d] != GFSHIP_SIDE_PERIODIC)
      keep = false;                 /* taken off the list, nothing puts it back */
    else {
      // periodic_bc_particle (:3189-3214), box of size 1 matching itself
      double size = 1.;
      double normal = (double) (d ^ 1) - (double) d;
      double box_face = 0., box_face_nbr = 0.;
      box_face += normal*size/2.;
      box_face_nbr -= normal*size/2.;
      double tolerance = size/1.e8;
      double distance = (p[d/2] - box_face)*normal;
      p[d/2] = box_face_nbr + distance + normal*tolerance;
      po[d/2] = p[d/2];
    }
  }
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    A.pos[c][q] = p[c];
    A.old[c][q] = po[c];
  }
  if (!keep)
    A.alive[q] = 0;
  else if (mig)
    A.alive[q] = (unsigned char) mig;
}

template <int DIM>
__global__ void __launch_bounds__(256)
particle_list_event_kernel (PartArgs A, int depth)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= A.n) return;
  if (A.alive[q] != 1) return;
  double p[3] = { A.pos[0][q], A.pos[1][q], DIM == 3 ? A.pos[2][q] : 0. };
  int cell[3];
  // remove_particles_not_in_domain, modules/particulatecommon.c:955-969
  if (!locate<DIM> (depth, p, cell)) {
    A.alive[q] = 0;
    return;
  }
  // gfs_particle_event (src/particle.c:31-44): pos_old = pos; gfs_domain_advect_point
  double po[3] = { p[0], p[1], p[2] };
  {
    double p1[3] = { p[0], p[1], p[2] };
#pragma unroll
    for (int c = 0; c < DIM; c++)
      p1[c] += A.dt*interpolate<DIM> (A.L, A.u[c], cell, po)/2.;
    int cell1[3];
    if (locate<DIM> (depth, p1, cell1)) {
#pragma unroll
      for (int c = 0; c < DIM; c++)
	p[c] += A.dt*interpolate<DIM> (A.L, A.u[c], cell1, p1);
    }
  }
  particle_bc_and_store<DIM> (A, q, depth, cell, p, po);
}

// ---------------------------------------------------------------------------------------------
// GfsParticulate with forces (modules/particulatecommon.c:91-842): uniform box, alpha = NULL
// (fluid_rho = 1.), constant viscosity (GfsSourceDiffusion on U), no user coefficient functions
// (cm = cl = 0.5, the default drag law).  Same operand order as the reference; pow (Re, 0.5) of
// the drag law is evaluated as sqrt (Re) (correctly rounded; glibc's pow is within 0.52 ulp of it).
// ---------------------------------------------------------------------------------------------
enum { FORCE_INERTIAL = GFSHIP_FORCE_INERTIAL, FORCE_ADDEDMASS = GFSHIP_FORCE_ADDEDMASS,
       FORCE_LIFT = GFSHIP_FORCE_LIFT, FORCE_DRAG = GFSHIP_FORCE_DRAG, FORCE_BUOY = GFSHIP_FORCE_BUOY };

struct ParticulateArgs {
  PartArgs P;
  const unsigned * orig;
  double * vel[3], * force[3], * mass;
  const double * volume, * dia;
  const double * uold[3];
  int nforces, forces[8];
  double gravity[3], viscosity;
  const double * coef[8];      // values of the GfsFunction of force f per slot; nullptr: the default
  double * cin[6];             // Rep, Urelp, Vrelp, Wrelp, Pdia per slot (coefficient inputs); [5]: the Rep of
                               // GfsForceDrag in 2-D (two-component norm, particulatecommon.c:549-556)
};

// gfs_center_gradient, src/fluid.c:434-475, both neighbours at the same level (x1 = x2 = 1.)
__device__ __forceinline__ double center_gradient (const double * __restrict__ v, long idx, long off)
{
  const double v0 = v[idx], v1 = v[idx - off], v2 = v[idx + off];
  return ((v2 - v0) + (v0 - v1))/2.;
}

// compute_inertial_force, :285-336
// (fvel: the fluid velocity interpolated at the particle, computed once per particle and shared with
// the lift and drag forces: the same gfs_interpolate of the same fields)
template <int DIM>
__device__ void inertial_force (const ParticulateArgs & A, const int cell[3], const double p[3],
				long idx, const double fvel[3], double force[3])
{
  const Layout & L = A.P.L;
  const long off[3] = { 1, (long) L.sy, (long) L.sz };
  const double size = 1./L.n;
  const double fluid_rho = 1.;
  force[0] = force[1] = force[2] = 0.;
  if (!(A.P.dt > 0.))
    return;
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    const double fluid_vel = fvel[c];
    const double fluid_veln = interpolate<DIM> (L, A.uold[c], cell, p);
    force[c] = fluid_rho*(fluid_vel - fluid_veln)/A.P.dt;
  }
#pragma unroll
  for (int c = 0; c < DIM; c++)
#pragma unroll
    for (int c2 = 0; c2 < DIM; c2++)
      force[c] += fluid_rho*center_gradient (A.P.u[c], idx, off[c2])*A.P.u[c2][idx]/size;
}

// the variables a GfsFunction of a GfsForceCoeff sees (:364-384,462-485,545-573): the particle
// Reynolds number from the relative velocity (all three components of the FttVectors), the sphere
// diameter and the viscosity (0.001 where the reference substitutes it for a zero viscosity in the
// added-mass and lift coefficients; the drag is then zero whatever its coefficient)
template <int DIM>
__global__ void __launch_bounds__(256)
particulate_coeff_inputs_kernel (ParticulateArgs A, int depth)
{
  const PartArgs & P = A.P;
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= P.n) return;
  if (P.alive[q] != 1) return;
  const Layout & L = P.L;
  double p[3] = { P.pos[0][q], P.pos[1][q], DIM == 3 ? P.pos[2][q] : 0. };
  int cell[3];
  if (!locate<DIM> (depth, p, cell)) return;
  const unsigned o = A.orig[q];
  const double fluid_rho = 1.;
  double rel[3] = { 0. - A.vel[0][o], 0. - A.vel[1][o], 0. - A.vel[2][o] };
#pragma unroll
  for (int c = 0; c < DIM; c++)
    rel[c] = interpolate<DIM> (L, P.u[c], cell, p) - A.vel[c][o];
  const double norm = sqrt (rel[0]*rel[0] + rel[1]*rel[1] + rel[2]*rel[2]);
  const double dia = A.dia[o];
  const double viscosity = A.viscosity == 0 ? 0.001 : A.viscosity;
  A.cin[0][q] = norm*dia*fluid_rho/viscosity;
  A.cin[1][q] = rel[0]; A.cin[2][q] = rel[1]; A.cin[3][q] = rel[2];
  A.cin[4][q] = dia;
  if (DIM == 2)      /* compute_drag_force takes the norm of the two components in 2-D (:549-556) */
    A.cin[5][q] = sqrt (rel[0]*rel[0] + rel[1]*rel[1])*dia*fluid_rho/viscosity;
}

// gfs_particulate_event (:768-842) in a gfs_particle_list_event (:980-1015)
template <int DIM>
__global__ void __launch_bounds__(256)
particulate_list_event_kernel (ParticulateArgs A, int depth)
{
  const PartArgs & P = A.P;
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= P.n) return;
  if (P.alive[q] != 1) return;
  const Layout & L = P.L;
  double p[3] = { P.pos[0][q], P.pos[1][q], DIM == 3 ? P.pos[2][q] : 0. };
  int cell[3];
  // remove_particles_not_in_domain, :955-969
  if (!locate<DIM> (depth, p, cell)) {
    P.alive[q] = 0;
    return;
  }
  const unsigned o = A.orig[q];
  const long idx = L.idx (cell[0], cell[1], DIM == 3 ? cell[2] : 0);
  const long off[3] = { 1, (long) L.sy, (long) L.sz };
  const double size = 1./L.n;
  const double fluid_rho = 1.;
  const double dt = P.dt;
  double po[3] = { p[0], p[1], p[2] };
  double vel[3] = { A.vel[0][o], A.vel[1][o], A.vel[2][o] };
  double mass = A.mass[o];
  const double volume = A.volume[o];
  double pf[3] = { 0., 0., 0. };
  // every force is a function of the state at the start of the event: what several of them need --
  // the fluid velocity at the particle, Du/Dt -- is evaluated once (the reference re-evaluates the
  // same expressions per force: same bits)
  bool need_fvel = false, need_dudt = false;
  for (int f = 0; f < A.nforces; f++) {
    need_fvel = need_fvel || A.forces[f] != FORCE_BUOY;
    need_dudt = need_dudt || A.forces[f] == FORCE_INERTIAL || A.forces[f] == FORCE_ADDEDMASS;
  }
  double fvel[3] = { 0., 0., 0. }, dudt[3] = { 0., 0., 0. };
  if (need_fvel) {
#pragma unroll
    for (int c = 0; c < DIM; c++)
      fvel[c] = interpolate<DIM> (L, P.u[c], cell, p);
  }
  if (need_dudt)
    inertial_force<DIM> (A, cell, p, idx, fvel, dudt);
  for (int f = 0; f < A.nforces; f++) {
    double force[3] = { 0., 0., 0. };
    switch (A.forces[f]) {
    case FORCE_INERTIAL:
      force[0] = dudt[0]; force[1] = dudt[1]; force[2] = dudt[2];
      break;
    case FORCE_ADDEDMASS: {     // compute_addedmass_force, :363-427
      force[0] = dudt[0]; force[1] = dudt[1]; force[2] = dudt[2];
      const double cm = A.coef[f] ? A.coef[f][q] : 0.5;
#pragma unroll
      for (int c = 0; c < DIM; c++)
	force[c] *= cm;
      mass += fluid_rho*volume*cm;
      break;
    }
    case FORCE_LIFT: {          // compute_lift_force, :455-524; vorticity_vector, :146-168
      double rel[3] = { 0., 0., 0. }, vort[3];
#pragma unroll
      for (int c = 0; c < DIM; c++)
	rel[c] = fvel[c] - vel[c];
      if (DIM == 2) {
	vort[0] = 0.; vort[1] = 0.;
	vort[2] = (center_gradient (P.u[1], idx, off[0]) - center_gradient (P.u[0], idx, off[1]))/size;
      }
      else {
	vort[0] = (center_gradient (P.u[2], idx, off[1]) - center_gradient (P.u[1], idx, off[2]))/size;
	vort[1] = (center_gradient (P.u[0], idx, off[2]) - center_gradient (P.u[2], idx, off[0]))/size;
	vort[2] = (center_gradient (P.u[1], idx, off[0]) - center_gradient (P.u[0], idx, off[1]))/size;
      }
      const double cl = A.coef[f] ? A.coef[f][q] : 0.5;
      if (DIM == 2) {
	force[0] = fluid_rho*cl*rel[1]*vort[2];
	force[1] = -fluid_rho*cl*rel[0]*vort[2];
      }
      else {
	force[0] = fluid_rho*cl*(rel[1]*vort[2] - rel[2]*vort[1]);
	force[1] = fluid_rho*cl*(rel[2]*vort[0] - rel[0]*vort[2]);
	force[2] = fluid_rho*cl*(rel[0]*vort[1] - rel[1]*vort[0]);
      }
      break;
    }
    case FORCE_DRAG: {          // compute_drag_force, :527-588
      double rel[3] = { 0., 0., 0. };
#pragma unroll
      for (int c = 0; c < DIM; c++)
	rel[c] = fvel[c] - vel[c];
      const double dia = A.dia[o];
      const double norm = DIM == 3 ? sqrt (rel[0]*rel[0] + rel[1]*rel[1] + rel[2]*rel[2]) :
	sqrt (rel[0]*rel[0] + rel[1]*rel[1]);
      if (A.viscosity == 0)
	break;
      const double Re = norm*dia*fluid_rho/A.viscosity;
      double cd;
      if (A.coef[f])
	cd = A.coef[f][q];
      else if (Re < 1e-8)
	break;
      else if (Re < 50.0)
	cd = 16.*(1. + 0.15*sqrt (Re))/Re;
      else
	cd = 48.*(1. - 2.21/sqrt (Re))/Re;
#pragma unroll
      for (int c = 0; c < DIM; c++)
	force[c] += 3./(4.*dia)*cd*norm*rel[c]*fluid_rho;
      break;
    }
    case FORCE_BUOY:            // compute_buoyancy_force, :619-653
#pragma unroll
      for (int c = 0; c < DIM; c++)
	force[c] += (mass/volume - fluid_rho)*A.gravity[c];
      break;
    }
    // compute_forces, :738-752
#pragma unroll
    for (int c = 0; c < DIM; c++)
      pf[c] = force[c]*volume + pf[c];
    if (DIM == 2) pf[2] = 0.;
  }
#pragma unroll
  for (int c = 0; c < DIM; c++) {
    p[c] += vel[c]*dt/2.;
    vel[c] += pf[c]*dt/mass;
    p[c] += vel[c]*dt/2.;
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    A.vel[c][o] = vel[c];
    A.force[c][o] = pf[c];
  }
  A.mass[o] = mass;
  particle_bc_and_store<DIM> (P, q, depth, cell, p, po);
}


// mpi_send_particle (modules/particulatecommon.c:3218-3222): the particles marked 2 + d go into the
// packet of side d (7 doubles: position and old position in the coordinates of the receiving
// box -- the same point of space, one box size less along the normal -- and the id) and leave
// the list
struct OutboxArgs {
  int n, out_cap, rs;          // rs doubles per record: 7, or 15 for particulates
  const double * pos[3], * old[3];
  const unsigned * id;
  unsigned char * alive;
  double * outbox;
  unsigned * out_count;
  const unsigned * orig;       // particulates: state indexed by the creation slot
  const double * vel[3], * force[3], * mass, * volume;
};

__global__ void __launch_bounds__(256)
particle_outbox_kernel (OutboxArgs A)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= A.n) return;
  int code = A.alive[q];
  if (code < 2) return;
  int d = code - 2;
  A.alive[q] = 0;
  unsigned k = atomicAdd (A.out_count + d, 1u);
  if (k >= (unsigned) A.out_cap) return;      /* the host sees the count and reports the overflow */
  double * r = A.outbox + (size_t) A.rs*((size_t) d*A.out_cap + k);
  double normal = (double) (d ^ 1) - (double) d;
#pragma unroll
  for (int c = 0; c < 3; c++) { r[c] = A.pos[c][q]; r[3 + c] = A.old[c][q]; }
  r[d/2] -= normal*1.;
  r[3 + d/2] -= normal*1.;
  r[6] = (double) A.id[q];
  if (A.rs > 7) {
    const unsigned o = A.orig[q];
#pragma unroll
    for (int c = 0; c < 3; c++) { r[7 + c] = A.vel[c][o]; r[12 + c] = A.force[c][o]; }
    r[10] = A.mass[o];
    r[11] = A.volume[o];
  }
}

// key = linear index of the containing leaf cell, dead or outside particles last
template <int DIM>
__global__ void __launch_bounds__(256)
particle_keys_kernel (Layout L, int depth, int n, const double * __restrict__ x,
		      const double * __restrict__ y, const double * __restrict__ z,
		      const unsigned char * __restrict__ alive, unsigned * __restrict__ key,
		      unsigned * __restrict__ slot)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= n) return;
  unsigned k = 0xFFFFFFFFu;
  if (alive[q] == 1) {
    double p[3] = { x[q], y[q], DIM == 3 ? z[q] : 0. };
    int c[3];
    if (locate<DIM> (depth, p, c))
      k = (unsigned) (c[0] - 1) + (unsigned) L.n*((unsigned) (c[1] - 1) +
						 (DIM == 3 ? (unsigned) L.n*(unsigned) (c[2] - 1) : 0u));
  }
  key[q] = k;
  slot[q] = q;
}

struct GatherArgs {
  int n;
  const unsigned * slot;
  const double * pos[3], * old[3];
  const unsigned * id, * orig;
  const unsigned char * alive;
  double * pos2[3], * old2[3];
  unsigned * id2, * orig2;
  unsigned char * alive2;
};

__global__ void __launch_bounds__(256)
particle_gather_kernel (GatherArgs G)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= G.n) return;
  unsigned s = G.slot[q];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    G.pos2[c][q] = G.pos[c][s];
    G.old2[c][q] = G.old[c][s];
  }
  G.id2[q] = G.id[s];
  G.orig2[q] = G.orig[s];
  G.alive2[q] = G.alive[s];
}

// GfsOutputLocation (src/output.c:1182-1199): gfs_domain_locate then gfs_interpolate of one
// variable at arbitrary points
template <int DIM>
__global__ void __launch_bounds__(256)
sample_kernel (Layout L, int depth, int np, const double * __restrict__ pos,
	       const double * __restrict__ v, double * __restrict__ out,
	       unsigned char * __restrict__ inside)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q >= np) return;
  double p[3] = { pos[3*(size_t) q], pos[3*(size_t) q + 1], DIM == 3 ? pos[3*(size_t) q + 2] : 0. };
  int cell[3];
  if (!locate<DIM> (depth, p, cell)) {
    inside[q] = 0;
    out[q] = 0.;
    return;
  }
  inside[q] = 1;
  out[q] = interpolate<DIM> (L, v, cell, p);
}

__global__ void __launch_bounds__(256)
count_alive_kernel (const unsigned char * alive, int n, unsigned * count)
{
  int q = blockIdx.x*blockDim.x + threadIdx.x;
  unsigned long long m = __ballot (q < n && alive[q] == 1);
  if ((threadIdx.x & 63) == 0 && m)
    atomicAdd (count, (unsigned) __popcll (m));
}

} // namespace gfship

using namespace gfship;

extern "C" {

struct gfship_sim_view { gfship_domain * dom; const gfship_field * u; double dt; double visc; };
gfship_sim_view gfship_sim_view_get (gfship_sim * s);   /* simulation.hip */

int gfship_particles_create (gfship_particles ** out, gfship_sim * sim, int np,
			     const double * pos, const unsigned * id)
{
  GFSHIP_CHECK (out && sim && (np == 0 || (pos && id)) && np >= 0, GFSHIP_EINVAL, "invalid argument");
  *out = nullptr;
  gfship_particles * pl = new gfship_particles;
  pl->sim = sim;
  pl->dom = gfship_sim_view_get (sim).dom;
  pl->n = np;
  pl->np0 = np;
  pl->cap = std::max (np, 1);
  size_t m = pl->cap;
  std::vector<double> tmp (m);
  for (int c = 0; c < 3; c++) {
    GFSHIP_HIP (hipMalloc ((void **) &pl->pos[c], m*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &pl->old[c], m*sizeof (double)));
    for (int q = 0; q < np; q++) tmp[q] = pos[3*(size_t) q + c];
    GFSHIP_HIP (hipMemcpy (pl->pos[c], tmp.data (), m*sizeof (double), hipMemcpyHostToDevice));
    GFSHIP_HIP (hipMemcpy (pl->old[c], tmp.data (), m*sizeof (double), hipMemcpyHostToDevice));
  }
  GFSHIP_HIP (hipMalloc ((void **) &pl->id, m*sizeof (unsigned)));
  if (np) GFSHIP_HIP (hipMemcpy (pl->id, id, np*sizeof (unsigned), hipMemcpyHostToDevice));
  GFSHIP_HIP (hipMalloc ((void **) &pl->alive, m));
  GFSHIP_HIP (hipMemset (pl->alive, 1, m));
  GFSHIP_HIP (hipMalloc ((void **) &pl->d_count, sizeof (unsigned)));
  for (int c = 0; c < 3; c++) {
    GFSHIP_HIP (hipMalloc ((void **) &pl->pos2[c], m*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &pl->old2[c], m*sizeof (double)));
  }
  unsigned ** ua[] = { &pl->orig, &pl->orig2, &pl->id2, &pl->key, &pl->key2, &pl->slot, &pl->slot2 };
  for (unsigned ** a : ua)
    GFSHIP_HIP (hipMalloc ((void **) a, m*sizeof (unsigned)));
  GFSHIP_HIP (hipMalloc ((void **) &pl->alive2, m));
  {
    std::vector<unsigned> iota (m);
    std::iota (iota.begin (), iota.end (), 0u);
    GFSHIP_HIP (hipMemcpy (pl->orig, iota.data (), m*sizeof (unsigned), hipMemcpyHostToDevice));
  }
  *out = pl;
  return GFSHIP_OK;
}

void gfship_particles_destroy (gfship_particles * pl)
{
  if (!pl) return;
  (void) hipStreamSynchronize (pl->dom->stream);
  for (int c = 0; c < 3; c++) {
    if (pl->pos[c]) (void) hipFree (pl->pos[c]);
    if (pl->old[c]) (void) hipFree (pl->old[c]);
  }
  if (pl->id) (void) hipFree (pl->id);
  if (pl->alive) (void) hipFree (pl->alive);
  if (pl->d_count) (void) hipFree (pl->d_count);
  for (int c = 0; c < 3; c++) {
    if (pl->pos2[c]) (void) hipFree (pl->pos2[c]);
    if (pl->old2[c]) (void) hipFree (pl->old2[c]);
  }
  void * extra[] = { pl->orig, pl->orig2, pl->id2, pl->key, pl->key2, pl->slot, pl->slot2,
		     pl->alive2, pl->sort_tmp, pl->outbox, pl->out_count };
  for (void * a : extra)
    if (a) (void) hipFree (a);
  for (int c = 0; c < 3; c++) {
    if (pl->vel[c]) (void) hipFree (pl->vel[c]);
    if (pl->force[c]) (void) hipFree (pl->force[c]);
    if (pl->uold[c] >= 0) gfship_field_free (pl->dom, pl->uold[c]);
  }
  if (pl->mass) (void) hipFree (pl->mass);
  if (pl->volume) (void) hipFree (pl->volume);
  if (pl->dia) (void) hipFree (pl->dia);
  for (int f = 0; f < 8; f++) {
    if (pl->coef[f]) (void) hipFree (pl->coef[f]);
    gfship::rtc_free (pl->coef_fn[f]);
  }
  for (int q = 0; q < 6; q++)
    if (pl->cin[q]) (void) hipFree (pl->cin[q]);
  delete pl;
}

int gfship_field_interpolate (gfship_domain * dom, gfship_field v, int np, const double * pos,
			      double * out, unsigned char * inside)
{
  GFSHIP_CHECK (dom && pos && out && inside && np >= 0, GFSHIP_EINVAL, "invalid argument");
  Field * V = get_field (dom, v);
  if (!V) return GFSHIP_EINVAL;
  if (np == 0) return GFSHIP_OK;
  double * dpos = nullptr, * dout = nullptr;
  unsigned char * din = nullptr;
  GFSHIP_HIP (hipMalloc ((void **) &dpos, 3*(size_t) np*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &dout, (size_t) np*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &din, (size_t) np));
  int r = GFSHIP_OK;
  hipError_t e = hipMemcpyAsync (dpos, pos, 3*(size_t) np*sizeof (double), hipMemcpyHostToDevice,
				 dom->stream);
  if (e == hipSuccess) {
    int block = 256, grid = (np + block - 1)/block;
    const Layout & L = dom->lay[dom->depth];
    if (dom->dim == 3)
      hipLaunchKernelGGL (sample_kernel<3>, dim3 (grid), dim3 (block), 0, dom->stream, L, dom->depth,
			  np, dpos, V->lev[dom->depth], dout, din);
    else
      hipLaunchKernelGGL (sample_kernel<2>, dim3 (grid), dim3 (block), 0, dom->stream, L, dom->depth,
			  np, dpos, V->lev[dom->depth], dout, din);
    e = hipGetLastError ();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync (out, dout, (size_t) np*sizeof (double), hipMemcpyDeviceToHost, dom->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync (inside, din, (size_t) np, hipMemcpyDeviceToHost, dom->stream);
  if (e == hipSuccess)
    e = hipStreamSynchronize (dom->stream);
  if (e != hipSuccess)
    r = hip_fail (e, "gfship_field_interpolate", __FILE__, __LINE__);
  (void) hipFree (dpos); (void) hipFree (dout); (void) hipFree (din);
  return r;
}

int gfship_particles_set_sort_interval (gfship_particles * pl, int every)
{
  GFSHIP_CHECK (pl != nullptr && every >= 0, GFSHIP_EINVAL, "invalid argument");
  pl->sort_every = every;
  return GFSHIP_OK;
}

int gfship_particles_sort (gfship_particles * pl)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  if (pl->n == 0) return GFSHIP_OK;
  gfship_domain * dom = pl->dom;
  const Layout & L = dom->lay[dom->depth];
  int block = 256, grid = (pl->n + block - 1)/block;
  if (dom->dim == 3)
    hipLaunchKernelGGL (particle_keys_kernel<3>, dim3 (grid), dim3 (block), 0, dom->stream, L,
			dom->depth, pl->n, pl->pos[0], pl->pos[1], pl->pos[2], pl->alive, pl->key,
			pl->slot);
  else
    hipLaunchKernelGGL (particle_keys_kernel<2>, dim3 (grid), dim3 (block), 0, dom->stream, L,
			dom->depth, pl->n, pl->pos[0], pl->pos[1], pl->pos[2], pl->alive, pl->key,
			pl->slot);
  GFSHIP_HIP (hipGetLastError ());
  size_t need = 0;
  GFSHIP_HIP (hipcub::DeviceRadixSort::SortPairs (nullptr, need, pl->key, pl->key2, pl->slot,
						  pl->slot2, pl->n, 0, 32, dom->stream));
  if (need > pl->sort_tmp_bytes) {
    if (pl->sort_tmp) GFSHIP_HIP (hipFree (pl->sort_tmp));
    pl->sort_tmp = nullptr;
    GFSHIP_HIP (hipMalloc (&pl->sort_tmp, need));
    pl->sort_tmp_bytes = need;
  }
  /* the key needs dim*depth bits (+1 so that the all-ones key of the dead sorts last) */
  int bits = dom->dim*dom->depth + 1;
  if (bits > 32) bits = 32;
  size_t tmp_bytes = pl->sort_tmp_bytes;
  GFSHIP_HIP (hipcub::DeviceRadixSort::SortPairs (pl->sort_tmp, tmp_bytes, pl->key, pl->key2,
						  pl->slot, pl->slot2, pl->n, 0, bits, dom->stream));
  GatherArgs G;
  G.n = pl->n;
  G.slot = pl->slot2;
  for (int c = 0; c < 3; c++) {
    G.pos[c] = pl->pos[c]; G.old[c] = pl->old[c];
    G.pos2[c] = pl->pos2[c]; G.old2[c] = pl->old2[c];
  }
  G.id = pl->id; G.orig = pl->orig; G.alive = pl->alive;
  G.id2 = pl->id2; G.orig2 = pl->orig2; G.alive2 = pl->alive2;
  hipLaunchKernelGGL (particle_gather_kernel, dim3 (grid), dim3 (block), 0, dom->stream, G);
  GFSHIP_HIP (hipGetLastError ());
  for (int c = 0; c < 3; c++) {
    std::swap (pl->pos[c], pl->pos2[c]);
    std::swap (pl->old[c], pl->old2[c]);
  }
  std::swap (pl->id, pl->id2);
  std::swap (pl->orig, pl->orig2);
  std::swap (pl->alive, pl->alive2);
  pl->events_since_sort = 0;
  return GFSHIP_OK;
}

// grow the per-particle arrays to at least `need` slots (contents of the live arrays kept)
static int particles_reserve (gfship_particles * pl, int need)
{
  if (need <= pl->cap) return GFSHIP_OK;
  gfship_domain * dom = pl->dom;
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  size_t m = std::max (need, 2*pl->cap), old = pl->cap;
  auto grow = [&] (void ** a, size_t elem, bool keep) -> hipError_t {
    void * nw = nullptr;
    hipError_t e = hipMalloc (&nw, m*elem);
    if (e != hipSuccess) return e;
    if (keep && *a) e = hipMemcpy (nw, *a, old*elem, hipMemcpyDeviceToDevice);
    if (*a) (void) hipFree (*a);
    *a = nw;
    return e;
  };
  for (int c = 0; c < 3; c++) {
    GFSHIP_HIP (grow ((void **) &pl->pos[c], sizeof (double), true));
    GFSHIP_HIP (grow ((void **) &pl->old[c], sizeof (double), true));
    GFSHIP_HIP (grow ((void **) &pl->pos2[c], sizeof (double), false));
    GFSHIP_HIP (grow ((void **) &pl->old2[c], sizeof (double), false));
  }
  GFSHIP_HIP (grow ((void **) &pl->id, sizeof (unsigned), true));
  GFSHIP_HIP (grow ((void **) &pl->orig, sizeof (unsigned), true));
  GFSHIP_HIP (grow ((void **) &pl->alive, 1, true));
  unsigned ** scratch[] = { &pl->orig2, &pl->id2, &pl->key, &pl->key2, &pl->slot, &pl->slot2 };
  for (unsigned ** a : scratch)
    GFSHIP_HIP (grow ((void **) a, sizeof (unsigned), false));
  GFSHIP_HIP (grow ((void **) &pl->alive2, 1, false));
  if (pl->particulate) {
    for (int c = 0; c < 3; c++) {
      GFSHIP_HIP (grow ((void **) &pl->vel[c], sizeof (double), true));
      GFSHIP_HIP (grow ((void **) &pl->force[c], sizeof (double), true));
    }
    GFSHIP_HIP (grow ((void **) &pl->mass, sizeof (double), true));
    GFSHIP_HIP (grow ((void **) &pl->volume, sizeof (double), true));
    GFSHIP_HIP (grow ((void **) &pl->dia, sizeof (double), true));
  }
  pl->cap = (int) m;
  return GFSHIP_OK;
}

// send_particles / rcv_particles through the GfsBoundaryMpi sides
// (modules/particulatecommon.c:3247-3312): packets per side out, the neighbours' packets in
static int particles_migrate (gfship_particles * pl)
{
  gfship_domain * dom = pl->dom;
  const int rs = pl->particulate ? 15 : 7;
  const int want = std::max (4096, pl->n/4);
  if (pl->out_cap < want) {
    if (pl->outbox) GFSHIP_HIP (hipFree (pl->outbox));
    pl->outbox = nullptr;
    GFSHIP_HIP (hipMalloc ((void **) &pl->outbox, 6*(size_t) want*15*sizeof (double)));
    pl->out_cap = want;
  }
  if (!pl->out_count)
    GFSHIP_HIP (hipMalloc ((void **) &pl->out_count, 6*sizeof (unsigned)));
  GFSHIP_HIP (hipMemsetAsync (pl->out_count, 0, 6*sizeof (unsigned), dom->stream));
  OutboxArgs O;
  O.n = pl->n; O.out_cap = pl->out_cap; O.rs = rs;
  O.orig = pl->orig; O.mass = pl->mass; O.volume = pl->volume;
  for (int c = 0; c < 3; c++) { O.vel[c] = pl->vel[c]; O.force[c] = pl->force[c]; }
  for (int c = 0; c < 3; c++) { O.pos[c] = pl->pos[c]; O.old[c] = pl->old[c]; }
  O.id = pl->id; O.alive = pl->alive; O.outbox = pl->outbox; O.out_count = pl->out_count;
  if (pl->n > 0) {
    hipLaunchKernelGGL (particle_outbox_kernel, dim3 ((pl->n + 255)/256), dim3 (256), 0, dom->stream, O);
    GFSHIP_HIP (hipGetLastError ());
  }
  unsigned cnt[6];
  GFSHIP_HIP (hipMemcpyAsync (cnt, pl->out_count, sizeof (cnt), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  std::vector<double> send[6];
  int nsend[6], nrecv[6] = { 0, 0, 0, 0, 0, 0 };
  const double * sp[6], * rp[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
  for (int d = 0; d < 6; d++) {
    GFSHIP_CHECK (cnt[d] <= (unsigned) pl->out_cap, GFSHIP_ENOMEM,
		  "%u particles leave through side %d in one step: more than the %d the packet holds",
		  cnt[d], d, pl->out_cap);
    nsend[d] = (int) cnt[d];
    send[d].resize ((size_t) rs*cnt[d]);
    if (cnt[d])
      GFSHIP_HIP (hipMemcpy (send[d].data (), pl->outbox + (size_t) rs*d*pl->out_cap,
			     (size_t) rs*cnt[d]*sizeof (double), hipMemcpyDeviceToHost));
    /* the device fills the packet in no particular order: sort it by id */
    std::vector<size_t> o (cnt[d]);
    std::iota (o.begin (), o.end (), (size_t) 0);
    std::sort (o.begin (), o.end (), [&] (size_t a, size_t b) { return send[d][rs*a + 6] < send[d][rs*b + 6]; });
    std::vector<double> sorted (send[d].size ());
    for (size_t q = 0; q < o.size (); q++)
      memcpy (&sorted[rs*q], &send[d][rs*o[q]], rs*sizeof (double));
    send[d].swap (sorted);
    sp[d] = send[d].data ();
  }
  int r;
  std::vector<double> inbox[6];
  if (pl->migrate) {
    r = pl->migrate (pl->migrate_ctx, nsend, sp, nrecv, rp);
    GFSHIP_CHECK (r == 0, GFSHIP_EHIP, "the particle migration hook failed (%d)", r);
  }
  else {
    /* the library's communicator */
    if ((r = comm_migrate (dom, rs, nsend, sp, nrecv, inbox))) return r;
    for (int d = 0; d < 6; d++) rp[d] = inbox[d].data ();
  }
  int total = 0;
  for (int d = 0; d < 6; d++) total += nrecv[d];
  if (total == 0) return GFSHIP_OK;
  if ((r = particles_reserve (pl, pl->n + total))) return r;
  std::vector<double> col (total);
  std::vector<unsigned> ids (total), orig (total);
  std::vector<unsigned char> one (total, 1);
  for (int c = 0; c < 6; c++) {
    size_t k = 0;
    for (int d = 0; d < 6; d++)             /* box_rcv_bc: side by side, in packet order */
      for (int q = 0; q < nrecv[d]; q++)
	col[k++] = rp[d][(size_t) rs*q + c];
    GFSHIP_HIP (hipMemcpy ((c < 3 ? pl->pos[c] : pl->old[c - 3]) + pl->n, col.data (),
			   total*sizeof (double), hipMemcpyHostToDevice));
  }
  size_t k = 0;
  for (int d = 0; d < 6; d++)
    for (int q = 0; q < nrecv[d]; q++, k++) {
      ids[k] = (unsigned) rp[d][(size_t) rs*q + 6];
      orig[k] = (unsigned) (pl->n + k);
    }
  GFSHIP_HIP (hipMemcpy (pl->id + pl->n, ids.data (), total*sizeof (unsigned), hipMemcpyHostToDevice));
  GFSHIP_HIP (hipMemcpy (pl->orig + pl->n, orig.data (), total*sizeof (unsigned), hipMemcpyHostToDevice));
  GFSHIP_HIP (hipMemcpy (pl->alive + pl->n, one.data (), total, hipMemcpyHostToDevice));
  if (pl->particulate) {
    /* the state of the newcomers lives at their creation slots pl->n .. pl->n + total - 1 */
    for (int c = 0; c < 9; c++) {      /* record columns 7 .. 14: vel, mass, volume, force; then dia */
      size_t kk = 0;
      for (int d = 0; d < 6; d++)
	for (int q = 0; q < nrecv[d]; q++) {
	  const double * r = rp[d] + (size_t) rs*q;
	  col[kk++] = c < 8 ? r[7 + c] : 2.*pow (3.0*r[11]/4.0/M_PI, 1./3.);
	}
      double * dst = c < 3 ? pl->vel[c] : c == 3 ? pl->mass : c == 4 ? pl->volume :
	c < 8 ? pl->force[c - 5] : pl->dia;
      GFSHIP_HIP (hipMemcpy (dst + pl->n, col.data (), total*sizeof (double), hipMemcpyHostToDevice));
    }
  }
  pl->n += total;
  return GFSHIP_OK;
}

int gfship_particles_set_migrate (gfship_particles * pl, gfship_particle_migrate_fn fn, void * ctx)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  pl->migrate = fn;
  pl->migrate_ctx = ctx;
  return GFSHIP_OK;
}

int gfship_particles_record_size (gfship_particles * pl)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  return pl->particulate ? 15 : 7;
}

int gfship_particles_slots (gfship_particles * pl)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  return pl->n;
}

// store_domain_previous_vel, modules/particulatecommon.c:100-114: copy on the leaves, then the BC
// of the new variable (a scalar with the default BC)
static int store_previous_vel (gfship_particles * pl)
{
  gfship_domain * dom = pl->dom;
  gfship_sim_view v = gfship_sim_view_get (pl->sim);
  const Layout & L = dom->lay[dom->depth];
  for (int c = 0; c < dom->dim; c++) {
    GFSHIP_HIP (hipMemcpyAsync (dom->fields[pl->uold[c]].lev[dom->depth],
				dom->fields[v.u[c]].lev[dom->depth], L.total*sizeof (double),
				hipMemcpyDeviceToDevice, dom->stream));
    int r = gfship_bc (dom, pl->uold[c], pl->uold[c], dom->depth);
    if (r) return r;
  }
  return GFSHIP_OK;
}

static int particulate_event (gfship_particles * pl, const PartArgs & P, double viscosity)
{
  gfship_domain * dom = pl->dom;
  ParticulateArgs A;
  A.P = P;
  A.orig = pl->orig;
  for (int c = 0; c < 3; c++) {
    A.vel[c] = pl->vel[c];
    A.force[c] = pl->force[c];
    A.uold[c] = c < dom->dim && pl->uold[c] >= 0 ? dom->fields[pl->uold[c]].lev[dom->depth] : nullptr;
    A.gravity[c] = pl->gravity[c];
  }
  A.mass = pl->mass; A.volume = pl->volume; A.dia = pl->dia;
  A.nforces = pl->nforces;
  for (int f = 0; f < 8; f++) A.forces[f] = pl->forces[f];
  A.viscosity = viscosity;
  int block = 256, grid = (pl->n + block - 1)/block;
  bool any = false;
  for (int f = 0; f < 8; f++) {
    A.coef[f] = nullptr;
    if (f < pl->nforces && pl->coef_fn[f]) any = true;
  }
  for (int q = 0; q < 6; q++) A.cin[q] = nullptr;
  if (any) {
    /* the GfsFunction coefficients: inputs of every particle, then one compiled kernel per function */
    if (pl->coef_cap < pl->cap) {
      for (int q = 0; q < 6; q++) {
	if (pl->cin[q]) GFSHIP_HIP (hipFree (pl->cin[q]));
	pl->cin[q] = nullptr;
	GFSHIP_HIP (hipMalloc ((void **) &pl->cin[q], (size_t) pl->cap*sizeof (double)));
      }
      for (int f = 0; f < 8; f++) {
	if (pl->coef[f]) GFSHIP_HIP (hipFree (pl->coef[f]));
	pl->coef[f] = nullptr;
	if (pl->coef_fn[f])
	  GFSHIP_HIP (hipMalloc ((void **) &pl->coef[f], (size_t) pl->cap*sizeof (double)));
      }
      pl->coef_cap = pl->cap;
    }
    for (int q = 0; q < 6; q++) A.cin[q] = pl->cin[q];
    if (dom->dim == 3)
      hipLaunchKernelGGL (particulate_coeff_inputs_kernel<3>, dim3 (grid), dim3 (block), 0, dom->stream,
			  A, dom->depth);
    else
      hipLaunchKernelGGL (particulate_coeff_inputs_kernel<2>, dim3 (grid), dim3 (block), 0, dom->stream,
			  A, dom->depth);
    GFSHIP_HIP (hipGetLastError ());
    const double * rel[3] = { pl->cin[1], pl->cin[2], pl->cin[3] };
    for (int f = 0; f < pl->nforces; f++)
      if (pl->coef_fn[f]) {
	if (!pl->coef[f])
	  GFSHIP_HIP (hipMalloc ((void **) &pl->coef[f], (size_t) pl->coef_cap*sizeof (double)));
	const double * rep = dom->dim == 2 && pl->forces[f] == GFSHIP_FORCE_DRAG ? pl->cin[5] : pl->cin[0];
	int r = rtc_launch_coefficient (pl->coef_fn[f], dom->stream, pl->n, P.alive, rep, rel, pl->cin[4],
					gfship_sim_time (pl->sim), pl->coef[f]);
	if (r) return r;
	A.coef[f] = pl->coef[f];
      }
  }
  if (dom->dim == 3)
    hipLaunchKernelGGL (particulate_list_event_kernel<3>, dim3 (grid), dim3 (block), 0, dom->stream,
			A, dom->depth);
  else
    hipLaunchKernelGGL (particulate_list_event_kernel<2>, dim3 (grid), dim3 (block), 0, dom->stream,
			A, dom->depth);
  GFSHIP_HIP (hipGetLastError ());
  // the velocity of this step for the inertial force of the next one (:1003-1012: only for
  // GfsForceInertial objects)
  for (int f = 0; f < pl->nforces; f++)
    if (pl->forces[f] == GFSHIP_FORCE_INERTIAL)
      return store_previous_vel (pl);
  return GFSHIP_OK;
}

int gfship_particles_set_particulate (gfship_particles * pl, const double * vel, const double * mass,
				      const double * volume)
{
  GFSHIP_CHECK (pl && vel && mass && volume, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (!pl->particulate, GFSHIP_EINVAL, "the list already holds particulates");
  size_t m = (size_t) pl->cap, np = (size_t) pl->n;
  GFSHIP_CHECK (pl->n == pl->np0, GFSHIP_EINVAL, "set the particulate state before the first migration");
  std::vector<double> tmp (m, 0.);
  for (int c = 0; c < 3; c++) {
    GFSHIP_HIP (hipMalloc ((void **) &pl->vel[c], m*sizeof (double)));
    GFSHIP_HIP (hipMalloc ((void **) &pl->force[c], m*sizeof (double)));
    for (size_t q = 0; q < np; q++) tmp[q] = vel[3*q + c];
    GFSHIP_HIP (hipMemcpy (pl->vel[c], tmp.data (), m*sizeof (double), hipMemcpyHostToDevice));
    GFSHIP_HIP (hipMemset (pl->force[c], 0, m*sizeof (double)));
  }
  GFSHIP_HIP (hipMalloc ((void **) &pl->mass, m*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &pl->volume, m*sizeof (double)));
  GFSHIP_HIP (hipMalloc ((void **) &pl->dia, m*sizeof (double)));
  if (np) {
    GFSHIP_HIP (hipMemcpy (pl->mass, mass, np*sizeof (double), hipMemcpyHostToDevice));
    GFSHIP_HIP (hipMemcpy (pl->volume, volume, np*sizeof (double), hipMemcpyHostToDevice));
    /* diameter of the sphere of that volume, on the host with the reference's expression
       (compute_drag_force, :552) */
    for (size_t q = 0; q < np; q++)
      tmp[q] = 2.*pow (3.0*volume[q]/4.0/M_PI, 1./3.);
    GFSHIP_HIP (hipMemcpy (pl->dia, tmp.data (), np*sizeof (double), hipMemcpyHostToDevice));
  }
  pl->particulate = true;
  return GFSHIP_OK;
}

int gfship_particles_set_forces (gfship_particles * pl, int nforces, const int * kinds,
				 const double gravity[3])
{
  GFSHIP_CHECK (pl && (nforces == 0 || kinds), GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (pl->particulate, GFSHIP_EINVAL, "forces act on particulates (gfship_particles_set_particulate)");
  GFSHIP_CHECK (nforces >= 0 && nforces <= 8, GFSHIP_EINVAL, "at most 8 forces");
  bool coeff = false;
  for (int f = 0; f < nforces; f++)
    GFSHIP_CHECK (kinds[f] >= GFSHIP_FORCE_INERTIAL && kinds[f] <= GFSHIP_FORCE_BUOY, GFSHIP_EINVAL,
		  "unknown force");
  /* a new list of forces: the compiled GfsFunction of a slot belonged to the force that sat there
     (gfship_particles_set_force_coefficient comes after this call) */
  GFSHIP_HIP (hipStreamSynchronize (pl->dom->stream));
  for (int f = 0; f < 8; f++) {
    gfship::rtc_free (pl->coef_fn[f]);
    pl->coef_fn[f] = nullptr;
    if (pl->coef[f]) (void) hipFree (pl->coef[f]);
    pl->coef[f] = nullptr;
  }
  for (int f = 0; f < nforces; f++) {
    pl->forces[f] = kinds[f];
    if (kinds[f] != GFSHIP_FORCE_BUOY) coeff = true;
  }
  pl->nforces = nforces;
  for (int c = 0; c < 3; c++) pl->gravity[c] = gravity ? gravity[c] : 0.;
  /* every GfsForceCoeff creates Un, Vn, Wn and stores the velocity when it is read (:181-187) */
  if (coeff && pl->uold[0] < 0) {
    for (int c = 0; c < pl->dom->dim; c++) {
      pl->uold[c] = gfship_field_alloc (pl->dom, -1);
      GFSHIP_CHECK (pl->uold[c] >= 0, GFSHIP_ENOMEM, "no field for the previous velocity");
    }
    return store_previous_vel (pl);
  }
  return GFSHIP_OK;
}

int gfship_particles_set_force_coefficient (gfship_particles * pl, int force, const char * function)
{
  GFSHIP_CHECK (pl && function, GFSHIP_EINVAL, "null argument");
  GFSHIP_CHECK (force >= 0 && force < pl->nforces, GFSHIP_EINVAL,
		"force %d: the list has %d forces (gfship_particles_set_forces)", force, pl->nforces);
  GFSHIP_CHECK (pl->forces[force] == GFSHIP_FORCE_ADDEDMASS || pl->forces[force] == GFSHIP_FORCE_LIFT ||
		pl->forces[force] == GFSHIP_FORCE_DRAG, GFSHIP_EINVAL,
		"only GfsForceAddedMass, GfsForceLift and GfsForceDrag read a coefficient");
  gfship::RtcKernel * k = nullptr;
  int r = gfship::rtc_compile_coefficient (pl->dom, function, &k);
  if (r) return r;
  GFSHIP_HIP (hipStreamSynchronize (pl->dom->stream));
  gfship::rtc_free (pl->coef_fn[force]);
  pl->coef_fn[force] = k;
  return GFSHIP_OK;
}

int gfship_particles_download_particulate (gfship_particles * pl, double * vel, double * mass,
					   double * force)
{
  GFSHIP_CHECK (pl && pl->particulate, GFSHIP_EINVAL, "not a list of particulates");
  if (pl->n == 0) return 0;
  gfship_domain * dom = pl->dom;
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  size_t m = pl->n, m0 = pl->n;      /* creation slots = slots in use */
  std::vector<unsigned> orig (m), where (m);
  std::vector<unsigned char> al (m);
  std::vector<double> v[3], f[3], ms (m0);
  GFSHIP_HIP (hipMemcpy (orig.data (), pl->orig, m*sizeof (unsigned), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (al.data (), pl->alive, m, hipMemcpyDeviceToHost));
  for (int c = 0; c < 3; c++) {
    v[c].resize (m0); f[c].resize (m0);
    GFSHIP_HIP (hipMemcpy (v[c].data (), pl->vel[c], m0*sizeof (double), hipMemcpyDeviceToHost));
    GFSHIP_HIP (hipMemcpy (f[c].data (), pl->force[c], m0*sizeof (double), hipMemcpyDeviceToHost));
  }
  GFSHIP_HIP (hipMemcpy (ms.data (), pl->mass, m0*sizeof (double), hipMemcpyDeviceToHost));
  for (size_t q = 0; q < m; q++)
    where[orig[q]] = (unsigned) q;
  int k = 0;
  for (size_t o = 0; o < m; o++)
    if (al[where[o]] == 1) {
      for (int c = 0; c < 3; c++) {
	if (vel) vel[3*(size_t) k + c] = v[c][o];
	if (force) force[3*(size_t) k + c] = f[c][o];
      }
      if (mass) mass[k] = ms[o];
      k++;
    }
  return k;
}

int gfship_particle_list_event (gfship_particles * pl)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  const bool migrates = pl->migrate != nullptr || (pl->dom->comm && pl->dom->has_external);
  if (pl->n == 0 && !migrates) return GFSHIP_OK;
  if (pl->sort_every > 0 &&
      (pl->events_since_sort < 0 || pl->events_since_sort >= pl->sort_every)) {
    int r = gfship_particles_sort (pl);
    if (r) return r;
  }
  if (pl->events_since_sort >= 0)
    pl->events_since_sort++;
  gfship_sim_view v = gfship_sim_view_get (pl->sim);
  gfship_domain * dom = pl->dom;
  PartArgs A;
  A.L = dom->lay[dom->depth];
  for (int d = 0; d < 6; d++) A.side[d] = dom->side[d];
  A.n = pl->n;
  for (int c = 0; c < 3; c++) {
    A.pos[c] = pl->pos[c];
    A.old[c] = pl->old[c];
    A.u[c] = c < dom->dim ? dom->fields[v.u[c]].lev[dom->depth] : nullptr;
  }
  A.alive = pl->alive;
  A.dt = v.dt;
  A.count = pl->d_count;
  A.migrate = migrates;
  int block = 256, grid = (pl->n + block - 1)/block;
  if (pl->n > 0 && pl->particulate && pl->nforces > 0) {
    int r = particulate_event (pl, A, v.visc);
    if (r) return r;
    return migrates ? particles_migrate (pl) : GFSHIP_OK;
  }
  if (pl->n > 0) {
    if (dom->dim == 3)
      hipLaunchKernelGGL (particle_list_event_kernel<3>, dim3 (grid), dim3 (block), 0, dom->stream,
			  A, dom->depth);
    else
      hipLaunchKernelGGL (particle_list_event_kernel<2>, dim3 (grid), dim3 (block), 0, dom->stream,
			  A, dom->depth);
    GFSHIP_HIP (hipGetLastError ());
  }
  if (migrates)
    return particles_migrate (pl);
  return GFSHIP_OK;
}

int gfship_particles_count (gfship_particles * pl)
{
  GFSHIP_CHECK (pl != nullptr, GFSHIP_EINVAL, "null particle list");
  if (pl->n == 0) return 0;
  gfship_domain * dom = pl->dom;
  GFSHIP_HIP (hipMemsetAsync (pl->d_count, 0, sizeof (unsigned), dom->stream));
  int block = 256, grid = (pl->n + block - 1)/block;
  hipLaunchKernelGGL (count_alive_kernel, dim3 (grid), dim3 (block), 0, dom->stream,
		      pl->alive, pl->n, pl->d_count);
  unsigned c = 0;
  GFSHIP_HIP (hipMemcpyAsync (&c, pl->d_count, sizeof (unsigned), hipMemcpyDeviceToHost, dom->stream));
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  return (int) c;
}

int gfship_particles_download (gfship_particles * pl, double * pos, unsigned * id)
{
  GFSHIP_CHECK (pl && pos && id, GFSHIP_EINVAL, "null argument");
  if (pl->n == 0) return 0;
  gfship_domain * dom = pl->dom;
  GFSHIP_HIP (hipStreamSynchronize (dom->stream));
  size_t m = pl->n;
  std::vector<double> x (m), y (m), z (m);
  std::vector<unsigned> ids (m), orig (m), where (m);
  std::vector<unsigned char> al (m);
  GFSHIP_HIP (hipMemcpy (orig.data (), pl->orig, m*sizeof (unsigned), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (x.data (), pl->pos[0], m*sizeof (double), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (y.data (), pl->pos[1], m*sizeof (double), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (z.data (), pl->pos[2], m*sizeof (double), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (ids.data (), pl->id, m*sizeof (unsigned), hipMemcpyDeviceToHost));
  GFSHIP_HIP (hipMemcpy (al.data (), pl->alive, m, hipMemcpyDeviceToHost));
  /* survivors in creation order (the reference's list order), whatever the storage order */
  for (size_t q = 0; q < m; q++)
    where[orig[q]] = (unsigned) q;
  int k = 0;
  for (size_t o = 0; o < m; o++) {
    size_t q = where[o];
    if (al[q] == 1) {
      pos[3*(size_t) k] = x[q]; pos[3*(size_t) k + 1] = y[q];
      pos[3*(size_t) k + 2] = dom->dim == 3 ? z[q] : 0.;
      id[k] = ids[q];
      k++;
    }
  }
  return k;
}

} // extern "C"
