/* go_particles.h -- oracle: GfsParticle list (src/particle.h:34-39). TEST INFRASTRUCTURE ONLY. */
#ifndef GO_PARTICLES_H
#define GO_PARTICLES_H
#include "go_sim.h"

typedef struct {
  int n, cap;
  double * pos, * pos_old;   /* 3 doubles per particle (z = 0 in 2-D) */
  unsigned * id;
  /* particles that left through a GfsBoundaryMpi side d, as sent to the neighbour
     (mpi_send_particle, particulatecommon.c:3218-3222): 7 doubles each (pos, pos_old, id) in
     the neighbour's coordinates, 15 for particulates (+ vel, mass, volume, force) */
  int nout[6];
  double * out[6];
  /* GfsParticulate (modules/particulatecommon.h:35-48): mass, volume, velocity, force; the list's
     forces (GfsParticleForce objects in the order they are applied) */
  int particulate;
  double * vel, * force;     /* 3 doubles per particle */
  double * mass, * volume;
  int nforces, forces[8];    /* GO_FORCE_* */
  /* the GfsFunction of a GfsForceCoeff (particulatecommon.c:166-210): called with the variables
     Rep, Urelp, Vrelp, Wrelp, Pdia as the reference sets them in the particle's cell; NULL: none */
  double (* coefficient[8]) (double rep, double urel, double vrel, double wrel, double pdia);
  double gravity[3];         /* sum of the GfsSource intensities on U, V, W (compute_buoyancy_force) */
  GoField * uold[3];         /* "Un", "Vn", "Wn" of GfsForceCoeff (particulatecommon.c:181-185) */
} GoParticles;

enum { GO_FORCE_INERTIAL = 1, GO_FORCE_ADDEDMASS = 2, GO_FORCE_LIFT = 3, GO_FORCE_DRAG = 4,
       GO_FORCE_BUOY = 5 };

int    go_locate (const GoDomain * dom, const double target[3], int ijk[3]);
double go_interpolate (const GoDomain * dom, const double * v, const int cell[3], const double p[3]);
void   go_advect_point (GoSim * s, double p[3], double dt);
GoParticles * go_particles_new (int np, const double * pos, const unsigned * id);
void   go_particles_destroy (GoParticles * pl);
int    go_particles_count (const GoParticles * pl);
int    go_particles_record_size (const GoParticles * pl);
int    go_particles_outbox (GoParticles * pl, int d, double ** rec);
void   go_particles_clear_outbox (GoParticles * pl);
void   go_particles_append (GoParticles * pl, int n, const double * rec);
double * go_particles_pos (GoParticles * pl);
unsigned * go_particles_id (GoParticles * pl);
void   go_particle_list_event (GoSim * s, GoParticles * pl);
void   go_particles_set_particulate (GoParticles * pl, const double * vel, const double * mass,
				     const double * volume);
void   go_particles_set_forces (GoParticles * pl, GoSim * s, int n, const int * kinds);
void   go_particles_set_gravity (GoParticles * pl, const double g[3]);
void   go_particles_set_coefficient (GoParticles * pl, int force,
				     double (* fn) (double, double, double, double, double));
double * go_particles_vel (GoParticles * pl);
double * go_particles_mass (GoParticles * pl);
double * go_particles_force (GoParticles * pl);
#endif
