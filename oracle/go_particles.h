/* go_particles.h -- oracle: GfsParticle list (src/particle.h:34-39). TEST INFRASTRUCTURE ONLY. */
#ifndef GO_PARTICLES_H
#define GO_PARTICLES_H
#include "go_sim.h"

typedef struct {
  int n;
  double * pos, * pos_old;   /* 3 doubles per particle (z = 0 in 2-D) */
  unsigned * id;
} GoParticles;

int    go_locate (const GoDomain * dom, const double target[3], int ijk[3]);
double go_interpolate (const GoDomain * dom, const double * v, const int cell[3], const double p[3]);
void   go_advect_point (GoSim * s, double p[3], double dt);
GoParticles * go_particles_new (int np, const double * pos, const unsigned * id);
void   go_particles_destroy (GoParticles * pl);
int    go_particles_count (const GoParticles * pl);
double * go_particles_pos (GoParticles * pl);
unsigned * go_particles_id (GoParticles * pl);
void   go_particle_list_event (GoSim * s, GoParticles * pl);
#endif
