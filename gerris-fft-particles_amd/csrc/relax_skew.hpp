// relax_skew.hpp -- definitions shared by the pipelined exact-order sweep kernels
#pragma once
#include "gfship_internal.hpp"

namespace gfship {

#define SK_T   16            /* tile edge (lines) */
#define SK_NL  (SK_T*SK_T)   /* lines = threads per tile */
#define SK_PAD (2*SK_T - 2)  /* extra rows of a tile: max skew */
#ifndef SK_D
#define SK_D   8             /* prefetch distance (steps) */
#endif
#ifndef SK_DH
#define SK_DH  4             /* prefetch distance of the halo streams (divides SK_D): the lag */
                             /* between neighbouring tiles grows with it                     */
#endif
#define SK_FP  16            /* rows of padding in front of and behind every tile, so that */
                             /* prefetch addresses never need clamping                      */

typedef unsigned long long u64;
#define SK_SENTINEL 0xFFFFFFFFFFFFFFFFull

struct SkewArgs {
  Layout L;
  int ntj;                 // tiles per side
  int RT;                  // rows per tile
  double * us;             // skewed u
  const double * rs;       // skewed rhs
  const double * ds;       // skewed dia (or nullptr)
  double * un;             // natural u (ghost layer + mirrored side cells)
  u64 * hbJ;               // [tile][n + SK_T - 1][SK_T] new values of line a = 15
  u64 * hbK;               // [tile][n + SK_T - 1][SK_T] new values of line b = 15
  const unsigned short * order; // ticket -> tile (anti-diagonal major)
  unsigned * ticket;       // ticket counter (zeroed before the launch)
  unsigned * err;          // set to 1 when a bounded spin gives up
  const u64 * dummy;       // 8 readable bytes for the streams a lane does not need
  u64 * stats;             // optional per-tile { start, end, spins, slow entries } (debug)
};

typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ u64 load_sc1 (const u64 * p)
{
  return __hip_atomic_load ((gu64 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_sc1 (u64 * p, u64 v)
{
  __hip_atomic_store ((gu64 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


} // namespace gfship
