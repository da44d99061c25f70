// wave_lab.hip -- micro-benchmark: ONE wave per 16x16-line tile of the exact-order sweep.
// Lane (a, q), a = 0..15, q = 0..3, owns the four lines b = 4q..4q+3 and works on
// I = t - a - b at step t: the neighbour a-1 is one DPP row shift away, the neighbour b-1 is the
// thread's own previous result except for b = 4q (one ds_bpermute from lane - 16); no LDS grid and
// no barrier on the chain.  Synthetic data, no inter-tile hand-off.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/lab/wave_lab.hip -o tools/lab/wave_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double div6 (double x)
{
  const double r = 0x1.5555555555555p-3;
  const double q = x*r;
  const double rem = __builtin_fma (- q, 6., x);
  return __builtin_fma (rem, r, q);
}

template <int CTRL>
__device__ __forceinline__ double dpp (double oldv, double v)
{
  const long long b = __double_as_longlong (v), o = __double_as_longlong (oldv);
  const int lo = __builtin_amdgcn_update_dpp ((int) o, (int) b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp ((int) (o >> 32), (int) (b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double (((long long) hi << 32) | (unsigned) lo);
}

__device__ __forceinline__ double bperm (int addr, double v)
{
  const long long b = __double_as_longlong (v);
  const int lo = __builtin_amdgcn_ds_bpermute (addr, (int) b);
  const int hi = __builtin_amdgcn_ds_bpermute (addr, (int) (b >> 32));
  return __longlong_as_double (((long long) hi << 32) | (unsigned) lo);
}

#define ROW_SHR1 0x111
#define ROW_SHL1 0x101

// V bit 0: prefetched global loads (u and rhs, 4 + 4 per thread and step), bit 1: 4 stores,
//   bit 2: halo values through an LDS ring written by a second wave (block of 128)
template <int V, int D>
__global__ void __launch_bounds__(128)
wave_kernel (const double * __restrict__ old, const double * __restrict__ rhs, const double * __restrict__ halo,
	     double * __restrict__ out, int steps, long long * cycles)
{
  __shared__ double ring[16][64];
  __shared__ volatile int produced, consumed;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int a = lane & 15, q = lane >> 4;
  const size_t region = (size_t) blockIdx.x*(size_t) (steps + 64)*256;
  if (threadIdx.x == 0) { produced = 0; consumed = 0; }
  __syncthreads ();
  long long t0 = __builtin_amdgcn_s_memtime ();
  if (wv == 1) {
    if (!(V & 4)) return;
    // halo wave: one value per lane and step into the ring, 4 steps ahead of the consumer at most 12
    const unsigned long long * qH = (const unsigned long long *) halo + lane;
    for (int s = 0; s < steps; s++) {
      while (s - consumed >= 12) __builtin_amdgcn_s_sleep (1);
      unsigned long long h = __hip_atomic_load (qH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      qH += 64;
      ring[s & 15][lane] = __longlong_as_double ((long long) h);
      asm volatile ("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) produced = s + 1;
    }
    return;
  }
  const double * qR = old + region + lane, * qRhs = rhs + region + lane;
  double * w = out + region + lane;
  double pR[D][4], pRhs[D][4];
#pragma unroll
  for (int d = 0; d < D; d++)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      pR[d][k] = (V & 1) ? qR[(d*4 + k)*64] : 1.;
      pRhs[d][k] = (V & 1) ? qRhs[(d*4 + k)*64] : 0.5;
    }
  qR += D*256; qRhs += D*256;
  typedef double d4 __attribute__((ext_vector_type(4)));
  const d4 * qR4 = (const d4 *) (old + region) + lane + D*64, * qRhs4 = (const d4 *) (rhs + region) + lane + D*64;
  d4 * w4 = (d4 *) (out + region) + lane;
  double N[4] = {0., 0., 0., 0.};
  const int up = ((lane - 16) & 63)*4, dn = ((lane + 16) & 63)*4;
  // halo lanes: a == 0 takes J- (new) for its 4 lines, a == 15 J+ (old); q == 0, k == 0 K- (new);
  // q == 3, k == 3 K+ (old).  The ring row holds: [0..15] J- by b, [16..31] J+ by b, [32..47] K- by a, [48..63] K+ by a
  for (int s0 = 0; s0 < steps; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      const int t = s0 + d;
      double hJm[4] = {0., 0., 0., 0.}, hJp[4] = {0., 0., 0., 0.}, hKm = 0., hKp = 0.;
      if (V & 4) {
	while (produced <= t) __builtin_amdgcn_s_sleep (1);
	const double * r = ring[t & 15];
#pragma unroll
	for (int k = 0; k < 4; k++) { hJm[k] = r[4*q + k]; hJp[k] = r[16 + 4*q + k]; }
	hKm = r[32 + a]; hKp = r[48 + a];
	if (lane == 0) consumed = t + 1;
      }
      const double * R1 = pR[(d + 1) % D];
      double top[4], bot[4], front[4], back[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
	top[k] = dpp<ROW_SHR1> (hJm[k], N[k]);
	if (V & 4) top[k] = a == 0 ? hJm[k] : top[k];
	bot[k] = dpp<ROW_SHL1> (hJp[k], R1[k]);
	if (V & 4) bot[k] = a == 15 ? hJp[k] : bot[k];
      }
      front[0] = bperm (up, N[3]);
      if (V & 4) front[0] = q == 0 ? hKm : front[0];
      back[3] = bperm (dn, R1[0]);
      if (V & 4) back[3] = q == 3 ? hKp : back[3];
#pragma unroll
      for (int k = 1; k < 4; k++) { front[k] = N[k - 1]; back[k - 1] = R1[k]; }
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
	double bb = 0.;
	bb += 1.*R1[k]; bb += 1.*N[k]; bb += 1.*top[k]; bb += 1.*bot[k]; bb += 1.*front[k]; bb += 1.*back[k];
	v[k] = div6 (bb - pRhs[d][k]);
      }
      if (V & 8) {
	const d4 r4 = *qR4, h4 = *qRhs4;
#pragma unroll
	for (int k = 0; k < 4; k++) { N[k] = v[k]; pR[d][k] = r4[k]; pRhs[d][k] = h4[k]; }
	d4 o4 = { v[0], v[1], v[2], v[3] };
	if (V & 2) *w4 = o4;
	qR4 += 64; qRhs4 += 64; w4 += 64;
      }
      else {
#pragma unroll
	for (int k = 0; k < 4; k++) {
	  N[k] = v[k];
	  if (V & 1) { pR[d][k] = qR[k*64]; pRhs[d][k] = qRhs[k*64]; }
	  if (V & 2) w[k*64] = v[k];
	}
	qR += 256; qRhs += 256; w += 256;
      }
      asm volatile ("" ::: "memory");     // keep the loads of a step in that step
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime ();
  if (lane == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  out[threadIdx.x] = N[0] + N[1] + N[2] + N[3];
}

int main (int argc, char ** argv)
{
  const int NB = argc > 1 ? atoi (argv[1]) : 1;
  printf ("%d workgroups (one compute wave each, 256 cells per step)\n", NB);
  const int steps = 4096;
  size_t n = (size_t) (steps + 64)*256*(size_t) (NB > 1 ? NB : 2);
  double * old, * rhs, * halo, * out;
  long long * cyc;
  hipMalloc (&old, n*8); hipMalloc (&rhs, n*8); hipMalloc (&halo, ((size_t) 1 << 22)*8); hipMalloc (&out, n*8);
  hipMalloc (&cyc, 8);
  hipMemset (old, 0, n*8); hipMemset (rhs, 0, n*8); hipMemset (out, 0, n*8);
  hipMemset (halo, 0, ((size_t) 1 << 22)*8);
  hipEvent_t e0, e1;
  hipEventCreate (&e0); hipEventCreate (&e1);
  auto report = [&] (const char * name) {
    hipEventSynchronize (e1);
    float ms = 0.f;
    hipEventElapsedTime (&ms, e0, e1);
    long long c;
    hipMemcpy (&c, cyc, 8, hipMemcpyDeviceToHost);
    printf ("%-60s %8.1f ticks/step, kernel %.3f us/step, %.0f GB/s algorithmic\n", name, (double) c/steps,
	    ms*1e3/steps, NB*256.*24./(ms*1e-3/steps)*1e-9);
  };
#define RUN(V, D, name) hipLaunchKernelGGL ((wave_kernel<V, D>), dim3 (NB), dim3 (128), 0, 0, old, rhs, halo, out, steps, cyc); hipEventRecord (e0, 0); hipLaunchKernelGGL ((wave_kernel<V, D>), dim3 (NB), dim3 (128), 0, 0, old, rhs, halo, out, steps, cyc); hipEventRecord (e1, 0); report (name)
  RUN (0, 8, "one wave, DPP/bpermute, no memory");
  RUN (1, 8, "+ prefetched loads (distance 8)");
  RUN (3, 8, "+ loads + stores (distance 8)");
  RUN (3, 16, "+ loads + stores (distance 16)");
  RUN (3, 4, "+ loads + stores (distance 4)");
  RUN (11, 4, "wide: 32-byte loads + stores (distance 4)");
  RUN (11, 8, "wide: 32-byte loads + stores (distance 8)");
  RUN (11, 16, "wide: 32-byte loads + stores (distance 16)");
  RUN (9, 16, "wide: 32-byte loads only (distance 16)");
  RUN (7, 8, "+ loads + stores + halo ring from a second wave (8)");
  RUN (7, 16, "+ loads + stores + halo ring from a second wave (16)");
  return 0;
}
