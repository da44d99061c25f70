// step_lab.hip -- micro-benchmark of ONE tile of the exact-order sweep: cycles per hyperplane step
// for variants of the exchange / halo / memory structure (no inter-tile hand-off, synthetic data).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/lab/step_lab.hip -o /tmp/step_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define T16 16
#define XS 17

__device__ __forceinline__ double div6 (double x)
{
  const double r = 0x1.5555555555555p-3;
  const double q = x*r;
  const double rem = __builtin_fma (- q, 6., x);
  return __builtin_fma (rem, r, q);
}

// V: bit 0 = global loads (prefetched) , bit 1 = global store, bit 2 = halo by wave 0 (sc1 loads + LDS),
//    bit 3 = true division instead of div6
template <int V>
__global__ void __launch_bounds__(256)
lds_kernel (const double * __restrict__ old, const double * __restrict__ rhs, const double * __restrict__ halo,
	    double * __restrict__ out, int steps, long long * cycles)
{
  __shared__ double X[2][XS*XS], Y[2][XS*XS];
  const int tid = threadIdx.x, a = tid & 15, b = tid >> 4;
  for (int q = tid; q < 2*XS*XS; q += 256) { (&X[0][0])[q] = 0.; (&Y[0][0])[q] = 0.; }
  __syncthreads ();
  const int iOwnX = (a + 1) + XS*(b + 1), iT = a + XS*(b + 1), iF = (a + 1) + XS*b;
  const int iOwnY = a + XS*b, iBo = (a + 1) + XS*b, iBk = a + XS*(b + 1);
  const bool loader = (tid >> 6) == 0;
  const int g = tid >> 4, m = tid & 15;
  const int xy = g == 0 ? XS*(m + 1) : g == 1 ? (m + 1) : g == 2 ? 16 + XS*m : m + XS*16;
  double * hdst = (g < 2 ? &X[0][0] : &Y[0][0]) + xy;
  const size_t boff = (size_t) (blockIdx.x % 8)*4096*256/8;   // spread the blocks over the buffers
  const double * qR = old + tid + boff % 65536, * qRhs = rhs + tid + boff % 65536, * qH = halo + tid;
  double * w = out + tid + (size_t) blockIdx.x*0;
  double pR[8], pRhs[8], pH[4];
#pragma unroll
  for (int q = 0; q < 8; q++) { pR[q] = (V & 1) ? qR[q*256] : 1.; pRhs[q] = (V & 1) ? qRhs[q*256] : 0.5; }
#pragma unroll
  for (int q = 0; q < 4; q++) pH[q] = (V & 4) ? qH[q*64] : 0.;
  qR += 8*256; qRhs += 8*256; qH += 4*64;
  double prev = 0.;
  long long t0 = __builtin_amdgcn_s_memtime ();
  for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int B = q & 1;
      const double Tn = X[B][iT], Fn = X[B][iF], Bo = Y[B][iBo], Bk = Y[B][iBk];
      double bb = 0.;
      bb += 1.*pR[q]; bb += 1.*prev; bb += 1.*Tn; bb += 1.*Bo; bb += 1.*Fn; bb += 1.*Bk;
      const double x = bb - pRhs[q];
      const double v = (V & 8) ? x/6. : div6 (x);
      prev = v;
      X[B ^ 1][iOwnX] = v;
      Y[B ^ 1][iOwnY] = pR[(q + 1) & 7];
      if (V & 4) {
	if (loader) {
	  double hv = pH[q & 3];
	  bool wv = __double_as_longlong (hv) == -1ll;
	  if (__builtin_expect (__any (wv), 0)) hv = 0.;
	  hdst[(B ^ 1)*XS*XS] = hv;
	  pH[q & 3] = __longlong_as_double (__hip_atomic_load ((const long long *) qH, __ATOMIC_RELAXED,
							      __HIP_MEMORY_SCOPE_AGENT));
	  qH += 64;
	}
      }
      if (V & 1) { pR[q] = *qR; qR += 256; pRhs[q] = *qRhs; qRhs += 256; }
      if (V & 2) { *w = v; w += 256; }
      asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime ();
  if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  out[tid] = prev;
}

// one wave per 8x8 tile... here: 256 threads, exchange inside the wave by DPP / bpermute, LDS only
// across waves with a barrier every step (variant 0) or no cross-wave exchange at all (variant 1:
// lower bound of a fully in-wave scheme)
template <int V>
__global__ void __launch_bounds__(256)
dpp_kernel (const double * __restrict__ old, const double * __restrict__ rhs, double * __restrict__ out,
	    int steps, long long * cycles)
{
  __shared__ double Xw[2][4][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const double * qR = old + tid, * qRhs = rhs + tid;
  double pR[8], pRhs[8];
#pragma unroll
  for (int q = 0; q < 8; q++) { pR[q] = qR[q*256]; pRhs[q] = qRhs[q*256]; }
  qR += 8*256; qRhs += 8*256;
  double prev = 0., Bo = 1., Bk = 1.;
  double * w = out + tid;
  if (tid < 128) (&Xw[0][0][0])[tid] = 0.;
  __syncthreads ();
  long long t0 = __builtin_amdgcn_s_memtime ();
  for (int s0 = 0; s0 < steps; s0 += 8) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      // neighbour a-1: row_shr:1 inside the 16-lane row; lane 0 of each row would take the halo
      int lo = __double2loint (prev), hi = __double2hiint (prev);
      int tlo = __builtin_amdgcn_update_dpp (0, lo, 0x111, 0xf, 0xf, false);
      int thi = __builtin_amdgcn_update_dpp (0, hi, 0x111, 0xf, 0xf, false);
      double Tn = __hiloint2double (thi, tlo);
      // neighbour b-1: 16 lanes down, through the LDS crossbar (no memory, no barrier)
      int flo = __builtin_amdgcn_ds_bpermute (((lane - 16) & 63) << 2, lo);
      int fhi = __builtin_amdgcn_ds_bpermute (((lane - 16) & 63) << 2, hi);
      double Fn = __hiloint2double (fhi, flo);
      if (V == 0) {
	// first row of the wave takes the value of the previous wave's last row
	if (lane < 16) Fn = Xw[q & 1][wv][lane];
      }
      double bb = 0.;
      bb += 1.*pR[q]; bb += 1.*prev; bb += 1.*Tn; bb += 1.*Bo; bb += 1.*Fn; bb += 1.*Bk;
      const double v = div6 (bb - pRhs[q]);
      prev = v;
      Bo = pR[(q + 1) & 7]; Bk = pR[(q + 3) & 7];
      if (V == 0) {
	if (lane >= 48 && wv < 3) Xw[(q & 1) ^ 1][wv + 1][lane - 48] = v;
      }
      pR[q] = *qR; qR += 256; pRhs[q] = *qRhs; qRhs += 256;
      *w = v; w += 256;
      if (V == 0)
	asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime ();
  if (tid == 0) cycles[0] = t1 - t0;
  out[tid] = prev;
}

// closer to the production step: bit 0 divide guard, bit 1 act / I logic and selects,
// bit 2 two predicated sc1 granule stores, bit 3 fifth wave doing the halo (block of 320)
template <int V>
__global__ void __launch_bounds__(320)
lds2_kernel (const double * __restrict__ old, const double * __restrict__ rhs, const double * __restrict__ halo,
	     double * __restrict__ out, int steps, long long * cycles, int n)
{
  __shared__ double X[2][XS*XS], Y[2][XS*XS];
  const int tid0 = threadIdx.x;
  const bool compute = (tid0 >> 6) < 4;
  const int tid = tid0 & 255, a = tid & 15, b = tid >> 4;
  for (int q = tid0; q < 2*XS*XS; q += blockDim.x) { (&X[0][0])[q] = 0.; (&Y[0][0])[q] = 0.; }
  __syncthreads ();
  const int iOwnX = (a + 1) + XS*(b + 1), iT = a + XS*(b + 1), iF = (a + 1) + XS*b;
  const int iOwnY = a + XS*b, iBo = (a + 1) + XS*b, iBk = a + XS*(b + 1);
  const int g = (tid >> 4) & 3, m = tid & 15;
  const int xy = g == 0 ? XS*(m + 1) : g == 1 ? (m + 1) : g == 2 ? 16 + XS*m : m + XS*16;
  double * hdst = (g < 2 ? &X[0][0] : &Y[0][0]) + xy;
  const size_t region = (size_t) blockIdx.x*(size_t) (4096 + 64)*256;   // a private stream per block
  const double * qR = old + region + tid, * qRhs = rhs + region + tid, * qH = halo + tid;
  double * w = out + region + tid;
  unsigned long long * p1 = (unsigned long long *) halo + 70000 + tid, * p2 = p1 + 4096;
  const bool r1 = a == 15 || a == 0, r2 = (b == 15 || b == 0) && r1;
  const int s = a + b;
  long long t0 = __builtin_amdgcn_s_memtime ();
  if (!compute) {
    if (!(V & 8)) return;
    double pH[4];
#pragma unroll
    for (int q = 0; q < 4; q++) pH[q] = qH[q*64];
    qH += 4*64;
    for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
      for (int q = 0; q < 16; q++) {
	double hv = pH[q & 3];
	bool wv = (unsigned) (s0 + q - m) < (unsigned) n && __double_as_longlong (hv) == -1ll;
	if (__builtin_expect (__any (wv), 0)) hv = 0.;
	hdst[((q & 1) ^ 1)*XS*XS] = hv;
	pH[q & 3] = __longlong_as_double (__hip_atomic_load ((const long long *) qH, __ATOMIC_RELAXED,
							    __HIP_MEMORY_SCOPE_AGENT));
	qH += 64;
	asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    return;
  }
  double pR[16], pRhs[16];
#pragma unroll
  for (int q = 0; q < 16; q++) { pR[q] = qR[q*256]; pRhs[q] = qRhs[q*256]; }
  qR += 16*256; qRhs += 16*256;
  double prev = 0., first = 0.;
  const double ghostR = 0.25;
  for (int s0 = 0; s0 < steps; s0 += 16) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int t = s0 + q, I = t - s;
      const bool act = (V & 2) ? (I >= 0 && I < n) : true;
      const int B = q & 1;
      const double Tn = X[B][iT], Fn = X[B][iF], Bo = Y[B][iBo], Bk = Y[B][iBk];
      const double Rv = (V & 2) ? ((I + 1 < n) ? pR[q] : ghostR) : pR[q];
      double bb = 0.;
      bb += 1.*Rv; bb += 1.*prev; bb += 1.*Tn; bb += 1.*Bo; bb += 1.*Fn; bb += 1.*Bk;
      const double x = bb - pRhs[q];
      double v = div6 (x);
      if (V & 1) {
	const double ax = __builtin_fabs (x);
	const bool ok = ax < 0x1p1000 && ax > 0x1p-1000;
	if (__builtin_expect (!__all (ok), 0)) v = x/6.;
      }
      prev = act ? v : prev;
      if (V & 2) first = I == 0 ? v : first;
      X[B ^ 1][iOwnX] = v;
      Y[B ^ 1][iOwnY] = pR[(q + 1) & 15];
      pR[q] = *qR; qR += 256; pRhs[q] = *qRhs; qRhs += 256;
      *w = v; w += 256;
      if (V & 4) {
	const unsigned long long bits = (unsigned long long) __double_as_longlong (v);
	if (act && r1) __hip_atomic_store (p1, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (act && r2) __hip_atomic_store (p2, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	p1 += 16; p2 += 16;
      }
      asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime ();
  if (tid0 == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  out[tid] = prev + first;
}

int main (int argc, char ** argv)
{
  const int NB = argc > 1 ? atoi (argv[1]) : 1;
  printf ("%d workgroups\n", NB);
  const int steps = 4096;
  size_t n = (size_t) (steps + 64)*256*(size_t) (NB > 1 ? NB : 2);
  std::vector<double> h (std::min (n, (size_t) 1 << 22), 1.0);
  double * old, * rhs, * halo, * out;
  long long * cyc;
  hipMalloc (&old, n*8); hipMalloc (&rhs, n*8); hipMalloc (&halo, ((size_t) 1 << 22)*8); hipMalloc (&out, n*8);
  hipMalloc (&cyc, 8);
  hipMemset (old, 0, n*8); hipMemset (rhs, 0, n*8); hipMemset (out, 0, n*8);
  hipMemcpy (halo, h.data (), h.size ()*8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate (&e0); hipEventCreate (&e1);
  auto report = [&] (const char * name) {
    hipEventSynchronize (e1);
    float ms = 0.f;
    hipEventElapsedTime (&ms, e0, e1);
    long long c;
    hipMemcpy (&c, cyc, 8, hipMemcpyDeviceToHost);
    printf ("%-56s %8.1f ticks/step, kernel %.3f us/step\n", name, (double) c/steps, ms*1e3/steps);
  };
#define RUN(K, name) hipLaunchKernelGGL (K, dim3 (NB), dim3 (256), 0, 0, old, rhs, halo, out, steps, cyc); hipEventRecord (e0, 0); hipLaunchKernelGGL (K, dim3 (NB), dim3 (256), 0, 0, old, rhs, halo, out, steps, cyc); hipEventRecord (e1, 0); report (name)
  RUN (lds_kernel<0>, "LDS+barrier, no memory, div6");
  RUN (lds_kernel<8>, "LDS+barrier, no memory, true division");
  RUN (lds_kernel<1>, "LDS+barrier + prefetched loads");
  RUN (lds_kernel<3>, "LDS+barrier + loads + store");
  RUN (lds_kernel<7>, "LDS+barrier + loads + store + wave-0 halo");
  RUN (lds_kernel<15>, "same, true division");
#undef RUN
#define RUN2(V, name) hipLaunchKernelGGL (lds2_kernel<V>, dim3 (NB), dim3 (320), 0, 0, old, rhs, halo, out, steps, cyc, 1 << 30); hipEventRecord (e0, 0); hipLaunchKernelGGL (lds2_kernel<V>, dim3 (NB), dim3 (320), 0, 0, old, rhs, halo, out, steps, cyc, 1 << 30); hipEventRecord (e1, 0); report (name)
  RUN2 (0, "lds2: base (prefetch 16, loads, store, div6)");
  RUN2 (1, "lds2: + divide guard");
  RUN2 (2, "lds2: + act / I logic");
  RUN2 (4, "lds2: + 2 predicated sc1 stores");
  RUN2 (8, "lds2: + fifth wave halo");
  RUN2 (15, "lds2: all");
#define RUN(K, name) hipLaunchKernelGGL (K, dim3 (1), dim3 (256), 0, 0, old, rhs, out, steps, cyc); hipEventRecord (e0, 0); hipLaunchKernelGGL (K, dim3 (1), dim3 (256), 0, 0, old, rhs, out, steps, cyc); hipEventRecord (e1, 0); report (name)
  RUN (dpp_kernel<0>, "DPP/bpermute in wave + LDS row across waves + barrier");
  RUN (dpp_kernel<1>, "DPP/bpermute only (no cross-wave, no barrier)");
  return 0;
}
