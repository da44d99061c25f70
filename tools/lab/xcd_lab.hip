// xcd_lab.hip -- hand-off latency between two workgroups through a global granule, for the cache
// scope bits of the accesses (sc0 / sc1 / nt) and for producer / consumer on the same or on
// different XCDs.   hipcc -O3 --offload-arch=gfx950 tools/lab/xcd_lab.hip -o tools/lab/xcd_lab
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;

template <int MODE> __device__ __forceinline__ u64 ld (const u64 * p)
{
  u64 v;
  if (MODE == 0) asm volatile ("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v" (v) : "v" (p) : "memory");
  if (MODE == 1) asm volatile ("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v" (v) : "v" (p) : "memory");
  if (MODE == 2) asm volatile ("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v" (v) : "v" (p) : "memory");
  if (MODE == 3) asm volatile ("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v" (v) : "v" (p) : "memory");
  return v;
}
template <int MODE> __device__ __forceinline__ void st (u64 * p, u64 v)
{
  if (MODE == 0) asm volatile ("global_store_dwordx2 %0, %1, off" :: "v" (p), "v" (v) : "memory");
  if (MODE == 1) asm volatile ("global_store_dwordx2 %0, %1, off sc0" :: "v" (p), "v" (v) : "memory");
  if (MODE == 2) asm volatile ("global_store_dwordx2 %0, %1, off sc1" :: "v" (p), "v" (v) : "memory");
  if (MODE == 3) asm volatile ("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v" (p), "v" (v) : "memory");
}

__device__ __forceinline__ unsigned xcc_id ()
{
  unsigned v;
  asm volatile ("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s" (v));
  return v & 0xf;
}

// block `prod` ping-pongs with block `cons`; everybody records its XCD
template <int LM, int SM>
__global__ void pingpong (u64 * slots, int prod, int cons, int iters, unsigned * xcd, long long * ticks,
			  unsigned * fail)
{
  if (threadIdx.x == 0) xcd[blockIdx.x] = xcc_id ();
  if (threadIdx.x != 0) return;
  u64 * A = slots, * B = slots + 64;     // different cache lines
  if ((int) blockIdx.x == prod) {
    long long t0 = __builtin_amdgcn_s_memrealtime ();
    for (int i = 1; i <= iters; i++) {
      st<SM> (A, (u64) i);
      unsigned spins = 0;
      while (ld<LM> (B) != (u64) i) if (++spins > 2000000u) { *fail = 1; return; }
    }
    *ticks = __builtin_amdgcn_s_memrealtime () - t0;
  }
  else if ((int) blockIdx.x == cons) {
    for (int i = 1; i <= iters; i++) {
      unsigned spins = 0;
      while (ld<LM> (A) != (u64) i) if (++spins > 2000000u) { *fail = 1; return; }
      st<SM> (B, (u64) i);
    }
  }
}

int main ()
{
  const int nb = 64, iters = 2000;
  u64 * slots; unsigned * xcd, * fail; long long * ticks;
  hipMalloc (&slots, 4096); hipMalloc (&xcd, nb*4); hipMalloc (&fail, 4); hipMalloc (&ticks, 8);
  unsigned hx[nb];
  // find the XCD of every block
  hipMemset (slots, 0, 4096); hipMemset (fail, 0, 4);
  hipLaunchKernelGGL ((pingpong<2, 2>), dim3 (nb), dim3 (64), 0, 0, slots, -1, -1, 0, xcd, ticks, fail);
  hipDeviceSynchronize ();
  hipMemcpy (hx, xcd, nb*4, hipMemcpyDeviceToHost);
  printf ("XCD of blocks 0..15:");
  for (int i = 0; i < 16; i++) printf (" %u", hx[i]);
  printf ("\n");
  int same = -1, other = -1;
  for (int i = 1; i < nb; i++) {
    if (same < 0 && hx[i] == hx[0]) same = i;
    if (other < 0 && hx[i] != hx[0]) other = i;
  }
  printf ("producer block 0 (XCD %u), same-XCD consumer block %d, other-XCD consumer block %d\n", hx[0], same, other);
#define RUN(LM, SM, name)						\
  for (int pass = 0; pass < 2; pass++) {				\
    int cons = pass == 0 ? same : other;				\
    hipMemset (slots, 0, 4096); hipMemset (fail, 0, 4);			\
    hipLaunchKernelGGL ((pingpong<LM, SM>), dim3 (nb), dim3 (64), 0, 0, slots, 0, cons, iters, xcd, ticks, fail); \
    hipDeviceSynchronize ();						\
    long long t; unsigned f;						\
    hipMemcpy (&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy (&f, fail, 4, hipMemcpyDeviceToHost); \
    printf ("%-34s %-10s one-way %7.3f us%s\n", name, pass == 0 ? "same XCD" : "other XCD", \
	    (double) t/100./iters/2., f ? "  (TIMED OUT: not coherent)" : "");	\
  }
  RUN (2, 2, "load sc1 / store sc1");
  RUN (3, 3, "load sc0 sc1 / store sc0 sc1");
  RUN (1, 1, "load sc0 / store sc0");
  RUN (1, 0, "load sc0 / store plain");
  RUN (1, 2, "load sc0 / store sc1");
  RUN (2, 1, "load sc1 / store sc0");
  RUN (2, 0, "load sc1 / store plain");
  return 0;
}
