// rtc.hip -- GfsFunction objects of the particle forces on the device.
//
// A GfsForceCoeff of the reference may carry a GfsFunction (modules/particulatecommon.c:166-210): C
// text of the simulation file -- an expression or a { block } with a return -- that gerris compiles
// with the host compiler and calls per particle with the variables Rep, Urelp, Vrelp, Wrelp, Pdia
// set in the particle's cell (:378-384,478-485,566-573).  Here the same text is compiled for the
// GPU the domain lives on with hipRTC (the runtime-compilation library of ROCm, opened with dlopen:
// a run without such functions never loads it) into a kernel that evaluates the function for every
// particle of the list; the event kernel then reads the coefficients from an array.  Same C
// arithmetic (-ffp-contract=off); the device's libm (pow, exp ...) in place of glibc's.
#include "gfship_internal.hpp"
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace gfship {

struct RtcApi {
  void * handle = nullptr;
  hiprtcResult (* CreateProgram) (hiprtcProgram *, const char *, const char *, int, const char **,
				  const char **) = nullptr;
  hiprtcResult (* CompileProgram) (hiprtcProgram, int, const char **) = nullptr;
  hiprtcResult (* GetProgramLogSize) (hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (* GetProgramLog) (hiprtcProgram, char *) = nullptr;
  hiprtcResult (* GetCodeSize) (hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (* GetCode) (hiprtcProgram, char *) = nullptr;
  hiprtcResult (* DestroyProgram) (hiprtcProgram *) = nullptr;
  const char * (* GetErrorString) (hiprtcResult) = nullptr;
};

static RtcApi g_rtc;

static int rtc_load ()
{
  if (g_rtc.handle) return GFSHIP_OK;
  const char * names[] = { "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so", "libhiprtc.so.7" };
  void * h = nullptr;
  for (const char * nm : names)
    if ((h = dlopen (nm, RTLD_NOW | RTLD_GLOBAL)))
      break;
  GFSHIP_CHECK (h != nullptr, GFSHIP_EUNSUPPORTED,
		"a GfsFunction of a particle force needs hipRTC: cannot open libhiprtc.so: %s", dlerror ());
#define SYM(field, name) do { \
    *(void **) &g_rtc.field = dlsym (h, name); \
    GFSHIP_CHECK (g_rtc.field != nullptr, GFSHIP_EUNSUPPORTED, "libhiprtc: no symbol %s", name); \
  } while (0)
  SYM (CreateProgram, "hiprtcCreateProgram");
  SYM (CompileProgram, "hiprtcCompileProgram");
  SYM (GetProgramLogSize, "hiprtcGetProgramLogSize");
  SYM (GetProgramLog, "hiprtcGetProgramLog");
  SYM (GetCodeSize, "hiprtcGetCodeSize");
  SYM (GetCode, "hiprtcGetCode");
  SYM (DestroyProgram, "hiprtcDestroyProgram");
  SYM (GetErrorString, "hiprtcGetErrorString");
#undef SYM
  g_rtc.handle = h;
  return GFSHIP_OK;
}

struct RtcKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};

void rtc_free (RtcKernel * k)
{
  if (!k) return;
  if (k->mod) (void) hipModuleUnload (k->mod);
  delete k;
}

// the kernel around a GfsFunction of the variables of gfs_force_coeff_read (:189-207) and the time
int rtc_compile_coefficient (gfship_domain * dom, const char * text, RtcKernel ** out)
{
  GFSHIP_CHECK (dom && text && out, GFSHIP_EINVAL, "null argument");
  *out = nullptr;
  int r = rtc_load ();
  if (r) return r;
  std::string t (text);
  size_t a = t.find_first_not_of (" \t\n\r"), b = t.find_last_not_of (" \t\n\r");
  GFSHIP_CHECK (a != std::string::npos, GFSHIP_EINVAL, "empty function");
  t = t.substr (a, b - a + 1);
  const bool block = t[0] == '{';
  std::string src =
    "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n"
    "static __device__ double gfship_f (double Rep, double Urelp, double Vrelp, double Wrelp,\n"
    "                                   double Pdia, double t)\n";
  if (block)
    src += t + "\n";
  else
    src += "{ return (" + t + "); }\n";
  src +=
    "extern \"C\" __global__ void gfship_coeff (int n, const unsigned char * alive,\n"
    "    const double * rep, const double * urel, const double * vrel, const double * wrel,\n"
    "    const double * pdia, double t, double * out)\n"
    "{\n"
    "  int q = blockIdx.x*blockDim.x + threadIdx.x;\n"
    "  if (q >= n || alive[q] != 1) return;\n"
    "  out[q] = gfship_f (rep[q], urel[q], vrel[q], wrel[q], pdia[q], t);\n"
    "}\n";
  hipDeviceProp_t prop;
  GFSHIP_HIP (hipGetDeviceProperties (&prop, dom->device));
  const std::string arch = std::string ("--offload-arch=") + prop.gcnArchName;
  const char * opts[] = { arch.c_str (), "-ffp-contract=off", "-fno-fast-math", "-O2" };
  hiprtcProgram prog;
  hiprtcResult e = g_rtc.CreateProgram (&prog, src.c_str (), "gfs_function.hip", 0, nullptr, nullptr);
  GFSHIP_CHECK (e == HIPRTC_SUCCESS, GFSHIP_EHIP, "hiprtcCreateProgram: %s", g_rtc.GetErrorString (e));
  e = g_rtc.CompileProgram (prog, 4, opts);
  if (e != HIPRTC_SUCCESS) {
    size_t ls = 0;
    std::string log;
    if (g_rtc.GetProgramLogSize (prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
      log.resize (ls);
      (void) g_rtc.GetProgramLog (prog, &log[0]);
    }
    (void) g_rtc.DestroyProgram (&prog);
    set_error ("the function `%s' does not compile for the device:\n%s", text, log.c_str ());
    return GFSHIP_EINVAL;
  }
  size_t cs = 0;
  e = g_rtc.GetCodeSize (prog, &cs);
  std::vector<char> code (cs);
  if (e == HIPRTC_SUCCESS) e = g_rtc.GetCode (prog, code.data ());
  (void) g_rtc.DestroyProgram (&prog);
  GFSHIP_CHECK (e == HIPRTC_SUCCESS, GFSHIP_EHIP, "hiprtcGetCode: %s", g_rtc.GetErrorString (e));
  RtcKernel * k = new RtcKernel;
  hipError_t he = hipModuleLoadData (&k->mod, code.data ());
  if (he == hipSuccess) he = hipModuleGetFunction (&k->fn, k->mod, "gfship_coeff");
  if (he != hipSuccess) {
    rtc_free (k);
    return hip_fail (he, "loading the compiled function", __FILE__, __LINE__);
  }
  *out = k;
  return GFSHIP_OK;
}

int rtc_launch_coefficient (RtcKernel * k, hipStream_t stream, int n, const unsigned char * alive,
			    const double * rep, const double * const rel[3], const double * pdia,
			    double t, double * out)
{
  if (n <= 0) return GFSHIP_OK;
  const double * u = rel[0], * v = rel[1], * w = rel[2];
  void * args[] = { &n, &alive, &rep, &u, &v, &w, &pdia, &t, &out };
  GFSHIP_HIP (hipModuleLaunchKernel (k->fn, (unsigned) ((n + 255)/256), 1, 1, 256, 1, 1, 0, stream, args,
				     nullptr));
  return GFSHIP_OK;
}

} // namespace gfship
