// gfsrun.cpp -- `gfship2D` / `gfship3D`: run a Gerris simulation file (.gfs) on libgfship.
//
// The host side of the drop-in: it reads the reference's simulation-file format, builds the
// domain and the simulation through the C ABI of include/gfship.h (nothing else is called),
// and drives the reference's event loop:
//   GfsSimulation   simulation_run     src/simulation.c:432-557
//   GfsPoisson      poisson_run        src/simulation.c:2213-2285
// with the events and outputs the reference's own test cases use (test/poisson, test/lid,
// test/reynolds, test/periodic ...), printed in the reference's formats so that the awk/python
// checks of test/*/*.sh read them unchanged.  Supported: one GfsBox (periodic through self
// edges `1 1 right`), uniform `Refine <int>`, Boundary { BcDirichlet | BcNeumann }, Time,
// ProjectionParams, ApproxProjectionParams, AdvectionParams, Init, SourceDiffusion (constant
// coefficient on U, V, W), VariableTracer, EventStop, EventScript, GModule (ignored: the device
// solver replaces hypre/agmg), OutputTime, OutputProjectionStats, OutputDiffusionStats,
// OutputScalarNorm, OutputScalarSum, OutputScalarStats, OutputErrorNorm, OutputLocation,
// GfsAdvection with a VariableStreamFunction, OutputSimulation (text format), OutputEnergySpectra (`GModule fft`), InitSpectra (`GModule
// turbulence`), GfsParticleList of GfsParticle / GfsParticulate objects with
// GfsForce{Inertial,AddedMass,Lift,Drag,Buoy} (`GModule particulates`; --particles FILE writes the
// lists at the end of the run the way the reference prints them).  Anything else fails loudly with
// the line number.
//
//   gfship2D [-D NAME=VALUE ...] [--device N] [--particles FILE] file.gfs     (gfship3D for three dimensions)
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <functional>
#include <sstream>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include "gfship.h"
#include "gfs_text.hpp"
#include "gfs_snapshot.hpp"
#include "gfs_function.hpp"

using namespace gfs;

namespace {

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) {			\
      fprintf (stderr, "gfship: %s failed: %s\n", #call, gfship_last_error ()); exit (1); } } while (0)

const char * side_name[6] = { "right", "left", "top", "bottom", "front", "back" };

// ----------------------------------------------------------------------------------------------
// variables: the reference's domain->variables list (P, Pmac, U, V[, W], then the ones the file
// adds).  Device variables live in libgfship; the others are host arrays used by Init, outputs
// and EventStop only.
// ----------------------------------------------------------------------------------------------
struct Variable {
  std::string name;
  gfship_field dev = -1;          // -1: host only
  std::vector<double> host;       // (n+2)^dim with ghosts, [k][j][i]
  double host_time = -1.;         // simulation step the host copy of a device variable is from
  std::function<void (Variable &)> derive;   // derived variables (Velocity, Divergence ...)
};

struct Event {                     // GfsEvent, src/event.c:60-169
  double t = 0., start = 0., end = DBL_MAX, step = DBL_MAX;
  unsigned i = 0, istart = 0, iend = INT_MAX, istep = INT_MAX;
  unsigned n = 0;
  bool end_event = false, realised = false, dead = false;
  std::string cls;
  int line = 0;
  std::function<void ()> action;
};

// Shell commands of the file (GfsEventScript) run in a helper process forked at program start, before
// anything touches the GPU: a process that has initialised the device never forks or execs.  The
// helper reads { length, script } records and answers with the status of system ().
struct ShellServer {
  int to = -1, from = -1;
  pid_t pid = -1;
  void start () {
    int a[2], b[2];
    if (pipe (a) != 0 || pipe (b) != 0) return;
    fflush (stdout); fflush (stderr);
    pid = fork ();
    if (pid == 0) {
      ::close (a[1]); ::close (b[0]);
      for (;;) {
	size_t len = 0;
	if (read (a[0], &len, sizeof (len)) != (ssize_t) sizeof (len)) _exit (0);
	std::string script (len, '\0');
	size_t got = 0;
	while (got < len) {
	  ssize_t q = read (a[0], &script[got], len - got);
	  if (q <= 0) _exit (0);
	  got += (size_t) q;
	}
	int status = system (script.c_str ());
	if (write (b[1], &status, sizeof (status)) != (ssize_t) sizeof (status)) _exit (0);
      }
    }
    ::close (a[0]); ::close (b[1]);
    if (pid < 0) { ::close (a[1]); ::close (b[0]); return; }
    to = a[1]; from = b[0];
    fcntl (to, F_SETFD, FD_CLOEXEC); fcntl (from, F_SETFD, FD_CLOEXEC);
  }
  int run (const std::string & script) {
    if (to < 0) return -1;
    size_t len = script.size ();
    int status = -1;
    if (write (to, &len, sizeof (len)) != (ssize_t) sizeof (len) ||
	write (to, script.data (), len) != (ssize_t) len ||
	read (from, &status, sizeof (status)) != (ssize_t) sizeof (status))
      return -1;
    return status;
  }
  void stop () {
    if (to >= 0) { ::close (to); ::close (from); to = from = -1; }
    if (pid > 0) { int st; waitpid (pid, &st, 0); pid = -1; }
  }
};
static ShellServer g_shell;

struct Output {                    // GfsOutput, src/output.c:143-212
  std::string format;             // stdout, stderr, a file name or a { shell script }
  FILE * fp = nullptr;
  bool pipe = false;
  FILE * open () {
    if (fp) return fp;
    if (format == "stdout") fp = stdout;
    else if (format == "stderr") fp = stderr;
    else if (!format.empty () && format[0] == '{') {
      std::string script = format.substr (1, format.size () - 2);
      fflush (stdout);
      fp = popen (script.c_str (), "w");
      pipe = true;
    }
    else
      fp = fopen (format.c_str (), "w");
    if (!fp) { fprintf (stderr, "gfship: cannot open output `%s'\n", format.c_str ()); exit (1); }
    return fp;
  }
  void close () {
    if (!fp) return;
    if (pipe) pclose (fp);
    else if (fp != stdout && fp != stderr) fclose (fp);
    else fflush (fp);
    fp = nullptr;
  }
};

struct BcSpec { int kind = GFSHIP_BC_SYMMETRY; Function * val = nullptr; };

// GfsParticleList (modules/particulatecommon.c:980-1093) of GfsParticle (src/particle.c:46-99) or
// GfsParticulate (:844-905) objects, with the list's GfsParticleForce objects
struct ParticleSpec {
  bool particulate = false;
  std::vector<unsigned> id;
  std::vector<double> pos, vel, mass, volume;    // 3 per particle for pos / vel
  std::vector<int> forces;                       // GFSHIP_FORCE_* in file order
  std::vector<std::string> force_functions;      // the GfsFunction of a GfsForceCoeff ("" = none)
  gfship_particles * pl = nullptr;
};

struct Run {
  int dim = 2, level = 0, device = 0;
  std::string sim_class = "Simulation";
  int side[6] = { GFSHIP_SIDE_BOUNDARY, GFSHIP_SIDE_BOUNDARY, GFSHIP_SIDE_BOUNDARY,
		  GFSHIP_SIDE_BOUNDARY, GFSHIP_SIDE_BOUNDARY, GFSHIP_SIDE_BOUNDARY };
  std::map<std::string, BcSpec> bc[6];       // per side: variable name -> condition
  // GfsTime, src/simulation.c:1660-1670
  double t = 0., end = DBL_MAX, dtmax = DBL_MAX;
  unsigned i = 0, iend = INT_MAX;
  std::map<std::string, std::string> proj_set, approx_set, adv_set;
  double visc[3] = { 0., 0., 0. };
  std::map<std::string, std::string> diff_set[3];
  std::vector<std::string> tracers;
  std::vector<int> tracer_gradient;              // 0 gfs_center_gradient, 1 van Leer (default)
  Function * stream_function = nullptr;          // GfsVariableStreamFunction (2-D, GfsAdvection)
  std::vector<std::pair<std::string, Function *>> init;   // Init {} { var = f }
  double source[3] = { 0., 0., 0. };    // GfsSource intensities on U, V, W
  // GfsPhysicalParams { alpha = f }: the inverse of the density, a function of x, y, z, t and tracers,
  // evaluated on the faces before every step (gfs_function_face_value, src/utils.c:1268-1297)
  Function * alpha = nullptr;
  int alpha_line = 0;
  gfship_field alpha_dev[3] = { -1, -1, -1 };
  bool alpha_static = false;             // no variable and no t in it: evaluated once
  std::vector<Variable> vars;
  std::vector<std::string> device_vars;   // variables of the file that live on the device (GfsVariableTurbulentViscosity)
  std::vector<std::unique_ptr<Event>> events;
  std::vector<std::unique_ptr<Output>> outputs;
  std::vector<std::unique_ptr<ParticleSpec>> plists;
  std::vector<std::pair<gfship_init_spectra_params, std::vector<std::string>>> init_spectra;
  std::string particles_out;                     // --particles FILE: the lists at the end of the run
  Function * refine_fn = nullptr;                // GfsRefine given as a function
  int refine_line = 0;
  // the text of the file, kept to write it back (GfsOutputSimulation): objects of the body, the
  // parameters of the GfsBox, the edges; and the cell data the file came with (a snapshot)
  std::vector<std::pair<std::string, std::string>> object_texts;   // class, text
  std::string box_text, edges_text;
  int nedges = 0;
  gfs::SimulationFile snapshot;
  FunctionSet functions;
  gfship_domain * dom = nullptr;
  gfship_sim * sim = nullptr;
  // a Refine function that asks for a non-uniform tree (one periodic box): gfship_tree.  The
  // host copy of a variable is then one value per leaf, the leaves in the order of
  // ftt_cell_traverse (src/ftt.c:689-926); (leaf_i, leaf_j, leaf_k) are the cell's indices on level
  // leaf_l.  for_each_cell hands the level over in the upper bits of its k (TREE_K)
  bool tree_mode = false;
  gfship_tree * tree = nullptr;
  std::vector<int> leaf_l, leaf_i, leaf_j, leaf_k;
  static int TREE_K (int k, int l) { return k | (l << 16); }
  size_t tree_index (size_t c) const {
    size_t r = (1 << leaf_l[c]) + 2;
    return leaf_i[c] + r*(leaf_j[c] + (dim == 3 ? r*leaf_k[c] : 0));
  }
  size_t tree_level_size (int l) const { size_t r = (1 << l) + 2; return dim == 3 ? r*r*r : r*r; }
  std::chrono::steady_clock::time_point clock0;

  int n () const { return 1 << level; }
  size_t total () const {
    if (tree_mode) return leaf_l.size ();
    size_t r = n () + 2; return dim == 3 ? r*r*r : r*r;
  }
  size_t idx (int i, int j, int k) const {
    size_t r = n () + 2;
    return i + r*(j + (dim == 3 ? r*(size_t) k : 0));
  }
  int var_index (const std::string & name) const {
    for (size_t q = 0; q < vars.size (); q++)
      if (vars[q].name == name) return (int) q;
    return -1;
  }
  int get_or_add_variable (const std::string & name) {
    int v = var_index (name);
    if (v >= 0) return v;
    Variable nv;
    nv.name = name;
    vars.push_back (nv);
    return (int) vars.size () - 1;
  }
  std::vector<std::string> var_names () const {
    std::vector<std::string> r;
    for (const Variable & v : vars) r.push_back (v.name);
    return r;
  }
};

// ----------------------------------------------------------------------------------------------
// host copies of the fields
// ----------------------------------------------------------------------------------------------
std::vector<double> & host_of (Run & R, int v)
{
  Variable & V = R.vars[v];
  if (R.tree_mode && V.dev >= 0) {
    if (V.host.size () != R.total () || V.host_time != (double) R.i) {
      V.host.resize (R.total ());
      int depth = gfship_tree_depth (R.tree);
      std::vector<std::vector<double>> lev (depth + 1);
      for (size_t c = 0; c < R.leaf_l.size (); c++) {
	int l = R.leaf_l[c];
	if (lev[l].empty ()) {
	  lev[l].resize (R.tree_level_size (l));
	  CHECK (gfship_tree_download (R.tree, V.dev, l, lev[l].data ()));
	}
	V.host[c] = lev[l][R.tree_index (c)];
      }
      V.host_time = (double) R.i;
    }
  }
  else if (V.dev >= 0) {
    if (V.host.size () != R.total () || V.host_time != (double) R.i) {
      V.host.resize (R.total ());
      CHECK (gfship_field_download (R.dom, V.dev, R.level, V.host.data ()));
      V.host_time = (double) R.i;
    }
  }
  else if (V.derive) {
    if (V.host.size () != R.total () || V.host_time != (double) R.i) {
      V.host.assign (R.total (), 0.);
      V.derive (V);
      V.host_time = (double) R.i;
    }
  }
  else if (V.host.size () != R.total ())
    V.host.assign (R.total (), 0.);
  return V.host;
}

void invalidate_device_copies (Run & R)
{
  for (Variable & V : R.vars)
    if (V.dev >= 0 || V.derive)
      V.host_time = -1.;
}

void cell_pos (const Run & R, int i, int j, int k, double p[3])
{
  // ftt_cell_pos on the unit box centred on the origin, src/ftt.c:349-367
  if (R.tree_mode) {       /* for_each_cell hands over the level of the leaf in the upper bits of k */
    double h = 1./(1 << (k >> 16));
    p[0] = -0.5 + (i - 0.5)*h;
    p[1] = -0.5 + (j - 0.5)*h;
    p[2] = R.dim == 3 ? -0.5 + ((k & 0xffff) - 0.5)*h : 0.;
    return;
  }
  double h = 1./R.n ();
  p[0] = -0.5 + (i - 0.5)*h;
  p[1] = -0.5 + (j - 0.5)*h;
  p[2] = R.dim == 3 ? -0.5 + (k - 0.5)*h : 0.;
}

double eval (Run & R, const Function * f, const double p[3], long cell)
{
  switch (f->kind) {
  case Function::CONSTANT: return f->val;
  case Function::VARIABLE:
    if (cell < 0) { fprintf (stderr, "gfship: a variable cannot be used in a boundary value\n"); exit (1); }
    return host_of (R, f->var)[cell];
  case Function::COMPILED: {
    double v[32];
    if (f->args.size () > 32) { fprintf (stderr, "gfship: too many variables in a function\n"); exit (1); }
    for (size_t q = 0; q < f->args.size (); q++) {
      if (cell < 0) { fprintf (stderr, "gfship: a variable cannot be used in a boundary value\n"); exit (1); }
      v[q] = host_of (R, f->args[q])[cell];
    }
    return f->fn (p[0], p[1], p[2], R.t, v);
  }
  default: break;
  }
  fprintf (stderr, "gfship: unresolved function\n");
  exit (1);
}

template <class F> void for_each_cell (const Run & R, F f)
{
  if (R.tree_mode) {
    for (size_t c = 0; c < R.leaf_l.size (); c++)
      f (R.leaf_i[c], R.leaf_j[c], Run::TREE_K (R.leaf_k[c], R.leaf_l[c]), c);
    return;
  }
  int n = R.n ();
  for (int k = 1; k <= (R.dim == 3 ? n : 1); k++)
    for (int j = 1; j <= n; j++)
      for (int i = 1; i <= n; i++) {
	int kk = R.dim == 3 ? k : 0;
	f (i, j, kk, R.idx (i, j, kk));
      }
}

// gfs_domain_norm_variable with volume weights, src/domain.c:2197-2232, fluid.c:2139-2171
gfship_norm norm_of (const Run & R, const std::vector<double> & a)
{
  gfship_norm nm = { 0., 0., 0., 0., 0. };
  double h = 1./R.n ();
  double w = R.dim == 3 ? h*h*h : h*h;
  for_each_cell (R, [&] (int, int, int k, size_t c) {
    if (R.tree_mode) { double hl = 1./(1 << (k >> 16)); w = R.dim == 3 ? hl*hl*hl : hl*hl; }  /* gfs_cell_volume of the leaf */
    double val = a[c];
    nm.bias += w*val;
    val = fabs (val);
    if (val > nm.infty) nm.infty = val;
    nm.first += w*val;
    nm.second += w*val*val;
    nm.w += w;
  });
  if (nm.w > 0.) {
    nm.bias /= nm.w;
    nm.first /= nm.w;
    nm.second = sqrt (nm.second/nm.w);
  }
  else
    nm.infty = 0.;
  return nm;
}

// ----------------------------------------------------------------------------------------------
// parsing
// ----------------------------------------------------------------------------------------------
void read_event_params (Reader & r, Event & e)
{
  // gfs_event_read, src/event.c:171-300
  if (r.peek (false) != '{') return;
  auto m = r.assignments ();
  bool set_start = m.count ("start"), set_end = m.count ("end"), set_step = m.count ("step"),
    set_istart = m.count ("istart"), set_iend = m.count ("iend"), set_istep = m.count ("istep");
  for (auto & kv : m)
    if (kv.first != "start" && kv.first != "end" && kv.first != "step" && kv.first != "istart" &&
	kv.first != "iend" && kv.first != "istep")
      r.fail ("unknown event parameter `" + kv.first + "'");
  if (set_end) e.end = atof (m["end"].c_str ());
  if (set_step) e.step = atof (m["step"].c_str ());
  if (set_istart) e.istart = (unsigned) atol (m["istart"].c_str ());
  if (set_iend) e.iend = (unsigned) atol (m["iend"].c_str ());
  if (set_istep) e.istep = (unsigned) atol (m["istep"].c_str ());
  if (set_start) {
    if (m["start"] == "end") {
      e.end_event = true;
      if (set_end || set_step || set_istart || set_iend || set_istep)
	r.fail ("no other parameter can be set for an `end' event");
    }
    else
      e.start = atof (m["start"].c_str ());
  }
  if (set_step && set_istep) r.fail ("step and istep cannot be set simultaneously");
  if (set_step) e.istep = INT_MAX;
  if (set_step && e.step <= 0.) r.fail ("step must be strictly positive");
  if (!set_step && !set_istep && set_end) r.fail ("expecting a number (step or istep)");
  if (set_end && e.end <= e.start) r.fail ("end must be larger than start");
  if (e.start < 0. && set_step) e.start = 0.;
  if (set_start || !set_istart) e.t = e.start;
  else e.t = e.start = DBL_MAX/2.;
  if (!set_istep && !set_step && set_iend) r.fail ("expecting a number (istep or step)");
  if (set_istart && e.iend <= e.istart) r.fail ("iend must be larger than istart");
  if (set_istart || !set_start) e.i = e.istart;
  else e.i = e.istart = INT_MAX/2;
}

void read_multilevel (Reader & r, std::map<std::string, std::string> & set)
{
  // gfs_multilevel_params_read, src/poisson.c:70-126
  static const char * keys[] = { "tolerance", "nrelax", "erelax", "minlevel", "nitermax",
				 "nitermin", "weighted", "beta", "omega", "function", nullptr };
  auto m = r.assignments ();
  for (auto & kv : m) {
    bool ok = false;
    for (const char ** k = keys; *k; k++) ok = ok || kv.first == *k;
    if (!ok) r.fail ("unknown keyword `" + kv.first + "'");
    set[kv.first] = kv.second;
  }
}

void apply_multilevel (gfship_multilevel_params * p, const std::map<std::string, std::string> & m)
{
  for (auto & kv : m) {
    const char * v = kv.second.c_str ();
    if (kv.first == "tolerance") p->tolerance = atof (v);
    else if (kv.first == "nrelax") p->nrelax = (unsigned) atol (v);
    else if (kv.first == "erelax") p->erelax = (unsigned) atol (v);
    else if (kv.first == "minlevel") p->minlevel = (unsigned) atol (v);
    else if (kv.first == "nitermax") p->nitermax = (unsigned) atol (v);
    else if (kv.first == "nitermin") p->nitermin = (unsigned) atol (v);
    else if (kv.first == "weighted") p->weighted = atoi (v);
    else if (kv.first == "beta") p->beta = atof (v);
    else if (kv.first == "omega") p->omega = atof (v);
  }
  if (p->tolerance <= 0. || p->nrelax == 0 || p->erelax == 0 || p->beta < 0.5 || p->beta > 1.) {
    fprintf (stderr, "gfship: invalid multilevel parameters\n");
    exit (1);
  }
}

Output * read_output (Run & R, Reader & r)
{
  R.outputs.emplace_back (new Output);
  Output * o = R.outputs.back ().get ();
  if (r.peek (false) == '{')
    o->format = "{" + r.braces () + "}";
  else
    o->format = r.word (false);
  return o;
}

// the `{ shell script }' outputs are started (popen: a fork) before the device is created; files and
// the standard streams are opened when their event first fires, as in the reference
void open_pipes (Run & R)
{
  for (auto & o : R.outputs)
    if (!o->format.empty () && o->format[0] == '{')
      o->open ();
}

double rate (double a, double b, unsigned n)
{
  if (a > 0. && b > 0. && n > 0) return exp (log (b/a)/n);
  return 0.;
}

void stats_write (const gfship_multilevel_params * par, FILE * fp)
{
  // gfs_multilevel_params_stats_write, src/poisson.c:142-172
  fprintf (fp,
	   "    niter: %4d\n"
	   "    residual.bias:   % 10.3e % 10.3e\n"
	   "    residual.first:  % 10.3e % 10.3e %6.2g\n"
	   "    residual.second: % 10.3e % 10.3e %6.2g\n"
	   "    residual.infty:  % 10.3e % 10.3e %6.2g\n",
	   par->niter, par->residual_before.bias, par->residual.bias,
	   par->residual_before.first, par->residual.first,
	   rate (par->residual.first, par->residual_before.first, par->niter),
	   par->residual_before.second, par->residual.second,
	   rate (par->residual.second, par->residual_before.second, par->niter),
	   par->residual_before.infty, par->residual.infty,
	   rate (par->residual.infty, par->residual_before.infty, par->niter));
}

// the scalar of a GfsOutputScalar: { v = function ... }, src/output.c:1670-1830
struct ScalarSpec { Function * f = nullptr; std::string name; Function * w = nullptr; std::string format; };

ScalarSpec read_scalar (Run & R, Reader & r)
{
  ScalarSpec s;
  int l0 = r.line ();
  Reader b (r.braces (), "simulation file", l0);
  while (!b.eof ()) {
    std::string k = b.word ();
    b.expect ('=');
    if (k == "v") {
      FunctionText t = b.function ();
      s.f = R.functions.add (t, b.line ());
      s.name = t.text;       /* gfs_function_description */
    }
    else if (k == "w")
      s.w = R.functions.add (b.function (), b.line ());
    else if (k == "format")
      s.format = b.word ();
    else if (k == "min" || k == "max" || k == "maxlevel")
      b.word ();
    else
      b.fail ("unsupported GfsOutputScalar keyword `" + k + "'");
  }
  if (!s.f) r.fail ("expecting `v = ...'");
  return s;
}

std::vector<double> scalar_values (Run & R, const ScalarSpec & s)
{
  std::vector<double> a (R.total (), 0.);
  for_each_cell (R, [&] (int i, int j, int k, size_t c) {
    double p[3];
    cell_pos (R, i, j, k, p);
    a[c] = eval (R, s.f, p, (long) c);
  });
  return a;
}

void add_event (Run & R, Event * e, const std::string & cls, int line)
{
  e->cls = cls;
  e->line = line;
  R.events.emplace_back (e);
}

// gfs_simulation_write (src/simulation.c:1343-1361): the file this run was read from, with the
// current GfsTime, without the objects that have done their work (GfsInit, GfsInitSpectra: events
// with t >= end are not written, simulation_write :125-137; GfsRefine: the tree follows, :113-123),
// and the cell data of the variables `list' between the braces of the GfsBox (gfs_box_write
// src/boundary.c:1819-1851).  The time is written with 17 significant digits (the reference prints
// %g, gfs_time_write src/simulation.c:1642-1660, and so cannot restart at exactly the same t).
void write_simulation (Run & R, FILE * fp, const std::vector<int> & list, bool binary)
{
  fprintf (fp, "# Gerris Flow Solver %dD version 1.3.2 (gfship)\n", R.dim);
  fprintf (fp, "1 %d GfsSimulation GfsBox GfsGEdge { version = 120812 variables = ", R.nedges);
  for (size_t q = 0; q < list.size (); q++)
    fprintf (fp, "%s%s", q ? "," : "", R.vars[list[q]].name.c_str ());
  fprintf (fp, " %s} {\n", binary ? "binary = 1 " : "");
  fprintf (fp, "  GfsTime { i = %u t = %.17g ", R.i, R.t);
  if (R.end < DBL_MAX) fprintf (fp, "end = %.17g ", R.end);
  if (R.iend < INT_MAX) fprintf (fp, "iend = %u ", R.iend);
  if (R.dtmax < DBL_MAX) fprintf (fp, "dtmax = %.17g ", R.dtmax);
  fputs ("}\n", fp);
  for (auto & ot : R.object_texts) {
    const std::string & c = ot.first;
    if (c == "Time" || c == "Refine" || c == "Init" || c == "InitSpectra") continue;
    fprintf (fp, "  %s\n", ot.second.c_str ());
  }
  fputs ("}\n", fp);
  // the cell data: device variables straight from the device image; host-only variables (Div of a
  // GfsPoisson file ...) go through a temporary device variable
  std::vector<gfship_field> f, tmp;
  for (int q : list) {
    if (R.vars[q].dev >= 0) f.push_back (R.vars[q].dev);
    else {
      gfship_field t = gfship_field_alloc (R.dom, -1);
      CHECK (t);
      CHECK (gfship_field_upload (R.dom, t, R.level, host_of (R, q).data ()));
      f.push_back (t);
      tmp.push_back (t);
    }
  }
  std::string image (gfship_snapshot_tree_bytes (R.dom, (int) f.size ()), '\0');
  CHECK (gfship_snapshot_tree_write (R.dom, (int) f.size (), f.data (), &image[0], image.size ()));
  for (gfship_field t : tmp) gfship_field_free (R.dom, t);
  size_t leaves = 1;
  for (int l = 0; l < R.level; l++) leaves *= R.dim == 3 ? 8 : 4;
  fprintf (fp, "GfsBox { id = 1 pid = -1 size = %zu x = 0 y = 0 z = 0%s%s } {\n", leaves,
	   R.box_text.empty () ? "" : " ", R.box_text.c_str ());
  if (binary)
    fwrite (image.data (), 1, image.size (), fp);
  else
    gfs::tree_write_text (fp, image, f.size ());
  fputs ("}\n", fp);
  fputs (R.edges_text.c_str (), fp);
}

// the gradient functions of GfsAdvectionParams / GfsVariableTracer (src/advection.c:944-1032):
// gfs_center_gradient, and the limited gradients of src/fluid.c:522-690
int gradient_kind (const std::string & name)
{
  static const char * names[] = { "gfs_center_gradient", "gfs_center_van_leer_gradient",
				  "gfs_center_minmod_gradient", "gfs_center_superbee_gradient",
				  "gfs_center_sweby_gradient" };
  for (int k = 0; k < 5; k++)
    if (name == names[k]) return k;
  return -1;
}

void parse_object (Run & R, Reader & r)
{
  int line = r.line ();
  std::string cls = strip_gfs (r.word ());
  if (cls == "Time") {
    auto m = r.assignments ();
    for (auto & kv : m) {
      const char * v = kv.second.c_str ();
      if (kv.first == "t") R.t = atof (v);
      else if (kv.first == "i") R.i = (unsigned) atol (v);
      else if (kv.first == "end") R.end = atof (v);
      else if (kv.first == "iend") R.iend = (unsigned) atol (v);
      else if (kv.first == "dtmax") R.dtmax = atof (v);
      else r.fail ("unknown GfsTime keyword `" + kv.first + "'");
    }
  }
  else if (cls == "Refine") {
    FunctionText t = r.function ();
    char * endp;
    long l = strtol (t.text.c_str (), &endp, 10);
    if (t.block || *endp != '\0') {
      // GfsRefine with a function of the position (refine_maxlevel, src/refine.c:34-43): accepted
      // when it asks for the same level everywhere (resolve_refine: e.g. test/periodic/periodic.gfs
      // with BOX = 0)
      R.refine_fn = R.functions.add (t, line);
      R.refine_line = line;
    }
    else {
      if (l < 0 || l > GFSHIP_MAXLEVEL)
	r.fail ("Refine: level out of range (got `" + t.text + "')");
      R.level = (int) l;
    }
  }
  else if (cls == "GModule") {
    std::string name = r.word (false);
    if (r.peek (false) == '{') r.braces ();
    if (name == "particulates" || name == "fft" || name == "turbulence")
      ;   /* GfsParticleList / GfsParticulate / GfsForce* are built in (libgfship) */
    else
      fprintf (stderr, "gfship: GModule %s ignored (the Poisson and diffusion solvers are libgfship's)\n",
	       name.c_str ());
  }
  else if (cls == "ApproxProjectionParams") read_multilevel (r, R.approx_set);
  else if (cls == "ProjectionParams") read_multilevel (r, R.proj_set);
  else if (cls == "AdvectionParams") {
    auto m = r.assignments ();
    for (auto & kv : m) {
      if (kv.first == "cfl" || kv.first == "gradient" || kv.first == "gc")
	R.adv_set[kv.first] = kv.second;
      else
	r.fail ("unsupported GfsAdvectionParams keyword `" + kv.first + "'");
    }
  }
  else if (cls == "Init") {
    Event e;
    read_event_params (r, e);
    int l0 = r.line ();
    Reader b (r.braces (), "simulation file", l0);
    while (!b.eof ()) {
      std::string name = b.word ();
      b.expect ('=');
      Function * f = R.functions.add (b.function (), b.line ());
      R.get_or_add_variable (name);
      R.init.emplace_back (name, f);
    }
  }
  else if (cls == "SourceDiffusion" || cls == "SourceViscosity") {
    Event e;
    read_event_params (r, e);
    std::vector<std::string> comps;
    if (cls == "SourceDiffusion") comps.push_back (r.word (false));
    else { comps = { "U", "V" }; if (R.dim == 3) comps.push_back ("W"); }
    FunctionText t = r.function ();
    if (t.block || !Reader::is_number (t.text))
      r.fail ("only a constant diffusion coefficient is supported");
    std::map<std::string, std::string> par;
    if (r.peek (false) == '{') read_multilevel (r, par);
    for (const std::string & v : comps) {
      int c = v == "U" ? 0 : v == "V" ? 1 : (v == "W" && R.dim == 3) ? 2 : -1;
      if (c < 0) r.fail ("diffusion is supported on the velocity components only (got `" + v + "')");
      R.visc[c] = atof (t.text.c_str ());
      R.diff_set[c] = par;
    }
  }
  else if (cls == "Source") {
    // GfsSource [{ event }] U|V|W intensity (src/source.c:405-446): a constant intensity on a velocity
    // component
    Event e;
    if (r.peek (false) == '{') read_event_params (r, e);
    std::string v = r.word (false);
    int c = v == "U" ? 0 : v == "V" ? 1 : (v == "W" && R.dim == 3) ? 2 : -1;
    if (c < 0) r.fail ("GfsSource is supported on the velocity components only (got `" + v + "')");
    FunctionText t = r.function ();
    if (t.block || !Reader::is_number (t.text))
      r.fail ("only a constant intensity is supported");
    R.source[c] += atof (t.text.c_str ());        /* several sources on a variable add up */
  }
  else if (cls == "PhysicalParams") {
    // gfs_physical_params_read, src/simulation.c:1760-1835: g, L constants, alpha a GfsFunction
    int l0 = r.line ();
    Reader b (r.braces (), "simulation file", l0);
    while (!b.eof ()) {
      std::string key = b.word ();
      b.expect ('=');
      if (key == "alpha") {
	R.alpha = R.functions.add (b.function (), b.line ());
	R.alpha_line = b.line ();
      }
      else if (key == "L" || key == "g") {
	FunctionText t = b.function ();
	if (t.block || !Reader::is_number (t.text) || atof (t.text.c_str ()) != 1.)
	  b.fail ("GfsPhysicalParams: only " + key + " = 1 is supported (got `" + t.text + "')");
      }
      else
	b.fail ("unknown keyword `" + key + "'");
    }
  }
  else if (cls == "VariableTracer") {
    std::string name = r.word (false);
    int gradient = 1;
    if (r.peek (false) == '{') {
      auto m = r.assignments ();
      for (auto & kv : m)
	if (kv.first == "gradient" && gradient_kind (kv.second) >= 0) gradient = gradient_kind (kv.second);
	else
	  r.fail ("unsupported GfsVariableTracer parameter " + kv.first + " = " + kv.second);
    }
    R.tracers.push_back (name);
    R.tracer_gradient.push_back (gradient);
    R.get_or_add_variable (name);
  }
  else if (cls == "VariableStreamFunction") {
    // GfsVariableStreamFunction name function (2-D; src/variable.c:905-1120)
    if (R.dim != 2) r.fail ("GfsVariableStreamFunction is 2-D only");
    Event e;
    if (r.peek (false) == '{') read_event_params (r, e);
    std::string name = r.word (false);
    R.get_or_add_variable (name);
    R.stream_function = R.functions.add (r.function (), r.line ());
  }
  else if (cls == "EventStop") {
    // gfs_event_stop_read / _event, src/event.c:1737-1835
    Event * e = new Event;
    read_event_params (r, *e);
    std::string vname = r.word (false);
    double max = r.number ();
    std::string dname;
    if (!r.at_end_of_object ()) dname = r.word (false);
    int v = R.var_index (vname);
    if (v < 0) r.fail ("unknown variable `" + vname + "'");
    int dv = dname.empty () ? -1 : R.get_or_add_variable (dname);
    auto oldv = std::make_shared<std::vector<double>> ();
    auto last = std::make_shared<double> (-1.);
    Run * pr = &R;
    e->action = [pr, v, dv, max, oldv, last] () {
      Run & R = *pr;
      const std::vector<double> & cur = host_of (R, v);
      if (*last >= 0.) {
	std::vector<double> d (R.total (), 0.);
	for_each_cell (R, [&] (int, int, int, size_t c) { d[c] = (*oldv)[c] - cur[c]; });
	gfship_norm nm = norm_of (R, d);
	if (nm.infty <= max)
	  R.end = R.t;
	if (dv >= 0) R.vars[dv].host = d;
      }
      *oldv = cur;
      *last = R.t;
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "ParticleList") {
    // gfs_event_list_read (src/event.c:2446-2506) + gfs_particle_list_read
    // (modules/particulatecommon.c:1022-1093):
    //   GfsParticleList { event } [GfsParticle|GfsParticulate] { objects } [{ forces }] [idlast]
    Event * e = new Event;
    read_event_params (r, *e);
    R.plists.emplace_back (new ParticleSpec);
    ParticleSpec * ps = R.plists.back ().get ();
    auto strip = [] (std::string w) { return w.compare (0, 3, "Gfs") == 0 ? w.substr (3) : w; };
    std::string item;
    if (r.peek (false) != '{') item = strip (r.word (false));
    auto numeric = [] (char c) { return (c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.'; };
    {
      int l0 = r.line ();
      Reader b (r.braces (), "simulation file", l0);
      bool first = true;
      while (!b.eof ()) {
	// every object starts with its class (gfs_event_read, src/event.c:171-196)
	std::string k = strip (b.word ());
	if (k != "Particle" && k != "Particulate")
	  b.fail ("expecting GfsParticle or GfsParticulate in a GfsParticleList (got `" + k + "')");
	if (!item.empty () && k != item)
	  b.fail ("the list holds Gfs" + item + " objects");
	bool particulate = k == "Particulate";
	if (!first && particulate != ps->particulate)
	  b.fail ("mixed lists of GfsParticle and GfsParticulate are not supported");
	ps->particulate = particulate;
	first = false;
	if (b.peek (false) == '{') b.braces ();          /* per-object event parameters: the list's apply */
	ps->id.push_back ((unsigned) b.number ());
	for (int c = 0; c < 3; c++) ps->pos.push_back (b.number ());
	if (particulate) {
	  ps->mass.push_back (b.number ());
	  ps->volume.push_back (b.number ());           /* physical_params.L = 1 */
	  for (int c = 0; c < 3; c++) ps->vel.push_back (b.number ());
	  for (int c = 0; c < 3 && numeric (b.peek (false)); c++)
	    b.number ();                                 /* force: recomputed at every event */
	}
      }
    }
    if (r.peek (false) == '{') {
      int l0 = r.line ();
      Reader b (r.braces (), "simulation file", l0);
      while (!b.eof ()) {
	std::string k = strip (b.word ());
	int kind = k == "ForceInertial" ? GFSHIP_FORCE_INERTIAL : k == "ForceAddedMass" ? GFSHIP_FORCE_ADDEDMASS :
	  k == "ForceLift" ? GFSHIP_FORCE_LIFT : k == "ForceDrag" ? GFSHIP_FORCE_DRAG :
	  k == "ForceBuoy" ? GFSHIP_FORCE_BUOY : 0;
	if (!kind) b.fail ("unsupported GfsParticleForce `" + k + "'");
	// gfs_force_coeff_read, modules/particulatecommon.c:189-207: what follows on the line is the
	// GfsFunction of the coefficient (variables Rep, Urelp, Vrelp, Wrelp, Pdia): compiled for the
	// device when the list is created
	std::string fn;
	char c = b.peek (false);
	if (c != 0 && c != '\n') {
	  if (kind == GFSHIP_FORCE_INERTIAL || kind == GFSHIP_FORCE_BUOY)
	    b.fail ("Gfs" + k + " takes no coefficient");
	  fn = b.function ().text;
	}
	ps->forces.push_back (kind);
	ps->force_functions.push_back (fn);
      }
      if (!ps->forces.empty () && !ps->particulate)
	r.fail ("GfsParticleForce objects act on GfsParticulate objects");
      if (ps->forces.size () > 8) r.fail ("at most 8 forces");
    }
    if (numeric (r.peek (false))) r.number ();           /* idlast */
    e->action = [ps] () {
      if (ps->pl && gfship_particle_list_event (ps->pl) != GFSHIP_OK) {
	fprintf (stderr, "gfship: %s\n", gfship_last_error ());
	exit (1);
      }
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputEnergySpectra") {
    // GfsOutputEnergySpectra (modules/fft.c:1360-1530): file { x0 = .. x1 = .. ... } [level]; the box
    // is transformed whole at the finest level
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    if (r.peek (false) == '{') r.braces ();
    auto digit = [] (char c) { return c >= '0' && c <= '9'; };
    if (digit (r.peek (false))) r.number ();
    Run * pr = &R;
    e->action = [pr, o] () {
      Run & R = *pr;
      int nk = gfship_energy_spectra_bins (R.dom);
      std::vector<double> Ek ((size_t) std::max (nk, 1));
      gfship_field u[3];
      const char * un[3] = { "U", "V", "W" };
      for (int c = 0; c < R.dim; c++) u[c] = R.vars[R.var_index (un[c])].dev;
      double Etot = 0., deltak = 0.;
      if (nk < 0 || gfship_energy_spectra (R.dom, R.dim, u, Ek.data (), &Etot, &deltak) != GFSHIP_OK) {
	fprintf (stderr, "gfship: %s\n", gfship_last_error ());
	exit (1);
      }
      // write_energy_spectra, modules/fft.c:1340-1348
      FILE * fp = o->open ();
      fprintf (fp, "# Total energy = %g \n", Etot);
      fputs ("# 1:k 2:Ek \n", fp);
      for (int i = 1; i < nk; i++)
	fprintf (fp, "%g %g \n", deltak*sqrt ((double) i), Ek[i]);
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputSpectra") {
    // GfsOutputSpectra (modules/fft.c:1101-1226): file v { x0 = .. x1 = .. ... } [level]; the 3-D box is
    // transformed whole at the finest level
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    std::string vname = r.word (false);
    int v = R.var_index (vname);
    if (v < 0) r.fail ("unknown variable `" + vname + "'");
    // { x0 = .. y0 = .. z0 = .. x1 = .. y1 = .. z1 = .. }: a box that is flat in one direction is a plane
    // (realdim == 2, modules/fft.c:1119-1141); anything else is taken as the whole box
    int normal = -1;
    double plane_pos = 0.;
    if (r.peek (false) == '{') {
      auto m = r.assignments ();
      const char * lo[3] = { "x0", "y0", "z0" }, * hi[3] = { "x1", "y1", "z1" };
      for (int c = 0; c < 3; c++)
	if (m.count (lo[c]) && m.count (hi[c]) && atof (m[lo[c]].c_str ()) == atof (m[hi[c]].c_str ())) {
	  if (normal >= 0) r.fail ("GfsOutputSpectra: FFT in 1 dimension is not implemented (modules/fft.c:1142)");
	  normal = c;
	  plane_pos = atof (m[lo[c]].c_str ());
	}
    }
    auto digit = [] (char c) { return c >= '0' && c <= '9'; };
    if (digit (r.peek (false))) r.number ();
    if (R.dim != 3) r.fail ("GfsOutputSpectra: the 3-D box or one of its planes is transformed");
    Run * pr = &R;
    if (normal >= 0) {
      e->action = [pr, o, v, normal, plane_pos] () {
	Run & R = *pr;
	const int N = gfship_output_spectra_side (R.dom);
	if (N <= 0) { fprintf (stderr, "gfship: %s\n", gfship_last_error ()); exit (1); }
	const int nh = N/2 + 1;
	std::vector<double> F ((size_t) 2*N*nh);
	double ks = 0.;
	if (gfship_output_spectra_plane (R.dom, R.vars[v].dev, normal, plane_pos, F.data (), &ks) != GFSHIP_OK) {
	  fprintf (stderr, "gfship: %s\n", gfship_last_error ());
	  exit (1);
	}
	// write_spectra, modules/fft.c:1047-1085 (L = 1): the flat direction first (one point, k = 0), then
	// the two others in coordinate order, the last one halved
	const int ca = normal == 0 ? 1 : 0, cb = normal == 2 ? 1 : 2;
	FILE * fp = o->open ();
	fprintf (fp, "# %i \n", N*N);
	fputs ("# 1:kx 2:ky 3:kz 4:real 5:img\n", fp);
	for (int j = 0; j < N; j++)
	  for (int l = 0; l < nh; l++) {
	    double k[3] = { 0., 0., 0. };
	    k[ca] = ks*(j < nh ? j : j - N);
	    k[cb] = ks*l;
	    const size_t q = 2*((size_t) j*nh + l);
	    fprintf (fp, "%g %g %g %g %g\n", k[0], k[1], k[2], F[q]*1., F[q + 1]*1.);
	  }
	fflush (fp);
      };
      add_event (R, e, cls, line);
      return;
    }
    e->action = [pr, o, v] () {
      Run & R = *pr;
      const int N = gfship_output_spectra_side (R.dom);
      if (N <= 0) { fprintf (stderr, "gfship: %s\n", gfship_last_error ()); exit (1); }
      const int nh = N/2 + 1;
      std::vector<double> F ((size_t) 2*N*N*nh);
      double ks = 0.;
      if (gfship_output_spectra (R.dom, R.vars[v].dev, F.data (), &ks) != GFSHIP_OK) {
	fprintf (stderr, "gfship: %s\n", gfship_last_error ());
	exit (1);
      }
      // write_spectra, modules/fft.c:1047-1085 (L = 1)
      FILE * fp = o->open ();
      fprintf (fp, "# %i \n", N*N*N);
      fputs ("# 1:kx 2:ky 3:kz 4:real 5:img\n", fp);
      for (int i = 0; i < N; i++) {
	const double kx = ks*(i < nh ? i : i - N);
	for (int j = 0; j < N; j++) {
	  const double ky = ks*(j < nh ? j : j - N);
	  for (int l = 0; l < nh; l++) {
	    const size_t q = 2*(((size_t) i*N + j)*nh + l);
	    fprintf (fp, "%g %g %g %g %g\n", kx, ky, ks*l, F[q]*1., F[q + 1]*1.);
	  }
	}
      }
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "VariableTurbulentViscosity") {
    // GfsVariableTurbulentViscosity [{ event }] name Cs (modules/turbulence.c:1068-1084): the Smagorinsky
    // eddy viscosity of the leaf cells, refreshed by its event (default: every step)
    Event * e = new Event;
    if (r.peek (false) == '{') read_event_params (r, *e);
    else { e->istep = 1; e->start = 0.; }
    std::string name = r.word (false);
    const double Cs = r.number ();
    const int v = R.get_or_add_variable (name);
    R.device_vars.push_back (name);
    Run * pr = &R;
    e->action = [pr, v, Cs] () {
      Run & R = *pr;
      gfship_field u[3] = { -1, -1, -1 };
      const char * un[3] = { "U", "V", "W" };
      for (int c = 0; c < R.dim; c++) u[c] = R.vars[R.var_index (un[c])].dev;
      if (R.dim == 2) u[2] = u[1];
      if (gfship_turbulent_viscosity (R.dom, u, Cs, 1, R.vars[v].dev) != GFSHIP_OK) {
	fprintf (stderr, "gfship: %s\n", gfship_last_error ());
	exit (1);
      }
      R.vars[v].host_time = -1.;      /* the host copy, if any, is out of date */
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "InitSpectra") {
    // gfs_init_spectra_read, modules/turbulence.c:277-346:
    //   GfsInitSpectra { event } { x0 y0 z0 L E } { alpha epsilon c1 c2 c3 [ReL kmax seed] } [level] U V W
    Event e;
    read_event_params (r, e);
    gfship_init_spectra_params p;
    memset (&p, 0, sizeof (p));
    p.kmax = DBL_MAX; p.ReL = 0.; p.seed = 0.;
    auto m1 = r.assignments ();
    for (auto & kv : m1) {
      double v = atof (kv.second.c_str ());
      if (kv.first == "x0") p.x0 = v; else if (kv.first == "y0") p.y0 = v; else if (kv.first == "z0") p.z0 = v;
      else if (kv.first == "L") p.L = v; else if (kv.first == "E") p.E = v;
      else r.fail ("unknown GfsInitSpectra keyword `" + kv.first + "'");
    }
    auto m2 = r.assignments ();
    for (auto & kv : m2) {
      double v = atof (kv.second.c_str ());
      if (kv.first == "alpha") p.alpha = v; else if (kv.first == "epsilon") p.epsilon = v;
      else if (kv.first == "c1") p.c1 = v; else if (kv.first == "c2") p.c2 = v; else if (kv.first == "c3") p.c3 = v;
      else if (kv.first == "ReL") p.ReL = v; else if (kv.first == "kmax") p.kmax = v; else if (kv.first == "seed") p.seed = v;
      else r.fail ("unknown GfsInitSpectra keyword `" + kv.first + "'");
    }
    p.level = -1;                                   /* default: the depth of the domain */
    char c0 = r.peek (false);
    if (c0 >= '0' && c0 <= '9') p.level = (int) r.number ();
    std::vector<std::string> names;
    for (int c = 0; c < 3; c++) names.push_back (r.word (false));
    if (R.dim != 3) r.fail ("GfsInitSpectra only works in 3-D (modules/turbulence.c:747)");
    R.init_spectra.emplace_back (p, names);
  }
  else if (cls == "EventScript") {
    Event * e = new Event;
    read_event_params (r, *e);
    std::string script = r.braces ();
    e->action = [script] () {
      fflush (stdout);
      if (g_shell.run (script) != 0)      /* in the helper forked before the device was touched */
	fprintf (stderr, "gfship: EventScript returned a non-zero status\n");
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputTime") {
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    Run * pr = &R;
    e->action = [pr, o] () {
      // time_event, src/output.c:370-392
      Run & R = *pr;
      double real = std::chrono::duration<double> (std::chrono::steady_clock::now () - R.clock0).count ();
      fprintf (o->open (), "step: %7u t: %15.8f dt: %13.6e cpu: %15.8f real: %15.8f\n",
	       R.i, R.t, R.tree ? gfship_tree_dt (R.tree) : R.sim ? gfship_sim_advection_params (R.sim)->dt : 0., real, real);
      fflush (o->fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputProjectionStats" || cls == "OutputDiffusionStats") {
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    Run * pr = &R;
    bool diffusion = cls == "OutputDiffusionStats";
    e->action = [pr, o, diffusion] () {
      Run & R = *pr;
      FILE * fp = o->open ();
      if (diffusion) {
	// gfs_output_diffusion_stats_event, src/output.c:560-600
	const char * names[3] = { "U", "V", "W" };
	for (int c = 0; c < R.dim; c++)
	  if (R.visc[c] != 0.) {
	    fprintf (fp, "%s diffusion\n", names[c]);
	    stats_write (gfship_sim_diffusion_params (R.sim, c), fp);
	  }
      }
      else {
	// projection_stats_event, src/output.c:486-500
	const gfship_multilevel_params * p = R.tree ? gfship_tree_projection_params (R.tree, 0) :
	  gfship_sim_projection_params (R.sim);
	if (p->niter > 0) {
	  fprintf (fp, "MAC projection        before     after       rate\n");
	  stats_write (p, fp);
	}
	fprintf (fp, "Approximate projection\n");
	stats_write (R.tree ? gfship_tree_projection_params (R.tree, 1) :
		     gfship_sim_approx_projection_params (R.sim), fp);
      }
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputScalarNorm" || cls == "OutputScalarSum" || cls == "OutputScalarStats") {
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    ScalarSpec s = read_scalar (R, r);
    Run * pr = &R;
    std::string kind = cls;
    e->action = [pr, o, s, kind] () {
      Run & R = *pr;
      std::vector<double> a = scalar_values (R, s);
      FILE * fp = o->open ();
      if (kind == "OutputScalarNorm") {
	// gfs_output_scalar_norm_event, src/output.c:1966-1986
	gfship_norm nm = norm_of (R, a);
	fprintf (fp, "%s time: %g first: % 10.3e second: % 10.3e infty: % 10.3e\n",
		 s.name.c_str (), R.t, nm.first, nm.second, nm.infty);
      }
      else if (kind == "OutputScalarSum") {
	// gfs_output_scalar_sum_event, src/output.c:2089-2123
	double h = 1./R.n (), vol = R.dim == 3 ? h*h*h : h*h, sum = 0.;
	for_each_cell (R, [&] (int i, int j, int k, size_t c) {
	  double w = vol;
	  if (R.tree_mode) { double hl = 1./(1 << (k >> 16)); w = R.dim == 3 ? hl*hl*hl : hl*hl; }
	  if (s.w) { double p[3]; cell_pos (R, i, j, k, p); w = eval (R, s.w, p, (long) c); }
	  sum += w*a[c];
	});
	if (!s.format.empty ()) {
	  std::string f = "%s time: " + s.format + " sum: " + s.format + "\n";
	  fprintf (fp, f.c_str (), s.name.c_str (), R.t, sum);
	}
	else
	  fprintf (fp, "%s time: %g sum: % 15.6e\n", s.name.c_str (), R.t, sum);
      }
      else {
	// gfs_output_scalar_stats_event, src/output.c:2030-2060 (GtsRange: min, avg, stddev, max)
	double mn = DBL_MAX, mx = -DBL_MAX, sum = 0., sum2 = 0.;
	size_t nn = 0;
	for_each_cell (R, [&] (int, int, int, size_t c) {
	  mn = std::min (mn, a[c]); mx = std::max (mx, a[c]);
	  sum += a[c]; sum2 += a[c]*a[c]; nn++;
	});
	double avg = sum/nn, sd = sqrt (std::max (0., (sum2 - sum*sum/nn)/nn));
	fprintf (fp, "%s time: %g min: %10.3e avg: %10.3e | %10.3e max: %10.3e\n",
		 s.name.c_str (), R.t, mn, avg, sd, mx);
      }
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputErrorNorm") {
    // gfs_output_error_norm_read / _event, src/output.c:2780-3035
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    ScalarSpec s = read_scalar (R, r);
    Function * ref = nullptr, * w = nullptr;
    int unbiased = 0, relative = 0, ev = -1;
    int l0 = r.line ();
    Reader b (r.braces (), "simulation file", l0);
    while (!b.eof ()) {
      std::string k = b.word ();
      b.expect ('=');
      if (k == "s") ref = R.functions.add (b.function (), b.line ());
      else if (k == "w") w = R.functions.add (b.function (), b.line ());
      else if (k == "unbiased") unbiased = atoi (b.word ().c_str ());
      else if (k == "relative") relative = atoi (b.word ().c_str ());
      else if (k == "v") ev = R.get_or_add_variable (b.word ());
      else b.fail ("unknown identifier `" + k + "'");
    }
    if (!ref) r.fail ("expecting `s = ...'");
    if (w) r.fail ("weighted error norms are not supported");
    Run * pr = &R;
    e->action = [pr, o, s, ref, unbiased, relative, ev] () {
      Run & R = *pr;
      std::vector<double> val = scalar_values (R, s), err (R.total (), 0.), sol (R.total (), 0.);
      for_each_cell (R, [&] (int i, int j, int k, size_t c) {
	double p[3];
	cell_pos (R, i, j, k, p);
	sol[c] = eval (R, ref, p, (long) c);
	err[c] = val[c] - sol[c];
      });
      gfship_norm snorm = { 0., 0., 0., 0., 0. };
      if (relative) snorm = norm_of (R, sol);
      gfship_norm nm = norm_of (R, err);
      if (unbiased) {
	for_each_cell (R, [&] (int, int, int, size_t c) { err[c] -= nm.bias; });
	nm = norm_of (R, err);
      }
      if (ev >= 0) { R.vars[ev].host = err; }
      if (relative) {
	if (snorm.first > 0.) nm.first /= snorm.first;
	if (snorm.second > 0.) nm.second /= snorm.second;
	if (snorm.infty > 0.) nm.infty /= snorm.infty;
      }
      FILE * fp = o->open ();
      if (!s.format.empty ()) {
	const std::string & f = s.format;
	std::string fmt = "%s time: " + f + " first: " + f + " second: " + f + " infty: " + f +
	  " bias: " + f + "\n";
	fprintf (fp, fmt.c_str (), s.name.c_str (), R.t, nm.first, nm.second, nm.infty, nm.bias);
      }
      else
	fprintf (fp, "%s time: %g first: %10.3e second: %10.3e infty: %10.3e bias: %10.3e\n",
		 s.name.c_str (), R.t, nm.first, nm.second, nm.infty, nm.bias);
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputLocation") {
    // gfs_output_location_read / _event, src/output.c:1037-1203
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    auto pts = std::make_shared<std::vector<double>> ();
    if (r.peek (false) == '{') {
      std::istringstream in (r.braces ());
      double x;
      while (in >> x) pts->push_back (x);
    }
    else {
      std::string w = r.word (false);
      if (Reader::is_number (w)) {
	pts->push_back (atof (w.c_str ()));
	pts->push_back (r.number ());
	pts->push_back (r.number ());
      }
      else {
	std::ifstream in (w);
	if (!in) r.fail ("cannot open file `" + w + "'");
	double x;
	while (in >> x) pts->push_back (x);
      }
    }
    if (pts->size () % 3) r.fail ("expecting x y z triplets");
    if (r.peek (false) == '{') r.braces ();     /* label, precision: defaults only */
    Run * pr = &R;
    auto first = std::make_shared<bool> (true);
    e->action = [pr, o, pts, first] () {
      Run & R = *pr;
      FILE * fp = o->open ();
      int np = (int) pts->size ()/3;
      if (*first) {
	fputs ("# 1:t 2:x 3:y 4:z", fp);
	int nv = 5;
	for (const Variable & V : R.vars)
	  if (!V.derive) fprintf (fp, " %d:%s", nv++, V.name.c_str ());
	fputc ('\n', fp);
	*first = false;
      }
      std::vector<std::vector<double>> cols;
      std::vector<unsigned char> inside (np, 1);
      for (Variable & V : R.vars) {
	if (V.derive) continue;
	std::vector<double> out (np, 0.);
	gfship_field f = V.dev;
	gfship_field tmp = -1;
	if (f < 0) {          /* host-only variable: sample it through a scratch device field */
	  tmp = gfship_field_alloc (R.dom, -1);
	  CHECK (tmp);
	  size_t vi = &V - &R.vars[0];
	  CHECK (gfship_field_upload (R.dom, tmp, R.level, host_of (R, (int) vi).data ()));
	  CHECK (gfship_bc (R.dom, tmp, tmp, R.level));
	  f = tmp;
	}
	CHECK (gfship_field_interpolate (R.dom, f, np, pts->data (), out.data (), inside.data ()));
	if (tmp >= 0) gfship_field_free (R.dom, tmp);
	cols.push_back (out);
      }
      for (int q = 0; q < np; q++) {
	if (!inside[q]) continue;
	fprintf (fp, "%g %g %g %g", R.t, (*pts)[3*q], (*pts)[3*q + 1], (*pts)[3*q + 2]);
	for (auto & c : cols) fprintf (fp, " %g", c[q]);
	fputc ('\n', fp);
      }
      fflush (fp);
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputSimulation") {
    // gfs_output_simulation (src/output.c:1354-1470, parameters :1480-1555): format = gfs (the
    // default: a simulation file with the cell data, text or `binary = 1') or text (columns)
    Event * e = new Event;
    read_event_params (r, *e);
    Output * o = read_output (R, r);
    std::string format = "gfs", variables;
    bool binary = false;
    if (r.peek (false) == '{') {
      int l0 = r.line ();
      Reader b (r.braces (), "simulation file", l0);
      while (!b.eof ()) {
	std::string k = b.word ();
	b.expect ('=');
	std::string v = b.word ();
	if (k == "format") format = v;
	else if (k == "binary") binary = atoi (v.c_str ()) != 0;
	else if (k == "variables") variables = v;
	else if (k == "depth" || k == "solid" || k == "precision") ;
	else b.fail ("unknown GfsOutputSimulation keyword `" + k + "'");
      }
      if (format != "gfs" && format != "text")
	r.fail ("GfsOutputSimulation: format `" + format + "' is not produced (gfs and text are)");
    }
    Run * pr = &R;
    e->action = [pr, o, format, binary, variables] () {
      Run & R = *pr;
      FILE * fp = o->open ();
      std::vector<int> list;
      if (variables.empty ()) {
	for (size_t q = 0; q < R.vars.size (); q++)
	  if (!R.vars[q].derive) list.push_back ((int) q);
      }
      else
	for (const std::string & nm : gfs::split_commas (variables)) {
	  int q = R.var_index (nm);
	  if (q < 0) { fprintf (stderr, "gfship: GfsOutputSimulation: unknown variable `%s'\n", nm.c_str ()); exit (1); }
	  list.push_back (q);
	}
      if (format == "text") {
	fputs ("# 1:x 2:y 3:z", fp);
	int nv = 4;
	for (int q : list) fprintf (fp, " %d:%s", nv++, R.vars[q].name.c_str ());
	fputc ('\n', fp);
	for_each_cell (R, [&] (int i, int j, int k, size_t c) {
	  double p[3];
	  cell_pos (R, i, j, k, p);
	  fprintf (fp, "%g %g %g", p[0], p[1], p[2]);
	  for (int q : list) fprintf (fp, " %g", host_of (R, q)[c]);
	  fputc ('\n', fp);
	});
      }
      else
	write_simulation (R, fp, list, binary);
      o->close ();
    };
    add_event (R, e, cls, line);
  }
  else if (cls == "OutputPPM" || cls == "OutputGRD" || cls == "OutputTiming" ||
	   cls == "OutputBalance" || cls == "OutputSolidStats" || cls == "OutputAdaptStats") {
    std::string rest = r.rest_of_object ();
    fprintf (stderr, "gfship: line %d: %s is not produced (skipped)\n", line, cls.c_str ());
  }
  else
    r.fail ("unsupported object `" + cls + "'");
  if (!r.at_end_of_object ())
    r.fail ("unexpected text after " + cls);
}

int side_from_name (const std::string & s)
{
  for (int d = 0; d < 6; d++)
    if (s == side_name[d]) return d;
  return -1;
}

void parse_box (Run & R, Reader & r)
{
  // gfs_box_read, src/domain.c:3608-3730
  std::string cls = strip_gfs (r.word ());
  if (cls != "Box") r.fail ("expecting GfsBox");
  int l0 = r.line ();
  const std::string raw = r.braces ();
  Reader b (raw, "simulation file", l0);
  while (!b.eof ()) {
    b.peek ();
    const size_t o0 = b.offset ();
    std::string k = b.word ();
    b.expect ('=');
    int d = side_from_name (k);
    if (d < 0) {
      if (k == "id" || k == "pid" || k == "size" || k == "x" || k == "y" || k == "z") { b.word (); continue; }
      b.fail ("unknown GfsBox keyword `" + k + "'");
    }
    if (d >= 2*R.dim) b.fail ("direction `" + k + "' does not exist in 2-D");
    std::string bcls = strip_gfs (b.word ());
    if (bcls != "Boundary") b.fail ("unsupported boundary class `" + bcls + "'");
    R.side[d] = GFSHIP_SIDE_BOUNDARY;
    if (b.peek (false) == '{') {
      int l1 = b.line ();
      Reader c (b.braces (), "simulation file", l1);
      while (!c.eof ()) {
	// gfs_bc_value_read, src/boundary.c:130-180
	std::string bc = strip_gfs (c.word ());
	BcSpec spec;
	if (bc == "BcDirichlet") spec.kind = GFSHIP_BC_DIRICHLET;
	else if (bc == "BcNeumann") spec.kind = GFSHIP_BC_NEUMANN;
	else c.fail ("unsupported boundary condition `" + bc + "'");
	std::string v = c.word ();
	spec.val = R.functions.add (c.function (), c.line ());
	R.bc[d][v] = spec;
      }
    }
    // kept for GfsOutputSimulation: the boundaries as the file has them (new lines folded)
    std::string piece = raw.substr (o0, b.offset () - o0);
    for (char & ch : piece) if (ch == '\n') ch = ' ';
    R.box_text += (R.box_text.empty () ? "" : " ") + piece;
  }
}

// whole-word -D substitution (the test scripts of the reference do this with sed / m4)
std::string substitute (const std::string & text, const std::map<std::string, std::string> & defs)
{
  if (defs.empty ()) return text;
  std::string out;
  size_t p = 0;
  while (p < text.size ()) {
    if (isalpha ((unsigned char) text[p]) || text[p] == '_') {
      size_t b = p;
      while (p < text.size () && (isalnum ((unsigned char) text[p]) || text[p] == '_')) p++;
      std::string id = text.substr (b, p - b);
      auto it = defs.find (id);
      out += it != defs.end () ? it->second : id;
    }
    else
      out += text[p++];
  }
  return out;
}

void parse_file (Run & R, const std::string & text, const std::string & name)
{
  Reader r (text, name);
  // `nboxes nedges SimClass BoxClass EdgeClass { graph parameters } {`, src/simulation.c:1563-1600
  int nboxes = (int) r.number ();
  int nedges = (int) r.number ();
  R.sim_class = strip_gfs (r.word ());
  r.word (); r.word ();
  if (r.peek () != '{') r.fail ("expecting an opening brace");
  r.braces ();
  if (nboxes != 1)
    r.fail ("one GfsBox per process: multi-box files map to one box per GPU through "
	    "gfship/distributed.py, not through this front end");
  if (R.sim_class != "Simulation" && R.sim_class != "Poisson" && R.sim_class != "Advection")
    r.fail ("unsupported simulation class Gfs" + R.sim_class);
  // default variables of gfs_simulation_init (src/simulation.c:958-985)
  R.get_or_add_variable ("P");
  R.get_or_add_variable ("Pmac");
  R.get_or_add_variable ("U");
  R.get_or_add_variable ("V");
  if (R.dim == 3) R.get_or_add_variable ("W");
  if (R.sim_class == "Poisson") R.get_or_add_variable ("Div");
  {
    int l0 = r.line ();
    const std::string body_text = r.braces ();
    Reader body (body_text, name, l0);
    while (!body.eof ()) {
      const size_t o0 = body.offset ();      /* eof () has skipped the space in front */
      parse_object (R, body);
      std::string text = body_text.substr (o0, body.offset () - o0);
      std::string cls = strip_gfs (Reader (text, name).word ());
      R.object_texts.emplace_back (cls, text);
    }
  }
  parse_box (R, r);
  R.nedges = nedges;
  for (int e = 0; e < nedges; e++) {
    int a = (int) r.number (), b = (int) r.number ();
    std::string dn = r.word ();
    int d = side_from_name (dn);
    if (a != 1 || b != 1 || d < 0 || d >= 2*R.dim || (d & 1))
      r.fail ("expecting a periodic self edge `1 1 right|top|front'");
    R.side[d] = R.side[d + 1] = GFSHIP_SIDE_PERIODIC;
    R.edges_text += "1 1 " + dn + "\n";
  }
}

// ----------------------------------------------------------------------------------------------
// derived variables, src/simulation.c:660-905
// ----------------------------------------------------------------------------------------------
void add_derived (Run & R)
{
  Run * pr = &R;
  auto add = [&] (const std::string & name, std::function<void (Variable &)> f) {
    int v = R.get_or_add_variable (name);
    R.vars[v].derive = f;
  };
  auto vel = [pr] (int c) -> const std::vector<double> & {
    static const char * nm[3] = { "U", "V", "W" };
    return host_of (*pr, pr->var_index (nm[c]));
  };
  add ("Velocity2", [pr, vel] (Variable & V) {
    for_each_cell (*pr, [&] (int, int, int, size_t c) {
      double s = 0.;
      for (int q = 0; q < pr->dim; q++) s += vel (q)[c]*vel (q)[c];
      V.host[c] = s;
    });
  });
  add ("Velocity", [pr, vel] (Variable & V) {
    for_each_cell (*pr, [&] (int, int, int, size_t c) {
      double s = 0.;
      for (int q = 0; q < pr->dim; q++) s += vel (q)[c]*vel (q)[c];
      V.host[c] = sqrt (s);
    });
  });
  add ("Divergence", [pr, vel] (Variable & V) {
    // gfs_divergence: sum of gfs_center_gradient (u_c) / size, src/fluid.c:2356-2372
    double h = 1./pr->n ();
    size_t r = pr->n () + 2, off[3] = { 1, r, r*r };
    for_each_cell (*pr, [&] (int, int, int, size_t c) {
      double div = 0.;
      for (int q = 0; q < pr->dim; q++)
	div += (vel (q)[c + off[q]] - vel (q)[c - off[q]])/2.;
      V.host[c] = div/h;
    });
  });
  if (R.dim == 2)
    add ("Vorticity", [pr, vel] (Variable & V) {
      // gfs_vorticity, src/fluid.c:2391-2402: (dV/dx - dU/dy)/size
      double h = 1./pr->n ();
      size_t r = pr->n () + 2;
      for_each_cell (*pr, [&] (int, int, int, size_t c) {
	V.host[c] = ((vel (1)[c + 1] - vel (1)[c - 1])/2. - (vel (0)[c + r] - vel (0)[c - r])/2.)/h;
      });
    });
}

// ----------------------------------------------------------------------------------------------
// events, src/event.c:60-127,410-459
// ----------------------------------------------------------------------------------------------
bool event_fires (Run & R, Event & e)
{
  if (e.dead) return false;
  if (e.t >= e.end || e.i >= e.iend || R.t > e.end || R.i > e.iend) { e.dead = true; return false; }
  if (e.end_event) {
    if (e.n == 0 && (R.t >= R.end || R.i >= R.iend)) { e.n = 1; return (e.realised = true); }
    return (e.realised = false);
  }
  if (R.t >= e.t) {
    if (e.istep < INT_MAX) {
      if (e.n == 0) { e.i = R.i + e.istep; e.n++; return (e.realised = true); }
    }
    else {
      e.n++;
      e.t = e.start + e.n*e.step;
      return (e.realised = true);
    }
  }
  if (R.i >= e.i) {
    if (e.step < DBL_MAX) {
      if (e.n == 0) { e.start = R.t; e.t = e.start + e.step; e.n = 1; return (e.realised = true); }
    }
    else {
      e.n++;
      e.i += e.istep;
      return (e.realised = true);
    }
  }
  return (e.realised = false);
}

void events_init (Run & R)
{
  for (auto & pe : R.events) {
    Event & e = *pe;
    if (e.end_event) e.t = e.start = DBL_MAX/2.;
    else if (e.istep < INT_MAX)
      while (e.i < R.i) { e.n++; e.i += e.istep; }
    else
      while (e.t < R.t) { e.n++; e.t = e.start + e.n*e.step; }
  }
}

void events_do (Run & R)
{
  invalidate_device_copies (R);
  for (auto & pe : R.events)
    if (event_fires (R, *pe))
      pe->action ();
}

// gfs_event_next, src/event.c:46-71
double event_next (const Event & e, double t, unsigned i)
{
  if (t < e.t) return e.t;
  if (e.t >= e.end || e.i >= e.iend || t > e.end || i > e.iend) return DBL_MAX;
  if (e.end_event) return DBL_MAX;
  if (t >= e.t) {
    if (e.istep < INT_MAX) {
      if (e.n == 0) return DBL_MAX;
    }
    else
      return e.start + (e.n + 1)*e.step;
  }
  if (i >= e.i && e.step < DBL_MAX && e.n == 0)
    return t + e.step;
  return DBL_MAX;
}

// the event loop of gfs_simulation_set_timestep (src/simulation.c:1603-1610), called back by
// libgfship from inside gfship_set_timestep
double next_event_hook (void * ctx, double t, unsigned i)
{
  const Run & R = *(const Run *) ctx;
  double tnext = INT_MAX;
  for (auto & pe : R.events) {
    if (pe->dead) continue;
    double next = event_next (*pe, t, i);
    if (t < next && next < tnext)
      tnext = next + 1e-9;
  }
  return tnext;
}

// ----------------------------------------------------------------------------------------------
// set-up of the device simulation
// ----------------------------------------------------------------------------------------------
void set_boundary_conditions (Run & R)
{
  int n = R.n ();
  size_t nface = R.dim == 3 ? (size_t) n*n : n;
  for (int d = 0; d < 2*R.dim; d++)
    for (auto & kv : R.bc[d]) {
      int v = R.var_index (kv.first);
      if (v < 0 || R.vars[v].dev < 0) {
	fprintf (stderr, "gfship: boundary condition on unknown variable `%s'\n", kv.first.c_str ());
	exit (1);
      }
      // value at the centre of every boundary face, first tangential axis fastest
      std::vector<double> val (nface);
      int c = d/2, ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
      double h = 1./n;
      for (size_t f = 0; f < nface; f++) {
	int ijk[3] = { 1, 1, 1 };
	ijk[c] = (d & 1) ? 1 : n;
	ijk[ta] = (int) (f % n) + 1;
	if (R.dim == 3) ijk[tb] = (int) (f / n) + 1;
	double p[3];
	cell_pos (R, ijk[0], ijk[1], R.dim == 3 ? ijk[2] : 0, p);
	p[c] += (d & 1) ? -h/2. : h/2.;      /* ftt_face_pos */
	val[f] = eval (R, kv.second.val, p, -1);
      }
      CHECK (gfship_field_set_bc (R.dom, R.vars[v].dev, d, kv.second.kind, val.data ()));
    }
}

void apply_init (Run & R)
{
  // gfs_init_event, src/init.c: every `var = function` in file order, on the leaf cells
  for (auto & kv : R.init) {
    int v = R.var_index (kv.first);
    std::vector<double> a = host_of (R, v);     /* copy: the function may read the variable */
    for_each_cell (R, [&] (int i, int j, int k, size_t c) {
      double p[3];
      cell_pos (R, i, j, k, p);
      a[c] = eval (R, kv.second, p, (long) c);
    });
    R.vars[v].host = a;
    if (R.tree_mode && R.vars[v].dev >= 0) {
      int depth = gfship_tree_depth (R.tree);
      for (int l = 0; l <= depth; l++) {
	std::vector<double> lev (R.tree_level_size (l), 0.);
	bool any = false;
	for (size_t c = 0; c < R.leaf_l.size (); c++)
	  if (R.leaf_l[c] == l) { lev[R.tree_index (c)] = a[c]; any = true; }
	if (any)
	  CHECK (gfship_tree_upload (R.tree, R.vars[v].dev, l, lev.data ()));
      }
      R.vars[v].host_time = (double) R.i;
    }
    else if (R.vars[v].dev >= 0) {
      CHECK (gfship_field_upload (R.dom, R.vars[v].dev, R.level, a.data ()));
      R.vars[v].host_time = (double) R.i;
    }
  }
}

// gfs_refine_refine (src/refine.c:45-60): a cell of level l is refined while l < f (x, y, z) at its
// centre.  Only uniform trees exist here: at every level all cells must agree.
void resolve_refine (Run & R)
{
  if (!R.refine_fn) return;
  int level = 0;
  for (;;) {
    const int n = 1 << level;
    int yes = 0, no = 0;
    for (int k = 1; k <= (R.dim == 3 ? n : 1); k++)
      for (int j = 1; j <= n; j++)
	for (int i = 1; i <= n; i++) {
	  double p[3] = { -0.5 + (i - 0.5)/n, -0.5 + (j - 0.5)/n, R.dim == 3 ? -0.5 + (k - 0.5)/n : 0. };
	  if (level < eval (R, R.refine_fn, p, -1)) yes++; else no++;
	}
    if (yes && no) {
      // a statically refined tree: gfship_tree (GfsSimulation, one box, all sides periodic)
      bool periodic = true;
      for (int d = 0; d < 2*R.dim; d++)
	if (R.side[d] != GFSHIP_SIDE_PERIODIC) periodic = false;
      bool sides_ok = true;      /* GfsPoisson: periodic or GfsBoundary sides */
      for (int d = 0; d < 2*R.dim; d++)
	if (R.side[d] != GFSHIP_SIDE_PERIODIC && R.side[d] != GFSHIP_SIDE_BOUNDARY) sides_ok = false;
      (void) periodic;
      if ((R.sim_class == "Simulation" || R.sim_class == "Poisson") && sides_ok) {
	R.tree_mode = true;
	R.level = level;     /* the coarsest leaves */
	return;
      }
      fprintf (stderr, "gfship: line %d: the Refine function asks for a non-uniform tree at level %d "
	       "(refined trees: GfsSimulation or GfsPoisson in one box)\n", R.refine_line, level);
      exit (1);
    }
    if (!yes) break;
    if (++level > GFSHIP_MAXLEVEL) {
      fprintf (stderr, "gfship: line %d: Refine: more than %d levels\n", R.refine_line, GFSHIP_MAXLEVEL);
      exit (1);
    }
  }
  R.level = level;
}

double refine_hook (double x, double y, double z, void * ctx)
{
  Run & R = *(Run *) ctx;
  double p[3] = { x, y, z };
  return eval (R, R.refine_fn, p, -1);
}

// the leaves of the tree in the order of ftt_cell_traverse (pre-order, children 0..3: bit 0 = +x,
// bit 1 = -y, src/ftt.c:301-316)
void tree_leaves (Run & R, const std::vector<std::vector<unsigned char>> & flag, int l, int i, int j, int k)
{
  size_t r = (1 << l) + 2;
  unsigned char f = flag[l][i + r*(j + (R.dim == 3 ? r*k : 0))];
  if (f == 1) {
    R.leaf_l.push_back (l); R.leaf_i.push_back (i); R.leaf_j.push_back (j); R.leaf_k.push_back (k);
  }
  else if (f == 2)
    for (int c = 0; c < (1 << R.dim); c++)
      tree_leaves (R, flag, l + 1, 2*i - 1 + (c & 1), 2*j - ((c >> 1) & 1), 2*k - ((c >> 2) & 1));
}

// poisson_run (src/simulation.c:2213-2285) on a statically refined tree, sides periodic or GfsBoundary
// the values of the conditions of variable `name' on the GfsBoundary sides of a tree: the GfsFunction at the
// centre of the face of every leaf ghost cell (gfs_function_face_value at ftt_face_pos), uploaded to `var'
int upload_tree_bc_values (Run & R, const std::vector<std::vector<unsigned char>> & flag, const std::string & name, int var)
{
  const int depth = gfship_tree_depth (R.tree);
  for (int l = 0; l <= depth; l++) {
    const int n = 1 << l, r = n + 2;
    const double h = 1./n;
    std::vector<double> val (R.tree_level_size (l), 0.);
    bool any = false;
    for (int d = 0; d < 2*R.dim; d++) {
      if (R.side[d] != GFSHIP_SIDE_BOUNDARY || !R.bc[d].count (name) || !R.bc[d][name].val) continue;
      const int c = d/2, ta = c == 0 ? 1 : 0, tb = c == 2 ? 1 : 2;
      for (int b = 1; b <= (R.dim == 3 ? n : 1); b++)
	for (int a = 1; a <= n; a++) {
	  int g[3] = { 0, 0, 0 };
	  g[c] = (d & 1) ? 0 : n + 1;
	  g[ta] = a;
	  g[tb] = R.dim == 3 ? b : 0;
	  size_t G = g[0] + (size_t) r*(g[1] + (R.dim == 3 ? (size_t) r*g[2] : 0));
	  if (flag[l][G] != 1) continue;          /* leaf ghost cells only */
	  double p[3] = { -0.5 + (g[0] - 0.5)*h, -0.5 + (g[1] - 0.5)*h, R.dim == 3 ? -0.5 + (g[2] - 0.5)*h : 0. };
	  p[c] = (d & 1) ? -0.5 : 0.5;            /* the face between the ghost cell and the box */
	  val[G] = eval (R, R.bc[d][name].val, p, -1);
	  any = true;
	}
    }
    if (any)
      CHECK (gfship_tree_upload (R.tree, var, l, val.data ()));
  }
  return 0;
}

int run_tree_poisson (Run & R)
{
  static const char * ok[] = { "OutputErrorNorm", "OutputScalarNorm", "OutputScalarSum", "OutputScalarStats",
			       "OutputTime", "OutputProjectionStats", "EventScript" };
  for (auto & e : R.events) {
    bool found = false;
    for (const char * c : ok) if (e->cls == c) found = true;
    if (e->cls == "OutputSimulation") {
      fprintf (stderr, "gfship: line %d: GfsOutputSimulation is not written for a refined tree (skipped)\n", e->line);
      e->action = [] () {};
      found = true;
    }
    if (!found) {
      fprintf (stderr, "gfship: line %d: Gfs%s is not supported on a refined tree\n", e->line, e->cls.c_str ());
      return 1;
    }
  }
  open_pipes (R);
  CHECK (gfship_tree_create_sides (&R.tree, R.dim, refine_hook, &R, R.side, R.device));
  int depth = gfship_tree_depth (R.tree);
  std::vector<std::vector<unsigned char>> flag (depth + 1);
  for (int l = 0; l <= depth; l++) {
    flag[l].resize (R.tree_level_size (l));
    CHECK (gfship_tree_flags (R.tree, l, flag[l].data ()));
  }
  tree_leaves (R, flag, 0, 1, 1, 1);
  R.vars[R.var_index ("P")].dev = GFSHIP_TREE_P;
  // the conditions of P: kind per side, value of the GfsFunction at the centre of the face of every
  // leaf ghost cell (gfs_function_face_value at ftt_face_pos)
  bool dirichlet = false;
  for (int d = 0; d < 2*R.dim; d++) {
    if (R.side[d] != GFSHIP_SIDE_BOUNDARY) continue;
    for (auto & kv : R.bc[d])
      if (kv.first != "P") {
	fprintf (stderr, "gfship: a boundary condition on `%s' is not supported on a refined tree\n", kv.first.c_str ());
	return 1;
      }
    int kind = R.bc[d].count ("P") ? R.bc[d]["P"].kind : GFSHIP_BC_SYMMETRY;
    if (kind == GFSHIP_BC_DIRICHLET) dirichlet = true;
    CHECK (gfship_tree_set_bc (R.tree, d, kind));
  }
  if (upload_tree_bc_values (R, flag, "P", GFSHIP_TREE_BCVAL)) return 1;
  apply_multilevel (gfship_tree_projection_params (R.tree, 1), R.approx_set);
  apply_init (R);
  events_init (R);
  gfship_multilevel_params * par = gfship_tree_projection_params (R.tree, 1);
  while (R.i < R.iend && R.t < R.end) {
    {
      // correct_div, src/simulation.c:2170-2190: div = Div size^2 on the leaves; without a Dirichlet
      // condition its mean is removed (GtsRange means: sums in traversal order over the leaves)
      const std::vector<double> & Div = host_of (R, R.var_index ("Div"));
      std::vector<double> div (R.total (), 0.);
      double sum = 0., vol = 0.;
      for (size_t c = 0; c < R.leaf_l.size (); c++) {
	double size = 1./(1 << R.leaf_l[c]);
	double a = size*size*1.;
	div[c] = Div[c]*a;
	vol += a;
      }
      if (!dirichlet) {
	for (size_t c = 0; c < R.leaf_l.size (); c++) sum += div[c];
	double nn = (double) R.leaf_l.size ();
	double ddiv = - (sum/nn)/(vol/nn);
	for (size_t c = 0; c < R.leaf_l.size (); c++) {
	  double size = 1./(1 << R.leaf_l[c]);
	  div[c] += size*size*ddiv*1.;
	}
      }
      for (int l = 0; l <= depth; l++) {
	std::vector<double> lev (R.tree_level_size (l), 0.);
	bool any = false;
	for (size_t c = 0; c < R.leaf_l.size (); c++)
	  if (R.leaf_l[c] == l) { lev[R.tree_index (c)] = div[c]; any = true; }
	if (any)
	  CHECK (gfship_tree_upload (R.tree, GFSHIP_TREE_DIV, l, lev.data ()));
      }
    }
    CHECK (gfship_tree_poisson_solve (R.tree, par, 1.));
    R.t = 0.;
    R.i++;
    events_do (R);
  }
  for (auto & o : R.outputs) o->close ();
  gfship_tree_destroy (R.tree);
  R.tree = nullptr;
  return 0;
}

// simulation_run (src/simulation.c:432-557) on a statically refined tree
int run_tree (Run & R)
{
  if (R.sim_class == "Poisson")
    return run_tree_poisson (R);
  auto refuse = [] (const char * what) {
    fprintf (stderr, "gfship: %s is not supported on a refined tree\n", what);
    return 1;
  };
  if (R.tracers.size () > 2) return refuse ("more than two tracers");
  for (int g : R.tracer_gradient)
    if (g > 1) return refuse ("this gradient of a GfsVariableTracer");
  if (!R.plists.empty ()) return refuse ("a particle list");
  if (!R.init_spectra.empty ()) return refuse ("GfsInitSpectra");
  if (!R.device_vars.empty ()) return refuse ("a turbulent-viscosity variable");
  if (R.snapshot.has_tree) return refuse ("cell data in the simulation file");
  if (R.dtmax != DBL_MAX) return refuse ("Time { dtmax }");
  for (int c = 0; c < 3; c++) {
    if (R.visc[c] != 0. && R.dim == 3) return refuse ("GfsSourceDiffusion (octrees)");
  }
  for (auto & kv : R.adv_set)
    if (kv.first != "cfl" && !(kv.first == "gradient" && kv.second == "gfs_center_gradient") &&
	!(kv.first == "gc" && atoi (kv.second.c_str ()) == 1))
      return refuse ("this AdvectionParams setting");
  static const char * ok[] = { "OutputErrorNorm", "OutputScalarNorm", "OutputScalarSum", "OutputScalarStats",
			       "OutputTime", "OutputProjectionStats", "EventScript", "EventStop" };
  for (auto & e : R.events) {
    bool found = false;
    for (const char * c : ok) if (e->cls == c) found = true;
    if (!found) {
      fprintf (stderr, "gfship: line %d: Gfs%s is not supported on a refined tree\n", e->line, e->cls.c_str ());
      return 1;
    }
  }
  {
    int v = R.var_index ("Vorticity");
    if (v >= 0)
      R.vars[v].derive = [] (Variable &) {
	fprintf (stderr, "gfship: the variable Vorticity is not supported on a refined tree\n");
	exit (1);
      };
    v = R.var_index ("Divergence");
    Run * pr = &R;
    if (v >= 0)      /* gfs_divergence on the device, then one value per leaf */
      R.vars[v].derive = [pr] (Variable & V) {
	Run & R = *pr;
	CHECK (gfship_tree_divergence (R.tree));
	std::vector<std::vector<double>> lev (gfship_tree_depth (R.tree) + 1);
	for (size_t c = 0; c < R.leaf_l.size (); c++) {
	  int l = R.leaf_l[c];
	  if (lev[l].empty ()) {
	    lev[l].resize (R.tree_level_size (l));
	    CHECK (gfship_tree_download (R.tree, GFSHIP_TREE_DIV, l, lev[l].data ()));
	  }
	  V.host[c] = lev[l][R.tree_index (c)];
	}
      };
  }
  static const char * vel[3] = { "U", "V", "W" };
  for (int d = 0; d < 2*R.dim; d++)
    for (auto & kv : R.bc[d]) {     /* GfsBoundary sides: conditions on P and on the velocity components */
      bool known = kv.first == "P";
      for (int c = 0; c < R.dim; c++) if (kv.first == vel[c]) known = true;
      if (!known) {
	fprintf (stderr, "gfship: a boundary condition on `%s' is not supported on a refined tree (side %s)\n",
		 kv.first.c_str (), side_name[d]);
	return 1;
      }
    }
  open_pipes (R);
  CHECK (gfship_tree_create_sides (&R.tree, R.dim, refine_hook, &R, R.side, R.device));
  int depth = gfship_tree_depth (R.tree);
  std::vector<std::vector<unsigned char>> flag (depth + 1);
  for (int l = 0; l <= depth; l++) {
    flag[l].resize (R.tree_level_size (l));
    CHECK (gfship_tree_flags (R.tree, l, flag[l].data ()));
  }
  tree_leaves (R, flag, 0, 1, 1, 1);
  R.vars[R.var_index ("P")].dev = GFSHIP_TREE_P;
  R.vars[R.var_index ("Pmac")].dev = GFSHIP_TREE_PMAC;
  R.vars[R.var_index ("U")].dev = GFSHIP_TREE_U;
  R.vars[R.var_index ("V")].dev = GFSHIP_TREE_V;
  if (R.dim == 3)
    R.vars[R.var_index ("W")].dev = GFSHIP_TREE_W;
  for (size_t k = 0; k < R.tracers.size (); k++) {
    int v = gfship_tree_add_tracer (R.tree, R.tracer_gradient[k]);
    CHECK (v);
    R.vars[R.var_index (R.tracers[k])].dev = v;
  }
  for (int d = 0; d < 2*R.dim; d++) {
    if (R.side[d] != GFSHIP_SIDE_BOUNDARY) continue;
    if (R.bc[d].count ("P")) CHECK (gfship_tree_set_bc (R.tree, d, R.bc[d]["P"].kind));
    for (int c = 0; c < R.dim; c++)
      if (R.bc[d].count (vel[c])) CHECK (gfship_tree_set_bc_u (R.tree, c, d, R.bc[d][vel[c]].kind));
  }
  if (upload_tree_bc_values (R, flag, "P", GFSHIP_TREE_BCVAL)) return 1;
  for (int c = 0; c < R.dim; c++)
    if (upload_tree_bc_values (R, flag, vel[c], GFSHIP_TREE_BCU + c)) return 1;
  for (int c = 0; c < R.dim; c++)
    if (R.source[c] != 0.)
      CHECK (gfship_tree_set_source (R.tree, c, R.source[c]));
  for (int c = 0; c < R.dim; c++)
    if (R.visc[c] != 0.) {
      CHECK (gfship_tree_set_viscosity (R.tree, c, R.visc[c]));
      apply_multilevel (gfship_tree_diffusion_params (R.tree, c), R.diff_set[c]);
    }
  apply_multilevel (gfship_tree_projection_params (R.tree, 0), R.proj_set);
  apply_multilevel (gfship_tree_projection_params (R.tree, 1), R.approx_set);
  double cfl = 0.8;        /* gfs_advection_params_init, src/advection.c:922-942 */
  for (auto & kv : R.adv_set)
    if (kv.first == "cfl") cfl = atof (kv.second.c_str ());
  apply_init (R);
  events_init (R);
  CHECK (gfship_tree_set_time (R.tree, R.end, cfl));
  CHECK (gfship_tree_set_next_event (R.tree, next_event_hook, &R));
  CHECK (gfship_tree_start (R.tree));
  while (R.t < R.end && R.i < R.iend) {
    events_do (R);
    CHECK (gfship_tree_set_time (R.tree, R.end, cfl));
    CHECK (gfship_tree_step (R.tree));
    R.t = gfship_tree_time (R.tree);
    R.i = gfship_tree_iter (R.tree);
  }
  events_do (R);
  for (auto & o : R.outputs) o->close ();
  gfship_tree_destroy (R.tree);
  R.tree = nullptr;
  return 0;
}

// GfsPhysicalParams { alpha }: gfs_function_face_value (alpha, face) of every leaf face -- the function at
// the centre of the face (ftt_face_pos), the variables it names interpolated onto the face
// (gfs_face_interpolated_value, src/fluid.c:2186-2198: ((x1 - 0.5) v0 + 0.5 v1)/x1 with x1 = 1 between
// cells of one level) -- in the layout gfship_poisson_coefficients_alpha takes: the entry of a cell is
// its + face along c, the ghost entry in front of the first cell its - face.  Both projections of a
// step evaluate alpha with the same time and the same tracers (gfs_advance_tracers comes last in the
// loop body, src/simulation.c:479-548): once before every step.
int refresh_alpha (Run & R)
{
  if (!R.alpha || (R.alpha_static && R.alpha_dev[0] >= 0))
    return 0;
  const Function * f = R.alpha;
  const int n = R.n ();
  const double h = 1./n;
  std::vector<const std::vector<double> *> vals;
  std::vector<int> used;
  if (f->kind == Function::VARIABLE) used.push_back (f->var);
  else if (f->kind == Function::COMPILED) used = f->args;
  for (int a : used) {
    /* the values beyond the box sides are those of the boundary conditions (the reference's ghost cells
       hold them at this point: gfs_simulation_init / the end of the tracer's advection) */
    Variable & V = R.vars[a];
    CHECK (gfship_bc (R.dom, V.dev, V.dev, R.level));
    V.host_time = -1.;
    vals.push_back (&host_of (R, a));
  }
  if (vals.size () > 32) { fprintf (stderr, "gfship: too many variables in alpha\n"); return 1; }
  bool first = R.alpha_dev[0] < 0;
  std::vector<double> a (R.total ());
  for (int c = 0; c < R.dim; c++) {
    std::fill (a.begin (), a.end (), 0.);
    const size_t off = c == 0 ? 1 : c == 1 ? R.idx (0, 1, 0) - R.idx (0, 0, 0) :
      R.idx (0, 0, 1) - R.idx (0, 0, 0);
    int lo[3] = { 1, 1, R.dim == 3 ? 1 : 0 }, hi[3] = { n, n, R.dim == 3 ? n : 0 };
    lo[c] = 0;
    for (int k = lo[2]; k <= hi[2]; k++)
      for (int j = lo[1]; j <= hi[1]; j++)
	for (int i = lo[0]; i <= hi[0]; i++) {
	  const size_t cell = R.idx (i, j, k);
	  double p[3];
	  cell_pos (R, i, j, k, p);
	  p[c] += h/2.;
	  double v[32];
	  for (size_t q = 0; q < vals.size (); q++) {
	    const double x1 = 1., v0 = (*vals[q])[cell], v1 = (*vals[q])[cell + off];
	    v[q] = ((x1 - 0.5)*v0 + 0.5*v1)/x1;
	  }
	  a[cell] = f->kind == Function::CONSTANT ? f->val : f->kind == Function::VARIABLE ? v[0] :
	    f->fn (p[0], p[1], p[2], R.t, v);
	}
    if (first) {
      R.alpha_dev[c] = gfship_field_alloc (R.dom, -1);
      CHECK (R.alpha_dev[c]);
    }
    CHECK (gfship_field_upload (R.dom, R.alpha_dev[c], R.level, a.data ()));
  }
  if (first)
    CHECK (gfship_sim_set_alpha (R.sim, R.alpha_dev));
  return 0;
}

int run (Run & R)
{
  R.clock0 = std::chrono::steady_clock::now ();
  /* the functions of the file are compiled (a child `cc') before anything touches the GPU: a file
     whose functions do not compile exits here without a device context or handles left behind */
  add_derived (R);
  R.functions.resolve (R.var_names ());
  resolve_refine (R);
  if (R.alpha) {
    if (R.tree_mode || R.sim_class != "Simulation") {
      fprintf (stderr, "gfship: line %d: GfsPhysicalParams { alpha } is supported for a GfsSimulation on a uniform box\n", R.alpha_line);
      return 1;
    }
    std::vector<int> used;
    if (R.alpha->kind == Function::VARIABLE) used.push_back (R.alpha->var);
    else if (R.alpha->kind == Function::COMPILED) used = R.alpha->args;
    for (int v : used)
      if (std::find (R.tracers.begin (), R.tracers.end (), R.vars[v].name) == R.tracers.end ()) {
	/* anything else changes between the two projections of a step */
	fprintf (stderr, "gfship: line %d: alpha may depend on x, y, z, t and tracers (got `%s')\n",
		 R.alpha_line, R.vars[v].name.c_str ());
	return 1;
      }
    R.alpha_static = used.empty () && !(R.alpha->kind == Function::COMPILED && R.alpha->uses_t);
  }
  if (R.tree_mode)
    return run_tree (R);
  open_pipes (R);
  CHECK (gfship_domain_create (&R.dom, R.dim, R.level, R.side, R.device));
  CHECK (gfship_sim_create (&R.sim, R.dom));
  R.vars[R.var_index ("P")].dev = gfship_sim_variable (R.sim, GFSHIP_VAR_P, 0);
  R.vars[R.var_index ("Pmac")].dev = gfship_sim_variable (R.sim, GFSHIP_VAR_PMAC, 0);
  const char * un[3] = { "U", "V", "W" };
  for (int c = 0; c < R.dim; c++)
    R.vars[R.var_index (un[c])].dev = gfship_sim_variable (R.sim, GFSHIP_VAR_U, c);
  for (const std::string & t : R.tracers) {
    int k = gfship_sim_add_tracer (R.sim);
    CHECK (k);
    R.vars[R.var_index (t)].dev = gfship_sim_variable (R.sim, GFSHIP_VAR_TRACER, k);
    CHECK (gfship_sim_set_tracer_gradient (R.sim, k, R.tracer_gradient[(size_t) k]));
  }
  for (const std::string & dv : R.device_vars) {
    gfship_field f = gfship_field_alloc (R.dom, -1);
    CHECK (f);
    R.vars[R.var_index (dv)].dev = f;
  }
  gfship_field div = -1;
  if (R.sim_class == "Poisson") {
    div = gfship_field_alloc (R.dom, -1);     /* the temporary `div` of poisson_run */
    CHECK (div);
  }
  apply_multilevel (gfship_sim_projection_params (R.sim), R.proj_set);
  apply_multilevel (gfship_sim_approx_projection_params (R.sim), R.approx_set);
  gfship_advection_params * adv = gfship_sim_advection_params (R.sim);
  for (auto & kv : R.adv_set) {
    if (kv.first == "cfl") adv->cfl = atof (kv.second.c_str ());
    else if (kv.first == "gc") adv->gc = atoi (kv.second.c_str ());
    else if (kv.first == "gradient") {
      if (gradient_kind (kv.second) >= 0) adv->gradient = gradient_kind (kv.second);
      else { fprintf (stderr, "gfship: unsupported gradient `%s'\n", kv.second.c_str ()); return 1; }
    }
  }
  for (int c = 0; c < R.dim; c++)
    if (R.source[c] != 0.)
      CHECK (gfship_sim_set_source (R.sim, c, R.source[c]));
  for (int c = 0; c < R.dim; c++)
    if (R.visc[c] != 0.) {
      CHECK (gfship_sim_set_viscosity (R.sim, c, R.visc[c]));
      apply_multilevel (gfship_sim_diffusion_params (R.sim, c), R.diff_set[c]);
    }
  // the particle lists: created (and the previous velocity of the GfsForceCoeff objects stored)
  // while the fields still hold the zeros of a fresh simulation, like the reference, which reads
  // the list before any GfsInit event runs (gfs_force_coeff_read, :181-187)
  for (auto & ps : R.plists) {
    int np = (int) ps->id.size ();
    CHECK (gfship_particles_create (&ps->pl, R.sim, np, ps->pos.data (), ps->id.data ()));
    if (ps->particulate) {
      CHECK (gfship_particles_set_particulate (ps->pl, ps->vel.data (), ps->mass.data (), ps->volume.data ()));
      /* gravity of GfsForceBuoy = the GfsSource intensities on U, V, W (compute_buoyancy_force) */
      CHECK (gfship_particles_set_forces (ps->pl, (int) ps->forces.size (), ps->forces.data (), R.source));
      for (size_t f = 0; f < ps->force_functions.size (); f++)
	if (!ps->force_functions[f].empty ())
	  CHECK (gfship_particles_set_force_coefficient (ps->pl, (int) f, ps->force_functions[f].c_str ()));
    }
  }
  set_boundary_conditions (R);
  apply_init (R);
  for (auto & is : R.init_spectra) {
    gfship_init_spectra_params p = is.first;
    if (p.level < 0) p.level = R.level;
    gfship_field v[3];
    for (int c = 0; c < 3; c++) {
      int q = R.var_index (is.second[c]);
      if (q < 0 || R.vars[q].dev < 0) {
	fprintf (stderr, "gfship: GfsInitSpectra: `%s' is not a variable of the simulation\n", is.second[c].c_str ());
	return 1;
      }
      v[c] = R.vars[q].dev;
      R.vars[q].host_time = -1.;
    }
    CHECK (gfship_init_spectra (R.dom, &p, v));
  }
  if (R.snapshot.has_tree) {
    // the cell data the file came with (gfs_box_read -> ftt_cell_read_binary / ftt_cell_read +
    // gfs_cell_read*, src/boundary.c:1925-1944): all levels of the variables of `variables = ...'
    if (R.snapshot.depth != R.level) {
      fprintf (stderr, "gfship: the cell data of the file is refined to level %d\n", R.snapshot.depth);
      return 1;
    }
    std::vector<gfship_field> f, tmp;
    std::vector<int> host_only;
    for (const std::string & nm : R.snapshot.variables) {
      int q = R.get_or_add_variable (nm);
      if (R.vars[q].dev >= 0) { f.push_back (R.vars[q].dev); R.vars[q].host_time = -1.; host_only.push_back (-1); }
      else {
	gfship_field t = gfship_field_alloc (R.dom, -1);
	CHECK (t);
	f.push_back (t); tmp.push_back (t); host_only.push_back (q);
      }
    }
    CHECK (gfship_snapshot_tree_read (R.dom, (int) f.size (), f.data (), R.snapshot.tree.data (),
				      R.snapshot.tree.size ()));
    for (size_t k = 0; k < f.size (); k++)
      if (host_only[k] >= 0) {
	Variable & V = R.vars[host_only[k]];
	V.host.assign (R.total (), 0.);
	CHECK (gfship_field_download (R.dom, f[k], R.level, V.host.data ()));
      }
    for (gfship_field t : tmp) gfship_field_free (R.dom, t);
    R.snapshot.tree.clear ();
    if (R.sim_class == "Simulation")
      CHECK (gfship_sim_restart (R.sim, R.t, R.i));
  }
  events_init (R);

  if (R.sim_class == "Poisson") {
    // poisson_run, src/simulation.c:2213-2285 (P has a Dirichlet condition somewhere, or the
    // mean of Div is removed by correct_div: only the Dirichlet case is supported here)
    gfship_field p = R.vars[R.var_index ("P")].dev;
    bool dirichlet = false;
    for (int d = 0; d < 2*R.dim; d++)
      if (R.bc[d].count ("P") && R.bc[d]["P"].kind == GFSHIP_BC_DIRICHLET) dirichlet = true;
    if (!dirichlet) { fprintf (stderr, "gfship: GfsPoisson needs a Dirichlet condition on P\n"); return 1; }
    gfship_field res = gfship_field_alloc (R.dom, -1), dia = gfship_field_alloc (R.dom, -1);
    CHECK (res); CHECK (dia);
    CHECK (gfship_bc (R.dom, p, p, R.level));
    gfship_multilevel_params * par = gfship_sim_approx_projection_params (R.sim);
    while (R.i < R.iend && R.t < R.end) {
      {
	/* correct_div with a Dirichlet condition: rescale_div only, div = Div*size*size*fraction
	   (src/simulation.c:2156-2162,2170-2211) */
	const std::vector<double> & Div = host_of (R, R.var_index ("Div"));
	std::vector<double> scaled (R.total (), 0.);
	double size = 1./R.n ();
	for_each_cell (R, [&] (int, int, int, size_t c) { scaled[c] = Div[c]*(size*size*1.); });
	CHECK (gfship_field_upload (R.dom, div, R.level, scaled.data ()));
      }
      CHECK (gfship_poisson_coefficients (R.dom));
      for (int l = 0; l <= R.level; l++)
	CHECK (gfship_field_fill (R.dom, dia, l, 0.));
      CHECK (gfship_poisson_solve (R.dom, par, p, div, res, dia, 1.));
      R.t = 0.;         /* sim->time.t = sim->tnext, and tnext stays 0. (src/simulation.c:1014,2269) */
      R.i++;
      events_do (R);
    }
  }
  else if (R.sim_class == "Advection") {
    // advection_run, src/simulation.c:2061-2116, with the velocity of a GfsVariableStreamFunction
    // (variable_stream_function_event, src/variable.c:1041-1086: psi at the four corners of every
    // leaf -> MAC velocities, centred velocities = means of the two faces, BC)
    if (!R.stream_function) {
      fprintf (stderr, "gfship: GfsAdvection needs a GfsVariableStreamFunction (prescribed velocity)\n");
      return 1;
    }
    {
      const int n = R.n ();
      std::vector<double> un[2], uc[2];
      for (int c = 0; c < 2; c++) { un[c].assign (R.total (), 0.); uc[c].assign (R.total (), 0.); }
      for (int j = 1; j <= n; j++)
	for (int i = 1; i <= n; i++) {
	  double o[3], p[3] = { 0., 0., 0. };
	  cell_pos (R, i, j, 0, o);
	  double h = (1./n)/2.;
	  p[0] = o[0] - h; p[1] = o[1] - h; double psi0 = eval (R, R.stream_function, p, -1);
	  p[0] = o[0] + h; p[1] = o[1] - h; double psi1 = eval (R, R.stream_function, p, -1);
	  p[0] = o[0] + h; p[1] = o[1] + h; double psi2 = eval (R, R.stream_function, p, -1);
	  p[0] = o[0] - h; p[1] = o[1] + h; double psi3 = eval (R, R.stream_function, p, -1);
	  double hh = 2.*h;
	  double f0 = (psi2 - psi1)*1./hh, f1 = (psi3 - psi0)*1./hh;
	  double f2 = (psi3 - psi2)*1./hh, f3 = (psi0 - psi1)*1./hh;
	  un[0][R.idx (i, j, 0)] = f0;
	  if (i == 1) un[0][R.idx (0, j, 0)] = f1;
	  un[1][R.idx (i, j, 0)] = f2;
	  if (j == 1) un[1][R.idx (i, 0, 0)] = f3;
	  uc[0][R.idx (i, j, 0)] = (f0 + f1)/2.;
	  uc[1][R.idx (i, j, 0)] = (f2 + f3)/2.;
	}
      const char * unames[2] = { "U", "V" };
      for (int c = 0; c < 2; c++) {
	gfship_field fu = R.vars[R.var_index (unames[c])].dev;
	CHECK (gfship_field_upload (R.dom, gfship_sim_variable (R.sim, GFSHIP_VAR_UN, c), R.level, un[c].data ()));
	CHECK (gfship_field_upload (R.dom, fu, R.level, uc[c].data ()));
	CHECK (gfship_bc (R.dom, fu, fu, R.level));
	R.vars[R.var_index (unames[c])].host_time = -1.;
      }
    }
    CHECK (gfship_sim_set_time (R.sim, R.end, R.dtmax));
    CHECK (gfship_sim_set_next_event (R.sim, next_event_hook, &R));
    for (const std::string & t : R.tracers) {
      gfship_field ft = R.vars[R.var_index (t)].dev;
      CHECK (gfship_bc (R.dom, ft, ft, R.level));
    }
    while (R.t < R.end && R.i < R.iend) {
      events_do (R);
      CHECK (gfship_sim_set_time (R.sim, R.end, R.dtmax));
      CHECK (gfship_sim_advection_step (R.sim));
      R.t = gfship_sim_time (R.sim);
      R.i = gfship_sim_iter (R.sim);
    }
    events_do (R);
  }
  else {
    // simulation_run, src/simulation.c:432-557
    CHECK (gfship_sim_set_time (R.sim, R.end, R.dtmax));
    CHECK (gfship_sim_set_next_event (R.sim, next_event_hook, &R));
    if (refresh_alpha (R)) return 1;
    CHECK (gfship_sim_start (R.sim));
    if (R.alpha && !R.alpha_static) invalidate_device_copies (R);   /* the half step of the tracers */
    while (R.t < R.end && R.i < R.iend) {
      events_do (R);
      /* a GfsEventStop may just have set time.end = time.t: the reference still completes this
	 iteration of the loop and stops at the next test of its condition */
      CHECK (gfship_sim_set_time (R.sim, R.end, R.dtmax));
      if (refresh_alpha (R)) return 1;
      CHECK (gfship_sim_step (R.sim));
      R.t = gfship_sim_time (R.sim);
      R.i = gfship_sim_iter (R.sim);
    }
    events_do (R);
  }
  if (!R.particles_out.empty ()) {
    // the lists as gfs_event_list_write / gfs_particle_write / gfs_particulate_write print them
    // (src/event.c:2508-2524, src/particle.c:88-99, modules/particulatecommon.c:907-921)
    FILE * fp = fopen (R.particles_out.c_str (), "w");
    if (!fp) { fprintf (stderr, "gfship: cannot open `%s'\n", R.particles_out.c_str ()); return 1; }
    for (auto & ps : R.plists) {
      int m = gfship_particles_slots (ps->pl);
      CHECK (m);
      std::vector<double> pos (3*(size_t) std::max (m, 1)), vel (pos.size ()), force (pos.size ()), mass (std::max (m, 1));
      std::vector<unsigned> id (std::max (m, 1));
      int k = gfship_particles_download (ps->pl, pos.data (), id.data ());
      CHECK (k);
      if (ps->particulate)
	CHECK (gfship_particles_download_particulate (ps->pl, vel.data (), mass.data (), force.data ()));
      std::map<unsigned, double> volume;
      for (size_t q = 0; q < ps->id.size () && ps->particulate; q++) volume[ps->id[q]] = ps->volume[q];
      fprintf (fp, "GfsParticleList %s {\n", ps->particulate ? "GfsParticulate" : "GfsParticle");
      for (int q = 0; q < k; q++) {
	fprintf (fp, "    %s %d %g %g %g", ps->particulate ? "GfsParticulate" : "GfsParticle", (int) id[q],
		 pos[3*q], pos[3*q + 1], pos[3*q + 2]);
	if (ps->particulate) {
	  fprintf (fp, " %g %g %g %g %g", mass[q], volume[id[q]], vel[3*q], vel[3*q + 1], vel[3*q + 2]);
	  fprintf (fp, " %g %g %g", force[3*q], force[3*q + 1], force[3*q + 2]);
	}
	fputc ('\n', fp);
      }
      fputs ("}\n", fp);
    }
    fclose (fp);
  }
  for (auto & ps : R.plists) gfship_particles_destroy (ps->pl);
  for (auto & o : R.outputs) o->close ();
  gfship_sim_destroy (R.sim);
  gfship_domain_destroy (R.dom);
  return 0;
}

} // namespace

// --check: parse, compile the functions and describe the run without touching a device
int check (Run & R)
{
  add_derived (R);
  R.functions.resolve (R.var_names ());
  resolve_refine (R);
  printf ("class Gfs%s dim %d level %d%s\n", R.sim_class.c_str (), R.dim, R.level,
	  R.tree_mode ? " (coarsest leaves of a refined tree)" : "");
  if (R.tree_mode)        /* no host copies of fields without the tree, which needs the device */
    R.init.clear ();
  printf ("sides");
  for (int d = 0; d < 2*R.dim; d++)
    printf (" %s=%s", side_name[d], R.side[d] == GFSHIP_SIDE_PERIODIC ? "periodic" : "boundary");
  printf ("\ntime t %g i %u end %g iend %u\n", R.t, R.i, R.end, R.iend);
  double p[3] = { 0.125, -0.25, R.dim == 3 ? 0.375 : 0. };
  for (auto & kv : R.init) {
    bool reads_vars = kv.second->kind == Function::VARIABLE || !kv.second->args.empty ();
    if (reads_vars) printf ("init %s = <function of variables>\n", kv.first.c_str ());
    else printf ("init %s (0.125,-0.25,%g) = %.17g\n", kv.first.c_str (), p[2], eval (R, kv.second, p, -1));
  }
  for (int d = 0; d < 2*R.dim; d++)
    for (auto & kv : R.bc[d])
      printf ("bc %s %s %s %.17g\n", side_name[d], kv.first.c_str (),
	      kv.second.kind == GFSHIP_BC_DIRICHLET ? "dirichlet" : "neumann",
	      eval (R, kv.second.val, p, -1));
  for (int c = 0; c < R.dim; c++)
    if (R.visc[c] != 0.) printf ("viscosity %d %g\n", c, R.visc[c]);
  for (auto & e : R.events)
    printf ("event %s line %d start %g step %g istep %u end_event %d\n", e->cls.c_str (), e->line,
	    e->start, e->step == DBL_MAX ? -1. : e->step, e->istep == INT_MAX ? 0 : e->istep,
	    (int) e->end_event);
  return 0;
}

int main (int argc, char ** argv)
{
  Run R;
  bool check_only = false;
  // dimension from the program name, like gerris2D / gerris3D
  std::string prog = argv[0];
  R.dim = prog.find ("3D") != std::string::npos ? 3 : 2;
  std::map<std::string, std::string> defs;
  std::string file;
  for (int a = 1; a < argc; a++) {
    std::string s = argv[a];
    if (s == "-2") R.dim = 2;
    else if (s == "-3") R.dim = 3;
    else if (s == "--device" && a + 1 < argc) R.device = atoi (argv[++a]);
    else if (s == "--check") check_only = true;
    else if (s == "--particles" && a + 1 < argc) R.particles_out = argv[++a];
    else if (s.compare (0, 2, "-D") == 0) {
      std::string d = s.size () > 2 ? s.substr (2) : (a + 1 < argc ? argv[++a] : "");
      size_t eq = d.find ('=');
      if (eq == std::string::npos) defs[d] = "1";
      else defs[d.substr (0, eq)] = d.substr (eq + 1);
    }
    else if (s == "-h" || s == "--help") {
      printf ("Usage: %s [-2|-3] [-DNAME=VALUE] [--device N] [--particles FILE] file.gfs\n", argv[0]);
      return 0;
    }
    else file = s;
  }
  if (file.empty ()) { fprintf (stderr, "gfship: no simulation file given\n"); return 1; }
  g_shell.start ();
  struct ShellStop { ~ShellStop () { g_shell.stop (); } } shell_stop;
  std::stringstream ss;
  if (file == "-")                       /* `gerris2D -': the simulation file on standard input */
    ss << std::cin.rdbuf ();
  else {
    std::ifstream in (file);
    if (!in) { fprintf (stderr, "gfship: cannot open `%s'\n", file.c_str ()); return 1; }
    ss << in.rdbuf ();
  }
  try {
    // a file with cell data (a snapshot written by GfsOutputSimulation): the data is cut out
    // before the text is parsed
    R.snapshot = gfs::split_simulation_file (ss.str (), file, R.dim);
    parse_file (R, substitute (R.snapshot.text, defs), file);
    R.snapshot.text.clear ();
    for (const std::string & nm : R.snapshot.variables)
      R.get_or_add_variable (nm);        /* gfs_domain_add_variable of domain_read, src/domain.c:283-293 */
    if (R.snapshot.has_tree && R.level == 0 && R.snapshot.depth > 0)
      R.level = R.snapshot.depth;        /* no GfsRefine in a file that carries its tree */
    return check_only ? check (R) : run (R);
  }
  catch (const ParseError & e) {
    fprintf (stderr, "gfship: %s\n", e.what ());
    return 1;
  }
}
