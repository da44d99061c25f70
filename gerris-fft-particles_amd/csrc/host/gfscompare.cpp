// gfscompare.cpp -- gfshipcompare2D / gfshipcompare3D: difference between the solutions of two
// simulation files for one variable, the way the reference's tools/gfscompare.c does it for two
// files with the same (here: uniform) tree:
//
//   gfshipcompare3D [-C] [-w] [-v] [-H] FILE1 FILE2 VAR
//
//   e = VAR(FILE1) - VAR(FILE2) on every leaf (difference_tree, tools/gfscompare.c:163-214);
//   -C  --constant      subtract the weighted mean of e first: "apply a constant shift to one of
//                       the field, minimizing the error between the two fields (useful for
//                       pressure)" (difference_constant :223-243, :657-669);
//   -w  --not-weighted  weights 1 instead of the cell volumes (:239,:266);
//   -v  --verbose       prints the norms of both fields and
//                       "total err first: %10.3e second: %10.3e infty: %10.3e w: %g" on stderr
//                       (:567-595, :683-686; gfs_norm_add / gfs_norm_update src/fluid.c:2139-2171);
//   -H  --histogram     "(error, volume fraction)" pairs on stdout (:268-269).
// Exit status 0; 1 when the files cannot be compared (different trees, unknown variable).
// Host arithmetic over two files: no device is involved.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include "gfs_snapshot.hpp"

using namespace gfs;

struct Norm { double bias = 0., first = 0., second = 0., infty = -1e300, w = 0.; };

static void norm_add (Norm & n, double val, double weight)   /* gfs_norm_add, src/fluid.c:2139-2154 */
{
  n.bias += weight*val;
  val = fabs (val);
  if (weight != 0. && val > n.infty) n.infty = val;
  n.first += weight*val;
  n.second += weight*val*val;
  n.w += weight;
}

static void norm_update (Norm & n)                           /* gfs_norm_update, :2156-2171 */
{
  if (n.w > 0.) {
    n.bias /= n.w;
    n.first /= n.w;
    n.second = sqrt (n.second/n.w);
  }
  else
    n.infty = 0.;
}

static bool load (const char * name, int dim, SimulationFile & F)
{
  std::ifstream in (name, std::ios::binary);
  if (!in) { fprintf (stderr, "gfscompare: cannot open file `%s'\n", name); return false; }
  std::stringstream ss;
  ss << in.rdbuf ();
  try {
    F = split_simulation_file (ss.str (), name, dim);
  }
  catch (const ParseError & e) {
    fprintf (stderr, "gfscompare: file `%s' is not a valid simulation file\n%s\n", name, e.what ());
    return false;
  }
  if (!F.has_tree) {
    fprintf (stderr, "gfscompare: file `%s' holds no cell data\n", name);
    return false;
  }
  return true;
}

static int variable_index (const SimulationFile & F, const std::string & v)
{
  for (size_t q = 0; q < F.variables.size (); q++)
    if (F.variables[q] == v) return (int) q;
  return -1;
}

int main (int argc, char ** argv)
{
  std::string prog = argv[0];
  const int dim = prog.find ("3D") != std::string::npos ? 3 : 2;
  bool constant = false, weighted = true, verbose = false, histogram = false;
  std::vector<const char *> pos;
  for (int a = 1; a < argc; a++) {
    std::string s = argv[a];
    if (s == "-C" || s == "--constant") constant = true;
    else if (s == "-w" || s == "--not-weighted") weighted = false;
    else if (s == "-v" || s == "--verbose") verbose = true;
    else if (s == "-H" || s == "--histogram") histogram = true;
    else if (s == "-h" || s == "--help") {
      fprintf (stderr,
	       "Usage: gfscompare [OPTION] FILE1 FILE2 VAR\n"
	       "Computes the difference between the solutions in FILE1 and FILE2\n"
	       "for variable VAR.\n\n"
	       "  -C    --constant    apply a constant shift to one of the field, minimizing\n"
	       "                      the error between the two fields (useful for pressure)\n"
	       "  -w    --not-weighted do not use area-weighted norm estimation\n"
	       "  -H    --histogram   output (error,volume) pairs for each cell used\n"
	       "                      to compute the error norms\n"
	       "  -v    --verbose     display difference statistics and other info\n"
	       "  -h    --help        display this help and exit\n");
      return 0;
    }
    else if (!s.empty () && s[0] == '-') {
      fprintf (stderr, "gfscompare: option `%s' is not available here\n"
	       "Try `gfscompare --help' for more information.\n", s.c_str ());
      return 1;
    }
    else pos.push_back (argv[a]);
  }
  if (pos.size () < 1) { fprintf (stderr, "gfscompare: missing FILE1\nTry `gfscompare --help' for more information.\n"); return 1; }
  if (pos.size () < 2) { fprintf (stderr, "gfscompare: missing FILE2\nTry `gfscompare --help' for more information.\n"); return 1; }
  if (pos.size () < 3) { fprintf (stderr, "gfscompare: missing VAR\nTry `gfscompare --help' for more information.\n"); return 1; }
  SimulationFile F1, F2;
  if (!load (pos[0], dim, F1) || !load (pos[1], dim, F2)) return 1;
  const std::string var = pos[2];
  int v1 = variable_index (F1, var), v2 = variable_index (F2, var);
  if (v1 < 0) { fprintf (stderr, "gfscompare: unknown variable `%s' for `%s'\nTry `gfscompare --help' for more information.\n", var.c_str (), pos[0]); return 1; }
  if (v2 < 0) { fprintf (stderr, "gfscompare: unknown variable `%s' for `%s'\nTry `gfscompare --help' for more information.\n", var.c_str (), pos[1]); return 1; }
  if (F1.depth != F2.depth) {
    /* difference_tree locates every cell of FILE1 in FILE2 at its own level: on uniform trees of
       different depths no leaf has a counterpart */
    fprintf (stderr, "gfscompare: the files are not comparable\n");
    return 1;
  }
  std::vector<double> a = tree_leaves (F1.tree, F1.variables.size (), (size_t) v1);
  std::vector<double> b = tree_leaves (F2.tree, F2.variables.size (), (size_t) v2);
  const double h = 1./(1 << F1.depth);
  const double volume = dim == 3 ? h*h*h : h*h;       /* ftt_cell_volume */
  const double w = weighted ? volume*1. : 1.;
  if (verbose)
    for (int f = 0; f < 2; f++) {
      const std::vector<double> & x = f ? b : a;
      Norm n;
      double mn = 1e300, mx = -1e300, sum = 0., sum2 = 0.;
      for (double v : x) {
	norm_add (n, v, volume);
	if (v < mn) mn = v;
	if (v > mx) mx = v;
	sum += v; sum2 += v*v;
      }
      norm_update (n);
      double mean = sum/x.size ();
      double sd = sqrt (fmax (0., sum2/x.size () - mean*mean));
      fprintf (stderr, "%s:\n  first: %g second: %g infty: %g w: %g\n  min: %g avg: %g | %g max: %g\n",
	       pos[f], n.first, n.second, n.infty, n.w, mn, mean, sd, mx);
    }
  std::vector<double> e (a.size ());
  for (size_t q = 0; q < a.size (); q++) e[q] = a[q] - b[q];
  double shift = 0.;
  if (constant) {
    double sum = 0., weight = 0.;
    for (double x : e) { sum += w*x; weight += w; }
    shift = weight > 0. ? sum/weight : 0.;
  }
  Norm n;
  for (double x : e) {
    norm_add (n, x - shift, w);
    if (histogram) printf ("%g %g\n", x, 1.);
  }
  norm_update (n);
  if (verbose)
    fprintf (stderr, "total err first: %10.3e second: %10.3e infty: %10.3e w: %g\n",
	     n.first, n.second, n.infty, n.w);
  return 0;
}
