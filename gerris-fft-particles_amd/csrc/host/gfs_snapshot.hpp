// gfs_snapshot.hpp -- a Gerris simulation file that carries cell data (what GfsOutputSimulation
// writes and `gerris' restarts from): taking the file apart and putting one together.
//
//   # Gerris Flow Solver 3D version ...
//   1 3 GfsSimulation GfsBox GfsGEdge { version = 120812 variables = P,Pmac,U,V,W binary = 1 } {
//     GfsTime { i = 12 t = 0.3 } ...objects...
//   }
//   GfsBox { id = 1 pid = -1 size = 4096 x = 0 y = 0 z = 0 ... } {
//   <cell tree: ftt_cell_write_binary / ftt_cell_write, src/ftt.c:1728-1799>}
//   1 1 right
//
// (simulation_write src/simulation.c:77-170, domain_write src/domain.c:168-209, gfs_box_write
// src/boundary.c:1819-1851).  The tree is binary (`binary = 1': guint flags, double -1., one double
// per variable, pre-order) or text (one line per cell: `flags -1 v1 v2 ...' in %g).  Shared by the
// front end (restart, OutputSimulation) and by the comparison tool; host code only.
#pragma once
#include "gfs_text.hpp"
#include <cstdint>
#include <cstdio>

namespace gfs {

struct SimulationFile {
  std::string text;                     // the file without the cell data (what the parser reads)
  bool has_tree = false, binary = false;
  std::vector<std::string> variables;   // variables = a,b,c of the graph parameters
  std::string tree;                     // the cell data as the binary image (converted if text)
  int dim = 0, depth = -1;
};

inline std::vector<std::string> split_commas (const std::string & s)
{
  std::vector<std::string> out;
  size_t b = 0;
  while (b <= s.size ()) {
    size_t e = s.find (',', b);
    if (e == std::string::npos) e = s.size ();
    if (e > b) out.push_back (s.substr (b, e - b));
    b = e + 1;
  }
  return out;
}

inline size_t tree_record (size_t nvars) { return 4 + 8 + 8*nvars; }

inline size_t tree_cells (int dim, int depth)
{
  size_t cells = 0, c = 1;
  for (int l = 0; l <= depth; l++) { cells += c; c *= dim == 3 ? 8 : 4; }
  return cells;
}

// depth of the (uniform) tree whose binary image starts at p: follow the first children
inline int tree_depth_binary (const char * p, size_t avail, size_t rec)
{
  int depth = 0;
  size_t o = 0;
  while (o + rec <= avail) {
    uint32_t flags;
    memcpy (&flags, p + o, 4);
    if (flags & 16u) return depth;
    depth++;
    o += rec;
  }
  return -1;
}

// a text tree (ftt_cell_write + gfs_cell_write) into the binary image; returns the characters used
inline size_t tree_text_to_binary (const std::string & s, size_t b, size_t nvars, int dim,
				   std::string & out, int & depth)
{
  size_t p = b;
  depth = -1;
  // iterative pre-order walk: the stack holds the number of children still to come per level
  std::vector<int> todo;
  int level = 0;
  auto number = [&] (double & v) -> bool {
    while (p < s.size () && isspace ((unsigned char) s[p])) p++;
    char * end;
    v = strtod (s.c_str () + p, &end);
    if (end == s.c_str () + p) return false;
    p = (size_t) (end - s.c_str ());
    return true;
  };
  for (;;) {
    double f, m;
    if (!number (f) || !number (m)) throw ParseError ("cell data: expecting `flags -1 values...'");
    if (m != -1.) throw ParseError ("cell data: solid fractions (mixed cells) are not supported");
    uint32_t flags = (uint32_t) f;
    out.append ((const char *) &flags, 4);
    out.append ((const char *) &m, 8);
    for (size_t v = 0; v < nvars; v++) {
      double x;
      if (!number (x)) throw ParseError ("cell data: expecting a number");
      out.append ((const char *) &x, 8);
    }
    if (flags & 16u) {
      if (depth < 0) depth = level;
      else if (depth != level) throw ParseError ("cell data: the tree is not uniform");
      // next sibling, or up
      while (!todo.empty () && --todo.back () == 0) { todo.pop_back (); level--; }
      if (todo.empty ()) break;
    }
    else {
      todo.push_back (dim == 3 ? 8 : 4);
      level++;
    }
  }
  return p - b;
}

// Cuts the cell data out of the text of a simulation file.  dim: 2 or 3 (the program's).
inline SimulationFile split_simulation_file (const std::string & all, const std::string & name,
					     int dim)
{
  SimulationFile F;
  F.dim = dim;
  Reader r (all, name);
  r.number (); r.number ();
  r.word (); r.word (); r.word ();
  {
    int l0 = r.line ();
    Reader g (r.braces (), name, l0);
    while (!g.eof ()) {
      std::string k = g.word ();
      g.expect ('=');
      std::string v = g.word ();
      if (k == "variables") F.variables = split_commas (v);
      else if (k == "binary") F.binary = atoi (v.c_str ()) != 0;
    }
  }
  r.braces ();                       // the simulation body
  if (r.eof ()) { F.text = all; return F; }
  r.word ();                         // GfsBox
  r.braces ();                       // its parameters
  if (r.peek (false) != '{') { F.text = all; return F; }
  // `{', then the data, then `}'
  size_t open = all.find ('{', r.offset ());
  size_t b = open + 1;
  size_t used;
  const size_t rec = tree_record (F.variables.size ());
  if (F.binary) {
    if (b >= all.size () || all[b] != '\n') throw ParseError (name + ": expecting a newline before the binary cell data");
    b++;
    F.depth = tree_depth_binary (all.data () + b, all.size () - b, rec);
    if (F.depth < 0) throw ParseError (name + ": truncated binary cell data");
    used = tree_cells (dim, F.depth)*rec;
    if (b + used > all.size ()) throw ParseError (name + ": truncated binary cell data");
    F.tree.assign (all, b, used);
  }
  else
    used = tree_text_to_binary (all, b, F.variables.size (), dim, F.tree, F.depth);
  size_t close = b + used;
  while (close < all.size () && isspace ((unsigned char) all[close])) close++;
  if (close >= all.size () || all[close] != '}')
    throw ParseError (name + ": expecting a closing brace after the cell data "
		      "(make sure the file has " + std::to_string (dim) + " spatial dimensions)");
  F.has_tree = true;
  F.text = all.substr (0, open) + all.substr (close + 1);
  return F;
}

// the leaves of variable v of a binary image, in the order of the file (the traversal order)
inline std::vector<double> tree_leaves (const std::string & tree, size_t nvars, size_t v)
{
  const size_t rec = tree_record (nvars);
  std::vector<double> out;
  for (size_t o = 0; o + rec <= tree.size (); o += rec) {
    uint32_t flags;
    memcpy (&flags, tree.data () + o, 4);
    if (flags & 16u) {
      double x;
      memcpy (&x, tree.data () + o + 12 + 8*v, 8);
      out.push_back (x);
    }
  }
  return out;
}

// ftt_cell_write + gfs_cell_write (text) of a binary image
inline void tree_write_text (FILE * fp, const std::string & tree, size_t nvars)
{
  const size_t rec = tree_record (nvars);
  for (size_t o = 0; o + rec <= tree.size (); o += rec) {
    uint32_t flags;
    memcpy (&flags, tree.data () + o, 4);
    fprintf (fp, "%u -1", flags);
    for (size_t v = 0; v < nvars; v++) {
      double x;
      memcpy (&x, tree.data () + o + 12 + 8*v, 8);
      fprintf (fp, " %g", x);
    }
    fputc ('\n', fp);
  }
}

} // namespace gfs
