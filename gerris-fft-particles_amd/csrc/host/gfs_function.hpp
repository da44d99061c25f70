// gfs_function.hpp -- GfsFunction: constants, variable names and C expressions.
//
// The reference turns every non-trivial function of a simulation file into C source, compiles
// it with the system compiler into a shared object and dlopen()s it (gfs_module_new /
// function_compile, src/utils.c:443-866).  The same is done here: all functions of a file go
// into one translation unit, compiled once with `cc -shared` and loaded with dlopen, so an
// expression means exactly what the C compiler says it means (double arithmetic, libm).
// x, y, z, t are the cell (or face) centre and the simulation time; the domain variables an
// expression names are passed by value.
#pragma once
#include "gfs_text.hpp"
#include <cstdio>
#include <dlfcn.h>
#include <unistd.h>
#include <memory>

namespace gfs {

typedef double (*CompiledFn) (double x, double y, double z, double t, const double * v);

struct Function {
  FunctionText src;
  enum Kind { NONE, CONSTANT, VARIABLE, COMPILED } kind = NONE;
  double val = 0.;                 // CONSTANT
  int var = -1;                    // VARIABLE: index in the variable table
  std::vector<int> args;           // COMPILED: variables passed in v[]
  bool uses_t = false;
  CompiledFn fn = nullptr;
  int where = 0;                   // line of the file, for messages
};

class FunctionSet {
public:
  ~FunctionSet () {
    if (handle_) dlclose (handle_);
    if (!dir_.empty ()) {
      unlink ((dir_ + "/functions.c").c_str ());
      unlink ((dir_ + "/functions.so").c_str ());
      rmdir (dir_.c_str ());
    }
  }

  Function * add (const FunctionText & t, int line) {
    fns_.emplace_back (new Function);
    fns_.back ()->src = t;
    fns_.back ()->where = line;
    return fns_.back ().get ();
  }

  // classify every function against the variable names, then compile what needs compiling
  void resolve (const std::vector<std::string> & names) {
    std::string code = "#include <math.h>\n#include <stdlib.h>\n#include <stdio.h>\n"
      "#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n";
    int ncompiled = 0;
    for (size_t k = 0; k < fns_.size (); k++) {
      Function & f = *fns_[k];
      if (f.kind != Function::NONE) continue;
      const std::string & e = f.src.text;
      if (!f.src.block && Reader::is_number (e)) {
	f.kind = Function::CONSTANT;
	f.val = atof (e.c_str ());
	continue;
      }
      if (!f.src.block) {
	int v = index_of (names, e);
	if (v >= 0) { f.kind = Function::VARIABLE; f.var = v; continue; }
      }
      f.kind = Function::COMPILED;
      std::string decl;
      for (const std::string & id : identifiers (e)) {
	if (id == "t") f.uses_t = true;
	int v = index_of (names, id);
	if (v >= 0) {
	  decl += "  double " + id + " = _v[" + std::to_string (f.args.size ()) + "];\n";
	  f.args.push_back (v);
	}
      }
      code += "double gfs_f" + std::to_string (k) +
	" (double x, double y, double z, double t, const double * _v) {\n"
	"  (void) x; (void) y; (void) z; (void) t; (void) _v;\n" + decl +
	"#line " + std::to_string (f.where) + " \"simulation file\"\n";
      if (f.src.block)
	code += "  " + e + "\n  return 0.;\n}\n";
      else
	code += "  return (double) (" + e + ");\n}\n";
      ncompiled++;
    }
    if (ncompiled == 0) return;
    char tmpl[] = "/tmp/gfshipXXXXXX";
    if (!mkdtemp (tmpl)) throw ParseError ("cannot create a temporary directory");
    dir_ = tmpl;
    std::string c = dir_ + "/functions.c", so = dir_ + "/functions.so";
    FILE * fp = fopen (c.c_str (), "w");
    if (!fp) throw ParseError ("cannot write " + c);
    fputs (code.c_str (), fp);
    fclose (fp);
    const char * cc = getenv ("GFSHIP_CC");
    // -ffp-contract=off: no fused multiply-add, the arithmetic of the reference's x86-64 build
    std::string cmd = std::string (cc ? cc : "cc") +
      " -O1 -fPIC -shared -ffp-contract=off -o " + so + " " + c + " -lm";
    if (system (cmd.c_str ()) != 0)
      throw ParseError ("error compiling the functions of the simulation file (" + cmd + ")");
    handle_ = dlopen (so.c_str (), RTLD_NOW);
    if (!handle_) throw ParseError (std::string ("dlopen: ") + dlerror ());
    for (size_t k = 0; k < fns_.size (); k++)
      if (fns_[k]->kind == Function::COMPILED && !fns_[k]->fn) {
	fns_[k]->fn = (CompiledFn) dlsym (handle_, ("gfs_f" + std::to_string (k)).c_str ());
	if (!fns_[k]->fn) throw ParseError ("missing compiled function");
      }
  }

private:
  static int index_of (const std::vector<std::string> & names, const std::string & n) {
    for (size_t q = 0; q < names.size (); q++)
      if (names[q] == n) return (int) q;
    return -1;
  }
  // identifiers of a C expression, each once, in order of appearance
  static std::vector<std::string> identifiers (const std::string & e) {
    std::vector<std::string> ids;
    size_t p = 0;
    while (p < e.size ()) {
      if (isalpha ((unsigned char) e[p]) || e[p] == '_') {
	size_t b = p;
	while (p < e.size () && (isalnum ((unsigned char) e[p]) || e[p] == '_')) p++;
	bool member = b > 0 && (e[b - 1] == '.' || isdigit ((unsigned char) e[b - 1]));
	std::string id = e.substr (b, p - b);
	if (!member && std::find (ids.begin (), ids.end (), id) == ids.end ())
	  ids.push_back (id);
      }
      else if (e[p] == '"') {
	p++;
	while (p < e.size () && e[p] != '"') p++;
	p++;
      }
      else
	p++;
    }
    return ids;
  }

  std::vector<std::unique_ptr<Function>> fns_;
  void * handle_ = nullptr;
  std::string dir_;
};

} // namespace gfs
