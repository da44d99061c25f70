// gfs_text.hpp -- reading the text of a Gerris simulation file (.gfs).
//
// The format is the GTS token stream the reference reads with GtsFile (objects one per line,
// `key = value` blocks in braces, `#` comments), see src/simulation.c:1265-1341 for the
// simulation body and src/utils.c:868-983 (gfs_function_expression) for where a function ends.
// This is an independent reader over the characters of the file; it keeps the reference's
// conventions where they decide the meaning of a file:
//   * a new object starts on a new line at brace depth 0;
//   * a function is a number, a variable name, a parenthesised / operator-chained C expression
//     (it goes on while inside parentheses, across spaces that are followed by an operator or
//     an opening parenthesis, and after an operator), or a { C block };
//   * class names are accepted with or without the `Gfs` prefix.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <stdexcept>
#include <cctype>
#include <cstdlib>
#include <cstring>

namespace gfs {

struct ParseError : std::runtime_error {
  explicit ParseError (const std::string & m) : std::runtime_error (m) {}
};

// the text of a function and how it was written
struct FunctionText {
  std::string text;      // "{ ... }" block, or an expression
  bool block = false;
  bool empty () const { return text.empty (); }
};

class Reader {
public:
  Reader (const std::string & text, const std::string & name, int first_line = 1)
    : s_ (text), name_ (name), line_ (first_line) {}

  bool eof () { skip_space (true); return p_ >= s_.size (); }
  int line () const { return line_; }
  size_t offset () const { return p_; }        // position in the text (for callers that cut it)

  [[noreturn]] void fail (const std::string & msg) const {
    throw ParseError (name_ + ":" + std::to_string (line_) + ": " + msg);
  }

  // spaces, comments and (optionally) newlines
  void skip_space (bool newlines) {
    while (p_ < s_.size ()) {
      char c = s_[p_];
      if (c == '#') {
	while (p_ < s_.size () && s_[p_] != '\n') p_++;
      }
      else if (c == '\n') {
	if (!newlines) return;
	line_++; p_++;
      }
      else if (c == ' ' || c == '\t' || c == '\r' || c == '\f')
	p_++;
      else
	return;
    }
  }

  // next significant character without consuming it (0 at end of text)
  char peek (bool newlines = true) {
    skip_space (newlines);
    return p_ < s_.size () ? s_[p_] : 0;
  }

  bool at_end_of_object () {
    char c = peek (false);
    return c == 0 || c == '\n';
  }

  void expect (char c) {
    if (peek () != c)
      fail (std::string ("expecting `") + c + "'");
    p_++;
  }

  bool accept (char c) {
    if (peek () == c) { p_++; return true; }
    return false;
  }

  // a word: everything up to a space or one of { } = newline
  std::string word (bool newlines = true) {
    skip_space (newlines);
    size_t b = p_;
    if (p_ < s_.size () && s_[p_] == '"') {      // quoted string
      p_++;
      while (p_ < s_.size () && s_[p_] != '"') { if (s_[p_] == '\n') line_++; p_++; }
      if (p_ >= s_.size ()) fail ("unterminated string");
      p_++;
      return s_.substr (b + 1, p_ - b - 2);
    }
    while (p_ < s_.size () && !strchr (" \t\r\f\n{}=#", s_[p_]))
      p_++;
    if (p_ == b)
      fail ("expecting a word");
    return s_.substr (b, p_ - b);
  }

  double number () {
    std::string w = word ();
    char * end;
    double v = strtod (w.c_str (), &end);
    if (*end != '\0')
      fail ("expecting a number, got `" + w + "'");
    return v;
  }

  // the raw text between balanced braces (the braces are consumed, not returned)
  std::string braces () {
    expect ('{');
    size_t b = p_;
    int depth = 1;
    while (p_ < s_.size ()) {
      char c = s_[p_];
      if (c == '\n') line_++;
      if (c == '{') depth++;
      else if (c == '}' && --depth == 0) {
	std::string r = s_.substr (b, p_ - b);
	p_++;
	return r;
      }
      p_++;
    }
    fail ("unbalanced braces");
  }

  // { key = value ... } into a map (values are words)
  std::map<std::string, std::string> assignments () {
    std::map<std::string, std::string> m;
    int l0 = line_;
    Reader r (braces (), name_, l0);
    while (!r.eof ()) {
      std::string k = r.word ();
      r.expect ('=');
      m[k] = r.word ();
    }
    return m;
  }

  // a GfsFunction (src/utils.c:868-983)
  FunctionText function () {
    static const char operators[] = "+-*/%<>=&^|?:!";
    FunctionText f;
    if (peek () == '{') {
      f.block = true;
      f.text = "{" + braces () + "}";
      return f;
    }
    int scope = 0;
    std::string & e = f.text;
    while (p_ < s_.size ()) {
      char c = s_[p_];
      if (scope > 0) {
	if (c == '(') scope++;
	else if (c == ')') scope--;
	if (c == '\n') line_++;
	e += c; p_++;
	continue;
      }
      if (c == '{' || c == '}' || c == '\n' || c == '#')
	break;
      if (c == ' ' || c == '\t' || c == '\r' || c == '\f') {
	size_t q = p_;
	while (q < s_.size () && strchr (" \t\r\f", s_[q])) q++;
	char d = q < s_.size () ? s_[q] : 0;
	bool constant = is_number (e);
	if (d == '(' && !e.empty () && !constant) {
	  e.append (s_, p_, q - p_ + 1);
	  p_ = q + 1;
	  scope++;
	}
	else if (d != 0 && strchr (operators, d) && !e.empty ()) {
	  // `a - b` goes on; but `U 1e-4 -1` style lists do not occur in this grammar
	  e.append (s_, p_, q - p_ + 1);
	  p_ = q + 1;
	  while (p_ < s_.size () && strchr (" \t\r\f", s_[p_])) e += s_[p_++];
	}
	else
	  break;
	continue;
      }
      if (strchr (operators, c)) {
	e += c; p_++;
	while (p_ < s_.size () && strchr (" \t\r\f", s_[p_])) e += s_[p_++];
	continue;
      }
      if (c == '(') scope++;
      e += c; p_++;
    }
    if (scope != 0) fail ("unbalanced parentheses in expression");
    while (!e.empty () && isspace ((unsigned char) e.back ())) e.pop_back ();
    if (e.empty ()) fail ("expecting an expression");
    return f;
  }

  // the rest of the current object: up to a newline at brace depth 0
  std::string rest_of_object () {
    size_t b = p_;
    int depth = 0;
    while (p_ < s_.size ()) {
      char c = s_[p_];
      if (c == '{') depth++;
      else if (c == '}') { if (depth == 0) break; depth--; }
      else if (c == '\n') { if (depth == 0) break; line_++; }
      else if (c == '#' && depth == 0) {
	while (p_ < s_.size () && s_[p_] != '\n') p_++;
	continue;
      }
      p_++;
    }
    return s_.substr (b, p_ - b);
  }

  static bool is_number (const std::string & w) {
    if (w.empty ()) return false;
    char * end;
    strtod (w.c_str (), &end);
    return *end == '\0';
  }

private:
  std::string s_, name_;
  size_t p_ = 0;
  int line_;
};

inline std::string strip_gfs (const std::string & cls) {
  return cls.compare (0, 3, "Gfs") == 0 ? cls.substr (3) : cls;
}

} // namespace gfs
