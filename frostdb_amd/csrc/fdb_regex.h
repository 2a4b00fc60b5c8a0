// fdb_regex.h — the library's built-in regular-expression engine: RE2 SYNTAX (Go's regexp, which the reference compiles `=~` / `!~`
// literals with, filter.go:105-124), matched the way regexp.Regexp.Match does (unanchored, on the value's bytes read as UTF-8,
// regexpfilter.go:84-166). Used once per DISTINCT value of a filtered column (per dictionary entry), never per row, when the host
// application passes no matcher of its own (fdb_plan_desc.regex_match == NULL).
//
// Covered: literals and escapes (\n \t \x41 \x{1F600} \123 octal, \Q…\E, escaped punctuation), `.`, character classes with ranges,
// negation, Perl classes \d \D \s \S \w \W and POSIX classes [[:alpha:]] / [[:^alpha:]], anchors ^ $ \A \z, word boundaries \b \B,
// capturing / non-capturing / named groups, alternation, repetition * + ? {n} {n,} {n,m} (lazy forms accepted: laziness does not
// change WHETHER something matches), flags i m s U — set `(?i)`, scoped `(?i:…)`, negated `(?-i)`. Like RE2: no backreferences, no
// look-around (both are syntax errors), a repeat count above 1 000 is an error, matching is a Thompson / Pike simulation —
// linear in the value's length, no backtracking blow-up whatever the pattern.
// Unicode: general-category classes \pL \p{Lu} \P{Nd} \p{^Zs} \p{Any} (inside brackets too), and (?i) folds by case-folding ORBITS the
// way regexp/syntax does with unicode.SimpleFold (k ↔ K ↔ U+212A, σ ↔ ς ↔ Σ, ß ↔ ẞ); tables generated from Unicode 13.0
// (tools/gen_unicode_tables.py → fdb_unicode_tables.inc; Go 1.22 carries 15.0: runes assigned since are unassigned here).
// Not covered (a pattern using them does not compile → FDB_ERR_INVALID at fdb_plan_create, like any bad pattern): Unicode SCRIPT
// classes (\p{Greek}, \p{Han} …).
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace fdb {

class Regex {
 public:
  // nullptr + *error ("regexp compile: …") when the pattern is not valid RE2 syntax (or uses what is not covered).
  static std::shared_ptr<const Regex> compile(const std::string& pattern, std::string* error);
  // ≙ regexp.Regexp.Match: does the pattern match anywhere in s[0, n)?
  bool match(const char* s, size_t n) const;
  bool match(const std::string& s) const { return match(s.data(), s.size()); }

  struct Range { uint32_t lo, hi; };
  enum Op : uint8_t { CLASS, ANY, ANY_NOT_NL, ASSERT, SPLIT, JMP, MATCH };
  enum Assert : uint32_t { BEGIN_TEXT, END_TEXT, BEGIN_LINE, END_LINE, WORD_B, NOT_WORD_B };
  struct Inst { Op op; uint32_t x, y; };  // CLASS: x = class index; ASSERT: x = kind; SPLIT: x, y = targets (x preferred); JMP: x

 private:
  std::vector<Inst> prog_;
  std::vector<std::vector<Range>> classes_;
  friend struct RegexCompiler;
};

}  // namespace fdb
