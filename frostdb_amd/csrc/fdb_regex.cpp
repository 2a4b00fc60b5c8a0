// fdb_regex.cpp — see fdb_regex.h. Parser (regexp/syntax's Perl flavour) → syntax tree → Thompson program → Pike simulation.
#include "fdb_regex.h"

#include <algorithm>
#include <cstring>
#include <stdexcept>

namespace fdb {

namespace {

constexpr uint32_t kMaxRune = 0x10FFFF;
constexpr int kMaxRepeat = 1000;        // regexp/syntax: "invalid repeat count"
constexpr size_t kMaxProgram = 200000;  // instructions (RE2 bounds the program too: a{1000}{1000} must not compile into gigabytes)

struct ParseError : std::runtime_error { using std::runtime_error::runtime_error; };

typedef Regex::Range Range;

void normalise(std::vector<Range>* r) {
  std::sort(r->begin(), r->end(), [](const Range& a, const Range& b) { return a.lo < b.lo; });
  std::vector<Range> out;
  for (const Range& x : *r) {
    if (!out.empty() && x.lo <= out.back().hi + 1 && out.back().hi != kMaxRune) out.back().hi = std::max(out.back().hi, x.hi);
    else if (!out.empty() && out.back().hi == kMaxRune) break;
    else out.push_back(x);
  }
  r->swap(out);
}
std::vector<Range> negate(std::vector<Range> r) {
  normalise(&r);
  std::vector<Range> out;
  uint32_t next = 0;
  for (const Range& x : r) {
    if (x.lo > next) out.push_back(Range{next, x.lo - 1});
    if (x.hi == kMaxRune) return out;
    next = x.hi + 1;
  }
  out.push_back(Range{next, kMaxRune});
  return out;
}
void add_fold(std::vector<Range>* r) {  // ASCII case folding: every letter range also in the other case
  const size_t n = r->size();
  for (size_t i = 0; i < n; i++) {
    const Range x = (*r)[i];
    const uint32_t lo1 = std::max<uint32_t>(x.lo, 'a'), hi1 = std::min<uint32_t>(x.hi, 'z');
    if (lo1 <= hi1) r->push_back(Range{lo1 - 32, hi1 - 32});
    const uint32_t lo2 = std::max<uint32_t>(x.lo, 'A'), hi2 = std::min<uint32_t>(x.hi, 'Z');
    if (lo2 <= hi2) r->push_back(Range{lo2 + 32, hi2 + 32});
  }
}
const std::vector<Range>& perl_class(char c) {  // \d \s \w (ASCII, like Go without the Unicode tables)
  static const std::vector<Range> d{{'0', '9'}}, s{{'\t', '\n'}, {'\f', '\r'}, {' ', ' '}}, w{{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}};
  return c == 'd' ? d : c == 's' ? s : w;
}
bool posix_class(const std::string& name, std::vector<Range>* out) {
  struct P { const char* n; std::vector<Range> r; };
  static const P table[] = {
      {"alnum", {{'0', '9'}, {'A', 'Z'}, {'a', 'z'}}}, {"alpha", {{'A', 'Z'}, {'a', 'z'}}}, {"ascii", {{0, 0x7F}}}, {"blank", {{'\t', '\t'}, {' ', ' '}}},
      {"cntrl", {{0, 0x1F}, {0x7F, 0x7F}}}, {"digit", {{'0', '9'}}}, {"graph", {{'!', '~'}}}, {"lower", {{'a', 'z'}}}, {"print", {{' ', '~'}}},
      {"punct", {{'!', '/'}, {':', '@'}, {'[', '`'}, {'{', '~'}}}, {"space", {{'\t', '\r'}, {' ', ' '}}}, {"upper", {{'A', 'Z'}}},
      {"word", {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}}, {"xdigit", {{'0', '9'}, {'A', 'F'}, {'a', 'f'}}}};
  for (const P& p : table) if (name == p.n) { *out = p.r; return true; }
  return false;
}

// ---- syntax tree ------------------------------------------------------------------------------------------------------------------
struct Node {
  enum Kind { EMPTY, CLASS, ANY, ANY_NOT_NL, ASSERT, CAT, ALT, REPEAT } kind = EMPTY;
  std::vector<Range> ranges;           // CLASS
  uint32_t assert_kind = 0;            // ASSERT
  std::vector<std::unique_ptr<Node>> kids;  // CAT / ALT: any number; REPEAT: one
  int min = 0, max = -1;               // REPEAT: max −1 = unbounded
};
typedef std::unique_ptr<Node> NodeP;
NodeP mk(Node::Kind k) { NodeP n(new Node()); n->kind = k; return n; }

struct Flags { bool i = false, m = false, s = false; };  // (U only changes which match is preferred: irrelevant here)

struct Parser {
  const std::string& p;
  size_t at = 0;
  int depth = 0;
  explicit Parser(const std::string& pat) : p(pat) {}
  bool end() const { return at >= p.size(); }
  [[noreturn]] void fail(const std::string& what) const { throw ParseError(what); }

  uint32_t rune() {  // next UTF-8 rune of the PATTERN (invalid bytes are taken as themselves: Go rejects them, "invalid UTF-8")
    const unsigned char c = (unsigned char)p[at];
    if (c < 0x80) { at++; return c; }
    int n = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : -1;
    if (n < 0 || at + (size_t)n + 1 > p.size()) fail("invalid UTF-8");
    uint32_t r = c & (0x3F >> n);
    for (int k = 1; k <= n; k++) {
      const unsigned char d = (unsigned char)p[at + (size_t)k];
      if ((d & 0xC0) != 0x80) fail("invalid UTF-8");
      r = (r << 6) | (d & 0x3F);
    }
    at += (size_t)n + 1;
    return r;
  }

  NodeP literal(uint32_t r, const Flags& f) {
    NodeP n = mk(Node::CLASS);
    n->ranges.push_back(Range{r, r});
    if (f.i) { add_fold(&n->ranges); normalise(&n->ranges); }
    return n;
  }

  // after a backslash; class escapes return true and fill *cls, single runes return false and set *r
  bool escape(bool in_class, uint32_t* r, std::vector<Range>* cls, NodeP* assert_out) {
    if (end()) fail("trailing backslash at end of expression");
    const char c = p[at++];
    switch (c) {
      case 'a': *r = 7; return false;
      case 'f': *r = '\f'; return false;
      case 't': *r = '\t'; return false;
      case 'n': *r = '\n'; return false;
      case 'r': *r = '\r'; return false;
      case 'v': *r = '\v'; return false;
      case 'd': case 's': case 'w': *cls = perl_class(c); return true;
      case 'D': case 'S': case 'W': *cls = negate(perl_class((char)(c + 32))); return true;
      case 'p': case 'P': fail("invalid character class range: Unicode classes (\\p) are not supported by the built-in engine");
      case 'x': {
        if (end()) fail("invalid escape sequence: `\\x`");
        uint32_t v = 0;
        if (p[at] == '{') {
          at++;
          int digits = 0;
          while (!end() && p[at] != '}') {
            const int h = hex(p[at++]);
            if (h < 0 || ++digits > 8) fail("invalid escape sequence: `\\x{`");
            v = v * 16 + (uint32_t)h;
          }
          if (end() || digits == 0 || v > kMaxRune) fail("invalid escape sequence: `\\x{`");
          at++;
        } else {
          if (at + 2 > p.size() || hex(p[at]) < 0 || hex(p[at + 1]) < 0) fail("invalid escape sequence: `\\x`");
          v = (uint32_t)(hex(p[at]) * 16 + hex(p[at + 1]));
          at += 2;
        }
        *r = v;
        return false;
      }
      case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': {
        // \1 … \7 alone would be backreferences (not in RE2); \0 and \12 / \123 are octal
        if (c != '0' && (end() || p[at] < '0' || p[at] > '7')) fail(std::string("invalid escape sequence: `\\") + c + "`");
        uint32_t v = (uint32_t)(c - '0');
        for (int k = 0; k < 2 && !end() && p[at] >= '0' && p[at] <= '7'; k++) v = v * 8 + (uint32_t)(p[at++] - '0');
        *r = v;
        return false;
      }
      default:
        break;
    }
    if (!in_class && assert_out != nullptr) {
      uint32_t k = 0xFFFFFFFFu;
      if (c == 'A') k = Regex::BEGIN_TEXT; else if (c == 'z') k = Regex::END_TEXT; else if (c == 'b') k = Regex::WORD_B; else if (c == 'B') k = Regex::NOT_WORD_B;
      if (k != 0xFFFFFFFFu) { *assert_out = mk(Node::ASSERT); (*assert_out)->assert_kind = k; return false; }
    }
    const unsigned char uc = (unsigned char)c;
    if (uc < 0x80 && !((uc >= '0' && uc <= '9') || (uc >= 'A' && uc <= 'Z') || (uc >= 'a' && uc <= 'z'))) { *r = uc; return false; }  // escaped punctuation
    fail(std::string("invalid escape sequence: `\\") + c + "`");
  }
  static int hex(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

  NodeP char_class(const Flags& f) {  // after '['
    NodeP n = mk(Node::CLASS);
    bool neg = false;
    if (!end() && p[at] == '^') { neg = true; at++; }
    bool first = true;
    for (;;) {
      if (end()) fail("missing closing ]");
      if (p[at] == ']' && !first) { at++; break; }
      first = false;
      if (p[at] == '[' && at + 1 < p.size() && p[at + 1] == ':') {  // [:alpha:] / [:^alpha:]
        const size_t close = p.find(":]", at + 2);
        if (close != std::string::npos) {
          std::string name = p.substr(at + 2, close - at - 2);
          bool pneg = false;
          if (!name.empty() && name[0] == '^') { pneg = true; name.erase(0, 1); }
          std::vector<Range> pc;
          if (!posix_class(name, &pc)) fail("invalid character class range: `[:" + name + ":]`");
          if (pneg) pc = negate(pc);
          n->ranges.insert(n->ranges.end(), pc.begin(), pc.end());
          at = close + 2;
          continue;
        }
      }
      uint32_t lo = 0;
      std::vector<Range> cls;
      bool is_cls = false;
      if (p[at] == '\\') { at++; is_cls = escape(true, &lo, &cls, nullptr); }
      else lo = rune();
      if (is_cls) { n->ranges.insert(n->ranges.end(), cls.begin(), cls.end()); continue; }
      uint32_t hi = lo;
      if (at + 1 < p.size() && p[at] == '-' && p[at + 1] != ']') {
        at++;
        std::vector<Range> c2;
        if (p[at] == '\\') { at++; if (escape(true, &hi, &c2, nullptr)) fail("invalid character class range"); }
        else hi = rune();
        if (hi < lo) fail("invalid character class range");
      }
      n->ranges.push_back(Range{lo, hi});
    }
    if (f.i) add_fold(&n->ranges);
    normalise(&n->ranges);
    if (neg) n->ranges = negate(n->ranges);
    return n;
  }

  bool repeat_counts(int* mn, int* mx) {  // at '{': parses {n} {n,} {n,m}; false (nothing consumed) if this is a literal brace
    size_t q = at + 1;
    auto num = [&](int* v) -> bool {
      if (q >= p.size() || p[q] < '0' || p[q] > '9') return false;
      long x = 0;
      while (q < p.size() && p[q] >= '0' && p[q] <= '9') { x = x * 10 + (p[q++] - '0'); if (x > 100000) x = 100000; }
      *v = (int)x;
      return true;
    };
    if (!num(mn)) return false;
    *mx = *mn;
    if (q < p.size() && p[q] == ',') { q++; if (q < p.size() && p[q] == '}') *mx = -1; else if (!num(mx)) return false; }
    if (q >= p.size() || p[q] != '}') return false;
    at = q + 1;
    if (*mn > kMaxRepeat || *mx > kMaxRepeat || (*mx >= 0 && *mx < *mn)) fail("invalid repeat count");
    return true;
  }

  NodeP alternation(Flags f, bool in_group) {  // until ')' (in a group) or the end
    if (++depth > 200) fail("expression nests too deeply");
    std::vector<NodeP> alts;
    std::vector<NodeP> seq;
    auto close_seq = [&] {
      NodeP cat = mk(Node::CAT);
      cat->kids = std::move(seq);
      seq.clear();
      alts.push_back(std::move(cat));
    };
    bool can_repeat = false;  // the previous item can take a repetition operator
    for (;;) {
      if (end()) { if (in_group) fail("missing closing )"); break; }
      const char c = p[at];
      if (c == ')') { if (!in_group) fail("unexpected )"); at++; break; }
      if (c == '|') { at++; close_seq(); can_repeat = false; continue; }
      if (c == '*' || c == '+' || c == '?' || c == '{') {
        int mn = 0, mx = -1;
        bool is_rep = true;
        if (c == '{') is_rep = repeat_counts(&mn, &mx);
        else { at++; mn = c == '+' ? 1 : 0; mx = c == '?' ? 1 : -1; }
        if (is_rep) {
          if (!can_repeat || seq.empty()) fail(std::string("missing argument to repetition operator: `") + c + "`");
          if (!end() && p[at] == '?') at++;  // lazy: same set of matching inputs
          if (!end() && (p[at] == '*' || p[at] == '+' || p[at] == '?')) fail("invalid nested repetition operator");
          NodeP rep = mk(Node::REPEAT);
          rep->min = mn; rep->max = mx;
          rep->kids.push_back(std::move(seq.back()));
          seq.back() = std::move(rep);
          can_repeat = false;
          continue;
        }
        // a literal '{'
        at++;
        seq.push_back(literal('{', f));
        can_repeat = true;
        continue;
      }
      can_repeat = true;
      if (c == '(') {
        at++;
        Flags inner = f;
        bool consumed_flags_only = false;
        if (!end() && p[at] == '?') {
          at++;
          if (end()) fail("missing closing )");
          if (p[at] == 'P' || p[at] == '<') {  // (?P<name>…) / (?<name>…)
            if (p[at] == 'P') at++;
            if (end() || p[at] != '<') fail("invalid named capture");
            const size_t gt = p.find('>', at);
            if (gt == std::string::npos || gt == at + 1) fail("invalid named capture");
            for (size_t k = at + 1; k < gt; k++) { const char ch = p[k]; if (!(ch == '_' || (ch >= '0' && ch <= '9') || (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'))) fail("invalid named capture"); }
            at = gt + 1;
          } else {  // flags: (?i) (?i:…) (?-s) (?im-s:…)
            bool on = true, any = false;
            for (;;) {
              if (end()) fail("missing closing )");
              const char fc = p[at++];
              if (fc == 'i') { inner.i = on; any = true; }
              else if (fc == 'm') { inner.m = on; any = true; }
              else if (fc == 's') { inner.s = on; any = true; }
              else if (fc == 'U') { any = true; }
              else if (fc == '-') { if (!on) fail("invalid or unsupported Perl syntax"); on = false; any = false; }
              else if (fc == ':') { if (!on && !any) fail("invalid or unsupported Perl syntax"); break; }
              else if (fc == ')') { if (!on && !any) fail("invalid or unsupported Perl syntax"); consumed_flags_only = true; break; }
              else fail("invalid or unsupported Perl syntax");  // (?=…) (?!…) (?<=…) … : no look-around in RE2
            }
          }
        }
        if (consumed_flags_only) { f = inner; can_repeat = false; continue; }  // flags for the rest of the enclosing group
        seq.push_back(alternation(inner, true));
        continue;
      }
      if (c == '[') { at++; seq.push_back(char_class(f)); continue; }
      if (c == '.') { at++; seq.push_back(mk(f.s ? Node::ANY : Node::ANY_NOT_NL)); continue; }
      if (c == '^') { at++; NodeP a = mk(Node::ASSERT); a->assert_kind = f.m ? Regex::BEGIN_LINE : Regex::BEGIN_TEXT; seq.push_back(std::move(a)); continue; }
      if (c == '$') { at++; NodeP a = mk(Node::ASSERT); a->assert_kind = f.m ? Regex::END_LINE : Regex::END_TEXT; seq.push_back(std::move(a)); continue; }
      if (c == '\\') {
        at++;
        if (!end() && p[at] == 'Q') {  // \Q…\E: everything literal
          at++;
          while (!end() && !(p[at] == '\\' && at + 1 < p.size() && p[at + 1] == 'E')) seq.push_back(literal(rune(), f));
          if (!end()) at += 2;
          continue;
        }
        uint32_t r = 0;
        std::vector<Range> cls;
        NodeP as;
        if (escape(false, &r, &cls, &as)) {
          NodeP n = mk(Node::CLASS);
          n->ranges = cls;
          if (f.i) add_fold(&n->ranges);
          normalise(&n->ranges);
          seq.push_back(std::move(n));
        } else if (as) {
          seq.push_back(std::move(as));
        } else {
          seq.push_back(literal(r, f));
        }
        continue;
      }
      seq.push_back(literal(rune(), f));
    }
    close_seq();
    depth--;
    if (alts.size() == 1) return std::move(alts[0]);
    NodeP alt = mk(Node::ALT);
    alt->kids = std::move(alts);
    return alt;
  }
};

}  // namespace

// ---- program ----------------------------------------------------------------------------------------------------------------------
struct RegexCompiler {
  Regex* re;
  void emit(Regex::Op op, uint32_t x = 0, uint32_t y = 0) {
    if (re->prog_.size() >= kMaxProgram) throw ParseError("expression too large");
    re->prog_.push_back(Regex::Inst{op, x, y});
  }
  uint32_t pc() const { return (uint32_t)re->prog_.size(); }
  void gen(const Node& n) {
    switch (n.kind) {
      case Node::EMPTY: return;
      case Node::CLASS: re->classes_.push_back(n.ranges); emit(Regex::CLASS, (uint32_t)re->classes_.size() - 1); return;
      case Node::ANY: emit(Regex::ANY); return;
      case Node::ANY_NOT_NL: emit(Regex::ANY_NOT_NL); return;
      case Node::ASSERT: emit(Regex::ASSERT, n.assert_kind); return;
      case Node::CAT: for (const NodeP& k : n.kids) gen(*k); return;
      case Node::ALT: {
        std::vector<uint32_t> jumps;
        for (size_t i = 0; i < n.kids.size(); i++) {
          if (i + 1 < n.kids.size()) {
            const uint32_t split = pc();
            emit(Regex::SPLIT);
            re->prog_[split].x = pc();
            gen(*n.kids[i]);
            jumps.push_back(pc());
            emit(Regex::JMP);
            re->prog_[split].y = pc();
          } else gen(*n.kids[i]);
        }
        for (uint32_t j : jumps) re->prog_[j].x = pc();
        return;
      }
      case Node::REPEAT: {
        const Node& k = *n.kids[0];
        for (int i = 0; i < n.min; i++) gen(k);
        if (n.max < 0) {  // k*: L: split(body, out); body; jmp L
          const uint32_t split = pc();
          emit(Regex::SPLIT);
          re->prog_[split].x = pc();
          gen(k);
          emit(Regex::JMP, split);
          re->prog_[split].y = pc();
        } else {
          std::vector<uint32_t> splits;
          for (int i = n.min; i < n.max; i++) {  // (k(k(k)?)?)?
            splits.push_back(pc());
            emit(Regex::SPLIT);
            re->prog_[splits.back()].x = pc();
            gen(k);
          }
          for (uint32_t s : splits) re->prog_[s].y = pc();
        }
        return;
      }
    }
  }
};

std::shared_ptr<const Regex> Regex::compile(const std::string& pattern, std::string* error) {
  std::shared_ptr<Regex> re(new Regex());
  try {
    Parser ps(pattern);
    NodeP root = ps.alternation(Flags(), false);
    RegexCompiler c{re.get()};
    c.gen(*root);
    c.emit(Regex::MATCH);
  } catch (const ParseError& e) {
    if (error) *error = std::string("regexp compile: error parsing regexp: ") + e.what() + ": `" + pattern + "`";
    return nullptr;
  }
  return re;
}

// ---- matching -----------------------------------------------------------------------------------------------------------------------
namespace {
inline bool is_word(uint32_t r) { return (r >= '0' && r <= '9') || (r >= 'A' && r <= 'Z') || (r >= 'a' && r <= 'z') || r == '_'; }
constexpr uint32_t kNoRune = 0xFFFFFFFFu;
// the rune at s[i] (Go's utf8.DecodeRune: an invalid or truncated sequence is U+FFFD of width 1)
inline uint32_t decode(const unsigned char* s, size_t n, size_t i, size_t* width) {
  const unsigned char c = s[i];
  if (c < 0x80) { *width = 1; return c; }
  int k = c >= 0xF0 && c <= 0xF4 ? 3 : c >= 0xE0 ? 2 : c >= 0xC2 ? 1 : -1;
  if (k < 0 || i + (size_t)k >= n) { *width = 1; return 0xFFFD; }  // not a lead byte, or the sequence is cut off
  uint32_t r = c & (0x3F >> k);
  for (int j = 1; j <= k; j++) {
    const unsigned char d = s[i + (size_t)j];
    if ((d & 0xC0) != 0x80) { *width = 1; return 0xFFFD; }
    r = (r << 6) | (d & 0x3F);
  }
  if ((k == 2 && (r < 0x800 || (r >= 0xD800 && r <= 0xDFFF))) || (k == 3 && (r < 0x10000 || r > kMaxRune))) { *width = 1; return 0xFFFD; }  // overlong / surrogate
  *width = (size_t)k + 1;
  return r;
}
}  // namespace

bool Regex::match(const char* sp, size_t n) const {
  const unsigned char* s = reinterpret_cast<const unsigned char*>(sp);
  const size_t np = prog_.size();
  std::vector<uint32_t> clist, nlist, stack;
  std::vector<uint32_t> mark(np, 0);  // generation in which a pc was last added
  uint32_t gen = 0;
  clist.reserve(np); nlist.reserve(np); stack.reserve(np);
  uint32_t prev = kNoRune;
  size_t i = 0;
  // adds pc and everything reachable through empty transitions at the position (prev | next) to `list`; true if MATCH is reachable
  auto add = [&](std::vector<uint32_t>& list, uint32_t pc0, uint32_t next_rune) -> bool {
    stack.clear();
    stack.push_back(pc0);
    while (!stack.empty()) {
      const uint32_t pc = stack.back();
      stack.pop_back();
      if (mark[pc] == gen) continue;
      mark[pc] = gen;
      const Inst& in = prog_[pc];
      switch (in.op) {
        case JMP: stack.push_back(in.x); break;
        case SPLIT: stack.push_back(in.y); stack.push_back(in.x); break;
        case ASSERT: {
          bool ok = false;
          switch (in.x) {
            case BEGIN_TEXT: ok = prev == kNoRune; break;
            case END_TEXT: ok = next_rune == kNoRune; break;
            case BEGIN_LINE: ok = prev == kNoRune || prev == '\n'; break;
            case END_LINE: ok = next_rune == kNoRune || next_rune == '\n'; break;
            case WORD_B: ok = (prev != kNoRune && is_word(prev)) != (next_rune != kNoRune && is_word(next_rune)); break;
            case NOT_WORD_B: ok = (prev != kNoRune && is_word(prev)) == (next_rune != kNoRune && is_word(next_rune)); break;
          }
          if (ok) stack.push_back(pc + 1);
          break;
        }
        case MATCH: return true;
        default: list.push_back(pc); break;
      }
    }
    return false;
  };
  for (;;) {
    size_t w = 0;
    const uint32_t r = i < n ? decode(s, n, i, &w) : kNoRune;
    // a new attempt starts at every position (unanchored search); threads carried over were added with the same generation scheme
    gen++;
    nlist.clear();
    // carry: re-close the surviving threads at this position (their successors were queued as raw pcs)
    for (uint32_t pc : clist) if (add(nlist, pc, r)) return true;
    if (add(nlist, 0, r)) return true;
    if (r == kNoRune) return false;
    clist.clear();
    for (uint32_t pc : nlist) {
      const Inst& in = prog_[pc];
      bool take = false;
      if (in.op == ANY) take = true;
      else if (in.op == ANY_NOT_NL) take = r != '\n';
      else if (in.op == CLASS) {
        const std::vector<Range>& c = classes_[in.x];
        size_t lo = 0, hi = c.size();
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (c[mid].hi < r) lo = mid + 1; else hi = mid; }
        take = lo < c.size() && c[lo].lo <= r;
      }
      if (take) clist.push_back(pc + 1);
    }
    prev = r;
    i += w;
  }
}

}  // namespace fdb
