// Small RE2-style regular-expression engine (Pike VM, linear time, no backtracking) for re_match / regex.match.
// The reference evaluates these through Go's regexp package inside OPA (third-party, regexp/syntax Perl flags); this
// engine follows that syntax and matches UTF-8 text rune by rune while running on BYTES (a rune set compiles to an
// alternation of UTF-8 byte sequences), so that the same program becomes the byte-class DFA the device runs (P_REGEX).
//
// Three outcomes for a pattern, never an approximation:
//   * compiles           literals, '.', classes (ranges, negation, \d \w \s, [[:alpha:]] ...), ^ $ \A \z \b \B, groups
//                        (capturing, (?:..), (?P<n>..)), alternation, * + ? {m,n} (greedy/lazy are equivalent for
//                        matching), flags i m s U ((?i), (?i:..), (?-i)), \x41 \x{263a} \012 \Q..\E escapes
//   * RegexError         the pattern is INVALID in Go (regexp.Compile fails): re_match is a builtin error -> undefined
//   * RegexUnsupported   the pattern is valid Go but outside this engine (\pL Unicode classes, non-ASCII runes in ranges or
//                        under (?i), negated classes that fold onto non-ASCII runes, repeat counts > 100): the caller reports
//                        GK_ERR_UNSUPPORTED at AddConstraint -- the constraint stays on the reference CPU driver
#pragma once
#include <algorithm>
#include <bitset>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gk {

struct RegexError : std::runtime_error { using std::runtime_error::runtime_error; };
struct RegexUnsupported : std::runtime_error { using std::runtime_error::runtime_error; };

class Regex {
 public:
  explicit Regex(const std::string& pat) : p_(pat) {
    Frag f = parse_alt();
    if (i_ != p_.size()) throw RegexError("regex: unexpected ')'");
    patch(f.out, (int)prog_.size());
    emit(MATCH);
    start_ = f.start;
  }

  // unanchored search (Go regexp.MatchString)
  bool search(const std::string& s) const {
    std::vector<int> carried, cur, nxt;
    std::vector<uint32_t> mark(prog_.size(), 0);
    uint32_t gen = 0;
    size_t n = s.size();
    for (size_t pos = 0;; pos++) {
      Ctx cx;
      cx.at_start = pos == 0; cx.at_end = pos == n;
      if (pos > 0) { unsigned char p = (unsigned char)s[pos - 1]; cx.prev_word = is_word(p); cx.prev_nl = p == '\n'; }
      if (pos < n) { unsigned char c = (unsigned char)s[pos]; cx.next_word = is_word(c); cx.next_nl = c == '\n'; }
      gen++;
      cur.clear();
      for (int pc : carried) add(cur, pc, cx, mark, gen);
      add(cur, start_, cx, mark, gen);   // a new thread may start at every position
      nxt.clear();
      for (int pc : cur) {
        const Inst& in = prog_[pc];
        if (in.op == MATCH) return true;
        if (pos < n) {
          unsigned char c = (unsigned char)s[pos];
          if ((in.op == CHAR && c == in.c) || (in.op == CLASS && classes_[in.x][c])) nxt.push_back(pc + 1);
        }
      }
      if (pos >= n) return false;
      carried.swap(nxt);
    }
  }

  // The same search as a byte-class DFA for the device (P_REGEX, vm_core.hpp):
  //   table = [u32 n_states][u32 n_classes][u8 class_of_byte[256]][u8 accept_at_end[n_states]][u8 next[n_states][n_classes]]
  // state 0 = start of input; an early match moves to an absorbing accepting state.  A state is the set of threads
  // carried over from the previous byte plus what the zero-width assertions need to know about that byte (start of
  // input / newline / word byte / other); `accept_at_end` applies the end-of-input assertions.
  // Returns false when the automaton needs more than `max_states` states (the caller reports "unsupported").
  bool to_dfa(std::vector<uint8_t>* table, size_t max_states = 255) const {
    // byte equivalence classes
    std::vector<int> cls(256, 0);
    int ncls = 1;
    auto refine = [&](const std::bitset<256>& bs) {
      std::map<std::pair<int, bool>, int> remap;
      for (int b = 0; b < 256; b++) {
        auto key = std::make_pair(cls[b], (bool)bs[b]);
        auto it = remap.find(key);
        if (it == remap.end()) it = remap.emplace(key, (int)remap.size()).first;
        cls[b] = it->second;
      }
      ncls = (int)remap.size();
    };
    bool uses_word = false, uses_nl = false;
    for (const Inst& in : prog_) {
      std::bitset<256> bs;
      if (in.op == CHAR) bs.set(in.c);
      else if (in.op == CLASS) bs = classes_[in.x];
      else { uses_word = uses_word || in.op == WORDB || in.op == NWORDB; uses_nl = uses_nl || in.op == BOL_ML || in.op == EOL_ML; continue; }
      refine(bs);
    }
    if (uses_word) { std::bitset<256> w; for (int b = 0; b < 256; b++) if (is_word((unsigned char)b)) w.set(b); refine(w); }
    if (uses_nl) { std::bitset<256> nl; nl.set('\n'); refine(nl); }
    if (ncls > 255) return false;
    std::vector<int> rep(ncls, -1);
    for (int b = 0; b < 256; b++) if (rep[cls[b]] < 0) rep[cls[b]] = b;
    enum Prev { P_START = 0, P_NL = 1, P_WORD = 2, P_OTHER = 3 };
    typedef std::pair<std::vector<int>, int> Key;   // carried threads (sorted), kind of the previous byte
    std::map<Key, int> ids;
    std::vector<Key> states;
    auto intern = [&](const Key& k) { auto it = ids.find(k); if (it != ids.end()) return it->second; ids[k] = (int)states.size(); states.push_back(k); return (int)states.size() - 1; };
    auto prev_kind = [&](unsigned char b) -> int {   // only as fine as the assertions in the program need
      if (uses_nl && b == '\n') return P_NL;
      if (uses_word && is_word(b)) return P_WORD;
      return P_OTHER;
    };
    auto closure = [&](const Key& k, bool at_end, unsigned char next, std::vector<int>* cur) {
      std::vector<uint32_t> mark(prog_.size(), 0);
      cur->clear();
      Ctx cx;
      cx.at_start = k.second == P_START; cx.at_end = at_end;
      cx.prev_word = k.second == P_WORD; cx.prev_nl = k.second == P_NL;
      if (!at_end) { cx.next_word = is_word(next); cx.next_nl = next == '\n'; }
      for (int pc : k.first) add(*cur, pc, cx, mark, 1);
      add(*cur, start_, cx, mark, 1);
    };
    auto has_match = [&](const std::vector<int>& cur) { for (int pc : cur) if (prog_[pc].op == MATCH) return true; return false; };
    intern(Key{{}, P_START});
    const int ACCEPT = -2;
    std::vector<std::vector<int>> next;
    std::vector<uint8_t> accept;
    std::vector<int> cur;
    for (size_t s = 0; s < states.size(); s++) {
      if (states.size() > max_states) return false;
      Key k = states[s];
      closure(k, true, 0, &cur);
      accept.push_back(has_match(cur) ? 1 : 0);
      std::vector<int> row(ncls, 0);
      for (int c = 0; c < ncls; c++) {
        unsigned char b = (unsigned char)rep[c];
        closure(k, false, b, &cur);
        if (has_match(cur)) { row[c] = ACCEPT; continue; }
        std::vector<int> nxt;
        for (int pc : cur) {
          const Inst& in = prog_[pc];
          if ((in.op == CHAR && b == in.c) || (in.op == CLASS && classes_[in.x][b])) nxt.push_back(pc + 1);
        }
        std::sort(nxt.begin(), nxt.end());
        nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
        row[c] = intern(Key{nxt, prev_kind(b)});
      }
      next.push_back(row);
    }
    if (states.size() + 1 > max_states) return false;
    const int acc_id = (int)states.size();   // absorbing accepting state
    const uint32_t ns = (uint32_t)states.size() + 1, nc = (uint32_t)ncls;
    table->clear();
    auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; k++) table->push_back((uint8_t)(v >> (8 * k))); };
    put32(ns); put32(nc);
    for (int b = 0; b < 256; b++) table->push_back((uint8_t)cls[b]);
    for (uint8_t a : accept) table->push_back(a);
    table->push_back(1);
    for (auto& row : next) for (int t : row) table->push_back((uint8_t)(t == ACCEPT ? acc_id : t));
    for (uint32_t c = 0; c < nc; c++) table->push_back((uint8_t)acc_id);
    return true;
  }

 private:
  enum Op { CHAR, CLASS, SPLIT, JMP, MATCH, BOT, EOT, BOL_ML, EOL_ML, WORDB, NWORDB };
  struct Inst { Op op; unsigned char c; int x, y; };
  struct Frag { int start; std::vector<int*> out; };
  struct Ctx { bool at_start = false, at_end = false, prev_word = false, prev_nl = false, next_word = false, next_nl = false; };
  struct Flags { bool fold = false, multiline = false, dotall = false; };
  // a set of runes: ASCII members, "every non-ASCII rune", or a few single non-ASCII runes
  struct RuneSet { std::bitset<128> ascii; bool rest = false; std::vector<uint32_t> extra; };

  std::string p_;
  size_t i_ = 0;
  Flags fl_;
  std::vector<Inst> prog_;
  std::vector<std::bitset<256>> classes_;
  int start_ = 0;
  std::vector<std::unique_ptr<int>> holes_;

  static bool is_word(unsigned char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }

  int emit(Op op, unsigned char c = 0, int x = -1, int y = -1) { prog_.push_back({op, c, x, y}); return (int)prog_.size() - 1; }

  void add(std::vector<int>& list, int pc, const Ctx& cx, std::vector<uint32_t>& mark, uint32_t gen) const {
    if (mark[pc] == gen) return;
    mark[pc] = gen;
    const Inst& in = prog_[pc];
    bool pass;
    switch (in.op) {
      case JMP: add(list, in.x, cx, mark, gen); return;
      case SPLIT: add(list, in.x, cx, mark, gen); add(list, in.y, cx, mark, gen); return;
      case BOT: pass = cx.at_start; break;
      case EOT: pass = cx.at_end; break;
      case BOL_ML: pass = cx.at_start || cx.prev_nl; break;
      case EOL_ML: pass = cx.at_end || cx.next_nl; break;
      case WORDB: pass = cx.prev_word != cx.next_word; break;
      case NWORDB: pass = cx.prev_word == cx.next_word; break;
      default: list.push_back(pc); return;
    }
    if (pass) add(list, pc + 1, cx, mark, gen);
  }

  // The program is built in a relocatable way: fragments are emitted in order, dangling exits are recorded as
  // (instruction index, field) pairs encoded in a side table.
  struct Hole { int inst; int field; };
  std::vector<Hole> hole_tab_;
  int* hole(int inst, int field) {
    hole_tab_.push_back({inst, field});
    holes_.emplace_back(new int((int)hole_tab_.size() - 1));
    return holes_.back().get();
  }
  void patch(const std::vector<int*>& outs, int target) {
    for (int* h : outs) {
      const Hole& ho = hole_tab_[*h];
      if (ho.field == 0) prog_[ho.inst].x = target; else prog_[ho.inst].y = target;
    }
  }

  bool more() const { return i_ < p_.size(); }
  char peek() const { return p_[i_]; }

  // ---- fragments
  Frag single(int inst) {   // one instruction that falls through to a JMP with a dangling target
    Frag f;
    f.start = inst;
    int j = emit(JMP);
    f.out = {hole(j, 0)};
    return f;
  }
  Frag empty_frag() { int j = emit(JMP); Frag f; f.start = j; f.out = {hole(j, 0)}; return f; }
  Frag byte_seq(const std::vector<std::bitset<256>>& bytes) {   // consecutive byte tests, then the exit
    int first = -1;
    for (const auto& b : bytes) {
      int pc;
      if (b.count() == 1) { int c = 0; while (!b[c]) c++; pc = emit(CHAR, (unsigned char)c); }
      else { classes_.push_back(b); pc = emit(CLASS, 0, (int)classes_.size() - 1); }
      if (first < 0) first = pc;
    }
    return single_from(first);
  }
  Frag single_from(int first) { Frag f; f.start = first; int j = emit(JMP); f.out = {hole(j, 0)}; return f; }
  Frag alternation(const std::vector<Frag>& alts) {
    if (alts.size() == 1) return alts[0];
    Frag out;
    int prev = -1;
    for (size_t k = 0; k + 1 < alts.size(); k++) {   // split0 -> alt0 | split1 -> alt1 | ... -> altN
      int sp = emit(SPLIT, 0, alts[k].start, -1);
      if (prev >= 0) prog_[prev].y = sp; else out.start = sp;
      prev = sp;
    }
    prog_[prev].y = alts.back().start;
    for (auto& a : alts) out.out.insert(out.out.end(), a.out.begin(), a.out.end());
    return out;
  }
  static std::bitset<256> range256(int a, int b) { std::bitset<256> s; for (int k = a; k <= b; k++) s.set(k); return s; }
  static std::string utf8(uint32_t r) {
    std::string o;
    if (r < 0x80) o.push_back((char)r);
    else if (r < 0x800) { o.push_back((char)(0xC0 | (r >> 6))); o.push_back((char)(0x80 | (r & 0x3F))); }
    else if (r < 0x10000) { o.push_back((char)(0xE0 | (r >> 12))); o.push_back((char)(0x80 | ((r >> 6) & 0x3F))); o.push_back((char)(0x80 | (r & 0x3F))); }
    else { o.push_back((char)(0xF0 | (r >> 18))); o.push_back((char)(0x80 | ((r >> 12) & 0x3F))); o.push_back((char)(0x80 | ((r >> 6) & 0x3F))); o.push_back((char)(0x80 | (r & 0x3F))); }
    return o;
  }
  Frag literal_bytes(const std::string& bytes) {
    std::vector<std::bitset<256>> seq;
    for (unsigned char c : bytes) { std::bitset<256> b; b.set(c); seq.push_back(b); }
    return byte_seq(seq);
  }
  // one rune out of `rs`, as an alternation of UTF-8 byte sequences
  Frag rune_set(const RuneSet& rs) {
    std::vector<Frag> alts;
    if (rs.ascii.any()) { std::bitset<256> b; for (int k = 0; k < 128; k++) if (rs.ascii[k]) b.set(k); alts.push_back(byte_seq({b})); }
    if (rs.rest) {   // every well-formed multi-byte sequence (Unicode Table 3-7)
      const std::bitset<256> cont = range256(0x80, 0xBF);
      alts.push_back(byte_seq({range256(0xC2, 0xDF), cont}));
      alts.push_back(byte_seq({range256(0xE0, 0xE0), range256(0xA0, 0xBF), cont}));
      alts.push_back(byte_seq({range256(0xE1, 0xEC) | range256(0xEE, 0xEF), cont, cont}));
      alts.push_back(byte_seq({range256(0xED, 0xED), range256(0x80, 0x9F), cont}));
      alts.push_back(byte_seq({range256(0xF0, 0xF0), range256(0x90, 0xBF), cont, cont}));
      alts.push_back(byte_seq({range256(0xF1, 0xF3), cont, cont, cont}));
      alts.push_back(byte_seq({range256(0xF4, 0xF4), range256(0x80, 0x8F), cont, cont}));
    } else for (uint32_t r : rs.extra) alts.push_back(literal_bytes(utf8(r)));
    if (alts.empty()) {   // empty set (e.g. [^\x00-\x{10FFFF}]): can never match
      std::bitset<256> none;
      return byte_seq({none});
    }
    return alternation(alts);
  }
  // simple case folding restricted to what can reach ASCII: A-Z <-> a-z, k/K <-> U+212A KELVIN SIGN, s/S <-> U+017F LONG S
  void fold_ascii(RuneSet& rs, bool negated) const {
    for (int c = 'a'; c <= 'z'; c++) {
      bool in = rs.ascii[c] || rs.ascii[c - 32];
      if (!in) continue;
      rs.ascii.set(c); rs.ascii.set(c - 32);
      if (c == 'k' || c == 's') {
        if (negated) throw RegexUnsupported("regex: negated class under (?i) folds onto a non-ASCII rune");
        if (!rs.rest) rs.extra.push_back(c == 'k' ? 0x212Au : 0x017Fu);
      }
    }
    for (uint32_t r : rs.extra) if (r != 0x212A && r != 0x017F) throw RegexUnsupported("regex: case folding of a non-ASCII rune");
    if (std::find(rs.extra.begin(), rs.extra.end(), 0x212Au) != rs.extra.end()) { rs.ascii.set('k'); rs.ascii.set('K'); }
    if (std::find(rs.extra.begin(), rs.extra.end(), 0x017Fu) != rs.extra.end()) { rs.ascii.set('s'); rs.ascii.set('S'); }
    std::sort(rs.extra.begin(), rs.extra.end());
    rs.extra.erase(std::unique(rs.extra.begin(), rs.extra.end()), rs.extra.end());
  }

  // ---- parser (regexp/syntax, Perl flags)
  Frag parse_alt() {
    std::vector<Frag> alts{parse_concat()};
    while (more() && peek() == '|') { i_++; alts.push_back(parse_concat()); }
    return alternation(alts);
  }

  Frag parse_concat() {
    Frag f;
    bool first = true;
    while (more() && peek() != '|' && peek() != ')') {
      Frag g = parse_repeat();
      if (first) { f = g; first = false; }
      else { patch(f.out, g.start); f.out = g.out; }
    }
    if (first) return empty_frag();
    return f;
  }

  Frag parse_repeat() {
    size_t atom_begin = i_;
    Flags atom_flags = fl_;
    bool is_group_flags = false;
    Frag f = parse_atom(&is_group_flags);
    if (is_group_flags) {   // (?i) and friends: not an operand
      // Go leaves whatever precedes the flag group on its parse stack, so a repetition here would bind to THAT (or fail
      // when there is nothing): an edge this engine does not reproduce
      if (at_repeat_op()) throw RegexUnsupported("regex: repetition operator after a flag group");
      return f;
    }
    if (!more()) return f;
    char c = peek();
    bool repeated = false;
    if (c == '*' || c == '+' || c == '?') {
      i_++;
      if (more() && peek() == '?') i_++;   // lazy marker: irrelevant for matching
      f = apply(f, c);
      repeated = true;
    } else if (c == '{') {
      size_t save = i_;
      int lo = 0, hi = -1;
      i_++;
      bool ok = parse_int(&lo);
      if (ok && more() && peek() == ',') { i_++; if (!parse_int(&hi)) hi = -2; } else hi = lo;
      if (!ok || !more() || peek() != '}') i_ = save;   // not a repetition: '{' is a literal
      else {
        i_++;
        if (more() && peek() == '?') i_++;
        if (lo > 1000 || hi > 1000 || (hi >= 0 && hi < lo)) throw RegexError("regex: invalid repeat count");
        if (lo > 100 || hi > 100) throw RegexUnsupported("regex: repeat count above 100");
        f = repeat_range(atom_begin, save, lo, hi, atom_flags);
        repeated = true;
      }
    }
    if (repeated && at_repeat_op()) throw RegexError("regex: invalid nested repetition operator");   // Perl mode: a** / a+* / a{2}* are errors
    return f;
  }

  // is the next token a repetition operator (* + ? or a well-formed {n} / {n,} / {n,m})?
  bool at_repeat_op() {
    if (!more()) return false;
    char d = peek();
    if (d == '*' || d == '+' || d == '?') return true;
    if (d != '{') return false;
    size_t save = i_;
    int x;
    i_++;
    bool ok = parse_int(&x);
    if (ok && more() && peek() == ',') { i_++; parse_int(&x); }
    bool op = ok && more() && peek() == '}';
    i_ = save;
    return op;
  }

  bool parse_int(int* v) {
    size_t s = i_;
    long x = 0;
    while (more() && isdigit((unsigned char)peek())) { x = std::min(x * 10 + (peek() - '0'), 100000L); i_++; }
    *v = (int)x;
    return i_ > s;
  }

  Frag apply(Frag f, char q) {
    Frag r;
    int sp = emit(SPLIT, 0, f.start, -1);
    if (q == '*') { patch(f.out, sp); r.start = sp; r.out = {hole(sp, 1)}; }
    else if (q == '+') { patch(f.out, sp); r.start = f.start; r.out = {hole(sp, 1)}; }
    else { r.start = sp; r.out = f.out; r.out.push_back(hole(sp, 1)); }
    return r;
  }

  // x{lo,hi}: re-parse the atom text lo..hi times (hi == -2: unbounded) under the flags the atom started with
  Frag repeat_range(size_t atom_begin, size_t atom_end, int lo, int hi, Flags atom_flags) {
    std::string atom = p_.substr(atom_begin, atom_end - atom_begin);
    std::string expanded;
    for (int k = 0; k < lo; k++) expanded += "(?:" + atom + ")";
    if (hi == -2) expanded += "(?:" + atom + ")*";
    else for (int k = lo; k < hi; k++) expanded += "(?:" + atom + ")?";
    // compile the expansion in place of the already-emitted atom copy (the earlier copy becomes dead code)
    std::string saved = p_;
    size_t saved_i = i_;
    Flags saved_fl = fl_;
    p_ = expanded; i_ = 0; fl_ = atom_flags;
    Frag f = parse_concat();
    p_ = saved; i_ = saved_i; fl_ = saved_fl;
    return f;
  }

  // (?flags) / (?flags:  -- returns true when the group is non-capturing-with-body (':' seen)
  bool parse_flags() {
    bool neg = false, any = false;
    for (;;) {
      if (!more()) throw RegexError("regex: missing ')'");
      char c = p_[i_++];
      switch (c) {
        case 'i': fl_.fold = !neg; any = true; break;
        case 'm': fl_.multiline = !neg; any = true; break;
        case 's': fl_.dotall = !neg; any = true; break;
        case 'U': any = true; break;   // swaps greedy / lazy: irrelevant for matching
        case '-': if (neg) throw RegexError("regex: invalid group flags"); neg = true; any = false; break;
        case ':': if (neg && !any) throw RegexError("regex: invalid group flags"); return true;
        case ')': if (neg && !any) throw RegexError("regex: invalid group flags"); return false;
        default: throw RegexError("regex: invalid or unsupported Perl syntax");
      }
    }
  }

  Frag literal_rune(uint32_t r) {
    if (r < 0x80) {
      if (fl_.fold && ((r >= 'a' && r <= 'z') || (r >= 'A' && r <= 'Z'))) { RuneSet rs; rs.ascii.set(r); fold_ascii(rs, false); return rune_set(rs); }
      return single(emit(CHAR, (unsigned char)r));
    }
    if (fl_.fold) {
      if (r == 0x212A || r == 0x017F) { RuneSet rs; rs.extra.push_back(r); fold_ascii(rs, false); return rune_set(rs); }
      throw RegexUnsupported("regex: non-ASCII literal under (?i)");
    }
    return literal_bytes(utf8(r));
  }

  // decodes the UTF-8 rune starting at p_[i_ - 1] (first byte already consumed as `c`)
  uint32_t rest_of_rune(unsigned char c) {
    int extra = c >= 0xF0 ? 3 : c >= 0xE0 ? 2 : c >= 0xC0 ? 1 : -1;
    if (extra < 0) throw RegexError("regex: invalid UTF-8");
    uint32_t r = c & (0x3F >> extra);
    for (int k = 0; k < extra; k++) {
      if (!more() || ((unsigned char)peek() & 0xC0) != 0x80) throw RegexError("regex: invalid UTF-8");
      r = (r << 6) | ((unsigned char)p_[i_++] & 0x3F);
    }
    return r;
  }

  Frag parse_atom(bool* is_group_flags) {
    unsigned char c = (unsigned char)p_[i_++];
    switch (c) {
      case '(': {
        Flags saved = fl_;
        if (more() && peek() == '?') {
          i_++;
          if (more() && peek() == 'P') {   // (?P<name>re)
            size_t e = p_.find('>', i_);
            if (i_ + 1 >= p_.size() || p_[i_ + 1] != '<' || e == std::string::npos || e == i_ + 2) throw RegexError("regex: invalid named capture");
            i_ = e + 1;
          } else if (more() && peek() == '<') {   // (?<name>re)
            size_t e = p_.find('>', i_);
            if (e == std::string::npos || e == i_ + 1) throw RegexError("regex: invalid named capture");
            i_ = e + 1;
          } else if (!parse_flags()) { *is_group_flags = true; return empty_frag(); }   // (?i): flags stay in force until the enclosing group ends
        }
        Frag f = parse_alt();
        if (!more() || peek() != ')') throw RegexError("regex: missing ')'");
        i_++;
        fl_ = saved;
        return f;
      }
      case '.': {
        RuneSet rs;
        rs.ascii.set(); rs.rest = true;
        if (!fl_.dotall) rs.ascii.reset('\n');
        return rune_set(rs);
      }
      case '^': return single(emit(fl_.multiline ? BOL_ML : BOT));
      case '$': return single(emit(fl_.multiline ? EOL_ML : EOT));
      case '[': return parse_class();
      case '\\': return parse_escape();
      case '*': case '+': case '?': throw RegexError("regex: missing argument to repetition operator");
      default:
        if (c < 0x80) return literal_rune(c);
        return literal_rune(rest_of_rune(c));
    }
  }

  static void perl_class(char e, RuneSet& rs) {   // \d \w \s and their negations (ASCII definitions, as in RE2)
    RuneSet s;
    auto range = [&](int a, int b) { for (int k = a; k <= b; k++) s.ascii.set(k); };
    switch (e) {
      case 'd': case 'D': range('0', '9'); break;
      case 'w': case 'W': range('0', '9'); range('a', 'z'); range('A', 'Z'); s.ascii.set('_'); break;
      default: for (char c : std::string("\t\n\f\r ")) s.ascii.set((unsigned char)c); break;   // \s == [\t\n\f\r ]
    }
    if (e == 'D' || e == 'W' || e == 'S') { s.ascii.flip(); s.rest = true; }
    rs.ascii |= s.ascii;
    rs.rest = rs.rest || s.rest;
  }
  static bool is_perl_class(char e) { return e == 'd' || e == 'D' || e == 'w' || e == 'W' || e == 's' || e == 'S'; }

  // escapes that denote ONE rune (shared by atoms and classes); i_ is just past the backslash's letter `e`
  uint32_t escape_rune(char e) {
    switch (e) {
      case 'a': return 7;
      case 'f': return '\f';
      case 'n': return '\n';
      case 'r': return '\r';
      case 't': return '\t';
      case 'v': return '\v';
      case 'x': {
        if (!more()) throw RegexError("regex: invalid escape sequence");
        auto hex = [](char h) { return h >= '0' && h <= '9' ? h - '0' : h >= 'a' && h <= 'f' ? h - 'a' + 10 : h >= 'A' && h <= 'F' ? h - 'A' + 10 : -1; };
        uint32_t r = 0;
        if (peek() == '{') {
          i_++;
          int nd = 0;
          while (more() && peek() != '}') { int h = hex(p_[i_++]); if (h < 0) throw RegexError("regex: invalid escape sequence"); r = r * 16 + (uint32_t)h; if (r > 0x10FFFF) throw RegexError("regex: invalid escape sequence"); nd++; }
          if (!more() || nd == 0) throw RegexError("regex: invalid escape sequence");
          i_++;
          return r;
        }
        for (int k = 0; k < 2; k++) { if (!more()) throw RegexError("regex: invalid escape sequence"); int h = hex(p_[i_++]); if (h < 0) throw RegexError("regex: invalid escape sequence"); r = r * 16 + (uint32_t)h; }
        return r;
      }
      case '1': case '2': case '3': case '4': case '5': case '6': case '7':
        if (!more() || peek() < '0' || peek() > '7') throw RegexError("regex: invalid escape sequence");   // a lone digit is a backreference
        [[fallthrough]];   // octal
      case '0': {
        uint32_t r = (uint32_t)(e - '0');
        for (int k = 0; k < 2 && more() && peek() >= '0' && peek() <= '7'; k++) r = r * 8 + (uint32_t)(p_[i_++] - '0');
        return r;
      }
      default:
        if ((unsigned char)e < 0x80 && !isalnum((unsigned char)e)) return (unsigned char)e;   // escaped punctuation is itself
        throw RegexError("regex: invalid escape sequence");
    }
  }

  Frag parse_escape() {
    if (!more()) throw RegexError("regex: trailing backslash");
    char e = p_[i_++];
    if (is_perl_class(e)) { RuneSet rs; perl_class(e, rs); if (fl_.fold) fold_ascii(rs, e == 'D' || e == 'W' || e == 'S'); return rune_set(rs); }
    switch (e) {
      case 'A': return single(emit(BOT));
      case 'z': return single(emit(EOT));
      case 'b': return single(emit(WORDB));
      case 'B': return single(emit(NWORDB));
      case 'p': case 'P': throw RegexUnsupported("regex: Unicode character classes (\\p) are not implemented");
      case 'Q': {   // literal text up to \E (or the end of the pattern)
        size_t e2 = p_.find("\\E", i_);
        std::string lit = p_.substr(i_, e2 == std::string::npos ? std::string::npos : e2 - i_);
        i_ = e2 == std::string::npos ? p_.size() : e2 + 2;
        if (lit.empty()) return empty_frag();
        // every rune of the quoted text is an atom of its own: only the last one takes a following repetition operator
        std::string saved = p_; size_t saved_i = i_;
        auto quote = [](const std::string& t) {
          std::string o;
          for (unsigned char ch : t) { if (ch < 0x80 && !isalnum(ch)) o.push_back('\\'); o.push_back((char)ch); }
          return o;
        };
        size_t k = lit.size();
        do { k--; } while (k > 0 && ((unsigned char)lit[k] & 0xC0) == 0x80);   // first byte of the last rune
        p_ = quote(lit.substr(0, k)) + "(?:" + quote(lit.substr(k)) + ")"; i_ = 0;
        Frag f = parse_concat();
        p_ = saved; i_ = saved_i;
        return f;
      }
      default: return literal_rune(escape_rune(e));
    }
  }

  Frag parse_class() {
    RuneSet rs;
    bool neg = false;
    if (more() && peek() == '^') { neg = true; i_++; }
    bool first = true;
    auto range = [&](int a, int b) { for (int k = a; k <= b; k++) rs.ascii.set(k); };
    for (;;) {
      if (!more()) throw RegexError("regex: missing ']'");
      unsigned char c = (unsigned char)p_[i_++];
      if (c == ']' && !first) break;
      first = false;
      if (c == '[' && more() && peek() == ':') {
        size_t e = p_.find(":]", i_);
        if (e == std::string::npos) throw RegexError("regex: bad POSIX class");
        std::string name = p_.substr(i_ + 1, e - i_ - 1);
        i_ = e + 2;
        bool cneg = !name.empty() && name[0] == '^';
        if (cneg) name = name.substr(1);
        RuneSet saved = rs;
        rs = RuneSet();
        if (name == "alpha") { range('a', 'z'); range('A', 'Z'); }
        else if (name == "digit") range('0', '9');
        else if (name == "alnum") { range('a', 'z'); range('A', 'Z'); range('0', '9'); }
        else if (name == "upper") range('A', 'Z');
        else if (name == "lower") range('a', 'z');
        else if (name == "space") { for (char ch : std::string(" \t\n\r\f\v")) rs.ascii.set((unsigned char)ch); }
        else if (name == "blank") { rs.ascii.set(' '); rs.ascii.set('\t'); }
        else if (name == "cntrl") { range(0, 31); rs.ascii.set(127); }
        else if (name == "graph") range('!', '~');
        else if (name == "print") range(' ', '~');
        else if (name == "ascii") range(0, 127);
        else if (name == "xdigit") { range('0', '9'); range('a', 'f'); range('A', 'F'); }
        else if (name == "punct") { range('!', '/'); range(':', '@'); range('[', '`'); range('{', '~'); }
        else if (name == "word") { range('a', 'z'); range('A', 'Z'); range('0', '9'); rs.ascii.set('_'); }
        else throw RegexError("regex: invalid character class range " + name);
        if (cneg) { rs.ascii.flip(); rs.rest = true; }
        rs.ascii |= saved.ascii; rs.rest = rs.rest || saved.rest;
        rs.extra.insert(rs.extra.end(), saved.extra.begin(), saved.extra.end());
        continue;
      }
      uint32_t lo;
      if (c == '\\') {
        if (!more()) throw RegexError("regex: trailing backslash");
        char e = p_[i_++];
        if (is_perl_class(e)) { perl_class(e, rs); continue; }
        if (e == 'p' || e == 'P') throw RegexUnsupported("regex: Unicode character classes (\\p) are not implemented");
        lo = escape_rune(e);
      } else lo = c < 0x80 ? c : rest_of_rune(c);
      if (more() && peek() == '-' && i_ + 1 < p_.size() && p_[i_ + 1] != ']') {
        i_++;
        unsigned char h = (unsigned char)p_[i_++];
        uint32_t hi;
        if (h == '\\') {
          if (!more()) throw RegexError("regex: trailing backslash");
          char e = p_[i_++];
          if (is_perl_class(e) || e == 'p' || e == 'P') throw RegexError("regex: invalid character class range");
          hi = escape_rune(e);
        } else hi = h < 0x80 ? h : rest_of_rune(h);
        if (hi < lo) throw RegexError("regex: invalid character class range");
        if (hi >= 0x80) {
          if (lo == 0x80 && hi == 0x10FFFF) { rs.rest = true; continue; }
          if (lo < 0x80 && hi == 0x10FFFF) { range((int)lo, 127); rs.rest = true; continue; }
          throw RegexUnsupported("regex: non-ASCII rune in a character class range");
        }
        range((int)lo, (int)hi);
      } else if (lo < 0x80) rs.ascii.set(lo);
      else rs.extra.push_back(lo);
    }
    if (fl_.fold) fold_ascii(rs, neg);
    if (neg) {
      if (!rs.extra.empty() && !rs.rest) throw RegexUnsupported("regex: negated class with non-ASCII members");
      RuneSet out;
      out.ascii = ~rs.ascii;
      out.rest = !rs.rest;
      return rune_set(out);
    }
    if (rs.rest) rs.extra.clear();
    return rune_set(rs);
  }
};

}  // namespace gk
