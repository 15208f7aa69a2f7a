// Small RE2-style regular-expression engine (Pike VM, linear time, no backtracking) for re_match / regex.match.
// The reference evaluates these through Go's regexp package inside OPA (third-party); this covers the syntax its
// in-tree templates use: literals, '.', classes (ranges, negation, \d \w \s, [[:alpha:]]), ^ $ anchors, groups,
// alternation, * + ? {m,n} (greedy/lazy are equivalent for matching).  Byte-oriented.
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
      gen++;
      cur.clear();
      for (int pc : carried) add(cur, pc, pos, n, mark, gen);
      add(cur, start_, pos, n, mark, gen);   // a new thread may start at every position
      gen++;
      nxt.clear();
      for (int pc : cur) {
        const Inst& in = prog_[pc];
        if (in.op == MATCH) return true;
        if (pos < n) {
          unsigned char c = (unsigned char)s[pos];
          bool ok = (in.op == CHAR && c == in.c) || (in.op == ANY && c != '\n') || (in.op == CLASS && classes_[in.x][c]);
          if (ok) add(nxt, pc + 1, pos + 1, n, mark, gen);
        }
      }
      if (pos >= n) return false;
      carried.swap(nxt);
    }
  }

  // The same search as a byte-class DFA for the device (P_REGEX, vm_core.hpp):
  //   table = [u32 n_states][u32 n_classes][u8 class_of_byte[256]][u8 accept_at_end[n_states]][u8 next[n_states][n_classes]]
  // state 0 = start of input; an early match moves to an absorbing accepting state.  A state is the set of threads
  // carried over from the previous byte plus "still at position 0" (for ^); `accept_at_end` applies $ at the end.
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
    for (const Inst& in : prog_) {
      std::bitset<256> bs;
      if (in.op == CHAR) bs.set(in.c);
      else if (in.op == ANY) { bs.set(); bs.reset('\n'); }
      else if (in.op == CLASS) bs = classes_[in.x];
      else continue;
      refine(bs);
    }
    if (ncls > 255) return false;
    std::vector<int> rep(ncls, -1);
    for (int b = 0; b < 256; b++) if (rep[cls[b]] < 0) rep[cls[b]] = b;
    typedef std::pair<std::vector<int>, bool> Key;   // carried threads (sorted), at position 0
    std::map<Key, int> ids;
    std::vector<Key> states;
    auto intern = [&](const Key& k) { auto it = ids.find(k); if (it != ids.end()) return it->second; ids[k] = (int)states.size(); states.push_back(k); return (int)states.size() - 1; };
    auto closure = [&](const Key& k, bool at_end, std::vector<int>* cur) {
      std::vector<uint32_t> mark(prog_.size(), 0);
      cur->clear();
      size_t pos = k.second ? 0 : 1, n = at_end ? pos : pos + 1;
      for (int pc : k.first) add(*cur, pc, pos, n, mark, 1);
      add(*cur, start_, pos, n, mark, 1);
    };
    auto has_match = [&](const std::vector<int>& cur) { for (int pc : cur) if (prog_[pc].op == MATCH) return true; return false; };
    intern(Key{{}, true});
    const int ACCEPT = -2;
    std::vector<std::vector<int>> next;
    std::vector<uint8_t> accept;
    std::vector<int> cur;
    for (size_t s = 0; s < states.size(); s++) {
      if (states.size() > max_states) return false;
      Key k = states[s];
      closure(k, true, &cur);
      accept.push_back(has_match(cur) ? 1 : 0);
      closure(k, false, &cur);
      std::vector<int> row(ncls, 0);
      if (has_match(cur)) { std::fill(row.begin(), row.end(), ACCEPT); next.push_back(row); continue; }
      for (int c = 0; c < ncls; c++) {
        unsigned char b = (unsigned char)rep[c];
        std::vector<int> nxt;
        for (int pc : cur) {
          const Inst& in = prog_[pc];
          bool ok = (in.op == CHAR && b == in.c) || (in.op == ANY && b != '\n') || (in.op == CLASS && classes_[in.x][b]);
          if (ok) nxt.push_back(pc + 1);
        }
        std::sort(nxt.begin(), nxt.end());
        nxt.erase(std::unique(nxt.begin(), nxt.end()), nxt.end());
        row[c] = intern(Key{nxt, false});
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
  enum Op { CHAR, ANY, CLASS, SPLIT, JMP, MATCH, BOL, EOL };
  struct Inst { Op op; unsigned char c; int x, y; };
  struct Frag { int start; std::vector<int*> out; bool empty = false; };
  std::string p_;
  size_t i_ = 0;
  std::vector<Inst> prog_;
  std::vector<std::bitset<256>> classes_;
  int start_ = 0;
  std::vector<std::unique_ptr<int>> holes_;

  int emit(Op op, unsigned char c = 0, int x = -1, int y = -1) { prog_.push_back({op, c, x, y}); return (int)prog_.size() - 1; }

  void add(std::vector<int>& list, int pc, size_t pos, size_t n, std::vector<uint32_t>& mark, uint32_t gen) const {
    if (mark[pc] == gen) return;
    mark[pc] = gen;
    const Inst& in = prog_[pc];
    switch (in.op) {
      case JMP: add(list, in.x, pos, n, mark, gen); break;
      case SPLIT: add(list, in.x, pos, n, mark, gen); add(list, in.y, pos, n, mark, gen); break;
      case BOL: if (pos == 0) add(list, pc + 1, pos, n, mark, gen); break;
      case EOL: if (pos == n) add(list, pc + 1, pos, n, mark, gen); break;
      default: list.push_back(pc);
    }
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

  Frag parse_alt() {
    Frag f = parse_concat();
    if (!(more() && peek() == '|')) return f;
    // alternation needs a SPLIT in front: rebuild as  SPLIT L1, L2 ; since code is already emitted for f, we wrap
    // by emitting a JMP trampoline: prog layout  [f ...] -> we cannot insert in front, so compile alternatives into
    // separate regions and dispatch with a SPLIT emitted afterwards.
    std::vector<Frag> alts{f};
    while (more() && peek() == '|') { i_++; alts.push_back(parse_concat()); }
    Frag out;
    int prev = -1;
    // chain of SPLITs: split0 -> alt0 | split1 -> alt1 | ... -> altN
    for (size_t k = 0; k + 1 < alts.size(); k++) {
      int sp = emit(SPLIT, 0, alts[k].start, -1);
      if (prev >= 0) prog_[prev].y = sp; else out.start = sp;
      prev = sp;
    }
    prog_[prev].y = alts.back().start;
    for (auto& a : alts) out.out.insert(out.out.end(), a.out.begin(), a.out.end());
    return out;
  }

  Frag parse_concat() {
    Frag f;
    bool first = true;
    while (more() && peek() != '|' && peek() != ')') {
      Frag g = parse_repeat();
      if (first) { f = g; first = false; }
      else { patch(f.out, g.start); f.out = g.out; }
    }
    if (first) {   // empty: a JMP to whatever follows
      int j = emit(JMP);
      f.start = j;
      f.out = {hole(j, 0)};
    }
    return f;
  }

  Frag parse_repeat() {
    size_t atom_begin = i_;
    Frag f = parse_atom();
    while (more()) {
      char c = peek();
      if (c == '*' || c == '+' || c == '?') {
        i_++;
        if (more() && peek() == '?') i_++;   // lazy marker: irrelevant for matching
        f = apply(f, c, atom_begin);
      } else if (c == '{') {
        size_t save = i_;
        int lo = 0, hi = -1;
        i_++;
        bool ok = parse_int(&lo);
        if (ok && more() && peek() == ',') { i_++; if (!parse_int(&hi)) hi = -2; } else hi = lo;
        if (!ok || !more() || peek() != '}') { i_ = save; break; }
        i_++;
        if (more() && peek() == '?') i_++;
        f = repeat_range(atom_begin, save, lo, hi);
      } else break;
    }
    return f;
  }

  bool parse_int(int* v) {
    size_t s = i_;
    int x = 0;
    while (more() && isdigit((unsigned char)peek())) { x = x * 10 + (peek() - '0'); i_++; }
    *v = x;
    return i_ > s;
  }

  Frag apply(Frag f, char q, size_t) {
    Frag r;
    if (q == '*') {
      int sp = emit(SPLIT, 0, f.start, -1);
      patch(f.out, sp);
      r.start = sp;
      r.out = {hole(sp, 1)};
    } else if (q == '+') {
      int sp = emit(SPLIT, 0, f.start, -1);
      patch(f.out, sp);
      r.start = f.start;
      r.out = {hole(sp, 1)};
    } else {
      int sp = emit(SPLIT, 0, f.start, -1);
      r.start = sp;
      r.out = f.out;
      r.out.push_back(hole(sp, 1));
    }
    return r;
  }

  // x{lo,hi}: re-parse the atom text lo..hi times (hi == -2: unbounded)
  Frag repeat_range(size_t atom_begin, size_t atom_end, int lo, int hi) {
    std::string atom = p_.substr(atom_begin, atom_end - atom_begin);
    if (lo > 100 || hi > 100) throw RegexError("regex: repeat count too large");
    std::string expanded;
    for (int k = 0; k < lo; k++) expanded += "(?:" + atom + ")";
    if (hi == -2) expanded += "(?:" + atom + ")*";
    else for (int k = lo; k < hi; k++) expanded += "(?:" + atom + ")?";
    // compile the expansion in place of the already-emitted atom copy (the earlier copy becomes dead code)
    std::string saved = p_;
    size_t saved_i = i_;
    p_ = expanded;
    i_ = 0;
    Frag f = parse_concat();
    p_ = saved;
    i_ = saved_i;
    return f;
  }

  Frag single(int inst) {
    Frag f;
    f.start = inst;
    int j = emit(JMP);
    f.out = {hole(j, 0)};
    return f;
  }

  Frag parse_atom() {
    char c = p_[i_++];
    switch (c) {
      case '(': {
        if (i_ + 1 < p_.size() && p_[i_] == '?') {
          if (p_[i_ + 1] == ':') i_ += 2;
          else if (p_[i_ + 1] == 'P' || p_[i_ + 1] == '<') { size_t e = p_.find('>', i_); if (e == std::string::npos) throw RegexError("regex: bad group name"); i_ = e + 1; }
          else throw RegexError("regex: unsupported group flags");
        }
        Frag f = parse_alt();
        if (!more() || peek() != ')') throw RegexError("regex: missing ')'");
        i_++;
        return f;
      }
      case '.': return single(emit(ANY));
      case '^': return single(emit(BOL));
      case '$': return single(emit(EOL));
      case '[': return single(emit(CLASS, 0, parse_class()));
      case '\\': {
        if (!more()) throw RegexError("regex: trailing backslash");
        char e = p_[i_++];
        std::bitset<256> bs;
        if (escape_class(e, bs)) { classes_.push_back(bs); return single(emit(CLASS, 0, (int)classes_.size() - 1)); }
        if (e == 'A') return single(emit(BOL));
        if (e == 'z') return single(emit(EOL));
        return single(emit(CHAR, (unsigned char)escape_char(e)));
      }
      case '*': case '+': case '?': throw RegexError("regex: missing argument to repetition operator");
      default: return single(emit(CHAR, (unsigned char)c));
    }
  }

  static char escape_char(char e) {
    switch (e) {
      case 'n': return '\n';
      case 't': return '\t';
      case 'r': return '\r';
      case 'f': return '\f';
      case 'v': return '\v';
      default: return e;
    }
  }
  static bool escape_class(char e, std::bitset<256>& bs) {
    auto range = [&](int a, int b) { for (int k = a; k <= b; k++) bs.set(k); };
    switch (e) {
      case 'd': range('0', '9'); return true;
      case 'D': range('0', '9'); bs.flip(); return true;
      case 'w': range('0', '9'); range('a', 'z'); range('A', 'Z'); bs.set('_'); return true;
      case 'W': range('0', '9'); range('a', 'z'); range('A', 'Z'); bs.set('_'); bs.flip(); return true;
      case 's': for (char c : std::string(" \t\n\r\f\v")) bs.set((unsigned char)c); return true;
      case 'S': for (char c : std::string(" \t\n\r\f\v")) bs.set((unsigned char)c); bs.flip(); return true;
      default: return false;
    }
  }

  int parse_class() {
    std::bitset<256> bs;
    bool neg = false;
    if (more() && peek() == '^') { neg = true; i_++; }
    bool first = true;
    for (;;) {
      if (!more()) throw RegexError("regex: missing ']'");
      char c = p_[i_++];
      if (c == ']' && !first) break;
      first = false;
      if (c == '[' && more() && peek() == ':') {
        size_t e = p_.find(":]", i_);
        if (e == std::string::npos) throw RegexError("regex: bad POSIX class");
        std::string name = p_.substr(i_ + 1, e - i_ - 1);
        i_ = e + 2;
        auto range = [&](int a, int b) { for (int k = a; k <= b; k++) bs.set(k); };
        if (name == "alpha") { range('a', 'z'); range('A', 'Z'); }
        else if (name == "digit") range('0', '9');
        else if (name == "alnum") { range('a', 'z'); range('A', 'Z'); range('0', '9'); }
        else if (name == "upper") range('A', 'Z');
        else if (name == "lower") range('a', 'z');
        else if (name == "space") { for (char ch : std::string(" \t\n\r\f\v")) bs.set((unsigned char)ch); }
        else if (name == "xdigit") { range('0', '9'); range('a', 'f'); range('A', 'F'); }
        else if (name == "punct") { range('!', '/'); range(':', '@'); range('[', '`'); range('{', '~'); }
        else if (name == "word") { range('a', 'z'); range('A', 'Z'); range('0', '9'); bs.set('_'); }
        else throw RegexError("regex: unknown POSIX class " + name);
        continue;
      }
      unsigned char lo;
      if (c == '\\') {
        if (!more()) throw RegexError("regex: trailing backslash");
        char e = p_[i_++];
        std::bitset<256> sub;
        if (escape_class(e, sub)) { bs |= sub; continue; }
        lo = (unsigned char)escape_char(e);
      } else lo = (unsigned char)c;
      if (more() && peek() == '-' && i_ + 1 < p_.size() && p_[i_ + 1] != ']') {
        i_++;
        char h = p_[i_++];
        if (h == '\\') { if (!more()) throw RegexError("regex: trailing backslash"); h = escape_char(p_[i_++]); }
        unsigned char hi = (unsigned char)h;
        if (hi < lo) throw RegexError("regex: invalid character class range");
        for (int k = lo; k <= hi; k++) bs.set(k);
      } else bs.set(lo);
    }
    if (neg) bs.flip();
    classes_.push_back(bs);
    return (int)classes_.size() - 1;
  }
};

}  // namespace gk
