// Rego builtins over concrete Values (constant folding in the AOT compiler; message rendering on the host).
// Restates the OPA v1.17.1 builtins the reference's in-tree templates call.  A builtin
// returning Undefined makes the calling expression undefined (OPA's default non-strict error mode).
#include "builtins.hpp"

#include <cmath>
#include <map>
#include <mutex>
#include <set>
#include <unordered_map>

#include "regex.hpp"

namespace gk {
namespace {

const Value U;   // undefined

bool is_num(const Value& v) { return v.is_number(); }

// ---------------------------------------------------------------------------------------------- sprintf
struct GoArg { int kind; i128 i; double f; std::string s; };   // 0 int, 1 float64, 2 string

GoArg go_arg(const Value& v) {
  GoArg a{2, 0, 0, ""};
  if (v.is_number()) {
    if (v.is_int) { a.kind = 0; a.i = v.i; }
    else { a.kind = 1; a.f = v.d; }
  } else if (v.is_string()) a.s = v.str();
  else a.s = to_term_string(v);
  return a;
}

std::string bad_verb(char verb, const GoArg& a) {
  std::string o = "%!";
  o.push_back(verb);
  if (a.kind == 2) return o + "(string=" + a.s + ")";
  if (a.kind == 0) return o + "(int=" + i128_to_string(a.i) + ")";
  return o + "(float64=" + go_float_v(a.f) + ")";
}

// fmt counts width and precision in runes (fmt/print.go padString: utf8.RuneCountInString; fmtS truncates at rune boundaries)
size_t fmt_rune_count(const std::string& s) {
  size_t n = 0;
  for (unsigned char c : s) n += (c & 0xC0) != 0x80;
  return n;
}
std::string rune_prefix(const std::string& s, size_t runes) {
  size_t b = 0;
  for (; b < s.size() && runes; runes--) { b++; while (b < s.size() && ((unsigned char)s[b] & 0xC0) == 0x80) b++; }
  return s.substr(0, b);
}

// fmt %c (fmt/format.go fmtC): the rune's UTF-8; anything that is no valid code point prints as U+FFFD
void append_rune(std::string& o, i128 r) {
  if (r < 0 || r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF)) r = 0xFFFD;
  const uint32_t c = (uint32_t)r;
  if (c < 0x80) o.push_back((char)c);
  else if (c < 0x800) { o.push_back((char)(0xC0 | (c >> 6))); o.push_back((char)(0x80 | (c & 0x3F))); }
  else if (c < 0x10000) { o.push_back((char)(0xE0 | (c >> 12))); o.push_back((char)(0x80 | ((c >> 6) & 0x3F))); o.push_back((char)(0x80 | (c & 0x3F))); }
  else { o.push_back((char)(0xF0 | (c >> 18))); o.push_back((char)(0x80 | ((c >> 12) & 0x3F))); o.push_back((char)(0x80 | ((c >> 6) & 0x3F))); o.push_back((char)(0x80 | (c & 0x3F))); }
}

std::string pad(std::string s, const std::string& flags, int width) {
  const size_t len = fmt_rune_count(s);
  if (width < 0 || len >= (size_t)width) return s;
  const size_t fill = (size_t)width - len;
  if (flags.find('-') != std::string::npos) return s + std::string(fill, ' ');
  if (flags.find('0') != std::string::npos && !s.empty() && (isdigit((unsigned char)s[0]) || s[0] == '+' || s[0] == '-')) {
    std::string sign;
    if (s[0] == '+' || s[0] == '-') { sign = s.substr(0, 1); s = s.substr(1); }
    return sign + std::string(fill, '0') + s;
  }
  return std::string(fill, ' ') + s;
}

std::string to_base(i128 v, int base, bool upper) {
  if (v == 0) return "0";
  bool neg = v < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
  const char* digs = upper ? "0123456789ABCDEF" : "0123456789abcdef";
  std::string s;
  while (u) { s.push_back(digs[(int)(u % base)]); u /= base; }
  if (neg) s.push_back('-');
  std::reverse(s.begin(), s.end());
  return s;
}

}  // namespace

std::string go_sprintf(const std::string& fmt, const ValueVec& argv) {
  std::vector<GoArg> args;
  for (const Value& v : argv) args.push_back(go_arg(v));
  std::string out;
  size_t ai = 0, n = fmt.size();
  for (size_t i = 0; i < n;) {
    if (fmt[i] != '%') { out.push_back(fmt[i++]); continue; }
    size_t j = i + 1;
    std::string flags;
    while (j < n && strchr("-+# 0", fmt[j])) flags.push_back(fmt[j++]);
    int width = -1, prec = -1;
    // fmt/print.go parsenum: a number that is already beyond 1e6 when a further digit arrives is "crazy long": the directive -- and with
    // it the rest of the format -- is given up as %!(NOVERB) (no width or precision of billions ever reaches an allocation)
    bool too_large = false;
    auto number = [&](int* v) { *v = 0; while (j < n && isdigit((unsigned char)fmt[j])) { if (*v > 1000000) { too_large = true; return; } *v = *v * 10 + (fmt[j++] - '0'); } };
    if (j < n && isdigit((unsigned char)fmt[j])) number(&width);
    if (!too_large && j < n && fmt[j] == '.') { j++; number(&prec); }
    if (too_large || j >= n) { out += "%!(NOVERB)"; break; }
    char verb = fmt[j++];
    std::string wide;   // a verb beyond ASCII is one whole rune (doPrintf decodes it); never a valid verb
    if ((unsigned char)verb >= 0x80) { wide.push_back(verb); while (j < n && ((unsigned char)fmt[j] & 0xC0) == 0x80) wide.push_back(fmt[j++]); }
    i = j;
    if (verb == '%') { out.push_back('%'); continue; }
    if (ai >= args.size()) { out += "%!"; if (wide.empty()) out.push_back(verb); else out += wide; out += "(MISSING)"; continue; }
    const GoArg& a = args[ai++];
    std::string s;
    const bool plus = flags.find('+') != std::string::npos, space = flags.find(' ') != std::string::npos;
    const bool sharp = flags.find('#') != std::string::npos, minus = flags.find('-') != std::string::npos;
    const bool zero = flags.find('0') != std::string::npos && !minus;
    bool bad = false;
    // fmt/format.go fmtInteger: precision = minimum digits (%.0d of 0 prints nothing), the 0 flag = precision from the width, '#' the
    // base prefix, then the sign; whatever is left of the width is spaces
    auto integer = [&](int base, bool upper, bool o_prefix) {
      const bool neg = a.i < 0;
      std::string d = to_base(neg ? -a.i : a.i, base, upper);
      int p = 0;
      if (prec >= 0) { p = prec; if (prec == 0 && a.i == 0) d.clear(); }
      else if (zero && width >= 0) { p = width; if (neg || plus || space) p--; }
      if ((int)d.size() < p) d.insert(0, (size_t)p - d.size(), '0');
      if (sharp) {
        if (base == 2) d.insert(0, "0b");
        else if (base == 8) { if (d.empty() || d[0] != '0') d.insert(0, "0"); }
        else if (base == 16) d.insert(0, upper ? "0X" : "0x");
      }
      if (o_prefix) d.insert(0, "0o");
      if (neg) d.insert(0, "-"); else if (plus) d.insert(0, "+"); else if (space) d.insert(0, " ");
      const size_t len = d.size();
      if (width >= 0 && len < (size_t)width) d = minus ? d + std::string((size_t)width - len, ' ') : std::string((size_t)width - len, ' ') + d;
      return d;
    };
    // fmtFloat: the sign is '-', '+' with the plus flag, ' ' with the space flag; the 0 flag pads between sign and digits
    auto floating = [&](std::string num) {
      if (num[0] != '-') { if (plus) num.insert(0, "+"); else if (space) num.insert(0, " "); }
      if (width < 0 || num.size() >= (size_t)width) return num;
      const size_t fill = (size_t)width - num.size();
      if (minus) return num + std::string(fill, ' ');
      if (!zero) return std::string(fill, ' ') + num;
      const bool sign = num[0] == '-' || num[0] == '+' || num[0] == ' ';
      return (sign ? num.substr(0, 1) : std::string()) + std::string(fill, '0') + num.substr(sign ? 1 : 0);
    };
    auto c_float = [&](char conv, int p) {
      char f[24];
      snprintf(f, sizeof f, "%%.%d%c", p, conv);
      const int need = snprintf(nullptr, 0, f, a.f);   // %f of 1.8e308 is 300+ digits
      std::string t((size_t)need + 1, '\0');
      snprintf(&t[0], t.size(), f, a.f);
      t.resize((size_t)need);
      return t;
    };
    // the operand under %v with this directive's flags, width and precision (printArg(arg, 'v')): a string is cut to the precision
    // (fmtS), an integer takes it as minimum digits, a float64 prints %g -- the shortest text, or that many significant digits
    auto as_v = [&]() {
      if (a.kind == 2) return pad(prec >= 0 ? rune_prefix(a.s, (size_t)prec) : a.s, flags, width);
      if (a.kind == 0) return integer(10, false, false);
      return floating(prec >= 0 ? c_float('g', prec) : go_float_v(a.f));
    };
    switch (wide.empty() ? verb : '\0') {
      case 'v': s = as_v(); break;
      case 's': if (a.kind == 2) s = as_v(); else bad = true; break;
      case 'q': if (a.kind == 2) s = pad(go_quote(prec >= 0 ? rune_prefix(a.s, (size_t)prec) : a.s), flags, width); else bad = true; break;
      case 'd': if (a.kind == 0) s = integer(10, false, false); else bad = true; break;
      case 'x': case 'X':
        if (a.kind == 0) s = integer(16, verb == 'X', false);
        else if (a.kind == 2) {   // fmtSbx: the precision counts bytes, ' ' separates them, '#' prefixes (each byte when separated)
          const char* d = verb == 'X' ? "0123456789ABCDEF" : "0123456789abcdef";
          const size_t nb = prec >= 0 && (size_t)prec < a.s.size() ? (size_t)prec : a.s.size();
          for (size_t k = 0; k < nb; k++) {
            const unsigned char c = (unsigned char)a.s[k];
            if (space && k) s.push_back(' ');
            if (sharp && (space || k == 0)) s += verb == 'X' ? "0X" : "0x";
            s.push_back(d[c >> 4]); s.push_back(d[c & 15]);
          }
          s = pad(s, flags, width);
        }
        else bad = true;
        break;
      case 'o': if (a.kind == 0) s = integer(8, false, false); else bad = true; break;
      case 'O': if (a.kind == 0) s = integer(8, false, true); else bad = true; break;
      case 'b': if (a.kind == 0) s = integer(2, false, false); else bad = true; break;
      case 'c': if (a.kind == 0) { append_rune(s, a.i); s = pad(s, flags, width); } else bad = true; break;
      case 'U':
        if (a.kind == 0 && a.i >= 0) { s = to_base(a.i, 16, true); if (s.size() < 4) s.insert(0, 4 - s.size(), '0'); s = pad("U+" + s, flags, width); }
        else bad = true;
        break;
      case 'T': {   // the operand's Go type (printArg: fmtS(reflect.TypeOf(arg).String()))
        const std::string t = a.kind == 0 ? "int" : a.kind == 1 ? "float64" : "string";
        s = pad(prec >= 0 ? rune_prefix(t, (size_t)prec) : t, flags, width);
        break;
      }
      case 'f': case 'F': case 'e': case 'E': case 'g': case 'G':
        if (a.kind == 1) {
          if ((verb == 'g' || verb == 'G') && prec < 0) {   // %g without a precision is the shortest text that round-trips, as %v
            s = go_float_v(a.f);
            if (verb == 'G') for (char& c : s) if (c == 'e') c = 'E';
          } else s = c_float(verb == 'F' ? 'f' : verb, prec < 0 ? 6 : prec);
          s = floating(s);
        } else bad = true;
        break;
      default: bad = true;
    }
    if (bad) {   // badVerb (fmt/print.go): %!verb(type=value) with the value printed as %v UNDER THIS DIRECTIVE'S flags, width and precision
      out += "%!";
      if (wide.empty()) out.push_back(verb); else out += wide;
      out += a.kind == 0 ? "(int=" : a.kind == 1 ? "(float64=" : "(string=";
      out += as_v() + ")";
    } else out += s;
  }
  if (ai < args.size()) {
    out += "%!(EXTRA ";
    for (size_t k = ai; k < args.size(); k++) {
      if (k > ai) out += ", ";
      const GoArg& a = args[k];
      if (a.kind == 0) out += "int=" + i128_to_string(a.i);
      else if (a.kind == 1) out += "float64=" + go_float_v(a.f);
      else out += "string=" + a.s;
    }
    out += ")";
  }
  return out;
}

namespace {
std::mutex g_re_mu;
std::unordered_map<std::string, std::shared_ptr<Regex>> g_re_cache;
}  // namespace

std::shared_ptr<Regex> get_regex(const std::string& pat) {
  std::lock_guard<std::mutex> l(g_re_mu);
  auto it = g_re_cache.find(pat);
  if (it != g_re_cache.end()) return it->second;
  std::shared_ptr<Regex> r;
  try { r = std::make_shared<Regex>(pat); } catch (const RegexError&) { r = nullptr; }   // invalid in Go: builtin error -> undefined
  // RegexUnsupported (valid Go, outside this engine) propagates: GK_ERR_UNSUPPORTED at AddConstraint, an error when rendering
  g_re_cache[pat] = r;
  return r;
}

bool builtin_regex_search(const std::string& pat, const char* s, size_t n, bool* valid) {
  // (a per-thread front of the locked cache: the ingest asks for the same handful of patterns millions of times)
  static thread_local std::unordered_map<std::string, std::shared_ptr<Regex>> tl;
  auto it = tl.find(pat);
  if (it == tl.end()) it = tl.emplace(pat, get_regex(pat)).first;
  if (!it->second) { *valid = false; return false; }
  *valid = true;
  return it->second->search(std::string(s, n));
}

namespace {

// utf-8 helpers: strings are byte strings; Rego's count/substring work on code points
size_t rune_count(const std::string& s) {
  size_t n = 0;
  for (unsigned char c : s) if ((c & 0xC0) != 0x80) n++;
  return n;
}
size_t rune_offset(const std::string& s, size_t runes) {
  size_t i = 0, n = 0;
  while (i < s.size() && n < runes) { i++; while (i < s.size() && ((unsigned char)s[i] & 0xC0) == 0x80) i++; n++; }
  return i;
}

Value norm_num(double d) { return Value::real(d); }

bool coll_items(const Value& c, ValueVec* out) {
  if (c.is_array() || c.is_set()) { *out = c.items(); return true; }
  return false;
}

std::string trim_set(const std::string& s, const std::string& cut, bool left, bool right) {
  size_t lo = 0, hi = s.size();
  if (left) while (lo < hi && cut.find(s[lo]) != std::string::npos) lo++;
  if (right) while (hi > lo && cut.find(s[hi - 1]) != std::string::npos) hi--;
  return s.substr(lo, hi - lo);
}

bool to_strs(const Value& v, std::vector<std::string>* out) {
  if (v.is_string()) { out->push_back(v.str()); return true; }
  ValueVec it;
  if (!coll_items(v, &it)) return false;
  for (const Value& x : it) { if (!x.is_string()) return false; out->push_back(x.str()); }
  return true;
}

Value parse_number_str(const std::string& x) {
  if (x.empty()) return U;
  size_t i = 0;
  if (x[i] == '+' || x[i] == '-') i++;
  size_t digits = 0, dot = 0, exp = 0, exp_digits = 0;
  for (; i < x.size(); i++) {
    char c = x[i];
    if (isdigit((unsigned char)c)) { digits++; exp_digits += exp; }
    else if (c == '.' && !dot && !exp) dot = 1;
    else if ((c == 'e' || c == 'E') && digits && !exp) { exp = 1; if (i + 1 < x.size() && (x[i + 1] == '+' || x[i + 1] == '-')) i++; }
    else return U;
  }
  if (!digits || (exp && !exp_digits)) return U;   // ("1e", "1e+": strconv.ParseFloat's syntax error)
  if (!dot && !exp && x.size() <= 37) {
    i128 v = 0; size_t k = 0; bool neg = false;
    if (x[0] == '-') { neg = true; k = 1; } else if (x[0] == '+') k = 1;
    for (; k < x.size(); k++) v = v * 10 + (x[k] - '0');
    return Value::integer(neg ? -v : v);
  }
  return norm_num(strtod(x.c_str(), nullptr));
}

typedef Value (*Fn)(const ValueVec&);
#define ARGS const ValueVec& a
#define NEED(n) if (a.size() != (n)) return U
#define STR(k) if (!a[k].is_string()) return U
#define NUM(k) if (!a[k].is_number()) return U

Value b_count(ARGS) {
  NEED(1);
  if (a[0].is_string()) return Value::integer((i128)rune_count(a[0].str()));
  if (a[0].is_array() || a[0].is_set() || a[0].is_object()) return Value::integer((i128)a[0].size());
  return U;
}
Value b_sum(ARGS) {
  NEED(1); ValueVec it; if (!coll_items(a[0], &it)) return U;
  bool all_int = true; i128 si = 0; double sd = 0;
  for (auto& v : it) { if (!v.is_number()) return U; if (v.is_int) si += v.i; else all_int = false; sd += v.as_double(); }
  return all_int ? Value::integer(si) : norm_num(sd);
}
Value b_product(ARGS) {
  NEED(1); ValueVec it; if (!coll_items(a[0], &it)) return U;
  bool all_int = true; i128 pi = 1; double pd = 1;
  for (auto& v : it) { if (!v.is_number()) return U; if (v.is_int) pi *= v.i; else all_int = false; pd *= v.as_double(); }
  return all_int ? Value::integer(pi) : norm_num(pd);
}
Value b_max(ARGS) { NEED(1); ValueVec it; if (!coll_items(a[0], &it) || it.empty()) return U; return *std::max_element(it.begin(), it.end()); }
Value b_min(ARGS) { NEED(1); ValueVec it; if (!coll_items(a[0], &it) || it.empty()) return U; return *std::min_element(it.begin(), it.end()); }
Value b_sort(ARGS) { NEED(1); ValueVec it; if (!coll_items(a[0], &it)) return U; std::sort(it.begin(), it.end()); return Value::array(it); }
Value b_any(ARGS) { NEED(1); ValueVec it; if (!coll_items(a[0], &it)) return U; for (auto& v : it) if (v.is_bool() && v.b) return Value::boolean(true); return Value::boolean(false); }
Value b_all(ARGS) { NEED(1); ValueVec it; if (!coll_items(a[0], &it)) return U; for (auto& v : it) if (!(v.is_bool() && v.b)) return Value::boolean(false); return Value::boolean(true); }
Value b_abs(ARGS) { NEED(1); NUM(0); return a[0].is_int ? Value::integer(a[0].i < 0 ? -a[0].i : a[0].i) : norm_num(std::fabs(a[0].d)); }
Value b_round(ARGS) { NEED(1); NUM(0); return a[0].is_int ? a[0] : norm_num(std::round(a[0].d)); }
Value b_ceil(ARGS) { NEED(1); NUM(0); return a[0].is_int ? a[0] : norm_num(std::ceil(a[0].d)); }
Value b_floor(ARGS) { NEED(1); NUM(0); return a[0].is_int ? a[0] : norm_num(std::floor(a[0].d)); }
Value b_sprintf(ARGS) { NEED(2); STR(0); if (!a[1].is_array()) return U; return Value::string(go_sprintf(a[0].str(), a[1].items())); }
Value b_concat(ARGS) {
  NEED(2); STR(0); ValueVec it; if (!coll_items(a[1], &it)) return U;
  std::string o;
  for (size_t k = 0; k < it.size(); k++) { if (!it[k].is_string()) return U; if (k) o += a[0].str(); o += it[k].str(); }
  return Value::string(o);
}
Value b_contains(ARGS) { NEED(2); STR(0); STR(1); return Value::boolean(a[0].str().find(a[1].str()) != std::string::npos); }
Value b_startswith(ARGS) { NEED(2); STR(0); STR(1); const std::string &s = a[0].str(), &p = a[1].str(); return Value::boolean(s.size() >= p.size() && s.compare(0, p.size(), p) == 0); }
Value b_endswith(ARGS) { NEED(2); STR(0); STR(1); const std::string &s = a[0].str(), &p = a[1].str(); return Value::boolean(s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0); }
// lower / upper: Go's strings.ToLower / ToUpper -- rune by rune, Unicode SIMPLE case mappings (unicode_case.inc)
#include "unicode_case.inc"
static uint32_t map_case(uint32_t cp, const uint32_t (*tab)[2], size_t n) {
  size_t lo = 0, hi = n;
  while (lo < hi) { size_t mid = (lo + hi) / 2; if (tab[mid][0] < cp) lo = mid + 1; else hi = mid; }
  return lo < n && tab[lo][0] == cp ? tab[lo][1] : cp;
}
static void put_utf8(std::string* o, uint32_t cp) {
  if (cp < 0x80) o->push_back((char)cp);
  else if (cp < 0x800) { o->push_back((char)(0xC0 | (cp >> 6))); o->push_back((char)(0x80 | (cp & 0x3F))); }
  else if (cp < 0x10000) { o->push_back((char)(0xE0 | (cp >> 12))); o->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o->push_back((char)(0x80 | (cp & 0x3F))); }
  else { o->push_back((char)(0xF0 | (cp >> 18))); o->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o->push_back((char)(0x80 | (cp & 0x3F))); }
}
static std::string map_case_string(const std::string& s, bool upper) {
  const uint32_t (*tab)[2] = upper ? kSimpleUpper : kSimpleLower;
  const size_t n = upper ? sizeof(kSimpleUpper) / sizeof(kSimpleUpper[0]) : sizeof(kSimpleLower) / sizeof(kSimpleLower[0]);
  std::string o;
  o.reserve(s.size());
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    size_t len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
    bool ok = len != 0 && i + len <= s.size();
    for (size_t k = 1; ok && k < len; k++) if (((unsigned char)s[i + k] & 0xC0) != 0x80) ok = false;
    if (!ok) { o.push_back((char)c); i++; continue; }   // (not UTF-8: the byte passes through; JSON strings are UTF-8)
    uint32_t cp = len == 1 ? c : len == 2 ? (c & 0x1F) : len == 3 ? (c & 0x0F) : (c & 0x07);
    for (size_t k = 1; k < len; k++) cp = (cp << 6) | ((unsigned char)s[i + k] & 0x3F);
    put_utf8(&o, map_case(cp, tab, n));
    i += len;
  }
  return o;
}
Value b_lower(ARGS) { NEED(1); STR(0); return Value::string(map_case_string(a[0].str(), false)); }
Value b_upper(ARGS) { NEED(1); STR(0); return Value::string(map_case_string(a[0].str(), true)); }
Value b_trim(ARGS) { NEED(2); STR(0); STR(1); return Value::string(trim_set(a[0].str(), a[1].str(), true, true)); }
Value b_trim_left(ARGS) { NEED(2); STR(0); STR(1); return Value::string(trim_set(a[0].str(), a[1].str(), true, false)); }
Value b_trim_right(ARGS) { NEED(2); STR(0); STR(1); return Value::string(trim_set(a[0].str(), a[1].str(), false, true)); }
Value b_trim_prefix(ARGS) { NEED(2); STR(0); STR(1); const std::string &s = a[0].str(), &p = a[1].str(); return Value::string(s.compare(0, p.size(), p) == 0 && s.size() >= p.size() ? s.substr(p.size()) : s); }
Value b_trim_suffix(ARGS) { NEED(2); STR(0); STR(1); const std::string &s = a[0].str(), &p = a[1].str(); return Value::string(s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0 ? s.substr(0, s.size() - p.size()) : s); }
// trim_space: Go's strings.TrimSpace = TrimFunc(s, unicode.IsSpace) -- White_Space code points, not only ASCII
static bool go_is_space(uint32_t cp) {
  return (cp >= 0x09 && cp <= 0x0D) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 ||
         cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
static bool decode_rune_at(const std::string& s, size_t i, uint32_t* cp, size_t* len) {   // false: not UTF-8 at i (the byte counts as itself)
  const unsigned char c = (unsigned char)s[i];
  size_t n = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
  if (n == 0 || i + n > s.size()) { *cp = c; *len = 1; return false; }
  uint32_t v = n == 1 ? c : n == 2 ? (c & 0x1F) : n == 3 ? (c & 0x0F) : (c & 0x07);
  for (size_t k = 1; k < n; k++) { if (((unsigned char)s[i + k] & 0xC0) != 0x80) { *cp = c; *len = 1; return false; } v = (v << 6) | ((unsigned char)s[i + k] & 0x3F); }
  *cp = v; *len = n;
  return true;
}
Value b_trim_space(ARGS) {
  NEED(1); STR(0);
  const std::string& s = a[0].str();
  size_t lo = 0, hi = s.size();
  while (lo < hi) { uint32_t cp; size_t n; decode_rune_at(s, lo, &cp, &n); if (!go_is_space(cp)) break; lo += n; }
  while (hi > lo) {
    size_t st = hi - 1;
    while (st > lo && ((unsigned char)s[st] & 0xC0) == 0x80 && hi - st < 4) st--;   // back to the rune's lead byte
    uint32_t cp; size_t n;
    decode_rune_at(s, st, &cp, &n);
    if (st + n != hi) { st = hi - 1; cp = (unsigned char)s[st]; }                    // (not UTF-8: the last byte by itself)
    if (!go_is_space(cp)) break;
    hi = st;
  }
  return Value::string(s.substr(lo, hi - lo));
}
Value b_split(ARGS) {
  NEED(2); STR(0); STR(1);
  const std::string &s = a[0].str(), &d = a[1].str();
  ValueVec out;
  if (d.empty()) { size_t i = 0; while (i < s.size()) { size_t j = rune_offset(s.substr(i), 1); out.push_back(Value::string(s.substr(i, j))); i += j; } return Value::array(out); }
  size_t pos = 0;
  for (;;) {
    size_t k = s.find(d, pos);
    if (k == std::string::npos) { out.push_back(Value::string(s.substr(pos))); break; }
    out.push_back(Value::string(s.substr(pos, k - pos)));
    pos = k + d.size();
  }
  return Value::array(out);
}
Value b_replace(ARGS) {
  NEED(3); STR(0); STR(1); STR(2);
  const std::string &s = a[0].str(), &o = a[1].str(), &n = a[2].str();
  std::string r;
  if (o.empty()) { r = n; size_t i = 0; while (i < s.size()) { size_t j = rune_offset(s.substr(i), 1); r += s.substr(i, j); r += n; i += j; } return Value::string(r); }
  size_t pos = 0;
  for (;;) {
    size_t k = s.find(o, pos);
    if (k == std::string::npos) { r += s.substr(pos); break; }
    r += s.substr(pos, k - pos);
    r += n;
    pos = k + o.size();
  }
  return Value::string(r);
}
Value b_substring(ARGS) {
  NEED(3); STR(0); NUM(1); NUM(2);
  const std::string& s = a[0].str();
  long long off = (long long)a[1].as_double(), len = (long long)a[2].as_double();
  if (off < 0) return U;
  size_t rc = rune_count(s);
  if ((size_t)off >= rc) return Value::string("");
  size_t b = rune_offset(s, off);
  if (len < 0) return Value::string(s.substr(b));
  size_t e = b + rune_offset(s.substr(b), len);
  return Value::string(s.substr(b, e - b));
}
Value b_indexof(ARGS) { NEED(2); STR(0); STR(1); size_t k = a[0].str().find(a[1].str()); return Value::integer(k == std::string::npos ? -1 : (i128)rune_count(a[0].str().substr(0, k))); }
Value b_format_int(ARGS) {
  NEED(2); NUM(0); NUM(1);
  int base = (int)a[1].as_double();
  if (base != 2 && base != 8 && base != 10 && base != 16) return U;
  if (a[0].is_int) return Value::string(to_base(a[0].i, base, false));
  // builtinFormatInt (OPA topdown/strings.go, restated): big.Float.Int -- TRUNCATED towards zero (-2.5 -> -2, -0.5 -> 0),
  // then %b / %o / %d / %x of the big.Int: a float64 beyond 128 bits still prints its exact integer value
  const double t = std::trunc(a[0].d);
  if (!std::isfinite(t)) return U;
  if (std::fabs(t) < 0x1p126) return Value::string(to_base((i128)t, base, false));
  int e2 = 0;
  const double m = std::frexp(std::fabs(t), &e2);                  // |t| = m * 2^e2, m in [0.5, 1): 53 bits of m are an integer
  unsigned long long mant = (unsigned long long)std::ldexp(m, 53);
  e2 -= 53;
  std::vector<uint32_t> dig;                                       // little-endian digits in `base`
  for (; mant; mant /= (unsigned)base) dig.push_back((uint32_t)(mant % (unsigned)base));
  for (int k = 0; k < e2; k++) {
    uint32_t carry = 0;
    for (auto& d : dig) { const uint32_t x = d * 2 + carry; d = x % (uint32_t)base; carry = x / (uint32_t)base; }
    if (carry) dig.push_back(carry);
  }
  std::string o = t < 0 ? "-" : "";
  for (size_t k = dig.size(); k-- > 0;) o += "0123456789abcdef"[dig[k]];
  return Value::string(o);
}
Value b_reverse(ARGS) {   // strings.reverse: rune by rune (builtinReverse converts to []rune)
  NEED(1); STR(0);
  const std::string& s = a[0].str();
  std::string o;
  o.reserve(s.size());
  for (size_t end = s.size(); end > 0;) {
    size_t b = end - 1;
    while (b > 0 && ((unsigned char)s[b] & 0xC0) == 0x80) b--;
    o.append(s, b, end - b);
    end = b;
  }
  return Value::string(o);
}
Value b_any_prefix(ARGS) {
  NEED(2); std::vector<std::string> ss, bs; if (!to_strs(a[0], &ss) || !to_strs(a[1], &bs)) return U;
  for (auto& s : ss) for (auto& b : bs) if (s.size() >= b.size() && s.compare(0, b.size(), b) == 0) return Value::boolean(true);
  return Value::boolean(false);
}
Value b_any_suffix(ARGS) {
  NEED(2); std::vector<std::string> ss, bs; if (!to_strs(a[0], &ss) || !to_strs(a[1], &bs)) return U;
  for (auto& s : ss) for (auto& b : bs) if (s.size() >= b.size() && s.compare(s.size() - b.size(), b.size(), b) == 0) return Value::boolean(true);
  return Value::boolean(false);
}
Value b_re_match(ARGS) { NEED(2); STR(0); STR(1); auto r = get_regex(a[0].str()); if (!r) return U; return Value::boolean(r->search(a[1].str())); }
Value b_regex_valid(ARGS) { NEED(1); if (!a[0].is_string()) return Value::boolean(false); return Value::boolean(get_regex(a[0].str()) != nullptr); }
Value b_is_string(ARGS) { NEED(1); return Value::boolean(a[0].is_string()); }
Value b_is_number(ARGS) { NEED(1); return Value::boolean(a[0].is_number()); }
Value b_is_boolean(ARGS) { NEED(1); return Value::boolean(a[0].is_bool()); }
Value b_is_array(ARGS) { NEED(1); return Value::boolean(a[0].is_array()); }
Value b_is_object(ARGS) { NEED(1); return Value::boolean(a[0].is_object()); }
Value b_is_set(ARGS) { NEED(1); return Value::boolean(a[0].is_set()); }
Value b_is_null(ARGS) { NEED(1); return Value::boolean(a[0].is_null()); }
Value b_type_name(ARGS) {
  NEED(1);
  static const char* names[] = {"null", "boolean", "number", "string", "array", "object", "set"};
  if (!a[0].defined()) return U;
  return Value::string(names[a[0].kind]);
}
Value b_to_number(ARGS) {
  NEED(1);
  if (a[0].is_null()) return Value::integer(0);
  if (a[0].is_bool()) return Value::integer(a[0].b ? 1 : 0);
  if (a[0].is_number()) return a[0];
  if (a[0].is_string()) return parse_number_str(a[0].str());
  return U;
}
Value b_object_get(ARGS) {
  NEED(3);
  if (!a[0].is_object()) return U;
  if (a[1].is_array()) {
    Value cur = a[0];
    for (const Value& k : a[1].items()) {
      if (cur.is_object()) { const Value* v = cur.get(k); if (!v) return a[2]; cur = *v; }
      else if (cur.is_array() && k.is_number() && k.is_int && k.i >= 0 && (size_t)k.i < cur.size()) cur = cur.items()[(size_t)k.i];
      else return a[2];
    }
    return cur;
  }
  const Value* v = a[0].get(a[1]);
  return v ? *v : a[2];
}
Value b_object_keys(ARGS) { NEED(1); if (!a[0].is_object()) return U; ValueVec ks; for (auto& kv : a[0].pairs()) ks.push_back(kv.first); return Value::set(ks); }
Value b_object_remove(ARGS) {
  NEED(2); if (!a[0].is_object()) return U;
  ValueVec ks;
  if (a[1].is_object()) for (auto& kv : a[1].pairs()) ks.push_back(kv.first); else if (!coll_items(a[1], &ks)) return U;
  Value kset = Value::set(ks);
  ValuePairs out;
  for (auto& kv : a[0].pairs()) if (!kset.set_has(kv.first)) out.push_back(kv);
  return Value::object(out);
}
Value obj_union(const Value& x, const Value& y) {
  ValuePairs out = x.pairs();
  for (auto& kv : y.pairs()) {
    const Value* cur = x.get(kv.first);
    if (cur && cur->is_object() && kv.second.is_object()) out.emplace_back(kv.first, obj_union(*cur, kv.second));
    else out.push_back(kv);
  }
  return Value::object(out);
}
Value b_object_union(ARGS) { NEED(2); if (!a[0].is_object() || !a[1].is_object()) return U; return obj_union(a[0], a[1]); }
Value b_array_concat(ARGS) { NEED(2); if (!a[0].is_array() || !a[1].is_array()) return U; ValueVec o = a[0].items(); o.insert(o.end(), a[1].items().begin(), a[1].items().end()); return Value::array(o); }
Value b_array_slice(ARGS) {
  NEED(3); if (!a[0].is_array()) return U; NUM(1); NUM(2);
  long long lo = std::max(0LL, (long long)a[1].as_double()), hi = std::min((long long)a[0].size(), (long long)a[2].as_double());
  ValueVec o;
  for (long long k = lo; k < hi; k++) o.push_back(a[0].items()[k]);
  return Value::array(o);
}
Value b_array_reverse(ARGS) { NEED(1); if (!a[0].is_array()) return U; ValueVec o = a[0].items(); std::reverse(o.begin(), o.end()); return Value::array(o); }
Value b_union(ARGS) { NEED(1); if (!a[0].is_set()) return U; ValueVec o; for (auto& s : a[0].items()) { if (!s.is_set()) return U; o.insert(o.end(), s.items().begin(), s.items().end()); } return Value::set(o); }
Value b_intersection(ARGS) {
  NEED(1); if (!a[0].is_set()) return U;
  if (a[0].size() == 0) return Value::set({});
  ValueVec cur = a[0].items()[0].items();
  for (size_t k = 1; k < a[0].size(); k++) { const Value& s = a[0].items()[k]; if (!s.is_set()) return U; ValueVec nx; for (auto& v : cur) if (s.set_has(v)) nx.push_back(v); cur = nx; }
  return Value::set(cur);
}
Value b_json_marshal(ARGS) { NEED(1); std::string o; to_json(a[0], o, /*go_marshal=*/true); return Value::string(o); }   // encoding/json.Marshal(ast.JSON(x))
Value b_json_unmarshal(ARGS) { NEED(1); STR(0); try { return parse_json(a[0].str()); } catch (const JsonError&) { return U; } }
Value b_true(ARGS) { (void)a; return Value::boolean(true); }

const std::map<std::string, Fn>& table() {
  static const std::map<std::string, Fn> t = {
      {"count", b_count}, {"sum", b_sum}, {"product", b_product}, {"max", b_max}, {"min", b_min}, {"sort", b_sort},
      {"any", b_any}, {"all", b_all}, {"abs", b_abs}, {"round", b_round}, {"ceil", b_ceil}, {"floor", b_floor},
      {"sprintf", b_sprintf}, {"concat", b_concat}, {"contains", b_contains}, {"startswith", b_startswith},
      {"endswith", b_endswith}, {"lower", b_lower}, {"upper", b_upper}, {"trim", b_trim}, {"trim_left", b_trim_left},
      {"trim_right", b_trim_right}, {"trim_prefix", b_trim_prefix}, {"trim_suffix", b_trim_suffix},
      {"trim_space", b_trim_space}, {"split", b_split}, {"replace", b_replace}, {"substring", b_substring},
      {"indexof", b_indexof}, {"format_int", b_format_int}, {"strings.reverse", b_reverse},
      {"strings.any_prefix_match", b_any_prefix}, {"strings.any_suffix_match", b_any_suffix},
      {"re_match", b_re_match}, {"regex.match", b_re_match}, {"regex.is_valid", b_regex_valid},
      {"is_string", b_is_string}, {"is_number", b_is_number}, {"is_boolean", b_is_boolean}, {"is_array", b_is_array},
      {"is_object", b_is_object}, {"is_set", b_is_set}, {"is_null", b_is_null}, {"type_name", b_type_name},
      {"to_number", b_to_number}, {"object.get", b_object_get}, {"object.keys", b_object_keys},
      {"object.remove", b_object_remove}, {"object.union", b_object_union}, {"array.concat", b_array_concat},
      {"array.slice", b_array_slice}, {"array.reverse", b_array_reverse}, {"union", b_union},
      {"intersection", b_intersection}, {"json.marshal", b_json_marshal}, {"json.unmarshal", b_json_unmarshal},
      {"print", b_true}, {"trace", b_true},
  };
  return t;
}

}  // namespace

bool has_builtin(const std::string& name) { return table().count(name) != 0; }

// Names OPA v1 defines as builtins (topdown + ast/builtins.go; third-party, restated from the published builtin reference)
// plus gatekeeper's own external_data.  A template calling one of these that this engine does not implement is VALID Rego:
// it must be refused as unsupported (the stock driver keeps it), not rejected as a type error.  http.send is left out on
// purpose: the reference deployment disables it (--disable-opa-builtin={http.send}) and pins the resulting
// "undefined function http.send" (test/bats/test.bats:492-498).
bool is_opa_builtin(const std::string& name) {
  static const char* const families[] = {"io.jwt.", "crypto.", "graphql.", "time.", "net.", "urlquery.", "base64url.", "base64.", "hex.", "yaml.", "json.",
                                         "providers.", "rego.", "opa.", "bits.", "units.", "semver.", "uuid.", "glob.", "graph.", "regex.", "strings.", "numbers.",
                                         "object.", "array.", "rand."};
  for (const char* f : families) if (name.rfind(f, 0) == 0) return true;
  static const std::set<std::string> names = {
      "abs", "ceil", "floor", "round", "count", "sum", "product", "max", "min", "sort", "all", "any", "and", "or", "intersection", "union", "concat", "contains",
      "endswith", "format_int", "indexof", "indexof_n", "lower", "replace", "split", "sprintf", "startswith", "substring", "trim", "trim_left", "trim_prefix",
      "trim_right", "trim_suffix", "trim_space", "upper", "re_match", "to_number", "cast_array", "cast_boolean", "cast_null", "cast_object", "cast_set",
      "cast_string", "is_array", "is_boolean", "is_null", "is_number", "is_object", "is_set", "is_string", "type_name", "walk", "trace", "print", "external_data",
      "set_diff"};
  return names.count(name) != 0;
}

Value call_builtin(const std::string& name, const ValueVec& args) {
  auto it = table().find(name);
  if (it == table().end()) return Value();
  for (const Value& v : args) if (!v.defined()) return Value();
  return it->second(args);
}

Value rego_arith(const std::string& op, const Value& a, const Value& b) {
  if (op == "-" && a.is_set() && b.is_set()) { ValueVec o; for (auto& v : a.items()) if (!b.set_has(v)) o.push_back(v); return Value::set(o); }
  if (op == "&") { if (!a.is_set() || !b.is_set()) return U; ValueVec o; for (auto& v : a.items()) if (b.set_has(v)) o.push_back(v); return Value::set(o); }
  if (op == "|") { if (!a.is_set() || !b.is_set()) return U; ValueVec o = a.items(); o.insert(o.end(), b.items().begin(), b.items().end()); return Value::set(o); }
  if (!a.is_number() || !b.is_number()) return U;
  bool ii = a.is_int && b.is_int;
  // (integers are 128-bit here; a result beyond that continues as a float instead of wrapping)
  i128 r;
  if (op == "+") return ii && !__builtin_add_overflow(a.i, b.i, &r) ? Value::integer(r) : norm_num(a.as_double() + b.as_double());
  if (op == "-") return ii && !__builtin_sub_overflow(a.i, b.i, &r) ? Value::integer(r) : norm_num(a.as_double() - b.as_double());
  if (op == "*") return ii && !__builtin_mul_overflow(a.i, b.i, &r) ? Value::integer(r) : norm_num(a.as_double() * b.as_double());
  if (op == "/") {
    if (b.as_double() == 0) return U;
    if (ii && a.i % b.i == 0) return Value::integer(a.i / b.i);
    return norm_num(a.as_double() / b.as_double());
  }
  if (op == "%") {   // builtinRem: both through NumberToInt (an integral float such as 1e21 IS an integer there), big.Int.Rem (truncated)
    auto as_int = [](const Value& v, i128* o) {
      if (v.is_int) { *o = v.i; return true; }
      const double d = v.as_double();
      if (!std::isfinite(d) || std::floor(d) != d || std::fabs(d) >= 1e38) return false;
      *o = (i128)d; return true;
    };
    i128 x, y;
    if (!as_int(a, &x) || !as_int(b, &y) || y == 0) return U;
    return Value::integer(x % y);
  }
  return U;
}

}  // namespace gk
