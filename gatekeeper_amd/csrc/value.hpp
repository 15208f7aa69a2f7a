// Value model shared by the JSON reader, the Rego front end (parser / AOT compiler) and the host-side message
// renderer.  Immutable, cheaply copyable (shared_ptr payloads).
//
// Replaces, for the MI355X engine, what the reference gets from OPA's ast.Value (github.com/open-policy-agent/opa
// v1.17.1, go.mod:19): null < boolean < number < string < array < object < set total order, value equality,
// and ast term String() rendering used by sprintf("%v") (pinned by website/docs/constrainttemplates.md:118).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace gk {

typedef __int128 i128;

struct Value;
typedef std::vector<Value> ValueVec;
typedef std::vector<std::pair<Value, Value>> ValuePairs;

struct Value {
  enum Kind : uint8_t { Null = 0, Bool = 1, Number = 2, String = 3, Array = 4, Object = 5, Set = 6, Undefined = 7 };
  Kind kind = Undefined;
  bool b = false;
  bool is_int = true;   // Number: exact integer in `i`, otherwise double in `d`
  i128 i = 0;
  double d = 0;
  std::shared_ptr<const std::string> s;
  std::shared_ptr<const ValueVec> arr;     // Array: in order; Set: sorted unique
  std::shared_ptr<const ValuePairs> obj;   // Object: sorted by key

  Value() {}
  static Value null() { Value v; v.kind = Null; return v; }
  static Value boolean(bool x) { Value v; v.kind = Bool; v.b = x; return v; }
  static Value integer(i128 x) { Value v; v.kind = Number; v.is_int = true; v.i = x; v.d = (double)x; return v; }
  static Value real(double x) {
    Value v; v.kind = Number;
    if (std::isfinite(x) && std::floor(x) == x && std::fabs(x) < 1e21) { v.is_int = true; v.i = (i128)x; v.d = x; }
    else { v.is_int = false; v.d = x; }
    return v;
  }
  static Value string(std::string x) { Value v; v.kind = String; v.s = std::make_shared<const std::string>(std::move(x)); return v; }
  static Value array(ValueVec x) { Value v; v.kind = Array; v.arr = std::make_shared<const ValueVec>(std::move(x)); return v; }
  static Value set(ValueVec x);
  static Value object(ValuePairs x);

  bool defined() const { return kind != Undefined; }
  bool is_null() const { return kind == Null; }
  bool is_bool() const { return kind == Bool; }
  bool is_number() const { return kind == Number; }
  bool is_string() const { return kind == String; }
  bool is_array() const { return kind == Array; }
  bool is_object() const { return kind == Object; }
  bool is_set() const { return kind == Set; }
  const std::string& str() const { return *s; }
  const ValueVec& items() const { return *arr; }
  const ValuePairs& pairs() const { return *obj; }
  double as_double() const { return is_int ? (double)i : d; }
  size_t size() const {
    if (kind == Array || kind == Set) return arr->size();
    if (kind == Object) return obj->size();
    if (kind == String) return s->size();
    return 0;
  }
  const Value* get(const Value& key) const;       // object lookup
  const Value* get(const char* key) const;        // object lookup by a string key (no Value is made for the key)
  bool set_has(const Value& v) const;
};

inline int compare(const Value& a, const Value& b);

inline int cmp_num(const Value& a, const Value& b) {
  if (a.is_int && b.is_int) return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
  double x = a.as_double(), y = b.as_double();
  return x < y ? -1 : (x > y ? 1 : 0);
}

inline int compare(const Value& a, const Value& b) {
  if (a.kind != b.kind) return a.kind < b.kind ? -1 : 1;
  switch (a.kind) {
    case Value::Null: case Value::Undefined: return 0;
    case Value::Bool: return (int)a.b - (int)b.b;
    case Value::Number: return cmp_num(a, b);
    case Value::String: { int c = a.s->compare(*b.s); return c < 0 ? -1 : (c > 0 ? 1 : 0); }
    case Value::Array: case Value::Set: {
      size_t n = std::min(a.arr->size(), b.arr->size());
      for (size_t k = 0; k < n; k++) { int c = compare((*a.arr)[k], (*b.arr)[k]); if (c) return c; }
      return a.arr->size() < b.arr->size() ? -1 : (a.arr->size() > b.arr->size() ? 1 : 0);
    }
    case Value::Object: {
      size_t n = std::min(a.obj->size(), b.obj->size());
      for (size_t k = 0; k < n; k++) {
        int c = compare((*a.obj)[k].first, (*b.obj)[k].first); if (c) return c;
        c = compare((*a.obj)[k].second, (*b.obj)[k].second); if (c) return c;
      }
      return a.obj->size() < b.obj->size() ? -1 : (a.obj->size() > b.obj->size() ? 1 : 0);
    }
  }
  return 0;
}
inline bool operator==(const Value& a, const Value& b) { return compare(a, b) == 0; }
inline bool operator!=(const Value& a, const Value& b) { return compare(a, b) != 0; }
inline bool operator<(const Value& a, const Value& b) { return compare(a, b) < 0; }

inline Value Value::set(ValueVec x) {
  std::sort(x.begin(), x.end());
  x.erase(std::unique(x.begin(), x.end()), x.end());
  Value v; v.kind = Set; v.arr = std::make_shared<const ValueVec>(std::move(x)); return v;
}
inline Value Value::object(ValuePairs x) {
  // sorted by key, stable (the later of two equal keys stays behind the earlier one), then last write wins -- all in place: small
  // objects (nearly all of them) by insertion, which needs no scratch buffer; no second vector for the result
  bool sorted = true;
  for (size_t i = 1; i < x.size() && sorted; i++) if (!(x[i - 1].first < x[i].first)) sorted = false;   // (strictly ascending: nothing to do at all)
  if (!sorted) {
    if (x.size() <= 24) {
      for (size_t i = 1; i < x.size(); i++) {
        size_t j = i;
        while (j > 0 && x[i].first < x[j - 1].first) j--;
        if (j != i) std::rotate(x.begin() + j, x.begin() + i, x.begin() + i + 1);
      }
    } else std::stable_sort(x.begin(), x.end(), [](const std::pair<Value, Value>& p, const std::pair<Value, Value>& q) { return p.first < q.first; });
    size_t w = 0;
    for (size_t i = 0; i < x.size(); i++) {
      if (w > 0 && x[w - 1].first == x[i].first) x[w - 1].second = std::move(x[i].second);
      else { if (w != i) x[w] = std::move(x[i]); w++; }
    }
    x.resize(w);
  }
  Value v; v.kind = Object; v.obj = std::make_shared<const ValuePairs>(std::move(x)); return v;
}
inline const Value* Value::get(const Value& key) const {
  if (kind != Object) return nullptr;
  auto it = std::lower_bound(obj->begin(), obj->end(), key, [](const std::pair<Value, Value>& p, const Value& k) { return p.first < k; });
  if (it != obj->end() && it->first == key) return &it->second;
  return nullptr;
}
inline const Value* Value::get(const char* key) const {
  if (kind != Object) return nullptr;
  const size_t kn = strlen(key);
  // keys sort by kind first (null < boolean < number < string < ..), strings by their bytes
  auto less = [&](const std::pair<Value, Value>& p, int) {
    if (p.first.kind != String) return p.first.kind < String;
    return p.first.s->compare(0, std::string::npos, key, kn) < 0;
  };
  auto it = std::lower_bound(obj->begin(), obj->end(), 0, less);
  if (it != obj->end() && it->first.kind == String && it->first.s->size() == kn && memcmp(it->first.s->data(), key, kn) == 0) return &it->second;
  return nullptr;
}
inline bool Value::set_has(const Value& v) const {
  if (kind != Set) return false;
  return std::binary_search(arr->begin(), arr->end(), v);
}

// ---------------------------------------------------------------------------------------------- rendering
inline std::string i128_to_string(i128 x) {
  if (x == 0) return "0";
  bool neg = x < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-(x + 1)) + 1 : (unsigned __int128)x;
  std::string s;
  while (u) { s.push_back('0' + (int)(u % 10)); u /= 10; }
  if (neg) s.push_back('-');
  std::reverse(s.begin(), s.end());
  return s;
}

// shortest round-trip decimal digits of f (finite, != 0): sign, digits without trailing zeros, decimal exponent of the first digit
inline void shortest_digits(double f, std::string* sign, std::string* digits, int* x) {
  char buf[64];
  for (int prec = 1; prec <= 17; prec++) { snprintf(buf, sizeof buf, "%.*e", prec - 1, f); if (strtod(buf, nullptr) == f) break; }
  std::string r(buf);
  const size_t epos = r.find('e');
  std::string mant = r.substr(0, epos);
  *x = atoi(r.c_str() + epos + 1);
  sign->clear();
  if (mant[0] == '-') { *sign = "-"; mant = mant.substr(1); }
  digits->clear();
  for (char c : mant) if (c != '.') digits->push_back(c);
  while (digits->size() > 1 && digits->back() == '0') digits->pop_back();
}
inline std::string float_e_form(const std::string& sign, const std::string& digits, int x) {
  std::string m = digits.substr(0, 1);
  if (digits.size() > 1) m += "." + digits.substr(1);
  char e[16]; snprintf(e, sizeof e, "e%c%02d", x >= 0 ? '+' : '-', std::abs(x));
  return sign + m + e;
}
inline std::string float_f_form(const std::string& sign, const std::string& digits, int x) {
  if (x >= 0) {
    if ((int)digits.size() <= x + 1) return sign + digits + std::string(x + 1 - digits.size(), '0');
    return sign + digits.substr(0, x + 1) + "." + digits.substr(x + 1);
  }
  return sign + "0." + std::string(-x - 1, '0') + digits;
}

// Go fmt %v of a float64 = strconv.FormatFloat(f, 'g', -1, 64): shortest digits, the %e form when the decimal exponent is
// < -4 or >= 6 (with the shortest precision strconv's %g decision uses precision 6): 6e+11, 1.2345675e+06, 123456.5
inline std::string go_float_v(double f) {
  if (f != f) return "NaN";
  if (std::isinf(f)) return f > 0 ? "+Inf" : "-Inf";
  if (f == 0) return std::signbit(f) ? "-0" : "0";
  std::string sign, digits; int x;
  shortest_digits(f, &sign, &digits, &x);
  return (x < -4 || x >= 6) ? float_e_form(sign, digits, x) : float_f_form(sign, digits, x);
}

// encoding/json's floatEncoder (the text a float64 of a review object carries as an ast.Number: objects reach OPA through
// a JSON round trip): 'f' with the shortest digits; 'e' when abs < 1e-6 or abs >= 1e21, a one-digit negative exponent
// without its leading zero (e-09 -> e-9)
inline std::string json_float_text(double f) {
  if (f != f || std::isinf(f)) return go_float_v(f);
  if (f == 0) return std::signbit(f) ? "-0" : "0";
  std::string sign, digits; int x;
  shortest_digits(f, &sign, &digits, &x);
  if (std::fabs(f) < 1e-6 || std::fabs(f) >= 1e21) {
    std::string s = float_e_form(sign, digits, x);
    const size_t n = s.size();
    if (n >= 4 && s[n - 4] == 'e' && s[n - 3] == '-' && s[n - 2] == '0') s = s.substr(0, n - 2) + s[n - 1];
    return s;
  }
  return float_f_form(sign, digits, x);
}

inline std::string num_to_string(const Value& v) { return v.is_int ? i128_to_string(v.i) : json_float_text(v.d); }   // ast.Number.String()

// Go strconv.Quote (ast.String.String()).
// strconv.IsPrint for runes >= 0x80 (categories L, M, N, P, S): tools/gen_unicode_print.py
#include "unicode_print.inc"
inline bool go_is_print(uint32_t r) {
  const size_t n = sizeof kPrintRanges / sizeof kPrintRanges[0];
  size_t lo = 0, hi = n;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (kPrintRanges[mid][1] < r) lo = mid + 1; else hi = mid; }
  return lo < n && kPrintRanges[lo][0] <= r;
}

// strconv.Quote (ast.String.String(), fmt %q): a printable rune is written as it is; \a \b \f \n \r \t \v; \xNN for the other ASCII
// controls and for a byte that is no UTF-8; \uNNNN / \UNNNNNNNN for any other rune IsPrint refuses (NBSP, zero-width and other format
// characters, line separators, unassigned and private-use code points)
inline std::string go_quote(const std::string& s) {
  std::string o = "\"";
  char b[16];
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\t': o += "\\t"; break;
        case '\r': o += "\\r"; break;
        case '\a': o += "\\a"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\v': o += "\\v"; break;
        default:
          if (c < 0x20 || c == 0x7f) { snprintf(b, sizeof b, "\\x%02x", c); o += b; }
          else o.push_back((char)c);
      }
      i++;
      continue;
    }
    const int len = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 0;
    uint32_t r = len == 4 ? c & 7u : len == 3 ? c & 15u : c & 31u;
    bool ok = len != 0 && i + (size_t)len <= s.size();
    for (int k = 1; ok && k < len; k++) {
      const unsigned char d = (unsigned char)s[i + (size_t)k];
      if ((d & 0xC0) != 0x80) ok = false; else r = (r << 6) | (d & 63u);
    }
    if (ok && (r < (len == 2 ? 0x80u : len == 3 ? 0x800u : 0x10000u) || r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF))) ok = false;
    if (!ok) { snprintf(b, sizeof b, "\\x%02x", c); o += b; i++; continue; }   // (DecodeRune's RuneError of width 1)
    if (go_is_print(r)) o.append(s, i, (size_t)len);
    else { snprintf(b, sizeof b, r < 0x10000 ? "\\u%04x" : "\\U%08x", r); o += b; }
    i += (size_t)len;
  }
  o.push_back('"');
  return o;
}

// OPA ast term String()
inline std::string to_term_string(const Value& v) {
  switch (v.kind) {
    case Value::Null: return "null";
    case Value::Undefined: return "undefined";
    case Value::Bool: return v.b ? "true" : "false";
    case Value::Number: return num_to_string(v);
    case Value::String: return go_quote(*v.s);
    case Value::Array: {
      std::string o = "[";
      for (size_t k = 0; k < v.arr->size(); k++) { if (k) o += ", "; o += to_term_string((*v.arr)[k]); }
      return o + "]";
    }
    case Value::Set: {
      if (v.arr->empty()) return "set()";
      std::string o = "{";
      for (size_t k = 0; k < v.arr->size(); k++) { if (k) o += ", "; o += to_term_string((*v.arr)[k]); }
      return o + "}";
    }
    case Value::Object: {
      std::string o = "{";
      for (size_t k = 0; k < v.obj->size(); k++) {
        if (k) o += ", ";
        o += to_term_string((*v.obj)[k].first) + ": " + to_term_string((*v.obj)[k].second);
      }
      return o + "}";
    }
  }
  return "";
}

// ---------------------------------------------------------------------------------------------- JSON
struct JsonError : std::runtime_error { using std::runtime_error::runtime_error; };

// go_marshal: as encoding/json's Marshal writes strings (the json.marshal builtin) -- additionally < > & as \u003c \u003e \u0026
// (EscapeHTML is on by default) and U+2028 / U+2029 as \u2028 / \u2029
inline void json_escape(const std::string& s, std::string& o, bool go_marshal = false) {
  o.push_back('"');
  for (size_t i = 0; i < s.size(); i++) {
    const unsigned char c = (unsigned char)s[i];
    if (go_marshal) {
      if (c == '<' || c == '>' || c == '&') { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; continue; }
      if (c == 0xE2 && i + 2 < s.size() && (unsigned char)s[i + 1] == 0x80 && ((unsigned char)s[i + 2] == 0xA8 || (unsigned char)s[i + 2] == 0xA9)) {
        o += (unsigned char)s[i + 2] == 0xA8 ? "\\u2028" : "\\u2029";
        i += 2;
        continue;
      }
    }
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\t': o += "\\t"; break;
      case '\r': o += "\\r"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
  }
  o.push_back('"');
}

// JSON text of a value; sets become sorted arrays (OPA ast.JSON), non-string object keys use their term string.
inline void to_json(const Value& v, std::string& o, bool go_marshal = false) {
  switch (v.kind) {
    case Value::Null: case Value::Undefined: o += "null"; break;
    case Value::Bool: o += v.b ? "true" : "false"; break;
    case Value::Number: o += num_to_string(v); break;
    case Value::String: json_escape(*v.s, o, go_marshal); break;
    case Value::Array: case Value::Set:
      o.push_back('[');
      for (size_t k = 0; k < v.arr->size(); k++) { if (k) o.push_back(','); to_json((*v.arr)[k], o, go_marshal); }
      o.push_back(']');
      break;
    case Value::Object:
      o.push_back('{');
      for (size_t k = 0; k < v.obj->size(); k++) {
        if (k) o.push_back(',');
        const Value& key = (*v.obj)[k].first;
        json_escape(key.is_string() ? *key.s : to_term_string(key), o, go_marshal);
        o.push_back(':');
        to_json((*v.obj)[k].second, o, go_marshal);
      }
      o.push_back('}');
      break;
  }
}
inline std::string to_json(const Value& v) { std::string o; to_json(v, o); return o; }

class JsonParser {
 public:
  JsonParser(const char* p, size_t n) : p_(p), e_(p + n) {}
  Value parse() {
    ws();
    Value v = value(0);
    ws();
    if (p_ != e_) fail("trailing characters");
    return v;
  }

 private:
  const char *p_, *e_;
  [[noreturn]] void fail(const char* m) { throw JsonError(std::string("invalid JSON: ") + m); }
  void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++; }
  Value value(int depth) {
    if (depth > 512) fail("nesting too deep");
    if (p_ >= e_) fail("unexpected end");
    char c = *p_;
    if (c == '{') {
      p_++; ws();
      ValuePairs pairs;
      if (p_ < e_ && *p_ == '}') { p_++; return Value::object(std::move(pairs)); }
      pairs.reserve(8);
      for (;;) {
        ws();
        if (p_ >= e_ || *p_ != '"') fail("expected object key");
        Value kv = key();
        ws();
        if (p_ >= e_ || *p_ != ':') fail("expected ':'");
        p_++; ws();
        Value v = value(depth + 1);
        pairs.emplace_back(std::move(kv), std::move(v));
        ws();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == '}') { p_++; break; }
        fail("expected ',' or '}'");
      }
      return Value::object(std::move(pairs));
    }
    if (c == '[') {
      p_++; ws();
      ValueVec items;
      if (p_ < e_ && *p_ == ']') { p_++; return Value::array(std::move(items)); }
      for (;;) {
        ws();
        items.push_back(value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == ']') { p_++; break; }
        fail("expected ',' or ']'");
      }
      return Value::array(std::move(items));
    }
    if (c == '"') return key();   // (string values repeat as well -- images, protocols, label values: the same table serves them)
    if (c == 't') { lit("true"); return Value::boolean(true); }
    if (c == 'f') { lit("false"); return Value::boolean(false); }
    if (c == 'n') { lit("null"); return Value::null(); }
    return number();
  }
  // A string as a Value.  Member names (and many values) repeat from document to document: short strings without escapes come
  // from a small per-thread table of shared string values (no allocation, no copy of the bytes); anything else is decoded.
  Value key() {
    const char* q = p_ + 1;
    while (q < e_ && *q != '"' && *q != '\\' && q - p_ <= 40) q++;
    if (q >= e_ || *q != '"') return Value::string(str());
    const char* s = p_ + 1;
    const size_t n = (size_t)(q - s);
    struct Ent { uint64_t h = 0; Value v; };
    static thread_local std::vector<Ent> tab(1024);
    uint64_t h = 1469598103934665603ull ^ (n * 0x9E3779B97F4A7C15ull);
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ull; }
    h |= 1;
    Ent& e = tab[(size_t)(h >> 20) & 1023];
    if (e.h != h || e.v.s->size() != n || memcmp(e.v.s->data(), s, n) != 0) { e.h = h; e.v = Value::string(std::string(s, n)); }
    p_ = q + 1;
    return e.v;
  }
  void lit(const char* w) {
    size_t n = strlen(w);
    if ((size_t)(e_ - p_) < n || memcmp(p_, w, n) != 0) fail("bad literal");
    p_ += n;
  }
  Value number() {
    const char* s = p_;
    bool is_int = true;
    if (p_ < e_ && *p_ == '-') p_++;
    if (p_ >= e_ || !(*p_ >= '0' && *p_ <= '9')) fail("bad number");
    while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
    if (p_ < e_ && *p_ == '.') { is_int = false; p_++; while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++; }
    if (p_ < e_ && (*p_ == 'e' || *p_ == 'E')) {
      is_int = false; p_++;
      if (p_ < e_ && (*p_ == '+' || *p_ == '-')) p_++;
      while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
    }
    std::string t(s, p_ - s);
    if (is_int && t.size() <= 37) {
      i128 x = 0; size_t k = 0; bool neg = false;
      if (t[0] == '-') { neg = true; k = 1; }
      for (; k < t.size(); k++) x = x * 10 + (t[k] - '0');
      return Value::integer(neg ? -x : x);
    }
    return Value::real(strtod(t.c_str(), nullptr));
  }
  static void utf8(std::string& o, uint32_t cp) {
    if (cp < 0x80) o.push_back((char)cp);
    else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
  }
  uint32_t hex4() {
    if (e_ - p_ < 4) fail("bad \\u escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string str() {
    p_++;  // opening quote
    std::string o;
    for (;;) {
      if (p_ >= e_) fail("unterminated string");
      const char* q = p_;
      while (q < e_ && *q != '"' && *q != '\\') q++;
      o.append(p_, q - p_);
      p_ = q;
      if (p_ >= e_) fail("unterminated string");
      if (*p_ == '"') { p_++; return o; }
      p_++;
      if (p_ >= e_) fail("bad escape");
      char c = *p_++;
      switch (c) {
        case '"': o.push_back('"'); break;
        case '\\': o.push_back('\\'); break;
        case '/': o.push_back('/'); break;
        case 'b': o.push_back('\b'); break;
        case 'f': o.push_back('\f'); break;
        case 'n': o.push_back('\n'); break;
        case 'r': o.push_back('\r'); break;
        case 't': o.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            p_ += 2;
            uint32_t lo = hex4();
            if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            else { utf8(o, 0xFFFD); cp = lo; }
          }
          utf8(o, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
  }
};

inline Value parse_json(const char* p, size_t n) { return JsonParser(p, n).parse(); }
inline Value parse_json(const std::string& s) { return parse_json(s.data(), s.size()); }

}  // namespace gk
