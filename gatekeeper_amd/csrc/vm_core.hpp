// Host/device evaluation core: predicate evaluation for one row (phase 1) and the formula interpreter for one
// review (phase 2).  Compiled by hipcc into kernels.hip (the product path) and by g++ into the TEST-ONLY CPU
// emulator tests/native/hostemu.cpp, which exists so the compiler + flattener can be checked against the oracle
// in the GPU-less build container.  The product library never links the emulator.
#pragma once
#include "plan.hpp"

#if defined(__HIPCC__)
#define GK_HD __host__ __device__ inline
#define GK_HD_COLD __host__ __device__ inline __attribute__((noinline))   // rare slow paths: keep them out of line
#else
#define GK_HD inline
#define GK_HD_COLD inline
#endif
// The formula interpreter's control flow is wave-uniform by construction (same bytecode, same loop bounds for all 64
// lanes).  GK_UNI makes that visible to the compiler so the program counter, the decoded instruction and the loop
// counters live in SGPRs and the dispatch is scalar branching instead of exec-mask divergence.
#if defined(__HIPCC__)
#define GK_CONST_ARRAY __device__ const
#else
#define GK_CONST_ARRAY static const
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define GK_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define GK_UNI(x) (x)
#endif

namespace gk {

#ifdef GK_COUNT_OPS
static unsigned long long gk_op_counter = 0;   // test-only instrumentation (hostemu)
#endif

struct PlanView {
  const uint32_t* ptab;        // [n_paths] (first << 8 | count) into preds; 0 = no predicates
  const Pred* preds;           // predicates grouped by path (a predicate whose pattern matches k paths appears k times)
  const Scope* scopes;
  const uint32_t* code;
  const uint8_t* cheap;        // constant heap
  PlanDims dims;
};

GK_HD uint32_t row_type(const Row& r) { return r.meta & ROW_TYPE_MASK; }
GK_HD uint32_t row_ordinal(const Row& r, uint32_t level) { return level >= GK_LEVEL_ROOT ? 0u : (r.meta >> (ROW_E_SHIFT0 + 8 * level)) & ROW_E_MASK; }
GK_HD uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
GK_HD int64_t row_i64(const Row& r) { return (int64_t)(((uint64_t)r.hi << 32) | r.lo); }
GK_HD double bits_f64(uint64_t b) { return __builtin_bit_cast(double, b); }
GK_HD double row_f64(const Row& r) { return bits_f64(((uint64_t)r.hi << 32) | r.lo); }

// ------------------------------------------------------------------------------------------------ strings
// A string row is either INLINE (ROW_STR_INLINE: length <= 7, bytes packed in lo/hi, no memory access at all -- kinds,
// short names, label values ...) or a HEAP string: lo = byte offset of a 16-byte aligned, zero-padded entry
// [u32 len][bytes][pad], hi = hash32.  The 16-byte entry header (length + first 12 bytes) is fetched with ONE aligned
// load, convergently for all lanes of a wave BEFORE the divergent predicate dispatch (kernel_body.inc), so most
// predicates are decided without a dependent memory access inside a divergent branch.

GK_HD bool row_needs_hdr(const Row& r) { return (r.meta & ROW_TYPE_MASK) == T_STRING && !(r.meta & ROW_STR_INLINE); }
GK_HD StrHdr load_hdr(const Row& r, const uint8_t* heap) {
  StrHdr h;
  const uint8_t* p = heap + r.lo - 4;
  h.w[0] = ld32(p); h.w[1] = ld32(p + 4); h.w[2] = ld32(p + 8); h.w[3] = ld32(p + 12);
  return h;
}

struct StrRef {
  uint32_t n;          // length in bytes
  uint64_t bits;       // first 8 bytes, zero padded (all of an inline string)
  uint32_t w2;         // bytes 8..11 (heap strings)
  uint32_t hash;       // heap strings only
  const uint8_t* p;    // heap bytes (nullptr for inline strings)
};
GK_HD StrRef make_str(const Row& r, const StrHdr& h, const uint8_t* heap) {
  StrRef s;
  if (r.meta & ROW_STR_INLINE) {
    s.n = r.hi >> 24; s.bits = ((uint64_t)(r.hi & 0x00FFFFFFu) << 32) | r.lo; s.w2 = 0; s.hash = 0; s.p = nullptr;
  } else {
    s.n = h.w[0]; s.bits = ((uint64_t)h.w[2] << 32) | h.w[1]; s.w2 = h.w[3]; s.hash = r.hi; s.p = heap + r.lo;
#ifdef GK_NO_HEAP   // TIMING AID (wrong answers): string bytes beyond the header come from one cached line instead of the row's heap entry
    s.p = heap + (r.lo & 48u);
#endif
  }
  return s;
}
GK_HD uint32_t sbyte(const StrRef& s, uint32_t i) {
  if (i < 8) return (uint32_t)(s.bits >> (8 * i)) & 0xFFu;
  if (i < 12) return (s.w2 >> (8 * (i - 8))) & 0xFFu;
  return s.p[i];
}
GK_HD uint64_t mask_bytes(uint32_t m) { return m >= 8 ? ~0ull : ((1ull << (8 * m)) - 1ull); }

// constant strings: bytes at cheap + off (16-byte aligned, zero padded), length len, key = packed bytes (len <= 7) or hash32
GK_HD bool str_eq_c(const StrRef& s, const uint8_t* c, uint32_t len, uint64_t key) {
  if (s.n != len) return false;
  if (len <= 7) return s.bits == key;
  if (s.hash != (uint32_t)key) return false;
  uint32_t d = ((uint32_t)s.bits ^ ld32(c)) | ((uint32_t)(s.bits >> 32) ^ ld32(c + 4)) | (s.w2 ^ ld32(c + 8));
  for (uint32_t j = 12; j < len; j += 4) d |= ld32(s.p + j) ^ ld32(c + j);
  return d == 0;
}
GK_HD bool str_prefix_c(const StrRef& s, const uint8_t* c, uint32_t m, uint64_t key) {
  if (m == 0) return true;
  if (s.n < m) return false;
  if (m <= 7) return ((s.bits ^ key) & mask_bytes(m)) == 0;
  uint32_t d = ((uint32_t)s.bits ^ ld32(c)) | ((uint32_t)(s.bits >> 32) ^ ld32(c + 4));
  if (m <= 12) {
    uint32_t r = m - 8;
    uint32_t mk = r == 4 ? ~0u : ((1u << (8 * r)) - 1u);
    return (d | ((s.w2 ^ ld32(c + 8)) & mk)) == 0;
  }
  d |= s.w2 ^ ld32(c + 8);
  uint32_t full = m & ~3u;
  for (uint32_t j = 12; j < full; j += 4) d |= ld32(s.p + j) ^ ld32(c + j);
  uint32_t r = m & 3u;
  if (r) d |= (ld32(s.p + full) ^ ld32(c + full)) & ((1u << (8 * r)) - 1u);
  return d == 0;
}
// Word access.  A heap entry is 16-byte aligned and zero padded ([u32 len][bytes][pad]) and the table heap ends in 16 B of
// slack, so whole aligned words -- also the one that straddles the end of the string -- can be read; bytes beyond the
// string are masked by the callers.  Byte-wise access costs one dependent memory round trip PER BYTE on the device (the
// compiler does not merge byte loads): with words, the loads of one comparison are independent and wait once.
GK_HD uint32_t sword(const StrRef& s, uint32_t j) {   // bytes [j, j+4), j a multiple of 4
  if (j < 8) return (uint32_t)(s.bits >> (8 * j));
  if (j == 8) return s.w2;
  return s.p ? ld32(s.p + j) : 0u;
}
GK_HD uint64_t swin(const StrRef& s, uint32_t at, uint32_t m) {   // bytes [at, at+m), m <= 8, any alignment; the bytes above m are unspecified
  const uint32_t a = at & ~3u, sh = (at & 3u) * 8u, end = at + m;   // only the words that hold wanted bytes are read: nothing beyond
  const uint32_t w0 = sword(s, a);                                  // the word of the string's last byte is ever touched
  const uint32_t w1 = end > a + 4u ? sword(s, a + 4u) : 0u, w2 = end > a + 8u ? sword(s, a + 8u) : 0u;
  const uint64_t lo = ((uint64_t)w1 << 32) | w0;
  return sh ? (lo >> sh) | ((uint64_t)w2 << (64u - sh)) : lo;
}
GK_HD uint64_t cwin(const uint8_t* c, uint32_t m) {   // up to 8 constant bytes as a little-endian word (folds for constexpr predicates)
  uint64_t v = 0;
  for (uint32_t i = 0; i < m && i < 8; i++) v |= (uint64_t)c[i] << (8 * i);
  return v;
}
// m bytes of s starting at byte `at` equal the constant bytes c[0..m)
GK_HD bool str_at_c(const StrRef& s, uint32_t at, const uint8_t* c, uint32_t m) {
  uint64_t d = 0;
  for (uint32_t i = 0; i < m; i += 8) {
    const uint32_t k = m - i < 8 ? m - i : 8;
    d |= (swin(s, at + i, k) ^ cwin(c + i, k)) & mask_bytes(k);
  }
  return d == 0;
}
// does the constant c[0..m), 1 <= m <= 8, occur anywhere in s?  One new aligned word per four positions.
GK_HD bool str_contains_short(const StrRef& s, const uint8_t* c, uint32_t m) {
  if (m > s.n) return false;
  const uint64_t want = cwin(c, m), mk = mask_bytes(m);
  const uint32_t last = s.n - m;   // last start position
  uint32_t w0 = sword(s, 0), w1 = sword(s, 4), w2 = sword(s, 8);   // (header words: no memory access)
  bool any = false;
  for (uint32_t a = 0; a <= last; a += 4) {
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    any = any || ((lo ^ want) & mk) == 0;
    if (a + 1 <= last) any = any || ((((lo >> 8) | ((uint64_t)w2 << 56)) ^ want) & mk) == 0;
    if (a + 2 <= last) any = any || ((((lo >> 16) | ((uint64_t)w2 << 48)) ^ want) & mk) == 0;
    if (a + 3 <= last) any = any || ((((lo >> 24) | ((uint64_t)w2 << 40)) ^ want) & mk) == 0;
    w0 = w1; w1 = w2; w2 = a + 12u < s.n ? sword(s, a + 12u) : 0u;   // (a start position in the next round needs bytes < n only)
  }
  return any;
}
GK_HD int str_cmp_c(const StrRef& s, uint32_t at, uint32_t n, const uint8_t* c, uint32_t nc) {   // ordering (rare)
  uint32_t k = n < nc ? n : nc;
  for (uint32_t i = 0; i < k; i++) {
    uint32_t x = sbyte(s, at + i), y = c[i];
    if (x != y) return x < y ? -1 : 1;
  }
  return n < nc ? -1 : (n > nc ? 1 : 0);
}

// Rego type rank: null < boolean < number < string < array < object < set
GK_HD int type_rank(uint32_t t) {
  switch (t) {
    case T_NULL: return 0;
    case T_BOOL: return 1;
    case T_INT: case T_FLOAT: return 2;
    case T_STRING: return 3;
    case T_ARRAY: return 4;
    default: return 5;
  }
}

GK_HD bool cmp_test(int c, uint32_t op) {
  switch (op) {
    case C_EQ: return c == 0;
    case C_NE: return c != 0;
    case C_LT: return c < 0;
    case C_LE: return c <= 0;
    case C_GT: return c > 0;
    default: return c >= 0;
  }
}

// three-way compare(row, scalar constant of predicate p). Composite rows only compare by rank (the compiler never
// emits equality between a row and a composite constant).
GK_HD int cmp_row_const(const Row& r, const Pred& p, const StrHdr& h, const uint8_t* heap, const uint8_t* cheap) {
  uint32_t t = row_type(r);
  int ra = type_rank(t), rb = type_rank(p.ctype);
  if (ra != rb) return ra < rb ? -1 : 1;
  switch (t) {
    case T_NULL: return 0;
    case T_BOOL: { int a = (int)r.lo, b = (int)p.k; return a - b; }
    case T_INT:
      if (p.ctype == T_INT) { int64_t a = row_i64(r), b = (int64_t)p.k; return a < b ? -1 : (a > b ? 1 : 0); }
      else { double a = (double)row_i64(r), b = bits_f64(p.k); return a < b ? -1 : (a > b ? 1 : 0); }
    case T_FLOAT: {
      double a = row_f64(r), b = p.ctype == T_INT ? (double)(int64_t)p.k : bits_f64(p.k);
      return a < b ? -1 : (a > b ? 1 : 0);
    }
    case T_STRING: {
      StrRef sr = make_str(r, h, heap);
      if (p.cmp == C_EQ || p.cmp == C_NE) return str_eq_c(sr, cheap + p.a, p.b, p.k) ? 0 : 1;   // equality never needs the ordering
      return str_cmp_c(sr, 0, sr.n, cheap + p.a, p.b);
    }
    default: return 0;
  }
}

// ---- byte-position masks.  Scanning a string byte by byte costs one DEPENDENT memory round trip per byte beyond the 12 header
// bytes (split() on a 35-byte image reference: ~50 of them per predicate, measured as 88 % of the 200-template corpus sweep).
// For strings of up to 64 bytes -- names, images, paths -- the positions of a byte value are ONE 64-bit mask instead: the header
// bytes come from registers, the rest from at most three independent 16-byte loads and one word (a heap entry is 16-byte
// aligned and zero padded: bytes 12.. of the string sit at entry offset 16..), compared four bytes at a time; splitting and
// trimming are bit arithmetic on the masks.  Longer strings take the byte-wise path.
GK_HD uint32_t eq4(uint32_t w, uint32_t pat) {   // bit k = byte k of w equals the pattern byte (exact per byte: no borrow between bytes)
  const uint32_t x = w ^ pat;
  uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
  t >>= 7;
  return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu;
}
GK_HD uint64_t low_mask64(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }
GK_HD uint32_t ctz64(uint64_t v) { return v ? (uint32_t)__builtin_ctzll(v) : 64u; }
struct StrWords { uint32_t w[16]; };   // bytes 0..63 of a string, zero beyond its end
GK_HD StrWords str_words64(const StrRef& s) {
  StrWords o;
  o.w[0] = (uint32_t)s.bits; o.w[1] = (uint32_t)(s.bits >> 32); o.w[2] = s.w2;
  for (int j = 3; j < 16; j++) o.w[j] = 0u;
  if (s.p) {
    // (separate conditions, no use in between: the loads are issued back to back and waited for once)
    if (s.n > 12u) { o.w[3] = ld32(s.p + 12); o.w[4] = ld32(s.p + 16); o.w[5] = ld32(s.p + 20); o.w[6] = ld32(s.p + 24); }
    if (s.n > 28u) { o.w[7] = ld32(s.p + 28); o.w[8] = ld32(s.p + 32); o.w[9] = ld32(s.p + 36); o.w[10] = ld32(s.p + 40); }
    if (s.n > 44u) { o.w[11] = ld32(s.p + 44); o.w[12] = ld32(s.p + 48); o.w[13] = ld32(s.p + 52); o.w[14] = ld32(s.p + 56); }
    if (s.n > 60u) o.w[15] = ld32(s.p + 60);
  }
  return o;
}
GK_HD uint64_t eq_mask64(const StrWords& sw, uint32_t n, uint32_t ch) {   // bit i, i < min(n, 64): byte i equals ch
  const uint32_t pat = (ch & 0xFFu) * 0x01010101u;
  uint64_t m = 0;
  for (int j = 0; j < 16; j++) m |= (uint64_t)eq4(sw.w[j], pat) << (4 * j);
  return m & low_mask64(n);
}
GK_HD uint32_t select_bit64(uint64_t m, uint32_t k) {   // position of the k-th (0-based) set bit; 64 if there is none
  for (uint32_t i = 0; i < k; i++) m &= m - 1ull;
  return ctz64(m);
}
// [lo, hi) of trim(s, cut) from the mask of the positions that hold `cut` (n <= 64)
GK_HD void trim_bounds64(uint64_t cut_mask, uint32_t n, uint32_t* lo, uint32_t* hi) {
  const uint64_t keep = ~cut_mask & low_mask64(n);   // positions that are not the cut byte
  if (!keep) { *lo = n; *hi = n; return; }
  *lo = ctz64(keep);
  *hi = 64u - (uint32_t)__builtin_clzll(keep);
}
// what split(trim(s, cut), sep) looks like: the separator positions inside [lo, hi).  One per (row, cut, sep): every
// predicate on a component of the same split shares it (the plan-specialised build computes it once per class body).
struct SplitMask { uint64_t seps; uint32_t lo, hi; bool fast; };
GK_HD SplitMask split_mask(const StrRef& s, uint8_t cut, uint8_t sep) {
  SplitMask o;
  o.seps = 0; o.lo = 0; o.hi = s.n; o.fast = s.n <= 64u;
  if (!o.fast) return o;
  const StrWords sw = str_words64(s);
  if (cut) trim_bounds64(eq_mask64(sw, s.n, cut), s.n, &o.lo, &o.hi);
  o.seps = eq_mask64(sw, s.n, sep) & low_mask64(o.hi) & ~low_mask64(o.lo);
  return o;
}
GK_HD bool split_component_fast(const SplitMask& sm, int32_t idx, uint32_t* off, uint32_t* len, uint32_t* count) {
  const uint32_t cnt = (uint32_t)__builtin_popcountll(sm.seps) + 1u;
  *count = cnt;
  const int32_t want = idx >= 0 ? idx : (int32_t)cnt + idx;
  if (want < 0 || want >= (int32_t)cnt) return false;
  const uint32_t start = want == 0 ? sm.lo : select_bit64(sm.seps, (uint32_t)want - 1u) + 1u;
  const uint32_t end = (uint32_t)want == cnt - 1u ? sm.hi : select_bit64(sm.seps, (uint32_t)want);
  *off = start; *len = end - start;
  return true;
}

// component `idx` of split(trim(s, cut), sep): returns false when it does not exist.  (byte-wise: strings beyond 64 bytes)
GK_HD_COLD bool split_component_slow(const StrRef& s, uint8_t cut, uint8_t sep, int32_t idx, uint32_t* off, uint32_t* len, uint32_t* count) {
  uint32_t lo = 0, hi = s.n;
  if (cut) {
    while (lo < hi && sbyte(s, lo) == cut) lo++;
    while (hi > lo && sbyte(s, hi - 1) == cut) hi--;
  }
  uint32_t cnt = 1;
  for (uint32_t i = lo; i < hi; i++) cnt += (sbyte(s, i) == sep);
  *count = cnt;
  int32_t want = idx >= 0 ? idx : (int32_t)cnt + idx;
  if (want < 0 || want >= (int32_t)cnt) return false;
  uint32_t start = lo;
  int32_t k = 0;
  for (uint32_t i = lo; i <= hi; i++) {
    if (i == hi || sbyte(s, i) == sep) {
      if (k == want) { *off = start; *len = i - start; return true; }
      k++;
      start = i + 1;
    }
  }
  return false;
}
GK_HD bool split_component(const StrRef& s, const SplitMask& sm, uint8_t cut, uint8_t sep, int32_t idx, uint32_t* off, uint32_t* len, uint32_t* count) {
  if (sm.fast) return split_component_fast(sm, idx, off, len, count);
  return split_component_slow(s, cut, sep, idx, off, len, count);
}
// P_SPLIT_CMP / P_SPLIT_COUNT on a string row whose split is already known
GK_HD bool eval_split_pred(const StrRef& s, const SplitMask& sm, const Pred& p, const uint8_t* cheap) {
  uint32_t off = 0, len = 0, cnt = 0;
  const uint8_t cut = (uint8_t)(p.pad >> 8), sep = (uint8_t)(p.pad & 0xFF);
  const bool have = split_component(s, sm, cut, sep, p.idx, &off, &len, &cnt);
  if (p.op == P_SPLIT_COUNT) { int64_t a = cnt, b = (int64_t)p.k; return cmp_test(a < b ? -1 : (a > b ? 1 : 0), p.cmp); }
  if (!have) return false;
  if (p.cmp == C_EQ || p.cmp == C_NE) { bool eq = len == p.b && str_at_c(s, off, cheap + p.a, len); return (p.cmp == C_EQ) == eq; }
  return cmp_test(str_cmp_c(s, off, len, cheap + p.a, p.b), p.cmp);
}
// P_SPLIT_PREFIX: trim(s, cut) == P  or  trim(s, cut) starts with P + sep      (P = components joined by sep)
GK_HD bool eval_split_prefix(const StrRef& s, const SplitMask& sm, const Pred& p, const uint8_t* cheap) {
  const uint8_t cut = (uint8_t)(p.pad >> 8), sep = (uint8_t)(p.pad & 0xFF);
  uint32_t lo = sm.lo, hi = sm.hi;
  if (!sm.fast && cut) {
    lo = 0; hi = s.n;
    while (lo < hi && sbyte(s, lo) == cut) lo++;
    while (hi > lo && sbyte(s, hi - 1) == cut) hi--;
  }
  const uint32_t len = hi - lo, m = p.b;
  if (len < m) return false;
  if (!str_at_c(s, lo, cheap + p.a, m)) return false;
  if (len == m) return true;
  return sm.fast ? ((sm.seps >> (lo + m)) & 1ull) != 0 : sbyte(s, lo + m) == sep;
}

GK_HD bool eval_pred(const Row& r, const Pred& p, const StrHdr& h, const uint8_t* heap, const uint8_t* cheap) {
  uint32_t t = row_type(r);
  if (t == T_ABSENT) return p.op == P_PRESENT;   // a carrier row of an element without the member: it exists for the element marker alone (plan.hpp)
  switch (p.op) {
    case P_DEFINED: case P_PRESENT: case P_STORE: return true;
    case P_TRUTHY: return !(t == T_BOOL && r.lo == 0);
    case P_CMP: return cmp_test(cmp_row_const(r, p, h, heap, cheap), p.cmp);
    case P_TYPE: return ((1u << t) & p.ctype) != 0;
    case P_STR_PREFIX: case P_STR_SUFFIX: case P_STR_CONTAINS: {
      if (t != T_STRING) return false;
      uint32_t m = p.b;
      if (m == 0) return true;
      StrRef s = make_str(r, h, heap);
      if (m > s.n) return false;
      const uint8_t* c = cheap + p.a;
      if (p.op == P_STR_PREFIX) return str_prefix_c(s, c, m, p.k);
      if (p.op == P_STR_SUFFIX) return str_at_c(s, s.n - m, c, m);
      if (m <= 8) return str_contains_short(s, c, m);
      bool any = false;
      for (uint32_t i = 0; i + m <= s.n; i++) any = any || str_at_c(s, i, c, m);
      return any;
    }
    case P_STR_IN_SET: {
      if (t != T_STRING) return false;
      // set record in the const heap at p.a (4-aligned): p.b entries {u32 a, u32 b, u32 len}:
      //   len <= 7: (a, b) = packed bytes;  else a = hash32, b = const-heap offset of the bytes
      StrRef s = make_str(r, h, heap);
      const uint8_t* e = cheap + p.a;
      bool hit = false;
      for (uint32_t i = 0; i < p.b; i++, e += 12) {
        uint32_t len = ld32(e + 8);
        if (len != s.n) continue;
        uint32_t ea = ld32(e), eb = ld32(e + 4);
        if (len <= 7) hit = hit || (s.bits == (((uint64_t)eb << 32) | ea));
        else if (ea == s.hash) hit = hit || str_eq_c(s, cheap + eb, len, ea);
      }
      return hit;
    }
    case P_SPLIT_PREFIX: {
      if (t != T_STRING) return false;
      StrRef s = make_str(r, h, heap);
      return eval_split_prefix(s, split_mask(s, (uint8_t)(p.pad >> 8), (uint8_t)(p.pad & 0xFF)), p, cheap);
    }
    case P_SPLIT_CMP: case P_SPLIT_COUNT: {
      if (t != T_STRING) return false;
      StrRef s = make_str(r, h, heap);
      return eval_split_pred(s, split_mask(s, (uint8_t)(p.pad >> 8), (uint8_t)(p.pad & 0xFF)), p, cheap);
    }
    case P_REGEX: {
      // DFA table at cheap + p.a: [u32 n_states][u32 n_classes][u8 class_of_byte[256]][u8 accept[n_states]][u8 next[][]]
      if (t != T_STRING) return false;
      StrRef s = make_str(r, h, heap);
      const uint8_t* d = cheap + p.a;
      const uint32_t ns = ld32(d), nc = ld32(d + 4);
      const uint8_t* cls = d + 8;
      const uint8_t* acc = cls + 256;
      const uint8_t* nxt = acc + ns;
      // four bytes per round: one word of the string (header words come from registers), four INDEPENDENT byte-class lookups
      // (one wait), then the four dependent transitions -- a byte at a time every step waits for three chained loads
      uint32_t st = 0;
      for (uint32_t i = 0; i < s.n; i += 4) {
        const uint32_t k = s.n - i, w = sword(s, i);
        const uint32_t c0 = cls[w & 0xFFu], c1 = cls[(w >> 8) & 0xFFu], c2 = cls[(w >> 16) & 0xFFu], c3 = cls[w >> 24];
        st = nxt[st * nc + c0];
        if (k > 1) st = nxt[st * nc + c1];
        if (k > 2) st = nxt[st * nc + c2];
        if (k > 3) st = nxt[st * nc + c3];
      }
      return acc[st] != 0;
    }
    case P_BITS: return t == T_INT && ((((uint64_t)r.hi << 32) | r.lo) & p.k) != 0;
    case P_COUNT_CMP: {
      int64_t a;
      if (t == T_OBJECT || t == T_ARRAY) a = r.lo;
      else return false;
      int64_t b = (int64_t)p.k;
      return cmp_test(a < b ? -1 : (a > b ? 1 : 0), p.cmp);
    }
    default: return false;
  }
}

// does the predicate read the bytes of a string row (so heap strings need their header)?
GK_HD bool pred_needs_str(const Pred& p) {
  switch (p.op) {
    case P_CMP: return p.ctype == T_STRING;
    case P_STR_PREFIX: case P_STR_SUFFIX: case P_STR_CONTAINS: case P_STR_IN_SET:
    case P_SPLIT_CMP: case P_SPLIT_COUNT: case P_SPLIT_PREFIX: case P_REGEX: return true;
    default: return false;
  }
}

GK_HD uint32_t row_vid(const Row& r) { return r.rev >> ROW_VID_SHIFT; }   // the row's value id (plan.hpp), 0 = none

// Accumulator word index helpers ----------------------------------------------------------------------------
// global bit g lives in word g>>5. Global bit 0 is reserved: ELEMENT OVERFLOW (an ordinal >= scope capacity).
constexpr uint32_t GBIT_OVERFLOW = 0;
// element word layout: word0 = [0] present | [1..19] leaf bits | [31:24] parent ordinal; leaf bits >= 20 spill to word 1+.
// A scope with ONE value slot and at most 8 element bits keeps the slot's value id in bits [23:8] of word0
// (Scope::val_off == GK_VAL_PACKED): a join then reads nothing but the element words its loops hold in registers.
constexpr uint32_t ELEM_W0_BITS = 20;
constexpr uint32_t ELEM_PACK_BITS = 8;     // element bits (present included) a scope may use and still pack its value id
GK_HD uint32_t elem_word_of_bit(uint32_t bit) { return bit < ELEM_W0_BITS ? 0 : 1 + ((bit - ELEM_W0_BITS) >> 5); }
GK_HD uint32_t elem_mask_of_bit(uint32_t bit) { return bit < ELEM_W0_BITS ? (1u << bit) : (1u << ((bit - ELEM_W0_BITS) & 31)); }
// value slots of an element: one word each (the stored row's value id), unless packed into the element word
GK_HD uint32_t val_stride(uint32_t nvals) { return nvals; }
GK_HD bool scope_packed(const Scope& sc) { return sc.val_off == GK_VAL_PACKED; }

// Phase 1 for one row. `Acc` provides or_word(w, mask), max_word(w, v), store_word(w, v) for THIS row's review.
template <class Acc>
GK_HD void eval_row_ent(const Row& r, uint32_t row_index, uint32_t ent, const StrHdr& h, const PlanView& pv, const uint8_t* heap, Acc& acc);
template <class Acc>
GK_HD void eval_row(const Row& r, uint32_t path, uint32_t row_index, const PlanView& pv, const uint8_t* heap, Acc& acc) {
  if (path >= pv.dims.n_paths) return;
  uint32_t ent = pv.ptab[path];
  if (ent == 0) return;
  StrHdr h = {{0, 0, 0, 0}};
  if (row_needs_hdr(r)) h = load_hdr(r, heap);
  eval_row_ent(r, row_index, ent, h, pv, heap, acc);
}
// `ent` = the row's path-table entry (first << 8 | count), already fetched
template <class Acc>
GK_HD void eval_row_ent(const Row& r, uint32_t row_index, uint32_t ent, const StrHdr& h, const PlanView& pv, const uint8_t* heap, Acc& acc) {
  uint32_t first = ent >> 8, cnt = ent & 0xFF;
  for (uint32_t i = 0; i < cnt; i++) {
    const Pred& p = pv.preds[first + i];
    if (!eval_pred(r, p, h, heap, pv.cheap)) continue;
    if (p.dst == D_GLOBAL) {
      acc.or_word(p.bit >> 5, 1u << (p.bit & 31));
      continue;
    }
    const Scope& sc = pv.scopes[p.scope];
    uint32_t ord = row_ordinal(r, p.level);
    if (ord >= sc.cap || (r.meta & ROW_ORD_OVERFLOW)) {
      acc.or_word(GBIT_OVERFLOW >> 5, 1u << (GBIT_OVERFLOW & 31));
      continue;
    }
    uint32_t wpe = sc.wpe;
    if (p.op == P_STORE) {
      // A stored value is compared with another one (Rego `==` between two review values): the slot takes the row's VALUE
      // ID.  A row without one -- a non-empty container (its equality would need a deep comparison), a table flattened
      // before the constraint registered the path -- or a review with more compared values than ids flags the review
      // beyond the engine's limits (reported in too_big, the caller fails closed) -- never guessed.
      const uint32_t vid = row_vid(r);
      if (vid == 0u || vid >= GK_VID_OVERFLOW) {
        acc.or_word(GBIT_OVERFLOW >> 5, 1u << (GBIT_OVERFLOW & 31));
        continue;
      }
      if (p.level >= GK_LEVEL_ROOT) { acc.max_word(sc.count_off, 1u); acc.or_word(sc.word_off, 1u); }   // the root scope's element exists once a value is stored
      if (scope_packed(sc)) acc.or_word(sc.word_off + ord * wpe, vid << ELEM_VID_SHIFT);
      else acc.store_word(sc.val_off + ord * val_stride(sc.nvals) + p.bit, vid);
    } else if (p.op == P_PRESENT) {
      uint32_t parent = p.level > 0 ? row_ordinal(r, p.level - 1) : 0;
      acc.or_word(sc.word_off + ord * wpe, 1u | (parent << 24));
      acc.max_word(sc.count_off, ord + 1);
    } else {
      acc.or_word(sc.word_off + ord * wpe + elem_word_of_bit(p.bit), elem_mask_of_bit(p.bit));
    }
  }
}

// value-slot equality (joins): two stored values are equal iff their value ids are (ids are per distinct Rego value within
// the review, plan.hpp); an empty slot (0) equals nothing
GK_HD bool vid_eq(uint32_t a, uint32_t b) { return (a == b) & (a != 0u); }

struct Results {
  uint64_t viol[GK_VIOL_WORDS];   // bit (s & 63) of word (s >> 6): violation formula s
  uint64_t match, err;
  GK_HD bool viol_bit(uint32_t s) const { return (viol[s >> 6] >> (s & 63u)) & 1ull; }
};

// Phase 2 for one review. `bounds[s]` = loop trip count for scope s (any value >= this review's element count;
// the HIP kernel passes the wave-wide maximum so control flow stays uniform).
template <class Acc>
GK_HD Results eval_formulas(const PlanView& pv, Acc& acc, uint32_t flags, const Row* rows, const uint8_t* heap, const uint32_t* bounds) {
  uint64_t B = 0;
  Results res = {};
  uint32_t cur[GK_MAX_SCOPES];
  uint32_t loop_pc[8];
  uint32_t loop_scope[8];
  int depth = 0;
  const uint32_t* code = pv.code;
  uint32_t pc = 0;
  for (;;) {
    uint32_t ins = GK_UNI(code[pc++]);
#ifdef GK_COUNT_OPS
    gk_op_counter++;
#endif
    uint32_t op = ins & 0xFF, a = (ins >> 8) & 0xFF, b = (ins >> 16) & 0xFF, c = ins >> 24;
    switch (op) {
      case F_LDG: {
        uint32_t bit = b | (c << 8);
        uint64_t v = (acc.load(bit >> 5) >> (bit & 31)) & 1;
        B = (B & ~(1ull << a)) | (v << a);
        break;
      }
      case F_LDF: {
        uint64_t v = (flags >> b) & 1;
        B = (B & ~(1ull << a)) | (v << a);
        break;
      }
      case F_LDE: {
        const Scope& sc = pv.scopes[b];
        uint32_t wpe = sc.wpe;
        uint32_t w = sc.word_off + cur[b] * wpe + elem_word_of_bit(c);
        uint64_t v = (acc.load(w) & elem_mask_of_bit(c)) ? 1 : 0;
        B = (B & ~(1ull << a)) | (v << a);
        break;
      }
      case F_AND: { uint64_t v = (B >> b) & (B >> c) & 1; B = (B & ~(1ull << a)) | (v << a); break; }
      case F_OR: { uint64_t v = ((B >> b) | (B >> c)) & 1; B = (B & ~(1ull << a)) | (v << a); break; }
      case F_NOT: { uint64_t v = (~(B >> b)) & 1; B = (B & ~(1ull << a)) | (v << a); break; }
      case F_ANDN: { uint64_t v = (B >> b) & ~(B >> c) & 1; B = (B & ~(1ull << a)) | (v << a); break; }
      case F_CONST: B = (B & ~(1ull << a)) | ((uint64_t)(b & 1) << a); break;
      case F_MOV: { uint64_t v = (B >> b) & 1; B = (B & ~(1ull << a)) | (v << a); break; }
      case F_LOOP: {
        // a = scope, b = parent scope + 1, c = accumulator register (cleared here)
        B &= ~(1ull << c);
        if (GK_UNI(bounds[a]) == 0) {
          // skip to the matching ENDLOOP
          int nest = 1;
          while (nest) {
            uint32_t w = GK_UNI(code[pc++]);
            uint32_t o = w & 0xFF;
            if (o == F_LOOP) nest++;
            else if (o == F_ENDLOOP || o == F_ENDLOOP2) nest--;
            else if (o == F_VEQ) pc++;
          }
          break;
        }
        cur[a] = 0;
        loop_pc[depth] = pc;
        loop_scope[depth] = a | (b << 8);
        depth++;
        break;
      }
      case F_ENDLOOP: {
        uint32_t s = loop_scope[depth - 1] & 0xFF, par = loop_scope[depth - 1] >> 8;
        const Scope& sc = pv.scopes[s];
        uint32_t wpe = sc.wpe;
        uint32_t w0 = acc.load(sc.word_off + cur[s] * wpe);
        bool valid = (w0 & 1) != 0;
        if (par) valid = valid && ((w0 >> 24) == cur[par - 1]);
        uint64_t v = valid ? ((B >> b) & 1) : 0;
        B |= v << a;
        cur[s]++;
        if (cur[s] < GK_UNI(bounds[s])) pc = loop_pc[depth - 1];
        else depth--;
        break;
      }
      case F_ENDLOOP2: {   // counting loop: a = once, b = body, c = twice
        uint32_t s = loop_scope[depth - 1] & 0xFF, par = loop_scope[depth - 1] >> 8;
        const Scope& sc = pv.scopes[s];
        uint32_t w0 = acc.load(sc.word_off + cur[s] * sc.wpe);
        bool valid = (w0 & 1) != 0;
        if (par) valid = valid && ((w0 >> 24) == cur[par - 1]);
        uint64_t v = valid ? ((B >> b) & 1) : 0;
        B |= (v & (B >> a) & 1) << c;
        B |= v << a;
        cur[s]++;
        if (cur[s] < GK_UNI(bounds[s])) pc = loop_pc[depth - 1];
        else depth--;
        break;
      }
      case F_VEQ: {
        uint32_t x = GK_UNI(code[pc++]);
        uint32_t sa = x & 0xFF, la = (x >> 8) & 0xFF, sb = (x >> 16) & 0xFF, lb = x >> 24;
        const Scope& A = pv.scopes[sa];
        const Scope& Bs = pv.scopes[sb];
        const uint32_t ia = scope_packed(A) ? (acc.load(A.word_off + cur[sa] * A.wpe) >> ELEM_VID_SHIFT) & GK_VID_OVERFLOW
                                            : acc.load(A.val_off + cur[sa] * val_stride(A.nvals) + la);
        const uint32_t ib = scope_packed(Bs) ? (acc.load(Bs.word_off + cur[sb] * Bs.wpe) >> ELEM_VID_SHIFT) & GK_VID_OVERFLOW
                                             : acc.load(Bs.val_off + cur[sb] * val_stride(Bs.nvals) + lb);
        uint64_t v = vid_eq(ia, ib) ? 1 : 0;
        B = (B & ~(1ull << a)) | (v << a);
        break;
      }
      case F_STE: {   // derived element bit: bit c of the current element of scope b := reg a
        const Scope& sc = pv.scopes[b];
        if ((B >> a) & 1) acc.or_word(sc.word_off + cur[b] * sc.wpe + elem_word_of_bit(c), elem_mask_of_bit(c));
        break;
      }
      case F_STG: {   // derived global bit (b | c<<8) := reg a
        uint32_t bit = b | (c << 8);
        if ((B >> a) & 1) acc.or_word(bit >> 5, 1u << (bit & 31));
        break;
      }
      case F_RES: {
        uint64_t v = (B >> a) & 1;
        if (b == 0) res.viol[c >> 6] |= v << (c & 63u);
        else if (b == 1) res.match |= v << c;
        else res.err |= v << c;
        break;
      }
      default: return res;   // F_END
    }
  }
}

}  // namespace gk
