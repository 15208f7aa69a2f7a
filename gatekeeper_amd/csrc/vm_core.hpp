// Host/device evaluation core: predicate evaluation for one row (phase 1) and the formula interpreter for one
// review (phase 2).  Compiled by hipcc into kernels.hip (the product path) and by g++ into the TEST-ONLY CPU
// emulator tests/native/hostemu.cpp, which exists so the compiler + flattener can be checked against the oracle
// in the GPU-less build container.  The product library never links the emulator.
#pragma once
#include "plan.hpp"

#if defined(__HIPCC__)
#define GK_HD __host__ __device__ inline
#else
#define GK_HD inline
#endif
// The formula interpreter's control flow is wave-uniform by construction (same bytecode, same loop bounds for all 64
// lanes).  GK_UNI makes that visible to the compiler so the program counter, the decoded instruction and the loop
// counters live in SGPRs and the dispatch is scalar branching instead of exec-mask divergence.
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
GK_HD uint32_t row_ordinal(const Row& r, uint32_t level) { return (r.meta >> (ROW_E_SHIFT0 + 8 * level)) & ROW_E_MASK; }
// Strings live in 4-byte aligned, zero-padded heap entries [u32 len][bytes][pad]: everything below uses aligned
// 32-bit loads and accumulates differences without data-dependent early exits, so the loads of one predicate pipeline.
GK_HD uint32_t ld32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }
GK_HD uint32_t heap_len(const uint8_t* heap, uint32_t off) { return ld32(heap + off - 4); }
GK_HD int64_t row_i64(const Row& r) { return (int64_t)(((uint64_t)r.hi << 32) | r.lo); }
GK_HD double bits_f64(uint64_t b) {
  union { uint64_t u; double d; } x;
  x.u = b;
  return x.d;
}
GK_HD double row_f64(const Row& r) { return bits_f64(((uint64_t)r.hi << 32) | r.lo); }

GK_HD int bytes_cmp(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {   // ordering compares only (rare)
  uint32_t n = na < nb ? na : nb;
  for (uint32_t i = 0; i < n; i++) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}
// n bytes at two 4-aligned, zero-padded addresses are equal (whole strings of equal length)
GK_HD bool words_eq(const uint8_t* a, const uint8_t* b, uint32_t n) {
  uint32_t d = 0;
  for (uint32_t j = 0; j < n; j += 4) d |= ld32(a + j) ^ ld32(b + j);
  return d == 0;
}
// first m bytes at 4-aligned s equal the 4-aligned constant c
GK_HD bool prefix_eq(const uint8_t* s, const uint8_t* c, uint32_t m) {
  uint32_t d = 0, full = m & ~3u;
  for (uint32_t j = 0; j < full; j += 4) d |= ld32(s + j) ^ ld32(c + j);
  uint32_t r = m & 3u;
  if (r) d |= (ld32(s + full) ^ ld32(c + full)) & ((1u << (8 * r)) - 1u);
  return d == 0;
}
// m bytes at an arbitrary (unaligned) s equal the 4-aligned constant c
GK_HD bool bytes_eq(const uint8_t* s, const uint8_t* c, uint32_t m) {
  uint32_t d = 0;
  for (uint32_t i = 0; i < m; i++) d |= (uint32_t)(s[i] ^ c[i]);
  return d == 0;
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
GK_HD int cmp_row_const(const Row& r, const Pred& p, const uint8_t* heap, const uint8_t* cheap) {
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
      if (p.cmp == C_EQ || p.cmp == C_NE) {   // equality never needs the ordering: hash is a fast reject, bytes decide
        if (r.hi != (uint32_t)p.k) return 1;
        uint32_t n = heap_len(heap, r.lo);
        return (n == p.b && words_eq(heap + r.lo, cheap + p.a, n)) ? 0 : 1;
      }
      return bytes_cmp(heap + r.lo, heap_len(heap, r.lo), cheap + p.a, p.b);
    }
    default: return 0;
  }
}

// component `idx` of split(trim(s, cut), sep): returns false when it does not exist.
GK_HD bool split_component(const uint8_t* s, uint32_t n, uint8_t cut, uint8_t sep, int32_t idx, uint32_t* off, uint32_t* len,
                           uint32_t* count) {
  uint32_t lo = 0, hi = n;
  if (cut) {
    while (lo < hi && s[lo] == cut) lo++;
    while (hi > lo && s[hi - 1] == cut) hi--;
  }
  uint32_t cnt = 1;
  for (uint32_t i = lo; i < hi; i++) cnt += (s[i] == sep);
  *count = cnt;
  int32_t want = idx >= 0 ? idx : (int32_t)cnt + idx;
  if (want < 0 || want >= (int32_t)cnt) return false;
  uint32_t start = lo;
  int32_t k = 0;
  for (uint32_t i = lo; i <= hi; i++) {
    if (i == hi || s[i] == sep) {
      if (k == want) { *off = start; *len = i - start; return true; }
      k++;
      start = i + 1;
    }
  }
  return false;
}

GK_HD bool eval_pred(const Row& r, const Pred& p, const uint8_t* heap, const uint8_t* cheap) {
  uint32_t t = row_type(r);
  switch (p.op) {
    case P_DEFINED: case P_PRESENT: case P_STORE: return true;
    case P_TRUTHY: return !(t == T_BOOL && r.lo == 0);
    case P_CMP: return cmp_test(cmp_row_const(r, p, heap, cheap), p.cmp);
    case P_TYPE: return ((1u << t) & p.ctype) != 0;
    case P_STR_PREFIX: case P_STR_SUFFIX: case P_STR_CONTAINS: {
      if (t != T_STRING) return false;
      uint32_t m = p.b;
      if (m == 0) return true;
      uint32_t n = heap_len(heap, r.lo);
      if (m > n) return false;
      const uint8_t* s = heap + r.lo;
      const uint8_t* c = cheap + p.a;
      if (p.op == P_STR_PREFIX) return prefix_eq(s, c, m);
      if (p.op == P_STR_SUFFIX) return bytes_eq(s + (n - m), c, m);
      bool any = false;
      for (uint32_t i = 0; i + m <= n; i++) any = any || bytes_eq(s + i, c, m);
      return any;
    }
    case P_STR_IN_SET: {
      if (t != T_STRING) return false;
      // set record in the const heap at p.a (4-aligned): p.b entries of {u32 hash, u32 off, u32 len}
      const uint8_t* e = cheap + p.a;
      bool hit = false;
      for (uint32_t i = 0; i < p.b; i++, e += 12) {
        if (ld32(e) != r.hi) continue;
        uint32_t n = heap_len(heap, r.lo);
        if (ld32(e + 8) == n && words_eq(heap + r.lo, cheap + ld32(e + 4), n)) hit = true;
      }
      return hit;
    }
    case P_SPLIT_PREFIX: {
      // trim(s, cut) == P  or  trim(s, cut) starts with P + sep      (P = components joined by sep)
      if (t != T_STRING) return false;
      uint32_t n = heap_len(heap, r.lo);
      const uint8_t* s = heap + r.lo;
      uint8_t cut = (uint8_t)(p.pad >> 8), sep = (uint8_t)(p.pad & 0xFF);
      uint32_t lo = 0, hi = n;
      if (cut) {
        while (lo < hi && s[lo] == cut) lo++;
        while (hi > lo && s[hi - 1] == cut) hi--;
      }
      uint32_t len = hi - lo, m = p.b;
      if (len < m) return false;
      bool pre = (lo & 3u) == 0 ? prefix_eq(s + lo, cheap + p.a, m) : bytes_eq(s + lo, cheap + p.a, m);
      if (!pre) return false;
      return len == m || s[lo + m] == sep;
    }
    case P_SPLIT_CMP: case P_SPLIT_COUNT: {
      if (t != T_STRING) return false;
      uint32_t n = heap_len(heap, r.lo), off = 0, len = 0, cnt = 0;
      uint8_t cut = (uint8_t)(p.pad >> 8), sep = (uint8_t)(p.pad & 0xFF);
      bool have = split_component(heap + r.lo, n, cut, sep, p.idx, &off, &len, &cnt);
      if (p.op == P_SPLIT_COUNT) { int64_t a = cnt, b = (int64_t)p.k; return cmp_test(a < b ? -1 : (a > b ? 1 : 0), p.cmp); }
      if (!have) return false;
      return cmp_test(bytes_cmp(heap + r.lo + off, len, cheap + p.a, p.b), p.cmp);
    }
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

// Accumulator word index helpers ----------------------------------------------------------------------------
// global bit g lives in word g>>5. Global bit 0 is reserved: ELEMENT OVERFLOW (an ordinal >= scope capacity).
constexpr uint32_t GBIT_OVERFLOW = 0;
// element word layout: word0 = [0] present | [1..23] leaf bits | [31:24] parent ordinal; bits >= 24 spill to word 1+
GK_HD uint32_t elem_word_of_bit(uint32_t bit) { return bit < 24 ? 0 : 1 + ((bit - 24) >> 5); }
GK_HD uint32_t elem_mask_of_bit(uint32_t bit) { return bit < 24 ? (1u << bit) : (1u << ((bit - 24) & 31)); }

// Phase 1 for one row. `Acc` provides or_word(w, mask), max_word(w, v), store_word(w, v) for THIS row's review.
template <class Acc>
GK_HD void eval_row_ent(const Row& r, uint32_t row_index, uint32_t ent, const PlanView& pv, const uint8_t* heap, Acc& acc);
template <class Acc>
GK_HD void eval_row(const Row& r, uint32_t row_index, const PlanView& pv, const uint8_t* heap, Acc& acc) {
  if (r.path >= pv.dims.n_paths) return;
  uint32_t ent = pv.ptab[r.path];
  if (ent == 0) return;
  eval_row_ent(r, row_index, ent, pv, heap, acc);
}
// `ent` = the row's path-table entry (first << 8 | count), already fetched
template <class Acc>
GK_HD void eval_row_ent(const Row& r, uint32_t row_index, uint32_t ent, const PlanView& pv, const uint8_t* heap, Acc& acc) {
  uint32_t first = ent >> 8, cnt = ent & 0xFF;
  for (uint32_t i = 0; i < cnt; i++) {
    const Pred& p = pv.preds[first + i];
    if (!eval_pred(r, p, heap, pv.cheap)) continue;
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
      acc.store_word(sc.val_off + ord * sc.nvals + p.bit, row_index + 1u);   // value slot = row index + 1 (0 = empty)
    } else if (p.op == P_PRESENT) {
      uint32_t parent = p.level > 0 ? row_ordinal(r, p.level - 1) : 0;
      acc.or_word(sc.word_off + ord * wpe, 1u | (parent << 24));
      acc.max_word(sc.count_off, ord + 1);
    } else {
      acc.or_word(sc.word_off + ord * wpe + elem_word_of_bit(p.bit), elem_mask_of_bit(p.bit));
    }
  }
}

// value-slot equality (joins): slots hold row index + 1. Both rows must hold values of the same Rego type with
// equal content (strings: hash fast-reject, then bytes).
GK_HD bool val_eq(uint32_t sa, uint32_t sb, const Row* rows, const uint8_t* heap) {
  if (sa == 0 || sb == 0) return false;
  const Row a = rows[sa - 1], b = rows[sb - 1];
  uint32_t ta = row_type(a), tb = row_type(b);
  if (type_rank(ta) != type_rank(tb)) return false;
  switch (ta) {
    case T_NULL: return true;
    case T_BOOL: return a.lo == b.lo;
    case T_INT: case T_FLOAT: {
      if (ta == T_INT && tb == T_INT) return a.lo == b.lo && a.hi == b.hi;
      double x = ta == T_INT ? (double)row_i64(a) : row_f64(a);
      double y = tb == T_INT ? (double)row_i64(b) : row_f64(b);
      return x == y;
    }
    case T_STRING: {
      if (a.hi != b.hi) return false;
      uint32_t na = heap_len(heap, a.lo), nb = heap_len(heap, b.lo);
      return na == nb && (a.lo == b.lo || words_eq(heap + a.lo, heap + b.lo, na));
    }
    default: return false;   // composite joins are rejected by the compiler
  }
}

struct Results {
  uint64_t viol, match, err;
};

// Phase 2 for one review. `bounds[s]` = loop trip count for scope s (any value >= this review's element count;
// the HIP kernel passes the wave-wide maximum so control flow stays uniform).
template <class Acc>
GK_HD Results eval_formulas(const PlanView& pv, Acc& acc, uint32_t flags, const Row* rows, const uint8_t* heap, const uint32_t* bounds) {
  uint64_t B = 0;
  Results res = {0, 0, 0};
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
            else if (o == F_ENDLOOP) nest--;
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
      case F_VEQ: {
        uint32_t x = GK_UNI(code[pc++]);
        uint32_t sa = x & 0xFF, la = (x >> 8) & 0xFF, sb = (x >> 16) & 0xFF, lb = x >> 24;
        const Scope& A = pv.scopes[sa];
        const Scope& Bs = pv.scopes[sb];
        uint32_t wa = A.val_off + cur[sa] * A.nvals + la;
        uint32_t wb = Bs.val_off + cur[sb] * Bs.nvals + lb;
        uint64_t v = val_eq(acc.load(wa), acc.load(wb), rows, heap) ? 1 : 0;
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
        if (b == 0) res.viol |= v << c;
        else if (b == 1) res.match |= v << c;
        else res.err |= v << c;
        break;
      }
      default: return res;   // F_END
    }
  }
}

}  // namespace gk
