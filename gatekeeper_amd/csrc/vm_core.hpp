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

namespace gk {

struct PlanView {
  const uint32_t* ptab;        // [n_paths] (first << 8 | count) into pred_list; 0 = no predicates
  const uint32_t* pred_list;   // predicate indices
  const Pred* preds;
  const Scope* scopes;
  const uint32_t* code;
  const uint8_t* cheap;        // constant heap
  PlanDims dims;
};

GK_HD uint32_t row_type(const Row& r) { return r.meta & ROW_TYPE_MASK; }
GK_HD uint32_t row_ordinal(const Row& r, uint32_t level) { return (r.meta >> (ROW_E_SHIFT0 + 8 * level)) & ROW_E_MASK; }
GK_HD uint32_t heap_len(const uint8_t* heap, uint32_t off) {
  return (uint32_t)heap[off - 4] | ((uint32_t)heap[off - 3] << 8) | ((uint32_t)heap[off - 2] << 16) | ((uint32_t)heap[off - 1] << 24);
}
GK_HD int64_t row_i64(const Row& r) { return (int64_t)(((uint64_t)r.hi << 32) | r.lo); }
GK_HD double bits_f64(uint64_t b) {
  union { uint64_t u; double d; } x;
  x.u = b;
  return x.d;
}
GK_HD double row_f64(const Row& r) { return bits_f64(((uint64_t)r.hi << 32) | r.lo); }

GK_HD int bytes_cmp(const uint8_t* a, uint32_t na, const uint8_t* b, uint32_t nb) {
  uint32_t n = na < nb ? na : nb;
  for (uint32_t i = 0; i < n; i++) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return na < nb ? -1 : (na > nb ? 1 : 0);
}
GK_HD bool bytes_eq(const uint8_t* a, const uint8_t* b, uint32_t n) {
  for (uint32_t i = 0; i < n; i++)
    if (a[i] != b[i]) return false;
  return true;
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
      uint32_t n = heap_len(heap, r.lo);
      if (r.hi == (uint32_t)p.k && n == p.b && bytes_eq(heap + r.lo, cheap + p.a, n)) return 0;
      return bytes_cmp(heap + r.lo, n, cheap + p.a, p.b);
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
      uint32_t n = heap_len(heap, r.lo), m = p.b;
      if (m > n) return false;
      const uint8_t* s = heap + r.lo;
      const uint8_t* c = cheap + p.a;
      if (p.op == P_STR_PREFIX) return bytes_eq(s, c, m);
      if (p.op == P_STR_SUFFIX) return bytes_eq(s + (n - m), c, m);
      for (uint32_t i = 0; i + m <= n; i++)
        if (bytes_eq(s + i, c, m)) return true;
      return false;
    }
    case P_STR_IN_SET: {
      if (t != T_STRING) return false;
      // set record in const heap at p.a: p.b entries of {u32 hash, u32 off, u32 len}
      uint32_t n = heap_len(heap, r.lo);
      const uint8_t* e = cheap + p.a;
      for (uint32_t i = 0; i < p.b; i++, e += 12) {
        uint32_t h = (uint32_t)e[0] | ((uint32_t)e[1] << 8) | ((uint32_t)e[2] << 16) | ((uint32_t)e[3] << 24);
        if (h != r.hi) continue;
        uint32_t off = (uint32_t)e[4] | ((uint32_t)e[5] << 8) | ((uint32_t)e[6] << 16) | ((uint32_t)e[7] << 24);
        uint32_t len = (uint32_t)e[8] | ((uint32_t)e[9] << 8) | ((uint32_t)e[10] << 16) | ((uint32_t)e[11] << 24);
        if (len == n && bytes_eq(heap + r.lo, cheap + off, n)) return true;
      }
      return false;
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
GK_HD void eval_row(const Row& r, const PlanView& pv, const uint8_t* heap, Acc& acc) {
  if (r.path >= pv.dims.n_paths) return;
  uint32_t ent = pv.ptab[r.path];
  if (ent == 0) return;
  uint32_t first = ent >> 8, cnt = ent & 0xFF;
  for (uint32_t i = 0; i < cnt; i++) {
    const Pred& p = pv.preds[pv.pred_list[first + i]];
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
      uint32_t w = sc.val_off + (ord * sc.nvals + p.bit) * 3;
      acc.store_word(w, r.lo);
      acc.store_word(w + 1, r.hi);
      acc.store_word(w + 2, 0x80u | row_type(r));
    } else if (p.op == P_PRESENT) {
      uint32_t parent = p.level > 0 ? row_ordinal(r, p.level - 1) : 0;
      acc.or_word(sc.word_off + ord * wpe, 1u | (parent << 24));
      acc.max_word(sc.count_off, ord + 1);
    } else {
      acc.or_word(sc.word_off + ord * wpe + elem_word_of_bit(p.bit), elem_mask_of_bit(p.bit));
    }
  }
}

// value-slot equality (joins). Both slots must hold a value of the same Rego type with equal content.
template <class Acc>
GK_HD bool val_eq(const Acc& acc, uint32_t wa, uint32_t wb, const uint8_t* heap) {
  uint32_t ta = acc.load(wa + 2), tb = acc.load(wb + 2);
  if (!(ta & 0x80) || !(tb & 0x80)) return false;
  ta &= 7; tb &= 7;
  uint32_t alo = acc.load(wa), ahi = acc.load(wa + 1), blo = acc.load(wb), bhi = acc.load(wb + 1);
  if (type_rank(ta) != type_rank(tb)) return false;
  switch (ta) {
    case T_NULL: return true;
    case T_BOOL: return alo == blo;
    case T_INT: case T_FLOAT: {
      if (ta == T_INT && tb == T_INT) return alo == blo && ahi == bhi;
      double x = ta == T_INT ? (double)(int64_t)(((uint64_t)ahi << 32) | alo) : bits_f64(((uint64_t)ahi << 32) | alo);
      double y = tb == T_INT ? (double)(int64_t)(((uint64_t)bhi << 32) | blo) : bits_f64(((uint64_t)bhi << 32) | blo);
      return x == y;
    }
    case T_STRING: {
      if (ahi != bhi) return false;
      uint32_t na = heap_len(heap, alo), nb = heap_len(heap, blo);
      return na == nb && (alo == blo || bytes_eq(heap + alo, heap + blo, na));
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
GK_HD Results eval_formulas(const PlanView& pv, const Acc& acc, uint32_t flags, const uint8_t* heap, const uint32_t* bounds) {
  uint64_t B = 0;
  Results res = {0, 0, 0};
  uint32_t cur[GK_MAX_SCOPES];
  uint32_t loop_pc[8];
  uint32_t loop_scope[8];
  int depth = 0;
  const uint32_t* code = pv.code;
  uint32_t pc = 0;
  for (;;) {
    uint32_t ins = code[pc++];
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
        if (bounds[a] == 0) {
          // skip to the matching ENDLOOP
          int nest = 1;
          while (nest) {
            uint32_t w = code[pc++];
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
        if (cur[s] < bounds[s]) pc = loop_pc[depth - 1];
        else depth--;
        break;
      }
      case F_VEQ: {
        uint32_t x = code[pc++];
        uint32_t sa = x & 0xFF, la = (x >> 8) & 0xFF, sb = (x >> 16) & 0xFF, lb = x >> 24;
        const Scope& A = pv.scopes[sa];
        const Scope& Bs = pv.scopes[sb];
        uint32_t wa = A.val_off + (cur[sa] * A.nvals + la) * 3;
        uint32_t wb = Bs.val_off + (cur[sb] * Bs.nvals + lb) * 3;
        uint64_t v = val_eq(acc, wa, wb, heap) ? 1 : 0;
        B = (B & ~(1ull << a)) | (v << a);
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
