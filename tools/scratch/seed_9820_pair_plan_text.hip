typedef unsigned char uint8_t; typedef unsigned short uint16_t; typedef unsigned int uint32_t; typedef unsigned long long uint64_t;
typedef short int16_t; typedef int int32_t; typedef long long int64_t;
// Flattened-review table layout and the compiled predicate plan, shared by host (compiler, flattener) and device
// (kernels.hip).  Everything here is POD with fixed-width fields.
//
// This is the MI355X engine's replacement for what the reference holds as Go structs + OPA ASTs on the hot path:
//   pkg/target/matcher.go:73-93 re-unmarshals object/oldObject per (constraint, review); here a review is flattened
//   ONCE into 16-byte rows that every constraint program reads, and match blocks (pkg/mutation/match/match.go:32-65)
//   plus template Rego are compiled into the same predicate/formula plan.

namespace gk {

// ------------------------------------------------------------------------------------------------ rows
// One row per JSON node (scalars AND containers) of the review documents.  The table is stored as ROW GROUPS: reviews
// are grouped in tiles of `rpt` consecutive reviews (64 .. 512, per table), and within a tile the rows are sorted by key path (stable:
// review order, then document order).  The rows of one (tile, path) pair form a SEGMENT.  A plan touches only the
// segments of the paths it has predicates on -- typically a fifth of a Pod's rows -- and every row of a segment takes
// the same predicates, so a wave evaluates them without divergence.
//
// Table arrays (HostTable / DevTable):
//   rows[n_rows]                 16 B each, see Row
//   shdr[n_rows]                 16 B each, parallel to rows: for heap strings the entry header [u32 len][first 12
//                                bytes], so string predicates get their operand with the same index as the row (no
//                                dependent heap access unless the string is longer than 12 bytes); zero otherwise
//   tile_idx[n_tiles][S + 1]     first row of slot s in tile t; slots = the table's distinct key paths in path-id order
//                                (= row order inside a tile), so slot s of tile t is [idx[t][s], idx[t][s+1])
//   rflags[n_reviews]            RF_*: match-layer facts computed once by the flattener
//   heap                         string bytes, 16-byte aligned zero-padded entries [u32 len][bytes]
struct Row {
  uint32_t rev;    // [8:0] review index within its row group (0 .. rpt-1) | [31:9] VALUE ID of the row (see ROW_VID_* below), 0 = none
  uint32_t meta;   // see ROW_* below
  uint32_t lo;     // value payload
  uint32_t hi;
};
static_assert(sizeof(Row) == 16, "Row must be 16 bytes");

struct StrHdr { uint32_t w[4]; };   // [len][bytes 0..11] of a heap string
static_assert(sizeof(StrHdr) == 16, "StrHdr must be 16 bytes");

// One plan path bound to a table: which slot holds its rows and which predicate list (path-table entry) they take.
struct Bind {
  uint32_t slot;
  uint32_t ent;
};

// Chunk lists (chunks.hpp): what a plan reads of a table, per row group, fixed when the plan is bound to the table.  A
// group's list = the 64-row chunks of the segments of the plan's paths, in the order the group's waves take them;
// entry 0 is the header.  Every group owns `capg` consecutive entries (header + chunks + unused tail).
struct ChunkDesc {
  uint32_t st;     // first row of the chunk                       | header: number of chunks of the group
  uint32_t info;   // (rows - 1) | entry << GK_DESC_ENT_SHIFT       | header: GK_LIST_OVERFLOW
};
static_assert(sizeof(ChunkDesc) == 8, "ChunkDesc must be 8 bytes");
constexpr uint32_t GK_DESC_ENT_SHIFT = 6;
constexpr uint32_t GK_DESC_ENT_MASK = 0x01FFFFFFu;   // of info >> GK_DESC_ENT_SHIFT: path-table entry (first << 8 | count) or class id
constexpr uint32_t GK_DESC_NEEDS_STR = 1u << 25;     // of info >> GK_DESC_ENT_SHIFT: some predicate of the class reads string bytes
constexpr uint32_t GK_DESC_NULL = 0xFFFFFFFFu;      // info of a padding entry (run-dealt lists, chunks.hpp): nothing to load, nothing to evaluate
constexpr uint32_t GK_LIST_OVERFLOW = 1u;            // header: the group has more chunks than a list holds -> its reviews take the big path

// VALUE IDS.  Rows that the loaded constraints compare with OTHER review values (Rego `==` between two review values: joins
// between array elements, object vs oldObject ...) carry an id that is unique per distinct Rego value WITHIN THEIR REVIEW:
// the flattener interns the review's compared values (numbers by numeric value -- 1 == 1.0 --, strings by bytes, null / true
// / false / empty array / empty object by kind), so that equality on the device is ONE integer compare, exact, with no
// payload, type or heap access.  0 = the row carries no id (its path is not compared, or it is a non-empty container, whose
// equality would need a deep comparison): a predicate that wants one flags the review beyond the engine's limits.
// GK_VID_OVERFLOW = the review holds more distinct compared values than ids: same treatment.
constexpr uint32_t ROW_REV_MASK = 0x1FFu;        // GK_RPT_MAX = 512 reviews per group
constexpr uint32_t ROW_VID_SHIFT = 9;
constexpr uint32_t GK_VID_BITS = 16;             // ids fit the element word (Scope::val_off == GK_VAL_PACKED)
constexpr uint32_t GK_VID_OVERFLOW = (1u << GK_VID_BITS) - 1u;
constexpr uint32_t GK_VID_NULL = 1, GK_VID_FALSE = 2, GK_VID_TRUE = 3, GK_VID_EMPTY_ARRAY = 4, GK_VID_EMPTY_OBJECT = 5, GK_VID_FIRST = 6;

enum RowType : uint32_t { T_NULL = 0, T_BOOL = 1, T_INT = 2, T_FLOAT = 3, T_STRING = 4, T_OBJECT = 5, T_ARRAY = 6,
                          // ELEMENT CARRIERS (round 6).  Half of the rows a sweep read were pairs: an array element's own row (the element
                          // marker: presence, parent ordinal, count) and the row of its `name` member (a value id, a test).  The plans now
                          // hang the marker on the rows of ONE member of the element -- the carrier, registered per element pattern
                          // (flatten.hpp DictRegistry::add_carrier) -- and the flattener guarantees exactly one row at the carrier's path per
                          // element: the member's own row, or, for an element without the member (or one that is no object), a row of THIS
                          // type, which exists for the marker alone: every other predicate treats it as "no row" (vm_core.hpp eval_pred).
                          T_ABSENT = 7 };

constexpr uint32_t ROW_TYPE_MASK = 0x7;
constexpr uint32_t ROW_RESERVED3 = 1u << 3;    // unused
constexpr uint32_t ROW_E_SHIFT0 = 4;           // ordinal of the enclosing element at array-nesting level 0
constexpr uint32_t ROW_E_SHIFT1 = 12;          // ... level 1
constexpr uint32_t ROW_E_SHIFT2 = 20;          // ... level 2
constexpr uint32_t ROW_E_MASK = 0xFF;
constexpr uint32_t ROW_ORD_OVERFLOW = 1u << 28;  // an enclosing ordinal did not fit in 8 bits (saturated at 255)
constexpr uint32_t ROW_DEEP = 1u << 29;          // more than 3 enclosing arrays
constexpr uint32_t ROW_INEXACT = 1u << 30;       // number not exactly representable (bigint / lossy float)
constexpr uint32_t ROW_STR_INLINE = 1u << 31;    // string of <= 7 bytes packed into lo/hi (no heap entry)
// value payload:  bool: lo=0/1 | int: hi:lo = int64 | float: hi:lo = f64 bits
//                 string (<= 7 bytes, ROW_STR_INLINE): lo = bytes 0..3, hi = bytes 4..6 | len << 24
//                 string (longer): lo = byte offset in the table heap of a 16-byte aligned entry [u32 len][bytes][pad]
//                                  (so off-4 is 16-byte aligned), hi = hash32(bytes)
//                 object/array: lo = member count

enum ReviewFlag : uint32_t {
  RF_HAS_OBJ = 1u << 0,          // request.object present (after setObjectOnDelete, pkg/target/target.go:269-287)
  RF_HAS_OLD = 1u << 1,          // request.oldObject present
  RF_NS_PRESENT = 1u << 2,       // Matchable.Namespace != nil (review namespace or nsCache hit, matcher.go:37-39)
  RF_OBJ_IS_NS = 1u << 3,        // match.IsNamespace(object)  (match.go:255-258)
  RF_OLD_IS_NS = 1u << 4,
  RF_OBJ_HAS_NSFIELD = 1u << 5,  // object.metadata.namespace != ""
  RF_OLD_HAS_NSFIELD = 1u << 6,
  RF_SRC_ORIGINAL = 1u << 7,     // Matchable.Source (mutator.go:14-26); neither bit set => ""
  RF_SRC_GENERATED = 1u << 8,
  RF_SRC_INVALID = 1u << 9,      // non-empty source outside {All,Original,Generated}
  RF_OBJ_HAS_NSNAME = 1u << 10,  // an effective namespace name exists for object (match.go:150-179 switch)
  RF_OLD_HAS_NSNAME = 1u << 11,
  RF_TOO_BIG = 1u << 12,         // some array of the review has > 255 elements (informational: the kernels decide per ROW --
                                 // ROW_ORD_OVERFLOW on a row an element predicate reads -- whether a review is beyond the engine's limits)
  RF_SRC_ALL = 1u << 13,
  RF_OBJ_LABELS_BAD = 1u << 14,  // metadata.labels is not a string map: unstructured GetLabels() yields none
  RF_OLD_LABELS_BAD = 1u << 15,
  RF_NS_LABELS_BAD = 1u << 16,
  RF_OBJ_BAD = 1u << 17,         // request.object is a JSON object that Unstructured.UnmarshalJSON rejects (no `kind`):
  RF_OLD_BAD = 1u << 18,
  RF_REFUSE = 1u << 20,          // a non-empty OBJECT sits where the loaded constraints iterate array elements (flatten.hpp,
                                 //   DictRegistry guards): reported in too_big, never evaluated
  RF_HOST_CAND = 1u << 21,       // a compared value of the review has no value id (a non-empty container, or more distinct values than ids):
                                 //   if a predicate wants it the review ends up in too_big -- the engine keeps such a review's text and
                                 //   evaluates it on the host then (engine.cpp complete_on_host); no kernel reads the bit
  RF_PREMATCHED = 1u << 22,      // the CALLER ran Matcher.Match (Client.Review, pkg/target/matcher.go:21-42) and asks for the violation sets of
                                 //   the constraints it hands over (Driver.Query's contract, pkg/drivers/k8scel/driver.go:162-251): every match
                                 //   formula counts as true for this review and no autoreject bit is written (kernel_body.inc output stage)
  RF_SKIP = 1u << 19,            // the review is not evaluated: HandleReview rejected it, or the process excluder skips its
                                 //   namespace (engine.cpp) -- no violation, match or autoreject bit for any constraint         //   gkReviewToObject fails with ErrRequestObject (pkg/target/matcher.go:73-93)
};

// ------------------------------------------------------------------------------------------------ predicates
// Phase 1: every row whose path has predicates evaluates them and ORs result bits into per-review accumulators.
enum PredOp : uint32_t {
  P_DEFINED = 1,     // row exists
  P_TRUTHY = 2,      // row exists and is not `false`
  P_CMP = 3,         // compare(row, const) <op> 0 under Rego's total order
  P_TYPE = 4,        // (1<<type) & mask
  P_STR_PREFIX = 5,  // string row startswith const
  P_STR_SUFFIX = 6,
  P_STR_CONTAINS = 7,
  P_STR_IN_SET = 8,  // string row is a member of a const string set
  P_SPLIT_CMP = 9,   // component idx of split(trim(row, cut), sep)  <op> const string
  P_SPLIT_COUNT = 10,  // count(split(trim(row, cut), sep)) <op> const int
  P_STORE = 11,      // store row value into an element value slot (joins)
  P_COUNT_CMP = 12,  // member count of container / byte length of string <op> const int
  P_PRESENT = 13,    // element marker: sets bit 0 and parent ordinal of the element word
  P_SPLIT_PREFIX = 14,  // split(trim(row, cut), sep) starts with the constant component list (fused path-prefix test)
  P_REGEX = 15,      // string row matches a constant regular expression (unanchored search): byte-class DFA in the const heap
  P_BITS = 16,       // integer row (a <leaf>.$d dictionary row, dexpr.hpp) has one of the bits of the mask k set
};
enum CmpOp : uint32_t { C_EQ = 0, C_NE = 1, C_LT = 2, C_LE = 3, C_GT = 4, C_GE = 5 };
enum PredDst : uint32_t { D_GLOBAL = 0, D_ELEM = 1 };
// (host side only) comparison codes beyond CmpOp for tests on the member NAME of a key iteration, resolved against the
// table's key paths when a plan is built: startswith / endswith / contains(key, const), "the key is a member name"
constexpr int KC_PREFIX = 6, KC_SUFFIX = 7, KC_CONTAINS = 8, KC_ISNAME = 9;
// Pred::level of a value stored for the ROOT scope: the one-element scope that holds review values compared with each other
// outside any iteration (object.spec.x != oldObject.spec.x).  Its "element" 0 exists as soon as one of its values is stored.
constexpr uint32_t GK_LEVEL_ROOT = 3;

struct Pred {
  uint8_t op;       // PredOp
  uint8_t dst;      // PredDst
  uint8_t scope;    // D_ELEM: scope index
  uint8_t level;    // D_ELEM: which ordinal of the row addresses the element (0..2)
  uint16_t bit;     // D_GLOBAL: bit index in the global bitset; D_ELEM: bit in the element word (P_STORE: value slot)
  uint8_t cmp;      // CmpOp for P_CMP / P_SPLIT_* / P_COUNT_CMP
  uint8_t ctype;    // RowType of the constant (P_CMP); type mask (P_TYPE)
  uint32_t a;       // const-heap byte offset (string / set)
  uint32_t b;       // const length / set size
  uint64_t k;       // immediate: int64 / f64 bits / string constant key (packed bytes if len <= 7, else hash32)
  int32_t idx;      // P_SPLIT_CMP component index (negative = from the end)
  uint32_t pad;     // P_SPLIT_*: (cut << 8) | sep
};
static_assert(sizeof(Pred) == 32, "Pred must be 32 bytes");

struct Scope {
  uint32_t word_off;   // first accumulator word (per review) of this scope's element words
  uint32_t val_off;    // first accumulator word of value slots; GK_VAL_PACKED: the scope's single slot lives in the element word
  uint32_t count_off;  // accumulator word holding max ordinal + 1
  uint16_t cap;        // element capacity in this variant
  uint8_t nvals;       // value slots per element: one word each, holding the stored row's VALUE ID (0 = empty)
  uint8_t wpe;         // accumulator words per element
};
constexpr uint32_t GK_VAL_PACKED = 0xFFFFFFFFu;   // one value slot, <= 8 element bits: the id sits in bits [23:8] of element word 0
constexpr uint32_t ELEM_VID_SHIFT = 8;

// ------------------------------------------------------------------------------------------------ formulas
// Phase 2: one lane per review runs this wave-uniform bytecode over the accumulators. 64 boolean registers.
enum FOp : uint32_t {
  F_LDG = 1,    // a = global bit (b | c<<8)
  F_LDF = 2,    // a = review flag bit b
  F_LDE = 3,    // a = bit c of the current element of scope b
  F_AND = 4,    // a = b & c
  F_OR = 5,     // a = b | c
  F_NOT = 6,    // a = !b
  F_ANDN = 7,   // a = b & !c
  F_CONST = 8,  // a = b
  F_MOV = 9,    // a = b
  F_LOOP = 10,  // begin loop over elements of scope a; b = parent scope + 1 (restrict to children of its current elem)
  F_ENDLOOP = 11,  // a = accumulator reg, b = body result reg:  a |= b & valid(elem); next element
  F_VEQ = 12,   // a = (value slot == value slot); followed by one extra word scopeA | slotA<<8 | scopeB<<16 | slotB<<24
  F_RES = 13,   // result[b (0 viol, 1 match, 2 error)][c] = reg a
  F_END = 14,
  F_STE = 15,   // derived element bit: bit c of the current element of scope b |= reg a  (common-subformula cache)
  F_STG = 16,   // derived global bit (b | c<<8) |= reg a
  F_ENDLOOP2 = 17,  // counting loop end: a = "once" reg (the loop's accumulator), b = body reg, c = "twice" reg:
                    //   twice |= once & b & valid(elem);  once |= b & valid(elem);  next element
};
inline constexpr uint32_t finst(uint32_t op, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0) {
  return op | (a << 8) | (b << 16) | (c << 24);
}

struct ConstraintSlot {
  uint16_t viol;    // index into the violation-result bits
  uint16_t match;   // index into the match-result bits (and match-error bits)
};

constexpr int GK_TILE = 64;            // reviews per bitmap word = lanes of a wave (one lane per review in phase 2)
// Reviews per ROW GROUP ("tile") are a property of each table, fixed when it is flattened (HostTable::rpt): 64 for small
// batches (admission), 256 / 512 for resident sets -- one workgroup of the dominant kernel per group, see kernel_body.inc.
constexpr int GK_RPT_MIN = 64;
constexpr int GK_RPT_MAX = 512;
constexpr int GK_PARTS_MIN_RPT = 4;    // formula shares per 64-review half in the 64-review geometry (256 threads)
inline constexpr int gk_block_of(int rpt) { return rpt <= 128 ? 256 : rpt * 2; }          // threads per row group
inline constexpr int gk_parts_of(int rpt) { return gk_block_of(rpt) / GK_TILE / (rpt / GK_TILE); }   // formula shares per half
constexpr int GK_MAX_RES = 64;         // distinct MATCH formulas (and their error formulas) per plan: one 64-bit result word per review
// distinct VIOLATION formulas per plan (round 6): GK_VIOL_WORDS banks of 64 result slots -- a policy set of a few hundred templates is ONE
// plan and one walk of the table (the 200-template corpus: 102 violation formulas, 6 match formulas), where rounds 1-5 cut it into
// groups of <= 64 constraints that each walked the table
constexpr int GK_VIOL_WORDS = 4;
constexpr int GK_MAX_VIOL = 64 * GK_VIOL_WORDS;
constexpr int GK_MAX_SCOPES = 32;
constexpr int GK_WAVE_CHUNKS = 64;      // 64-row chunks one wave queues per tile (LDS); beyond: the tile's reviews take the big path
constexpr uint32_t GK_ENT_NEEDS_STR = 0x80000000u;   // class entry flag (plan-specialised build): some predicate reads string bytes

struct PlanDims {
  uint32_t n_paths;       // entries in ptab
  uint32_t n_preds;
  uint32_t n_scopes;
  uint32_t n_code;        // formula words
  uint32_t n_constraints;
  uint32_t n_gwords;      // global bitset words
  uint32_t acc_words;     // accumulator words per review (globals + scopes)
  uint32_t const_bytes;
  uint32_t n_viol, n_match;   // result slots in use (distinct violation / match formulas)
};

}  // namespace gk
// Host/device evaluation core: predicate evaluation for one row (phase 1) and the formula interpreter for one
// review (phase 2).  Compiled by hipcc into kernels.hip (the product path) and by g++ into the TEST-ONLY CPU
// emulator tests/native/hostemu.cpp, which exists so the compiler + flattener can be checked against the oracle
// in the GPU-less build container.  The product library never links the emulator.

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
#define GK_LANE_ID() __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u))
static __device__ inline void gk_writelane2(unsigned long long m, const uint32_t l, uint32_t& lo, uint32_t& hi) {
  asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(l)); }
#define GK_WRITELANE2(m, l, lo, hi) gk_writelane2((m), (l), (lo), (hi))
#define GK_RES_BASE(kind) ((kind) == 0 ? 0u : (kind) == 1 ? (uint32_t)GK_RES_KV : (kind) == 2 ? (uint32_t)(GK_RES_KV + GK_RES_KM) : ((uint32_t)(kind) - 2u) * 64u)
#define GK_RES_PROLOGUE uint32_t gk_rl0 = 0u, gk_rh0 = 0u, gk_rl1 = 0u, gk_rh1 = 0u, gk_rl2 = 0u, gk_rh2 = 0u, gk_rl3 = 0u, gk_rh3 = 0u, gk_rl4 = 0u, gk_rh4 = 0u, gk_rl5 = 0u, gk_rh5 = 0u;
#define GK_RES(kind, slot, b) do { const unsigned long long m_ = __ballot((b) != 0u); GK_WRITELANE2(m_, slot, gk_rl##kind, gk_rh##kind); } while (0)
#define GK_RES_FLUSH(m0, m1, m2, m3, m4, m5) do { const uint32_t l_ = GK_LANE_ID() & 63u; if (((unsigned long long)(m0) >> l_) & 1ull) masks[GK_RES_BASE(0) + l_] = ((unsigned long long)gk_rh0 << 32) | gk_rl0; if (((unsigned long long)(m1) >> l_) & 1ull) masks[GK_RES_BASE(1) + l_] = ((unsigned long long)gk_rh1 << 32) | gk_rl1; if (((unsigned long long)(m2) >> l_) & 1ull) masks[GK_RES_BASE(2) + l_] = ((unsigned long long)gk_rh2 << 32) | gk_rl2; if (((unsigned long long)(m3) >> l_) & 1ull) masks[GK_RES_BASE(3) + l_] = ((unsigned long long)gk_rh3 << 32) | gk_rl3; if (((unsigned long long)(m4) >> l_) & 1ull) masks[GK_RES_BASE(4) + l_] = ((unsigned long long)gk_rh4 << 32) | gk_rl4; if (((unsigned long long)(m5) >> l_) & 1ull) masks[GK_RES_BASE(5) + l_] = ((unsigned long long)gk_rh5 << 32) | gk_rl5; (void)gk_rl0; (void)gk_rh0; (void)gk_rl1; (void)gk_rh1; (void)gk_rl2; (void)gk_rh2; (void)gk_rl3; (void)gk_rh3; (void)gk_rl4; (void)gk_rh4; (void)gk_rl5; (void)gk_rh5; } while (0)
namespace gk {
GK_CONST_ARRAY unsigned char gk_plan_consts[32] = {121,121,0,0,0,0,0,0,0,0,0,0,0,0,0,0,120,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
#define GK_HAS_ZERO_RANGES 1
constexpr uint32_t GK_N_ZERO_RANGES = 1u;
GK_CONST_ARRAY uint32_t gk_zero_lo[1] = {0u};
GK_CONST_ARRAY uint32_t gk_zero_hi[1] = {19u};
#define GK_RES_KV 4
#define GK_RES_KM 4
#define GK_N_SCOPES_K 2
GK_CONST_ARRAY uint32_t gk_count_off[2] = {1u,10u};
GK_CONST_ARRAY uint32_t gk_scope_cap[2] = {8u,8u};
template <class Acc>
GK_HD __attribute__((always_inline)) void jit_row(Row r, uint32_t cls, StrHdr h, const uint8_t* heap, Acc acc, bool on) {
  const uint8_t* cheap = gk_plan_consts;
  (void)cheap; (void)h;
  cls = GK_UNI(cls) & ~GK_ENT_NEEDS_STR;   // one class per call: the dispatch is a scalar branch
  switch (cls) {
    case 1: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      if (t != 7u) {
      uint32_t mg0 = 0u;
      mg0 |= 64u;
      if (mg0) acc.or_word(0u, mg0);
      }
    }
    } while (false);
    break;
    case 2: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      if (t != 7u) {
      uint32_t me0_0_0 = 0u;
      me0_0_0 |= 16u;
      { constexpr Pred P = Pred{3,1,0,0,5,0,1,0u,0u,1ull,0,0u}; if (eval_pred(r, P, h, heap, cheap)) me0_0_0 |= 32u; }
      if (((r.meta & (7u | ROW_STR_INLINE)) == (4u | ROW_STR_INLINE) && r.lo == 31097u && r.hi == 33554432u)) me0_0_0 |= 64u;
      if (true) {
        const uint32_t ord = row_ordinal(r, 0u);
        if (ord >= 8u || (r.meta & ROW_ORD_OVERFLOW)) acc.or_word(0u, 1u);
        else {
          acc.or_word(2u + ord * 1u, me0_0_0);
        }
      }
      }
    }
    } while (false);
    break;
    case 3: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      if (t != 7u) {
      uint32_t me1_0_0 = 0u;
      if (((r.meta & (7u | ROW_STR_INLINE)) == (4u | ROW_STR_INLINE) && r.lo == 120u && r.hi == 16777216u)) me1_0_0 |= 2u;
      if ((me1_0_0) != 0u) {
        const uint32_t ord = row_ordinal(r, 0u);
        if (ord >= 8u || (r.meta & ROW_ORD_OVERFLOW)) acc.or_word(0u, 1u);
        else {
          acc.or_word(11u + ord * 1u, me1_0_0);
        }
      }
      }
    }
    } while (false);
    break;
    case 4: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      const bool real = t != 7u; (void)real;
      uint32_t me0_0_0 = 0u;
      if (real) me0_0_0 |= 2u;
      { constexpr Pred P = Pred{3,1,0,0,2,0,1,0u,0u,1ull,0,0u}; if (eval_pred(r, P, h, heap, cheap)) me0_0_0 |= 4u; }
      if (real && (((r.meta & (7u | ROW_STR_INLINE)) == (4u | ROW_STR_INLINE) && r.lo == 31097u && r.hi == 33554432u))) me0_0_0 |= 8u;
      if (true) {
        const uint32_t ord = row_ordinal(r, 0u);
        if (ord >= 8u || (r.meta & ROW_ORD_OVERFLOW)) acc.or_word(0u, 1u);
        else {
          acc.max_word(1u, ord + 1u);
          acc.or_word(2u + ord * 1u, me0_0_0 | 1u);
        }
      }
    }
    } while (false);
    break;
    case 5: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      const bool real = t != 7u; (void)real;
      uint32_t me1_0_0 = 0u;
      if (real) me1_0_0 |= 4u;
      if (true) {
        const uint32_t ord = row_ordinal(r, 0u);
        if (ord >= 8u || (r.meta & ROW_ORD_OVERFLOW)) acc.or_word(0u, 1u);
        else {
          acc.max_word(10u, ord + 1u);
          acc.or_word(11u + ord * 1u, me1_0_0 | 1u);
        }
      }
    }
    } while (false);
    break;
    case 6: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      if (t != 7u) {
      uint32_t mg0 = 0u;
      if ((t == T_INT) && (((r.lo & 1u) | (r.hi & 0u)) != 0u)) mg0 |= 16u;
      if ((t == T_INT) && (((r.lo & 2u) | (r.hi & 0u)) != 0u)) mg0 |= 32u;
      if (mg0) acc.or_word(0u, mg0);
      }
    }
    } while (false);
    break;
    case 7: do { if (on) {
      const uint32_t t = r.meta & 7u; (void)t;
      if (t != 7u) {
      uint32_t mg0 = 0u;
      if ((t == T_INT) && (((r.lo & 1u) | (r.hi & 0u)) != 0u)) mg0 |= 128u;
      if (mg0) acc.or_word(0u, mg0);
      }
    }
    } while (false);
    break;
    default: break;
  }

}

template <class Acc>
GK_HD Results jit_formulas(const PlanView& pv, Acc& acc, uint32_t flags, const Row* rows, const uint8_t* heap, const uint32_t* bounds) {
  (void)pv; (void)rows; (void)heap; (void)flags;
  Results res = {};
  uint32_t b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u, b4 = 0u, b5 = 0u, b6 = 0u, b7 = 0u, b8 = 0u, b9 = 0u, b10 = 0u, b11 = 0u, b12 = 0u, b13 = 0u, b14 = 0u, b15 = 0u, b16 = 0u, b17 = 0u, b18 = 0u, b19 = 0u, b20 = 0u, b21 = 0u, b22 = 0u, b23 = 0u, b24 = 0u, b25 = 0u, b26 = 0u, b27 = 0u, b28 = 0u, b29 = 0u, b30 = 0u, b31 = 0u, b32 = 0u, b33 = 0u, b34 = 0u, b35 = 0u, b36 = 0u, b37 = 0u, b38 = 0u, b39 = 0u, b40 = 0u, b41 = 0u, b42 = 0u, b43 = 0u, b44 = 0u, b45 = 0u, b46 = 0u, b47 = 0u, b48 = 0u, b49 = 0u, b50 = 0u, b51 = 0u, b52 = 0u, b53 = 0u, b54 = 0u, b55 = 0u, b56 = 0u, b57 = 0u, b58 = 0u, b59 = 0u, b60 = 0u, b61 = 0u, b62 = 0u, b63 = 0u;
  uint32_t g0 = acc.load(0u);
  b0 = 0u;
  { uint32_t t_ = 0u;
    const uint32_t nq_ = GK_UNI(bounds[0]);
    for (uint32_t eq_ = 0; eq_ < nq_; eq_++) { const uint32_t wq_ = acc.load(2u + eq_ * 1u); t_ |= (uint32_t)((wq_ & 15u) == 3u); }
    b0 = t_; }
  if (b0) g0 |= 2u;
  b1 = 0u;
  { uint32_t t_ = 0u;
    const uint32_t nq_ = GK_UNI(bounds[1]);
    for (uint32_t eq_ = 0; eq_ < nq_; eq_++) { const uint32_t wq_ = acc.load(11u + eq_ * 1u); t_ |= (uint32_t)((wq_ & 7u) == 7u); }
    b1 = t_; }
  if (b1) g0 |= 4u;
  b2 = 0u;
  { uint32_t t_ = 0u;
    const uint32_t nq_ = GK_UNI(bounds[0]);
    for (uint32_t eq_ = 0; eq_ < nq_; eq_++) { const uint32_t wq_ = acc.load(2u + eq_ * 1u); t_ |= (uint32_t)((wq_ & 113u) == 17u); }
    b2 = t_; }
  if (b2) g0 |= 8u;
  b0 = (g0 >> 1) & 1u;
  b0 = b0 ^ 1u;
  b1 = (g0 >> 2) & 1u;
  b2 = (g0 >> 1) & 1u;
  b1 = b1 & b2;
  b2 = (g0 >> 3) & 1u;
  b2 = b2 ^ 1u;
  b1 = b1 & b2;
  b0 = b0 | b1;
  res.viol[0] |= (uint64_t)b0 << 0;
  b0 = (g0 >> 4) & 1u;
  b1 = (g0 >> 5) & 1u;
  b1 = b1 ^ 1u;
  b2 = (g0 >> 6) & 1u;
  b1 = b1 & b2;
  b2 = (g0 >> 7) & 1u;
  b1 = b1 & b2;
  b0 = b0 | b1;
  res.viol[0] |= (uint64_t)b0 << 1;
  b0 = 1u;
  res.match |= (uint64_t)b0 << 0;
  b0 = 0u;
  res.err |= (uint64_t)b0 << 0;
  return res;
}

#define GK_HAS_STAGES 1
constexpr uint32_t GK_N_STAGES = 2u;
constexpr uint32_t GK_GEN_PARTS = 4u;
#ifndef GK_RES
#define GK_RES(kind, slot, b) do { if ((kind) == 0) res.viol[0] |= (uint64_t)(b) << (slot); else if ((kind) == 1) res.match |= (uint64_t)(b) << (slot); else if ((kind) == 2) res.err |= (uint64_t)(b) << (slot); else res.viol[(kind) - 2] |= (uint64_t)(b) << (slot); } while (0)
#define GK_RES_PROLOGUE
#endif
#ifndef GK_RES_FLUSH
#define GK_RES_FLUSH(m0, m1, m2, m3, m4, m5)
#endif
template <class Acc>
GK_HD void jit_formula_part(uint32_t part, Acc& acc, uint32_t flags, const uint8_t* heap, const uint32_t* bounds, Results& res, unsigned long long* masks) {
  (void)heap; (void)flags; (void)bounds; (void)res; (void)masks;
  GK_RES_PROLOGUE
  uint32_t b0 = 0u, b1 = 0u, b2 = 0u, b3 = 0u, b4 = 0u, b5 = 0u, b6 = 0u, b7 = 0u, b8 = 0u, b9 = 0u, b10 = 0u, b11 = 0u, b12 = 0u, b13 = 0u, b14 = 0u, b15 = 0u, b16 = 0u, b17 = 0u, b18 = 0u, b19 = 0u, b20 = 0u, b21 = 0u, b22 = 0u, b23 = 0u, b24 = 0u, b25 = 0u, b26 = 0u, b27 = 0u, b28 = 0u, b29 = 0u, b30 = 0u, b31 = 0u, b32 = 0u, b33 = 0u, b34 = 0u, b35 = 0u, b36 = 0u, b37 = 0u, b38 = 0u, b39 = 0u, b40 = 0u, b41 = 0u, b42 = 0u, b43 = 0u, b44 = 0u, b45 = 0u, b46 = 0u, b47 = 0u, b48 = 0u, b49 = 0u, b50 = 0u, b51 = 0u, b52 = 0u, b53 = 0u, b54 = 0u, b55 = 0u, b56 = 0u, b57 = 0u, b58 = 0u, b59 = 0u, b60 = 0u, b61 = 0u, b62 = 0u, b63 = 0u;
  uint32_t g0 = acc.load(0u);
  switch (part) {
    case 0: {
      {
      uint32_t W0_0 = acc.load(2u);
      uint32_t W0_1 = acc.load(3u);
      uint32_t W0_2 = acc.load(4u);
      uint32_t W0_3 = acc.load(5u);
      uint32_t W0_4 = acc.load(6u);
      uint32_t W0_5 = acc.load(7u);
      uint32_t W0_6 = acc.load(8u);
      uint32_t W0_7 = acc.load(9u);
      b0 = 0u;
      { uint32_t t_ = 0u;
        t_ |= (uint32_t)((W0_0 & 15u) == 3u);
        t_ |= (uint32_t)((W0_1 & 15u) == 3u);
        t_ |= (uint32_t)((W0_2 & 15u) == 3u);
        t_ |= (uint32_t)((W0_3 & 15u) == 3u);
        t_ |= (uint32_t)((W0_4 & 15u) == 3u);
        t_ |= (uint32_t)((W0_5 & 15u) == 3u);
        t_ |= (uint32_t)((W0_6 & 15u) == 3u);
        t_ |= (uint32_t)((W0_7 & 15u) == 3u);
        b0 = t_; }
      if (b0) { g0 |= 2u; acc.or_word(0u, 2u); }
      }
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 1: {
      {
      uint32_t W0_0 = acc.load(2u);
      uint32_t W0_1 = acc.load(3u);
      uint32_t W0_2 = acc.load(4u);
      uint32_t W0_3 = acc.load(5u);
      uint32_t W0_4 = acc.load(6u);
      uint32_t W0_5 = acc.load(7u);
      uint32_t W0_6 = acc.load(8u);
      uint32_t W0_7 = acc.load(9u);
      b2 = 0u;
      { uint32_t t_ = 0u;
        t_ |= (uint32_t)((W0_0 & 113u) == 17u);
        t_ |= (uint32_t)((W0_1 & 113u) == 17u);
        t_ |= (uint32_t)((W0_2 & 113u) == 17u);
        t_ |= (uint32_t)((W0_3 & 113u) == 17u);
        t_ |= (uint32_t)((W0_4 & 113u) == 17u);
        t_ |= (uint32_t)((W0_5 & 113u) == 17u);
        t_ |= (uint32_t)((W0_6 & 113u) == 17u);
        t_ |= (uint32_t)((W0_7 & 113u) == 17u);
        b2 = t_; }
      if (b2) { g0 |= 8u; acc.or_word(0u, 8u); }
      }
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 2: {
      {
      uint32_t W1_0 = acc.load(11u);
      uint32_t W1_1 = acc.load(12u);
      uint32_t W1_2 = acc.load(13u);
      uint32_t W1_3 = acc.load(14u);
      uint32_t W1_4 = acc.load(15u);
      uint32_t W1_5 = acc.load(16u);
      uint32_t W1_6 = acc.load(17u);
      uint32_t W1_7 = acc.load(18u);
      b1 = 0u;
      { uint32_t t_ = 0u;
        t_ |= (uint32_t)((W1_0 & 7u) == 7u);
        t_ |= (uint32_t)((W1_1 & 7u) == 7u);
        t_ |= (uint32_t)((W1_2 & 7u) == 7u);
        t_ |= (uint32_t)((W1_3 & 7u) == 7u);
        t_ |= (uint32_t)((W1_4 & 7u) == 7u);
        t_ |= (uint32_t)((W1_5 & 7u) == 7u);
        t_ |= (uint32_t)((W1_6 & 7u) == 7u);
        t_ |= (uint32_t)((W1_7 & 7u) == 7u);
        b1 = t_; }
      if (b1) { g0 |= 4u; acc.or_word(0u, 4u); }
      }
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 3: {
      {
      b0 = (g0 >> 4) & 1u;
      b1 = (g0 >> 5) & 1u;
      b1 = b1 ^ 1u;
      b2 = (g0 >> 6) & 1u;
      b1 = b1 & b2;
      b2 = (g0 >> 7) & 1u;
      b1 = b1 & b2;
      b0 = b0 | b1;
      GK_RES(0, 1, b0);
      b0 = 1u;
      GK_RES(1, 0, b0);
      b0 = 0u;
      GK_RES(2, 0, b0);
      }
      GK_RES_FLUSH(0x2ull, 0x1ull, 0x1ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 4: {
      {
      b0 = (g0 >> 1) & 1u;
      b0 = b0 ^ 1u;
      b1 = (g0 >> 2) & 1u;
      b2 = (g0 >> 1) & 1u;
      b1 = b1 & b2;
      b2 = (g0 >> 3) & 1u;
      b2 = b2 ^ 1u;
      b1 = b1 & b2;
      b0 = b0 | b1;
      GK_RES(0, 0, b0);
      }
      GK_RES_FLUSH(0x1ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 5: {
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 6: {
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    case 7: {
      GK_RES_FLUSH(0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull, 0x0ull);
    } break;
    default: break;
  }
  (void)g0;
}
}  // namespace gk
#define GK_TILES_BOUNDS __launch_bounds__(256, 8)
#define GK_PREFETCH 1
#define GK_PRIO_LEVELS 3003
#define GK_PRIO_PART0 1
#define GK_RPT_K 64
#define GK_RPP_K 64
#define GK_LIST_CAP_K 256
#define GK_TOT_K 64
namespace gk {
#define GK_KERNEL_TILES gk_jit_tiles
#define GK_KERNEL_BIG gk_jit_big
#define GK_KERNEL_LINKAGE extern "C"
#define GK_SKIP_BIG
#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) jit_row(r, ent, h, heap, acc, on)
#define GK_BIND_ALWAYS_STR 0
#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) jit_formulas(pv, acc, flags, rows, heap, bounds)
// Kernel bodies shared by the ahead-of-time build (kernels.hip: generic bytecode interpreter) and the plan-specialised
// build (codegen.cpp -> hiprtc: GK_ROW_FN / GK_FORMULA_FN are functions generated for one plan).  See kernels.hip for
// the description of the two phases.
//
// Geometry (compile time, per instantiation): GK_RPT_K = reviews per ROW GROUP of the table this instantiation serves
// (64, 128, 256 or 512; a table's row-group size is fixed when it is flattened).  One workgroup per row group:
//     GK_RPT_K   threads  waves  halves (64-review bitmap words)  parts (formula shares per half)
//        64        256      4        1                              4
//       128        256      4        2                              2
//       256        512      8        4                              2
//       512       1024     16        8                              2
// Large row groups make the segments of sparse key paths long enough to fill a wave's 64 lanes (a 64-review group has
// ~130 bound segments of ~8 rows each: phase 1 then runs mostly empty lanes); small ones keep admission batches cheap.
#ifndef GK_RPT_K
#define GK_RPT_K 64
#endif
#ifndef GK_BLOCK_K
#define GK_BLOCK_K ((GK_RPT_K) <= 128 ? 256 : (GK_RPT_K) * 2)
#endif
#define GK_HALVES_K ((GK_RPT_K) / 64)
#define GK_PARTS_K (GK_BLOCK_K / 64 / GK_HALVES_K)
#ifndef GK_TILES_BOUNDS
#define GK_TILES_BOUNDS __launch_bounds__(GK_BLOCK_K)
#endif
// result slots kept per 64-review half: GK_RES_KV violation slots, then GK_RES_KM match slots, then GK_RES_KM error slots
// (plan-specialised build: what the plan uses, in steps of four -- codegen.cpp jit_res_kv / jit_res_km; generic build: the limits)
#ifndef GK_RES_KV
#define GK_RES_KV GK_MAX_VIOL
#define GK_RES_KM GK_MAX_RES
#endif
#define GK_RES_WORDS_K ((GK_RES_KV) + 2 * (GK_RES_KM))     // result words per half
#define GK_RES_BANKS_K (((GK_RES_KV) + 63) / 64)           // 64-slot banks of violation slots (1 for up to 64 distinct violation formulas)
#ifndef GK_KERNEL_TILES
#define GK_KERNEL_TILES gk_eval_tiles
#define GK_KERNEL_BIG gk_eval_big
#define GK_KERNEL_LINKAGE
#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) do { if (on) eval_row_ent(r, i, ent, h, pv, heap, acc); } while (0)
#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) eval_formulas(pv, acc, flags, rows, heap, bounds)
#endif

#ifndef GK_BODY_COMMON
#define GK_BODY_COMMON
// ------------------------------------------------------------------------------------------------ accumulators
template <uint32_t STRIDE>
struct LdsAccT {   // review `rl` of the pass; words are [w][STRIDE], STRIDE = reviews per pass (compile time: plan-specialised build)
  uint32_t* base;
  uint32_t rl;
  __device__ void or_word(uint32_t w, uint32_t m) { atomicOr(&base[w * STRIDE + rl], m); }
  __device__ void max_word(uint32_t w, uint32_t v) { atomicMax(&base[w * STRIDE + rl], v); }
  __device__ void store_word(uint32_t w, uint32_t v) { base[w * STRIDE + rl] = v; }
  __device__ uint32_t load(uint32_t w) const { return base[w * STRIDE + rl]; }
};
struct LdsAccR {   // the same with a run-time stride (generic build)
  uint32_t* base;
  uint32_t rl, stride;
  __device__ void or_word(uint32_t w, uint32_t m) { atomicOr(&base[w * stride + rl], m); }
  __device__ void max_word(uint32_t w, uint32_t v) { atomicMax(&base[w * stride + rl], v); }
  __device__ void store_word(uint32_t w, uint32_t v) { base[w * stride + rl] = v; }
  __device__ uint32_t load(uint32_t w) const { return base[w * stride + rl]; }
};
struct GlobalAcc {   // big variant: contiguous words of one review in HBM scratch
  uint32_t* base;
  __device__ void or_word(uint32_t w, uint32_t m) { atomicOr(&base[w], m); }
  __device__ void max_word(uint32_t w, uint32_t v) { atomicMax(&base[w], v); }
  __device__ void store_word(uint32_t w, uint32_t v) { base[w] = v; }
  __device__ uint32_t load(uint32_t w) const { return base[w]; }
};

__device__ inline uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// largest value of the wave's 64 lanes, in every lane
__device__ inline uint32_t gk_wave_max(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = (uint32_t)__shfl_xor((int)v, o); v = x > v ? x : v; }
  return v;
}
// The TEST-ONLY kernel emulator (tests/native/kernel_emu.hpp, g++) replaces what only the device compiler understands
#ifndef GK_DYN_LDS
#define GK_DYN_LDS(name) extern __shared__ uint32_t name[]
#define GK_OPAQUE_V2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define GK_OPAQUE_V1(a) asm volatile("" : "+v"(a))
#define GK_OPAQUE_S1(a) asm volatile("" : "+s"(a))
#define GK_OPAQUE() asm volatile("")
typedef uint32_t gk_u32x4 __attribute__((ext_vector_type(4)));
#endif
#ifndef GK_LDS_ADD   // an add into LDS whose result nobody reads (ds_add_u32 without return); the kernel emulator supplies its own
#define GK_LDS_ADD(p, v) ((void)__hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
#endif

struct OutPtrs {
  uint64_t* viol;
  uint64_t* err;
  uint64_t* match;       // may be null
  uint64_t* overflow;    // [n_tiles] reviews that must be re-run in the big variant
  uint64_t* too_big;     // [n_tiles] reviews beyond engine limits (reported, never guessed)
  uint32_t* counts;      // [n_constraints]
  uint32_t* list;        // pairs
  uint32_t* list_count;  // [0] = entries wanted, [1] = overflowed-review count (this launch's slot, zero on entry)
  uint32_t list_capacity;
  unsigned long long* prof;   // profiling aid (GK_KERNEL_PROF): 8 clock samples per row group, or null
  uint32_t* partial;     // [gridDim.x][GK_TOT_K] popcounts of the violation words each workgroup wrote in this launch (a GK_TOT_K build always gets the buffer)
};
// Profiling aids: the phase marks (GK_KERNEL_PROF) and the phase switches of the launch word (GK_DBG_PHASE).  The plan-specialised build
// carries them only when one of the two is set (jit_source.hpp defines GK_WITH_PROF then): the kernel has no scalars to spare -- 65
// spilled to vector lanes at the 64-VGPR budget -- and the seven marks with their pointer, test and address arithmetic cost
// configs[2] 6 % (0.0465 -> 0.0438 ms) and the 10 M-object table 3 % (0.420 -> 0.406): profiles/r06_variants_ai_*.log.
#if defined(GK_RPP_K) && !defined(GK_WITH_PROF)
#define GK_PROF(k) do { } while (0)
#define GK_DBG(m) 0u
#else
#define GK_PROF(k) do { if (out.prof && tid == 0) out.prof[(size_t)tile * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define GK_DBG(m) (dbg & (m))
#endif
#endif   // GK_BODY_COMMON

// ------------------------------------------------------------------------------------------------ dominant kernel
// PERSISTENT workgroups: the grid holds as many workgroups as fit on the device at once (kernels.hip), each walks the
// row groups  xcd * per_xcd + blockIdx / 8 + i * gridDim / 8  of "its" XCD.  Per row group (and PASS, when the group's
// accumulators do not fit in LDS: every pass walks the same chunks and takes the rows of its rpp reviews):
//   list:     the group's CHUNK LIST (chunks.hpp: the 64-row chunks of the segments of the plan's key paths, heaviest
//             predicate class first) is copied to LDS with one coalesced 8-byte load per thread -- issued one group
//             AHEAD, while the current group is in phase 1; the first row loads of the next group are issued as soon
//             as the current group's formulas are through, so they travel while the outputs are written and the
//             accumulators cleared.  No index lookups, no atomics, no barrier of its own: a group starts with its
//             first rows already on their way.
//   phase 1:  the waves take the chunks round robin, in batches of GK_PREFETCH (rows and, for classes with string
//             predicates, the parallel string headers -- same index, no dependent access).  One chunk = one key path = one
//             predicate class: the dispatch is a SCALAR branch and the predicates run with (nearly) full lanes.  Rows of
//             paths without predicates are never read.  Result bits are OR-ed into the per-review accumulators in LDS,
//             laid out [word][review].
//   phase 2:  lane = review.  Loop bounds per WAVE (the largest element count among the wave's 64 reviews: a wave
//             reduction, no LDS atomics, no barrier); formulas over the accumulators; ballots -> bitmaps.
#ifndef GK_BIND_ALWAYS_STR
#define GK_BIND_ALWAYS_STR 1   // generic build: path-table entries carry no "reads string bytes" flag
#endif
#ifndef GK_PREFETCH
#define GK_PREFETCH 2
#endif
GK_KERNEL_LINKAGE __global__ GK_TILES_BOUNDS void GK_KERNEL_TILES(PlanView pv, const Row* __restrict__ rows,
                                                         const StrHdr* __restrict__ shdr, const ChunkDesc* __restrict__ lists,
                                                         uint32_t capg, const uint32_t* __restrict__ rflags, const uint8_t* __restrict__ heap,
                                                         uint32_t n_reviews, uint32_t n_tiles, const ConstraintSlot* __restrict__ slots,
                                                         OutPtrs out, uint32_t dbg, uint32_t rpp) {
  // reviews per PASS: the plan-specialised build fixes it at compile time (GK_RPP_K: LDS offsets fold into immediates),
  // the generic build takes it from the launch
#ifdef GK_RPP_K
  typedef LdsAccT<GK_RPP_K> LdsAcc;
  constexpr uint32_t RPP = GK_RPP_K;
  (void)rpp;
#define GK_MAKE_ACC(rl) LdsAcc{lds, (rl)}
#else
  typedef LdsAccR LdsAcc;
  const uint32_t RPP = rpp;
#define GK_MAKE_ACC(rl) LdsAcc{lds, (rl), RPP}
#endif
  GK_DYN_LDS(lds);                           // [acc_words][RPP]
  constexpr uint32_t NW = GK_BLOCK_K / GK_TILE;
  // list capacity (header included): 64 entries per wave for up to 8 waves; two buffers -- the list of the NEXT group is
  // staged while the current one is walked
#ifdef GK_LIST_CAP_K
  constexpr uint32_t LIST_CAP = GK_LIST_CAP_K;   // plan-specialised build: what the table's longest list needs (jit_source.hpp jit_list_cap)
#else
  constexpr uint32_t LIST_CAP = (NW < 8u ? NW : 8u) * GK_WAVE_CHUNKS;
#endif
  static_assert(LIST_CAP <= (uint32_t)GK_BLOCK_K, "one list entry per thread");
  __shared__ uint2 s_work[2][LIST_CAP];
  // Results of phase 2: one 64-bit word per result slot and half -- bit l = the formula's value for review l of the
  // half ([half][kind][slot]).  They live in the list buffer of the group just walked (dead after phase 1) when they fit.
  constexpr uint32_t MASK_WORDS = (uint32_t)GK_HALVES_K * (uint32_t)GK_RES_WORDS_K;
  constexpr bool MASKS_ALIAS = MASK_WORDS <= LIST_CAP;
  __shared__ unsigned long long s_masks_own[MASKS_ALIAS ? 1u : MASK_WORDS];
  __shared__ uint32_t s_slw[GK_TILE];
#ifndef GK_N_SCOPES_K
  __shared__ uint32_t s_wb[NW][GK_MAX_SCOPES];   // generic build: the wave's loop bounds (written and read by the same wave)
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & (GK_TILE - 1);
  const uint32_t wave = GK_UNI(tid / GK_TILE);   // wave-uniform by construction: keep it (and everything derived from it) scalar
  // XCD-aware group order: workgroups are dealt round robin to the 8 XCDs, so workgroup b runs on XCD b % 8.  Giving
  // each XCD a contiguous range of groups keeps neighbouring groups -- whose 8-byte bitmap words share cache lines and
  // whose rows are adjacent -- behind the same L2 (the grid is a multiple of 8).
  constexpr uint32_t N_XCD = 8;
  const uint32_t n_groups = (n_reviews + GK_RPT_K - 1u) / GK_RPT_K;   // row groups; n_tiles = bitmap words per row = ceil(n / 64)
  const uint32_t per_xcd = (n_groups + N_XCD - 1u) / N_XCD;
  const uint32_t xcd = blockIdx.x % N_XCD, nbx = gridDim.x / N_XCD;
  uint32_t gl = blockIdx.x / N_XCD;                  // position in the XCD's range
  uint32_t tile = xcd * per_xcd + gl;
#ifdef GK_TOT_K
  // PER-CONSTRAINT TOTALS WITHOUT A SECOND KERNEL PER SWEEP (round 6).  The output stage adds the popcount of every violation word it
  // writes into s_tot (one LDS add without return per lane and batch of 64 constraints: lgkmcnt, not in the way of the rows in flight);
  // a workgroup leaves the launch with ONE row of plain stores, partial[blockIdx.x][0 .. GK_TOT_K) -- no global atomics, no fence, no contention --
  // and the collecting call adds the rows up (kernels.hip gk_sum_partials, once per collection instead of a popcount kernel over every
  // bitmap word behind every sweep: 4.9 us + a dispatch gap behind each 44 us sweep of configs[2]).  A workgroup without a row group
  // clears its row.  GK_TOT_K = the plan's constraints rounded up to 64 (jit_source.hpp jit_tot_k; plans beyond 256 constraints and
  // the bytecode build keep the popcount kernel).
  __shared__ uint32_t s_tot[GK_TOT_K];
  // (Measured, profiles/r06_variants_a{p,q,r,u}_*.log: kernel +1.0 us on configs[2] (0.0432 -> 0.0442 ms), level on configs[1] and the corpus;
  //  per step 0.0494 -> 0.0459 / 0.0135 -> 0.0110 / 0.0420 -> 0.0399 ms; 10 M objects 0.436 -> 0.422.  The array is 256 B of LDS and configs[2]
  //  has exactly that much to spare: 34 816 B of accumulators + 4 608 B static = 39 424 B per group keep FOUR groups resident per CU, eight
  //  bytes more -- a variant that parked the row's address in LDS -- and the fourth runs behind the other three, +25 % on the persistent
  //  grid.  jit_source.hpp jit_tot_k leaves the array out where it would cost a group.)
  if (gl >= per_xcd || tile >= n_groups) {
#pragma unroll
    for (uint32_t q = threadIdx.x; q < (uint32_t)GK_TOT_K; q += GK_BLOCK_K) out.partial[(size_t)blockIdx.x * GK_TOT_K + q] = 0u;   // (a compile-time trip count: one store)
    return;
  }
#pragma unroll
  for (uint32_t q = threadIdx.x; q < (uint32_t)GK_TOT_K; q += GK_BLOCK_K) s_tot[q] = 0u;   // (ordered before the first output stage by the prologue's barrier)
#else
  if (gl >= per_xcd || tile >= n_groups) return;
#endif
  // phase 2 geometry: the wave serves the 64 reviews of one half of the group and evaluates one share of the formulas
  const uint32_t half = wave / GK_PARTS_K, part = wave % GK_PARTS_K;
  const uint32_t acc_words = pv.dims.acc_words;
  const uint32_t n_pass = GK_RPT_K / RPP, halves_per_pass = RPP / GK_TILE;
  const bool wave_on = half < halves_per_pass;             // a multi-pass launch has fewer halves per pass than wave groups
  const uint32_t rl2 = (wave_on ? half : 0u) * GK_TILE + lane;   // "my" review within the pass

  // ---- chunk fetch.  Loads that land in a slot must not be followed by anything that reads the slot (a zero default
  // merged in at the end of a divergent `if`, a copy for a phi): the compiler would have to wait for the load right
  // there and the prefetch would be worth nothing.  So: wave-uniform (scalar) branches only, the row load is
  // unconditional with the lane's index clamped into the chunk (`on` remembers which lanes hold a row), and a slot whose
  // chunk does not need string headers simply keeps the stale ones.
  struct Chunk { gk_u32x4 rv, hv; uint32_t ent, j; bool on; };   // whole vectors, opaque until consumed
  auto fetch = [&](const uint2* wl, uint32_t nch_u, uint32_t j, Chunk& c) {
    c.j = j;
    if (j < nch_u) {
      const uint2 d = wl[1u + j];
      const uint32_t st = GK_UNI(d.x), info = GK_UNI(d.y);
      const uint32_t nr = (info & 63u) + 1u;                  // rows of the chunk
      c.ent = info >> GK_DESC_ENT_SHIFT;
      c.on = lane < nr;
      const uint32_t at = st + (lane < nr ? lane : nr - 1u);
      c.rv = reinterpret_cast<const gk_u32x4*>(rows)[at];
      if (GK_BIND_ALWAYS_STR || (c.ent & GK_DESC_NEEDS_STR)) c.hv = reinterpret_cast<const gk_u32x4*>(shdr)[at];
    } else {
      c.on = false;
    }
  };
  // BATCHES of GK_PREFETCH chunks per wave: the loads of the next batch are issued together and have the processing
  // of the whole current batch to land.  (A ring rotated by moves never has more than one load in flight: the moves
  // read the newest slot, so the compiler must wait for the newest load in every iteration; and its GK_PREFETCH x 8
  // moves per chunk were a third of the loop's vector instructions -- profiles/r02_variants_*.log.)  The slots of a
  // batch are static registers; the single copy of the row code picks its chunk with a wave-uniform branch.
  static_assert(GK_PREFETCH >= 1 && GK_PREFETCH <= 8, "GK_PREFETCH: 1..8 chunks per batch");
  Chunk cur[GK_PREFETCH], nxt[GK_PREFETCH];
#pragma unroll
  for (int k = 0; k < GK_PREFETCH; k++) { nxt[k].rv = gk_u32x4{0u, 0u, 0u, 0u}; nxt[k].hv = gk_u32x4{0u, 0u, 0u, 0u}; nxt[k].ent = 0; nxt[k].on = false; nxt[k].j = 0xFFFFFFFFu; }
  auto list_len = [&](const uint2* wl) -> uint32_t {   // chunks of the staged list (0: the group overflows the list)
    const uint2 hd = wl[0];
    return GK_DBG(16u) || (GK_UNI(hd.y) & GK_LIST_OVERFLOW) ? 0u : GK_UNI(hd.x);
  };

  // the result slots of constraints 0..63 (lane c: constraint c) -- the same for every group: staged in LDS once, so that
  // the output stage of the usual <= 64-constraint plan issues no global load (loads return in order: it would wait for
  // the rows requested just before it; and kept in a register the value gets spilled, which is a load again)
  if (tid < (uint32_t)GK_TILE) s_slw[tid] = tid < pv.dims.n_constraints ? *reinterpret_cast<const uint32_t*>(&slots[tid]) : 0u;
  // the per-constraint totals start at zero: the popcount kernel behind this launch ADDS its slices' counts (kernels.hip
  // gk_count_rows_sliced); the first workgroup always has a row group to walk
  if (blockIdx.x == 0 && out.counts) for (uint32_t q = tid; q < pv.dims.n_constraints; q += GK_BLOCK_K) out.counts[q] = 0u;

  // ---- the walk: ITEMS = (row group, pass); what follows an item: the group's next pass, or the first pass of this
  // workgroup's next group
  struct Item { uint32_t gl, pass; bool ok; };
  auto tile_of = [&](const Item& it) -> uint32_t { return xcd * per_xcd + it.gl; };
  auto next_of = [&](const Item& it) -> Item {
    if (!it.ok) return it;
    const uint32_t nr = min((uint32_t)GK_RPT_K, n_reviews - tile_of(it) * GK_RPT_K);
    if (it.pass + 1u < n_pass && (it.pass + 1u) * RPP < nr) return Item{it.gl, it.pass + 1u, true};
    const uint32_t g2 = it.gl + nbx;
    return Item{g2, 0u, g2 < per_xcd && xcd * per_xcd + g2 < n_groups};
  };
  // (A DYNAMIC group order -- every further group from a per-XCD ticket counter, or one workgroup per group left to the
  //  hardware dispatcher -- was measured three ways in round 3 and is 7-8 % SLOWER than this fixed stride, whether the ticket
  //  is drawn in front of the item's first rows or in phase 2, where no load of the drawing wave is in flight: 0.135-0.140
  //  against 0.126-0.130 ms on configs[2]; profiles/r03_variants_c_*.log, r03_variants_f_*.log.)
  // Loads are issued in this order -- they return in order --: [list entry of the item after next | flags of the next
  // item | first rows of the next item], all as soon as the current item's formulas are through; the first wait (start of
  // the next item's phase 1) then finds everything it covers requested a whole output + clearing stage earlier.
  auto flags_of = [&](const Item& it, uint32_t tid_now) -> uint32_t {
    (void)tid_now;
    const uint32_t t0 = tile_of(it) * GK_RPT_K, nr = min((uint32_t)GK_RPT_K, n_reviews - t0), g = it.pass * RPP + rl2;
    return (it.ok && wave_on && g < nr) ? rflags[t0 + g] : 0u;
  };
  auto entry_of = [&](const Item& it, uint32_t tid_now) -> ChunkDesc {
    ChunkDesc d = {0u, 0u};
    if (it.ok && tid_now < capg) d = lists[(size_t)tile_of(it) * capg + tid_now];
    return d;
  };

  // ---- stagger.  All workgroups of the (persistent) grid start together and take the same time per phase, so without
  // this the whole device walks in step: every CU streams rows at once (HBM saturated), then every CU evaluates formulas
  // (HBM idle) -- the phases ADD UP (profiles/r02_phases_ac_*.log: loading the rows costs the same 0.057 ms whether or not
  // anything else is done, close to the 378 MB / 8 TB/s floor).  Half of the workgroups therefore start late by about half
  // an item (`stagger` x 256 clocks, in the launch word `dbg`): their row phases fall into the others' formula phases.
  {
    const uint32_t stagger = (dbg >> 8) & 0xFFFu;
    const bool late = (dbg & (1u << 20)) ? gl >= (nbx + 1u) / 2u : (gl & 1u) != 0u;
    if (stagger && late) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < (unsigned long long)stagger * 256ull) __builtin_amdgcn_s_sleep(64);
    }
  }

  // ---- prologue: the first group's list, the next one's entry, the first flags and the first batch of rows
  uint32_t buf = 0;
  Item it = Item{gl, 0u, true}, nx = next_of(it);
  if (tid < capg) { const ChunkDesc d = lists[(size_t)tile * capg + tid]; s_work[0][tid] = make_uint2(d.st, d.info); }
  ChunkDesc dn = entry_of(nx, tid);
  uint32_t flags = flags_of(it, tid);   // "my" review's flags in the current item (re-loaded, for the next item, once their last use is behind)
  __syncthreads();
  {
    const uint32_t n0 = list_len(s_work[0]);
#pragma unroll
    for (int k = 0; k < GK_PREFETCH; k++) fetch(s_work[0], n0, wave + (uint32_t)k * NW, nxt[k]);
  }

  // WAVE PRIORITIES (round 6, visits s..y; profiles/r06_timeline_s_*.log, r06_variants_{t,u,v,w,x,y}_*.log).  A CU serves its resident
  // workgroups by wave priority, then AGE: with everything at priority 0 the k-th workgroup the dispatcher placed on a CU took
  // 20 / 23 / 28 / 35 k clocks per row group (k = 0..3, configs[2]), the oldest left early and the youngest finished alone.  What
  // pays on every table size measured (0.6 M .. 10 M objects: -5 .. -7 %; level on the small ones) is a priority per PHASE:
  // GK_PRIO_LEVELS = four decimal digits -- phase 1 (waves that request rows and wait: they issue little, and the sooner the
  // better) | bounds | formulas (the issue-heavy part) | outputs, the next item's requests and the clearing stage.  Default 3003.
  // (A priority per workgroup that rotates round by round -- (k - round) mod K -- balances the finishing times and wins 7-9 % at
  //  3.8 row groups per workgroup, but loses up to 20 % when the last round is thin: tools/scratch/wave_priority_rotation.patch.)
#if defined(GK_PRIO_LEVELS) && GK_PRIO_LEVELS
#define GK_PRIO_AT(where) __builtin_amdgcn_s_setprio((where) == 1 ? (GK_PRIO_LEVELS / 1000) % 10 : (where) == 2 ? (GK_PRIO_LEVELS / 100) % 10 : (where) == 3 ? (GK_PRIO_LEVELS / 10) % 10 : GK_PRIO_LEVELS % 10)
#else
#define GK_PRIO_AT(where) do { } while (0)
#endif
  for (;;) {
    tile = tile_of(it);
    const uint32_t pass = it.pass;
    const uint32_t r0 = tile * GK_RPT_K;
    const uint32_t nrev = min((uint32_t)GK_RPT_K, n_reviews - r0);
    const uint32_t rev_lo = pass * RPP;              // first review (within the group) of this pass
    const bool has_next = nx.ok;
    const uint32_t rg = rev_lo + rl2;                        // "my" review within the group
    const bool live = wave_on && rg < nrev;
    GK_PROF(0);
    // (opaque per item: addresses that are functions of the thread index alone -- the clearing ranges' start offsets, the
    //  thread's list entry -- get hoisted out of the group loop and spilled; the reload, a load, would then wait right here
    //  for the rows requested at the end of the last item)
    uint32_t tid_c = tid;
    GK_OPAQUE_V1(tid_c);
    if (!GK_DBG(128u)) {
      uint4* lds4 = reinterpret_cast<uint4*>(lds);
#ifdef GK_HAS_ZERO_RANGES
      // value-slot payload words are only read behind their type nibble: no need to clear them
      // (RPP is a multiple of 64: every range is a whole number of 16-byte stores)
      for (uint32_t q = 0; q < GK_N_ZERO_RANGES; q++)
        for (uint32_t w = gk_zero_lo[q] * (RPP / 4u) + tid_c; w < gk_zero_hi[q] * (RPP / 4u); w += GK_BLOCK_K) lds4[w] = make_uint4(0u, 0u, 0u, 0u);
#else
      for (uint32_t w = tid_c; w < acc_words * (RPP / 4u); w += GK_BLOCK_K) lds4[w] = make_uint4(0u, 0u, 0u, 0u);
#endif
    }
    __syncthreads();   // accumulators are clear (and the previous item's result words have been read)
    GK_PROF(1);
    const uint2* wl = s_work[buf];
    const uint32_t nch_u = list_len(wl);
    const bool list_ovf = (GK_UNI(wl[0].y) & GK_LIST_OVERFLOW) != 0u;   // absurdly large group: all its reviews take the big path

    // ---- phase 1: wave w walks chunks w, w + NW, ... of the list; its first batch is already in flight
    GK_PRIO_AT(1);
    for (uint32_t jb = wave; jb < nch_u; jb += (uint32_t)GK_PREFETCH * NW) {
#pragma unroll
      for (int k = 0; k < GK_PREFETCH; k++) cur[k] = nxt[k];
#pragma unroll
      for (int k = 0; k < GK_PREFETCH; k++) fetch(wl, nch_u, jb + (uint32_t)(GK_PREFETCH + k) * NW, nxt[k]);
#pragma nounroll
      for (uint32_t t = 0; t < (uint32_t)GK_PREFETCH; t++) {
        // (branches with an opaque statement inside: folded into `cur[t]` the slots would move to scratch memory)
        Chunk c0 = cur[0];
#define GK_PICK(K) else if (K < GK_PREFETCH && t == K##u) { c0 = cur[K < GK_PREFETCH ? K : 0]; GK_OPAQUE(); }
        if (t == 0u) { GK_OPAQUE(); }
        GK_PICK(1) GK_PICK(2) GK_PICK(3) GK_PICK(4) GK_PICK(5) GK_PICK(6) GK_PICK(7)
#undef GK_PICK
        if (c0.j >= nch_u) break;
        // the consumer's uses of single fields must not be hoisted to the loads (the compiler would shuffle the loaded
        // registers right behind the load and wait for it there): the vectors pass through an empty asm first
        GK_OPAQUE_V2(c0.rv, c0.hv);
        Row r; r.rev = c0.rv.x; r.meta = c0.rv.y; r.lo = c0.rv.z; r.hi = c0.rv.w;
        StrHdr h; h.w[0] = c0.hv.x; h.w[1] = c0.hv.y; h.w[2] = c0.hv.z; h.w[3] = c0.hv.w;
        // the rows of this pass's reviews (unsigned compare: also rejects rev < rev_lo); a single pass takes them all
        const uint32_t rrev = r.rev & ROW_REV_MASK;   // (the bits above hold the row's value id, plan.hpp)
#if defined(GK_RPP_K) && GK_RPP_K == GK_RPT_K
        const bool mine_now = c0.on && !GK_DBG(1u);   // (one pass: every row of the group is a row of the pass)
#else
        const bool mine_now = c0.on && (rrev - rev_lo) < RPP && !GK_DBG(1u);
#endif
        // (a lane without a row of this pass addresses the slot of review `lane`: the plan-specialised row code issues its LDS
        //  atomics unconditionally, with neutral operands for such lanes -- spread over the banks, not piled on one address)
        LdsAcc acc = GK_MAKE_ACC(mine_now ? rrev - rev_lo : lane);
        GK_ROW_FN(r, c0.j, c0.ent & GK_DESC_ENT_MASK, h, pv, heap, acc, mine_now);
      }
    }
    GK_PROF(2);
    if (has_next && tid_c < capg) s_work[buf ^ 1u][tid_c] = make_uint2(dn.st, dn.info);
    __syncthreads();   // phase 1 is complete, the next list is staged
    GK_PROF(3);

    // ---- phase 2: lane = review
    GK_PRIO_AT(2);
    LdsAcc acc = GK_MAKE_ACC(rl2);
    // beyond-limits reviews are found per ROW (an element predicate met an ordinal that does not fit, eval_row_ent): they
    // overflow here, are re-run by the big variant and, if they still overflow there, are reported in too_big -- an array
    // of > 255 elements that no element predicate reads does not disturb the review.  (Read now: once the last formula
    // stage is through, the waves without outputs start clearing the accumulators for the next item.)
    const bool ovf = live && ((acc.load(0) & 1u) || list_ovf);
    // loop bound of a scope = the largest element count among the wave's 64 reviews
#ifdef GK_N_SCOPES_K
    uint32_t wbounds[GK_N_SCOPES_K > 0 ? GK_N_SCOPES_K : 1];
    wbounds[0] = 0u;
    // small capacities (the usual case of a table-specialised plan): count the thresholds some review of the wave exceeds --
    // one compare + one scalar test each, independent of each other; a cross-lane max costs six dependent LDS-crossbar permutes
#pragma unroll
    for (uint32_t s = 0; s < (uint32_t)GK_N_SCOPES_K; s++) {
      const uint32_t cnt = wave_on ? lds[gk_count_off[s] * RPP + rl2] : 0u;
      if (gk_scope_cap[s] <= 16u) {
        uint32_t b = 0u;
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++) if (k < gk_scope_cap[s]) b += __ballot(cnt > k) != 0ull ? 1u : 0u;
        wbounds[s] = b;
      } else wbounds[s] = GK_UNI(gk_wave_max(cnt));
    }
#else
    uint32_t* wbounds = s_wb[wave];
    for (uint32_t s = 0; s < pv.dims.n_scopes; s++) {
      const uint32_t m = gk_wave_max(wave_on ? lds[pv.scopes[s].count_off * RPP + rl2] : 0u);
      if (lane == 0) wbounds[s] = m;
    }
#endif
    GK_PROF(4);
    GK_PRIO_AT(3);
#if defined(GK_PRIO_LEVELS) && GK_PRIO_LEVELS && defined(GK_PRIO_PART0)
    if (part == 0u) __builtin_amdgcn_s_setprio(GK_PRIO_PART0);   // (the formula share whose wave goes on to write the violation words: jit_source.hpp)
#endif
    Results res = {};
    // "lane s = words of slot s": every constraint's bitmap words are TWO cross-lane gathers away (lane c fetches the
    // words of its constraint's match / violation slots) -- instead of extracting 3 bits per lane and constraint
    unsigned long long* s_masks = MASKS_ALIAS ? reinterpret_cast<unsigned long long*>(s_work[buf]) : s_masks_own;
    unsigned long long* hm = s_masks + (wave_on ? half : 0u) * (uint32_t)GK_RES_WORDS_K;
#ifdef GK_HAS_STAGES
    // phase 2 on all waves: the formulas are cut into self-contained blocks (codegen.cpp); every wave evaluates its
    // share of each stage for its 64 reviews, stages are separated by barriers (derived bits live in LDS); a finished
    // formula is balloted straight into its slot's word (GK_RES)
    static_assert(GK_GEN_PARTS == GK_PARTS_K, "generated formula shares do not match the kernel geometry");
    if (!GK_DBG(2u)) {
      for (uint32_t st = 0; st < GK_N_STAGES; st++) {
        if (wave_on) jit_formula_part(st * GK_PARTS_K + part, acc, flags, heap, wbounds, res, hm);
        __syncthreads();
      }
    } else {
      if (wave_on && part == 0) for (uint32_t q = lane; q < (uint32_t)GK_RES_WORDS_K; q += GK_TILE) hm[q] = 0ull;
      __syncthreads();
    }
#else
    // generic build: one wave per half evaluates all formulas of its 64 reviews, then ballots them slot by slot
    if (wave_on && part == 0) {
      if (!GK_DBG(2u)) res = GK_FORMULA_FN(pv, acc, flags, rows, heap, wbounds);
      // (the slots the plan uses: the limits only size the buffers)
      const uint32_t nv_used = min(pv.dims.n_viol, (uint32_t)GK_RES_KV), nm_used = min(pv.dims.n_match, (uint32_t)GK_RES_KM);
      for (uint32_t sl = 0; sl < nv_used; sl++) {
        const unsigned long long mv = __ballot(res.viol_bit(sl));
        if (lane == 0) hm[sl] = mv;
      }
      for (uint32_t sl = 0; sl < nm_used; sl++) {
        const unsigned long long mm = __ballot((res.match >> sl) & 1ull), me = __ballot((res.err >> sl) & 1ull);
        if (lane == 0) { hm[GK_RES_KV + sl] = mm; hm[GK_RES_KV + GK_RES_KM + sl] = me; }
      }
    }
    __syncthreads();
#endif
    GK_PROF(5);
    GK_PRIO_AT(4);
    // the next item's first batch of rows: requested now -- the formulas' registers are free again -- and on their way
    // while the outputs are written and the accumulators are cleared
    // (last use of this item's flags: their register takes the next item's -- a second variable would be filled by a copy
    //  at the loop's back edge, and a copy of a value in flight waits for it, i.e. for everything requested here)
    const unsigned long long ovf_mask = __ballot(ovf);
    unsigned long long usable = __ballot(live && !ovf && !(flags & (RF_SKIP | RF_REFUSE)));
    const unsigned long long refused = __ballot(live && (flags & RF_REFUSE) != 0u);   // never evaluated: the caller fails closed
    // reviews whose caller ran Matcher.Match itself (Driver.Query's contract: the constraints it hands over are evaluated whatever
    // this engine's own match layer says, and nothing is autorejected): every match word gets their bits, every autoreject word loses them
    const unsigned long long prem = __ballot(live && (flags & RF_PREMATCHED) != 0u);
    if (GK_DBG(8u)) usable = 0ull;   // profiling aid: formulas run, outputs suppressed
    const Item nx2 = next_of(nx);
    {
      uint32_t tid_o = tid;   // (opaque, see above)
      GK_OPAQUE_V1(tid_o);
      dn = entry_of(nx2, tid_o);
      flags = flags_of(nx, tid_o);
    }
    {
      const uint32_t nn = has_next ? list_len(s_work[buf ^ 1u]) : 0u;
      // (the slots start from fresh zeros, not from what phase 1 left in them: a conditional load that "keeps the stale value"
      //  ties its destination to the registers the chunk loop ended in, the loop's entry expects the slots elsewhere, and the
      //  compiler reconciled the two with eight moves right behind the loads -- i.e. an s_waitcnt vmcnt(0) two instructions
      //  after the request, the whole output stage behind it, and three spilled thread invariants whose reloads waited once
      //  more.  `tools/jit_inspect.py --waits` lists where loads are issued and first waited for;
      //  tests/test_jit_source.py keeps the distance.)
#pragma unroll
      for (int k = 0; k < GK_PREFETCH; k++) {
        nxt[k].rv = gk_u32x4{0u, 0u, 0u, 0u}; nxt[k].hv = gk_u32x4{0u, 0u, 0u, 0u};
        fetch(s_work[buf ^ 1u], nn, wave + (uint32_t)k * NW, nxt[k]);
      }
    }
    // the output stage is split between two waves of the half (round 3; it was one wave's serial chain of LDS reads, cross-lane
    // gathers and stores while the other waves waited at the next barrier -- 8 % of a group's clocks, profiles/r03_variants_b_*):
    // share 0 writes the violation words (and the pair list), share OUT_B the autoreject / match words and the half's
    // overflow / too_big words
    constexpr uint32_t OUT_B = GK_PARTS_K > 1 ? 1u : 0u;
    if (wave_on && (part == 0 || part == OUT_B)) {
      // (opaque per item: otherwise the per-lane output addresses c * n_tiles are hoisted out of the group loop, spilled, and
      //  their reload -- a load -- waits for the rows just requested)
      uint32_t n_tiles_now = n_tiles;
      GK_OPAQUE_S1(n_tiles_now);
      const uint32_t word = tile * GK_HALVES_K + pass * halves_per_pass + half;     // the half's bitmap word
      const bool do_v = part == 0, do_e = part == OUT_B;
      if (do_e && lane == 0 && word < n_tiles_now) {
        out.overflow[word] = ovf_mask;
        if (ovf_mask) atomicAdd(&out.list_count[1], (uint32_t)__popcll(ovf_mask));
        out.too_big[word] = refused;   // every half writes its word: no clearing pass; the big variant ORs into it later
      }
      // lane s: the words of match / error slot s and of violation slots s, 64 + s, ... (one register pair per bank of 64 violation slots:
      // a plan with up to 64 distinct violation formulas has one)
      const uint32_t sl = lane < (uint32_t)GK_RES_KM ? lane : 0u;
      const unsigned long long wm = hm[GK_RES_KV + sl];
      const unsigned long long we = do_e ? hm[GK_RES_KV + GK_RES_KM + sl] : 0ull;
      unsigned long long wv[GK_RES_BANKS_K];
#pragma unroll
      for (uint32_t bk = 0; bk < (uint32_t)GK_RES_BANKS_K; bk++) wv[bk] = do_v ? hm[bk * 64u + lane < (uint32_t)GK_RES_KV ? bk * 64u + lane : 0u] : 0ull;
      const uint32_t nc = GK_DBG(64u) ? 0u : pv.dims.n_constraints;
      // one batch of 64 constraints: lane c of the batch = constraint cb + c, slw = its (violation slot | match slot << 16)
      auto emit = [&](uint32_t cb, uint32_t slw) {
        const uint32_t c = cb + lane;
        const int sv = (int)(slw & 0xFFFFu), sm = (int)(slw >> 16);
        const unsigned long long m = (((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(wm >> 32), sm) << 32) | (uint32_t)__shfl((int)(uint32_t)wm, sm)) | prem;
        const bool mine = c < nc && word < n_tiles_now;
        if (do_v) {
          unsigned long long v = 0ull;
#pragma unroll
          for (uint32_t bk = 0; bk < (uint32_t)GK_RES_BANKS_K; bk++) {
            const unsigned long long x = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(wv[bk] >> 32), sv & 63) << 32) | (uint32_t)__shfl((int)(uint32_t)wv[bk], sv & 63);
            if (GK_RES_BANKS_K == 1 || (uint32_t)(sv >> 6) == bk) v = x;
          }
          const unsigned long long vw = m & v & usable;
          if (mine) {
            out.viol[(size_t)c * n_tiles_now + word] = vw;
            if (vw && out.list_capacity) {
              // compacted (constraint, review) list: per-constraint totals come from gk_count_rows over the finished bitmap
              const uint32_t n = (uint32_t)__popcll(vw);
              uint32_t q = atomicAdd(&out.list_count[0], n);
              for (unsigned long long w = vw; w; w &= w - 1ull, q++)
                if (q < out.list_capacity) { out.list[2 * q] = c; out.list[2 * q + 1] = r0 + rev_lo + half * GK_TILE + (uint32_t)__builtin_ctzll(w); }
            }
          }
#ifdef GK_TOT_K
          GK_LDS_ADD(&s_tot[lane + (mine ? cb : 0u)], (uint32_t)__popcll(mine ? vw : 0ull));
#endif
        }
        if (do_e) {
          const unsigned long long e = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(we >> 32), sm) << 32) | (uint32_t)__shfl((int)(uint32_t)we, sm);
          const unsigned long long ew = e & usable & ~prem, mw = m & usable;
          if (mine) {
            out.err[(size_t)c * n_tiles_now + word] = ew;
            if (out.match) out.match[(size_t)c * n_tiles_now + word] = mw;
          }
        }
      };
      // the first batch takes its slots from a register (no load on the usual path: a load here would have to wait for
      // the rows requested above -- loads return in order); further batches read theirs
      uint32_t lane_o = lane;   // (opaque: a hoisted, spilled LDS address would come back through a load)
      GK_OPAQUE_V1(lane_o);
      if (nc) emit(0u, s_slw[lane_o]);
      for (uint32_t cb = GK_TILE; cb < nc; cb += GK_TILE) {
        uint32_t slw = 0;
        if (cb + lane < nc) slw = *reinterpret_cast<const uint32_t*>(&slots[cb + lane]);
        emit(cb, slw);
      }
    }
    GK_PROF(6);
    if (!has_next) break;
    it = nx; nx = nx2; buf ^= 1u;
  }
#ifdef GK_TOT_K
  __syncthreads();   // every output wave's adds are in
  {
#pragma unroll
    for (uint32_t q = threadIdx.x; q < (uint32_t)GK_TOT_K; q += GK_BLOCK_K) out.partial[(size_t)blockIdx.x * GK_TOT_K + q] = s_tot[q];
  }
#endif
#undef GK_MAKE_ACC
#undef GK_PRIO_AT
}

#ifndef GK_SKIP_BIG
// ------------------------------------------------------------------------------------------------ big variant
// One wave per review whose arrays exceed the LDS element capacity: walks the bound paths' segments of its tile, takes
// the rows that belong to the review, accumulators in HBM scratch.
GK_KERNEL_LINKAGE __global__ __launch_bounds__(GK_TILE) void GK_KERNEL_BIG(PlanView pv, const Row* __restrict__ rows, const StrHdr* __restrict__ shdr,
                                                       const uint32_t* __restrict__ tile_idx, uint32_t n_slots,
                                                       const Bind* __restrict__ bind, uint32_t n_bind, const uint32_t* __restrict__ rflags,
                                                       const uint8_t* __restrict__ heap, const uint32_t* __restrict__ review_ids,
                                                       uint32_t n_list, uint32_t n_tiles, const ConstraintSlot* __restrict__ slots,
                                                       uint32_t* scratch, OutPtrs out, uint32_t rpt) {
  __shared__ uint32_t s_bounds[GK_MAX_SCOPES];
  const uint32_t lane = threadIdx.x;
  const uint32_t r = review_ids[blockIdx.x];
  const uint32_t group = r / rpt, rev_in_group = r % rpt;   // row group and position in it
  const uint32_t tile = r / GK_TILE, bit = r % GK_TILE;           // bitmap word and bit
  uint32_t* accw = scratch + (size_t)blockIdx.x * pv.dims.acc_words;
  for (uint32_t w = lane; w < pv.dims.acc_words; w += GK_TILE) accw[w] = 0;
  __threadfence_block();
  __syncthreads();
  GlobalAcc acc{accw};
  const uint32_t* __restrict__ ix = tile_idx + (size_t)group * (n_slots + 1u);
  for (uint32_t k = 0; k < n_bind; k++) {
    const Bind b = bind[k];
    const uint32_t row_lo = ix[b.slot], row_hi = ix[b.slot + 1u];
    for (uint32_t i = row_lo + lane; i < row_hi; i += GK_TILE) {
      const uint4 v = reinterpret_cast<const uint4*>(rows)[i];
      Row rw{v.x, v.y, v.z, v.w};
      if ((rw.rev & ROW_REV_MASK) != rev_in_group) continue;
      const uint4 s = reinterpret_cast<const uint4*>(shdr)[i];
      StrHdr h = {{s.x, s.y, s.z, s.w}};
      GK_ROW_FN(rw, i, b.ent, h, pv, heap, acc, true);
    }
  }
  __threadfence_block();
  __syncthreads();
  if (lane < pv.dims.n_scopes) s_bounds[lane] = accw[pv.scopes[lane].count_off];
  __syncthreads();
  if (lane != 0) return;
  if (acc.load(0) & 1u) {   // still overflowing: report, never guess
    atomicOr((unsigned long long*)&out.too_big[tile], 1ull << bit);
    return;
  }
  Results res = GK_FORMULA_FN(pv, acc, rflags[r], rows, heap, s_bounds);
  const bool prem = (rflags[r] & RF_PREMATCHED) != 0u;   // (see the output stage of the dominant kernel)
  for (uint32_t c = 0; c < pv.dims.n_constraints; c++) {
    const ConstraintSlot sl = slots[c];
    const bool m = prem || ((res.match >> sl.match) & 1ull);
    const bool e = !prem && ((res.err >> sl.match) & 1ull);
    const bool v = m && res.viol_bit(sl.viol);
    if (out.match && m) atomicOr((unsigned long long*)&out.match[(size_t)c * n_tiles + tile], 1ull << bit);
    if (e) atomicOr((unsigned long long*)&out.err[(size_t)c * n_tiles + tile], 1ull << bit);
    if (v) {
      atomicOr((unsigned long long*)&out.viol[(size_t)c * n_tiles + tile], 1ull << bit);
      if (out.list_capacity) {
        const uint32_t k = atomicAdd(&out.list_count[0], 1u);
        if (k < out.list_capacity) { out.list[2 * k] = c; out.list[2 * k + 1] = r; }
      }
    }
  }
  (void)n_list;
}
#endif   // GK_SKIP_BIG
}
