// Flattened-review table layout and the compiled predicate plan, shared by host (compiler, flattener) and device
// (kernels.hip).  Everything here is POD with fixed-width fields.
//
// This is the MI355X engine's replacement for what the reference holds as Go structs + OPA ASTs on the hot path:
//   pkg/target/matcher.go:73-93 re-unmarshals object/oldObject per (constraint, review); here a review is flattened
//   ONCE into 16-byte rows that every constraint program reads, and match blocks (pkg/mutation/match/match.go:32-65)
//   plus template Rego are compiled into the same predicate/formula plan.
#pragma once
#include <cstdint>

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
