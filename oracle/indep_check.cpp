// ORACLE (test infrastructure only -- never linked into, loaded by or called from the product path).
//
// The INDEPENDENT COMPILED CHECKER of bench.py's full-size parity leg: a C++ restatement of the Python oracle
//   oracle/values.py  oracle/rego_parser.py  oracle/rego_interp.py  oracle/rego_builtins.py (the builtins the bench's policy
//   sets call)  oracle/match.py  oracle/target.py (object reviews)  oracle/client.py Client.review
// with its own JSON reader, its own value model, its own Rego parser and tree-walking interpreter and its own Match layer.
// NOTHING of gatekeeper_amd/csrc is compiled into it or linked with it (oracle/Makefile: this one file, the C++ standard
// library, pthreads); of the product it sees only the public C header's `gk_review_in` (a struct of pointers and lengths),
// so that the JSON TEXT of the batch the timed table was built from is read where it lies.
//
// What it restates (through the Python oracle, which cites them line by line): pkg/target/target.go:81-179 (HandleReview for
// unstructured objects), pkg/target/matcher.go:21-93, pkg/mutation/match/match.go:32-268, pkg/wildcard/wildcard.go:17-41,
// apimachinery's label selectors, and OPA's topdown evaluation of the template's `violation` set (frameworks Driver.Query,
// called from pkg/audit/manager.go:621,719).  It answers, per (constraint, object): does the constraint match and the template
// yield a result (violation bit), or does matching fail (autoreject bit) -- the two bitmaps the device produces.
// Scope: AugmentedUnstructured{object, namespace, source "Original"} reviews at the audit enforcement point, the Rego subset of
// oracle/rego_parser.py, the builtins listed in BUILTINS below; anything else is reported as an error, never guessed.
// Pinned by tests/test_indep_check.py against the Python oracle (fixtures, synthetic configs, the corpus).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../include/gkgpu.h"   // gk_review_in only (plain C struct: pointers and lengths)

namespace ic {

typedef __int128 i128;

// ================================================================================================ values (oracle/values.py)
struct V;
typedef std::shared_ptr<const V> VP;
struct V {
  enum K { Null = 0, Bool = 1, Num = 2, Str = 3, Arr = 4, Obj = 5, Set = 6 } k = Null;
  bool b = false;
  bool is_int = true;
  i128 i = 0;
  double d = 0;
  std::string s;
  std::vector<VP> a;                      // Arr: in order; Set: sorted, unique
  std::vector<std::pair<VP, VP>> o;       // Obj: sorted by key, unique keys
};
int compare(const VP& a, const VP& b);
static VP mk_null() { static const VP v = std::make_shared<const V>(); return v; }
static VP mk_bool(bool x) { static const VP t = [] { V v; v.k = V::Bool; v.b = true; return std::make_shared<const V>(v); }(), f = [] { V v; v.k = V::Bool; v.b = false; return std::make_shared<const V>(v); }(); return x ? t : f; }
static VP mk_int(i128 x) { V v; v.k = V::Num; v.is_int = true; v.i = x; v.d = (double)x; return std::make_shared<const V>(std::move(v)); }
static VP mk_float(double x) {   // (_norm: an integral float below 2^63 is the integer)
  V v; v.k = V::Num;
  if (std::isfinite(x) && std::floor(x) == x && std::fabs(x) < 9223372036854775808.0) { v.is_int = true; v.i = (i128)x; v.d = x; }
  else { v.is_int = false; v.d = x; }
  return std::make_shared<const V>(std::move(v));
}
static VP mk_str(std::string s) { V v; v.k = V::Str; v.s = std::move(s); return std::make_shared<const V>(std::move(v)); }
static VP mk_arr(std::vector<VP> a) { V v; v.k = V::Arr; v.a = std::move(a); return std::make_shared<const V>(std::move(v)); }
static VP mk_set(std::vector<VP> a) {
  std::sort(a.begin(), a.end(), [](const VP& x, const VP& y) { return compare(x, y) < 0; });
  a.erase(std::unique(a.begin(), a.end(), [](const VP& x, const VP& y) { return compare(x, y) == 0; }), a.end());
  V v; v.k = V::Set; v.a = std::move(a); return std::make_shared<const V>(std::move(v));
}
static VP mk_obj(std::vector<std::pair<VP, VP>> o) {   // RObj(pairs): a later pair with an equal key replaces the earlier one
  std::stable_sort(o.begin(), o.end(), [](const std::pair<VP, VP>& x, const std::pair<VP, VP>& y) { return compare(x.first, y.first) < 0; });
  std::vector<std::pair<VP, VP>> out;
  for (auto& p : o) { if (!out.empty() && compare(out.back().first, p.first) == 0) out.back().second = p.second; else out.push_back(p); }
  V v; v.k = V::Obj; v.o = std::move(out); return std::make_shared<const V>(std::move(v));
}
static int cmp_num(const V& a, const V& b) {
  if (a.is_int && b.is_int) return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
  const double x = a.is_int ? (double)a.i : a.d, y = b.is_int ? (double)b.i : b.d;
  return x < y ? -1 : (x > y ? 1 : 0);
}
int compare(const VP& a, const VP& b) {   // values.py compare: null < boolean < number < string < array < object < set
  if (a->k != b->k) return a->k < b->k ? -1 : 1;
  switch (a->k) {
    case V::Null: return 0;
    case V::Bool: return (int)a->b - (int)b->b;
    case V::Num: return cmp_num(*a, *b);
    case V::Str: { const int c = a->s.compare(b->s); return c < 0 ? -1 : (c > 0 ? 1 : 0); }
    case V::Arr: case V::Set: {
      const size_t n = std::min(a->a.size(), b->a.size());
      for (size_t i = 0; i < n; i++) { const int c = compare(a->a[i], b->a[i]); if (c) return c; }
      return a->a.size() < b->a.size() ? -1 : (a->a.size() > b->a.size() ? 1 : 0);
    }
    case V::Obj: {
      const size_t n = std::min(a->o.size(), b->o.size());
      for (size_t i = 0; i < n; i++) {
        int c = compare(a->o[i].first, b->o[i].first); if (c) return c;
        c = compare(a->o[i].second, b->o[i].second); if (c) return c;
      }
      return a->o.size() < b->o.size() ? -1 : (a->o.size() > b->o.size() ? 1 : 0);
    }
  }
  return 0;
}
static bool equal(const VP& a, const VP& b) { return compare(a, b) == 0; }
static const VP* obj_get(const V& o, const VP& key) {
  if (o.k != V::Obj) return nullptr;
  auto it = std::lower_bound(o.o.begin(), o.o.end(), key, [](const std::pair<VP, VP>& p, const VP& k) { return compare(p.first, k) < 0; });
  return it != o.o.end() && compare(it->first, key) == 0 ? &it->second : nullptr;
}
static const VP* obj_get(const V& o, const char* key) {
  if (o.k != V::Obj) return nullptr;
  for (auto& p : o.o) if (p.first->k == V::Str && p.first->s == key) return &p.second;
  return nullptr;
}
static bool set_has(const V& s, const VP& x) { return std::binary_search(s.a.begin(), s.a.end(), x, [](const VP& p, const VP& q) { return compare(p, q) < 0; }); }
static size_t utf8_len(const std::string& s) { size_t n = 0; for (unsigned char c : s) if ((c & 0xC0) != 0x80) n++; return n; }

// ast term String() (values.py to_string) -- what sprintf("%v") prints
// BEGIN GENERATED PRINT RANGES
// GENERATED by tools/gen_unicode_print.py from Python's unicodedata 13.0.0: the code points >= U+0080 that Go's strconv.IsPrint calls
// printable (categories L, M, N, P, S), as inclusive ranges in ascending order.
static const uint32_t kIcPrintRanges[][2] = {
  {0xA1,0xAC}, {0xAE,0x377}, {0x37A,0x37F}, {0x384,0x38A}, {0x38C,0x38C}, {0x38E,0x3A1}, {0x3A3,0x52F}, {0x531,0x556},
  {0x559,0x58A}, {0x58D,0x58F}, {0x591,0x5C7}, {0x5D0,0x5EA}, {0x5EF,0x5F4}, {0x606,0x61B}, {0x61E,0x6DC}, {0x6DE,0x70D},
  {0x710,0x74A}, {0x74D,0x7B1}, {0x7C0,0x7FA}, {0x7FD,0x82D}, {0x830,0x83E}, {0x840,0x85B}, {0x85E,0x85E}, {0x860,0x86A},
  {0x8A0,0x8B4}, {0x8B6,0x8C7}, {0x8D3,0x8E1}, {0x8E3,0x983}, {0x985,0x98C}, {0x98F,0x990}, {0x993,0x9A8}, {0x9AA,0x9B0},
  {0x9B2,0x9B2}, {0x9B6,0x9B9}, {0x9BC,0x9C4}, {0x9C7,0x9C8}, {0x9CB,0x9CE}, {0x9D7,0x9D7}, {0x9DC,0x9DD}, {0x9DF,0x9E3},
  {0x9E6,0x9FE}, {0xA01,0xA03}, {0xA05,0xA0A}, {0xA0F,0xA10}, {0xA13,0xA28}, {0xA2A,0xA30}, {0xA32,0xA33}, {0xA35,0xA36},
  {0xA38,0xA39}, {0xA3C,0xA3C}, {0xA3E,0xA42}, {0xA47,0xA48}, {0xA4B,0xA4D}, {0xA51,0xA51}, {0xA59,0xA5C}, {0xA5E,0xA5E},
  {0xA66,0xA76}, {0xA81,0xA83}, {0xA85,0xA8D}, {0xA8F,0xA91}, {0xA93,0xAA8}, {0xAAA,0xAB0}, {0xAB2,0xAB3}, {0xAB5,0xAB9},
  {0xABC,0xAC5}, {0xAC7,0xAC9}, {0xACB,0xACD}, {0xAD0,0xAD0}, {0xAE0,0xAE3}, {0xAE6,0xAF1}, {0xAF9,0xAFF}, {0xB01,0xB03},
  {0xB05,0xB0C}, {0xB0F,0xB10}, {0xB13,0xB28}, {0xB2A,0xB30}, {0xB32,0xB33}, {0xB35,0xB39}, {0xB3C,0xB44}, {0xB47,0xB48},
  {0xB4B,0xB4D}, {0xB55,0xB57}, {0xB5C,0xB5D}, {0xB5F,0xB63}, {0xB66,0xB77}, {0xB82,0xB83}, {0xB85,0xB8A}, {0xB8E,0xB90},
  {0xB92,0xB95}, {0xB99,0xB9A}, {0xB9C,0xB9C}, {0xB9E,0xB9F}, {0xBA3,0xBA4}, {0xBA8,0xBAA}, {0xBAE,0xBB9}, {0xBBE,0xBC2},
  {0xBC6,0xBC8}, {0xBCA,0xBCD}, {0xBD0,0xBD0}, {0xBD7,0xBD7}, {0xBE6,0xBFA}, {0xC00,0xC0C}, {0xC0E,0xC10}, {0xC12,0xC28},
  {0xC2A,0xC39}, {0xC3D,0xC44}, {0xC46,0xC48}, {0xC4A,0xC4D}, {0xC55,0xC56}, {0xC58,0xC5A}, {0xC60,0xC63}, {0xC66,0xC6F},
  {0xC77,0xC8C}, {0xC8E,0xC90}, {0xC92,0xCA8}, {0xCAA,0xCB3}, {0xCB5,0xCB9}, {0xCBC,0xCC4}, {0xCC6,0xCC8}, {0xCCA,0xCCD},
  {0xCD5,0xCD6}, {0xCDE,0xCDE}, {0xCE0,0xCE3}, {0xCE6,0xCEF}, {0xCF1,0xCF2}, {0xD00,0xD0C}, {0xD0E,0xD10}, {0xD12,0xD44},
  {0xD46,0xD48}, {0xD4A,0xD4F}, {0xD54,0xD63}, {0xD66,0xD7F}, {0xD81,0xD83}, {0xD85,0xD96}, {0xD9A,0xDB1}, {0xDB3,0xDBB},
  {0xDBD,0xDBD}, {0xDC0,0xDC6}, {0xDCA,0xDCA}, {0xDCF,0xDD4}, {0xDD6,0xDD6}, {0xDD8,0xDDF}, {0xDE6,0xDEF}, {0xDF2,0xDF4},
  {0xE01,0xE3A}, {0xE3F,0xE5B}, {0xE81,0xE82}, {0xE84,0xE84}, {0xE86,0xE8A}, {0xE8C,0xEA3}, {0xEA5,0xEA5}, {0xEA7,0xEBD},
  {0xEC0,0xEC4}, {0xEC6,0xEC6}, {0xEC8,0xECD}, {0xED0,0xED9}, {0xEDC,0xEDF}, {0xF00,0xF47}, {0xF49,0xF6C}, {0xF71,0xF97},
  {0xF99,0xFBC}, {0xFBE,0xFCC}, {0xFCE,0xFDA}, {0x1000,0x10C5}, {0x10C7,0x10C7}, {0x10CD,0x10CD}, {0x10D0,0x1248}, {0x124A,0x124D},
  {0x1250,0x1256}, {0x1258,0x1258}, {0x125A,0x125D}, {0x1260,0x1288}, {0x128A,0x128D}, {0x1290,0x12B0}, {0x12B2,0x12B5}, {0x12B8,0x12BE},
  {0x12C0,0x12C0}, {0x12C2,0x12C5}, {0x12C8,0x12D6}, {0x12D8,0x1310}, {0x1312,0x1315}, {0x1318,0x135A}, {0x135D,0x137C}, {0x1380,0x1399},
  {0x13A0,0x13F5}, {0x13F8,0x13FD}, {0x1400,0x167F}, {0x1681,0x169C}, {0x16A0,0x16F8}, {0x1700,0x170C}, {0x170E,0x1714}, {0x1720,0x1736},
  {0x1740,0x1753}, {0x1760,0x176C}, {0x176E,0x1770}, {0x1772,0x1773}, {0x1780,0x17DD}, {0x17E0,0x17E9}, {0x17F0,0x17F9}, {0x1800,0x180D},
  {0x1810,0x1819}, {0x1820,0x1878}, {0x1880,0x18AA}, {0x18B0,0x18F5}, {0x1900,0x191E}, {0x1920,0x192B}, {0x1930,0x193B}, {0x1940,0x1940},
  {0x1944,0x196D}, {0x1970,0x1974}, {0x1980,0x19AB}, {0x19B0,0x19C9}, {0x19D0,0x19DA}, {0x19DE,0x1A1B}, {0x1A1E,0x1A5E}, {0x1A60,0x1A7C},
  {0x1A7F,0x1A89}, {0x1A90,0x1A99}, {0x1AA0,0x1AAD}, {0x1AB0,0x1AC0}, {0x1B00,0x1B4B}, {0x1B50,0x1B7C}, {0x1B80,0x1BF3}, {0x1BFC,0x1C37},
  {0x1C3B,0x1C49}, {0x1C4D,0x1C88}, {0x1C90,0x1CBA}, {0x1CBD,0x1CC7}, {0x1CD0,0x1CFA}, {0x1D00,0x1DF9}, {0x1DFB,0x1F15}, {0x1F18,0x1F1D},
  {0x1F20,0x1F45}, {0x1F48,0x1F4D}, {0x1F50,0x1F57}, {0x1F59,0x1F59}, {0x1F5B,0x1F5B}, {0x1F5D,0x1F5D}, {0x1F5F,0x1F7D}, {0x1F80,0x1FB4},
  {0x1FB6,0x1FC4}, {0x1FC6,0x1FD3}, {0x1FD6,0x1FDB}, {0x1FDD,0x1FEF}, {0x1FF2,0x1FF4}, {0x1FF6,0x1FFE}, {0x2010,0x2027}, {0x2030,0x205E},
  {0x2070,0x2071}, {0x2074,0x208E}, {0x2090,0x209C}, {0x20A0,0x20BF}, {0x20D0,0x20F0}, {0x2100,0x218B}, {0x2190,0x2426}, {0x2440,0x244A},
  {0x2460,0x2B73}, {0x2B76,0x2B95}, {0x2B97,0x2C2E}, {0x2C30,0x2C5E}, {0x2C60,0x2CF3}, {0x2CF9,0x2D25}, {0x2D27,0x2D27}, {0x2D2D,0x2D2D},
  {0x2D30,0x2D67}, {0x2D6F,0x2D70}, {0x2D7F,0x2D96}, {0x2DA0,0x2DA6}, {0x2DA8,0x2DAE}, {0x2DB0,0x2DB6}, {0x2DB8,0x2DBE}, {0x2DC0,0x2DC6},
  {0x2DC8,0x2DCE}, {0x2DD0,0x2DD6}, {0x2DD8,0x2DDE}, {0x2DE0,0x2E52}, {0x2E80,0x2E99}, {0x2E9B,0x2EF3}, {0x2F00,0x2FD5}, {0x2FF0,0x2FFB},
  {0x3001,0x303F}, {0x3041,0x3096}, {0x3099,0x30FF}, {0x3105,0x312F}, {0x3131,0x318E}, {0x3190,0x31E3}, {0x31F0,0x321E}, {0x3220,0x9FFC},
  {0xA000,0xA48C}, {0xA490,0xA4C6}, {0xA4D0,0xA62B}, {0xA640,0xA6F7}, {0xA700,0xA7BF}, {0xA7C2,0xA7CA}, {0xA7F5,0xA82C}, {0xA830,0xA839},
  {0xA840,0xA877}, {0xA880,0xA8C5}, {0xA8CE,0xA8D9}, {0xA8E0,0xA953}, {0xA95F,0xA97C}, {0xA980,0xA9CD}, {0xA9CF,0xA9D9}, {0xA9DE,0xA9FE},
  {0xAA00,0xAA36}, {0xAA40,0xAA4D}, {0xAA50,0xAA59}, {0xAA5C,0xAAC2}, {0xAADB,0xAAF6}, {0xAB01,0xAB06}, {0xAB09,0xAB0E}, {0xAB11,0xAB16},
  {0xAB20,0xAB26}, {0xAB28,0xAB2E}, {0xAB30,0xAB6B}, {0xAB70,0xABED}, {0xABF0,0xABF9}, {0xAC00,0xD7A3}, {0xD7B0,0xD7C6}, {0xD7CB,0xD7FB},
  {0xF900,0xFA6D}, {0xFA70,0xFAD9}, {0xFB00,0xFB06}, {0xFB13,0xFB17}, {0xFB1D,0xFB36}, {0xFB38,0xFB3C}, {0xFB3E,0xFB3E}, {0xFB40,0xFB41},
  {0xFB43,0xFB44}, {0xFB46,0xFBC1}, {0xFBD3,0xFD3F}, {0xFD50,0xFD8F}, {0xFD92,0xFDC7}, {0xFDF0,0xFDFD}, {0xFE00,0xFE19}, {0xFE20,0xFE52},
  {0xFE54,0xFE66}, {0xFE68,0xFE6B}, {0xFE70,0xFE74}, {0xFE76,0xFEFC}, {0xFF01,0xFFBE}, {0xFFC2,0xFFC7}, {0xFFCA,0xFFCF}, {0xFFD2,0xFFD7},
  {0xFFDA,0xFFDC}, {0xFFE0,0xFFE6}, {0xFFE8,0xFFEE}, {0xFFFC,0xFFFD}, {0x10000,0x1000B}, {0x1000D,0x10026}, {0x10028,0x1003A}, {0x1003C,0x1003D},
  {0x1003F,0x1004D}, {0x10050,0x1005D}, {0x10080,0x100FA}, {0x10100,0x10102}, {0x10107,0x10133}, {0x10137,0x1018E}, {0x10190,0x1019C}, {0x101A0,0x101A0},
  {0x101D0,0x101FD}, {0x10280,0x1029C}, {0x102A0,0x102D0}, {0x102E0,0x102FB}, {0x10300,0x10323}, {0x1032D,0x1034A}, {0x10350,0x1037A}, {0x10380,0x1039D},
  {0x1039F,0x103C3}, {0x103C8,0x103D5}, {0x10400,0x1049D}, {0x104A0,0x104A9}, {0x104B0,0x104D3}, {0x104D8,0x104FB}, {0x10500,0x10527}, {0x10530,0x10563},
  {0x1056F,0x1056F}, {0x10600,0x10736}, {0x10740,0x10755}, {0x10760,0x10767}, {0x10800,0x10805}, {0x10808,0x10808}, {0x1080A,0x10835}, {0x10837,0x10838},
  {0x1083C,0x1083C}, {0x1083F,0x10855}, {0x10857,0x1089E}, {0x108A7,0x108AF}, {0x108E0,0x108F2}, {0x108F4,0x108F5}, {0x108FB,0x1091B}, {0x1091F,0x10939},
  {0x1093F,0x1093F}, {0x10980,0x109B7}, {0x109BC,0x109CF}, {0x109D2,0x10A03}, {0x10A05,0x10A06}, {0x10A0C,0x10A13}, {0x10A15,0x10A17}, {0x10A19,0x10A35},
  {0x10A38,0x10A3A}, {0x10A3F,0x10A48}, {0x10A50,0x10A58}, {0x10A60,0x10A9F}, {0x10AC0,0x10AE6}, {0x10AEB,0x10AF6}, {0x10B00,0x10B35}, {0x10B39,0x10B55},
  {0x10B58,0x10B72}, {0x10B78,0x10B91}, {0x10B99,0x10B9C}, {0x10BA9,0x10BAF}, {0x10C00,0x10C48}, {0x10C80,0x10CB2}, {0x10CC0,0x10CF2}, {0x10CFA,0x10D27},
  {0x10D30,0x10D39}, {0x10E60,0x10E7E}, {0x10E80,0x10EA9}, {0x10EAB,0x10EAD}, {0x10EB0,0x10EB1}, {0x10F00,0x10F27}, {0x10F30,0x10F59}, {0x10FB0,0x10FCB},
  {0x10FE0,0x10FF6}, {0x11000,0x1104D}, {0x11052,0x1106F}, {0x1107F,0x110BC}, {0x110BE,0x110C1}, {0x110D0,0x110E8}, {0x110F0,0x110F9}, {0x11100,0x11134},
  {0x11136,0x11147}, {0x11150,0x11176}, {0x11180,0x111DF}, {0x111E1,0x111F4}, {0x11200,0x11211}, {0x11213,0x1123E}, {0x11280,0x11286}, {0x11288,0x11288},
  {0x1128A,0x1128D}, {0x1128F,0x1129D}, {0x1129F,0x112A9}, {0x112B0,0x112EA}, {0x112F0,0x112F9}, {0x11300,0x11303}, {0x11305,0x1130C}, {0x1130F,0x11310},
  {0x11313,0x11328}, {0x1132A,0x11330}, {0x11332,0x11333}, {0x11335,0x11339}, {0x1133B,0x11344}, {0x11347,0x11348}, {0x1134B,0x1134D}, {0x11350,0x11350},
  {0x11357,0x11357}, {0x1135D,0x11363}, {0x11366,0x1136C}, {0x11370,0x11374}, {0x11400,0x1145B}, {0x1145D,0x11461}, {0x11480,0x114C7}, {0x114D0,0x114D9},
  {0x11580,0x115B5}, {0x115B8,0x115DD}, {0x11600,0x11644}, {0x11650,0x11659}, {0x11660,0x1166C}, {0x11680,0x116B8}, {0x116C0,0x116C9}, {0x11700,0x1171A},
  {0x1171D,0x1172B}, {0x11730,0x1173F}, {0x11800,0x1183B}, {0x118A0,0x118F2}, {0x118FF,0x11906}, {0x11909,0x11909}, {0x1190C,0x11913}, {0x11915,0x11916},
  {0x11918,0x11935}, {0x11937,0x11938}, {0x1193B,0x11946}, {0x11950,0x11959}, {0x119A0,0x119A7}, {0x119AA,0x119D7}, {0x119DA,0x119E4}, {0x11A00,0x11A47},
  {0x11A50,0x11AA2}, {0x11AC0,0x11AF8}, {0x11C00,0x11C08}, {0x11C0A,0x11C36}, {0x11C38,0x11C45}, {0x11C50,0x11C6C}, {0x11C70,0x11C8F}, {0x11C92,0x11CA7},
  {0x11CA9,0x11CB6}, {0x11D00,0x11D06}, {0x11D08,0x11D09}, {0x11D0B,0x11D36}, {0x11D3A,0x11D3A}, {0x11D3C,0x11D3D}, {0x11D3F,0x11D47}, {0x11D50,0x11D59},
  {0x11D60,0x11D65}, {0x11D67,0x11D68}, {0x11D6A,0x11D8E}, {0x11D90,0x11D91}, {0x11D93,0x11D98}, {0x11DA0,0x11DA9}, {0x11EE0,0x11EF8}, {0x11FB0,0x11FB0},
  {0x11FC0,0x11FF1}, {0x11FFF,0x12399}, {0x12400,0x1246E}, {0x12470,0x12474}, {0x12480,0x12543}, {0x13000,0x1342E}, {0x14400,0x14646}, {0x16800,0x16A38},
  {0x16A40,0x16A5E}, {0x16A60,0x16A69}, {0x16A6E,0x16A6F}, {0x16AD0,0x16AED}, {0x16AF0,0x16AF5}, {0x16B00,0x16B45}, {0x16B50,0x16B59}, {0x16B5B,0x16B61},
  {0x16B63,0x16B77}, {0x16B7D,0x16B8F}, {0x16E40,0x16E9A}, {0x16F00,0x16F4A}, {0x16F4F,0x16F87}, {0x16F8F,0x16F9F}, {0x16FE0,0x16FE4}, {0x16FF0,0x16FF1},
  {0x17000,0x187F7}, {0x18800,0x18CD5}, {0x18D00,0x18D08}, {0x1B000,0x1B11E}, {0x1B150,0x1B152}, {0x1B164,0x1B167}, {0x1B170,0x1B2FB}, {0x1BC00,0x1BC6A},
  {0x1BC70,0x1BC7C}, {0x1BC80,0x1BC88}, {0x1BC90,0x1BC99}, {0x1BC9C,0x1BC9F}, {0x1D000,0x1D0F5}, {0x1D100,0x1D126}, {0x1D129,0x1D172}, {0x1D17B,0x1D1E8},
  {0x1D200,0x1D245}, {0x1D2E0,0x1D2F3}, {0x1D300,0x1D356}, {0x1D360,0x1D378}, {0x1D400,0x1D454}, {0x1D456,0x1D49C}, {0x1D49E,0x1D49F}, {0x1D4A2,0x1D4A2},
  {0x1D4A5,0x1D4A6}, {0x1D4A9,0x1D4AC}, {0x1D4AE,0x1D4B9}, {0x1D4BB,0x1D4BB}, {0x1D4BD,0x1D4C3}, {0x1D4C5,0x1D505}, {0x1D507,0x1D50A}, {0x1D50D,0x1D514},
  {0x1D516,0x1D51C}, {0x1D51E,0x1D539}, {0x1D53B,0x1D53E}, {0x1D540,0x1D544}, {0x1D546,0x1D546}, {0x1D54A,0x1D550}, {0x1D552,0x1D6A5}, {0x1D6A8,0x1D7CB},
  {0x1D7CE,0x1DA8B}, {0x1DA9B,0x1DA9F}, {0x1DAA1,0x1DAAF}, {0x1E000,0x1E006}, {0x1E008,0x1E018}, {0x1E01B,0x1E021}, {0x1E023,0x1E024}, {0x1E026,0x1E02A},
  {0x1E100,0x1E12C}, {0x1E130,0x1E13D}, {0x1E140,0x1E149}, {0x1E14E,0x1E14F}, {0x1E2C0,0x1E2F9}, {0x1E2FF,0x1E2FF}, {0x1E800,0x1E8C4}, {0x1E8C7,0x1E8D6},
  {0x1E900,0x1E94B}, {0x1E950,0x1E959}, {0x1E95E,0x1E95F}, {0x1EC71,0x1ECB4}, {0x1ED01,0x1ED3D}, {0x1EE00,0x1EE03}, {0x1EE05,0x1EE1F}, {0x1EE21,0x1EE22},
  {0x1EE24,0x1EE24}, {0x1EE27,0x1EE27}, {0x1EE29,0x1EE32}, {0x1EE34,0x1EE37}, {0x1EE39,0x1EE39}, {0x1EE3B,0x1EE3B}, {0x1EE42,0x1EE42}, {0x1EE47,0x1EE47},
  {0x1EE49,0x1EE49}, {0x1EE4B,0x1EE4B}, {0x1EE4D,0x1EE4F}, {0x1EE51,0x1EE52}, {0x1EE54,0x1EE54}, {0x1EE57,0x1EE57}, {0x1EE59,0x1EE59}, {0x1EE5B,0x1EE5B},
  {0x1EE5D,0x1EE5D}, {0x1EE5F,0x1EE5F}, {0x1EE61,0x1EE62}, {0x1EE64,0x1EE64}, {0x1EE67,0x1EE6A}, {0x1EE6C,0x1EE72}, {0x1EE74,0x1EE77}, {0x1EE79,0x1EE7C},
  {0x1EE7E,0x1EE7E}, {0x1EE80,0x1EE89}, {0x1EE8B,0x1EE9B}, {0x1EEA1,0x1EEA3}, {0x1EEA5,0x1EEA9}, {0x1EEAB,0x1EEBB}, {0x1EEF0,0x1EEF1}, {0x1F000,0x1F02B},
  {0x1F030,0x1F093}, {0x1F0A0,0x1F0AE}, {0x1F0B1,0x1F0BF}, {0x1F0C1,0x1F0CF}, {0x1F0D1,0x1F0F5}, {0x1F100,0x1F1AD}, {0x1F1E6,0x1F202}, {0x1F210,0x1F23B},
  {0x1F240,0x1F248}, {0x1F250,0x1F251}, {0x1F260,0x1F265}, {0x1F300,0x1F6D7}, {0x1F6E0,0x1F6EC}, {0x1F6F0,0x1F6FC}, {0x1F700,0x1F773}, {0x1F780,0x1F7D8},
  {0x1F7E0,0x1F7EB}, {0x1F800,0x1F80B}, {0x1F810,0x1F847}, {0x1F850,0x1F859}, {0x1F860,0x1F887}, {0x1F890,0x1F8AD}, {0x1F8B0,0x1F8B1}, {0x1F900,0x1F978},
  {0x1F97A,0x1F9CB}, {0x1F9CD,0x1FA53}, {0x1FA60,0x1FA6D}, {0x1FA70,0x1FA74}, {0x1FA78,0x1FA7A}, {0x1FA80,0x1FA86}, {0x1FA90,0x1FAA8}, {0x1FAB0,0x1FAB6},
  {0x1FAC0,0x1FAC2}, {0x1FAD0,0x1FAD6}, {0x1FB00,0x1FB92}, {0x1FB94,0x1FBCA}, {0x1FBF0,0x1FBF9}, {0x20000,0x2A6DD}, {0x2A700,0x2B734}, {0x2B740,0x2B81D},
  {0x2B820,0x2CEA1}, {0x2CEB0,0x2EBE0}, {0x2F800,0x2FA1D}, {0x30000,0x3134A}, {0xE0100,0xE01EF},
};
// END GENERATED PRINT RANGES
static bool rune_is_print(uint32_t r) {   // strconv.IsPrint for r >= 0x80
  size_t lo = 0, hi = sizeof kIcPrintRanges / sizeof kIcPrintRanges[0];
  while (lo < hi) { const size_t mid = (lo + hi) / 2; if (kIcPrintRanges[mid][1] < r) lo = mid + 1; else hi = mid; }
  return lo < sizeof kIcPrintRanges / sizeof kIcPrintRanges[0] && kIcPrintRanges[lo][0] <= r;
}
// strconv.Quote: printable runes as they are, the C escapes, \xNN for other ASCII controls and for bytes that are no UTF-8, \uNNNN /
// \UNNNNNNNN for every other rune that is not printable (NBSP, format characters, unassigned and private-use code points ...)
static std::string quote(const std::string& s) {
  std::string o = "\"";
  char buf[16];
  for (size_t i = 0; i < s.size();) {
    const unsigned char c = (unsigned char)s[i];
    if (c < 0x80) {
      switch (c) {
        case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break; case '\t': o += "\\t"; break; case '\r': o += "\\r"; break;
        case '\a': o += "\\a"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break; case '\v': o += "\\v"; break;
        default: if (c < 0x20 || c == 0x7F) { snprintf(buf, sizeof buf, "\\x%02x", c); o += buf; } else o.push_back((char)c);
      }
      i++;
      continue;
    }
    int len = c >= 0xF0 ? 4 : c >= 0xE0 ? 3 : c >= 0xC0 ? 2 : 0;
    uint32_t r = len == 4 ? c & 7u : len == 3 ? c & 15u : c & 31u;
    bool ok = len && i + (size_t)len <= s.size();
    for (int k = 1; ok && k < len; k++) { const unsigned char d = (unsigned char)s[i + (size_t)k]; if ((d & 0xC0) != 0x80) ok = false; else r = (r << 6) | (d & 63u); }
    if (ok && (r < (len == 2 ? 0x80u : len == 3 ? 0x800u : 0x10000u) || r > 0x10FFFF || (r >= 0xD800 && r <= 0xDFFF))) ok = false;   // overlong, out of range, surrogate
    if (!ok) { snprintf(buf, sizeof buf, "\\x%02x", c); o += buf; i++; continue; }
    if (rune_is_print(r)) o.append(s, i, (size_t)len);
    else { snprintf(buf, sizeof buf, r < 0x10000 ? "\\u%04x" : "\\U%08x", r); o += buf; }
    i += (size_t)len;
  }
  return o + "\"";
}
static std::string i128_str(i128 x) {
  if (x == 0) return "0";
  const bool neg = x < 0;
  unsigned __int128 u = neg ? (unsigned __int128)(-(x + 1)) + 1 : (unsigned __int128)x;
  std::string s;
  while (u) { s.push_back((char)('0' + (int)(u % 10))); u /= 10; }
  if (neg) s.push_back('-');
  std::reverse(s.begin(), s.end());
  return s;
}
// shortest digits that round-trip (what Go's strconv and Python's repr print): (digits, decimal exponent of the first digit)
static void shortest_digits(double f, std::string* digits, int* x) {
  char buf[64];
  for (int p = 1; p <= 17; p++) {
    snprintf(buf, sizeof buf, "%.*e", p - 1, std::fabs(f));
    if (strtod(buf, nullptr) == std::fabs(f) || p == 17) break;
  }
  std::string m(buf);
  const size_t e = m.find('e');
  *x = atoi(m.c_str() + e + 1);
  std::string d;
  for (size_t i = 0; i < e; i++) if (m[i] != '.') d.push_back(m[i]);
  while (d.size() > 1 && d.back() == '0') d.pop_back();
  *digits = d;
}
static std::string fmt_e(const std::string& sign, const std::string& digits, int x, int min_exp_digits) {
  std::string m = digits.substr(0, 1) + (digits.size() > 1 ? "." + digits.substr(1) : "");
  char eb[16];
  snprintf(eb, sizeof eb, "e%c%0*d", x >= 0 ? '+' : '-', min_exp_digits, x >= 0 ? x : -x);
  return sign + m + eb;
}
static std::string fmt_f(const std::string& sign, const std::string& digits, int x) {
  if (x >= 0) {
    if ((int)digits.size() <= x + 1) return sign + digits + std::string((size_t)(x + 1 - (int)digits.size()), '0');
    return sign + digits.substr(0, (size_t)x + 1) + "." + digits.substr((size_t)x + 1);
  }
  return sign + "0." + std::string((size_t)(-x - 1), '0') + digits;
}
static std::string json_float_text(double f) {   // encoding/json floatEncoder (values.py json_float_text)
  if (f == 0) return std::signbit(f) ? "-0" : "0";
  std::string digits; int x;
  shortest_digits(f, &digits, &x);
  const std::string sign = f < 0 ? "-" : "";
  if (std::fabs(f) < 1e-6 || std::fabs(f) >= 1e21) {
    std::string s = fmt_e(sign, digits, x, 2);
    if (s.size() >= 4 && s[s.size() - 4] == 'e' && s[s.size() - 3] == '-' && s[s.size() - 2] == '0') s = s.substr(0, s.size() - 2) + s.back();
    return s;
  }
  return fmt_f(sign, digits, x);
}
static std::string go_float_v(double f) {   // fmt %v of a float64 (values.py go_float_v)
  if (f != f) return "NaN";
  if (std::isinf(f)) return f > 0 ? "+Inf" : "-Inf";
  if (f == 0) return std::signbit(f) ? "-0" : "0";
  std::string digits; int x;
  shortest_digits(f, &digits, &x);
  const std::string sign = f < 0 ? "-" : "";
  return (x < -4 || x >= 6) ? fmt_e(sign, digits, x, 2) : fmt_f(sign, digits, x);
}
static std::string num_str(const V& v) {   // ast.Number.String() (values.py num_to_string)
  if (v.is_int) return i128_str(v.i);
  if (std::isfinite(v.d) && std::floor(v.d) == v.d && std::fabs(v.d) < 1e21) return i128_str((i128)v.d);
  return json_float_text(v.d);
}
static std::string to_string(const VP& v) {
  switch (v->k) {
    case V::Null: return "null";
    case V::Bool: return v->b ? "true" : "false";
    case V::Num: return num_str(*v);
    case V::Str: return quote(v->s);
    case V::Arr: { std::string o = "["; for (size_t i = 0; i < v->a.size(); i++) { if (i) o += ", "; o += to_string(v->a[i]); } return o + "]"; }
    case V::Obj: { std::string o = "{"; for (size_t i = 0; i < v->o.size(); i++) { if (i) o += ", "; o += to_string(v->o[i].first) + ": " + to_string(v->o[i].second); } return o + "}"; }
    case V::Set: { if (v->a.empty()) return "set()"; std::string o = "{"; for (size_t i = 0; i < v->a.size(); i++) { if (i) o += ", "; o += to_string(v->a[i]); } return o + "}"; }
  }
  return "";
}

// ================================================================================================ JSON reader
struct JsonErr : std::runtime_error { using std::runtime_error::runtime_error; };
class Json {
 public:
  Json(const char* p, size_t n) : p_(p), e_(p + n) {}
  VP parse() { ws(); VP v = value(0); ws(); if (p_ != e_) throw JsonErr("trailing characters"); return v; }

 private:
  const char *p_, *e_;
  void ws() { while (p_ < e_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) p_++; }
  VP value(int depth) {
    if (depth > 512) throw JsonErr("too deep");
    if (p_ >= e_) throw JsonErr("unexpected end");
    const char c = *p_;
    if (c == '{') {
      p_++; ws();
      std::vector<std::pair<VP, VP>> pairs;
      if (p_ < e_ && *p_ == '}') { p_++; return mk_obj(std::move(pairs)); }
      for (;;) {
        ws();
        if (p_ >= e_ || *p_ != '"') throw JsonErr("expected key");
        VP k = mk_str(str());
        ws();
        if (p_ >= e_ || *p_ != ':') throw JsonErr("expected ':'");
        p_++; ws();
        VP v = value(depth + 1);
        pairs.emplace_back(std::move(k), std::move(v));
        ws();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == '}') { p_++; break; }
        throw JsonErr("expected ',' or '}'");
      }
      return mk_obj(std::move(pairs));   // (json.loads: the last of two equal keys wins)
    }
    if (c == '[') {
      p_++; ws();
      std::vector<VP> items;
      if (p_ < e_ && *p_ == ']') { p_++; return mk_arr(std::move(items)); }
      for (;;) {
        ws();
        items.push_back(value(depth + 1));
        ws();
        if (p_ < e_ && *p_ == ',') { p_++; continue; }
        if (p_ < e_ && *p_ == ']') { p_++; break; }
        throw JsonErr("expected ',' or ']'");
      }
      return mk_arr(std::move(items));
    }
    if (c == '"') return mk_str(str());
    if (c == 't') { lit("true"); return mk_bool(true); }
    if (c == 'f') { lit("false"); return mk_bool(false); }
    if (c == 'n') { lit("null"); return mk_null(); }
    return number();
  }
  void lit(const char* w) { const size_t n = strlen(w); if ((size_t)(e_ - p_) < n || memcmp(p_, w, n) != 0) throw JsonErr("bad literal"); p_ += n; }
  VP number() {   // (json.loads: int without fraction / exponent, else float)
    const char* s = p_;
    bool is_int = true;
    if (p_ < e_ && *p_ == '-') p_++;
    if (p_ >= e_ || !(*p_ >= '0' && *p_ <= '9')) throw JsonErr("bad number");
    while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++;
    if (p_ < e_ && *p_ == '.') { is_int = false; p_++; while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++; }
    if (p_ < e_ && (*p_ == 'e' || *p_ == 'E')) { is_int = false; p_++; if (p_ < e_ && (*p_ == '+' || *p_ == '-')) p_++; while (p_ < e_ && *p_ >= '0' && *p_ <= '9') p_++; }
    const std::string t(s, p_ - s);
    if (is_int && t.size() <= 37) {
      i128 x = 0; size_t k = t[0] == '-' ? 1 : 0;
      for (; k < t.size(); k++) x = x * 10 + (t[k] - '0');
      return mk_int(t[0] == '-' ? -x : x);
    }
    V v; v.k = V::Num; v.is_int = false; v.d = strtod(t.c_str(), nullptr);   // (a Python float: no normalisation to int at read time; compare / equal are numeric)
    return std::make_shared<const V>(std::move(v));
  }
  static void utf8(std::string& o, uint32_t cp) {
    if (cp < 0x80) o.push_back((char)cp);
    else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
  }
  uint32_t hex4() {
    if (e_ - p_ < 4) throw JsonErr("bad escape");
    uint32_t v = 0;
    for (int k = 0; k < 4; k++) { const char c = *p_++; v <<= 4; if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10; else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10; else throw JsonErr("bad escape"); }
    return v;
  }
  std::string str() {
    p_++;
    std::string o;
    for (;;) {
      if (p_ >= e_) throw JsonErr("unterminated string");
      const char c = *p_++;
      if (c == '"') return o;
      if (c != '\\') { o.push_back(c); continue; }
      if (p_ >= e_) throw JsonErr("bad escape");
      const char x = *p_++;
      switch (x) {
        case '"': o.push_back('"'); break; case '\\': o.push_back('\\'); break; case '/': o.push_back('/'); break; case 'b': o.push_back('\b'); break;
        case 'f': o.push_back('\f'); break; case 'n': o.push_back('\n'); break; case 'r': o.push_back('\r'); break; case 't': o.push_back('\t'); break;
        case 'u': {
          uint32_t cp = hex4();
          if (cp >= 0xD800 && cp < 0xDC00 && e_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            const char* save = p_;
            p_ += 2;
            const uint32_t lo = hex4();
            if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); else p_ = save;
          }
          utf8(o, cp);
          break;
        }
        default: throw JsonErr("bad escape");
      }
    }
  }
};
static VP parse_json(const char* p, size_t n) { return Json(p, n).parse(); }

// ================================================================================================ Rego AST + parser (oracle/rego_parser.py)
struct RegoErr : std::runtime_error { using std::runtime_error::runtime_error; };
struct Term;
typedef std::shared_ptr<const Term> TP;
struct Literal;
typedef std::vector<Literal> Body;
struct Term {
  enum K { Scalar, Var, Ref, Call, Array, Object, SetT, ArrComp, SetComp, ObjComp, BinOp } k = Scalar;
  VP val;                          // Scalar
  std::string name;                // Var name; BinOp operator
  TP head, head2;                  // Ref head; comprehension head (key for ObjComp) / value
  std::vector<TP> args;            // Ref operands; Call args; Array / Set elements; Object k, v, k, v..; BinOp l, r
  std::vector<std::string> path;   // Call: dotted name
  std::shared_ptr<const Body> body;
};
struct Literal {
  enum K { Expr, Assign, Unify, Not, Some, SomeIn, Every } k = Expr;
  TP a, b, c;                      // Expr: a; Assign / Unify: a, b; SomeIn / Every: key a (may be null), value b, collection c
  std::vector<std::string> names;  // Some
  std::shared_ptr<const Literal> inner;
  std::shared_ptr<const Body> body;
};
struct Rule {
  enum K { Complete, SetR, ObjectR, Func } k = Complete;
  std::string name;
  std::vector<TP> args;
  TP key, value;
  Body body;
  bool is_default = false;
  std::vector<std::pair<TP, Body>> elses;
  std::vector<std::string> pkg;
  std::map<std::string, std::vector<std::string>> imports;   // alias -> path
};
struct Module { std::vector<std::string> package; std::vector<std::pair<std::vector<std::string>, std::string>> imports; std::vector<Rule> rules; };

struct Tok { enum K { NL, Num, Ident, Kw, Str, Op, Eof } k; std::string s; VP num; int line; };
static const std::set<std::string> KEYWORDS = {"package", "import", "default", "not", "some", "every", "in", "if", "contains", "else", "with", "as", "true", "false", "null"};
static std::vector<Tok> tokenize(const std::string& src) {
  std::vector<Tok> toks;
  size_t pos = 0;
  int line = 1;
  auto isid0 = [](char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; };
  auto isdig = [](char c) { return c >= '0' && c <= '9'; };
  while (pos < src.size()) {
    const char c = src[pos];
    if (c == ' ' || c == '\t' || c == '\r') { pos++; continue; }
    if (c == '#') { while (pos < src.size() && src[pos] != '\n') pos++; continue; }
    if (c == '\n') { toks.push_back({Tok::NL, "\n", nullptr, line}); line++; pos++; continue; }
    if (isdig(c) || (c == '.' && pos + 1 < src.size() && isdig(src[pos + 1]))) {
      size_t q = pos;
      bool flt = false;
      while (q < src.size() && isdig(src[q])) q++;
      if (q < src.size() && src[q] == '.' && q + 1 < src.size() && isdig(src[q + 1])) { flt = true; q++; while (q < src.size() && isdig(src[q])) q++; }
      if (q < src.size() && (src[q] == 'e' || src[q] == 'E')) {
        size_t r = q + 1;
        if (r < src.size() && (src[r] == '+' || src[r] == '-')) r++;
        if (r < src.size() && isdig(src[r])) { flt = true; q = r; while (q < src.size() && isdig(src[q])) q++; }
      }
      const std::string t = src.substr(pos, q - pos);
      VP v;
      if (flt) { V x; x.k = V::Num; x.is_int = false; x.d = strtod(t.c_str(), nullptr); v = std::make_shared<const V>(std::move(x)); }
      else { i128 x = 0; for (char d : t) x = x * 10 + (d - '0'); v = mk_int(x); }
      toks.push_back({Tok::Num, t, v, line});
      pos = q;
      continue;
    }
    if (isid0(c)) {
      size_t q = pos;
      while (q < src.size() && (isid0(src[q]) || isdig(src[q]))) q++;
      const std::string t = src.substr(pos, q - pos);
      toks.push_back({KEYWORDS.count(t) ? Tok::Kw : Tok::Ident, t, nullptr, line});
      pos = q;
      continue;
    }
    if (c == '"') {
      size_t q = pos + 1;
      std::string o;
      for (;;) {
        if (q >= src.size() || src[q] == '\n') throw RegoErr("line " + std::to_string(line) + ": unterminated string");
        const char d = src[q++];
        if (d == '"') break;
        if (d != '\\') { o.push_back(d); continue; }
        if (q >= src.size()) throw RegoErr("bad escape");
        const char e = src[q++];
        switch (e) {
          case '"': o.push_back('"'); break; case '\\': o.push_back('\\'); break; case '/': o.push_back('/'); break; case 'b': o.push_back('\b'); break;
          case 'f': o.push_back('\f'); break; case 'n': o.push_back('\n'); break; case 'r': o.push_back('\r'); break; case 't': o.push_back('\t'); break;
          case 'u': {
            if (q + 4 > src.size()) throw RegoErr("bad escape");
            const uint32_t cp = (uint32_t)strtoul(src.substr(q, 4).c_str(), nullptr, 16);
            q += 4;
            if (cp < 0x80) o.push_back((char)cp);
            else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
            else { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
            break;
          }
          default: throw RegoErr(std::string("bad escape \\") + e);
        }
      }
      toks.push_back({Tok::Str, o, nullptr, line});
      pos = q;
      continue;
    }
    if (c == '`') {
      const size_t q = src.find('`', pos + 1);
      if (q == std::string::npos) throw RegoErr("unterminated raw string");
      const std::string t = src.substr(pos + 1, q - pos - 1);
      toks.push_back({Tok::Str, t, nullptr, line});
      line += (int)std::count(t.begin(), t.end(), '\n');
      pos = q + 1;
      continue;
    }
    static const char* two[] = {":=", "==", "!=", "<=", ">="};
    bool done = false;
    for (const char* t : two) if (src.compare(pos, 2, t) == 0) { toks.push_back({Tok::Op, t, nullptr, line}); pos += 2; done = true; break; }
    if (done) continue;
    if (strchr("{}[]().,;:|=<>+-*/%&", c)) { toks.push_back({Tok::Op, std::string(1, c), nullptr, line}); pos++; continue; }
    throw RegoErr("line " + std::to_string(line) + ": unexpected character");
  }
  toks.push_back({Tok::Eof, "", nullptr, line});
  return toks;
}

class Parser {
 public:
  explicit Parser(const std::string& src) : toks_(tokenize(src)) {}
  Module parse_module() {
    Module m;
    skip_nl();
    expect(Tok::Kw, "package");
    m.package = parse_dotted();
    for (;;) {
      skip_nl();
      if (at(Tok::Eof)) break;
      if (accept(Tok::Kw, "import")) {
        std::vector<std::string> path = parse_dotted();
        std::string alias;
        if (accept(Tok::Kw, "as")) alias = expect(Tok::Ident).s;
        m.imports.emplace_back(path, alias);
        continue;
      }
      m.rules.push_back(parse_rule());
    }
    return m;
  }

 private:
  std::vector<Tok> toks_;
  size_t i_ = 0;
  int wild_ = 0;
  const Tok& peek(bool skip = false) const { size_t i = i_; if (skip) while (toks_[i].k == Tok::NL) i++; return toks_[i]; }
  const Tok& next(bool skip = false) { if (skip) skip_nl(); return toks_[i_++]; }
  void skip_nl() { while (toks_[i_].k == Tok::NL) i_++; }
  bool at(Tok::K k, const char* v = nullptr, bool skip = false) const { const Tok& t = peek(skip); return t.k == k && (!v || t.s == v); }
  bool at_op(const char* v, bool skip = false) const { return at(Tok::Op, v, skip); }
  bool accept(Tok::K k, const char* v = nullptr, bool skip = false) { if (at(k, v, skip)) { next(skip); return true; } return false; }
  const Tok& expect(Tok::K k, const char* v = nullptr, bool skip = false) {
    const Tok& t = next(skip);
    if (t.k != k || (v && t.s != v)) throw RegoErr("line " + std::to_string(t.line) + ": expected " + (v ? v : "token") + ", got " + t.s);
    return t;
  }
  [[noreturn]] void err(const std::string& m) const { throw RegoErr("line " + std::to_string(peek().line) + ": " + m + " (at " + peek().s + ")"); }
  std::vector<std::string> parse_dotted() {
    const Tok& t = next();
    if (t.k != Tok::Ident && t.k != Tok::Kw) throw RegoErr("expected identifier");
    std::vector<std::string> parts{t.s};
    for (;;) {
      if (accept(Tok::Op, ".")) parts.push_back(next().s);
      else if (at_op("[")) { next(); parts.push_back(expect(Tok::Str).s); expect(Tok::Op, "]"); }
      else break;
    }
    return parts;
  }
  Rule parse_rule() {
    Rule r;
    r.is_default = accept(Tok::Kw, "default");
    r.name = expect(Tok::Ident).s;
    if (at_op("(")) {
      next(); skip_nl();
      while (!at_op(")", true)) { r.args.push_back(parse_term()); if (!accept(Tok::Op, ",", true)) break; }
      expect(Tok::Op, ")", true);
      r.k = Rule::Func;
    } else if (at_op("[")) {
      next();
      r.key = parse_term();
      expect(Tok::Op, "]", true);
      r.k = Rule::SetR;
    } else if (accept(Tok::Kw, "contains")) {
      skip_nl();
      r.key = parse_term();
      r.k = Rule::SetR;
    }
    if (at_op("=") || at_op(":=")) { next(); r.value = parse_term(); if (r.k == Rule::SetR) r.k = Rule::ObjectR; }
    const bool has_if = accept(Tok::Kw, "if", at(Tok::Kw, "if", true));
    if (has_if) skip_nl();
    if (at_op("{")) r.body = parse_braced_body();
    else if (has_if) r.body.push_back(parse_literal());
    while (at(Tok::Kw, "else", true)) {
      next(true);
      TP val;
      if (at_op("=") || at_op(":=")) { next(); val = parse_term(); }
      accept(Tok::Kw, "if");
      Body b;
      if (at_op("{")) b = parse_braced_body();
      else if (at(Tok::NL) || at(Tok::Eof)) {}
      else b.push_back(parse_literal());
      r.elses.emplace_back(val, b);
    }
    if (r.is_default && !r.value) err("default rule needs a value");
    return r;
  }
  Body parse_braced_body() { expect(Tok::Op, "{"); Body b = parse_body_until("}"); expect(Tok::Op, "}", true); return b; }
  Body parse_body_until(const char* closer) {
    Body lits;
    for (;;) {
      skip_nl();
      while (accept(Tok::Op, ";")) skip_nl();
      if (at_op(closer)) break;
      lits.push_back(parse_literal());
      if (!(at(Tok::NL) || at_op(";") || at_op(closer))) err("expected end of literal");
    }
    return lits;
  }
  Literal parse_literal() {
    Literal l;
    if (accept(Tok::Kw, "not")) { l.k = Literal::Not; l.inner = std::make_shared<const Literal>(parse_literal()); return l; }
    if (at(Tok::Kw, "some")) {
      next();
      TP first = parse_term(false, true);
      if (at_op(",")) {
        next();
        TP second = parse_term(false, true);
        if (accept(Tok::Kw, "in")) { l.k = Literal::SomeIn; l.a = first; l.b = second; l.c = parse_term(); return l; }
        l.k = Literal::Some;
        l.names = {first->name, second->name};
        while (accept(Tok::Op, ",")) l.names.push_back(parse_term(false, true)->name);
        return l;
      }
      if (accept(Tok::Kw, "in")) { l.k = Literal::SomeIn; l.b = first; l.c = parse_term(); return l; }
      l.k = Literal::Some;
      l.names = {first->name};
      return l;
    }
    if (at(Tok::Kw, "every")) {
      next();
      TP first = parse_term(false, true), key;
      if (accept(Tok::Op, ",")) { key = first; first = parse_term(false, true); }
      expect(Tok::Kw, "in");
      l.k = Literal::Every; l.a = key; l.b = first; l.c = parse_term();
      l.body = std::make_shared<const Body>(parse_braced_body());
      return l;
    }
    TP lhs = parse_term();
    if (at_op(":=")) { next(); l.k = Literal::Assign; l.a = lhs; l.b = parse_term(false, false, true); }
    else if (at_op("=")) { next(); l.k = Literal::Unify; l.a = lhs; l.b = parse_term(false, false, true); }
    else { l.k = Literal::Expr; l.a = lhs; }
    if (at(Tok::Kw, "with")) err("`with` is not supported");
    return l;
  }
  static TP binop(const std::string& op, TP l, TP r) { Term t; t.k = Term::BinOp; t.name = op; t.args = {l, r}; return std::make_shared<const Term>(std::move(t)); }
  TP parse_term(bool no_bitor = false, bool no_in = false, bool after_op = false) { if (after_op) skip_nl(); return parse_relation(no_bitor, no_in); }
  TP parse_relation(bool no_bitor, bool no_in) {
    TP l = parse_bitor(no_bitor);
    for (;;) {
      const Tok& t = peek();
      if (t.k == Tok::Op && (t.s == "==" || t.s == "!=" || t.s == "<" || t.s == "<=" || t.s == ">" || t.s == ">=")) {
        const std::string op = t.s;
        next(); skip_nl();
        l = binop(op, l, parse_bitor(no_bitor));
      } else if (t.k == Tok::Kw && t.s == "in" && !no_in) { next(); l = binop("in", l, parse_bitor(no_bitor)); }
      else return l;
    }
  }
  TP parse_bitor(bool no_bitor) { TP l = parse_bitand(); while (!no_bitor && at_op("|")) { next(); skip_nl(); l = binop("|", l, parse_bitand()); } return l; }
  TP parse_bitand() { TP l = parse_arith(); while (at_op("&")) { next(); skip_nl(); l = binop("&", l, parse_arith()); } return l; }
  TP parse_arith() { TP l = parse_factor(); while (at_op("+") || at_op("-")) { const std::string op = next().s; skip_nl(); l = binop(op, l, parse_factor()); } return l; }
  TP parse_factor() { TP l = parse_unary(); while (at_op("*") || at_op("/") || at_op("%")) { const std::string op = next().s; skip_nl(); l = binop(op, l, parse_unary()); } return l; }
  static TP scalar(VP v) { Term t; t.k = Term::Scalar; t.val = std::move(v); return std::make_shared<const Term>(std::move(t)); }
  static TP var(const std::string& n) { Term t; t.k = Term::Var; t.name = n; return std::make_shared<const Term>(std::move(t)); }
  TP parse_unary() {
    if (at_op("-")) {
      next();
      TP t = parse_unary();
      if (t->k == Term::Scalar && t->val->k == V::Num) {
        if (t->val->is_int) return scalar(mk_int(-t->val->i));
        V x; x.k = V::Num; x.is_int = false; x.d = -t->val->d;
        return scalar(std::make_shared<const V>(std::move(x)));
      }
      return binop("-", scalar(mk_int(0)), t);
    }
    return parse_postfix(parse_primary());
  }
  TP parse_postfix(TP head) {
    std::vector<TP> ops;
    for (;;) {
      if (at_op(".")) { next(); const Tok& t = next(); if (t.k != Tok::Ident && t.k != Tok::Kw) throw RegoErr("expected field name"); ops.push_back(scalar(mk_str(t.s))); }
      else if (at_op("[")) { next(); skip_nl(); ops.push_back(parse_term()); expect(Tok::Op, "]", true); }
      else if (at_op("(")) {
        if (head->k != Term::Var) err("call on non-name");
        std::vector<std::string> path{head->name};
        for (auto& o : ops) { if (o->k != Term::Scalar || o->val->k != V::Str) err("call on non-name"); path.push_back(o->val->s); }
        next(); skip_nl();
        Term c; c.k = Term::Call; c.path = path;
        while (!at_op(")", true)) { c.args.push_back(parse_term()); if (!accept(Tok::Op, ",", true)) break; }
        expect(Tok::Op, ")", true);
        head = std::make_shared<const Term>(std::move(c));
        ops.clear();
      } else break;
    }
    if (!ops.empty()) { Term r; r.k = Term::Ref; r.head = head; r.args = ops; return std::make_shared<const Term>(std::move(r)); }
    return head;
  }
  TP parse_primary() {
    const Tok t = next();
    if (t.k == Tok::Num) return scalar(t.num);
    if (t.k == Tok::Str) return scalar(mk_str(t.s));
    if (t.k == Tok::Kw) {
      if (t.s == "true") return scalar(mk_bool(true));
      if (t.s == "false") return scalar(mk_bool(false));
      if (t.s == "null") return scalar(mk_null());
      if (t.s == "contains") return var(t.s);
      throw RegoErr("line " + std::to_string(t.line) + ": unexpected keyword " + t.s);
    }
    if (t.k == Tok::Ident) {
      if (t.s == "_") return var("$w" + std::to_string(++wild_));
      if (t.s == "set" && at_op("(")) {
        const size_t save = i_;
        next();
        if (accept(Tok::Op, ")")) { Term s; s.k = Term::SetT; return std::make_shared<const Term>(std::move(s)); }
        i_ = save;
      }
      return var(t.s);
    }
    if (t.k == Tok::Op) {
      if (t.s == "(") { skip_nl(); TP e = parse_term(); expect(Tok::Op, ")", true); return e; }
      if (t.s == "[") return parse_array_or_comp();
      if (t.s == "{") return parse_brace_term();
    }
    throw RegoErr("line " + std::to_string(t.line) + ": unexpected token " + t.s);
  }
  TP parse_array_or_comp() {
    skip_nl();
    Term a; a.k = Term::Array;
    if (accept(Tok::Op, "]")) return std::make_shared<const Term>(std::move(a));
    TP first = parse_term(true);
    if (at_op("|", true)) {
      next(true);
      Term c; c.k = Term::ArrComp; c.head = first; c.body = std::make_shared<const Body>(parse_body_until("]"));
      expect(Tok::Op, "]", true);
      return std::make_shared<const Term>(std::move(c));
    }
    a.args.push_back(first);
    while (accept(Tok::Op, ",", true)) { skip_nl(); if (at_op("]")) break; a.args.push_back(parse_term()); }
    expect(Tok::Op, "]", true);
    return std::make_shared<const Term>(std::move(a));
  }
  TP parse_brace_term() {
    skip_nl();
    if (accept(Tok::Op, "}")) { Term o; o.k = Term::Object; return std::make_shared<const Term>(std::move(o)); }
    TP first = parse_term(true);
    if (at_op(":", true)) {
      next(true); skip_nl();
      TP val = parse_term(true);
      if (at_op("|", true)) {
        next(true);
        Term c; c.k = Term::ObjComp; c.head = first; c.head2 = val; c.body = std::make_shared<const Body>(parse_body_until("}"));
        expect(Tok::Op, "}", true);
        return std::make_shared<const Term>(std::move(c));
      }
      Term o; o.k = Term::Object; o.args = {first, val};
      while (accept(Tok::Op, ",", true)) {
        skip_nl();
        if (at_op("}")) break;
        TP k = parse_term();
        expect(Tok::Op, ":", true); skip_nl();
        o.args.push_back(k); o.args.push_back(parse_term());
      }
      expect(Tok::Op, "}", true);
      return std::make_shared<const Term>(std::move(o));
    }
    if (at_op("|", true)) {
      next(true);
      Term c; c.k = Term::SetComp; c.head = first; c.body = std::make_shared<const Body>(parse_body_until("}"));
      expect(Tok::Op, "}", true);
      return std::make_shared<const Term>(std::move(c));
    }
    Term s; s.k = Term::SetT; s.args.push_back(first);
    while (accept(Tok::Op, ",", true)) { skip_nl(); if (at_op("}")) break; s.args.push_back(parse_term()); }
    expect(Tok::Op, "}", true);
    return std::make_shared<const Term>(std::move(s));
  }
};

// ================================================================================================ builtins (oracle/rego_builtins.py, the ones the bench's policies call)
struct BuiltinErr {};   // the call is undefined
static const V& need(const VP& v, V::K k) { if (v->k != k) throw BuiltinErr(); return *v; }
static VP arith(const std::string& op, const VP& a, const VP& b) {
  if (op == "-" && a->k == V::Set && b->k == V::Set) { std::vector<VP> o; for (auto& x : a->a) if (!set_has(*b, x)) o.push_back(x); return mk_set(std::move(o)); }
  if (op == "&") { need(a, V::Set); need(b, V::Set); std::vector<VP> o; for (auto& x : a->a) if (set_has(*b, x)) o.push_back(x); return mk_set(std::move(o)); }
  if (op == "|") { need(a, V::Set); need(b, V::Set); std::vector<VP> o = a->a; o.insert(o.end(), b->a.begin(), b->a.end()); return mk_set(std::move(o)); }
  const V& x = need(a, V::Num); const V& y = need(b, V::Num);
  const bool ii = x.is_int && y.is_int;
  const double dx = x.is_int ? (double)x.i : x.d, dy = y.is_int ? (double)y.i : y.d;
  if (op == "+") return ii ? mk_int(x.i + y.i) : mk_float(dx + dy);
  if (op == "-") return ii ? mk_int(x.i - y.i) : mk_float(dx - dy);
  if (op == "*") return ii ? mk_int(x.i * y.i) : mk_float(dx * dy);
  if (op == "/") {
    if (dy == 0) throw BuiltinErr();
    if (ii && x.i % y.i == 0) return mk_int(x.i / y.i);
    return mk_float(dx / dy);
  }
  if (op == "%") {
    if ((!x.is_int && std::floor(x.d) != x.d) || (!y.is_int && std::floor(y.d) != y.d)) throw BuiltinErr();
    const i128 p = x.is_int ? x.i : (i128)x.d, q = y.is_int ? y.i : (i128)y.d;
    if (q == 0) throw BuiltinErr();
    const i128 r = (p < 0 ? -p : p) % (q < 0 ? -q : q);
    return mk_int(p < 0 ? -r : r);
  }
  throw BuiltinErr();
}
// builtinSprintf (rego_builtins.py go_sprintf) for the verbs messages are made of -- %v %s %d, flags and widths; any other verb is
// outside this checker's scope (an error, not a guess)
static std::string go_sprintf(const std::string& fmt, const std::vector<VP>& args) {
  struct Arg { int kind; std::string s; i128 i; double d; };   // 0 string, 1 int, 2 float64
  std::vector<Arg> as;
  for (const VP& a : args) {
    Arg x{0, "", 0, 0};
    if (a->k == V::Num) {
      if (a->is_int) { x.kind = 1; x.i = a->i; }
      else if (std::floor(a->d) == a->d && std::fabs(a->d) < 1e21) { x.kind = 1; x.i = (i128)a->d; }
      else { x.kind = 2; x.d = a->d; }
    } else if (a->k == V::Str) x.s = a->s;
    else x.s = to_string(a);
    as.push_back(x);
  }
  auto pad = [](std::string s, const std::string& flags, int width) {
    const size_t n = utf8_len(s);
    if (width < 0 || n >= (size_t)width) return s;
    if (flags.find('-') != std::string::npos) return s + std::string((size_t)width - n, ' ');
    if (flags.find('0') != std::string::npos && !s.empty() && ((s[0] >= '0' && s[0] <= '9') || s[0] == '+' || s[0] == '-')) {
      std::string sign;
      if (s[0] == '+' || s[0] == '-') { sign = s.substr(0, 1); s = s.substr(1); }
      return sign + std::string((size_t)width - n, '0') + s;
    }
    return std::string((size_t)width - n, ' ') + s;
  };
  std::string o;
  size_t ai = 0;
  for (size_t i = 0; i < fmt.size(); i++) {
    if (fmt[i] != '%') { o.push_back(fmt[i]); continue; }
    size_t j = i + 1;
    std::string flags;
    while (j < fmt.size() && strchr("+-# 0", fmt[j])) flags.push_back(fmt[j++]);
    int width = -1, prec = -1;
    bool too_large = false;   // (fmt parsenum: beyond 1e6 with a further digit to come, the directive and the rest of the format are %!(NOVERB))
    if (j < fmt.size() && fmt[j] >= '0' && fmt[j] <= '9') { width = 0; while (j < fmt.size() && fmt[j] >= '0' && fmt[j] <= '9') { if (width > 1000000) { too_large = true; break; } width = width * 10 + (fmt[j++] - '0'); } }
    if (!too_large && j < fmt.size() && fmt[j] == '.') { j++; prec = 0; while (j < fmt.size() && fmt[j] >= '0' && fmt[j] <= '9') { if (prec > 1000000) { too_large = true; break; } prec = prec * 10 + (fmt[j++] - '0'); } }
    if (too_large || j >= fmt.size()) { o += "%!(NOVERB)"; break; }
    std::string verb_rune(1, fmt[j]);   // the verb is the next rune, whatever it is
    while (j + 1 < fmt.size() && (unsigned char)verb_rune[0] >= 0x80 && ((unsigned char)fmt[j + 1] & 0xC0) == 0x80) verb_rune.push_back(fmt[++j]);
    const char verb = verb_rune.size() == 1 ? verb_rune[0] : '\0';
    i = j;
    if (verb == '%') { o.push_back('%'); continue; }
    if (ai >= as.size()) { o += "%!" + verb_rune + "(MISSING)"; continue; }
    const Arg& a = as[ai++];
    std::string s;
    const bool plus = flags.find('+') != std::string::npos, space = flags.find(' ') != std::string::npos, left = flags.find('-') != std::string::npos;
    const bool zero = flags.find('0') != std::string::npos && !left;
    // an integer operand (fmtInteger): precision or, with the 0 flag, the width = minimum digits; then the sign; spaces fill the width
    auto integer = [&]() {
      const bool neg = a.i < 0;
      std::string d = i128_str(neg ? -a.i : a.i);
      size_t min_digits = 0;
      if (prec >= 0) { min_digits = (size_t)prec; if (prec == 0 && a.i == 0) d.clear(); }
      else if (zero && width > 0) min_digits = (size_t)width - ((neg || plus || space) && width > 0 ? 1 : 0);
      while (d.size() < min_digits) d.insert(d.begin(), '0');
      d = (neg ? "-" : plus ? "+" : space ? " " : "") + d;
      if (width > 0 && d.size() < (size_t)width) { const std::string fill((size_t)width - d.size(), ' '); d = left ? d + fill : fill + d; }
      return d;
    };
    // a float64 operand (fmtFloat): sign by flag, zeros go between the sign and the digits
    auto floating = [&](std::string t) {
      if (t[0] != '-') t = (plus ? "+" : space ? " " : "") + t;
      if (width <= 0 || t.size() >= (size_t)width) return t;
      const std::string fill((size_t)width - t.size(), zero ? '0' : ' ');
      if (left) return t + fill;
      if (zero && (t[0] == '-' || t[0] == '+' || t[0] == ' ')) return t.substr(0, 1) + fill + t.substr(1);
      return fill + t;
    };
    auto cut = [&](const std::string& t) {   // fmtS / fmtQ: the precision counts runes
      if (prec < 0) return t;
      size_t b = 0, n = 0;
      while (b < t.size() && n < (size_t)prec) { b++; while (b < t.size() && ((unsigned char)t[b] & 0xC0) == 0x80) b++; n++; }
      return t.substr(0, b);
    };
    // the operand under %v with this directive's flags, width and precision; a float64 with a precision prints that many significant digits (%g)
    auto as_v = [&]() {
      if (a.kind == 0) return pad(cut(a.s), flags, width);
      if (a.kind == 1) return integer();
      if (prec < 0) return floating(go_float_v(a.d));
      char buf[64];
      snprintf(buf, sizeof buf, "%.*g", prec, a.d);
      return floating(buf);
    };
    // badVerb: %!verb(type=value), the value printed as %v under the same flags
    auto bad = [&]() { return "%!" + verb_rune + (a.kind == 0 ? "(string=" : a.kind == 1 ? "(int=" : "(float64=") + as_v() + ")"; };
    if (verb == 'v') s = as_v();
    else if (verb == 's') s = a.kind == 0 ? as_v() : bad();
    else if (verb == 'd') s = a.kind == 1 ? integer() : bad();
    else if (verb == 'q') s = a.kind == 0 ? pad(quote(cut(a.s)), flags, width) : bad();
    else if (verb == 'T') { std::string t = a.kind == 0 ? "string" : a.kind == 1 ? "int" : "float64"; if (prec >= 0 && (size_t)prec < t.size()) t.resize((size_t)prec); s = pad(t, flags, width); }
    else if (verb == '\0' || verb == 't' || verb == 'p' || !strchr("bcdeEfFgGoOqsUvxX", verb)) s = bad();   // no verb of fmt at all, or none for these operands
    else throw std::runtime_error(std::string("sprintf verb %") + verb + " is outside this checker's scope");
    o += s;   // (a bad verb is written outside the width)
  }
  if (ai < as.size()) {
    o += "%!(EXTRA ";
    for (size_t k = ai; k < as.size(); k++) { if (k > ai) o += ", "; o += as[k].kind == 0 ? "string=" + as[k].s : as[k].kind == 1 ? "int=" + i128_str(as[k].i) : "float64=" + go_float_v(as[k].d); }
    o += ")";
  }
  return o;
}
// Go regexp (RE2 syntax) through std::regex (ECMAScript): the policies' patterns -- anchors, classes, alternation, counted
// repetition, a leading (?i) -- mean the same in both.  What this translation does not cover is an ERROR of the checker (never
// "undefined", which is what an invalid pattern is in OPA): other inline flags, \pL classes, \z, named groups, a pattern std::regex
// rejects.
static std::shared_ptr<const std::regex> go_regex(const std::string& pat) {
  static std::mutex mu;
  static std::map<std::string, std::shared_ptr<const std::regex>> cache;
  std::lock_guard<std::mutex> l(mu);
  auto it = cache.find(pat);
  if (it != cache.end()) return it->second;
  std::string body = pat;
  auto flags = std::regex::ECMAScript;
  if (body.compare(0, 2, "(?") == 0) {
    const size_t close = body.find(')');
    if (close != std::string::npos && body.find_first_not_of("imsU-", 2) == close) {   // a flags-only group at the very start
      if (body.substr(2, close - 2) != "i") throw std::runtime_error("regex flags " + body.substr(0, close + 1) + " are outside this checker's scope");
      flags |= std::regex::icase;
      body = body.substr(close + 1);
    }
  }
  if (body.find("(?") != std::string::npos || body.find("\\p") != std::string::npos || body.find("\\z") != std::string::npos || body.find("\\A") != std::string::npos || body.find("\\Q") != std::string::npos || body.find("[[:") != std::string::npos)
    throw std::runtime_error("regex " + pat + " is outside this checker's scope");
  std::shared_ptr<const std::regex> r;
  try { r = std::make_shared<const std::regex>(body, flags); }
  catch (const std::regex_error&) { throw std::runtime_error("regex " + pat + " is outside this checker's scope (std::regex rejects it)"); }
  cache[pat] = r;
  return r;
}
typedef VP (*BuiltinFn)(const std::vector<VP>&);
static const std::map<std::string, std::pair<int, BuiltinFn>>& BUILTINS() {
  static const std::map<std::string, std::pair<int, BuiltinFn>> m = {
      {"count", {1, [](const std::vector<VP>& a) -> VP { const V& x = *a[0]; if (x.k == V::Str) return mk_int((i128)utf8_len(x.s)); if (x.k == V::Arr || x.k == V::Set) return mk_int((i128)x.a.size()); if (x.k == V::Obj) return mk_int((i128)x.o.size()); throw BuiltinErr(); }}},
      {"any", {1, [](const std::vector<VP>& a) -> VP { if (a[0]->k != V::Arr && a[0]->k != V::Set) throw BuiltinErr(); for (auto& x : a[0]->a) if (x->k == V::Bool && x->b) return mk_bool(true); return mk_bool(false); }}},
      {"all", {1, [](const std::vector<VP>& a) -> VP { if (a[0]->k != V::Arr && a[0]->k != V::Set) throw BuiltinErr(); for (auto& x : a[0]->a) if (!(x->k == V::Bool && x->b)) return mk_bool(false); return mk_bool(true); }}},
      {"startswith", {2, [](const std::vector<VP>& a) -> VP { const std::string& s = need(a[0], V::Str).s; const std::string& p = need(a[1], V::Str).s; return mk_bool(s.size() >= p.size() && s.compare(0, p.size(), p) == 0); }}},
      {"endswith", {2, [](const std::vector<VP>& a) -> VP { const std::string& s = need(a[0], V::Str).s; const std::string& p = need(a[1], V::Str).s; return mk_bool(s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0); }}},
      {"contains", {2, [](const std::vector<VP>& a) -> VP { return mk_bool(need(a[0], V::Str).s.find(need(a[1], V::Str).s) != std::string::npos); }}},
      {"is_string", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Str); }}},
      {"is_number", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Num); }}},
      {"is_boolean", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Bool); }}},
      {"is_array", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Arr); }}},
      {"is_object", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Obj); }}},
      {"is_set", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Set); }}},
      {"is_null", {1, [](const std::vector<VP>& a) -> VP { return mk_bool(a[0]->k == V::Null); }}},
      {"sprintf", {2, [](const std::vector<VP>& a) -> VP { return mk_str(go_sprintf(need(a[0], V::Str).s, need(a[1], V::Arr).a)); }}},
      {"re_match", {2, [](const std::vector<VP>& a) -> VP { const std::string& p = need(a[0], V::Str).s; const std::string& s = need(a[1], V::Str).s; return mk_bool(std::regex_search(s, *go_regex(p))); }}},
      {"regex.match", {2, [](const std::vector<VP>& a) -> VP { const std::string& p = need(a[0], V::Str).s; const std::string& s = need(a[1], V::Str).s; return mk_bool(std::regex_search(s, *go_regex(p))); }}},
      {"replace", {3, [](const std::vector<VP>& a) -> VP {
         const std::string& s = need(a[0], V::Str).s; const std::string& old = need(a[1], V::Str).s; const std::string& nw = need(a[2], V::Str).s;
         std::string o;
         if (old.empty()) {   // str.replace("", new): between every two code points and at both ends
           o = nw;
           for (size_t i = 0; i < s.size();) { size_t j = i + 1; while (j < s.size() && ((unsigned char)s[j] & 0xC0) == 0x80) j++; o.append(s, i, j - i); o += nw; i = j; }
           return mk_str(o);
         }
         for (size_t i = 0; i < s.size();) { if (s.compare(i, old.size(), old) == 0) { o += nw; i += old.size(); } else o.push_back(s[i++]); }
         return mk_str(o);
       }}},
      {"split", {2, [](const std::vector<VP>& a) -> VP {
         const std::string& s = need(a[0], V::Str).s; const std::string& d = need(a[1], V::Str).s;
         std::vector<VP> o;
         if (d.empty()) { for (size_t i = 0; i < s.size();) { size_t j = i + 1; while (j < s.size() && ((unsigned char)s[j] & 0xC0) == 0x80) j++; o.push_back(mk_str(s.substr(i, j - i))); i = j; } return mk_arr(std::move(o)); }
         size_t pos = 0;
         for (;;) { const size_t q = s.find(d, pos); if (q == std::string::npos) { o.push_back(mk_str(s.substr(pos))); break; } o.push_back(mk_str(s.substr(pos, q - pos))); pos = q + d.size(); }
         return mk_arr(std::move(o));
       }}},
      {"substring", {3, [](const std::vector<VP>& a) -> VP {   // by code point
         const std::string& s = need(a[0], V::Str).s; const V& off = need(a[1], V::Num); const V& ln = need(a[2], V::Num);
         const long long o = off.is_int ? (long long)off.i : (long long)off.d, l = ln.is_int ? (long long)ln.i : (long long)ln.d;
         if (o < 0) throw BuiltinErr();
         std::vector<size_t> starts;
         for (size_t i = 0; i < s.size(); i++) if (((unsigned char)s[i] & 0xC0) != 0x80) starts.push_back(i);
         if ((size_t)o >= starts.size()) return mk_str("");
         const size_t b = starts[(size_t)o];
         if (l < 0 || (size_t)(o + l) >= starts.size()) return mk_str(s.substr(b));
         return mk_str(s.substr(b, starts[(size_t)(o + l)] - b));
       }}},
      {"trim", {2, [](const std::vector<VP>& a) -> VP {   // (ASCII cut sets, as the policies use)
         const std::string& s = need(a[0], V::Str).s; const std::string& cut = need(a[1], V::Str).s;
         if (cut.empty()) return a[0];
         size_t lo = 0, hi = s.size();
         while (lo < hi && cut.find(s[lo]) != std::string::npos) lo++;
         while (hi > lo && cut.find(s[hi - 1]) != std::string::npos) hi--;
         return mk_str(s.substr(lo, hi - lo));
       }}},
      {"trim_prefix", {2, [](const std::vector<VP>& a) -> VP { const std::string& s = need(a[0], V::Str).s; const std::string& p = need(a[1], V::Str).s; return s.size() >= p.size() && s.compare(0, p.size(), p) == 0 ? mk_str(s.substr(p.size())) : a[0]; }}},
      {"trim_suffix", {2, [](const std::vector<VP>& a) -> VP { const std::string& s = need(a[0], V::Str).s; const std::string& p = need(a[1], V::Str).s; return !p.empty() && s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0 ? mk_str(s.substr(0, s.size() - p.size())) : a[0]; }}},
      {"lower", {1, [](const std::vector<VP>& a) -> VP { std::string s = need(a[0], V::Str).s; for (char& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c + 32); for (unsigned char c : s) if (c >= 0x80) throw std::runtime_error("lower: non-ASCII text is outside this checker's scope"); return mk_str(s); }}},
      {"upper", {1, [](const std::vector<VP>& a) -> VP { std::string s = need(a[0], V::Str).s; for (char& c : s) if (c >= 'a' && c <= 'z') c = (char)(c - 32); for (unsigned char c : s) if (c >= 0x80) throw std::runtime_error("upper: non-ASCII text is outside this checker's scope"); return mk_str(s); }}},
      {"concat", {2, [](const std::vector<VP>& a) -> VP {
         const std::string& d = need(a[0], V::Str).s;
         if (a[1]->k != V::Arr && a[1]->k != V::Set) throw BuiltinErr();
         std::string o;
         for (size_t i = 0; i < a[1]->a.size(); i++) { if (i) o += d; o += need(a[1]->a[i], V::Str).s; }
         return mk_str(o);
       }}},
      {"to_number", {1, [](const std::vector<VP>& a) -> VP {
         const V& x = *a[0];
         if (x.k == V::Null) return mk_int(0);
         if (x.k == V::Bool) return mk_int(x.b ? 1 : 0);
         if (x.k == V::Num) return a[0];
         if (x.k != V::Str) throw BuiltinErr();
         const std::string& s = x.s;   // [+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?|0[xX][0-9a-fA-F]+|[iI]nf|NaN)
         size_t i = 0;
         if (i < s.size() && (s[i] == '+' || s[i] == '-')) i++;
         const std::string body = s.substr(i);
         auto dig = [](char c) { return c >= '0' && c <= '9'; };
         bool ok = false, is_int = false;
         if (body == "inf" || body == "Inf" || body == "NaN") ok = true;
         else if (body.size() > 2 && body[0] == '0' && (body[1] == 'x' || body[1] == 'X')) { ok = true; for (size_t k = 2; k < body.size(); k++) if (!isxdigit((unsigned char)body[k])) ok = false; }
         else {
           size_t k = 0, nd = 0;
           while (k < body.size() && dig(body[k])) { k++; nd++; }
           bool frac = false;
           if (k < body.size() && body[k] == '.') { frac = true; k++; size_t fd = 0; while (k < body.size() && dig(body[k])) { k++; fd++; } if (nd == 0 && fd == 0) nd = 0; else if (nd == 0) nd = fd ? 1 : 0; }
           bool ex = false;
           if (nd && k < body.size() && (body[k] == 'e' || body[k] == 'E')) { size_t r = k + 1; if (r < body.size() && (body[r] == '+' || body[r] == '-')) r++; size_t ed = 0; while (r < body.size() && dig(body[r])) { r++; ed++; } if (ed) { ex = true; k = r; } }
           ok = nd > 0 && k == body.size();
           is_int = ok && !frac && !ex;
         }
         if (!ok) throw BuiltinErr();
         if (is_int && s.size() <= 37) { i128 v = 0; for (char c : body) v = v * 10 + (c - '0'); return mk_int(s[0] == '-' ? -v : v); }
         if (body.size() > 2 && (body[1] == 'x' || body[1] == 'X')) throw BuiltinErr();   // (int("0x..") and float("0x..") both fail in the Python oracle)
         const double d = strtod(s.c_str(), nullptr);
         return mk_float(d);
       }}},
  };
  return m;
}

// ================================================================================================ interpreter (oracle/rego_interp.py)
struct Unbound {};   // a variable used where it cannot be bound yet
struct EvalErr : std::runtime_error { using std::runtime_error::runtime_error; };
template <class Sig> class fref;
template <class R, class... A>
class fref<R(A...)> {
  void* obj_;
  R (*call_)(void*, A...);

 public:
  template <class F, class = typename std::enable_if<!std::is_same<typename std::decay<F>::type, fref>::value>::type>
  fref(F&& f) : obj_((void*)std::addressof(f)), call_([](void* o, A... a) -> R { return (*(typename std::remove_reference<F>::type*)o)(static_cast<A>(a)...); }) {}
  R operator()(A... a) const { return call_(obj_, static_cast<A>(a)...); }
};
// environments: an immutable chain, a binding is a new link (what dict(env) + assignment is in the Python oracle)
struct Env;
typedef std::shared_ptr<const Env> EP;
struct Env { const std::string* name; VP v; bool gone; EP up; };
static EP bind(const EP& e, const std::string& name, const VP& v) { return std::make_shared<const Env>(Env{&name, v, false, e}); }
static EP unbind(const EP& e, const std::string& name) { return std::make_shared<const Env>(Env{&name, nullptr, true, e}); }
static const VP* lookup(const EP& e, const std::string& name) {
  for (const Env* p = e.get(); p; p = p->up.get()) if (p->name == &name || *p->name == name) return p->gone ? nullptr : &p->v;
  return nullptr;
}
typedef fref<void(const EP&)> KE;
typedef fref<void(const VP&, const EP&)> KV;

struct Program {   // one compiled template: main module + libs (Interp.__init__)
  std::vector<std::shared_ptr<Module>> modules;
  std::map<std::pair<std::vector<std::string>, std::string>, std::vector<const Rule*>> rules;
  std::vector<std::string> main_pkg;
  VP data;
  void load(const std::vector<std::string>& sources) {
    for (auto& s : sources) modules.push_back(std::make_shared<Module>(Parser(s).parse_module()));
    for (auto& m : modules)
      for (Rule& r : m->rules) {
        r.pkg = m->package;
        for (auto& im : m->imports) r.imports[im.second.empty() ? im.first.back() : im.second] = im.first;
      }
    for (auto& m : modules) for (const Rule& r : m->rules) rules[{r.pkg, r.name}].push_back(&r);
    main_pkg = modules[0]->package;
    data = mk_obj({});
  }
  const std::vector<const Rule*>* find(const std::vector<std::string>& pkg, const std::string& name) const { auto it = rules.find({pkg, name}); return it == rules.end() ? nullptr : &it->second; }
};

class Query {
 public:
  Query(const Program& p, VP input) : P(p), input_(std::move(input)) {}
  // the `violation` partial set of the main package (Interp.violations)
  VP violations() {
    if (!P.find(P.main_pkg, "violation")) return mk_set({});
    VP s = rule_value(P.main_pkg, "violation");
    return s ? s : mk_set({});
  }

 private:
  const Program& P;
  VP input_;
  struct Cached { bool in_progress = false, done = false; VP v; };
  std::map<std::pair<std::vector<std::string>, std::string>, Cached> cache_;
  int depth_ = 0;

  VP rule_value(const std::vector<std::string>& pkg, const std::string& name) {   // nullptr: undefined
    Cached& c = cache_[{pkg, name}];
    if (c.done) return c.v;
    if (c.in_progress) throw EvalErr("recursive rule " + name);
    c.in_progress = true;
    const std::vector<const Rule*>& rules = *P.find(pkg, name);
    const Rule::K kind = rules[0]->k;
    VP res;
    if (kind == Rule::Func) throw EvalErr("function " + name + " referenced without call");
    if (kind == Rule::SetR) {
      std::vector<VP> out;
      for (const Rule* r : rules) eval_body(r->body, nullptr, *r, [&](const EP& env) { eval_term(*r->key, env, *r, [&](const VP& v, const EP&) { out.push_back(v); }); });
      res = mk_set(std::move(out));
    } else if (kind == Rule::ObjectR) {
      std::vector<std::pair<VP, VP>> pairs;
      for (const Rule* r : rules)
        eval_body(r->body, nullptr, *r, [&](const EP& env) { eval_term(*r->key, env, *r, [&](const VP& k, const EP& e2) { eval_term(*r->value, e2, *r, [&](const VP& v, const EP&) { pairs.emplace_back(k, v); }); }); });
      res = mk_obj(std::move(pairs));
    } else {
      VP def;
      for (const Rule* r : rules) {
        if (r->is_default) { eval_term(*r->value, nullptr, *r, [&](const VP& v, const EP&) { def = v; }); continue; }
        VP v = complete_def(*r, nullptr);
        if (v) { if (res && !equal(res, v)) throw EvalErr("complete rule " + name + " produced conflicting values"); res = v; }
      }
      if (!res) res = def;
    }
    Cached& c2 = cache_[{pkg, name}];
    c2.v = res; c2.done = true; c2.in_progress = false;
    return res;
  }
  VP complete_def(const Rule& r, const EP& env) {   // one definition with its else chain; nullptr: undefined
    for (size_t li = 0; li <= r.elses.size(); li++) {
      const TP& val_t = li == 0 ? r.value : r.elses[li - 1].first;
      const Body& body = li == 0 ? r.body : r.elses[li - 1].second;
      VP res;
      eval_body(body, env, r, [&](const EP& e) {
        VP v;
        if (!val_t) v = mk_bool(true);
        else { bool first = true; eval_term(*val_t, e, r, [&](const VP& x, const EP&) { if (first) { v = x; first = false; } }); if (!v) return; }
        if (res && !equal(res, v)) throw EvalErr("rule " + r.name + " produced conflicting values");
        res = v;
      });
      if (res) return res;
    }
    return nullptr;
  }
  VP call_function(const std::vector<std::string>& pkg, const std::string& name, const std::vector<VP>& args) {
    const std::vector<const Rule*>& rules = *P.find(pkg, name);
    if (++depth_ > 200) { depth_--; throw EvalErr("recursion too deep in " + name); }
    struct Dec { int& d; ~Dec() { d--; } } dec{depth_};
    VP res, def;
    for (const Rule* r : rules) {
      if (r->k != Rule::Func || r->args.size() != args.size()) continue;
      if (r->is_default) { eval_term(*r->value, nullptr, *r, [&](const VP& v, const EP&) { def = v; }); continue; }
      unify_args(*r, args, 0, nullptr, [&](const EP& e) {
        VP v = complete_def(*r, e);
        if (v) { if (res && !equal(res, v)) throw EvalErr("function " + name + " produced conflicting outputs"); res = v; }
      });
    }
    return res ? res : def;
  }
  void unify_args(const Rule& r, const std::vector<VP>& args, size_t i, const EP& env, KE k) {
    if (i == args.size()) { k(env); return; }
    unify_value(*r.args[i], args[i], env, r, [&](const EP& e) { unify_args(r, args, i + 1, e, k); });
  }

  // ---- bodies: the first literal that can be evaluated goes first (eval_body)
  void eval_body(const Body& lits, const EP& env, const Rule& rule, KE k) {
    std::vector<const Literal*> ptrs;
    for (auto& l : lits) ptrs.push_back(&l);
    eval_lits(ptrs, env, rule, k);
  }
  void eval_lits(const std::vector<const Literal*>& lits, const EP& env, const Rule& rule, KE k) {
    if (lits.empty()) { k(env); return; }
    for (size_t i = 0; i < lits.size(); i++) {
      std::vector<EP> sols;
      try { eval_literal(*lits[i], env, rule, [&](const EP& e) { sols.push_back(e); }); }
      catch (const Unbound&) { continue; }
      std::vector<const Literal*> rest;
      for (size_t j = 0; j < lits.size(); j++) if (j != i) rest.push_back(lits[j]);
      for (const EP& e : sols) eval_lits(rest, e, rule, k);
      return;
    }
    throw Unbound();
  }
  void eval_literal(const Literal& l, const EP& env, const Rule& rule, KE k) {
    switch (l.k) {
      case Literal::Expr: eval_term(*l.a, env, rule, [&](const VP& v, const EP& e) { if (!(v->k == V::Bool && !v->b)) k(e); }); break;
      case Literal::Assign: case Literal::Unify: unify_terms(*l.a, *l.b, env, rule, k); break;
      case Literal::Not: { bool any = false; eval_literal(*l.inner, env, rule, [&](const EP&) { any = true; }); if (!any) k(env); break; }
      case Literal::Some: { EP e = env; for (auto& n : l.names) e = unbind(e, n); k(e); break; }
      case Literal::SomeIn:
        eval_term(*l.c, env, rule, [&](const VP& coll, const EP& e) {
          iter_kv(coll, [&](const VP& key, const VP& val) {
            unify_value(*l.b, val, e, rule, [&](const EP& e2) { if (!l.a) k(e2); else unify_value(*l.a, key, e2, rule, k); });
          });
        });
        break;
      case Literal::Every:
        eval_term(*l.c, env, rule, [&](const VP& coll, const EP& e) {
          bool ok = true;
          iter_kv(coll, [&](const VP& key, const VP& val) {
            if (!ok) return;
            bool sat = false;
            unify_value(*l.b, val, e, rule, [&](const EP& e2) {
              auto run = [&](const EP& e3) { if (!sat) eval_body(*l.body, e3, rule, [&](const EP&) { sat = true; }); };
              if (!l.a) run(e2); else unify_value(*l.a, key, e2, rule, run);
            });
            if (!sat) ok = false;
          });
          if (ok) k(e);
        });
        break;
    }
  }

  // ---- unification
  bool is_global(const std::string& name, const Rule& rule) const { return name == "input" || name == "data" || P.find(rule.pkg, name) || rule.imports.count(name); }
  bool is_unbound_var(const Term& t, const EP& env, const Rule& rule) const { return t.k == Term::Var && !lookup(env, t.name) && !is_global(t.name, rule); }
  bool has_unbound(const Term& t, const EP& env, const Rule& rule) const {
    if (t.k == Term::Var) return is_unbound_var(t, env, rule);
    if (t.k == Term::Array) { for (auto& x : t.args) if (has_unbound(*x, env, rule)) return true; return false; }
    if (t.k == Term::Object) { for (size_t i = 1; i < t.args.size(); i += 2) if (has_unbound(*t.args[i], env, rule)) return true; return false; }   // (values only, as the Python oracle)
    return false;
  }
  void unify_terms(const Term& a, const Term& b, const EP& env, const Rule& rule, KE k) {
    if (is_unbound_var(a, env, rule)) eval_term(b, env, rule, [&](const VP& v, const EP& e) { k(bind(e, a.name, v)); });
    else if (is_unbound_var(b, env, rule)) eval_term(a, env, rule, [&](const VP& v, const EP& e) { k(bind(e, b.name, v)); });
    else if ((a.k == Term::Array || a.k == Term::Object) && has_unbound(a, env, rule)) eval_term(b, env, rule, [&](const VP& v, const EP& e) { unify_value(a, v, e, rule, k); });
    else if ((b.k == Term::Array || b.k == Term::Object) && has_unbound(b, env, rule)) eval_term(a, env, rule, [&](const VP& v, const EP& e) { unify_value(b, v, e, rule, k); });
    else eval_term(a, env, rule, [&](const VP& va, const EP& e) { eval_term(b, e, rule, [&](const VP& vb, const EP& e2) { if (equal(va, vb)) k(e2); }); });
  }
  void unify_array(const Term& pat, const V& val, size_t i, const EP& env, const Rule& rule, KE k) {
    if (i == pat.args.size()) { k(env); return; }
    unify_value(*pat.args[i], val.a[i], env, rule, [&](const EP& e) { unify_array(pat, val, i + 1, e, rule, k); });
  }
  void unify_object(const Term& pat, const V& val, size_t i, const EP& env, const Rule& rule, KE k) {
    if (2 * i >= pat.args.size()) { k(env); return; }
    eval_term(*pat.args[2 * i], env, rule, [&](const VP& kv, const EP& e1) {
      const VP* f = obj_get(val, kv);
      if (f) unify_value(*pat.args[2 * i + 1], *f, e1, rule, [&](const EP& e2) { unify_object(pat, val, i + 1, e2, rule, k); });
    });
  }
  void unify_value(const Term& pat, const VP& val, const EP& env, const Rule& rule, KE k) {
    if (pat.k == Term::Var && is_unbound_var(pat, env, rule)) { k(bind(env, pat.name, val)); return; }
    if (pat.k == Term::Array && has_unbound(pat, env, rule)) { if (val->k == V::Arr && val->a.size() == pat.args.size()) unify_array(pat, *val, 0, env, rule, k); return; }
    if (pat.k == Term::Object && has_unbound(pat, env, rule)) { if (val->k == V::Obj && val->o.size() == pat.args.size() / 2) unify_object(pat, *val, 0, env, rule, k); return; }
    eval_term(pat, env, rule, [&](const VP& v, const EP& e) { if (equal(v, val)) k(e); });
  }

  // ---- terms
  void eval_seq(const std::vector<TP>& ts, size_t i, std::vector<VP>& acc, const EP& env, const Rule& rule, fref<void(const std::vector<VP>&, const EP&)> k) {
    if (i == ts.size()) { k(acc, env); return; }
    eval_term(*ts[i], env, rule, [&](const VP& v, const EP& e) { acc.push_back(v); eval_seq(ts, i + 1, acc, e, rule, k); acc.pop_back(); });
  }
  static void iter_kv(const VP& c, fref<void(const VP&, const VP&)> fn) {
    if (c->k == V::Arr) { for (size_t i = 0; i < c->a.size(); i++) fn(mk_int((i128)i), c->a[i]); }
    else if (c->k == V::Obj) { for (auto& p : c->o) fn(p.first, p.second); }
    else if (c->k == V::Set) { for (auto& x : c->a) fn(x, x); }
  }
  static VP index(const VP& cur, const VP& key) {   // nullptr: nothing there (_index)
    if (cur->k == V::Obj) { const VP* v = obj_get(*cur, key); return v ? *v : nullptr; }
    if (cur->k == V::Arr) {
      if (key->k != V::Num) return nullptr;
      i128 i;
      if (key->is_int) i = key->i; else { if (std::floor(key->d) != key->d) return nullptr; i = (i128)key->d; }
      return i >= 0 && (size_t)i < cur->a.size() ? cur->a[(size_t)i] : nullptr;
    }
    if (cur->k == V::Set) return set_has(*cur, key) ? key : nullptr;
    return nullptr;
  }
  void walk(const VP& cur, const std::vector<TP>& ops, size_t i, const EP& env, const Rule& rule, KV k) {
    if (i == ops.size()) { k(cur, env); return; }
    const Term& op = *ops[i];
    if (op.k == Term::Var && is_unbound_var(op, env, rule)) { iter_kv(cur, [&](const VP& key, const VP& val) { walk(val, ops, i + 1, bind(env, op.name, key), rule, k); }); return; }
    if (op.k == Term::Scalar) { VP nxt = index(cur, op.val); if (nxt) walk(nxt, ops, i + 1, env, rule, k); return; }
    if ((op.k == Term::Array || op.k == Term::Object) && has_unbound(op, env, rule)) { iter_kv(cur, [&](const VP& key, const VP& val) { unify_value(op, key, env, rule, [&](const EP& e) { walk(val, ops, i + 1, e, rule, k); }); }); return; }
    eval_term(op, env, rule, [&](const VP& kv, const EP& e) { VP nxt = index(cur, kv); if (nxt) walk(nxt, ops, i + 1, e, rule, k); });
  }
  void eval_data_ref(const std::vector<TP>& ops, const EP& env, const Rule& rule, KV k) {
    std::vector<std::string> consts;
    for (auto& o : ops) { if (o->k == Term::Scalar && o->val->k == V::Str) consts.push_back(o->val->s); else break; }
    for (size_t n = consts.size(); n-- > 0;) {
      const std::vector<std::string> pkg(consts.begin(), consts.begin() + n);
      if (P.find(pkg, consts[n])) { VP v = rule_value(pkg, consts[n]); if (v) walk(v, ops, n + 1, env, rule, k); return; }
    }
    walk(P.data, ops, 0, env, rule, k);
  }
  void eval_call(const Term& t, const EP& env, const Rule& rule, KV k) {
    std::string name;
    for (size_t i = 0; i < t.path.size(); i++) { if (i) name += "."; name += t.path[i]; }
    const std::vector<std::string>* tpkg = nullptr;
    std::vector<std::string> pkgbuf;
    std::string tname;
    if (t.path.size() == 1 && P.find(rule.pkg, t.path[0])) { tpkg = &rule.pkg; tname = t.path[0]; }
    else if (t.path[0] == "data" && t.path.size() >= 2) { pkgbuf.assign(t.path.begin() + 1, t.path.end() - 1); if (P.find(pkgbuf, t.path.back())) { tpkg = &pkgbuf; tname = t.path.back(); } }
    else if (rule.imports.count(t.path[0])) {
      std::vector<std::string> full = rule.imports.at(t.path[0]);
      full.insert(full.end(), t.path.begin() + 1, t.path.end());
      if (full[0] == "data" && full.size() >= 2) { pkgbuf.assign(full.begin() + 1, full.end() - 1); if (P.find(pkgbuf, full.back())) { tpkg = &pkgbuf; tname = full.back(); } }
    }
    const std::pair<int, BuiltinFn>* bf = nullptr;
    if (!tpkg) { auto it = BUILTINS().find(name); if (it == BUILTINS().end()) throw EvalErr("undefined function " + name + " (outside this checker's builtins)"); bf = &it->second; }
    std::vector<VP> acc;
    eval_seq(t.args, 0, acc, env, rule, [&](const std::vector<VP>& args, const EP& e) {
      if (tpkg) { const std::vector<VP> copy = args; VP v = call_function(*tpkg, tname, copy); if (v) k(v, e); return; }
      if ((int)args.size() != bf->first) throw EvalErr(name + ": wrong number of arguments");
      VP v;
      try { v = bf->second(args); } catch (const BuiltinErr&) { return; }
      k(v, e);
    });
  }
  void eval_term(const Term& t, const EP& env, const Rule& rule, KV k) {
    switch (t.k) {
      case Term::Scalar: k(t.val, env); break;
      case Term::Var: {
        if (const VP* b = lookup(env, t.name)) { k(*b, env); return; }
        if (t.name == "input") { k(input_, env); return; }
        if (t.name == "data") { eval_data_ref({}, env, rule, k); return; }
        if (P.find(rule.pkg, t.name)) { VP v = rule_value(rule.pkg, t.name); if (v) k(v, env); return; }
        throw Unbound();
      }
      case Term::Ref: {
        const Term& head = *t.head;
        if (head.k == Term::Var && head.name == "data" && !lookup(env, "data")) { eval_data_ref(t.args, env, rule, k); return; }
        if (head.k == Term::Var && rule.imports.count(head.name) && !lookup(env, head.name)) {
          const std::vector<std::string>& path = rule.imports.at(head.name);
          std::vector<TP> full;
          for (size_t i = 1; i < path.size(); i++) { Term s; s.k = Term::Scalar; s.val = mk_str(path[i]); full.push_back(std::make_shared<const Term>(std::move(s))); }
          full.insert(full.end(), t.args.begin(), t.args.end());
          if (path[0] == "data") eval_data_ref(full, env, rule, k); else walk(input_, full, 0, env, rule, k);
          return;
        }
        eval_term(head, env, rule, [&](const VP& hv, const EP& e) { walk(hv, t.args, 0, e, rule, k); });
        break;
      }
      case Term::Call: eval_call(t, env, rule, k); break;
      case Term::BinOp: {
        const std::string& op = t.name;
        eval_term(*t.args[0], env, rule, [&](const VP& a, const EP& e) {
          eval_term(*t.args[1], e, rule, [&](const VP& b, const EP& e2) {
            if (op == "==" || op == "!=" || op == "<" || op == "<=" || op == ">" || op == ">=") {
              const int c = compare(a, b);
              const bool r = op == "==" ? c == 0 : op == "!=" ? c != 0 : op == "<" ? c < 0 : op == "<=" ? c <= 0 : op == ">" ? c > 0 : c >= 0;
              k(mk_bool(r), e2);
            } else if (op == "in") {
              bool any = false;
              iter_kv(b, [&](const VP&, const VP& v) { if (equal(a, v)) any = true; });
              k(mk_bool(any), e2);
            } else {
              VP v;
              try { v = arith(op, a, b); } catch (const BuiltinErr&) { return; }
              k(v, e2);
            }
          });
        });
        break;
      }
      case Term::Array: case Term::SetT: {
        std::vector<VP> acc;
        eval_seq(t.args, 0, acc, env, rule, [&](const std::vector<VP>& vals, const EP& e) { k(t.k == Term::Array ? mk_arr(vals) : mk_set(vals), e); });
        break;
      }
      case Term::Object: {
        std::vector<VP> acc;
        eval_seq(t.args, 0, acc, env, rule, [&](const std::vector<VP>& vals, const EP& e) {
          std::vector<std::pair<VP, VP>> p;
          for (size_t i = 0; i + 1 < vals.size(); i += 2) p.emplace_back(vals[i], vals[i + 1]);
          k(mk_obj(std::move(p)), e);
        });
        break;
      }
      case Term::ArrComp: case Term::SetComp: {
        std::vector<VP> out;
        eval_body(*t.body, env, rule, [&](const EP& e) { eval_term(*t.head, e, rule, [&](const VP& v, const EP&) { out.push_back(v); }); });
        k(t.k == Term::ArrComp ? mk_arr(std::move(out)) : mk_set(std::move(out)), env);
        break;
      }
      case Term::ObjComp: {
        std::vector<std::pair<VP, VP>> out;
        eval_body(*t.body, env, rule, [&](const EP& e) { eval_term(*t.head, e, rule, [&](const VP& kv, const EP& e2) { eval_term(*t.head2, e2, rule, [&](const VP& vv, const EP&) { out.emplace_back(kv, vv); }); }); });
        k(mk_obj(std::move(out)), env);
        break;
      }
    }
  }
};

// ================================================================================================ Match layer (oracle/match.py, oracle/target.py Matcher)
struct MatchErr : std::runtime_error { using std::runtime_error::runtime_error; };
static std::string s_of(const VP* v) { return v && (*v)->k == V::Str ? (*v)->s : std::string(); }
static const V* meta_of(const V& obj) { const VP* m = obj_get(obj, "metadata"); return m && (*m)->k == V::Obj ? m->get() : nullptr; }
static std::string obj_name(const V& o) { const V* m = meta_of(o); return m ? s_of(obj_get(*m, "name")) : ""; }
static std::string obj_generate_name(const V& o) { const V* m = meta_of(o); return m ? s_of(obj_get(*m, "generateName")) : ""; }
static std::string obj_namespace(const V& o) { const V* m = meta_of(o); return m ? s_of(obj_get(*m, "namespace")) : ""; }
static std::map<std::string, std::string> obj_labels(const V& o) {   // GetLabels -> NestedStringMap: one non-string value empties the map
  std::map<std::string, std::string> out;
  const V* m = meta_of(o);
  const VP* l = m ? obj_get(*m, "labels") : nullptr;
  if (!l || (*l)->k != V::Obj) return out;
  for (auto& p : (*l)->o) if (p.second->k != V::Str) return {};
  for (auto& p : (*l)->o) if (p.first->k == V::Str) out[p.first->s] = p.second->s;
  return out;
}
static void parse_gv(const std::string& av, std::string* g, std::string* v) {
  g->clear(); v->clear();
  if (av.empty() || av == "/") return;
  const size_t n = (size_t)std::count(av.begin(), av.end(), '/');
  if (n == 0) *v = av;
  else if (n == 1) { const size_t i = av.find('/'); *g = av.substr(0, i); *v = av.substr(i + 1); }
}
static void obj_gvk(const V& o, std::string* g, std::string* v, std::string* k) { parse_gv(s_of(obj_get(o, "apiVersion")), g, v); *k = s_of(obj_get(o, "kind")); }
static bool is_namespace(const V& o) { std::string g, v, k; obj_gvk(o, &g, &v, &k); return k == "Namespace" && g.empty(); }
static bool starts(const std::string& s, const std::string& p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
static bool ends(const std::string& s, const std::string& p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
static bool wildcard_matches(const std::string& w, const std::string& c) {   // wildcard.go:17-30
  const bool pre = !w.empty() && w[0] == '*', suf = !w.empty() && w.back() == '*';
  if (pre && suf) { std::string inner = w.substr(1); if (!inner.empty() && inner.back() == '*') inner.pop_back(); return c.find(inner) != std::string::npos; }
  if (pre) return ends(c, w.substr(1));
  if (suf) return starts(c, w.substr(0, w.size() - 1));
  return w == c;
}
static bool wildcard_matches_generate_name(const std::string& w, const std::string& c) {   // wildcard.go:32-41
  const bool pre = !w.empty() && w[0] == '*', suf = !w.empty() && w.back() == '*';
  if (pre && suf) { std::string inner = w.substr(1); if (!inner.empty() && inner.back() == '*') inner.pop_back(); return c.find(inner) != std::string::npos; }
  if (suf) return starts(c, w.substr(0, w.size() - 1));
  return false;
}
static bool alnum(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); }
static bool qname(const std::string& s) {   // ^([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]$
  if (s.empty() || !alnum(s.back()) || !alnum(s[0])) return false;
  for (char c : s) if (!(alnum(c) || c == '-' || c == '_' || c == '.')) return false;
  return true;
}
static bool dns1123_sub(const std::string& s) {
  size_t pos = 0;
  while (pos <= s.size()) {
    size_t q = s.find('.', pos);
    if (q == std::string::npos) q = s.size();
    const std::string lab = s.substr(pos, q - pos);
    auto lc = [](char c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); };
    if (lab.empty() || !lc(lab[0]) || !lc(lab.back())) return false;
    for (char c : lab) if (!(lc(c) || c == '-')) return false;
    pos = q + 1;
  }
  return true;
}
static bool valid_label_key(const std::string& k) {
  const size_t n = (size_t)std::count(k.begin(), k.end(), '/');
  std::string name = k;
  if (n == 1) { const size_t i = k.find('/'); const std::string prefix = k.substr(0, i); name = k.substr(i + 1); if (prefix.empty() || prefix.size() > 253 || !dns1123_sub(prefix)) return false; }
  else if (n > 1) return false;
  return !name.empty() && name.size() <= 63 && qname(name);
}
static bool valid_label_value(const std::string& v) { return v.size() <= 63 && (v.empty() || qname(v)); }
struct Req { std::string key, op; std::vector<std::string> vals; };
static Req new_requirement(const VP& key, const std::string& op, const std::vector<VP>& vals) {
  if (key->k != V::Str || !valid_label_key(key->s)) throw MatchErr("key: Invalid value");
  if (op == "In" || op == "NotIn") { if (vals.empty()) throw MatchErr("values: Invalid value: []"); }
  else if (op == "Equals") { if (vals.size() != 1) throw MatchErr("values: Invalid value: exact-match"); }
  else if (!vals.empty()) throw MatchErr("values: Invalid value: values set must be empty");
  Req r; r.key = key->s; r.op = op;
  for (auto& v : vals) { if (v->k != V::Str || !valid_label_value(v->s)) throw MatchErr("values: Invalid value"); r.vals.push_back(v->s); }
  return r;
}
static bool truthy_py(const VP* v) {   // Python `x or default`
  if (!v) return false;
  const V& x = **v;
  switch (x.k) { case V::Null: return false; case V::Bool: return x.b; case V::Num: return x.is_int ? x.i != 0 : x.d != 0; case V::Str: return !x.s.empty(); case V::Arr: case V::Set: return !x.a.empty(); case V::Obj: return !x.o.empty(); }
  return false;
}
// selector_requirements: false = the empty selector (everything)
static bool selector_requirements(const V& sel, std::vector<Req>* reqs) {
  const VP* ml = obj_get(sel, "matchLabels");
  const VP* me = obj_get(sel, "matchExpressions");
  const size_t nml = truthy_py(ml) && (*ml)->k == V::Obj ? (*ml)->o.size() : 0, nme = truthy_py(me) && (*me)->k == V::Arr ? (*me)->a.size() : 0;
  if (nml + nme == 0) return false;
  if (nml) for (auto& p : (*ml)->o) reqs->push_back(new_requirement(p.first, "Equals", {p.second}));   // (sorted keys: the object's own order)
  if (nme) for (auto& e : (*me)->a) {
    if (e->k != V::Obj) throw MatchErr("malformed requirement");
    const std::string op = s_of(obj_get(*e, "operator"));
    if (op != "In" && op != "NotIn" && op != "Exists" && op != "DoesNotExist") throw MatchErr("\"" + op + "\" is not a valid label selector operator");
    const VP* key = obj_get(*e, "key");
    const VP* vals = obj_get(*e, "values");
    reqs->push_back(new_requirement(key ? *key : mk_str(""), op, truthy_py(vals) && (*vals)->k == V::Arr ? (*vals)->a : std::vector<VP>()));
  }
  return true;
}
static bool selector_matches(const std::vector<Req>& reqs, const std::map<std::string, std::string>& labels) {
  for (const Req& r : reqs) {
    auto it = labels.find(r.key);
    const bool has = it != labels.end();
    const bool in = has && std::find(r.vals.begin(), r.vals.end(), it->second) != r.vals.end();
    bool ok;
    if (r.op == "In" || r.op == "Equals") ok = in;
    else if (r.op == "NotIn") ok = !has || !in;
    else if (r.op == "Exists") ok = has;
    else ok = !has;
    if (!ok) return false;
  }
  return true;
}
static std::vector<std::string> str_list(const VP* v) { std::vector<std::string> o; if (truthy_py(v) && (*v)->k == V::Arr) for (auto& x : (*v)->a) o.push_back(x->k == V::Str ? x->s : std::string("\x01not-a-string")); return o; }
static bool contains(const std::vector<std::string>& v, const std::string& s) { return std::find(v.begin(), v.end(), s) != v.end(); }
// match.Matches (match.go:32-65) for one object; throws MatchErr
static bool matches(const V& match, const V& obj, const V* ns, const std::string& source) {
  {   // kinds_match
    const VP* kinds = obj_get(match, "kinds");
    if (truthy_py(kinds) && (*kinds)->k == V::Arr && !(*kinds)->a.empty()) {
      std::string g, v, k; obj_gvk(obj, &g, &v, &k);
      bool any = false;
      for (auto& kk : (*kinds)->a) {
        if (kk->k != V::Obj) continue;
        const std::vector<std::string> ks = str_list(obj_get(*kk, "kinds")), gs = str_list(obj_get(*kk, "apiGroups"));
        if (!(ks.empty() || contains(ks, "*") || contains(ks, k))) continue;
        if (gs.empty() || contains(gs, "*") || contains(gs, g)) { any = true; break; }
      }
      if (!any) return false;
    }
  }
  const bool is_ns = is_namespace(obj);
  {   // scope_match
    const bool has_ns = !obj_namespace(obj).empty() || ns != nullptr;
    const std::string scope = s_of(obj_get(match, "scope"));
    if (scope == "Cluster" && !(is_ns || !has_ns)) return false;
    if (scope == "Namespaced" && !(!is_ns && has_ns)) return false;
  }
  bool have_name = true;
  std::string eff;
  if (is_ns) eff = obj_name(obj);
  else if (ns) { const V* m = meta_of(*ns); eff = m ? s_of(obj_get(*m, "name")) : ""; }
  else if (!obj_namespace(obj).empty()) eff = obj_namespace(obj);
  else have_name = false;
  {   // namespaces_match / excluded_namespaces_match
    const std::vector<std::string> nss = str_list(obj_get(match, "namespaces"));
    if (!nss.empty() && have_name) { bool any = false; for (auto& n : nss) if (wildcard_matches(n, eff)) any = true; if (!any) return false; }
    const std::vector<std::string> ex = str_list(obj_get(match, "excludedNamespaces"));
    if (!ex.empty() && have_name) { for (auto& n : ex) if (wildcard_matches(n, eff)) return false; }
  }
  {   // label_selector_match
    const VP* sel = obj_get(match, "labelSelector");
    if (sel && (*sel)->k != V::Null) {
      if ((*sel)->k != V::Obj) throw MatchErr("labelSelector is not a map");
      std::vector<Req> reqs;
      if (selector_requirements(**sel, &reqs) && !selector_matches(reqs, obj_labels(obj))) return false;
    }
  }
  {   // namespace_selector_match
    const VP* sel = obj_get(match, "namespaceSelector");
    if (sel && (*sel)->k != V::Null && !(!is_ns && !ns && obj_namespace(obj).empty())) {
      if ((*sel)->k != V::Obj) throw MatchErr("namespaceSelector is not a map");
      std::vector<Req> reqs;
      const bool some = selector_requirements(**sel, &reqs);
      if (is_ns) { if (some && !selector_matches(reqs, obj_labels(obj))) return false; }
      else {
        if (!ns) throw MatchErr("namespace selector for namespace-scoped object but missing Namespace");
        if (some && !selector_matches(reqs, obj_labels(*ns))) return false;
      }
    }
  }
  {   // names_match
    const std::string name = s_of(obj_get(match, "name"));
    if (!name.empty() && !(wildcard_matches(name, obj_name(obj)) || wildcard_matches_generate_name(name, obj_generate_name(obj)))) return false;
  }
  {   // source_match
    std::string m_src = s_of(obj_get(match, "source"));
    if (m_src.empty()) m_src = "All";
    else if (m_src != "All" && m_src != "Generated" && m_src != "Original") throw MatchErr("invalid source field");
    if (source.empty() && m_src != "All") throw MatchErr("source field not specified");
    if (m_src != "All") {
      if (source != "All" && source != "Generated" && source != "Original") throw MatchErr("invalid source field");
      if (m_src != source) return false;
    }
  }
  return true;
}

// ================================================================================================ the client (oracle/client.py Client.review for object reviews at the audit enforcement point)
struct Constraint { VP doc, match, params; std::string kind; const Program* prog = nullptr; };
struct Checker {
  std::map<std::string, std::unique_ptr<Program>> templates;   // lower(kind)
  std::vector<Constraint> constraints;                        // rows
};
static std::string lower_ascii(std::string s) { for (char& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c + 32); return s; }
static const VP* path_get(const VP& v, std::initializer_list<const char*> keys) {
  const VP* cur = &v;
  for (const char* k : keys) { if ((*cur)->k != V::Obj) return nullptr; cur = obj_get(**cur, k); if (!cur) return nullptr; }
  return cur;
}
static Checker* build(const char* templates_json, const char* constraints_json) {
  std::unique_ptr<Checker> c(new Checker());
  VP ts = parse_json(templates_json, strlen(templates_json));
  if (ts->k != V::Arr) throw std::runtime_error("templates: expected a JSON array");
  for (auto& t : ts->a) {
    const VP* kind = path_get(t, {"spec", "crd", "spec", "names", "kind"});
    const VP* targets = path_get(t, {"spec", "targets"});
    if (!kind || (*kind)->k != V::Str || !targets || (*targets)->k != V::Arr || (*targets)->a.size() != 1) throw std::runtime_error("invalid ConstraintTemplate");
    const VP& tg = (*targets)->a[0];
    const VP* rego = path_get(tg, {"rego"});
    if (!rego || (*rego)->k != V::Str) throw std::runtime_error("template " + (*kind)->s + " has no Rego source");
    std::vector<std::string> sources{(*rego)->s};
    if (const VP* libs = path_get(tg, {"libs"})) if ((*libs)->k == V::Arr) for (auto& l : (*libs)->a) if (l->k == V::Str) sources.push_back(l->s);
    std::unique_ptr<Program> p(new Program());
    p->load(sources);
    c->templates[lower_ascii((*kind)->s)] = std::move(p);
  }
  VP cs = parse_json(constraints_json, strlen(constraints_json));
  if (cs->k != V::Arr) throw std::runtime_error("constraints: expected a JSON array");
  for (auto& k : cs->a) {
    Constraint x;
    x.doc = k;
    x.kind = s_of(path_get(k, {"kind"}));
    auto it = c->templates.find(lower_ascii(x.kind));
    if (it == c->templates.end()) throw std::runtime_error("missing ConstraintTemplate: " + x.kind);
    x.prog = it->second.get();
    const VP* m = path_get(k, {"spec", "match"});
    if (m && (*m)->k != V::Null) { if ((*m)->k != V::Obj) throw std::runtime_error("unable to create matcher: spec.match is not a map"); x.match = *m; }
    const VP* p = path_get(k, {"spec", "parameters"});
    x.params = p && (*p)->k != V::Null ? *p : mk_obj({});
    c->constraints.push_back(std::move(x));
  }
  return c.release();
}
// ---- K8sValidationTarget.HandleReview (oracle/target.py handle_review; target.go:81-179, 269-287) for both shapes of gk_review_in
struct ReviewErr : std::runtime_error { using std::runtime_error::runtime_error; };   // HandleReview refuses the input (no results at all)
struct Rv {
  VP input;         // input.review: the AdmissionRequest as the Rego driver sees it (review_input_json)
  VP obj, old;      // request.object / request.oldObject when they hold a JSON object (RawExtension with bytes), else null pointers
  VP ns;            // gkReview.namespace
  std::string source;
};
static VP raw_of(const std::map<std::string, VP>& req, const char* key) {
  auto it = req.find(key);
  return it != req.end() && it->second->k == V::Obj ? it->second : nullptr;
}
static VP member(const std::map<std::string, VP>& req, const char* key) { auto it = req.find(key); return it == req.end() ? nullptr : it->second; }
static Rv handle_review(const gk_review_in& r) {
  static const char* const kSources[] = {"", "Original", "Generated", "All"};
  Rv rv;
  rv.source = r.source >= 0 && r.source <= 3 ? kSources[r.source] : "invalid";
  if (r.namespace_json && r.namespace_len) { rv.ns = parse_json(r.namespace_json, r.namespace_len); if (rv.ns->k == V::Null) rv.ns = nullptr; }
  const VP doc = parse_json(r.json, r.json_len);
  if (doc->k != V::Obj) throw ReviewErr("the review is not a JSON object");
  std::map<std::string, VP> req;   // the AdmissionRequest's members by name
  if (r.kind == 1) {   // unstructuredToAdmissionRequest / augmentedUnstructuredToAdmissionRequest (target.go:140-179)
    std::string g, v, k;
    obj_gvk(*doc, &g, &v, &k);
    req["kind"] = mk_obj({{mk_str("group"), mk_str(g)}, {mk_str("version"), mk_str(v)}, {mk_str("kind"), mk_str(k)}});
    req["object"] = doc;
    req["name"] = mk_str(obj_name(*doc));
    req["namespace"] = mk_str(obj_namespace(*doc));
    const std::string op = r.operation ? r.operation : "";
    if (!op.empty()) req["operation"] = mk_str(op);
    if (op == "DELETE") { req["oldObject"] = doc; req["object"] = mk_null(); }
  } else if (r.kind == 0) {
    for (auto& kv : doc->o) if (kv.first->k == V::Str) req[kv.first->s] = kv.second;
  } else throw ReviewErr("unknown review kind");
  // setObjectOnDelete (target.go:269-287)
  if (s_of(req.count("operation") ? &req["operation"] : nullptr) == "DELETE") {
    if (!raw_of(req, "oldObject")) throw ReviewErr("oldObject cannot be nil for DELETE operations");
    req["object"] = req["oldObject"];
  }
  rv.obj = raw_of(req, "object");
  rv.old = raw_of(req, "oldObject");
  // review_input_json: admissionv1.AdmissionRequest's struct tags -- uid / kind / resource / operation / userInfo always there, the
  // RawExtensions null when empty, omitempty on subResource, requestSubResource, name, namespace, requestKind, requestResource, dryRun
  auto text_of = [](const VP& o, const char* key) { if (!o || o->k != V::Obj) return mk_str(""); const VP* p = obj_get(*o, key); return p ? *p : mk_str(""); };
  const VP kind = member(req, "kind"), res = member(req, "resource");
  const VP kind_or_empty = kind && truthy_py(&kind) ? kind : nullptr, res_or_empty = res && truthy_py(&res) ? res : nullptr;
  std::vector<std::pair<VP, VP>> in;
  const VP uid = member(req, "uid"), oper = member(req, "operation"), user = member(req, "userInfo"), options = member(req, "options");
  in.emplace_back(mk_str("uid"), uid ? uid : mk_str(""));
  in.emplace_back(mk_str("kind"), mk_obj({{mk_str("group"), text_of(kind_or_empty, "group")}, {mk_str("version"), text_of(kind_or_empty, "version")}, {mk_str("kind"), text_of(kind_or_empty, "kind")}}));
  in.emplace_back(mk_str("resource"), mk_obj({{mk_str("group"), text_of(res_or_empty, "group")}, {mk_str("version"), text_of(res_or_empty, "version")}, {mk_str("resource"), text_of(res_or_empty, "resource")}}));
  in.emplace_back(mk_str("operation"), oper ? oper : mk_str(""));
  in.emplace_back(mk_str("userInfo"), user && truthy_py(&user) ? user : mk_obj({}));
  in.emplace_back(mk_str("object"), rv.obj ? rv.obj : mk_null());
  in.emplace_back(mk_str("oldObject"), rv.old ? rv.old : mk_null());
  in.emplace_back(mk_str("options"), options ? options : mk_null());
  for (const char* k : {"subResource", "requestSubResource", "name", "namespace"}) { const VP v = member(req, k); if (v && truthy_py(&v)) in.emplace_back(mk_str(k), v); }
  for (const char* k : {"requestKind", "requestResource", "dryRun"}) { const VP v = member(req, k); if (v && v->k != V::Null) in.emplace_back(mk_str(k), v); }
  if (r.ns_object_json && r.ns_object_len) in.emplace_back(mk_str("namespaceObject"), parse_json(r.ns_object_json, r.ns_object_len));
  rv.input = mk_obj(std::move(in));
  return rv;
}
// Matcher.Match (matcher.go:21-71; target.py Matcher.match_review): gkReviewToObject refuses a document without a `kind`, then matchAny
// over object and oldObject.  Throws MatchErr (the constraint's autoreject).  The Namespace is the one handed in with the review (this
// checker keeps no cache: callers pass what the cache would answer).
static bool match_review(const V& match, const Rv& rv) {
  for (const VP* o : {&rv.obj, &rv.old}) {
    if (!*o) continue;
    const VP* kind = obj_get(**o, "kind");
    if (!kind || (*kind)->k != V::Str || (*kind)->s.empty()) throw MatchErr("invalid request object: failed to unmarshal gkReview object");
  }
  int nil = 0;
  for (const VP* o : {&rv.obj, &rv.old}) {
    if (!*o) { nil++; continue; }
    if (matches(match, **o, rv.ns.get(), rv.source)) return true;
  }
  if (nil == 2) throw MatchErr("invalid request object: neither object nor old object are defined");
  return false;
}

// one review: which rows violate, which are autorejected (matching failed)
// results (may be null): per row, += the number of types.Results of this review -- one per distinct (msg, details), as the Rego driver
// dedupes them (what pkg/audit/manager.go:893-904 totals per constraint)
static void review_one(const Checker& c, const gk_review_in& r, std::vector<uint32_t>* viol, std::vector<uint32_t>* err, uint64_t* results = nullptr) {
  const Rv rv = handle_review(r);
  std::map<std::pair<const Program*, const V*>, uint32_t> memo;   // (program, parameters) -> how many results
  for (size_t row = 0; row < c.constraints.size(); row++) {
    const Constraint& x = c.constraints[row];
    if (x.match) {
      try { if (!match_review(*x.match, rv)) continue; } catch (const MatchErr&) { err->push_back((uint32_t)row); continue; }
    }
    auto key = std::make_pair(x.prog, x.params.get());
    auto it = memo.find(key);
    uint32_t count;
    if (it != memo.end()) count = it->second;
    else {
      Query q(*x.prog, mk_obj({{mk_str("review"), rv.input}, {mk_str("parameters"), x.params}}));
      const VP set = q.violations();
      count = 0;
      std::set<std::pair<std::string, std::string>> seen;
      for (auto& res : set->a) {
        if (res->k != V::Obj) continue;
        const VP* m = obj_get(*res, "msg");
        if (!m || (*m)->k != V::Str) continue;
        if (!results) { count = 1; break; }
        const VP* d = obj_get(*res, "details");
        if (seen.insert({(*m)->s, d ? to_string(*d) : std::string("{}")}).second) count++;
      }
      memo[key] = count;
    }
    if (count) viol->push_back((uint32_t)row);
    if (results) results[row] += count;
  }
}

// the messages of one review: row -> the msg of every result (one per distinct (msg, details), as Client.review's driver dedupes)
static void review_messages(const Checker& c, const gk_review_in& r, std::map<uint32_t, std::vector<std::string>>* out) {
  const Rv rv = handle_review(r);
  for (size_t row = 0; row < c.constraints.size(); row++) {
    const Constraint& x = c.constraints[row];
    if (x.match) {
      try { if (!match_review(*x.match, rv)) continue; } catch (const MatchErr&) { continue; }
    }
    Query q(*x.prog, mk_obj({{mk_str("review"), rv.input}, {mk_str("parameters"), x.params}}));
    const VP set = q.violations();
    std::set<std::pair<std::string, std::string>> seen;
    for (auto& res : set->a) {
      if (res->k != V::Obj) continue;
      const VP* m = obj_get(*res, "msg");
      if (!m || (*m)->k != V::Str) continue;
      const VP* d = obj_get(*res, "details");
      if (seen.insert({(*m)->s, d ? to_string(*d) : std::string("{}")}).second) (*out)[(uint32_t)row].push_back((*m)->s);
    }
  }
}
static std::string json_quote(const std::string& s) {
  std::string o = "\"";
  char buf[8];
  for (unsigned char ch : s) {
    if (ch == '"') o += "\\\""; else if (ch == '\\') o += "\\\\"; else if (ch < 0x20) { snprintf(buf, sizeof buf, "\\u%04x", ch); o += buf; } else o.push_back((char)ch);
  }
  return o + "\"";
}

}  // namespace ic

// ================================================================================================ C entry points (ctypes: oracle/indep_check.py)
static thread_local std::string g_ic_err;
extern "C" {
const char* ic_last_error() { return g_ic_err.c_str(); }
void* ic_create(const char* templates_json, const char* constraints_json) {
  try { return ic::build(templates_json, constraints_json); }
  catch (const std::exception& e) { g_ic_err = e.what(); return nullptr; }
}
void ic_destroy(void* h) { delete static_cast<ic::Checker*>(h); }
// the messages of ONE review as JSON {"<row>": ["msg", ..]} (rows with results only); free with ic_free; NULL with ic_last_error()
char* ic_messages(void* h, const gk_review_in* r) {
  try {
    std::map<uint32_t, std::vector<std::string>> out;
    ic::review_messages(*static_cast<ic::Checker*>(h), *r, &out);
    std::string js = "{";
    for (auto& kv : out) {
      if (js.size() > 1) js += ",";
      js += "\"" + std::to_string(kv.first) + "\":[";
      for (size_t i = 0; i < kv.second.size(); i++) { if (i) js += ","; js += ic::json_quote(kv.second[i]); }
      js += "]";
    }
    js += "}";
    char* buf = (char*)malloc(js.size() + 1);
    memcpy(buf, js.c_str(), js.size() + 1);
    return buf;
  } catch (const std::exception& e) { g_ic_err = e.what(); return nullptr; }
  catch (const ic::Unbound&) { g_ic_err = "unsafe variable"; return nullptr; }
}
void ic_free(void* p) { free(p); }
// bitmaps [n_constraints][words] (bit r of word r / 64 of row c: pair (c, review r)), zeroed by the caller; returns 0, or -1 with ic_last_error()
// `rejected` ([n] bytes, may be NULL): 1 where HandleReview refuses the review (no results, as the product's statuses[i]); with NULL
// such a review is an error of the call.
// results ([n_constraints], may be NULL, zeroed by the caller): RESULT totals per constraint over all n reviews
int ic_check_totals(void* h, const gk_review_in* reviews, size_t n, uint64_t* viol, uint64_t* err, uint8_t* rejected, uint64_t* results, size_t words, int threads);
int ic_check_reviews(void* h, const gk_review_in* reviews, size_t n, uint64_t* viol, uint64_t* err, uint8_t* rejected, size_t words, int threads) {
  return ic_check_totals(h, reviews, n, viol, err, rejected, nullptr, words, threads);
}
int ic_check_totals(void* h, const gk_review_in* reviews, size_t n, uint64_t* viol, uint64_t* err, uint8_t* rejected, uint64_t* results, size_t words, int threads) {
  const ic::Checker& c = *static_cast<ic::Checker*>(h);
  if (threads < 1) threads = 1;
  std::atomic<size_t> next{0};
  std::mutex mu;
  std::string first_err;
  auto fail = [&](size_t i, const std::string& what) { std::lock_guard<std::mutex> l(mu); if (first_err.empty()) first_err = "review " + std::to_string(i) + ": " + what; };
  auto work = [&]() {
    std::vector<uint32_t> v, e;
    std::vector<uint64_t> mine(results ? c.constraints.size() : 0, 0);
    struct Flush { std::vector<uint64_t>& m; uint64_t* out; std::mutex& mu; ~Flush() { if (!out) return; std::lock_guard<std::mutex> l(mu); for (size_t k = 0; k < m.size(); k++) out[k] += m[k]; } } flush{mine, results, mu};
    for (;;) {
      const size_t lo = next.fetch_add(64);   // (64 reviews = one word per row: no two threads share a word)
      if (lo >= n) return;
      for (size_t i = lo; i < std::min(n, lo + 64); i++) {
        v.clear(); e.clear();
        try { ic::review_one(c, reviews[i], &v, &e, results ? mine.data() : nullptr); }
        catch (const ic::ReviewErr& ex) { if (rejected) { rejected[i] = 1; continue; } fail(i, ex.what()); return; }
        catch (const std::exception& ex) { fail(i, ex.what()); return; }
        catch (const ic::Unbound&) { fail(i, "unsafe variable"); return; }
        for (uint32_t row : v) viol[(size_t)row * words + i / 64] |= 1ull << (i % 64);
        for (uint32_t row : e) err[(size_t)row * words + i / 64] |= 1ull << (i % 64);
      }
    }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < threads; t++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  if (!first_err.empty()) { g_ic_err = first_err; return -1; }
  return 0;
}
int ic_check(void* h, const gk_review_in* reviews, size_t n, uint64_t* viol, uint64_t* err, size_t words, int threads) {
  return ic_check_reviews(h, reviews, n, viol, err, nullptr, words, threads);
}
}
