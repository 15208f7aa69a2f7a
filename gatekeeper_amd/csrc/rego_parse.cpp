// Rego-subset lexer + parser. See rego_ast.hpp.
#include <cctype>
#include <set>

#include "rego_ast.hpp"

namespace gk {
namespace {

struct Tok {
  enum K { NL, Num, Ident, Kw, Str, Op, Eof } k;
  std::string text;
  Value num;
  int line;
};

const std::set<std::string> kKeywords = {"package", "import", "default", "not", "some", "every", "in", "if",
                                         "contains", "else", "with", "as", "true", "false", "null"};

[[noreturn]] void fail(int line, const std::string& msg) { throw RegoError("rego_parse_error: line " + std::to_string(line) + ": " + msg); }

void utf8_append(std::string& o, uint32_t cp) {
  if (cp < 0x80) o.push_back((char)cp);
  else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
  else { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
}

std::vector<Tok> tokenize(const std::string& s) {
  std::vector<Tok> out;
  size_t i = 0, n = s.size();
  int line = 1;
  while (i < n) {
    char c = s[i];
    if (c == ' ' || c == '\t' || c == '\r') { i++; continue; }
    if (c == '#') { while (i < n && s[i] != '\n') i++; continue; }
    if (c == '\n') { out.push_back({Tok::NL, "\n", Value(), line}); line++; i++; continue; }
    if (isdigit((unsigned char)c) || (c == '.' && i + 1 < n && isdigit((unsigned char)s[i + 1]))) {
      size_t j = i;
      bool is_int = true;
      while (j < n && isdigit((unsigned char)s[j])) j++;
      if (j < n && s[j] == '.' && j + 1 < n && isdigit((unsigned char)s[j + 1])) { is_int = false; j++; while (j < n && isdigit((unsigned char)s[j])) j++; }
      if (j < n && (s[j] == 'e' || s[j] == 'E')) {
        size_t k = j + 1;
        if (k < n && (s[k] == '+' || s[k] == '-')) k++;
        if (k < n && isdigit((unsigned char)s[k])) { is_int = false; j = k; while (j < n && isdigit((unsigned char)s[j])) j++; }
      }
      std::string t = s.substr(i, j - i);
      Value v;
      if (is_int && t.size() <= 37) { i128 x = 0; for (char d : t) x = x * 10 + (d - '0'); v = Value::integer(x); }
      else v = Value::real(strtod(t.c_str(), nullptr));
      out.push_back({Tok::Num, t, v, line});
      i = j;
      continue;
    }
    if (isalpha((unsigned char)c) || c == '_') {
      size_t j = i;
      while (j < n && (isalnum((unsigned char)s[j]) || s[j] == '_')) j++;
      std::string t = s.substr(i, j - i);
      out.push_back({kKeywords.count(t) ? Tok::Kw : Tok::Ident, t, Value(), line});
      i = j;
      continue;
    }
    if (c == '"') {
      std::string o;
      size_t j = i + 1;
      for (;;) {
        if (j >= n || s[j] == '\n') fail(line, "unterminated string");
        if (s[j] == '"') { j++; break; }
        if (s[j] == '\\') {
          j++;
          if (j >= n) fail(line, "bad escape");
          char e = s[j++];
          switch (e) {
            case '"': o.push_back('"'); break;
            case '\\': o.push_back('\\'); break;
            case '/': o.push_back('/'); break;
            case 'b': o.push_back('\b'); break;
            case 'f': o.push_back('\f'); break;
            case 'n': o.push_back('\n'); break;
            case 'r': o.push_back('\r'); break;
            case 't': o.push_back('\t'); break;
            case 'u': {
              if (j + 4 > n) fail(line, "bad \\u escape");
              utf8_append(o, (uint32_t)strtoul(s.substr(j, 4).c_str(), nullptr, 16));
              j += 4;
              break;
            }
            default: fail(line, std::string("bad escape \\") + e);
          }
        } else o.push_back(s[j++]);
      }
      out.push_back({Tok::Str, o, Value(), line});
      i = j;
      continue;
    }
    if (c == '`') {
      size_t j = s.find('`', i + 1);
      if (j == std::string::npos) fail(line, "unterminated raw string");
      std::string o = s.substr(i + 1, j - i - 1);
      out.push_back({Tok::Str, o, Value(), line});
      for (char ch : o) if (ch == '\n') line++;
      i = j + 1;
      continue;
    }
    static const char* two[] = {":=", "==", "!=", "<=", ">="};
    bool matched = false;
    for (const char* t : two)
      if (i + 1 < n && s[i] == t[0] && s[i + 1] == t[1]) { out.push_back({Tok::Op, t, Value(), line}); i += 2; matched = true; break; }
    if (matched) continue;
    if (strchr("{}[]().,;:|=<>+-*/%&", c)) { out.push_back({Tok::Op, std::string(1, c), Value(), line}); i++; continue; }
    fail(line, std::string("unexpected character '") + c + "'");
  }
  out.push_back({Tok::Eof, "", Value(), line});
  return out;
}

class Parser {
 public:
  explicit Parser(const std::string& src) : t_(tokenize(src)) {}

  Module module() {
    Module m;
    skip_nl();
    expect_kw("package");
    m.package = dotted();
    for (;;) {
      skip_nl();
      if (at(Tok::Eof)) break;
      if (accept_kw("import")) {
        auto path = dotted();
        std::string alias = path.back();
        if (accept_kw("as")) alias = expect(Tok::Ident).text;
        m.imports.emplace_back(path, alias);
        continue;
      }
      m.rules.push_back(rule());
    }
    return m;
  }

 private:
  std::vector<Tok> t_;
  size_t i_ = 0;
  int wild_ = 0;

  const Tok& peek(bool nl = false) {
    size_t j = i_;
    if (nl) while (t_[j].k == Tok::NL) j++;
    return t_[j];
  }
  void skip_nl() { while (t_[i_].k == Tok::NL) i_++; }
  const Tok& next(bool nl = false) { if (nl) skip_nl(); return t_[i_++]; }
  bool at(Tok::K k, bool nl = false) { return peek(nl).k == k; }
  bool at_op(const char* v, bool nl = false) { const Tok& t = peek(nl); return t.k == Tok::Op && t.text == v; }
  bool at_kw(const char* v, bool nl = false) { const Tok& t = peek(nl); return t.k == Tok::Kw && t.text == v; }
  bool accept_op(const char* v, bool nl = false) { if (at_op(v, nl)) { next(nl); return true; } return false; }
  bool accept_kw(const char* v, bool nl = false) { if (at_kw(v, nl)) { next(nl); return true; } return false; }
  const Tok& expect(Tok::K k, bool nl = false) {
    const Tok& t = next(nl);
    if (t.k != k) fail(t.line, "unexpected token '" + t.text + "'");
    return t;
  }
  void expect_op(const char* v, bool nl = false) {
    const Tok& t = next(nl);
    if (t.k != Tok::Op || t.text != v) fail(t.line, std::string("expected '") + v + "', got '" + t.text + "'");
  }
  void expect_kw(const char* v) {
    const Tok& t = next();
    if (t.k != Tok::Kw || t.text != v) fail(t.line, std::string("expected '") + v + "'");
  }

  std::vector<std::string> dotted() {
    const Tok& t = next();
    if (t.k != Tok::Ident && t.k != Tok::Kw) fail(t.line, "expected identifier");
    std::vector<std::string> parts{t.text};
    for (;;) {
      if (accept_op(".")) parts.push_back(next().text);
      else if (at_op("[")) { next(); parts.push_back(expect(Tok::Str).text); expect_op("]"); }
      else break;
    }
    return parts;
  }

  Rule rule() {
    Rule r;
    r.line = peek().line;
    r.is_default = accept_kw("default");
    r.name = expect(Tok::Ident).text;
    if (at_op(".")) throw Unsupported("unsupported on the device plan: a reference as rule head (line " + std::to_string(r.line) + ")");   // `a.b.c { .. }` (OPA >= 0.46)
    if (at_op("(")) {
      next();
      skip_nl();
      while (!at_op(")", true)) { r.args.push_back(term()); if (!accept_op(",", true)) break; }
      expect_op(")", true);
      r.kind = Rule::Function;
    } else if (at_op("[")) {
      next();
      r.key = term();
      expect_op("]", true);
      r.kind = Rule::PartialSet;
    } else if (accept_kw("contains")) {
      skip_nl();
      r.key = term();
      r.kind = Rule::PartialSet;
    }
    if (at_op("=") || at_op(":=")) {
      next();
      r.value = term();
      if (r.kind == Rule::PartialSet) r.kind = Rule::PartialObject;
    }
    bool has_if = false;
    if (at_kw("if", true)) { next(true); has_if = true; skip_nl(); }
    if (at_op("{")) r.body = braced_body();
    else if (has_if) r.body.push_back(literal());
    while (at_kw("else", true)) {
      next(true);
      TermP val;
      if (at_op("=") || at_op(":=")) { next(); val = term(); }
      accept_kw("if");
      Body b;
      if (at_op("{")) b = braced_body();
      else if (!(at(Tok::NL) || at(Tok::Eof))) b.push_back(literal());
      r.elses.emplace_back(val, b);
    }
    if (r.is_default && !r.value) fail(r.line, "default rule needs a value");
    return r;
  }

  Body braced_body() {
    expect_op("{");
    Body b = body_until("}");
    expect_op("}", true);
    return b;
  }

  Body body_until(const char* closer) {
    Body lits;
    for (;;) {
      skip_nl();
      while (accept_op(";")) skip_nl();
      if (at_op(closer)) break;
      lits.push_back(literal());
      if (!(at(Tok::NL) || at_op(";") || at_op(closer))) fail(peek().line, "expected end of literal, got '" + peek().text + "'");
    }
    return lits;
  }

  Literal literal() {
    Literal l;
    l.line = peek().line;
    if (accept_kw("not")) {
      l.kind = Literal::Not;
      l.inner = std::make_shared<const Literal>(literal());
      return l;
    }
    if (accept_kw("some")) {
      TermP first = term(false, true);
      if (at_op(",")) {
        next();
        TermP second = term(false, true);
        if (accept_kw("in")) { l.kind = Literal::SomeIn; l.a = first; l.b = second; l.c = term(); return l; }
        l.kind = Literal::Some;
        l.names = {first->name, second->name};
        while (accept_op(",")) l.names.push_back(term(false, true)->name);
        return l;
      }
      if (accept_kw("in")) { l.kind = Literal::SomeIn; l.b = first; l.c = term(); return l; }
      l.kind = Literal::Some;
      l.names = {first->name};
      return l;
    }
    if (accept_kw("every")) {
      TermP first = term(false, true);
      TermP key;
      if (accept_op(",")) { key = first; first = term(false, true); }
      expect_kw("in");
      l.kind = Literal::Every;
      l.a = key;
      l.b = first;
      l.c = term();
      l.body = std::make_shared<const Body>(braced_body());
      return l;
    }
    l.a = term();
    if (at_op(":=")) { next(); skip_nl(); l.kind = Literal::Assign; l.b = term(); }
    else if (at_op("=")) { next(); skip_nl(); l.kind = Literal::Unify; l.b = term(); }
    else l.kind = Literal::Expr;
    if (at_kw("with")) throw Unsupported("unsupported on the device plan: the `with` modifier (line " + std::to_string(l.line) + ")");
    return l;
  }

  static TermP mk(Term t) { return std::make_shared<const Term>(std::move(t)); }
  TermP binop(const std::string& op, TermP l, TermP r, int line) {
    Term t; t.kind = Term::BinOp; t.name = op; t.args = {l, r}; t.line = line;
    return mk(t);
  }

  TermP term(bool no_bitor = false, bool no_in = false) { return relation(no_bitor, no_in); }

  TermP relation(bool no_bitor, bool no_in) {
    TermP l = bitor_(no_bitor);
    for (;;) {
      const Tok& t = peek();
      if (t.k == Tok::Op && (t.text == "==" || t.text == "!=" || t.text == "<" || t.text == "<=" || t.text == ">" || t.text == ">=")) {
        std::string op = t.text; int line = t.line;
        next(); skip_nl();
        l = binop(op, l, bitor_(no_bitor), line);
      } else if (t.k == Tok::Kw && t.text == "in" && !no_in) {
        int line = t.line;
        next();
        l = binop("in", l, bitor_(no_bitor), line);
      } else return l;
    }
  }
  TermP bitor_(bool no_bitor) {
    TermP l = bitand_();
    while (!no_bitor && at_op("|")) { int line = next().line; skip_nl(); l = binop("|", l, bitand_(), line); }
    return l;
  }
  TermP bitand_() {
    TermP l = arith();
    while (at_op("&")) { int line = next().line; skip_nl(); l = binop("&", l, arith(), line); }
    return l;
  }
  TermP arith() {
    TermP l = factor();
    while (at_op("+") || at_op("-")) { const Tok& t = next(); std::string op = t.text; int line = t.line; skip_nl(); l = binop(op, l, factor(), line); }
    return l;
  }
  TermP factor() {
    TermP l = unary();
    while (at_op("*") || at_op("/") || at_op("%")) { const Tok& t = next(); std::string op = t.text; int line = t.line; skip_nl(); l = binop(op, l, unary(), line); }
    return l;
  }
  TermP unary() {
    if (at_op("-")) {
      int line = next().line;
      TermP t = unary();
      if (t->kind == Term::Scalar && t->value.is_number()) {
        Term n; n.kind = Term::Scalar; n.line = line;
        n.value = t->value.is_int ? Value::integer(-t->value.i) : Value::real(-t->value.d);
        return mk(n);
      }
      Term z; z.kind = Term::Scalar; z.value = Value::integer(0); z.line = line;
      return binop("-", mk(z), t, line);
    }
    return postfix(primary());
  }

  TermP postfix(TermP head) {
    std::vector<TermP> ops;
    for (;;) {
      if (at_op(".")) {
        next();
        const Tok& t = next();
        if (t.k != Tok::Ident && t.k != Tok::Kw) fail(t.line, "expected field name");
        Term s; s.kind = Term::Scalar; s.value = Value::string(t.text); s.line = t.line;
        ops.push_back(mk(s));
      } else if (at_op("[")) {
        next(); skip_nl();
        ops.push_back(term());
        expect_op("]", true);
      } else if (at_op("(")) {
        if (head->kind != Term::Var) fail(peek().line, "call on non-name");
        Term c; c.kind = Term::Call; c.line = peek().line;
        c.path.push_back(head->name);
        for (auto& o : ops) {
          if (o->kind != Term::Scalar || !o->value.is_string()) fail(peek().line, "call on non-name");
          c.path.push_back(o->value.str());
        }
        next(); skip_nl();
        while (!at_op(")", true)) { c.args.push_back(term()); if (!accept_op(",", true)) break; }
        expect_op(")", true);
        head = mk(c);
        ops.clear();
      } else break;
    }
    if (ops.empty()) return head;
    Term r; r.kind = Term::Ref; r.head = head; r.args = ops; r.line = head->line;
    return mk(r);
  }

  TermP primary() {
    const Tok& t = next();
    Term n; n.line = t.line;
    switch (t.k) {
      case Tok::Num: n.kind = Term::Scalar; n.value = t.num; return mk(n);
      case Tok::Str: n.kind = Term::Scalar; n.value = Value::string(t.text); return mk(n);
      case Tok::Kw:
        n.kind = Term::Scalar;
        if (t.text == "true") n.value = Value::boolean(true);
        else if (t.text == "false") n.value = Value::boolean(false);
        else if (t.text == "null") n.value = Value::null();
        else if (t.text == "contains") { n.kind = Term::Var; n.name = t.text; }   // OPA: `contains` anywhere BUT in rule heads gets no special treatment (the builtin contains(s, sub))
        else fail(t.line, "unexpected keyword '" + t.text + "'");
        return mk(n);
      case Tok::Ident:
        if (t.text == "_") { n.kind = Term::Var; n.name = "$w" + std::to_string(++wild_); return mk(n); }
        if (t.text == "set" && at_op("(")) {
          size_t save = i_;
          next();
          if (accept_op(")")) { n.kind = Term::SetLit; return mk(n); }
          i_ = save;
        }
        n.kind = Term::Var; n.name = t.text; return mk(n);
      case Tok::Op:
        if (t.text == "(") { skip_nl(); TermP e = term(); expect_op(")", true); return e; }
        if (t.text == "[") return array_or_comp(t.line);
        if (t.text == "{") return brace_term(t.line);
        // fallthrough
      default: fail(t.line, "unexpected token '" + t.text + "'");
    }
  }

  TermP array_or_comp(int line) {
    Term n; n.line = line;
    skip_nl();
    if (accept_op("]")) { n.kind = Term::Array; return mk(n); }
    TermP first = term(true);
    if (at_op("|", true)) {
      next(true);
      n.kind = Term::ArrComp; n.head = first;
      n.body = std::make_shared<const Body>(body_until("]"));
      expect_op("]", true);
      return mk(n);
    }
    n.kind = Term::Array;
    n.args.push_back(first);
    while (accept_op(",", true)) { skip_nl(); if (at_op("]")) break; n.args.push_back(term()); }
    expect_op("]", true);
    return mk(n);
  }

  TermP brace_term(int line) {
    Term n; n.line = line;
    skip_nl();
    if (accept_op("}")) { n.kind = Term::Object; return mk(n); }
    TermP first = term(true);
    if (at_op(":", true)) {
      next(true); skip_nl();
      TermP val = term(true);
      if (at_op("|", true)) {
        next(true);
        n.kind = Term::ObjComp; n.head = first; n.head2 = val;
        n.body = std::make_shared<const Body>(body_until("}"));
        expect_op("}", true);
        return mk(n);
      }
      n.kind = Term::Object;
      n.args = {first, val};
      while (accept_op(",", true)) {
        skip_nl();
        if (at_op("}")) break;
        TermP k = term();
        expect_op(":", true); skip_nl();
        n.args.push_back(k);
        n.args.push_back(term());
      }
      expect_op("}", true);
      return mk(n);
    }
    if (at_op("|", true)) {
      next(true);
      n.kind = Term::SetComp; n.head = first;
      n.body = std::make_shared<const Body>(body_until("}"));
      expect_op("}", true);
      return mk(n);
    }
    n.kind = Term::SetLit;
    n.args.push_back(first);
    while (accept_op(",", true)) { skip_nl(); if (at_op("}")) break; n.args.push_back(term()); }
    expect_op("}", true);
    return mk(n);
  }
};

}  // namespace

Module parse_rego(const std::string& src) { return Parser(src).module(); }

}  // namespace gk
