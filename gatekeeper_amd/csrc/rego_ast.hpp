// Rego-subset AST + recursive-descent parser (host side of the AOT policy compiler).
//
// The reference hands ConstraintTemplate Rego to OPA's parser (github.com/open-policy-agent/opa v1.17.1, go.mod:19;
// reached through Driver.AddTemplate -- boundary exemplar pkg/drivers/k8scel/driver.go:74).  This front end covers
// the language surface the reference's in-tree templates use: v0 and v1 rule heads, partial
// sets/objects, complete rules, functions with several bodies, `not`, `some`/`some..in`/`every`, comprehensions,
// infix arithmetic/set operators and comparisons.  `with` is rejected.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "value.hpp"

namespace gk {

struct RegoError : std::runtime_error { using std::runtime_error::runtime_error; };
struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };   // valid Rego this engine does not compile: GK_ERR_UNSUPPORTED, the stock driver keeps the template

struct Term;
typedef std::shared_ptr<const Term> TermP;
struct Literal;
typedef std::vector<Literal> Body;

struct Term {
  enum Kind { Scalar, Var, Ref, Call, Array, Object, SetLit, ArrComp, SetComp, ObjComp, BinOp } kind = Scalar;
  Value value;                       // Scalar
  std::string name;                  // Var name; BinOp operator
  TermP head;                        // Ref head; comprehension head (key for ObjComp)
  TermP head2;                       // ObjComp value
  std::vector<TermP> args;           // Ref operands; Call args; Array/Set elems; Object k,v,k,v...; BinOp l,r
  std::vector<std::string> path;     // Call dotted name
  std::shared_ptr<const Body> body;  // comprehensions
  int line = 0;
};

struct Literal {
  enum Kind { Expr, Assign, Unify, Not, Some, SomeIn, Every } kind = Expr;
  TermP a, b, c;                            // Expr: a; Assign/Unify: a,b; SomeIn/Every: key a (may be null), val b, coll c
  std::vector<std::string> names;           // Some
  std::shared_ptr<const Literal> inner;     // Not
  std::shared_ptr<const Body> body;         // Every
  int line = 0;
};

struct Rule {
  enum Kind { Complete, PartialSet, PartialObject, Function } kind = Complete;
  std::string name;
  std::vector<TermP> args;   // Function
  TermP key;                 // PartialSet / PartialObject
  TermP value;               // may be null (=> true)
  Body body;
  bool is_default = false;
  std::vector<std::pair<TermP, Body>> elses;
  int line = 0;
};

struct Module {
  std::vector<std::string> package;
  std::vector<std::pair<std::vector<std::string>, std::string>> imports;   // path, alias
  std::vector<Rule> rules;
};

Module parse_rego(const std::string& src);

}  // namespace gk
