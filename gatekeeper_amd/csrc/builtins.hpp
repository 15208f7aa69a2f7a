// Concrete Rego builtins (see builtins.cpp).
#pragma once
#include <memory>
#include <string>

#include "value.hpp"

namespace gk {
bool has_builtin(const std::string& name);
bool is_opa_builtin(const std::string& name);   // a builtin of OPA (or gatekeeper) whether or not this engine implements it
Value call_builtin(const std::string& name, const ValueVec& args);        // Undefined on error / unknown
Value rego_arith(const std::string& op, const Value& a, const Value& b);  // + - * / % & | (numbers and sets)
std::string go_sprintf(const std::string& fmt, const ValueVec& args);
class Regex;
std::shared_ptr<Regex> get_regex(const std::string& pat);   // compiled-pattern cache; nullptr for an invalid pattern
// re_match(pat, <the n bytes at s>) without a Value: *valid = false for a pattern Go refuses (the builtin is then undefined)
bool builtin_regex_search(const std::string& pat, const char* s, size_t n, bool* valid);
}  // namespace gk
