// Debug/test CLI for the partial evaluator: reads one JSON document
//   {"rego": "...", "libs": [...], "parameters": {...}, "review": {...}, "inventory": {...}, "mode": "render"|"compile"}
// from stdin; prints {"violations":[{"msg","details"}]} or {"formula": "..."} or {"error": "..."}.
#include <iostream>
#include <iterator>

#include "pe.hpp"

using namespace gk;

int main() {
  std::string in((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
  Value doc = parse_json(in);
  std::vector<std::string> libs;
  if (const Value* l = doc.get("libs")) for (auto& x : l->items()) libs.push_back(x.str());
  std::string out;
  try {
    Template t(doc.get("rego")->str(), libs);
    const Value* p = doc.get("parameters");
    Value params = p ? *p : Value::object({});
    const Value* m = doc.get("mode");
    if (m && m->str() == "compile") {
      int nq = 0;
      FP f = t.compile(params, &nq);
      ValuePairs o{{Value::string("formula"), Value::string(f_to_string(f))}};
      out = to_json(Value::object(o));
    } else {
      const Value* inv = doc.get("inventory");
      auto vs = t.render(*doc.get("review"), params, inv ? *inv : Value());
      ValueVec arr;
      for (auto& v : vs) {
        ValuePairs o{{Value::string("msg"), Value::string(v.msg)}};
        if (v.details.defined()) o.emplace_back(Value::string("details"), v.details);
        arr.push_back(Value::object(o));
      }
      ValuePairs o{{Value::string("violations"), Value::array(arr)}};
      out = to_json(Value::object(o));
    }
  } catch (const std::exception& e) {
    ValuePairs o{{Value::string("error"), Value::string(e.what())}};
    out = to_json(Value::object(o));
  }
  std::cout << out << std::endl;
  return 0;
}
