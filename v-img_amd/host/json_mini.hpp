// Minimal JSON reader for v-img scene files.  The reference parses with nlohmann::json 3.11.2
// (CMakeLists.txt, not vendored): numbers go through strtod (doubles) or integer parsing and are
// narrowed with static_cast on get<T>() — value() below does the same narrowing.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace jmini {

struct Value {
  enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
  bool b = false;
  int64_t i = 0;
  double d = 0.0;
  std::string s;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;  // file order kept

  bool is_array() const { return kind == Array; }
  bool is_object() const { return kind == Object; }
  bool is_number() const { return kind == Int || kind == Float; }
  bool is_string() const { return kind == String; }
  bool contains(const std::string& k) const {
    if (kind != Object) return false;
    for (auto& kv : obj)
      if (kv.first == k) return true;
    return false;
  }
  const Value& at(const std::string& k) const {
    for (auto& kv : obj)
      if (kv.first == k) return kv.second;
    throw std::runtime_error("json: missing key '" + k + "'");
  }
  const Value& at(size_t idx) const {
    if (kind != Array || idx >= arr.size()) throw std::runtime_error("json: bad array index");
    return arr[idx];
  }
  double as_double() const {
    if (kind == Int) return static_cast<double>(i);
    if (kind == Float) return d;
    throw std::runtime_error("json: not a number");
  }
  float as_float() const {
    if (kind == Int) return static_cast<float>(i);
    if (kind == Float) return static_cast<float>(d);
    throw std::runtime_error("json: not a number");
  }
  uint32_t as_u32() const {
    if (kind == Int) return static_cast<uint32_t>(i);
    if (kind == Float) return static_cast<uint32_t>(d);
    throw std::runtime_error("json: not a number");
  }
  const std::string& as_string() const {
    if (kind != String) throw std::runtime_error("json: not a string");
    return s;
  }
  float value_f(const std::string& k, float dflt) const {
    return contains(k) ? at(k).as_float() : dflt;
  }
  uint32_t value_u32(const std::string& k, uint32_t dflt) const {
    return contains(k) ? at(k).as_u32() : dflt;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : t_(text) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != t_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& t_;
  size_t p_ = 0;
  [[noreturn]] void fail(const char* what) {
    throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(p_));
  }
  void ws() {
    while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r'))
      ++p_;
  }
  Value value() {
    ws();
    if (p_ >= t_.size()) fail("unexpected end");
    char c = t_[p_];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') {
      Value v;
      v.kind = Value::String;
      v.s = string();
      return v;
    }
    if (t_.compare(p_, 4, "true") == 0) {
      p_ += 4;
      Value v;
      v.kind = Value::Bool;
      v.b = true;
      return v;
    }
    if (t_.compare(p_, 5, "false") == 0) {
      p_ += 5;
      Value v;
      v.kind = Value::Bool;
      return v;
    }
    if (t_.compare(p_, 4, "null") == 0) {
      p_ += 4;
      return Value{};
    }
    return number();
  }
  Value number() {
    size_t start = p_;
    bool is_float = false;
    if (p_ < t_.size() && (t_[p_] == '-' || t_[p_] == '+')) ++p_;
    while (p_ < t_.size()) {
      char c = t_[p_];
      if (c >= '0' && c <= '9') {
        ++p_;
      } else if (c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') {
        is_float = true;
        ++p_;
      } else {
        break;
      }
    }
    if (p_ == start) fail("bad value");
    std::string tok = t_.substr(start, p_ - start);
    Value v;
    if (is_float) {
      v.kind = Value::Float;
      v.d = std::strtod(tok.c_str(), nullptr);
    } else {
      v.kind = Value::Int;
      v.i = std::strtoll(tok.c_str(), nullptr, 10);
    }
    return v;
  }
  std::string string() {
    ++p_;  // opening quote
    std::string out;
    while (p_ < t_.size() && t_[p_] != '"') {
      char c = t_[p_++];
      if (c == '\\') {
        if (p_ >= t_.size()) fail("bad escape");
        char e = t_[p_++];
        switch (e) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': p_ += 4; out += '?'; break;
          default: out += e;
        }
      } else {
        out += c;
      }
    }
    if (p_ >= t_.size()) fail("unterminated string");
    ++p_;
    return out;
  }
  Value array() {
    Value v;
    v.kind = Value::Array;
    ++p_;
    ws();
    if (p_ < t_.size() && t_[p_] == ']') {
      ++p_;
      return v;
    }
    for (;;) {
      v.arr.push_back(value());
      ws();
      if (p_ >= t_.size()) fail("unterminated array");
      if (t_[p_] == ',') {
        ++p_;
        continue;
      }
      if (t_[p_] == ']') {
        ++p_;
        return v;
      }
      fail("expected , or ]");
    }
  }
  Value object() {
    Value v;
    v.kind = Value::Object;
    ++p_;
    ws();
    if (p_ < t_.size() && t_[p_] == '}') {
      ++p_;
      return v;
    }
    for (;;) {
      ws();
      if (p_ >= t_.size() || t_[p_] != '"') fail("expected key");
      std::string k = string();
      ws();
      if (p_ >= t_.size() || t_[p_] != ':') fail("expected :");
      ++p_;
      Value val = value();
      // nlohmann keeps the LAST duplicate key; emulate by overwriting
      bool replaced = false;
      for (auto& kv : v.obj)
        if (kv.first == k) {
          kv.second = val;
          replaced = true;
        }
      if (!replaced) v.obj.emplace_back(k, std::move(val));
      ws();
      if (p_ >= t_.size()) fail("unterminated object");
      if (t_[p_] == ',') {
        ++p_;
        continue;
      }
      if (t_[p_] == '}') {
        ++p_;
        return v;
      }
      fail("expected , or }");
    }
  }
};

}  // namespace jmini
