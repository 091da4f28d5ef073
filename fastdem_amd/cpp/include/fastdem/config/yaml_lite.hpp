// fastdem/config/yaml_lite.hpp — the YAML subset FastDEM configuration files use: nested block
// mappings of scalars, '#' comments, single/double quoted strings.  (The reference links yaml-cpp,
// fastdem/CMakeLists.txt:19; it is not available to this build, and the configuration schema of
// fastdem/src/config_fastdem.cpp:57-126 needs nothing beyond this subset.)  Anything else —
// sequences, flow collections, anchors, multi-line scalars — raises yaml::Error, never a silent
// misread.
#pragma once
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace fastdem {
namespace yaml {

class Error : public std::runtime_error {
 public:
  using std::runtime_error::runtime_error;
};

class Node {
 public:
  bool defined() const { return defined_; }
  explicit operator bool() const { return defined_; }
  bool isMap() const { return defined_ && !scalar_; }
  /// node["key"]: an undefined node when absent (like YAML::Node)
  const Node& operator[](const std::string& key) const {
    static const Node none;
    const auto it = children_.find(key);
    return it == children_.end() ? none : it->second;
  }
  template <typename T>
  T as() const;

 private:
  friend Node parse(const std::string&);
  bool defined_ = false, scalar_ = false;
  std::string text_;
  std::map<std::string, Node> children_;
};

namespace detail {
inline std::string trim(const std::string& s) {
  const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
// strip a trailing comment: '#' at line start or preceded by whitespace, outside quotes
inline std::string uncomment(const std::string& line) {
  char quote = 0;
  for (size_t i = 0; i < line.size(); ++i) {
    const char ch = line[i];
    if (quote) {
      if (ch == quote) quote = 0;
    } else if (ch == '"' || ch == '\'') {
      quote = ch;
    } else if (ch == '#' && (i == 0 || line[i - 1] == ' ' || line[i - 1] == '\t')) {
      return line.substr(0, i);
    }
  }
  return line;
}
inline std::string unquote(const std::string& v, int line_no) {
  if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'')) {
    if (v.back() != v.front()) throw Error("yaml: unterminated quoted string at line " + std::to_string(line_no));
    return v.substr(1, v.size() - 2);
  }
  return v;
}
}  // namespace detail

inline Node parse(const std::string& text) {
  Node root;
  root.defined_ = true;
  struct Frame { int indent; Node* node; };
  std::vector<Frame> stack{{-1, &root}};
  Node* pending = nullptr;  // "key:" with nothing after it: becomes a map if deeper lines follow
  int pending_indent = -1;
  std::istringstream in(text);
  std::string raw;
  int line_no = 0;
  while (std::getline(in, raw)) {
    ++line_no;
    const std::string body = detail::uncomment(raw);
    const std::string line = detail::trim(body);
    if (line.empty() || line == "---") continue;
    const size_t ind = body.find_first_not_of(' ');
    if (body[ind] == '\t') throw Error("yaml: tab indentation at line " + std::to_string(line_no));
    const int indent = int(ind);
    if (line[0] == '-' && (line.size() == 1 || line[1] == ' '))
      throw Error("yaml: sequences are not supported (line " + std::to_string(line_no) + ")");
    if (pending) {
      if (indent > pending_indent) stack.push_back({pending_indent, pending});  // it is a mapping
      pending = nullptr;
    }
    while (stack.size() > 1 && indent <= stack.back().indent) stack.pop_back();
    const size_t colon = line.find(':');
    if (colon == std::string::npos || (colon + 1 < line.size() && line[colon + 1] != ' '))
      throw Error("yaml: expected 'key: value' at line " + std::to_string(line_no));
    const std::string key = detail::unquote(detail::trim(line.substr(0, colon)), line_no);
    const std::string value = detail::trim(line.substr(colon + 1));
    if (key.empty()) throw Error("yaml: empty key at line " + std::to_string(line_no));
    Node& child = stack.back().node->children_[key];
    child = Node();
    child.defined_ = true;
    if (value.empty()) {
      pending = &child;
      pending_indent = indent;
      child.scalar_ = true;  // "key:" alone is a null scalar unless a deeper block follows
      child.text_.clear();
    } else {
      if (value[0] == '{' || value[0] == '[' || value[0] == '&' || value[0] == '*' || value[0] == '|' ||
          value[0] == '>')
        throw Error("yaml: unsupported construct at line " + std::to_string(line_no));
      child.scalar_ = true;
      child.text_ = detail::unquote(value, line_no);
    }
    stack.back().node->scalar_ = false;  // a node with children is a mapping
  }
  return root;
}

inline Node loadFile(const std::string& path) {
  std::ifstream fs(path);
  if (!fs.is_open()) throw Error("yaml: cannot open " + path);
  std::ostringstream ss;
  ss << fs.rdbuf();
  return parse(ss.str());
}

// yaml-cpp style conversions: the whole scalar must convert, otherwise Error ("bad conversion")
template <>
inline std::string Node::as<std::string>() const {
  if (!defined_ || !scalar_) throw Error("yaml: bad conversion (not a scalar)");
  return text_;
}
template <>
inline float Node::as<float>() const {
  const std::string s = as<std::string>();
  if (s == ".inf" || s == ".Inf" || s == ".INF" || s == "+.inf") return HUGE_VALF;
  if (s == "-.inf" || s == "-.Inf" || s == "-.INF") return -HUGE_VALF;
  char* end = nullptr;
  const float v = std::strtof(s.c_str(), &end);
  if (s.empty() || end != s.c_str() + s.size()) throw Error("yaml: bad conversion to float: '" + s + "'");
  return v;
}
template <>
inline int Node::as<int>() const {
  const std::string s = as<std::string>();
  char* end = nullptr;
  const long v = std::strtol(s.c_str(), &end, 0);
  if (s.empty() || end != s.c_str() + s.size()) throw Error("yaml: bad conversion to int: '" + s + "'");
  return int(v);
}
template <>
inline bool Node::as<bool>() const {
  std::string s = as<std::string>();
  for (auto& ch : s) ch = char(std::tolower(static_cast<unsigned char>(ch)));
  if (s == "true" || s == "yes" || s == "on" || s == "y") return true;
  if (s == "false" || s == "no" || s == "off" || s == "n") return false;
  throw Error("yaml: bad conversion to bool: '" + s + "'");
}

}  // namespace yaml
}  // namespace fastdem
