// oracle/shim — TEST INFRASTRUCTURE ONLY: a reader for the FLAT YAML the reference's config files use
// (legkilo/config/*.yaml: `key: scalar` or `key: [a, b, c]`, `#` comments) with the yaml-cpp calls
// common/yaml_helper.hpp makes: YAML::LoadFile, Node::operator[], conversion to bool, Node::as<T>().
#ifndef LK_SHIM_YAML
#define LK_SHIM_YAML
#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace YAML {
class Node {
    std::map<std::string, std::string> kv_;
    std::string scalar_;
    bool defined_ = false;

    static std::string trim(std::string s) {
        const char* ws = " \t\r\n";
        s.erase(0, s.find_first_not_of(ws));
        s.erase(s.find_last_not_of(ws) + 1);
        return s;
    }
    template <class T>
    static T conv(const std::string& s0) {
        std::string s = trim(s0);
        if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) s = s.substr(1, s.size() - 2);
        if constexpr (std::is_same<T, std::string>::value) {
            return s;
        } else if constexpr (std::is_same<T, bool>::value) {
            std::string l = s;
            std::transform(l.begin(), l.end(), l.begin(), ::tolower);
            if (l == "true" || l == "yes" || l == "on") return true;
            if (l == "false" || l == "no" || l == "off") return false;
            throw std::runtime_error("bad bool: " + s);
        } else {
            std::istringstream is(s);
            T v;
            is >> v;
            if (is.fail()) throw std::runtime_error("bad scalar: " + s);
            return v;
        }
    }

   public:
    Node() {}
    static Node fromFile(const std::string& path) {
        std::ifstream f(path);
        if (!f) throw std::runtime_error("cannot open " + path);
        Node n;
        n.defined_ = true;
        std::string line;
        while (std::getline(f, line)) {
            const size_t h = line.find('#');
            if (h != std::string::npos) line = line.substr(0, h);
            const size_t c = line.find(':');
            if (c == std::string::npos) continue;
            const std::string k = trim(line.substr(0, c)), v = trim(line.substr(c + 1));
            if (!k.empty()) n.kv_[k] = v;
        }
        return n;
    }
    Node operator[](const std::string& key) const {
        Node n;
        auto it = kv_.find(key);
        if (it != kv_.end()) n.defined_ = true, n.scalar_ = it->second;
        return n;
    }
    explicit operator bool() const { return defined_; }
    bool IsDefined() const { return defined_; }
    template <class T>
    T as() const {
        if (!defined_) throw std::runtime_error("undefined node");
        if constexpr (std::is_same<T, std::vector<double>>::value || std::is_same<T, std::vector<int>>::value ||
                      std::is_same<T, std::vector<float>>::value) {
            std::string s = trim(scalar_);
            if (s.empty() || s.front() != '[' || s.back() != ']') throw std::runtime_error("not a list: " + s);
            T out;
            std::istringstream is(s.substr(1, s.size() - 2));
            std::string item;
            while (std::getline(is, item, ',')) out.push_back(conv<typename T::value_type>(item));
            return out;
        } else {
            return conv<T>(scalar_);
        }
    }
};
inline Node LoadFile(const std::string& path) { return Node::fromFile(path); }
}  // namespace YAML
#endif
