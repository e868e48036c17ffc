// ConfigReader.h — reader for the reference's config.txt (the original header is absent from the
// reference; grammar from config.txt:1-23, API from the call sites at Path Tracer.cpp:137-158):
//   one `key = value` per line; `#` starts a comment anywhere on a line; spaces and tabs are
//   ignored everywhere; a missing file, a malformed line, a missing key or an unparsable number
//   throws ConfigException, which the caller answers with "all defaults" (Path Tracer.cpp:158-173).
#ifndef PT_HOST_CONFIGREADER_H
#define PT_HOST_CONFIGREADER_H

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dp {

class ConfigReader {
    std::map<std::string, std::string> m_values;

    static std::string squeeze(const std::string& s) {
        std::string out;
        for (char c : s)
            if (c != ' ' && c != '\t' && c != '\r' && c != '\n') out.push_back(c);
        return out;
    }

public:
    class ConfigException : public std::runtime_error {
    public:
        explicit ConfigException(const std::string& what) : std::runtime_error(what) {}
    };

    ConfigReader() = default;
    explicit ConfigReader(const char* path) { load(path); }
    explicit ConfigReader(const std::string& path) { load(path.c_str()); }

    void load(const char* path) {
        std::ifstream in(path);
        if (!in) throw ConfigException(std::string("cannot open ") + path);
        std::stringstream ss;
        ss << in.rdbuf();
        parse(ss.str());
    }

    // Parses the text of a config file (also the entry point the tests use).
    void parse(const std::string& text) {
        std::istringstream in(text);
        std::string line;
        int lineno = 0;
        while (std::getline(in, line)) {
            ++lineno;
            const std::size_t hash = line.find('#');
            if (hash != std::string::npos) line.erase(hash);
            const std::string s = squeeze(line);
            if (s.empty()) continue;
            const std::size_t eq = s.find('=');
            if (eq == std::string::npos || eq == 0 || eq + 1 >= s.size())
                throw ConfigException("line " + std::to_string(lineno) + " is not of the form key = value");
            m_values[s.substr(0, eq)] = s.substr(eq + 1);
        }
    }

    bool contains(const std::string& key) const { return m_values.count(key) != 0; }

    // The value as text, whitespace and trailing comment already stripped.
    std::string readValue(const std::string& key) const {
        auto it = m_values.find(key);
        if (it == m_values.end()) throw ConfigException("key not found: " + key);
        return it->second;
    }

    // The value converted to the type of the default argument (double or int at the call sites).
    template <class T>
    T readValue(const std::string& key, T /*defaultValue: fixes the return type*/) const {
        static_assert(std::is_arithmetic_v<T>, "numeric defaults only");
        const std::string s = readValue(key);
        char* end = nullptr;
        if constexpr (std::is_integral_v<T>) {
            const long v = std::strtol(s.c_str(), &end, 10);
            if (end == s.c_str() || *end != '\0') throw ConfigException("value of " + key + " is not an integer: " + s);
            return static_cast<T>(v);
        } else {
            const double v = std::strtod(s.c_str(), &end);
            if (end == s.c_str() || *end != '\0') throw ConfigException("value of " + key + " is not a number: " + s);
            return static_cast<T>(v);
        }
    }

    void clear() { m_values.clear(); }
};

}   // namespace dp
#endif
