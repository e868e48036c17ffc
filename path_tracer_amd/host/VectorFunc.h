// VectorFunc.h — host-side vector helpers that stay on the host: reading "(X,Y,Z)" from the
// config file (reference interface: VectorFunc.h:62-114) and the random draws the SCENE
// GENERATOR makes (VectorFunc.cpp:31-35). The scattering helpers of the reference's VectorFunc
// (sphere point, reflect, refract) live on the device: csrc/pt_device.h.
#ifndef PT_HOST_VECTORFUNC_H
#define PT_HOST_VECTORFUNC_H

#include <cstdlib>
#include <random>
#include <string>
#include <string_view>

#include "PhysicsVector.h"

namespace dp {

namespace detail {
inline bool isDigit(char c) { return c >= '0' && c <= '9'; }
// digits* '.'? digits*   (needDigit: at least one digit before the optional point)
inline bool isUnsignedDecimal(std::string_view s, bool needDigit) {
    std::size_t i = 0, lead = 0;
    while (i < s.size() && isDigit(s[i])) { ++i; ++lead; }
    if (needDigit && lead == 0) return false;
    if (i < s.size() && s[i] == '.') ++i;
    while (i < s.size() && isDigit(s[i])) ++i;
    return i == s.size();
}
inline bool isOpen(char c) { return c == '{' || c == '[' || c == '(' || c == '<'; }
inline bool isClose(char c) { return c == '}' || c == ']' || c == ')' || c == '>'; }
}   // namespace detail

// Reads "(X,Y,Z)" (any of ( [ { < or none) into inVector. Like the reference's validator
// (VectorFunc.h:65) it accepts UNSIGNED decimals only: a sign, an exponent or a space makes the
// call fail, zero the vector and return false (VectorFunc.h:67-77).
template <std::size_t dim>
bool readVector(std::string_view input, PhysicsVector<dim>& inVector) {
    auto reject = [&]() { inVector = PhysicsVector<dim>{}; return false; };
    std::string_view s = input;
    if (!s.empty() && detail::isOpen(s.front())) s.remove_prefix(1);
    if (!s.empty() && detail::isClose(s.back())) s.remove_suffix(1);
    // split at commas; every term but the last may be empty or lack digits, the last needs a digit first
    std::size_t commas = 0, start = 0;
    double vals[dim > 0 ? dim : 1] = {};
    for (std::size_t i = 0; i <= s.size(); ++i) {
        if (i < s.size() && s[i] != ',') continue;
        const std::string_view term = s.substr(start, i - start);
        const bool last = (i == s.size());
        if (!detail::isUnsignedDecimal(term, last)) return reject();
        if (commas < dim) vals[commas] = std::strtod(std::string(term).c_str(), nullptr);
        if (!last) ++commas;
        start = i + 1;
    }
    if (commas != dim - 1) return reject();
    for (std::size_t i = 0; i < dim; ++i) inVector.setAt(i, vals[i]);
    return true;
}

template <std::size_t dim>
PhysicsVector<dim> readVector(std::string_view input) {
    PhysicsVector<dim> out{};
    readVector(input, out);
    return out;
}

// Three draws from U(inMin, inMax), x first (VectorFunc.cpp:31-35), from the given engine.
template <class Engine>
PhysicsVector<3> randVector(Engine& engine, double inMin, double inMax) {
    std::uniform_real_distribution<double> distribution{inMin, inMax};
    const double a = distribution(engine);
    const double b = distribution(engine);
    const double c = distribution(engine);
    return PhysicsVector<3>{a, b, c};
}

}   // namespace dp
#endif
