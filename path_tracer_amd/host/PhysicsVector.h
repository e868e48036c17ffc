// PhysicsVector.h — host-side N-vector of double with the member names the reference's call
// sites use (the original header lives in the author's un-vendored Basic-Utilities library and
// is absent from the reference; this is an independent implementation of the subset listed in
// SURVEY.md 8(c)). On the device the same quantity is three fp32 registers (csrc/pt_device.h).
#ifndef PT_HOST_PHYSICSVECTOR_H
#define PT_HOST_PHYSICSVECTOR_H

#include <array>
#include <cmath>
#include <cstddef>
#include <type_traits>

namespace dp {

template <std::size_t N>
class PhysicsVector {
    std::array<double, N> m_v{};   // zero-filled by default (Ray.h:22, Path Tracer.cpp:266 rely on it)

public:
    PhysicsVector() = default;

    // { 8, 2, 3 } with ints and doubles mixed (Path Tracer.cpp:121, :182, :218)
    template <class... T, class = std::enable_if_t<sizeof...(T) == N && (std::is_arithmetic_v<T> && ...)>>
    PhysicsVector(T... vals) : m_v{static_cast<double>(vals)...} {}

    double& x() { return m_v[0]; }
    double& y() { static_assert(N > 1); return m_v[1]; }
    double& z() { static_assert(N > 2); return m_v[2]; }
    double x() const { return m_v[0]; }
    double y() const { static_assert(N > 1); return m_v[1]; }
    double z() const { static_assert(N > 2); return m_v[2]; }

    double operator[](std::size_t i) const { return m_v[i]; }
    void setAt(std::size_t i, double value) { m_v[i] = value; }
    static constexpr std::size_t dimension() { return N; }

    PhysicsVector operator+(const PhysicsVector& o) const { PhysicsVector r; for (std::size_t i = 0; i < N; ++i) r.m_v[i] = m_v[i] + o.m_v[i]; return r; }
    PhysicsVector operator-(const PhysicsVector& o) const { PhysicsVector r; for (std::size_t i = 0; i < N; ++i) r.m_v[i] = m_v[i] - o.m_v[i]; return r; }
    PhysicsVector operator-() const { PhysicsVector r; for (std::size_t i = 0; i < N; ++i) r.m_v[i] = -m_v[i]; return r; }
    PhysicsVector& operator+=(const PhysicsVector& o) { for (std::size_t i = 0; i < N; ++i) m_v[i] += o.m_v[i]; return *this; }
    PhysicsVector& operator-=(const PhysicsVector& o) { for (std::size_t i = 0; i < N; ++i) m_v[i] -= o.m_v[i]; return *this; }
    bool operator==(const PhysicsVector& o) const { return m_v == o.m_v; }

    PhysicsVector scaledBy(double s) const { PhysicsVector r; for (std::size_t i = 0; i < N; ++i) r.m_v[i] = m_v[i] * s; return r; }
    double innerProduct(const PhysicsVector& o) const { double s = 0; for (std::size_t i = 0; i < N; ++i) s += m_v[i] * o.m_v[i]; return s; }
    double lengthSquared() const { return innerProduct(*this); }
    double length() const { return std::sqrt(lengthSquared()); }
    PhysicsVector getUnitVector() const { const double l = length(); PhysicsVector r; for (std::size_t i = 0; i < N; ++i) r.m_v[i] = m_v[i] / l; return r; }

    static PhysicsVector vectorProduct(const PhysicsVector& a, const PhysicsVector& b) {
        static_assert(N == 3, "the vector product is defined for three dimensions");
        return PhysicsVector{a.m_v[1] * b.m_v[2] - a.m_v[2] * b.m_v[1], a.m_v[2] * b.m_v[0] - a.m_v[0] * b.m_v[2],
                             a.m_v[0] * b.m_v[1] - a.m_v[1] * b.m_v[0]};
    }
};

}   // namespace dp
#endif
