// Ray.h — origin and direction (reference interface: Ray.h:14-32). On the device a ray is two
// fp32 triples in registers.
#ifndef PT_HOST_RAY_H
#define PT_HOST_RAY_H

#include "PhysicsVector.h"

class Ray {
    dp::PhysicsVector<3> m_origin;
    dp::PhysicsVector<3> m_direction;

public:
    Ray() = default;
    Ray(const dp::PhysicsVector<3>& origin, const dp::PhysicsVector<3>& direction) : m_origin{origin}, m_direction{direction} {}

    const dp::PhysicsVector<3>& origin() const { return m_origin; }
    const dp::PhysicsVector<3>& direction() const { return m_direction; }
    dp::PhysicsVector<3> at(double distance) const { return m_origin + m_direction.scaledBy(distance); }
};
#endif
