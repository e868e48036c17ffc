// Lambertian.h — diffuse material (reference interface: Lambertian.h:13-23).
#ifndef PT_HOST_LAMBERTIAN_H
#define PT_HOST_LAMBERTIAN_H
#include "Material.h"

class Lambertian : public Material {
    dp::PhysicsVector<3> m_albedoColour;

public:
    Lambertian(const dp::PhysicsVector<3>& inColour) : m_albedoColour{inColour} {}
    MaterialDesc describe() const override {
        return MaterialDesc{PT_LAMBERTIAN, {m_albedoColour.x(), m_albedoColour.y(), m_albedoColour.z()}, 0.0, 0.0};
    }
};
#endif
