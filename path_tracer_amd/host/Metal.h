// Metal.h — reflective material with fuzz (reference interface: Metal.h:12-24; the constructor
// clamps fuzz to at most 1 like Metal.h:21).
#ifndef PT_HOST_METAL_H
#define PT_HOST_METAL_H
#include "Material.h"

class Metal : public Material {
    dp::PhysicsVector<3> m_albedoColour;
    double m_fuzz;

public:
    Metal(const dp::PhysicsVector<3>& inColour, double inFuzz) : m_albedoColour{inColour}, m_fuzz{inFuzz < 1 ? inFuzz : 1} {}
    MaterialDesc describe() const override {
        return MaterialDesc{PT_METAL, {m_albedoColour.x(), m_albedoColour.y(), m_albedoColour.z()}, m_fuzz, 0.0};
    }
};
#endif
