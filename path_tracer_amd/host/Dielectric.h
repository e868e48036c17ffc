// Dielectric.h — transparent material (reference interface: Dielectric.h:17-36).
#ifndef PT_HOST_DIELECTRIC_H
#define PT_HOST_DIELECTRIC_H
#include "Material.h"

class Dielectric : public Material {
    double m_refractiveIndex;

public:
    Dielectric(double inIndex) : m_refractiveIndex{inIndex} {}
    MaterialDesc describe() const override { return MaterialDesc{PT_DIELECTRIC, {1.0, 1.0, 1.0}, 0.0, m_refractiveIndex}; }
};
#endif
