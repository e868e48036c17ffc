// Material.h — the abstract material (reference interface: Material.h:14-23). isScattered is
// answered by the device shader (pt_scatter) with three host-drawn uniforms; describe() exports
// the parameters for the flattener.
#ifndef PT_HOST_MATERIAL_H
#define PT_HOST_MATERIAL_H

#include "DeviceSession.h"
#include "Hittable.h"
#include "PhysicsVector.h"
#include "Ray.h"

struct MaterialDesc {
    int kind;                 // PT_LAMBERTIAN, PT_METAL, PT_DIELECTRIC
    double albedo[3];
    double fuzz;
    double ior;
};

class Material {
public:
    virtual ~Material() = default;
    virtual MaterialDesc describe() const = 0;

    virtual bool isScattered(const Ray& inRay, const HitRecord& inRecord, dp::PhysicsVector<3>& inColourAtten,
                             Ray& scatteredRay) const {
        const double u[3] = {dp::hostUniform(), dp::hostUniform(), dp::hostUniform()};
        return isScatteredWith(inRay, inRecord, u, inColourAtten, scatteredRay);
    }

    // The same with the uniforms given: u[0], u[1] pick the point on the unit sphere, u[2] is the Fresnel coin.
    bool isScatteredWith(const Ray& inRay, const HitRecord& inRecord, const double u[3], dp::PhysicsVector<3>& inColourAtten,
                         Ray& scatteredRay) const {
        const MaterialDesc m = describe();
        dp::DeviceSession& s = dp::DeviceSession::instance();
        const double d[3] = {inRay.direction().x(), inRay.direction().y(), inRay.direction().z()};
        const double p[3] = {inRecord.m_point.x(), inRecord.m_point.y(), inRecord.m_point.z()};
        const double n[3] = {inRecord.m_normal.x(), inRecord.m_normal.y(), inRecord.m_normal.z()};
        const int front = inRecord.m_frontFace ? 1 : 0;
        double out[3], atten[3];
        int ok = 0;
        s.check(pt_scatter(s.handle(), 1, &m.kind, m.albedo, &m.fuzz, &m.ior, d, p, n, &front, u, out, atten, &ok), "pt_scatter");
        scatteredRay = Ray(inRecord.m_point, dp::PhysicsVector<3>{out[0], out[1], out[2]});
        inColourAtten = dp::PhysicsVector<3>{atten[0], atten[1], atten[2]};
        return ok != 0;
    }
};
#endif
