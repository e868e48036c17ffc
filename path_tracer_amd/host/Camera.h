// Camera.h — thin-lens camera (reference interface: Camera.h:17-79). The frame set-up
// (Camera.cpp:7-32) stays on the host in double: it runs once and its seven results are the
// constants of the device's ray generation. getCurrentRay is answered by the device (pt_gen_ray).
#ifndef PT_HOST_CAMERA_H
#define PT_HOST_CAMERA_H

#include <cmath>

#include "DeviceSession.h"
#include "Material.h"
#include "Ray.h"

class Camera final {
    using Vec3D = dp::PhysicsVector<3>;

    Vec3D m_cameraPosition{0, 0, 0};
    Vec3D m_cameraLookingAt{0, 0, -1};
    Vec3D m_cameraUpOrientation{0, 1, 0};
    Vec3D m_lowerLeftCorner, m_horizontalDirection, m_verticalDirection;
    Vec3D m_alongViewLine, m_positiveHorizontal, m_positiveVertical;
    double m_aspectRatio{16.0 / 9.0};
    double m_focalLength{1};
    double m_fieldOfView{60};
    double m_viewportHeight{0}, m_viewportWidth{0};
    double m_apertureSize{0.1};
    double m_focusDistance{10};
    double m_lensRadius{0};
    mutable bool m_usedDevice = false;

    static constexpr double pi = 3.14159265358979;     // the reference's truncated value (Camera.h:45)

    void setupCamera() {
        const double theta = m_fieldOfView * pi / 180.0;
        const double h = m_focalLength * std::tan(theta / 2);
        m_viewportHeight = 2 * h;
        m_viewportWidth = m_aspectRatio * m_viewportHeight;
        m_alongViewLine = (m_cameraPosition - m_cameraLookingAt).getUnitVector();
        m_positiveHorizontal = Vec3D::vectorProduct(m_cameraUpOrientation, m_alongViewLine).getUnitVector();
        m_positiveVertical = Vec3D::vectorProduct(m_alongViewLine, m_positiveHorizontal).getUnitVector();
        m_horizontalDirection = m_positiveHorizontal.scaledBy(m_viewportWidth).scaledBy(m_focusDistance);
        m_verticalDirection = m_positiveVertical.scaledBy(m_viewportHeight).scaledBy(m_focusDistance);
        m_lowerLeftCorner = m_cameraPosition - m_horizontalDirection.scaledBy(0.5) - m_verticalDirection.scaledBy(0.5) -
                            m_alongViewLine.scaledBy(m_focalLength).scaledBy(m_focusDistance);
        m_lensRadius = m_apertureSize / 2;
    }

public:
    Camera() { setupCamera(); }
    Camera(Vec3D inPosition, Vec3D inLookingAt, Vec3D inUp, double inRatio, double inFocalLength, double inVFoV,
           double inAperture, double inFocusDist)
        : m_cameraPosition{inPosition}, m_cameraLookingAt{inLookingAt}, m_cameraUpOrientation{inUp}, m_aspectRatio{inRatio},
          m_focalLength{inFocalLength}, m_fieldOfView{inVFoV}, m_apertureSize{inAperture}, m_focusDistance{inFocusDist} {
        setupCamera();
    }
    ~Camera() { if (m_usedDevice) dp::DeviceSession::instance().forgetCamera(this); }

    // position, lowerLeft, horizontal, vertical, u, v (3 each), lensRadius: what pt_set_camera takes.
    void deviceConstants(double out[19]) const {
        const Vec3D* v[6] = {&m_cameraPosition, &m_lowerLeftCorner, &m_horizontalDirection, &m_verticalDirection,
                             &m_positiveHorizontal, &m_positiveVertical};
        for (int i = 0; i < 6; ++i) { out[3 * i] = v[i]->x(); out[3 * i + 1] = v[i]->y(); out[3 * i + 2] = v[i]->z(); }
        out[18] = m_lensRadius;
    }

    // One ray through the normalised viewport point (inputX, inputY); the lens point is drawn on
    // the host by rejection like VectorFunc.cpp:57-64 and handed to the device.
    Ray getCurrentRay(double inputX, double inputY) const {
        double dx, dy;
        do {
            dx = 2 * dp::hostUniform() - 1; dy = 2 * dp::hostUniform() - 1; (void)dp::hostUniform();
        } while (dx * dx + dy * dy >= 1);
        return getCurrentRay(inputX, inputY, dx, dy);
    }
    Ray getCurrentRay(double inputX, double inputY, double diskX, double diskY) const {
        dp::DeviceSession& s = dp::DeviceSession::instance();
        double c[19];
        deviceConstants(c);
        s.bindCamera(this, c);
        m_usedDevice = true;
        // width = height = 2 makes (i + u) / (W - 1) equal to u for pixel 0
        const int zero = 0;
        double o[3], d[3];
        s.check(pt_gen_ray(s.handle(), 1, 2, 2, &zero, &zero, &inputX, &inputY, &diskX, &diskY, o, d), "pt_gen_ray");
        return Ray(Vec3D{o[0], o[1], o[2]}, Vec3D{d[0], d[1], d[2]});
    }

    double getHeight() const { return m_viewportHeight; }
    double getWidth() const { return m_viewportWidth; }
    double getRatio() const { return m_aspectRatio; }
};
#endif
