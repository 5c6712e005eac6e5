// Math.h — vector types of the host API.  The reference uses glm::dvec2 / glm::dvec3 in its public
// surface (Source/Hittable.h:10-12, Source/Camera.h:8-9).  When glm is on the include path it is
// used as-is; otherwise a minimal layout-compatible stand-in (x,y,z doubles) is provided so the host
// API builds on machines without glm (this image has none).
#pragma once
#include <cmath>

#if defined(POORAYTRACER_USE_GLM) || (__has_include(<glm/vec3.hpp>) && !defined(POORAYTRACER_NO_GLM))
#include <glm/vec2.hpp>
#include <glm/vec3.hpp>
namespace Pooraytracer {
using dvec2 = glm::dvec2;
using dvec3 = glm::dvec3;
} // namespace Pooraytracer
#else
namespace Pooraytracer {
struct dvec2 {
    double x, y;
    dvec2() : x(0), y(0) {}
    dvec2(double x_, double y_) : x(x_), y(y_) {}
    explicit dvec2(double s) : x(s), y(s) {}
    double& operator[](int i) { return i == 0 ? x : y; }
    double operator[](int i) const { return i == 0 ? x : y; }
};
struct dvec3 {
    union { double x, r; };
    union { double y, g; };
    union { double z, b; };
    dvec3() : x(0), y(0), z(0) {}
    dvec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    explicit dvec3(double s) : x(s), y(s), z(s) {}
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline dvec3 operator+(const dvec3& a, const dvec3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline dvec3 operator-(const dvec3& a, const dvec3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline dvec3 operator-(const dvec3& a) { return {-a.x, -a.y, -a.z}; }
inline dvec3 operator*(const dvec3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline dvec3 operator*(double s, const dvec3& a) { return {s * a.x, s * a.y, s * a.z}; }
inline dvec3 operator/(const dvec3& a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline dvec3& operator+=(dvec3& a, const dvec3& b) { a = a + b; return a; }
} // namespace Pooraytracer
#endif

namespace Pooraytracer {
using vec3 = dvec3;
using vec2 = dvec2;
using point3 = dvec3;
using color = dvec3;
} // namespace Pooraytracer
