// Ray.h — mirror of Source/Ray.h:7-17 (direction is NOT normalised for camera rays).
#pragma once
#include "Math.h"
namespace Pooraytracer {
class Ray {
public:
    Ray() = default;
    Ray(const vec3& origin_, const vec3& direction_) : origin(origin_), direction(direction_) {}
    vec3 operator()(double t) const { return origin + direction * t; }
    vec3 origin;
    vec3 direction;
};
} // namespace Pooraytracer
