// Material.h — mirror of the material classes of Source/Material.h (constructors and the
// HasEmission / GetEmission / SkipLightSampling queries).  Scatter / Sample / Eval / PDF are the
// per-bounce hot path (Material.h:101-521) and run on the device only (prt_device.h mat_scatter /
// mat_eval); the host classes export their parameters through Describe().
#pragma once
#include <memory>

#include "../prt.h"
#include "Math.h"
#include "Texture.h"

namespace Pooraytracer {
using std::make_shared;
using std::shared_ptr;

enum class MaterialType { Lambertian, PhoneReflectance, PerfectMirror, CookTorrance, DiffuseLight, DebugMaterial, Empty };

class Material {
public:
    virtual ~Material() = default;
    virtual bool HasEmission() const { return false; }
    virtual color GetEmission() const { return color(0., 0., 0.); }
    virtual bool SkipLightSampling() const { return false; }
    // parameters for the device material table; *tex receives the Kd map (or stays null)
    virtual void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const = 0;

protected:
    static void Zero(PrtMaterial& m, int type);
    static void Put(double* dst, const color& c) { dst[0] = c.x; dst[1] = c.y; dst[2] = c.z; }
    static color Solid(const shared_ptr<Texture>& t) {
        auto s = std::dynamic_pointer_cast<SolidColor>(t);
        return s ? s->Albedo() : color(0., 1., 1.);
    }
};

class Lambertian : public Material {
public:
    Lambertian(const color& albedo) : texture(make_shared<SolidColor>(albedo)) {}
    Lambertian(shared_ptr<Texture> texture_) : texture(texture_) {}
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;

private:
    shared_ptr<Texture> texture;
};

class DiffuseLight : public Material {
public:
    DiffuseLight(shared_ptr<Texture> texture_) : texture(texture_) {}
    DiffuseLight(const color& emit) : texture(make_shared<SolidColor>(emit)) {}
    bool HasEmission() const override { return true; }
    color GetEmission() const override { return Solid(texture); } // Emmited(0,0,p) of a SolidColor
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;

private:
    shared_ptr<Texture> texture;
};

class PhoneReflectance : public Material {
public:
    PhoneReflectance(const color& Kd_, const color& Ks_, double Ns_)
        : Kd(make_shared<SolidColor>(Kd_)), Ks(make_shared<SolidColor>(Ks_)), Ns(Ns_) {}
    // the reference stores mapKd in BOTH Kd and Ks (Material.h:178-181)
    PhoneReflectance(shared_ptr<Texture> mapKd, const color& /*Ks*/, double Ns_) : Kd(mapKd), Ks(mapKd), Ns(Ns_) {}
    bool SkipLightSampling() const override { return Ns > 1.; }
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;

private:
    shared_ptr<Texture> Kd, Ks;
    double Ns;
};

class PerfectMirror : public Material {
public:
    bool SkipLightSampling() const override { return true; }
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;
};

class CookTorrance : public Material {
public:
    CookTorrance(const color& Kd, double alphaX_ = 0.3, double alphaY_ = 0.3, vec3 eta_ = vec3(1.0), vec3 k_ = vec3(0.0))
        : eta(eta_), k(k_), alphaX(alphaX_), alphaY(alphaY_), texture(make_shared<SolidColor>(Kd)) {}
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;

private:
    vec3 eta, k;
    double alphaX, alphaY;
    shared_ptr<Texture> texture;
};

class DebugMaterial : public Material {
public:
    DebugMaterial(shared_ptr<Texture> texture_) : texture(texture_) {}
    DebugMaterial(const color& albedo) : texture(make_shared<SolidColor>(albedo)) {}
    bool HasEmission() const override { return true; }
    color GetEmission() const override { return Solid(texture); }
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;

private:
    shared_ptr<Texture> texture;
};

class EmptyMaterial : public Material {
public:
    bool SkipLightSampling() const override { return true; }
    void Describe(PrtMaterial& out, shared_ptr<Texture>* tex) const override;
};
} // namespace Pooraytracer
