// Texture.h — mirror of Source/Texture.h:12-43.  Texture::Value (Texture.cpp:22-71) is evaluated on
// the device (prt_device.h tex_value); the host classes hold the texels / colour.  Image decoding
// (stb_image in the reference, Texture.cpp:10-21) is out of scope: ImageTexture takes raw 8-bit texels.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "Math.h"

namespace Pooraytracer {
class Texture {
public:
    virtual ~Texture() = default;
    virtual bool IsImage() const { return false; }
};
class SolidColor : public Texture {
public:
    SolidColor(const color& albedo_) : albedo(albedo_) {}
    const color& Albedo() const { return albedo; }

private:
    color albedo;
};
class ImageTexture : public Texture {
public:
    // width*height*channels interleaved 8-bit texels, row 0 first (what stbi_load returns)
    ImageTexture(int width_, int height_, int channels_, const unsigned char* texels)
        : data(std::make_shared<std::vector<unsigned char>>(texels, texels + (size_t)width_ * height_ * channels_)),
          width(width_), height(height_), channels(channels_) {}
    bool IsImage() const override { return true; }
    std::shared_ptr<std::vector<unsigned char>> data;
    int width, height, channels;
};
} // namespace Pooraytracer
