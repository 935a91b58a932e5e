// host/image_writer.hpp — 8-bit image output, same entry points as the
// reference (reference pathtracer/png_writer.hpp:8-9): clip() and
// save_png_to_file(Bitmap*, path).  No libpng: the PNG is written with stored
// (uncompressed) deflate blocks; the bytes a decoder returns are identical.
#pragma once
#include <cstdint>

#include "../../../include/hpt/types.h"

namespace hpt {
struct Bitmap {            // reference common.hpp:25-29
    hpt_pixel *pixels;
    int width;
    int height;
};
float clip(float n, float lower, float upper);
// (uint8)(255*clip(c,0,1)) per channel, reference png_writer.cpp:11-17
void convert(const hpt_pixel *p, uint8_t rgb[3]);
int save_png_to_file(const Bitmap *bitmap, const char *path);   // 0 on success
int save_ppm_to_file(const Bitmap *bitmap, const char *path);
int save_png_rgb8(const uint8_t *rgb, int width, int height, const char *path);
}  // namespace hpt
