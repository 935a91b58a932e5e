// host/image_writer.cpp — see image_writer.hpp.
#include "image_writer.hpp"

#include <cstdio>
#include <vector>

namespace hpt {

float clip(float n, float lower, float upper) {       // std::max(lower, std::min(n, upper)), png_writer.cpp:7-9
    float mn = (upper < n) ? upper : n;
    return (lower < mn) ? mn : lower;
}

void convert(const hpt_pixel *p, uint8_t rgb[3]) {
    rgb[0] = (uint8_t)(255 * clip(p->red, 0.0f, 1.0f));
    rgb[1] = (uint8_t)(255 * clip(p->green, 0.0f, 1.0f));
    rgb[2] = (uint8_t)(255 * clip(p->blue, 0.0f, 1.0f));
}

static uint32_t crc_table[256];
static void crc_init() {
    if (crc_table[1]) return;
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
}
static uint32_t crc32_update(uint32_t c, const uint8_t *buf, size_t len) {
    for (size_t i = 0; i < len; i++) c = crc_table[(c ^ buf[i]) & 0xff] ^ (c >> 8);
    return c;
}
static void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
static void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32_update(0xffffffffu, out.data() + at, out.size() - at) ^ 0xffffffffu);
}

int save_png_rgb8(const uint8_t *rgb, int width, int height, const char *path) {
    if (!rgb || width <= 0 || height <= 0) return -1;
    crc_init();
    std::vector<uint8_t> raw;                       // filter byte 0 + row
    raw.reserve((size_t)height * (3 * (size_t)width + 1));
    for (int y = 0; y < height; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + 3 * (size_t)width * y, rgb + 3 * (size_t)width * (y + 1));
    }
    std::vector<uint8_t> z;                         // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t pos = 0;
    while (pos < raw.size() || raw.empty()) {
        size_t len = raw.size() - pos; if (len > 65535) len = 65535;
        z.push_back(pos + len == raw.size() ? 1 : 0);
        z.push_back((uint8_t)len); z.push_back((uint8_t)(len >> 8));
        z.push_back((uint8_t)~len); z.push_back((uint8_t)(~len >> 8));
        for (size_t i = 0; i < len; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
        pos += len;
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", std::vector<uint8_t>());
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return -1;
    size_t w = std::fwrite(out.data(), 1, out.size(), fp);
    std::fclose(fp);
    return w == out.size() ? 0 : -1;
}

int save_png_to_file(const Bitmap *bitmap, const char *path) {
    if (!bitmap || !bitmap->pixels) return -1;
    std::vector<uint8_t> rgb((size_t)bitmap->width * bitmap->height * 3);
    for (size_t i = 0; i < (size_t)bitmap->width * bitmap->height; i++) convert(bitmap->pixels + i, rgb.data() + 3 * i);
    return save_png_rgb8(rgb.data(), bitmap->width, bitmap->height, path);
}

int save_ppm_to_file(const Bitmap *bitmap, const char *path) {
    if (!bitmap || !bitmap->pixels) return -1;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return -1;
    std::fprintf(fp, "P6\n%d %d\n255\n", bitmap->width, bitmap->height);
    for (size_t i = 0; i < (size_t)bitmap->width * bitmap->height; i++) {
        uint8_t rgb[3];
        convert(bitmap->pixels + i, rgb);
        std::fwrite(rgb, 1, 3, fp);
    }
    std::fclose(fp);
    return 0;
}

}  // namespace hpt
