// image_io.hpp -- minimal raster input for the headless harnesses: binary PPM (P6) or raw RGBA8
// ("file.rgba:WxH").  The reference loads PNGs through SDL_image (software/triangulate/main.cpp:40);
// tools/png2ppm.py converts.  Optional GL_LINEAR-style resampling reproduces the reference's habit of
// rasterising into a window of image/1.5 (software/triangulate/main.cpp:53).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct Raster {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;
};

inline bool load_raster(const std::string& spec, Raster& img) {
    const size_t colon = spec.rfind(':');
    if (colon != std::string::npos && spec.find('x', colon) != std::string::npos) {  // raw: path:WxH
        const std::string path = spec.substr(0, colon);
        if (std::sscanf(spec.c_str() + colon + 1, "%dx%d", &img.w, &img.h) != 2) return false;
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        img.rgba.resize((size_t)img.w * img.h * 4);
        const size_t n = std::fread(img.rgba.data(), 1, img.rgba.size(), f);
        std::fclose(f);
        return n == img.rgba.size();
    }
    FILE* f = std::fopen(spec.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0;
    if (std::fscanf(f, "%2s", magic) != 1 || std::string(magic) != "P6") { std::fclose(f); return false; }
    int vals[3], got = 0;
    while (got < 3) {  // width, height, maxval with '#' comments
        int c = std::fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
        if (c == EOF) { std::fclose(f); return false; }
        if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        std::ungetc(c, f);
        if (std::fscanf(f, "%d", &vals[got]) != 1) { std::fclose(f); return false; }
        got++;
    }
    std::fgetc(f);  // single whitespace after maxval
    img.w = vals[0]; img.h = vals[1]; maxv = vals[2];
    if (maxv != 255 || img.w < 1 || img.h < 1) { std::fclose(f); return false; }
    std::vector<uint8_t> rgb((size_t)img.w * img.h * 3);
    const size_t n = std::fread(rgb.data(), 1, rgb.size(), f);
    std::fclose(f);
    if (n != rgb.size()) return false;
    img.rgba.resize((size_t)img.w * img.h * 4);
    for (size_t i = 0; i < (size_t)img.w * img.h; i++) {
        img.rgba[4 * i] = rgb[3 * i]; img.rgba[4 * i + 1] = rgb[3 * i + 1]; img.rgba[4 * i + 2] = rgb[3 * i + 2];
        img.rgba[4 * i + 3] = 255;
    }
    return true;
}

// bilinear resample (clamp to edge, texel centres at +0.5) into a w2 x h2 raster -- what sampling an
// RGBA8 GL_LINEAR texture at the centre of every pixel of a smaller window yields, rounded to 8 bits
inline Raster resample(const Raster& src, int w2, int h2) {
    Raster dst;
    dst.w = w2; dst.h = h2;
    dst.rgba.resize((size_t)w2 * h2 * 4);
    for (int y = 0; y < h2; y++)
        for (int x = 0; x < w2; x++) {
            const float u = ((float)x + 0.5f) / (float)w2 * (float)src.w - 0.5f;
            const float v = ((float)y + 0.5f) / (float)h2 * (float)src.h - 0.5f;
            const int x0 = (int)std::floor(u), y0 = (int)std::floor(v);
            const float fx = u - (float)x0, fy = v - (float)y0;
            auto at = [&](int xx, int yy, int c) {
                xx = xx < 0 ? 0 : xx >= src.w ? src.w - 1 : xx;
                yy = yy < 0 ? 0 : yy >= src.h ? src.h - 1 : yy;
                return (float)src.rgba[((size_t)yy * src.w + xx) * 4 + c];
            };
            for (int c = 0; c < 4; c++) {
                const float top = at(x0, y0, c) * (1 - fx) + at(x0 + 1, y0, c) * fx;
                const float bot = at(x0, y0 + 1, c) * (1 - fx) + at(x0 + 1, y0 + 1, c) * fx;
                const float val = top * (1 - fy) + bot * fy;
                dst.rgba[((size_t)y * w2 + x) * 4 + c] = (uint8_t)(val + 0.5f);
            }
        }
    return dst;
}
