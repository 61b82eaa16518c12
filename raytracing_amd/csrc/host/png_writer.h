// Minimal PNG encoder (8-bit gray / RGB / RGBA, stored deflate blocks): what
// SSBOBuffer::save_as_image needs from cv::imwrite (src/SSBOBuffer.h:41-62).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rta {

// pixels: h rows of w*c bytes, top row first.  Returns false and sets err on failure.
bool write_png(const std::string &path, int w, int h, int c, const uint8_t *pixels, std::string &err);

// The float -> byte step of save_as_image: min(255, int(x * 255)) truncated to a byte, rows
// flipped (the buffers hold the bottom row first, images the top row first).  NaN maps to 0
// (the reference's int(NaN) is undefined).  out: h*w*c bytes.
void quantize_flip(const float *data, int w, int h, int c, std::vector<uint8_t> &out);

}  // namespace rta
