// Environment-map side of the scene: Radiance RGBE (.hdr) codec and the
// importance-sampling cache the path tracer draws sky directions from.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace rta {

// Decodes a Radiance picture ("#?RADIANCE" header, "-Y h +X w" resolution line,
// new-style RLE, old-style RLE or flat scanlines) into w*h*3 floats, first file
// scanline first.  Component value = mantissa/256 * 2^(exponent-128), the convention
// of the reference's loader (include/hdrloader.cpp:99-116), which differs from the
// usual (m+0.5)/256.  Returns false and sets err on malformed input.
bool rgbe_decode(const uint8_t *bytes, size_t n, int &w, int &h, std::vector<float> &rgb, std::string &err);

// Encodes w*h*3 floats as a flat (non-RLE) Radiance picture that rgbe_decode and the
// reference loader both read.
void rgbe_encode(const float *rgb, int w, int h, std::vector<uint8_t> &bytes);

// src/texture/Skybox.cpp:12-53.  cache[(i*w+j)*3 + {0,1,2}] = (X/w, Y/h, pdf[Y][X]/lightSum)
// where (Y, X) is the texel whose cumulative luminance first reaches the fraction
// (i/h, j/w); luminance weight 0.2r + 0.7g + 0.1b accumulated in double.
void sky_build_cache(const float *rgb, int w, int h, std::vector<float> &cache, double &light_sum);

}  // namespace rta
