#include "png_writer.h"
#include <cstdio>
#include <cstring>

namespace rta {
namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
    for(uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for(int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
    crc_ready = true;
}
uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if(!crc_ready) crc_init();
    crc = ~crc;
    for(size_t i = 0; i < n; i++) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void put32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t> &file, const char type[4], const std::vector<uint8_t> &body) {
    put32(file, (uint32_t)body.size());
    const size_t at = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), body.begin(), body.end());
    put32(file, crc32(0, file.data() + at, 4 + body.size()));
}

}  // namespace

bool write_png(const std::string &path, int w, int h, int c, const uint8_t *pixels, std::string &err) {
    if(w <= 0 || h <= 0 || (c != 1 && c != 3 && c != 4) || !pixels) {
        err = "write_png: width/height must be positive and channels 1, 3 or 4";
        return false;
    }
    // scanlines with filter byte 0
    const size_t row = (size_t)w * c;
    std::vector<uint8_t> raw((row + 1) * (size_t)h);
    for(int y = 0; y < h; y++) {
        raw[(row + 1) * y] = 0;
        memcpy(&raw[(row + 1) * y + 1], pixels + row * y, row);
    }
    // zlib stream of stored blocks
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for(size_t off = 0; off < raw.size() || off == 0;) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        const bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for(size_t i = 0; i < n; i++) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        off += n;
        if(last) break;
    }
    put32(z, (b << 16) | a);

    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w);
    put32(ihdr, (uint32_t)h);
    ihdr.push_back(8);
    ihdr.push_back(c == 1 ? 0 : (c == 3 ? 2 : 6));
    ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(file, "IHDR", ihdr);
    chunk(file, "IDAT", z);
    chunk(file, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if(!f) {
        err = "write_png: cannot open " + path;
        return false;
    }
    const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
    fclose(f);
    if(!ok) err = "write_png: short write to " + path;
    return ok;
}

void quantize_flip(const float *data, int w, int h, int c, std::vector<uint8_t> &out) {
    const size_t row = (size_t)w * c;
    out.resize(row * (size_t)h);
    for(int y = 0; y < h; y++) {
        const float *src = data + row * (size_t)(h - 1 - y);
        uint8_t *dst = out.data() + row * (size_t)y;
        for(size_t i = 0; i < row; i++) {
            const float x = src[i] * 255.0f;
            int v;
            if(x != x) v = 0;
            else if(x >= 2147483520.0f) v = 255;           // int(x) would overflow; min(255, .) of the saturated value
            else if(x <= -2147483648.0f) v = (int)0x80000000;
            else v = (int)x;
            if(v > 255) v = 255;
            dst[i] = (uint8_t)v;
        }
    }
}

}  // namespace rta
