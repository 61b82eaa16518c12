#include "sky_host.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace rta {
namespace {

struct Cursor {
    const uint8_t *p;
    const uint8_t *end;
    bool eof() const { return p >= end; }
    int get() { return p < end ? *p++ : -1; }
};

// Flat pixels, or the old run-length form in which a (1,1,1,count) pixel repeats the
// previous pixel `count << shift` times (consecutive markers shift by 8 more bits).
bool read_flat_or_old_rle(Cursor &c, uint8_t (*px)[4], int len, int done) {
    int shift = 0;
    while(done < len) {
        if(c.end - c.p < 4) return false;
        uint8_t q[4] = {c.p[0], c.p[1], c.p[2], c.p[3]};
        c.p += 4;
        if(q[0] == 1 && q[1] == 1 && q[2] == 1) {
            if(done == 0) return false;
            long count = (long)q[3] << shift;
            for(long k = 0; k < count && done < len; k++, done++) memcpy(px[done], px[done - 1], 4);
            shift += 8;
        } else {
            memcpy(px[done++], q, 4);
            shift = 0;
        }
    }
    return true;
}

bool read_scanline(Cursor &c, uint8_t (*px)[4], int len) {
    if(len < 8 || len > 0x7fff) return read_flat_or_old_rle(c, px, len, 0);
    if(c.eof()) return false;
    if(*c.p != 2) return read_flat_or_old_rle(c, px, len, 0);
    if(c.end - c.p < 4) return false;
    uint8_t g = c.p[1], b = c.p[2], e = c.p[3];
    c.p += 4;
    if(g != 2 || (b & 128)) {
        // not an RLE marker after all: it was an ordinary first pixel
        px[0][0] = 2; px[0][1] = g; px[0][2] = b; px[0][3] = e;
        return read_flat_or_old_rle(c, px, len, 1);
    }
    for(int comp = 0; comp < 4; comp++) {
        int j = 0;
        while(j < len) {
            int code = c.get();
            if(code < 0) return false;
            if(code > 128) {
                int run = code & 127;
                int val = c.get();
                if(val < 0) return false;
                while(run-- > 0 && j < len) px[j++][comp] = (uint8_t)val;
            } else {
                int run = code;
                while(run-- > 0 && j < len) {
                    int val = c.get();
                    if(val < 0) return false;
                    px[j++][comp] = (uint8_t)val;
                }
            }
        }
    }
    return true;
}

}  // namespace

bool rgbe_decode(const uint8_t *bytes, size_t n, int &w, int &h, std::vector<float> &rgb, std::string &err) {
    static const char magic[] = "#?RADIANCE";
    if(n < 11 || memcmp(bytes, magic, 10) != 0) { err = "not a Radiance picture (missing #?RADIANCE)"; return false; }
    Cursor c{bytes + 11, bytes + n};
    // header ends at the first empty line
    int prev = 0;
    for(;;) {
        int ch = c.get();
        if(ch < 0) { err = "truncated header"; return false; }
        if(ch == '\n' && prev == '\n') break;
        prev = ch;
    }
    char reso[128];
    int k = 0;
    for(;;) {
        int ch = c.get();
        if(ch < 0) { err = "truncated resolution line"; return false; }
        if(ch == '\n' || k >= 127) break;
        reso[k++] = (char)ch;
    }
    reso[k] = 0;
    if(sscanf(reso, "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) {
        err = std::string("unsupported resolution line: ") + reso;
        return false;
    }
    rgb.assign((size_t)w * h * 3, 0.0f);
    std::vector<uint8_t> line((size_t)w * 4);
    uint8_t(*px)[4] = reinterpret_cast<uint8_t(*)[4]>(line.data());
    for(int y = 0; y < h; y++) {
        if(!read_scanline(c, px, w)) { err = "truncated pixel data"; return false; }
        float *o = rgb.data() + (size_t)y * w * 3;
        for(int x = 0; x < w; x++) {
            float d = std::ldexp(1.0f, (int)px[x][3] - 128);
            o[x * 3 + 0] = (px[x][0] / 256.0f) * d;
            o[x * 3 + 1] = (px[x][1] / 256.0f) * d;
            o[x * 3 + 2] = (px[x][2] / 256.0f) * d;
        }
    }
    return true;
}

void rgbe_encode(const float *rgb, int w, int h, std::vector<uint8_t> &bytes) {
    char head[128];
    int hn = snprintf(head, sizeof head, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    bytes.assign(head, head + hn);
    bytes.reserve(hn + (size_t)w * h * 4);
    for(size_t i = 0; i < (size_t)w * h; i++) {
        float r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        float v = std::max(r, std::max(g, b));
        uint8_t q[4] = {0, 0, 0, 0};
        if(v >= 1e-32f) {
            int e;
            float m = std::frexp(v, &e);      // v = m * 2^e, m in [0.5, 1)
            float s = m * 256.0f / v;
            q[0] = (uint8_t)(r * s);
            q[1] = (uint8_t)(g * s);
            q[2] = (uint8_t)(b * s);
            q[3] = (uint8_t)(e + 128);
        }
        // a flat scanline must not start with the byte pattern of an RLE marker
        if(w >= 8 && w <= 0x7fff && i % w == 0 && q[0] == 2 && q[1] == 2 && !(q[2] & 128)) q[0] = 3;
        bytes.insert(bytes.end(), q, q + 4);
    }
}

void sky_build_cache(const float *rgb, int w, int h, std::vector<float> &cache, double &light_sum) {
    std::vector<double> pdf((size_t)w * h), spdf((size_t)w * h), FY(h);
    light_sum = 0;
    for(int i = 0; i < h; i++)
        for(int j = 0; j < w; j++) {
            const float *col = rgb + ((size_t)i * w + j) * 3;
            double p = col[0] * 0.2 + col[1] * 0.7 + col[2] * 0.1;
            pdf[(size_t)i * w + j] = p;
            light_sum += p;
        }
    for(int i = 0; i < h; i++) {
        double *row = spdf.data() + (size_t)i * w;
        const double *prow = pdf.data() + (size_t)i * w;
        for(int j = 0; j < w; j++) row[j] = j ? prow[j] + row[j - 1] : prow[j];
        FY[i] = row[w - 1];
        if(i) FY[i] += FY[i - 1];
    }
    cache.assign((size_t)w * h * 3, 0.0f);
    for(int i = 0; i < h; i++)
        for(int j = 0; j < w; j++) {
            double u = (float)i / h;  // quotient formed in float, then widened
            double v = (float)j / w;
            int Y = (int)(std::lower_bound(FY.begin(), FY.end(), u * light_sum) - FY.begin());
            const double *row = spdf.data() + (size_t)Y * w;
            int X = (int)(std::lower_bound(row, row + w, v * row[w - 1]) - row);
            float *o = cache.data() + ((size_t)i * w + j) * 3;
            o[0] = (float)X / w;
            o[1] = (float)Y / h;
            o[2] = (float)(pdf[(size_t)Y * w + X] / light_sum);
        }
}

}  // namespace rta
