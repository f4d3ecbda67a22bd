// jpeg_decode.cpp -- incr_decode_jpeg: the JPEG texels the reference's loader sees, as integer arithmetic.
//
// The reference decodes glTF images through tinygltf's built-in loader (src/scene.cpp:40, 98-117: tinygltf::LoadImageData, i.e.
// stbi_load_from_memory with four requested components).  ext/tinygltf is an un-vendored submodule (empty in the reference checkout, pinned
// commit unknown), so this file restates stb_image's PUBLISHED JPEG algorithm (stb_image.h v2.x, public domain; unchanged in the steps below
// since v2.0): texels are bytes, and the contract for bytes is bit-exact -- libjpeg-turbo (PIL), which rounds 1 - 2 codes differently in the
// IDCT, the chroma up-sampling and the colour conversion, is not the reference's decoder.  PARITY UNPINNED: no output of stb_image itself is
// available here; tests/test_jpeg_decode.py holds hand-derived known answers and the +-few-codes agreement with libjpeg-turbo.
//   * entropy decoding: ITU-T T.81 baseline (SOF0 / SOF1) and progressive (SOF2) Huffman, restart intervals -- the coefficients are fixed by the
//     standard; stb keeps them as 16-bit values, dequantised at decode time (baseline) or before the IDCT (progressive): the (short) casts are kept
//   * stbi__idct_block: the integer "islow" IDCT with 12-bit constants, column pass to 2 extra bits (+ 512 >> 10), row pass + 65536 + (128 << 17) >> 17,
//     clamp; a column whose AC terms are all zero takes dc * 4
//   * chroma: stbi__resample_row_hv_2 (3:1 vertically, then 3:1 horizontally, + 8 >> 4), _v_2, _h_2, generic replication, chosen per component
//     from the sampling factors; the vertical neighbour by stb's row stepping (ystep starts at vs >> 1)
//   * stbi__YCbCr_to_RGB_row: 20-bit fixed point, the cb -> g product masked to its upper 16 bits (which makes stb's SIMD kernels produce the
//     same bytes as this scalar form: the reference's x64 build runs those)
// Output: RGBA8, alpha 255; a one-component file is grey replicated (stb's conversion to four requested components).
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../include/incr.h"

namespace {

const uint8_t kDezigzag[64 + 15] = {
    0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct Huffman {                 // canonical code, T.81 annex C / F.2.2.3
    bool defined = false;
    uint8_t values[256];
    int32_t mincode[17], maxcode[18], valptr[17];
    bool build(const uint8_t counts[16], const uint8_t* vals, int total) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; ++len) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memcpy(values, vals, (size_t)total);
        defined = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0, dcPred = 0;
    int x = 0, y = 0, w2 = 0, h2 = 0;            // size in samples, size padded to whole MCUs
    std::vector<uint8_t> data;                   // w2 x h2 samples
    std::vector<int16_t> coeff;                  // progressive: (w2 / 8) x (h2 / 8) blocks of 64
    int coeffW = 0;
};

struct Decoder {
    const uint8_t* p; const uint8_t* end;
    uint32_t codeBuffer = 0; int codeBits = 0; uint8_t marker = 0xff; bool nomore = false;
    int imgX = 0, imgY = 0, imgN = 0, hMax = 1, vMax = 1, mcuX = 0, mcuY = 0, mcuW = 0, mcuH = 0;
    bool progressive = false, jfif = false; int app14Transform = -1, rgb = 0;
    int specStart = 0, specEnd = 0, succHigh = 0, succLow = 0, eobRun = 0;
    int scanN = 0, order[4] = {0, 0, 0, 0};
    int restartInterval = 0, todo = 0;
    uint16_t dequant[4][64];
    Huffman hdc[4], hac[4];
    Component comp[4];
    bool bad = false;

    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    void grow() {
        do {
            unsigned b = nomore ? 0u : (unsigned)get8();
            if (b == 0xff) {
                int c = get8();
                while (c == 0xff) c = get8();            // fill bytes
                if (c != 0) { marker = (uint8_t)c; nomore = true; return; }
            }
            codeBuffer |= b << (24 - codeBits);
            codeBits += 8;
        } while (codeBits <= 24);
    }
    int bit() {
        if (codeBits < 1) grow();
        const int k = (int)(codeBuffer >> 31);
        codeBuffer <<= 1; --codeBits;
        return k;
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (codeBits < n) grow();
        const int k = (int)(codeBuffer >> (32 - n));
        codeBuffer <<= n; codeBits -= n;
        return k;
    }
    int extend_receive(int n) {                          // T.81 F.2.2.1 RECEIVE + EXTEND
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode(const Huffman& h) {
        if (!h.defined) { bad = true; return 0; }
        int code = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | bit();
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.values[h.valptr[len] + code - h.mincode[len]];
        }
        bad = true; return 0;
    }
    void reset_entropy() {
        codeBits = 0; codeBuffer = 0; nomore = false; marker = 0xff;
        for (int k = 0; k < 4; ++k) comp[k].dcPred = 0;
        todo = restartInterval ? restartInterval : 0x7fffffff;
        eobRun = 0;
    }

    // stbi__jpeg_decode_block: one baseline block, dequantised as it is decoded
    bool block_baseline(int16_t data[64], const Huffman& hd, const Huffman& ha, int c) {
        if (codeBits < 16) grow();
        const int t = decode(hd);
        if (bad || t > 15) return false;
        memset(data, 0, 64 * sizeof(int16_t));
        const int diff = t ? extend_receive(t) : 0;
        const int dc = comp[c].dcPred + diff;
        comp[c].dcPred = dc;
        const uint16_t* dq = dequant[comp[c].tq];
        data[0] = (int16_t)(dc * dq[0]);
        int k = 1;
        do {
            const int rs = decode(ha);
            if (bad) return false;
            const int s = rs & 15, r = rs >> 4;
            if (s == 0) {
                if (rs != 0xf0) break;                   // end of block
                k += 16;
            } else {
                k += r;
                const int zig = kDezigzag[k++];
                data[zig] = (int16_t)(extend_receive(s) * dq[zig]);
            }
        } while (k < 64);
        return true;
    }
    // stbi__jpeg_decode_block_prog_dc
    bool block_prog_dc(int16_t data[64], const Huffman& hd, int c) {
        if (specEnd != 0) return false;
        if (codeBits < 16) grow();
        if (succHigh == 0) {
            memset(data, 0, 64 * sizeof(int16_t));
            const int t = decode(hd);
            if (bad || t > 15) return false;
            const int diff = t ? extend_receive(t) : 0;
            const int dc = comp[c].dcPred + diff;
            comp[c].dcPred = dc;
            data[0] = (int16_t)(dc * (1 << succLow));
        } else if (bit()) {
            data[0] = (int16_t)(data[0] + (int16_t)(1 << succLow));
        }
        return true;
    }
    // stbi__jpeg_decode_block_prog_ac
    bool block_prog_ac(int16_t data[64], const Huffman& ha) {
        if (specStart == 0) return false;
        if (succHigh == 0) {
            const int shift = succLow;
            if (eobRun) { --eobRun; return true; }
            int k = specStart;
            do {
                if (codeBits < 16) grow();
                const int rs = decode(ha);
                if (bad) return false;
                const int s = rs & 15, r = rs >> 4;
                if (s == 0) {
                    if (r < 15) {
                        eobRun = 1 << r;
                        if (r) eobRun += bits(r);
                        --eobRun;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    const int zig = kDezigzag[k++];
                    data[zig] = (int16_t)(extend_receive(s) * (1 << shift));
                }
            } while (k <= specEnd);
        } else {
            const int16_t bitv = (int16_t)(1 << succLow);
            if (eobRun) {
                --eobRun;
                for (int k = specStart; k <= specEnd; ++k) {
                    int16_t* q = &data[kDezigzag[k]];
                    if (*q != 0 && bit() && (*q & bitv) == 0) *q = (int16_t)(*q > 0 ? *q + bitv : *q - bitv);
                }
            } else {
                int k = specStart;
                do {
                    if (codeBits < 16) grow();
                    const int rs = decode(ha);
                    if (bad) return false;
                    int s = rs & 15, r = rs >> 4;
                    if (s == 0) {
                        if (r < 15) {
                            eobRun = (1 << r) - 1;
                            if (r) eobRun += bits(r);
                            r = 64;                      // the rest of the band only refines
                        }
                    } else {
                        if (s != 1) return false;
                        s = bit() ? bitv : -bitv;
                    }
                    while (k <= specEnd) {
                        int16_t* q = &data[kDezigzag[k++]];
                        if (*q != 0) {
                            if (bit() && (*q & bitv) == 0) *q = (int16_t)(*q > 0 ? *q + bitv : *q - bitv);
                        } else {
                            if (r == 0) { *q = (int16_t)s; break; }
                            --r;
                        }
                    }
                } while (k <= specEnd);
            }
        }
        return true;
    }
};

inline uint8_t clamp8(int x) { return (unsigned)x > 255u ? (x < 0 ? 0 : 255) : (uint8_t)x; }
inline int f2f(double x) { return (int)(x * 4096 + 0.5); }

// stbi__idct_block
#define INCR_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                                    \
    int t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;                                                             \
    p2 = s2; p3 = s6;                                                                                                   \
    p1 = (p2 + p3) * f2f(0.5411961f);                                                                                   \
    t2 = p1 + p3 * f2f(-1.847759065f);                                                                                  \
    t3 = p1 + p2 * f2f(0.765366865f);                                                                                   \
    p2 = s0; p3 = s4;                                                                                                   \
    t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                                       \
    x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                                             \
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                                 \
    p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                                             \
    p5 = (p3 + p4) * f2f(1.175875602f);                                                                                 \
    t0 = t0 * f2f(0.298631336f); t1 = t1 * f2f(2.053119869f); t2 = t2 * f2f(3.072711026f); t3 = t3 * f2f(1.501321110f); \
    p1 = p5 + p1 * f2f(-0.899976223f); p2 = p5 + p2 * f2f(-2.562915447f);                                               \
    p3 = p3 * f2f(-1.961570560f); p4 = p4 * f2f(-0.390180644f);                                                         \
    t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

void idct_block(uint8_t* out, int outStride, const int16_t data[64]) {
    int val[64], *v = val;
    const int16_t* d = data;
    for (int i = 0; i < 8; ++i, ++d, ++v) {
        if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
            const int dcterm = d[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
        } else {
            INCR_IDCT_1D(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56])
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    v = val;
    uint8_t* o = out;
    for (int i = 0; i < 8; ++i, v += 8, o += outStride) {
        INCR_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
        x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

// stbi__resample_row_*
typedef const uint8_t* (*Resample)(uint8_t* out, const uint8_t* inNear, const uint8_t* inFar, int w, int hs);
const uint8_t* resample_1(uint8_t*, const uint8_t* inNear, const uint8_t*, int, int) { return inNear; }
const uint8_t* resample_v2(uint8_t* out, const uint8_t* inNear, const uint8_t* inFar, int w, int) {
    for (int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * inNear[i] + inFar[i] + 2) >> 2);
    return out;
}
const uint8_t* resample_h2(uint8_t* out, const uint8_t* in, const uint8_t*, int w, int) {
    if (w == 1) { out[0] = out[1] = in[0]; return out; }
    out[0] = in[0];
    out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
        const int n = 3 * in[i] + 2;
        out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
        out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
    }
    out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
    out[i * 2 + 1] = in[w - 1];
    return out;
}
const uint8_t* resample_hv2(uint8_t* out, const uint8_t* inNear, const uint8_t* inFar, int w, int) {
    if (w == 1) { out[0] = out[1] = (uint8_t)((3 * inNear[0] + inFar[0] + 2) >> 2); return out; }
    int t1 = 3 * inNear[0] + inFar[0];
    out[0] = (uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
        const int t0 = t1;
        t1 = 3 * inNear[i] + inFar[i];
        out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
        out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
    return out;
}
const uint8_t* resample_generic(uint8_t* out, const uint8_t* inNear, const uint8_t*, int w, int hs) {
    for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i * hs + j] = inNear[i];
    return out;
}

// stbi__YCbCr_to_RGB_row, four bytes per pixel
inline int float2fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }
void ycbcr_to_rgba_row(uint8_t* out, const uint8_t* y, const uint8_t* pcb, const uint8_t* pcr, int count) {
    for (int i = 0; i < count; ++i) {
        const int yFixed = (y[i] << 20) + (1 << 19);
        const int cr = pcr[i] - 128, cb = pcb[i] - 128;
        int r = yFixed + cr * float2fixed(1.40200f);
        int g = yFixed + (cr * -float2fixed(0.71414f)) + (int)((unsigned)(cb * -float2fixed(0.34414f)) & 0xffff0000u);
        int b = yFixed + cb * float2fixed(1.77200f);
        r >>= 20; g >>= 20; b >>= 20;
        out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b); out[3] = 255;
        out += 4;
    }
}

int fail() { return INCR_ERR_INVALID_ARGUMENT; }

} // namespace

extern "C" int incr_decode_jpeg(const uint8_t* bytes, size_t size, int32_t* width, int32_t* height, uint8_t* rgba, size_t capacity) {
    if (!bytes || size < 4 || !width || !height) return fail();
    Decoder D; D.p = bytes; D.end = bytes + size;
    if (D.get8() != 0xff || D.get8() != 0xd8) return fail();
    memset(D.dequant, 0, sizeof(D.dequant));
    bool haveFrame = false, finished = false;
    auto next_marker = [&]() -> int {
        if (D.marker != 0xff) { const int m = D.marker; D.marker = 0xff; return m; }
        int x = D.get8();
        if (x != 0xff) return 0xff;
        while (x == 0xff) x = D.get8();
        return x;
    };
    auto process_scan = [&]() -> bool {
        D.reset_entropy();
        int16_t block[64];
        if (!D.progressive) {
            if (D.scanN == 1) {
                const int n = D.order[0]; Component& c = D.comp[n];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
                    if (!D.block_baseline(block, D.hdc[c.hd], D.hac[c.ha], n)) return false;
                    idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, block);
                    if (--D.todo <= 0) {
                        if (D.codeBits < 24) D.grow();
                        if (!(D.marker >= 0xd0 && D.marker <= 0xd7)) return true;
                        D.reset_entropy();
                    }
                }
            } else {
                for (int j = 0; j < D.mcuY; ++j) for (int i = 0; i < D.mcuX; ++i) {
                    for (int k = 0; k < D.scanN; ++k) {
                        const int n = D.order[k]; Component& c = D.comp[n];
                        for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x) {
                            const int x2 = (i * c.h + x) * 8, y2 = (j * c.v + y) * 8;
                            if (!D.block_baseline(block, D.hdc[c.hd], D.hac[c.ha], n)) return false;
                            idct_block(c.data.data() + (size_t)c.w2 * y2 + x2, c.w2, block);
                        }
                    }
                    if (--D.todo <= 0) {
                        if (D.codeBits < 24) D.grow();
                        if (!(D.marker >= 0xd0 && D.marker <= 0xd7)) return true;
                        D.reset_entropy();
                    }
                }
            }
        } else {
            if (D.scanN == 1) {
                const int n = D.order[0]; Component& c = D.comp[n];
                const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
                for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
                    int16_t* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeffW);
                    if (D.specStart == 0) { if (!D.block_prog_dc(data, D.hdc[c.hd], n)) return false; }
                    else if (!D.block_prog_ac(data, D.hac[c.ha])) return false;
                    if (--D.todo <= 0) {
                        if (D.codeBits < 24) D.grow();
                        if (!(D.marker >= 0xd0 && D.marker <= 0xd7)) return true;
                        D.reset_entropy();
                    }
                }
            } else {
                for (int j = 0; j < D.mcuY; ++j) for (int i = 0; i < D.mcuX; ++i) {
                    for (int k = 0; k < D.scanN; ++k) {
                        const int n = D.order[k]; Component& c = D.comp[n];
                        for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x) {
                            const int x2 = i * c.h + x, y2 = j * c.v + y;
                            int16_t* data = c.coeff.data() + 64 * ((size_t)x2 + (size_t)y2 * c.coeffW);
                            if (!D.block_prog_dc(data, D.hdc[c.hd], n)) return false;
                        }
                    }
                    if (--D.todo <= 0) {
                        if (D.codeBits < 24) D.grow();
                        if (!(D.marker >= 0xd0 && D.marker <= 0xd7)) return true;
                        D.reset_entropy();
                    }
                }
            }
        }
        return !D.bad;
    };
    int m = next_marker();
    while (!finished) {
        if (m == 0xff) return fail();
        if (m == 0xd9) { finished = true; break; }
        if (m == 0xc0 || m == 0xc1 || m == 0xc2) {                      // SOF0 / SOF1 (Huffman, 8 bit) / SOF2
            if (haveFrame) return fail();
            D.progressive = m == 0xc2;
            const int L = D.get16();
            if (D.get8() != 8) return fail();
            D.imgY = D.get16(); D.imgX = D.get16(); D.imgN = D.get8();
            if (D.imgX <= 0 || D.imgY <= 0 || (D.imgN != 1 && D.imgN != 3) || L != 8 + 3 * D.imgN) return fail();          // (CMYK files are not textures: rejected)
            if ((int64_t)D.imgX * D.imgY > (int64_t)1 << 28) return fail();
            D.rgb = 0;
            for (int i = 0; i < D.imgN; ++i) {
                static const uint8_t rgbIds[3] = {'R', 'G', 'B'};
                Component& c = D.comp[i];
                c.id = D.get8();
                if (D.imgN == 3 && c.id == rgbIds[i]) ++D.rgb;
                const int q = D.get8();
                c.h = q >> 4; c.v = q & 15; c.tq = D.get8();
                if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail();
                if (c.h > D.hMax) D.hMax = c.h;
                if (c.v > D.vMax) D.vMax = c.v;
            }
            for (int i = 0; i < D.imgN; ++i) if (D.hMax % D.comp[i].h != 0 || D.vMax % D.comp[i].v != 0) return fail();
            D.mcuW = D.hMax * 8; D.mcuH = D.vMax * 8;
            D.mcuX = (D.imgX + D.mcuW - 1) / D.mcuW; D.mcuY = (D.imgY + D.mcuH - 1) / D.mcuH;
            for (int i = 0; i < D.imgN; ++i) {
                Component& c = D.comp[i];
                c.x = (D.imgX * c.h + D.hMax - 1) / D.hMax; c.y = (D.imgY * c.v + D.vMax - 1) / D.vMax;
                c.w2 = D.mcuX * c.h * 8; c.h2 = D.mcuY * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2, 0);
                if (D.progressive) { c.coeffW = c.w2 / 8; c.coeff.assign((size_t)c.w2 * c.h2, 0); }
            }
            haveFrame = true;
            if (!rgba) {                                              // size query
                *width = D.imgX; *height = D.imgY;
                return INCR_OK;
            }
        } else if (m == 0xc4) {                                        // DHT
            int L = D.get16() - 2;
            while (L > 0) {
                const int q = D.get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail();
                uint8_t counts[16]; int n = 0;
                for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)D.get8(); n += counts[i]; }
                if (n > 256) return fail();
                uint8_t vals[256];
                for (int i = 0; i < n; ++i) vals[i] = (uint8_t)D.get8();
                if (!(tc == 0 ? D.hdc[th] : D.hac[th]).build(counts, vals, n)) return fail();
                L -= 17 + n;
            }
            if (L != 0) return fail();
        } else if (m == 0xdb) {                                        // DQT
            int L = D.get16() - 2;
            while (L > 0) {
                const int q = D.get8(), pq = q >> 4, t = q & 15;
                if ((pq != 0 && pq != 1) || t > 3) return fail();
                for (int i = 0; i < 64; ++i) D.dequant[t][kDezigzag[i]] = (uint16_t)(pq ? D.get16() : D.get8());
                L -= pq ? 129 : 65;
            }
            if (L != 0) return fail();
        } else if (m == 0xdd) {                                        // DRI
            if (D.get16() != 4) return fail();
            D.restartInterval = D.get16();
        } else if (m == 0xda) {                                        // SOS
            if (!haveFrame) return fail();
            const int L = D.get16();
            D.scanN = D.get8();
            if (D.scanN < 1 || D.scanN > D.imgN || L != 6 + 2 * D.scanN) return fail();
            for (int i = 0; i < D.scanN; ++i) {
                const int id = D.get8(), q = D.get8();
                int which = 0;
                for (; which < D.imgN; ++which) if (D.comp[which].id == id) break;
                if (which == D.imgN) return fail();
                D.comp[which].hd = q >> 4; D.comp[which].ha = q & 15;
                if (D.comp[which].hd > 3 || D.comp[which].ha > 3) return fail();
                D.order[i] = which;
            }
            D.specStart = D.get8(); D.specEnd = D.get8();
            const int aa = D.get8(); D.succHigh = aa >> 4; D.succLow = aa & 15;
            if (D.progressive) {
                if (D.specStart > 63 || D.specEnd > 63 || D.specStart > D.specEnd || D.succHigh > 13 || D.succLow > 13) return fail();
            } else {
                if (D.specStart != 0 || D.succHigh != 0 || D.succLow != 0) return fail();
                D.specEnd = 63;
            }
            if (!process_scan()) return fail();
            if (D.marker == 0xff) {                                    // stb: skip whatever sits between the entropy-coded segment and the next marker
                while (D.p < D.end) {
                    const int x = D.get8();
                    if (x == 0xff) {
                        const int y = D.p < D.end ? D.get8() : 0xd9;
                        if (y != 0x00 && y != 0xff) { D.marker = (uint8_t)y; break; }
                    }
                }
                if (D.marker == 0xff) { finished = true; break; }
            }
        } else if (m == 0xe0) {                                        // APP0: JFIF
            const int L = D.get16();
            if (L >= 7) { char tag[5]; for (int i = 0; i < 5; ++i) tag[i] = (char)D.get8(); if (!memcmp(tag, "JFIF", 5)) D.jfif = true; for (int i = 0; i < L - 7; ++i) D.get8(); }
            else for (int i = 0; i < L - 2; ++i) D.get8();
        } else if (m == 0xee) {                                        // APP14: Adobe colour transform
            int L = D.get16() - 2;
            if (L >= 12) {
                char tag[6]; for (int i = 0; i < 6; ++i) tag[i] = (char)D.get8();
                L -= 6;
                if (!memcmp(tag, "Adobe", 6)) { D.get8(); D.get16(); D.get16(); D.app14Transform = D.get8(); L -= 6; }
            }
            for (int i = 0; i < L; ++i) D.get8();
        } else if ((m >= 0xe0 && m <= 0xef) || m == 0xfe || (m >= 0xd0 && m <= 0xd7) || m == 0x01) {
            if (!((m >= 0xd0 && m <= 0xd7) || m == 0x01)) { const int L = D.get16(); if (L < 2) return fail(); for (int i = 0; i < L - 2; ++i) D.get8(); }
        } else {
            return fail();                                              // arithmetic coding, lossless, hierarchical, DNL: not what a texture file holds
        }
        if (D.p >= D.end && D.marker == 0xff) { finished = true; break; }
        m = next_marker();
    }
    if (!haveFrame) return fail();
    *width = D.imgX; *height = D.imgY;
    if (!rgba) return INCR_OK;
    if (capacity < (size_t)D.imgX * D.imgY * 4) return fail();
    if (D.progressive) {                                              // stbi__jpeg_finish: dequantise, then the IDCT of every block
        for (int n = 0; n < D.imgN; ++n) {
            Component& c = D.comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            const uint16_t* dq = D.dequant[c.tq];
            for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
                int16_t* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.coeffW);
                for (int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * dq[k]);
                idct_block(c.data.data() + (size_t)c.w2 * j * 8 + i * 8, c.w2, data);
            }
        }
    }
    // load_jpeg_image: up-sample every component to the frame, row by row, then convert
    struct Row { Resample fn; const uint8_t* line0; const uint8_t* line1; int hs, vs, wLores, ystep, ypos; std::vector<uint8_t> buf; } R[3];
    for (int k = 0; k < D.imgN; ++k) {
        Component& c = D.comp[k];
        R[k].hs = D.hMax / c.h; R[k].vs = D.vMax / c.v;
        R[k].ystep = R[k].vs >> 1; R[k].wLores = (D.imgX + R[k].hs - 1) / R[k].hs; R[k].ypos = 0;
        R[k].line0 = R[k].line1 = c.data.data();
        R[k].buf.assign((size_t)D.imgX + 3, 0);
        R[k].fn = R[k].hs == 1 && R[k].vs == 1 ? resample_1 : (R[k].hs == 1 && R[k].vs == 2 ? resample_v2 : (R[k].hs == 2 && R[k].vs == 1 ? resample_h2 : (R[k].hs == 2 && R[k].vs == 2 ? resample_hv2 : resample_generic)));
    }
    const bool isRgb = D.imgN == 3 && (D.rgb == 3 || (D.app14Transform == 0 && !D.jfif));
    for (int j = 0; j < D.imgY; ++j) {
        const uint8_t* rows[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < D.imgN; ++k) {
            Row& r = R[k];
            const bool yBot = r.ystep >= (r.vs >> 1);
            rows[k] = r.fn(r.buf.data(), yBot ? r.line1 : r.line0, yBot ? r.line0 : r.line1, r.wLores, r.hs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < D.comp[k].y) r.line1 += D.comp[k].w2;
            }
        }
        uint8_t* out = rgba + (size_t)j * D.imgX * 4;
        if (D.imgN == 3) {
            if (isRgb) for (int i = 0; i < D.imgX; ++i) { out[4 * i] = rows[0][i]; out[4 * i + 1] = rows[1][i]; out[4 * i + 2] = rows[2][i]; out[4 * i + 3] = 255; }
            else ycbcr_to_rgba_row(out, rows[0], rows[1], rows[2], D.imgX);
        } else {
            for (int i = 0; i < D.imgX; ++i) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = rows[0][i]; out[4 * i + 3] = 255; }
        }
    }
    return INCR_OK;
}
