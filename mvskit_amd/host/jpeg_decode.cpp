// JPEG reader of the host mirror: what Image::readJpeg gets from CImg::load_jpeg -> libjpeg (image/image.cpp:827-879) --
// neither CImg nor libjpeg headers exist in this image, so the decoder is written out here.  8-bit Huffman JPEG, baseline /
// extended sequential / progressive (SOF0, SOF1, SOF2), grey or three components, restart intervals, interleaved and
// single-component scans.  The arithmetic follows the published libjpeg decompressor with its default settings
// (JDCT_ISLOW inverse DCT, "fancy" triangle up-sampling of 2:1 chroma, 16-bit fixed-point YCbCr -> RGB tables), which is
// what libjpeg-turbo -- the library behind CImg / PIL on current systems -- still implements bit for bit:
// tests/test_jpeg_decode.py compares the pixels with PIL's (libjpeg-turbo) on files of every supported kind.
// Not supported (an error, never a guess): arithmetic coding, lossless / hierarchical, 12-bit, four components,
// chroma sampling other than 1x1, 2x1, 2x2 relative to luma.
#include "jpeg_decode.hpp"

#include <cstdint>
#include <cstring>
#include <fstream>

namespace mvshost {
namespace {

const int kZigzag[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                              6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                              39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                              63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};  // a corrupt run past 63 lands here

struct Huff {
    bool defined = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int32_t maxcode[18];
    int32_t valptr[17];
    int32_t mincode[17];
    uint8_t look_n[512];  // 9-bit prefix -> code length (0 = longer than 9 bits)
    uint8_t look_v[512];
    bool build() {
        int code = 0, k = 0;
        memset(look_n, 0, sizeof look_n);
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                if (k >= 256 || code >= (1 << l)) return false;  // more codes of this length than the code space holds
                if (l <= 9) {
                    const int first = code << (9 - l);
                    for (int f = 0; f < (1 << (9 - l)); ++f) { look_n[first + f] = (uint8_t)l; look_v[first + f] = vals[k]; }
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            if (code > (1 << l)) return false;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int width = 0, height = 0;  // downsampled size in samples
    int bw = 0, bh = 0;         // blocks covering that size (single-component scans walk these)
    int pw = 0, ph = 0;         // blocks including MCU padding (allocation; interleaved scans)
    int dc_tbl = 0, ac_tbl = 0;
    int dc_pred = 0;
    std::vector<int16_t> coef;  // ph x pw blocks of 64, natural order
    std::vector<uint8_t> plane; // (ph*8) x (pw*8) samples after the inverse DCT
};

struct Decoder {
    const uint8_t* d;
    size_t n, pos = 0;
    std::string err;
    uint16_t quant[4][64];
    bool have_quant[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int W = 0, H = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false, have_frame = false, saw_jfif = false, saw_adobe = false;
    int adobe_transform = 0, restart_interval = 0;
    // bit reader
    uint32_t bitbuf = 0;
    int bitcnt = 0;
    bool hit_marker = false;
    int eobrun = 0;

    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    int u8() { return pos < n ? d[pos++] : -1; }
    int u16() { if (pos + 2 > n) { pos = n; return -1; } const int v = (d[pos] << 8) | d[pos + 1]; pos += 2; return v; }

    void bits_reset() { bitbuf = 0; bitcnt = 0; hit_marker = false; }
    void fill() {
        while (bitcnt <= 24) {
            int b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xff) {
                    const int b2 = pos + 1 < n ? d[pos + 1] : 0xd9;
                    if (b2 == 0) pos += 2;                  // stuffed zero
                    else { hit_marker = true; b = 0; }      // a marker ends the entropy-coded segment: feed zeros (as libjpeg does)
                } else ++pos;
            } else hit_marker = true;
            bitbuf |= (uint32_t)b << (24 - bitcnt);
            bitcnt += 8;
        }
    }
    int getbits(int k) {  // 0 <= k <= 16
        if (k == 0) return 0;
        if (bitcnt < k) fill();
        const int v = (int)(bitbuf >> (32 - k));
        bitbuf <<= k; bitcnt -= k;
        return v;
    }
    int getbit() { return getbits(1); }
    int decode(const Huff& h) {
        if (bitcnt < 16) fill();
        const int p = (int)(bitbuf >> 23);
        if (h.look_n[p]) { const int l = h.look_n[p]; bitbuf <<= l; bitcnt -= l; return h.look_v[p]; }
        int code = (int)(bitbuf >> 22), l = 10;
        while (l <= 16 && code > h.maxcode[l]) { ++l; code = (int)(bitbuf >> (32 - l)); }
        if (l > 16) { fail("corrupt JPEG: bad Huffman code"); return 0; }
        bitbuf <<= l; bitcnt -= l;
        return h.vals[(h.valptr[l] + code - h.mincode[l]) & 255];
    }
    static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

    // ---- marker segments
    bool read_dqt(int len) {
        const size_t end = pos + len;
        while (pos < end) {
            const int pq = u8();
            const int prec = pq >> 4, t = pq & 15;
            if (t > 3 || prec > 1) return fail("corrupt JPEG: bad quantisation table header");
            for (int i = 0; i < 64; ++i) {
                const int q = prec ? u16() : u8();
                if (q < 0) return fail("corrupt JPEG: short quantisation table");
                quant[t][kZigzag[i]] = (uint16_t)q;
            }
            have_quant[t] = true;
        }
        return pos == end ? true : fail("corrupt JPEG: quantisation table length");
    }
    bool read_dht(int len) {
        const size_t end = pos + len;
        while (pos < end) {
            const int tc = u8();
            const int cls = tc >> 4, t = tc & 15;
            if (cls > 1 || t > 3) return fail("corrupt JPEG: bad Huffman table header");
            Huff& h = cls ? ac[t] : dc[t];
            int total = 0;
            h.bits[0] = 0;
            for (int l = 1; l <= 16; ++l) { const int b = u8(); if (b < 0) return fail("corrupt JPEG: short Huffman table"); h.bits[l] = (uint8_t)b; total += b; }
            if (total > 256 || pos + total > end) return fail("corrupt JPEG: bad Huffman table");
            memset(h.vals, 0, sizeof h.vals);
            for (int i = 0; i < total; ++i) h.vals[i] = (uint8_t)u8();
            if (!h.build()) return fail("corrupt JPEG: bad Huffman code lengths");
            h.defined = true;
        }
        return pos == end ? true : fail("corrupt JPEG: Huffman table length");
    }
    bool read_sof(int len) {
        if (have_frame) return fail("unsupported JPEG: more than one frame");
        const int prec = u8();
        H = u16(); W = u16();
        const int nc = u8();
        if (prec != 8) return fail("unsupported JPEG: only 8-bit samples");
        if (W <= 0 || H <= 0) return fail("unsupported JPEG: image size missing from the frame header");
        if ((long long)W * H > (1ll << 28)) return fail("unsupported JPEG: more than 2^28 pixels");
        if (nc != 1 && nc != 3) return fail("unsupported JPEG: component count is not 1 or 3");
        if (len != 6 + 3 * nc) return fail("corrupt JPEG: frame header length");
        comps.assign(nc, Component());
        for (auto& c : comps) {
            c.id = u8();
            const int hv = u8();
            c.h = hv >> 4; c.v = hv & 15; c.tq = u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq < 0 || c.tq > 3) return fail("corrupt JPEG: bad component in the frame header");
            hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
        }
        if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }  // a single component is never interleaved: its factors do not matter
        mcux = (W + 8 * hmax - 1) / (8 * hmax);
        mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        for (auto& c : comps) {
            c.width = (W * c.h + hmax - 1) / hmax;
            c.height = (H * c.v + vmax - 1) / vmax;
            c.bw = (c.width + 7) / 8; c.bh = (c.height + 7) / 8;
            c.pw = mcux * c.h; c.ph = mcuy * c.v;
            c.coef.assign((size_t)c.pw * c.ph * 64, 0);
        }
        have_frame = true;
        return true;
    }

    // ---- entropy-coded blocks
    bool block_baseline(Component& c, int16_t* b) {
        const Huff &hd = dc[c.dc_tbl], &ha = ac[c.ac_tbl];
        const int t = decode(hd);
        if (t > 15) return fail("corrupt JPEG: bad DC size");
        const int diff = t ? extend(getbits(t), t) : 0;
        c.dc_pred += diff;
        b[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64; ++k) {
            const int rs = decode(ha), r = rs >> 4, s = rs & 15;
            if (s) { k += r; b[kZigzag[k]] = (int16_t)extend(getbits(s), s); }
            else { if (r != 15) break; k += 15; }
        }
        return err.empty();
    }
    bool block_dc_first(Component& c, int16_t* b, int al) {
        const int t = decode(dc[c.dc_tbl]);
        if (t > 15) return fail("corrupt JPEG: bad DC size");
        const int diff = t ? extend(getbits(t), t) : 0;
        c.dc_pred += diff;
        b[0] = (int16_t)(c.dc_pred * (1 << al));
        return err.empty();
    }
    void block_dc_refine(int16_t* b, int al) { if (getbit()) b[0] = (int16_t)(b[0] | (1 << al)); }
    bool block_ac_first(Component& c, int16_t* b, int ss, int se, int al) {
        if (eobrun > 0) { --eobrun; return true; }
        const Huff& ha = ac[c.ac_tbl];
        for (int k = ss; k <= se; ++k) {
            const int rs = decode(ha), r = rs >> 4, s = rs & 15;
            if (s) { k += r; b[kZigzag[k]] = (int16_t)(extend(getbits(s), s) * (1 << al)); }
            else {
                if (r == 15) { k += 15; continue; }
                eobrun = 1 << r;
                if (r) eobrun += getbits(r);
                --eobrun;
                break;
            }
        }
        return err.empty();
    }
    bool block_ac_refine(Component& c, int16_t* b, int ss, int se, int al) {
        const Huff& ha = ac[c.ac_tbl];
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        auto refine = [&](int16_t& v) {
            if (getbit() && (v & p1) == 0) v = (int16_t)(v >= 0 ? v + p1 : v + m1);
        };
        if (eobrun == 0) {
            for (; k <= se; ++k) {
                const int rs = decode(ha);
                int r = rs >> 4, s = rs & 15;
                if (s) {
                    if (s != 1) return fail("corrupt JPEG: bad refinement symbol");
                    s = getbit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += getbits(r);
                    break;
                }
                do {
                    int16_t& v = b[kZigzag[k]];
                    if (v != 0) refine(v);
                    else if (--r < 0) break;
                    ++k;
                } while (k <= se);
                if (s && k <= se) b[kZigzag[k]] = (int16_t)s;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; ++k) { int16_t& v = b[kZigzag[k]]; if (v != 0) refine(v); }
            --eobrun;
        }
        return err.empty();
    }

    bool restart(int& next_rst) {
        // the entropy-coded segment ends on a byte boundary, followed by RSTn
        bits_reset();
        while (pos < n && d[pos] != 0xff) ++pos;  // libjpeg skips garbage up to the marker
        while (pos + 1 < n && d[pos] == 0xff && d[pos + 1] == 0xff) ++pos;
        if (pos + 1 >= n || d[pos] != 0xff || d[pos + 1] != (0xd0 + next_rst)) return fail("corrupt JPEG: restart marker missing");
        pos += 2;
        next_rst = (next_rst + 1) & 7;
        for (auto& c : comps) c.dc_pred = 0;
        eobrun = 0;
        return true;
    }

    bool read_scan(int len) {
        if (!have_frame) return fail("corrupt JPEG: scan before the frame header");
        const int ns = u8();
        if (ns < 1 || ns > (int)comps.size() || len != 4 + 2 * ns) return fail("corrupt JPEG: bad scan header");
        Component* sc[3];
        for (int i = 0; i < ns; ++i) {
            const int id = u8(), t = u8();
            sc[i] = nullptr;
            for (auto& c : comps) if (c.id == id) sc[i] = &c;
            if (!sc[i]) return fail("corrupt JPEG: scan names an unknown component");
            sc[i]->dc_tbl = t >> 4; sc[i]->ac_tbl = t & 15;
            if (sc[i]->dc_tbl > 3 || sc[i]->ac_tbl > 3) return fail("corrupt JPEG: bad table selector");
        }
        const int ss = u8(), se = u8(), ahal = u8();
        const int ah = ahal >> 4, al = ahal & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) return fail("corrupt JPEG: bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ahal != 0) return fail("corrupt JPEG: bad sequential scan parameters");
        for (int i = 0; i < ns; ++i) {
            const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
            if (need_dc && !dc[sc[i]->dc_tbl].defined) return fail("corrupt JPEG: DC Huffman table missing");
            if (need_ac && !ac[sc[i]->ac_tbl].defined) return fail("corrupt JPEG: AC Huffman table missing");
        }
        bits_reset();
        for (auto& c : comps) c.dc_pred = 0;
        eobrun = 0;
        int next_rst = 0, todo = restart_interval;
        auto one = [&](Component& c, int16_t* b) -> bool {
            if (!progressive) return block_baseline(c, b);
            if (ss == 0) { if (ah == 0) return block_dc_first(c, b, al); block_dc_refine(b, al); return true; }
            return ah == 0 ? block_ac_first(c, b, ss, se, al) : block_ac_refine(c, b, ss, se, al);
        };
        if (ns == 1) {
            Component& c = *sc[0];
            for (int by = 0; by < c.bh; ++by) for (int bx = 0; bx < c.bw; ++bx) {
                if (restart_interval && todo == 0) { if (!restart(next_rst)) return false; todo = restart_interval; }
                if (!one(c, &c.coef[((size_t)by * c.pw + bx) * 64])) return false;
                --todo;
            }
        } else {
            for (int my = 0; my < mcuy; ++my) for (int mx = 0; mx < mcux; ++mx) {
                if (restart_interval && todo == 0) { if (!restart(next_rst)) return false; todo = restart_interval; }
                for (int i = 0; i < ns; ++i) {
                    Component& c = *sc[i];
                    for (int y = 0; y < c.v; ++y) for (int x = 0; x < c.h; ++x)
                        if (!one(c, &c.coef[((size_t)(my * c.v + y) * c.pw + (mx * c.h + x)) * 64])) return false;
                }
                --todo;
            }
        }
        // the bit reader may have read ahead of the segment's end; the next marker is searched from the last byte it took
        while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] != 0 && !(d[pos + 1] >= 0xd0 && d[pos + 1] <= 0xd7))) ++pos;
        return true;
    }

    // ---- reconstruction
    static inline uint8_t range_limit(int64_t v) {  // libjpeg's post-IDCT table: index (v & 1023), centred on 128
        const int i = (int)(v & 1023);
        return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
    }
    static void idct_islow(const int16_t* in, const uint16_t* q, uint8_t* out, int stride) {  // jidctint.c, jpeg_idct_islow
        enum { CB = 13, P1 = 2 };
        const int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                      F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
        auto descale = [](int64_t x, int nb) { return (x + ((int64_t)1 << (nb - 1))) >> nb; };  // JLONG is 64 bits wide on LP64: a corrupt stream cannot overflow it
        int64_t ws[64];
        for (int c = 0; c < 8; ++c) {
            const int16_t* ip = in + c;
            const uint16_t* qp = q + c;
            int64_t* wp = ws + c;
            if (ip[8] == 0 && ip[16] == 0 && ip[24] == 0 && ip[32] == 0 && ip[40] == 0 && ip[48] == 0 && ip[56] == 0) {
                const int64_t dcv = (int64_t)ip[0] * qp[0] * (1 << P1);
                for (int r = 0; r < 8; ++r) wp[8 * r] = dcv;
                continue;
            }
            int64_t z2 = (int64_t)ip[16] * qp[16], z3 = (int64_t)ip[48] * qp[48];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            z2 = (int64_t)ip[0] * qp[0]; z3 = (int64_t)ip[32] * qp[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = (int64_t)ip[56] * qp[56]; tmp1 = (int64_t)ip[40] * qp[40]; tmp2 = (int64_t)ip[24] * qp[24]; tmp3 = (int64_t)ip[8] * qp[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
            wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
            wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
            wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; ++r) {
            const int64_t* wp = ws + 8 * r;
            uint8_t* op = out + (size_t)r * stride;
            int64_t z2 = wp[2], z3 = wp[6];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            int64_t tmp0 = (wp[0] + wp[4]) * (1 << CB), tmp1 = (wp[0] - wp[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const int S = CB + P1 + 3;
            op[0] = range_limit(descale(tmp10 + tmp3, S)); op[7] = range_limit(descale(tmp10 - tmp3, S));
            op[1] = range_limit(descale(tmp11 + tmp2, S)); op[6] = range_limit(descale(tmp11 - tmp2, S));
            op[2] = range_limit(descale(tmp12 + tmp1, S)); op[5] = range_limit(descale(tmp12 - tmp1, S));
            op[3] = range_limit(descale(tmp13 + tmp0, S)); op[4] = range_limit(descale(tmp13 - tmp0, S));
        }
    }
    bool inverse_dct() {
        for (auto& c : comps) {
            if (!have_quant[c.tq]) return fail("corrupt JPEG: quantisation table missing");
            const int stride = c.pw * 8;
            c.plane.assign((size_t)stride * c.ph * 8, 0);
            for (int by = 0; by < c.ph; ++by) for (int bx = 0; bx < c.pw; ++bx)
                idct_islow(&c.coef[((size_t)by * c.pw + bx) * 64], quant[c.tq], &c.plane[(size_t)by * 8 * stride + bx * 8], stride);
            std::vector<int16_t>().swap(c.coef);
        }
        return true;
    }
    // jdsample.c: full-size planes of W x H from a component (rows of the true downsampled size, edges replicated)
    bool upsample(const Component& c, std::vector<uint8_t>& out) const {
        out.resize((size_t)W * H);
        const int stride = c.pw * 8, cw = c.width, ch = c.height;
        const uint8_t* src = c.plane.data();
        if (c.h == hmax && c.v == vmax) {  // fullsize_upsample
            for (int y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], src + (size_t)y * stride, W);
            return true;
        }
        const bool fancy = cw > 2;  // jinit_upsampler: the triangle filters need more than two columns
        std::vector<uint8_t> row((size_t)2 * cw + 2);
        if (2 * c.h == hmax && c.v == vmax) {  // h2v1_fancy_upsample / h2v1_upsample
            for (int y = 0; y < H; ++y) {
                const uint8_t* in = src + (size_t)y * stride;
                if (!fancy) { for (int x = 0; x < cw; ++x) row[2 * x] = row[2 * x + 1] = in[x]; }
                else {
                    row[0] = in[0];
                    row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                    for (int x = 1; x < cw - 1; ++x) {
                        const int v = in[x] * 3;
                        row[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
                        row[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
                    }
                    row[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2);
                    row[2 * cw - 1] = in[cw - 1];
                }
                memcpy(&out[(size_t)y * W], row.data(), W);
            }
            return true;
        }
        if (2 * c.h == hmax && 2 * c.v == vmax) {  // h2v2_fancy_upsample / h2v2_upsample
            for (int y = 0; y < H; ++y) {
                const int r = y >> 1;
                const uint8_t* in0 = src + (size_t)r * stride;
                if (!fancy) { for (int x = 0; x < cw; ++x) row[2 * x] = row[2 * x + 1] = in0[x]; }
                else {
                    int rn = (y & 1) ? r + 1 : r - 1;  // the nearer neighbour row; the image's first / last row stands in for a missing one
                    rn = rn < 0 ? 0 : rn > ch - 1 ? ch - 1 : rn;
                    const uint8_t* in1 = src + (size_t)rn * stride;
                    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
                    row[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
                    row[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                    lastcol = thiscol; thiscol = nextcol;
                    for (int x = 1; x < cw - 1; ++x) {
                        nextcol = in0[x + 1] * 3 + in1[x + 1];
                        row[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                        row[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
                        lastcol = thiscol; thiscol = nextcol;
                    }
                    row[2 * cw - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
                    row[2 * cw - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
                }
                memcpy(&out[(size_t)y * W], row.data(), W);
            }
            return true;
        }
        return false;
    }

    bool run(std::vector<unsigned char>& pixels, int& width, int& height, int& channels) {
        if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return fail("not a JPEG file (no SOI marker)");
        pos = 2;
        bool done = false;
        while (!done) {
            // next marker
            while (pos < n && d[pos] != 0xff) ++pos;
            while (pos < n && d[pos] == 0xff) ++pos;
            if (pos >= n) break;
            const int m = d[pos++];
            if (m == 0xd9) { done = true; break; }
            if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
            const int len = u16();
            if (len < 2 || pos + (size_t)(len - 2) > n) return fail("corrupt JPEG: marker segment runs past the end of the file");
            const size_t next = pos + (size_t)(len - 2);
            bool ok = true, is_scan = false;
            switch (m) {
                case 0xc0: case 0xc1: ok = read_sof(len - 2); break;
                case 0xc2: progressive = true; ok = read_sof(len - 2); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    return fail("unsupported JPEG: lossless, hierarchical or arithmetic-coded frame");
                case 0xc4: ok = read_dht(len - 2); break;
                case 0xdb: ok = read_dqt(len - 2); break;
                case 0xdd: restart_interval = u16(); break;
                case 0xe0: if (len >= 7 && !memcmp(d + pos, "JFIF", 5)) saw_jfif = true; break;
                case 0xee: if (len >= 14 && !memcmp(d + pos, "Adobe", 5)) { saw_adobe = true; adobe_transform = d[pos + 11]; } break;
                case 0xda: ok = read_scan(len - 2); is_scan = true; break;
                default: break;
            }
            if (!ok) return false;
            if (!is_scan) pos = next;
        }
        if (!have_frame) return fail("corrupt JPEG: no frame header");
        if (!err.empty()) return false;
        if (!inverse_dct()) return false;
        width = W; height = H; channels = (int)comps.size();
        if (channels == 1) {
            pixels.resize((size_t)W * H);
            const int stride = comps[0].pw * 8;
            for (int y = 0; y < H; ++y) memcpy(&pixels[(size_t)y * W], &comps[0].plane[(size_t)y * stride], W);
            return true;
        }
        std::vector<uint8_t> p[3];
        for (int i = 0; i < 3; ++i)
            if (!upsample(comps[i], p[i])) return fail("unsupported JPEG: chroma sampling other than 1x1, 2x1 or 2x2");
        // jdapimin.c default_decompress_parms: which colour space three components are in
        bool ycc = true;
        if (saw_jfif) ycc = true;
        else if (saw_adobe) ycc = adobe_transform != 0;
        else if (comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B') ycc = false;
        pixels.resize((size_t)W * H * 3);
        if (!ycc) {
            for (size_t i = 0; i < (size_t)W * H; ++i) { pixels[3 * i] = p[0][i]; pixels[3 * i + 1] = p[1][i]; pixels[3 * i + 2] = p[2][i]; }
            return true;
        }
        // jdcolor.c build_ycc_rgb_table: 16-bit fixed point, FIX(x) = x * 65536 + 0.5
        int cr_r[256], cb_b[256];
        int32_t cr_g[256], cb_g[256];
        for (int i = 0; i < 256; ++i) {
            const int32_t x = i - 128;
            cr_r[i] = (int)((91881 * x + 32768) >> 16);
            cb_b[i] = (int)((116130 * x + 32768) >> 16);
            cr_g[i] = -46802 * x;
            cb_g[i] = -22554 * x + 32768;
        }
        auto clamp = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            const int y = p[0][i], cb = p[1][i], cr = p[2][i];
            pixels[3 * i] = clamp(y + cr_r[cr]);
            pixels[3 * i + 1] = clamp(y + (int)((cb_g[cb] + cr_g[cr]) >> 16));
            pixels[3 * i + 2] = clamp(y + cb_b[cb]);
        }
        return true;
    }
};

}  // namespace

int decodeJpeg(const unsigned char* data, size_t size, std::vector<unsigned char>& pixels, int& width, int& height, int& channels,
               std::string* error) {
    Decoder dec{};
    dec.d = data; dec.n = size;
    if (!dec.run(pixels, width, height, channels)) {
        if (error) *error = dec.err.empty() ? "corrupt JPEG" : dec.err;
        pixels.clear();
        return -1;
    }
    return 0;
}

int readJpegFile(const std::string& file, std::vector<unsigned char>& pixels, int& width, int& height, int& channels, std::string* error) {
    std::ifstream is(file.c_str(), std::ios::binary);
    if (!is.is_open()) { if (error) *error = "cannot open " + file; return -1; }
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    return decodeJpeg(bytes.data(), bytes.size(), pixels, width, height, channels, error);
}

}  // namespace mvshost

extern "C" int mvshost_jpeg_decode(const unsigned char* data, long long size, unsigned char* out, long long out_capacity, int* width,
                                   int* height, int* channels, char* err, int err_capacity) {
    std::vector<unsigned char> px;
    std::string e;
    int w = 0, h = 0, c = 0;
    if (mvshost::decodeJpeg(data, (size_t)size, px, w, h, c, &e) != 0) {
        if (err && err_capacity > 0) { strncpy(err, e.c_str(), (size_t)err_capacity - 1); err[err_capacity - 1] = 0; }
        return -1;
    }
    if (width) *width = w;
    if (height) *height = h;
    if (channels) *channels = c;
    if (out) {
        if ((long long)px.size() > out_capacity) return -2;
        memcpy(out, px.data(), px.size());
    }
    return 0;
}
