// JPEG (baseline + progressive Huffman, 8 bit) and TGA decoders for texture / sky ingestion.
//
// The reference reads images through FreeImage 3.18 with flags = 0 (reference render/image.cc:159-160).  For JPEG
// that means FreeImage's "fast" settings: libjpeg's JDCT_IFAST inverse DCT and no fancy up-sampling
// (PluginJPEG.cpp: without JPEG_ACCURATE, dct_method = JDCT_IFAST and do_fancy_upsampling = FALSE).  The decoder
// below restates those published libjpeg algorithms -- jidctfst.c (AA&N, 8-bit constants, truncating shifts),
// jddctmgr.c's IFAST multiplier table, jdsample.c's replication up-sampling, jdcolor.c's fixed-point YCbCr -> RGB --
// so a texel gets the same 8-bit value.  FreeImage itself is absent here ("parity unpinned" for the codec); the tests
// pin the decoder against libjpeg-turbo driven the same way (Pillow's draft mode: JDCT_FASTEST, no fancy up-sampling).
#include "rl_host.h"

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace rl {
namespace {

const uint8_t kZigzag[64] = {
	0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
	35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// jddctmgr.c: AA&N scale factors, 14 fractional bits, natural order
const int16_t kAanScales[64] = {
	16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 22725, 31521, 29692, 26722, 22725, 17855, 12299, 6270,
	21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906, 19266, 26722, 25172, 22654, 19266, 15137, 10426, 5315,
	16384, 22725, 21407, 19266, 16384, 12873, 8867, 4520, 12873, 17855, 16819, 15137, 12873, 10114, 6967, 3552,
	8867, 12299, 11585, 10426, 8867, 6967, 4799, 2446, 4520, 6270, 5906, 5315, 4520, 3552, 2446, 1247 };

struct Huff {
	bool present = false;
	// canonical decoding tables (ITU T.81 F.2.2.3)
	int32_t mincode[17], maxcode[18], valptr[17];
	uint8_t vals[256];
	void build(const uint8_t* counts, const uint8_t* symbols, int n)
	{
		present = true;
		memset(vals, 0, sizeof(vals)); memcpy(vals, symbols, (size_t)n);
		int code = 0, k = 0;
		for (int l = 1; l <= 16; ++l) {
			valptr[l] = k; mincode[l] = code;
			code += counts[l - 1]; k += counts[l - 1];
			maxcode[l] = counts[l - 1] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7fffffff;
	}
};

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int blocksW = 0, blocksH = 0; std::vector<int16_t> coef; int pred = 0; };

struct BitReader {
	const uint8_t* p; const uint8_t* end;
	uint32_t acc = 0; int bits = 0;
	bool hitMarker = false;
	void reset() { acc = 0; bits = 0; hitMarker = false; }
	void fill()
	{
		while (bits <= 24) {
			uint32_t b = 0;
			if (!hitMarker && p < end) {
				b = *p;
				if (b == 0xFF) {
					if (p + 1 < end && p[1] == 0x00) { p += 2; }
					else { hitMarker = true; b = 0; }          // a marker: feed zeros, leave p on it
				} else ++p;
			}
			acc |= b << (24 - bits);
			bits += 8;
		}
	}
	// n: 1..16 in a valid stream (a magnitude category or an EOB-run length); a corrupt Huffman table can name any byte -> clamped
	int get(int n) { if (n <= 0) return 0; if (n > 16) n = 16; fill(); int v = (int)(acc >> (32 - n)); acc <<= n; bits -= n; return v; }
	int bit() { return get(1); }
	int decode(const Huff& h)
	{
		fill();
		int code = 0;
		for (int l = 1; l <= 16; ++l) {
			code = (code << 1) | (int)(acc >> 31); acc <<= 1; --bits;
			if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
		}
		return 0;   // corrupt stream
	}
};
inline int Extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

// 32-bit integer with two's-complement wrap-around: what libjpeg's INT32 arithmetic does in practice.  On valid streams nothing
// wraps; on corrupt ones (coefficients far outside the 8-bit range) the result is garbage pixels, as with libjpeg, instead of
// undefined behaviour.
struct W32 {
	uint32_t u;
	W32() : u(0) {}
	W32(int32_t v) : u((uint32_t)v) {}
	int32_t s() const { return (int32_t)u; }
};
inline W32 operator+(W32 a, W32 b) { W32 r; r.u = a.u + b.u; return r; }
inline W32 operator-(W32 a, W32 b) { W32 r; r.u = a.u - b.u; return r; }
inline W32 operator*(W32 a, W32 b) { W32 r; r.u = a.u * b.u; return r; }
inline W32 Sar8(W32 a) { return W32(a.s() >> 8); }

// jidctfst.c with the IFAST multipliers of jddctmgr.c; out: 64 samples 0..255
void IdctIfast(const int16_t* coef, const int32_t* mult, uint8_t* out)
{
	const W32 F1_082 = 277, F1_414 = 362, F1_847 = 473, F2_613 = 669, NEG_F2_613 = -669;
	#define JMUL(v, c) Sar8((v) * (c))
	W32 ws[64];
	for (int c = 0; c < 8; ++c) {
		const int16_t* in = coef + c; const int32_t* q = mult + c; W32* w = ws + c;
		W32 tmp0 = W32(in[0]) * W32(q[0]), tmp1 = W32(in[16]) * W32(q[16]), tmp2 = W32(in[32]) * W32(q[32]), tmp3 = W32(in[48]) * W32(q[48]);
		W32 tmp10 = tmp0 + tmp2, tmp11 = tmp0 - tmp2;
		W32 tmp13 = tmp1 + tmp3, tmp12 = JMUL(tmp1 - tmp3, F1_414) - tmp13;
		tmp0 = tmp10 + tmp13; tmp3 = tmp10 - tmp13; tmp1 = tmp11 + tmp12; tmp2 = tmp11 - tmp12;
		W32 tmp4 = W32(in[8]) * W32(q[8]), tmp5 = W32(in[24]) * W32(q[24]), tmp6 = W32(in[40]) * W32(q[40]), tmp7 = W32(in[56]) * W32(q[56]);
		W32 z13 = tmp6 + tmp5, z10 = tmp6 - tmp5, z11 = tmp4 + tmp7, z12 = tmp4 - tmp7;
		tmp7 = z11 + z13; tmp11 = JMUL(z11 - z13, F1_414);
		W32 z5 = JMUL(z10 + z12, F1_847);
		tmp10 = JMUL(z12, F1_082) - z5;
		tmp12 = JMUL(z10, NEG_F2_613) + z5;
		tmp6 = tmp12 - tmp7; tmp5 = tmp11 - tmp6; tmp4 = tmp10 + tmp5;
		w[0] = tmp0 + tmp7; w[56] = tmp0 - tmp7; w[8] = tmp1 + tmp6; w[48] = tmp1 - tmp6;
		w[16] = tmp2 + tmp5; w[40] = tmp2 - tmp5; w[32] = tmp3 + tmp4; w[24] = tmp3 - tmp4;
	}
	auto limit = [](W32 xw) -> uint8_t {
		const int32_t x = xw.s();
		// range_limit[(x >> 5) & RANGE_MASK] of the IDCT table (jdmaster.c prepare_range_limit_table), CENTERJSAMPLE folded in
		const int idx = (x >> 5) & 1023;
		if (idx < 128) return (uint8_t)(128 + idx);
		if (idx < 512) return 255;
		if (idx < 896) return 0;
		return (uint8_t)(idx - 896);
	};
	for (int r = 0; r < 8; ++r) {
		const W32* w = ws + 8 * r; uint8_t* o = out + 8 * r;
		W32 tmp10 = w[0] + w[4], tmp11 = w[0] - w[4];
		W32 tmp13 = w[2] + w[6], tmp12 = JMUL(w[2] - w[6], F1_414) - tmp13;
		W32 tmp0 = tmp10 + tmp13, tmp3 = tmp10 - tmp13, tmp1 = tmp11 + tmp12, tmp2 = tmp11 - tmp12;
		W32 z13 = w[5] + w[3], z10 = w[5] - w[3], z11 = w[1] + w[7], z12 = w[1] - w[7];
		W32 tmp7 = z11 + z13; tmp11 = JMUL(z11 - z13, F1_414);
		W32 z5 = JMUL(z10 + z12, F1_847);
		tmp10 = JMUL(z12, F1_082) - z5;
		tmp12 = JMUL(z10, NEG_F2_613) + z5;
		W32 tmp6 = tmp12 - tmp7, tmp5 = tmp11 - tmp6, tmp4 = tmp10 + tmp5;
		o[0] = limit(tmp0 + tmp7); o[7] = limit(tmp0 - tmp7); o[1] = limit(tmp1 + tmp6); o[6] = limit(tmp1 - tmp6);
		o[2] = limit(tmp2 + tmp5); o[5] = limit(tmp2 - tmp5); o[4] = limit(tmp3 + tmp4); o[3] = limit(tmp3 - tmp4);
	}
	#undef JMUL
}

inline uint8_t Clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

struct Decoder {
	const uint8_t* d; size_t n;
	int width = 0, height = 0, ncomp = 0; bool progressive = false;
	uint16_t qt[4][64]; bool qtPresent[4] = { false, false, false, false };
	Huff dc[4], ac[4];
	Component comp[4];
	int hmax = 1, vmax = 1, mcuW = 0, mcuH = 0, restart = 0;
	int adobeTransform = -1;
	int eobrun = 0;

	bool parse(std::vector<uint8_t>& rgba)
	{
		if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
		size_t p = 2;
		bool sawFrame = false;
		while (p + 4 <= n) {
			if (d[p] != 0xFF) { ++p; continue; }
			const uint8_t m = d[p + 1];
			if (m == 0xFF) { ++p; continue; }
			if (m == 0xD9) break;
			if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
			const size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
			if (len < 2 || p + 2 + len > n) return false;
			const uint8_t* b = d + p + 4; const size_t bl = len - 2;
			if (m == 0xDB) {                       // DQT
				size_t k = 0;
				while (k < bl) {
					const int pq = b[k] >> 4, tq = b[k] & 15; ++k;
					if (tq > 3 || k + (pq ? 128 : 64) > bl) return false;
					for (int i = 0; i < 64; ++i) { qt[tq][kZigzag[i]] = pq ? (uint16_t)((b[k] << 8) | b[k + 1]) : b[k]; k += pq ? 2 : 1; }
					qtPresent[tq] = true;
				}
			} else if (m == 0xC4) {                // DHT
				size_t k = 0;
				while (k + 17 <= bl) {
					const int tc = b[k] >> 4, th = b[k] & 15;
					int total = 0; for (int i = 0; i < 16; ++i) total += b[k + 1 + i];
					if (th > 3 || tc > 1 || total > 256 || k + 17 + (size_t)total > bl) return false;
					(tc ? ac[th] : dc[th]).build(b + k + 1, b + k + 17, total);
					k += 17 + (size_t)total;
				}
			} else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {   // SOF0/1/2
				if (sawFrame || bl < 6 || b[0] != 8) return false;
				progressive = (m == 0xC2);
				height = (b[1] << 8) | b[2]; width = (b[3] << 8) | b[4]; ncomp = b[5];
				if (!width || !height || (ncomp != 1 && ncomp != 3) || bl < 6 + 3 * (size_t)ncomp) return false;
				if (!PlausibleImageSize((uint64_t)width, (uint64_t)height, (uint64_t)n, 4096)) return false;   // a scan cannot be that much smaller than its picture
				for (int c = 0; c < ncomp; ++c) {
					comp[c].id = b[6 + 3 * c]; comp[c].h = b[7 + 3 * c] >> 4; comp[c].v = b[7 + 3 * c] & 15; comp[c].tq = b[8 + 3 * c] & 3;
					if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4) return false;
					hmax = std::max(hmax, comp[c].h); vmax = std::max(vmax, comp[c].v);
				}
				if (ncomp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }   // a single component is never interleaved
				mcuW = (width + 8 * hmax - 1) / (8 * hmax); mcuH = (height + 8 * vmax - 1) / (8 * vmax);
				for (int c = 0; c < ncomp; ++c) {
					comp[c].blocksW = mcuW * comp[c].h; comp[c].blocksH = mcuH * comp[c].v;
					comp[c].coef.assign((size_t)comp[c].blocksW * comp[c].blocksH * 64, 0);
				}
				sawFrame = true;
			} else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
				return false;                      // lossless / arithmetic / hierarchical
			} else if (m == 0xDD) {                // DRI
				if (bl >= 2) restart = (b[0] << 8) | b[1];
			} else if (m == 0xEE) {                // APP14 "Adobe"
				if (bl >= 12 && !memcmp(b, "Adobe", 5)) adobeTransform = b[11];
			} else if (m == 0xDA) {                // SOS
				if (!sawFrame) return false;
				size_t consumed = 0;
				if (!scan(b, bl, d + p + 2 + len, d + n, consumed)) return false;
				p += 2 + len + consumed;
				continue;
			}
			p += 2 + len;
		}
		if (!sawFrame) return false;
		return finish(rgba);
	}

	bool scan(const uint8_t* hdr, size_t hl, const uint8_t* data, const uint8_t* end, size_t& consumed)
	{
		if (hl < 1) return false;
		const int ns = hdr[0];
		if (ns < 1 || ns > ncomp || hl < 1 + 2 * (size_t)ns + 3) return false;
		int idx[4];
		for (int i = 0; i < ns; ++i) {
			int c = -1;
			for (int k = 0; k < ncomp; ++k) if (comp[k].id == hdr[1 + 2 * i]) c = k;
			if (c < 0) return false;
			idx[i] = c; comp[c].td = hdr[2 + 2 * i] >> 4; comp[c].ta = hdr[2 + 2 * i] & 15;
			if (comp[c].td > 3 || comp[c].ta > 3) return false;
		}
		const int Ss = hdr[1 + 2 * ns], Se = hdr[2 + 2 * ns], Ah = hdr[3 + 2 * ns] >> 4, Al = hdr[3 + 2 * ns] & 15;
		if (progressive) { if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) return false; }
		else if (Ss != 0 || Se != 63) { /* tolerated: baseline ignores them */ }
		BitReader br; br.p = data; br.end = end;
		for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
		eobrun = 0;
		int untilRestart = restart;
		auto restartCheck = [&]() -> bool {
			if (!restart) return true;
			if (--untilRestart > 0) return true;
			// byte-align, expect RSTn
			br.reset();
			while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) {
				if (br.p[0] == 0xFF && br.p[1] != 0x00 && br.p[1] != 0xFF) return true;   // another marker: the scan is over
				++br.p;
			}
			if (br.p + 1 < br.end) br.p += 2;
			for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
			eobrun = 0;
			untilRestart = restart;
			return true;
		};
		if (ns > 1) {
			// interleaved: one MCU = h x v blocks of every scan component
			const int total = mcuW * mcuH;
			for (int m = 0; m < total; ++m) {
				const int mx = m % mcuW, my = m / mcuW;
				for (int i = 0; i < ns; ++i) {
					Component& C = comp[idx[i]];
					for (int by = 0; by < C.v; ++by) for (int bx = 0; bx < C.h; ++bx) {
						int16_t* blk = &C.coef[((size_t)(my * C.v + by) * C.blocksW + (mx * C.h + bx)) * 64];
						if (!block(br, C, blk, Ss, Se, Ah, Al)) return false;
					}
				}
				if (m + 1 < total) restartCheck();
			}
		} else {
			// non-interleaved: the component's own block raster, clipped to the image (T.81 A.2.2)
			Component& C = comp[idx[0]];
			const int bw = (int)(((size_t)width * C.h + hmax * 8 - 1) / ((size_t)hmax * 8));
			const int bh = (int)(((size_t)height * C.v + vmax * 8 - 1) / ((size_t)vmax * 8));
			const int total = bw * bh;
			for (int m = 0; m < total; ++m) {
				int16_t* blk = &C.coef[((size_t)(m / bw) * C.blocksW + (m % bw)) * 64];
				if (!block(br, C, blk, Ss, Se, Ah, Al)) return false;
				if (m + 1 < total) restartCheck();
			}
		}
		// leave the parser at the next marker (the bit reader never steps over one)
		const uint8_t* q = br.p;
		while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && !(q[1] >= 0xD0 && q[1] <= 0xD7) && q[1] != 0xFF)) ++q;
		consumed = (size_t)(q - data);
		return true;
	}

	bool block(BitReader& br, Component& C, int16_t* blk, int Ss, int Se, int Ah, int Al)
	{
		if (!progressive) {
			const Huff& hd = dc[C.td]; const Huff& ha = ac[C.ta];
			if (!hd.present || !ha.present) return false;
			int s = br.decode(hd); if (s > 16) s = 16;   // DC category: <= 11 in a valid 8-bit stream; a corrupt table can name any byte
			int diff = s ? Extend(br.get(s), s) : 0;
			C.pred += diff; blk[0] = (int16_t)C.pred;
			for (int k = 1; k < 64;) {
				const int rs = br.decode(ha); const int r = rs >> 4; s = rs & 15;
				if (s == 0) { if (r == 15) { k += 16; continue; } break; }
				k += r; if (k > 63) break;
				blk[kZigzag[k]] = (int16_t)Extend(br.get(s), s); ++k;
			}
			return true;
		}
		if (Ss == 0) {
			if (Ah == 0) {
				const Huff& hd = dc[C.td]; if (!hd.present) return false;
				int s = br.decode(hd); if (s > 16) s = 16;
				const int diff = s ? Extend(br.get(s), s) : 0;
				C.pred += diff; blk[0] = (int16_t)(C.pred * (1 << Al));
			} else if (br.bit()) blk[0] = (int16_t)(blk[0] | (1 << Al));
			return true;
		}
		const Huff& ha = ac[C.ta]; if (!ha.present) return false;
		if (Ah == 0) {
			if (eobrun > 0) { --eobrun; return true; }
			for (int k = Ss; k <= Se;) {
				const int rs = br.decode(ha); const int r = rs >> 4, s = rs & 15;
				if (s == 0) {
					if (r == 15) { k += 16; continue; }
					eobrun = (1 << r) - 1; if (r) eobrun += br.get(r);
					break;
				}
				k += r; if (k > 63) break;
				blk[kZigzag[k]] = (int16_t)(Extend(br.get(s), s) * (1 << Al)); ++k;
			}
			return true;
		}
		// AC refinement (T.81 G.1.2.3; control flow of libjpeg's decode_mcu_AC_refine)
		const int p1 = 1 << Al, m1 = -(1 << Al);
		auto correct = [&](int16_t& cf) { if (br.bit() && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1); };
		int k = Ss;
		if (eobrun == 0) {
			for (; k <= Se; ++k) {
				const int rs = br.decode(ha); int r = rs >> 4; const int s = rs & 15;
				int val = 0;
				if (s) val = br.bit() ? p1 : m1;
				else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.get(r); break; }
				do {
					int16_t& cf = blk[kZigzag[k]];
					if (cf != 0) correct(cf);
					else if (--r < 0) break;
					++k;
				} while (k <= Se);
				if (val && k <= 63) blk[kZigzag[k]] = (int16_t)val;
			}
		}
		if (eobrun > 0) {
			for (; k <= Se; ++k) { int16_t& cf = blk[kZigzag[k]]; if (cf != 0) correct(cf); }
			--eobrun;
		}
		return true;
	}

	bool finish(std::vector<uint8_t>& rgba)
	{
		// planes at component resolution
		std::vector<uint8_t> plane[4];
		for (int c = 0; c < ncomp; ++c) {
			Component& C = comp[c];
			if (!qtPresent[C.tq]) return false;
			int32_t mult[64];
			for (int i = 0; i < 64; ++i) mult[i] = (int32_t)(((int32_t)qt[C.tq][i] * (int32_t)kAanScales[i] + (1 << 11)) >> 12);   // DESCALE(q * aan, 14 - 2)
			const int pw = C.blocksW * 8;
			plane[c].assign((size_t)pw * C.blocksH * 8, 0);
			uint8_t px[64];
			for (int by = 0; by < C.blocksH; ++by) for (int bx = 0; bx < C.blocksW; ++bx) {
				IdctIfast(&C.coef[((size_t)by * C.blocksW + bx) * 64], mult, px);
				for (int y = 0; y < 8; ++y) memcpy(&plane[c][(size_t)(by * 8 + y) * pw + bx * 8], px + 8 * y, 8);
			}
		}
		// replication up-sampling (jdsample.c int_upsample / h2v1 / h2v2 without fancy) + jdcolor.c
		// jdapimin.c default_decompress_parms: Adobe transform 0 or component ids 'R','G','B' mean RGB, else YCbCr
		const bool rgbIds = ncomp == 3 && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B';
		const bool ycc = ncomp == 3 && (adobeTransform >= 0 ? adobeTransform != 0 : !rgbIds);
		int crR[256], cbB[256], crG[256], cbG[256];
		for (int i = 0; i < 256; ++i) {
			const int x = i - 128;
			crR[i] = (int)((91881 * x + 32768) >> 16);      // FIX(1.40200)
			cbB[i] = (int)((116130 * x + 32768) >> 16);     // FIX(1.77200)
			crG[i] = -46802 * x;                             // FIX(0.71414)
			cbG[i] = -22554 * x + 32768;                     // FIX(0.34414) + ONE_HALF
		}
		rgba.resize((size_t)width * height * 4);
		for (int y = 0; y < height; ++y) for (int x = 0; x < width; ++x) {
			int v[3] = { 0, 0, 0 };
			for (int c = 0; c < ncomp; ++c) {
				const Component& C = comp[c];
				const int sx = x * C.h / hmax, sy = y * C.v / vmax;
				v[c] = plane[c][(size_t)sy * (C.blocksW * 8) + sx];
			}
			uint8_t* o = &rgba[((size_t)y * width + x) * 4];
			if (ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)v[0]; }
			else if (ycc) {
				o[0] = Clamp255(v[0] + crR[v[2]]);
				o[1] = Clamp255(v[0] + ((cbG[v[1]] + crG[v[2]]) >> 16));
				o[2] = Clamp255(v[0] + cbB[v[1]]);
			} else { o[0] = (uint8_t)v[0]; o[1] = (uint8_t)v[1]; o[2] = (uint8_t)v[2]; }
			o[3] = 255;
		}
		return true;
	}
};

} // namespace

bool DecodeJPEG(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba)
{
	Decoder D; D.d = d.data(); D.n = d.size();
	memset(D.qt, 0, sizeof(D.qt));
	if (!D.parse(rgba)) return false;
	w = (uint32_t)D.width; h = (uint32_t)D.height;
	return true;
}

// Truevision TGA: types 2 / 10 (true colour, raw / RLE) with 24 or 32 bits, 3 / 11 (grey), 1 / 9 (colour-mapped, 24/32-bit
// palette); origin bit honoured.  FreeImage's ConvertTo32Bits keeps a 32-bit file's alpha and sets 255 otherwise.
bool DecodeTGA(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba)
{
	if (d.size() < 18) return false;
	const int idLen = d[0], cmapType = d[1], type = d[2];
	const int cmFirst = d[3] | (d[4] << 8), cmLen = d[5] | (d[6] << 8), cmBits = d[7];
	w = (uint32_t)(d[12] | (d[13] << 8)); h = (uint32_t)(d[14] | (d[15] << 8));
	const int bpp = d[16], desc = d[17];
	if (!w || !h || cmapType > 1) return false;
	if (!PlausibleImageSize(w, h, d.size(), 128)) return false;   // RLE packets: at most 128 pixels per 2 bytes
	const bool rle = type >= 9;
	const int base = rle ? type - 8 : type;
	if (base < 1 || base > 3) return false;
	if (base == 2 && bpp != 24 && bpp != 32) return false;
	if (base == 3 && bpp != 8) return false;
	if (base == 1 && (bpp != 8 || !cmapType || (cmBits != 24 && cmBits != 32))) return false;
	size_t p = 18 + (size_t)idLen;
	const size_t cmBytes = cmapType ? (size_t)cmLen * (cmBits / 8) : 0;
	if (p + cmBytes > d.size()) return false;
	const uint8_t* cmap = d.data() + p; p += cmBytes;
	const int bytes = bpp / 8;
	std::vector<uint8_t> raw((size_t)w * h * bytes);
	if (!rle) {
		if (p + raw.size() > d.size()) return false;
		memcpy(raw.data(), d.data() + p, raw.size());
	} else {
		size_t o = 0;
		while (o < raw.size()) {
			if (p >= d.size()) return false;
			const int hdr = d[p++]; const size_t cnt = (size_t)(hdr & 127) + 1;
			if (hdr & 128) {
				if (p + bytes > d.size()) return false;
				for (size_t i = 0; i < cnt && o < raw.size(); ++i) { memcpy(&raw[o], &d[p], (size_t)bytes); o += bytes; }
				p += bytes;
			} else {
				const size_t nb = cnt * bytes;
				if (p + nb > d.size()) return false;
				const size_t take = std::min(nb, raw.size() - o);
				memcpy(&raw[o], &d[p], take); o += take; p += nb;
			}
		}
	}
	rgba.resize((size_t)w * h * 4);
	const bool topDown = (desc & 0x20) != 0, rightLeft = (desc & 0x10) != 0;
	for (uint32_t y = 0; y < h; ++y) for (uint32_t x = 0; x < w; ++x) {
		const uint32_t sy = topDown ? y : h - 1 - y, sx = rightLeft ? w - 1 - x : x;
		const uint8_t* s = &raw[((size_t)sy * w + sx) * bytes];
		uint8_t* o = &rgba[((size_t)y * w + x) * 4];
		if (base == 2) { o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = bytes == 4 ? s[3] : 255; }
		else if (base == 3) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
		else {
			const int k = (int)s[0] - cmFirst;
			if (k < 0 || k >= cmLen) { o[0] = o[1] = o[2] = 0; o[3] = 255; }
			else { const uint8_t* e = cmap + (size_t)k * (cmBits / 8); o[0] = e[2]; o[1] = e[1]; o[2] = e[0]; o[3] = cmBits == 32 ? e[3] : 255; }
		}
	}
	return true;
}

// ---------------------------------------------------------------------------
// Baseline JPEG writer for Raylib_WriteImageToDisk(..., RAYLIB_IMAGEFILETYPE_Jpg) -- the console front-end saves its
// results as .jpg (reference src/main.cc:480-510).  What FreeImage's flags = 0 produce: quality 75, 4:2:0 chroma,
// the Annex K tables; the exact bytes of a lossy writer are not part of parity (tests: the file decodes, with Pillow and with
// the decoder above, to within the quantisation error of the source).
namespace {

const uint8_t kStdLumaQ[64] = {   // ITU T.81 Table K.1, natural order
	16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
	18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99 };
const uint8_t kStdChromaQ[64] = { // Table K.2
	17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
	99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99 };
// Tables K.3 - K.6: code-length counts and symbols
const uint8_t kDcLumaBits[16] = { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 };
const uint8_t kDcChromaBits[16] = { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 };
const uint8_t kDcVals[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
const uint8_t kAcLumaBits[16] = { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d };
const uint8_t kAcLumaVals[162] = {
	0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
	0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
	0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
	0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
	0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
	0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
	0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };
const uint8_t kAcChromaBits[16] = { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 };
const uint8_t kAcChromaVals[162] = {
	0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
	0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
	0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
	0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
	0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
	0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
	0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa };

struct EncTable { uint16_t code[256]; uint8_t len[256]; };
void BuildEnc(const uint8_t* bits, const uint8_t* vals, EncTable& t)
{
	memset(&t, 0, sizeof(t));
	int code = 0, k = 0;
	for (int l = 1; l <= 16; ++l) { for (int i = 0; i < bits[l - 1]; ++i) { t.code[vals[k]] = (uint16_t)code; t.len[vals[k]] = (uint8_t)l; ++code; ++k; } code <<= 1; }
}
struct BitWriter {
	std::vector<uint8_t>& out; uint32_t acc = 0; int n = 0;
	explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
	void put(uint32_t v, int len) { acc = (acc << len) | (v & ((1u << len) - 1u)); n += len; while (n >= 8) { const uint8_t b = (uint8_t)(acc >> (n - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); n -= 8; } }
	void flush() { if (n) put(0x7F, 8 - n); }
};
void Fdct(const float* in, float* out)
{
	static float c[8][8]; static bool init = false;
	if (!init) { for (int k = 0; k < 8; ++k) for (int x = 0; x < 8; ++x) c[k][x] = (k ? 0.5f : 0.35355339f) * cosf((2 * x + 1) * k * 3.14159265358979f / 16.0f); init = true; }
	float tmp[64];
	for (int y = 0; y < 8; ++y) for (int k = 0; k < 8; ++k) { float s = 0; for (int x = 0; x < 8; ++x) s += in[8 * y + x] * c[k][x]; tmp[8 * y + k] = s; }
	for (int k = 0; k < 8; ++k) for (int x = 0; x < 8; ++x) { float s = 0; for (int y = 0; y < 8; ++y) s += tmp[8 * y + x] * c[k][y]; out[8 * k + x] = s; }
}
void EncodeBlock(BitWriter& bw, const float* px, const uint8_t* q, const EncTable& dc, const EncTable& ac, int& pred)
{
	float f[64]; Fdct(px, f);
	int zz[64];
	for (int i = 0; i < 64; ++i) { const float v = f[kZigzag[i]] / (float)q[kZigzag[i]]; zz[i] = (int)(v < 0 ? v - 0.5f : v + 0.5f); }
	auto magnitude = [](int v, int& bits) { int a = v < 0 ? -v : v, s = 0; while (a) { ++s; a >>= 1; } bits = v < 0 ? v - 1 : v; return s; };
	int bits; const int diff = zz[0] - pred; pred = zz[0];
	int s = magnitude(diff, bits);
	bw.put(dc.code[s], dc.len[s]); if (s) bw.put((uint32_t)bits, s);
	int run = 0;
	for (int k = 1; k < 64; ++k) {
		if (zz[k] == 0) { ++run; continue; }
		while (run > 15) { bw.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
		s = magnitude(zz[k], bits);
		bw.put(ac.code[(run << 4) | s], ac.len[(run << 4) | s]); bw.put((uint32_t)bits, s);
		run = 0;
	}
	if (run) bw.put(ac.code[0], ac.len[0]);
}

} // namespace

bool EncodeJPEG(uint32_t w, uint32_t h, const uint8_t* rgb /* top-down, 3 bytes per pixel */, std::vector<uint8_t>& out)
{
	if (!w || !h || w > 65535 || h > 65535) return false;
	uint8_t ql[64], qc[64];
	for (int i = 0; i < 64; ++i) {   // jcparam.c jpeg_quality_scaling(75) = 50, jpeg_add_quant_table with force_baseline
		int a = (kStdLumaQ[i] * 50 + 50) / 100, b = (kStdChromaQ[i] * 50 + 50) / 100;
		ql[i] = (uint8_t)(a < 1 ? 1 : (a > 255 ? 255 : a)); qc[i] = (uint8_t)(b < 1 ? 1 : (b > 255 ? 255 : b));
	}
	EncTable dcL, dcC, acL, acC;
	BuildEnc(kDcLumaBits, kDcVals, dcL); BuildEnc(kDcChromaBits, kDcVals, dcC); BuildEnc(kAcLumaBits, kAcLumaVals, acL); BuildEnc(kAcChromaBits, kAcChromaVals, acC);
	auto marker = [&](uint8_t m, const std::vector<uint8_t>& body) { out.push_back(0xFF); out.push_back(m); const size_t n = body.size() + 2; out.push_back((uint8_t)(n >> 8)); out.push_back((uint8_t)n); out.insert(out.end(), body.begin(), body.end()); };
	out.clear(); out.push_back(0xFF); out.push_back(0xD8);
	marker(0xE0, { 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0 });
	for (int t = 0; t < 2; ++t) { std::vector<uint8_t> b; b.push_back((uint8_t)t); for (int i = 0; i < 64; ++i) b.push_back((t ? qc : ql)[kZigzag[i]]); marker(0xDB, b); }
	marker(0xC0, { 8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1 });
	auto dht = [&](uint8_t id, const uint8_t* bits, const uint8_t* vals, int n) { std::vector<uint8_t> b; b.push_back(id); b.insert(b.end(), bits, bits + 16); b.insert(b.end(), vals, vals + n); marker(0xC4, b); };
	dht(0x00, kDcLumaBits, kDcVals, 12); dht(0x10, kAcLumaBits, kAcLumaVals, 162); dht(0x01, kDcChromaBits, kDcVals, 12); dht(0x11, kAcChromaBits, kAcChromaVals, 162);
	marker(0xDA, { 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0 });
	BitWriter bw(out);
	int predY = 0, predCb = 0, predCr = 0;
	const uint32_t mw = (w + 15) / 16, mh = (h + 15) / 16;
	for (uint32_t my = 0; my < mh; ++my) for (uint32_t mx = 0; mx < mw; ++mx) {
		float Y[16][16], Cb[16][16], Cr[16][16];
		for (int y = 0; y < 16; ++y) for (int x = 0; x < 16; ++x) {
			const uint32_t sx = std::min(w - 1, mx * 16 + (uint32_t)x), sy = std::min(h - 1, my * 16 + (uint32_t)y);   // edge replication
			const uint8_t* p = rgb + ((size_t)sy * w + sx) * 3;
			const float r = p[0], g = p[1], b = p[2];
			Y[y][x] = 0.299f * r + 0.587f * g + 0.114f * b - 128.0f;
			Cb[y][x] = -0.168736f * r - 0.331264f * g + 0.5f * b;
			Cr[y][x] = 0.5f * r - 0.418688f * g - 0.081312f * b;
		}
		float blk[64];
		for (int by = 0; by < 2; ++by) for (int bx = 0; bx < 2; ++bx) {
			for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) blk[8 * y + x] = Y[by * 8 + y][bx * 8 + x];
			EncodeBlock(bw, blk, ql, dcL, acL, predY);
		}
		for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) blk[8 * y + x] = 0.25f * (Cb[2 * y][2 * x] + Cb[2 * y][2 * x + 1] + Cb[2 * y + 1][2 * x] + Cb[2 * y + 1][2 * x + 1]);
		EncodeBlock(bw, blk, qc, dcC, acC, predCb);
		for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) blk[8 * y + x] = 0.25f * (Cr[2 * y][2 * x] + Cr[2 * y][2 * x + 1] + Cr[2 * y + 1][2 * x] + Cr[2 * y + 1][2 * x + 1]);
		EncodeBlock(bw, blk, qc, dcC, acC, predCr);
	}
	bw.flush();
	out.push_back(0xFF); out.push_back(0xD9);
	return true;
}

} // namespace rl
