// Image container + file codecs for the MI355X raylib (host C++).
//
// The reference goes through FreeImage 3.18 loaded at run time (reference
// loader/dll_loader.cc:21-54, render/image.cc:152-257), which is Windows-only
// plumbing around a third-party library.  Here: self-contained BMP and PNG
// (zlib) codecs that produce the same in-memory image the reference builds from
// FreeImage's output: float RGBA = byte/255 (render/image.h:37-43), alpha 1 for
// 24-bit sources (ConvertTo32Bits), row 0 = top (image.cc:203-226).
// JPEG is not decoded (returns null, like a failed FreeImage load: image.cc:161-165).
#include "rl_host.h"
#include <exception>
#include <limits.h>

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>
#include <string>

#include <atomic>

namespace rl {

uint64_t NextImageVersion()
{
	static std::atomic<uint64_t> next{1};
	return next.fetch_add(1, std::memory_order_relaxed);
}

Image::~Image() { if (devPixels) DeviceFreePixels(devPixels); }

void Image::SyncHost() const
{
	if (!hostStale) return;
	Image& self = const_cast<Image&>(*this);
	if (!DeviceReadback(self)) Log("Image: the device copy could not be read back; host pixels are stale");
	self.hostStale = false;
}

void Image::Reallocate(uint32_t w, uint32_t h, float r, float g, float b, float a)
{
	SyncHost();
	devValid = false;
	Touch();
	// reference render/image.cc:29-34: vector::resize keeps existing pixels, new ones get the clear colour
	width = w; height = h;
	size_t old = rgba.size() / 4, now = (size_t)w * h;
	rgba.resize(now * 4);
	for (size_t i = old; i < now; ++i) { rgba[4 * i] = r; rgba[4 * i + 1] = g; rgba[4 * i + 2] = b; rgba[4 * i + 3] = a; }
}

namespace {

bool ReadFile(const char* path, std::vector<uint8_t>& out)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	if (n < 0) { fclose(f); return false; }
	out.resize((size_t)n);
	size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
	fclose(f);
	return got == (size_t)n;
}
inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

Image* FromBytesRGBA(uint32_t w, uint32_t h, const std::vector<uint8_t>& px /* top-down RGBA8 */)
{
	Image* img = new Image;
	img->width = w; img->height = h;
	img->rgba.resize((size_t)w * h * 4);
	for (size_t i = 0; i < (size_t)w * h * 4; ++i) img->rgba[i] = (float)px[i] / 255.0f;
	return img;
}

Image* LoadBMP(const std::vector<uint8_t>& d)
{
	if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return nullptr;
	uint32_t off = le32(&d[10]);
	int32_t w = (int32_t)le32(&d[18]), h = (int32_t)le32(&d[22]);
	uint16_t bpp = le16(&d[28]);
	uint32_t comp = le32(&d[30]);
	if ((bpp != 24 && bpp != 32) || (comp != 0 && comp != 3) || w <= 0 || h == 0) return nullptr;
	if (h == INT32_MIN) return nullptr;
	bool bottomUp = h > 0; if (h < 0) h = -h;
	if (!PlausibleImageSize((uint64_t)w, (uint64_t)h, d.size(), 1)) return nullptr;   // uncompressed: at least 3 bytes per pixel
	size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
	if (off > d.size() || d.size() - off < stride * (size_t)h) return nullptr;
	std::vector<uint8_t> px((size_t)w * h * 4);
	for (int32_t y = 0; y < h; ++y) {
		const uint8_t* row = &d[off + stride * (bottomUp ? (h - 1 - y) : y)];
		for (int32_t x = 0; x < w; ++x) {
			const uint8_t* s = row + (size_t)x * (bpp / 8);
			uint8_t* o = &px[((size_t)y * w + x) * 4];
			o[0] = s[2]; o[1] = s[1]; o[2] = s[0]; o[3] = (bpp == 32) ? s[3] : 255;
		}
	}
	return FromBytesRGBA((uint32_t)w, (uint32_t)h, px);
}

inline int Paeth(int a, int b, int c) { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

Image* LoadPNG(const std::vector<uint8_t>& d)
{
	static const uint8_t sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
	if (d.size() < 33 || memcmp(d.data(), sig, 8) != 0) return nullptr;
	uint32_t w = 0, h = 0; uint8_t depth = 0, color = 0, interlace = 0;
	std::vector<uint8_t> idat, palette, trns;
	size_t p = 8;
	while (p + 12 <= d.size()) {
		uint32_t len = be32(&d[p]); const uint8_t* type = &d[p + 4]; const uint8_t* body = &d[p + 8];
		if (p + 12 + len > d.size()) return nullptr;
		if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12]; }
		else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
		else if (!memcmp(type, "IEND", 4)) break;
		p += 12 + len;
	}
	if (!w || !h || depth != 8 || interlace != 0) return nullptr;
	int ch = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
	if (!ch) return nullptr;
	if (!PlausibleImageSize(w, h, idat.size(), 1100)) return nullptr;   // deflate expands by at most ~1032:1
	size_t stride = (size_t)w * ch;
	std::vector<uint8_t> raw((stride + 1) * h);
	uLongf rawLen = (uLongf)raw.size();
	if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return nullptr;
	std::vector<uint8_t> img(stride * h);
	for (uint32_t y = 0; y < h; ++y) {
		uint8_t ft = raw[(stride + 1) * y];
		const uint8_t* s = &raw[(stride + 1) * y + 1];
		uint8_t* o = &img[stride * y];
		const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
		for (size_t i = 0; i < stride; ++i) {
			int a = i >= (size_t)ch ? o[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
			int v = s[i];
			switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += Paeth(a, b, c); break; default: break; }
			o[i] = (uint8_t)v;
		}
	}
	std::vector<uint8_t> px((size_t)w * h * 4);
	for (size_t i = 0; i < (size_t)w * h; ++i) {
		const uint8_t* s = &img[i * ch]; uint8_t* o = &px[i * 4];
		switch (color) {
			case 0: o[0] = o[1] = o[2] = s[0]; o[3] = 255; break;
			case 2: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = 255; break;
			case 3: { size_t k = s[0]; o[0] = 3 * k + 2 < palette.size() ? palette[3 * k] : 0; o[1] = 3 * k + 2 < palette.size() ? palette[3 * k + 1] : 0;
			          o[2] = 3 * k + 2 < palette.size() ? palette[3 * k + 2] : 0; o[3] = k < trns.size() ? trns[k] : 255; break; }
			case 4: o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; break;
			default: o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; o[3] = s[3]; break;
		}
	}
	return FromBytesRGBA(w, h, px);
}

// Radiance .hdr (RGBE), the usual container of sky panoramas.  The reference reads it through FreeImage and
// ConvertToRGBAF (render/image.cc:168-193): float RGB = mantissa * 2^(e - 136) (no +0.5 bias), alpha 1, file rows top to bottom.
// [parity unpinned: FreeImage is absent; the decode is restated from the format's definition]
Image* LoadHDR(const std::vector<uint8_t>& d)
{
	if (d.size() < 11 || (memcmp(d.data(), "#?RADIANCE", 10) != 0 && memcmp(d.data(), "#?RGBE", 6) != 0)) return nullptr;
	size_t p = 0;
	auto line = [&](std::string& out) { out.clear(); while (p < d.size() && d[p] != '\n') out.push_back((char)d[p++]); if (p < d.size()) ++p; return p <= d.size(); };
	std::string ln;
	bool formatOk = false;
	while (line(ln)) {
		if (ln.empty()) break;
		if (ln.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) formatOk = true;
	}
	if (!formatOk || !line(ln)) return nullptr;
	int w = 0, h = 0;
	if (sscanf(ln.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return nullptr;
	if (!PlausibleImageSize((uint64_t)w, (uint64_t)h, d.size(), 128)) return nullptr;   // RLE: a run byte pair covers at most 127 pixels of one channel
	std::vector<uint8_t> row((size_t)w * 4);
	Image* img = new Image;
	img->width = (uint32_t)w; img->height = (uint32_t)h;
	img->rgba.resize((size_t)w * h * 4);
	for (int y = 0; y < h; ++y) {
		if (p + 4 > d.size()) { delete img; return nullptr; }
		if (w >= 8 && w < 32768 && d[p] == 2 && d[p + 1] == 2 && ((d[p + 2] << 8) | d[p + 3]) == w) {
			p += 4;   // new-style RLE: the four channels of the scanline one after the other
			for (int ch = 0; ch < 4; ++ch) {
				int x = 0;
				while (x < w) {
					if (p >= d.size()) { delete img; return nullptr; }
					int cnt = d[p++];
					if (cnt > 128) { cnt -= 128; if (p >= d.size() || x + cnt > w) { delete img; return nullptr; } uint8_t v = d[p++]; while (cnt--) row[4 * (x++) + ch] = v; }
					else { if (cnt == 0 || p + cnt > d.size() || x + cnt > w) { delete img; return nullptr; } while (cnt--) row[4 * (x++) + ch] = d[p++]; }
				}
			}
		} else {
			if (p + (size_t)w * 4 > d.size()) { delete img; return nullptr; }
			memcpy(row.data(), &d[p], (size_t)w * 4); p += (size_t)w * 4;   // flat scanline
		}
		for (int x = 0; x < w; ++x) {
			float* o = &img->rgba[((size_t)y * w + x) * 4];
			const uint8_t e = row[4 * x + 3];
			if (e) { const float f = ldexpf(1.0f, (int)e - (128 + 8)); o[0] = row[4 * x] * f; o[1] = row[4 * x + 1] * f; o[2] = row[4 * x + 2] * f; }
			else o[0] = o[1] = o[2] = 0.0f;
			o[3] = 1.0f;
		}
	}
	return img;
}

// reference render/image.h:62-69: (uint32)(c * 255.0f) & 0xff, no clamping
inline uint8_t ToByte(float c) { return (uint8_t)((uint32_t)(c * 255.0f) & 0xff); }

void Put32(std::vector<uint8_t>& v, uint32_t x, bool be) {
	if (be) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
	else { v.push_back(x); v.push_back(x >> 8); v.push_back(x >> 16); v.push_back(x >> 24); }
}

bool WriteBMP(const Image& img, const char* path)
{
	const uint32_t w = img.width, h = img.height;
	const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
	std::vector<uint8_t> out;
	out.push_back('B'); out.push_back('M');
	Put32(out, (uint32_t)(54 + stride * h), false); Put32(out, 0, false); Put32(out, 54, false);
	Put32(out, 40, false); Put32(out, w, false); Put32(out, h, false);
	out.push_back(1); out.push_back(0); out.push_back(24); out.push_back(0);
	Put32(out, 0, false); Put32(out, (uint32_t)(stride * h), false); Put32(out, 2835, false); Put32(out, 2835, false); Put32(out, 0, false); Put32(out, 0, false);
	for (int32_t y = (int32_t)h - 1; y >= 0; --y) {
		size_t rowStart = out.size();
		for (uint32_t x = 0; x < w; ++x) {
			const float* p = &img.rgba[((size_t)y * w + x) * 4];
			out.push_back(ToByte(p[2])); out.push_back(ToByte(p[1])); out.push_back(ToByte(p[0]));
		}
		while (out.size() - rowStart < stride) out.push_back(0);
	}
	FILE* f = fopen(path, "wb");
	if (!f) return false;
	bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
	fclose(f);
	return ok;
}

void PngChunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body)
{
	Put32(out, (uint32_t)body.size(), true);
	size_t start = out.size();
	out.insert(out.end(), type, type + 4);
	out.insert(out.end(), body.begin(), body.end());
	uint32_t crc = (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start));
	Put32(out, crc, true);
}

bool WritePNG(const Image& img, const char* path)
{
	const uint32_t w = img.width, h = img.height;
	std::vector<uint8_t> raw; raw.reserve(((size_t)w * 3 + 1) * h);
	for (uint32_t y = 0; y < h; ++y) {
		raw.push_back(0);
		for (uint32_t x = 0; x < w; ++x) {
			const float* p = &img.rgba[((size_t)y * w + x) * 4];
			raw.push_back(ToByte(p[0])); raw.push_back(ToByte(p[1])); raw.push_back(ToByte(p[2]));
		}
	}
	uLongf clen = compressBound((uLong)raw.size());
	std::vector<uint8_t> comp(clen);
	if (compress(comp.data(), &clen, raw.data(), (uLong)raw.size()) != Z_OK) return false;
	comp.resize(clen);
	std::vector<uint8_t> out = { 137, 80, 78, 71, 13, 10, 26, 10 };
	std::vector<uint8_t> ihdr; Put32(ihdr, w, true); Put32(ihdr, h, true);
	ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
	PngChunk(out, "IHDR", ihdr); PngChunk(out, "IDAT", comp); PngChunk(out, "IEND", {});
	FILE* f = fopen(path, "wb");
	if (!f) return false;
	bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
	fclose(f);
	return ok;
}

} // namespace

Image* LoadImageFile(const char* path)
{
	if (path == nullptr) return nullptr;
	try {
	std::vector<uint8_t> d;
	if (!ReadFile(path, d)) return nullptr;
	if (Image* i = LoadPNG(d)) return i;
	if (Image* i = LoadBMP(d)) return i;
	if (Image* i = LoadHDR(d)) return i;
	{
		uint32_t w = 0, h = 0; std::vector<uint8_t> px;
		if (DecodeJPEG(d, w, h, px)) return FromBytesRGBA(w, h, px);
		const size_t len = strlen(path);
		if (len > 4 && (path[len - 3] == 't' || path[len - 3] == 'T') && (path[len - 2] == 'g' || path[len - 2] == 'G') && (path[len - 1] == 'a' || path[len - 1] == 'A') &&
		    DecodeTGA(d, w, h, px)) return FromBytesRGBA(w, h, px);   // TGA has no signature: by extension, as FreeImage_GetFIFFromFilename does
	}
	Log("LoadImage: unsupported image format: %s (BMP, 8-bit PNG, JPEG (8-bit Huffman), TGA and Radiance HDR are decoded)", path);
	} catch (const std::exception& e) {   // out of memory on a huge (but plausible) image: never across the C ABI
		Log("LoadImage: %s: %s", path, e.what());
	}
	return nullptr;
}

bool WriteImageFile(const Image& img, const char* path, uint32_t fileType)
{
	img.SyncHost();
	switch (fileType) {
		case RAYLIB_IMAGEFILETYPE_Bitmap: return WriteBMP(img, path);
		case RAYLIB_IMAGEFILETYPE_Png:    return WritePNG(img, path);
		case RAYLIB_IMAGEFILETYPE_Jpg: {
			std::vector<uint8_t> rgb((size_t)img.width * img.height * 3), file;
			for (size_t i = 0; i < (size_t)img.width * img.height; ++i) for (int k = 0; k < 3; ++k) rgb[3 * i + k] = ToByte(img.rgba[4 * i + k]);
			if (!EncodeJPEG(img.width, img.height, rgb.data(), file)) return false;
			FILE* f = fopen(path, "wb");
			if (!f) return false;
			const bool ok = fwrite(file.data(), 1, file.size(), f) == file.size();
			fclose(f);
			return ok;
		}
		default: Log("WriteImageToDisk: unknown file type %u", fileType); return false;
	}
}

} // namespace rl
