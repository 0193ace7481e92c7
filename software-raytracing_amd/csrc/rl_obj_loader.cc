// Wavefront OBJ / MTL ingestion for the MI355X raylib (host C++, as north_star asks:
// "host-side scene build and OBJ load stay C++").
//
// The reference delegates parsing to tinyobjloader v2.0.0rc10 (reference
// loader/obj_loader.cc:10-11,91) which is not vendored; this is an independent
// parser that yields what the reference's loader consumes at its call sites
// (obj_loader.cc:133-234): per shape, triangles with positions, optional
// texcoords / normals, and a per-face material id.  What the reference then does
// with that data IS reproduced rule by rule:
//   - missing vertex normal on any corner -> flat face normal for all three (:199-203)
//   - missing texcoord -> (0,0)                                          (:163-173)
//   - face without a valid material -> Lambertian(0.5)                   (:113,206-211)
//   - non-triangle faces: the reference skips them after tinyobjloader has already
//     triangulated (triangulate defaults to true), so polygons arrive as fans
//   - MTL -> material mapping                                             (:354-397)
#include "rl_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <stdint.h>
#include <string.h>
#include <map>

namespace rl {
namespace {

struct MtlRecord {
	std::string name;
	float Kd[3] = { 0, 0, 0 }, Ks[3] = { 0, 0, 0 }, Ke[3] = { 0, 0, 0 }, Tf[3] = { 0, 0, 0 };
	float Ns = 1.0f, Ni = 1.0f, Pr = 0.0f, Pm = 0.0f;   // tinyobjloader InitMaterial defaults
	int illum = 0;
	bool hasKd = false;
	std::string map_Kd, map_Pr, map_Pm, map_Ke, norm, bump;
};

// Tokeniser over one line: whitespace separated, '#' starts a comment.
struct Tokens {
	std::vector<std::string> t;
	explicit Tokens(const char* line) {
		const char* p = line;
		while (*p) {
			while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
			if (!*p || *p == '#') break;
			const char* b = p;
			while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p;
			t.emplace_back(b, p - b);
		}
	}
	size_t size() const { return t.size(); }
	const std::string& operator[](size_t i) const { return t[i]; }
	float f(size_t i, float dflt = 0.0f) const { return i < t.size() ? strtof(t[i].c_str(), nullptr) : dflt; }
};

bool ReadLines(const std::string& path, std::vector<std::string>& out)
{
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return false;
	std::string cur;
	char buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof(buf), f)) > 0) {
		for (size_t i = 0; i < n; ++i) {
			if (buf[i] == '\n') { out.push_back(cur); cur.clear(); }
			else cur.push_back(buf[i]);
		}
	}
	if (!cur.empty()) out.push_back(cur);
	fclose(f);
	return true;
}

std::string DirOf(const std::string& path)
{
	size_t p = path.find_last_of("/\\");
	return p == std::string::npos ? std::string() : path.substr(0, p + 1);
}

void ParseMTL(const std::string& path, std::vector<MtlRecord>& out)
{
	std::vector<std::string> lines;
	if (!ReadLines(path, lines)) { Log("OBJ: cannot open material library %s", path.c_str()); return; }
	MtlRecord* cur = nullptr;
	for (const std::string& line : lines) {
		Tokens tk(line.c_str());
		if (tk.size() == 0) continue;
		const std::string& k = tk[0];
		if (k == "newmtl") { out.emplace_back(); cur = &out.back(); cur->name = tk.size() > 1 ? tk[1] : ""; continue; }
		if (!cur) continue;
		auto rgb = [&](float* dst) { dst[0] = tk.f(1); dst[1] = tk.f(2, dst[0]); dst[2] = tk.f(3, dst[0]); };
		if (k == "Kd") { rgb(cur->Kd); cur->hasKd = true; }
		else if (k == "Ks") rgb(cur->Ks);
		else if (k == "Ke") rgb(cur->Ke);
		else if (k == "Tf" || k == "Kt") rgb(cur->Tf);
		else if (k == "Ns") cur->Ns = tk.f(1);
		else if (k == "Ni") cur->Ni = tk.f(1);
		else if (k == "Pr") cur->Pr = tk.f(1);
		else if (k == "Pm") cur->Pm = tk.f(1);
		else if (k == "illum") cur->illum = tk.size() > 1 ? atoi(tk[1].c_str()) : 0;
		else if (k == "map_Kd") cur->map_Kd = tk[tk.size() - 1];
		else if (k == "map_Pr") cur->map_Pr = tk[tk.size() - 1];
		else if (k == "map_Pm") cur->map_Pm = tk[tk.size() - 1];
		else if (k == "map_Ke") cur->map_Ke = tk[tk.size() - 1];
		else if (k == "norm") cur->norm = tk[tk.size() - 1];
		else if (k == "map_bump" || k == "map_Bump" || k == "bump") cur->bump = tk[tk.size() - 1];
	}
	for (MtlRecord& m : out)
		if (!m.map_Kd.empty() && !m.hasKd) m.Kd[0] = m.Kd[1] = m.Kd[2] = 0.6f;   // tinyobjloader default for textured materials without Kd
}

inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

// reference loader/obj_loader.cc:354-397
HostMaterial MaterialFromMTL(const MtlRecord& r)
{
	HostMaterial m; memset(&m, 0, sizeof(m));
	for (int i = 0; i < 5; ++i) m.tex[i] = -1;
	float albedo[3];
	for (int i = 0; i < 3; ++i) albedo[i] = r.Kd[i] < 0.95f ? r.Kd[i] : 0.95f;   // min(MAX_ALBEDO, Kd), :29,:354
	const bool bTransparentIllum = (r.illum == 4 || r.illum == 6);
	const bool bZeroDiffuse = r.map_Kd.empty() && albedo[0] == 0.0f && albedo[1] == 0.0f && albedo[2] == 0.0f;
	if (bTransparentIllum && bZeroDiffuse) {
		m.type = MAT_DIELECTRIC;
		m.ior = r.Ni;
		for (int i = 0; i < 3; ++i) m.transmission[i] = r.Tf[i];
	} else if (r.illum == 3) {
		m.type = MAT_MIRROR;
		for (int i = 0; i < 3; ++i) m.albedo[i] = albedo[i];
	} else {
		m.type = MAT_MICROFACET;
		for (int i = 0; i < 3; ++i) m.albedo[i] = clamp01(albedo[i]);             // SetAlbedoFallback saturates (material.h:236)
		float rough;
		if (r.Pr > 0.0f) rough = r.Pr;
		else {
			float intensity = (r.Ks[0] + r.Ks[1] + r.Ks[2]) / 3.0f;                 // PhongSpecularToRoughness, :37-41
			rough = sqrtf(2.0f / (r.Ns * intensity + 2.0f));
		}
		m.roughness = clamp01(rough);
		m.metallic = clamp01(r.Pm);
		for (int i = 0; i < 3; ++i) m.emissive[i] = r.Ke[i];
	}
	return m;
}

// strtof's value for the token [b, e).  Fast paths (Clinger) for plain decimals "digits[.digits]": (1) the digits form an
// integer m < 2^24 and the scale is 10^k, k <= 10 -- both exact floats, one IEEE division gives the correctly rounded result;
// (2) m < 2^53, k <= 22 -- the same in double, then a rounding to float that is provably single unless the double lies within
// an ulp of a float midpoint.  Everything else (exponents, hex, inf/nan, 17+ digits, midpoints) goes to strtof itself.
float ParseFloat(const char* b, const char* e)
{
	static const float kPow10f[11] = { 1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f };
	static const double kPow10d[23] = { 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22 };
	const char* p = b;
	bool neg = false;
	if (p < e && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
	uint64_t m = 0; int digits = 0, frac = 0; bool ok = p < e;
	for (; p < e && *p >= '0' && *p <= '9'; ++p) { if (m > 900719925474098ull) { ok = false; break; } m = m * 10u + (uint64_t)(*p - '0'); ++digits; }
	if (ok && p < e && *p == '.') {
		++p;
		for (; p < e && *p >= '0' && *p <= '9'; ++p) { if (m > 900719925474098ull) { ok = false; break; } m = m * 10u + (uint64_t)(*p - '0'); ++digits; ++frac; }
	}
	if (!ok || p != e || digits == 0 || frac > 22) return strtof(b, nullptr);
	if (m < (1ull << 24) && frac <= 10) { const float v = (float)(uint32_t)m / kPow10f[frac]; return neg ? -v : v; }
	// m < 2^53 and 10^frac are exact doubles: d is the correctly rounded double of the decimal.  Rounding d to float gives the
	// correctly rounded float unless d sits within one double-ulp of the midpoint of two floats; those go to strtof.
	const double d = (double)m / kPow10d[frac];
	if (!(d > 1e-30 && d < 1e30)) return strtof(b, nullptr);
	uint64_t bits; memcpy(&bits, &d, 8);
	const uint64_t low = bits & ((1ull << 29) - 1);
	if (low >= (1ull << 28) - 1 && low <= (1ull << 28) + 1) return strtof(b, nullptr);
	const float v = (float)d;
	return neg ? -v : v;
}

struct Corner { int v, vt, vn; };

bool ParseCorner(const char* p, int nV, int nVT, int nVN, Corner& c)
{
	c.v = c.vt = c.vn = -1;
	char* end;
	long a = strtol(p, &end, 10);
	if (end == p) return false;
	c.v = a > 0 ? (int)a - 1 : nV + (int)a;
	if (*end == '/') {
		p = end + 1;
		if (*p != '/') { long b = strtol(p, &end, 10); if (end != p) c.vt = b > 0 ? (int)b - 1 : nVT + (int)b; }
		else end = (char*)p;
		if (*end == '/') {
			p = end + 1;
			long d = strtol(p, &end, 10);
			if (end != p) c.vn = d > 0 ? (int)d - 1 : nVN + (int)d;
		}
	}
	return c.v >= 0 && c.v < nV;
}

} // namespace

float ParseDecimalFloat(const char* token) { return ParseFloat(token, token + strlen(token)); }

bool LoadOBJ(const char* path, OBJModel& out)
{
	if (path == nullptr) { Log("LoadOBJ: filepath was null"); return false; }
	// The whole file in one NUL-terminated buffer, parsed in place: no per-line / per-token allocations (a 10 M-triangle
	// OBJ is 2.7 GB of text; the string-per-token parser spent 3x the BVH build's time here).  Token rules as before:
	// whitespace separated, a token starting with '#' ends the line, numbers by strtof / strtol at the token start.
	const auto tl0 = std::chrono::steady_clock::now();
	std::vector<char> text;
	{
		FILE* fp = fopen(path, "rb");
		if (!fp) { Log("LoadOBJ: cannot open %s", path); return false; }
		fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
		if (sz < 0) { fclose(fp); Log("LoadOBJ: cannot open %s", path); return false; }
		text.resize((size_t)sz + 1);
		const size_t got = sz ? fread(text.data(), 1, (size_t)sz, fp) : 0;
		fclose(fp);
		text[got] = 0;
		for (size_t i = 0; i < got; ++i) if (text[i] == 0) text[i] = ' ';   // stray NULs would end the C-string parsing early
	}
	const std::string dir = DirOf(path);

	std::vector<float> V, VT, VN;
	std::vector<MtlRecord> mtl;
	std::map<std::string, int> mtlIndex;
	struct Face { Corner c[3]; int material; int shape; };
	std::vector<Face> faces;
	int curMaterial = -1;
	int curShape = -1, numShapes = 0;
	bool shapeHasFaces = false;
	std::vector<Corner> cs;

	auto isSpace = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\r'; };
	// next token of the current line: [b, e); false at end of line / comment
	auto nextToken = [&](const char*& p, const char*& b, const char*& e) -> bool {
		while (isSpace(*p)) ++p;
		if (*p == 0 || *p == '\n' || *p == '#') return false;
		b = p;
		while (*p && *p != '\n' && !isSpace(*p)) ++p;
		e = p;
		return true;
	};
	auto number = [&](const char*& p, float dflt) -> float {
		const char* b; const char* e;
		if (!nextToken(p, b, e)) return dflt;
		return ParseFloat(b, e);
	};
	auto is = [](const char* b, const char* e, const char* word) { const size_t n = strlen(word); return (size_t)(e - b) == n && memcmp(b, word, n) == 0; };

	const auto tl1 = std::chrono::steady_clock::now();
	const char* p = text.data();
	while (*p) {
		const char* b; const char* e;
		if (nextToken(p, b, e)) {
			if (is(b, e, "v")) { const float x = number(p, 0.0f), y = number(p, 0.0f), z = number(p, 0.0f); V.push_back(x); V.push_back(y); V.push_back(z); }
			else if (is(b, e, "vt")) { const float u = number(p, 0.0f), v = number(p, 0.0f); VT.push_back(u); VT.push_back(v); }
			else if (is(b, e, "vn")) { const float x = number(p, 0.0f), y = number(p, 0.0f), z = number(p, 0.0f); VN.push_back(x); VN.push_back(y); VN.push_back(z); }
			else if (is(b, e, "o") || is(b, e, "g")) {
				// a new shape starts; shapes that end up without faces are dropped
				if (shapeHasFaces || curShape < 0) { curShape = numShapes++; }
				shapeHasFaces = false;
			}
			else if (is(b, e, "usemtl")) {
				const char* nb; const char* ne;
				auto it = nextToken(p, nb, ne) ? mtlIndex.find(std::string(nb, ne)) : mtlIndex.end();
				curMaterial = it == mtlIndex.end() ? -1 : it->second;
			}
			else if (is(b, e, "mtllib")) {
				const char* nb; const char* ne;
				while (nextToken(p, nb, ne)) {
					size_t before = mtl.size();
					ParseMTL(dir + std::string(nb, ne), mtl);
					for (size_t j = before; j < mtl.size(); ++j) mtlIndex[mtl[j].name] = (int)j;
				}
			}
			else if (is(b, e, "f")) {
				if (curShape < 0) { curShape = numShapes++; }
				cs.clear();
				bool ok = true;
				const char* cb; const char* ce;
				while (nextToken(p, cb, ce)) {
					Corner c;
					if (!ParseCorner(cb, (int)V.size() / 3, (int)VT.size() / 2, (int)VN.size() / 3, c)) { ok = false; break; }
					cs.push_back(c);
				}
				if (ok && cs.size() >= 3) {
					for (size_t j = 1; j + 1 < cs.size(); ++j) {   // triangle fan
						Face f; f.c[0] = cs[0]; f.c[1] = cs[j]; f.c[2] = cs[j + 1]; f.material = curMaterial; f.shape = curShape;
						faces.push_back(f);
					}
					shapeHasFaces = true;
				}
			}
		}
		while (*p && *p != '\n') ++p;   // rest of the line
		if (*p == '\n') ++p;
	}
	text.clear(); text.shrink_to_fit();
	const auto tl2 = std::chrono::steady_clock::now();
	if (faces.empty()) { Log("LoadOBJ: No shapes found in: %s", path); return false; }

	// compact shape ids (drop empty shapes)
	std::vector<int> remap(numShapes, -1);
	int nShapes = 0;
	for (const Face& f : faces) if (remap[f.shape] < 0) remap[f.shape] = 0;
	for (int i = 0; i < numShapes; ++i) if (remap[i] == 0) remap[i] = nShapes++;

	out.materials.clear(); out.materialNames.clear(); out.images.clear(); out.triangles.clear();
	std::map<std::string, int> imageIndex;
	auto texture = [&](const std::string& fn) -> int {
		if (fn.empty()) return -1;
		auto it = imageIndex.find(fn);
		if (it != imageIndex.end()) return it->second;
		Image* img = LoadImageFile((dir + fn).c_str());
		int ix = -1;
		if (img) { ix = (int)out.images.size(); out.images.emplace_back(img); }
		else Log("LoadOBJ: texture %s could not be loaded; falling back to constants", (dir + fn).c_str());
		imageIndex[fn] = ix;
		return ix;
	};
	for (const MtlRecord& r : mtl) {
		HostMaterial m = MaterialFromMTL(r);
		if (m.type == MAT_MICROFACET) {
			m.tex[0] = texture(r.map_Kd);
			m.tex[1] = texture(r.norm); if (m.tex[1] < 0) m.tex[1] = texture(r.bump);
			m.tex[2] = texture(r.map_Pr);
			m.tex[3] = texture(r.map_Pm);
			m.tex[4] = texture(r.map_Ke);
		}
		out.materials.push_back(m);
		out.materialNames.push_back(r.name);
	}
	// fallback for faces without a material: Lambertian(0.5) (obj_loader.cc:113)
	{
		HostMaterial fb; memset(&fb, 0, sizeof(fb));
		fb.type = MAT_LAMBERTIAN; fb.albedo[0] = fb.albedo[1] = fb.albedo[2] = 0.5f;
		for (int i = 0; i < 5; ++i) fb.tex[i] = -1;
		out.materials.push_back(fb);
		out.materialNames.push_back("");
	}
	const int fallback = (int)out.materials.size() - 1;

	out.triangles.reserve(faces.size());
	int numInvalidTexcoords = 0;
	for (const Face& f : faces) {
		HostTriangle t; memset(&t, 0, sizeof(t));
		f3 pos[3], nrm[3]; float tu[3], tv[3];
		bool validNormal = true;
		for (int c = 0; c < 3; ++c) {
			const Corner& k = f.c[c];
			pos[c] = F3(V[3 * k.v], V[3 * k.v + 1], V[3 * k.v + 2]);
			if (k.vt >= 0 && k.vt < (int)VT.size() / 2) { tu[c] = VT[2 * k.vt]; tv[c] = VT[2 * k.vt + 1]; }
			else { tu[c] = tv[c] = 0.0f; ++numInvalidTexcoords; }
			if (k.vn >= 0 && k.vn < (int)VN.size() / 3) nrm[c] = F3(VN[3 * k.vn], VN[3 * k.vn + 1], VN[3 * k.vn + 2]);
			else { nrm[c] = F3(0, 0, 0); validNormal = false; }
		}
		if (!validNormal) {
			f3 n = normalize(cross(pos[1] - pos[0], pos[2] - pos[0]));
			nrm[0] = nrm[1] = nrm[2] = n;
		}
		t.v0 = pos[0]; t.v1 = pos[1]; t.v2 = pos[2];
		t.n0 = nrm[0]; t.n1 = nrm[1]; t.n2 = nrm[2];
		t.s0 = tu[0]; t.t0 = tv[0]; t.s1 = tu[1]; t.t1 = tv[1]; t.s2 = tu[2]; t.t2 = tv[2];
		t.material = (f.material >= 0 && f.material < fallback) ? f.material : fallback;
		t.shape = remap[f.shape];
		out.triangles.push_back(t);
	}
	out.numShapes = nShapes;
	out.finalized = false;
	if (getenv("RAYLIB_BUILD_TIMING")) Log("LoadOBJ: read %.2f s, parse %.2f s, materials + triangle assembly %.2f s", std::chrono::duration<double>(tl1 - tl0).count(),
		std::chrono::duration<double>(tl2 - tl1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tl2).count());
	Log("LoadOBJ: Load %s", path);
	Log("\tTotal shapes: %d", nShapes);
	Log("\tTotal vertices: %d", (int)(V.size() / 3));
	Log("\tTotal materials: %d", (int)mtl.size());
	if (numInvalidTexcoords > 0) Log("WARNING: Num triangles with invalid UVs: %d", numInvalidTexcoords);
	return true;
}

// reference raylib.cc:71-90 + geom/static_mesh.cc:54-78 + geom/transform.cc:47-65,88-95:
// positions: rotate -> scale -> translate; normals: rotate only.  Ignored once finalized.
void TransformOBJ(OBJModel& m, float tx, float ty, float tz, float yaw, float pitch, float roll, float sx, float sy, float sz)
{
	if (m.finalized) return;
	const float pi_f = (float)3.1415926535897932385;
	const float rad_yaw = yaw * pi_f / 180.0f, rad_pitch = pitch * pi_f / 180.0f, rad_roll = roll * pi_f / 180.0f;
	const float ch = cosf(rad_yaw), sh = sinf(rad_yaw), cp = cosf(rad_pitch), sp = sinf(rad_pitch), cb = cosf(rad_roll), sb = sinf(rad_roll);
	const f3 M0 = F3(ch * cb + sh * sp * sb, sb * cp, -sh * cb + ch * sp * sb);
	const f3 M1 = F3(-ch * sb + sh * sp * cb, cb * cp, sb * sh + ch * sp * cb);
	const f3 M2 = F3(sh * cp, -sp, ch * cp);
	auto rot = [&](f3 p) { return F3(dot(M0, p), dot(M1, p), dot(M2, p)); };
	const f3 scale = F3(sx, sy, sz), loc = F3(tx, ty, tz);
	for (HostTriangle& t : m.triangles) {
		t.v0 = (rot(t.v0) * scale) + loc; t.v1 = (rot(t.v1) * scale) + loc; t.v2 = (rot(t.v2) * scale) + loc;
		// the normal transform of the reference is rotation, then *(1,1,1) + (0,0,0)
		t.n0 = (rot(t.n0) * F3(1, 1, 1)) + F3(0, 0, 0); t.n1 = (rot(t.n1) * F3(1, 1, 1)) + F3(0, 0, 0); t.n2 = (rot(t.n2) * F3(1, 1, 1)) + F3(0, 0, 0);
	}
}

} // namespace rl
