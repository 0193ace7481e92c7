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

struct Corner { int v, vt, vn; };

bool ParseCorner(const std::string& s, int nV, int nVT, int nVN, Corner& c)
{
	c.v = c.vt = c.vn = -1;
	const char* p = s.c_str();
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

bool LoadOBJ(const char* path, OBJModel& out)
{
	if (path == nullptr) { Log("LoadOBJ: filepath was null"); return false; }
	std::vector<std::string> lines;
	if (!ReadLines(path, lines)) { Log("LoadOBJ: cannot open %s", path); return false; }
	const std::string dir = DirOf(path);

	std::vector<float> V, VT, VN;
	std::vector<MtlRecord> mtl;
	std::map<std::string, int> mtlIndex;
	struct Face { Corner c[3]; int material; int shape; };
	std::vector<Face> faces;
	int curMaterial = -1;
	int curShape = -1, numShapes = 0;
	bool shapeHasFaces = false;

	for (const std::string& line : lines) {
		Tokens tk(line.c_str());
		if (tk.size() == 0) continue;
		const std::string& k = tk[0];
		if (k == "v") { V.push_back(tk.f(1)); V.push_back(tk.f(2)); V.push_back(tk.f(3)); }
		else if (k == "vt") { VT.push_back(tk.f(1)); VT.push_back(tk.f(2)); }
		else if (k == "vn") { VN.push_back(tk.f(1)); VN.push_back(tk.f(2)); VN.push_back(tk.f(3)); }
		else if (k == "o" || k == "g") {
			// a new shape starts; shapes that end up without faces are dropped
			if (shapeHasFaces || curShape < 0) { curShape = numShapes++; }
			shapeHasFaces = false;
		}
		else if (k == "usemtl") {
			auto it = tk.size() > 1 ? mtlIndex.find(tk[1]) : mtlIndex.end();
			curMaterial = it == mtlIndex.end() ? -1 : it->second;
		}
		else if (k == "mtllib") {
			for (size_t i = 1; i < tk.size(); ++i) {
				size_t before = mtl.size();
				ParseMTL(dir + tk[i], mtl);
				for (size_t j = before; j < mtl.size(); ++j) mtlIndex[mtl[j].name] = (int)j;
			}
		}
		else if (k == "f") {
			if (curShape < 0) { curShape = numShapes++; }
			std::vector<Corner> cs;
			bool ok = true;
			for (size_t i = 1; i < tk.size(); ++i) {
				Corner c;
				if (!ParseCorner(tk[i], (int)V.size() / 3, (int)VT.size() / 2, (int)VN.size() / 3, c)) { ok = false; break; }
				cs.push_back(c);
			}
			if (!ok || cs.size() < 3) continue;
			for (size_t j = 1; j + 1 < cs.size(); ++j) {   // triangle fan
				Face f; f.c[0] = cs[0]; f.c[1] = cs[j]; f.c[2] = cs[j + 1]; f.material = curMaterial; f.shape = curShape;
				faces.push_back(f);
			}
			shapeHasFaces = true;
		}
	}
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
