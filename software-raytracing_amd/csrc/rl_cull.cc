// Host side of the "cells that cannot see the scene" rule (rl_runtime.inl EnqueueRender uses it per frame; RaylibAMD_CullCells exposes it to the tests).
#include "rl_host.h"

#include <algorithm>
#include <cmath>
#include <stdlib.h>

namespace rl {

// Which of a rank's cells can a camera ray meet the scene in?  With no sky panorama (a pinhole camera or a thin lens: round 4; a sky panorama adds each sample's own texel, looked up by k_resolve) every sample of every other cell ends in the
// miss shader with the same value (nothing, or the sun's illuminance when the sun is not hidden from the camera either), and the megakernel used to
// find that out sample by sample: generate the ray, test it against the root's boxes, store the constant -- 63 % of the Cornell frame's camera samples,
// 89 % of the 298 k-triangle frame's.  Here the scene's bounding box is projected onto the image plane once per frame (double precision, the eight
// corners, all of which must lie in front of the camera) and a cell is dropped from the job list when its pixels, the +-1 pixel of the jitter and a
// further 2 pixels of margin (five orders of magnitude more than the rounding of the device's ray set-up and of its widened box tests) stay
// outside that rectangle.  Dropped cells are flagged for k_resolve, which adds the constant up sample by sample as the stored samples would have been;
// the counters get the camera samples and root-box queries those samples stand for (FinishRender).  The frame cannot change: a listed or a
// dropped cell's pixels come to the same bits either way (tests/test_gpu_parity.py renders both).  RAYLIB_CULL_CELLS=0: every cell is listed.
// Returns false when the frame is not eligible.
bool CullCells(const CullScene& DS, const DCamera& cam, int32_t maxPathLength, float rayTMin, uint32_t W, uint32_t H,
               uint32_t cellsX, uint32_t cellFirst, uint32_t stride, uint32_t numLocalCells, CullResult& out)
{
	if (const char* e = getenv("RAYLIB_CULL_CELLS")) if (atoi(e) == 0) return false;
	// (a sky panorama does not stand in the way: the dropped cells' samples then differ by their sky texel, and k_resolve looks it up per sample -- rl_render.hip)
	if (DS.prims || !DS.boundsValid || maxPathLength <= 0 || numLocalCells == 0) return false;
	const double lensR = std::fabs((double)cam.lensRadius);   // the console front-end renders with aperture 0.01 (reference src/main.cc:24,421-425)
	if (!std::isfinite(lensR)) return false;
	// A negative rayTMin lets a query find hits BEHIND its origin (the tree walk and the tests support it): a camera ray that points away from the box, or a sun
	// ray from a camera downstream of the box, may then meet the scene after all.  Everything below reasons about t >= 0 only, so such a frame is not culled.
	if (!(rayTMin >= 0.0f)) return false;
	auto sub3 = [](const double* a, const double* b, double* r) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; };
	auto dot3 = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
	const double O[3] = { cam.origin[0], cam.origin[1], cam.origin[2] }, TL[3] = { cam.top_left[0], cam.top_left[1], cam.top_left[2] };
	const double Hh[3] = { cam.horizontal[0], cam.horizontal[1], cam.horizontal[2] }, Vv[3] = { cam.vertical[0], cam.vertical[1], cam.vertical[2] };
	for (int k = 0; k < 3; ++k) if (!std::isfinite(O[k]) || !std::isfinite(TL[k]) || !std::isfinite(Hh[k]) || !std::isfinite(Vv[k])) return false;
	// direction of sample (u, v): top_left + u H + (1 - v) V - origin = E + u H - v V with E = top_left + V - origin (rl_render.hip CameraRay)
	double E[3]; sub3(TL, O, E); for (int k = 0; k < 3; ++k) E[k] += Vv[k];
	double N[3] = { Hh[1] * Vv[2] - Hh[2] * Vv[1], Hh[2] * Vv[0] - Hh[0] * Vv[2], Hh[0] * Vv[1] - Hh[1] * Vv[0] };
	const double hh = dot3(Hh, Hh), vv = dot3(Vv, Vv), nn = dot3(N, N);
	if (!(hh > 0.0) || !(vv > 0.0) || !(nn > 0.0) || std::fabs(dot3(Hh, Vv)) > 1e-6 * std::sqrt(hh * vv)) return false;   // (the reference's camera basis is orthogonal)
	double planeDist = dot3(E, N);                        // every point of the image plane has this component along N
	if (planeDist < 0.0) { planeDist = -planeDist; for (int k = 0; k < 3; ++k) N[k] = -N[k]; }
	if (!(planeDist > 1e-12 * std::sqrt(nn) * std::sqrt(dot3(E, E)))) return false;
	const double ext = std::max({ DS.boundsMax[0] - DS.boundsMin[0], DS.boundsMax[1] - DS.boundsMin[1], DS.boundsMax[2] - DS.boundsMin[2], 1e-30 });
	double uLo = 1e300, uHi = -1e300, vLo = 1e300, vHi = -1e300;
	// A thin lens (reference render/camera.h:44-53: origin + u * rd.x + v * rd.y with |rd| <= lensRadius, aimed at the sample's point F on the focal plane, which is
	// the plane top_left / horizontal / vertical span): the ray from lens point L through F meets a scene point X of depth z where F = X * (zf / z) + L * (1 - zf / z),
	// i.e. at the pinhole's projection of X moved by (L - O) * (1 - zf / z) within the focal plane -- at most lensRadius * |1 - zf / z|, largest at the box's nearest
	// or farthest depth.  The rectangle of the pinhole projection grows by that circle of confusion.
	double lensFac = 0.0;
	for (int c = 0; c < 8; ++c) {
		double X[3], Q[3];
		for (int k = 0; k < 3; ++k) X[k] = (c >> k & 1) ? DS.boundsMax[k] + 1e-6 * ext : DS.boundsMin[k] - 1e-6 * ext;
		sub3(X, O, Q);
		const double depth = dot3(Q, N);
		if (!(depth > 1e-9 * std::sqrt(nn) * (std::sqrt(dot3(Q, Q)) + ext))) return false;   // a corner beside or behind the camera: no rectangle bounds the box
		const double sc = planeDist / depth;
		lensFac = std::max(lensFac, std::fabs(1.0 - sc));
		double R3[3]; for (int k = 0; k < 3; ++k) R3[k] = Q[k] * sc - E[k];             // on the image plane, relative to the direction of (u, v) = (0, 0)
		const double u = dot3(R3, Hh) / hh, v = -dot3(R3, Vv) / vv;
		if (!std::isfinite(u) || !std::isfinite(v)) return false;
		uLo = std::min(uLo, u); uHi = std::max(uHi, u); vLo = std::min(vLo, v); vHi = std::max(vHi, v);
	}
	// in pixels: a sample of pixel x has u * W in (x - 1, x + 1)
	const double margin = 2.0;
	const double cocU = lensR * lensFac / std::sqrt(hh) * 1.0001, cocV = lensR * lensFac / std::sqrt(vv) * 1.0001;   // the circle of confusion in u and in v
	if (!std::isfinite(cocU) || !std::isfinite(cocV)) return false;
	const double xLo = (uLo - cocU) * W - margin, xHi = (uHi + cocU) * W + margin, yLo = (vLo - cocV) * H - margin, yHi = (vHi + cocV) * H + margin;
	out.raysPerSample = 1;
	out.L[0] = out.L[1] = out.L[2] = 0.0f;
	if (DS.hasSun) {
		// the miss shader asks whether the sun is hidden from the ray's ORIGIN -- the same point for every sample of a pinhole camera.  Culling needs the
		// answer to be "no" without a traversal: the sun ray must miss the scene's box, enlarged by a percent, altogether
		const double D[3] = { -(double)DS.sunDirection[0], -(double)DS.sunDirection[1], -(double)DS.sunDirection[2] };
		double t0 = 0.0, t1 = 1e300;
		bool miss = false;
		for (int k = 0; k < 3 && !miss; ++k) {
			const double lo = DS.boundsMin[k] - 0.01 * ext - lensR, hi = DS.boundsMax[k] + 0.01 * ext + lensR;   // (every lens point is within lensR of the camera origin)
			if (D[k] == 0.0) { if (O[k] < lo || O[k] > hi) miss = true; continue; }
			double a = (lo - O[k]) / D[k], b = (hi - O[k]) / D[k]; if (a > b) std::swap(a, b);
			t0 = std::max(t0, a); t1 = std::min(t1, b);
			if (t0 > t1) miss = true;
		}
		if (!miss || !std::isfinite(D[0]) || !std::isfinite(D[1]) || !std::isfinite(D[2])) return false;
		out.raysPerSample = 2;
		// radiance = (0 + sunIlluminance), rl_render.hip MissShader
		for (int k = 0; k < 3; ++k) out.L[k] = 0.0f + DS.sunIlluminance[k];
	}
	const uint32_t numCells = cellsX * ((H + 7) / 8);
	out.active.clear(); out.active.reserve(numLocalCells);
	out.empty.assign(numLocalCells, 0);
	out.emptyPixels = 0;
	for (uint32_t k = 0; k < numLocalCells; ++k) {
		const uint32_t cell = cellFirst + k * stride;
		if (cell >= numCells) { out.active.push_back(k); continue; }
		const uint32_t cx = cell % cellsX, cy = cell / cellsX;
		const double px0 = 8.0 * cx - 1.0, px1 = 8.0 * cx + 8.0, py0 = 8.0 * cy - 1.0, py1 = 8.0 * cy + 8.0;
		const bool outside = px1 < xLo || px0 > xHi || py1 < yLo || py0 > yHi;
		if (outside) {
			out.empty[k] = 1;
			out.emptyPixels += (uint64_t)std::min(8u, W - cx * 8) * std::min(8u, H - cy * 8);
		} else out.active.push_back(k);
	}
	return out.active.size() < numLocalCells;   // nothing to drop: the plain list
}


} // namespace rl
