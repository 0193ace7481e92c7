// Host BVH builder: binned-SAH BVH2 over triangle bounds, emitted as 64-byte
// two-box nodes (rl_device.h DNode).
//
// The reference builds three nested levels of random-axis median-split trees with
// a std::sort per node (reference geom/bvh.cc:10-80, geom/static_mesh.cc:80-95,
// loader/obj_loader.cc:236-245, geom/scene.cc:23-31).  Tree topology is not part
// of results parity -- closest hit is the minimum t over all triangles whichever
// way they are grouped -- so this builder is free to produce
// ONE flat tree whose shape suits a GPU: SAH splits (fewer nodes touched per
// ray), <= 4 triangles per leaf stored contiguously, both child boxes in the
// parent so a lane decides two children per 64-byte fetch.
#include "rl_host.h"

#include <algorithm>
#include <float.h>
#include <string.h>

namespace rl {
namespace {

struct Box {
	f3 mn, mx;
	void reset() { mn = F3(FLT_MAX, FLT_MAX, FLT_MAX); mx = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX); }
	void grow(const f3& p) { mn = fmin3(mn, p); mx = fmax3(mx, p); }
	void grow(const Box& b) { mn = fmin3(mn, b.mn); mx = fmax3(mx, b.mx); }
	float halfArea() const {
		f3 d = mx - mn;
		if (d.x < 0 || d.y < 0 || d.z < 0) return 0.0f;
		return d.x * d.y + d.y * d.z + d.z * d.x;
	}
};
inline float axisOf(const f3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct TmpNode {
	Box box;
	int32_t left = -1, right = -1;   // TmpNode indices; -1 => leaf
	uint32_t first = 0, count = 0;
};

constexpr int kBins = 16;
constexpr uint32_t kMaxLeaf = 4;
constexpr float kCostTraverse = 1.0f, kCostTri = 1.0f;

struct Builder {
	std::vector<uint8_t> kind;      // per primitive: PRIM_TRIANGLE / PRIM_SPHERE / PRIM_CUBE
	std::vector<Box> triBox;
	std::vector<f3> centroid;
	std::vector<uint32_t> order;
	std::vector<TmpNode> tmp;
	uint32_t maxDepth = 0;

	// a leaf is either <= kMaxLeaf triangles or exactly one analytic primitive
	bool leafAllowed(uint32_t b, uint32_t e) const {
		if (e - b == 1) return true;
		if (e - b > kMaxLeaf) return false;
		for (uint32_t i = b; i < e; ++i) if (kind[order[i]] != PRIM_TRIANGLE) return false;
		return true;
	}

	int32_t build(uint32_t b, uint32_t e, uint32_t depth) {
		TmpNode node;
		node.box.reset();
		Box cb; cb.reset();
		for (uint32_t i = b; i < e; ++i) { node.box.grow(triBox[order[i]]); cb.grow(centroid[order[i]]); }
		const uint32_t n = e - b;
		int32_t self = (int32_t)tmp.size();
		tmp.push_back(node);

		auto makeLeaf = [&]() {
			tmp[self].first = b; tmp[self].count = n;
			if (depth > maxDepth) maxDepth = depth;
			return self;
		};
		if (n <= 1) return makeLeaf();

		// best binned split over the three axes
		float bestCost = FLT_MAX; int bestAxis = -1; int bestBin = -1;
		const float parentArea = std::max(node.box.halfArea(), 1e-30f);
		for (int a = 0; a < 3; ++a) {
			float lo = axisOf(cb.mn, a), hi = axisOf(cb.mx, a);
			if (!(hi > lo)) continue;
			Box binBox[kBins]; uint32_t binCount[kBins];
			for (int k = 0; k < kBins; ++k) { binBox[k].reset(); binCount[k] = 0; }
			const float scale = (float)kBins / (hi - lo);
			for (uint32_t i = b; i < e; ++i) {
				int k = (int)((axisOf(centroid[order[i]], a) - lo) * scale);
				k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				binBox[k].grow(triBox[order[i]]); binCount[k]++;
			}
			float rightArea[kBins]; uint32_t rightCount[kBins];
			Box acc; acc.reset(); uint32_t cnt = 0;
			for (int k = kBins - 1; k >= 1; --k) { acc.grow(binBox[k]); cnt += binCount[k]; rightArea[k] = acc.halfArea(); rightCount[k] = cnt; }
			acc.reset(); cnt = 0;
			for (int k = 0; k < kBins - 1; ++k) {
				acc.grow(binBox[k]); cnt += binCount[k];
				if (cnt == 0 || rightCount[k + 1] == 0) continue;
				float cost = kCostTraverse + kCostTri * (acc.halfArea() * cnt + rightArea[k + 1] * rightCount[k + 1]) / parentArea;
				if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
			}
		}

		uint32_t mid;
		if (bestAxis < 0) {
			// all centroids coincide: split by index
			if (leafAllowed(b, e)) return makeLeaf();
			mid = b + n / 2;
		} else {
			if (leafAllowed(b, e) && bestCost >= kCostTri * n) return makeLeaf();
			float lo = axisOf(cb.mn, bestAxis), hi = axisOf(cb.mx, bestAxis);
			const float scale = (float)kBins / (hi - lo);
			auto it = std::partition(order.begin() + b, order.begin() + e, [&](uint32_t t) {
				int k = (int)((axisOf(centroid[t], bestAxis) - lo) * scale);
				k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				return k <= bestBin;
			});
			mid = (uint32_t)(it - order.begin());
			if (mid == b || mid == e) mid = b + n / 2;
		}
		int32_t l = build(b, mid, depth + 1);
		int32_t r = build(mid, e, depth + 1);
		tmp[self].left = l; tmp[self].right = r;
		return self;
	}
};

inline void storeBox(float* mn, float* mx, const Box& b) {
	mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z;
	mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z;
}
inline int32_t leafRef(uint32_t first, uint32_t kind, uint32_t count) { return ~(int32_t)((first << 6) | (kind << 4) | (count - 1)); }

} // namespace

void BuildBVH(const std::vector<PrimRef>& prims, BVH& out)
{
	out.nodes.clear(); out.triOrder.clear(); out.depth = 0; out.sahCost = 0.0f;
	const uint32_t n = (uint32_t)prims.size();
	Box empty; empty.mn = F3(FLT_MAX, FLT_MAX, FLT_MAX); empty.mx = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);

	if (n == 0) {
		DNode root; memset(&root, 0, sizeof(root));
		storeBox(root.lmin, root.lmax, empty); storeBox(root.rmin, root.rmax, empty);
		root.left = root.right = DNODE_EMPTY;
		out.nodes.push_back(root);
		return;
	}

	Builder B;
	B.kind.resize(n); B.triBox.resize(n); B.centroid.resize(n); B.order.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		Box b; b.mn = prims[i].mn; b.mx = prims[i].mx;
		B.kind[i] = prims[i].kind;
		B.triBox[i] = b;
		B.centroid[i] = F3(0.5f * (b.mn.x + b.mx.x), 0.5f * (b.mn.y + b.mx.y), 0.5f * (b.mn.z + b.mx.z));
		B.order[i] = i;
	}
	B.tmp.reserve(2 * (size_t)n);
	int32_t root = B.build(0, n, 0);

	// Leaf references.  Triangle leaves index the triangle arrays in leaf order (out.triOrder lists the
	// original triangle index of every slot); an analytic primitive's leaf carries its index in its own array.
	const std::vector<TmpNode>& T = B.tmp;
	std::vector<int32_t> leafCode(T.size(), 0);
	for (size_t t = 0; t < T.size(); ++t) {
		if (T[t].left >= 0) continue;
		const uint32_t k = B.kind[B.order[T[t].first]];
		if (k == PRIM_TRIANGLE) {
			const uint32_t first = (uint32_t)out.triOrder.size();
			for (uint32_t i = 0; i < T[t].count; ++i) out.triOrder.push_back(prims[B.order[T[t].first + i]].index);
			leafCode[t] = leafRef(first, PRIM_TRIANGLE, T[t].count);
		} else {
			leafCode[t] = leafRef(prims[B.order[T[t].first]].index, k, 1);
		}
	}

	// Emit two-box nodes in depth-first order.  A tree that is a single leaf still
	// gets one inner node (left = the leaf, right = empty).
	if (T[root].left < 0) {
		DNode nd; memset(&nd, 0, sizeof(nd));
		storeBox(nd.lmin, nd.lmax, T[root].box); storeBox(nd.rmin, nd.rmax, empty);
		nd.left = leafCode[root]; nd.right = DNODE_EMPTY;
		out.nodes.push_back(nd);
		out.depth = 1;
		return;
	}
	std::vector<int32_t> emitIndex(T.size(), -1);
	int32_t next = 0;
	{
		std::vector<int32_t> st; st.push_back(root);
		while (!st.empty()) {
			int32_t t = st.back(); st.pop_back();
			if (T[t].left < 0) continue;
			emitIndex[t] = next++;
			st.push_back(T[t].right); st.push_back(T[t].left);
		}
	}
	out.nodes.resize((size_t)next);
	double sah = 0.0; const float rootArea = std::max(T[root].box.halfArea(), 1e-30f);
	for (size_t t = 0; t < T.size(); ++t) {
		if (T[t].left < 0) { sah += kCostTri * T[t].count * T[t].box.halfArea() / rootArea; continue; }
		sah += kCostTraverse * T[t].box.halfArea() / rootArea;
		DNode nd; memset(&nd, 0, sizeof(nd));
		const TmpNode& L = T[T[t].left]; const TmpNode& R = T[T[t].right];
		storeBox(nd.lmin, nd.lmax, L.box); storeBox(nd.rmin, nd.rmax, R.box);
		nd.left = (L.left < 0) ? leafCode[T[t].left] : emitIndex[T[t].left];
		nd.right = (R.left < 0) ? leafCode[T[t].right] : emitIndex[T[t].right];
		out.nodes[emitIndex[t]] = nd;
	}
	out.depth = B.maxDepth;   // leaves at depth d => at most d inner nodes above them
	out.sahCost = (float)sah;
}

bool ValidateBVH(const BVH& bvh, const std::vector<HostTriangle>& tris)
{
	if (bvh.nodes.empty()) return false;
	std::vector<uint8_t> seen(tris.size(), 0);
	struct Item { int32_t ref; f3 mn, mx; };
	std::vector<Item> st;
	const DNode& r = bvh.nodes[0];
	st.push_back({ r.left, F3(r.lmin[0], r.lmin[1], r.lmin[2]), F3(r.lmax[0], r.lmax[1], r.lmax[2]) });
	st.push_back({ r.right, F3(r.rmin[0], r.rmin[1], r.rmin[2]), F3(r.rmax[0], r.rmax[1], r.rmax[2]) });
	auto inside = [](const f3& p, const f3& mn, const f3& mx) {
		return p.x >= mn.x && p.y >= mn.y && p.z >= mn.z && p.x <= mx.x && p.y <= mx.y && p.z <= mx.z;
	};
	while (!st.empty()) {
		Item it = st.back(); st.pop_back();
		if (it.ref == DNODE_EMPTY) continue;
		if (it.ref < 0) {
			uint32_t code = (uint32_t)~it.ref, first = code >> 6, count = (code & 7u) + 1;
			if (((code >> 4) & 3u) != PRIM_TRIANGLE) continue;   // analytic primitive: nothing to check against triangles
			for (uint32_t k = 0; k < count; ++k) {
				if (first + k >= bvh.triOrder.size()) return false;
				uint32_t ti = bvh.triOrder[first + k];
				if (ti >= tris.size() || seen[ti]) return false;
				seen[ti] = 1;
				const HostTriangle& t = tris[ti];
				if (!inside(t.v0, it.mn, it.mx) || !inside(t.v1, it.mn, it.mx) || !inside(t.v2, it.mn, it.mx)) return false;
			}
			continue;
		}
		if ((size_t)it.ref >= bvh.nodes.size()) return false;
		const DNode& n = bvh.nodes[it.ref];
		f3 lmn = F3(n.lmin[0], n.lmin[1], n.lmin[2]), lmx = F3(n.lmax[0], n.lmax[1], n.lmax[2]);
		f3 rmn = F3(n.rmin[0], n.rmin[1], n.rmin[2]), rmx = F3(n.rmax[0], n.rmax[1], n.rmax[2]);
		if (n.left != DNODE_EMPTY && (!inside(lmn, it.mn, it.mx) || !inside(lmx, it.mn, it.mx))) return false;
		if (n.right != DNODE_EMPTY && (!inside(rmn, it.mn, it.mx) || !inside(rmx, it.mn, it.mx))) return false;
		st.push_back({ n.left, lmn, lmx });
		st.push_back({ n.right, rmn, rmx });
	}
	for (uint8_t s : seen) if (!s) return false;
	return true;
}

} // namespace rl
