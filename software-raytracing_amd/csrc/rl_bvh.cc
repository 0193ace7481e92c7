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
#include <atomic>
#include <chrono>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

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
	// soft: a leaf of the binary builder (<= kMaxLeaf triangles, first / count) that carries a split of its triangles below it for the 8-wide plan to open where a
	// node has slots to spare (SplitLeaves); dead: below a soft node the plan left closed (not part of any tree)
	uint8_t soft = 0, dead = 0;
};

constexpr int kBins = 16;
constexpr uint32_t kMaxLeaf = 4;
constexpr float kCostTraverse = 1.0f, kCostTri = 1.0f;

// Work shared by the build threads.  The tree is the one a single thread would build, whatever the thread count:
// bin boxes and counts are min/max/integer sums (order-free), the partition is the serial std::partition on the same
// input order, sub-trees below a size threshold are built as independent tasks on disjoint ranges of `order`, and the
// final node array is re-emitted in the serial build's pre-order.
struct Shared {
	std::vector<uint8_t> kind;      // per primitive: PRIM_TRIANGLE / PRIM_SPHERE / PRIM_CUBE
	std::vector<Box> triBox;
	std::vector<f3> centroid;
	std::vector<uint32_t> order;
	unsigned threads = 1;
	uint32_t taskSize = 0;          // ranges of at most this many primitives become tasks (0: no tasks)
};

struct Task { uint32_t b, e, depth; };

struct Ctx {
	std::vector<TmpNode> tmp;
	uint32_t maxDepth = 0;
	std::vector<Task>* tasks = nullptr;   // only the top-level context spawns tasks
};

struct BinSet {
	Box box[3][kBins]; uint32_t count[3][kBins];
	void reset() { for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { box[a][k].reset(); count[a][k] = 0; } }
	void merge(const BinSet& o) { for (int a = 0; a < 3; ++a) for (int k = 0; k < kBins; ++k) { box[a][k].grow(o.box[a][k]); count[a][k] += o.count[a][k]; } }
};

constexpr size_t kLeafListMax = 4 * RL_LEAFLIST_RECORDS;   // leaves of the leaf list (rl_device.h)
constexpr uint32_t kMedianSplitDepth = 36;      // see build(): SAH splits above, median splits from here on -> depth <= 36 + 25 < 64
constexpr uint32_t kParallelRange = 1u << 17;   // ranges at least this long are scanned by all threads

template <typename F>
void ParallelChunks(unsigned threads, uint32_t b, uint32_t e, F&& fn)
{
	const uint32_t n = e - b;
	if (threads <= 1 || n < kParallelRange) { fn(b, e, 0u); return; }
	std::vector<std::thread> pool;
	const uint32_t per = (n + threads - 1) / threads;
	for (unsigned t = 1; t < threads; ++t) {
		const uint32_t cb = b + std::min(n, t * per), ce = b + std::min(n, (t + 1) * per);
		if (cb < ce) pool.emplace_back([&fn, cb, ce, t]() { fn(cb, ce, t); });
	}
	fn(b, b + std::min(n, per), 0u);
	for (std::thread& th : pool) th.join();
}

// a leaf is either <= kMaxLeaf triangles or exactly one analytic primitive
bool leafAllowed(const Shared& S, uint32_t b, uint32_t e)
{
	if (e - b == 1) return true;
	if (e - b > kMaxLeaf) return false;
	for (uint32_t i = b; i < e; ++i) if (S.kind[S.order[i]] != PRIM_TRIANGLE) return false;
	return true;
}

int32_t build(Shared& S, Ctx& C, uint32_t b, uint32_t e, uint32_t depth)
{
	const uint32_t n = e - b;
	const int32_t self = (int32_t)C.tmp.size();
	C.tmp.push_back(TmpNode());
	if (C.tasks && n <= S.taskSize && n > 1) {
		// built later by a worker; the placeholder keeps this node's place in the pre-order
		C.tmp[self].left = -2 - (int32_t)C.tasks->size();
		C.tasks->push_back({ b, e, depth });
		return self;
	}
	const unsigned threads = C.tasks ? S.threads : 1u;

	// long ranges of the top levels are scanned by all threads (per-thread partial results, merged in thread order:
	// min / max / integer sums, so the merge order does not matter)
	const bool wide = threads > 1 && n >= kParallelRange;

	// bounds of the boxes and of the centroids
	Box nodeBox, cb;
	if (!wide) {
		nodeBox.reset(); cb.reset();
		for (uint32_t i = b; i < e; ++i) { nodeBox.grow(S.triBox[S.order[i]]); cb.grow(S.centroid[S.order[i]]); }
	} else {
		std::vector<Box> nb(threads), cbs(threads);
		for (unsigned t = 0; t < threads; ++t) { nb[t].reset(); cbs[t].reset(); }
		ParallelChunks(threads, b, e, [&](uint32_t lo, uint32_t hi, unsigned t) {
			Box x, y; x.reset(); y.reset();
			for (uint32_t i = lo; i < hi; ++i) { x.grow(S.triBox[S.order[i]]); y.grow(S.centroid[S.order[i]]); }
			nb[t] = x; cbs[t] = y;
		});
		nodeBox = nb[0]; cb = cbs[0];
		for (unsigned t = 1; t < threads; ++t) { nodeBox.grow(nb[t]); cb.grow(cbs[t]); }
	}
	C.tmp[self].box = nodeBox;

	auto makeLeaf = [&]() {
		C.tmp[self].first = b; C.tmp[self].count = n;
		if (depth > C.maxDepth) C.maxDepth = depth;
		return self;
	};
	if (n <= 1) return makeLeaf();

	// binned SAH over the three axes, one pass over the range
	float lo[3], scale[3]; bool live[3];
	for (int a = 0; a < 3; ++a) {
		lo[a] = axisOf(cb.mn, a); const float hi = axisOf(cb.mx, a);
		live[a] = hi > lo[a];
		scale[a] = live[a] ? (float)kBins / (hi - lo[a]) : 0.0f;
	}
	BinSet bins; bins.reset();
	auto scan = [&](BinSet& bs, uint32_t cbeg, uint32_t cend) {
		for (uint32_t i = cbeg; i < cend; ++i) {
			const uint32_t p = S.order[i];
			const f3& c = S.centroid[p]; const Box& bx = S.triBox[p];
			for (int a = 0; a < 3; ++a) {
				if (!live[a]) continue;
				int k = (int)((axisOf(c, a) - lo[a]) * scale[a]);
				k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
				bs.box[a][k].grow(bx); bs.count[a][k]++;
			}
		}
	};
	if (!wide) scan(bins, b, e);
	else {
		std::vector<BinSet> sets(threads);
		for (BinSet& bs : sets) bs.reset();
		ParallelChunks(threads, b, e, [&](uint32_t cbeg, uint32_t cend, unsigned t) { scan(sets[t], cbeg, cend); });
		for (unsigned t = 0; t < threads; ++t) bins.merge(sets[t]);
	}

	float bestCost = FLT_MAX; int bestAxis = -1; int bestBin = -1;
	const float parentArea = std::max(nodeBox.halfArea(), 1e-30f);
	for (int a = 0; a < 3; ++a) {
		if (!live[a]) continue;
		float rightArea[kBins]; uint32_t rightCount[kBins];
		Box acc; acc.reset(); uint32_t cnt = 0;
		for (int k = kBins - 1; k >= 1; --k) { acc.grow(bins.box[a][k]); cnt += bins.count[a][k]; rightArea[k] = acc.halfArea(); rightCount[k] = cnt; }
		acc.reset(); cnt = 0;
		for (int k = 0; k < kBins - 1; ++k) {
			acc.grow(bins.box[a][k]); cnt += bins.count[a][k];
			if (cnt == 0 || rightCount[k + 1] == 0) continue;
			float cost = kCostTraverse + kCostTri * (acc.halfArea() * cnt + rightArea[k + 1] * rightCount[k + 1]) / parentArea;
			if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
		}
	}

	uint32_t mid;
	if (depth >= kMedianSplitDepth) {
		// A SAH tree has no depth bound (a pathological input can peel one primitive per level) and the traversal stacks hold
		// 64 entries: from this depth on the range is halved by count along the longest centroid axis, which ends within
		// log2(n) <= 25 more levels (kMaxPrimitives), i.e. below 64.  Deterministic for any thread count (nth_element on a range
		// whose order is already fixed).
		if (leafAllowed(S, b, e)) return makeLeaf();
		int axis = 0;
		const float ex = cb.mx.x - cb.mn.x, ey = cb.mx.y - cb.mn.y, ez = cb.mx.z - cb.mn.z;
		if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
		mid = b + n / 2;
		std::nth_element(S.order.begin() + b, S.order.begin() + mid, S.order.begin() + e, [&](uint32_t x, uint32_t y) {
			const float cx = axisOf(S.centroid[x], axis), cy = axisOf(S.centroid[y], axis);
			return cx < cy || (cx == cy && x < y);
		});
	} else if (bestAxis < 0) {
		// all centroids coincide: split by index
		if (leafAllowed(S, b, e)) return makeLeaf();
		mid = b + n / 2;
	} else {
		if (leafAllowed(S, b, e) && bestCost >= kCostTri * n) return makeLeaf();
		const float l0 = lo[bestAxis], sc = scale[bestAxis];
		auto it = std::partition(S.order.begin() + b, S.order.begin() + e, [&](uint32_t t) {
			int k = (int)((axisOf(S.centroid[t], bestAxis) - l0) * sc);
			k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
			return k <= bestBin;
		});
		mid = (uint32_t)(it - S.order.begin());
		if (mid == b || mid == e) mid = b + n / 2;
	}
	const int32_t l = build(S, C, b, mid, depth + 1);
	const int32_t r = build(S, C, mid, e, depth + 1);
	C.tmp[self].left = l; C.tmp[self].right = r;
	return self;
}

// pre-order copy of (top tree + task sub-trees) into one array: the order a single-threaded build pushes nodes in
int32_t Splice(const std::vector<Ctx>& sub, const std::vector<std::pair<uint32_t, int32_t>>& where, const Ctx& src, int32_t idx, std::vector<TmpNode>& dst)
{
	const TmpNode& n = src.tmp[idx];
	if (n.left <= -2) { const auto& w = where[(size_t)(-2 - n.left)]; return Splice(sub, where, sub[w.first], w.second, dst); }
	const int32_t self = (int32_t)dst.size();
	dst.push_back(n);
	if (n.left >= 0) {
		const int32_t l = Splice(sub, where, src, n.left, dst);
		const int32_t r = Splice(sub, where, src, n.right, dst);
		dst[self].left = l; dst[self].right = r;
	}
	return self;
}

inline void storeBox(float* mn, float* mx, const Box& b) {
	mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z;
	mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z;
}
inline int32_t leafRef(uint32_t first, uint32_t kind, uint32_t count) { return ~(int32_t)((first << 6) | (kind << 4) | (count - 1)); }

} // namespace

// Leaf references keep the first primitive slot in 25 bits (DNode, rl_device.h): more primitives than that cannot be addressed.
bool BVHCapacityOk(size_t numPrimitives) { return numPrimitives < ((size_t)1 << 25); }

// nodes4 -> nodes4q (DNode4Q, rl_device.h).  Every decision is made in double, where origin + q * step is exact, so "the grid
// box contains the float box" holds exactly.
static void QuantizeWide(BVH& out)
{
	out.nodes4q.assign(out.nodes4.size(), DNode4Q());
	auto run = [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; ++i) {
		const DNode4& n = out.nodes4[i];
		DNode4Q q; memset(&q, 0, sizeof(q));
		float steps[3] = { 0.0f, 0.0f, 0.0f };
		for (int a = 0; a < 3; ++a) {
			float lo = FLT_MAX, hi = -FLT_MAX;
			for (int k = 0; k < 4; ++k) if (n.child[k] != DNODE_EMPTY) { lo = std::min(lo, n.lo[a][k]); hi = std::max(hi, n.hi[a][k]); }
			if (!(lo <= hi)) { lo = hi = 0.0f; }
			q.origin[a] = lo;
			// the smallest power of two whose 255 steps span the node
			const double extent = (double)hi - (double)lo;
			int e = 1;   // biased exponent; 1 = 2^-126, the smallest normal
			if (extent > 0.0) { int x; (void)frexp(extent / 255.0, &x); e = std::max(1, std::min(254, x + 127)); }   // 2^x >= extent / 255
			double step = ldexp(1.0, e - 127);
			while (255.0 * step < extent && e < 254) { ++e; step *= 2.0; }
			steps[a] = (float)step;   // a power of two between 2^-126 and 2^127: exact
			for (int k = 0; k < 4; ++k) {
				if (n.child[k] == DNODE_EMPTY) { q.qlo[a] |= 255u << (8 * k); continue; }   // inverted: lower 255, upper 0
				double l = floor(((double)n.lo[a][k] - (double)lo) / step), h = ceil(((double)n.hi[a][k] - (double)lo) / step);
				l = std::max(0.0, std::min(255.0, l)); h = std::max(0.0, std::min(255.0, h));
				while (l > 0.0 && (double)lo + l * step > (double)n.lo[a][k]) l -= 1.0;
				while (h < 255.0 && (double)lo + h * step < (double)n.hi[a][k]) h += 1.0;
				q.qlo[a] |= (uint32_t)l << (8 * k);
				q.qhi[a] |= (uint32_t)h << (8 * k);
			}
		}
		q.stepX = steps[0]; q.stepY = steps[1]; q.stepZ = steps[2];
		for (int k = 0; k < 4; ++k) q.child[k] = n.child[k];
		out.nodes4q[i] = q;
	} };
	const size_t n = out.nodes4.size();
	unsigned threads = n >= (1u << 16) ? std::max(1u, std::min(32u, std::thread::hardware_concurrency())) : 1u;
	if (const char* e = getenv("RAYLIB_BUILD_THREADS")) { int v = atoi(e); if (v > 0 && n >= (1u << 16)) threads = (unsigned)std::min(v, 32); }
	std::vector<std::thread> pool;
	const size_t per = (n + threads - 1) / threads;
	for (unsigned t = 1; t < threads; ++t) { const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per); if (a < b) pool.emplace_back(run, a, b); }
	run(0, std::min(n, per));
	for (std::thread& th : pool) th.join();
}

// ---- the 8-wide tree (DNode8, rl_device.h) ---------------------------------------------------------------------------------------------------------
// Collapse: starting from a node's two children, the inner child with the largest surface area is replaced by its own two children until there are eight (or
// only leaves) -- the BVH4's rule.  Slots: child c goes to the slot whose three bits (bit 0: +x, bit 1: +y, bit 2: +z) point from the node's centre towards
// the child's, assigned greedily by the largest dot product of (child centre - node centre) with the slot's (+-1, +-1, +-1): a ray then visits the slots in the
// order slot XOR (signs of its direction) without sorting anything.  Nodes are numbered breadth first, so the inner children of a node are consecutive; the
// leaf children of a node, taken in slot order, define the order of the triangle slots (BuildBVH assigns them in W.leafOrder's order).
struct Wide8Node { int32_t kid[8]; uint32_t firstChild; };   // TmpNode index per slot (-1: none); node index of the first inner child
struct Wide8 { std::vector<Wide8Node> nodes; std::vector<int32_t> leafOrder; uint32_t depth = 0; double sah = 0.0; };

// Which binary nodes become 8-wide nodes is decided for the whole tree at once (the dynamic programme of Ylitie, Karras, Laine, "Efficient incoherent ray
// traversal on GPUs through compressed wide BVHs", HPG 2017, section 3.1, without its leaf merging: the binary tree's leaves stay what they are), minimising the
// sum of the 8-wide nodes' surface areas -- the node steps a random ray is expected to take.  c[n][i - 1]: the least cost of the sub-tree under n when it may
// occupy i slots of its parent 8-wide node (i = 1: n is itself an 8-wide node, or a leaf).  Opening the largest child first (what the 4-wide tree does; here
// RAYLIB_WIDE_GREEDY=1) leaves 4.1 of the 8 slots used on the 298 k-triangle room -- 63 655 nodes, an expected 39.5 steps; the plan: 45 455 nodes, 5.4 slots used,
// 38.5 steps, and frames 0.3 ... 4.7 % shorter (tools/gpu_w8_plan_ab.py).  For the 4-wide tree the same plan buys 0.7 ... 2 % of expected steps and a deeper
// worst-case stack (colonnade 29 -> 33 entries: the next kernel instance): not used there.
template <int W> struct WidePlan {
	float c[W - 1];
	uint32_t bits;   // k (slots of the left sub-tree) when n's children share j = 2 ... W slots: 3 bits each from bit 0; bit 21 + i: c[i - 1] is c[i - 2] (i = 2 ... W - 1)
	int kAt(int j) const { return (int)((bits >> (3 * (j - 2))) & 7u); }
	bool sameAsFewer(int i) const { return ((bits >> (21 + i)) & 1u) != 0u; }
};
template <int W>
static void PlanWide(const std::vector<TmpNode>& T, int32_t root, std::vector<WidePlan<W>>& plan, float triCost)
{
	plan.assign(T.size(), WidePlan<W>());
	const double rootArea = std::max((double)T[root].box.halfArea(), 1e-30);
	for (size_t t = T.size(); t-- > 0;) {   // children follow their parent in T: a reverse sweep sees them first
		WidePlan<W>& P = plan[t];
		// a leaf child costs its expected triangle tests: (its box's area / the root's) x its triangles x triCost node steps
		const float leafCost = triCost * (float)((double)T[t].box.halfArea() / rootArea) * (float)T[t].count;
		if (T[t].left < 0) { for (int i = 0; i < W - 1; ++i) P.c[i] = leafCost; P.bits = 0; continue; }
		const WidePlan<W>& L = plan[T[t].left]; const WidePlan<W>& R = plan[T[t].right];
		float d[W + 1]; uint32_t bits = 0;   // d[j]: n's two children share j slots
		for (int j = 2; j <= W; ++j) {
			float best = FLT_MAX; int bestK = 1;
			for (int k = 1; k < j; ++k) { if (k > W - 1 || j - k > W - 1) continue; const float v = L.c[k - 1] + R.c[j - k - 1]; if (v < best) { best = v; bestK = k; } }
			d[j] = best; bits |= (uint32_t)bestK << (3 * (j - 2));
		}
		// one slot: an 8-wide node of its own -- or, for a split leaf (soft), the leaf as the binary builder made it
		P.c[0] = T[t].soft ? leafCost : (float)((double)T[t].box.halfArea() / rootArea) + d[W];
		for (int i = 2; i <= W - 1; ++i) {
			if (d[i] < P.c[i - 2]) P.c[i - 1] = d[i];
			else { P.c[i - 1] = P.c[i - 2]; bits |= 1u << (21 + i); }
		}
		P.bits = bits;
	}
}
// the planned children of the W-wide node rooted at t: its two sub-trees share W slots as the plan says; a sub-tree given i slots is either opened (its own two
// children share them) or, with one slot, a child of this node.  Returns their number.
template <int W>
static int PlannedChildren(std::vector<TmpNode>& T, const std::vector<WidePlan<W>>& plan, int32_t t, int32_t* kids)
{
	struct Share { int32_t t; int slots; } todo[2 * W]; int top = 0, nk = 0;
	{ const int k = plan[t].kAt(W); todo[top++] = { T[t].right, W - k }; todo[top++] = { T[t].left, k }; }
	while (top > 0) {
		const Share sh = todo[--top];
		if (T[sh.t].left < 0) { kids[nk++] = sh.t; continue; }
		int i = sh.slots;
		while (i > 1 && plan[sh.t].sameAsFewer(i)) --i;
		if (i == 1) { kids[nk++] = sh.t; continue; }
		const int k = plan[sh.t].kAt(i);
		T[sh.t].soft = 0;   // (a split leaf the plan opens is an inner node from here on: the binary and the 4-wide tree are emitted from the same T)
		todo[top++] = { T[sh.t].right, i - k }; todo[top++] = { T[sh.t].left, k };
	}
	return nk;
}

// Leaves of the binary tree hold up to kMaxLeaf triangles: below that the surface-area heuristic finds a node not worth its step.  In an 8-wide node a step tests
// eight boxes whether or not eight children exist (5.4 did on the 298 k-triangle room), so a spare slot is a free box: every triangle leaf gets a binary split
// of its triangles (sorted along the longest axis of their centroids; each range cut where the two halves' area x count is least) down to single triangles,
// flagged `soft`; the plan (PlanWide) opens such a node where its parent has slots left and the smaller boxes save expected triangle tests, and leaves it the
// leaf it was elsewhere (CollapseWide8 closes what the plan did not open).  No node is added to the 8-wide tree by this.
static void SplitLeaves(std::vector<TmpNode>& T, Shared& B)
{
	const size_t n0 = T.size();
	for (size_t t = 0; t < n0; ++t) {
		if (T[t].left >= 0 || T[t].count < 2 || B.kind[B.order[T[t].first]] != PRIM_TRIANGLE) continue;
		const uint32_t b = T[t].first, e = b + T[t].count;
		Box cb; cb.reset();
		for (uint32_t i = b; i < e; ++i) cb.grow(B.centroid[B.order[i]]);
		int axis = 0;
		const float ex = cb.mx.x - cb.mn.x, ey = cb.mx.y - cb.mn.y, ez = cb.mx.z - cb.mn.z;
		if (ey > ex && ey >= ez) axis = 1; else if (ez > ex && ez > ey) axis = 2;
		std::sort(B.order.begin() + b, B.order.begin() + e, [&](uint32_t x, uint32_t y) {
			const float cx = axisOf(B.centroid[x], axis), cy = axisOf(B.centroid[y], axis);
			return cx < cy || (cx == cy && x < y);
		});
		// ranges to split, depth first; a range's node exists before its children are appended (children follow their parent in T)
		struct Range { int32_t node; uint32_t b, e; } todo[2 * kMaxLeaf]; int top = 0;
		todo[top++] = { (int32_t)t, b, e };
		while (top > 0) {
			const Range r = todo[--top];
			if (r.e - r.b < 2) continue;
			uint32_t bestCut = r.b + 1; float bestCost = FLT_MAX;
			for (uint32_t cut = r.b + 1; cut < r.e; ++cut) {
				Box l, rr; l.reset(); rr.reset();
				for (uint32_t i = r.b; i < cut; ++i) l.grow(B.triBox[B.order[i]]);
				for (uint32_t i = cut; i < r.e; ++i) rr.grow(B.triBox[B.order[i]]);
				const float cost = l.halfArea() * (float)(cut - r.b) + rr.halfArea() * (float)(r.e - cut);
				if (cost < bestCost) { bestCost = cost; bestCut = cut; }
			}
			TmpNode kids[2];
			const uint32_t lim[3] = { r.b, bestCut, r.e };
			for (int k = 0; k < 2; ++k) {
				kids[k].box.reset();
				for (uint32_t i = lim[k]; i < lim[k + 1]; ++i) kids[k].box.grow(B.triBox[B.order[i]]);
				kids[k].first = lim[k]; kids[k].count = lim[k + 1] - lim[k];
			}
			const int32_t li = (int32_t)T.size(); T.push_back(kids[0]);
			const int32_t ri = (int32_t)T.size(); T.push_back(kids[1]);
			T[r.node].left = li; T[r.node].right = ri; T[r.node].soft = 1;
			todo[top++] = { ri, bestCut, r.e }; todo[top++] = { li, r.b, bestCut };
		}
	}
}

static void CollapseWide8(std::vector<TmpNode>& T, int32_t root, Wide8& W, bool greedy, float triCost)
{
	std::vector<WidePlan<8>> plan;
	if (!greedy) PlanWide<8>(T, root, plan, triCost);
	struct Item { int32_t tmp; uint32_t level; };
	std::vector<Item> queue;
	queue.push_back({ root, 1u });
	W.nodes.clear(); W.leafOrder.clear(); W.depth = 0; W.sah = 0.0;
	const double rootArea = std::max((double)T[root].box.halfArea(), 1e-30);
	for (size_t at = 0; at < queue.size(); ++at) {
		const Item it = queue[at];
		W.depth = std::max(W.depth, it.level);
		W.sah += (double)T[it.tmp].box.halfArea() / rootArea;
		int32_t kids[8]; int nk = 0;
		if (!plan.empty()) nk = PlannedChildren<8>(T, plan, it.tmp, kids);
		else {
		kids[nk++] = T[it.tmp].left; kids[nk++] = T[it.tmp].right;
		while (nk < 8) {
			int best = -1; float bestArea = -1.0f;
			for (int k = 0; k < nk; ++k) if (T[kids[k]].left >= 0) { const float a = T[kids[k]].box.halfArea(); if (a > bestArea) { bestArea = a; best = k; } }
			if (best < 0) break;
			const int32_t open = kids[best];
			kids[best] = T[open].left; kids[nk++] = T[open].right;
		}
		}
		const Box& nb = T[it.tmp].box;
		const double cx = 0.5 * ((double)nb.mn.x + nb.mx.x), cy = 0.5 * ((double)nb.mn.y + nb.mx.y), cz = 0.5 * ((double)nb.mn.z + nb.mx.z);
		double cost[8][8];
		for (int c = 0; c < nk; ++c) {
			const Box& b = T[kids[c]].box;
			const double dx = 0.5 * ((double)b.mn.x + b.mx.x) - cx, dy = 0.5 * ((double)b.mn.y + b.mx.y) - cy, dz = 0.5 * ((double)b.mn.z + b.mx.z) - cz;
			for (int sl = 0; sl < 8; ++sl) cost[c][sl] = ((sl & 1) ? dx : -dx) + ((sl & 2) ? dy : -dy) + ((sl & 4) ? dz : -dz);
		}
		Wide8Node nd; for (int sl = 0; sl < 8; ++sl) nd.kid[sl] = -1;
		bool childDone[8] = { false, false, false, false, false, false, false, false };
		for (int round = 0; round < nk; ++round) {
			int bc = -1, bs = -1; double bv = -1e300;
			for (int c = 0; c < nk; ++c) { if (childDone[c]) continue; for (int sl = 0; sl < 8; ++sl) { if (nd.kid[sl] >= 0) continue; if (cost[c][sl] > bv) { bv = cost[c][sl]; bc = c; bs = sl; } } }
			nd.kid[bs] = kids[bc]; childDone[bc] = true;
		}
		nd.firstChild = (uint32_t)queue.size();
		for (int sl = 0; sl < 8; ++sl) {
			const int32_t k = nd.kid[sl];
			if (k < 0) continue;
			if (T[k].left >= 0 && !T[k].soft) queue.push_back({ k, it.level + 1u }); else W.leafOrder.push_back(k);
		}
		W.nodes.push_back(nd);
	}
	// split leaves the plan left closed are leaves again; what hangs below them belongs to no tree
	for (size_t t = 0; t < T.size(); ++t) {
		if (!T[t].soft || T[t].dead) continue;
		int32_t st[4 * kMaxLeaf]; int top = 0;
		st[top++] = T[t].left; st[top++] = T[t].right;
		while (top > 0) { const int32_t k = st[--top]; T[k].dead = 1; if (T[k].left >= 0) { st[top++] = T[k].left; st[top++] = T[k].right; } }
		T[t].left = T[t].right = -1; T[t].soft = 0;
	}
}

// the expected node steps of a random ray through the 4-wide collapse of T (sum of the wide nodes' areas over the root's), without building it: what decides
// whether a scene's rays walk the 8-wide tree by default (rl_runtime.inl RL_BVH8_MIN_STEPS) -- and only then are its leaves split for that tree
static double ExpectedSteps4(const std::vector<TmpNode>& T, int32_t root)
{
	const double rootArea = std::max((double)T[root].box.halfArea(), 1e-30);
	double sum = 0.0;
	std::vector<int32_t> work; work.push_back(root);
	while (!work.empty()) {
		const int32_t t = work.back(); work.pop_back();
		sum += (double)T[t].box.halfArea() / rootArea;
		int32_t kids[4]; int nk = 0;
		kids[nk++] = T[t].left; kids[nk++] = T[t].right;
		while (nk < 4) {
			int best = -1; float bestArea = -1.0f;
			for (int k = 0; k < nk; ++k) if (T[kids[k]].left >= 0) { const float a = T[kids[k]].box.halfArea(); if (a > bestArea) { bestArea = a; best = k; } }
			if (best < 0) break;
			const int32_t open = kids[best];
			kids[best] = T[open].left; kids[nk++] = T[open].right;
		}
		for (int k = 0; k < nk; ++k) if (T[kids[k]].left >= 0) work.push_back(kids[k]);
	}
	return sum;
}

// W + the leaf references (first triangle slot, count) -> out.nodes8.  Grid boxes as in QuantizeWide: every decision in double.
static void EmitWide8(const std::vector<TmpNode>& T, const Wide8& W, const std::vector<int32_t>& leafCode, BVH& out)
{
	out.nodes8.assign(W.nodes.size(), DNode8());
	out.depth8 = W.depth; out.sahNodes8 = (float)W.sah;
	auto run = [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; ++i) {
		const Wide8Node& w = W.nodes[i];
		DNode8 n; memset(&n, 0, sizeof(n));
		uint32_t imask = 0, leafMask = 0, triBase = 0; bool haveTri = false;
		for (int c = 0; c < 8; ++c) {
			const int32_t k = w.kid[c];
			if (k < 0) continue;
			if (T[k].left >= 0) { imask |= 1u << c; continue; }
			const uint32_t code = (uint32_t)~leafCode[k], first = code >> 6, count = (code & 7u) + 1;
			leafMask |= ((1u << count) - 1u) << (4 * c);
			if (!haveTri) { triBase = first; haveTri = true; }
		}
		uint32_t exps[3] = { 0, 0, 0 };
		for (int a = 0; a < 3; ++a) {
			float lo = FLT_MAX, hi = -FLT_MAX;
			for (int c = 0; c < 8; ++c) if (w.kid[c] >= 0) { lo = std::min(lo, axisOf(T[w.kid[c]].box.mn, a)); hi = std::max(hi, axisOf(T[w.kid[c]].box.mx, a)); }
			if (!(lo <= hi)) { lo = hi = 0.0f; }
			n.origin[a] = lo;
			const double extent = (double)hi - (double)lo;
			int e = 1;
			if (extent > 0.0) { int x; (void)frexp(extent / 255.0, &x); e = std::max(1, std::min(254, x + 127)); }
			double step = ldexp(1.0, e - 127);
			while (255.0 * step < extent && e < 254) { ++e; step *= 2.0; }
			exps[a] = (uint32_t)e;
			for (int c = 0; c < 8; ++c) {
				if (w.kid[c] < 0) { n.qlo[a][c >> 2] |= 255u << (8 * (c & 3)); continue; }   // inverted: lower 255, upper 0
				const double blo = axisOf(T[w.kid[c]].box.mn, a), bhi = axisOf(T[w.kid[c]].box.mx, a);
				double l = floor((blo - (double)lo) / step), h = ceil((bhi - (double)lo) / step);
				l = std::max(0.0, std::min(255.0, l)); h = std::max(0.0, std::min(255.0, h));
				while (l > 0.0 && (double)lo + l * step > blo) l -= 1.0;
				while (h < 255.0 && (double)lo + h * step < bhi) h += 1.0;
				n.qlo[a][c >> 2] |= (uint32_t)l << (8 * (c & 3));
				n.qhi[a][c >> 2] |= (uint32_t)h << (8 * (c & 3));
			}
		}
		n.meta = exps[0] | (exps[1] << 8) | (exps[2] << 16) | (imask << 24);
		n.childBase = w.firstChild; n.triBase = triBase; n.leafMask = leafMask; n.alphaMask = 0;
		out.nodes8[i] = n;
	} };
	const size_t n = W.nodes.size();
	unsigned threads = n >= (1u << 15) ? std::max(1u, std::min(32u, std::thread::hardware_concurrency())) : 1u;
	if (const char* e = getenv("RAYLIB_BUILD_THREADS")) { int v = atoi(e); if (v > 0 && n >= (1u << 15)) threads = (unsigned)std::min(v, 32); }
	std::vector<std::thread> pool;
	const size_t per = (n + threads - 1) / threads;
	for (unsigned t = 1; t < threads; ++t) { const size_t a = std::min(n, t * per), b = std::min(n, (t + 1) * per); if (a < b) pool.emplace_back(run, a, b); }
	run(0, std::min(n, per));
	for (std::thread& th : pool) th.join();
}

void BuildBVH(const std::vector<PrimRef>& prims, BVH& out, const BVHBuildOptions& opt)
{
	out.nodes.clear(); out.nodes4.clear(); out.nodes4q.clear(); out.nodes8.clear(); out.depth8 = 0; out.leafList.clear(); out.stackNeed4 = 0; out.triOrder.clear(); out.depth = 0; out.sahCost = 0.0f;
	const uint32_t n = (uint32_t)prims.size();
	Box empty; empty.mn = F3(FLT_MAX, FLT_MAX, FLT_MAX); empty.mx = F3(-FLT_MAX, -FLT_MAX, -FLT_MAX);

	if (n == 0) {
		DNode root; memset(&root, 0, sizeof(root));
		storeBox(root.lmin, root.lmax, empty); storeBox(root.rmin, root.rmax, empty);
		root.left = root.right = DNODE_EMPTY;
		out.nodes.push_back(root);
		return;
	}

	Shared B;
	B.kind.resize(n); B.triBox.resize(n); B.centroid.resize(n); B.order.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		Box b; b.mn = prims[i].mn; b.mx = prims[i].mx;
		B.kind[i] = prims[i].kind;
		B.triBox[i] = b;
		B.centroid[i] = F3(0.5f * (b.mn.x + b.mx.x), 0.5f * (b.mn.y + b.mx.y), 0.5f * (b.mn.z + b.mx.z));
		B.order[i] = i;
	}
	// threads: RAYLIB_BUILD_THREADS, else the host's (capped at 32); small scenes build on the calling thread
	unsigned threads = std::thread::hardware_concurrency();
	if (const char* e = getenv("RAYLIB_BUILD_THREADS")) { int v = atoi(e); if (v > 0) threads = (unsigned)v; }
	threads = std::max(1u, std::min(32u, threads));
	if (n < (1u << 16)) threads = 1;
	B.threads = threads;
	// sub-trees of at most this many primitives are tasks.  Not far below kParallelRange: between the two sizes a range is
	// binned by ONE thread while the others wait (10 M triangles, 32 threads: 1.3 s of a 2.3 s build went there with n / 16T)
	B.taskSize = threads > 1 ? std::max<uint32_t>(4096u, std::min<uint32_t>(kParallelRange - 1u, n / (threads * 4u))) : 0u;
	// large scenes: everything below the all-thread ranges is a task -- no range is left for one thread to bin while the others wait
	if (threads > 1 && n >= 8u * kParallelRange) B.taskSize = kParallelRange - 1u;

	const auto tb0 = std::chrono::steady_clock::now();
	Ctx top;
	std::vector<Task> tasks;
	if (threads > 1) top.tasks = &tasks;
	top.tmp.reserve(threads > 1 ? 4096 : 2 * (size_t)n);
	const int32_t topRoot = build(B, top, 0, n, 0);
	const auto tb1 = std::chrono::steady_clock::now();
	// every worker appends the sub-trees of the tasks it takes to its own arena (one growing vector per thread: a
	// vector per task meant one mmap/munmap pair per task, and the threads queued on the kernel's mm lock)
	std::vector<Ctx> sub(threads);
	std::vector<std::pair<uint32_t, int32_t>> where(tasks.size());   // task -> (arena, root index in it)
	if (!tasks.empty()) {
		std::atomic<size_t> nextTask{ 0 };
		auto worker = [&](unsigned me) {
			sub[me].tmp.reserve((size_t)n / threads);
			for (;;) {
				const size_t t = nextTask.fetch_add(1);
				if (t >= tasks.size()) return;
				const int32_t r = build(B, sub[me], tasks[t].b, tasks[t].e, tasks[t].depth);
				where[t] = { me, r };
			}
		};
		std::vector<std::thread> pool;
		for (unsigned t = 1; t < threads; ++t) pool.emplace_back(worker, t);
		worker(0u);
		for (std::thread& th : pool) th.join();
	}
	const auto tb2 = std::chrono::steady_clock::now();
	uint32_t maxDepth = top.maxDepth;
	for (const Ctx& c : sub) maxDepth = std::max(maxDepth, c.maxDepth);
	std::vector<TmpNode> merged;
	int32_t root = topRoot;
	if (!tasks.empty()) {
		merged.reserve(2 * (size_t)n);
		root = Splice(sub, where, top, topRoot, merged);
		top.tmp.clear(); top.tmp.shrink_to_fit();
		for (Ctx& c : sub) { c.tmp.clear(); c.tmp.shrink_to_fit(); }
	} else {
		merged.swap(top.tmp);
	}

	// Leaf references.  Triangle leaves index the triangle arrays in leaf order (out.triOrder lists the
	// original triangle index of every slot); an analytic primitive's leaf carries its index in its own array.
	std::vector<TmpNode>& T = merged;
	// Triangle-only scenes of at least 8 triangles get the wide trees; the 8-wide one decides the order of the triangle slots (the leaf children of one of its
	// nodes hold consecutive slots), every other format refers to the same slots through its leaf references.
	bool trianglesOnly = true;
	for (uint32_t i = 0; i < n && trianglesOnly; ++i) if (prims[i].kind != PRIM_TRIANGLE) trianglesOnly = false;
	const bool wideTrees = n >= 8 && trianglesOnly && T[root].left >= 0;
	// (not for the scenes small enough for the leaf list: its leaves are sub-trees of this tree, whose triangles must stay one range of slots -- the depth-first order)
	const bool wide8 = wideTrees && n > RL_LEAFLIST_MAXTRIS;
	Wide8 W8;
	if (wide8) {
		// leaves split for the 8-wide plan (SplitLeaves) where that tree is the one the scene's rays will walk
		if (!opt.wideGreedy && opt.splitLeaves8 && ExpectedSteps4(T, root) >= opt.minSteps8) SplitLeaves(T, B);
		CollapseWide8(T, root, W8, opt.wideGreedy, opt.triCost8);
	}
	std::vector<int32_t> leafCode(T.size(), 0);
	auto codeLeaf = [&](size_t t) {
		const uint32_t k = B.kind[B.order[T[t].first]];
		if (k == PRIM_TRIANGLE) {
			const uint32_t first = (uint32_t)out.triOrder.size();
			for (uint32_t i = 0; i < T[t].count; ++i) out.triOrder.push_back(prims[B.order[T[t].first + i]].index);
			leafCode[t] = leafRef(first, PRIM_TRIANGLE, T[t].count);
		} else {
			leafCode[t] = leafRef(prims[B.order[T[t].first]].index, k, 1);
		}
	};
	if (wide8) { for (int32_t t : W8.leafOrder) codeLeaf((size_t)t); }
	else for (size_t t = 0; t < T.size(); ++t) if (T[t].left < 0) codeLeaf(t);

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
		// (the depth is taken here, from the tree that is emitted: split leaves the 8-wide plan opened are inner nodes of it)
		std::vector<std::pair<int32_t, uint32_t>> st; st.push_back({ root, 0u });
		maxDepth = 0;
		while (!st.empty()) {
			const int32_t t = st.back().first; const uint32_t d = st.back().second; st.pop_back();
			if (T[t].left < 0) { maxDepth = std::max(maxDepth, d); continue; }
			emitIndex[t] = next++;
			st.push_back({ T[t].right, d + 1u }); st.push_back({ T[t].left, d + 1u });
		}
	}
	out.nodes.resize((size_t)next);
	double sah = 0.0; const float rootArea = std::max(T[root].box.halfArea(), 1e-30f);
	for (size_t t = 0; t < T.size(); ++t) {
		if (T[t].dead) continue;
		if (T[t].left < 0) { sah += kCostTri * T[t].count * T[t].box.halfArea() / rootArea; continue; }
		sah += kCostTraverse * T[t].box.halfArea() / rootArea;
		DNode nd; memset(&nd, 0, sizeof(nd));
		const TmpNode& L = T[T[t].left]; const TmpNode& R = T[T[t].right];
		storeBox(nd.lmin, nd.lmax, L.box); storeBox(nd.rmin, nd.rmax, R.box);
		nd.left = (L.left < 0) ? leafCode[T[t].left] : emitIndex[T[t].left];
		nd.right = (R.left < 0) ? leafCode[T[t].right] : emitIndex[T[t].right];
		out.nodes[emitIndex[t]] = nd;
	}
	if (getenv("RAYLIB_BUILD_TIMING")) Log("BVH build: %u threads, %zu tasks; top levels %.3f s, tasks %.3f s, merge + emit %.3f s", threads, tasks.size(),
		std::chrono::duration<double>(tb1 - tb0).count(), std::chrono::duration<double>(tb2 - tb1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tb2).count());
	out.depth = maxDepth;     // leaves at depth d => at most d inner nodes above them
	out.sahCost = (float)sah;

	// ---- the wide tree: collapse to <= 4 children per node ----
	// Starting from a node's two children, the inner child with the largest surface area is replaced by its own two
	// children until there are four (or only leaves).  Only for scenes the pool schedule can run (triangles only, not tiny).
	if (wideTrees) {
		if (wide8) EmitWide8(T, W8, leafCode, out);
		struct Item { int32_t tmp; int32_t slot; uint32_t need; };   // a BVH2 inner node that becomes wide node `slot`
		std::vector<Item> work;
		out.nodes4.clear();
		out.nodes4.reserve(T.size() / 3 + 1);
		out.nodes4.emplace_back();
		work.push_back({ root, 0, 0u });
		uint32_t needMax = 0;
		double sah4 = 0.0; const double rootArea4 = std::max((double)T[root].box.halfArea(), 1e-30);
		while (!work.empty()) {
			const Item it = work.back(); work.pop_back();
			sah4 += (double)T[it.tmp].box.halfArea() / rootArea4;
			int32_t kids[4]; int nk = 0;
			kids[nk++] = T[it.tmp].left; kids[nk++] = T[it.tmp].right;
			while (nk < 4) {
				int best = -1; float bestArea = -1.0f;
				for (int k = 0; k < nk; ++k) if (T[kids[k]].left >= 0) { const float a = T[kids[k]].box.halfArea(); if (a > bestArea) { bestArea = a; best = k; } }
				if (best < 0) break;
				const int32_t open = kids[best];
				kids[best] = T[open].left; kids[nk++] = T[open].right;
			}
			DNode4 nd; memset(&nd, 0, sizeof(nd));
			const uint32_t need = it.need + (uint32_t)(nk - 1);   // entries this node can leave on the stack while one child is followed
			if (need > needMax) needMax = need;
			for (int k = 0; k < 4; ++k) {
				if (k >= nk) {
					nd.lo[0][k] = nd.lo[1][k] = nd.lo[2][k] = FLT_MAX; nd.hi[0][k] = nd.hi[1][k] = nd.hi[2][k] = -FLT_MAX;
					nd.child[k] = DNODE_EMPTY;
					continue;
				}
				const TmpNode& c = T[kids[k]];
				nd.lo[0][k] = c.box.mn.x; nd.lo[1][k] = c.box.mn.y; nd.lo[2][k] = c.box.mn.z;
				nd.hi[0][k] = c.box.mx.x; nd.hi[1][k] = c.box.mx.y; nd.hi[2][k] = c.box.mx.z;
				if (c.left < 0) nd.child[k] = leafCode[kids[k]];
				else {
					const int32_t slot = (int32_t)out.nodes4.size();
					out.nodes4.emplace_back();
					nd.child[k] = slot;
					work.push_back({ kids[k], slot, need });
				}
			}
			out.nodes4[it.slot] = nd;
		}
		out.stackNeed4 = needMax; out.sahNodes4 = (float)sah4;
		QuantizeWide(out);
		// ---- the leaf list: a scene that 4 * RL_LEAFLIST_RECORDS leaves of <= 8 triangles can hold is walked without a tree ----
		// Every ray tests every leaf's box once (4 records of 4 boxes, in lockstep across a wave: no stack, no divergence), then visits the
		// leaves it touched nearest first.  In a tree this small a wave's rays take different turns at every node, and the wave pays for
		// the union of their walks (Cornell frame: 10 node steps per wave and bounce for 3.5 per ray).  The leaves are a cut through the
		// SAH tree: starting from the root, the sub-tree with the largest area x triangle count is opened until the list is full or only the tree's own leaves are left.
		// Up to 4.5 triangles per leaf on average: beyond, the leaves of the cut grow towards 8 triangles and the tree wins again (tools/gpu_leaflist.py,
		// leaf list / BVH4 walk, first version: 36 triangles 0.86, 72: 0.92, 84: 0.88, 96: 0.89, 108: 0.90, 120 (8-triangle leaves): 1.05; final version: 0.79, 0.80, 0.79,
		// 0.77, 0.78 -- the limit is also what the kernel's LDS layout holds, rl_device.h RL_LEAFLIST_MAXTRIS).
		if (n <= RL_LEAFLIST_MAXTRIS) {
			std::vector<uint32_t> triFirst(T.size(), 0), triCount(T.size(), 0);
			for (size_t t = T.size(); t-- > 0;) {   // children follow their parent in T (pre-order): a reverse sweep sees them first
				if (T[t].left < 0) { triFirst[t] = ((uint32_t)~leafCode[t]) >> 6; triCount[t] = T[t].count; }
				else { triFirst[t] = triFirst[T[t].left]; triCount[t] = triCount[T[t].left] + triCount[T[t].right]; }
			}
			std::vector<int32_t> cut; cut.push_back(root);
			for (;;) {
				int best = -1; double bestCost = -1.0; bool bestOver = false;
				for (size_t k = 0; k < cut.size(); ++k) {
					const int32_t t = cut[k];
					if (T[t].left < 0) continue;
					const bool over = triCount[t] > 8;   // too large for one leaf: goes first
					const double cost = ((double)T[t].box.halfArea() + 1e-30) * triCount[t];
					if (best < 0 || (over && !bestOver) || (over == bestOver && cost > bestCost)) { bestCost = cost; best = (int)k; bestOver = over; }
				}
				if (best < 0 || (cut.size() >= kLeafListMax && !bestOver)) break;
				if (cut.size() >= kLeafListMax) { cut.clear(); break; }   // does not fit
				const int32_t open = cut[(size_t)best];
				cut[(size_t)best] = T[open].left; cut.push_back(T[open].right);
			}
			bool fits = !cut.empty();
			for (int32_t t : cut) if (triCount[t] > 8) fits = false;
			if (fits) {
				out.leafList.assign((cut.size() + 3) / 4, DNode4());
				for (DNode4& nd : out.leafList) {
					memset(&nd, 0, sizeof(nd));
					// an unused slot is the box [+inf, -inf]: whatever the ray, one of its axes enters it at +inf (rl_render.hip TraverseLeafList has no other test for it)
					for (int k = 0; k < 4; ++k) { nd.lo[0][k] = nd.lo[1][k] = nd.lo[2][k] = INFINITY; nd.hi[0][k] = nd.hi[1][k] = nd.hi[2][k] = -INFINITY; nd.child[k] = DNODE_EMPTY; }
				}
				for (size_t at = 0; at < cut.size(); ++at) {
					const int32_t t = cut[at];
					DNode4& nd = out.leafList[at / 4]; const int k = (int)(at % 4);
					nd.lo[0][k] = T[t].box.mn.x; nd.lo[1][k] = T[t].box.mn.y; nd.lo[2][k] = T[t].box.mn.z;
					nd.hi[0][k] = T[t].box.mx.x; nd.hi[1][k] = T[t].box.mx.y; nd.hi[2][k] = T[t].box.mx.z;
					nd.child[k] = leafRef(triFirst[t], PRIM_TRIANGLE, triCount[t]);
				}
			}
		}
	} else { out.nodes4.clear(); out.nodes4q.clear(); out.nodes8.clear(); out.depth8 = 0; out.leafList.clear(); out.stackNeed4 = 0; }
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

// The wide tree must hold every triangle exactly once, inside the box its parent records for it, every child box inside
// its parent's, and never need more stack than stackNeed4 says.
bool ValidateBVH4(const BVH& bvh, const std::vector<HostTriangle>& tris)
{
	if (bvh.nodes4.empty()) return false;
	std::vector<uint8_t> seen(tris.size(), 0);
	struct Item { int32_t ref; f3 mn, mx; uint32_t need; bool haveBox; };
	std::vector<Item> st;
	st.push_back({ 0, F3(0, 0, 0), F3(0, 0, 0), 0u, false });
	auto inside = [](const f3& p, const f3& mn, const f3& mx) {
		return p.x >= mn.x && p.y >= mn.y && p.z >= mn.z && p.x <= mx.x && p.y <= mx.y && p.z <= mx.z;
	};
	while (!st.empty()) {
		const Item it = st.back(); st.pop_back();
		if (it.ref < 0) {
			const uint32_t code = (uint32_t)~it.ref, first = code >> 6, count = (code & 7u) + 1;
			if (((code >> 4) & 3u) != PRIM_TRIANGLE) return false;
			for (uint32_t k = 0; k < count; ++k) {
				if (first + k >= bvh.triOrder.size()) return false;
				const uint32_t ti = bvh.triOrder[first + k];
				if (ti >= tris.size() || seen[ti]) return false;
				seen[ti] = 1;
				const HostTriangle& t = tris[ti];
				if (!inside(t.v0, it.mn, it.mx) || !inside(t.v1, it.mn, it.mx) || !inside(t.v2, it.mn, it.mx)) return false;
			}
			continue;
		}
		if ((size_t)it.ref >= bvh.nodes4.size()) return false;
		const DNode4& n = bvh.nodes4[it.ref];
		{   // the grid node: same children, every grid box around its float box, 255 steps at most
			if (bvh.nodes4q.size() != bvh.nodes4.size()) return false;
			const DNode4Q& q = bvh.nodes4q[it.ref];
			for (int k = 0; k < 4; ++k) {
				if (q.child[k] != n.child[k]) return false;
				if (n.child[k] == DNODE_EMPTY) continue;
				for (int a = 0; a < 3; ++a) {
					const double step = (double)(a == 0 ? q.stepX : (a == 1 ? q.stepY : q.stepZ));
					{ int e2 = 0; if (!(step > 0.0) || frexp(step, &e2) != 0.5) return false; }   // a power of two
					const double lo = (double)q.origin[a] + (double)((q.qlo[a] >> (8 * k)) & 255u) * step, hi = (double)q.origin[a] + (double)((q.qhi[a] >> (8 * k)) & 255u) * step;
					if (!(lo <= (double)n.lo[a][k] && hi >= (double)n.hi[a][k])) return false;
				}
			}
		}
		int nk = 0;
		for (int k = 0; k < 4; ++k) if (n.child[k] != DNODE_EMPTY) ++nk;
		if (nk < 2 && !(it.ref == 0 && nk >= 1)) return false;
		const uint32_t need = it.need + (uint32_t)(nk - 1);
		if (need > bvh.stackNeed4) return false;
		for (int k = 0; k < 4; ++k) {
			if (n.child[k] == DNODE_EMPTY) continue;
			const f3 mn = F3(n.lo[0][k], n.lo[1][k], n.lo[2][k]), mx = F3(n.hi[0][k], n.hi[1][k], n.hi[2][k]);
			if (it.haveBox && (!inside(mn, it.mn, it.mx) || !inside(mx, it.mn, it.mx))) return false;
			st.push_back({ n.child[k], mn, mx, need, true });
		}
	}
	for (uint8_t v : seen) if (!v) return false;
	// the leaf list, if the scene has one: every triangle exactly once, inside its leaf's box; unused slots are the box [+inf, -inf]
	if (!bvh.leafList.empty()) {
		if (bvh.leafList.size() > RL_LEAFLIST_RECORDS) return false;
		std::fill(seen.begin(), seen.end(), 0);
		for (const DNode4& n : bvh.leafList) for (int k = 0; k < 4; ++k) {
			if (n.child[k] == DNODE_EMPTY) {
				for (int a = 0; a < 3; ++a) if (!(n.lo[a][k] == INFINITY && n.hi[a][k] == -INFINITY)) return false;
				continue;
			}
			if (n.child[k] >= 0) return false;
			const uint32_t code = (uint32_t)~n.child[k], first = code >> 6, count = (code & 7u) + 1;
			if (((code >> 4) & 3u) != PRIM_TRIANGLE) return false;
			const f3 mn = F3(n.lo[0][k], n.lo[1][k], n.lo[2][k]), mx = F3(n.hi[0][k], n.hi[1][k], n.hi[2][k]);
			for (uint32_t i = 0; i < count; ++i) {
				if (first + i >= bvh.triOrder.size()) return false;
				const uint32_t ti = bvh.triOrder[first + i];
				if (ti >= tris.size() || seen[ti]) return false;
				seen[ti] = 1;
				const HostTriangle& t = tris[ti];
				if (!inside(t.v0, mn, mx) || !inside(t.v1, mn, mx) || !inside(t.v2, mn, mx)) return false;
			}
		}
		for (uint8_t v : seen) if (!v) return false;
	}
	return true;
}

// The 8-wide tree must hold every triangle slot exactly once, and every child's grid box must contain every triangle below it.
bool ValidateBVH8(const BVH& bvh, const std::vector<HostTriangle>& tris)
{
	if (bvh.nodes8.empty()) return true;
	std::vector<uint8_t> seen(bvh.triOrder.size(), 0);
	struct Frame { uint32_t node; int child; double lo[3], hi[3]; double clo[3], chi[3]; };   // lo / hi: bounds of what has been seen below this node so far; clo / chi: ... below the child being walked
	// iterative post-order: for every node, the bounds of the triangles below each child are checked against that child's grid box
	std::vector<Frame> st;
	auto gridBox = [&](const DNode8& n, int c, double* lo, double* hi) {
		for (int a = 0; a < 3; ++a) {
			const double step = ldexp(1.0, (int)((n.meta >> (8 * a)) & 255u) - 127);
			lo[a] = (double)n.origin[a] + (double)((n.qlo[a][c >> 2] >> (8 * (c & 3))) & 255u) * step;
			hi[a] = (double)n.origin[a] + (double)((n.qhi[a][c >> 2] >> (8 * (c & 3))) & 255u) * step;
		}
	};
	Frame f0; f0.node = 0; f0.child = -1; for (int a = 0; a < 3; ++a) { f0.lo[a] = 1e300; f0.hi[a] = -1e300; }
	st.push_back(f0);
	uint32_t depth = 0;
	while (!st.empty()) {
		Frame& F = st.back();
		depth = std::max<uint32_t>(depth, (uint32_t)st.size());
		if (F.node >= bvh.nodes8.size()) return false;
		const DNode8& n = bvh.nodes8[F.node];
		const uint32_t imask = n.meta >> 24;
		if (++F.child >= 8) {
			// done: hand this node's bounds to the parent's current child
			const Frame done = F; st.pop_back();
			if (!st.empty()) {
				Frame& P = st.back();
				const DNode8& pn = bvh.nodes8[P.node];
				double glo[3], ghi[3]; gridBox(pn, P.child, glo, ghi);
				for (int a = 0; a < 3; ++a) {
					if (!(glo[a] <= done.lo[a] && ghi[a] >= done.hi[a])) return false;
					P.lo[a] = std::min(P.lo[a], done.lo[a]); P.hi[a] = std::max(P.hi[a], done.hi[a]);
				}
			}
			continue;
		}
		const int c = F.child;
		const uint32_t nib = (n.leafMask >> (4 * c)) & 15u;
		if (imask & (1u << c)) {
			if (nib) return false;
			Frame ch; ch.node = n.childBase + (uint32_t)__builtin_popcount(imask & ((1u << c) - 1u)); ch.child = -1;
			for (int a = 0; a < 3; ++a) { ch.lo[a] = 1e300; ch.hi[a] = -1e300; }
			st.push_back(ch);
			continue;
		}
		if (!nib) {   // unused: the inverted box
			for (int a = 0; a < 3; ++a) if (((n.qlo[a][c >> 2] >> (8 * (c & 3))) & 255u) != 255u || ((n.qhi[a][c >> 2] >> (8 * (c & 3))) & 255u) != 0u) return false;
			continue;
		}
		if (nib != 1u && nib != 3u && nib != 7u && nib != 15u) return false;
		const uint32_t first = n.triBase + (uint32_t)__builtin_popcount(n.leafMask & ((1u << (4 * c)) - 1u)), count = (uint32_t)__builtin_popcount(nib);
		double glo[3], ghi[3]; gridBox(n, c, glo, ghi);
		for (uint32_t k = 0; k < count; ++k) {
			if (first + k >= bvh.triOrder.size() || seen[first + k]) return false;
			seen[first + k] = 1;
			const uint32_t ti = bvh.triOrder[first + k];
			if (ti >= tris.size()) return false;
			const HostTriangle& t = tris[ti];
			const f3 vs[3] = { t.v0, t.v1, t.v2 };
			for (const f3& v : vs) {
				const double p[3] = { v.x, v.y, v.z };
				for (int a = 0; a < 3; ++a) {
					if (!(glo[a] <= p[a] && ghi[a] >= p[a])) return false;
					F.lo[a] = std::min(F.lo[a], p[a]); F.hi[a] = std::max(F.hi[a], p[a]);
				}
			}
		}
	}
	for (uint8_t v : seen) if (!v) return false;
	return depth <= bvh.depth8;
}

// The 8-wide walk of rl_render.hip (NodeStep8 / LeafStep8) on the host, operation by operation in float -- A = step * inv and B = (corner - o) * inv per axis, the
// rounding bound E = (|B| + 255 |A|) 2^-21, one fma per 8-bit plane, the NEGATED entry distance, the sign of fma(exit, widen, -entry), groups of hit children in
// visiting order, the stack of groups -- with the exit distance fixed at tMax[i] and every triangle of every leaf child reached tested by a tolerant
// double-precision test: outT[i] = the least distance among them (FLT_MAX: none).  What the box arithmetic must never do is skip the leaf that holds the closest
// hit; tests compare outT with the oracle's closest hit on the same rays without a GPU (tests/test_host_logic.py).  outSteps (optional): node steps taken.
bool Walk8Host(const BVH& bvh, const std::vector<HostTriangle>& tris, const float* rays, int n, float tMin, const float* tMax, float* outT, uint32_t* outSteps)
{
	if (bvh.nodes8.empty()) return false;
	const float widen = 1.00001f;
	auto clampInv = [](float x) { return std::isinf(x) ? copysignf(1e30f, x) : x; };
	auto plane = [](const uint32_t q[2], int c) { return (float)((q[c >> 2] >> (8 * (c & 3))) & 255u); };
	for (int r = 0; r < n; ++r) {
		const float* ray = rays + 6 * (size_t)r;
		const float o[3] = { ray[0], ray[1], ray[2] }, d[3] = { ray[3], ray[4], ray[5] };
		float inv[3]; bool neg[3]; uint32_t oct = 0;
		for (int a = 0; a < 3; ++a) { inv[a] = clampInv(1.0f / d[a]); neg[a] = inv[a] < 0.0f; if (!neg[a]) oct |= 1u << a; }
		const float ntMin = -tMin, tmx = std::min(tMax[r], FLT_MAX);
		std::vector<std::pair<uint32_t, uint32_t>> stack;
		uint32_t gx = 0, gy = (1u << (24 + oct)) | 1u, steps = 0;
		double best = (double)FLT_MAX;
		for (;;) {
			if ((gy >> 24) == 0u) { if (stack.empty()) break; gx = stack.back().first; gy = stack.back().second; stack.pop_back(); continue; }
			const uint32_t pos = 31u - (uint32_t)__builtin_clz(gy);
			gy &= ~(1u << pos);
			const uint32_t slot = (pos - 24u) ^ oct;
			const uint32_t node = gx + (uint32_t)__builtin_popcount(gy & 0xffu & ((1u << slot) - 1u));
			if ((gy >> 24) != 0u) stack.push_back({ gx, gy });
			if (node >= bvh.nodes8.size() || stack.size() > 64) return false;
			const DNode8& nd = bvh.nodes8[node];
			++steps;
			float A[3], nB[3], Bf[3];
			for (int a = 0; a < 3; ++a) {
				uint32_t sb = ((nd.meta >> (8 * a)) & 255u) << 23; float st; memcpy(&st, &sb, 4);
				A[a] = st * inv[a];
				const float B = (nd.origin[a] - o[a]) * inv[a];
				const float E = fabsf(A[a] * 1.21593475e-4f) + fabsf(B * 4.76837158e-7f);
				nB[a] = E - B; Bf[a] = B + E;
			}
			uint32_t hit = 0;
			const uint32_t imask = nd.meta >> 24;
			for (int ch = 0; ch < 8; ++ch) {
				float ntn = ntMin, tf = tmx;
				for (int a = 0; a < 3; ++a) {
					const float qn = plane(neg[a] ? nd.qhi[a] : nd.qlo[a], ch), qf = plane(neg[a] ? nd.qlo[a] : nd.qhi[a], ch);
					ntn = std::min(ntn, fmaf(qn, -A[a], nB[a])); tf = std::min(tf, fmaf(qf, A[a], Bf[a]));
				}
				if (!std::signbit(fmaf(tf, widen, ntn))) hit |= 1u << ch;
			}
			uint32_t innerP = 0;
			for (uint32_t b = 0; b < 8u; ++b) if (((hit & imask) >> b) & 1u) innerP |= 1u << (b ^ oct);
			for (uint32_t ch = 0; ch < 8u; ++ch) {
				if (!((hit & ~imask) >> ch & 1u)) continue;
				const uint32_t nib = (nd.leafMask >> (4 * ch)) & 15u;
				const uint32_t first = nd.triBase + (uint32_t)__builtin_popcount(nd.leafMask & ((1u << (4 * ch)) - 1u));
				for (uint32_t k = 0; k < (uint32_t)__builtin_popcount(nib); ++k) {
					if (first + k >= bvh.triOrder.size()) return false;
					const HostTriangle& t = tris[bvh.triOrder[first + k]];
					// tolerant double-precision test (edges and vertices count as inside by 1e-6): a superset of what the reference's float test accepts
					const double e1[3] = { (double)t.v1.x - t.v0.x, (double)t.v1.y - t.v0.y, (double)t.v1.z - t.v0.z }, e2[3] = { (double)t.v2.x - t.v0.x, (double)t.v2.y - t.v0.y, (double)t.v2.z - t.v0.z };
					const double dd[3] = { d[0], d[1], d[2] };
					const double pv[3] = { dd[1] * e2[2] - dd[2] * e2[1], dd[2] * e2[0] - dd[0] * e2[2], dd[0] * e2[1] - dd[1] * e2[0] };
					const double det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
					if (det == 0.0) continue;
					const double tv[3] = { (double)o[0] - t.v0.x, (double)o[1] - t.v0.y, (double)o[2] - t.v0.z };
					const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
					const double qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
					const double v = (dd[0] * qv[0] + dd[1] * qv[1] + dd[2] * qv[2]) / det;
					const double tt = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) / det;
					if (u >= -1e-6 && v >= -1e-6 && u + v <= 1.0 + 1e-6 && tt >= (double)tMin * 0.999 && tt < best) best = tt;
				}
			}
			gx = nd.childBase; gy = (innerP << 24) | imask;
		}
		outT[r] = best >= (double)FLT_MAX ? FLT_MAX : (float)best;
		if (outSteps) outSteps[r] = steps;
	}
	return true;
}

} // namespace rl
