// Host runtime of the MI355X raylib: devices, streams, per-rank work buffers, scene upload, launches, and the frame
// split over several GPUs behind Raylib_Render.  Included at the end of rl_render.hip (the kernels launched here are
// templates defined there).
//
// Ranks.  RAYLIB_NUM_GPUS = N (default 1) makes the library drive N devices from this one process: the frame's 8x8
// cells are dealt round-robin to N logical ranks (rank r renders cells r, r + N, ...), every rank has its own
// device, stream, work buffers and scene copy and a host thread that enqueues its work, and the ranks' cell buffers are
// gathered on rank 0's device -- RCCL grouped send / recv over xGMI (librccl is loaded at run time, only then), or
// hipMemcpyPeerAsync pushes (RAYLIB_GATHER=peer, and the fallback when RCCL cannot be initialised) -- where one small
// kernel scatters them into the row-major frame.  Streams are keyed by (seed, pixel, sample), so the assembled frame is
// bit-identical to the one-device frame.  Raylib_Render keeps the reference's shape (raylib.cc:231-239): synchronous, no
// new arguments.  RAYLIB_GPU_MAP = "d0,d1,..." names the physical device of every logical rank; naming one device
// several times (e.g. "0,0,0,0") runs the whole N-rank path on one GPU, which is how the GPU test suite covers it.
// RAYLIB_GATHER_SELF=1 (tests) sends rank 0's own cells through the gather mechanism too.

namespace rl {

#define HIP_OK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	Log("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__, __LINE__, #expr); return false; } } while (0)
// the same without leaving the function: clears the local `ok` and goes on (code that has work enqueued and must still reach the place that waits for it)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	Log("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__, __LINE__, #expr); ok = false; } } while (0)

// One physical device's copy of a scene.
struct DeviceSceneCopy {
	int device = 0;
	DNode4Q* nodes4 = nullptr; DNode4* nodes4f = nullptr; DNode4* leafList = nullptr; DNode8* nodes8 = nullptr;
	DNode* nodes = nullptr; DTriIsect* isect = nullptr; DTriShade* shade = nullptr; int32_t* alphaTex = nullptr;
	DMaterial* materials = nullptr; DTexture* textures = nullptr; float* texels = nullptr;
	DSphere* spheres = nullptr; DCube* cubes = nullptr;
	// the sky panorama is read when a render starts, as the reference does (renderer.cc:159-176 dereferences the handle per miss)
	float4* sky = nullptr; size_t skyBytes = 0;
	const Image* skyImage = nullptr; uint64_t skyVersion = 0;
	DSceneView view;
};
struct DeviceScene {
	std::vector<DeviceSceneCopy*> copy;   // by device slot (Runtime::devices)
	SkyRot skyRot;
	uint32_t bvhDepth = 0, stackNeed4 = 0;
	bool hasNodes4 = false, hasNodes8 = false; uint32_t depth8 = 0; float sahNodes4 = 0.0f;
	double boundsMin[3] = { 0, 0, 0 }, boundsMax[3] = { 0, 0, 0 };   // of all triangle vertices (CullCells)
	bool boundsValid = false;                                        // triangles only, every coordinate finite
};

namespace {

// ---- RCCL, bound at run time (a one-device render never loads it) -------------------------------------------------
struct RcclApi {
	bool tried = false, ok = false;
	void* lib = nullptr;
	int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
	int (*CommDestroy)(void* comm) = nullptr;
	int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) = nullptr;
	int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	std::vector<void*> comms;   // one per device slot
};
constexpr int kRcclFloat = 7;   // ncclFloat32 (rccl.h)

// ---- a host thread per rank beyond the first: enqueues that rank's work on its device --------------------------------
struct Worker {
	std::mutex m;
	std::condition_variable cv;
	std::function<bool()> job;
	bool pending = false, result = true;
	void Start(int device)
	{
		std::thread([this, device]() {
			(void)hipSetDevice(device);
			for (;;) {
				std::function<bool()> f;
				{ std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return pending; }); f = job; }
				const bool r = f();
				{ std::lock_guard<std::mutex> lk(m); result = r; pending = false; }
				cv.notify_all();
			}
		}).detach();
	}
	void Post(std::function<bool()> f) { { std::lock_guard<std::mutex> lk(m); job = std::move(f); pending = true; } cv.notify_all(); }
	bool Wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !pending; }); return result; }
};

struct RankCtx {
	int rank = 0, device = 0, devSlot = 0, numCUs = 0;
	hipStream_t stream = nullptr;
	// Two frame slots (a whole-frame render over several ranks keeps frame i in flight while frame i + 1 is enqueued; everything else uses slot 0).
	// Per slot: 0/1 render, 2/3 megakernel, 4 "my cells are on rank 0's device", (rank 0) 5 every rank's cells are here, 6 frame assembled, 7 counters are on the host
	hipEvent_t ev[2][8] = {};
	unsigned long long* cntHost[2] = { nullptr, nullptr };   // pinned: the counter read-back must not block the enqueuing thread
	// reusable work buffers
	SampleRGB* samples = nullptr; size_t samplesBytes = 0;
	float4* accum = nullptr; size_t accumBytes = 0;
	float4* image = nullptr; size_t imageBytes = 0;
	float* pathStack = nullptr; size_t pathStackBytes = 0;
	float4* cells = nullptr; size_t cellsBytes = 0;   // N > 1: this rank's cells back to back, when they are not rendered into the gather buffer
	unsigned long long* counters = nullptr;
	unsigned int* jobCounter = nullptr;
	// cells outside the scene's silhouette (CullCells), per frame slot: the list of the others and a flag per cell, pinned on the host and on the device;
	// cullKey = what they were computed from (camera, frame, cells, bounds): an unchanged view re-uses them without a copy
	uint32_t* cellListHost[2] = { nullptr, nullptr }; uint32_t* cellList[2] = { nullptr, nullptr }; size_t cellListCells[2] = { 0, 0 };
	std::vector<unsigned char> cullKey[2];
	uint32_t cullActive[2] = { 0, 0 }; uint64_t cullEmptyPixels[2] = { 0, 0 }; float cullL[2][3] = { { 0, 0, 0 }, { 0, 0, 0 } }; uint32_t cullRays[2] = { 1, 1 };
	std::map<const void*, int> occupancy;   // blocks per CU, asked once per kernel
	Worker* worker = nullptr;
};

struct Runtime {
	bool probed = false, ok = false;
	std::vector<RankCtx*> ranks;
	std::vector<int> devices;          // the distinct physical devices, rank 0's first
	float4* gather[2] = { nullptr, nullptr }; size_t gatherBytes[2] = { 0, 0 };   // on rank 0's device: every rank's cells, rank by rank; one per frame slot
	hipStream_t gatherStream = nullptr;                  // rank 0's device: receives / waits for the ranks' cells and assembles the frame, beside rank 0's own rendering
	bool gatherSelf = false, wantRccl = true, pipeline = true;
	uint64_t frameNo = 0;                                // whole-frame renders over several ranks so far: slot = frameNo & 1
	struct Inflight;
	Inflight* inflight[2] = { nullptr, nullptr };        // enqueued, not yet waited for (oldest first by frameNo)
	RcclApi rccl;
	std::mutex lock;
};
Runtime g_rt;
inline RankCtx& Rank0() { return *g_rt.ranks[0]; }
bool ReadbackLocked(Image& img);   // device copy -> img.rgba; the runtime lock is held by the caller
bool DrainLocked();                // waits for the multi-rank frames in flight (RenderMulti); the runtime lock is held by the caller

bool EnsureRuntime()
{
	Runtime& R = g_rt;
	if (R.probed) return R.ok;
	R.probed = true;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		Log("raylib(MI355X): no HIP device visible -- Raylib_Render cannot run (there is no CPU fallback)");
		return false;
	}
	int n = 1;
	if (const char* e = getenv("RAYLIB_NUM_GPUS")) n = atoi(e);
	if (n < 1 || n > 16) { Log("raylib(MI355X): RAYLIB_NUM_GPUS=%d is outside 1..16", n); return false; }
	std::vector<int> map;
	if (const char* m = getenv("RAYLIB_GPU_MAP")) {
		for (const char* p = m; *p; ) { char* end; const long v = strtol(p, &end, 10); if (end == p) break; map.push_back((int)v); p = (*end == ',') ? end + 1 : end; }
		if ((int)map.size() < n) { Log("raylib(MI355X): RAYLIB_GPU_MAP names %d device(s) for RAYLIB_NUM_GPUS=%d", (int)map.size(), n); return false; }
		map.resize((size_t)n);
		for (int d : map) if (d < 0 || d >= count) { Log("raylib(MI355X): RAYLIB_GPU_MAP names device %d, %d visible", d, count); return false; }
	} else {
		int base = 0;
		if (const char* e = getenv("RAYLIB_DEVICE")) base = atoi(e);
		else if (n == 1) { if (const char* l = getenv("LOCAL_RANK")) base = atoi(l); }
		base = ((base % count) + count) % count;
		if (n > count) {
			Log("raylib(MI355X): RAYLIB_NUM_GPUS=%d but %d device(s) visible (RAYLIB_GPU_MAP may name a device more than once, for tests)", n, count);
			return false;
		}
		for (int r = 0; r < n; ++r) map.push_back((base + r) % count);
	}
	for (int d : map) if (std::find(R.devices.begin(), R.devices.end(), d) == R.devices.end()) R.devices.push_back(d);
	if (const char* g = getenv("RAYLIB_GATHER")) R.wantRccl = strcmp(g, "peer") != 0;
	if (const char* g = getenv("RAYLIB_GATHER_SELF")) R.gatherSelf = atoi(g) != 0;
	if (const char* g = getenv("RAYLIB_PIPELINE")) R.pipeline = atoi(g) != 0;
	for (int r = 0; r < n; ++r) {
		RankCtx* C = new RankCtx;
		C->rank = r; C->device = map[(size_t)r];
		C->devSlot = (int)(std::find(R.devices.begin(), R.devices.end(), C->device) - R.devices.begin());
		HIP_OK(hipSetDevice(C->device));
		hipDeviceProp_t prop;
		HIP_OK(hipGetDeviceProperties(&prop, C->device));
		C->numCUs = prop.multiProcessorCount;
		HIP_OK(hipStreamCreateWithFlags(&C->stream, hipStreamNonBlocking));
		for (int q = 0; q < 2; ++q) {
			for (int i = 0; i < 8; ++i) HIP_OK(hipEventCreate(&C->ev[q][i]));
			HIP_OK(hipHostMalloc((void**)&C->cntHost[q], (CNT_COUNT + 24) * sizeof(unsigned long long), hipHostMallocDefault));
		}
		HIP_OK(hipMalloc(&C->counters, (CNT_COUNT + 24 + RL_TIMELINE_SLOTS) * sizeof(unsigned long long)));
		HIP_OK(hipMalloc(&C->jobCounter, RL_MAX_HEADS * RL_HEAD_STRIDE * sizeof(unsigned int)));   // the heads of the job list, one per XCD, 128 B apart
		if (r > 0) { C->worker = new Worker; C->worker->Start(C->device); }
		R.ranks.push_back(C);
		Log("raylib(MI355X): rank %d of %d on device %d %s (%s), %d CUs", r, n, C->device, prop.name, prop.gcnArchName, C->numCUs);
	}
	if (n > 1) {   // once per process: what a scaling number will have run on
		std::string matrix;
		for (size_t a = 0; a < R.devices.size(); ++a) {
			matrix += a ? " | " : "";
			for (size_t b2 = 0; b2 < R.devices.size(); ++b2) {
				int can = a == b2 ? 1 : 0;
				if (a != b2 && hipDeviceCanAccessPeer(&can, R.devices[a], R.devices[b2]) != hipSuccess) can = -1;
				matrix += can < 0 ? "?" : (can ? "1" : "0");
			}
		}
		(void)hipGetLastError();
		Log("raylib(MI355X): %d logical rank(s) on %d distinct device(s) of %d visible; peer access (row: from, column: to) %s; gather %s, pipeline %s",
		    n, (int)R.devices.size(), count, matrix.c_str(), R.wantRccl ? "rccl (peer copies if it cannot be initialised)" : "peer copies", R.pipeline ? "two frames in flight" : "off");
	}
	// peers write their cells straight into rank 0's gather buffer
	for (size_t s = 1; s < R.devices.size(); ++s) {
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, R.devices[s], R.devices[0]) == hipSuccess && can) {
			(void)hipSetDevice(R.devices[s]);
			const hipError_t e = hipDeviceEnablePeerAccess(R.devices[0], 0);
			if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) Log("raylib(MI355X): peer access %d -> %d could not be enabled (%s); copies will be staged", R.devices[s], R.devices[0], hipGetErrorName(e));
			(void)hipGetLastError();
		}
	}
	HIP_OK(hipSetDevice(R.devices[0]));
	if (n > 1 || R.gatherSelf) HIP_OK(hipStreamCreateWithFlags(&R.gatherStream, hipStreamNonBlocking));
	R.ok = true;
	return true;
}

// RCCL communicators over the distinct devices (single process: ncclCommInitAll).  False: use peer copies.
bool EnsureRccl()
{
	Runtime& R = g_rt;
	RcclApi& A = R.rccl;
	if (A.tried) return A.ok;
	A.tried = true;
	// An RCCL the process already holds (a PyTorch process maps its own) is ADOPTED, never doubled: two copies of the library in one process would each keep
	// their own device state.  /proc/self/maps names the file that is mapped; dlopen(that path, RTLD_NOLOAD) returns the handle of exactly that copy, whatever
	// soname it was loaded under.  Only a process without any librccl loads one -- and when a mapped copy cannot be adopted the gather falls back to peer
	// copies and says so, instead of loading a second one beside it.
	std::string mapped;
	if (FILE* f = fopen("/proc/self/maps", "r")) {
		char line[1024];
		while (fgets(line, sizeof(line), f)) {
			const char* path = strchr(line, '/');
			if (!path) continue;
			const char* base = strrchr(path, '/');
			if (base && strncmp(base + 1, "librccl.so", 10) == 0) { mapped.assign(path); while (!mapped.empty() && (mapped.back() == '\n' || mapped.back() == ' ')) mapped.pop_back(); break; }
		}
		fclose(f);
	}
	const char* how = "adopted (already mapped by this process)";
	if (!mapped.empty()) {
		A.lib = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD);
		if (!A.lib) A.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
		if (!A.lib) A.lib = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
		if (!A.lib) { Log("raylib(MI355X): %s is mapped by this process but could not be adopted (%s); not loading a second RCCL -- the gather uses peer copies", mapped.c_str(), dlerror()); return false; }
	} else {
		how = "loaded by the library (no RCCL was mapped)";
		A.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!A.lib) A.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
		if (!A.lib) { Log("raylib(MI355X): librccl could not be loaded (%s); the gather uses peer copies", dlerror()); return false; }
	}
	Log("raylib(MI355X): RCCL %s: %s", how, mapped.empty() ? "librccl.so.1" : mapped.c_str());
	A.CommInitAll = (int (*)(void**, int, const int*))dlsym(A.lib, "ncclCommInitAll");
	A.CommDestroy = (int (*)(void*))dlsym(A.lib, "ncclCommDestroy");
	A.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(A.lib, "ncclSend");
	A.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(A.lib, "ncclRecv");
	A.GroupStart = (int (*)())dlsym(A.lib, "ncclGroupStart");
	A.GroupEnd = (int (*)())dlsym(A.lib, "ncclGroupEnd");
	A.GetErrorString = (const char* (*)(int))dlsym(A.lib, "ncclGetErrorString");
	if (!A.CommInitAll || !A.CommDestroy || !A.Send || !A.Recv || !A.GroupStart || !A.GroupEnd) { Log("raylib(MI355X): librccl lacks an expected symbol; the gather uses peer copies"); return false; }
	A.comms.assign(R.devices.size(), nullptr);
	const int rc = A.CommInitAll(A.comms.data(), (int)R.devices.size(), R.devices.data());
	if (rc != 0) { Log("raylib(MI355X): ncclCommInitAll failed (%s); the gather uses peer copies", A.GetErrorString ? A.GetErrorString(rc) : "?"); A.comms.clear(); return false; }
	(void)hipSetDevice(R.devices[0]);
	Log("raylib(MI355X): RCCL communicator over %d device(s)", (int)R.devices.size());
	A.ok = true;
	return true;
}

template <typename T>
bool Grow(T*& ptr, size_t& have, size_t need)
{
	if (need <= have && ptr) return true;
	if (ptr) { (void)hipFree(ptr); ptr = nullptr; have = 0; }
	HIP_OK(hipMalloc(&ptr, need));
	have = need;
	return true;
}

template <typename T>
bool Upload(T*& dst, const T* src, size_t count)
{
	size_t bytes = (count ? count : 1) * sizeof(T);
	HIP_OK(hipMalloc(&dst, bytes));
	if (count) HIP_OK(hipMemcpy(dst, src, count * sizeof(T), hipMemcpyHostToDevice));
	return true;
}

#define RL_BVH8_MIN_STEPS 40.0f
#define RL_POOL_DEFAULT_MIN_TRIS 256u
// what SelectTraceKernel's default choice comes to for a scene (no environment overrides): the pool schedule on the 8-wide tree
static bool Walks8ByDefault(const DeviceScene* D, size_t numTriangles)
{
	return D->hasNodes8 && D->depth8 <= RL_POOL8_MAXLEVELS && D->sahNodes4 >= RL_BVH8_MIN_STEPS && numTriangles >= RL_POOL_DEFAULT_MIN_TRIS;
}
// the wide tree a render is about to walk, on this rank's device (uploaded with the scene only when it is the default choice)
static bool EnsureWideTree(DeviceSceneCopy* C, const Scene& sc, int width, bool gridNodes)
{
	if (width == 8 && !C->nodes8 && !sc.bvh.nodes8.empty()) {
		if (!Upload(C->nodes8, sc.bvh.nodes8.data(), sc.bvh.nodes8.size())) return false;
		C->view.nodes8 = C->nodes8;
	}
	if (width == 4 && gridNodes && RL_Q4 && !C->nodes4 && !sc.bvh.nodes4q.empty()) {
		if (!Upload(C->nodes4, sc.bvh.nodes4q.data(), sc.bvh.nodes4q.size())) return false;
		C->view.nodes4 = C->nodes4;
	}
	return true;
}
void FreeCopy(DeviceSceneCopy* C)
{
	if (!C) return;
	(void)hipSetDevice(C->device);
	(void)hipFree(C->nodes); if (C->nodes4) (void)hipFree(C->nodes4); if (C->nodes4f) (void)hipFree(C->nodes4f); if (C->nodes8) (void)hipFree(C->nodes8); if (C->leafList) (void)hipFree(C->leafList); (void)hipFree(C->isect); (void)hipFree(C->shade); if (C->alphaTex) (void)hipFree(C->alphaTex);
	(void)hipFree(C->materials); (void)hipFree(C->textures); (void)hipFree(C->texels); (void)hipFree(C->spheres); (void)hipFree(C->cubes);
	if (C->sky) (void)hipFree(C->sky);
	delete C;
}
void FreeScene(DeviceScene* D)
{
	if (!D) return;
	for (DeviceSceneCopy* C : D->copy) FreeCopy(C);
	delete D;
	if (g_rt.ok) (void)hipSetDevice(g_rt.devices[0]);
}

// Flatten the host scene into device records (leaf order) and upload a copy to every device in use.
bool UploadScene(Scene& sc)
{
	if (sc.device) return true;
	const size_t n = sc.triangles.size();
	std::vector<DTriIsect> isect(n);
	std::atomic<int> fastBary(1);
	if (const char* e = getenv("RAYLIB_FAST_BARY")) fastBary.store(atoi(e) != 0 ? 1 : 0);   // 0: the divisions, whatever the scene (parity tests compare the two)
	std::vector<DTriShade> shade(n);
	auto flatten = [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; ++k) {
		const HostTriangle& t = sc.triangles[sc.bvh.triOrder[k]];
		DTriIsect& I = isect[k];
		const f3 nrm = normalize(cross(t.v1 - t.v0, t.v2 - t.v0));   // geom/triangle.h:34-38
		const f3 u = t.v1 - t.v0, v = t.v2 - t.v0;                   // geom/triangle.cc:30-31
		const float uv = dot(u, v), uu = dot(u, u), vv = dot(v, v);  // :34-38
		const float uvuv = uv * uv, uuvv = uu * vv;                  // :39-40
		I.v0[0] = t.v0.x; I.v0[1] = t.v0.y; I.v0[2] = t.v0.z;
		I.n[0] = nrm.x; I.n[1] = nrm.y; I.n[2] = nrm.z;
		I.v1[0] = t.v1.x; I.v1[1] = t.v1.y; I.v1[2] = t.v1.z;
		I.v2[0] = t.v2.x; I.v2[1] = t.v2.y; I.v2[2] = t.v2.z;
		I.uv = uv; I.uu = uu; I.vv = vv;
		{   // the reciprocal of denom = uvuv - uuvv for the short barycentric divisions (rl_render.hip Barycentric, which states the conditions)
			const float denom = uvuv - uuvv, mag = fabsf(denom);
			if (denom == 0.0f || denom != denom) I.rden = std::numeric_limits<float>::quiet_NaN();
			else if (mag >= 0x1p-62f && mag <= 0x1p125f) I.rden = 1.0f / denom;
			else { I.rden = std::numeric_limits<float>::quiet_NaN(); fastBary.store(0, std::memory_order_relaxed); }
		}
		DTriShade& Sh = shade[k];
		Sh.n0[0] = t.n0.x; Sh.n0[1] = t.n0.y; Sh.n0[2] = t.n0.z;
		Sh.n1[0] = t.n1.x; Sh.n1[1] = t.n1.y; Sh.n1[2] = t.n1.z;
		Sh.n2[0] = t.n2.x; Sh.n2[1] = t.n2.y; Sh.n2[2] = t.n2.z;
		Sh.s0 = t.s0; Sh.t0 = t.t0; Sh.s1 = t.s1; Sh.t1 = t.t1; Sh.s2 = t.s2; Sh.t2 = t.t2;
		Sh.material = t.material;
	} };
	{   // per-triangle records are independent: all host threads for large scenes (10 M triangles: 0.6 s on one thread)
		unsigned threads = n >= (1u << 17) ? std::min(32u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
		if (const char* e = getenv("RAYLIB_BUILD_THREADS")) { int v = atoi(e); if (v > 0 && n >= (1u << 17)) threads = (unsigned)std::min(v, 32); }
		std::vector<std::thread> pool;
		const size_t per = (n + threads - 1) / threads;
		for (unsigned t = 1; t < threads; ++t) { const size_t k0 = std::min(n, t * per), k1 = std::min(n, (t + 1) * per); if (k0 < k1) pool.emplace_back(flatten, k0, k1); }
		flatten(0, std::min(n, per));
		for (std::thread& th : pool) th.join();
	}
	std::vector<DMaterial> mats(sc.materials.size());
	for (size_t i = 0; i < mats.size(); ++i) {
		const HostMaterial& h = sc.materials[i];
		DMaterial& m = mats[i]; memset(&m, 0, sizeof(m));
		m.type = h.type;
		memcpy(m.albedo, h.albedo, 12); m.roughness = h.roughness; m.metallic = h.metallic;
		memcpy(m.emissive, h.emissive, 12); m.ior = h.ior; memcpy(m.transmission, h.transmission, 12);
		m.fuzziness = h.fuzziness; memcpy(m.tex, h.tex, 20);
	}
	std::vector<DTexture> texs(sc.textures.size());
	std::vector<float> pool;
	for (size_t i = 0; i < texs.size(); ++i) {
		const Image& im = *sc.textures[i];
		texs[i].offset = (uint32_t)(pool.size() / 4); texs[i].width = (int32_t)im.width; texs[i].height = (int32_t)im.height; texs[i].pad = 0;
		if (im.hostStale) {   // a rendered image used as a texture: fetch it (the runtime lock is held here)
			Image& w = const_cast<Image&>(im);
			if (!ReadbackLocked(w)) Log("UploadScene: texture %zu could not be read back from the device", i);
			w.hostStale = false;
		}
		pool.insert(pool.end(), im.rgba.begin(), im.rgba.end());
	}
	// Albedo maps are read through Texture2D::Sample(bSRGB = true): nearest texel, then pow(texel, 2.2) on all four channels
	// (reference render/texture.cc:44-50, material.cc:383,400) -- four powf per shading event and per alpha-tested candidate.
	// The power of a texel does not depend on the ray: every texture some material uses as albedo gets a converted copy here
	// (host powf = the reference's own function, the one csrc/rl_glibc_math.h restates), and the material points at the copy.
	{
		std::vector<int32_t> converted(texs.size(), -1);
		for (DMaterial& m : mats) {
			if (m.type != MAT_MICROFACET || m.tex[0] < 0 || (size_t)m.tex[0] >= converted.size()) continue;
			const size_t src = (size_t)m.tex[0];
			if (converted[src] < 0) {
				DTexture t = texs[src];
				const size_t count = (size_t)t.width * t.height * 4, from = (size_t)t.offset * 4;
				t.offset = (uint32_t)(pool.size() / 4);
				pool.resize(pool.size() + count);
				float* dst = pool.data() + (size_t)t.offset * 4; const float* in = pool.data() + from;
				unsigned threads = count >= (1u << 20) ? std::min(32u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
				std::vector<std::thread> workers;
				const size_t per = (count + threads - 1) / threads;
				auto run = [dst, in](size_t a, size_t b) { for (size_t i = a; i < b; ++i) dst[i] = powf(in[i], 2.2f); };
				for (unsigned w = 1; w < threads; ++w) { const size_t a = std::min(count, w * per), b = std::min(count, (w + 1) * per); if (a < b) workers.emplace_back(run, a, b); }
				run(0, std::min(count, per));
				for (std::thread& th : workers) th.join();
				converted[src] = (int32_t)texs.size();
				texs.push_back(t);
			}
			m.tex[0] = converted[src];
		}
	}
	// The cut-out test of a candidate (geom/triangle.cc:54 -> MicrofacetMaterial::AlphaTest) needs one texel of the triangle's material's albedo map: as a table per
	// triangle slot the texture is known from the triangle alone, and the test's chain of dependent loads inside the walk is triangle -> texture -> texel instead
	// of triangle -> material -> texture -> texel (round 5).
	std::vector<int32_t> alphaTex;
	{
		bool any = false;
		for (const DMaterial& m : mats) if (m.type == MAT_MICROFACET && m.tex[0] >= 0) any = true;
		if (any) {
			alphaTex.resize(n);
			for (size_t k = 0; k < n; ++k) {
				const int32_t mi = shade[k].material;
				alphaTex[k] = (mi >= 0 && (size_t)mi < mats.size() && mats[(size_t)mi].type == MAT_MICROFACET && mats[(size_t)mi].tex[0] >= 0) ? mats[(size_t)mi].tex[0] : -1;
			}
		}
	}
	std::vector<DSphere> dsph(sc.spheres.size());
	for (size_t i = 0; i < dsph.size(); ++i) {
		memset(&dsph[i], 0, sizeof(DSphere));
		dsph[i].center[0] = sc.spheres[i].center.x; dsph[i].center[1] = sc.spheres[i].center.y; dsph[i].center[2] = sc.spheres[i].center.z;
		dsph[i].radius = sc.spheres[i].radius; dsph[i].material = sc.spheres[i].material;
	}
	std::vector<DCube> dcub(sc.cubes.size());
	for (size_t i = 0; i < dcub.size(); ++i) {
		memset(&dcub[i], 0, sizeof(DCube));
		const HostCube& h = sc.cubes[i];
		dcub[i].minBounds[0] = h.minBounds.x; dcub[i].minBounds[1] = h.minBounds.y; dcub[i].minBounds[2] = h.minBounds.z; dcub[i].timeStartMove = h.timeStartMove;
		dcub[i].maxBounds[0] = h.maxBounds.x; dcub[i].maxBounds[1] = h.maxBounds.y; dcub[i].maxBounds[2] = h.maxBounds.z; dcub[i].material = h.material;
		dcub[i].velocity[0] = h.velocity.x; dcub[i].velocity[1] = h.velocity.y; dcub[i].velocity[2] = h.velocity.z;
	}

	DeviceScene* D = new DeviceScene;
	D->bvhDepth = sc.bvh.depth; D->stackNeed4 = sc.bvh.stackNeed4; D->hasNodes4 = !sc.bvh.nodes4.empty(); D->hasNodes8 = !sc.bvh.nodes8.empty(); D->depth8 = sc.bvh.depth8; D->sahNodes4 = sc.bvh.sahNodes4;
	{   // Rotator(yaw = 90).rotate rows, reference geom/transform.cc:47-65 (host libm, as the reference)
		const float pi_f = (float)3.1415926535897932385;
		const float ry = 90.0f * pi_f / 180.0f, rp = 0.0f * pi_f / 180.0f, rr = 0.0f * pi_f / 180.0f;
		const float ch = cosf(ry), sh = sinf(ry), cp = cosf(rp), sp = sinf(rp), cb = cosf(rr), sb = sinf(rr);
		D->skyRot.m0[0] = ch * cb + sh * sp * sb; D->skyRot.m0[1] = sb * cp; D->skyRot.m0[2] = -sh * cb + ch * sp * sb;
		D->skyRot.m1[0] = -ch * sb + sh * sp * cb; D->skyRot.m1[1] = cb * cp; D->skyRot.m1[2] = sb * sh + ch * sp * cb;
		D->skyRot.m2[0] = sh * cp; D->skyRot.m2[1] = -sp; D->skyRot.m2[2] = ch * cp;
	}
	{   // the scene's bounding box: the root node's two child boxes (CullCells)
		D->boundsValid = false;
		if (sc.spheres.empty() && sc.cubes.empty() && !sc.bvh.nodes.empty() && n > 0) {
			const DNode& root = sc.bvh.nodes[0];
			bool ok = true;
			for (int k = 0; k < 3; ++k) {
				double lo = 1e300, hi = -1e300;
				if (root.left != DNODE_EMPTY) { lo = std::min(lo, (double)root.lmin[k]); hi = std::max(hi, (double)root.lmax[k]); }
				if (root.right != DNODE_EMPTY) { lo = std::min(lo, (double)root.rmin[k]); hi = std::max(hi, (double)root.rmax[k]); }
				if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) ok = false;
				D->boundsMin[k] = lo; D->boundsMax[k] = hi;
			}
			D->boundsValid = ok;
		}
	}
	for (size_t slot = 0; slot < g_rt.devices.size(); ++slot) {
		DeviceSceneCopy* C = new DeviceSceneCopy;
		C->device = g_rt.devices[slot];
		memset(&C->view, 0, sizeof(C->view));
		D->copy.push_back(C);
		bool ok = hipSetDevice(C->device) == hipSuccess;
		ok = ok && Upload(C->nodes, sc.bvh.nodes.data(), sc.bvh.nodes.size());
		// the grid nodes for the pool schedule; the float-box nodes for scenes small enough to stay in the caches (k_trace's class),
		// where the grid's extra arithmetic buys nothing (Cornell frame: 22.8 ms on float boxes, 23.8 ms on the grid)
		const bool wantFull = D->hasNodes4 && (!RL_Q4 || sc.triangles.size() < 4096);
		// Of the two wide trees the pool schedule can walk, the one its default choice walks (SelectTraceKernel: the 8-wide tree for scenes whose rays are expected to take
		// many steps) goes to the device now; the other follows the first time a render asks for it (EnsureWideTree: RAYLIB_BVH8 / RAYLIB_BVH4 / RAYLIB_POOL are read
		// per render).  A 10 M-triangle scene keeps 190 MB of nodes per device instead of 290.
		const bool walks8 = Walks8ByDefault(D, sc.triangles.size());
		if (ok && D->hasNodes4 && RL_Q4 && !walks8) ok = Upload(C->nodes4, sc.bvh.nodes4q.data(), sc.bvh.nodes4q.size());
		if (ok && D->hasNodes8 && walks8) ok = Upload(C->nodes8, sc.bvh.nodes8.data(), sc.bvh.nodes8.size());
		if (ok && wantFull) ok = Upload(C->nodes4f, sc.bvh.nodes4.data(), sc.bvh.nodes4.size());
		if (ok && wantFull && !sc.bvh.leafList.empty()) ok = Upload(C->leafList, sc.bvh.leafList.data(), sc.bvh.leafList.size());
		ok = ok && Upload(C->isect, isect.data(), n) && Upload(C->shade, shade.data(), n);
		if (ok && !alphaTex.empty()) ok = Upload(C->alphaTex, alphaTex.data(), alphaTex.size());
		ok = ok && Upload(C->materials, mats.data(), mats.size()) && Upload(C->textures, texs.data(), texs.size()) && Upload(C->texels, pool.data(), pool.size());
		ok = ok && Upload(C->spheres, dsph.data(), dsph.size()) && Upload(C->cubes, dcub.data(), dcub.size());
		if (!ok) { Log("UploadScene: device %d could not take the scene", C->device); FreeScene(D); return false; }   // nothing of a failed upload is left behind
		DSceneView& V = C->view;
		V.nodes = C->nodes; V.nodes4 = C->nodes4; V.nodes4f = C->nodes4f; V.nodes8 = C->nodes8; V.isect = C->isect; V.shade = C->shade; V.alphaTex = C->alphaTex; V.materials = C->materials;
		V.textures = C->textures; V.numTextures = (int32_t)texs.size(); V.texels = C->texels; V.spheres = C->spheres; V.cubes = C->cubes;
		V.sunIlluminance[0] = sc.sunIlluminance.x; V.sunIlluminance[1] = sc.sunIlluminance.y; V.sunIlluminance[2] = sc.sunIlluminance.z;
		V.sunDirection[0] = sc.sunDirection.x; V.sunDirection[1] = sc.sunDirection.y; V.sunDirection[2] = sc.sunDirection.z;
		V.sky = nullptr; V.skyWidth = V.skyHeight = 0;
		V.hasSun = !(sc.sunIlluminance.x == 0.0f && sc.sunIlluminance.y == 0.0f && sc.sunIlluminance.z == 0.0f);   // renderer.cc:192
		V.numTriangles = (int32_t)n;
		V.numNodes4 = (int32_t)sc.bvh.nodes4.size(); V.numMaterials = (int32_t)mats.size();
		V.leafList = C->leafList; V.numLeafRecords = C->leafList ? (int32_t)sc.bvh.leafList.size() : 0;
		V.fastBary = fastBary.load();
		V.numNodes8 = (int32_t)sc.bvh.nodes8.size();

	}
	HIP_OK(hipSetDevice(g_rt.devices[0]));
	sc.device = D;
	return true;
}

// The sky panorama as it is NOW: the reference reads the image through the handle at every miss (renderer.cc:159-176), so pixels
// written after Raylib_FinalizeScene / Raylib_SetSkyPanorama are seen by the next render.  Only the sky texels move; triangles,
// BVH and textures stay where they are.  A frame that was rendered into the image and never left the device is copied on it.
bool SyncSky(Scene& sc)
{
	DeviceScene* D = sc.device;
	Image* sky = sc.sky;
	for (DeviceSceneCopy* C : D->copy) {
		if (!sky || (size_t)sky->width * sky->height == 0) { C->view.sky = nullptr; C->view.skyWidth = C->view.skyHeight = 0; C->skyImage = nullptr; continue; }
		if (C->skyImage == sky && C->skyVersion == sky->version && C->view.sky) continue;
		const size_t bytes = (size_t)sky->width * sky->height * sizeof(float4);
		(void)DrainLocked();   // a frame in flight reads the texels that are about to be replaced
		HIP_OK(hipSetDevice(C->device));
		if (!Grow(C->sky, C->skyBytes, bytes)) return false;
		if (sky->hostStale && sky->devValid && sky->devPixels) {
			HIP_OK(hipMemcpyPeer(C->sky, C->device, sky->devPixels, Rank0().device, bytes));   // images live on rank 0's device
		} else {
			HIP_OK(hipMemcpy(C->sky, sky->rgba.data(), bytes, hipMemcpyHostToDevice));
		}
		C->skyImage = sky; C->skyVersion = sky->version;
		C->view.sky = (const float*)C->sky; C->view.skyWidth = (int32_t)sky->width; C->view.skyHeight = (int32_t)sky->height;
	}
	HIP_OK(hipSetDevice(g_rt.devices[0]));
	return true;
}

typedef void (*TraceKernel)(const DRenderParams, const DSceneView, const SkyRot, SampleRGB*, float*, unsigned long long*, unsigned int*);

// poolK = 0: k_trace (one path per lane); poolK = K: k_trace_pool with 64*K paths per wave
template <int STACK, bool PRIMS>
TraceKernel SelectTraceKernel(int& poolK, const DeviceScene* D, bool& shortStack, int& width)
{
	shortStack = false; width = 2;
	if constexpr (STACK <= 32 && !PRIMS) {
		const char* e = getenv("RAYLIB_POOL_SHORT_STACK");
		// the wide tree: default whenever the scene carries one whose worst-case stack fits; RAYLIB_BVH4=0|1 overrides
		const char* w = getenv("RAYLIB_BVH4");
		const bool haveWide = D && D->hasNodes4 && D->stackNeed4 <= 64;   // (the pool's node format is a build-time choice: RL_Q4)
		const bool wantWide = haveWide && (w ? atoi(w) != 0 : true);
		if (poolK == 2 && wantWide) {
			width = 4; shortStack = true;
			if (e && atoi(e) == 0) { shortStack = false; return D->stackNeed4 <= 32 ? (TraceKernel)k_trace_pool<32, PRIMS, 2, 32, 1> : (TraceKernel)k_trace_pool<64, PRIMS, 2, 32, 1>; }
			// the 8-wide tree (0.67 x the steps of the 4-wide one, each 1.4 x as long) when the scene carries one of at most 16 levels -- a group of hit children per
			// level is all its stack ever holds -- and its rays are expected to take many steps: measured over rooms and colonnades of 1 k ... 10 M triangles
			// (tools/gpu_bvh8_sweep.py, profiles/r04_bvh8_sweep.log) the 8-wide walk loses 2 - 8 % below ~30 expected steps of the 4-wide tree (the builder's sum of
			// node areas over the root's), breaks even between 30 and 42 and wins 3 - 9 % from 59 up.  RAYLIB_BVH8=0|1 overrides the choice.
			const char* w8 = getenv("RAYLIB_BVH8");
			const bool want8 = D->hasNodes8 && (w8 ? atoi(w8) != 0 : D->sahNodes4 >= RL_BVH8_MIN_STEPS);
			if (want8 && D->depth8 <= RL_POOL8_MAXLEVELS) { width = 8; return (TraceKernel)k_trace_pool<2 * RL_POOL8_MAXLEVELS, PRIMS, 2, RL_POOL8_LSTACK, 3>; }
			if (want8) {   // (the kernel's stack holds one group of hit children per level: a deeper 8-wide tree is not walked, and that is said once per scene)
				static const DeviceScene* told = nullptr;
				if (told != D) { told = D; Log("Raylib_Render: the scene's 8-wide tree has %u levels, the pool kernel's stack holds %d: walking the 4-wide tree", D->depth8, (int)RL_POOL8_MAXLEVELS); }
			}
			return D->stackNeed4 <= 32 ? (TraceKernel)k_trace_pool<32, PRIMS, 2, RL_POOL_SHORT_LSTACK, 1> : (TraceKernel)k_trace_pool<64, PRIMS, 2, RL_POOL_SHORT_LSTACK, 1>;
		}
		if constexpr (STACK == 32) {
			const bool wantShort = e ? atoi(e) != 0 : D->bvhDepth <= RL_POOL_SHORT_MAXDEPTH;
			if (poolK == 2 && e && atoi(e) == 4) { shortStack = true; return k_trace_pool<STACK, PRIMS, 2, 4>; }   // tests: nearly every push overflows
			if (poolK == 2 && wantShort) { shortStack = true; return k_trace_pool<STACK, PRIMS, 2, RL_POOL_SHORT_LSTACK>; }
		}
		if (poolK == 2) return k_trace_pool<STACK, PRIMS, 2>;
		if (poolK == 3) return k_trace_pool<STACK, PRIMS, 3>;
		if (poolK == 4) return k_trace_pool<STACK, PRIMS, 4>;
	}
	poolK = 0;
	return nullptr;   // k_trace: the caller picks the instantiation for the node format the scene carries
}

// What EnqueueRender leaves for FinishRender: everything is queued on the rank's stream, nothing has been waited for.
struct PendingRender {
	RankCtx* ctx = nullptr;
	bool pathTrace = false, lastBatchPending = false;
	float traceMs = 0.0f;
	uint32_t launches = 0, schedulePaths = 1, jobHeads = 0, treeWidth = 2;
	uint64_t pixels = 0;
	uint64_t culledSamples = 0; uint32_t culledRaysPerSample = 0, culledSkyTexels = 0;   // camera samples of cells outside the scene's silhouette: reported apart, not traced (CullCells)
	uint32_t culledCells = 0, listedCells = 0;
	bool enqueuedToEnd = false;                // EnqueueRender reached the ev[slot][7] record (FinishRender waits for it; otherwise for the whole stream)
	float4* out = nullptr; size_t outBytes = 0;
	int slot = 0;                              // frame slot: which of the rank's event sets / pinned counter buffers this render uses
	const unsigned long long* cnt = nullptr;   // -> ctx->cntHost[slot], valid once ev[slot][7] has fired
};

// One rank's share of a render, queued on its stream: counters reset, the megakernel (or k_aov) per sample batch, k_resolve,
// end event, counter read-back.  `req.outDevice` receives the row-major frame (cellStride 1) or the rank's cells back to back.
template <int STACK, bool PRIMS>
bool EnqueueRender(RankCtx& R, Scene& sc, const RenderRequest& req, PendingRender& pend)
{
	DeviceScene* DS = sc.device;
	DeviceSceneCopy* D = DS->copy[(size_t)R.devSlot];
	const RendererSettings& st = req.settings;
	const uint32_t W = st.viewportWidth, H = st.viewportHeight;
	const uint32_t cellsX = (W + 7) / 8, cellsY = (H + 7) / 8, numCells = cellsX * cellsY;
	const uint32_t stride = req.cellStride ? req.cellStride : 1;
	const uint32_t numLocalCells = req.cellFirst < numCells ? (numCells - req.cellFirst + stride - 1) / stride : 0;
	const uint32_t numSlots = numLocalCells * 64u;
	const bool rowMajor = (stride == 1 && req.cellFirst == 0 && !req.cellMajor);
	const uint32_t SPP = (uint32_t)(st.samplesPerPixel > 1 ? st.samplesPerPixel : 1);
	const bool pathTrace = (st.renderMode == RAYLIB_RENDERMODE_Default);
	pend.ctx = &R; pend.pathTrace = pathTrace;
	const int q = req.slot & 1;
	pend.slot = q; pend.cnt = R.cntHost[q];

	DRenderParams P; memset(&P, 0, sizeof(P));
	P.width = W; P.height = H; P.spp = SPP; P.maxPathLength = st.maxPathLength; P.rayTMin = st.rayTMin;
	P.invWidth = 1.0f / (float)W; P.invHeight = 1.0f / (float)H;   // correctly rounded (IEEE division on the host): rl_render.hip PixelUV
	P.renderMode = st.renderMode; P.seed = req.seed; P.cellsX = cellsX; P.cellsY = cellsY;
	P.cellFirst = req.cellFirst; P.cellStride = stride; P.numLocalCells = numLocalCells;
	P.rowMajorOutput = rowMajor ? 1u : 0u; P.camera = req.camera;
	P.seedMixed = raylib_rng_mix64(req.seed);
	P.magicCellsX = cellsX > 1 ? (uint32_t)(0x100000000ull / cellsX) : 0xFFFFFFFFu;

	const size_t outBytes = rowMajor ? (size_t)W * H * sizeof(float4) : (size_t)numSlots * sizeof(float4);
	float4* out = (float4*)req.outDevice;
	if (!out) { if (!Grow(R.image, R.imageBytes, outBytes ? outBytes : 16)) return false; out = R.image; }
	pend.out = out; pend.outBytes = outBytes;

	HIP_OK(hipMemsetAsync(R.counters, 0, (CNT_COUNT + 24 + RL_TIMELINE_SLOTS) * sizeof(unsigned long long), R.stream));
	HIP_OK(hipEventRecord(R.ev[q][0], R.stream));
	if (numSlots == 0) {
		// nothing to do for this rank
	} else if (!pathTrace) {
		const uint32_t blocks = (numSlots + RL_BLOCK - 1) / RL_BLOCK;
		hipLaunchKernelGGL((k_aov<STACK, PRIMS>), dim3(blocks), dim3(RL_BLOCK), 0, R.stream, P, D->view, out, R.counters);
		HIP_OK(hipGetLastError());
	} else {
		// sample batches: one launch per <= 16 GiB of sample buffer (288 GB of HBM: few, large launches -- every launch pays its
		// ramp-up and its tail once; measured on the 298 k-triangle scene at 128 spp: 1 launch 61.3 ms, 2 launches 68.9, 4 launches 90.1)
		const size_t perSample = (size_t)numSlots * sizeof(SampleRGB);
		size_t capBytes = (size_t)16 << 30;
		if (const char* e = getenv("RAYLIB_SAMPLE_BUFFER_GIB")) { const int v = atoi(e); if (v > 0) capBytes = (size_t)v << 30; }
		uint32_t batch = (uint32_t)std::max<size_t>(1, std::min<size_t>(SPP, capBytes / perSample));
		if (const char* e = getenv("RAYLIB_SAMPLE_BATCH")) { int v = atoi(e); if (v > 0) batch = std::min<uint32_t>((uint32_t)v, SPP); }
		if (!Grow(R.samples, R.samplesBytes, perSample * batch)) return false;
		if (batch < SPP && !Grow(R.accum, R.accumBytes, (size_t)numSlots * sizeof(float4))) return false;
		// Scheduling of the megakernel: the pool schedule for triangle scenes from RAYLIB_POOL_MIN_TRIS triangles on, else k_trace
		// (the Cornell class: tens of triangles, shading-bound).  Measured crossover (tools/gpu_crossover.py, tessellated rooms at
		// 1080p x 16 spp, pool time / k_trace time): 36 triangles 1.07, 144: 0.97, 324: 0.95, 1296: 0.89, 5184: 0.80, 20736: 0.67.
		// RAYLIB_POOL=0|2|3|4 overrides.
		uint32_t minTris = RL_POOL_DEFAULT_MIN_TRIS; if (const char* e = getenv("RAYLIB_POOL_MIN_TRIS")) minTris = (uint32_t)atoi(e);
		int poolK = (STACK <= 32 && !PRIMS && sc.triangles.size() >= minTris) ? 2 : 0;
		if (const char* e = getenv("RAYLIB_POOL")) poolK = atoi(e);
		bool shortStack = false; int width = 2;
		TraceKernel traceKernel = SelectTraceKernel<STACK, PRIMS>(poolK, DS, shortStack, width);
		// k_trace walks the 4-wide tree too when the scene has one whose worst-case stack fits this instantiation's LDS stack:
		// on float boxes if the scene carries them (small scenes), else on the grid nodes
		// (k_trace walks the grid nodes too when the scene carries no float-box ones)
		if (!EnsureWideTree(D, sc, poolK > 0 ? width : (DS->hasNodes4 ? 4 : 2), poolK > 0 || !D->nodes4f)) return false;
		DSceneView traceView = D->view;
#if !RL_Q4
		traceView.nodes4 = nullptr;
#endif
		if (poolK == 0) {
			const char* w = getenv("RAYLIB_BVH4");
			const bool baseWide = !PRIMS && DS->hasNodes4 && DS->stackNeed4 <= (uint32_t)STACK && (w ? atoi(w) != 0 : true);
			if (!baseWide) { traceView.nodes4 = nullptr; traceView.nodes4f = nullptr; }
			width = baseWide ? 4 : 2;
			const bool full = traceView.nodes4f != nullptr;
			if (full) traceView.nodes4 = nullptr;
			// the whole scene in LDS when it fits the fixed layout (rl_render.hip RL_LDS_*); RAYLIB_LDS_SCENE=0 keeps it in global memory
			bool lds = false;
			if constexpr (STACK == 16 && !PRIMS) {
				const char* e = getenv("RAYLIB_LDS_SCENE");
				lds = full && (e ? atoi(e) != 0 : true) && sc.bvh.nodes4.size() <= RL_LDS_MAXNODES && sc.triangles.size() <= RL_LDS_MAXTRIS && sc.materials.size() <= RL_LDS_MAXMATS;
				// ... and a scene of few leaves without a tree (rl_bvh.cc "the leaf list"); RAYLIB_LEAF_LIST=0 walks its BVH4 instead
				const char* f = getenv("RAYLIB_LEAF_LIST");
				const bool flat = lds && traceView.leafList != nullptr && traceView.numLeafRecords <= RL_LEAFLIST_RECORDS && sc.triangles.size() <= RL_LEAFLIST_MAXTRIS && (f ? atoi(f) != 0 : true)
				                  && st.rayTMin >= 0.0f;   // its sortable keys are entry distances, never negative (rl_render.hip TraverseLeafList)
				if (flat) { traceKernel = (TraceKernel)k_trace<STACK, PRIMS, true, 2>; width = 0; }
				else if (lds) traceKernel = (TraceKernel)k_trace<STACK, PRIMS, true, 1>;
			}
			if (!lds) traceKernel = full ? (TraceKernel)k_trace<STACK, PRIMS, true> : (TraceKernel)k_trace<STACK, PRIMS, false>;
		}
		const uint32_t pathsPerThread = poolK > 0 ? (uint32_t)poolK : 1u;
		pend.schedulePaths = pathsPerThread; pend.treeWidth = (uint32_t)width;
		// ---- cells that cannot see the scene leave the job list (CullCells) ----
		uint32_t numActive = numLocalCells;
		{
			// what the decision depends on: an unchanged view keeps the lists the slot already holds on the device
			std::vector<unsigned char> key;
			auto put = [&](const void* ptr, size_t n) { const unsigned char* b = (const unsigned char*)ptr; key.insert(key.end(), b, b + n); };
			const uint32_t geo[7] = { W, H, req.cellFirst, stride, numLocalCells, (uint32_t)st.maxPathLength, __builtin_bit_cast(uint32_t, st.rayTMin) };
			const int flags[3] = { traceView.sky ? 1 : 0, traceView.hasSun, getenv("RAYLIB_CULL_CELLS") ? atoi(getenv("RAYLIB_CULL_CELLS")) : 1 };
			put(&req.camera, sizeof(req.camera)); put(geo, sizeof(geo)); put(flags, sizeof(flags)); put(traceView.sunDirection, sizeof(traceView.sunDirection));
			put(traceView.sunIlluminance, sizeof(traceView.sunIlluminance)); put(DS->boundsMin, sizeof(DS->boundsMin)); put(DS->boundsMax, sizeof(DS->boundsMax));
			const void* sceneId = DS; put(&sceneId, sizeof(sceneId));
			if (key != R.cullKey[q]) {
				CullResult cr;
				CullScene cs;
				for (int k = 0; k < 3; ++k) { cs.boundsMin[k] = DS->boundsMin[k]; cs.boundsMax[k] = DS->boundsMax[k]; cs.sunDirection[k] = traceView.sunDirection[k]; cs.sunIlluminance[k] = traceView.sunIlluminance[k]; }
				cs.boundsValid = DS->boundsValid; cs.prims = PRIMS; cs.hasSky = traceView.sky != nullptr; cs.hasSun = traceView.hasSun != 0;
				const bool culled = CullCells(cs, req.camera, st.maxPathLength, st.rayTMin, W, H, cellsX, req.cellFirst, stride, numLocalCells, cr);
				R.cullKey[q].clear();   // (valid again once the slot's buffers hold this view)
				R.cullActive[q] = numLocalCells; R.cullEmptyPixels[q] = 0;
				if (culled) {
					if (R.cellListCells[q] < numLocalCells) {
						if (R.cellListHost[q]) { (void)hipHostFree(R.cellListHost[q]); R.cellListHost[q] = nullptr; }
						if (R.cellList[q]) { (void)hipFree(R.cellList[q]); R.cellList[q] = nullptr; }
						R.cellListCells[q] = 0;
						// the list (uint32 per cell) and the flags (one byte per cell, behind it) in one buffer
						HIP_OK(hipHostMalloc((void**)&R.cellListHost[q], (size_t)numLocalCells * 5 + 16, hipHostMallocDefault));
						HIP_OK(hipMalloc((void**)&R.cellList[q], (size_t)numLocalCells * 5 + 16));
						R.cellListCells[q] = numLocalCells;
					}
					// (the slot's staging buffer is free: the frame that used it last has been waited for -- FinishRender, or FinishInflight before a slot is re-used)
					memcpy(R.cellListHost[q], cr.active.data(), cr.active.size() * sizeof(uint32_t));
					memcpy((unsigned char*)(R.cellListHost[q] + R.cellListCells[q]), cr.empty.data(), numLocalCells);
					HIP_OK(hipMemcpyAsync(R.cellList[q], R.cellListHost[q], (size_t)R.cellListCells[q] * 5, hipMemcpyHostToDevice, R.stream));
					R.cullActive[q] = (uint32_t)cr.active.size(); R.cullEmptyPixels[q] = cr.emptyPixels;
					R.cullL[q][0] = cr.L[0]; R.cullL[q][1] = cr.L[1]; R.cullL[q][2] = cr.L[2]; R.cullRays[q] = cr.raysPerSample;
				}
				R.cullKey[q] = key;
			}
			if (R.cullActive[q] < numLocalCells) {
				numActive = R.cullActive[q];
				P.activeCells = R.cellList[q]; P.cellEmpty = (const uint8_t*)(R.cellList[q] + R.cellListCells[q]); P.numActiveCells = numActive;
				P.emptyL[0] = R.cullL[q][0]; P.emptyL[1] = R.cullL[q][1]; P.emptyL[2] = R.cullL[q][2];
				P.emptySky = traceView.sky ? 1u : 0u;
				pend.culledSamples = R.cullEmptyPixels[q] * (uint64_t)SPP; pend.culledRaysPerSample = R.cullRays[q]; pend.culledSkyTexels = traceView.sky ? 1u : 0u;
			}
			pend.culledCells = numLocalCells - numActive; pend.listedCells = numActive;
		}
		int blocksPerCU = 0;
		{
			auto it = R.occupancy.find((const void*)traceKernel);
			if (it == R.occupancy.end()) {
				HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocksPerCU, traceKernel, RL_BLOCK, 0));
				R.occupancy[(const void*)traceKernel] = blocksPerCU;
				if (getenv("RAYLIB_PRINT_OCCUPANCY")) Log("megakernel (%u paths per lane, tree width %d): %d workgroups per CU", pathsPerThread, width, blocksPerCU);
			} else blocksPerCU = it->second;
		}
		if (blocksPerCU < 1) blocksPerCU = 1;
		if (const char* e = getenv("RAYLIB_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) blocksPerCU = v; }
		const int depthSlots = st.maxPathLength > 1 ? st.maxPathLength : 1;
		for (uint32_t s0 = 0; s0 < SPP; s0 += batch) {
			const uint32_t cnt = std::min(batch, SPP - s0);
			P.sampleBegin = s0; P.sampleCount = cnt;
			P.magicSamples = cnt > 1 ? (uint32_t)(0x100000000ull / cnt) : 0xFFFFFFFFu;
			const uint64_t jobs64 = (uint64_t)numActive * cnt * 64u;
			if (jobs64 > 0xF0000000ull) { Log("Raylib_Render: job count overflow"); return false; }   // (a head overshoots its band by one chunk per wave and attempt)
			P.numJobs = (uint32_t)jobs64;
			uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)R.numCUs * blocksPerCU, (jobs64 + RL_BLOCK * pathsPerThread - 1) / (RL_BLOCK * pathsPerThread));
			if (blocks < 1) blocks = 1;
			P.stackStride = blocks * RL_BLOCK * pathsPerThread;
			{   // jobs per global atomic: ~1/16 of a wave's share, rounded to a multiple of 64 (one cell at one sample), 64..1024.
				// Measured on the slice one of 8 ranks renders of the 1080p x 64 spp Cornell frame (16.6 M jobs): 64 -> 4.10 ms,
				// 128 -> 3.62, 256 -> 3.45, 512 -> 3.53, 1024 -> 4.07; on the whole frame 1024 is best (64 -> 33.6 ms: the atomic saturates).
				const uint64_t waves = (uint64_t)blocks * (RL_BLOCK / 64);
				uint64_t chunk = ((jobs64 / (waves * 16)) + 32) & ~63ull;
				if (const char* e = getenv("RAYLIB_JOB_CHUNK")) chunk = (uint64_t)std::max(0, atoi(e));
				P.jobChunk = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(64, chunk));
				// The pool schedule's jobs differ by two orders of magnitude (a cell that misses the scene's root box against one full of geometry): the launch's
				// tail is what a wave needs for its LAST chunk, and on the 298 k-triangle frame a chunk of 1024 heavy jobs is 6 ms of a 45 ms launch (wave timeline,
				// tools/gpu_timeline_pool.py: first wave out of jobs at 39.7 ms, last at 46.0).  One head took chunks no smaller than 1024 (256: 48.1 ms, the atomic's
				// queue); with a head per XCD 256 is the best: 1024 -> 46.5 ms, 512 -> 45.0, 256 -> 44.0, 128 -> 44.0, 64 -> 48.4 (one head, 1024: 44.9).
				if (poolK > 0 && !getenv("RAYLIB_JOB_CHUNK")) {
					uint32_t h = RL_MAX_HEADS; if (const char* e = getenv("RAYLIB_JOB_HEADS")) h = (uint32_t)std::max(1, atoi(e));
					if (h >= 4) P.jobChunk = std::min<uint32_t>(P.jobChunk, 256u);
					// ... and with the cells that cannot see the scene out of the list (CullCells) every job is a heavy one and there are far fewer of them: 128
					// as long as that stays under ~400 k draws per launch (298 k frame, 29 M jobs: 256 -> 37.75 ms, 128 -> 37.25, 64 -> 37.2, 512 -> 38.7)
					if (h >= 4 && numActive < numLocalCells && jobs64 / 128u <= 400000u) P.jobChunk = std::min<uint32_t>(P.jobChunk, 128u);
				}
				// the leaf-list kernel's chunk belongs to a workgroup, whose four waves draw batches of 64 from it (RL_QUEUE_SHARED_CHUNK): four waves' worth,
				// RAYLIB_JOB_CHUNK_MAX (default 1024) at most
				if (RL_QUEUE_SHARED_CHUNK && traceKernel == (TraceKernel)k_trace<16, false, true, 2> && !getenv("RAYLIB_JOB_CHUNK")) {
					uint64_t cap = 1024; if (const char* e = getenv("RAYLIB_JOB_CHUNK_MAX")) cap = (uint64_t)std::max(64, atoi(e));
					P.jobChunk = (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(256, 4 * chunk));
				}
				// a chunk is whole batches of 64 jobs (one cell at one sample: DecodeJobBatch decodes base >> 6, TakeJobs packs the head's number into the low
				// bits of a band's job count) whatever the environment asked for
				P.jobChunk = std::max(64u, P.jobChunk & ~63u);
			}
			if (!Grow(R.pathStack, R.pathStackBytes, (size_t)depthSlots * 8 * P.stackStride * sizeof(float))) return false;
			{   // the job list in bands of whole cells, one head per XCD (rl_render.hip TakeJobs); heads count from their band's first job: one memset
				uint32_t heads = RL_MAX_HEADS; if (const char* e = getenv("RAYLIB_JOB_HEADS")) heads = (uint32_t)std::min<int>((int)RL_MAX_HEADS, std::max(1, atoi(e)));
				const uint32_t cellsPerHead = (std::max(1u, numActive) + heads - 1) / heads;
				heads = (std::max(1u, numActive) + cellsPerHead - 1) / cellsPerHead;   // no empty band: every head's first job exists (and h * jobsPerHead < numJobs < 2^32)
				P.numHeads = heads; P.jobsPerHead = cellsPerHead * cnt * 64u;
				{   // guided draws at the end of a band: 2^shift ~ twice the drawers per head (waves; workgroups in the leaf-list kernel, whose chunk is shared)
					const bool perBlockChunk = RL_QUEUE_SHARED_CHUNK && traceKernel == (TraceKernel)k_trace<16, false, true, 2>;
					const uint32_t drawers = std::max(1u, blocks * (perBlockChunk ? 1u : (uint32_t)(RL_BLOCK / 64)) / heads);
					uint32_t shift = 1; while ((1u << shift) < 2u * drawers && shift < 24u) ++shift;
					int guided = 0;   // measured: no gain on either bench workload -- a heavy chunk drawn three rounds before the end outlasts the guided ones
					if (const char* e = getenv("RAYLIB_GUIDED")) guided = atoi(e);
					P.guideShift = guided > 0 ? shift + (uint32_t)(guided - 1) : 0u;
				}
				pend.jobHeads = heads;
				HIP_OK(hipMemsetAsync(R.jobCounter, 0, RL_MAX_HEADS * RL_HEAD_STRIDE * sizeof(unsigned int), R.stream));
			}
			HIP_OK(hipEventRecord(R.ev[q][2], R.stream));
			if (jobs64 > 0) {   // (no cell of this rank sees the scene: k_resolve has all it needs)
				hipLaunchKernelGGL(traceKernel, dim3(blocks), dim3(RL_BLOCK), 0, R.stream,
				                   P, traceView, DS->skyRot, R.samples, R.pathStack, R.counters, R.jobCounter);
				HIP_OK(hipGetLastError());
			}
			HIP_OK(hipEventRecord(R.ev[q][3], R.stream));
			const uint32_t rblocks = (numSlots + RL_BLOCK - 1) / RL_BLOCK;
			hipLaunchKernelGGL(k_resolve, dim3(rblocks), dim3(RL_BLOCK), 0, R.stream,
			                   P, traceView, DS->skyRot, R.samples, R.accum, out, (int)(s0 == 0), (int)(s0 + cnt >= SPP));
			HIP_OK(hipGetLastError());
			++pend.launches;
			if (s0 + cnt < SPP) {   // the event pair is reused by the next batch; the last batch's pair is read after the one final sync
				HIP_OK(hipEventSynchronize(R.ev[q][3]));
				float ms = 0.0f;
				HIP_OK(hipEventElapsedTime(&ms, R.ev[q][2], R.ev[q][3]));
				pend.traceMs += ms;
			} else pend.lastBatchPending = true;
		}
	}
	HIP_OK(hipEventRecord(R.ev[q][1], R.stream));
	HIP_OK(hipMemcpyAsync(R.cntHost[q], R.counters, (CNT_COUNT + 24) * sizeof(unsigned long long), hipMemcpyDeviceToHost, R.stream));
	HIP_OK(hipEventRecord(R.ev[q][7], R.stream));
	pend.enqueuedToEnd = true;
	uint64_t px = 0;
	for (uint32_t k = 0; k < numLocalCells; ++k) {
		const uint32_t cell = req.cellFirst + k * stride, cx = cell % cellsX, cy = cell / cellsX;
		px += (uint64_t)std::min(8u, W - cx * 8) * std::min(8u, H - cy * 8);
	}
	pend.pixels = px;
	if (!pathTrace) pend.listedCells = numLocalCells;
	return true;
}

bool EnqueueDispatch(RankCtx& R, Scene& sc, const RenderRequest& req, PendingRender& pend)
{
	const bool prims = !sc.spheres.empty() || !sc.cubes.empty();
	if (sc.bvh.depth <= 16 && !prims) return EnqueueRender<16, false>(R, sc, req, pend);
	if (sc.bvh.depth <= 32) return prims ? EnqueueRender<32, true>(R, sc, req, pend) : EnqueueRender<32, false>(R, sc, req, pend);
	if (sc.bvh.depth <= 64) return prims ? EnqueueRender<64, true>(R, sc, req, pend) : EnqueueRender<64, false>(R, sc, req, pend);
	Log("Raylib_Render: BVH depth %u exceeds the traversal stack (64)", sc.bvh.depth);
	return false;
}

// The one host synchronisation of a rank's render, then its numbers.  Adds to `stats` (counters are summed over ranks, times
// are the slowest rank's).
bool FinishRender(PendingRender& pend, RaylibAMDStats& stats)
{
	RankCtx& R = *pend.ctx;
	const int q = pend.slot;
	HIP_OK(hipSetDevice(R.device));
	if (!pend.enqueuedToEnd) {
		// the enqueue failed half-way: whatever it did queue is waited for (the stream is drained whatever happened), there are no numbers to report
		(void)hipStreamSynchronize(R.stream);
		return false;
	}
	HIP_OK(hipEventSynchronize(R.ev[q][7]));   // this render's last operation on the rank's stream (a later frame may already be queued behind it)
	if (pend.lastBatchPending) {
		float ms = 0.0f;
		HIP_OK(hipEventElapsedTime(&ms, R.ev[q][2], R.ev[q][3]));
		pend.traceMs += ms;
	}
	float totalMs = 0.0f;
	HIP_OK(hipEventElapsedTime(&totalMs, R.ev[q][0], R.ev[q][1]));
	const unsigned long long* cnt = pend.cnt;
	stats.rays += cnt[CNT_RAYS]; stats.nodesVisited += cnt[CNT_NODES]; stats.trisTested += cnt[CNT_TRIS];
	stats.shadedHits += cnt[CNT_SHADED]; stats.texFetches += cnt[CNT_TEXELS]; stats.cameraSamples += cnt[CNT_SAMPLES];
	// camera samples of the cells outside the scene's silhouette (CullCells) were never generated or traced: they are reported next to the executed work, not in it
	// (each would have been one root-box query, two with a sun -- what the megakernel counts for a sample it decides at the root)
	stats.culledCells += pend.culledCells; stats.listedCells += pend.listedCells;
	stats.culledSamples += pend.culledSamples; stats.culledRays += pend.culledSamples * pend.culledRaysPerSample;
	stats.texFetches += pend.culledSamples * pend.culledSkyTexels;   // (k_resolve DOES look up the sky texel of every sample of a dropped cell: executed, counted)
	stats.waveTrips += cnt[CNT_TRIPS];
	stats.pathsPerWave = 64u * pend.schedulePaths;
	stats.treeWidth = pend.treeWidth; stats.nodeBytes = pend.treeWidth == 8 ? (uint32_t)sizeof(DNode8) : 64u;
	stats.pixels += pend.pixels;
	stats.kernelMs = std::max(stats.kernelMs, (double)totalMs);
	stats.traceKernelMs = std::max(stats.traceKernelMs, (double)(pend.pathTrace ? pend.traceMs : totalMs));
	stats.traceLaunches = std::max(stats.traceLaunches, pend.pathTrace ? pend.launches : 1u);
	if (R.rank >= 0 && R.rank < 16) { stats.rankKernelMs[R.rank] = (double)totalMs; stats.rankTraceMs[R.rank] = (double)(pend.pathTrace ? pend.traceMs : totalMs); }
	if (pend.jobHeads) stats.jobHeads = pend.jobHeads;
#ifdef RL_DIAG_TIMELINE
	if (getenv("RAYLIB_PRINT_STAMPS")) {
		std::vector<unsigned long long> tl(RL_TIMELINE_SLOTS);
		HIP_OK(hipMemcpy(tl.data(), R.counters + CNT_COUNT + 24, tl.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
		std::vector<double> st, ex, en;
		unsigned long long t0 = ~0ull;
		for (int w = 0; w < 8192; ++w) if (tl[w] && tl[w] < t0) t0 = tl[w];
		for (int w = 0; w < 8192; ++w) if (tl[w]) { st.push_back((tl[w] - t0) * 0.01); if (tl[8192 + w]) ex.push_back((tl[8192 + w] - t0) * 0.01); en.push_back((tl[16384 + w] - t0) * 0.01); }
		auto pct = [](std::vector<double>& v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
		Log("timeline (us from the first wave's start; last launch, %d waves): start p50 %.1f max %.1f | queue seen empty min %.1f p50 %.1f max %.1f | end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f",
			(int)st.size(), pct(st, 0.5), pct(st, 1.0), pct(ex, 0.0), pct(ex, 0.5), pct(ex, 1.0), pct(en, 0.0), pct(en, 0.1), pct(en, 0.5), pct(en, 0.9), pct(en, 1.0));
		for (unsigned x = 0; x < 8; ++x) {   // per XCD: which waves ran there, when they found the job list empty, when they ended
			std::vector<double> xe, xn;
			for (int w = 0; w < 8192; ++w) if (tl[w] && tl[24576 + w] == x) { if (tl[8192 + w]) xe.push_back((tl[8192 + w] - t0) * 0.01); xn.push_back((tl[16384 + w] - t0) * 0.01); }
			if (!xn.empty()) Log("   XCC %u: %d waves | job list seen empty min %.1f p50 %.1f max %.1f | end min %.1f p50 %.1f max %.1f", x, (int)xn.size(), pct(xe, 0.0), pct(xe, 0.5), pct(xe, 1.0), pct(xn, 0.0), pct(xn, 0.5), pct(xn, 1.0));
		}
	}
#endif
#ifdef RL_DIAG_TOPN
	if (getenv("RAYLIB_PRINT_STAMPS")) {
		const double n = (double)cnt[CNT_NODES] + 1e-9;
		Log("node steps by node number (breadth first): < 9: %.3f  < 22: %.3f  < 53: %.3f  < 73: %.3f  < 128: %.3f  < 256: %.3f  < 1024: %.3f  beyond: %.3f (cumulative shares of %.0f steps)",
		    cnt[CNT_COUNT + 4] / n, (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5]) / n, (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5] + cnt[CNT_COUNT + 6]) / n,
		    (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5] + cnt[CNT_COUNT + 6] + cnt[CNT_COUNT + 7]) / n, (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5] + cnt[CNT_COUNT + 6] + cnt[CNT_COUNT + 7] + cnt[CNT_COUNT + 8]) / n,
		    (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5] + cnt[CNT_COUNT + 6] + cnt[CNT_COUNT + 7] + cnt[CNT_COUNT + 8] + cnt[CNT_COUNT + 9]) / n,
		    (cnt[CNT_COUNT + 4] + cnt[CNT_COUNT + 5] + cnt[CNT_COUNT + 6] + cnt[CNT_COUNT + 7] + cnt[CNT_COUNT + 8] + cnt[CNT_COUNT + 9] + cnt[CNT_COUNT + 10]) / n, cnt[CNT_COUNT + 11] / n, n);
		Log("groups on the stack at a node step: 0: %.3f  1: %.3f  2: %.3f  3: %.3f  4: %.3f  5: %.3f  6-7: %.3f  8+: %.3f", cnt[CNT_COUNT + 16] / n, cnt[CNT_COUNT + 17] / n, cnt[CNT_COUNT + 18] / n,
		    cnt[CNT_COUNT + 19] / n, cnt[CNT_COUNT + 20] / n, cnt[CNT_COUNT + 21] / n, cnt[CNT_COUNT + 22] / n, cnt[CNT_COUNT + 23] / n);
	}
#else
	if (getenv("RAYLIB_PRINT_STAMPS")) {
		const double tot = (double)(cnt[CNT_COUNT] + cnt[CNT_COUNT + 1] + cnt[CNT_COUNT + 2] + cnt[CNT_COUNT + 3]);
		Log("wave steps: node %llu (lane steps %llu, eff %.3f)  tri %llu (lane %llu, eff %.3f)  leaf rounds %llu  trips %llu", cnt[CNT_COUNT + 4], cnt[CNT_NODES], cnt[CNT_NODES] / (64.0 * cnt[CNT_COUNT + 4] + 1), cnt[CNT_COUNT + 5], cnt[CNT_TRIS], cnt[CNT_TRIS] / (64.0 * cnt[CNT_COUNT + 5] + 1), cnt[CNT_COUNT + 6], cnt[CNT_TRIPS]);
#if defined(RL_DIAG_STAMPS) && RL_DIAG_STAMPS >= 2
		if (cnt[CNT_COUNT + 7]) Log("leaf list: %llu triangle wave steps taken; %llu if every round's (ray, triangle) pairs were dealt evenly to the wave's 64 lanes (the bound of any regrouping: a round cannot take less than one step)", cnt[CNT_COUNT + 5], cnt[CNT_COUNT + 7]);
#endif
		Log("diagnostic slots (wave level): [4] %llu [5] %llu [6] %llu [7] %llu [16] %llu [17] %llu [18] %llu [19] %llu trips %llu", cnt[CNT_COUNT + 4], cnt[CNT_COUNT + 5], cnt[CNT_COUNT + 6], cnt[CNT_COUNT + 7],
		    cnt[CNT_COUNT + 16], cnt[CNT_COUNT + 17], cnt[CNT_COUNT + 18], cnt[CNT_COUNT + 19], cnt[CNT_TRIPS]);
		if (tot > 0) Log("shade split (of all): surface+material %.3f scatter %.3f emit+store %.3f", cnt[CNT_COUNT + 8] / tot, cnt[CNT_COUNT + 9] / tot, cnt[CNT_COUNT + 10] / tot);
		if (tot > 0) Log("microfacet split (of all): setup %.3f beckmann sample %.3f brdf+pdf %.3f | newton wave iters %llu lane iters %llu (eff %.3f) | microfacet wave calls %llu lanes %llu (eff %.3f)", cnt[CNT_COUNT + 12] / tot, cnt[CNT_COUNT + 13] / tot, cnt[CNT_COUNT + 14] / tot, cnt[CNT_COUNT + 16], cnt[CNT_COUNT + 17], cnt[CNT_COUNT + 17] / (64.0 * cnt[CNT_COUNT + 16] + 1), cnt[CNT_COUNT + 18], cnt[CNT_COUNT + 19], cnt[CNT_COUNT + 19] / (64.0 * cnt[CNT_COUNT + 18] + 1));
		{
			static const char* nm[4] = { "traverse", "shade a hit", "miss shader", "fold + store" };
			for (int k = 0; k < 4; ++k) if (cnt[CNT_COUNT + 4 + k] && cnt[CNT_COUNT + 20 + k])
				Log("  %-12s clock share %.3f, lanes taking part %.3f (level-1 diagnostic build)", nm[k], cnt[CNT_COUNT + 4 + k] / tot, cnt[CNT_COUNT + 20 + k] / (64.0 * cnt[CNT_COUNT + 4 + k]));
		}
		if (tot > 0) Log("phase shares (shader clock): refill %.3f traverse %.3f shade %.3f fold %.3f", cnt[CNT_COUNT] / tot, cnt[CNT_COUNT + 1] / tot, cnt[CNT_COUNT + 2] / tot, cnt[CNT_COUNT + 3] / tot);
	}
#endif
	return true;
}

// ---- whole frames over N ranks -------------------------------------------------------------------------------------
// cells round-robin, one gather to rank 0's device, one scatter kernel -- and up to two frames in flight: Raylib_Render returns when frame i
// is ENQUEUED on every rank's stream (after waiting for frame i - 1's predecessor, whose buffers frame i reuses), so that while rank 0's gather
// stream still receives and assembles frame i the ranks already render frame i + 1.  Whoever reads the pixels, the stats or changes anything
// a frame in flight uses goes through DrainLocked() first: to a front-end the call is as synchronous as the reference's (raylib.cc:231-239),
// it just finds out later.  RAYLIB_PIPELINE=0 waits at the end of every call.
struct Runtime::Inflight {
	std::vector<PendingRender> pend;
	RaylibAMDStats stats;                 // what is known when the frame is enqueued; FinishInflight adds counters and times
	int slot = 0;
	bool ok = true, timed = false;
	std::chrono::steady_clock::time_point t0;
};
RaylibAMDStats g_deferredStats;           // of the last frame FinishInflight completed
bool g_deferredUnreported = false;

bool FinishInflight(int slot)
{
	Runtime& R = g_rt;
	Runtime::Inflight* F = R.inflight[slot];
	if (!F) return true;
	R.inflight[slot] = nullptr;
	RankCtx& R0 = Rank0();
	bool ok = F->ok;
	for (int r = (int)F->pend.size() - 1; r >= 0; --r) {
		if (F->pend[(size_t)r].ctx) ok = FinishRender(F->pend[(size_t)r], F->stats) && ok;
		else { (void)hipSetDevice(R.ranks[(size_t)r]->device); (void)hipStreamSynchronize(R.ranks[(size_t)r]->stream); }
	}
	(void)hipSetDevice(R0.device);
	if (F->timed) {
		if (hipEventSynchronize(R0.ev[slot][6]) != hipSuccess) ok = false;   // frame assembled (and copied to the host, if asked for)
		float g = 0.0f, sc = 0.0f;
		if (hipEventElapsedTime(&g, R0.ev[slot][1], R0.ev[slot][5]) == hipSuccess) F->stats.gatherMs = (double)g;
		if (hipEventElapsedTime(&sc, R0.ev[slot][5], R0.ev[slot][6]) == hipSuccess) F->stats.scatterMs = (double)sc;
	} else if (R.gatherStream) (void)hipStreamSynchronize(R.gatherStream);
	F->stats.wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - F->t0).count();
	g_deferredStats = F->stats; g_deferredUnreported = true;
	delete F;
	return ok;
}
// waits for every frame in flight (oldest first); the runtime lock is held by the caller
bool DrainLocked()
{
	Runtime& R = g_rt;
	if (!R.ok) return true;
	const int older = (int)(R.frameNo & 1);   // the slot the NEXT frame would take holds the older of two frames in flight
	bool ok = FinishInflight(older);
	return FinishInflight(older ^ 1) && ok;
}

bool RenderMulti(Scene& sc, const RenderRequest& req, RaylibAMDStats& stats, bool& deferred)
{
	Runtime& R = g_rt;
	const int N = (int)R.ranks.size();
	RankCtx& R0 = Rank0();
	const int b = (int)(R.frameNo & 1);
	deferred = false;
	// this slot's previous frame (two calls ago) gives up its buffers, events and counter blocks
	bool ok = FinishInflight(b);
	const uint32_t W = req.settings.viewportWidth, H = req.settings.viewportHeight;
	const uint32_t cellsX = (W + 7) / 8, numCells = cellsX * ((H + 7) / 8);
	ScatterPlan plan; memset(&plan, 0, sizeof(plan));
	plan.ranks = (uint32_t)N;
	std::vector<uint32_t> local((size_t)N);
	uint32_t total = 0;
	for (int r = 0; r < N; ++r) {
		local[(size_t)r] = (uint32_t)r < numCells ? (numCells - (uint32_t)r + (uint32_t)N - 1) / (uint32_t)N : 0;
		plan.offset[r] = total * 64u;
		total += local[(size_t)r];
	}
	HIP_OK(hipSetDevice(R0.device));
	if (!Grow(R.gather[b], R.gatherBytes[b], std::max<size_t>(16, (size_t)total * 64 * sizeof(float4)))) return false;
	float4* out = (float4*)req.outDevice;
	const size_t frameBytes = (size_t)W * H * sizeof(float4);
	if (!out) {
		if (frameBytes > R0.imageBytes) ok = FinishInflight(b ^ 1) && ok;   // the frame in flight may be writing the library's own image: not while it is re-allocated
		if (!Grow(R0.image, R0.imageBytes, frameBytes)) return false;
		out = R0.image;
	}

	// which ranks need a copy: those on another device than rank 0 (and rank 0 itself under RAYLIB_GATHER_SELF)
	std::vector<char> remote((size_t)N, 0);
	bool anyRemote = false;
	for (int r = 0; r < N; ++r) { remote[(size_t)r] = (R.ranks[(size_t)r]->device != R0.device) || (r == 0 && R.gatherSelf); anyRemote = anyRemote || remote[(size_t)r]; }
	const bool useRccl = anyRemote && R.wantRccl && EnsureRccl();

	// From here on work is enqueued that only FinishInflight waits for: no early return -- a failing call clears `ok` (HIP_TRY) and the function still
	// reaches the place that registers the frame and drains every stream.
	Runtime::Inflight* F = new Runtime::Inflight;
	F->pend.resize((size_t)N);
	F->slot = b; F->t0 = std::chrono::steady_clock::now();
	memset(&F->stats, 0, sizeof(F->stats));
	std::vector<PendingRender>& pend = F->pend;
	float4* gather = R.gather[b];
	auto run = [&](int r) -> bool {
		RankCtx& C = *R.ranks[(size_t)r];
		HIP_OK(hipSetDevice(C.device));
		RenderRequest q = req;
		q.cellFirst = (uint32_t)r; q.cellStride = (uint32_t)N; q.cellMajor = true; q.outHostRGBA = nullptr; q.slot = b;
		const size_t bytes = (size_t)local[(size_t)r] * 64 * sizeof(float4);
		float4* dst = gather + plan.offset[r];
		if (remote[(size_t)r]) { if (!Grow(C.cells, C.cellsBytes, std::max<size_t>(16, bytes))) return false; q.outDevice = C.cells; }
		else q.outDevice = dst;   // same device as rank 0: rendered in place, nothing to move
		if (!EnqueueDispatch(C, sc, q, pend[(size_t)r])) return false;
		if (remote[(size_t)r] && !useRccl && bytes) HIP_OK(hipMemcpyPeerAsync(dst, R0.device, C.cells, C.device, bytes, C.stream));
		HIP_OK(hipEventRecord(C.ev[b][4], C.stream));
		return true;
	};
	for (int r = 1; r < N; ++r) R.ranks[(size_t)r]->worker->Post([&run, r]() { return run(r); });
	ok = run(0) && ok;
	for (int r = 1; r < N; ++r) ok = R.ranks[(size_t)r]->worker->Wait() && ok;
	HIP_TRY(hipSetDevice(R0.device));
	if (ok && useRccl) {
		// one group: rank 0's GATHER stream receives every remote rank's cells, each remote rank's stream sends them (behind its kernels)
		RcclApi& A = R.rccl;
		// One stream per communicator inside the group: all ranks of a device send on the stream of that device's FIRST rank (the lead), which waits
		// for the others' kernels; rank 0 sending to itself (RAYLIB_GATHER_SELF, tests) sends and receives on the gather stream.  Afterwards the
		// other ranks' streams wait for the lead's sends, so that their next frame does not overwrite cells that are still being sent.
		std::vector<int> lead(R.devices.size(), -1);
		for (int r = 0; r < N; ++r) if (remote[(size_t)r] && local[(size_t)r] && lead[(size_t)R.ranks[(size_t)r]->devSlot] < 0) lead[(size_t)R.ranks[(size_t)r]->devSlot] = r;
		auto sendStream = [&](const RankCtx& C) { return C.devSlot == R0.devSlot ? R.gatherStream : R.ranks[(size_t)lead[(size_t)C.devSlot]]->stream; };
		for (int r = 0; r < N; ++r) {
			RankCtx& C = *R.ranks[(size_t)r];
			if (!remote[(size_t)r] || !local[(size_t)r] || lead[(size_t)C.devSlot] == r) continue;
			(void)hipSetDevice(C.device);
			HIP_TRY(hipStreamWaitEvent(sendStream(C), C.ev[b][4], 0));
		}
		if (remote[0] && local[0]) { (void)hipSetDevice(R0.device); HIP_TRY(hipStreamWaitEvent(R.gatherStream, R0.ev[b][4], 0)); }
		int rc = A.GroupStart();
		for (int r = 0; r < N && rc == 0; ++r) {
			if (!remote[(size_t)r] || !local[(size_t)r]) continue;
			RankCtx& C = *R.ranks[(size_t)r];
			const size_t floats = (size_t)local[(size_t)r] * 64 * 4;
			rc = A.Recv(gather + plan.offset[r], floats, kRcclFloat, C.devSlot, A.comms[(size_t)R0.devSlot], R.gatherStream);
			if (rc == 0) rc = A.Send(C.cells, floats, kRcclFloat, R0.devSlot, A.comms[(size_t)C.devSlot], sendStream(C));
		}
		const int rcEnd = A.GroupEnd();
		if (rc != 0 || rcEnd != 0) { Log("Raylib_Render: RCCL gather failed (%s)", A.GetErrorString ? A.GetErrorString(rc ? rc : rcEnd) : "?"); ok = false; }
		for (size_t sl = 0; sl < lead.size() && ok; ++sl) {
			if (lead[sl] < 0 || (int)sl == R0.devSlot) continue;
			RankCtx& L = *R.ranks[(size_t)lead[sl]];
			(void)hipSetDevice(L.device);
			HIP_TRY(hipEventRecord(L.ev[b][5], L.stream));   // (slots 5 and 6 belong to rank 0 on ITS device; a lead of another device uses its own 5 for "sends done")
			for (int r = 0; r < N; ++r) if (r != lead[sl] && remote[(size_t)r] && R.ranks[(size_t)r]->devSlot == (int)sl) HIP_TRY(hipStreamWaitEvent(R.ranks[(size_t)r]->stream, L.ev[b][5], 0));
		}
		(void)hipSetDevice(R0.device);
	}
	if (ok) {
		// the gather stream waits for every rank's "my cells are there" (rank 0's own render included), assembles the frame, and rank 0's
		// render stream is free for the next frame meanwhile
		for (int r = 0; r < N; ++r) HIP_TRY(hipStreamWaitEvent(R.gatherStream, R.ranks[(size_t)r]->ev[b][4], 0));
		HIP_TRY(hipEventRecord(R0.ev[b][5], R.gatherStream));
		const uint32_t blocks = (uint32_t)(((size_t)W * H + RL_BLOCK - 1) / RL_BLOCK);
		hipLaunchKernelGGL(k_scatter_cells, dim3(blocks), dim3(RL_BLOCK), 0, R.gatherStream, (const float4*)gather, out, W, H, cellsX, plan);
		HIP_TRY(hipGetLastError());
		if (req.outHostRGBA) HIP_TRY(hipMemcpyAsync(req.outHostRGBA, out, frameBytes, hipMemcpyDeviceToHost, R.gatherStream));
		HIP_TRY(hipEventRecord(R0.ev[b][6], R.gatherStream));
		if (remote[0] && useRccl) HIP_TRY(hipStreamWaitEvent(R0.stream, R0.ev[b][5], 0));   // rank 0's self-send has read its cell buffer before the next render writes it
		F->timed = ok;   // (events 5 and 6 are only read when both were recorded)
	}
	F->ok = ok;
	F->stats.ranks = (uint32_t)N; F->stats.devices = (uint32_t)R.devices.size();
	F->stats.gatherMode = !anyRemote ? 0u : (useRccl ? 1u : 2u);
	F->stats.rcclCommSize = R.rccl.ok ? (uint32_t)R.rccl.comms.size() : 0u;
	F->stats.numNodes = (uint32_t)sc.bvh.nodes.size(); F->stats.numTriangles = (uint32_t)sc.triangles.size(); F->stats.bvhDepth = sc.bvh.depth;
	R.inflight[b] = F;
	++R.frameNo;
	if (!ok || !R.pipeline || req.outHostRGBA || req.callerOwnsOut) {
		// synchronous after all: a failure (every stream is drained whatever happened), the switch, pixels wanted in host memory now, or a frame
		// into device memory of the caller's (RaylibAMD_RenderDevice: "the library's stream has been synchronised when it returns" -- nothing the
		// library owns would keep a reader or a free of that buffer behind the gather stream's scatter)
		ok = DrainLocked() && ok;
		stats = g_deferredStats; g_deferredUnreported = false;
		return ok;
	}
	stats = F->stats;   // counters and times follow when the frame is waited for (RaylibAMD_GetLastStats, any reader of the pixels, the call after next)
	deferred = true;
	return ok;
}

} // namespace

bool DeviceAvailable()
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	return EnsureRuntime();
}

int DeviceNumRanks()
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	return EnsureRuntime() ? (int)g_rt.ranks.size() : 0;
}

bool DeviceRender(Scene& sc, const RenderRequest& req, RaylibAMDStats& stats)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	const auto t0 = std::chrono::steady_clock::now();
	if (!EnsureRuntime()) return false;
	HIP_OK(hipSetDevice(Rank0().device));
	if (!UploadScene(sc)) return false;
	if (!SyncSky(sc)) return false;
	bool ok;
	const bool whole = req.cellFirst == 0 && (req.cellStride == 0 || req.cellStride == 1);
	if (whole && (g_rt.ranks.size() > 1 || g_rt.gatherSelf)) {
		bool deferred = false;
		ok = RenderMulti(sc, req, stats, deferred);
		if (deferred) return ok;    // scene numbers are in; counters, times and wallMs follow at the drain
	} else {
		(void)DrainLocked();        // this path uses rank 0's slot-0 events and counter block
		g_deferredUnreported = false;   // ... and its numbers are the ones the caller reads next
		PendingRender pend;
		ok = EnqueueDispatch(Rank0(), sc, req, pend);
		if (ok && req.outHostRGBA) HIP_OK(hipMemcpyAsync(req.outHostRGBA, pend.out, pend.outBytes, hipMemcpyDeviceToHost, Rank0().stream));
		if (pend.ctx) ok = FinishRender(pend, stats) && ok;
		else (void)hipStreamSynchronize(Rank0().stream);
		stats.ranks = 1; stats.devices = 1;
	}
	stats.numNodes = (uint32_t)sc.bvh.nodes.size(); stats.numTriangles = (uint32_t)sc.triangles.size(); stats.bvhDepth = sc.bvh.depth;
	if (!(whole && (g_rt.ranks.size() > 1 || g_rt.gatherSelf))) stats.wallMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return ok;
}

// Waits for whatever Raylib_Render left in flight.  True with `out` filled when that completed the LAST render call's numbers (counters, times)
// that the call itself could not report yet.
bool DeviceDrain(RaylibAMDStats* out)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!g_rt.ok) return false;
	(void)DrainLocked();
	if (!g_deferredUnreported) return false;
	g_deferredUnreported = false;
	if (out) *out = g_deferredStats;
	return true;
}

bool DeviceClosestHit(Scene& sc, const float* rays, int32_t n, float tMin, void* outHits)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!EnsureRuntime()) return false;
	RankCtx& R = Rank0();
	HIP_OK(hipSetDevice(R.device));
	if (sc.bvh.depth > 64) { Log("RaylibAMD_ClosestHit: BVH depth %u exceeds the traversal stack (64)", sc.bvh.depth); return false; }
	if (!UploadScene(sc)) return false;
	if (n <= 0) return true;
	float* dRays = nullptr; DHitOut* dOut = nullptr;
	HIP_OK(hipMalloc(&dRays, (size_t)n * 6 * sizeof(float)));
	if (hipMalloc(&dOut, (size_t)n * sizeof(DHitOut)) != hipSuccess) { (void)hipFree(dRays); Log("RaylibAMD_ClosestHit: out of device memory"); return false; }
	const DSceneView& view = sc.device->copy[(size_t)R.devSlot]->view;
	bool ok = hipMemcpy(dRays, rays, (size_t)n * 6 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		const uint32_t blocks = ((uint32_t)n + RL_BLOCK - 1) / RL_BLOCK;
		if (sc.bvh.depth <= 32) hipLaunchKernelGGL((k_closest_hit<32, true>), dim3(blocks), dim3(RL_BLOCK), 0, R.stream, view, dRays, n, tMin, dOut);
		else hipLaunchKernelGGL((k_closest_hit<64, true>), dim3(blocks), dim3(RL_BLOCK), 0, R.stream, view, dRays, n, tMin, dOut);
		ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(R.stream) == hipSuccess;
		ok = ok && hipMemcpy(outHits, dOut, (size_t)n * sizeof(DHitOut), hipMemcpyDeviceToHost) == hipSuccess;
	}
	(void)hipFree(dRays); (void)hipFree(dOut);
	if (!ok) Log("RaylibAMD_ClosestHit: a HIP call failed");
	return ok;
}

// kind 0: scatter (in 16 / out 16 floats per record, a = material), 1: camera rays (in 2 / out 7), 2: texture (in 2 / out 4, a = texture, b = sRGB)
bool DeviceEvalHook(int kind, Scene* sc, const DCamera* cam, int a, int b, const float* in, int n, uint64_t seed, float* out)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!EnsureRuntime()) return false;
	RankCtx& R = Rank0();
	HIP_OK(hipSetDevice(R.device));
	if (sc && !UploadScene(*sc)) return false;
	if (n <= 0) return true;
	const int inW = kind == 0 ? 16 : 2, outW = kind == 0 ? 16 : (kind == 1 ? 7 : 4);
	float *din = nullptr, *dout = nullptr;
	HIP_OK(hipMalloc(&din, (size_t)n * inW * 4));
	if (hipMalloc(&dout, (size_t)n * outW * 4) != hipSuccess) { (void)hipFree(din); return false; }
	bool ok = hipMemcpy(din, in, (size_t)n * inW * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		const dim3 grid(((uint32_t)n + RL_BLOCK - 1) / RL_BLOCK), block(RL_BLOCK);
		if (kind == 0) hipLaunchKernelGGL(k_eval_scatter, grid, block, 0, R.stream, sc->device->copy[(size_t)R.devSlot]->view, a, din, n, (unsigned long long)seed, dout);
		else if (kind == 1) hipLaunchKernelGGL(k_eval_camera, grid, block, 0, R.stream, *cam, din, n, (unsigned long long)seed, dout);
		else hipLaunchKernelGGL(k_eval_texture, grid, block, 0, R.stream, sc->device->copy[(size_t)R.devSlot]->view, a, b, din, n, dout);
		ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(R.stream) == hipSuccess;
		ok = ok && hipMemcpy(out, dout, (size_t)n * outW * 4, hipMemcpyDeviceToHost) == hipSuccess;
	}
	(void)hipFree(din); (void)hipFree(dout);
	return ok;
}

bool DeviceEvalMath(int fn, const float* x, const float* y, int n, float* out)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!EnsureRuntime()) return false;
	RankCtx& R = Rank0();
	HIP_OK(hipSetDevice(R.device));
	if (n <= 0) return true;
	float *dx = nullptr, *dy = nullptr, *dout = nullptr;
	bool ok = hipMalloc(&dx, (size_t)n * 4) == hipSuccess && hipMalloc(&dout, (size_t)n * 4) == hipSuccess;
	ok = ok && hipMemcpy(dx, x, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok && y) ok = hipMalloc(&dy, (size_t)n * 4) == hipSuccess && hipMemcpy(dy, y, (size_t)n * 4, hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		hipLaunchKernelGGL(k_eval_math, dim3(((uint32_t)n + RL_BLOCK - 1) / RL_BLOCK), dim3(RL_BLOCK), 0, R.stream, fn, dx, dy, n, dout);
		ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(R.stream) == hipSuccess;
		ok = ok && hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess;
	}
	if (dx) (void)hipFree(dx); if (dout) (void)hipFree(dout); if (dy) (void)hipFree(dy);
	return ok;
}

bool DeviceVerifyExactMath(int which, uint64_t* outMismatches, uint64_t* outFirst)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!EnsureRuntime()) return false;
	RankCtx& R = Rank0();
	HIP_OK(hipSetDevice(R.device));
	unsigned long long* d = nullptr;
	HIP_OK(hipMalloc(&d, 2 * sizeof(unsigned long long)));
	unsigned long long h[2] = { 0ull, ~0ull };
	bool ok = hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) == hipSuccess;
	if (ok) {
		hipLaunchKernelGGL(k_verify_exact_math, dim3(4096), dim3(RL_BLOCK), 0, R.stream, which, d);
		ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(R.stream) == hipSuccess && hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
	}
	(void)hipFree(d);
	if (outMismatches) *outMismatches = h[0];
	if (outFirst) *outFirst = h[1];
	return ok;
}

// Images live on rank 0's device.
void* DeviceImagePixels(Image& img)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (!EnsureRuntime()) return nullptr;
	const size_t need = (size_t)img.width * img.height * sizeof(float4);
	if (need == 0) return nullptr;
	if (img.devPixels && img.devBytes >= need) return img.devPixels;
	(void)DrainLocked();   // a frame in flight may be writing the buffer that is about to be freed
	if (hipSetDevice(Rank0().device) != hipSuccess) return nullptr;
	if (img.devPixels) { (void)hipFree(img.devPixels); img.devPixels = nullptr; img.devBytes = 0; }
	if (hipMalloc(&img.devPixels, need) != hipSuccess) { img.devPixels = nullptr; return nullptr; }
	img.devBytes = need;
	img.devValid = false;
	return img.devPixels;
}

bool DeviceReadback(Image& img)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	return ReadbackLocked(img);
}
namespace {
bool ReadbackLocked(Image& img)
{
	const size_t n = (size_t)img.width * img.height;
	if (n == 0) return true;
	if (!g_rt.ok || !img.devPixels || !img.devValid || img.devBytes < n * sizeof(float4)) return false;
	if (!DrainLocked()) Log("Image read-back: a frame in flight did not complete");
	HIP_OK(hipSetDevice(Rank0().device));
	img.rgba.resize(n * 4);
	HIP_OK(hipMemcpyAsync(img.rgba.data(), img.devPixels, n * sizeof(float4), hipMemcpyDeviceToHost, Rank0().stream));
	HIP_OK(hipStreamSynchronize(Rank0().stream));
	return true;
}
} // namespace

// Raylib_DumpImageData for a frame that lives on the device (reference raylib.h:90-93: the caller's buffer of 3 * 4 * w * h bytes; in the reference the pixels
// are in host memory when Raylib_Render returns, renderer.cc:292-296, so this copy is part of what a front-end sees of a render).  k_pack_rgb packs RGB on
// the device; the packed frame crosses the bus in chunks into PINNED staging memory (a pageable destination makes the runtime stage the copy through its own
// small buffers: 12 GB/s measured in round 3); helper threads copy every chunk on to the caller's memory as soon as its event has fired, so the host copies
// run beside the rest of the transfer.  img.rgba stays stale: another reader fetches its own copy (Image::SyncHost).
namespace {
struct DumpStage {
	float* packed = nullptr; size_t packedBytes = 0;     // device: RGB, 12 bytes per pixel
	float* pinned = nullptr; size_t pinnedBytes = 0;     // host, page-locked
	hipEvent_t ev[16] = {};
	bool evReady = false;
	std::vector<Worker*> helpers;
};
DumpStage g_dump;
}
bool DeviceDumpRGB(Image& img, float* outDest)
{
	std::lock_guard<std::mutex> lk(g_rt.lock);
	const size_t n = (size_t)img.width * img.height;
	if (n == 0) return true;
	if (!g_rt.ok || !img.devPixels || !img.devValid || img.devBytes < n * sizeof(float4)) return false;
	if (const char* e = getenv("RAYLIB_FAST_DUMP")) if (atoi(e) == 0) return false;   // the plain path: read RGBA back, pack on the host
	if (!DrainLocked()) Log("Raylib_DumpImageData: a frame in flight did not complete");
	RankCtx& R = Rank0();
	DumpStage& D = g_dump;
	HIP_OK(hipSetDevice(R.device));
	const size_t bytes = n * 3 * sizeof(float);
	if (!Grow(D.packed, D.packedBytes, (bytes + 15) & ~(size_t)15)) return false;
	if (D.pinnedBytes < bytes) {
		if (D.pinned) { (void)hipHostFree(D.pinned); D.pinned = nullptr; D.pinnedBytes = 0; }
		HIP_OK(hipHostMalloc((void**)&D.pinned, bytes, hipHostMallocDefault));
		D.pinnedBytes = bytes;
	}
	if (!D.evReady) { for (hipEvent_t& e : D.ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); D.evReady = true; }
	const uint32_t blocks = (uint32_t)(((n + 3) / 4 + RL_BLOCK - 1) / RL_BLOCK);
	hipLaunchKernelGGL(k_pack_rgb, dim3(blocks), dim3(RL_BLOCK), 0, R.stream, (const float4*)img.devPixels, (float4*)D.packed, D.packed, n);
	HIP_OK(hipGetLastError());
	// chunks of ~2 MB (at most 16): the first host copy starts after 1/chunks of the transfer, the last one is all that is left when the transfer ends
	const size_t chunks = std::max<size_t>(1, std::min<size_t>(16, bytes / (2u << 20)));
	const size_t per = ((bytes / chunks) + 63) & ~(size_t)63;
	size_t nChunks = 0;
	for (size_t off = 0; off < bytes; off += per, ++nChunks) {
		const size_t len = std::min(per, bytes - off);
		HIP_OK(hipMemcpyAsync((char*)D.pinned + off, (const char*)D.packed + off, len, hipMemcpyDeviceToHost, R.stream));
		HIP_OK(hipEventRecord(D.ev[nChunks], R.stream));
	}
	// helper threads: chunk c is copied by helper c % (helpers + 1), the calling thread takes its share too
	const size_t wantHelpers = nChunks > 1 ? std::min<size_t>(3, std::max(1u, std::thread::hardware_concurrency()) - 1) : 0;
	while (D.helpers.size() < wantHelpers) { Worker* w = new Worker; w->Start(R.device); D.helpers.push_back(w); }
	const size_t lanes = wantHelpers + 1;
	auto copyLane = [&D, outDest, per, bytes, nChunks, lanes](size_t lane) -> bool {
		bool ok = true;
		for (size_t c = lane; c < nChunks; c += lanes) {
			if (hipEventSynchronize(D.ev[c]) != hipSuccess) { ok = false; break; }
			const size_t off = c * per;
			memcpy((char*)outDest + off, (const char*)D.pinned + off, std::min(per, bytes - off));
		}
		return ok;
	};
	for (size_t h = 0; h < wantHelpers; ++h) D.helpers[h]->Post([&copyLane, h]() { return copyLane(h + 1); });
	bool ok = copyLane(0);
	for (size_t h = 0; h < wantHelpers; ++h) ok = D.helpers[h]->Wait() && ok;
	if (!ok) { Log("Raylib_DumpImageData: the device copy failed"); (void)hipStreamSynchronize(R.stream); return false; }
	return true;
}

void DeviceFreePixels(void* p)
{
	if (!p) return;
	std::lock_guard<std::mutex> lk(g_rt.lock);
	if (g_rt.ok) { (void)DrainLocked(); (void)hipSetDevice(Rank0().device); (void)hipFree(p); }
}

// Image2D::PostProcess (reference render/image.cc:44-103) on the device: k_pp_max finds the white point
// (max is exact in any order), k_pp_map applies extended Reinhard on luminance, the clamp and gamma 1/2.2
// with glibc's exact powf.  Bit-identical to the host statement of the same function.
bool DevicePostProcess(Image& img)
{
	const size_t n = (size_t)img.width * img.height;
	if (n == 0) return true;
	float4* px = (float4*)DeviceImagePixels(img);
	if (!px) return false;
	std::lock_guard<std::mutex> lk(g_rt.lock);
	(void)DrainLocked();   // the frame this works on may still be in flight on the gather stream
	RankCtx& R = Rank0();
	HIP_OK(hipSetDevice(R.device));
	if (!img.devValid) HIP_OK(hipMemcpyAsync(px, img.rgba.data(), n * sizeof(float4), hipMemcpyHostToDevice, R.stream));   // (never stale here: stale implies devValid)
	unsigned int one; { float f = 1.0f; memcpy(&one, &f, 4); }
	HIP_OK(hipMemcpyAsync(R.jobCounter, &one, sizeof(one), hipMemcpyHostToDevice, R.stream));
	const uint32_t blocks = (uint32_t)std::min<size_t>((n + RL_BLOCK - 1) / RL_BLOCK, 2048);
	hipLaunchKernelGGL(k_pp_max, dim3(blocks), dim3(RL_BLOCK), 0, R.stream, px, n, R.jobCounter);
	HIP_OK(hipGetLastError());
	hipLaunchKernelGGL(k_pp_map, dim3((uint32_t)((n + RL_BLOCK - 1) / RL_BLOCK)), dim3(RL_BLOCK), 0, R.stream, px, n, R.jobCounter);
	HIP_OK(hipGetLastError());
	float white = 1.0f;
	HIP_OK(hipMemcpyAsync(&white, R.jobCounter, 4, hipMemcpyDeviceToHost, R.stream));
	HIP_OK(hipStreamSynchronize(R.stream));
	img.devValid = true;
	img.hostStale = true;   // read back when the pixels are asked for (Image::SyncHost)
	img.Touch();
	Log("Max white luminance: %f", white);
	return true;
}

void DeviceReleaseScene(DeviceScene* D)
{
	if (!D) return;
	std::lock_guard<std::mutex> lk(g_rt.lock);
	(void)DrainLocked();   // frames in flight read this scene
	FreeScene(D);
}

} // namespace rl
