// C ABI of libpt_hip.so (include/pt_hip.h): context, scene upload, frame loop.  This is the
// replacement for `class VulkanRaytracing` (VulkanRaytracing.h:88-117) minus everything that only
// exists because of Vulkan (descriptor sets, pipeline, shader binding table): kernel arguments
// take the place of descriptor sets, and the frame counter m_pcRay.frame lives in pt_ctx.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "pt_build.h"
#include "pt_trace.h"

extern const char *pt_seam_last_error();

namespace
{
thread_local std::string g_error;

// One bounce of a frame under PT_PROFILE_KERNELS: begin -> (extend) -> mid -> (shade) -> end.
struct EventPair
{
	hipEvent_t begin, mid, end;
};
}  // namespace

// ---- scratch of the allocation-free entry point pt_render_frames_noalloc (include/pt_hip.h) ----
// What pt_render_frames allocates on demand (the LCG skip-ahead table, the per-path state and radiance records, the
// wave-private survivor rings), at a fixed maximum size, in a file-scope device array: every GPU the library runs on holds
// its own copy at its own address until the process ends.  Host code never uses the array's name as an address: pt_create
// asks hipGetSymbolAddress once per device (g_scratchPtr) and everything below works with that pointer.
#define PT_NOALLOC_WAVES 1024u                 // single-wave workgroups of the launch (4 per CU)
#define PT_NOALLOC_MAX_RECORDS (1u << 18)      // paths per launch: ceil(tile pixels / 64) * 64 * spp_per_frame
#define PT_NOALLOC_MAX_SPP 256u
namespace
{
constexpr size_t kNaRingBytes = (size_t)PT_NOALLOC_WAVES * PT_PATHS_RING * sizeof(float4);  // 4 MiB
constexpr size_t kNaRecBytes = (size_t)PT_NOALLOC_MAX_RECORDS * sizeof(float4);             // 4 MiB
constexpr size_t kNaOffRingO = 0, kNaOffRingD = kNaRingBytes, kNaOffState = 2 * kNaRingBytes,
                 kNaOffRad = 2 * kNaRingBytes + kNaRecBytes, kNaOffSkip = 2 * kNaRingBytes + 2 * kNaRecBytes;
constexpr size_t kNaOffLit = kNaOffSkip + PT_NOALLOC_MAX_SPP * sizeof(uint2);               // one bit per path: 32 KiB
constexpr size_t kNaBytes = kNaOffLit + PT_NOALLOC_MAX_RECORDS / 8;                         // 16 MiB + 34 KiB
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[kNaBytes];
namespace
{
std::mutex g_scratchMutex;
unsigned char *g_scratchPtr[kMaxDevices] = {};  // device address of `scratch` on each device, set by pt_create
}  // namespace

struct pt_ctx
{
	int device = 0;
	hipStream_t ownStream = nullptr, stream = nullptr;
	int numCUs = 256;
	std::string error;

	// scene (reference layout) + acceleration structure
	PtVertex *dVerts = nullptr;
	uint32_t *dIndices = nullptr, *dMatIdx = nullptr;
	PtMaterial *dMats = nullptr;
	uint8_t *dPrimEmissive = nullptr;
	PtShadeRec *dShade = nullptr;  // per-primitive shading records (pt_device_common.h)
	uint32_t nVerts = 0, nIdx = 0, nMats = 0;
	bool haveScene = false, haveAccel = false, brute = false;
	float4 *dTexture = nullptr;  // optional diffuse map (pt_set_texture), linear RGBA
	uint32_t texW = 0, texH = 0;
	// one map per Model (pt_set_model_texture): texels of model m (nullptr: none) and the device table the shading searches
	struct ModelTex
	{
		float4 *texels = nullptr;
		uint32_t w = 0, h = 0;
	};
	std::vector<ModelTex> modelTex;
	PtTexSlot *dTexSlots = nullptr;
	uint32_t nTexSlots = 0;
	PtBuildOutput accel;
	PtQuadOutput quadAccel;  // quad leaves + the node array that refers to them (persistent pipeline only)
	uint4 *dEntry = nullptr;  // entry node of the persistent pipeline (flattened tree top, pt_wave_traverse.h) or nullptr
	bool entryFlat = false;   // the entry node holds every leaf of the scene
	double buildMs = 0.0;
	uint64_t accelBytes = 0, accelBytesUncompacted = 0;  // acceleration-structure memory after / before compaction
	// two-level scenes (pt_set_instances): 24 floats per instance (object-to-world 3x4, world-to-object 3x4)
	std::vector<float> instances;
	float *dInstances = nullptr;
	uint32_t nInst = 0, tlasRoot = 0, blasNodes = 0, tlasDepth = 0;
	// several Models (pt_upload_models): the buffers above hold them one after the other; model m owns the triangles
	// [modelTriFirst[m], + modelTriCount[m]) and, after pt_build_accel, the bottom-level tree rooted at node modelRoot[m]
	std::vector<uint32_t> modelTriFirst{ 0u }, modelTriCount{ 0u }, modelRoot{ 0u };
	std::vector<uint32_t> instModel;  // custom_index of every instance
	bool implicitInst = false;        // the instance list is pt_build_accel's own (one identity instance per Model)
	uint32_t *dInstRoot = nullptr;    // per instance: root node of its model's tree

	// frame state
	uint32_t W = 0, H = 0;
	PtTile tile = { 0, 0, 0, 0, 0, 0 };  // w == 0: the whole image
	int spp = 10, maxDepth = 20, maxFrames = 1000;
	float tmin = 0.001f, tmax = 10000.0f;
	uint32_t seed = 0;
	uint32_t firstFrame = 0, frameStride = 1;
	int accumMode = PT_ACCUM_MEAN;
	int pipeline = PT_PIPELINE_PERSISTENT;
	int frame = -1;  // m_pcRay.frame
	bool haveCamera = false;
	PtCamera cam;
	PtCamera *dCam = nullptr;  // device copy for the persistent pipeline, rewritten before every launch (stream-ordered)

	PtQueues q = {};
	size_t queueCap = 0;  // records
	float4 *image = nullptr;
	PtFrameCounters *counters = nullptr;
	unsigned long long *dStats = nullptr;  // [0] rays_total [1] rays_last_frame [2..7] traversal counters
	PtLaunchCfg cfg = {};
	uint2 *dLcgSkip = nullptr;  // persistent pipeline: LCG skip-ahead pairs for 0, 2, 4, .. steps
	int lcgSkipSpp = 0;
	float4 *pState = nullptr, *pRad = nullptr, *pRingO = nullptr, *pRingD = nullptr;
	unsigned long long *pLit = nullptr;     // one bit per record of pRad: the path has written it (PtPathsArgs::lit)
	size_t pCap = 0;                        // records in pRad
	size_t pStateCap = 0;                   // records in pState (a flat scene only needs the ring part: ensurePersistent)
	uint32_t lastLaunchFrames = 1;          // frames the most recent launch covered
	int framesPerLaunch = 8;                // PT_FRAMES_PER_LAUNCH (16 buys +0.9 % at the headline config for 8.5 GB more)
	bool framesPerLaunchSet = false;        // (set by the environment: then flat scenes do not double it)
	unsigned char *naScratch = nullptr;     // this device's copy of `scratch` (pt_render_frames_noalloc)
	size_t persistentBytes = 12ull << 30;   // budget for pState + pRad (PT_PERSISTENT_GB): 32 B per path per frame of a launch

	int profiling = 0;
	std::vector<EventPair> pending;
	std::vector<hipEvent_t> freeEvents;
	double extendMs = 0, shadeMs = 0;
	uint64_t extendLaunches = 0, shadeLaunches = 0, framesRendered = 0;
};

namespace
{
int fail(pt_ctx *c, int code, const char *fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (c) c->error = buf;
	else g_error = buf;
	return code;
}

#define PT_HIP(c, call)                                                                             \
	do                                                                                              \
	{                                                                                               \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) return fail(c, PT_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
	} while (0)

void freeQueues(pt_ctx *c)
{
	for (int i = 0; i < 2; ++i)
	{
		hipFree(c->q.rayO[i]);
		hipFree(c->q.rayD[i]);
		hipFree(c->q.thr[i]);
	}
	hipFree(c->q.hits);
	hipFree(c->q.rad);
	hipFree(c->q.chunkCount[0]);
	hipFree(c->q.chunkCount[1]);
	c->q = PtQueues();
	c->queueCap = 0;
}

void freeAccel(pt_ctx *c)
{
	hipFree(c->dInstances);
	c->dInstances = nullptr;
	hipFree(c->dInstRoot);
	c->dInstRoot = nullptr;
	c->tlasRoot = c->blasNodes = c->tlasDepth = 0;
	hipFree(c->accel.nodes);
	hipFree(c->accel.tris);
	hipFree(c->accel.codes);
	hipFree(c->accel.order);
	c->accel = PtBuildOutput();
	hipFree(c->quadAccel.nodes);
	hipFree(c->quadAccel.quads);
	c->quadAccel = PtQuadOutput();
	hipFree(c->dEntry);
	c->dEntry = nullptr;
	c->entryFlat = false;
	c->haveAccel = false;
}

void freeModelTextures(pt_ctx *c)
{
	for (pt_ctx::ModelTex &t : c->modelTex) hipFree(t.texels);
	c->modelTex.clear();
	hipFree(c->dTexSlots);
	c->dTexSlots = nullptr;
	c->nTexSlots = 0;
}

void freeScene(pt_ctx *c)
{
	freeModelTextures(c);  // (their slots follow the Models' primitive ranges; the scene-wide map stays)
	hipFree(c->dVerts);
	hipFree(c->dIndices);
	hipFree(c->dMats);
	hipFree(c->dMatIdx);
	hipFree(c->dPrimEmissive);
	hipFree(c->dShade);
	c->dShade = nullptr;
	c->dVerts = nullptr;
	c->dIndices = c->dMatIdx = nullptr;
	c->dMats = nullptr;
	c->dPrimEmissive = nullptr;
	c->haveScene = false;
	freeAccel(c);
}

int ensureQueues(pt_ctx *c, size_t records)
{
	if (records <= c->queueCap) return PT_OK;
	freeQueues(c);
	const size_t bytes = records * sizeof(float4);
	// the fused pipeline allocates output chunks densely: at most one partially filled chunk per wave
	const size_t chunks = (records + PT_CHUNK - 1) / PT_CHUNK + c->cfg.bounceWaves;
	const size_t qbytes = chunks * PT_CHUNK * sizeof(float4);
	for (int i = 0; i < 2; ++i)
	{
		PT_HIP(c, hipMalloc(&c->q.rayO[i], qbytes));
		PT_HIP(c, hipMalloc(&c->q.rayD[i], qbytes));
		PT_HIP(c, hipMalloc(&c->q.thr[i], qbytes));
		PT_HIP(c, hipMalloc(&c->q.chunkCount[i], chunks * sizeof(uint32_t)));
	}
	PT_HIP(c, hipMalloc(&c->q.hits, bytes));
	PT_HIP(c, hipMalloc(&c->q.rad, qbytes));  // item-major layout of the persistent pipeline pads the last pixel group
	c->queueCap = records;
	return PT_OK;
}

PtSceneDev sceneDev(const pt_ctx *c)
{
	PtSceneDev s;
	s.vertices = c->dVerts;
	s.indices = c->dIndices;
	s.materials = c->dMats;
	s.matIndices = c->dMatIdx;
	s.shade = c->dShade;
	s.nodes = c->accel.nodes;
	s.tris = c->accel.tris;  // plane 0; plane r at tris + r * nTris
	s.triPlanes = (uint32_t)c->accel.triPlanes;
	s.nTris = c->nIdx / 3;
	s.nNodes = c->accel.nNodes;
	s.texture = c->dTexture;
	s.texW = c->texW;
	s.texH = c->texH;
	s.texSlots = c->dTexSlots;
	s.nTexSlots = c->nTexSlots;
	if (c->dTexSlots)  // per-Model maps: any of them marks the scene as one that samples maps
		for (const pt_ctx::ModelTex &t : c->modelTex)
			if (t.texels)
			{
				s.texture = t.texels;
				s.texW = t.w;
				s.texH = t.h;
				break;
			}
	s.quads = nullptr;  // only the persistent pipeline traverses the quad version of the tree (renderPersistent)
	s.nQuads = 0;
	return s;
}

hipEvent_t getEvent(pt_ctx *c)
{
	if (!c->freeEvents.empty())
	{
		hipEvent_t e = c->freeEvents.back();
		c->freeEvents.pop_back();
		return e;
	}
	hipEvent_t e = nullptr;
	if (hipEventCreate(&e) != hipSuccess) return nullptr;  // callers treat a null event as "profiling unavailable"
	return e;
}

// Three events for one profiled launch, or none (event creation failed: that launch goes untimed).
bool getEventTriple(pt_ctx *c, hipEvent_t &e0, hipEvent_t &e1, hipEvent_t &e2)
{
	e0 = getEvent(c);
	e1 = getEvent(c);
	e2 = getEvent(c);
	if (e0 && e1 && e2) return true;
	for (hipEvent_t e : { e0, e1, e2 })
		if (e) c->freeEvents.push_back(e);
	return false;
}

// Folds finished event pairs into the per-kernel-class totals.  Caller has synchronised the stream.
void drainEvents(pt_ctx *c)
{
	for (EventPair &p : c->pending)
	{
		float ms = 0.0f;
		if (hipEventElapsedTime(&ms, p.begin, p.mid) == hipSuccess)
		{
			c->extendMs += ms;
			c->extendLaunches++;
		}
		if (hipEventElapsedTime(&ms, p.mid, p.end) == hipSuccess)
		{
			c->shadeMs += ms;
			c->shadeLaunches++;
		}
		c->freeEvents.push_back(p.begin);
		c->freeEvents.push_back(p.mid);
		c->freeEvents.push_back(p.end);
	}
	c->pending.clear();
}

int checkReady(pt_ctx *c, bool needImage)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!c->haveScene) return fail(c, PT_EINVAL, "no scene uploaded (pt_upload_scene)");
	if (!c->haveAccel) return fail(c, PT_EINVAL, "acceleration structure not built (pt_build_accel)");
	if (needImage && (!c->W || !c->H)) return fail(c, PT_EINVAL, "no image (pt_resize)");
	if (needImage && !c->haveCamera) return fail(c, PT_EINVAL, "no camera (pt_set_camera)");
	return PT_OK;
}

// Traces the rays in q.rayO[0]/q.rayD[0][0..n) into q.hits (debug entry points).
int traceQueue0(pt_ctx *c, uint32_t n, float tmin, float tmax)
{
	if (c->nInst) return fail(c, PT_EINVAL, "the ray-level debug entry points trace single-level scenes only (pt_set_instances(ctx, 0, 0))");
	ptLaunchFrameBegin(c->counters, n, 0, c->stream);
	ptLaunchExtend(sceneDev(c), c->brute, &c->counters->count[0], &c->counters->headExtend[0], c->q.rayO[0],
	               c->q.rayD[0], c->q.hits, tmin, tmax, c->cfg, nullptr, c->stream);
	PT_HIP(c, hipGetLastError());
	return PT_OK;
}
}  // namespace

extern "C" const char *pt_last_error(pt_ctx *ctx)
{
	if (ctx) return ctx->error.c_str();
	if (!g_error.empty()) return g_error.c_str();
	return pt_seam_last_error();
}

extern "C" pt_ctx *pt_create(int device)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n == 0)
	{
		fail(nullptr, PT_EHIP, "pt_create: no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
		return nullptr;
	}
	if (device < 0 || device >= n)
	{
		fail(nullptr, PT_EINVAL, "pt_create: device ordinal %d out of range [0,%d)", device, n);
		return nullptr;
	}
	if ((e = hipSetDevice(device)) != hipSuccess)
	{
		fail(nullptr, PT_EHIP, "pt_create: hipSetDevice: %s", hipGetErrorString(e));
		return nullptr;
	}
	pt_ctx *c = new pt_ctx();
	c->device = device;
	if (device < kMaxDevices)
	{
		// once per device, with that device current: the address of this device's copy of the scratch array
		std::lock_guard<std::mutex> lock(g_scratchMutex);
		if (!g_scratchPtr[device])
		{
			void *p = nullptr;
			e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
			if (e != hipSuccess || !p)
			{
				fail(nullptr, PT_EHIP, "pt_create: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
				delete c;
				return nullptr;
			}
			g_scratchPtr[device] = static_cast<unsigned char *>(p);
		}
		c->naScratch = g_scratchPtr[device];
	}
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->numCUs = prop.multiProcessorCount;
	bool ok = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) == hipSuccess;
	c->stream = c->ownStream;
	ok = ok && hipMalloc(&c->counters, sizeof(PtFrameCounters)) == hipSuccess;
	ok = ok && hipMalloc(&c->dStats, 32 * sizeof(unsigned long long)) == hipSuccess;
	ok = ok && hipMalloc(&c->dCam, sizeof(PtCamera)) == hipSuccess;
	ok = ok && hipMemset(c->dStats, 0, 32 * sizeof(unsigned long long)) == hipSuccess;
	// persistent grid: blocks of 256 threads per CU (tuning knob PT_BLOCKS_PER_CU, default 4)
	int blocksPerCU = 4;
	if (const char *e = getenv("PT_BLOCKS_PER_CU"))
	{
		int v = atoi(e);
		if (v >= 1 && v <= 8) blocksPerCU = v;
	}
	c->cfg.persistentBlocks = (uint32_t)c->numCUs * (uint32_t)blocksPerCU;
	// fused kernel: single-wave blocks, 10 KB of LDS each -> 16 per CU (tuning knob PT_BOUNCE_WAVES_PER_CU)
	int wavesPerCU = 16;
	if (const char *e = getenv("PT_BOUNCE_WAVES_PER_CU"))
	{
		int v = atoi(e);
		if (v >= 1 && v <= 32) wavesPerCU = v;
	}
	c->cfg.bounceWaves = (uint32_t)c->numCUs * (uint32_t)wavesPerCU;
	if (const char *e = getenv("PT_FRAMES_PER_LAUNCH"))
	{
		int v = atoi(e);
		if (v >= 1 && v <= 256)
		{
			c->framesPerLaunch = v;
			c->framesPerLaunchSet = true;
		}
	}
	if (const char *e = getenv("PT_PERSISTENT_GB"))
	{
		int v = atoi(e);
		if (v >= 1 && v <= 256) c->persistentBytes = (size_t)v << 30;
	}
	c->cfg.spillStride = c->cfg.persistentBlocks * 256u;
	if (c->cfg.bounceWaves * 64u > c->cfg.spillStride) c->cfg.spillStride = c->cfg.bounceWaves * 64u;
	ok = ok && hipMalloc(&c->cfg.waveRays, sizeof(uint32_t) * c->cfg.bounceWaves) == hipSuccess;
	ok = ok && hipMemset(c->cfg.waveRays, 0, sizeof(uint32_t) * c->cfg.bounceWaves) == hipSuccess;
	ok = ok && hipMalloc(&c->cfg.spill, sizeof(uint32_t) * (size_t)c->cfg.spillStride * PT_SPILL_STACK) == hipSuccess;
	if (!ok)
	{
		fail(nullptr, PT_EHIP, "pt_create: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
		pt_destroy(c);
		return nullptr;
	}
	return c;
}

extern "C" void pt_destroy(pt_ctx *c)
{
	if (!c) return;
	hipSetDevice(c->device);
	if (c->stream) hipStreamSynchronize(c->stream);
	drainEvents(c);
	for (hipEvent_t e : c->freeEvents) hipEventDestroy(e);
	freeScene(c);
	freeQueues(c);
	hipFree(c->dTexture);
	hipFree(c->image);
	hipFree(c->counters);
	hipFree(c->dStats);
	hipFree(c->dCam);
	hipFree(c->cfg.spill);
	hipFree(c->cfg.waveRays);
	hipFree(c->dLcgSkip);
	hipFree(c->pState);
	hipFree(c->pRad);
	hipFree(c->pLit);
	hipFree(c->pRingO);
	hipFree(c->pRingD);
	if (c->ownStream) hipStreamDestroy(c->ownStream);
	delete c;
}

extern "C" int pt_set_stream(pt_ctx *c, void *s)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	drainEvents(c);
	c->stream = s ? (hipStream_t)s : c->ownStream;
	return PT_OK;
}

extern "C" int pt_synchronize(pt_ctx *c)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	drainEvents(c);
	return PT_OK;
}

extern "C" int pt_upload_scene(pt_ctx *c, const PtVertex *v, uint32_t nV, const uint32_t *idx, uint32_t nIdx,
                               const PtMaterial *m, uint32_t nM, const uint32_t *matIdx) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (nIdx % 3) return fail(c, PT_EINVAL, "pt_upload_scene: n_indices (%u) is not a multiple of 3", nIdx);
	if (nIdx && (!v || !idx || !m || !matIdx || !nV || !nM))
		return fail(c, PT_EINVAL, "pt_upload_scene: null/empty buffer with %u indices", nIdx);
	if (nIdx / 3 > 0x0FFFFFFFu) return fail(c, PT_ELIMIT, "pt_upload_scene: more than 2^28-1 triangles");
	const uint32_t nTris = nIdx / 3;
	for (uint32_t i = 0; i < nIdx; ++i)
		if (idx[i] >= nV) return fail(c, PT_EINVAL, "pt_upload_scene: index %u = %u out of range (%u vertices)", i, idx[i], nV);
	std::vector<uint8_t> emissive(nTris ? nTris : 1);
	for (uint32_t t = 0; t < nTris; ++t)
	{
		if (matIdx[t] >= nM)
			return fail(c, PT_EINVAL, "pt_upload_scene: mat_indices[%u] = %u out of range (%u materials)", t, matIdx[t], nM);
		const float *e = m[matIdx[t]].emission;
		emissive[t] = ((e[0] + e[1]) + e[2] > 0.0f) ? 1 : 0;  // raytrace.rchit:56
	}
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	freeScene(c);
	c->nVerts = nV;
	c->nIdx = nIdx;
	c->nMats = nM;
	PT_HIP(c, hipMalloc(&c->dVerts, sizeof(PtVertex) * (size_t)(nV ? nV : 1)));
	PT_HIP(c, hipMalloc(&c->dIndices, 4 * (size_t)(nIdx ? nIdx : 1)));
	PT_HIP(c, hipMalloc(&c->dMats, sizeof(PtMaterial) * (size_t)(nM ? nM : 1)));
	PT_HIP(c, hipMalloc(&c->dMatIdx, 4 * (size_t)(nTris ? nTris : 1)));
	PT_HIP(c, hipMalloc(&c->dPrimEmissive, nTris ? nTris : 1));
	if (nIdx)
	{
		PT_HIP(c, hipMemcpy(c->dVerts, v, sizeof(PtVertex) * (size_t)nV, hipMemcpyHostToDevice));
		PT_HIP(c, hipMemcpy(c->dIndices, idx, 4 * (size_t)nIdx, hipMemcpyHostToDevice));
		PT_HIP(c, hipMemcpy(c->dMats, m, sizeof(PtMaterial) * (size_t)nM, hipMemcpyHostToDevice));
		PT_HIP(c, hipMemcpy(c->dMatIdx, matIdx, 4 * (size_t)nTris, hipMemcpyHostToDevice));
		PT_HIP(c, hipMemcpy(c->dPrimEmissive, emissive.data(), nTris, hipMemcpyHostToDevice));
	}
	c->haveScene = true;
	c->modelTriFirst.assign(1, 0u);
	c->modelTriCount.assign(1, nTris);
	if (c->implicitInst)
	{
		c->instances.clear();
		c->instModel.clear();
		c->nInst = 0;
		c->implicitInst = false;
	}
	if (nTris && !getenv("PT_NO_SHADE_RECORDS"))
	{
		PT_HIP(c, hipMalloc(&c->dShade, sizeof(PtShadeRec) * (size_t)nTris));
		PtSceneDev sc = sceneDev(c);
		sc.nTris = nTris;
		ptLaunchBuildShadeRecords(sc, c->dShade, c->stream);
		PT_HIP(c, hipStreamSynchronize(c->stream));
	}
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_upload_scene: out of host memory");
}

extern "C" int pt_upload_models(pt_ctx *c, const PtModelDesc *models, uint32_t nModels) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!models || !nModels) return fail(c, PT_EINVAL, "pt_upload_models: no models");
	if (nModels == 1)
		return pt_upload_scene(c, models[0].vertices, models[0].n_vertices, models[0].indices, models[0].n_indices, models[0].materials,
		                       models[0].n_materials, models[0].mat_indices);
	// One set of buffers: vertices, materials and triangles of the models one after the other, indices and material indices
	// rebased -- so a primitive id that runs through all models reaches the right vertices and material without any per-model
	// address table on the device (what raytrace.rchit:27-37 does through Model's four buffer addresses).
	uint64_t nV = 0, nIdx = 0, nM = 0;
	for (uint32_t m = 0; m < nModels; ++m)
	{
		const PtModelDesc &d = models[m];
		if (d.n_indices % 3) return fail(c, PT_EINVAL, "pt_upload_models: model %u: n_indices (%u) is not a multiple of 3", m, d.n_indices);
		if (!d.n_indices || !d.vertices || !d.indices || !d.materials || !d.mat_indices || !d.n_vertices || !d.n_materials)
			return fail(c, PT_EINVAL, "pt_upload_models: model %u: null/empty buffer", m);
		for (uint32_t i = 0; i < d.n_indices; ++i)
			if (d.indices[i] >= d.n_vertices)
				return fail(c, PT_EINVAL, "pt_upload_models: model %u: index %u = %u out of range (%u vertices)", m, i, d.indices[i], d.n_vertices);
		for (uint32_t t = 0; t < d.n_indices / 3; ++t)
			if (d.mat_indices[t] >= d.n_materials)
				return fail(c, PT_EINVAL, "pt_upload_models: model %u: mat_indices[%u] = %u out of range (%u materials)", m, t, d.mat_indices[t],
				            d.n_materials);
		nV += d.n_vertices;
		nIdx += d.n_indices;
		nM += d.n_materials;
	}
	if (nIdx / 3 > 0x0FFFFFFFull || nV > 0xFFFFFFF0ull || nM > 0xFFFFFFF0ull)
		return fail(c, PT_ELIMIT, "pt_upload_models: more than 2^28-1 triangles (or 2^32 vertices / materials) in all");
	std::vector<PtVertex> v((size_t)nV);
	std::vector<uint32_t> idx((size_t)nIdx), mid((size_t)(nIdx / 3));
	std::vector<PtMaterial> mat((size_t)nM);
	std::vector<uint32_t> first(nModels), count(nModels);
	size_t v0 = 0, i0 = 0, m0 = 0;
	for (uint32_t m = 0; m < nModels; ++m)
	{
		const PtModelDesc &d = models[m];
		memcpy(v.data() + v0, d.vertices, sizeof(PtVertex) * d.n_vertices);
		memcpy(mat.data() + m0, d.materials, sizeof(PtMaterial) * d.n_materials);
		for (uint32_t i = 0; i < d.n_indices; ++i) idx[i0 + i] = d.indices[i] + (uint32_t)v0;
		for (uint32_t t = 0; t < d.n_indices / 3; ++t) mid[i0 / 3 + t] = d.mat_indices[t] + (uint32_t)m0;
		first[m] = (uint32_t)(i0 / 3);
		count[m] = d.n_indices / 3;
		v0 += d.n_vertices;
		i0 += d.n_indices;
		m0 += d.n_materials;
	}
	int rc = pt_upload_scene(c, v.data(), (uint32_t)nV, idx.data(), (uint32_t)nIdx, mat.data(), (uint32_t)nM, mid.data());
	if (rc) return rc;
	c->modelTriFirst.swap(first);
	c->modelTriCount.swap(count);
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_upload_models: out of host memory");
}

namespace
{
// sRGB8 image -> linear float4 texels on the device (alpha is not a colour; the shading does not use it)
int uploadTexels(pt_ctx *c, const char *who, const uint8_t *rgba8, uint32_t w, uint32_t h, float4 **out)
{
	const size_t n = (size_t)w * h;
	std::vector<float> lin(4 * n);
	pt_srgb8_to_linear(rgba8, (uint32_t)(4 * n), lin.data());
	for (size_t i = 0; i < n; ++i) lin[4 * i + 3] = (float)rgba8[4 * i + 3] / 255.0f;
	float4 *d = nullptr;
	PT_HIP(c, hipMalloc(&d, sizeof(float4) * n));
	hipError_t e = hipMemcpy(d, lin.data(), sizeof(float4) * n, hipMemcpyHostToDevice);
	if (e != hipSuccess)
	{
		hipFree(d);  // never leave a map without an extent behind: the sampler divides by it
		return fail(c, PT_EHIP, "%s: upload failed: %s", who, hipGetErrorString(e));
	}
	*out = d;
	return PT_OK;
}

}  // namespace

extern "C" int pt_set_texture(pt_ctx *c, const uint8_t *rgba8, uint32_t w, uint32_t h) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if ((rgba8 == nullptr) != (w == 0 || h == 0)) return fail(c, PT_EINVAL, "pt_set_texture: pixels and extent must both be given or both be empty");
	if (w > 16384 || h > 16384) return fail(c, PT_ELIMIT, "pt_set_texture: %u x %u exceeds 16384 x 16384", w, h);
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	hipFree(c->dTexture);
	c->dTexture = nullptr;
	c->texW = c->texH = 0;
	freeModelTextures(c);  // the scene-wide map replaces per-Model ones
	if (!rgba8) return PT_OK;
	int rc = uploadTexels(c, "pt_set_texture", rgba8, w, h, &c->dTexture);
	if (rc) return rc;
	c->texW = w;
	c->texH = h;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_set_texture: out of host memory");
}

extern "C" int pt_set_model_texture(pt_ctx *c, uint32_t model, const uint8_t *rgba8, uint32_t w, uint32_t h) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!c->haveScene) return fail(c, PT_EINVAL, "pt_set_model_texture: no scene uploaded");
	const uint32_t nModels = (uint32_t)c->modelTriCount.size();
	if (model >= nModels) return fail(c, PT_EINVAL, "pt_set_model_texture: model %u of %u", model, nModels);
	if ((rgba8 == nullptr) != (w == 0 || h == 0))
		return fail(c, PT_EINVAL, "pt_set_model_texture: pixels and extent must both be given or both be empty");
	if (w > 16384 || h > 16384) return fail(c, PT_ELIMIT, "pt_set_model_texture: %u x %u exceeds 16384 x 16384", w, h);
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	// the new map first: a failed upload leaves every map, and the table that points to them, as it was
	pt_ctx::ModelTex fresh;
	if (rgba8)
	{
		int rc = uploadTexels(c, "pt_set_model_texture", rgba8, w, h, &fresh.texels);
		if (rc) return rc;
		fresh.w = w;
		fresh.h = h;
	}
	// per-Model maps replace the scene-wide one
	hipFree(c->dTexture);
	c->dTexture = nullptr;
	c->texW = c->texH = 0;
	c->modelTex.resize(nModels);
	// (the table goes before the texels it points to: nothing may find a freed map through it if a step below fails)
	hipFree(c->dTexSlots);
	c->dTexSlots = nullptr;
	c->nTexSlots = 0;
	hipFree(c->modelTex[model].texels);
	c->modelTex[model] = fresh;
	// the table the shading searches: one slot per Model, in primitive order
	bool any = false;
	std::vector<PtTexSlot> slots(nModels);
	for (uint32_t m = 0; m < nModels; ++m)
	{
		slots[m].texels = c->modelTex[m].texels;
		slots[m].w = c->modelTex[m].w;
		slots[m].h = c->modelTex[m].h;
		slots[m].primEnd = c->modelTriFirst[m] + c->modelTriCount[m];
		slots[m].pad = 0;
		any = any || c->modelTex[m].texels;
	}
	if (!any)
	{
		freeModelTextures(c);
		return PT_OK;
	}
	PT_HIP(c, hipMalloc(&c->dTexSlots, sizeof(PtTexSlot) * nModels));
	PT_HIP(c, hipMemcpy(c->dTexSlots, slots.data(), sizeof(PtTexSlot) * nModels, hipMemcpyHostToDevice));
	c->nTexSlots = nModels;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_set_model_texture: out of host memory");
}

namespace
{
// Top-level structure over the instances (buildTlas, VulkanRaytracing.cpp:382-527): world-space box of every
// instance = transformed corners of the bottom-level root box, a BVH4 over those boxes (same builder, box mode),
// placed behind the bottom-level nodes in one array so that node references are global.
int buildTopLevel(pt_ctx *c, int builder)
{
	// object-space box of every Model = the box of its tree's root node
	const size_t nModels = c->modelRoot.size();
	std::vector<float> mlo(3 * nModels, INFINITY), mhi(3 * nModels, -INFINITY);
	for (size_t m = 0; m < nModels; ++m)
	{
		PtNode4 root;
		PT_HIP(c, hipMemcpy(&root, c->accel.nodes + c->modelRoot[m], sizeof(root), hipMemcpyDeviceToHost));
		const float scale[3] = { root.scaleX, root.scaleY, root.scaleZ };
		for (int ch = 0; ch < 4; ++ch)
		{
			if (root.child[ch] == PT_EMPTY_CHILD) continue;
			for (int a = 0; a < 3; ++a)
			{
				float l = fmaf((float)((root.qlo[a] >> (8 * ch)) & 0xFFu), scale[a], root.origin[a]);
				float h = fmaf((float)((root.qhi[a] >> (8 * ch)) & 0xFFu), scale[a], root.origin[a]);
				mlo[3 * m + a] = fminf(mlo[3 * m + a], l);
				mhi[3 * m + a] = fmaxf(mhi[3 * m + a], h);
			}
		}
	}
	const uint32_t n = c->nInst;
	std::vector<float4> bl(n), bh(n);
	std::vector<uint32_t> roots(n);
	for (uint32_t i = 0; i < n; ++i)
	{
		const float *t = &c->instances[(size_t)i * 24];
		const uint32_t model = c->instModel[i];
		const float *lo = &mlo[3 * (size_t)model], *hi = &mhi[3 * (size_t)model];
		roots[i] = c->modelRoot[model];
		float wl[3] = { INFINITY, INFINITY, INFINITY }, wh[3] = { -INFINITY, -INFINITY, -INFINITY };
		for (int corner = 0; corner < 8; ++corner)
		{
			const float p[3] = { (corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2] };
			for (int r = 0; r < 3; ++r)
			{
				float w = ((t[4 * r] * p[0] + t[4 * r + 1] * p[1]) + t[4 * r + 2] * p[2]) + t[4 * r + 3];
				wl[r] = fminf(wl[r], w);
				wh[r] = fmaxf(wh[r], w);
			}
		}
		bl[i] = make_float4(wl[0], wl[1], wl[2], 0.0f);
		bh[i] = make_float4(wh[0], wh[1], wh[2], 0.0f);
	}
	float4 *dLo = nullptr, *dHi = nullptr;
	hipError_t e = hipMalloc(&dLo, sizeof(float4) * n);
	if (e == hipSuccess) e = hipMalloc(&dHi, sizeof(float4) * n);
	if (e == hipSuccess) e = hipMemcpy(dLo, bl.data(), sizeof(float4) * n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(dHi, bh.data(), sizeof(float4) * n, hipMemcpyHostToDevice);
	PtBuildInput in = { nullptr, nullptr, n, 1, builder };
	in.boxLo = dLo;
	in.boxHi = dHi;
	in.nodeBase = c->blasNodes;
	in.padExp = 14;  // the object-space hit test and the world-space box disagree by the rounding of two transforms
	PtBuildOutput top;
	if (e == hipSuccess) e = ptBuildBvh4(in, top, c->stream);
	// Which builder makes the better tree over a set of instance boxes depends on the layout: on a regular grid the Morton
	// order IS the right hierarchy and LBVH wins (teapot x 64 x 64: 10.1 against 9.1 Grays/s), on scattered or clustered
	// instances PLOC does (+3...6 %, scripts/gpu_tlas_builder.py).  The top level is a few thousand boxes and a few
	// milliseconds, so BOTH are built and the one with the smaller surface-area cost -- the sum of its wide nodes' box areas,
	// i.e. the expected number of node visits of a random ray, up to a constant -- is kept.  PT_TLAS_BUILDER=lbvh|ploc forces one.
	const char *forced = getenv("PT_TLAS_BUILDER");
	if (e == hipSuccess && n > 4 && !forced)
	{
		auto cost = [&](const PtBuildOutput &t, double &out) -> hipError_t {
			std::vector<PtNode4> h(t.nNodes);
			hipError_t ce = hipMemcpy(h.data(), t.nodes, sizeof(PtNode4) * t.nNodes, hipMemcpyDeviceToHost);
			out = 0.0;
			for (const PtNode4 &nd : h)
			{
				const float scale[3] = { nd.scaleX, nd.scaleY, nd.scaleZ };
				double lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
				for (int ch = 0; ch < 4; ++ch)
				{
					if (nd.child[ch] == PT_EMPTY_CHILD) continue;
					for (int a = 0; a < 3; ++a)
					{
						const double l = (double)nd.origin[a] + (double)((nd.qlo[a] >> (8 * ch)) & 0xFFu) * scale[a];
						const double u = (double)nd.origin[a] + (double)((nd.qhi[a] >> (8 * ch)) & 0xFFu) * scale[a];
						if (l < lo[a]) lo[a] = l;
						if (u > hi[a]) hi[a] = u;
					}
				}
				const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
				if (dx >= 0.0 && dy >= 0.0 && dz >= 0.0) out += dx * dy + dy * dz + dz * dx;
			}
			return ce;
		};
		PtBuildInput in2 = in;
		in2.builder = builder == PT_BUILDER_PLOC ? PT_BUILDER_LBVH : PT_BUILDER_PLOC;
		PtBuildOutput other;
		e = ptBuildBvh4(in2, other, c->stream);
		double costTop = 0.0, costOther = 0.0;
		if (e == hipSuccess) e = cost(top, costTop);
		if (e == hipSuccess) e = cost(other, costOther);
		PtBuildOutput &drop = (e == hipSuccess && costOther < costTop) ? top : other;
		hipFree(drop.nodes);
		hipFree(drop.tris);
		hipFree(drop.codes);
		hipFree(drop.order);
		if (&drop == &top) top = other;
	}
	else if (e == hipSuccess && forced)
	{
		const int want = (forced[0] == 'l' || forced[0] == 'L') ? PT_BUILDER_LBVH : PT_BUILDER_PLOC;
		if (want != builder)
		{
			hipFree(top.nodes);
			hipFree(top.tris);
			hipFree(top.codes);
			hipFree(top.order);
			in.builder = want;
			e = ptBuildBvh4(in, top, c->stream);
		}
	}
	hipFree(dLo);
	hipFree(dHi);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_build_accel: top-level build failed: %s", hipGetErrorString(e));
	PtNode4 *all = nullptr;
	e = hipMalloc(&all, sizeof(PtNode4) * ((size_t)c->blasNodes + top.nNodes));
	if (e == hipSuccess) e = hipMemcpy(all, c->accel.nodes, sizeof(PtNode4) * c->blasNodes, hipMemcpyDeviceToDevice);
	if (e == hipSuccess) e = hipMemcpy(all + c->blasNodes, top.nodes, sizeof(PtNode4) * top.nNodes, hipMemcpyDeviceToDevice);
	if (e == hipSuccess) e = hipMalloc(&c->dInstances, sizeof(float) * c->instances.size());
	if (e == hipSuccess)
		e = hipMemcpy(c->dInstances, c->instances.data(), sizeof(float) * c->instances.size(), hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMalloc(&c->dInstRoot, sizeof(uint32_t) * roots.size());
	if (e == hipSuccess) e = hipMemcpy(c->dInstRoot, roots.data(), sizeof(uint32_t) * roots.size(), hipMemcpyHostToDevice);
	hipFree(top.nodes);
	hipFree(top.tris);
	hipFree(top.codes);
	hipFree(top.order);
	if (e != hipSuccess)
	{
		hipFree(all);
		return fail(c, PT_EHIP, "pt_build_accel: top-level assembly failed: %s", hipGetErrorString(e));
	}
	if (c->quadAccel.nQuads)
	{
		// the quad version of the bottom-level nodes gets the same top level behind it
		PtNode4 *allQ = nullptr;
		hipError_t qe = hipMalloc(&allQ, sizeof(PtNode4) * ((size_t)c->blasNodes + top.nNodes));
		if (qe == hipSuccess) qe = hipMemcpy(allQ, c->quadAccel.nodes, sizeof(PtNode4) * c->blasNodes, hipMemcpyDeviceToDevice);
		if (qe == hipSuccess) qe = hipMemcpy(allQ + c->blasNodes, all + c->blasNodes, sizeof(PtNode4) * top.nNodes, hipMemcpyDeviceToDevice);
		if (qe != hipSuccess)
		{
			hipFree(allQ);
			hipFree(all);
			return fail(c, PT_EHIP, "pt_build_accel: top-level assembly failed: %s", hipGetErrorString(qe));
		}
		hipFree(c->quadAccel.nodes);
		c->quadAccel.nodes = allQ;
	}
	hipFree(c->accel.nodes);
	c->accel.nodes = all;
	c->accel.nNodes = c->blasNodes + top.nNodes;
	c->tlasRoot = c->blasNodes;
	c->tlasDepth = top.depth;
	return PT_OK;
}
}  // namespace

extern "C" int pt_build_accel(pt_ctx *c, int flags) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!c->haveScene) return fail(c, PT_EINVAL, "pt_build_accel: no scene uploaded");
	const int mode = flags & 0xFF, leafMax = (flags >> 8) & 0xF;
	const bool compact = !(flags & PT_ACCEL_NO_COMPACTION);
	if ((mode != PT_ACCEL_BVH4 && mode != PT_ACCEL_BRUTE_FORCE && mode != PT_ACCEL_BVH4_PLOC) || (flags & ~0xFFFF) || leafMax > 8 ||
	    ((flags & PT_ACCEL_TRI_PLANES_1) && (flags & PT_ACCEL_TRI_PLANES_3)))
		return fail(c, PT_EINVAL, "pt_build_accel: bad flags 0x%x", flags);
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	freeAccel(c);
	c->brute = (mode == PT_ACCEL_BRUTE_FORCE);
	const uint32_t nModels = (uint32_t)c->modelTriCount.size();
	if (nModels > 1)
	{
		if (c->brute) return fail(c, PT_EINVAL, "pt_build_accel: PT_ACCEL_BRUTE_FORCE traces one Model");
		if (c->nInst == 0 || c->implicitInst)
		{
			// the reference's own instance list (createInstancesBuffer, VulkanRaytracing.cpp:382-411): instance i = Model i,
			// identity transform
			c->instances.assign((size_t)nModels * 24, 0.0f);
			c->instModel.resize(nModels);
			for (uint32_t m = 0; m < nModels; ++m)
			{
				float *t = &c->instances[(size_t)m * 24];
				t[0] = t[5] = t[10] = t[12] = t[17] = t[22] = 1.0f;
				c->instModel[m] = m;
			}
			c->nInst = nModels;
			c->implicitInst = true;
		}
	}
	for (uint32_t i = 0; i < c->nInst; ++i)
		if (c->instModel[i] >= nModels)
			return fail(c, PT_EINVAL, "pt_build_accel: instance %u: custom_index %u, but the scene has %u Model%s", i, c->instModel[i], nModels,
			            nModels == 1 ? "" : "s");
	// The Morton codes / sorted order / BVH are built in both modes (the debug entry points read
	// them); brute-force mode just does not traverse the hierarchy.
	PtBuildInput in = { c->dVerts, c->dIndices, c->nIdx / 3, leafMax,
	                    mode == PT_ACCEL_BVH4_PLOC ? PT_BUILDER_PLOC : PT_BUILDER_LBVH };
	// Leaf triangle records once per dominant ray axis (the traversal then spends nothing per triangle on the permutation) --
	// or, for one big Model, the identity plane only: at PT_ONE_PLANE_TRIS triangles and above the traversal waits for memory
	// (DESIGN.md 5.3c), the selects are free and a third of the footprint caches better (measured, profiles/r03p/q_ab_planes*:
	// soup of 1 M / 4 M / 20 M triangles +1.0 / +4.1 / +1.3 % with one plane, viking_room and the teapot grid -1 %), while the
	// two extra planes are 96 B per triangle to write and to keep (100 M triangles: 9.6 GB, 2.3 ms of the build).  Scenes of
	// several Models keep three planes.
	in.triPlanes = 3;
	if (nModels == 1 && !(flags & PT_ACCEL_TRI_PLANES_3) && ((flags & PT_ACCEL_TRI_PLANES_1) || c->nIdx / 3 >= PT_ONE_PLANE_TRIS)) in.triPlanes = 1;
	hipEvent_t a = getEvent(c), b = getEvent(c);
	const bool timed = a && b;  // without events the build is simply not timed
	if (timed) hipEventRecord(a, c->stream);
	hipError_t e = hipSuccess;
	c->modelRoot.assign(nModels, 0u);
	if (nModels == 1)
	{
		in.doneEvent = timed ? b : nullptr;  // build_ms ends behind the build's last kernel, not behind the release of its scratch memory
		e = ptBuildBvh4(in, c->accel, c->stream);
		in.doneEvent = nullptr;
	}
	else
	{
		// one bottom-level tree per Model (buildBlas loops over blasInputs, VulkanRaytracing.cpp:158-262), each built over its
		// own slice of the index buffer, then moved into ONE node array and ONE triangle array so that node and leaf
		// references are global and the traversal needs nothing per model but its root
		std::vector<PtBuildOutput> parts(nModels);
		uint64_t nodes = 0;
		for (uint32_t m = 0; m < nModels && e == hipSuccess; ++m)
		{
			PtBuildInput im = in;
			im.indices = c->dIndices + 3 * (size_t)c->modelTriFirst[m];
			im.nTris = c->modelTriCount[m];
			e = ptBuildBvh4(im, parts[m], c->stream);
			if (e == hipSuccess && parts[m].nNodes == 0) e = hipErrorInvalidValue;
			nodes += parts[m].nNodes;
		}
		PtBuildOutput all;
		const uint32_t total = c->nIdx / 3;
		if (e == hipSuccess && nodes > 0x7FFFFFF0ull) e = hipErrorInvalidValue;
		if (e == hipSuccess) e = hipMalloc(&all.nodes, sizeof(PtNode4) * (size_t)nodes);
		if (e == hipSuccess) e = hipMalloc(&all.tris, sizeof(PtTri48) * 3 * (size_t)total);
		if (e == hipSuccess) e = hipMalloc(&all.codes, 4 * (size_t)total);
		if (e == hipSuccess) e = hipMalloc(&all.order, 4 * (size_t)total);
		uint32_t nodeBase = 0;
		for (uint32_t m = 0; m < nModels && e == hipSuccess; ++m)
		{
			c->modelRoot[m] = nodeBase;
			e = ptRelocateBlas(parts[m], nodeBase, c->modelTriFirst[m], total, all.nodes, all.tris, all.codes, all.order, c->stream);
			nodeBase += parts[m].nNodes;
			if (parts[m].depth > all.depth) all.depth = parts[m].depth;
			for (int k = 0; k < 3; ++k)
			{
				all.cmin[k] = m ? fminf(all.cmin[k], parts[m].cmin[k]) : parts[m].cmin[k];
				all.cmax[k] = m ? fmaxf(all.cmax[k], parts[m].cmax[k]) : parts[m].cmax[k];
			}
		}
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		for (PtBuildOutput &p : parts)
		{
			hipFree(p.nodes);
			hipFree(p.tris);
			hipFree(p.codes);
			hipFree(p.order);
		}
		if (e == hipSuccess)
		{
			all.nNodes = nodeBase;
			all.nTris = total;
			c->accel = all;
		}
		else
		{
			hipFree(all.nodes);
			hipFree(all.tris);
			hipFree(all.codes);
			hipFree(all.order);
		}
	}
	float ms = 0.0f;
	if (timed)
	{
		if (nModels != 1 || e != hipSuccess || c->accel.nTris == 0) hipEventRecord(b, c->stream);  // (else: recorded by the builder)
		hipEventSynchronize(b);
		hipEventElapsedTime(&ms, a, b);
	}
	else hipStreamSynchronize(c->stream);
	c->buildMs = ms;
	if (a) c->freeEvents.push_back(a);
	if (b) c->freeEvents.push_back(b);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_build_accel: BVH build failed: %s", hipGetErrorString(e));
	c->blasNodes = c->accel.nNodes;
	{
		// cmdCompactBlas (VulkanRaytracing.cpp:300-380): the builder sizes the node array for the worst case (one wide
		// node per primitive); copy the nodes actually produced into an exact-size allocation and release the rest
		const uint64_t nPrims = c->nIdx / 3;
		const uint64_t fixed = nPrims * ((uint64_t)c->accel.triPlanes * sizeof(PtTri48) + 8);  // leaf triangles (three planes, or one) + codes + order
		c->accelBytesUncompacted = fixed + nPrims * sizeof(PtNode4);
		c->accelBytes = c->accelBytesUncompacted;
		if (nModels > 1) c->accelBytes = fixed + (uint64_t)c->accel.nNodes * sizeof(PtNode4);  // assembled at its exact size
		else if (compact && c->accel.nNodes && c->accel.nNodes < nPrims)
		{
			PtNode4 *exact = nullptr;
			hipError_t ce = hipMalloc(&exact, sizeof(PtNode4) * (size_t)c->accel.nNodes);
			if (ce == hipSuccess) ce = hipMemcpy(exact, c->accel.nodes, sizeof(PtNode4) * (size_t)c->accel.nNodes, hipMemcpyDeviceToDevice);
			if (ce != hipSuccess)
			{
				hipFree(exact);
				freeAccel(c);
				return fail(c, PT_EHIP, "pt_build_accel: compaction failed: %s", hipGetErrorString(ce));
			}
			hipFree(c->accel.nodes);
			c->accel.nodes = exact;
			c->accelBytes = fixed + (uint64_t)c->accel.nNodes * sizeof(PtNode4);
		}
	}
	if (!c->brute && c->accel.nNodes && (leafMax == 0 ? PT_LEAF_DEFAULT : leafMax) <= 7 && !getenv("PT_NO_QUADS"))
	{
		// scenes made of quad faces: two-triangle fan leaves become quad leaves for k_paths (pt_build.h)
		hipError_t qe = ptBuildQuadLeaves(c->accel.nodes, c->accel.nNodes, c->accel.tris + (size_t)(c->accel.triPlanes - 1) * c->accel.nTris,
		                                  c->accel.nTris, 0.9f, c->quadAccel, c->stream);
		if (qe != hipSuccess)
		{
			freeAccel(c);
			return fail(c, PT_EHIP, "pt_build_accel: quad leaves failed: %s", hipGetErrorString(qe));
		}
	}
	// (not for trees of millions of nodes: the children of the entry node are visited in a fixed order, and where a ray
	// crosses many subtrees the nearest-first descent of the ordinary root is worth more: 20 M-triangle soup 733 vs 672 Mrays/s)
	if (c->nInst && !c->brute && c->accel.nNodes)
	{
		int rc = buildTopLevel(c, in.builder);
		if (rc)
		{
			freeAccel(c);
			return rc;
		}
	}
	// (two-level scenes: only on request -- a ray crosses many instance boxes there, and visiting the entry node's children
	// in a fixed order instead of nearest first costs more than the wide first step saves: 64 x 64 grid 5.41 vs 5.78 Grays/s)
	if (!c->brute && c->accel.nNodes && (c->accel.nNodes <= (1u << 20) || getenv("PT_ENTRY_ANY_SIZE")) && !(flags & PT_ACCEL_NO_ENTRY_NODE) &&
	    !getenv("PT_NO_ENTRY") &&
	    (!c->tlasDepth || getenv("PT_ENTRY_TWO_LEVEL")))
	{
		// the tree top (two-level scenes: the top of the top-level tree) as one wide node that k_paths tests while it stages
		// a ray (of the node array k_paths traverses)
		const bool two = c->tlasDepth != 0;
		hipError_t ee = ptBuildEntryNode(c->quadAccel.nQuads ? c->quadAccel.nodes : c->accel.nodes, c->accel.nNodes, two ? c->tlasRoot : 0u,
		                                 c->accel.tris + (size_t)(c->accel.triPlanes - 1) * c->accel.nTris, c->accel.nTris, c->quadAccel.quads,
		                                 c->quadAccel.nQuads, two, &c->dEntry,
		                                 &c->entryFlat, c->stream);
		if (getenv("PT_NO_FLAT")) c->entryFlat = false;
		if (ee != hipSuccess)
		{
			freeAccel(c);
			return fail(c, PT_EHIP, "pt_build_accel: entry node failed: %s", hipGetErrorString(ee));
		}
	}
	// stack entries a traversal can need: 3 per level (+ the sentinel), per tree (+ the instance marker)
	const uint32_t need = 3 * c->accel.depth + 1 + (c->tlasDepth ? 3 * c->tlasDepth + 1 : 0);
	if (need > PT_STACK_CAPACITY)
	{
		uint32_t d = c->accel.depth, dt = c->tlasDepth;
		freeAccel(c);
		return fail(c, PT_ELIMIT, "pt_build_accel: BVH4 depth %u (+ top level %u) exceeds the traversal stack (%d entries)", d,
		            dt, (int)PT_STACK_CAPACITY);
	}
	c->haveAccel = true;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_build_accel: out of host memory");
}

extern "C" int pt_set_instances(pt_ctx *c, const PtInstance *inst, uint32_t n) try
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (n && !inst) return fail(c, PT_EINVAL, "pt_set_instances: null array");
	if (n > 0x0FFFFFF0u) return fail(c, PT_ELIMIT, "pt_set_instances: too many instances");
	std::vector<float> v((size_t)n * 24);
	std::vector<uint32_t> model(n);
	for (uint32_t i = 0; i < n; ++i)
	{
		model[i] = inst[i].custom_index;  // checked against the number of Models by pt_build_accel
		const float *t = inst[i].transform;
		for (int k = 0; k < 12; ++k)
			if (!(fabsf(t[k]) <= 3.0e38f)) return fail(c, PT_EINVAL, "pt_set_instances: instance %u: non-finite transform", i);
		// glm::inverse of the 4x4 (column-major m[c][r]) whose last row is (0,0,0,1): what gl_WorldToObjectEXT holds
		float m[16], inv[16];
		for (int col = 0; col < 4; ++col)
		{
			for (int r = 0; r < 3; ++r) m[4 * col + r] = t[4 * r + col];
			m[4 * col + 3] = col == 3 ? 1.0f : 0.0f;
		}
		pt_inverse4(m, inv);
		float *o = &v[(size_t)i * 24];
		for (int k = 0; k < 12; ++k) o[k] = t[k];
		for (int r = 0; r < 3; ++r)
			for (int col = 0; col < 4; ++col)
			{
				float x = inv[4 * col + r];
				if (!(fabsf(x) <= 3.0e38f)) return fail(c, PT_EINVAL, "pt_set_instances: instance %u: singular transform", i);
				o[12 + 4 * r + col] = x;
			}
	}
	c->instances.swap(v);
	c->instModel.swap(model);
	c->implicitInst = false;
	c->nInst = n;
	c->haveAccel = false;  // the top-level structure must be rebuilt: pt_build_accel
	c->frame = -1;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_set_instances: out of host memory");
}

extern "C" int pt_resize(pt_ctx *c, uint32_t W, uint32_t H)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!W || !H) return fail(c, PT_EINVAL, "pt_resize: zero extent %ux%u", W, H);
	if ((uint64_t)W * H * (uint64_t)c->spp > 0xFFFFFFF0ull)
		return fail(c, PT_ELIMIT, "pt_resize: W*H*spp_per_frame exceeds 2^32 path ids");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	hipFree(c->image);
	c->image = nullptr;
	PT_HIP(c, hipMalloc(&c->image, sizeof(float4) * (size_t)W * H));
	PT_HIP(c, hipMemset(c->image, 0, sizeof(float4) * (size_t)W * H));
	c->W = W;
	c->H = H;
	c->tile = PtTile{ 0, 0, 0, 0 };
	c->frame = -1;
	return PT_OK;
}

extern "C" int pt_image_extent(pt_ctx *c, uint32_t *W, uint32_t *H)
{
	if (!c || !W || !H) return fail(c, PT_EINVAL, "pt_image_extent: null argument");
	*W = c->W;
	*H = c->H;
	return PT_OK;
}

extern "C" int pt_set_tile(pt_ctx *c, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!c->W) return fail(c, PT_EINVAL, "pt_set_tile: no image (pt_resize)");
	if (w == 0 && h == 0)
	{
		c->tile = PtTile{ 0, 0, 0, 0 };
		return PT_OK;
	}
	if (!w || !h || (uint64_t)x0 + w > c->W || (uint64_t)y0 + h > c->H)
		return fail(c, PT_EINVAL, "pt_set_tile: %ux%u at (%u,%u) does not fit %ux%u", w, h, x0, y0, c->W, c->H);
	c->tile = PtTile{ x0, y0, w, h };
	return PT_OK;
}

// Blocks of block_w columns dealt round-robin: rank r of `world` gets the blocks r, r + world, ... (all rows).
extern "C" int pt_set_tile_interleaved(pt_ctx *c, uint32_t rank, uint32_t world, uint32_t block_w)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!c->W) return fail(c, PT_EINVAL, "pt_set_tile_interleaved: no image (pt_resize)");
	if (!world || rank >= world || !block_w) return fail(c, PT_EINVAL, "pt_set_tile_interleaved: rank %u of %u, blocks of %u columns", rank, world, block_w);
	uint64_t columns = 0;
	for (uint64_t x = (uint64_t)rank * block_w; x < c->W; x += (uint64_t)world * block_w)
		columns += (c->W - x < block_w) ? c->W - x : block_w;
	if (!columns)
		return fail(c, PT_EINVAL, "pt_set_tile_interleaved: rank %u of %u gets no column of %u in blocks of %u", rank, world, c->W, block_w);
	c->tile = PtTile{ rank * block_w, 0, (uint32_t)columns, c->H, block_w, world };
	return PT_OK;
}

extern "C" int pt_set_camera(pt_ctx *c, const float view[16], const float proj[16])
{
	if (!c || !view || !proj) return fail(c, PT_EINVAL, "pt_set_camera: null argument");
	PtCamera cam;
	pt_inverse4(view, cam.invView);
	pt_inverse4(proj, cam.invProj);
	pt_camera_fill_constants(cam);
	// VulkanRaytracing::updateFrame's stated intent ("Resets the frame counter if the camera has changed",
	// VulkanRaytracing.cpp:903-908, left as a TODO there): a different camera restarts the accumulation; the same
	// matrices again (main.cpp calls updateUniformBuffer every frame) change nothing.
	if (c->haveCamera && memcmp(&cam, &c->cam, sizeof(cam)) != 0) c->frame = -1;
	c->cam = cam;
	c->haveCamera = true;
	return PT_OK;
}

extern "C" int pt_set_params(pt_ctx *c, int spp, int depth, float tmin, float tmax, uint32_t seed, int maxFrames)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (spp < 1) return fail(c, PT_EINVAL, "pt_set_params: spp_per_frame %d < 1", spp);
	if (depth < 1 || depth > PT_MAX_DEPTH) return fail(c, PT_ELIMIT, "pt_set_params: max_depth %d outside [1,%d]", depth, PT_MAX_DEPTH);
	if (!(tmin >= 0.0f) || !(tmax > tmin)) return fail(c, PT_EINVAL, "pt_set_params: need 0 <= tmin < tmax");
	if (c->W && (uint64_t)c->W * c->H * (uint64_t)spp > 0xFFFFFFF0ull)
		return fail(c, PT_ELIMIT, "pt_set_params: W*H*spp_per_frame exceeds 2^32 path ids");
	c->spp = spp;
	c->maxDepth = depth;
	c->tmin = tmin;
	c->tmax = tmax;
	c->seed = seed;
	c->maxFrames = maxFrames;
	return PT_OK;
}

extern "C" int pt_set_frame_schedule(pt_ctx *c, uint32_t first, uint32_t stride)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (!stride) return fail(c, PT_EINVAL, "pt_set_frame_schedule: stride 0");
	c->firstFrame = first;
	c->frameStride = stride;
	return PT_OK;
}

extern "C" int pt_set_accum_mode(pt_ctx *c, int mode)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (mode != PT_ACCUM_MEAN && mode != PT_ACCUM_SUM) return fail(c, PT_EINVAL, "pt_set_accum_mode: bad mode %d", mode);
	c->accumMode = mode;
	return PT_OK;
}

namespace
{
int ensureLcgSkip(pt_ctx *c)
{
	if (c->lcgSkipSpp == c->spp) return PT_OK;
	// lcg^(2i)(s) = A_i * s + C_i (mod 2^32): lets a wave enter a pixel's seed chain at sample i (rgen:28-37)
	std::vector<uint2> tab((size_t)c->spp);
	uint32_t A = 1u, C = 0u;
	for (int i = 0; i < c->spp; ++i)
	{
		tab[(size_t)i] = make_uint2(A, C);
		for (int k = 0; k < 2; ++k)
		{
			A = 1664525u * A;
			C = 1664525u * C + 1013904223u;
		}
	}
	PT_HIP(c, hipStreamSynchronize(c->stream));  // frames in flight still read the old table
	hipFree(c->dLcgSkip);
	c->dLcgSkip = nullptr;
	PT_HIP(c, hipMalloc(&c->dLcgSkip, sizeof(uint2) * tab.size()));
	PT_HIP(c, hipMemcpy(c->dLcgSkip, tab.data(), sizeof(uint2) * tab.size(), hipMemcpyHostToDevice));
	c->lcgSkipSpp = c->spp;
	return PT_OK;
}

// Buffers of the persistent pipeline: per-path state and radiance for every frame of a launch, wave-private rings.
// outOfMemory: set when the state/radiance pair does not fit -- the caller retries with fewer frames per launch.
// flatScene: the flat-scene kernel (pt_paths.hip) carries throughput and RNG state with the ray; of the state buffer it uses the
// first waves * PT_PATHS_RING records as a third ring array and nothing else
int ensurePersistent(pt_ctx *c, size_t records, bool flatScene, bool *outOfMemory)
{
	*outOfMemory = false;
	const size_t ringRecords = (size_t)c->cfg.bounceWaves * PT_PATHS_RING;
	// (the flat kernel keeps no state record per path only when it is BUILT to carry the state with the ray)
	const bool ringOnly = flatScene && PT_FLAT_CARRY_STATE != 0;
	size_t stateRecords = ringOnly ? ringRecords : (records < ringRecords ? ringRecords : records);
	if (records > c->pCap || stateRecords > c->pStateCap)
	{
		PT_HIP(c, hipStreamSynchronize(c->stream));
		hipFree(c->pState);
		hipFree(c->pRad);
		hipFree(c->pLit);
		c->pState = c->pRad = nullptr;
		c->pLit = nullptr;
		// Both buffers are sized for THIS request (the caller planned it against the memory budget): a request that fits what
		// is there never gets here, so smaller calls between larger ones do not reallocate -- and a context that goes from a
		// flat scene (radiance for twice the frames, no state table) to one with a tree does not keep the larger radiance
		// buffer beside a full state table.
		c->pCap = c->pStateCap = 0;
		hipError_t e = hipMalloc(&c->pState, stateRecords * sizeof(float4));
		if (e == hipSuccess) e = hipMalloc(&c->pRad, records * sizeof(float4));
		if (e == hipSuccess) e = hipMalloc(&c->pLit, (records + 63) / 64 * sizeof(unsigned long long));
		if (e != hipSuccess)
		{
			hipFree(c->pState);
			hipFree(c->pRad);
			c->pState = c->pRad = nullptr;
			(void)hipGetLastError();
			*outOfMemory = (e == hipErrorOutOfMemory);
			return fail(c, e == hipErrorOutOfMemory ? PT_ENOMEM : PT_EHIP, "pt_render_frame: %zu bytes of path state: %s",
			            records * 2 * sizeof(float4), hipGetErrorString(e));
		}
		c->pCap = records;
		c->pStateCap = stateRecords;
	}
	if (!c->pRingO)
	{
		const size_t ring = (size_t)c->cfg.bounceWaves * PT_PATHS_RING * sizeof(float4);
		PT_HIP(c, hipMalloc(&c->pRingO, ring));
		PT_HIP(c, hipMalloc(&c->pRingD, ring));
	}
	return PT_OK;
}

// `n` consecutive active frames (c->frame + 1 ...) on the persistent pipeline, several frames per launch.
// noAlloc: pt_render_frames_noalloc -- the buffers this function otherwise allocates on demand are slices of the device's
// scratch array, the grid is PT_NOALLOC_WAVES, no event is created and nothing is allocated or freed, on the device or
// through the HIP runtime; the caller has checked the limits.
int renderPersistent(pt_ctx *c, int n, bool noAlloc = false)
{
	const PtTile tile = c->tile.w ? c->tile : PtTile{ 0, 0, c->W, c->H };
	const uint32_t WH = tile.w * tile.h;  // pixels a dispatch covers
	const size_t nPadded = (size_t)((WH + PT_WAVE - 1) / PT_WAVE) * PT_WAVE * c->spp;  // whole pixel groups of 64
	// frames per launch: path ids q = item * 64 + lane stay below 2^32, state + radiance stay below the memory budget
	// the flat-scene kernel keeps no state record per path (ptLaunchPaths: the same test), so the budget buys it twice the frames
	const bool flatScene = c->dEntry && c->entryFlat && !(c->tlasDepth ? c->nInst : 0) && c->nIdx / 3 < 255u;
	const bool ringOnly = flatScene && PT_FLAT_CARRY_STATE != 0;
	const size_t pathBytes = ringOnly && !noAlloc ? sizeof(float4) : 2 * sizeof(float4);  // (the scratch array's slices are fixed)
	size_t perLaunch = (size_t)c->framesPerLaunch * (ringOnly && !c->framesPerLaunchSet ? 2 : 1);
	if (perLaunch * nPadded > 0xFFFFFF00ull) perLaunch = 0xFFFFFF00ull / nPadded;
	size_t budget = c->persistentBytes;
	if (noAlloc) budget = 2 * kNaRecBytes;
	if (perLaunch * nPadded * pathBytes > budget) perLaunch = budget / (nPadded * pathBytes);
	if (perLaunch < 1) perLaunch = 1;
	const size_t firstRequest = nPadded * (n > 1 ? 2 * perLaunch - 1 : (size_t)1);  // (at most what the loop below asks ensurePersistent for)
	if (!noAlloc && (firstRequest > c->pCap || (!ringOnly && firstRequest > c->pStateCap)))
	{
		// the buffers will have to grow (only then is the driver asked): never plan for more than half of what the device has
		// left, plus what these buffers already hold -- a frame-by-frame caller whose buffers are in place pays no query
		size_t freeB = 0, totalB = 0;
		if (hipMemGetInfo(&freeB, &totalB) == hipSuccess)
		{
			const size_t room = (freeB + (c->pCap + c->pStateCap) * sizeof(float4)) / 2;
			if (room < budget) budget = room;
			if (perLaunch * nPadded * pathBytes > budget) perLaunch = budget / (nPadded * pathBytes);
			if (perLaunch < 1) perLaunch = 1;
		}
	}
	if (nPadded > 0xFFFFFF00ull) return fail(c, PT_EINVAL, "pt_render_frame: %zu paths per frame exceed 2^32", nPadded);
	// frames ONE launch may hold at most (budget and path ids): a call that ends with less than a full launch beyond a full one
	// runs them together -- 20 frames are one launch of 20, not 16 + 4, where the short launch pays the whole drain (~0.3 ms)
	// for a quarter of the frames
	size_t maxFit = budget / (nPadded * pathBytes);
	if (maxFit * nPadded > 0xFFFFFF00ull) maxFit = 0xFFFFFF00ull / nPadded;
	hipStream_t s = c->stream;
	int rc = PT_OK;
	float4 *state = nullptr, *rad = nullptr, *ringO = nullptr, *ringD = nullptr;
	unsigned long long *lit = nullptr;
	const uint2 *lcgSkip = nullptr;
	uint32_t waves = c->cfg.bounceWaves;
	if (noAlloc)
	{
		unsigned char *base = c->naScratch;
		ringO = reinterpret_cast<float4 *>(base + kNaOffRingO);
		ringD = reinterpret_cast<float4 *>(base + kNaOffRingD);
		state = reinterpret_cast<float4 *>(base + kNaOffState);
		rad = reinterpret_cast<float4 *>(base + kNaOffRad);
		lit = reinterpret_cast<unsigned long long *>(base + kNaOffLit);
		// the table is rebuilt on the device for every call: another context on this GPU may have left its own behind
		ptLaunchLcgSkip((uint32_t)c->spp, reinterpret_cast<uint2 *>(base + kNaOffSkip), s);
		lcgSkip = reinterpret_cast<const uint2 *>(base + kNaOffSkip);
		if (waves > PT_NOALLOC_WAVES) waves = PT_NOALLOC_WAVES;
	}
	else
	{
		rc = ensureLcgSkip(c);
		if (rc) return rc;
		lcgSkip = c->dLcgSkip;
	}
	PtSceneDev sc = sceneDev(c);
	if (c->quadAccel.nQuads)
	{
		sc.nodes = c->quadAccel.nodes;
		sc.quads = c->quadAccel.quads;
		sc.nQuads = c->quadAccel.nQuads;
	}
	unsigned long long *trav = (c->profiling & PT_PROFILE_TRAVERSAL) ? c->dStats + 2 : nullptr;
#ifdef PT_PATHS_TIMING
	const bool statsKernel = trav != nullptr;
	trav = c->dStats + 2;  // tuning build: the section timers live behind the traversal counters
#else
	const bool statsKernel = trav != nullptr;
#endif
	while (n > 0)
	{
		uint32_t batch = (uint32_t)((size_t)n < perLaunch ? (size_t)n : perLaunch);
		if (!noAlloc && (size_t)n > perLaunch && (size_t)n < 2 * perLaunch && (size_t)n <= maxFit) batch = (uint32_t)n;
		// a multi-frame call sizes the buffers for full launches at once, so that a later, longer call (warm-up of 4 frames,
		// then 64) does not reallocate 8 GB in the middle of a render; frame-by-frame callers keep a one-frame buffer
		// (a second context on the GPU, or a 4K frame, may leave less room than the budget: halve and try again)
		if (!noAlloc)
		{
			bool oom = false;
			// (a multi-frame call sizes them for the longest launch a later call may make: a stretched one)
			size_t longest = 2 * perLaunch - 1 < maxFit ? 2 * perLaunch - 1 : maxFit;
			if (longest < batch) longest = batch;
			rc = ensurePersistent(c, nPadded * (n > 1 ? longest : (size_t)batch), flatScene, &oom);
			if (rc && oom && perLaunch > 1 && n > 1)
			{
				perLaunch = (perLaunch + 1) / 2;
				maxFit = perLaunch;  // (no stretched launches either)
				continue;
			}
			if (rc) return rc;
			state = c->pState;
			rad = c->pRad;
			lit = c->pLit;
			ringO = c->pRingO;
			ringD = c->pRingD;
		}
		const uint32_t frame0 = (uint32_t)(c->frame + 1);
		const uint32_t gframe0 = c->firstFrame + frame0 * c->frameStride;
		uint32_t centre = 0xFFFFFFFFu;
		for (uint32_t f = 0; f < batch; ++f)
			if (gframe0 + f * c->frameStride == 0) centre = f;
		ptLaunchFrameBegin(c->counters, (uint32_t)0, 0, s);
		hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
		if ((c->profiling & PT_PROFILE_KERNELS) && !noAlloc)
		{
			if (!getEventTriple(c, e0, e1, e2)) e0 = e1 = e2 = nullptr;
			else hipEventRecord(e0, s);
		}
		PtPathsArgs pa;
		pa.sc = sc;
		ptLaunchStoreCamera(c->cam, c->dCam, s);
		pa.cam = c->dCam;
		pa.W = c->W;
		pa.H = c->H;
		pa.tile = tile;
		pa.spp = (uint32_t)c->spp;
		pa.seedWord = c->seed + gframe0;
		pa.frameStride = c->frameStride;
		pa.nFrames = batch;
		pa.centreFrame = centre;
		pa.maxDepth = c->maxDepth;
		pa.tmin = c->tmin;
		pa.tmax = c->tmax;
		pa.lcgSkip = lcgSkip;
		pa.ringO = ringO;
		pa.ringD = ringD;
		pa.state = state;
		pa.rad = rad;
		pa.lit = lit;
		pa.itemHead = &c->counters->chunkHead[0];
		pa.waveRays = c->cfg.waveRays;
		pa.spill = c->cfg.spill;
		pa.spillStride = c->cfg.spillStride;
		pa.trav = trav;
		pa.instances = c->dInstances;
		pa.nInst = c->tlasDepth ? c->nInst : 0;
		pa.tlasRoot = c->tlasRoot;
		pa.instRoot = c->dInstRoot;
		pa.entry = c->dEntry;
		pa.entryFlat = c->entryFlat ? 1u : 0u;
		ptLaunchPaths(pa, waves, statsKernel, s);
		if (e1) hipEventRecord(e1, s);
		ptLaunchAccumulateItems(rad, lit, tile, c->W, (uint32_t)c->spp, c->image, frame0, batch, c->accumMode, s);
		if (e2)
		{
			hipEventRecord(e2, s);
			c->pending.push_back({ e0, e1, e2 });
		}
		ptLaunchFrameEnd(c->counters, 0, c->dStats, c->cfg.waveRays, c->cfg.bounceWaves, s);
		PT_HIP(c, hipGetLastError());
		c->frame += (int)batch;
		c->framesRendered += batch;
		c->lastLaunchFrames = batch;
		n -= (int)batch;
	}
	return PT_OK;
}

// One frame (c->frame, already advanced) on the per-bounce pipelines.
int renderClassic(pt_ctx *c)
{
	const PtTile tile = c->tile.w ? c->tile : PtTile{ 0, 0, c->W, c->H };
	const uint32_t WH = tile.w * tile.h;
	const size_t nPaths = (size_t)WH * c->spp;
	int rc = ensureQueues(c, nPaths);
	if (rc) return rc;
	const uint32_t gframe = c->firstFrame + (uint32_t)c->frame * c->frameStride;
	hipStream_t s = c->stream;
	PtSceneDev sc = sceneDev(c);
	const bool fused = (c->pipeline == PT_PIPELINE_FUSED) && !c->brute;
	unsigned long long *trav = (c->profiling & PT_PROFILE_TRAVERSAL) ? c->dStats + 2 : nullptr;
	ptLaunchFrameBegin(c->counters, (uint32_t)nPaths, fused ? 1 : 0, s);
	ptLaunchGenerate(c->cam, c->W, c->H, tile, (uint32_t)c->spp, c->seed + gframe, gframe == 0, c->q, s);
	int in = 0;
	for (int d = 0; d < c->maxDepth; ++d)
	{
		hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
		if (c->profiling & PT_PROFILE_KERNELS)
		{
			if (!getEventTriple(c, e0, e1, e2)) e0 = e1 = e2 = nullptr;
			else hipEventRecord(e0, s);
		}
		if (fused)
		{
			ptLaunchBounce(sc, c->q, in, c->counters, d, d == 0 ? (uint32_t)nPaths : 0u, c->tmin, c->tmax,
			               d == c->maxDepth - 1, c->cfg, trav, s);
			if (e1) hipEventRecord(e1, s);
		}
		else
		{
			ptLaunchExtend(sc, c->brute, &c->counters->count[d], &c->counters->headExtend[d], c->q.rayO[in],
			               c->q.rayD[in], c->q.hits, c->tmin, c->tmax, c->cfg, trav, s);
			if (e1) hipEventRecord(e1, s);
			ptLaunchShade(sc, c->dPrimEmissive, &c->counters->count[d], &c->counters->headShade[d], c->q, in,
			              &c->counters->count[d + 1], d == c->maxDepth - 1, c->cfg, s);
		}
		if (e2)
		{
			hipEventRecord(e2, s);
			c->pending.push_back({ e0, e1, e2 });
		}
		in ^= 1;
	}
	ptLaunchAccumulate(c->q, tile, c->W, (uint32_t)c->spp, c->image, (uint32_t)c->frame, c->accumMode, s);
	ptLaunchFrameEnd(c->counters, c->maxDepth, c->dStats, c->cfg.waveRays, c->cfg.bounceWaves, s);
	PT_HIP(c, hipGetLastError());
	c->framesRendered++;
	c->lastLaunchFrames = 1;
	return PT_OK;
}
}  // namespace

// VulkanRaytracing::raytrace, n times: the frame counter always advances (updateFrame); frames at or beyond
// max_frames are no-ops ("Max frames reached", VulkanRaytracing.cpp:877-886).
extern "C" int pt_render_frames(pt_ctx *c, int n) try
{
	int rc = checkReady(c, true);
	if (rc) return rc;
	if (n <= 0) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	const bool persistent = (c->pipeline == PT_PIPELINE_PERSISTENT) && !c->brute;
	if (c->nInst && !persistent)
		return fail(c, PT_EINVAL, "pt_render_frame: instanced scenes run on PT_PIPELINE_PERSISTENT with a BVH only");
	const long long next = (long long)c->frame + 1;
	long long active = n;
	if (c->maxFrames > 0) active = next >= c->maxFrames ? 0 : (c->maxFrames - next < n ? c->maxFrames - next : n);
	if (persistent)
	{
		rc = renderPersistent(c, (int)active);
		if (rc) return rc;
	}
	else
	{
		for (long long i = 0; i < active; ++i)
		{
			c->frame++;
			rc = renderClassic(c);
			if (rc) return rc;
		}
	}
	c->frame += (int)(n - active);
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_render_frames: out of host memory");
}

extern "C" int pt_render_frame(pt_ctx *c) { return pt_render_frames(c, 1); }

// pt_render_frames for callers that cannot allocate (include/pt_hip.h): same frames, same image, same counters, on the
// device's fixed scratch array.  No hipMalloc / hipFree / event creation on this path, and no host allocation either.
extern "C" int pt_render_frames_noalloc(pt_ctx *c, int n)
{
	int rc = checkReady(c, true);
	if (rc) return rc;
	if (n <= 0) return PT_OK;
	if (!c->naScratch) return fail(c, PT_EHIP, "pt_render_frames_noalloc: no scratch on device %d", c->device);
	if (c->pipeline != PT_PIPELINE_PERSISTENT || c->brute)
		return fail(c, PT_EINVAL, "pt_render_frames_noalloc: PT_PIPELINE_PERSISTENT over a BVH only");
	if ((uint32_t)c->spp > PT_NOALLOC_MAX_SPP)
		return fail(c, PT_ELIMIT, "pt_render_frames_noalloc: spp_per_frame %d exceeds %u", c->spp, PT_NOALLOC_MAX_SPP);
	const uint64_t pixels = c->tile.w ? (uint64_t)c->tile.w * c->tile.h : (uint64_t)c->W * c->H;
	const uint64_t records = (pixels + PT_WAVE - 1) / PT_WAVE * PT_WAVE * (uint64_t)c->spp;
	if (records > PT_NOALLOC_MAX_RECORDS)
		return fail(c, PT_ELIMIT,
		            "pt_render_frames_noalloc: %llu paths per frame (pixels rounded up to 64, times spp_per_frame) exceed the "
		            "scratch array's %u; render tiles (pt_set_tile) or use pt_render_frames",
		            (unsigned long long)records, PT_NOALLOC_MAX_RECORDS);
	PT_HIP(c, hipSetDevice(c->device));
	const long long next = (long long)c->frame + 1;
	long long active = n;
	if (c->maxFrames > 0) active = next >= c->maxFrames ? 0 : (c->maxFrames - next < n ? c->maxFrames - next : n);
	if (active > 0)
	{
		rc = renderPersistent(c, (int)active, true);
		if (rc) return rc;
	}
	c->frame += (int)(n - active);
	return PT_OK;
}

extern "C" int pt_set_pipeline(pt_ctx *c, int mode)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	if (mode != PT_PIPELINE_FUSED && mode != PT_PIPELINE_SPLIT && mode != PT_PIPELINE_PERSISTENT) return fail(c, PT_EINVAL, "pt_set_pipeline: bad mode %d", mode);
	c->pipeline = mode;
	return PT_OK;
}

extern "C" int pt_reset_accumulation(pt_ctx *c)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	c->frame = -1;
	return PT_OK;
}

extern "C" int pt_frame_index(pt_ctx *c) { return c ? c->frame : -1; }

extern "C" int pt_read_image(pt_ctx *c, float *rgba)
{
	if (!c || !rgba) return fail(c, PT_EINVAL, "pt_read_image: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_read_image: no image (pt_resize)");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipMemcpyAsync(rgba, c->image, sizeof(float4) * (size_t)c->W * c->H, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	drainEvents(c);
	return PT_OK;
}

extern "C" int pt_image_device_ptr(pt_ctx *c, void **p)
{
	if (!c || !p) return fail(c, PT_EINVAL, "pt_image_device_ptr: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_image_device_ptr: no image (pt_resize)");
	*p = c->image;
	return PT_OK;
}

extern "C" int pt_write_image_from_device(pt_ctx *c, const void *src)
{
	if (!c || !src) return fail(c, PT_EINVAL, "pt_write_image_from_device: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_write_image_from_device: no image (pt_resize)");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipMemcpyAsync(c->image, src, sizeof(float4) * (size_t)c->W * c->H, hipMemcpyDeviceToDevice, c->stream));
	return PT_OK;
}

extern "C" int pt_write_image(pt_ctx *c, const float *rgba)
{
	if (!c || !rgba) return fail(c, PT_EINVAL, "pt_write_image: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_write_image: no image (pt_resize)");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipMemcpyAsync(c->image, rgba, sizeof(float4) * (size_t)c->W * c->H, hipMemcpyHostToDevice, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));  // the caller's buffer is free to go when the call returns
	return PT_OK;
}

// ---- accumulation checkpoint ----
namespace
{
struct CheckpointHeader
{
	char magic[8];  // "PTCKPT03" ("PTCKPT02": without tileBlock; "PTCKPT01": without the tile, read as the whole image)
	uint32_t W, H;
	int32_t frame, accumMode;
	uint32_t firstFrame, frameStride, seed;
	int32_t spp, maxDepth, maxFrames;
	float tmin, tmax;
	uint32_t tile[4];  // x0, y0, w, h of pt_set_tile; w = 0: the whole image
	uint32_t tileBlock[2];  // "PTCKPT03": block width and stride of pt_set_tile_interleaved (0, 0: a rectangle)
};
constexpr size_t kCheckpointHeaderV1 = offsetof(CheckpointHeader, tile);
constexpr size_t kCheckpointHeaderV2 = offsetof(CheckpointHeader, tileBlock);
}  // namespace

extern "C" int pt_save_checkpoint(pt_ctx *c, const char *path)
{
	if (!c || !path) return fail(c, PT_EINVAL, "pt_save_checkpoint: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_save_checkpoint: no image (pt_resize)");
	try
	{
		std::vector<float> px(4 * (size_t)c->W * c->H);
		int rc = pt_read_image(c, px.data());
		if (rc) return rc;
		CheckpointHeader h;
		memcpy(h.magic, "PTCKPT03", 8);
		h.W = c->W;
		h.H = c->H;
		h.frame = c->frame;
		h.accumMode = c->accumMode;
		h.firstFrame = c->firstFrame;
		h.frameStride = c->frameStride;
		h.seed = c->seed;
		h.spp = c->spp;
		h.maxDepth = c->maxDepth;
		h.maxFrames = c->maxFrames;
		h.tmin = c->tmin;
		h.tmax = c->tmax;
		h.tile[0] = c->tile.x0;
		h.tile[1] = c->tile.y0;
		h.tile[2] = c->tile.w;
		h.tile[3] = c->tile.h;
		h.tileBlock[0] = c->tile.bw;
		h.tileBlock[1] = c->tile.stride;
		FILE *f = fopen(path, "wb");
		if (!f) return fail(c, PT_EIO, "pt_save_checkpoint: cannot open %s", path);
		bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(px.data(), sizeof(float), px.size(), f) == px.size();
		ok = (fclose(f) == 0) && ok;
		return ok ? PT_OK : fail(c, PT_EIO, "pt_save_checkpoint: write to %s failed", path);
	}
	catch (const std::bad_alloc &)
	{
		return fail(c, PT_ENOMEM, "pt_save_checkpoint: out of host memory");
	}
}

// Everything is read and validated BEFORE the context is touched: a file that is refused leaves the context as it was.
extern "C" int pt_load_checkpoint(pt_ctx *c, const char *path)
{
	if (!c || !path) return fail(c, PT_EINVAL, "pt_load_checkpoint: null argument");
	FILE *f = fopen(path, "rb");
	if (!f) return fail(c, PT_EIO, "pt_load_checkpoint: cannot open %s", path);
	CheckpointHeader h;
	memset(&h, 0, sizeof(h));
	size_t headerBytes = 0;
	if (fread(&h, kCheckpointHeaderV1, 1, f) == 1)
	{
		if (memcmp(h.magic, "PTCKPT01", 8) == 0) headerBytes = kCheckpointHeaderV1;
		else if (memcmp(h.magic, "PTCKPT02", 8) == 0 && fread(h.tile, sizeof(h.tile), 1, f) == 1) headerBytes = kCheckpointHeaderV2;
		else if (memcmp(h.magic, "PTCKPT03", 8) == 0 && fread(h.tile, sizeof(h.tile) + sizeof(h.tileBlock), 1, f) == 1)
			headerBytes = sizeof(h);
	}
	if (!headerBytes || !h.W || !h.H || (uint64_t)h.W * h.H > 0x3FFFFFFFull)
	{
		fclose(f);
		return fail(c, PT_EINVAL, "pt_load_checkpoint: %s is not a checkpoint", path);
	}
	const char *bad = nullptr;
	if (h.accumMode != PT_ACCUM_MEAN && h.accumMode != PT_ACCUM_SUM) bad = "bad accumulation mode";
	else if (!h.frameStride) bad = "frame stride 0";
	else if (h.frame < -1 || h.frame > 0x7FFFFFFF - 4096) bad = "frame index out of range";
	else if (h.spp < 1 || h.maxDepth < 1 || h.maxDepth > PT_MAX_DEPTH) bad = "sampling parameters out of range";
	else if (!(h.tmin >= 0.0f) || !(h.tmax > h.tmin)) bad = "ray interval out of range";
	else if ((uint64_t)h.W * h.H * (uint64_t)h.spp > 0xFFFFFFF0ull) bad = "W*H*spp_per_frame exceeds 2^32 path ids";
	else if (h.tileBlock[0] || h.tileBlock[1])
	{
		// an interleaved tile: its last column must lie inside the image (pt_tile_column of column w - 1)
		const uint64_t bw = h.tileBlock[0], st = h.tileBlock[1], w = h.tile[2];
		if (!bw || !st || !w || h.tile[1] != 0 || h.tile[3] != h.H || h.tile[0] % bw || h.tile[0] / bw >= st ||
		    h.tile[0] + ((w - 1) / bw) * st * bw + (w - 1) % bw >= h.W)
			bad = "interleaved tile does not fit the image";
	}
	else if (h.tile[2] || h.tile[3])
	{
		if (!h.tile[2] || !h.tile[3] || (uint64_t)h.tile[0] + h.tile[2] > h.W || (uint64_t)h.tile[1] + h.tile[3] > h.H)
			bad = "tile does not fit the image";
	}
	// the payload must be there before anything of its size is allocated
	const uint64_t payload = 16ull * h.W * h.H;
	if (!bad)
	{
		long here = ftell(f);
		if (here < 0 || fseek(f, 0, SEEK_END) != 0) bad = "cannot seek";
		else
		{
			long end = ftell(f);
			if (end < 0 || (uint64_t)(end - here) != payload) bad = "truncated or oversized payload";
			else if (fseek(f, here, SEEK_SET) != 0) bad = "cannot seek";
		}
	}
	if (bad)
	{
		fclose(f);
		return fail(c, PT_EINVAL, "pt_load_checkpoint: %s: %s", path, bad);
	}
	std::vector<float> px;
	try
	{
		px.resize(4 * (size_t)h.W * h.H);
	}
	catch (const std::bad_alloc &)
	{
		fclose(f);
		return fail(c, PT_ENOMEM, "pt_load_checkpoint: out of host memory for %u x %u", h.W, h.H);
	}
	size_t got = fread(px.data(), sizeof(float), px.size(), f);
	fclose(f);
	if (got != px.size()) return fail(c, PT_EINVAL, "pt_load_checkpoint: %s is truncated", path);
	// the image first (the only step that can still fail: device memory), then the scalars
	int rc = pt_resize(c, h.W, h.H);  // also drops frames still in flight; resets the tile
	if (rc) return rc;
	PT_HIP(c, hipMemcpy(c->image, px.data(), sizeof(float) * px.size(), hipMemcpyHostToDevice));
	c->spp = h.spp;
	c->maxDepth = h.maxDepth;
	c->tmin = h.tmin;
	c->tmax = h.tmax;
	c->seed = h.seed;
	c->maxFrames = h.maxFrames;
	c->firstFrame = h.firstFrame;
	c->frameStride = h.frameStride;
	c->accumMode = h.accumMode;
	c->tile = PtTile{ h.tile[0], h.tile[1], h.tile[2], h.tile[3], h.tileBlock[0], h.tileBlock[1] };
	c->frame = h.frame;  // the next pt_render_frame renders frame + 1, as if the run had never stopped
	return PT_OK;
}

extern "C" int pt_read_image_srgb8(pt_ctx *c, uint8_t *rgba8) try
{
	if (!c || !rgba8) return fail(c, PT_EINVAL, "pt_read_image_srgb8: null argument");
	if (!c->image) return fail(c, PT_EINVAL, "pt_read_image_srgb8: no image (pt_resize)");
	PT_HIP(c, hipSetDevice(c->device));
	const uint32_t n = c->W * c->H;
	void *tmp = nullptr;
	PT_HIP(c, hipMalloc(&tmp, 4 * (size_t)n));
	ptLaunchSrgb8(c->image, n, tmp, c->stream);
	hipError_t e = hipMemcpyAsync(rgba8, tmp, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	hipFree(tmp);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_read_image_srgb8: %s", hipGetErrorString(e));
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	return fail(c, PT_ENOMEM, "pt_read_image_srgb8: out of host memory");
}

extern "C" int pt_set_profiling(pt_ctx *c, int flags)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	c->profiling = flags;
	return PT_OK;
}

extern "C" int pt_get_stats(pt_ctx *c, PtStats *out)
{
	if (!c || !out) return fail(c, PT_EINVAL, "pt_get_stats: null argument");
	PT_HIP(c, hipSetDevice(c->device));
	unsigned long long h[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
	PT_HIP(c, hipMemcpyAsync(h, c->dStats, sizeof(h), hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	drainEvents(c);
	memset(out, 0, sizeof(*out));
	out->rays_total = h[0];
	out->rays_last_frame = h[1] / c->lastLaunchFrames;  // mean over the frames of the last launch
	out->paths_last_frame = (c->tile.w ? (uint64_t)c->tile.w * c->tile.h : (uint64_t)c->W * c->H) * (uint64_t)c->spp;
	out->frames_rendered = c->framesRendered;
	out->extend_ms_total = c->extendMs;
	out->extend_launches = c->extendLaunches;
	out->shade_ms_total = c->shadeMs;
	out->shade_launches = c->shadeLaunches;
	out->bvh_nodes = c->accel.nNodes;
	out->bvh_depth = c->accel.depth;
	out->bvh_tris = c->accel.nTris;
	out->build_ms = c->buildMs;
	out->accel_bytes = c->accelBytes;
	out->accel_bytes_uncompacted = c->accelBytesUncompacted;
	out->quad_leaves = c->quadAccel.nQuads;
	out->trav_nodes = h[2];
	out->trav_tris = h[3];
	out->trav_hits = h[4];
	out->trav_leaf_visits = h[5];
	out->trav_node_steps = h[6];
	out->trav_leaf_steps = h[7];
#ifdef PT_PATHS_TIMING
	{
		unsigned long long t[16] = {};
		hipMemcpy(t, c->dStats + 8, sizeof(t), hipMemcpyDeviceToHost);
		double sum = 0;
		for (unsigned long long x : t) sum += (double)x;
		static const char *names[16] = { "loop", "retire", "flush", "ring", "gen", "take", "node", "leaf", "s:prepare", "s:entry", "s:leaf",
			                             "s:store", "f:loads", "f:math", "f:store", "-" };
		fprintf(stderr, "[k_paths section shares]");
		for (int i = 0; i < 16; ++i) fprintf(stderr, " %s %.1f%%", names[i], sum > 0 ? 100.0 * (double)t[i] / sum : 0.0);
		fprintf(stderr, "  (%.3g wave-cycles)\n", sum);
	}
#endif
	return PT_OK;
}

extern "C" int pt_reset_stats(pt_ctx *c)
{
	if (!c) return fail(nullptr, PT_EINVAL, "null context");
	PT_HIP(c, hipSetDevice(c->device));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	drainEvents(c);
	PT_HIP(c, hipMemset(c->dStats, 0, 32 * sizeof(unsigned long long)));
	c->extendMs = c->shadeMs = 0.0;
	c->extendLaunches = c->shadeLaunches = c->framesRendered = 0;
	return PT_OK;
}

// ------------------------------------------------------------------- debug entry points ----
extern "C" int pt_debug_rng(pt_ctx *c, uint32_t n, const uint32_t *v0, const uint32_t *v1, uint32_t nSteps,
                            uint32_t *outTea, uint32_t *outLcg, float *outUnit)
{
	if (!c || !v0 || !v1 || !outTea || !outLcg || !outUnit) return fail(c, PT_EINVAL, "pt_debug_rng: null argument");
	if (!n) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	uint32_t *d0 = nullptr, *d1 = nullptr, *dt = nullptr, *dl = nullptr;
	float *du = nullptr;
	hipError_t e = hipMalloc(&d0, 4 * (size_t)n);
	if (e == hipSuccess) e = hipMalloc(&d1, 4 * (size_t)n);
	if (e == hipSuccess) e = hipMalloc(&dt, 4 * (size_t)n);
	if (e == hipSuccess) e = hipMalloc(&dl, 4 * (size_t)n * (nSteps ? nSteps : 1));
	if (e == hipSuccess) e = hipMalloc(&du, 12 * (size_t)n);
	if (e == hipSuccess) e = hipMemcpy(d0, v0, 4 * (size_t)n, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d1, v1, 4 * (size_t)n, hipMemcpyHostToDevice);
	if (e == hipSuccess)
	{
		ptLaunchDebugRng(n, d0, d1, nSteps, dt, dl, du, c->stream);
		e = hipStreamSynchronize(c->stream);
	}
	if (e == hipSuccess) e = hipMemcpy(outTea, dt, 4 * (size_t)n, hipMemcpyDeviceToHost);
	if (e == hipSuccess && nSteps) e = hipMemcpy(outLcg, dl, 4 * (size_t)n * nSteps, hipMemcpyDeviceToHost);
	if (e == hipSuccess) e = hipMemcpy(outUnit, du, 12 * (size_t)n, hipMemcpyDeviceToHost);
	hipFree(d0); hipFree(d1); hipFree(dt); hipFree(dl); hipFree(du);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_debug_rng: %s", hipGetErrorString(e));
	return PT_OK;
}

extern "C" int pt_debug_math(pt_ctx *c, uint64_t n, uint32_t seed, uint64_t mismatches[4])
{
	if (!c || !mismatches) return fail(c, PT_EINVAL, "pt_debug_math: null argument");
	if (n > (1ull << 38)) return fail(c, PT_ELIMIT, "pt_debug_math: at most 2^38 samples");
	mismatches[0] = mismatches[1] = mismatches[2] = mismatches[3] = 0;
	if (!n) return PT_OK;
	PT_HIP(c, hipSetDevice(c->device));
	unsigned long long *d = nullptr;
	hipError_t e = hipMalloc(&d, 4 * sizeof(unsigned long long));
	if (e == hipSuccess) e = hipMemsetAsync(d, 0, 4 * sizeof(unsigned long long), c->stream);
	if (e == hipSuccess)
	{
		ptLaunchDebugMath(n, seed, d, c->stream);
		e = hipStreamSynchronize(c->stream);
	}
	unsigned long long h[4] = { 0, 0, 0, 0 };
	if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	hipFree(d);
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_debug_math: %s", hipGetErrorString(e));
	for (int k = 0; k < 4; ++k) mismatches[k] = h[k];
	return PT_OK;
}

extern "C" int pt_debug_primary_hits(pt_ctx *c, PtHit *hits)
{
	int rc = checkReady(c, true);
	if (rc) return rc;
	if (!hits) return fail(c, PT_EINVAL, "pt_debug_primary_hits: null argument");
	PT_HIP(c, hipSetDevice(c->device));
	const uint32_t WH = c->W * c->H;
	rc = ensureQueues(c, (size_t)WH * c->spp);
	if (rc) return rc;
	// frame-0 rays: jitter = pixel centre (raytrace.rgen:38), one sample per pixel
	ptLaunchGenerate(c->cam, c->W, c->H, PtTile{ 0, 0, c->W, c->H }, 1, c->seed, 1, c->q, c->stream);
	rc = traceQueue0(c, WH, c->tmin, c->tmax);
	if (rc) return rc;
	PT_HIP(c, hipMemcpyAsync(hits, c->q.hits, sizeof(PtHit) * (size_t)WH, hipMemcpyDeviceToHost, c->stream));
	PT_HIP(c, hipStreamSynchronize(c->stream));
	return PT_OK;
}

extern "C" int pt_debug_trace(pt_ctx *c, const float *rays, uint32_t n, float tmin, float tmax, PtHit *hits)
{
	int rc = checkReady(c, false);
	if (rc) return rc;
	if (!rays || !hits) return fail(c, PT_EINVAL, "pt_debug_trace: null argument");
	if (!n) return PT_OK;
	if (!(tmin >= 0.0f) || !(tmax > tmin)) return fail(c, PT_EINVAL, "pt_debug_trace: need 0 <= tmin < tmax");
	PT_HIP(c, hipSetDevice(c->device));
	rc = ensureQueues(c, n);
	if (rc) return rc;
	float *dr = nullptr;
	PT_HIP(c, hipMalloc(&dr, 24 * (size_t)n));
	hipError_t e = hipMemcpyAsync(dr, rays, 24 * (size_t)n, hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess)
	{
		ptLaunchPackRays(dr, n, c->q.rayO[0], c->q.rayD[0], c->stream);
		rc = traceQueue0(c, n, tmin, tmax);
		if (rc == PT_OK) e = hipMemcpyAsync(hits, c->q.hits, sizeof(PtHit) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	hipFree(dr);
	if (rc) return rc;
	if (e != hipSuccess) return fail(c, PT_EHIP, "pt_debug_trace: %s", hipGetErrorString(e));
	return PT_OK;
}

extern "C" int pt_debug_morton(pt_ctx *c, uint32_t *codes, uint32_t *order, float cmin[3], float cmax[3])
{
	int rc = checkReady(c, false);
	if (rc) return rc;
	const uint32_t n = c->nIdx / 3;
	PT_HIP(c, hipSetDevice(c->device));
	if (codes && n) PT_HIP(c, hipMemcpy(codes, c->accel.codes, 4 * (size_t)n, hipMemcpyDeviceToHost));
	if (order && n) PT_HIP(c, hipMemcpy(order, c->accel.order, 4 * (size_t)n, hipMemcpyDeviceToHost));
	for (int k = 0; k < 3; ++k)
	{
		if (cmin) cmin[k] = c->accel.cmin[k];
		if (cmax) cmax[k] = c->accel.cmax[k];
	}
	return PT_OK;
}

extern "C" int pt_debug_read_bvh(pt_ctx *c, void *nodes, uint32_t *nNodes, void *tris, uint32_t *nTris)
{
	int rc = checkReady(c, false);
	if (rc) return rc;
	PT_HIP(c, hipSetDevice(c->device));
	if (nNodes) *nNodes = c->accel.nNodes;
	if (nTris) *nTris = c->accel.nTris;
	if (nodes && c->accel.nNodes)
		PT_HIP(c, hipMemcpy(nodes, c->accel.nodes, sizeof(PtNode4) * (size_t)c->accel.nNodes, hipMemcpyDeviceToHost));
	if (tris && c->accel.nTris)
		PT_HIP(c, hipMemcpy(tris, c->accel.tris + (size_t)(c->accel.triPlanes - 1) * c->accel.nTris, sizeof(PtTri48) * (size_t)c->accel.nTris, hipMemcpyDeviceToHost));
	return PT_OK;
}
