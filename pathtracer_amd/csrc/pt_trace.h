// Host-callable launchers of the wavefront path-tracing kernels (pt_trace.hip).
#pragma once
#include "pt_device_common.h"

#define PT_MAX_DEPTH 64     // upper bound for max_depth (reference: 20)
#define PT_SPILL_STACK 72   // traversal stack entries per lane beyond the LDS-resident ones
#define PT_STACK_CAPACITY (10 + PT_SPILL_STACK)  // smallest traversal stack of any kernel: LDS entries + spill
#define PT_CHUNK 1024u      // queue records per chunk (work-claim and compaction granularity)
// k_paths on flat scenes (every child of the entry node a leaf): throughput and RNG state travel with the ray (third ring
// record, four more staging and done rows) instead of living in a per-path table that the shading pass has to gather from.
// Kernel AND host read this switch: with it the host allocates only the ring part of the state buffer for such a scene
// (pt_api.hip, ensurePersistent); a kernel built without it writes one state record per path.
#ifndef PT_FLAT_CARRY_STATE
#define PT_FLAT_CARRY_STATE 1
#endif

// Part of the image a dispatch covers (default = the whole image); seeds always use the full-image linear index y * W + x
// (rgen:28), so a pixel's value does not depend on the tiling.
//   bw == 0 (pt_set_tile): the w x h rectangle at (x0, y0); tile pixel t = image pixel (x0 + t % w, y0 + t / w).
//   bw != 0 (pt_set_tile_interleaved): blocks of bw columns dealt round-robin -- this tile holds every `stride`-th block
//   beginning with the one at column x0; w counts the tile's columns (the image's last block may be cut short), and tile
//   column c = image column x0 + (c / bw) * stride * bw + c % bw.
struct PtTile
{
	uint32_t x0, y0, w, h;
	uint32_t bw, stride;
};
__host__ __device__ inline uint32_t pt_tile_column(const PtTile &t, uint32_t c)
{
	return t.bw ? t.x0 + (c / t.bw) * t.stride * t.bw + c % t.bw : t.x0 + c;
}

// Per-frame device counters, zeroed by k_frame_begin.  count[d] = rays queued for bounce d.
struct PtFrameCounters
{
	uint32_t count[PT_MAX_DEPTH + 1];       // split pipeline: queue length; fused: rays traced at bounce d
	uint32_t headExtend[PT_MAX_DEPTH + 1];
	uint32_t headShade[PT_MAX_DEPTH + 1];
	uint32_t numChunks[PT_MAX_DEPTH + 2];   // fused pipeline: chunks in the queue of bounce d
	uint32_t chunkHead[PT_MAX_DEPTH + 1];   // fused pipeline: next input chunk to claim
};

// Path/ray queues in HBM (SoA float4 streams, capacity = W*H*spp_per_frame records each).
struct PtQueues
{
	float4 *rayO[2];  // (origin.xyz, rng state)          ping-pong across bounces
	float4 *rayD[2];  // (direction.xyz, path id)
	float4 *thr[2];   // (throughput.xyz, -)
	float4 *hits;     // PtHit per queued ray
	float4 *rad;      // radiance per path id (sample-major: id = sample*W*H + pixel)
	uint32_t *chunkCount[2];  // fused pipeline: valid records at the front of each 1024-record chunk
};

struct PtLaunchCfg
{
	uint32_t persistentBlocks;  // grid of the persistent extend/shade kernels
	uint32_t *spill;            // persistentBlocks*256*PT_SPILL_STACK words
	uint32_t spillStride;       // lanes the spill strip is laid out for
	uint32_t bounceWaves;       // grid of the fused k_bounce kernel (single-wave blocks)
	uint32_t *waveRays;         // bounceWaves counters: rays fetched by each wave slot during the frame
};

void ptLaunchFrameBegin(PtFrameCounters *fc, uint32_t nPaths, int fused, hipStream_t s);
void ptLaunchFrameEnd(const PtFrameCounters *fc, int depth, unsigned long long *stats, uint32_t *waveRays,
                      uint32_t nWaveRays, hipStream_t s);
void ptLaunchGenerate(const PtCamera &cam, uint32_t W, uint32_t H, PtTile tile, uint32_t spp, uint32_t seedWord, int centreJitter,
                      const PtQueues &q, hipStream_t s);
void ptLaunchExtend(const PtSceneDev &sc, bool brute, const uint32_t *count, uint32_t *head, const float4 *rayO,
                    const float4 *rayD, float4 *hits, float tmin, float tmax, const PtLaunchCfg &cfg,
                    unsigned long long *travStats /* null = uninstrumented kernel */, hipStream_t s);
void ptLaunchShade(const PtSceneDev &sc, const uint8_t *primEmissive, const uint32_t *count, uint32_t *head,
                   const PtQueues &q, int in, uint32_t *countOut, int lastBounce, const PtLaunchCfg &cfg, hipStream_t s);
// Fused extend+shade+compact for bounce `depth`: reads queue `in`, writes queue `in^1`.  denseN != 0: the
// input queue is the dense output of k_generate with denseN records (chunk counts implied).
void ptLaunchBounce(const PtSceneDev &sc, const PtQueues &q, int in, PtFrameCounters *fc, int depth, uint32_t denseN,
                    float tmin, float tmax, int lastBounce, const PtLaunchCfg &cfg, unsigned long long *travStats,
                    hipStream_t s);
// Persistent path kernel (pt_paths.hip): one launch traces every path of the frame from camera ray to
// termination.  Buffers: ringO/ringD = waves * PT_PATHS_RING records (wave-private survivor rings), state and
// rad = nFrames * ceil(tile pixels/64) * 64 * spp records in item-major order q = ((t / 64) * spp + sample) * 64 + t % 64
// (t = pixel index inside the tile), lit = one 64-bit word per 64 of them.  A flat scene uses the first waves *
// PT_PATHS_RING records of `state` as a third ring (throughput, RNG state) and the rest of it not at all.
#define PT_PATHS_RING 256u
struct PtPathsArgs
{
	PtSceneDev sc;
	// in device memory (ptLaunchStoreCamera), not in the argument block: 32 SGPRs that only the camera-ray generation
	// needs would otherwise be held -- and spilled around -- through the whole kernel
	const PtCamera *cam;
	uint32_t W, H, spp;
	// One launch renders nFrames consecutive frames of the schedule (the drain at the end of a launch -- waves
	// finishing their last paths at low lane utilisation, ~0.3 ms -- is paid once per launch, not per frame):
	// frame f of the launch seeds with seedWord + f * frameStride (rgen:28, val1 = seed + global frame) and uses
	// the pixel-centre jitter iff f == centreFrame (global frame 0; 0xFFFFFFFF: none).
	uint32_t seedWord, frameStride, nFrames, centreFrame;
	PtTile tile;
	int maxDepth;
	float tmin, tmax;
	const uint2 *lcgSkip;  // [spp]: (A, C) with lcg^(2i)(s) = A * s + C
	float4 *ringO, *ringD, *state, *rad;
	// one bit per path of the launch (word = work item, bit = lane): the path has added light to its radiance record.  The
	// generator clears an item's word instead of zeroing 64 records; k_accumulate_items reads a record only where the bit is set.
	unsigned long long *lit;
	uint32_t *itemHead;    // zeroed per frame
	uint32_t *waveRays;
	uint32_t *spill;
	uint32_t spillStride;
	unsigned long long *trav;  // traversal statistics (stats variant only)
	// two-level scenes (pt_set_instances): 24 floats per instance (object-to-world 3x4, world-to-object 3x4, row-major);
	// sc.nodes holds the bottom-level tree at index 0 and the top-level tree at tlasRoot; nInst == 0: single level
	const float *instances;
	uint32_t nInst, tlasRoot;
	const uint32_t *instRoot;  // per instance: root node of the bottom-level tree of its Model (pt_upload_models)
	const uint4 *entry;  // entry node (PT_ENTRY_REC 16-byte units, pt_wave_traverse.h) or nullptr: every ray starts at the root
	uint32_t entryFlat;  // every child of the entry node is a leaf: the whole scene is that one node
};
void ptLaunchPaths(const PtPathsArgs &args, uint32_t waves, bool stats, hipStream_t s);
void ptLaunchStoreCamera(const PtCamera &cam, PtCamera *dst, hipStream_t s);
// dst[i] = (A, C) with lcg^(2i)(s) = A * s + C, i < spp: the skip-ahead table of PtPathsArgs::lcgSkip, computed on the device
void ptLaunchLcgSkip(uint32_t spp, uint2 *dst, hipStream_t s);
void ptLaunchAccumulateItems(const float4 *rad, const unsigned long long *lit, PtTile tile, uint32_t W, uint32_t spp, float4 *image,
                             uint32_t lframe0, uint32_t nFrames, int mode, hipStream_t s);
void ptLaunchAccumulate(const PtQueues &q, PtTile tile, uint32_t W, uint32_t spp, float4 *image, uint32_t lframe, int mode,
                        hipStream_t s);
void ptLaunchDebugRng(uint32_t n, const uint32_t *v0, const uint32_t *v1, uint32_t nSteps, uint32_t *outTea,
                      uint32_t *outLcg, float *outUnit, hipStream_t s);
void ptLaunchDebugMath(unsigned long long n, uint32_t seed, unsigned long long *mismatches, hipStream_t s);
void ptLaunchPackRays(const float *rays, uint32_t n, float4 *rayO, float4 *rayD, hipStream_t s);
void ptLaunchSetU32(uint32_t *p, uint32_t v, hipStream_t s);
void ptLaunchBuildShadeRecords(const PtSceneDev &sc, PtShadeRec *out, hipStream_t s);
void ptLaunchSrgb8(const float4 *image, uint32_t n, void *out, hipStream_t s);
