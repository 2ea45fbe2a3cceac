// k_paths: a whole frame of paths in ONE persistent kernel -- raytrace.rgen:28-80 (seed, jitter, camera
// ray, the bounce loop), traceRayEXT, raytrace.rchit / raytrace.rmiss -- "path regeneration inside the wave".
//
// Why.  With one k_bounce launch per bounce (pt_bounce.hip) every bounce pays a tail (waves finish a claim
// of >= 256 rays ~ 60 us apart: measured ~80 us fixed cost per launch, 10 % of the frame), k_generate is a
// separate HBM-bound pass (6 %), and all survivors make a round trip through global chunked queues.  Here
//   * a wave generates its primary rays itself, 64 at a time at full lane utilisation (the staging phase
//     of k_bounce), from work items claimed with one atomic per up to 8 items;
//   * lanes whose traversal ended are refilled inside the wave, finished hits are parked in an LDS ring
//     and shaded 64 at a time (both exactly as in k_bounce);
//   * survivors of the shading go to a small WAVE-PRIVATE ring in global memory (256 records, no atomics,
//     no chunk bookkeeping) and come back through the staging phase, preferred over new paths -- so a path
//     stays in the wave that started it from its first ray to its last;
//   * the per-path state that is only needed at a hit (throughput, rng state) lives in one 16-byte record
//     state[q], q = item * 64 + lane (a staging batch touches 1 KB contiguous) -- or, for a flat scene (every
//     child of the entry node a leaf), travels with the ray: a third ring record, four more staging and done
//     rows, four registers of the lane, and no table the shading pass would have to gather from;
//   * the radiance rad[q] is written only by a path that reaches a light; lit[item] holds one bit per path
//     ("written") that the generator clears and k_accumulate_items consults before it reads a record.
// The ring holds at most 254 records.  Rays of a wave are in the ring (R), in the staging area (S), in lanes (L <= 64) or parked
// for the shading pass (D); only the generator adds rays, 64 at a time, and it runs only when R <= 63 (survivors are preferred),
// S = 0 (the staging area is empty) and D <= 63 (64 parked hits are shaded before anything else is staged): R + S + L + D <= 190
// before it and <= 254 after it, and every other step moves rays between the four or drops them.
// One tail per frame instead of one per bounce.  Every arithmetic step that reaches a result is the shared
// specification (pt_device_common.h, pt_wave_traverse.h): images are bit-identical to the other pipelines
// and to the CPU oracle.
//
// Work item = (pixel group g of 64 consecutive pixels, sample i): item = g * spp + i.  The raygen seed chain
// of a pixel is sequential over its samples (rgen:28-37: seed = tea(pixel, frame); two rnd() per sample); a
// wave that processes consecutive items of one group continues the chain in a register, a wave that starts
// in the middle of a group (claims are not aligned to groups) jumps there with the LCG skip-ahead pair
// (A^2i, C_2i) from a table the host computes.
#include <stdlib.h>

#include <type_traits>

#include "pt_trace.h"
#include "pt_wave_traverse.h"

#ifndef PT_PATHS_MIN_ITEMS
#define PT_PATHS_MIN_ITEMS 2u
#endif
#ifndef PT_PATHS_MAX_ITEMS
#define PT_PATHS_MAX_ITEMS 8u
#endif
#ifndef PT_REFILL_LANES
#define PT_REFILL_LANES 32
#endif

namespace
{
#ifndef PT_STACK_LDS
#define PT_STACK_LDS 12
#endif
#ifdef PT_PATHS_WAVES_PER_EU  // tuning builds: ask the compiler for that many waves per SIMD (register budget)
#define PT_PATHS_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PT_PATHS_WAVES_PER_EU, PT_PATHS_WAVES_PER_EU)))
#else
#define PT_PATHS_OCCUPANCY
#endif
constexpr int kStackLdsFull = PT_STACK_LDS;   // traversal stack entries per lane held in LDS; deeper ones spill to global
// the variants that hold a node copy AND the entry node in LDS keep 16 waves per CU (10 240 B each) with a shorter stack
// and done ring: their trees have at most 16 nodes (depth <= 3 below the entry node: 9 pending entries + the sentinel)
constexpr int kStackLdsTiny = 10;
constexpr int kInFields = 11;   // ox oy oz ix iy iz Sx Sy Sz kpack(kz | signs | depth << 16) q   (+ entry children hit) (+ dx dy dz: two-level)
constexpr int kDoneFields = 6;  // V W det prim q kpack                                          (+ instance: two-level)
constexpr int kSaveFields = 7;  // two-level: prepared world ray (ix iy iz Sx Sy Sz kpack) while a lane is inside an instance
#ifndef PT_DONE_RING
#define PT_DONE_RING 128
#endif
constexpr int kDoneRingFull = PT_DONE_RING;  // >= 64 entries; below 127 a retire may have to shade a partial batch first
#ifndef PT_DONE_RING_TINY
#define PT_DONE_RING_TINY 112
#endif
#ifndef PT_INST_RING
#define PT_INST_RING 80
#endif
#ifndef PT_INST_VOTE3
#define PT_INST_VOTE3 0
#endif
#ifndef PT_INST_STACK
#define PT_INST_STACK 10
#endif
// two-level scenes, A/B switches (profiles/r03b..r03f_ab*.log):
//   PT_INST_EXIT_HEAD   a lane that pops PT_INSTANCE_END leaves the instance at the head of the next iteration (restore + pop)
//                       instead of spending a leaf step of its own on it: teapot grid +9.4 %
//   PT_INST_ENTRY_HEAD  K > 0: lanes at a top-level leaf enter their instance at the head of the iteration as soon as K of them
//                       wait (and then take part in the vote as lanes at the bottom-level root); 0: inside the leaf step.
//                       Measured K = 1 / 4 / 8 / 16: -11 % / -5 % / -2 % / -2 % -- the entry program (transform + ray
//                       preparation, ~130 instructions) is too long to run for a handful of lanes in every iteration
// LDS: a stack of 10 and a done ring of 80 entries make the kernel's 10 176 B fit 16 times into a CU (+4.6 % against 12 / 96
// = 14 waves, once the exit no longer costs a leaf step; before that the same squeeze measured within noise)
#ifndef PT_INST_EXIT_HEAD
#define PT_INST_EXIT_HEAD 1
#endif
#ifndef PT_INST_ENTRY_HEAD
#define PT_INST_ENTRY_HEAD 0
#endif
constexpr bool kInstExitHead = PT_INST_EXIT_HEAD != 0;
constexpr uint32_t kInstEntryHead = PT_INST_ENTRY_HEAD;
constexpr int kDoneRingInst = PT_INST_RING;  // two-level scenes: their 1.8 KB of saved world rays cost occupancy otherwise
constexpr int kStackLdsInst = PT_INST_STACK;
constexpr bool kInstVote3 = PT_INST_VOTE3 != 0;
constexpr int kDoneRingTiny = PT_DONE_RING_TINY;             // (only when more than 32 lanes retire at once onto more than 32 parked hits)
// ring position p + i for p < kDoneRing, i <= kDoneRing (the ring length need not be a power of two)
template <int kDoneRing>
__device__ __forceinline__ uint32_t pt_ring_at(uint32_t p)
{
	if ((kDoneRing & (kDoneRing - 1)) == 0) return p & (uint32_t)(kDoneRing - 1);
	return p >= (uint32_t)kDoneRing ? p - (uint32_t)kDoneRing : p;
}
constexpr uint32_t kRefillLanesFlat = PT_REFILL_LANES;  // flat scenes (a ray makes one or two leaf steps): 16 .. 40 measure the same
#ifndef PT_REFILL_LANES_TREE
#define PT_REFILL_LANES_TREE 16
#endif
// scenes with a tree to traverse: a ray makes ten and more steps, a refill pass is amortised over more of them, so idle lanes are
// replaced sooner (viking_room +2.5 %, teapot grid +1.7 %, 2 M-triangle mesh +0.3 % against 32)
constexpr uint32_t kRefillLanesTree = PT_REFILL_LANES_TREE;
constexpr uint32_t kNoPath = 0xFFFFFFFFu;
// flat scenes: a leaf the box test selected is skipped when its box begins behind what the ray has hit meanwhile
#ifndef PT_FLAT_CULL
#define PT_FLAT_CULL 1
#endif
constexpr bool kFlatCull = PT_FLAT_CULL != 0;
// the survivors of a wave are a stack, not a queue: the batch staged next is the one shaded last, whose ring records and
// path state were written a moment ago
// (flat scenes: +2.7 %; scenes with a tree lose 1-2 % -- a queue keeps the coherent camera rays of a batch together)
#ifndef PT_RING_LIFO
#define PT_RING_LIFO 1
#endif
constexpr bool kRingLifoOn = PT_RING_LIFO != 0;
// (PT_FLAT_CARRY_STATE: pt_trace.h -- the host sizes the state buffer by it)
#ifndef PT_RING_PREFETCH
#define PT_RING_PREFETCH 1
#endif
constexpr bool kRingPrefetch = PT_RING_PREFETCH != 0;
#ifndef PT_FLAT_LDS_QUADS
#define PT_FLAT_LDS_QUADS 1
#endif

// A kernel argument that only a cold part of k_paths needs (camera-ray generation, the gather fallback of the shading,
// the epilogue), read from the kernarg segment WHERE it is used: the block of arguments is ~60 SGPRs, and what the
// compiler loads at entry it keeps -- or spills to VGPR lanes and restores with v_readlane_b32, a VALU instruction each --
// for the whole kernel.  The empty asm makes the address opaque, so the scalar load cannot be hoisted back to the entry.
// (k_paths has ONE by-value argument, so PtPathsArgs starts at offset 0 of the kernarg segment.)
template <typename T>
__device__ __forceinline__ T pt_cold_arg(uint32_t offset)
{
	static_assert(sizeof(T) % 4 == 0, "whole dwords");
	unsigned long long p = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
	asm volatile("" : "+s"(p));
	const uint32_t __attribute__((address_space(4))) *w =
	    reinterpret_cast<const uint32_t __attribute__((address_space(4))) *>(p + offset);
	union
	{
		T value;
		uint32_t word[sizeof(T) / 4];
	} r;
#pragma unroll
	for (uint32_t i = 0; i < sizeof(T) / 4; ++i) r.word[i] = w[i];
	if constexpr (std::is_pointer<T>::value)
	{
		// a pointer argument points to device memory: say so, or its loads become flat_load instead of global_load
		typedef typename std::remove_pointer<T>::type __attribute__((address_space(1))) *GlobalPtr;
		return (T) reinterpret_cast<GlobalPtr>(reinterpret_cast<unsigned long long>(r.value));
	}
	else return r.value;
}
#define PT_COLD(field) pt_cold_arg<decltype(PtPathsArgs::field)>((uint32_t)offsetof(PtPathsArgs, field))

__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }

// Population count of a wave-uniform mask (a ballot) as a 32-bit scalar.  Without the empty asm the optimiser compares two
// such counts as the 64-bit values __popcll returns, and a 64-bit ordered compare only exists on the VALU (v_cmp_ge_u64).
__device__ __forceinline__ uint32_t popcWave(unsigned long long m)
{
	uint32_t n = (uint32_t)__popcll(m);
	asm("" : "+s"(n));
	return n;
}

// Tuning builds only (-DPT_PATHS_TIMING): wall cycles of every section of the loop, summed per wave into a.trav[6 + section]
// (s_memtime stamps; the stamps themselves cost ~10 % -- shares, not absolute times).
#ifdef PT_PATHS_TIMING
#define PT_T_SECTIONS 16
#define PT_T_MARK(sec)                                               \
	{                                                                \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		tcyc[tsec] += now_ - tlast;                                  \
		tlast = now_;                                                \
		tsec = (sec);                                                \
	}
#else
#define PT_T_MARK(sec)
#endif
enum
{
	PT_T_LOOP = 0,   // loop head, votes, bookkeeping
	PT_T_RETIRE = 1, // parking finished hits
	PT_T_FLUSH = 2,  // shading
	PT_T_RING = 3,   // staging survivors from the ring
	PT_T_GEN = 4,    // claims + camera rays + staging
	PT_T_TAKE = 5,   // idle lanes pick prepared rays up
	PT_T_NODE = 6,
	PT_T_LEAF = 7,
	// parts of the flat staging (the time before the first of them is the wait for the ring's / the generator's data) ...
	PT_T_S_PREPARE = 8,
	PT_T_S_ENTRY = 9,
	PT_T_S_LEAF = 10,
	PT_T_S_STORE = 11,
	// ... and of the shading pass: until its loads have arrived, arithmetic, stores
	PT_T_F_LOADS = 12,
	PT_T_F_MATH = 13,
	PT_T_F_STORE = 14
};
#ifdef PT_PATHS_TIMING
#define PT_T_PIN(x) asm volatile("" ::"v"(x));
#else
#define PT_T_PIN(x)
#endif

// kInst: two-level scene (pt_set_instances).  A lane starts in the top-level tree (inst == PT_NO_INSTANCE), where
// a leaf reference is an instance index: the lane saves its prepared world ray in LDS, transforms the ray with the
// instance's world-to-object matrix (direction NOT renormalised: t means the same in both spaces), pushes
// PT_INSTANCE_END and descends into the bottom-level tree; popping the marker restores the world ray.
// kEntryMode: PT_ENTRY_NONE, or the scene has an entry node (a.entry; pt_wave_traverse.h): the staging phase tests every
// fresh ray against it and hands the lane that takes the ray the set of children it hits; the lane visits them one after
// the other.  PT_ENTRY_FLAT: every child of the entry node is a leaf (a scene of at most PT_ENTRY_WIDTH leaves, the Cornell
// box's 16 quads) -- there is no tree left to traverse, so no stack and no node step; the staging phase also tests the
// FIRST leaf the ray's box test selects, at full lane utilisation, and only a ray with further candidates (one in three
// on Cornell) is handed to a lane at all: the others go straight to the shading ring, or nowhere when they hit nothing.
#ifndef PT_ENTRY_NEAREST
#define PT_ENTRY_NEAREST 1
#endif
constexpr bool kEntryNearest = PT_ENTRY_NEAREST != 0;  // tree form: the lane enters the nearest hit child first
enum
{
	PT_ENTRY_NONE = 0,
	PT_ENTRY_TREE = 1,
	PT_ENTRY_FLAT = 2
};
template <bool kStats, bool kInst, int kNodeMode, int kEntryMode>
__global__ void __launch_bounds__(PT_WAVE) PT_PATHS_OCCUPANCY k_paths(PtPathsArgs a)
{
	constexpr bool kEntry = kEntryMode != PT_ENTRY_NONE, kFlat = kEntryMode == PT_ENTRY_FLAT;
	static_assert(!(kFlat && kInst), "flat: single-level scenes");
	static_assert(!kFlat || kNodeMode == PT_NODES_GLOBAL, "a flat scene has no nodes to copy");
	constexpr bool kTiny = kEntry && kNodeMode != PT_NODES_GLOBAL;
	constexpr int kStackLds = kInst ? kStackLdsInst : (kTiny ? kStackLdsTiny : kStackLdsFull);
	constexpr int kDoneRing = kInst ? kDoneRingInst : (kTiny ? kDoneRingTiny : kDoneRingFull);
	constexpr int kInEntry = kInFields;  // entry children the ray hits (kEntry)
	constexpr int kInDir = kInFields + (kEntry ? 1 : 0);  // first row of the world direction (kInst)
	constexpr int kInBest = kInFields + 1;  // kFlat: best hit so far (t, prim, V, W, det) of a ray that was tested while staged
	__shared__ uint32_t sStack[kFlat ? 1 : kStackLds][PT_WAVE];
	// kCarry (flat scenes): throughput and RNG state travel with the ray.  To stay within 10 240 B of LDS the rows are packed
	// -- a flat scene has at most 16 leaves of at most 8 triangles, so a primitive id is a byte (0xFF: none): it shares the word
	// of the axis bits and the depth (staging row 9: kz | signs | depth << 16 | primitive << 24; done row 3: primitive | depth
	// << 8, done row 5 is free for the state) -- and the staging area holds 48 rays instead of 64 (a batch that hands more than
	// that to the lanes tests another leaf first: stage()).
	constexpr bool kCarry = kFlat && PT_FLAT_CARRY_STATE;
	constexpr int kInCap = kCarry ? 48 : PT_WAVE;                    // rays the staging area holds
	constexpr int kInState = kInBest + 4;                           // kCarry: throughput x, y, z and RNG state (best hit: t V W det)
	constexpr int kDoneState = 5;                                   // kCarry: the same of a parked hit, rows 5 .. 8
	__shared__ uint32_t sIn[kInFields + (kInst ? 3 : 0) + (kEntry ? 1 : 0) + (kFlat ? (kCarry ? 8 : 5) : 0)][kInCap];
	__shared__ uint32_t sDone[kDoneFields + (kInst ? 1 : 0) + (kCarry ? 3 : 0)][kDoneRing];
	__shared__ uint4 sEntry[kEntry ? PT_ENTRY_REC : 1];
	if (kEntry)
	{
		if (threadIdx.x < PT_ENTRY_REC) sEntry[threadIdx.x] = a.entry[threadIdx.x];
		__syncthreads();
	}
	// flat scenes: their (at most 16) quads, one copy in x, y, z order (pt_leaf_test picks the ray's axis order by address): a
	// leaf test waits for LDS instead of for a divergent load from memory
	constexpr bool kLdsQuads = kFlat && PT_FLAT_LDS_QUADS;
	__shared__ float sQuads[kLdsQuads ? PT_LDS_QUAD_WORDS : 1];
	if (kLdsQuads)
	{
		const uint32_t nq = a.sc.nQuads <= PT_LDS_QUADS ? a.sc.nQuads : 0u;  // (a flat scene has at most 16 leaves)
		const uint32_t *src = reinterpret_cast<const uint32_t *>(a.sc.quads + 2 * (size_t)a.sc.nQuads);  // plane 2: x, y, z
		uint32_t *dst = reinterpret_cast<uint32_t *>(sQuads);
		for (uint32_t i = threadIdx.x; i < nq * 14u; i += PT_WAVE)
		{
			const uint32_t q = i / 14u, w = i % 14u;
			dst[w < 12u ? q * 12u + w : PT_LDS_QUADS * 12u + 2u * q + (w - 12u)] = src[q * 16u + w];
		}
		__syncthreads();
	}
	__shared__ uint32_t sSave[kInst ? kSaveFields : 1][PT_WAVE];
	// tiny scenes (e.g. the Cornell box's 7 wide nodes): a wave-private copy of the node array, decoded (PT_NODES_PLANES) or
	// as it is (PT_NODES_PACKED); see pt_wave_traverse.h
	constexpr uint32_t kNodeWords = kNodeMode == PT_NODES_PLANES ? PT_LDS_NODES_PLANES * PT_PLANE_REC * 4u
	                                                             : (kNodeMode == PT_NODES_PACKED ? PT_LDS_NODES_PACKED * 16u : 4u);
	__shared__ uint4 sNodeMem[kNodeWords / 4u];
	const uint4 *const sNodes = kNodeMode != PT_NODES_GLOBAL ? sNodeMem : nullptr;
	if (kNodeMode == PT_NODES_PLANES)
	{
		uint32_t *dst = reinterpret_cast<uint32_t *>(sNodeMem);
		const uint32_t nRec = a.sc.nNodes < PT_LDS_NODES_PLANES ? a.sc.nNodes : PT_LDS_NODES_PLANES;
		for (uint32_t i = threadIdx.x; i < nRec * (4u * PT_PLANE_REC); i += PT_WAVE)
			dst[i] = pt_plane_record_word(a.sc.nodes, i / (4u * PT_PLANE_REC), i % (4u * PT_PLANE_REC));
		__syncthreads();
	}
	else if (kNodeMode == PT_NODES_PACKED)
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(a.sc.nodes);
		for (uint32_t i = threadIdx.x; i < a.sc.nNodes * 4u && i < PT_LDS_NODES_PACKED * 4u; i += PT_WAVE) sNodeMem[i] = src[i];
		__syncthreads();
	}

	const PtSceneDev &sc = a.sc;
	const int lane = threadIdx.x;
	const unsigned long long laneBit = 1ull << lane, lanesBelow = laneBit - 1ull;
	uint32_t *mySpill = a.spill + (size_t)blockIdx.x * PT_WAVE + lane;
	const float tmin = a.tmin, tmax = a.tmax;
	const uint32_t WH = a.tile.w * a.tile.h;  // pixels of the tile this dispatch covers
	const uint32_t itemsPerFrame = ((WH + PT_WAVE - 1) / PT_WAVE) * a.spp;
	const uint32_t totalItems = itemsPerFrame * a.nFrames;

	// ---- generator (wave-uniform cursors + the per-lane raygen seed chain) ----
	const uint32_t firstShare = totalItems / (2u * gridDim.x);
	const uint32_t firstK = firstShare >= PT_PATHS_MAX_ITEMS ? PT_PATHS_MAX_ITEMS : (firstShare == 0 ? 1u : firstShare);
	const uint32_t headBias = gridDim.x * firstK;  // items handed out statically (first claim of every wave)
	bool firstClaim = true, genDone = (totalItems == 0), needSeed = true;
	uint32_t lastSeen = headBias, itemCur = 0, itemEnd = 0;
	uint32_t genSeed = 0, genPx = 0, genPy = 0;
	bool genValid = false;

	// ---- wave-private survivor ring in global memory ----
	float4 *ringO = a.ringO + (size_t)blockIdx.x * PT_PATHS_RING;
	float4 *ringD = a.ringD + (size_t)blockIdx.x * PT_PATHS_RING;
	float4 *ringT = a.state + (size_t)blockIdx.x * PT_PATHS_RING;  // kCarry: (throughput, RNG state) of the survivors; the table is not used
	uint32_t ringHead = 0, ringCount = 0;
	float cTx = 1.0f, cTy = 1.0f, cTz = 1.0f;  // kCarry: throughput and RNG state of the lane's ray
	uint32_t cSeed = 0;
	// the next full batch of the ring, requested as soon as the ring holds it (end of a shading pass, or after a batch was
	// staged) and used when the staging area next runs empty: the load's latency passes while the wave traverses
	constexpr bool kRingLifo = kRingLifoOn && kFlat;
	constexpr bool kPrefetch = kRingPrefetch && kFlat && !(kFlat && PT_FLAT_CARRY_STATE);  // (carried state: no registers left for it)  // (scenes with a tree: no difference measured, eight registers more)
	float4 pfO = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pfD = pfO, pfT = pfO;
	bool pfValid = false;
	auto prefetchRing = [&]() {
		if (kPrefetch && !pfValid && ringCount >= PT_WAVE)
		{
			const uint32_t slot = kRingLifo ? ringCount - PT_WAVE + lane : (ringHead + lane) & (PT_PATHS_RING - 1);
			pfO = ringO[slot];
			pfD = ringD[slot];
			if (kCarry) pfT = ringT[slot];
			pfValid = true;
			if (kRingLifo) ringCount -= PT_WAVE;  // taken off the stack now: a later shading pass pushes on top of what is left
		}
	};

	uint32_t inCount = 0, inPos = 0;      // prepared rays in sIn
	uint32_t doneHead = 0, doneCount = 0;  // LDS ring of finished rays: first entry (< kDoneRing) and entries parked
	uint32_t raysFetched = 0;
	uint32_t nodeFetches = 0, triTests = 0, nHits = 0, leafVisits = 0, nodeSteps = 0, leafSteps = 0;

#ifdef PT_PATHS_TIMING
	unsigned long long tcyc[PT_T_SECTIONS] = {}, tlast = __builtin_amdgcn_s_memtime();
	int tsec = PT_T_LOOP;
#endif
	// per-lane ray state
	float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0, Sx = 0, Sy = 0, Sz = 0;
	uint32_t kpack = 0, q = kNoPath;
	float wox = 0, woy = 0, woz = 0, wdx = 0, wdy = 0, wdz = 0;  // world ray (two-level scenes)
	uint32_t inst = PT_NO_INSTANCE, bestInst = 0;
	uint32_t pend = 0;  // entry-node children the lane's ray hits and has not visited yet (kEntry)
	PtBest best = { tmax, PT_NO_HIT, 0.0f, 0.0f, 1.0f };
	uint32_t cur = PT_TRAVERSAL_DONE;
	int sp = 0;

	// Shades `count` (<= 64) parked hits and sends the survivors to the wave's ring.
	auto flush = [&](uint32_t count) {
		bool alive = false;
		// (what follows is read by survivors only, and every survivor has written it: left uninitialised on purpose -- zeros
		// for the other lanes are a dozen moves per pass)
		uint32_t litOut;
		float4 nT;
		float3x no, nd;
		uint32_t pq, depth;
		const int maxDepth = PT_COLD(maxDepth);
		// what the shading reads: the per-primitive records, or (no records / a diffuse map) the reference's gather
		PtSceneDev fsc = {};
		fsc.shade = sc.shade;
		fsc.texture = PT_COLD(sc.texture);
		if (!fsc.shade || fsc.texture)  // wave-uniform
		{
			fsc.vertices = PT_COLD(sc.vertices);
			fsc.indices = PT_COLD(sc.indices);
			fsc.materials = PT_COLD(sc.materials);
			fsc.matIndices = PT_COLD(sc.matIndices);
			fsc.texW = PT_COLD(sc.texW);
			fsc.texH = PT_COLD(sc.texH);
			fsc.texSlots = PT_COLD(sc.texSlots);
			fsc.nTexSlots = PT_COLD(sc.nTexSlots);
		}
		PT_T_MARK(PT_T_F_LOADS)
		if ((uint32_t)lane < count)
		{
			uint32_t e = pt_ring_at<kDoneRing>(doneHead + (uint32_t)lane);
			uint32_t prim = sDone[3][e];
			pq = sDone[4][e];
			if (kCarry)
			{
				depth = prim >> 8;
				prim &= 0xFFu;
			}
			else depth = sDone[5][e] >> 16;
			// bit 7 of the depth byte: the path has written its radiance record before (possible only where an emission does
			// not end the path, i.e. its components do not sum to more than zero)
			uint32_t lit = (depth >> 7) & 1u;
			depth &= 0x7Fu;
			if (prim != PT_NO_HIT)  // rmiss: hitValue = 0, emission = 0, terminate -> nothing to do
			{
				float V = __uint_as_float(sDone[0][e]), W = __uint_as_float(sDone[1][e]), det = __uint_as_float(sDone[2][e]);
				float u, v;
				if (pt_all_in_division_range(det, V, W))  // same quotients, the reciprocal refined once (pt_device_common.h)
				{
#ifdef PT_PROBE_RCPDET
					const float rcpDet = PtSharedDivisor(det).quotient(1.0f);  // (speed probe only: changes results)
					u = V * rcpDet;
					v = W * rcpDet;
#else
					const PtSharedDivisor by(det);
					u = by.quotient(V);
					v = by.quotient(W);
#endif
				}
				else
				{
					u = V / det;
					v = W / det;
				}
				float4 T;  // throughput, rng state
				if (kCarry)
					T = make_float4(__uint_as_float(sDone[kDoneState + 0][e]), __uint_as_float(sDone[kDoneState + 1][e]),
					                __uint_as_float(sDone[kDoneState + 2][e]), __uint_as_float(sDone[kDoneState + 3][e]));
				else T = a.state[pq];
				uint32_t seed = __float_as_uint(T.w);
				float3x kd, ke;
#ifdef PT_PATHS_TIMING
				{
					// (tuning build: wait for the state and for the first and last word of the shading record)
					const float4 *r = reinterpret_cast<const float4 *>(fsc.shade + prim);
					float4 r0 = r[0], r5 = r[5];
					PT_T_PIN(T.w) PT_T_PIN(r0.x) PT_T_PIN(r5.w) PT_T_PIN(u)
					PT_T_MARK(PT_T_F_MATH)
				}
#endif
				pt_shade_hit<true>(fsc, prim, u, v, seed, no, nd, kd, ke,
				             kInst ? a.instances + 24 * (size_t)sDone[kDoneFields][e] : nullptr);
				PT_T_PIN(nd.x) PT_T_PIN(nd.z)
				float ex = ke.x, ey = ke.y, ez = ke.z;
				if (ex != 0.0f || ey != 0.0f || ez != 0.0f)
				{
					// accumDepthHitValue += emission * curAttenuation (rgen:78); one thread owns a path
					// the record is not zeroed when the path starts: the first addition is a store of 0 + e * T and sets the
					// path's bit in the launch's mask, which is what k_accumulate_items looks at before it reads a record
					float4 L = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					if (lit) L = a.rad[pq];
					else __hip_atomic_fetch_or(PT_COLD(lit) + (pq >> 6), 1ull << (pq & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					L.x += ex * T.x;
					L.y += ey * T.y;
					L.z += ez * T.z;
					a.rad[pq] = L;
					lit = 1u;
				}
				alive = (int)(depth + 1) < maxDepth && !((ex + ey) + ez > 0.0f);  // rgen:61 loop bound, rchit:56-58
				if (alive)  // curAttenuation *= hitValue (rgen:79)
				{
					nT = make_float4(T.x * kd.x, T.y * kd.y, T.z * kd.z, __uint_as_float(seed));
					if (!kCarry) a.state[pq] = nT;
				}
				litOut = lit;
				if (kStats) ++nHits;
			}
		}
		PT_T_MARK(PT_T_F_STORE)
		doneHead = pt_ring_at<kDoneRing>(doneHead + count);
		doneCount -= count;
		unsigned long long am = __ballot(alive);
		if (alive)
		{
			uint32_t slot = kRingLifo ? ringCount + popc64(am & lanesBelow)
			                          : (ringHead + ringCount + popc64(am & lanesBelow)) & (PT_PATHS_RING - 1);
			ringO[slot] = make_float4(no.x, no.y, no.z, __uint_as_float((depth + 1) | (litOut << 7)));
			ringD[slot] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(pq));
			if (kCarry) ringT[slot] = nT;
		}
		ringCount += popcWave(am);
		prefetchRing();
	};

	// Prepares the rays of a staging batch (lanes with `valid`; every lane of the wave calls) and writes them into the
	// staging area; returns how many rays wait there for a lane.
	auto stage = [&](bool valid, float3x o, float3x d, uint32_t depth, uint32_t pq, float4 st) -> uint32_t {
		if (!kFlat)
		{
			// With an entry node a ray whose box test selects no child has missed the scene: it is finished here (rmiss adds
			// nothing) and never takes a lane; the others are ballot-packed.  Without one every valid ray is kept (the valid
			// lanes are 0 .. n-1, so the rank is the lane).
			PtLaneRay pr = {};
			uint32_t m = 0;
			if (valid)
			{
				pr = pt_prepare_ray<true>(o, d);  // every lane of the staging batch is here: the shared-reciprocal form
				if (kEntry)
				{
					m = pt_entry_test<kEntryNearest>(sEntry, pr, tmin, tmax);
					if (kStats) ++nodeFetches;  // one (wide) node visit
				}
			}
			const bool keep = valid && (!kEntry || (m & 0xFFFFu) != 0u);
			const unsigned long long km = __ballot(keep);
			if (keep)
			{
				const uint32_t r = kEntry ? popc64(km & lanesBelow) : (uint32_t)lane;
				sIn[0][r] = __float_as_uint(pr.ox);
				sIn[1][r] = __float_as_uint(pr.oy);
				sIn[2][r] = __float_as_uint(pr.oz);
				sIn[3][r] = __float_as_uint(pr.ix);
				sIn[4][r] = __float_as_uint(pr.iy);
				sIn[5][r] = __float_as_uint(pr.iz);
				sIn[6][r] = __float_as_uint(pr.Sx);
				sIn[7][r] = __float_as_uint(pr.Sy);
				sIn[8][r] = __float_as_uint(pr.Sz);
				sIn[9][r] = pr.kpack | (depth << 16);
				sIn[10][r] = pq;
				if (kEntry) sIn[kInEntry][r] = m;
				if (kInst)
				{
					sIn[kInDir + 0][r] = __float_as_uint(d.x);
					sIn[kInDir + 1][r] = __float_as_uint(d.y);
					sIn[kInDir + 2][r] = __float_as_uint(d.z);
				}
			}
			return popcWave(km);
		}
		// flat scene: box test and the first leaf here, with all 64 lanes; then the ray is finished (hit: parked for the
		// shading pass; nothing hit: gone) or goes -- ballot-packed -- to the staging area with what it has found so far
		// (pr, b, m: read behind `valid &&` only -- no zeros for the lanes of a partial batch)
		PtLaneRay pr;
		PtBest b;
		uint32_t m;
		PT_T_PIN(o.x) PT_T_PIN(d.x) PT_T_MARK(PT_T_S_PREPARE)
		if (valid)
		{
			b = { tmax, PT_NO_HIT, 0.0f, 0.0f, 1.0f };
			pr = pt_prepare_ray<true>(o, d);
#ifdef PT_PROBE_VALU
			{
				// tuning probe: PT_PROBE_VALU independent vector instructions that do nothing
				float x0 = pr.ox, x1 = pr.oy, x2 = pr.oz, x3 = pr.ix;
#pragma unroll
				for (int i = 0; i < PT_PROBE_VALU / 4; ++i)
					asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
					             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
			}
#endif
			PT_T_PIN(pr.Sx) PT_T_PIN(pr.ix) PT_T_MARK(PT_T_S_ENTRY)
			m = pt_entry_test(sEntry, pr, tmin, tmax);
			PT_T_PIN(m) PT_T_MARK(PT_T_S_LEAF)
			if (kStats) ++nodeFetches;
			if (m)
			{
				const uint32_t ref = pt_entry_next(sEntry, m);
				if (kStats) ++leafVisits;
				pt_leaf_test<true, kLdsQuads>(sc, pr, tmin, b, ref, triTests, 0u, nullptr, sQuads);
			}
			PT_T_PIN(b.t) PT_T_PIN(b.prim)
		}
		PT_T_MARK(PT_T_S_STORE)
		if (kInCap < PT_WAVE)
		{
			// the staging area is shorter than a batch: while more rays than it holds have candidates left, every one of them
			// tests its next leaf here (skipping those behind its hit, as a lane would)
			while (popcWave(__ballot(valid && m != 0u)) > (uint32_t)kInCap)
			{
				uint32_t ref = PT_TRAVERSAL_DONE;
				if (valid)
				{
					const float limit = b.t * 1.000001f;
					while (m)
					{
						const uint32_t bit = (uint32_t)__builtin_ctz(m);
						m &= m - 1u;
						const uint32_t child = (uint32_t)(PT_ENTRY_WIDTH - 1) - bit;
						if (!kFlatCull || b.prim == PT_NO_HIT || pt_entry_child_t0(sEntry, child, pr, tmin) <= limit)
						{
							ref = reinterpret_cast<const uint32_t *>(sEntry + 2)[child];
							break;
						}
					}
				}
				if (ref != PT_TRAVERSAL_DONE)
				{
					if (kStats) ++leafVisits;
					pt_leaf_test<true, kLdsQuads>(sc, pr, tmin, b, ref, triTests, 0u, nullptr, sQuads);
				}
			}
		}
		const bool more = valid && m != 0u, parked = valid && m == 0u && b.prim != PT_NO_HIT;
		const unsigned long long mm = __ballot(more), hm = __ballot(parked);
		if (parked)
		{
			const uint32_t e = pt_ring_at<kDoneRing>(pt_ring_at<kDoneRing>(doneHead + doneCount) + popc64(hm & lanesBelow));
			sDone[0][e] = __float_as_uint(b.V);
			sDone[1][e] = __float_as_uint(b.W);
			sDone[2][e] = __float_as_uint(b.det);
			sDone[3][e] = kCarry ? (b.prim | (depth << 8)) : b.prim;
			sDone[4][e] = pq;
			if (!kCarry) sDone[5][e] = pr.kpack | (depth << 16);
			if (kCarry)
			{
				sDone[kDoneState + 0][e] = __float_as_uint(st.x);
				sDone[kDoneState + 1][e] = __float_as_uint(st.y);
				sDone[kDoneState + 2][e] = __float_as_uint(st.z);
				sDone[kDoneState + 3][e] = __float_as_uint(st.w);
			}
		}
		doneCount += popcWave(hm);  // <= 63 before (the caller shades as soon as 64 are parked), so <= 127 now
		if (more)
		{
			const uint32_t r = popc64(mm & lanesBelow);
			sIn[0][r] = __float_as_uint(pr.ox);
			sIn[1][r] = __float_as_uint(pr.oy);
			sIn[2][r] = __float_as_uint(pr.oz);
			if (kFlatCull)
			{
				sIn[3][r] = __float_as_uint(pr.ix);
				sIn[4][r] = __float_as_uint(pr.iy);
				sIn[5][r] = __float_as_uint(pr.iz);
			}
			sIn[6][r] = __float_as_uint(pr.Sx);
			sIn[7][r] = __float_as_uint(pr.Sy);
			sIn[8][r] = __float_as_uint(pr.Sz);
			sIn[9][r] = pr.kpack | (depth << 16) | (kCarry ? (b.prim == PT_NO_HIT ? 0xFFu : b.prim) << 24 : 0u);
			sIn[10][r] = pq;
			sIn[kInEntry][r] = m;
			sIn[kInBest + 0][r] = __float_as_uint(b.t);
			if (!kCarry) sIn[kInBest + 1][r] = b.prim;
			sIn[kInBest + (kCarry ? 1 : 2)][r] = __float_as_uint(b.V);
			sIn[kInBest + (kCarry ? 2 : 3)][r] = __float_as_uint(b.W);
			sIn[kInBest + (kCarry ? 3 : 4)][r] = __float_as_uint(b.det);
			if (kCarry)
			{
				sIn[kInState + 0][r] = __float_as_uint(st.x);
				sIn[kInState + 1][r] = __float_as_uint(st.y);
				sIn[kInState + 2][r] = __float_as_uint(st.z);
				sIn[kInState + 3][r] = __float_as_uint(st.w);
			}
		}
		return popcWave(mm);
	};

	// two-level scenes: the bottom-level traversal of the lane's instance is finished -- back to the world ray and on with
	// whatever the top-level traversal left on the stack
	auto leaveInstance = [&]() {
		if constexpr (!kInst) return;  // (sSave has one row in the single-level instantiations)
		else
		{
			ox = wox;
			oy = woy;
			oz = woz;
			ix = __uint_as_float(sSave[0][lane]);
			iy = __uint_as_float(sSave[1][lane]);
			iz = __uint_as_float(sSave[2][lane]);
			Sx = __uint_as_float(sSave[3][lane]);
			Sy = __uint_as_float(sSave[4][lane]);
			Sz = __uint_as_float(sSave[5][lane]);
			kpack = sSave[6][lane];
			inst = PT_NO_INSTANCE;
			cur = pt_stack_pop<kStackLds>(sp, sStack, lane, mySpill, a.spillStride);
		}
	};

	// top-level leaf = instance: ray into object space, o' = W2O * (o,1), d' = W2O * (d,0), each row
	// ((m0*x + m1*y) + m2*z) + m3 -- then the same preparation as in the staging phase
	struct InstLoads
	{
		float4 r0, r1, r2;
		uint32_t root;
	};
	auto loadInstance = [&](InstLoads &L) {
		const uint32_t i = cur & 0x0FFFFFFFu;
		const float4 *w = reinterpret_cast<const float4 *>(a.instances + 24 * (size_t)i + 12);
		L.r0 = w[0];
		L.r1 = w[1];
		L.r2 = w[2];
		L.root = a.instRoot[i];  // root of the bottom-level tree of the instance's Model
	};
	auto enterInstance = [&](const InstLoads &L) {
		if constexpr (!kInst) return;
		else
		{
			inst = cur & 0x0FFFFFFFu;
			const float4 r0 = L.r0, r1 = L.r1, r2 = L.r2;
			float3x oo, od;
			oo.x = ((r0.x * wox + r0.y * woy) + r0.z * woz) + r0.w;
			oo.y = ((r1.x * wox + r1.y * woy) + r1.z * woz) + r1.w;
			oo.z = ((r2.x * wox + r2.y * woy) + r2.z * woz) + r2.w;
			od.x = (r0.x * wdx + r0.y * wdy) + r0.z * wdz;
			od.y = (r1.x * wdx + r1.y * wdy) + r1.z * wdz;
			od.z = (r2.x * wdx + r2.y * wdy) + r2.z * wdz;
			sSave[0][lane] = __float_as_uint(ix);
			sSave[1][lane] = __float_as_uint(iy);
			sSave[2][lane] = __float_as_uint(iz);
			sSave[3][lane] = __float_as_uint(Sx);
			sSave[4][lane] = __float_as_uint(Sy);
			sSave[5][lane] = __float_as_uint(Sz);
			sSave[6][lane] = kpack;
			const PtLaneRay pr = pt_prepare_ray(oo, od);
			ox = pr.ox;
			oy = pr.oy;
			oz = pr.oz;
			ix = pr.ix;
			iy = pr.iy;
			iz = pr.iz;
			Sx = pr.Sx;
			Sy = pr.Sy;
			Sz = pr.Sz;
			kpack = pr.kpack | (kpack & 0xFFFF0000u);  // keep the depth bits
			pt_stack_push<kStackLds>(PT_INSTANCE_END, sp, sStack, lane, mySpill, a.spillStride);
			cur = L.root;
			if (kStats) ++leafVisits;
		}
	};
	static_assert(!kInstEntryHead || kInstExitHead, "entry at the head of the iteration: the exit too");

	for (;;)
	{
		PT_T_MARK(PT_T_LOOP)
		if (kInst && kInstExitHead && __ballot(cur == PT_INSTANCE_END))
		{
			if (cur == PT_INSTANCE_END) leaveInstance();
		}
		if (kEntry && !kFlat && cur == PT_TRAVERSAL_DONE && pend != 0u)
		{
			// the subtree of one entry child is finished (the sentinel was popped): on to the next child the ray hits
			cur = pt_entry_next(sEntry, pend);
			sp = 1;
		}
		// Retire + refill only when enough lanes wait for it (or nothing else is left to do): the section
		// costs about half a node step, so it is not run for every single finished lane.
		const uint32_t nIdle = popcWave(__ballot(cur == PT_TRAVERSAL_DONE));
		if (nIdle >= (kFlat ? kRefillLanesFlat : kRefillLanesTree))
		{
			PT_T_MARK(PT_T_RETIRE)
			// ------------------------------------------------------------------ C: retire finished lanes ----
			// (a ray that hit nothing is finished for good -- rmiss adds nothing and ends the path -- and is not parked)
			const bool ended = (q != kNoPath) && (cur == PT_TRAVERSAL_DONE);
			if (ended && best.prim == PT_NO_HIT) q = kNoPath;
			bool fin = ended && best.prim != PT_NO_HIT;
			unsigned long long fm = __ballot(fin);
			if (fm)
			{
				if (kDoneRing < 2 * PT_WAVE - 1 && doneCount + popcWave(fm) > (uint32_t)kDoneRing)
				{
					// a ring shorter than 127 entries cannot always take the finished lanes: shade what is parked first
					PT_T_MARK(PT_T_FLUSH)
					__syncthreads();
					flush(doneCount);
					__syncthreads();
					PT_T_MARK(PT_T_RETIRE)
				}
				if (fin)
				{
					uint32_t e = pt_ring_at<kDoneRing>(pt_ring_at<kDoneRing>(doneHead + doneCount) + popc64(fm & lanesBelow));
					sDone[0][e] = __float_as_uint(best.V);
					sDone[1][e] = __float_as_uint(best.W);
					sDone[2][e] = __float_as_uint(best.det);
					sDone[3][e] = kCarry ? (best.prim | ((kpack >> 16) << 8)) : best.prim;
					sDone[4][e] = q;
					if (!kCarry) sDone[5][e] = kpack;
					if (kInst) sDone[kDoneFields][e] = bestInst;
					if (kCarry)
					{
						sDone[kDoneState + 0][e] = __float_as_uint(cTx);
						sDone[kDoneState + 1][e] = __float_as_uint(cTy);
						sDone[kDoneState + 2][e] = __float_as_uint(cTz);
						sDone[kDoneState + 3][e] = cSeed;
					}
					q = kNoPath;
				}
				doneCount += popcWave(fm);
				if (doneCount >= PT_WAVE)
				{
					PT_T_MARK(PT_T_FLUSH)
					__syncthreads();
					flush(PT_WAVE);
					__syncthreads();
				}
			}
			PT_T_MARK(PT_T_TAKE)
			// ------------------------------------------------------------------ A: refill idle lanes ----
			unsigned long long idle = __ballot(cur == PT_TRAVERSAL_DONE);
			while (idle)
			{
				if (inPos == inCount)
				{
					uint32_t n = 0, staged = 0;
					if ((kRingLifo && pfValid) || ringCount >= PT_WAVE || (genDone && ringCount > 0))
					{
						// continue paths of this wave: the next (up to) 64 survivors
						PT_T_MARK(PT_T_RING)
						const bool fromRegs = kPrefetch && pfValid;
						n = fromRegs ? PT_WAVE : (ringCount < PT_WAVE ? ringCount : PT_WAVE);
						__syncthreads();  // staging contents consumed; ring stores of earlier flushes complete
						float4 ro = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rd = ro, rt = ro;
						if (fromRegs)
						{
							ro = pfO;
							rd = pfD;
							rt = pfT;
							pfValid = false;
						}
						else if ((uint32_t)lane < n)
						{
							uint32_t slot = kRingLifo ? ringCount - n + lane : (ringHead + lane) & (PT_PATHS_RING - 1);
							ro = ringO[slot];
							rd = ringD[slot];
							if (kCarry) rt = ringT[slot];
						}
						staged = stage((uint32_t)lane < n, { ro.x, ro.y, ro.z }, { rd.x, rd.y, rd.z }, __float_as_uint(ro.w),
						               __float_as_uint(rd.w), rt);
						if (!(kRingLifo && fromRegs))
						{
							ringHead += n;
							ringCount -= n;
						}
						prefetchRing();
					}
					else if (!genDone)
					{
						PT_T_MARK(PT_T_GEN)
						if (itemCur == itemEnd)
						{
							// Guided self-scheduling over the work items: a static first claim, then one atomic for
							// clamp(remaining / (2 * waves), MIN, MAX) items; the wave never polls the counter, it uses
							// the value its own previous claim returned.
							uint32_t c, k;
							if (firstClaim)
							{
								firstClaim = false;
								k = firstK;
								c = blockIdx.x * firstK;
							}
							else
							{
								if (lastSeen >= totalItems)
								{
									genDone = true;
									continue;
								}
								const uint32_t share = (totalItems - lastSeen) / (2u * gridDim.x);
								k = share >= PT_PATHS_MAX_ITEMS ? PT_PATHS_MAX_ITEMS
								                                : (share <= PT_PATHS_MIN_ITEMS ? PT_PATHS_MIN_ITEMS : share);
								c = 0;
								if (lane == 0) c = atomicAdd(PT_COLD(itemHead), k);
								c = __builtin_amdgcn_readfirstlane(c) + headBias;
							}
							if (c >= totalItems)
							{
								genDone = true;
								continue;
							}
							lastSeen = c + k;
							itemCur = c;
							itemEnd = c + k < totalItems ? c + k : totalItems;
							needSeed = true;
						}
						const uint32_t item = itemCur++;
						const uint32_t f = item / itemsPerFrame, local = item - f * itemsPerFrame;  // frame of the launch
						const uint32_t spp = PT_COLD(spp);
						const uint32_t g = local / spp, i = local - g * spp;
						if (needSeed || i == 0)
						{
							needSeed = false;
							uint32_t p = g * PT_WAVE + lane;
							genValid = p < WH;
							const PtTile tile = PT_COLD(tile);
							const uint32_t ty = p / tile.w;
							genPy = tile.y0 + ty;
							genPx = pt_tile_column(tile, p - ty * tile.w);
							genSeed = pt_tea(genPy * PT_COLD(W) + genPx, PT_COLD(seedWord) + f * PT_COLD(frameStride));  // rgen:28, val1 = seed + global frame
							if (i)
							{
								uint2 ac = PT_COLD(lcgSkip)[i];  // state after 2*i lcg steps = A * s + C
								genSeed = ac.x * genSeed + ac.y;
							}
						}
						n = popcWave(__ballot(genValid));  // valid lanes are 0 .. n-1 (only the last group is partial)
						__syncthreads();
						float3x go = { 0, 0, 0 }, gd = { 0, 0, 1 };
						float4 gst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
						uint32_t gq = 0;
						if (genValid)
						{
							float r1 = pt_rnd(genSeed);
							float r2 = pt_rnd(genSeed);
							const bool centre = f == PT_COLD(centreFrame);  // rgen:38: frame 0 samples the pixel centre
							float jx = centre ? 0.5f : r1, jy = centre ? 0.5f : r2;
							float3x o, d;
							// the matrices are fetched here, by scalar loads; the empty asm keeps the loads from being hoisted
							// out of the loop (where they would occupy 32 SGPRs for good)
							unsigned long long camAddr = reinterpret_cast<unsigned long long>(PT_COLD(cam));
							asm volatile("" : "+s"(camAddr));
							pt_const_float *cm = reinterpret_cast<pt_const_float *>(camAddr);
							pt_camera_ray_const<pt_const_float *>(cm, PT_COLD(W), PT_COLD(H), genPx, genPy, jx, jy, o, d);
							uint32_t pq = item * PT_WAVE + lane;
							// payload.seed = seed: the bounce stream starts from the state after the two jitter
							// draws, while the raygen-local seed continues from the same state for the next sample
							if (!kCarry) a.state[pq] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(genSeed));
							gst = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(genSeed));
							if (lane == 0) PT_COLD(lit)[item] = 0ull;  // (instead of zeroing the item's 64 radiance records)
							go = o;
							gd = d;
							gq = pq;
						}
						staged = stage(genValid, go, gd, 0u, gq, gst);
						if (n == 0) continue;
					}
					else
					{
						// nothing to stage: in the drain, shade a partial batch early so its survivors keep lanes busy
						if (doneCount && popcWave(idle) >= 32u)
						{
							__syncthreads();
							flush(doneCount);
							__syncthreads();
							if (ringCount || pfValid) continue;
						}
						break;
					}
					__syncthreads();
					inCount = staged;
					inPos = 0;
					raysFetched += n;
					if (kFlat)
					{
						// rays that were finished while staged are parked already: shade as soon as a full batch waits
						if (doneCount >= PT_WAVE)
						{
							PT_T_MARK(PT_T_FLUSH)
							flush(PT_WAVE);
							__syncthreads();
						}
					}
					if (kEntry && staged == 0) continue;  // none of the batch needs a lane: stage the next one
					PT_T_MARK(PT_T_TAKE)
				}
				uint32_t avail = inCount - inPos, want = popcWave(idle);
				uint32_t n = want < avail ? want : avail;
				uint32_t rank = popc64(idle & lanesBelow);
				bool take = (idle & laneBit) && rank < n;
				if (take)
				{
					uint32_t s = inPos + rank;
					ox = __uint_as_float(sIn[0][s]);
					oy = __uint_as_float(sIn[1][s]);
					oz = __uint_as_float(sIn[2][s]);
					ix = __uint_as_float(sIn[3][s]);
					iy = __uint_as_float(sIn[4][s]);
					iz = __uint_as_float(sIn[5][s]);
					Sx = __uint_as_float(sIn[6][s]);
					Sy = __uint_as_float(sIn[7][s]);
					Sz = __uint_as_float(sIn[8][s]);
					kpack = sIn[9][s];
					q = sIn[10][s];
					if (kInst)
					{
						wox = ox;
						woy = oy;
						woz = oz;
						wdx = __uint_as_float(sIn[kInDir + 0][s]);
						wdy = __uint_as_float(sIn[kInDir + 1][s]);
						wdz = __uint_as_float(sIn[kInDir + 2][s]);
						inst = PT_NO_INSTANCE;
						bestInst = 0;
					}
					best.t = tmax;
					best.prim = PT_NO_HIT;
					best.V = 0.0f;
					best.W = 0.0f;
					best.det = 1.0f;
					sStack[0][lane] = PT_TRAVERSAL_DONE;  // sentinel: popping an empty stack ends the traversal
					sp = 1;
					cur = sc.nNodes ? (kInst ? a.tlasRoot : 0u) : PT_TRAVERSAL_DONE;
					if (kEntry)
					{
						// the children of the entry node the ray hits (tested while it was staged): the first one now, the
						// others when the traversal below it has emptied the stack (loop head); none: the ray misses the scene
						const uint32_t m = sIn[kInEntry][s];
						pend = m & 0xFFFFu;
						if (kEntryNearest && !kFlat) cur = pend ? pt_entry_first(sEntry, m, pend) : PT_TRAVERSAL_DONE;
						else cur = pend ? pt_entry_next(sEntry, pend) : PT_TRAVERSAL_DONE;
					}
					if (kCarry)
					{
						cTx = __uint_as_float(sIn[kInState + 0][s]);
						cTy = __uint_as_float(sIn[kInState + 1][s]);
						cTz = __uint_as_float(sIn[kInState + 2][s]);
						cSeed = sIn[kInState + 3][s];
					}
					if (kFlat)
					{
						best.t = __uint_as_float(sIn[kInBest + 0][s]);
						if (kCarry)
						{
							best.prim = (kpack >> 24) == 0xFFu ? PT_NO_HIT : kpack >> 24;
							kpack &= 0x00FFFFFFu;
						}
						else best.prim = sIn[kInBest + 1][s];
						best.V = __uint_as_float(sIn[kInBest + (kCarry ? 1 : 2)][s]);
						best.W = __uint_as_float(sIn[kInBest + (kCarry ? 2 : 3)][s]);
						best.det = __uint_as_float(sIn[kInBest + (kCarry ? 3 : 4)][s]);
					}
				}
				inPos += n;
				idle &= ~__ballot(take);
			}
			if (!__ballot(q != kNoPath))
			{
				// no lane holds a path and nothing could be staged: shade what is parked (its survivors come
				// back through the ring), or leave when that is empty too
				if (doneCount)
				{
					__syncthreads();
					flush(doneCount);
					__syncthreads();
					continue;
				}
				if (ringCount == 0 && !pfValid && inPos == inCount && genDone) break;
				continue;
			}
		}

		// ------------------------------------------------------------------ B: traverse ----
		if (kFlat)
		{
			// every lane that holds a ray is at a leaf: test it, then on to the next leaf the ray's box test selected
			PT_T_MARK(PT_T_LEAF)
			const PtLaneRay fray = { ox, oy, oz, ix, iy, iz, Sx, Sy, Sz, kpack };
			if (kStats) ++leafSteps;
			if (cur != PT_TRAVERSAL_DONE)
			{
				if (kStats) ++leafVisits;
				pt_leaf_test<true, kLdsQuads>(sc, fray, tmin, best, cur, triTests, 0u, nullptr, sQuads);
				if (kFlatCull)
				{
					// the other leaves the box test selected were selected against tmax: once something is hit, one whose box
					// begins behind that hit is skipped (the node step's test, t0 <= t * 1.000001f, with its near planes only)
					cur = PT_TRAVERSAL_DONE;
					const float limit = best.t * 1.000001f;
					while (pend)
					{
						const uint32_t bit = (uint32_t)__builtin_ctz(pend);
						pend &= pend - 1u;
						const uint32_t child = (uint32_t)(PT_ENTRY_WIDTH - 1) - bit;
						if (best.prim == PT_NO_HIT || pt_entry_child_t0(sEntry, child, fray, tmin) <= limit)
						{
							cur = reinterpret_cast<const uint32_t *>(sEntry + 2)[child];
							break;
						}
					}
				}
				else cur = pend ? pt_entry_next(sEntry, pend) : PT_TRAVERSAL_DONE;
			}
			continue;
		}
		// One step for the LARGER of the two groups (majority-phase stepping, see pt_bounce.hip).
		PT_T_MARK(PT_T_LOOP)
		bool waitsForEntry = false;
		if (kInst && kInstEntryHead)
		{
			// lanes at a top-level leaf (reference = PT_LEAF_FLAG | instance) enter their instance now -- all that wait, as soon as
			// K wait or no other lane has anything else to do -- and are lanes at a node (the bottom-level root) in the vote
			waitsForEntry = inst == PT_NO_INSTANCE && (cur & PT_LEAF_FLAG) && cur < PT_INSTANCE_END;
			const unsigned long long wm = __ballot(waitsForEntry);
			if (wm && (popcWave(wm) >= kInstEntryHead || !__ballot(!waitsForEntry && cur != PT_TRAVERSAL_DONE)))
			{
				if (waitsForEntry)
				{
					InstLoads L;
					loadInstance(L);
					enterInstance(L);
				}
				waitsForEntry = false;
			}
		}
		const unsigned long long nodeLanes = __ballot(!(cur & PT_LEAF_FLAG));
		const bool atLeaf = (cur & PT_LEAF_FLAG) && cur != PT_TRAVERSAL_DONE && !waitsForEntry;
		bool doNode, mine = atLeaf;  // mine: this lane belongs to the leaf group that steps
		if (kInst && kInstVote3)
		{
			// Two-level scenes have three leaf programs -- triangle leaf, instance entry (transform + ray preparation with its
			// divisions), instance exit -- and a wave pays for each that any of its lanes needs: vote among THREE groups
			// (nodes, triangle leaves, instance entries / exits) and run the largest.
			const bool instOp = atLeaf && (cur == PT_INSTANCE_END || inst == PT_NO_INSTANCE);
			const uint32_t nN = popcWave(nodeLanes), nT = popcWave(__ballot(atLeaf && !instOp)), nI = popcWave(__ballot(instOp));
			doNode = nN >= nT && nN >= nI;
			const bool doTri = !doNode && nT >= nI;
			mine = atLeaf && (instOp != doTri);
		}
		else
		{
			const unsigned long long leafLanes = __ballot(atLeaf);
			doNode = popcWave(nodeLanes) >= popcWave(leafLanes);
		}
		const PtLaneRay ray = { ox, oy, oz, ix, iy, iz, Sx, Sy, Sz, kpack };
		if (kStats)
		{
			if (doNode) ++nodeSteps;  // wave-uniform
			else ++leafSteps;
		}
		PT_T_MARK(doNode ? PT_T_NODE : PT_T_LEAF)
		// A/B switch, off (pt_wave_traverse.h): the lanes of a quad fetch each other's node records; all 64 lanes take part
		constexpr bool kCoop = kNodeMode == PT_NODES_GLOBAL && PT_NODE_COOP != 0;
		PtNodeRegs pre;
		if (kCoop && doNode) pt_node_fetch_coop(sc.nodes, cur, lane, pre);
		if (doNode && !(cur & PT_LEAF_FLAG))
		{
			if (kStats) ++nodeFetches;
			pt_node_step<kStackLds, kNodeMode, kCoop>(sc, ray, tmin, best, cur, sp, sStack, lane, mySpill, a.spillStride, sNodes,
			                                          pre);
		}
		if (!doNode && mine)
		{
			if (kInst && !kInstExitHead && cur == PT_INSTANCE_END) leaveInstance();
			else if (kInst && !kInstEntryHead && inst == PT_NO_INSTANCE)
			{
				if (cur == PT_EMPTY_CHILD) cur = pt_stack_pop<kStackLds>(sp, sStack, lane, mySpill, a.spillStride);
				else
				{
					InstLoads L;
					loadInstance(L);
					enterInstance(L);
				}
			}
			else
			{
				if (kStats && cur != PT_EMPTY_CHILD) ++leafVisits;
				pt_leaf_step<kStackLds, true>(sc, ray, tmin, best, cur, sp, sStack, lane, mySpill, a.spillStride, triTests,
				                        kInst ? inst : 0u, kInst ? &bestInst : nullptr);
			}
		}
	}

#ifdef PT_PATHS_TIMING
	PT_T_MARK(PT_T_LOOP)
	if (lane == 0 && a.trav)
		for (int i = 0; i < PT_T_SECTIONS; ++i) atomicAdd(&a.trav[6 + i], tcyc[i]);
#endif
	if (lane == 0 && raysFetched) PT_COLD(waveRays)[blockIdx.x] += raysFetched;  // one wave per slot: no atomic needed
	if (kStats)
	{
		unsigned long long x = nodeFetches, y = triTests, z = nHits, w = leafVisits;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
		{
			x += __shfl_xor(x, o);
			y += __shfl_xor(y, o);
			z += __shfl_xor(z, o);
			w += __shfl_xor(w, o);
		}
		if (lane == 0)
		{
			atomicAdd(&a.trav[0], x);
			atomicAdd(&a.trav[1], y);
			atomicAdd(&a.trav[2], z);
			atomicAdd(&a.trav[3], w);
			atomicAdd(&a.trav[4], (unsigned long long)nodeSteps);
			atomicAdd(&a.trav[5], (unsigned long long)leafSteps);
		}
	}
}

// k_accumulate for the item-major radiance layout of k_paths: q(t, i) = ((t / 64) * spp + i) * 64 + t % 64,
// t = pixel index inside the tile.
__global__ void __launch_bounds__(256)
    k_accumulate_items(const float4 *rad, const unsigned long long *lit, PtTile tile, uint32_t W, uint32_t spp, float4 *image,
                       uint32_t lframe0, uint32_t nFrames, int mode)
{
	// (a lane past the last pixel stays: the lanes of a wave fetch the mask words for one another; a WAVE past it leaves --
	// its pixel group does not exist, nor do its mask words)
	const uint32_t WH = tile.w * tile.h;
	const uint32_t tt = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & (PT_WAVE - 1);
	if (tt - lane >= WH) return;
	const bool valid = tt < WH;
	const uint32_t t = valid ? tt : 0u;
	const uint32_t p = (tile.y0 + t / tile.w) * W + pt_tile_column(tile, t % tile.w);  // image pixel of tile pixel t
	const size_t framePitch = (size_t)((WH + PT_WAVE - 1) / PT_WAVE) * spp * PT_WAVE;
	size_t base = (size_t)(tt / PT_WAVE) * spp * PT_WAVE + lane;  // (tt / 64 is the same for the 64 lanes of a wave)
	const float S = (float)spp;
	float4 out = (lframe0 && valid) ? image[p] : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
	// the frames of the launch one after the other, exactly as if each had been a dispatch of its own
	for (uint32_t f = 0; f < nFrames; ++f, base += framePitch)
	{
		float ax = 0.0f, ay = 0.0f, az = 0.0f;
		// A path that never reached a light has no record to read: its radiance is the +0 the sums start from.  The words of
		// the pixel group's samples are consecutive: one coalesced load per 64 samples, lane j holding sample j's word; the
		// records of four samples are requested together and added in sample order.
		const unsigned long long *words = lit + (base >> 6);
		for (uint32_t c = 0; c < spp; c += PT_WAVE)
		{
			const uint32_t n = spp - c < PT_WAVE ? spp - c : PT_WAVE;
			const unsigned long long mine = lane < n ? words[c + lane] : 0ull;
			const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
			for (uint32_t j = 0; j < n; j += 4)
			{
				float4 L[4];
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k)
				{
					const uint32_t jj = j + k < n ? j + k : n - 1u;  // (uniform)
					const uint32_t wl = __builtin_amdgcn_readlane(lo, jj), wh = __builtin_amdgcn_readlane(hi, jj);
					const bool on = valid && j + k < n && (((lane < 32u ? wl : wh) >> (lane & 31u)) & 1u);
					L[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					if (on) L[k] = rad[base + (size_t)(c + j + k) * PT_WAVE];
				}
#pragma unroll
				for (uint32_t k = 0; k < 4; ++k)
					if (j + k < n)
					{
						ax += L[k].x;
						ay += L[k].y;
						az += L[k].z;
					}
			}
		}
		float cx = ax / S, cy = ay / S, cz = az / S;
		const uint32_t lframe = lframe0 + f;
		if (lframe == 0) out = make_float4(cx, cy, cz, 1.0f);
		else if (mode == PT_ACCUM_SUM) out = make_float4(out.x + cx, out.y + cy, out.z + cz, 1.0f);
		else
		{
			float al = 1.0f / (float)(lframe + 1);
			float be = 1.0f - al;
			out = make_float4(out.x * be + cx * al, out.y * be + cy * al, out.z * be + cz * al, 1.0f);  // mix(old,new,a)
		}
	}
	if (valid) image[p] = out;
}
}  // namespace

namespace
{
__global__ void k_store_camera(PtCamera cam, PtCamera *dst)
{
	const float *src = cam.invView;  // the struct is 44 contiguous floats
	if (threadIdx.x < sizeof(PtCamera) / 4) dst->invView[threadIdx.x] = src[threadIdx.x];
}
}  // namespace

void ptLaunchStoreCamera(const PtCamera &cam, PtCamera *dst, hipStream_t s) { k_store_camera<<<1, 64, 0, s>>>(cam, dst); }

namespace
{
// The host's table of ensureLcgSkip (pt_api.hip), entry by entry: u32 wrap-around products, so the same numbers.
__global__ void k_lcg_skip(uint32_t spp, uint2 *dst)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= spp) return;
	uint32_t A = 1u, C = 0u;
	for (uint32_t k = 0; k < 2u * i; ++k)
	{
		A = 1664525u * A;
		C = 1664525u * C + 1013904223u;
	}
	dst[i] = make_uint2(A, C);
}
}  // namespace

void ptLaunchLcgSkip(uint32_t spp, uint2 *dst, hipStream_t s) { k_lcg_skip<<<(spp + 63) / 64, 64, 0, s>>>(spp, dst); }

void ptLaunchPaths(const PtPathsArgs &args, uint32_t waves, bool stats, hipStream_t s)
{
	// A/B knobs: PT_LDS_NODES_OFF = nodes from memory whatever the size, PT_LDS_PLANES_OFF = packed LDS copy only
	static const bool ldsOff = getenv("PT_LDS_NODES_OFF") != nullptr, planesOff = getenv("PT_LDS_PLANES_OFF") != nullptr;
	const uint32_t nNodes = args.sc.nNodes;
#define PT_LAUNCH(inst, mode)                                                                        \
	{                                                                                                \
		if (args.entry)                                                                              \
		{                                                                                            \
			if (stats) k_paths<true, inst, mode, PT_ENTRY_TREE><<<waves, PT_WAVE, 0, s>>>(args);     \
			else k_paths<false, inst, mode, PT_ENTRY_TREE><<<waves, PT_WAVE, 0, s>>>(args);          \
		}                                                                                            \
		else if (stats) k_paths<true, inst, mode, PT_ENTRY_NONE><<<waves, PT_WAVE, 0, s>>>(args);    \
		else k_paths<false, inst, mode, PT_ENTRY_NONE><<<waves, PT_WAVE, 0, s>>>(args);              \
	}
	if (args.nInst) PT_LAUNCH(true, PT_NODES_GLOBAL)
	else if (args.entry && args.entryFlat && !args.nInst && args.sc.nTris < 255u)  // (a primitive id is a byte in the flat kernel's rows)
	{
		if (stats) k_paths<true, false, PT_NODES_GLOBAL, PT_ENTRY_FLAT><<<waves, PT_WAVE, 0, s>>>(args);
		else k_paths<false, false, PT_NODES_GLOBAL, PT_ENTRY_FLAT><<<waves, PT_WAVE, 0, s>>>(args);
	}
	else if (nNodes && nNodes <= PT_LDS_NODES_PLANES && !ldsOff && !planesOff) PT_LAUNCH(false, PT_NODES_PLANES)
	else if (nNodes && nNodes <= PT_LDS_NODES_PACKED && !ldsOff) PT_LAUNCH(false, PT_NODES_PACKED)
	else PT_LAUNCH(false, PT_NODES_GLOBAL)
#undef PT_LAUNCH
}

void ptLaunchAccumulateItems(const float4 *rad, const unsigned long long *lit, PtTile tile, uint32_t W, uint32_t spp, float4 *image,
                             uint32_t lframe0, uint32_t nFrames, int mode, hipStream_t s)
{
	k_accumulate_items<<<(tile.w * tile.h + 255) / 256, 256, 0, s>>>(rad, lit, tile, W, spp, image, lframe0, nFrames, mode);
}
