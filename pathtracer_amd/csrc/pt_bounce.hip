// k_bounce: one bounce of every queued path in ONE persistent kernel -- traceRayEXT
// (raytrace.rgen:65-76) + raytrace.rchit / raytrace.rmiss + the throughput update of
// raytrace.rgen:78-79 -- with stream compaction into the next bounce's queue.
//
// Why one kernel.  Measured on MI355X (profiles/r01a_*): the split extend/shade pair spends its
// time in VALU at ~34 % lane utilisation (lanes of a wave finish their traversal at different
// times) and the shade kernel is a pure HBM round trip of the hit and ray records.  Here
//   * every wave is persistent and refills lanes the moment they finish ("in-wave wavefront"):
//     rays are PREPARED 64 at a time at full utilisation (coalesced queue read, shear constants,
//     reciprocal direction) into an LDS staging area, idle lanes pick prepared rays up by ballot
//     rank, finished lanes park their hit in an LDS ring, and the ring is SHADED 64 at a time at
//     full utilisation -- only the traversal loop itself runs with partially filled waves;
//   * the hit record never goes to HBM, the ray record is read once;
//   * compaction costs no per-batch atomics: the queue is a sequence of 1024-record chunks with
//     a count each.  A wave claims input chunks with one atomic per chunk and allocates output
//     chunks with one atomic per 1024 survivors; survivors are packed by ballot + prefix popcount.
// All arithmetic that reaches a result is the shared specification of pt_device_common.h, so
// images stay bit-identical to the split pipeline and to the CPU oracle.
#include "pt_trace.h"
#include "pt_wave_traverse.h"

#ifndef PT_NODE_WEIGHT
#define PT_NODE_WEIGHT 1u
#define PT_LEAF_WEIGHT 1u
#endif
#ifndef PT_REFILL_LANES
#define PT_REFILL_LANES 16
#endif

namespace
{
constexpr int kStackLds = 12;  // traversal stack entries per lane held in LDS; deeper ones spill to global
constexpr int kInFields = 13;  // ox oy oz ix iy iz Sx Sy Sz kpack(kz | sign bits) seed id rayIdx
constexpr int kDoneFields = 7; // V W det prim seed id rayIdx
constexpr int kDoneRing = 128;
constexpr uint32_t kRefillLanes = PT_REFILL_LANES;  // finished lanes that trigger a retire + refill pass
constexpr uint32_t kUnit = 64;   // records per work claim unit (= one staging batch)
constexpr uint32_t kUnitsPerChunk = PT_CHUNK / kUnit;

__device__ __forceinline__ uint32_t popc64(unsigned long long m) { return (uint32_t)__popcll(m); }

template <bool kStats>
__global__ void __launch_bounds__(PT_WAVE)
    k_bounce(PtSceneDev sc, const float4 *__restrict__ rayOIn, const float4 *__restrict__ rayDIn,
             const float4 *__restrict__ thrIn, float4 *__restrict__ rayOOut, float4 *__restrict__ rayDOut,
             float4 *__restrict__ thrOut, const uint32_t *__restrict__ chunkCountIn, uint32_t *__restrict__ chunkCountOut,
             const uint32_t *numChunksInPtr, uint32_t *chunkHead, uint32_t *numChunksOut, uint32_t *waveRays,
             uint32_t denseN, float4 *__restrict__ rad, float tmin, float tmax, int lastBounce, uint32_t *spill,
             uint32_t spillStride, unsigned long long *trav)
{
	__shared__ uint32_t sStack[kStackLds][PT_WAVE];
	__shared__ uint32_t sIn[kInFields][PT_WAVE];
	__shared__ uint32_t sDone[kDoneFields][kDoneRing];

	const int lane = threadIdx.x;
	const unsigned long long laneBit = 1ull << lane, lanesBelow = laneBit - 1ull;
	const uint32_t numChunksIn = *numChunksInPtr;
	uint32_t *mySpill = spill + (size_t)blockIdx.x * PT_WAVE + lane;

	// wave-uniform cursors
	uint32_t chunkBase = 0, chunkPos = 0, chunkCnt = 0;  // input unit being consumed (a quarter of a chunk)
	const uint32_t totalUnits = numChunksIn * kUnitsPerChunk;
	uint32_t unitCur = 0, unitEnd = 0;                   // claimed units [unitCur, unitEnd)
	const uint32_t firstShare = totalUnits / (2u * gridDim.x);
	const uint32_t firstK = firstShare >= kUnitsPerChunk ? kUnitsPerChunk : (firstShare == 0 ? 1u : firstShare);
	const uint32_t headBias = gridDim.x * firstK;        // units handed out statically
	bool firstClaim = true;
	uint32_t lastSeen = headBias;
	bool exhausted = (numChunksIn == 0);
	uint32_t inCount = 0, inPos = 0;                     // prepared rays in sIn
	uint32_t doneHead = 0, doneTail = 0;                 // ring of finished rays
	uint32_t outChunk = 0xFFFFFFFFu, outPos = PT_CHUNK;  // output chunk being filled
	uint32_t raysFetched = 0;
	uint32_t nodeFetches = 0, triTests = 0, nHits = 0;

	// per-lane ray state
	float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0, Sx = 0, Sy = 0, Sz = 0;
	uint32_t kpack = 0, seed = 0, id = 0, rayIdx = 0xFFFFFFFFu;
	PtBest best = { tmax, PT_NO_HIT, 0.0f, 0.0f, 1.0f };
	uint32_t cur = PT_TRAVERSAL_DONE;
	int sp = 0;

	// Shades up to 64 parked hits (ring entries doneHead .. doneHead+count) and emits the survivors.
	auto flush = [&](uint32_t count) {
		bool alive = false;
		float3x no = { 0, 0, 0 }, nd = { 0, 0, 0 };
		float tx = 0, ty = 0, tz = 0;
		uint32_t nseed = 0, nid = 0;
		if ((uint32_t)lane < count)
		{
			uint32_t e = (doneHead + lane) & (kDoneRing - 1);
			uint32_t prim = sDone[3][e];
			if (prim != PT_NO_HIT)  // rmiss: hitValue = 0, emission = 0, terminate -> nothing to do
			{
				float V = __uint_as_float(sDone[0][e]), W = __uint_as_float(sDone[1][e]), det = __uint_as_float(sDone[2][e]);
				nseed = sDone[4][e];
				nid = sDone[5][e];
				uint32_t ridx = sDone[6][e];
				float u = V / det, v = W / det;
				float4 T = thrIn[ridx];
				float3x kd, ke;
				pt_shade_hit(sc, prim, u, v, nseed, no, nd, kd, ke);
				float ex = ke.x, ey = ke.y, ez = ke.z;
				if (ex != 0.0f || ey != 0.0f || ez != 0.0f)
				{
					// accumDepthHitValue += emission * curAttenuation (rgen:78); one thread owns a path
					float4 L = rad[nid];
					L.x += ex * T.x;
					L.y += ey * T.y;
					L.z += ez * T.z;
					rad[nid] = L;
				}
				alive = !lastBounce && !((ex + ey) + ez > 0.0f);  // rchit:56-58
				tx = T.x * kd.x;  // curAttenuation *= hitValue (rgen:79)
				ty = T.y * kd.y;
				tz = T.z * kd.z;
				if (kStats) ++nHits;
			}
		}
		doneHead += count;
		unsigned long long am = __ballot(alive);
		uint32_t total = popc64(am), rank = popc64(am & lanesBelow), emitted = 0;
		while (emitted < total)
		{
			if (outPos == PT_CHUNK)
			{
				// close the full chunk, open a new one (one atomic per 1024 survivors)
				uint32_t c = 0;
				if (lane == 0)
				{
					if (outChunk != 0xFFFFFFFFu) chunkCountOut[outChunk] = PT_CHUNK;
					c = atomicAdd(numChunksOut, 1u);
				}
				outChunk = __builtin_amdgcn_readfirstlane(c);
				outPos = 0;
			}
			uint32_t take = total - emitted;
			if (take > PT_CHUNK - outPos) take = PT_CHUNK - outPos;
			if (alive && rank >= emitted && rank < emitted + take)
			{
				size_t dst = (size_t)outChunk * PT_CHUNK + outPos + (rank - emitted);
				rayOOut[dst] = make_float4(no.x, no.y, no.z, __uint_as_float(nseed));
				rayDOut[dst] = make_float4(nd.x, nd.y, nd.z, __uint_as_float(nid));
				thrOut[dst] = make_float4(tx, ty, tz, 0.0f);
			}
			outPos += take;
			emitted += take;
		}
	};

	for (;;)
	{
		// Retire + refill only when enough lanes wait for it (or nothing else is left to do): the section
		// costs about half a node step, so it is not run for every single finished lane.
		const uint32_t nIdle = popc64(__ballot(cur == PT_TRAVERSAL_DONE));
		if (nIdle >= kRefillLanes)
		{
			// ------------------------------------------------------------------ C: retire finished lanes ----
			bool fin = (rayIdx != 0xFFFFFFFFu) && (cur == PT_TRAVERSAL_DONE);
			unsigned long long fm = __ballot(fin);
			if (fm)
			{
				if (fin)
				{
					uint32_t e = (doneTail + popc64(fm & lanesBelow)) & (kDoneRing - 1);
					sDone[0][e] = __float_as_uint(best.V);
					sDone[1][e] = __float_as_uint(best.W);
					sDone[2][e] = __float_as_uint(best.det);
					sDone[3][e] = best.prim;
					sDone[4][e] = seed;
					sDone[5][e] = id;
					sDone[6][e] = rayIdx;
					rayIdx = 0xFFFFFFFFu;
				}
				doneTail += popc64(fm);
				if (doneTail - doneHead >= PT_WAVE)
				{
					__syncthreads();
					flush(PT_WAVE);
					__syncthreads();
				}
			}
			// ------------------------------------------------------------------ A: refill idle lanes ----
			unsigned long long idle = __ballot(cur == PT_TRAVERSAL_DONE);
			while (idle)
			{
				if (inPos == inCount)
				{
					if (exhausted) break;
					if (chunkPos == chunkCnt)
					{
						// Work is claimed in units of 64 records (one staging batch; 16 per chunk) with guided
						// self-scheduling: up to 16 units per atomic while plenty of work is left (a single counter
						// sustains ~88 claims/us), fewer near the end so the last waves finish within a batch of
						// each other.  The first claim of every wave is static (no atomic: the counter is read as
						// if it started at gridDim * firstK), and a wave that can see the queue is exhausted
						// leaves without touching the counter.
						if (unitCur == unitEnd)
						{
							uint32_t c, k;
							if (firstClaim)
							{
								firstClaim = false;
								k = firstK;
								c = blockIdx.x * firstK;
							}
							else
							{
								const uint32_t seen = lastSeen;  // counter value at this wave's previous claim
								if (seen >= totalUnits)
								{
									exhausted = true;
									break;
								}
								// guided: half of an even share of what is left, clamped to [4, 16] units: >= 256 records
								// per atomic keeps 4096 waves below the counter's throughput, <= 1024 bounds the tail
								const uint32_t share = (totalUnits - seen) / (2u * gridDim.x);
								k = share >= kUnitsPerChunk ? kUnitsPerChunk : (share <= 4u ? 4u : share);
								c = 0;
								if (lane == 0) c = atomicAdd(chunkHead, k);
								c = __builtin_amdgcn_readfirstlane(c) + headBias;
							}
							if (c >= totalUnits)
							{
								exhausted = true;
								break;
							}
							lastSeen = c + k;
							unitCur = c;
							unitEnd = c + k < totalUnits ? c + k : totalUnits;
						}
						const uint32_t u = unitCur++;
						const uint32_t chunk = u / kUnitsPerChunk, lo = (u % kUnitsPerChunk) * kUnit;
						uint32_t cnt;
						if (denseN)
						{
							uint32_t rem = denseN - chunk * PT_CHUNK;
							cnt = rem < PT_CHUNK ? rem : PT_CHUNK;
						}
						else cnt = __builtin_amdgcn_readfirstlane(chunkCountIn[chunk]);
						if (lo >= cnt)
						{
							// the rest of this chunk holds no records: skip to the next chunk boundary
							uint32_t nextChunkUnit = (chunk + 1) * kUnitsPerChunk;
							unitCur = nextChunkUnit < unitEnd ? nextChunkUnit : unitEnd;
							continue;
						}
						chunkBase = chunk * PT_CHUNK + lo;
						chunkPos = 0;
						chunkCnt = (cnt - lo < kUnit) ? cnt - lo : kUnit;
					}
					// prepare the next (up to) 64 rays of the chunk at full utilisation
					uint32_t n = chunkCnt - chunkPos;
					if (n > PT_WAVE) n = PT_WAVE;
					__syncthreads();  // all lanes are done reading the previous staging contents
					if ((uint32_t)lane < n)
					{
						uint32_t ridx = chunkBase + chunkPos + lane;
						float4 ro = rayOIn[ridx], rd = rayDIn[ridx];
						float3x d = { rd.x, rd.y, rd.z };
						PtLaneRay pr = pt_prepare_ray({ ro.x, ro.y, ro.z }, d);
						sIn[0][lane] = __float_as_uint(ro.x);
						sIn[1][lane] = __float_as_uint(ro.y);
						sIn[2][lane] = __float_as_uint(ro.z);
						sIn[3][lane] = __float_as_uint(pr.ix);
						sIn[4][lane] = __float_as_uint(pr.iy);
						sIn[5][lane] = __float_as_uint(pr.iz);
						sIn[6][lane] = __float_as_uint(pr.Sx);
						sIn[7][lane] = __float_as_uint(pr.Sy);
						sIn[8][lane] = __float_as_uint(pr.Sz);
						sIn[9][lane] = pr.kpack;
						sIn[10][lane] = __float_as_uint(ro.w);
						sIn[11][lane] = __float_as_uint(rd.w);
						sIn[12][lane] = ridx;
					}
					__syncthreads();
					chunkPos += n;
					inCount = n;
					inPos = 0;
					raysFetched += n;
				}
				uint32_t avail = inCount - inPos, want = popc64(idle);
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
					seed = sIn[10][s];
					id = sIn[11][s];
					rayIdx = sIn[12][s];
					best.t = tmax;
					best.prim = PT_NO_HIT;
					best.V = 0.0f;
					best.W = 0.0f;
					best.det = 1.0f;
					sStack[0][lane] = PT_TRAVERSAL_DONE;  // sentinel: popping an empty stack ends the traversal
					sp = 1;
					cur = sc.nNodes ? 0u : PT_TRAVERSAL_DONE;
				}
				inPos += n;
				idle &= ~__ballot(take);
			}
			if (!__ballot(rayIdx != 0xFFFFFFFFu)) break;  // input exhausted and every lane retired
		}

		// ------------------------------------------------------------------ B: traverse ----
		// One step for the LARGER of the two groups -- lanes on an inner node or lanes on a leaf -- the other
		// group waits: at least half of the busy lanes work in every step (a phase that runs until its group
		// is empty decays towards a single lane).
		const unsigned long long nodeLanes = __ballot(!(cur & PT_LEAF_FLAG));
		const unsigned long long leafLanes = __ballot((cur & PT_LEAF_FLAG) && cur != PT_TRAVERSAL_DONE);
		const bool doNode = popc64(nodeLanes) * PT_NODE_WEIGHT >= popc64(leafLanes) * PT_LEAF_WEIGHT;
		const PtLaneRay ray = { ox, oy, oz, ix, iy, iz, Sx, Sy, Sz, kpack };
		if (doNode && !(cur & PT_LEAF_FLAG))
		{
			if (kStats) ++nodeFetches;
			pt_node_step<kStackLds>(sc, ray, tmin, best, cur, sp, sStack, lane, mySpill, spillStride);
		}
		if (!doNode && (cur & PT_LEAF_FLAG) && cur != PT_TRAVERSAL_DONE)
			pt_leaf_step<kStackLds>(sc, ray, tmin, best, cur, sp, sStack, lane, mySpill, spillStride, triTests);
	}
	// drain the ring
	__syncthreads();
	if (doneTail != doneHead) flush(doneTail - doneHead);
	if (lane == 0)
	{
		if (outChunk != 0xFFFFFFFFu) chunkCountOut[outChunk] = outPos;
		if (raysFetched) waveRays[blockIdx.x] += raysFetched;  // one wave per slot per launch: no atomic needed
	}
	if (kStats)
	{
		unsigned long long a = nodeFetches, b = triTests, c = nHits;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
		{
			a += __shfl_xor(a, o);
			b += __shfl_xor(b, o);
			c += __shfl_xor(c, o);
		}
		if (lane == 0)
		{
			atomicAdd(&trav[0], a);
			atomicAdd(&trav[1], b);
			atomicAdd(&trav[2], c);
		}
	}
}
}  // namespace

void ptLaunchBounce(const PtSceneDev &sc, const PtQueues &q, int in, PtFrameCounters *fc, int depth, uint32_t denseN,
                    float tmin, float tmax, int lastBounce, const PtLaunchCfg &cfg, unsigned long long *travStats,
                    hipStream_t s)
{
	const int out = in ^ 1;
#define PT_BOUNCE(S)                                                                                                   \
	k_bounce<S><<<cfg.bounceWaves, PT_WAVE, 0, s>>>(sc, q.rayO[in], q.rayD[in], q.thr[in], q.rayO[out], q.rayD[out],   \
	                                                q.thr[out], q.chunkCount[in], q.chunkCount[out], &fc->numChunks[depth], \
	                                                &fc->chunkHead[depth], &fc->numChunks[depth + 1], cfg.waveRays, \
	                                                denseN, q.rad, tmin, tmax, lastBounce, cfg.spill, cfg.spillStride,   \
	                                                travStats)
	if (travStats) PT_BOUNCE(true);
	else PT_BOUNCE(false);
#undef PT_BOUNCE
}
