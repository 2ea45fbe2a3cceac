// Wavefront path tracer: the device side of VulkanRaytracing::raytrace (VulkanRaytracing.cpp:868-901),
// i.e. what vkCmdTraceRaysKHR(W,H,1) ran: shaders/raytrace.rgen (generate + bounce loop + accumulate),
// traceRayEXT (BVH traversal, done by the driver/RT hardware in the reference), raytrace.rchit and
// raytrace.rmiss (shade).  One dispatch of the reference = one "frame" here:
//
//   k_generate    raytrace.rgen:28-59   seed, jitter, camera ray for every (pixel, sample) path
//   k_extend      raytrace.rgen:65-76   closest hit per queued ray (stack-based quantised-BVH4 traversal,
//                                       watertight triangle test); persistent waves pull 1024-ray chunks
//   k_shade       rchit:24-59, rmiss:10-15, rgen:78-79   material + Lambert bounce, emission/miss terminate,
//                                       survivors compacted into the next queue (ballot + prefix popcount)
//   k_accumulate  raytrace.rgen:82-95   per-pixel sample sum in sample order, /spp, progressive mix
//
// Queue records are SoA float4 streams: rayO = (origin, rng state), rayD = (direction, path id),
// thr = (throughput, -), hits = PtHit.  Path id = sample * W*H + pixel.
#include "pt_trace.h"

namespace
{
constexpr int kBlock = 256;
constexpr int kChunk = PT_CHUNK;    // rays per work-queue claim (one atomic per wave per chunk)
constexpr int kLdsStack = 24;       // stack entries kept in LDS per lane; deeper entries spill to global

struct Stack
{
	uint32_t *lds;    // &ldsStack[0][threadIdx.x], stride kBlock
	uint32_t *spill;  // &spill[globalThread], stride spillStride
	uint32_t spillStride;
	int sp;
	__device__ __forceinline__ void push(uint32_t v)
	{
		if (sp < kLdsStack) lds[sp * kBlock] = v;
		else if (sp < kLdsStack + PT_SPILL_STACK) spill[(size_t)(sp - kLdsStack) * spillStride] = v;
		++sp;
	}
	__device__ __forceinline__ uint32_t pop()
	{
		if (sp == 0) return PT_TRAVERSAL_DONE;
		--sp;
		if (sp < kLdsStack) return lds[sp * kBlock];
		if (sp < kLdsStack + PT_SPILL_STACK) return spill[(size_t)(sp - kLdsStack) * spillStride];
		return PT_EMPTY_CHILD;  // entry was dropped on overflow (build rejects trees that can get here)
	}
};

// Closest hit against the wide BVH.  Culling is conservative (quantised boxes padded at build time,
// far plane padded here), every surviving triangle runs the exact pt_tri_test, and acceptance is
// order independent (t, then primitive id), so the result equals the brute-force definition.
template <bool kStats>
__device__ __forceinline__ PtHit traverseBvh4(const PtSceneDev &sc, float3x o, float3x d, float tmin, float tmax,
                                              Stack &st, uint32_t &nodeFetches, uint32_t &triTests)
{
	PtRayShear rs = pt_ray_shear(d);
	PtBest best = { tmax, PT_NO_HIT, 0.0f, 0.0f, 1.0f };
	// box test only: avoid 0 * inf by nudging zero components (conservative: interval stays huge)
	float3x dc = { fabsf(d.x) > 1e-30f ? d.x : copysignf(1e-30f, d.x), fabsf(d.y) > 1e-30f ? d.y : copysignf(1e-30f, d.y),
		           fabsf(d.z) > 1e-30f ? d.z : copysignf(1e-30f, d.z) };
	float3x inv = { 1.0f / dc.x, 1.0f / dc.y, 1.0f / dc.z };
	const bool negx = inv.x < 0.0f, negy = inv.y < 0.0f, negz = inv.z < 0.0f;
	st.sp = 0;
	uint32_t cur = sc.nNodes ? 0u : PT_TRAVERSAL_DONE;
	// while-while: all lanes that sit on an inner node step together, then all lanes on a leaf
	while (cur != PT_TRAVERSAL_DONE)
	{
		while (!(cur & PT_LEAF_FLAG))
		{
			if (kStats) ++nodeFetches;
			const uint4 *np = reinterpret_cast<const uint4 *>(sc.nodes + cur);
			uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
			float sx = __uint_as_float(n0.w), sy = __uint_as_float(n3.z), sz = __uint_as_float(n3.w);
			float ax = sx * inv.x, ay = sy * inv.y, az = sz * inv.z;
			float bx = (__uint_as_float(n0.x) - o.x) * inv.x, by = (__uint_as_float(n0.y) - o.y) * inv.y,
			      bz = (__uint_as_float(n0.z) - o.z) * inv.z;
			// n2 = qlo[0..2], qhi[0]; n3 = qhi[1..2], pad
			uint32_t nqx = negx ? n2.w : n2.x, fqx = negx ? n2.x : n2.w;
			uint32_t nqy = negy ? n3.x : n2.y, fqy = negy ? n2.y : n3.x;
			uint32_t nqz = negz ? n3.y : n2.z, fqz = negz ? n2.z : n3.y;
			uint32_t refs[4] = { n1.x, n1.y, n1.z, n1.w };
			uint32_t key[4];
#pragma unroll
			for (int c = 0; c < 4; ++c)
			{
				float tnx = __builtin_fmaf((float)((nqx >> (8 * c)) & 0xFFu), ax, bx);
				float tny = __builtin_fmaf((float)((nqy >> (8 * c)) & 0xFFu), ay, by);
				float tnz = __builtin_fmaf((float)((nqz >> (8 * c)) & 0xFFu), az, bz);
				float tfx = __builtin_fmaf((float)((fqx >> (8 * c)) & 0xFFu), ax, bx);
				float tfy = __builtin_fmaf((float)((fqy >> (8 * c)) & 0xFFu), ay, by);
				float tfz = __builtin_fmaf((float)((fqz >> (8 * c)) & 0xFFu), az, bz);
				float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
				float tf = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
				bool hit = (tn <= tf * 1.000001f) && (refs[c] != PT_EMPTY_CHILD);
				key[c] = hit ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : 0xFFFFFFFFu;
			}
			// sort the four keys ascending (nearest first): 5 compare-exchanges
#define PT_CE(a, b)                      \
	{                                    \
		uint32_t lo_ = min(key[a], key[b]); \
		uint32_t hi_ = max(key[a], key[b]); \
		key[a] = lo_;                    \
		key[b] = hi_;                    \
	}
			PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
#define PT_REF(k) ((k & 2u) ? ((k & 1u) ? refs[3] : refs[2]) : ((k & 1u) ? refs[1] : refs[0]))
			if (key[3] != 0xFFFFFFFFu) st.push(PT_REF(key[3]));
			if (key[2] != 0xFFFFFFFFu) st.push(PT_REF(key[2]));
			if (key[1] != 0xFFFFFFFFu) st.push(PT_REF(key[1]));
			cur = (key[0] != 0xFFFFFFFFu) ? PT_REF(key[0]) : st.pop();
#undef PT_REF
		}
		while ((cur & PT_LEAF_FLAG) && cur != PT_TRAVERSAL_DONE)
		{
			if (cur == PT_EMPTY_CHILD)
			{
				cur = st.pop();  // placeholder of an entry dropped on stack overflow
				continue;
			}
			uint32_t first = cur & 0x0FFFFFFFu;
			uint32_t cnt = ((cur >> 28) & 7u) + 1u;
			if (kStats) triTests += cnt;
			for (uint32_t i = 0; i < cnt; ++i)
			{
				const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + (size_t)(sc.triPlanes - 1u) * sc.nTris + first + i);  // identity plane
				float4 a = tp[0], b = tp[1], c = tp[2];
				float3x p0 = { a.x, a.y, a.z }, p1 = { a.w, b.x, b.y }, p2 = { b.z, b.w, c.x };
				pt_tri_test(o, rs, tmin, p0, p1, p2, __float_as_uint(c.y), best);
			}
			cur = st.pop();
		}
	}
	return pt_finish_hit(best, tmax);
}

// PT_ACCEL_BRUTE_FORCE: the definition itself, every triangle tested in primitive order.
__device__ __forceinline__ PtHit traverseBrute(const PtSceneDev &sc, float3x o, float3x d, float tmin, float tmax)
{
	PtRayShear rs = pt_ray_shear(d);
	PtBest best = { tmax, PT_NO_HIT, 0.0f, 0.0f, 1.0f };
	for (uint32_t p = 0; p < sc.nTris; ++p)
	{
		const float *a = sc.vertices[sc.indices[3 * (size_t)p + 0]].pos;
		const float *b = sc.vertices[sc.indices[3 * (size_t)p + 1]].pos;
		const float *c = sc.vertices[sc.indices[3 * (size_t)p + 2]].pos;
		pt_tri_test(o, rs, tmin, { a[0], a[1], a[2] }, { b[0], b[1], b[2] }, { c[0], c[1], c[2] }, p, best);
	}
	return pt_finish_hit(best, tmax);
}

// ------------------------------------------------------------------------ generate ----
__global__ void __launch_bounds__(kBlock)
    k_generate(PtCamera cam, uint32_t W, uint32_t H, PtTile tile, uint32_t spp, uint32_t seedWord, int centreJitter,
               float4 *rayO, float4 *rayD, float4 *thr, float4 *rad)
{
	uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t WH = tile.w * tile.h;  // paths are numbered inside the tile
	if (p >= WH) return;
	uint32_t px = pt_tile_column(tile, p % tile.w), py = tile.y0 + p / tile.w;
	uint32_t seed = pt_tea(py * W + px, seedWord);  // raytrace.rgen:28 with val1 = seed + global frame
	for (uint32_t i = 0; i < spp; ++i)
	{
		float r1 = pt_rnd(seed);
		float r2 = pt_rnd(seed);
		float jx = centreJitter ? 0.5f : r1, jy = centreJitter ? 0.5f : r2;
		float3x o, d;
		pt_camera_ray(cam, W, H, px, py, jx, jy, o, d);
		size_t id = (size_t)i * WH + p;
		// payload.seed = seed: the bounce stream starts from the state after the two jitter draws,
		// while the raygen-local seed continues from the same state for the next sample.
		rayO[id] = make_float4(o.x, o.y, o.z, __uint_as_float(seed));
		rayD[id] = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)id));
		thr[id] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
		rad[id] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	}
}

// -------------------------------------------------------------------------- extend ----
template <bool kBrute, bool kStats>
__global__ void __launch_bounds__(kBlock)
    k_extend(PtSceneDev sc, const uint32_t *count, uint32_t *head, const float4 *rayO, const float4 *rayD,
             float4 *hits, float tmin, float tmax, uint32_t *spill, uint32_t spillStride, unsigned long long *trav)
{
	uint32_t nodeFetches = 0, triTests = 0, nHits = 0;
	__shared__ uint32_t ldsStack[kBrute ? 1 : kLdsStack][kBlock];
	const uint32_t n = *count;
	const int lane = threadIdx.x & 63;
	Stack st;
	st.lds = &ldsStack[0][threadIdx.x];
	st.spill = spill + (size_t)blockIdx.x * kBlock + threadIdx.x;
	st.spillStride = spillStride;
	st.sp = 0;
	for (;;)
	{
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(head, (uint32_t)kChunk);
		base = __builtin_amdgcn_readfirstlane(base);
		if (base >= n) break;
		uint32_t end = base + kChunk < n ? base + kChunk : n;
		for (uint32_t i = base + lane; i < end; i += PT_WAVE)
		{
			float4 ro = rayO[i], rd = rayD[i];
			float3x o = { ro.x, ro.y, ro.z }, d = { rd.x, rd.y, rd.z };
			PtHit h;
			if (kBrute)
			{
				h = traverseBrute(sc, o, d, tmin, tmax);
				if (kStats) triTests += sc.nTris;
			}
			else h = traverseBvh4<kStats>(sc, o, d, tmin, tmax, st, nodeFetches, triTests);
			if (kStats && h.prim != PT_NO_HIT) ++nHits;
			hits[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.prim));
		}
	}
	if (kStats)
	{
		// per-wave sums, three atomics per wave at kernel end
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

// --------------------------------------------------------------------------- shade ----
__global__ void __launch_bounds__(kBlock)
    k_shade(PtSceneDev sc, const uint8_t *primEmissive, const uint32_t *count, uint32_t *head, const float4 *hits,
            const float4 *rayOIn, const float4 *rayDIn, const float4 *thrIn, float4 *rayOOut, float4 *rayDOut,
            float4 *thrOut, uint32_t *countOut, float4 *rad, int lastBounce)
{
	const uint32_t n = *count;
	const int lane = threadIdx.x & 63;
	const unsigned long long lanesBelow = (1ull << lane) - 1ull;
	constexpr int kRounds = kChunk / PT_WAVE;
	for (;;)
	{
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(head, (uint32_t)kChunk);
		base = __builtin_amdgcn_readfirstlane(base);
		if (base >= n) break;
		uint32_t end = base + kChunk < n ? base + kChunk : n;
		// pass 1: which paths of the chunk continue?  (hit a non-emissive primitive, not the last bounce)
		uint32_t aliveBits = 0;  // bit r = this lane's path of round r continues
		uint32_t total = 0;      // wave-uniform survivor count of the chunk
		if (!lastBounce)
		{
#pragma unroll 4
			for (int r = 0; r < kRounds; ++r)
			{
				uint32_t i = base + r * PT_WAVE + lane;
				bool alive = false;
				if (i < end)
				{
					uint32_t prim = __float_as_uint(hits[i].w);
					alive = prim != PT_NO_HIT && !primEmissive[prim];
				}
				unsigned long long m = __ballot(alive);
				aliveBits |= (alive ? 1u : 0u) << r;
				total += (uint32_t)__popcll(m);
			}
		}
		uint32_t outBase = 0;
		if (lane == 0 && total) outBase = atomicAdd(countOut, total);
		outBase = __builtin_amdgcn_readfirstlane(outBase);
		// pass 2: shade
		for (int r = 0; r < kRounds; ++r)
		{
			uint32_t i = base + r * PT_WAVE + lane;
			bool alive = (aliveBits >> r) & 1u;
			unsigned long long m = __ballot(alive);
			if (i < end)
			{
				float4 h = hits[i];
				uint32_t prim = __float_as_uint(h.w);
				if (prim != PT_NO_HIT)  // rmiss: hitValue = 0, emission = 0, terminate -> nothing to write
				{
					float4 ro = rayOIn[i], rd = rayDIn[i], T = thrIn[i];
					uint32_t seed = __float_as_uint(ro.w);
					uint32_t id = __float_as_uint(rd.w);
					float3x o, d;
					float3x kd, ke;
					pt_shade_hit(sc, prim, h.y, h.z, seed, o, d, kd, ke);
					float ex = ke.x, ey = ke.y, ez = ke.z;
					if (ex != 0.0f || ey != 0.0f || ez != 0.0f)
					{
						// accumDepthHitValue += emission * curAttenuation (rgen:78); each path owns rad[id]
						float4 L = rad[id];
						L.x += ex * T.x;
						L.y += ey * T.y;
						L.z += ez * T.z;
						rad[id] = L;
					}
					if (alive)
					{
						uint32_t dst = outBase + (uint32_t)__popcll(m & lanesBelow);
						rayOOut[dst] = make_float4(o.x, o.y, o.z, __uint_as_float(seed));
						rayDOut[dst] = make_float4(d.x, d.y, d.z, __uint_as_float(id));
						thrOut[dst] = make_float4(T.x * kd.x, T.y * kd.y, T.z * kd.z, 0.0f);  // rgen:79
					}
				}
			}
			outBase += (uint32_t)__popcll(m);
		}
	}
}

// ---------------------------------------------------------------------- accumulate ----
// raytrace.rgen:82-95.  mode PT_ACCUM_MEAN: running mean with a = 1/(lframe+1); PT_ACCUM_SUM: raw sums.
__global__ void __launch_bounds__(kBlock)
    k_accumulate(const float4 *rad, PtTile tile, uint32_t W, uint32_t spp, float4 *image, uint32_t lframe, int mode)
{
	uint32_t t = blockIdx.x * kBlock + threadIdx.x;
	const uint32_t WH = tile.w * tile.h;
	if (t >= WH) return;
	const uint32_t p = (tile.y0 + t / tile.w) * W + pt_tile_column(tile, t % tile.w);  // image pixel of tile pixel t
	float ax = 0.0f, ay = 0.0f, az = 0.0f;
	for (uint32_t i = 0; i < spp; ++i)
	{
		float4 L = rad[(size_t)i * WH + t];
		ax += L.x;
		ay += L.y;
		az += L.z;
	}
	float S = (float)spp;
	float cx = ax / S, cy = ay / S, cz = az / S;
	float4 out;
	if (lframe == 0) out = make_float4(cx, cy, cz, 1.0f);
	else
	{
		float4 old = image[p];
		if (mode == PT_ACCUM_SUM) out = make_float4(old.x + cx, old.y + cy, old.z + cz, 1.0f);
		else
		{
			float a = 1.0f / (float)(lframe + 1);
			float b = 1.0f - a;
			out = make_float4(old.x * b + cx * a, old.y * b + cy * a, old.z * b + cz * a, 1.0f);  // mix(old,new,a)
		}
	}
	image[p] = out;
}

// ------------------------------------------------------------------- frame counters ----
__global__ void k_frame_begin(PtFrameCounters *fc, uint32_t nPaths, int fused)
{
	uint32_t *w = reinterpret_cast<uint32_t *>(fc);
	for (uint32_t i = threadIdx.x; i < sizeof(PtFrameCounters) / 4; i += blockDim.x) w[i] = 0;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		if (fused) fc->numChunks[0] = (nPaths + PT_CHUNK - 1) / PT_CHUNK;  // count[d] is summed by the waves
		else fc->count[0] = nPaths;
	}
}

__global__ void k_frame_end(const PtFrameCounters *fc, int depth, unsigned long long *stats, uint32_t *waveRays,
                            uint32_t nWaveRays)
{
	// split pipeline: rays = sum of the queue lengths; fused pipeline: sum of the per-wave fetch counters
	__shared__ unsigned long long part[256];
	unsigned long long r = 0;
	for (uint32_t i = threadIdx.x; i < nWaveRays; i += blockDim.x)
	{
		r += waveRays[i];
		waveRays[i] = 0;
	}
	part[threadIdx.x] = r;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		unsigned long long total = 0;
		for (int i = 0; i < 256; ++i) total += part[i];
		for (int d = 0; d < depth; ++d) total += fc->count[d];
		stats[0] += total;  // rays_total
		stats[1] = total;   // rays_last_frame
	}
}

// ------------------------------------------------------------------- debug kernels ----
__global__ void k_debug_rng(uint32_t n, const uint32_t *v0, const uint32_t *v1, uint32_t nSteps, uint32_t *outTea,
                            uint32_t *outLcg, float *outUnit)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t s = pt_tea(v0[i], v1[i]);
	outTea[i] = s;
	uint32_t st = s;
	for (uint32_t k = 0; k < nSteps; ++k) outLcg[(size_t)i * nSteps + k] = pt_lcg(st);
	uint32_t s2 = s;
	float3x u = pt_rnd_unit_vector(s2);
	outUnit[3 * (size_t)i + 0] = u.x;
	outUnit[3 * (size_t)i + 1] = u.y;
	outUnit[3 * (size_t)i + 2] = u.z;
}

// In-range sqrt / division forms (pt_device_common.h) against the plain operators, on operands that sweep the exponent
// ranges the callers' checks admit (2^+-40 for divisions, 2^+-60 for roots) with random signs and mantissas; the exact
// ends of the ranges are visited too.  mismatches[0]: quotients, [1]: roots, [2]: reciprocal-of-root (normalize), [3]: sets checked.
__global__ void k_debug_math(unsigned long long n, uint32_t seed, unsigned long long *mismatches)
{
	const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (threadIdx.x == 0)
	{
		// [3]: operand sets this launch really went through (a launch that did not run must not read as "no mismatch")
		const unsigned long long first = (unsigned long long)blockIdx.x * blockDim.x;
		atomicAdd(&mismatches[3], n - first < blockDim.x ? n - first : (unsigned long long)blockDim.x);
	}
	if (i >= n) return;
	uint32_t st = pt_tea((uint32_t)i, seed + (uint32_t)(i >> 32));
	auto draw = [&](int lim, uint32_t k) {
		// exponent uniformly in [-lim, lim] (every 64th sample pinned to an end), 23 random mantissa bits, random sign
		pt_lcg(st);
		uint32_t r = st;
		int e = (int)((r >> 8) % (uint32_t)(2 * lim + 1)) - lim;
		if (((i + k) & 63u) == 0u) e = (r & 1u) ? lim : -lim;
		pt_lcg(st);
		uint32_t m = st & 0x7FFFFFu;
		if (e == lim) m = 0u;  // the upper end is 2^lim itself
		return __uint_as_float(((st >> 23) & 1u) << 31 | (uint32_t)(e + 127) << 23 | m);
	};
	const float d = draw(40, 0), n1 = draw(40, 1), n2 = draw(40, 2);
	unsigned long long bad0 = 0, bad1 = 0, bad2 = 0;
	if (pt_all_in_division_range(d, n1, n2))
	{
		const PtSharedDivisor by(d);
		bad0 += __float_as_uint(by.quotient(n1)) != __float_as_uint(n1 / d);
		bad0 += __float_as_uint(by.quotient(n2)) != __float_as_uint(n2 / d);
		bad0 += __float_as_uint(by.quotient(1.0f)) != __float_as_uint(1.0f / d);
	}
	else bad0 += 1;  // the generator stays inside the range: the check must have passed
	const float x = fabsf(draw(60, 3));
	if (pt_all_in_range(x))
	{
		const float root = pt_sqrt_in_range(x);
		bad1 += __float_as_uint(root) != __float_as_uint(__builtin_sqrtf(x));
		bad2 += __float_as_uint(PtSharedDivisor(root).quotient(1.0f)) != __float_as_uint(1.0f / __builtin_sqrtf(x));
	}
	else bad1 += 1;
	if (bad0) atomicAdd(&mismatches[0], bad0);
	if (bad1) atomicAdd(&mismatches[1], bad1);
	if (bad2) atomicAdd(&mismatches[2], bad2);
}

__global__ void k_pack_rays(const float *rays, uint32_t n, float4 *rayO, float4 *rayD)
{
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	rayO[i] = make_float4(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2], 0.0f);
	rayD[i] = make_float4(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5], 0.0f);
}

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// post.frag + sRGB attachment: linear -> 8-bit sRGB (alpha linear)
__global__ void __launch_bounds__(kBlock) k_srgb8(const float4 *image, uint32_t n, uchar4 *out)
{
	uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	float4 c = image[i];
	float v[4] = { c.x, c.y, c.z, c.w };
	unsigned char o[4];
#pragma unroll
	for (int k = 0; k < 4; ++k)
	{
		float x = v[k];
		if (!(x > 0.0f)) x = 0.0f;
		if (x > 1.0f) x = 1.0f;
		float e = x;
		if (k < 3) e = x <= 0.0031308f ? x * 12.92f : 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
		o[k] = (unsigned char)(e * 255.0f + 0.5f);
	}
	out[i] = make_uchar4(o[0], o[1], o[2], o[3]);
}
}  // namespace

static inline uint32_t divUp(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

void ptLaunchFrameBegin(PtFrameCounters *fc, uint32_t nPaths, int fused, hipStream_t s)
{
	k_frame_begin<<<1, 256, 0, s>>>(fc, nPaths, fused);
}

void ptLaunchFrameEnd(const PtFrameCounters *fc, int depth, unsigned long long *stats, uint32_t *waveRays,
                      uint32_t nWaveRays, hipStream_t s)
{
	k_frame_end<<<1, 256, 0, s>>>(fc, depth, stats, waveRays, nWaveRays);
}

void ptLaunchGenerate(const PtCamera &cam, uint32_t W, uint32_t H, PtTile tile, uint32_t spp, uint32_t seedWord,
                      int centreJitter, const PtQueues &q, hipStream_t s)
{
	k_generate<<<divUp((size_t)tile.w * tile.h, kBlock), kBlock, 0, s>>>(cam, W, H, tile, spp, seedWord, centreJitter, q.rayO[0],
	                                                          q.rayD[0], q.thr[0], q.rad);
}

void ptLaunchExtend(const PtSceneDev &sc, bool brute, const uint32_t *count, uint32_t *head, const float4 *rayO,
                    const float4 *rayD, float4 *hits, float tmin, float tmax, const PtLaunchCfg &cfg,
                    unsigned long long *travStats, hipStream_t s)
{
#define PT_EXT(B, S)                                                                                                 \
	k_extend<B, S><<<cfg.persistentBlocks, kBlock, 0, s>>>(sc, count, head, rayO, rayD, hits, tmin, tmax, cfg.spill, \
	                                                       cfg.spillStride, travStats)
	if (brute)
	{
		if (travStats) PT_EXT(true, true);
		else PT_EXT(true, false);
	}
	else
	{
		if (travStats) PT_EXT(false, true);
		else PT_EXT(false, false);
	}
#undef PT_EXT
}

void ptLaunchShade(const PtSceneDev &sc, const uint8_t *primEmissive, const uint32_t *count, uint32_t *head,
                   const PtQueues &q, int in, uint32_t *countOut, int lastBounce, const PtLaunchCfg &cfg, hipStream_t s)
{
	k_shade<<<cfg.persistentBlocks, kBlock, 0, s>>>(sc, primEmissive, count, head, q.hits, q.rayO[in], q.rayD[in],
	                                                q.thr[in], q.rayO[in ^ 1], q.rayD[in ^ 1], q.thr[in ^ 1], countOut,
	                                                q.rad, lastBounce);
}

void ptLaunchAccumulate(const PtQueues &q, PtTile tile, uint32_t W, uint32_t spp, float4 *image, uint32_t lframe, int mode,
                        hipStream_t s)
{
	k_accumulate<<<divUp((size_t)tile.w * tile.h, kBlock), kBlock, 0, s>>>(q.rad, tile, W, spp, image, lframe, mode);
}

void ptLaunchDebugRng(uint32_t n, const uint32_t *v0, const uint32_t *v1, uint32_t nSteps, uint32_t *outTea,
                      uint32_t *outLcg, float *outUnit, hipStream_t s)
{
	k_debug_rng<<<divUp(n, 256), 256, 0, s>>>(n, v0, v1, nSteps, outTea, outLcg, outUnit);
}

void ptLaunchDebugMath(unsigned long long n, uint32_t seed, unsigned long long *mismatches, hipStream_t s)
{
	k_debug_math<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(n, seed, mismatches);
}

void ptLaunchPackRays(const float *rays, uint32_t n, float4 *rayO, float4 *rayD, hipStream_t s)
{
	k_pack_rays<<<divUp(n, 256), 256, 0, s>>>(rays, n, rayO, rayD);
}

// raytrace.rchit:27-37's fetches, done once per primitive (PtShadeRec, pt_device_common.h)
__global__ void __launch_bounds__(kBlock) k_build_shade_records(PtSceneDev sc, PtShadeRec *out)
{
	uint32_t prim = blockIdx.x * kBlock + threadIdx.x;
	if (prim >= sc.nTris) return;
	PtShadeRec r;
	const float *mat = sc.materials[sc.matIndices[prim]].diffuse;
	for (int j = 0; j < 3; ++j)
	{
		const float *v = sc.vertices[sc.indices[3 * (size_t)prim + j]].pos;
		for (int k = 0; k < 3; ++k)
		{
			r.pos[j][k] = v[k];
			r.nrm[j][k] = v[3 + k];
		}
		r.kd[j] = mat[j];
		r.ke[j] = mat[3 + j];
	}
	out[prim] = r;
}

void ptLaunchBuildShadeRecords(const PtSceneDev &sc, PtShadeRec *out, hipStream_t s)
{
	if (sc.nTris) k_build_shade_records<<<divUp(sc.nTris, kBlock), kBlock, 0, s>>>(sc, out);
}

void ptLaunchSetU32(uint32_t *p, uint32_t v, hipStream_t s) { k_set_u32<<<1, 1, 0, s>>>(p, v); }

void ptLaunchSrgb8(const float4 *image, uint32_t n, void *out, hipStream_t s)
{
	k_srgb8<<<divUp(n, kBlock), kBlock, 0, s>>>(image, n, reinterpret_cast<uchar4 *>(out));
}
