// Acceleration-structure build on the device: what the Vulkan driver did inside
// vkCmdBuildAccelerationStructuresKHR for VulkanRaytracing::buildBlas/buildTlas
// (VulkanRaytracing.cpp:147-262, 448-527).  Nothing here has a counterpart in the reference's
// own sources; the pipeline is
//
//   k_tri_setup      triangle AABBs + centroid bounds (wave reduce, one atomic set per block)
//   k_morton         30-bit Morton code per triangle (definition: DESIGN.md 5.1, oracle pto_morton_codes)
//   radix sort       4 passes x 8 bits, LDS histogram / scan / stable LDS-ranked scatter
//   k_lbvh_topology  Karras 2012 binary radix tree over the sorted codes
//   k_refit          bottom-up AABB union, one agent-scope counter per internal node
//   k_collapse_level BVH2 -> quantised BVH4 (64-byte nodes), level by level, leaf triangles
//                    re-laid out as 48-byte records in leaf order
#include "pt_build.h"

#include <float.h>
#include <stdlib.h>
#include <string.h>

namespace
{
constexpr int kBlock = 256;

// ---- order-preserving float <-> int so integer atomics give float min/max ----
__device__ __forceinline__ int floatToOrdered(float f)
{
	int i = __float_as_int(f);
	return i ^ ((i >> 31) & 0x7FFFFFFF);
}
__host__ __device__ __forceinline__ float orderedToFloat(int i)
{
	int b = i ^ ((i >> 31) & 0x7FFFFFFF);
	union
	{
		int i;
		float f;
	} u;
	u.i = b;
	return u.f;
}

__device__ __forceinline__ float waveMin(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
	return v;
}
__device__ __forceinline__ float waveMax(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
	return v;
}

// bounds[0..2] = centroid min (ordered int), [3..5] = centroid max, [6] = max |coordinate| (float bits)
// triPos: the three positions of every primitive as one 48-byte record (the layout of a plane-2 leaf triangle without the
// id), written here where the vertices are gathered anyway in primitive order; k_write_tris then needs ONE gathered
// record per triangle in leaf order instead of an index triple and three vertices.
__global__ void __launch_bounds__(kBlock) k_tri_setup(const PtVertex *verts, const uint32_t *indices, uint32_t nTris,
                                                      float4 *triLo, float4 *triHi, int *bounds, float4 *triPos, float *centroid)
{
	float cmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, cmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
	float amax = 0.0f;
	for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < nTris; i += gridDim.x * kBlock)
	{
		float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
		float q[3][3];
#pragma unroll
		for (int j = 0; j < 3; ++j)
		{
			const float *p = verts[indices[3 * (size_t)i + j]].pos;
#pragma unroll
			for (int k = 0; k < 3; ++k)
			{
				float c = p[k];
				q[j][k] = c;
				if (c < lo[k]) lo[k] = c;
				if (c > hi[k]) hi[k] = c;
			}
		}
		triPos[3 * (size_t)i + 0] = make_float4(q[0][0], q[0][1], q[0][2], q[1][0]);
		triPos[3 * (size_t)i + 1] = make_float4(q[1][1], q[1][2], q[2][0], q[2][1]);
		triPos[3 * (size_t)i + 2] = make_float4(q[2][2], 0.0f, 0.0f, 0.0f);
		if (!centroid)  // (uniform)
		{
			triLo[2 * (size_t)i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
			triHi[2 * (size_t)i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
		}
#pragma unroll
		for (int k = 0; k < 3; ++k)
		{
			float c = (lo[k] + hi[k]) * 0.5f;
			if (centroid) centroid[3 * (size_t)i + k] = c;
			if (c < cmin[k]) cmin[k] = c;
			if (c > cmax[k]) cmax[k] = c;
			amax = fmaxf(amax, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
		}
	}
	__shared__ float red[7][kBlock / PT_WAVE];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		float a = waveMin(cmin[k]), b = waveMax(cmax[k]);
		if (lane == 0)
		{
			red[k][wave] = a;
			red[3 + k][wave] = b;
		}
	}
	amax = waveMax(amax);
	if (lane == 0) red[6][wave] = amax;
	__syncthreads();
	if (threadIdx.x < 7)
	{
		float v = red[threadIdx.x][0];
		for (int w = 1; w < kBlock / PT_WAVE; ++w)
			v = (threadIdx.x < 3) ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
		if (threadIdx.x < 3)
			atomicMin(&bounds[threadIdx.x], floatToOrdered(v));
		else if (threadIdx.x < 6)
			atomicMax(&bounds[threadIdx.x], floatToOrdered(v));
		else
			atomicMax(reinterpret_cast<unsigned int *>(&bounds[6]), __float_as_uint(v));
	}
}

// Box mode: the primitive boxes are given; only the centroid bounds and max |coordinate| are reduced.
__global__ void __launch_bounds__(kBlock) k_box_setup(const float4 *inLo, const float4 *inHi, uint32_t n, float4 *triLo,
                                                      float4 *triHi, int *bounds)
{
	uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	float4 lo = inLo[i], hi = inHi[i];
	triLo[2 * (size_t)i] = lo;
	triHi[2 * (size_t)i] = hi;
	float l[3] = { lo.x, lo.y, lo.z }, h[3] = { hi.x, hi.y, hi.z };
	float amax = 0.0f;
	for (int k = 0; k < 3; ++k)
	{
		float c = (l[k] + h[k]) * 0.5f;
		atomicMin(&bounds[k], floatToOrdered(c));
		atomicMax(&bounds[3 + k], floatToOrdered(c));
		amax = fmaxf(amax, fmaxf(fabsf(l[k]), fabsf(h[k])));
	}
	atomicMax(reinterpret_cast<unsigned int *>(&bounds[6]), __float_as_uint(amax));
}

__device__ __forceinline__ uint32_t expandBits10(uint32_t v)
{
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}

// centroid: the box centres as k_tri_setup formed them (12 bytes per triangle; the LBVH over triangles needs nothing else of
// the primitive boxes -- k_refit takes them from the position records), else they are formed here from the boxes.
__global__ void __launch_bounds__(kBlock) k_morton(const float4 *triLo, const float4 *triHi, const float *centroid, uint32_t nTris,
                                                   const int *bounds, uint32_t *codes)
{
	uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= nTris) return;
	float c[3];
	if (centroid)  // (uniform)
	{
#pragma unroll
		for (int k = 0; k < 3; ++k) c[k] = centroid[3 * (size_t)i + k];
	}
	else
	{
		float4 lo = triLo[2 * (size_t)i], hi = triHi[2 * (size_t)i];
		c[0] = (lo.x + hi.x) * 0.5f;
		c[1] = (lo.y + hi.y) * 0.5f;
		c[2] = (lo.z + hi.z) * 0.5f;
	}
	uint32_t q[3];
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		float mn = orderedToFloat(bounds[k]), mx = orderedToFloat(bounds[3 + k]);
		float ext = mx - mn;
		float n = ext > 0.0f ? (c[k] - mn) / ext : 0.0f;
		float s = n * 1024.0f;
		if (!(s > 0.0f)) s = 0.0f;
		if (s > 1023.0f) s = 1023.0f;
		q[k] = (uint32_t)s;
	}
	uint32_t code = (expandBits10(q[0]) << 2) | (expandBits10(q[1]) << 1) | expandBits10(q[2]);
	codes[i] = code;  // the first sort pass reads the codes where they lie and numbers the primitives itself
}

// ------------------------------------------------------------------- radix sort ----
// LSD, kRadixBits per pass.  A block owns one tile of kTile consecutive keys; wave w of the block owns the
// w-th quarter of the tile and walks it in rounds of 64 consecutive keys, so that (wave, round, lane) order == input
// order and the scatter is stable.
constexpr int kItems = 8;
constexpr int kTile = kBlock * kItems;
// A block sorts kSuper consecutive tiles one after the other, carrying the running digit offsets in LDS, so the histogram
// array -- and the scan over it -- has one entry per digit and SUPER tile: at 100 M keys 1.5 M counters per pass instead of
// 12.5 M (whose strided 4-byte stores cost the histogram kernel 1.6 GB of partial-line writes, profiles/r03m, and whose scan
// cost as much again as the histogram).
#ifndef PT_SORT_SUPER
#define PT_SORT_SUPER 8
#endif
constexpr int kSuper = PT_SORT_SUPER;
constexpr int kSuperTile = kTile * kSuper;
constexpr int kRadixBits = 8;   // measured at 100 M keys: 10-bit digits (3 passes) scatter 2 keys per digit and tile -- 1.85 ms a pass against 1.0
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kSortBits = 32;  // bits of the key that are sorted (30-bit Morton codes: four 8-bit passes)

__global__ void __launch_bounds__(kBlock) k_radix_hist(const uint32_t *keys, uint32_t n, int shift, uint32_t *hist,
                                                       uint32_t nBlocks)
{
	__shared__ uint32_t bins[kRadixBins];
	for (int i = threadIdx.x; i < kRadixBins; i += kBlock) bins[i] = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * kSuperTile;
#pragma unroll 8
	for (int r = 0; r < kItems * kSuper; ++r)
	{
		size_t i = base + r * kBlock + threadIdx.x;
		if (i < n) atomicAdd(&bins[(keys[i] >> shift) & (uint32_t)(kRadixBins - 1)], 1u);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < kRadixBins; i += kBlock) hist[(size_t)i * nBlocks + blockIdx.x] = bins[i];
}

// ------------------------------------------------------------------------- scan ----
// Exclusive scan of `len` counters of type T (uint32_t, or two packed 32-bit counts in a uint64_t), any length:
// tiles of 4096 (256 threads x 16 consecutive items); k_scan_reduce writes one sum per tile, the sums are scanned
// by the same routine (recursion on the host: 100 M -> 24 K -> 6), k_scan_apply rescans each tile in LDS-free wave
// arithmetic with the tile's offset.  HBM traffic 3 x sizeof(T) per item (read, read, write).
constexpr int kScanItems = 16;
constexpr int kScanTile = kBlock * kScanItems;

template <typename T> __device__ __forceinline__ T waveInclusiveScan(T v, int lane)
{
#pragma unroll
	for (int o = 1; o < PT_WAVE; o <<= 1)
	{
		T u = __shfl_up(v, o);
		if (lane >= o) v += u;
	}
	return v;
}

// Exclusive scan of one value per thread across the block (kBlock threads); total = sum over the block.
template <typename T> __device__ __forceinline__ T blockExclusiveScan(T v, T *sWave /*[kBlock / PT_WAVE]*/, T &total)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	T inc = waveInclusiveScan(v, lane);
	if (lane == PT_WAVE - 1) sWave[wave] = inc;
	__syncthreads();
	T before = 0, all = 0;
#pragma unroll
	for (int w = 0; w < kBlock / PT_WAVE; ++w)
	{
		T c = sWave[w];
		if (w < wave) before += c;
		all += c;
	}
	__syncthreads();
	total = all;
	return before + inc - v;
}

template <typename T> __global__ void __launch_bounds__(kBlock) k_scan_reduce(const T *data, size_t len, T *sums)
{
	__shared__ T sWave[kBlock / PT_WAVE];
	const size_t base = (size_t)blockIdx.x * kScanTile;
	T s = 0;
#pragma unroll
	for (int r = 0; r < kScanItems; ++r)
	{
		size_t i = base + (size_t)r * kBlock + threadIdx.x;  // coalesced; the order inside the tile does not matter here
		if (i < len) s += data[i];
	}
	T total;
	blockExclusiveScan(s, sWave, total);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// In-place exclusive scan of the tile, plus offsets[blockIdx.x] when given; totalOut (single-tile launches only)
// receives the sum of everything.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_scan_apply(T *data, size_t len, const T *offsets, T *totalOut)
{
	__shared__ T sWave[kBlock / PT_WAVE];
	const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
	T v[kScanItems];
	T s = 0;
#pragma unroll
	for (int r = 0; r < kScanItems; ++r)
	{
		v[r] = base + r < len ? data[base + r] : (T)0;
		s += v[r];
	}
	T total;
	T run = blockExclusiveScan(s, sWave, total) + (offsets ? offsets[blockIdx.x] : (T)0);
#pragma unroll
	for (int r = 0; r < kScanItems; ++r)
	{
		if (base + r < len) data[base + r] = run;
		run += v[r];
	}
	if (totalOut && threadIdx.x == 0) *totalOut = total;
}

// Number of scratch elements scanExclusive needs for `len` items.
static size_t scanScratchLen(size_t len)
{
	size_t total = 0;
	while (len > (size_t)kScanTile)
	{
		len = (len + kScanTile - 1) / kScanTile;
		total += (len + 63) & ~(size_t)63;
	}
	return total + 64;
}

template <typename T> static void scanExclusive(T *data, size_t len, T *scratch, T *totalOut, hipStream_t stream)
{
	if (len <= (size_t)kScanTile)
	{
		k_scan_apply<T><<<1, kBlock, 0, stream>>>(data, len, nullptr, totalOut);
		return;
	}
	const size_t nb = (len + kScanTile - 1) / kScanTile;
	k_scan_reduce<T><<<(uint32_t)nb, kBlock, 0, stream>>>(data, len, scratch);
	scanExclusive<T>(scratch, nb, scratch + ((nb + 63) & ~(size_t)63), totalOut, stream);
	k_scan_apply<T><<<(uint32_t)nb, kBlock, 0, stream>>>(data, len, scratch, nullptr);
}

__global__ void __launch_bounds__(kBlock) k_radix_scatter(const uint32_t *keysIn, const uint32_t *valsIn,
                                                          uint32_t *keysOut, uint32_t *valsOut, uint32_t n, int shift,
                                                          const uint32_t *histScanned, uint32_t nBlocks)
{
	constexpr int kWaves = kBlock / PT_WAVE;
	constexpr uint32_t kMask = (uint32_t)(kRadixBins - 1);
	__shared__ uint32_t waveCount[kWaves][kRadixBins];
	__shared__ uint32_t digitBase[kRadixBins], tileStart[kRadixBins], sScan[kBlock / PT_WAVE];
	__shared__ uint32_t sKey[kTile], sVal[kTile];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const unsigned long long lanesBelow = (1ull << lane) - 1ull;
	static_assert(kRadixBins == kBlock, "one digit per thread below");
	digitBase[threadIdx.x] = histScanned[(size_t)threadIdx.x * nBlocks + blockIdx.x];  // first output position of digit d of this super tile
	for (int sub = 0; sub < kSuper; ++sub)
	{
	const size_t tileBase = (size_t)blockIdx.x * kSuperTile + (size_t)sub * kTile;
	if (tileBase >= n) break;  // (uniform)
	for (int i = threadIdx.x; i < kWaves * kRadixBins; i += kBlock) (&waveCount[0][0])[i] = 0;
	__syncthreads();

	uint32_t key[kItems], val[kItems], rank[kItems];
	const size_t waveBase = tileBase + (size_t)wave * (kItems * PT_WAVE);
#pragma unroll
	for (int r = 0; r < kItems; ++r)
	{
		size_t i = waveBase + r * PT_WAVE + lane;
		bool valid = i < n;
		key[r] = valid ? keysIn[i] : 0xFFFFFFFFu;
		val[r] = valid ? (valsIn ? valsIn[i] : (uint32_t)i) : 0u;  // valsIn == nullptr: the first pass, value = position
		uint32_t digit = (key[r] >> shift) & kMask;
		// lanes of this round holding the same digit (and the same validity)
		unsigned long long peers = __ballot(valid);
		if (!valid) peers = ~peers;
#pragma unroll
		for (int b = 0; b < kRadixBits; ++b)
		{
			unsigned long long m = __ballot((digit >> b) & 1u);
			peers &= ((digit >> b) & 1u) ? m : ~m;
		}
		uint32_t before = waveCount[wave][digit];
		rank[r] = before + __popcll(peers & lanesBelow);
		// the lowest peer lane publishes the new running count for this digit
		if (valid && (peers & lanesBelow) == 0) waveCount[wave][digit] = before + __popcll(peers);
	}
	__syncthreads();
	// per digit: exclusive scan across the waves of the block, and the digit's total in this tile
	uint32_t digitTotal;
	{
		const int d = threadIdx.x;
		uint32_t run = 0;
		for (int w = 0; w < kWaves; ++w)
		{
			uint32_t c = waveCount[w][d];
			waveCount[w][d] = run;
			run += c;
		}
		digitTotal = run;
	}
	// Position of every key inside the tile once the tile is sorted by digit (stable): the keys go through LDS in that
	// order, so that consecutive lanes then store consecutive addresses of one digit's run -- a wave store touches ~9 runs
	// instead of 64 single words (measured at 100 M keys: 1.0 ms -> see DESIGN.md 5.1 per pass).
	uint32_t tileTotal;
	const uint32_t start = blockExclusiveScan(digitTotal, sScan, tileTotal);
	tileStart[threadIdx.x] = start;
	__syncthreads();
#pragma unroll
	for (int r = 0; r < kItems; ++r)
	{
		size_t i = waveBase + r * PT_WAVE + lane;
		if (i < n)
		{
			uint32_t digit = (key[r] >> shift) & kMask;
			uint32_t pos = tileStart[digit] + waveCount[wave][digit] + rank[r];
			sKey[pos] = key[r];
			sVal[pos] = val[r];
		}
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < kItems; ++r)
	{
		const uint32_t pos = r * kBlock + threadIdx.x;
		if (pos < tileTotal)
		{
			const uint32_t k = sKey[pos], digit = (k >> shift) & kMask;
			const uint32_t dst = digitBase[digit] + (pos - tileStart[digit]);
			keysOut[dst] = k;
			valsOut[dst] = sVal[pos];
		}
	}
	__syncthreads();  // every thread has read digitBase and the staging area
	digitBase[threadIdx.x] += digitTotal;  // the next tile of this block continues behind this one's keys, digit by digit
	}
}

// ----------------------------------------------------------------- LBVH topology ----
// BVH2 node numbering: internal nodes 0 .. n-2, leaf for sorted position k is (n-1) + k.
// delta(i, j) of Karras 2012 over the sorted keys: common leading bits, the index as a tie-break; -1 outside the array.
// `at` reads a key by position.
template <typename KeyAt> __device__ __forceinline__ int lbvhDelta(const KeyAt &at, int n, uint32_t a, int i, int j)
{
	if (j < 0 || j >= n) return -1;
	const uint32_t b = at(j);
	if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
	return __clz(a ^ b);
}

// One 16-byte record per internal node: (left child, right child, parent, other end of the node's range of sorted positions).
// A Karras node i sits at one end of its range, so range = [min(i, j), max(i, j)] and the subtree holds |j - i| + 1 leaves:
// the refit walk reads ONE record per level (children, range for the LDS test, next parent) where three arrays cost it
// three dependent loads.  The parent word of record c is written by c's parent's thread, the other three by c's own:
// separate stores to separate words, no conflict.  leafParent[k] = parent of the leaf at sorted position k.
// The searches of a node read keys near its own position -- a node's range is short for all but the top of the tree -- so the
// block keeps the keys of its 256 positions and of 256 either side in LDS (3 KB) and only the long ranges go to memory: the
// kernel was a chain of dependent loads (2.5 ms at 100 M keys, 1.3 TB/s of traffic).
__global__ void __launch_bounds__(kBlock) k_lbvh_topology(const uint32_t *keys, int n, int4 *topo, int *leafParent, unsigned int *arrive)
{
	__shared__ uint32_t sKeys[3 * kBlock];
	const int winFirst = (int)(blockIdx.x * kBlock) - kBlock;  // position of sKeys[0]
	for (int k = threadIdx.x; k < 3 * kBlock; k += kBlock)
	{
		const int pos = winFirst + k;
		sKeys[k] = (pos >= 0 && pos < n) ? keys[pos] : 0u;
	}
	__syncthreads();
	int i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n - 1) return;
	arrive[i] = 0u;  // the refit's arrival counter of this node (instead of a 4n-byte fill of its own)
	auto at = [&](int j) -> uint32_t {
		const int k = j - winFirst;
		return (k >= 0 && k < 3 * kBlock) ? sKeys[k] : keys[j];
	};
	const uint32_t a = sKeys[kBlock + threadIdx.x];
	int d = (lbvhDelta(at, n, a, i, i + 1) - lbvhDelta(at, n, a, i, i - 1)) >= 0 ? 1 : -1;
	int dmin = lbvhDelta(at, n, a, i, i - d);
	int lmax = 2;
	while (lbvhDelta(at, n, a, i, i + lmax * d) > dmin) lmax <<= 1;
	int l = 0;
	for (int t = lmax >> 1; t >= 1; t >>= 1)
		if (lbvhDelta(at, n, a, i, i + (l + t) * d) > dmin) l += t;
	int j = i + l * d;
	int dnode = lbvhDelta(at, n, a, i, j);
	int s = 0;
	int t = l;
	do
	{
		t = (t + 1) >> 1;
		if (lbvhDelta(at, n, a, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	int gamma = i + s * d + (d < 0 ? d : 0);
	int first = i < j ? i : j, last = i < j ? j : i;
	int left = (first == gamma) ? (n - 1) + gamma : gamma;
	int right = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
	int *rec = reinterpret_cast<int *>(topo + i);
	*reinterpret_cast<int2 *>(rec) = make_int2(left, right);
	rec[3] = j;
	if (left >= n - 1) leafParent[left - (n - 1)] = i;
	else reinterpret_cast<int *>(topo + left)[2] = i;
	if (right >= n - 1) leafParent[right - (n - 1)] = i;
	else reinterpret_cast<int *>(topo + right)[2] = i;
	if (i == 0) rec[2] = -1;
}

// The BVH2 as the collapse sees it: LBVH = the topology records (children, leaf count and first sorted position all come
// out of one record), PLOC = children + subtree leaf counts in two arrays.  Leaf for sorted position k = node (n-1) + k.
struct Bvh2View
{
	const int4 *topo;          // LBVH, else nullptr
	const int2 *children;      // PLOC
	const uint32_t *subCount;  // PLOC
	int n;
	__device__ __forceinline__ int2 kids(int nd) const
	{
		return topo ? *reinterpret_cast<const int2 *>(topo + nd) : children[nd];
	}
	__device__ __forceinline__ uint32_t count(int nd) const
	{
		if (!topo) return subCount[nd];
		if (nd >= n - 1) return 1u;
		const int j = reinterpret_cast<const int *>(topo + nd)[3];
		return (uint32_t)(j > nd ? j - nd : nd - j) + 1u;
	}
	// children and leaf count of node nd with ONE load where the layout allows (LBVH: both come out of the 16-byte record)
	__device__ __forceinline__ void load(int nd, int2 &k, uint32_t &c) const
	{
		k = make_int2(0, 0);
		if (nd >= n - 1)
		{
			c = topo ? 1u : subCount[nd];
			return;
		}
		if (topo)
		{
			const int4 r = topo[nd];
			k = make_int2(r.x, r.y);
			c = (uint32_t)(r.w > nd ? r.w - nd : nd - r.w) + 1u;
		}
		else
		{
			c = subCount[nd];
			k = children[nd];
		}
	}
	// LBVH only: first sorted position of the subtree
	__device__ __forceinline__ uint32_t first(int nd) const
	{
		if (nd >= n - 1) return (uint32_t)(nd - (n - 1));
		const int j = reinterpret_cast<const int *>(topo + nd)[3];
		return (uint32_t)(j < nd ? j : nd);
	}
};

// ------------------------------------------------------------------------- refit ----
// boxes: 2 float4 per BVH2 node: (lo.xyz, -) (hi.xyz, -).  Boxes cross CUs inside this launch,
// so every box store/load is an agent-scope (sc1) access and the arrival counter is an
// agent-scope atomic issued after the storing lane's vmcnt has drained.
__device__ __forceinline__ void storeBox(float *dst, const float lo[3], const float hi[3])
{
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		__hip_atomic_store(dst + k, lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__hip_atomic_store(dst + 4 + k, hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
__device__ __forceinline__ void loadBox(const float *src, float lo[3], float hi[3])
{
#pragma unroll
	for (int k = 0; k < 3; ++k)
	{
		lo[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		hi[k] = __hip_atomic_load(src + 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// One thread per leaf walks towards the root; at every internal node the first arrival stops and the second one
// (which then knows both children are finished) forms the union.  A thread carries its own box in registers and
// reads only the sibling's.  Two protocols:
//   * LOCAL: while the parent's range of sorted positions lies inside this block's 256 leaves, its index does too
//     (a Karras node sits at one end of its range), so arrival counters and sibling boxes live in LDS -- 99 % of the
//     nodes, no global atomics, no cross-CU traffic;
//   * GLOBAL (the top of the tree): the box is published with agent-scope stores, the lane drains its stores
//     (s_waitcnt vmcnt(0)), then an agent-scope atomic counts the arrival and the sibling's box is read with
//     agent-scope loads (MI355X_MICROARCH.md, cross-CU hand-off inside a launch).
// Every node's box is also written to `boxes` for the collapse kernel.
// The leaf's own box: triangles -- from the gathered 48-byte position record triPos[prim], the SAME gather the leaf-ordered
// triangle array needs (LBVH: leaf order is the sorted order), so the thread writes its three rotated PtTri48 records here
// too and the primitive boxes are never gathered; boxes (top level) -- from triLo/triHi as given.
__global__ void __launch_bounds__(kBlock) k_refit(const float4 *triPos, const float4 *triLo, const float4 *triHi,
                                                  const uint32_t *sortedPrim, int n, const int4 *topo, const int *leafParent,
                                                  const int *bounds, float *boxes, unsigned int *arrive, float padScale,
                                                  PtTri48 *tris, int planes)
{
	__shared__ float sBox[6][2 * kBlock];  // slot: internal node i -> i - base, leaf k -> kBlock + k - base
	__shared__ unsigned int sArrive[kBlock];
	const int base = blockIdx.x * kBlock, k = base + threadIdx.x;
	sArrive[threadIdx.x] = 0;
	__syncthreads();
	const bool valid = k < n;
	const float pad = __uint_as_float((unsigned int)bounds[6]) * padScale;  // 2^-18 * max|coord| for triangles
	const uint32_t prim = valid ? sortedPrim[k] : 0u;
	int cur = (n - 1) + k;
	int p = valid ? leafParent[k] : -1;
	float lo[3], hi[3];
	float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
	if (triPos)  // (uniform)
	{
		if (valid)
		{
			r0 = triPos[3 * (size_t)prim + 0];
			r1 = triPos[3 * (size_t)prim + 1];
			r2 = triPos[3 * (size_t)prim + 2];
		}
		const float q[3][3] = { { r0.x, r0.y, r0.z }, { r0.w, r1.x, r1.y }, { r1.z, r1.w, r2.x } };
		// The block's 256 records of a plane are 12 KB contiguous: staged through LDS (the box slots, not in use yet) so that
		// every store instruction of a wave writes 1 KB of consecutive bytes -- written straight from the lanes, 16 bytes at
		// a 48-byte stride, a store instruction touches 24 lines a third at a time.
		float4 *stage = reinterpret_cast<float4 *>(&sBox[0][0]);
		static_assert(sizeof(sBox) >= (size_t)kBlock * sizeof(PtTri48), "staging area");
		const int nRec = n - base < kBlock ? n - base : kBlock;
#pragma unroll
		for (int r = 0; r < 3; ++r)
		{
			if (planes == 1 && r != 2) continue;  // (uniform) one plane: the identity
			const int x = (r + 1) % 3, y = (r + 2) % 3, z = r;
			stage[3 * threadIdx.x + 0] = make_float4(q[0][x], q[0][y], q[0][z], q[1][x]);
			stage[3 * threadIdx.x + 1] = make_float4(q[1][y], q[1][z], q[2][x], q[2][y]);
			stage[3 * threadIdx.x + 2] = make_float4(q[2][z], __uint_as_float(prim), 0.0f, 0.0f);
			__syncthreads();
			float4 *dst = reinterpret_cast<float4 *>(tris + (planes == 1 ? (size_t)0 : (size_t)r * n) + base);
			for (int i = threadIdx.x; i < 3 * nRec; i += kBlock) dst[i] = stage[i];
			__syncthreads();
		}
	}
	if (!valid) return;
	if (triPos)
	{
		const float q[3][3] = { { r0.x, r0.y, r0.z }, { r0.w, r1.x, r1.y }, { r1.z, r1.w, r2.x } };
		// exactly k_tri_setup's box of the primitive (same compares in the same order)
#pragma unroll
		for (int a = 0; a < 3; ++a)
		{
			float l = INFINITY, h = -INFINITY;
#pragma unroll
			for (int j = 0; j < 3; ++j)
			{
				const float c = q[j][a];
				if (c < l) l = c;
				if (c > h) h = c;
			}
			lo[a] = l - pad;
			hi[a] = h + pad;
		}
	}
	else
	{
		const float4 l4 = triLo[2 * (size_t)prim], h4 = triHi[2 * (size_t)prim];
		lo[0] = l4.x - pad; lo[1] = l4.y - pad; lo[2] = l4.z - pad;
		hi[0] = h4.x + pad; hi[1] = h4.y + pad; hi[2] = h4.z + pad;
	}
	bool local = true;
	bool skipStore = false;
	{
		// A triangle leaf whose parent lies inside this block hands its box over in LDS, and the collapse recomputes the box
		// of a single-triangle leaf child from the triangle record (k_collapse_level): no 32-byte store for 99 % of the leaves.
		if (triPos && p >= 0)
		{
			const int j = reinterpret_cast<const int *>(topo + p)[3];
			const int first = j < p ? j : p, last = j < p ? p : j;
			skipStore = first >= base && last < base + kBlock;
		}
	}
	for (;;)
	{
		// Every lane that arrives, first or second, stores the node's box here: consecutive lanes write consecutive 32-byte
		// boxes, whole lines at a time.  (Measured at 100 M triangles, this kernel 12.8 ms: the same stores split between the
		// two arrivals 31 ms; the record requested a level ahead and LDS-only fences, so that no wait covers a store, 22 ms --
		// the full waits pace the stores, and the memory system does better paced.)
		if (!skipStore)
		{
			float4 *dst = reinterpret_cast<float4 *>(boxes + 8 * (size_t)cur);
			dst[0] = make_float4(lo[0], lo[1], lo[2], 0.0f);
			dst[1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
		}
		skipStore = false;
		if (p < 0) return;  // cur is the root
		const int4 t = topo[p];  // children, parent of p, other end of p's range: one load per level
		if (local)
		{
			const int first = t.w < p ? t.w : p, last = t.w < p ? p : t.w;
			local = first >= base && last < base + kBlock;
		}
		const int sib = t.x == cur ? t.y : t.x;
		if (local)
		{
			const int slot = cur >= n - 1 ? kBlock + (cur - (n - 1) - base) : cur - base;
#pragma unroll
			for (int a = 0; a < 3; ++a)
			{
				sBox[a][slot] = lo[a];
				sBox[3 + a][slot] = hi[a];
			}
			__threadfence_block();
			if (atomicAdd(&sArrive[p - base], 1u) == 0u) return;  // the sibling subtree finishes this node
			__threadfence_block();
			const int ss = sib >= n - 1 ? kBlock + (sib - (n - 1) - base) : sib - base;
#pragma unroll
			for (int a = 0; a < 3; ++a)
			{
				lo[a] = fminf(lo[a], sBox[a][ss]);
				hi[a] = fmaxf(hi[a], sBox[3 + a][ss]);
			}
		}
		else
		{
			storeBox(boxes + 8 * (size_t)cur, lo, hi);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			unsigned int old = __hip_atomic_fetch_add(&arrive[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (old == 0) return;
			float lo2[3], hi2[3];
			loadBox(boxes + 8 * (size_t)sib, lo2, hi2);
#pragma unroll
			for (int a = 0; a < 3; ++a)
			{
				lo[a] = fminf(lo[a], lo2[a]);
				hi[a] = fmaxf(hi[a], hi2[a]);
			}
		}
		cur = p;
		p = t.z;
	}
}

// ---------------------------------------------------------------------- collapse ----
struct CollapseTask
{
	int bvh2;     // BVH2 node this wide node is rooted at
	uint32_t out; // wide-node slot to write
};

__device__ __forceinline__ float halfArea(const float *box)
{
	float dx = box[4] - box[0], dy = box[5] - box[1], dz = box[6] - box[2];
	return (dx * dy + dy * dz) + dz * dx;
}

// One thread per wide node of the level.  Children that stay inner nodes become the tasks of the next level; their
// node ids are childBase + position in that queue (breadth-first numbering), so ONE block-aggregated atomic hands
// out both (a per-child atomic on a single counter serialises: 4 ms of a 20 M-triangle build).
// The level loop runs on the device: levelCount[L] = wide nodes of level L (levelCount[0] = 1, the root), written by
// level L - 1's launch; the host enqueues kMaxLevels launches back to back and reads the counts once at the end (a
// read-back per level cost ~0.1 ms each, 16 of them at 100 M triangles).  Empty levels return at once.
constexpr int kMaxLevels = 32;  // trees deeper than the traversal stack allows (28 levels) are rejected by the caller anyway

__global__ void __launch_bounds__(kBlock)
    k_collapse_level(const CollapseTask *tasksIn, CollapseTask *tasksOut, uint32_t *levelCount, int level, Bvh2View bvh,
                     const float *boxes, const uint32_t *sortedPrim, bool boxMode, uint32_t *leafPrim, PtNode4 *nodes,
                     uint32_t *triCount, int leafMax, uint32_t nodeBase, const PtTri48 *leafTris, const int *bounds, float padScale)
{
	const float leafPad = leafTris ? __uint_as_float((unsigned int)bounds[6]) * padScale : 0.0f;  // as k_refit
	__shared__ unsigned long long sWave[kBlock / PT_WAVE];
	__shared__ uint32_t sBase[2];
	const int n = bvh.n;
	const uint32_t nIn = levelCount[level];
	if (nIn == 0) return;
	uint32_t childBase = 0;  // wide nodes of the levels 0 .. level = id of the first node of the next level
	for (int l = 0; l <= level; ++l) childBase += levelCount[l];
	uint32_t *nOut = levelCount + level + 1;
	for (uint32_t t0 = blockIdx.x * kBlock; t0 < nIn; t0 += gridDim.x * kBlock)
	{
	const uint32_t ti = t0 + threadIdx.x;
	const bool active = ti < nIn;
	CollapseTask task = { 0, 0u };
	int slot[4] = { 0, 0, 0, 0 };
	uint32_t cnt[4] = { 0, 0, 0, 0 };
	int nSlots = 0;
	if (active)
	{
		task = tasksIn[ti];
		uint32_t rootCount;
		int2 ch;
		bvh.load(task.bvh2, ch, rootCount);
		if ((int)rootCount <= leafMax)
		{
			slot[0] = task.bvh2;  // whole subtree is one leaf child
			cnt[0] = rootCount;
			nSlots = 1;
		}
		else
		{
			// slot[], cnt[], area[] are only ever indexed by unrolled loop counters (a run-time index would put them in
			// scratch memory): "write entry s" is a select over the four entries
			float area[4] = { -1.0f, -1.0f, -1.0f, -1.0f };
			int2 kid[4] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } };  // children of every slot: opening one costs no further load
			auto put = [&](int s, int nd) {
				uint32_t c;
				int2 kk;
				bvh.load(nd, kk, c);
				const float ar = (int)c <= leafMax ? -1.0f : halfArea(boxes + 8 * (size_t)nd);  // leaf children are never opened
#pragma unroll
				for (int k = 0; k < 4; ++k)
					if (k == s)
					{
						slot[k] = nd;
						cnt[k] = c;
						area[k] = ar;
						kid[k] = kk;
					}
			};
			put(0, ch.x);
			put(1, ch.y);
			nSlots = 2;
#pragma unroll
			for (int round = 0; round < 2; ++round)
			{
				int best = -1;
				int2 ch2 = make_int2(0, 0);
				float bestArea = -1.0f;
#pragma unroll
				for (int k = 0; k < 4; ++k)
					if (k < nSlots && area[k] > bestArea)
					{
						bestArea = area[k];
						best = k;
						ch2 = kid[k];
					}
				if (best >= 0)  // otherwise every child is a leaf: nothing to open, in this round or the next
				{
					put(best, ch2.x);
					put(nSlots++, ch2.y);
				}
			}
		}
	}
	// queue positions / node ids of the inner children and (PLOC) triangle slots of the leaf children
	uint32_t nInner = 0, nLeafTris = 0;
#pragma unroll
	for (int s = 0; s < 4; ++s)
		if (s < nSlots)
		{
			if ((int)cnt[s] <= leafMax) nLeafTris += cnt[s];
			else ++nInner;
		}
	unsigned long long total;
	const unsigned long long ex = blockExclusiveScan((unsigned long long)nInner | ((unsigned long long)nLeafTris << 32), sWave, total);
	if (threadIdx.x == 0)
	{
		sBase[0] = (uint32_t)total ? atomicAdd(nOut, (uint32_t)total) : 0u;
		sBase[1] = (leafPrim && (uint32_t)(total >> 32)) ? atomicAdd(triCount, (uint32_t)(total >> 32)) : 0u;
	}
	__syncthreads();
	if (!active) continue;
	uint32_t q = sBase[0] + (uint32_t)ex, triBase = sBase[1] + (uint32_t)(ex >> 32);

	// node frame: origin = min corner, power-of-two scale per axis covering the extent in 255 steps
	float4 blo[4], bhi[4];
	float org[3] = { INFINITY, INFINITY, INFINITY }, top[3] = { -INFINITY, -INFINITY, -INFINITY };
#pragma unroll
	for (int s = 0; s < 4; ++s)
		if (s < nSlots)
		{
			if (leafTris && slot[s] >= n - 1)
			{
				// LBVH, a child that is ONE triangle: k_refit does not store such a leaf's box (it travels through LDS); it is
				// recomputed here from the leaf-ordered record (plane 2 = x, y, z order; sorted position = leaf slot), with
				// k_refit's compares and padding
				const PtTri48 t = leafTris[slot[s] - (n - 1)];
				float l3[3], h3[3];
#pragma unroll
				for (int a = 0; a < 3; ++a)
				{
					float l = INFINITY, h = -INFINITY;
					const float cs[3] = { t.v0[a], t.v1[a], t.v2[a] };
#pragma unroll
					for (int j = 0; j < 3; ++j)
					{
						if (cs[j] < l) l = cs[j];
						if (cs[j] > h) h = cs[j];
					}
					l3[a] = l - leafPad;
					h3[a] = h + leafPad;
				}
				blo[s] = make_float4(l3[0], l3[1], l3[2], 0.0f);
				bhi[s] = make_float4(h3[0], h3[1], h3[2], 0.0f);
			}
			else
			{
				const float4 *b = reinterpret_cast<const float4 *>(boxes + 8 * (size_t)slot[s]);
				blo[s] = b[0];
				bhi[s] = b[1];
			}
			org[0] = fminf(org[0], blo[s].x);
			org[1] = fminf(org[1], blo[s].y);
			org[2] = fminf(org[2], blo[s].z);
			top[0] = fmaxf(top[0], bhi[s].x);
			top[1] = fmaxf(top[1], bhi[s].y);
			top[2] = fmaxf(top[2], bhi[s].z);
		}
	float scale[3];
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		float ext = top[a] - org[a];
		int e = 1;
		if (ext > 0.0f)
		{
			int ex2;
			float m = frexpf(ext / 255.0f, &ex2);  // ext/255 = m * 2^ex2, m in [0.5,1)
			e = (m == 0.5f ? ex2 - 1 : ex2) + 127;
		}
		if (e < 1) e = 1;
		if (e > 254) e = 254;
		float sc = __uint_as_float((uint32_t)e << 23);
		while (e < 254 && !(__builtin_fmaf(255.0f, sc, org[a]) >= top[a]))
		{
			++e;
			sc = __uint_as_float((uint32_t)e << 23);
		}
		scale[a] = sc;
	}

	PtNode4 node;
	node.origin[0] = org[0];
	node.origin[1] = org[1];
	node.origin[2] = org[2];
	node.scaleX = scale[0];
	node.scaleY = scale[1];
	node.scaleZ = scale[2];
#pragma unroll
	for (int a = 0; a < 3; ++a)
	{
		node.qlo[a] = 0xFFFFFFFFu;  // empty slots: lo = 255, hi = 0
		node.qhi[a] = 0u;
	}
#pragma unroll
	for (int c = 0; c < 4; ++c) node.child[c] = PT_EMPTY_CHILD;

#pragma unroll
	for (int s = 0; s < 4; ++s)
	{
		if (s >= nSlots) continue;
		const float lo3[3] = { blo[s].x, blo[s].y, blo[s].z }, hi3[3] = { bhi[s].x, bhi[s].y, bhi[s].z };
#pragma unroll
		for (int a = 0; a < 3; ++a)
		{
			float ql = floorf((lo3[a] - org[a]) / scale[a]);
			ql = fminf(fmaxf(ql, 0.0f), 255.0f);
			while (ql > 0.0f && __builtin_fmaf(ql, scale[a], org[a]) > lo3[a]) ql -= 1.0f;
			float qh = ceilf((hi3[a] - org[a]) / scale[a]);
			qh = fminf(fmaxf(qh, 0.0f), 255.0f);
			while (qh < 255.0f && __builtin_fmaf(qh, scale[a], org[a]) < hi3[a]) qh += 1.0f;
			node.qlo[a] = (node.qlo[a] & ~(0xFFu << (8 * s))) | ((uint32_t)ql << (8 * s));
			node.qhi[a] = (node.qhi[a] & ~(0xFFu << (8 * s))) | ((uint32_t)qh << (8 * s));
		}
		const int c = (int)cnt[s];
		if (c <= leafMax)
		{
			if (!leafPrim)
			{
				// Box mode (leafMax == 1): the leaf reference is the primitive (instance) index itself.
				// LBVH: a subtree covers consecutive sorted positions and the triangle array is in sorted order,
				// so the leaf is (first position, count) -- nothing to gather, nothing to allocate.
				const uint32_t first = bvh.first(slot[s]);
				node.child[s] = boxMode ? (PT_LEAF_FLAG | sortedPrim[first]) : (PT_LEAF_FLAG | ((uint32_t)(c - 1) << 28) | first);
				continue;
			}
			// PLOC: the primitives of the subtree, left to right, into consecutive slots
			int np = 0, dfs[PT_LEAF_MAX + 1], dsp = 0;
			dfs[dsp++] = slot[s];
			while (dsp > 0 && np < c)
			{
				int nd = dfs[--dsp];
				if (nd >= n - 1) leafPrim[triBase + np++] = sortedPrim[nd - (n - 1)];
				else
				{
					int2 ch = bvh.kids(nd);
					dfs[dsp++] = ch.y;
					dfs[dsp++] = ch.x;
				}
			}
			node.child[s] = PT_LEAF_FLAG | ((uint32_t)(c - 1) << 28) | triBase;
			triBase += (uint32_t)c;
		}
		else
		{
			node.child[s] = childBase + q + nodeBase;
			tasksOut[q].bvh2 = slot[s];
			tasksOut[q].out = childBase + q;
			++q;
		}
	}
	nodes[task.out] = node;
	}
}


// Leaf-ordered triangle records: slot j holds primitive leafPrim[j]; plane r holds the coordinates in the order
// ((r+1)%3, (r+2)%3, r) so that the watertight test of a ray whose dominant axis is r reads them without per-lane
// selects (plane 2 is the identity).  One thread per slot: three 16-byte stores per plane, consecutive lanes write
// consecutive 48-byte records.
__global__ void __launch_bounds__(kBlock) k_write_tris(const uint32_t *leafPrim, uint32_t n, const float4 *triPos,
                                                       PtTri48 *tris, int planes)
{
	uint32_t j = blockIdx.x * kBlock + threadIdx.x;
	if (j >= n) return;
	const uint32_t prim = leafPrim[j];
	const float4 r0 = triPos[3 * (size_t)prim + 0], r1 = triPos[3 * (size_t)prim + 1], r2 = triPos[3 * (size_t)prim + 2];
	const float a[3] = { r0.x, r0.y, r0.z }, b[3] = { r0.w, r1.x, r1.y }, c[3] = { r1.z, r1.w, r2.x };
#pragma unroll
	for (int r = 0; r < 3; ++r)
	{
		if (planes == 1 && r != 2) continue;  // one plane: the identity
		const int x = (r + 1) % 3, y = (r + 2) % 3, z = r;
		float4 *dst = reinterpret_cast<float4 *>(tris + (planes == 1 ? (size_t)0 : (size_t)r * n) + j);
		dst[0] = make_float4(a[x], a[y], a[z], b[x]);
		dst[1] = make_float4(b[y], b[z], c[x], c[y]);
		dst[2] = make_float4(c[z], __uint_as_float(prim), 0.0f, 0.0f);
	}
}

// ----------------------------------------------------------------------------- PLOC ----
// Parallel locally-ordered clustering (Meister & Bittner 2017) over the Morton-sorted leaves: every
// cluster looks kPlocRadius positions to either side for the neighbour whose merged box has the least
// surface area; mutual nearest neighbours merge; the survivors are compacted in order; repeat until
// one cluster is left.  BVH2 node ids: leaf for sorted position k = (n-1)+k, internal nodes are numbered
// 0.. in creation order (deterministic: exclusive scan of the merge flags), the root is the last one.
#ifndef PT_PLOC_RADIUS
#define PT_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = PT_PLOC_RADIUS;

__global__ void __launch_bounds__(kBlock) k_ploc_leaves(const float4 *triLo, const float4 *triHi,
                                                        const uint32_t *sortedPrim, int n, const int *bounds,
                                                        float *boxes, uint32_t *subCount, uint32_t *cluster,
                                                        float padScale)
{
	int k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= n) return;
	const float pad = __uint_as_float((unsigned int)bounds[6]) * padScale;  // as k_refit
	uint32_t prim = sortedPrim[k];
	float4 l4 = triLo[2 * (size_t)prim], h4 = triHi[2 * (size_t)prim];
	float *b = boxes + 8 * (size_t)((n - 1) + k);
	b[0] = l4.x - pad; b[1] = l4.y - pad; b[2] = l4.z - pad; b[3] = 0.0f;
	b[4] = h4.x + pad; b[5] = h4.y + pad; b[6] = h4.z + pad; b[7] = 0.0f;
	subCount[(n - 1) + k] = 1u;
	cluster[k] = (uint32_t)((n - 1) + k);
}

__global__ void __launch_bounds__(kBlock) k_ploc_nearest(const uint32_t *cluster, uint32_t m, const float *boxes,
                                                         uint32_t *nearest)
{
	// the block's window of cluster boxes (256 + 16 either side) is gathered once into LDS; every thread then scans
	// its 32 neighbours there instead of issuing 32 dependent global gathers
	constexpr int kWin = kBlock + 2 * kPlocRadius;
	__shared__ float sB[6][kWin];
	const long long base = (long long)blockIdx.x * kBlock - kPlocRadius;
	for (int t = threadIdx.x; t < kWin; t += kBlock)
	{
		const long long g = base + t;
		if (g >= 0 && g < (long long)m)
		{
			const float4 *b = reinterpret_cast<const float4 *>(boxes + 8 * (size_t)cluster[g]);
			const float4 l = b[0], h = b[1];
			sB[0][t] = l.x; sB[1][t] = l.y; sB[2][t] = l.z;
			sB[3][t] = h.x; sB[4][t] = h.y; sB[5][t] = h.z;
		}
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= m) return;
	const int me = threadIdx.x + kPlocRadius;
	const float lo[3] = { sB[0][me], sB[1][me], sB[2][me] }, hi[3] = { sB[3][me], sB[4][me], sB[5][me] };
	const uint32_t j0 = i > (uint32_t)kPlocRadius ? i - kPlocRadius : 0u;
	const uint32_t j1 = i + kPlocRadius < m - 1 ? i + kPlocRadius : m - 1;
	float best = INFINITY;
	uint32_t bestJ = i;
	for (uint32_t j = j0; j <= j1; ++j)
	{
		if (j == i) continue;
		const int t = (int)((long long)j - base);
		float dx = fmaxf(hi[0], sB[3][t]) - fminf(lo[0], sB[0][t]);
		float dy = fmaxf(hi[1], sB[4][t]) - fminf(lo[1], sB[1][t]);
		float dz = fmaxf(hi[2], sB[5][t]) - fminf(lo[2], sB[2][t]);
		float a = (dx * dy + dy * dz) + dz * dx;  // symmetric in (i, j): both sides compute the same value
		if (a < best)
		{
			best = a;
			bestJ = j;
		}
	}
	nearest[i] = bestJ;
}

// Flags of cluster i (recomputed from `nearest` wherever they are needed): keep = survives at its position (alone,
// or as the lower index of a merged pair); merge = lower index of a mutual pair, creates an internal node.
// Packed as keep | merge << 32 so that one 64-bit scan carries both counts.
__device__ __forceinline__ unsigned long long plocFlags(const uint32_t *nearest, uint32_t i, uint32_t &j, bool &mutual)
{
	j = nearest[i];
	mutual = j != i && nearest[j] == i;
	const unsigned long long keep = (mutual && j < i) ? 0ull : 1ull, merge = (mutual && i < j) ? 1ull : 0ull;
	return keep | (merge << 32);
}

// One (keep, merge) sum per block of kBlock clusters.
__global__ void __launch_bounds__(kBlock) k_ploc_count(const uint32_t *nearest, uint32_t m, unsigned long long *sums)
{
	__shared__ unsigned long long sWave[kBlock / PT_WAVE];
	uint32_t i = blockIdx.x * kBlock + threadIdx.x, j;
	bool mutual;
	unsigned long long f = i < m ? plocFlags(nearest, i, j, mutual) : 0ull, total;
	blockExclusiveScan(f, sWave, total);
	if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// sums: exclusive scan of k_ploc_count's output.  Compacts the survivors in order and creates the internal nodes
// (numbered internalBase + exclusive scan of the merge flags: deterministic).
__global__ void __launch_bounds__(kBlock)
    k_ploc_apply(const uint32_t *cluster, const uint32_t *nearest, uint32_t m, const unsigned long long *sums,
                 uint32_t internalBase, float *boxes, uint32_t *subCount, int2 *children, uint32_t *clusterOut)
{
	__shared__ unsigned long long sWave[kBlock / PT_WAVE];
	uint32_t i = blockIdx.x * kBlock + threadIdx.x, j = 0;
	bool mutual = false;
	unsigned long long f = i < m ? plocFlags(nearest, i, j, mutual) : 0ull, total;
	unsigned long long ex = blockExclusiveScan(f, sWave, total) + sums[blockIdx.x];
	if (i >= m || (mutual && j < i)) return;  // absorbed by its partner
	uint32_t id = cluster[i];
	if (mutual)
	{
		uint32_t a = id, b = cluster[j];
		id = internalBase + (uint32_t)(ex >> 32);
		const float4 *ba = reinterpret_cast<const float4 *>(boxes + 8 * (size_t)a);
		const float4 *bb = reinterpret_cast<const float4 *>(boxes + 8 * (size_t)b);
		float4 *bo = reinterpret_cast<float4 *>(boxes + 8 * (size_t)id);
		float4 alo = ba[0], ahi = ba[1], blo = bb[0], bhi = bb[1];
		bo[0] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.0f);
		bo[1] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.0f);
		subCount[id] = subCount[a] + subCount[b];
		children[id] = make_int2((int)a, (int)b);
	}
	clusterOut[(uint32_t)ex] = id;
}

__global__ void k_init_bounds(int *bounds)
{
	if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7FFFFFFF;
	else if (threadIdx.x < 6) bounds[threadIdx.x] = (int)0x80000000;
	else if (threadIdx.x == 6) bounds[6] = 0;
}
}  // namespace

#define HIPCHK(x)                    \
	do                               \
	{                                \
		hipError_t e_ = (x);         \
		if (e_ != hipSuccess) return e_; \
	} while (0)

static inline uint32_t divUp(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

hipError_t ptBuildBvh4(const PtBuildInput &in, PtBuildOutput &out, hipStream_t stream)
{
	const uint32_t n = in.nTris;
	const bool boxMode = in.boxLo != nullptr;
	const int leafMax = boxMode ? 1 : (in.leafMax < 1 ? PT_LEAF_DEFAULT : (in.leafMax > PT_LEAF_MAX ? PT_LEAF_MAX : in.leafMax));
	const float padScale = ldexpf(1.0f, -in.padExp);
	out = PtBuildOutput();
	if (n == 0) return hipSuccess;
	if (n > 0x0FFFFFFFu) return hipErrorInvalidValue;
	const bool ploc = in.builder == PT_BUILDER_PLOC;

	// ---- scratch: ONE allocation carved into 256-byte aligned pieces (hipMalloc/hipFree of two dozen multi-GB
	// buffers cost more than the build itself), freed on exit ----
	const uint32_t nSortBlocks = divUp(n, kSuperTile);
	const size_t histLen = (size_t)kRadixBins * nSortBlocks;
	const size_t nNodes2 = 2 * (size_t)n - 1;
	const uint32_t plocBlocks = divUp(n, kBlock);
	size_t arenaBytes = 0;
	auto reserve = [&](size_t bytes) {
		size_t at = arenaBytes;
		arenaBytes += (bytes + 255) & ~(size_t)255;
		return at;
	};
	// primitive boxes interleaved (lo, hi) in ONE array: the gathers through the sorted order touch one 32-byte record per
	// primitive instead of two 16-byte ones in different arrays; the kernels see it as two pointers with a stride of 2
	// (the LBVH over triangles keeps only the box centres: 12 bytes per triangle instead of 32 to write and to read back)
	const bool needPrimBoxes = ploc || boxMode;
	const size_t oTriLo = reserve(needPrimBoxes ? 2 * sizeof(float4) * (size_t)n : 0), oBounds = reserve(64);
	const size_t oCentroid = reserve(needPrimBoxes ? 0 : 3 * sizeof(float) * (size_t)n);
	const size_t oTriPos = reserve(boxMode ? 0 : 3 * sizeof(float4) * (size_t)n);
	const size_t oKeysA = reserve(4 * (size_t)n), oKeysB = reserve(4 * (size_t)n), oValsX = reserve(4 * (size_t)n);
	const size_t oHist = reserve(4 * histLen), oScan = reserve(4 * scanScratchLen(histLen));
	// LBVH: one 16-byte topology record per internal node + the leaves' parents; PLOC: children + subtree leaf counts
	const size_t oTopo = reserve(ploc ? 0 : sizeof(int4) * n), oLeafParent = reserve(ploc ? 0 : sizeof(int) * n);
	const size_t oChildren = reserve(ploc ? sizeof(int2) * n : 0), oSubCount = reserve(ploc ? sizeof(uint32_t) * nNodes2 : 0);
	const size_t oBoxes = reserve(sizeof(float) * 8 * nNodes2);
	const size_t oArrive = reserve(ploc ? 0 : sizeof(unsigned int) * n);
	const size_t oClusterA = reserve(ploc ? 4 * (size_t)n : 0), oClusterB = reserve(ploc ? 4 * (size_t)n : 0),
	             oNearest = reserve(ploc ? 4 * (size_t)n : 0), oLeafPrim = reserve(ploc && !boxMode ? 4 * (size_t)n : 0);
	const size_t oPlocSums = reserve(ploc ? 8 * (((size_t)plocBlocks + 63) / 64 * 64 + scanScratchLen(plocBlocks)) : 0);
	const size_t oPlocTotal = reserve(64);
	const size_t oTaskA = reserve(sizeof(CollapseTask) * n), oTaskB = reserve(sizeof(CollapseTask) * n), oCounters = reserve(256);
	char *arena = nullptr;
	hipError_t err = hipMalloc(&arena, arenaBytes);
#define TRY(x)                      \
	if (err == hipSuccess) err = (x)
#define AT(T, off) reinterpret_cast<T *>(arena + (off))
	float4 *triLo = AT(float4, oTriLo), *triHi = triLo + 1, *triPos = AT(float4, oTriPos);
	int *bounds = AT(int, oBounds);
	float *centroid = needPrimBoxes ? nullptr : AT(float, oCentroid);
	uint32_t *keysA = AT(uint32_t, oKeysA), *keysB = AT(uint32_t, oKeysB), *valsX = AT(uint32_t, oValsX),
	         *hist = AT(uint32_t, oHist), *scanScratch = AT(uint32_t, oScan);
	int4 *topo = AT(int4, oTopo);
	int *leafParent = AT(int, oLeafParent);
	int2 *children = AT(int2, oChildren);
	float *boxes = AT(float, oBoxes);
	unsigned int *arrive = AT(unsigned int, oArrive);
	uint32_t *subCount = AT(uint32_t, oSubCount), *clusterA = AT(uint32_t, oClusterA), *clusterB = AT(uint32_t, oClusterB),
	         *nearest = AT(uint32_t, oNearest), *leafPrim = ploc && !boxMode ? AT(uint32_t, oLeafPrim) : nullptr;
	unsigned long long *plocSums = AT(unsigned long long, oPlocSums);
	uint32_t *plocTotal = AT(uint32_t, oPlocTotal);  // one packed (keep, merge) pair = two uint32
	CollapseTask *taskA = AT(CollapseTask, oTaskA), *taskB = AT(CollapseTask, oTaskB);
	uint32_t *counters = AT(uint32_t, oCounters);  // [1] = triangle slots handed out (PLOC); [8 + L] = wide nodes of level L
#undef AT
	// outputs
	PtNode4 *nodes = nullptr;
	PtTri48 *tris = nullptr;
	uint32_t *codes = nullptr, *order = nullptr;
	const int planes = in.triPlanes == 1 ? 1 : 3;

	uint32_t depth = 0, nodeCount = 0;
	if (err == hipSuccess)
	{
		const uint32_t gridTris = divUp(n, kBlock);
		k_init_bounds<<<1, 64, 0, stream>>>(bounds);
		if (boxMode) k_box_setup<<<gridTris, kBlock, 0, stream>>>(in.boxLo, in.boxHi, n, triLo, triHi, bounds);
		else
			k_tri_setup<<<gridTris < 2048 ? gridTris : 2048, kBlock, 0, stream>>>(in.vertices, in.indices, n, triLo, triHi,
			                                                                       bounds, triPos, centroid);
		// The four outputs are allocated while the device runs the set-up kernel, which needs none of them: a hipMalloc is
		// ~0.3 ms of host time whatever its size (scripts/ubench/malloc_time.hip), and four of them in front of the first
		// launch were 1.2 ms of every build in which the device did nothing.
		TRY(hipMalloc(&codes, 4 * (size_t)n));
		TRY(hipMalloc(&order, 4 * (size_t)n));
		TRY(hipMalloc(&nodes, sizeof(PtNode4) * (size_t)n));  // a wide node has >= 2 children unless it is the root
		if (!boxMode) TRY(hipMalloc(&tris, (size_t)planes * sizeof(PtTri48) * (size_t)n));  // three axis-rotated planes, or the identity
	}
	if (err == hipSuccess)
	{
		const uint32_t gridTris = divUp(n, kBlock);
		k_morton<<<gridTris, kBlock, 0, stream>>>(triLo, triHi, centroid, n, bounds, codes);
		// The first pass reads the codes where k_morton left them (the `codes` output, never written again) and takes a key's
		// position as its value; the value buffers alternate so that the LAST pass scatters into the `order` output -- no copy
		// of the keys in, none of the sorted order out (0.4 GB each way per copy at 100 M primitives).
		constexpr int kPasses = kSortBits / kRadixBits;
		const uint32_t *kin = codes, *vin = nullptr;
		for (int pass = 0; pass < kPasses; ++pass)
		{
			const int shift = pass * kRadixBits;
			uint32_t *kout = (pass & 1) ? keysB : keysA, *vout = ((kPasses - 1 - pass) & 1) ? valsX : order;
			k_radix_hist<<<nSortBlocks, kBlock, 0, stream>>>(kin, n, shift, hist, nSortBlocks);
			scanExclusive<uint32_t>(hist, histLen, scanScratch, nullptr, stream);
			k_radix_scatter<<<nSortBlocks, kBlock, 0, stream>>>(kin, vin, kout, vout, n, shift, hist, nSortBlocks);
			kin = kout;
			vin = vout;
		}
		// (kin, vin) hold the sorted keys and primitives; vin == order
		int rootNode = 0;  // LBVH: internal node 0 (or the only leaf, index 0, when n == 1)
		if (err == hipSuccess && !ploc)
		{
			if (n > 1) k_lbvh_topology<<<divUp(n - 1, kBlock), kBlock, 0, stream>>>(kin, (int)n, topo, leafParent, arrive);
			else TRY(hipMemsetAsync(leafParent, 0xFF, sizeof(int), stream));  // single leaf: parent = -1
			// leaf boxes, the bottom-up union and (triangles) the leaf-ordered triangle array: the LBVH's leaf order is the
			// sorted order, so the thread that gathers a primitive's positions for its box writes its three records too
			static const int refitPad = getenv("PT_REFIT_LDS_PAD") ? atoi(getenv("PT_REFIT_LDS_PAD")) : 0;  // tuning: occupancy limiter
			k_refit<<<gridTris, kBlock, (size_t)refitPad, stream>>>(boxMode ? nullptr : triPos, triLo, triHi, vin, (int)n, topo, leafParent, bounds,
			                                         boxes, arrive, padScale, tris, planes);
		}
		if (err == hipSuccess && ploc)
		{
			k_ploc_leaves<<<gridTris, kBlock, 0, stream>>>(triLo, triHi, vin, (int)n, bounds, boxes, subCount, clusterA,
			                                               padScale);
			uint32_t m = n, created = 0;
			uint32_t *cin = clusterA, *cout = clusterB;
			while (err == hipSuccess && m > 1)
			{
				const uint32_t g = divUp(m, kBlock);
				k_ploc_nearest<<<g, kBlock, 0, stream>>>(cin, m, boxes, nearest);
				k_ploc_count<<<g, kBlock, 0, stream>>>(nearest, m, plocSums);
				scanExclusive<unsigned long long>(plocSums, g, plocSums + (((size_t)g + 63) & ~(size_t)63),
				                                  reinterpret_cast<unsigned long long *>(plocTotal), stream);
				k_ploc_apply<<<g, kBlock, 0, stream>>>(cin, nearest, m, plocSums, created, boxes, subCount, children, cout);
				uint32_t h[2];
				err = hipMemcpyAsync(h, plocTotal, sizeof(h), hipMemcpyDeviceToHost, stream);  // (clusters left, merges)
				TRY(hipStreamSynchronize(stream));
				if (err != hipSuccess) break;
				// the globally closest pair inside the window is always mutual, so every pass merges at least once
				if (h[1] == 0 || h[0] + h[1] != m || created + h[1] > n - 1) { err = hipErrorAssert; break; }
				created += h[1];
				m = h[0];
				uint32_t *t = cin;
				cin = cout;
				cout = t;
			}
			rootNode = n > 1 ? (int)(created - 1) : 0;
			if (err == hipSuccess && n > 1 && created != n - 1) err = hipErrorAssert;
		}
		// collapse: every level's launch is enqueued at once, the level sizes stay on the device (k_collapse_level)
		uint32_t init[64] = {};
		init[8] = 1u;  // level 0 = the root
		TRY(hipMemcpyAsync(counters, init, sizeof(init), hipMemcpyHostToDevice, stream));
		CollapseTask root = { rootNode, 0u };
		TRY(hipMemcpyAsync(taskA, &root, sizeof(root), hipMemcpyHostToDevice, stream));
		if (err == hipSuccess)
		{
			const Bvh2View view = { ploc ? nullptr : topo, ploc ? children : nullptr, ploc ? subCount : nullptr, (int)n };
			const uint32_t grid = gridTris < 2048u ? gridTris : 2048u;
			CollapseTask *tin = taskA, *tout = taskB;
			for (int level = 0; level < kMaxLevels; ++level)
			{
				// a level cannot hold more wide nodes than 4^level: small grids for the top of the tree
				const uint32_t cap = level < 10 ? divUp((size_t)1 << (2 * level), kBlock) : grid;
				k_collapse_level<<<cap < grid ? cap : grid, kBlock, 0, stream>>>(tin, tout, counters + 8, level, view, boxes, vin, boxMode,
				                                                                 leafPrim, nodes, counters + 1, leafMax, in.nodeBase,
				                                                                 (!ploc && !boxMode) ? tris + (size_t)(planes - 1) * n : nullptr, bounds, padScale);
				CollapseTask *t = tin;
				tin = tout;
				tout = t;
			}
			uint32_t h[64];
			err = hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, stream);
			TRY(hipStreamSynchronize(stream));
			if (err == hipSuccess)
			{
				nodeCount = 0;
				for (int level = 0; level <= kMaxLevels; ++level)
				{
					if (h[8 + level] == 0) break;
					++depth;  // (kMaxLevels + 1 = the tree did not fit: the caller rejects depths like that)
					nodeCount += h[8 + level];
				}
				if (h[1] > n || nodeCount > n) err = hipErrorAssert;
			}
		}
		if (err == hipSuccess && leafPrim) k_write_tris<<<gridTris, kBlock, 0, stream>>>(leafPrim, n, triPos, tris, planes);
		TRY(hipGetLastError());
		if (in.doneEvent) TRY(hipEventRecord(in.doneEvent, stream));
	}
	if (err == hipSuccess)
	{
		int hb[8];
		err = hipMemcpy(hb, bounds, sizeof(int) * 7, hipMemcpyDeviceToHost);
		for (int k = 0; k < 3; ++k)
		{
			out.cmin[k] = orderedToFloat(hb[k]);
			out.cmax[k] = orderedToFloat(hb[3 + k]);
		}
	}
#undef TRY
	hipStreamSynchronize(stream);
	hipFree(arena);
	if (err != hipSuccess)
	{
		hipFree(nodes); hipFree(tris); hipFree(codes); hipFree(order);
		return err;
	}
	out.nodes = nodes;
	out.tris = tris;
	out.codes = codes;
	out.order = order;
	out.nNodes = nodeCount;
	out.nTris = n;
	out.triPlanes = planes;
	out.depth = depth;
	return hipSuccess;
}

// ------------------------------------------------------------------------- quad leaves ----
namespace
{
// 0: not a fan pair; 1: A = first triangle, B = second; 2: A = second, B = first.  x, y: the two records of the leaf in
// plane 2 (coordinates in x, y, z order).  Vertices are compared bit for bit (the shared edge function must be exact).
__device__ __forceinline__ int quadPattern(const PtTri48 &x, const PtTri48 &y)
{
	auto same = [](const float *p, const float *q) {
		return __float_as_uint(p[0]) == __float_as_uint(q[0]) && __float_as_uint(p[1]) == __float_as_uint(q[1]) &&
		       __float_as_uint(p[2]) == __float_as_uint(q[2]);
	};
	if (same(y.v0, x.v0) && same(y.v1, x.v2)) return 1;
	if (same(x.v0, y.v0) && same(x.v1, y.v2)) return 2;
	return 0;
}

__device__ __forceinline__ bool quadCandidate(uint32_t ref)
{
	return (ref & PT_LEAF_FLAG) && ref < PT_INSTANCE_END && ((ref >> 28) & 7u) == 1u;  // a leaf of exactly two triangles
}

// One thread per child slot: flags[4 * node + c] = 1 if the slot holds a fan pair.
__global__ void __launch_bounds__(kBlock) k_quad_flag(const PtNode4 *nodes, uint32_t nNodes, const PtTri48 *plane2, uint32_t *flags)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= 4u * nNodes) return;
	const uint32_t ref = nodes[i >> 2].child[i & 3u];
	uint32_t f = 0;
	if (quadCandidate(ref))
	{
		const uint32_t first = ref & 0x0FFFFFFFu;
		f = quadPattern(plane2[first], plane2[first + 1]) != 0 ? 1u : 0u;
	}
	flags[i] = f;
}

// The same test over a SAMPLE of the node array (the last `nSample` nodes: breadth-first numbering puts the leaves there),
// counted: count[0] += fan pairs, count[1] += triangles in leaves.  A scene of millions of nodes whose leaves do not pair up
// (a triangle soup) is told apart here for the price of the sample, not of a pass over every node and triangle.
__global__ void __launch_bounds__(kBlock) k_quad_sample(const PtNode4 *nodes, uint32_t firstNode, uint32_t nSample, const PtTri48 *plane2,
                                                        unsigned long long *count)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	uint32_t pair = 0, tris = 0;
	if (i < 4u * nSample)
	{
		const uint32_t ref = nodes[firstNode + (i >> 2)].child[i & 3u];
		if ((ref & PT_LEAF_FLAG) && ref < PT_INSTANCE_END)
		{
			tris = ((ref >> 28) & 7u) + 1u;
			if (quadCandidate(ref))
			{
				const uint32_t first = ref & 0x0FFFFFFFu;
				pair = quadPattern(plane2[first], plane2[first + 1]) != 0 ? 1u : 0u;
			}
		}
	}
	const unsigned long long pm = __ballot(pair != 0);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) tris += __shfl_xor(tris, o);
	if ((threadIdx.x & 63) == 0)
	{
		if (pm) atomicAdd(&count[0], (unsigned long long)__popcll(pm));
		if (tris) atomicAdd(&count[1], (unsigned long long)tris);
	}
}

// index = exclusive scan of the flags.  Writes the quad records (all three planes, from the triangle records of the same
// plane) and patches the child reference in the copy of the node array.
__global__ void __launch_bounds__(kBlock)
    k_quad_write(const PtNode4 *nodes, uint32_t nNodes, const PtTri48 *plane2, const uint32_t *index,
                 PtNode4 *nodesOut, PtQuad64 *quads, uint32_t nQuads)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= 4u * nNodes) return;
	const uint32_t ref = nodes[i >> 2].child[i & 3u];
	if (!quadCandidate(ref)) return;
	const uint32_t first = ref & 0x0FFFFFFFu;
	const int pat = quadPattern(plane2[first], plane2[first + 1]);
	if (pat == 0) return;
	const uint32_t qi = index[i];
	if (qi >= nQuads) return;  // cannot happen (the scan counted exactly these slots); keeps a wrong count from writing outside
	const uint32_t ia = pat == 1 ? first : first + 1, ib = pat == 1 ? first + 1 : first;
	// the quad's three rotated records from the triangles' identity records (plane r holds the coordinates ((r+1)%3,
	// (r+2)%3, r), as the triangle planes do): the same values whether the scene keeps three triangle planes or one
	const PtTri48 a = plane2[ia], b = plane2[ib];
	for (int r = 0; r < 3; ++r)
	{
		const int ax[3] = { (r + 1) % 3, (r + 2) % 3, r };
		PtQuad64 q;
		for (int k = 0; k < 3; ++k)
		{
			q.v0[k] = a.v0[ax[k]];
			q.v1[k] = a.v1[ax[k]];
			q.v2[k] = a.v2[ax[k]];
			q.v3[k] = b.v2[ax[k]];
		}
		q.primA = a.prim;
		q.primB = b.prim;
		q.pad[0] = q.pad[1] = 0u;
		quads[(size_t)r * nQuads + qi] = q;
	}
	nodesOut[i >> 2].child[i & 3u] = PT_LEAF_FLAG | (PT_LEAF_QUAD << 28) | qi;
}
}  // namespace

hipError_t ptBuildQuadLeaves(const PtNode4 *nodes, uint32_t nNodes, const PtTri48 *tris, uint32_t nTris, float minShare,
                             PtQuadOutput &out, hipStream_t stream)
{
	out = PtQuadOutput();
	if (!nNodes || nTris < 2 || nNodes > 0x3FFFFFFFu) return hipSuccess;
	const size_t nSlots = 4 * (size_t)nNodes;
	uint32_t *flags = nullptr;
	const size_t words = ((nSlots + 63) & ~(size_t)63) + scanScratchLen(nSlots) + 64;
	hipError_t e = hipSuccess;
	constexpr uint32_t kSample = 1u << 16;
	if (nNodes > 8u * kSample)
	{
		// big trees: look at the last 65 536 nodes first; when not even half of the share asked for pairs up there, the scene
		// is not made of quads (100 M-triangle soup: 2 ms of flagging, scanning and reading 7 GB saved per build)
		unsigned long long *count = nullptr, h[2] = { 0, 0 };
		e = hipMalloc(&count, sizeof(h));
		if (e == hipSuccess) e = hipMemsetAsync(count, 0, sizeof(h), stream);
		if (e == hipSuccess)
		{
			k_quad_sample<<<divUp(4u * kSample, kBlock), kBlock, 0, stream>>>(nodes, nNodes - kSample, kSample, tris, count);
			e = hipMemcpyAsync(h, count, sizeof(h), hipMemcpyDeviceToHost, stream);
		}
		if (e == hipSuccess) e = hipStreamSynchronize(stream);
		hipFree(count);
		if (e != hipSuccess) return e;
		if (2.0 * (double)h[0] < 0.5 * (double)minShare * (double)h[1]) return hipSuccess;
	}
	e = hipMalloc(&flags, 4 * words);
	if (e != hipSuccess) return e;
	uint32_t *scratch = flags + ((nSlots + 63) & ~(size_t)63), *total = flags + words - 64;
	k_quad_flag<<<divUp((uint32_t)nSlots, kBlock), kBlock, 0, stream>>>(nodes, nNodes, tris, flags);
	scanExclusive<uint32_t>(flags, nSlots, scratch, total, stream);  // the grand total comes out of the top level of the recursion
	uint32_t nQuads = 0;
	e = hipMemcpyAsync(&nQuads, total, 4, hipMemcpyDeviceToHost, stream);
	if (e == hipSuccess) e = hipStreamSynchronize(stream);
	if (e == hipSuccess && nQuads && 2.0 * (double)nQuads >= (double)minShare * (double)nTris)
	{
		e = hipMalloc(&out.nodes, sizeof(PtNode4) * (size_t)nNodes);
		if (e == hipSuccess) e = hipMalloc(&out.quads, sizeof(PtQuad64) * 3 * (size_t)nQuads);
		if (e == hipSuccess) e = hipMemcpyAsync(out.nodes, nodes, sizeof(PtNode4) * (size_t)nNodes, hipMemcpyDeviceToDevice, stream);
		if (e == hipSuccess)
		{
			k_quad_write<<<divUp((uint32_t)nSlots, kBlock), kBlock, 0, stream>>>(nodes, nNodes, tris, flags, out.nodes, out.quads, nQuads);
			e = hipStreamSynchronize(stream);
		}
		if (e == hipSuccess) e = hipGetLastError();
		if (e == hipSuccess) out.nQuads = nQuads;
		else
		{
			hipFree(out.nodes);
			hipFree(out.quads);
			out = PtQuadOutput();
		}
	}
	hipFree(flags);
	return e;
}

// -------------------------------------------------------------------------- entry node ----
namespace
{
// One thread: the top of the tree is a handful of nodes.  Starting from the root's children, the inner child with the
// largest box is replaced by its own children while the list still fits PT_ENTRY_WIDTH.  Boxes: an inner child keeps the
// box its parent holds for it (dequantised with the fmaf the builder verified it with); a LEAF child gets the exact box of
// its triangles -- on the 255-step grid of a room-sized node a wall is a slab 1/128 of the room thick that contains the
// origin of every ray leaving that wall, so each bounce ray would test the quad it starts on; the exact slab is left
// within t < tmin and the interval test discards it.  All boxes are padded by 2^-18 * max|coordinate| (as the builder pads
// its leaves), stored relative to the node's origin (rounded outwards): the traversal evaluates
// t = fma(plane, 1/d, (origin - o)/d), whose rounding errors are relative to the distance travelled (covered by the test's
// 1e-6 slack) plus 2^-23 of the node's extent (covered by the padding, 2^-18).
__global__ void k_entry_node(const PtNode4 *nodes, uint32_t nNodes, uint32_t root, const PtTri48 *plane2, const PtQuad64 *quads2,
                             bool instanceLeaves, float4 *rec)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	constexpr int W = PT_ENTRY_WIDTH, G = W / 4;
	uint32_t ref[W];
	float lo[W][3], hi[W][3];
	int n = 0;
	auto childrenOf = [&](uint32_t node, int at, int &count) {
		// children of `node`: the first replaces entry `at`, the others are appended; returns how many there were
		const PtNode4 nd = nodes[node];
		const float sc[3] = { nd.scaleX, nd.scaleY, nd.scaleZ };
		count = 0;
		for (int c = 0; c < 4; ++c)
		{
			if (nd.child[c] == PT_EMPTY_CHILD) continue;
			const int dst = count == 0 ? at : n++;
			ref[dst] = nd.child[c];
			for (int a = 0; a < 3; ++a)
			{
				lo[dst][a] = __builtin_fmaf((float)((nd.qlo[a] >> (8 * c)) & 0xFFu), sc[a], nd.origin[a]);
				hi[dst][a] = __builtin_fmaf((float)((nd.qhi[a] >> (8 * c)) & 0xFFu), sc[a], nd.origin[a]);
			}
			++count;
		}
	};
	auto liveChildren = [&](uint32_t node) {
		int k = 0;
		for (int c = 0; c < 4; ++c) k += nodes[node].child[c] != PT_EMPTY_CHILD ? 1 : 0;
		return k;
	};
	{
		int k;
		n = 1;  // entry 0 is a placeholder the root's first child replaces
		childrenOf(root, 0, k);
		if (k == 0) n = 0;
	}
	for (;;)
	{
		int best = -1;
		float bestArea = -1.0f;
		for (int i = 0; i < n; ++i)
		{
			if (ref[i] & PT_LEAF_FLAG) continue;
			const int k = liveChildren(ref[i]);
			if (k == 0 || n - 1 + k > W) continue;
			const float dx = hi[i][0] - lo[i][0], dy = hi[i][1] - lo[i][1], dz = hi[i][2] - lo[i][2];
			const float area = (dx * dy + dy * dz) + dz * dx;
			if (area > bestArea)
			{
				bestArea = area;
				best = i;
			}
		}
		if (best < 0) break;
		int k;
		childrenOf(ref[best], best, k);
	}
	// inner children first: the lane visits the children last to first, so subtrees are entered after the leaves have
	// shortened the ray (a node step culls with the best hit so far; a leaf is tested whatever the order)
	for (int i = 0, at = 0; i < n; ++i)
		if (!(ref[i] & PT_LEAF_FLAG))
		{
			for (int j = i; j > at; --j)
			{
				const uint32_t r = ref[j];
				ref[j] = ref[j - 1];
				ref[j - 1] = r;
				for (int a = 0; a < 3; ++a)
				{
					float t = lo[j][a];
					lo[j][a] = lo[j - 1][a];
					lo[j - 1][a] = t;
					t = hi[j][a];
					hi[j][a] = hi[j - 1][a];
					hi[j - 1][a] = t;
				}
			}
			++at;
		}
	float maxAbs = 0.0f;
	for (int i = 0; i < n; ++i)
	{
		if ((ref[i] & PT_LEAF_FLAG) && !instanceLeaves)  // (a top-level leaf is an instance: its box stays the tree's)
		{
			// exact box of the leaf's triangles (plane 2 holds the coordinates in x, y, z order)
			float l3[3] = { INFINITY, INFINITY, INFINITY }, h3[3] = { -INFINITY, -INFINITY, -INFINITY };
			auto add = [&](const float *v) {
				for (int a = 0; a < 3; ++a)
				{
					l3[a] = fminf(l3[a], v[a]);
					h3[a] = fmaxf(h3[a], v[a]);
				}
			};
			const uint32_t first = ref[i] & 0x0FFFFFFFu, cnt = ((ref[i] >> 28) & 7u) + 1u;
			if (quads2 && ((ref[i] >> 28) & 7u) == PT_LEAF_QUAD)
			{
				const PtQuad64 &q = quads2[first];
				add(q.v0);
				add(q.v1);
				add(q.v2);
				add(q.v3);
			}
			else
				for (uint32_t k = 0; k < cnt; ++k)
				{
					const PtTri48 &t = plane2[first + k];
					add(t.v0);
					add(t.v1);
					add(t.v2);
				}
			for (int a = 0; a < 3; ++a)
			{
				// never larger than what the tree says (the tree's box is what the other pipelines cull with)
				lo[i][a] = fmaxf(lo[i][a], l3[a]);
				hi[i][a] = fminf(hi[i][a], h3[a]);
			}
		}
		for (int a = 0; a < 3; ++a) maxAbs = fmaxf(maxAbs, fmaxf(fabsf(lo[i][a]), fabsf(hi[i][a])));
	}
	uint32_t nInnerKids = 0;
	for (int i = 0; i < n; ++i) nInnerKids += (ref[i] & PT_LEAF_FLAG) ? 0u : 1u;
	if (nInnerKids == 0 && !instanceLeaves)
	{
		// all leaves (k_paths' flat form): largest box first, i.e. visited LAST -- the small leaves (the things standing in a
		// room) are tested before the large ones (its walls), which are then skipped when they lie behind what was hit
		// (insertion sort, stable; any order gives the same image)
		auto areaOf = [&](int i) {
			const float dx = hi[i][0] - lo[i][0], dy = hi[i][1] - lo[i][1], dz = hi[i][2] - lo[i][2];
			return (dx * dy + dy * dz) + dz * dx;
		};
		for (int i = 1; i < n; ++i)
			for (int j = i; j > 0 && areaOf(j) > areaOf(j - 1); --j)
			{
				const uint32_t r = ref[j];
				ref[j] = ref[j - 1];
				ref[j - 1] = r;
				for (int a = 0; a < 3; ++a)
				{
					float t = lo[j][a];
					lo[j][a] = lo[j - 1][a];
					lo[j - 1][a] = t;
					t = hi[j][a];
					hi[j][a] = hi[j - 1][a];
					hi[j - 1][a] = t;
				}
			}
	}
	const float pad = maxAbs * 3.814697265625e-06f;  // 2^-18
	float org[3] = { INFINITY, INFINITY, INFINITY };
	for (int i = 0; i < n; ++i)
		for (int a = 0; a < 3; ++a) org[a] = fminf(org[a], lo[i][a]);
	float *out = reinterpret_cast<float *>(rec);
	uint32_t *outU = reinterpret_cast<uint32_t *>(rec);
	for (int a = 0; a < 3; ++a) out[a] = n ? org[a] : 0.0f;
	out[3] = 0.0f;
	outU[4] = (uint32_t)n;
	uint32_t nInner = 0;
	for (int i = 0; i < n; ++i) nInner += (ref[i] & PT_LEAF_FLAG) ? 0u : 1u;
	outU[5] = nInner;
	outU[6] = outU[7] = 0u;
	for (int i = 0; i < W; ++i)
	{
		outU[8 + i] = i < n ? ref[i] : PT_EMPTY_CHILD;
		for (int a = 0; a < 3; ++a)
		{
			float pl = INFINITY, ph = -INFINITY;
			if (i < n)
			{
				// relative to the origin, padded, and one more step outwards for the rounding of the two operations
				pl = (lo[i][a] - pad) - org[a];
				ph = (hi[i][a] + pad) - org[a];
				pl = nextafterf(pl, -INFINITY);
				ph = nextafterf(ph, INFINITY);
			}
			out[4 * (2 + G) + (0 + a) * W + i] = pl;
			out[4 * (2 + G) + (3 + a) * W + i] = ph;
		}
	}
}
}  // namespace

hipError_t ptBuildEntryNode(const PtNode4 *nodes, uint32_t nNodes, uint32_t root, const PtTri48 *tris, uint32_t nTris,
                            const PtQuad64 *quads, uint32_t nQuads, bool instanceLeaves, uint4 **rec, bool *flat, hipStream_t stream)
{
	*rec = nullptr;
	*flat = false;
	if (!nNodes || root >= nNodes) return hipSuccess;
	uint4 *d = nullptr;
	hipError_t e = hipMalloc(&d, sizeof(uint4) * PT_ENTRY_REC);
	if (e != hipSuccess) return e;
	k_entry_node<<<1, 64, 0, stream>>>(nodes, nNodes, root, tris, nQuads ? quads + 2 * (size_t)nQuads : nullptr,
	                                   instanceLeaves, reinterpret_cast<float4 *>(d));
	e = hipGetLastError();
	uint32_t head[8] = {};
	if (e == hipSuccess) e = hipMemcpyAsync(head, d, sizeof(head), hipMemcpyDeviceToHost, stream);
	if (e == hipSuccess) e = hipStreamSynchronize(stream);
	if (e != hipSuccess)
	{
		hipFree(d);
		return e;
	}
	*rec = d;
	*flat = head[4] != 0u && head[5] == 0u && !instanceLeaves;  // children, inner children
	return hipSuccess;
}

// ------------------------------------------------------------- several bottom-level trees in one set of arrays ----
namespace
{
__global__ void __launch_bounds__(kBlock) k_relocate_nodes(const PtNode4 *src, uint32_t n, uint32_t nodeBase, uint32_t triBase, PtNode4 *dst)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= n) return;
	PtNode4 nd = src[i];
#pragma unroll
	for (int c = 0; c < 4; ++c)
	{
		const uint32_t r = nd.child[c];
		if (r == PT_EMPTY_CHILD) continue;
		nd.child[c] = (r & PT_LEAF_FLAG) ? ((r & 0xF0000000u) | ((r & 0x0FFFFFFFu) + triBase)) : r + nodeBase;
	}
	dst[nodeBase + i] = nd;
}

__global__ void __launch_bounds__(kBlock)
    k_relocate_tris(const PtTri48 *src, uint32_t n, uint32_t triBase, uint32_t total, PtTri48 *dst)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= 3u * n) return;
	const uint32_t plane = i / n, j = i - plane * n;
	PtTri48 t = src[i];
	t.prim += triBase;  // primitive ids run through all models in order
	dst[(size_t)plane * total + triBase + j] = t;
}

__global__ void __launch_bounds__(kBlock) k_relocate_u32(const uint32_t *src, uint32_t n, uint32_t add, uint32_t *dst)
{
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i < n) dst[i] = src[i] + add;
}
}  // namespace

hipError_t ptRelocateBlas(const PtBuildOutput &part, uint32_t nodeBase, uint32_t triBase, uint32_t totalTris, PtNode4 *nodes,
                          PtTri48 *tris, uint32_t *codes, uint32_t *order, hipStream_t stream)
{
	if (part.nNodes) k_relocate_nodes<<<divUp(part.nNodes, kBlock), kBlock, 0, stream>>>(part.nodes, part.nNodes, nodeBase, triBase, nodes);
	if (part.nTris)
	{
		k_relocate_tris<<<divUp(3u * part.nTris, kBlock), kBlock, 0, stream>>>(part.tris, part.nTris, triBase, totalTris, tris);
		k_relocate_u32<<<divUp(part.nTris, kBlock), kBlock, 0, stream>>>(part.codes, part.nTris, 0u, codes + triBase);
		k_relocate_u32<<<divUp(part.nTris, kBlock), kBlock, 0, stream>>>(part.order, part.nTris, triBase, order + triBase);
	}
	return hipGetLastError();
}
