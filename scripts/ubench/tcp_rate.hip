// Micro-benchmark 3: how fast does a CU's vector L1 (TA + TCP) work through 16-byte loads, by address pattern?
//   coalesced   lane l reads element base + l: 1 KB contiguous per wave instruction (8 lines of 128 B)
//   quad        lanes 4k .. 4k+3 read the four 16-byte pieces of ONE 64-byte record, a different (random) record per quad
//   divergent   every lane reads one 16-byte piece of its OWN (random) 64-byte record: 64 lines per wave instruction
//   pair        lanes 2k, 2k+1 read two adjacent pieces of one record, a different record per pair: 32 lines
//   node        every lane reads the four pieces of its own record with four instructions (what pt_node_step does)
// Working sets: 16 KB (L1-resident), 2 MB (L2-resident).  16 waves per CU, four loads in flight per lane.
// Output: cycles of one CU per wave instruction (at the clock rocm-smi reports; 2.4 GHz assumed) = 1 / throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_ITER 512

template <int kMode>
__global__ void __launch_bounds__(256) k_loads(const uint4 *buf, uint32_t recMask, uint4 *out)
{
	const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
	uint32_t s = wave * 2654435761u + 12345u;
	uint4 acc = make_uint4(0, 0, 0, 0);
	for (int i = 0; i < N_ITER; ++i)
	{
		uint4 v[4];
		s = s * 1664525u + 1013904223u;
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			const uint32_t sk = kMode == 3 ? s : s + (uint32_t)k * 0x632BE5ABu;  // node: the same record four times
			uint32_t idx;
			if (kMode == 0) idx = (((sk * 0x85EBCA6Bu) >> 8) & recMask & ~15u) * 4u + lane;  // 64 consecutive pieces
			else if (kMode == 1)
			{
				const uint32_t h = (sk ^ ((lane >> 2) * 0x9E3779B9u)) * 0x85EBCA6Bu;  // one random record per quad of lanes
				idx = ((h >> 8) & recMask) * 4u + (lane & 3u);
			}
			else if (kMode == 4)
			{
				const uint32_t h = (sk ^ ((lane >> 1) * 0x9E3779B9u)) * 0x85EBCA6Bu;  // one random record per pair of lanes
				idx = ((h >> 8) & recMask) * 4u + (lane & 1u) + 2u * ((uint32_t)k & 1u);
			}
			else
			{
				const uint32_t h = (sk ^ (lane * 0x9E3779B9u)) * 0x85EBCA6Bu;  // one random record per lane
				idx = ((h >> 8) & recMask) * 4u + (kMode == 3 ? (uint32_t)k : ((h >> 4) & 3u));
			}
			v[k] = buf[idx];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
		{
			acc.x ^= v[k].x;
			acc.y += v[k].y;
			acc.z ^= v[k].z;
			acc.w += v[k].w;
		}
	}
	if (acc.x == 0x12345678u) out[wave * 64 + lane] = acc;
}

typedef void (*kfn)(const uint4 *, uint32_t, uint4 *);
static void run(const char *name, kfn f, const uint4 *buf, uint32_t records, uint4 *out)
{
	const int blocks = 256 * 4;  // 16 waves per CU
	hipEvent_t a, b;
	hipEventCreate(&a);
	hipEventCreate(&b);
	hipLaunchKernelGGL(f, blocks, 256, 0, 0, buf, records - 1, out);
	hipEventRecord(a);
	hipLaunchKernelGGL(f, blocks, 256, 0, 0, buf, records - 1, out);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	const double instrPerCu = (double)N_ITER * 4 * 16;  // wave instructions each CU executes
	const double ns = ms * 1e6 / instrPerCu;
	printf("%-10s %7u records (%6u KB): %7.2f ns = %6.1f cycles at 2.4 GHz per wave instruction and CU\n", name, records,
	       records * 64 / 1024, ns, ns * 2.4);
}

int main()
{
	const uint32_t maxRec = 1u << 15;  // 2 MB
	uint4 *buf, *out;
	hipMalloc(&buf, (size_t)maxRec * 64);
	hipMemset(buf, 1, (size_t)maxRec * 64);
	hipMalloc(&out, (size_t)256 * 4 * 4 * 64 * sizeof(uint4));
	for (uint32_t rec : { 256u, 2048u, maxRec })
	{
		run("coalesced", k_loads<0>, buf, rec, out);
		run("quad", k_loads<1>, buf, rec, out);
		run("divergent", k_loads<2>, buf, rec, out);
		run("node", k_loads<3>, buf, rec, out);
		run("pair", k_loads<4>, buf, rec, out);
	}
	return 0;
}
