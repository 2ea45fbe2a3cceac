// Micro-benchmark: sustained issue rate of common VALU instructions on gfx950, wave64,
// at 1/2/4 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, float a, float b, unsigned u)
{
	float x0 = threadIdx.x, x1 = a, x2 = b, x3 = a + b, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
	unsigned q = u + threadIdx.x;
	for (int i = 0; i < N_ITER; ++i)
	{
		if (OP == 0) { // fma
			x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
			x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
		} else if (OP == 1) { // cndmask via compare+select
			x0 = x0 > a ? x1 : x2; x1 = x1 > a ? x2 : x3; x2 = x2 > a ? x3 : x4; x3 = x3 > a ? x4 : x5;
			x4 = x4 > a ? x5 : x6; x5 = x5 > a ? x6 : x7; x6 = x6 > a ? x7 : x0; x7 = x7 > a ? x0 : x1;
		} else if (OP == 2) { // cvt ubyte
			x0 += (float)(q & 0xFF); x1 += (float)((q >> 8) & 0xFF); x2 += (float)((q >> 16) & 0xFF); x3 += (float)(q >> 24);
			q = q * 3u + 1u;
			x4 += (float)(q & 0xFF); x5 += (float)((q >> 8) & 0xFF); x6 += (float)((q >> 16) & 0xFF); x7 += (float)(q >> 24);
		} else if (OP == 3) { // min/max u32
			unsigned p0 = __float_as_uint(x0), p1 = __float_as_uint(x1), p2 = __float_as_uint(x2), p3 = __float_as_uint(x3);
			unsigned l = min(p0, p1), h = max(p0, p1), l2 = min(p2, p3), h2 = max(p2, p3);
			x0 = __uint_as_float(min(l, l2) + q); x1 = __uint_as_float(max(l, l2)); x2 = __uint_as_float(min(h, h2)); x3 = __uint_as_float(max(h, h2) ^ q);
		} else if (OP == 4) { // IEEE division
			x0 = x0 / a; x1 = x1 / b; x2 = x2 / a; x3 = x3 / b;
		} else if (OP == 5) { // mul+add unfused
			x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)q;
}
template <int OP>
void run(const char *name, int opsPerIter, float *d)
{
	int cus = 256;
	for (int wps : { 1, 2, 4, 8 })
	{
		int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		k<OP><<<blocks, 256>>>(d, 1.0001f, 0.5f, 7u);
		hipEventRecord(a);
		k<OP><<<blocks, 256>>>(d, 1.0001f, 0.5f, 7u);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		double instrPerSimd = (double)N_ITER * opsPerIter * wps;
		printf("%-10s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4GHz = %.2f cyc)\n", name, wps, ms,
		       ms * 1e6 / instrPerSimd, ms * 1e6 / instrPerSimd * 2.4);
	}
}
int main()
{
	float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
	run<0>("fma", 8, d); run<1>("cmp+cnd", 16, d); run<2>("cvt_ubyte+add", 17, d); run<3>("minmax", 10, d); run<4>("div", 4, d); run<5>("mul,add", 8, d);
	return 0;
}
