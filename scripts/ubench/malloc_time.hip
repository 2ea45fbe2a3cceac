// How long do hipMalloc / hipFree of multi-GB buffers take on this box?  (the BVH build allocates its scratch arena and its
// outputs between the two events that time it)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
	hipFree(nullptr);
	const double gb[] = { 0.4, 0.4, 4.8, 6.4, 17.0, 17.0, 6.4, 4.8 };
	for (double g : gb)
	{
		size_t bytes = (size_t)(g * 1e9);
		void *p = nullptr;
		auto t0 = std::chrono::steady_clock::now();
		hipError_t e = hipMalloc(&p, bytes);
		auto t1 = std::chrono::steady_clock::now();
		if (e != hipSuccess) { printf("hipMalloc %.1f GB failed\n", g); return 1; }
		hipMemsetAsync(p, 0, 256, 0);
		hipDeviceSynchronize();
		auto t2 = std::chrono::steady_clock::now();
		hipFree(p);
		auto t3 = std::chrono::steady_clock::now();
		printf("%.1f GB: hipMalloc %.3f ms, hipFree %.3f ms\n", g, std::chrono::duration<double, std::milli>(t1 - t0).count(),
		       std::chrono::duration<double, std::milli>(t3 - t2).count());
	}
	return 0;
}
