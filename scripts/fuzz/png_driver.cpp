#include "pt_hip.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv)
{
	int ok = 0, bad = 0;
	for (int i = 1; i < argc; ++i)
	{
		uint8_t *px = nullptr;
		uint32_t w = 0, h = 0;
		int rc = pt_load_png(argv[i], &px, &w, &h);
		if (rc == PT_OK) { ++ok; volatile uint8_t s = 0; for (size_t k = 0; k < (size_t)w * h * 4; ++k) s += px[k]; pt_free(px); }
		else ++bad;
	}
	printf("ok %d rejected %d\n", ok, bad);
	return 0;
}
