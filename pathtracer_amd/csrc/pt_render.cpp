// pt_render -- headless host driver over the C ABI: the reference's main.cpp call sequence
// (initVulkan -> loadModel -> buildBlas/buildTlas -> mainLoop{updateUniformBuffer, raytrace})
// without the window: loads an OBJ, renders N frames, writes the accumulation image.
//
//   pt_render scene.obj [-w 800] [-h 600] [-spp 10] [-frames 100] [-depth 20] [-seed 0] [-tex map.png]
//             (default: the first map_Kd of the OBJ's MTL file; `-tex none` renders untextured)
//             [-o out.pfm|out.ppm|out.png] [-gpus G] [-devices 0,1,..] [-split streams|tiles|strips] [-reduce rccl|host]
//
// Defaults are the reference's constants (VulkanUtils.h:11-15, raytrace.rgen:22-23).
// -gpus G: one host thread and one context per GPU (the rule of pathtracer_amd/multigpu.py): with
// `streams` GPU r renders the global frames r, r+G, ... (frames rounded up to a multiple of G), with `tiles` every GPU renders
// all frames of its own share of the image -- blocks of 32 columns dealt round-robin (pt_set_tile_interleaved; `strips`: one
// column strip per GPU, which balances badly: profiles/r03j_tile_balance.log); each accumulates raw sums (PT_ACCUM_SUM) against its own copy of the scene and BVH.
// The G sum images are then combined on GPU 0:
//   -reduce rccl (default for G > 1)  ONE ncclReduce(sum, fp32, 4*W*H, root 0) over xGMI on the images where they lie
//                (pt_image_device_ptr), a communicator per GPU from ncclCommInitAll -- the only exchange between the GPUs;
//   -reduce host                      the images are read back and added on the host in rank order.  Also the fall-back when
//                RCCL cannot set the communicators up (e.g. the same ordinal named twice in -devices, which is how one
//                GPU rehearses the G-context path: RCCL wants one rank per device).
// Then the mean over all frames (host) and alpha back to 1.  -gpus 1 -reduce rccl runs the same code with one rank.
// -devices names the ordinals (default 0 .. G-1).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "pt_hip.h"

#define CHECK(call)                                                        \
	do                                                                     \
	{                                                                      \
		int rc_ = (call);                                                  \
		if (rc_ != PT_OK)                                                  \
		{                                                                  \
			fprintf(stderr, "%s: error %d: %s\n", #call, rc_, pt_last_error(ctx)); \
			return EXIT_FAILURE;                                           \
		}                                                                  \
	} while (0)

struct Job
{
	const PtSceneData *sd = nullptr;
	std::string texPath;
	uint32_t W = 800, H = 600, seed = 0;
	int spp = 10, depth = 20;
	int frames = 100;       // frames THIS context renders
	uint32_t first = 0, stride = 1;
	uint32_t tile[4] = { 0, 0, 0, 0 };  // w == 0: whole image
	uint32_t ileave[3] = { 0, 0, 0 };   // rank, world, block width of pt_set_tile_interleaved (block width 0: not used)
	int accum = PT_ACCUM_MEAN;
	int device = 0;
	// results
	pt_ctx *ctx = nullptr;  // kept open until the images are combined
	std::vector<float> image;
	PtStats stats = {};
	double seconds = 0.0;
	int rc = EXIT_SUCCESS;
};

// Everything one GPU does up to the last frame: context, scene, acceleration structure, frames.  The context stays open.
static int renderOn(Job &j)
{
	pt_ctx *ctx = pt_create(j.device);
	if (!ctx)
	{
		fprintf(stderr, "pt_create(%d): %s\n", j.device, pt_last_error(nullptr));
		return EXIT_FAILURE;
	}
	const PtSceneData *sd = j.sd;
	CHECK(pt_upload_scene(ctx, pt_scene_vertices(sd), pt_scene_n_vertices(sd), pt_scene_indices(sd), pt_scene_n_indices(sd),
	                      pt_scene_materials(sd), pt_scene_n_materials(sd), pt_scene_mat_indices(sd)));
	if (!j.texPath.empty())
	{
		// diffuse map over Vertex::texCoord (the reference ships textures/viking_room.png but never samples it)
		uint8_t *texels = nullptr;
		uint32_t tw = 0, th = 0;
		CHECK(pt_load_png(j.texPath.c_str(), &texels, &tw, &th));
		CHECK(pt_set_texture(ctx, texels, tw, th));
		pt_free(texels);
	}
	CHECK(pt_build_accel(ctx, PT_ACCEL_BVH4_PLOC));
	CHECK(pt_resize(ctx, j.W, j.H));
	float view[16], proj[16];
	pt_reference_camera(j.W, j.H, view, proj);
	CHECK(pt_set_camera(ctx, view, proj));
	CHECK(pt_set_params(ctx, j.spp, j.depth, 0.001f, 10000.0f, j.seed, j.frames));
	CHECK(pt_set_frame_schedule(ctx, j.first, j.stride));
	CHECK(pt_set_accum_mode(ctx, j.accum));
	if (j.tile[2]) CHECK(pt_set_tile(ctx, j.tile[0], j.tile[1], j.tile[2], j.tile[3]));
	if (j.ileave[2]) CHECK(pt_set_tile_interleaved(ctx, j.ileave[0], j.ileave[1], j.ileave[2]));
	auto t0 = std::chrono::steady_clock::now();
	CHECK(pt_render_frames(ctx, j.frames));
	CHECK(pt_synchronize(ctx));
	j.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	CHECK(pt_get_stats(ctx, &j.stats));
	j.ctx = ctx;
	return EXIT_SUCCESS;
}

// ONE ncclReduce(sum) of the G accumulation images onto GPU 0, in place.  false: RCCL could not be set up
// (the images are untouched; the caller adds them on the host).
static bool reduceWithRccl(std::vector<Job> &jobs, size_t count, double *seconds)
{
	const int G = (int)jobs.size();
	std::vector<int> devs((size_t)G);
	for (int r = 0; r < G; ++r) devs[(size_t)r] = jobs[(size_t)r].device;
	for (int a = 0; a < G; ++a)
		for (int b = a + 1; b < G; ++b)
			if (devs[(size_t)a] == devs[(size_t)b])
			{
				fprintf(stderr, "-reduce rccl: device %d named twice (RCCL wants one rank per device): adding on the host\n", devs[(size_t)a]);
				return false;
			}
	std::vector<ncclComm_t> comms((size_t)G, nullptr);
	ncclResult_t e = ncclCommInitAll(comms.data(), G, devs.data());
	if (e != ncclSuccess)
	{
		fprintf(stderr, "-reduce rccl: ncclCommInitAll: %s: adding on the host\n", ncclGetErrorString(e));
		return false;
	}
	std::vector<void *> img((size_t)G, nullptr);
	bool ok = true;
	for (int r = 0; r < G && ok; ++r) ok = pt_image_device_ptr(jobs[(size_t)r].ctx, &img[(size_t)r]) == PT_OK;
	auto t0 = std::chrono::steady_clock::now();
	if (ok)
	{
		// every context has been synchronised (renderOn): the reduce runs on each device's null stream
		ncclGroupStart();
		for (int r = 0; r < G; ++r)
		{
			if (hipSetDevice(devs[(size_t)r]) != hipSuccess) ok = false;
			e = ncclReduce(img[(size_t)r], img[(size_t)r], count, ncclFloat, ncclSum, 0, comms[(size_t)r], nullptr);
			if (e != ncclSuccess) ok = false;
		}
		e = ncclGroupEnd();
		if (e != ncclSuccess) ok = false;
		for (int r = 0; r < G; ++r)
		{
			if (hipSetDevice(devs[(size_t)r]) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) ok = false;
		}
		if (!ok) fprintf(stderr, "-reduce rccl: ncclReduce failed: %s\n", ncclGetErrorString(e));
	}
	*seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	for (ncclComm_t c : comms)
		if (c) ncclCommDestroy(c);
	if (!ok)
	{
		// a failed collective may have left partial sums behind: nothing to fall back to
		fprintf(stderr, "-reduce rccl: the images are not usable after a failed collective\n");
		exit(EXIT_FAILURE);
	}
	return true;
}

int main(int argc, char **argv)
{
	pt_ctx *ctx = nullptr;
	if (argc < 2)
	{
		fprintf(stderr, "usage: %s scene.obj [-w W] [-h H] [-spp S] [-frames F] [-depth D] [-seed X] [-tex map.png] [-o out.pfm|.ppm|.png] [-gpus G] [-devices a,b,..] [-split streams|tiles|strips] [-reduce rccl|host]\n", argv[0]);
		return EXIT_FAILURE;
	}
	std::string objPath = argv[1], out = "out.pfm", split = "streams", reduce = "";
	Job base;
	int gpus = 1;
	std::vector<int> devices;
	for (int i = 2; i + 1 < argc; i += 2)
	{
		std::string k = argv[i];
		const char *v = argv[i + 1];
		if (k == "-w") base.W = (uint32_t)atoi(v);
		else if (k == "-h") base.H = (uint32_t)atoi(v);
		else if (k == "-spp") base.spp = atoi(v);
		else if (k == "-frames") base.frames = atoi(v);
		else if (k == "-depth") base.depth = atoi(v);
		else if (k == "-seed") base.seed = (uint32_t)strtoul(v, nullptr, 0);
		else if (k == "-o") out = v;
		else if (k == "-tex") base.texPath = v;
		else if (k == "-gpus") gpus = atoi(v);
		else if (k == "-split") split = v;
		else if (k == "-reduce") reduce = v;
		else if (k == "-devices")
		{
			for (const char *p = v; *p;)
			{
				devices.push_back((int)strtol(p, const_cast<char **>(&p), 10));
				if (*p == ',') ++p;
				else if (*p) break;
			}
		}
		else
		{
			fprintf(stderr, "unknown option %s\n", k.c_str());
			return EXIT_FAILURE;
		}
	}
	if (gpus < 1 || (split != "streams" && split != "tiles" && split != "strips") || (!devices.empty() && (int)devices.size() != gpus) ||
	    (split != "streams" && (uint32_t)gpus > base.W) || base.frames < 1 || (!reduce.empty() && reduce != "rccl" && reduce != "host"))
	{
		fprintf(stderr, "bad -gpus / -devices / -split / -frames / -reduce\n");
		return EXIT_FAILURE;
	}
	PtSceneData *sd = nullptr;
	CHECK(pt_load_obj(objPath.c_str(), &sd));
	if (*pt_scene_warnings(sd)) fprintf(stderr, "%s", pt_scene_warnings(sd));
	base.sd = sd;
	// no -tex: the first material's map_Kd, when the MTL names one and the file is there (one texture slot: pt_set_texture)
	for (uint32_t m = 0; base.texPath.empty() && m < pt_scene_n_materials(sd); ++m)
	{
		const char *map = pt_scene_diffuse_map(sd, m);
		if (!*map) continue;
		if (FILE *f = fopen(map, "rb"))
		{
			fclose(f);
			base.texPath = map;
			fprintf(stderr, "diffuse map %s (map_Kd of material %u)\n", map, m);
		}
		else fprintf(stderr, "WARN: map_Kd %s of material %u not found\n", map, m);
	}
	if (base.texPath == "none") base.texPath.clear();
	const uint32_t W = base.W, H = base.H;
	const bool tiles = split != "streams", strips = split == "strips";
	const uint32_t kBlock = 32;  // columns per block of the interleaved schedule (multigpu.TILE_BLOCK_COLUMNS)
	const int perGpu = tiles ? base.frames : (base.frames + gpus - 1) / gpus;
	const int totalFrames = tiles ? base.frames : perGpu * gpus;
	std::vector<Job> jobs((size_t)gpus, base);
	for (int r = 0; r < gpus; ++r)
	{
		Job &j = jobs[(size_t)r];
		j.device = devices.empty() ? r : devices[(size_t)r];
		j.frames = perGpu;
		if (gpus > 1 || reduce == "rccl")
		{
			j.accum = PT_ACCUM_SUM;
			if (tiles && !strips && (uint64_t)gpus * kBlock <= W)
			{
				j.ileave[0] = (uint32_t)r;
				j.ileave[1] = (uint32_t)gpus;
				j.ileave[2] = kBlock;
			}
			else if (tiles)
			{
				const uint32_t b = W / (uint32_t)gpus, extra = W % (uint32_t)gpus, ur = (uint32_t)r;
				j.tile[0] = ur * b + (ur < extra ? ur : extra);
				j.tile[1] = 0;
				j.tile[2] = b + (ur < extra ? 1u : 0u);
				j.tile[3] = H;
			}
			else
			{
				j.first = (uint32_t)r;
				j.stride = (uint32_t)gpus;
			}
		}
	}
	auto t0 = std::chrono::steady_clock::now();
	if (gpus == 1) jobs[0].rc = renderOn(jobs[0]);
	else
	{
		std::vector<std::thread> th;
		for (int r = 0; r < gpus; ++r) th.emplace_back([&jobs, r] { jobs[(size_t)r].rc = renderOn(jobs[(size_t)r]); });
		for (std::thread &t : th) t.join();
	}
	for (const Job &j : jobs)
		if (j.rc != EXIT_SUCCESS) return j.rc;
	ctx = jobs[0].ctx;
	const bool sums = gpus > 1 || reduce == "rccl";  // the contexts hold raw sums over their frames
	double renderSec = 0.0, reduceSec = 0.0;
	unsigned long long rays = 0;
	for (const Job &j : jobs)
	{
		if (j.seconds > renderSec) renderSec = j.seconds;
		rays += j.stats.rays_total;
	}
	const char *how = "one context";
	std::vector<float> &img = jobs[0].image;
	img.resize(4 * (size_t)W * H);
	if (sums)
	{
		const bool viaRccl = reduce != "host" && reduceWithRccl(jobs, 4 * (size_t)W * H, &reduceSec);
		how = viaRccl ? "RCCL reduce" : "host sum";
		CHECK(pt_read_image(ctx, img.data()));
		if (!viaRccl)
		{
			// the reduce(sum) of multigpu.py on the host, in rank order
			auto h0 = std::chrono::steady_clock::now();
			std::vector<float> o(img.size());
			for (int r = 1; r < gpus; ++r)
			{
				if (pt_read_image(jobs[(size_t)r].ctx, o.data()) != PT_OK)
				{
					fprintf(stderr, "pt_read_image: %s\n", pt_last_error(jobs[(size_t)r].ctx));
					return EXIT_FAILURE;
				}
				for (size_t i = 0; i < img.size(); ++i) img[i] += o[i];
			}
			reduceSec = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
		}
		// the mean over all frames; alpha back to 1 (rgen:92)
		const float n = (float)totalFrames;
		for (size_t i = 0; i < img.size(); i += 4)
		{
			img[i] /= n;
			img[i + 1] /= n;
			img[i + 2] /= n;
			img[i + 3] = 1.0f;
		}
		CHECK(pt_write_image(ctx, img.data()));  // GPU 0 holds the final image: the sRGB encode below is the library's
	}
	else CHECK(pt_read_image(ctx, img.data()));
	const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	const PtStats &st = jobs[0].stats;
	printf("%u triangles, BVH4 %u nodes (build %.3f ms); %d frames x %d spp at %ux%u on %d GPU%s (%s, %s): %.3f s render + %.4f s reduce (%.3f s with set-up), %.1f Mrays/s, %.2f spp/px/s\n",
	       st.bvh_tris, st.bvh_nodes, st.build_ms, totalFrames, base.spp, W, H, gpus, gpus > 1 ? "s" : "", gpus > 1 ? split.c_str() : "streams", how,
	       renderSec, reduceSec, sec, rays / (renderSec + reduceSec) / 1e6, totalFrames * base.spp / (renderSec + reduceSec));
	if (st.accel_bytes_uncompacted)  // the reference's log line, VulkanRaytracing.cpp:254-255
		printf("Reduced size of BLAS by %.1f%% from %llu to %llu\n", 100.0 * (double)(st.accel_bytes_uncompacted - st.accel_bytes) / (double)st.accel_bytes_uncompacted,
		       (unsigned long long)st.accel_bytes_uncompacted, (unsigned long long)st.accel_bytes);
	const std::string ext = out.size() > 4 ? out.substr(out.size() - 4) : "";
	if (ext == ".ppm" || ext == ".png")
	{
		// post.frag + the sRGB attachment: the library's device-side encode, whatever the number of GPUs
		std::vector<uint8_t> px(4 * (size_t)W * H);
		CHECK(pt_read_image_srgb8(ctx, px.data()));
		if (ext == ".png") CHECK(pt_write_png(out.c_str(), px.data(), W, H));
		else CHECK(pt_write_ppm(out.c_str(), px.data(), W, H));
	}
	else CHECK(pt_write_pfm(out.c_str(), img.data(), W, H));
	for (Job &j : jobs) pt_destroy(j.ctx);
	pt_scene_free(sd);
	return EXIT_SUCCESS;
}
