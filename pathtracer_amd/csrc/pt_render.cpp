// pt_render -- headless host driver over the C ABI: the reference's main.cpp call sequence
// (initVulkan -> loadModel -> buildBlas/buildTlas -> mainLoop{updateUniformBuffer, raytrace})
// without the window: loads an OBJ, renders N frames, writes the accumulation image.
//
//   pt_render scene.obj [-w 800] [-h 600] [-spp 10] [-frames 100] [-depth 20] [-seed 0] [-tex map.png]
//             [-o out.pfm|out.ppm|out.png]
//
// Defaults are the reference's constants (VulkanUtils.h:11-15, raytrace.rgen:22-23).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

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

int main(int argc, char **argv)
{
	pt_ctx *ctx = nullptr;
	if (argc < 2)
	{
		fprintf(stderr, "usage: %s scene.obj [-w W] [-h H] [-spp S] [-frames F] [-depth D] [-seed X] [-tex map.png] [-o out.pfm|.ppm|.png]\n", argv[0]);
		return EXIT_FAILURE;
	}
	std::string objPath = argv[1], out = "out.pfm", texPath;
	uint32_t W = 800, H = 600, seed = 0;
	int spp = 10, frames = 100, depth = 20;
	for (int i = 2; i + 1 < argc; i += 2)
	{
		std::string k = argv[i];
		const char *v = argv[i + 1];
		if (k == "-w") W = (uint32_t)atoi(v);
		else if (k == "-h") H = (uint32_t)atoi(v);
		else if (k == "-spp") spp = atoi(v);
		else if (k == "-frames") frames = atoi(v);
		else if (k == "-depth") depth = atoi(v);
		else if (k == "-seed") seed = (uint32_t)strtoul(v, nullptr, 0);
		else if (k == "-o") out = v;
		else if (k == "-tex") texPath = v;
		else
		{
			fprintf(stderr, "unknown option %s\n", k.c_str());
			return EXIT_FAILURE;
		}
	}
	PtSceneData *sd = nullptr;
	CHECK(pt_load_obj(objPath.c_str(), &sd));
	if (*pt_scene_warnings(sd)) fprintf(stderr, "%s", pt_scene_warnings(sd));
	ctx = pt_create(0);
	if (!ctx)
	{
		fprintf(stderr, "pt_create: %s\n", pt_last_error(nullptr));
		return EXIT_FAILURE;
	}
	CHECK(pt_upload_scene(ctx, pt_scene_vertices(sd), pt_scene_n_vertices(sd), pt_scene_indices(sd), pt_scene_n_indices(sd),
	                      pt_scene_materials(sd), pt_scene_n_materials(sd), pt_scene_mat_indices(sd)));
	if (!texPath.empty())
	{
		// diffuse map over Vertex::texCoord (the reference ships textures/viking_room.png but never samples it)
		uint8_t *texels = nullptr;
		uint32_t tw = 0, th = 0;
		CHECK(pt_load_png(texPath.c_str(), &texels, &tw, &th));
		CHECK(pt_set_texture(ctx, texels, tw, th));
		pt_free(texels);
	}
	CHECK(pt_build_accel(ctx, PT_ACCEL_BVH4_PLOC));
	CHECK(pt_resize(ctx, W, H));
	float view[16], proj[16];
	pt_reference_camera(W, H, view, proj);
	CHECK(pt_set_camera(ctx, view, proj));
	CHECK(pt_set_params(ctx, spp, depth, 0.001f, 10000.0f, seed, frames));
	auto t0 = std::chrono::steady_clock::now();
	CHECK(pt_render_frames(ctx, frames));
	CHECK(pt_synchronize(ctx));
	double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	PtStats st;
	CHECK(pt_get_stats(ctx, &st));
	printf("%u triangles, BVH4 %u nodes (build %.3f ms); %d frames x %d spp at %ux%u: %.3f s, %.1f Mrays/s, %.2f spp/px/s\n",
	       st.bvh_tris, st.bvh_nodes, st.build_ms, frames, spp, W, H, sec, st.rays_total / sec / 1e6, frames * spp / sec);
	if (st.accel_bytes_uncompacted)  // the reference's log line, VulkanRaytracing.cpp:254-255
		printf("Reduced size of BLAS by %.1f%% from %llu to %llu\n", 100.0 * (double)(st.accel_bytes_uncompacted - st.accel_bytes) / (double)st.accel_bytes_uncompacted,
		       (unsigned long long)st.accel_bytes_uncompacted, (unsigned long long)st.accel_bytes);
	const std::string ext = out.size() > 4 ? out.substr(out.size() - 4) : "";
	if (ext == ".ppm" || ext == ".png")
	{
		std::vector<uint8_t> px(4 * (size_t)W * H);
		CHECK(pt_read_image_srgb8(ctx, px.data()));
		if (ext == ".png") CHECK(pt_write_png(out.c_str(), px.data(), W, H));
		else CHECK(pt_write_ppm(out.c_str(), px.data(), W, H));
	}
	else
	{
		std::vector<float> px(4 * (size_t)W * H);
		CHECK(pt_read_image(ctx, px.data()));
		CHECK(pt_write_pfm(out.c_str(), px.data(), W, H));
	}
	pt_scene_free(sd);
	pt_destroy(ctx);
	return EXIT_SUCCESS;
}
