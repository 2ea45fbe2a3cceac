/*
 * pt_hip.h -- C ABI of libpt_hip.so: the MI355X (gfx950) replacement for the reference's
 * Vulkan KHR ray-tracing pipeline.
 *
 * The reference (clarlzx/PathTracer) has no plugin/FFI interface; its seam is the public
 * surface of `class VulkanRaytracing` (VulkanRaytracing.h:88-117) as driven by
 * main.cpp:272-317,433-435,1411-1412, plus the host-side scene/camera code of main.cpp that
 * feeds it.  Every entry point below names the reference interface it replaces.
 *
 * Conventions: plain C types only; every int-returning function returns PT_OK (0) or a
 * negative PT_E* code, with a human-readable message available from pt_last_error();
 * nothing throws across the boundary; a context is used from one host thread at a time;
 * one context per GPU.  All matrices are column-major float[16] exactly as glm::mat4.
 * Device work is asynchronous on the context's stream unless the call returns data to the
 * host.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK 0
#define PT_EINVAL (-1)   /* bad argument / call order */
#define PT_EHIP (-2)     /* HIP runtime error */
#define PT_ENOMEM (-3)
#define PT_EIO (-4)      /* file not found / parse error (host seam) */
#define PT_ELIMIT (-5)   /* a documented size limit was exceeded */

#define PT_NO_HIT 0xFFFFFFFFu

/* Byte-identical to `struct Vertex` (main.cpp:60-65, shaders/raycommon.glsl:11-17): 44 B. */
typedef struct PtVertex
{
	float pos[3];
	float normal[3];
	float color[3];
	float texCoord[2];
} PtVertex;

/* Byte-identical to `struct Material` (main.cpp:54-58, shaders/raycommon.glsl:19-23): 24 B. */
typedef struct PtMaterial
{
	float diffuse[3];
	float emission[3];
} PtMaterial;

/* Closest-hit record: what traceRayEXT hands to raytrace.rchit (gl_PrimitiveID +
 * hitAttributeEXT vec2 attribs) plus the hit distance. 16 B. */
typedef struct PtHit
{
	float t;
	float u, v;     /* barycentric weights of triangle vertex 1 and 2 */
	uint32_t prim;  /* PT_NO_HIT on a miss */
} PtHit;

typedef struct PtStats
{
	uint64_t rays_total;        /* closest-hit queries since the last reset (one per traceRayEXT) */
	uint64_t rays_last_frame;   /* mean over the frames of the most recent launch (several frames per launch) */
	uint64_t paths_last_frame;  /* W*H*spp_per_frame */
	uint64_t frames_rendered;
	double extend_ms_total;     /* HIP-event time of the traversal kernels (only with PT_PROFILE_KERNELS) */
	uint64_t extend_launches;
	double shade_ms_total;
	uint64_t shade_launches;
	uint32_t bvh_nodes;         /* wide nodes in the acceleration structure */
	uint32_t bvh_depth;         /* wide-node levels */
	uint32_t bvh_tris;
	uint32_t reserved;
	double build_ms;            /* last pt_build_accel: device time from the call's first launch (scratch allocation included) to the tree's last kernel; compaction, quad leaves and entry node are outside */
	/* Traversal work counters, only collected under PT_PROFILE_TRAVERSAL (instrumented kernel): */
	uint64_t trav_nodes;        /* wide nodes fetched (the entry node of the persistent pipeline counts as one per ray) */
	uint64_t trav_tris;         /* ray/triangle tests */
	uint64_t trav_hits;         /* rays that found a hit */
	uint64_t trav_leaf_visits;  /* leaves entered (lane level) */
	uint64_t trav_node_steps;   /* wave-level node steps of the traversal loop: trav_nodes / (64 * trav_node_steps) = lane
	                               utilisation when the scene has no entry node (whose visits happen outside the loop) */
	uint64_t trav_leaf_steps;   /* wave-level leaf steps (persistent pipeline only, like the two above) */
	/* cmdCompactBlas's log line (VulkanRaytracing.cpp:247-256): bytes of the bottom-level structure (nodes, leaf
	 * triangles, codes, order) after compaction and as the builder had to size it before knowing the node count */
	uint64_t accel_bytes;
	uint64_t accel_bytes_uncompacted;
	/* Two-triangle leaves stored as quads for the persistent pipeline (0: the scene is not made of quad faces, or
	 * PT_NO_QUADS is set); they cost 192 bytes each plus one more copy of the node array, not counted above */
	uint64_t quad_leaves;
} PtStats;

typedef struct pt_ctx pt_ctx;

/* ---- context -------------------------------------------------------------------------- */

/* VulkanRaytracing::setContext (VulkanRaytracing.cpp:9-12) + device selection
 * (VulkanDevice.cpp:16-43).  Returns NULL on failure (pt_last_error(NULL) tells why). */
pt_ctx *pt_create(int device_ordinal);

/* VulkanRaytracing::cleanup (VulkanRaytracing.cpp:915-935). */
void pt_destroy(pt_ctx *ctx);

/* Replaces the C++ exceptions of the reference (main.cpp:1628-1636).  ctx may be NULL for
 * errors of pt_create and of the context-free host-seam functions. */
const char *pt_last_error(pt_ctx *ctx);

/* Optional: run on a caller-owned hipStream_t (passed as void*) instead of the context's
 * own stream; NULL restores the internal stream. */
int pt_set_stream(pt_ctx *ctx, void *hip_stream);
int pt_synchronize(pt_ctx *ctx);

/* ---- scene + acceleration structure ---------------------------------------------------- */

/* VulkanRaytracing::buildTlas / createInstancesBuffer (VulkanRaytracing.cpp:382-527): the instances of the
 * bottom-level structures.  transform = VkTransformMatrixKHR (object-to-world, 3x4 row-major); custom_index selects
 * the Model the instance shows AND its bottom-level structure (raytrace.rchit:27 model.m[gl_InstanceCustomIndexEXT];
 * createInstancesBuffer pairs instanceCustomIndex = i with m_blases[i]) -- 0 with pt_upload_scene, < n_models with
 * pt_upload_models (checked by pt_build_accel); mask is kept for layout only (every instance is visible, as with
 * the reference's 0xFF).  n = 0 (default) is the reference's own instance list: one identity instance per Model --
 * for one Model that is a single-level scene and costs nothing.
 * Semantics (Vulkan spec + raytrace.rchit:45-46): the ray is taken into object space with the world-to-object
 * matrix (glm::inverse of the 4x4, fp32 cofactors -- this build's definition of gl_WorldToObjectEXT) WITHOUT
 * renormalising the direction, so hit distances compare across instances; ties go to the lower instance index;
 * hit position = ObjectToWorld * p, normal = normalize(n * WorldToObject).  Call before pt_build_accel (which
 * builds the top-level BVH4 over the instances' world boxes).  Instanced scenes render on
 * PT_PIPELINE_PERSISTENT. */
typedef struct PtInstance
{
	float transform[12];
	uint32_t custom_index;
	uint32_t mask;
} PtInstance;
int pt_set_instances(pt_ctx *ctx, const PtInstance *instances, uint32_t n);

/* PathTracer::createModelBuffers (main.cpp:1192-1279): uploads the four scene buffers the
 * `Model` address table points at.  n_indices must be a multiple of 3; mat_indices has
 * n_indices/3 entries, each < n_materials.  Host pointers are copied; nothing is borrowed. */
int pt_upload_scene(pt_ctx *ctx, const PtVertex *vertices, uint32_t n_vertices, const uint32_t *indices,
                    uint32_t n_indices, const PtMaterial *materials, uint32_t n_materials,
                    const uint32_t *mat_indices);

/* Several Models in one scene: the array `model.m[]` of raytrace.rchit:21,27 (struct Model, main.cpp:46-52 /
 * raycommon.glsl:25; the reference fills in exactly one, main.cpp:1199-1208, and builds one BLAS and one instance,
 * VulkanRaytracing.cpp:158, 385).  Replaces pt_upload_scene; Model m gets its own bottom-level structure at
 * pt_build_accel and is shown by every instance whose custom_index is m.  Every model obeys pt_upload_scene's rules
 * on its own (indices < its n_vertices, mat_indices < its n_materials); gl_PrimitiveID is local to the model as in
 * Vulkan (PtStats / debug entry points number primitives through all models in order).  n_models >= 1; at most
 * 2^28-1 triangles in all.  A scene of several models renders on PT_PIPELINE_PERSISTENT. */
typedef struct PtModelDesc
{
	const PtVertex *vertices;
	uint32_t n_vertices;
	const uint32_t *indices;
	uint32_t n_indices;
	const PtMaterial *materials;
	uint32_t n_materials;
	const uint32_t *mat_indices;
} PtModelDesc;
int pt_upload_models(pt_ctx *ctx, const PtModelDesc *models, uint32_t n_models);

/* Diffuse map of the scene (BASELINE config 3 "textured").  No counterpart in the reference's
 * shaders: it ships textures/viking_room.png and carries Vertex::texCoord (main.cpp:1166-1176, V flipped) through to
 * raytrace.rchit but never samples it.  Defined here: hitValue (raytrace.rchit:54) = material diffuse x texel, texel =
 * bilinear, repeat-addressed lookup at the barycentrically interpolated texCoord in the sRGB-decoded image
 * (pt_srgb8_to_linear), row 0 = v 0.  This map covers every primitive of the scene (the reference has one Model;
 * pt_set_model_texture: one per Model); rgba8 = NULL, w = h = 0 removes it.  May be called at any time before a frame; pt_upload_scene keeps it. */
int pt_set_texture(pt_ctx *ctx, const uint8_t *rgba8, uint32_t width, uint32_t height);

/* One diffuse map per Model (raytrace.rchit:27: every per-Model buffer is reached through
 * model.m[gl_InstanceCustomIndexEXT]; main.cpp:1192-1251 fills in one Model).  `model` indexes the array given to
 * pt_upload_models (0 after pt_upload_scene); a hit on a primitive of that Model multiplies the material's diffuse colour
 * with this map as pt_set_texture defines, a Model without a map keeps the plain colour.  rgba8 = NULL, w = h = 0 removes
 * the Model's map.  Per-Model maps and the scene-wide map replace each other (the call that comes last wins); a new
 * pt_upload_scene / pt_upload_models drops the per-Model maps (their primitive ranges are the old scene's). */
int pt_set_model_texture(pt_ctx *ctx, uint32_t model, const uint8_t *rgba8, uint32_t width, uint32_t height);

#define PT_ACCEL_BVH4 0         /* LBVH (Morton + LDS radix sort) collapsed to a quantised BVH4 */
#define PT_ACCEL_BRUTE_FORCE 1  /* no hierarchy: every ray tests every triangle (parity/debug mode) */
#define PT_ACCEL_BVH4_PLOC 2    /* same sort, BVH2 by parallel locally-ordered clustering (radius 16), same BVH4 */
#define PT_ACCEL_LEAF_SIZE(n) ((n) << 8) /* OR into flags: at most n (1..8) triangles per leaf; 0 = default */
#define PT_ACCEL_NO_COMPACTION 0x1000    /* buildBlas(..., allowCompaction = false), VulkanRaytracing.cpp:150,170 */
#define PT_ACCEL_NO_ENTRY_NODE 0x2000    /* no flattened tree top: every ray of the persistent pipeline starts at the root (same
                                            images; node-visit statistics then equal the other pipelines') */
/* Leaf triangle records: by default three copies, one per dominant ray axis with the coordinates rotated for it (the
 * traversal then spends nothing per triangle on the permutation) -- except for ONE Model of PT_ONE_PLANE_TRIS triangles or
 * more, whose traversal waits for memory anyway and which keeps the plain (x, y, z) copy only (a third of the bytes to
 * build and to keep; traversal measured 1-4 % FASTER on such scenes, 1 % slower on cache-resident ones).  These two flags force one or the other.  Same hits, same images either way. */
#define PT_ACCEL_TRI_PLANES_1 0x4000
#define PT_ACCEL_TRI_PLANES_3 0x8000
#define PT_ONE_PLANE_TRIS 2000000u

/* VulkanRaytracing::buildBlas + buildTlas (VulkanRaytracing.cpp:147-262,448-527): what the
 * Vulkan driver did inside vkCmdBuildAccelerationStructuresKHR.  One BLAS, one identity
 * instance.  Blocks until the structure is built (the reference waits with vkQueueWaitIdle).
 * Compaction (cmdCompactBlas, VulkanRaytracing.cpp:300-380; on by default as allowCompaction is): the node array is
 * re-allocated at its exact size after the build; PtStats reports both sizes. */
int pt_build_accel(pt_ctx *ctx, int flags);

/* ---- frame state ---------------------------------------------------------------------- */

/* PathTracer::createColorResources (main.cpp:908-931) + updateDescriptorSets + resetFrame
 * (main.cpp:433-435): (re)allocates the RGBA32F accumulation image and the path queues. */
int pt_resize(pt_ctx *ctx, uint32_t width, uint32_t height);

/* PathTracer::updateUniformBuffer (main.cpp:1465-1480): UniformBufferObject.view/.proj.  The
 * two inverses raytrace.rgen:44-46 recomputes per sample are hoisted here (glm algorithm).
 * A camera that differs from the current one restarts the accumulation (frame = -1): what updateFrame's comment
 * promises and its TODO leaves out (VulkanRaytracing.cpp:903-908); setting the same matrices again is a no-op. */
int pt_set_camera(pt_ctx *ctx, const float view[16], const float proj[16]);

/* The reference's compile-time constants made parameters: MAX_SAMPLES (raytrace.rgen:22,
 * 10), MAX_RAY_DEPTH (:23, 20), tMin/tMax (:51-52, 0.001/10000), maxRTFrames
 * (VulkanUtils.h:15, 1000; <=0 = unlimited) and an explicit `seed` that replaces the
 * non-reproducible int(clockARB()) of raytrace.rgen:28: val1 = seed + global frame index. */
int pt_set_params(pt_ctx *ctx, int spp_per_frame, int max_depth, float tmin, float tmax, uint32_t seed,
                  int max_frames);

/* Multi-GPU sample-stream split (no reference counterpart: the reference is single-device).
 * The k-th pt_render_frame after a reset renders GLOBAL frame first + k*stride (seeding and
 * the frame-0 pixel-centre rule use the global index).  Default first=0, stride=1. */
int pt_set_frame_schedule(pt_ctx *ctx, uint32_t first, uint32_t stride);

/* Current extent of the accumulation image (0 x 0 before pt_resize; pt_load_checkpoint may change it). */
int pt_image_extent(pt_ctx *ctx, uint32_t *width, uint32_t *height);

/* Image-tile split: restrict pt_render_frame to the w x h rectangle at (x0, y0);
 * pixels outside keep their value (0 after pt_resize), so per-GPU tiles combine with the same sum-reduce as
 * sample streams.  Every pixel's value is independent of the tiling (seeds use the full-image pixel index).
 * w = h = 0 restores the whole image; pt_resize does too. */
int pt_set_tile(pt_ctx *ctx, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h);
/* The same split with the load spread evenly (no reference counterpart): the image's columns in blocks of block_w, dealt
 * round-robin -- rank r of `world` renders the blocks r, r + world, ... over all rows.  A column strip per GPU leaves the
 * GPUs whose strip shows sky idle (measured on configs[4]: the slowest of 8 strips takes 2.1x the mean, profiles/
 * r03j_tile_balance.log); blocks of 16-64 columns even that out.  Same image, same reduce.  pt_set_tile(0,0,0,0) or
 * pt_resize restore the whole image. */
int pt_set_tile_interleaved(pt_ctx *ctx, uint32_t rank, uint32_t world, uint32_t block_w);

#define PT_ACCUM_MEAN 0  /* running mean, mix(old,new,1/(frame+1)): raytrace.rgen:88-95 */
#define PT_ACCUM_SUM 1   /* raw per-frame sums (exact, order-independent multi-GPU reduce) */
int pt_set_accum_mode(pt_ctx *ctx, int mode);

/* How one dispatch is executed on the device (no reference counterpart; results are bit-identical):
 * PERSISTENT (default) = one kernel per frame; FUSED = one persistent kernel per bounce (traverse + shade +
 * compact, lanes refilled in flight); SPLIT = the classic wavefront pair k_extend / k_shade with the hit
 * records in HBM.  PT_ACCEL_BRUTE_FORCE always runs on SPLIT. */
#define PT_PIPELINE_FUSED 0
#define PT_PIPELINE_SPLIT 1
#define PT_PIPELINE_PERSISTENT 2 /* one persistent kernel per FRAME: waves generate camera rays, keep their paths
                                   * from first ray to last (survivors in wave-private rings), no per-bounce launches */
int pt_set_pipeline(pt_ctx *ctx, int mode);

/* VulkanRaytracing::raytrace (VulkanRaytracing.cpp:868-901): ++frame, dispatch W x H x 1,
 * accumulate; a no-op once frame >= max_frames.  Asynchronous. */
int pt_render_frame(pt_ctx *ctx);
/* n consecutive pt_render_frame calls (convenience for the host loop, main.cpp:320-329). */
int pt_render_frames(pt_ctx *ctx, int n);

/* pt_render_frames for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc.  Same frames, same image, same frame counter and
 * statistics as pt_render_frames(ctx, n) on PT_PIPELINE_PERSISTENT, but every buffer pt_render_frames allocates on demand
 * (LCG skip-ahead table, per-path state and radiance, wave-private survivor rings) is a slice of a file-scope device array
 * of 16 MiB + 34 KiB that each GPU the library runs on holds until the process ends; the call makes no hipMalloc / hipFree /
 * event creation and no host allocation.  Scene, acceleration structure, camera and the image (pt_resize) are the caller's
 * inputs and outputs as before.  Limits (PT_ELIMIT beyond them, nothing rendered): ceil(pixels of the image or tile / 64)
 * * 64 * spp_per_frame <= 2^18 paths per frame (e.g. 512 x 512 x 1 spp, 256 x 256 x 4 spp; larger images tile by tile with
 * pt_set_tile), spp_per_frame <= 256.  PT_PROFILE_KERNELS is ignored.  All contexts of one GPU share that GPU's array:
 * calls that overlap on one GPU (two streams, two host threads) race; calls that follow each other, from any context,
 * do not depend on what the previous one left behind.  The grid is 1 024 waves (4 per CU, a quarter of pt_render_frames'
 * occupancy: the rings of more do not fit the array), so the call is slower per frame than pt_render_frames.
 * Under stream capture (pt_set_stream + hipStreamBeginCapture; verified in global capture mode,
 * tests/test_zz_no_alloc.py::test_noalloc_under_hip_graph_capture): the call records five kernel launches per batch of
 * frames and nothing else.  Frame numbers, seed words and the accumulation position are kernel ARGUMENTS computed from
 * the context's frame counter when the call is RECORDED (the counter advances then, not at replay): a graph replays
 * exactly the frames it recorded -- it is not a progressive loop.  To accumulate with graphs, record one call per launch
 * position (graph k renders frames kn .. kn+n-1 and continues the image graph k-1 left) and launch them in order. */
int pt_render_frames_noalloc(pt_ctx *ctx, int n);

/* VulkanRaytracing::resetFrame (VulkanRaytracing.cpp:910-913): frame = -1. */
int pt_reset_accumulation(pt_ctx *ctx);
/* m_pcRay.frame (VulkanRaytracing.h:35-38): -1 before the first dispatch. */
int pt_frame_index(pt_ctx *ctx);

/* The image post.frag samples (shaders/post.frag:12, main.cpp:908-931): W*H*4 floats, row 0
 * = top of the picture.  Synchronises. */
int pt_read_image(pt_ctx *ctx, float *rgba32f);
/* Device address of that image (for an RCCL reduce or zero-copy consumers). */
int pt_image_device_ptr(pt_ctx *ctx, void **dev_ptr);
/* Overwrites the accumulation image from device memory (e.g. a reduced multi-GPU result). */
int pt_write_image_from_device(pt_ctx *ctx, const void *dev_rgba32f);
/* The same from host memory, the counterpart of pt_read_image (m_colorImage, main.cpp:908-931, is device-only in the
 * reference): pt_render writes the mean of G GPUs' sums back so that the sRGB encode is the library's.  Synchronises. */
int pt_write_image(pt_ctx *ctx, const float *rgba32f);

/* Accumulation checkpoint: the RGBA32F image plus everything the frame counter's meaning depends on (frame
 * index, extent, sampling parameters, seed, frame schedule, accumulation mode).  The reference keeps this
 * state only on the GPU and loses it on exit or resize (VulkanRaytracing.cpp:904-913, main.cpp:435).  Load
 * into a context whose scene, acceleration structure and camera are set: the next pt_render_frame continues
 * with frame + 1 and the final image is bit-identical to an uninterrupted run. */
int pt_save_checkpoint(pt_ctx *ctx, const char *path);
int pt_load_checkpoint(pt_ctx *ctx, const char *path);

/* post.frag + the B8G8R8A8_SRGB attachment (VulkanSwapchain.cpp:8): linear RGBA32F -> 8-bit
 * sRGB RGBA, W*H*4 bytes.  Synchronises. */
int pt_read_image_srgb8(pt_ctx *ctx, uint8_t *rgba8);

#define PT_PROFILE_KERNELS 1    /* bracket traversal/shade launches with HIP events */
#define PT_PROFILE_TRAVERSAL 2  /* run the instrumented traversal kernel (counts node fetches / triangle tests) */
int pt_set_profiling(pt_ctx *ctx, int flags);
int pt_get_stats(pt_ctx *ctx, PtStats *out);
int pt_reset_stats(pt_ctx *ctx);

/* ---- parity / inspection entry points (test + bench support, no reference counterpart) -- */

/* tea()/lcg() of shaders/random.glsl:27-51 evaluated ON THE DEVICE: out_tea[i] =
 * tea(v0[i], v1[i]); out_lcg[i*n_steps + k] = k-th lcg output (24 bit) from state out_tea[i];
 * out_unit[i*3..] = rnd_unit_vector drawn from state out_tea[i]. */
int pt_debug_rng(pt_ctx *ctx, uint32_t n, const uint32_t *v0, const uint32_t *v1, uint32_t n_steps,
                 uint32_t *out_tea, uint32_t *out_lcg, float *out_unit);
/* Self-check of the arithmetic specification on the device: n pseudo-random operand sets (exponents sweeping 2^+-40 for
 * divisions, 2^+-60 for square roots, the ends included) through the in-range instruction sequences the path kernel
 * issues for IEEE division and sqrt, compared bit for bit with the plain operators.  mismatches[0] quotients,
 * [1] roots, [2] reciprocal roots -- all zero or the build is not what DESIGN.md section 4 says it is; [3] = operand sets
 * the device went through (= n). */
int pt_debug_math(pt_ctx *ctx, uint64_t n, uint32_t seed, uint64_t mismatches[4]);
/* Closest hits of the frame-0 primary rays (jitter = pixel centre), W*H records. */
int pt_debug_primary_hits(pt_ctx *ctx, PtHit *hits);
/* Closest hits for caller rays: rays = n * 6 floats (origin, direction). */
int pt_debug_trace(pt_ctx *ctx, const float *rays, uint32_t n, float tmin, float tmax, PtHit *hits);
/* Morton codes per primitive, the sorted primitive order, and the centroid bounds. */
int pt_debug_morton(pt_ctx *ctx, uint32_t *codes, uint32_t *order, float cmin[3], float cmax[3]);
/* Copies the wide BVH out: nodes (64 B each) and leaf triangles (48 B each). Pass NULL to
 * query the counts only. */
int pt_debug_read_bvh(pt_ctx *ctx, void *nodes, uint32_t *n_nodes, void *tris, uint32_t *n_tris);

/* ---- host seam (context-free) ------------------------------------------------------------ */

typedef struct PtSceneData PtSceneData;

/* PathTracer::loadModel (main.cpp:1101-1190) over an OBJ/MTL parser that follows
 * tinyobjloader v1.0.6 semantics (the reference's vendored external/tinyobjloader).
 * Extension: faces without normals get flat geometric normals (the reference has UB there). */
int pt_load_obj(const char *obj_path, PtSceneData **out);
/* The second half of PathTracer::loadModel alone (main.cpp:1122-1189), for callers that already hold what
 * tinyobj::LoadObj (main.cpp:1117) returned: attrib_t::vertices / normals / texcoords (3, 3, 2 floats per element),
 * the shapes' mesh.indices concatenated in shape order as tinyobj::index_t triples { vertex_index, normal_index,
 * texcoord_index } (n_indices corners, three per triangle; -1 = absent), mesh.material_ids likewise (n_indices / 3
 * entries, -1 = none -> the grey fallback) and the .mtl materials as Material{diffuse, emission}.  Same result as
 * pt_load_obj on the file those arrays came from (tests/test_host_seam.py). */
int pt_scene_from_tinyobj(const float *vertices, uint32_t n_vertices, const float *normals, uint32_t n_normals,
                          const float *texcoords, uint32_t n_texcoords, const int32_t *indices, uint32_t n_indices,
                          const int32_t *material_ids, const PtMaterial *materials, uint32_t n_materials,
                          PtSceneData **out);
void pt_scene_free(PtSceneData *s);
uint32_t pt_scene_n_vertices(const PtSceneData *s);
uint32_t pt_scene_n_indices(const PtSceneData *s);
uint32_t pt_scene_n_materials(const PtSceneData *s);
const PtVertex *pt_scene_vertices(const PtSceneData *s);
const uint32_t *pt_scene_indices(const PtSceneData *s);
const PtMaterial *pt_scene_materials(const PtSceneData *s);
const uint32_t *pt_scene_mat_indices(const PtSceneData *s);
const char *pt_scene_warnings(const PtSceneData *s);
/* tinyobj::material_t::diffuse_texname (the MTL file's `map_Kd [-options] file`, tiny_obj_loader.h:1204-1208) of one
 * material, relative names resolved against the OBJ's directory; "" when the material has none or the index is out of
 * range.  The reference's loadModel never reads it (main.cpp:1125-1133 takes only diffuse and emission), which is why
 * models/viking_room renders untextured there; pt_render and HipRaytracing.setTextureFromScene hand it to pt_set_texture
 * (SURVEY.md 8(f) row 3).  The pointer lives as long as the PtSceneData. */
const char *pt_scene_diffuse_map(const PtSceneData *s, uint32_t material);

/* glm::lookAt / glm::perspective (RH, -1..1 depth: SURVEY.md section 0) / glm::inverse as the
 * reference's host uses them, and the fixed camera of main.cpp:1465-1480 (incl. proj[1][1]*=-1). */
void pt_look_at(const float eye[3], const float center[3], const float up[3], float out[16]);
void pt_perspective(float fovy_radians, float aspect, float z_near, float z_far, float out[16]);
void pt_inverse4(const float m[16], float out[16]);
void pt_reference_camera(uint32_t width, uint32_t height, float view[16], float proj[16]);

/* Writers for the accumulation image (the reference only presents to a window). */
int pt_write_pfm(const char *path, const float *rgba32f, uint32_t width, uint32_t height);
int pt_write_ppm(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height);
/* 8-bit RGBA PNG (stored deflate blocks, sRGB chunk) of what the sRGB swapchain would show
 * (VulkanSwapchain.cpp:8, shaders/post.frag:9-13): feed it pt_read_image_srgb8's output. */
int pt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height);

/* stb_image's role in the reference's lineage (external/stb, textures/viking_room.png): decodes an 8-bit
 * non-interlaced PNG (grey, grey+alpha, RGB, RGBA, palette) to RGBA8 with an own inflate.  *out_rgba8 is
 * malloc'ed: release it with pt_free. */
int pt_load_png(const char *path, uint8_t **out_rgba8, uint32_t *width, uint32_t *height);
void pt_free(void *p);
/* IEC 61966-2-1 decode of n 8-bit values: c = b/255; c <= 0.04045 ? c/12.92 : pow((c+0.055)/1.055, 2.4), evaluated
 * in double and rounded once to float (a 256-entry table). */
void pt_srgb8_to_linear(const uint8_t *srgb8, uint32_t n, float *out);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
