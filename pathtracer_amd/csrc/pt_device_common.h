// Device-side building blocks shared by the HIP kernels (gfx950 only).
//
// ARITHMETIC CONTRACT.  Everything here is compiled with -ffp-contract=off: IEEE binary32,
// round-to-nearest-even, correctly rounded '/' and sqrt (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt), fp32 denormals kept, and a fused multiply-add
// ONLY where __builtin_fmaf is written out.  The expression order of every function that
// feeds a result (RNG, camera ray, triangle test, shading) is part of the specification in
// DESIGN.md section 4, which the CPU oracle implements independently; the two agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pt_hip.h"

#define PT_WAVE 64

struct float3x
{
	float x, y, z;
};

// ------------------------------------------------------------------------------ RNG ----
// shaders/random.glsl:27-41
__device__ __forceinline__ uint32_t pt_tea(uint32_t val0, uint32_t val1)
{
	uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
	for (int n = 0; n < 16; n++)
	{
		s0 += 0x9e3779b9u;
		v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
		v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
	}
	return v0;
}

// shaders/random.glsl:45-51
__device__ __forceinline__ uint32_t pt_lcg(uint32_t &prev)
{
	prev = 1664525u * prev + 1013904223u;
	return prev & 0x00FFFFFFu;
}

// shaders/random.glsl:54-57 (exact: 24-bit integer scaled by 2^-24)
__device__ __forceinline__ float pt_rnd(uint32_t &prev)
{
	return (float)pt_lcg(prev) * (1.0f / 16777216.0f);
}

// sin/cos of 2*pi*u, u = k/2^24 in [0,1): exact quadrant reduction, then the Taylor
// polynomials of sin(pi/2 x), cos(pi/2 x) on |x| <= 1/2 (Horner, fused).  This DEFINES the
// implementation-defined GLSL sin/cos of random.glsl:59-64 for this build.
__device__ __forceinline__ void pt_sincos_2pi(float u, float &s_out, float &c_out)
{
	const float S0 = 1.57079632679489662f, S1 = -0.645964097506246254f, S2 = 0.0796926262461670451f,
	            S3 = -0.00468175413531868810f, S4 = 0.000160441184787359822f;
	const float C1 = -1.23370055013616983f, C2 = 0.253669507901048014f, C3 = -0.0208634807633529609f,
	            C4 = 0.000919260274839426580f;
	float t = u * 4.0f;
	int q = (int)(t + 0.5f);
	float x = t - (float)q;
	float x2 = x * x;
	float sp = __builtin_fmaf(x2, S4, S3);
	sp = __builtin_fmaf(x2, sp, S2);
	sp = __builtin_fmaf(x2, sp, S1);
	sp = __builtin_fmaf(x2, sp, S0);
	float s = x * sp;
	float cp = __builtin_fmaf(x2, C4, C3);
	cp = __builtin_fmaf(x2, cp, C2);
	cp = __builtin_fmaf(x2, cp, C1);
	float c = __builtin_fmaf(x2, cp, 1.0f);
	bool swap = q & 1;
	float ss = swap ? c : s;
	float cc = swap ? s : c;
	// q&3: 0 -> (s, c); 1 -> (c, -s); 2 -> (-s, -c); 3 -> (-c, s)
	s_out = (q & 2) ? -ss : ss;
	c_out = (((q + 1) & 2) ? -cc : cc);
}

// ---- IEEE sqrt, reciprocal and division, in-range forms ----
// hipcc expands __builtin_sqrtf into: scale x by 2^32 if it is tiny, v_sqrt_f32, one correction step to the correctly
// rounded root (the neighbours one ulp below and above are tried against x with an fma each), scale back, pass 0 / inf
// through -- 16 instructions; and n / d into v_div_scale x2, v_rcp_f32, two Newton steps on the reciprocal, three on
// the quotient, v_div_fmas, v_div_fixup -- 11.  For operands far from the exponent limits the scaling, the fix-up and the
// class check are identities; the functions below are the remaining instructions, in the same order, so they return the
// same bits.  Callers use them only when a wave-uniform range check has passed for every active lane, and fall back to the
// plain operators otherwise.
__device__ __forceinline__ float pt_sqrt_in_range(float x)
{
	float s = __builtin_amdgcn_sqrtf(x);
	float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
	float rb = __builtin_fmaf(-below, s, x);
	float ra = __builtin_fmaf(-above, s, x);
	float r = (0.0f >= rb) ? below : s;
	return (0.0f < ra) ? above : r;
}
// n / d for several numerators and one denominator: the reciprocal part of the expansion is computed once
struct PtSharedDivisor
{
	float d, rc;
	__device__ __forceinline__ explicit PtSharedDivisor(float den) : d(den)
	{
		rc = __builtin_amdgcn_rcpf(den);
		float e = __builtin_fmaf(-den, rc, 1.0f);
		rc = __builtin_fmaf(e, rc, rc);
	}
	__device__ __forceinline__ float quotient(float n) const
	{
		float q = n * rc;
		float e = __builtin_fmaf(-d, q, n);
		q = __builtin_fmaf(e, rc, q);
		e = __builtin_fmaf(-d, q, n);
		return __builtin_fmaf(e, rc, q);
	}
};
// true when |a|, |b|, |c| all lie in [2^-40, 2^40] for every active lane of the wave (quotients of two such numbers stay
// 2^46 away from the exponent limits: no scaling in v_div_scale, nothing to fix in v_div_fixup; a NaN fails the test)
__device__ __forceinline__ bool pt_all_in_division_range(float a, float b, float c)
{
	const float lo = 9.09494702e-13f, hi = 1.09951163e12f;
	const float x = fabsf(a), y = fabsf(b), z = fabsf(c);
	const bool ok = x >= lo && x <= hi && y >= lo && y <= hi && z >= lo && z <= hi;
	return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}
// true when |x| lies in [2^-60, 2^60] for every active lane of the wave
__device__ __forceinline__ bool pt_all_in_range(float x)
{
	const float a = fabsf(x);
	const bool ok = a >= 8.67361738e-19f && a <= 1.15292150e18f;
	return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
}

// shaders/random.glsl:59-64 (theta draw first, then z).  kInRange: called by all lanes of a full shading pass (k_paths):
// the in-range sqrt when it applies to every one of them.
template <bool kInRange = false>
__device__ __forceinline__ float3x pt_rnd_unit_vector(uint32_t &prev)
{
	float u = pt_rnd(prev);
	float z = pt_rnd(prev) * 2.0f - 1.0f;
	float x = 1.0f - z * z;
	float r;
	if (kInRange && pt_all_in_range(x)) r = pt_sqrt_in_range(x);
	else r = __builtin_sqrtf(x);
	float s, c;
	pt_sincos_2pi(u, s, c);
	return { r * c, r * s, z };
}

// glm-style normalize: v * (1 / sqrt((x*x + y*y) + z*z))
template <bool kInRange = false>
__device__ __forceinline__ float3x pt_normalize(float3x v)
{
	float x = (v.x * v.x + v.y * v.y) + v.z * v.z;
	float inv;
	if (kInRange && pt_all_in_range(x)) inv = PtSharedDivisor(pt_sqrt_in_range(x)).quotient(1.0f);
	else inv = 1.0f / __builtin_sqrtf(x);
	return { v.x * inv, v.y * inv, v.z * inv };
}

// ---------------------------------------------------------------------- camera ray ----
struct PtCamera
{
	float invView[16];
	float invProj[16];
	// the ray-independent parts of raytrace.rgen:40-46, evaluated once on the host with the same float expressions
	// (pt_camera_fill_constants): origin = invView * (0,0,0,1); projConst[k] = P[8+k]*1 + P[12+k]*1; viewZero[k] = V[12+k]*0
	float origin[3];
	float projConst[3];
	float viewZero[3];
	float pad[3];
};
static_assert(sizeof(PtCamera) == 44 * 4, "PtCamera is 44 floats");

// Host and device: the constant sub-expressions of pt_camera_ray_from, in its association and precision.
__host__ __device__ inline void pt_camera_fill_constants(PtCamera &cam)
{
	const float *V = cam.invView, *P = cam.invProj;
	for (int k = 0; k < 3; ++k)
	{
		cam.origin[k] = (V[0 + k] * 0.0f + V[4 + k] * 0.0f) + (V[8 + k] * 0.0f + V[12 + k] * 1.0f);
		cam.projConst[k] = P[8 + k] * 1.0f + P[12 + k] * 1.0f;
		cam.viewZero[k] = V[12 + k] * 0.0f;
		cam.pad[k] = 0.0f;
	}
}

// raytrace.rgen:40-46 with inverse(view)/inverse(proj) hoisted to the host.  Matrix * vector in
// glm's association: (m0*x + m1*y) + (m2*z + m3*w).  V, P: the two inverse matrices (column-major), through any pointer type.
template <typename FP>
__device__ __forceinline__ void pt_camera_ray_from(FP V, FP P, uint32_t W, uint32_t H, uint32_t px, uint32_t py, float jx,
                                                   float jy, float3x &o, float3x &d)
{
	float uvx = ((float)px + jx) / (float)W;
	float uvy = ((float)py + jy) / (float)H;
	float dx = uvx * 2.0f - 1.0f;
	float dy = uvy * 2.0f - 1.0f;
	o.x = (V[0] * 0.0f + V[4] * 0.0f) + (V[8] * 0.0f + V[12] * 1.0f);
	o.y = (V[1] * 0.0f + V[5] * 0.0f) + (V[9] * 0.0f + V[13] * 1.0f);
	o.z = (V[2] * 0.0f + V[6] * 0.0f) + (V[10] * 0.0f + V[14] * 1.0f);
	float3x t;
	t.x = (P[0] * dx + P[4] * dy) + (P[8] * 1.0f + P[12] * 1.0f);
	t.y = (P[1] * dx + P[5] * dy) + (P[9] * 1.0f + P[13] * 1.0f);
	t.z = (P[2] * dx + P[6] * dy) + (P[10] * 1.0f + P[14] * 1.0f);
	float3x n = pt_normalize(t);
	d.x = (V[0] * n.x + V[4] * n.y) + (V[8] * n.z + V[12] * 0.0f);
	d.y = (V[1] * n.x + V[5] * n.y) + (V[9] * n.z + V[13] * 0.0f);
	d.z = (V[2] * n.x + V[6] * n.y) + (V[10] * n.z + V[14] * 0.0f);
}

// The same ray from a PtCamera behind `c` whose constants are filled in (k_paths: 44 floats in device memory).
template <typename FP>
__device__ __forceinline__ void pt_camera_ray_const(FP c, uint32_t W, uint32_t H, uint32_t px, uint32_t py, float jx, float jy,
                                                    float3x &o, float3x &d)
{
	FP V = c, P = c + 16;
	float uvx = ((float)px + jx) / (float)W;
	float uvy = ((float)py + jy) / (float)H;
	float dx = uvx * 2.0f - 1.0f;
	float dy = uvy * 2.0f - 1.0f;
	o.x = c[32];
	o.y = c[33];
	o.z = c[34];
	float3x t;
	t.x = (P[0] * dx + P[4] * dy) + c[35];
	t.y = (P[1] * dx + P[5] * dy) + c[36];
	t.z = (P[2] * dx + P[6] * dy) + c[37];
	float3x n = pt_normalize<true>(t);  // k_paths: every valid lane of the item generates here
	d.x = (V[0] * n.x + V[4] * n.y) + (V[8] * n.z + c[38]);
	d.y = (V[1] * n.x + V[5] * n.y) + (V[9] * n.z + c[39]);
	d.z = (V[2] * n.x + V[6] * n.y) + (V[10] * n.z + c[40]);
}

__device__ __forceinline__ void pt_camera_ray(const PtCamera &cam, uint32_t W, uint32_t H, uint32_t px, uint32_t py,
                                              float jx, float jy, float3x &o, float3x &d)
{
	pt_camera_ray_from<const float *>(cam.invView, cam.invProj, W, H, px, py, jx, jy, o, d);
}

// Uniform read-only data behind a pointer in the constant address space: read with scalar loads where it is used
// instead of living in SGPRs for the whole kernel.
typedef const float __attribute__((address_space(4))) pt_const_float;

// ------------------------------------------------------- watertight ray/triangle test ----
// Woop, Benthin, Wald, "Watertight Ray/Triangle Intersection" (JCGT 2013), without backface
// culling (the instance sets TRIANGLE_FACING_CULL_DISABLE, VulkanRaytracing.cpp:405).
struct PtRayShear
{
	int kx, ky, kz;
	float Sx, Sy, Sz;
};

__device__ __forceinline__ float pt_sel3(float3x v, int k)
{
	return k == 0 ? v.x : (k == 1 ? v.y : v.z);
}

__device__ __forceinline__ PtRayShear pt_ray_shear(float3x d)
{
	PtRayShear rs;
	int kz = 0;
	float m = __builtin_fabsf(d.x);
	if (__builtin_fabsf(d.y) > m)
	{
		kz = 1;
		m = __builtin_fabsf(d.y);
	}
	if (__builtin_fabsf(d.z) > m) kz = 2;
	int kx = kz + 1;
	if (kx == 3) kx = 0;
	int ky = kx + 1;
	if (ky == 3) ky = 0;
	float dz = pt_sel3(d, kz);
	if (dz < 0.0f)
	{
		int t = kx;
		kx = ky;
		ky = t;
	}
	rs.kx = kx;
	rs.ky = ky;
	rs.kz = kz;
	rs.Sx = pt_sel3(d, kx) / dz;
	rs.Sy = pt_sel3(d, ky) / dz;
	rs.Sz = 1.0f / dz;
	return rs;
}

struct PtBest
{
	float t;        // closest accepted distance so far (initially tmax)
	uint32_t prim;  // PT_NO_HIT until a candidate is accepted
	float V, W, det;
};

// Candidate accepted iff tmin < t and (t < best.t, or t == best.t with a lower primitive id).
__device__ __forceinline__ void pt_tri_test(float3x o, const PtRayShear &rs, float tmin, float3x p0, float3x p1,
                                            float3x p2, uint32_t prim, PtBest &best)
{
	float3x A = { p0.x - o.x, p0.y - o.y, p0.z - o.z };
	float3x B = { p1.x - o.x, p1.y - o.y, p1.z - o.z };
	float3x C = { p2.x - o.x, p2.y - o.y, p2.z - o.z };
	float Akz = pt_sel3(A, rs.kz), Bkz = pt_sel3(B, rs.kz), Ckz = pt_sel3(C, rs.kz);
	float Ax = pt_sel3(A, rs.kx) - rs.Sx * Akz, Ay = pt_sel3(A, rs.ky) - rs.Sy * Akz;
	float Bx = pt_sel3(B, rs.kx) - rs.Sx * Bkz, By = pt_sel3(B, rs.ky) - rs.Sy * Bkz;
	float Cx = pt_sel3(C, rs.kx) - rs.Sx * Ckz, Cy = pt_sel3(C, rs.ky) - rs.Sy * Ckz;
	float U = Cx * By - Cy * Bx;
	float V = Ax * Cy - Ay * Cx;
	float W = Bx * Ay - By * Ax;
	if (U == 0.0f || V == 0.0f || W == 0.0f)
	{
		U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
		V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
		W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
	}
	if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return;
	float det = (U + V) + W;
	if (det == 0.0f) return;
	float Az = rs.Sz * Akz, Bz = rs.Sz * Bkz, Cz = rs.Sz * Ckz;
	float T = (U * Az + V * Bz) + W * Cz;
	float t = T / det;
	if (t > tmin && (t < best.t || (t == best.t && best.prim != PT_NO_HIT && prim < best.prim)))
	{
		best.t = t;
		best.prim = prim;
		best.V = V;
		best.W = W;
		best.det = det;
	}
}

__device__ __forceinline__ PtHit pt_finish_hit(const PtBest &b, float tmax)
{
	PtHit h;
	h.prim = b.prim;
	if (b.prim != PT_NO_HIT)
	{
		h.t = b.t;
		h.u = b.V / b.det;
		h.v = b.W / b.det;
	}
	else
	{
		h.t = tmax;
		h.u = 0.0f;
		h.v = 0.0f;
	}
	return h;
}

// ------------------------------------------------------------------ wide BVH layout ----
// 64-byte quantised BVH4 node: one 64 B-aligned record = four 16-byte loads per lane.
//   child box c, axis a:  lo = origin[a] + qlo[a][c] * scale[a],  hi likewise with qhi; scale[a] is a power of two
//   stored as a float (no decode instructions in the traversal).
//   An empty child slot has qlo = 255, qhi = 0 on every axis and child = PT_EMPTY_CHILD: an inverted box, which fails the
//   interval test of the wave kernels by itself -- except for a ray so far away that the node's whole extent vanishes in the
//   test's 1e-6 relative slack; such a ray pushes PT_EMPTY_CHILD, which the leaf step pops without touching memory
//   (tests/test_gpu_edge.py::test_tiny_far_geometry_matches_brute_force).  The split kernel tests the reference instead.
//   child[c]: bit 31 = leaf.  inner: node index.  leaf: bits 30..28 = triangle count - 1,
//   bits 27..0 = first triangle in the leaf-ordered triangle array.
struct __attribute__((aligned(64))) PtNode4
{
	float origin[3];
	float scaleX;        // power-of-two quantisation step per axis (x here, y and z at the end of the record)
	uint32_t child[4];
	uint32_t qlo[3];     // qlo[a] = 4 bytes, one per child
	uint32_t qhi[3];
	float scaleY, scaleZ;
};
static_assert(sizeof(PtNode4) == 64, "PtNode4 must be 64 bytes");

#define PT_LEAF_FLAG 0x80000000u
#define PT_EMPTY_CHILD 0xFFFFFFFEu  // unused child slot (never reported as hit)
#define PT_TRAVERSAL_DONE 0xFFFFFFFFu
#define PT_INSTANCE_END 0xFFFFFFFDu  // stack marker: the bottom-level traversal of an instance is finished
#define PT_NO_INSTANCE 0xFFFFFFFFu   // lane is traversing the top-level structure
#define PT_LEAF_MAX 8      // the leaf reference has 3 bits for (count - 1)
#define PT_LEAF_DEFAULT 2

// Entry node (pt_wave_traverse.h): children of the flattened tree top, and the size of its decoded record in 16-byte units.
#ifndef PT_ENTRY_WIDTH
#define PT_ENTRY_WIDTH 16
#endif
#define PT_ENTRY_REC (2u + 7u * PT_ENTRY_WIDTH / 4u)

// 48-byte leaf triangle: three 16-byte loads.  (v0.xyz, v1.x) (v1.yz, v2.xy) (v2.z, prim, -, -)
struct __attribute__((aligned(16))) PtTri48
{
	float v0[3];
	float v1[3];
	float v2[3];
	uint32_t prim;
	uint32_t pad[2];
};
static_assert(sizeof(PtTri48) == 48, "PtTri48 must be 48 bytes");

// Quad leaf (k_paths only; pt_build.h, ptBuildQuadLeaves): the two triangles (a, b, c), (a, c, d) that a fan-triangulated
// quad face gives, as four vertices: v0 v1 v2 = triangle A in its own vertex order, triangle B = (v0, v2, v3).  Four
// 16-byte loads instead of six, four vertex transforms instead of six, and B's third edge function is A's second, negated
// (exactly: fl(x*y) is commutative and negation is exact), so t, u, v of both triangles keep their bits.
struct __attribute__((aligned(16))) PtQuad64
{
	float v0[3], v1[3], v2[3], v3[3];
	uint32_t primA, primB;
	uint32_t pad[2];
};
static_assert(sizeof(PtQuad64) == 64, "PtQuad64 must be 64 bytes");
#define PT_LEAF_QUAD 7u  // count field (count - 1) of a quad leaf reference; such scenes are built with leaves of <= 7 triangles

// Everything raytrace.rchit fetches for one primitive through matIndices -> Material and indices -> 3 x Vertex
// (rchit:27-37), gathered once at upload into one 96-byte record: six 16-byte loads of one lane instead of ten
// dependent ones (a divergent-address load in a hot phase costs the frame as much as ~10 VALU instructions,
// DESIGN.md 5.2).  Same values, so the shading arithmetic is unchanged.  Texture coordinates are not in it:
// scenes with a diffuse map take the gather path.
struct __attribute__((aligned(16))) PtShadeRec
{
	float pos[3][3];  // v0, v1, v2
	float nrm[3][3];
	float kd[3];      // Material::diffuse
	float ke[3];      // Material::emission
};
static_assert(sizeof(PtShadeRec) == 96, "PtShadeRec must be 96 bytes");

struct PtTexSlot  // the diffuse map of one Model
{
	const float4 *texels;  // nullptr: the Model has no map (hitValue = the material's diffuse colour)
	uint32_t w, h;
	uint32_t primEnd;  // first primitive id behind the Model's
	uint32_t pad;
};
static_assert(sizeof(PtTexSlot) == 24, "PtTexSlot must be 24 bytes");

struct PtSceneDev
{
	const PtVertex *vertices;
	const uint32_t *indices;
	const PtMaterial *materials;
	const uint32_t *matIndices;
	const PtShadeRec *shade;  // per-primitive shading records (nullptr: gather as raytrace.rchit does)
	const PtNode4 *nodes;
	const PtTri48 *tris;  // triPlanes planes of nTris records; the identity plane is the last
	uint32_t nTris;
	uint32_t nNodes;
	const PtQuad64 *quads;  // quad leaves (three planes of nQuads records, rotated like tris) or nullptr
	uint32_t nQuads;
	// optional diffuse map (pt_set_texture): texW x texH linear RGBA texels, row 0 = v 0; nullptr = none
	const float4 *texture;
	uint32_t texW, texH;
	uint32_t triPlanes;  // 3 (one plane per dominant ray axis) or 1 (identity only: the leaf test permutes with selects)
	// one map per Model (pt_set_model_texture; raytrace.rchit:27 reaches every per-Model buffer through
	// model.m[gl_InstanceCustomIndexEXT]): nTexSlots records in primitive order, slot m covers the primitives below its
	// primEnd that no earlier slot covers; nullptr = the scene-wide map above, or none.  `texture` is then not null (the
	// "this scene samples maps" flag that selects the gather path).
	const PtTexSlot *texSlots;
	uint32_t nTexSlots;
};

// Diffuse map lookup (no counterpart in the reference's shaders, which never sample textures/viking_room.png:
// SURVEY.md 8(f) row 3).  Repeat addressing, bilinear filter, every step in fp32 in exactly this order (the oracle's
// pto_sample_texture is the same expression tree).
__device__ __forceinline__ float3x pt_sample_texture(const float4 *tex, uint32_t w, uint32_t h, float u, float v)
{
	float su = u - floorf(u), sv = v - floorf(v);
	float x = su * (float)w - 0.5f, y = sv * (float)h - 0.5f;
	float x0f = floorf(x), y0f = floorf(y);
	float fx = x - x0f, fy = y - y0f;
	int ix0 = (int)x0f, iy0 = (int)y0f;
	int ix1 = ix0 + 1, iy1 = iy0 + 1;
	if (ix0 < 0) ix0 += (int)w;
	if (iy0 < 0) iy0 += (int)h;
	if (ix1 >= (int)w) ix1 -= (int)w;
	if (iy1 >= (int)h) iy1 -= (int)h;
	const float4 t00 = tex[(size_t)iy0 * w + ix0], t10 = tex[(size_t)iy0 * w + ix1];
	const float4 t01 = tex[(size_t)iy1 * w + ix0], t11 = tex[(size_t)iy1 * w + ix1];
	const float gx = 1.0f - fx, gy = 1.0f - fy;
	float3x r;
	r.x = (t00.x * gx + t10.x * fx) * gy + (t01.x * gx + t11.x * fx) * fy;
	r.y = (t00.y * gx + t10.y * fx) * gy + (t01.y * gx + t11.y * fx) * fy;
	r.z = (t00.z * gx + t10.z * fx) * gy + (t01.z * gx + t11.z * fx) * fy;
	return r;
}

// ---------------------------------------------------------------------- closest hit ----
// raytrace.rchit:24-59 for one hit.  Returns the new ray, updates seed; mat = material record.
// xf != nullptr: the hit lies in an instance with xf = object-to-world 3x4 followed by world-to-object 3x4 (both
// row-major): rchit:45-46 -- worldHitPos = vec3(gl_ObjectToWorldEXT * vec4(localHitPos, 1)) in glm's association
// (c0*x + c1*y) + (c2*z + c3*1); worldHitNormal = normalize(vec3(localHitNormal * gl_WorldToObjectEXT)), a row
// vector times the matrix.  xf == nullptr is the reference's single identity instance (no transform arithmetic).
// kd = hitValue of rchit:54 (the material's diffuse colour, times the diffuse map when the scene has one), ke = emission.
// The arithmetic of a hit once its three vertices are known (rchit:39-53): interpolation, the instance's transform, the new ray.
template <bool kInRange = false>
__device__ __forceinline__ void pt_shade_math(const float p0[3], const float p1[3], const float p2[3], const float n0[3],
                                              const float n1[3], const float n2[3], float bx, float by, float bz,
                                              uint32_t &seed, float3x &o, float3x &d, const float *xf)
{
	float3x N, P;
	N.x = (n0[0] * bx + n1[0] * by) + n2[0] * bz;
	N.y = (n0[1] * bx + n1[1] * by) + n2[1] * bz;
	N.z = (n0[2] * bx + n1[2] * by) + n2[2] * bz;
	P.x = (p0[0] * bx + p1[0] * by) + p2[0] * bz;
	P.y = (p0[1] * bx + p1[1] * by) + p2[1] * bz;
	P.z = (p0[2] * bx + p1[2] * by) + p2[2] * bz;
	if (xf)
	{
		const float *w = xf + 12;
		float3x Pw, Nw;
		Pw.x = (xf[0] * P.x + xf[1] * P.y) + (xf[2] * P.z + xf[3] * 1.0f);
		Pw.y = (xf[4] * P.x + xf[5] * P.y) + (xf[6] * P.z + xf[7] * 1.0f);
		Pw.z = (xf[8] * P.x + xf[9] * P.y) + (xf[10] * P.z + xf[11] * 1.0f);
		Nw.x = (N.x * w[0] + N.y * w[4]) + N.z * w[8];
		Nw.y = (N.x * w[1] + N.y * w[5]) + N.z * w[9];
		Nw.z = (N.x * w[2] + N.y * w[6]) + N.z * w[10];
		P = Pw;
		N = Nw;
	}
	float3x Nn = pt_normalize<kInRange>(N);
	float3x rv = pt_rnd_unit_vector<kInRange>(seed);
	d = pt_normalize<kInRange>({ Nn.x + rv.x, Nn.y + rv.y, Nn.z + rv.z });
	o = P;
}

template <bool kInRange = false>
__device__ __forceinline__ void pt_shade_hit(const PtSceneDev &sc, uint32_t prim, float u, float v, uint32_t &seed,
                                         float3x &o, float3x &d, float3x &kd, float3x &ke, const float *xf = nullptr)
{
	float bx = (1.0f - u) - v, by = u, bz = v;
	float p0[3], p1[3], p2[3], n0[3], n1[3], n2[3];
	if (sc.shade && !sc.texture)
	{
		const float4 *r = reinterpret_cast<const float4 *>(sc.shade + prim);
		const float4 a = r[0], b = r[1], c = r[2], e = r[3], f = r[4], g = r[5];
		p0[0] = a.x; p0[1] = a.y; p0[2] = a.z; p1[0] = a.w; p1[1] = b.x; p1[2] = b.y; p2[0] = b.z; p2[1] = b.w; p2[2] = c.x;
		n0[0] = c.y; n0[1] = c.z; n0[2] = c.w; n1[0] = e.x; n1[1] = e.y; n1[2] = e.z; n2[0] = e.w; n2[1] = f.x; n2[2] = f.y;
		kd = { f.z, f.w, g.x };
		ke = { g.y, g.z, g.w };
		// (the arithmetic is instantiated in both branches: joined behind them, the eighteen loaded values are first copied
		// into the registers the other branch would have put them in)
		pt_shade_math<kInRange>(p0, p1, p2, n0, n1, n2, bx, by, bz, seed, o, d, xf);
		return;
	}
	else
	{
		const float *mat = sc.materials[sc.matIndices[prim]].diffuse;
		kd = { mat[0], mat[1], mat[2] };
		ke = { mat[3], mat[4], mat[5] };
		const float *v0 = sc.vertices[sc.indices[3 * (size_t)prim + 0]].pos;
		const float *v1 = sc.vertices[sc.indices[3 * (size_t)prim + 1]].pos;
		const float *v2 = sc.vertices[sc.indices[3 * (size_t)prim + 2]].pos;
#pragma unroll
		for (int k = 0; k < 3; ++k)
		{
			p0[k] = v0[k]; p1[k] = v1[k]; p2[k] = v2[k];
			n0[k] = v0[3 + k]; n1[k] = v1[3 + k]; n2[k] = v2[3 + k];
		}
		if (sc.texture)
		{
			const float4 *tex = sc.texture;
			uint32_t tw = sc.texW, th = sc.texH;
			if (sc.texSlots)  // (wave-uniform) one map per Model: the slot whose primitive range holds the hit
			{
				uint32_t m = 0;
				while (m + 1u < sc.nTexSlots && prim >= sc.texSlots[m].primEnd) ++m;
				const PtTexSlot slot = sc.texSlots[m];
				tex = slot.texels;
				tw = slot.w;
				th = slot.h;
			}
			if (tex)
			{
				float tu = (v0[9] * bx + v1[9] * by) + v2[9] * bz, tv = (v0[10] * bx + v1[10] * by) + v2[10] * bz;
				float3x tx = pt_sample_texture(tex, tw, th, tu, tv);
				kd = { kd.x * tx.x, kd.y * tx.y, kd.z * tx.z };
			}
		}
	}
	pt_shade_math<kInRange>(p0, p1, p2, n0, n1, n2, bx, by, bz, seed, o, d, xf);
}

