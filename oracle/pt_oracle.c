/*
 * pt_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY: may be loaded by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
 * (pathtracer_amd/, libpt_hip.so) never links, loads or calls it.
 *
 * Plain-C restatement of the reference's ray-tracing hot path:
 *   shaders/random.glsl:27-64      tea / lcg / rnd / rnd_unit_vector
 *   shaders/raytrace.rgen:25-96    seed, sample loop, camera ray, bounce loop, progressive mix
 *   shaders/raytrace.rchit:24-59   material + vertex fetch, barycentric interpolation, Lambert bounce
 *   shaders/raytrace.rmiss:10-15   black environment, terminate
 *   main.cpp:1465-1480             camera (glm::lookAt / glm::perspective RH_NO / proj[1][1]*=-1)
 *   glm/detail/func_matrix.inl:347-404 (inverse), glm/ext/matrix_transform.inl:153-173 (lookAtRH),
 *   glm/ext/matrix_clip_space.inl:249-262 (perspectiveRH_NO)
 * plus the pieces the reference delegates to the Vulkan driver and which this
 * build therefore DEFINES (closest-hit query = brute force over all triangles with
 * a watertight Woop/Benthin/Wald test, deterministic tie-break = lowest primitive
 * id) and the build's own Morton-code definition.
 *
 * PARITY STATUS.  Host seam (OBJ parse, camera matrices, inverse) is pinned against
 * the reference's real tinyobjloader+glm through oracle/_ref -> tests/golden/.
 * The shader path has no reference tests, golden vectors or runnable reference
 * (no GLSL compiler / Vulkan ICD in the image): its parity is UNPINNED by the
 * reference; the restatement is pinned only by its own committed known-answer
 * vectors.
 *
 * ARITHMETIC CONTRACT (shared, by specification, with the HIP kernels so that
 * GPU and oracle agree bit-for-bit): IEEE-754 binary32, round-to-nearest-even,
 * no contraction (-ffp-contract=off) except where fmaf() is written out,
 * correctly-rounded / and sqrtf, denormals kept.  Expression order is exactly as
 * written below.  sin/cos of the reference (implementation-defined precision in
 * GLSL) are DEFINED here as the degree-9/8 Taylor polynomials of pto_sincos_2pi.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_NONE 0xFFFFFFFFu

/* ------------------------------------------------------------------ RNG -- */
/* random.glsl:27-41 */
uint32_t pto_tea(uint32_t val0, uint32_t val1)
{
	uint32_t v0 = val0, v1 = val1, s0 = 0;
	for (uint32_t n = 0; n < 16; n++)
	{
		s0 += 0x9e3779b9u;
		v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
		v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
	}
	return v0;
}

/* random.glsl:45-51 */
uint32_t pto_lcg(uint32_t *prev)
{
	*prev = 1664525u * (*prev) + 1013904223u;
	return *prev & 0x00FFFFFFu;
}

/* random.glsl:54-57 -- exact in binary32 (24-bit integer / 2^24) */
float pto_rnd(uint32_t *prev)
{
	return (float)pto_lcg(prev) / 16777216.0f;
}

/* sin and cos of 2*pi*u for u in [0,1).  The reference computes theta = rnd*2.0*M_PI
 * and calls GLSL cos/sin (random.glsl:59-64), whose precision is implementation
 * defined.  Definition used by this build (oracle and HIP alike): exact quadrant
 * reduction on u (u is k/2^24, so 4u-q is exact), then Taylor polynomials of
 * sin(pi/2 x), cos(pi/2 x) on |x| <= 1/2 evaluated in Horner form with fmaf.  Max
 * abs error vs the real sin/cos of the exact angle < 1.2e-7. */
void pto_sincos_2pi(float u, float *s_out, float *c_out)
{
	const float S0 = 1.57079632679489662f;   /*  (pi/2)      */
	const float S1 = -0.645964097506246254f; /* -(pi/2)^3/3! */
	const float S2 = 0.0796926262461670451f; /*  (pi/2)^5/5! */
	const float S3 = -0.00468175413531868810f;
	const float S4 = 0.000160441184787359822f;
	const float C1 = -1.23370055013616983f;  /* -(pi/2)^2/2! */
	const float C2 = 0.253669507901048014f;
	const float C3 = -0.0208634807633529609f;
	const float C4 = 0.000919260274839426580f;
	float t = u * 4.0f;                 /* exact */
	int q = (int)(t + 0.5f);            /* t + 0.5 exact for t < 4 with 22 fractional bits */
	float x = t - (float)q;             /* exact, in [-0.5, 0.5] */
	float x2 = x * x;
	float sp = fmaf(x2, S4, S3);
	sp = fmaf(x2, sp, S2);
	sp = fmaf(x2, sp, S1);
	sp = fmaf(x2, sp, S0);
	float s = x * sp;
	float cp = fmaf(x2, C4, C3);
	cp = fmaf(x2, cp, C2);
	cp = fmaf(x2, cp, C1);
	float c = fmaf(x2, cp, 1.0f);
	switch (q & 3)
	{
	case 0: *s_out = s; *c_out = c; break;
	case 1: *s_out = c; *c_out = -s; break;
	case 2: *s_out = -s; *c_out = -c; break;
	default: *s_out = -c; *c_out = s; break;
	}
}

/* random.glsl:59-64: theta draw first, then z */
void pto_rnd_unit_vector(uint32_t *prev, float out[3])
{
	float u = pto_rnd(prev);
	float z = pto_rnd(prev) * 2.0f - 1.0f;
	float r = sqrtf(1.0f - z * z);
	float s, c;
	pto_sincos_2pi(u, &s, &c);
	out[0] = r * c;
	out[1] = r * s;
	out[2] = z;
}

/* ------------------------------------------------------ small vector ops -- */
static inline float dot3(const float a[3], const float b[3])
{
	/* glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z */
	return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static inline void normalize3(const float v[3], float out[3])
{
	/* glm: v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
	float inv = 1.0f / sqrtf(dot3(v, v));
	out[0] = v[0] * inv;
	out[1] = v[1] * inv;
	out[2] = v[2] * inv;
}
static inline void cross3(const float x[3], const float y[3], float out[3])
{
	/* glm compute_cross */
	out[0] = x[1] * y[2] - y[1] * x[2];
	out[1] = x[2] * y[0] - y[2] * x[0];
	out[2] = x[0] * y[1] - y[0] * x[1];
}

/* ---------------------------------------------------------------- camera -- */
/* Column-major 4x4 as glm: m[c*4 + r]. */
/* glm/ext/matrix_transform.inl:153-173 (lookAtRH) */
void pto_lookat(const float eye[3], const float center[3], const float up[3], float m[16])
{
	float fv[3] = { center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] };
	float f[3], sv[3], s[3], u[3];
	normalize3(fv, f);
	cross3(f, up, sv);
	normalize3(sv, s);
	cross3(s, f, u);
	for (int i = 0; i < 16; ++i) m[i] = 0.0f;
	m[0] = s[0]; m[4] = s[1]; m[8] = s[2];
	m[1] = u[0]; m[5] = u[1]; m[9] = u[2];
	m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
	m[12] = -dot3(s, eye);
	m[13] = -dot3(u, eye);
	m[14] = dot3(f, eye);
	m[15] = 1.0f;
}

/* glm/ext/matrix_clip_space.inl:249-262 (perspectiveRH_NO); tan() is libm tanf as
 * std::tan(float) resolves in glm. */
void pto_perspective(float fovy, float aspect, float zNear, float zFar, float m[16])
{
	float tanHalfFovy = tanf(fovy / 2.0f);
	for (int i = 0; i < 16; ++i) m[i] = 0.0f;
	m[0] = 1.0f / (aspect * tanHalfFovy);
	m[5] = 1.0f / (tanHalfFovy);
	m[10] = -(zFar + zNear) / (zFar - zNear);
	m[11] = -1.0f;
	m[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
}

/* glm/detail/func_matrix.inl:347-404 (compute_inverse<4,4>) */
void pto_inverse4(const float a[16], float out[16])
{
#define M(c, r) a[(c) * 4 + (r)]
	float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
	float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
	float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
	float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
	float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
	float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
	float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
	float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
	float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
	float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
	float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
	float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
	float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
	float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
	float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
	float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
	float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
	float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
	float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
	float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
	float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
	float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
	float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
	float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
	float Vec0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) };
	float Vec1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
	float Vec2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) };
	float Vec3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
	const float SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
	float Inv[16];
	for (int i = 0; i < 4; ++i)
	{
		float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
		float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
		float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
		float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
		Inv[0 * 4 + i] = Inv0 * SignA[i];
		Inv[1 * 4 + i] = Inv1 * SignB[i];
		Inv[2 * 4 + i] = Inv2 * SignA[i];
		Inv[3 * 4 + i] = Inv3 * SignB[i];
	}
	/* Row0 = (Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]); Dot0 = m[0] * Row0 */
	float d0 = M(0, 0) * Inv[0], d1 = M(0, 1) * Inv[4], d2 = M(0, 2) * Inv[8], d3 = M(0, 3) * Inv[12];
	float Dot1 = (d0 + d1) + (d2 + d3);
	float OneOverDeterminant = 1.0f / Dot1;
	for (int i = 0; i < 16; ++i) out[i] = Inv[i] * OneOverDeterminant;
#undef M
}

/* main.cpp:1465-1480 */
void pto_reference_camera(uint32_t W, uint32_t H, float view[16], float proj[16])
{
	const float eye[3] = { 0.0f, 1.0f, 3.5f }, center[3] = { 0.0f, 1.0f, 0.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
	pto_lookat(eye, center, up, view);
	/* glm::radians(45.0f) = degrees * 0.01745329251994329576923690768489 (as float) */
	float fovy = 45.0f * 0.01745329251994329576923690768489f;
	pto_perspective(fovy, (float)W / (float)H, 0.1f, 10.0f, proj);
	proj[5] *= -1.0f;
}

/* glm operator*(mat4, vec4): (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w) */
static inline void mat4_mul_vec4(const float m[16], float x, float y, float z, float w, float out[4])
{
	for (int i = 0; i < 4; ++i) out[i] = (m[0 + i] * x + m[4 + i] * y) + (m[8 + i] * z + m[12 + i] * w);
}

/* raytrace.rgen:40-46 with the two inverses hoisted (invView = inverse(ubo.view) ...) */
void pto_camera_ray(const float invView[16], const float invProj[16], uint32_t W, uint32_t H, uint32_t px,
                    uint32_t py, float jx, float jy, float o[3], float d[3])
{
	float uvx = ((float)px + jx) / (float)W;
	float uvy = ((float)py + jy) / (float)H;
	float dx = uvx * 2.0f - 1.0f;
	float dy = uvy * 2.0f - 1.0f;
	float origin[4], target[4], tn[3], dir[4];
	mat4_mul_vec4(invView, 0.0f, 0.0f, 0.0f, 1.0f, origin);
	mat4_mul_vec4(invProj, dx, dy, 1.0f, 1.0f, target);
	normalize3(target, tn);
	mat4_mul_vec4(invView, tn[0], tn[1], tn[2], 0.0f, dir);
	o[0] = origin[0]; o[1] = origin[1]; o[2] = origin[2];
	d[0] = dir[0]; d[1] = dir[1]; d[2] = dir[2];
}

/* ----------------------------------------------------------------- scene -- */
/* Byte-identical to the reference's buffers: Vertex = 11 floats (main.cpp:60-65),
 * Material = 6 floats (main.cpp:54-58). */
typedef struct
{
	const float *verts;     /* nVerts * 11 */
	uint32_t nVerts;
	const uint32_t *indices; /* nIdx */
	uint32_t nIdx;
	const float *materials; /* nMat * 6 */
	uint32_t nMat;
	const uint32_t *matIdx; /* nIdx/3 */
	/* Two-level instancing (reference data model: VulkanRaytracing.cpp:382-411 builds ONE
	 * instance with the identity transform; raytrace.rchit:45-46 already applies gl_ObjectToWorldEXT /
	 * gl_WorldToObjectEXT).  nInst == 0: the reference's single identity instance (no transform arithmetic at
	 * all).  Otherwise nInst records of 24 floats: object-to-world 3x4 row-major (VkTransformMatrixKHR), then
	 * world-to-object 3x4 row-major (glm::inverse of the 4x4, computed by the caller with pto_inverse4). */
	const float *instances;
	uint32_t nInst;
	/* Diffuse map (no counterpart in the reference's shaders, which carry Vertex::texCoord
	 * to raytrace.rchit but never sample textures/viking_room.png): texW x texH linear RGBA float texels, row 0 =
	 * v 0; NULL = none.  hitValue = material diffuse x pto_sample_texture(interpolated texCoord). */
	const float *texture;
	uint32_t texW, texH;
	/* Several Models (raytrace.rchit:27 model.m[gl_InstanceCustomIndexEXT]; instance i refers to m_blases[i] AND
	 * Model i, VulkanRaytracing.cpp:395-399): the buffers above hold the models one after the other (indices and material
	 * indices rebased by the caller), and instance ii contains only the primitives [instRange[2*ii], + instRange[2*ii+1]).
	 * NULL: every instance shows the whole scene (one Model). */
	const uint32_t *instRange;
	/* One diffuse map per Model (the product's pt_set_model_texture): nTexSlots records in primitive order, slot m covers
	 * the primitives below its primEnd that no earlier slot covers; texels NULL = that Model has no map.  NULL: the
	 * scene-wide map above (or none). */
	const struct PtoTexSlot *texSlots;
	uint32_t nTexSlots;
} PtoScene;

typedef struct PtoTexSlot
{
	const float *texels;
	uint32_t w, h, primEnd, pad;
} PtoTexSlot;

/* Repeat addressing, bilinear filter; every step in fp32 in exactly this order (this build's definition of the
 * lookup; the device's pt_sample_texture is the same expression tree). */
void pto_sample_texture(const float *tex, uint32_t w, uint32_t h, float u, float v, float out[3])
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
	const float *t00 = tex + 4 * ((size_t)iy0 * w + ix0), *t10 = tex + 4 * ((size_t)iy0 * w + ix1);
	const float *t01 = tex + 4 * ((size_t)iy1 * w + ix0), *t11 = tex + 4 * ((size_t)iy1 * w + ix1);
	float gx = 1.0f - fx, gy = 1.0f - fy;
	for (int k = 0; k < 3; ++k) out[k] = (t00[k] * gx + t10[k] * fx) * gy + (t01[k] * gx + t11[k] * fx) * fy;
}

typedef struct
{
	float t, u, v;  /* u,v = barycentric weights of vertex 1 and 2 (hitAttributeEXT attribs) */
	uint32_t prim;  /* PTO_NONE on miss */
} PtoHit;

/* Per-ray constants of the watertight test (Woop, Benthin, Wald 2013). */
typedef struct
{
	int kx, ky, kz;
	float Sx, Sy, Sz;
} PtoRayShear;

static inline void ray_shear(const float d[3], PtoRayShear *rs)
{
	int kz = 0;
	if (fabsf(d[1]) > fabsf(d[kz])) kz = 1;
	if (fabsf(d[2]) > fabsf(d[kz])) kz = 2;
	int kx = kz + 1; if (kx == 3) kx = 0;
	int ky = kx + 1; if (ky == 3) ky = 0;
	if (d[kz] < 0.0f) { int t = kx; kx = ky; ky = t; }
	rs->kx = kx; rs->ky = ky; rs->kz = kz;
	rs->Sx = d[kx] / d[kz];
	rs->Sy = d[ky] / d[kz];
	rs->Sz = 1.0f / d[kz];
}

/* One ray/triangle test.  Updates *best (t, prim) and the (V, W, det) of the best
 * candidate.  Candidate accepted iff tmin < t and (t < best.t or equal t with a
 * lower primitive id).  Returns 1 if the edge tests passed (for statistics). */
static inline int tri_test_inst(const float o[3], const PtoRayShear *rs, float tmin, const float *p0, const float *p1,
                                const float *p2, uint32_t prim, uint32_t inst, float *best_t, uint32_t *best_prim,
                                uint32_t *best_inst, float *bV, float *bW, float *bDet);

static inline int tri_test(const float o[3], const PtoRayShear *rs, float tmin, const float *p0, const float *p1,
                           const float *p2, uint32_t prim, float *best_t, uint32_t *best_prim, float *bV, float *bW,
                           float *bDet)
{
	uint32_t bi = 0;
	return tri_test_inst(o, rs, tmin, p0, p1, p2, prim, 0u, best_t, best_prim, &bi, bV, bW, bDet);
}

/* Ties between instances: equal t -> lower instance index, then lower primitive id (order independent). */
static inline int tri_test_inst(const float o[3], const PtoRayShear *rs, float tmin, const float *p0, const float *p1,
                                const float *p2, uint32_t prim, uint32_t inst, float *best_t, uint32_t *best_prim,
                                uint32_t *best_inst, float *bV, float *bW, float *bDet)
{
	float A[3] = { p0[0] - o[0], p0[1] - o[1], p0[2] - o[2] };
	float B[3] = { p1[0] - o[0], p1[1] - o[1], p1[2] - o[2] };
	float C[3] = { p2[0] - o[0], p2[1] - o[1], p2[2] - o[2] };
	float Ax = A[rs->kx] - rs->Sx * A[rs->kz], Ay = A[rs->ky] - rs->Sy * A[rs->kz];
	float Bx = B[rs->kx] - rs->Sx * B[rs->kz], By = B[rs->ky] - rs->Sy * B[rs->kz];
	float Cx = C[rs->kx] - rs->Sx * C[rs->kz], Cy = C[rs->ky] - rs->Sy * C[rs->kz];
	float U = Cx * By - Cy * Bx;
	float V = Ax * Cy - Ay * Cx;
	float W = Bx * Ay - By * Ax;
	if (U == 0.0f || V == 0.0f || W == 0.0f)
	{
		U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
		V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
		W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
	}
	if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
	float det = (U + V) + W;
	if (det == 0.0f) return 0;
	float Az = rs->Sz * A[rs->kz], Bz = rs->Sz * B[rs->kz], Cz = rs->Sz * C[rs->kz];
	float T = (U * Az + V * Bz) + W * Cz;
	float t = T / det;
	if (t > tmin && (t < *best_t || (t == *best_t && *best_prim != PTO_NONE &&
	                                 (inst < *best_inst || (inst == *best_inst && prim < *best_prim)))))
	{
		*best_t = t; *best_prim = prim; *best_inst = inst; *bV = V; *bW = W; *bDet = det;
	}
	return 1;
}

/* The closest-hit query that traceRayEXT (raytrace.rgen:65-76) delegates to the
 * driver: opaque triangles, no culling (VulkanRaytracing.cpp:405), tmin < t < tmax. */
void pto_intersect_brute(const PtoScene *sc, const float o[3], const float d[3], float tmin, float tmax, PtoHit *hit)
{
	PtoRayShear rs;
	ray_shear(d, &rs);
	float best_t = tmax, bV = 0, bW = 0, bDet = 1;
	uint32_t best_prim = PTO_NONE;
	uint32_t nTri = sc->nIdx / 3;
	for (uint32_t p = 0; p < nTri; ++p)
	{
		const float *p0 = sc->verts + 11 * (size_t)sc->indices[3 * p + 0];
		const float *p1 = sc->verts + 11 * (size_t)sc->indices[3 * p + 1];
		const float *p2 = sc->verts + 11 * (size_t)sc->indices[3 * p + 2];
		tri_test(o, &rs, tmin, p0, p1, p2, p, &best_t, &best_prim, &bV, &bW, &bDet);
	}
	hit->prim = best_prim;
	if (best_prim != PTO_NONE)
	{
		hit->t = best_t;
		hit->u = bV / bDet;
		hit->v = bW / bDet;
	}
	else
	{
		hit->t = tmax; hit->u = 0.0f; hit->v = 0.0f;
	}
}

/* ------------------------------------------------- oracle-side BVH2 ------ */
/* A plain median-split BVH used ONLY to make the oracle fast enough on scenes with
 * thousands of triangles.  It never changes results: leaves run the same tri_test
 * and the slab test is conservative (boxes padded, t-interval padded); tests check
 * it against pto_intersect_brute. */
typedef struct
{
	float lo[3], hi[3];
	uint32_t left, right; /* children, or (first, count) when count != 0 */
	uint32_t first, count;
} PtoNode;

typedef struct
{
	PtoNode *nodes;
	uint32_t nNodes;
	uint32_t *prims;
	float *cent; /* scratch */
	const PtoScene *sc;
} PtoBvh;

static void prim_bounds(const PtoScene *sc, uint32_t p, float lo[3], float hi[3])
{
	for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
	for (int j = 0; j < 3; ++j)
	{
		const float *v = sc->verts + 11 * (size_t)sc->indices[3 * p + j];
		for (int k = 0; k < 3; ++k)
		{
			if (v[k] < lo[k]) lo[k] = v[k];
			if (v[k] > hi[k]) hi[k] = v[k];
		}
	}
}

static uint32_t bvh_build_rec(PtoBvh *b, uint32_t first, uint32_t count, float pad)
{
	uint32_t id = b->nNodes++;
	PtoNode *n = &b->nodes[id];
	float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
	for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
	for (uint32_t i = first; i < first + count; ++i)
	{
		float lo[3], hi[3];
		prim_bounds(b->sc, b->prims[i], lo, hi);
		for (int k = 0; k < 3; ++k)
		{
			if (lo[k] < n->lo[k]) n->lo[k] = lo[k];
			if (hi[k] > n->hi[k]) n->hi[k] = hi[k];
			float c = 0.5f * (lo[k] + hi[k]);
			if (c < clo[k]) clo[k] = c;
			if (c > chi[k]) chi[k] = c;
		}
	}
	for (int k = 0; k < 3; ++k) { n->lo[k] -= pad; n->hi[k] += pad; }
	n->first = first; n->count = count; n->left = n->right = 0;
	if (count <= 4) return id;
	int axis = 0;
	if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
	if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
	float split = 0.5f * (clo[axis] + chi[axis]);
	uint32_t i = first, j = first + count;
	while (i < j)
	{
		float lo[3], hi[3];
		prim_bounds(b->sc, b->prims[i], lo, hi);
		if (0.5f * (lo[axis] + hi[axis]) < split) ++i;
		else { --j; uint32_t t = b->prims[i]; b->prims[i] = b->prims[j]; b->prims[j] = t; }
	}
	uint32_t nl = i - first;
	if (nl == 0 || nl == count) nl = count / 2;
	n->count = 0;
	uint32_t l = bvh_build_rec(b, first, nl, pad);
	uint32_t r = bvh_build_rec(b, first + nl, count - nl, pad);
	n = &b->nodes[id];
	n->left = l; n->right = r;
	return id;
}

PtoBvh *pto_bvh_build(const PtoScene *sc)
{
	uint32_t nTri = sc->nIdx / 3;
	PtoBvh *b = (PtoBvh *)calloc(1, sizeof(PtoBvh));
	b->sc = sc;
	b->nodes = (PtoNode *)malloc(sizeof(PtoNode) * (2 * (size_t)nTri + 1));
	b->prims = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)nTri + 1));
	float m = 0.0f;
	for (uint32_t i = 0; i < sc->nVerts; ++i)
		for (int k = 0; k < 3; ++k)
		{
			float a = fabsf(sc->verts[11 * (size_t)i + k]);
			if (a > m) m = a;
		}
	for (uint32_t i = 0; i < nTri; ++i) b->prims[i] = i;
	if (nTri) bvh_build_rec(b, 0, nTri, m * (1.0f / 65536.0f));
	return b;
}

void pto_bvh_free(PtoBvh *b)
{
	if (!b) return;
	free(b->nodes); free(b->prims); free(b);
}

static inline int slab(const PtoNode *n, const float o[3], const float inv[3], float tmin, float tmax)
{
	float t0 = tmin, t1 = tmax;
	for (int k = 0; k < 3; ++k)
	{
		float a = (n->lo[k] - o[k]) * inv[k], b = (n->hi[k] - o[k]) * inv[k];
		float tn = fminf(a, b), tf = fmaxf(a, b); /* fmin/fmax drop NaN (0*inf) */
		tn = tn * 0.99999f; tf = tf * 1.00001f;
		if (tn < 0.0f) tn = tn * 1.00003f;
		if (tf < 0.0f) tf = tf * 0.99997f;
		t0 = fmaxf(t0, tn); t1 = fminf(t1, tf);
	}
	return t0 <= t1;
}

void pto_intersect_bvh(const PtoBvh *b, const float o[3], const float d[3], float tmin, float tmax, PtoHit *hit)
{
	const PtoScene *sc = b->sc;
	PtoRayShear rs;
	ray_shear(d, &rs);
	float best_t = tmax, bV = 0, bW = 0, bDet = 1;
	uint32_t best_prim = PTO_NONE;
	float inv[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
	uint32_t stack[128]; int sp = 0;
	if (b->nNodes) stack[sp++] = 0;
	while (sp)
	{
		const PtoNode *n = &b->nodes[stack[--sp]];
		if (!slab(n, o, inv, tmin, best_t)) continue;
		if (n->count)
		{
			for (uint32_t i = n->first; i < n->first + n->count; ++i)
			{
				uint32_t p = b->prims[i];
				const float *p0 = sc->verts + 11 * (size_t)sc->indices[3 * p + 0];
				const float *p1 = sc->verts + 11 * (size_t)sc->indices[3 * p + 1];
				const float *p2 = sc->verts + 11 * (size_t)sc->indices[3 * p + 2];
				tri_test(o, &rs, tmin, p0, p1, p2, p, &best_t, &best_prim, &bV, &bW, &bDet);
			}
		}
		else if (sp < 126) { stack[sp++] = n->left; stack[sp++] = n->right; }
	}
	hit->prim = best_prim;
	if (best_prim != PTO_NONE) { hit->t = best_t; hit->u = bV / bDet; hit->v = bW / bDet; }
	else { hit->t = tmax; hit->u = 0.0f; hit->v = 0.0f; }
}

/* Ray into the object space of an instance: o' = W2O * (o,1), d' = W2O * (d,0), the direction NOT renormalised,
 * so t means the same in both spaces (Vulkan spec, "Ray Intersection Candidate Determination").  Row r of the
 * 3x4 matrix m: ((m[r][0]*x + m[r][1]*y) + m[r][2]*z) + m[r][3]. */
static inline void ray_to_object(const float *w2o, const float o[3], const float d[3], float oo[3], float od[3])
{
	for (int r = 0; r < 3; ++r)
	{
		const float *m = w2o + 4 * r;
		oo[r] = ((m[0] * o[0] + m[1] * o[1]) + m[2] * o[2]) + m[3];
		od[r] = (m[0] * d[0] + m[1] * d[1]) + m[2] * d[2];
	}
}

/* Closest hit over all instances: the definition is "every instance, every triangle" -- no top-level
 * structure on the oracle side; the per-BLAS BVH2 only culls conservatively.  *inst = instance of the hit. */
void pto_intersect_instances(const PtoScene *sc, const PtoBvh *b, const float o[3], const float d[3], float tmin,
                             float tmax, PtoHit *hit, uint32_t *inst)
{
	float best_t = tmax, bV = 0, bW = 0, bDet = 1;
	uint32_t best_prim = PTO_NONE, best_inst = 0;
	uint32_t nTri = sc->nIdx / 3;
	for (uint32_t ii = 0; ii < sc->nInst; ++ii)
	{
		float oo[3], od[3];
		ray_to_object(sc->instances + 24 * (size_t)ii + 12, o, d, oo, od);
		PtoRayShear rs;
		ray_shear(od, &rs);
		const uint32_t pFirst = sc->instRange ? sc->instRange[2 * ii] : 0u;
		const uint32_t pEnd = sc->instRange ? pFirst + sc->instRange[2 * ii + 1] : nTri;
		if (!b)
		{
			for (uint32_t p = pFirst; p < pEnd; ++p)
			{
				const float *p0 = sc->verts + 11 * (size_t)sc->indices[3 * p + 0];
				const float *p1 = sc->verts + 11 * (size_t)sc->indices[3 * p + 1];
				const float *p2 = sc->verts + 11 * (size_t)sc->indices[3 * p + 2];
				tri_test_inst(oo, &rs, tmin, p0, p1, p2, p, ii, &best_t, &best_prim, &best_inst, &bV, &bW, &bDet);
			}
			continue;
		}
		float inv[3] = { 1.0f / od[0], 1.0f / od[1], 1.0f / od[2] };
		uint32_t stack[128]; int sp = 0;
		if (b->nNodes) stack[sp++] = 0;
		while (sp)
		{
			const PtoNode *n = &b->nodes[stack[--sp]];
			if (!slab(n, oo, inv, tmin, best_t)) continue;
			if (n->count)
			{
				for (uint32_t i = n->first; i < n->first + n->count; ++i)
				{
					uint32_t p = b->prims[i];
					if (p < pFirst || p >= pEnd) continue; /* a primitive of another Model */
					const float *p0 = sc->verts + 11 * (size_t)sc->indices[3 * p + 0];
					const float *p1 = sc->verts + 11 * (size_t)sc->indices[3 * p + 1];
					const float *p2 = sc->verts + 11 * (size_t)sc->indices[3 * p + 2];
					tri_test_inst(oo, &rs, tmin, p0, p1, p2, p, ii, &best_t, &best_prim, &best_inst, &bV, &bW, &bDet);
				}
			}
			else if (sp < 126) { stack[sp++] = n->left; stack[sp++] = n->right; }
		}
	}
	hit->prim = best_prim;
	*inst = best_inst;
	if (best_prim != PTO_NONE) { hit->t = best_t; hit->u = bV / bDet; hit->v = bW / bDet; }
	else { hit->t = tmax; hit->u = 0.0f; hit->v = 0.0f; }
}

static inline void intersect(const PtoScene *sc, const PtoBvh *bvh, const float o[3], const float d[3], float tmin,
                             float tmax, PtoHit *hit, uint32_t *inst)
{
	*inst = 0;
	if (sc->nInst) pto_intersect_instances(sc, bvh, o, d, tmin, tmax, hit, inst);
	else if (bvh) pto_intersect_bvh(bvh, o, d, tmin, tmax, hit);
	else pto_intersect_brute(sc, o, d, tmin, tmax, hit);
}

/* ------------------------------------------------------------- rendering -- */
typedef struct
{
	int32_t spp_per_frame; /* MAX_SAMPLES   (raytrace.rgen:22, reference 10) */
	int32_t max_depth;     /* MAX_RAY_DEPTH (raytrace.rgen:23, reference 20) */
	float tmin, tmax;      /* raytrace.rgen:51-52: 0.001, 10000 */
	uint32_t seed;         /* replaces int(clockARB()) of raytrace.rgen:28: val1 = seed + global frame */
} PtoParams;

/* raytrace.rchit:24-59.  Returns 1 if the path continues. */
static inline int closest_hit(const PtoScene *sc, const PtoHit *h, uint32_t inst, uint32_t *seed, float o[3], float d[3],
                              float hitValue[3], float emission[3])
{
	const float *mat = sc->materials + 6 * (size_t)sc->matIdx[h->prim];
	const float *v0 = sc->verts + 11 * (size_t)sc->indices[3 * h->prim + 0];
	const float *v1 = sc->verts + 11 * (size_t)sc->indices[3 * h->prim + 1];
	const float *v2 = sc->verts + 11 * (size_t)sc->indices[3 * h->prim + 2];
	float bx = (1.0f - h->u) - h->v, by = h->u, bz = h->v;
	float N[3], P[3], Nn[3], rv[3], nd[3];
	for (int k = 0; k < 3; ++k)
	{
		N[k] = (v0[3 + k] * bx + v1[3 + k] * by) + v2[3 + k] * bz;
		P[k] = (v0[k] * bx + v1[k] * by) + v2[k] * bz;
	}
	if (sc->nInst)
	{
		/* rchit:45-46: worldHitPos = vec3(gl_ObjectToWorldEXT * vec4(localHitPos, 1.0)) -- mat4x3 * vec4 in glm's
		 * association (c0*x + c1*y) + (c2*z + c3*w); worldHitNormal = normalize(vec3(localHitNormal *
		 * gl_WorldToObjectEXT)) -- row vector times matrix: component c = dot(N, column c of W2O). */
		const float *o2w = sc->instances + 24 * (size_t)inst, *w2o = o2w + 12;
		float Pw[3], Nw[3];
		for (int r = 0; r < 3; ++r)
		{
			const float *m = o2w + 4 * r;
			Pw[r] = (m[0] * P[0] + m[1] * P[1]) + (m[2] * P[2] + m[3] * 1.0f);
			Nw[r] = (N[0] * w2o[0 + r] + N[1] * w2o[4 + r]) + N[2] * w2o[8 + r];
		}
		for (int k = 0; k < 3; ++k) { P[k] = Pw[k]; N[k] = Nw[k]; }
	}
	/* nInst == 0: gl_ObjectToWorldEXT / gl_WorldToObjectEXT are identity (VulkanRaytracing.cpp:391-393) */
	normalize3(N, Nn);
	pto_rnd_unit_vector(seed, rv);
	nd[0] = Nn[0] + rv[0]; nd[1] = Nn[1] + rv[1]; nd[2] = Nn[2] + rv[2];
	normalize3(nd, d);
	o[0] = P[0]; o[1] = P[1]; o[2] = P[2];
	emission[0] = mat[3]; emission[1] = mat[4]; emission[2] = mat[5];
	hitValue[0] = mat[0]; hitValue[1] = mat[1]; hitValue[2] = mat[2];
	const float *tex = sc->texture;
	uint32_t tw = sc->texW, th = sc->texH;
	if (sc->texSlots)
	{
		uint32_t m = 0;
		while (m + 1u < sc->nTexSlots && h->prim >= sc->texSlots[m].primEnd) ++m;
		tex = sc->texSlots[m].texels;
		tw = sc->texSlots[m].w;
		th = sc->texSlots[m].h;
	}
	if (tex)
	{
		float tu = (v0[9] * bx + v1[9] * by) + v2[9] * bz, tv = (v0[10] * bx + v1[10] * by) + v2[10] * bz, tx[3];
		pto_sample_texture(tex, tw, th, tu, tv, tx);
		for (int k = 0; k < 3; ++k) hitValue[k] = hitValue[k] * tx[k];
	}
	return !((mat[3] + mat[4]) + mat[5] > 0.0f);
}

/* One pixel of one dispatch, raytrace.rgen:25-86: returns accumHitValue / MAX_SAMPLES. */
static void render_pixel(const PtoScene *sc, const PtoBvh *bvh, const float invView[16], const float invProj[16],
                         uint32_t W, uint32_t H, uint32_t px, uint32_t py, const PtoParams *pp, uint32_t gframe,
                         float out[3], uint64_t *rays)
{
	uint32_t seed = pto_tea(py * W + px, pp->seed + gframe);
	float acc[3] = { 0.0f, 0.0f, 0.0f };
	uint64_t nr = 0;
	for (int i = 0; i < pp->spp_per_frame; ++i)
	{
		float r1 = pto_rnd(&seed);
		float r2 = pto_rnd(&seed);
		float jx = gframe == 0 ? 0.5f : r1, jy = gframe == 0 ? 0.5f : r2;
		float o[3], d[3];
		pto_camera_ray(invView, invProj, W, H, px, py, jx, jy, o, d);
		/* payload.seed = seed (copy; the raygen-local seed is NOT advanced by the bounces) */
		uint32_t pseed = seed;
		float accD[3] = { 0, 0, 0 }, att[3] = { 1.0f, 1.0f, 1.0f };
		for (int depth = 0; depth < pp->max_depth; ++depth)
		{
			PtoHit h;
			uint32_t hinst;
			float hv[3], em[3];
			intersect(sc, bvh, o, d, pp->tmin, pp->tmax, &h, &hinst);
			++nr;
			int cont;
			if (h.prim == PTO_NONE)
			{
				hv[0] = hv[1] = hv[2] = 0.0f; em[0] = em[1] = em[2] = 0.0f; cont = 0; /* rmiss */
			}
			else cont = closest_hit(sc, &h, hinst, &pseed, o, d, hv, em);
			for (int k = 0; k < 3; ++k)
			{
				accD[k] += em[k] * att[k];
				att[k] *= hv[k];
			}
			if (!cont) break; /* depth = 100 */
		}
		for (int k = 0; k < 3; ++k) acc[k] += accD[k];
	}
	float S = (float)pp->spp_per_frame;
	out[0] = acc[0] / S; out[1] = acc[1] / S; out[2] = acc[2] / S;
	*rays += nr;
}

/* One dispatch (raytrace.rgen main over W x H) + progressive accumulation
 * (raytrace.rgen:88-95).  `gframe` is the global frame index (seeding + the
 * frame-0 pixel-centre rule); `lframe` the count of frames already accumulated in
 * `image` (the a = 1/(frame+1) of the running mean).  accum_sum != 0 keeps raw sums
 * (multi-GPU mode) instead of the running mean.  image: W*H*4 floats RGBA. */
uint64_t pto_render_frame(const PtoScene *sc, const PtoBvh *bvh, const float view[16], const float proj[16],
                          uint32_t W, uint32_t H, const PtoParams *pp, uint32_t gframe, uint32_t lframe,
                          int accum_sum, float *image, int nthreads)
{
	float invView[16], invProj[16];
	pto_inverse4(view, invView);
	pto_inverse4(proj, invProj);
	uint64_t rays = 0;
#ifdef _OPENMP
	if (nthreads > 0) omp_set_num_threads(nthreads);
#else
	(void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : rays)
	for (int64_t py = 0; py < (int64_t)H; ++py)
	{
		for (uint32_t px = 0; px < W; ++px)
		{
			float c[3];
			render_pixel(sc, bvh, invView, invProj, W, H, px, (uint32_t)py, pp, gframe, c, &rays);
			float *dst = image + 4 * ((size_t)py * W + px);
			if (accum_sum)
			{
				if (lframe == 0) { dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; }
				else { dst[0] += c[0]; dst[1] += c[1]; dst[2] += c[2]; }
			}
			else if (lframe > 0)
			{
				float a = 1.0f / (float)(lframe + 1);
				for (int k = 0; k < 3; ++k) dst[k] = dst[k] * (1.0f - a) + c[k] * a; /* mix(old,new,a) */
			}
			else { dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; }
			dst[3] = 1.0f;
		}
	}
	return rays;
}

/* The per-dispatch colour (accumHitValue / MAX_SAMPLES, raytrace.rgen:85) of selected pixels only:
 * lets a test compare a full-size GPU frame against the oracle on a pixel sample. out: n*3. */
uint64_t pto_render_pixels(const PtoScene *sc, const PtoBvh *bvh, const float view[16], const float proj[16],
                           uint32_t W, uint32_t H, const PtoParams *pp, uint32_t gframe, const uint32_t *pixels,
                           uint32_t n, float *out)
{
	float invView[16], invProj[16];
	pto_inverse4(view, invView);
	pto_inverse4(proj, invProj);
	uint64_t rays = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : rays)
	for (int64_t i = 0; i < (int64_t)n; ++i)
		render_pixel(sc, bvh, invView, invProj, W, H, pixels[i] % W, pixels[i] / W, pp, gframe, out + 3 * i, &rays);
	return rays;
}

/* Primary-ray closest hits at frame 0 (jitter = pixel centre): hits[W*H]. */
void pto_primary_hits(const PtoScene *sc, const PtoBvh *bvh, const float view[16], const float proj[16], uint32_t W,
                      uint32_t H, float tmin, float tmax, PtoHit *hits)
{
	float invView[16], invProj[16];
	pto_inverse4(view, invView);
	pto_inverse4(proj, invProj);
#pragma omp parallel for schedule(dynamic, 4)
	for (int64_t py = 0; py < (int64_t)H; ++py)
		for (uint32_t px = 0; px < W; ++px)
		{
			float o[3], d[3];
			uint32_t hinst;
			pto_camera_ray(invView, invProj, W, H, px, (uint32_t)py, 0.5f, 0.5f, o, d);
			intersect(sc, bvh, o, d, tmin, tmax, &hits[(size_t)py * W + px], &hinst);
		}
}

/* Closest hits for an arbitrary ray batch: rays = n * 6 floats (o, d). */
void pto_trace_rays(const PtoScene *sc, const PtoBvh *bvh, const float *rays, uint32_t n, float tmin, float tmax,
                    PtoHit *hits)
{
#pragma omp parallel for schedule(dynamic, 256)
	for (int64_t i = 0; i < (int64_t)n; ++i)
	{
		uint32_t hinst;
		intersect(sc, bvh, rays + 6 * i, rays + 6 * i + 3, tmin, tmax, &hits[i], &hinst);
	}
}

/* Same with the instance index of every hit (instanced scenes): inst[n]. */
void pto_trace_rays_inst(const PtoScene *sc, const PtoBvh *bvh, const float *rays, uint32_t n, float tmin, float tmax,
                         PtoHit *hits, uint32_t *inst)
{
#pragma omp parallel for schedule(dynamic, 256)
	for (int64_t i = 0; i < (int64_t)n; ++i)
		intersect(sc, bvh, rays + 6 * i, rays + 6 * i + 3, tmin, tmax, &hits[i], &inst[i]);
}

/* ------------------------------------------------------------ BVH build -- */
/* The build's own Morton-code definition (the reference has none: the driver builds
 * its BVH).  Triangle AABB -> centroid 0.5*(lo+hi); centroid bounds over the scene;
 * 10 bits per axis, x in the most significant position of each triple. */
static inline uint32_t expand10(uint32_t v)
{
	v = (v * 0x00010001u) & 0xFF0000FFu;
	v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u;
	v = (v * 0x00000005u) & 0x49249249u;
	return v;
}

void pto_morton_codes(const PtoScene *sc, uint32_t *codes, float cmin[3], float cmax[3])
{
	uint32_t nTri = sc->nIdx / 3;
	for (int k = 0; k < 3; ++k) { cmin[k] = INFINITY; cmax[k] = -INFINITY; }
	float *cent = (float *)malloc(sizeof(float) * 3 * ((size_t)nTri + 1));
	for (uint32_t p = 0; p < nTri; ++p)
	{
		float lo[3], hi[3];
		prim_bounds(sc, p, lo, hi);
		for (int k = 0; k < 3; ++k)
		{
			float c = (lo[k] + hi[k]) * 0.5f;
			cent[3 * (size_t)p + k] = c;
			if (c < cmin[k]) cmin[k] = c;
			if (c > cmax[k]) cmax[k] = c;
		}
	}
	for (uint32_t p = 0; p < nTri; ++p)
	{
		uint32_t q[3];
		for (int k = 0; k < 3; ++k)
		{
			float ext = cmax[k] - cmin[k];
			float n = ext > 0.0f ? (cent[3 * (size_t)p + k] - cmin[k]) / ext : 0.0f;
			float s = n * 1024.0f;
			if (!(s > 0.0f)) s = 0.0f;
			if (s > 1023.0f) s = 1023.0f;
			q[k] = (uint32_t)s;
		}
		codes[p] = (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
	}
	free(cent);
}

/* Stable ascending sort of primitive ids by Morton code (ties keep primitive order):
 * the order an LSD radix sort produces. */
void pto_morton_order(const uint32_t *codes, uint32_t n, uint32_t *order)
{
	uint32_t *tmp = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
	for (uint32_t i = 0; i < n; ++i) order[i] = i;
	for (int shift = 0; shift < 32; shift += 8)
	{
		uint32_t cnt[257] = { 0 };
		for (uint32_t i = 0; i < n; ++i) cnt[((codes[order[i]] >> shift) & 255u) + 1]++;
		for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
		for (uint32_t i = 0; i < n; ++i) tmp[cnt[(codes[order[i]] >> shift) & 255u]++] = order[i];
		memcpy(order, tmp, sizeof(uint32_t) * n);
	}
	free(tmp);
}

/* post.frag:9-13 + B8G8R8A8_SRGB attachment (VulkanSwapchain.cpp:8): linear float ->
 * 8-bit sRGB, the encode the framebuffer format performs.  out: W*H*4 bytes RGBA. */
void pto_srgb8(const float *rgba, uint32_t n, uint8_t *out)
{
	for (uint32_t i = 0; i < n; ++i)
	{
		for (int k = 0; k < 4; ++k)
		{
			float c = rgba[4 * (size_t)i + k];
			if (!(c > 0.0f)) c = 0.0f;
			if (c > 1.0f) c = 1.0f;
			float e = c;
			if (k < 3) e = c <= 0.0031308f ? c * 12.92f : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
			out[4 * (size_t)i + k] = (uint8_t)(e * 255.0f + 0.5f);
		}
	}
}

int pto_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}
