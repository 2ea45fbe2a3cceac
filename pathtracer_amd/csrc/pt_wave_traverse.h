// One traversal step of one wave over the quantised BVH4: shared by the per-bounce kernel
// (pt_bounce.hip) and the persistent path kernel (pt_paths.hip).  Lanes that are not in the stepping group
// simply do not call.  The per-lane stack lives in LDS as sStack[depth][lane] (conflict-free); entry 0 is
// the PT_TRAVERSAL_DONE sentinel, so popping an empty stack ends the traversal.
#pragma once
#include "pt_trace.h"

// Ray as the traversal wants it: origin, reciprocal direction (zero components nudged), shear constants of
// the watertight test, kpack = kz | (rx<0)<<6 | (ry<0)<<7 | (rz<0)<<8 (upper bits are the caller's).
struct PtLaneRay
{
	float ox, oy, oz, ix, iy, iz, Sx, Sy, Sz;
	uint32_t kpack;
};

// Prepares a ray for traversal (runs at full lane utilisation in the staging phase: the divisions live here).
// Woop's winding swap for d[kz] < 0 is omitted: it negates U, V, W, det and T exactly and leaves t, u, v
// unchanged, so results equal the oracle's (which keeps the swap) bit for bit.
template <bool kShareReciprocal = false>
__device__ __forceinline__ PtLaneRay pt_prepare_ray(float3x o, float3x d)
{
	PtLaneRay r;
	float ad0 = fabsf(d.x), ad1 = fabsf(d.y), ad2 = fabsf(d.z);
	uint32_t kz = 0;
	float am = ad0;
	if (ad1 > am)
	{
		kz = 1;
		am = ad1;
	}
	if (ad2 > am) kz = 2;
	float dkz = kz == 0 ? d.x : (kz == 1 ? d.y : d.z);
	float dkx = kz == 0 ? d.y : (kz == 1 ? d.z : d.x);
	float dky = kz == 0 ? d.z : (kz == 1 ? d.x : d.y);
	// Three IEEE divisions by the same number.  hipcc expands each into v_div_scale x2, v_rcp, two Newton steps on the
	// reciprocal, three on the quotient, v_div_fmas, v_div_fixup (11 instructions); the reciprocal part depends on the
	// denominator only.  When every lane's operands are far from the exponent limits (|dkz| and the numerators within
	// 2^+-40: then v_div_scale scales nothing, v_div_fmas is a plain fma and v_div_fixup returns its input) the same
	// instructions are issued with that part shared -- 18 instead of 33, bit-identical quotients; otherwise, and always where
	// lanes call this one by one (kShareReciprocal == false), the plain divisions.
#ifdef PT_PROBE_RCPDET
	// speed probe only (changes results): Sz = 1 / d[kz] once, Sx = d[kx] * Sz, Sy = d[ky] * Sz
	if (kShareReciprocal && pt_all_in_division_range(dkz, dkx, dky))
	{
		r.Sz = PtSharedDivisor(dkz).quotient(1.0f);
		r.Sx = dkx * r.Sz;
		r.Sy = dky * r.Sz;
	}
#else
	if (kShareReciprocal && pt_all_in_division_range(dkz, dkx, dky))
	{
		const PtSharedDivisor by(dkz);
		r.Sx = by.quotient(dkx);
		r.Sy = by.quotient(dky);
		r.Sz = by.quotient(1.0f);
	}
#endif
	else
	{
		r.Sx = dkx / dkz;
		r.Sy = dky / dkz;
		r.Sz = 1.0f / dkz;
	}
	// box test only: zero components nudged so that 0 * inf cannot occur (interval stays huge)
	float dcx = fabsf(d.x) > 1e-30f ? d.x : copysignf(1e-30f, d.x);
	float dcy = fabsf(d.y) > 1e-30f ? d.y : copysignf(1e-30f, d.y);
	float dcz = fabsf(d.z) > 1e-30f ? d.z : copysignf(1e-30f, d.z);
	// v_rcp_f32 (1 ulp), not an IEEE division: the reciprocals only feed the box test, whose far plane carries a
	// 1e-6 relative slack (8 ulp) for exactly this kind of error; |dc| >= 1e-30 keeps operand and result normal.
	// Saves three of the six divisions of the staging phase (a division is ~13 average VALU slots).
	r.ix = __builtin_amdgcn_rcpf(dcx);
	r.iy = __builtin_amdgcn_rcpf(dcy);
	r.iz = __builtin_amdgcn_rcpf(dcz);
	r.kpack = kz | ((r.ix < 0.0f) ? 64u : 0u) | ((r.iy < 0.0f) ? 128u : 0u) | ((r.iz < 0.0f) ? 256u : 0u);
	r.ox = o.x;
	r.oy = o.y;
	r.oz = o.z;
	return r;
}

// Where a node step finds its nodes.
//   PT_NODES_GLOBAL  the 64-byte records in memory: four divergent-address global loads (each of which costs the frame as
//                    much as ~10 VALU instructions)
//   PT_NODES_PACKED  the same records copied into LDS by the caller (node arrays up to PT_LDS_NODES_PACKED): four ds_read_b128
//   PT_NODES_PLANES  decoded records in LDS (node arrays up to PT_LDS_NODES_PLANES; layout below): the 24 quantised planes are
//                    stored as the floats (float)q the packed path converts them to on every visit -- same fma, same
//                    result, 24 v_cvt_f32_ubyte fewer -- near and far planes are picked by address instead of by select,
//                    and an empty child slot is a box no ray can hit (lo = +inf, hi = -inf) instead of a test on its reference.
enum
{
	PT_NODES_GLOBAL = 0,
	PT_NODES_PACKED = 1,
	PT_NODES_PLANES = 2
};
#define PT_LDS_NODES_PACKED 16u
#ifndef PT_LDS_NODES_PLANES
#define PT_LDS_NODES_PLANES 8u
#endif
// decoded record, in 16-byte units: [0] origin.xyz, scaleX  [1] child references  [2] scaleY, scaleZ, -, -
// [3..5] low planes x, y, z (one float per child)  [6..8] high planes x, y, z
#define PT_PLANE_REC 9u

// Decodes node `n` of the packed array into word `k` (0 .. 4*PT_PLANE_REC-1) of its decoded record.
__device__ __forceinline__ uint32_t pt_plane_record_word(const PtNode4 *nodes, uint32_t n, uint32_t k)
{
	const uint32_t *w = reinterpret_cast<const uint32_t *>(nodes + n);  // origin[3] scaleX child[4] qlo[3] qhi[3] scaleY scaleZ
	if (k < 8u) return w[k];
	if (k < 12u) return k == 8u ? w[14] : (k == 9u ? w[15] : 0u);
	const uint32_t p = (k - 12u) >> 2, c = (k - 12u) & 3u;  // p: lo x, y, z, hi x, y, z
	if (w[4u + c] == PT_EMPTY_CHILD) return __float_as_uint(p < 3u ? INFINITY : -INFINITY);
	return __float_as_uint((float)((w[8u + p] >> (8u * c)) & 0xFFu));
}


// ------------------------------------------------------------------- cooperative node fetch ----
// A/B switch (round 3, OFF: measured slower on every BASELINE config; profiles/r03h3_ab_coop.log).  The
// counters of a tree-form k_paths show a busy vector L1 (viking_room: TA busy 0.70, 0.91 line look-ups per cycle and CU,
// profiles/r03h2_viking_ta_pmc.log) and scripts/ubench/tcp_rate.hip prices its access patterns: a wave instruction whose 64
// lanes address 64 different lines costs the TA/TCP ~44 cycles whether the lines hit or not -- the four loads of a node step
// 177 cycles for 4 KB -- while an instruction whose lanes 4k .. 4k+3 read the four 16-byte pieces of ONE 64-byte record
// costs 18 (hit) to 37 (miss): the texture addresser merges the lanes of a quad that fall into one line into one tag
// look-up.  So the lanes of a quad fetch each other's nodes: instruction i has lane 4k+j load piece j of the node lane 4k+i
// wants (address by a DPP quad broadcast of `cur`), and a 4x4 transposition inside the quad -- two butterfly stages of
// v_cndmask_b32 with a DPP-permuted operand, 32 instructions for the 16 dwords -- hands every lane the four pieces of its
// own node.  Same bytes, same arithmetic after it, images bit-identical.  Result: viking_room -11.5 %, teapot grid -14 %,
// 1 M soup -19 %, 5 M soup +-0, 20 M soup +5.7 %, 100 M soup -6.4 %: the look-ups were not what those kernels wait for --
// the 48 instructions of the exchange lengthen each wave's dependent chain by more than the cheaper fetch shortens it.
// Every lane of the wave must call (EXEC full: a DPP operand reads the other lanes' registers).
#ifndef PT_NODE_COOP
#define PT_NODE_COOP 0
#endif
struct PtNodeRegs  // a 64-byte node record in registers
{
	uint4 n[4];
};

// One butterfly stage over two registers of four dwords each: lanes whose index has the stage's bit clear keep a and take
// b from the partner lane, lanes with the bit set keep b and take a from the partner.  keepA = lanes with the bit clear.
#define PT_QUAD_STAGE(PERM, KEEP_A_LO, KEEP_B_LO)                                                                      \
	__device__ __forceinline__ void pt_quad_stage_##KEEP_A_LO(uint4 &a, uint4 &b)                                      \
	{                                                                                                                  \
		uint4 na, nb;                                                                                                  \
		asm volatile("s_nop 1\n\t"                                                                                     \
		             "s_mov_b32 vcc_lo, " #KEEP_A_LO "\n\ts_mov_b32 vcc_hi, " #KEEP_A_LO "\n\t"                        \
		             "v_cndmask_b32_dpp %0, %12, %8, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                      \
		             "v_cndmask_b32_dpp %1, %13, %9, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                      \
		             "v_cndmask_b32_dpp %2, %14, %10, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
		             "v_cndmask_b32_dpp %3, %15, %11, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
		             "s_mov_b32 vcc_lo, " #KEEP_B_LO "\n\ts_mov_b32 vcc_hi, " #KEEP_B_LO "\n\t"                        \
		             "v_cndmask_b32_dpp %4, %8, %12, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                      \
		             "v_cndmask_b32_dpp %5, %9, %13, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                      \
		             "v_cndmask_b32_dpp %6, %10, %14, vcc " PERM " row_mask:0xf bank_mask:0xf\n\t"                     \
		             "v_cndmask_b32_dpp %7, %11, %15, vcc " PERM " row_mask:0xf bank_mask:0xf"                         \
		             : "=&v"(na.x), "=&v"(na.y), "=&v"(na.z), "=&v"(na.w), "=&v"(nb.x), "=&v"(nb.y), "=&v"(nb.z),      \
		               "=&v"(nb.w)                                                                                     \
		             : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w)                  \
		             : "vcc");                                                                                         \
		a = na;                                                                                                        \
		b = nb;                                                                                                        \
	}
PT_QUAD_STAGE("quad_perm:[1,0,3,2]", 0x55555555, 0xaaaaaaaa)  // partner = lane ^ 1
PT_QUAD_STAGE("quad_perm:[2,3,0,1]", 0x33333333, 0xcccccccc)  // partner = lane ^ 2
#undef PT_QUAD_STAGE

// R[i] of lane 4k+j holds piece j of record i of the quad  ->  R[j] of lane 4k+i holds piece j of record i
__device__ __forceinline__ void pt_quad_transpose(uint4 (&R)[4])
{
	pt_quad_stage_0x55555555(R[0], R[1]);
	pt_quad_stage_0x55555555(R[2], R[3]);
	pt_quad_stage_0x33333333(R[0], R[2]);
	pt_quad_stage_0x33333333(R[1], R[3]);
}

// The node `cur` of every lane that is at an inner node (bit 31 clear), fetched by the lane's quad; the other lanes get
// whatever.  Called by ALL 64 lanes.
__device__ __forceinline__ void pt_node_fetch_coop(const PtNode4 *nodes, uint32_t cur, int lane, PtNodeRegs &out)
{
	uint4 (&R)[4] = out.n;
	const uint32_t piece = (uint32_t)lane & 3u;
#define PT_COOP_LOAD(i)                                                                                              \
	{                                                                                                                \
		const uint32_t ci = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, (i) * 0x55, 0xF, 0xF, true); /* lane 4k+i's */ \
		/* (a defined value at no cost for the quads whose lane i is not at a node: nothing reads it) */               \
		asm volatile("" : "=v"(R[i].x), "=v"(R[i].y), "=v"(R[i].z), "=v"(R[i].w));                                   \
		if (!(ci & PT_LEAF_FLAG)) R[i] = reinterpret_cast<const uint4 *>(nodes + ci)[piece];                         \
	}
	PT_COOP_LOAD(0) PT_COOP_LOAD(1) PT_COOP_LOAD(2) PT_COOP_LOAD(3)
#undef PT_COOP_LOAD
	pt_quad_transpose(R);
}

// kPre: `pre` holds the node record, fetched already (pt_node_fetch_coop; kNodeMode == PT_NODES_GLOBAL only)
template <int kStackLds, int kNodeMode = PT_NODES_GLOBAL, bool kPre = false>
__device__ __forceinline__ void pt_node_step(const PtSceneDev &sc, const PtLaneRay &ray, float tmin, const PtBest &best,
                                             uint32_t &cur, int &sp, uint32_t (*sStack)[PT_WAVE], int lane,
                                             uint32_t *mySpill, uint32_t spillStride, const uint4 *sNodes = nullptr,
                                             const PtNodeRegs &pre = PtNodeRegs())
{
	const float ox = ray.ox, oy = ray.oy, oz = ray.oz, ix = ray.ix, iy = ray.iy, iz = ray.iz;
	const uint32_t kpack = ray.kpack;
	// the entry that is popped if no child is hit; read early so its latency hides under the box tests
	const bool fastStack = sp <= kStackLds - 3;
	uint32_t top = sStack[sp - 1 < kStackLds ? sp - 1 : 0][lane];
	const bool negx = kpack & 64u, negy = kpack & 128u, negz = kpack & 256u;
	uint32_t ref[4];
	float tn[4];
	if (kNodeMode == PT_NODES_PLANES)
	{
		const float4 *rec = reinterpret_cast<const float4 *>(sNodes) + cur * PT_PLANE_REC;
		const float4 h0 = rec[0], h2 = rec[2];
		const uint4 r4 = *reinterpret_cast<const uint4 *>(rec + 1);
		const int dx = negx ? 3 : 0, dy = negy ? 3 : 0, dz = negz ? 3 : 0;
		const float4 qnx = rec[3 + dx], qfx = rec[6 - dx];
		const float4 qny = rec[4 + dy], qfy = rec[7 - dy];
		const float4 qnz = rec[5 + dz], qfz = rec[8 - dz];
		const float ax = h0.w * ix, ay = h2.x * iy, az = h2.y * iz;
		const float bx = (h0.x - ox) * ix, by = (h0.y - oy) * iy, bz = (h0.z - oz) * iz;
		ref[0] = r4.x;
		ref[1] = r4.y;
		ref[2] = r4.z;
		ref[3] = r4.w;
		const float vnx[4] = { qnx.x, qnx.y, qnx.z, qnx.w }, vfx[4] = { qfx.x, qfx.y, qfx.z, qfx.w };
		const float vny[4] = { qny.x, qny.y, qny.z, qny.w }, vfy[4] = { qfy.x, qfy.y, qfy.z, qfy.w };
		const float vnz[4] = { qnz.x, qnz.y, qnz.z, qnz.w }, vfz[4] = { qfz.x, qfz.y, qfz.z, qfz.w };
#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			float tnx = __builtin_fmaf(vnx[c], ax, bx);
			float tny = __builtin_fmaf(vny[c], ay, by);
			float tnz = __builtin_fmaf(vnz[c], az, bz);
			float tfx = __builtin_fmaf(vfx[c], ax, bx);
			float tfy = __builtin_fmaf(vfy[c], ay, by);
			float tfz = __builtin_fmaf(vfz[c], az, bz);
			float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
			float t1 = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
			// an empty slot has t0 = +inf and t1 = -inf whatever the ray (a = scale/d is finite and not zero)
			tn[c] = (t0 <= t1 * 1.000001f) ? t0 : INFINITY;
		}
	}
	else
	{
		uint4 n0, n1, n2, n3;
		if (kNodeMode == PT_NODES_PACKED)
		{
			n0 = sNodes[4u * cur + 0u];
			n1 = sNodes[4u * cur + 1u];
			n2 = sNodes[4u * cur + 2u];
			n3 = sNodes[4u * cur + 3u];
		}
		else if (kPre)
		{
			n0 = pre.n[0];
			n1 = pre.n[1];
			n2 = pre.n[2];
			n3 = pre.n[3];
		}
		else
		{
			const uint4 *np = reinterpret_cast<const uint4 *>(sc.nodes + cur);
			n0 = np[0];
			n1 = np[1];
			n2 = np[2];
			n3 = np[3];
		}
		float sx = __uint_as_float(n0.w), sy = __uint_as_float(n3.z), sz = __uint_as_float(n3.w);
		float ax = sx * ix, ay = sy * iy, az = sz * iz;
		float bx = (__uint_as_float(n0.x) - ox) * ix, by = (__uint_as_float(n0.y) - oy) * iy,
		      bz = (__uint_as_float(n0.z) - oz) * iz;
		// n2 = qlo[0..2], qhi[0]; n3 = qhi[1..2], scaleY, scaleZ
		uint32_t nqx = negx ? n2.w : n2.x, fqx = negx ? n2.x : n2.w;
		uint32_t nqy = negy ? n3.x : n2.y, fqy = negy ? n2.y : n3.x;
		uint32_t nqz = negz ? n3.y : n2.z, fqz = negz ? n2.z : n3.y;
		ref[0] = n1.x;
		ref[1] = n1.y;
		ref[2] = n1.z;
		ref[3] = n1.w;
#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			float tnx = __builtin_fmaf((float)((nqx >> (8 * c)) & 0xFFu), ax, bx);
			float tny = __builtin_fmaf((float)((nqy >> (8 * c)) & 0xFFu), ay, by);
			float tnz = __builtin_fmaf((float)((nqz >> (8 * c)) & 0xFFu), az, bz);
			float tfx = __builtin_fmaf((float)((fqx >> (8 * c)) & 0xFFu), ax, bx);
			float tfy = __builtin_fmaf((float)((fqy >> (8 * c)) & 0xFFu), ay, by);
			float tfz = __builtin_fmaf((float)((fqz >> (8 * c)) & 0xFFu), az, bz);
			float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
			float t1 = fminf(fminf(tfx, tfy), fminf(tfz, best.t));
			// an empty slot is the inverted box lo = 255, hi = 0 on every axis: its entry lies 255 steps behind its exit and
			// the test fails by itself (the stray case -- a ray so far away that 255 steps vanish in the 1e-6 slack -- pushes
			// PT_EMPTY_CHILD, which the leaf step skips)
			tn[c] = (t0 <= t1 * 1.000001f) ? t0 : INFINITY;
		}
	}
	// sort the (entry distance, reference) pairs ascending: 5 compare-exchanges; missed children (inf) sink
#define PT_CE(a, b)                           \
	{                                         \
		bool sw_ = tn[b] < tn[a];             \
		float tl_ = sw_ ? tn[b] : tn[a];      \
		float th_ = sw_ ? tn[a] : tn[b];      \
		uint32_t rl_ = sw_ ? ref[b] : ref[a]; \
		uint32_t rh_ = sw_ ? ref[a] : ref[b]; \
		tn[a] = tl_;                          \
		tn[b] = th_;                          \
		ref[a] = rl_;                         \
		ref[b] = rh_;                         \
	}
#ifndef PT_SORT_FULL
	// nearest child to slot 0 only; the pending ones are pushed in slot order (exact when <= 2 children are hit)
	PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2)
#else
	PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#endif
#undef PT_CE
	const bool h0 = tn[0] < INFINITY, h1 = tn[1] < INFINITY, h2 = tn[2] < INFINITY, h3 = tn[3] < INFINITY;
	if (fastStack)
	{
		// branch-free: far children are stored unconditionally above the top, the stack pointer only
		// moves past the ones that were hit (far -> near, so the nearest pending child is popped first)
		sStack[sp][lane] = ref[3];
		sp += h3 ? 1 : 0;
		sStack[sp][lane] = ref[2];
		sp += h2 ? 1 : 0;
		sStack[sp][lane] = ref[1];
		sp += h1 ? 1 : 0;
		cur = h0 ? ref[0] : top;
		sp -= h0 ? 0 : 1;
	}
	else
	{
#define PT_PUSH(v)                                                                                       \
	{                                                                                                    \
		if (sp < kStackLds) sStack[sp][lane] = (v);                                                      \
		else if (sp < kStackLds + PT_SPILL_STACK) mySpill[(size_t)(sp - kStackLds) * spillStride] = (v); \
		++sp;                                                                                            \
	}
		if (h3) PT_PUSH(ref[3]);
		if (h2) PT_PUSH(ref[2]);
		if (h1) PT_PUSH(ref[1]);
#undef PT_PUSH
		if (h0) cur = ref[0];
		else
		{
			--sp;
			cur = sp < kStackLds ? sStack[sp][lane]
			                     : (sp < kStackLds + PT_SPILL_STACK ? mySpill[(size_t)(sp - kStackLds) * spillStride]
			                                                        : PT_EMPTY_CHILD);
		}
	}
}

// ---------------------------------------------------------------------------------- entry node ----
// The top of the tree flattened into ONE wide node (pt_build.h, ptBuildEntryNode) that k_paths tests in its STAGING phase,
// where all 64 lanes hold a fresh ray -- the first node visits of every ray at full lane utilisation instead of inside the
// traversal loop at ~0.68, and as one wide step instead of two or three 4-wide ones (Cornell: root + the two room-sized
// nodes every ray is inside = 3.04 visits of 105 instructions -> one 16-wide test).  Record, in 16-byte units (W = width):
//   [0] origin.xyz, -   [1] children, -, -, -   [2 .. 2+W/4) child references
//   then six plane arrays of W floats: lo x, lo y, lo z, hi x, hi y, hi z -- distances of the planes from the origin (the
//   fma of the decoded node records above with a scale of one, near/far picked by address, an empty slot is lo = +inf /
//   hi = -inf).  Leaf children carry the exact (padded) boxes of their triangles, not the node grid's (pt_bvh_build.hip).
// The children keep their ordinary references (leaves, quad leaves, inner nodes of the ordinary array), so what follows the
// entry test is the ordinary traversal; the flattened nodes stay in the array for rays that do not use the entry node.
// (PT_ENTRY_WIDTH, PT_ENTRY_REC: pt_device_common.h)
// Tests `ray` against the entry node (best hit so far: none, so the interval is [tmin, tmax]) and returns the set of
// children it hits as a bit mask: child c is bit PT_ENTRY_WIDTH-1-c.  The lane that traverses the ray keeps the mask in a
// register and visits the children one after the other, lowest bit first (pt_entry_next), each with the ordinary stack
// discipline -- nothing is stored per child, and there is no limit on how many a ray may hit.
// kNearest: bits 16..19 of the result also name the hit child the ray enters FIRST (bit position in the mask), for
// pt_entry_first -- three more instructions per child; worth it where the children are subtrees (the order decides how
// much the later ones are culled), not where they are leaves.
template <bool kNearest = false>
__device__ __forceinline__ uint32_t pt_entry_test(const uint4 *e, const PtLaneRay &ray, float tmin, float tmax)
{
	constexpr int W = PT_ENTRY_WIDTH, G = W / 4;
	const float4 *rec = reinterpret_cast<const float4 *>(e);
	const float4 h0 = rec[0];
	const bool negx = ray.kpack & 64u, negy = ray.kpack & 128u, negz = ray.kpack & 256u;
	const float ax = ray.ix, ay = ray.iy, az = ray.iz;  // planes are distances from the origin, not grid indices
	const float bx = (h0.x - ray.ox) * ray.ix, by = (h0.y - ray.oy) * ray.iy, bz = (h0.z - ray.oz) * ray.iz;
	const float4 *planes = rec + 2 + G;
	const int nx = negx ? 3 * G : 0, ny = negy ? 3 * G : 0, nz = negz ? 3 * G : 0;
	// The 1e-6 slack of the far side (node step: t0 <= t1 * 1.000001f) is folded into the far planes' coefficients -- the
	// factor is positive, so it commutes with the min up to one rounding per operand, which the slack itself (8 ulp) and the
	// padding of the boxes cover -- one multiplication per child less.  Without the nearest key the outcome goes into the
	// set as the sign bit of t1 - t0 (set: missed; neither operand is a NaN, they are never infinities of the same sign, and
	// equal operands give +0), one v_alignbit per child instead of a compare and a two-instruction insert.
	const float slack = 1.000001f;
	const float axf = ax * slack, ayf = ay * slack, azf = az * slack;
	const float bxf = bx * slack, byf = by * slack, bzf = bz * slack, tmaxf = tmax * slack;
	uint32_t mask = 0, nearest = 0xFFFFFFFFu;
#pragma unroll
	for (int g = 0; g < G; ++g)
	{
		// one group of four children at a time: left alone the compiler issues the loads of all groups first (28 x 16 bytes
		// per lane for 16 children) and the kernel loses a wave of occupancy to the registers that hold them
		asm volatile("" ::: "memory");
		const float4 qnx = planes[0 * G + nx + g], qfx = planes[3 * G - nx + g];
		const float4 qny = planes[1 * G + ny + g], qfy = planes[4 * G - ny + g];
		const float4 qnz = planes[2 * G + nz + g], qfz = planes[5 * G - nz + g];
		const float vnx[4] = { qnx.x, qnx.y, qnx.z, qnx.w }, vfx[4] = { qfx.x, qfx.y, qfx.z, qfx.w };
		const float vny[4] = { qny.x, qny.y, qny.z, qny.w }, vfy[4] = { qfy.x, qfy.y, qfy.z, qfy.w };
		const float vnz[4] = { qnz.x, qnz.y, qnz.z, qnz.w }, vfz[4] = { qfz.x, qfz.y, qfz.z, qfz.w };
#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			float tnx = __builtin_fmaf(vnx[c], ax, bx);
			float tny = __builtin_fmaf(vny[c], ay, by);
			float tnz = __builtin_fmaf(vnz[c], az, bz);
			float tfx = __builtin_fmaf(vfx[c], axf, bxf);
			float tfy = __builtin_fmaf(vfy[c], ayf, byf);
			float tfz = __builtin_fmaf(vfz[c], azf, bzf);
			float t0 = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
			float t1 = fminf(fminf(tfx, tfy), fminf(tfz, tmaxf));
			if (!kNearest)
			{
				mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(t1 - t0), 31);  // (mask << 1) | missed
				continue;
			}
			const bool hit = t0 <= t1;  // the node step's test (slack folded in above); an empty slot fails it whatever the ray
			mask = (mask + mask) + (hit ? 1u : 0u);
			if (kNearest)
			{
				// entry distances are >= tmin >= 0, so their bit patterns order like unsigned integers; the low four bits
				// carry the child's bit position instead of mantissa
				const uint32_t key = hit ? ((__float_as_uint(t0) & ~15u) | (uint32_t)(W - 1 - (4 * g + c))) : 0xFFFFFFFFu;
				nearest = key < nearest ? key : nearest;
			}
		}
	}
	if (!kNearest) mask = ~mask & ((1u << W) - 1u);
	static_assert(!kNearest || W <= 16, "four bits for the nearest child");
	return kNearest ? (mask | ((nearest & 15u) << 16)) : mask;
}

// The reference of the child a ray enters first (result `m` of pt_entry_test<true>, at least one child hit); `pend` gets the
// other hit children.
__device__ __forceinline__ uint32_t pt_entry_first(const uint4 *e, uint32_t m, uint32_t &pend)
{
	const uint32_t bit = (m >> 16) & 15u;
	pend = (m & 0xFFFFu) & ~(1u << bit);
	return reinterpret_cast<const uint32_t *>(e + 2)[(uint32_t)(PT_ENTRY_WIDTH - 1) - bit];
}

// Entry distance of `ray` into child `child` of the entry node: the near-plane half of pt_entry_test, same arithmetic.
__device__ __forceinline__ float pt_entry_child_t0(const uint4 *e, uint32_t child, const PtLaneRay &ray, float tmin)
{
	constexpr uint32_t W = PT_ENTRY_WIDTH, G = W / 4;
	const float4 h0 = reinterpret_cast<const float4 *>(e)[0];
	const float *planes = reinterpret_cast<const float *>(e + 2 + G);
	const float bx = (h0.x - ray.ox) * ray.ix, by = (h0.y - ray.oy) * ray.iy, bz = (h0.z - ray.oz) * ray.iz;
	const float vnx = planes[((ray.kpack & 64u) ? 3u * W : 0u) + child];
	const float vny = planes[((ray.kpack & 128u) ? 4u * W : W) + child];
	const float vnz = planes[((ray.kpack & 256u) ? 5u * W : 2u * W) + child];
	const float tnx = __builtin_fmaf(vnx, ray.ix, bx), tny = __builtin_fmaf(vny, ray.iy, by), tnz = __builtin_fmaf(vnz, ray.iz, bz);
	return fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
}

// The reference of the next child in `pend` (not zero), which it removes from the set.
__device__ __forceinline__ uint32_t pt_entry_next(const uint4 *e, uint32_t &pend)
{
	const uint32_t bit = (uint32_t)__builtin_ctz(pend);
	pend &= pend - 1u;
	return reinterpret_cast<const uint32_t *>(e + 2)[(uint32_t)(PT_ENTRY_WIDTH - 1) - bit];
}

template <int kStackLds>
__device__ __forceinline__ uint32_t pt_stack_pop(int &sp, uint32_t (*sStack)[PT_WAVE], int lane, uint32_t *mySpill,
                                                 uint32_t spillStride)
{
	--sp;
	return sp < kStackLds ? sStack[sp][lane]
	                      : (sp < kStackLds + PT_SPILL_STACK ? mySpill[(size_t)(sp - kStackLds) * spillStride]
	                                                         : PT_EMPTY_CHILD);
}

template <int kStackLds>
__device__ __forceinline__ void pt_stack_push(uint32_t v, int &sp, uint32_t (*sStack)[PT_WAVE], int lane,
                                              uint32_t *mySpill, uint32_t spillStride)
{
	if (sp < kStackLds) sStack[sp][lane] = v;
	else if (sp < kStackLds + PT_SPILL_STACK) mySpill[(size_t)(sp - kStackLds) * spillStride] = v;
	++sp;
}

// curInst / bestInst: instance the lane is inside and instance of the best hit (two-level scenes; ties at equal t
// go to the lower instance index, then the lower primitive id).  Single-level callers pass 0 and a dummy.
// kQuads: the tree may hold quad leaf references (PtQuad64; k_paths on scenes made of quad faces).
// The triangles (or the quad) of leaf reference `cur` against the ray: the leaf step without its stack pop.
// ldsQuads (kLdsQuads): the scene's quads in LDS, PT_LDS_QUADS records of twelve floats (v0 .. v3 in x, y, z order) followed by
// their primitive id pairs -- ONE copy, not three rotated ones: the lane picks the components in its ray's axis order by
// address, so the values are the ones the rotated records in memory hold.
#define PT_LDS_QUADS 16u
#define PT_LDS_QUAD_WORDS (PT_LDS_QUADS * 14u)
template <bool kQuads = false, bool kLdsQuads = false>
__device__ __forceinline__ void pt_leaf_test(const PtSceneDev &sc, const PtLaneRay &ray, float tmin, PtBest &best,
                                             const uint32_t cur, uint32_t &triTests, uint32_t curInst = 0u,
                                             uint32_t *bestInst = nullptr, const float *ldsQuads = nullptr)
{
	const float ox = ray.ox, oy = ray.oy, oz = ray.oz, Sx = ray.Sx, Sy = ray.Sy, Sz = ray.Sz;
	const uint32_t kpack = ray.kpack;
	// Second half of the Woop/Benthin/Wald watertight test (pt_tri_test without the selects), in two parts.
	// inside(): the edge functions U, V, W of the sheared vertices (Ax, Ay) (Bx, By) (Cx, Cy) are recomputed in double
	// precision when one of them is zero; true when the ray passes inside the triangle (det = U + V + W, not zero).
	auto inside = [](float &U, float &V, float &W, float &det, float Ax, float Ay, float Bx, float By, float Cx, float Cy) {
		if (U == 0.0f || V == 0.0f || W == 0.0f)
		{
			// rare (a ray through an edge or a vertex).  The empty asm pins the conversions inside the branch: left alone the
			// compiler hoists part of the double-precision work (v_cvt_f64_f32, v_mul_f64: half rate) into every leaf step
			asm volatile("" : "+v"(Ax), "+v"(Ay), "+v"(Bx), "+v"(By), "+v"(Cx), "+v"(Cy));
			U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
			V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
			W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
		}
		bool anyNeg = U < 0.0f || V < 0.0f || W < 0.0f, anyPos = U > 0.0f || V > 0.0f || W > 0.0f;
		det = (U + V) + W;
		return !(anyNeg && anyPos) && det != 0.0f;
	};
	// accept(): distance of the hit; it replaces the best one when it is closer, or equally far with the lower primitive
	// (instance) id.
	auto accept = [&](float U, float V, float W, float det, float Akz, float Bkz, float Ckz, uint32_t prim) {
		float Az = Sz * Akz, Bz = Sz * Bkz, Cz = Sz * Ckz;
		float T = (U * Az + V * Bz) + W * Cz;
		float t = T / det;
		bool tie = t == best.t && best.prim != PT_NO_HIT;
		if (bestInst) tie = tie && (curInst < *bestInst || (curInst == *bestInst && prim < best.prim));
		else tie = tie && prim < best.prim;
		if (t > tmin && (t < best.t || tie))
		{
			if (bestInst) *bestInst = curInst;
			best.t = t;
			best.prim = prim;
			best.V = V;
			best.W = W;
			best.det = det;
		}
	};
	if (cur != PT_EMPTY_CHILD)
	{
		const uint32_t kz = kpack & 3u;
		// origin in the triangle plane's axis order (kx, ky, kz)
		float rox = kz == 0 ? oy : (kz == 1 ? oz : ox);
		float roy = kz == 0 ? oz : (kz == 1 ? ox : oy);
		float roz = kz == 0 ? ox : (kz == 1 ? oy : oz);
		uint32_t first = cur & 0x0FFFFFFFu;
		// count field 7 is a quad reference only in a tree that HAS quad leaves (wave-uniform test); without them it is an
		// ordinary leaf of eight triangles (PT_ACCEL_LEAF_SIZE(8), for which pt_build_accel builds no quads)
		const bool quad = kQuads && sc.nQuads != 0u && ((cur >> 28) & 7u) == PT_LEAF_QUAD;
		if (!quad)
		{
			uint32_t cnt = ((cur >> 28) & 7u) + 1u;
			triTests += cnt;
			// three planes: the record of the ray's dominant axis holds the coordinates in (kx, ky, kz) order already; one plane
			// (big scenes, whose traversal waits for memory: a third of the bytes built and kept): permuted here (wave-uniform test)
			const bool onePlane = sc.triPlanes == 1u;
			const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + (onePlane ? (size_t)0 : (size_t)kz * sc.nTris) + first);
			for (uint32_t i = 0; i < cnt; ++i, tp += 3)
			{
				float4 a = tp[0], b = tp[1], c = tp[2];
				if (onePlane)
				{
					// (x y z x) (y z x y) (z prim - -)  ->  vertex j as (kx, ky, kz)
					const float v0x = a.x, v0y = a.y, v0z = a.z, v1x = a.w, v1y = b.x, v1z = b.y, v2x = b.z, v2y = b.w, v2z = c.x;
					a.x = kz == 0 ? v0y : (kz == 1 ? v0z : v0x);
					a.y = kz == 0 ? v0z : (kz == 1 ? v0x : v0y);
					a.z = kz == 0 ? v0x : (kz == 1 ? v0y : v0z);
					a.w = kz == 0 ? v1y : (kz == 1 ? v1z : v1x);
					b.x = kz == 0 ? v1z : (kz == 1 ? v1x : v1y);
					b.y = kz == 0 ? v1x : (kz == 1 ? v1y : v1z);
					b.z = kz == 0 ? v2y : (kz == 1 ? v2z : v2x);
					b.w = kz == 0 ? v2z : (kz == 1 ? v2x : v2y);
					c.x = kz == 0 ? v2x : (kz == 1 ? v2y : v2z);
				}
				float Akx = a.x - rox, Aky = a.y - roy, Akz = a.z - roz;
				float Bkx = a.w - rox, Bky = b.x - roy, Bkz = b.y - roz;
				float Ckx = b.z - rox, Cky = b.w - roy, Ckz = c.x - roz;
				float Ax = Akx - Sx * Akz, Ay = Aky - Sy * Akz;
				float Bx = Bkx - Sx * Bkz, By = Bky - Sy * Bkz;
				float Cx = Ckx - Sx * Ckz, Cy = Cky - Sy * Ckz;
				float U = Cx * By - Cy * Bx;
				float V = Ax * Cy - Ay * Cx;
				float W = Bx * Ay - By * Ax;
				float det;
				if (inside(U, V, W, det, Ax, Ay, Bx, By, Cx, Cy)) accept(U, V, W, det, Akz, Bkz, Ckz, __float_as_uint(c.y));
			}
		}
		else
		{
			// quad leaf: triangle A = (v0, v1, v2), triangle B = (v0, v2, v3); four vertices sheared once
			triTests += 2u;
			float4 a, b, c;
			uint2 prims;
			if (kLdsQuads)
			{
				const uint32_t kx = kz == 2u ? 0u : kz + 1u, ky = kz == 0u ? 2u : kz - 1u;
				const float *qx = ldsQuads + first * 12u + kx, *qy = ldsQuads + first * 12u + ky, *qz = ldsQuads + first * 12u + kz;
				a = make_float4(qx[0], qy[0], qz[0], qx[3]);
				b = make_float4(qy[3], qz[3], qx[6], qy[6]);
				c = make_float4(qz[6], qx[9], qy[9], qz[9]);
				prims = reinterpret_cast<const uint2 *>(ldsQuads + PT_LDS_QUADS * 12u)[first];
			}
			else
			{
				const float4 *qp = reinterpret_cast<const float4 *>(sc.quads + (size_t)kz * sc.nQuads + first);
				a = qp[0];
				b = qp[1];
				c = qp[2];
				prims = *reinterpret_cast<const uint2 *>(qp + 3);
			}
			float P0x = a.x - rox, P0y = a.y - roy, P0z = a.z - roz;
			float P1x = a.w - rox, P1y = b.x - roy, P1z = b.y - roz;
			float P2x = b.z - rox, P2y = b.w - roy, P2z = c.x - roz;
			float P3x = c.y - rox, P3y = c.z - roy, P3z = c.w - roz;
			float X0 = P0x - Sx * P0z, Y0 = P0y - Sy * P0z;
			float X1 = P1x - Sx * P1z, Y1 = P1y - Sy * P1z;
			float X2 = P2x - Sx * P2z, Y2 = P2y - Sy * P2z;
			float X3 = P3x - Sx * P3z, Y3 = P3y - Sy * P3z;
			// A: (A, B, C) = (0, 1, 2)
			float UA = X2 * Y1 - Y2 * X1;
			float VA = X0 * Y2 - Y0 * X2;
			float WA = X1 * Y0 - Y1 * X0;
			// B: (A, B, C) = (0, 2, 3): W = Bx*Ay - By*Ax = X2*Y0 - Y2*X0 = -(X0*Y2 - Y0*X2) = -VA, bit for bit
			float UB = X3 * Y2 - Y3 * X2;
			float VB = X0 * Y3 - Y0 * X3;
			float WB = -VA;
			float detA, detB;
			const bool inA = inside(UA, VA, WA, detA, X0, Y0, X1, Y1, X2, Y2);
			const bool inB = inside(UB, VB, WB, detB, X0, Y0, X2, Y2, X3, Y3);
			// a ray is inside at most one of the two (both only on the shared edge or across a folded quad): one pass through
			// the distance arithmetic with A's or B's operands selected, and a second one for the rare lane inside both
			if (inA || inB)
				accept(inA ? UA : UB, inA ? VA : VB, inA ? WA : WB, inA ? detA : detB, P0z, inA ? P1z : P2z, inA ? P2z : P3z,
				       inA ? prims.x : prims.y);
			if (inA && inB) accept(UB, VB, WB, detB, P0z, P2z, P3z, prims.y);
		}
	}
}

template <int kStackLds, bool kQuads = false>
__device__ __forceinline__ void pt_leaf_step(const PtSceneDev &sc, const PtLaneRay &ray, float tmin, PtBest &best,
                                             uint32_t &cur, int &sp, uint32_t (*sStack)[PT_WAVE], int lane,
                                             uint32_t *mySpill, uint32_t spillStride, uint32_t &triTests,
                                             uint32_t curInst = 0u, uint32_t *bestInst = nullptr)
{
	// the next entry is needed whatever the triangles say: read it first
	--sp;
	uint32_t next = sp < kStackLds ? sStack[sp][lane]
	                               : (sp < kStackLds + PT_SPILL_STACK ? mySpill[(size_t)(sp - kStackLds) * spillStride]
	                                                                  : PT_EMPTY_CHILD);
	pt_leaf_test<kQuads>(sc, ray, tmin, best, cur, triTests, curInst, bestInst);
	cur = next;
}
