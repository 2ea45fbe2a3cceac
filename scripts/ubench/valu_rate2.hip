// Micro-benchmark 2: issue rate per VALU opcode (inline asm, 8 independent chains), 4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define DEFK(NAME, ASM)                                                                  \
	__global__ void __launch_bounds__(256) NAME(float *out, float a, float b)            \
	{                                                                                    \
		float x[8];                                                                      \
		for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;                     \
		float y = a, z = b;                                                              \
		asm volatile("v_cmp_lt_f32 vcc, %0, %1\nv_cmp_lt_f32 s[20:21], %0, %1" :: "v"(x[0]), "v"(y) : "vcc", "s20", "s21"); \
		for (int i = 0; i < N_ITER; ++i)                                                 \
		{                                                                                \
			asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)         \
			             : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
			             : "v"(y), "v"(z) : "vcc", "s20", "s21");                                      \
		}                                                                                \
		float s = 0;                                                                     \
		for (int i = 0; i < 8; ++i) s += x[i];                                           \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                  \
	}
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MINF(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define A_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MINU(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define A_CVTUB(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
#define A_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 8, 8\n"
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define A_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define A_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define A_CVTU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define A_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_MADU24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_FMAMIX(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9 op_sel_hi:[1,0,0]\n"
#define A_FMAMIXHI(i) "v_fma_mix_f32 %" #i ", %" #i ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_MAXE32(i) "v_max_f32_e32 %" #i ", %" #i ", %8\n"
#define A_MAXE64(i) "v_max_f32_e64 %" #i ", %" #i ", %8\n"
#define A_CVTSDWA(i) "v_cvt_f32_ubyte0_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
#define A_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
#define A_CNDPAIR(i) "v_cmp_lt_f32 vcc, %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_CNDPAIRS(i) "v_cmp_lt_f32 s[20:21], %" #i ", %9\nv_cndmask_b32 %" #i ", %" #i ", %8, s[20:21]\n"
DEFK(k_fma, A_FMA) DEFK(k_minf, A_MINF) DEFK(k_max3, A_MAX3) DEFK(k_minu, A_MINU) DEFK(k_and, A_AND) DEFK(k_addu, A_ADDU)
DEFK(k_lshladd, A_LSHLADD) DEFK(k_cvtub, A_CVTUB) DEFK(k_cmp, A_CMP) DEFK(k_cnd, A_CND) DEFK(k_bfe, A_BFE) DEFK(k_mov, A_MOV)
DEFK(k_mul, A_MUL) DEFK(k_sub, A_SUB) DEFK(k_rcp, A_RCP) DEFK(k_sqrt, A_SQRT) DEFK(k_cvtu, A_CVTU) DEFK(k_med3, A_MED3)
DEFK(k_fmamix, A_FMAMIX) DEFK(k_fmamixhi, A_FMAMIXHI) DEFK(k_maxe32, A_MAXE32) DEFK(k_maxe64, A_MAXE64) DEFK(k_cvtsdwa, A_CVTSDWA) DEFK(k_cnds, A_CNDS) DEFK(k_cndpair, A_CNDPAIR) DEFK(k_cndpairs, A_CNDPAIRS) DEFK(k_perm, A_PERM) DEFK(k_madu24, A_MADU24) DEFK(k_xor, A_XOR) DEFK(k_mullo, A_MULLO)
typedef void (*kfn)(float *, float, float);
static void run(const char *name, kfn f, float *d)
{
	const int wps = 4, blocks = 256 * wps;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(f, blocks, 256, 0, 0, d, 1.0001f, 0.5f);
	hipEventRecord(a);
	hipLaunchKernelGGL(f, blocks, 256, 0, 0, d, 1.0001f, 0.5f);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	double n = (double)N_ITER * 8 * wps;
	printf("%-14s %.3f ns per wave-instr per SIMD\n", name, ms * 1e6 / n);
}
int main()
{
	float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(x) run(#x, x, d);
	R(k_fma) R(k_mul) R(k_sub) R(k_minf) R(k_max3) R(k_med3) R(k_minu) R(k_and) R(k_xor) R(k_addu) R(k_lshladd) R(k_madu24) R(k_mullo)
	R(k_fmamix) R(k_fmamixhi) R(k_maxe32) R(k_maxe64) R(k_cvtsdwa) R(k_bfe) R(k_perm) R(k_cvtub) R(k_cvtu) R(k_cmp) R(k_cnd) R(k_cnds) R(k_cndpair) R(k_cndpairs) R(k_mov) R(k_rcp) R(k_sqrt)
	return 0;
}
