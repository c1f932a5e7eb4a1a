// Issue rate of single vector instructions on gfx950: cycles per wave64 instruction per SIMD, with
// 1, 2 and 4 waves resident per SIMD.  Every lane runs UNROLL independent chains of one opcode.
//   hipcc --offload-arch=gfx950 -O3 -o bin/valu_rate_probe tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int ITERS = 4096;

enum Op { FMA, PK_FMA, MAD_U64, MUL_LO, MUL_HI, CNDMASK, CMP, RCP, SQRT, SIN, MAX3, MUL_U24, PK_MUL, PK_ADD, CVT_UBYTE, MOV, XOR_, ADD_U32, MIXED, MIX_MOV, MIX_2TO1, MIX_AND64, MIX_BRANCH, MIX_WAIT, MIX_NOP, MIX_XOR, CNDMASK_S, ADD_F32, MUL_F32, AND_B32, LSHL, MIX_CMPX, SALU_ONLY, MIX_INT };

template <int OP> __global__ void __launch_bounds__(256) k_rate(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    unsigned long long q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, q4 = q0 + 4, q5 = q0 + 5, q6 = q0 + 6, q7 = q0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    const float c = seed * 0.5f;
    const unsigned m = 0xD2511F53u;
    for (int i = 0; i < ITERS; ++i) {
#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
        if (OP == FMA) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == PK_FMA) {
#define S(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##k) : "v"(p0));
            R8(S) R8(S)
#undef S
        } else if (OP == PK_MUL) {
#define S(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##k) : "v"(p0));
            R8(S) R8(S)
#undef S
        } else if (OP == PK_ADD) {
#define S(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##k) : "v"(p0));
            R8(S) R8(S)
#undef S
        } else if (OP == MAD_U64) {
#define S(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q##k) : "v"(u##k), "v"(m) : "vcc");
            R8(S) R8(S)
#undef S
        } else if (OP == MUL_LO) {
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == MUL_HI) {
#define S(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == MUL_U24) {
#define S(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == CNDMASK) {
#define S(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == CMP) {
#define S(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a##k), "v"(c) : "vcc");
            R8(S) R8(S)
#undef S
        } else if (OP == RCP) {
#define S(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##k));
            R8(S) R8(S)
#undef S
        } else if (OP == SQRT) {
#define S(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##k));
            R8(S) R8(S)
#undef S
        } else if (OP == SIN) {
#define S(k) asm volatile("v_sin_f32 %0, %0" : "+v"(a##k));
            R8(S) R8(S)
#undef S
        } else if (OP == MAX3) {
#define S(k) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == CVT_UBYTE) {
#define S(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a##k));
            R8(S) R8(S)
#undef S
        } else if (OP == MOV) {
#define S(k) asm volatile("v_mov_b32 %0, %1" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == XOR_) {
#define S(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == ADD_U32) {
#define S(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_MOV) { // fma + an independent scalar move
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_mov_b32 s2" #k ", 7" : "+v"(a##k) : "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_2TO1) { // two fma per scalar instruction
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n s_mov_b32 s2" #k ", 7" : "+v"(a##k) : "v"(c) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            R8(S)
#undef S
        } else if (OP == MIX_AND64) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_and_b64 s[20:21], s[22:23], exec" : "+v"(a##k) : "v"(c) : "s20", "s21", "scc");
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_BRANCH) { // fma + a branch that is never taken
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_cbranch_execz 1f\n1:" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_WAIT) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_NOP) {
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_nop 0" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_XOR) { // fma + a cheap integer vector instruction
#define S(k) asm volatile("v_fma_f32 %0, %0, %2, %2\n v_xor_b32 %1, %1, %3" : "+v"(a##k), "+v"(u##k) : "v"(c), "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_INT) { // two cheap integer vector instructions
#define S(k) asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == MIX_CMPX) { // compare into an SGPR pair + select on it
#define S(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(c) : "s20", "s21");
            R8(S) R8(S)
#undef S
        } else if (OP == SALU_ONLY) {
#define S(k) asm volatile("s_mov_b32 s2" #k ", 7" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            R8(S) R8(S)
#undef S
        } else if (OP == CNDMASK_S) {
#define S(k) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == ADD_F32) {
#define S(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == MUL_F32) {
#define S(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##k) : "v"(c));
            R8(S) R8(S)
#undef S
        } else if (OP == AND_B32) {
#define S(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u##k) : "v"(m));
            R8(S) R8(S)
#undef S
        } else if (OP == LSHL) {
#define S(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u##k));
            R8(S) R8(S)
#undef S
        } else if (OP == MIXED) { // 8 fma + 8 scalar instructions interleaved: does the scalar unit take vector issue slots?
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_add_u32 s20, s20, 1" : "+v"(a##k) : "v"(c) : "s20", "scc");
            R8(S) R8(S)
#undef S
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    unsigned long long q = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    unsigned u = u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7;
    if (r == 12345.678f && q == 77 && u == 99) out[0] = r;
}

template <int OP> void run(const char *name, float *out, int cus, double ghz) {
    printf("%-14s", name);
    for (int waves = 1; waves <= 8; waves *= 2) {
        // `waves` blocks of 256 threads per CU: one wave of each block per SIMD
        dim3 grid(cus * waves);
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        k_rate<OP><<<grid, 256>>>(out, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_rate<OP><<<grid, 256>>>(out, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = double(waves) * ITERS * 16.0;
        printf("  %dw: %5.2f", waves, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
    printf("   cycles per wave-instruction per SIMD\n");
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, %.2f GHz nominal\n", p.gcnArchName, cus, ghz);
    float *out;
    CHECK(hipMalloc(&out, 4));
    run<FMA>("v_fma_f32", out, cus, ghz);
    run<PK_FMA>("v_pk_fma_f32", out, cus, ghz);
    run<PK_MUL>("v_pk_mul_f32", out, cus, ghz);
    run<PK_ADD>("v_pk_add_f32", out, cus, ghz);
    run<MAX3>("v_max3_f32", out, cus, ghz);
    run<CNDMASK>("v_cndmask_b32", out, cus, ghz);
    run<CMP>("v_cmp_lt_f32", out, cus, ghz);
    run<MOV>("v_mov_b32", out, cus, ghz);
    run<XOR_>("v_xor_b32", out, cus, ghz);
    run<ADD_U32>("v_add_u32", out, cus, ghz);
    run<MAD_U64>("v_mad_u64_u32", out, cus, ghz);
    run<MUL_LO>("v_mul_lo_u32", out, cus, ghz);
    run<MUL_HI>("v_mul_hi_u32", out, cus, ghz);
    run<MUL_U24>("v_mul_u32_u24", out, cus, ghz);
    run<RCP>("v_rcp_f32", out, cus, ghz);
    run<SQRT>("v_sqrt_f32", out, cus, ghz);
    run<SIN>("v_sin_f32", out, cus, ghz);
    run<CVT_UBYTE>("v_cvt_f32_ub1", out, cus, ghz);
    run<CNDMASK_S>("v_cndmask sgpr", out, cus, ghz);
    run<ADD_F32>("v_add_f32", out, cus, ghz);
    run<MUL_F32>("v_mul_f32", out, cus, ghz);
    run<AND_B32>("v_and_b32", out, cus, ghz);
    run<LSHL>("v_lshlrev_b32", out, cus, ghz);
    printf("pairs (cycles per pair of instructions):\n");
    run<MIXED>("fma+s_add dep", out, cus, ghz);
    run<MIX_MOV>("fma+s_mov", out, cus, ghz);
    run<MIX_2TO1>("2fma+s_mov", out, cus, ghz);
    run<MIX_AND64>("fma+s_and_b64", out, cus, ghz);
    run<MIX_BRANCH>("fma+s_cbranch", out, cus, ghz);
    run<MIX_WAIT>("fma+s_waitcnt", out, cus, ghz);
    run<MIX_NOP>("fma+s_nop", out, cus, ghz);
    run<MIX_XOR>("fma+v_xor", out, cus, ghz);
    run<MIX_INT>("v_add+v_xor", out, cus, ghz);
    run<MIX_CMPX>("v_cmp+v_cndmask", out, cus, ghz);
    run<SALU_ONLY>("s_mov only", out, cus, ghz);
    CHECK(hipFree(out));
    return 0;
}
