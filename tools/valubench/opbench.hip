// opbench.hip -- issue cost of single gfx950 vector instructions as the sparse scan uses them (four independent chains, 8 waves per SIMD):
// ns of SIMD time per wave-instruction, relative to v_add_u32.  Build: hipcc -O3 --offload-arch=gfx950 opbench.hip -o opbench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define OP4(ins) asm volatile(ins(0) ins(1) ins(2) ins(3) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(c), "s"(sm) : "vcc");
#define I_ADD(k)   "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_SHL64(k) "v_lshlrev_b64 %" "4" ", %8, %" "4" "\n"
#define I_BFE(k)   "v_bfe_u32 %" #k ", %" #k ", %8, 1\n"
#define I_PKMAD(k) "v_pk_mad_u16 %" #k ", %" #k ", %8, %8\n"
#define I_PKSUB(k) "v_pk_sub_u16 %" #k ", %" #k ", %8\n"
#define I_PKMIN(k) "v_pk_min_u16 %" #k ", %" #k ", %8\n"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %9, %" #k "\n"
#define I_SDWA(k)  "v_add_u32_sdwa %" #k ", %8, %" #k " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_PERM(k)  "v_perm_b32 %" #k ", %" #k ", %8, %8\n"
#define I_MAD24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %8\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define I_CMP(k)   "v_cmp_ne_u32 vcc, %" #k ", %8\n"
#define I_CMPS(k)  "v_cmp_gt_i32_e64 s[20:21], 0, %" #k "\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 16, %8\n"
#define I_ANDLIT(k) "v_and_b32 %" #k ", 0xfff8fff8, %" #k "\n"
#define I_NOP(k)   "s_nop 0\n"
template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters) {
    uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, c = threadIdx.x & 7;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 w0 = {threadIdx.x, 1u}, w1 = w0, w2 = w0, w3 = w0;
    uint32_t sm = 0x55555555u + blockIdx.x;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP16(OP4(I_ADD)) }
        else if (MODE == 1) {
            REP16(asm volatile("v_lshlrev_b64 %0, %4, %0\n v_lshlrev_b64 %1, %4, %1\n v_lshlrev_b64 %2, %4, %2\n v_lshlrev_b64 %3, %4, %3\n" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(c));)
        }
        else if (MODE == 2) { REP16(OP4(I_BFE)) }
        else if (MODE == 3) { REP16(OP4(I_PKMAD)) }
        else if (MODE == 4) { REP16(OP4(I_PKSUB)) }
        else if (MODE == 5) { REP16(OP4(I_PKMIN)) }
        else if (MODE == 6) { REP16(OP4(I_MBCNT)) }
        else if (MODE == 7) { REP16(OP4(I_SDWA)) }
        else if (MODE == 8) { REP16(OP4(I_PERM)) }
        else if (MODE == 9) { REP16(OP4(I_MAD24)) }
        else if (MODE == 10) { REP16(OP4(I_LSHLADD)) }
        else if (MODE == 11) { REP16(OP4(I_CMP)) }
        else if (MODE == 12) { REP16(asm volatile("v_cmp_gt_i32_e64 s[20:21], 0, %0\n v_cmp_gt_i32_e64 s[20:21], 0, %1\n v_cmp_gt_i32_e64 s[20:21], 0, %2\n v_cmp_gt_i32_e64 s[20:21], 0, %3\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "s20", "s21");) }
        else if (MODE == 13) { REP16(OP4(I_LSHLOR)) }
        else if (MODE == 14) { REP16(OP4(I_ANDLIT)) }
        else if (MODE == 15) { REP16(OP4(I_NOP)) }
        else if (MODE == 16) {  // the packed chain with its dependences: pk_sub -> pk_min -> and -> pk_mad -> sdwa add (does the wait state cost a slot?)
            REP16(asm volatile("v_pk_sub_u16 %0, %0, %4\n v_pk_sub_u16 %1, %1, %4\n v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(c));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + w0.x + w1.y + w2.x + w3.y;
}
int main() {
    uint32_t *d;
    (void)hipMalloc(&d, 1024 * 1024 * 4 * 4);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2000;
    const char *names[] = {"v_add_u32", "v_lshlrev_b64", "v_bfe_u32", "v_pk_mad_u16", "v_pk_sub_u16", "v_pk_min_u16", "v_mbcnt_lo (sgpr mask)", "v_add_u32_sdwa", "v_perm_b32",
                           "v_mad_u32_u24", "v_lshl_add_u32", "v_cmp_ne_u32 (vcc)", "v_cmp_gt_i32_e64 (sgpr)", "v_lshl_or_b32", "v_and_b32 literal", "s_nop 0", "pk_sub->pk_min dependent pairs"};
    double base = 0;
    for (int mode = 0; mode < 17; mode++) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        void (*kern)(uint32_t *, int) = nullptr;
        switch (mode) {
#define C(M) case M: kern = k<M>; break;
            C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
#undef C
        }
        hipLaunchKernelGGL(kern, dim3(cus * 2), dim3(1024), 0, 0, d, 10);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus * 2), dim3(1024), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / iters / 8.0 / 64.0;   // per wave-instruction per SIMD (8 waves per SIMD, 64 instructions per iteration)
        if (mode == 0) base = ns;
        printf("%-34s %6.3f ns per wave-instruction per SIMD = %4.2f x v_add_u32\n", names[mode], ns, ns / base);
    }
    return 0;
}
