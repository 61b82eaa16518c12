// Issue cost of the vector instructions the traversal's wide-node step is made of, measured: one
// wave per SIMD (and four, to see what interleaving hides), each running a long unrolled stream of
// ONE instruction over eight independent register chains; cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s\n", #x); exit(1); } } while(0)

typedef float pair_t __attribute__((ext_vector_type(2)));
constexpr int REPS = 16384, CHAINS = 8;   // ~0.3 ms per launch: every block is resident long before the first one ends

#define BODY(NAME, DECL, STMT)                                                                     \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *cycles, float *sink, float seed) { \
        DECL;                                                                                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                \
        _Pragma("unroll 16") for(int r = 0; r < REPS; r++) {                                       \
            _Pragma("unroll") for(int c = 0; c < CHAINS; c++) { STMT; }                            \
        }                                                                                          \
        const unsigned long long t1 = __builtin_readcyclecounter();                                \
        float acc = 0;                                                                             \
        for(int c = 0; c < CHAINS; c++) acc += out_of(c);                                          \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                         \
        if(threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;                                         \
    }

#define DECL_F float a[CHAINS]; for(int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x; float k1 = seed * 1.0001f, k2 = seed * 0.5f; auto out_of = [&](int c) { return a[c]; }
#define DECL_P pair_t a[CHAINS]; for(int c = 0; c < CHAINS; c++) a[c] = pair_t{seed + c + threadIdx.x, seed - c}; pair_t k1 = {seed * 1.0001f, seed}, k2 = {seed * 0.5f, seed}; auto out_of = [&](int c) { return a[c].x + a[c].y; }
#define DECL_U unsigned u[CHAINS]; float a[CHAINS]; for(int c = 0; c < CHAINS; c++) { u[c] = __float_as_uint(seed) + c * 77u + threadIdx.x; a[c] = 0; } float k1 = seed; auto out_of = [&](int c) { return a[c] + __uint_as_float(u[c]); }

BODY(k_fma, DECL_F, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(k1), "v"(k2)))
BODY(k_mul, DECL_F, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k1)))
BODY(k_max3, DECL_F, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(k1), "v"(k2)))
BODY(k_min, DECL_F, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k1)))
BODY(k_rcp, DECL_F, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[c])))
BODY(k_pkfma, DECL_P, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(k1), "v"(k2)))
BODY(k_pkmul, DECL_P, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k1)))
BODY(k_pkadd, DECL_P, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(k1)))
BODY(k_cvtub, DECL_U, asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[c]) : "v"(u[c])))
BODY(k_perm, DECL_U, asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(u[c]), "v"(k1), "v"(k1)))
BODY(k_bfe, DECL_U, asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(a[c]) : "v"(u[c])))
BODY(k_andor, DECL_U, asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(a[c]) : "v"(u[c]), "v"(k1), "v"(k1)))
BODY(k_lshlor, DECL_U, asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(a[c]) : "v"(u[c]), "v"(k1)))
BODY(k_cvtu32, DECL_U, asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[c]) : "v"(u[c])))
BODY(k_cndmask, DECL_U, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(k1)))
BODY(k_minu, DECL_U, asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[c]) : "v"(k1)))
BODY(k_cmp, DECL_U, asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[c]), "v"(k1) : "vcc"))
BODY(k_madu64, DECL_U, { unsigned long long w; asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, 0" : "=v"(w) : "v"(u[c]) : "vcc"); a[c] += (float)(unsigned)w; })

template <class K>
void run(const char *name, K kernel, int flops_per_inst) {
    unsigned long long *cyc;
    float *sink;
    CK(hipMalloc(&cyc, 4096 * 8));
    CK(hipMalloc(&sink, 4096 * 256 * 4));
    for(int waves_per_simd : {1, 2, 4}) {
        // 256 CUs; a block of 256 threads = 4 waves = one wave per SIMD of a CU
        const int blocks = 256 * waves_per_simd;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, cyc, sink, 1.5f);
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, cyc, sink, 1.5f);
        CK(hipDeviceSynchronize());
        static unsigned long long h[1024];
        CK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
        double sum = 0;
        for(int b = 0; b < blocks; b++) sum += (double)h[b];
        const double per = sum / blocks / ((double)REPS * CHAINS);
        printf("%-14s %d wave(s)/SIMD: %6.2f ticks per instruction per wave = %5.2f ticks of the SIMD per instruction\n", name, waves_per_simd, per,
               per / waves_per_simd);
    }
    CK(hipFree(cyc));
    CK(hipFree(sink));
}

int main() {
    run("v_fma_f32", k_fma, 2);
    run("v_mul_f32", k_mul, 1);
    run("v_max3_f32", k_max3, 1);
    run("v_min_f32", k_min, 1);
    run("v_rcp_f32", k_rcp, 1);
    run("v_pk_fma_f32", k_pkfma, 4);
    run("v_pk_mul_f32", k_pkmul, 2);
    run("v_pk_add_f32", k_pkadd, 2);
    run("v_cvt_ubyte", k_cvtub, 1);
    run("v_perm_b32", k_perm, 1);
    run("v_bfe_u32", k_bfe, 1);
    run("v_and_or_b32", k_andor, 1);
    run("v_lshl_or_b32", k_lshlor, 1);
    run("v_cvt_u32", k_cvtu32, 1);
    run("v_cndmask", k_cndmask, 1);
    run("v_min_u32", k_minu, 1);
    run("v_cmp_lt_u32", k_cmp, 1);
    run("v_mad_u64_u32", k_madu64, 1);
    return 0;
}
