// How fast does HBM serve independent random gathers of 64-, 128- and 256-byte records (one
// record per lane per step, table far larger than every cache)?  Tells whether a traversal is
// limited by the number of requests or by the bytes they move.
//   hipcc --offload-arch=gfx950 -O3 -o gather_sizes gather_sizes.hip && ./gather_sizes
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { if((x) != hipSuccess) { fprintf(stderr, "HIP call failed: %s\n", #x); exit(1); } } while(0)
#include <cstdlib>

template <int QUADS>  // record = QUADS * 16 bytes
__global__ void gather(const float4 *table, unsigned long long n_records, unsigned n_per_lane, float *out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (gid + 1);
    float acc = 0.0f;
    for(unsigned k = 0; k < n_per_lane; k++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const unsigned long long r = (x * 0x2545F4914F6CDD1Dull) % n_records;
        const float4 *p = table + r * QUADS;
        float4 v[QUADS];
#pragma unroll
        for(int q = 0; q < QUADS; q++) v[q] = p[q];
#pragma unroll
        for(int q = 0; q < QUADS; q++) acc += v[q].x + v[q].w;
    }
    out[gid] = acc;
}

// The same 64-byte records, fetched by four lanes each (16 bytes per lane): one load
// instruction covers 16 whole records instead of a quarter of 64.
__global__ void gather_coop(const float4 *table, unsigned long long n_records, unsigned n_per_lane, float *out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (gid + 1);
    float acc = 0.0f;
    for(unsigned k = 0; k < n_per_lane; k++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const unsigned r = (unsigned)((x * 0x2545F4914F6CDD1Dull) % n_records);
        float4 v[4];
#pragma unroll
        for(int j = 0; j < 4; j++) {
            const unsigned ro = __shfl(r, j * 16 + (lane >> 2));
            v[j] = table[(unsigned long long)ro * 4 + (lane & 3u)];
        }
#pragma unroll
        for(int j = 0; j < 4; j++) acc += v[j].x + v[j].w;
    }
    out[gid] = acc;
}

void run_coop(const float4 *table, unsigned long long bytes, float *out) {
    const unsigned threads = 256, per_lane = 16, blocks = 256 * 8 * 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned long long n_records = bytes / 64;
    hipLaunchKernelGGL(gather_coop, dim3(blocks), dim3(threads), 0, 0, table, n_records, per_lane, out);
    CK(hipEventRecord(a, 0));
    for(int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(gather_coop, dim3(blocks), dim3(threads), 0, 0, table, n_records, per_lane, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = 5.0 * blocks * threads * per_lane;
    printf("record  64 B, four lanes per record: %6.2f G records/s, %6.2f TB/s\n", recs / ms / 1e6, recs * 64 / ms / 1e9);
}

template <int QUADS>
void run(const float4 *table, unsigned long long bytes, float *out, int waves_per_simd) {
    const unsigned threads = 256, per_lane = 16;
    const unsigned blocks = 256 * waves_per_simd * 8;  // a few rounds of the whole chip
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned long long n_records = bytes / (QUADS * 16);
    hipLaunchKernelGGL(gather<QUADS>, dim3(blocks), dim3(threads), 0, 0, table, n_records, per_lane, out);
    CK(hipEventRecord(a, 0));
    for(int rep = 0; rep < 5; rep++) hipLaunchKernelGGL(gather<QUADS>, dim3(blocks), dim3(threads), 0, 0, table, n_records, per_lane, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double recs = 5.0 * blocks * threads * per_lane;
    printf("record %3d B, one lane per record:   %6.2f G records/s, %6.2f TB/s\n", QUADS * 16, recs / ms / 1e6,
           recs * QUADS * 16 / ms / 1e9);
}

int main() {
    const unsigned long long bytes = 8ull << 30;
    float4 *table = nullptr;
    float *out = nullptr;
    if(hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, (size_t)256 * 64 * 256 * 4) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    CK(hipMemset(table, 0, bytes));
    CK(hipDeviceSynchronize());
    run<2>(table, bytes, out, 8);
    run<4>(table, bytes, out, 8);
    run<8>(table, bytes, out, 8);
    run<16>(table, bytes, out, 8);
    run_coop(table, bytes, out);
    if(hipDeviceSynchronize() != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        return 1;
    }
    CK(hipFree(table));
    CK(hipFree(out));
    return 0;
}
