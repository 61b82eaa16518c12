// Calibrates rocprofv3's FETCH_SIZE for the traversal kernels' access pattern: every lane
// reads one 64-byte record (four 16-byte loads) at a pseudo-random place in a table far
// larger than the L2 and the 256 MiB Infinity Cache, so (nearly) every record comes from HBM.
// Known bytes per launch = lanes * 64; compare with the counter.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void gather64(const float4 *table, unsigned long long n_records, unsigned n_per_lane, float *out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (gid + 1);
    float acc = 0.0f;
    for(unsigned k = 0; k < n_per_lane; k++) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;  // xorshift64*
        const unsigned long long r = (x * 0x2545F4914F6CDD1Dull) % n_records;
        const float4 *p = table + r * 4;
        const float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[gid] = acc;
}

int main() {
    const unsigned long long bytes = 8ull << 30, n_records = bytes / 64;
    float4 *table = nullptr;
    float *out = nullptr;
    const unsigned blocks = 8192, threads = 256, per_lane = 16;
    if(hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, (size_t)blocks * threads * 4) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(table, 0, bytes);
    hipDeviceSynchronize();
    for(int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(gather64, dim3(blocks), dim3(threads), 0, 0, table, n_records, per_lane, out);
    if(hipDeviceSynchronize() != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        return 1;
    }
    printf("known bytes per launch: %llu (%u lanes x %u records x 64 B)\n", (unsigned long long)blocks * threads * per_lane * 64,
           blocks * threads, per_lane);
    hipFree(table);
    hipFree(out);
    return 0;
}
