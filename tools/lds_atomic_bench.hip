// Microbenchmark (design aid, not product): throughput of ds_add_f64 / ds_read_b64 under different address
// patterns on gfx950.  Prints entries per cycle per CU (assuming 2.4 GHz is NOT needed: we report ns and Gop/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE, bool ATOMIC>
__global__ __launch_bounds__(512) void k(const int *__restrict__ idx, double *out, int iters) {
    __shared__ double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) buf[i] = 1.0;
    __syncthreads();
    int lane = threadIdx.x & 63;
    double acc = 0;
    const int *p = idx + (size_t)blockIdx.x * 512 * 16 + threadIdx.x;
    int a[16];
    for (int j = 0; j < 16; j++) {
        int r = p[j * 512] & 4095;
        if (MODE == 0) a[j] = (lane + j * 64) & 4095;          // distinct consecutive
        else if (MODE == 1) a[j] = r;                           // random in 4096
        else if (MODE == 2) a[j] = (j * 7) & 4095;              // wave-uniform address
        else if (MODE == 3) a[j] = ((lane >> 5) + j * 2) & 4095; // 2 distinct (32 each)
        else if (MODE == 4) a[j] = ((lane >> 3) + j * 8) & 4095; // 8 distinct (8 each)
        else a[j] = (r & 63) + 100;                             // random among 64 hot addresses
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (ATOMIC) __hip_atomic_fetch_add(&buf[a[j]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += buf[a[j]];
        }
        if (!ATOMIC) { for (int j = 0; j < 16; j++) a[j] = (a[j] + (int)(acc == -1.0)) & 4095; }
    }
    __syncthreads();
    if (ATOMIC) acc = buf[threadIdx.x];
    if (acc == 12345.678) out[0] = acc;
}

template <int MODE, bool ATOMIC>
void run(const char *name, const int *d_idx, double *d_out) {
    const int blocks = 512, iters = 200;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, ATOMIC>), dim3(blocks), dim3(512), 0, 0, d_idx, d_out, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, ATOMIC>), dim3(blocks), dim3(512), 0, 0, d_idx, d_out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)blocks * 512 * 16 * iters;
    printf("%-28s %s  %8.3f ms  %8.1f Gop/s  = %.2f lane-ops/clk/CU @2.4GHz\n", name, ATOMIC ? "ds_add_f64 " : "ds_read_b64", ms,
           ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    std::vector<int> h(512 * 512 * 16);
    srand(1); for (auto &x : h) x = rand();
    int *d_idx; double *d_out;
    CHECK(hipMalloc(&d_idx, h.size() * 4)); CHECK(hipMalloc(&d_out, 8));
    CHECK(hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0, true>("distinct consecutive", d_idx, d_out);  run<0, false>("distinct consecutive", d_idx, d_out);
    run<1, true>("random in 4096", d_idx, d_out);        run<1, false>("random in 4096", d_idx, d_out);
    run<2, true>("wave-uniform", d_idx, d_out);          run<2, false>("wave-uniform", d_idx, d_out);
    run<3, true>("2 distinct x32", d_idx, d_out);        run<3, false>("2 distinct x32", d_idx, d_out);
    run<4, true>("8 distinct x8", d_idx, d_out);         run<4, false>("8 distinct x8", d_idx, d_out);
    run<5, true>("random among 64 hot", d_idx, d_out);   run<5, false>("random among 64 hot", d_idx, d_out);
    return 0;
}
