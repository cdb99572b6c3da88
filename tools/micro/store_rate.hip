// Microbenchmark: how fast can ONE workgroup (512 threads) on a CU store a 256 x 256 fp32 tile (256 KB), as a function of how
// many CUs do it at once and of the store shape.  Prints shader cycles (s_memtime) per workgroup: issue only / stores drained.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// shape 0: a wave-instruction = 4 rows x 256 contiguous bytes (16 lanes x 16 B);  shape 1: 4 rows x two 128-byte segments 512 B apart
// (the eight-phase kernel's wave geometry);  shape 2: one row, 1024 contiguous bytes
template <int SHAPE>
__global__ __launch_bounds__(512) void store_tile(float* out, int ld, int tiles_n, unsigned long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    float* base = out + (size_t)tm * 256 * ld + tn * 256;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const float4 v = make_float4((float)tid, 1.f, 2.f, 3.f);
    const int wr = wave >> 2, wc = wave & 3;
    if (SHAPE == 2) {
        // wave w stores rows w * 32 .. + 31, a whole 1 KiB row per instruction
        for (int r = 0; r < 32; ++r) *reinterpret_cast<float4*>(base + (size_t)(wave * 32 + r) * ld + lane * 4) = v;
    } else {
        for (int ps = 0; ps < 2; ++ps)
            for (int it = 0; it < 16; ++it) {
                const int row = ps * 128 + wr * 64 + it * 4 + (lane >> 4);
                const int c4 = (lane & 15) * 4;
                const int col = SHAPE == 0 ? wc * 64 + c4 : wc * 32 + (c4 & 31) + (c4 >> 5) * 128;
                *reinterpret_cast<float4*>(base + (size_t)row * ld + col) = v;
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        stamps[blockIdx.x * 2] = t1 - t0;
        stamps[blockIdx.x * 2 + 1] = t2 - t0;
    }
}

int main() {
    const int M = 5120, N = 30208;
    float* out;
    unsigned long long* st;
    hipMalloc(&out, (size_t)M * N * 4);
    hipMalloc(&st, 4096 * 16);
    for (int shape = 0; shape < 3; ++shape)
        for (int tiles : {4, 80, 256, 1024}) {
            const int tiles_n = (tiles + 19) / 20;  // at most 20 row tiles (M = 5120), at most 52 column tiles (13 312 <= N columns)
            for (int rep = 0; rep < 3; ++rep) {
                if (shape == 0) hipLaunchKernelGGL(store_tile<0>, dim3(tiles), dim3(512), 0, 0, out, N, tiles_n, st);
                if (shape == 1) hipLaunchKernelGGL(store_tile<1>, dim3(tiles), dim3(512), 0, 0, out, N, tiles_n, st);
                if (shape == 2) hipLaunchKernelGGL(store_tile<2>, dim3(tiles), dim3(512), 0, 0, out, N, tiles_n, st);
            }
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> h(tiles * 2);
            hipMemcpy(h.data(), st, tiles * 16, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> a, b;
            for (int i = 0; i < tiles; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
            std::sort(a.begin(), a.end());
            std::sort(b.begin(), b.end());
            printf("shape %d, %4d workgroups (256 KB each): issue median %6llu  drained median %6llu  p90 %6llu cycles -> %.1f B/clk per workgroup\n", shape, tiles,
                   a[tiles / 2], b[tiles / 2], b[tiles * 9 / 10], 262144.0 / (double)b[tiles / 2]);
        }
    return 0;
}
