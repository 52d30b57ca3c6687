// LDS micro-benchmark for gfx950: LDS-array cycles per wave-instruction of ds_add_f64 / ds_read_b64 under the address
// patterns K1-3D produces.  One workgroup per CU; every wave issues N back-to-back operations; the time per operation
// and CU (in cycles at the measured clock) is printed.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_lds.hip -o gpurun_out/ubench_lds && gpurun_out/ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NDBL = 16384;   // 128 KB of doubles

template <int MODE>
__global__ void __launch_bounds__(1024) k(const int *__restrict__ idx, double *out, int iters, long long *cyc, int every) {
    __shared__ double s[NDBL];
    for (int t = threadIdx.x; t < NDBL; t += blockDim.x) s[t] = MODE == 0 ? 0.0 : out[2048 + (t & 7)];   // run-time data: nothing to fold
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int a = idx[lane] + wave * 600;      // per-wave base: waves work on different regions (like different cells)
    const bool active = (lane % every) == 0;   // exec-masked atomics: what a sparse wave instruction costs
    double acc = 0.0, acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        asm volatile("" ::: "memory");        // the reads of an iteration are not hoisted out of the loop
        const int ai = a + (it & 1) * 720;
#pragma unroll
        for (int u = 0; u < 27; u++) {
            int o = MODE == 0 ? ai + (u / 9) * 240 + ((u / 3) % 3) * 24 + (u % 3) : ai + u * 300;   // reads: too far apart for ds_read2_b64
            if (MODE == 0) { if (active) atomicAdd(&s[o], 1.0); }
            else acc9[u % 9] += s[o];        // nine independent chains: the reads stay in flight
        }
    }
    for (int q = 0; q < 9; q++) acc += acc9[q];
    __syncthreads();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (acc == 1.2345) out[0] = acc;
    if (MODE == 0 && threadIdx.x == 0) out[1 + blockIdx.x] = s[idx[0]];
}

int main() {
    int dev = 0; hipSetDevice(dev);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, dev);
    printf("# %s CUs=%d clock=%d kHz wallclock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate, wc_khz);
    struct Pat { const char *name; std::vector<int> v; };
    std::vector<Pat> pats;
    auto mk = [&](const char *n, auto f) { Pat p; p.name = n; p.v.resize(64); for (int l = 0; l < 64; l++) p.v[l] = f(l); pats.push_back(p); };
    mk("consecutive (lane)", [](int l) { return l; });
    mk("4 columns of 16 z, y stride 24", [](int l) { return (l >> 4) * 24 + (l & 15); });
    mk("4 columns of 16 z, y stride 21", [](int l) { return (l >> 4) * 21 + (l & 15); });
    mk("4 columns of 16 z, y stride 48", [](int l) { return (l >> 4) * 48 + (l & 15); });
    mk("all lanes one address", [](int l) { return 0; });
    mk("pairs share an address (l/2)", [](int l) { return l / 2; });
    mk("8 lanes per address (l/8)", [](int l) { return l / 8; });
    mk("stride 2 doubles", [](int l) { return 2 * l; });
    mk("stride 32 doubles (same bank)", [](int l) { return 32 * (l & 15) + (l >> 4); });
    mk("16-lane group: 2-way bank conflict", [](int l) { return (l & 7) + 32 * ((l >> 3) & 1) + 64 * (l >> 4); });
    mk("random cells, 2 columns mixed", [](int l) { return ((l * 7) % 16) + 24 * ((l * 5) % 3); });
    int *d_idx; double *d_out; long long *d_cyc;
    hipMalloc(&d_idx, 64 * 4); hipMalloc(&d_out, 8 * 4096); hipMalloc(&d_cyc, 8);
    const int iters = 200, nblk = pr.multiProcessorCount;
    for (int mode = 0; mode < 2; mode++)
      for (int every : {1, 2, 4, 16, 64}) {
        if (mode == 1 && every > 1) continue;
        for (int threads : {256, 768}) {
            if (every > 1 && threads == 256) continue;
            printf("== %s, %d threads per CU, 1 lane in %d active (27 ops per iteration, %d iterations)\n", mode == 0 ? "ds_add_f64" : "ds_read_b64", threads, every, iters);
            for (auto &p : pats) {
                hipMemcpy(d_idx, p.v.data(), 64 * 4, hipMemcpyHostToDevice);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(threads), 0, 0, d_idx, d_out, iters, d_cyc, every);
                    else hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(threads), 0, 0, d_idx, d_out, iters, d_cyc, every);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long wc; hipMemcpy(&wc, d_cyc, 8, hipMemcpyDeviceToHost);
                double ops = (double)iters * 27 * (threads / 64);          // wave-instructions per CU
                double ns_per_op = ms * 1e6 / ops;
                printf("   %-40s %8.3f ms  %6.2f ns per wave-instruction per CU = %5.1f cycles at 2.4 GHz (in-kernel %.1f wallclock ticks/op)\n",
                       p.name, ms, ns_per_op, ns_per_op * 2.4, (double)wc / ops);
            }
        }
      }
    return 0;
}
