// microbenchmark: cycles per ds_add_f64 / ds_read_b64 wave instruction on gfx950 for the address
// patterns of the tiled deposit.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int N = 8192;   // doubles of LDS (64 KB): 2 blocks of 8 waves per CU, like K1
constexpr int REP = 256;

template <int MODE>
__global__ void __launch_bounds__(512) k(const int *idx, double *out, long long *cyc, int nactive) {
    __shared__ double s[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) s[i] = 0.0;
    __syncthreads();
    int lane = threadIdx.x & 63;
    int a = idx[threadIdx.x];
    double v = 1.0 + lane;
    bool act = nactive < 0 ? ((lane & 1) == 0) : lane < nactive;   // nactive < 0: even lanes only
    long long t0 = __builtin_amdgcn_s_memtime();
    double acc = 0.0;
    // immediate offsets only: no address arithmetic between the LDS instructions
    double *base = s + a;
    for (int r = 0; r < REP / 32; r++) {
        if (MODE == 0) {
            if (act) {
#pragma unroll
                for (int c = 0; c < 32; c++) atomicAdd(base + c * 64, v);
            }
        } else if (MODE == 3) {   // ds_read_b128 at an address that is 8- but not 16-byte aligned
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 t[16];
            unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) double *)(s + 2 * a + 1);
#pragma unroll
            for (int c = 0; c < 16; c++)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t[c]) : "v"(addr), "n"(c * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int c = 0; c < 16; c++) acc += t[c].x + t[c].y;
        } else if (MODE == 2) {   // 16 bytes per lane: ds_read_b128
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 t[16];
            const d2 *b2 = (const d2 *)(s + 2 * a);
#pragma unroll
            for (int c = 0; c < 16; c++) t[c] = b2[c * 64];
#pragma unroll
            for (int c = 0; c < 16; c++) acc += t[c].x + t[c].y;
        } else {
            double t[32];
#pragma unroll
            for (int c = 0; c < 32; c++) t[c] = base[c * 64];
#pragma unroll
            for (int c = 0; c < 32; c++) acc += t[c];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s[threadIdx.x];
}

int main() {
    const int T = 512, B = 512;
    int *h = (int *)malloc(T * sizeof(int)), *d;
    double *o; long long *c, hc[B];
    hipMalloc(&d, T * sizeof(int)); hipMalloc(&o, sizeof(double) * T * B); hipMalloc(&c, sizeof(long long) * B);
    const char *names[] = {"consecutive (lane -> lane)", "consecutive, wave offset 64*w", "random in 16x64 tile", "stride 2", "all lanes same address", "16-lane groups same row, rows 64 apart",
        "consecutive, every 4th lane drifted +1 column", "quarters = even / odd columns, every 4th lane drifted +1",
        "quarters = even / odd columns, no drift", "consecutive, every 4th lane drifted +1 ROW (64 doubles)"};
    for (int pat = 0; pat < 10; pat++) {
        srand(1);
        for (int t = 0; t < T; t++) {
            int lane = t & 63, w = t >> 6;
            if (pat == 0) h[t] = lane;
            if (pat == 1) h[t] = lane + 64 * w;
            if (pat == 2) h[t] = rand() % 1024;
            if (pat == 3) h[t] = 2 * lane + 128 * w;
            if (pat == 4) h[t] = 5;
            if (pat == 5) h[t] = (lane & 15) + 64 * (lane >> 4) + 256 * w;
            if (pat == 6) h[t] = lane + ((lane & 3) == 1 ? 1 : 0);
            if (pat == 7 || pat == 8) {   // lanes 0-15: even columns of row 0, 16-31: odd columns, 32-63: row 1
                int q = lane >> 4, i = lane & 15;
                int col = 2 * i + (q & 1), row = q >> 1;
                if (pat == 7 && (lane & 3) == 1) col += 1;
                h[t] = row * 64 + col;
            }
            if (pat == 9) h[t] = lane + ((lane & 3) == 1 ? 64 : 0);
        }
        hipMemcpy(d, h, T * sizeof(int), hipMemcpyHostToDevice);
        for (int mode = 0; mode < 4; mode++)
            for (int na : {64, 32, -32, 8, 2}) {
                if (mode >= 1 && na != 64) continue;
                if (mode >= 2 && pat != 0 && pat != 1) continue;   // b128: 2 * a must stay inside the array
                for (int it = 0; it < 2; it++) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(B), dim3(T), 0, 0, d, o, c, na);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(T), 0, 0, d, o, c, na);
                    else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(T), 0, 0, d, o, c, na);
                    else hipLaunchKernelGGL(k<3>, dim3(B), dim3(T), 0, 0, d, o, c, na);
                    hipDeviceSynchronize();
                }
                hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
                double m = 0; for (int b = 0; b < B; b++) m += hc[b]; m /= B;
                // 8 waves per block, 2 blocks per CU co-resident: cycles per wave instruction seen by the CU
                printf("%-42s %s active=%2d : %8.1f cyc/block -> %6.2f cyc per wave-instr per CU (16 waves = 2 blocks/CU)\n",
                       names[pat], mode == 3 ? "ds_read_b128 +8B misaligned (per 2 doubles /2)" : mode == 2 ? "ds_read_b128 (16 instr = 32 doubles)" : mode ? "ds_read_b64" : "ds_add_f64 ", na, m, m / (REP * 16.0));
            }
    }
    return 0;
}
