// microbenchmark: do FP64 VALU work and ds_add_f64 / ds_read_b64 traffic of DIFFERENT waves of a SIMD
// overlap on gfx950?  Each wave runs REP iterations of {NF independent v_fma_f64, NL LDS ops}; the
// launch puts 4 waves on every SIMD (512-thread blocks, 64 KB LDS -> 2 blocks per CU) like K1.
// If the time of the mixed loop is max(t_valu, t_lds) the pipes overlap; if it is the sum they do not.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics valu_lds_overlap.hip -o valu_lds_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int N = 8192;
constexpr int REP = 256;

template <int NF, int NL, int KIND>   // KIND 0: ds_add_f64, 1: ds_read_b64
__global__ void __launch_bounds__(512) k(double *out, long long *cyc, double seed, long long *rt) {
    __shared__ double s[N];
    for (int i = threadIdx.x; i < N; i += blockDim.x) s[i] = 0.0;
    __syncthreads();
    double *base = s + threadIdx.x;                 // conflict free: lane -> consecutive doubles
    double a[8];
#pragma unroll
    for (int c = 0; c < 8; c++) a[c] = seed + c + threadIdx.x;
    double m = seed * 1.0000001, acc = 0.0;
    long long r0 = __builtin_amdgcn_s_memrealtime();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; r++) {
        constexpr int G = NL > 0 ? NL : 1;          // interleave: NF/G FMAs after every LDS op
#pragma unroll
        for (int l = 0; l < G; l++) {
            if (NL > 0) {
                if (KIND == 0) atomicAdd(base + (l & 15) * 512, a[l & 7]);
                else acc += base[(l & 15) * 512];
            }
#pragma unroll
            for (int f = 0; f < NF / G; f++) a[f & 7] = __builtin_fma(a[f & 7], m, 1.0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
    double t = acc;
#pragma unroll
    for (int c = 0; c < 8; c++) t += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t + s[threadIdx.x];
}

template <int NF, int NL, int KIND>
static double run(double *o, long long *c, long long *rt) {
    // event-timed whole launches: B = 256 (1 block per CU) and B = 512 (2 per CU, 4 waves per SIMD)
    double res[2];
    int bi = 0;
    for (int B : {256, 512}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<NF, NL, KIND>), dim3(B), dim3(512), 0, 0, o, c, 1.0, rt);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int it = 0; it < 4; it++) hipLaunchKernelGGL((k<NF, NL, KIND>), dim3(B), dim3(512), 0, 0, o, c, 1.0, rt);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        res[bi++] = ms * 1e6 / 4 / REP;              // ns per iteration of the whole grid
    }
    printf("NF=%4d f64 FMA  NL=%3d %s : %8.1f ns/iter at 2 waves/SIMD, %8.1f ns/iter at 4 waves/SIMD -> per SIMD: %6.3f ns per wave-FMA, %6.2f ns per CU per LDS op\n",
           NF, NL, KIND ? "ds_read_b64" : "ds_add_f64 ", res[0], res[1], NF ? res[1] / (4.0 * NF) : 0.0,
           NL ? res[1] / (16.0 * NL) : 0.0);
    return res[1];
}

int main() {
    double *o; long long *c, *rt;
    hipMalloc(&o, sizeof(double) * 512 * 512); hipMalloc(&c, sizeof(long long) * 512); hipMalloc(&rt, sizeof(long long) * 512);
    run<512, 0, 0>(o, c, rt);
    run<0, 64, 0>(o, c, rt);
    run<512, 64, 0>(o, c, rt);
    run<256, 64, 0>(o, c, rt);
    run<1024, 64, 0>(o, c, rt);
    run<0, 64, 1>(o, c, rt);
    run<512, 64, 1>(o, c, rt);
    run<512, 32, 0>(o, c, rt);
    run<0, 32, 0>(o, c, rt);
    return 0;
}
