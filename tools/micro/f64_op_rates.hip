// Issue rate of the f64 vector instructions the kernel function uses (exp_nonpos: v_rndne_f64, v_cvt_i32_f64, v_ldexp_f64 next to
// v_fma_f64 / v_mul_f64; potrf_inv16: v_rsq_f64, v_rcp_f64), 8 waves per SIMD, 8 independent chains per lane: cycles per wave
// instruction and SIMD.   hipcc --offload-arch=gfx950 -O3 tools/micro/f64_op_rates.hip -o tools/micro/f64_op_rates && tools/micro/f64_op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + 0.001 * (threadIdx.x + 17 * i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) v[i] = fma(v[i], 0.999999, 1e-9);
            else if (OP == 1) v[i] = v[i] * 1.0000001;
            else if (OP == 2) v[i] = __builtin_rint(v[i] * 1.0000001);                 // mul + rndne: subtract OP 1
            else if (OP == 3) v[i] = ldexp(v[i], (it & 1) ? 1 : -1);
            else if (OP == 4) v[i] = (double)(int)v[i] + 0.5;                           // cvt_i32_f64 + cvt_f64_i32 + add
            else if (OP == 5) v[i] = __builtin_amdgcn_rsq(v[i] + 1.5);                  // add + rsq
            else if (OP == 6) v[i] = __builtin_amdgcn_rcp(v[i] + 1.5);                  // add + rcp
            else if (OP == 7) v[i] = v[i] + 1.5;
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char* name, double* out, int blocks, int iters, double ghz) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, iters, 1.25);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, iters, 1.25);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: blocks / CUs workgroups x 1 wave each per SIMD; wave instructions per wave = iters x 8
    const double waves_per_simd = blocks / 256.0;
    const double cyc = ms * 1e-3 * ghz * 1e9 / (waves_per_simd * iters * 8.0);
    printf("%-44s %8.3f ms  -> %6.2f cycles per wave instruction group\n", name, ms, cyc);
    return cyc;
}

int main() {
    double* out;
    const int blocks = 256 * 8;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(double));
    const double ghz = 2.4;     // nominal; the ratios are what matter
    const int it = 20000;
    run<0>("v_fma_f64", out, blocks, it, ghz);
    run<1>("v_mul_f64", out, blocks, it, ghz);
    run<7>("v_add_f64", out, blocks, it, ghz);
    run<2>("v_mul_f64 + v_rndne_f64", out, blocks, it, ghz);
    run<3>("v_ldexp_f64", out, blocks, it, ghz);
    run<4>("v_cvt_i32_f64 + v_cvt_f64_i32 + v_add_f64", out, blocks, it, ghz);
    run<5>("v_add_f64 + v_rsq_f64", out, blocks, it, ghz);
    run<6>("v_add_f64 + v_rcp_f64", out, blocks, it, ghz);
    hipFree(out);
    return 0;
}
