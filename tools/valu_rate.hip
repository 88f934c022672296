// microbench: fp32 VALU rates on gfx950 (v_fma_f32 vs v_pk_fma_f32 vs pk_mul+pk_add), SGPR tap operand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* taps, int iters) {
    v2f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.0f};
    float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[0])));
    float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[1])));
    v2f x = (v2f){1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) {            // packed fma, scalar tap
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"((v2f){t0, t1}), "v"(x));
                } else if (MODE == 1) {     // two plain fma, scalar tap
                    float a = acc[i].x, b = acc[i].y;
                    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %2, %4, %1" : "+v"(a), "+v"(b) : "s"(t0), "v"(x.x), "v"(x.y));
                    acc[i].x = a; acc[i].y = b;
                } else if (MODE == 2) {     // packed mul + packed add (unfused)
                    v2f p;
                    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"((v2f){t0, t1}), "v"(x));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(p));
                } else {                    // plain mul + add x2
                    float a = acc[i].x, b = acc[i].y, p, q;
                    asm volatile("v_mul_f32 %0, %2, %3\n v_mul_f32 %1, %2, %4" : "=v"(p), "=v"(q) : "s"(t0), "v"(x.x), "v"(x.y));
                    asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3" : "+v"(a), "+v"(b) : "v"(p), "v"(q));
                    acc[i].x = a; acc[i].y = b;
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int blocks, int iters, float* out, float* taps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, taps, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double macs = (double)blocks * 256 * iters * 4 * 16 * 2;   // complex-component MACs
    return macs * 2 / (ms * 1e-3) / 1e12;   // TFLOP/s (mul+add = 2 flop)
}

int main() {
    float *out, *taps; hipMalloc(&out, 4 * 256 * 8192); hipMalloc(&taps, 64);
    float h[2] = {0.999f, 1.001f}; hipMemcpy(taps, h, 8, hipMemcpyHostToDevice);
    for (int wpc : {1, 2, 4, 8}) {   // workgroups per CU (x4 waves)
        int blocks = 256 * wpc;
        printf("wg/CU=%d  pk_fma %.1f TF | fma %.1f TF | pk_mul+pk_add %.1f TF | mul+add %.1f TF\n", wpc,
               run<0>(blocks, 2000, out, taps), run<1>(blocks, 2000, out, taps),
               run<2>(blocks, 2000, out, taps), run<3>(blocks, 2000, out, taps));
    }
    return 0;
}
