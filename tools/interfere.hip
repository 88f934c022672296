// interfere.hip -- what does a streaming read kernel (the front end's shape: one single-wave workgroup per
// 2 KiB tile, non-temporal loads) lose to a second kernel running beside it on another stream?  The second
// kernel is a persistent "disturber" of one kind at a time (ALU, LDS chains, barriers, dependent global
// loads, scattered global stores, global atomics, scratch traffic, a train of tiny kernels); both with and
// without disjoint CU masks.
//   hipcc --offload-arch=gfx950 -O3 tools/interfere.hip -o tools/interfere && ./tools/interfere [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void fold(v2s &mx, v4u a) {
    mx = __builtin_elementwise_max(mx, __builtin_bit_cast(v2s, a.x));
    mx = __builtin_elementwise_max(mx, __builtin_bit_cast(v2s, a.y));
    mx = __builtin_elementwise_max(mx, __builtin_bit_cast(v2s, a.z));
    mx = __builtin_elementwise_max(mx, __builtin_bit_cast(v2s, a.w));
}

__global__ __launch_bounds__(64) void stream_kernel(const v4u *src, uint32_t *out) {
    const v4u *p = src + (uint64_t)blockIdx.x * 128 + threadIdx.x;
    v4u a = __builtin_nontemporal_load(p);
    v4u c = __builtin_nontemporal_load(p + 64);
    v2s mx = (v2s){0, 0};
    fold(mx, a);
    fold(mx, c);
    if (mx.x == 12345 && mx.y == 321) out[0] = 1;
}

struct DArgs {
    uint32_t *buf;          // 256 MiB of words
    uint32_t words;         // power of two
    uint32_t iters;
    uint32_t *sink;
    uint32_t *stop;         // device word: the disturber runs until it is set (or for iters rounds, whichever comes first)
};

// looked at by one lane of every wave on every 16th round, past the L1 (agent scope)
#define STOPPED(a, i) ((((i) & 15u) == 0) && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load((a).stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0)

__global__ __launch_bounds__(256) void d_alu(DArgs a) {
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
#pragma unroll 16
        for (int k = 0; k < 256; ++k) x = x * y + 0.5f;
    }
    if (x == 123.f) a.sink[0] = 1;
}

__global__ __launch_bounds__(256) void d_lds(DArgs a) {
    __shared__ uint16_t tab[32768];
    for (uint32_t i = threadIdx.x; i < 32768; i += 256) tab[i] = (uint16_t)((i * 40503u + 17u) & 32767u);
    __syncthreads();
    uint32_t v[6];
    for (int j = 0; j < 6; ++j) v[j] = (threadIdx.x * 97u + j * 1031u) & 32767u;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 64; ++k) {
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = tab[v[j]];
        }
    }
    if (v[0] + v[1] + v[2] + v[3] + v[4] + v[5] == 0xffffffffu) a.sink[0] = 1;
}

__global__ __launch_bounds__(256) void d_barrier(DArgs a) {
    __shared__ uint32_t s[256];
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 64; ++k) {
            s[threadIdx.x] = x;
            __syncthreads();
            x += s[(threadIdx.x + 1) & 255];
            __syncthreads();
        }
    }
    if (x == 0xffffffffu) a.sink[0] = 1;
}

__global__ __launch_bounds__(256) void d_chase(DArgs a) {      // dependent scattered global loads (one dword each)
    uint32_t v = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 16; ++k) v = a.buf[v & (a.words - 1)] + v * 1664525u + 1013904223u;
    }
    if (v == 0xffffffffu) a.sink[0] = 1;
}

__global__ __launch_bounds__(256) void d_scatter(DArgs a) {    // scattered 4-byte global stores
    uint32_t v = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 16; ++k) {
            v = v * 1664525u + 1013904223u;
            a.buf[v & (a.words - 1)] = v;
        }
    }
}

__global__ __launch_bounds__(256) void d_rows(DArgs a) {       // coalesced 1 KiB rows at scattered places, read + write
    uint32_t v = blockIdx.x * 2654435761u;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 16; ++k) {
            v = v * 1664525u + 1013904223u;
            const uint32_t row = (v & (a.words - 1)) & ~255u;
            acc += a.buf[row + threadIdx.x];
            a.buf[(row ^ 0x100000u) + threadIdx.x] = acc;
        }
    }
    if (acc == 0xffffffffu) a.sink[0] = 1;
}

__global__ __launch_bounds__(256) void d_atomic(DArgs a) {
    uint32_t v = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 4; ++k) {
            v = v * 1664525u + 1013904223u;
            if ((threadIdx.x & 63u) == 0) atomicAdd(a.buf + (v & 1023u) * 32u, 1u);
        }
    }
}

__global__ __launch_bounds__(256) void d_scratch(DArgs a) {    // private arrays indexed at run time: scratch traffic
    uint32_t loc[64];
    for (int k = 0; k < 64; ++k) loc[k] = threadIdx.x + k;
    uint32_t v = threadIdx.x;
    for (uint32_t i = 0; i < a.iters && !STOPPED(a, i); ++i) {
        for (int k = 0; k < 64; ++k) {
            v = v * 1664525u + 1013904223u;
            loc[v & 63u] += v;
            v += loc[(v >> 8) & 63u];
        }
    }
    if (v == 0xffffffffu) a.sink[0] = 1;
}

__global__ void d_tiny(uint32_t *sink) {
    if (sink[1] == 0xffffffffu) sink[0] = 1;
}

static hipStream_t make_stream(uint32_t pattern) {
    hipStream_t s = nullptr;
    if (pattern == 0) {
        CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    } else {
        uint32_t mask[8];
        for (auto &w : mask) w = pattern;
        CHK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    }
    return s;
}

int main(int argc, char **argv) {
    const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 4;
    const uint64_t bytes = gib << 30, ntiles = bytes / 2048;
    v4u *src = nullptr;
    uint32_t *buf = nullptr, *sink = nullptr, *stop = nullptr;
    CHK(hipMalloc(&src, bytes));
    CHK(hipMemset(src, 1, bytes));
    const uint32_t words = 1u << 26;
    CHK(hipMalloc(&buf, (size_t)words * 4));
    CHK(hipMemset(buf, 0, (size_t)words * 4));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(sink, 0, 64));
    uint32_t *h_flag = nullptr;
    CHK(hipHostMalloc(&h_flag, 64));
    h_flag[0] = 0;
    h_flag[1] = 1;
    stop = sink + 8;
    hipStream_t sc = nullptr;
    CHK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const char *names[] = {"nothing", "alu spin", "lds chains", "barriers", "dependent global loads", "scattered dword stores",
                           "scattered 1 KiB rows r+w", "global atomics", "scratch arrays", "train of tiny kernels"};
    struct Mode { uint32_t ms, md; const char *name; };
    const Mode modes[] = {{0, 0, "no masks"}, {0x55555555u, 0xAAAAAAAAu, "disjoint halves"}};
    for (const Mode &m : modes) {
        hipStream_t ss = make_stream(m.ms), sd = make_stream(m.md);
        printf("== %s: stream kernel over %llu GiB, disturber = 512 workgroups of 256 ==\n", m.name, (unsigned long long)gib);
        for (int d = 0; d < 10; ++d) {
            float best = 1e9f, dist_ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipMemcpyAsync(stop, h_flag, 4, hipMemcpyHostToDevice, sc));
                CHK(hipStreamSynchronize(sc));
                DArgs a{buf, words, 1u << 15, sink, stop};     // (bounded: a few hundred ms at most even if the flag were lost)
                hipEvent_t d0, d1;
                CHK(hipEventCreate(&d0));
                CHK(hipEventCreate(&d1));
                CHK(hipEventRecord(d0, sd));
                switch (d) {
                case 1: hipLaunchKernelGGL(d_alu, 512, 256, 0, sd, a); break;
                case 2: hipLaunchKernelGGL(d_lds, 512, 256, 0, sd, a); break;
                case 3: hipLaunchKernelGGL(d_barrier, 512, 256, 0, sd, a); break;
                case 4: hipLaunchKernelGGL(d_chase, 512, 256, 0, sd, a); break;
                case 5: hipLaunchKernelGGL(d_scatter, 512, 256, 0, sd, a); break;
                case 6: hipLaunchKernelGGL(d_rows, 512, 256, 0, sd, a); break;
                case 7: hipLaunchKernelGGL(d_atomic, 512, 256, 0, sd, a); break;
                case 8: hipLaunchKernelGGL(d_scratch, 512, 256, 0, sd, a); break;
                case 9: for (int k = 0; k < 400; ++k) hipLaunchKernelGGL(d_tiny, 64, 64, 0, sd, sink); break;
                default: break;
                }
                CHK(hipEventRecord(d1, sd));
                CHK(hipEventRecord(e0, ss));
                hipLaunchKernelGGL(stream_kernel, dim3((uint32_t)ntiles), dim3(64), 0, ss, src, sink);
                CHK(hipEventRecord(e1, ss));
                CHK(hipEventSynchronize(e1));
                CHK(hipMemcpyAsync(stop, h_flag + 1, 4, hipMemcpyHostToDevice, sc));    // the disturber outlived the stream kernel
                CHK(hipStreamSynchronize(sc));
                CHK(hipStreamSynchronize(sd));
                float ms = 0, dm = 0;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                CHK(hipEventElapsedTime(&dm, d0, d1));
                if (ms < best) best = ms, dist_ms = dm;
                CHK(hipEventDestroy(d0));
                CHK(hipEventDestroy(d1));
            }
            printf("  beside %-28s %.3f ms  %.0f GB/s   (disturber ran %.3f ms)\n", names[d], best, bytes / best * 1e-6, dist_ms);
            fflush(stdout);
        }
        CHK(hipStreamDestroy(ss));
        CHK(hipStreamDestroy(sd));
    }
    return 0;
}
