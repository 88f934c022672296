// stream_bw2.hip -- which feature of the streaming front end costs bandwidth?  Persistent waves, 1024 ticket heads.
//   V0 double-buffered groups (next group's loads issued before this group is folded)
//   V1 single-buffered: ticket for the next group, then this group's loads, fold, repeat
//   V2 = V1 + per-tile quiet test (max/min, ballot, branch) + zero-word and tile-info stores
//   V3 = V2 with 5 KiB of dynamic LDS per wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v2s(uint32_t w) { return __builtin_bit_cast(v2s, w); }

template <int G, int V>
__global__ __launch_bounds__(64) void k(const v4u *src, uint64_t ntiles, uint32_t *heads, uint32_t H, uint64_t *words, uint32_t *info, uint32_t *out, int L) {
    extern __shared__ unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t h = blockIdx.x % H;
    uint32_t *head = heads + 16 * h;
    const uint64_t ng = ntiles / G;
    uint32_t acc = 0;
    auto ticket = [&]() { uint32_t t = 0; if (tid == 0) t = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return t; };
    auto resolve = [&](uint32_t t) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t) * H + h; };
    uint64_t g = resolve(ticket());
    if (V == 0) {
        if (g >= ng) return;
        v4u q[G][2];
        { const v4u *b = src + g * G * 128 + tid;
#pragma unroll
          for (int d = 0; d < G; ++d) { q[d][0] = __builtin_nontemporal_load(b + d * 128); q[d][1] = __builtin_nontemporal_load(b + d * 128 + 64); } }
        for (;;) {
            const uint64_t gn = resolve(ticket());
            v4u r[G][2];
            const bool more = gn < ng;
            if (more) { const v4u *b = src + gn * G * 128 + tid;
#pragma unroll
                for (int d = 0; d < G; ++d) { r[d][0] = __builtin_nontemporal_load(b + d * 128); r[d][1] = __builtin_nontemporal_load(b + d * 128 + 64); } }
#pragma unroll
            for (int d = 0; d < G; ++d) { acc |= q[d][0].x | q[d][0].y | q[d][0].z | q[d][0].w | q[d][1].x | q[d][1].y | q[d][1].z | q[d][1].w; }
            if (!more) break;
#pragma unroll
            for (int d = 0; d < G; ++d) { q[d][0] = r[d][0]; q[d][1] = r[d][1]; }
        }
    } else {
        while (g < ng) {
            const uint32_t tn = ticket();
            const v4u *b = src + g * G * 128 + tid;
            v4u q[G][2];
#pragma unroll
            for (int d = 0; d < G; ++d) { q[d][0] = __builtin_nontemporal_load(b + d * 128); q[d][1] = __builtin_nontemporal_load(b + d * 128 + 64); }
            if (V == 1) {
#pragma unroll
                for (int d = 0; d < G; ++d) { acc |= q[d][0].x | q[d][0].y | q[d][0].z | q[d][0].w | q[d][1].x | q[d][1].y | q[d][1].z | q[d][1].w; }
            } else {
                uint32_t quiet = 0;
#pragma unroll
                for (int d = 0; d < G; ++d) {
                    v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
                    const uint32_t w[8] = {q[d][0].x, q[d][0].y, q[d][0].z, q[d][0].w, q[d][1].x, q[d][1].y, q[d][1].z, q[d][1].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) { mx = __builtin_elementwise_max(mx, as_v2s(w[i])); mn = __builtin_elementwise_min(mn, as_v2s(w[i])); }
                    const bool loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
                    if (__ballot(loud) == 0) quiet |= 1u << d;
                    else {
                        if (V == 3) { reinterpret_cast<uint32_t *>(smem)[tid] = w[0]; acc += reinterpret_cast<uint32_t *>(smem)[(tid + 1) & 63]; }
                        acc += w[3];
                    }
                }
                // V2: both stores; V4: test only; V5: words only; V6: info only
                if (V != 4 && V != 6 && tid < 4u * G && ((quiet >> (tid >> 2)) & 1u)) *reinterpret_cast<uint4 *>(words + g * G * 8 + 2 * tid) = make_uint4(0, 0, 0, 0);
                if (V != 4 && V != 5 && tid < G) info[g * G + tid] = quiet;
                if (V == 4) acc += quiet;
            }
            g = resolve(tn);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
float best_of(F launch, uint32_t *heads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        (void)hipMemset(heads, 0, 1024 * 16 * 4);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const uint64_t bytes = (argc > 1 ? atoll(argv[1]) : 4ll) << 30;
    v4u *src; uint32_t *heads, *out, *info; uint64_t *words;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&heads, 1024 * 16 * 4); (void)hipMalloc(&out, 4);
    (void)hipMalloc(&words, bytes / 32 + 4096); (void)hipMalloc(&info, bytes / 512 + 4096);
    (void)hipMemset(src, 0, bytes);
    const uint64_t ntiles = bytes / 2048;
    auto tbs = [&](float ms) { return bytes / ms / 1e9; };
    const uint32_t H = 1024;
    for (int waves : {8, 16, 32}) {
        const uint32_t g = 256 * waves;
#define RUN(G, V, lds) tbs(best_of([&] { hipLaunchKernelGGL((k<G, V>), dim3(g), dim3(64), lds, 0, src, ntiles, heads, H, words, info, out, 100); }, heads))
        printf("waves/CU %2d: G4: V1 %.2f V2(both) %.2f V4(test only) %.2f V5(words) %.2f V6(info) %.2f TB/s\n", waves, RUN(4, 1, 0),
               RUN(4, 2, 0), RUN(4, 4, 0), RUN(4, 5, 0), RUN(4, 6, 0));
    }
    return 0;
}
