// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the OOK rx path.
//
//   front end : SC16Q11 unpack -> FIR -> |.|^2 >= P* -> packed bit words
//               (reference: src/complexf.h:68-77, src/fir.c:302-395,
//                src/ookiedokie.c:171-179 with src/complexf.h:43-58)
//   edges     : bit words -> sorted list of level changes
//               (the content of --rx-rec-dig, src/ookiedokie.c:146-169)
//   fsm       : table-driven symbol state machine run over the edge list,
//               segment-parallel with a fix-point on the carried state
//               (reference per-sample form: src/state_machine.c:421-556,
//                buffer-skip rule: src/device.c:634-658)
//
// Compiled with -ffp-contract=off: every a*b+c below is a separately
// rounded multiply and add, exactly as the reference's scalar build; fused
// multiply-adds are written explicitly with __builtin_fmaf.
#include "kernels.hpp"
#include "common.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <utility>
#include <vector>

#pragma clang fp contract(off)

namespace ookd {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// complexf.h:68-77: (float)v * (1.0f/2048.0f), exact.
__device__ __forceinline__ float2 unpack_iq(uint32_t w) {
    const float s = 1.0f / 2048.0f;
    float2 r;
    r.x = (float)(int16_t)(w & 0xffffu) * s;
    r.y = (float)(int16_t)(w >> 16) * s;
    return r;
}

// One input sample of a capture as float2, honouring halo (index < 0) and
// zero padding (index >= n_valid; bladeRF_file.c:113-117).
__device__ __forceinline__ float2 fetch_sample(const FrontParams &p, const uint32_t *src,
                                               const float2 *srcf, int64_t n) {
    if (n < 0) {
        const int64_t h = (int64_t)p.halo_len + n;
        if (h < 0) return make_float2(0.0f, 0.0f);
        if (p.halo_f32) return reinterpret_cast<const float2 *>(p.halo_f32)[h];
        if (p.halo) return unpack_iq(reinterpret_cast<const uint32_t *>(p.halo)[h]);
        return make_float2(0.0f, 0.0f);
    }
    if ((uint64_t)n >= p.n_valid) return make_float2(0.0f, 0.0f);
    if (srcf) return srcf[n];
    return unpack_iq(src[n]);
}

// Same for int16 inputs, returned raw (packed I | Q << 16).
__device__ __forceinline__ uint32_t fetch_raw(const FrontParams &p, const uint32_t *src, int64_t n) {
    if (n < 0) {
        const int64_t h = (int64_t)p.halo_len + n;
        if (h < 0 || !p.halo) return 0u;
        return reinterpret_cast<const uint32_t *>(p.halo)[h];
    }
    if ((uint64_t)n >= p.n_valid) return 0u;
    return src[n];
}

// ookiedokie.c:171-179: bit = sqrtf(re*re + im*im) >= thr.  sqrtf is
// correctly rounded and monotone, so this equals power >= P*, P* being the
// smallest float whose sqrtf is >= thr (host computes it).  The power keeps
// the reference's three roundings (complexf.h:45).
__device__ __forceinline__ float power_ref(float re, float im) {
    const float rr = re * re;
    const float ii = im * im;
    return rr + ii;
}

// ---------------------------------------------------------------------------
// front end, 1 stage / decimation 1 (fs32_fs4, the 255-tap config)
// ---------------------------------------------------------------------------
//
// One wavefront = 64 lanes = 1024 consecutive outputs, working alone (a
// workgroup is just kFirWaves of them sharing an LDS allocation, no barrier);
// lane t owns outputs 16t..16t+15 and keeps their 32 partial sums in
// registers.  The wave loads its inputs plus the (padded) tap history raw,
// decides from them whether the filter can be skipped (quiet test), and
// otherwise unpacks them ONCE into its LDS window as float2.
// Taps are consumed in chunks of 32 held in SGPRs; within a chunk the lane
// streams 47 consecutive LDS samples, newest first, each feeding up to 16
// accumulators -- so every output sees its taps in the reference order
// (tap 0 / newest sample first, fir.c:313-318).
//
// LDS layout: sample j lives at slot j + (j >> 4): one pad slot per 16
// samples makes the lane stride 17 float2 = 34 dwords, which is conflict
// free for ds_read_b64 (32-lane halves, 64 banks), and keeps every read of
// the unrolled body at  lane_base + compile-time immediate.

// one pad slot per R samples (R = outputs per lane, a power of two)
template <int R>
__host__ __device__ __forceinline__ uint32_t slot(uint32_t j) { return j + j / (uint32_t)R; }

// float2 slots of one wavefront's private window (kept a multiple of 2 = 16 B)
template <int R>
__host__ __device__ __forceinline__ uint32_t fir1_wave_slots(uint32_t Tp) {
    return (slot<R>(64u * R + Tp) + 2u) & ~1u;
}

// Sequential, unfused recomputation of one output (guard-band path).
template <int R>
__device__ __forceinline__ float2 fir1_exact_body(const float2 *lds, uint32_t j_out,
                                                const float *taps, uint32_t ntaps) {
    float re = 0.0f, im = 0.0f;
    for (uint32_t k = 0; k < ntaps; ++k) {
        const float2 x = lds[slot<R>(j_out - k)];
        const float t = taps[k];
        const float pr = t * x.x;
        const float pi = t * x.y;
        re = re + pr;
        im = im + pi;
    }
    return make_float2(re, im);
}

template <int R>
__device__ __noinline__ float2 fir1_exact_output(const float2 *lds, uint32_t j_out,
                                                 const float *taps, uint32_t ntaps) {
    float re = 0.0f, im = 0.0f;
    for (uint32_t k = 0; k < ntaps; ++k) {
        const float2 x = lds[slot<R>(j_out - k)];
        const float t = taps[k];
        const float pr = t * x.x;
        const float pi = t * x.y;
        re = re + pr;
        im = im + pi;
    }
    return make_float2(re, im);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v8f __attribute__((ext_vector_type(8)));

// One complex multiply-accumulate acc(re,im) += tap * x(re,im) as a packed
// fp32 instruction with the tap read from an SGPR pair: `tp` holds two
// consecutive taps, HI picks which one is broadcast to both halves.  Plain
// v_fma_f32 issues at half the packed rate on gfx950 (measured: 77 vs 147
// TFLOP/s), so the packed form is what reaches the fp32 roof.
//   fused : v_pk_fma_f32                     (one rounding per step)
//   exact : v_pk_mul_f32 then v_pk_add_f32   (the reference's two roundings)
template <bool EXACT, bool HI>
__device__ __forceinline__ void cmac(v2f &acc, v2f tp, v2f x) {
    if (EXACT) {
        v2f prod;
        if (HI) {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(prod) : "s"(tp), "v"(x));
        } else {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(prod) : "s"(tp), "v"(x));
        }
        asm("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(prod));
    } else {
        if (HI) {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "s"(tp), "v"(x));
        } else {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(tp), "v"(x));
        }
    }
}

// Compile-time unrolled body of one 32-tap chunk.  Window position W
// (newest first) feeds output r with tap kk = r - W when 0 <= kk < 32, so
// every output receives its taps in ascending order (fir.c:313-318).
template <bool EXACT, int R, int W, int... Rs>
__device__ __forceinline__ void fir1_wstep(v2f *acc, const v2f *tpair, const v2f *base,
                                           std::integer_sequence<int, Rs...>) {
    constexpr int cp = W + kTapChunk;                   // 1 .. R + 31
    const v2f x = base[cp + cp / R];
    ((void)((Rs - W >= 0 && Rs - W < kTapChunk)
                ? (cmac<EXACT, ((Rs - W) & 1) != 0>(acc[Rs], tpair[((Rs - W) & 31) >> 1], x), 0)
                : 0),
     ...);
}

template <bool EXACT, int R, int... Ws>
__device__ __forceinline__ void fir1_chunk(v2f *acc, const v2f *tpair, const v2f *base,
                                           std::integer_sequence<int, Ws...>) {
    (fir1_wstep<EXACT, R, R - 1 - Ws>(acc, tpair, base, std::make_integer_sequence<int, R>{}), ...);
}

typedef short v2s __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// non-temporal 16 B load: the capture is streamed through once (tools/stream_bw.hip: +7 % over plain loads)
__device__ __forceinline__ uint4 ld_nt4(const uint4 *p) {
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ v2s as_v2s(uint32_t w) { return __builtin_bit_cast(v2s, w); }

// Everything behind the staged window of one wave tile: packed MACs over the
// tap chunks, threshold + guard band, bit packing, optional float output.
// Stores the tile's bit words (WT: write-through, visible to a kernel that
// starts while this one still runs) and returns the tile info word
// (level changes inside the tile | first bit << 30 | last bit << 31).
template <bool EXACT, int R, bool WT, bool NOCALL = false>
__device__ __forceinline__ uint32_t fir1_tile_compute(const FrontParams &p, float2 *lds, uint32_t Tp, uint64_t t0,
                                                      uint32_t tid, uint32_t cap, uint64_t *words) {
    uint32_t info = 0;
    // ---- accumulate ----------------------------------------------------------
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (v2f){0.0f, 0.0f};

    const uint32_t nchunks = Tp / kTapChunk;
    for (uint32_t c = 0; c < nchunks; ++c) {
        // 32 taps of this chunk -> 16 SGPR pairs
        const float *tp = p.taps + c * kTapChunk;
        v8f ta, tb, tc, td;
        asm volatile("s_load_dwordx8 %0, %4, 0x0\n\t"
                     "s_load_dwordx8 %1, %4, 0x20\n\t"
                     "s_load_dwordx8 %2, %4, 0x40\n\t"
                     "s_load_dwordx8 %3, %4, 0x60\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(ta), "=&s"(tb), "=&s"(tc), "=&s"(td)
                     : "s"(tp)
                     : "memory");
        const v2f tpair[16] = {
            __builtin_shufflevector(ta, ta, 0, 1), __builtin_shufflevector(ta, ta, 2, 3),
            __builtin_shufflevector(ta, ta, 4, 5), __builtin_shufflevector(ta, ta, 6, 7),
            __builtin_shufflevector(tb, tb, 0, 1), __builtin_shufflevector(tb, tb, 2, 3),
            __builtin_shufflevector(tb, tb, 4, 5), __builtin_shufflevector(tb, tb, 6, 7),
            __builtin_shufflevector(tc, tc, 0, 1), __builtin_shufflevector(tc, tc, 2, 3),
            __builtin_shufflevector(tc, tc, 4, 5), __builtin_shufflevector(tc, tc, 6, 7),
            __builtin_shufflevector(td, td, 0, 1), __builtin_shufflevector(td, td, 2, 3),
            __builtin_shufflevector(td, td, 4, 5), __builtin_shufflevector(td, td, 6, 7)};

        // output r of this lane sits at window index Tp + R*tid + r; tap kc+kk reads
        // Tp + R*tid + r - kc - kk = R*tid + 32*m + (w + 32),  w = r - kk, m = (Tp - kc - 32)/32;
        // R*tid and 32*m are multiples of R, so their pad slots add up separately
        const uint32_t m = nchunks - 1 - c;
        const v2f *base = reinterpret_cast<const v2f *>(lds + (uint32_t)(R + 1) * tid + (32u + 32u / R) * m);
        fir1_chunk<EXACT, R>(acc, tpair, base, std::make_integer_sequence<int, R + kTapChunk - 1>{});
    }

    // ---- threshold, guard band, pack -----------------------------------------
    // Per output: power (packed square + add), one compare per bound whose
    // wave-wide result lands in an SGPR pair, and the lane's own bit shifted
    // into `mask` through the carry (r runs downwards so bit r ends at position
    // r).  The "inside the guard band" masks are OR-ed on the scalar unit.
    const uint64_t o0 = t0 + (uint64_t)tid * R;
    uint32_t mask = 0;
    uint64_t any_unsure = 0;
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
        float pw;
        if (EXACT) {
            pw = power_ref(acc[r].x, acc[r].y);
        } else {
            v2f sq;
            asm("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(acc[r]));
            pw = sq.x + sq.y;
        }
        const uint64_t ge_hi = __ballot(pw >= (EXACT ? p.p_star : p.p_hi));
        asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(mask) : "s"(ge_hi) : "vcc");
        if (!EXACT) any_unsure |= __ballot(pw >= p.p_lo) & ~ge_hi;
    }
    if (!EXACT && any_unsure != 0) {
        // some lane of this wave has a sample inside the band: those lanes redo
        // their borderline samples in the reference's exact order
        uint32_t todo = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v2f sq;
            asm("v_pk_mul_f32 %0, %1, %1" : "=v"(sq) : "v"(acc[r]));
            const float pf = sq.x + sq.y;
            if (pf >= p.p_lo && !(pf >= p.p_hi) && o0 + r < p.n_out) todo |= 1u << r;
        }
        const uint32_t redo = (uint32_t)__popc(todo);
        while (todo) {                          // one copy of the recompute, whichever outputs need it
            const uint32_t r = (uint32_t)__ffs((int)todo) - 1u;
            todo &= todo - 1u;
            // (NOCALL: the streaming kernel keeps loads in flight in registers the compiler does not
            //  know to be busy -- nothing may save / restore them around a call)
            const float2 y = NOCALL ? fir1_exact_body<R>(lds, Tp + R * tid + r, p.taps, p.stage[0].ntaps)
                                    : fir1_exact_output<R>(lds, Tp + R * tid + r, p.taps, p.stage[0].ntaps);
            const float pe = power_ref(y.x, y.y);
            mask = (mask & ~(1u << r)) | ((pe >= p.p_star ? 1u : 0u) << r);
        }
        if (redo && p.recompute_count) atomicAdd(p.recompute_count, (unsigned long long)redo);
    }
    // outputs past the end of the (padded) capture do not exist
    if (o0 + R > p.n_out) {
        const uint32_t keep = o0 >= p.n_out ? 0u : (uint32_t)(p.n_out - o0);
        mask &= (keep >= 32 ? 0xffffffffu : ((1u << keep) - 1u));
    }

    if (p.fir_out) {
        float2 *out = reinterpret_cast<float2 *>(p.fir_out) + (uint64_t)cap * p.n_out;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (o0 + r < p.n_out) out[o0 + r] = make_float2(acc[r].x, acc[r].y);
        }
    }

    // level changes inside the tile (what edge_count would find in these 1024 bits,
    // minus the comparison of the tile's first bit with the tile before)
    {
        const uint32_t keep = o0 >= p.n_out ? 0u : (o0 + R <= p.n_out ? (uint32_t)R : (uint32_t)(p.n_out - o0));
        const uint32_t prev_top = __shfl_up(mask >> (R - 1), 1);       // bit 15 of the lane before
        uint32_t ch = (mask ^ (mask << 1)) & ((1u << R) - 2u);
        if (tid != 0) ch |= (mask ^ prev_top) & 1u;
        ch &= keep >= 32 ? 0xffffffffu : ((1u << keep) - 1u);
        uint32_t cnt = __popc(ch);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(mask & 1u));
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)((mask >> (R - 1)) & 1u), 63);
        // the word that holds the first change: lane l holds bits R l .. R l + R - 1 of the tile
        const uint64_t chl = __ballot(ch != 0);
        const uint32_t widx = chl ? ((uint32_t)__builtin_ctzll(chl) * (uint32_t)R) >> 6 : 0u;
        info = cnt | (widx << kTileWordShift) | (first << 30) | (last << 31) | p.stamp_bits;
    }

    // 64 / R lanes x R bits -> one 64-bit word
    constexpr uint32_t kLanesPerWord = 64u / R;
    uint64_t w64 = (uint64_t)mask << (R * (tid % kLanesPerWord));
#pragma unroll
    for (uint32_t d = 1; d < kLanesPerWord; d <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)w64, (int)d), hi = __shfl_xor((uint32_t)(w64 >> 32), (int)d);
        w64 |= (uint64_t)lo | ((uint64_t)hi << 32);
    }
    if (tid % kLanesPerWord == 0) {
        if (WT) __hip_atomic_store(words + (t0 >> 6) + tid / kLanesPerWord, w64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else words[(t0 >> 6) + tid / kLanesPerWord] = w64;
    }
    return info;
}

template <bool EXACT, int R>
__global__ __launch_bounds__(64 * kFirWgWaves) void fir1_bits_kernel(const FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    constexpr uint32_t kTile = 64u * R;                 // outputs per wavefront
    constexpr int kRounds = (kTile + 256 + 255) / 256;  // 16 B loads per lane (taps <= 256)
    const uint32_t tid = threadIdx.x & 63u;             // lane: every wavefront works alone
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t cap = blockIdx.y;
    const uint64_t t0 = (((uint64_t)blockIdx.x + p.tile_base) * kFirWgWaves + wave) * kTile;
    const uint32_t Tp = p.stage[0].ntaps_pad;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
    float2 *lds = reinterpret_cast<float2 *>(smem_raw) + wave * fir1_wave_slots<R>(Tp);
    uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;

    // ---- load the wave's window: slot j <-> input index t0 - Tp + j ------------
    // 64R + Tp samples in vectors of 4, lane + 64*i, kept RAW in
    // registers until the quiet test has decided whether they are needed.
    const uint32_t nvec = (kTile + Tp) >> 2;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const bool interior = aligned16 && t0 >= Tp && t0 + kTile <= p.n_valid;
    if (interior) {
        const uint4 *src4 = reinterpret_cast<const uint4 *>(src + (t0 - Tp));
        uint4 q[kRounds];
#pragma unroll
        for (int i = 0; i < kRounds; ++i) {
            const uint32_t v = tid + 64u * i;
            q[i] = (64u * (i + 1) <= kTile / 4 || v < nvec) ? ld_nt4(src4 + v) : make_uint4(0, 0, 0, 0);     // read once
        }
        // ---- quiet test ----------------------------------------------------------
        // |y_re|,|y_im| <= sum|h| * max|component|, so a wave whose whole window
        // stays below quiet_lsb cannot reach the threshold: its 1024 bits are 0
        // without running the filter -- exactly what the reference computes.
        v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
#pragma unroll
        for (int i = 0; i < kRounds; ++i) {
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2s(q[i].x), as_v2s(q[i].y)));
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2s(q[i].z), as_v2s(q[i].w)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2s(q[i].x), as_v2s(q[i].y)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2s(q[i].z), as_v2s(q[i].w)));
        }
        const int L = p.quiet_lsb;
        const bool loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
        if (!p.fir_out && __ballot(loud) == 0) {
            // sparse output: nothing is stored; whatever the tile's words and info hold carries an older
            // run's stamp and reads as quiet (tile_live)
            if (!p.sparse) {
                if (tid < kTile / 64) words[(t0 >> 6) + tid] = 0;
                if (tid == 0) p.tile_info[(uint64_t)cap * p.tiles_per_cap + t0 / kTile] = 0;
            }
            if (p.quiet_count && tid == 0) atomicAdd(p.quiet_count + (blockIdx.x % kQuietCounters), 1u);
            return;
        }
#pragma unroll
        for (int i = 0; i < kRounds; ++i) {
            const uint32_t v = tid + 64u * i;
            if (64u * (i + 1) <= kTile / 4 || v < nvec) {
                float2 *dst = lds + slot<R>(4 * v);        // 4 slots, never straddle a pad
                dst[0] = unpack_iq(q[i].x);
                dst[1] = unpack_iq(q[i].y);
                dst[2] = unpack_iq(q[i].z);
                dst[3] = unpack_iq(q[i].w);
            }
        }
    } else {
        // first / last tiles of a capture, halo, unaligned host pointers
        for (uint32_t v = tid; v < nvec; v += 64) {
            const int64_t n = (int64_t)t0 - (int64_t)Tp + 4 * (int64_t)v;
            float2 *dst = lds + slot<R>(4 * v);
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = fetch_sample(p, src, nullptr, n + i);
        }
    }
    // the window is private to this wavefront and the LDS executes one wave's
    // accesses in order: no workgroup barrier, only keep the compiler from
    // moving reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const uint32_t info = fir1_tile_compute<EXACT, R, false>(p, lds, Tp, t0, tid, cap, words);
    if (tid == 0) p.tile_info[(uint64_t)cap * p.tiles_per_cap + t0 / kTile] = info;
}

// ---------------------------------------------------------------------------
// front end, 1 stage / decimation 1, STREAMING form
// ---------------------------------------------------------------------------
//
// The grid form above is one workgroup per wave tile: half a million tiny
// workgroups per GiB, which fill every wave slot and nearly all of the LDS of
// every CU for as long as the grid lasts -- the edge / state machine kernels
// of the chunk (or capture) before starve beside it.  This form is a
// PERSISTENT grid of single-wave workgroups, N per CU (the launcher picks N),
// that pull GROUPS of kStreamGroup = 4 consecutive tiles (2048 outputs, 8 KiB
// of input) from ticket heads:
//   * many heads (head = workgroup % H owns the groups = head mod H): one
//     returning atomic per group, far below what a head sustains, and the
//     active window of the capture stays dense and moves front to back like a
//     hardware-dispatched grid's (tools/stream_bw.hip: 6.6-6.7 TB/s at 8..32
//     waves per CU against 6.9 for the grid shape and 5.3-6.2 for strided or
//     few-head persistent shapes);
//   * residency is capped at N waves and N windows of LDS per CU: the rest of
//     the CU stays free for whatever else is queued on the device;
//   * a wave issues the nine loads of its group at once (nt: the stream is read
//     once; 8.5 KiB in flight per waiting wave) with the ticket of its next
//     group in front of them; inside a group the tap history of a tile is the
//     tail of the tile before, in registers;
//   * a quiet group costs its loads, ONE store of zero words and ONE store of
//     tile infos;
//   * chunk pipelining: a wave counts the groups it finished per chunk and adds
//     them to done[chunk] when it moves on to a later chunk (its stores are
//     write-through and complete by then), so a stream waiting for
//     done[c] == groups of chunk c may read that prefix of the bit words while
//     this kernel still runs.
// Tile arithmetic is the grid form's (fir1_tile_compute), bit for bit.
// (Deeper software pipelines -- a ring of tiles in registers, double-buffered
//  groups -- were tried: the compiler's wait counts turn conservative across the
//  loop (vmcnt(0) behind the issue), and inline-asm loads are unsafe because the
//  register allocator copies their destination registers while they are in flight.)

struct StreamGroup {            // raw samples of one group: lane L of c[i][0] holds samples 512 i + 4L .., c[i][1] + 256
    v4u c[kStreamGroup][2];
    v4u hal;                    // lanes hal0..63: the Tp samples before the group
};

__device__ __forceinline__ v4u ld_nt(const uint4 *p) {
    return __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
}

constexpr uint32_t kNoGroup = 0xffffffffu;

template <bool EXACT, bool WT>
__global__ __launch_bounds__(64) void fir1_stream_kernel(const FrontParams p, const StreamCtl ctl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int R = kFir1RShort;
    constexpr uint32_t kTile = 64u * R;                 // 512 outputs
    constexpr uint32_t kGroupOut = kTile * kStreamGroup;
    static_assert(kTile == 512, "two 16 B loads per lane and tile");
    const uint32_t tid = threadIdx.x;
    const uint32_t Tp = p.stage[0].ntaps_pad;           // <= 256
    const uint32_t hal0 = 64u - Tp / 4u;                // lanes hal0 .. 63 of a history register are valid
    float2 *lds = reinterpret_cast<float2 *>(smem_raw);
    const uint32_t H = ctl.num_heads;
    const uint32_t h = blockIdx.x % H;
    uint32_t *head = ctl.heads + (size_t)kStreamHeadStride * h;
    const uint32_t ngroups = ctl.groups_per_cap * ctl.num_caps;
    const int L = p.quiet_lsb;
    const bool do_quiet = L > 0 && !p.fir_out;
    const bool aligned16 = (reinterpret_cast<uintptr_t>(p.iq) & 15u) == 0 && (p.cap_stride & 3u) == 0;

    // (lane 0 only, written over the zero the other lanes keep -- no merge instruction, so the wait for the atomic
    //  falls where the ticket is resolved.  The build disables the compiler's atomic optimizer,
    //  which would rewrite this as a wave reduction and wait for the result on the spot.)
    auto issue_ticket = [&]() {
        uint32_t t = 0;
        if (tid == 0) t = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    };
    auto resolve_ticket = [&](uint32_t t) {
        const uint64_t g = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t) * H + h;
        return g < ngroups ? (uint32_t)g : kNoGroup;
    };

    // chunk accounting (write-through runs): groups this wave finished in chunk `chunk`
    uint32_t chunk = 0, chunk_count = 0;
    auto flush_chunk = [&]() {
        if (WT && ctl.done && chunk_count) {
            // every store of those groups has reached memory before the chunk counter moves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_fetch_add(ctl.done + chunk, chunk_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        chunk_count = 0;
    };

    const bool strided = ctl.static_stride != 0;        // experiment: no tickets, group = workgroup + k * grid
    uint32_t g = strided ? (blockIdx.x < ngroups ? blockIdx.x : kNoGroup) : resolve_ticket(issue_ticket());
    while (g != kNoGroup) {
        const uint32_t t_next = strided ? 0u : issue_ticket();         // returns while the group's loads are waited for
        const uint32_t cap = g / ctl.groups_per_cap;
        const uint64_t g0 = (uint64_t)(g - cap * ctl.groups_per_cap) * kGroupOut;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
        uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
        // a group whose window lies inside the capture and can be fetched 16 B at a time
        const bool inside = aligned16 && g0 >= Tp && g0 + kGroupOut <= p.n_valid;
        StreamGroup q{};
        if (inside) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(src + g0) + tid;
#pragma unroll
            for (int i = 0; i < kStreamGroup; ++i) {
                q.c[i][0] = ld_nt(s4 + 128 * i);
                q.c[i][1] = ld_nt(s4 + 128 * i + 64);
            }
            q.hal = ld_nt(s4 - 64);     // (lanes below hal0 fetch samples further back: never used)
        }

        uint32_t quiet_mask = 0, info_vec = 0;
        bool info_vec_any = false;
#pragma unroll
        for (int i = 0; i < kStreamGroup; ++i) {
            const uint64_t t0 = g0 + (uint64_t)i * kTile;
            const v4u hal = i == 0 ? q.hal : q.c[i > 0 ? i - 1 : 0][1];    // the tile's tap history: the tail of the tile before
            bool loud = true;
            if (inside) {
                if (do_quiet) {
                    v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
                    const uint32_t w[12] = {q.c[i][0].x, q.c[i][0].y, q.c[i][0].z, q.c[i][0].w, q.c[i][1].x, q.c[i][1].y,
                                            q.c[i][1].z, q.c[i][1].w, hal.x,       hal.y,       hal.z,       hal.w};
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        // lanes below hal0 hold no history (older samples: masked)
                        const uint32_t v = (k >= 8 && tid < hal0) ? 0u : w[k];
                        mx = __builtin_elementwise_max(mx, as_v2s(v));
                        mn = __builtin_elementwise_min(mn, as_v2s(v));
                    }
                    loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
                    loud = __ballot(loud) != 0;
                }
                if (loud) {
                    // window slot j <-> input index t0 - Tp + j
                    if (tid >= hal0) {
                        float2 *dst = lds + slot<R>(4u * (tid - hal0));
                        dst[0] = unpack_iq(hal.x);
                        dst[1] = unpack_iq(hal.y);
                        dst[2] = unpack_iq(hal.z);
                        dst[3] = unpack_iq(hal.w);
                    }
                    float2 *d0 = lds + slot<R>(Tp + 4u * tid);
                    d0[0] = unpack_iq(q.c[i][0].x);
                    d0[1] = unpack_iq(q.c[i][0].y);
                    d0[2] = unpack_iq(q.c[i][0].z);
                    d0[3] = unpack_iq(q.c[i][0].w);
                    float2 *d1 = lds + slot<R>(Tp + 256u + 4u * tid);
                    d1[0] = unpack_iq(q.c[i][1].x);
                    d1[1] = unpack_iq(q.c[i][1].y);
                    d1[2] = unpack_iq(q.c[i][1].z);
                    d1[3] = unpack_iq(q.c[i][1].w);
                }
            } else {
                // first / last groups of a capture, halo of a shard, unaligned pointers
                const uint32_t nvec = (kTile + Tp) >> 2;
                for (uint32_t v = tid; v < nvec; v += 64) {
                    const int64_t n = (int64_t)t0 - (int64_t)Tp + 4 * (int64_t)v;
                    float2 *dst = lds + slot<R>(4 * v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[k] = fetch_sample(p, src, nullptr, n + k);
                }
            }
            if (loud) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t info = fir1_tile_compute<EXACT, R, WT, true>(p, lds, Tp, t0, tid, cap, words);
                if (tid == (uint32_t)i) info_vec = info;
                info_vec_any = info_vec_any || info != 0;
                // the next loud tile rewrites the window: the reads above are done (the LDS is in order)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                quiet_mask |= 1u << i;
            }
        }
        // sparse output (p.sparse): quiet tiles store nothing -- their words and infos are zero already
        // (launch_clear_tiles); small stores into the read stream cost 3 x their share of the bytes
        // (tools/stream_bw2.hip).  Loud tiles' words were stored by fir1_tile_compute.
        if (!p.sparse) {
            // zero words of the quiet tiles: 8 words per tile, lane l covers words 2l, 2l+1 of the group = tile l / 4
            if (tid < 4u * kStreamGroup && ((quiet_mask >> (tid >> 2)) & 1u)) {
                uint64_t *w = words + (g0 >> 6) + 2u * tid;
                if (WT) {
                    __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(w + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *reinterpret_cast<uint4 *>(w) = make_uint4(0, 0, 0, 0);
                }
            }
            if (tid < (uint32_t)kStreamGroup) {
                uint32_t *ti = p.tile_info + (uint64_t)cap * p.tiles_per_cap + (g0 / kTile) + tid;
                if (WT) __hip_atomic_store(ti, info_vec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *ti = info_vec;
            }
        } else if (info_vec_any) {
            // infos of the loud tiles only (lane i holds tile i's)
            if (tid < (uint32_t)kStreamGroup && !((quiet_mask >> tid) & 1u)) {
                uint32_t *ti = p.tile_info + (uint64_t)cap * p.tiles_per_cap + (g0 / kTile) + tid;
                if (WT) __hip_atomic_store(ti, info_vec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *ti = info_vec;
            }
        }
        if (p.quiet_count && tid == 0 && quiet_mask) atomicAdd(p.quiet_count + (g % kQuietCounters), (uint32_t)__popc(quiet_mask));
        if (WT && ctl.done) {
            uint32_t c = chunk;
            while (c + 1 < ctl.num_chunks && g >= ctl.chunk_end[c]) ++c;
            if (c != chunk) {
                flush_chunk();
                chunk = c;
            }
            chunk_count++;
        }
        if (strided) g = (uint64_t)g + gridDim.x < ngroups ? g + gridDim.x : kNoGroup;
        else g = resolve_ticket(t_next);
    }
    flush_chunk();
}

// ---------------------------------------------------------------------------
// front end, two decimating stages (the backend default fs128_fs16_dec4:
// 16 taps / 2 then 32 taps / 2) -- SURVEY.md 8(f) row f1
// ---------------------------------------------------------------------------
//
// Same execution model as the 1-stage kernel: one wavefront = one tile, no
// workgroup barrier, raw loads -> quiet test -> unpack into a private LDS
// window -> register-blocked packed MACs with the taps in SGPRs.  Here the
// tile is F = 64*R2 final outputs; stage 1 first produces the L1 intermediate
// outputs stage 2 needs (its own halo recomputed per tile, ~12 %) into a
// second LDS window.  Stage output J reads inputs D*(J+1)-1-k (fir.c:290:
// the countdown starts at D); every stage's geometry is a compile-time
// constant, so all LDS addresses in the MAC bodies are lane_base + immediate.
//
// Arithmetic is always the reference's (separately rounded multiply and
// add): floats and bits are bit-identical, no guard band needed.
//
// LDS layout per level: lane t's block of P = D*R consecutive samples starts
// at slot t*(P+1) -- one pad per block makes the lane stride odd, i.e.
// conflict free for ds_read_b64.

template <int D1_, int N1_, int D2_, int N2_, int R2_>
struct Fir2Geom {
    static constexpr int D1 = D1_, D2 = D2_, R2 = R2_;
    static constexpr int T1 = 16 * N1_, T2 = 16 * N2_;         // padded tap counts
    static constexpr int N1 = N1_, N2 = N2_;
    static constexpr int F = 64 * R2;                           // final outputs per wave
    static constexpr int L1need = D2 * (F - 1) + T2;            // stage-1 outputs stage 2 reads
    static constexpr int R1 = (L1need + 63) / 64;
    static constexpr int L1 = 64 * R1;                          // stage-1 outputs computed
    static constexpr int L0 = D1 * (L1 - 1) + T1;               // input samples read
    static constexpr int P1 = D1 * R1, P2 = D2 * R2;
    static constexpr int kVecs = (L0 + 3 + 3) / 4;              // 16 B loads (the window may start mid-vector)
    static constexpr int kVecRounds = (kVecs + 63) / 64;
    static constexpr int slots0 = (L0 + L0 / P1 + 2 + 1) & ~1;  // level 0 stays RAW: 4 B per sample
    static constexpr int slots1 = L1 + L1 / P2 + 2;             // level 1: float2
    static constexpr int wave_bytes = ((slots0 * 4 + slots1 * 8) + 15) & ~15;
};

// One 16-tap chunk of a decimating stage: window position W (newest first)
// feeds output r with tap kk = D*r - D*(R-1) + W when 0 <= kk < 16.
__device__ __forceinline__ v2f lds_sample(const v2f *base, int i) { return base[i]; }
__device__ __forceinline__ v2f lds_sample(const uint32_t *base, int i) {       // raw SC16Q11 level
    const float2 v = unpack_iq(base[i]);
    return (v2f){v.x, v.y};
}

template <bool EXACT, int D, int R, int P, int CHUNK_OFF, int W, typename In, int... Rs>
__device__ __forceinline__ void fir2_wstep(v2f *acc, const v2f *tpair, const In *base,
                                           std::integer_sequence<int, Rs...>) {
    // sample index inside the lane's window: c = Tpad-1 + D*(R-1) - 16*chunk - W
    constexpr int c = CHUNK_OFF + D * (R - 1) - W;
    static_assert(c >= 0, "window underflow");
    const v2f x = lds_sample(base, c + c / P);
    ((void)((D * Rs - D * (R - 1) + W >= 0 && D * Rs - D * (R - 1) + W < 16)
                ? (cmac<EXACT, ((D * Rs - D * (R - 1) + W) & 1) != 0>(
                       acc[Rs], tpair[((D * Rs - D * (R - 1) + W) & 15) >> 1], x),
                   0)
                : 0),
     ...);
}

template <bool EXACT, int D, int R, int P, int CHUNK_OFF, typename In, int... Ws>
__device__ __forceinline__ void fir2_chunk_body(v2f *acc, const v2f *tpair, const In *base,
                                                std::integer_sequence<int, Ws...>) {
    (fir2_wstep<EXACT, D, R, P, CHUNK_OFF, Ws>(acc, tpair, base, std::make_integer_sequence<int, R>{}), ...);
}

// All chunks of one stage, taps kk ascending across chunks (reference order).
template <bool EXACT, int D, int R, int P, int TPAD, int CH, typename In>
__device__ __forceinline__ void fir2_stage_chunk(v2f *acc, const float *taps, const In *base) {
    v8f ta, tb;
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\t"
                 "s_load_dwordx8 %1, %2, 0x20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(ta), "=&s"(tb)
                 : "s"(taps + 16 * CH)
                 : "memory");
    const v2f tpair[8] = {
        __builtin_shufflevector(ta, ta, 0, 1), __builtin_shufflevector(ta, ta, 2, 3),
        __builtin_shufflevector(ta, ta, 4, 5), __builtin_shufflevector(ta, ta, 6, 7),
        __builtin_shufflevector(tb, tb, 0, 1), __builtin_shufflevector(tb, tb, 2, 3),
        __builtin_shufflevector(tb, tb, 4, 5), __builtin_shufflevector(tb, tb, 6, 7)};
    // tap k = 16*CH + kk of output r reads window sample (TPAD-1) + D*r - k
    fir2_chunk_body<EXACT, D, R, P, TPAD - 1 - 16 * CH>(acc, tpair, base,
                                                 std::make_integer_sequence<int, D*(R - 1) + 16>{});
}

template <bool EXACT, int D, int R, int P, int TPAD, typename In, int... CHs>
__device__ __forceinline__ void fir2_stage(v2f *acc, const float *taps, const In *base,
                                           std::integer_sequence<int, CHs...>) {
    (fir2_stage_chunk<EXACT, D, R, P, TPAD, CHs>(acc, taps, base), ...);
}

// Reference-order recomputation of one final output (guard-band path): the
// T2 stage-1 outputs it reads, each from the raw level-0 window, then stage 2.
// `j` = final output index inside the tile.
template <typename G>
__device__ __noinline__ float2 fir2_exact_output(const uint32_t *lds0, const float *taps1, uint32_t ntaps1,
                                                 const float *taps2, uint32_t ntaps2, uint32_t j) {
    float re2 = 0.0f, im2 = 0.0f;
    for (uint32_t k2 = 0; k2 < ntaps2; ++k2) {
        const uint32_t j1 = G::D2 * j + (G::T2 - 1) - k2;               // local stage-1 index
        float re1 = 0.0f, im1 = 0.0f;
        for (uint32_t k1 = 0; k1 < ntaps1; ++k1) {
            const uint32_t i = G::D1 * j1 + (G::T1 - 1) - k1;           // local input index
            const float2 x = unpack_iq(lds0[i + i / G::P1]);
            const float t = taps1[k1];
            const float pr = t * x.x;
            const float pi = t * x.y;
            re1 = re1 + pr;
            im1 = im1 + pi;
        }
        const float t2 = taps2[k2];
        const float pr = t2 * re1;
        const float pi = t2 * im1;
        re2 = re2 + pr;
        im2 = im2 + pi;
    }
    return make_float2(re2, im2);
}

constexpr int kFir2Waves = 1;           // single-wave workgroups: a loud wave must not pin the LDS / wave slots of finished quiet ones

template <typename G, bool EXACT>
__global__ __launch_bounds__(64 * kFir2Waves) void fir2_bits_kernel(const FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const uint32_t tid = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t cap = blockIdx.y;
    const uint64_t J0 = (((uint64_t)blockIdx.x + p.tile_base) * kFir2Waves + wave) * G::F;     // first final output
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
    uint32_t *lds0 = reinterpret_cast<uint32_t *>(smem_raw + wave * G::wave_bytes);    // raw I,Q pairs
    float2 *lds1 = reinterpret_cast<float2 *>(lds0 + G::slots0);
    uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;

    // global index of local stage-1 output 0 and of local input sample 0
    const int64_t j1_0 = (int64_t)G::D2 * (int64_t)J0 + (G::D2 - 1) - (G::T2 - 1);
    const int64_t a0 = (int64_t)G::D1 * j1_0 + (G::D1 - 1) - (G::T1 - 1);

    // ---- load + quiet test + unpack (as in fir1_bits_kernel) ---------------------
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const int64_t va = a0 & ~(int64_t)3;
    const bool interior = aligned16 && a0 >= 0 && (uint64_t)(va + 4 * G::kVecs) <= p.n_valid;
    if (interior) {
        const uint4 *src4 = reinterpret_cast<const uint4 *>(src + va);
        uint4 q[G::kVecRounds];
#pragma unroll
        for (int i = 0; i < G::kVecRounds; ++i) {
            const uint32_t v = tid + 64u * i;
            q[i] = (64 * (i + 1) <= G::kVecs || v < (uint32_t)G::kVecs) ? src4[v] : make_uint4(0, 0, 0, 0);
        }
        v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
#pragma unroll
        for (int i = 0; i < G::kVecRounds; ++i) {
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2s(q[i].x), as_v2s(q[i].y)));
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2s(q[i].z), as_v2s(q[i].w)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2s(q[i].x), as_v2s(q[i].y)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2s(q[i].z), as_v2s(q[i].w)));
        }
        const int L = p.quiet_lsb;
        const bool loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
        if (!p.fir_out && __ballot(loud) == 0) {
            // (sparse output: nothing is stored -- what the tile's words and info hold carries an older run's stamp
            //  and reads as quiet, tile_live)
            if (!p.sparse) {
                if (tid < G::F / 64) words[(J0 >> 6) + tid] = 0;
                if (tid == 0) p.tile_info[(uint64_t)cap * p.tiles_per_cap + J0 / G::F] = 0;
            }
            if (p.quiet_count && tid == 0) atomicAdd(p.quiet_count + (blockIdx.x % kQuietCounters), 1u);
            return;
        }
        const int shift = (int)(a0 - va);           // 0..3 samples of the first vector precede the window
#pragma unroll
        for (int i = 0; i < G::kVecRounds; ++i) {
            const int i0 = 4 * (int)(tid + 64u * i) - shift;        // local index of q[i].x
            const uint32_t w4[4] = {q[i].x, q[i].y, q[i].z, q[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int li = i0 + e;
                if (li >= 0 && li < G::L0) lds0[li + li / G::P1] = w4[e];
            }
        }
    } else {
        // first / last tiles, halo: values outside the capture are zeros or the
        // previous shard's samples -- all of them int16, so the level stays raw
        for (int li = (int)tid; li < G::L0; li += 64) {
            lds0[li + li / G::P1] = fetch_raw(p, src, a0 + li);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 1: lane t -> local outputs R1*t .. R1*t + R1-1 ---------------------------
    {
        v2f acc[G::R1];
#pragma unroll
        for (int r = 0; r < G::R1; ++r) acc[r] = (v2f){0.0f, 0.0f};
        const uint32_t *base = lds0 + (G::P1 + 1) * tid;
        fir2_stage<EXACT, G::D1, G::R1, G::P1, G::T1>(acc, p.taps + p.stage[0].tap_off, base,
                                               std::make_integer_sequence<int, G::N1>{});
#pragma unroll
        for (int r = 0; r < G::R1; ++r) {
            const int j = G::R1 * (int)tid + r;
            lds1[j + j / G::P2] = make_float2(acc[r].x, acc[r].y);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- stage 2: lane t -> final outputs J0 + R2*t .. + R2-1 ------------------------------
    v2f acc[G::R2];
#pragma unroll
    for (int r = 0; r < G::R2; ++r) acc[r] = (v2f){0.0f, 0.0f};
    {
        const v2f *base = reinterpret_cast<const v2f *>(lds1 + (G::P2 + 1) * tid);
        fir2_stage<EXACT, G::D2, G::R2, G::P2, G::T2>(acc, p.taps + p.stage[1].tap_off, base,
                                               std::make_integer_sequence<int, G::N2>{});
    }

    // ---- threshold + pack: R2 bits per lane, 64 / R2 lanes per word ----------------------------
    static_assert(G::R2 == 4, "bit packing below assumes 4 outputs per lane");
    const uint64_t o0 = J0 + (uint64_t)tid * G::R2;
    uint32_t nib = 0;
    float2 *fout = p.fir_out ? reinterpret_cast<float2 *>(p.fir_out) + (uint64_t)cap * p.n_out : nullptr;
#pragma unroll
    for (int r = 0; r < G::R2; ++r) {
        const bool valid = o0 + r < p.n_out;
        const float pw = power_ref(acc[r].x, acc[r].y);
        bool bit = pw >= (EXACT ? p.p_star : p.p_hi);
        if (!EXACT && valid && !bit && pw >= p.p_lo) {
            // inside the guard band: redo this output in the reference's exact order
            const float2 y = fir2_exact_output<G>(lds0, p.taps + p.stage[0].tap_off, p.stage[0].ntaps,
                                                  p.taps + p.stage[1].tap_off, p.stage[1].ntaps, G::R2 * tid + r);
            bit = power_ref(y.x, y.y) >= p.p_star;
            if (p.recompute_count) atomicAdd(p.recompute_count, 1ull);
        }
        nib |= ((valid && bit) ? 1u : 0u) << r;
        if (fout && valid) fout[o0 + r] = make_float2(acc[r].x, acc[r].y);
    }
    {
        // level changes inside the tile, as in the 1-stage kernel
        const uint32_t keep = o0 >= p.n_out ? 0u : (o0 + G::R2 <= p.n_out ? (uint32_t)G::R2 : (uint32_t)(p.n_out - o0));
        const uint32_t prev_top = __shfl_up(nib >> (G::R2 - 1), 1);
        uint32_t ch = (nib ^ (nib << 1)) & ((1u << G::R2) - 2u);
        if (tid != 0) ch |= (nib ^ prev_top) & 1u;
        ch &= (1u << keep) - 1u;
        uint32_t cnt = __popc(ch);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(nib & 1u));
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)((nib >> (G::R2 - 1)) & 1u), 63);
        const uint64_t chl = __ballot(ch != 0);         // (the word that holds the first change: R2 bits per lane)
        const uint32_t widx = chl ? ((uint32_t)__builtin_ctzll(chl) * (uint32_t)G::R2) >> 6 : 0u;
        if (tid == 0) p.tile_info[(uint64_t)cap * p.tiles_per_cap + J0 / G::F] = cnt | (widx << kTileWordShift) | (first << 30) | (last << 31) | p.stamp_bits;
    }
    uint32_t half = nib << (4u * (tid & 7u));
    half |= __shfl_xor(half, 1);
    half |= __shfl_xor(half, 2);
    half |= __shfl_xor(half, 4);                // lanes 8g..8g+7 hold outputs 32g..32g+31
    const uint32_t hi = __shfl_down(half, 8);
    if ((tid & 15u) == 0) words[(J0 >> 6) + (tid >> 4)] = (uint64_t)half | ((uint64_t)hi << 32);
}

typedef Fir2Geom<2, 1, 2, 2, 4> Fir2Dec4;      // fs128_fs16_dec4: (D 2, 16 taps), (D 2, 32 taps)

// ---------------------------------------------------------------------------
// front end, generic: any number of stages / decimations, exact arithmetic
// ---------------------------------------------------------------------------
//
// One workgroup = kGenTile final outputs.  Level 0 is the unpacked input,
// level s+1 the output of stage s; the slice of every level the tile needs
// is produced in LDS, ping-ponging between two buffers.  Stage output J
// (global) reads level-s inputs D*(J+1)-1-k (fir.c:290: the countdown starts
// at D, so the first output is at input index D-1).

struct GenLevel {
    int64_t a;          // first local index needed at this level
    uint32_t len;       // samples needed
};

__device__ __forceinline__ void gen_levels(const FrontParams &p, int64_t j0, uint32_t L,
                                           GenLevel *lv, int64_t *off) {
    // global origin of each level: g_{s+1} = floor(g_s / D_s)
    uint64_t g = p.origin;
    const int S = (int)p.num_stages;
    for (int s = 0; s < S; ++s) {
        const uint64_t D = p.stage[s].decim;
        const uint64_t gn = g / D;
        off[s] = (int64_t)(D * gn + D - 1) - (int64_t)g;    // = D-1-(g mod D)
        g = gn;
    }
    lv[S].a = j0;
    lv[S].len = L;
    for (int s = S - 1; s >= 0; --s) {
        const int64_t D = p.stage[s].decim;
        const int64_t T = p.stage[s].ntaps;
        lv[s].a = D * lv[s + 1].a + off[s] - (T - 1);
        lv[s].len = (uint32_t)(D * ((int64_t)lv[s + 1].len - 1) + T);
    }
}

template <bool F32IN>
__global__ __launch_bounds__(256) void fir_generic_kernel(const FrontParams p, uint32_t lds_b_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *buf[2] = {reinterpret_cast<float2 *>(smem_raw),
                      reinterpret_cast<float2 *>(smem_raw) + lds_b_off};

    const uint32_t tid = threadIdx.x;
    const uint32_t cap = blockIdx.y;
    const int S = (int)p.num_stages;
    const int64_t j0 = (int64_t)blockIdx.x * kGenTile;
    GenLevel lv[kMaxStages + 1];
    int64_t off[kMaxStages];
    gen_levels(p, j0, kGenTile, lv, off);

    const uint32_t *src = F32IN ? nullptr
                                : reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
    const float2 *srcf = F32IN ? reinterpret_cast<const float2 *>(p.iq_f32) + (uint64_t)cap * p.cap_stride
                               : nullptr;

    for (uint32_t i = tid; i < lv[0].len; i += 256) {
        buf[0][i] = fetch_sample(p, src, srcf, lv[0].a + (int64_t)i);
    }
    __syncthreads();

    uint64_t *words = p.bits ? p.bits + (uint64_t)cap * p.words_per_cap : nullptr;
    float2 *fout = p.fir_out ? reinterpret_cast<float2 *>(p.fir_out) + (uint64_t)cap * p.n_out : nullptr;

    for (int s = 0; s < S; ++s) {
        const float2 *in = buf[s & 1];
        float2 *out = buf[(s + 1) & 1];
        const float *taps = p.taps + p.stage[s].tap_off;
        const int64_t D = p.stage[s].decim;
        const uint32_t T = p.stage[s].ntaps;
        const bool last = (s == S - 1);
        const uint32_t n = lv[s + 1].len;
        for (uint32_t base = 0; base < n; base += 256) {
            const uint32_t i = base + tid;
            float re = 0.0f, im = 0.0f;
            if (i < n) {
                const int64_t jl = lv[s + 1].a + (int64_t)i;
                const int64_t newest = D * jl + off[s] - lv[s].a;   // index into `in`
                for (uint32_t k = 0; k < T; ++k) {
                    const float2 x = in[newest - (int64_t)k];
                    const float t = taps[k];
                    const float pr = t * x.x;
                    const float pi = t * x.y;
                    re = re + pr;
                    im = im + pi;
                }
            }
            if (!last) {
                if (i < n) out[i] = make_float2(re, im);
            } else {
                const int64_t o = j0 + (int64_t)i;
                const bool valid = (i < n) && o >= 0 && (uint64_t)o < p.n_out;
                const bool bit = valid && (power_ref(re, im) >= p.p_star);
                const uint64_t ball = __ballot(bit);
                if (words && lane_id() == 0) {
                    words[((uint64_t)j0 + base + (tid & ~63u)) >> 6] = ball;
                }
                if (fout && valid) fout[o] = make_float2(re, im);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// front end, no filter ("-F none", ookiedokie.c:260-263): threshold the
// unpacked samples directly.  16 B per lane loads; a wave covers 256 samples.
// ---------------------------------------------------------------------------
// One wavefront = 1024 samples (16 per lane, four 16 B loads issued up
// front), same tile / tile-info convention as the 1-stage FIR kernel.
__global__ __launch_bounds__(64) void nofir_bits_kernel(const FrontParams p) {
    const uint32_t tid = threadIdx.x;
    const uint32_t cap = blockIdx.y;
    const uint64_t t0 = ((uint64_t)blockIdx.x + p.tile_base) * kWaveTile;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
    uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    float2 *fout = p.fir_out ? reinterpret_cast<float2 *>(p.fir_out) + (uint64_t)cap * p.n_out : nullptr;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const uint64_t o0 = t0 + 16ull * tid;          // lane owns samples o0 .. o0+15
    uint32_t raw[16];
    if (aligned16 && t0 + kWaveTile <= p.n_valid) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src + o0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = s4[i];
            raw[4 * i] = q.x;
            raw[4 * i + 1] = q.y;
            raw[4 * i + 2] = q.z;
            raw[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) raw[i] = fetch_raw(p, src, (int64_t)(o0 + i));
    }
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 v = unpack_iq(raw[i]);
        const bool valid = o0 + i < p.n_out;
        mask |= ((valid && power_ref(v.x, v.y) >= p.p_star) ? 1u : 0u) << i;
        if (fout && valid) fout[o0 + i] = v;
    }
    {
        // level changes inside the tile (see fir1_bits_kernel)
        const uint32_t keep = o0 >= p.n_out ? 0u : (o0 + 16 <= p.n_out ? 16u : (uint32_t)(p.n_out - o0));
        const uint32_t prev_top = __shfl_up(mask >> 15, 1);
        uint32_t ch = (mask ^ (mask << 1)) & 0xfffeu;
        if (tid != 0) ch |= (mask ^ prev_top) & 1u;
        ch &= (1u << keep) - 1u;
        uint32_t cnt = __popc(ch);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(mask & 1u));
        const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)((mask >> 15) & 1u), 63);
        const uint64_t chl = __ballot(ch != 0);         // (the word that holds the first change: 16 bits per lane)
        const uint32_t widx = chl ? (uint32_t)__builtin_ctzll(chl) >> 2 : 0u;
        if (tid == 0) p.tile_info[(uint64_t)cap * p.tiles_per_cap + (t0 >> 10)] = cnt | (widx << kTileWordShift) | (first << 30) | (last << 31) | p.stamp_bits;
    }
    const uint32_t pair = mask | (__shfl_xor(mask, 1) << 16);      // valid on even lanes
    const uint32_t hi = __shfl_xor(pair, 2);
    if ((tid & 3u) == 0) words[(t0 >> 6) + (tid >> 2)] = (uint64_t)pair | ((uint64_t)hi << 32);
}

// ---------------------------------------------------------------------------
// unpack only (SDR backend rx: complexf.h:68-77 on the GPU)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t *iq, float2 *out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        out[i] = unpack_iq(iq[i]);
    }
}

// ---------------------------------------------------------------------------
// pack only (post-filter recorder: complexf.h:87-96 on the GPU)
// ---------------------------------------------------------------------------
// (int16_t)(x * 2048.0f): truncate towards zero to a 32-bit integer, keep the
// low 16 bits (what the reference's cast compiles to on x86-64).
__device__ __forceinline__ uint32_t pack_iq(float2 v) {
    const int32_t re = (int32_t)(v.x * 2048.0f);
    const int32_t im = (int32_t)(v.y * 2048.0f);
    return ((uint32_t)re & 0xffffu) | ((uint32_t)im << 16);
}

__global__ __launch_bounds__(256) void pack_kernel(const float2 *in, uint32_t *iq, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        iq[i] = pack_iq(in[i]);
    }
}

// ---------------------------------------------------------------------------
// synthetic capture generator
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void synth_kernel(const SynthRun *runs, uint64_t num_runs, uint64_t seed,
                                                    uint32_t noise, uint64_t first, uint64_t count,
                                                    uint32_t *iq) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; base < count; base += stride) {
        const uint64_t n0 = first + base;
        // last run with start <= n0
        uint64_t lo = 0, hi = num_runs;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (runs[mid].start <= n0) lo = mid;
            else hi = mid;
        }
        uint64_t ri = lo;
        uint32_t out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t n = n0 + i;
            while (ri + 1 < num_runs && runs[ri + 1].start <= n) ri++;
            int ni, nq;
            synth_noise(seed, n, noise, ni, nq);
            int vi = (num_runs && runs[ri].start <= n ? runs[ri].i_level : 0) + ni;
            int vq = (num_runs && runs[ri].start <= n ? runs[ri].q_level : 0) + nq;
            vi = max(-32768, min(32767, vi));
            vq = max(-32768, min(32767, vq));
            out[i] = ((uint32_t)vi & 0xffffu) | ((uint32_t)vq << 16);
        }
        if (base + 3 < count && ((reinterpret_cast<uintptr_t>(iq + base) & 15u) == 0)) {
            *reinterpret_cast<uint4 *>(iq + base) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
            for (int i = 0; i < 4; ++i) {
                if (base + i < count) iq[base + i] = out[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

// small tiles only pay through the quiet shortcut: none without it (quiet_lsb 0)
static int fir1_R(const FrontParams &p) {
    if (front_uses_mfma(p)) return (int)kMfmaTile / 64;
    return (p.stage[0].ntaps_pad <= kFir1ShortTaps && p.quiet_lsb > 0) ? kFir1RShort : kFir1RLong;
}

static size_t fir1_lds_bytes(const FrontParams &p) {
    const uint32_t Tp = p.stage[0].ntaps_pad;
    const uint32_t slots = fir1_R(p) == kFir1RShort ? fir1_wave_slots<kFir1RShort>(Tp) : fir1_wave_slots<kFir1RLong>(Tp);
    return (size_t)kFirWgWaves * slots * sizeof(float2);
}

static void gen_level_sizes(const FrontParams &p, uint32_t len[kMaxStages + 1]) {
    const int S = (int)p.num_stages;
    len[S] = kGenTile;
    for (int s = S - 1; s >= 0; --s) {
        len[s] = p.stage[s].decim * (len[s + 1] - 1) + p.stage[s].ntaps;
    }
}

size_t generic_lds_bytes(const FrontParams &p) {
    uint32_t len[kMaxStages + 1];
    gen_level_sizes(p, len);
    uint32_t even = 0, odd = 0;
    for (int s = 0; s < (int)p.num_stages; ++s) {   // the final level is not stored
        if (s & 1) odd = len[s] > odd ? len[s] : odd;
        else even = len[s] > even ? len[s] : even;
    }
    return (size_t)(even + odd + 2) * sizeof(float2);
}

hipError_t launch_front_generic(const FrontParams &p, uint32_t num_captures, hipStream_t stream) {
    uint32_t len[kMaxStages + 1];
    gen_level_sizes(p, len);
    uint32_t even = 0;
    for (int s = 0; s < (int)p.num_stages; s += 2) even = len[s] > even ? len[s] : even;
    const size_t lds = generic_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // cover every bit word of the capture so the tail words are written (as zeros)
    uint64_t tiles = (p.n_out + kGenTile - 1) / kGenTile;
    if (p.bits && p.words_per_cap * 64 / kGenTile > tiles) tiles = p.words_per_cap * 64 / kGenTile;
    if (tiles == 0) return hipSuccess;
    dim3 grid((uint32_t)tiles, num_captures);
    hipError_t e;
    if (p.iq_f32) {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&fir_generic_kernel<true>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_generic_kernel<true>, grid, dim3(256), lds, stream, p, even + 1);
    } else {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&fir_generic_kernel<false>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_generic_kernel<false>, grid, dim3(256), lds, stream, p, even + 1);
    }
    return hipGetLastError();
}

hipError_t ensure_dynamic_lds(const void *func, size_t bytes) {
    struct Granted {
        const void *func;
        int dev;
        size_t bytes;
    };
    static thread_local std::vector<Granted> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (Granted &g : granted) {
        if (g.func == func && g.dev == dev) {
            if (g.bytes >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) g.bytes = bytes;
            return e;
        }
    }
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted.push_back({func, dev, bytes});
    return e;
}

static bool use_fir1(const FrontParams &p) {
    // (the tuned kernels' load rounds cover a tap history of up to 256 samples; longer filters: generic kernel)
    return p.num_stages == 1 && p.stage[0].decim == 1 && p.origin == 0 && !p.iq_f32 &&
           p.stage[0].ntaps_pad <= 256u && fir1_lds_bytes(p) <= 160 * 1024;
}

static bool use_fir2(const FrontParams &p) {
    // every level's phase is D-1 when the origin is a multiple of the total decimation
    return p.num_stages == 2 && !p.iq_f32 && !p.halo_f32 && p.stage[0].decim == 2 && p.stage[1].decim == 2 &&
           p.stage[0].ntaps <= Fir2Dec4::T1 && p.stage[1].ntaps <= Fir2Dec4::T2 && p.origin % 4 == 0;
}

uint64_t front_wave_tiles(const FrontParams &p) {
    if (p.num_stages == 0) return ((p.n_out + kFirTile - 1) / kFirTile) * (kFirTile / kWaveTile);
    if (use_fir1(p)) return ((p.n_out + kFirTile - 1) / kFirTile) * (kFirTile / (64 * fir1_R(p)));
    if (use_fir2(p)) {
        const uint64_t per_wg = (uint64_t)kFir2Waves * Fir2Dec4::F;
        return ((p.n_out + per_wg - 1) / per_wg) * kFir2Waves;
    }
    return 0;
}

uint32_t front_tile_bits(const FrontParams &p) {
    if (p.num_stages == 0) return kWaveTile;
    if (use_fir1(p)) return 64 * fir1_R(p);
    if (use_fir2(p)) return Fir2Dec4::F;
    return 0;
}

// t0 / t1 (optional): events that take the kernel's own start / end time stamps.  For the
// one-kernel front ends they ride on the dispatch itself (hipExtLaunchKernel): no marker
// packets in front of and behind the dominant kernel, and their difference is the kernel's
// duration as a profiler sees it.
hipError_t launch_front(const FrontParams &p, uint32_t num_captures, bool exact, hipStream_t stream,
                        hipEvent_t t0, hipEvent_t t1, uint64_t tile_begin, uint64_t tile_count) {
    if (p.n_out == 0) {
        if (t0 && hipEventRecord(t0, stream) != hipSuccess) return hipGetLastError();
        if (t1 && hipEventRecord(t1, stream) != hipSuccess) return hipGetLastError();
        return hipSuccess;
    }
    // [tile_begin, tile_begin + tile_count) of the capture's wave tiles (tuned kernels only), default: all
    auto range = [&](uint64_t all, uint64_t &grid, FrontParams &pp) {
        const uint64_t b = tile_begin < all ? tile_begin : all;
        grid = tile_count < all - b ? tile_count : all - b;
        pp.tile_base = (uint32_t)b;
    };
    FrontParams pp = p;
    void *args[] = {&pp};
    uint64_t grid = 0;
    if (p.num_stages == 0) {
        // whole 4096-sample blocks, so every bit word of the capture is written
        range((p.n_out + kFirTile - 1) / kFirTile * kFirWaves, grid, pp);
        if (grid == 0) return hipSuccess;
        const hipError_t e = hipExtLaunchKernel(reinterpret_cast<const void *>(&nofir_bits_kernel),
                                                dim3((uint32_t)grid, num_captures), dim3(64), args, 0, stream, t0, t1, 0);
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (use_fir1(p) && front_uses_mfma(p) && !exact) {
        return launch_front_mfma(p, num_captures, stream, t0, t1, tile_begin, tile_count);
    }
    if (use_fir1(p)) {
        const size_t lds = fir1_lds_bytes(p);
        // whole 4096-output blocks, so every bit word of the capture is written
        const int R = fir1_R(p);
        range((p.n_out + kFirTile - 1) / kFirTile * (kFirTile / (64 * R) / kFirWgWaves), grid, pp);
        if (grid == 0) return hipSuccess;
        const void *fn;
        if (R == kFir1RShort) {
            fn = exact ? reinterpret_cast<const void *>(&fir1_bits_kernel<true, kFir1RShort>)
                       : reinterpret_cast<const void *>(&fir1_bits_kernel<false, kFir1RShort>);
        } else {
            fn = exact ? reinterpret_cast<const void *>(&fir1_bits_kernel<true, kFir1RLong>)
                       : reinterpret_cast<const void *>(&fir1_bits_kernel<false, kFir1RLong>);
        }
        static const size_t lds_pad = dev_getenv("OOKD_FIR1_LDS_PAD") ? (size_t)atoi(dev_getenv("OOKD_FIR1_LDS_PAD")) : 0;   // experiment: caps the waves per CU
        const size_t lds_req = lds + lds_pad;
        hipError_t e = ensure_dynamic_lds(fn, lds_req);
        if (e != hipSuccess) return e;
        e = hipExtLaunchKernel(fn, dim3((uint32_t)grid, num_captures), dim3(64 * kFirWgWaves), args, lds_req, stream, t0, t1, 0);
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (use_fir2(p) && front_uses_mfma2(p) && !exact) {
        return launch_front_mfma2(p, num_captures, stream, t0, t1, tile_begin, tile_count);
    }
    if (use_fir2(p)) {
        const size_t lds = (size_t)kFir2Waves * Fir2Dec4::wave_bytes;
        range((p.n_out + (uint64_t)kFir2Waves * Fir2Dec4::F - 1) / ((uint64_t)kFir2Waves * Fir2Dec4::F), grid, pp);
        if (grid == 0) return hipSuccess;
        const void *fn = exact ? reinterpret_cast<const void *>(&fir2_bits_kernel<Fir2Dec4, true>)
                               : reinterpret_cast<const void *>(&fir2_bits_kernel<Fir2Dec4, false>);
        const hipError_t e = hipExtLaunchKernel(fn, dim3((uint32_t)grid, num_captures), dim3(64 * kFir2Waves), args, lds,
                                                stream, t0, t1, 0);
        return e != hipSuccess ? e : hipGetLastError();
    }
    if (tile_begin != 0 || tile_count != ~0ull) return hipErrorInvalidValue;     // the generic kernel runs whole captures
    // several kernels: bracket them
    if (t0 && hipEventRecord(t0, stream) != hipSuccess) return hipGetLastError();
    const hipError_t e = launch_front_generic(p, num_captures, stream);
    if (e != hipSuccess) return e;
    if (t1 && hipEventRecord(t1, stream) != hipSuccess) return hipGetLastError();
    return hipSuccess;
}

bool front_sparse_capable(const FrontParams &p) {
    // (round 3: the two-stage kernels too -- their quiet tiles stored 36 bytes each, 150 MB of small stores per
    //  16 GiB capture beside the read stream: the backend default filter ran 15 % behind the 1-stage one for it)
    return (use_fir1(p) || use_fir2(p)) && p.quiet_lsb > 0 && !p.fir_out;
}

bool front_streams(const FrontParams &p) {
    return !front_uses_mfma(p) && use_fir1(p) && fir1_R(p) == kFir1RShort && p.stage[0].ntaps_pad <= 256u;
}

static int device_cu_count() {
    static thread_local int cached_dev = -1, cached = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != cached_dev) {
        hipDeviceProp_t prop;
        cached = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                     ? prop.multiProcessorCount : 256;
        cached_dev = dev;
    }
    return cached;
}

hipError_t launch_front_stream(const FrontParams &p, StreamCtl ctl, bool exact, bool write_through,
                               hipStream_t stream, hipEvent_t t0, hipEvent_t t1) {
    if (!front_streams(p) || p.tiles_per_cap % kStreamGroup != 0) return hipErrorInvalidValue;
    const size_t lds = (size_t)fir1_wave_slots<kFir1RShort>(p.stage[0].ntaps_pad) * sizeof(float2);
    const uint64_t ngroups = (uint64_t)(p.tiles_per_cap / kStreamGroup) * ctl.num_caps;
    if (ngroups >= 0xfffffff0ull) return hipErrorInvalidValue;
    if (ngroups == 0) {
        if (t0 && hipEventRecord(t0, stream) != hipSuccess) return hipGetLastError();
        if (t1 && hipEventRecord(t1, stream) != hipSuccess) return hipGetLastError();
        return hipSuccess;
    }
    ctl.groups_per_cap = p.tiles_per_cap / kStreamGroup;
    uint64_t grid = (uint64_t)device_cu_count() * (ctl.waves_per_cu ? ctl.waves_per_cu : 12u);
    if (grid > ngroups) grid = ngroups;
    // heads: as many as the grid has waves (each head still hands out groups in order), at most kStreamHeads
    ctl.num_heads = (uint32_t)std::min<uint64_t>(grid, (uint64_t)kStreamHeads);
    const void *fn;
    if (write_through) {
        fn = exact ? reinterpret_cast<const void *>(&fir1_stream_kernel<true, true>)
                   : reinterpret_cast<const void *>(&fir1_stream_kernel<false, true>);
    } else {
        fn = exact ? reinterpret_cast<const void *>(&fir1_stream_kernel<true, false>)
                   : reinterpret_cast<const void *>(&fir1_stream_kernel<false, false>);
    }
    hipError_t e = ensure_dynamic_lds(fn, lds);
    if (e != hipSuccess) return e;
    FrontParams pp = p;
    void *args[] = {&pp, &ctl};
    e = hipExtLaunchKernel(fn, dim3((uint32_t)grid), dim3(64), args, lds, stream, t0, t1, 0);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// sparse front-end output: making the bit words dense again
// ---------------------------------------------------------------------------
// With FrontParams::sparse the tuned kernels (1 stage; 2 x decimate-by-2) store nothing for quiet
// tiles, and what an earlier run left there is told apart by the run stamp in
// the tile info (tile_live): no pass over the tiles between runs (round 2 first
// zeroed the previous run's tiles before every run: 60 us and 107 MB of
// stores per 16 GiB capture, beside another context's front end).  This kernel
// zeroes words + info of every tile that does not carry `keep_stamp_bits` (0:
// every tile with a non-zero info): for readers of the raw words
// (ookd_rx_get_bits), when the run geometry changes, when the stamp wraps.
__global__ __launch_bounds__(256) void clear_tiles_kernel(uint32_t *tile_info, uint64_t *bits, uint64_t ntiles,
                                                          uint32_t tiles_per_cap, uint64_t words_per_cap,
                                                          uint32_t words_per_tile, uint32_t keep_stamp_bits) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; t < ntiles; t += stride) {
        const uint4 q = *reinterpret_cast<const uint4 *>(tile_info + t);       // tiles_per_cap is a multiple of 8
        if ((q.x | q.y | q.z | q.w) == 0) continue;
        uint32_t info[4] = {q.x, q.y, q.z, q.w};
        for (uint32_t k = 0; k < 4; ++k) {
            if (!info[k]) continue;
            if (keep_stamp_bits && !((info[k] ^ keep_stamp_bits) & kTileStampMask)) continue;      // this run's
            const uint64_t tt = t + k;
            const uint64_t cap = tt / tiles_per_cap, tile = tt - cap * tiles_per_cap;
            uint64_t *w = bits + cap * words_per_cap + tile * words_per_tile;
            for (uint32_t i = 0; i < words_per_tile; i += 2) *reinterpret_cast<uint4 *>(w + i) = make_uint4(0, 0, 0, 0);
            info[k] = 0;
        }
        *reinterpret_cast<uint4 *>(tile_info + t) = make_uint4(info[0], info[1], info[2], info[3]);
    }
}

hipError_t launch_clear_tiles(uint32_t *tile_info, uint64_t *bits, uint64_t ntiles, uint32_t tiles_per_cap,
                              uint64_t words_per_cap, uint32_t tile_bits, uint32_t keep_stamp_bits, hipStream_t stream) {
    if (ntiles == 0) return hipSuccess;
    uint64_t blocks = (ntiles / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(clear_tiles_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, tile_info, bits, ntiles,
                       tiles_per_cap, words_per_cap, tile_bits / 64u, keep_stamp_bits);
    return hipGetLastError();
}

hipError_t launch_unpack(const int16_t *iq, float *out, uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpack_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const uint32_t *>(iq), reinterpret_cast<float2 *>(out), n);
    return hipGetLastError();
}

hipError_t launch_pack(const float *in, int16_t *iq, uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream,
                       reinterpret_cast<const float2 *>(in), reinterpret_cast<uint32_t *>(iq), n);
    return hipGetLastError();
}

hipError_t launch_synth(const SynthRun *runs, uint64_t num_runs, uint64_t seed, uint32_t noise,
                        uint64_t first, uint64_t count, int16_t *iq, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, runs, num_runs, seed,
                       noise, first, count, reinterpret_cast<uint32_t *>(iq));
    return hipGetLastError();
}

}  // namespace ookd
