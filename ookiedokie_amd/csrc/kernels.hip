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
// One workgroup = 256 lanes = 4096 consecutive outputs; lane t owns outputs
// 16t..16t+15 and keeps their 32 partial sums in registers.  The tile's
// inputs plus the (padded) tap history are unpacked ONCE into LDS as float2.
// Taps are consumed in chunks of 32 held in SGPRs; within a chunk the lane
// streams 47 consecutive LDS samples, newest first, each feeding up to 16
// accumulators -- so every output sees its taps in the reference order
// (tap 0 / newest sample first, fir.c:313-318).
//
// LDS layout: sample j lives at slot j + (j >> 4): one pad slot per 16
// samples makes the lane stride 17 float2 = 34 dwords, which is conflict
// free for ds_read_b64 (32-lane halves, 64 banks), and keeps every read of
// the unrolled body at  lane_base + compile-time immediate.

__device__ __forceinline__ uint32_t slot(uint32_t j) { return j + (j >> 4); }

// Sequential, unfused recomputation of one output (guard-band path).
__device__ __noinline__ float2 fir1_exact_output(const float2 *lds, uint32_t j_out,
                                                 const float *taps, uint32_t ntaps) {
    float re = 0.0f, im = 0.0f;
    for (uint32_t k = 0; k < ntaps; ++k) {
        const float2 x = lds[slot(j_out - k)];
        const float t = taps[k];
        const float pr = t * x.x;
        const float pi = t * x.y;
        re = re + pr;
        im = im + pi;
    }
    return make_float2(re, im);
}

template <bool EXACT>
__global__ __launch_bounds__(kFirThreads) void fir1_bits_kernel(const FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2 *lds = reinterpret_cast<float2 *>(smem_raw);

    constexpr int R = kFirR;
    const uint32_t tid = threadIdx.x;
    const uint32_t cap = blockIdx.y;
    const uint64_t t0 = (uint64_t)blockIdx.x * kFirTile;
    const uint32_t Tp = p.stage[0].ntaps_pad;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;

    // ---- stage the window: slot j <-> input index t0 - Tp + j ----------------
    const uint32_t nvec = (kFirTile + Tp) >> 2;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    for (uint32_t v = tid; v < nvec; v += kFirThreads) {
        const int64_t n = (int64_t)t0 - (int64_t)Tp + 4 * (int64_t)v;
        float2 s0, s1, s2, s3;
        if (aligned16 && n >= 0 && (uint64_t)(n + 3) < p.n_valid) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + n);
            s0 = unpack_iq(q.x);
            s1 = unpack_iq(q.y);
            s2 = unpack_iq(q.z);
            s3 = unpack_iq(q.w);
        } else {
            s0 = fetch_sample(p, src, nullptr, n);
            s1 = fetch_sample(p, src, nullptr, n + 1);
            s2 = fetch_sample(p, src, nullptr, n + 2);
            s3 = fetch_sample(p, src, nullptr, n + 3);
        }
        float2 *dst = lds + slot(4 * v);        // 4 slots, never straddle a pad
        dst[0] = s0;
        dst[1] = s1;
        dst[2] = s2;
        dst[3] = s3;
    }
    __syncthreads();

    // ---- accumulate ----------------------------------------------------------
    float acc_re[R], acc_im[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc_re[r] = 0.0f;
        acc_im[r] = 0.0f;
    }

    const uint32_t nchunks = Tp / kTapChunk;
    for (uint32_t c = 0; c < nchunks; ++c) {
        const float *tp = p.taps + c * kTapChunk;       // wave-uniform => SGPRs
        float tap[kTapChunk];
#pragma unroll
        for (int i = 0; i < kTapChunk; ++i) tap[i] = tp[i];

        // output r of this lane sits at window slot-index Tp + 16*tid + r; tap
        // kc+kk reads Tp + 16*tid + r - kc - kk = 16*tid + 32*m + (w + 32),
        // w = r - kk, m = (Tp - kc - 32)/32.
        const uint32_t m = nchunks - 1 - c;
        const float2 *base = lds + 17u * tid + 34u * m;
#pragma unroll
        for (int w = R - 1; w >= -(kTapChunk - 1); --w) {
            const int cp = w + kTapChunk;               // 1 .. 47
            const float2 x = base[cp + (cp >> 4)];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int kk = r - w;
                if (kk >= 0 && kk < kTapChunk) {
                    if (EXACT) {
                        const float pr = tap[kk] * x.x;
                        const float pi = tap[kk] * x.y;
                        acc_re[r] = acc_re[r] + pr;
                        acc_im[r] = acc_im[r] + pi;
                    } else {
                        acc_re[r] = __builtin_fmaf(tap[kk], x.x, acc_re[r]);
                        acc_im[r] = __builtin_fmaf(tap[kk], x.y, acc_im[r]);
                    }
                }
            }
        }
    }

    // ---- threshold, guard band, pack -----------------------------------------
    const uint64_t o0 = t0 + (uint64_t)tid * R;
    uint32_t mask = 0;
    if (EXACT) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float pw = power_ref(acc_re[r], acc_im[r]);
            mask |= (pw >= p.p_star ? 1u : 0u) << r;
        }
    } else {
        uint32_t unsure = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float pw = power_ref(acc_re[r], acc_im[r]);
            mask |= (pw >= p.p_hi ? 1u : 0u) << r;
            unsure |= ((pw >= p.p_lo && !(pw >= p.p_hi)) ? 1u : 0u) << r;
        }
        if (unsure) {
            uint32_t redo = 0;
            for (int r = 0; r < R; ++r) {
                if ((unsure >> r) & 1u) {
                    if (o0 + r < p.n_out) {
                        const float2 y = fir1_exact_output(lds, Tp + R * tid + r, p.taps,
                                                           p.stage[0].ntaps);
                        const float pw = power_ref(y.x, y.y);
                        mask |= (pw >= p.p_star ? 1u : 0u) << r;
                        redo++;
                    }
                }
            }
            if (redo && p.recompute_count) atomicAdd(p.recompute_count, (unsigned long long)redo);
        }
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
            if (o0 + r < p.n_out) out[o0 + r] = make_float2(acc_re[r], acc_im[r]);
        }
    }

    // four lanes x 16 bits -> one 64-bit word
    const uint32_t pair = mask | (__shfl_xor(mask, 1) << 16);      // valid on even lanes
    const uint32_t hi = __shfl_xor(pair, 2);                       // lane+2's pair
    if ((tid & 3u) == 0) {
        uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
        words[(t0 >> 6) + (tid >> 2)] = (uint64_t)pair | ((uint64_t)hi << 32);
    }
}

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
__global__ __launch_bounds__(256) void nofir_bits_kernel(const FrontParams p) {
    const uint32_t cap = blockIdx.y;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride;
    uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    float2 *fout = p.fir_out ? reinterpret_cast<float2 *>(p.fir_out) + (uint64_t)cap * p.n_out : nullptr;
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    const uint64_t nquads = (p.words_per_cap * 64) >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // every lane of a wave runs the same number of iterations (nquads is a
    // multiple of 64 because words_per_cap is a multiple of 64 words)
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nquads; v += stride) {
        const uint64_t n = 4 * v;
        float2 s[4];
        if (aligned16 && n + 3 < p.n_valid) {
            const uint4 q = *reinterpret_cast<const uint4 *>(src + n);
            s[0] = unpack_iq(q.x);
            s[1] = unpack_iq(q.y);
            s[2] = unpack_iq(q.z);
            s[3] = unpack_iq(q.w);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = fetch_sample(p, src, nullptr, (int64_t)(n + i));
        }
        uint32_t nib = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = n + i < p.n_out;
            nib |= ((valid && power_ref(s[i].x, s[i].y) >= p.p_star) ? 1u : 0u) << i;
            if (fout && valid) fout[n + i] = s[i];
        }
        const uint32_t l = lane_id();
        uint32_t half = nib << (4u * (l & 7u));
        half |= __shfl_xor(half, 1);
        half |= __shfl_xor(half, 2);
        half |= __shfl_xor(half, 4);            // lanes 8g..8g+7 hold samples 32g..32g+31
        const uint32_t hi = __shfl_down(half, 8);
        if ((l & 15u) == 0) {
            words[(n >> 6)] = (uint64_t)half | ((uint64_t)hi << 32);
        }
    }
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
// edges: count per 4096-bit block, exclusive scan, compact
// ---------------------------------------------------------------------------

// word of level changes: bit b set <=> sample 64*w+b differs from its
// predecessor (the sample before a capture counts as 0, as record_dig's
// first line does for sample 0, ookiedokie.c:150-153).
__device__ __forceinline__ uint64_t change_word(const uint64_t *words, uint64_t w) {
    const uint64_t cur = words[w];
    const uint64_t prev_top = (w == 0) ? 0ull : (words[w - 1] >> 63);
    return cur ^ ((cur << 1) | prev_top);
}

__global__ __launch_bounds__(256) void edge_count_kernel(const EdgeParams p) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (wave >= total_blocks) return;
    const uint32_t cap = wave / p.blocks_per_cap;
    const uint32_t blk = wave % p.blocks_per_cap;
    const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    const uint64_t e = change_word(words, (uint64_t)blk * kBlockWords + lane_id());
    uint32_t c = (uint32_t)__popcll(e);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane_id() == 0) p.blk_count[wave] = c;
}

// single workgroup, 1024 lanes: exclusive scan of blk_count into blk_offset[0..n]
__global__ __launch_bounds__(1024) void edge_scan_kernel(const EdgeParams p) {
    __shared__ uint32_t part[1024];
    const uint32_t n = p.num_captures * p.blocks_per_cap;
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (n + 1023u) / 1024u;
    const uint32_t lo = min(tid * chunk, n);
    const uint32_t hi = min(lo + chunk, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += p.blk_count[i];
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = (tid >= d) ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - sum;     // exclusive prefix of this lane's chunk
    for (uint32_t i = lo; i < hi; ++i) {
        p.blk_offset[i] = run;
        run += p.blk_count[i];
    }
    if (tid == 1023) {
        p.blk_offset[n] = part[1023];
        if ((uint64_t)part[1023] > p.edge_capacity) *p.overflow = 1;
    }
}

__global__ __launch_bounds__(256) void edge_write_kernel(const EdgeParams p) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (wave >= total_blocks) return;
    if (p.blk_count[wave] == 0) return;
    const uint32_t cap = wave / p.blocks_per_cap;
    const uint32_t blk = wave % p.blocks_per_cap;
    const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    const uint64_t w = (uint64_t)blk * kBlockWords + lane_id();
    uint64_t e = change_word(words, w);
    const uint32_t c = (uint32_t)__popcll(e);
    // exclusive prefix over the wave
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(inc, d);
        if ((int)lane_id() >= d) inc += v;
    }
    uint64_t at = (uint64_t)p.blk_offset[wave] + (inc - c);
    while (e) {
        const int b = __ffsll((long long)e) - 1;
        if (at < p.edge_capacity) p.edges[at] = w * 64 + (uint64_t)b;
        ++at;
        e &= e - 1;
    }
}

// ---------------------------------------------------------------------------
// symbol state machine over the edge list
// ---------------------------------------------------------------------------
//
// Integer restatement of handle_rx_triggers / process (state_machine.c:
// 421-539).  `k` = number of elapsed_us increments since it was last zeroed;
// the host turned every duration window / timeout into a range of k by
// replaying the reference's double accumulation, so the tests below are the
// reference's float comparisons exactly.

enum { kCondAlways = 1, kCondPulseStart, kCondPulseEnd, kCondTimeout, kCondMsgComplete };
enum { kActNone = 1, kActAppend0, kActAppend1, kActOutput };
enum { kResError = -1, kResNone = 0, kResOutput = 1 };

struct Fsm {
    const FsmTablesDev *t;      // LDS copy
    uint32_t cur, nbits, prev;
    uint64_t k;
    uint64_t data[kPayloadWords];
};

__device__ __forceinline__ bool in_range(uint64_t k, uint64_t lo, uint64_t hi) {
    return k >= lo && k <= hi;
}

// state_machine.c:421-519
__device__ int fsm_eval(Fsm &f, uint32_t b) {
    const FsmTablesDev *t = f.t;
    const uint32_t s = f.cur;
    int fired = -1;
    bool edge = false;
    for (uint32_t i = t->trig_begin[s]; i < t->trig_begin[s + 1] && fired < 0; ++i) {
        if (!in_range(f.k, t->trig_kmin[i], t->trig_kmax[i])) continue;    // :119-133
        switch (t->trig_cond[i]) {
        case kCondAlways:
            fired = (int)i;
            break;
        case kCondPulseStart:
            if (!f.prev && b) {
                fired = (int)i;
                edge = true;
            }
            break;
        case kCondPulseEnd:
            if (f.prev && !b) {
                fired = (int)i;
                edge = true;
            }
            break;
        case kCondTimeout:
            if (f.k >= t->state_kto[s]) fired = (int)i;     // kto = MAX when no timeout
            break;
        case kCondMsgComplete:
            if (f.nbits >= t->max_bits) fired = (int)i;
            break;
        default:
            break;
        }
    }
    if (fired < 0) {
        f.k += 1;                                           // :513-515
        return kResNone;
    }
    int result = kResNone;
    const bool ok = !edge || in_range(f.k, t->state_kmin[s], t->state_kmax[s]);   // :100-117
    if (ok) {
        const uint32_t act = t->trig_action[fired];
        if (act == kActAppend0 || act == kActAppend1) {
            // :365-385 stores while num_bits <= max_bits, always counts
            if (f.nbits <= t->max_bits) {
                const uint32_t wi = f.nbits >> 6;
                const uint64_t m = 1ull << (f.nbits & 63u);
                if (wi < kPayloadWords) {
                    if (act == kActAppend1) f.data[wi] |= m;
                    else f.data[wi] &= ~m;
                }
            }
            f.nbits += 1;
        } else if (act == kActOutput) {
            result = kResOutput;
        }
        f.cur = t->trig_next[fired];
    } else {
        result = kResError;
        f.cur = 0;                                          // :505-509
    }
    f.k = 0;                                                // :511
    return result;
}

// state_machine.c:521-539: reset clears the payload and is evaluated, then
// the (possibly new) state is evaluated on the same sample.
__device__ int fsm_step(Fsm &f, uint32_t b) {
    if (f.cur == 0) {
        f.nbits = 0;
        // memset(data, 0, (max_bits+7)/8); the spare word past it is never output
        for (int i = 0; i < kPayloadWords; ++i) f.data[i] = 0;
        const int r = fsm_eval(f, b);
        if (r != kResNone) return r;
    }
    return fsm_eval(f, b);
}

// Number of upcoming trigger EVALUATIONS (at k, k+1, ...) that certainly do
// not fire while the input level stays constant (b == prev, so pulse
// triggers cannot fire).  UINT64_MAX = never.
__device__ uint64_t fsm_quiet_evals(const Fsm &f) {
    const FsmTablesDev *t = f.t;
    const uint32_t s = f.cur;
    uint64_t best = ~0ull;
    for (uint32_t i = t->trig_begin[s]; i < t->trig_begin[s + 1]; ++i) {
        uint64_t lo = t->trig_kmin[i];
        const uint64_t hi = t->trig_kmax[i];
        const uint32_t c = t->trig_cond[i];
        if (c == kCondAlways) {
        } else if (c == kCondTimeout) {
            const uint64_t kto = t->state_kto[s];
            if (kto == ~0ull) continue;
            if (kto > lo) lo = kto;
        } else if (c == kCondMsgComplete) {
            if (f.nbits < t->max_bits) continue;
        } else {
            continue;
        }
        const uint64_t first = f.k > lo ? f.k : lo;         // first k >= max(k, lo)
        if (first > hi) continue;
        const uint64_t wait = first - f.k;
        if (wait < best) best = wait;
    }
    return best;
}

// First decimated index of buffer `buf`: floor(buf * spb / D)
// (decimated sample j comes from input D*(j+1)-1).
__device__ __forceinline__ uint64_t buffer_start(uint64_t buf, uint32_t spb, uint32_t D) {
    return (buf * (uint64_t)spb) / D;      // buf*spb ~ input samples, fits 64 bits
}

constexpr int kEdgeWin = 256;

__global__ __launch_bounds__(64) void fsm_prepare_kernel(const FsmParams p, const FsmStateDev first,
                                                         int have_first) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (seg >= nseg) return;
    const uint32_t cap = seg / p.segs_per_cap;
    const uint32_t ls = seg % p.segs_per_cap;
    FsmStateDev st;
    st.cur = 0;
    st.nbits = 0;
    st.k = 0;
    st.prev = 0;
    st.pad = 0;
    for (int i = 0; i < kPayloadWords; ++i) st.data[i] = 0;
    if (ls == 0) {
        if (have_first) st = first;
    } else {
        // speculative: reset state, previous bit = the sample before the segment
        const uint64_t start = buffer_start((uint64_t)ls * p.seg_buffers, p.spb, p.total_decim);
        if (start > 0 && start <= p.n_out) {
            const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
            st.prev = (uint32_t)((words[(start - 1) >> 6] >> ((start - 1) & 63)) & 1ull);
        }
    }
    p.state_in[seg] = st;
    p.seg_msg_count[seg] = 0;
    p.seg_err_count[seg] = 0;
}

__device__ __forceinline__ bool state_equal(const FsmStateDev &a, const FsmStateDev &b) {
    bool eq = a.cur == b.cur && a.nbits == b.nbits && a.k == b.k && a.prev == b.prev;
    for (int i = 0; i < kPayloadWords; ++i) eq = eq && a.data[i] == b.data[i];
    return eq;
}

// One wave per segment; lane 0 walks the state machine, all lanes fetch
// edges cooperatively into an LDS window.
// mode 0: run every segment (fresh run).  mode 1: rerun only segments whose
// incoming state (the previous segment's outgoing state of the last round)
// changed.  mode 2: as 1, and a capture's first segment reruns too (its
// incoming state was just replaced by the host: shard refine).
__global__ __launch_bounds__(64) void fsm_run_kernel(const FsmParams p, uint32_t parity, uint32_t mode,
                                                     uint32_t slot) {
    __shared__ FsmTablesDev tab;
    __shared__ uint64_t win[kEdgeWin];
    __shared__ uint64_t sh_ci;
    __shared__ uint64_t sh_pos;
    __shared__ int sh_done;

    const uint32_t seg = blockIdx.x;
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    const uint32_t cap = seg / p.segs_per_cap;
    const uint32_t ls = seg % p.segs_per_cap;
    const uint32_t lane = threadIdx.x;
    const uint32_t par = parity & 1u;
    FsmStateDev *out_cur = p.state_out + (size_t)par * nseg;
    const FsmStateDev *out_prev = p.state_out + (size_t)(par ^ 1u) * nseg;

    // ---- does this segment need to run? --------------------------------------
    if (mode != 0) {
        bool rerun = false;
        if (ls != 0) {
            const FsmStateDev nin = out_prev[seg - 1];
            rerun = !state_equal(nin, p.state_in[seg]);
            if (rerun && lane == 0) p.state_in[seg] = nin;
        } else {
            rerun = (mode == 2);
        }
        if (!rerun) {
            if (lane == 0) out_cur[seg] = out_prev[seg];
            return;
        }
        if (lane == 0) atomicAdd(&p.changed[slot], 1u);
        __syncthreads();
    }

    // ---- tables to LDS -----------------------------------------------------------
    {
        const uint32_t *srcw = reinterpret_cast<const uint32_t *>(p.tables);
        uint32_t *dstw = reinterpret_cast<uint32_t *>(&tab);
        for (uint32_t i = lane; i < sizeof(FsmTablesDev) / 4; i += 64) dstw[i] = srcw[i];
    }

    const uint64_t seg_start = min(buffer_start((uint64_t)ls * p.seg_buffers, p.spb, p.total_decim), p.n_out);
    const uint64_t seg_end = (ls + 1 == p.segs_per_cap)
                                 ? p.n_out
                                 : min(buffer_start((uint64_t)(ls + 1) * p.seg_buffers, p.spb, p.total_decim),
                                       p.n_out);
    const uint32_t blk0 = cap * p.blocks_per_cap;
    const uint64_t cap_e0 = p.blk_offset[blk0];
    const uint64_t ne = (uint64_t)p.blk_offset[blk0 + p.blocks_per_cap] - cap_e0;   // edges of this capture
    const uint64_t *edges = p.edges + cap_e0;

    // ---- first edge at or after seg_start (capture-local index) ------------------
    uint64_t ci;
    {
        const uint64_t blk = min(seg_start >> 12, (uint64_t)p.blocks_per_cap - 1);
        uint64_t i = (uint64_t)p.blk_offset[blk0 + blk] - cap_e0;
        const uint64_t iend = (uint64_t)p.blk_offset[blk0 + blk + 1] - cap_e0;
        // count edges of that block below seg_start, 64 at a time
        uint64_t below = 0;
        for (; i < iend; i += 64) {
            const uint64_t idx = i + lane;
            const bool lt = idx < iend && edges[idx] < seg_start;
            below += (uint64_t)__popcll(__ballot(lt));
        }
        ci = ((uint64_t)p.blk_offset[blk0 + blk] - cap_e0) + below;
    }
    __syncthreads();

    Fsm f;
    uint64_t pos = seg_start;
    uint32_t nmsg = 0, nerr = 0;
    uint32_t flags = 0;
    if (lane == 0) {
        const FsmStateDev st = p.state_in[seg];
        f.t = &tab;
        f.cur = st.cur;
        f.nbits = st.nbits;
        f.prev = st.prev;
        f.k = st.k;
        for (int i = 0; i < kPayloadWords; ++i) f.data[i] = st.data[i];
        sh_ci = ci;
        sh_done = (pos >= seg_end) ? 1 : 0;
    }
    __syncthreads();

    MsgDev *msgs = p.seg_msgs + (size_t)seg * p.msg_slots;
    uint64_t *errs = p.seg_errs + (size_t)seg * p.err_slots;

    while (!sh_done) {
        // refill the window with edges[win_base .. win_base + kEdgeWin)
        const uint64_t win_base = sh_ci;
        for (uint32_t q = lane; q < kEdgeWin; q += 64) {
            const uint64_t idx = win_base + q;
            win[q] = idx < ne ? edges[idx] : ~0ull;
        }
        __syncthreads();
        if (lane == 0) {
            ci = win_base;
            // needs win[ci - win_base] and win[ci + 1 - win_base]
            while (pos < seg_end && ci + 1 < win_base + kEdgeWin) {
                const uint64_t e0 = win[ci - win_base];
                const bool at_edge = (e0 == pos);
                const uint64_t cia = ci + (at_edge ? 1 : 0);
                const uint32_t b = (uint32_t)(cia & 1ull);
                if (b == f.prev) {
                    // constant input: skip samples that cannot fire a trigger
                    const uint64_t e1 = win[cia - win_base];
                    const uint64_t run_end = e1 < seg_end ? e1 : seg_end;
                    const uint64_t n = run_end - pos;
                    const uint64_t quiet = fsm_quiet_evals(f);
                    uint64_t m;
                    if (f.cur == 0) {
                        m = quiet >> 1;                 // two evaluations per sample in reset
                        if (m > n) m = n;
                        if (m > 0) f.k += 2 * m;
                    } else {
                        m = quiet < n ? quiet : n;
                        if (m > 0) f.k += m;
                    }
                    if (m > 0) {
                        pos += m;
                        ci = cia;
                        continue;
                    }
                }
                const int r = fsm_step(f, b);
                f.prev = b;                             // sm_process: prev_bit = data[i]
                if (r == kResOutput) {
                    if (nmsg < p.msg_slots) {
                        MsgDev mm;
                        mm.capture = cap;
                        mm.reserved = 0;
                        mm.sample = pos;
                        // the first (max_bits+7)/8 bytes are the message
                        const uint32_t nbytes = (tab.max_bits + 7u) >> 3;
                        for (uint32_t i = 0; i < 4; ++i) {
                            uint64_t v = f.data[i];
                            if (8 * i >= nbytes) v = 0;
                            else if (8 * (i + 1) > nbytes) v &= (1ull << ((nbytes - 8 * i) * 8)) - 1ull;
                            mm.payload[i] = v;
                        }
                        msgs[nmsg] = mm;
                    } else {
                        flags |= 1u;
                    }
                    nmsg++;
                    pos += 1;
                    ci = cia;
                } else if (r == kResError) {
                    if (nerr < p.err_slots) errs[nerr] = pos;
                    nerr++;
                    // device.c:646: the rest of this buffer is never fed in
                    const uint64_t in_idx = (uint64_t)p.total_decim * (pos + 1) - 1;
                    const uint64_t buf = in_idx / p.spb;
                    const uint64_t nb = buffer_start(buf + 1, p.spb, p.total_decim);
                    pos = nb > pos ? nb : pos + 1;
                    ci = cia;
                    // first edge at or after pos
                    while (ci + 1 < win_base + kEdgeWin && win[ci - win_base] < pos) ci++;
                    if (win[ci - win_base] < pos) break;    // ran off the window: refill
                } else {
                    pos += 1;
                    ci = cia;
                }
            }
            // after a window break caused by the skip loop, make sure ci is exact
            sh_ci = ci;
            sh_done = (pos >= seg_end) ? 1 : 0;
        }
        __syncthreads();
        // an error skip can leave win[ci] < pos with the window exhausted: advance
        // through the list cooperatively until the first edge >= pos
        if (!sh_done) {
            if (lane == 0) sh_pos = pos;
            __syncthreads();
            const uint64_t target = sh_pos;
            uint64_t c2 = sh_ci;
            for (;;) {
                const uint64_t idx = c2 + lane;
                const bool lt = idx < ne && edges[idx] < target;
                const uint64_t bal = __ballot(lt);
                c2 += (uint64_t)__popcll(bal);
                if (bal != ~0ull) break;
            }
            __syncthreads();
            if (lane == 0) sh_ci = c2;
            __syncthreads();
        }
    }

    if (lane == 0) {
        FsmStateDev st;
        st.cur = f.cur;
        st.nbits = f.nbits;
        st.k = f.k;
        st.prev = f.prev;
        st.pad = 0;
        for (int i = 0; i < kPayloadWords; ++i) st.data[i] = f.data[i];
        out_cur[seg] = st;
        p.seg_msg_count[seg] = nmsg;
        p.seg_err_count[seg] = nerr;
        if (flags) atomicOr(p.flags, flags);
    }
}

// Compacts per-segment messages into one list (single workgroup).
__global__ __launch_bounds__(1024) void fsm_gather_kernel(const FsmParams p) {
    __shared__ uint32_t part[1024];
    __shared__ unsigned long long err_total;
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (nseg + 1023u) / 1024u;
    const uint32_t lo = min(tid * chunk, nseg);
    const uint32_t hi = min(lo + chunk, nseg);
    if (tid == 0) err_total = 0;
    __syncthreads();
    uint32_t sum = 0;
    unsigned long long esum = 0;
    for (uint32_t s = lo; s < hi; ++s) {
        sum += min(p.seg_msg_count[s], p.msg_slots);
        esum += p.seg_err_count[s];
    }
    if (esum) atomicAdd(&err_total, esum);
    part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = (tid >= d) ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint64_t at = part[tid] - sum;
    for (uint32_t s = lo; s < hi; ++s) {
        const uint32_t c = min(p.seg_msg_count[s], p.msg_slots);
        for (uint32_t i = 0; i < c; ++i) {
            if (at < p.msg_capacity) p.msgs[at] = p.seg_msgs[(size_t)s * p.msg_slots + i];
            ++at;
        }
    }
    if (tid == 1023) {
        p.totals[0] = part[1023];
        p.totals[1] = err_total;
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

static size_t fir1_lds_bytes(uint32_t Tp) {
    const uint32_t n = kFirTile + Tp;
    return (size_t)(n + (n >> 4) + 1) * sizeof(float2);
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
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fir_generic_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_generic_kernel<true>, grid, dim3(256), lds, stream, p, even + 1);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fir_generic_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fir_generic_kernel<false>, grid, dim3(256), lds, stream, p, even + 1);
    }
    return hipGetLastError();
}

hipError_t launch_front(const FrontParams &p, uint32_t num_captures, bool exact, hipStream_t stream) {
    if (p.n_out == 0) return hipSuccess;
    if (p.num_stages == 0) {
        const uint64_t nquads = (p.words_per_cap * 64) >> 2;
        uint64_t blocks = (nquads + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(nofir_bits_kernel, dim3((uint32_t)blocks, num_captures), dim3(256), 0, stream, p);
        return hipGetLastError();
    }
    if (p.num_stages == 1 && p.stage[0].decim == 1 && p.origin == 0 && !p.iq_f32) {
        const size_t lds = fir1_lds_bytes(p.stage[0].ntaps_pad);
        if (lds <= 160 * 1024) {
            const uint64_t tiles = (p.n_out + kFirTile - 1) / kFirTile;
            dim3 grid((uint32_t)tiles, num_captures);
            hipError_t e;
            if (exact) {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fir1_bits_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(fir1_bits_kernel<true>, grid, dim3(kFirThreads), lds, stream, p);
            } else {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(&fir1_bits_kernel<false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(fir1_bits_kernel<false>, grid, dim3(kFirThreads), lds, stream, p);
            }
            return hipGetLastError();
        }
    }
    return launch_front_generic(p, num_captures, stream);
}

hipError_t launch_edges(const EdgeParams &p, hipStream_t stream) {
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (total_blocks == 0) return hipSuccess;
    const uint32_t wgs = (total_blocks + 3) / 4;        // 4 waves per workgroup
    hipLaunchKernelGGL(edge_count_kernel, dim3(wgs), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(edge_scan_kernel, dim3(1), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(edge_write_kernel, dim3(wgs), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_fsm_prepare(const FsmParams &p, const FsmStateDev *first_state, hipStream_t stream) {
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (nseg == 0) return hipSuccess;
    FsmStateDev first{};
    if (first_state) first = *first_state;
    hipLaunchKernelGGL(fsm_prepare_kernel, dim3((nseg + 63) / 64), dim3(64), 0, stream, p, first,
                       first_state ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_fsm_iteration(const FsmParams &p, uint32_t parity, uint32_t mode, uint32_t slot,
                                hipStream_t stream) {
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (nseg == 0) return hipSuccess;
    hipLaunchKernelGGL(fsm_run_kernel, dim3(nseg), dim3(64), 0, stream, p, parity, mode, slot);
    return hipGetLastError();
}

hipError_t launch_fsm_gather(const FsmParams &p, uint32_t, hipStream_t stream) {
    hipLaunchKernelGGL(fsm_gather_kernel, dim3(1), dim3(1024), 0, stream, p);
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
