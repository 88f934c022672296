// fir_mfma.hip -- the 1-stage / decimation-1 front end on the matrix cores.
//
//   SC16Q11 unpack -> FIR -> |.|^2 >= P* -> packed bit words
//   (reference: src/complexf.h:68-77, src/fir.c:302-334, src/ookiedokie.c:171-179)
//
// Why: the direct form of fir.c:313-318 is fp32 bound on this part -- 128 flop per
// 4-byte sample with 32 taps (above the 19.7 flop/B ridge), 1020 with 255 taps -- and
// the packed-VALU kernel (kernels.hip) tops out at half of the fp32 peak.  The matrix
// cores run fp16 products with fp32 accumulation at 16 x the fp32 rate, and for THIS
// filter the products can be made exact:
//   * the samples are integers.  A nominal SC16Q11 sample (|x| <= 2048) is one fp16
//     number exactly; any int16 splits exactly into x & ~31 (a multiple of 32, 11
//     significant bits) + x & 31 -- two fp16 numbers;
//   * a tap scaled by a power of two S splits into two fp16 pieces h1 + h2 that carry
//     22 of its 24 significant bits -- 11 + 11 bits -- and usually all the filter needs:
//     the host computes what is left over, exactly, and puts it into the error bound;
//   * an 11-bit x 11-bit product is exact in fp32, so  y = c * sum (h1 + h2) * X  differs
//     from the real sum only by the fp32 roundings of the accumulation -- like the
//     reference's own sequential sum does (in another order).
// The contract is the fused kernel's: the power is compared against a GUARD BAND
// [p_lo, p_hi) around P* derived from a forward bound on |y_mfma - y_ref|, and a sample
// that lands inside is recomputed in the reference's exact order (tap 0 first, separately
// rounded multiply and add).  Bits are the reference's by construction; floats within
// 1e-5 of the scale sum|h| max|x| (measured: a few 1e-7).
//
// The filter as a matrix product: a wave tile is 1024 outputs = 32 columns of 32
// consecutive outputs.  With window sample j <-> input index t0 - Tp + j,
//   y[32 n + i] = sum_kk A[i][kk] * win[32 n + kk],  A[i][kk] = h[i + Tp - kk]  (0 outside 0..T-1),
// kk = 0 .. Tp + 31: a 32 x (Tp + 32) Toeplitz matrix of taps -- the SAME for every column
// and every tile -- times a (Tp + 32) x 32 matrix whose column n is the window shifted by
// 32 n.  v_mfma_f32_32x32x16_f16 takes 16 of the kk per instruction: KS = Tp / 16 + 2
// K-steps, x 2 tap pieces x {re, im} (x 2 sample pieces for a tile that holds a sample
// beyond +-2048).  The Toeplitz matrix holds T taps in Tp + 32 columns: 50 % of the
// products are structural zeros with 32 taps, 11 % with 255 -- against a 16-fold rate.
//   A fragments: constants, built by the host (mfma_prepare_taps), 4 VGPRs per K-step
//                and piece, loaded once per workgroup and kept in registers over the
//                workgroup's G consecutive tiles;
//   B fragments: lane (n, half) reads the 8 consecutive window samples 32 n + 16 s +
//                8 half of one plane (re or im) as fp16 from the wave's LDS window --
//                one ds_read_b128; 8 pad halfs per 32 samples make the 16 lanes of a
//                read group hit 16 different bank quads;
//   C:           column on the lane, rows in the 16 registers: re and im of an output
//                sit in the same lane and register, so power, threshold and guard band
//                are per-lane VALU work, and a column's 32 bits are one lane pair's.
// Everything around the product is the packed-VALU kernel's: raw non-temporal 16 B
// loads, the quiet shortcut on the raw samples, sparse output with run stamps, tile
// infos for the edge stage, no workgroup barrier (one wave = one workgroup).
#include "kernels.hpp"
#include "common.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#pragma clang fp contract(off)

namespace ookd {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float v2fm __attribute__((ext_vector_type(2)));
typedef short v2s __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// half index of window sample j inside a plane: 8 pad halfs per 32 samples
__host__ __device__ constexpr uint32_t mslot(uint32_t j) { return j + 8u * (j >> 5); }

template <int KS>
struct MfmaGeom {
    static constexpr uint32_t Tp = 16u * (KS - 2);              // tap history the window holds
    static constexpr uint32_t W = kMfmaTile + Tp;               // window samples (a multiple of 32)
    static constexpr uint32_t plane = mslot(W);                 // halfs per plane (a multiple of 8)
    static constexpr uint32_t lds_bytes = plane * 2u * 2u;      // re + im
    static constexpr uint32_t nvec = W / 4u;                    // 16 B raw vectors
    static constexpr int rounds = (int)((nvec + 63u) / 64u);
};

typedef const __attribute__((address_space(1))) v4u *gptr128;
__device__ __forceinline__ uint4 ld_nt4m(gptr128 p) {
    const v4u v = __builtin_nontemporal_load(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// uniform base + 32-bit byte offset: lets the compiler use the SGPR-base addressing form instead of
// keeping a 64-bit address per lane in registers (and hoisting a dozen of them out of the tile loop)
typedef const __attribute__((address_space(1))) unsigned char *gbytes;
typedef __attribute__((address_space(1))) unsigned char *gbytes_w;
__device__ __forceinline__ uint4 ld_nt4_at(gbytes base, uint32_t byte_off) {
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<gptr128>(base + byte_off));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Keeps a uniform pointer in SGPRs and opaque: without it the compiler adds every lane's offset to it as a
// 64-bit VGPR pair OUTSIDE the tile loop (a dozen pairs), runs out of registers and reloads them from scratch
// in the loop head -- in front of the loads everything waits for.
template <typename P>
__device__ __forceinline__ P uniform_ptr(P ptr) {
    uint64_t v = (uint64_t)ptr;
    asm volatile("" : "+s"(v));
    return (P)v;
}

__device__ __forceinline__ v2s as_v2sm(uint32_t w) { return __builtin_bit_cast(v2s, w); }

typedef const __attribute__((address_space(1))) uint32_t *gptr32;
typedef const __attribute__((address_space(1))) float *gptrf;

// what the boundary handling needs, as scalars (a reference to the kernel's parameter block would put
// the whole block on the stack)
struct RawSrc {
    gptr32 src;             // the capture
    gptr32 halo;            // samples in front of it (newest last) or null
    uint32_t halo_len;
    uint64_t n_valid;       // samples present; beyond: zeros (bladeRF_file.c:113-117)
};

__device__ __forceinline__ uint32_t fetch_raw_m(const RawSrc &rs, int64_t n) {
    if (n < 0) {
        const int64_t h = (int64_t)rs.halo_len + n;
        if (h < 0 || !rs.halo) return 0u;
        return rs.halo[h];
    }
    if ((uint64_t)n >= rs.n_valid) return 0u;
    return rs.src[n];
}

// Reference-order recomputation of output `n` (= input index: one stage, decimation 1)
// from the capture itself: fir.c:313-318, separately rounded multiply and add.
__device__ __forceinline__ float2 mfma_exact_output(const RawSrc &rs, gptrf taps, uint32_t T, int64_t n) {
    const float s = 1.0f / 2048.0f;
    float re = 0.0f, im = 0.0f;
    for (uint32_t k = 0; k < T; ++k) {
        const uint32_t w = fetch_raw_m(rs, n - (int64_t)k);
        const float xr = (float)(int16_t)(w & 0xffffu) * s;
        const float xi = (float)(int16_t)(w >> 16) * s;
        const float t = taps[k];
        const float pr = t * xr;
        const float pi = t * xi;
        re = re + pr;
        im = im + pi;
    }
    return make_float2(re, im);
}

// int16 -> fp16 of one half of each of two dwords, packed: one SDWA conversion per output half (the second
// writes the upper half and preserves the lower), no separate pack -- 8 instructions per 4 samples.
template <int HALF>
__device__ __forceinline__ uint32_t cvt2(uint32_t a, uint32_t b) {
    uint32_t r;
    if (HALF == 0) {
        asm("v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(r) : "v"(a));
        asm("v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(r) : "v"(b));
    } else {
        asm("v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(r) : "v"(a));
        asm("v_cvt_f16_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(r) : "v"(b));
    }
    return r;
}

// four raw samples (I | Q << 16 each), masked, -> four fp16 I and four fp16 Q (exact: see top)
__device__ __forceinline__ void cvt4(uint4 q, uint32_t mask, uint2 &re, uint2 &im) {
    const uint32_t w0 = q.x & mask, w1 = q.y & mask, w2 = q.z & mask, w3 = q.w & mask;
    re = make_uint2(cvt2<0>(w0, w1), cvt2<0>(w2, w3));
    im = make_uint2(cvt2<1>(w0, w1), cvt2<1>(w2, w3));
}

// Workgroup = kMfmaWaves wavefronts that share ONE copy of the A-fragment image in LDS (8 KB with 32
// taps, 36 KB with 255) and otherwise work alone: each wave pulls tiles (FrontParams::mfma_g per wave)
// from a ticket in LDS, so a wave that drew loud tiles does not hold the others up, and a quiet tile costs
// nothing but its loads.  The image is fetched by the first wave that meets a loud tile (another wave
// doing the same at the same time stores the same bytes); the ready flag is set behind that wave's own
// stores -- the LDS executes a wave's accesses in order.  No workgroup barrier after the one that
// publishes the zeroed ticket.
//
// Ticket k of workgroup b is tile k * gridDim + b: the workgroups that are resident together (consecutive
// b, at about the same k) read one dense, moving window of the capture, like a hardware-dispatched grid
// of one-tile workgroups does.
//
// Software pipeline: the next tile's raw window is requested as soon as this tile's raw samples are dead
// -- at once for a quiet tile, behind the (last) conversion for a loud one -- into the same registers; it
// is in flight during the product and everything behind it.  (Anything that puts a second dependent
// memory round trip in front of those loads -- a register spilled to scratch and reloaded in the loop
// head, a ticket in global memory -- costs a whole loaded-memory latency per tile: measured 4.3 us per
// tile and wave instead of 1.7.)
//
// Registers (96: five waves per SIMD): 32 accumulators + two A / B fragment sets + the raw window.
// Waves per workgroup: as many as it takes to have 16 waves per CU beside ONE image per workgroup in the
// 160 KB of LDS -- 4 up to 64 taps (image 8 / 12 KB), 8 up to 128 (20 KB), 16 up to 256 (36 KB: one
// workgroup per CU; with four waves per workgroup only two workgroups fit, two waves per SIMD, and the
// matrix pipe idles while both convert or threshold: measured 41 % MFMA + 33 % VALU busy, hardly overlapping).
template <int KS>
__host__ __device__ constexpr int mfma_waves() { return KS <= 6 ? 4 : KS <= 10 ? 8 : 16; }
constexpr uint32_t kMfmaCtlBytes = 16;

template <int KS>
__host__ __device__ constexpr uint32_t mfma_lds_bytes() {
    return (uint32_t)KS * 2u * 1024u + kMfmaCtlBytes + (uint32_t)mfma_waves<KS>() * MfmaGeom<KS>::lds_bytes;
}

struct MfmaTileCtx {            // uniform per workgroup
    RawSrc rs;
    uint32_t *ctl;              // LDS: [0] next ticket, [1] image ready
    uint32_t tickets;           // per workgroup
    uint64_t tile_base, tile_end;
    uint32_t xcd_span;          // != 0: the launch's tiles in eight runs of this many, one per XCD (see mfma_take_ticket)
    bool aligned16;
};

// Ticket tk of workgroup b is position v = tk * gridDim + b of the launch: workgroups that run at the same time hold
// neighbouring positions.  Position -> tile: v itself -- or, xcd_span != 0 (the launch holds 8 * xcd_span tiles):
// tile (v & 7) * xcd_span + (v >> 3).  Workgroups are dealt out to the eight XCDs in turn, so v & 7 is the XCD: each
// XCD then goes through ONE contiguous eighth of the launch, and the halo a tile shares with the next one (96 of 1120
// samples for the decimate-by-4 filter) is read by the same XCD twice -- the second time from its L2 -- instead of by
// two XCDs from memory once each.
__device__ __forceinline__ bool mfma_take_ticket(const MfmaTileCtx &c, uint32_t tid, uint64_t &tile) {
    uint32_t tk = 0;
    if (tid == 0) tk = __hip_atomic_fetch_add(&c.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
    const uint64_t v = (uint64_t)tk * gridDim.x + blockIdx.x;
    tile = (c.xcd_span ? (v & 7u) * (uint64_t)c.xcd_span + (v >> 3) : v) + c.tile_base;
    return tk < c.tickets && v + c.tile_base < c.tile_end;
}

template <int KS>
__device__ __forceinline__ bool mfma_interior(const MfmaTileCtx &c, uint64_t tile) {
    const uint64_t t0 = tile * kMfmaTile;
    return c.aligned16 && t0 >= MfmaGeom<KS>::Tp && t0 + kMfmaTile <= c.rs.n_valid;
}

// vector v <-> input samples t0 - Tp + 4v .. + 3
template <int KS>
__device__ __forceinline__ void mfma_issue_loads(const MfmaTileCtx &c, uint64_t tile, uint32_t tid,
                                                 uint4 (&q)[MfmaGeom<KS>::rounds]) {
    using Gm = MfmaGeom<KS>;
    const gbytes src4 = uniform_ptr((gbytes)(c.rs.src + (tile * kMfmaTile - Gm::Tp)));
#pragma unroll
    for (int i = 0; i < Gm::rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        // (a partial last round: the lanes past the window read its last vector again -- same cache line,
        //  harmless to the min / max of the quiet test, never stored; a branch here would need its own address)
        q[i] = ld_nt4_at(src4, 16u * ((64u * (i + 1) <= Gm::nvec || v < Gm::nvec) ? v : Gm::nvec - 1u));
    }
}

// first / last tiles of a capture, halo of a shard, unaligned pointers: sample by sample, staged raw
// through the (free) LDS window so that the code stays a compact loop
template <int KS>
__device__ __noinline__ void mfma_boundary_stage(RawSrc rs, uint64_t t0, uint32_t tid, uint4 *stage) {
    using Gm = MfmaGeom<KS>;
    for (uint32_t v = tid; v < Gm::nvec; v += 64u) {
        const int64_t s0 = (int64_t)t0 - (int64_t)Gm::Tp + 4 * (int64_t)v;
        uint4 w;
        w.x = fetch_raw_m(rs, s0);
        w.y = fetch_raw_m(rs, s0 + 1);
        w.z = fetch_raw_m(rs, s0 + 2);
        w.w = fetch_raw_m(rs, s0 + 3);
        stage[v] = w;
    }
}

template <int KS>
__device__ __forceinline__ void mfma_boundary_loads(const MfmaTileCtx &c, uint64_t tile, uint32_t tid, unsigned char *win,
                                                    uint4 (&q)[MfmaGeom<KS>::rounds]) {
    using Gm = MfmaGeom<KS>;
    uint4 *stage = reinterpret_cast<uint4 *>(win);
    mfma_boundary_stage<KS>(c.rs, tile * kMfmaTile, tid, stage);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < Gm::rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        q[i] = stage[(64u * (i + 1) <= Gm::nvec || v < Gm::nvec) ? v : Gm::nvec - 1u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// raw window -> fp16 planes in LDS (masked: a wide tile's upper / lower bits)
template <int KS>
__device__ __forceinline__ void mfma_convert(const uint4 (&q)[MfmaGeom<KS>::rounds], uint32_t mask, uint32_t tid,
                                             _Float16 *pl_re, _Float16 *pl_im) {
    using Gm = MfmaGeom<KS>;
#pragma unroll
    for (int i = 0; i < Gm::rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        if (64u * (i + 1) <= Gm::nvec || v < Gm::nvec) {
            uint2 r4, i4;
            cvt4(q[i], mask, r4, i4);
            *reinterpret_cast<uint2 *>(pl_re + mslot(4u * v)) = r4;
            *reinterpret_cast<uint2 *>(pl_im + mslot(4u * v)) = i4;
        }
    }
    // the window is private to this wavefront and the LDS executes one wave's accesses in
    // order: no workgroup barrier, only keep the compiler from moving reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the K-steps over the planes; FIRST: the accumulators start from zero (the instruction's own C = 0).
// Two fragment sets: step s + 1 is read while step s multiplies; the scheduling barriers keep the compiler
// from hoisting every step's reads to the front (64 registers with 32 taps -- and then the raw window of
// the NEXT tile, in flight, is what gets spilled: a wait in front of the product).
template <int KS, bool FIRST>
__device__ __forceinline__ void mfma_ksteps(const h8 *a_img, const _Float16 *pl_re, const _Float16 *pl_im, uint32_t tid,
                                            f16x &are, f16x &aim) {
    const uint32_t n = tid & 31u, hh = tid >> 5;
    const _Float16 *bre = pl_re + 40u * n + 8u * hh;
    const _Float16 *bim = pl_im + 40u * n + 8u * hh;
    const h8 *af = a_img + tid;
    f16x zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
    h8 xr = *reinterpret_cast<const h8 *>(bre), xi = *reinterpret_cast<const h8 *>(bim);
    h8 a0 = af[0], a1 = af[64];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        h8 xr_n = xr, xi_n = xi, a0_n = a0, a1_n = a1;
        if (s + 1 < KS) {
            xr_n = *reinterpret_cast<const h8 *>(bre + 16 * (s + 1) + 8 * ((s + 1) >> 1));
            xi_n = *reinterpret_cast<const h8 *>(bim + 16 * (s + 1) + 8 * ((s + 1) >> 1));
            a0_n = af[((s + 1) * 2 + 0) * 64];
            a1_n = af[((s + 1) * 2 + 1) * 64];
        }
        are = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xr, (FIRST && s == 0) ? zero : are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, xi, (FIRST && s == 0) ? zero : aim, 0, 0, 0);
        are = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xr, are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, xi, aim, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        xr = xr_n;
        xi = xi_n;
        a0 = a0_n;
        a1 = a1_n;
    }
    // the next pass / tile rewrites the window: the reads above are done (the LDS is in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the A-fragment image -> LDS (idempotent; four fragments per turn in flight, not unrolled further: every
// unrolled load would keep a hoisted 64-bit address pair alive across the whole tile loop)
template <int KS>
__device__ __noinline__ void mfma_fetch_image(const void *image, unsigned char *smem, uint32_t tid) {
    v4u *dst = reinterpret_cast<v4u *>(smem) + tid;
    const gptr128 src = reinterpret_cast<gptr128>((gbytes)image) + tid;
    for (int i = 0; i < 2 * KS; i += 4) {
        const v4u f0 = src[64 * i], f1 = src[64 * i + 64], f2 = src[64 * i + 128], f3 = src[64 * i + 192];
        dst[64 * i] = f0;
        dst[64 * i + 64] = f1;
        dst[64 * i + 128] = f2;
        dst[64 * i + 192] = f3;
    }
}

template <int KS>
__global__ __launch_bounds__(64 * mfma_waves<KS>()) __attribute__((amdgpu_waves_per_eu(4)))
void fir1_mfma_kernel(const FrontParams p) {
    constexpr int kMfmaWaves = mfma_waves<KS>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using Gm = MfmaGeom<KS>;
    constexpr uint32_t kImgBytes = (uint32_t)KS * 2u * 1024u;
    uint32_t tid = threadIdx.x & 63u;       // (made opaque once per tile: see the loop head)
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t cap = blockIdx.y;
    MfmaTileCtx c;
    c.rs.src = (gptr32)(reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride);
    c.rs.halo = (gptr32)reinterpret_cast<const uint32_t *>(p.halo);
    c.rs.halo_len = p.halo_len;
    c.rs.n_valid = p.n_valid;
    c.ctl = reinterpret_cast<uint32_t *>(smem_raw + kImgBytes);
    c.tickets = p.mfma_g * (uint32_t)kMfmaWaves;
    c.tile_base = p.tile_base;
    c.tile_end = p.tile_end;
    c.xcd_span = p.mfma_xcd_span;
    c.aligned16 = (((uintptr_t)c.rs.src & 15u) == 0);
    typedef __attribute__((address_space(1))) uint64_t *gptr64;
    typedef __attribute__((address_space(1))) uint32_t *gptr32w;
    const gptr64 words = (gptr64)(p.bits + (uint64_t)cap * p.words_per_cap);
    const gptr32w tile_info = (gptr32w)(p.tile_info + (uint64_t)cap * p.tiles_per_cap);
    const h8 *a_img = reinterpret_cast<const h8 *>(smem_raw);
    unsigned char *win = smem_raw + kImgBytes + kMfmaCtlBytes + wave * Gm::lds_bytes;
    _Float16 *pl_re = reinterpret_cast<_Float16 *>(win);
    _Float16 *pl_im = pl_re + Gm::plane;
    typedef __attribute__((address_space(1))) v2fm *gptrf2;
    const gptrf2 fout = p.fir_out ? (gptrf2)(reinterpret_cast<v2fm *>(p.fir_out) + (uint64_t)cap * p.n_out) : (gptrf2)nullptr;

    if (threadIdx.x == 0) {
        c.ctl[0] = 0;
        c.ctl[1] = 0;
    }
    __syncthreads();


    uint64_t tile = 0;
    if (!mfma_take_ticket(c, tid, tile)) return;
    uint4 q[Gm::rounds];
    if (mfma_interior<KS>(c, tile)) mfma_issue_loads<KS>(c, tile, tid, q);
    else mfma_boundary_loads<KS>(c, tile, tid, win, q);
    for (;;) {
        // Everything per lane below is a function of the lane id; left alone the compiler computes a dozen
        // offsets and addresses from it ONCE, keeps them alive across the whole loop and, out of registers,
        // reloads them from scratch right in front of the loads.  Opaque lane id => recomputed per tile
        // (a handful of VALU instructions), one live register.
        asm volatile("" : "+v"(tid));
        const uint32_t n = tid & 31u, hh = tid >> 5;
        const uint64_t t0 = tile * kMfmaTile;
        // ---- quiet test (exact: kernels.hip) and range of the window ------------------------------
        v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
#pragma unroll
        for (int i = 0; i < Gm::rounds; ++i) {
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2sm(q[i].x), as_v2sm(q[i].y)));
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2sm(q[i].z), as_v2sm(q[i].w)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2sm(q[i].x), as_v2sm(q[i].y)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2sm(q[i].z), as_v2sm(q[i].w)));
        }
        const int L = p.quiet_lsb;
        const bool loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
        const bool quiet = (!fout && __ballot(loud) == 0) || (p.mfma_debug & 1u);
        const bool wide = __ballot(mx.x > 2047 || mx.y > 2047 || mn.x < -2048 || mn.y < -2048) != 0;

        uint64_t tile_n = 0;
        bool more = false, pre = false;
        if (quiet) {
            more = mfma_take_ticket(c, tid, tile_n);
            pre = more && mfma_interior<KS>(c, tile_n);
            if (pre) mfma_issue_loads<KS>(c, tile_n, tid, q);
            if (!p.sparse) {
                if (tid < kMfmaTile / 64) *reinterpret_cast<gptr64>(uniform_ptr((gbytes_w)(words + (t0 >> 6))) + 8u * tid) = 0;
                if (tid == 0) *uniform_ptr(tile_info + tile) = 0;
            }
            if (p.quiet_count && tid == 0) atomicAdd(p.quiet_count + ((blockIdx.x * kMfmaWaves + wave) % kQuietCounters), 1u);
        } else {
            if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
                mfma_fetch_image<KS>(p.mfma_a, smem_raw, tid);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) __hip_atomic_store(&c.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            f16x are, aim;
            if (!wide) {
                mfma_convert<KS>(q, 0xffffffffu, tid, pl_re, pl_im);
                more = mfma_take_ticket(c, tid, tile_n);
                pre = more && mfma_interior<KS>(c, tile_n);
                if (pre) mfma_issue_loads<KS>(c, tile_n, tid, q);
                mfma_ksteps<KS, true>(a_img, pl_re, pl_im, tid, are, aim);
            } else {
                // a sample beyond +-2048 somewhere in the window: upper bits, then the low five bits
                mfma_convert<KS>(q, 0xffe0ffe0u, tid, pl_re, pl_im);
                mfma_ksteps<KS, true>(a_img, pl_re, pl_im, tid, are, aim);
                mfma_convert<KS>(q, 0x001f001fu, tid, pl_re, pl_im);
                more = mfma_take_ticket(c, tid, tile_n);
                pre = more && mfma_interior<KS>(c, tile_n);
                if (pre) mfma_issue_loads<KS>(c, tile_n, tid, q);
                mfma_ksteps<KS, false>(a_img, pl_re, pl_im, tid, are, aim);
            }

            // ---- power, threshold, guard band: register r of lane (n, hh) is output 32 n + row(r) ------
            // Per pair of outputs: two packed squares and a packed add; per output: one compare per bound
            // whose wave-wide result lands in an SGPR pair, the lane's own bit shifted into `m16` through the
            // carry (r runs downwards so bit r ends at position r), the "inside the band" masks OR-ed on the
            // scalar unit.
            // (the bands arrive in accumulator units: y = c z with c a power of two, so fl(y^2) = c^2 fl(z^2) and
            //  these comparisons are the comparisons of the powers against the band, bit for bit; the host only
            //  takes this path when the scaling of the band edges is exact: mfma_scale_band)
            const float plo = wide ? p.p_lo_w : p.p_lo_n;
            const float phi = wide ? p.p_hi_w : p.p_hi_n;
            const uint32_t ocol = 32u * n + 4u * hh;
            uint32_t m16 = 0;
            uint64_t any_unsure = 0;
#pragma unroll
            for (int r2 = 7; r2 >= 0; --r2) {
                const v2fm zr = (v2fm){are[2 * r2], are[2 * r2 + 1]}, zi = (v2fm){aim[2 * r2], aim[2 * r2 + 1]};
                const v2fm rr = zr * zr, ii = zi * zi;
                const v2fm pw = rr + ii;
                {
                    const uint64_t ge_hi = __ballot(pw.y >= phi);
                    asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(m16) : "s"(ge_hi) : "vcc");
                    any_unsure |= __ballot(pw.y >= plo) & ~ge_hi;
                }
                {
                    const uint64_t ge_hi = __ballot(pw.x >= phi);
                    asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(m16) : "s"(ge_hi) : "vcc");
                    any_unsure |= __ballot(pw.x >= plo) & ~ge_hi;
                }
            }
            if (any_unsure != 0) {
                // some lane of this wave has an output inside the band: those lanes redo their borderline
                // outputs in the reference's exact order (from the capture itself)
                uint32_t todo = 0, redo = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float rr = are[r] * are[r], ii = aim[r] * aim[r];
                    const float pw = rr + ii;
                    if (pw >= plo && !(pw >= phi)) todo |= 1u << r;
                }
                while (todo) {
                    const uint32_t r = (uint32_t)__ffs((int)todo) - 1u;
                    todo &= todo - 1u;
                    const uint64_t o = t0 + ocol + (r & 3u) + 8u * (r >> 2);
                    if (o >= p.n_out) continue;
                    const float2 y = mfma_exact_output(c.rs, (gptrf)p.taps, p.stage[0].ntaps, (int64_t)o);
                    const float rr = y.x * y.x, ii = y.y * y.y;
                    const float pe = rr + ii;
                    m16 = (m16 & ~(1u << r)) | ((pe >= p.p_star ? 1u : 0u) << r);
                    redo++;
                }
                if (redo && p.recompute_count) atomicAdd(p.recompute_count, (unsigned long long)redo);
            }
            if (fout) {
                const float cs = p.mfma_c;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint64_t o = t0 + ocol + (r & 3) + 8 * (r >> 2);
                    if (o < p.n_out) fout[o] = (v2fm){are[r] * cs, aim[r] * cs};
                }
            }

            // ---- 32 bits per column: rows (r & 3) + 8 (r >> 2) + 4 hh ---------------------------------
            uint32_t m32 = (m16 & 0xfu) | ((m16 & 0xf0u) << 4) | ((m16 & 0xf00u) << 8) | ((m16 & 0xf000u) << 12);
            m32 <<= 4u * hh;
            uint32_t w32 = m32 | (uint32_t)__shfl_xor((int)m32, 32);
            {
                // outputs past the end of the (padded) capture do not exist
                const uint64_t c0 = t0 + 32u * n;
                if (c0 + 32u > p.n_out) {
                    const uint32_t keep = c0 >= p.n_out ? 0u : (uint32_t)(p.n_out - c0);
                    w32 &= (1u << keep) - 1u;       // keep < 32 here
                }
            }
            // level changes inside the tile (the tile's first bit against the tile before NOT included)
            {
                const uint32_t prev_top = (uint32_t)__shfl_up((int)(w32 >> 31), 1);
                uint32_t ch = w32 ^ (w32 << 1);
                if (n != 0) ch ^= prev_top & 1u;
                else ch &= ~1u;
                uint32_t cnt = hh == 0 ? (uint32_t)__popc(ch) : 0u;
#pragma unroll
                for (int d = 16; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
                const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(w32 & 1u));
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)(w32 >> 31), 31);
                // the word that holds the first change: lane n < 32 holds bits 32 n .. 32 n + 31 of the tile
                const uint64_t chl = __ballot(hh == 0 && ch != 0);
                const uint32_t widx = chl ? (uint32_t)__builtin_ctzll(chl) >> 1 : 0u;
                if (tid == 0) *uniform_ptr(tile_info + tile) = cnt | (widx << kTileWordShift) | (first << 30) | (last << 31) | p.stamp_bits;
            }
            const uint32_t up = (uint32_t)__shfl_xor((int)w32, 1);
            if (hh == 0 && (n & 1u) == 0) {
                const gbytes_w wb = uniform_ptr((gbytes_w)(words + (t0 >> 6)));
                *reinterpret_cast<gptr64>(wb + 4u * n) = (uint64_t)w32 | ((uint64_t)up << 32);
            }
        }   // loud tile

        if (!more) break;
        if (!pre) mfma_boundary_loads<KS>(c, tile_n, tid, win, q);       // (the LDS window is free: every read of it is done)
        tile = tile_n;
    }
}

// ---------------------------------------------------------------------------
// two decimate-by-2 stages (the backend default fs128_fs16_dec4: 16 taps / 2, 32 taps / 2) on the matrix cores
// ---------------------------------------------------------------------------
//
// y2[m] = sum_k2 h2[k2] y1[2m + 1 - k2],  y1[j] = sum_k1 h1[k1] x[2j + 1 - k1]   (fir.c:290: the countdown starts at D)
//       = sum_t g[t] x[4m + 3 - t],  g[t] = sum_{2 k2 + k1 = t} h2[k2] h1[k1],  t = 0 .. 2 (T2 - 1) + T1 - 1 <= 77:
// ONE decimate-by-4 filter of up to 78 taps.  In real arithmetic the two are the same; the reference rounds every
// multiply and add of both stages to fp32, and what that (and the fp32 accumulation here, and the part of g the two
// fp16 pieces do not carry) can amount to goes into the guard band: an output inside it is recomputed exactly as the
// reference does -- both stages, tap 0 first, separately rounded multiply and add -- from the capture itself.
//
// Product: a wave tile is 256 final outputs = 1024 input samples = 16 columns of 16 outputs (64 inputs each),
// v_mfma_f32_16x16x32_f16.  With window sample j <-> input a0 + j, a0 = 4 M0 - 96:
//   y2[M0 + 16 n + i] = sum_kk A[i][kk] win[64 n + kk],  A[i][kk] = g[4 i + 99 - kk],  kk = 0 .. 159: 5 K-steps of 32,
// x 2 tap pieces x {re, im} = 20 MFMAs of 16 cycles per 1024 input samples (x 2 for a tile with samples beyond +-2048).
// C: lane (n = l & 15, q = l >> 4), register r = output 16 n + 4 q + r.
constexpr uint32_t kF2Out = 256;                // final outputs per wave tile
constexpr uint32_t kF2Tp = 96;                  // input history in the window (>= 77)
constexpr uint32_t kF2W = 4 * kF2Out + kF2Tp;   // 1120 window samples
constexpr uint32_t kF2Nvec = kF2W / 4;          // 280 raw vectors
constexpr int kF2Rounds = (int)((kF2Nvec + 63) / 64);
constexpr int kF2KS = 5;
constexpr int kF2Waves = 4;
// one pad chunk (8 halfs) per 64 samples: column stride 72 halfs
__host__ __device__ constexpr uint32_t f2slot(uint32_t j) { return j + 8u * (j >> 6); }
constexpr uint32_t kF2Plane = f2slot(kF2W) + 8u;
constexpr uint32_t kF2WaveBytes = kF2Plane * 2u * 2u;
constexpr uint32_t kF2ImgBytes = kF2KS * 2u * 1024u;
constexpr uint32_t kF2LdsBytes = kF2ImgBytes + kMfmaCtlBytes + kF2Waves * kF2WaveBytes;

typedef float f4x __attribute__((ext_vector_type(4)));

struct F2Taps {                 // what the exact recompute needs
    gptrf taps1, taps2;
    uint32_t n1, n2;
};

// Reference-order recomputation of final output m: the n2 stage-1 outputs it reads, each from the capture, then stage 2
__device__ __forceinline__ float2 fir2_mfma_exact_output(const RawSrc &rs, const F2Taps &ft, int64_t m) {
    const float s = 1.0f / 2048.0f;
    float re2 = 0.0f, im2 = 0.0f;
    for (uint32_t k2 = 0; k2 < ft.n2; ++k2) {
        const int64_t j1 = 2 * m + 1 - (int64_t)k2;             // stage-1 output index
        float re1 = 0.0f, im1 = 0.0f;
        // (a stage-1 output in front of the capture comes out of the samples in front of it: zeros -- the reference's
        //  stage buffers start as zeros, fir.c:282-293 -- or the previous shard's samples, the halo)
        for (uint32_t k1 = 0; k1 < ft.n1; ++k1) {
            const uint32_t w = fetch_raw_m(rs, 2 * j1 + 1 - (int64_t)k1);
            const float xr = (float)(int16_t)(w & 0xffffu) * s;
            const float xi = (float)(int16_t)(w >> 16) * s;
            const float t = ft.taps1[k1];
            const float pr = t * xr;
            const float pi = t * xi;
            re1 = re1 + pr;
            im1 = im1 + pi;
        }
        const float t2 = ft.taps2[k2];
        const float pr = t2 * re1;
        const float pi = t2 * im1;
        re2 = re2 + pr;
        im2 = im2 + pi;
    }
    return make_float2(re2, im2);
}

__device__ __forceinline__ bool f2_interior(const MfmaTileCtx &c, uint64_t tile) {
    const uint64_t i0 = tile * (4ull * kF2Out);
    return c.aligned16 && i0 >= kF2Tp && i0 + 4ull * kF2Out <= c.rs.n_valid;
}

__device__ __forceinline__ void f2_issue_loads(const MfmaTileCtx &c, uint64_t tile, uint32_t tid, uint4 (&q)[kF2Rounds]) {
    const gbytes src4 = uniform_ptr((gbytes)(c.rs.src + (tile * (4ull * kF2Out) - kF2Tp)));
#pragma unroll
    for (int i = 0; i < kF2Rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        q[i] = ld_nt4_at(src4, 16u * ((64u * (i + 1) <= kF2Nvec || v < kF2Nvec) ? v : kF2Nvec - 1u));
    }
}

__device__ __noinline__ void f2_boundary_stage(RawSrc rs, uint64_t i0, uint32_t tid, uint4 *stage) {
    for (uint32_t v = tid; v < kF2Nvec; v += 64u) {
        const int64_t s0 = (int64_t)i0 - (int64_t)kF2Tp + 4 * (int64_t)v;
        uint4 w;
        w.x = fetch_raw_m(rs, s0);
        w.y = fetch_raw_m(rs, s0 + 1);
        w.z = fetch_raw_m(rs, s0 + 2);
        w.w = fetch_raw_m(rs, s0 + 3);
        stage[v] = w;
    }
}

__device__ __forceinline__ void f2_boundary_loads(const MfmaTileCtx &c, uint64_t tile, uint32_t tid, unsigned char *win,
                                                  uint4 (&q)[kF2Rounds]) {
    uint4 *stage = reinterpret_cast<uint4 *>(win);
    f2_boundary_stage(c.rs, tile * (4ull * kF2Out), tid, stage);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < kF2Rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        q[i] = stage[(64u * (i + 1) <= kF2Nvec || v < kF2Nvec) ? v : kF2Nvec - 1u];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void f2_convert(const uint4 (&q)[kF2Rounds], uint32_t mask, uint32_t tid, _Float16 *pl_re,
                                           _Float16 *pl_im) {
#pragma unroll
    for (int i = 0; i < kF2Rounds; ++i) {
        const uint32_t v = tid + 64u * i;
        if (64u * (i + 1) <= kF2Nvec || v < kF2Nvec) {
            uint2 r4, i4;
            cvt4(q[i], mask, r4, i4);
            *reinterpret_cast<uint2 *>(pl_re + f2slot(4u * v)) = r4;
            *reinterpret_cast<uint2 *>(pl_im + f2slot(4u * v)) = i4;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool FIRST>
__device__ __forceinline__ void f2_ksteps(const h8 *a_img, const _Float16 *pl_re, const _Float16 *pl_im, uint32_t tid,
                                          f4x &are, f4x &aim) {
    const uint32_t n = tid & 15u, g = tid >> 4;
    // window sample 64 n + 32 s + 8 g: slot = that + 8 * (n + (32 s + 8 g) / 64) = 72 n + 32 s + 8 g  (32 s + 8 g < 64 only
    // for s < 2: the pad of the NEXT 64-block comes in with s >= 2)
    const _Float16 *bre = pl_re + 72u * n + 8u * g;
    const _Float16 *bim = pl_im + 72u * n + 8u * g;
    const h8 *af = a_img + tid;
    f4x zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < kF2KS; ++s) {
        // (32 s + 8 g) / 64 depends on g only when 32 s + 8 g crosses a multiple of 64: it does not (32 s is a multiple
        //  of 32, 8 g <= 24), so the pad count is (32 s) / 64 = s / 2 for every lane
        const h8 xr = *reinterpret_cast<const h8 *>(bre + 32 * s + 8 * (s >> 1));
        const h8 xi = *reinterpret_cast<const h8 *>(bim + 32 * s + 8 * (s >> 1));
        const h8 a0 = af[(s * 2 + 0) * 64];
        const h8 a1 = af[(s * 2 + 1) * 64];
        are = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xr, (FIRST && s == 0) ? zero : are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xi, (FIRST && s == 0) ? zero : aim, 0, 0, 0);
        are = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xr, are, 0, 0, 0);
        aim = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xi, aim, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __noinline__ void f2_fetch_image(const void *image, unsigned char *smem, uint32_t tid) {
    v4u *dst = reinterpret_cast<v4u *>(smem) + tid;
    const gptr128 src = reinterpret_cast<gptr128>((gbytes)image) + tid;
    for (int i = 0; i < 2 * kF2KS; i += 2) {
        const v4u f0 = src[64 * i], f1 = src[64 * i + 64];
        dst[64 * i] = f0;
        dst[64 * i + 64] = f1;
    }
}

__global__ __launch_bounds__(64 * kF2Waves) __attribute__((amdgpu_waves_per_eu(4)))
void fir2_mfma_kernel(const FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t tid = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t cap = blockIdx.y;
    MfmaTileCtx c;
    c.rs.src = (gptr32)(reinterpret_cast<const uint32_t *>(p.iq) + (uint64_t)cap * p.cap_stride);
    c.rs.halo = (gptr32)reinterpret_cast<const uint32_t *>(p.halo);
    c.rs.halo_len = p.halo_len;
    c.rs.n_valid = p.n_valid;
    c.ctl = reinterpret_cast<uint32_t *>(smem_raw + kF2ImgBytes);
    c.tickets = p.mfma_g * (uint32_t)kF2Waves;
    c.tile_base = p.tile_base;
    c.tile_end = p.tile_end;
    c.xcd_span = p.mfma_xcd_span;
    c.aligned16 = (((uintptr_t)c.rs.src & 15u) == 0);
    F2Taps ft;
    ft.taps1 = (gptrf)(p.taps + p.stage[0].tap_off);
    ft.taps2 = (gptrf)(p.taps + p.stage[1].tap_off);
    ft.n1 = p.stage[0].ntaps;
    ft.n2 = p.stage[1].ntaps;
    typedef __attribute__((address_space(1))) uint64_t *gptr64;
    typedef __attribute__((address_space(1))) uint32_t *gptr32w;
    const gptr64 words = (gptr64)(p.bits + (uint64_t)cap * p.words_per_cap);
    const gptr32w tile_info = (gptr32w)(p.tile_info + (uint64_t)cap * p.tiles_per_cap);
    const h8 *a_img = reinterpret_cast<const h8 *>(smem_raw);
    unsigned char *win = smem_raw + kF2ImgBytes + kMfmaCtlBytes + wave * kF2WaveBytes;
    _Float16 *pl_re = reinterpret_cast<_Float16 *>(win);
    _Float16 *pl_im = pl_re + kF2Plane;
    typedef __attribute__((address_space(1))) v2fm *gptrf2;
    const gptrf2 fout = p.fir_out ? (gptrf2)(reinterpret_cast<v2fm *>(p.fir_out) + (uint64_t)cap * p.n_out) : (gptrf2)nullptr;

    if (threadIdx.x == 0) {
        c.ctl[0] = 0;
        c.ctl[1] = 0;
    }
    __syncthreads();

    uint64_t tile = 0;
    if (!mfma_take_ticket(c, tid, tile)) return;
    uint4 q[kF2Rounds];
    if (f2_interior(c, tile)) f2_issue_loads(c, tile, tid, q);
    else f2_boundary_loads(c, tile, tid, win, q);
    for (;;) {
        asm volatile("" : "+v"(tid));
        const uint32_t n = tid & 15u, g = tid >> 4;
        const uint64_t M0 = tile * kF2Out;
        v2s mx = (v2s){0, 0}, mn = (v2s){0, 0};
#pragma unroll
        for (int i = 0; i < kF2Rounds; ++i) {
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2sm(q[i].x), as_v2sm(q[i].y)));
            mx = __builtin_elementwise_max(mx, __builtin_elementwise_max(as_v2sm(q[i].z), as_v2sm(q[i].w)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2sm(q[i].x), as_v2sm(q[i].y)));
            mn = __builtin_elementwise_min(mn, __builtin_elementwise_min(as_v2sm(q[i].z), as_v2sm(q[i].w)));
        }
        const int L = p.quiet_lsb;
        const bool loud = !(mx.x < L && mx.y < L && mn.x > -L && mn.y > -L);
        const bool quiet = (!fout && __ballot(loud) == 0) || (p.mfma_debug & 1u);
        const bool wide = __ballot(mx.x > 2047 || mx.y > 2047 || mn.x < -2048 || mn.y < -2048) != 0;

        uint64_t tile_n = 0;
        bool more = false, pre = false;
        if (quiet) {
            more = mfma_take_ticket(c, tid, tile_n);
            pre = more && f2_interior(c, tile_n);
            if (pre) f2_issue_loads(c, tile_n, tid, q);
            if (!p.sparse) {
                if (tid < kF2Out / 64) *reinterpret_cast<gptr64>(uniform_ptr((gbytes_w)(words + (M0 >> 6))) + 8u * tid) = 0;
                if (tid == 0) *uniform_ptr(tile_info + tile) = 0;
            }
            if (p.quiet_count && tid == 0) atomicAdd(p.quiet_count + ((blockIdx.x * kF2Waves + wave) % kQuietCounters), 1u);
        } else {
            if (__hip_atomic_load(&c.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
                f2_fetch_image(p.mfma_a, smem_raw, tid);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) __hip_atomic_store(&c.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            f4x are, aim;
            if (!wide) {
                f2_convert(q, 0xffffffffu, tid, pl_re, pl_im);
                more = mfma_take_ticket(c, tid, tile_n);
                pre = more && f2_interior(c, tile_n);
                if (pre) f2_issue_loads(c, tile_n, tid, q);
                f2_ksteps<true>(a_img, pl_re, pl_im, tid, are, aim);
            } else {
                f2_convert(q, 0xffe0ffe0u, tid, pl_re, pl_im);
                f2_ksteps<true>(a_img, pl_re, pl_im, tid, are, aim);
                f2_convert(q, 0x001f001fu, tid, pl_re, pl_im);
                more = mfma_take_ticket(c, tid, tile_n);
                pre = more && f2_interior(c, tile_n);
                if (pre) f2_issue_loads(c, tile_n, tid, q);
                f2_ksteps<false>(a_img, pl_re, pl_im, tid, are, aim);
            }
            // ---- power, threshold, guard band: register r of lane (n, g) is output 16 n + 4 g + r ------------
            const float plo = wide ? p.p_lo_w : p.p_lo_n;
            const float phi = wide ? p.p_hi_w : p.p_hi_n;
            const uint32_t o_lane = 16u * n + 4u * g;
            uint32_t nib = 0, unsure = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rr = are[r] * are[r], ii = aim[r] * aim[r];
                const float pw = rr + ii;
                const bool hi = pw >= phi;
                nib |= (hi ? 1u : 0u) << r;
                unsure |= ((!hi && pw >= plo) ? 1u : 0u) << r;
            }
            if (__ballot(unsure != 0) != 0) {
                uint32_t todo = unsure, redo = 0;
                while (todo) {
                    const uint32_t r = (uint32_t)__ffs((int)todo) - 1u;
                    todo &= todo - 1u;
                    const uint64_t o = M0 + o_lane + r;
                    if (o >= p.n_out) continue;
                    const float2 y = fir2_mfma_exact_output(c.rs, ft, (int64_t)o);
                    const float rr = y.x * y.x, ii = y.y * y.y;
                    const float pe = rr + ii;
                    nib = (nib & ~(1u << r)) | ((pe >= p.p_star ? 1u : 0u) << r);
                    redo++;
                }
                if (redo && p.recompute_count) atomicAdd(p.recompute_count, (unsigned long long)redo);
            }
            if (fout) {
                const float cs = p.mfma_c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t o = M0 + o_lane + r;
                    if (o < p.n_out) fout[o] = (v2fm){are[r] * cs, aim[r] * cs};
                }
            }
            // outputs past the end of the (padded) capture do not exist
            if (M0 + o_lane + 4u > p.n_out) {
                const uint32_t keep = M0 + o_lane >= p.n_out ? 0u : (uint32_t)(p.n_out - (M0 + o_lane));
                nib &= (1u << keep) - 1u;
            }
            // ---- the tile's four 64-bit words: lane (n, g) holds bits 16 (n & 3) + 4 g .. + 3 of word n >> 2 ------
            uint64_t w64 = (uint64_t)nib << (16u * (n & 3u) + 4u * g);
#pragma unroll
            for (int d = 1; d <= 32; d <<= 1) {
                if (d == 4 || d == 8) continue;         // lanes that differ in n >> 2 hold other words
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)w64, d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(w64 >> 32), d);
                w64 |= (uint64_t)lo | ((uint64_t)hi << 32);
            }
            // level changes inside the tile (the tile's first bit against the tile before NOT included)
            {
                const uint32_t wq = n >> 2;                                     // this lane's word
                const uint32_t top_prev = (uint32_t)__shfl((int)(uint32_t)(w64 >> 63), (int)((wq ? wq - 1u : 0u) << 2));   // lane 4 (wq - 1)
                uint64_t ch = w64 ^ ((w64 << 1) | (wq ? (uint64_t)(top_prev & 1u) : 0ull));
                if (wq == 0) ch &= ~1ull;
                uint32_t cnt = ((n & 3u) == 0 && g == 0) ? (uint32_t)__popcll(ch) : 0u;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
                const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w64 & 1ull));
                const uint32_t last = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w64 >> 63), 12);
                // the word that holds the first change: lanes 0, 4, 8, 12 hold the tile's words 0 .. 3
                const uint64_t chl = __ballot((n & 3u) == 0 && g == 0 && ch != 0);
                const uint32_t widx = chl ? (uint32_t)__builtin_ctzll(chl) >> 2 : 0u;
                if (tid == 0) *uniform_ptr(tile_info + tile) = cnt | (widx << kTileWordShift) | (first << 30) | (last << 31) | p.stamp_bits;
            }
            if ((n & 3u) == 0 && g == 0) {
                const gbytes_w wb = uniform_ptr((gbytes_w)(words + (M0 >> 6)));
                *reinterpret_cast<gptr64>(wb + 2u * n) = w64;           // word n >> 2 at byte 8 (n >> 2) = 2 n
            }
        }   // loud tile

        if (!more) break;
        if (!pre) f2_boundary_loads(c, tile_n, tid, win, q);
        tile = tile_n;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

static uint16_t half_bits(_Float16 h) {
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

static int mfma_ksteps_for(uint32_t ntaps) {
    // compiled window lengths: Tp = 32, 64, 128, 256 (history of Tp >= T - 1 samples)
    if (ntaps == 0) return 0;
    if (ntaps <= 32) return 4;
    if (ntaps <= 64) return 6;
    if (ntaps <= 128) return 10;
    if (ntaps <= 256) return 18;
    return 0;
}

bool mfma_prepare_taps(const float *taps, uint32_t ntaps, MfmaTaps &out) {
    out = MfmaTaps();
    const int KS = mfma_ksteps_for(ntaps);
    if (KS == 0) return false;
    double hmax = 0.0, sum_abs = 0.0;
    for (uint32_t k = 0; k < ntaps; ++k) {
        if (!std::isfinite(taps[k])) return false;
        hmax = std::max(hmax, std::fabs((double)taps[k]));
        sum_abs += std::fabs((double)taps[k]);
    }
    if (!(hmax > 0.0)) return false;
    int e = 0;
    (void)std::frexp(hmax, &e);                 // hmax = m * 2^e, m in [0.5, 1)
    const int sh = 15 - e;                      // hmax * 2^sh in [2^14, 2^15)
    // c = 2^-11 / S must be a normal float, and so must S itself
    if (sh > 100 || sh < -100) return false;
    const double S = std::ldexp(1.0, sh);
    const double kMinNormal = std::ldexp(1.0, -14);
    std::vector<_Float16> p1(ntaps), p2(ntaps);
    double delta = 0.0, sum_hat = 0.0;
    for (uint32_t k = 0; k < ntaps; ++k) {
        const double hs = (double)taps[k] * S;          // exact: a power of two
        _Float16 a = (_Float16)hs;
        if (std::fabs((double)a) < kMinNormal) a = (_Float16)0.0;       // no fp16 subnormals on the matrix cores
        const double r1 = hs - (double)a;
        _Float16 b = (_Float16)r1;
        if (std::fabs((double)b) < kMinNormal) b = (_Float16)0.0;
        const double r2 = r1 - (double)b;
        if (!std::isfinite((double)a) || !std::isfinite((double)b)) return false;
        p1[k] = a;
        p2[k] = b;
        delta += std::fabs(r2) / S;
        sum_hat += std::fabs((double)a + (double)b) / S;
    }
    const uint32_t Tp = 16u * (uint32_t)(KS - 2);
    out.ksteps = (uint32_t)KS;
    out.image.assign((size_t)KS * 2 * 64 * 8, 0);
    for (int s = 0; s < KS; ++s) {
        for (int pc = 0; pc < 2; ++pc) {
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t r = l & 31u, hh = l >> 5;
                for (uint32_t j = 0; j < 8; ++j) {
                    const int64_t kk = 16 * s + 8 * (int64_t)hh + j;
                    const int64_t t = (int64_t)r + Tp - kk;
                    _Float16 v = (_Float16)0.0;
                    if (t >= 0 && t < (int64_t)ntaps) v = pc == 0 ? p1[t] : p2[t];
                    out.image[(((size_t)s * 2 + pc) * 64 + l) * 8 + j] = half_bits(v);
                }
            }
        }
    }
    out.c = (float)std::ldexp(1.0, -11 - sh);
    out.delta = delta;
    out.sum_abs = sum_abs;
    out.sum_hat = sum_hat;
    return true;
}

// Two decimate-by-2 stages folded into one decimate-by-4 filter g (see fir2_mfma_kernel): pieces, image, residue.
bool mfma_prepare_taps2(const float *taps1, uint32_t n1, const float *taps2, uint32_t n2, MfmaTaps &out) {
    out = MfmaTaps();
    if (n1 == 0 || n2 == 0 || n1 > 16 || n2 > 32) return false;
    const uint32_t ng = 2 * (n2 - 1) + n1;                      // <= 78
    std::vector<double> g(ng, 0.0), gabs(ng, 0.0);
    for (uint32_t k2 = 0; k2 < n2; ++k2) {
        for (uint32_t k1 = 0; k1 < n1; ++k1) {
            if (!std::isfinite(taps1[k1]) || !std::isfinite(taps2[k2])) return false;
            g[2 * k2 + k1] += (double)taps2[k2] * (double)taps1[k1];            // exact products; the sums round at 2^-53
            gabs[2 * k2 + k1] += std::fabs((double)taps2[k2] * (double)taps1[k1]);
        }
    }
    double gmax = 0.0, sum_abs = 0.0;
    for (uint32_t t = 0; t < ng; ++t) {
        gmax = std::max(gmax, std::fabs(g[t]));
        sum_abs += gabs[t];
    }
    if (!(gmax > 0.0)) return false;
    int e = 0;
    (void)std::frexp(gmax, &e);
    const int sh = 15 - e;
    if (sh > 100 || sh < -100) return false;
    const double S = std::ldexp(1.0, sh);
    const double kMinNormal = std::ldexp(1.0, -14);
    std::vector<_Float16> p1(ng), p2(ng);
    double delta = 0.0, sum_hat = 0.0;
    for (uint32_t t = 0; t < ng; ++t) {
        const double hs = g[t] * S;
        _Float16 a = (_Float16)hs;
        if (std::fabs((double)a) < kMinNormal) a = (_Float16)0.0;
        const double r1 = hs - (double)a;
        _Float16 b = (_Float16)r1;
        if (std::fabs((double)b) < kMinNormal) b = (_Float16)0.0;
        const double r2 = r1 - (double)b;
        if (!std::isfinite((double)a) || !std::isfinite((double)b)) return false;
        p1[t] = a;
        p2[t] = b;
        delta += std::fabs(r2) / S + gabs[t] * std::ldexp(1.0, -50);       // (+ the double rounding of the fold)
        sum_hat += std::fabs((double)a + (double)b) / S;
    }
    out.ksteps = (uint32_t)kF2KS;
    out.image.assign((size_t)kF2KS * 2 * 64 * 8, 0);
    for (int s = 0; s < kF2KS; ++s) {
        for (int pc = 0; pc < 2; ++pc) {
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t i = l & 15u, q = l >> 4;
                for (uint32_t j = 0; j < 8; ++j) {
                    const int64_t kk = 32 * s + 8 * (int64_t)q + j;
                    const int64_t t = 4 * (int64_t)i + 99 - kk;
                    _Float16 v = (_Float16)0.0;
                    if (t >= 0 && t < (int64_t)ng) v = pc == 0 ? p1[t] : p2[t];
                    out.image[(((size_t)s * 2 + pc) * 64 + l) * 8 + j] = half_bits(v);
                }
            }
        }
    }
    out.c = (float)std::ldexp(1.0, -11 - sh);
    out.delta = delta;
    out.sum_abs = sum_abs;
    out.sum_hat = sum_hat;
    return true;
}

// forward bound on |y_mfma - y_ref| per component for the folded two-stage filter: `e_ref` = what the two-stage
// reference chain may differ from real arithmetic by (rx.cpp: guard_error, for samples up to x_max), plus the
// accumulation here and the residue of the pieces
double mfma_error_bound2(const MfmaTaps &t, double e_ref, bool wide) {
    const double u = std::ldexp(1.0, -24);
    const double xmax = wide ? 16.0 : 1.0;
    const double chain = (double)t.ksteps * 32.0 * 2.0 * (wide ? 2.0 : 1.0);
    const double e_acc = chain * 2.0 * u * t.sum_hat;
    return 1.1 * (e_ref + (e_acc + t.delta) * xmax) + 78.0 * std::ldexp(1.0, -140);
}

double mfma_error_bound(const MfmaTaps &t, uint32_t ntaps, bool wide) {
    // |y_mfma - y_ref| per component, x in units of 1 (= 2048 LSB):
    //   reference chain against the real sum: (T + 1) u sum|h| xmax   (T products, T sums)
    //   matrix-core accumulation against the real sum of the split taps: every product enters
    //   through one addition; allow each of them a full truncation (2 u) of a partial sum that is
    //   bounded by sum|h^| xmax:  chain length = K-steps x 16 x pieces (x 2 sample pieces)
    //   taps not captured by the two pieces: delta xmax
    const double u = std::ldexp(1.0, -24);
    const double xmax = wide ? 16.0 : 1.0;
    const double chain = (double)t.ksteps * 16.0 * 2.0 * (wide ? 2.0 : 1.0);
    const double e_ref = ((double)ntaps + 1.0) * u * t.sum_abs;
    const double e_acc = chain * 2.0 * u * t.sum_hat;
    return 1.1 * (e_ref + e_acc + t.delta) * xmax + (double)ntaps * std::ldexp(1.0, -140);
}

// band edge (a power, filter output units) -> accumulator units: p / c^2.  False when that is not exact in
// float (the caller then keeps the packed-VALU form)
bool mfma_scale_band(const MfmaTaps &t, float p, float &out) {
    if (std::isnan(p) || std::isinf(p) || p == 0.0f) {
        out = p;
        return true;
    }
    int e = 0;
    (void)std::frexp((double)t.c, &e);          // c = 2^(e-1)
    const double scaled = std::ldexp((double)p, -2 * (e - 1));
    const float f = (float)scaled;
    if (!std::isfinite(f) || (double)f != scaled || std::fabs(f) < 1.1754944e-38f) return false;
    out = f;
    return true;
}

bool front_uses_mfma(const FrontParams &p) {
    return p.mfma_a != nullptr && p.num_stages == 1 && p.stage[0].decim == 1 && p.origin == 0 && !p.iq_f32 &&
           mfma_ksteps_for(p.stage[0].ntaps) != 0;
}

bool front_uses_mfma2(const FrontParams &p) {
    return p.mfma_a != nullptr && p.num_stages == 2 && !p.iq_f32 && !p.halo_f32 && p.stage[0].decim == 2 &&
           p.stage[1].decim == 2 && p.stage[0].ntaps <= 16 && p.stage[1].ntaps <= 32 && p.origin % 4 == 0;
}

hipError_t launch_front_mfma2(const FrontParams &p, uint32_t num_captures, hipStream_t stream, hipEvent_t t0,
                              hipEvent_t t1, uint64_t tile_begin, uint64_t tile_count) {
    const uint64_t all = (p.n_out + kF2Out - 1) / kF2Out;
    const uint64_t b = tile_begin < all ? tile_begin : all;
    const uint64_t cnt = tile_count < all - b ? tile_count : all - b;
    if (cnt == 0) return hipSuccess;
    FrontParams pp = p;
    pp.tile_base = (uint32_t)b;
    pp.tile_end = b + cnt;
    if (pp.mfma_g == 0) pp.mfma_g = 1;
    pp.mfma_xcd_span = (p.mfma_xcd & 2u) && cnt % 8 == 0 && cnt / 8 <= 0xffffffffull ? (uint32_t)(cnt / 8) : 0u;
    const uint64_t grid = (cnt + (uint64_t)pp.mfma_g * kF2Waves - 1) / ((uint64_t)pp.mfma_g * kF2Waves);
    const void *fn = reinterpret_cast<const void *>(&fir2_mfma_kernel);
    hipError_t e = ensure_dynamic_lds(fn, kF2LdsBytes);
    if (e != hipSuccess) return e;
    void *args[] = {&pp};
    e = hipExtLaunchKernel(fn, dim3((uint32_t)grid, num_captures), dim3(64 * kF2Waves), args, kF2LdsBytes, stream, t0, t1, 0);
    return e != hipSuccess ? e : hipGetLastError();
}

template <int KS>
static hipError_t launch_mfma_ks(FrontParams &pp, uint32_t num_captures, uint64_t grid, hipStream_t stream,
                                 hipEvent_t t0, hipEvent_t t1) {
    const void *fn = reinterpret_cast<const void *>(&fir1_mfma_kernel<KS>);
    // (experiment: OOKD_MFMA_LDS_PAD caps the workgroups per CU, leaving registers / wave slots to other streams' kernels)
    static const size_t lds_pad = dev_getenv("OOKD_MFMA_LDS_PAD") ? (size_t)atoi(dev_getenv("OOKD_MFMA_LDS_PAD")) : 0;
    const size_t lds = mfma_lds_bytes<KS>() + lds_pad;
    hipError_t e = ensure_dynamic_lds(fn, lds);
    if (e != hipSuccess) return e;
    void *args[] = {&pp};
    e = hipExtLaunchKernel(fn, dim3((uint32_t)grid, num_captures), dim3(64 * mfma_waves<KS>()), args, lds, stream, t0, t1, 0);
    return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_front_mfma(const FrontParams &p, uint32_t num_captures, hipStream_t stream, hipEvent_t t0,
                             hipEvent_t t1, uint64_t tile_begin, uint64_t tile_count) {
    // whole 4096-output blocks, so every bit word of the capture is written (dense output)
    const uint64_t all = (p.n_out + kFirTile - 1) / kFirTile * (kFirTile / kMfmaTile);
    const uint64_t b = tile_begin < all ? tile_begin : all;
    const uint64_t cnt = tile_count < all - b ? tile_count : all - b;
    if (cnt == 0) return hipSuccess;
    FrontParams pp = p;
    pp.tile_base = (uint32_t)b;
    pp.tile_end = b + cnt;
    if (pp.mfma_g == 0) pp.mfma_g = 1;
    pp.mfma_xcd_span = (p.mfma_xcd & 1u) && cnt % 8 == 0 && cnt / 8 <= 0xffffffffull ? (uint32_t)(cnt / 8) : 0u;
    // pp.mfma_g = tickets per wave; a workgroup of W waves hands out W times as many
    auto grid_for = [&](int waves) { return (cnt + (uint64_t)pp.mfma_g * waves - 1) / ((uint64_t)pp.mfma_g * waves); };
    switch (mfma_ksteps_for(p.stage[0].ntaps)) {
    case 4: return launch_mfma_ks<4>(pp, num_captures, grid_for(mfma_waves<4>()), stream, t0, t1);
    case 6: return launch_mfma_ks<6>(pp, num_captures, grid_for(mfma_waves<6>()), stream, t0, t1);
    case 10: return launch_mfma_ks<10>(pp, num_captures, grid_for(mfma_waves<10>()), stream, t0, t1);
    case 18: return launch_mfma_ks<18>(pp, num_captures, grid_for(mfma_waves<18>()), stream, t0, t1);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace ookd
