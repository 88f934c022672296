// edges_fsm.hip -- level-change extraction and the symbol state machine.
//
//   edges : packed bit words -> sorted list of level changes
//           (the content of --rx-rec-dig, src/ookiedokie.c:146-169)
//   fsm   : the table-driven symbol state machine run over that list,
//           many segments of a capture in parallel, with a fix-point on the
//           state carried from one segment to the next
//           (reference, per sample: src/state_machine.c:421-556; the
//            drop-rest-of-buffer rule: src/device.c:634-658)
//
// Why edges: between two level changes only `always` / `timeout` /
// `msg_complete` triggers can fire, each at a sample count known from the
// tables, so a run of constant level is advanced in O(1) instead of sample
// by sample; the result is identical to feeding every sample
// (SURVEY.md section 7 step 6).
#include "kernels.hpp"

namespace ookd {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t rfl(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v);
}
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ uint64_t rl64(uint64_t v, uint32_t lane) {
    return ((uint64_t)rl((uint32_t)(v >> 32), lane) << 32) | rl((uint32_t)v, lane);
}

// ---------------------------------------------------------------------------
// edges: count per 4096-bit block, two-level exclusive scan, compact
// ---------------------------------------------------------------------------

// word of level changes: bit b set <=> sample 64*w+b differs from its
// predecessor (the sample before a capture counts as 0, as record_dig's
// first line does for sample 0, ookiedokie.c:150-153).
// Changes at or beyond n_out do not exist (the words past the capture are
// zero padding, not samples).
// has_prev: the words continue in front of `words` (a chunk of a pipelined run): the level before
// its first sample is the last bit of the chunk before, not 0.
__device__ __forceinline__ uint64_t change_word(const uint64_t *words, uint64_t w, uint64_t n_out, uint32_t has_prev = 0) {
    const uint64_t cur = words[w];
    const uint64_t prev_top = (w == 0 && !has_prev) ? 0ull : (*(words + w - 1) >> 63);
    const uint64_t e = cur ^ ((cur << 1) | prev_top);
    const uint64_t base = w * 64;
    if (base + 64 <= n_out) return e;
    if (base >= n_out) return 0ull;
    return e & ((1ull << (n_out - base)) - 1ull);
}

__global__ __launch_bounds__(256) void edge_count_kernel(const EdgeParams p) {
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (wave >= total_blocks) return;
    const uint32_t cap = wave / p.blocks_per_cap;
    const uint32_t blk = wave % p.blocks_per_cap;
    const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    const uint64_t e = change_word(words, (uint64_t)blk * kBlockWords + lane_id(), p.n_out, p.has_prev);
    uint32_t c = (uint32_t)__popcll(e);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if (lane_id() == 0) p.blk_count[wave] = c;
}

// level 1: each workgroup scans kScanGroup consecutive block counts
// (coalesced 16 B per lane), writes group-local exclusive prefixes and the
// group total.
__global__ __launch_bounds__(256) void edge_scan_local_kernel(const EdgeParams p) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t wave_sum[4];
    const uint32_t n = p.num_captures * p.blocks_per_cap;
    const uint32_t tid = threadIdx.x;
    const uint32_t base = blockIdx.x * kScanGroup + tid * 4;
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (base + i >= n) {
            v[i] = 0u;
        } else if (!p.tile_info) {
            v[i] = p.blk_count[base + i];
        } else {
            // block count from the wave tiles the front end already counted: changes
            // inside each tile + one per tile whose first bit differs from the last bit
            // of the tile before (nothing precedes a capture: level 0)
            const uint32_t b = base + i;
            const uint32_t cap = b / p.blocks_per_cap, blk = b - cap * p.blocks_per_cap;
            const uint32_t tpb = p.tiles_per_block;
            const uint32_t tile_bits = (uint32_t)(kBlockWords * 64) / tpb;
            const uint32_t *ti = p.tile_info + (uint64_t)cap * p.blocks_per_cap * tpb;
            const uint32_t t0 = blk * tpb;
            uint32_t prev_last = (t0 || p.has_prev) ? tile_live(*(ti + t0 - 1), p.stamp_bits) >> 31 : 0u;
            uint32_t c = 0;
            // a block's tiles are 4, 8 or 16 consecutive words: fetch them 16 B at a time
            for (uint32_t t4 = 0; t4 < tpb; t4 += 4) {
                const uint4 q = *reinterpret_cast<const uint4 *>(ti + t0 + t4);
                const uint32_t info4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) {
                    // a tile that holds no sample was not written by the front end; one without this
                    // run's stamp is a quiet tile (sparse output)
                    const bool live = (uint64_t)(t0 + t4 + j) * tile_bits < p.n_out;
                    const uint32_t info = live ? tile_live(info4[j], p.stamp_bits) : 0u;
                    c += live ? (info & 0x3ffu) + (((info >> 30) & 1u) ^ prev_last) : 0u;
                    prev_last = live ? info >> 31 : prev_last;
                }
            }
            v[i] = c;
        }
    }
    // (one 16-byte store per lane where the lists allow it: four dword stores per lane are four
    //  partial writes of every line)
    const bool quad = base + 3 < n;
    if (p.tile_info) {                  // edge_write skips empty blocks by the count
        if (quad && ((uintptr_t)(p.blk_count + base) & 15u) == 0) {
            *reinterpret_cast<uint4 *>(p.blk_count + base) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (base + i < n) p.blk_count[base + i] = v[i];
        }
    }
    const uint32_t mine = v[0] + v[1] + v[2] + v[3];
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if ((int)lane_id() >= d) inc += t;
    }
    if (lane_id() == 63) wave_sum[tid >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) before += wave_sum[w];
    uint32_t run = before + inc - mine;
    if (quad && ((uintptr_t)(p.blk_offset + base) & 15u) == 0) {
        *reinterpret_cast<uint4 *>(p.blk_offset + base) = make_uint4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
        run += mine;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (base + i < n) p.blk_offset[base + i] = run;
            run += v[i];
        }
    }
    if (tid == 255) p.group_total[blockIdx.x] = run;
}

// level 2: one workgroup scans the group totals in place (exclusive) and
// publishes the grand total in blk_offset[n].
__global__ __launch_bounds__(1024) void edge_scan_groups_kernel(const EdgeParams p) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t wtot[16];
    const uint32_t n = p.num_captures * p.blocks_per_cap;
    const uint32_t ng = (n + kScanGroup - 1) / kScanGroup;
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = (ng + 1023u) / 1024u;
    const uint32_t lo = min(tid * chunk, ng);
    const uint32_t hi = min(lo + chunk, ng);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += p.group_total[i];
    uint32_t total = 0;
    uint32_t run = wg_inclusive_sum(sum, wtot, &total) - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t t = p.group_total[i];
        p.group_total[i] = run;
        run += t;
    }
    if (tid == 1023) {
        p.blk_offset[n] = total;
        if ((uint64_t)total > p.edge_capacity) *p.overflow = 1;
        if (p.total_acc) atomicAdd(p.total_acc, total);
    }
}

// level 3 fused with the compaction: add the group base, keep the global
// prefix for later readers, write the positions.  A wavefront owns
// kWriteSpan (4) blocks, interleaved with the other wavefronts' (blocks with
// edges come in runs -- a message -- and would otherwise pile up on a few
// waves): their counts / offsets are fetched by its first lanes in one go,
// then only the blocks that hold edges (a minority: OOK is mostly constant
// level) get the 64-word treatment.
constexpr uint32_t kWriteSpan = 4;

__global__ __launch_bounds__(256) void edge_write_kernel(const EdgeParams p) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = lane_id();
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    const uint32_t nwaves = (total_blocks + kWriteSpan - 1) / kWriteSpan;
    if (wave >= nwaves) return;
    uint32_t off = 0, cnt = 0;
    if (lane < kWriteSpan && wave + lane * nwaves < total_blocks) {
        const uint32_t b = wave + lane * nwaves;
        off = p.blk_offset[b] + p.group_total[b / kScanGroup];
        p.blk_offset[b] = off;
        cnt = p.blk_count[b];
    }
    uint64_t todo = __ballot(cnt != 0);
    while (todo) {
        const uint32_t j = (uint32_t)__ffsll((long long)todo) - 1u;
        todo &= todo - 1;
        const uint32_t b = wave + j * nwaves;
        const uint32_t boff = rl(off, j);
        const uint32_t cap = b / p.blocks_per_cap;
        const uint32_t blk = b % p.blocks_per_cap;
        const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
        const uint64_t w = (uint64_t)blk * kBlockWords + lane;
        uint64_t e = change_word(words, w, p.n_out, p.has_prev);
        const uint32_t c = (uint32_t)__popcll(e);
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t v = __shfl_up(inc, d);
            if ((int)lane >= d) inc += v;
        }
        uint64_t at = (uint64_t)boff + (inc - c);
        while (e) {
            const int bit = __ffsll((long long)e) - 1;
            if (at < p.edge_capacity) p.edges[at] = w * 64 + (uint64_t)bit;
            ++at;
            e &= e - 1;
        }
    }
}

// The same behind the tuned front-end kernels, which leave a count of level changes per wave
// tile: ONE LANE per wave tile (round 3; one lane per 4096-bit block of 4, 8 or 16 tiles before).
// The lanes of a block sit next to each other in a wave: each reads the block's offset / count
// and its tile infos (the same 16-byte pieces: one request per block), adds up what the tiles in
// front of its own hold, and only a lane whose tile holds a change reads the tile's words (4, 8
// or 16 of them, one go) -- a capture is mostly constant level, so that is a few percent of the
// bit words, where edge_write_kernel reads every word of every block that holds an edge.
// Positions come out ascending: tiles in order, words in order, bits in order.  (A lane per block
// went through its loud tiles one after the other -- up to four dependent trips to memory where a
// message keeps every tile of every block of a wave busy: 50 us at 16 GiB.)
template <uint32_t TPB>
__global__ __launch_bounds__(256) void edge_write_tiles_kernel(const EdgeParams p) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr uint32_t tpb = TPB;           // wave tiles per 4096-bit block: 4, 8 or 16 (compile time: everything unrolls)
    const uint32_t b = gid / tpb, t = gid % tpb;
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (b >= total_blocks) return;
    const uint32_t cap = b / p.blocks_per_cap, blk = b - cap * p.blocks_per_cap;
    constexpr uint32_t words_per_tile = (uint32_t)kBlockWords / tpb;
    constexpr uint32_t tile_bits = words_per_tile * 64u;
    const uint32_t *ti = p.tile_info + (uint64_t)cap * p.blocks_per_cap * tpb;
    const uint64_t *words = p.bits + (uint64_t)cap * p.words_per_cap;
    const uint32_t t0 = blk * tpb;
    // one round trip: the block's offset, count and tile infos (and the info of the tile in front of the block)
    uint32_t infos[tpb];
    const uint32_t off_local = p.blk_offset[b], gbase = p.group_total[b / kScanGroup], cnt = p.blk_count[b];
    const uint32_t prev_info = (t0 || p.has_prev) ? *(ti + t0 - 1) : 0u;
    {
        const uint4 *ti4 = reinterpret_cast<const uint4 *>(ti + t0);
#pragma unroll
        for (uint32_t q = 0; q < tpb / 4; ++q) {
            const uint4 v = ti4[q];
            infos[4 * q + 0] = v.x;
            infos[4 * q + 1] = v.y;
            infos[4 * q + 2] = v.z;
            infos[4 * q + 3] = v.w;
        }
    }
    const uint32_t off = off_local + gbase;
    // (the block's lanes are neighbours in one wave: all of them have read the local offset before this store)
    if (t == 0) p.blk_offset[b] = off;
    if (cnt == 0) return;
    // what the tiles in front of this one hold, and the level in front of it
    uint32_t prev_last = (t0 || p.has_prev) ? tile_live(prev_info, p.stamp_bits) >> 31 : 0u;
    uint64_t at = off;
    uint32_t mine = 0, carry0 = 0, my_info = 0;
#pragma unroll
    for (uint32_t q = 0; q < tpb; ++q) {
        const bool live = (uint64_t)(t0 + q) * tile_bits < p.n_out;
        const uint32_t info = live ? tile_live(infos[q], p.stamp_bits) : 0u;
        const uint32_t c = live ? (info & 0x3ffu) + (((info >> 30) & 1u) ^ prev_last) : 0u;
        if (q < t) at += c;
        if (q == t) {
            mine = c;
            carry0 = prev_last;
            my_info = info;
        }
        prev_last = live ? info >> 31 : prev_last;
    }
    if (mine == 0) return;
    const uint64_t w0 = (uint64_t)(t0 + t) * words_per_tile;
    const uint32_t inner = my_info & 0x3ffu;
    if (inner <= 1u) {
        // The usual tile of a message: one pulse edge.  The change against the tile before sits at the tile's first
        // bit and needs no word at all; ONE change inside is the first bit of word `widx` that differs from the
        // tile's first bit (everything in front of the first change equals it) -- 8 bytes instead of the tile's
        // 128: reading every loud tile whole was 94 MB of 128-byte pieces per 16 GiB capture, most of this
        // kernel's 55 us.
        if (mine != inner) {
            if (at < p.edge_capacity) p.edges[at] = w0 * 64;
            ++at;
        }
        if (inner) {
            const uint32_t widx = (my_info >> kTileWordShift) & 15u;
            const uint64_t cur = words[w0 + widx];
            const uint64_t x = cur ^ (((my_info >> 30) & 1u) ? ~0ull : 0ull);
            if (at < p.edge_capacity) p.edges[at] = (w0 + widx) * 64 + (uint64_t)(__ffsll((long long)x) - 1);
        }
        return;
    }
    // the tile's words (4, 8 or 16 of them), requested together
    uint4 wv[words_per_tile / 2];
    const uint4 *w4 = reinterpret_cast<const uint4 *>(words + w0);
#pragma unroll
    for (uint32_t q = 0; q < words_per_tile / 2; ++q) wv[q] = w4[q];
    uint64_t carry = carry0;
#pragma unroll
    for (uint32_t i = 0; i < words_per_tile; ++i) {
        const uint4 v = wv[i >> 1];
        const uint64_t cur = (i & 1u) ? ((uint64_t)v.w << 32) | v.z : ((uint64_t)v.y << 32) | v.x;
        uint64_t e = cur ^ ((cur << 1) | carry);
        carry = cur >> 63;
        const uint64_t base = (w0 + i) * 64;
        if (base + 64 > p.n_out) e &= base >= p.n_out ? 0ull : ((1ull << (p.n_out - base)) - 1ull);
        while (e) {
            const int bit = __ffsll((long long)e) - 1;
            if (at < p.edge_capacity) p.edges[at] = base + (uint64_t)bit;
            ++at;
            e &= e - 1;
        }
    }
}

// ---------------------------------------------------------------------------
// symbol state machine
// ---------------------------------------------------------------------------
//
// Integer restatement of handle_rx_triggers / process (state_machine.c:
// 421-539).  `k` = number of elapsed_us increments since it was last zeroed;
// the host turned every duration window / timeout into a range of k by
// replaying the reference's double accumulation, so the comparisons below
// ARE the reference's float comparisons.  Counters are 32 bit and saturate
// at kSat; the host refuses devices with a finite bound above 2^31, so a
// saturated counter is "beyond every bound" exactly like the real one.
//
// Execution model: one wave per segment, control flow and state machine
// state wave-uniform (SGPRs); the tables live in VGPRs, lane i holding
// trigger i and state i, so "first trigger in file order that matches" is a
// ballot + find-first-set and a table row is a v_readlane.

enum { kCondAlways = 1, kCondPulseStart, kCondPulseEnd, kCondTimeout, kCondMsgComplete };
enum { kActNone = 1, kActAppend0, kActAppend1, kActOutput };
enum { kResError = -1, kResNone = 0, kResOutput = 1 };

constexpr uint32_t kNone = 0xffffffffu;     // "no bound"
constexpr uint32_t kSat = 0xfffffffeu;      // counters saturate here (< kNone)

struct LaneTables {             // per-lane copies
    uint32_t tkmin, tkmax, tinfo;       // trigger `lane`: cond | action << 8 | next << 16
    uint32_t skmin, skmax, skto, srow;  // state `lane`: tbeg | tend << 8 | flags << 16
    uint32_t max_bits;                  // wave-uniform
};

struct Fsm {                    // wave-uniform
    uint32_t cur, nbits, prev, k;
    uint64_t d0, d1, d2, d3, d4;
};

__device__ __forceinline__ uint32_t sat_add(uint32_t k, uint64_t m) {
    const uint64_t s = (uint64_t)k + m;
    return s > kSat ? kSat : (uint32_t)s;
}

// Branch-free so the payload words stay in scalar registers (an address-taken
// or dynamically indexed payload ends up in scratch memory).
__device__ __forceinline__ uint64_t put_bit(uint64_t d, bool here, uint64_t m, bool one) {
    const uint64_t v = one ? (d | m) : (d & ~m);
    return here ? v : d;
}

__device__ __forceinline__ void set_bit(Fsm &f, uint32_t idx, bool one) {
    const uint32_t wi = idx >> 6;
    const uint64_t m = 1ull << (idx & 63u);
    f.d0 = put_bit(f.d0, wi == 0, m, one);
    f.d1 = put_bit(f.d1, wi == 1, m, one);
    f.d2 = put_bit(f.d2, wi == 2, m, one);
    f.d3 = put_bit(f.d3, wi == 3, m, one);
    f.d4 = put_bit(f.d4, wi == 4, m, one);
}

// k has no influence in states flagged so; pin it to 0 there so that two
// trajectories that differ only in how long they idled compare equal.
__device__ __forceinline__ void canon(const LaneTables &t, Fsm &f) {
    if (rl(t.srow, f.cur) & 0x10000u) f.k = 0;
}

// What happens once trigger `fired` of state s has matched with counter k
// (state_machine.c:483-511).
__device__ __forceinline__ int fsm_fire(const LaneTables &t, Fsm &f, uint32_t s, uint32_t fired) {
    const uint32_t info = rl(t.tinfo, fired);
    const uint32_t fc = info & 0xffu, act = (info >> 8) & 0xffu, next = info >> 16;
    int result = kResNone;
    bool ok = true;
    if (fc == kCondPulseStart || fc == kCondPulseEnd) {     // :100-117
        ok = f.k >= rl(t.skmin, s) && f.k <= rl(t.skmax, s);
    }
    if (ok) {
        if (act == kActAppend0 || act == kActAppend1) {
            // :365-385 stores while num_bits <= max_bits, always counts
            if (f.nbits <= t.max_bits) set_bit(f, f.nbits, act == kActAppend1);
            f.nbits += 1;
        } else if (act == kActOutput) {
            result = kResOutput;
        }
        f.cur = next;
    } else {
        result = kResError;
        f.cur = 0;                                          // :505-509
    }
    f.k = 0;                                                // :511
    return result;
}

// Per-lane: does my trigger match on a sample with counter k, previous level
// prev and level b?  (state_machine.c:430-481)
__device__ __forceinline__ bool trig_match(const LaneTables &t, uint32_t row, uint32_t kto, uint32_t k,
                                           uint32_t prev, uint32_t b, uint32_t nbits) {
    const uint32_t lane = lane_id();
    const uint32_t cond = t.tinfo & 0xffu;
    const bool in_row = lane >= (row & 0xffu) && lane < ((row >> 8) & 0xffu);
    bool c = cond == kCondAlways;
    c = c || (cond == kCondPulseStart && !prev && b);
    c = c || (cond == kCondPulseEnd && prev && !b);
    c = c || (cond == kCondTimeout && k >= kto);            // kto = kNone when no timeout
    c = c || (cond == kCondMsgComplete && nbits >= t.max_bits);
    return in_row && k >= t.tkmin && k <= t.tkmax && c;     // :119-133
}

// state_machine.c:421-519: one evaluation of the current state's triggers.
__device__ __forceinline__ int fsm_eval(const LaneTables &t, Fsm &f, uint32_t b) {
    const uint32_t s = f.cur;
    const uint32_t row = rl(t.srow, s), kto = rl(t.skto, s);
    const uint64_t ball = __ballot(trig_match(t, row, kto, f.k, f.prev, b, f.nbits));
    if (ball == 0) {
        f.k = sat_add(f.k, 1);                              // :513-515
        return kResNone;
    }
    return fsm_fire(t, f, s, (uint32_t)__builtin_ctzll(ball));     // first in file order
}

// state_machine.c:521-539: reset clears the payload and is evaluated, then
// the (possibly new) state is evaluated on the same sample.
__device__ __forceinline__ int fsm_step(const LaneTables &t, Fsm &f, uint32_t b) {
    if (f.cur == 0) {
        f.nbits = 0;
        f.d0 = f.d1 = f.d2 = f.d3 = f.d4 = 0;   // memset(data, 0, (max_bits+7)/8)
        const int r = fsm_eval(t, f, b);
        if (r != kResNone) return r;
    }
    return fsm_eval(t, f, b);
}

// Per-lane: with the input level constant (pulse triggers cannot fire), how
// many evaluations from now (at k, k+1, ...) until my trigger fires?
// kNone = never.
__device__ __forceinline__ uint32_t trig_wait(const LaneTables &t, uint32_t row, uint32_t kto, uint32_t k,
                                              uint32_t nbits) {
    const uint32_t lane = lane_id();
    const uint32_t cond = t.tinfo & 0xffu;
    bool can = lane >= (row & 0xffu) && lane < ((row >> 8) & 0xffu);
    uint32_t lo = t.tkmin;
    if (cond == kCondTimeout) {
        can = can && kto != kNone;
        lo = lo > kto ? lo : kto;
    } else if (cond == kCondMsgComplete) {
        can = can && nbits >= t.max_bits;
    } else {
        can = can && cond == kCondAlways;
    }
    const uint32_t first = k > lo ? k : lo;
    can = can && first <= t.tkmax && first <= kSat;
    return can ? first - k : kNone;
}

// wave minimum of a per-lane u32
__device__ __forceinline__ uint32_t wave_min(uint32_t w) {
#define OOKD_ROW_MIN(SH)                                                                              \
    {                                                                                                 \
        const uint32_t o =                                                                            \
            (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)w, 0x110 + (SH), 0xf, 0xf, false); \
        w = o < w ? o : w;                                                                            \
    }
    OOKD_ROW_MIN(1)
    OOKD_ROW_MIN(2)
    OOKD_ROW_MIN(4)
    OOKD_ROW_MIN(8)
#undef OOKD_ROW_MIN
    const uint32_t a = rl(w, 15), b = rl(w, 31), c = rl(w, 47), d = rl(w, 63);
    const uint32_t m1 = a < b ? a : b, m2 = c < d ? c : d;
    return m1 < m2 ? m1 : m2;
}

__device__ __forceinline__ uint64_t payload_word(uint64_t v, uint32_t i, uint32_t nbytes) {
    if (8 * i >= nbytes) return 0;
    if (8 * (i + 1) > nbytes) return v & ((1ull << ((nbytes - 8 * i) * 8)) - 1ull);
    return v;
}

__device__ __forceinline__ uint32_t clamp32(uint64_t v) {
    return v == ~0ull ? kNone : (uint32_t)v;    // host guarantees finite bounds < 2^31
}

__device__ __forceinline__ LaneTables load_tables(const FsmTablesDev *g) {
    LaneTables t;
    const uint32_t l = lane_id();
    t.tkmin = clamp32(g->trig_kmin[l]);
    t.tkmax = clamp32(g->trig_kmax[l]);
    t.tinfo = g->trig_info[l];
    t.skmin = clamp32(g->state_kmin[l]);
    t.skmax = clamp32(g->state_kmax[l]);
    t.skto = clamp32(g->state_kto[l]);
    t.srow = (g->state_tbeg[l] & 0xffu) | ((g->state_tend[l] & 0xffu) << 8) | ((g->state_flags[l] & 1u) << 16);
    t.max_bits = rfl(g->max_bits);
    return t;
}

// ---- devices with more than 64 states / triggers: the same evaluation with the tables in LDS ----------
// (kernels.hpp: kMaxStatesBig).  A state's triggers are looked at 64 at a time, lane l taking trigger
// begin + 64 c + l: "first in file order" is the first set bit of the first chunk that has one.
struct BigTab {
    const uint32_t *st;         // [states][6]
    const uint32_t *tr;         // [triggers][3]
    uint32_t max_bits;
};

__device__ __forceinline__ bool big_cond(uint32_t cond, uint32_t kto, uint32_t k, uint32_t prev, uint32_t b,
                                         uint32_t nbits, uint32_t max_bits) {
    bool c = cond == kCondAlways;
    c = c || (cond == kCondPulseStart && !prev && b);
    c = c || (cond == kCondPulseEnd && prev && !b);
    c = c || (cond == kCondTimeout && k >= kto);
    c = c || (cond == kCondMsgComplete && nbits >= max_bits);
    return c;
}

// index of the first trigger of state s that matches, or -1 (wave-uniform)
__device__ __forceinline__ int big_match(const BigTab &t, uint32_t s, uint32_t k, uint32_t prev, uint32_t b, uint32_t nbits) {
    const uint32_t *row = t.st + kBigStateWords * s;
    const uint32_t kto = row[2], beg = row[3], end = row[4];
    for (uint32_t base = beg; base < end; base += 64) {
        const uint32_t i = base + lane_id();
        bool m = false;
        if (i < end) {
            const uint32_t *tg = t.tr + kBigTrigWords * i;
            m = k >= tg[0] && k <= tg[1] && big_cond(tg[2] & 0xffu, kto, k, prev, b, nbits, t.max_bits);
        }
        const uint64_t ball = __ballot(m);
        if (ball) return (int)(base + (uint32_t)__builtin_ctzll(ball));
    }
    return -1;
}

// with the level constant: evaluations from now until a trigger of state s fires (kNone = never), wave-uniform
__device__ __forceinline__ uint32_t big_wait(const BigTab &t, uint32_t s, uint32_t k, uint32_t nbits) {
    const uint32_t *row = t.st + kBigStateWords * s;
    const uint32_t kto = row[2], beg = row[3], end = row[4];
    uint32_t best = kNone;
    for (uint32_t base = beg; base < end; base += 64) {
        const uint32_t i = base + lane_id();
        uint32_t w = kNone;
        if (i < end) {
            const uint32_t *tg = t.tr + kBigTrigWords * i;
            const uint32_t cond = tg[2] & 0xffu;
            bool can = true;
            uint32_t lo = tg[0];
            if (cond == kCondTimeout) {
                can = kto != kNone;
                lo = lo > kto ? lo : kto;
            } else if (cond == kCondMsgComplete) {
                can = nbits >= t.max_bits;
            } else {
                can = cond == kCondAlways;
            }
            const uint32_t first = k > lo ? k : lo;
            can = can && first <= tg[1] && first <= kSat;
            w = can ? first - k : kNone;
        }
        const uint32_t m = wave_min(w);
        best = m < best ? m : best;
    }
    return best;
}

__device__ __forceinline__ int big_fire(const BigTab &t, Fsm &f, uint32_t s, uint32_t fired) {
    const uint32_t info = t.tr[kBigTrigWords * fired + 2];
    const uint32_t fc = info & 0xffu, act = (info >> 8) & 0xffu, next = info >> 16;
    int result = kResNone;
    bool ok = true;
    if (fc == kCondPulseStart || fc == kCondPulseEnd) {
        ok = f.k >= t.st[kBigStateWords * s] && f.k <= t.st[kBigStateWords * s + 1];
    }
    if (ok) {
        if (act == kActAppend0 || act == kActAppend1) {
            if (f.nbits <= t.max_bits) set_bit(f, f.nbits, act == kActAppend1);
            f.nbits += 1;
        } else if (act == kActOutput) {
            result = kResOutput;
        }
        f.cur = next;
    } else {
        result = kResError;
        f.cur = 0;
    }
    f.k = 0;
    return result;
}

__device__ __forceinline__ int big_eval(const BigTab &t, Fsm &f, uint32_t b) {
    const uint32_t s = f.cur;
    const int m = big_match(t, s, f.k, f.prev, b, f.nbits);
    if (m < 0) {
        f.k = sat_add(f.k, 1);
        return kResNone;
    }
    return big_fire(t, f, s, (uint32_t)m);
}

__device__ __forceinline__ int big_step(const BigTab &t, Fsm &f, uint32_t b) {
    if (f.cur == 0) {
        f.nbits = 0;
        f.d0 = f.d1 = f.d2 = f.d3 = f.d4 = 0;
        const int r = big_eval(t, f, b);
        if (r != kResNone) return r;
    }
    return big_eval(t, f, b);
}

__device__ __forceinline__ void big_canon(const BigTab &t, Fsm &f) {
    if (t.st[kBigStateWords * f.cur + 5] & 1u) f.k = 0;
}

// First decimated index of input buffer `buf`: floor(buf * spb / D)
// (decimated sample j comes from input D*(j+1)-1).
__device__ __forceinline__ uint64_t buffer_start(uint64_t buf, uint32_t spb, uint32_t D) {
    return (buf * (uint64_t)spb) / D;       // buf*spb ~ input samples, fits 64 bits
}

__device__ __forceinline__ uint32_t bit_at(const uint64_t *words, uint64_t i) {
    return (uint32_t)((words[i >> 6] >> (i & 63)) & 1ull);
}

// ---- segment boundaries -------------------------------------------------------
//
// A capture is cut every ~seg_len samples, but each cut is moved (within
// +-seg_len/2) to the rising edge that ends the longest low-level gap in
// that window: after a long gap the true state machine has almost certainly
// timed out into its quiet state, which is what a segment assumes about its
// incoming state.  The choice only affects how many fix-point rounds are
// needed, never the result.
__global__ __launch_bounds__(64) void fsm_cuts_kernel(const FsmParams p) {
    if (p.edge_overflow && *p.edge_overflow) return;        // blk_offset counts edges that were never written
    const uint32_t id = blockIdx.x;                 // capture * (segs+1) + boundary
    const uint32_t per = p.segs_per_cap + 1;
    const uint32_t cap = id / per, bi = id % per;
    uint64_t *bounds = p.seg_bounds + (size_t)cap * per;
    const uint32_t lane = lane_id();
    if (bi == 0 || bi == p.segs_per_cap) {
        if (lane == 0) bounds[bi] = bi == 0 ? 0 : p.n_out;
        return;
    }
    const uint64_t nominal = (uint64_t)bi * p.seg_len;
    const uint64_t half = p.seg_len / 2;
    const uint64_t w0 = nominal - half, w1 = min(nominal + half, p.n_out);
    const uint32_t blk0 = cap * p.blocks_per_cap;
    const uint64_t cap_e0 = p.blk_offset[blk0];
    const uint64_t ne = (uint64_t)p.blk_offset[blk0 + p.blocks_per_cap] - cap_e0;
    const uint64_t *edges = p.edges + cap_e0;
    const uint64_t b0 = min(w0 >> 12, (uint64_t)p.blocks_per_cap - 1);
    const uint64_t b1 = min((w1 >> 12) + 1, (uint64_t)p.blocks_per_cap);
    const uint64_t i0 = (uint64_t)p.blk_offset[blk0 + b0] - cap_e0;
    const uint64_t i1 = (uint64_t)p.blk_offset[blk0 + b1] - cap_e0;
    uint64_t best_gap = 0, best_cut = 0;
    for (uint64_t i = i0 + lane; i + 1 < ne && i < i1; i += 64) {
        // level after edge i is (i+1)&1: low after odd-indexed edges
        if ((i & 1ull) == 0) continue;
        const uint64_t a = edges[i], b = edges[i + 1];
        if (a < w0 || b >= w1) continue;
        if (b - a > best_gap) {
            best_gap = b - a;
            best_cut = b;
        }
    }
    // wave argmax (ties: smallest cut)
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t og = __shfl_xor(best_gap, d), oc = __shfl_xor(best_cut, d);
        if (og > best_gap || (og == best_gap && og != 0 && oc < best_cut)) {
            best_gap = og;
            best_cut = oc;
        }
    }
    if (lane == 0) bounds[bi] = best_gap ? best_cut : min(nominal, p.n_out);
}

__global__ __launch_bounds__(64) void fsm_prepare_kernel(const FsmParams p, const FsmStateDev first,
                                                         int have_first) {
    if (p.edge_overflow && *p.edge_overflow) return;
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (seg >= nseg) return;
    const uint32_t cap = seg / p.segs_per_cap;
    const uint32_t ls = seg % p.segs_per_cap;
    SegState ss;
    ss.st.cur = 0;
    ss.st.nbits = 0;
    ss.st.k = 0;
    ss.st.prev = 0;
    ss.st.pad = 0;
    for (int i = 0; i < kPayloadWords; ++i) ss.st.data[i] = 0;
    ss.skip_to = 0;
    ss.pad = 0;
    if (ls == 0) {
        if (have_first) ss.st = first;
    } else {
        const uint64_t start = p.seg_bounds[(size_t)cap * (p.segs_per_cap + 1) + ls];
        if (start > 0 && start <= p.n_out) {
            ss.st.prev = fsm_level_at(p, cap, (int64_t)start - 1);
        }
        // assume the quiet state when the level before the segment is low
        if (ss.st.prev == 0) ss.st.cur = p.tables->quiet_state;
    }
    p.state_in[seg] = ss;
    p.seg_msg_count[seg] = 0;
    p.seg_err_count[seg] = 0;
}

__device__ __forceinline__ bool seg_state_equal(const SegState &a, const SegState &b) {
    bool eq = a.st.cur == b.st.cur && a.st.nbits == b.st.nbits && a.st.k == b.st.k &&
              a.st.prev == b.st.prev && a.skip_to == b.skip_to;
    for (int i = 0; i < kPayloadWords; ++i) eq = eq && a.st.data[i] == b.st.data[i];
    return eq;
}

template <bool BIG>
__global__ __launch_bounds__(64) void fsm_round_kernel(const FsmParams p, uint32_t parity, uint32_t mode,
                                                       uint32_t slot) {
    extern __shared__ __attribute__((aligned(16))) uint32_t big_lds[];
    if (p.edge_overflow && *p.edge_overflow) return;        // blk_offset counts edges that were never written
    const uint32_t seg = blockIdx.x;
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    const uint32_t cap = seg / p.segs_per_cap;
    const uint32_t ls = seg % p.segs_per_cap;
    const uint32_t lane = lane_id();
    const uint32_t par = parity & 1u;
    SegState *out_cur = p.state_out + (size_t)par * nseg;
    const SegState *out_prev = p.state_out + (size_t)(par ^ 1u) * nseg;

    // a finished fix-point makes the remaining rounds of a batch no-ops
    if (mode == 1 && slot > 0 && p.changed[slot - 1] == 0) {
        if (lane == 0) out_cur[seg] = out_prev[seg];
        return;
    }
    if (mode != 0) {
        bool rerun = false;
        if (ls != 0) {
            const SegState nin = out_prev[seg - 1];
            rerun = !seg_state_equal(nin, p.state_in[seg]);
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
    } else if (lane == 0) {
        atomicAdd(&p.changed[slot], 1u);            // a full round counts every segment
    }

    LaneTables t{};
    BigTab bt{};
    if constexpr (BIG) {
        for (uint32_t i = lane; i < p.big_words; i += 64) big_lds[i] = p.big[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        bt.st = big_lds + kBigHeaderWords;
        bt.tr = bt.st + kBigStateWords * big_lds[0];
        bt.max_bits = big_lds[1];
        t.max_bits = bt.max_bits;
    } else {
        t = load_tables(p.tables);
    }
    const uint64_t *bounds = p.seg_bounds + (size_t)cap * (p.segs_per_cap + 1);
    const uint64_t seg_start = rfl64(bounds[ls]);
    const uint64_t seg_end = rfl64(bounds[ls + 1]);
    const uint32_t seg_n = (uint32_t)(seg_end - seg_start);     // host keeps segments < 2^31 samples
    const uint32_t blk0 = cap * p.blocks_per_cap;
    const uint64_t cap_e0 = rfl(p.blk_offset[blk0]);
    const uint64_t ne = (uint64_t)rfl(p.blk_offset[blk0 + p.blocks_per_cap]) - cap_e0;
    const uint64_t *edges = p.edges + cap_e0;

    const SegState sin = p.state_in[seg];
    Fsm f;
    f.cur = rfl(sin.st.cur);
    f.nbits = rfl(sin.st.nbits);
    f.prev = rfl(sin.st.prev);
    {
        const uint64_t k64 = rfl64(sin.st.k);
        f.k = k64 > kSat ? kSat : (uint32_t)k64;
    }
    f.d0 = rfl64(sin.st.data[0]);
    f.d1 = rfl64(sin.st.data[1]);
    f.d2 = rfl64(sin.st.data[2]);
    f.d3 = rfl64(sin.st.data[3]);
    f.d4 = rfl64(sin.st.data[4]);
    const uint64_t skip_to = rfl64(sin.skip_to);
    uint64_t skip_out = 0;
    // positions inside the segment are 32-bit offsets from seg_start
    uint32_t pos = 0;
    if (skip_to > seg_start) {
        if (skip_to >= seg_end) {
            pos = seg_n;                            // whole segment skipped
            skip_out = skip_to;
        } else {
            pos = (uint32_t)(skip_to - seg_start);
        }
    }

    // first edge at or after the start position (capture-local index): count
    // the edges below it from the start of its 4096-sample block
    uint64_t ci = 0;
    if (pos < seg_n) {
        const uint64_t apos = seg_start + pos;
        const uint64_t blk = min(apos >> 12, (uint64_t)p.blocks_per_cap - 1);
        uint64_t i = (uint64_t)rfl(p.blk_offset[blk0 + blk]) - cap_e0;
        const uint64_t iend = (uint64_t)rfl(p.blk_offset[blk0 + blk + 1]) - cap_e0;
        ci = i;
        for (; i < iend; i += 64) {
            const uint64_t idx = i + lane;
            const bool lt = idx < iend && edges[idx] < apos;
            ci += (uint64_t)__popcll(__ballot(lt));
        }
    }

    MsgDev *msgs = p.seg_msgs + (size_t)seg * p.msg_slots;
    uint64_t *errs = p.seg_errs + (size_t)seg * p.err_slots;
    uint32_t nmsg = 0, nerr = 0, flags = 0;

    // edge window: lane i holds the offset of edge wbase + i (kNone past the
    // segment or the list); every in-segment edge offset is < seg_n
    uint64_t wbase = ci;
    uint32_t ev;
#define OOKD_LOAD_WINDOW()                                                          \
    {                                                                               \
        const uint64_t idx_ = wbase + lane;                                         \
        const uint64_t e_ = idx_ < ne ? edges[idx_] : ~0ull;                        \
        ev = (e_ >= seg_end) ? kNone : (uint32_t)(e_ - seg_start);                  \
    }
    OOKD_LOAD_WINDOW()

    uint32_t dbg_turns = 0, dbg_fused = 0, dbg_loads = 1;
    const uint64_t dbg_t0 = p.debug ? __builtin_amdgcn_s_memtime() : 0;
    while (pos < seg_n) {
        dbg_turns++;
        if (ci + 1 >= wbase + 64 || ci < wbase) {   // needs entries ci and ci+1
            wbase = ci;
            dbg_loads++;
            OOKD_LOAD_WINDOW()
        }
        const uint32_t wi = (uint32_t)(ci - wbase);
        const uint32_t e0 = rl(ev, wi);
        const bool at_edge = (e0 == pos);
        const uint32_t wia = wi + (at_edge ? 1u : 0u);
        const uint64_t cia = ci + (at_edge ? 1 : 0);
        const uint32_t b = (uint32_t)(cia & 1ull);  // level = parity of edges <= pos
        int r;
        uint32_t at;                                // offset of the sample that produced r
        if (b == f.prev) {
            // ---- constant level up to the next edge (or the segment end) ----------
            const uint32_t e1 = rl(ev, wia);
            const uint32_t run_end = e1 < seg_n ? e1 : seg_n;
            const uint32_t n = run_end - pos;       // >= 1 samples with level b
            const uint32_t s = f.cur;
            uint32_t row = 0, kto = 0, wait = 0, wmin = kNone;
            if constexpr (BIG) {
                wmin = big_wait(bt, s, f.k, f.nbits);
            } else {
                row = rl(t.srow, s);
                kto = rl(t.skto, s);
                wait = trig_wait(t, row, kto, f.k, f.nbits);
            }
            if (s != 0) {
                // one evaluation per sample: any always/timeout/msg_complete
                // trigger due inside the run?
                if (BIG ? !(wmin < n) : __ballot(wait < n) == 0) {
                    f.k = sat_add(f.k, n);
                    if (e1 >= seg_n) {              // ran into the end of the segment
                        pos = seg_n;
                        ci = cia;
                        if constexpr (BIG) big_canon(bt, f);
                        else canon(t, f);
                        continue;
                    }
                    // the edge sample itself: level flips, prev = b
                    const uint32_t b2 = b ^ 1u;
                    int fired;
                    if constexpr (BIG) {
                        fired = big_match(bt, s, f.k, b, b2, f.nbits);
                    } else {
                        const uint64_t ball = __ballot(trig_match(t, row, kto, f.k, b, b2, f.nbits));
                        fired = ball ? (int)__builtin_ctzll(ball) : -1;
                    }
                    pos = e1 + 1;
                    ci = cia + 1;
                    f.prev = b2;
                    if (fired < 0) {                // edge ignored by this state
                        f.k = sat_add(f.k, 1);
                        if constexpr (BIG) big_canon(bt, f);
                        else canon(t, f);
                        continue;
                    }
                    r = BIG ? big_fire(bt, f, s, (uint32_t)fired) : fsm_fire(t, f, s, (uint32_t)fired);
                    at = e1;
                    dbg_fused++;
                } else {
                    const uint32_t w = BIG ? wmin : wave_min(wait);      // < n
                    f.k = sat_add(f.k, w);
                    pos += w;
                    ci = cia;
                    r = BIG ? big_step(bt, f, b) : fsm_step(t, f, b);          // fires
                    at = pos;
                    pos += 1;
                }
            } else {
                // reset evaluates twice per sample (state_machine.c:526-538)
                const uint32_t q = BIG ? wmin : wave_min(wait);
                uint32_t m = q == kNone ? n : (q >> 1);
                if (m > n) m = n;
                if (m > 0) {
                    f.k = sat_add(f.k, 2ull * m);
                    pos += m;
                    ci = cia;
                    if constexpr (BIG) big_canon(bt, f);
                    else canon(t, f);
                    continue;
                }
                r = BIG ? big_step(bt, f, b) : fsm_step(t, f, b);
                at = pos;
                pos += 1;
                ci = cia;
            }
        } else {
            // ---- the state machine sees a level change at pos -----------------------
            r = BIG ? big_step(bt, f, b) : fsm_step(t, f, b);
            f.prev = b;                             // sm_process: prev_bit = data[i]
            at = pos;
            pos += 1;
            ci = cia;
        }
        if constexpr (BIG) big_canon(bt, f);
        else canon(t, f);
        if (r == kResOutput) {
            if (nmsg < p.msg_slots) {
                if (lane == 0) {
                    MsgDev mm;
                    mm.capture = cap;
                    mm.reserved = 0;
                    mm.sample = seg_start + at;
                    // the first (max_bits+7)/8 bytes are the message
                    const uint32_t nbytes = (t.max_bits + 7u) >> 3;
                    mm.payload[0] = payload_word(f.d0, 0, nbytes);
                    mm.payload[1] = payload_word(f.d1, 1, nbytes);
                    mm.payload[2] = payload_word(f.d2, 2, nbytes);
                    mm.payload[3] = payload_word(f.d3, 3, nbytes);
                    msgs[nmsg] = mm;
                }
            } else {
                flags |= 1u;
            }
            nmsg++;
        } else if (r == kResError) {
            const uint64_t apos = seg_start + at;
            if (nerr < p.err_slots && lane == 0) errs[nerr] = apos;
            nerr++;
            // device.c:646: the rest of this buffer is never fed in
            const uint64_t in_idx = (uint64_t)p.total_decim * (apos + 1) - 1;
            const uint64_t buf = in_idx / p.spb;
            uint64_t nb = buffer_start(buf + 1, p.spb, p.total_decim);
            if (nb <= apos) nb = apos + 1;
            if (nb >= seg_end) {
                skip_out = nb;
                pos = seg_n;
                break;
            }
            pos = (uint32_t)(nb - seg_start);
            // first edge at or after pos
            for (;;) {
                if (ci >= wbase + 64 || ci < wbase) {
                    wbase = ci;
                    OOKD_LOAD_WINDOW()
                }
                const uint32_t from = (uint32_t)(ci - wbase);
                const uint64_t bal = __ballot(lane >= from && ev < pos);
                const uint32_t cnt = (uint32_t)__popcll(bal);
                ci += cnt;
                if (from + cnt < 64) break;         // stopped inside the window
            }
        }
    }
#undef OOKD_LOAD_WINDOW

    if (p.debug && lane == 0) {
        p.debug[4 * (size_t)seg + 0] = dbg_turns;
        p.debug[4 * (size_t)seg + 1] = __builtin_amdgcn_s_memtime() - dbg_t0;
        p.debug[4 * (size_t)seg + 2] = dbg_fused;
        p.debug[4 * (size_t)seg + 3] = dbg_loads;
    }
    if (lane == 0) {
        SegState so;
        so.st.cur = f.cur;
        so.st.nbits = f.nbits;
        so.st.k = f.k;
        so.st.prev = f.prev;
        so.st.pad = 0;
        so.st.data[0] = f.d0;
        so.st.data[1] = f.d1;
        so.st.data[2] = f.d2;
        so.st.data[3] = f.d3;
        so.st.data[4] = f.d4;
        so.skip_to = skip_out;
        so.pad = 0;
        out_cur[seg] = so;
        p.seg_msg_count[seg] = nmsg;
        p.seg_err_count[seg] = nerr;
        if (flags) atomicOr(p.flags, flags);
    }
}

// Compacts per-segment messages into one list (single workgroup).
__global__ __launch_bounds__(1024) void fsm_gather_kernel(const FsmParams p) {
    if (p.edge_overflow && *p.edge_overflow) return;
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
        const uint32_t v = (tid >= d) ? part[tid - d] : 0u;
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
// publish: results straight into pinned host memory
// ---------------------------------------------------------------------------
// One small kernel instead of three device-to-host copy commands: the result
// header (plus the edge total, which lives at the end of the block offsets)
// and the first messages are written through the host-mapped pointers, then
// the device header is zeroed for the next run.
__global__ __launch_bounds__(256) void publish_kernel(PublishParams p) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint64_t s_total;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_total = p.d_hdr[p.totals_word] | ((uint64_t)p.d_hdr[p.totals_word + 1] << 32);
    __syncthreads();
    if (p.d_msgs) {
        const uint64_t n = (s_total < p.first_msgs ? s_total : p.first_msgs) * (sizeof(MsgDev) / 16);
        for (uint64_t i = tid; i < n; i += blockDim.x) p.h_msgs[i] = p.d_msgs[i];
    }
    if (tid < p.hdr_words) {
        uint32_t v = p.d_hdr[tid];
        if (tid == p.edges_word && p.total_edges) v = *p.total_edges;
        p.h_hdr[tid] = v;
    }
    __syncthreads();
    if (tid < p.hdr_words) p.d_hdr[tid] = 0;
}

hipError_t launch_publish(const PublishParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------

hipError_t launch_edges(const EdgeParams &p, hipStream_t stream) {
    const uint32_t total_blocks = p.num_captures * p.blocks_per_cap;
    if (total_blocks == 0) return hipSuccess;
    const uint32_t wgs = (total_blocks + 3) / 4;        // 4 waves per workgroup
    const uint32_t groups = (total_blocks + kScanGroup - 1) / kScanGroup;
    if (!p.tile_info) hipLaunchKernelGGL(edge_count_kernel, dim3(wgs), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(edge_scan_local_kernel, dim3(groups), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(edge_scan_groups_kernel, dim3(1), dim3(1024), 0, stream, p);
    if (p.tile_info) {
        const dim3 grid((uint32_t)(((uint64_t)total_blocks * p.tiles_per_block + 255) / 256));      // a lane per tile
        if (p.tiles_per_block == 4) hipLaunchKernelGGL(edge_write_tiles_kernel<4>, grid, dim3(256), 0, stream, p);
        else if (p.tiles_per_block == 8) hipLaunchKernelGGL(edge_write_tiles_kernel<8>, grid, dim3(256), 0, stream, p);
        else if (p.tiles_per_block == 16) hipLaunchKernelGGL(edge_write_tiles_kernel<16>, grid, dim3(256), 0, stream, p);
        else return hipErrorInvalidValue;
    } else {
        const uint32_t write_waves = (total_blocks + kWriteSpan - 1) / kWriteSpan;
        hipLaunchKernelGGL(edge_write_kernel, dim3((write_waves + 3) / 4), dim3(256), 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_fsm_prepare(const FsmParams &p, const FsmStateDev *first_state, hipStream_t stream) {
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (nseg == 0) return hipSuccess;
    FsmStateDev first{};
    if (first_state) first = *first_state;
    hipLaunchKernelGGL(fsm_cuts_kernel, dim3(p.num_captures * (p.segs_per_cap + 1)), dim3(64), 0, stream, p);
    hipLaunchKernelGGL(fsm_prepare_kernel, dim3((nseg + 63) / 64), dim3(64), 0, stream, p, first,
                       first_state ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_fsm_round(const FsmParams &p, uint32_t parity, uint32_t mode, uint32_t slot,
                            hipStream_t stream) {
    const uint32_t nseg = p.num_captures * p.segs_per_cap;
    if (nseg == 0) return hipSuccess;
    if (p.big) {
        const size_t lds = (size_t)p.big_words * sizeof(uint32_t);
        const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(&fsm_round_kernel<true>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(fsm_round_kernel<true>, dim3(nseg), dim3(64), lds, stream, p, parity, mode, slot);
    } else {
        hipLaunchKernelGGL(fsm_round_kernel<false>, dim3(nseg), dim3(64), 0, stream, p, parity, mode, slot);
    }
    return hipGetLastError();
}

hipError_t launch_fsm_gather(const FsmParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(fsm_gather_kernel, dim3(1), dim3(1024), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ookd
