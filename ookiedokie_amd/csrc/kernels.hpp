// kernels.hpp -- launch interface between the rx context (rx.cpp) and the
// gfx950 kernels (kernels.hip).  All pointers are device pointers unless
// noted.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include <cstdint>

namespace ookd {

constexpr int kMaxStages = 8;
constexpr int kTapChunk = 32;           // taps are padded to a multiple of this
constexpr int kFirThreads = 256;        // lanes behind one 4096-output block of bit words
constexpr int kFirR = 16;               // outputs per lane in the 1-stage kernel
constexpr int kFirTile = kFirThreads * kFirR;   // 4096 outputs per workgroup
constexpr int kFirWaves = kFirThreads / 64;     // 1024-output wave tiles per 4096-output block
constexpr int kFirWgWaves = 1;                  // wave tiles per workgroup of the 1-stage kernel
// outputs per lane of the 1-stage kernel (tile = 64 * R): with the quiet shortcut the small tile
// (finer shortcut, smaller LDS window = more waves per CU: -6 % with 32 taps on the bench capture,
// -8 % with 255 taps); without it the MAC-efficient large tile
constexpr int kFir1RShort = 8, kFir1RLong = 16;
constexpr uint32_t kFir1ShortTaps = 256;        // padded tap count up to which the small tile is used (all)
constexpr int kWaveTile = 64 * kFirR;           // 1024 outputs per wavefront
constexpr int kQuietCounters = 1024;
constexpr int kGenTile = 1024;          // final outputs per workgroup, generic kernel
constexpr int kBlockWords = 64;         // one edge block = 64 words = 4096 bits
constexpr int kPayloadWords = 5;        // 4 x u64 payload + 1 spare (bit index == max_bits)
// ---- front end -------------------------------------------------------------

struct FirStageDev {
    uint32_t decim;
    uint32_t ntaps;         // true tap count
    uint32_t ntaps_pad;     // padded to kTapChunk (1-stage kernel)
    uint32_t tap_off;       // offset into the taps array (floats)
};

struct FrontParams {
    const int16_t *iq;          // captures, interleaved I,Q
    const float *iq_f32;        // alternative float2 input (streaming FIR API)
    uint64_t cap_stride;        // samples between consecutive captures
    uint64_t n_valid;           // samples present per capture; beyond: zeros
    uint64_t n_in;              // samples consumed per capture (padded)
    uint64_t n_out;             // decimated outputs per capture
    uint64_t origin;            // global index of local input sample 0
    const int16_t *halo;        // samples preceding sample 0 (newest last) or null
    const float *halo_f32;
    uint32_t halo_len;
    uint32_t num_stages;        // 0 = no filter
    FirStageDev stage[kMaxStages];
    const float *taps;          // all stages, 1-stage kernel: zero padded
    uint64_t *bits;             // [captures][words_per_cap]
    uint64_t words_per_cap;     // multiple of kBlockWords, tile-padded
    float *fir_out;             // optional float2 [captures][n_out]
    float p_star;               // smallest power whose sqrtf >= threshold
    float p_lo, p_hi;           // guard band (fast mode): p<p_lo => 0, p>=p_hi => 1
    int quiet_lsb;              // all |I|,|Q| of a window below this (raw LSB) => outputs provably < threshold
    unsigned long long *recompute_count;
    uint32_t *quiet_count;      // kQuietCounters spread counters of waves that skipped the filter, or null
    uint32_t *tile_info;        // tuned kernels: per wave tile, level changes inside the tile
                                // (its first bit vs. the tile before NOT included)
                                // | first bit << 30 | last bit << 31; [captures][tiles_per_cap]
    uint32_t tiles_per_cap;
    uint32_t tile_base;         // first wave tile of this launch (set by launch_front: chunked runs)
    uint32_t sparse;            // tuned 1-stage kernels: quiet tiles store nothing -- what an earlier run left in
                                // their words / infos carries that run's stamp and reads as quiet (tile_live)
    uint32_t stamp_bits;        // this run's stamp << kTileStampShift, OR-ed into every tile info written
    // matrix-core form of the 1-stage kernel (fir_mfma.hip); mfma_a == null: packed-VALU form
    const void *mfma_a;         // A-fragment image of the split taps (mfma_prepare_taps)
    float mfma_c;               // accumulator * mfma_c = filter output
    float p_lo_n, p_hi_n;       // guard band of a tile whose samples all lie within +-2048 (one sample piece),
                                // in ACCUMULATOR units (power / mfma_c^2: mfma_scale_band)
    float p_lo_w, p_hi_w;       // ... of a tile with samples beyond (two sample pieces)
    uint32_t mfma_g;            // wave tiles (tickets) per wave of a workgroup (the A-fragment image is fetched once for them)
    uint32_t mfma_xcd;          // bit 0 / 1: the 1-stage / decimate-by-4 launches give every XCD one contiguous run of tiles
    uint32_t mfma_xcd_span;     // (set by the launchers: tiles per XCD of this launch, 0 = positions are tiles)
    uint64_t tile_end;          // first wave tile past this launch (set by launch_front_mfma)
    uint32_t mfma_debug;        // experiments (OOKD_MFMA_DEBUG): bit 0 = every tile takes the quiet exit (timing only)
};

// A tile info word:  level changes inside the tile (10 bits: tiles hold at most 1024 outputs; the change between the
// tile's first bit and the tile before is NOT in it) | which of the tile's 64-bit words holds the first of them
// << 10 (4 bits; round 3: edge_write reads that one word of a tile with one change, not the tile) | run stamp << 14
// (16 bits) | first bit << 30 | last bit << 31.  Words and info of a tile are only meaningful when the info carries
// the current run's stamp; anything else -- never written, or left by an earlier run (sparse output) -- is a quiet
// tile: all bits zero.
constexpr uint32_t kTileWordShift = 10;
constexpr uint32_t kTileStampShift = 14;
constexpr uint32_t kTileStampMask = 0x3fffc000u;
constexpr uint32_t kTileStampMax = 0xffffu;
// -> count | first word << 10 | first << 30 | last << 31 of a live tile, 0 of a quiet / stale one
__host__ __device__ __forceinline__ uint32_t tile_live(uint32_t info, uint32_t stamp_bits) {
    return ((info ^ stamp_bits) & kTileStampMask) ? 0u : (info & ~kTileStampMask);
}

// Sparse front-end output: zero the words and infos of every tile whose info does not carry `keep_stamp_bits`
// (0: of every tile with a non-zero info) -- makes the bit words dense again (ookd_rx_get_bits, a change of
// the run geometry, the stamp wrapping around).  Not part of a normal run.
hipError_t launch_clear_tiles(uint32_t *tile_info, uint64_t *bits, uint64_t ntiles, uint32_t tiles_per_cap,
                              uint64_t words_per_cap, uint32_t tile_bits, uint32_t keep_stamp_bits, hipStream_t stream);
// does this shape run on a kernel that honours FrontParams::sparse?
bool front_sparse_capable(const FrontParams &p);

// ---- streaming (persistent) form of the tuned front end ------------------------------
constexpr int kStreamGroup = 2;         // wave tiles per ticket (1024 outputs, 4 KiB of input)
constexpr int kStreamHeads = 1024;      // ticket heads (head = workgroup % heads owns groups = head mod heads)
constexpr int kStreamHeadStride = 16;   // dwords between heads (own 64-B lines: atomics execute at the memory side)
constexpr int kMaxChunks = 256;

struct StreamCtl {
    uint32_t *heads;            // [kStreamHeads * kStreamHeadStride] ticket heads, zero at launch
    uint32_t *done;             // [num_chunks] groups finished per chunk (zero at launch), or null
    const uint32_t *chunk_end;  // [num_chunks] first group (global, capture-major) past each chunk
    uint32_t num_chunks;
    uint32_t num_caps;
    uint32_t waves_per_cu;      // persistent single-wave workgroups per CU (0 = default)
    uint32_t groups_per_cap;    // filled in by the launcher: tiles_per_cap / kStreamGroup
    uint32_t num_heads;         // filled in by the launcher
    uint32_t static_stride;     // experiment: groups dealt statically (workgroup + k * grid) instead of by ticket
};

// ---- matrix-core form of the tuned 1-stage front end (fir_mfma.hip) ---------------------
constexpr uint32_t kMfmaTile = 1024;    // outputs per wave tile: 32 columns x 32 rows of v_mfma_f32_32x32x16_f16

struct MfmaTaps {
    std::vector<uint16_t> image;    // [K-steps][2 pieces][64 lanes][8] fp16 bit patterns
    uint32_t ksteps = 0;
    float c = 0.0f;                 // accumulator -> filter output
    double delta = 0.0;             // sum |h - (h1 + h2) / S|: what the two pieces do not carry
    double sum_abs = 0.0;           // sum |h|
    double sum_hat = 0.0;           // sum |h1 + h2| / S
};
// false: this filter does not go through the matrix cores (more than 256 taps, non-finite taps, ...)
bool mfma_prepare_taps(const float *taps, uint32_t ntaps, MfmaTaps &out);
// forward bound on |y_mfma - y_ref| per component (filter output units)
double mfma_error_bound(const MfmaTaps &t, uint32_t ntaps, bool wide);
// a band edge in accumulator units (p / c^2); false when the scaling is not exact in float
bool mfma_scale_band(const MfmaTaps &t, float p, float &out);
bool front_uses_mfma(const FrontParams &p);
// two decimate-by-2 stages (<= 16 and <= 32 taps) folded into one decimate-by-4 product (fir2_mfma_kernel)
bool mfma_prepare_taps2(const float *taps1, uint32_t n1, const float *taps2, uint32_t n2, MfmaTaps &out);
double mfma_error_bound2(const MfmaTaps &t, double e_ref, bool wide);
bool front_uses_mfma2(const FrontParams &p);
hipError_t launch_front_mfma2(const FrontParams &p, uint32_t num_captures, hipStream_t stream, hipEvent_t t0,
                              hipEvent_t t1, uint64_t tile_begin, uint64_t tile_count);
hipError_t launch_front_mfma(const FrontParams &p, uint32_t num_captures, hipStream_t stream, hipEvent_t t0,
                             hipEvent_t t1, uint64_t tile_begin, uint64_t tile_count);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when a kernel needs
// more than it was last granted (the call costs microseconds per launch).
hipError_t ensure_dynamic_lds(const void *func, size_t bytes);

// tile_begin / tile_count: a range of the capture's wave tiles (front_tile_bits() outputs each; tuned
// kernels only) -- the chunks of a pipelined run
hipError_t launch_front(const FrontParams &p, uint32_t num_captures, bool exact, hipStream_t stream,
                        hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr, uint64_t tile_begin = 0,
                        uint64_t tile_count = ~0ull);
// Streaming form (1 stage / decimation 1 / <= 256 taps only: front_streams()).  ctl.heads (and
// ctl.done) must be zero when the kernel starts; write_through: bit words / tile infos are stored
// past the L2 so that a kernel started while this one runs reads them (ctl.done tells when).
bool front_streams(const FrontParams &p);
hipError_t launch_front_stream(const FrontParams &p, StreamCtl ctl, bool exact, bool write_through,
                               hipStream_t stream, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);
// 1024-ish output windows ("wave tiles") the tuned kernels split a capture into
// (0 when the generic kernel serves this shape).
uint64_t front_wave_tiles(const FrontParams &p);
// bits per wave tile of the tuned kernel that serves this shape (0 = generic)
uint32_t front_tile_bits(const FrontParams &p);
// Generic multi-stage kernel regardless of shape (cross-check / streaming FIR).
hipError_t launch_front_generic(const FrontParams &p, uint32_t num_captures,
                                hipStream_t stream);
size_t generic_lds_bytes(const FrontParams &p);

// ---- edges -----------------------------------------------------------------

constexpr int kScanGroup = 1024;        // blocks per first-level scan group

struct EdgeParams {
    const uint64_t *bits;
    uint64_t words_per_cap;
    uint64_t n_out;             // decimated samples per capture: no edge at or beyond it
    uint32_t num_captures;
    uint32_t blocks_per_cap;    // words_per_cap / kBlockWords
    uint32_t *blk_count;        // [captures * blocks_per_cap]
    uint32_t *blk_offset;       // exclusive prefix over all blocks, size + 1
    uint32_t *group_total;      // [ceil(blocks / kScanGroup)] scratch
    uint64_t *edges;            // capture-local decimated indices
    uint64_t edge_capacity;
    uint32_t *overflow;         // set to 1 when total edges > capacity
    const uint32_t *tile_info;  // per wave tile counts from the tuned front-end kernels, or null
    uint32_t tiles_per_block;   // wave tiles per 4096-bit block (4 or 16)
    // chunked (pipelined) runs: this launch covers one chunk of a capture, positions are chunk-local
    uint32_t *total_acc;        // += the chunk's level changes, or null
    uint32_t has_prev;          // bits / tile_info continue in front of the chunk: the level before its first
                                // sample is the last bit of the chunk before (0 for a capture's or shard's start)
    uint32_t stamp_bits;        // the front end's (FrontParams::stamp_bits): tile infos without it are quiet tiles
};

hipError_t launch_edges(const EdgeParams &p, hipStream_t stream);

// ---- state machine ---------------------------------------------------------
//
// The kernel keeps the tables in VGPR lanes (lane i = trigger i / state i),
// hence at most 64 of each.

constexpr int kMaxStates = 64;
constexpr int kMaxTriggers = 64;
// Devices beyond that (the reference allocates states and triggers dynamically, state_machine.c:135-235)
// run through the round form with the tables in LDS instead of in the lanes' registers:
//   big[0..3] = states, max_bits, triggers, quiet state;
//   per state   (6 words): kmin, kmax, kto (0xffffffff = none), first trigger, end trigger, flags
//   per trigger (3 words): kmin, kmax, cond | action << 8 | next << 16
constexpr int kMaxStatesBig = 1024;
constexpr int kMaxTriggersBig = 2048;
constexpr uint32_t kBigHeaderWords = 4, kBigStateWords = 6, kBigTrigWords = 3;

struct FsmTablesDev {
    uint32_t num_states, max_bits, num_triggers;
    uint32_t quiet_state;       // state every trajectory settles in on a long 0 level
    uint64_t state_kmin[kMaxStates], state_kmax[kMaxStates], state_kto[kMaxStates];
    uint32_t state_tbeg[kMaxStates], state_tend[kMaxStates];
    uint32_t state_flags[kMaxStates];       // bit 0: k never influences this state
    uint64_t trig_kmin[kMaxTriggers], trig_kmax[kMaxTriggers];
    uint32_t trig_info[kMaxTriggers];       // cond | action << 8 | next << 16
};

struct FsmStateDev {            // same layout as ookd_fsm_state
    uint32_t cur, nbits;
    uint64_t k;
    uint32_t prev, pad;
    uint64_t data[kPayloadWords];
};
static_assert(sizeof(FsmStateDev) == 64, "FsmStateDev layout");

struct SegState {               // state carried between segments of one capture
    FsmStateDev st;
    uint64_t skip_to;           // samples below this index are not fed in (device.c:646)
    uint64_t pad;
};
static_assert(sizeof(SegState) == 80, "SegState layout");

struct MsgDev {                 // same layout as ookd_message
    uint32_t capture, reserved;
    uint64_t sample;
    uint64_t payload[4];
};
static_assert(sizeof(MsgDev) == 48, "MsgDev layout");

struct FsmParams {
    const FsmTablesDev *tables;
    const uint64_t *bits;
    uint64_t words_per_cap;
    const uint64_t *edges;
    const uint32_t *blk_offset;
    uint32_t blocks_per_cap;
    uint32_t num_captures;
    uint64_t n_out;             // decimated samples per capture
    uint32_t spb;               // input samples per buffer
    uint32_t total_decim;
    uint64_t seg_len;           // nominal decimated samples per segment
    uint32_t segs_per_cap;
    uint32_t msg_slots, err_slots;
    uint64_t *seg_bounds;       // [captures][segs_per_cap + 1]
    SegState *state_in;         // [segs]
    SegState *state_out;        // [2][segs] ping-pong by round parity
    MsgDev *seg_msgs;           // [segs][msg_slots]
    uint32_t *seg_msg_count;    // [segs]
    uint64_t *seg_errs;         // [segs][err_slots]
    uint32_t *seg_err_count;    // [segs]
    uint32_t *changed;          // [rounds of one batch]
    uint32_t *flags;            // bit0: message slot overflow
    // compaction
    MsgDev *msgs;               // [msg_capacity]
    uint64_t msg_capacity;
    uint64_t *totals;           // [0] messages, [1] errors
    uint64_t *debug;            // optional [segs][4]: loop turns, cycles, fused edges, window loads
    // the bit words are only valid in tiles whose info carries the run's stamp (tile_live); null: everywhere
    const uint32_t *tile_info;  // [captures][tiles_per_cap], offset like `bits` for a chunk
    uint32_t tiles_per_cap;
    uint32_t tile_shift;        // log2(outputs per tile)
    uint32_t stamp_bits;
    const uint32_t *big;        // tables of a device with more than 64 states / triggers (see kMaxStatesBig), or null
    uint32_t big_words;
    // the edge stage's overflow flag (device): set = the list's tail was never written and blk_offset counts
    // edges that are not there -- the round-form kernels return at once (the host reports OOKD_ERR_CAPACITY)
    const uint32_t *edge_overflow;
};

// level of decimated sample `pos` of capture `cap` (pos may be -1 for a chunk: the sample in front of it)
__device__ __forceinline__ uint32_t fsm_level_at(const FsmParams &p, uint32_t cap, int64_t pos) {
    if (p.tile_info) {
        const int64_t tile = pos >> p.tile_shift;
        if (!tile_live(p.tile_info[(int64_t)cap * p.tiles_per_cap + tile], p.stamp_bits)) return 0u;
    }
    const uint64_t *w = p.bits + (uint64_t)cap * p.words_per_cap;
    return (uint32_t)((w[pos >> 6] >> (pos & 63)) & 1ull);
}

hipError_t launch_fsm_prepare(const FsmParams &p, const FsmStateDev *first_state,
                              hipStream_t stream);
// mode 0: run every segment; 1: rerun segments whose incoming state changed;
// 2: as 1 and the first segment of each capture reruns (shard refine).
hipError_t launch_fsm_round(const FsmParams &p, uint32_t parity, uint32_t mode, uint32_t slot,
                            hipStream_t stream);
hipError_t launch_fsm_gather(const FsmParams &p, hipStream_t stream);

// ---- results -> pinned host memory ----------------------------------------------
struct PublishParams {
    uint32_t *d_hdr;                // device result header, as dwords; zeroed afterwards
    uint32_t *h_hdr;                // host-mapped copy
    uint32_t hdr_words;             // <= 256
    uint32_t totals_word;           // dword index of the u64 message total
    uint32_t edges_word;            // dword index that receives *total_edges
    const uint32_t *total_edges;    // or null
    const uint4 *d_msgs;            // or null (no state machine)
    uint4 *h_msgs;
    uint64_t first_msgs;            // messages published with the header
    uint32_t done_word;             // dword index of a zero-initialised completion counter (multi-workgroup publishers)
};
hipError_t launch_publish(const PublishParams &p, hipStream_t stream);

// ---- state machine as a scan over edges (fsm_scan.hip) -----------------------------

struct LeafEvDev {              // what happened in one edge-to-edge span
    uint8_t napp, nout, nerr;
    uint8_t flags;              // bit0: passed through reset; bit2: ends inside a skipped rest-of-buffer
    uint8_t apps_at_reset;      // appends of this span before its last reset
    uint8_t out_ab[2];          // appends of this span before output j
    uint8_t out_rb[2];          // apps_at_reset as of output j (0xff: no reset before it in this span)
    uint8_t end_cur, end_prev, pad;
    uint32_t appvals;           // bit j = value of the j-th append
    uint32_t end_k;
    uint64_t out_pos[2];
    uint64_t err_pos;
};
static_assert(sizeof(LeafEvDev) == 48, "LeafEvDev layout");

struct FsmScanArgs {
    FsmParams f;                // tables, edges, geometry, msgs / totals
    uint32_t D, S;              // abstract states (with stuck codes), machine states
    uint32_t SNB;               // S * (max_bits + 2): the normal codes
    uint32_t leaf_block;        // from fsm_scan_leaf_block()
    uint32_t grid_blocks;       // persistent workgroups for the leaf / emit kernels
    uint16_t *block_tab;        // [total_blocks_cap][D rounded up to 8]
    const uint32_t *lt_off, *lt_n0, *lt_pk;     // span tables from build_leaf_tables, or null
    uint32_t lt_words;          // their size in words (offsets + 2 x intervals)
    const uint32_t *lt_merged;  // build_merged_rows of them, or null
    uint32_t lt_merged_words;
    const void *ltab;           // device copy of the kernels' table layout (fsm_scan_fill_ltab)
    PublishParams publish;      // d_hdr != null: the scan's last kernel also publishes the results
    const uint16_t *reach;      // codes a span can be entered in (ascending), or null = all
    uint32_t nreach, nreach_base;       // all / those below the stuck codes (S * (max_bits + 2) + 3)
    uint32_t nreach_lv[2];      // behind the nreach entries: the codes met at level 0, then those met at level 1
    uint32_t *cap_block_off;    // [captures + 1]
    uint32_t total_blocks_cap;
    LeafEvDev *events;          // [edges + captures]
    uint32_t *ev_hot;           // same count: a leaf's record in one word (the 48-byte one only where it is needed)
    uint8_t *app_vals;          // append pool
    uint64_t app_capacity;
    uint64_t *errs;             // flat error list
    uint64_t err_capacity;
    const FsmStateDev *first;   // incoming state (host pointer) or null = reset
    // chunked (pipelined) runs, one launch per chunk of one capture (all null / 0 otherwise):
    const SegState *first_dev;  // incoming state in device memory: the chunk before's final_state
    uint64_t pos_origin;        // the chunk's first decimated sample
    const uint64_t *totals_in;  // [2] messages / errors of the chunks before (device)
    const uint32_t *edge_overflow;      // the edge stage's overflow flag (device), or null
    uint32_t *cap_fallback;     // [captures] per-capture refusal bits (zero at launch), or null
    uint16_t *pre_codes;        // [total_blocks_cap][leaf_block] entry code of every leaf (scan_entry_kernel)
    uint16_t *blk_in;           // [total_blocks_cap] entry code of every block
    uint16_t *rowz;             // [total_blocks_cap][leaf_block] scratch of the entry passes: merged-rows interval of every leaf
    uint32_t *skipc;            // [total_blocks_cap][leaf_block] every leaf applied to the two skip codes (wave leaf kernel -> entry walk)
    SegState *final_state;      // [captures]
    uint32_t *fallback;         // device word: non-zero => result invalid, use the round path
    uint32_t *fin_off;          // [captures + 1]
    void *fsum;                 // [fin_blocks_cap] x 32 B: stamped block aggregates
    unsigned long long *fin_ticket;     // the finish kernel's work counter: count | run stamp << 32, never zeroed
    uint32_t run_stamp;         // != 0, changes every launch
    uint32_t fin_blocks_cap;
    uint32_t *cap_group_off;    // [captures + 1]
    uint16_t *group_tab;        // [total_blocks_cap / 16 + captures + 1][D rounded up to 8]
    uint32_t *cap_super_off;    // [captures + 1]
    uint16_t *super_tab;        // [total_blocks_cap / 64 + captures + 1][D rounded up to 8]
    uint16_t *super_in;         // same count
    uint16_t *cap_end;          // [captures]
    uint16_t *cap_first;        // [captures]
    uint32_t *sync_rec;         // [total_blocks_cap][8] the walk's block records (fsm_scan.hip, kSyncRec*)
    uint64_t pre_plane;         // pre_codes holds 4 planes this many elements apart (0: one plane, no such walk)
    uint32_t lt_sync_words;     // lt_merged's size with append_sync_codes' tables (lt_merged_words: the rows alone)
    uint32_t *sync_fail;        // device word, zero at launch: the walk from the synchronising spans gave up
    uint32_t sync_try;          // 1: try that walk first, the composing kernels queued behind it (they return at once unless
                                // it gave up); 2: the walk alone -- giving up refuses the run with kScanFbSync and the host
                                // queues it again with 0: the composing kernels only
};

constexpr uint32_t kScanFbSync = 16;      // refusal bit: the walk from synchronising spans gave up in a launch without composing kernels
uint32_t fsm_scan_leaf_block(uint32_t D, uint32_t S, uint32_t SNB);
// The trigger / state tables in the layout the scan kernels keep in LDS: size, and
// a host-side fill (16-byte aligned destination) to be uploaded once per context.
size_t fsm_scan_ltab_bytes();
// returns the size of the abstract domain (states x bit counts + 3 + stuck codes)
uint32_t fsm_scan_fill_ltab(void *dst, const FsmTablesDev &tables, uint32_t spb, uint32_t decim,
                            const std::vector<uint16_t> &stuck_src, const std::vector<uint8_t> &stuck_rows);
// Packed result of a span as a step function of its length, per (row, level)
// (host side; false = not tabulated, the kernels simulate).
// reach: the abstract codes a span can be entered in (closure of the tables' results);
// empty when that cannot be told.  stuck_src / stuck_rows: the normal codes / table rows with
// a "no trigger fired on the edge" result (they extend the domain: fsm_scan_fill_ltab).
bool build_leaf_tables(const FsmTablesDev &tables, uint32_t spb, uint32_t decim, std::vector<uint32_t> &off,
                       std::vector<uint32_t> &n0, std::vector<uint32_t> &pk, std::vector<uint16_t> &reach,
                       std::vector<uint16_t> &stuck_src, std::vector<uint8_t> &stuck_rows);
// The span tables merged over the rows: per level the sorted union of all (state, class) rows' breakpoints,
// and for every interval between two of them the 2S packed rows a span of that length has -- ONE search per
// leaf, independent of the state the leaf is entered in.  Layout (32-bit words):
//   [0] nbp level 0, [1] nbp level 1, [2] 2S, [3] 0 | bp level 0 | bp level 1 | rows level 0 [nbp0][2S] |
//   rows level 1 [nbp1][2S]
std::vector<uint32_t> build_merged_rows(uint32_t S, const std::vector<uint32_t> &off, const std::vector<uint32_t> &n0,
                                        const std::vector<uint32_t> &pk);
// Marks the intervals of the merged rows in which a span is SYNCHRONISING (every row a span of that level can be
// entered in holds the same absolute normal code): one word per interval behind the rows (the code, or 0xffff),
// header word [3] = where they start.  reach: code | level mask << 14 as build_leaf_tables returns it.
void append_sync_codes(std::vector<uint32_t> &merged, uint32_t S, uint32_t NB1, uint32_t max_bits,
                       const std::vector<uint16_t> &reach);
uint32_t fsm_scan_fin_block();
// t_end (optional): event that takes the end time stamp of the scan's last kernel
hipError_t launch_fsm_scan(const FsmScanArgs &a, hipStream_t stream, hipEvent_t t_end = nullptr);

// ---- workgroup inclusive sum (device code) ---------------------------------------------
// Shuffles inside each wavefront + one exchange of the wave totals through
// `wtot` (>= blockDim/64 entries of shared memory).  Returns the inclusive sum
// of v over the workgroup's lanes 0..tid; *total = the sum over all lanes.
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t wg_inclusive_sum(uint32_t v, uint32_t *wtot, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63u) >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if ((int)lane >= d) v += t;
    }
    __syncthreads();                    // wtot may still be read from a previous call
    if (lane == 63u || threadIdx.x == blockDim.x - 1) wtot[wave] = v;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < nwaves; ++w) {
        const uint32_t t = wtot[w];
        if (w < wave) before += t;
        all += t;
    }
    if (total) *total = all;
    return v + before;
}
#endif

// ---- unpack (backend rx) ----------------------------------------------------
hipError_t launch_unpack(const int16_t *iq, float *out, uint64_t n, hipStream_t stream);
// ---- pack (post-filter recorder) ----------------------------------------------
hipError_t launch_pack(const float *in, int16_t *iq, uint64_t n, hipStream_t stream);

// ---- synthetic generator ------------------------------------------------------

struct SynthRun {               // one constant-envelope run of the capture
    uint64_t start;             // first sample of the run
    int16_t i_level, q_level;   // carrier-rotated on level (0,0 when off)
    uint32_t pad;
};

// Same integer function on host and device: sample = level + noise.
__host__ __device__ inline uint64_t synth_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ inline void synth_noise(uint64_t seed, uint64_t idx, uint32_t amp,
                                            int &ni, int &nq) {
    if (amp == 0) {
        ni = nq = 0;
        return;
    }
    const uint64_t h = synth_mix(seed ^ (idx * 0xD1342543DE82EF95ull));
    const uint32_t span = 2 * amp + 1;
    ni = (int)(((h & 0xffffffffull) * span) >> 32) - (int)amp;
    nq = (int)(((h >> 32) * span) >> 32) - (int)amp;
}

hipError_t launch_synth(const SynthRun *runs, uint64_t num_runs, uint64_t seed,
                        uint32_t noise, uint64_t first, uint64_t count, int16_t *iq,
                        hipStream_t stream);

}  // namespace ookd
