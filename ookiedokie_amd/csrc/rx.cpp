// rx.cpp -- rx context: HBM buffers, kernel sequencing, C ABI of the fused
// demodulation entry (replaces the body of the reference loop,
// src/ookiedokie.c:243-288, for whole captures resident in HBM).
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>

#include "common.hpp"
#include "ingest.hpp"
#include "kernels.hpp"

using namespace ookd;

namespace {

#define HIPCHK(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)_e,                  \
                      hipGetErrorString(_e), __FILE__, __LINE__, #expr);          \
            return OOKD_ERR_HIP;                                                  \
        }                                                                         \
    } while (0)

constexpr uint32_t kIterBatch = 4;      // FSM fix-point rounds queued per host sync

constexpr uint64_t kHostMsgFirst = 2048;    // messages copied with the header
// control block of the streaming front end: ticket heads | per-chunk counters | chunk ends
constexpr size_t kCtlHeads = 0, kCtlDone = (size_t)kStreamHeads * kStreamHeadStride, kCtlChunkEnd = kCtlDone + kMaxChunks,
                 kCtlWords = kCtlChunkEnd + kMaxChunks;

struct ResultHeader {
    uint32_t changed[kIterBatch];
    uint32_t flags;
    uint32_t edge_overflow;
    uint64_t totals[2];
    unsigned long long recompute;
    uint32_t total_edges;
    uint32_t scan_fallback;
    uint32_t sync_fail;         // the scan's walk from synchronising spans gave up (the composing kernels ran)
    uint32_t publish_done;      // workgroups of the scan's publishing kernel that are through
};

// one chunk of a pipelined single-capture run
struct Chunk {
    uint64_t out0, nout;        // decimated samples [out0, out0 + nout) of the capture
    uint32_t blk0, nblk;        // 4096-output blocks
    uint64_t edge_off, edge_cap;        // its region of the edge list
};
constexpr uint64_t kPipeDefaultChunk = 1ull << 28;      // input samples
constexpr uint64_t kPipeTailChunk = 1ull << 25;         // the last chunks shrink down to this: a short exposed chain

}  // namespace

// Front-end kernels of contexts that share a gate take turns: they are HBM bound, so running two
// at once only makes both slower, while everything after them (edges, state machine: latency
// bound) overlaps the next context's front end.  The gate remembers the stop event of the
// front-end launch queued last (it rides on that kernel's dispatch: launch_front); a context
// makes its stream wait for it before launching its own -- no marker packets.  An explicit
// object handed to ookd_rx_create by the caller: there is no process-wide state.
struct ookd_rx_gate {
    std::mutex m;
    hipEvent_t last = nullptr;
};

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        n = count;
        if (count == 0) return OOKD_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
            p = nullptr;
            return OOKD_ERR_NOMEM;
        }
        return OOKD_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// smallest float p with sqrtf(p) >= thr  (SURVEY.md hard part 3)
float power_threshold(float thr) {
    if (std::isnan(thr)) return NAN;
    if (thr <= 0.0f) return 0.0f;
    if (std::isinf(thr)) return INFINITY;
    float p = (float)((double)thr * (double)thr);
    while (p > 0.0f && sqrtf(nextafterf(p, 0.0f)) >= thr) p = nextafterf(p, 0.0f);
    while (!std::isinf(p) && sqrtf(p) < thr) p = nextafterf(p, INFINITY);
    return p;
}

// Guard band for the fused-multiply-add FIR (1 stage): any sample whose
// FMA-computed power lies in [p_lo, p_hi) is recomputed in reference order.
// e bounds |y_fma - y_ref| per component: both chains are within
// gamma_T * sum|h||x| of the exact sum (one rounding per step for fma, two
// for mul+add), inputs are bounded by 32768/2048 = 16.
//
// Several stages: the fused and the reference chain of stage s+1 start from
// inputs that already differ by e_s, which the stage amplifies by at most
// sum|h_{s+1}|, and add their own rounding difference on values bounded by
// 16 * prod sum|h|:  e = 2.2 u * 16 * prod_s S_s * sum_s (T_s + 1).
void band_from_error(double e, float p_star, float &p_lo, float &p_hi);
double guard_error(const std::vector<std::vector<float>> &stages, double x_max);
void guard_band(const std::vector<std::vector<float>> &stages, float p_star, float &p_lo, float &p_hi) {
    if (std::isnan(p_star) || std::isinf(p_star) || p_star <= 0.0f) {
        p_lo = p_hi = p_star;
        return;
    }
    const double u = std::ldexp(1.0, -24);
    double S = 1.0, T = 0.0, Tsum = 0.0;
    for (const auto &taps : stages) {
        double ss = 0.0;
        for (float t : taps) ss += std::fabs((double)t);
        S *= std::max(ss, 1.0);         // a stage with gain < 1 still adds its own roundings
        T += (double)taps.size() + 1.0;
        Tsum += (double)taps.size();
    }
    const double e = 2.2 * T * u * S * 16.0 * (stages.size() > 1 ? 1.01 : 1.0) + Tsum * std::ldexp(1.0, -140);
    band_from_error(e, p_star, p_lo, p_hi);
}

// the same bound for samples up to x_max (in units of 2048 LSB) instead of 16
double guard_error(const std::vector<std::vector<float>> &stages, double x_max) {
    const double u = std::ldexp(1.0, -24);
    double S = 1.0, T = 0.0, Tsum = 0.0;
    for (const auto &taps : stages) {
        double ss = 0.0;
        for (float t : taps) ss += std::fabs((double)t);
        S *= std::max(ss, 1.0);
        T += (double)taps.size() + 1.0;
        Tsum += (double)taps.size();
    }
    return 2.2 * T * u * S * x_max * (stages.size() > 1 ? 1.01 : 1.0) + Tsum * std::ldexp(1.0, -140);
}

// [p_lo, p_hi) around p_star for a filter output known to within e per component
void band_from_error(double e, float p_star, float &p_lo, float &p_hi) {
    if (std::isnan(p_star) || std::isinf(p_star) || p_star <= 0.0f) {
        p_lo = p_hi = p_star;
        return;
    }
    const double u = std::ldexp(1.0, -24);
    const double P = (double)p_star;
    // |p_ref - p_fma| <= m(p) = 3.003*e*sqrt(p) + 3e^2 + 6u*p
    // upper edge: smallest s = sqrt(p) with (1-6u)s^2 - 3.003e s - (3e^2 + P) >= 0
    {
        const double a = 1.0 - 6.0 * u, b = 3.003 * e, c = 3.0 * e * e + P;
        double s = (b + std::sqrt(b * b + 4.0 * a * c)) / (2.0 * a);
        double ph = s * s * (1.0 + 1e-6);
        ph = std::max(ph, 4.0 * e * e);     // p - m(p) is increasing beyond ~2.3e^2
        float f = (float)ph;
        if ((double)f < ph) f = nextafterf(f, INFINITY);
        f = nextafterf(f, INFINITY);
        p_hi = std::max(f, p_star);
    }
    // lower edge: largest s with (1+6u)s^2 + 3.003e s + 3e^2 - P < 0
    {
        const double a = 1.0 + 6.0 * u, b = 3.003 * e, c = 3.0 * e * e - P;
        if (c >= 0.0) {
            p_lo = 0.0f;
        } else {
            double s = (-b + std::sqrt(b * b - 4.0 * a * c)) / (2.0 * a);
            double pl = s > 0.0 ? s * s * (1.0 - 1e-6) : 0.0;
            float f = (float)pl;
            if ((double)f > pl) f = nextafterf(f, 0.0f);
            f = nextafterf(f, 0.0f);
            p_lo = std::max(f, 0.0f);
        }
    }
}

// Which state does every trajectory settle in while the input stays low?
// Walks the tables from reset with no edges: only always / timeout /
// msg_complete triggers can fire.  Used only as the ASSUMED incoming state of
// a state machine segment (a wrong guess costs fix-point rounds, never
// correctness).
uint32_t quiet_state(const ookd_device &d) {
    const uint64_t NONE = ~0ull;
    uint32_t cur = 0, nbits = 0;
    uint64_t k = 0;
    const size_t ns = d.state_duration_us.size();
    for (size_t step = 0; step < 4 * ns + 8; ++step) {
        int fire = -1;
        uint64_t best = NONE;
        for (uint32_t t = d.trig_begin[cur]; t < d.trig_begin[cur + 1]; ++t) {
            uint64_t lo = d.trig_kmin[t];
            const uint8_t c = d.trig_cond[t];
            if (c == 4) {               // timeout
                if (d.state_kto[cur] == NONE) continue;
                lo = std::max(lo, d.state_kto[cur]);
            } else if (c == 5) {        // msg_complete
                if (nbits < d.num_bits) continue;
            } else if (c != 1) {        // pulse triggers need an edge
                continue;
            }
            const uint64_t first = std::max(k, lo);
            if (first > d.trig_kmax[t]) continue;
            if (first - k < best) {     // earliest in time, then first in file order
                best = first - k;
                fire = (int)t;
            }
        }
        if (fire < 0) return cur;
        const uint8_t act = d.trig_action[fire];
        if (act == 2 || act == 3) nbits++;
        cur = d.trig_next[fire];
        if (cur == 0) nbits = 0;
        k = 0;
    }
    return 0;
}

// kernels.hpp: kMaxStatesBig -- the packed tables of a device with more than 64 states / triggers
std::vector<uint32_t> big_tables(const ookd_device &d, uint32_t quiet) {
    const uint64_t NONE = ~0ull;
    auto c32 = [&](uint64_t v) { return v == NONE ? 0xffffffffu : (uint32_t)v; };       // finite bounds are < 2^31
    const size_t ns = d.state_duration_us.size(), nt = d.trig_cond.size();
    std::vector<uint32_t> w(kBigHeaderWords + kBigStateWords * ns + kBigTrigWords * nt);
    w[0] = (uint32_t)ns;
    w[1] = d.num_bits;
    w[2] = (uint32_t)nt;
    w[3] = quiet;
    for (size_t s = 0; s < ns; ++s) {
        uint32_t *r = &w[kBigHeaderWords + kBigStateWords * s];
        r[0] = c32(d.state_kmin[s]);
        r[1] = c32(d.state_kmax[s]);
        r[2] = c32(d.state_kto[s]);
        r[3] = d.trig_begin[s];
        r[4] = d.trig_begin[s + 1];
        bool irrelevant = d.state_kmin[s] == 0 && d.state_kmax[s] == NONE;
        for (uint32_t i = d.trig_begin[s]; i < d.trig_begin[s + 1]; ++i) {
            if (d.trig_kmin[i] != 0 || d.trig_kmax[i] != NONE) irrelevant = false;
            if (d.trig_cond[i] == 4 && d.state_kto[s] != NONE) irrelevant = false;
        }
        r[5] = irrelevant ? 1u : 0u;
    }
    for (size_t i = 0; i < nt; ++i) {
        uint32_t *r = &w[kBigHeaderWords + kBigStateWords * ns + kBigTrigWords * i];
        r[0] = c32(d.trig_kmin[i]);
        r[1] = c32(d.trig_kmax[i]);
        r[2] = (uint32_t)d.trig_cond[i] | ((uint32_t)d.trig_action[i] << 8) | ((uint32_t)d.trig_next[i] << 16);
    }
    return w;
}

void fill_fsm_tables(const ookd_device &d, FsmTablesDev &t) {
    const uint64_t NONE = ~0ull;
    const size_t ns = d.state_duration_us.size();
    const size_t nt = d.trig_cond.size();
    t.num_states = (uint32_t)ns;
    t.max_bits = d.num_bits;
    t.num_triggers = (uint32_t)nt;
    for (size_t i = 0; i < (size_t)kMaxTriggers; ++i) {
        t.trig_kmin[i] = 1;             // empty window: never matches
        t.trig_kmax[i] = 0;
    }
    for (size_t s = 0; s < (size_t)kMaxStates; ++s) {
        t.state_kmax[s] = NONE;
        t.state_kto[s] = NONE;
    }
    t.quiet_state = quiet_state(d);
    if (ns > (size_t)kMaxStates || nt > (size_t)kMaxTriggers) return;    // a big device: only the counts (its tables: big_tables)
    for (size_t s = 0; s < ns; ++s) {
        t.state_kmin[s] = d.state_kmin[s];
        t.state_kmax[s] = d.state_kmax[s];
        t.state_kto[s] = d.state_kto[s];
        t.state_tbeg[s] = d.trig_begin[s];
        t.state_tend[s] = d.trig_begin[s + 1];
        // k is irrelevant in a state with no duration, no live timeout and no
        // trigger duration windows
        bool irrelevant = d.state_kmin[s] == 0 && d.state_kmax[s] == NONE;
        for (uint32_t i = d.trig_begin[s]; i < d.trig_begin[s + 1]; ++i) {
            if (d.trig_kmin[i] != 0 || d.trig_kmax[i] != NONE) irrelevant = false;
            if (d.trig_cond[i] == 4 && d.state_kto[s] != NONE) irrelevant = false;
        }
        t.state_flags[s] = irrelevant ? 1u : 0u;
    }
    for (size_t i = 0; i < nt; ++i) {
        t.trig_kmin[i] = d.trig_kmin[i];
        t.trig_kmax[i] = d.trig_kmax[i];
        t.trig_info[i] = (uint32_t)d.trig_cond[i] | ((uint32_t)d.trig_action[i] << 8) |
                         ((uint32_t)d.trig_next[i] << 16);
    }
    t.quiet_state = quiet_state(d);
}

uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

}  // namespace

struct ookd_rx {
    ookd_rx_config cfg{};
    int dev = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};

    // filter
    uint32_t num_stages = 0;
    FirStageDev stage[kMaxStages]{};
    uint32_t total_decim = 1;
    uint64_t halo_needed = 0;
    std::vector<float> taps0;       // stage-0 true taps (guard band)
    DevBuf<float> d_taps;
    float p_star = 0, p_lo = 0, p_hi = 0;
    // matrix-core form of the 1-stage front end (fir_mfma.hip): A-fragment image, scale, bands; empty = packed-VALU form
    DevBuf<uint16_t> d_mfma_a;
    float mfma_c = 0, p_lo_n = 0, p_hi_n = 0, p_lo_w = 0, p_hi_w = 0;
    uint32_t mfma_g = 0;
    uint32_t mfma_xcd = 2;          // FrontParams::mfma_xcd (OOKD_MFMA_XCD)
    int quiet_lsb = 0;              // 0 = the quiet shortcut never applies
    bool exact = false;
    bool count_quiet = false;
    DevBuf<uint32_t> d_quiet;       // kQuietCounters spread counters (diagnostics), running totals
    std::vector<uint32_t> quiet_prev;       // what they held after the run before
    DevBuf<uint32_t> d_tile_info;   // per wave tile edge counts written by the tuned front-end kernels
    DevBuf<uint32_t> d_ctl;         // streaming front end: ticket heads | chunk counters | chunk ends
    // sparse front-end output: quiet tiles store nothing; the tile infos carry the run's stamp and tiles
    // without it read as quiet (kernels.hpp: tile_live).  Extents of the previous run: a change of the
    // geometry zeroes them first.
    bool sparse = false;
    uint64_t dirty_tiles = 0, dirty_words_per_cap = 0;
    uint32_t dirty_tiles_per_cap = 0;
    uint32_t tile_stamp = 0;            // of the current run's front end, 1 .. kTileStampMax
    mutable bool bits_dense = false;    // stale tiles of the current run were zeroed (ookd_rx_get_bits)
    int densify_bits() const;
    // chunk pipeline (single-capture runs): front end on s_front, edges + state machine on s_chain, each
    // on its own half of the CUs
    bool pipe_ok = false;
    uint64_t pipe_chunk_in = 0;     // target input samples per chunk
    hipStream_t s_front = nullptr, s_chain = nullptr;
    std::vector<hipStream_t> dummy_streams;     // OOKD_PIPE_DUMMY experiment
    hipEvent_t ev_start = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_c0, ev_c1;       // per chunk: front-end kernel start / stop
    std::vector<Chunk> chunks;      // of the last run (empty: not pipelined)
    DevBuf<SegState> d_carry;       // [2] state handed from chunk to chunk
    DevBuf<uint64_t> d_chunk_totals;    // [2][2] messages / errors so far
    DevBuf<unsigned long long> d_fin_tickets;   // [kMaxChunks + 1] the finish kernel's stamped work counters: one per
                                                // chunk of a pipelined run, the last one for whole runs
    const void *last_iq = nullptr;  // arguments of the last run (a refused pipelined run is redone whole)
    uint64_t last_stride = 0;
    bool no_pipeline_once = false;
    uint32_t front_launches = 1;    // of the last run
    DevBuf<uint32_t> d_cap_fallback;        // [captures] the scan's per-capture refusal bits (batched runs)
    DevBuf<uint16_t> d_pre, d_blk_in;       // entry code of every leaf / block (scan_entry_kernel)
    DevBuf<uint16_t> d_rowz;                // merged-rows interval of every leaf (scan_entry_kernel)
    DevBuf<uint32_t> d_skipc;               // every leaf applied to the two skip codes (leaf kernel -> entry walk)
    DevBuf<uint32_t> d_sync_rec;            // the block records of the walk from synchronising spans
    uint64_t pre_plane = 0;                 // elements per plane of d_pre (4 planes when that walk can run)
    uint32_t lt_merged_rows = 0;            // size of d_lt_merged without append_sync_codes' tables
    bool scan_sync = false;                 // try the walk from synchronising spans first
    // How the next scan is queued.  The walk alone while it works (the composing kernels behind it would only return
    // at once: five launches, ~25 us of the chain).  A run where it gives up is refused and queued again with the
    // composing kernels only; so are the next kSyncBackoff - 1 runs, then one run carries both, and its verdict
    // decides (a stream of captures of one kind settles in one form or the other).
    static constexpr uint32_t kSyncBackoff = 8;
    uint32_t sync_backoff = 0;              // runs left in the composing form
    uint64_t sync_min_edges = 20000;        // edge lists expected shorter than this are composed (OOKD_SYNC_MIN_EDGES: tests)
    uint32_t sync_mode = 0;                 // of the scan in flight: FsmScanArgs::sync_try
    std::vector<uint64_t> mixed_errs;       // error positions of a run whose refused captures were redone (host side)
    bool mixed_valid = false;
    bool front_grid = false;        // OOKD_RX_FRONT_GRID: one workgroup per wave tile instead of the streaming form
    uint32_t stream_waves = 12;     // persistent front-end waves per CU
    uint64_t front_launch_outputs = 1ull << 29;    // decimated samples per front-end grid launch (all captures together)

    // device (state machine)
    bool have_fsm = false;
    uint32_t num_bits = 0;
    DevBuf<FsmTablesDev> d_tables;
    DevBuf<uint32_t> d_big;         // packed tables of a device with more than 64 states / triggers (kernels.hpp)
    bool big_device = false;

    // capacity
    uint64_t max_samples = 0;
    uint32_t max_captures = 1;
    uint64_t max_n_in = 0, max_n_out = 0, max_words = 0;
    uint32_t max_blocks = 0, max_segs_per_cap = 0;
    uint64_t seg_len = 0;
    uint32_t msg_slots = 0, err_slots = 0;
    uint64_t edge_capacity = 0, msg_capacity = 0;

    DevBuf<uint64_t> d_bits;
    DevBuf<float> d_fir;
    DevBuf<int16_t> d_halo;
    DevBuf<uint32_t> d_blk_count, d_blk_offset, d_group_total;
    DevBuf<uint64_t> d_edges, d_seg_bounds;
    DevBuf<SegState> d_state_in, d_state_out;
    DevBuf<MsgDev> d_seg_msgs, d_msgs;
    DevBuf<uint32_t> d_seg_msg_count, d_seg_err_count;
    DevBuf<uint64_t> d_seg_errs;
    DevBuf<ResultHeader> d_hdr;
    DevBuf<uint64_t> d_debug;

    // scan form of the state machine (fsm_scan.hip)
    bool scan_ok = false;           // device fits the scan's tables and it was not disabled
    std::unique_ptr<FsmTablesDev> h_tables;     // host copy of the device tables
    bool scan_used = false;         // the results of the last run come from the scan
    bool scan_pending = false;      // a scan is queued; its verdict is read with the results
    bool pending_first_valid = false;
    FsmStateDev pending_first{};
    uint32_t scan_reach_base = 0;   // reach entries below the stuck codes
    uint32_t scan_reach_n = 0;      // reach entries; behind them the per-level lists (scan_reach_lv[level] codes each)
    uint32_t scan_reach_lv[2] = {0, 0};
    uint32_t scan_D = 0, scan_S = 0, scan_leaf_block = 0, scan_blocks_cap = 0;
    uint32_t scan_max_bits = 0;
    DevBuf<uint16_t> d_block_tab;
    DevBuf<uint32_t> d_lt_off, d_lt_n0, d_lt_pk;    // span tables (empty = the scan simulates)
    DevBuf<uint32_t> d_lt_merged;                    // build_merged_rows of them
    DevBuf<uint4> d_ltab;           // the scan kernels' LDS table image
    DevBuf<uint16_t> d_reach;       // abstract codes a span can be entered in (empty = all)
    DevBuf<uint32_t> d_cap_group_off, d_cap_super_off;
    DevBuf<uint16_t> d_group_tab, d_super_tab, d_super_in, d_cap_end;
    DevBuf<uint32_t> d_cap_block_off;
    DevBuf<LeafEvDev> d_events;
    DevBuf<uint32_t> d_ev_hot;
    DevBuf<uint8_t> d_app_vals;
    DevBuf<uint64_t> d_scan_errs;
    DevBuf<SegState> d_final_state;
    DevBuf<uint32_t> d_fin_off;
    DevBuf<uint64_t> d_fsum;                // 32 B per finish block: stamped aggregates
    uint32_t scan_fin_cap = 0;
    uint32_t scan_stamp = 0;        // stamps the finish kernel's block aggregates, never 0
    DevBuf<int16_t> d_stage_in;     // process_host staging (lazy)
    Ingest ingest;                  // its pinned double buffer (lazy)

    ResultHeader *h_hdr = nullptr;  // pinned
    MsgDev *h_msgs = nullptr;       // pinned, msg_capacity
    ResultHeader *h_hdr_dev = nullptr;      // the same two through the device's mapping
    MsgDev *h_msgs_dev = nullptr;
    bool hdr_dirty = true;          // device header needs zeroing before the next run
    uint64_t num_msgs = 0;

    // geometry of the last run
    uint32_t run_caps = 0;
    uint64_t run_n_valid = 0, run_n_in = 0, run_n_out = 0, run_words = 0;
    uint32_t run_blocks = 0, run_segs_per_cap = 0;
    uint32_t iter_next = 0;         // next FSM iteration number (parity continues)
    uint32_t final_parity = 0;
    ookd_rx_stats stats{};

    ookd_rx_gate *gate = nullptr;   // shared with other contexts by the caller, or null
    ~ookd_rx() {
        if (gate) {
            std::lock_guard<std::mutex> lock(gate->m);
            if (gate->last == ev[1]) gate->last = nullptr;      // ev[1] is destroyed below
        }
        (void)hipSetDevice(dev);
        d_taps.release();
        d_mfma_a.release();
        d_tables.release();
        d_big.release();
        d_bits.release();
        d_fir.release();
        d_halo.release();
        d_tile_info.release();
        d_ctl.release();
        d_quiet.release();
        d_carry.release();
        d_cap_fallback.release();
        d_pre.release();
        d_blk_in.release();
        d_rowz.release();
        d_skipc.release();
        d_sync_rec.release();
        d_chunk_totals.release();
        d_fin_tickets.release();
        for (auto &e : ev_c0) if (e) (void)hipEventDestroy(e);
        for (auto &e : ev_c1) if (e) (void)hipEventDestroy(e);
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_end) (void)hipEventDestroy(ev_end);
        for (hipStream_t d : dummy_streams) if (d) (void)hipStreamDestroy(d);
        if (s_front) (void)hipStreamDestroy(s_front);
        if (s_chain) (void)hipStreamDestroy(s_chain);
        d_blk_count.release();
        d_blk_offset.release();
        d_group_total.release();
        d_seg_bounds.release();
        d_edges.release();
        d_state_in.release();
        d_state_out.release();
        d_seg_msgs.release();
        d_msgs.release();
        d_seg_msg_count.release();
        d_seg_err_count.release();
        d_seg_errs.release();
        d_hdr.release();
        d_debug.release();
        d_block_tab.release();
        d_lt_off.release();
        d_lt_n0.release();
        d_lt_pk.release();
        d_lt_merged.release();
        d_ltab.release();
        d_reach.release();
        d_cap_group_off.release();
        d_group_tab.release();
        d_super_tab.release();
        d_super_in.release();
        d_cap_super_off.release();
        d_cap_end.release();
        d_cap_block_off.release();
        d_events.release();
        d_ev_hot.release();
        d_app_vals.release();
        d_scan_errs.release();
        d_final_state.release();
        d_fin_off.release();
        d_fsum.release();
        d_stage_in.release();
        if (h_hdr) (void)hipHostFree(h_hdr);
        if (h_msgs) (void)hipHostFree(h_msgs);
        for (auto &e : ev) {
            if (e) (void)hipEventDestroy(e);
        }
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }

    void geometry(uint64_t n_valid, bool pad_to_buffer, uint64_t &n_in, uint64_t &n_out,
                  uint64_t &words, uint32_t &blocks, uint32_t &segs) const {
        const uint64_t spb = cfg.samples_per_buffer;
        n_in = pad_to_buffer ? ((n_valid + spb - 1) / spb) * spb : n_valid;
        n_out = n_in / total_decim;
        const uint64_t tiles = (n_out + kFirTile - 1) / kFirTile;
        words = tiles * (kFirTile / 64);
        blocks = (uint32_t)(words / kBlockWords);
        segs = (uint32_t)((n_out + seg_len - 1) / seg_len);
        if (segs == 0 && n_out > 0) segs = 1;
    }

    FrontParams front_params(const void *d_iq, uint64_t stride) const {
        FrontParams p{};
        p.iq = static_cast<const int16_t *>(d_iq);
        p.cap_stride = stride;
        p.n_valid = run_n_valid;
        p.n_in = run_n_in;
        p.n_out = run_n_out;
        p.num_stages = num_stages;
        for (uint32_t s = 0; s < num_stages; ++s) p.stage[s] = stage[s];
        p.taps = d_taps.p;
        p.bits = d_bits.p;
        p.words_per_cap = run_words;
        p.fir_out = (cfg.flags & OOKD_RX_KEEP_FIR) ? d_fir.p : nullptr;
        p.p_star = p_star;
        p.p_lo = p_lo;
        p.p_hi = p_hi;
        p.mfma_a = d_mfma_a.p;
        p.mfma_c = mfma_c;
        p.p_lo_n = p_lo_n;
        p.p_hi_n = p_hi_n;
        p.p_lo_w = p_lo_w;
        p.p_hi_w = p_hi_w;
        p.mfma_g = mfma_g;
        p.mfma_xcd = mfma_xcd;
        {
            static const uint32_t dbg = dev_getenv("OOKD_MFMA_DEBUG") ? (uint32_t)atoi(dev_getenv("OOKD_MFMA_DEBUG")) : 0u;
            p.mfma_debug = dbg;
        }
        p.recompute_count = &d_hdr.p->recompute;
        p.quiet_lsb = quiet_lsb;
        p.quiet_count = count_quiet ? d_quiet.p : nullptr;
        p.tile_info = d_tile_info.p;
        {
            const uint32_t tile_bits = front_tile_bits(p);
            p.tiles_per_cap = tile_bits ? (uint32_t)(run_words * 64 / tile_bits) : 0;
        }
        p.sparse = sparse ? 1u : 0u;
        p.stamp_bits = tile_stamp << kTileStampShift;
        return p;
    }

    EdgeParams edge_params() const {
        EdgeParams e{};
        e.bits = d_bits.p;
        e.words_per_cap = run_words;
        e.n_out = run_n_out;
        e.num_captures = run_caps;
        e.blocks_per_cap = run_blocks;
        e.blk_count = d_blk_count.p;
        e.blk_offset = d_blk_offset.p;
        e.group_total = d_group_total.p;
        e.edges = d_edges.p;
        e.edge_capacity = edge_capacity;
        e.overflow = &d_hdr.p->edge_overflow;
        const uint32_t tile_bits = front_tile_bits(front_params(nullptr, 0));
        if (tile_bits && d_tile_info.p) {
            e.tile_info = d_tile_info.p;
            e.tiles_per_block = (uint32_t)(kBlockWords * 64) / tile_bits;
            e.stamp_bits = tile_stamp << kTileStampShift;
        }
        return e;
    }

    FsmParams fsm_params() const {
        FsmParams f{};
        f.tables = d_tables.p;
        f.big = big_device ? d_big.p : nullptr;
        f.big_words = big_device ? (uint32_t)d_big.n : 0u;
        f.bits = d_bits.p;
        f.words_per_cap = run_words;
        f.edges = d_edges.p;
        f.blk_offset = d_blk_offset.p;
        f.blocks_per_cap = run_blocks;
        f.num_captures = run_caps;
        f.n_out = run_n_out;
        f.spb = cfg.samples_per_buffer;
        f.total_decim = total_decim;
        f.seg_len = seg_len;
        f.segs_per_cap = run_segs_per_cap;
        f.seg_bounds = d_seg_bounds.p;
        f.msg_slots = msg_slots;
        f.err_slots = err_slots;
        f.state_in = d_state_in.p;
        f.state_out = d_state_out.p;
        f.seg_msgs = d_seg_msgs.p;
        f.seg_msg_count = d_seg_msg_count.p;
        f.seg_errs = d_seg_errs.p;
        f.seg_err_count = d_seg_err_count.p;
        f.changed = d_hdr.p->changed;
        f.flags = &d_hdr.p->flags;
        f.msgs = d_msgs.p;
        f.msg_capacity = msg_capacity;
        f.totals = d_hdr.p->totals;
        f.debug = d_debug.p;
        f.edge_overflow = &d_hdr.p->edge_overflow;
        const uint32_t tile_bits = front_tile_bits(front_params(nullptr, 0));
        if (tile_bits && d_tile_info.p) {
            f.tile_info = d_tile_info.p;
            f.tiles_per_cap = (uint32_t)(run_words * 64 / tile_bits);
            f.tile_shift = (uint32_t)__builtin_ctz(tile_bits);
            f.stamp_bits = tile_stamp << kTileStampShift;
        }
        return f;
    }

    uint32_t next_sync_mode() {
        if (!scan_sync) return sync_mode = 0;
        // a short edge list goes through the composing kernels faster: the walk ends with its longest region (a few
        // hundred leaves at 0.14 us, whatever the capture's size) -- chain 131 against 153 us for the 46 000 edges of a
        // 1 GiB bench capture, 275 against 400 for the 737 000 of 16 GiB, about even at 20 000.  By the edge count of
        // this context's last run; before there is one, by the capture's length
        const uint64_t expect = stats.num_edges ? stats.num_edges : ((uint64_t)run_n_out * run_caps) >> 12;
        if (expect < sync_min_edges) return sync_mode = 0;
        if (sync_backoff == 0) return sync_mode = 2;
        return sync_mode = (--sync_backoff == 0) ? 1u : 0u;
    }

    int front_and_edges(const void *d_iq, uint64_t stride, const int16_t *d_halo_ptr,
                        uint32_t halo_len);
    bool plan_chunks();
    int run_pipelined(const void *d_iq);
    int prepare_front(FrontParams &fp);
    int run_state_machine(const FsmStateDev *first, bool fresh);
    int fsm_scan(const FsmStateDev *first);
    int fsm_to_fixpoint(const FsmStateDev *first, bool fresh, bool force_first, const FsmParams *view = nullptr);
    int redo_refused_captures();
    int fetch_results();
    int enqueue_publish();
    PublishParams publish_params() const;
    bool scan_published = false;    // the queued scan ends with the publish step
    int collect_results();
    bool submitted = false;         // a run is queued (ookd_rx_submit_device) and not yet waited for
};

// A new run of the front end: the next stamp.  Sparse output leaves the quiet tiles' words and infos
// alone -- whatever they hold carries an older stamp and reads as quiet -- so nothing is zeroed between
// runs, except when the geometry changes (rare; keeps "a zero info means zero words" simple to reason
// about) or the 16-bit stamp wraps (every 2^16 - 1 runs): then every tile the buffer may hold is zeroed.
int ookd_rx::prepare_front(FrontParams &fp) {
    const uint32_t tile_bits = front_tile_bits(fp);
    if (!tile_bits) return OOKD_OK;
    if (sparse) {
        const uint64_t tiles = (uint64_t)run_caps * fp.tiles_per_cap;
        const bool same = tiles == dirty_tiles && fp.tiles_per_cap == dirty_tiles_per_cap && run_words == dirty_words_per_cap;
        if (tile_stamp >= kTileStampMax) {
            const uint64_t wpt = tile_bits / 64u;
            const uint64_t all = std::min<uint64_t>(std::min<uint64_t>(d_tile_info.n, d_bits.n / wpt), 0xfffffff8ull) & ~(uint64_t)7;
            // (the tile -> words mapping is linear: the whole buffer as one capture)
            HIPCHK(launch_clear_tiles(d_tile_info.p, d_bits.p, all, (uint32_t)all, all * wpt, tile_bits, 0u, stream));
            tile_stamp = 0;
        } else if (!same && dirty_tiles) {
            HIPCHK(launch_clear_tiles(d_tile_info.p, d_bits.p, dirty_tiles, dirty_tiles_per_cap, dirty_words_per_cap,
                                      tile_bits, 0u, stream));
        }
        dirty_tiles = tiles;
        dirty_tiles_per_cap = fp.tiles_per_cap;
        dirty_words_per_cap = run_words;
    } else if (tile_stamp >= kTileStampMax) {
        tile_stamp = 0;         // dense output rewrites every tile of a run
    }
    ++tile_stamp;
    bits_dense = !sparse;
    fp.stamp_bits = tile_stamp << kTileStampShift;
    return OOKD_OK;
}

// Readers of the raw bit words (ookd_rx_get_bits) want the quiet tiles zero: zero what earlier runs
// left in the current run's extents.
int ookd_rx::densify_bits() const {
    if (bits_dense || !sparse || !dirty_tiles) return OOKD_OK;
    const uint32_t tile_bits = front_tile_bits(front_params(nullptr, 0));
    HIPCHK(launch_clear_tiles(d_tile_info.p, d_bits.p, dirty_tiles, dirty_tiles_per_cap, dirty_words_per_cap, tile_bits,
                              tile_stamp << kTileStampShift, stream));
    HIPCHK(hipStreamSynchronize(stream));
    bits_dense = true;
    return OOKD_OK;
}

// ---------------------------------------------------------------------------
// chunk pipeline
// ---------------------------------------------------------------------------
// A long single capture is cut into chunks at multiples of lcm(4096 outputs,
// samples_per_buffer): chunk boundaries are buffer boundaries (the
// drop-rest-of-buffer rule stays chunk-local) and edge-block boundaries.  The
// front end of chunk c+1 (s_front) runs while the edges + state machine of
// chunk c (s_chain) run; each stream owns one half of the CUs, spread evenly
// over the XCDs -- the front end is HBM bound and as fast on half the chip
// (tools/cu_mask_probe.py), the chain's kernels no longer wait for a wave slot
// beside a grid that refills every one it frees.  Every chunk is a shard of
// the capture for the state machine: positions chunk-local, the level in
// front of it counted as 0, the incoming state = the chunk before's outgoing
// state, in device memory.  Messages / errors are appended behind the chunks
// before's.  A refusal of the scan anywhere sends the whole capture through
// the unchunked path.
bool ookd_rx::plan_chunks() {
    chunks.clear();
    if (!pipe_ok || no_pipeline_once || run_caps != 1 || run_n_out == 0) return false;
    const uint64_t spb = cfg.samples_per_buffer;
    // boundary (decimated index) o: o % 4096 == 0 and o * D % spb == 0
    const uint64_t per_buf = spb / gcd64(spb, total_decim);
    const uint64_t align = (uint64_t)kFirTile / gcd64(kFirTile, per_buf) * per_buf;
    const uint64_t target = std::max<uint64_t>(align, pipe_chunk_in / total_decim / align * align);
    if (run_n_out < 2 * target) return false;
    std::vector<uint64_t> sizes;
    uint64_t left = run_n_out;
    while (left > target + target / 2) {
        sizes.push_back(target);
        left -= target;
    }
    // the tail shrinks: the chain of the last chunk is all that is not hidden behind a front end
    const uint64_t tail_min = std::max<uint64_t>(align, kPipeTailChunk / total_decim / align * align);
    while (left > 2 * tail_min && sizes.size() + 2 < (size_t)kMaxChunks) {
        uint64_t half = left / 2 / align * align;
        if (half < tail_min) break;
        sizes.push_back(half);
        left -= half;
    }
    sizes.push_back(left);
    if (sizes.size() < 2 || sizes.size() > (size_t)kMaxChunks) return false;
    uint64_t at = 0, eoff = 0;
    for (uint64_t sz : sizes) {
        Chunk c{};
        c.out0 = at;
        c.nout = sz;
        c.blk0 = (uint32_t)(at / kFirTile);
        c.nblk = (uint32_t)((sz + kFirTile - 1) / kFirTile);
        c.edge_off = eoff;
        c.edge_cap = edge_capacity / run_blocks * c.nblk;       // its share of the list, by blocks
        eoff += c.edge_cap;
        at += sz;
        chunks.push_back(c);
    }
    return true;
}

int ookd_rx::run_pipelined(const void *d_iq) {
    const size_t nc = chunks.size();
    while (ev_c0.size() < nc) {
        hipEvent_t a = nullptr, b = nullptr;
        HIPCHK(hipEventCreate(&a));
        ev_c0.push_back(a);
        HIPCHK(hipEventCreate(&b));
        ev_c1.push_back(b);
    }
    if (hdr_dirty) HIPCHK(hipMemsetAsync(d_hdr.p, 0, sizeof(ResultHeader), stream));
    hdr_dirty = true;
    FrontParams fp = front_params(d_iq, run_n_valid);
    {
        const int rc = prepare_front(fp);
        if (rc != OOKD_OK) return rc;
    }
    HIPCHK(hipEventRecord(ev_start, stream));
    HIPCHK(hipStreamWaitEvent(s_front, ev_start, 0));
    HIPCHK(hipStreamWaitEvent(s_chain, ev_start, 0));

    const uint32_t tile_bits = front_tile_bits(fp);
    const uint32_t tiles_per_block = (uint32_t)kFirTile / tile_bits;
    scan_used = false;
    scan_pending = true;
    stats.fsm_path = 0;
    stats.scan_entry_form = 0;
    stats.fsm_fallback_reason = 0;
    pending_first_valid = false;
    // every front-end launch first: the device starts on the capture at once and is not held up by
    // the host still queueing the (ten times as many) chain kernels behind
    for (size_t c = 0; c < nc; ++c) {
        const Chunk &ch = chunks[c];
        HIPCHK(launch_front(fp, 1, exact, s_front, ev_c0[c], ev_c1[c], (uint64_t)ch.blk0 * tiles_per_block,
                            (uint64_t)ch.nblk * tiles_per_block));
    }
    for (size_t c = 0; c < nc; ++c) {
        const Chunk &ch = chunks[c];
        const bool last = c + 1 == nc;
        HIPCHK(hipStreamWaitEvent(s_chain, ev_c1[c], 0));
        // ---- its edges: chunk-local positions ----------------------------------------------------------
        EdgeParams e{};
        e.bits = d_bits.p + (size_t)ch.blk0 * kBlockWords;
        e.words_per_cap = (uint64_t)ch.nblk * kBlockWords;
        e.n_out = ch.nout;
        e.num_captures = 1;
        e.blocks_per_cap = ch.nblk;
        e.blk_count = d_blk_count.p + ch.blk0;
        e.blk_offset = d_blk_offset.p + ch.blk0 + c;            // nblk + 1 entries per chunk
        e.group_total = d_group_total.p;
        e.edges = d_edges.p + ch.edge_off;
        e.edge_capacity = ch.edge_cap;
        e.overflow = &d_hdr.p->edge_overflow;
        e.tile_info = d_tile_info.p + (size_t)ch.blk0 * tiles_per_block;
        e.tiles_per_block = tiles_per_block;
        e.total_acc = &d_hdr.p->total_edges;
        e.has_prev = c ? 1u : 0u;
        e.stamp_bits = fp.stamp_bits;
        HIPCHK(launch_edges(e, s_chain));
        // ---- its state machine, from the chunk before's outgoing state --------------------------------------
        FsmScanArgs a{};
        a.f = fsm_params();
        a.f.bits = e.bits;
        a.f.tile_info = e.tile_info;
        a.f.words_per_cap = e.words_per_cap;
        a.f.edges = e.edges;
        a.f.blk_offset = e.blk_offset;
        a.f.blocks_per_cap = ch.nblk;
        a.f.num_captures = 1;
        a.f.n_out = ch.nout;
        a.f.totals = last ? d_hdr.p->totals : d_chunk_totals.p + 2 * (c & 1);
        a.D = scan_D;
        a.S = scan_S;
        a.SNB = scan_S * (scan_max_bits + 2);
        a.leaf_block = scan_leaf_block;
        a.grid_blocks = 1024;
        a.block_tab = d_block_tab.p;
        a.lt_off = d_lt_off.p;
        a.lt_n0 = d_lt_n0.p;
        a.lt_pk = d_lt_pk.p;
        a.lt_words = (uint32_t)(d_lt_off.n + d_lt_n0.n + d_lt_pk.n);
        a.lt_merged = d_lt_merged.p;
        a.lt_merged_words = lt_merged_rows;
        a.lt_sync_words = (uint32_t)d_lt_merged.n;
        a.ltab = d_ltab.p;
        a.reach = d_reach.p;
        a.nreach = scan_reach_n;
        a.nreach_base = scan_reach_base;
        a.nreach_lv[0] = scan_reach_lv[0];
        a.nreach_lv[1] = scan_reach_lv[1];
        if (last) {
            a.publish = publish_params();
            a.publish.total_edges = nullptr;        // accumulated in the header by the edge stages
        } else {
            // messages go to the host as they are resolved; the header only with the last chunk
            PublishParams pp{};
            pp.d_msgs = reinterpret_cast<const uint4 *>(d_msgs.p);
            pp.h_msgs = reinterpret_cast<uint4 *>(h_msgs_dev);
            pp.first_msgs = std::min<uint64_t>(kHostMsgFirst, msg_capacity);
            a.publish = pp;
        }
        a.cap_group_off = d_cap_group_off.p;
        a.group_tab = d_group_tab.p;
        a.cap_super_off = d_cap_super_off.p;
        a.super_tab = d_super_tab.p;
        a.super_in = d_super_in.p;
        a.cap_end = d_cap_end.p;
        a.cap_first = d_cap_end.p + (max_captures + 8);
        a.cap_block_off = d_cap_block_off.p;
        a.total_blocks_cap = scan_blocks_cap;
        a.events = d_events.p;
        a.ev_hot = d_ev_hot.p;
        a.app_vals = d_app_vals.p;
        a.app_capacity = d_app_vals.n - 64;
        a.errs = d_scan_errs.p;
        a.err_capacity = d_scan_errs.n;
        a.first = nullptr;
        a.first_dev = c ? d_carry.p + ((c - 1) & 1) : nullptr;
        a.pos_origin = ch.out0;
        a.totals_in = c ? d_chunk_totals.p + 2 * ((c - 1) & 1) : nullptr;
        a.edge_overflow = &d_hdr.p->edge_overflow;
        a.pre_codes = d_pre.p;
        a.blk_in = d_blk_in.p;
        a.rowz = d_rowz.p;
        a.skipc = d_skipc.p;
    a.sync_rec = d_sync_rec.p;
    a.pre_plane = pre_plane;
    a.sync_fail = &d_hdr.p->sync_fail;
    a.sync_try = next_sync_mode();
        a.sync_rec = d_sync_rec.p;
        a.pre_plane = pre_plane;
        a.sync_fail = &d_hdr.p->sync_fail;
        a.sync_try = next_sync_mode();
        a.final_state = d_carry.p + (c & 1);
        a.fallback = &d_hdr.p->scan_fallback;
        a.fin_off = d_fin_off.p;
        a.fsum = d_fsum.p;
        a.fin_ticket = d_fin_tickets.p + c;
        if (++scan_stamp == 0) scan_stamp = 1;
        a.run_stamp = scan_stamp;
        a.fin_blocks_cap = scan_fin_cap;
        HIPCHK(launch_fsm_scan(a, s_chain, last ? ev[2] : nullptr));
    }
    scan_published = true;
    HIPCHK(hipEventRecord(ev_end, s_chain));
    HIPCHK(hipStreamWaitEvent(stream, ev_end, 0));
    return OOKD_OK;
}

int ookd_rx::front_and_edges(const void *d_iq, uint64_t stride, const int16_t *d_halo_ptr,
                             uint32_t halo_len) {
    if (hdr_dirty) HIPCHK(hipMemsetAsync(d_hdr.p, 0, sizeof(ResultHeader), stream));
    hdr_dirty = true;
    FrontParams fp = front_params(d_iq, stride);
    fp.halo = d_halo_ptr;
    fp.halo_len = halo_len;
    {
        const int rc = prepare_front(fp);
        if (rc != OOKD_OK) return rc;
    }
    if (!front_grid && front_streams(fp)) {
        // persistent streaming form: capped residency, other contexts' kernels run beside it
        HIPCHK(hipMemsetAsync(d_ctl.p, 0, sizeof(uint32_t) * kCtlChunkEnd, stream));
        StreamCtl ctl{};
        ctl.heads = d_ctl.p + kCtlHeads;
        ctl.done = nullptr;
        ctl.chunk_end = d_ctl.p + kCtlChunkEnd;
        ctl.num_chunks = 0;
        ctl.num_caps = run_caps;
        ctl.waves_per_cu = stream_waves;
        ctl.static_stride = dev_getenv("OOKD_STREAM_STRIDE") ? 1u : 0u;
        front_launches = 1;
        HIPCHK(launch_front_stream(fp, ctl, exact, false, stream, ev[0], ev[1]));
    } else {
        // The tuned kernels go out as several grid launches of front_launch_tiles wave tiles: while a
        // grid has workgroups left to dispatch, kernels of OTHER queues (the edges / state machine
        // chain of the capture before, on another context's stream) are hardly dispatched at all --
        // they get their turn when a front-end launch has drained (profiles/r02_pipeline_trace.txt),
        // and once dispatched they run beside the next launch.  ev[0] / ev[1] = start of the first,
        // end of the last launch.
        const uint32_t tile_bits = front_tile_bits(fp);
        const uint64_t tiles = tile_bits ? (uint64_t)fp.tiles_per_cap : 0;
        const uint64_t per = tile_bits ? std::max<uint64_t>(1, front_launch_outputs / tile_bits / std::max(1u, run_caps)) : 0;
        auto launch_all = [&]() -> hipError_t {
            front_launches = 1;
            if (!tile_bits || tiles <= per + per / 2) return launch_front(fp, run_caps, exact, stream, ev[0], ev[1]);
            front_launches = (uint32_t)((tiles + per - 1) / per);
            for (uint64_t t = 0; t < tiles; t += per) {
                const bool first = t == 0, last = t + per >= tiles;
                const hipError_t e = launch_front(fp, run_caps, exact, stream, first ? ev[0] : nullptr,
                                                  last ? ev[1] : nullptr, t, per);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        };
        if (gate) {
            // wait + launch + publish under the lock: the event must be on its way before another
            // context may wait for it
            std::lock_guard<std::mutex> lock(gate->m);
            hipEvent_t prev = gate->last;
            if (prev && prev != ev[1]) HIPCHK(hipStreamWaitEvent(stream, prev, 0));
            HIPCHK(launch_all());
            gate->last = ev[1];
        } else {
            HIPCHK(launch_all());
        }
    }
    if (run_n_out > 0) HIPCHK(launch_edges(edge_params(), stream));
    return OOKD_OK;
}

// Runs segment-parallel state machine rounds until no segment's incoming
// state changes.  fresh: (re)initialise every segment's assumed state.
// view: the geometry to run on instead of the whole run's (one capture of a batch: redo_refused_captures)
int ookd_rx::fsm_to_fixpoint(const FsmStateDev *first, bool fresh, bool force_first, const FsmParams *view) {
    if (!have_fsm || run_n_out == 0) {
        HIPCHK(hipEventRecord(ev[2], stream));
        return OOKD_OK;
    }
    const FsmParams fp = view ? *view : fsm_params();
    if (fresh) {
        HIPCHK(launch_fsm_prepare(fp, first, stream));
        iter_next = 0;
    } else if (force_first) {
        // refine: new incoming state for segment 0 of the (single) capture
        // (the FsmStateDev is the leading member of SegState; skip_to stays 0:
        // shards start on buffer boundaries)
        HIPCHK(hipMemcpyAsync(d_state_in.p, first, sizeof(FsmStateDev), hipMemcpyHostToDevice, stream));
    }
    uint32_t rounds = 0;
    uint32_t mode = fresh ? 0u : (force_first ? 2u : 1u);
    for (;;) {
        HIPCHK(hipMemsetAsync(d_hdr.p->changed, 0, sizeof(uint32_t) * kIterBatch, stream));
        const uint32_t batch_mode0 = mode;
        for (uint32_t i = 0; i < kIterBatch; ++i) {
            HIPCHK(launch_fsm_round(fp, iter_next & 1u, mode, i, stream));
            iter_next++;
            mode = 1;
        }
        HIPCHK(hipMemcpyAsync(h_hdr->changed, d_hdr.p->changed, sizeof(uint32_t) * kIterBatch,
                              hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        rounds += kIterBatch;
        if (getenv("OOKD_DEBUG")) {
            fprintf(stderr, "[ookd] fsm batch mode0=%u changed:", batch_mode0);
            for (uint32_t i = 0; i < kIterBatch; ++i) fprintf(stderr, " %u", h_hdr->changed[i]);
            fprintf(stderr, " (segments %u)\n", run_caps * run_segs_per_cap);
        }
        // converged once an incremental round reran nothing (a mode-0 round
        // reruns everything and reports no count; a mode-2 round counts the
        // forced first segments)
        bool conv = false;
        for (uint32_t i = 0; i < kIterBatch && !conv; ++i) {
            const bool incremental = !(i == 0 && batch_mode0 != 1u);
            if (incremental && h_hdr->changed[i] == 0) conv = true;
        }
        if (conv) break;
        // every round settles at least one more segment's incoming state, left to right: a capture of
        // S segments is through after at most S + 1 rounds -- more would be a bug, not a hard input
        // (captures are independent chains of segs_per_cap segments: the bound is per capture, and it IS the
        //  check -- a fix-point that has not closed by then is a bug, not a hard input)
        if (rounds > fp.segs_per_cap + 2 * kIterBatch) {
            set_error("state machine fix-point did not converge in %u rounds", rounds);
            return OOKD_ERR_ARG;
        }
    }
    if (d_debug.p && !view) {
        const size_t nseg = (size_t)run_caps * run_segs_per_cap;
        std::vector<uint64_t> dbg(nseg * 4);
        HIPCHK(hipMemcpy(dbg.data(), d_debug.p, dbg.size() * 8, hipMemcpyDeviceToHost));
        uint64_t turns = 0, cyc = 0, fused = 0, loads = 0, maxcyc = 0, maxturns = 0;
        for (size_t i = 0; i < nseg; ++i) {
            turns += dbg[4 * i];
            cyc += dbg[4 * i + 1];
            fused += dbg[4 * i + 2];
            loads += dbg[4 * i + 3];
            maxcyc = std::max(maxcyc, dbg[4 * i + 1]);
            maxturns = std::max(maxturns, dbg[4 * i]);
        }
        fprintf(stderr, "[ookd] fsm last-run per segment: turns avg %.1f max %llu, fused %.1f, window loads %.1f, "
                        "memtime ticks avg %.0f max %llu\n",
                (double)turns / nseg, (unsigned long long)maxturns, (double)fused / nseg, (double)loads / nseg,
                (double)cyc / nseg, (unsigned long long)maxcyc);
    }
    final_parity = (iter_next - 1) & 1u;
    stats.fsm_iterations = rounds;
    HIPCHK(launch_fsm_gather(fp, stream));
    if (!view) HIPCHK(hipEventRecord(ev[2], stream));
    return OOKD_OK;
}

// A batched run in which the scan refused some captures (their path left its model): the scan's
// results for all the others stand; each refused capture is redone alone in the round form and
// its messages / errors are merged in, in capture order.  (Round 1 sent the whole call -- every
// capture of the batch -- through the rounds.)
int ookd_rx::redo_refused_captures() {
    std::vector<uint32_t> flags(run_caps);
    HIPCHK(hipMemcpy(flags.data(), d_cap_fallback.p, run_caps * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint64_t first = std::min<uint64_t>(kHostMsgFirst, msg_capacity);
    const uint64_t total = std::min<uint64_t>(h_hdr->totals[0], msg_capacity);
    std::vector<MsgDev> merged(h_msgs, h_msgs + std::min(total, first));
    if (total > first) {
        merged.resize(total);
        HIPCHK(hipMemcpy(merged.data() + first, d_msgs.p + first, (total - first) * sizeof(MsgDev), hipMemcpyDeviceToHost));
    }
    mixed_errs.resize(std::min<uint64_t>(h_hdr->totals[1], d_scan_errs.n));
    if (!mixed_errs.empty()) {
        HIPCHK(hipMemcpy(mixed_errs.data(), d_scan_errs.p, mixed_errs.size() * 8, hipMemcpyDeviceToHost));
    }
    uint64_t nerr = h_hdr->totals[1];
    uint32_t reasons = 0, rounds = 0;
    for (uint32_t cap = 0; cap < run_caps; ++cap) {
        if (!flags[cap]) continue;
        reasons |= flags[cap];
        FsmParams one = fsm_params();
        one.bits += (size_t)cap * run_words;
        if (one.tile_info) one.tile_info += (size_t)cap * one.tiles_per_cap;
        one.blk_offset += (size_t)cap * run_blocks;     // (its entries are offsets into the one edge list)
        one.num_captures = 1;
        HIPCHK(hipMemsetAsync(d_hdr.p, 0, sizeof(ResultHeader), stream));
        int rc = fsm_to_fixpoint(nullptr, true, false, &one);
        if (rc != OOKD_OK) return rc;
        rounds += stats.fsm_iterations;
        ResultHeader hdr;
        HIPCHK(hipMemcpy(&hdr, d_hdr.p, sizeof(hdr), hipMemcpyDeviceToHost));
        if (hdr.flags & 1u) {
            set_error("a state machine segment produced more than %u messages "
                      "(raise ookd_rx_config.message_slots)", msg_slots);
            return OOKD_ERR_CAPACITY;
        }
        const uint64_t m = std::min<uint64_t>(hdr.totals[0], msg_capacity);
        const size_t at = merged.size();
        merged.resize(at + m);
        if (m) HIPCHK(hipMemcpy(merged.data() + at, d_msgs.p, m * sizeof(MsgDev), hipMemcpyDeviceToHost));
        for (size_t i = at; i < merged.size(); ++i) merged[i].capture = cap;
        // its errors, from the per-segment lists
        const size_t nseg = run_segs_per_cap;
        std::vector<uint32_t> counts(nseg);
        std::vector<uint64_t> errs(nseg * err_slots);
        HIPCHK(hipMemcpy(counts.data(), d_seg_err_count.p, nseg * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(errs.data(), d_seg_errs.p, errs.size() * 8, hipMemcpyDeviceToHost));
        for (size_t sgi = 0; sgi < nseg; ++sgi) {
            for (uint32_t i = 0; i < std::min<uint32_t>(counts[sgi], err_slots); ++i) mixed_errs.push_back(errs[sgi * err_slots + i]);
        }
        nerr += hdr.totals[1];
    }
    HIPCHK(hipMemsetAsync(d_hdr.p, 0, sizeof(ResultHeader), stream));
    HIPCHK(hipStreamSynchronize(stream));
    hdr_dirty = false;
    std::stable_sort(merged.begin(), merged.end(), [](const MsgDev &a, const MsgDev &b) { return a.capture < b.capture; });
    if (merged.size() > msg_capacity) {
        set_error("message list overflow: %zu messages, capacity %llu", merged.size(), (unsigned long long)msg_capacity);
        return OOKD_ERR_CAPACITY;
    }
    if (!merged.empty()) memcpy(h_msgs, merged.data(), merged.size() * sizeof(MsgDev));
    num_msgs = merged.size();
    stats.num_messages = merged.size();
    stats.num_errors = nerr;
    stats.fsm_path = 3;                 // the scan, with the rounds for what it refused
    stats.fsm_fallback_reason = reasons;
    stats.fsm_iterations = rounds;
    mixed_valid = true;
    return OOKD_OK;
}

// The state machine over the current edge list: scan form when possible,
// segment/round form otherwise (or when the scan refuses the capture).
int ookd_rx::run_state_machine(const FsmStateDev *first, bool fresh) {
    (void)fresh;
    scan_used = false;
    scan_pending = false;
    stats.fsm_path = 0;
    stats.scan_entry_form = 0;
    stats.fsm_fallback_reason = 0;
    if (!have_fsm || run_n_out == 0) {
        HIPCHK(hipEventRecord(ev[2], stream));
        return OOKD_OK;
    }
    bool try_scan = scan_ok;
    if (first && (first->cur >= scan_S || first->nbits > scan_max_bits + 1)) try_scan = false;
    if (try_scan) {
        // queued without a host sync; fetch_results() looks at the scan's verdict
        // and, if it refused the capture, runs the round path instead
        pending_first_valid = first != nullptr;
        if (first) pending_first = *first;
        int rc = fsm_scan(first);
        if (rc != OOKD_OK) return rc;
        scan_pending = true;
        return OOKD_OK;
    }
    int rc = fsm_to_fixpoint(first, true, false);
    if (rc != OOKD_OK) return rc;
    stats.fsm_path = 2;
    return OOKD_OK;
}

int ookd_rx::fsm_scan(const FsmStateDev *first) {
    FsmScanArgs a{};
    a.f = fsm_params();
    a.D = scan_D;
    a.S = scan_S;
    a.SNB = scan_S * (scan_max_bits + 2);
    a.leaf_block = scan_leaf_block;
    a.grid_blocks = 1024;
    a.block_tab = d_block_tab.p;
    a.lt_off = d_lt_off.p;
    a.lt_n0 = d_lt_n0.p;
    a.lt_pk = d_lt_pk.p;
    a.lt_words = (uint32_t)(d_lt_off.n + d_lt_n0.n + d_lt_pk.n);
    a.lt_merged = d_lt_merged.p;
    a.lt_merged_words = lt_merged_rows;
    a.lt_sync_words = (uint32_t)d_lt_merged.n;
    a.ltab = d_ltab.p;
    a.reach = d_reach.p;
    a.nreach = scan_reach_n;
    a.nreach_base = scan_reach_base;
    a.nreach_lv[0] = scan_reach_lv[0];
    a.nreach_lv[1] = scan_reach_lv[1];
    a.publish = publish_params();
    a.cap_group_off = d_cap_group_off.p;
    a.group_tab = d_group_tab.p;
    a.cap_super_off = d_cap_super_off.p;
    a.super_tab = d_super_tab.p;
    a.super_in = d_super_in.p;
    a.cap_end = d_cap_end.p;
    a.cap_first = d_cap_end.p + (max_captures + 8);
    a.cap_block_off = d_cap_block_off.p;
    a.total_blocks_cap = scan_blocks_cap;
    a.events = d_events.p;
    a.ev_hot = d_ev_hot.p;
    a.app_vals = d_app_vals.p;
    a.app_capacity = d_app_vals.n - 64;     // fin_msg reads whole 8-byte groups
    a.errs = d_scan_errs.p;
    a.err_capacity = d_scan_errs.n;
    a.first = first;
    a.final_state = d_final_state.p;
    a.fallback = &d_hdr.p->scan_fallback;
    if (run_caps > 1) {
        HIPCHK(hipMemsetAsync(d_cap_fallback.p, 0, run_caps * sizeof(uint32_t), stream));
        a.cap_fallback = d_cap_fallback.p;
    }
    a.edge_overflow = &d_hdr.p->edge_overflow;
    a.pre_codes = d_pre.p;
    a.blk_in = d_blk_in.p;
    a.rowz = d_rowz.p;
    a.skipc = d_skipc.p;
    a.sync_rec = d_sync_rec.p;
    a.pre_plane = pre_plane;
    a.sync_fail = &d_hdr.p->sync_fail;
    a.sync_try = next_sync_mode();
    a.fin_off = d_fin_off.p;
    a.fsum = d_fsum.p;
    a.fin_ticket = d_fin_tickets.p + kMaxChunks;
    if (++scan_stamp == 0) scan_stamp = 1;
    a.run_stamp = scan_stamp;
    a.fin_blocks_cap = scan_fin_cap;
    // totals / scan_fallback are still zero from the header memset of front_and_edges
    HIPCHK(launch_fsm_scan(a, stream, ev[2]));      // ev[2]: end of its last kernel
    scan_published = true;          // its last kernel also publishes
    static const char *const debug_scan = getenv("OOKD_DEBUG_SCAN");     // read once: this is the hot path
    if (debug_scan) HIPCHK(hipStreamSynchronize(stream));
    if (debug_scan && d_debug.p) {
        uint64_t dbg[64];
        HIPCHK(hipMemcpy(dbg, d_debug.p, sizeof(dbg), hipMemcpyDeviceToHost));
        fprintf(stderr, "[scan] block_sims phases: resume %llu gap+rep %llu uniq %llu sims %llu\n",
                (unsigned long long)(dbg[41] - dbg[40]), (unsigned long long)(dbg[42] - dbg[41]),
                (unsigned long long)(dbg[43] - dbg[42]), (unsigned long long)(dbg[44] - dbg[43]));
        {
            const uint64_t *c = dbg + 48;       // the leaf kernel's stamps (block 2)
            fprintf(stderr, "[scan] leaf (wave form) block 2: rows %llu stuck marks %llu compose %llu stuck starts %llu block table %llu ticks\n",
                    (unsigned long long)(c[1] - c[0]), (unsigned long long)(c[2] - c[1]), (unsigned long long)(c[3] - c[2]),
                    (unsigned long long)(c[4] - c[3]), (unsigned long long)(c[5] - c[4]));
        }
        uint64_t dbg2[8];
        HIPCHK(hipMemcpy(dbg2, d_debug.p + 56, sizeof(dbg2), hipMemcpyDeviceToHost));
        HIPCHK(hipMemset(d_debug.p + 56, 0, sizeof(dbg2)));
        fprintf(stderr, "[scan] sync walk: %llu arrivals outside the image; first: block %u cand %u code %u len %u, image %08x %08x, info %08x next %08x, "
                        "next block %u f0 %u, own cands %08x %08x; nothing to hold on to at block %u (%u)\n",
                (unsigned long long)dbg2[0], (uint32_t)dbg2[1], (uint32_t)(dbg2[1] >> 32), (uint32_t)dbg2[2], (uint32_t)(dbg2[2] >> 32),
                (uint32_t)dbg2[3], (uint32_t)(dbg2[3] >> 32), (uint32_t)dbg2[4], (uint32_t)(dbg2[4] >> 32), (uint32_t)dbg2[5],
                (uint32_t)(dbg2[5] >> 32), (uint32_t)dbg2[6], (uint32_t)(dbg2[6] >> 32), (uint32_t)dbg2[7], (uint32_t)(dbg2[7] >> 32));
        for (int i = 0; i < 4; ++i)
            fprintf(stderr, "[scan] leaf block %d: sims %llu expand %llu compose %llu ticks, %llu unique spans, cap %llx\n", i,
                    (unsigned long long)dbg[4 * i], (unsigned long long)dbg[4 * i + 1],
                    (unsigned long long)dbg[4 * i + 2], (unsigned long long)(dbg[4 * i + 3] & 0xffffffffu),
                    (unsigned long long)(dbg[4 * i + 3] >> 32));
    }
    if (debug_scan && debug_scan[0] == '2') {
        uint32_t off[2];
        HIPCHK(hipMemcpy(off, d_cap_block_off.p, 8, hipMemcpyDeviceToHost));
        uint32_t ne32[2];
        HIPCHK(hipMemcpy(&ne32[0], d_blk_offset.p, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&ne32[1], d_blk_offset.p + run_blocks, 4, hipMemcpyDeviceToHost));
        const uint32_t ne = ne32[1] - ne32[0];
        fprintf(stderr, "[scan] blocks %u..%u ne %u D %u LB %u\n", off[0], off[1], ne, scan_D, scan_leaf_block);
        std::vector<LeafEvDev> evs(ne + 1);
        HIPCHK(hipMemcpy(evs.data(), d_events.p, (ne + 1) * sizeof(LeafEvDev), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i <= ne && i < 14; ++i)
            fprintf(stderr, "[scan] leaf %u napp %u nout %u nerr %u flags %u end cur %u k %u prev %u\n", i,
                    evs[i].napp, evs[i].nout, evs[i].nerr, evs[i].flags, evs[i].end_cur, evs[i].end_k, evs[i].end_prev);
    }
    return OOKD_OK;
}

int ookd_rx::fetch_results() {
    int rc = enqueue_publish();
    if (rc != OOKD_OK) return rc;
    return collect_results();
}

PublishParams ookd_rx::publish_params() const {
    static_assert(sizeof(ResultHeader) % 4 == 0 && sizeof(ResultHeader) / 4 <= 256, "header is published by one workgroup");
    PublishParams pp{};
    pp.d_hdr = reinterpret_cast<uint32_t *>(d_hdr.p);
    pp.h_hdr = reinterpret_cast<uint32_t *>(h_hdr_dev);
    pp.hdr_words = sizeof(ResultHeader) / 4;
    pp.totals_word = offsetof(ResultHeader, totals) / 4;
    pp.edges_word = offsetof(ResultHeader, total_edges) / 4;
    pp.done_word = offsetof(ResultHeader, publish_done) / 4;
    pp.total_edges = run_n_out > 0 ? d_blk_offset.p + (size_t)run_caps * run_blocks : nullptr;
    if (have_fsm && run_n_out > 0) {
        pp.d_msgs = reinterpret_cast<const uint4 *>(d_msgs.p);
        pp.h_msgs = reinterpret_cast<uint4 *>(h_msgs_dev);
    }
    pp.first_msgs = std::min<uint64_t>(kHostMsgFirst, msg_capacity);
    return pp;
}

// Last kernel of a run: header and first messages into pinned host memory (the
// scan form's last kernel has already done it).
int ookd_rx::enqueue_publish() {
    if (scan_published) {
        scan_published = false;
        return OOKD_OK;
    }
    HIPCHK(launch_publish(publish_params(), stream));
    return OOKD_OK;
}

// Waits for the run queued on the stream and turns it into host-side results.
int ookd_rx::collect_results() {
    const uint64_t first = std::min<uint64_t>(kHostMsgFirst, msg_capacity);
    uint32_t total_edges = 0;
    bool redo_some = false;             // the scan left some captures of the batch to the rounds
    mixed_valid = false;
    HIPCHK(hipStreamSynchronize(stream));
    hdr_dirty = false;              // the publish kernel left the device header zeroed
    if (scan_pending) {
        scan_pending = false;
        if (h_hdr->scan_fallback == 0) {
            scan_used = true;
            stats.fsm_path = 1;
            stats.scan_entry_form = (h_hdr->sync_fail & 3u) == 2u ? 1u : 2u;
            if (sync_mode == 1 && (h_hdr->sync_fail & 1u)) sync_backoff = kSyncBackoff;     // tried with both queued: not yet
            redo_some = run_caps > 1 && (h_hdr->flags & 2u) != 0;
        } else if (h_hdr->scan_fallback == kScanFbSync && !h_hdr->edge_overflow && chunks.empty()) {
            // the walk from synchronising spans gave up and nothing was queued behind it: the scan again, composing
            // (and so for the next runs of this context)
            sync_backoff = kSyncBackoff;
            ResultHeader keep = *h_hdr;
            keep.totals[0] = keep.totals[1] = 0;
            keep.scan_fallback = 0;
            keep.sync_fail = 0;
            keep.publish_done = 0;          // (the scan's last kernel counts its workgroups here to find the one that publishes)
            HIPCHK(hipMemcpyAsync(d_hdr.p, &keep, sizeof(keep), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            hdr_dirty = true;
            int rc = run_state_machine(pending_first_valid ? &pending_first : nullptr, true);
            if (rc != OOKD_OK) return rc;
            return fetch_results();
        } else if (h_hdr->edge_overflow) {
            // refused because the edge list overflowed: there is nothing to run the rounds on (blk_offset counts
            // edges that were never written); reported below
            stats.fsm_fallback_reason = h_hdr->scan_fallback;
        } else if (!chunks.empty()) {
            if (h_hdr->scan_fallback & kScanFbSync) sync_backoff = kSyncBackoff;
            // the scan refused a chunk of a pipelined run: the whole capture again, unchunked (the
            // last chunk's publishing kernel left the device header zeroed)
            stats.fsm_fallback_reason = h_hdr->scan_fallback;
            if (getenv("OOKD_DEBUG")) {
                fprintf(stderr, "[ookd] pipelined run refused (reason %u), running the capture whole\n", h_hdr->scan_fallback);
            }
            chunks.clear();
            no_pipeline_once = true;
            const uint32_t reason = stats.fsm_fallback_reason;
            int rc = front_and_edges(last_iq, last_stride, nullptr, 0);
            if (rc == OOKD_OK) rc = run_state_machine(nullptr, true);
            if (rc == OOKD_OK) rc = enqueue_publish();
            no_pipeline_once = false;
            if (rc != OOKD_OK) return rc;
            rc = collect_results();
            if (stats.fsm_fallback_reason == 0) stats.fsm_fallback_reason = reason;
            return rc;
        } else {
            // the scan refused this capture: run the round path and fetch again
            stats.fsm_fallback_reason = h_hdr->scan_fallback;
            if (getenv("OOKD_DEBUG")) {
                fprintf(stderr, "[ookd] fsm scan refused (reason %u), using rounds\n", h_hdr->scan_fallback);
            }
            // the publish kernel zeroed the device header: put the front end's
            // counters back (minus the scan's verdict and totals) for the second pass
            ResultHeader keep = *h_hdr;
            keep.totals[0] = keep.totals[1] = 0;
            keep.scan_fallback = 0;
            keep.sync_fail = 0;
            HIPCHK(hipMemcpyAsync(d_hdr.p, &keep, sizeof(keep), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            hdr_dirty = true;
            int rc = fsm_to_fixpoint(pending_first_valid ? &pending_first : nullptr, true, false);
            if (rc != OOKD_OK) return rc;
            stats.fsm_path = 3;
            return fetch_results();
        }
    }
    total_edges = run_n_out > 0 ? h_hdr->total_edges : 0;
    num_msgs = 0;
    stats.num_edges = total_edges;
    stats.num_messages = 0;
    stats.num_errors = 0;
    stats.guard_recomputes = h_hdr->recompute;
    stats.total_waves = front_wave_tiles(front_params(nullptr, 0)) * run_caps;
    if (stats.total_waves) {
        if (count_quiet) {
            // running counters (never zeroed: a memset per run is a 20 us fill kernel on the critical path):
            // this run's share is the difference to what the run before left, modulo 2^32 per counter
            std::vector<uint32_t> q(kQuietCounters);
            HIPCHK(hipMemcpyAsync(q.data(), d_quiet.p, q.size() * 4, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (quiet_prev.size() != q.size()) quiet_prev.assign(q.size(), 0u);
            for (size_t i = 0; i < q.size(); ++i) {
                stats.quiet_waves += (uint32_t)(q[i] - quiet_prev[i]);
                quiet_prev[i] = q[i];
            }
        }
    }
    stats.input_samples = run_n_in;
    stats.decimated_samples = run_n_out;
    stats.num_segments = run_caps * run_segs_per_cap;
    float ms = 0.0f;
    if (!chunks.empty()) {
        // pipelined: the front-end kernels of all chunks; first kernel start -> last kernel end
        stats.pipeline_chunks = (uint32_t)chunks.size();
        stats.front_launches = (uint32_t)chunks.size();
        float sum = 0.0f;
        for (size_t c = 0; c < chunks.size(); ++c) {
            if (hipEventElapsedTime(&ms, ev_c0[c], ev_c1[c]) == hipSuccess) sum += ms;
        }
        stats.fir_kernel_ms = sum;
        if (hipEventElapsedTime(&ms, ev_c0[0], ev[2]) == hipSuccess) stats.total_device_ms = ms;
    } else {
        stats.front_launches = front_launches;
        if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) stats.fir_kernel_ms = ms;
        if (hipEventElapsedTime(&ms, ev[0], ev[2]) == hipSuccess) stats.total_device_ms = ms;
    }
    if (h_hdr->edge_overflow) {
        set_error("edge list overflow: %u level changes found, capacity %llu "
                  "(raise ookd_rx_config.edge_capacity)",
                  total_edges, (unsigned long long)edge_capacity);
        return OOKD_ERR_CAPACITY;
    }
    if (have_fsm && run_n_out > 0) {
        if (!scan_used && (h_hdr->flags & 1u)) {
            set_error("a state machine segment produced more than %u messages "
                      "(raise ookd_rx_config.message_slots)", msg_slots);
            return OOKD_ERR_CAPACITY;
        }
        const uint64_t total = h_hdr->totals[0];
        if (total > msg_capacity) {
            set_error("message list overflow: %llu messages, capacity %llu",
                      (unsigned long long)total, (unsigned long long)msg_capacity);
            return OOKD_ERR_CAPACITY;
        }
        if (total > first) {
            // (on the context's stream, not the null stream: that one joins every blocking stream of the process)
            HIPCHK(hipMemcpyAsync(h_msgs + first, d_msgs.p + first, (total - first) * sizeof(MsgDev),
                                  hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        num_msgs = total;
        stats.num_messages = total;
        stats.num_errors = h_hdr->totals[1];
        if (redo_some) {
            const int rc = redo_refused_captures();
            if (rc != OOKD_OK) return rc;
        }
        // (the scan's finish kernel numbers message slots capture-major, the round
        //  form's gather does too: the list is already in capture order)
    }
    return OOKD_OK;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

ookd_rx *ookd_rx_create(const ookd_rx_config *cfg, const ookd_filter *filter,
                        const ookd_device *device) {
    clear_error();
    if (!cfg || cfg->samples_per_buffer == 0 || cfg->max_samples == 0) {
        set_error("ookd_rx_create: samples_per_buffer and max_samples must be non-zero");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: libookiedokie_amd has no CPU fallback");
        return nullptr;
    }
    if (cfg->hip_device < 0 || cfg->hip_device >= ndev) {
        set_error("hip_device %d out of range (%d devices)", cfg->hip_device, ndev);
        return nullptr;
    }
    std::unique_ptr<ookd_rx> rx(new ookd_rx());
    rx->cfg = *cfg;
    rx->dev = cfg->hip_device;
    if (hipSetDevice(rx->dev) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", rx->dev);
        return nullptr;
    }
    if (cfg->stream) {
        rx->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
        if (hipStreamCreateWithFlags(&rx->stream, hipStreamNonBlocking) != hipSuccess) {
            set_error("hipStreamCreate failed");
            return nullptr;
        }
        rx->own_stream = true;
    }
    for (auto &e : rx->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            set_error("hipEventCreate failed");
            return nullptr;
        }
    }
    rx->exact = (cfg->flags & OOKD_RX_EXACT_FIR) != 0;
    rx->max_captures = cfg->max_captures ? cfg->max_captures : 1;
    rx->max_samples = cfg->max_samples;

    // ---- filter ---------------------------------------------------------------
    std::vector<float> taps_dev;
    if (filter) {
        if (filter->stages.size() > (size_t)kMaxStages) {
            set_error("filter has %zu stages, this build supports %d", filter->stages.size(), kMaxStages);
            return nullptr;
        }
        rx->num_stages = (uint32_t)filter->stages.size();
        rx->total_decim = filter->total_decimation;
        uint64_t mult = 1;
        for (uint32_t s = 0; s < rx->num_stages; ++s) {
            const auto &st = filter->stages[s];
            FirStageDev d{};
            d.decim = st.decimation;
            d.ntaps = (uint32_t)st.taps.size();
            d.ntaps_pad = ((d.ntaps + kTapChunk - 1) / kTapChunk) * kTapChunk;
            d.tap_off = (uint32_t)taps_dev.size();
            taps_dev.insert(taps_dev.end(), st.taps.begin(), st.taps.end());
            // zero padding keeps sums bit-identical: acc + (+-0) == acc
            taps_dev.resize(d.tap_off + d.ntaps_pad, 0.0f);
            rx->stage[s] = d;
            rx->halo_needed += (uint64_t)(d.ntaps - 1) * mult;
            mult *= d.decim;
        }
        rx->taps0 = filter->stages[0].taps;
        if (rx->d_taps.alloc(taps_dev.size()) != OOKD_OK) return nullptr;
        if (hipMemcpy(rx->d_taps.p, taps_dev.data(), taps_dev.size() * sizeof(float),
                      hipMemcpyHostToDevice) != hipSuccess) {
            set_error("tap upload failed");
            return nullptr;
        }
    }
    rx->p_star = power_threshold(cfg->threshold);
    rx->p_lo = rx->p_hi = rx->p_star;
    if (!rx->exact && filter) {
        // used by the tuned kernels (1 stage / decimation 1, and 2 x decimation 2); the
        // generic kernel always computes in reference order and ignores the band
        std::vector<std::vector<float>> st;
        for (const auto &f : filter->stages) st.push_back(f.taps);
        guard_band(st, rx->p_star, rx->p_lo, rx->p_hi);
        // 1 stage, decimation 1, <= 256 taps: the product runs on the matrix cores (fir_mfma.hip) unless the
        // caller asks for the packed-VALU loop (or for the experimental streaming form, which only exists for it)
        MfmaTaps mt;
        bool use_mfma = false;
        if (rx->num_stages == 2 && rx->stage[0].decim == 2 && rx->stage[1].decim == 2 && !(cfg->flags & OOKD_RX_FIR_VALU) &&
            !dev_getenv("OOKD_FIR_VALU") &&
            mfma_prepare_taps2(filter->stages[0].taps.data(), rx->stage[0].ntaps, filter->stages[1].taps.data(),
                               rx->stage[1].ntaps, mt)) {
            // the backend default shape (two decimate-by-2 stages) folded into one decimate-by-4 product
            float lo_n, hi_n, lo_w, hi_w;
            band_from_error(mfma_error_bound2(mt, guard_error(st, 1.0), false), rx->p_star, lo_n, hi_n);
            band_from_error(mfma_error_bound2(mt, guard_error(st, 16.0), true), rx->p_star, lo_w, hi_w);
            use_mfma = mfma_scale_band(mt, lo_n, rx->p_lo_n) && mfma_scale_band(mt, hi_n, rx->p_hi_n) &&
                       mfma_scale_band(mt, lo_w, rx->p_lo_w) && mfma_scale_band(mt, hi_w, rx->p_hi_w);
            if (rx->p_star > 0.0f && !(rx->p_star >= 0x1p-100f && rx->p_star <= 0x1p100f)) use_mfma = false;
        }
        if (rx->num_stages == 1 && rx->stage[0].decim == 1 && !(cfg->flags & OOKD_RX_FIR_VALU) &&
            !dev_getenv("OOKD_FRONT_STREAM") && !dev_getenv("OOKD_FIR_VALU") &&
            mfma_prepare_taps(filter->stages[0].taps.data(), rx->stage[0].ntaps, mt)) {
            float lo_n, hi_n, lo_w, hi_w;
            band_from_error(mfma_error_bound(mt, rx->stage[0].ntaps, false), rx->p_star, lo_n, hi_n);
            band_from_error(mfma_error_bound(mt, rx->stage[0].ntaps, true), rx->p_star, lo_w, hi_w);
            // the kernel compares in accumulator units; thresholds so far from the filter's range that the
            // power-of-two scaling leaves the normal floats stay on the packed-VALU loop
            use_mfma = mfma_scale_band(mt, lo_n, rx->p_lo_n) && mfma_scale_band(mt, hi_n, rx->p_hi_n) &&
                       mfma_scale_band(mt, lo_w, rx->p_lo_w) && mfma_scale_band(mt, hi_w, rx->p_hi_w);
            // fl(y^2) = c^2 fl(z^2) needs y^2 clear of the subnormals (and of overflow) wherever it decides a bit
            if (rx->p_star > 0.0f && !(rx->p_star >= 0x1p-100f && rx->p_star <= 0x1p100f)) use_mfma = false;
        }
        if (use_mfma) {
            if (rx->d_mfma_a.alloc(mt.image.size()) != OOKD_OK) return nullptr;
            if (hipMemcpy(rx->d_mfma_a.p, mt.image.data(), mt.image.size() * sizeof(uint16_t),
                          hipMemcpyHostToDevice) != hipSuccess) {
                set_error("tap image upload failed");
                return nullptr;
            }
            rx->mfma_c = mt.c;
            // wave tiles per wave of a workgroup: more for the long filters, whose workgroups fill a CU and
            // fetch a 20 / 36 KB image each (config2 sweep: 474 / 545 / 599 / 623 / 635 Gsamples/s at 2 / 4 / 8 / 16 / 32)
            rx->mfma_g = mt.ksteps <= 6 ? 4u : mt.ksteps <= 10 ? 16u : 32u;
            if (const char *g = dev_getenv("OOKD_MFMA_G")) rx->mfma_g = (uint32_t)std::min(4096, std::max(1, atoi(g)));
            // one contiguous run of tiles per XCD: where the halo is a good share of a tile's window -- the decimate-by-4
            // filter (96 of 1120 samples: 3.0 -> 2.8-2.9 ms per 16 GiB) and the long 1-stage filters (272 of 1296: 1 %)
            rx->mfma_xcd = 2u | (mt.ksteps >= 10 ? 1u : 0u);
            if (const char *x = dev_getenv("OOKD_MFMA_XCD")) rx->mfma_xcd = (uint32_t)atoi(x);
        }
    }
    if (filter && cfg->threshold > 0.0f && std::isfinite(cfg->threshold) && !(cfg->flags & OOKD_RX_NO_QUIET_SKIP)) {
        // |y_re|, |y_im| <= S * m with S = prod over stages of sum|h|, m = max |component| in
        // the window, so |y| <= sqrt(2) * S * m; 0.1 % slack covers every rounding of the
        // reference's float arithmetic (relative 1e-5 at most) many times over
        double S = 1.0;
        for (const auto &st : filter->stages) {
            double ss = 0.0;
            for (float t : st.taps) ss += std::fabs((double)t);
            S *= ss;
        }
        if (S > 0.0) {
            // |v| < quiet_lsb  <=>  |v|/2048 < level (complexf.h:68-77 scaling)
            const double lvl = (double)cfg->threshold * 0.999 / (1.41421356237309515 * S) * 2048.0;
            rx->quiet_lsb = lvl >= 32767.0 ? 32767 : (int)std::ceil(lvl);
        }
    }

    // ---- state machine ------------------------------------------------------------
    if (device) {
        const size_t ns = device->state_duration_us.size();
        const size_t nt = device->trig_cond.size();
        const bool big = ns > (size_t)kMaxStates || nt > (size_t)kMaxTriggers;
        if (ns > (size_t)kMaxStatesBig || nt > (size_t)kMaxTriggersBig || device->num_bits > 254) {
            set_error("device has %zu states / %zu triggers / %u bits, this build supports %d / %d / 254", ns, nt,
                      device->num_bits, kMaxStatesBig, kMaxTriggersBig);
            return nullptr;
        }
        // the kernel counts elapsed samples in 32 bits (saturating)
        auto too_long = [](const std::vector<uint64_t> &v) {
            for (uint64_t x : v) {
                if (x != ~0ull && x >= (1ull << 31)) return true;
            }
            return false;
        };
        if (too_long(device->state_kmin) || too_long(device->state_kmax) || too_long(device->state_kto) ||
            too_long(device->trig_kmin) || too_long(device->trig_kmax)) {
            set_error("a device duration/timeout exceeds 2^31 samples at this rate: unsupported");
            return nullptr;
        }
        std::unique_ptr<FsmTablesDev> t(new FsmTablesDev());
        memset(t.get(), 0, sizeof(FsmTablesDev));
        fill_fsm_tables(*device, *t);
        if (rx->d_tables.alloc(1) != OOKD_OK) return nullptr;
        if (hipMemcpy(rx->d_tables.p, t.get(), sizeof(FsmTablesDev), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("table upload failed");
            return nullptr;
        }
        if (big) {
            // more than 64 states / triggers: the round form with the tables in LDS (the scan's tables and the
            // lane-resident ones stop at 64)
            const std::vector<uint32_t> w = big_tables(*device, t->quiet_state);
            if (rx->d_big.alloc(w.size()) != OOKD_OK) return nullptr;
            if (hipMemcpy(rx->d_big.p, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
                set_error("table upload failed");
                return nullptr;
            }
            rx->big_device = true;
        }
        rx->have_fsm = true;
        rx->num_bits = device->num_bits;
        rx->h_tables = std::move(t);
    }

    // ---- capacities -------------------------------------------------------------------
    const uint64_t spb = cfg->samples_per_buffer;
    // nominal decimated samples per state machine segment (~2^19 by default;
    // segment_buffers expresses it in input buffers)
    if (cfg->segment_buffers) {
        rx->seg_len = std::max<uint64_t>(1, (uint64_t)cfg->segment_buffers * spb / rx->total_decim);
    } else {
        rx->seg_len = 1ull << 19;
    }
    rx->seg_len = std::min<uint64_t>(rx->seg_len, 1ull << 30);   // 32-bit offsets inside a segment
    rx->msg_slots = cfg->message_slots ? cfg->message_slots : 32;
    rx->err_slots = 32;
    uint32_t blocks = 0, segs = 0;
    rx->geometry(rx->max_samples, true, rx->max_n_in, rx->max_n_out, rx->max_words, blocks, segs);
    rx->max_blocks = blocks;
    rx->max_segs_per_cap = std::max<uint32_t>(segs, 1);
    const uint64_t total_out = rx->max_n_out * rx->max_captures;
    rx->edge_capacity = cfg->edge_capacity ? cfg->edge_capacity : total_out / 32 + (1u << 20);
    if (rx->edge_capacity > 0xfffffff0ull) rx->edge_capacity = 0xfffffff0ull;
    rx->msg_capacity = cfg->message_capacity
                           ? cfg->message_capacity
                           : std::max<uint64_t>(1u << 16, (uint64_t)rx->max_captures * 64);

    const size_t caps = rx->max_captures;
    const size_t nseg = caps * rx->max_segs_per_cap;
    int rc = OOKD_OK;
    rc |= rx->d_bits.alloc(caps * rx->max_words + 64);
    if (cfg->flags & OOKD_RX_KEEP_FIR) rc |= rx->d_fir.alloc(2 * caps * rx->max_n_out + 2);
    rc |= rx->d_halo.alloc(2 * (rx->halo_needed + 4));
    rc |= rx->d_blk_count.alloc(caps * blocks + 1);
    rc |= rx->d_tile_info.alloc(caps * blocks * 16 + 16);       // smallest wave tile: 256 bits
    rc |= rx->d_ctl.alloc(kCtlWords);
    // the streaming (persistent) form is experimental: it caps the front end's residency, but is
    // slower than the hardware-dispatched grid; OOKD_FRONT_STREAM=1 selects it
    rx->front_grid = (cfg->flags & OOKD_RX_FRONT_GRID) != 0 || !dev_getenv("OOKD_FRONT_STREAM");
    if (const char *w = dev_getenv("OOKD_STREAM_WAVES")) rx->stream_waves = (uint32_t)std::max(1, atoi(w));
    if (const char *w = dev_getenv("OOKD_FRONT_LAUNCH_LOG2")) rx->front_launch_outputs = 1ull << std::min(40, std::max(16, atoi(w)));
    rc |= rx->d_blk_offset.alloc(caps * blocks + 1 + kMaxChunks);     // (a pipelined run keeps one total per chunk)
    rc |= rx->d_group_total.alloc((caps * blocks + kScanGroup - 1) / kScanGroup + 1);
    rc |= rx->d_edges.alloc(rx->edge_capacity + 64);
    rc |= rx->d_hdr.alloc(1);
    rx->count_quiet = (cfg->flags & OOKD_RX_COUNT_QUIET) != 0;
    if (rx->count_quiet) rc |= rx->d_quiet.alloc(kQuietCounters);
    if (rx->count_quiet && rc == OOKD_OK && hipMemset(rx->d_quiet.p, 0, kQuietCounters * sizeof(uint32_t)) != hipSuccess) rc = OOKD_ERR_HIP;
    if (rx->have_fsm) {
        rc |= rx->d_seg_bounds.alloc(caps * (rx->max_segs_per_cap + 1));
        rc |= rx->d_state_in.alloc(nseg);
        rc |= rx->d_state_out.alloc(2 * nseg);
        rc |= rx->d_seg_msgs.alloc(nseg * rx->msg_slots);
        rc |= rx->d_msgs.alloc(rx->msg_capacity);
        rc |= rx->d_seg_msg_count.alloc(nseg);
        rc |= rx->d_seg_err_count.alloc(nseg);
        rc |= rx->d_seg_errs.alloc(nseg * rx->err_slots);
        if (getenv("OOKD_DEBUG")) rc |= rx->d_debug.alloc(nseg * 4 + 128);
        // scan form: abstract states = states x bit counts + skip x2 + poison
        rx->scan_S = (uint32_t)device->state_duration_us.size();
        rx->scan_max_bits = device->num_bits;
        rx->scan_D = rx->scan_S * (device->num_bits + 2) + 3;
        rx->scan_ok = !(cfg->flags & OOKD_RX_FSM_ROUNDS) && rx->scan_D <= 384 && device->num_bits <= 254 && !rx->big_device;
        std::vector<uint16_t> stuck_src;        // normal codes an inert edge can leave stuck (domain extension)
        std::vector<uint8_t> stuck_rows;
        if (rx->scan_ok && !(cfg->flags & OOKD_RX_SCAN_SIMS)) {
            // span tables: packed result of a span as a step function of its length
            std::vector<uint32_t> off, n0, pk;
            std::vector<uint16_t> reach;
            const auto t0 = std::chrono::steady_clock::now();
            const bool ok = build_leaf_tables(*rx->h_tables, cfg->samples_per_buffer, rx->total_decim, off, n0, pk,
                                              reach, stuck_src, stuck_rows);
            if (!ok) {
                stuck_src.clear();
                stuck_rows.clear();
            }
            if (getenv("OOKD_DEBUG")) {
                size_t zeros = 0;
                for (uint32_t v : pk) zeros += v == 0;
                fprintf(stderr, "[ookd] span tables: %s, %zu intervals (%zu need simulation), %zu codes can get stuck, "
                                "%zu codes reachable, %.1f ms\n",
                        ok ? "built" : "REFUSED", n0.size(), zeros, stuck_src.size(), reach.size(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (getenv("OOKD_DEBUG")[0] == '2') {
                    for (size_t t = 0; t + 1 < off.size(); ++t) {
                        fprintf(stderr, "[ookd]  row %zu L %zu:", t / 2, t % 2);
                        for (uint32_t i = off[t]; i < off[t + 1]; ++i) fprintf(stderr, " %u:%08x", n0[i], pk[i]);
                        fprintf(stderr, "\n");
                    }
                }
            }
            if (ok && !reach.empty()) {
                const uint32_t d0 = rx->scan_S * (device->num_bits + 2) + 3;
                rx->scan_reach_base = 0;                // entries are code | level mask << 14, ascending in the code
                for (uint16_t v : reach) rx->scan_reach_base += (v & 0x3fffu) < d0 ? 1u : 0u;
                // ... then, for the composition of chunk tables, the codes met at level 0 and those met at
                // level 1 as two plain lists (a chunk starts at one level: only that list is walked)
                rx->scan_reach_n = (uint32_t)reach.size();
                {
                    std::vector<uint16_t> l0, l1;
                    for (uint32_t i = 0; i < rx->scan_reach_base; ++i) {
                        if (reach[i] & 0x4000u) l0.push_back((uint16_t)(reach[i] & 0x3fffu));
                        if (reach[i] & 0x8000u) l1.push_back((uint16_t)(reach[i] & 0x3fffu));
                    }
                    rx->scan_reach_lv[0] = (uint32_t)l0.size();
                    rx->scan_reach_lv[1] = (uint32_t)l1.size();
                    reach.insert(reach.end(), l0.begin(), l0.end());
                    reach.insert(reach.end(), l1.begin(), l1.end());
                }
                rc |= rx->d_reach.alloc(reach.size());
                if (rc == OOKD_OK && hipMemcpy(rx->d_reach.p, reach.data(), reach.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                    rc = OOKD_ERR_HIP;
                }
            }
            if (ok && !n0.empty()) {
                rc |= rx->d_lt_off.alloc(off.size());
                rc |= rx->d_lt_n0.alloc(n0.size());
                rc |= rx->d_lt_pk.alloc(pk.size());
                if (rc == OOKD_OK &&
                    (hipMemcpy(rx->d_lt_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                     hipMemcpy(rx->d_lt_n0.p, n0.data(), n0.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                     hipMemcpy(rx->d_lt_pk.p, pk.data(), pk.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
                    rc = OOKD_ERR_HIP;
                }
                std::vector<uint32_t> merged = build_merged_rows(rx->scan_S, off, n0, pk);
                rx->lt_merged_rows = (uint32_t)merged.size();
                {
                    // (reach: the level lists were appended behind the scan_reach_n masked codes above)
                    const std::vector<uint16_t> masked(reach.begin(), reach.begin() + (reach.empty() ? 0 : rx->scan_reach_n));
                    append_sync_codes(merged, rx->scan_S, device->num_bits + 2, device->num_bits, masked);
                    rx->scan_sync = !(cfg->flags & OOKD_RX_SCAN_TABLES) && !dev_getenv("OOKD_SCAN_NO_SYNC");
                    if (const char *e = dev_getenv("OOKD_SYNC_MIN_EDGES")) rx->sync_min_edges = strtoull(e, nullptr, 0);
                }
                rc |= rx->d_lt_merged.alloc(merged.size());
                if (rc == OOKD_OK && hipMemcpy(rx->d_lt_merged.p, merged.data(), merged.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                    rc = OOKD_ERR_HIP;
                }
            }
        }
        if (rx->scan_ok) {
            std::vector<uint4> image((fsm_scan_ltab_bytes() + 15) / 16);
            rx->scan_D = fsm_scan_fill_ltab(image.data(), *rx->h_tables, cfg->samples_per_buffer, rx->total_decim,
                                            stuck_src, stuck_rows);
            rc |= rx->d_ltab.alloc(image.size());
            if (rc == OOKD_OK &&
                hipMemcpy(rx->d_ltab.p, image.data(), image.size() * 16, hipMemcpyHostToDevice) != hipSuccess) {
                rc = OOKD_ERR_HIP;
            }
            rx->scan_leaf_block = fsm_scan_leaf_block(rx->scan_D, rx->scan_S, rx->scan_S * (rx->scan_max_bits + 2));
            rx->scan_blocks_cap = (uint32_t)(rx->edge_capacity / rx->scan_leaf_block + caps + 8);
            rc |= rx->d_block_tab.alloc((size_t)rx->scan_blocks_cap * ((rx->scan_D + 7u) & ~7u) + 64);
            {
                const size_t ngroups = rx->scan_blocks_cap / 16 + caps + 8;
                rc |= rx->d_cap_group_off.alloc(caps + 1);
                rc |= rx->d_group_tab.alloc(ngroups * ((rx->scan_D + 7u) & ~7u) + 64);
                const size_t nsuper = rx->scan_blocks_cap / 64 + caps + 8;
                rc |= rx->d_cap_super_off.alloc(caps + 1);
                rc |= rx->d_super_tab.alloc(nsuper * ((rx->scan_D + 7u) & ~7u) + 64);
                rc |= rx->d_super_in.alloc(nsuper);
                rc |= rx->d_cap_end.alloc(2 * (caps + 8));         // + cap_first
            }
            rc |= rx->d_cap_block_off.alloc(caps + 1);
            rc |= rx->d_events.alloc(rx->edge_capacity + caps + 8);
            rc |= rx->d_ev_hot.alloc(rx->edge_capacity + caps + 8);
            rc |= rx->d_app_vals.alloc(2 * (rx->edge_capacity + caps) + 512 * caps + 1024);
            rc |= rx->d_scan_errs.alloc(1u << 16);
            rc |= rx->d_cap_fallback.alloc(caps);
            // (the walk from synchronising spans keeps a plane of entry codes per candidate)
            rx->pre_plane = rx->scan_sync ? (size_t)rx->scan_blocks_cap * rx->scan_leaf_block + 64 : 0;
            rc |= rx->d_pre.alloc(((size_t)rx->scan_blocks_cap * rx->scan_leaf_block + 64) * (rx->scan_sync ? 4 : 1));
            rc |= rx->d_rowz.alloc((size_t)rx->scan_blocks_cap * rx->scan_leaf_block + 64);
            rc |= rx->d_skipc.alloc((size_t)rx->scan_blocks_cap * rx->scan_leaf_block + 64);
            rc |= rx->d_sync_rec.alloc(((size_t)rx->scan_blocks_cap + 8) * 10);     // records, digests, selections
            rc |= rx->d_blk_in.alloc((size_t)rx->scan_blocks_cap + 16);
            rc |= rx->d_final_state.alloc(caps);
            rx->scan_fin_cap = (uint32_t)((rx->edge_capacity + caps) / fsm_scan_fin_block() + caps + 8);
            rc |= rx->d_fin_off.alloc(caps + 1);
            rc |= rx->d_fsum.alloc(4 * (size_t)rx->scan_fin_cap);
            // stamped work counters (fsm_scan.hip: take_stamped_ticket): zeroed ONCE -- stamp 0 is no run's
            rc |= rx->d_fin_tickets.alloc(kMaxChunks + 1);
            if (rc == OOKD_OK && hipMemset(rx->d_fin_tickets.p, 0, (kMaxChunks + 1) * sizeof(unsigned long long)) != hipSuccess) rc = OOKD_ERR_HIP;
            // stamped aggregates: the stamp half of every word must start out as "no run"
            if (rc == OOKD_OK && hipMemset(rx->d_fsum.p, 0, rx->d_fsum.n * sizeof(uint64_t)) != hipSuccess) rc = OOKD_ERR_HIP;
        }
    }
    if (rc != OOKD_OK) return nullptr;
    {
        // sparse front-end output (1-stage kernels with the quiet shortcut, no float dump): the bit
        // words and tile infos start out zero and every run zeroes what the run before wrote
        FrontParams probe = rx->front_params(nullptr, 0);
        probe.n_out = rx->max_n_out;
        rx->sparse = front_sparse_capable(probe) && !dev_getenv("OOKD_DENSE_BITS");
        if (rx->sparse &&
            (hipMemset(rx->d_bits.p, 0, rx->d_bits.n * sizeof(uint64_t)) != hipSuccess ||
             hipMemset(rx->d_tile_info.p, 0, rx->d_tile_info.n * sizeof(uint32_t)) != hipSuccess)) {
            set_error("hipMemset of the bit words failed");
            return nullptr;
        }
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&rx->h_hdr), sizeof(ResultHeader)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&rx->h_msgs), rx->msg_capacity * sizeof(MsgDev)) != hipSuccess) {
        set_error("hipHostMalloc failed");
        return nullptr;
    }
    memset(rx->h_hdr, 0, sizeof(ResultHeader));
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&rx->h_hdr_dev), rx->h_hdr, 0) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void **>(&rx->h_msgs_dev), rx->h_msgs, 0) != hipSuccess) {
        set_error("pinned result buffers are not mapped into the device");
        return nullptr;
    }
    // ---- chunk pipeline: two internal streams, each on its own half of the CUs -----------------------
    {
        FrontParams probe = rx->front_params(nullptr, 0);
        probe.n_out = rx->max_n_out;
        const bool tuned = front_tile_bits(probe) != 0;
        // off unless asked for (pipeline_chunk_samples, or OOKD_PIPELINE=1 for the default chunk): with the
        // hardware-dispatched front end the chain's kernels are hardly dispatched while a front-end grid
        // has workgroups pending, and the chunked run is slower than the whole one
        uint64_t want = cfg->pipeline_chunk_samples;
        if (!want && dev_getenv("OOKD_PIPELINE")) want = kPipeDefaultChunk;
        rx->pipe_chunk_in = want ? want : kPipeDefaultChunk;
        rx->pipe_ok = want != 0 && rx->have_fsm && rx->scan_ok && tuned && !(cfg->flags & OOKD_RX_NO_PIPELINE) &&
                      cfg->pipeline_chunk_samples != ~0ull && !dev_getenv("OOKD_NO_PIPELINE") &&
                      rx->max_n_out >= 2 * (rx->pipe_chunk_in / rx->total_decim);
    }
    if (rx->pipe_ok) {
        // every other CU of every XCD for the front end, the rest for the chain (an UNEVEN mask
        // slows the front end: the dispatcher deals workgroups evenly over the XCDs).
        // OOKD_PIPE_MASKS=front,chain (hex, per 32 CUs; 0 = no mask) overrides.
        uint32_t mf = 0x55555555u, mc = 0xAAAAAAAAu;
        if (const char *m = dev_getenv("OOKD_PIPE_MASKS")) {
            char *end = nullptr;
            mf = (uint32_t)strtoul(m, &end, 16);
            mc = (end && *end == ',') ? (uint32_t)strtoul(end + 1, nullptr, 16) : 0u;
        }
        auto make = [&](hipStream_t &s, uint32_t pattern) {
            if (pattern == 0) return hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            uint32_t mask[8];
            for (auto &w : mask) w = pattern;
            hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
            if (e != hipSuccess) {          // (no CU masking on this system: plain streams still pipeline)
                (void)hipGetLastError();
                e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            }
            return e;
        };
        bool made = make(rx->s_front, mf) == hipSuccess;
        if (const char *k = dev_getenv("OOKD_PIPE_DUMMY")) {       // experiment: shift the queue -> pipe assignment
            for (int i = 0; i < atoi(k); ++i) {
                hipStream_t d = nullptr;
                (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking);
                // a queue only exists once something was submitted to it
                (void)hipMemsetAsync(rx->d_hdr.p, 0, 4, d);
                (void)hipStreamSynchronize(d);
                rx->dummy_streams.push_back(d);         // destroyed with the context
            }
        }
        if (!made || make(rx->s_chain, mc) != hipSuccess ||
            hipEventCreateWithFlags(&rx->ev_start, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&rx->ev_end, hipEventDisableTiming) != hipSuccess) {
            set_error("creating the pipeline streams failed");
            return nullptr;
        }
        if (rx->d_carry.alloc(2) != OOKD_OK || rx->d_chunk_totals.alloc(4) != OOKD_OK) {
            return nullptr;
        }
    }
    rx->gate = static_cast<ookd_rx_gate *>(cfg->front_gate);
    // (tests: start the tile stamp near its wrap-around)
    if (const char *e = dev_getenv("OOKD_TILE_STAMP_START")) rx->tile_stamp = std::min<uint32_t>((uint32_t)strtoul(e, nullptr, 0), kTileStampMax);
    return rx.release();
}

void ookd_rx_destroy(ookd_rx *rx) { delete rx; }

ookd_rx_gate *ookd_rx_gate_create(void) { return new (std::nothrow) ookd_rx_gate(); }

void ookd_rx_gate_destroy(ookd_rx_gate *gate) { delete gate; }

int ookd_rx_submit_device(ookd_rx *rx, const void *d_iq, uint32_t num_captures,
                          uint64_t samples_per_capture, uint64_t capture_stride_samples) {
    clear_error();
    if (!rx || (!d_iq && samples_per_capture)) {
        set_error("ookd_rx_submit_device: null argument");
        return OOKD_ERR_ARG;
    }
    if (rx->submitted) {
        set_error("ookd_rx_submit_device: the previous run has not been waited for");
        return OOKD_ERR_ARG;
    }
    if (num_captures == 0 || num_captures > rx->max_captures || samples_per_capture > rx->max_samples) {
        set_error("run of %u captures x %llu samples exceeds the context capacity (%u x %llu)",
                  num_captures, (unsigned long long)samples_per_capture, rx->max_captures,
                  (unsigned long long)rx->max_samples);
        return OOKD_ERR_ARG;
    }
    if (num_captures > 1 && capture_stride_samples < samples_per_capture) {
        set_error("capture stride smaller than the capture");
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    rx->run_caps = num_captures;
    rx->run_n_valid = samples_per_capture;
    rx->geometry(samples_per_capture, true, rx->run_n_in, rx->run_n_out, rx->run_words,
                 rx->run_blocks, rx->run_segs_per_cap);
    rx->stats = ookd_rx_stats{};
    rx->last_iq = d_iq;
    rx->last_stride = capture_stride_samples;
    if (rx->plan_chunks()) {
        const int rcp = rx->run_pipelined(d_iq);
        if (rcp != OOKD_OK) return rcp;
        rx->submitted = true;
        return OOKD_OK;
    }
    int rc = rx->front_and_edges(d_iq, capture_stride_samples, nullptr, 0);
    if (rc != OOKD_OK) return rc;
    rc = rx->run_state_machine(nullptr, true);
    if (rc != OOKD_OK) return rc;
    rc = rx->enqueue_publish();
    if (rc != OOKD_OK) return rc;
    rx->submitted = true;
    return OOKD_OK;
}

int ookd_rx_wait(ookd_rx *rx) {
    clear_error();
    if (!rx || !rx->submitted) {
        set_error("ookd_rx_wait: nothing was submitted");
        return OOKD_ERR_ARG;
    }
    rx->submitted = false;
    HIPCHK(hipSetDevice(rx->dev));
    return rx->collect_results();
}

int ookd_rx_process_device(ookd_rx *rx, const void *d_iq, uint32_t num_captures,
                           uint64_t samples_per_capture, uint64_t capture_stride_samples) {
    static const bool timing = getenv("OOKD_DEBUG_TIME") != nullptr;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    int rc = ookd_rx_submit_device(rx, d_iq, num_captures, samples_per_capture, capture_stride_samples);
    if (rc != OOKD_OK) return rc;
    const auto t1 = clk::now();
    rc = ookd_rx_wait(rx);
    if (timing) {
        const auto t2 = clk::now();
        auto us = [](clk::time_point a, clk::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        fprintf(stderr, "[ookd] host us: enqueue %.1f, wait + collect %.1f\n", us(t0, t1), us(t1, t2));
    }
    return rc;
}

int ookd_rx_process_host(ookd_rx *rx, const int16_t *iq, uint64_t num_samples) {
    clear_error();
    if (!rx || (!iq && num_samples)) {
        set_error("ookd_rx_process_host: null argument");
        return OOKD_ERR_ARG;
    }
    if (num_samples > rx->max_samples) {
        set_error("capture of %llu samples exceeds max_samples %llu", (unsigned long long)num_samples,
                  (unsigned long long)rx->max_samples);
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    if (!rx->d_stage_in.p) {
        int rc = rx->d_stage_in.alloc(2 * rx->max_samples + 8);
        if (rc != OOKD_OK) return rc;
    }
    if (num_samples && rx->ingest.from_host(iq, rx->d_stage_in.p, num_samples * 4) < 0) return OOKD_ERR_HIP;
    return ookd_rx_process_device(rx, rx->d_stage_in.p, 1, num_samples, num_samples);
}

uint64_t ookd_rx_halo_samples(const ookd_rx *rx) { return rx ? rx->halo_needed : 0; }

int ookd_scan_domain_info(const ookd_device *device, uint32_t samples_per_buffer, uint32_t total_decimation,
                          uint32_t out[8]) {
    clear_error();
    if (!device || !out || !samples_per_buffer || !total_decimation) {
        set_error("ookd_scan_domain_info: bad argument");
        return OOKD_ERR_ARG;
    }
    std::unique_ptr<FsmTablesDev> t(new FsmTablesDev());
    memset(t.get(), 0, sizeof(FsmTablesDev));
    fill_fsm_tables(*device, *t);
    std::vector<uint32_t> off, n0, pk;
    std::vector<uint16_t> reach, stuck_src;
    std::vector<uint8_t> stuck_rows;
    const bool ok = build_leaf_tables(*t, samples_per_buffer, total_decimation, off, n0, pk, reach, stuck_src,
                                      stuck_rows);
    size_t zeros = 0;
    for (uint32_t v : pk) zeros += v == 0;
    const uint32_t S = (uint32_t)device->state_duration_us.size();
    // the merged rows (one search per leaf for scan_entry_kernel) must say what the per-row searches say, at
    // every breakpoint, next to it and far beyond: a mismatch reports the tables as not built
    bool merged_ok = true;
    if (ok && !n0.empty()) {
        const std::vector<uint32_t> m = build_merged_rows(S, off, n0, pk);
        auto lookup = [&](uint32_t row, uint32_t L, uint32_t n) -> uint32_t {
            uint32_t lo = off[2 * row + L], hi = off[2 * row + L + 1];
            if (lo >= hi) return 0u;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (n0[mid] <= n) lo = mid;
                else hi = mid;
            }
            return pk[lo];
        };
        std::vector<uint32_t> probes = {0u, 1u, 0xfffffff0u, 0x7fffffffu};
        for (uint32_t v : n0) {
            probes.push_back(v);
            probes.push_back(v + 1);
            if (v) probes.push_back(v - 1);
        }
        for (uint32_t L = 0; L < 2 && merged_ok; ++L) {
            const uint32_t nbp = m[L];
            const uint32_t *bp = m.data() + 4 + (L ? m[0] : 0u);
            for (uint32_t n : probes) {
                uint32_t lo = 0, hi = nbp;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (bp[mid] <= n) lo = mid;
                    else hi = mid;
                }
                const uint32_t *rows = m.data() + 4 + m[0] + m[1] + ((L ? m[0] : 0u) + lo) * 2u * S;
                for (uint32_t k = 0; k < S; ++k) {
                    const uint32_t p0 = lookup(2 * k, L, n);
                    const uint32_t p1 = (p0 & 0x10000000u) ? p0 : lookup(2 * k + 1, L, n);      // kPkShared
                    if (rows[2 * k] != (p0 & ~0x10000000u) || rows[2 * k + 1] != (p1 & ~0x10000000u)) merged_ok = false;
                }
            }
        }
    }
    // ... and the sync walk's second copy of the rows (append_sync_codes) must step every normal code exactly as the
    // rows do, and every interval's image must hold what its reachable codes end in
    if (ok && merged_ok && !n0.empty()) {
        std::vector<uint32_t> m = build_merged_rows(S, off, n0, pk);
        const uint32_t NB1 = device->num_bits + 2, max_bits = device->num_bits, SNB = S * NB1;
        append_sync_codes(m, S, NB1, max_bits, reach);
        const uint32_t nbp[2] = {m[0], m[1]}, rows0 = 4 + m[0] + m[1], sync_off = m[3];
        const uint32_t rows2 = sync_off + 2 * (nbp[0] + nbp[1]);
        if (m.size() != rows2 + (size_t)(nbp[0] + nbp[1]) * 2 * S) merged_ok = false;
        for (uint32_t L = 0; L < 2 && merged_ok; ++L) {
            for (uint32_t z = 0; z < nbp[L] && merged_ok; ++z) {
                const size_t at = (size_t)((L ? nbp[0] : 0u) + z) * 2 * S;
                const uint32_t w0 = m[sync_off + 2 * ((L ? nbp[0] : 0u) + z)], w1 = m[sync_off + 2 * ((L ? nbp[0] : 0u) + z) + 1];
                const uint32_t nimg = (w1 >> 16) & 0xfu, img[3] = {w0 & 0xffffu, w0 >> 16, w1 & 0xffffu};
                for (uint32_t c = 0; c < SNB && merged_ok; ++c) {
                    const uint32_t cur = c / NB1, nb = c - cur * NB1, r = 2 * cur + (nb >= max_bits ? 1u : 0u);
                    const uint32_t pp = m[rows0 + at + r], q = m[rows2 + at + r];
                    uint32_t want = 0xffffffffu;
                    if (pp & 0x80000000u) {
                        want = pp & 0xffffu;
                    } else if (pp & 0x20000000u) {
                        const uint32_t nbo = nb + ((pp >> 8) & 0xffffu);
                        want = (pp & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                    }
                    if (want == 0xffffffffu || want >= SNB + 3) {
                        if (q != 0) merged_ok = false;
                        continue;
                    }
                    if (!(q & 0x80000000u)) {
                        merged_ok = false;
                        continue;
                    }
                    const uint32_t nb2 = std::min(((q & 0x40000000u) ? nb : 0u) + ((q >> 8) & 0xffffu), NB1 - 1u), cur2 = q & 0xffu;
                    const uint32_t got = cur2 < S ? cur2 * NB1 + nb2 : SNB + nb2;
                    if (got != want) merged_ok = false;
                    // a reachable code's result is in the image (when the interval has one)
                    bool reachable = reach.empty();
                    for (uint16_t v : reach) reachable = reachable || ((v & 0x3fffu) == c && (v & (0x4000u << L)));
                    if (nimg && reachable && want != img[0] && want != img[1] && want != img[2]) merged_ok = false;
                }
            }
        }
    }
    out[0] = (ok && merged_ok) ? 1u : 0u;
    out[1] = (uint32_t)n0.size();
    out[2] = (uint32_t)zeros;
    out[3] = (uint32_t)reach.size();
    out[4] = ok ? (uint32_t)stuck_src.size() : 0u;
    out[5] = ok ? (uint32_t)stuck_rows.size() : 0u;
    std::vector<uint4> image((fsm_scan_ltab_bytes() + 15) / 16);
    if (!ok) {
        stuck_src.clear();
        stuck_rows.clear();
    }
    out[6] = fsm_scan_fill_ltab(image.data(), *t, samples_per_buffer, total_decimation, stuck_src, stuck_rows);
    out[7] = S;
    return OOKD_OK;
}

int ookd_rx_shard_begin(ookd_rx *rx, const void *d_iq, uint64_t num_samples, const int16_t *halo,
                        uint64_t halo_samples, int last_shard, const ookd_fsm_state *state_in,
                        ookd_fsm_state *state_out) {
    clear_error();
    if (!rx || (!d_iq && num_samples) || num_samples > rx->max_samples) {
        set_error("ookd_rx_shard_begin: bad argument");
        return OOKD_ERR_ARG;
    }
    const uint64_t spb = rx->cfg.samples_per_buffer;
    const uint64_t align = spb / gcd64(spb, rx->total_decim) * rx->total_decim;
    if (!last_shard && (num_samples % align) != 0) {
        set_error("shard of %llu samples is not a multiple of lcm(samples_per_buffer, decimation) = %llu",
                  (unsigned long long)num_samples, (unsigned long long)align);
        return OOKD_ERR_ARG;
    }
    if (halo && halo_samples < rx->halo_needed) {
        set_error("halo of %llu samples is shorter than the %llu the filter needs",
                  (unsigned long long)halo_samples, (unsigned long long)rx->halo_needed);
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    uint32_t hl = 0;
    if (halo && rx->halo_needed) {
        hl = (uint32_t)rx->halo_needed;
        // hipMemcpyDefault: the halo may be a host array or a device buffer an
        // RCCL recv landed in
        HIPCHK(hipMemcpyAsync(rx->d_halo.p, halo + 2 * (halo_samples - hl), (size_t)hl * 4,
                              hipMemcpyDefault, rx->stream));
    }
    rx->run_caps = 1;
    rx->run_n_valid = num_samples;
    rx->geometry(num_samples, last_shard != 0, rx->run_n_in, rx->run_n_out, rx->run_words,
                 rx->run_blocks, rx->run_segs_per_cap);
    rx->chunks.clear();
    rx->stats = ookd_rx_stats{};
    int rc = rx->front_and_edges(d_iq, num_samples, hl ? rx->d_halo.p : nullptr, hl);
    if (rc != OOKD_OK) return rc;
    static_assert(sizeof(ookd_fsm_state) == sizeof(FsmStateDev), "fsm state layout");
    rc = rx->run_state_machine(reinterpret_cast<const FsmStateDev *>(state_in), true);
    if (rc != OOKD_OK) return rc;
    rc = rx->fetch_results();
    if (rc != OOKD_OK) return rc;
    if (state_out && rx->have_fsm && rx->run_n_out > 0) {
        const size_t nseg = rx->run_segs_per_cap;
        const SegState *src = rx->scan_used ? rx->d_final_state.p
                                            : &rx->d_state_out.p[(size_t)rx->final_parity * nseg + (nseg - 1)];
        HIPCHK(hipMemcpy(state_out, &src->st, sizeof(FsmStateDev), hipMemcpyDeviceToHost));
    } else if (state_out) {
        if (state_in) *state_out = *state_in;
        else memset(state_out, 0, sizeof(*state_out));
    }
    return OOKD_OK;
}

int ookd_rx_shard_refine(ookd_rx *rx, const ookd_fsm_state *state_in, ookd_fsm_state *state_out) {
    clear_error();
    if (!rx || !state_in) {
        set_error("ookd_rx_shard_refine: null argument");
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    if (!rx->have_fsm || rx->run_n_out == 0) {
        if (state_out) *state_out = *state_in;
        return OOKD_OK;
    }
    int rc;
    if (rx->scan_used) {
        // the scan is cheap enough to simply run again from the new incoming state
        HIPCHK(hipMemsetAsync(rx->d_hdr.p->totals, 0, sizeof(uint64_t) * 2, rx->stream));
        HIPCHK(hipMemsetAsync(&rx->d_hdr.p->scan_fallback, 0, 2 * sizeof(uint32_t), rx->stream));   // + sync_fail
        rc = rx->run_state_machine(reinterpret_cast<const FsmStateDev *>(state_in), true);
    } else {
        rc = rx->fsm_to_fixpoint(reinterpret_cast<const FsmStateDev *>(state_in), false, true);
    }
    if (rc != OOKD_OK) return rc;
    rc = rx->fetch_results();
    if (rc != OOKD_OK) return rc;
    if (state_out) {
        const size_t nseg = rx->run_segs_per_cap;
        const SegState *src = rx->scan_used ? rx->d_final_state.p
                                            : &rx->d_state_out.p[(size_t)rx->final_parity * nseg + (nseg - 1)];
        HIPCHK(hipMemcpy(state_out, &src->st, sizeof(FsmStateDev), hipMemcpyDeviceToHost));
    }
    return OOKD_OK;
}

uint64_t ookd_rx_num_messages(const ookd_rx *rx) { return rx ? rx->num_msgs : 0; }

const ookd_message *ookd_rx_messages(const ookd_rx *rx) {
    static_assert(sizeof(ookd_message) == sizeof(MsgDev), "message layout");
    return rx ? reinterpret_cast<const ookd_message *>(rx->h_msgs) : nullptr;
}

int ookd_rx_get_stats(const ookd_rx *rx, ookd_rx_stats *out) {
    if (!rx || !out) return OOKD_ERR_ARG;
    *out = rx->stats;
    return OOKD_OK;
}

uint64_t ookd_rx_bit_words(const ookd_rx *rx) { return rx ? rx->run_words : 0; }

int ookd_rx_get_bits(const ookd_rx *rx, uint32_t capture, uint64_t *words, uint64_t capacity_words) {
    clear_error();
    if (!rx || !words || capture >= rx->run_caps) return OOKD_ERR_ARG;
    const uint64_t n = std::min<uint64_t>(capacity_words, rx->run_words);
    HIPCHK(hipSetDevice(rx->dev));
    {
        const int rc = rx->densify_bits();      // sparse front-end output: quiet tiles hold older runs' words
        if (rc != OOKD_OK) return rc;
    }
    HIPCHK(hipMemcpy(words, rx->d_bits.p + (size_t)capture * rx->run_words, n * 8, hipMemcpyDeviceToHost));
    return OOKD_OK;
}

int ookd_rx_get_edges(const ookd_rx *rx, uint32_t capture, uint64_t *edges, uint64_t capacity,
                      uint64_t *num_edges) {
    clear_error();
    if (!rx || capture >= rx->run_caps) return OOKD_ERR_ARG;
    HIPCHK(hipSetDevice(rx->dev));
    if (rx->run_n_out == 0) {
        if (num_edges) *num_edges = 0;
        return OOKD_OK;
    }
    if (!rx->chunks.empty()) {
        // pipelined run: one list per chunk, positions chunk-local
        std::vector<uint64_t> all;
        for (size_t c = 0; c < rx->chunks.size(); ++c) {
            const Chunk &ch = rx->chunks[c];
            uint32_t ne = 0;
            HIPCHK(hipMemcpy(&ne, rx->d_blk_offset.p + ch.blk0 + c + ch.nblk, 4, hipMemcpyDeviceToHost));
            std::vector<uint64_t> loc(std::min<uint64_t>(ne, ch.edge_cap));
            if (!loc.empty()) {
                HIPCHK(hipMemcpy(loc.data(), rx->d_edges.p + ch.edge_off, loc.size() * 8, hipMemcpyDeviceToHost));
            }
            for (uint64_t e : loc) all.push_back(e + ch.out0);
        }
        if (num_edges) *num_edges = all.size();
        if (edges && !all.empty()) memcpy(edges, all.data(), std::min<uint64_t>(all.size(), capacity) * 8);
        return OOKD_OK;
    }
    uint32_t off[2];
    HIPCHK(hipMemcpy(&off[0], rx->d_blk_offset.p + (size_t)capture * rx->run_blocks, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&off[1], rx->d_blk_offset.p + (size_t)(capture + 1) * rx->run_blocks, 4,
                     hipMemcpyDeviceToHost));
    const uint64_t n = off[1] - off[0];
    if (num_edges) *num_edges = n;
    if (edges && n) {
        HIPCHK(hipMemcpy(edges, rx->d_edges.p + off[0], std::min<uint64_t>(n, capacity) * 8,
                         hipMemcpyDeviceToHost));
    }
    return OOKD_OK;
}

int ookd_rx_get_fir(const ookd_rx *rx, uint32_t capture, ookd_complexf *out, uint64_t capacity) {
    clear_error();
    if (!rx || !out || capture >= rx->run_caps) return OOKD_ERR_ARG;
    if (!rx->d_fir.p) {
        set_error("ookd_rx_get_fir needs OOKD_RX_KEEP_FIR");
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    const uint64_t n = std::min<uint64_t>(capacity, rx->run_n_out);
    HIPCHK(hipMemcpy(out, rx->d_fir.p + 2 * (size_t)capture * rx->run_n_out, n * 8, hipMemcpyDeviceToHost));
    return OOKD_OK;
}

int ookd_rx_get_errors(const ookd_rx *rx, uint64_t *samples, uint64_t capacity, uint64_t *num) {
    clear_error();
    if (!rx) return OOKD_ERR_ARG;
    if (num) *num = rx->stats.num_errors;
    if (!rx->have_fsm || rx->run_n_out == 0 || !samples || capacity == 0) return OOKD_OK;
    HIPCHK(hipSetDevice(rx->dev));
    if (rx->mixed_valid) {              // scan + rounds for the captures it refused: the list lives on the host
        const uint64_t n = std::min<uint64_t>(rx->mixed_errs.size(), capacity);
        if (n) memcpy(samples, rx->mixed_errs.data(), n * 8);
        return OOKD_OK;
    }
    if (rx->scan_used) {
        const uint64_t n = std::min<uint64_t>(std::min<uint64_t>(rx->stats.num_errors, capacity), rx->d_scan_errs.n);
        if (n) HIPCHK(hipMemcpy(samples, rx->d_scan_errs.p, n * 8, hipMemcpyDeviceToHost));
        return OOKD_OK;
    }
    const size_t nseg = (size_t)rx->run_caps * rx->run_segs_per_cap;
    std::vector<uint32_t> counts(nseg);
    std::vector<uint64_t> errs(nseg * rx->err_slots);
    HIPCHK(hipMemcpy(counts.data(), rx->d_seg_err_count.p, nseg * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(errs.data(), rx->d_seg_errs.p, errs.size() * 8, hipMemcpyDeviceToHost));
    uint64_t at = 0;
    for (size_t s = 0; s < nseg && at < capacity; ++s) {
        const uint32_t c = std::min<uint32_t>(counts[s], rx->err_slots);
        for (uint32_t i = 0; i < c && at < capacity; ++i) samples[at++] = errs[s * rx->err_slots + i];
    }
    return OOKD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// recorders
// ---------------------------------------------------------------------------

namespace {

// record_dig (ookiedokie.c:146-169) restated over the edge list: the first
// line is sample 0's level, every later level change at i writes the pair
// "i-1, old" / "i, new".  The edge list counts a capture that starts high as
// a change at 0 (level before the capture = 0); record_dig does not.
template <typename Sink>
void dig_lines(const uint64_t *edges, uint64_t n, Sink &&sink) {
    char line[96];
    bool level = n > 0 && edges[0] == 0;
    int m = snprintf(line, sizeof(line), "0, %c\n", level ? '1' : '0');
    sink(line, (size_t)m);
    for (uint64_t e = level ? 1 : 0; e < n; ++e) {
        const uint64_t i = edges[e];
        m = snprintf(line, sizeof(line), "%" PRIu64 ", %c\n%" PRIu64 ", %c\n", i - 1, level ? '1' : '0', i,
                     level ? '0' : '1');
        sink(line, (size_t)m);
        level = !level;
    }
}

int fetch_edges(const ookd_rx *rx, uint32_t capture, std::vector<uint64_t> &edges) {
    uint64_t n = 0;
    int rc = ookd_rx_get_edges(rx, capture, nullptr, 0, &n);
    if (rc != OOKD_OK) return rc;
    if (n > rx->edge_capacity) {
        set_error("edge list overflowed its capacity: raise ookd_rx_config.edge_capacity");
        return OOKD_ERR_CAPACITY;
    }
    edges.resize(n);
    if (n) rc = ookd_rx_get_edges(rx, capture, edges.data(), n, &n);
    return rc;
}

}  // namespace

extern "C" {

size_t ookd_rx_dig_text(const ookd_rx *rx, uint32_t capture, char *out, size_t capacity) {
    clear_error();
    if (out && capacity) out[0] = '\0';
    if (!rx || capture >= rx->run_caps) {
        set_error("ookd_rx_dig_text: bad argument");
        return 0;
    }
    if (rx->run_n_out == 0) return 0;           // no buffer was ever processed: record_dig never ran
    std::vector<uint64_t> edges;
    if (fetch_edges(rx, capture, edges) != OOKD_OK) return 0;
    size_t len = 0;
    dig_lines(edges.data(), edges.size(), [&](const char *s, size_t m) {
        if (out && len < capacity) memcpy(out + len, s, std::min(m, capacity - len));
        len += m;
    });
    if (out && capacity) out[std::min(len, capacity - 1)] = '\0';
    return len;
}

int ookd_rx_record_dig(const ookd_rx *rx, uint32_t capture, const char *path) {
    clear_error();
    if (!rx || !path || capture >= rx->run_caps) {
        set_error("ookd_rx_record_dig: bad argument");
        return OOKD_ERR_ARG;
    }
    std::vector<uint64_t> edges;
    if (rx->run_n_out) {
        int rc = fetch_edges(rx, capture, edges);
        if (rc != OOKD_OK) return rc;
    }
    FILE *f = fopen(path, "w");                 // ookiedokie.c:112
    if (!f) {
        set_error("Failed to open %s: %s", path, strerror(errno));
        return OOKD_ERR_IO;
    }
    bool ok = true;
    if (rx->run_n_out) {
        dig_lines(edges.data(), edges.size(), [&](const char *s, size_t m) { ok = ok && fwrite(s, 1, m, f) == m; });
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error("Failed to write %s", path);
        return OOKD_ERR_IO;
    }
    return OOKD_OK;
}

int ookd_rx_get_fir_sc16q11(const ookd_rx *rx, uint32_t capture, int16_t *out, uint64_t capacity) {
    clear_error();
    if (!rx || !out || capture >= rx->run_caps) return OOKD_ERR_ARG;
    if (!rx->d_fir.p) {
        set_error("ookd_rx_get_fir_sc16q11 needs OOKD_RX_KEEP_FIR");
        return OOKD_ERR_ARG;
    }
    HIPCHK(hipSetDevice(rx->dev));
    const uint64_t n = std::min<uint64_t>(capacity, rx->run_n_out);
    if (n == 0) return OOKD_OK;
    int16_t *d_tmp = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d_tmp), n * 4));
    hipError_t e = launch_pack(rx->d_fir.p + 2 * (size_t)capture * rx->run_n_out, d_tmp, n, rx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_tmp, n * 4, hipMemcpyDeviceToHost, rx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rx->stream);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) {
        set_error("ookd_rx_get_fir_sc16q11: %s", hipGetErrorString(e));
        return OOKD_ERR_HIP;
    }
    return OOKD_OK;
}

int ookd_rx_record_fir(const ookd_rx *rx, uint32_t capture, const char *path) {
    clear_error();
    if (!rx || !path || capture >= rx->run_caps) {
        set_error("ookd_rx_record_fir: bad argument");
        return OOKD_ERR_ARG;
    }
    if (!rx->d_fir.p) {
        set_error("ookd_rx_record_fir needs OOKD_RX_KEEP_FIR");
        return OOKD_ERR_ARG;
    }
    std::vector<int16_t> buf(2 * rx->run_n_out);
    if (rx->run_n_out) {
        int rc = ookd_rx_get_fir_sc16q11(rx, capture, buf.data(), rx->run_n_out);
        if (rc != OOKD_OK) return rc;
    }
    FILE *f = fopen(path, "wb");                // bladeRF_file.c:82 (tx direction)
    if (!f) {
        set_error("Failed to open %s: %s", path, strerror(errno));
        return OOKD_ERR_IO;
    }
    bool ok = fwrite(buf.data(), 4, rx->run_n_out, f) == rx->run_n_out;
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        set_error("Failed to write %s", path);
        return OOKD_ERR_IO;
    }
    return OOKD_OK;
}

}  // extern "C"
