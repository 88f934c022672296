// fsm_scan.hip -- the symbol state machine as a parallel scan over edges.
//
// The reference runs its state machine sample by sample
// (src/state_machine.c:421-556).  Between two level changes nothing but
// always / timeout / msg_complete triggers can fire, and at almost every
// level change some trigger fires and zeroes the elapsed-time counter, so
// right after edge i the machine is fully described by a SMALL abstract
// state
//        (current state, number of collected bits)      [counter k = 0]
// plus two "rest of this buffer is being skipped" states (device.c:646),
// one "assumption broken" state (poison), and -- for the edges on which NO
// trigger fires although the state is timing something (a glitch inside a bit
// gap) -- a few "stuck" twins STUCK_d(s): "was in (state, bits) = s d edges
// ago and nothing has fired since", whose next leaf is simply the span table
// row of s at the merged length (see kStuckDepth).  The effect of the samples between edge
// i-1 and edge i (inclusive) is then a function L_i on that finite set; the
// machine's trajectory is the prefix composition L_i o ... o L_1 applied to
// the state after the first edge -- a scan, computed blockwise:
//
//   leaf   : per block of edges, build the tables L_i (a handful of tiny
//            simulations per edge: every state x {few bits, all bits}), and
//            compose them into one table per block;
//   blocks : one workgroup per capture walks the block tables from the true
//            start state;
//   entry  : the state every leaf is entered in, from the block tables;
//   (sync  : or -- for edge lists long enough to pay for it -- without any
//            tables: a leaf that ends in one of at most four states whatever
//            it was entered in (the silence between two messages) splits the
//            capture; every stretch between two of them is walked once per
//            such state, and a scan over the few-entry maps picks the true
//            one: scan_sync / scan_syncwalk / scan_syncpick, further down;)
//   emit   : every edge re-simulates its own span from its now-known
//            incoming state and records what happened (appended bits,
//            resets, OUTPUT_READY, ERROR);
//   finish : prefix sums over those records rebuild the payloads (bit t of a
//            message is the t-th append since the last reset), the message
//            list, the error list and the outgoing state.
//
// Everything is exact or refuses: if the true path ever leaves the abstract
// model (more ignored edges in a row than the stuck twins hold, a span with
// too many events) the `fallback` word is set and the host reruns the
// capture with the segment/round path of edges_fsm.hip, which handles
// anything.  Both paths give identical results (tests run both against the
// oracle).
#include "kernels.hpp"
#include "common.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

namespace ookd {

namespace {

enum { kCondAlways = 1, kCondPulseStart, kCondPulseEnd, kCondTimeout, kCondMsgComplete };
enum { kActNone = 1, kActAppend0, kActAppend1, kActOutput };
enum { kResError = -1, kResNone = 0, kResOutput = 1 };

constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kSat = 0xfffffffeu;
constexpr uint32_t kMaxFires = 48;      // trigger firings simulated per span before giving up
constexpr uint32_t kMaxLeafApps = 30;
constexpr int kScanThreads = 1024;
// "Stuck" codes: an edge on which no trigger fires, in a state whose counter matters,
// leaves the machine outside (state, bit count).  But nothing fired, so the span simply
// goes on: STUCK_d(s) = "entered d leaves ago in normal state s, no trigger fired on the
// d edges since" -- the next leaf is row(s) evaluated at the merged length.
constexpr uint32_t kStuckDepth = 8;     // inert edges in a row that stay representable, at most (LTab.depth)
constexpr uint32_t kStuckDomain = 512;  // the stuck codes may grow the domain up to this size
constexpr uint32_t kMaxStuck = 256;     // normal codes that can get stuck (p3l-nexa2012: 37)
constexpr uint32_t kMaxStuckRows = 32;

// reasons in the fallback word
enum { kFbPoison = 1, kFbOverflow = 2, kFbBlocks = 4, kFbPool = 8, kFbSync = kScanFbSync };

// Tables in LDS, one 16-byte record per state / trigger: a simulation step is
// a chain of dependent table reads (~100 cycles each), so what belongs
// together is fetched with one ds_read_b128 and the triggers of a state in
// batches of independent reads.
struct LTab {
    uint4 st[kMaxStates];       // kmin, kmax, kto, row = tbeg | tend << 8 | k-free << 16 | has msg_complete << 17
    uint4 tr[kMaxTriggers];     // kmin, kmax, info = cond | action << 8 | next << 16, -
    uint32_t max_bits, S, NB1, D;       // NB1 = max_bits + 2 bit-count values; D = S*NB1 + 3 + stuck codes
    uint32_t spb, decim;
    uint32_t NS, nstuck_rows;           // stuck codes: S*NB1 + 3 + (d-1)*NS + index in stuck_src
    uint32_t depth;                     // d = 1 .. depth <= kStuckDepth
    uint32_t lvl0;                      // level in front of the capture's first sample: 0, or (chunk of a pipelined run,
                                        // set by the kernels) the last bit of the chunk before; leaf i runs at lvl0 ^ (i & 1)
    uint32_t buf_shift;                 // a buffer holds exactly 2^buf_shift decimated samples, or 0xffffffff: divide
    uint32_t pad_[3];
    uint16_t stuck_src[kMaxStuck];      // ascending: the normal codes that can get stuck
    uint8_t stuck_row[kMaxStuckRows];   // row | level << 7: span-table rows with a stuck result, met at that level
};

__host__ __device__ __forceinline__ uint32_t clamp32(uint64_t v) { return v == ~0ull ? kNone : (uint32_t)v; }

// The tables are laid out once on the host (fsm_scan_fill_ltab): the kernels
// fetch the image with 16-byte loads.
static_assert(sizeof(LTab) % 16 == 0, "LTab is copied in 16-byte pieces");
__device__ __forceinline__ void copy_ltab(LTab &T, const void *g) {
    const uint4 *src = static_cast<const uint4 *>(g);
    uint4 *dst = reinterpret_cast<uint4 *>(&T);
    for (uint32_t i = threadIdx.x; i < (uint32_t)(sizeof(LTab) / 16); i += blockDim.x) dst[i] = src[i];
}

// per-lane concrete machine + what happened in the span
struct PSim {
    uint32_t cur, nbits, k, prev;
};

struct Acc {
    uint32_t napp, appvals, apps_at_reset;
    uint32_t nout, nerr, fires;
    uint32_t out_ab0, out_ab1, out_rb0, out_rb1;
    uint64_t out_pos0, out_pos1, err_pos;
    bool reset_seen, sensitive, overflow, msgc_seen;
    bool last_fired;            // a trigger fired on the last sample that was run
};

__host__ __device__ __forceinline__ void acc_init(Acc &a) {
    a.napp = a.appvals = a.apps_at_reset = 0;
    a.nout = a.nerr = a.fires = 0;
    a.out_ab0 = a.out_ab1 = 0;
    a.out_rb0 = a.out_rb1 = 0xffu;
    a.out_pos0 = a.out_pos1 = 0;
    a.err_pos = 0;
    a.reset_seen = a.sensitive = a.overflow = a.msgc_seen = false;
    a.last_fired = false;
}

__host__ __device__ __forceinline__ uint32_t sat_add(uint32_t k, uint64_t m) {
    const uint64_t s = (uint64_t)k + m;
    return s > kSat ? kSat : (uint32_t)s;
}

__host__ __device__ __forceinline__ void canon(const LTab &T, PSim &f) {
    if (T.st[f.cur].w & 0x10000u) f.k = 0;
}

constexpr int kTrigBatch = 4;           // trigger records fetched together

// state_machine.c:421-519, one evaluation
__host__ __device__ __forceinline__ int p_eval(const LTab &T, PSim &f, Acc &a, uint32_t b, uint64_t pos) {
    const uint32_t s = f.cur;
    const uint4 srec = T.st[s];
    const uint32_t te = (srec.w >> 8) & 0xffu;
    const uint32_t max_bits = T.max_bits;
    bool fired = false;
    uint32_t info = 0;
    for (uint32_t t0 = srec.w & 0xffu; t0 < te && !fired; t0 += kTrigBatch) {
        uint4 rec[kTrigBatch];
#pragma unroll
        for (int j = 0; j < kTrigBatch; ++j) rec[j] = T.tr[(t0 + j) & (kMaxTriggers - 1)];
#pragma unroll
        for (int j = 0; j < kTrigBatch; ++j) {
            if (fired || t0 + j >= te) continue;
            if (f.k < rec[j].x || f.k > rec[j].y) continue;
            const uint32_t c = rec[j].z & 0xffu;
            bool m;
            if (c == kCondAlways) {
                m = true;
            } else if (c == kCondPulseStart) {
                m = !f.prev && b;
            } else if (c == kCondPulseEnd) {
                m = f.prev && !b;
            } else if (c == kCondTimeout) {
                m = f.k >= srec.z;
            } else {
                m = f.nbits >= max_bits;
                a.msgc_seen = true;
                // table rows are simulated for a whole class of bit counts; that is
                // only sound if this test is never reached after an append of the
                // same span (and before a reset)
                if (!a.reset_seen && a.napp > 0) a.sensitive = true;
            }
            if (m) {
                fired = true;
                info = rec[j].z;
            }
        }
    }
    if (!fired) {
        f.k = sat_add(f.k, 1);
        return kResNone;
    }
    a.fires++;
    const uint32_t fc = info & 0xffu, act = (info >> 8) & 0xffu, next = info >> 16;
    int result = kResNone;
    bool ok = true;
    if (fc == kCondPulseStart || fc == kCondPulseEnd) ok = f.k >= srec.x && f.k <= srec.y;
    if (ok) {
        if (act == kActAppend0 || act == kActAppend1) {
            if (a.napp < kMaxLeafApps) {
                if (act == kActAppend1) a.appvals |= 1u << a.napp;
            } else {
                a.overflow = true;
            }
            a.napp++;
            f.nbits = f.nbits >= kSat ? kSat : f.nbits + 1;
        } else if (act == kActOutput) {
            result = kResOutput;
            const uint32_t rb = a.reset_seen ? a.apps_at_reset : 0xffu;
            if (a.nout == 0) {
                a.out_pos0 = pos;
                a.out_ab0 = a.napp;
                a.out_rb0 = rb;
            } else if (a.nout == 1) {
                a.out_pos1 = pos;
                a.out_ab1 = a.napp;
                a.out_rb1 = rb;
            } else {
                a.overflow = true;
            }
            a.nout++;
        }
        f.cur = next;
    } else {
        result = kResError;
        f.cur = 0;
        if (a.nerr == 0) a.err_pos = pos;
        else a.overflow = true;                     // one error per span is all the record holds
        a.nerr++;
    }
    f.k = 0;
    return result;
}

// state_machine.c:521-539
__host__ __device__ __forceinline__ int p_step(const LTab &T, PSim &f, Acc &a, uint32_t b, uint64_t pos) {
    if (f.cur == 0) {
        f.nbits = 0;
        a.reset_seen = true;
        a.apps_at_reset = a.napp;
        const int r = p_eval(T, f, a, b, pos);
        if (r != kResNone) return r;
    }
    return p_eval(T, f, a, b, pos);
}

// evaluations until an always / timeout / msg_complete trigger fires while
// the level stays constant (kNone = never)
__host__ __device__ __forceinline__ uint32_t p_quiet(const LTab &T, const PSim &f, Acc &a) {
    const uint4 srec = T.st[f.cur];
    const uint32_t te = (srec.w >> 8) & 0xffu;
    const uint32_t kto = srec.z;
    const uint32_t max_bits = T.max_bits;
    uint32_t best = kNone;
    for (uint32_t t0 = srec.w & 0xffu; t0 < te; t0 += kTrigBatch) {
        uint4 rec[kTrigBatch];
#pragma unroll
        for (int j = 0; j < kTrigBatch; ++j) rec[j] = T.tr[(t0 + j) & (kMaxTriggers - 1)];
#pragma unroll
        for (int j = 0; j < kTrigBatch; ++j) {
            if (t0 + j >= te) continue;
            const uint32_t c = rec[j].z & 0xffu;
            uint32_t lo = rec[j].x;
            if (c == kCondTimeout) {
                if (kto == kNone) continue;
                lo = lo > kto ? lo : kto;
            } else if (c == kCondMsgComplete) {
                a.msgc_seen = true;
                if (!a.reset_seen && a.napp > 0) a.sensitive = true;
                if (f.nbits < max_bits) continue;
            } else if (c != kCondAlways) {
                continue;
            }
            const uint32_t first = f.k > lo ? f.k : lo;
            if (first > rec[j].y || first > kSat) continue;
            const uint32_t w = first - f.k;
            best = w < best ? w : best;
        }
    }
    return best;
}

__host__ __device__ __forceinline__ uint64_t next_buffer_start(const LTab &T, uint64_t pos) {
    // spb = decim << buf_shift: buffer b holds the decimated samples [b << shift, (b + 1) << shift)
    if (T.buf_shift != 0xffffffffu) return (pos | ((1ull << T.buf_shift) - 1ull)) + 1ull;
    const uint64_t in_idx = (uint64_t)T.decim * (pos + 1) - 1;
    const uint64_t buf = in_idx / T.spb;
    const uint64_t nb = ((buf + 1) * (uint64_t)T.spb) / T.decim;
    return nb > pos ? nb : pos + 1;
}

// Runs `n` samples of constant level L starting at absolute position pos0,
// then (has_edge) one sample of level !L.  Returns false when the span ends
// inside a skipped rest-of-buffer (f.prev = the level of the error sample).
__host__ __device__ __forceinline__ bool sim_span(const LTab &T, PSim &f, Acc &a, uint64_t pos0, uint32_t L, uint64_t n, bool has_edge) {
    uint64_t pos = pos0;
    const uint64_t end_const = pos0 + n;
    const uint64_t last = end_const + (has_edge ? 1 : 0);
    while (pos < last) {
        const uint32_t b = pos < end_const ? L : (L ^ 1u);
        if (b == f.prev && pos < end_const) {
            const uint64_t room = end_const - pos;
            const uint32_t q = p_quiet(T, f, a);
            uint64_t m;
            if (f.cur == 0) {
                m = q == kNone ? room : (uint64_t)(q >> 1);    // reset evaluates twice per sample
                if (m > room) m = room;
                f.k = sat_add(f.k, 2 * m);
            } else {
                m = q == kNone ? room : (uint64_t)q;
                if (m > room) m = room;
                f.k = sat_add(f.k, m);
            }
            if (m > 0) {
                canon(T, f);
                pos += m;
                a.last_fired = false;
                continue;
            }
        }
        if (a.fires > kMaxFires) {
            a.overflow = true;
            return true;
        }
        const uint32_t fires0 = a.fires;
        const int r = p_step(T, f, a, b, pos);
        a.last_fired = a.fires != fires0;
        f.prev = b;
        canon(T, f);
        if (r == kResError) {
            if (pos + 1 >= last) return false;      // error on the span's last sample: skip starts beyond it
            const uint64_t nb = next_buffer_start(T, pos);
            if (nb >= last) return false;           // still skipping when the span ends
            pos = nb;                               // resume inside this span: reset, k = 0
            continue;
        }
        pos += 1;
    }
    return true;
}

struct Span {                   // the samples a leaf covers
    uint64_t pos0, n;
    uint32_t L;
    bool has_edge;
    uint64_t prefix;            // entered stuck: samples before pos0 since the state was last normal
};

// leaf i (1 <= i < ne): samples e[i-1]+1 .. e[i]
__device__ __forceinline__ Span span_of(const LTab &T, const uint64_t *edges, uint64_t i) {
    Span s;
    s.pos0 = edges[i - 1] + 1;
    s.n = edges[i] - edges[i - 1] - 1;
    s.L = (uint32_t)(i & 1ull) ^ T.lvl0;        // level after edge i-1
    s.has_edge = true;
    s.prefix = 0;
    return s;
}

// abstract codes:  cur * NB1 + nb | skip(prev 0/1) | poison
__host__ __device__ __forceinline__ uint32_t code_skip(const LTab &T, uint32_t prev) { return T.S * T.NB1 + prev; }
__host__ __device__ __forceinline__ uint32_t code_poison(const LTab &T) { return T.S * T.NB1 + 2; }

// Concrete run of span sp from abstract state `code`; `resume` = first sample
// after the buffer of the edge that precedes the span (where a skip state
// starts feeding samples again).  Returns alive (false = ends inside a
// skip); f is the machine at the end.  Must not be called with the poison code.
// What a simulation hands back, squeezed into 12 dwords so that the
// out-of-line simulator returns it in registers.  (Passing the span and the
// records by reference put them in scratch memory: two round trips of
// several hundred cycles per call.)
struct SimRes {
    uint32_t w[12];
};

__host__ __device__ __forceinline__ SimRes sim_pack(const PSim &f, const Acc &a, bool alive) {
    SimRes r;
    r.w[0] = (f.cur & 0xffu) | ((f.prev & 1u) << 8) | ((alive ? 1u : 0u) << 9) | ((a.reset_seen ? 1u : 0u) << 10) |
             ((a.sensitive ? 1u : 0u) << 11) | ((a.overflow ? 1u : 0u) << 12) | ((a.msgc_seen ? 1u : 0u) << 13) |
             ((a.last_fired ? 1u : 0u) << 14);
    r.w[1] = f.k;
    r.w[2] = f.nbits;
    r.w[3] = (a.napp > 255u ? 255u : a.napp) | ((a.nout > 255u ? 255u : a.nout) << 8) |
             ((a.nerr > 255u ? 255u : a.nerr) << 16) | ((a.apps_at_reset & 0xffu) << 24);
    r.w[4] = a.appvals;
    r.w[5] = (a.out_ab0 & 0xffu) | ((a.out_ab1 & 0xffu) << 8) | ((a.out_rb0 & 0xffu) << 16) | ((a.out_rb1 & 0xffu) << 24);
    r.w[6] = (uint32_t)a.out_pos0;
    r.w[7] = (uint32_t)(a.out_pos0 >> 32);
    r.w[8] = (uint32_t)a.out_pos1;
    r.w[9] = (uint32_t)(a.out_pos1 >> 32);
    r.w[10] = (uint32_t)a.err_pos;
    r.w[11] = (uint32_t)(a.err_pos >> 32);
    return r;
}

// -> alive
__host__ __device__ __forceinline__ bool sim_unpack(const SimRes &r, PSim &f, Acc &a) {
    f.cur = r.w[0] & 0xffu;
    f.prev = (r.w[0] >> 8) & 1u;
    f.k = r.w[1];
    f.nbits = r.w[2];
    a.reset_seen = (r.w[0] >> 10) & 1u;
    a.sensitive = (r.w[0] >> 11) & 1u;
    a.overflow = (r.w[0] >> 12) & 1u;
    a.last_fired = (r.w[0] >> 14) & 1u;
    a.msgc_seen = (r.w[0] >> 13) & 1u;
    a.napp = r.w[3] & 0xffu;
    a.nout = (r.w[3] >> 8) & 0xffu;
    a.nerr = (r.w[3] >> 16) & 0xffu;
    a.apps_at_reset = r.w[3] >> 24;
    a.appvals = r.w[4];
    a.out_ab0 = r.w[5] & 0xffu;
    a.out_ab1 = (r.w[5] >> 8) & 0xffu;
    a.out_rb0 = (r.w[5] >> 16) & 0xffu;
    a.out_rb1 = r.w[5] >> 24;
    a.out_pos0 = r.w[6] | ((uint64_t)r.w[7] << 32);
    a.out_pos1 = r.w[8] | ((uint64_t)r.w[9] << 32);
    a.err_pos = r.w[10] | ((uint64_t)r.w[11] << 32);
    a.fires = 0;
    return (r.w[0] >> 9) & 1u;
}

// Concrete run of a span from abstract state `code`; `resume` = first sample
// after the buffer of the edge that precedes the span (where a skip state
// starts feeding samples again).  alive = false: ends inside a skip.  Must not
// be called with the poison code.  lvl_edge = level | has_edge << 1.
__host__ __device__ __noinline__ SimRes run_leaf_raw(const LTab &T, uint32_t code, uint64_t pos0, uint64_t n, uint32_t lvl_edge,
                                            uint64_t resume, uint64_t prefix = 0) {
    const uint32_t NB1 = T.NB1;
    const uint32_t nstates = T.S * NB1;
    const uint32_t L = lvl_edge & 1u;
    const bool has_edge = (lvl_edge & 2u) != 0;
    PSim f;
    Acc a;
    acc_init(a);
    bool alive;
    if (code >= nstates) {
        f.cur = 0;
        f.nbits = 0;
        f.k = 0;
        f.prev = code - nstates;
        const uint64_t end_const = pos0 + n;
        const uint64_t last = end_const + (has_edge ? 1 : 0);
        if (resume >= last) alive = false;
        else if (resume >= end_const) alive = sim_span(T, f, a, resume, L, 0, has_edge);
        else alive = sim_span(T, f, a, resume, L, end_const - resume, has_edge);
    } else {
        f.cur = code / NB1;
        f.nbits = code - f.cur * NB1;
        f.k = 0;
        f.prev = L;
        bool lost = false;
        if (prefix) {
            // entered stuck: no trigger fired on the edges inside the prefix, so the machine
            // went through it as through a constant level; what happened there belongs to
            // the leaves before
            lost = !sim_span(T, f, a, pos0 - prefix, L, prefix, false) || a.overflow;
            acc_init(a);
        }
        alive = sim_span(T, f, a, pos0, L, n, has_edge);
        if (lost) a.overflow = true;
    }
    // the 8-bit event counters of the record saturate; more than the record can
    // hold is an overflow anyway (kMaxLeafApps, two outputs, one error)
    return sim_pack(f, a, alive);
}

__device__ __forceinline__ bool run_leaf(const LTab &T, uint32_t code, const Span &sp, uint64_t resume, PSim &f,
                                         Acc &a) {
    const SimRes r = run_leaf_raw(T, code, sp.pos0, sp.n, sp.L | (sp.has_edge ? 2u : 0u), resume, sp.prefix);
    return sim_unpack(r, f, a);
}

__host__ __device__ __forceinline__ uint32_t encode_post(const LTab &T, const PSim &f, const Acc &a, bool alive) {
    if (a.overflow) return code_poison(T);
    if (!alive) return code_skip(T, f.prev);
    if (f.k != 0 && !(T.st[f.cur].w & 0x10000u)) return code_poison(T);   // counter not zeroed: not representable
    return f.cur * T.NB1 + (f.nbits >= T.NB1 ? T.NB1 - 1 : f.nbits);
}

}  // namespace

struct ScanParams {
    FsmParams f;
    uint16_t *block_tab;
    uint32_t *cap_block_off;
    LeafEvDev *events;
    uint8_t *app_vals;
    uint64_t app_capacity;
    uint64_t *errs;
    uint64_t err_capacity;
    FsmStateDev first;
    int have_first;
    // chunked (pipelined) runs: one launch of the scan per chunk of ONE capture, in stream order
    const SegState *first_dev;  // incoming state in device memory (the chunk before's final_state), or null
    uint32_t has_prev;          // the bit words continue in front of f.bits (chunk > 0): lvl0 = bits[-1] >> 63
    uint64_t pos_origin;        // the chunk's first decimated sample: added to message / error positions
    const uint64_t *totals_in;  // [2] messages / errors of the chunks before (device), or null = 0
    const uint32_t *edge_overflow;      // the edge stage's overflow flag: the scan refuses such a run
    // entry code of every leaf (scan_entry_kernel): the emit kernel then needs none of the tables
    uint16_t *pre_codes;        // [blocks][leaf_block] entry code of every leaf (scan_entry_kernel)
    uint16_t *blk_in;           // [blocks] entry code of every block
    uint16_t *rowz;             // [blocks][leaf_block] merged-rows interval of every leaf (0xffff: span too long for the tables)
    uint32_t *skipc;            // [blocks][leaf_block] the leaf applied to the two skip codes: out(skip 0) | out(skip 1) << 16
    uint32_t skipc_valid;       // the leaf kernel of this launch fills skipc (wave-per-block form)
    uint32_t entry_phase;       // scan_entry_kernel: 0 groups -> blocks, 1 blocks -> chunks -> leaves
    uint32_t *cap_fallback;     // [captures] refusal bits per capture (batched runs; zero at launch), or null: a
                                // capture whose path leaves the model is left out of the results -- the host
                                // redoes it alone in the round form -- instead of voiding the whole call
    SegState *final_state;
    uint32_t *fallback;
    uint32_t total_blocks_cap;
    uint32_t leaf_block;        // leaves per block (<= 256)
    uint32_t *fin_off;          // [captures + 1] prefix of finish-block counts
    unsigned long long *fagg;   // [finish blocks][4, two used] appends | outputs << 20, errors | (last epoch start + 1) << 12; each | stamp << 32
    unsigned long long *fin_ticket;     // next finish block to hand out: count | run stamp << 32 (take_stamped_ticket)
    uint32_t run_stamp;         // != 0, different from the previous launch's
    uint32_t fin_blocks_cap;
    // span tables (build_leaf_tables): packed result of a span as a function of its
    // length, per (row, level); null = simulate
    const uint32_t *lt_off, *lt_n0, *lt_pk;
    const void *ltab;           // device copy of the LTab (fsm_scan_fill_ltab)
    const uint16_t *reach;      // abstract codes a span can be entered in (from the span tables), or null = all
    uint32_t nreach, nreach_base;       // all of them / the normal, skip and poison codes among them (they come first)
    uint32_t nreach_lv[2];              // reach + nreach: the base codes met at level 0, then those met at level 1
    uint32_t lt_words;                  // size of the span tables (offsets + 2 x intervals)
    const uint32_t *lt_merged;          // build_merged_rows of the span tables, or null
    uint32_t lt_merged_words;
    uint32_t Dp, D;             // block table row pitch (D rounded up to 8); D = the domain with the stuck codes
    uint32_t S;                 // machine states
    uint32_t *cap_group_off;    // [captures + 1] prefix of group counts
    uint16_t *group_tab;        // [groups][Dp]
    uint32_t *cap_super_off;    // [captures + 1] prefix of supergroup counts (a supergroup = kSuper groups)
    uint16_t *super_tab;        // [supergroups][Dp]
    uint16_t *super_in;         // [supergroups] entry code (scan_walk_kernel)
    uint16_t *cap_end;          // [captures] state after the last regular leaf
    uint16_t *cap_first;        // [captures] state after the first span (leaf kernel -> walk kernel)
    // entry codes from synchronising spans (scan_sync_kernel / scan_syncwalk_kernel / scan_syncpick_kernel)
    uint32_t *ev_hot;           // [edges + captures] a leaf's record in one word (put_event); the 48-byte record only for
                                // the leaves that need it
    uint32_t *sync_rec;         // [blocks][kSyncRecWords]: see kSyncRec*
    uint32_t *sync_dig;         // [blocks] what scan_syncpick_kernel needs of a record, in one word (kSyncDig*)
    uint32_t *sync_sel;         // [blocks] split | plane of the leaves below it << 8 | plane of the others << 12
    uint32_t *sync_fail;        // device word, zero at launch: bit 1 = the walk from the synchronising spans was tried, bit 0 = it gave up
    uint32_t sync_try;          // this launch tries that walk first; the composing kernels only run when it gave up
    uint32_t sync_only;         // ... and are not even queued: giving up refuses the run (kFbSync), the host composes
    uint64_t pre_plane;         // elements between two planes of pre_codes (the walk keeps one per candidate)
    uint32_t lt_sync_words;     // lt_merged with append_sync_codes' tables behind the rows (lt_merged_words: without)
};

namespace {

__device__ __forceinline__ uint64_t cap_edges(const FsmParams &p, uint32_t cap, uint64_t &e0) {
    const uint32_t blk0 = cap * p.blocks_per_cap;
    e0 = p.blk_offset[blk0];
    return (uint64_t)p.blk_offset[blk0 + p.blocks_per_cap] - e0;
}


// position p of a block -> leaf index, even leaves first then odd ones
__device__ __forceinline__ uint32_t parity_order(uint32_t p, uint32_t count) {
    const uint32_t evens = (count + 1) >> 1;
    return p < evens ? 2 * p : (p < count ? 2 * (p - evens) + 1 : count);
}

// Builds the transition tables of leaves [first, first+count) into LDS:
// tab[l * D + d].
// Phase A of a block's tables: per leaf, the packed results of the class
// simulations, res[l][2S+2] (two per state: "few bits" nb = 0 standing for
// every nb < max_bits, "all bits" nb = max_bits standing for nb >= max_bits;
// then the two skip states).  Lanes of a wave share the start state and take
// leaves of equal level (even leaves first, then odd), so they follow nearly
// the same path.

// packed result of a class simulation
constexpr uint32_t kPkAbsolute = 0x80000000u;   // [15:0] end code, [23:16] its state (S for skip / poison)
constexpr uint32_t kPkSensitive = 0x40000000u;
constexpr uint32_t kPkRelative = 0x20000000u;   // [7:0] end state, [23:8] appended bits
constexpr uint32_t kPkShared = 0x10000000u;     // transient: class "all bits" takes this result too
constexpr uint32_t kPkStuck = 0x08000000u;      // no trigger fired on the edge, the counter runs on
constexpr uint32_t kPkEventShift = 24;          // relative results: bits 24..25 = the span's events in short (pack_normal)

__host__ __device__ __forceinline__ uint32_t pack_absolute(uint32_t code, uint32_t NB1) {
    return code | ((code / NB1) << 16) | kPkAbsolute;
}

// Packed result of a simulation from a normal state (state k, bit count nb0 =
// representative of class cls).
__host__ __device__ __forceinline__ uint32_t pack_normal(const LTab &T, const PSim &f, const Acc &a, bool alive,
                                                         uint32_t nb0, uint32_t cls) {
    const uint32_t NB1 = T.NB1;
    const uint32_t out = encode_post(T, f, a, alive);
    uint32_t packed;
    if (!a.overflow && alive && !a.sensitive && !a.last_fired && f.k != 0 && !(T.st[f.cur].w & 0x10000u)) {
        packed = kPkStuck;                                  // see kStuckDepth
    } else if (out >= T.S * NB1) {
        packed = pack_absolute(out, NB1);                   // skip / poison
    } else if (a.sensitive) {
        packed = kPkSensitive;                              // row needs one simulation per bit count
    } else if (a.reset_seen) {
        packed = pack_absolute(out, NB1);                   // bit count restarted inside the span
    } else {
        const uint32_t ocur = out / NB1;
        const uint32_t nbo = out - ocur * NB1;
        const uint32_t delta = nbo >= nb0 ? nbo - nb0 : 0u;
        packed = ocur | (delta << 8) | kPkRelative;         // relative: nb + delta (saturating)
        // what the span leaves behind for scan_emit_kernel, when that is all of it: nothing, or ONE appended bit
        // (no output, no error, no reset -- and, the row not being bit-count sensitive, the same for every bit count
        // of the class): bits 24..25 = 1 nothing, 2 / 3 a 0 / 1 appended; 0 = simulate
        if (a.nout == 0 && a.nerr == 0 && !a.overflow && a.napp <= 1u)
            packed |= (a.napp == 0 ? 1u : 2u + (a.appvals & 1u)) << kPkEventShift;
    }
    // no dependence on the bit count at all: class "all bits" takes the "few bits" result
    if (cls == 0 && !a.msgc_seen && !a.overflow) packed |= kPkShared;
    return packed;
}

// Span tables: table (row, L) -> entries [off[2*row+L], off[2*row+L+1]); entry i
// covers span lengths n0[i] .. n0[i+1]-1 (the last one to infinity).  Rows
// 0 .. 2S-1 = (state, class); rows 2S, 2S+1 = start in reset with previous level
// 0 / 1 DIFFERENT from the span's level (pk 0 = position dependent, simulate).
__device__ __forceinline__ uint32_t lt_lookup(const uint32_t *off, const uint32_t *n0, const uint32_t *pk, uint32_t row,
                                              uint32_t L, uint32_t n) {
    uint32_t lo = off[2 * row + L], hi = off[2 * row + L + 1];
    if (lo >= hi) return 0u;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (n0[mid] <= n) lo = mid;
        else hi = mid;
    }
    return pk[lo];
}

// ---- stuck codes (kStuckDepth) ---------------------------------------------------------
struct StuckCtx {
    const uint64_t *edges;                      // the capture's edge list
    const uint32_t *lt_off, *lt_n0, *lt_pk;     // span tables (null: stuck stays poison)
};

// normal code s got stuck on the edge of its leaf: STUCK_1(s), or poison if it is not in the domain
__device__ __forceinline__ uint32_t stuck_enter(const LTab &T, const StuckCtx &c, uint32_t s) {
    if (!c.lt_off) return code_poison(T);
    uint32_t lo = 0, hi = T.NS;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (T.stuck_src[mid] < s) lo = mid + 1;
        else hi = mid;
    }
    return (lo < T.NS && T.stuck_src[lo] == s) ? T.S * T.NB1 + 3 + lo : code_poison(T);
}

__device__ __forceinline__ void stuck_decode(const LTab &T, uint32_t code, uint32_t &d, uint32_t &src) {
    uint32_t rel = code - (T.S * T.NB1 + 3);
    d = 1;
    while (rel >= T.NS && d < kStuckDepth) {            // d <= kStuckDepth: cheaper than a division
        rel -= T.NS;
        ++d;
    }
    src = T.stuck_src[rel < T.NS ? rel : 0u];
}

// Leaf i (edge index in the capture) entered in a stuck code: no trigger fired on the d
// edges before it, so the machine saw one span from the start of leaf i - d on: the row
// of the code it had there, at the merged length, ended by this leaf's edge.
__device__ __noinline__ uint32_t stuck_step(const LTab &T, const StuckCtx &c, uint64_t i, uint32_t code) {
    const uint32_t NB1 = T.NB1;
    if (!c.lt_off || !T.NS) return code_poison(T);
    uint32_t d, src;
    stuck_decode(T, code, d, src);
    if (i < (uint64_t)d + 1) return code_poison(T);
    const uint64_t n = c.edges[i] - c.edges[i - d - 1] - 1;
    if (n > 0xfffffff0ull) return code_poison(T);
    const uint32_t cur = src / NB1, nb = src - cur * NB1;
    const uint32_t pk = lt_lookup(c.lt_off, c.lt_n0, c.lt_pk, 2 * cur + (nb >= T.max_bits ? 1u : 0u),
                                  (uint32_t)(i & 1ull) ^ T.lvl0, (uint32_t)n);
    if (pk & kPkAbsolute) return pk & 0xffffu;
    if (pk & kPkRelative) {
        const uint32_t nbo = nb + ((pk >> 8) & 0xffffu);
        return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
    }
    if ((pk & kPkStuck) && d < T.depth) return code + T.NS;
    return code_poison(T);              // deeper, position dependent or bit-count sensitive: not followed
}


constexpr uint32_t kCapWords = (256 + kStuckDepth + 31) / 32;
constexpr uint32_t kLtLdsWords = 768;   // span tables up to this size are searched from LDS (shipped devices: 409)
constexpr uint32_t kMergedLdsWords = 6144;      // merged rows (+ the sync walk's tables) up to this size are read from LDS
#define STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
// cap: bit kStuckDepth + l = leaf l of the block (l = -kStuckDepth .. count-1) can end stuck
__device__ void block_sims(const LTab &T, const uint64_t *edges, uint64_t first, uint32_t count, uint32_t *res,
                           uint64_t *resume, uint16_t *rep /* [count] */, uint16_t *uniq /* [count + 1] */,
                           const uint32_t *lt_off, const uint32_t *lt_n0, const uint32_t *lt_pk,
                           uint32_t *cap /* [kCapWords] */, uint64_t *dbg = nullptr) {
    STAMP(0);
    for (uint32_t w = threadIdx.x; w < kCapWords; w += blockDim.x) cap[w] = 0;
    const uint32_t S = T.S, NB1 = T.NB1, max_bits = T.max_bits;
    const uint32_t nsim = 2 * S + 2;
    // one 64-bit division per leaf instead of one per simulation
    for (uint32_t l = threadIdx.x; l < count; l += blockDim.x) resume[l] = next_buffer_start(T, edges[first + l - 1]);
    // From a normal state the outcome of a span depends only on its level and
    // length, not on where it lies: simulate each distinct (level, length) of the
    // block once.  rep[l] = first leaf of the block with the same key.
    STAMP(1);
    uint32_t *gap = res;                        // scratch: res is rewritten below
    for (uint32_t l = threadIdx.x; l < count; l += blockDim.x) {
        const uint64_t n = edges[first + l] - edges[first + l - 1];
        gap[l] = n > 0xfffffffeull ? 0xffffffffu - l : (uint32_t)n;     // giant gaps never match
    }
    __syncthreads();
    for (uint32_t l = threadIdx.x; l < count; l += blockDim.x) {
        const uint32_t n = gap[l];
        uint32_t r = l;
        for (uint32_t m = l & 1u; m < l; m += 2) {          // leaves of equal parity share the level
            if (gap[m] == n) {
                r = m;
                break;
            }
        }
        rep[l] = (uint16_t)r;
    }
    __syncthreads();
    STAMP(2);
    // list of representatives, even leaves first then odd ones (neighbouring lanes
    // then share the level); position = rank in that order, from per-wave ballots
    {
        __shared__ uint64_t s_mask[16][2];              // [wave][parity] representatives
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
        uint32_t total[2] = {0, 0};
        for (uint32_t l0 = 0; l0 < count; l0 += blockDim.x) {       // one pass unless count > blockDim
            const uint32_t l = l0 + threadIdx.x;
            const bool isrep = l < count && rep[l] == l;
            const uint64_t me = __ballot(isrep && !(l & 1u)), mo = __ballot(isrep && (l & 1u));
            if (lane == 0) {
                s_mask[wave][0] = me;
                s_mask[wave][1] = mo;
            }
            __syncthreads();
            uint32_t before[2] = {0, 0}, all[2] = {0, 0};
            for (uint32_t w = 0; w < nwaves; ++w) {
                const uint32_t ce = (uint32_t)__popcll(s_mask[w][0]), co = (uint32_t)__popcll(s_mask[w][1]);
                if (w < wave) {
                    before[0] += ce;
                    before[1] += co;
                }
                all[0] += ce;
                all[1] += co;
            }
            __syncthreads();
            // odd representatives come after ALL even ones of the block: they are placed in a
            // second sweep once the even total is known (single pass when count <= blockDim)
            if (isrep) {
                const uint64_t below = (1ull << lane) - 1ull;
                const uint32_t par = l & 1u;
                const uint32_t r = total[par] + before[par] + (uint32_t)__popcll((par ? mo : me) & below);
                // store parity-local rank now, add the even total below
                uniq[1 + l] = (uint16_t)r;              // scratch: slot l, compacted afterwards
            }
            total[0] += all[0];
            total[1] += all[1];
        }
        __syncthreads();
        // compact: rank among even reps, or (#even reps) + rank among odd reps
        uint16_t mine = 0xffffu;
        uint32_t mine_rank = 0;
        // (count <= 256 = at most one leaf per lane and pass; larger blocks loop)
        for (uint32_t l0 = 0; l0 < count; l0 += blockDim.x) {
            const uint32_t l = l0 + threadIdx.x;
            const bool isrep = l < count && rep[l] == l;
            if (isrep) {
                mine = (uint16_t)l;
                mine_rank = uniq[1 + l] + ((l & 1u) ? total[0] : 0u);
            }
            __syncthreads();
            if (isrep) uniq[1 + mine_rank] = mine;
            __syncthreads();
        }
        if (threadIdx.x == 0) uniq[0] = (uint16_t)(total[0] + total[1]);
        __syncthreads();
    }
    STAMP(3);
    const uint32_t nu = uniq[0];
    // Dense tasks, one per lane: (start state, distinct span, class) for the normal
    // rows, then the two skip states of every leaf (position dependent).  The
    // machine is busy with the instruction streams of these waves and nothing
    // else, so lanes must not idle: neighbouring lanes share the start state
    // (same triggers, mostly the same path), and both classes run side by side.
    if (lt_off) {
        // ---- table driven: the packed result is a step function of the span length ----------
        // normal rows: one lane per (distinct span, state); spans too long for the
        // tables' 32-bit lengths are simulated
        for (uint32_t task = threadIdx.x; task < nu * S; task += blockDim.x) {
            const uint32_t k = task / nu, up = task - k * nu;
            const uint32_t l = uniq[1 + up];
            const Span sp = span_of(T, edges, first + l);
            uint32_t p0, p1;
            if (sp.n <= 0xfffffff0ull) {
                p0 = lt_lookup(lt_off, lt_n0, lt_pk, 2 * k, sp.L, (uint32_t)sp.n);
                p1 = (p0 & kPkShared) ? p0 : lt_lookup(lt_off, lt_n0, lt_pk, 2 * k + 1, sp.L, (uint32_t)sp.n);
            } else {
                PSim f;
                Acc a;
                bool alive = run_leaf(T, k * NB1, sp, resume[l], f, a);
                p0 = pack_normal(T, f, a, alive, 0u, 0u);
                alive = run_leaf(T, k * NB1 + max_bits, sp, resume[l], f, a);
                p1 = pack_normal(T, f, a, alive, max_bits, 1u);
            }
            res[l * nsim + 2 * k] = p0;
            res[l * nsim + 2 * k + 1] = p1;
        }
        // skip rows: per leaf.  Skipping ends at `resume`; from there the machine starts
        // in reset -- the normal row (reset, few bits) of a shorter span when the level
        // before the skip equals the span's, a special row otherwise.
        for (uint32_t task = threadIdx.x; task < 2 * count; task += blockDim.x) {
            const uint32_t kk = task / count, lp = task - kk * count;
            const uint32_t l = parity_order(lp, count);
            const Span sp = span_of(T, edges, first + l);
            const uint64_t end_const = sp.pos0 + sp.n, last = end_const + 1;
            const uint64_t rs = resume[l];
            uint32_t out;
            if (rs >= last) {
                out = S * NB1 + kk;                              // still skipping when the span ends
            } else {
                const uint64_t n2 = rs >= end_const ? 0 : end_const - rs;
                uint32_t pk = 0;
                if (n2 <= 0xfffffff0ull) {
                    pk = lt_lookup(lt_off, lt_n0, lt_pk, kk == sp.L ? 0u : 2 * S + kk, sp.L, (uint32_t)n2);
                }
                if (pk & kPkAbsolute) {
                    out = pk & 0xffffu;
                } else if (pk & kPkRelative) {
                    const uint32_t nbo = (pk >> 8) & 0xffffu;       // from a bit count of 0
                    out = (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                } else {
                    PSim f;                                         // position dependent or sensitive
                    Acc a;
                    const bool alive = run_leaf(T, S * NB1 + kk, sp, rs, f, a);
                    out = encode_post(T, f, a, alive);
                }
            }
            res[l * nsim + 2 * S + kk] = pack_absolute(out, NB1);
        }
    } else {
    const uint32_t nnorm = 2 * nu * S;
    const uint32_t ntask = nnorm + 2 * count;
    for (uint32_t task = threadIdx.x; task < ntask; task += blockDim.x) {
        PSim f;
        Acc a;
        if (task >= nnorm) {
            const uint32_t t2 = task - nnorm;
            const uint32_t k = t2 / count, lp = t2 - k * count;
            const uint32_t l = parity_order(lp, count);
            const bool alive = run_leaf(T, S * NB1 + k, span_of(T, edges, first + l), resume[l], f, a);
            res[l * nsim + 2 * S + k] = pack_absolute(encode_post(T, f, a, alive), NB1);
            continue;
        }
        const uint32_t cls = task & 1u, pair = task >> 1;
        const uint32_t k = pair / nu, up = pair - k * nu;
        const uint32_t l = uniq[1 + up];
        const uint32_t nb0 = cls ? max_bits : 0u;
        const bool alive = run_leaf(T, k * NB1 + nb0, span_of(T, edges, first + l), resume[l], f, a);
        res[l * nsim + 2 * k + cls] = pack_normal(T, f, a, alive, nb0, cls);
    }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nu * S; e += blockDim.x) {
        const uint32_t up = e / S, k = e - up * S;
        uint32_t *r = res + (uint32_t)uniq[1 + up] * nsim + 2 * k;
        const uint32_t p0 = r[0];
        if (p0 & kPkShared) {
            r[0] = p0 & ~kPkShared;
            r[1] = p0 & ~kPkShared;
        }
    }
    __syncthreads();
    STAMP(4);
    // the other leaves take their representative's rows
    for (uint32_t e = threadIdx.x; e < count * 2 * S; e += blockDim.x) {
        const uint32_t l = e / (2 * S), c = e - l * (2 * S);
        const uint32_t r = rep[l];
        if (r != l) res[l * nsim + c] = res[r * nsim + c];
    }
    __syncthreads();
    if (lt_off && T.NS) {
        // which leaves can end stuck: those of the block from their rows, the few before it
        // (whose stuck codes may enter the block) from the tables
        // (only rows of states that can be met at the leaf's level count: stuck_row = row | level << 7)
        for (uint32_t e = threadIdx.x; e < count * T.nstuck_rows; e += blockDim.x) {
            const uint32_t l = e / T.nstuck_rows, rl = T.stuck_row[e - l * T.nstuck_rows];
            if ((rl >> 7) != ((uint32_t)((first + l) & 1ull) ^ T.lvl0)) continue;
            if (res[l * nsim + (rl & 127u)] & kPkStuck) atomicOr(&cap[(l + kStuckDepth) >> 5], 1u << ((l + kStuckDepth) & 31u));
        }
        for (uint32_t t = threadIdx.x; t < T.depth * T.nstuck_rows; t += blockDim.x) {
            const uint32_t j = t / T.nstuck_rows + 1, rl = T.stuck_row[t - (j - 1) * T.nstuck_rows];
            const uint32_t r = rl & 127u;
            if (first < (uint64_t)j + 1) continue;                  // leaf first - j >= 1
            const uint64_t i = first - j;
            if ((rl >> 7) != ((uint32_t)(i & 1ull) ^ T.lvl0)) continue;
            const uint64_t n = edges[i] - edges[i - 1] - 1;
            if (n > 0xfffffff0ull) continue;
            if (lt_lookup(lt_off, lt_n0, lt_pk, r, (uint32_t)(i & 1ull) ^ T.lvl0, (uint32_t)n) & kPkStuck) {
                atomicOr(&cap[(kStuckDepth - j) >> 5], 1u << ((kStuckDepth - j) & 31u));
            }
        }
        __syncthreads();
    }
}

// Phase B: tables.  Leaves with the same (level, length) share their normal
// rows, so a dense table tab[l * D + code] is built for the representatives
// only (a handful per block); the two skip entries are per leaf.  One step of
// a walk through leaf l is then   code < S*NB1 ? tab[rep[l] * D + code]
//                                               : skip[l][code - S*NB1]   (poison stays).
__device__ void block_expand(const LTab &T, const StuckCtx &sc, const uint64_t *edges, uint64_t first, uint32_t count,
                             uint16_t *tab, const uint32_t *res, const uint64_t *resume, const uint16_t *rep,
                             uint16_t (*skip)[2]) {
    const uint32_t S = T.S, NB1 = T.NB1, max_bits = T.max_bits;
    const uint32_t nsim = 2 * S + 2;
    const uint32_t SNB = S * NB1;
    for (uint32_t l = threadIdx.x; l < count; l += blockDim.x) {
        skip[l][0] = (uint16_t)(res[l * nsim + 2 * S] & 0xffffu);
        skip[l][1] = (uint16_t)(res[l * nsim + 2 * S + 1] & 0xffffu);
    }
    for (uint32_t l0 = 0; l0 < count; l0 += 64) {
      // representatives of this group of 64 leaves, as a wave-uniform bit mask
      const uint32_t lane = threadIdx.x & 63u;
      uint64_t reps = __ballot(l0 + lane < count && rep[l0 + lane] == l0 + lane);
      while (reps) {
        const uint32_t l = l0 + (uint32_t)__ffsll((long long)reps) - 1u;
        reps &= reps - 1;
        const uint32_t *r = res + l * nsim;
        for (uint32_t e = threadIdx.x; e < SNB; e += blockDim.x) {
            const uint32_t cur = e / NB1, nb = e - cur * NB1;
            const uint32_t pk = r[2 * cur + (nb >= max_bits ? 1u : 0u)];
            uint32_t out;
            if (pk & kPkAbsolute) {
                out = pk & 0xffffu;
            } else if (pk & kPkRelative) {
                uint32_t nbo = nb + ((pk >> 8) & 0xffffu);
                if (nbo >= NB1) nbo = NB1 - 1;
                out = (pk & 0xffu) * NB1 + nbo;
            } else if (pk & kPkStuck) {
                out = stuck_enter(T, sc, e);
            } else {
                // exact per-count simulation (rare)
                PSim f;
                Acc a;
                const bool alive = run_leaf(T, e, span_of(T, edges, first + l), resume[l], f, a);
                out = encode_post(T, f, a, alive);
            }
            tab[l * SNB + e] = (uint16_t)out;
        }
      }
    }
    __syncthreads();
}

// one leaf (l-th of the block that starts at edge `first`) applied to an abstract state code
__device__ __forceinline__ uint32_t leaf_step(const LTab &T, const StuckCtx &sc, uint64_t first, const uint16_t *tab,
                                              const uint16_t *rep, const uint16_t (*skip)[2], uint32_t SNB,
                                              uint32_t l, uint32_t s) {
    if (s < SNB) return tab[(uint32_t)rep[l] * SNB + s];
    if (s < SNB + 2) return skip[l][s - SNB];
    return s == SNB + 2 ? s : stuck_step(T, sc, first + l, s);
}

__device__ __forceinline__ void locate_block(const ScanParams &sp, uint32_t gb, uint32_t &cap, uint32_t &lb) {
    uint32_t lo = 0, hi = sp.f.num_captures;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sp.cap_block_off[mid] <= gb) lo = mid;
        else hi = mid;
    }
    cap = lo;
    lb = gb - sp.cap_block_off[lo];
}

// capture `cap` cannot be decoded by the scan: per capture in a batched run, else the whole call
__device__ __forceinline__ void scan_refuse(const ScanParams &sp, uint32_t cap, uint32_t reason) {
    if (sp.cap_fallback && sp.f.num_captures > 1) {
        atomicOr(&sp.cap_fallback[cap], reason);
        atomicOr(sp.f.flags, 2u);               // bit 1 of the header flags: some capture was refused
    } else {
        atomicOr(sp.fallback, reason);
    }
}

// the concrete incoming state: reset, the host's (shards), or the chunk before's (pipelined runs)
__device__ __forceinline__ FsmStateDev scan_first(const ScanParams &sp) {
    if (sp.first_dev) return sp.first_dev->st;
    return sp.first;            // zeros without have_first
}

// A capture's first span (samples 0 .. first edge; the whole capture when it
// has no edge) from the concrete incoming state.
__device__ __forceinline__ bool first_leaf(const LTab &T, const ScanParams &sp, const uint64_t *edges, uint64_t ne, PSim &f,
                           Acc &a) {
    const FsmStateDev fs = scan_first(sp);
    f.cur = sp.have_first ? fs.cur : 0u;
    f.nbits = sp.have_first ? fs.nbits : 0u;
    f.prev = sp.have_first ? fs.prev : 0u;
    const uint64_t k64 = sp.have_first ? fs.k : 0ull;
    f.k = k64 > kSat ? kSat : (uint32_t)k64;
    acc_init(a);
    const uint64_t n = ne ? edges[0] : sp.f.n_out;
    return sim_span(T, f, a, 0, T.lvl0, n, ne != 0);
}

__device__ __forceinline__ void write_event(LeafEvDev &ev, const Acc &a, const PSim &f, bool alive) {
    ev.napp = (uint8_t)(a.napp > 255u ? 255u : a.napp);
    ev.nout = (uint8_t)(a.nout > 255u ? 255u : a.nout);
    ev.nerr = (uint8_t)(a.nerr > 255u ? 255u : a.nerr);
    ev.flags = (uint8_t)((a.reset_seen ? 1u : 0u) | (alive ? 0u : 4u));
    ev.apps_at_reset = (uint8_t)a.apps_at_reset;
    ev.out_ab[0] = (uint8_t)a.out_ab0;
    ev.out_ab[1] = (uint8_t)a.out_ab1;
    ev.out_rb[0] = (uint8_t)a.out_rb0;
    ev.out_rb[1] = (uint8_t)a.out_rb1;
    ev.end_cur = (uint8_t)f.cur;
    ev.end_prev = (uint8_t)f.prev;
    ev.pad = 0;
    ev.appvals = a.appvals;
    ev.end_k = f.k;
    ev.out_pos[0] = a.out_pos0;
    ev.out_pos[1] = a.out_pos1;
    ev.err_pos = a.err_pos;
}

// A leaf's record.  Nearly every span leaves nothing behind, or a few appended bits: ONE word per leaf says so -- appends
// (8 bits) | 0x400: the 48-byte record holds the rest | passed through reset << 11 | appends before its last reset
// << 16 | the first eight appended bits << 24 -- and only a leaf with an output, an error, more than eight appends, a
// skipped rest-of-buffer at its end, or the capture's tail (whose end state goes out) gets the 48-byte record too.
// (Round 2 stored and read the 48 bytes of every leaf: 35 MB each way per 16 GiB capture.)
constexpr uint32_t kEvFull = 0x400u;
__device__ __forceinline__ void put_event(const ScanParams &sp, size_t at, const Acc &a, const PSim &f, bool alive, bool tail) {
    const uint32_t napp = a.napp > 255u ? 255u : a.napp;
    const bool full = tail || !alive || a.nout != 0 || a.nerr != 0 || napp > 8u;
    sp.ev_hot[at] = napp | (full ? kEvFull : 0u) | (a.reset_seen ? 0x800u : 0u) | ((a.apps_at_reset & 0xffu) << 16) |
                    ((a.appvals & 0xffu) << 24);
    if (full) write_event(sp.events[at], a, f, alive);
}

// ... and back (fin_block_scan)
__device__ __forceinline__ LeafEvDev get_event(const ScanParams &sp, size_t at) {
    const uint32_t hot = sp.ev_hot[at];
    if (hot & kEvFull) return sp.events[at];
    LeafEvDev ev;
    ev.napp = (uint8_t)(hot & 0xffu);
    ev.nout = ev.nerr = 0;
    ev.flags = (uint8_t)((hot >> 11) & 1u);
    ev.apps_at_reset = (uint8_t)((hot >> 16) & 0xffu);
    ev.out_ab[0] = ev.out_ab[1] = 0;
    ev.out_rb[0] = ev.out_rb[1] = 0xffu;
    ev.end_cur = ev.end_prev = ev.pad = 0;
    ev.appvals = hot >> 24;
    ev.end_k = 0;
    ev.out_pos[0] = ev.out_pos[1] = 0;
    ev.err_pos = 0;
    return ev;
}

}  // namespace

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

constexpr int kSimThreads = 256;        // leaf / emit kernels (register-heavy simulations)
constexpr int kChunk = 16;              // leaves per composition chunk inside a block
constexpr int kFinBlock = 1024;         // leaves per finish block

extern __shared__ __attribute__((aligned(16))) unsigned char scan_smem[];

struct BlockLds {
    uint16_t *tab;      // [LB][D]   leaf tables
    uint16_t *ctab;     // [LB/16][D] chunk tables
    uint32_t *res;      // [LB][2S+2]
};

// leaf tables hold the normal codes only (SNB = S * NB1); chunk tables the whole domain
// (the leaf-table area also stages up to kGroup - 1 block tables in the emit kernel)
__host__ __device__ __forceinline__ size_t leaf_tab_bytes(uint32_t LB, uint32_t D, uint32_t SNB) {
    const size_t a = (size_t)LB * SNB * 2, g = (size_t)16 * ((D + 7u) & ~7u) * 2;
    return ((a > g ? a : g) + 15) & ~(size_t)15;
}

__device__ __forceinline__ BlockLds carve(uint32_t LB, uint32_t D, uint32_t SNB) {
    BlockLds b;
    size_t off = 0;
    b.tab = reinterpret_cast<uint16_t *>(scan_smem + off);
    off += leaf_tab_bytes(LB, D, SNB);
    b.ctab = reinterpret_cast<uint16_t *>(scan_smem + off);
    off += ((size_t)((LB + kChunk - 1) / kChunk) * D * 2 + 15) & ~(size_t)15;
    b.res = reinterpret_cast<uint32_t *>(scan_smem + off);
    return b;
}

static size_t block_lds_bytes(uint32_t LB, uint32_t D, uint32_t S, uint32_t SNB) {
    size_t off = leaf_tab_bytes(LB, D, SNB);
    off += ((size_t)((LB + kChunk - 1) / kChunk) * D * 2 + 15) & ~(size_t)15;
    off += (size_t)LB * (2 * S + 2) * 4;
    return off;
}

// One leaf applied to an abstract state code without any stored row: the span tables are searched
// for the leaf's length (what block_sims does for the block's classes), position-dependent and
// count-dependent cases are simulated.  i = the leaf (edge index in the capture), e_before / e_at =
// edges[i - 1], edges[i].  Without span tables every step is a simulation.
__device__ __noinline__ uint32_t leaf_step_fly(const LTab &T, const StuckCtx &sc, uint64_t i, uint64_t e_before,
                                               uint64_t e_at, uint32_t s) {
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1;
    if (s == SNB + 2) return s;                                 // poison stays
    if (s > SNB + 2) return stuck_step(T, sc, i, s);
    Span span;
    span.pos0 = e_before + 1;
    span.n = e_at - e_before - 1;
    span.L = (uint32_t)(i & 1ull) ^ T.lvl0;
    span.has_edge = true;
    span.prefix = 0;
    uint64_t resume = 0;
    if (sc.lt_off) {
        if (s < SNB) {
            if (span.n <= 0xfffffff0ull) {
                const uint32_t cur = s / NB1, nb = s - cur * NB1;
                const uint32_t p0 = lt_lookup(sc.lt_off, sc.lt_n0, sc.lt_pk, 2 * cur, span.L, (uint32_t)span.n);
                uint32_t pk = p0 & ~kPkShared;
                if (!(p0 & kPkShared) && nb >= T.max_bits) pk = lt_lookup(sc.lt_off, sc.lt_n0, sc.lt_pk, 2 * cur + 1, span.L, (uint32_t)span.n);
                if (pk & kPkAbsolute) return pk & 0xffffu;
                if (pk & kPkRelative) {
                    const uint32_t nbo = nb + ((pk >> 8) & 0xffffu);
                    return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                }
                if (pk & kPkStuck) return stuck_enter(T, sc, s);
            }
            resume = next_buffer_start(T, e_before);            // depends on the exact bit count, or a giant span
        } else {
            // a skip state: skipping ends at `resume`; from there the machine starts in reset -- the normal
            // row (reset, no bits) of a shorter span when the level before the skip equals the span's, a
            // special row otherwise (block_sims)
            const uint32_t kk = s - SNB;
            const uint64_t end_const = span.pos0 + span.n, last = end_const + 1;
            resume = next_buffer_start(T, e_before);
            if (resume >= last) return s;                        // still skipping when the span ends
            const uint64_t n2 = resume >= end_const ? 0 : end_const - resume;
            if (n2 <= 0xfffffff0ull) {
                const uint32_t pk = lt_lookup(sc.lt_off, sc.lt_n0, sc.lt_pk, kk == span.L ? 0u : 2 * S + kk, span.L, (uint32_t)n2);
                if (pk & kPkAbsolute) return pk & 0xffffu;
                if (pk & kPkRelative) {
                    const uint32_t nbo = (pk >> 8) & 0xffffu;   // from a bit count of 0
                    return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                }
            }
        }
    } else {
        resume = next_buffer_start(T, e_before);
    }
    PSim f;
    Acc a;
    const bool alive = run_leaf(T, s, span, resume, f, a);
    return encode_post(T, f, a, alive);
}

// chunk tables: ctab[c][d] = the block's leaves 16c .. 16c+15 composed
//
// Only the codes a span can actually be entered in are walked: reach_lv = the two plain lists
// "normal / skip / poison codes met at level 0 / at level 1" (the closure of the span tables'
// results, split by level on the host); every other entry is poison, so a path that leaves the
// closure after all makes the capture fall back.  Without the lists every base code is walked.
//
// A chain is up to 16 dependent LDS reads.  One wave takes one chunk (and a slab of 64 * kIlp
// codes) at a time: the leaf is then the same in every lane -- its row base and skip pair are
// fetched one step ahead, off the dependent path -- and a lane walks kIlp codes side by side,
// branch-free on the state (stuck codes wait like poison), so that one step costs one LDS
// latency.  The kernel is bound by the LDS (random 2-byte reads, bank conflicts), so the reads
// are what is saved: after the chunk's FIRST leaf the few hundred codes have collapsed to a
// few dozen distinct states (the machine's states times the bit counts that survive), and
// only those walk the other 15 leaves (`scratch`: a bit set per wave to find them; the chunk's
// own ctab row holds their results until the codes have picked them up).
// A chain that gets stuck (rare) stops and is finished out of the hot loop.
constexpr int kComposeIlp = 7;

// st[] through the leaves la .. lb-1 of the block (uniform bounds), kIlp chains per lane
template <int kIlp>
__device__ __forceinline__ void compose_walk(const LTab &T, const StuckCtx &sc, uint64_t first, const BlockLds &b, uint32_t SNB,
                                             const uint16_t *rep, const uint16_t (*skip)[2], uint32_t la, uint32_t lb,
                                             uint32_t (&st)[kIlp]) {
    const uint32_t *skipw = reinterpret_cast<const uint32_t *>(skip);     // [l]: skip[l][0] | skip[l][1] << 16
    uint32_t at[kIlp];          // leaf in front of which the chain got stuck (lb: it did not)
#pragma unroll
    for (int j = 0; j < kIlp; ++j) at[j] = lb;
    if (la < lb) {
        uint32_t rb = (uint32_t)rep[la] * SNB, sk = skipw[la];
        for (uint32_t l = la; l < lb; ++l) {
            const uint32_t ln = l + 1 < lb ? l + 1 : l;
            const uint32_t repn = rep[ln], skn = skipw[ln];
            uint32_t a[kIlp];
#pragma unroll
            for (int j = 0; j < kIlp; ++j) a[j] = b.tab[rb + (st[j] < SNB ? st[j] : 0u)];
#pragma unroll
            for (int j = 0; j < kIlp; ++j) {
                const uint32_t v = st[j];
                const uint32_t k = v == SNB + 1 ? sk >> 16 : sk & 0xffffu;
                const uint32_t nv = v < SNB ? a[j] : (v < SNB + 2 ? k : v);
                at[j] = (nv > SNB + 2 && v <= SNB + 2) ? l + 1 : at[j];
                st[j] = nv;
            }
            rb = repn * SNB;
            sk = skn;
        }
    }
#pragma unroll
    for (int j = 0; j < kIlp; ++j) {
        if (st[j] > SNB + 2) {
            for (uint32_t l = at[j]; l < lb; ++l) st[j] = leaf_step(T, sc, first, b.tab, rep, skip, SNB, l, st[j]);
        }
    }
}

__device__ __forceinline__ void wave_sync_lds() {
    // the LDS executes one wave's accesses in order: only keep the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ void compose_chunks(const LTab &T, const StuckCtx &sc, uint64_t first, const uint32_t *cap,
                               const BlockLds &b, uint32_t D, uint32_t SNB, uint32_t count, const uint16_t *rep,
                               const uint16_t (*skip)[2], const uint16_t *reach_lv, uint32_t nreach_lv,
                               uint32_t *scratch, uint32_t scratch_words, uint64_t *dbg = nullptr) {
    STAMP(0);
    constexpr int kIlp = kComposeIlp;
    const uint32_t nch = (count + kChunk - 1) / kChunk;
    const uint32_t D0 = SNB + 3;
    const uint32_t NR = reach_lv ? nreach_lv : D0;
    for (uint32_t i = threadIdx.x; i < nch * D; i += blockDim.x) b.ctab[i] = (uint16_t)(SNB + 2);
    __syncthreads();
    STAMP(1);
    const uint32_t lane = threadIdx.x & 63u, nwaves = blockDim.x >> 6;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // scalar: uniform loop bounds
    const uint32_t slabs = (NR + 64u * kIlp - 1u) / (64u * kIlp);
    // distinct states after the first leaf: a bit set of the base codes + their list, per wave
    const uint32_t set_words = (D0 + 31u) / 32u;
    const uint32_t per_wave = set_words + (64u * kIlp + 1u) / 2u;
    const bool dedupe = slabs == 1 && set_words <= 64u && per_wave * nwaves <= scratch_words;
    uint32_t *seen = scratch + wave * per_wave;
    uint16_t *list = reinterpret_cast<uint16_t *>(seen + set_words);
    for (uint32_t unit = wave; unit < nch * slabs; unit += nwaves) {
        const uint32_t c = unit / slabs, slab = unit - c * slabs;
        const uint32_t l0 = c * kChunk, l1 = min(l0 + (uint32_t)kChunk, count);
        uint32_t st[kIlp], code[kIlp];
        bool valid[kIlp];
#pragma unroll
        for (int j = 0; j < kIlp; ++j) {
            const uint32_t idx = slab * 64u * kIlp + (uint32_t)j * 64u + lane;
            valid[j] = idx < NR;
            code[j] = reach_lv ? (uint32_t)reach_lv[valid[j] ? idx : 0u] : idx;
            st[j] = valid[j] ? code[j] : SNB + 2;
        }
        STAMP(2);
        uint16_t *row = b.ctab + c * D;
        if (!dedupe || l1 - l0 < 3) {
            compose_walk<kIlp>(T, sc, first, b, SNB, rep, skip, l0, l1, st);
        } else {
            compose_walk<kIlp>(T, sc, first, b, SNB, rep, skip, l0, l0 + 1, st);
            // ---- the distinct base codes among st[] ---------------------------------------------
            if (lane < set_words) seen[lane] = 0;
            wave_sync_lds();
#pragma unroll
            for (int j = 0; j < kIlp; ++j)
                if (valid[j] && st[j] < D0) atomicOr(&seen[st[j] >> 5], 1u << (st[j] & 31u));
            wave_sync_lds();
            uint32_t bits = lane < set_words ? seen[lane] : 0u;
            const uint32_t cnt = (uint32_t)__popc(bits);
            uint32_t inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d);
                if ((int)lane >= d) inc += t;
            }
            const uint32_t nd = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            uint32_t at = inc - cnt;
            while (bits) {
                const uint32_t bit = (uint32_t)__ffs((int)bits) - 1u;
                bits &= bits - 1u;
                list[at++] = (uint16_t)(lane * 32u + bit);
            }
            wave_sync_lds();
            // ---- they walk the rest of the chunk; results parked in the chunk's own row -----------
            for (uint32_t base = 0; base < nd; base += 64u * kIlp) {
                uint32_t ds[kIlp], dv[kIlp];
#pragma unroll
                for (int j = 0; j < kIlp; ++j) {
                    const uint32_t i = base + (uint32_t)j * 64u + lane;
                    dv[j] = i < nd ? (uint32_t)list[i] : SNB + 2;
                    ds[j] = dv[j];
                }
                compose_walk<kIlp>(T, sc, first, b, SNB, rep, skip, l0 + 1, l1, ds);
#pragma unroll
                for (int j = 0; j < kIlp; ++j)
                    if (base + (uint32_t)j * 64u + lane < nd) row[dv[j]] = (uint16_t)ds[j];
            }
            wave_sync_lds();
            // ---- every code picks up the result of the state its first leaf led to ------------------
            uint32_t stuck1 = 0;
#pragma unroll
            for (int j = 0; j < kIlp; ++j) {
                if (st[j] < D0) st[j] = row[st[j]];
                else stuck1 |= 1u << j;         // got stuck on the first leaf (rare): finished below
            }
            wave_sync_lds();
            for (uint32_t i = lane; i < nd; i += 64) row[list[i]] = (uint16_t)(SNB + 2);
            wave_sync_lds();
            if (stuck1) {
#pragma unroll
                for (int j = 0; j < kIlp; ++j) {
                    if (!((stuck1 >> j) & 1u)) continue;
                    for (uint32_t l = l0 + 1; l < l1; ++l) st[j] = leaf_step(T, sc, first, b.tab, rep, skip, SNB, l, st[j]);
                }
            }
        }
        STAMP(3);
#pragma unroll
        for (int j = 0; j < kIlp; ++j)
            if (valid[j]) row[code[j]] = (uint16_t)st[j];
    }
    STAMP(4);
    // chains that START in a stuck code: only behind a leaf that can end stuck -- nowhere in
    // a clean capture
    bool any = false;
    for (uint32_t w = 0; w < kCapWords; ++w) any = any || cap[w] != 0;
    if (any && T.NS) {
        const uint32_t nst = T.NS * T.depth;
        for (uint32_t item = threadIdx.x; item < nch * nst; item += blockDim.x) {
            const uint32_t c = item / nst, code = D0 + (item - c * nst);
            const uint32_t la = c * kChunk, lb = min((c + 1) * kChunk, count);
            // STUCK_d enters the chunk only if the leaf d before it can end stuck
            const uint32_t ob = la + kStuckDepth - ((code - D0) / T.NS + 1);
            if (!(cap[ob >> 5] & (1u << (ob & 31u)))) continue;
            uint32_t v = code;
            for (uint32_t l = la; l < lb; ++l) v = leaf_step(T, sc, first, b.tab, rep, skip, SNB, l, v);
            b.ctab[c * D + code] = (uint16_t)v;
        }
    }
    __syncthreads();
}

// prefix of per-capture block counts: scan blocks (regular leaves) and finish
// blocks (all leaves); one workgroup
__global__ __launch_bounds__(kScanThreads) void scan_layout_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t wtot[kScanThreads / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t nc = sp.f.num_captures;
    const uint32_t chunk = (nc + kScanThreads - 1) / kScanThreads;
    const uint32_t lo = min(tid * chunk, nc), hi = min(lo + chunk, nc);
    uint32_t sum = 0, sum2 = 0, sum3 = 0, sum4 = 0;
    for (uint32_t c = lo; c < hi; ++c) {
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, c, e0);
        const uint64_t regular = ne > 1 ? ne - 1 : 0;
        const uint32_t nblk = (uint32_t)((regular + sp.leaf_block - 1) / sp.leaf_block);
        sum += nblk;
        sum2 += (uint32_t)((ne + 1 + kFinBlock - 1) / kFinBlock);
        sum3 += (nblk + 15u) / 16u;
        sum4 += (nblk + 63u) / 64u;
    }
    uint32_t tot = 0, tot2 = 0, tot3 = 0, tot4 = 0;
    uint32_t run = wg_inclusive_sum(sum, wtot, &tot) - sum;
    uint32_t run2 = wg_inclusive_sum(sum2, wtot, &tot2) - sum2;
    uint32_t run3 = wg_inclusive_sum(sum3, wtot, &tot3) - sum3;
    uint32_t run4 = wg_inclusive_sum(sum4, wtot, &tot4) - sum4;
    for (uint32_t c = lo; c < hi; ++c) {
        sp.cap_block_off[c] = run;
        sp.fin_off[c] = run2;
        sp.cap_group_off[c] = run3;
        sp.cap_super_off[c] = run4;
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, c, e0);
        const uint64_t regular = ne > 1 ? ne - 1 : 0;
        const uint32_t nblk = (uint32_t)((regular + sp.leaf_block - 1) / sp.leaf_block);
        run += nblk;
        run2 += (uint32_t)((ne + 1 + kFinBlock - 1) / kFinBlock);
        run3 += (nblk + 15u) / 16u;
        run4 += (nblk + 63u) / 64u;
    }
    if (tid == kScanThreads - 1) {
        sp.cap_block_off[nc] = tot;
        sp.fin_off[nc] = tot2;
        sp.cap_group_off[nc] = tot3;
        sp.cap_super_off[nc] = tot4;
        if (tot > sp.total_blocks_cap || tot2 > sp.fin_blocks_cap) atomicOr(sp.fallback, (uint32_t)kFbBlocks);
        // more level changes than the edge list holds: its tail was never written and the buffers
        // sized by the capacity would be overrun -- no scan (the host reports OOKD_ERR_CAPACITY)
        if (sp.edge_overflow && *sp.edge_overflow) atomicOr(sp.fallback, (uint32_t)kFbBlocks);
    }
}

__global__ __launch_bounds__(kSimThreads) void scan_leaf_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ LTab T;
    __shared__ uint64_t s_resume[256];
    __shared__ uint16_t s_rep[256], s_uniq[258];
    __shared__ __attribute__((aligned(4))) uint16_t s_skip[256][2];
    __shared__ uint32_t s_cap[kCapWords];
    __shared__ uint32_t s_lt[kLtLdsWords];
    copy_ltab(T, sp.ltab);
    __syncthreads();
    if (threadIdx.x == 0) T.lvl0 = sp.has_prev ? fsm_level_at(sp.f, 0, -1) : 0u;
    __syncthreads();
    if (*sp.fallback) return;
    const uint32_t D = T.D, LB = sp.leaf_block;
    const BlockLds b = carve(LB, D, T.S * T.NB1);
    // the span tables (a few hundred words for the shipped devices) are searched from LDS
    const uint32_t *lt_off = sp.lt_off, *lt_n0 = sp.lt_n0, *lt_pk = sp.lt_pk;
    if (lt_off) {
        const uint32_t noff = 2 * (2 * T.S + 2) + 1, nint = lt_off[noff - 1];
        if (noff + 2 * nint <= kLtLdsWords) {
            for (uint32_t i = threadIdx.x; i < noff; i += blockDim.x) s_lt[i] = sp.lt_off[i];
            for (uint32_t i = threadIdx.x; i < nint; i += blockDim.x) {
                s_lt[noff + i] = sp.lt_n0[i];
                s_lt[noff + nint + i] = sp.lt_pk[i];
            }
            lt_off = s_lt;
            lt_n0 = s_lt + noff;
            lt_pk = s_lt + noff + nint;
            __syncthreads();
        }
    }
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    for (uint32_t gb = blockIdx.x; gb < total + sp.f.num_captures; gb += gridDim.x) {
        if (gb >= total) {
            // one extra item per capture: its first span (samples 0 .. first edge) from the
            // concrete incoming state, for the walk kernel (a cold simulation there would sit
            // on the critical path; here it runs beside the blocks)
            if (threadIdx.x == 0) {
                const uint32_t cap = gb - total;
                uint64_t e0;
                const uint64_t ne = cap_edges(sp.f, cap, e0);
                PSim f;
                Acc a;
                const bool alive = first_leaf(T, sp, sp.f.edges + e0, ne, f, a);
                sp.cap_first[cap] = (uint16_t)encode_post(T, f, a, alive);
            }
            continue;
        }
        uint32_t cap, lb;
        locate_block(sp, gb, cap, lb);
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, cap, e0);
        const uint64_t *edges = sp.f.edges + e0;
        const uint64_t first = 1 + (uint64_t)lb * LB;
        const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
        const uint64_t st0 = __builtin_amdgcn_s_memtime();
        const StuckCtx sc{edges, lt_off, lt_n0, lt_pk};
        block_sims(T, edges, first, count, b.res, s_resume, s_rep, s_uniq, lt_off, lt_n0, lt_pk, s_cap,
                   (sp.f.debug && gb == 2) ? sp.f.debug + 40 : nullptr);
        const uint64_t st1 = __builtin_amdgcn_s_memtime();
        block_expand(T, sc, edges, first, count, b.tab, b.res, s_resume, s_rep, s_skip);
        const uint64_t st2 = __builtin_amdgcn_s_memtime();
        {
            // (a chunk starts on a multiple of 16 leaves: all chunks of the block start at the block's level)
            const uint32_t lvl = (uint32_t)(first & 1ull) ^ T.lvl0;
            const uint16_t *lv = sp.reach ? sp.reach + sp.nreach + (lvl ? sp.nreach_lv[0] : 0u) : nullptr;
            // the packed rows are dead once the leaf tables are expanded: their LDS is the composition's scratch
            compose_chunks(T, sc, first, s_cap, b, D, T.S * T.NB1, count, s_rep, s_skip, lv, sp.reach ? sp.nreach_lv[lvl] : 0u,
                           b.res, LB * (2 * T.S + 2), (sp.f.debug && gb == 2) ? sp.f.debug + 48 : nullptr);
        }
        const uint64_t st3 = __builtin_amdgcn_s_memtime();
        if (sp.f.debug && threadIdx.x == 0 && gb < 8) {
            sp.f.debug[4 * gb + 0] = st1 - st0;
            sp.f.debug[4 * gb + 1] = st2 - st1;
            sp.f.debug[4 * gb + 2] = st3 - st2;
            sp.f.debug[4 * gb + 3] = s_uniq[0] | ((uint64_t)(s_cap[0] | s_cap[1] | s_cap[2]) << 32);
        }
        // the block's table: every abstract state walks the chunk tables -- all this kernel leaves behind
        // (2 * Dp bytes per 64 leaves; the leaves' entry codes come from scan_entry_kernel)
        const uint32_t nch = (count + kChunk - 1) / kChunk;
        for (uint32_t d = threadIdx.x; d < D; d += blockDim.x) {
            uint32_t s = d;
            for (uint32_t c = 0; c < nch; ++c) s = b.ctab[c * D + s];
            sp.block_tab[(size_t)gb * sp.Dp + d] = (uint16_t)s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// leaf kernel, table-driven form: ONE WAVE per block of 64 leaves
// ---------------------------------------------------------------------------
// scan_leaf_kernel above gives a block to a workgroup of four waves, and most of a block's
// steps are narrow (64 leaves = one wave's worth; the other three wait at barriers) and long
// chains of dependent, cheap instructions: it is bound by instruction issue and latency on the
// one busy wave, with the expanded leaf tables (2 * S * NB1 bytes per leaf) limiting a CU to
// two or three blocks at a time.  With span tables a row of a leaf is one table search, so here
// lane l simply owns leaf l: its 2S+2 packed rows straight from the span tables (no search for
// equal spans, no expanded tables), and the chunk composition evaluates the packed rows
// directly (a multiply-high for state / bit count, one LDS read, a few selects per step).  LDS
// per block: the packed rows + four chunk tables (~6 KB for the shipped devices), no workgroup
// barrier anywhere, a dozen blocks per CU at a time.  The block tables are the same bit for bit.

struct WaveLds {
    uint32_t *res;      // [64][2S+4]: (state, class) rows, skip 0, skip 1, poison, spare
    uint16_t *ctab;     // [64 / kChunk][D]
};
__host__ __device__ __forceinline__ size_t wave_lds_bytes(uint32_t D, uint32_t S) {
    return (size_t)64 * (2 * S + 4) * 4 + ((size_t)(64 / kChunk) * D * 2 + 15) / 16 * 16;
}

// one step of the slow path: everything leaf_step_fly knows, from the block's packed rows
__device__ __forceinline__ uint32_t wave_step_slow(const LTab &T, const StuckCtx &sc, const uint64_t *edges, uint64_t first,
                                                uint32_t l, const uint32_t *res, const uint64_t *resume, uint32_t s) {
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1;
    const uint32_t *row = res + l * (2 * S + 4);
    if (s >= SNB) {
        if (s < SNB + 2) return row[2 * S + (s - SNB)] & 0xffffu;
        return s == SNB + 2 ? s : stuck_step(T, sc, first + l, s);
    }
    const uint32_t cur = s / NB1, nb = s - cur * NB1;
    const uint32_t pk = row[2 * cur + (nb >= T.max_bits ? 1u : 0u)];
    if (pk & kPkAbsolute) return pk & 0xffffu;
    if (pk & kPkRelative) {
        const uint32_t nbo = nb + ((pk >> 8) & 0xffffu);
        return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
    }
    if (pk & kPkStuck) return stuck_enter(T, sc, s);
    PSim f;                     // row depends on the exact bit count (rare): simulate
    Acc a;
    const bool alive = run_leaf(T, s, span_of(T, edges, first + l), resume[l], f, a);
    return encode_post(T, f, a, alive);
}

// A whole chain through the slow path, leaves la .. lb-1: ONE call (the kernel that calls it keeps ~190 registers
// live; a call per step had it save and restore them sixteen times per chain).
__device__ __noinline__ uint32_t wave_chain_slow(const LTab &T, const StuckCtx &sc, const uint64_t *edges, uint64_t first,
                                                 uint32_t la, uint32_t lb, const uint32_t *res, const uint64_t *resume, uint32_t v) {
    for (uint32_t l = la; l < lb; ++l) v = wave_step_slow(T, sc, edges, first, l, res, resume, v);
    return v;
}

// K chains per lane, each through the 16 leaves of its chunk (a short last chunk is padded with identity rows).
// A chain's state is kept as (state, bit count) -- skip / poison codes as state S with "bit count" 0 / 1 / 2,
// so that code = state * NB1 + bit count throughout -- which is what the packed rows are made of: no division
// per step, every product fits 24 bits.  Rows are 2S + 4 words: (state, class) pairs, skip 0, skip 1, poison
// (-> poison), spare.  A row that is neither absolute nor relative (stuck, bit-count sensitive, position
// dependent) is walked as if it were relative -- garbage, but inside the tables -- and flags the chain, which is
// then redone from its start by the slow path.  ~22 instructions per chain and step.
template <int K>
__device__ __forceinline__ void wave_walk(const LTab &T, const uint32_t *res, uint32_t nsim, uint32_t rcpNB1,
                                          const uint32_t (&la)[K], uint32_t (&st)[K], uint32_t &trapped) {
    const uint32_t S = T.S, NB1 = T.NB1, max_bits = T.max_bits;
    uint32_t cur[K], nb[K], ro[K], good[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        cur[j] = __umulhi(st[j], rcpNB1);                       // st / NB1 (exact below 2^16), once per chain
        nb[j] = st[j] - cur[j] * NB1;
        ro[j] = __umul24(la[j], nsim);
        good[j] = 0x80000000u;
    }
    for (uint32_t step = 0; step < (uint32_t)kChunk; ++step) {
        uint32_t pk[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t col = cur[j] < S ? 2u * cur[j] + (nb[j] >= max_bits ? 1u : 0u) : 2u * S + (nb[j] & 3u);
            pk[j] = res[ro[j] + col];
            ro[j] += nsim;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t p = pk[j];
            const uint32_t acur = (p >> 16) & 0xffu;                                    // absolute: the end code's state
            const uint32_t anb = (p & 0xffffu) - __umul24(acur, NB1);
            const uint32_t rnb = min(nb[j] + ((p >> 8) & 0xffffu), NB1 - 1u);           // relative: bits appended
            const bool isabs = (int32_t)p < 0;                                          // kPkAbsolute = bit 31
            good[j] &= p | (p << 2);                                                    // bit 31 | bit 29 (kPkRelative)
            cur[j] = isabs ? acur : (p & 0xffu);
            nb[j] = isabs ? anb : rnb;
        }
    }
    trapped = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        st[j] = __umul24(cur[j], NB1) + nb[j];
        trapped |= (good[j] & 0x80000000u) ? 0u : 1u << j;
    }
}
static_assert(kPkAbsolute == 0x80000000u && kPkRelative == 0x20000000u, "wave_walk tests these bits by position");

template <int K>
__device__ __forceinline__ void wave_compose(const LTab &T, const StuckCtx &sc, const uint64_t *edges, uint64_t first,
                                             const WaveLds &b, const uint64_t *resume, uint32_t D, uint32_t count,
                                             const uint16_t *reach_lv, uint32_t NR, uint32_t rcpNB1, uint32_t base) {
    const uint32_t nsim = 2 * T.S + 4;
    const uint32_t nch = (count + kChunk - 1) / kChunk, nitem = nch * NR;
    const uint32_t lane = threadIdx.x;
    uint32_t la[K], st[K], code[K], dst[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const uint32_t item = base + (uint32_t)j * 64u + lane;
        const bool valid = item < nitem;
        const uint32_t it = valid ? item : 0u;
        const uint32_t c = it / NR, idx = it - c * NR;
        code[j] = reach_lv ? (uint32_t)reach_lv[idx] : idx;
        la[j] = c * kChunk;
        st[j] = code[j];
        dst[j] = valid ? c * D + code[j] : 0xffffffffu;
    }
    uint32_t trapped;
    wave_walk<K>(T, b.res, nsim, rcpNB1, la, st, trapped);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (dst[j] == 0xffffffffu) continue;
        if ((trapped >> j) & 1u) {
            // met a row the fast walk does not know: again from the start, one full step at a time
            st[j] = wave_chain_slow(T, sc, edges, first, la[j], min(la[j] + (uint32_t)kChunk, count), b.res, resume, code[j]);
        }
        b.ctab[dst[j]] = (uint16_t)st[j];
    }
}

// (four chains per lane and three waves per SIMD instead of eight and two: 150 -> 142 us at 16 GiB, but 39 -> 44 us
//  on a 1 GiB capture, where a block's own latency is all there is: not kept)
__global__ __launch_bounds__(64) void scan_leaf_wave_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);
    if (sp.sync_try && !(*sp.sync_fail & 1u)) return;          // the entry codes are there already (scan_syncwalk_kernel)
    __shared__ LTab T;
    __shared__ uint64_t s_resume[64];
    __shared__ uint32_t s_cap[kCapWords];
    __shared__ uint32_t s_lt[kLtLdsWords];
    // Start-up in ONE round trip to memory: a workgroup handles one or two blocks (the grid is several times what
    // the chip holds -- the hardware hands workgroups out as others finish, which is what evens out the blocks
    // behind a glitch), so what it does before its first block counts.  Every word of the table image, the span
    // tables (their size comes with the launch: the launcher picks this kernel only when they fit the LDS) and
    // the fallback word is requested before any of them is waited for.
    const uint32_t lane = threadIdx.x;
    const uint32_t noff = 2 * (2 * sp.S + 2) + 1, nint = (sp.lt_words - noff) / 2u;
    {
        constexpr uint32_t kT4 = (uint32_t)(sizeof(LTab) / 16), kTper = (kT4 + 63u) / 64u, kLper = (kLtLdsWords + 63u) / 64u;
        const uint4 *tsrc = static_cast<const uint4 *>(sp.ltab);
        uint4 tv[kTper];
        uint32_t lv[kLper];
#pragma unroll
        for (uint32_t r = 0; r < kTper; ++r) tv[r] = lane + 64u * r < kT4 ? tsrc[lane + 64u * r] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t r = 0; r < kLper; ++r) {
            const uint32_t i = lane + 64u * r;
            lv[r] = i < noff ? sp.lt_off[i] : (i < noff + nint ? sp.lt_n0[i - noff] : (i < noff + 2u * nint ? sp.lt_pk[i - noff - nint] : 0u));
        }
        const uint32_t fb = *sp.fallback;
        const uint32_t lvl0 = sp.has_prev ? fsm_level_at(sp.f, 0, -1) : 0u;
        uint4 *tdst = reinterpret_cast<uint4 *>(&T);
#pragma unroll
        for (uint32_t r = 0; r < kTper; ++r)
            if (lane + 64u * r < kT4) tdst[lane + 64u * r] = tv[r];
#pragma unroll
        for (uint32_t r = 0; r < kLper; ++r)
            if (lane + 64u * r < noff + 2u * nint) s_lt[lane + 64u * r] = lv[r];
        wave_sync_lds();
        if (lane == 0) T.lvl0 = lvl0;
        wave_sync_lds();
        if (fb) return;
    }
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1, D = T.D, nsim = 2 * S + 4, max_bits = T.max_bits;
    const uint32_t D0 = SNB + 3;
    const uint32_t rcpNB1 = (uint32_t)((0x100000000ull + NB1 - 1) / NB1);
    WaveLds b;
    b.res = reinterpret_cast<uint32_t *>(scan_smem);
    b.ctab = reinterpret_cast<uint16_t *>(scan_smem + (size_t)64 * nsim * 4);
    const uint32_t *const lt_off = s_lt, *const lt_n0 = s_lt + noff, *const lt_pk = s_lt + noff + nint;
    const uint32_t LB = sp.leaf_block;          // 64
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    for (uint32_t gb = blockIdx.x; gb < total + sp.f.num_captures; gb += gridDim.x) {
        if (gb >= total) {
            // one extra item per capture: its first span from the concrete incoming state (for the walk kernel)
            if (lane == 0) {
                const uint32_t cap = gb - total;
                uint64_t e0;
                const uint64_t ne = cap_edges(sp.f, cap, e0);
                PSim f;
                Acc a;
                const bool alive = first_leaf(T, sp, sp.f.edges + e0, ne, f, a);
                sp.cap_first[cap] = (uint16_t)encode_post(T, f, a, alive);
            }
            continue;
        }
        uint32_t cap, lb;
        locate_block(sp, gb, cap, lb);
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, cap, e0);
        const uint64_t *edges = sp.f.edges + e0;
        const uint64_t first = 1 + (uint64_t)lb * LB;
        const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
        const StuckCtx sc{edges, lt_off, lt_n0, lt_pk};
        const uint32_t nch = (count + kChunk - 1) / kChunk;
        uint64_t *dbg = (sp.f.debug && gb == 2) ? sp.f.debug + 48 : nullptr;
        STAMP(0);
        // ---- lane l = leaf l: its packed rows --------------------------------------------------
        if (lane < kCapWords) s_cap[lane] = 0;
        for (uint32_t i = lane; i < nch * D; i += 64) b.ctab[i] = (uint16_t)(SNB + 2);
        if (lane < count) {
            const uint32_t l = lane;
            const Span span = span_of(T, edges, first + l);
            const uint64_t rs = next_buffer_start(T, span.pos0 - 1);
            s_resume[l] = rs;
            uint32_t *row = b.res + l * nsim;
            if (span.n <= 0xfffffff0ull) {
                const uint32_t n = (uint32_t)span.n;
                for (uint32_t k = 0; k < S; ++k) {
                    const uint32_t p0 = lt_lookup(lt_off, lt_n0, lt_pk, 2 * k, span.L, n);
                    const uint32_t p1 = (p0 & kPkShared) ? p0 : lt_lookup(lt_off, lt_n0, lt_pk, 2 * k + 1, span.L, n);
                    row[2 * k] = p0 & ~kPkShared;
                    row[2 * k + 1] = p1 & ~kPkShared;
                }
            } else {
                // too long for the tables' 32-bit lengths: simulate both classes of every state
                for (uint32_t k = 0; k < S; ++k) {
                    PSim f;
                    Acc a;
                    bool alive = run_leaf(T, k * NB1, span, rs, f, a);
                    const uint32_t p0 = pack_normal(T, f, a, alive, 0u, 0u);
                    alive = run_leaf(T, k * NB1 + max_bits, span, rs, f, a);
                    const uint32_t p1 = (p0 & kPkShared) ? p0 : pack_normal(T, f, a, alive, max_bits, 1u);
                    row[2 * k] = p0 & ~kPkShared;
                    row[2 * k + 1] = p1 & ~kPkShared;
                }
            }
            // skip rows.  Skipping ends at `rs`; from there the machine starts in reset -- the normal
            // row (reset, few bits) of a shorter span when the level before the skip equals the span's, a
            // special row otherwise.
            const uint64_t end_const = span.pos0 + span.n, last = end_const + 1;
            uint32_t skip_pair = 0;
            for (uint32_t kk = 0; kk < 2; ++kk) {
                uint32_t out;
                if (rs >= last) {
                    out = SNB + kk;                                 // still skipping when the span ends
                } else {
                    const uint64_t n2 = rs >= end_const ? 0 : end_const - rs;
                    uint32_t pk = 0;
                    if (n2 <= 0xfffffff0ull) pk = lt_lookup(lt_off, lt_n0, lt_pk, kk == span.L ? 0u : 2 * S + kk, span.L, (uint32_t)n2);
                    if (pk & kPkAbsolute) {
                        out = pk & 0xffffu;
                    } else if (pk & kPkRelative) {
                        const uint32_t nbo = (pk >> 8) & 0xffffu;       // from a bit count of 0
                        out = (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                    } else {
                        PSim f;                                         // position dependent or sensitive
                        Acc a;
                        const bool alive = run_leaf(T, SNB + kk, span, rs, f, a);
                        out = encode_post(T, f, a, alive);
                    }
                }
                row[2 * S + kk] = pack_absolute(out, NB1);
                skip_pair |= out << (16u * kk);
            }
            row[2 * S + 2] = row[2 * S + 3] = pack_absolute(SNB + 2, NB1);
            // (kept for the entry walk: a leaf entered skipping is the one case it would otherwise have to redo --
            //  a lookup at the remaining length or a simulation; 4 bytes per leaf, one coalesced store per block)
            if (sp.skipc) sp.skipc[(size_t)gb * LB + l] = skip_pair;
        } else if (lane < nch * kChunk) {
            // padding of a short last chunk: identity rows (the fast walk always takes 16 steps)
            uint32_t *row = b.res + lane * nsim;
            for (uint32_t k = 0; k < S; ++k) row[2 * k] = row[2 * k + 1] = k | kPkRelative;
            row[2 * S] = pack_absolute(SNB, NB1);
            row[2 * S + 1] = pack_absolute(SNB + 1, NB1);
            row[2 * S + 2] = row[2 * S + 3] = pack_absolute(SNB + 2, NB1);
        }
        wave_sync_lds();
        STAMP(1);
        // ---- which leaves can end stuck (block_sims): those of the block from their rows, the few
        //      before it (whose stuck codes may enter the block) from the tables ------------------------
        if (T.NS) {
            for (uint32_t e = lane; e < count * T.nstuck_rows; e += 64) {
                const uint32_t l = e / T.nstuck_rows, rl = T.stuck_row[e - l * T.nstuck_rows];
                if ((rl >> 7) != ((uint32_t)((first + l) & 1ull) ^ T.lvl0)) continue;
                if (b.res[l * nsim + (rl & 127u)] & kPkStuck) atomicOr(&s_cap[(l + kStuckDepth) >> 5], 1u << ((l + kStuckDepth) & 31u));
            }
            for (uint32_t t = lane; t < T.depth * T.nstuck_rows; t += 64) {
                const uint32_t j = t / T.nstuck_rows + 1, rl = T.stuck_row[t - (j - 1) * T.nstuck_rows];
                const uint32_t r = rl & 127u;
                if (first < (uint64_t)j + 1) continue;                  // leaf first - j >= 1
                const uint64_t i = first - j;
                if ((rl >> 7) != ((uint32_t)(i & 1ull) ^ T.lvl0)) continue;
                const uint64_t n = edges[i] - edges[i - 1] - 1;
                if (n > 0xfffffff0ull) continue;
                if (lt_lookup(lt_off, lt_n0, lt_pk, r, (uint32_t)(i & 1ull) ^ T.lvl0, (uint32_t)n) & kPkStuck) {
                    atomicOr(&s_cap[(kStuckDepth - j) >> 5], 1u << ((kStuckDepth - j) & 31u));
                }
            }
            wave_sync_lds();
        }
        STAMP(2);
        // ---- chunk tables: every code a span can be entered in at this level, through its chunk ---------
        {
            const uint32_t lvl = (uint32_t)(first & 1ull) ^ T.lvl0;        // (a chunk starts on a multiple of 16 leaves)
            const uint16_t *lv = sp.reach ? sp.reach + sp.nreach + (lvl ? sp.nreach_lv[0] : 0u) : nullptr;
            const uint32_t NR = sp.reach ? sp.nreach_lv[lvl] : D0;
            const uint32_t nitem = nch * NR;
            uint32_t base = 0;
            while (base < nitem) {
                const uint32_t left = (nitem - base + 63u) / 64u;
                if (left >= 8) {
                    wave_compose<8>(T, sc, edges, first, b, s_resume, D, count, lv, NR, rcpNB1, base);
                    base += 8 * 64;
                } else if (left > 5) {
                    wave_compose<8>(T, sc, edges, first, b, s_resume, D, count, lv, NR, rcpNB1, base);
                    base = nitem;
                } else if (left > 3) {
                    wave_compose<5>(T, sc, edges, first, b, s_resume, D, count, lv, NR, rcpNB1, base);
                    base = nitem;
                } else if (left > 1) {
                    wave_compose<3>(T, sc, edges, first, b, s_resume, D, count, lv, NR, rcpNB1, base);
                    base = nitem;
                } else {
                    wave_compose<1>(T, sc, edges, first, b, s_resume, D, count, lv, NR, rcpNB1, base);
                    base = nitem;
                }
            }
            STAMP(3);
            // chains that START in a stuck code: only behind a leaf that can end stuck -- nowhere in a clean capture
            bool any = false;
            for (uint32_t w = 0; w < kCapWords; ++w) any = any || s_cap[w] != 0;
            if (any && T.NS) {
                const uint32_t nst = T.NS * T.depth;
                for (uint32_t item = lane; item < nch * nst; item += 64) {
                    const uint32_t c = item / nst, code = D0 + (item - c * nst);
                    const uint32_t la = c * kChunk, lbb = min((c + 1) * (uint32_t)kChunk, count);
                    // STUCK_d enters the chunk only if the leaf d before it can end stuck
                    const uint32_t ob = la + kStuckDepth - ((code - D0) / T.NS + 1);
                    if (!(s_cap[ob >> 5] & (1u << (ob & 31u)))) continue;
                    b.ctab[c * D + code] = (uint16_t)wave_chain_slow(T, sc, edges, first, la, lbb, b.res, s_resume, code);
                }
            }
            wave_sync_lds();
        }
        STAMP(4);
        // ---- the block's table: every abstract state walks the chunk tables ------------------------------
        for (uint32_t d = lane; d < D; d += 64) {
            uint32_t s = d;
            for (uint32_t c = 0; c < nch; ++c) s = b.ctab[c * D + s];
            sp.block_tab[(size_t)gb * sp.Dp + d] = (uint16_t)s;
        }
        wave_sync_lds();
        STAMP(5);
    }
}

// Walk of the block tables from the true start state, three small kernels:
// supergroups of 64 block tables are composed in parallel (one workgroup each:
// four groups of 16, then the four group tables), one lane per capture walks
// the supergroup tables, then (scan_entry_kernel, phase 0) one lane per group
// walks the group tables in front of it in its supergroup and its 16 blocks.
// (Round 2 stopped at the groups: the walk's one lane had 720 tables to go
// through at 16 GiB, 128 staged at a time -- 47 us, every one of them on the
// chain's critical path.)
constexpr int kGroup = 16;              // block tables per group
constexpr int kSuper = 4;               // groups per supergroup
constexpr int kSuperBlocks = kGroup * kSuper;
constexpr int kGroupsThreads = 1024;

__device__ __forceinline__ void locate_group(const ScanParams &sp, uint32_t gg, uint32_t &cap, uint32_t &lg) {
    uint32_t lo = 0, hi = sp.f.num_captures;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sp.cap_group_off[mid] <= gg) lo = mid;
        else hi = mid;
    }
    cap = lo;
    lg = gg - sp.cap_group_off[lo];
}

__device__ __forceinline__ void locate_super(const ScanParams &sp, uint32_t ss, uint32_t &cap, uint32_t &ls) {
    uint32_t lo = 0, hi = sp.f.num_captures;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sp.cap_super_off[mid] <= ss) lo = mid;
        else hi = mid;
    }
    cap = lo;
    ls = ss - sp.cap_super_off[lo];
}

// stage the block tables [b0, b0+nb) into LDS
__device__ __forceinline__ void stage_group(const ScanParams &sp, uint32_t b0, uint32_t nb, uint16_t *stage) {
    const uint4 *src = reinterpret_cast<const uint4 *>(sp.block_tab + (size_t)b0 * sp.Dp);
    uint4 *dst = reinterpret_cast<uint4 *>(stage);
    for (uint32_t i = threadIdx.x; i < nb * (sp.Dp / 8); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

__host__ __device__ __forceinline__ size_t groups_lds_bytes(uint32_t Dp) { return (size_t)(kSuperBlocks + kSuper) * Dp * 2; }

__global__ __launch_bounds__(kGroupsThreads) void scan_groups_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    if (*sp.fallback || (sp.sync_try && !(*sp.sync_fail & 1u))) return;
    uint16_t *stage = reinterpret_cast<uint16_t *>(scan_smem);      // [kSuperBlocks][Dp] block tables
    uint16_t *gt = stage + (size_t)kSuperBlocks * sp.Dp;            // [kSuper][Dp] the group tables
    const uint32_t D = sp.D;            // the whole domain, stuck codes included
    const uint32_t total = sp.cap_super_off[sp.f.num_captures];
    for (uint32_t ss = blockIdx.x; ss < total; ss += gridDim.x) {
        uint32_t cap, ls;
        locate_super(sp, ss, cap, ls);
        const uint32_t b0 = sp.cap_block_off[cap] + ls * kSuperBlocks;
        const uint32_t nb = min((uint32_t)kSuperBlocks, sp.cap_block_off[cap + 1] - b0);
        const uint32_t g0 = sp.cap_group_off[cap] + ls * kSuper;
        const uint32_t ng = (nb + kGroup - 1) / kGroup;
        stage_group(sp, b0, nb, stage);
        for (uint32_t item = threadIdx.x; item < ng * D; item += blockDim.x) {
            const uint32_t j = item / D, d = item - j * D;
            const uint32_t q0 = j * kGroup, q1 = min(q0 + (uint32_t)kGroup, nb);
            uint32_t s = d;
            for (uint32_t q = q0; q < q1; ++q) s = stage[q * sp.Dp + s];
            gt[j * sp.Dp + d] = (uint16_t)s;
            sp.group_tab[(size_t)(g0 + j) * sp.Dp + d] = (uint16_t)s;
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < D; d += blockDim.x) {
            uint32_t s = d;
            for (uint32_t j = 0; j < ng; ++j) s = gt[j * sp.Dp + s];
            sp.super_tab[(size_t)ss * sp.Dp + d] = (uint16_t)s;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kScanThreads) void scan_walk_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t x;
    if (*sp.fallback || (sp.sync_try && !(*sp.sync_fail & 1u))) return;
    uint16_t *stage = reinterpret_cast<uint16_t *>(scan_smem);      // up to 128 supergroup tables at a time
    const uint32_t max_stage = 128;
    for (uint32_t cap = blockIdx.x; cap < sp.f.num_captures; cap += gridDim.x) {
        const uint32_t g0 = sp.cap_super_off[cap], g1 = sp.cap_super_off[cap + 1];
        if (threadIdx.x == 0) x = sp.cap_first[cap];        // from the leaf kernel
        __syncthreads();
        for (uint32_t gs = g0; gs < g1; gs += max_stage) {
            const uint32_t ng = min(max_stage, g1 - gs);
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(sp.super_tab + (size_t)gs * sp.Dp);
                uint4 *dst = reinterpret_cast<uint4 *>(stage);
                for (uint32_t i = threadIdx.x; i < ng * (sp.Dp / 8); i += blockDim.x) dst[i] = src[i];
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t s = x;
                for (uint32_t g = 0; g < ng; ++g) {
                    sp.super_in[gs + g] = (uint16_t)s;
                    s = stage[g * sp.Dp + s];
                }
                x = s;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) sp.cap_end[cap] = (uint16_t)x;     // state after the last regular leaf
        __syncthreads();
    }
}

// LDS copies of the span tables and the merged rows for scan_entry_kernel, at file scope: the out-of-line step
// function below reads them as the LDS arrays they are.  (Through a pointer that may also point to global
// memory every read of a table search is a FLAT load at twice the latency, and a step of the walk is a dozen
// dependent reads: 1.8 us per step, 115 us for the pass; `tools`-less experiment in DESIGN.md 4.6a.)
__shared__ uint32_t g_lt[kLtLdsWords];          // offsets [noff] | n0 [nint] | pk [nint]
__shared__ uint32_t g_mr[kMergedLdsWords];

__device__ __forceinline__ uint32_t lt_lookup_g(uint32_t noff, uint32_t nint, uint32_t row, uint32_t L, uint32_t n) {
    uint32_t lo = g_lt[2 * row + L], hi = g_lt[2 * row + L + 1];
    if (lo >= hi) return 0u;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g_lt[noff + mid] <= n) lo = mid;
        else hi = mid;
    }
    return g_lt[noff + nint + lo];
}

// leaf_step_fly with the span tables in g_lt (stuck codes and simulations: the generic functions, through sc)
__device__ __noinline__ uint32_t leaf_step_fly_g(const LTab &T, const StuckCtx &sc, uint32_t noff, uint32_t nint, uint64_t i,
                                                 uint64_t e_before, uint64_t e_at, uint32_t s) {
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1;
    if (s == SNB + 2) return s;                                 // poison stays
    if (s > SNB + 2) return stuck_step(T, sc, i, s);
    Span span;
    span.pos0 = e_before + 1;
    span.n = e_at - e_before - 1;
    span.L = (uint32_t)(i & 1ull) ^ T.lvl0;
    span.has_edge = true;
    span.prefix = 0;
    const uint64_t resume = next_buffer_start(T, e_before);
    if (s < SNB) {
        if (span.n <= 0xfffffff0ull) {
            const uint32_t cur = s / NB1, nb = s - cur * NB1;
            const uint32_t p0 = lt_lookup_g(noff, nint, 2 * cur, span.L, (uint32_t)span.n);
            uint32_t pk = p0 & ~kPkShared;
            if (!(p0 & kPkShared) && nb >= T.max_bits) pk = lt_lookup_g(noff, nint, 2 * cur + 1, span.L, (uint32_t)span.n);
            if (pk & kPkAbsolute) return pk & 0xffffu;
            if (pk & kPkRelative) {
                const uint32_t nbo = nb + ((pk >> 8) & 0xffffu);
                return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
            }
            if (pk & kPkStuck) return stuck_enter(T, sc, s);
        }
    } else {
        const uint32_t kk = s - SNB;
        const uint64_t end_const = span.pos0 + span.n, last = end_const + 1;
        if (resume >= last) return s;                            // still skipping when the span ends
        const uint64_t n2 = resume >= end_const ? 0 : end_const - resume;
        if (n2 <= 0xfffffff0ull) {
            const uint32_t pk = lt_lookup_g(noff, nint, kk == span.L ? 0u : 2 * S + kk, span.L, (uint32_t)n2);
            if (pk & kPkAbsolute) return pk & 0xffffu;
            if (pk & kPkRelative) {
                const uint32_t nbo = (pk >> 8) & 0xffffu;       // from a bit count of 0
                return (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
            }
        }
    }
    PSim f;
    Acc a;
    const bool alive = run_leaf(T, s, span, resume, f, a);
    return encode_post(T, f, a, alive);
}

// The entry code of every leaf, exactly, from what leaf / groups / walk left behind -- in two small
// lane-parallel passes instead of inside every emit workgroup (where thread 0 walked up to 15 block
// tables and 4 chunk tables and 4 threads 16 leaves each, behind the staging of all those tables:
// half of that kernel's time).
//   phase 0: one lane per GROUP walks its <= 16 block tables from the group's entry state (scan_walk)
//            -> entry code of every block;
//   phase 1: one lane per BLOCK walks its leaves from the block's entry code, every step one search of
//            the span tables (LDS) for the leaf's length -- the same packed row the leaf kernel built the
//            block's tables from -> entry code of every leaf.  (First form: one lane per 16-leaf chunk
//            from rows and chunk tables the leaf kernel stored for it -- 123 MB of stores per 16 GiB
//            capture, and stores beside another context's front end cost it several times their share
//            of the bandwidth: profiles/r02_interfere.txt.)
constexpr uint32_t kEntryP0Blocks = 64;
__global__ __launch_bounds__(256) void scan_entry_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);
    if (sp.sync_try && !(*sp.sync_fail & 1u)) return;          // the entry codes are there already (scan_syncwalk_kernel)
    __shared__ LTab T;
    copy_ltab(T, sp.ltab);
    __syncthreads();
    if (threadIdx.x == 0) T.lvl0 = sp.has_prev ? fsm_level_at(sp.f, 0, -1) : 0u;
    __syncthreads();
    if (*sp.fallback) return;
    const uint32_t LB = sp.leaf_block;
    // Phases 0 and 2 do not depend on each other: one launch, the first kEntryP0Blocks workgroups take phase 0,
    // the others phase 2 (one launch boundary and one table staging less on the chain's critical path).
    uint32_t phase = sp.entry_phase, bid = blockIdx.x, nblk = gridDim.x;
    if (phase == 0 && gridDim.x > kEntryP0Blocks) {
        if (bid >= kEntryP0Blocks) {
            phase = 2;
            bid -= kEntryP0Blocks;
            nblk -= kEntryP0Blocks;
        } else {
            nblk = kEntryP0Blocks;
        }
    }
    if (phase == 0) {
        const uint32_t ngroups = sp.cap_group_off[sp.f.num_captures];
        for (uint32_t gg = bid * blockDim.x + threadIdx.x; gg < ngroups; gg += nblk * blockDim.x) {
            uint32_t cap, lg;
            locate_group(sp, gg, cap, lg);
            const uint32_t b0 = sp.cap_block_off[cap] + lg * kGroup;
            const uint32_t nb = min((uint32_t)kGroup, sp.cap_block_off[cap + 1] - b0);
            // the group's entry state: its supergroup's, through the group tables in front of it
            uint32_t s = sp.super_in[sp.cap_super_off[cap] + lg / kSuper];
            for (uint32_t j = gg - (lg % kSuper); j < gg; ++j) s = sp.group_tab[(size_t)j * sp.Dp + s];
            for (uint32_t j = 0; j < nb; ++j) {
                sp.blk_in[b0 + j] = (uint16_t)s;
                s = sp.block_tab[(size_t)(b0 + j) * sp.Dp + s];
            }
        }
        return;
    }
    // ---- phases 2 and 1 -----------------------------------------------------------------------------
    // Tables in LDS (g_lt, g_mr: file scope, see above) when both fit -- the fast form; otherwise the walk
    // calls leaf_step_fly on the tables where they are, one search or two per step.
    const uint32_t noff = 2 * (2 * T.S + 2) + 1;
    const bool fast = sp.lt_off && sp.lt_merged && sp.rowz && sp.lt_words <= kLtLdsWords && sp.lt_merged_words <= kMergedLdsWords;
    const uint32_t nint = fast ? (sp.lt_words - noff) / 2u : 0u;
    if (fast) {
        for (uint32_t i = threadIdx.x; i < noff; i += blockDim.x) g_lt[i] = sp.lt_off[i];
        for (uint32_t i = threadIdx.x; i < nint; i += blockDim.x) {
            g_lt[noff + i] = sp.lt_n0[i];
            g_lt[noff + nint + i] = sp.lt_pk[i];
        }
        for (uint32_t i = threadIdx.x; i < sp.lt_merged_words; i += blockDim.x) g_mr[i] = sp.lt_merged[i];
        __syncthreads();
    }
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    const uint32_t SNB = T.S * T.NB1;
    const uint32_t rcpNB1 = (uint32_t)((0x100000000ull + T.NB1 - 1) / T.NB1);
    if (phase == 2) {
        // one lane per LEAF: which interval of its level's merged breakpoints the leaf's length falls into -- the
        // only search a leaf needs, done for all of them at once instead of inside the sequential walk of phase 1
        if (!fast) return;
        for (uint64_t t = (uint64_t)bid * blockDim.x + threadIdx.x; t < (uint64_t)total * LB; t += (uint64_t)nblk * blockDim.x) {
            const uint32_t gb = (uint32_t)(t / LB), l = (uint32_t)(t - (uint64_t)gb * LB);
            uint32_t cap, lb;
            locate_block(sp, gb, cap, lb);
            uint64_t e0;
            const uint64_t ne = cap_edges(sp.f, cap, e0);
            const uint64_t first = 1 + (uint64_t)lb * LB;
            const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
            if (l >= count) continue;
            const uint64_t *edges = sp.f.edges + e0;
            const uint64_t i = first + l;
            const uint64_t n = edges[i] - edges[i - 1] - 1;
            const uint32_t L = (uint32_t)(i & 1ull) ^ T.lvl0;
            const uint32_t b0 = 4 + (L ? g_mr[0] : 0u);
            uint32_t lo = 0, hi = g_mr[L];
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (g_mr[b0 + mid] <= (uint32_t)n) lo = mid;
                else hi = mid;
            }
            sp.rowz[t] = n <= 0xfffffff0ull ? (uint16_t)lo : (uint16_t)0xffffu;
        }
        return;
    }
    // One lane in eight walks a block: a step that needs the full function (the first leaf after a skipped rest of
    // a buffer, stuck codes: under one per cent of them, but whole blocks of them after a glitch) is paid by the
    // whole wave; the walks are latency-bound chains, so more, emptier waves cost nothing.
    if (fast && (threadIdx.x & 7u)) return;
    const uint32_t per = fast ? 8u : 1u;
    for (uint32_t gb = (blockIdx.x * blockDim.x + threadIdx.x) / per; gb < total; gb += (gridDim.x * blockDim.x) / per) {
        uint32_t cap, lb;
        locate_block(sp, gb, cap, lb);
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, cap, e0);
        const uint64_t first = 1 + (uint64_t)lb * LB;
        const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
        const uint64_t *edges = sp.f.edges + e0;
        // (fast form: stuck codes search the LDS copy of the span tables too -- a stuck step through the global
        //  ones is seven dependent trips to memory that the whole wave waits for)
        const StuckCtx sc = fast ? StuckCtx{edges, g_lt, g_lt + noff, g_lt + noff + nint} : StuckCtx{edges, sp.lt_off, sp.lt_n0, sp.lt_pk};
        uint32_t s = sp.blk_in[gb];
        uint4 *pre = reinterpret_cast<uint4 *>(sp.pre_codes + (size_t)gb * LB);
        if (fast && LB == 64) {
            // Fast form.  ONE round trip to memory for the whole block: the 64 leaves' row intervals (phase 2) and
            // what the leaf kernel found for the two skip codes are requested together, before the first step; a
            // step is then one LDS read of the row the state selects plus its decode.  The edges themselves are
            // only read by a step that needs the full function (under one per cent of them).  (First form: eight
            // leaves' edges, intervals and skip results per eight steps -- eight dependent round trips per block,
            // 45 us for the pass at 16 GiB.)
            const uint4 *z4p = reinterpret_cast<const uint4 *>(sp.rowz + (size_t)gb * LB);
            const uint4 *sk4p = reinterpret_cast<const uint4 *>(sp.skipc + (size_t)gb * LB);
            uint4 zq[8], skq[16];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) zq[q] = z4p[q];
            if (sp.skipc_valid) {
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) skq[q] = sk4p[q];
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 16; ++q) skq[q] = make_uint4(0, 0, 0, 0);
            }
            const uint32_t rows0 = 4 + g_mr[0] + g_mr[1], nbp0 = g_mr[0], twoS = 2u * T.S;
#pragma unroll
            for (uint32_t g = 0; g < 8; ++g) {
                if (8 * g >= count) break;
                const uint32_t zz[4] = {zq[g].x, zq[g].y, zq[g].z, zq[g].w};
                const uint32_t sk[8] = {skq[2 * g].x, skq[2 * g].y, skq[2 * g].z, skq[2 * g].w,
                                        skq[2 * g + 1].x, skq[2 * g + 1].y, skq[2 * g + 1].z, skq[2 * g + 1].w};
                uint32_t code[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; ++k) {
                    code[k] = s;
                    const uint32_t l = 8 * g + k;
                    if (l < count) {
                        const uint32_t z = (zz[k >> 1] >> (16u * (k & 1u))) & 0xffffu;
                        bool done = false;
                        if (s < SNB && z != 0xffffu) {
                            const uint32_t L = (uint32_t)((first + l) & 1ull) ^ T.lvl0;
                            const uint32_t roff = rows0 + ((L ? nbp0 : 0u) + z) * twoS;
                            const uint32_t cur = __umulhi(s, rcpNB1), nb = s - cur * T.NB1;
                            const uint32_t p = g_mr[roff + 2u * cur + (nb >= T.max_bits ? 1u : 0u)];
                            if (p & kPkAbsolute) {
                                s = p & 0xffffu;
                                done = true;
                            } else if (p & kPkRelative) {
                                const uint32_t nbo = nb + ((p >> 8) & 0xffffu);
                                s = (p & 0xffu) * T.NB1 + (nbo >= T.NB1 ? T.NB1 - 1 : nbo);
                                done = true;
                            }
                        } else if (s >= SNB && s < SNB + 2 && sp.skipc_valid) {
                            s = (sk[k] >> (16u * (s - SNB))) & 0xffffu;     // what the leaf kernel found for this leaf
                            done = true;
                        }
                        if (!done) {
                            const uint64_t e_before = edges[first + l - 1], e_at = edges[first + l];
                            if (s >= SNB && s < SNB + 2 && next_buffer_start(T, e_before) > e_at) {
                                // still skipping when the span ends
                            } else {
                                s = leaf_step_fly_g(T, sc, noff, nint, first + l, e_before, e_at, s);
                            }
                        }
                    }
                }
                pre[g] = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16), code[4] | (code[5] << 16),
                                    code[6] | (code[7] << 16));
            }
            continue;
        }
        // General form (no tables in LDS, or blocks of another size): eight leaves at a time, their edges in one go
        // (one memory latency per eight steps, not per step), their codes in one 16-byte store (block-major list:
        // 2 * LB bytes per block, aligned).
        uint64_t before = edges[first - 1];
        for (uint32_t l0 = 0; l0 < count; l0 += 8) {
            uint64_t ev[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) ev[k] = edges[first + min(l0 + k, count - 1)];
            uint32_t code[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) {
                code[k] = s;
                if (l0 + k < count) {
                    s = fast ? leaf_step_fly_g(T, sc, noff, nint, first + l0 + k, before, ev[k], s)
                             : leaf_step_fly(T, sc, first + l0 + k, before, ev[k], s);
                    before = ev[k];
                }
            }
            pre[l0 >> 3] = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16), code[4] | (code[5] << 16),
                                      code[6] | (code[7] << 16));
        }
    }
}

// ---------------------------------------------------------------------------
// entry codes from SYNCHRONISING spans (round 3)
// ---------------------------------------------------------------------------
// The composing kernels above (leaf -> groups -> walk -> entry) find the state every leaf is entered in by composing
// functions on the whole abstract domain (116 reachable codes for p3l-nexa2012) over every leaf: 223 us of the chain
// at 16 GiB, two thirds of it the block tables.  But an OOK capture is messages with silence between them, and a
// long span leaves very little of the state it was entered in: every state times out into reset, a skipped rest-of-
// buffer has ended, and the edge that ends the span finds the machine idle -- or, entered in the one state where
// that edge is an error, dropping the next buffer.  The IMAGE of such a leaf -- the codes it can end in, over every
// code it can be entered in -- is two or three codes.  So:
//   scan_sync_kernel      one lane per leaf (a wave = a block): the leaf's merged-rows interval (phase 2 of
//                         scan_entry_kernel), its two skip rows (as the leaf kernel computes them), and its image:
//                         the interval's (host: append_sync_codes) plus the two skip rows' results.  A leaf with at
//                         most kSyncK codes in its image, and none of the T.depth leaves before it able to end stuck
//                         (no stuck code can enter it), is a SYNC leaf; the first of a block goes into the block's
//                         record with its candidates.
//   scan_syncwalk_kernel  one lane per (block with a sync leaf, candidate) -- and one for the first block of every
//                         capture, from the capture's first span, a concrete simulation --: from behind the block's
//                         sync leaf in that candidate through the block and on, until it has stepped through the
//                         sync leaf of a later block: which of THAT leaf's candidates it arrives at goes into the
//                         record.  The entry codes it passes go into the candidate's plane of pre_codes.  A step
//                         is what a step of scan_entry_kernel's phase 1 is.
//   scan_syncpick_kernel  one workgroup per capture: the records are maps on kSyncK candidates -- composed in a
//                         scan (a thread per run of blocks, then 1024 maps by doubling) from the first block's one
//                         candidate: the candidate every region was really entered in, i.e. which plane holds a
//                         leaf's entry code (scan_emit_kernel reads that plane), and the state behind the last leaf.
// Nothing is assumed: the image covers every code the leaf can be entered in (normal codes from the tables' closure,
// the two skip codes evaluated for the leaf itself, stuck codes excluded by the leaves before).  What can happen is
// that there is nothing to hold on to -- kSyncMaxRun blocks without a sync leaf (dense noise: the composing scan's
// depth is logarithmic in such a stretch, a walk's linear) -- or a walk ends in a code the image does not list (a
// stuck code met on the way): then sync_fail is set and the composing kernels, queued behind and otherwise returning
// at once, run as if this had not been tried.
constexpr uint32_t kSyncK = 4;                  // candidates per sync leaf (planes of pre_codes)
constexpr uint32_t kSyncMaxRun = 8;             // blocks a walk runs on beyond its own
constexpr uint32_t kSyncNone = 0xffffu;
// a block's record (32-bit words)
constexpr uint32_t kSyncRecWords = 8;
constexpr uint32_t kSyncRecInfo = 0;            // split | candidates << 8 | 0x8000: leaves >= split belong to the region
                                                // that starts in this block (0: no sync leaf, nothing starts here)
constexpr uint32_t kSyncRecCand = 2;            // [2] the candidates, 16 bits each
                                                // (words 1, 4, 5: spare)
constexpr uint32_t kSyncRecEnd = 6;             // [2] per candidate, 16 bits: the code behind the capture's last leaf

// a record's digest: split | 0x80 a region starts here | candidates << 8 | per candidate 3 bits from bit 11: index among
// the next region's candidates, | 4: its walk reached the end of the capture
constexpr uint32_t kSyncDigMap = 11;

// T, the span tables and the merged rows into LDS (every thread of the workgroup; ends with a barrier)
__device__ __forceinline__ void stage_sync_tables(const ScanParams &sp, LTab &T, uint32_t noff, uint32_t nint) {
    copy_ltab(T, sp.ltab);
    for (uint32_t i = threadIdx.x; i < noff; i += blockDim.x) g_lt[i] = sp.lt_off[i];
    for (uint32_t i = threadIdx.x; i < nint; i += blockDim.x) {
        g_lt[noff + i] = sp.lt_n0[i];
        g_lt[noff + nint + i] = sp.lt_pk[i];
    }
    for (uint32_t i = threadIdx.x; i < sp.lt_sync_words; i += blockDim.x) g_mr[i] = sp.lt_merged[i];
    __syncthreads();
    if (threadIdx.x == 0) T.lvl0 = sp.has_prev ? fsm_level_at(sp.f, 0, -1) : 0u;
    __syncthreads();
}

// interval of the level's merged breakpoints a length falls into (g_mr)
__device__ __forceinline__ uint32_t merged_interval(uint32_t L, uint64_t n) {
    const uint32_t b0 = 4 + (L ? g_mr[0] : 0u);
    uint32_t lo = 0, hi = g_mr[L];
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g_mr[b0 + mid] <= (uint32_t)n) lo = mid;
        else hi = mid;
    }
    return n <= 0xfffffff0ull ? lo : kSyncNone;
}

__global__ __launch_bounds__(256) void scan_sync_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ LTab T;
    const uint32_t noff = 2 * (2 * sp.S + 2) + 1, nint = (sp.lt_words - noff) / 2u;
    stage_sync_tables(sp, T, noff, nint);
    if (*sp.fallback) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(sp.sync_fail, 2u);        // bit 1: tried (bit 0: gave up)
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1, depth = T.NS ? T.depth : 0u;
    const uint32_t LB = sp.leaf_block;          // 64: a wave is a block
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t nbp0 = g_mr[0], sync_off = g_mr[3];
    // (one capture -- the usual run: its extent once, not a dependent round trip per block)
    const bool one = sp.f.num_captures == 1;
    uint64_t e0_one = 0;
    const uint64_t ne_one = one ? cap_edges(sp.f, 0, e0_one) : 0;
    for (uint32_t gb = blockIdx.x * 4u + wave; gb < total + sp.f.num_captures; gb += gridDim.x * 4u) {
        if (gb >= total) {
            // one extra item per capture: its first span from the concrete incoming state
            if (lane == 0) {
                const uint32_t cap = gb - total;
                uint64_t e0;
                const uint64_t ne = cap_edges(sp.f, cap, e0);
                PSim f;
                Acc a;
                const bool alive = first_leaf(T, sp, sp.f.edges + e0, ne, f, a);
                sp.cap_first[cap] = (uint16_t)encode_post(T, f, a, alive);
            }
            continue;
        }
        uint32_t cap = 0, lb = gb;
        uint64_t e0 = e0_one, ne = ne_one;
        if (!one) {
            locate_block(sp, gb, cap, lb);
            ne = cap_edges(sp.f, cap, e0);
        }
        const uint64_t *edges = sp.f.edges + e0;
        const uint64_t first = 1 + (uint64_t)lb * LB;
        const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
        bool few = false, stuckable = false;
        uint32_t cand[kSyncK] = {kSyncNone, kSyncNone, kSyncNone, kSyncNone}, nc = 0;
        if (lane < count) {
            const uint64_t i = first + lane;
            const Span span = span_of(T, edges, i);
            const uint32_t z = merged_interval(span.L, span.n);
            // (for the walk: where the leaf's row set starts in the rows, in words -- not the interval, which is what
            //  scan_entry_kernel's phase 2 leaves in the same list when the composing kernels run)
            sp.rowz[(size_t)gb * LB + lane] = z != kSyncNone ? (uint16_t)(((span.L ? nbp0 : 0u) + z) * 2u * S) : (uint16_t)kSyncNone;
            // the two skip rows (scan_leaf_wave_kernel's): skipping ends at `rs`; from there the machine starts in
            // reset -- the normal row (reset, few bits) of a shorter span when the level before the skip equals
            // the span's, a special row otherwise
            const uint64_t rs = next_buffer_start(T, span.pos0 - 1);
            const uint64_t end_const = span.pos0 + span.n, last = end_const + 1;
            uint32_t skip_out[2];
            for (uint32_t kk = 0; kk < 2; ++kk) {
                uint32_t out;
                if (rs >= last) {
                    out = SNB + kk;                                 // still skipping when the span ends
                } else {
                    const uint64_t n2 = rs >= end_const ? 0 : end_const - rs;
                    uint32_t pk = 0;
                    if (n2 <= 0xfffffff0ull) pk = lt_lookup_g(noff, nint, kk == span.L ? 0u : 2 * S + kk, span.L, (uint32_t)n2);
                    if (pk & kPkAbsolute) {
                        out = pk & 0xffffu;
                    } else if (pk & kPkRelative) {
                        const uint32_t nbo = (pk >> 8) & 0xffffu;       // from a bit count of 0
                        out = (pk & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                    } else {
                        PSim f;                                         // position dependent or sensitive
                        Acc a;
                        const bool alive = run_leaf(T, SNB + kk, span, rs, f, a);
                        out = encode_post(T, f, a, alive);
                    }
                }
                skip_out[kk] = out;
            }
            sp.skipc[(size_t)gb * LB + lane] = skip_out[0] | (skip_out[1] << 16);
            // the image: the interval's codes and the two skip results
            if (z != kSyncNone && sync_off) {
                const uint32_t w0 = g_mr[sync_off + 2u * ((span.L ? nbp0 : 0u) + z)];
                const uint32_t w1 = g_mr[sync_off + 2u * ((span.L ? nbp0 : 0u) + z) + 1u];
                const uint32_t n = (w1 >> 16) & 0xfu;
                stuckable = ((w1 >> 20) & 1u) != 0;
                if (n) {
                    cand[0] = w0 & 0xffffu;
                    cand[1] = w0 >> 16;
                    cand[2] = w1 & 0xffffu;
                    nc = n;
                    few = true;
#pragma unroll
                    for (uint32_t kk = 0; kk < 2; ++kk) {
                        const uint32_t c = skip_out[kk];
                        if (c == cand[0] || c == cand[1] || c == cand[2] || c == cand[3]) continue;
                        if (nc == kSyncK || c >= SNB + 2) {
                            few = false;
                        } else {
                            // (cand[nc] = c with nc in registers)
                            cand[3] = nc == 3 ? c : cand[3];
                            cand[2] = nc == 2 ? c : cand[2];
                            cand[1] = nc == 1 ? c : cand[1];
                            ++nc;
                        }
                    }
                }
            } else {
                stuckable = depth != 0;         // (a span beyond the tables' lengths: say it can)
            }
        }
        // leaves that a stuck code can enter: one of the `depth` leaves before them can end stuck -- those of the
        // block by ballot, those in front of it looked up by the last `depth` lanes
        uint64_t dirty = 0;
        if (depth) {
            const uint64_t sw = __ballot(stuckable);
            bool pre = false;
            if (lane >= 64u - depth) {
                const uint64_t back = 64u - lane;           // 1 .. depth leaves in front of the block
                if (first > back) {                         // leaf first - back >= 1
                    const uint64_t i = first - back;
                    const uint64_t n = edges[i] - edges[i - 1] - 1;
                    const uint32_t L = (uint32_t)(i & 1ull) ^ T.lvl0;
                    const uint32_t z = merged_interval(L, n);
                    pre = z == kSyncNone || !sync_off || ((g_mr[sync_off + 2u * ((L ? nbp0 : 0u) + z) + 1u] >> 20) & 1u);
                }
            }
            const uint64_t pw = __ballot(pre);
            for (uint32_t d = 1; d <= depth; ++d) dirty |= (sw << d) | (pw >> (64u - d));
        }
        const uint64_t valid = __ballot(few) & ~dirty;
        uint32_t *rec = sp.sync_rec + (size_t)gb * kSyncRecWords;
        if (lb == 0) {
            // the first block of a capture: its region starts at its first leaf, in the code behind the first span
            if (lane == 0) rec[kSyncRecInfo] = 0u | (1u << 8) | 0x8000u;
        } else if (valid == 0) {
            if (lane == 0) {
                rec[kSyncRecInfo] = 0u;
                sp.sync_dig[gb] = 0u;           // (a block where a region starts gets its digest from the walk)
            }
        } else if (lane == (uint32_t)__builtin_ctzll(valid)) {
            rec[kSyncRecInfo] = (lane + 1u) | (nc << 8) | 0x8000u;
            rec[kSyncRecCand] = cand[0] | (cand[1] << 16);
            rec[kSyncRecCand + 1] = cand[2] | (cand[3] << 16);
        }
    }
}

// abstract code -> the walk's (state, bit count): see scan_syncwalk_kernel
__device__ __forceinline__ void sync_split(uint32_t code, uint32_t SNB, uint32_t NB1, uint32_t S, uint32_t rcpNB1,
                                           uint32_t &cur, uint32_t &nb) {
    if (code < SNB) {
        cur = __umulhi(code, rcpNB1);
        nb = code - __umul24(cur, NB1);
    } else if (code < SNB + 3u) {
        cur = S;
        nb = code - SNB;
    } else {
        cur = 0x7fu;
        nb = code;
    }
}

// The walk.  Eight lanes per block (a "slot"; eight slots per wave), the first kSyncK of them its candidates, all
// eight fetch for it.  A region is a run of consecutive leaves -- rowz / skipc / the planes of pre_codes are
// [block][leaf], blocks of a capture follow each other -- from behind the block's sync leaf through the sync leaf of
// the next block that has one (found up front: the records of the kSyncMaxRun blocks behind, one round trip).  The
// wave goes through its slots' regions 64 leaves at a time: the slot's lanes fetch the next 64 intervals and skip
// results of THEIR region into LDS (one round trip), then a ROLLED loop of 64 steps, every slot on its own leaves.
// (First form: every lane over the blocks it crosses, 64 unrolled steps per block with the leaf data in registers:
//  10 000 instructions, 158 per step with its predicates and an out-of-line call site each, and a wave paid for the
//  union of its eight blocks' ranges -- 1 200 cycles per useful step, 114 us at 16 GiB.)
__global__ __launch_bounds__(256) void scan_syncwalk_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ LTab T;
    __shared__ __attribute__((aligned(16))) uint16_t s_z[4][8][64];
    __shared__ uint32_t s_sk[4][8][64];
    const uint32_t noff = 2 * (2 * sp.S + 2) + 1, nint = (sp.lt_words - noff) / 2u;
    stage_sync_tables(sp, T, noff, nint);
    if (*sp.fallback) return;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, slot = lane >> 3, k = lane & 7u;
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1, max_bits = T.max_bits;
    const uint32_t rcpNB1 = (uint32_t)((0x100000000ull + NB1 - 1) / NB1);
    const uint32_t rows2 = g_mr[3] + 2u * (g_mr[0] + g_mr[1]);     // append_sync_codes' second copy of the rows
    uint16_t *const zq = s_z[wave][slot];
    uint32_t *const skq = s_sk[wave][slot];
    for (uint32_t base = (blockIdx.x * 4u + wave) * 8u; base < total; base += gridDim.x * 32u) {
        const uint32_t gb = base + slot;
        uint32_t *rec = sp.sync_rec + (size_t)min(gb, total - 1u) * kSyncRecWords;
        const uint32_t info = gb < total ? rec[kSyncRecInfo] : 0u;
        const bool region = (info & 0x8000u) != 0;              // (the same for the slot's eight lanes)
        const bool walker = region && k < ((info >> 8) & 7u);
        // ---- the region: leaves [f0, f0 + len) of the capture's [block][leaf] lists ------------------------
        uint32_t cap = 0, lb = 0, gcap0 = 0, nblk = 0, len = 0, g2 = 0, info2 = 0;
        uint64_t ne = 0;
        const uint64_t *edges = sp.f.edges;
        bool stop = false, bad = false;
        if (region) {
            locate_block(sp, gb, cap, lb);
            uint64_t e0;
            ne = cap_edges(sp.f, cap, e0);
            edges = sp.f.edges + e0;
            gcap0 = gb - lb;
            nblk = sp.cap_block_off[cap + 1] - gcap0;
            // the next block with a sync leaf: the records behind, all asked for at once
            uint32_t inf[kSyncMaxRun];
#pragma unroll
            for (uint32_t j = 0; j < kSyncMaxRun; ++j)
                inf[j] = lb + 1 + j < nblk ? sp.sync_rec[(size_t)(gb + 1 + j) * kSyncRecWords + kSyncRecInfo] : 0u;
            uint32_t dist = 0;
#pragma unroll
            for (uint32_t j = kSyncMaxRun; j-- > 0;) {
                if (inf[j] & 0x8000u) {
                    dist = j + 1;
                    info2 = inf[j];
                }
            }
            const uint32_t f0 = lb * 64u + (info & 0x7fu);
            uint32_t fend;                                      // one past the region's last leaf
            if (dist) {
                stop = true;
                g2 = gb + dist;
                fend = (lb + dist) * 64u + (info2 & 0x7fu);     // through that block's sync leaf (split - 1)
            } else if (lb + 1 + kSyncMaxRun >= nblk) {
                fend = (uint32_t)(ne - 1);                      // to the capture's last regular leaf (leaf i at i - 1)
            } else {
                bad = true;                                     // nothing to hold on to
                fend = f0;
            }
            len = fend > f0 ? fend - f0 : 0u;
        }
        const uint32_t f0 = lb * 64u + (info & 0x7fu);
        const size_t flat0 = (size_t)gcap0 * 64u + f0;          // where the region starts in rowz / skipc / a plane
        uint16_t *plane = sp.pre_codes + (size_t)min(k, kSyncK - 1u) * sp.pre_plane + flat0;
        const StuckCtx sc{edges, g_lt, g_lt + noff, g_lt + noff + nint};
        // the walk's state: machine state / bit count; skip and poison as state S with bit count 0 / 1 / 2 (the rows
        // are made of that: no division per step); a stuck code as state 0x7f with the code for a bit count
        uint32_t cur = 0, nb = 0;
        if (walker) {
            const uint32_t s0 = lb == 0 ? (uint32_t)sp.cap_first[cap] : (rec[kSyncRecCand + (k >> 1)] >> (16u * (k & 1u))) & 0xffffu;
            sync_split(s0, SNB, NB1, S, rcpNB1, cur, nb);
        }
        // ---- 64 leaves at a time -----------------------------------------------------------------------------
        for (uint32_t c0 = 0; __ballot(c0 < len) != 0; c0 += 64) {
            if (c0 < len) {
                // the slot's eight lanes fetch its next 64 leaves (beyond the region's end: whatever is there --
                // the lists are padded --, never read)
                uint32_t zv[8], sv[8];
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    zv[j] = sp.rowz[flat0 + c0 + 8u * k + j];
                    sv[j] = sp.skipc[flat0 + c0 + 8u * k + j];
                }
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    zq[8u * k + j] = (uint16_t)zv[j];
                    skq[8u * k + j] = sv[j];
                }
            }
            wave_sync_lds();
            const uint32_t left = len > c0 ? min(len - c0, 64u) : 0u;
            // eight steps at a time: their row offsets in ONE 16-byte read, off the chain of dependent reads (a step
            // is then one read of the row entry the state selects, and its decode)
            for (uint32_t j0 = 0; __ballot(walker && j0 < left) != 0; j0 += 8) {
                const uint4 r8 = *reinterpret_cast<const uint4 *>(zq + j0);
                const uint32_t rr[4] = {r8.x, r8.y, r8.z, r8.w};
#pragma unroll
                for (uint32_t jj = 0; jj < 8; ++jj) {
                    const uint32_t j = j0 + jj;
                    if (walker && j < left) {
                        const uint32_t rz = (rr[jj >> 1] >> (16u * (jj & 1u))) & 0xffffu;   // where the leaf's row set starts
                        plane[c0 + j] = (uint16_t)(cur | (nb << 7));
                        const bool plain = cur < S && rz != kSyncNone;
                        uint32_t q = g_mr[rows2 + (plain ? rz : 0u) + 2u * min(cur, S - 1u) + (nb >= max_bits ? 1u : 0u)];
                        q = plain ? q : 0u;
                        if ((int32_t)q < 0) {
                            // state' | bit count (absolute) or bits appended (relative) << 8
                            nb = min(((q & 0x40000000u) ? nb : 0u) + ((q >> 8) & 0xffffu), NB1 - 1u);
                            cur = q & 0xffu;
                        } else {
                            // skip codes: scan_sync_kernel's skip rows; stuck codes, rows that need a simulation: the full step
                            uint32_t code = cur < S ? __umul24(cur, NB1) + nb : (cur == S ? SNB + nb : nb);
                            if (cur == S && nb < 2u) {
                                code = (skq[j] >> (16u * nb)) & 0xffffu;
                            } else {
                                const uint32_t f = f0 + c0 + j;         // leaf i = f + 1 of the capture
                                code = leaf_step_fly_g(T, sc, noff, nint, (uint64_t)f + 1u, edges[f], edges[f + 1u], code);
                            }
                            sync_split(code, SNB, NB1, S, rcpNB1, cur, nb);
                        }
                    }
                }
            }
            wave_sync_lds();
        }
        const uint32_t s = cur < S ? __umul24(cur, NB1) + nb : (cur == S ? SNB + nb : nb);
        // ---- where the walk arrived ----------------------------------------------------------------------------
        uint32_t mine = 0;
        if (walker) {
            uint32_t arrive = 0;
            if (stop) {
                // which of that leaf's candidates it is
                const uint32_t c01 = sp.sync_rec[(size_t)g2 * kSyncRecWords + kSyncRecCand];
                const uint32_t c23 = sp.sync_rec[(size_t)g2 * kSyncRecWords + kSyncRecCand + 1];
                const uint32_t nc2 = (info2 >> 8) & 7u;
                const uint32_t cc[kSyncK] = {c01 & 0xffffu, c01 >> 16, c23 & 0xffffu, c23 >> 16};
                arrive = kSyncK;
#pragma unroll
                for (uint32_t j = 0; j < kSyncK; ++j)
                    if (j < nc2 && cc[j] == s && arrive == kSyncK) arrive = j;
                if (arrive == kSyncK) {         // not in the image: a stuck code (or poison) met on the way
                    bad = true;
                    arrive = 0;
                    if (sp.f.debug && atomicAdd(reinterpret_cast<unsigned long long *>(sp.f.debug + 56), 1ull) == 0) {
                        sp.f.debug[57] = gb | ((uint64_t)k << 32);
                        sp.f.debug[58] = s | ((uint64_t)len << 32);
                        sp.f.debug[59] = c01 | ((uint64_t)c23 << 32);
                        sp.f.debug[60] = info | ((uint64_t)info2 << 32);
                        sp.f.debug[61] = g2 | ((uint64_t)f0 << 32);
                        sp.f.debug[62] = rec[kSyncRecCand] | ((uint64_t)rec[kSyncRecCand + 1] << 32);
                    }
                }
            }
            if (bad && !stop && sp.f.debug) sp.f.debug[63] = gb | ((uint64_t)lb << 32);
            // (16-bit stores: the candidates of a block write the halves of the same words)
            reinterpret_cast<uint16_t *>(rec + kSyncRecEnd)[k] = (uint16_t)s;
            if (bad) atomicOr(sp.sync_fail, 1u);
            mine = (arrive | (stop ? 0u : 4u)) << (3u * k);
        }
        // the slot's digest: its candidates' three bits each, gathered over its eight lanes
        mine |= (uint32_t)__shfl_xor((int)mine, 1);
        mine |= (uint32_t)__shfl_xor((int)mine, 2);
        mine |= (uint32_t)__shfl_xor((int)mine, 4);
        if (region && k == 0) sp.sync_dig[gb] = (info & 0x7fu) | 0x80u | (info & 0x700u) | (mine << kSyncDigMap);
    }
}

// maps on kSyncK candidates, 8 bits per entry
__device__ __forceinline__ uint32_t sync_map_identity() { return 0x03020100u; }
__device__ __forceinline__ uint32_t sync_map_apply(uint32_t m, uint32_t x) { return (m >> (8u * x)) & 0xffu; }
// first a, then b
__device__ __forceinline__ uint32_t sync_map_then(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t x = 0; x < kSyncK; ++x) r |= sync_map_apply(b, sync_map_apply(a, x) & 3u) << (8u * x);
    return r;
}
// blocks per thread of scan_syncpick_kernel, at most (1024 threads: captures up to 2.1 M edges -- a 32 GiB shard of
// the bench capture has 1.5 M; beyond, the composing kernels).  The digests of a capture's blocks go through the LDS: fetched and stored in order, coalesced
// (a thread reading its own run of 32-byte records from memory touched a cache line per lane and word: 73 000 line
// requests from one CU, 32 us), read and rewritten there by the thread that owns the run.
constexpr uint32_t kPickPer = 32;

__device__ __forceinline__ uint32_t sync_dig_map(uint32_t dig) {
    const uint32_t mb = dig >> kSyncDigMap;
    return (mb & 3u) | (((mb >> 3) & 3u) << 8) | (((mb >> 6) & 3u) << 16) | (((mb >> 9) & 3u) << 24);
}

__global__ __launch_bounds__(kScanThreads) void scan_syncpick_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint32_t maps[2][kScanThreads / 64];
    __shared__ uint32_t lastc[kScanThreads];
    uint32_t *const dig = reinterpret_cast<uint32_t *>(scan_smem);          // [kPickPer * kScanThreads]
    const uint32_t tid = threadIdx.x;
    if (*sp.fallback) return;
    if (*sp.sync_fail & 1u) {
        // no composing kernels behind this launch: the run is refused, the host queues it again with them
        if (sp.sync_only && tid == 0) atomicOr(sp.fallback, (uint32_t)kFbSync);
        return;
    }
    for (uint32_t cap = blockIdx.x; cap < sp.f.num_captures; cap += gridDim.x) {
        const uint32_t gb0 = sp.cap_block_off[cap], nblk = sp.cap_block_off[cap + 1] - gb0;
        const uint32_t per = (nblk + kScanThreads - 1) / kScanThreads;
        if (per > kPickPer) {                   // (uniform: every thread leaves)
            if (tid == 0) {
                atomicOr(sp.sync_fail, 1u);
                if (sp.sync_only) atomicOr(sp.fallback, (uint32_t)kFbSync);
            }
            return;
        }
        for (uint32_t i = tid; i < nblk; i += kScanThreads) dig[i] = sp.sync_dig[gb0 + i];
        __syncthreads();
        const uint32_t b0 = min(tid * per, nblk), b1 = min(b0 + per, nblk);
        // this thread's run of blocks as ONE map: candidate at the first region that starts at or behind b0 ->
        // candidate at the first region that starts behind the run (no region starts in the run: the same one)
        uint32_t m = sync_map_identity();
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t d = dig[b];
            if (d & 0x80u) m = sync_map_then(m, sync_dig_map(d));
        }
        // exclusive scan of the maps: inside the waves by shuffles, the sixteen wave totals by the first wave
        const uint32_t lane = tid & 63u, wv = tid >> 6;
        uint32_t incl = m;
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t other = (uint32_t)__shfl_up((int)incl, (int)d);
            if (lane >= d) incl = sync_map_then(other, incl);
        }
        if (lane == 63) maps[0][wv] = incl;
        __syncthreads();
        if (wv == 0) {
            uint32_t t = lane < (uint32_t)(kScanThreads / 64) ? maps[0][lane] : sync_map_identity();
#pragma unroll
            for (uint32_t d = 1; d < (uint32_t)(kScanThreads / 64); d <<= 1) {
                const uint32_t other = (uint32_t)__shfl_up((int)t, (int)d);
                if (lane >= d) t = sync_map_then(other, t);
            }
            if (lane < (uint32_t)(kScanThreads / 64)) maps[1][lane] = t;        // inclusive over the waves
        }
        __syncthreads();
        uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) excl = sync_map_identity();
        if (wv) excl = sync_map_then(maps[1][wv - 1], excl);
        // the capture's first region has one candidate (index 0): what it has become in front of this thread's run
        uint32_t c = sync_map_apply(excl, 0u) & 3u;
        // the run again: every region's true candidate, left in its digest's place (| 0x80)
        uint32_t last = 0xffffffffu;
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t d = dig[b];
            if (d & 0x80u) {
                last = c;
                const uint32_t me = (d >> (kSyncDigMap + 3u * c)) & 7u;
                if (me & 4u) {                  // its walk reached the end of the capture: the state behind the last leaf
                    const uint32_t e = sp.sync_rec[(size_t)(gb0 + b) * kSyncRecWords + kSyncRecEnd + (c >> 1)];
                    sp.cap_end[cap] = (uint16_t)((e >> (16u * (c & 1u))) & 0xffffu);
                }
                dig[b] = (d & 0xffu) | (c << 12);
                c = me & 3u;
            } else {
                dig[b] = 64u;                   // every leaf below the split
            }
        }
        // the plane of a block's leaves below its split is that of the region before: the last one that started
        // in front of the run -- a few threads back at most (a region is kSyncMaxRun + 1 blocks at most)
        lastc[tid] = last;
        __syncthreads();
        uint32_t before = 0;
        if (b0 < b1) {                          // (a thread without blocks has nothing to look for: with 720 blocks for
                                                //  1024 threads the idle ones went back over 300 entries -- 15 us)
            for (uint32_t t = tid; t-- > 0;) {
                if (lastc[t] != 0xffffffffu) {
                    before = lastc[t];
                    break;
                }
            }
        }
        for (uint32_t b = b0; b < b1; ++b) {
            const uint32_t d = dig[b];
            dig[b] = (d & 0x7fu) | (before << 8) | (d & 0x3000u);
            if (d & 0x80u) before = (d >> 12) & 3u;
        }
        __syncthreads();
        for (uint32_t i = tid; i < nblk; i += kScanThreads) sp.sync_sel[gb0 + i] = dig[i];
        __syncthreads();
    }
}

// leaves of capture c live at events[e0 + c + i], i = 0 .. ne  (ne + 1 leaves)

// regular leaf i of capture `cap`, entered in code `in`: simulate the span, record what happened
__device__ __forceinline__ void emit_simulate(const ScanParams &sp, const LTab &T, uint32_t cap, const uint64_t *edges,
                                              size_t ev0, uint64_t i, uint32_t in) {
    const uint64_t e_before = edges[i - 1], e_at = edges[i];
    PSim f;
    Acc a;
    bool alive = true;
    if (in == code_poison(T)) {
        scan_refuse(sp, cap, (uint32_t)kFbPoison);
        acc_init(a);
        f.cur = f.nbits = f.k = f.prev = 0;
    } else {
        Span span;
        span.pos0 = e_before + 1;
        span.n = e_at - e_before - 1;
        span.L = (uint32_t)(i & 1ull) ^ T.lvl0;
        span.has_edge = true;
        span.prefix = 0;
        if (in >= T.S * T.NB1 + 3) {            // entered stuck: from where the state was normal (span_entered)
            uint32_t d, src;
            stuck_decode(T, in, d, src);
            span.prefix = e_before - edges[i - d - 1];
            in = src;
        }
        alive = run_leaf(T, in, span, next_buffer_start(T, e_before), f, a);
        if (a.overflow) scan_refuse(sp, cap, (uint32_t)kFbOverflow);
    }
    put_event(sp, ev0 + i, a, f, alive, false);
}

// leaves a workgroup of scan_emit_kernel puts aside to simulate together at its end (sync form)
constexpr uint32_t kEmitQueue = 1024;
struct EmitQ {
    uint32_t cap, in;
    uint64_t i;
};

__global__ __launch_bounds__(kSimThreads) void scan_emit_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ LTab T;
    __shared__ EmitQ s_q[kEmitQueue];
    __shared__ uint32_t s_qn;
    copy_ltab(T, sp.ltab);
    __syncthreads();
    if (threadIdx.x == 0) {
        T.lvl0 = sp.has_prev ? fsm_level_at(sp.f, 0, -1) : 0u;
        s_qn = 0;
    }
    __syncthreads();
    if (*sp.fallback) return;
    const bool sync_form = sp.sync_try && !(*sp.sync_fail & 1u);
    const uint32_t LB = sp.leaf_block;
    const uint32_t total = sp.cap_block_off[sp.f.num_captures];
    // Sync form: most leaves need no simulation.  The walk's copy of the rows (append_sync_codes) says per (state,
    // class) and length interval what the span leaves behind when that is nothing or one appended bit -- a pulse, a
    // bit gap: all but the leaves at a message's ends --, and scan_sync_kernel left every leaf's row offset: the rows
    // into LDS (g_mr, from its start: 1 200 words for the shipped devices), one read per leaf.  The few that are left
    // (2.6 % on the bench capture) are put aside and simulated TOGETHER at the workgroup's end: one or two of them in
    // every wave cost every wave the latency of a whole simulation (44 us, as much as simulating all of them).
    if (sync_form) {
        const uint32_t r2 = sp.lt_merged[3] + 2u * (sp.lt_merged[0] + sp.lt_merged[1]);
        const uint32_t n2 = (sp.lt_merged[0] + sp.lt_merged[1]) * 2u * T.S;
        for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) g_mr[i] = sp.lt_merged[r2 + i];
        __syncthreads();
    }
    // work items: groups of FOUR blocks, then one "ends" item per capture (first span + tail).
    // A workgroup takes four blocks = 256 leaves, one per lane: waves 0 / 1 the even / odd leaves of blocks 0 and 1,
    // waves 2 / 3 those of blocks 2 and 3 -- leaves of one parity run at one level, so a wave's lanes sit in the
    // same few states and run the same triggers, and every lane has a leaf.  (Round 2 gave a workgroup ONE block
    // and each of its four waves the leaves entered in "its" states: a quarter of the lanes busy, four times the
    // workgroup rounds -- 130 us per 16 GiB capture.)
    const uint32_t nquad = (total + 3u) / 4u;
    for (uint32_t wq = blockIdx.x; wq < nquad + sp.f.num_captures; wq += gridDim.x) {
        if (wq < nquad) {
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
            // sync form: a wave takes a block, lane = leaf -- few leaves are simulated at all (below), and a wave's
            // 64 records are 3 KB in a row; composing form: leaves of one parity (one level) per wave
            const uint32_t w = sync_form ? 4u * wq + wave : 4u * wq + 2u * (wave >> 1) + (lane >> 5);
            const uint32_t l = sync_form ? lane : 2u * (lane & 31u) + (wave & 1u);
            if (w < total) {
                uint32_t cap, lb;
                locate_block(sp, w, cap, lb);
                uint64_t e0;
                const uint64_t ne = cap_edges(sp.f, cap, e0);
                const uint64_t *edges = sp.f.edges + e0;
                const size_t ev0 = (size_t)e0 + cap;
                const uint64_t first = 1 + (uint64_t)lb * LB;
                const uint32_t count = (uint32_t)min((uint64_t)LB, ne - first);
                // The leaves' entry codes are there already (scan_entry_kernel): no table is staged, nothing goes
                // through the LDS, no barrier -- every lane asks for its leaf's code and its two edges at once.
                if (l < count) {
                    const uint64_t i = first + l;
                    const size_t at = (size_t)w * LB + l;
                    uint32_t in, kind = 0, to = 0;
                    if (sync_form) {
                        // the walk from synchronising spans keeps a plane per candidate and scan_syncpick_kernel says
                        // which: all four asked for together with the selection (one round trip, not two)
                        const uint32_t se = sp.sync_sel[w];
                        const uint32_t c0 = sp.pre_codes[at], c1 = sp.pre_codes[at + sp.pre_plane];
                        const uint32_t c2 = sp.pre_codes[at + 2 * sp.pre_plane], c3 = sp.pre_codes[at + 3 * sp.pre_plane];
                        const uint32_t rz = sp.rowz[at];
                        const uint32_t pl = (l < (se & 0x7fu) ? se >> 8 : se >> 12) & 3u;
                        in = pl == 0 ? c0 : pl == 1 ? c1 : pl == 2 ? c2 : c3;
                        const uint32_t pc = in & 0x7fu, pn = in >> 7;       // state | bit count << 7 (scan_syncwalk_kernel)
                        in = pc < T.S ? pc * T.NB1 + pn : (pc == T.S ? T.S * T.NB1 + pn : pn);
                        if (pc < T.S && rz != kSyncNone) {
                            const uint32_t q = g_mr[rz + 2u * pc + (pn >= T.max_bits ? 1u : 0u)];
                            kind = (int32_t)q < 0 ? (q >> 24) & 3u : 0u;
                            to = q & 0xffu;
                        }
                    } else {
                        in = sp.pre_codes[at];
                    }
                    if (kind) {
                        // nothing happened, or one bit was appended (the rows say which): no simulation
                        PSim f;
                        Acc a;
                        acc_init(a);
                        a.napp = kind >= 2u ? 1u : 0u;
                        a.appvals = kind == 3u ? 1u : 0u;
                        f.cur = to;
                        f.nbits = f.k = 0;
                        f.prev = ((uint32_t)(i & 1ull) ^ T.lvl0) ^ 1u;
                        put_event(sp, ev0 + i, a, f, true, false);
                    } else {
                        const uint32_t slot = sync_form ? atomicAdd(&s_qn, 1u) : kEmitQueue;
                        if (slot < kEmitQueue) {
                            s_q[slot].cap = cap;
                            s_q[slot].in = in;
                            s_q[slot].i = i;
                        } else {
                            emit_simulate(sp, T, cap, edges, ev0, i, in);
                        }
                    }
                }
            }
        } else {
            const uint32_t cap = wq - nquad;
            uint64_t e0;
            const uint64_t ne = cap_edges(sp.f, cap, e0);
            const uint64_t *edges = sp.f.edges + e0;
            const size_t ev0 = (size_t)e0 + cap;
            if (threadIdx.x == 0) {
                // first span, from the concrete incoming state
                PSim f;
                Acc a;
                const bool alive = first_leaf(T, sp, edges, ne, f, a);
                if (a.overflow) scan_refuse(sp, cap, (uint32_t)kFbOverflow);
                put_event(sp, ev0, a, f, alive, ne == 0);
            }
            if (threadIdx.x == 64 && ne > 0) {
                // tail: the samples after the last edge
                uint32_t in;
                if (ne == 1) {
                    PSim f0;
                    Acc a0;
                    const bool al0 = first_leaf(T, sp, edges, ne, f0, a0);
                    in = encode_post(T, f0, a0, al0);
                } else {
                    in = sp.cap_end[cap];
                }
                const uint64_t pos0 = edges[ne - 1] + 1;
                Span tail;
                tail.prefix = 0;
                tail.pos0 = pos0;
                tail.n = sp.f.n_out > pos0 ? sp.f.n_out - pos0 : 0;
                tail.L = (uint32_t)(ne & 1ull) ^ T.lvl0;
                tail.has_edge = false;
                if (in > code_poison(T)) {              // entered stuck: run it from where the state was normal
                    uint32_t d, src;
                    stuck_decode(T, in, d, src);
                    if (ne >= (uint64_t)d + 1) {
                        tail.prefix = edges[ne - 1] - edges[ne - d - 1];
                        in = src;
                    } else {
                        in = code_poison(T);
                    }
                }
                PSim f;
                Acc a;
                bool alive = true;
                if (in == code_poison(T)) {
                    scan_refuse(sp, cap, (uint32_t)kFbPoison);
                    acc_init(a);
                    f.cur = f.nbits = f.k = f.prev = 0;
                } else {
                    alive = run_leaf(T, in, tail, next_buffer_start(T, edges[ne - 1]), f, a);
                    if (a.overflow) scan_refuse(sp, cap, (uint32_t)kFbOverflow);
                }
                put_event(sp, ev0 + ne, a, f, alive, true);
            }
            __syncthreads();
        }
    }
    // the leaves put aside: simulated together
    __syncthreads();
    const uint32_t nq = min(s_qn, kEmitQueue);
    for (uint32_t j = threadIdx.x; j < nq; j += blockDim.x) {
        const EmitQ e = s_q[j];
        uint64_t e0;
        (void)cap_edges(sp.f, e.cap, e0);
        emit_simulate(sp, T, e.cap, sp.f.edges + e0, (size_t)e0 + e.cap, e.i, e.in);
    }
}

// ---- finish: records -> payloads / messages / errors / outgoing state --------------
//
// Appends get ordinals in time order (bits the machine already held when the
// capture / shard began count as ordinals 0 .. nb0-1).  `epoch` = ordinal of
// the first append after the last pass through reset: the payload at any time
// is appends epoch .. (epoch + max_bits), bit t = append epoch + t
// (state_machine.c:365-385: stored while num_bits <= max_bits).

__device__ __forceinline__ void locate_fin(const ScanParams &sp, uint32_t gfb, uint32_t &cap, uint32_t &fb) {
    uint32_t lo = 0, hi = sp.f.num_captures;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sp.fin_off[mid] <= gfb) lo = mid;
        else hi = mid;
    }
    cap = lo;
    fb = gfb - sp.fin_off[lo];
}

// workgroup inclusive scans over 1024 lanes: sums of (a, o, e) and max of r.
// Shuffles inside each wavefront, one exchange of the 16 wave totals; the
// inclusive values are also left in sh[k][lane] for the callers.
__device__ __forceinline__ void wg_scan4(uint32_t &a, uint32_t &o, uint32_t &e, uint32_t &r, uint32_t (*sh)[kFinBlock]) {
    __shared__ uint32_t wtot[4][kFinBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t va = __shfl_up(a, d), vo = __shfl_up(o, d), ve = __shfl_up(e, d), vr = __shfl_up(r, d);
        if ((int)lane >= d) {
            a += va;
            o += vo;
            e += ve;
            r = max(r, vr);
        }
    }
    if (lane == 63) {
        wtot[0][wave] = a;
        wtot[1][wave] = o;
        wtot[2][wave] = e;
        wtot[3][wave] = r;
    }
    __syncthreads();
    uint32_t pa = 0, po = 0, pe = 0, pr = 0;
    for (uint32_t w = 0; w < wave; ++w) {
        pa += wtot[0][w];
        po += wtot[1][w];
        pe += wtot[2][w];
        pr = max(pr, wtot[3][w]);
    }
    a += pa;
    o += po;
    e += pe;
    r = max(r, pr);
    sh[0][tid] = a;
    sh[1][tid] = o;
    sh[2][tid] = e;
    sh[3][tid] = r;
    __syncthreads();
}

struct FinLeaf {                // one lane = one leaf of a finish block
    LeafEvDev ev;
    bool have;
    uint32_t a_in, o_in, e_in;  // block-local exclusive prefixes
    uint32_t r_in;              // block-local (ordinal of last epoch start before this leaf) + 1, 0 = none
    uint32_t a_tot, o_tot, e_tot, r_tot;
};

__device__ __forceinline__ void fin_block_scan(const ScanParams &sp, uint32_t cap, uint32_t fb, FinLeaf &L,
                               uint32_t (*sh)[kFinBlock]) {
    uint64_t e0;
    const uint64_t ne = cap_edges(sp.f, cap, e0);
    const uint64_t i = (uint64_t)fb * kFinBlock + threadIdx.x;
    L.have = i <= ne && !(sp.cap_fallback && sp.f.num_captures > 1 && sp.cap_fallback[cap]);
    if (L.have) L.ev = get_event(sp, (size_t)e0 + cap + i);
    uint32_t a = L.have ? L.ev.napp : 0u, o = L.have ? L.ev.nout : 0u, e = L.have ? L.ev.nerr : 0u;
    const uint32_t na = a, no = o, nerr = e;
    uint32_t r = 0;
    // first get the append prefix, then the epoch-start candidates need it
    uint32_t dummy = 0;
    wg_scan4(a, o, e, dummy, sh);
    L.a_in = a - na;
    L.o_in = o - no;
    L.e_in = e - nerr;
    L.a_tot = sh[0][kFinBlock - 1];
    L.o_tot = sh[1][kFinBlock - 1];
    L.e_tot = sh[2][kFinBlock - 1];
    __syncthreads();
    r = (L.have && (L.ev.flags & 1u)) ? L.a_in + L.ev.apps_at_reset + 1 : 0u;
    uint32_t z0 = 0, z1 = 0, z2 = 0;
    const uint32_t mine = r;
    wg_scan4(z0, z1, z2, r, sh);
    L.r_tot = sh[3][kFinBlock - 1];
    // exclusive: the value of the lane before
    __syncthreads();
    L.r_in = threadIdx.x ? sh[3][threadIdx.x - 1] : 0u;
    (void)mine;
    __syncthreads();
}

// Finish blocks are numbered capture-major, so "everything before block g" is
// a plain prefix: outputs / errors over ALL earlier blocks give the global
// message / error slot (deterministic order, no atomics), appends and the
// epoch only over the earlier blocks of the same capture.
//
// One pass: a workgroup takes a ticket g, scans its 1024 leaves, publishes the
// block aggregate, then adds up its predecessors' aggregates (they took their
// tickets earlier, publish before they wait for anything, so they arrive) and
// writes.  An aggregate is four 64-bit words  value | run stamp << 32: a reader
// simply re-reads a word (device-scope atomic load, past the L1) until it
// carries this run's stamp.

__device__ __forceinline__ void agg_store(unsigned long long *slot, uint32_t v, uint32_t stamp) {
    __hip_atomic_store(slot, (unsigned long long)v | ((unsigned long long)stamp << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t agg_load(const unsigned long long *slot, uint32_t stamp) {
    for (;;) {
        const unsigned long long v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) == stamp) return (uint32_t)v;
        __builtin_amdgcn_s_sleep(2);
    }
}

// A work counter that needs no zeroing between launches: count | run stamp << 32.  Whoever finds another
// run's stamp in it swaps in  0 | this run's stamp  (one of the finders wins, the others see the new stamp
// and carry on); counts are only ever added to a word that carries the current stamp.  Tickets 0, 1, 2, ...
// are handed out exactly once per launch whatever the word held before -- nothing depends on a memset
// having been ordered in front of the kernel, or on its visibility.
__device__ __forceinline__ uint32_t take_stamped_ticket(unsigned long long *w, uint32_t stamp) {
    for (;;) {
        const unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(v >> 32) != stamp) {
            unsigned long long expect = v;
            (void)__hip_atomic_compare_exchange_strong(w, &expect, (unsigned long long)stamp << 32, __ATOMIC_RELAXED,
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const unsigned long long old = __hip_atomic_fetch_add(w, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(old >> 32) == stamp) return (uint32_t)old;
    }
}

__device__ __forceinline__ uint64_t pool_start(uint64_t e0, uint32_t cap) {
    return 2 * (e0 + cap) + (uint64_t)cap * 512;    // 2 per leaf + 512 per capture
}

// append values by ordinal, error list, message descriptors (sample, epoch, count)
__global__ __launch_bounds__(kFinBlock) void fin_write_kernel(ScanParams sp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t sh[4][kFinBlock];
    __shared__ uint32_t s_g;
    __shared__ uint32_t s_acc[4];       // appends (same capture), outputs, errors (all), last block with a reset + 1
    // bits other kernels may have raised are final by now; kFbPool is raised in here and must
    // not make a later workgroup skip its publication
    if (*sp.fallback & ~(uint32_t)kFbPool) return;
    const uint32_t total = sp.fin_off[sp.f.num_captures];
    const uint32_t max_bits = sp.f.tables->max_bits;
    const uint32_t stamp = sp.run_stamp;
    const uint32_t tid = threadIdx.x;
    for (;;) {
        if (tid == 0) {
            s_g = take_stamped_ticket(sp.fin_ticket, stamp);
            s_acc[0] = s_acc[1] = s_acc[2] = s_acc[3] = 0;
        }
        __syncthreads();
        const uint32_t g = s_g;
        if (g >= total) break;
        uint32_t cap, fb;
        locate_fin(sp, g, cap, fb);
        FinLeaf L;
        fin_block_scan(sp, cap, fb, L, sh);
        if (tid == 0) {
            // two stamped words: appends (20 bits) | outputs << 20 (12), errors (12) | epoch << 12 (20)
            // -- 1024 leaves of at most 255 appends, 2 outputs, 1 error each
            unsigned long long *slot = sp.fagg + 4 * (size_t)g;
            agg_store(slot + 0, L.a_tot | (L.o_tot << 20), stamp);
            agg_store(slot + 1, L.e_tot | (L.r_tot << 12), stamp);
        }
        // ---- predecessors -----------------------------------------------------------------
        const uint32_t g0 = sp.fin_off[cap];
        {
            uint32_t a = 0, o = 0, e = 0, last = 0;
            for (uint32_t j = tid; j < g; j += blockDim.x) {
                const unsigned long long *slot = sp.fagg + 4 * (size_t)j;
                const uint32_t w0 = agg_load(slot + 0, stamp), w1 = agg_load(slot + 1, stamp);
                o += w0 >> 20;
                e += w1 & 0xfffu;
                if (j >= g0) {
                    a += w0 & 0xfffffu;
                    if (w1 >> 12) last = j + 1;
                }
            }
            if (a) atomicAdd(&s_acc[0], a);
            if (o) atomicAdd(&s_acc[1], o);
            if (e) atomicAdd(&s_acc[2], e);
            if (last) atomicMax(&s_acc[3], last);
        }
        __syncthreads();
        const FsmStateDev fs = scan_first(sp);
        const uint32_t nb0 = sp.have_first ? min(fs.nbits, max_bits + 1) : 0u;
        const uint32_t a_before = nb0 + s_acc[0], o_before = s_acc[1], e_before = s_acc[2];
        const uint64_t base_m = sp.totals_in ? sp.totals_in[0] : 0ull, base_e = sp.totals_in ? sp.totals_in[1] : 0ull;
        const uint32_t jr = s_acc[3];           // block jr-1 holds the latest reset before this block
        __syncthreads();
        uint32_t epoch_in = 0;
        if (jr) {
            // appends of this capture before block jr-1, then that block's last epoch start
            if (tid == 0) s_acc[0] = 0;
            __syncthreads();
            uint32_t a = 0;
            for (uint32_t j = g0 + tid; j + 1 < jr; j += blockDim.x) a += agg_load(sp.fagg + 4 * (size_t)j, stamp) & 0xfffffu;
            if (a) atomicAdd(&s_acc[0], a);
            __syncthreads();
            epoch_in = nb0 + s_acc[0] + (agg_load(sp.fagg + 4 * (size_t)(jr - 1) + 1, stamp) >> 12) - 1;
            __syncthreads();
        }
        uint64_t e0;
        const uint64_t ne = cap_edges(sp.f, cap, e0);
        const bool last_of_cap = g + 1 == sp.fin_off[cap + 1];
        if (tid == 0) {
            if (last_of_cap && (uint64_t)(a_before + L.a_tot) > 2 * (ne + 1) + 512) atomicOr(sp.fallback, (uint32_t)kFbPool);
            if (g + 1 == total) {
                sp.f.totals[0] = base_m + o_before + L.o_tot;
                sp.f.totals[1] = base_e + e_before + L.e_tot;
            }
        }
        const uint64_t pool0 = pool_start(e0, cap);
        if (pool0 + 2 * (ne + 1) + 512 > sp.app_capacity) {
            if (tid == 0) atomicOr(sp.fallback, (uint32_t)kFbPool);
            continue;
        }
        uint8_t *vals = sp.app_vals + pool0;
        if (fb == 0 && tid < nb0) {
            // (selects, not fs.data[tid >> 6]: a run-time index would put the state into scratch memory)
            static_assert(kPayloadWords == 5, "select chain below");
            const uint32_t q = tid >> 6;
            const uint64_t w = q == 0 ? fs.data[0] : q == 1 ? fs.data[1] : q == 2 ? fs.data[2] : q == 3 ? fs.data[3] : fs.data[4];
            vals[tid] = (uint8_t)((w >> (tid & 63)) & 1ull);
        }
        if (fb == 0 && tid == 0 && nb0 > (uint32_t)kFinBlock) atomicOr(sp.fallback, (uint32_t)kFbPool);
        if (!L.have) continue;
        const uint32_t ai = a_before + L.a_in;
        const uint32_t epoch = L.r_in ? a_before + L.r_in - 1 : epoch_in;
        // (the pool check above runs in the capture's last block only, after the earlier ones have
        //  written: a leaf that would write past its capture's pool stores nothing and refuses here)
        const uint64_t pool_len = 2 * (ne + 1) + 512;
        if ((uint64_t)ai + L.ev.napp > pool_len) {
            atomicOr(sp.fallback, (uint32_t)kFbPool);
        } else {
            for (uint32_t j = 0; j < L.ev.napp && j < 32; ++j) vals[ai + j] = (uint8_t)((L.ev.appvals >> j) & 1u);
        }
        if (L.ev.nerr) {
            const uint64_t slot = base_e + e_before + L.e_in;
            if (slot < sp.err_capacity) sp.errs[slot] = L.ev.err_pos + sp.pos_origin;
        }
        // (constant indices: a run-time index into the record's arrays puts the whole record into
        //  scratch memory -- 160 bytes per lane, written and read back: 118 MB per 16 GiB capture)
#pragma unroll
        for (uint32_t j = 0; j < 2; ++j) {
            if (j >= L.ev.nout) break;
            const uint32_t rb = j ? L.ev.out_rb[1] : L.ev.out_rb[0], ab = j ? L.ev.out_ab[1] : L.ev.out_ab[0];
            const uint64_t opos = j ? L.ev.out_pos[1] : L.ev.out_pos[0];
            const uint32_t ep = rb != 0xffu ? ai + rb : epoch;
            const uint32_t have = ai + ab - ep;
            const uint64_t slot = base_m + o_before + L.o_in + j;
            if (slot < sp.f.msg_capacity) {
                uint4 *dst = reinterpret_cast<uint4 *>(sp.f.msgs + slot);
                const uint64_t sample = opos + sp.pos_origin;
                // capture, reserved | sample | payload[0] = epoch start | bits << 32 (resolved by fin_msg_kernel) | 0 ...
                dst[0] = make_uint4(cap, 0u, (uint32_t)sample, (uint32_t)(sample >> 32));
                dst[1] = make_uint4(ep, have, 0u, 0u);
                dst[2] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // outgoing state: the lane that owns the tail leaf
        if ((uint64_t)fb * kFinBlock + tid == ne) {
            SegState so;
            so.st.cur = L.ev.end_cur;
            so.st.k = L.ev.end_k;
            so.st.prev = L.ev.end_prev;
            so.st.pad = 0;
            const uint32_t ep_end = (L.ev.flags & 1u) ? ai + L.ev.apps_at_reset : epoch;
            const uint32_t have = ai + L.ev.napp - ep_end;
            so.st.nbits = have;
            so.st.data[0] = ep_end;             // resolved by fin_msg_kernel
            for (int q = 1; q < kPayloadWords; ++q) so.st.data[q] = 0;
            if (L.ev.flags & 4u) {
                // ended inside a skipped rest-of-buffer: reset, k = 0
                so.st.cur = 0;
                so.st.k = 0;
            }
            so.skip_to = 0;
            so.pad = 0;
            sp.final_state[cap] = so;
        }
    }
}

// gathers payload bits: one lane per message, then one per capture for the outgoing state
// Last kernel of the scan: resolves the payloads and, with `pp`, also does the
// publish step (edges_fsm.hip: publish_kernel) -- resolved messages go to the
// host copy as they are produced, the workgroup that finishes last copies the
// header and zeroes the device one.
__global__ __launch_bounds__(256) void fin_msg_kernel(ScanParams sp, PublishParams pp) {
    __builtin_amdgcn_s_setprio(3);      // latency chain: issue ahead of a front-end kernel sharing the CU
    __shared__ uint32_t s_last;
    const bool refused = *sp.fallback != 0;
    const uint32_t max_bits = sp.f.tables->max_bits;
    const uint32_t nbytes = (max_bits + 7u) >> 3;
    const uint64_t nmsg = refused ? 0 : min((uint64_t)sp.f.totals[0], sp.f.msg_capacity);
    const uint64_t nitems = refused ? 0 : nmsg + sp.f.num_captures;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // (a chunk of a pipelined run resolves its own messages: those behind the chunks before's)
    const uint64_t m0 = refused ? 0 : min(sp.totals_in ? sp.totals_in[0] : 0ull, nmsg);
    MsgDev *h_msgs = reinterpret_cast<MsgDev *>(pp.h_msgs);
    for (uint64_t m = m0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < nitems; m += stride) {
        if (m < nmsg) {
            MsgDev mm = sp.f.msgs[m];
            uint64_t e0;
            cap_edges(sp.f, mm.capture, e0);
            const uint8_t *vals = sp.app_vals + pool_start(e0, mm.capture);
            const uint32_t ep = (uint32_t)mm.payload[0], have = (uint32_t)(mm.payload[0] >> 32);
            const uint32_t take = min(have, min(max_bits + 1, 256u));
            // append values are bytes 0 / 1: eight at a time, squeezed to a byte of bits
            uint64_t wv[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                uint64_t w = 0;
#pragma unroll
                for (uint32_t g = 0; g < 8; ++g) {
                    const uint32_t t = 64 * q + 8 * g;
                    if (t < take) {
                        uint64_t x = 0;
                        __builtin_memcpy(&x, vals + ep + t, 8);        // the pool is padded past its end
                        const uint32_t left = take - t;
                        if (left < 8) x &= (1ull << (8 * left)) - 1ull;
                        w |= ((x * 0x0102040810204080ull) >> 56) << (8 * g);
                    }
                }
                wv[q] = w;
            }
            for (uint32_t q = 0; q < 4; ++q) {
                uint64_t v = wv[q];
                if (8 * q >= nbytes) v = 0;
                else if (8 * (q + 1) > nbytes) v &= (1ull << ((nbytes - 8 * q) * 8)) - 1ull;
                mm.payload[q] = v;
            }
            sp.f.msgs[m] = mm;
            if (h_msgs && m < pp.first_msgs) h_msgs[m] = mm;
        } else {
            const uint32_t cap = (uint32_t)(m - nmsg);
            SegState so = sp.final_state[cap];
            uint64_t e0;
            cap_edges(sp.f, cap, e0);
            const uint8_t *vals = sp.app_vals + pool_start(e0, cap);
            const uint32_t ep = (uint32_t)so.st.data[0];
            const uint32_t take = min(so.st.nbits, min(max_bits + 1, 320u));
            for (int q = 0; q < kPayloadWords; ++q) so.st.data[q] = 0;
            for (uint32_t t = 0; t < take; ++t) {
                if (vals[ep + t]) so.st.data[t >> 6] |= 1ull << (t & 63);
            }
            sp.final_state[cap] = so;
        }
    }
    if (!pp.d_hdr) return;
    // ---- publish: the last workgroup to get here owns the header ----------------------------
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(pp.d_hdr + pp.done_word, 1u) + 1u == gridDim.x ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const uint32_t tid = threadIdx.x;
    if (tid < pp.hdr_words) {
        uint32_t v = pp.d_hdr[tid];
        if (tid == pp.edges_word && pp.total_edges) v = *pp.total_edges;
        pp.h_hdr[tid] = v;
    }
    __syncthreads();
    if (tid < pp.hdr_words) pp.d_hdr[tid] = 0;
}

// ---------------------------------------------------------------------------
// span tables (host): the same simulator, run once per interval of span lengths
// ---------------------------------------------------------------------------
//
// From a normal state the packed result of a span depends only on its level
// and its length n.  While the level is constant the machine follows a fixed
// trajectory of always / timeout / msg_complete firings; the edge that ends
// the span is then judged with a counter k = n - (time of the last firing)
// against a handful of bounds of the state it meets.  So the result is a step
// function of n whose steps can only lie next to   firing time + bound.  Those
// candidates are enumerated from the trajectory, the simulator -- the very
// code the kernels run -- is evaluated at each, equal neighbours are merged,
// and every merged interval is probed again inside; any surprise disables the
// tables (the kernels then simulate, as they do for rows marked 0).

namespace {

void fill_ltab_host(LTab &T, const FsmTablesDev &g, uint32_t spb, uint32_t decim) {
    for (uint32_t i = 0; i < (uint32_t)kMaxTriggers; ++i) {
        T.tr[i] = make_uint4(clamp32(g.trig_kmin[i]), clamp32(g.trig_kmax[i]), g.trig_info[i], 0u);
    }
    for (uint32_t i = 0; i < (uint32_t)kMaxStates; ++i) {
        uint32_t msgc = 0;
        for (uint32_t t = g.state_tbeg[i]; t < g.state_tend[i] && t < (uint32_t)kMaxTriggers; ++t) {
            if ((g.trig_info[t] & 0xffu) == kCondMsgComplete) msgc = 1;
        }
        const uint32_t row = (g.state_tbeg[i] & 0xffu) | ((g.state_tend[i] & 0xffu) << 8) |
                             ((g.state_flags[i] & 1u) << 16) | (msgc << 17);
        T.st[i] = make_uint4(clamp32(g.state_kmin[i]), clamp32(g.state_kmax[i]), clamp32(g.state_kto[i]), row);
    }
    T.max_bits = g.max_bits;
    T.S = g.num_states;
    T.NB1 = g.max_bits + 2;
    T.D = g.num_states * (g.max_bits + 2) + 3;
    T.spb = spb;
    T.decim = decim;
    T.buf_shift = 0xffffffffu;
    if (decim && spb % decim == 0) {
        const uint32_t q = spb / decim;
        if (q && !(q & (q - 1))) T.buf_shift = (uint32_t)__builtin_ctz(q);
    }
    T.NS = 0;
    T.nstuck_rows = 0;
    T.depth = 0;
    T.lvl0 = 0;
}

// adds the stuck codes (found by build_leaf_tables) to the domain
void add_stuck(LTab &T, const std::vector<uint16_t> &stuck_src, const std::vector<uint8_t> &stuck_rows) {
    if (stuck_src.empty() || stuck_src.size() > kMaxStuck || stuck_rows.size() > kMaxStuckRows) return;
    const uint32_t room = T.D < kStuckDomain ? (kStuckDomain - T.D) / (uint32_t)stuck_src.size() : 0u;
    if (room == 0) return;                              // no room in the domain: they stay poison
    T.depth = room < kStuckDepth ? room : kStuckDepth;
    T.NS = (uint32_t)stuck_src.size();
    T.nstuck_rows = (uint32_t)stuck_rows.size();
    for (size_t i = 0; i < stuck_src.size(); ++i) T.stuck_src[i] = stuck_src[i];
    for (size_t i = 0; i < stuck_rows.size(); ++i) T.stuck_row[i] = stuck_rows[i];
    T.D += T.NS * T.depth;
}

constexpr uint64_t kProbePos = 1ull << 40;      // spans of normal rows do not depend on where they lie

// packed result of a span of n constant samples + its edge; row < 2S: (state, class);
// otherwise start in reset with previous level row - 2S != L (0 = position dependent)
uint32_t eval_row(const LTab &T, uint32_t row, uint32_t L, uint64_t n) {
    PSim f;
    Acc a;
    if (row < 2 * T.S) {
        const uint32_t k = row >> 1, cls = row & 1u;
        const uint32_t nb0 = cls ? T.max_bits : 0u;
        const SimRes r = run_leaf_raw(T, k * T.NB1 + nb0, kProbePos, n, L | 2u, 0);
        const bool alive = sim_unpack(r, f, a);
        return pack_normal(T, f, a, alive, nb0, cls);
    }
    const uint32_t code = T.S * T.NB1 + (row - 2 * T.S);
    uint32_t out[2];
    for (int v = 0; v < 2; ++v) {
        const uint64_t pos = kProbePos + (v ? 12345 : 0);
        const SimRes r = run_leaf_raw(T, code, pos, n, L | 2u, pos);
        const bool alive = sim_unpack(r, f, a);
        // an error before the span's last sample resumes at the next buffer boundary:
        // where that is depends on the span's position (one on the last sample ends the span skipping)
        if (a.nerr && a.err_pos + 1 < pos + n + 1) return 0u;
        out[v] = encode_post(T, f, a, alive);
    }
    return out[0] == out[1] ? pack_absolute(out[0], T.NB1) : 0u;
}

void add_candidates(std::vector<uint64_t> &c, uint64_t t, uint32_t bound) {
    if (bound == kNone) return;
    for (uint64_t b : {(uint64_t)bound, (uint64_t)bound / 2}) {         // reset counts twice per sample
        for (int d = -3; d <= 3; ++d) {
            const int64_t v = (int64_t)(t + b) + d;
            if (v >= 0 && v <= 0xfffffff0ll) c.push_back((uint64_t)v);
        }
    }
}

// span lengths next to which the result of row (start state f) can change
void row_candidates(const LTab &T, PSim f, uint32_t L, bool first_is_edge, std::vector<uint64_t> &c) {
    Acc a;
    acc_init(a);
    uint64_t pos = 0;
    for (uint64_t v = 0; v < 8; ++v) c.push_back(v);
    if (first_is_edge) {                                // previous level != L: the first sample is an edge
        p_step(T, f, a, L, pos);
        f.prev = L;
        canon(T, f);
        pos = 1;
    }
    for (uint32_t it = 0; it < 4 * kMaxFires + 8; ++it) {
        const uint4 srec = T.st[f.cur];
        add_candidates(c, pos, 0);
        add_candidates(c, pos, srec.x);
        add_candidates(c, pos, srec.y);
        add_candidates(c, pos, srec.z);
        for (uint32_t t = srec.w & 0xffu; t < ((srec.w >> 8) & 0xffu) && t < (uint32_t)kMaxTriggers; ++t) {
            add_candidates(c, pos, T.tr[t].x);
            add_candidates(c, pos, T.tr[t].y);
        }
        const uint32_t q = p_quiet(T, f, a);
        if (q == kNone) break;
        const uint64_t m = f.cur == 0 ? (uint64_t)(q >> 1) : (uint64_t)q;     // as sim_span advances
        f.k = sat_add(f.k, f.cur == 0 ? 2 * m : m);
        if (m > 0) canon(T, f);
        pos += m;
        p_step(T, f, a, L, pos);
        f.prev = L;
        canon(T, f);
        pos += 1;
        if (pos > 0xfffffff0ull) break;
    }
}

}  // namespace

size_t fsm_scan_ltab_bytes() { return sizeof(LTab); }

uint32_t fsm_scan_fill_ltab(void *dst, const FsmTablesDev &g, uint32_t spb, uint32_t decim,
                            const std::vector<uint16_t> &stuck_src, const std::vector<uint8_t> &stuck_rows) {
    memset(dst, 0, sizeof(LTab));
    LTab &T = *static_cast<LTab *>(dst);
    fill_ltab_host(T, g, spb, decim);
    add_stuck(T, stuck_src, stuck_rows);
    return T.D;
}

// Abstract codes a span can be entered in, and with which level: closure of {reset at
// level 0, skip x2, poison} under every result the span tables hold (any length).  Entries
// that need a simulation add nothing: they are spans with an error before their last
// sample, which end either skipping or as a run from (reset, no bits) at a buffer start
// with the previous level equal to the span's -- a result of row (reset, few bits) again.
// A normal code whose (row, level) holds a "stuck" result while it is reachable at that
// level joins stuck_src: those codes get STUCK_1..kStuckDepth twins in the domain (results
// of a stuck code are results of its row again, at either level).
// reach empty = unknown (a bit-count sensitive result): the kernels then compose over all
// codes.  (A code missed here would only cost speed: its entries are poison, and a
// poisoned capture falls back to the rounds.)
static void reachable_codes(LTab &T, const std::vector<uint32_t> &off, const std::vector<uint32_t> &pk,
                            std::vector<uint16_t> &reach, std::vector<uint16_t> &stuck_src,
                            std::vector<uint8_t> &stuck_rows) {
    const uint32_t S = T.S, NB1 = T.NB1, SNB = S * NB1, D0 = SNB + 3;
    std::vector<char> in(2 * D0, 0), stuck(SNB, 0);
    std::vector<uint32_t> todo;
    bool unknown = false;
    auto add = [&](uint32_t code, uint32_t L) {
        if (code < D0 && !in[2 * code + L]) {
            in[2 * code + L] = 1;
            todo.push_back(2 * code + L);
        }
    };
    // results of table (row, L) entered with nb bits; they are entered at level `next`
    auto follow = [&](uint32_t row, uint32_t L, uint32_t nb, uint32_t next) -> bool {
        bool any_stuck = false;
        for (uint32_t i = off[2 * row + L]; i < off[2 * row + L + 1]; ++i) {
            const uint32_t v = pk[i];
            uint32_t c = D0;
            if (v & kPkAbsolute) {
                c = v & 0xffffu;
            } else if (v & kPkRelative) {
                const uint32_t nbo = nb + ((v >> 8) & 0xffffu);
                c = (v & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
            } else if (v & kPkSensitive) {
                unknown = true;
            } else if (v & kPkStuck) {
                any_stuck = true;
            }
            if (c < D0) add(c, next);
        }
        return any_stuck;
    };
    reach.clear();
    stuck_src.clear();
    stuck_rows.clear();
    add(0, 0);                                          // a capture starts in reset, at level 0
    for (uint32_t L = 0; L < 2; ++L) {
        add(SNB, L);
        add(SNB + 1, L);
        add(SNB + 2, L);
    }
    while (!todo.empty() && !unknown) {
        const uint32_t code = todo.back() >> 1, L = todo.back() & 1u;
        todo.pop_back();
        if (code == SNB + 2) continue;
        if (code >= SNB) {
            const uint32_t kk = code - SNB;             // skipping ends inside the span, or goes on
            follow(kk == L ? 0u : 2 * S + kk, L, 0u, L ^ 1u);
            add(code, L ^ 1u);
            continue;
        }
        const uint32_t cur = code / NB1, nb = code - cur * NB1;
        const uint32_t row = 2 * cur + (nb >= T.max_bits ? 1u : 0u);
        if (follow(row, L, nb, L ^ 1u)) {
            stuck[code] = 1;
            const uint8_t rl = (uint8_t)(row | (L << 7));
            if (row < 128 && std::find(stuck_rows.begin(), stuck_rows.end(), rl) == stuck_rows.end()) stuck_rows.push_back(rl);
            follow(row, 0, nb, 1);                      // what its stuck twins can end in
            follow(row, 1, nb, 0);
        }
    }
    if (getenv("OOKD_DEBUG_REACH")) {
        size_t ns = 0, nr = 0;
        for (char c : stuck) ns += c;
        for (char c : in) nr += c;
        fprintf(stderr, "[reach] unknown %d stuck %zu in %zu todo %zu\n", (int)unknown, ns, nr, todo.size());
    }
    if (unknown) return;
    for (uint32_t c = 0; c < SNB; ++c) {
        if (!stuck[c]) continue;
        stuck_src.push_back((uint16_t)c);
    }
    add_stuck(T, stuck_src, stuck_rows);
    if (!T.NS) {
        stuck_src.clear();                              // too many: they stay unrepresentable (poison)
        stuck_rows.clear();
    }
    // entry = code | levels it is met at << 14 (bit 14: level 0, bit 15: level 1)
    for (uint32_t c = 0; c < D0; ++c) {
        if (in[2 * c] || in[2 * c + 1]) reach.push_back((uint16_t)(c | (in[2 * c] ? 0x4000u : 0u) | (in[2 * c + 1] ? 0x8000u : 0u)));
    }
    for (uint32_t c = D0; c < T.D; ++c) reach.push_back((uint16_t)(c | 0xc000u));
}

bool build_leaf_tables(const FsmTablesDev &g, uint32_t spb, uint32_t decim, std::vector<uint32_t> &off,
                       std::vector<uint32_t> &n0, std::vector<uint32_t> &pk, std::vector<uint16_t> &reach,
                       std::vector<uint16_t> &stuck_src, std::vector<uint8_t> &stuck_rows) {
    std::unique_ptr<LTab> Tp(new LTab());
    LTab &T = *Tp;
    fill_ltab_host(T, g, spb, decim);
    const uint32_t S = T.S, rows = 2 * S + 2;
    off.assign(2 * rows + 1, 0);
    n0.clear();
    pk.clear();
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    for (uint32_t row = 0; row < rows; ++row) {
        for (uint32_t L = 0; L < 2; ++L) {
            off[2 * row + L] = (uint32_t)n0.size();
            const bool special = row >= 2 * S;
            if (special && (row - 2 * S) == L) continue;        // same level: the kernels use row 0
            PSim f;
            f.k = 0;
            if (special) {
                f.cur = 0;
                f.nbits = 0;
                f.prev = row - 2 * S;
            } else {
                f.cur = row >> 1;
                f.nbits = (row & 1u) ? T.max_bits : 0u;
                f.prev = L;
            }
            std::vector<uint64_t> c;
            row_candidates(T, f, L, special, c);
            c.push_back(0);
            std::sort(c.begin(), c.end());
            c.erase(std::unique(c.begin(), c.end()), c.end());
            if (c.size() > 20000) return false;                 // keep create time bounded: simulate instead
            const size_t begin = n0.size();
            uint32_t prev_pk = 0;
            for (size_t i = 0; i < c.size(); ++i) {
                const uint32_t v = eval_row(T, row, L, c[i]);
                if (i == 0 || v != prev_pk) {
                    n0.push_back((uint32_t)c[i]);
                    pk.push_back(v);
                    prev_pk = v;
                }
            }
            // safety net: probe every interval again -- all of it when short, evenly spread
            // points (plus both ends) when long; any surprise disables the tables
            for (size_t i = begin; i < n0.size(); ++i) {
                const uint64_t a = n0[i], b = i + 1 < n0.size() ? n0[i + 1] : 0xfffffff0ull;
                const uint64_t len = b - a;
                const uint64_t steps = len <= 512 ? len : 128;
                for (uint64_t j = 0; j < steps; ++j) {
                    rng ^= rng << 13;
                    rng ^= rng >> 7;
                    rng ^= rng << 17;
                    const uint64_t x = len <= 512 ? a + j : a + (len / steps) * j + rng % (len / steps);
                    if (eval_row(T, row, L, x) != pk[i]) return false;
                }
                if (eval_row(T, row, L, b - 1) != pk[i]) return false;
            }
            if (n0.size() > (1u << 16)) return false;
        }
    }
    off[2 * rows] = (uint32_t)n0.size();
    reachable_codes(T, off, pk, reach, stuck_src, stuck_rows);
    return true;
}

// ---------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------

uint32_t fsm_scan_leaf_block(uint32_t D, uint32_t S, uint32_t SNB) {
    // 64 leaves per block: about one simulation task per lane, and the tables
    // (~35 KiB for the shipped devices) let several workgroups share a CU
    uint32_t lb = 64;
    while (lb > 16 && block_lds_bytes(lb, D, S, SNB) > 140u * 1024u) lb >>= 1;
    return lb;
}

std::vector<uint32_t> build_merged_rows(uint32_t S, const std::vector<uint32_t> &off, const std::vector<uint32_t> &n0,
                                        const std::vector<uint32_t> &pk) {
    auto lookup = [&](uint32_t row, uint32_t L, uint32_t n) -> uint32_t {      // lt_lookup
        uint32_t lo = off[2 * row + L], hi = off[2 * row + L + 1];
        if (lo >= hi) return 0u;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (n0[mid] <= n) lo = mid;
            else hi = mid;
        }
        return pk[lo];
    };
    std::vector<uint32_t> bp[2];
    for (uint32_t L = 0; L < 2; ++L) {
        bp[L].push_back(0);
        for (uint32_t r = 0; r < 2 * S; ++r)
            for (uint32_t i = off[2 * r + L]; i < off[2 * r + L + 1]; ++i) bp[L].push_back(n0[i]);
        std::sort(bp[L].begin(), bp[L].end());
        bp[L].erase(std::unique(bp[L].begin(), bp[L].end()), bp[L].end());
    }
    std::vector<uint32_t> out = {(uint32_t)bp[0].size(), (uint32_t)bp[1].size(), 2 * S, 0u};
    out.insert(out.end(), bp[0].begin(), bp[0].end());
    out.insert(out.end(), bp[1].begin(), bp[1].end());
    for (uint32_t L = 0; L < 2; ++L) {
        for (uint32_t n : bp[L]) {
            for (uint32_t k = 0; k < S; ++k) {
                // (what the leaf kernels make of a leaf's two lookups: a shared class-0 result also stands for class 1)
                const uint32_t p0 = lookup(2 * k, L, n);
                const uint32_t p1 = (p0 & kPkShared) ? p0 : lookup(2 * k + 1, L, n);
                out.push_back(p0 & ~kPkShared);
                out.push_back(p1 & ~kPkShared);
            }
        }
    }
    return out;
}

// What a span does to the codes it can be entered in, per interval of the merged rows -- for the walk from
// synchronising spans (scan_sync_kernel).  Two words per interval behind the rows, header word [3] = where they
// start:  c0 | c1 << 16,  c2 | (n | stuckable << 4) << 16
//   n (1..3): every normal code a span of that level can be entered in (reach: code | level mask << 14 -- bit 14:
//             met at level 0, bit 15: at level 1; empty = every normal code at either level) ends in one of the n
//             codes c0..c2 (absolute results, relative ones evaluated; a skip code where the entry runs into an
//             error on the span's last sample);  n = 0: more than three, or a row that is stuck / bit-count
//             sensitive / position dependent;
//   stuckable: some such entry finds no trigger on the edge (the leaf can end in a stuck code).
void append_sync_codes(std::vector<uint32_t> &merged, uint32_t S, uint32_t NB1, uint32_t max_bits,
                       const std::vector<uint16_t> &reach) {
    if (merged.size() < 4) return;
    const uint32_t nbp[2] = {merged[0], merged[1]}, twoS = 2 * S, SNB = S * NB1;
    const uint32_t rows0 = 4 + nbp[0] + nbp[1];
    std::vector<uint16_t> codes[2];
    for (uint32_t L = 0; L < 2; ++L) {
        if (reach.empty()) {
            for (uint32_t c = 0; c < SNB; ++c) codes[L].push_back((uint16_t)c);
        } else {
            for (uint16_t v : reach)
                if ((v & 0x3fffu) < SNB && (v & (0x4000u << L))) codes[L].push_back((uint16_t)(v & 0x3fffu));
        }
    }
    std::vector<uint32_t> out;
    for (uint32_t L = 0; L < 2; ++L) {
        for (uint32_t z = 0; z < nbp[L]; ++z) {
            const uint32_t *row = merged.data() + rows0 + (size_t)((L ? nbp[0] : 0u) + z) * twoS;
            uint32_t img[3] = {0xffffu, 0xffffu, 0xffffu}, n = 0, stuckable = 0;
            bool ok = !codes[L].empty();
            for (uint16_t c : codes[L]) {
                const uint32_t cur = c / NB1, nb = c - cur * NB1;
                const uint32_t p = row[2 * cur + (nb >= max_bits ? 1u : 0u)];
                uint32_t end;
                if (p & kPkAbsolute) {
                    end = p & 0xffffu;
                    if (end >= SNB + 2) ok = false;         // poison: not an exit the walk carries
                } else if (p & kPkRelative) {
                    const uint32_t nbo = nb + ((p >> 8) & 0xffffu);
                    end = (p & 0xffu) * NB1 + (nbo >= NB1 ? NB1 - 1 : nbo);
                } else {
                    if (p & kPkStuck) stuckable = 1;
                    ok = false;                             // stuck, bit-count sensitive, position dependent
                    continue;
                }
                uint32_t j = 0;
                while (j < n && img[j] != end) ++j;
                if (j == n) {
                    if (n == 3) ok = false;
                    else img[n++] = end;
                }
            }
            if (!ok) n = 0;
            out.push_back(img[0] | (img[1] << 16));
            out.push_back(img[2] | ((n | (stuckable << 4)) << 16));
        }
    }
    // ... and the rows once more in the form the walk steps through (a state is kept as state / bit count, skip and
    // poison as state S with bit count 0 / 1 / 2): state' | (bit count or bits appended) << 8 | the span's events in
    // short << 24 (pack_normal: what scan_emit_kernel writes without simulating) | relative << 30 | 0x80000000;
    // 0 = not a plain result (stuck, bit-count sensitive, position dependent): the full step
    std::vector<uint32_t> rows2;
    for (size_t i = rows0; i < merged.size(); ++i) {
        const uint32_t p = merged[i];
        uint32_t q = 0;
        if (p & kPkAbsolute) {
            const uint32_t code = p & 0xffffu;
            const uint32_t cur = code < SNB ? code / NB1 : S, nb = code < SNB ? code - cur * NB1 : code - SNB;
            if (code < SNB + 3) q = cur | (nb << 8) | 0x80000000u;
        } else if (p & kPkRelative) {
            const uint32_t add = (p >> 8) & 0xffffu;
            q = (p & 0xffu) | ((add > NB1 ? NB1 : add) << 8) | (((p >> kPkEventShift) & 3u) << 24) | 0x40000000u | 0x80000000u;
        }
        rows2.push_back(q);
    }
    merged[3] = (uint32_t)merged.size();
    merged.insert(merged.end(), out.begin(), out.end());
    merged.insert(merged.end(), rows2.begin(), rows2.end());
}

uint32_t fsm_scan_fin_block() { return (uint32_t)kFinBlock; }

hipError_t launch_fsm_scan(const FsmScanArgs &a, hipStream_t stream, hipEvent_t t_end) {
    ScanParams sp{};
    sp.f = a.f;
    sp.block_tab = a.block_tab;
    sp.cap_block_off = a.cap_block_off;
    sp.events = a.events;
    sp.ev_hot = a.ev_hot;
    sp.app_vals = a.app_vals;
    sp.app_capacity = a.app_capacity;
    sp.errs = a.errs;
    sp.err_capacity = a.err_capacity;
    sp.have_first = (a.first || a.first_dev) ? 1 : 0;
    if (a.first) sp.first = *a.first;
    sp.first_dev = a.first_dev;
    sp.has_prev = a.first_dev ? 1u : 0u;
    sp.pos_origin = a.pos_origin;
    sp.totals_in = a.totals_in;
    sp.edge_overflow = a.edge_overflow;
    sp.cap_fallback = a.cap_fallback;
    sp.final_state = a.final_state;
    sp.fallback = a.fallback;
    sp.total_blocks_cap = a.total_blocks_cap;
    sp.leaf_block = a.leaf_block;
    sp.fin_off = a.fin_off;
    sp.fagg = reinterpret_cast<unsigned long long *>(a.fsum);
    sp.fin_ticket = a.fin_ticket;
    sp.run_stamp = a.run_stamp;
    sp.fin_blocks_cap = a.fin_blocks_cap;
    sp.lt_off = a.lt_off;
    sp.lt_n0 = a.lt_n0;
    sp.lt_pk = a.lt_pk;
    sp.ltab = a.ltab;
    // with a concrete incoming state from outside (shards) the first span may leave the closure: no
    // pruning.  The state a chunk of a pipelined run hands to the next is one the whole capture's
    // run passes through: the code behind the chunk's first edge lies in the closure.
    sp.reach = a.first ? nullptr : a.reach;
    sp.nreach = a.first ? 0 : a.nreach;
    sp.nreach_base = a.first ? 0 : a.nreach_base;
    sp.nreach_lv[0] = a.nreach_lv[0];
    sp.nreach_lv[1] = a.nreach_lv[1];
    sp.lt_words = a.lt_words;
    sp.lt_merged = a.lt_merged;
    sp.lt_merged_words = a.lt_merged_words;
    sp.Dp = (a.D + 7u) & ~7u;
    sp.D = a.D;
    sp.S = a.S;
    sp.cap_group_off = a.cap_group_off;
    sp.group_tab = a.group_tab;
    sp.cap_super_off = a.cap_super_off;
    sp.super_tab = a.super_tab;
    sp.super_in = a.super_in;
    sp.sync_rec = a.sync_rec;
    sp.sync_dig = a.sync_rec ? a.sync_rec + (size_t)(a.total_blocks_cap + 8) * kSyncRecWords : nullptr;
    sp.sync_sel = a.sync_rec ? sp.sync_dig + a.total_blocks_cap + 8 : nullptr;
    sp.sync_fail = a.sync_fail;
    sp.sync_try = 0;
    sp.pre_plane = a.pre_plane;
    sp.lt_sync_words = a.lt_sync_words;
    sp.cap_end = a.cap_end;
    sp.cap_first = a.cap_first;
    const size_t lds = block_lds_bytes(a.leaf_block, a.D, a.S, a.SNB);
    const size_t lds_group = groups_lds_bytes(sp.Dp);
    const size_t lds_walk = (size_t)128 * sp.Dp * 2;
    if (lds_walk > 150u * 1024u) return hipErrorInvalidValue;
    hipError_t e;
    e = ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_leaf_kernel), lds);
    if (e != hipSuccess) return e;
    e = ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_walk_kernel), lds_walk);
    if (e != hipSuccess) return e;
    e = ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_groups_kernel), lds_group);
    if (e != hipSuccess) return e;
    const size_t lds_pick = (size_t)kPickPer * kScanThreads * 4;
    e = ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_syncpick_kernel), lds_pick);
    if (e != hipSuccess) return e;
    // with span tables and 64-leaf blocks the leaf kernel runs one wave per block (OOKD_SCAN_LEAF=block: the
    // workgroup-per-block form, which is also what runs without span tables)
    const char *const leaf_env = dev_getenv("OOKD_SCAN_LEAF");       // (looked up per launch: the tests switch it)
    const size_t lds_wave = wave_lds_bytes(a.D, a.S);
    const bool wave_form = a.lt_off && a.lt_words && a.lt_words <= kLtLdsWords && a.leaf_block == 64 && lds_wave <= 60u * 1024u &&
                           !(leaf_env && leaf_env[0] == 'b');
    if (wave_form) {
        e = ensure_dynamic_lds(reinterpret_cast<const void *>(&scan_leaf_wave_kernel), lds_wave);
        if (e != hipSuccess) return e;
    }
    const uint32_t caps = a.f.num_captures;
    const uint32_t cap_grid = caps < 256 ? caps : 256;
    // leaf / emit are persistent grids: exactly as many workgroups as the chip holds at once (a
    // second, thinner round of the 1024 there used to be cost a third of their time)
    static thread_local int sim_dev = -1;
    static thread_local uint32_t leaf_grid = 0, emit_grid = 0, wave_grid = 0;
    static thread_local size_t wave_lds_seen = ~(size_t)0;
    if (wave_form) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev != sim_dev || lds_wave != wave_lds_seen) {
            int per_cu = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, scan_leaf_wave_kernel, 64, lds_wave) == hipSuccess &&
                hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 && per_cu > 0) {
                // four times what the chip holds at once: the hardware's hand-out of workgroups balances the blocks
                wave_grid = 4u * (uint32_t)(per_cu * prop.multiProcessorCount);
            } else {
                (void)hipGetLastError();
                wave_grid = 4 * a.grid_blocks;
            }
            wave_lds_seen = lds_wave;
        }
    }
    static thread_local size_t sim_lds = 0;
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev != sim_dev || lds != sim_lds) {
            int per_cu_leaf = 0, per_cu_emit = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_leaf, scan_leaf_kernel, kSimThreads, lds) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_emit, scan_emit_kernel, kSimThreads, 0) == hipSuccess &&
                hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
                const int cus = prop.multiProcessorCount;
                leaf_grid = per_cu_leaf > 0 ? (uint32_t)(per_cu_leaf * cus) : a.grid_blocks;
                emit_grid = per_cu_emit > 0 ? (uint32_t)(per_cu_emit * cus) : a.grid_blocks;
            } else {
                (void)hipGetLastError();
                leaf_grid = emit_grid = a.grid_blocks;
            }
            sim_dev = dev;
            sim_lds = lds;
        }
    }
    static const char *const grid_env = dev_getenv("OOKD_SCAN_GRID");
    const uint32_t leaf_blocks = grid_env ? (uint32_t)atoi(grid_env) : leaf_grid;
    const uint32_t emit_blocks = grid_env ? (uint32_t)atoi(grid_env) : emit_grid;
    sp.skipc = wave_form ? a.skipc : nullptr;
    sp.skipc_valid = sp.skipc ? 1u : 0u;
    sp.pre_codes = a.pre_codes;
    sp.blk_in = a.blk_in;
    sp.rowz = a.rowz;
    // entry codes from synchronising spans first; the composing kernels behind them return at once unless that gave up
    sp.sync_try = (a.sync_try && wave_form && a.lt_merged && a.lt_sync_words && a.lt_sync_words <= kMergedLdsWords && a.rowz && a.skipc &&
                   a.pre_codes && a.pre_plane && a.sync_rec && a.sync_fail) ? 1u : 0u;
    sp.sync_only = sp.sync_try && a.sync_try == 2 ? 1u : 0u;
    hipLaunchKernelGGL(scan_layout_kernel, dim3(1), dim3(kScanThreads), 0, stream, sp);
    if (sp.sync_try) {
        hipLaunchKernelGGL(scan_sync_kernel, dim3(512), dim3(256), 0, stream, sp);
        hipLaunchKernelGGL(scan_syncwalk_kernel, dim3(512), dim3(256), 0, stream, sp);
        hipLaunchKernelGGL(scan_syncpick_kernel, dim3(cap_grid), dim3(kScanThreads), lds_pick, stream, sp);
    }
    // the composing kernels (with a walk in front of them they return at once unless it gave up; after a walk in
    // sync_only form they are not queued at all: five launches less on the chain, ~25 us)
    if (!(sp.sync_try && sp.sync_only)) {
        if (wave_form) {
            hipLaunchKernelGGL(scan_leaf_wave_kernel, dim3(grid_env ? (uint32_t)atoi(grid_env) : wave_grid), dim3(64), lds_wave, stream, sp);
        } else {
            hipLaunchKernelGGL(scan_leaf_kernel, dim3(leaf_blocks), dim3(kSimThreads), lds, stream, sp);
        }
        hipLaunchKernelGGL(scan_groups_kernel, dim3(512), dim3(kGroupsThreads), lds_group, stream, sp);
        hipLaunchKernelGGL(scan_walk_kernel, dim3(cap_grid), dim3(kScanThreads), lds_walk, stream, sp);
        // entry codes of all blocks, then of all leaves, in two small passes; the emit kernel stages no tables
        sp.entry_phase = 0;
        const bool with_phase2 = a.lt_off && a.lt_merged && a.rowz;
        hipLaunchKernelGGL(scan_entry_kernel, dim3(kEntryP0Blocks + (with_phase2 ? 512u : 0u)), dim3(256), 0, stream, sp);
        sp.entry_phase = 1;
        hipLaunchKernelGGL(scan_entry_kernel, dim3(512), dim3(256), 0, stream, sp);
    }
    hipLaunchKernelGGL(scan_emit_kernel, dim3(emit_blocks), dim3(kSimThreads), 0, stream, sp);
    // the finish workgroups wait for each other: no more of them than fit the chip at once
    hipLaunchKernelGGL(fin_write_kernel, dim3(256), dim3(kFinBlock), 0, stream, sp);
    // t_end takes the last kernel's own end time stamp (no marker packet behind the chain)
    hipExtLaunchKernelGGL(fin_msg_kernel, dim3(64), dim3(256), 0, stream, nullptr, t_end, 0, sp, a.publish);
    return hipGetLastError();
}

}  // namespace ookd
