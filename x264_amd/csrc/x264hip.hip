// x264hip.hip -- device context and C ABI (include/x264hip.h) of the gfx950 lookahead path.
//
// Data layout in HBM, per frame slot (all resident for the life of the frame in the lookahead window):
//   planes      4 x (lines+64) x stride pixels   half-res luma: full, H, V, HV half-pel planes incl. 32 px border
//   luma        height x width pixels            staging copy of the input luma (source of lowres + AQ)
//   inv_qscale  n_mb u16                         Q8 AQ factors (input of the cost_est_aq / row satd sums)
//   mvq[l][d]   n_mb u64 granules {mv, tag}      lowres_mvs[l][d] of the reference + in-kernel hand-off tag
//   mvcost[l][d] n_mb i32                        lowres_mv_costs[l][d]
//   lowres_costs[d0][d1] n_mb u16                cell maps; [0][0] doubles as i_intra_cost (frame.c:283)
//   row_satds[d0][d1] mb_h i32
// plus per-slot weighted copies of a reference's plane 0, taken from a small pool.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <map>

#include "x264hip.h"
#include "device_common.h"
#include "me_search.h"
#include "me_latency.h"
#include "la_kernels.h"
#include "block_metrics.h"
#include "dct_quant_block.h"
#include "me_full.h"
#include "integral.h"
#include "vtable_blocks.h"

#define HIPCK( call )                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = ( call );                                                                            \
        if( e_ != hipSuccess )                                                                               \
        {                                                                                                    \
            fprintf( stderr, "x264hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
            ctx->broken = 1;                                                                                 \
            return X264HIP_EDEVICE;                                                                          \
        }                                                                                                    \
    } while( 0 )

// one cell summary travelling between ranks (x264hip_export_cells / x264hip_import_cells): where its parts live on this side
struct CellXfer
{
    int *acc_host;            // import: pinned record x264hip_frame_cost reads; export: unused
    int *acc_dev;             // the five sums on the device
    int *rows, *rows_intra;   // [mb_h] each
    int skip, pad_;           // import: this context already has the cell
};

struct CellEntry
{
    unsigned char valid = 0;      // a speculative result for this cell sits in cell_acc
    unsigned char requested = 0;  // the caller has asked for this cell: never speculate it again
    unsigned char variant = 1;    // B cells: evaluated with (1) or without (0) the list-1 reference's own L0 vectors
    unsigned tag0 = 0, tag1 = 0, tagr = 0; // tags of the fields the speculative evaluation consumed
    unsigned batch = 0;
    // window shard (include/x264hip.h "one lookahead window over several GPUs"): the sums came from the rank that owns the frame,
    // the per-block map (lowres_costs) is not in this context; the frames the cell refers to, for a local re-evaluation
    unsigned char map_remote = 0;
    int slot_p0 = -1, slot_p1 = -1;
};

struct FrameSlot
{
    int in_use = 0;
    int frame_no = -1;
    int rowmajor_mask = 0;        // bit p: plane p exists row-major (plane 0 always; H / V / HV on demand, x264hip_get_lowres)
    uint64_t mbt_last_use = 0;    // serial of the last MB-tree launch whose steps name this slot (0: none); mbt_queued: named by steps still in the queue
    bool mbt_queued = false;
    char *planes = nullptr;       // padded planes base (4 planes)
    char *luma = nullptr;
    uint16_t *inv_qscale = nullptr;
    unsigned long long *frame_sums = nullptr;
    uint2 *mb_sums = nullptr;
    unsigned long long *mvq[3][X264HIP_BFRAME_MAX + 1];   // [2]: a second list-0 field per distance, searched speculatively on a WEIGHTED
    int *mvcost[3][X264HIP_BFRAME_MAX + 1];                 // reference (x264hip_prefetch_weighted_fields); a request that brings that weight swaps it in
    uint16_t *lowres_costs = nullptr; // [(bf+2)*(bf+2)][n_mb]
    int *row_satds = nullptr;         // [(bf+2)*(bf+2)][mb_h]
    int *blk = nullptr;               // [(bf+2)*(bf+2)][n_mb] unclamped block cost | b_intra << 30 per cell
    int *prop = nullptr;              // [n_mb] MB-tree i_propagate_cost accumulator
    int *prop_view = nullptr;         // where the accumulator's current contents are: prop, or this slot's array in one of the context's banks
    float *qp_aq = nullptr, *qp = nullptr; // [n_mb] f_qp_offset_aq, f_qp_offset
    unsigned field_tag[2][X264HIP_BFRAME_MAX + 1]; // serial of the search that last wrote the field
    std::vector<CellEntry> cells;
    // Callers ask for some B cell classes now with, now without the list-1 reference's own vectors (slicetype.c:629: it depends on
    // whether that frame has been searched as a P frame by then).  When both happen often the minority variant is speculated as well,
    // into the slot's one spare cell; a request for it copies map and sums over (no evaluation, no wait for an on-demand launch).
    std::vector<CellEntry> alts;      // [n_cells]: the speculative OTHER variant of B cell idx, evaluated into spare cell ctx->spare_at[idx]
    // a B cell's THIRD evaluation (x264hip_prefetch_weighted_fields): over the speculative weighted list-0 field of this frame and / or of
    // its list-1 reference, into the cell's second spare (ctx->spare2_at)
    std::vector<CellEntry> alts2;
    std::vector<int> cell_at;         // [n_cells]: where the data of cell idx lives -- idx, or spare_at[idx] once the caller asked for the variant in the spare
    // host-side state of the device fields
    unsigned char field_ready[2][X264HIP_BFRAME_MAX + 1]; // searched (any variant) and complete on the stream
    unsigned char field_prefetched[2][X264HIP_BFRAME_MAX + 1]; // unweighted field computed speculatively, not yet claimed
    // window shard: the field was searched on the rank that owns this frame -- 1: nothing of it is here (the tag stands for it),
    // 2: its vectors are here (x264hip_import_cell_map), its costs are not.  0: an ordinary local field
    unsigned char field_remote[2][X264HIP_BFRAME_MAX + 1];
    int *cell_sums = nullptr;         // [n_store][8] device copy of the cell sums (x264hip_export_cells)
    int *cell_work = nullptr;         // [n_store][8] work words of cell_reduce_kernel
    // speculation by position (x264hip_gop_hint): the ( period, position ) this frame was speculated under, and what it was asked for
    int pos_key = 0;                  // period * 32 + position, 0 = no expectation
    unsigned req_fields[2] = { 0, 0 };// bit d: (list, distance d + 1) requested
    std::vector<unsigned char> req_cells; // [(bf+2)*(bf+2)]
    // the speculative weighted list-0 field of distance d + 1, if any: the weight it was searched with, its tag and batch; its P cell sits
    // in the spare of cell ( d + 1, 0 ) (alts)
    struct WSpec { unsigned char valid = 0; x264hip_weight w = { 0, 1, 0, 0 }; unsigned tag = 0, batch = 0; };
    WSpec wspec[X264HIP_BFRAME_MAX + 1];
    int wplane_idx = -1;          // weighted-plane pool entry in use by this slot's current weighted search
    uint64_t sum = 0, ssd = 0;
    int stats_valid = 0;
    unsigned gen = 0;             // bumped whenever the slot receives a new frame
};

// Pinned-host / device descriptor tables are handed out round-robin; an entry is reused only after the event
// recorded behind its last consumer has completed, so filling the next table never waits for the stream.
struct DescRing
{
    static const int K = 4;
    char *host[K] = { nullptr, nullptr, nullptr, nullptr }, *dev[K] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev[K] = { nullptr, nullptr, nullptr, nullptr };
    bool used[K] = { false, false, false, false };
    int next = 0;
};

struct x264hip_ctx
{
    x264hip_params p;
    LaP P;
    int device = 0;
    int n_cu = 256;
    int broken = 0;
    int psz = 1;                  // sizeof(pixel)
    int lw = 0, lh = 0;           // lowres dims (mod16 / 2)
    int n_mb = 0;
    size_t plane_bytes = 0;
    hipStream_t stream = nullptr;
    std::vector<FrameSlot> slots;
    uint16_t *cost_mv_dev = nullptr; // base (not centred)
    AqLuts *luts_dev = nullptr;
    unsigned *sync_words = nullptr;  // device: row-ticket counters of the search kernel
    unsigned long long *me_prof = nullptr; // ME_PROFILE builds: 8 cycle accumulators of the search kernel (device), else unused
    int n_cells = 0;                 // (bframes+2)^2
    std::vector<int> spare_at;       // [n_cells]: storage index of the spare of B cell idx (n_cells + its rank among the B classes; only they have one)
    int n_store = 0;                 // cell maps per slot: n_cells own places + one spare per B class + one nobody reads (spare_at of the other classes)
    int *cell_acc_host = nullptr;    // pinned [slots][n_cells][8]: sums of every cell evaluation, written by the device directly
    int *cell_alt_host = nullptr;    // pinned [slots][n_cells][8]: sums of the spare cells (FrameSlot alts)
    int *cell_alt2_host = nullptr;   // pinned [slots][n_cells][8]: sums of the second spares (FrameSlot alts2)
    std::vector<int> spare2_at;      // [n_cells]: storage index of the second spare of B cell idx
    DescRing cell_ring, put_ring, search_ring, xfer_ring;
    int xfer_cap = 2048;
    unsigned *err_host = nullptr;    // pinned: sticky in-kernel timeout flag, written by the device directly
    int launches_since_big = 1000;   // search launches since the last one that could fill the chip (launch_searches_t: which form a small one takes)
    int put_desc_cap = 256;
    int cell_desc_cap = 0;
    unsigned batch_serial = 0, batch_synced = 0;
    static const int BATCH_EVS = 64;
    hipEvent_t batch_ev[64] = { nullptr }; // batch_ev[b % 64] is recorded behind speculative batch b
    unsigned long long *stats_host = nullptr; // pinned [slots][2]
    // MB-tree: own stream, ring of pinned/device step tables
    hipStream_t stream2 = nullptr;
    static const int MBT_RING = 16, MBT_CAP = 6144;
    MbtOpDev *mbt_host[32] = { nullptr }, *mbt_dev[32] = { nullptr };
    hipEvent_t mbt_done[32] = { nullptr };
    hipEvent_t ev_cross = nullptr, ev_mbt_last = nullptr;
    hipEvent_t ev_ingest = nullptr;  // behind the most recent ingest kernels: frame totals are readable after it
    int mbt_next = 0, mbt_pending = 0;
    uint64_t mbt_serial = 0, mbt_ring_serial[16] = { 0 }; // launches so far; the launch that last used ring entry r (its event: mbt_done[r])
    unsigned *mbt_bar = nullptr;      // device [MBT_RING][MBT_MAX_GROUPS][4]: barrier arrivals, unused, exits, unused; then one error word
    // step lists waiting for their launch (x264hip_mbtree queues, mbt_flush launches): list g of the queue is steps
    // [mbt_q.beg[g], mbt_q.beg[g+1]) of ring entry mbt_q_ring and adds into accumulator bank g
    MbtGroups mbt_q = { 0, { 0 } };
    int mbt_q_ring = -1;
    bool counted_open = false;
    // descriptor tables go to the device through a stream of their own (upload_async)
    hipStream_t stream_up = nullptr;
    static const int UP_EVS = 64;
    hipEvent_t up_ev[64] = { nullptr };
    int up_next = 0;
    std::vector<std::pair<int, int>> mbt_q_finished;  // (slot, step index in the ring entry) of the FINISH steps of the queued lists
    std::map<int, std::pair<uint64_t, uint64_t>> flush_sites; // source line of an mbt_flush_at call -> (flushes of a non-empty queue, lists launched)
    int *prop_bank[MBT_MAX_GROUPS] = { nullptr }; // bank g > 0: [max_frames][n_mb] accumulators of list g of a launch (bank 0 = the slots' own)
    int mbt_bank_limit = MBT_MAX_GROUPS;          // lists per launch this context has memory for (lowered when a bank cannot be allocated)
    int desc_cap = 0;
    // weight costs: WCAP job entries, each with device counters [2][2] and a pinned result pair; entry 0 serves the
    // on-demand call, the others hold speculative pairs (x264hip_prefetch_weight_costs) until their frames go away
    static const int WCAP = 1024;
    unsigned *wcost_dev = nullptr;   // device [WCAP][2][2]
    unsigned *wcost_host = nullptr;  // pinned [WCAP][2]
    DescRing wjob_ring;
    struct WEntry { int slot_fenc = -1, slot_ref = -1; unsigned gen_fenc = 0, gen_ref = 0; x264hip_weight w; unsigned batch = 0; };
    std::vector<WEntry> wcache;      // [WCAP], entry 0 unused
    int wcache_next = 1;
    char *staging = nullptr;         // pinned luma staging (entry 0 of the ring below)
    // pictures handed over in HOST memory (x264hip_frame_put with is_device == 0, host pointers in x264hip_frame_put_batch*): a DMA
    // stream of their own -- pinned buffers go straight from where they are, pageable ones through a ring of pinned staging copies --
    // and the compute stream waits for an event behind the copies, never the host for the compute stream
    static const int STAGE_RING = 4;
    hipStream_t stream_h2d = nullptr;
    hipEvent_t h2d_done = nullptr;
    char *stage[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t stage_ev[4] = { nullptr, nullptr, nullptr, nullptr };
    bool stage_used[4] = { false, false, false, false };
    int stage_next = 0;
    uint64_t weighted_speculated = 0, weighted_claimed = 0, weighted_cells = 0, weighted_cells_used = 0; // x264hip_prefetch_weighted_fields: searches enqueued, fields a request took, P cells with them
    uint64_t h2d_bytes = 0, h2d_direct = 0, h2d_staged = 0; // bytes copied, pictures taken from pinned memory as they were / through the ring
    // Pictures that follow each other in pinned host memory cross PCIe as ONE transfer per group of a batch (a 2 MB copy reaches 21 GB/s,
    // a 33 MB one the link's 50+, profiles/r06_bench_kernel_stats_solo.csv): into a ring of device group buffers the ingest kernels read
    // in place of the slots' own luma buffers.  h2d_group_free[k]: behind the ingest kernels that last read buffer k (compute stream).
    // Twelve buffers of at most sixteen pictures or 64 MB, allocated as they are first needed: a pass of 160 pictures finds a free
    // buffer for every group.  A buffer whose last readers have not run yet is waited for by the HOST THREAD, outside the queue's lock:
    // a device-side wait in the device's ONE transfer queue holds up every other context's pictures behind it (six buffers, waits in
    // the queue: the link was busy 59 % of a host-fed step, gpurun_out/r06w).
    static const int H2D_GROUPS = 12, H2D_GROUP_PICS = 16;
    char *h2d_group[12] = {};
    hipEvent_t h2d_group_free[12] = {};
    bool h2d_group_used[12] = {};
    int h2d_group_next = 0, h2d_group_pics = 16;
    // X264HIP_H2D_TRACE=1: host time of the host-picture calls -- waiting for the transfer queue's lock, holding it, between two calls
    uint64_t h2d_t_wait = 0, h2d_t_hold = 0, h2d_t_prep = 0, h2d_t_between = 0, h2d_t_last = 0, h2d_calls = 0;
    uint64_t h2d_merged = 0, h2d_by_kernel = 0; // transfers that carried a whole group; single pictures fetched by a copy kernel on the compute stream
    char *chroma_staging = nullptr, *chroma_dev = nullptr; // host-buffer ingest with chroma: pinned + device copies of Cb and Cr (allocated on first use)
    size_t staging_bytes = 0;
    std::vector<char *> wplanes;     // weighted plane pool
    char *me_pool = nullptr;         // device memory of x264hip_me_search_batch calls (request table, results, TESA lists), grow-only
    size_t me_pool_bytes = 0;
    char *me_host = nullptr;         // ... and the pinned host block its tables are written to and its results come back to
    size_t me_host_bytes = 0;
    char *vt_pool = nullptr;         // staging memory of the function-table members (x264hip_mc_fill & co.), grow-only
    size_t vt_pool_bytes = 0;
    std::vector<int> wplane_owner;
    unsigned tag_serial = 1;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    int last_n_search = 0, last_n_blocks = 0, ev_valid = 0;
    // profile ring: one event pair per search launch while profiling is on
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_n;
    int prof_on = 0, prof_used = 0;
    double prof_ms = 0; uint64_t prof_launches = 0, prof_searches = 0;
    double prof_lat_ms = 0; uint64_t prof_lat_launches = 0, prof_lat_searches = 0; // the search launches that went to me_latency_kernel, for themselves
    double prof_cell_ms = 0; uint64_t prof_cell_launches = 0, prof_cells = 0; // the same for the cost cell launches (prof_n entries < 0)
    // prof_on & 2: an event pair around every kernel of the ingest and cell launches as well (x264hip_kernel_profile; class = X264HIP_KPROF_*)
    std::vector<int> prof_kind;       // parallel to prof_n: -1 = a search / cell launch as above, else the kernel class
    double kprof_ms[X264HIP_KPROF_CLASSES] = { 0 }; uint64_t kprof_launches[X264HIP_KPROF_CLASSES] = { 0 }, kprof_units[X264HIP_KPROF_CLASSES] = { 0 };
    uint64_t counters[16] = { 0 }; // [8] remote fields searched here after all [9] remote cell maps recomputed here [10] maps imported [11] cells imported [12] fields registered as remote [13] searches on demand (x264hip_frame_cost) [14] second variants of B cells speculated [15] ... and used
    // how often callers asked for each B cell class with / without a searched L0 field of the list-1 reference
    // (slicetype.c:629-642): speculation evaluates the variant asked for more often so far
    uint32_t variant_req[( X264HIP_BFRAME_MAX + 2 ) * ( X264HIP_BFRAME_MAX + 2 )][2] = { { 0 } };
    // Which field classes (list, distance) and cell classes (d0, d1) the caller's decision flow ever asks for depends on
    // its configuration (b-adapt mode, pyramid, bframes): after a learning period classes that were never requested are
    // no longer speculated (a later request is still served, on demand).
    uint32_t field_req[2][X264HIP_BFRAME_MAX + 1] = { { 0 } };
    uint32_t cell_req[( X264HIP_BFRAME_MAX + 2 ) * ( X264HIP_BFRAME_MAX + 2 )] = { 0 };
    uint32_t field_spec[2][X264HIP_BFRAME_MAX + 1] = { { 0 } };  // speculative searches / cells per class (X264HIP_TRACE_CLASSES prints them next to the requests)
    uint32_t cell_spec[( X264HIP_BFRAME_MAX + 2 ) * ( X264HIP_BFRAME_MAX + 2 )] = { 0 };
    uint32_t n_requests = 0;
    bool mbt_q_lds = false;           // the lists queued for the next MB-tree launch are in the one-workgroup LDS form (mbtree_lds_kernel)
    static const uint32_t LEARN_REQUESTS = 400;
    // ... and what the caller can say ahead of time about its flow (x264hip_spec_classes): classes outside these are never speculated
    uint8_t cell_allowed[( X264HIP_BFRAME_MAX + 2 ) * ( X264HIP_BFRAME_MAX + 2 )];
    x264hip_ctx() { memset( cell_allowed, 1, sizeof( cell_allowed ) ); }
    uint32_t field_allowed[2] = { ~0u, ~0u };
    // speculation by position: hint of the caller, and per ( period, position ) key the number of frames that have come and gone and
    // how many of them asked for each field / cell class
    int hint_anchor = 0, hint_period = 0;
    static const int POS_KEYS = 32 * ( X264HIP_BFRAME_MAX + 2 ), POS_LEARN_FRAMES = 24;
    std::vector<uint32_t> pos_frames;             // [POS_KEYS]
    std::vector<uint32_t> pos_field_req;          // [POS_KEYS][2][BFRAME_MAX + 1]
    std::vector<uint32_t> pos_cell_req;           // [POS_KEYS][n_cells]
};

static const char *const g_errstr[] = { "ok", "no usable HIP device", "invalid argument", "out of memory", "device failure",
                                        "in-kernel wait timed out", "bad call sequence", "another rank of the window shard failed" };
extern "C" const char *x264hip_strerror( int code )
{
    int i = -code;
    return i >= 0 && i < 8 ? g_errstr[i] : "unknown";
}

template <typename T>
static inline T *plane_origin( x264hip_ctx *ctx, FrameSlot &s, int p )
{
    return (T *)( s.planes + (size_t)p * ctx->plane_bytes ) + LA_PAD * ctx->P.stride + LA_PAD;
}

// Table in pinned host memory (16-byte aligned, a multiple of 4 bytes) -> device memory, in front of the launch on stream s that reads
// it: a kernel on s (la_kernels.h: upload_kernel).  Measured (rocprofv3 --hip-runtime-trace, eight contexts, 1080p): hipMemcpyAsync on s
// does not return while s has an unresolved hipStreamWaitEvent in front of it -- 12-41 ms per call, ~28 ms per 160-frame pass of a
// context, 20 700 frames/s; the kernel 21 400-21 900; hipMemcpyAsync on a stream of its own that never waits, s waiting for the copy's
// event (X264HIP_UPLOAD=stream), 18 300; the kernel on a high-priority stream of its own (X264HIP_UPLOAD=prio) 21 700, but 15 700 against 21 400
// with four contexts.
static hipError_t upload_async( x264hip_ctx *ctx, void *dst_dev, const void *src_pinned, size_t bytes, hipStream_t s )
{
    if( !bytes ) return hipSuccess;
    static const int mode = !getenv( "X264HIP_UPLOAD" ) ? 0 : !strcmp( getenv( "X264HIP_UPLOAD" ), "stream" ) ? 1 : !strcmp( getenv( "X264HIP_UPLOAD" ), "prio" ) ? 2 : 0;
    const unsigned n_words = (unsigned)( ( bytes + 3 ) / 4 );
    const unsigned wgs = std::max( 1u, std::min( 64u, ( n_words / 4 + 255 ) / 256 ) );
    if( mode == 0 )
    {
        upload_kernel<<<wgs, 256, 0, s>>>( (uint32_t *)dst_dev, (const uint32_t *)src_pinned, n_words );
        return hipGetLastError();
    }
    // (the previous reader of the destination is done: ring_acquire / mbt_ring_acquire)
    hipEvent_t ev = ctx->up_ev[ctx->up_next];
    ctx->up_next = ( ctx->up_next + 1 ) % x264hip_ctx::UP_EVS;
    hipError_t e = hipSuccess;
    if( mode == 1 )
        e = hipMemcpyAsync( dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream_up );
    else
    {
        upload_kernel<<<wgs, 256, 0, ctx->stream_up>>>( (uint32_t *)dst_dev, (const uint32_t *)src_pinned, n_words );
        e = hipGetLastError();
    }
    if( e == hipSuccess ) e = hipEventRecord( ev, ctx->stream_up );
    if( e == hipSuccess ) e = hipStreamWaitEvent( s, ev, 0 );
    return e;
}

static void ring_free( DescRing &r )
{
    for( int i = 0; i < DescRing::K; i++ )
    {
        (void)hipHostFree( r.host[i] ); (void)hipFree( r.dev[i] );
        if( r.ev[i] ) (void)hipEventDestroy( r.ev[i] );
        r.host[i] = r.dev[i] = nullptr; r.ev[i] = nullptr;
    }
}
static hipError_t ring_alloc( DescRing &r, size_t bytes )
{
    for( int i = 0; i < DescRing::K; i++ )
    {
        hipError_t e = hipHostMalloc( &r.host[i], bytes );
        if( e == hipSuccess ) e = hipMalloc( &r.dev[i], bytes );
        if( e == hipSuccess ) e = hipEventCreateWithFlags( &r.ev[i], hipEventDisableTiming );
        if( e != hipSuccess ) return e;
    }
    return hipSuccess;
}
// next table pair; blocks only if the device is still K tables behind
static int ring_acquire( DescRing &r, int *idx )
{
    const int i = r.next;
    r.next = ( i + 1 ) % DescRing::K;
    if( r.used[i] && hipEventSynchronize( r.ev[i] ) != hipSuccess ) return -1;
    *idx = i;
    return 0;
}
static int ring_commit( DescRing &r, int i, hipStream_t s )
{
    r.used[i] = true;
    return hipEventRecord( r.ev[i], s ) == hipSuccess ? 0 : -1;
}

// contexts open on each device: the MB-tree launches size themselves so that the lists of every context fit the chip together
#include <atomic>
static std::atomic<int> g_open_contexts[64];
// ONE host-to-device stream per device, shared by its contexts.  With a DMA stream per context the transfers of eight contexts share the
// link: every context's pictures arrive late and together, all of them compute together, and link and device take turns (eight host-fed
// segments: 88 ms per step = 46 ms of transfers + 43 ms of device work, gpurun_out/r06r).  In ONE queue the transfers of a context arrive
// at the link's full rate, its kernels start while the next context's pictures are on their way, and the contexts fall out of step.
#include <mutex>
static std::mutex g_h2d_mutex;
static std::mutex g_h2d_batch_mutex[64]; // a batch of host pictures is enqueued as a whole: the transfers of ONE context follow each other in the device's queue
static hipStream_t g_h2d_stream[64];
static int g_h2d_users[64];
static int h2d_stream_acquire( int device, hipStream_t *out )
{
    std::lock_guard<std::mutex> lock( g_h2d_mutex );
    const int d = device & 63;
    if( !g_h2d_users[d] && hipStreamCreateWithFlags( &g_h2d_stream[d], hipStreamNonBlocking ) != hipSuccess ) return X264HIP_EDEVICE;
    g_h2d_users[d]++;
    *out = g_h2d_stream[d];
    return X264HIP_OK;
}
static void h2d_stream_release( int device )
{
    std::lock_guard<std::mutex> lock( g_h2d_mutex );
    const int d = device & 63;
    if( g_h2d_users[d] > 0 && !--g_h2d_users[d] )
    {
        (void)hipStreamDestroy( g_h2d_stream[d] );
        g_h2d_stream[d] = nullptr;
    }
}

static inline uint64_t host_now_ns()
{
    struct timespec ts;
    clock_gettime( CLOCK_MONOTONIC, &ts );
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
static void free_all( x264hip_ctx *ctx )
{
    if( ctx->h2d_calls && getenv( "X264HIP_H2D_TRACE" ) )
        fprintf( stderr, "x264hip h2d trace: %llu calls, per call: before the lock %.3f ms, waiting for it %.3f ms, holding it %.3f ms, between calls %.3f ms\n",
                 (unsigned long long)ctx->h2d_calls, ctx->h2d_t_prep / 1e6 / ctx->h2d_calls, ctx->h2d_t_wait / 1e6 / ctx->h2d_calls,
                 ctx->h2d_t_hold / 1e6 / ctx->h2d_calls, ctx->h2d_t_between / 1e6 / ctx->h2d_calls );
    if( ctx->counted_open ) { g_open_contexts[ctx->device & 63]--; ctx->counted_open = false; }
    if( ctx->stream ) (void)hipStreamSynchronize( ctx->stream );
    for( auto &s : ctx->slots )
    {
        (void)hipFree( s.planes ); // one allocation per slot holds everything
    }
    for( auto w : ctx->wplanes ) (void)hipFree( w );
    if( ctx->vt_pool ) (void)hipFree( ctx->vt_pool );
    if( ctx->me_pool ) (void)hipFree( ctx->me_pool );
    if( ctx->me_host ) (void)hipHostFree( ctx->me_host );
    (void)hipFree( ctx->cost_mv_dev ); (void)hipFree( ctx->luts_dev ); (void)hipFree( ctx->sync_words );
    ring_free( ctx->cell_ring ); ring_free( ctx->put_ring ); ring_free( ctx->search_ring ); ring_free( ctx->wjob_ring ); ring_free( ctx->xfer_ring );
    (void)hipHostFree( ctx->err_host );
    (void)hipHostFree( ctx->stats_host );
    if( ctx->stream_up ) (void)hipStreamSynchronize( ctx->stream_up );
    if( ctx->stream2 ) (void)hipStreamSynchronize( ctx->stream2 );
    for( int i = 0; i < x264hip_ctx::MBT_RING; i++ )
    {
        (void)hipHostFree( ctx->mbt_host[i] ); (void)hipFree( ctx->mbt_dev[i] );
        if( ctx->mbt_done[i] ) (void)hipEventDestroy( ctx->mbt_done[i] );
    }
    (void)hipFree( ctx->mbt_bar );
    for( int g = 1; g < MBT_MAX_GROUPS; g++ ) (void)hipFree( ctx->prop_bank[g] );
    if( ctx->ev_cross ) (void)hipEventDestroy( ctx->ev_cross );
    if( ctx->ev_mbt_last ) (void)hipEventDestroy( ctx->ev_mbt_last );
    if( ctx->ev_ingest ) (void)hipEventDestroy( ctx->ev_ingest );
    for( int i = 0; i < x264hip_ctx::BATCH_EVS; i++ )
        if( ctx->batch_ev[i] ) (void)hipEventDestroy( ctx->batch_ev[i] );
    if( ctx->stream2 ) (void)hipStreamDestroy( ctx->stream2 );
    for( int i = 0; i < x264hip_ctx::UP_EVS; i++ )
        if( ctx->up_ev[i] ) (void)hipEventDestroy( ctx->up_ev[i] );
    if( ctx->stream_up ) (void)hipStreamDestroy( ctx->stream_up );
    (void)hipHostFree( ctx->cell_acc_host );
    (void)hipHostFree( ctx->cell_alt_host ); (void)hipHostFree( ctx->cell_alt2_host );
    (void)hipFree( ctx->wcost_dev );
    (void)hipHostFree( ctx->wcost_host ); (void)hipHostFree( ctx->staging );
    for( int k = 1; k < x264hip_ctx::STAGE_RING; k++ ) if( ctx->stage[k] ) (void)hipHostFree( ctx->stage[k] );
    for( int k = 0; k < x264hip_ctx::STAGE_RING; k++ ) if( ctx->stage_ev[k] ) (void)hipEventDestroy( ctx->stage_ev[k] );
    if( ctx->h2d_done ) (void)hipEventDestroy( ctx->h2d_done );
    for( int k = 0; k < x264hip_ctx::H2D_GROUPS; k++ )
    {
        if( ctx->h2d_group[k] ) (void)hipFree( ctx->h2d_group[k] );
        if( ctx->h2d_group_free[k] ) (void)hipEventDestroy( ctx->h2d_group_free[k] );
    }
    if( ctx->stream_h2d )
    {
        (void)hipStreamSynchronize( ctx->stream_h2d ); // (shared with the device's other contexts: nothing of THIS context may still be queued on it)
        h2d_stream_release( ctx->device );
    }
    if( ctx->chroma_staging ) (void)hipHostFree( ctx->chroma_staging );
    if( ctx->chroma_dev ) (void)hipFree( ctx->chroma_dev );
    for( auto e : ctx->prof_ev ) (void)hipEventDestroy( e );
    if( ctx->ev_start ) (void)hipEventDestroy( ctx->ev_start );
    if( ctx->ev_stop ) (void)hipEventDestroy( ctx->ev_stop );
    if( ctx->stream ) (void)hipStreamDestroy( ctx->stream );
}

extern "C" void x264hip_mc_unbind( x264hip_ctx *ctx );
extern "C" void x264hip_close( x264hip_ctx *ctx )
{
    if( !ctx ) return;
    x264hip_mc_unbind( ctx );
    if( getenv( "X264HIP_TRACE_CLASSES" ) )
    {
        const int bf = ctx->p.bframes, ns = bf + 2;
        fprintf( stderr, "x264hip classes (requested / speculated): fields" );
        for( int l = 0; l < 2; l++ )
            for( int d = 0; d <= bf; d++ )
                fprintf( stderr, " L%d d%d %u/%u", l, d + 1, ctx->field_req[l][d], ctx->field_spec[l][d] );
        fprintf( stderr, " | cells" );
        for( int d0 = 0; d0 <= bf + 1; d0++ )
            for( int d1 = 0; d0 + d1 <= bf + 1; d1++ )
                fprintf( stderr, " (%d,%d) %u/%u", d0, d1, ctx->cell_req[d0 * ns + d1], ctx->cell_spec[d0 * ns + d1] );
        fprintf( stderr, "\n" );
        fprintf( stderr, "x264hip MB-tree queue flushed at (source line: flushes, lists launched):" );
        for( auto &kv : ctx->flush_sites )
            fprintf( stderr, " %d: %llu, %llu;", kv.first, (unsigned long long)kv.second.first, (unsigned long long)kv.second.second );
        fprintf( stderr, "\n" );
    }
#ifdef ME_PROFILE
    if( ctx->me_prof )
    {
        unsigned long long v[32] = { 0 };
        (void)hipStreamSynchronize( ctx->stream );
        (void)hipMemcpy( v, ctx->me_prof, sizeof( v ), hipMemcpyDeviceToHost );
        if( v[7] )
            fprintf( stderr, "ME_PROFILE waves %llu steps/wave %.1f cycles/wave %.0f | per step: wait-below %.0f pre %.0f search %.0f store+rest %.0f | spins/step %.3f\n", v[7],
                     (double)v[6] / v[7], (double)v[0] / v[7], (double)v[1] / v[6], (double)v[2] / v[6], (double)v[3] / v[6], (double)v[4] / v[6], (double)v[5] / v[6] );
        if( v[7] )
            fprintf( stderr, "ME_PROFILE search phases per step (group 0): start candidates %.0f pattern %.0f half-pel %.0f quarter-pel %.0f\n",
                     (double)v[8] / v[6], (double)v[9] / v[6], (double)v[10] / v[6], (double)v[11] / v[6] );
        if( v[26] )
            fprintf( stderr, "ME_PROFILE latency form: steps searched on a guessed below-left vector %.3f of all, guesses that missed %.3f\n", (double)v[26] / v[6], (double)v[27] / v[26] );
        if( v[7] && v[23] )
        {
            fprintf( stderr, "ME_PROFILE neighbour candidates kept per block, share of blocks (0..4):" );
            unsigned long long tb = v[12] + v[13] + v[14] + v[15] + v[16], tw = v[17] + v[18] + v[19] + v[20] + v[21];
            for( int i = 0; i < 5; i++ ) fprintf( stderr, " %.3f", (double)v[12 + i] / ( tb ? tb : 1 ) );
            fprintf( stderr, " | largest of a wave's eight blocks:" );
            for( int i = 0; i < 5; i++ ) fprintf( stderr, " %.3f", (double)v[17 + i] / ( tw ? tw : 1 ) );
            fprintf( stderr, " | single-tap share of the half-pel set's candidates %.3f, of the start set's %.3f\n", (double)v[22] / ( v[23] ? v[23] : 1 ), (double)v[24] / ( v[25] ? v[25] : 1 ) );
        }
        (void)hipFree( ctx->me_prof );
    }
#endif
    free_all( ctx );
    delete ctx;
}

static size_t align_up( size_t v, size_t a ) { return ( v + a - 1 ) / a * a; }

extern "C" int x264hip_open( x264hip_ctx **out, int device, const x264hip_params *params )
{
    if( !out || !params ) return X264HIP_EINVAL;
    *out = nullptr;
    const x264hip_params &p = *params;
    if( ( p.bit_depth != 8 && p.bit_depth != 10 ) || p.width < 16 || p.height < 16 || p.bframes < 0 || p.bframes > X264HIP_BFRAME_MAX ||
        !p.cost_mv || p.mv_range < 1 || ( p.subpel_refine != 2 && p.subpel_refine != 4 ) || p.max_frames < 2 ||
        ( p.me_method != X264HIP_ME_DIA && p.me_method != X264HIP_ME_HEX ) || p.aq_mode < 0 || p.aq_mode > 3 ||
        p.lookahead_slices < 0 || p.lookahead_slices > X264HIP_LOOKAHEAD_SLICES_MAX || p.chroma_format < 0 || p.chroma_format > 3 )
        return X264HIP_EINVAL;
    int ndev = 0;
    if( hipGetDeviceCount( &ndev ) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev )
    {
        fprintf( stderr, "x264hip: no usable HIP device (count=%d); this path has no CPU fallback\n", ndev );
        return X264HIP_ENODEV;
    }
    if( hipSetDevice( device ) != hipSuccess )
        return X264HIP_ENODEV;
    x264hip_ctx *ctx = new x264hip_ctx();
    {
        hipDeviceProp_t prop;
        if( hipGetDeviceProperties( &prop, device ) == hipSuccess && prop.multiProcessorCount > 0 )
            ctx->n_cu = prop.multiProcessorCount;
    }
    ctx->p = p;
    ctx->p.cost_mv = nullptr;
    ctx->device = device;
    ctx->psz = p.bit_depth == 8 ? 1 : 2;
    const int mb_w = ( p.width + 15 ) / 16, mb_h = ( p.height + 15 ) / 16;
    ctx->lw = 8 * mb_w; ctx->lh = 8 * mb_h; ctx->n_mb = mb_w * mb_h;
    LaP &P = ctx->P;
    P.mb_w = mb_w; P.mb_h = mb_h;
    P.stride = (int)align_up( ctx->lw + 2 * LA_PAD, 64 );
    P.plane_elems = P.stride * ( ctx->lh + 2 * LA_PAD );
    ctx->plane_bytes = (size_t)P.plane_elems * ctx->psz;
    if( (long long)P.stride * ( ctx->lh + 2 * LA_PAD ) >= ( 1ll << 24 ) )
    {
        // the kernels address samples with 24-bit multiplies and 32-bit offsets: pictures up to 8K (16 M samples per padded plane)
        delete ctx;
        return X264HIP_EINVAL;
    }
    P.lambda = p.lambda; P.me_method = p.me_method; P.subpel_refine = p.subpel_refine; P.me_range = p.me_range;
    P.mv_range = p.mv_range; P.subme = p.subme; P.mbcmp_satd = p.mbcmp_satd; P.fpelcmp_satd = p.fpelcmp_satd;
    P.weighted_bipred = p.weighted_bipred; P.aq_mode = p.aq_mode; P.depth_shift = p.bit_depth - 8;
    P.pixel_max = ( 1 << p.bit_depth ) - 1;
    // frames of at most two blocks in a direction are always evaluated whole (slicetype.c:823)
    P.no_edges = p.no_edges && P.mb_w > 2 && P.mb_h > 2;
    P.n_slices = std::max( 1, p.lookahead_slices );

#define OPENCK( call ) do { if( ( call ) != hipSuccess ) { fprintf( stderr, "x264hip_open: %s failed\n", #call ); free_all( ctx ); delete ctx; return X264HIP_ENOMEM; } } while( 0 )
    // (the main stream at high priority and the chip-filling search launches detoured to a low-priority stream, so that other contexts'
    //  small kernels get freed wave slots first: 28.8 k against 33.8 k frames/s with eight contexts, gpurun_out/r07f -- not kept)
    OPENCK( hipStreamCreateWithFlags( &ctx->stream, hipStreamNonBlocking ) );
    OPENCK( hipEventCreate( &ctx->ev_start ) );
    OPENCK( hipEventCreate( &ctx->ev_stop ) );
    const int n_tab = 2 * 4 * p.mv_range;
    OPENCK( hipMalloc( &ctx->cost_mv_dev, ( 2 * n_tab + 1 ) * sizeof( uint16_t ) ) );
    OPENCK( hipMemcpy( ctx->cost_mv_dev, p.cost_mv - n_tab, ( 2 * n_tab + 1 ) * sizeof( uint16_t ), hipMemcpyHostToDevice ) );
    P.cost_mv = ctx->cost_mv_dev + n_tab;
    {
        // the reference's 5-decimal log2 table and 8-bit exp2 table (common/tables.c:58-90), regenerated
        AqLuts l;
        for( int i = 0; i < 128; i++ ) l.log2_lut[i] = (float)( floor( log2( 1.0 + i / 128.0 ) * 100000.0 + 0.5 ) / 100000.0 );
        for( int i = 0; i < 64; i++ ) l.exp2_lut[i] = (unsigned char)floor( ( pow( 2.0, i / 64.0 ) - 1.0 ) * 256.0 + 0.5 );
        OPENCK( hipMalloc( &ctx->luts_dev, sizeof( AqLuts ) ) );
        OPENCK( hipMemcpy( ctx->luts_dev, &l, sizeof( l ), hipMemcpyHostToDevice ) );
    }
    OPENCK( hipMalloc( &ctx->sync_words, 2 * ME_QUEUES * ME_QUEUE_STRIDE * sizeof( unsigned ) ) ); // row tickets of the two search kernels
    OPENCK( hipMemset( ctx->sync_words, 0, 2 * ME_QUEUES * ME_QUEUE_STRIDE * sizeof( unsigned ) ) );
#ifdef ME_PROFILE
    OPENCK( hipMalloc( &ctx->me_prof, 32 * sizeof( unsigned long long ) ) );
    OPENCK( hipMemset( ctx->me_prof, 0, 32 * sizeof( unsigned long long ) ) );
#endif
    ctx->n_cells = ( p.bframes + 2 ) * ( p.bframes + 2 );
    {
        const int ns = p.bframes + 2;
        int n_b = 0;
        for( int d0 = 1; d0 <= p.bframes + 1; d0++ )
            for( int d1 = 1; d0 + d1 <= p.bframes + 1; d1++ )
                n_b++;
        const int n_p = p.bframes + 1; // the P classes have a spare too: the cell over a speculative WEIGHTED field
        ctx->spare_at.assign( ctx->n_cells, ctx->n_cells + n_b + n_p );
        for( int d0 = 1, k = 0; d0 <= p.bframes + 1; d0++ )
            for( int d1 = 1; d0 + d1 <= p.bframes + 1; d1++ )
                ctx->spare_at[d0 * ns + d1] = ctx->n_cells + k++;
        for( int d0 = 1; d0 <= p.bframes + 1; d0++ )
            ctx->spare_at[d0 * ns] = ctx->n_cells + n_b + d0 - 1;
        ctx->n_store = ctx->n_cells + n_b + n_p + 1;
        ctx->spare2_at.assign( ctx->n_cells, ctx->n_store - 1 );
        for( int d0 = 1; d0 <= p.bframes + 1; d0++ )
            for( int d1 = 1; d0 + d1 <= p.bframes + 1; d1++ )
                ctx->spare2_at[d0 * ns + d1] = ctx->n_store++;
    }
    ctx->pos_frames.assign( x264hip_ctx::POS_KEYS, 0 );
    ctx->pos_field_req.assign( (size_t)x264hip_ctx::POS_KEYS * 2 * ( X264HIP_BFRAME_MAX + 1 ), 0 );
    ctx->pos_cell_req.assign( (size_t)x264hip_ctx::POS_KEYS * ctx->n_cells, 0 );
    OPENCK( hipHostMalloc( &ctx->cell_acc_host, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) ) );
    OPENCK( hipHostMalloc( &ctx->cell_alt_host, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) ) );
    memset( ctx->cell_alt_host, 0, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) );
    OPENCK( hipHostMalloc( &ctx->cell_alt2_host, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) ) );
    memset( ctx->cell_alt2_host, 0, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) );
    memset( ctx->cell_acc_host, 0, (size_t)p.max_frames * ctx->n_cells * 8 * sizeof( int ) );
    OPENCK( hipHostMalloc( &ctx->err_host, 16 * sizeof( unsigned ) ) ); // (report_wait_timeout, device_common.h)
    memset( ctx->err_host, 0, 16 * sizeof( unsigned ) );
    OPENCK( hipHostMalloc( &ctx->stats_host, (size_t)p.max_frames * 2 * sizeof( unsigned long long ) ) );
    // (MB-tree launches on the main stream instead, no cross-stream waits at all: 20 700 frames/s against 21 400, eight contexts)
    OPENCK( hipStreamCreateWithFlags( &ctx->stream2, hipStreamNonBlocking ) );
    {
        int prio_low = 0, prio_high = 0;
        OPENCK( hipDeviceGetStreamPriorityRange( &prio_low, &prio_high ) );
        OPENCK( hipStreamCreateWithPriority( &ctx->stream_up, hipStreamNonBlocking, prio_high ) );
    }
    for( int i = 0; i < x264hip_ctx::UP_EVS; i++ )
        OPENCK( hipEventCreateWithFlags( &ctx->up_ev[i], hipEventDisableTiming ) );
    OPENCK( hipEventCreateWithFlags( &ctx->ev_cross, hipEventDisableTiming ) );
    OPENCK( hipEventCreateWithFlags( &ctx->ev_mbt_last, hipEventDisableTiming ) );
    OPENCK( hipEventCreateWithFlags( &ctx->ev_ingest, hipEventDisableTiming ) );
    for( int i = 0; i < x264hip_ctx::BATCH_EVS; i++ )
        OPENCK( hipEventCreateWithFlags( &ctx->batch_ev[i], hipEventDisableTiming ) );
    OPENCK( hipMalloc( &ctx->mbt_bar, ( x264hip_ctx::MBT_RING * MBT_MAX_GROUPS * 4 + 4 ) * sizeof( unsigned ) ) );
    OPENCK( hipMemset( ctx->mbt_bar, 0, ( x264hip_ctx::MBT_RING * MBT_MAX_GROUPS * 4 + 4 ) * sizeof( unsigned ) ) );
    for( int i = 0; i < x264hip_ctx::MBT_RING; i++ )
    {
        // (the step table, then the steps' indices sorted by level: mbt_flush)
        OPENCK( hipHostMalloc( &ctx->mbt_host[i], x264hip_ctx::MBT_CAP * ( sizeof( MbtOpDev ) + sizeof( int ) ) ) );
        OPENCK( hipMalloc( &ctx->mbt_dev[i], x264hip_ctx::MBT_CAP * ( sizeof( MbtOpDev ) + sizeof( int ) ) ) );
        OPENCK( hipEventCreateWithFlags( &ctx->mbt_done[i], hipEventDisableTiming ) );
    }
    OPENCK( ring_alloc( ctx->put_ring, (size_t)ctx->put_desc_cap * sizeof( PutDesc ) ) );
    ctx->cell_desc_cap = 4096;
    OPENCK( ring_alloc( ctx->cell_ring, (size_t)ctx->cell_desc_cap * sizeof( CellArgs ) ) );
    OPENCK( hipMalloc( &ctx->wcost_dev, (size_t)x264hip_ctx::WCAP * 4 * sizeof( unsigned ) ) );
    OPENCK( hipMemset( ctx->wcost_dev, 0, (size_t)x264hip_ctx::WCAP * 4 * sizeof( unsigned ) ) );
    OPENCK( hipHostMalloc( &ctx->wcost_host, (size_t)x264hip_ctx::WCAP * 2 * sizeof( unsigned ) ) );
    OPENCK( ring_alloc( ctx->wjob_ring, (size_t)x264hip_ctx::WCAP * sizeof( WeightJob ) ) );
    OPENCK( ring_alloc( ctx->xfer_ring, (size_t)ctx->xfer_cap * sizeof( CellXfer ) ) );
    ctx->wcache.assign( x264hip_ctx::WCAP, x264hip_ctx::WEntry() );
    ctx->desc_cap = 2 * ( p.bframes + 1 ) * p.max_frames + 16;
    OPENCK( ring_alloc( ctx->search_ring, (size_t)ctx->desc_cap * sizeof( SearchDesc<uint8_t> ) ) );
    ctx->staging_bytes = (size_t)p.width * p.height * ctx->psz;
    OPENCK( hipHostMalloc( &ctx->staging, ctx->staging_bytes ) );
    ctx->stage[0] = ctx->staging;
    if( h2d_stream_acquire( ctx->device, &ctx->stream_h2d ) ) { ctx->stream_h2d = nullptr; OPENCK( hipErrorUnknown ); }
    OPENCK( hipEventCreateWithFlags( &ctx->h2d_done, hipEventDisableTiming ) );
    for( int k = 0; k < x264hip_ctx::STAGE_RING; k++ )
        OPENCK( hipEventCreateWithFlags( &ctx->stage_ev[k], hipEventDisableTiming ) );
    for( int k = 0; k < x264hip_ctx::H2D_GROUPS; k++ )
        OPENCK( hipEventCreateWithFlags( &ctx->h2d_group_free[k], hipEventDisableTiming ) );

    const int nd = p.bframes + 1, nc = ( p.bframes + 2 ) * ( p.bframes + 2 );
    ctx->slots.resize( p.max_frames );
    for( auto &s : ctx->slots )
    {
        size_t off = 0;
        const size_t o_planes = off; off += align_up( 12 * ctx->plane_bytes, 256 ); // four row-major planes, then their strip copies (twice the size)
        const size_t o_luma = off; off += align_up( ctx->staging_bytes, 256 );
        const size_t o_inv = off; off += align_up( ctx->n_mb * sizeof( uint16_t ), 256 );
        const size_t o_mbs = off; off += align_up( ctx->n_mb * sizeof( uint2 ), 256 );
        const size_t o_mvq = off; off += align_up( (size_t)3 * nd * ctx->n_mb * sizeof( unsigned long long ), 256 );
        const size_t o_mvc = off; off += align_up( (size_t)3 * nd * ctx->n_mb * sizeof( int ), 256 );
        // (more cells than the reference has: spare_at[idx] holds the speculative SECOND variant of B cell idx, see FrameSlot alts)
        const int nst = ctx->n_store;
        const size_t o_lc = off; off += align_up( (size_t)nst * ctx->n_mb * sizeof( uint16_t ), 256 );
        const size_t o_rows = off; off += align_up( (size_t)nst * mb_h * sizeof( int ), 256 );
        const size_t o_blk = off; off += align_up( (size_t)nst * ctx->n_mb * sizeof( int ), 256 );
        const size_t o_prop = off; off += align_up( (size_t)ctx->n_mb * sizeof( int ), 256 );
        const size_t o_qpa = off; off += align_up( (size_t)ctx->n_mb * sizeof( float ), 256 );
        const size_t o_qp = off; off += align_up( (size_t)ctx->n_mb * sizeof( float ), 256 );
        const size_t o_sums = off; off += align_up( (size_t)nst * 8 * sizeof( int ), 256 );
        const size_t o_work = off; off += align_up( (size_t)nst * 8 * sizeof( int ), 256 ); // cell_reduce_kernel's work words (zero between launches)
        char *base = nullptr;
        OPENCK( hipMalloc( &base, off ) );
        OPENCK( hipMemset( base, 0, off ) );
        s.planes = base + o_planes; s.luma = base + o_luma; s.inv_qscale = (uint16_t *)( base + o_inv );
        s.frame_sums = ctx->stats_host + 2 * ( &s - &ctx->slots[0] ); // pinned: read by the host after a stream sync
        s.mb_sums = (uint2 *)( base + o_mbs );
        for( int l = 0; l < 3; l++ )
            for( int d = 0; d < nd; d++ )
            {
                s.mvq[l][d] = (unsigned long long *)( base + o_mvq ) + (size_t)( l * nd + d ) * ctx->n_mb;
                s.mvcost[l][d] = (int *)( base + o_mvc ) + (size_t)( l * nd + d ) * ctx->n_mb;
            }
        s.lowres_costs = (uint16_t *)( base + o_lc );
        s.row_satds = (int *)( base + o_rows );
        s.blk = (int *)( base + o_blk );
        s.prop = s.prop_view = (int *)( base + o_prop ); s.qp_aq = (float *)( base + o_qpa ); s.qp = (float *)( base + o_qp );
        s.cell_sums = (int *)( base + o_sums );
        s.cell_work = (int *)( base + o_work );
        s.cells.assign( nc, CellEntry() );
        s.alts.assign( nc, CellEntry() );
        s.alts2.assign( nc, CellEntry() );
        s.cell_at.resize( nc );
        for( int c = 0; c < nc; c++ ) s.cell_at[c] = c;
        s.req_cells.assign( nc, 0 );
        memset( s.field_tag, 0, sizeof( s.field_tag ) );
        memset( s.field_ready, 0, sizeof( s.field_ready ) );
        memset( s.field_prefetched, 0, sizeof( s.field_prefetched ) );
        memset( s.field_remote, 0, sizeof( s.field_remote ) );
    }
    // hipMemset of device memory returns before the fill has run (it is ordered on the NULL stream, which the context's own streams --
    // hipStreamNonBlocking -- do not wait for): without this wait the fill of a slot could land AFTER the slot's first search had begun to
    // publish its vectors there, and the waves that wait for those vectors waited until the spin limit.  Seen as an "in-kernel wait timed
    // out" once in some five runs of the bench, always within the first launches of freshly opened contexts, a granule of the row below
    // reading zero by every kind of access (gpurun_out/r07r-r07t; rounds 1-6 were lucky, or slower to start).
    if( !getenv( "X264HIP_NO_OPEN_SYNC" ) ) // (the switch: to see tests/test_gpu_fresh_contexts.py fail without the wait)
        OPENCK( hipStreamSynchronize( nullptr ) );
#undef OPENCK
    g_open_contexts[ctx->device & 63]++; ctx->counted_open = true;
    *out = ctx;
    return X264HIP_OK;
}

extern "C" int x264hip_device_name( x264hip_ctx *ctx, char *buf, size_t cap )
{
    if( !ctx || !buf || !cap ) return X264HIP_EINVAL;
    hipDeviceProp_t prop;
    HIPCK( hipGetDeviceProperties( &prop, ctx->device ) );
    snprintf( buf, cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount );
    return X264HIP_OK;
}

static int mbt_flush( x264hip_ctx *ctx );
// do the ingest kernels of this process write all four planes row-major?  (split ingest or X264HIP_ROWMAJOR=4; default: plane 0 only)
static bool rowmajor_all_at_ingest()
{
    static const bool v = ( getenv( "X264HIP_INGEST" ) && !strcmp( getenv( "X264HIP_INGEST" ), "split" ) ) ||
                          ( getenv( "X264HIP_ROWMAJOR" ) && atoi( getenv( "X264HIP_ROWMAJOR" ) ) == 4 );
    return v;
}
static int mbt_flush_at( x264hip_ctx *ctx, int line );
static int mbt_guard_slot( x264hip_ctx *ctx, const FrameSlot &s );
extern "C" int x264hip_synchronize( x264hip_ctx *ctx )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    {
        int rc = mbt_flush_at( ctx, __LINE__ ); // queued MB-tree lists are work the caller has handed over
        if( rc ) return rc;
    }
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    return X264HIP_OK;
}

extern "C" int x264hip_flush( x264hip_ctx *ctx )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    return mbt_flush_at( ctx, __LINE__ );
}

extern "C" int x264hip_geometry( x264hip_ctx *ctx, int *mb_w, int *mb_h, int *lowres_stride )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( mb_w ) *mb_w = ctx->P.mb_w;
    if( mb_h ) *mb_h = ctx->P.mb_h;
    if( lowres_stride ) *lowres_stride = ctx->P.stride;
    return X264HIP_OK;
}

static int sync_stream( x264hip_ctx *ctx );
static int prof_drain( x264hip_ctx *ctx );
// x264hip_kernel_profile: the first event of a pair in front of one kernel of class `kind` (units: frames / cells it works on); the
// second one is recorded by kprof_end.  Both are no-ops unless bit 1 of the profile switch is set.
static int kprof_begin( x264hip_ctx *ctx, int kind, int units, hipEvent_t *end_ev )
{
    *end_ev = nullptr;
    if( !( ctx->prof_on & 2 ) ) return X264HIP_OK;
    if( ctx->prof_used + 2 > (int)ctx->prof_ev.size() )
    {
        int rc = prof_drain( ctx );
        if( rc ) return rc;
    }
    HIPCK( hipEventRecord( ctx->prof_ev[ctx->prof_used], ctx->stream ) );
    *end_ev = ctx->prof_ev[ctx->prof_used + 1];
    while( ctx->prof_kind.size() < ctx->prof_n.size() ) ctx->prof_kind.push_back( -1 );
    ctx->prof_n.push_back( units ); ctx->prof_kind.push_back( kind );
    ctx->prof_used += 2;
    return X264HIP_OK;
}
static int kprof_end( x264hip_ctx *ctx, hipEvent_t end_ev )
{
    if( end_ev ) HIPCK( hipEventRecord( end_ev, ctx->stream ) );
    return X264HIP_OK;
}
#define KPROF( kind, units, launch ) do { hipEvent_t kp_; { int rc_ = kprof_begin( ctx, kind, units, &kp_ ); if( rc_ ) return rc_; } launch; { int rc_ = kprof_end( ctx, kp_ ); if( rc_ ) return rc_; } } while( 0 )

static inline bool slot_ok( x264hip_ctx *ctx, int s ) { return s >= 0 && s < (int)ctx->slots.size(); }

// ---- frame ingest ------------------------------------------------------------------------------------
static PutDesc make_put_desc( x264hip_ctx *ctx, FrameSlot &s, const void *src, int src_stride, const void *cb, const void *cr, int cstride,
                              int aq_on )
{
    PutDesc d;
    memset( &d, 0, sizeof( d ) );
    d.src = src; d.src_stride = src_stride; d.cb = cb; d.cr = cr; d.cstride = cstride;
    d.planes = s.planes; d.inv_qscale = s.inv_qscale; d.mb_sums = s.mb_sums; d.frame_sums = s.frame_sums;
    d.qp_aq = s.qp_aq; d.qp = s.qp; d.intra_cost = s.lowres_costs; d.aq_on = aq_on;
    return d;
}

// lowres + AQ + intra of n frames: one launch of each kernel (blockIdx.z / blockIdx.x = frame)
template <typename T>
static int launch_ingest_t( x264hip_ctx *ctx, const PutDesc *descs_dev, const PutDesc &single, int n )
{
    const x264hip_params &p = ctx->p;
    const LaP &P = ctx->P;
    const float strength = p.aq_strength * 1.0397f;
    const float bias = 14.427f + 2 * ( p.bit_depth - 8 );
    static const bool split_ingest = getenv( "X264HIP_INGEST" ) && !strcmp( getenv( "X264HIP_INGEST" ), "split" );
    const int rows = ctx->lh + 2 * LA_PAD;
    const bool rowmajor_all = rowmajor_all_at_ingest(); // A/B runs: H / V / HV row-major at ingest
    if( !split_ingest && rowmajor_all )
        KPROF( X264HIP_KPROF_LOWRES, n, ( lowres_tiles_kernel<T, 4><<<dim3( ( P.stride + LT_COLS - 1 ) / LT_COLS, ( rows + LT_ROWS - 1 ) / LT_ROWS, n ), 256, 0, ctx->stream>>>(
            descs_dev, single, p.width, p.height, P.plane_elems, P.stride, ctx->lw, ctx->lh ) ) );
    else if( !split_ingest )
        KPROF( X264HIP_KPROF_LOWRES, n, ( lowres_tiles_kernel<T, 1><<<dim3( ( P.stride + LT_COLS - 1 ) / LT_COLS, ( rows + LT_ROWS - 1 ) / LT_ROWS, n ), 256, 0, ctx->stream>>>(
            descs_dev, single, p.width, p.height, P.plane_elems, P.stride, ctx->lw, ctx->lh ) ) );
    else
    {
        // the two-kernel form (planes, then their strip copy read back from the planes): kept for comparison
        dim3 grd( ( ( ctx->lw + 2 * LA_PAD ) / 4 + 255 ) / 256, rows, n );
        lowres_kernel<T><<<grd, 256, 0, ctx->stream>>>( descs_dev, single, p.width, p.height, P.plane_elems, P.stride, ctx->lw, ctx->lh );
        strips_kernel<T><<<dim3( ( rows + 63 ) / 64, ( 4 * ( P.stride / 8 ) + 3 ) / 4, n ), 256, 0, ctx->stream>>>( descs_dev, single, P.plane_elems, P.stride, rows );
    }
    const int wg_per_frame = ( ( ctx->n_mb + AQ_MBS_PER_WG - 1 ) / AQ_MBS_PER_WG + 7 ) / 8 * 8; // a multiple of 8: contiguous runs of macroblocks per XCD
    KPROF( X264HIP_KPROF_AQ, n, ( aq_kernel<T><<<dim3( wg_per_frame, 1, n ), 64, 0, ctx->stream>>>( descs_dev, single, p.width, p.height, P.mb_w, P.mb_h, strength, bias, ctx->luts_dev,
                                                                       p.aq_mode, 1.f / ( 1 << ( 2 * ( p.bit_depth - 8 ) ) ), p.chroma_format ) ) );
    if( p.aq_mode >= 2 && p.aq_strength != 0.f )
        aq_auto_kernel<<<n, 1024, 0, ctx->stream>>>( descs_dev, single, ctx->n_mb, p.aq_mode, p.aq_strength, ctx->luts_dev );
    // (256 threads: beside other contexts' search waves a 1 024-thread workgroup waits for sixteen free wave slots on ONE compute unit --
    //  3.5 ms on average for an 8 us kernel with eight contexts in flight, profiles/r05_bench_kernel_stats.csv -- and every frame's
    //  ingest stands behind it)
    aq_reduce_kernel<<<n, 256, 0, ctx->stream>>>( descs_dev, single, ctx->n_mb );
    const int intra_wgs = ( ( ctx->n_mb + INTRA_BLOCKS_PER_WG - 1 ) / INTRA_BLOCKS_PER_WG + 7 ) / 8 * 8;
    if( P.subme > 1 )
        KPROF( X264HIP_KPROF_INTRA, n, ( intra_kernel<T, 10><<<dim3( intra_wgs, 1, n ), 64, 0, ctx->stream>>>( P, descs_dev, single ) ) );
    else
        KPROF( X264HIP_KPROF_INTRA, n, ( intra_kernel<T, 3><<<dim3( intra_wgs, 1, n ), 64, 0, ctx->stream>>>( P, descs_dev, single ) ) );
    HIPCK( hipGetLastError() );
    HIPCK( hipEventRecord( ctx->ev_ingest, ctx->stream ) );
    return X264HIP_OK;
}

static void slot_reset( x264hip_ctx *ctx, FrameSlot &s )
{
    s.rowmajor_mask = rowmajor_all_at_ingest() ? 0xF : 1;
    if( s.in_use && s.pos_key )
    {
        // the frame that leaves this slot has been asked for everything it will ever be asked for: count it under its position
        const int k = s.pos_key;
        ctx->pos_frames[k]++;
        for( int l = 0; l < 2; l++ )
            for( int d = 0; d <= X264HIP_BFRAME_MAX; d++ )
                if( s.req_fields[l] >> d & 1 )
                    ctx->pos_field_req[( (size_t)k * 2 + l ) * ( X264HIP_BFRAME_MAX + 1 ) + d]++;
        for( int c = 0; c < ctx->n_cells; c++ )
            if( s.req_cells[c] )
                ctx->pos_cell_req[(size_t)k * ctx->n_cells + c]++;
    }
    s.pos_key = 0; s.req_fields[0] = s.req_fields[1] = 0;
    std::fill( s.req_cells.begin(), s.req_cells.end(), 0 );
    s.in_use = 1;
    s.frame_no = -1; // (known again once the frame has been named in an x264hip_prefetch call)
    s.gen++;
    s.stats_valid = 0;
    memset( s.field_ready, 0, sizeof( s.field_ready ) );
    memset( s.field_prefetched, 0, sizeof( s.field_prefetched ) );
    memset( s.field_tag, 0, sizeof( s.field_tag ) );
    memset( s.field_remote, 0, sizeof( s.field_remote ) );
    s.cells.assign( ctx->n_cells, CellEntry() );
    s.alts.assign( ctx->n_cells, CellEntry() );
    s.alts2.assign( ctx->n_cells, CellEntry() );
    for( int c = 0; c < ctx->n_cells; c++ ) s.cell_at[c] = c;
    if( s.wplane_idx >= 0 ) { ctx->wplane_owner[s.wplane_idx] = -1; s.wplane_idx = -1; }
    for( auto &ws : s.wspec ) ws = FrameSlot::WSpec();
    ctx->counters[3]++;
}

// ---- pictures from host memory ------------------------------------------------------------------------------------------------------
// Is p something the DMA engines can read where it is (hipHostMalloc / hipHostRegister memory), or a device pointer, or plain pageable
// host memory?  0 = pageable host, 1 = pinned host, 2 = device.
static int pointer_kind( const void *p )
{
    hipPointerAttribute_t a;
    if( hipPointerGetAttributes( &a, p ) != hipSuccess )
    {
        (void)hipGetLastError(); // (an unregistered host pointer is reported as an error)
        return 0;
    }
    return a.type == hipMemoryTypeDevice ? 2 : a.type == hipMemoryTypeHost ? 1 : a.type == hipMemoryTypeManaged ? 2 : 0;
}
// Opens a group of host-to-device copies: the slots' luma buffers are read by the ingest kernels of their previous pictures
// (ev_ingest stands behind the most recent ones), nothing else touches them.
static int h2d_begin( x264hip_ctx *ctx )
{
    HIPCK( hipStreamWaitEvent( ctx->stream_h2d, ctx->ev_ingest, 0 ) );
    return X264HIP_OK;
}
// one picture (width x height samples, `stride` samples per row in the caller's buffer) into the slot's luma buffer, on the DMA stream
static int h2d_picture( x264hip_ctx *ctx, FrameSlot &s, const void *luma, int stride, int kind )
{
    const x264hip_params &p = ctx->p;
    const size_t row = (size_t)p.width * ctx->psz;
    if( kind == 1 )
    {
        // (rows that follow each other in the caller's buffer are ONE transfer: the strided form goes row by row)
        if( stride == p.width )
            HIPCK( hipMemcpyAsync( s.luma, luma, ctx->staging_bytes, hipMemcpyHostToDevice, ctx->stream_h2d ) );
        else
            HIPCK( hipMemcpy2DAsync( s.luma, row, luma, (size_t)stride * ctx->psz, row, p.height, hipMemcpyHostToDevice, ctx->stream_h2d ) );
        ctx->h2d_direct++;
    }
    else
    {
        const int k = ctx->stage_next++ % x264hip_ctx::STAGE_RING;
        if( !ctx->stage[k] && hipHostMalloc( &ctx->stage[k], ctx->staging_bytes ) != hipSuccess ) return X264HIP_ENOMEM;
        if( ctx->stage_used[k] )
            HIPCK( hipEventSynchronize( ctx->stage_ev[k] ) ); // the copy that last read this staging buffer (three pictures ago) is done
        if( stride == p.width )
            memcpy( ctx->stage[k], luma, ctx->staging_bytes );
        else
            for( int y = 0; y < p.height; y++ )
                memcpy( ctx->stage[k] + (size_t)y * row, (const char *)luma + (size_t)y * stride * ctx->psz, row );
        HIPCK( hipMemcpyAsync( s.luma, ctx->stage[k], ctx->staging_bytes, hipMemcpyHostToDevice, ctx->stream_h2d ) );
        HIPCK( hipEventRecord( ctx->stage_ev[k], ctx->stream_h2d ) );
        ctx->stage_used[k] = true;
        ctx->h2d_staged++;
    }
    ctx->h2d_bytes += ctx->staging_bytes;
    return X264HIP_OK;
}
// closes the group: whatever the compute stream is given next runs behind the copies
static int h2d_end( x264hip_ctx *ctx )
{
    HIPCK( hipEventRecord( ctx->h2d_done, ctx->stream_h2d ) );
    HIPCK( hipStreamWaitEvent( ctx->stream, ctx->h2d_done, 0 ) );
    return X264HIP_OK;
}

// One picture through a copy kernel on the compute stream; X264HIP_ESTATE: not possible for this buffer (the caller takes the DMA road).
static int h2d_single_by_kernel( x264hip_ctx *ctx, FrameSlot &s, const void *luma, int stride, int kind )
{
    static const bool dma_only = getenv( "X264HIP_H2D" ) && !strcmp( getenv( "X264HIP_H2D" ), "dma" );
    const x264hip_params &p = ctx->p;
    if( dma_only || ( ctx->staging_bytes & 15 ) || ( (uintptr_t)s.luma & 15 ) ) return X264HIP_ESTATE;
    const char *host = (const char *)luma;
    int k = -1;
    if( kind != 1 || stride != p.width || ( (uintptr_t)luma & 15 ) )
    {
        // pageable (or strided / unaligned) pictures: one memcpy into the pinned ring, read from there
        if( kind == 1 && stride == p.width ) return X264HIP_ESTATE; // pinned but unaligned: the DMA engine takes it as it is
        k = ctx->stage_next++ % x264hip_ctx::STAGE_RING;
        if( !ctx->stage[k] && hipHostMalloc( &ctx->stage[k], ctx->staging_bytes ) != hipSuccess ) return X264HIP_ENOMEM;
        if( ctx->stage_used[k] )
            HIPCK( hipEventSynchronize( ctx->stage_ev[k] ) ); // the copy that last read this staging buffer is done
        const size_t row = (size_t)p.width * ctx->psz;
        if( stride == p.width )
            memcpy( ctx->stage[k], luma, ctx->staging_bytes );
        else
            for( int y = 0; y < p.height; y++ )
                memcpy( ctx->stage[k] + (size_t)y * row, (const char *)luma + (size_t)y * stride * ctx->psz, row );
        host = ctx->stage[k];
    }
    void *mapped = nullptr;
    if( hipHostGetDevicePointer( &mapped, (void *)host, 0 ) != hipSuccess || !mapped )
    {
        (void)hipGetLastError();
        if( k >= 0 ) ctx->stage_next--;
        return X264HIP_ESTATE;
    }
    const size_t n16 = ctx->staging_bytes / 16;
    const int grid = (int)std::min<size_t>( ( n16 + 1023 ) / 1024, 512 );
    copy16_kernel<4, false><<<grid, 256, 0, ctx->stream>>>( (const copy_v4u *)mapped, (copy_v4u *)s.luma, n16 );
    HIPCK( hipGetLastError() );
    if( k >= 0 )
    {
        HIPCK( hipEventRecord( ctx->stage_ev[k], ctx->stream ) );
        ctx->stage_used[k] = true;
        ctx->h2d_staged++;
    }
    else
        ctx->h2d_direct++;
    ctx->h2d_by_kernel++;
    ctx->h2d_bytes += ctx->staging_bytes;
    return X264HIP_OK;
}

extern "C" int x264hip_frame_put( x264hip_ctx *ctx, int slot, const void *luma, int stride, int is_device, const void *cb, const void *cr,
                                  int cstride, const uint16_t *inv_qscale )
{
    if( !ctx || !slot_ok( ctx, slot ) || !luma || stride < ctx->p.width || ( !cb ) != ( !cr ) ) return X264HIP_EINVAL;
    if( cb && cstride < ( ctx->p.chroma_format == 3 ? ctx->p.width : ( ctx->p.width + 1 ) >> 1 ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
    }
    {
        int rc = mbt_guard_slot( ctx, s ); // MB-tree steps may still read this slot's maps
        if( rc ) return rc;
    }
    slot_reset( ctx, s );
    const x264hip_params &p = ctx->p;
    const void *src = luma;
    int src_stride = stride;
    if( is_device && pointer_kind( luma ) != 2 )
        is_device = 0; // (the pointer's own attributes decide: a host buffer announced as device memory still takes the DMA road)
    if( !is_device )
    {
        // ONE picture: fetched by a copy kernel on the compute stream, which reads the caller's pinned buffer (or the pinned staging
        // copy of a pageable one) across PCIe -- no DMA stream, no event between two streams in front of the ingest kernels, which
        // are what an encoder-paced caller waits for (a 2 MB DMA transfer takes 98 us and two cross-stream waits; X264HIP_H2D=dma
        // keeps that road).  Buffers the kernel cannot address (an odd size or alignment, no device mapping) take the DMA road.
        int rc = h2d_single_by_kernel( ctx, s, luma, stride, pointer_kind( luma ) == 1 ? 1 : 0 );
        if( rc == X264HIP_ESTATE )
        {
            rc = h2d_begin( ctx );
            if( !rc ) rc = h2d_picture( ctx, s, luma, stride, pointer_kind( luma ) == 1 ? 1 : 0 );
            if( !rc ) rc = h2d_end( ctx );
        }
        if( rc ) return rc;
        src = s.luma;
        src_stride = p.width;
        if( cb && cr )
        {
            // the two chroma planes (4:2:0) follow through their own staging pair, which is reused picture after picture: the device copy
            // of the previous picture's chroma has to have been consumed (AQ with chroma from host buffers is the occasional caller)
            HIPCK( hipStreamSynchronize( ctx->stream ) );
            const int cw = p.chroma_format == 3 ? p.width : ( p.width + 1 ) >> 1, ch = p.chroma_format >= 2 ? p.height : ( p.height + 1 ) >> 1;
            const size_t crow = (size_t)cw * ctx->psz, cplane = crow * ch;
            if( !ctx->chroma_staging )
            {
                HIPCK( hipHostMalloc( &ctx->chroma_staging, 2 * cplane ) );
                HIPCK( hipMalloc( &ctx->chroma_dev, 2 * cplane ) );
            }
            for( int y = 0; y < ch; y++ )
            {
                memcpy( ctx->chroma_staging + (size_t)y * crow, (const char *)cb + (size_t)y * cstride * ctx->psz, crow );
                memcpy( ctx->chroma_staging + cplane + (size_t)y * crow, (const char *)cr + (size_t)y * cstride * ctx->psz, crow );
            }
            HIPCK( hipMemcpyAsync( ctx->chroma_dev, ctx->chroma_staging, 2 * cplane, hipMemcpyHostToDevice, ctx->stream ) );
            cb = ctx->chroma_dev; cr = ctx->chroma_dev + cplane; cstride = cw;
        }
    }
    const int aq_on = p.aq_mode >= 1 && p.aq_strength != 0.f && !inv_qscale;
    // the chroma planes take part in the AQ energy (ac_energy_mb, ratecontrol.c:258-276) when the caller supplies them
    const PutDesc d = make_put_desc( ctx, s, src, src_stride, cb, cr, cstride, aq_on );
    int rc = p.bit_depth == 8 ? launch_ingest_t<uint8_t>( ctx, nullptr, d, 1 ) : launch_ingest_t<uint16_t>( ctx, nullptr, d, 1 );
    if( rc ) return rc;
    if( inv_qscale )
        HIPCK( hipMemcpyAsync( s.inv_qscale, inv_qscale, ctx->n_mb * sizeof( uint16_t ), hipMemcpyHostToDevice, ctx->stream ) );
    return X264HIP_OK;
}

// Batch form for frames already resident on the device: one launch per ingest kernel for all n frames.
extern "C" int x264hip_frame_put_batch( x264hip_ctx *ctx, int n, const int *slots, const void *const *luma_dev, int stride )
{
    return x264hip_frame_put_batch_yuv( ctx, n, slots, luma_dev, stride, nullptr, nullptr, 0 );
}

extern "C" int x264hip_frame_put_batch_yuv( x264hip_ctx *ctx, int n, const int *slots, const void *const *luma_dev, int stride,
                                            const void *const *cb_dev, const void *const *cr_dev, int cstride )
{
    if( !ctx || n <= 0 || !slots || !luma_dev || stride < ctx->p.width || ( !cb_dev ) != ( !cr_dev ) ||
        ( cb_dev && cstride < ( ctx->p.chroma_format == 3 ? ctx->p.width : ( ctx->p.width + 1 ) / 2 ) ) )
        return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const uint64_t t_enter = host_now_ns();
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
    }
    for( int i = 0; i < n; i++ )
        if( slot_ok( ctx, slots[i] ) )
        {
            int rc = mbt_guard_slot( ctx, ctx->slots[slots[i]] );
            if( rc ) return rc;
        }
    const x264hip_params &p = ctx->p;
    const int aq_on = p.aq_mode >= 1 && p.aq_strength != 0.f;
    // Pictures may also be HOST pointers (luma only): they travel on the DMA stream, sixteen to a group, and the ingest kernels of a
    // group run behind its copies while the next group is still on its way
    const int kind0 = pointer_kind( luma_dev[0] );
    if( kind0 != 2 && cb_dev ) return X264HIP_EINVAL;
    ctx->h2d_group_pics = (int)std::max<size_t>( 1, std::min<size_t>( x264hip_ctx::H2D_GROUP_PICS, ( (size_t)64 << 20 ) / ctx->staging_bytes ) );
    const int group = kind0 != 2 ? std::min( ctx->h2d_group_pics, ctx->put_desc_cap ) : ctx->put_desc_cap;
    static const bool no_merge = getenv( "X264HIP_H2D" ) && !strcmp( getenv( "X264HIP_H2D" ), "dma" ); // A/B runs: a transfer per picture
    // (eight threads that enqueue their groups at the same moment would interleave them, and every context's pictures would arrive
    //  late and together again: the lock is held while this call ENQUEUES -- microseconds per group, nothing waits for the device)
    std::unique_lock<std::mutex> batch_lock( g_h2d_batch_mutex[ctx->device & 63], std::defer_lock );
    bool h2d_begun = false; // (the wait for the slots' last readers: once per call, and only when a picture goes into a slot's own buffer)
    // ONE descriptor table per call (per put_desc_cap pictures): its groups' ingest launches read their part of it.  A table per group
    // ran the ring of four dry at the fifth group of a 160-picture call, and the wait for the first group's ingest kernels -- behind
    // seven other contexts' searches -- was made with the transfer queue's lock held: the link stood idle 40 % of a host-fed step
    // (gpurun_out/r07b/timeline.txt).  Whatever this call may have to wait for, it waits for without the lock.
    for( int o0 = 0; o0 < n; o0 += ctx->put_desc_cap )
    {
        const int n0 = std::min( n - o0, ctx->put_desc_cap );
        int ri = 0;
        if( ring_acquire( ctx->put_ring, &ri ) ) return X264HIP_EDEVICE;
        PutDesc *const dh0 = (PutDesc *)ctx->put_ring.host[ri], *const dd0 = (PutDesc *)ctx->put_ring.dev[ri];
        const uint64_t t_before = host_now_ns();
        if( kind0 != 2 )
            batch_lock.lock();
        const uint64_t t_locked = host_now_ns();
        int rc_launch = X264HIP_OK;
        // Beside other contexts the ingest kernels of the call are launched ONCE, behind its last transfer: a launch per group is five small
        // kernels in stream order, each of which waits its turn among the other contexts' search waves (fifty launches per 160 pictures:
        // a pass of twelve host-fed contexts took 145 ms, the link needs 72, gpurun_out/r07d) -- what a launch per group buys, kernels under
        // the context's own transfers, the other contexts' work provides anyway.  A context alone on its device keeps the launch per group.
        static const char *ingest_env = getenv( "X264HIP_H2D_INGEST" ); // "group" / "call": A/B runs
        const bool ingest_once = kind0 != 2 && ( ingest_env ? !strcmp( ingest_env, "call" ) : g_open_contexts[ctx->device & 63].load() > 1 );
        int used_groups[x264hip_ctx::H2D_GROUPS], n_used_groups = 0;
        int launched = o0; // pictures of this call whose ingest kernels have been launched
        for( int o = o0; o < o0 + n0 && !rc_launch; o += group )
        {
            const int m = std::min( o0 + n0 - o, group );
            PutDesc *dh = dh0 + ( o - o0 ), *dd = dd0 + ( o - o0 );
            // a group whose pictures follow each other in pinned memory is ONE transfer into a device group buffer, read there by the ingest kernels
            int gk = -1;
            if( kind0 == 1 && stride == p.width && m > 1 && !no_merge )
            {
                bool run = true;
                for( int i = 1; i < m && run; i++ )
                    run = (const char *)luma_dev[o + i] == (const char *)luma_dev[o] + (size_t)i * ctx->staging_bytes;
                // (one allocation: its first and last byte are pinned memory, so is everything between)
                run = run && pointer_kind( (const char *)luma_dev[o] + (size_t)m * ctx->staging_bytes - 1 ) == 1;
                if( run )
                {
                    gk = ctx->h2d_group_next++ % x264hip_ctx::H2D_GROUPS;
                    if( !ctx->h2d_group[gk] && hipMalloc( &ctx->h2d_group[gk], (size_t)ctx->h2d_group_pics * ctx->staging_bytes ) != hipSuccess )
                    {
                        (void)hipGetLastError();
                        gk = -1; // no memory for the group buffer: picture by picture
                    }
                }
                if( gk >= 0 )
                {
                    if( ctx->h2d_group_used[gk] && hipEventQuery( ctx->h2d_group_free[gk] ) != hipSuccess )
                    {
                        (void)hipGetLastError(); // (not ready is reported as an error)
                        batch_lock.unlock();     // the other contexts' transfers go ahead meanwhile
                        HIPCK( hipEventSynchronize( ctx->h2d_group_free[gk] ) );
                        batch_lock.lock();
                    }
                    HIPCK( hipMemcpyAsync( ctx->h2d_group[gk], luma_dev[o], (size_t)m * ctx->staging_bytes, hipMemcpyHostToDevice, ctx->stream_h2d ) );
                    ctx->h2d_merged++;
                    ctx->h2d_direct += m;
                    ctx->h2d_bytes += (size_t)m * ctx->staging_bytes;
                }
            }
            for( int i = 0; i < m; i++ )
            {
                if( !slot_ok( ctx, slots[o + i] ) || !luma_dev[o + i] ) return X264HIP_EINVAL;
                FrameSlot &s = ctx->slots[slots[o + i]];
                slot_reset( ctx, s );
                if( gk >= 0 )
                {
                    dh[i] = make_put_desc( ctx, s, ctx->h2d_group[gk] + (size_t)i * ctx->staging_bytes, p.width, nullptr, nullptr, 0, aq_on );
                    continue;
                }
                if( kind0 != 2 )
                {
                    if( !h2d_begun )
                    {
                        int rc = h2d_begin( ctx ); // (the slots of one call are distinct, a group's copies may overlap the previous group's kernels)
                        if( rc ) return rc;
                        h2d_begun = true;
                    }
                    int rc = h2d_picture( ctx, s, luma_dev[o + i], stride, pointer_kind( luma_dev[o + i] ) == 1 ? 1 : 0 );
                    if( rc ) return rc;
                    dh[i] = make_put_desc( ctx, s, s.luma, p.width, nullptr, nullptr, 0, aq_on );
                    continue;
                }
                dh[i] = make_put_desc( ctx, s, luma_dev[o + i], stride, cb_dev ? cb_dev[o + i] : nullptr, cb_dev ? cr_dev[o + i] : nullptr, cstride, aq_on );
            }
            if( ingest_once )
            {
                if( gk >= 0 && n_used_groups < x264hip_ctx::H2D_GROUPS ) used_groups[n_used_groups++] = gk;
                if( o + group < o0 + n0 && n_used_groups < x264hip_ctx::H2D_GROUPS - 1 )
                    continue; // (the launch comes with the call's last group -- or before the ring of group buffers would come round)
            }
            if( kind0 != 2 )
            {
                int rc = h2d_end( ctx );
                if( rc ) return rc;
            }
            // (ingest_once: everything since the last launch -- the groups' descriptors follow each other in the table)
            PutDesc *const lh = ingest_once ? dh0 + ( launched - o0 ) : dh, *const ld = ingest_once ? dd0 + ( launched - o0 ) : dd;
            const int lm = ingest_once ? o + m - launched : m;
            HIPCK( upload_async( ctx, ld, lh, (size_t)lm * sizeof( PutDesc ), ctx->stream ) );
            PutDesc none;
            memset( &none, 0, sizeof( none ) );
            rc_launch = p.bit_depth == 8 ? launch_ingest_t<uint8_t>( ctx, ld, none, lm ) : launch_ingest_t<uint16_t>( ctx, ld, none, lm );
            launched = o + m;
            if( ingest_once )
            {
                for( int k = 0; k < n_used_groups; k++ )
                {
                    HIPCK( hipEventRecord( ctx->h2d_group_free[used_groups[k]], ctx->stream ) );
                    ctx->h2d_group_used[used_groups[k]] = true;
                }
                n_used_groups = 0;
            }
            else if( gk >= 0 )
            {
                HIPCK( hipEventRecord( ctx->h2d_group_free[gk], ctx->stream ) );
                ctx->h2d_group_used[gk] = true;
            }
        }
        if( batch_lock.owns_lock() )
        {
            batch_lock.unlock();
            const uint64_t t_done = host_now_ns();
            ctx->h2d_calls++;
            ctx->h2d_t_prep += t_before - t_enter; ctx->h2d_t_wait += t_locked - t_before; ctx->h2d_t_hold += t_done - t_locked;
            if( ctx->h2d_t_last ) ctx->h2d_t_between += t_enter - ctx->h2d_t_last;
            ctx->h2d_t_last = t_done;
        }
        if( ring_commit( ctx->put_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
        if( rc_launch ) return rc_launch;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_frame_stats( x264hip_ctx *ctx, int slot, uint64_t *pixel_sum, uint64_t *pixel_ssd )
{
    if( !ctx || !slot_ok( ctx, slot ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    if( !s.in_use ) return X264HIP_ESTATE;
    if( !s.stats_valid )
    {
        // the ingest kernels write the totals straight into pinned host memory: one wait covers every pending frame
        std::vector<int> pend;
        for( int i = 0; i < (int)ctx->slots.size(); i++ )
            if( ctx->slots[i].in_use && !ctx->slots[i].stats_valid )
                pend.push_back( i );
        HIPCK( hipEventSynchronize( ctx->ev_ingest ) );
        const uint64_t n = (uint64_t)( 16 * ctx->P.mb_w ) * ( 16 * ctx->P.mb_h );
        for( int i : pend )
        {
            FrameSlot &f = ctx->slots[i];
            const unsigned long long sum = ctx->stats_host[2 * i], sq = ctx->stats_host[2 * i + 1];
            f.sum = (uint32_t)sum;
            // i_pixel_sum is a uint32_t in the reference (common/frame.h:140): the total has wrapped before it is squared
            const uint64_t s32 = (uint32_t)sum;
            f.ssd = sq - ( s32 * s32 + n / 2 ) / n; // ratecontrol.c:405-414
            f.stats_valid = 1;
        }
    }
    if( pixel_sum ) *pixel_sum = s.sum;
    if( pixel_ssd ) *pixel_ssd = s.ssd;
    return X264HIP_OK;
}

// ---- searches ---------------------------------------------------------------------------------------
static WtD make_wt( x264hip_ctx *ctx, const x264hip_weight *w )
{
    WtD d = { 0, 1, 0, 0 };
    if( w && w->on )
    {
        d.on = 1; d.scale = w->scale; d.denom = w->denom; d.offset = w->offset * ( 1 << ctx->P.depth_shift );
    }
    return d;
}

static int acquire_wplane( x264hip_ctx *ctx, int owner_slot )
{
    for( size_t i = 0; i < ctx->wplanes.size(); i++ )
        if( ctx->wplane_owner[i] < 0 ) { ctx->wplane_owner[i] = owner_slot; return (int)i; }
    char *pl = nullptr;
    if( hipMalloc( &pl, 2 * ctx->plane_bytes ) != hipSuccess ) return -1; // plane 0 in the strip layout of me_search.h
    ctx->wplanes.push_back( pl );
    ctx->wplane_owner.push_back( owner_slot );
    return (int)ctx->wplanes.size() - 1;
}

// fold finished event pairs of the profile ring into the running totals
static int prof_drain( x264hip_ctx *ctx )
{
    if( ctx->prof_used )
    {
        HIPCK( hipStreamSynchronize( ctx->stream ) );
        for( int i = 0; i < ctx->prof_used; i += 2 )
        {
            float ms = 0;
            if( hipEventElapsedTime( &ms, ctx->prof_ev[i], ctx->prof_ev[i + 1] ) != hipSuccess )
            {
                (void)hipGetLastError(); // a pair whose second event was never recorded (its launch failed half way): not a measurement
                continue;
            }
            const int k = ctx->prof_n[i / 2];
            const int kind = i / 2 < (int)ctx->prof_kind.size() ? ctx->prof_kind[i / 2] : -1;
            if( kind >= 0 ) { ctx->kprof_ms[kind] += ms; ctx->kprof_launches[kind]++; ctx->kprof_units[kind] += (uint64_t)k; }
            else if( kind == -3 ) { ctx->prof_lat_ms += ms; ctx->prof_lat_launches++; ctx->prof_lat_searches += k; }
            else if( k < 0 ) { ctx->prof_cell_ms += ms; ctx->prof_cell_launches++; ctx->prof_cells += (uint64_t)-k; }
            else { ctx->prof_ms += ms; ctx->prof_launches++; ctx->prof_searches += k; }
        }
    }
    ctx->prof_used = 0;
    ctx->prof_n.clear();
    ctx->prof_kind.clear();
    return X264HIP_OK;
}

static bool same_weight( const x264hip_weight &a, const x264hip_weight &b );
struct SearchReq
{
    int slot_b, slot_ref, list, dist_m1;
    WtD wt;
    unsigned keep_tag = 0; // non-zero: search under this tag (a field that so far existed on another rank only keeps its identity)
    int to_spare = 0;      // a speculative weighted search: into the slot's second list-0 field of that distance, its tag into wspec
};

static void report_timeout( x264hip_ctx *ctx )
{
    if( ctx->broken ) return; // (said once)
    const volatile unsigned *e = ctx->err_host;
    fprintf( stderr, "x264hip: in-kernel wait timed out: wait %u, search %u of the launch, row (group) %u, step/column 0x%x, waited for tag %u, last saw tag %u vector 0x%08x; "
                     "%u waves gave up, the last one row (group) %u of search %u; contexts open on the device: %d\n", e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[8], e[9] >> 16, e[9] & 0xFFFF,
             g_open_contexts[ctx->device & 63].load() );
    fprintf( stderr, "x264hip:   (last writer's view) xcc %u, queue %u, home %u; the same granule by compare-exchange: tag %u vector 0x%08x, by a system-scope load: tag %u vector 0x%08x; "
                     "queue's ticket counter %u\n", e[10] & 255, ( e[10] >> 8 ) & 255, e[10] >> 16, e[11], e[12], e[13], e[14], e[15] );
}
// wait for the stream, latch in-kernel timeouts (the flag lives in pinned host memory); every completed batch is now readable
static int sync_stream( x264hip_ctx *ctx )
{
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    ctx->batch_synced = ctx->batch_serial;
    if( *(volatile unsigned *)ctx->err_host )
    {
        report_timeout( ctx );
        ctx->broken = 1;
        return X264HIP_ETIMEOUT;
    }
    return X264HIP_OK;
}

// A speculative batch has been enqueued: give it the next serial and an event to wait on.
static int batch_close( x264hip_ctx *ctx )
{
    ctx->batch_serial++;
    HIPCK( hipEventRecord( ctx->batch_ev[ctx->batch_serial % x264hip_ctx::BATCH_EVS], ctx->stream ) );
    return X264HIP_OK;
}
// Wait until batch b is complete -- only that far, so that work queued behind it keeps running while the host decides.
static int batch_wait( x264hip_ctx *ctx, unsigned b )
{
    if( b <= ctx->batch_synced ) return X264HIP_OK;
    if( ctx->batch_serial - b >= (unsigned)x264hip_ctx::BATCH_EVS ) return sync_stream( ctx ); // its event has been reused
    HIPCK( hipEventSynchronize( ctx->batch_ev[b % x264hip_ctx::BATCH_EVS] ) );
    if( b > ctx->batch_synced ) ctx->batch_synced = b;
    if( *(volatile unsigned *)ctx->err_host )
    {
        report_timeout( ctx );
        ctx->broken = 1;
        return X264HIP_ETIMEOUT;
    }
    return X264HIP_OK;
}

template <typename T>
static int launch_searches_t( x264hip_ctx *ctx, const std::vector<SearchReq> &reqs )
{
    const LaP &P = ctx->P;
    const int n = (int)reqs.size();
    if( !n ) return X264HIP_OK;
    if( n > ctx->desc_cap ) return X264HIP_EINVAL;
    int rc = 0, ri = 0;
    if( ring_acquire( ctx->search_ring, &ri ) ) return X264HIP_EDEVICE;
    SearchDesc<T> *dh = (SearchDesc<T> *)ctx->search_ring.host[ri], *dd = (SearchDesc<T> *)ctx->search_ring.dev[ri];
    // The table holds the searches on unweighted planes first (in request = frame order), then the weighted ones: the kernels are
    // compiled once without and once with the weighting code.
    static const bool use_rows = getenv( "X264HIP_SEARCH" ) && !strcmp( getenv( "X264HIP_SEARCH" ), "rows" ); // A/B runs: the rows kernel for every launch
    std::vector<int> order( n );
    int n_plain = 0;
    for( int i = 0; i < n; i++ )
        if( !reqs[i].wt.on ) order[n_plain++] = i;
    for( int i = 0, k = n_plain; i < n; i++ )
        if( reqs[i].wt.on ) order[k++] = i;
    {
        // The searches that read the same REFERENCE stand next to each other in the table: their waves take neighbouring tickets, start
        // together and find some of each other's lines in the L2 (four fifths of a search's bytes are its reference's planes; request order
        // kept the searches of one SOURCE together): 3.13 against 3.23 us per search alone, 34.5-34.9 k against 33.9-34.0 k frames/s with
        // eight contexts (gpurun_out/r07h/order.txt; X264HIP_SEARCH_ORDER=req: request order).  Slots stand for frames: both run in put order.
        static const bool by_ref = !( getenv( "X264HIP_SEARCH_ORDER" ) && !strcmp( getenv( "X264HIP_SEARCH_ORDER" ), "req" ) );
        if( by_ref )
            std::stable_sort( order.begin(), order.begin() + n_plain, [&]( int a, int b ) {
                return reqs[a].slot_ref != reqs[b].slot_ref ? reqs[a].slot_ref < reqs[b].slot_ref : reqs[a].slot_b < reqs[b].slot_b; } );
    }
    // Which kernel ("two search kernels"): a launch that cannot fill the chip is as long as its dependency chain
    // -- W + 2 (H - 1) block searches one after the other -- whatever its width, so it goes to the LATENCY form of the search out of LDS
    // (me_latency_kernel: a wave per search and block row, ~5 x shorter per block, ~3 x more instructions per block);
    // everything larger goes to the throughput kernel (me_rows_kernel).
    static const int lat_waves = getenv( "X264HIP_LAT_WAVES" ) ? atoi( getenv( "X264HIP_LAT_WAVES" ) ) : 4096;
    // ... unless the device is shared: beside the launches of other contexts a small launch does not own the chip, what it leaves idle
    // is theirs to fill, and the latency form's three-fold instruction count per block comes out of everybody's throughput (eight
    // contexts, the 45 weighted searches of a fade per pass: 27.1 k against 26.0 k frames/s, profiles/r06_fade.txt)
    static const bool lat_always = getenv( "X264HIP_LAT_ALWAYS" ) != nullptr;
    // ... and only where the device is kept full: a context that has just sent a chip-filling launch is one of several working through
    // whole segments (the weighted searches of its fade follow within a launch or two).  A context whose launches are ALL small is an
    // encoder-paced stream -- a frame's searches per launch, its caller waiting for them -- and beside other such streams the device
    // is not full: there the latency form's shorter chain is what counts (eight paced streams 7.6 k -> 8.0 k frames/s, four 5.8 k -> 6.5 k,
    // gpurun_out/r08k).
    {
        const int n_rg = ( P.mb_h + ME_ROWS - 1 ) / ME_ROWS;
        if( (long long)std::max( n_plain, n - n_plain ) * n_rg >= 4096 ) ctx->launches_since_big = 0;
        else if( ctx->launches_since_big < 1000 ) ctx->launches_since_big++;
    }
    const bool crowded = !lat_always && g_open_contexts[ctx->device & 63].load() >= 4 && ctx->launches_since_big <= 4;
    const bool lat[2] = { !crowded && (long long)n_plain * P.mb_h <= lat_waves, !crowded && (long long)( n - n_plain ) * P.mb_h <= lat_waves };
    const bool rows[2] = { use_rows || !lat[0], use_rows || !lat[1] };
    std::vector<int> spare_planes;
    for( int i = 0; i < n; i++ )
    {
        const SearchReq &r = reqs[order[i]];
        FrameSlot &b = ctx->slots[r.slot_b], &rf = ctx->slots[r.slot_ref];
        SearchDesc<T> d;
        d.fenc0 = (const T *)( b.planes + 4 * ctx->plane_bytes );
        d.ref_strips = (const T *)( rf.planes + 4 * ctx->plane_bytes );
        d.refw_strips = nullptr;
        d.wt = r.wt;
        if( r.wt.on )
        {
            // the weighted copy of the reference's plane 0 is read by this launch only: the speculative searches of a launch take one pool
            // entry each and give it back at once (whoever takes it next writes it behind this launch, on the same stream)
            int wi = r.to_spare ? acquire_wplane( ctx, r.slot_b ) : b.wplane_idx;
            if( wi < 0 ) wi = b.wplane_idx = acquire_wplane( ctx, r.slot_b );
            if( wi < 0 ) return X264HIP_ENOMEM;
            T *wp = (T *)ctx->wplanes[wi];
            d.refw_strips = wp; // (filled by ONE launch for all the weighted searches of the table, below)
            if( r.to_spare ) spare_planes.push_back( wi );
        }
        const int fl = r.to_spare ? 2 : r.list;
        d.mvq = b.mvq[fl][r.dist_m1];
        d.costs = b.mvcost[fl][r.dist_m1];
        if( r.keep_tag )
            d.tag = r.keep_tag;
        else
        {
            d.tag = ctx->tag_serial++;
            if( !ctx->tag_serial ) ctx->tag_serial = 1;
        }
        if( r.to_spare )
            b.wspec[r.dist_m1].tag = d.tag;
        else
        {
            b.field_tag[r.list][r.dist_m1] = d.tag;
            b.field_remote[r.list][r.dist_m1] = 0;
        }
        d.pad = 0;
        dh[i] = d;
    }
    HIPCK( upload_async( ctx, dd, dh, (size_t)n * sizeof( SearchDesc<T> ), ctx->stream ) );
    if( n > n_plain )
        weight_strips_multi_kernel<T, SearchDesc<T>><<<dim3( ( P.plane_elems + 255 ) / 256, n - n_plain ), 256, 0, ctx->stream>>>( dd + n_plain, P.plane_elems, P.stride, P.pixel_max );
    // (the row tickets in sync_words are cleared by the last wave of the previous launch: me_search.h)
    hipEvent_t e0 = ctx->ev_start, e1 = ctx->ev_stop;
    if( ctx->prof_on )
    {
        if( ctx->prof_used + 2 > (int)ctx->prof_ev.size() )
        {
            rc = prof_drain( ctx );
            if( rc ) return rc;
        }
        e0 = ctx->prof_ev[ctx->prof_used]; e1 = ctx->prof_ev[ctx->prof_used + 1];
        while( ctx->prof_kind.size() < ctx->prof_n.size() ) ctx->prof_kind.push_back( -1 );
        // (-1: a launch that can fill the chip -- more waves than the 4 096 wave slots -- on the throughput kernel: what the roofline is
        //  about; -3: a small launch, as long as its dependency chain whatever runs it, counted for itself)
        const int n_rg = ( P.mb_h + ME_ROWS - 1 ) / ME_ROWS;
        const bool fills = ( !n_plain || ( rows[0] && (long long)n_plain * n_rg >= 4096 ) ) && ( n == n_plain || ( rows[1] && (long long)( n - n_plain ) * n_rg >= 4096 ) );
        ctx->prof_n.push_back( n ); ctx->prof_kind.push_back( fills ? -1 : -3 );
        ctx->prof_used += 2;
    }
    HIPCK( hipEventRecord( e0, ctx->stream ) );
    {
        const bool hex = P.me_method == X264HIP_ME_HEX, r4 = P.subpel_refine >= 3;
        const int mode = !r4 && !P.mbcmp_satd && !P.fpelcmp_satd ? 0 : r4 && P.mbcmp_satd ? ( P.fpelcmp_satd ? 2 : 1 ) : 3;
        MeQueues Q;
        // latency kernel: one wave per (search, block row); rows kernel: one wave per (search, group of ME_ROWS block rows).  Both are
        // specialised on the search pattern, the sub-pel depth and on whether their searches read weighted references
        const int n_rowgroups = ( P.mb_h + ME_ROWS - 1 ) / ME_ROWS;
        for( int part = 0; part < 2; part++ )
        {
            const int first = part ? n_plain : 0, count = part ? n - n_plain : n_plain;
            if( !count ) continue;
            for( int q = 0; q <= ME_QUEUES; q++ )
                Q.base[q] = (int)( (long long)count * q / ME_QUEUES ); // the ticket queues hand out searches; contiguous groups: the table is in frame order
            unsigned *tickets = ctx->sync_words + part * ME_QUEUES * ME_QUEUE_STRIDE;
            const int grid_rows = count * n_rowgroups, grid_lat = count * ( ( P.mb_h + ME_LAT_ROWS - 1 ) / ME_LAT_ROWS );
            // (spin limit: ~8 s of polling for a vector of the row below before a wave gives up and the context is marked broken; 2 s
            //  were once not enough beside fifteen other contexts, gpurun_out/r07i -- nothing a wave waits for is that far away unless the
            //  device itself stands still)
#define ME_ARGS_ROWS P, dd + first, Q, tickets, ctx->err_host, 1u << 24, ctx->me_prof
#define ME_LAUNCH( HEXV, MODEV ) do { \
                if( rows[part] ) { \
                    if( part ) me_rows_kernel<T, HEXV, MODEV, 1><<<grid_rows, 64, 0, ctx->stream>>>( ME_ARGS_ROWS ); \
                    else me_rows_kernel<T, HEXV, MODEV, 0><<<grid_rows, 64, 0, ctx->stream>>>( ME_ARGS_ROWS ); \
                } else { \
                    if( part ) me_latency_kernel<T, HEXV, MODEV, 1, ME_LAT_ROWS><<<grid_lat, 64 * ME_LAT_ROWS, 0, ctx->stream>>>( ME_ARGS_ROWS ); \
                    else me_latency_kernel<T, HEXV, MODEV, 0, ME_LAT_ROWS><<<grid_lat, 64 * ME_LAT_ROWS, 0, ctx->stream>>>( ME_ARGS_ROWS ); \
                } } while( 0 )
            switch( 4 * hex + mode )
            {
                case 0: ME_LAUNCH( 0, 0 ); break;
                case 1: ME_LAUNCH( 0, 1 ); break;
                case 2: ME_LAUNCH( 0, 2 ); break;
                case 3: ME_LAUNCH( 0, 3 ); break;
                case 4: ME_LAUNCH( 1, 0 ); break;
                case 5: ME_LAUNCH( 1, 1 ); break;
                case 6: ME_LAUNCH( 1, 2 ); break;
                default: ME_LAUNCH( 1, 3 ); break;
            }
#undef ME_LAUNCH
#undef ME_ARGS_ROWS
        }
    }
    HIPCK( hipEventRecord( e1, ctx->stream ) );
    if( ring_commit( ctx->search_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
    HIPCK( hipGetLastError() );
    for( int wi : spare_planes ) ctx->wplane_owner[wi] = -1;
    ctx->ev_valid = 1;
    ctx->last_n_search = n;
    ctx->last_n_blocks = n * ctx->n_mb;
    ctx->counters[0] += n;
    return X264HIP_OK;
}

static int launch_searches( x264hip_ctx *ctx, const std::vector<SearchReq> &reqs )
{
    return ctx->p.bit_depth == 8 ? launch_searches_t<uint8_t>( ctx, reqs ) : launch_searches_t<uint16_t>( ctx, reqs );
}

// ---- cost cells ---------------------------------------------------------------------------------------
// Descriptor of the cell (slot_b, d0, d1) with the fields as they stand now.  kind: 0 intra sums only, 1 real cell.
template <typename T>
static CellArgs make_cell( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int with_intra, int ref1_l0_valid,
                           int sums_only, int to_spare = 0, int wfield = 0 /* list 0 from the slot's speculative weighted field */ )
{
    const LaP &P = ctx->P;
    FrameSlot &b = ctx->slots[slot_b], &f0 = ctx->slots[slot_p0], &f1 = ctx->slots[slot_p1];
    const int intra_only = d0 == 0 && d1 == 0, b_bidir = d1 > 0;
    const int idx = d0 * ( ctx->p.bframes + 2 ) + d1;
    CellArgs A;
    memset( &A, 0, sizeof( A ) );
    A.b_bidir = b_bidir; A.with_intra = with_intra; A.is_intra_only = intra_only; A.ref1_l0_valid = b_bidir && ref1_l0_valid;
    A.sums_only = sums_only;
    A.dist_scale_factor = intra_only ? 128 : ( ( d0 << 8 ) + ( ( d0 + d1 ) >> 1 ) ) / ( d0 + d1 );
    if( !intra_only )
    {
        // wfield bit 0: list 0 from this frame's speculative weighted field, bit 1: the list-1 reference's vectors from ITS speculative weighted field
        A.mvq0 = b.mvq[( wfield & 1 ) ? 2 : 0][d0 - 1]; A.costs0 = b.mvcost[( wfield & 1 ) ? 2 : 0][d0 - 1];
        if( b_bidir )
        {
            A.mvq1 = b.mvq[1][d1 - 1]; A.costs1 = b.mvcost[1][d1 - 1];
            A.ref1_l0 = A.ref1_l0_valid ? f1.mvq[( wfield & 2 ) ? 2 : 0][d0 + d1 - 1] : nullptr;
            A.fenc0 = b.planes + 4 * ctx->plane_bytes; A.ref0_0 = f0.planes + 4 * ctx->plane_bytes; A.ref1_0 = f1.planes + 4 * ctx->plane_bytes;
        }
    }
    A.intra_cost = b.lowres_costs; // cell [0][0] (frame.c:283)
    A.inv_qscale = b.inv_qscale;
    A.lowres_costs = b.lowres_costs + (size_t)idx * ctx->n_mb;
    A.row_satds = b.row_satds + (size_t)idx * P.mb_h;
    A.row_satds_intra = b.row_satds;
    A.blk = b.blk + (size_t)idx * ctx->n_mb;
    A.acc = ctx->cell_acc_host + ( (size_t)slot_b * ctx->n_cells + idx ) * 8; // pinned, device-visible
    A.acc_dev = b.cell_sums + (size_t)idx * 8;
    A.work = b.cell_work + (size_t)idx * 8;
    if( !to_spare )
        b.cell_at[idx] = idx; // an evaluation into the cell's own place is what the cell is from now on
    if( to_spare )
    {
        // the cell's spare: its own map, block words, row sums and sums; the intra row sums are the frame's (same values)
        const int sp = to_spare == 2 ? ctx->spare2_at[idx] : ctx->spare_at[idx];
        A.lowres_costs = b.lowres_costs + (size_t)sp * ctx->n_mb;
        A.row_satds = b.row_satds + (size_t)sp * P.mb_h;
        A.blk = b.blk + (size_t)sp * ctx->n_mb;
        A.acc = ( to_spare == 2 ? ctx->cell_alt2_host : ctx->cell_alt_host ) + ( (size_t)slot_b * ctx->n_cells + idx ) * 8;
        A.acc_dev = b.cell_sums + (size_t)sp * 8;
        A.work = b.cell_work + (size_t)sp * 8;
    }
    return A;
}

// workgroups per cell of cell_reduce_kernel: enough of them over the whole launch to spread it over the chip, each with at least 4 rows
static int reduce_bands( const LaP &P, int n_cells_in_launch )
{
    const int by_rows = std::max( 1, P.mb_h / 4 ), by_chip = std::max( 1, 512 / std::max( 1, n_cells_in_launch ) );
    return std::min( std::min( by_rows, by_chip ), 32 );
}

struct SpecCell
{
    int slot_p0, slot_p1, slot_b, d0, d1, sums_only, ref1_valid;
    int to_spare = 0;
    int dual = 0; // B cell with the list-1 reference's vectors: the outcome WITHOUT them is produced in the same pass, into the cell's spare
    int wfield = 0; // P cell over the slot's speculative weighted list-0 field (into the cell's spare)
};

// one batch: all P cells, all B cells, then one reduction launch (a workgroup per cell)
template <typename T>
static int launch_cells_t( x264hip_ctx *ctx, const std::vector<SpecCell> &cells )
{
    const LaP &P = ctx->P;
    const size_t per_launch = (size_t)ctx->cell_desc_cap / 2; // a cell evaluated both ways takes a second descriptor (its spare half's sums)
    for( size_t o = 0; o < cells.size(); o += per_launch )
    {
        const int n = (int)std::min( cells.size() - o, per_launch );
        int ri = 0;
        if( ring_acquire( ctx->cell_ring, &ri ) ) return X264HIP_EDEVICE;
        CellArgs *dh = (CellArgs *)ctx->cell_ring.host[ri], *dd = (CellArgs *)ctx->cell_ring.dev[ri];
        // order: [P cells | B cells | sums-only]; the reduction walks all of them
        std::vector<const SpecCell *> ord;
        int n_p = 0, n_b = 0;
        for( int pass = 0; pass < 3; pass++ )
            for( int i = 0; i < n; i++ )
            {
                const SpecCell &c = cells[o + i];
                int kind = c.sums_only ? 2 : c.d1 > 0 ? 1 : 0;
                if( kind != pass ) continue;
                ord.push_back( &c );
                if( pass == 0 ) n_p++;
                if( pass == 1 ) n_b++;
            }
        int n_red = n; // descriptors the reduction walks: every cell, then the spare halves of the cells evaluated both ways
        for( int i = 0; i < n; i++ )
        {
            const SpecCell &c = *ord[i];
            dh[i] = make_cell<T>( ctx, c.slot_p0, c.slot_p1, c.slot_b, c.d0, c.d1, 1, c.ref1_valid, c.sums_only, c.to_spare, c.wfield );
            if( c.dual )
            {
                const CellArgs S = make_cell<T>( ctx, c.slot_p0, c.slot_p1, c.slot_b, c.d0, c.d1, 1, 0, 0, 1 );
                dh[i].dual = 1; dh[i].lowres_costs2 = S.lowres_costs; dh[i].blk2 = S.blk;
                dh[n_red++] = S;
            }
        }
        HIPCK( upload_async( ctx, dd, dh, (size_t)n_red * sizeof( CellArgs ), ctx->stream ) );
        CellArgs none;
        memset( &none, 0, sizeof( none ) );
        hipEvent_t pe1 = nullptr;
        if( ctx->prof_on )
        {
            if( ctx->prof_used + 2 > (int)ctx->prof_ev.size() )
            {
                int rc = prof_drain( ctx );
                if( rc ) return rc;
            }
            HIPCK( hipEventRecord( ctx->prof_ev[ctx->prof_used], ctx->stream ) );
            pe1 = ctx->prof_ev[ctx->prof_used + 1];
            while( ctx->prof_kind.size() < ctx->prof_n.size() ) ctx->prof_kind.push_back( -1 );
            ctx->prof_n.push_back( -n ); ctx->prof_kind.push_back( -1 );
            ctx->prof_used += 2;
        }
        if( n_p )
            KPROF( X264HIP_KPROF_CELL_P, n_p, ( cell_p_kernel<<<dim3( ( ctx->n_mb + 255 ) / 256, n_p ), 256, 0, ctx->stream>>>( P, dd, none ) ) );
        if( n_b )
            KPROF( X264HIP_KPROF_CELL_B, n_b, ( cell_b_kernel<T><<<dim3( ( P.mb_w + CELLB_BPW - 1 ) / CELLB_BPW, P.mb_h, n_b ), 64, 0, ctx->stream>>>( P, dd + n_p, none ) ) );
        KPROF( X264HIP_KPROF_CELL_REDUCE, n_red, ( cell_reduce_kernel<<<dim3( n_red, reduce_bands( P, n_red ) ), 256, 0, ctx->stream>>>( P, dd, none ) ) ); // sums go straight to pinned host memory
        if( pe1 )
            HIPCK( hipEventRecord( pe1, ctx->stream ) );
        HIPCK( hipGetLastError() );
        if( ring_commit( ctx->cell_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_gop_hint( x264hip_ctx *ctx, int anchor_frame, int period )
{
    if( !ctx || period < 0 ) return X264HIP_EINVAL;
    ctx->hint_anchor = anchor_frame;
    ctx->hint_period = period <= X264HIP_BFRAME_MAX + 1 ? period : 0;
    return X264HIP_OK;
}

extern "C" int x264hip_prefetch( x264hip_ctx *ctx, const int *slots, const int *frame_numbers, int n )
{
    return x264hip_prefetch_ex( ctx, slots, frame_numbers, n, 0 );
}

extern "C" int x264hip_prefetch_ex( x264hip_ctx *ctx, const int *slots, const int *frame_numbers, int n, int flags )
{
    if( !ctx || !slots || !frame_numbers || n < 0 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    std::vector<SearchReq> reqs;
    const WtD none = { 0, 1, 0, 0 };
    const int bf = ctx->p.bframes, nstride = bf + 2;
    static const bool no_learn = getenv( "X264HIP_NO_CLASS_LEARNING" ) != nullptr; // debugging aid: speculate everything
    const bool learned = !no_learn && ctx->n_requests >= x264hip_ctx::LEARN_REQUESTS;
    static const bool no_pos = getenv( "X264HIP_NO_POSITION_CLASSES" ) != nullptr; // debugging aid: ignore x264hip_gop_hint
    // A class is speculated when at least this share of the frames seen so far asked for it (percent; 0 = any request at all, the
    // default for fields: a field that is missing costs a whole search launch -- a dependency chain of W + 2 H steps -- on demand,
    // a missing cell only a cell launch).  X264HIP_FIELD_RATE / X264HIP_CELL_RATE override them (experiments).
    static const unsigned field_rate = getenv( "X264HIP_FIELD_RATE" ) ? (unsigned)atoi( getenv( "X264HIP_FIELD_RATE" ) ) : 0;
    static const unsigned cell_rate = getenv( "X264HIP_CELL_RATE" ) ? (unsigned)atoi( getenv( "X264HIP_CELL_RATE" ) ) : 0;
    const uint64_t frames_seen = ctx->counters[3];
    // a class is worth speculating for a frame at a known position if at least one in ten of the frames seen there asked for it
    auto pos_wants_field = [&]( const FrameSlot &f, int list, int dm1 ) {
        const int k = f.pos_key;
        if( !k || ctx->pos_frames[k] < (uint32_t)x264hip_ctx::POS_LEARN_FRAMES ) return true;
        return ctx->pos_field_req[( (size_t)k * 2 + list ) * ( X264HIP_BFRAME_MAX + 1 ) + dm1] * 10 >= ctx->pos_frames[k];
    };
    auto pos_wants_cell = [&]( const FrameSlot &f, int idx ) {
        const int k = f.pos_key;
        if( !k || ctx->pos_frames[k] < (uint32_t)x264hip_ctx::POS_LEARN_FRAMES ) return true;
        return ctx->pos_cell_req[(size_t)k * ctx->n_cells + idx] * 10 >= ctx->pos_frames[k];
    };
    for( int i = 0; i < n; i++ )
    {
        if( !slot_ok( ctx, slots[i] ) || !ctx->slots[slots[i]].in_use ) return X264HIP_ESTATE;
        ctx->slots[slots[i]].frame_no = frame_numbers[i];
        {
            // the position this frame is expected to take between two anchors (0 = an anchor itself), under the caller's current hint
            FrameSlot &f = ctx->slots[slots[i]];
            const int per = no_pos ? 0 : ctx->hint_period;
            int key = 0;
            if( per > 0 && per <= X264HIP_BFRAME_MAX + 1 && frame_numbers[i] > ctx->hint_anchor )
                key = per * 32 + ( frame_numbers[i] - ctx->hint_anchor ) % per + 1;
            // a frame keeps the key it was first speculated under unless that was "no expectation": its requests are counted there
            if( !f.pos_key ) f.pos_key = key;
        }
        for( int j = 0; j < n; j++ )
        {
            const int d = frame_numbers[j] - frame_numbers[i]; // reference j relative to source i
            if( !d || abs( d ) > bf + 1 ) continue;
            const int list = d > 0, dm1 = abs( d ) - 1;
            if( list && !bf ) continue;
            FrameSlot &b = ctx->slots[slots[i]];
            if( b.field_ready[list][dm1] || b.field_prefetched[list][dm1] ) continue;
            if( !( ctx->field_allowed[list] >> dm1 & 1 ) ) continue;  // a class the caller has ruled out (x264hip_spec_classes)
            if( learned && !ctx->field_req[list][dm1] ) continue; // a class this caller never asks for
            if( learned && field_rate && (uint64_t)ctx->field_req[list][dm1] * 100 < field_rate * frames_seen ) continue; // ... or too rarely
            if( !pos_wants_field( b, list, dm1 ) ) continue;       // ... or hardly ever for a frame at this position
            if( flags & X264HIP_PREFETCH_CELLS_ONLY ) continue;     // the fields come from elsewhere (x264hip_import_field)
            b.field_prefetched[list][dm1] = 1;
            ctx->field_spec[list][dm1]++;
            reqs.push_back( SearchReq{ slots[i], slots[j], list, dm1, none } );
        }
    }
    // split into launches that fit the descriptor table
    for( size_t o = 0; o < reqs.size(); o += ctx->desc_cap )
    {
        std::vector<SearchReq> part( reqs.begin() + o, reqs.begin() + std::min( reqs.size(), o + (size_t)ctx->desc_cap ) );
        int r = launch_searches( ctx, part );
        if( r ) return r;
    }
    // speculative cells over the fields that now exist: intra sums, P cells, B cells whose three inputs are present
    // (unweighted speculative or already claimed fields).  Entries remember the field tags they consumed.
    std::vector<SpecCell> cells;
    if( getenv( "X264HIP_NO_SPEC_CELLS" ) ) return X264HIP_OK; // debugging aid: searches only
    static const bool no_dual = getenv( "X264HIP_NO_DUAL" ) != nullptr; // A/B runs: one variant per B cell, chosen by the requests so far
    std::vector<int> by_number; // frame number -> index in the lists (small window: linear scans are fine)
    auto find = [&]( int number ) { for( int k = 0; k < n; k++ ) if( frame_numbers[k] == number ) return k; return -1; };
    auto has_field = [&]( FrameSlot &f, int list, int dm1 ) { return ( f.field_ready[list][dm1] || f.field_prefetched[list][dm1] ) && !f.field_remote[list][dm1]; };
    for( int i = 0; i < n; i++ )
    {
        FrameSlot &b = ctx->slots[slots[i]];
        CellEntry &e0 = b.cells[0];
        if( !e0.valid && !e0.requested )
        {
            e0.valid = 1; e0.batch = ctx->batch_serial + 1; e0.tag0 = e0.tag1 = e0.tagr = 0;
            cells.push_back( SpecCell{ slots[i], slots[i], slots[i], 0, 0, 1, 0 } );
        }
        for( int d0 = 1; d0 <= bf + 1; d0++ )
        {
            const int j0 = find( frame_numbers[i] - d0 );
            if( j0 < 0 || !has_field( b, 0, d0 - 1 ) ) continue;
            for( int d1 = 0; d0 + d1 <= bf + 1; d1++ )
            {
                CellEntry &e = b.cells[d0 * nstride + d1];
                if( e.valid || e.requested ) continue;
                if( !ctx->cell_allowed[d0 * nstride + d1] ) continue;
                if( learned && !ctx->cell_req[d0 * nstride + d1] ) continue;
                if( learned && cell_rate && (uint64_t)ctx->cell_req[d0 * nstride + d1] * 100 < cell_rate * frames_seen ) continue;
                if( !pos_wants_cell( b, d0 * nstride + d1 ) ) continue;
                int j1 = j0, variant = 1, dual = 0;
                unsigned t1 = 0, tr = 0;
                if( d1 )
                {
                    j1 = find( frame_numbers[i] + d1 );
                    if( j1 < 0 || !has_field( b, 1, d1 - 1 ) ) continue;
                    FrameSlot &f1 = ctx->slots[slots[j1]];
                    // Which way the caller will ask for a B cell -- with or without the list-1 reference's own list-0 vectors
                    // (slicetype.c:629) -- depends on the order of its requests.  The two outcomes share their three candidate costs, so
                    // one pass of the cell kernel produces both: with the vectors into the cell's own place, without them into its spare
                    // (FrameSlot alts); if the reference's field does not exist only the second is possible.  (Rounds 2-3 speculated the
                    // variant asked for more often so far and the other one into ONE spare per frame: a fresh context -- every pass of a
                    // single stream -- knew nothing, and 766 of the 2 281 requests of a 250-frame 4K pass were evaluated on demand.)
                    const bool with_r = has_field( f1, 0, d0 + d1 - 1 );
                    variant = with_r ? 1 : 0;
                    dual = with_r && !no_dual;
                    if( no_dual )
                    {
                        const uint32_t *rq = ctx->variant_req[d0 * nstride + d1];
                        variant = rq[0] > rq[1] ? 0 : 1;
                        if( variant && !with_r ) continue;
                    }
                    t1 = b.field_tag[1][d1 - 1];
                    tr = variant ? f1.field_tag[0][d0 + d1 - 1] : 0;
                }
                e.valid = 1; e.batch = ctx->batch_serial + 1; e.variant = (unsigned char)variant;
                e.tag0 = b.field_tag[0][d0 - 1]; e.tag1 = t1; e.tagr = tr;
                ctx->cell_spec[d0 * nstride + d1]++;
                SpecCell sc{ slots[j0], slots[d1 ? j1 : i], slots[i], d0, d1, 0, variant };
                sc.dual = dual;
                cells.push_back( sc );
                if( dual )
                {
                    CellEntry &a = b.alts[d0 * nstride + d1];
                    a = CellEntry();
                    a.valid = 1; a.batch = ctx->batch_serial + 1; a.variant = 0;
                    a.tag0 = e.tag0; a.tag1 = t1; a.tagr = 0;
                    ctx->counters[14]++;
                }
            }
        }
    }
    if( !cells.empty() )
    {
        // entries carry batch_serial + 1; the serial moves only once the batch is enqueued
        int r = ctx->p.bit_depth == 8 ? launch_cells_t<uint8_t>( ctx, cells ) : launch_cells_t<uint16_t>( ctx, cells );
        if( r ) return r;
        r = batch_close( ctx );
        if( r ) return r;
        ctx->counters[5] += cells.size();
    }
    return X264HIP_OK;
}

// ---- window shard: data that lives on the owner rank until somebody here needs it ----------------------------------------------
// The fields a cell (p0, p1, b) reads, searched locally if this context only knows them by tag (x264hip_fields_remote).  The result is
// what the owner computed: a search is a pure function of its two frames.
static int ensure_fields_local( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int with_ref1_l0 )
{
    std::vector<SearchReq> reqs;
    const WtD none = { 0, 1, 0, 0 };
    auto need = [&]( int slot_f, int slot_ref, int list, int dm1 ) {
        FrameSlot &f = ctx->slots[slot_f];
        if( f.field_remote[list][dm1] )
        {
            reqs.push_back( SearchReq{ slot_f, slot_ref, list, dm1, none, f.field_tag[list][dm1] } );
            ctx->counters[8]++;
        }
    };
    if( d0 > 0 ) need( slot_b, slot_p0, 0, d0 - 1 );
    if( d1 > 0 )
    {
        need( slot_b, slot_p1, 1, d1 - 1 );
        if( with_ref1_l0 ) need( slot_p1, slot_p0, 0, d0 + d1 - 1 );
    }
    if( reqs.empty() ) return X264HIP_OK;
    const uint64_t n_before = ctx->counters[0];
    int r = launch_searches( ctx, reqs );
    ctx->counters[0] = n_before; // counted under [8]
    return r;
}

template <typename T>
static CellArgs make_cell( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int with_intra, int ref1_l0_valid, int sums_only, int to_spare, int wfield );

// The per-block map of a cell whose sums came from its owner rank, evaluated here after all (MB-tree reads it, so do
// x264hip_frame_cost_recalculate and the getters): fields first, then the cell kernel; the sums are known already.
template <typename T>
static int ensure_cell_local_t( x264hip_ctx *ctx, int slot_b, int d0, int d1 )
{
    FrameSlot &b = ctx->slots[slot_b];
    CellEntry &e = b.cells[d0 * ( ctx->p.bframes + 2 ) + d1];
    if( !e.map_remote ) return X264HIP_OK;
    if( !slot_ok( ctx, e.slot_p0 ) || !slot_ok( ctx, e.slot_p1 ) || !ctx->slots[e.slot_p0].in_use || !ctx->slots[e.slot_p1].in_use ) return X264HIP_ESTATE;
    const int with_l0 = d1 > 0 && e.variant;
    int r = ensure_fields_local( ctx, e.slot_p0, e.slot_p1, slot_b, d0, d1, with_l0 );
    if( r ) return r;
    const CellArgs A = make_cell<T>( ctx, e.slot_p0, e.slot_p1, slot_b, d0, d1, 0, with_l0, 0 );
    if( d1 > 0 )
        cell_b_kernel<T><<<dim3( ( ctx->P.mb_w + CELLB_BPW - 1 ) / CELLB_BPW, ctx->P.mb_h, 1 ), 64, 0, ctx->stream>>>( ctx->P, nullptr, A );
    else
        cell_p_kernel<<<dim3( ( ctx->n_mb + 255 ) / 256, 1 ), 256, 0, ctx->stream>>>( ctx->P, nullptr, A );
    HIPCK( hipGetLastError() );
    e.map_remote = 0;
    ctx->counters[9]++;
    return X264HIP_OK;
}
static int ensure_cell_local( x264hip_ctx *ctx, int slot_b, int d0, int d1 )
{
    return ctx->p.bit_depth == 8 ? ensure_cell_local_t<uint8_t>( ctx, slot_b, d0, d1 ) : ensure_cell_local_t<uint16_t>( ctx, slot_b, d0, d1 );
}
// a field this context only knows by tag, for a getter: the reference frame is found by its number
static int ensure_field_local_by_number( x264hip_ctx *ctx, int slot, int list, int dm1 )
{
    FrameSlot &f = ctx->slots[slot];
    if( !f.field_remote[list][dm1] ) return X264HIP_OK;
    const int want = f.frame_no + ( list ? dm1 + 1 : -( dm1 + 1 ) );
    for( int i = 0; i < (int)ctx->slots.size(); i++ )
        if( ctx->slots[i].in_use && ctx->slots[i].frame_no == want )
        {
            std::vector<SearchReq> reqs( 1, SearchReq{ slot, i, list, dm1, WtD{ 0, 1, 0, 0 }, f.field_tag[list][dm1] } );
            const uint64_t n_before = ctx->counters[0];
            int r = launch_searches( ctx, reqs );
            ctx->counters[0] = n_before;
            ctx->counters[8]++;
            return r;
        }
    return X264HIP_ESTATE;
}

// ---- evaluation -------------------------------------------------------------------------------------
template <typename T>
static int frame_cost_t( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int d0, int d1, const int do_search[2],
                         const x264hip_weight *w, int with_intra, int ref1_l0_valid, x264hip_cost *out )
{
    const LaP &P = ctx->P;
    FrameSlot &b = ctx->slots[slot_b], &f1 = ctx->slots[slot_p1];
    const int intra_only = d0 == 0 && d1 == 0;
    const int b_bidir = d1 > 0;
    const int idx = d0 * ( ctx->p.bframes + 2 ) + d1;
    std::vector<SearchReq> reqs;
    ctx->n_requests++;
    ctx->cell_req[idx]++;
    b.req_cells[idx] = 1;
    if( !intra_only )
    {
        const WtD wt = make_wt( ctx, w );
        if( do_search[0] ) { ctx->field_req[0][d0 - 1]++; b.req_fields[0] |= 1u << ( d0 - 1 ); }
        if( b_bidir && do_search[1] ) { ctx->field_req[1][d1 - 1]++; b.req_fields[1] |= 1u << ( d1 - 1 ); }
        if( do_search[0] )
        {
            FrameSlot::WSpec &ws = b.wspec[d0 - 1];
            if( b.field_prefetched[0][d0 - 1] && !wt.on )
                ctx->counters[2]++;
            else if( wt.on && ws.valid && same_weight( ws.w, *w ) )
            {
                // the field was searched ahead of time on this very weighted reference (x264hip_prefetch_weighted_fields): it becomes the
                // slot's list-0 field of this distance, the unweighted speculative one (and every cell evaluated over it) is left behind
                int r = batch_wait( ctx, ws.batch );
                if( r ) return r;
                std::swap( b.mvq[0][d0 - 1], b.mvq[2][d0 - 1] );
                std::swap( b.mvcost[0][d0 - 1], b.mvcost[2][d0 - 1] );
                b.field_tag[0][d0 - 1] = ws.tag;
                b.field_remote[0][d0 - 1] = 0;
                ctx->weighted_claimed++;
            }
            else
                reqs.push_back( SearchReq{ slot_b, slot_p0, 0, d0 - 1, wt } );
            ws.valid = 0;
            b.field_prefetched[0][d0 - 1] = 0;
            b.field_ready[0][d0 - 1] = 1;
        }
        else if( !b.field_ready[0][d0 - 1] )
            return X264HIP_ESTATE;
        if( b_bidir )
        {
            if( do_search[1] )
            {
                if( b.field_prefetched[1][d1 - 1] )
                    ctx->counters[2]++;
                else
                    reqs.push_back( SearchReq{ slot_b, slot_p1, 1, d1 - 1, WtD{ 0, 1, 0, 0 } } );
                b.field_prefetched[1][d1 - 1] = 0;
                b.field_ready[1][d1 - 1] = 1;
            }
            else if( !b.field_ready[1][d1 - 1] )
                return X264HIP_ESTATE;
            if( ref1_l0_valid && !f1.field_ready[0][d0 + d1 - 1] )
                return X264HIP_ESTATE;
        }
        ctx->counters[13] += reqs.size(); // searches on demand
        int r = launch_searches( ctx, reqs );
        if( r ) return r;
    }
    // speculative result usable?  Same input fields (tags) as the cell would read now.
    CellEntry &e = b.cells[idx];
    const unsigned t0 = intra_only ? 0 : b.field_tag[0][d0 - 1];
    const unsigned t1 = b_bidir ? b.field_tag[1][d1 - 1] : 0;
    const unsigned tr = b_bidir && ref1_l0_valid ? f1.field_tag[0][d0 + d1 - 1] : 0;
    const bool hit = e.valid && e.tag0 == t0 && e.tag1 == t1 && e.tagr == tr && ( !b_bidir || e.variant == ( ref1_l0_valid ? 1 : 0 ) );
    if( b_bidir )
        ctx->variant_req[idx][ref1_l0_valid ? 1 : 0]++;
    for( int which = 0; which < 2 && !hit && !intra_only; which++ )
    {
        // another evaluation of this cell sits in one of its spares -- the other variant of a B cell (with / without the list-1 reference's
        // vectors), or the cell over fields that were searched ahead of time on weighted references -- and it read exactly the fields
        // the cell would read now: the spare becomes the cell (nothing to wait for beyond the batch that evaluated it) and the answer
        // comes from its sums
        CellEntry &a = which ? b.alts2[idx] : b.alts[idx];
        if( !( a.valid && a.tag0 == t0 && a.tag1 == t1 && a.tagr == tr && ( !b_bidir || a.variant == ( ref1_l0_valid ? 1 : 0 ) ) ) )
            continue;
        int r = batch_wait( ctx, a.batch );
        if( r ) return r;
        b.cell_at[idx] = which ? ctx->spare2_at[idx] : ctx->spare_at[idx]; // every reader goes through cell_at (no copy, no launch)
        const int *ra = ( which ? ctx->cell_alt2_host : ctx->cell_alt_host ) + ( (size_t)slot_b * ctx->n_cells + idx ) * 8;
        out->cost_est = ra[0]; out->cost_est_aq = ra[1]; out->intra_mbs = ra[2];
        out->intra_cost_est = ra[3]; out->intra_cost_est_aq = ra[4];
        e.requested = 1; e.valid = 0;
        e.variant = a.variant;
        e.map_remote = a.map_remote; e.slot_p0 = a.slot_p0; e.slot_p1 = a.slot_p1; // (window shard: the spare's map may still be with its owner)
        a.valid = 0;
        ctx->counters[4]++; ctx->counters[1]++;
        if( which ) ctx->weighted_cells_used++; else ctx->counters[15]++;
        return X264HIP_OK;
    }
    const int was_valid = e.valid, was_requested = e.requested;
    e.requested = 1;
    e.valid = 0;
    const int *res = ctx->cell_acc_host + ( (size_t)slot_b * ctx->n_cells + idx ) * 8;
    CellArgs A = make_cell<T>( ctx, slot_p0, slot_p1, slot_b, d0, d1, with_intra, ref1_l0_valid, 0 );
    if( hit )
    {
        ctx->counters[4]++;
        static const bool trace_hit = getenv( "X264HIP_TRACE_MISS" ) != nullptr;
        if( trace_hit && b_bidir )
            fprintf( stderr, "hit b=%d d0=%d d1=%d ref1_ok=%d\n", b.frame_no, d0, d1, ref1_l0_valid );
        int r = batch_wait( ctx, e.batch );
        if( r ) return r;
        if( intra_only && res[5] )
        {
            // the sums are known; the map still has to take the reference's 14-bit clamp (aliases the intra costs, which queued
            // MB-tree lists read as they were when they were handed over).  res[5] = the frame's intra costs above 14 bits, counted with
            // the sums (cell_reduce_kernel): none -- nearly every frame -- and the clamp is the identity: no launch, and above all no
            // flush of the MB-tree queue, which this request used to cut into launches of one to six lists (one intra-only request per
            // mini-GOP: a flush and a clamp launch per MB-tree call, profiles/r04_bench_kernel_stats.csv)
            int rf = mbt_flush_at( ctx, __LINE__ );
            if( rf ) return rf;
            cell_p_kernel<<<dim3( ( ctx->n_mb + 255 ) / 256, 1 ), 256, 0, ctx->stream>>>( P, nullptr, A );
            HIPCK( hipGetLastError() );
        }
    }
    else
    {
        ctx->counters[7]++;
        if( was_requested )
        {
            // a cell the caller asks for AGAIN may be one a queued MB-tree list reads (x264hip_mbtree queues lists and reads their inputs
            // when they are launched): the lists go first, then the map is rewritten
            int rf = mbt_flush_at( ctx, __LINE__ );
            if( rf ) return rf;
        }
        if( !intra_only )
        {
            // inputs that so far exist on another rank only (window shard) are searched here now, under the tag they are known by
            int r = ensure_fields_local( ctx, slot_p0, slot_p1, slot_b, d0, d1, ref1_l0_valid );
            if( r ) return r;
        }
        e.map_remote = 0;
        static const bool trace_miss = getenv( "X264HIP_TRACE_MISS" ) != nullptr;
        if( trace_miss )
            fprintf( stderr, "miss b=%d d0=%d d1=%d valid=%d tags have %u/%u/%u want %u/%u/%u ref1_ok=%d wi=%d search=%d,%d w=%d\n", b.frame_no, d0, d1, was_valid,
                     e.tag0, e.tag1, e.tagr, t0, t1, tr, ref1_l0_valid, with_intra, do_search[0], do_search[1], w && w->on );
        if( b_bidir )
            cell_b_kernel<T><<<dim3( ( P.mb_w + CELLB_BPW - 1 ) / CELLB_BPW, P.mb_h, 1 ), 64, 0, ctx->stream>>>( P, nullptr, A );
        else
            cell_p_kernel<<<dim3( ( ctx->n_mb + 255 ) / 256, 1 ), 256, 0, ctx->stream>>>( P, nullptr, A );
        cell_reduce_kernel<<<dim3( 1, reduce_bands( P, 1 ) ), 256, 0, ctx->stream>>>( P, nullptr, A );
        HIPCK( hipGetLastError() );
        int r = sync_stream( ctx );
        if( r ) return r;
    }
    out->cost_est = res[0]; out->cost_est_aq = res[1]; out->intra_mbs = res[2];
    out->intra_cost_est = res[3]; out->intra_cost_est_aq = res[4];
    ctx->counters[1]++;
    return X264HIP_OK;
}

extern "C" int x264hip_frame_cost( x264hip_ctx *ctx, int slot_p0, int slot_p1, int slot_b, int dist_p0, int dist_p1, const int do_search[2],
                                   const x264hip_weight *w, int with_intra, int ref1_l0_valid, x264hip_cost *out )
{
    if( !ctx || !out || !do_search || !slot_ok( ctx, slot_p0 ) || !slot_ok( ctx, slot_p1 ) || !slot_ok( ctx, slot_b ) ) return X264HIP_EINVAL;
    if( dist_p0 < 0 || dist_p1 < 0 || dist_p0 + dist_p1 > ctx->p.bframes + 1 || ( dist_p0 == 0 && dist_p1 != 0 ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    if( !ctx->slots[slot_b].in_use || !ctx->slots[slot_p0].in_use || !ctx->slots[slot_p1].in_use ) return X264HIP_ESTATE;
    return ctx->p.bit_depth == 8
               ? frame_cost_t<uint8_t>( ctx, slot_p0, slot_p1, slot_b, dist_p0, dist_p1, do_search, w, with_intra, ref1_l0_valid, out )
               : frame_cost_t<uint16_t>( ctx, slot_p0, slot_p1, slot_b, dist_p0, dist_p1, do_search, w, with_intra, ref1_l0_valid, out );
}

// ---- MB-tree ----------------------------------------------------------------------------------------------
// The ring entry the next launch's step table is written to (its previous launch has consumed it)
static int mbt_ring_acquire( x264hip_ctx *ctx, int *r_out )
{
    const int ring = ctx->mbt_next;
    ctx->mbt_next = ( ring + 1 ) % x264hip_ctx::MBT_RING;
    if( ctx->mbt_pending >= x264hip_ctx::MBT_RING )
        HIPCK( hipEventSynchronize( ctx->mbt_done[ring] ) );
    *r_out = ring;
    return X264HIP_OK;
}

// A launch has been enqueued on ring entry r: the slots its steps name are in use until mbt_done[r] has passed.
static void mbt_mark_launch( x264hip_ctx *ctx, int r )
{
    ctx->mbt_serial++;
    ctx->mbt_ring_serial[r] = ctx->mbt_serial;
    for( auto &f : ctx->slots )
        if( f.mbt_queued ) { f.mbt_queued = false; f.mbt_last_use = ctx->mbt_serial; }
}
// A picture is about to be written into slot s (maps, factors, offsets: what MB-tree steps read and write).  Only a launch whose steps NAME
// the slot can still be reading it -- and only that launch is waited for, on the device.  (Until round 6 every ingest waited for the most
// recent MB-tree launch whatever it worked on: an encoder-paced stream spent the 200 us of every mini-GOP's propagation with the pictures
// of the next one waiting in front of the device, profiles/r06_paced_trace.txt.)  A ring entry that has been taken over by a later launch
// belongs to a launch that is complete (mbt_ring_acquire waits for it).
static int mbt_guard_slot( x264hip_ctx *ctx, const FrameSlot &s )
{
    static const bool coarse = getenv( "X264HIP_MBT_GUARD" ) && !strcmp( getenv( "X264HIP_MBT_GUARD" ), "coarse" ); // A/B runs: the round-5 wait
    if( !ctx->mbt_pending ) return X264HIP_OK;
    if( coarse )
    {
        HIPCK( hipStreamWaitEvent( ctx->stream, ctx->ev_mbt_last, 0 ) );
        return X264HIP_OK;
    }
    if( !s.mbt_last_use ) return X264HIP_OK;
    for( int r = 0; r < x264hip_ctx::MBT_RING; r++ )
        if( ctx->mbt_ring_serial[r] == s.mbt_last_use )
        {
            if( hipEventQuery( ctx->mbt_done[r] ) == hipSuccess ) return X264HIP_OK;
            (void)hipGetLastError(); // (hipErrorNotReady is an answer, not a failure)
            HIPCK( hipStreamWaitEvent( ctx->stream, ctx->mbt_done[r], 0 ) );
            return X264HIP_OK;
        }
    return X264HIP_OK;
}

// Workgroups per list.  The workgroups of a list wait for each other at its barriers, so all of them have to be resident together:
// with 1024 threads and 80 registers a CU holds one, and the lists of a launch of every open context must fit the chip at once
// (otherwise workgroups that spin at a barrier could keep the ones they wait for off the CUs): a context's share of the CUs, divided
// among the lists of the launch, and no more than leaves a thread four macroblocks per step (at least four workgroups).  Measured, 1080p, 16 lists of ~80 steps
// per launch: one context 14 750 frames/s with 4 workgroups per list, 13 100 with 2, 10 200 with 1; eight contexts 21 900 / 21 800 /
// 20 800.  Few lists of large pictures (BASELINE configs[4]: 8K, 12 frames per segment, one to three lists per launch) get up to 16.
static int mbt_wgs_per_list( x264hip_ctx *ctx, int n_lists )
{
    static const int forced = getenv( "X264HIP_MBT_WGS" ) ? std::max( 1, std::min( 64, atoi( getenv( "X264HIP_MBT_WGS" ) ) ) ) : 0;
    if( forced ) return forced;
    const int contexts = std::max( 1, g_open_contexts[ctx->device & 63].load() );
    // never more than a quarter of the chip for the workgroups of one launch that wait for each other (with up to 48 lists per launch
    // since round 5, four workgroups per list would hold 192 CUs at their barriers: a latency-form search launch beside them, whose
    // waves also wait for each other, then cannot become resident -- seen once as an in-kernel timeout)
    const int share = std::max( 1, std::min( ctx->n_cu / contexts, ctx->n_cu / 4 ) );
    const int by_work = std::max( 4, ( ctx->n_mb + 4095 ) / 4096 ); // 1080p 4, 4K 8, 8K 16
    return std::max( 1, std::min( std::min( share / std::max( 1, n_lists ), by_work ), 16 ) );
}

// Launch the queued step lists: one kernel, every list on its own workgroups and accumulator bank.  Everything that reads what the
// lists write (quantiser offsets, accumulators), rewrites what they read (slot reuse, the intra clamp) or must see them finished
// (synchronize) calls this first, so that a caller cannot tell the queue from a launch per call.
static int mbt_flush( x264hip_ctx *ctx )
{
    if( ctx->mbt_q.n == 0 ) return X264HIP_OK;
    const int r = ctx->mbt_q_ring;
    const MbtGroups G = ctx->mbt_q;
    ctx->mbt_q.n = 0; ctx->mbt_q.beg[0] = 0; ctx->mbt_q_ring = -1;
    ctx->mbt_q_finished.clear();
    // Threads per workgroup: a list alone on the device wants its steps short (1024 threads: a step of 1080p is two rounds of loads); beside
    // the searches of several contexts what counts is how little a waiting list holds.  Measured, 1080p slow+dia, 40 timed steps
    // (scripts/r05_mbt_shape2.sh): eight contexts 38 000 frames/s at 1024, 38 400 at 512, 38 800 at 256; one context 27 800 / 26 500 /
    // 23 500.  (No MB-tree at all: 40 700 -- the propagation costs the full device 6 %.)
    static const int forced_threads = getenv( "X264HIP_MBT_THREADS" ) ? std::max( 64, std::min( 1024, atoi( getenv( "X264HIP_MBT_THREADS" ) ) & ~63 ) ) : 0;
    const int open_contexts = std::max( 1, g_open_contexts[ctx->device & 63].load() );
    const int mbt_threads = forced_threads ? forced_threads : open_contexts >= 4 ? 256 : open_contexts >= 2 ? 512 : MBT_THREADS;
    const int mbt_wgs = mbt_wgs_per_list( ctx, G.n );
    // inputs come from the main stream (cells, clamp kernels): order the MB-tree stream behind it
    HIPCK( hipEventRecord( ctx->ev_cross, ctx->stream ) );
    HIPCK( hipStreamWaitEvent( ctx->stream2, ctx->ev_cross, 0 ) );
    // Measured (two segments in flight, 1080p): copying the step list to the device in front of the launch gives 8500 frames/s,
    // letting every workgroup pull it from pinned host memory into LDS 8070
    HIPCK( upload_async( ctx, ctx->mbt_dev[r], ctx->mbt_host[r], (size_t)G.beg[G.n] * sizeof( MbtOpDev ), ctx->stream2 ) );
    static const bool skip_kernel = getenv( "X264HIP_MBT_SKIP" ) != nullptr; // timing experiments only: what the stream costs the others (offsets are then wrong)
    // One launch with counter barriers (default) or one launch per level (X264HIP_MBT=levels).  Measured, 1080p slow+dia, eight contexts
    // (profiles/r05_mbtree_forms.txt): 31 600 - 32 200 frames/s against 22 700 - 23 200 -- a level is a small launch that has to find free
    // wave slots on a chip full of search waves that live for a millisecond, thirty times per flush, where the barrier form takes its
    // CUs once; one context alone: 19 900 either way.  (No MB-tree at all: 36 100 / 27 800 -- the propagation is a tenth of the device work.)
    static const bool spin_form = !( getenv( "X264HIP_MBT" ) && !strcmp( getenv( "X264HIP_MBT" ), "levels" ) );
    const bool lds_form = ctx->mbt_q_lds;
    ctx->mbt_q_lds = false;
    if( skip_kernel )
        ;
    else if( lds_form )
    {
        static bool attr_set = false;
        if( !attr_set )
        {
            HIPCK( hipFuncSetAttribute( (const void *)mbtree_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 ) );
            attr_set = true;
        }
        const int lds_slots = std::min( 6, (int)( ( 156 * 1024 ) / ( (size_t)ctx->n_mb * sizeof( int ) ) ) );
        mbtree_lds_kernel<<<G.n, 1024, (size_t)lds_slots * ctx->n_mb * sizeof( int ), ctx->stream2>>>( ctx->P, ctx->mbt_dev[r], G, ctx->luts_dev );
    }
    else if( spin_form )
        mbtree_kernel<<<G.n * mbt_wgs, mbt_threads, 0, ctx->stream2>>>( ctx->P, ctx->mbt_dev[r], G, mbt_wgs, ctx->luts_dev,
                                                                        ctx->mbt_bar + (size_t)r * MBT_MAX_GROUPS * 4,
                                                                        ctx->mbt_bar + (size_t)x264hip_ctx::MBT_RING * MBT_MAX_GROUPS * 4 );
    else
    {
        // one launch per level (la_kernels.h, mbtree_level_kernel): the steps sorted by the number of barriers in front of them in their list
        const MbtOpDev *dh = ctx->mbt_host[r];
        const int n_ops = G.beg[G.n];
        std::vector<int> level( n_ops );
        int n_levels = 0;
        for( int g = 0; g < G.n; g++ )
        {
            int lv = 0;
            for( int k = G.beg[g]; k < G.beg[g + 1]; k++ )
            {
                lv += dh[k].barrier_before;
                level[k] = lv;
            }
            n_levels = std::max( n_levels, lv + 1 );
        }
        std::vector<int> first( n_levels + 1, 0 );
        for( int k = 0; k < n_ops; k++ ) first[level[k] + 1]++;
        for( int l = 0; l < n_levels; l++ ) first[l + 1] += first[l];
        int *order_host = (int *)( ctx->mbt_host[r] + x264hip_ctx::MBT_CAP );
        {
            std::vector<int> at( first.begin(), first.end() - 1 );
            for( int k = 0; k < n_ops; k++ ) order_host[at[level[k]]++] = k;
        }
        int *order_dev = (int *)( ctx->mbt_dev[r] + x264hip_ctx::MBT_CAP );
        HIPCK( upload_async( ctx, order_dev, order_host, (size_t)n_ops * sizeof( int ), ctx->stream2 ) );
        const unsigned gx = (unsigned)( ( ctx->n_mb + 256 * MBT_UNROLL - 1 ) / ( 256 * MBT_UNROLL ) );
        for( int l = 0; l < n_levels; l++ )
            if( first[l + 1] > first[l] )
                mbtree_level_kernel<<<dim3( gx, (unsigned)( first[l + 1] - first[l] ) ), 256, 0, ctx->stream2>>>( ctx->P, ctx->mbt_dev[r], order_dev + first[l], ctx->luts_dev );
    }
    HIPCK( hipGetLastError() );
    HIPCK( hipEventRecord( ctx->mbt_done[r], ctx->stream2 ) );
    HIPCK( hipEventRecord( ctx->ev_mbt_last, ctx->stream2 ) );
    ctx->mbt_pending++;
    mbt_mark_launch( ctx, r );
    return X264HIP_OK;
}
// (X264HIP_TRACE_CLASSES: which call sites cut the MB-tree queue into launches, printed when the context closes)
static int mbt_flush_at( x264hip_ctx *ctx, int line )
{
    if( ctx->mbt_q.n )
    {
        ctx->flush_sites[line].first++;
        ctx->flush_sites[line].second += ctx->mbt_q.n;
    }
    return mbt_flush( ctx );
}

// the accumulator of (bank, slot)
static int *mbt_bank_acc( x264hip_ctx *ctx, int bank, int slot )
{
    return bank == 0 ? ctx->slots[slot].prop : ctx->prop_bank[bank] + (size_t)slot * ctx->n_mb;
}

extern "C" int x264hip_mbtree( x264hip_ctx *ctx, const x264hip_mbtree_op *ops, int n )
{
    if( !ctx || !ops || n <= 0 || n > x264hip_ctx::MBT_CAP ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int nstride = ctx->p.bframes + 2;
    // validate the whole list first: nothing below (accumulator swaps, the ring entry) may happen for a list that is rejected
    for( int i = 0; i < n; i++ )
    {
        const x264hip_mbtree_op &o = ops[i];
        if( !slot_ok( ctx, o.slot_b ) || !slot_ok( ctx, o.slot_p0 ) || !slot_ok( ctx, o.slot_p1 ) || o.type < 0 || o.type > X264HIP_MBT_RESET_QP )
            return X264HIP_EINVAL;
        if( o.type == X264HIP_MBT_PROPAGATE || o.type == X264HIP_MBT_FINISH )
            if( o.dist_p0 < 0 || o.dist_p1 < 0 || o.dist_p0 + o.dist_p1 > ctx->p.bframes + 1 || ( o.type == X264HIP_MBT_PROPAGATE && o.dist_p0 < 1 ) )
                return X264HIP_EINVAL;
    }
    // window shard: a propagation reads the cell's map and vectors; whatever the caller did not fetch from the owner rank
    // (x264hip_import_cell_map) is evaluated here
    for( int i = 0; i < n; i++ )
        if( ops[i].type == X264HIP_MBT_PROPAGATE )
        {
            int rc = ensure_cell_local( ctx, ops[i].slot_b, ops[i].dist_p0, ops[i].dist_p1 );
            if( rc ) return rc;
        }
    // (mbt_guard_slot: the slots a list names belong to the launch that carries it until that launch's event has passed; marked where the
    //  list joins a launch -- a flush on the way there launches OTHER lists)
    auto name_slots = [&]() {
        for( int i = 0; i < n; i++ )
            ctx->slots[ops[i].slot_b].mbt_queued = ctx->slots[ops[i].slot_p0].mbt_queued = ctx->slots[ops[i].slot_p1].mbt_queued = true;
    };
    // A list that clears every accumulator it reads before it adds to it (every list of a lookahead with frames in it,
    // slicetype.c:1108-1135) depends on the lists before it through nothing but the buffers: it is queued, gets an accumulator bank
    // of its own and runs beside the other lists of the launch.  Lists that carry state across calls (the lookahead-less form swaps
    // accumulators and resets offsets, slicetype.c:1096-1121) run at once on the slots' own accumulators, as before.
    static const bool no_lds = getenv( "X264HIP_MBT_LDS" ) == nullptr; // measured slower than the multi-workgroup kernel: opt-in for experiments
    static const int max_groups = getenv( "X264HIP_MBT_GROUPS" ) ? std::max( 0, std::min( MBT_MAX_GROUPS, atoi( getenv( "X264HIP_MBT_GROUPS" ) ) ) ) : MBT_MAX_GROUPS;
    bool queued_form = no_lds && max_groups > 0;
    {
        std::vector<char> cleared( ctx->slots.size(), 0 );
        for( int i = 0; i < n && queued_form; i++ )
        {
            const x264hip_mbtree_op &o = ops[i];
            if( o.type == X264HIP_MBT_ZERO ) cleared[o.slot_b] = 1;
            else if( o.type == X264HIP_MBT_SWAP || o.type == X264HIP_MBT_RESET_QP ) queued_form = false;
        }
        for( int i = 0; i < n && queued_form; i++ )
        {
            const x264hip_mbtree_op &o = ops[i];
            if( o.type == X264HIP_MBT_FINISH ) queued_form = cleared[o.slot_b];
            else if( o.type == X264HIP_MBT_PROPAGATE )
                queued_form = ( !o.referenced || cleared[o.slot_b] ) && cleared[o.slot_p0] && ( o.dist_p1 == 0 || cleared[o.slot_p1] );
        }
    }
    if( queued_form )
    {
        // Two lists of a launch must not write the offsets of the same frame: the later one has to win.  Nothing can have read the
        // earlier list's offsets in between (every reader launches the queue first), so its FINISH step is dead: it becomes a no-op.
        // (Until round 5 the queue was launched instead -- 120 times per 1 900 frames of the bench clip, 5.5 lists per launch on
        // average where 16 fit; profiles/r05_mbtree_forms.txt.)
        if( ctx->mbt_q_ring >= 0 )
            for( int i = 0; i < n; i++ )
                if( ops[i].type == X264HIP_MBT_FINISH )
                    for( auto &f : ctx->mbt_q_finished )
                        if( f.first == ops[i].slot_b && f.second >= 0 )
                        {
                            ctx->mbt_host[ctx->mbt_q_ring][f.second].type = MBT_NOP;
                            f.second = -1;
                        }
        if( ctx->mbt_q.n >= max_groups || ctx->mbt_q.beg[ctx->mbt_q.n] + n > x264hip_ctx::MBT_CAP )
        {
            int rc = mbt_flush_at( ctx, __LINE__ );
            if( rc ) return rc;
        }
        // the list's accumulator bank: bank 0 is the slots' own accumulators, the others are allocated when first used.  A context that cannot
        // get another bank (max_frames x n_mb x 4 bytes each: 130 MB for a 250-frame window at 8K) is not broken by that: it runs with the
        // banks it has -- fewer lists side by side -- from then on
        if( ctx->mbt_q.n >= ctx->mbt_bank_limit )
        {
            int rc = mbt_flush_at( ctx, __LINE__ );
            if( rc ) return rc;
        }
        int bank = ctx->mbt_q.n;
        if( bank > 0 && !ctx->prop_bank[bank] && hipMalloc( &ctx->prop_bank[bank], ctx->slots.size() * (size_t)ctx->n_mb * sizeof( int ) ) != hipSuccess )
        {
            (void)hipGetLastError();
            ctx->prop_bank[bank] = nullptr;
            ctx->mbt_bank_limit = bank;
            int rc = mbt_flush_at( ctx, __LINE__ );
            if( rc ) return rc;
            bank = 0;
        }
        if( ctx->mbt_q_ring < 0 )
        {
            int rc = mbt_ring_acquire( ctx, &ctx->mbt_q_ring );
            if( rc ) return rc;
        }
        MbtOpDev *dh = ctx->mbt_host[ctx->mbt_q_ring] + ctx->mbt_q.beg[bank];
        // The form of the launch: every list on several workgroups that meet at counter barriers and add into the bank with global atomics
        // (mbtree_kernel, the default), or on ONE workgroup with the accumulators in play in LDS (mbtree_lds_kernel, X264HIP_MBT=lds;
        // pictures whose accumulators fit three at a time: the two anchors of a mini-GOP and its B-reference).  Measured, 1080p slow+dia
        // (scripts/r05_mbt_lds.sh, two runs each): eight contexts 37 200 / 36 100 frames/s in LDS against 36 100 / 38 000, one context
        // 26 900 / 27 000 against 27 700 / 27 800 -- the lists take a tenth of the device whichever way they are walked (18 M macroblock
        // steps per 160-frame pass, each a round trip with a handful of instructions behind it: what they cost is the registers and wave
        // slots they hold while they wait), and one workgroup per list is the longer chain for the one stream that ends on it.
        static const bool want_lds = getenv( "X264HIP_MBT" ) && !strcmp( getenv( "X264HIP_MBT" ), "lds" );
        const int lds_slots = std::min( 6, (int)( ( 156 * 1024 ) / ( (size_t)ctx->n_mb * sizeof( int ) ) ) );
        const bool in_lds = want_lds && lds_slots >= 3 && ( ctx->mbt_q.n == 0 || ctx->mbt_q_lds );
        if( in_lds )
        {
            // steps in the caller's order; accumulators (identified by their buffer in the bank) mapped onto LDS slots, least recently
            // used out.  A cleared accumulator costs nothing until a step adds to it (then its slot is cleared in LDS); one that never
            // enters LDS -- a frame nothing refers to -- has its buffer cleared at the end, for whoever reads it afterwards.
            std::vector<MbtOpDev> L;
            std::vector<int *> held( lds_slots, nullptr ), zero_pending;
            std::vector<int> stamp( lds_slots, 0 );
            std::vector<std::pair<int, int>> fin;
            int clock = 0;
            auto find = [&]( int *acc ) { for( int c = 0; c < lds_slots; c++ ) if( held[c] == acc ) return c; return -1; };
            auto move_op = [&]( int type, int *acc, int slot ) {
                MbtOpDev d;
                memset( &d, 0, sizeof( d ) );
                d.type = type; d.prop_b = acc; d.lds_b = slot;
                L.push_back( d );
            };
            auto hold = [&]( int *acc, int pin0, int pin1 ) {
                int c = find( acc );
                if( c < 0 )
                {
                    for( int q = 0; q < lds_slots && c < 0; q++ )
                        if( !held[q] && q != pin0 && q != pin1 ) c = q;
                    for( int q = 0; q < lds_slots && !( c >= 0 && !held[c] ); q++ )
                        if( held[q] && q != pin0 && q != pin1 && ( c < 0 || stamp[q] < stamp[c] ) ) c = q;
                    if( held[c] ) move_op( MBT_LDS_STORE, held[c], c );
                    held[c] = acc;
                    auto z = std::find( zero_pending.begin(), zero_pending.end(), acc );
                    if( z != zero_pending.end() )
                    {
                        zero_pending.erase( z );
                        move_op( X264HIP_MBT_ZERO, acc, c );
                    }
                    else
                        move_op( MBT_LDS_LOAD, acc, c );
                }
                stamp[c] = ++clock;
                return c;
            };
            for( int i = 0; i < n; i++ )
            {
                const x264hip_mbtree_op &o = ops[i];
                FrameSlot &b = ctx->slots[o.slot_b];
                int *acc_b = mbt_bank_acc( ctx, bank, o.slot_b );
                if( o.type == X264HIP_MBT_ZERO )
                {
                    b.prop_view = acc_b; // the frame's accumulator is what this list leaves in its bank
                    const int c = find( acc_b );
                    if( c >= 0 ) move_op( X264HIP_MBT_ZERO, acc_b, c );
                    else if( std::find( zero_pending.begin(), zero_pending.end(), acc_b ) == zero_pending.end() ) zero_pending.push_back( acc_b );
                    continue;
                }
                MbtOpDev d;
                memset( &d, 0, sizeof( d ) );
                d.type = o.type; d.referenced = o.referenced; d.bipred_weight = o.bipred_weight; d.fps_factor_i = o.fps_factor_i;
                d.fps_factor = o.fps_factor; d.weightdelta = o.weightdelta; d.strength = o.strength;
                d.b_bidir = o.dist_p1 > 0;
                d.prop_b = acc_b; d.prop_p0 = mbt_bank_acc( ctx, bank, o.slot_p0 ); d.prop_p1 = mbt_bank_acc( ctx, bank, o.slot_p1 );
                d.intra_cost = b.lowres_costs; d.inv_qscale = b.inv_qscale;
                d.qp_aq = b.qp_aq; d.qp = b.qp;
                d.lowres_costs = b.lowres_costs + (size_t)b.cell_at[o.dist_p0 * nstride + o.dist_p1] * ctx->n_mb;
                if( o.type == X264HIP_MBT_FINISH )
                {
                    d.lds_b = hold( acc_b, -1, -1 );
                    fin.push_back( std::make_pair( o.slot_b, (int)L.size() ) );
                }
                else
                {
                    d.mvq0 = b.mvq[0][o.dist_p0 - 1];
                    d.mvq1 = o.dist_p1 > 0 ? b.mvq[1][o.dist_p1 - 1] : nullptr;
                    int kb = -1;
                    if( o.referenced ) kb = hold( acc_b, -1, -1 ); // its own total is read; an unreferenced B-frame has none
                    const int k0 = hold( d.prop_p0, kb, -1 );
                    const int k1 = d.b_bidir ? hold( d.prop_p1, kb, k0 ) : k0;
                    d.lds_b = kb < 0 ? 0 : kb; d.lds_p0 = k0; d.lds_p1 = k1;
                }
                L.push_back( d );
            }
            for( int c = 0; c < lds_slots; c++ )
                if( held[c] ) move_op( MBT_LDS_STORE, held[c], c );
            for( int *acc : zero_pending )
                move_op( X264HIP_MBT_ZERO, acc, -1 );
            if( ctx->mbt_q.beg[bank] + (int)L.size() <= x264hip_ctx::MBT_CAP )
            {
                memcpy( dh, L.data(), L.size() * sizeof( MbtOpDev ) );
                for( const auto &f : fin ) // ( frame slot, index of its FINISH step in the launch's table )
                    ctx->mbt_q_finished.push_back( std::make_pair( f.first, ctx->mbt_q.beg[bank] + f.second ) );
                ctx->mbt_q_lds = true;
                ctx->mbt_q.beg[bank + 1] = ctx->mbt_q.beg[bank] + (int)L.size();
                ctx->mbt_q.n = bank + 1;
                name_slots();
                return X264HIP_OK;
            }
            if( bank != 0 )
            {
                // the list with its moves does not fit behind the ones queued: they go first, this call starts the next launch
                int rc = mbt_flush_at( ctx, __LINE__ );
                if( rc ) return rc;
                return x264hip_mbtree( ctx, ops, n );
            }
            // (a single list too long for the table with its moves: the multi-workgroup form below takes it as it is)
        }
        ctx->mbt_q_lds = false;
        // Step order on the device: every ZERO first (a buffer is always cleared before anything is added to it in the
        // reference's order too), then the rest in order.  A barrier is only needed where a step reads what earlier
        // steps accumulated: referenced PROPAGATEs and FINISH; runs of B-frame propagations overlap freely.
        int k = 0;
        for( int pass = 0; pass < 2; pass++ )
            for( int i = 0; i < n; i++ )
            {
                const x264hip_mbtree_op &o = ops[i];
                if( ( o.type == X264HIP_MBT_ZERO ) != ( pass == 0 ) ) continue;
                FrameSlot &b = ctx->slots[o.slot_b];
                MbtOpDev d;
                memset( &d, 0, sizeof( d ) );
                d.type = o.type; d.referenced = o.referenced; d.bipred_weight = o.bipred_weight; d.fps_factor_i = o.fps_factor_i;
                d.fps_factor = o.fps_factor; d.weightdelta = o.weightdelta; d.strength = o.strength;
                d.b_bidir = o.dist_p1 > 0;
                d.prop_b = mbt_bank_acc( ctx, bank, o.slot_b ); d.prop_p0 = mbt_bank_acc( ctx, bank, o.slot_p0 ); d.prop_p1 = mbt_bank_acc( ctx, bank, o.slot_p1 );
                d.intra_cost = b.lowres_costs; d.inv_qscale = b.inv_qscale;
                d.qp_aq = b.qp_aq; d.qp = b.qp;
                d.lowres_costs = b.lowres_costs;
                if( o.type == X264HIP_MBT_ZERO )
                    b.prop_view = d.prop_b; // the frame's accumulator is what this list leaves in its bank
                else
                {
                    d.barrier_before = k == 0 || dh[k - 1].type == X264HIP_MBT_ZERO || ( o.type == X264HIP_MBT_PROPAGATE && o.referenced ) || o.type == X264HIP_MBT_FINISH;
                    d.lowres_costs = b.lowres_costs + (size_t)b.cell_at[o.dist_p0 * nstride + o.dist_p1] * ctx->n_mb;
                    if( o.type == X264HIP_MBT_PROPAGATE )
                    {
                        d.mvq0 = b.mvq[0][o.dist_p0 - 1];
                        d.mvq1 = o.dist_p1 > 0 ? b.mvq[1][o.dist_p1 - 1] : nullptr;
                    }
                    else
                        ctx->mbt_q_finished.push_back( std::make_pair( o.slot_b, ctx->mbt_q.beg[bank] + k ) );
                }
                dh[k++] = d;
            }
        ctx->mbt_q.beg[bank + 1] = ctx->mbt_q.beg[bank] + k;
        ctx->mbt_q.n = bank + 1;
        name_slots();
        return X264HIP_OK;
    }
    // the immediate form: behind everything queued, on the slots' own accumulators -- contents that live in a bank move home first
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
        name_slots();
        for( int i = 0; i < n; i++ )
            for( int slot : { ops[i].slot_b, ops[i].slot_p0, ops[i].slot_p1 } )
            {
                FrameSlot &f = ctx->slots[slot];
                if( f.prop_view != f.prop )
                {
                    HIPCK( hipMemcpyAsync( f.prop, f.prop_view, ctx->n_mb * sizeof( int ), hipMemcpyDeviceToDevice, ctx->stream2 ) );
                    f.prop_view = f.prop;
                }
            }
    }
    int r = 0;
    {
        int rc = mbt_ring_acquire( ctx, &r );
        if( rc ) return rc;
    }
    MbtOpDev *dh = ctx->mbt_host[r];
    // The lookahead-less form exchanges the accumulators of two frames (X264HIP_MBT_SWAP): that is a swap of the two slots' buffer
    // pointers, applied here in the caller's order, so that every step below addresses the buffer the reference would.  RESET_QP
    // is a device-to-device copy queued behind the kernel (it only ever comes with a ZERO of the same frame, slicetype.c:1117-1121).
    std::vector<int *> res_b( n ), res_p0( n ), res_p1( n );
    std::vector<int> resets;
    for( int i = 0; i < n; i++ )
    {
        const x264hip_mbtree_op &o = ops[i];
        if( o.type == X264HIP_MBT_SWAP )
        {
            std::swap( ctx->slots[o.slot_b].prop, ctx->slots[o.slot_p0].prop );
            ctx->slots[o.slot_b].prop_view = ctx->slots[o.slot_b].prop; ctx->slots[o.slot_p0].prop_view = ctx->slots[o.slot_p0].prop;
        }
        else if( o.type == X264HIP_MBT_RESET_QP )
            resets.push_back( o.slot_b );
        res_b[i] = ctx->slots[o.slot_b].prop; res_p0[i] = ctx->slots[o.slot_p0].prop; res_p1[i] = ctx->slots[o.slot_p1].prop;
    }
    // step order and barriers as in the queued form
    std::vector<int> order;
    for( int i = 0; i < n; i++ ) if( ops[i].type == X264HIP_MBT_ZERO ) order.push_back( i );
    const int n_zero = (int)order.size();
    for( int i = 0; i < n; i++ ) if( ops[i].type == X264HIP_MBT_PROPAGATE || ops[i].type == X264HIP_MBT_FINISH ) order.push_back( i );
    const int n_host = n;
    n = (int)order.size(); // steps that run on the device
    for( int k = 0; k < n; k++ )
    {
        const int i = k;
        const x264hip_mbtree_op &o = ops[order[k]];
        FrameSlot &b = ctx->slots[o.slot_b];
        MbtOpDev d;
        memset( &d, 0, sizeof( d ) );
        d.type = o.type; d.referenced = o.referenced; d.bipred_weight = o.bipred_weight; d.fps_factor_i = o.fps_factor_i;
        d.fps_factor = o.fps_factor; d.weightdelta = o.weightdelta; d.strength = o.strength;
        d.b_bidir = o.dist_p1 > 0;
        d.barrier_before = k == n_zero || ( o.type == X264HIP_MBT_PROPAGATE && o.referenced ) || o.type == X264HIP_MBT_FINISH;
        d.prop_b = res_b[order[k]]; d.prop_p0 = res_p0[order[k]]; d.prop_p1 = res_p1[order[k]];
        d.intra_cost = b.lowres_costs; d.inv_qscale = b.inv_qscale;
        d.lowres_costs = b.lowres_costs + (size_t)b.cell_at[o.dist_p0 * nstride + o.dist_p1] * ctx->n_mb;
        if( o.type == X264HIP_MBT_PROPAGATE )
        {
            d.mvq0 = b.mvq[0][o.dist_p0 - 1];
            d.mvq1 = o.dist_p1 > 0 ? b.mvq[1][o.dist_p1 - 1] : nullptr;
        }
        d.qp_aq = b.qp_aq; d.qp = b.qp;
        dh[i] = d;
    }
    // inputs come from the main stream (cells, clamp kernels): order the MB-tree stream behind it
    HIPCK( hipEventRecord( ctx->ev_cross, ctx->stream ) );
    HIPCK( hipStreamWaitEvent( ctx->stream2, ctx->ev_cross, 0 ) );
    // One workgroup with the accumulators in LDS when at least three of them fit (three are in play in a mini-GOP with a
    // B-reference: the two anchors and the middle frame); larger pictures use the multi-workgroup kernel below
    const int lds_slots = std::min( 6, (int)( ( 156 * 1024 ) / ( (size_t)ctx->n_mb * sizeof( int ) ) ) );
    bool done_in_lds = false;
    if( n > 0 && !no_lds && lds_slots >= 3 )
    {
        // steps in the caller's order; accumulators (identified by their global buffer) mapped onto LDS slots, least recently used out
        std::vector<MbtOpDev> L;
        std::vector<int *> held( lds_slots, nullptr );
        std::vector<int> stamp( lds_slots, 0 );
        int clock = 0;
        auto find = [&]( int *acc ) { for( int k = 0; k < lds_slots; k++ ) if( held[k] == acc ) return k; return -1; };
        auto move_op = [&]( int type, int *acc, int slot ) {
            MbtOpDev d;
            memset( &d, 0, sizeof( d ) );
            d.type = type; d.prop_b = acc; d.lds_b = slot;
            L.push_back( d );
        };
        // make acc resident (pinned = slots that must stay); load = its current global contents matter
        auto hold = [&]( int *acc, bool load, int pin0, int pin1 ) {
            int k = find( acc );
            if( k < 0 )
            {
                // a free slot if there is one, else the slot used longest ago; never one this step still needs
                for( int c = 0; c < lds_slots && k < 0; c++ )
                    if( !held[c] && c != pin0 && c != pin1 ) k = c;
                for( int c = 0; c < lds_slots && !( k >= 0 && !held[k] ); c++ )
                    if( held[c] && c != pin0 && c != pin1 && ( k < 0 || stamp[c] < stamp[k] ) ) k = c;
                if( held[k] ) move_op( MBT_LDS_STORE, held[k], k );
                held[k] = acc;
                if( load ) move_op( MBT_LDS_LOAD, acc, k );
            }
            stamp[k] = ++clock;
            return k;
        };
        for( int q = 0; q < n_host; q++ )
        {
            const x264hip_mbtree_op &o = ops[q];
            if( o.type != X264HIP_MBT_ZERO && o.type != X264HIP_MBT_PROPAGATE && o.type != X264HIP_MBT_FINISH ) continue;
            FrameSlot &b = ctx->slots[o.slot_b];
            MbtOpDev d;
            memset( &d, 0, sizeof( d ) );
            d.type = o.type; d.referenced = o.referenced; d.bipred_weight = o.bipred_weight; d.fps_factor_i = o.fps_factor_i;
            d.fps_factor = o.fps_factor; d.weightdelta = o.weightdelta; d.strength = o.strength;
            d.b_bidir = o.dist_p1 > 0;
            d.prop_b = res_b[q]; d.prop_p0 = res_p0[q]; d.prop_p1 = res_p1[q];
            d.intra_cost = b.lowres_costs; d.inv_qscale = b.inv_qscale;
            d.lowres_costs = b.lowres_costs + (size_t)b.cell_at[o.dist_p0 * nstride + o.dist_p1] * ctx->n_mb;
            d.qp_aq = b.qp_aq; d.qp = b.qp;
            if( o.type == X264HIP_MBT_ZERO )
                d.lds_b = hold( res_b[q], false, -1, -1 );
            else if( o.type == X264HIP_MBT_FINISH )
                d.lds_b = hold( res_b[q], true, -1, -1 );
            else
            {
                d.mvq0 = b.mvq[0][o.dist_p0 - 1];
                d.mvq1 = o.dist_p1 > 0 ? b.mvq[1][o.dist_p1 - 1] : nullptr;
                int kb = -1;
                if( o.referenced ) kb = hold( res_b[q], true, -1, -1 ); // its own total is read; an unreferenced B-frame has none
                const int k0 = hold( res_p0[q], true, kb, -1 );
                const int k1 = d.b_bidir ? hold( res_p1[q], true, kb, k0 ) : k0;
                d.lds_b = kb < 0 ? 0 : kb; d.lds_p0 = k0; d.lds_p1 = k1;
            }
            L.push_back( d );
        }
        for( int k = 0; k < lds_slots; k++ )
            if( held[k] ) move_op( MBT_LDS_STORE, held[k], k );
        if( (int)L.size() <= x264hip_ctx::MBT_CAP )
        {
            memcpy( dh, L.data(), L.size() * sizeof( MbtOpDev ) );
            static bool attr_set = false;
            if( !attr_set )
            {
                HIPCK( hipFuncSetAttribute( (const void *)mbtree_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 ) );
                attr_set = true;
            }
            HIPCK( upload_async( ctx, ctx->mbt_dev[r], dh, L.size() * sizeof( MbtOpDev ), ctx->stream2 ) );
            MbtGroups G1;
            memset( &G1, 0, sizeof( G1 ) );
            G1.n = 1; G1.beg[1] = (int)L.size();
            mbtree_lds_kernel<<<1, 1024, (size_t)lds_slots * ctx->n_mb * sizeof( int ), ctx->stream2>>>( ctx->P, ctx->mbt_dev[r], G1, ctx->luts_dev );
            HIPCK( hipGetLastError() );
            done_in_lds = true;
        }
    }
    if( n > 0 && !done_in_lds )
    {
    static const int mbt_threads = getenv( "X264HIP_MBT_THREADS" ) ? std::max( 64, std::min( 1024, atoi( getenv( "X264HIP_MBT_THREADS" ) ) & ~63 ) ) : MBT_THREADS;
    const int mbt_wgs = mbt_wgs_per_list( ctx, 1 );
    MbtGroups G;
    memset( &G, 0, sizeof( G ) );
    G.n = 1; G.beg[1] = n;
    HIPCK( upload_async( ctx, ctx->mbt_dev[r], dh, (size_t)n * sizeof( MbtOpDev ), ctx->stream2 ) );
    mbtree_kernel<<<mbt_wgs, mbt_threads, 0, ctx->stream2>>>( ctx->P, ctx->mbt_dev[r], G, mbt_wgs, ctx->luts_dev, ctx->mbt_bar + (size_t)r * MBT_MAX_GROUPS * 4,
                                                            ctx->mbt_bar + (size_t)x264hip_ctx::MBT_RING * MBT_MAX_GROUPS * 4 );
    HIPCK( hipGetLastError() );
    }
    for( int slot : resets )
        HIPCK( hipMemcpyAsync( ctx->slots[slot].qp, ctx->slots[slot].qp_aq, ctx->n_mb * sizeof( float ), hipMemcpyDeviceToDevice, ctx->stream2 ) );
    HIPCK( hipEventRecord( ctx->mbt_done[r], ctx->stream2 ) );
    HIPCK( hipEventRecord( ctx->ev_mbt_last, ctx->stream2 ) );
    ctx->mbt_pending++;
    mbt_mark_launch( ctx, r );
    return X264HIP_OK;
}

extern "C" int x264hip_get_qp_offsets( x264hip_ctx *ctx, int slot, float *qp_offset )
{
    if( !ctx || !slot_ok( ctx, slot ) || !qp_offset ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
    }
    HIPCK( hipEventSynchronize( ctx->ev_ingest ) ); // f_qp_offset starts as the AQ offsets written at ingest (main stream)
    HIPCK( hipMemcpyAsync( qp_offset, ctx->slots[slot].qp, ctx->n_mb * sizeof( float ), hipMemcpyDeviceToHost, ctx->stream2 ) );
    unsigned err = 0;
    HIPCK( hipMemcpyAsync( &err, ctx->mbt_bar + (size_t)x264hip_ctx::MBT_RING * MBT_MAX_GROUPS * 4, sizeof( unsigned ), hipMemcpyDeviceToHost, ctx->stream2 ) );
    HIPCK( hipStreamSynchronize( ctx->stream2 ) );
    ctx->mbt_pending = 0;
    if( err ) // a barrier of the step walk gave up
    {
        ctx->broken = 1;
        return X264HIP_ETIMEOUT;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_frame_cost_recalculate( x264hip_ctx *ctx, int slot_b, int dist_p0, int dist_p1, int use_aq_offsets, int *score )
{
    if( !ctx || !slot_ok( ctx, slot_b ) || !score || dist_p0 < 0 || dist_p1 < 0 || dist_p0 + dist_p1 > ctx->p.bframes + 1 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &b = ctx->slots[slot_b];
    if( !b.in_use ) return X264HIP_ESTATE;
    const int idx = dist_p0 * ( ctx->p.bframes + 2 ) + dist_p1;
    {
        int rc0 = ensure_cell_local( ctx, slot_b, dist_p0, dist_p1 ); // (window shard: the map may still be with the owner rank)
        if( rc0 ) return rc0;
    }
    // f_qp_offset is written by the MB-tree stream: order this stream behind it
    {
        int rc1 = mbt_flush_at( ctx, __LINE__ );
        if( rc1 ) return rc1;
    }
    if( ctx->mbt_pending )
        HIPCK( hipStreamWaitEvent( ctx->stream, ctx->ev_mbt_last, 0 ) );
    int *res = ctx->cell_acc_host + ( (size_t)slot_b * ctx->n_cells + idx ) * 8 + 7; // a spare word of the cell's pinned result record
    recalc_kernel<<<1, 256, 0, ctx->stream>>>( ctx->P, b.lowres_costs + (size_t)b.cell_at[idx] * ctx->n_mb, use_aq_offsets ? b.qp_aq : b.qp, ctx->luts_dev,
                                               b.row_satds + (size_t)b.cell_at[idx] * ctx->P.mb_h, res );
    HIPCK( hipGetLastError() );
    int rc = sync_stream( ctx );
    if( rc ) return rc;
    *score = *(volatile int *)res;
    return X264HIP_OK;
}

extern "C" int x264hip_frame_add_quant_offsets( x264hip_ctx *ctx, int slot, const float *quant_offsets )
{
    if( !ctx || !slot_ok( ctx, slot ) || !quant_offsets ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    if( !ctx->p.aq_mode ) return X264HIP_OK; // the reference has no offset maps without AQ (frame.c:217-226)
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
    }
    FrameSlot &s = ctx->slots[slot];
    const int n = ctx->n_mb;
    std::vector<float> qp( n );
    std::vector<uint16_t> inv( n );
    HIPCK( hipMemcpyAsync( qp.data(), s.qp_aq, n * sizeof( float ), hipMemcpyDeviceToHost, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    uint8_t exp2_lut[64]; // the table behind x264_exp2fix8 (common/base.h:217-223): round( ( 2^(i/64) - 1 ) * 256 )
    for( int i = 0; i < 64; i++ )
        exp2_lut[i] = (uint8_t)floor( ( pow( 2.0, i / 64.0 ) - 1.0 ) * 256.0 + 0.5 );
    for( int i = 0; i < n; i++ )
    {
        qp[i] += quant_offsets[i];                                         // qp_adj += quant_offsets[mb_xy]
        const int k = (int)( qp[i] * ( -64.f / 6.f ) + 512.5f );           // x264_exp2fix8, common/base.h:217-223
        inv[i] = (uint16_t)( k < 0 ? 0 : k > 1023 ? 0xffff : ( ( exp2_lut[k & 63] + 256 ) << ( k >> 6 ) >> 8 ) );
    }
    HIPCK( hipMemcpyAsync( s.qp_aq, qp.data(), n * sizeof( float ), hipMemcpyHostToDevice, ctx->stream ) );
    HIPCK( hipMemcpyAsync( s.qp, qp.data(), n * sizeof( float ), hipMemcpyHostToDevice, ctx->stream ) );
    HIPCK( hipMemcpyAsync( s.inv_qscale, inv.data(), n * sizeof( uint16_t ), hipMemcpyHostToDevice, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) ); // the host vectors go out of scope
    return X264HIP_OK;
}

extern "C" int x264hip_get_propagate_cost( x264hip_ctx *ctx, int slot, uint16_t *propagate )
{
    if( !ctx || !slot_ok( ctx, slot ) || !propagate ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    {
        int rc = mbt_flush_at( ctx, __LINE__ );
        if( rc ) return rc;
    }
    std::vector<int> tmp( ctx->n_mb );
    HIPCK( hipMemcpyAsync( tmp.data(), ctx->slots[slot].prop_view, ctx->n_mb * sizeof( int ), hipMemcpyDeviceToHost, ctx->stream2 ) );
    HIPCK( hipStreamSynchronize( ctx->stream2 ) );
    for( int i = 0; i < ctx->n_mb; i++ )
        propagate[i] = (uint16_t)( tmp[i] < 32767 ? tmp[i] : 32767 );
    return X264HIP_OK;
}

static WeightJob make_wjob( x264hip_ctx *ctx, int entry, FrameSlot &f, FrameSlot &r, const WtD &wt )
{
    WeightJob j;
    memset( &j, 0, sizeof( j ) );
    if( ctx->p.bit_depth == 8 ) { j.fenc0 = plane_origin<uint8_t>( ctx, f, 0 ); j.ref0 = plane_origin<uint8_t>( ctx, r, 0 ); }
    else { j.fenc0 = plane_origin<uint16_t>( ctx, f, 0 ); j.ref0 = plane_origin<uint16_t>( ctx, r, 0 ); }
    j.intra_cost = f.lowres_costs;
    j.w = wt;
    j.accum = ctx->wcost_dev + 4 * entry;
    j.out_host = ctx->wcost_host + 2 * entry;
    return j;
}

static bool same_weight( const x264hip_weight &a, const x264hip_weight &b )
{
    return a.on == b.on && a.scale == b.scale && a.denom == b.denom && a.offset == b.offset;
}

extern "C" int x264hip_prefetch_weight_costs( x264hip_ctx *ctx, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w )
{
    if( !ctx || n < 0 || ( n && ( !slot_fenc || !slot_ref || !w ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    if( !n ) return X264HIP_OK;
    if( n > x264hip_ctx::WCAP - 1 ) n = x264hip_ctx::WCAP - 1; // a cache: dropping work is always allowed
    int ri = 0;
    if( ring_acquire( ctx->wjob_ring, &ri ) ) return X264HIP_EDEVICE;
    WeightJob *jh = (WeightJob *)ctx->wjob_ring.host[ri], *jd = (WeightJob *)ctx->wjob_ring.dev[ri];
    int m = 0;
    for( int i = 0; i < n; i++ )
    {
        if( !slot_ok( ctx, slot_fenc[i] ) || !slot_ok( ctx, slot_ref[i] ) ) return X264HIP_EINVAL;
        FrameSlot &f = ctx->slots[slot_fenc[i]], &r = ctx->slots[slot_ref[i]];
        if( !f.in_use || !r.in_use || !w[i].on ) continue;
        const int e = ctx->wcache_next;
        ctx->wcache_next = e + 1 < x264hip_ctx::WCAP ? e + 1 : 1;
        x264hip_ctx::WEntry &we = ctx->wcache[e];
        // an entry about to be reused may still be running only if more than WCAP pairs are in flight: wait then
        if( we.slot_fenc >= 0 )
        {
            int rc = batch_wait( ctx, we.batch );
            if( rc ) return rc;
        }
        we.slot_fenc = slot_fenc[i]; we.slot_ref = slot_ref[i]; we.gen_fenc = f.gen; we.gen_ref = r.gen; we.w = w[i];
        we.batch = ctx->batch_serial + 1;
        jh[m++] = make_wjob( ctx, e, f, r, make_wt( ctx, &w[i] ) );
    }
    if( !m ) return X264HIP_OK;
    HIPCK( upload_async( ctx, jd, jh, (size_t)m * sizeof( WeightJob ), ctx->stream ) );
    const dim3 grid( ( ctx->n_mb + WCOST_BLOCKS_PER_WG - 1 ) / WCOST_BLOCKS_PER_WG, m, 1 );
    WeightJob none;
    memset( &none, 0, sizeof( none ) );
    if( ctx->p.bit_depth == 8 )
        weight_cost_kernel<uint8_t><<<grid, 256, 0, ctx->stream>>>( ctx->P, jd, none, 2 );
    else
        weight_cost_kernel<uint16_t><<<grid, 256, 0, ctx->stream>>>( ctx->P, jd, none, 2 );
    HIPCK( hipGetLastError() );
    if( ring_commit( ctx->wjob_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
    return batch_close( ctx );
}

// The list-0 searches a P request would make WITH a weight, ahead of the request: pair i = frame slot_fenc[i] searched on slot_ref[i]
// weighted by w[i] (the weight x264_weights_analyse is going to arrive at -- the caller has the frame totals and the two cost sums it
// needs, x264hip_prefetch_weight_costs).  Each goes into the frame's second list-0 field of that distance, and the P cell over it into
// the spare of cell ( distance, 0 ); an x264hip_frame_cost call that first-triggers the field with the same weight takes both over
// (no launch, no wait beyond this batch).  A request with another weight, or without one, ignores them.  Never changes results.
extern "C" int x264hip_prefetch_weighted_fields( x264hip_ctx *ctx, int n, const int *slot_fenc, const int *slot_ref, const x264hip_weight *w )
{
    if( !ctx || n < 0 || ( n && ( !slot_fenc || !slot_ref || !w ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    std::vector<SearchReq> reqs;
    std::vector<SpecCell> cells;
    const int ns = ctx->p.bframes + 2;
    for( int i = 0; i < n; i++ )
    {
        if( !slot_ok( ctx, slot_fenc[i] ) || !slot_ok( ctx, slot_ref[i] ) ) return X264HIP_EINVAL;
        FrameSlot &b = ctx->slots[slot_fenc[i]], &r = ctx->slots[slot_ref[i]];
        if( !b.in_use || !r.in_use || !w[i].on ) continue;
        const int d = b.frame_no - r.frame_no; // (frame numbers: x264hip_prefetch)
        if( b.frame_no < 0 || r.frame_no < 0 || d < 1 || d > ctx->p.bframes + 1 ) continue;
        FrameSlot::WSpec &ws = b.wspec[d - 1];
        if( b.field_ready[0][d - 1] || ws.valid || (int)reqs.size() >= ctx->desc_cap ) continue; // requested already / searched already
        ws.valid = 1; ws.w = w[i]; ws.batch = ctx->batch_serial + 1;
        SearchReq q{ slot_fenc[i], slot_ref[i], 0, d - 1, make_wt( ctx, &w[i] ) };
        q.to_spare = 1;
        reqs.push_back( q );
    }
    if( reqs.empty() ) return X264HIP_OK;
    int rc = launch_searches( ctx, reqs );
    if( rc ) return rc;
    ctx->counters[0] -= reqs.size(); // (counted for themselves, not among the unweighted speculative searches)
    ctx->weighted_speculated += reqs.size();
    for( const SearchReq &q : reqs )
    {
        FrameSlot &b = ctx->slots[q.slot_b];
        const int idx = ( q.dist_m1 + 1 ) * ns;
        if( b.cells[idx].requested ) continue;
        CellEntry &a = b.alts[idx];
        a = CellEntry();
        a.valid = 1; a.batch = ctx->batch_serial + 1; a.variant = 1;
        a.tag0 = b.wspec[q.dist_m1].tag; a.tag1 = 0; a.tagr = 0;
        SpecCell sc{ q.slot_ref, q.slot_b, q.slot_b, q.dist_m1 + 1, 0, 0, 0 };
        sc.to_spare = 1; sc.wfield = 1;
        cells.push_back( sc );
    }
    // ... and the B cells that read one of the new fields -- as the frame's own list-0 field (it was asked for as a P frame first, over
    // that distance) or as the list-1 reference's vectors (slicetype.c:629; the reference, a P frame, keeps a weight) -- evaluated over
    // them, into the cell's second spare.  Where both exist the cell is evaluated over both: under a fade every P request keeps a weight.
    {
        std::map<int, int> by_number;
        for( int i = 0; i < (int)ctx->slots.size(); i++ )
            if( ctx->slots[i].in_use && ctx->slots[i].frame_no >= 0 ) by_number[ctx->slots[i].frame_no] = i;
        auto has_field = [&]( FrameSlot &f, int list, int dm1 ) { return ( f.field_ready[list][dm1] || f.field_prefetched[list][dm1] ) && !f.field_remote[list][dm1]; };
        const unsigned fresh = ctx->batch_serial + 1;
        for( auto &kv : by_number )
        {
            FrameSlot &b = ctx->slots[kv.second];
            for( int d0 = 1; d0 <= ctx->p.bframes + 1; d0++ )
                for( int d1 = 1; d0 + d1 <= ctx->p.bframes + 1; d1++ )
                {
                    const int idx = d0 * ns + d1;
                    auto i0 = by_number.find( kv.first - d0 ), i1 = by_number.find( kv.first + d1 );
                    if( i0 == by_number.end() || i1 == by_number.end() || !ctx->cell_allowed[idx] ) continue;
                    FrameSlot &f1 = ctx->slots[i1->second];
                    const bool wb = b.wspec[d0 - 1].valid, wr = f1.wspec[d0 + d1 - 1].valid;
                    if( !( ( wb && b.wspec[d0 - 1].batch == fresh ) || ( wr && f1.wspec[d0 + d1 - 1].batch == fresh ) ) ) continue;
                    if( b.cells[idx].requested || b.alts2[idx].valid || !has_field( b, 1, d1 - 1 ) ) continue;
                    if( ( !wb && !has_field( b, 0, d0 - 1 ) ) || ( !wr && !has_field( f1, 0, d0 + d1 - 1 ) ) ) continue;
                    CellEntry &a = b.alts2[idx];
                    a = CellEntry();
                    a.valid = 1; a.batch = fresh; a.variant = 1;
                    a.tag0 = wb ? b.wspec[d0 - 1].tag : b.field_tag[0][d0 - 1];
                    a.tag1 = b.field_tag[1][d1 - 1];
                    a.tagr = wr ? f1.wspec[d0 + d1 - 1].tag : f1.field_tag[0][d0 + d1 - 1];
                    SpecCell sc{ i0->second, i1->second, kv.second, d0, d1, 0, 1 };
                    sc.to_spare = 2; sc.wfield = ( wb ? 1 : 0 ) | ( wr ? 2 : 0 );
                    cells.push_back( sc );
                }
        }
    }
    if( !cells.empty() )
    {
        rc = ctx->p.bit_depth == 8 ? launch_cells_t<uint8_t>( ctx, cells ) : launch_cells_t<uint16_t>( ctx, cells );
        if( rc ) return rc;
        ctx->weighted_cells += cells.size();
    }
    return batch_close( ctx );
}

extern "C" int x264hip_weighted_stats( x264hip_ctx *ctx, uint64_t out[4] )
{
    if( !ctx || !out ) return X264HIP_EINVAL;
    out[0] = ctx->weighted_speculated; out[1] = ctx->weighted_claimed; out[2] = ctx->weighted_cells; out[3] = ctx->weighted_cells_used;
    return X264HIP_OK;
}

extern "C" int x264hip_weight_cost( x264hip_ctx *ctx, int slot_fenc, int slot_ref, const x264hip_weight *w, unsigned *cost )
{
    if( !ctx || !cost || !slot_ok( ctx, slot_fenc ) || !slot_ok( ctx, slot_ref ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &f = ctx->slots[slot_fenc], &r = ctx->slots[slot_ref];
    if( !f.in_use || !r.in_use ) return X264HIP_ESTATE;
    const bool weighted = w && w->on;
    // speculative result?  (unweighted sum: any pair of these two frames; weighted: the same weight)
    for( int e = 1; e < x264hip_ctx::WCAP; e++ )
    {
        const x264hip_ctx::WEntry &we = ctx->wcache[e];
        if( we.slot_fenc != slot_fenc || we.slot_ref != slot_ref || we.gen_fenc != f.gen || we.gen_ref != r.gen ) continue;
        if( weighted && !same_weight( we.w, *w ) ) continue;
        int rc = batch_wait( ctx, we.batch );
        if( rc ) return rc;
        *cost = ( (volatile unsigned *)ctx->wcost_host )[2 * e + ( weighted ? 1 : 0 )];
        ctx->counters[6]++;
        return X264HIP_OK;
    }
    const WeightJob j = make_wjob( ctx, 0, f, r, make_wt( ctx, w ) );
    const dim3 grid( ( ctx->n_mb + WCOST_BLOCKS_PER_WG - 1 ) / WCOST_BLOCKS_PER_WG, 1, 1 );
    if( ctx->p.bit_depth == 8 )
        weight_cost_kernel<uint8_t><<<grid, 256, 0, ctx->stream>>>( ctx->P, nullptr, j, weighted ? 1 : 0 );
    else
        weight_cost_kernel<uint16_t><<<grid, 256, 0, ctx->stream>>>( ctx->P, nullptr, j, weighted ? 1 : 0 );
    HIPCK( hipGetLastError() );
    int rc = sync_stream( ctx );
    if( rc ) return rc;
    *cost = ( (volatile unsigned *)ctx->wcost_host )[weighted ? 1 : 0];
    return X264HIP_OK;
}

// ---- getters ----------------------------------------------------------------------------------------
extern "C" int x264hip_get_lowres( x264hip_ctx *ctx, int slot, int plane, void *dst, int dst_stride )
{
    if( !ctx || !dst || !slot_ok( ctx, slot ) || plane < 0 || plane > 3 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    const int w = ctx->lw + 2 * LA_PAD, h = ctx->lh + 2 * LA_PAD;
    if( plane && !( s.rowmajor_mask & ( 1 << plane ) ) )
    {
        // the H / V / HV planes exist as strips only until somebody asks for them row-major (la_kernels.h lowres_tiles_kernel)
        const int pieces = h * ( ctx->P.stride >> 3 );
        if( ctx->p.bit_depth == 8 )
            strips_to_plane_kernel<uint8_t><<<( pieces + 255 ) / 256, 256, 0, ctx->stream>>>( (uint8_t *)s.planes, plane, ctx->P.plane_elems, ctx->P.stride, h );
        else
            strips_to_plane_kernel<uint16_t><<<( pieces + 255 ) / 256, 256, 0, ctx->stream>>>( (uint16_t *)s.planes, plane, ctx->P.plane_elems, ctx->P.stride, h );
        HIPCK( hipGetLastError() );
        s.rowmajor_mask |= 1 << plane;
    }
    HIPCK( hipMemcpy2DAsync( dst, (size_t)dst_stride * ctx->psz, s.planes + (size_t)plane * ctx->plane_bytes, (size_t)ctx->P.stride * ctx->psz,
                             (size_t)w * ctx->psz, h, hipMemcpyDeviceToHost, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    return X264HIP_OK;
}

extern "C" int x264hip_get_mvs( x264hip_ctx *ctx, int slot, int list, int dist_minus1, int16_t *mvs, int *mv_costs )
{
    if( !ctx || !slot_ok( ctx, slot ) || list < 0 || list > 1 || dist_minus1 < 0 || dist_minus1 > ctx->p.bframes ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    {
        int rc0 = ensure_field_local_by_number( ctx, slot, list, dist_minus1 );
        if( rc0 ) return rc0;
    }
    std::vector<unsigned long long> g( ctx->n_mb );
    HIPCK( hipMemcpyAsync( g.data(), s.mvq[list][dist_minus1], ctx->n_mb * sizeof( unsigned long long ), hipMemcpyDeviceToHost, ctx->stream ) );
    if( mv_costs )
        HIPCK( hipMemcpyAsync( mv_costs, s.mvcost[list][dist_minus1], ctx->n_mb * sizeof( int ), hipMemcpyDeviceToHost, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    if( mvs )
        for( int i = 0; i < ctx->n_mb; i++ )
        {
            unsigned w = (unsigned)g[i];
            mvs[2 * i] = (int16_t)( w & 0xFFFF );
            mvs[2 * i + 1] = (int16_t)( w >> 16 );
        }
    return X264HIP_OK;
}

extern "C" int x264hip_get_lowres_costs( x264hip_ctx *ctx, int slot, int dist_p0, int dist_p1, uint16_t *costs, int *row_satds )
{
    if( !ctx || !slot_ok( ctx, slot ) || dist_p0 < 0 || dist_p1 < 0 || dist_p0 > ctx->p.bframes + 1 || dist_p1 > ctx->p.bframes + 1 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    const int idx = dist_p0 * ( ctx->p.bframes + 2 ) + dist_p1;
    if( costs && dist_p0 + dist_p1 <= ctx->p.bframes + 1 )
    {
        int rc0 = ensure_cell_local( ctx, slot, dist_p0, dist_p1 );
        if( rc0 ) return rc0;
    }
    if( costs )
        HIPCK( hipMemcpyAsync( costs, s.lowres_costs + (size_t)s.cell_at[idx] * ctx->n_mb, ctx->n_mb * sizeof( uint16_t ), hipMemcpyDeviceToHost, ctx->stream ) );
    if( row_satds )
        HIPCK( hipMemcpyAsync( row_satds, s.row_satds + (size_t)s.cell_at[idx] * ctx->P.mb_h, ctx->P.mb_h * sizeof( int ), hipMemcpyDeviceToHost, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    return X264HIP_OK;
}

extern "C" int x264hip_get_intra_costs( x264hip_ctx *ctx, int slot, uint16_t *intra_costs )
{
    return x264hip_get_lowres_costs( ctx, slot, 0, 0, intra_costs, nullptr );
}

extern "C" int x264hip_get_inv_qscale( x264hip_ctx *ctx, int slot, uint16_t *inv_qscale )
{
    if( !ctx || !slot_ok( ctx, slot ) || !inv_qscale ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    HIPCK( hipMemcpyAsync( inv_qscale, ctx->slots[slot].inv_qscale, ctx->n_mb * sizeof( uint16_t ), hipMemcpyDeviceToHost, ctx->stream ) );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    return X264HIP_OK;
}

extern "C" int x264hip_last_search_ms( x264hip_ctx *ctx, float *ms, int *n_searches, int *n_blocks )
{
    if( !ctx || !ms ) return X264HIP_EINVAL;
    if( !ctx->ev_valid ) return X264HIP_ESTATE;
    HIPCK( hipEventSynchronize( ctx->ev_stop ) );
    HIPCK( hipEventElapsedTime( ms, ctx->ev_start, ctx->ev_stop ) );
    if( n_searches ) *n_searches = ctx->last_n_search;
    if( n_blocks ) *n_blocks = ctx->last_n_blocks;
    return X264HIP_OK;
}

extern "C" int x264hip_search_profile( x264hip_ctx *ctx, int enable, double *total_ms, uint64_t *launches, uint64_t *searches )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    int rc = prof_drain( ctx );
    if( rc ) return rc;
    if( total_ms ) *total_ms = ctx->prof_ms;
    if( launches ) *launches = ctx->prof_launches;
    if( searches ) *searches = ctx->prof_searches;
    if( enable >= 0 )
    {
        ctx->prof_ms = 0; ctx->prof_launches = 0; ctx->prof_searches = 0;
        ctx->prof_lat_ms = 0; ctx->prof_lat_launches = 0; ctx->prof_lat_searches = 0;
        ctx->prof_cell_ms = 0; ctx->prof_cell_launches = 0; ctx->prof_cells = 0;
        for( int k = 0; k < X264HIP_KPROF_CLASSES; k++ ) { ctx->kprof_ms[k] = 0; ctx->kprof_launches[k] = 0; ctx->kprof_units[k] = 0; }
        ctx->prof_on = enable;
        if( enable && ctx->prof_ev.empty() )
        {
            ctx->prof_ev.resize( 2048 );
            for( auto &e : ctx->prof_ev )
                HIPCK( hipEventCreate( &e ) );
        }
    }
    return X264HIP_OK;
}

// the search launches of the profiled region that cannot fill the chip (fewer waves than wave slots; me_latency_kernel runs them unless the device is shared);
// call before x264hip_search_profile, which resets the totals
extern "C" int x264hip_search_profile_latency( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *searches )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE;
    int rc = prof_drain( ctx );
    if( rc ) return rc;
    if( total_ms ) *total_ms = ctx->prof_lat_ms;
    if( launches ) *launches = ctx->prof_lat_launches;
    if( searches ) *searches = ctx->prof_lat_searches;
    return X264HIP_OK;
}

extern "C" int x264hip_cell_profile( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *cells )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    int rc = prof_drain( ctx );
    if( rc ) return rc;
    if( total_ms ) *total_ms = ctx->prof_cell_ms;
    if( launches ) *launches = ctx->prof_cell_launches;
    if( cells ) *cells = ctx->prof_cells;
    return X264HIP_OK;
}

extern "C" int x264hip_kernel_profile( x264hip_ctx *ctx, double *total_ms, uint64_t *launches, uint64_t *units )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    int rc = prof_drain( ctx );
    if( rc ) return rc;
    for( int k = 0; k < X264HIP_KPROF_CLASSES; k++ )
    {
        if( total_ms ) total_ms[k] = ctx->kprof_ms[k];
        if( launches ) launches[k] = ctx->kprof_launches[k];
        if( units ) units[k] = ctx->kprof_units[k];
    }
    return X264HIP_OK;
}

extern "C" int x264hip_host_transfer_stats( x264hip_ctx *ctx, uint64_t out[3] )
{
    if( !ctx || !out ) return X264HIP_EINVAL;
    out[0] = ctx->h2d_bytes; out[1] = ctx->h2d_direct; out[2] = ctx->h2d_staged;
    return X264HIP_OK;
}

extern "C" int x264hip_host_transfer_stats2( x264hip_ctx *ctx, uint64_t out[2] )
{
    if( !ctx || !out ) return X264HIP_EINVAL;
    out[0] = ctx->h2d_merged; out[1] = ctx->h2d_by_kernel;
    return X264HIP_OK;
}

extern "C" int x264hip_counters( x264hip_ctx *ctx, uint64_t *out, int n )
{
    if( !ctx || !out ) return X264HIP_EINVAL;
    for( int i = 0; i < n && i < 16; i++ ) out[i] = ctx->counters[i];
    return X264HIP_OK;
}

// ---- batched primitives ------------------------------------------------------------------------------
static int fill_multi( MultiPtrs &M, int n, const void *const *a, const void *const *b, const void *const *c, const void *const *d )
{
    memset( &M, 0, sizeof( M ) );
    if( n < 1 || n > MULTI_PLANES_MAX || !a || !b || !c || !d ) return X264HIP_EINVAL;
    for( int i = 0; i < n; i++ )
    {
        if( !a[i] || !b[i] || !c[i] || !d[i] ) return X264HIP_EINVAL;
        M.p[0][i] = (void *)a[i]; M.p[1][i] = (void *)b[i]; M.p[2][i] = (void *)c[i]; M.p[3][i] = (void *)d[i];
    }
    M.n = n;
    return X264HIP_OK;
}

static int pixel_cmp_launch( x264hip_ctx *ctx, int satd, int size_idx, const void *fenc_plane, const void *ref_plane, int stride,
                             int blocks_w, int blocks_h, const int16_t *mv_dev, int *out_dev, const MultiPtrs &M );
extern "C" int x264hip_pixel_cmp_batch( x264hip_ctx *ctx, int satd, int size_idx, const void *fenc_plane, const void *ref_plane, int stride,
                                        int blocks_w, int blocks_h, const int16_t *mv_dev, int *out_dev )
{
    MultiPtrs none;
    memset( &none, 0, sizeof( none ) );
    return pixel_cmp_launch( ctx, satd, size_idx, fenc_plane, ref_plane, stride, blocks_w, blocks_h, mv_dev, out_dev, none );
}
// n_pairs independent frame pairs of one geometry in ONE launch (pair = blockIdx.z): the BASELINE-defined primitive input -- all blocks of one
// 4K frame pair -- is a 5 us launch, half of it launch and ramp; a window's worth of pairs runs at the rate of a large field
extern "C" int x264hip_pixel_cmp_batch_multi( x264hip_ctx *ctx, int satd, int size_idx, int n_pairs, const void *const *fenc_planes, const void *const *ref_planes, int stride,
                                              int blocks_w, int blocks_h, const int16_t *const *mv_dev, int *const *out_dev )
{
    MultiPtrs M;
    int rc = fill_multi( M, n_pairs, fenc_planes, ref_planes, (const void *const *)mv_dev, (const void *const *)out_dev );
    if( rc ) return rc;
    return pixel_cmp_launch( ctx, satd, size_idx, fenc_planes[0], ref_planes[0], stride, blocks_w, blocks_h, mv_dev[0], out_dev[0], M );
}
static int pixel_cmp_launch( x264hip_ctx *ctx, int satd, int size_idx, const void *fenc_plane, const void *ref_plane, int stride,
                             int blocks_w, int blocks_h, const int16_t *mv_dev, int *out_dev, const MultiPtrs &M )
{
    if( !ctx || !fenc_plane || !ref_plane || !mv_dev || !out_dev ) return X264HIP_EINVAL;
    static const int sz_w[7] = { 16, 16, 8, 8, 8, 4, 4 }, sz_h[7] = { 16, 8, 16, 8, 4, 8, 4 }; // common/pixel.h:37-59, PIXEL_16x16 .. PIXEL_4x4
    if( size_idx < 0 || size_idx > 6 || blocks_w <= 0 || blocks_h <= 0 ) return X264HIP_EINVAL;
    const int bwid = sz_w[size_idx], bhgt = sz_h[size_idx];
    if( ( blocks_w * bwid ) % 16 || ( blocks_h * bhgt ) % 16 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int rw = blocks_w * bwid / 16, rh = blocks_h * bhgt / 16;
    // region rows per lane (loads in flight per lane): 4 unless the field is too small to fill the chip with a quarter of the
    // workgroups (X264HIP_CMP_ROWS = 1 / 2 / 4 overrides it: A/B aid)
    static const int rows_env = getenv( "X264HIP_CMP_ROWS" ) ? atoi( getenv( "X264HIP_CMP_ROWS" ) ) : 0;
    const int wgs1 = ( ( rw + 15 ) / 16 ) * rh * ( M.n ? M.n : 1 );
    const int rr = rows_env == 1 || rows_env == 2 || rows_env == 4 || rows_env == 8 ? rows_env : wgs1 >= 32 * ctx->n_cu ? 4 : wgs1 >= 4 * ctx->n_cu ? 2 : 1;
    const dim3 grd( ( rw + 15 ) / 16, ( rh + rr - 1 ) / rr, M.n ? M.n : 1 );
    static const int xcd_bands = !( getenv( "X264HIP_CMP_BANDS" ) && atoi( getenv( "X264HIP_CMP_BANDS" ) ) == 0 ); // 0: workgroups in launch order (A/B aid)
#define CMP_LAUNCH( T, BW, BH, D ) \
    do { \
        if( rr == 8 ) pixel_cmp_batch_kernel<T, BW, BH, D, 8><<<grd, 256, 0, ctx->stream>>>( (const T *)fenc_plane, (const T *)ref_plane, stride, rw, rh, mv_dev, out_dev, xcd_bands, M ); \
        else if( rr == 4 ) pixel_cmp_batch_kernel<T, BW, BH, D, 4><<<grd, 256, 0, ctx->stream>>>( (const T *)fenc_plane, (const T *)ref_plane, stride, rw, rh, mv_dev, out_dev, xcd_bands, M ); \
        else if( rr == 2 ) pixel_cmp_batch_kernel<T, BW, BH, D, 2><<<grd, 256, 0, ctx->stream>>>( (const T *)fenc_plane, (const T *)ref_plane, stride, rw, rh, mv_dev, out_dev, xcd_bands, M ); \
        else pixel_cmp_batch_kernel<T, BW, BH, D, 1><<<grd, 256, 0, ctx->stream>>>( (const T *)fenc_plane, (const T *)ref_plane, stride, rw, rh, mv_dev, out_dev, xcd_bands, M ); \
    } while( 0 )
#define CMP_METRIC( T, BW, BH ) do { if( satd ) CMP_LAUNCH( T, BW, BH, true ); else CMP_LAUNCH( T, BW, BH, false ); } while( 0 )
#define CMP_SIZE( T ) \
    do { \
        switch( size_idx ) \
        { \
            case 0: CMP_METRIC( T, 16, 16 ); break; \
            case 1: CMP_METRIC( T, 16, 8 ); break; \
            case 2: CMP_METRIC( T, 8, 16 ); break; \
            case 3: CMP_METRIC( T, 8, 8 ); break; \
            case 4: CMP_METRIC( T, 8, 4 ); break; \
            case 5: CMP_METRIC( T, 4, 8 ); break; \
            default: CMP_METRIC( T, 4, 4 ); break; \
        } \
    } while( 0 )
    if( ctx->p.bit_depth == 8 )
        CMP_SIZE( uint8_t );
    else
        CMP_SIZE( uint16_t );
#undef CMP_METRIC
#undef CMP_SIZE
#undef CMP_LAUNCH
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

// reqs_host / out_host: the caller's table and result array in host memory (x264hip_me_search_batch: synchronous), or
// reqs_dev / out_dev: both on the device (x264hip_me_search_batch_dev: enqueued; one class of methods, declared by the caller).
template <typename T>
static int me_search_batch_t( x264hip_ctx *ctx, int n, const x264hip_me_request *reqs_host, const x264hip_me_request *reqs_dev, int dev_method, int dev_range,
                              const void *fenc_plane, intptr_t fenc_stride,
                              const void *const ref_planes[4], intptr_t ref_stride, const uint16_t *integral, intptr_t integral_lower,
                              const uint16_t *cost_mv, int *out_host_user, int *out_dev_user )
{
    // TESA keeps the candidates that pass its SAD threshold: at most ( width + 4 ) * ( rows + 1 ) of them, 12 bytes each, with
    // width <= 2 * me_range + 4 and rows <= 2 * me_range + 1 (me.c:651-655)
    auto tesa_bytes = []( int me_range ) { return (size_t)( 2 * me_range + 8 ) * ( 2 * me_range + 2 ) * 12; };
    const bool on_dev = reqs_dev != nullptr;
    size_t scratch_total = 0;
    if( on_dev )
        scratch_total = dev_method == 4 ? tesa_bytes( dev_range ) * (size_t)n : 0;
    else
        for( int i = 0; i < n; i++ )
            if( reqs_host[i].me_method == 4 ) scratch_total += tesa_bytes( reqs_host[i].me_range );
    // One device allocation and one pinned host allocation for the call, kept by the context and grown when a call needs more (an
    // allocation per call cost more than the searches of a 1080p frame).  What crosses the host link is the caller's table as it is
    // (116 bytes per request), two ints per request and the results; the kernels' own records are made from it on the device
    // (me_translate_kernel).
    const size_t b_scratch = align_up( scratch_total, 256 ), b_table = align_up( sizeof( MfReq<T> ) * n, 256 ), b_mvc = align_up( (size_t)n * MF_MVC_MAX * 2 * sizeof( int16_t ), 256 ),
                 b_nmvc = align_up( sizeof( int ) * n, 256 ), b_out = align_up( sizeof( int ) * 4 * n, 256 ), b_index = align_up( sizeof( int ) * n, 256 ),
                 b_raw = on_dev ? 0 : align_up( sizeof( x264hip_me_request ) * (size_t)n, 256 ), b_off = on_dev ? 0 : align_up( sizeof( unsigned long long ) * n, 256 );
    const size_t b_host = b_raw + b_off + b_index; // what the host writes: contiguous
    const size_t need = b_scratch + b_table + b_mvc + b_nmvc + b_host + b_out;
    if( need > ctx->me_pool_bytes )
    {
        HIPCK( hipStreamSynchronize( ctx->stream ) ); // (an enqueued batch may still be using the block)
        if( ctx->me_pool ) (void)hipFree( ctx->me_pool );
        ctx->me_pool = nullptr; ctx->me_pool_bytes = 0;
        if( hipMalloc( &ctx->me_pool, need + ( need >> 2 ) ) != hipSuccess ) return X264HIP_ENOMEM;
        ctx->me_pool_bytes = need + ( need >> 2 );
    }
    if( !on_dev && b_host + b_out > ctx->me_host_bytes )
    {
        if( ctx->me_host ) (void)hipHostFree( ctx->me_host );
        ctx->me_host = nullptr; ctx->me_host_bytes = 0;
        const size_t want = b_host + b_out + ( ( b_host + b_out ) >> 2 );
        if( hipHostMalloc( &ctx->me_host, want ) != hipSuccess ) return X264HIP_ENOMEM;
        ctx->me_host_bytes = want;
    }
    char *scratch = ctx->me_pool;
    MfReq<T> *table_dev = (MfReq<T> *)( scratch + b_scratch );
    int16_t *mvc_dev = (int16_t *)( (char *)table_dev + b_table );
    int *n_mvc_dev = (int *)( (char *)mvc_dev + b_mvc );
    char *host_dev = (char *)n_mvc_dev + b_nmvc; // the device copy of the host-written block: raw requests, TESA offsets, index
    const x264hip_me_request *raw_dev = on_dev ? reqs_dev : (const x264hip_me_request *)host_dev;
    unsigned long long *off_dev = (unsigned long long *)( host_dev + b_raw ); // 64-bit: a batch's TESA scratch passes 4 GiB at ~20 k requests of range 64
    int *index_dev = (int *)( host_dev + b_raw + b_off ), *out_dev = on_dev ? out_dev_user : (int *)( host_dev + b_host );
    int rc = X264HIP_OK;
#define MECK( call ) do { if( ( call ) != hipSuccess ) { ctx->broken = 1; return X264HIP_EDEVICE; } } while( 0 )
    int n_pat = on_dev ? ( dev_method < 3 ? n : 0 ) : 0;
    if( !on_dev )
    {
        x264hip_me_request *raw = (x264hip_me_request *)ctx->me_host;
        unsigned long long *off = (unsigned long long *)( ctx->me_host + b_raw );
        int *idx = (int *)( ctx->me_host + b_raw + b_off );
        memcpy( raw, reqs_host, sizeof( x264hip_me_request ) * (size_t)n );
        size_t t = 0;
        for( int i = 0; i < n; i++ )
        {
            off[i] = (unsigned long long)t;
            if( reqs_host[i].me_method == 4 ) t += tesa_bytes( reqs_host[i].me_range );
        }
        // one launch per class of methods: the pattern searches (DIA, HEX, UMH) and the exhaustive ones (ESA, TESA) are separate builds
        for( int i = 0; i < n; i++ ) if( reqs_host[i].me_method < 3 ) idx[n_pat++] = i;
        for( int i = 0, k = n_pat; i < n; i++ ) if( reqs_host[i].me_method >= 3 ) idx[k++] = i;
        MECK( upload_async( ctx, host_dev, ctx->me_host, b_host, ctx->stream ) );
    }
    MeTranslate A;
    memset( &A, 0, sizeof( A ) );
    A.fenc_plane = fenc_plane; A.fenc_stride = (long)fenc_stride; A.ref_stride = (long)ref_stride; A.integral_lower = (long)integral_lower;
    for( int k = 0; k < 4; k++ ) A.ref[k] = ref_planes[k];
    A.integral = integral; A.cost_mv = cost_mv; A.scratch = scratch;
    A.scratch_off = on_dev ? nullptr : off_dev; A.uniform_scratch = on_dev ? tesa_bytes( dev_range ) : 0;
    A.force_method = on_dev ? dev_method : -1; A.n = n;
    me_translate_kernel<T><<<( n + 255 ) / 256, 256, 0, ctx->stream>>>( raw_dev, A, table_dev, mvc_dev, n_mvc_dev, on_dev ? index_dev : nullptr );
    MECK( hipEventRecord( ctx->ev_start, ctx->stream ) );
    {
        // a wave per request: its 64 lanes run the search in lock step, block costs are computed across the wave (four samples per
        // lane) and the exhaustive scans (ESA, TESA) cost 64 candidates per step.  X264HIP_ME_FULL_SCALAR=1 sends everything through
        // the one-thread form instead (the device reference the cooperative form is checked against)
        static const bool all_scalar = getenv( "X264HIP_ME_FULL_SCALAR" ) != nullptr;
        if( all_scalar )
            me_full_list_kernel<T><<<( n + 63 ) / 64, 64, 0, ctx->stream>>>( table_dev, mvc_dev, n_mvc_dev, index_dev, n, out_dev );
        else
        {
            if( n_pat )
                me_full_coop_kernel<T, 1><<<n_pat, 64, 0, ctx->stream>>>( table_dev, mvc_dev, n_mvc_dev, index_dev, n_pat, out_dev );
            if( n - n_pat )
                me_full_coop_kernel<T, 2><<<n - n_pat, 64, 0, ctx->stream>>>( table_dev, mvc_dev, n_mvc_dev, index_dev + n_pat, n - n_pat, out_dev );
        }
    }
    MECK( hipGetLastError() );
    // (x264hip_last_search_ms reports the device time of the batch's kernels: n searches of one block each)
    MECK( hipEventRecord( ctx->ev_stop, ctx->stream ) );
    ctx->ev_valid = 1; ctx->last_n_search = n; ctx->last_n_blocks = n;
    if( !on_dev )
    {
        int *out_pinned = (int *)( ctx->me_host + b_host );
        MECK( hipMemcpyAsync( out_pinned, out_dev, sizeof( int ) * 4 * n, hipMemcpyDeviceToHost, ctx->stream ) );
        MECK( hipStreamSynchronize( ctx->stream ) );
        memcpy( out_host_user, out_pinned, sizeof( int ) * 4 * n );
    }
#undef MECK
    return rc;
}

extern "C" int x264hip_me_search_batch( x264hip_ctx *ctx, int n, const x264hip_me_request *reqs, const void *fenc_plane_dev, intptr_t fenc_stride,
                                        const void *const ref_planes_dev[4], intptr_t ref_stride, const uint16_t *integral_dev, intptr_t integral_lower,
                                        const uint16_t *cost_mv_dev, int *out )
{
    if( !ctx || n <= 0 || !reqs || !fenc_plane_dev || !ref_planes_dev || !cost_mv_dev || !out ) return X264HIP_EINVAL;
    for( int k = 0; k < 4; k++ )
        if( !ref_planes_dev[k] ) return X264HIP_EINVAL;
    for( int i = 0; i < n; i++ )
    {
        const x264hip_me_request &q = reqs[i];
        if( q.i_pixel < 0 || q.i_pixel > 6 || q.me_method < 0 || q.me_method > 4 || q.subpel_refine < 0 || q.subpel_refine > 11 ||
            q.n_mvc < 0 || q.n_mvc > X264HIP_ME_MVC_MAX || q.me_range < 1 || ( q.me_method == 4 && ( !integral_dev || q.me_range > 64 ) ) )
            return X264HIP_EINVAL;
    }
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    return ctx->p.bit_depth == 8
           ? me_search_batch_t<uint8_t>( ctx, n, reqs, nullptr, 0, 0, fenc_plane_dev, fenc_stride, ref_planes_dev, ref_stride, integral_dev, integral_lower, cost_mv_dev, out, nullptr )
           : me_search_batch_t<uint16_t>( ctx, n, reqs, nullptr, 0, 0, fenc_plane_dev, fenc_stride, ref_planes_dev, ref_stride, integral_dev, integral_lower, cost_mv_dev, out, nullptr );
}

// the same with the request table and the results ON the device (a front end that generates its requests there): enqueued on the
// context's stream, nothing crosses the host link.  The table holds one method (me_method: every request is searched with it) and
// me_range_max bounds the requests' me_range (it sizes the TESA candidate lists).  The fields of the requests are not validated.
extern "C" int x264hip_me_search_batch_dev( x264hip_ctx *ctx, int n, const x264hip_me_request *reqs_dev, const void *fenc_plane_dev, intptr_t fenc_stride,
                                            const void *const ref_planes_dev[4], intptr_t ref_stride, const uint16_t *integral_dev, intptr_t integral_lower,
                                            const uint16_t *cost_mv_dev, int me_method, int me_range_max, int *out_dev )
{
    if( !ctx || n <= 0 || !reqs_dev || !fenc_plane_dev || !ref_planes_dev || !cost_mv_dev || !out_dev || me_method < 0 || me_method > 4 || me_range_max < 1 ||
        ( me_method == 4 && ( !integral_dev || me_range_max > 64 ) ) )
        return X264HIP_EINVAL;
    for( int k = 0; k < 4; k++ )
        if( !ref_planes_dev[k] ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    return ctx->p.bit_depth == 8
           ? me_search_batch_t<uint8_t>( ctx, n, nullptr, reqs_dev, me_method, me_range_max, fenc_plane_dev, fenc_stride, ref_planes_dev, ref_stride, integral_dev, integral_lower,
                                         cost_mv_dev, nullptr, out_dev )
           : me_search_batch_t<uint16_t>( ctx, n, nullptr, reqs_dev, me_method, me_range_max, fenc_plane_dev, fenc_stride, ref_planes_dev, ref_stride, integral_dev, integral_lower,
                                          cost_mv_dev, nullptr, out_dev );
}

extern "C" int x264hip_integral_init( x264hip_ctx *ctx, const void *plane_dev, intptr_t stride, int width, int height, uint16_t *sum8_dev, uint16_t *sum4_dev )
{
    if( !ctx || !plane_dev || !sum8_dev || !sum4_dev || width < 8 || height < 8 || stride < width ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    uint16_t *h = nullptr; // row sums of 8 and of 4, one plane each
    const size_t plane = (size_t)stride * height;
    if( hipMalloc( &h, 2 * plane * sizeof( uint16_t ) ) != hipSuccess ) return X264HIP_ENOMEM;
    const dim3 grd( ( width + 255 ) / 256, height );
    if( ctx->p.bit_depth == 8 )
        integral_rows_kernel<uint8_t><<<grd, 256, 0, ctx->stream>>>( (const uint8_t *)plane_dev, (long)stride, width, height, h, h + plane );
    else
        integral_rows_kernel<uint16_t><<<grd, 256, 0, ctx->stream>>>( (const uint16_t *)plane_dev, (long)stride, width, height, h, h + plane );
    integral_cols_kernel<<<grd, 256, 0, ctx->stream>>>( h, h + plane, (long)stride, width, height, sum8_dev, sum4_dev );
    int rc = hipGetLastError() == hipSuccess && hipStreamSynchronize( ctx->stream ) == hipSuccess ? X264HIP_OK : X264HIP_EDEVICE;
    (void)hipFree( h );
    return rc;
}

extern "C" int x264hip_pixel_metric_batch( x264hip_ctx *ctx, int metric, int size_idx, const void *a_plane, const void *b_plane, intptr_t stride,
                                           int blocks_w, int blocks_h, uint64_t *out_dev )
{
    static const int sizes[7][2] = { { 16, 16 }, { 16, 8 }, { 8, 16 }, { 8, 8 }, { 8, 4 }, { 4, 8 }, { 4, 4 } }; // PIXEL_16x16 .. PIXEL_4x4
    if( !ctx || !a_plane || !out_dev || size_idx < 0 || size_idx > 6 || blocks_w <= 0 || blocks_h <= 0 ) return X264HIP_EINVAL;
    const int w = sizes[size_idx][0], h = sizes[size_idx][1];
    const bool two = metric == X264HIP_METRIC_SSD || metric == X264HIP_METRIC_SA8D || metric == X264HIP_METRIC_ASD8;
    if( ( two && !b_plane ) || stride < (intptr_t)blocks_w * w ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const dim3 grd( ( blocks_w + 63 ) / 64, blocks_h );
    bool launched = false;
#define BM_LAUNCH( T, M, W, H ) \
    if( !launched && metric == M && w == W && h == H ) \
    { \
        block_metric_kernel<T, M, W, H><<<grd, 64, 0, ctx->stream>>>( (const T *)a_plane, two ? (const T *)b_plane : (const T *)nullptr, (long)stride, \
                                                                      blocks_w, (unsigned long long *)out_dev ); \
        launched = true; \
    }
#define BM_ALL( T ) \
    BM_LAUNCH( T, BM_SSD, 16, 16 ) BM_LAUNCH( T, BM_SSD, 16, 8 ) BM_LAUNCH( T, BM_SSD, 8, 16 ) BM_LAUNCH( T, BM_SSD, 8, 8 ) \
    BM_LAUNCH( T, BM_SSD, 8, 4 ) BM_LAUNCH( T, BM_SSD, 4, 8 ) BM_LAUNCH( T, BM_SSD, 4, 4 ) \
    BM_LAUNCH( T, BM_SA8D, 16, 16 ) BM_LAUNCH( T, BM_SA8D, 8, 8 ) \
    BM_LAUNCH( T, BM_VAR, 16, 16 ) BM_LAUNCH( T, BM_VAR, 8, 16 ) BM_LAUNCH( T, BM_VAR, 8, 8 ) \
    BM_LAUNCH( T, BM_HADAMARD_AC, 16, 16 ) BM_LAUNCH( T, BM_HADAMARD_AC, 16, 8 ) BM_LAUNCH( T, BM_HADAMARD_AC, 8, 16 ) BM_LAUNCH( T, BM_HADAMARD_AC, 8, 8 ) \
    BM_LAUNCH( T, BM_VSAD, 16, 16 ) BM_LAUNCH( T, BM_VSAD, 16, 8 ) \
    BM_LAUNCH( T, BM_ASD8, 8, 16 ) BM_LAUNCH( T, BM_ASD8, 8, 8 )
    if( ctx->p.bit_depth == 8 ) { BM_ALL( uint8_t ) } else { BM_ALL( uint16_t ) }
#undef BM_ALL
#undef BM_LAUNCH
    if( !launched ) return X264HIP_EINVAL; // the reference has no such metric / size pair
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

static int dct_quant4x4_launch( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev, const MultiPtrs &M );
extern "C" int x264hip_frame_dct_quant4x4( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                           const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev )
{
    MultiPtrs none;
    memset( &none, 0, sizeof( none ) );
    return dct_quant4x4_launch( ctx, fenc, fenc_stride, fdec, fdec_stride, width, height, mf, bias, coefs_dev, nz_dev, none );
}
extern "C" int x264hip_frame_dct_quant4x4_multi( x264hip_ctx *ctx, int n, const void *const *fenc, intptr_t fenc_stride, const void *const *fdec, intptr_t fdec_stride,
                                                 int width, int height, const void *mf, const void *bias, void *const *coefs_dev, uint8_t *const *nz_dev )
{
    MultiPtrs M;
    int rc = fill_multi( M, n, fenc, fdec, (const void *const *)coefs_dev, (const void *const *)nz_dev );
    if( rc ) return rc;
    for( int i = 0; i < n; i++ ) if( (uintptr_t)coefs_dev[i] & 15 ) return X264HIP_EINVAL;
    return dct_quant4x4_launch( ctx, fenc[0], fenc_stride, fdec[0], fdec_stride, width, height, mf, bias, coefs_dev[0], nz_dev[0], M );
}
static int dct_quant4x4_launch( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev, const MultiPtrs &M )
{
    if( !ctx || !fenc || !fdec || !mf || !bias || !coefs_dev || !nz_dev || width <= 0 || height <= 0 || ( width & 3 ) || ( height & 3 ) ||
        fenc_stride < width || fdec_stride < width || ( (uintptr_t)coefs_dev & 15 ) )
        return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    QuantTab q;
    for( int i = 0; i < 16; i++ )
    {
        q.mf[i] = ctx->p.bit_depth == 8 ? ( (const uint16_t *)mf )[i] : ( (const uint32_t *)mf )[i];     // udctcoef (common/common.h)
        q.bias[i] = ctx->p.bit_depth == 8 ? ( (const uint16_t *)bias )[i] : ( (const uint32_t *)bias )[i];
    }
    const int bw = width / 4, bh = height / 4;
    const dim3 grd( ( bw + 255 ) / 256, bh, M.n ? M.n : 1 );
    if( ctx->p.bit_depth == 8 )
        frame_dct_quant4x4_kernel<uint8_t, int16_t><<<grd, 256, 0, ctx->stream>>>( (const uint8_t *)fenc, (long)fenc_stride, (const uint8_t *)fdec, (long)fdec_stride, bw, bh, q,
                                                                                   (int16_t *)coefs_dev, nz_dev, M );
    else
        frame_dct_quant4x4_kernel<uint16_t, int32_t><<<grd, 256, 0, ctx->stream>>>( (const uint16_t *)fenc, (long)fenc_stride, (const uint16_t *)fdec, (long)fdec_stride, bw, bh,
                                                                                    q, (int32_t *)coefs_dev, nz_dev, M );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

extern "C" int x264hip_frame_dct_quant8x8( x264hip_ctx *ctx, const void *fenc, intptr_t fenc_stride, const void *fdec, intptr_t fdec_stride, int width, int height,
                                           const void *mf, const void *bias, void *coefs_dev, uint8_t *nz_dev )
{
    if( !ctx || !fenc || !fdec || !mf || !bias || !coefs_dev || !nz_dev || width <= 0 || height <= 0 || ( width & 7 ) || ( height & 7 ) ||
        fenc_stride < width || fdec_stride < width )
        return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    QuantTab8 q;
    for( int i = 0; i < 64; i++ )
    {
        q.mf[i] = ctx->p.bit_depth == 8 ? ( (const uint16_t *)mf )[i] : ( (const uint32_t *)mf )[i];     // udctcoef (common/common.h)
        q.bias[i] = ctx->p.bit_depth == 8 ? ( (const uint16_t *)bias )[i] : ( (const uint32_t *)bias )[i];
    }
    const int bw = width / 8, bh = height / 8;
    const dim3 grd( ( bw + 63 ) / 64, bh );
    if( ctx->p.bit_depth == 8 )
        frame_dct_quant8x8_kernel<uint8_t, int16_t><<<grd, 64, 0, ctx->stream>>>( (const uint8_t *)fenc, (long)fenc_stride, (const uint8_t *)fdec, (long)fdec_stride, bw, q,
                                                                                  (int16_t *)coefs_dev, nz_dev );
    else
        frame_dct_quant8x8_kernel<uint16_t, int32_t><<<grd, 64, 0, ctx->stream>>>( (const uint16_t *)fenc, (long)fenc_stride, (const uint16_t *)fdec, (long)fdec_stride, bw, q,
                                                                                   (int32_t *)coefs_dev, nz_dev );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

static int hpel_filter_launch( x264hip_ctx *ctx, void *dsth, void *dstv, void *dstc, const void *src, intptr_t stride, int width, int height, const MultiPtrs &M );
extern "C" int x264hip_hpel_filter( x264hip_ctx *ctx, void *dsth, void *dstv, void *dstc, const void *src, intptr_t stride, int width, int height )
{
    MultiPtrs none;
    memset( &none, 0, sizeof( none ) );
    return hpel_filter_launch( ctx, dsth, dstv, dstc, src, stride, width, height, none );
}
// the three half-pel planes of n planes of one geometry in one launch (plane = blockIdx.z)
extern "C" int x264hip_hpel_filter_multi( x264hip_ctx *ctx, int n, void *const *dsth, void *const *dstv, void *const *dstc, const void *const *src, intptr_t stride,
                                          int width, int height )
{
    MultiPtrs M;
    int rc = fill_multi( M, n, (const void *const *)dsth, (const void *const *)dstv, (const void *const *)dstc, src );
    if( rc ) return rc;
    return hpel_filter_launch( ctx, dsth[0], dstv[0], dstc[0], src[0], stride, width, height, M );
}
static int hpel_filter_launch( x264hip_ctx *ctx, void *dsth, void *dstv, void *dstc, const void *src, intptr_t stride, int width, int height, const MultiPtrs &M )
{
    if( !ctx || !dsth || !dstv || !dstc || !src || width <= 0 || height <= 0 || stride < width ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const dim3 grd( ( width + HPEL_TW - 1 ) / HPEL_TW, ( height + HPEL_TH - 1 ) / HPEL_TH );
    static const bool tiled8 = getenv( "X264HIP_HPEL_TILED" ) != nullptr; // the LDS-tiled kernel instead of the streaming ones (A/B runs)
    if( tiled8 && M.n ) return X264HIP_EINVAL; // (the comparison kernel has no multi-plane form)
    const unsigned nz = M.n ? M.n : 1;
    if( ctx->p.bit_depth == 8 && !tiled8 )
        hpel_stream_kernel<<<dim3( ( width + HPS_W - 1 ) / HPS_W, ( height + HPS_R - 1 ) / HPS_R, nz ), 64, 0, ctx->stream>>>(
            (uint8_t *)dsth, (uint8_t *)dstv, (uint8_t *)dstc, (const uint8_t *)src, (int)stride, width, height, M );
    else if( ctx->p.bit_depth == 8 )
        hpel_filter_kernel<uint8_t><<<grd, 256, 0, ctx->stream>>>( (uint8_t *)dsth, (uint8_t *)dstv, (uint8_t *)dstc, (const uint8_t *)src, (long)stride, width, height,
                                                                   ctx->P.pixel_max );
    else if( !tiled8 )
        hpel_stream16_kernel<<<dim3( ( width + HPS_W - 1 ) / HPS_W, ( height + HPS_R - 1 ) / HPS_R, nz ), 64, 0, ctx->stream>>>(
            (uint16_t *)dsth, (uint16_t *)dstv, (uint16_t *)dstc, (const uint16_t *)src, (int)stride, width, height, ctx->P.pixel_max, M );
    else
        hpel_filter_kernel<uint16_t><<<grd, 256, 0, ctx->stream>>>( (uint16_t *)dsth, (uint16_t *)dstv, (uint16_t *)dstc, (const uint16_t *)src, (long)stride, width,
                                                                    height, ctx->P.pixel_max );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

// x264_frame_filter for a whole frame (common/mc.c:704-784) with the border work around it (x264_frame_expand_border,
// x264_frame_expand_border_filtered, common/frame.c:556-623): what a reconstructed frame goes through before it can serve as a
// reference.  The reference does this macroblock row by macroblock row as rows finish; every output depends only on the luma
// samples, so the whole-frame order gives the same planes (checked against planes recorded from the reference).
template <typename T>
static int frame_filter_t( x264hip_ctx *ctx, const T *luma, intptr_t luma_stride, int width, int height, T *const planes[4], intptr_t stride, int padh, int padv,
                           uint16_t *sum8, uint16_t *sum4 )
{
    const int pw = width + 2 * padh, ph = height + 2 * padv;
    const dim3 grd( ( pw + 255 ) / 256, ph );
    // 1. the picture into plane 0, its border replicated (x264_frame_expand_border)
    expand_border_kernel<T><<<grd, 256, 0, ctx->stream>>>( planes[0], (long)stride, luma, (long)luma_stride, 0, 0, width - 1, 0, height - 1, -padh, width + padh - 1, -padv );
    // 2. the three half-pel planes over the picture plus 8 samples all round (mc.c:706-726)
    const long offs = -8 * (long)stride - 8;
    MultiPtrs one_plane;
    memset( &one_plane, 0, sizeof( one_plane ) );
    if constexpr( sizeof( T ) == 1 )
        hpel_stream_kernel<<<dim3( ( width + 16 + HPS_W - 1 ) / HPS_W, ( height + 16 + HPS_R - 1 ) / HPS_R ), 64, 0, ctx->stream>>>(
            planes[1] + offs, planes[2] + offs, planes[3] + offs, planes[0] + offs, (int)stride, width + 16, height + 16, one_plane );
    else
        hpel_stream16_kernel<<<dim3( ( width + 16 + HPS_W - 1 ) / HPS_W, ( height + 16 + HPS_R - 1 ) / HPS_R ), 64, 0, ctx->stream>>>(
            planes[1] + offs, planes[2] + offs, planes[3] + offs, planes[0] + offs, (int)stride, width + 16, height + 16, ctx->P.pixel_max, one_plane );
    // 3. their borders from the last trustworthy filtered samples: 4 columns / 8 rows outside the picture (frame.c:599-623)
    for( int k = 1; k < 4; k++ )
        expand_border_kernel<T><<<grd, 256, 0, ctx->stream>>>( planes[k], (long)stride, planes[k], (long)stride, 1, -4, width + 3, -8, height + 7, -padh, width + padh - 1, -padv );
    HIPCK( hipGetLastError() );
    // 4. the integral planes of the padded luma plane (mc.c:757-783)
    if( sum8 && sum4 )
        return x264hip_integral_init( ctx, planes[0] - (long)padv * stride - padh, stride, pw, ph, sum8, sum4 );
    return X264HIP_OK;
}

extern "C" int x264hip_frame_filter( x264hip_ctx *ctx, const void *luma_dev, intptr_t luma_stride, int width, int height, void *const planes_dev[4], intptr_t stride,
                                     int padh, int padv, uint16_t *sum8_dev, uint16_t *sum4_dev )
{
    if( !ctx || !luma_dev || !planes_dev || width < 16 || height < 16 || luma_stride < width || padh < 8 || padv < 8 || stride < width + 2 * padh ||
        ( !sum8_dev ) != ( !sum4_dev ) )
        return X264HIP_EINVAL;
    for( int k = 0; k < 4; k++ )
        if( !planes_dev[k] ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    if( ctx->p.bit_depth == 8 )
    {
        uint8_t *pl[4] = { (uint8_t *)planes_dev[0], (uint8_t *)planes_dev[1], (uint8_t *)planes_dev[2], (uint8_t *)planes_dev[3] };
        return frame_filter_t<uint8_t>( ctx, (const uint8_t *)luma_dev, luma_stride, width, height, pl, stride, padh, padv, sum8_dev, sum4_dev );
    }
    uint16_t *pl[4] = { (uint16_t *)planes_dev[0], (uint16_t *)planes_dev[1], (uint16_t *)planes_dev[2], (uint16_t *)planes_dev[3] };
    return frame_filter_t<uint16_t>( ctx, (const uint16_t *)luma_dev, luma_stride, width, height, pl, stride, padh, padv, sum8_dev, sum4_dev );
}

extern "C" int x264hip_device_copy( x264hip_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes )
{
    if( !ctx || !dst_dev || !src_dev || ( bytes & 15 ) || ( (uintptr_t)dst_dev & 15 ) || ( (uintptr_t)src_dev & 15 ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const size_t n16 = bytes / 16;
    // X264HIP_COPY = "<unroll><n|t>[,<workgroups per CU>]" picks another form of the kernel (A/B aid); default: 2 loads per lane in flight,
    // non-temporal, 16 workgroups of 256 per CU (measured on MI355X over 512 MB: 6.27 TB/s; one load per lane, 32 workgroups per CU,
    // the round-2 form: 5.35; four or eight loads per lane in 4 ... 8 workgroups per CU: 5.0 ... 5.9)
    static const char *cv = getenv( "X264HIP_COPY" );
    int unroll = 2, nt = 1, per_cu = 16;
    if( cv && cv[0] )
    {
        unroll = cv[0] - '0'; nt = cv[1] == 't';
        if( cv[1] && cv[2] == ',' ) per_cu = std::max( 1, atoi( cv + 3 ) );
    }
    if( n16 )
    {
        const size_t chunk = (size_t)256 * unroll;
        const int grid = (int)std::min<size_t>( ( n16 + chunk - 1 ) / chunk, (size_t)ctx->n_cu * per_cu );
        const copy_v4u *sp = (const copy_v4u *)src_dev; copy_v4u *dp = (copy_v4u *)dst_dev;
        switch( unroll * 2 + nt )
        {
            case 2: copy16_kernel<1, false><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 3: copy16_kernel<1, true><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 4: copy16_kernel<2, false><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 8: copy16_kernel<4, false><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 16: copy16_kernel<8, false><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 17: copy16_kernel<8, true><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            case 9: copy16_kernel<4, true><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
            default: copy16_kernel<2, true><<<grid, 256, 0, ctx->stream>>>( sp, dp, n16 ); break;
        }
    }
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

extern "C" int x264hip_frame_init_lowres_core( x264hip_ctx *ctx, const void *src0, void *dst0, void *dsth, void *dstv, void *dstc,
                                               intptr_t src_stride, intptr_t dst_stride, int width, int height )
{
    if( !ctx || !src0 || !dst0 || !dsth || !dstv || !dstc || width <= 0 || height <= 0 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    dim3 grd( ( width + 255 ) / 256, height );
    if( ctx->p.bit_depth == 8 )
        lowres_core_kernel<uint8_t><<<grd, 256, 0, ctx->stream>>>( (const uint8_t *)src0, (uint8_t *)dst0, (uint8_t *)dsth, (uint8_t *)dstv,
                                                                   (uint8_t *)dstc, (long)src_stride, (long)dst_stride, width, height );
    else
        lowres_core_kernel<uint16_t><<<grd, 256, 0, ctx->stream>>>( (const uint16_t *)src0, (uint16_t *)dst0, (uint16_t *)dsth, (uint16_t *)dstv,
                                                                    (uint16_t *)dstc, (long)src_stride, (long)dst_stride, width, height );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

extern "C" int x264hip_dct_quant_batch( x264hip_ctx *ctx, int is8x8, int n_blocks, const void *fenc, const void *fdec, const void *mf,
                                        const void *bias, void *coefs_out, int *nz_out )
{
    if( !ctx || n_blocks <= 0 || !fenc || !fdec || !mf || !bias || !coefs_out || !nz_out ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int N = is8x8 ? 8 : 4, psz = ctx->psz, csz = ctx->p.bit_depth == 8 ? 2 : 4;
    const size_t fe_b = (size_t)n_blocks * N * 16 * psz, fd_b = (size_t)n_blocks * N * 32 * psz, tab_b = (size_t)N * N * csz,
                 co_b = (size_t)n_blocks * N * N * csz;
    char *dev = nullptr;
    HIPCK( hipMalloc( &dev, fe_b + fd_b + 2 * tab_b + co_b + n_blocks * sizeof( int ) + 1024 ) );
    char *d_fe = dev, *d_fd = d_fe + align_up( fe_b, 64 ), *d_mf = d_fd + align_up( fd_b, 64 ), *d_bias = d_mf + align_up( tab_b, 64 ),
         *d_co = d_bias + align_up( tab_b, 64 ), *d_nz = d_co + align_up( co_b, 64 );
    int rc = X264HIP_OK;
    if( hipMemcpy( d_fe, fenc, fe_b, hipMemcpyHostToDevice ) != hipSuccess || hipMemcpy( d_fd, fdec, fd_b, hipMemcpyHostToDevice ) != hipSuccess ||
        hipMemcpy( d_mf, mf, tab_b, hipMemcpyHostToDevice ) != hipSuccess || hipMemcpy( d_bias, bias, tab_b, hipMemcpyHostToDevice ) != hipSuccess )
        rc = X264HIP_EDEVICE;
    if( !rc )
    {
        const int grid = ( n_blocks + 63 ) / 64;
        if( ctx->p.bit_depth == 8 )
            dct_quant_kernel<uint8_t, int16_t, uint16_t><<<grid, 64, 0, ctx->stream>>>( is8x8, n_blocks, (const uint8_t *)d_fe, (const uint8_t *)d_fd,
                                                                                        (const uint16_t *)d_mf, (const uint16_t *)d_bias, (int16_t *)d_co, (int *)d_nz );
        else
            dct_quant_kernel<uint16_t, int32_t, uint32_t><<<grid, 64, 0, ctx->stream>>>( is8x8, n_blocks, (const uint16_t *)d_fe, (const uint16_t *)d_fd,
                                                                                         (const uint32_t *)d_mf, (const uint32_t *)d_bias, (int32_t *)d_co, (int *)d_nz );
        if( hipStreamSynchronize( ctx->stream ) != hipSuccess || hipMemcpy( coefs_out, d_co, co_b, hipMemcpyDeviceToHost ) != hipSuccess ||
            hipMemcpy( nz_out, d_nz, n_blocks * sizeof( int ), hipMemcpyDeviceToHost ) != hipSuccess )
            rc = X264HIP_EDEVICE;
    }
    (void)hipFree( dev );
    if( rc ) ctx->broken = 1;
    return rc;
}

// ---- the remaining per-macroblock vtable entries in batch form (vtable_blocks.h): host buffers staged through one device allocation ----
namespace {
struct Staged
{
    char *dev = nullptr;
    std::vector<size_t> off;
    size_t total = 0;
    size_t add( size_t bytes ) { off.push_back( total ); total += align_up( bytes ? bytes : 1, 256 ); return off.size() - 1; }
    char *at( size_t i ) const { return dev + off[i]; }
    ~Staged() { if( dev ) (void)hipFree( dev ); }
};
}

extern "C" int x264hip_dct_batch( x264hip_ctx *ctx, int kind, int n, const void *fenc, const void *fdec, void *coefs )
{
    if( !ctx || n <= 0 || !coefs || !vt_dct_coefs( kind ) || ( !vt_dct_in_place( kind ) && ( !fenc || !fdec ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int psz = ctx->psz, csz = ctx->p.bit_depth == 8 ? 2 : 4;
    const size_t fe_b = (size_t)n * 16 * VT_FENC_STRIDE * psz, fd_b = (size_t)n * 16 * VT_FDEC_STRIDE * psz, co_b = (size_t)n * vt_dct_coefs( kind ) * csz;
    Staged st;
    const size_t i_fe = st.add( fe_b ), i_fd = st.add( fd_b ), i_co = st.add( co_b );
    HIPCK( hipMalloc( &st.dev, st.total ) );
    if( vt_dct_in_place( kind ) )
        HIPCK( hipMemcpy( st.at( i_co ), coefs, co_b, hipMemcpyHostToDevice ) );
    else
    {
        HIPCK( hipMemcpy( st.at( i_fe ), fenc, fe_b, hipMemcpyHostToDevice ) );
        HIPCK( hipMemcpy( st.at( i_fd ), fdec, fd_b, hipMemcpyHostToDevice ) );
    }
    const int grid = ( n + 63 ) / 64;
    if( ctx->p.bit_depth == 8 )
        vt_dct_kernel<uint8_t, int16_t><<<grid, 64, 0, ctx->stream>>>( kind, n, (const uint8_t *)st.at( i_fe ), (const uint8_t *)st.at( i_fd ), (int16_t *)st.at( i_co ) );
    else
        vt_dct_kernel<uint16_t, int32_t><<<grid, 64, 0, ctx->stream>>>( kind, n, (const uint16_t *)st.at( i_fe ), (const uint16_t *)st.at( i_fd ), (int32_t *)st.at( i_co ) );
    HIPCK( hipGetLastError() );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    HIPCK( hipMemcpy( coefs, st.at( i_co ), co_b, hipMemcpyDeviceToHost ) );
    return X264HIP_OK;
}

extern "C" int x264hip_quant_batch( x264hip_ctx *ctx, int kind, int n, void *coefs, const void *mf, const void *bias, int mf_dc, int bias_dc, int *nz )
{
    const bool dc = kind == VT_QUANT_4X4_DC || kind == VT_QUANT_2X2_DC;
    if( !ctx || n <= 0 || !coefs || !nz || !vt_quant_coefs( kind ) || ( !dc && ( !mf || !bias ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int csz = ctx->p.bit_depth == 8 ? 2 : 4; // dctcoef and udctcoef have the same size (common/common.h:93-105)
    const int n_tab = kind == VT_QUANT_8X8 ? 64 : 16;
    const size_t co_b = (size_t)n * vt_quant_coefs( kind ) * csz, tab_b = (size_t)n_tab * csz;
    Staged st;
    const size_t i_co = st.add( co_b ), i_mf = st.add( tab_b ), i_bias = st.add( tab_b ), i_nz = st.add( (size_t)n * sizeof( int ) );
    HIPCK( hipMalloc( &st.dev, st.total ) );
    HIPCK( hipMemcpy( st.at( i_co ), coefs, co_b, hipMemcpyHostToDevice ) );
    if( !dc )
    {
        HIPCK( hipMemcpy( st.at( i_mf ), mf, tab_b, hipMemcpyHostToDevice ) );
        HIPCK( hipMemcpy( st.at( i_bias ), bias, tab_b, hipMemcpyHostToDevice ) );
    }
    const int grid = ( n + 63 ) / 64;
    if( ctx->p.bit_depth == 8 )
        vt_quant_kernel<int16_t, uint16_t><<<grid, 64, 0, ctx->stream>>>( kind, n, (int16_t *)st.at( i_co ), (const uint16_t *)st.at( i_mf ), (const uint16_t *)st.at( i_bias ),
                                                                          mf_dc, bias_dc, (int *)st.at( i_nz ) );
    else
        vt_quant_kernel<int32_t, uint32_t><<<grid, 64, 0, ctx->stream>>>( kind, n, (int32_t *)st.at( i_co ), (const uint32_t *)st.at( i_mf ), (const uint32_t *)st.at( i_bias ),
                                                                          mf_dc, bias_dc, (int *)st.at( i_nz ) );
    HIPCK( hipGetLastError() );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    HIPCK( hipMemcpy( coefs, st.at( i_co ), co_b, hipMemcpyDeviceToHost ) );
    HIPCK( hipMemcpy( nz, st.at( i_nz ), (size_t)n * sizeof( int ), hipMemcpyDeviceToHost ) );
    return X264HIP_OK;
}

extern "C" int x264hip_var2_batch( x264hip_ctx *ctx, int height, int n, const void *fenc, const void *fdec, int *var, int *ssd )
{
    if( !ctx || n <= 0 || !fenc || !fdec || !var || !ssd || ( height != 8 && height != 16 ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const size_t fe_b = (size_t)n * 16 * VT_FENC_STRIDE * ctx->psz, fd_b = (size_t)n * 16 * VT_FDEC_STRIDE * ctx->psz;
    Staged st;
    const size_t i_fe = st.add( fe_b ), i_fd = st.add( fd_b ), i_var = st.add( (size_t)n * sizeof( int ) ), i_ssd = st.add( (size_t)n * 2 * sizeof( int ) );
    HIPCK( hipMalloc( &st.dev, st.total ) );
    HIPCK( hipMemcpy( st.at( i_fe ), fenc, fe_b, hipMemcpyHostToDevice ) );
    HIPCK( hipMemcpy( st.at( i_fd ), fdec, fd_b, hipMemcpyHostToDevice ) );
    const int grid = ( n + 63 ) / 64;
    if( ctx->p.bit_depth == 8 )
        vt_var2_kernel<uint8_t><<<grid, 64, 0, ctx->stream>>>( height, n, (const uint8_t *)st.at( i_fe ), (const uint8_t *)st.at( i_fd ), (int *)st.at( i_var ), (int *)st.at( i_ssd ) );
    else
        vt_var2_kernel<uint16_t><<<grid, 64, 0, ctx->stream>>>( height, n, (const uint16_t *)st.at( i_fe ), (const uint16_t *)st.at( i_fd ), (int *)st.at( i_var ), (int *)st.at( i_ssd ) );
    HIPCK( hipGetLastError() );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    HIPCK( hipMemcpy( var, st.at( i_var ), (size_t)n * sizeof( int ), hipMemcpyDeviceToHost ) );
    HIPCK( hipMemcpy( ssd, st.at( i_ssd ), (size_t)n * 2 * sizeof( int ), hipMemcpyDeviceToHost ) );
    return X264HIP_OK;
}

static_assert( sizeof( x264hip_ads_call ) == sizeof( VtAdsCall ), "x264hip_ads_call mirrors VtAdsCall" );
extern "C" int x264hip_ads_batch( x264hip_ctx *ctx, int n, const x264hip_ads_call *calls, const uint16_t *sums, size_t n_sums, const uint16_t *cost_mvx, size_t n_cost,
                                  int16_t *mvs, size_t n_mvs, int *counts )
{
    if( !ctx || n <= 0 || !calls || !sums || !cost_mvx || !mvs || !counts ) return X264HIP_EINVAL;
    for( int i = 0; i < n; i++ )
    {
        const x264hip_ads_call &c = calls[i];
        const long long span = ( c.n_dc == 4 ? c.delta + 8 : c.n_dc == 2 ? c.delta : 0 ) + c.width;
        if( ( c.n_dc != 1 && c.n_dc != 2 && c.n_dc != 4 ) || c.width < 0 || c.delta < 0 || c.sums_off < 0 || c.cost_off < 0 || c.mvs_off < 0 ||
            c.sums_off + span > (long long)n_sums || c.cost_off + c.width > (long long)n_cost || c.mvs_off + c.width > (long long)n_mvs )
            return X264HIP_EINVAL;
    }
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    Staged st;
    const size_t i_calls = st.add( (size_t)n * sizeof( VtAdsCall ) ), i_sums = st.add( n_sums * 2 ), i_cost = st.add( n_cost * 2 ), i_mvs = st.add( n_mvs * 2 ),
                 i_cnt = st.add( (size_t)n * sizeof( int ) );
    HIPCK( hipMalloc( &st.dev, st.total ) );
    HIPCK( hipMemcpy( st.at( i_calls ), calls, (size_t)n * sizeof( VtAdsCall ), hipMemcpyHostToDevice ) );
    HIPCK( hipMemcpy( st.at( i_sums ), sums, n_sums * 2, hipMemcpyHostToDevice ) );
    HIPCK( hipMemcpy( st.at( i_cost ), cost_mvx, n_cost * 2, hipMemcpyHostToDevice ) );
    HIPCK( hipMemsetAsync( st.at( i_mvs ), 0, n_mvs * 2, ctx->stream ) ); // (on the kernel's stream: hipMemset does not wait for itself)
    vt_ads_kernel<<<n, 64, 0, ctx->stream>>>( n, (const VtAdsCall *)st.at( i_calls ), (const uint16_t *)st.at( i_sums ), (const uint16_t *)st.at( i_cost ),
                                             (int16_t *)st.at( i_mvs ), (int *)st.at( i_cnt ) );
    HIPCK( hipGetLastError() );
    HIPCK( hipStreamSynchronize( ctx->stream ) );
    HIPCK( hipMemcpy( mvs, st.at( i_mvs ), n_mvs * 2, hipMemcpyDeviceToHost ) );
    HIPCK( hipMemcpy( counts, st.at( i_cnt ), (size_t)n * sizeof( int ), hipMemcpyDeviceToHost ) );
    return X264HIP_OK;
}

// ---- function-table members with the reference's exact signatures (common/mc.h:292,306-307,326-327,333-337; common/pixel.h:78-100;
// common/dct.h:29-59; common/quant.h:30-45) ---------------------------------------------------------------------------------------------
// The table functions of the reference carry no context argument.  The filler therefore binds them through a process-wide registry:
// one context per bit depth (the reference's tables are bit-depth templated too: x264_8_*, x264_10_*), plus, for the one member that
// receives the encoder handle (mbtree_propagate_list), contexts registered per handle (x264hip_mc_bind_handle) so that two encoders
// with different picture sizes can coexist.  The members may be called from any thread (the reference calls hpel_filter from its frame
// threads, common/mc.c:704-784): every call takes the registry lock, so calls are serialised, the bound context cannot be closed under a
// running call, and all of them share the context's staging buffer.  The pointers are HOST pointers like the ones the encoder passes,
// staged through device memory on every call (correct and signature-compatible; a caller that cares about speed keeps its planes on the
// device and uses the x264hip_* entries that take device pointers).  The functions return like the originals: a device failure latches
// the context (x264hip_synchronize and every other call then return X264HIP_EDEVICE), which is how slicetype-cl.c:44-56 reports errors.
#include <mutex>
#include <map>
namespace {
std::mutex g_vt_mutex;
x264hip_ctx *g_vt_ctx[2] = { nullptr, nullptr };          // [bit depth 8 / 10]
std::map<const void *, x264hip_ctx *> g_vt_by_handle;     // x264_t * -> context

struct VtFail {};
#define VTCK( call ) do { if( ( call ) != hipSuccess ) throw VtFail(); } while( 0 )
#define VTRC( call ) do { if( ( call ) != X264HIP_OK ) throw VtFail(); } while( 0 )

// staging memory of a table call: carved out of the context's grow-only pool (calls are serialised by the registry lock and every call
// drains the stream before it returns, so one pool serves them all)
struct VtStage
{
    x264hip_ctx *ctx;
    std::vector<size_t> off;
    size_t total = 0;
    explicit VtStage( x264hip_ctx *c ) : ctx( c ) {}
    size_t add( size_t bytes ) { off.push_back( total ); total += align_up( bytes ? bytes : 1, 256 ); return off.size() - 1; }
    void commit()
    {
        if( total > ctx->vt_pool_bytes )
        {
            if( ctx->vt_pool ) VTCK( hipFree( ctx->vt_pool ) );
            ctx->vt_pool = nullptr; ctx->vt_pool_bytes = 0;
            VTCK( hipMalloc( &ctx->vt_pool, total + ( total >> 1 ) ) );
            ctx->vt_pool_bytes = total + ( total >> 1 );
        }
    }
    char *at( size_t i ) const { return ctx->vt_pool + off[i]; }
};

template <typename F>
void vt_guard_ctx( x264hip_ctx *ctx, F body )
{
    if( !ctx || ctx->broken ) return;
    if( hipSetDevice( ctx->device ) != hipSuccess ) { ctx->broken = 1; return; }
    try { body( ctx ); }
    catch( const VtFail & ) { ctx->broken = 1; }
    catch( ... ) { ctx->broken = 1; }
}
template <int D, typename F>
void vt_guard( F body )
{
    std::lock_guard<std::mutex> lock( g_vt_mutex );
    vt_guard_ctx( g_vt_ctx[D], body );
}

template <int D>
void vt_plane_copy( void *dst, intptr_t i_dst, void *src, intptr_t i_src, int w, int h )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        if( w <= 0 || h <= 0 ) return;
        const size_t row = (size_t)w * ctx->psz, pitch = align_up( row, 16 );
        VtStage st( ctx );
        const size_t a = st.add( pitch * h ), b = st.add( pitch * h );
        st.commit();
        VTCK( hipMemcpy2D( st.at( a ), pitch, src, (size_t)i_src * ctx->psz, row, h, hipMemcpyHostToDevice ) );
        VTRC( x264hip_device_copy( ctx, st.at( b ), st.at( a ), pitch * h ) );
        VTCK( hipStreamSynchronize( ctx->stream ) );
        VTCK( hipMemcpy2D( dst, (size_t)i_dst * ctx->psz, st.at( b ), pitch, row, h, hipMemcpyDeviceToHost ) );
    } );
}

template <int D>
void vt_hpel_filter( void *dsth, void *dstv, void *dstc, void *src, intptr_t stride, int width, int height, int16_t *buf )
{
    (void)buf; // the C version's scratch row; the kernel keeps its intermediate sums in LDS
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        if( width <= 0 || height <= 0 ) return;
        const int psz = ctx->psz;
        // the four planes get the caller's geometry on the device: rows -2 .. height+2 of `stride` samples, 8 samples of slack in front
        const size_t rows = height + 5, plane_b = ( rows * stride + 16 ) * psz;
        VtStage st( ctx );
        const size_t i_s = st.add( plane_b ), i_h = st.add( plane_b ), i_v = st.add( plane_b ), i_c = st.add( plane_b );
        st.commit();
        auto org = [&]( size_t i ) { return st.at( i ) + ( (size_t)2 * stride + 8 ) * psz; }; // sample (0,0)
        // what the C version reads: columns -2 .. width+2 of rows -2 .. height+2 (mc.c:172-196)
        VTCK( hipMemcpy2D( org( i_s ) - ( 2 * stride + 2 ) * psz, (size_t)stride * psz, (const char *)src - ( 2 * stride + 2 ) * psz, (size_t)stride * psz,
                           (size_t)( width + 5 ) * psz, rows, hipMemcpyHostToDevice ) );
        VTRC( x264hip_hpel_filter( ctx, org( i_h ), org( i_v ), org( i_c ), org( i_s ), stride, width, height ) );
        VTCK( hipStreamSynchronize( ctx->stream ) );
        VTCK( hipMemcpy2D( dsth, (size_t)stride * psz, org( i_h ), (size_t)stride * psz, (size_t)width * psz, height, hipMemcpyDeviceToHost ) );
        VTCK( hipMemcpy2D( dstc, (size_t)stride * psz, org( i_c ), (size_t)stride * psz, (size_t)width * psz, height, hipMemcpyDeviceToHost ) );
        // dstv also receives columns -2, -1 and width .. width+2 (the C version filters them for the centre plane and stores them)
        VTCK( hipMemcpy2D( (char *)dstv - 2 * psz, (size_t)stride * psz, org( i_v ) - 2 * psz, (size_t)stride * psz, (size_t)( width + 5 ) * psz, height, hipMemcpyDeviceToHost ) );
    } );
}

template <int D>
void vt_frame_init_lowres_core( void *src0, void *dst0, void *dsth, void *dstv, void *dstc, intptr_t src_stride, intptr_t dst_stride, int width, int height )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        if( width <= 0 || height <= 0 ) return;
        const int psz = ctx->psz;
        // reads rows 0 .. 2*height and columns 0 .. 2*width of the source (mc.c:484-507)
        const size_t s_rows = 2 * (size_t)height + 1, s_row_b = ( 2 * (size_t)width + 1 ) * psz, s_pitch = align_up( s_row_b, 16 );
        const size_t d_row_b = (size_t)width * psz, d_pitch = align_up( d_row_b, 16 );
        VtStage st( ctx );
        const size_t i_s = st.add( s_pitch * s_rows ), i0 = st.add( d_pitch * height ), i1 = st.add( d_pitch * height ), i2 = st.add( d_pitch * height ),
                     i3 = st.add( d_pitch * height );
        st.commit();
        VTCK( hipMemcpy2D( st.at( i_s ), s_pitch, src0, (size_t)src_stride * psz, s_row_b, s_rows, hipMemcpyHostToDevice ) );
        VTRC( x264hip_frame_init_lowres_core( ctx, st.at( i_s ), st.at( i0 ), st.at( i1 ), st.at( i2 ), st.at( i3 ), (intptr_t)( s_pitch / psz ), (intptr_t)( d_pitch / psz ),
                                              width, height ) );
        VTCK( hipStreamSynchronize( ctx->stream ) );
        void *dst[4] = { dst0, dsth, dstv, dstc };
        const size_t idx[4] = { i0, i1, i2, i3 };
        for( int k = 0; k < 4; k++ )
            VTCK( hipMemcpy2D( dst[k], (size_t)dst_stride * psz, st.at( idx[k] ), d_pitch, d_row_b, height, hipMemcpyDeviceToHost ) );
    } );
}

template <int D>
void vt_mbtree_propagate_cost( int16_t *dst, uint16_t *propagate_in, uint16_t *intra_costs, uint16_t *inter_costs, uint16_t *inv_qscales, float *fps_factor, int len )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        if( len <= 0 ) return;
        const size_t b = (size_t)len * 2;
        VtStage st( ctx );
        const size_t i_d = st.add( b ), i_p = st.add( b ), i_i = st.add( b ), i_e = st.add( b ), i_q = st.add( b );
        st.commit();
        VTCK( hipMemcpy( st.at( i_p ), propagate_in, b, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_i ), intra_costs, b, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_e ), inter_costs, b, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_q ), inv_qscales, b, hipMemcpyHostToDevice ) );
        mbt_cost_row_kernel<<<( len + 255 ) / 256, 256, 0, ctx->stream>>>( (int16_t *)st.at( i_d ), (const uint16_t *)st.at( i_p ), (const uint16_t *)st.at( i_i ),
                                                                         (const uint16_t *)st.at( i_e ), (const uint16_t *)st.at( i_q ), *fps_factor, len );
        VTCK( hipGetLastError() );
        VTCK( hipStreamSynchronize( ctx->stream ) );
        VTCK( hipMemcpy( dst, st.at( i_d ), b, hipMemcpyDeviceToHost ) );
    } );
}

template <int D>
void vt_mbtree_propagate_list( void *h, uint16_t *ref_costs, int16_t ( *mvs )[2], int16_t *propagate_amount, uint16_t *lowres_costs, int bipred_weight, int mb_y, int len,
                               int list )
{
    // x264_t *: the C version reads the macroblock geometry from it; here that is the context registered for this handle
    // (x264hip_mc_bind_handle), or the one bound for the bit depth
    std::lock_guard<std::mutex> lock( g_vt_mutex );
    auto it = g_vt_by_handle.find( h );
    vt_guard_ctx( it != g_vt_by_handle.end() ? it->second : g_vt_ctx[D], [&]( x264hip_ctx *ctx ) {
        const int W = ctx->P.mb_w, H = ctx->P.mb_h, n_mb = W * H;
        if( len <= 0 || len > W || mb_y < 0 || mb_y >= H ) throw VtFail();
        VtStage st( ctx );
        const size_t i_r16 = st.add( (size_t)n_mb * 2 ), i_r32 = st.add( (size_t)n_mb * 4 ), i_mv = st.add( (size_t)len * 4 ), i_pa = st.add( (size_t)len * 2 ),
                     i_lc = st.add( (size_t)len * 2 ), i_o16 = st.add( (size_t)n_mb * 2 );
        st.commit();
        VTCK( hipMemcpy( st.at( i_r16 ), ref_costs, (size_t)n_mb * 2, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_mv ), mvs, (size_t)len * 4, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_pa ), propagate_amount, (size_t)len * 2, hipMemcpyHostToDevice ) );
        VTCK( hipMemcpy( st.at( i_lc ), lowres_costs, (size_t)len * 2, hipMemcpyHostToDevice ) );
        widen_u16_kernel<<<( n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( (int *)st.at( i_r32 ), (const uint16_t *)st.at( i_r16 ), n_mb );
        mbt_list_row_kernel<<<( len + 255 ) / 256, 256, 0, ctx->stream>>>( (int *)st.at( i_r32 ), (const int16_t *)st.at( i_mv ), (const int16_t *)st.at( i_pa ),
                                                                         (const uint16_t *)st.at( i_lc ), bipred_weight, mb_y, len, list, W, H );
        // only the entries this row added to are saturated (MC_CLIP_ADD, mc.c:527-598): everything else keeps the caller's value
        narrow_changed_kernel<<<( n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( (uint16_t *)st.at( i_o16 ), (const int *)st.at( i_r32 ), (const uint16_t *)st.at( i_r16 ), n_mb );
        VTCK( hipGetLastError() );
        VTCK( hipStreamSynchronize( ctx->stream ) );
        VTCK( hipMemcpy( ref_costs, st.at( i_o16 ), (size_t)n_mb * 2, hipMemcpyDeviceToHost ) );
    } );
}

// ---- x264_dct_function_t / x264_quant_function_t / x264_pixel_function_t members: ONE call staged through the batch entries above.  The
// encoder's pointers point into its macroblock buffers (fenc stride 16, fdec stride 32): exactly the rows a member reads are copied.
template <int D, int KIND, int BW, int BH>
void vt_sub_dct( void *dct, void *pix1, void *pix2 )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        const int psz = ctx->psz;
        std::vector<char> fe( (size_t)16 * VT_FENC_STRIDE * psz, 0 ), fd( (size_t)16 * VT_FDEC_STRIDE * psz, 0 );
        for( int y = 0; y < BH; y++ )
        {
            memcpy( &fe[(size_t)y * VT_FENC_STRIDE * psz], (const char *)pix1 + (size_t)y * VT_FENC_STRIDE * psz, (size_t)BW * psz );
            memcpy( &fd[(size_t)y * VT_FDEC_STRIDE * psz], (const char *)pix2 + (size_t)y * VT_FDEC_STRIDE * psz, (size_t)BW * psz );
        }
        VTRC( x264hip_dct_batch( ctx, KIND, 1, fe.data(), fd.data(), dct ) );
    } );
}
template <int D>
void vt_dct4x4dc( void *d )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) { VTRC( x264hip_dct_batch( ctx, 7, 1, nullptr, nullptr, d ) ); } );
}
template <int D, typename C>
void vt_dct2x4dc( void *dct, void *dct4x4 )
{
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        C *out = (C *)dct, ( *blk )[16] = (C( * )[16])dct4x4;
        for( int i = 0; i < 8; i++ ) out[i] = blk[i][0];
        VTRC( x264hip_dct_batch( ctx, 8, 1, nullptr, nullptr, out ) );
        for( int i = 0; i < 8; i++ ) blk[i][0] = 0; // the DC terms move to dct[] (dct.c:109-143)
    } );
}
template <int D, int KIND>
int vt_quant( void *dct, void *mf, void *bias )
{
    int nz = 0;
    vt_guard<D>( [&]( x264hip_ctx *ctx ) { VTRC( x264hip_quant_batch( ctx, KIND, 1, dct, mf, bias, 0, 0, &nz ) ); } );
    return nz;
}
template <int D, int KIND>
int vt_quant_dc( void *dct, int mf, int bias )
{
    int nz = 0;
    vt_guard<D>( [&]( x264hip_ctx *ctx ) { VTRC( x264hip_quant_batch( ctx, KIND, 1, dct, nullptr, nullptr, mf, bias, &nz ) ); } );
    return nz;
}

static const int vt_size_w[7] = { 16, 16, 8, 8, 8, 4, 4 }, vt_size_h[7] = { 16, 8, 16, 8, 4, 8, 4 }; // PIXEL_16x16 .. PIXEL_4x4 (common/pixel.h:37-59)
// one block pair on the device: both blocks in the top-left corner of a 16x16 plane pair (what the batch entries index)
template <int D>
uint64_t vt_block_metric( int metric /* -1 sad, -2 satd, else X264HIP_METRIC_* */, int size_idx, void *pix1, intptr_t s1, void *pix2, intptr_t s2 )
{
    uint64_t result = 0;
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        const int psz = ctx->psz, w = vt_size_w[size_idx], h = vt_size_h[size_idx];
        VtStage st( ctx );
        const size_t i_a = st.add( (size_t)16 * 16 * psz ), i_b = st.add( (size_t)16 * 16 * psz ), i_mv = st.add( 256 * 4 ), i_out = st.add( 256 * 8 );
        st.commit();
        VTCK( hipMemsetAsync( st.at( i_a ), 0, st.total, ctx->stream ) );
        VTCK( hipMemcpy2DAsync( st.at( i_a ), (size_t)16 * psz, pix1, (size_t)s1 * psz, (size_t)w * psz, h, hipMemcpyHostToDevice, ctx->stream ) );
        if( pix2 )
            VTCK( hipMemcpy2DAsync( st.at( i_b ), (size_t)16 * psz, pix2, (size_t)s2 * psz, (size_t)w * psz, h, hipMemcpyHostToDevice, ctx->stream ) );
        if( metric < 0 )
        {
            VTRC( x264hip_pixel_cmp_batch( ctx, metric == -2, size_idx, st.at( i_a ), st.at( i_b ), 16, 16 / w, 16 / h, (const int16_t *)st.at( i_mv ), (int *)st.at( i_out ) ) );
            int v = 0;
            VTCK( hipMemcpyAsync( &v, st.at( i_out ), 4, hipMemcpyDeviceToHost, ctx->stream ) );
            VTCK( hipStreamSynchronize( ctx->stream ) );
            result = (uint64_t)(unsigned)v;
        }
        else
        {
            VTRC( x264hip_pixel_metric_batch( ctx, metric, size_idx, st.at( i_a ), pix2 ? st.at( i_b ) : nullptr, 16, 1, 1, (uint64_t *)st.at( i_out ) ) );
            VTCK( hipMemcpyAsync( &result, st.at( i_out ), 8, hipMemcpyDeviceToHost, ctx->stream ) );
            VTCK( hipStreamSynchronize( ctx->stream ) );
        }
    } );
    return result;
}
template <int D, int METRIC, int SIZE>
int vt_cmp( void *pix1, intptr_t s1, void *pix2, intptr_t s2 ) { return (int)vt_block_metric<D>( METRIC, SIZE, pix1, s1, pix2, s2 ); }
template <int D, int METRIC, int SIZE>
uint64_t vt_one( void *pix, intptr_t stride ) { return vt_block_metric<D>( METRIC, SIZE, pix, stride, nullptr, 0 ); }

// several candidates against one source block: one staged call per candidate (fenc has FENC_STRIDE)
template <int D, int METRIC, int SIZE>
void vt_x3( void *fenc, void *p0, void *p1, void *p2, intptr_t stride, int scores[3] )
{
    void *p[3] = { p0, p1, p2 };
    for( int k = 0; k < 3; k++ )
        scores[k] = (int)vt_block_metric<D>( METRIC, SIZE, fenc, VT_FENC_STRIDE, p[k], stride );
}
template <int D, int METRIC, int SIZE>
void vt_x4( void *fenc, void *p0, void *p1, void *p2, void *p3, intptr_t stride, int scores[4] )
{
    void *p[4] = { p0, p1, p2, p3 };
    for( int k = 0; k < 4; k++ )
        scores[k] = (int)vt_block_metric<D>( METRIC, SIZE, fenc, VT_FENC_STRIDE, p[k], stride );
}
template <int D>
int vt_vsad( void *pix, intptr_t stride, int height )
{
    if( height != 16 && height != 8 ) return 0; // the reference calls it with the rows of a macroblock (pair) only
    return (int)vt_block_metric<D>( X264HIP_METRIC_VSAD, height == 16 ? 0 : 1, pix, stride, nullptr, 0 );
}
template <int D>
int vt_asd8( void *pix1, intptr_t s1, void *pix2, intptr_t s2, int height )
{
    if( height != 16 && height != 8 ) return 0;
    return (int)vt_block_metric<D>( X264HIP_METRIC_ASD8, height == 16 ? 2 : 3, pix1, s1, pix2, s2 );
}
template <int D, int H>
int vt_var2( void *fenc, void *fdec, int ssd[2] )
{
    int var = 0;
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        const int psz = ctx->psz;
        std::vector<char> fe( (size_t)16 * VT_FENC_STRIDE * psz, 0 ), fd( (size_t)16 * VT_FDEC_STRIDE * psz, 0 );
        for( int y = 0; y < H; y++ )
        {
            // U at column 0, V at column stride / 2 of the encoder's chroma buffers (common/macroblock.c: p_fenc[1] / p_fenc[2])
            memcpy( &fe[(size_t)y * VT_FENC_STRIDE * psz], (const char *)fenc + (size_t)y * VT_FENC_STRIDE * psz, (size_t)VT_FENC_STRIDE * psz );
            memcpy( &fd[(size_t)y * VT_FDEC_STRIDE * psz], (const char *)fdec + (size_t)y * VT_FDEC_STRIDE * psz, (size_t)( VT_FDEC_STRIDE / 2 + 8 ) * psz );
        }
        VTRC( x264hip_var2_batch( ctx, H, 1, fe.data(), fd.data(), &var, ssd ) );
    } );
    return var;
}
template <int D, int NDC>
int vt_ads( int enc_dc[4], uint16_t *sums, int delta, uint16_t *cost_mvx, int16_t *mvs, int width, int thresh )
{
    int count = 0;
    vt_guard<D>( [&]( x264hip_ctx *ctx ) {
        if( width <= 0 ) return;
        x264hip_ads_call c;
        memset( &c, 0, sizeof( c ) );
        c.n_dc = NDC; c.delta = delta; c.width = width; c.thresh = thresh;
        for( int k = 0; k < NDC; k++ ) c.enc_dc[k] = enc_dc[k];
        const size_t n_sums = (size_t)( NDC == 4 ? delta + 8 : NDC == 2 ? delta : 0 ) + width;
        std::vector<int16_t> out( (size_t)width );
        VTRC( x264hip_ads_batch( ctx, 1, &c, sums, n_sums, cost_mvx, (size_t)width, out.data(), (size_t)width, &count ) );
        memcpy( mvs, out.data(), (size_t)count * sizeof( int16_t ) ); // the reference writes the surviving candidates only
    } );
    return count;
}

template <int D>
void fill_mc( x264hip_mc_functions *pf )
{
    pf->plane_copy = vt_plane_copy<D>;
    pf->hpel_filter = vt_hpel_filter<D>;
    pf->frame_init_lowres_core = vt_frame_init_lowres_core<D>;
    pf->mbtree_propagate_cost = vt_mbtree_propagate_cost<D>;
    pf->mbtree_propagate_list = vt_mbtree_propagate_list<D>;
}
template <int D, typename C>
void fill_dct( x264hip_dct_functions *pf )
{
    pf->sub4x4_dct = vt_sub_dct<D, 0, 4, 4>;       pf->sub8x8_dct = vt_sub_dct<D, 1, 8, 8>;       pf->sub16x16_dct = vt_sub_dct<D, 2, 16, 16>;
    pf->sub8x8_dct8 = vt_sub_dct<D, 3, 8, 8>;      pf->sub16x16_dct8 = vt_sub_dct<D, 4, 16, 16>;
    pf->sub8x8_dct_dc = vt_sub_dct<D, 5, 8, 8>;    pf->sub8x16_dct_dc = vt_sub_dct<D, 6, 8, 16>;
    pf->dct4x4dc = vt_dct4x4dc<D>;                 pf->dct2x4dc = vt_dct2x4dc<D, C>;
}
template <int D>
void fill_quant( x264hip_quant_functions *pf )
{
    pf->quant_8x8 = vt_quant<D, 1>; pf->quant_4x4 = vt_quant<D, 0>; pf->quant_4x4x4 = vt_quant<D, 2>;
    pf->quant_4x4_dc = vt_quant_dc<D, 3>; pf->quant_2x2_dc = vt_quant_dc<D, 4>;
}
template <int D>
void fill_pixel( x264hip_pixel_functions *pf, bool satd, bool satd_fpel )
{
    memset( pf, 0, sizeof( *pf ) );
#define VT_SIZE( S ) pf->sad[S] = vt_cmp<D, -1, S>; pf->satd[S] = vt_cmp<D, -2, S>; pf->ssd[S] = vt_cmp<D, X264HIP_METRIC_SSD, S>;
    VT_SIZE( 0 ) VT_SIZE( 1 ) VT_SIZE( 2 ) VT_SIZE( 3 ) VT_SIZE( 4 ) VT_SIZE( 5 ) VT_SIZE( 6 )
#undef VT_SIZE
    pf->sa8d[0] = vt_cmp<D, X264HIP_METRIC_SA8D, 0>; pf->sa8d[3] = vt_cmp<D, X264HIP_METRIC_SA8D, 3>;
    pf->var[0] = vt_one<D, X264HIP_METRIC_VAR, 0>; pf->var[2] = vt_one<D, X264HIP_METRIC_VAR, 2>; pf->var[3] = vt_one<D, X264HIP_METRIC_VAR, 3>;
    pf->hadamard_ac[0] = vt_one<D, X264HIP_METRIC_HADAMARD_AC, 0>; pf->hadamard_ac[1] = vt_one<D, X264HIP_METRIC_HADAMARD_AC, 1>;
    pf->hadamard_ac[2] = vt_one<D, X264HIP_METRIC_HADAMARD_AC, 2>; pf->hadamard_ac[3] = vt_one<D, X264HIP_METRIC_HADAMARD_AC, 3>;
#define VT_MULTI( S ) pf->sad_x3[S] = vt_x3<D, -1, S>; pf->sad_x4[S] = vt_x4<D, -1, S>; pf->satd_x3[S] = vt_x3<D, -2, S>; pf->satd_x4[S] = vt_x4<D, -2, S>;
    VT_MULTI( 0 ) VT_MULTI( 1 ) VT_MULTI( 2 ) VT_MULTI( 3 ) VT_MULTI( 4 ) VT_MULTI( 5 ) VT_MULTI( 6 )
#undef VT_MULTI
    pf->vsad = vt_vsad<D>; pf->asd8 = vt_asd8<D>;
    pf->var2[2] = vt_var2<D, 16>; pf->var2[3] = vt_var2<D, 8>;                       // PIXEL_8x16, PIXEL_8x8 (pixel.c:880-881)
    pf->ads[0] = vt_ads<D, 4>; pf->ads[1] = vt_ads<D, 2>; pf->ads[3] = vt_ads<D, 1>; // PIXEL_16x16, PIXEL_16x8, PIXEL_8x8 (pixel.c:883-885)
    // mbcmp_init (encoder/encoder.c:1409-1427): satd for sub-pel refinement and mode decision from subme 2 on, for the full-pel search with
    // --me tesa on top of that
    for( int i = 0; i < 8; i++ )
    {
        pf->sad_aligned[i] = pf->sad[i];
        pf->mbcmp[i] = satd ? pf->satd[i] : pf->sad_aligned[i];
        pf->mbcmp_unaligned[i] = satd ? pf->satd[i] : pf->sad[i];
        pf->fpelcmp[i] = satd_fpel ? pf->satd[i] : pf->sad[i];
    }
    for( int i = 0; i < 7; i++ )
    {
        pf->fpelcmp_x3[i] = satd_fpel ? pf->satd_x3[i] : pf->sad_x3[i];
        pf->fpelcmp_x4[i] = satd_fpel ? pf->satd_x4[i] : pf->sad_x4[i];
    }
}
int vt_bind( x264hip_ctx *ctx )
{
    if( !ctx ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    std::lock_guard<std::mutex> lock( g_vt_mutex );
    g_vt_ctx[ctx->p.bit_depth == 8 ? 0 : 1] = ctx;
    return X264HIP_OK;
}
} // namespace

extern "C" int x264hip_mc_fill( x264hip_ctx *ctx, x264hip_mc_functions *pf )
{
    if( !pf ) return X264HIP_EINVAL;
    int rc = vt_bind( ctx );
    if( rc ) return rc;
    if( ctx->p.bit_depth == 8 ) fill_mc<0>( pf ); else fill_mc<1>( pf );
    return X264HIP_OK;
}
extern "C" int x264hip_dct_fill( x264hip_ctx *ctx, x264hip_dct_functions *pf )
{
    if( !pf ) return X264HIP_EINVAL;
    int rc = vt_bind( ctx );
    if( rc ) return rc;
    if( ctx->p.bit_depth == 8 ) fill_dct<0, int16_t>( pf ); else fill_dct<1, int32_t>( pf );
    return X264HIP_OK;
}
extern "C" int x264hip_quant_fill( x264hip_ctx *ctx, x264hip_quant_functions *pf )
{
    if( !pf ) return X264HIP_EINVAL;
    int rc = vt_bind( ctx );
    if( rc ) return rc;
    if( ctx->p.bit_depth == 8 ) fill_quant<0>( pf ); else fill_quant<1>( pf );
    return X264HIP_OK;
}
extern "C" int x264hip_pixel_fill( x264hip_ctx *ctx, x264hip_pixel_functions *pf )
{
    if( !pf ) return X264HIP_EINVAL;
    int rc = vt_bind( ctx );
    if( rc ) return rc;
    // (the context's parameters say which metric the tables select: x264hip_params mbcmp_satd / fpelcmp_satd)
    if( ctx->p.bit_depth == 8 ) fill_pixel<0>( pf, ctx->p.mbcmp_satd != 0, ctx->p.fpelcmp_satd != 0 ); else fill_pixel<1>( pf, ctx->p.mbcmp_satd != 0, ctx->p.fpelcmp_satd != 0 );
    return X264HIP_OK;
}
extern "C" int x264hip_mc_bind_handle( x264hip_ctx *ctx, const void *encoder_handle )
{
    if( !ctx || !encoder_handle ) return X264HIP_EINVAL;
    std::lock_guard<std::mutex> lock( g_vt_mutex );
    g_vt_by_handle[encoder_handle] = ctx;
    return X264HIP_OK;
}
extern "C" void x264hip_mc_unbind( x264hip_ctx *ctx )
{
    // (waits for a table call that is running on this context: they hold the lock)
    std::lock_guard<std::mutex> lock( g_vt_mutex );
    for( int d = 0; d < 2; d++ )
        if( g_vt_ctx[d] == ctx ) g_vt_ctx[d] = nullptr;
    for( auto it = g_vt_by_handle.begin(); it != g_vt_by_handle.end(); )
        it = it->second == ctx ? g_vt_by_handle.erase( it ) : std::next( it );
}

// ---- one lookahead window sharded over several GPUs: the unweighted motion searches of a frame run on the rank that
// owns it, the finished fields travel to the deciding rank (x264_amd/shard.py drives the exchange with torch.distributed / RCCL) --------
__global__ __launch_bounds__( 256 ) void export_field_kernel( const unsigned long long *__restrict__ mvq, const int *__restrict__ costs, int2 *__restrict__ dst, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n )
        dst[i] = make_int2( (int)(unsigned)mvq[i], costs[i] );
}
__global__ __launch_bounds__( 256 ) void import_field_kernel( unsigned long long *__restrict__ mvq, int *__restrict__ costs, const int2 *__restrict__ src, unsigned tag, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n )
    {
        const int2 v = src[i];
        mvq[i] = ( (unsigned long long)tag << 32 ) | (unsigned)v.x;
        costs[i] = v.y;
    }
}

extern "C" int x264hip_search_fields( x264hip_ctx *ctx, int n, const int *slot_b, const int *slot_ref, const int *list, const int *dist_minus1 )
{
    if( !ctx || n < 0 || ( n && ( !slot_b || !slot_ref || !list || !dist_minus1 ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    std::vector<SearchReq> reqs;
    const WtD none = { 0, 1, 0, 0 };
    for( int i = 0; i < n; i++ )
    {
        if( !slot_ok( ctx, slot_b[i] ) || !slot_ok( ctx, slot_ref[i] ) || list[i] < 0 || list[i] > 1 || dist_minus1[i] < 0 || dist_minus1[i] > ctx->p.bframes )
            return X264HIP_EINVAL;
        FrameSlot &b = ctx->slots[slot_b[i]];
        if( !b.in_use || !ctx->slots[slot_ref[i]].in_use ) return X264HIP_ESTATE;
        if( b.field_ready[list[i]][dist_minus1[i]] || b.field_prefetched[list[i]][dist_minus1[i]] ) continue;
        b.field_prefetched[list[i]][dist_minus1[i]] = 1;
        reqs.push_back( SearchReq{ slot_b[i], slot_ref[i], list[i], dist_minus1[i], none } );
    }
    for( size_t o = 0; o < reqs.size(); o += ctx->desc_cap )
    {
        std::vector<SearchReq> part( reqs.begin() + o, reqs.begin() + std::min( reqs.size(), o + (size_t)ctx->desc_cap ) );
        int r = launch_searches( ctx, part );
        if( r ) return r;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_export_field( x264hip_ctx *ctx, int slot, int list, int dist_minus1, void *dst_dev )
{
    if( !ctx || !slot_ok( ctx, slot ) || list < 0 || list > 1 || dist_minus1 < 0 || dist_minus1 > ctx->p.bframes || !dst_dev ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    if( ( !s.field_ready[list][dist_minus1] && !s.field_prefetched[list][dist_minus1] ) || s.field_remote[list][dist_minus1] ) return X264HIP_ESTATE;
    export_field_kernel<<<( ctx->n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( s.mvq[list][dist_minus1], s.mvcost[list][dist_minus1], (int2 *)dst_dev, ctx->n_mb );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

extern "C" int x264hip_import_field( x264hip_ctx *ctx, int slot, int list, int dist_minus1, const void *src_dev )
{
    if( !ctx || !slot_ok( ctx, slot ) || list < 0 || list > 1 || dist_minus1 < 0 || dist_minus1 > ctx->p.bframes || !src_dev ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &s = ctx->slots[slot];
    if( !s.in_use ) return X264HIP_ESTATE;
    unsigned tag;
    if( s.field_remote[list][dist_minus1] )
        tag = s.field_tag[list][dist_minus1]; // known by tag so far (x264hip_fields_remote): the data arrives, the identity stays
    else if( s.field_ready[list][dist_minus1] || s.field_prefetched[list][dist_minus1] )
        return X264HIP_OK; // this context already has the field (searched here on demand): identical by construction, keep it
    else
    {
        tag = ctx->tag_serial++;
        if( !ctx->tag_serial ) ctx->tag_serial = 1;
    }
    s.field_remote[list][dist_minus1] = 0;
    import_field_kernel<<<( ctx->n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( s.mvq[list][dist_minus1], s.mvcost[list][dist_minus1], (const int2 *)src_dev, tag, ctx->n_mb );
    HIPCK( hipGetLastError() );
    s.field_tag[list][dist_minus1] = tag;
    s.field_prefetched[list][dist_minus1] = 1;
    return X264HIP_OK;
}

// ---- cells on the owner rank, summaries to the deciding rank (include/x264hip.h, "one lookahead window over several GPUs") -------------
extern "C" int x264hip_stream_handle( x264hip_ctx *ctx, void **hip_stream )
{
    if( !ctx || !hip_stream ) return X264HIP_EINVAL;
    *hip_stream = (void *)ctx->stream;
    return X264HIP_OK;
}

extern "C" int x264hip_cell_classes( x264hip_ctx *ctx, unsigned char *cell_class )
{
    if( !ctx || !cell_class ) return X264HIP_EINVAL;
    static const bool no_learn = getenv( "X264HIP_NO_CLASS_LEARNING" ) != nullptr;
    const bool learned = !no_learn && ctx->n_requests >= x264hip_ctx::LEARN_REQUESTS;
    const int bf = ctx->p.bframes, ns = bf + 2;
    memset( cell_class, 0, (size_t)ns * ns );
    for( int d0 = 1; d0 <= bf + 1; d0++ )
        for( int d1 = 0; d0 + d1 <= bf + 1; d1++ )
        {
            const int idx = d0 * ns + d1;
            if( !ctx->cell_allowed[idx] || ( learned && !ctx->cell_req[idx] ) ) continue;
            const uint32_t *rq = ctx->variant_req[idx];
            // B cells: 3 = both ways in one pass wherever the list-1 reference's field exists (what x264hip_prefetch does); X264HIP_NO_DUAL:
            // the variant asked for more often so far
            static const bool no_dual = getenv( "X264HIP_NO_DUAL" ) != nullptr;
            cell_class[idx] = !d1 ? 1 : !no_dual ? 3 : rq[0] <= rq[1] ? 2 : 1;
        }
    return X264HIP_OK;
}

static bool cell_ref_ok( x264hip_ctx *ctx, const x264hip_cell_ref &c )
{
    return slot_ok( ctx, c.slot_b ) && slot_ok( ctx, c.slot_p0 ) && slot_ok( ctx, c.slot_p1 ) && c.dist_p0 >= 0 && c.dist_p1 >= 0 &&
           c.dist_p0 + c.dist_p1 <= ctx->p.bframes + 1 && !( c.dist_p0 == 0 && c.dist_p1 != 0 );
}

extern "C" int x264hip_spec_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells )
{
    if( !ctx || n < 0 || ( n && !cells ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const int ns = ctx->p.bframes + 2;
    std::vector<SpecCell> list;
    auto local = [&]( FrameSlot &f, int l, int dm1 ) { return ( f.field_ready[l][dm1] || f.field_prefetched[l][dm1] ) && !f.field_remote[l][dm1]; };
    for( int i = 0; i < n; i++ )
    {
        const x264hip_cell_ref &c = cells[i];
        if( !cell_ref_ok( ctx, c ) ) return X264HIP_EINVAL;
        FrameSlot &b = ctx->slots[c.slot_b], &f1 = ctx->slots[c.slot_p1];
        if( !b.in_use || !ctx->slots[c.slot_p0].in_use || !f1.in_use ) return X264HIP_ESTATE;
        const int d0 = c.dist_p0, d1 = c.dist_p1;
        CellEntry &e = b.cells[d0 * ns + d1];
        if( e.valid || e.requested ) continue;
        unsigned t0 = 0, t1 = 0, tr = 0;
        const int variant = d1 && ( c.with_ref1_l0 & X264HIP_CELL_WITH_L0 ), dual = variant && ( c.with_ref1_l0 & X264HIP_CELL_BOTH );
        if( d0 )
        {
            // every field the cell reads has to be in this context (searched here, or imported)
            if( !local( b, 0, d0 - 1 ) || ( d1 && ( !local( b, 1, d1 - 1 ) || ( variant && !local( f1, 0, d0 + d1 - 1 ) ) ) ) ) return X264HIP_ESTATE;
            t0 = b.field_tag[0][d0 - 1];
            if( d1 ) { t1 = b.field_tag[1][d1 - 1]; tr = variant ? f1.field_tag[0][d0 + d1 - 1] : 0; }
            ctx->cell_spec[d0 * ns + d1]++;
        }
        e.valid = 1; e.batch = ctx->batch_serial + 1; e.variant = (unsigned char)( d1 ? variant : 1 );
        e.tag0 = t0; e.tag1 = t1; e.tagr = tr; e.map_remote = 0;
        SpecCell sc{ c.slot_p0, c.slot_p1, c.slot_b, d0, d1, d0 == 0, variant };
        sc.dual = dual;
        list.push_back( sc );
        if( dual )
        {
            CellEntry &a = b.alts[d0 * ns + d1];
            a = CellEntry();
            a.valid = 1; a.batch = ctx->batch_serial + 1; a.variant = 0;
            a.tag0 = t0; a.tag1 = t1; a.tagr = 0;
            ctx->counters[14]++;
        }
    }
    if( list.empty() ) return X264HIP_OK;
    int r = ctx->p.bit_depth == 8 ? launch_cells_t<uint8_t>( ctx, list ) : launch_cells_t<uint16_t>( ctx, list );
    if( r ) return r;
    ctx->counters[5] += list.size();
    return batch_close( ctx );
}

__global__ __launch_bounds__( 64 ) void export_cells_kernel( const CellXfer *__restrict__ x, int mb_h, int *__restrict__ dst )
{
    const CellXfer X = load_uniform( x + blockIdx.x );
    int *d = dst + (size_t)blockIdx.x * X264HIP_CELL_SUMMARY_INTS( mb_h );
    if( threadIdx.x < 8 ) d[threadIdx.x] = threadIdx.x < 5 ? X.acc_dev[threadIdx.x] : 0;
    for( int i = threadIdx.x; i < mb_h; i += 64 )
    {
        d[8 + i] = X.rows[i];
        d[8 + mb_h + i] = X.rows_intra[i];
    }
}
__global__ __launch_bounds__( 64 ) void import_cells_kernel( const CellXfer *__restrict__ x, int mb_h, const int *__restrict__ src )
{
    const CellXfer X = load_uniform( x + blockIdx.x );
    if( X.skip ) return;
    const int *s = src + (size_t)blockIdx.x * X264HIP_CELL_SUMMARY_INTS( mb_h );
    if( threadIdx.x < 5 )
    {
        const int v = s[threadIdx.x];
        X.acc_dev[threadIdx.x] = v;
        __hip_atomic_store( X.acc_host + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ); // pinned host record
    }
    // (the intra row sums of the summary are NOT taken over: the rank that decides evaluates the intra sums of every frame itself --
    // x264hip_spec_cells with a d0 == 0 entry -- and an owner that never did would overwrite them with zeros)
    for( int i = threadIdx.x; i < mb_h; i += 64 )
        X.rows[i] = s[8 + i];
}

static CellXfer make_xfer( x264hip_ctx *ctx, const x264hip_cell_ref &c, bool importing )
{
    FrameSlot &b = ctx->slots[c.slot_b];
    const int idx = c.dist_p0 * ( ctx->p.bframes + 2 ) + c.dist_p1;
    const bool spare = ( c.with_ref1_l0 & X264HIP_CELL_SPARE ) != 0;
    // a summary from the owner rank goes to the cell's own place (the caller re-points cell_at there once it has decided to take
    // the entry: an entry that is skipped -- the cell was already answered here, possibly from its spare half -- must not move it)
    const int at = spare ? ctx->spare_at[idx] : importing ? idx : b.cell_at[idx];
    CellXfer X;
    X.acc_host = ( spare ? ctx->cell_alt_host : ctx->cell_acc_host ) + ( (size_t)c.slot_b * ctx->n_cells + idx ) * 8;
    X.acc_dev = b.cell_sums + (size_t)at * 8;
    X.rows = b.row_satds + (size_t)at * ctx->P.mb_h;
    X.rows_intra = b.row_satds;
    X.skip = 0; X.pad_ = 0;
    return X;
}

extern "C" int x264hip_export_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, void *dst_dev )
{
    if( !ctx || n < 0 || ( n && ( !cells || !dst_dev ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const size_t per = X264HIP_CELL_SUMMARY_INTS( ctx->P.mb_h );
    for( int o = 0; o < n; o += ctx->xfer_cap )
    {
        const int m = std::min( n - o, ctx->xfer_cap );
        int ri = 0;
        if( ring_acquire( ctx->xfer_ring, &ri ) ) return X264HIP_EDEVICE;
        CellXfer *xh = (CellXfer *)ctx->xfer_ring.host[ri], *xd = (CellXfer *)ctx->xfer_ring.dev[ri];
        for( int i = 0; i < m; i++ )
        {
            if( !cell_ref_ok( ctx, cells[o + i] ) ) return X264HIP_EINVAL;
            xh[i] = make_xfer( ctx, cells[o + i], false );
        }
        HIPCK( upload_async( ctx, xd, xh, (size_t)m * sizeof( CellXfer ), ctx->stream ) );
        export_cells_kernel<<<m, 64, 0, ctx->stream>>>( xd, ctx->P.mb_h, (int *)dst_dev + (size_t)o * per );
        HIPCK( hipGetLastError() );
        if( ring_commit( ctx->xfer_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_import_cells( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, const void *src_dev )
{
    if( !ctx || n < 0 || ( n && ( !cells || !src_dev ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    const size_t per = X264HIP_CELL_SUMMARY_INTS( ctx->P.mb_h );
    const int ns = ctx->p.bframes + 2;
    int taken = 0;
    for( int o = 0; o < n; o += ctx->xfer_cap )
    {
        const int m = std::min( n - o, ctx->xfer_cap );
        int ri = 0;
        if( ring_acquire( ctx->xfer_ring, &ri ) ) return X264HIP_EDEVICE;
        CellXfer *xh = (CellXfer *)ctx->xfer_ring.host[ri], *xd = (CellXfer *)ctx->xfer_ring.dev[ri];
        for( int i = 0; i < m; i++ )
        {
            const x264hip_cell_ref &c = cells[o + i];
            if( !cell_ref_ok( ctx, c ) || !c.dist_p0 ) return X264HIP_EINVAL; // (the intra sums of a frame are computed where they are needed)
            FrameSlot &b = ctx->slots[c.slot_b], &f1 = ctx->slots[c.slot_p1];
            if( !b.in_use ) return X264HIP_ESTATE;
            const int d0 = c.dist_p0, d1 = c.dist_p1;
            const bool spare = d1 && ( c.with_ref1_l0 & X264HIP_CELL_SPARE );
            CellEntry &e = spare ? b.alts[d0 * ns + d1] : b.cells[d0 * ns + d1];
            xh[i] = make_xfer( ctx, c, true );
            // the cell stands for the fields as this context knows them now (registered with x264hip_fields_remote, or local)
            auto known = [&]( FrameSlot &f, int l, int dm1 ) { return f.field_ready[l][dm1] || f.field_prefetched[l][dm1]; };
            const int variant = d1 && !spare && ( c.with_ref1_l0 & X264HIP_CELL_WITH_L0 );
            const bool inputs = known( b, 0, d0 - 1 ) && ( !d1 || ( known( b, 1, d1 - 1 ) && ( !variant || known( f1, 0, d0 + d1 - 1 ) ) ) );
            if( e.valid || e.requested || b.cells[d0 * ns + d1].requested || !inputs ) { xh[i].skip = 1; continue; }
            if( !spare ) b.cell_at[d0 * ns + d1] = d0 * ns + d1;
            e.valid = 1; e.batch = ctx->batch_serial + 1; e.variant = (unsigned char)( d1 ? variant : 1 );
            e.tag0 = b.field_tag[0][d0 - 1];
            e.tag1 = d1 ? b.field_tag[1][d1 - 1] : 0;
            e.tagr = variant ? f1.field_tag[0][d0 + d1 - 1] : 0;
            e.map_remote = 1; e.slot_p0 = c.slot_p0; e.slot_p1 = c.slot_p1;
            taken++;
        }
        HIPCK( upload_async( ctx, xd, xh, (size_t)m * sizeof( CellXfer ), ctx->stream ) );
        import_cells_kernel<<<m, 64, 0, ctx->stream>>>( xd, ctx->P.mb_h, (const int *)src_dev + (size_t)o * per );
        HIPCK( hipGetLastError() );
        if( ring_commit( ctx->xfer_ring, ri, ctx->stream ) ) return X264HIP_EDEVICE;
    }
    ctx->counters[11] += taken;
    return n ? batch_close( ctx ) : X264HIP_OK;
}

extern "C" int x264hip_fields_remote( x264hip_ctx *ctx, int n, const int *slot, const int *frame_number, const int *list, const int *dist_minus1 )
{
    if( !ctx || n < 0 || ( n && ( !slot || !frame_number || !list || !dist_minus1 ) ) ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    for( int i = 0; i < n; i++ )
    {
        if( !slot_ok( ctx, slot[i] ) || list[i] < 0 || list[i] > 1 || dist_minus1[i] < 0 || dist_minus1[i] > ctx->p.bframes ) return X264HIP_EINVAL;
        FrameSlot &f = ctx->slots[slot[i]];
        if( !f.in_use ) return X264HIP_ESTATE;
        f.frame_no = frame_number[i];
        if( f.field_ready[list[i]][dist_minus1[i]] || f.field_prefetched[list[i]][dist_minus1[i]] ) continue; // this context has it
        f.field_prefetched[list[i]][dist_minus1[i]] = 1;
        f.field_remote[list[i]][dist_minus1[i]] = 1;
        f.field_tag[list[i]][dist_minus1[i]] = ctx->tag_serial++;
        if( !ctx->tag_serial ) ctx->tag_serial = 1;
        ctx->counters[12]++;
    }
    return X264HIP_OK;
}

extern "C" int x264hip_cells_missing( x264hip_ctx *ctx, int n, const x264hip_cell_ref *cells, unsigned char *missing )
{
    if( !ctx || n < 0 || ( n && ( !cells || !missing ) ) ) return X264HIP_EINVAL;
    const int ns = ctx->p.bframes + 2;
    for( int i = 0; i < n; i++ )
    {
        if( !cell_ref_ok( ctx, cells[i] ) ) return X264HIP_EINVAL;
        const FrameSlot &b = ctx->slots[cells[i].slot_b];
        const int idx = cells[i].dist_p0 * ns + cells[i].dist_p1;
        // 1: the map of the cell's own evaluation is with its owner, 2: the map of its spare half (the caller asked for that variant)
        missing[i] = b.cells[idx].map_remote ? ( b.cell_at[idx] != idx ? 2 : 1 ) : 0;
    }
    return X264HIP_OK;
}

__global__ __launch_bounds__( 256 ) void export_map_kernel( const uint16_t *__restrict__ costs, const unsigned long long *__restrict__ mvq0, const unsigned long long *__restrict__ mvq1,
                                                            int *__restrict__ dst, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n )
    {
        dst[i] = costs[i];
        dst[n + i] = (int)(unsigned)mvq0[i];
        dst[2 * n + i] = mvq1 ? (int)(unsigned)mvq1[i] : 0;
    }
}
__global__ __launch_bounds__( 256 ) void import_map_kernel( uint16_t *__restrict__ costs, unsigned long long *__restrict__ mvq0, unsigned tag0, unsigned long long *__restrict__ mvq1,
                                                            unsigned tag1, const int *__restrict__ src, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n )
    {
        costs[i] = (uint16_t)src[i];
        if( mvq0 ) mvq0[i] = ( (unsigned long long)tag0 << 32 ) | (unsigned)src[n + i];
        if( mvq1 ) mvq1[i] = ( (unsigned long long)tag1 << 32 ) | (unsigned)src[2 * n + i];
    }
}

extern "C" int x264hip_export_cell_map( x264hip_ctx *ctx, const x264hip_cell_ref *cell, void *dst_dev )
{
    if( !ctx || !cell || !dst_dev || !cell_ref_ok( ctx, *cell ) || !cell->dist_p0 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &b = ctx->slots[cell->slot_b];
    const int d0 = cell->dist_p0, d1 = cell->dist_p1, idx = d0 * ( ctx->p.bframes + 2 ) + d1;
    const bool spare = d1 && ( cell->with_ref1_l0 & X264HIP_CELL_SPARE );
    const CellEntry &e = spare ? b.alts[idx] : b.cells[idx];
    if( !b.in_use || e.map_remote || ( !e.valid && !e.requested ) ) return X264HIP_ESTATE; // the map has to have been evaluated HERE
    const int at = spare ? ctx->spare_at[idx] : b.cell_at[idx];
    export_map_kernel<<<( ctx->n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( b.lowres_costs + (size_t)at * ctx->n_mb, b.mvq[0][d0 - 1], d1 ? b.mvq[1][d1 - 1] : nullptr,
                                                                          (int *)dst_dev, ctx->n_mb );
    HIPCK( hipGetLastError() );
    return X264HIP_OK;
}

extern "C" int x264hip_import_cell_map( x264hip_ctx *ctx, const x264hip_cell_ref *cell, const void *src_dev )
{
    if( !ctx || !cell || !src_dev || !cell_ref_ok( ctx, *cell ) || !cell->dist_p0 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE; // the current device is per host thread
    FrameSlot &b = ctx->slots[cell->slot_b];
    const int d0 = cell->dist_p0, d1 = cell->dist_p1, idx = d0 * ( ctx->p.bframes + 2 ) + d1;
    CellEntry &e = b.cells[idx];
    if( !b.in_use ) return X264HIP_ESTATE;
    if( !e.map_remote ) return X264HIP_OK; // evaluated here in the meantime: identical by construction
    // vectors of fields this context only knows by tag arrive with the map (their costs stay with the owner); local fields are left alone
    unsigned long long *q0 = b.field_remote[0][d0 - 1] ? b.mvq[0][d0 - 1] : nullptr;
    unsigned long long *q1 = d1 && b.field_remote[1][d1 - 1] ? b.mvq[1][d1 - 1] : nullptr;
    import_map_kernel<<<( ctx->n_mb + 255 ) / 256, 256, 0, ctx->stream>>>( b.lowres_costs + (size_t)b.cell_at[idx] * ctx->n_mb, q0, b.field_tag[0][d0 - 1], q1,
                                                                          d1 ? b.field_tag[1][d1 - 1] : 0u, (const int *)src_dev, ctx->n_mb );
    HIPCK( hipGetLastError() );
    if( q0 ) b.field_remote[0][d0 - 1] = 2;
    if( q1 ) b.field_remote[1][d1 - 1] = 2;
    e.map_remote = 0;
    ctx->counters[10]++;
    return X264HIP_OK;
}

template <typename T>
static int mc_luma_probe_t( x264hip_ctx *ctx, int slot, int n, const x264hip_mc_probe *req, const x264hip_weight *w, void *out )
{
    FrameSlot &f = ctx->slots[slot];
    McProbe *rd = nullptr;
    T *od = nullptr;
    int rc = X264HIP_OK;
    if( hipMalloc( &rd, (size_t)n * sizeof( McProbe ) ) != hipSuccess || hipMalloc( &od, (size_t)n * 64 * sizeof( T ) ) != hipSuccess )
        rc = X264HIP_ENOMEM;
    else
    {
        static_assert( sizeof( McProbe ) == sizeof( x264hip_mc_probe ), "the request record is the public one" );
        const WtD wt = w ? make_wt( ctx, w ) : WtD{ 0, 1, 0, 0 };
        if( hipMemcpyAsync( rd, req, (size_t)n * sizeof( McProbe ), hipMemcpyHostToDevice, ctx->stream ) != hipSuccess )
            rc = X264HIP_EDEVICE;
        else
        {
            mc_probe_kernel<T><<<( n + 7 ) / 8, 64, 0, ctx->stream>>>( ctx->P, (const T *)( f.planes + 4 * ctx->plane_bytes ), rd, n, wt, od );
            if( hipMemcpyAsync( out, od, (size_t)n * 64 * sizeof( T ), hipMemcpyDeviceToHost, ctx->stream ) != hipSuccess ||
                hipStreamSynchronize( ctx->stream ) != hipSuccess || hipGetLastError() != hipSuccess )
                rc = X264HIP_EDEVICE;
        }
    }
    if( rd ) (void)hipFree( rd );
    if( od ) (void)hipFree( od );
    return rc;
}

extern "C" int x264hip_mc_luma_probe( x264hip_ctx *ctx, int slot, int n, const x264hip_mc_probe *req, const x264hip_weight *w, void *out )
{
    if( !ctx || !req || !out || n < 0 ) return X264HIP_EINVAL;
    if( ctx->broken ) return X264HIP_EDEVICE;
    if( !slot_ok( ctx, slot ) || !ctx->slots[slot].in_use ) return X264HIP_ESTATE;
    if( !n ) return X264HIP_OK;
    // what a legal candidate can reach: the picture plus the 8-sample block inside the padding, one sample spare for the quarter-pel partner
    const int W8 = 8 * ctx->P.mb_w, H8 = 8 * ctx->P.mb_h;
    for( int i = 0; i < n; i++ )
    {
        const int x0 = req[i].x + ( req[i].mvx >> 2 ), y0 = req[i].y + ( req[i].mvy >> 2 );
        if( x0 < -LA_PAD || x0 + 9 > W8 + LA_PAD || y0 < -LA_PAD || y0 + 9 > H8 + LA_PAD ) return X264HIP_EINVAL;
    }
    if( hipSetDevice( ctx->device ) != hipSuccess ) return X264HIP_EDEVICE;
    return ctx->p.bit_depth == 8 ? mc_luma_probe_t<uint8_t>( ctx, slot, n, req, w, out ) : mc_luma_probe_t<uint16_t>( ctx, slot, n, req, w, out );
}

extern "C" int x264hip_spec_classes( x264hip_ctx *ctx, const unsigned char *cell_allowed, unsigned mask_l0, unsigned mask_l1 )
{
    if( !ctx ) return X264HIP_EINVAL;
    static const bool off = getenv( "X264HIP_NO_STATIC_CLASSES" ) != nullptr; // A/B runs: learn everything from the requests
    if( off ) return X264HIP_OK;
    const int ns = ctx->p.bframes + 2;
    if( cell_allowed ) memcpy( ctx->cell_allowed, cell_allowed, (size_t)ns * ns );
    else memset( ctx->cell_allowed, 1, sizeof( ctx->cell_allowed ) );
    ctx->field_allowed[0] = mask_l0; ctx->field_allowed[1] = mask_l1;
    return X264HIP_OK;
}

extern "C" int x264hip_class_requests( x264hip_ctx *ctx, uint32_t *field_req, uint32_t *cell_req, unsigned char *cell_allowed, unsigned *field_allowed )
{
    if( !ctx ) return X264HIP_EINVAL;
    const int ns = ctx->p.bframes + 2;
    if( field_req )
        for( int l = 0; l < 2; l++ )
            for( int d = 0; d <= ctx->p.bframes; d++ )
                field_req[l * ( ctx->p.bframes + 1 ) + d] = ctx->field_req[l][d];
    if( cell_req ) memcpy( cell_req, ctx->cell_req, sizeof( uint32_t ) * ns * ns );
    if( cell_allowed ) memcpy( cell_allowed, ctx->cell_allowed, (size_t)ns * ns );
    if( field_allowed ) { field_allowed[0] = ctx->field_allowed[0]; field_allowed[1] = ctx->field_allowed[1]; }
    return X264HIP_OK;
}

extern "C" int x264hip_field_classes( x264hip_ctx *ctx, unsigned *mask_l0, unsigned *mask_l1 )
{
    if( !ctx || !mask_l0 || !mask_l1 ) return X264HIP_EINVAL;
    static const bool no_learn = getenv( "X264HIP_NO_CLASS_LEARNING" ) != nullptr;
    const bool learned = !no_learn && ctx->n_requests >= x264hip_ctx::LEARN_REQUESTS;
    unsigned m[2] = { 0, 0 };
    for( int l = 0; l < 2; l++ )
        for( int d = 0; d <= ctx->p.bframes; d++ )
            if( ( ctx->field_allowed[l] >> d & 1 ) && ( !learned || ctx->field_req[l][d] ) )
                m[l] |= 1u << d;
    *mask_l0 = m[0]; *mask_l1 = m[1];
    return X264HIP_OK;
}
