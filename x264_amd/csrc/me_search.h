// me_search.h -- the lookahead motion search of whole frames on gfx950: four block rows per wave64, one 8x8 block per 16-lane group.
//
// Behaviour follows the reference's slicetype_mb_cost search part (encoder/slicetype.c:654-709) over x264_me_search_ref
// (encoder/me.c:182-420,774-798, DIA and HEX) + refine_subpel (me.c:865-992); the decision logic is me_logic.h.
//
// Work decomposition.  A search (source frame, reference frame, list, distance) is a W x H field of 8x8 blocks scanned from the
// bottom right; block (x, y) takes its predictors from (x+1, y) and (x-1..x+1, y+1).  A wave owns ME_ROWS = 4 consecutive block rows:
// lane group g (16 lanes, the Px4 geometry of device_common.h) walks row y0 - g from right to left, two blocks behind the group
// below it, so in step t group g searches block x = W-1 - (t - 2g) and finds the three vectors of the row below in the registers
// of group g-1 (its results of steps t-1, t-2, t-3; one ds_bpermute each, no memory).  Each group runs the whole block search as
// its own instruction stream (SIMT over groups: candidates one after the other, costs reduced over the 16 lanes with DPP): the
// selection logic is vector code shared by four blocks instead of scalar code serving one, and a row of hand-offs through
// memory is needed only every fourth row.  Only the bottom group of a wave waits for another wave: the top row of the wave below
// publishes self-validating 8-byte granules { mv, tag } with agent-scope relaxed atomics (`sc1` write-through stores /
// L1-bypassing loads, MI355X guide G16 form R2); the other rows' results leave as plain stores.
// Row groups are claimed bottom-up through ticket counters, so the wave a group depends on always holds an earlier ticket and is
// running or done: the in-kernel waits cannot deadlock whatever the dispatch order.
#pragma once
#include "device_common.h"
#define ME_HD __device__ __forceinline__
#ifdef ME_PROFILE
#define ME_MARK( ev, k ) ( ev ).mark( k )
#endif
#include "me_logic.h"

#define ME_ROWS 4

template <typename T>
struct SearchDesc
{
    const T *fenc0;             // source frame plane 0 origin
    const T *ref0;              // reference frame plane 0 origin (planes 1..3 at +plane_elems*k)
    const T *refw;              // weighted plane 0 origin or nullptr
    WtD wt;
    unsigned long long *mvq;    // [n_mb] granules: low 32 = mvx | mvy<<16, high 32 = tag
    int *costs;                 // [n_mb]
    unsigned tag;               // non-zero, unique per use of mvq
    int pad;
    const T *ref_strips;        // strip copy of the reference's four planes (me_search8.h), start of the allocation
    const T *refw_strips;       // strip copy of the weighted plane 0 or nullptr
};

// The searches of a launch are split into ME_QUEUES contiguous groups (neighbouring frames), one ticket counter
// each, and a wave serves the group of the XCD it runs on: the planes of a group then live in ONE of the eight
// L2s instead of being fetched into all of them (a row band of every frame of the launch is active at any time,
// far more than a 4 MB L2 holds).  The XCD id is used for locality only: a wave whose own group has no rows left
// takes rows from the other groups, so every row is claimed whatever the dispatcher's workgroup placement is,
// and the order inside a group still guarantees that the rows below hold an earlier ticket.
#define ME_QUEUES 8
#define ME_QUEUE_STRIDE 16 // counters 64 bytes apart
struct MeQueues
{
    int base[ME_QUEUES + 1]; // searches [base[q], base[q+1]) belong to group q
};

__device__ __forceinline__ int xcc_id()
{
    return __builtin_amdgcn_s_getreg( 20 | ( 0 << 6 ) | ( ( 4 - 1 ) << 11 ) ) & ( ME_QUEUES - 1 ); // HW_REG_XCC_ID[3:0]
}

// the evaluator of me_logic.h on the 16-lane block geometry: every cost is the same in the 16 lanes of a group
// mv costs come from the cost_mv table (h->cost_mv[X264_LOOKAHEAD_QP], analyse.c:151-157), indexed by the quarter-pel difference to
// the block's predictor.  Two table loads per candidate made the vector-memory pipe the busiest unit of the kernel (every wave64 load
// instruction occupies the address unit for 16 cycles whatever it fetches), so each wave keeps the entries for differences of
// less than ME_TAB_HALF quarter-pels in LDS; a block whose candidates could reach beyond that window (predictor hundreds of
// samples long) is searched with the table in memory instead.
#define ME_TAB_HALF 1024
template <typename T, int LDS_TAB>
struct GroupEval
{
    const uint16_t *lds_tab; // this wave's window: entry ME_TAB_HALF + d is the cost of difference d
    const T *rbase;      // wave-uniform: start of the reference frame's four-plane allocation (unweighted)
    const T *wbase;      // wave-uniform: plane read by full-pel candidates (weighted copy of plane 0, or rbase)
    const uint16_t *tab; // wave-uniform: first entry of the cost_mv table
    int plane_elems, stride, pixel_max;
    int fpelcmp_satd;
    WtD wt;
    int lane_off;        // element offset of this lane's 4 samples of the block at zero displacement (plane 0)
    int tab_x, tab_y;    // table index of a quarter-pel component q is q + tab_x / q + tab_y
    Px4 f;               // this lane's 4 source pixels

    __device__ __forceinline__ int bits( int qx, int qy ) const
    {
        if( LDS_TAB ) // tab_x / tab_y address the window around a zero mv difference that sits in LDS (me_rows_kernel)
            return lds_tab[qx + tab_x] + lds_tab[qy + tab_y];
        return gload_u16( tab, 2u * (unsigned)( qx + tab_x ) ) + gload_u16( tab, 2u * (unsigned)( qy + tab_y ) );
    }
    __device__ __forceinline__ int fpel( int x, int y ) const
    {
        const Px4 r = load_px4_at( wbase, lane_off + mad24( y, stride, x ) );
        return block_cost8x8<T>( f, r, fpelcmp_satd );
    }
    __device__ __forceinline__ int qpel( int qx, int qy, int use_satd ) const
    {
        Px4 r = qpel_px4_at( rbase, plane_elems, stride, lane_off, qx, qy );
        if( wt.on )
            r = weight_px4<T>( r, wt, pixel_max );
        return block_cost8x8<T>( f, r, use_satd );
    }
    __device__ __forceinline__ bool any( bool c ) const { return __builtin_amdgcn_ballot_w64( c ) != 0ull; }
#ifdef ME_PROFILE
    unsigned long long pf_last;
    unsigned pf_phase[5]; // [k] = cycles between mark k-1 and mark k: 1 start candidates, 2 pattern, 3 half-pel, 4 quarter-pel
    __device__ __forceinline__ void mark( int k )
    {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if( k ) pf_phase[k] += (unsigned)( now - pf_last );
        pf_last = now;
    }
#endif
};

// value of the same lane position one group (16 lanes) further down; group 0 gets its own value back
__device__ __forceinline__ int from_group_below( int v, int lane )
{
    return __builtin_amdgcn_ds_bpermute( ( ( lane - 16 ) & 63 ) << 2, v );
}

// MODE fixes the metric pair at compile time (every cost evaluation would otherwise carry both metrics behind a branch):
//   0  sub-pel depth 2, mbcmp = fpelcmp = SAD            (subme <= 1, encoder.c:1409-1427 + slicetype.c:45-61)
//   1  sub-pel depth 4, mbcmp = SATD, fpelcmp = SAD      (subme >= 2)
//   2  sub-pel depth 4, mbcmp = fpelcmp = SATD           (subme >= 2 with --me tesa)
//   3  any other combination a caller configures: depth and metrics read from the parameters at run time
#ifndef ME_MIN_WAVES
#define ME_MIN_WAVES 4
#endif
template <typename T, int HEX, int MODE>
__global__ __launch_bounds__( 64, ME_MIN_WAVES ) void me_rows_kernel( LaP P, const SearchDesc<T> *descs, MeQueues Q, unsigned *tickets /* [ME_QUEUES * ME_QUEUE_STRIDE] */,
                                                         unsigned *err_host /* pinned sticky timeout flag */, unsigned spin_limit,
                                                         unsigned long long *prof /* ME_PROFILE builds: cycle accumulators, else unused */ )
{
    const int lane = lane_id();
#ifdef ME_PROFILE
    unsigned long long pf_wait = 0, pf_pre = 0, pf_search = 0, pf_store = 0, pf_spins = 0, pf_steps = 0;
    unsigned long long pf_ph[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long pf_begin = __builtin_amdgcn_s_memtime();
#define PF_NOW() __builtin_amdgcn_s_memtime()
#endif
    const int W = P.mb_w, H = P.mb_h;
    const int n_rowgroups = ( H + ME_ROWS - 1 ) / ME_ROWS;
    // the ticket is wave-uniform: fetch it on lane 0 and broadcast through an SGPR so that the row group, the descriptor and
    // everything derived from them stay scalar
    const int home = xcc_id();
    int j = 0, s = -1;
    for( int k = 0; k < ME_QUEUES && s < 0; k++ )
    {
        const int q = ( home + k ) & ( ME_QUEUES - 1 );
        const int n_q = Q.base[q + 1] - Q.base[q];
        if( !n_q )
            continue;
        unsigned t0 = 0;
        if( lane == 0 )
            t0 = atomicAdd( &tickets[q * ME_QUEUE_STRIDE], 1u );
        const unsigned t = __builtin_amdgcn_readfirstlane( t0 );
        if( t < (unsigned)( n_q * n_rowgroups ) )
        {
            j = t / n_q;
            s = Q.base[q] + ( t - j * n_q );
        }
    }
    if( s < 0 )
        return;
    const SearchDesc<T> D = descs[s];
    const int g = lane >> 4;
    const int by0 = H - 1 - ME_ROWS * j;  // row of group 0 (scalar)
    const int by = by0 - g;               // this group's row
    const bool row_ok = by >= 0;

    // the cost table window of this wave
    __shared__ uint16_t tab_window[2 * ME_TAB_HALF];
    {
        const int centre = 2 * 4 * P.mv_range; // P.cost_mv is centred: valid differences are -centre .. +centre
        for( int i = lane; i < 2 * ME_TAB_HALF; i += 64 )
        {
            const int d = i - ME_TAB_HALF;
            tab_window[i] = d >= -centre && d <= centre ? P.cost_mv[d] : (uint16_t)0;
        }
        __syncthreads(); // one wave per workgroup: orders the LDS writes before the first block's reads
    }
    MeCfg C;
    C.hex = HEX; C.me_range = P.me_range;
    C.refine4 = MODE == 3 ? P.subpel_refine >= 3 : MODE >= 1;
    C.mbcmp_satd = MODE == 3 ? P.mbcmp_satd : MODE >= 1;
    C.fpelcmp_satd = MODE == 3 ? P.fpelcmp_satd : MODE == 2;
    GroupEval<T, 0> ev;
#ifdef ME_PROFILE
    for( int k = 0; k < 5; k++ ) ev.pf_phase[k] = 0;
#endif
    ev.lds_tab = tab_window;
    const int border = LA_PAD * P.stride + LA_PAD;
    const T *fbase = D.fenc0 - border;
    ev.rbase = D.ref0 - border;
    ev.wbase = D.wt.on ? D.refw - border : ev.rbase;
    const int tab_centre = 2 * 4 * P.mv_range;
    ev.tab = P.cost_mv - tab_centre;
    ev.plane_elems = P.plane_elems; ev.stride = P.stride; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd;
    ev.wt = D.wt;
    int row_off; // this lane's 4 samples inside an 8x8 block
    {
        const int l = lane & 15, q = l >> 2;
        row_off = ( ( q >> 1 ) * 4 + ( l & 3 ) ) * P.stride + ( q & 1 ) * 4;
    }
    // end row of the band this row belongs to (slicetype.c:917-918): rows of one band do not see the vectors of the band below
    // (slicetype.c:668).  One band: H.
    int band_end = H;
    for( int sl = P.n_slices - 1; sl >= 1; sl-- )
    {
        const int start = ( H * sl + P.n_slices / 2 ) / P.n_slices;
        if( by < start )
            band_end = start;
    }
    const bool has_below = row_ok && by < band_end - 1;
    const int zero_bits = P.cost_mv[0];

    int r1 = 0, r2 = 0, r3 = 0; // packed vectors this group found in the last three steps
    const int n_steps = W + 2 * ( ME_ROWS - 1 );
    for( int t = 0; t < n_steps; t++ )
    {
        const int bx = W - 1 - ( t - 2 * g );
        const bool active = row_ok && bx >= 0 && bx < W;
#ifdef ME_PROFILE
        const unsigned long long pf_t0 = PF_NOW();
#endif
        // the row below: (x-1, y+1), (x, y+1), (x+1, y+1) are what the group below found one, two and three steps ago
        int below_left = from_group_below( r1, lane ), below = from_group_below( r2, lane ), below_right = from_group_below( r3, lane );
        {
            // group 0's row below belongs to another wave: wait for its three granules
            const int bx0 = W - 1 - t;
            const bool need0 = bx0 >= 0 && (bool)__builtin_amdgcn_readfirstlane( (int)has_below );
            if( need0 )
            {
                unsigned long long gq = 0;
                const int nb = lane == 1 ? ( bx0 > 0 ? -1 : 0 ) : lane == 2 ? ( bx0 < W - 1 ? 1 : 0 ) : 0;
                const unsigned long long *gp = D.mvq + ( ( by0 + 1 ) * W + bx0 + nb );
                unsigned spins = 0;
                while( 1 )
                {
                    bool ok = true;
                    if( lane < 3 )
                    {
                        gq = __hip_atomic_load( gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                        ok = (unsigned)( gq >> 32 ) == D.tag;
                    }
                    if( __all( ok ) )
                        break;
                    if( ++spins > spin_limit )
                    {
                        if( lane == 0 )
                            __hip_atomic_store( err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
                        return;
                    }
                    __builtin_amdgcn_s_sleep( 4 );
#ifdef ME_PROFILE
                    pf_spins++;
#endif
                }
                const int lo = (int)(unsigned)gq;
                const int w0 = __builtin_amdgcn_readlane( lo, 0 ), w1 = __builtin_amdgcn_readlane( lo, 1 ), w2 = __builtin_amdgcn_readlane( lo, 2 );
                if( g == 0 ) { below = w0; below_left = w1; below_right = w2; }
            }
        }
        int mvx = 0, mvy = 0, cost = 0;
#ifdef ME_PROFILE
        const unsigned long long pf_t1 = PF_NOW();
        unsigned long long pf_t2 = pf_t1, pf_t3 = pf_t1;
#endif
        if( active )
        {
            const int xy = by * W + bx;
            if( la_visited( P, bx, by ) )
            {
                MeLim L;
                melogic::block_limits( L, bx, by, W, H, P.mv_range );
                int mvcx[4], mvcy[4];
                const int n = melogic::neighbour_list( bx, W, has_below, r1, below, below_left, below_right, mvcx, mvcy );
                int mvpx, mvpy;
                if( n <= 1 ) { mvpx = mvcx[0]; mvpy = mvcy[0]; }
                else
                {
                    mvpx = melogic::median3( mvcx[0], mvcx[1], mvcx[2] );
                    mvpy = melogic::median3( mvcy[0], mvcy[1], mvcy[2] );
                }
                ev.lane_off = border + 8 * ( by * P.stride + bx ) + row_off;
                ev.tab_x = tab_centre - mvpx;
                ev.tab_y = tab_centre - mvpy;
                ev.f = load_px4_at( fbase, ev.lane_off );
                bool done = false;
                if( !( mvpx | mvpy ) )
                {
                    // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692)
                    const Px4 r = load_px4_at( ev.rbase, ev.lane_off );
                    cost = block_cost8x8<T>( ev.f, r, C.mbcmp_satd );
                    done = cost < 64;
                }
#ifdef ME_PROFILE
                pf_t2 = PF_NOW();
#endif
                if( !done )
                {
                    // how far from the predictor can a candidate of this block be?  The start candidates (zero, the neighbours, the
                    // predictor clipped to the full-pel and to the sub-pel limits) plus what the pattern search and the sub-pel
                    // refinement can move away from them: me_range full-pel steps (DIA; HEX: me_range + 1) + 3 quarter-pels
                    int reach = imax2( iabs( mvpx ), iabs( mvpy ) );
#pragma unroll
                    for( int i = 0; i < 4; i++ )
                        if( i < n )
                            reach = imax2( reach, imax2( iabs( mvcx[i] - mvpx ), iabs( mvcy[i] - mvpy ) ) );
                    reach = imax2( reach, imax2( iabs( iclip3( mvpx, 4 * L.fmin_x, 4 * L.fmax_x ) - mvpx ), iabs( iclip3( mvpy, 4 * L.fmin_y, 4 * L.fmax_y ) - mvpy ) ) );
                    reach = imax2( reach, imax2( iabs( iclip3( mvpx, L.smin_x + 2, L.smax_x - 2 ) - mvpx ), iabs( iclip3( mvpy, L.smin_y + 2, L.smax_y - 2 ) - mvpy ) ) );
                    const bool far = reach + 4 * ( P.me_range + 4 ) >= ME_TAB_HALF;
                    if( __builtin_amdgcn_ballot_w64( far ) == 0ull )
                    {
                        GroupEval<T, 1> evl;
                        evl.lds_tab = tab_window; evl.rbase = ev.rbase; evl.wbase = ev.wbase; evl.tab = ev.tab; evl.plane_elems = ev.plane_elems;
                        evl.stride = ev.stride; evl.pixel_max = ev.pixel_max; evl.fpelcmp_satd = ev.fpelcmp_satd; evl.wt = ev.wt;
                        evl.lane_off = ev.lane_off; evl.f = ev.f;
                        evl.tab_x = ME_TAB_HALF - mvpx; evl.tab_y = ME_TAB_HALF - mvpy;
#ifdef ME_PROFILE
                        for( int k = 0; k < 5; k++ ) evl.pf_phase[k] = 0;
#endif
                        melogic::search( C, L, evl, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
#ifdef ME_PROFILE
                        for( int k = 1; k < 5; k++ ) pf_ph[k] += evl.pf_phase[k];
#endif
                    }
                    else
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                    cost -= zero_bits;
                    if( mvx | mvy )
                        cost += 5 * P.lambda;
                }
            }
#ifdef ME_PROFILE
            pf_t3 = PF_NOW();
#endif
            // blocks slicetype_slice_cost never visits (slicetype.c:823-833) keep zero vectors (frame.c:283-285)
            if( ( lane & 15 ) == 0 )
            {
                const unsigned long long gv = ( (unsigned long long)D.tag << 32 ) | (unsigned)( ( mvx & 0xFFFF ) | ( mvy << 16 ) );
                if( g == ME_ROWS - 1 )
                    __hip_atomic_store( D.mvq + xy, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); // read by the wave above
                else
                    D.mvq[xy] = gv;
                D.costs[xy] = cost;
            }
        }
        r3 = r2; r2 = r1;
        r1 = ( mvx & 0xFFFF ) | ( mvy << 16 );
#ifdef ME_PROFILE
        {
            // wave-level view: the slowest group of the step (times are taken by whichever lane reads the counter last)
            const unsigned long long pf_t4 = PF_NOW();
            const unsigned long long a2 = __builtin_amdgcn_readfirstlane( (unsigned)( pf_t2 - pf_t1 ) ), a3 = __builtin_amdgcn_readfirstlane( (unsigned)( pf_t3 - pf_t2 ) );
            pf_wait += pf_t1 - pf_t0; pf_pre += a2; pf_search += a3; pf_store += pf_t4 - pf_t1 - a2 - a3; pf_steps++;
        }
#endif
    }
#ifdef ME_PROFILE
    if( lane == 0 && prof )
    {
        atomicAdd( prof + 0, PF_NOW() - pf_begin ); atomicAdd( prof + 1, pf_wait ); atomicAdd( prof + 2, pf_pre ); atomicAdd( prof + 3, pf_search );
        atomicAdd( prof + 4, pf_store ); atomicAdd( prof + 5, pf_spins ); atomicAdd( prof + 6, pf_steps ); atomicAdd( prof + 7, 1ull );
        for( int k = 1; k < 5; k++ ) atomicAdd( prof + 7 + k, pf_ph[k] ); // lane 0's view: group 0 of the wave
    }
#endif
}
