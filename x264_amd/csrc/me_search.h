// me_search.h -- device-side motion search of one 8x8 lowres block by one wave64, and the persistent
// "row wave" kernel that runs whole frame searches.
//
// Behaviour follows the reference's x264_me_search_ref (encoder/me.c:182-420,774-798, DIA and HEX
// branches) + refine_subpel (me.c:865-992) as driven by slicetype_mb_cost (encoder/slicetype.c:654-709).
// Candidates are evaluated four at a time (one per 16-lane group); the selection among them is
// wave-uniform scalar code that applies the candidates in the reference's order with strict '<', which
// is what its packed (cost<<k)+index comparisons implement.
#pragma once
#include "device_common.h"

template <typename T>
struct MeBlk
{
    const T *rbase; // wave-uniform: start of the reference frame's four-plane allocation (unweighted)
    const T *wbase; // wave-uniform: plane read by full-pel candidates (weighted copy of plane 0, or rbase)
    const uint16_t *tab; // wave-uniform: first entry of the cost_mv table
    int lane_off;   // element offset of this lane's 4 samples of the block at zero displacement (plane 0)
    int tab_x, tab_y; // table index of a quarter-pel component q is q + tab_x / q + tab_y
    Px4 f;          // this lane's 4 source pixels
    int g;          // candidate group 0..3
    int mvpx, mvpy;
    int smin_x, smin_y, smax_x, smax_y; // quarter-pel limits
    int fmin_x, fmin_y, fmax_x, fmax_y; // full-pel limits
};

template <typename T>
__device__ __forceinline__ int mv_bits( const LaP &P, const MeBlk<T> &B, int qx, int qy )
{
    return gload_u16( B.tab, 2u * (unsigned)( qx + B.tab_x ) ) + gload_u16( B.tab, 2u * (unsigned)( qy + B.tab_y ) );
}

// cost of this lane group's full-pel candidate (cx,cy); bits included when with_bits
template <typename T>
__device__ __forceinline__ int fpel_cost( const LaP &P, const MeBlk<T> &B, int cx, int cy, int with_bits )
{
    // the table lookups are issued before the pixel load is consumed: one memory latency per round
    const int bits = mv_bits( P, B, 4 * cx, 4 * cy );
    const Px4 r = load_px4_at( B.wbase, B.lane_off + mad24( cy, P.stride, cx ) );
    return block_cost8x8<T>( B.f, r, P.fpelcmp_satd ) + ( with_bits ? bits : 0 );
}

// cost of this lane group's quarter-pel candidate (qx,qy) with get_ref semantics (mc.c:218-249)
template <typename T>
__device__ __forceinline__ int qpel_cost( const LaP &P, const MeBlk<T> &B, const WtD &wt, int qx, int qy, int use_satd )
{
    const int bits = mv_bits( P, B, qx, qy );
    Px4 r = qpel_px4_at( B.rbase, P.plane_elems, P.stride, B.lane_off, qx, qy );
    if( wt.on )
        r = weight_px4<T>( r, wt, P.pixel_max );
    return block_cost8x8<T>( B.f, r, use_satd ) + bits;
}

#define GRP_COST( v, k ) __builtin_amdgcn_readlane( v, 16 * ( k ) )

template <typename T>
__device__ __forceinline__ bool in_fpel_range( const MeBlk<T> &B, int x, int y )
{
    return x >= B.fmin_x && x <= B.fmax_x && y >= B.fmin_y && y <= B.fmax_y;
}

// hexagon offsets, index 0..5: (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2)
__device__ __forceinline__ int hex_dx( int k ) { return (int)( ( 0x134310u >> ( 4 * k ) ) & 15 ) - 2; }
__device__ __forceinline__ int hex_dy( int k ) { return (int)( ( 0x002442u >> ( 4 * k ) ) & 15 ) - 2; }
__device__ __forceinline__ int mod6( int v ) { return v < 0 ? v + 6 : v >= 6 ? v - 6 : v; }

template <typename T>
__device__ void me_block( const LaP &P, MeBlk<T> &B, const WtD &wt, int n_mvc, const int *mvcx, const int *mvcy,
                          int &out_mvx, int &out_mvy, int &out_cost )
{
    const int g = B.g;
    int bmx, bmy, bcost;
    int bpred_cost = COST_MAX_I, bpred_mx = 0, bpred_my = 0;
    int pmvx, pmvy;
    int cx[4], cy[4], nc = 0;

    if( P.subpel_refine >= 3 )
    {
        // predictor and neighbour candidates at quarter-pel precision (me.c:216-275)
        bpred_mx = iclip3( B.mvpx, 4 * B.fmin_x, 4 * B.fmax_x );
        bpred_my = iclip3( B.mvpy, 4 * B.fmin_y, 4 * B.fmax_y );
        pmvx = bpred_mx; pmvy = bpred_my;
        for( int i = 0; i < n_mvc; i++ )
        {
            int mx = mvcx[i], my = mvcy[i];
            if( ( !mx && !my ) || ( mx == pmvx && my == pmvy ) )
                continue;
            cx[nc] = iclip3( mx, 4 * B.fmin_x, 4 * B.fmax_x );
            cy[nc] = iclip3( my, 4 * B.fmin_y, 4 * B.fmax_y );
            nc++;
        }
        // round A: pmv + first three candidates
        int qx = sel4( g, pmvx, nc > 0 ? cx[0] : pmvx, nc > 1 ? cx[1] : pmvx, nc > 2 ? cx[2] : pmvx );
        int qy = sel4( g, pmvy, nc > 0 ? cy[0] : pmvy, nc > 1 ? cy[1] : pmvy, nc > 2 ? cy[2] : pmvy );
        int v = qpel_cost( P, B, wt, qx, qy, P.fpelcmp_satd );
        bpred_cost = GRP_COST( v, 0 );
        const int pmv_cost = bpred_cost;
        for( int i = 0; i < nc && i < 3; i++ )
        {
            int c = i == 0 ? GRP_COST( v, 1 ) : i == 1 ? GRP_COST( v, 2 ) : GRP_COST( v, 3 );
            if( c < bpred_cost ) { bpred_cost = c; bpred_mx = cx[i]; bpred_my = cy[i]; }
        }
        if( nc > 3 )
        {
            int v2 = qpel_cost( P, B, wt, cx[3], cy[3], P.fpelcmp_satd );
            int c = GRP_COST( v2, 0 );
            if( c < bpred_cost ) { bpred_cost = c; bpred_mx = cx[3]; bpred_my = cy[3]; }
        }
        bmx = ( bpred_mx + 2 ) >> 2;
        bmy = ( bpred_my + 2 ) >> 2;
        // round B: rounded best predictor (group 0) and the zero vector (group 1), applied in order
        const bool need_round = ( ( bpred_mx | bpred_my ) & 3 ) != 0;
        const bool need_zero = ( pmvx | pmvy ) && ( bmx | bmy );
        bcost = need_round ? COST_MAX_I : bpred_cost;
        if( need_round || need_zero )
        {
            int fx = g == 0 ? bmx : 0, fy = g == 0 ? bmy : 0;
            int v3 = fpel_cost( P, B, fx, fy, 1 );
            if( need_round ) bcost = GRP_COST( v3, 0 );
            if( need_zero )
            {
                int c = GRP_COST( v3, 1 );
                if( c < bcost ) { bcost = c; bmx = 0; bmy = 0; }
            }
        }
        if( !( pmvx | pmvy ) && pmv_cost < bcost )
        {
            bcost = pmv_cost; bmx = 0; bmy = 0;
        }
    }
    else
    {
        // predictor rounded to full-pel (me.c:276-318)
        bmx = iclip3( ( B.mvpx + 2 ) >> 2, B.fmin_x, B.fmax_x );
        bmy = iclip3( ( B.mvpy + 2 ) >> 2, B.fmin_y, B.fmax_y );
        pmvx = bmx; pmvy = bmy;
        for( int i = 0; i < n_mvc; i++ )
        {
            int mx = ( mvcx[i] + 2 ) >> 2, my = ( mvcy[i] + 2 ) >> 2;
            if( ( !mx && !my ) || ( mx == pmvx && my == pmvy ) )
                continue;
            cx[nc] = iclip3( mx, B.fmin_x, B.fmax_x );
            cy[nc] = iclip3( my, B.fmin_y, B.fmax_y );
            nc++;
        }
        int fx = sel4( g, pmvx, nc > 0 ? cx[0] : pmvx, nc > 1 ? cx[1] : pmvx, nc > 2 ? cx[2] : pmvx );
        int fy = sel4( g, pmvy, nc > 0 ? cy[0] : pmvy, nc > 1 ? cy[1] : pmvy, nc > 2 ? cy[2] : pmvy );
        int v = fpel_cost( P, B, fx, fy, g != 0 ); // the predictor itself carries no mv bits
        bcost = GRP_COST( v, 0 );
        for( int i = 0; i < nc && i < 3; i++ )
        {
            int c = i == 0 ? GRP_COST( v, 1 ) : i == 1 ? GRP_COST( v, 2 ) : GRP_COST( v, 3 );
            if( c < bcost ) { bcost = c; bmx = cx[i]; bmy = cy[i]; }
        }
        const bool need_zero = ( pmvx | pmvy ) != 0;
        if( nc > 3 || need_zero )
        {
            int gx = g == 0 && nc > 3 ? cx[3] : 0, gy = g == 0 && nc > 3 ? cy[3] : 0;
            int v2 = fpel_cost( P, B, gx, gy, 1 );
            if( nc > 3 )
            {
                int c = GRP_COST( v2, 0 );
                if( c < bcost ) { bcost = c; bmx = cx[3]; bmy = cy[3]; }
            }
            if( need_zero )
            {
                int c = GRP_COST( v2, 1 );
                if( c < bcost ) { bcost = c; bmx = 0; bmy = 0; }
            }
        }
    }

    if( P.me_method == X264HIP_ME_DIA )
    {
        // radius-1 diamond: up, down, left, right (me.c:322-342)
        const int ddx = sel4( g, 0, 0, -1, 1 ), ddy = sel4( g, -1, 1, 0, 0 );
        int iters = P.me_range;
        do
        {
            int v = fpel_cost( P, B, bmx + ddx, bmy + ddy, 1 );
            int best = -1;
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                int c = GRP_COST( v, k );
                if( c < bcost ) { bcost = c; best = k; }
            }
            if( best < 0 )
                break;
            bmx += best == 2 ? -1 : best == 3 ? 1 : 0;
            bmy += best == 0 ? -1 : best == 1 ? 1 : 0;
        } while( --iters && in_fpel_range( B, bmx, bmy ) );
    }
    else
    {
        // hexagon (me.c:344-420)
        int dir = -1;
        {
            int k0 = g, k1 = 4 + ( g & 1 );
            int v0 = fpel_cost( P, B, bmx + hex_dx( k0 ), bmy + hex_dy( k0 ), 1 );
            int v1 = fpel_cost( P, B, bmx + hex_dx( k1 ), bmy + hex_dy( k1 ), 1 );
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                int c = GRP_COST( v0, k );
                if( c < bcost ) { bcost = c; dir = k; }
            }
#pragma unroll
            for( int k = 0; k < 2; k++ )
            {
                int c = GRP_COST( v1, k );
                if( c < bcost ) { bcost = c; dir = 4 + k; }
            }
        }
        if( dir >= 0 )
        {
            bmx += hex_dx( dir ); bmy += hex_dy( dir );
            for( int i = ( P.me_range >> 1 ) - 1; i > 0 && in_fpel_range( B, bmx, bmy ); i-- )
            {
                int kd = mod6( dir + ( g > 2 ? 1 : g ) - 1 ); // groups 0,1,2 -> dir-1, dir, dir+1
                int v = fpel_cost( P, B, bmx + hex_dx( kd ), bmy + hex_dy( kd ), 1 );
                int best = -2;
#pragma unroll
                for( int k = 0; k < 3; k++ )
                {
                    int c = GRP_COST( v, k );
                    if( c < bcost ) { bcost = c; best = k - 1; }
                }
                if( best == -2 )
                    break;
                dir = mod6( dir + best );
                bmx += hex_dx( dir ); bmy += hex_dy( dir );
            }
        }
        // square refine: (0,-1) (0,1) (-1,0) (1,0) then (-1,-1) (-1,1) (1,-1) (1,1)
        {
            const int ax = sel4( g, 0, 0, -1, 1 ), ay = sel4( g, -1, 1, 0, 0 );
            const int bx2 = sel4( g, -1, -1, 1, 1 ), by2 = sel4( g, -1, 1, -1, 1 );
            int v0 = fpel_cost( P, B, bmx + ax, bmy + ay, 1 );
            int v1 = fpel_cost( P, B, bmx + bx2, bmy + by2, 1 );
            int best = -1;
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                int c = GRP_COST( v0, k );
                if( c < bcost ) { bcost = c; best = k; }
            }
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                int c = GRP_COST( v1, k );
                if( c < bcost ) { bcost = c; best = 4 + k; }
            }
            if( best >= 0 )
            {
                bmx += best == 2 ? -1 : best == 3 ? 1 : best >= 4 ? ( best < 6 ? -1 : 1 ) : 0;
                bmy += best == 0 ? -1 : best == 1 ? 1 : best >= 4 ? ( ( best & 1 ) ? 1 : -1 ) : 0;
            }
        }
    }

    // back to quarter-pel units (me.c:774-789)
    int mvx, mvy, cost;
    if( P.subpel_refine < 3 )
    {
        cost = bcost;
        if( bmx == pmvx && bmy == pmvy )
            cost += mv_bits( P, B, 4 * bmx, 4 * bmy );
        mvx = 4 * bmx; mvy = 4 * bmy;
    }
    else if( bpred_cost < bcost )
    {
        mvx = bpred_mx; mvy = bpred_my; cost = bpred_cost;
    }
    else
    {
        mvx = 4 * bmx; mvy = 4 * bmy; cost = bcost;
    }

    // ---- refine_subpel (me.c:865-992); lookahead rows of subpel_iterations: refine 2 -> hpel 1 / qpel 0,
    // refine 4 -> hpel 1 / qpel 1
    if( P.subpel_refine >= 2 )
    {
        const int qpel_iters = P.subpel_refine >= 4 ? 1 : 0;
        if( P.subpel_refine < 3 )
        {
            int mx = iclip3( B.mvpx, B.smin_x + 2, B.smax_x - 2 );
            int my = iclip3( B.mvpy, B.smin_y + 2, B.smax_y - 2 );
            if( mx != mvx || my != mvy )
            {
                int v = qpel_cost( P, B, wt, mx, my, P.fpelcmp_satd );
                int c = GRP_COST( v, 0 );
                if( c < cost ) { cost = c; mvx = mx; mvy = my; }
            }
        }
        {
            // half-pel diamond, one iteration: up, down, left, right
            const int hx = sel4( g, 0, 0, -2, 2 ), hy = sel4( g, -2, 2, 0, 0 );
            int v = qpel_cost( P, B, wt, mvx + hx, mvy + hy, P.fpelcmp_satd );
            int best = -1;
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                int c = GRP_COST( v, k );
                if( c < cost ) { cost = c; best = k; }
            }
            if( best >= 0 )
            {
                mvx += best == 2 ? -2 : best == 3 ? 2 : 0;
                mvy += best == 0 ? -2 : best == 1 ? 2 : 0;
            }
        }
        if( P.mbcmp_satd != P.fpelcmp_satd )
        {
            int v = qpel_cost( P, B, wt, mvx, mvy, P.mbcmp_satd );
            cost = GRP_COST( v, 0 );
        }
        int bdir = -1;
        for( int i = qpel_iters; i > 0; i-- )
        {
            if( mvy <= B.smin_y || mvy >= B.smax_y || mvx <= B.smin_x || mvx >= B.smax_x )
                break;
            const int odir = bdir, omx = mvx, omy = mvy;
            const int qx = sel4( g, 0, 0, -1, 1 ), qy = sel4( g, -1, 1, 0, 0 );
            int v = qpel_cost( P, B, wt, omx + qx, omy + qy, P.mbcmp_satd );
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                if( ( k ^ 1 ) == odir )
                    continue;
                int c = GRP_COST( v, k );
                if( c < cost )
                {
                    cost = c; bdir = k;
                    mvx = omx + ( k == 2 ? -1 : k == 3 ? 1 : 0 );
                    mvy = omy + ( k == 0 ? -1 : k == 1 ? 1 : 0 );
                }
            }
            if( mvx == omx && mvy == omy )
                break;
        }
    }
    out_mvx = mvx; out_mvy = mvy; out_cost = cost;
}

// ---- persistent row-wave search kernel ---------------------------------------------------------------
// One wave64 per (search, block row).  Rows are claimed bottom-up through a ticket counter, so the wave a
// row depends on (the row below of the same search) always holds an earlier ticket and is running or done:
// the in-kernel waits cannot deadlock whatever the dispatch order.  A row trails the row below by two
// blocks: block (x,y) needs the final vectors of (x+1,y) [own registers] and of (x-1..x+1, y+1), which the
// lower wave publishes as self-validating 8-byte granules { mv, tag } with agent-scope relaxed atomics
// (write-through `sc1` stores / L1-bypassing `sc1` loads, no fences; MI355X guide, G16 form R2).
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
};

// The searches of a launch are split into ME_QUEUES contiguous groups (neighbouring frames), one ticket counter
// each, and a wave serves the group of the XCD it runs on: the planes of a group then live in ONE of the eight
// L2s instead of being fetched into all of them (a row band of every frame of the launch is active at any time,
// far more than a 4 MB L2 holds).  The XCD id is used for locality only: a wave whose own group has no rows left
// takes rows from the other groups, so every row is claimed whatever the dispatcher's workgroup placement is,
// and the order inside a group still guarantees that the row below holds an earlier ticket.
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

template <typename T>
__global__ __launch_bounds__( 64 ) __attribute__( ( amdgpu_waves_per_eu( 8, 8 ) ) ) void me_rows_kernel( LaP P, const SearchDesc<T> *descs, MeQueues Q,
                                                                    unsigned *tickets /* [ME_QUEUES * ME_QUEUE_STRIDE] */,
                                                                    unsigned *err_host /* pinned sticky timeout flag */, unsigned spin_limit )
{
    const int lane = lane_id();
    // the ticket is wave-uniform: fetch it on lane 0 and broadcast through an SGPR so that the row index, the
    // descriptor and everything derived from them stay scalar
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
        if( t < (unsigned)( n_q * P.mb_h ) )
        {
            j = t / n_q;
            s = Q.base[q] + ( t - j * n_q );
        }
    }
    if( s < 0 )
        return;
    const int by = P.mb_h - 1 - j;
    const SearchDesc<T> D = descs[s];
    const int W = P.mb_w, H = P.mb_h;

    MeBlk<T> B;
    B.g = lane >> 4;
    int row_off; // this lane's 4 samples inside an 8x8 block
    {
        int l = lane & 15, q = l >> 2;
        row_off = ( ( q >> 1 ) * 4 + ( l & 3 ) ) * P.stride + ( q & 1 ) * 4;
    }
    const int border = LA_PAD * P.stride + LA_PAD;
    const T *fbase = D.fenc0 - border;
    B.rbase = D.ref0 - border;
    B.wbase = D.wt.on ? D.refw - border : B.rbase;
    const int tab_centre = 2 * 4 * P.mv_range;
    B.tab = P.cost_mv - tab_centre;
    const int range = 2 * P.mv_range;
    B.smin_y = imax2( 4 * ( -8 * by - 12 ), -range );
    B.smax_y = imin2( 4 * ( 8 * ( H - by - 1 ) + 12 ), range - 1 );
    B.fmin_y = B.smin_y >> 2;
    B.fmax_y = B.smax_y >> 2;

    // Hb: end row of the band this row belongs to (slicetype.c:917-918); rows of one band do not see the vectors of the
    // band below (slicetype.c:668).  One band: Hb == H.
    int Hb = H;
    for( int sl = P.n_slices - 1; sl >= 1; sl-- )
    {
        const int start = ( H * sl + P.n_slices / 2 ) / P.n_slices;
        if( by < start )
            Hb = start;
    }
    // blocks slicetype_slice_cost never visits (slicetype.c:823-833): their vectors stay zero (frame.c:283-285) and are
    // marked ready at once, nothing reads their costs
    const int e = P.no_edges;
    if( e )
    {
        const unsigned long long unvisited = (unsigned long long)D.tag << 32;
        const bool whole_row = by == 0 || by == H - 1;
        for( int bx = lane; bx < W; bx += 64 )
            if( whole_row || bx == 0 || bx == W - 1 )
            {
                __hip_atomic_store( D.mvq + by * W + bx, unvisited, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                D.costs[by * W + bx] = 0;
            }
        if( whole_row )
            return;
    }

    int right_x = 0, right_y = 0;
    for( int bx = W - 1 - e; bx >= e; bx-- )
    {
        const int xy = by * W + bx;
        B.lane_off = border + 8 * ( by * P.stride + bx ) + row_off;
        int nbx[3] = { 0, 0, 0 }, nby[3] = { 0, 0, 0 }; // below, below-left, below-right
        if( by < Hb - 1 )
        {
            unsigned long long gq = 0;
            const int nb = lane == 1 ? ( bx > 0 ? -1 : 0 ) : lane == 2 ? ( bx < W - 1 ? 1 : 0 ) : 0;
            const unsigned long long *gp = D.mvq + ( xy + W + nb );
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
            }
            const int lo = (int)(unsigned)gq;
#pragma unroll
            for( int k = 0; k < 3; k++ )
            {
                int w = __builtin_amdgcn_readlane( lo, k );
                nbx[k] = (int)(short)( w & 0xFFFF );
                nby[k] = w >> 16;
            }
        }
        // predictor list in the reference's order: right, below, below-left, below-right
        int mvcx[4] = { 0, 0, 0, 0 }, mvcy[4] = { 0, 0, 0, 0 }, n = 0;
        if( bx < W - 1 ) { mvcx[n] = right_x; mvcy[n] = right_y; n++; }
        if( by < Hb - 1 )
        {
            mvcx[n] = nbx[0]; mvcy[n] = nby[0]; n++;
            if( bx > 0 ) { mvcx[n] = nbx[1]; mvcy[n] = nby[1]; n++; }
            if( bx < W - 1 ) { mvcx[n] = nbx[2]; mvcy[n] = nby[2]; n++; }
        }
        if( n <= 1 ) { B.mvpx = mvcx[0]; B.mvpy = mvcy[0]; }
        else
        {
            B.mvpx = median3i( mvcx[0], mvcx[1], mvcx[2] );
            B.mvpy = median3i( mvcy[0], mvcy[1], mvcy[2] );
        }
        B.smin_x = imax2( 4 * ( -8 * bx - 12 ), -range );
        B.smax_x = imin2( 4 * ( 8 * ( W - bx - 1 ) + 12 ), range - 1 );
        B.fmin_x = B.smin_x >> 2;
        B.fmax_x = B.smax_x >> 2;
        B.tab_x = tab_centre - B.mvpx;
        B.tab_y = tab_centre - B.mvpy;
        B.f = load_px4_at( fbase, B.lane_off );

        int mvx = 0, mvy = 0, cost = 0;
        bool done = false;
        if( !( B.mvpx | B.mvpy ) )
        {
            // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692)
            const Px4 r = load_px4_at( B.rbase, B.lane_off );
            cost = GRP_COST( block_cost8x8<T>( B.f, r, P.mbcmp_satd ), 0 );
            done = cost < 64;
        }
        if( !done )
        {
            me_block( P, B, D.wt, n, mvcx, mvcy, mvx, mvy, cost );
            cost -= P.cost_mv[0];
            if( mvx | mvy )
                cost += 5 * P.lambda;
        }
        if( lane == 0 )
        {
            unsigned long long gv = ( (unsigned long long)D.tag << 32 ) | (unsigned)( ( mvx & 0xFFFF ) | ( mvy << 16 ) );
            __hip_atomic_store( D.mvq + xy, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
            D.costs[xy] = cost;
        }
        right_x = mvx; right_y = mvy;
    }
}
