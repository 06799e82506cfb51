// me_search8.h -- the lookahead motion search, eight block rows per wave64: one 8x8 block per 8-lane group, one block row per lane.
//
// Same behaviour and the same decision logic (me_logic.h) as me_rows_kernel of me_search.h; what changes is the geometry.  The
// search is bound by vector-ALU issue once the CU is full (experiments/README.md, "what bounds the search", and the issue-rate
// table measured with experiments/gen_valu_rate.py: everything but plain add/and/or/mov/cndmask costs 4 cycles per wave64), and
// most of those instructions are the selection logic, address arithmetic and reductions, whose count per wave does not depend on
// how many pixels a lane holds.  With 8 pixels per lane (one row of the block: two dwords, the byte-wise v_sad_u8 / v_lerp_u8
// take them as they are) a block needs 8 lanes instead of 16, the wave carries eight blocks through the same instruction stream
// instead of four, and the cost reduction is three DPP steps instead of four.  SATD is the sum of four 4x4 transforms: the two
// column halves of a lane go through the quad-wide transform of device_common.h one after the other.
//
// Reference samples come from the STRIP copy of the half-pel planes (written next to the row-major planes by lowres_kernel):
// strip k of a plane holds columns 8k .. 8k+15 of every row, 16 samples per row, rows one after the other.  The eight rows of a
// block candidate are then 128 consecutive bytes (2-3 cache lines) instead of one line per row, and because every column
// exists in two strips, eight samples starting at any column -- and the quarter-pel partner one column to the right -- are one
// unaligned load inside one strip.  The row-major kernel asks the L1 for ~9 lines per block and candidate, which saturates the
// texture cache pipe as soon as the vector ALU stops being the limit (experiments/README.md).
//
// A wave owns ME8_ROWS = 8 consecutive block rows; group g (lanes 8g..8g+7) walks row y0 - g from right to left two blocks behind
// the group below it and finds its three lower neighbours in the registers of group g-1; only group 0 waits for another wave
// (granules { mv, tag } published by the top row of the wave below, as in me_search.h).  W + 14 steps per wave.
#pragma once
#include "me_search.h"

#define ME8_ROWS 8
#define DPP_ROW_HALF_MIRROR 0x141 // lane i of every 8 reads lane 7 - i

struct Px8
{
    Px4 lo, hi; // samples 0..3 and 4..7 of this lane's row
};

__device__ __forceinline__ Px8 load_px8_at( const uint8_t *ubase, int elem_off )
{
    const uint2 w = gload_u64( ubase, (unsigned)elem_off ); // one global_load_dwordx2 at any byte alignment
    Px8 r;
    r.lo = px4_from_raw( w.x ); r.hi = px4_from_raw( w.y );
    return r;
}
__device__ __forceinline__ Px8 load_px8_at( const uint16_t *ubase, int elem_off )
{
    Px8 r;
    r.lo = load_px4_at( ubase, elem_off ); r.hi = load_px4_at( ubase, elem_off + 4 );
    return r;
}
// element offset inside a plane's strips of the 8 samples starting at padded column c of the row whose strip-row offset is row16
__device__ __forceinline__ int strip_off( int c, int row16, int strip_elems )
{
    return mad24( c >> 3, strip_elems, ( c & 7 ) + row16 );
}
// the quarter-pel samples of device_common.h's qpel_px4_at, eight per lane, out of the strip copy: sbase = strips of plane 0,
// cx0 / row16 = padded column of the block and strip-row offset of this lane's row at zero displacement
template <typename T>
__device__ __forceinline__ Px8 qpel_px8_strips( const T *sbase, int plane_elems, int strip_elems, int cx0, int row16, int mvx, int mvy )
{
    const int fx = mvx & 3, fy = mvy & 3;
    const int sh = 2 * ( fx | ( fy << 2 ) );
    const unsigned pa = ( 0x54FE5454u >> sh ) & 3u, pb = ( 0xBABABA10u >> sh ) & 3u; // plane pair of the phase (device_common.h)
    const int o = strip_off( cx0 + ( mvx >> 2 ), row16 + ( ( mvy >> 2 ) << 4 ), strip_elems );
    // a plane's strips take twice the plane; the partner column is in the same strip (offsets 0..8 + 8 samples <= 16)
    const int oa = ( (int)__umul24( pa, (unsigned)plane_elems ) << 1 ) + o + ( fy == 3 ? 16 : 0 );
    const int ob = ( (int)__umul24( pb, (unsigned)plane_elems ) << 1 ) + o + ( fx == 3 );
    const Px8 a = load_px8_at( sbase, oa ), b = load_px8_at( sbase, ob );
    Px8 r;
    r.lo = avg_px4( a.lo, b.lo, (const T *)nullptr ); r.hi = avg_px4( a.hi, b.hi, (const T *)nullptr );
    return r;
}
// sum over the 8 lanes of a group, result in every lane of the group
__device__ __forceinline__ int reduce8( int v )
{
    v = reduce_quad( v );
    v += dpp_mov<DPP_ROW_HALF_MIRROR>( v );
    return v;
}
__device__ __forceinline__ int sad_partial_px8( const Px8 &f, const Px8 &r, const uint8_t * )
{
    return (int)__builtin_amdgcn_sad_u8( f.hi.raw, r.hi.raw, __builtin_amdgcn_sad_u8( f.lo.raw, r.lo.raw, 0u ) );
}
__device__ __forceinline__ int sad_partial_px8( const Px8 &f, const Px8 &r, const uint16_t * )
{
    return sad_partial16( f.lo, r.lo ) + sad_partial16( f.hi, r.hi );
}
// cost of the 8x8 block this 8-lane group holds, in every lane of the group
template <typename T>
__device__ __forceinline__ int block_cost8( const Px8 &f, const Px8 &r, int use_satd )
{
    if( use_satd )
        return reduce8( satd_partial_px4( f.lo, r.lo ) + satd_partial_px4( f.hi, r.hi ) ) >> 1;
    return reduce8( sad_partial_px8( f, r, (const T *)nullptr ) );
}

// the evaluator of me_logic.h on the 8-lane geometry
template <typename T, int LDS_TAB, int WEIGHTED>
struct GroupEval8
{
    const uint16_t *lds_tab; // this wave's window of the mv cost table: entry ME_TAB_HALF + d is the cost of difference d
    const T *sbase;          // wave-uniform: strips of the reference frame's four planes (unweighted)
    const T *wsbase;         // wave-uniform: strips read by full-pel candidates (weighted copy of plane 0, or sbase)
    const uint16_t *tab;     // wave-uniform: first entry of the cost_mv table in memory
    int plane_elems, strip_elems, pixel_max;
    int fpelcmp_satd;
    WtD wt;
    int cx0, row16;          // padded column of the block, strip-row offset of this lane's row, both at zero displacement
    int tab_x, tab_y;
    Px8 f;                   // this lane's 8 source pixels

    __device__ __forceinline__ int bits( int qx, int qy ) const
    {
        if( LDS_TAB )
            return lds_tab[qx + tab_x] + lds_tab[qy + tab_y];
        return gload_u16( tab, 2u * (unsigned)( qx + tab_x ) ) + gload_u16( tab, 2u * (unsigned)( qy + tab_y ) );
    }
    __device__ __forceinline__ int fpel( int x, int y ) const
    {
        const Px8 r = load_px8_at( WEIGHTED ? wsbase : sbase, strip_off( cx0 + x, row16 + ( y << 4 ), strip_elems ) );
        return block_cost8<T>( f, r, fpelcmp_satd );
    }
    __device__ __forceinline__ int qpel( int qx, int qy, int use_satd ) const
    {
        Px8 r = qpel_px8_strips( sbase, plane_elems, strip_elems, cx0, row16, qx, qy );
        if( WEIGHTED )
        {
            r.lo = weight_px4<T>( r.lo, wt, pixel_max ); r.hi = weight_px4<T>( r.hi, wt, pixel_max );
        }
        return block_cost8<T>( f, r, use_satd );
    }
    __device__ __forceinline__ bool any( bool c ) const { return __builtin_amdgcn_ballot_w64( c ) != 0ull; }
#ifdef ME_PROFILE
    __device__ __forceinline__ void mark( int ) {}
#endif
};

// value of the same lane position one group (8 lanes) further down; group 0 gets garbage it never uses
__device__ __forceinline__ int from_group_below8( int v, int lane )
{
    return __builtin_amdgcn_ds_bpermute( ( ( lane - 8 ) & 63 ) << 2, v );
}

// MODE as in me_rows_kernel; WEIGHTED: every search of the launch reads a weighted copy of its reference (D.refw, D.wt)
template <typename T, int HEX, int MODE, int WEIGHTED>
__global__ __launch_bounds__( 64, ME_MIN_WAVES ) void me_rows8_kernel( LaP P, const SearchDesc<T> *descs, MeQueues Q, unsigned *tickets /* [ME_QUEUES * ME_QUEUE_STRIDE] */,
                                                                        unsigned *err_host /* pinned sticky timeout flag */, unsigned spin_limit )
{
    const int lane = lane_id();
    const int W = P.mb_w, H = P.mb_h;
    const int n_rowgroups = ( H + ME8_ROWS - 1 ) / ME8_ROWS;
    // the ticket is wave-uniform: fetched on lane 0 and broadcast through an SGPR, so that the row group, the descriptor and
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
    const int g = lane >> 3;
    const int by0 = H - 1 - ME8_ROWS * j; // row of group 0 (scalar)
    const int by = by0 - g;               // this group's row
    const bool row_ok = by >= 0;

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
    const int border = LA_PAD * P.stride + LA_PAD;
    const T *fbase = D.fenc0 - border;
    const T *sbase = D.ref_strips;
    const T *wsbase = WEIGHTED ? D.refw_strips : sbase;
    const int strip_elems = ( P.plane_elems / P.stride ) * 16; // rows of the padded plane x 16 samples
    const int tab_centre = 2 * 4 * P.mv_range;
    const int row_off = ( lane & 7 ) * P.stride; // this lane's row inside an 8x8 block
    // end row of the band this row belongs to (slicetype.c:917-918): rows of one band do not see the vectors of the band below
    int band_end = H;
    for( int sl = P.n_slices - 1; sl >= 1; sl-- )
    {
        const int start = ( H * sl + P.n_slices / 2 ) / P.n_slices;
        if( by < start )
            band_end = start;
    }
    const bool has_below = row_ok && by < band_end - 1;
    // group 0's row is the only one whose lower neighbours live in another wave
    const bool below_is_remote = (bool)__builtin_amdgcn_readfirstlane( (int)has_below );
    const int zero_bits = P.cost_mv[0];

    int r1 = 0, r2 = 0, r3 = 0; // packed vectors this group found in the last three steps
    const int n_steps = W + 2 * ( ME8_ROWS - 1 );
    for( int t = 0; t < n_steps; t++ )
    {
        const int bx = W - 1 - ( t - 2 * g );
        const bool active = row_ok && bx >= 0 && bx < W;
        // the row below: (x-1, y+1), (x, y+1), (x+1, y+1) are what the group below found one, two and three steps ago
        int below_left = from_group_below8( r1, lane ), below = from_group_below8( r2, lane ), below_right = from_group_below8( r3, lane );
        {
            const int bx0 = W - 1 - t;
            if( bx0 >= 0 && below_is_remote )
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
                }
                const int lo = (int)(unsigned)gq;
                const int w0 = __builtin_amdgcn_readlane( lo, 0 ), w1 = __builtin_amdgcn_readlane( lo, 1 ), w2 = __builtin_amdgcn_readlane( lo, 2 );
                if( g == 0 ) { below = w0; below_left = w1; below_right = w2; }
            }
        }
        int mvx = 0, mvy = 0, cost = 0;
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
                const int lane_off = border + 8 * ( by * P.stride + bx ) + row_off;
                const int cx0 = 8 * bx + LA_PAD, row16 = ( 8 * by + ( lane & 7 ) + LA_PAD ) << 4;
                const Px8 f = load_px8_at( fbase, lane_off );
                bool done = false;
                if( !( mvpx | mvpy ) )
                {
                    // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692)
                    const Px8 r = load_px8_at( sbase, strip_off( cx0, row16, strip_elems ) );
                    cost = block_cost8<T>( f, r, C.mbcmp_satd );
                    done = cost < 64;
                }
                if( !done )
                {
                    // how far from the predictor can a candidate of this block be?  (me_search.h)
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
                        GroupEval8<T, 1, WEIGHTED> ev;
                        ev.lds_tab = tab_window; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = nullptr; ev.plane_elems = P.plane_elems;
                        ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = D.wt;
                        ev.cx0 = cx0; ev.row16 = row16; ev.f = f;
                        ev.tab_x = ME_TAB_HALF - mvpx; ev.tab_y = ME_TAB_HALF - mvpy;
                        melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                    }
                    else
                    {
                        GroupEval8<T, 0, WEIGHTED> ev;
                        ev.lds_tab = nullptr; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = P.cost_mv - tab_centre; ev.plane_elems = P.plane_elems;
                        ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = D.wt;
                        ev.cx0 = cx0; ev.row16 = row16; ev.f = f;
                        ev.tab_x = tab_centre - mvpx; ev.tab_y = tab_centre - mvpy;
                        melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                    }
                    cost -= zero_bits;
                    if( mvx | mvy )
                        cost += 5 * P.lambda;
                }
            }
            // blocks slicetype_slice_cost never visits (slicetype.c:823-833) keep zero vectors (frame.c:283-285)
            if( ( lane & 7 ) == 0 )
            {
                const unsigned long long gv = ( (unsigned long long)D.tag << 32 ) | (unsigned)( ( mvx & 0xFFFF ) | ( mvy << 16 ) );
                if( g == ME8_ROWS - 1 )
                    __hip_atomic_store( D.mvq + xy, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); // read by the wave above
                else
                    D.mvq[xy] = gv;
                D.costs[xy] = cost;
            }
        }
        r3 = r2; r2 = r1;
        r1 = ( mvx & 0xFFFF ) | ( mvy << 16 );
    }
}
