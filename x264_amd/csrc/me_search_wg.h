// me_search_wg.h -- the lookahead motion search with the reference window staged in LDS, shared by the searches that read it.
//
// me_rows_kernel (me_search.h) is bound by the latency of the dependent candidate rounds of a step, and every round is a trip to
// the L2: the 32 KB L1 of a CU holds 16 lines per resident wave, a wave touches ~200 per step (profiles/r02_search_pmc.json:
// 55 % L1 misses, waves parked on s_waitcnt 73 % of their life).  A window of the reference in LDS makes a round an LDS read -- but
// a window per wave costs ~28 KB, i.e. 5 waves per CU, too few to keep the vector ALU busy.  The way out is that a frame is the
// reference of up to 2 x (bframes + 1) searches (every source frame within reach, both lists): the searches of a launch are grouped
// by reference frame, a workgroup of MEW_WAVES waves takes the same row group of up to MEW_WAVES searches that share a reference,
// and all of them read ONE window.  The waves advance in lock step (two barriers per step), each exactly as a wave of
// me_rows_kernel does (four block rows, one block per 16-lane group, melogic::search per group); a 8-sample column strip of the
// four half-pel planes is requested one step ahead by the whole workgroup and written into the circular window at the next
// step's start, so the plane data is fetched from memory once per row group instead of once per candidate.  A candidate whose
// samples lie outside the window (vectors beyond ~ +-12 lowres samples) is read from memory as before (wave-uniform choice per
// candidate): results never depend on which path served a candidate.  8-bit, unweighted searches; everything else runs
// me_rows_kernel.
#pragma once
#include "me_search.h"

#ifndef MEW_WAVES
#define MEW_WAVES 8 // searches per workgroup (they share a reference frame)
#endif
#define MEW_SLOTS 16                                 // 8-sample column slots of the circular window
#define MEW_COLS ( 8 * MEW_SLOTS )                   // 128 samples
#define MEW_MARGIN_Y 12
#define MEW_ROWS ( 8 * ME_ROWS + 2 * MEW_MARGIN_Y )  // 56 rows: the four block rows of the workgroup + 12 above and below
#define MEW_LEFT 2                                   // the strip of a step lies this many slots left of group 0's block (16 samples)
#define MEW_PITCH ( MEW_COLS + 8 )                   // bytes per window row: the samples of slot 0 are repeated behind slot 15, so that four samples starting
                                                     // at any column are one (unaligned) 32-bit LDS read; 34 dwords per row also spread the 8 rows x 2 halves of a
                                                     // block over 16 different banks
#define MEW_PLANE ( MEW_ROWS * MEW_PITCH )

struct RefGroup
{
    int n;                  // searches in the group (1 .. MEW_WAVES)
    int search[MEW_WAVES];  // indices into the launch's SearchDesc table
};

// evaluator of me_logic.h: 16-lane block geometry, samples from the LDS window where it covers them
struct WinEval
{
    const uint8_t *rbase;    // reference frame, four planes (memory path)
    const uint16_t *tab;     // cost table in memory (blocks with far predictors)
    const uint16_t *lds_tab; // cost table window in LDS
    const uint8_t *win;      // LDS: [4][MEW_ROWS][MEW_PITCH]
    int use_lds_tab;
    int plane_elems, stride, fpelcmp_satd;
    int lane_off;            // element offset of this lane's samples at zero displacement (memory path)
    int px, py;              // picture position of this lane's first sample at zero displacement
    int x_lo, x_hi, y_lo, y_hi, y0; // window coverage (picture coordinates), y0 = picture row of window row 0
    int tab_x, tab_y;
    Px4 f;

#ifdef ME_PROFILE
    __device__ __forceinline__ void mark( int ) {}
#endif
    __device__ __forceinline__ bool any( bool c ) const { return __builtin_amdgcn_ballot_w64( c ) != 0ull; }
    __device__ __forceinline__ int bits( int qx, int qy ) const
    {
        if( use_lds_tab )
            return lds_tab[qx + tab_x] + lds_tab[qy + tab_y];
        return gload_u16( tab, 2u * (unsigned)( qx + tab_x ) ) + gload_u16( tab, 2u * (unsigned)( qy + tab_y ) );
    }
    // four samples of plane p starting at picture position (x, y) out of the window
    __device__ __forceinline__ uint32_t win4( int p, int x, int y ) const
    {
        uint32_t w;
        __builtin_memcpy( &w, win + ( p * MEW_ROWS + ( y - y0 ) ) * MEW_PITCH + ( ( x + 4096 ) & ( MEW_COLS - 1 ) ), 4 ); // ds_read_b32, any alignment
        return w;
    }
    __device__ __forceinline__ int fpel( int x, int y ) const
    {
        const int sx = px + x, sy = py + y;
        const bool out = sx < x_lo || sx + 4 > x_hi || sy < y_lo || sy >= y_hi;
        Px4 r;
        if( !any( out ) )
            r = px4_from_raw( win4( 0, sx, sy ) );
        else
            r = load_px4_at( rbase, lane_off + mad24( y, stride, x ) );
        return block_cost8x8<uint8_t>( f, r, fpelcmp_satd );
    }
    __device__ __forceinline__ int qpel( int qx, int qy, int use_satd ) const
    {
        const int fx = qx & 3, fy = qy & 3;
        const int sh = 2 * ( fx | ( fy << 2 ) );
        const int pa = ( 0x54FE5454u >> sh ) & 3u, pb = ( 0xBABABA10u >> sh ) & 3u; // plane pair of the phase (device_common.h)
        const int ix = px + ( qx >> 2 ), iy = py + ( qy >> 2 );
        const bool out = ix < x_lo || ix + 5 > x_hi || iy < y_lo || iy + 1 >= y_hi;
        Px4 r;
        if( !any( out ) )
        {
            const uint32_t a = win4( pa, ix, iy + ( fy == 3 ) ), b = win4( pb, ix + ( fx == 3 ), iy );
            r = px4_from_raw( __builtin_amdgcn_lerp( a, b, 0x01010101u ) );
        }
        else
            r = qpel_px4_at( rbase, plane_elems, stride, lane_off, qx, qy );
        return block_cost8x8<uint8_t>( f, r, use_satd );
    }
};

// 8-byte pieces of the window: piece i of a strip = plane i / MEW_ROWS, window row i % MEW_ROWS, the 8 samples of column slot `col`
__device__ __forceinline__ uint2 mew_load_piece( const uint8_t *rbase, int plane_elems, int stride, int border, int y0, int lh, int i, int col )
{
    const int p = i / MEW_ROWS, r = i - p * MEW_ROWS;
    const int y = iclip3( y0 + r, -LA_PAD, lh + LA_PAD - 1 ); // rows outside the padded plane are never read by a legal candidate
    return gload_u64( rbase, (unsigned)( p * plane_elems + border + y * stride + col ) );
}
__device__ __forceinline__ void mew_store_piece( uint8_t *win, int i, int col, uint2 v )
{
    const int p = i / MEW_ROWS, r = i - p * MEW_ROWS;
    const int c = ( col + 4096 ) & ( MEW_COLS - 1 );
    *(uint2 *)( win + ( p * MEW_ROWS + r ) * MEW_PITCH + c ) = v;
    if( !c )
        *(uint2 *)( win + ( p * MEW_ROWS + r ) * MEW_PITCH + MEW_COLS ) = v;
}

template <int HEX, int MODE>
__global__ __launch_bounds__( 64 * MEW_WAVES, 4 ) void me_rows_wg_kernel( LaP P, const SearchDesc<uint8_t> *descs, const RefGroup *groups, MeQueues Q /* over groups */,
                                                                           unsigned *tickets, unsigned *err_host, unsigned spin_limit,
                                                                           unsigned long long *prof /* ME_PROFILE builds */ )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t win[4 * MEW_PLANE];
    __shared__ uint16_t tab_window[2 * ME_TAB_HALF];
    __shared__ int sh_ticket[2];
    const int lane = lane_id(), wave = threadIdx.x >> 6, tid = threadIdx.x;
#ifdef ME_PROFILE
    unsigned long long pf_wait = 0, pf_bar = 0, pf_search = 0, pf_spins = 0, pf_steps = 0;
    const unsigned long long pf_begin = __builtin_amdgcn_s_memtime();
#endif
    const int W = P.mb_w, H = P.mb_h, lw = 8 * W, lh = 8 * H;
    const int n_rowgroups = ( H + ME_ROWS - 1 ) / ME_ROWS;
    if( tid == 0 )
    {
        const int home = xcc_id();
        int j = 0, gi = -1;
        for( int k = 0; k < ME_QUEUES && gi < 0; k++ )
        {
            const int q = ( home + k ) & ( ME_QUEUES - 1 );
            const int n_q = Q.base[q + 1] - Q.base[q];
            if( !n_q )
                continue;
            const unsigned t = atomicAdd( &tickets[q * ME_QUEUE_STRIDE], 1u );
            if( t < (unsigned)( n_q * n_rowgroups ) )
            {
                j = t / n_q;
                gi = Q.base[q] + ( t - j * n_q );
            }
        }
        sh_ticket[0] = j; sh_ticket[1] = gi;
    }
    {
        const int centre = 2 * 4 * P.mv_range;
        for( int i = tid; i < 2 * ME_TAB_HALF; i += 64 * MEW_WAVES )
        {
            const int d = i - ME_TAB_HALF;
            tab_window[i] = d >= -centre && d <= centre ? P.cost_mv[d] : (uint16_t)0;
        }
    }
    __syncthreads();
    const int j = sh_ticket[0], gi = sh_ticket[1];
    if( gi < 0 )
        return; // the whole workgroup
    const bool has_search = wave < groups[gi].n;
    const SearchDesc<uint8_t> D = descs[groups[gi].search[has_search ? wave : 0]];
    const int g = lane >> 4;
    const int by0 = H - 1 - ME_ROWS * j;
    const int by = by0 - g;
    const bool row_ok = has_search && by >= 0;

    MeCfg C;
    C.hex = HEX; C.me_range = P.me_range;
    C.refine4 = MODE == 3 ? P.subpel_refine >= 3 : MODE >= 1;
    C.mbcmp_satd = MODE == 3 ? P.mbcmp_satd : MODE >= 1;
    C.fpelcmp_satd = MODE == 3 ? P.fpelcmp_satd : MODE == 2;
    const int border = LA_PAD * P.stride + LA_PAD;
    const uint8_t *fbase = D.fenc0 - border;
    const uint8_t *rbase = D.ref0 - border; // the same frame for every wave of the workgroup
    const int tab_centre = 2 * 4 * P.mv_range;
    WinEval ev;
    ev.rbase = rbase; ev.tab = P.cost_mv - tab_centre; ev.lds_tab = tab_window; ev.win = win;
    ev.plane_elems = P.plane_elems; ev.stride = P.stride; ev.fpelcmp_satd = C.fpelcmp_satd;
    const int l = lane & 15, q4 = l >> 2;
    const int lrow = ( q4 >> 1 ) * 4 + ( l & 3 ), lcol = ( q4 & 1 ) * 4; // this lane's samples inside an 8x8 block
    const int row_off = lrow * P.stride + lcol;
    int band_end = H;
    for( int sl = P.n_slices - 1; sl >= 1; sl-- )
    {
        const int start = ( H * sl + P.n_slices / 2 ) / P.n_slices;
        if( by < start )
            band_end = start;
    }
    const bool has_below = row_ok && by < band_end - 1;
    const int zero_bits = P.cost_mv[0];

    // the window: rows y0 .. y0 + MEW_ROWS - 1 of the picture, columns in 8-sample slots, circular
    const int y0 = 8 * ( by0 - ( ME_ROWS - 1 ) ) - MEW_MARGIN_Y;
    ev.y0 = y0;
    ev.y_lo = imax2( y0, -LA_PAD ); ev.y_hi = imin2( y0 + MEW_ROWS, lh + LA_PAD );
    const int n_pieces = 4 * MEW_ROWS;
    // first fill: the 16 slots in front of step 0 (columns 8 (W-1-MEW_LEFT) ... ), as far as the padded plane reaches
    for( int k = tid; k < n_pieces * MEW_SLOTS; k += 64 * MEW_WAVES )
    {
        const int slot = k / n_pieces, i = k - slot * n_pieces;
        const int col = 8 * ( W - 1 - MEW_LEFT + slot );
        if( col >= -LA_PAD && col + 8 <= lw + LA_PAD )
            mew_store_piece( win, i, col, mew_load_piece( rbase, P.plane_elems, P.stride, border, y0, lh, i, col ) );
    }
    bool failed = false;
    int r1 = 0, r2 = 0, r3 = 0;
    const int n_steps = W + 2 * ( ME_ROWS - 1 );
    uint2 strip = make_uint2( 0, 0 ); // the piece this lane fetched for the next step
    for( int t = 0; t < n_steps; t++ )
    {
        const int bx0 = W - 1 - t; // group 0's block column in this step (may be negative once it has finished)
#ifdef ME_PROFILE
        const unsigned long long pf_t0 = __builtin_amdgcn_s_memtime();
#endif
        // 1. the strip requested during the previous step goes into the window (nobody reads the window between the barriers)
        __syncthreads();
        {
            const int col = 8 * ( bx0 - MEW_LEFT );
            if( t > 0 && tid < n_pieces && col >= -LA_PAD && col + 8 <= lw + LA_PAD )
                mew_store_piece( win, tid, col, strip );
        }
        __syncthreads();
        ev.x_lo = imax2( 8 * ( bx0 - MEW_LEFT ), -LA_PAD );
        ev.x_hi = imin2( 8 * ( bx0 - MEW_LEFT + MEW_SLOTS ), lw + LA_PAD );
        // 2. request the next step's strip: it has the whole step to arrive
        {
            const int col = 8 * ( bx0 - 1 - MEW_LEFT );
            if( t + 1 < n_steps && tid < n_pieces && col >= -LA_PAD && col + 8 <= lw + LA_PAD )
                strip = mew_load_piece( rbase, P.plane_elems, P.stride, border, y0, lh, tid, col );
        }
#ifdef ME_PROFILE
        const unsigned long long pf_t1 = __builtin_amdgcn_s_memtime();
#endif
        // 3. this wave's step: the body of me_rows_kernel
        const int bx = W - 1 - ( t - 2 * g );
        const bool active = row_ok && !failed && bx >= 0 && bx < W;
        int below_left = from_group_below( r1, lane ), below = from_group_below( r2, lane ), below_right = from_group_below( r3, lane );
        {
            const bool need0 = has_search && !failed && bx0 >= 0 && (bool)__builtin_amdgcn_readfirstlane( (int)has_below );
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
                        failed = true; // keep walking (the barriers of the workgroup must line up), search nothing
                        break;
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
        const unsigned long long pf_t2 = __builtin_amdgcn_s_memtime();
#endif
        if( active && !failed )
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
                ev.px = 8 * bx + lcol; ev.py = 8 * by + lrow;
                ev.f = load_px4_at( fbase, ev.lane_off );
                bool done = false;
                if( !( mvpx | mvpy ) )
                {
                    // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692); the block itself is always in the window
                    const Px4 r = px4_from_raw( ev.win4( 0, ev.px, ev.py ) );
                    cost = block_cost8x8<uint8_t>( ev.f, r, C.mbcmp_satd );
                    done = cost < 64;
                }
                if( !done )
                {
                    int reach = imax2( iabs( mvpx ), iabs( mvpy ) );
#pragma unroll
                    for( int i = 0; i < 4; i++ )
                        if( i < n )
                            reach = imax2( reach, imax2( iabs( mvcx[i] - mvpx ), iabs( mvcy[i] - mvpy ) ) );
                    reach = imax2( reach, imax2( iabs( iclip3( mvpx, 4 * L.fmin_x, 4 * L.fmax_x ) - mvpx ), iabs( iclip3( mvpy, 4 * L.fmin_y, 4 * L.fmax_y ) - mvpy ) ) );
                    reach = imax2( reach, imax2( iabs( iclip3( mvpx, L.smin_x + 2, L.smax_x - 2 ) - mvpx ), iabs( iclip3( mvpy, L.smin_y + 2, L.smax_y - 2 ) - mvpy ) ) );
                    const bool far = reach + 4 * ( P.me_range + 4 ) >= ME_TAB_HALF;
                    ev.use_lds_tab = __builtin_amdgcn_ballot_w64( far ) == 0ull;
                    ev.tab_x = ( ev.use_lds_tab ? ME_TAB_HALF : tab_centre ) - mvpx;
                    ev.tab_y = ( ev.use_lds_tab ? ME_TAB_HALF : tab_centre ) - mvpy;
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                    cost -= zero_bits;
                    if( mvx | mvy )
                        cost += 5 * P.lambda;
                }
            }
            if( ( lane & 15 ) == 0 )
            {
                const unsigned long long gv = ( (unsigned long long)D.tag << 32 ) | (unsigned)( ( mvx & 0xFFFF ) | ( mvy << 16 ) );
                if( g == ME_ROWS - 1 )
                    __hip_atomic_store( D.mvq + xy, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); // read by the workgroup above
                else
                    D.mvq[xy] = gv;
                D.costs[xy] = cost;
            }
        }
        r3 = r2; r2 = r1;
        r1 = ( mvx & 0xFFFF ) | ( mvy << 16 );
#ifdef ME_PROFILE
        {
            const unsigned long long pf_t3 = __builtin_amdgcn_s_memtime();
            pf_bar += pf_t1 - pf_t0; pf_wait += pf_t2 - pf_t1; pf_search += pf_t3 - pf_t2; pf_steps++;
        }
#endif
    }
#ifdef ME_PROFILE
    if( lane == 0 && prof && has_search )
    {
        // slot 2 ("pre" of me_rows_kernel) holds the time in the two barriers and the strip store
        atomicAdd( prof + 0, __builtin_amdgcn_s_memtime() - pf_begin ); atomicAdd( prof + 1, pf_wait ); atomicAdd( prof + 2, pf_bar ); atomicAdd( prof + 3, pf_search );
        atomicAdd( prof + 5, pf_spins ); atomicAdd( prof + 6, pf_steps ); atomicAdd( prof + 7, 1ull );
    }
#endif
}
