// me_logic.h -- the decision logic of one lookahead block search, written once for the device and for the host.
//
// Behaviour: x264_me_search_ref (encoder/me.c:182-420,774-798, DIA and HEX branches) + refine_subpel (me.c:865-992) as
// slicetype_mb_cost drives them for an 8x8 lowres block (encoder/slicetype.c:654-709).  The candidates of a pattern are
// costed one after the other by an evaluator E and applied in the reference's order with strict '<' -- what its packed
// (cost << k) + index comparisons implement (me.c:330-333,370-375,912-915).
//
// On the device (me_search.h) the "thread" that runs this code is a group of 8 lanes holding one block: E's cost functions
// reduce over the group with DPP and return the same value in its 8 lanes, so control flow is uniform inside a group and may
// differ between the eight groups of a wave (eight block rows searched in lock step).  The costs of one pattern are requested back
// to back, before any of them is consumed: their loads are in flight together.  On the host (tests/tools/me_logic_host.cpp) E is
// a plain scalar evaluator, which is how this logic is checked against the oracle without a GPU.
//
// E provides:   int  fpel( int x, int y )                 pixel cost (fpelcmp) of the full-pel candidate, read from the weighted plane
//               int  qpel( int qx, int qy, int use_satd ) pixel cost of the quarter-pel candidate (get_ref semantics, mc.c:218-249)
//               int  bits( int qx, int qy )               p_cost_mvx[qx] + p_cost_mvy[qy]
//               bool any( bool c )                        true if c holds for any thread that shares this instruction stream
#pragma once

#ifndef ME_HD
#define ME_HD __host__ __device__ __forceinline__
#endif

#define ME_COST_MAX ( 1 << 28 )

struct MeCfg
{
    int hex;            // lookahead h->mb.i_me_method == X264_ME_HEX (else DIA)
    int refine4;        // lookahead h->mb.i_subpel_refine >= 3 (the lookahead only uses 2 and 4, slicetype.c:45-61)
    int me_range;
    int mbcmp_satd, fpelcmp_satd;
};

struct MeLim
{
    int smin_x, smin_y, smax_x, smax_y; // quarter-pel limits (h->mb.mv_min_spel / mv_max_spel)
    int fmin_x, fmin_y, fmax_x, fmax_y; // full-pel limits (h->mb.mv_limit_fpel)
};

namespace melogic {

ME_HD int clip3( int v, int lo, int hi ) { return v < lo ? lo : v > hi ? hi : v; }
ME_HD bool in_fpel_range( const MeLim &L, int x, int y ) { return x >= L.fmin_x && x <= L.fmax_x && y >= L.fmin_y && y <= L.fmax_y; }

// hexagon offsets, index 0..5: (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2)  (me.c:344-350 hex2 without its duplicated ends)
ME_HD int hex_dx( int k ) { return (int)( ( 0x134310u >> ( 4 * k ) ) & 15 ) - 2; }
ME_HD int hex_dy( int k ) { return (int)( ( 0x002442u >> ( 4 * k ) ) & 15 ) - 2; }
ME_HD int mod6( int v ) { return v < 0 ? v + 6 : v >= 6 ? v - 6 : v; }

// the four neighbours of a diamond in the reference's order: up, down, left, right
ME_HD int dia_dx( int k ) { return k == 2 ? -1 : k == 3 ? 1 : 0; }
ME_HD int dia_dy( int k ) { return k == 0 ? -1 : k == 1 ? 1 : 0; }

// motion vector limits of block (bx, by) of a W x H block picture (slicetype.c:518-531 with the lowres 8x8 geometry): the block may
// leave the picture by 12 samples (the padded border is 32), and never by more than the level's vertical / horizontal range
ME_HD void block_limits( MeLim &L, int bx, int by, int W, int H, int mv_range )
{
    const int range = 2 * mv_range;
    L.smin_x = 4 * ( -8 * bx - 12 ); if( L.smin_x < -range ) L.smin_x = -range;
    L.smin_y = 4 * ( -8 * by - 12 ); if( L.smin_y < -range ) L.smin_y = -range;
    L.smax_x = 4 * ( 8 * ( W - bx - 1 ) + 12 ); if( L.smax_x > range - 1 ) L.smax_x = range - 1;
    L.smax_y = 4 * ( 8 * ( H - by - 1 ) + 12 ); if( L.smax_y > range - 1 ) L.smax_y = range - 1;
    L.fmin_x = L.smin_x >> 2; L.fmin_y = L.smin_y >> 2;
    L.fmax_x = L.smax_x >> 2; L.fmax_y = L.smax_y >> 2;
}

ME_HD int median3( int a, int b, int c )
{
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    const int m = hi < c ? hi : c;
    return lo > m ? lo : m;
}

// The neighbour list of slicetype_mb_cost (slicetype.c:662-680): right, below, below-left, below-right, as far as they exist
// (has_below: the row below belongs to the same band and has been searched).  Returns the count; absent entries are zero.
ME_HD int neighbour_list( int bx, int W, bool has_below, int right, int below, int below_left, int below_right, int mvcx[4], int mvcy[4] )
{
    int n = 0;
    int v[4] = { 0, 0, 0, 0 };
    // compaction without dynamic indexing: each candidate lands in slot n, n is at most 3 here
    if( bx < W - 1 ) { v[0] = right; n = 1; }
    if( has_below )
    {
        if( n == 0 ) v[0] = below; else v[1] = below;
        n++;
        if( bx > 0 )
        {
            if( n == 1 ) v[1] = below_left; else v[2] = below_left;
            n++;
        }
        if( bx < W - 1 )
        {
            if( n == 2 ) v[2] = below_right; else v[3] = below_right; // n == 1 cannot happen: bx < W-1 put `right` first
            n++;
        }
    }
#pragma unroll
    for( int i = 0; i < 4; i++ )
    {
        mvcx[i] = (int)(short)( v[i] & 0xFFFF );
        mvcy[i] = v[i] >> 16;
    }
    return n;
}

#ifndef ME_MARK
#define ME_MARK( ev, k ) // profiling builds of the device kernel time the phases of a search
#endif

template <class E>
ME_HD void search( const MeCfg &C, const MeLim &L, E &ev, int mvpx, int mvpy, int n_mvc, const int mvcx[4], const int mvcy[4],
                   int &out_mvx, int &out_mvy, int &out_cost )
{
    int bmx, bmy, bcost;
    int bpred_cost = ME_COST_MAX, bpred_mx = 0, bpred_my = 0;
    int pmvx, pmvy;
    ME_MARK( ev, 0 );

    if( C.refine4 )
    {
        // predictor and neighbour candidates at quarter-pel precision (me.c:216-275)
        bpred_mx = clip3( mvpx, 4 * L.fmin_x, 4 * L.fmax_x );
        bpred_my = clip3( mvpy, 4 * L.fmin_y, 4 * L.fmax_y );
        pmvx = bpred_mx; pmvy = bpred_my;
        // x264_predictor_clip (common/common.h:774-805): candidates equal to zero or to the predictor are dropped, the rest clipped
        bool ok[4];
        int cx[4], cy[4];
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            const int mx = mvcx[i], my = mvcy[i];
            ok[i] = i < n_mvc && ( mx | my ) && !( mx == pmvx && my == pmvy );
            cx[i] = ok[i] ? clip3( mx, 4 * L.fmin_x, 4 * L.fmax_x ) : pmvx;
            cy[i] = ok[i] ? clip3( my, 4 * L.fmin_y, 4 * L.fmax_y ) : pmvy;
        }
        int pmv_cost;
        if( ev.any( ok[0] | ok[1] | ok[2] | ok[3] ) )
        {
            // a dropped candidate is costed at the predictor's position (no new memory traffic) and never applied
            const int v = ev.qpel( pmvx, pmvy, C.fpelcmp_satd ) + ev.bits( pmvx, pmvy );
            int c[4];
#pragma unroll
            for( int i = 0; i < 4; i++ )
                c[i] = ev.qpel( cx[i], cy[i], C.fpelcmp_satd ) + ev.bits( cx[i], cy[i] );
            bpred_cost = pmv_cost = v;
#pragma unroll
            for( int i = 0; i < 4; i++ )
                if( ok[i] && c[i] < bpred_cost ) { bpred_cost = c[i]; bpred_mx = cx[i]; bpred_my = cy[i]; }
        }
        else
            bpred_cost = pmv_cost = ev.qpel( pmvx, pmvy, C.fpelcmp_satd ) + ev.bits( pmvx, pmvy );
        bmx = ( bpred_mx + 2 ) >> 2;
        bmy = ( bpred_my + 2 ) >> 2;
        // the rounded best predictor, then the zero vector, in that order (me.c:258-275)
        const bool need_round = ( ( bpred_mx | bpred_my ) & 3 ) != 0;
        const bool need_zero = ( pmvx | pmvy ) && ( bmx | bmy );
        bcost = need_round ? ME_COST_MAX : bpred_cost;
        if( ev.any( need_round || need_zero ) )
        {
            const int rx = need_round ? bmx : 0, ry = need_round ? bmy : 0;
            const int vr = ev.fpel( rx, ry ) + ev.bits( 4 * rx, 4 * ry );
            const int vz = ev.fpel( 0, 0 ) + ev.bits( 0, 0 );
            if( need_round ) bcost = vr;
            if( need_zero && vz < bcost ) { bcost = vz; bmx = 0; bmy = 0; }
        }
        if( !( pmvx | pmvy ) && pmv_cost < bcost )
        {
            bcost = pmv_cost; bmx = 0; bmy = 0;
        }
    }
    else
    {
        // predictor rounded to full-pel; it carries no mv bits here (me.c:276-318)
        bmx = clip3( ( mvpx + 2 ) >> 2, L.fmin_x, L.fmax_x );
        bmy = clip3( ( mvpy + 2 ) >> 2, L.fmin_y, L.fmax_y );
        pmvx = bmx; pmvy = bmy;
        bool ok[4];
        int cx[4], cy[4];
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            // x264_predictor_roundclip (common/common.h:789-805)
            const int mx = ( mvcx[i] + 2 ) >> 2, my = ( mvcy[i] + 2 ) >> 2;
            ok[i] = i < n_mvc && ( mx | my ) && !( mx == pmvx && my == pmvy );
            cx[i] = ok[i] ? clip3( mx, L.fmin_x, L.fmax_x ) : pmvx;
            cy[i] = ok[i] ? clip3( my, L.fmin_y, L.fmax_y ) : pmvy;
        }
        const bool need_zero = ( pmvx | pmvy ) != 0;
        if( ev.any( ok[0] | ok[1] | ok[2] | ok[3] ) )
        {
            const int v = ev.fpel( pmvx, pmvy );
            int c[4];
#pragma unroll
            for( int i = 0; i < 4; i++ )
                c[i] = ev.fpel( cx[i], cy[i] ) + ev.bits( 4 * cx[i], 4 * cy[i] );
            const int vz = ev.fpel( 0, 0 ) + ev.bits( 0, 0 );
            bcost = v;
            const int px = pmvx, py = pmvy;
#pragma unroll
            for( int i = 0; i < 4; i++ )
                if( ok[i] && c[i] < bcost ) { bcost = c[i]; bmx = cx[i]; bmy = cy[i]; }
            (void)px; (void)py;
            if( need_zero && vz < bcost ) { bcost = vz; bmx = 0; bmy = 0; }
        }
        else
        {
            bcost = ev.fpel( pmvx, pmvy );
            if( ev.any( need_zero ) )
            {
                const int vz = ev.fpel( 0, 0 ) + ev.bits( 0, 0 );
                if( need_zero && vz < bcost ) { bcost = vz; bmx = 0; bmy = 0; }
            }
        }
    }

    ME_MARK( ev, 1 );
    if( !C.hex )
    {
        // radius-1 diamond: up, down, left, right (me.c:322-342)
        int iters = C.me_range;
        do
        {
            int c[4];
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                const int x = bmx + dia_dx( k ), y = bmy + dia_dy( k );
                c[k] = ev.fpel( x, y ) + ev.bits( 4 * x, 4 * y );
            }
            int best = -1;
#pragma unroll
            for( int k = 0; k < 4; k++ )
                if( c[k] < bcost ) { bcost = c[k]; best = k; }
            if( best < 0 )
                break;
            bmx += dia_dx( best );
            bmy += dia_dy( best );
        } while( --iters && in_fpel_range( L, bmx, bmy ) );
    }
    else
    {
        // hexagon (me.c:344-420)
        int dir = -1;
        {
            int c[6];
#pragma unroll
            for( int k = 0; k < 6; k++ )
            {
                const int x = bmx + hex_dx( k ), y = bmy + hex_dy( k );
                c[k] = ev.fpel( x, y ) + ev.bits( 4 * x, 4 * y );
            }
#pragma unroll
            for( int k = 0; k < 6; k++ )
                if( c[k] < bcost ) { bcost = c[k]; dir = k; }
        }
        if( dir >= 0 )
        {
            bmx += hex_dx( dir ); bmy += hex_dy( dir );
            // half hexagons: the three new points in the direction of the last move
            for( int i = ( C.me_range >> 1 ) - 1; i > 0 && in_fpel_range( L, bmx, bmy ); i-- )
            {
                int c[3];
#pragma unroll
                for( int k = 0; k < 3; k++ )
                {
                    const int kd = mod6( dir + k - 1 );
                    const int x = bmx + hex_dx( kd ), y = bmy + hex_dy( kd );
                    c[k] = ev.fpel( x, y ) + ev.bits( 4 * x, 4 * y );
                }
                int best = -2;
#pragma unroll
                for( int k = 0; k < 3; k++ )
                    if( c[k] < bcost ) { bcost = c[k]; best = k - 1; }
                if( best == -2 )
                    break;
                dir = mod6( dir + best );
                bmx += hex_dx( dir ); bmy += hex_dy( dir );
            }
        }
        // square refine: (0,-1) (0,1) (-1,0) (1,0) then (-1,-1) (-1,1) (1,-1) (1,1)
        {
            int c[8];
#pragma unroll
            for( int k = 0; k < 8; k++ )
            {
                const int dx = k < 4 ? dia_dx( k ) : ( k < 6 ? -1 : 1 ), dy = k < 4 ? dia_dy( k ) : ( ( k & 1 ) ? 1 : -1 );
                c[k] = ev.fpel( bmx + dx, bmy + dy ) + ev.bits( 4 * ( bmx + dx ), 4 * ( bmy + dy ) );
            }
            int best = -1;
#pragma unroll
            for( int k = 0; k < 8; k++ )
                if( c[k] < bcost ) { bcost = c[k]; best = k; }
            if( best >= 0 )
            {
                bmx += best < 4 ? dia_dx( best ) : ( best < 6 ? -1 : 1 );
                bmy += best < 4 ? dia_dy( best ) : ( ( best & 1 ) ? 1 : -1 );
            }
        }
    }

    ME_MARK( ev, 2 );
    // back to quarter-pel units (me.c:774-789)
    int mvx, mvy, cost;
    if( !C.refine4 )
    {
        cost = bcost;
        if( bmx == pmvx && bmy == pmvy )
            cost += ev.bits( 4 * bmx, 4 * bmy );
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

    // ---- refine_subpel (me.c:865-992); lookahead rows of subpel_iterations: refine 2 -> hpel 1 / qpel 0, refine 4 -> hpel 1 / qpel 1
    {
        if( !C.refine4 )
        {
            // the clipped predictor itself, if the search did not end on it (me.c:886-893)
            const int mx = clip3( mvpx, L.smin_x + 2, L.smax_x - 2 );
            const int my = clip3( mvpy, L.smin_y + 2, L.smax_y - 2 );
            const bool differs = mx != mvx || my != mvy;
            if( ev.any( differs ) )
            {
                const int c = ev.qpel( mx, my, C.fpelcmp_satd ) + ev.bits( mx, my );
                if( differs && c < cost ) { cost = c; mvx = mx; mvy = my; }
            }
        }
        {
            // half-pel diamond, one iteration: up, down, left, right
            int c[4];
#pragma unroll
            for( int k = 0; k < 4; k++ )
            {
                const int x = mvx + 2 * dia_dx( k ), y = mvy + 2 * dia_dy( k );
                c[k] = ev.qpel( x, y, C.fpelcmp_satd ) + ev.bits( x, y );
            }
            int best = -1;
#pragma unroll
            for( int k = 0; k < 4; k++ )
                if( c[k] < cost ) { cost = c[k]; best = k; }
            if( best >= 0 )
            {
                mvx += 2 * dia_dx( best );
                mvy += 2 * dia_dy( best );
            }
        }
        ME_MARK( ev, 3 );
        if( C.refine4 )
        {
            // quarter-pel diamond, one iteration, costs with mbcmp (me.c:935-976)
            const bool inside = !( mvy <= L.smin_y || mvy >= L.smax_y || mvx <= L.smin_x || mvx >= L.smax_x );
            int c[4];
            int base = cost;
            if( C.mbcmp_satd != C.fpelcmp_satd )
                base = ev.qpel( mvx, mvy, C.mbcmp_satd ) + ev.bits( mvx, mvy );
            if( ev.any( inside ) )
            {
#pragma unroll
                for( int k = 0; k < 4; k++ )
                {
                    const int x = inside ? mvx + dia_dx( k ) : mvx, y = inside ? mvy + dia_dy( k ) : mvy;
                    c[k] = ev.qpel( x, y, C.mbcmp_satd ) + ev.bits( x, y );
                }
                cost = base;
                if( inside )
                {
                    const int omx = mvx, omy = mvy;
#pragma unroll
                    for( int k = 0; k < 4; k++ )
                        if( c[k] < cost ) { cost = c[k]; mvx = omx + dia_dx( k ); mvy = omy + dia_dy( k ); }
                }
            }
            else
                cost = base;
        }
        else if( C.mbcmp_satd != C.fpelcmp_satd )
            cost = ev.qpel( mvx, mvy, C.mbcmp_satd ) + ev.bits( mvx, mvy );
    }
    ME_MARK( ev, 4 );
    out_mvx = mvx; out_mvy = mvy; out_cost = cost;
}

} // namespace melogic
