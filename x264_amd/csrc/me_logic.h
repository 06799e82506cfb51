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
// Candidates are handed to the evaluator in SETS (the four points of a diamond, the six of a hexagon, the start candidates ...): a
// set is described by a generator gen( k, x, y, ok, with_bits ) giving candidate k's position, whether it takes part, and whether its
// vector costs bits.  The evaluator returns the cheapest participating candidate as  ( cost << 3 ) | k  -- the lowest k among equal
// costs -- which is what applying the candidates in order with strict '<' gives (ME_PACK_MAX when none takes part).  On the device a
// set is costed ACROSS the lanes of the group: lane k derives candidate k's address and mv bits, the addresses are handed round, the
// per-lane partial costs of all candidates are reduced together (transposed butterfly) and the arg-min is a packed minimum.  A
// candidate that does not take part must still have a readable position (callers give the set's centre).
//
// E provides:   template <int N, class G> int fpel_set( G gen )                        N <= 8 full-pel candidates: fpelcmp + bits( 4x, 4y )
//               template <int N, class G> int qpel_set( int use_satd, G gen, int &c0 ) N <= 8 quarter-pel candidates (get_ref semantics,
//                                                                                       mc.c:218-249) + bits( qx, qy ); c0 = cost of candidate 0
//               int  bits( int qx, int qy )               p_cost_mvx[qx] + p_cost_mvy[qy] of a vector every thread of the group agrees on
//               bool any( bool c )                        true if c holds for any thread that shares this instruction stream
//               template <int N, class G> bool more_than_first( G gen )   false only if NO thread that shares this instruction stream has a
//                                                         candidate k >= 1 of the set gen describes (may answer true when in doubt): lets the
//                                                         caller evaluate candidate 0 alone
//               template <int N, class G> int qpel_fused( int use_satd, G gen, int &c0, int &c1, int &c2 )
//                                                         N <= 8 quarter-pel candidates (full-pel ones given in quarter-pel units) costed together:
//                                                         c0..c2 = the costs of candidates 0..2 (meaningful where the candidate takes part), the
//                                                         result = the cheapest of candidates 3..N-1 packed with k - 3 (ME_PACK_MAX for N == 3)
// melogic::ScalarSets<E> implements the set functions over E's scalar fpel( x, y ) / qpel( qx, qy, use_satd ) / bits( qx, qy ): the
// host evaluators of the tests use it (candidates one after the other).
#pragma once

#ifndef ME_HD
#define ME_HD __host__ __device__ __forceinline__
#endif

#define ME_COST_MAX ( 1 << 28 )
#ifndef ME_QSTAR
#define ME_QSTAR 1 // 0: the quarter-pel diamond always as a general set (A/B builds)
#endif
#ifndef ME_FUSED_START
#define ME_FUSED_START 0 // 1: predictor, its rounding, the zero vector and the first diamond costed as ONE set where the predictor is the only start
                         // candidate (bit-exact, tests/test_me_logic_host.py runs it; measured in round 6: no gain, see me_search.h)
#endif

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
// (also the first four of the eight points of the square refinement, me.c:411-420: then (-1,-1) (-1,1) (1,-1) (1,1); offset + 1 in
// two bits per index, so that an index that differs from lane to lane costs a shift and a mask)
ME_HD int square_dx( int k ) { return (int)( ( 0xA085u >> ( 2 * k ) ) & 3 ) - 1; }
ME_HD int square_dy( int k ) { return (int)( ( 0x8858u >> ( 2 * k ) ) & 3 ) - 1; }
ME_HD int dia_dx( int k ) { return square_dx( k ); }
ME_HD int dia_dy( int k ) { return square_dy( k ); }

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

// ---- candidate sets -----------------------------------------------------------------------------------------------------------
#define ME_PACK_MAX 0x7FFFFFFF
ME_HD int pk_cost( int p ) { return p >> 3; }
ME_HD int pk_idx( int p ) { return p & 7; }
// entry i (0..3) of a four-entry list as a chain of selects: i may differ from lane to lane on the device
// (the list is passed by value: with the entries behind a pointer the compiler may turn the selects into an indexed load, which on the
// device puts the list into scratch memory)
struct Mv4 { int v0, v1, v2, v3; };
ME_HD Mv4 mv4_of( const int v[4] ) { Mv4 r; r.v0 = v[0]; r.v1 = v[1]; r.v2 = v[2]; r.v3 = v[3]; return r; }
ME_HD int pick4( int i, const Mv4 v ) { return i == 0 ? v.v0 : i == 1 ? v.v1 : i == 2 ? v.v2 : v.v3; }

// The set functions over a scalar evaluator (candidates one after the other): D provides fpel( x, y ), qpel( qx, qy, use_satd ),
// bits( qx, qy ).  Used by the host evaluators of the tests.
template <class D>
struct ScalarSets
{
    template <int N, class G>
    int fpel_set( G gen )
    {
        D &d = *static_cast<D *>( this );
        int best = ME_PACK_MAX;
        for( int k = 0; k < N; k++ )
        {
            int x = 0, y = 0;
            bool ok = false, wb = true;
            gen( k, x, y, ok, wb );
            if( !ok ) continue;
            const int p = ( ( d.fpel( x, y ) + ( wb ? d.bits( 4 * x, 4 * y ) : 0 ) ) << 3 ) | k;
            if( p < best ) best = p;
        }
        return best;
    }
    template <int N, class G>
    bool more_than_first( G gen )
    {
        for( int k = 1; k < N; k++ )
        {
            int x = 0, y = 0;
            bool ok = false, wb = true;
            gen( k, x, y, ok, wb );
            if( ok ) return true;
        }
        return false;
    }
    // the quarter-pel diamond around ( mvx, mvy ) with the centre as candidate 0 (see search()): here simply the set it stands for
    int qpel_star5( int use_satd, int mvx, int mvy, bool inside )
    {
        int c0;
        return qpel_set<5>( use_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
            const bool moved = k > 0 && inside;
            x = moved ? mvx + dia_dx( k - 1 ) : mvx; y = moved ? mvy + dia_dy( k - 1 ) : mvy;
            ok = k == 0 || inside; wb = true;
        }, c0 );
    }
    template <int N, class G>
    int qpel_fused( int use_satd, G gen, int &c0, int &c1, int &c2 )
    {
        D &d = *static_cast<D *>( this );
        int best = ME_PACK_MAX;
        int c[3] = { ME_COST_MAX, ME_COST_MAX, ME_COST_MAX };
        for( int k = 0; k < N; k++ )
        {
            int x = 0, y = 0;
            bool ok = false, wb = true;
            gen( k, x, y, ok, wb );
            if( !ok ) continue;
            const int v = d.qpel( x, y, use_satd ) + ( wb ? d.bits( x, y ) : 0 );
            if( k < 3 ) c[k] = v;
            else if( ( ( v << 3 ) | ( k - 3 ) ) < best ) best = ( v << 3 ) | ( k - 3 );
        }
        c0 = c[0]; c1 = c[1]; c2 = c[2];
        return best;
    }
    template <int N, class G>
    int qpel_set( int use_satd, G gen, int &cost0 )
    {
        D &d = *static_cast<D *>( this );
        int best = ME_PACK_MAX;
        cost0 = ME_COST_MAX;
        for( int k = 0; k < N; k++ )
        {
            int x = 0, y = 0;
            bool ok = false, wb = true;
            gen( k, x, y, ok, wb );
            if( !ok ) continue;
            const int c = d.qpel( x, y, use_satd ) + ( wb ? d.bits( x, y ) : 0 );
            if( k == 0 ) cost0 = c;
            const int p = ( c << 3 ) | k;
            if( p < best ) best = p;
        }
        return best;
    }
};

template <class E>
ME_HD void search( const MeCfg &C, const MeLim &L, E &ev, int mvpx, int mvpy, int n_mvc, const int mvcx[4], const int mvcy[4],
                   int &out_mvx, int &out_mvy, int &out_cost )
{
    int bmx, bmy, bcost;
    int bpred_cost = ME_COST_MAX, bpred_mx = 0, bpred_my = 0;
    int pmvx, pmvy;
    int unused_c0 = 0;
    bool have_first = false; // the first diamond has been costed with the start candidates: p_first
    int p_first = ME_PACK_MAX;
    const Mv4 cand_x = mv4_of( mvcx ), cand_y = mv4_of( mvcy );
    ME_MARK( ev, 0 );

    if( C.refine4 )
    {
        // predictor and neighbour candidates at quarter-pel precision (me.c:216-275): one set, the clipped predictor first, then the
        // neighbours x264_predictor_clip keeps (common/common.h:774-805: not zero, not the predictor), clipped
        pmvx = clip3( mvpx, 4 * L.fmin_x, 4 * L.fmax_x );
        pmvy = clip3( mvpy, 4 * L.fmin_y, 4 * L.fmax_y );
        int pmv_cost = ME_COST_MAX;
        auto start_set = [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
            const int i = k - 1;
            const int vx = pick4( i, cand_x ), vy = pick4( i, cand_y );
            const bool keep = k > 0 && i < n_mvc && ( vx | vy ) && !( vx == pmvx && vy == pmvy );
            x = keep ? clip3( vx, 4 * L.fmin_x, 4 * L.fmax_x ) : pmvx;
            y = keep ? clip3( vy, 4 * L.fmin_y, 4 * L.fmax_y ) : pmvy;
            ok = k == 0 || keep; wb = true;
        };
        // Where motion is uniform every neighbour repeats the predictor and x264_predictor_clip drops them all: the predictor alone.
        // Then everything the search does next is known before anything has been costed: the predictor's full-pel rounding, the zero
        // vector (me.c:258-275) and -- unless the zero vector wins -- the first diamond around the rounding (me.c:322-342).  One set
        // of seven, one round of loads instead of three; the candidates are applied below in the reference's order all the same.
        const bool many = ev.template more_than_first<5>( start_set );
        if( ME_FUSED_START && !many )
        {
            bpred_mx = pmvx; bpred_my = pmvy;
            bmx = ( pmvx + 2 ) >> 2; bmy = ( pmvy + 2 ) >> 2;
            const int rbx = bmx, rby = bmy;
            const bool need_round = ( ( pmvx | pmvy ) & 3 ) != 0;
            const bool need_zero = ( pmvx | pmvy ) && ( bmx | bmy );
            int c_pmv = 0, c_round = 0, c_zero = 0;
            auto fused = [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                // 0 the predictor, 1 its rounding, 2 the zero vector, 3..6 the diamond around the rounding (full-pel points in quarter-pel units)
                const int dx = k >= 3 ? dia_dx( k - 3 ) : 0, dy = k >= 3 ? dia_dy( k - 3 ) : 0;
                x = k == 0 ? pmvx : k == 2 ? 0 : 4 * ( rbx + dx );
                y = k == 0 ? pmvy : k == 2 ? 0 : 4 * ( rby + dy );
                ok = k == 0 || ( k == 1 && need_round ) || ( k == 2 && need_zero ) || k >= 3;
                wb = true;
            };
            p_first = C.hex ? ev.template qpel_fused<3>( C.fpelcmp_satd, fused, c_pmv, c_round, c_zero )
                            : ev.template qpel_fused<7>( C.fpelcmp_satd, fused, c_pmv, c_round, c_zero );
            bpred_cost = c_pmv;
            bcost = bpred_cost;
            if( need_round )
            {
                bcost = c_round;
                if( need_zero && c_zero < c_round ) { bcost = c_zero; bmx = 0; bmy = 0; }
            }
            else if( need_zero && c_zero < bcost ) { bcost = c_zero; bmx = 0; bmy = 0; }
            // (the predictor being the zero vector and cheaper than what stands: me.c:270-275 -- cannot hold here, bcost IS its cost then)
            have_first = !C.hex && bmx == rbx && bmy == rby;
        }
        else
        {
        const int p = many ? ev.template qpel_set<5>( C.fpelcmp_satd, start_set, pmv_cost )
                           : ev.template qpel_set<1>( C.fpelcmp_satd, start_set, pmv_cost );
#ifdef ME_PROFILE
        {
            int kept = 0, single = ( ( pmvx | pmvy ) & 1 ) == 0, total = 1;
            for( int i = 0; i < 4; i++ )
            {
                const int vx = pick4( i, cand_x ), vy = pick4( i, cand_y );
                if( i < n_mvc && ( vx | vy ) && !( vx == pmvx && vy == pmvy ) )
                {
                    kept++; total++;
                    single += ( ( clip3( vx, 4 * L.fmin_x, 4 * L.fmax_x ) | clip3( vy, 4 * L.fmin_y, 4 * L.fmax_y ) ) & 1 ) == 0;
                }
            }
            ev.pf_kept = kept; ev.pf_single_start = single; ev.pf_total_start = total;
        }
#endif
        bpred_cost = pk_cost( p );
        bpred_mx = pmvx; bpred_my = pmvy;
        if( pk_idx( p ) )
        {
            bpred_mx = clip3( pick4( pk_idx( p ) - 1, cand_x ), 4 * L.fmin_x, 4 * L.fmax_x );
            bpred_my = clip3( pick4( pk_idx( p ) - 1, cand_y ), 4 * L.fmin_y, 4 * L.fmax_y );
        }
        bmx = ( bpred_mx + 2 ) >> 2;
        bmy = ( bpred_my + 2 ) >> 2;
        // the rounded best predictor, then the zero vector, in that order (me.c:258-275): the rounded one replaces the cost whatever
        // it is, the zero vector only if it is cheaper
        const bool need_round = ( ( bpred_mx | bpred_my ) & 3 ) != 0;
        const bool need_zero = ( pmvx | pmvy ) && ( bmx | bmy );
        bcost = need_round ? ME_COST_MAX : bpred_cost;
        if( ev.any( need_round || need_zero ) )
        {
            const int p2 = ev.template fpel_set<2>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                x = k == 0 ? bmx : 0; y = k == 0 ? bmy : 0;
                ok = k == 0 ? need_round : need_zero; wb = true;
            } );
            if( need_round || ( need_zero && pk_cost( p2 ) < bcost ) )
            {
                bcost = pk_cost( p2 );
                if( pk_idx( p2 ) == 1 ) { bmx = 0; bmy = 0; }
            }
        }
        if( !( pmvx | pmvy ) && pmv_cost < bcost )
        {
            bcost = pmv_cost; bmx = 0; bmy = 0;
        }
        }
    }
    else
    {
        // predictor rounded to full-pel (it carries no mv bits here), the neighbours x264_predictor_roundclip keeps
        // (common/common.h:789-805), the zero vector (me.c:276-318)
        bmx = clip3( ( mvpx + 2 ) >> 2, L.fmin_x, L.fmax_x );
        bmy = clip3( ( mvpy + 2 ) >> 2, L.fmin_y, L.fmax_y );
        pmvx = bmx; pmvy = bmy;
        const bool need_zero = ( pmvx | pmvy ) != 0;
        const int p = ev.template fpel_set<6>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
            const int i = k - 1;
            const int mx = ( pick4( i, cand_x ) + 2 ) >> 2, my = ( pick4( i, cand_y ) + 2 ) >> 2;
            const bool keep = k > 0 && k < 5 && i < n_mvc && ( mx | my ) && !( mx == pmvx && my == pmvy );
            x = keep ? clip3( mx, L.fmin_x, L.fmax_x ) : k == 5 ? 0 : pmvx;
            y = keep ? clip3( my, L.fmin_y, L.fmax_y ) : k == 5 ? 0 : pmvy;
            ok = k == 0 || keep || ( k == 5 && need_zero );
            wb = k != 0;
        } );
        bcost = pk_cost( p );
        const int kb = pk_idx( p );
        if( kb == 5 ) { bmx = 0; bmy = 0; }
        else if( kb )
        {
            bmx = clip3( ( pick4( kb - 1, cand_x ) + 2 ) >> 2, L.fmin_x, L.fmax_x );
            bmy = clip3( ( pick4( kb - 1, cand_y ) + 2 ) >> 2, L.fmin_y, L.fmax_y );
        }
    }

    ME_MARK( ev, 1 );
    if( !C.hex )
    {
        // radius-1 diamond: up, down, left, right (me.c:322-342)
        int iters = C.me_range;
        do
        {
            int p = p_first;
            if( !have_first )
                p = ev.template fpel_set<4>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                    x = bmx + dia_dx( k ); y = bmy + dia_dy( k ); ok = true; wb = true;
                } );
            have_first = false;
            if( pk_cost( p ) >= bcost )
                break;
            bcost = pk_cost( p );
            bmx += dia_dx( pk_idx( p ) );
            bmy += dia_dy( pk_idx( p ) );
        } while( --iters && in_fpel_range( L, bmx, bmy ) );
    }
    else
    {
        // hexagon (me.c:344-420)
        int dir = -1;
        {
            const int p = ev.template fpel_set<6>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                x = bmx + hex_dx( k ); y = bmy + hex_dy( k ); ok = true; wb = true;
            } );
            if( pk_cost( p ) < bcost ) { bcost = pk_cost( p ); dir = pk_idx( p ); }
        }
        if( dir >= 0 )
        {
            bmx += hex_dx( dir ); bmy += hex_dy( dir );
            // half hexagons: the three new points in the direction of the last move
            for( int i = ( C.me_range >> 1 ) - 1; i > 0 && in_fpel_range( L, bmx, bmy ); i-- )
            {
                const int p = ev.template fpel_set<3>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                    const int kd = mod6( dir + k - 1 );
                    x = bmx + hex_dx( kd ); y = bmy + hex_dy( kd ); ok = true; wb = true;
                } );
                if( pk_cost( p ) >= bcost )
                    break;
                bcost = pk_cost( p );
                dir = mod6( dir + pk_idx( p ) - 1 );
                bmx += hex_dx( dir ); bmy += hex_dy( dir );
            }
        }
        // square refine: (0,-1) (0,1) (-1,0) (1,0) then (-1,-1) (-1,1) (1,-1) (1,1)
        {
            const int p = ev.template fpel_set<8>( [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                x = bmx + square_dx( k ); y = bmy + square_dy( k ); ok = true; wb = true;
            } );
            if( pk_cost( p ) < bcost )
            {
                bcost = pk_cost( p );
                bmx += square_dx( pk_idx( p ) );
                bmy += square_dy( pk_idx( p ) );
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
            cost += ev.bits( 4 * bmx, 4 * bmy ); // the predictor was costed without its bits (me.c:781-782)
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
                const int p = ev.template qpel_set<1>( C.fpelcmp_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                    x = mx; y = my; ok = differs; wb = true;
                }, unused_c0 );
                if( differs && pk_cost( p ) < cost ) { cost = pk_cost( p ); mvx = mx; mvy = my; }
            }
        }
#ifdef ME_PROFILE
        ev.pf_single_hpel = ( ( mvx | mvy ) & 1 ) == 0;
#endif
        {
            // half-pel diamond, one iteration: up, down, left, right
            const int p = ev.template qpel_set<4>( C.fpelcmp_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                x = mvx + 2 * dia_dx( k ); y = mvy + 2 * dia_dy( k ); ok = true; wb = true;
            }, unused_c0 );
            if( pk_cost( p ) < cost )
            {
                cost = pk_cost( p );
                mvx += 2 * dia_dx( pk_idx( p ) );
                mvy += 2 * dia_dy( pk_idx( p ) );
            }
        }
        ME_MARK( ev, 3 );
        if( C.refine4 )
        {
            // quarter-pel diamond, one iteration, costs with mbcmp (me.c:935-976); when mbcmp is not the metric the cost so far was
            // measured with, the current vector is re-costed first (me.c:925-929): it joins the set as candidate 0
            const bool inside = !( mvy <= L.smin_y || mvy >= L.smax_y || mvx <= L.smin_x || mvx >= L.smax_x );
            if( C.mbcmp_satd != C.fpelcmp_satd )
            {
                // (at a half-pel position the evaluator may know a cheaper way to the same five costs: E::qpel_star5)
                const int p = ME_QSTAR && !ev.any( ( ( mvx | mvy ) & 1 ) != 0 ) ? ev.qpel_star5( C.mbcmp_satd, mvx, mvy, inside ) :
                              ev.template qpel_set<5>( C.mbcmp_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                    const bool moved = k > 0 && inside;
                    x = moved ? mvx + dia_dx( k - 1 ) : mvx; y = moved ? mvy + dia_dy( k - 1 ) : mvy;
                    ok = k == 0 || inside; wb = true;
                }, unused_c0 );
                cost = pk_cost( p );
                if( pk_idx( p ) )
                {
                    const int kb = pk_idx( p ) - 1;
                    mvx += dia_dx( kb ); mvy += dia_dy( kb );
                }
            }
            else if( ev.any( inside ) )
            {
                const int p = ev.template qpel_set<4>( C.mbcmp_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                    x = inside ? mvx + dia_dx( k ) : mvx; y = inside ? mvy + dia_dy( k ) : mvy; ok = inside; wb = true;
                }, unused_c0 );
                if( inside && pk_cost( p ) < cost )
                {
                    cost = pk_cost( p );
                    mvx += dia_dx( pk_idx( p ) ); mvy += dia_dy( pk_idx( p ) );
                }
            }
        }
        else if( C.mbcmp_satd != C.fpelcmp_satd )
        {
            const int p = ev.template qpel_set<1>( C.mbcmp_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
                x = mvx; y = mvy; ok = true; wb = true;
            }, unused_c0 );
            cost = pk_cost( p );
        }
    }
    ME_MARK( ev, 4 );
    out_mvx = mvx; out_mvy = mvy; out_cost = cost;
}

} // namespace melogic
