// me_search.h -- the lookahead motion search of whole frames on gfx950: eight block rows per wave64, one 8x8 block per 8-lane group,
// one block row per lane, reference samples out of a strip copy of the half-pel planes.
//
// Behaviour follows the reference's slicetype_mb_cost search part (encoder/slicetype.c:654-709) over x264_me_search_ref
// (encoder/me.c:182-420,774-798, DIA and HEX) + refine_subpel (me.c:865-992); the decision logic is me_logic.h.
//
// Work decomposition.  A search (source frame, reference frame, list, distance) is a W x H field of 8x8 blocks scanned from the
// bottom right; block (x, y) takes its predictors from (x+1, y) and (x-1..x+1, y+1).  A wave owns ME_ROWS = 8 consecutive block
// rows: lane group g (lanes 8g..8g+7) walks row y0 - g from right to left, two blocks behind the group below it, so in step t
// group g searches block x = W-1 - (t - 2g) and finds the three vectors of the row below in the registers of group g-1 (its
// results of steps t-1, t-2, t-3; one ds_bpermute each, no memory).  Each group runs the whole block search as its own
// instruction stream (SIMT over groups: candidates one after the other, costs reduced over the 8 lanes with DPP): the selection
// logic is vector code shared by eight blocks, and a row of hand-offs through memory is needed only every eighth row.  Only the
// bottom group of a wave waits for another wave: the top row of the wave below publishes self-validating 8-byte granules
// { mv, tag } with agent-scope relaxed atomics (`sc1` write-through stores / L1-bypassing loads, MI355X guide G16 form R2); the
// other rows' results leave as plain stores.  Row groups are claimed bottom-up through ticket counters, so the wave a group
// depends on always holds an earlier ticket and is running or done: the in-kernel waits cannot deadlock whatever the dispatch
// order.  W + 14 steps per wave.
//
// Why this geometry.  Once the CU is full the search is bound by vector-ALU issue and by the L1 (texture cache) pipe at the same
// time (experiments/README.md "what bounds the search"; issue costs measured with experiments/gen_valu_rate.py: everything but
// plain add/sub/and/or/mov/ashr/cndmask takes 4 cycles per wave64 instruction, not 2).
// * Most instructions are selection logic, address arithmetic and reductions, whose count per wave does not depend on how many
//   pixels a lane holds.  With 8 pixels per lane (one row of the block: two dwords, the byte-wise v_sad_u8 / v_lerp_u8 take them as
//   they are) a block needs 8 lanes, the wave carries eight blocks through one instruction stream and the cost reduction is three
//   DPP steps.  SATD is the sum of four 4x4 transforms: the two column halves of a lane go through the quad-wide transform of
//   device_common.h one after the other.  (Round 2 started with 16 lanes x 4 pixels and four rows per wave: 249 VALU
//   instructions per block against 171 now.)
// * Reference samples come from the STRIP copy of the planes (written next to the row-major planes by lowres_tiles_kernel): strip k
//   of a plane holds columns 8k .. 8k+15 of every row, 16 samples per row, rows one after the other.  The eight rows of a
//   block candidate are then 128 consecutive bytes (2-3 cache lines) instead of one line per row (~9 per block and candidate
//   in row-major planes: the eight-row kernel on row-major planes was SLOWER than the four-row one, 11.6 against 8.7 us per
//   search, because the L1 pipe saturated), and because every column exists in two strips, eight samples starting at any column
//   -- and the quarter-pel partner one column to the right -- are one unaligned load inside one strip.
#pragma once
#include "device_common.h"
#define ME_HD __device__ __forceinline__
#ifdef ME_PROFILE
#define ME_MARK( ev, k ) ( ev ).mark( k )
#endif
#include "me_logic.h"
#include "strip_layout.h"

#define ME_ROWS 8
#ifndef ME_TAPS
#define ME_TAPS 2
#endif
#define DPP_ROW_HALF_MIRROR 0x141 // lane i of every 8 reads lane 7 - i

template <typename T>
struct SearchDesc
{
    const T *fenc0;             // source frame: strip copy of its plane 0 (a block's eight rows are 128 consecutive samples: one or two cache lines
                                // where the row-major plane costs eight -- the kernel is bound by L1 line accesses once the chip is full)
    const T *ref_strips;        // reference frame: strip copy of its four planes, start of the allocation
    const T *refw_strips;       // strip copy of the weighted plane 0 or nullptr
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

// mv costs come from the cost_mv table (h->cost_mv[X264_LOOKAHEAD_QP], analyse.c:151-157), indexed by the quarter-pel difference to
// the block's predictor.  Two table loads per candidate made the vector-memory pipe the busiest unit of the kernel, so each wave
// keeps the entries for differences of less than ME_TAB_HALF quarter-pels in LDS; a block whose candidates could reach beyond that
// window (predictor hundreds of samples long) is searched with the table in memory instead.
#define ME_TAB_HALF 1024

struct Px8
{
    Px4 lo, hi; // samples 0..3 and 4..7 of this lane's row
};

// A wave-wide load whose lanes are not dword-aligned is split by the L1's address unit: 30 CU-cycles per instruction for the 8 bytes of a
// candidate row at its own byte offset against 17.3 for ANY dword-aligned 4 / 8 / 12 / 16 bytes per lane -- the rate of a fully coalesced
// load (experiments/mem_rates, profiles/r05_mem_rates.json).  ME_ALIGN=1 reads a tap as the three aligned dwords that cover it (they
// never leave the 16-sample strip row: the offset inside the row is at most 8) and shifts it into place with two v_alignbyte.  Measured
// (profiles/r05_search_ab.txt): 3.31 against 3.28 us per 1080p search -- the four extra vector instructions per load cost what the L1
// time saves, the kernel is not bound by the L1's address unit -- so the plain unaligned load stays the default.
#ifndef ME_ALIGN
#define ME_ALIGN 0
#endif
__device__ __forceinline__ Px8 load_px8_at( const uint8_t *ubase, int elem_off )
{
    Px8 r;
#if ME_ALIGN
    const u32x3 w = gload_u96( ubase, (unsigned)elem_off & ~3u );
    const unsigned t = (unsigned)elem_off & 3u;
    r.lo = px4_from_raw( __builtin_amdgcn_alignbyte( w.y, w.x, t ) ); r.hi = px4_from_raw( __builtin_amdgcn_alignbyte( w.z, w.y, t ) );
#else
    const uint2 w = gload_u64( ubase, (unsigned)elem_off ); // one global_load_dwordx2 at any byte alignment
    r.lo = px4_from_raw( w.x ); r.hi = px4_from_raw( w.y );
#endif
    return r;
}
__device__ __forceinline__ Px8 load_px8_at( const uint16_t *ubase, int elem_off )
{
    Px8 r;
    r.lo = load_px4_at( ubase, elem_off ); r.hi = load_px4_at( ubase, elem_off + 4 );
    return r;
}
// element offset inside a plane's strips of the 8 samples starting at padded column c of the row whose strip-row offset is row16
// (strip_layout.h, with the 24-bit multiply of the device)
__device__ __forceinline__ int strip_off( int c, int row16, int strip_elems )
{
    return mad24( c >> 3, strip_elems, ( c & 7 ) + row16 );
}
// eight quarter-pel samples per lane out of the strip copy (rounded average of two of the four half-pel planes): sbase = strips of plane 0,
// cx0 / row16 = padded column of the block and strip-row offset of this lane's row at zero displacement (taps: strip_layout.h)
template <typename T>
__device__ __forceinline__ Px8 qpel_px8_strips( const T *sbase, int plane_elems, int strip_elems, int cx0, int row16, int mvx, int mvy )
{
    int oa, ob;
    strip_layout::qpel_taps( plane_elems, strip_off( cx0 + ( mvx >> 2 ), row16 + ( ( mvy >> 2 ) << 4 ), strip_elems ), mvx, mvy, oa, ob );
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
        return reduce8( satd_partial_px4( f.hi, r.hi, (unsigned)satd_partial_px4( f.lo, r.lo ) ) ) >> 1;
    return reduce8( sad_partial_px8( f, r, (const T *)nullptr ) );
}

// ---- candidate sets across the lanes of a group (the evaluator of me_logic.h on the 8-lane geometry) ---------------------------------
// Lane l (0..7) of a group owns candidate slot SLOT(l) = l for l < 4 and 11 - l for l >= 4 (7, 6, 5, 4): with p, q, r the three bits
// of l, SLOT = P | Q << 1 | r << 2 where P = p ^ r and Q = q ^ r.  The two lanes a half-row mirror pairs (l and 7 - l) agree in P and
// Q, so a set of at most four candidates is held twice (once per quad) and one mirror step completes its sums.
#define DPP_QUAD_XOR1_ 0xB1
#define DPP_QUAD_XOR2_ 0x4E
struct LaneSlots
{
    bool P, Q, R;   // this lane's slot bits
    int slot;       // 0..7
    int bp0;        // ds_bpermute byte address of lane 0 of this lane's group
    int row16;      // ( lane & 7 ) << 4: this lane's row of the block in strip-row units
};
__device__ __forceinline__ LaneSlots make_lane_slots( int lane )
{
    LaneSlots S;
    const int l = lane & 7, r = l >> 2;
    S.R = r != 0;
    S.P = ( ( l ^ r ) & 1 ) != 0;
    S.Q = ( ( ( l >> 1 ) ^ r ) & 1 ) != 0;
    S.slot = r ? 11 - l : l;
    S.bp0 = ( lane & ~7 ) << 2;
    S.row16 = l << 4;
    return S;
}
// value held by the lane that owns candidate slot J (a compile-time constant) of this lane's group
template <int J>
__device__ __forceinline__ int from_slot( const LaneSlots &S, int v )
{
    return __builtin_amdgcn_ds_bpermute( S.bp0 + 4 * ( J < 4 ? J : 11 - J ), v );
}
// Sums over the 8 lanes of a group of N per-lane partial costs c[0..N-1] at once: every lane ends up with the total of the candidate
// in ITS slot (slot & 3 for N <= 4).  Each butterfly step halves the number of live values: a lane keeps the half its slot bit selects
// and receives the partner's partial of that half (6 selects + 4 DPP adds for four candidates, 14 + 7 for eight, against 3 DPP adds
// per candidate one at a time).
template <int N>
__device__ __forceinline__ int reduce_slots( const LaneSlots &S, const int *c )
{
    int s[4];
#pragma unroll
    for( int m = 0; m < 4; m++ )
    {
        const int lo = 2 * m < N ? c[2 * m] : 0, hi = 2 * m + 1 < N ? c[2 * m + 1] : 0;
        if( 2 * m + 1 < N )
            s[m] = ( S.P ? hi : lo ) + dpp_mov<DPP_QUAD_XOR1_>( S.P ? lo : hi );
        else if( 2 * m < N )
            s[m] = ( S.P ? 0 : lo ) + dpp_mov<DPP_QUAD_XOR1_>( S.P ? lo : 0 );
        else
            s[m] = 0;
    }
    int u[2];
#pragma unroll
    for( int m = 0; m < 2; m++ )
    {
        if( 4 * m + 2 < N )
            u[m] = ( S.Q ? s[2 * m + 1] : s[2 * m] ) + dpp_mov<DPP_QUAD_XOR2_>( S.Q ? s[2 * m] : s[2 * m + 1] );
        else if( 4 * m < N )
            u[m] = ( S.Q ? 0 : s[2 * m] ) + dpp_mov<DPP_QUAD_XOR2_>( S.Q ? s[2 * m] : 0 );
        else
            u[m] = 0;
    }
    if( N <= 4 )
        return u[0] + dpp_mov<DPP_ROW_HALF_MIRROR>( u[0] );
    return ( S.R ? u[1] : u[0] ) + dpp_mov<DPP_ROW_HALF_MIRROR>( S.R ? u[0] : u[1] );
}
// minimum over the lanes of a group (over a quad for N <= 4: both quads hold the same four slots)
template <int N>
__device__ __forceinline__ int min_slots( int v )
{
    v = imin2( v, dpp_mov<DPP_QUAD_XOR1_>( v ) );
    v = imin2( v, dpp_mov<DPP_QUAD_XOR2_>( v ) );
    if( N > 4 )
        v = imin2( v, dpp_mov<DPP_ROW_HALF_MIRROR>( v ) );
    return v;
}
// per-lane partial cost of the block this group holds against the 8 reference samples r of this lane's row
template <typename T>
__device__ __forceinline__ int block_partial8( const Px8 &f, const Px8 &r, int use_satd )
{
    if( use_satd )
        return satd_partial_px4( f.hi, r.hi, (unsigned)satd_partial_px4( f.lo, r.lo ) );
    return sad_partial_px8( f, r, (const T *)nullptr );
}

template <typename T, int LDS_TAB, int WEIGHTED>
struct GroupEval
{
    const uint16_t *lds_tab; // this wave's window of the mv cost table: entry ME_TAB_HALF + d is the cost of difference d
    const T *sbase;          // wave-uniform: strips of the reference frame's four planes (unweighted)
    const T *wsbase;         // wave-uniform: strips read by full-pel candidates (weighted copy of plane 0, or sbase)
    const uint16_t *tab;     // wave-uniform: first entry of the cost_mv table in memory
    int plane_elems, strip_elems, pixel_max;
    int fpelcmp_satd;
    WtD wt;
    int cx0, row16;          // padded column of the block, strip-row offset of the block's row 0, both at zero displacement
    int tab_x, tab_y;
    Px8 f;                   // this lane's 8 source pixels
    LaneSlots S;

    __device__ __forceinline__ int bits( int qx, int qy ) const
    {
        if( LDS_TAB )
            return lds_tab[qx + tab_x] + lds_tab[qy + tab_y];
        return gload_u16( tab, 2u * (unsigned)( qx + tab_x ) ) + gload_u16( tab, 2u * (unsigned)( qy + tab_y ) );
    }
    // the total cost of the candidate in this lane's slot -> the cheapest of the set, packed with its index
    template <int N>
    __device__ __forceinline__ int pack_min( int total, bool ok ) const
    {
        const int k = N <= 4 ? ( S.slot & 3 ) : S.slot;
        return min_slots<N>( ok && k < N ? ( total << 3 ) | k : ME_PACK_MAX );
    }
    template <int N, class G>
    __device__ __forceinline__ int fpel_set( G gen ) const
    {
        // this lane's own candidate: position, offset of its row 0 in the strips, mv bits
        const int slot = N <= 4 ? ( S.slot & 3 ) : S.slot, k = imin2( slot, N - 1 );
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( k, x, y, ok, wb );
        const int off = strip_off( cx0 + x, row16 + ( y << 4 ), strip_elems );
        const int b = wb ? bits( 4 * x, 4 * y ) : 0;
        // every candidate's row for this lane: the owner's offset plus this lane's row
        const T *base = WEIGHTED ? wsbase : sbase;
        Px8 r[N];
        r[0] = load_px8_at( base, from_slot<0>( S, off ) + S.row16 );
        if constexpr( N > 1 ) r[1] = load_px8_at( base, from_slot<1>( S, off ) + S.row16 );
        if constexpr( N > 2 ) r[2] = load_px8_at( base, from_slot<2>( S, off ) + S.row16 );
        if constexpr( N > 3 ) r[3] = load_px8_at( base, from_slot<3>( S, off ) + S.row16 );
        if constexpr( N > 4 ) r[4] = load_px8_at( base, from_slot<4>( S, off ) + S.row16 );
        if constexpr( N > 5 ) r[5] = load_px8_at( base, from_slot<5>( S, off ) + S.row16 );
        if constexpr( N > 6 ) r[6] = load_px8_at( base, from_slot<6>( S, off ) + S.row16 );
        if constexpr( N > 7 ) r[7] = load_px8_at( base, from_slot<7>( S, off ) + S.row16 );
        int c[N];
#pragma unroll
        for( int j = 0; j < N; j++ )
            c[j] = block_partial8<T>( f, r[j], fpelcmp_satd );
        int total = reduce_slots<N>( S, c );
        if( fpelcmp_satd ) total >>= 1;
        return pack_min<N>( total + b, ok );
    }
    template <int N, class G>
    __device__ __forceinline__ bool more_than_first( G gen ) const
    {
        const int slot = N <= 4 ? ( S.slot & 3 ) : S.slot;
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( imin2( slot, N - 1 ), x, y, ok, wb );
        return any( ok && slot >= 1 && slot < N );
    }
    template <int N, class G>
    __device__ __forceinline__ int qpel_totals( int use_satd, G gen, bool &ok_out ) const
    {
        const int slot = N <= 4 ? ( S.slot & 3 ) : S.slot, k = imin2( slot, N - 1 );
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( k, x, y, ok, wb );
        int oa, ob;
        strip_layout::qpel_taps( plane_elems, strip_off( cx0 + ( x >> 2 ), row16 + ( ( y >> 2 ) << 4 ), strip_elems ), x, y, oa, ob );
        const int b = wb ? bits( x, y ) : 0;
        // does any candidate of any block of the wave average two different samples?  (after a full-pel search the half-pel diamond
        // never does: 98 % of its candidates on the bench clip; a wave-uniform answer, so the branch below costs no masking)
        const bool two_taps = any( oa != ob );
        int ta[N], tb[N];
        ta[0] = from_slot<0>( S, oa ); tb[0] = from_slot<0>( S, ob );
        if constexpr( N > 1 ) { ta[1] = from_slot<1>( S, oa ); tb[1] = from_slot<1>( S, ob ); }
        if constexpr( N > 2 ) { ta[2] = from_slot<2>( S, oa ); tb[2] = from_slot<2>( S, ob ); }
        if constexpr( N > 3 ) { ta[3] = from_slot<3>( S, oa ); tb[3] = from_slot<3>( S, ob ); }
        if constexpr( N > 4 ) { ta[4] = from_slot<4>( S, oa ); tb[4] = from_slot<4>( S, ob ); }
        if constexpr( N > 5 ) { ta[5] = from_slot<5>( S, oa ); tb[5] = from_slot<5>( S, ob ); }
        if constexpr( N > 6 ) { ta[6] = from_slot<6>( S, oa ); tb[6] = from_slot<6>( S, ob ); }
        if constexpr( N > 7 ) { ta[7] = from_slot<7>( S, oa ); tb[7] = from_slot<7>( S, ob ); }
        // Both taps of every candidate are requested before anything waits (ME_TAPS: 2).  A full- or half-pel position has the same
        // sample twice and the second request hits the line the first one brought; skipping it (ME_TAPS: 1) needs a copy of the
        // first tap's registers, which makes every candidate wait for its own load before the next one is issued.
        int c[N];
#if ME_TAPS == 2
        Px8 pa[N], pb[N];
#pragma unroll
        for( int j = 0; j < N; j++ )
            pa[j] = load_px8_at( sbase, ta[j] + S.row16 );
        if( two_taps )
        {
#pragma unroll
            for( int j = 0; j < N; j++ )
                pb[j] = load_px8_at( sbase, tb[j] + S.row16 );
        }
        else
        {
#pragma unroll
            for( int j = 0; j < N; j++ )
                pb[j] = pa[j]; // the copies wait for the loads, but every load of the set has been issued by now
        }
#endif
#pragma unroll
        for( int j = 0; j < N; j++ )
        {
#if ME_TAPS == 2
            const Px8 a = pa[j], bb = pb[j];
#else
            const Px8 a = load_px8_at( sbase, ta[j] + S.row16 );
            Px8 bb = a;
            if( tb[j] != ta[j] )
                bb = load_px8_at( sbase, tb[j] + S.row16 );
#endif
            Px8 r;
            r.lo = avg_px4( a.lo, bb.lo, (const T *)nullptr ); r.hi = avg_px4( a.hi, bb.hi, (const T *)nullptr );
            if( WEIGHTED )
            {
                r.lo = weight_px4<T>( r.lo, wt, pixel_max ); r.hi = weight_px4<T>( r.hi, wt, pixel_max );
            }
            c[j] = block_partial8<T>( f, r, use_satd );
        }
        int total = reduce_slots<N>( S, c );
        if( use_satd ) total >>= 1;
        ok_out = ok;
        return total + b;
    }
    template <int N, class G>
    __device__ __forceinline__ int qpel_set( int use_satd, G gen, int &cost0 ) const
    {
        bool ok;
        const int total = qpel_totals<N>( use_satd, gen, ok );
        cost0 = from_slot<0>( S, total );
        return pack_min<N>( total, ok );
    }
    // candidates 0..2 costed for themselves, the cheapest of candidates 3..N-1 packed with k - 3 (me_logic.h: the start of a search
    // whose only start candidate is the predictor)
    template <int N, class G>
    __device__ __forceinline__ int qpel_fused( int use_satd, G gen, int &c0, int &c1, int &c2 ) const
    {
        bool ok;
        const int total = qpel_totals<N>( use_satd, gen, ok );
        c0 = from_slot<0>( S, total ); c1 = from_slot<1>( S, total ); c2 = from_slot<2>( S, total );
        if( N <= 3 )
            return ME_PACK_MAX;
        return min_slots<8>( ok && S.slot >= 3 && S.slot < N ? ( total << 3 ) | ( S.slot - 3 ) : ME_PACK_MAX );
    }

    // The quarter-pel diamond around a vector at a half-pel position (both components even), its centre re-costed as candidate 0
    // (me_logic.h): each of the four neighbours is the rounded average of the centre's sample run and of the run at the half-pel
    // position two quarter-pels further on (get_ref, mc.c:218-249 with the tables of tables.c:183-184: one tap of an odd position is
    // the even position next to it in each direction) -- five single-tap runs instead of ten taps, one address each.
    __device__ __forceinline__ int qpel_star5( int use_satd, int mvx, int mvy, bool inside ) const
    {
        const int k = imin2( S.slot, 4 );
        const int dx = k && inside ? melogic::dia_dx( k - 1 ) : 0, dy = k && inside ? melogic::dia_dy( k - 1 ) : 0;
        const int px = mvx + 2 * dx, py = mvy + 2 * dy; // the half-pel point this lane addresses
        int oa, ob;
        strip_layout::qpel_taps( plane_elems, strip_off( cx0 + ( px >> 2 ), row16 + ( ( py >> 2 ) << 4 ), strip_elems ), px, py, oa, ob );
        const int b = bits( mvx + dx, mvy + dy );
        const int t0 = from_slot<0>( S, oa ), t1 = from_slot<1>( S, oa ), t2 = from_slot<2>( S, oa ), t3 = from_slot<3>( S, oa ), t4 = from_slot<4>( S, oa );
        const Px8 p0 = load_px8_at( sbase, t0 + S.row16 ), p1 = load_px8_at( sbase, t1 + S.row16 ), p2 = load_px8_at( sbase, t2 + S.row16 ),
                  p3 = load_px8_at( sbase, t3 + S.row16 ), p4 = load_px8_at( sbase, t4 + S.row16 );
        const Px8 *const p[5] = { &p0, &p1, &p2, &p3, &p4 };
        int c[5];
#pragma unroll
        for( int j = 0; j < 5; j++ )
        {
            Px8 r = p0;
            if( j )
            {
                r.lo = avg_px4( p0.lo, p[j]->lo, (const T *)nullptr ); r.hi = avg_px4( p0.hi, p[j]->hi, (const T *)nullptr );
            }
            if( WEIGHTED )
            {
                r.lo = weight_px4<T>( r.lo, wt, pixel_max ); r.hi = weight_px4<T>( r.hi, wt, pixel_max );
            }
            c[j] = block_partial8<T>( f, r, use_satd );
        }
        int total = reduce_slots<5>( S, c );
        if( use_satd ) total >>= 1;
        return pack_min<5>( total + b, k == 0 || inside );
    }
    __device__ __forceinline__ bool any( bool c ) const { return __builtin_amdgcn_ballot_w64( c ) != 0ull; }
#ifdef ME_PROFILE
    unsigned long long pf_last;
    int pf_kept, pf_single_start, pf_total_start, pf_single_hpel;
    unsigned pf_phase[5]; // [k] = cycles between mark k-1 and mark k: 1 start candidates, 2 pattern, 3 half-pel, 4 quarter-pel
    __device__ __forceinline__ void mark( int k )
    {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if( k ) pf_phase[k] += (unsigned)( now - pf_last );
        pf_last = now;
    }
#endif
};

// value of the same lane position one group (8 lanes) further down; group 0 gets garbage it never uses
__device__ __forceinline__ int from_group_below( int v, int lane )
{
    return __builtin_amdgcn_ds_bpermute( ( ( lane - 8 ) & 63 ) << 2, v );
}

// MODE fixes the metric pair at compile time (every cost evaluation would otherwise carry both metrics behind a branch):
//   0  sub-pel depth 2, mbcmp = fpelcmp = SAD            (subme <= 1, encoder.c:1409-1427 + slicetype.c:45-61)
//   1  sub-pel depth 4, mbcmp = SATD, fpelcmp = SAD      (subme >= 2)
//   2  sub-pel depth 4, mbcmp = fpelcmp = SATD           (subme >= 2 with --me tesa)
//   3  any other combination a caller configures: depth and metrics read from the parameters at run time
// WEIGHTED: every search of the launch reads a weighted copy of its reference (D.refw_strips, D.wt)
#ifndef ME_MIN_WAVES
#define ME_MIN_WAVES 4
#endif
// Round 6 measured three ways to take round trips out of a step (profiles/r06_search_ab.txt, r06_search_pmc_*.json): the granule of
// the row below and the source block requested a step ahead (ME_POLL_AHEAD, ME_FENC_AHEAD: the wait for the row below 3 250 -> 1 900
// cycles per step, the source load 1 200 -> 540 in the profiling build) and me_logic.h's fused start set (ME_FUSED_START).  None of them
// moves the kernel: 3.35 against 3.30 us per search alone, 36-37 k against 38-39 k frames/s with eight contexts, 111 vector instructions
// per block either way.  A wave that waits costs its slot, not issue cycles, and with eight launches in flight the slots are full of other
// searches' waves.  They stay in the source, off (ME_SLACK: a wave starts that many blocks later than the vectors allow; no effect either).
#ifndef ME_POLL_AHEAD
#define ME_POLL_AHEAD 0 // 1: the granule of the row below requested a step ahead (0: three granules polled for at the start of every step)
#endif
#ifndef ME_SLACK
#define ME_SLACK 0      // blocks a wave lets the wave below get ahead before it starts, beyond what its first block needs (ME_POLL_AHEAD builds)
#endif
#ifndef ME_FENC_AHEAD
#define ME_FENC_AHEAD 0 // 1: the source block requested a step ahead
#endif
template <typename T, int HEX, int MODE, int WEIGHTED>
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
    // the ticket is wave-uniform: fetched on lane 0 and broadcast through an SGPR, so that the row group, the descriptor and
    // everything derived from them stay scalar
    // Every wave reports its exit on a second counter; the last one out clears the tickets for the next launch on this stream
    // (all ticket requests of a wave have returned before it exits, so nothing can arrive after the clear).  Saves the memset
    // dispatch in front of every search launch.
    auto leave = [&]() {
        if( lane == 0 && atomicAdd( &tickets[1], 1u ) == gridDim.x - 1 )
        {
            for( int q = 0; q < ME_QUEUES; q++ )
                atomicExch( &tickets[q * ME_QUEUE_STRIDE], 0u );
            atomicExch( &tickets[1], 0u );
        }
    };
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
    {
        leave();
        return;
    }
    const SearchDesc<T> D = load_uniform( descs + s ); // s is wave-uniform (the ticket went through an SGPR)
    const int g = lane >> 3;
    const int by0 = H - 1 - ME_ROWS * j; // row of group 0 (scalar)
    const int by = by0 - g;               // this group's row
    const bool row_ok = by >= 0;

    __shared__ uint16_t tab_window[2 * ME_TAB_HALF];
#ifdef ME_LDS_PAD
    // (A/B builds: LDS nobody reads, to hold the number of resident waves per SIMD below what the registers allow)
    __shared__ unsigned me_lds_pad[ME_LDS_PAD / 4];
    if( spin_limit == 0xFFFFFFFFu ) me_lds_pad[lane] = 0;
#endif
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
    // the descriptor came through the scalar cache (load_uniform: the table was written by an upload kernel of an EARLIER dispatch on this
    // stream, and the scalar cache is invalidated at the start of every dispatch -- device_common.h); the compiler still cannot prove
    // the pointers uniform, and without these every load below would move its base address into scalar registers again (two
    // v_readfirstlane per load, ~10 % of the vector instructions)
    const T *fbase = uniform_ptr( D.fenc0 );
    const T *sbase = uniform_ptr( D.ref_strips );
    const T *wsbase = WEIGHTED ? uniform_ptr( D.refw_strips ) : sbase;
    unsigned long long *const mvq = uniform_ptr( D.mvq );
    int *const costs_out = uniform_ptr( D.costs );
    const unsigned tag = __builtin_amdgcn_readfirstlane( D.tag );
    const int strip_elems = ( P.plane_elems / P.stride ) * 16; // rows of the padded plane x 16 samples
    const int tab_centre = 2 * 4 * P.mv_range;
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

#ifndef ME_REMAT
    const LaneSlots LS = make_lane_slots( lane );
#endif
    int r1 = 0, r2 = 0, r3 = 0; // packed vectors this group found in the last three steps
    int keep_mv = 0, keep_cost = 0; // lanes 0..3 of a group: the result of the block with x % 4 == lane, until the four leave together
    const int n_steps = W + 2 * ( ME_ROWS - 1 );
#if ME_POLL_AHEAD
    // The row below group 0 belongs to another wave.  A step needs its vectors at x - 1, x, x + 1; two of the three were the last step's
    // x - 1 and x, so ONE new granule per step, and that one is requested a step ahead: the wave below holds an earlier ticket and is
    // normally several blocks further on, so the granule carries the tag when it is looked at and the trip to the L2 (the loads
    // bypass the L1) is hidden behind a whole block search.  Only a granule that does not carry the tag yet is polled for.
    // Every lane requests the same granule (one address: one L2 request) and the tag test is scalar.
    const unsigned long long *const below_row = mvq + ( by0 + 1 ) * W;
    auto remote_granule = [&]( int x ) -> unsigned long long { return __hip_atomic_load( below_row + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); };
    bool timed_out = false;
    auto remote_vector = [&]( int x, unsigned long long gq ) -> int { // the vector of granule x: gq if it carries the tag, else polled for
        unsigned spins = 0;
        while( (unsigned)__builtin_amdgcn_readfirstlane( (int)( gq >> 32 ) ) != tag )
        {
            if( ++spins > spin_limit )
            {
                timed_out = true;
                return 0;
            }
            __builtin_amdgcn_s_sleep( 4 );
#ifdef ME_PROFILE
            pf_spins++;
#endif
            gq = remote_granule( x );
        }
        return __builtin_amdgcn_readfirstlane( (int)(unsigned)gq );
    };
    int rem_c = 0, rem_r = 0; // the remote row's vectors at the coming step's x and x + 1
    unsigned long long g_next = 0;
    if( below_is_remote )
    {
#if ME_SLACK
        // start a few blocks later than the vectors allow: a wave that runs at the heels of the wave below waits for it in every step
        (void)remote_vector( imax2( W - 1 - ME_SLACK, 0 ), remote_granule( imax2( W - 1 - ME_SLACK, 0 ) ) );
#endif
        rem_c = remote_vector( W - 1, remote_granule( W - 1 ) );
        g_next = remote_granule( imax2( W - 2, 0 ) );
        if( timed_out )
        {
            if( lane == 0 )
                __hip_atomic_store( err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
            leave();
            return;
        }
    }
#endif
#if ME_FENC_AHEAD
    // the source block of a step is requested a step ahead (its position does not depend on anything found so far) and kept as loaded:
    // unpacking it would wait for the load
    const int frow16 = ( ( 8 * imax2( by, 0 ) + LA_PAD ) << 4 ) + LS.row16;
    auto source_raw = [&]( int x ) -> uint4 {
        const int o = strip_off( 8 * iclip3( x, 0, W - 1 ) + LA_PAD, frow16, strip_elems );
        uint4 w = { 0, 0, 0, 0 };
        if( sizeof( T ) == 1 ) { const uint2 h = gload_u64( fbase, (unsigned)o ); w.x = h.x; w.y = h.y; }
        else { const uint2 h0 = gload_u64( fbase, 2u * (unsigned)o ), h1 = gload_u64( fbase, 2u * (unsigned)o + 8u ); w.x = h0.x; w.y = h0.y; w.z = h1.x; w.w = h1.y; }
        return w;
    };
    auto source_px8 = [&]( const uint4 &w ) -> Px8 {
        Px8 r;
        if( sizeof( T ) == 1 ) { r.lo = px4_from_raw( w.x ); r.hi = px4_from_raw( w.y ); }
        else { r.lo.a = w.x; r.lo.b = w.y; r.lo.raw = 0; r.hi.a = w.z; r.hi.b = w.w; r.hi.raw = 0; }
        return r;
    };
    uint4 f_next = source_raw( W - 1 + 2 * g ); // (step 0: x = W - 1 + 2 g, clamped: only group 0 is inside the picture yet)
#endif
    for( int t = 0; t < n_steps; t++ )
    {
#ifdef ME_REMAT
        // (A/B builds: the lane-derived constants of a step rebuilt here, behind an opaque copy of the lane id, instead of living in registers
        //  across the loop: 96 registers and no spill under ME_MIN_WAVES=5 -- and five waves per SIMD no faster than four: 3.25 against
        //  3.18-3.27 us per search, profiles/r06_search_order_latency.txt)
        int lane_r = lane;
        asm volatile( "" : "+v"( lane_r ) );
        const LaneSlots LS = make_lane_slots( lane_r );
        const int g = lane_r >> 3;
#endif
        const int bx = W - 1 - ( t - 2 * g );
        const bool active = row_ok && bx >= 0 && bx < W;
#ifdef ME_PROFILE
        const unsigned long long pf_t0 = PF_NOW();
#endif
        // the row below: (x-1, y+1), (x, y+1), (x+1, y+1) are what the group below found one, two and three steps ago
        int below_left = from_group_below( r1, lane ), below = from_group_below( r2, lane ), below_right = from_group_below( r3, lane );
#if ME_POLL_AHEAD
        {
            const int bx0 = W - 1 - t;
            if( bx0 >= 0 && below_is_remote )
            {
                int rem_l = 0;
                if( bx0 > 0 )
                    rem_l = remote_vector( bx0 - 1, g_next );
                if( timed_out )
                {
                    if( lane == 0 )
                        __hip_atomic_store( err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
                    leave();
                    return;
                }
                if( bx0 > 1 )
                    g_next = remote_granule( bx0 - 2 );
                if( g == 0 ) { below = rem_c; below_left = rem_l; below_right = rem_r; }
                rem_r = rem_c; rem_c = rem_l;
            }
        }
#else
        {
            const int bx0 = W - 1 - t;
            if( bx0 >= 0 && below_is_remote )
            {
                unsigned long long gq = 0;
                const int nb = lane == 1 ? ( bx0 > 0 ? -1 : 0 ) : lane == 2 ? ( bx0 < W - 1 ? 1 : 0 ) : 0;
                const unsigned long long *gp = mvq + ( ( by0 + 1 ) * W + bx0 + nb );
                unsigned spins = 0;
                while( 1 )
                {
                    bool ok = true;
                    if( lane < 3 )
                    {
                        gq = __hip_atomic_load( gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                        ok = (unsigned)( gq >> 32 ) == tag;
                    }
                    if( __all( ok ) )
                        break;
                    if( ++spins > spin_limit )
                    {
                        if( lane == 0 )
                        {
                            // what other kinds of access see at the same address, the unit this wave runs on, its queue's ticket counter
                            unsigned long long expect = ~0ull;
                            (void)__hip_atomic_compare_exchange_strong( const_cast<unsigned long long *>( gp ), &expect, ~0ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                            const unsigned long long sys = __hip_atomic_load( gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
                            int qq = 0;
                            for( int k = 0; k < ME_QUEUES; k++ )
                                if( s >= Q.base[k] && s < Q.base[k + 1] ) qq = k;
                            err_host[10] = (unsigned)xcc_id() | ( (unsigned)qq << 8 ) | ( (unsigned)home << 16 );
                            err_host[11] = (unsigned)( expect >> 32 ); err_host[12] = (unsigned)expect;
                            err_host[13] = (unsigned)( sys >> 32 ); err_host[14] = (unsigned)sys;
                            err_host[15] = __hip_atomic_load( &tickets[qq * ME_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                            report_wait_timeout( err_host, 2u, (unsigned)s, (unsigned)j, ( (unsigned)t << 16 ) | (unsigned)bx0, tag, gq );
                        }
                        leave();
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
#endif
#if ME_FENC_AHEAD
        const uint4 f_now = f_next;
        f_next = source_raw( bx - 1 );
#endif
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
                const int cx0 = 8 * bx + LA_PAD, row16 = ( 8 * by + LA_PAD ) << 4; // the block's row 0; this lane's row is LS.row16 further
#if ME_FENC_AHEAD
                const Px8 f = source_px8( f_now );
#else
                const Px8 f = load_px8_at( fbase, strip_off( cx0, row16 + LS.row16, strip_elems ) );
#endif
                bool done = false;
                if( !( mvpx | mvpy ) )
                {
                    // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692)
                    const Px8 r = load_px8_at( sbase, strip_off( cx0, row16 + LS.row16, strip_elems ) );
                    cost = block_cost8<T>( f, r, C.mbcmp_satd );
                    done = cost < 64;
                }
#ifdef ME_PROFILE
                pf_t2 = PF_NOW();
#endif
                if( !done )
                {
                    // How far from the predictor can a candidate of this block be?  Every start candidate is the predictor, a neighbour's
                    // vector, one of them clipped (towards zero: the limits include the zero vector) or the zero vector, and the predictor
                    // is a neighbour's vector or a median of them: with m the largest component of the neighbours (absent ones are zero),
                    // no difference to the predictor exceeds 2 m.  (The exact maximum took 59 vector instructions per step for a test that
                    // fails only for vectors beyond a hundred samples; min / max of the eight components take 8.)
                    const int hi = imax2( imax3( mvcx[0], mvcx[1], mvcx[2] ), imax3( mvcx[3], mvcy[0], imax3( mvcy[1], mvcy[2], mvcy[3] ) ) );
                    const int lo = imin2( imin3( mvcx[0], mvcx[1], mvcx[2] ), imin3( mvcx[3], mvcy[0], imin3( mvcy[1], mvcy[2], mvcy[3] ) ) );
                    const int reach = 2 * imax2( hi, -lo );
                    const bool far = reach + 4 * ( P.me_range + 4 ) >= ME_TAB_HALF;
                    if( __builtin_amdgcn_ballot_w64( far ) == 0ull )
                    {
                        GroupEval<T, 1, WEIGHTED> ev;
                        ev.lds_tab = tab_window; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = nullptr; ev.plane_elems = P.plane_elems;
                        ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = D.wt;
                        ev.cx0 = cx0; ev.row16 = row16; ev.f = f; ev.S = LS;
                        ev.tab_x = ME_TAB_HALF - mvpx; ev.tab_y = ME_TAB_HALF - mvpy;
#ifdef ME_PROFILE
                        for( int k = 0; k < 5; k++ ) ev.pf_phase[k] = 0;
#endif
                        melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
#ifdef ME_PROFILE
                        for( int k = 1; k < 5; k++ ) pf_ph[k] += ev.pf_phase[k];
#ifdef ME_PROFILE_HIST // (-DME_PROFILE_HIST: the atomics distort the cycle counters above)
                        if( prof )
                        {
                            if( ( lane & 7 ) == 0 )
                            {
                                atomicAdd( prof + 12 + ev.pf_kept, 1ull );
                                atomicAdd( prof + 22, (unsigned long long)( ev.pf_single_hpel ? 4 : 0 ) ); atomicAdd( prof + 23, 4ull );
                                atomicAdd( prof + 24, (unsigned long long)ev.pf_single_start ); atomicAdd( prof + 25, (unsigned long long)ev.pf_total_start );
                            }
                            int wmax = 0;
                            for( int v = 4; v > 0 && !wmax; v-- )
                                if( __builtin_amdgcn_ballot_w64( ev.pf_kept == v ) ) wmax = v;
                            if( lane == __builtin_ctzll( __builtin_amdgcn_ballot_w64( true ) ) )
                                atomicAdd( prof + 17 + wmax, 1ull );
                        }
#endif
#endif
                    }
                    else
                    {
                        GroupEval<T, 0, WEIGHTED> ev;
                        ev.lds_tab = nullptr; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = P.cost_mv - tab_centre; ev.plane_elems = P.plane_elems;
                        ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = D.wt;
                        ev.cx0 = cx0; ev.row16 = row16; ev.f = f; ev.S = LS;
                        ev.tab_x = tab_centre - mvpx; ev.tab_y = tab_centre - mvpy;
#ifdef ME_PROFILE
                        for( int k = 0; k < 5; k++ ) ev.pf_phase[k] = 0;
#endif
                        melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                    }
                    cost -= zero_bits;
                    if( mvx | mvy )
                        cost += 5 * P.lambda;
                }
            }
#ifdef ME_PROFILE
            pf_t3 = PF_NOW();
#endif
            // blocks slicetype_slice_cost never visits (slicetype.c:823-833) keep zero vectors (frame.c:283-285)
            // Results leave four blocks at a time: lane j (< 4) of the group keeps the block with x % 4 == j, and when the group
            // reaches x % 4 == 0 the four lanes store four neighbouring granules (one 32-byte sector) and four costs.  One 8-byte
            // and one 4-byte store per block dirtied a sector each, and with a step every ~10 us the L2 had usually written a
            // sector back before its next store arrived (0.58 MB written per search for 98 KB of results).  The top row of the
            // wave is the exception: the wave above is waiting for its vectors, so they go out at once (sc1), only its costs wait.
            const int packed = ( mvx & 0xFFFF ) | ( mvy << 16 );
            const int j = lane & 7;
            if( j == ( bx & 3 ) )
            {
                keep_mv = packed; keep_cost = cost;
            }
            if( g == ME_ROWS - 1 && j == 0 )
                __hip_atomic_store( mvq + xy, ( (unsigned long long)tag << 32 ) | (unsigned)packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
            if( !( bx & 3 ) && j < 4 && bx + j < W )
            {
                if( g != ME_ROWS - 1 )
                    mvq[xy + j] = ( (unsigned long long)tag << 32 ) | (unsigned)keep_mv;
                costs_out[xy + j] = keep_cost;
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
    leave();
#ifdef ME_PROFILE
    if( lane == 0 && prof )
    {
        atomicAdd( prof + 0, PF_NOW() - pf_begin ); atomicAdd( prof + 1, pf_wait ); atomicAdd( prof + 2, pf_pre ); atomicAdd( prof + 3, pf_search );
        atomicAdd( prof + 4, pf_store ); atomicAdd( prof + 5, pf_spins ); atomicAdd( prof + 6, pf_steps ); atomicAdd( prof + 7, 1ull );
        for( int k = 1; k < 5; k++ ) atomicAdd( prof + 7 + k, pf_ph[k] ); // lane 0's view: group 0 of the wave
    }
#endif
}

// ---- the prediction the searches and cost cells read, block by block (test entry x264hip_mc_luma_probe) ---------------------------------
// out[i] = the 8x8 block at lowres position (x, y) displaced by (mvx, mvy) quarter-pels, through qpel_px8_strips -- the tap arithmetic of
// every candidate of me_rows_kernel, me_latency_kernel (outside its window) and cell_b_kernel -- then weighted like mc_luma does
// (common/mc.c:198-218: average first, weight afterwards).  A request per 8-lane group, a row per lane.
struct McProbe
{
    int x, y, mvx, mvy;
};
template <typename T>
__global__ __launch_bounds__( 64 ) void mc_probe_kernel( LaP P, const T *strips, const McProbe *req, int n, WtD wt, T *out )
{
    const int i = blockIdx.x * 8 + ( threadIdx.x >> 3 ), r = threadIdx.x & 7;
    if( i >= n )
        return;
    const McProbe q = req[i];
    const int strip_elems = ( P.plane_elems / P.stride ) * 16;
    Px8 p = qpel_px8_strips<T>( strips, P.plane_elems, strip_elems, q.x + LA_PAD, ( q.y + LA_PAD + r ) << 4, q.mvx, q.mvy );
    if( wt.on )
    {
        p.lo = weight_px4<T>( p.lo, wt, P.pixel_max ); p.hi = weight_px4<T>( p.hi, wt, P.pixel_max );
    }
    int v[8];
    px4_to_ints( p.lo, v ); px4_to_ints( p.hi, v + 4 );
#pragma unroll
    for( int k = 0; k < 8; k++ )
        out[(size_t)i * 64 + r * 8 + k] = (T)v[k];
}
