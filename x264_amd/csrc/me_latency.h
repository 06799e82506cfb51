// me_latency.h -- the LATENCY form of the lookahead motion search on gfx950, searched out of LDS: a wave64 holds ONE 8x8 block of ONE
// search, lane group g (lanes 8g .. 8g+7) costs candidate g of a set, one block row (8 samples) per lane; the wave walks ONE block row
// of the picture from right to left and keeps the reference window of that row -- the four half-pel planes, +-8 samples around the
// block -- in wave-private LDS, filled one 8-column strip ahead by LDS-DMA (global_load_lds_dwordx4), so that the dependent rounds of a
// block search (me_logic.h: load -> reduce -> choose, ~8 per block) are LDS round trips instead of L1 / L2 round trips.  It serves the
// launches that cannot fill the chip (x264hip.hip launch_searches_t): those are as long as their dependency chain of W + 2 (H - 1)
// block searches whatever their width; everything larger goes to me_rows_kernel (me_search.h).
//
// Behaviour follows the reference's slicetype_mb_cost search part (encoder/slicetype.c:654-709) over x264_me_search_ref
// (encoder/me.c:182-420,774-798, DIA and HEX) + refine_subpel (me.c:865-992); the decision logic is me_logic.h.
//
// Work decomposition.  A search (source frame, reference frame, list, distance) is a W x H field of 8x8 blocks scanned from the
// bottom right; block (x, y) takes its predictors from (x+1, y) and (x-1..x+1, y+1).  Wave (search, y) runs block row y: in step t it
// searches block x = W-1-t, so the block position, the vector limits, the window AND the decisions between two candidate sets are
// wave-uniform (scalar).  Rows are claimed bottom-up through ticket counters; row y takes the vectors of row y+1 from memory
// (self-validating 8-byte granules { mv, tag }, sc1 stores / L1-bypassing loads, one new granule per step, requested a step ahead), so
// it trails the row below by two blocks plus one hand-off; the wave a row depends on always holds an earlier ticket and is running or
// done: the waits cannot deadlock whatever the dispatch order.  W steps per wave.
//
// The window (WIN_*).  The reference's strip copy (strip_layout.h: strip k = columns 8k .. 8k+15 of every row, 16 samples per row)
// is mirrored piece by piece: an LDS slot holds one strip column of the window -- rows Y0-8 .. Y0+16 of the NP planes, 16 samples
// each, plane after plane -- and the window is the slots of strips k-1, k, k+1 (k = the block's own strip) in a ring of four indexed
// by the strip number modulo 4; the fourth slot receives strip k-2 while block k is searched.  A slot is contiguous in the order the
// DMA lanes write it (lane i -> byte 16 i), the lanes pick their source rows.  8 samples starting at ANY column of the window (and
// the quarter-pel partner one column / one row further) lie inside one 16-sample row piece: a tap is read as the three (8-bit) / five
// (16-bit) aligned dwords that cover it and shifted into place with v_alignbyte (unaligned wide DS reads are replayed at 64 cycles).
// Candidates outside the window (|mv| > 8 full samples in x or y) are read from the global strip copy instead (same arithmetic):
// results do not depend on the window.
//
// (Round 4 also had a throughput form on this window -- eight searches of one reference per wave -- measured slower than
// me_rows_kernel at every launch size, experiments/README.md; it was removed in round 5.)
#pragma once
#include "me_search.h"

#define WIN_R 8
#define WIN_ROWS ( 8 + 2 * WIN_R + 1 ) // block rows + reach above and below + the quarter-pel partner row
#define WIN_RING 4
#define WIN_TAB_HALF 512

template <typename T, int NP>
struct WinGeo
{
    static constexpr int E = (int)sizeof( T );
    static constexpr int ROWB = 16 * E;                      // bytes of one row piece
    static constexpr int SLOTB = NP * WIN_ROWS * ROWB;       // bytes of one strip column of the window
    static constexpr int BYTES = WIN_RING * SLOTB;
    static constexpr int NCH = NP * WIN_ROWS * E;            // 16-byte chunks per slot
    static constexpr int NI = ( NCH + 63 ) / 64;             // DMA instructions per slot
};

// one LDS-DMA instruction: lane i of the wave copies 16 bytes from sbase + voff (per lane) to LDS byte lds_dst + 16 i (M0 is the
// destination base; it is compiler-reserved, so it is set and restored inside the statement).  hipcc does not count this load:
// me_dma_drain() before the window is read.
__device__ __forceinline__ void me_dma16( const void *sbase, unsigned voff, unsigned lds_dst )
{
    unsigned keep;
    asm volatile( "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                  : "=&s"( keep )
                  : "v"( voff ), "s"( sbase ), "s"( lds_dst )
                  : "memory" );
}
// wait for every outstanding vector-memory operation, the DMA included
__device__ __forceinline__ void me_dma_drain()
{
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
}

__device__ __forceinline__ uint4 gload_u128( const void *ubase, unsigned byte_off )
{
    uint4 w;
    __builtin_memcpy( &w, (const AS_GLOBAL char *)ubase + byte_off, 16 );
    return w;
}
typedef __attribute__( ( address_space( 3 ) ) ) unsigned char lds_byte;
__device__ __forceinline__ uint32_t lds_u32( const unsigned char *lds, int byte_off, int imm )
{
    return *(const uint32_t *)( lds + byte_off + imm );
}
// 8 samples starting at LDS byte address v (any alignment, inside one row piece) + rowb (a multiple of 16)
__device__ __forceinline__ Px8 win_px8( const unsigned char *lds, int v, int rowb, const uint8_t * )
{
    const int a = ( v & ~3 ) + rowb;
    const uint32_t w0 = lds_u32( lds, a, 0 ), w1 = lds_u32( lds, a, 4 ), w2 = lds_u32( lds, a, 8 );
    const unsigned t = (unsigned)v & 3u;
    Px8 r;
    r.lo = px4_from_raw( __builtin_amdgcn_alignbyte( w1, w0, t ) );
    r.hi = px4_from_raw( __builtin_amdgcn_alignbyte( w2, w1, t ) );
    return r;
}
__device__ __forceinline__ Px8 win_px8( const unsigned char *lds, int v, int rowb, const uint16_t * )
{
    const int a = ( v & ~3 ) + rowb;
    const uint32_t w0 = lds_u32( lds, a, 0 ), w1 = lds_u32( lds, a, 4 ), w2 = lds_u32( lds, a, 8 ), w3 = lds_u32( lds, a, 12 ), w4 = lds_u32( lds, a, 16 );
    const unsigned t = (unsigned)v & 3u; // 0 or 2
    Px8 r;
    r.lo.a = __builtin_amdgcn_alignbyte( w1, w0, t ); r.lo.b = __builtin_amdgcn_alignbyte( w2, w1, t ); r.lo.raw = 0;
    r.hi.a = __builtin_amdgcn_alignbyte( w3, w2, t ); r.hi.b = __builtin_amdgcn_alignbyte( w4, w3, t ); r.hi.raw = 0;
    return r;
}

#define WIN_NONE ( (int)0x80000000 ) // no packed vector looks like this (mvy = -32768)

// ---- one search per wave: the candidates of a set across the eight lane groups ---------------------------------------------------------
// The evaluator of me_logic.h on this geometry: lane = one row of candidate (lane >> 3), a group sum is three DPP steps, the cheapest
// candidate a packed minimum over the groups (two row broadcasts) moved to a scalar register -- so a set of up to eight candidates is
// ONE pass of ~45 instructions whatever its size, and everything between two sets (the decision logic of me_logic.h) is wave-uniform
// and runs on the scalar unit.  A block search is then ~400 instructions instead of ~1 900 for eight blocks side by side
// (me_rows_kernel): a fifth of the latency per block at five times the instructions per block.
#define DPP_ROW_ROR8_ 0x128
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
__device__ __forceinline__ int wave_min_groups( int v ) // minimum over the eight groups (every lane of a group holds the group's value) -> scalar
{
    v = imin2( v, dpp_mov<DPP_ROW_ROR8_>( v ) );
    v = imin2( v, __builtin_amdgcn_update_dpp( v, v, DPP_ROW_BCAST15, 0xA, 0xF, false ) );
    v = imin2( v, __builtin_amdgcn_update_dpp( v, v, DPP_ROW_BCAST31, 0xC, 0xF, false ) );
    return __builtin_amdgcn_readlane( v, 63 );
}
template <typename T, int LDS_TAB, int WEIGHTED>
struct WaveEval
{
    static constexpr int NP = WEIGHTED ? 5 : 4;
    typedef WinGeo<T, NP> G;
    const unsigned char *win;
    const uint16_t *lds_tab;
    const T *sbase, *wsbase;
    const uint16_t *tab;
    int plane_elems, strip_elems, pixel_max;
    int fpelcmp_satd;
    WtD wt;
    int cx0, row16;
    int tab_x, tab_y;
    Px8 f;
    LaneSlots S;   // only row16 (this lane's row in strip-row units) is used
    int rowb;
    int grp;       // lane >> 3: the candidate this lane works on

    __device__ __forceinline__ int bits( int qx, int qy ) const
    {
        if( LDS_TAB )
            return lds_tab[qx + tab_x] + lds_tab[qy + tab_y];
        return gload_u16( tab, 2u * (unsigned)( qx + tab_x ) ) + gload_u16( tab, 2u * (unsigned)( qy + tab_y ) );
    }
    __device__ __forceinline__ int win_addr( int p, int x, int y ) const
    {
        const int c = cx0 + x;
        return mad24( ( c >> 3 ) & ( WIN_RING - 1 ), G::SLOTB, mad24( mad24( p, WIN_ROWS, y + WIN_R ), G::ROWB, ( c & 7 ) * G::E ) );
    }
    static __device__ __forceinline__ bool in_window( int x, int y )
    {
        return (unsigned)( x + WIN_R ) <= 2u * WIN_R && (unsigned)( y + WIN_R ) <= 2u * WIN_R;
    }
    template <int N, class GEN>
    __device__ __forceinline__ int fpel_set( GEN gen ) const
    {
        const int k = imin2( grp, N - 1 );
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( k, x, y, ok, wb );
        const int b = wb ? bits( 4 * x, 4 * y ) : 0;
        Px8 r;
        if( __builtin_amdgcn_ballot_w64( !in_window( x, y ) ) == 0ull )
            r = win_px8( win, win_addr( WEIGHTED ? 4 : 0, x, y ), rowb, (const T *)nullptr );
        else
            r = load_px8_at( WEIGHTED ? wsbase : sbase, strip_off( cx0 + x, row16 + ( y << 4 ), strip_elems ) + S.row16 );
        int total = reduce8( block_partial8<T>( f, r, fpelcmp_satd ) );
        if( fpelcmp_satd ) total >>= 1;
        return wave_min_groups( ok && grp < N ? ( ( total + b ) << 3 ) | k : ME_PACK_MAX );
    }
    // (the candidates of a set are evaluated side by side, one is as dear as eight: the answer only matters to me_logic.h's fused start)
    template <int N, class GEN>
    __device__ __forceinline__ bool more_than_first( GEN gen ) const
    {
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( imin2( grp, N - 1 ), x, y, ok, wb );
        return __builtin_amdgcn_ballot_w64( ok && grp >= 1 && grp < N ) != 0ull;
    }
    template <int N, class GEN>
    __device__ __forceinline__ int qpel_set( int use_satd, GEN gen, int &cost0 ) const
    {
        bool ok;
        const int total = qpel_totals<N>( use_satd, gen, ok );
        cost0 = __builtin_amdgcn_readlane( total, 0 );
        return wave_min_groups( ok && grp < N ? ( total << 3 ) | imin2( grp, N - 1 ) : ME_PACK_MAX );
    }
    // candidates 0..2 costed for themselves (groups 0..2), the cheapest of candidates 3..N-1 packed with k - 3
    template <int N, class GEN>
    __device__ __forceinline__ int qpel_fused( int use_satd, GEN gen, int &c0, int &c1, int &c2 ) const
    {
        bool ok;
        const int total = qpel_totals<N>( use_satd, gen, ok );
        c0 = __builtin_amdgcn_readlane( total, 0 ); c1 = __builtin_amdgcn_readlane( total, 8 ); c2 = __builtin_amdgcn_readlane( total, 16 );
        if( N <= 3 )
            return ME_PACK_MAX;
        return wave_min_groups( ok && grp >= 3 && grp < N ? ( total << 3 ) | ( grp - 3 ) : ME_PACK_MAX );
    }
    // cost (metric + vector bits) of the candidate this lane's group works on, in every lane of the group
    template <int N, class GEN>
    __device__ __forceinline__ int qpel_totals( int use_satd, GEN gen, bool &ok_out ) const
    {
        const int k = imin2( grp, N - 1 );
        int x = 0, y = 0;
        bool ok = false, wb = true;
        gen( k, x, y, ok, wb );
        const int b = wb ? bits( x, y ) : 0;
        const int fx = x & 3, fy = y & 3, ix = x >> 2, iy = y >> 2;
        Px8 a, bb;
        if( __builtin_amdgcn_ballot_w64( !in_window( ix, iy ) ) == 0ull )
        {
            const int sh = 2 * ( fx | ( fy << 2 ) );
            const int pa = (int)( ( 0x54FE5454u >> sh ) & 3u ), pb = (int)( ( 0xBABABA10u >> sh ) & 3u );
            const int va = win_addr( pa, ix, iy + ( fy == 3 ) ), vb = win_addr( pb, ix + ( fx == 3 ), iy );
            // both taps are read even where they are the same sample (full- and half-pel positions): skipping the second read needs a
            // copy of the first one's registers, i.e. a wait for it, and this kernel is about the latency of a block
            a = win_px8( win, va, rowb, (const T *)nullptr );
            bb = win_px8( win, vb, rowb, (const T *)nullptr );
        }
        else
        {
            int oa, ob;
            strip_layout::qpel_taps( plane_elems, strip_off( cx0 + ix, row16 + ( iy << 4 ), strip_elems ), x, y, oa, ob );
            a = load_px8_at( sbase, oa + S.row16 );
            bb = load_px8_at( sbase, ob + S.row16 );
        }
        Px8 r;
        r.lo = avg_px4( a.lo, bb.lo, (const T *)nullptr ); r.hi = avg_px4( a.hi, bb.hi, (const T *)nullptr );
        if( WEIGHTED )
        {
            r.lo = weight_px4<T>( r.lo, wt, pixel_max ); r.hi = weight_px4<T>( r.hi, wt, pixel_max );
        }
        int total = reduce8( block_partial8<T>( f, r, use_satd ) );
        if( use_satd ) total >>= 1;
        ok_out = ok;
        return total + b;
    }
    // me_logic.h's shortcut for the quarter-pel diamond at a half-pel position: the window serves the ten taps as they are
    __device__ __forceinline__ int qpel_star5( int use_satd, int mvx, int mvy, bool inside ) const
    {
        int c0;
        return qpel_set<5>( use_satd, [&]( int k, int &x, int &y, bool &ok, bool &wb ) {
            const bool moved = k > 0 && inside;
            x = moved ? mvx + melogic::dia_dx( k - 1 ) : mvx; y = moved ? mvy + melogic::dia_dy( k - 1 ) : mvy;
            ok = k == 0 || inside; wb = true;
        }, c0 );
    }
    __device__ __forceinline__ bool any( bool c ) const { return c; } // uniform
#ifdef ME_PROFILE
    unsigned long long pf_last;
    unsigned pf_phase[5];
    int pf_kept, pf_single_start, pf_total_start, pf_single_hpel;
    __device__ __forceinline__ void mark( int k )
    {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if( k ) pf_phase[k] += (unsigned)( now - pf_last );
        pf_last = now;
    }
#endif
};

// MODE / WEIGHTED as in me_rows_kernel (me_search.h).  Q.base[] counts searches.
// Block rows (waves) per workgroup of the latency form.  RW > 1 hands vectors from row to row through LDS inside a workgroup; measured with
// RW = 4 (round 4, 1080p, launches of 24 searches): launch 1.024 ms against 1.019 ms, wait for the row below 2 660 against 2 380 cycles per
// step -- what a row waits for is the block below-left being SEARCHED (the spread of the step times along the dependency chain), not the
// trip of its vector through memory.  So the default stays one row per workgroup; -DME_LAT_ROWS=4 builds the other form.
#ifndef ME_LAT_ROWS
#define ME_LAT_ROWS 1
#endif
#define WIN_HAND_W 512 // widest row (blocks) handed over through LDS inside a workgroup (8K pictures: 480)
// Round 6, measured and kept OFF: a row GUESSES the one vector it would have to wait for.  Block (x, y) needs the vector of (x - 1, y + 1),
// which the row below finds two blocks after the last vector this row already holds: a row trails the row below by two block searches
// plus a hand-off, H - 1 times along the chain.  Where motion is uniform that vector repeats its right-hand neighbour's, so with
// -DME_LAT_SPEC=1 a step whose granule has not arrived searches on below_left := below and looks at the granule afterwards: the guess
// held -> the result stands, one block search earlier than it could have started; it did not -> the block is searched again with the
// real vector.  Bit-exact (a block is committed only when its neighbours are the real ones; the whole GPU parity suite ran on it), 75 %
// of the steps of the bench clip guess and 8 % of the guesses miss -- and a launch of 25 searches takes 1 146 us instead of 984, the paced
// stream 2 520 instead of 2 780 frames/s (profiles/r06_latency_ab.txt).  The chain is not W + 2 ( H - 1 ) times an average block: block
// searches differ by a factor of ten (a block that ends at the zero test against one that walks the diamond), a row's pace is the row
// below's, and the two blocks of distance are the slack that absorbs the differences; at one block every slow block below stalls every
// row above it at once (cycles per step: the wait for the row below 2 110 -> 1 290, but everything else + 1 680).
#ifndef ME_LAT_SPEC
#define ME_LAT_SPEC 0
#endif
template <typename T, int HEX, int MODE, int WEIGHTED, int RW>
__global__ __launch_bounds__( 64 * RW, ME_MIN_WAVES ) void me_latency_kernel( LaP P, const SearchDesc<T> *descs, MeQueues Q,
                                                                          unsigned *tickets /* [ME_QUEUES * ME_QUEUE_STRIDE] */, unsigned *err_host /* pinned sticky timeout flag */,
                                                                       unsigned spin_limit, unsigned long long *prof /* ME_PROFILE builds: cycle accumulators, else unused */ )
{
    constexpr int NP = WEIGHTED ? 5 : 4;
    typedef WinGeo<T, NP> G;
    const int lane = lane_id();
#ifdef ME_PROFILE
    unsigned long long pf_wait = 0, pf_pre = 0, pf_search = 0, pf_store = 0, pf_spins = 0, pf_steps = 0, pf_guess = 0, pf_miss = 0;
    unsigned long long pf_ph[5] = { 0, 0, 0, 0, 0 };
    const unsigned long long pf_begin = __builtin_amdgcn_s_memtime();
#define PF_NOW() __builtin_amdgcn_s_memtime()
#endif
    const int W = P.mb_w, H = P.mb_h;
    // RW > 1: a workgroup is RW waves on RW consecutive block rows of one search (wave w on the row w above wave 0's): row to row hand-offs
    // inside the workgroup go through LDS (a few hundred cycles instead of a round trip to memory), only wave 0 waits for another workgroup
    const int wv = RW > 1 ? __builtin_amdgcn_readfirstlane( (int)( threadIdx.x >> 6 ) ) : 0;
    // Every wave reports its exit on a second counter; the last one out clears the tickets for the next launch on this stream.
    auto leave = [&]() {
        if( lane == 0 && atomicAdd( &tickets[1], 1u ) == gridDim.x * RW - 1 )
        {
            for( int q = 0; q < ME_QUEUES; q++ )
                atomicExch( &tickets[q * ME_QUEUE_STRIDE], 0u );
            atomicExch( &tickets[1], 0u );
        }
    };
    const int n_bands = ( H + RW - 1 ) / RW; // tickets per search: one per workgroup
    __shared__ int ticket_sh[2];
    int j = 0, si = -1;
    if( wv == 0 )
    {
        const int home = xcc_id();
        for( int k = 0; k < ME_QUEUES && si < 0; k++ )
        {
            const int q = ( home + k ) & ( ME_QUEUES - 1 );
            const int n_q = Q.base[q + 1] - Q.base[q];
            if( !n_q )
                continue;
            unsigned t0 = 0;
            if( lane == 0 )
                t0 = atomicAdd( &tickets[q * ME_QUEUE_STRIDE], 1u );
            const unsigned t = __builtin_amdgcn_readfirstlane( t0 );
            if( t < (unsigned)( n_q * n_bands ) )
            {
                j = t / n_q;
                si = Q.base[q] + ( t - j * n_q );
            }
        }
        if( RW > 1 && lane == 0 ) { ticket_sh[0] = j; ticket_sh[1] = si; }
    }
    if( RW > 1 )
    {
        __syncthreads();
        j = __builtin_amdgcn_readfirstlane( ticket_sh[0] ); si = __builtin_amdgcn_readfirstlane( ticket_sh[1] );
    }
    if( si < 0 )
    {
        leave();
        return;
    }
    const int by = H - 1 - ( RW * j + wv ); // this wave's block row (scalar); above the picture: nothing to do for this wave
    const bool idle = by < 0;
    const int g = lane >> 3;
    const SearchDesc<T> *dp = descs + si;
    const bool leader = lane == 0; // the lane that talks to memory for the block; lanes 0..3 keep its costs
    const T *fbase = dp->fenc0;
    AS_GLOBAL unsigned long long *mvq = (AS_GLOBAL unsigned long long *)dp->mvq;
    AS_GLOBAL int *costs = (AS_GLOBAL int *)dp->costs;
    const unsigned tag = dp->tag;
    const WtD wt = dp->wt;
    const T *sbase = uniform_ptr( dp->ref_strips );
    const T *wsbase = WEIGHTED ? uniform_ptr( dp->refw_strips ) : sbase;

    __shared__ __attribute__( ( aligned( 16 ) ) ) unsigned char win_all[RW * G::BYTES];
    __shared__ uint16_t tab_window[2 * WIN_TAB_HALF];
    __shared__ int hand[RW > 1 ? RW : 1][RW > 1 ? WIN_HAND_W : 1]; // hand[w][x]: the vector wave w found for block x of its row (WIN_NONE: not yet)
    unsigned char *win = win_all + wv * G::BYTES;
    const bool lds_hand = RW > 1 && W <= WIN_HAND_W;
    {
        const int centre = 2 * 4 * P.mv_range; // P.cost_mv is centred: valid differences are -centre .. +centre
        for( int i = threadIdx.x; i < 2 * WIN_TAB_HALF; i += 64 * RW )
        {
            const int d = i - WIN_TAB_HALF;
            tab_window[i] = d >= -centre && d <= centre ? P.cost_mv[d] : (uint16_t)0;
        }
        if( RW > 1 )
            for( int i = threadIdx.x; i < RW * WIN_HAND_W; i += 64 * RW )
                hand[i / WIN_HAND_W][i % WIN_HAND_W] = WIN_NONE;
    }
    __syncthreads(); // orders the cost table (and the hand-off words) before the first block's reads; the last barrier of the kernel
    if( idle )
    {
        leave();
        return;
    }
    MeCfg C;
    C.hex = HEX; C.me_range = P.me_range;
    C.refine4 = MODE == 3 ? P.subpel_refine >= 3 : MODE >= 1;
    C.mbcmp_satd = MODE == 3 ? P.mbcmp_satd : MODE >= 1;
    C.fpelcmp_satd = MODE == 3 ? P.fpelcmp_satd : MODE == 2;
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
    const bool has_below = by < band_end - 1; // scalar
    const int zero_bits = P.cost_mv[0];
    const LaneSlots LS = make_lane_slots( lane );
    const int rowb = ( lane & 7 ) * G::ROWB;

    // ---- the window: DMA source offsets of this lane's chunks of a slot, relative to the strip's first row (bytes) ----
    // chunk q = lane + 64 i of a slot: piece q / E (plane piece / WIN_ROWS, window row piece % WIN_ROWS), 16-byte half q % E
    const unsigned win_lds = (unsigned)(size_t)(const lds_byte *)win;
    unsigned goff[G::NI];
    const int Yw0 = 8 * by + LA_PAD - WIN_R; // first padded row of the window
#pragma unroll
    for( int i = 0; i < G::NI; i++ )
    {
        const int q = lane + 64 * i, piece = q / G::E, half = q % G::E;
        const int p = piece / WIN_ROWS, r = piece - p * WIN_ROWS;
        // plane 4 (weighted builds) = the weighted copy of plane 0, which is a separate allocation: its offset is taken against wsbase
        const long elem = ( p < 4 ? (long)p * 2 * P.plane_elems : 0 ) + (long)( Yw0 + r ) * 16 + half * 8;
        goff[i] = (unsigned)( elem * G::E );
    }
    auto fill_slot = [&]( int ks ) { // strip ks of every plane -> slot ks & 3
        const unsigned strip_byte = (unsigned)ks * (unsigned)strip_elems * G::E;
        const unsigned dst = __builtin_amdgcn_readfirstlane( win_lds + ( ks & ( WIN_RING - 1 ) ) * G::SLOTB );
#pragma unroll
        for( int i = 0; i < G::NI; i++ )
        {
            const int q = lane + 64 * i;
            if( q < G::NCH )
            {
                if( WEIGHTED && q >= 4 * WIN_ROWS * G::E )
                    me_dma16( wsbase, goff[i] + strip_byte, dst + 1024 * i );
                else
                    me_dma16( sbase, goff[i] + strip_byte, dst + 1024 * i );
            }
        }
    };
    const int k_first = W - 1 + ( LA_PAD >> 3 );
    fill_slot( k_first + 1 ); fill_slot( k_first ); fill_slot( k_first - 1 );

    // ---- hand-off state: the vectors of the row below at x+1, x, x-1 (scalars) ----
    const AS_GLOBAL unsigned long long *below_row = mvq + ( by + 1 ) * W;
    const bool below_in_lds = lds_hand && wv > 0; // the row below is wave wv - 1 of this workgroup
    auto granule = [&]( int x ) -> unsigned long long { // lane 0: granule of block (x, by+1), L1-bypassing
        unsigned long long gq = 0;
        if( has_below && leader )
        {
            if( below_in_lds )
            {
                const int v = __hip_atomic_load( &hand[RW > 1 ? wv - 1 : 0][x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );
                gq = v == WIN_NONE ? 0ull : ( (unsigned long long)tag << 32 ) | (unsigned)v;
            }
            else
                gq = __hip_atomic_load( below_row + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        }
        return gq;
    };
    auto granule_ok = [&]( unsigned long long gq ) -> bool { return !leader || (unsigned)( gq >> 32 ) == tag; };
    bool timed_out = false;
    // The granule requested a step ahead normally carries the tag already (the row below is two blocks ahead): the check is then
    // the only cost.  Otherwise spin, reloading, until the granule carries the tag.
    auto granule_spin = [&]( int x ) -> unsigned long long {
        unsigned spins = 0;
        unsigned long long gq;
        do
        {
            if( ++spins > spin_limit )
            {
                timed_out = true;
                return 0ull;
            }
            if( below_in_lds ) __builtin_amdgcn_s_sleep( 1 ); else __builtin_amdgcn_s_sleep( 4 );
#ifdef ME_PROFILE
            pf_spins++;
#endif
            gq = granule( x );
        } while( !__all( granule_ok( gq ) ) );
        return gq;
    };
    // lane 0's vector, as a scalar
    auto granule_mv = [&]( unsigned long long gq ) -> int { return __builtin_amdgcn_readfirstlane( (int)(unsigned)gq ); };
    int below_right = 0, below = 0, below_left = 0;
    unsigned long long g_next = 0;
    if( has_below )
    {
        unsigned long long gq = granule( W - 1 );
        if( !__all( granule_ok( gq ) ) )
            gq = granule_spin( W - 1 );
        below = granule_mv( gq );
        g_next = granule( imax2( W - 2, 0 ) );
    }
    if( timed_out )
    {
        if( lane == 0 )
            report_wait_timeout( err_host, 3u, (unsigned)si, (unsigned)by, (unsigned)W, tag, 0ull );
        me_dma_drain();
        leave();
        return;
    }
    // this lane's source row of block x (strip copy of the source frame: 8 rows of a block are 128 consecutive samples)
    const int frow16 = ( 8 * by + LA_PAD ) << 4;
    // (requested a block ahead and kept as loaded: unpacking it would wait for the load)
    auto source_raw = [&]( int x ) -> uint4 {
        const int o = strip_off( 8 * x + LA_PAD, frow16 + LS.row16, strip_elems );
        uint4 w = { 0, 0, 0, 0 };
        if( sizeof( T ) == 1 ) { const uint2 h = gload_u64( fbase, (unsigned)o ); w.x = h.x; w.y = h.y; }
        else w = gload_u128( fbase, 2u * (unsigned)o );
        return w;
    };
    auto source_px8 = [&]( const uint4 &w ) -> Px8 {
        Px8 r;
        if( sizeof( T ) == 1 ) { r.lo = px4_from_raw( w.x ); r.hi = px4_from_raw( w.y ); }
        else { r.lo.a = w.x; r.lo.b = w.y; r.lo.raw = 0; r.hi.a = w.z; r.hi.b = w.w; r.hi.raw = 0; }
        return r;
    };
    uint4 f_next = source_raw( W - 1 );

    int r1 = 0;                     // packed vector of the block to the right (the previous result)
    int keep_cost = 0;              // lanes 0..3: the cost of the block with x % 4 == lane, until the four leave together
    for( int bx = W - 1; bx >= 0; bx-- )
    {
#ifdef ME_PROFILE
        const unsigned long long pf_t0 = PF_NOW();
#endif
        // Everything the last step requested has landed (the strip of the window, the source block, the granule) and everything it
        // stored is acknowledged: all of it went out at the START of that step, a whole block search ago.
        me_dma_drain();
        const int k = bx + ( LA_PAD >> 3 );
        // What this step consumes of the last step's requests first.  hipcc does not see the DMA instructions, so every s_waitcnt vmcnt(n)
        // it places after them is two operations too strict and would wait for the window: the values are pinned here (the empty asm
        // statements make them "used"), before the new requests go out, and the DMA goes out last.
        Px8 f = source_px8( f_next );
        asm volatile( "" : "+v"( f.lo.raw ), "+v"( f.hi.raw ), "+v"( f.lo.a ), "+v"( f.lo.b ), "+v"( f.hi.a ), "+v"( f.hi.b ) );
        bool guessed = false; // this step searches on a guess of below_left and checks it afterwards (scalar)
        if( has_below && bx > 0 )
        {
            unsigned long long gq = g_next;
            if( __all( granule_ok( gq ) ) )
                below_left = granule_mv( gq );
            else if( ME_LAT_SPEC )
            {
                guessed = true;
                below_left = below;
            }
            else
            {
                gq = granule_spin( bx - 1 );
                below_left = granule_mv( gq );
            }
            // (kept in a scalar register: an asm result in a VGPR counts as divergent, and everything derived from it -- predictor,
            // candidates, the whole decision logic -- would be vector code again)
            asm volatile( "" : "+s"( below_left ) );
        }
        if( timed_out )
            break;
        // Then everything this step sends to memory, in one go: the vector of the block just searched (sc1: the row above is waiting for
        // it), the costs of the four blocks to the right when they are complete (lane j < 4 keeps the block with
        // x % 4 == j; four neighbouring costs are one 16-byte store), and the requests for the next step: the source block, the granule,
        // the strip the window moves onto.  Nothing else touches memory until the next step's wait, which therefore never waits for
        // anything younger than a block search.
        if( bx + 1 < W && leader )
            __hip_atomic_store( mvq + by * W + bx + 1, ( (unsigned long long)tag << 32 ) | (unsigned)r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        if( !( ( bx + 1 ) & 3 ) && bx + 1 < W && lane < 4 && bx + 1 + lane < W )
            costs[by * W + bx + 1 + lane] = keep_cost;
        f_next = source_raw( imax2( bx - 1, 0 ) );
        g_next = granule( imax2( bx - 2, 0 ) );
        if( bx > 0 )
            fill_slot( k - 2 );
#ifdef ME_PROFILE
        const unsigned long long pf_t1 = PF_NOW();
        unsigned long long pf_t2 = pf_t1, pf_t3 = pf_t1;
#endif
        int mvx = 0, mvy = 0, cost = 0;
        for( ;; )
        {
        mvx = 0; mvy = 0; cost = 0;
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
            const int cx0 = 8 * bx + LA_PAD, row16 = frow16;
            bool done = false;
            if( !( mvpx | mvpy ) )
            {
                // near-zero residual shortcut on the unweighted plane (slicetype.c:684-692): plane 0 at zero displacement
                const int v0 = mad24( k & ( WIN_RING - 1 ), G::SLOTB, WIN_R * G::ROWB );
                const Px8 r = win_px8( win, v0, rowb, (const T *)nullptr );
                cost = block_cost8<T>( f, r, C.mbcmp_satd );
                cost = __builtin_amdgcn_readfirstlane( cost );
                done = cost < 64;
            }
#ifdef ME_PROFILE
            pf_t2 = PF_NOW();
#endif
            if( !done )
            {
                // how far from the predictor can a candidate of this block be?  (the cost table window)
                int reach = imax2( iabs( mvpx ), iabs( mvpy ) );
#pragma unroll
                for( int i = 0; i < 4; i++ )
                    if( i < n )
                        reach = imax2( reach, imax2( iabs( mvcx[i] - mvpx ), iabs( mvcy[i] - mvpy ) ) );
                reach = imax2( reach, imax2( iabs( iclip3( mvpx, 4 * L.fmin_x, 4 * L.fmax_x ) - mvpx ), iabs( iclip3( mvpy, 4 * L.fmin_y, 4 * L.fmax_y ) - mvpy ) ) );
                reach = imax2( reach, imax2( iabs( iclip3( mvpx, L.smin_x + 2, L.smax_x - 2 ) - mvpx ), iabs( iclip3( mvpy, L.smin_y + 2, L.smax_y - 2 ) - mvpy ) ) );
                const bool far = reach + 4 * ( P.me_range + 4 ) >= WIN_TAB_HALF;
                if( __builtin_amdgcn_ballot_w64( far ) == 0ull )
                {
                    WaveEval<T, 1, WEIGHTED> ev;
                    ev.grp = g;
                    ev.win = win; ev.lds_tab = tab_window; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = nullptr; ev.plane_elems = P.plane_elems;
                    ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = wt;
                    ev.cx0 = cx0; ev.row16 = row16; ev.f = f; ev.S = LS; ev.rowb = rowb;
                    ev.tab_x = WIN_TAB_HALF - mvpx; ev.tab_y = WIN_TAB_HALF - mvpy;
#ifdef ME_PROFILE
                    for( int i = 0; i < 5; i++ ) ev.pf_phase[i] = 0;
#endif
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
#ifdef ME_PROFILE
                    for( int i = 1; i < 5; i++ ) pf_ph[i] += ev.pf_phase[i];
#endif
                }
                else
                {
                    WaveEval<T, 0, WEIGHTED> ev;
                    ev.grp = g;
                    ev.win = win; ev.lds_tab = nullptr; ev.sbase = sbase; ev.wsbase = wsbase; ev.tab = P.cost_mv - tab_centre; ev.plane_elems = P.plane_elems;
                    ev.strip_elems = strip_elems; ev.pixel_max = P.pixel_max; ev.fpelcmp_satd = C.fpelcmp_satd; ev.wt = wt;
                    ev.cx0 = cx0; ev.row16 = row16; ev.f = f; ev.S = LS; ev.rowb = rowb;
                    ev.tab_x = tab_centre - mvpx; ev.tab_y = tab_centre - mvpy;
#ifdef ME_PROFILE
                    for( int i = 0; i < 5; i++ ) ev.pf_phase[i] = 0;
#endif
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                }
                cost -= zero_bits;
                if( mvx | mvy )
                    cost += 5 * P.lambda;
            }
        }
        if( !guessed )
            break;
        // the guess against the vector the row below has found by now (it is one block search ahead: normally there, else waited for)
        {
            unsigned long long gq = granule( bx - 1 );
            if( !__all( granule_ok( gq ) ) )
                gq = granule_spin( bx - 1 );
            if( timed_out )
                break;
            int real = granule_mv( gq );
            asm volatile( "" : "+s"( real ) );
            guessed = false;
#ifdef ME_PROFILE
            pf_guess++;
#endif
            if( real == below_left )
                break;
#ifdef ME_PROFILE
            pf_miss++;
#endif
            below_left = real;
        }
        }
        if( timed_out )
            break;
#ifdef ME_PROFILE
        pf_t3 = PF_NOW();
#endif
        // blocks slicetype_slice_cost never visits (slicetype.c:823-833) keep zero vectors (frame.c:283-285); the vector leaves at the
        // start of the next step
        const int packed = ( mvx & 0xFFFF ) | ( mvy << 16 );
        if( lds_hand && leader )
            __hip_atomic_store( &hand[wv][bx], packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP ); // the wave above polls this word
        if( lane == ( bx & 3 ) )
            keep_cost = cost;
        r1 = packed;
        below_right = below; below = below_left;
#ifdef ME_PROFILE
        {
            const unsigned long long pf_t4 = PF_NOW();
            const unsigned long long a2 = __builtin_amdgcn_readfirstlane( (unsigned)( pf_t2 - pf_t1 ) ), a3 = __builtin_amdgcn_readfirstlane( (unsigned)( pf_t3 - pf_t2 ) );
            pf_wait += pf_t1 - pf_t0; pf_pre += a2; pf_search += a3; pf_store += pf_t4 - pf_t1 - a2 - a3; pf_steps++;
        }
#endif
    }
    if( timed_out && lane == 0 )
        report_wait_timeout( err_host, 4u, (unsigned)si, (unsigned)by, (unsigned)W, tag, 0ull );
    if( !timed_out )
    {
        if( leader )
            __hip_atomic_store( mvq + by * W, ( (unsigned long long)tag << 32 ) | (unsigned)r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        if( lane < 4 && lane < W )
            costs[by * W + lane] = keep_cost;
    }
    me_dma_drain();
    leave();
#ifdef ME_PROFILE
    if( lane == 0 && prof )
    {
        atomicAdd( prof + 0, PF_NOW() - pf_begin ); atomicAdd( prof + 1, pf_wait ); atomicAdd( prof + 2, pf_pre ); atomicAdd( prof + 3, pf_search );
        atomicAdd( prof + 4, pf_store ); atomicAdd( prof + 5, pf_spins ); atomicAdd( prof + 6, pf_steps ); atomicAdd( prof + 7, 1ull );
        for( int i = 1; i < 5; i++ ) atomicAdd( prof + 7 + i, pf_ph[i] );
        atomicAdd( prof + 26, pf_guess ); atomicAdd( prof + 27, pf_miss );
    }
#endif
}
