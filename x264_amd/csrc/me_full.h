// Main-encode motion search: x264_me_search_ref with DIA / HEX / UMH / ESA / TESA and refine_subpel
// (encoder/me.c:182-798, :865-992) for any partition size, luma only, one reference, no weights.  The search is written once as plain
// C++ (BM_HD): compiled for the host by tests/tools it is checked against the oracle and the recordings of the reference without a GPU,
// one candidate after the other (Coop = CoopNone, C = false) -- and that scalar form also exists on the device (me_full_list_kernel,
// X264HIP_ME_FULL_SCALAR=1) as the reference of the form that ships: a WAVE per request (me_full_coop_kernel, Coop = CoopWave, C = true).
// There a block cost is computed across the 64 lanes (four samples per lane, a wave sum), the candidates of a pattern's set are
// requested together (mef_costs_f / mef_costs_q: one memory round trip per set), the ESA scan takes 256 candidates per step with
// v_qsad_pk_u16_u8 (8-bit) or 64 with a lane per candidate, and TESA keeps its ads survivors and its SAD-stage thresholds in scan order
// through ballots and a DPP prefix minimum.  Rates and what bounds them: DESIGN.md section 3 (table) and section 8.
#pragma once
#include <stdint.h>

#ifndef BM_HD
#define BM_HD __host__ __device__ __forceinline__
#endif

#define MF_COST_MAX ( 1 << 28 )
#define MF_TESA_WIDTH_MAX 160   // ( 2 * me_range + 4 ) & ~3 columns, me_range <= 64 ... plus rounding
#define MF_TESA_ROWS_MAX 132    // 2 * me_range + 1 rows + 1

BM_HD int clip3( int v, int lo, int hi ) { return v < lo ? lo : v > hi ? hi : v; }
BM_HD int imin( int a, int b ) { return a < b ? a : b; }
BM_HD int imax( int a, int b ) { return a > b ? a : b; }
BM_HD int rnd_avg( int a, int b ) { return ( a + b + 1 ) >> 1; }
BM_HD int mf_abs( int v ) { return v < 0 ? -v : v; }

// one call of the search (x264_me_t + the h->mb limits it reads); every pointer is addressable by the code that runs the search
template <typename T>
struct MfReq
{
    int i_pixel;                // PIXEL_16x16 .. PIXEL_4x4 (0..6)
    int me_method;              // 0 dia, 1 hex, 2 umh, 3 esa, 4 tesa
    int subpel_refine;          // h->mb.i_subpel_refine
    int me_range;
    int mbcmp_satd, fpelcmp_satd;
    const T *fenc;              // the block to match
    int fenc_stride;
    const T *ref[4];            // full / H / V / HV planes at the block origin
    int stride;
    const uint16_t *integral;   // 8x8-sum plane at the block origin (TESA only)
    long integral_lower;        // elements from there to the 4x4-sum plane
    int mvp[2];
    int lim_min[2], lim_max[2]; // h->mb.mv_limit_fpel
    int spel_min[2], spel_max[2];
    const uint16_t *cost_mv;    // centred
    void *scratch;              // TESA: MF_TESA_ROWS_MAX * MF_TESA_WIDTH_MAX entries of 12 bytes
};

namespace mefull {

template <typename T>
BM_HD int mf_sad( const T *a, int sa, const T *b, int sb, int w, int h )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( sizeof( T ) == 1 )
    {
        // A row per load where the row is 16 or 8 samples (one dwordx4 / dwordx2 at any byte alignment), four samples per v_sad_u8.  The
        // load count is what these searches cost: a lane's candidate is its own address, and the L1's address unit takes 17-53 cycles per
        // wave-load whatever its width (experiments/mem_rates) -- 16 row loads per 16x16 candidate instead of 64 dword loads.
        typedef uint32_t u32x4_u __attribute__( ( ext_vector_type( 4 ), aligned( 1 ) ) );
        typedef uint32_t u32x2_u __attribute__( ( ext_vector_type( 2 ), aligned( 1 ) ) );
        unsigned acc = 0;
        if( w == 16 )
        {
            for( int y = 0; y < h; y++ )
            {
                const u32x4_u wa = *(const u32x4_u *)( a + y*sa ), wb = *(const u32x4_u *)( b + y*sb );
                acc = __builtin_amdgcn_sad_u8( wa.x, wb.x, acc ); acc = __builtin_amdgcn_sad_u8( wa.y, wb.y, acc );
                acc = __builtin_amdgcn_sad_u8( wa.z, wb.z, acc ); acc = __builtin_amdgcn_sad_u8( wa.w, wb.w, acc );
            }
            return (int)acc;
        }
        if( w == 8 )
        {
            for( int y = 0; y < h; y++ )
            {
                const u32x2_u wa = *(const u32x2_u *)( a + y*sa ), wb = *(const u32x2_u *)( b + y*sb );
                acc = __builtin_amdgcn_sad_u8( wa.x, wb.x, acc ); acc = __builtin_amdgcn_sad_u8( wa.y, wb.y, acc );
            }
            return (int)acc;
        }
        for( int y = 0; y < h; y++ )
        {
            uint32_t wa, wb;
            __builtin_memcpy( &wa, a + y*sa, 4 );
            __builtin_memcpy( &wb, b + y*sb, 4 );
            acc = __builtin_amdgcn_sad_u8( wa, wb, acc );
        }
        return (int)acc;
    }
#endif
    int s = 0;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
            s += mf_abs( a[y*sa+x] - b[y*sb+x] );
    return s;
}

template <typename T>
BM_HD int mf_ssd( const T *a, int sa, const T *b, int sb, int w, int h )
{
    int s = 0;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
        {
            int d = a[y*sa+x] - b[y*sb+x];
            s += d*d;
        }
    return s;
}

BM_HD void hadamard4( int *v, int step )
{
    int s01 = v[0] + v[step], d01 = v[0] - v[step], s23 = v[2*step] + v[3*step], d23 = v[2*step] - v[3*step];
    v[0] = s01 + s23; v[step] = d01 + d23; v[2*step] = s01 - s23; v[3*step] = d01 - d23;
}

template <typename T>
BM_HD int hadamard_abs_4x4( const T *a, int sa, const T *b, int sb )
{
    int d[16], s = 0;
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            d[4*y+x] = a[y*sa+x] - b[y*sb+x];
    for( int y = 0; y < 4; y++ ) hadamard4( d + 4*y, 1 );
    for( int x = 0; x < 4; x++ ) hadamard4( d + x, 4 );
    for( int i = 0; i < 16; i++ ) s += mf_abs( d[i] );
    return s;
}

/* satd of any WxH made of 4x4 tiles: sum over tiles of (sum |H4 D H4^T|), halved per 8x4 / 4x4 unit
 * exactly as PIXEL_SATD_C composes x264_pixel_satd_8x4 / _4x4 (T.c:265-332). */
template <typename T>
BM_HD int mf_satd( const T *a, int sa, const T *b, int sb, int w, int h )
{
    int total = 0;
    if( w == 4 )
    {
        for( int y = 0; y < h; y += 4 )
            total += hadamard_abs_4x4( a + y*sa, sa, b + y*sb, sb ) >> 1;
        return total;
    }
    for( int y = 0; y < h; y += 4 )
        for( int x = 0; x < w; x += 8 )
            total += ( hadamard_abs_4x4( a + y*sa + x, sa, b + y*sb + x, sb )
                     + hadamard_abs_4x4( a + y*sa + x + 4, sa, b + y*sb + x + 4, sb ) ) >> 1;
    return total;
}


/* Successive elimination, common/T.c:759-803 (ads4/ads2/ads1): candidates i of a row whose lower bound
 * sum|enc_dc - box sum| + cost_mvx[i] stays below thresh, in order. n_dc = 4, 2 or 1. */
BM_HD int mf_ads( int n_dc, const int *enc_dc, const uint16_t *sums, int delta, const uint16_t *cost_mvx, int16_t *mvs, int width, int thresh )
{
    int nmv = 0;
    for( int i = 0; i < width; i++, sums++ )
    {
        int ads = mf_abs( enc_dc[0] - sums[0] ) + cost_mvx[i];
        if( n_dc == 2 )
            ads += mf_abs( enc_dc[1] - sums[delta] );
        else if( n_dc == 4 )
            ads += mf_abs( enc_dc[1] - sums[8] ) + mf_abs( enc_dc[2] - sums[delta] ) + mf_abs( enc_dc[3] - sums[delta + 8] );
        if( ads < thresh )
            mvs[nmv++] = (int16_t)i;
    }
    return nmv;
}


/* One interpolated sample at lowres integer position (x,y) displaced by quarter-pel (mvx,mvy).
 * Phase (fx,fy): first tap from plane (fx?H:0)+(fy==2?V:0) one row lower when fy==3; second tap
 * from plane (fx==2?H:0)+(fy?V:0) one column further when fx==3; taps equal when fx,fy are even. */
template <typename T>
BM_HD int qpel_sample( const T *const planes[4], int stride, int x, int y, int mvx, int mvy, const void *wt )
{
    int fx = mvx & 3, fy = mvy & 3;
    int ix = x + ( mvx >> 2 ), iy = y + ( mvy >> 2 );
    int pa = ( fx ? 1 : 0 ) + ( fy == 2 ? 2 : 0 );
    int v = planes[pa][( iy + ( fy == 3 ) ) * stride + ix];
    if( ( fx | fy ) & 1 )
    {
        int pb = ( fx == 2 ? 1 : 0 ) + ( fy ? 2 : 0 );
        v = rnd_avg( v, planes[pb][iy * stride + ix + ( fx == 3 )] );
    }
    return v;
}

template <typename T>
BM_HD void mf_mc_luma( T *dst, int ds, const T *const planes[4], int stride, int mvx, int mvy, int w, int h, const void *wt )
{
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
            dst[y*ds+x] = (T)qpel_sample( planes, stride, x, y, mvx, mvy, wt );
}


// C: the search is run by a whole wave in lock step (device only, me_full_coop_kernel): block costs are then computed ACROSS the wave --
// lane l owns four horizontally adjacent samples of the block (quad q = l >> 2 is a 4x4 tile, l & 3 its row, so the vertical Hadamard
// of a SATD is a quad exchange) and the totals come back wave-uniform -- instead of by every lane over the whole block.
template <typename T, bool C = false>
struct Mef
{
    const MfReq<T> *p;
    int bw, bh;
    int bmx, bmy, bcost;
#ifdef __HIPCC__
    int l_active, l_row, l_col; // this lane's four samples: ( l_col .. l_col + 3, l_row ), taking part if l_active
    Px4 l_f;                    // ... of the source block
#endif
};

// Cooperation policy of the exhaustive branches (ESA / TESA).  CoopNone: one thread runs the whole search -- the host check and the
// scalar device reference.  CoopWave (device, below): the 64 lanes of a wave run the search of ONE request in lock step with
// identical state; in the exhaustive scans lane l costs candidate l of every chunk of 64, the chunk's winner is the smallest cost
// and among equal costs the earliest candidate -- exactly what scanning the chunk in order with strict '<' keeps.
struct CoopNone
{
    static constexpr int W = 1;
    int16_t xs_[MF_TESA_WIDTH_MAX + 64];
    BM_HD int lane() const { return 0; }
    BM_HD void argmin( int &, int & ) const {}
    BM_HD unsigned long long ballot( bool p ) const { return p ? 1ull : 0ull; }
    BM_HD unsigned long long max64( unsigned long long v ) const { return v; }
    BM_HD int bcast( int v, int ) const { return v; }
    // before = the smallest v among the lanes in front of this one (none here), all = the smallest v of all lanes
    BM_HD void min_scan( int v, int &before, int &all ) const { before = 1 << 28; all = v; }
    BM_HD void sync() const {}
    BM_HD int16_t *xs() { return xs_; }
};
BM_HD int mf_popc64( unsigned long long v )
{
    int n = 0;
    for( ; v; v &= v - 1 ) n++;
    return n;
}

// TESA, between the SAD stage and the full costs (behaviour: encoder/me.c:704-752): a list of (SAD, position) survivors in scan order, the
// best SAD among them, a bar (the SAD stage's last threshold) and a quota.  While more than twice the quota remain and the bar is above
// the best SAD, the bar moves half way down towards it and every entry above it leaves -- the rest keep their order; then the most expensive
// entry leaves (the first one among equals), its place taken by the LAST entry of the list, until the quota is met.  Written for a group
// of Coop::W cooperating lanes that all hold the same scalars: a pass over the list is ceil( count / W ) chunks, the entries that stay are
// placed by a ballot and the number of staying lanes below each (ordered compaction), the most expensive one is a maximum over packed
// ( SAD, first-index-wins ) keys.  With W = 1 the same code is the scalar form the host tests run.
template <class Coop, class Entry>
BM_HD int mf_thin_survivors( Coop &coop, Entry *list, int count, int best, int bar, int quota )
{
    const unsigned long long below = coop.lane() ? ( 1ull << coop.lane() ) - 1 : 0ull; // the lanes in front of this one
    while( count > 2 * quota && bar > best )
    {
        bar = ( bar + best ) >> 1;
        int kept = 0;
        for( int base = 0; base < count; base += Coop::W )
        {
            const int i = base + coop.lane();
            Entry e = list[i < count ? i : base];
            const bool stays = i < count && e.sad <= bar;
            const unsigned long long who = coop.ballot( stays );
            coop.sync(); // the whole chunk has been read: its entries may be overwritten now (a kept entry never moves up the list)
            if( stays )
                list[kept + mf_popc64( who & below )] = e;
            kept += mf_popc64( who );
        }
        coop.sync();
        count = kept;
    }
    while( count > quota )
    {
        unsigned long long top = 0; // ( SAD << 32 ) | ( 0x7FFFFFFF - index ): the largest key is the largest SAD at the smallest index
        for( int base = 0; base < count; base += Coop::W )
        {
            const int i = base + coop.lane();
            if( i < count )
            {
                const unsigned long long key = ( (unsigned long long)(unsigned)list[i].sad << 32 ) | (unsigned)( 0x7FFFFFFF - i );
                top = key > top ? key : top;
            }
        }
        top = coop.max64( top );
        const int out = 0x7FFFFFFF - (int)(unsigned)top;
        count--;
        coop.sync();
        if( coop.lane() == 0 )
            list[out] = list[count];
        coop.sync();
    }
    return count;
}

#if defined( __HIP_DEVICE_COMPILE__ )
// cost of the candidate block at b (row-major, stride sb) against the source block, over the wave: every lane gets the total
template <typename T, bool C>
__device__ __forceinline__ int mef_wave_cmp( const Mef<T, C> *s, const T *b, int sb, int use_satd )
{
    const Px4 r = load_px4( b + (long)s->l_row * sb + s->l_col );
    int v = use_satd ? satd_partial_px4( s->l_f, r ) : sad_partial_px4( s->l_f, r, (const T *)nullptr );
    v = (int)wave_sum_u32( (unsigned)( s->l_active ? v : 0 ) );
    return use_satd ? v >> 1 : v; // every 4x4 sum of absolute Hadamard coefficients is even: halving the total == PIXEL_SATD_C's halves
}
// the same for the quarter-pel position ( qx, qy ) (get_ref semantics: one plane tap or the rounded average of two, mc.c:218-249)
template <typename T, bool C>
__device__ __forceinline__ int mef_wave_qpel( const Mef<T, C> *s, int qx, int qy, int use_satd )
{
    const MfReq<T> *p = s->p;
    const int fx = qx & 3, fy = qy & 3;
    const long o = (long)( s->l_row + ( qy >> 2 ) ) * p->stride + s->l_col + ( qx >> 2 );
    const int pa = ( fx ? 1 : 0 ) + ( fy == 2 ? 2 : 0 ), pb = ( fx == 2 ? 1 : 0 ) + ( fy ? 2 : 0 );
    Px4 r = load_px4( p->ref[pa] + o + ( fy == 3 ? p->stride : 0 ) );
    if( ( fx | fy ) & 1 )
        r = avg_px4( r, load_px4( p->ref[pb] + o + ( fx == 3 ) ), (const T *)nullptr );
    int v = use_satd ? satd_partial_px4( s->l_f, r ) : sad_partial_px4( s->l_f, r, (const T *)nullptr );
    v = (int)wave_sum_u32( (unsigned)( s->l_active ? v : 0 ) );
    return use_satd ? v >> 1 : v;
}
#endif

template <typename T, bool C>
BM_HD int mef_fpelcmp( const Mef<T, C> *s, const T *b, int sb )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( C ) return mef_wave_cmp( s, b, sb, s->p->fpelcmp_satd );
#endif
    return s->p->fpelcmp_satd ? mf_satd( s->p->fenc, s->p->fenc_stride, b, sb, s->bw, s->bh ) : mf_sad( s->p->fenc, s->p->fenc_stride, b, sb, s->bw, s->bh );
}
template <typename T, bool C>
BM_HD int mef_mbcmp( const Mef<T, C> *s, const T *b, int sb )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( C ) return mef_wave_cmp( s, b, sb, s->p->mbcmp_satd );
#endif
    return s->p->mbcmp_satd ? mf_satd( s->p->fenc, s->p->fenc_stride, b, sb, s->bw, s->bh ) : mf_sad( s->p->fenc, s->p->fenc_stride, b, sb, s->bw, s->bh );
}
template <typename T, bool C>
BM_HD int mef_bits_q( const Mef<T, C> *s, int qx, int qy ) { return s->p->cost_mv[qx - s->p->mvp[0]] + s->p->cost_mv[qy - s->p->mvp[1]]; }
template <typename T, bool C>
BM_HD int mef_bits_f( const Mef<T, C> *s, int fx, int fy ) { return mef_bits_q( s, 4*fx, 4*fy ); } /* BITS_MVD */
template <typename T, bool C>
BM_HD int mef_cost_f( const Mef<T, C> *s, int fx, int fy ) /* the cost COST_MV computes */
{
    return mef_fpelcmp( s, s->p->ref[0] + (long)fy * s->p->stride + fx, s->p->stride ) + mef_bits_f( s, fx, fy );
}
// the same cost computed by THIS lane alone whatever C is: the exhaustive scans hand every lane of a wave its own candidate
template <typename T, bool C>
BM_HD int mef_cost_f_lane( const Mef<T, C> *s, int fx, int fy )
{
    const T *b = s->p->ref[0] + (long)fy * s->p->stride + fx;
    return ( s->p->fpelcmp_satd ? mf_satd( s->p->fenc, s->p->fenc_stride, b, s->p->stride, s->bw, s->bh )
                                : mf_sad( s->p->fenc, s->p->fenc_stride, b, s->p->stride, s->bw, s->bh ) ) + mef_bits_f( s, fx, fy );
}
template <typename T, bool C>
BM_HD void mef_try_f( Mef<T, C> *s, int fx, int fy ) /* COST_MV */
{
    int c = mef_cost_f( s, fx, fy );
    if( c < s->bcost ) { s->bcost = c; s->bmx = fx; s->bmy = fy; }
}
template <typename T, bool C>
BM_HD int mef_cost_q( const Mef<T, C> *s, int qx, int qy, int use_mbcmp ) /* COST_MV_HPEL / COST_MV_SAD / COST_MV_SATD */
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( C ) return mef_wave_qpel( s, qx, qy, use_mbcmp ? s->p->mbcmp_satd : s->p->fpelcmp_satd ) + mef_bits_q( s, qx, qy );
#endif
    T pix[16*16];
    mf_mc_luma( pix, 16, s->p->ref, s->p->stride, qx, qy, s->bw, s->bh, NULL );
    return ( use_mbcmp ? mef_mbcmp( s, pix, 16 ) : mef_fpelcmp( s, pix, 16 ) ) + mef_bits_q( s, qx, qy );
}
// the same for N quarter-pel positions that do not depend on each other (see mef_costs_f): every tap of the set is requested before any
// candidate is averaged and reduced
template <int N, typename T, bool C>
BM_HD void mef_costs_q( const Mef<T, C> *s, const int qx[N], const int qy[N], int use_mbcmp, int c[N] )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( C )
    {
        const MfReq<T> *p = s->p;
        const int use_satd = use_mbcmp ? p->mbcmp_satd : p->fpelcmp_satd;
        Px4 ra[N], rb[N];
#pragma unroll
        for( int k = 0; k < N; k++ )
        {
            const int fx = qx[k] & 3, fy = qy[k] & 3;
            const long o = (long)( s->l_row + ( qy[k] >> 2 ) ) * p->stride + s->l_col + ( qx[k] >> 2 );
            const int pa = ( fx ? 1 : 0 ) + ( fy == 2 ? 2 : 0 ), pb = ( fx == 2 ? 1 : 0 ) + ( fy ? 2 : 0 );
            ra[k] = load_px4( p->ref[pa] + o + ( fy == 3 ? p->stride : 0 ) );
            rb[k].a = rb[k].b = rb[k].raw = 0;
            if( ( fx | fy ) & 1 )
                rb[k] = load_px4( p->ref[pb] + o + ( fx == 3 ) );
        }
#pragma unroll
        for( int k = 0; k < N; k++ )
        {
            Px4 r = ra[k];
            if( ( qx[k] | qy[k] ) & 1 )
                r = avg_px4( r, rb[k], (const T *)nullptr );
            int v = use_satd ? satd_partial_px4( s->l_f, r ) : sad_partial_px4( s->l_f, r, (const T *)nullptr );
            v = (int)wave_sum_u32( (unsigned)( s->l_active ? v : 0 ) );
            c[k] = ( use_satd ? v >> 1 : v ) + mef_bits_q( s, qx[k], qy[k] );
        }
        return;
    }
#endif
    for( int k = 0; k < N; k++ )
        c[k] = mef_cost_q( s, qx[k], qy[k], use_mbcmp );
}
template <typename T, bool C>
BM_HD int mef_in_range( const Mef<T, C> *s, int fx, int fy ) /* CHECK_MVRANGE */
{
    return fx >= s->p->lim_min[0] && fx <= s->p->lim_max[0] && fy >= s->p->lim_min[1] && fy <= s->p->lim_max[1];
}
// The costs of N full-pel candidates that do not depend on each other.  In a wave every candidate's samples are requested before any
// of them is reduced: the set is ONE memory round trip instead of N (a pattern search is a chain of ~50 such evaluations, and the chain
// is what a request costs).
template <int N, typename T, bool C>
BM_HD void mef_costs_f( const Mef<T, C> *s, const int x[N], const int y[N], int c[N] )
{
#if defined( __HIP_DEVICE_COMPILE__ )
    if( C )
    {
        const MfReq<T> *p = s->p;
        Px4 r[N];
#pragma unroll
        for( int k = 0; k < N; k++ )
            r[k] = load_px4( p->ref[0] + (long)( y[k] + s->l_row ) * p->stride + x[k] + s->l_col );
#pragma unroll
        for( int k = 0; k < N; k++ )
        {
            int v = p->fpelcmp_satd ? satd_partial_px4( s->l_f, r[k] ) : sad_partial_px4( s->l_f, r[k], (const T *)nullptr );
            v = (int)wave_sum_u32( (unsigned)( s->l_active ? v : 0 ) );
            c[k] = ( p->fpelcmp_satd ? v >> 1 : v ) + mef_bits_f( s, x[k], y[k] );
        }
        return;
    }
#endif
    for( int k = 0; k < N; k++ )
        c[k] = mef_cost_f( s, x[k], y[k] );
}
// ... and applied in order with strict '<' (COST_MV one after the other); candidates whose bit in `ok` is clear do not take part
template <int N, typename T, bool C>
BM_HD void mef_try_set( Mef<T, C> *s, const int x[N], const int y[N], unsigned ok = ~0u )
{
    int xx[N], yy[N], c[N];
    for( int k = 0; k < N; k++ )
    {
        const bool on = ( ok >> k ) & 1;
        xx[k] = on ? x[k] : s->bmx; yy[k] = on ? y[k] : s->bmy; // (a readable position for the ones left out)
    }
    mef_costs_f<N>( s, xx, yy, c );
    for( int k = 0; k < N; k++ )
        if( ( ( ok >> k ) & 1 ) && c[k] < s->bcost ) { s->bcost = c[k]; s->bmx = x[k]; s->bmy = y[k]; }
}
/* COST_MV_X4 relative to (omx, omy), candidates applied in order */
template <typename T, bool C>
BM_HD void mef_x4( Mef<T, C> *s, int omx, int omy, const int d[4][2] )
{
    int x[4], y[4];
    for( int k = 0; k < 4; k++ ) { x[k] = omx + d[k][0]; y[k] = omy + d[k][1]; }
    mef_try_set<4>( s, x, y );
}
template <typename T, bool C>
BM_HD void mef_cross( Mef<T, C> *s, int omx, int omy, int start, int x_max, int y_max ) /* CROSS, me.c:139-166 */
{
    const MfReq<T> *p = s->p;
    int i = start;
    if( x_max <= imin( p->lim_max[0] - omx, omx - p->lim_min[0] ) )
        for( ; i < x_max - 2; i += 4 )
        {
            const int d[4][2] = { { i, 0 }, { -i, 0 }, { i+2, 0 }, { -i-2, 0 } };
            mef_x4( s, omx, omy, d );
        }
    for( ; i < x_max; i += 2 )
    {
        if( omx + i <= p->lim_max[0] ) mef_try_f( s, omx + i, omy );
        if( omx - i >= p->lim_min[0] ) mef_try_f( s, omx - i, omy );
    }
    i = start;
    if( y_max <= imin( p->lim_max[1] - omy, omy - p->lim_min[1] ) )
        for( ; i < y_max - 2; i += 4 )
        {
            const int d[4][2] = { { 0, i }, { 0, -i }, { 0, i+2 }, { 0, -i-2 } };
            mef_x4( s, omx, omy, d );
        }
    for( ; i < y_max; i += 2 )
    {
        if( omy + i <= p->lim_max[1] ) mef_try_f( s, omx, omy + i );
        if( omy - i >= p->lim_min[1] ) mef_try_f( s, omx, omy - i );
    }
}

template <typename T, bool C>
BM_HD void mef_hex2( Mef<T, C> *s, int me_range ) /* the HEX branch incl. the square refine, me.c:344-420 */
{
    const int8_t hex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
    const uint8_t mod6m1[8] = { 5,0,1,2,3,4,5,0 };
    const int8_t square1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
    const int8_t first[6][2] = { {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2} };
    int bmx = s->bmx, bmy = s->bmy, bcost = s->bcost, dir = -1;
    {
        int x[6], y[6], c[6];
        for( int k = 0; k < 6; k++ ) { x[k] = bmx + first[k][0]; y[k] = bmy + first[k][1]; }
        mef_costs_f<6>( s, x, y, c );
        for( int k = 0; k < 6; k++ ) /* packed (cost<<3)+k+2 with COPY1_IF_LT: lowest cost, first on ties */
            if( c[k] < bcost ) { bcost = c[k]; dir = k; }
    }
    if( dir >= 0 )
    {
        bmx += hex2[dir+1][0]; bmy += hex2[dir+1][1];
        s->bmx = bmx; s->bmy = bmy;
        for( int i = ( me_range >> 1 ) - 1; i > 0 && mef_in_range( s, bmx, bmy ); i-- )
        {
            int best = -1;
            int x[3], y[3], c[3];
            for( int k = 0; k < 3; k++ ) { x[k] = bmx + hex2[dir+k][0]; y[k] = bmy + hex2[dir+k][1]; }
            mef_costs_f<3>( s, x, y, c );
            for( int k = 0; k < 3; k++ )
                if( c[k] < bcost ) { bcost = c[k]; best = k; }
            if( best < 0 )
                break;
            dir += best - 1;
            dir = mod6m1[dir+1];
            bmx += hex2[dir+1][0]; bmy += hex2[dir+1][1];
        }
    }
    int sq = 0;
    {
        int x[8], y[8], c[8];
        for( int k = 0; k < 8; k++ ) { x[k] = bmx + square1[k + 1][0]; y[k] = bmy + square1[k + 1][1]; }
        mef_costs_f<8>( s, x, y, c );
        for( int k = 0; k < 8; k++ )
            if( c[k] < bcost ) { bcost = c[k]; sq = k + 1; }
    }
    s->bmx = bmx + square1[sq][0]; s->bmy = bmy + square1[sq][1]; s->bcost = bcost;
}

typedef struct { int sad; int mx, my; } mef_mvsad;

template <typename T, bool C>
BM_HD void mef_refine_subpel( Mef<T, C> *s, int mv[2], int *cost, int *cost_mv, int hpel_iters, int qpel_iters )
{
    const MfReq<T> *p = s->p;
    int bmx = mv[0], bmy = mv[1], bcost = *cost;
    if( hpel_iters )
    {
        if( p->subpel_refine < 3 )
        {
            int mx = clip3( p->mvp[0], p->spel_min[0] + 2, p->spel_max[0] - 2 );
            int my = clip3( p->mvp[1], p->spel_min[1] + 2, p->spel_max[1] - 2 );
            if( ( mx - bmx ) | ( my - bmy ) )
            {
                int c = mef_cost_q( s, mx, my, 0 );
                if( c < bcost ) { bcost = c; bmx = mx; bmy = my; }
            }
        }
        const int d2[4][2] = { {0,-2}, {0,2}, {-2,0}, {2,0} };
        for( int i = hpel_iters; i > 0; i-- )
        {
            int best = -1, omx = bmx, omy = bmy;
            int x[4], y[4], c[4];
            for( int k = 0; k < 4; k++ ) { x[k] = omx + d2[k][0]; y[k] = omy + d2[k][1]; }
            mef_costs_q<4>( s, x, y, 0, c );
            for( int k = 0; k < 4; k++ )
                if( c[k] < bcost ) { bcost = c[k]; best = k; }
            if( best < 0 )
                break;
            bmx = omx + d2[best][0]; bmy = omy + d2[best][1];
        }
    }
    if( p->mbcmp_satd != p->fpelcmp_satd ) /* h->pixf.mbcmp_unaligned[0] != h->pixf.fpelcmp[0] */
        bcost = mef_cost_q( s, bmx, bmy, 1 );
    const int d1[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
    if( p->subpel_refine != 1 )
    {
        int bdir = -1;
        for( int i = qpel_iters; i > 0; i-- )
        {
            if( bmy <= p->spel_min[1] || bmy >= p->spel_max[1] || bmx <= p->spel_min[0] || bmx >= p->spel_max[0] )
                break;
            int odir = bdir, omx = bmx, omy = bmy;
            int x[4], y[4], c[4];
            for( int k = 0; k < 4; k++ )
            {
                const bool back = ( k ^ 1 ) == odir; // the point the last step came from is not costed again (it is read, nothing more)
                x[k] = back ? omx : omx + d1[k][0]; y[k] = back ? omy : omy + d1[k][1];
            }
            mef_costs_q<4>( s, x, y, 1, c );
            for( int k = 0; k < 4; k++ )
            {
                if( ( k ^ 1 ) == odir )
                    continue;
                if( c[k] < bcost ) { bcost = c[k]; bmx = x[k]; bmy = y[k]; bdir = k; }
            }
            if( bmx == omx && bmy == omy )
                break;
        }
    }
    else if( bmy > p->spel_min[1] && bmy < p->spel_max[1] && bmx > p->spel_min[0] && bmx < p->spel_max[0] )
    {
        int omx = bmx, omy = bmy; /* subme 1: one quarter-pel diamond with fpelcmp */
        int x[4], y[4], c[4];
        for( int k = 0; k < 4; k++ ) { x[k] = omx + d1[k][0]; y[k] = omy + d1[k][1]; }
        mef_costs_q<4>( s, x, y, 0, c );
        for( int k = 0; k < 4; k++ )
            if( c[k] < bcost ) { bcost = c[k]; bmx = x[k]; bmy = y[k]; }
    }
    mv[0] = bmx; mv[1] = bmy; *cost = bcost;
    *cost_mv = mef_bits_q( s, bmx, bmy );
}

#if defined( __HIP_DEVICE_COMPILE__ )
// smallest value over the wave, wave-uniform (DPP inside the 16-lane rows, the four row results as scalars)
__device__ __forceinline__ unsigned mef_wave_min_u32( unsigned v )
{
    auto mn = []( unsigned a, unsigned b ) { return a < b ? a : b; };
    v = mn( v, (unsigned)dpp_mov<DPP_QUAD_XOR1>( (int)v ) );
    v = mn( v, (unsigned)dpp_mov<DPP_QUAD_XOR2>( (int)v ) );
    v = mn( v, (unsigned)dpp_mov<0x141>( (int)v ) ); // row_half_mirror
    v = mn( v, (unsigned)dpp_mov<0x140>( (int)v ) ); // row_mirror
    return mn( mn( (unsigned)__builtin_amdgcn_readlane( (int)v, 0 ), (unsigned)__builtin_amdgcn_readlane( (int)v, 16 ) ),
               mn( (unsigned)__builtin_amdgcn_readlane( (int)v, 32 ), (unsigned)__builtin_amdgcn_readlane( (int)v, 48 ) ) );
}
// The exhaustive scan of ESA (me.c:620-660) by a whole wave: the candidates of the window in scan order, 64 per step -- whatever row
// they are in --, every lane the full SAD of its own candidate with the source block held in registers (the candidates of a step are
// neighbours: their loads share cache lines), then one packed minimum ( cost << 6 | lane: the earliest of equal costs ).  Skipping
// rows whose vertical mv cost alone reaches the best cost, as the reference does, cannot change the result (every cost of such a row
// is at least that) and is left out.
template <typename T, int BW, int BH>
__device__ __forceinline__ void mef_esa_scan_wave( Mef<T, true> *s, int min_x, int min_y, int max_y, int width )
{
    const MfReq<T> *p = s->p;
    const int lane = threadIdx.x & 63;
    Px4 fe[BH][BW / 4];
#pragma unroll
    for( int y = 0; y < BH; y++ )
#pragma unroll
        for( int k = 0; k < BW / 4; k++ )
            fe[y][k] = load_px4( p->fenc + (long)y * p->fenc_stride + 4 * k );
    const int total = ( max_y - min_y + 1 ) * width;
    for( int base = 0; base < total; base += 64 )
    {
        const int t = base + lane < total ? base + lane : total - 1;
        const int row = t / width, col = t - row * width;
        const T *b = p->ref[0] + (long)( min_y + row ) * p->stride + min_x + col;
        int acc = 0;
#pragma unroll
        for( int y = 0; y < BH; y++ )
        {
            // (the row in one load where it is 16 or 8 bytes: see mf_sad)
            if constexpr( sizeof( T ) == 1 && BW == 16 )
            {
                typedef uint32_t u32x4_u __attribute__( ( ext_vector_type( 4 ), aligned( 1 ) ) );
                const u32x4_u w = *(const u32x4_u *)( b + (long)y * p->stride );
                acc = (int)__builtin_amdgcn_sad_u8( fe[y][0].raw, w.x, (unsigned)acc ); acc = (int)__builtin_amdgcn_sad_u8( fe[y][1].raw, w.y, (unsigned)acc );
                acc = (int)__builtin_amdgcn_sad_u8( fe[y][2].raw, w.z, (unsigned)acc ); acc = (int)__builtin_amdgcn_sad_u8( fe[y][3].raw, w.w, (unsigned)acc );
            }
            else if constexpr( sizeof( T ) == 1 && BW == 8 )
            {
                typedef uint32_t u32x2_u __attribute__( ( ext_vector_type( 2 ), aligned( 1 ) ) );
                const u32x2_u w = *(const u32x2_u *)( b + (long)y * p->stride );
                acc = (int)__builtin_amdgcn_sad_u8( fe[y][0].raw, w.x, (unsigned)acc ); acc = (int)__builtin_amdgcn_sad_u8( fe[y][1].raw, w.y, (unsigned)acc );
            }
            else
            {
#pragma unroll
                for( int k = 0; k < BW / 4; k++ )
                    acc += sad_partial_px4( fe[y][k], load_px4( b + (long)y * p->stride + 4 * k ), (const T *)nullptr );
            }
        }
        const int c = acc + p->cost_mv[4 * ( min_x + col ) - p->mvp[0]] + p->cost_mv[4 * ( min_y + row ) - p->mvp[1]];
        const unsigned key = mef_wave_min_u32( base + lane < total ? ( (unsigned)c << 6 ) | (unsigned)lane : 0xFFFFFFFFu );
        const int cmin = (int)( key >> 6 );
        if( cmin < s->bcost )
        {
            const int tt = base + (int)( key & 63 ), r2 = tt / width;
            s->bcost = cmin; s->bmx = min_x + tt - r2 * width; s->bmy = min_y + r2;
        }
    }
}
// The same scan for 8-bit samples with v_qsad_pk_u16_u8: ONE instruction gives the SADs of four source samples against the four byte
// positions of an 8-byte window -- four horizontally adjacent candidates at once, accumulated as four packed 16-bit sums (a 16x16 block
// of 8-bit samples cannot exceed 65 280).  A lane takes four neighbouring candidates (the width of the window is a multiple of four, so
// they share a row), a step 256 candidates: a quarter of the SAD instructions and of the steps of the form above.  The winner is the
// smallest ( cost << 8 | place in the step ): the earliest candidate among equal costs, as scanning in order with strict '<' keeps.
template <int BW, int BH>
__device__ __forceinline__ void mef_esa_scan_wave_q( Mef<uint8_t, true> *s, int min_x, int min_y, int max_y, int width )
{
    typedef uint32_t u32x4_u __attribute__( ( ext_vector_type( 4 ), aligned( 1 ) ) );
    typedef uint32_t u32x3_u __attribute__( ( ext_vector_type( 3 ), aligned( 1 ) ) );
    typedef uint32_t u32x2_u __attribute__( ( ext_vector_type( 2 ), aligned( 1 ) ) );
    typedef uint32_t u32x1_u __attribute__( ( aligned( 1 ) ) );
    const MfReq<uint8_t> *p = s->p;
    const int lane = threadIdx.x & 63;
    uint32_t fe[BH][BW / 4];
#pragma unroll
    for( int y = 0; y < BH; y++ )
#pragma unroll
        for( int k = 0; k < BW / 4; k++ )
            fe[y][k] = *(const u32x1_u *)( p->fenc + (long)y * p->fenc_stride + 4 * k );
    const int total = ( max_y - min_y + 1 ) * width;
    for( int base = 0; base < total; base += 256 )
    {
        const bool live = base + 4 * lane < total;
        const int t = live ? base + 4 * lane : total - 4;
        const int row = t / width, col = t - row * width;
        const uint8_t *b = p->ref[0] + (long)( min_y + row ) * p->stride + min_x + col;
        unsigned long long acc = 0;
#pragma unroll
        for( int y = 0; y < BH; y++ )
        {
            // the BW + 3 samples the four candidates' rows cover, as BW / 4 + 1 dwords (the last one's top byte belongs to nobody)
            uint32_t w[BW / 4 + 1];
            const uint8_t *r = b + (long)y * p->stride;
            if constexpr( BW == 16 )
            {
                const u32x4_u v = *(const u32x4_u *)r;
                w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; w[4] = *(const u32x1_u *)( r + 16 );
            }
            else if constexpr( BW == 8 )
            {
                const u32x3_u v = *(const u32x3_u *)r;
                w[0] = v.x; w[1] = v.y; w[2] = v.z;
            }
            else
            {
                const u32x2_u v = *(const u32x2_u *)r;
                w[0] = v.x; w[1] = v.y;
            }
#pragma unroll
            for( int k = 0; k < BW / 4; k++ )
                acc = __builtin_amdgcn_qsad_pk_u16_u8( ( (unsigned long long)w[k + 1] << 32 ) | w[k], fe[y][k], acc );
        }
        const int ycost = p->cost_mv[4 * ( min_y + row ) - p->mvp[1]];
        unsigned key = 0xFFFFFFFFu;
#pragma unroll
        for( int j = 3; j >= 0; j-- )
        {
            const int c = (int)( ( acc >> ( 16 * j ) ) & 0xFFFF ) + p->cost_mv[4 * ( min_x + col + j ) - p->mvp[0]] + ycost;
            const unsigned kj = ( (unsigned)c << 8 ) | (unsigned)( 4 * lane + j );
            key = kj < key ? kj : key;
        }
        key = mef_wave_min_u32( live ? key : 0xFFFFFFFFu );
        const int cmin = (int)( key >> 8 );
        if( cmin < s->bcost )
        {
            const int tt = base + (int)( key & 255 ), r2 = tt / width;
            s->bcost = cmin; s->bmx = min_x + tt - r2 * width; s->bmy = min_y + r2;
        }
    }
}
template <typename T>
__device__ __forceinline__ void mef_esa_scan_dispatch( Mef<T, true> *s, int min_x, int min_y, int max_y, int width )
{
    if constexpr( sizeof( T ) == 1 )
    {
        switch( s->p->i_pixel )
        {
            case 0: mef_esa_scan_wave_q<16, 16>( s, min_x, min_y, max_y, width ); break;
            case 1: mef_esa_scan_wave_q<16, 8>( s, min_x, min_y, max_y, width ); break;
            case 2: mef_esa_scan_wave_q<8, 16>( s, min_x, min_y, max_y, width ); break;
            case 3: mef_esa_scan_wave_q<8, 8>( s, min_x, min_y, max_y, width ); break;
            case 4: mef_esa_scan_wave_q<8, 4>( s, min_x, min_y, max_y, width ); break;
            case 5: mef_esa_scan_wave_q<4, 8>( s, min_x, min_y, max_y, width ); break;
            default: mef_esa_scan_wave_q<4, 4>( s, min_x, min_y, max_y, width ); break;
        }
        return;
    }
    switch( s->p->i_pixel )
    {
        case 0: mef_esa_scan_wave<T, 16, 16>( s, min_x, min_y, max_y, width ); break;
        case 1: mef_esa_scan_wave<T, 16, 8>( s, min_x, min_y, max_y, width ); break;
        case 2: mef_esa_scan_wave<T, 8, 16>( s, min_x, min_y, max_y, width ); break;
        case 3: mef_esa_scan_wave<T, 8, 8>( s, min_x, min_y, max_y, width ); break;
        case 4: mef_esa_scan_wave<T, 8, 4>( s, min_x, min_y, max_y, width ); break;
        case 5: mef_esa_scan_wave<T, 4, 8>( s, min_x, min_y, max_y, width ); break;
        default: mef_esa_scan_wave<T, 4, 4>( s, min_x, min_y, max_y, width ); break;
    }
}
template <typename T>
__device__ __forceinline__ void mef_esa_scan_dispatch( Mef<T, false> *, int, int, int, int ) {}
#endif

// METHODS: which branches are compiled in -- bit 0 the pattern searches (DIA, HEX, UMH), bit 1 the exhaustive ones (ESA, TESA).  The
// device builds one kernel per class: the exhaustive scan keeps the source block in registers, which would cost the pattern searches
// half their occupancy.
template <typename T, class Coop = CoopNone, int METHODS = 3>
BM_HD void mf_me_search_full( const MfReq<T> *p, const int16_t (*mvc)[2], int n_mvc, int out[4], Coop coop = Coop() )
{
    const uint8_t mef_size[7][2] = { {16,16}, {16,8}, {8,16}, {8,8}, {8,4}, {4,8}, {4,4} };
    const uint8_t mef_subpel_iterations[12][4] = /* me.c:38-50 */
        { {0,0,0,0}, {1,1,0,0}, {0,1,1,0}, {0,2,1,0}, {0,2,1,1}, {0,2,1,2}, {0,0,2,2}, {0,0,2,2}, {0,0,4,10}, {0,0,4,10}, {0,0,4,10}, {0,0,4,10} };
    Mef<T, ( Coop::W > 1 )> S, *s = &S;
    s->p = p; s->bw = mef_size[p->i_pixel][0]; s->bh = mef_size[p->i_pixel][1];
#if defined( __HIP_DEVICE_COMPILE__ )
    if( Coop::W > 1 )
    {
        const int l = coop.lane(), tiles_per_row = s->bw >> 2, q = l >> 2;
        const int ty = q / tiles_per_row, tx = q - ty * tiles_per_row;
        s->l_active = 4 * ty < s->bh;
        s->l_row = s->l_active ? 4 * ty + ( l & 3 ) : ( l & 3 ); // (lanes beyond the block read its first tile: always addressable)
        s->l_col = s->l_active ? 4 * tx : 0;
        s->l_f = load_px4( p->fenc + (long)s->l_row * p->fenc_stride + s->l_col );
    }
#endif
    const int mv_x_min = p->lim_min[0], mv_y_min = p->lim_min[1], mv_x_max = p->lim_max[0], mv_y_max = p->lim_max[1];
    int me_range = p->me_range;
    int bpred_cost = MF_COST_MAX, bpred_mx = 0, bpred_my = 0, pmx, pmy, pmv_nonzero;
    s->bcost = MF_COST_MAX; s->bmx = s->bmy = 0;
    int pmv_q[2];

    if( p->subpel_refine >= 3 )
    {
        bpred_mx = clip3( p->mvp[0], 4*mv_x_min, 4*mv_x_max );
        bpred_my = clip3( p->mvp[1], 4*mv_y_min, 4*mv_y_max );
        pmv_q[0] = bpred_mx; pmv_q[1] = bpred_my;
        pmv_nonzero = ( bpred_mx | bpred_my ) != 0;
        pmx = ( bpred_mx + 2 ) >> 2; pmy = ( bpred_my + 2 ) >> 2;
        bpred_cost = mef_cost_q( s, bpred_mx, bpred_my, 0 );
        const int pmv_cost = bpred_cost;
        for( int i = 0; i < n_mvc; i++ ) /* x264_predictor_clip + the packed minimum (first of equal costs wins) */
        {
            int mx = mvc[i][0], my = mvc[i][1];
            if( ( !mx && !my ) || ( mx == pmv_q[0] && my == pmv_q[1] ) )
                continue;
            mx = clip3( mx, 4*mv_x_min, 4*mv_x_max ); my = clip3( my, 4*mv_y_min, 4*mv_y_max );
            int c = mef_cost_q( s, mx, my, 0 );
            if( c < bpred_cost ) { bpred_cost = c; bpred_mx = mx; bpred_my = my; }
        }
        s->bmx = ( bpred_mx + 2 ) >> 2; s->bmy = ( bpred_my + 2 ) >> 2;
        if( ( bpred_mx | bpred_my ) & 3 )
            mef_try_f( s, s->bmx, s->bmy ); /* bcost is MF_COST_MAX here: always taken */
        else
            s->bcost = bpred_cost;
        if( pmv_nonzero )
        {
            if( s->bmx | s->bmy ) mef_try_f( s, 0, 0 );
        }
        else if( pmv_cost < s->bcost ) { s->bcost = pmv_cost; s->bmx = 0; s->bmy = 0; }
    }
    else
    {
        s->bmx = pmx = clip3( ( p->mvp[0] + 2 ) >> 2, mv_x_min, mv_x_max );
        s->bmy = pmy = clip3( ( p->mvp[1] + 2 ) >> 2, mv_y_min, mv_y_max );
        pmv_q[0] = pmx; pmv_q[1] = pmy; /* full-pel units in this branch */
        pmv_nonzero = ( pmx | pmy ) != 0;
        s->bcost = mef_fpelcmp( s, p->ref[0] + (long)s->bmy * p->stride + s->bmx, p->stride ); /* no mv bits for the rounded predictor */
        for( int i = 0; i < n_mvc; i++ ) /* x264_predictor_roundclip */
        {
            int mx = ( mvc[i][0] + 2 ) >> 2, my = ( mvc[i][1] + 2 ) >> 2;
            if( ( !mx && !my ) || ( mx == pmx && my == pmy ) )
                continue;
            mx = clip3( mx, mv_x_min, mv_x_max ); my = clip3( my, mv_y_min, mv_y_max );
            int c = mef_cost_f( s, mx, my );
            if( c < s->bcost ) { s->bcost = c; s->bmx = mx; s->bmy = my; }
        }
        if( pmv_nonzero )
            mef_try_f( s, 0, 0 );
    }

    // (a build for one class of methods never sees a request of the other: tells the compiler to drop those branches)
    if( ( !( METHODS & 2 ) && p->me_method >= 3 ) || ( !( METHODS & 1 ) && p->me_method < 3 ) )
        __builtin_unreachable();
    switch( p->me_method )
    {
        case 0: /* DIA, me.c:322-342 */
        {
            const int d[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
            int i = me_range;
            do
            {
                int best = -1, bmx = s->bmx, bmy = s->bmy;
                int x[4], y[4], c[4];
                for( int k = 0; k < 4; k++ ) { x[k] = bmx + d[k][0]; y[k] = bmy + d[k][1]; }
                mef_costs_f<4>( s, x, y, c );
                for( int k = 0; k < 4; k++ )
                    if( c[k] < s->bcost ) { s->bcost = c[k]; best = k; }
                if( best < 0 )
                    break;
                s->bmx = bmx + d[best][0]; s->bmy = bmy + d[best][1];
            } while( --i && mef_in_range( s, s->bmx, s->bmy ) );
            break;
        }
        case 1:
            mef_hex2( s, me_range );
            break;
        case 2:
        {
            // Uneven multi-hexagon search (behaviour of me.c:422-618) as a programme over probe sets.  The search keeps a best vector and
            // its cost; every stage probes a fixed set of offsets around an origin (probes apply in order, strict '<').  Three cost
            // marks decide the exits: `entry` (before anything of this method ran), `settled` (after the small diamonds around the
            // predictor, zero and -- if that moved the best -- the new best).  "Nothing better since a mark" is the test bcost == mark.
            // Thresholds are stated for a 16x16 block and scale down with the partition's area class.
            static const int8_t kDiamond[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
            // the eight points at city-block distance 2, the eight knight moves, the four diagonal corners at distance 2 (two sets of four each)
            static const int8_t kProbe[5][4][2] = { { {0,-2}, {-1,-1}, {1,-1}, {-2,0} }, { {2,0}, {-1,1}, {1,1}, {0,2} },
                                                    { {-1,-2}, {1,-2}, {-2,-1}, {2,-1} }, { {-2,1}, {2,1}, {-1,2}, {1,2} },
                                                    { {-2,-2}, {-2,2}, {2,-2}, {2,2} } };
            auto probe = [&]( int cx, int cy, const int8_t ( *set )[2] ) {
                int x[4], y[4];
                for( int k = 0; k < 4; k++ ) { x[k] = cx + set[k][0]; y[k] = cy + set[k][1]; }
                mef_try_set<4>( s, x, y );
            };
            const int area_class = (int)( ( 0x4332110u >> ( 4 * p->i_pixel ) ) & 15 ); // 16x16 0, 16x8 / 8x16 1, 8x8 2, 8x4 / 4x8 3, 4x4 4
            auto good_match = [&]( int limit16 ) { return s->bcost < ( limit16 >> area_class ); };
            const int entry = s->bcost;
            probe( pmx, pmy, kDiamond );
            if( pmx | pmy )
                probe( 0, 0, kDiamond );
            if( p->i_pixel == 6 ) // 4x4 partitions go straight to the hexagon
            {
                mef_hex2( s, me_range );
                break;
            }
            const int settled = s->bcost;
            if( ( s->bmx | s->bmy ) && ( s->bmx != pmx || s->bmy != pmy ) )
                probe( s->bmx, s->bmy, kDiamond );
            int cross_first = s->bcost == settled ? 3 : 1; // the cross skips what the diamond around the origin has just covered
            const int ox = s->bmx, oy = s->bmy;            // origin of every stage up to the hexagon rings
            bool finished = false;
            if( s->bcost == settled && good_match( 2000 ) )
            {
                // the start is already a good match and its diamond found nothing: look a little further before paying for the full programme
                probe( ox, oy, kProbe[0] );
                probe( ox, oy, kProbe[1] );
                if( s->bcost == entry && good_match( 500 ) )
                    finished = true;
                else if( s->bcost == settled )
                {
                    const int reach = ( me_range >> 1 ) | 1;
                    mef_cross( s, ox, oy, 3, reach, reach );
                    probe( ox, oy, kProbe[2] );
                    probe( ox, oy, kProbe[3] );
                    if( s->bcost == settled )
                        finished = true;
                    else
                        cross_first = reach + 2;
                }
            }
            if( finished )
                break;
            if( n_mvc )
            {
                // the range adapts to how well the predictors agree (spread per predictor pair, 4 classes) and to how good the match
                // is (4 classes): quarters of the configured range, one nibble per (agreement, match) pair
                int spread, pairs = 1;
                const int to_mvp = mf_abs( p->mvp[0] - mvc[0][0] ) + mf_abs( p->mvp[1] - mvc[0][1] );
                if( n_mvc == 1 )
                    spread = p->i_pixel == 0 ? 25 : to_mvp;
                else
                {
                    pairs = n_mvc - 1 + ( p->i_pixel != 0 );
                    spread = p->i_pixel != 0 ? to_mvp : 0;
                    for( int i = 0; i + 1 < n_mvc; i++ )
                        spread += mf_abs( mvc[i][0] - mvc[i + 1][0] ) + mf_abs( mvc[i][1] - mvc[i + 1][1] );
                }
                const int match = good_match( 1000 ) ? 0 : good_match( 2000 ) ? 1 : good_match( 4000 ) ? 2 : 3;
                const int agree = spread < 10 * pairs ? 0 : spread < 20 * pairs ? 1 : spread < 40 * pairs ? 2 : 3;
                static const uint16_t kQuarters[4] = { 0x4433, 0x4443, 0x5444, 0x6544 }; // [agree], nibble `match`
                me_range = me_range * (int)( ( kQuarters[agree] >> ( 4 * match ) ) & 15 ) >> 2;
            }
            mef_cross( s, ox, oy, cross_first, me_range, me_range >> 1 );
            probe( ox, oy, kProbe[4] );
            // sixteen-point hexagon rings of growing radius around the best so far: (0, -4) and (0, 4), then the pairs (-x, y), (x, y)
            // for y = -3 .. 3 with x = 2 on the two outer rows and 4 between them, all scaled by the ring number
            {
                const int rx = s->bmx, ry = s->bmy;
                int ring = 1;
                do // (at least one ring, also where the adapted range is below four)
                {
                    const int room = imin( imin( mv_x_max - rx, rx - mv_x_min ), imin( mv_y_max - ry, ry - mv_y_min ) );
                    const bool clipped = 4 * ring > room;
                    for( int j0 = 0; j0 < 16; j0 += 8 ) // (two sets of eight: sixteen blocks of samples in flight do not fit the registers)
                    {
                        int x[8], y[8];
                        unsigned ok = 0;
                        for( int k = 0; k < 8; k++ )
                        {
                            const int j = j0 + k, row = ( j - 2 ) >> 1;
                            const int dy = j < 2 ? ( j ? 4 : -4 ) : row - 3;
                            const int ax = j < 2 ? 0 : ( row == 0 || row == 6 ) ? 2 : 4;
                            x[k] = rx + ( ( j & 1 ) ? ax : -ax ) * ring; y[k] = ry + dy * ring;
                            if( !clipped || mef_in_range( s, x[k], y[k] ) ) ok |= 1u << k;
                        }
                        mef_try_set<8>( s, x, y, ok );
                    }
                } while( ++ring <= me_range >> 2 );
            }
            if( s->bmy <= mv_y_max && s->bmy >= mv_y_min && s->bmx <= mv_x_max && s->bmx >= mv_x_min )
                mef_hex2( s, me_range );
            break;
        }
        default: /* ESA (3) / TESA (4), me.c:620-772 */
        {
            if( !( METHODS & 2 ) )
                break; // (compiled out of the pattern-search build)
            const int min_x = imax( s->bmx - me_range, mv_x_min ), min_y = imax( s->bmy - me_range, mv_y_min );
            const int max_x = imin( s->bmx + me_range, mv_x_max ), max_y = imin( s->bmy + me_range, mv_y_max );
            const int width = ( max_x - min_x + 3 ) & ~3;
#if defined( __HIP_DEVICE_COMPILE__ )
            if( p->me_method == 3 && Coop::W > 1 && !p->fpelcmp_satd )
            {
                mef_esa_scan_dispatch( s, min_x, min_y, max_y, width );
                break;
            }
#endif
            if( p->me_method == 3 )
            {
                /* successive elimination only discards candidates that cannot beat the current best (sum|d| >= |sum d|),
                 * so the result is the plain exhaustive scan in its order -- including the up to three columns past
                 * max_x that rounding the width to a multiple of four adds */
                for( int my = min_y; my <= max_y; my++ )
                {
                    if( s->bcost <= p->cost_mv[4*my - p->mvp[1]] )
                        continue;
                    for( int base = 0; base < width; base += Coop::W )
                    {
                        int idx = base + coop.lane(), c = MF_COST_MAX;
                        if( idx < width )
                            c = mef_cost_f_lane( s, min_x + idx, my );
                        coop.argmin( c, idx );
                        if( c < s->bcost ) { s->bcost = c; s->bmx = min_x + idx; s->bmy = my; }
                    }
                }
                break;
            }
            /* TESA: ADS threshold, SAD threshold, keep the best few SADs, then SATD */
            const uint16_t *sums_base = p->integral;
            int enc_dc[4];
            const int small = p->i_pixel > 3; /* sad_size: 8x8 quadrants for sizes >= 8x8, else 4x4 */
            int delta = small ? 4 : 8;
            {
                const T *f = p->fenc;
                const int q[4][2] = { {0,0}, {delta,0}, {0,delta}, {delta,delta} };
                for( int k = 0; k < 4; k++ )
                {
                    int sum = 0;
                    for( int y = 0; y < delta; y++ )
                        for( int x = 0; x < delta; x++ )
                            sum += f[( q[k][1] + y ) * p->fenc_stride + q[k][0] + x];
                    enc_dc[k] = sum;
                }
            }
            if( small )
                sums_base += p->integral_lower;
            int ads_n; /* ads[i_pixel]: 16x16 -> ads4; 16x8, 8x16, 8x4, 4x8 -> ads2; 8x8, 4x4 -> ads1 */
            if( p->i_pixel == 0 ) ads_n = 4; else if( p->i_pixel == 3 || p->i_pixel == 6 ) ads_n = 1; else ads_n = 2;
            if( p->i_pixel == 0 || p->i_pixel == 2 || p->i_pixel == 5 )
                delta *= p->stride;
            if( p->i_pixel == 2 || p->i_pixel == 5 )
                enc_dc[1] = enc_dc[2];
            // candidate list of the SAD stage: in the request's scratch area (MF_TESA_ROWS_MAX x MF_TESA_WIDTH_MAX entries)
            mef_mvsad *mvsads = (mef_mvsad *)p->scratch;
            int16_t *xs = coop.xs();
            // the horizontal vector costs of the window's columns: a table for the one-thread form; a wave keeps the (at most three) columns each
            // lane looks at in registers (a table indexed by the lane lives in scratch memory: a memory round trip per ads step)
            // (NCH ads steps cover the widest window: three in a wave of 64; the thread-group form of the host tests has more, narrower ones)
            constexpr int NCH = Coop::W == 1 ? 1 : ( MF_TESA_WIDTH_MAX + Coop::W - 1 ) / Coop::W;
            uint16_t cost_fpel_mvx[Coop::W == 1 ? MF_TESA_WIDTH_MAX + 4 : 1];
            int cmx_lane[NCH] = { 0 };
            if( Coop::W == 1 )
                for( int x = 0; x < width; x++ )
                    cost_fpel_mvx[x] = p->cost_mv[4*( min_x + x ) - p->mvp[0]];
            else
                for( int k = 0; k < NCH; k++ )
                    if( k * Coop::W + coop.lane() < width )
                        cmx_lane[k] = p->cost_mv[4*( min_x + k * Coop::W + coop.lane() ) - p->mvp[0]];
            int nmvsad = 0;
            int sad_thresh = me_range <= 16 ? 10 : me_range <= 24 ? 11 : 12;
            int bsad = mf_sad( p->fenc, p->fenc_stride, p->ref[0] + (long)s->bmy * p->stride + s->bmx, p->stride, s->bw, s->bh ) + mef_bits_f( s, s->bmx, s->bmy );
            // What a row reads before any of its decisions -- its vertical vector cost and, in a wave, the box sums of its first 64 columns --
            // is requested while the row before it is still being worked on: a row is then one memory round trip (its SADs) instead of three.
            struct RowAhead { int ycost, s0, s1, s2, s3; };
            auto row_ahead = [&]( int my ) {
                RowAhead r = { 0, 0, 0, 0, 0 };
                r.ycost = p->cost_mv[4*my - p->mvp[1]];
                const int i = coop.lane();
                if( Coop::W > 1 && i < width )
                {
                    const uint16_t *sums = sums_base + min_x + (long)my * p->stride;
                    r.s0 = sums[i];
                    if( ads_n == 2 ) r.s1 = sums[i + delta];
                    else if( ads_n == 4 ) { r.s1 = sums[i + 8]; r.s2 = sums[i + delta]; r.s3 = sums[i + delta + 8]; }
                }
                return r;
            };
            RowAhead row = row_ahead( min_y ), row_next = row;
            for( int my = min_y; my <= max_y; my++, row = row_next )
            {
                if( my < max_y )
                    row_next = row_ahead( my + 1 );
                int ycost = row.ycost;
                if( bsad <= ycost )
                    continue;
                bsad -= ycost;
                /* ads stage (pixel.c:756-803): the candidates of the row below the threshold, kept in scan order -- with several
                 * lanes through a ballot and the count of kept candidates in front of each */
                int xn = 0;
                {
                    const uint16_t *sums = sums_base + min_x + (long)my * p->stride;
                    const int thresh = bsad * 17 >> 4;
                    coop.sync();
                    for( int base = 0; base < width; base += Coop::W )
                    {
                        const int i = base + coop.lane();
                        bool keep = false;
                        if( i < width )
                        {
                            int cmx;
                            if constexpr( Coop::W == 1 ) cmx = cost_fpel_mvx[i];
                            else if constexpr( NCH == 3 ) cmx = base == 0 ? cmx_lane[0] : base == Coop::W ? cmx_lane[1] : cmx_lane[2]; // (selects, not an indexed array)
                            else cmx = cmx_lane[base / Coop::W];
                            int ads;
                            if( Coop::W > 1 && base == 0 )
                            {
                                ads = mf_abs( enc_dc[0] - row.s0 ) + cmx;
                                if( ads_n == 2 ) ads += mf_abs( enc_dc[1] - row.s1 );
                                else if( ads_n == 4 ) ads += mf_abs( enc_dc[1] - row.s1 ) + mf_abs( enc_dc[2] - row.s2 ) + mf_abs( enc_dc[3] - row.s3 );
                            }
                            else
                            {
                                ads = mf_abs( enc_dc[0] - sums[i] ) + cmx;
                                if( ads_n == 2 ) ads += mf_abs( enc_dc[1] - sums[i + delta] );
                                else if( ads_n == 4 ) ads += mf_abs( enc_dc[1] - sums[i + 8] ) + mf_abs( enc_dc[2] - sums[i + delta] ) + mf_abs( enc_dc[3] - sums[i + delta + 8] );
                            }
                            keep = ads < thresh;
                        }
                        const unsigned long long m = coop.ballot( keep );
                        if( keep )
                            xs[xn + mf_popc64( m & ( ( 1ull << coop.lane() ) - 1 ) )] = (int16_t)i;
                        xn += mf_popc64( m );
                    }
                    coop.sync();
                }
                /* SAD stage: the SADs of a chunk of survivors side by side, the running thresholds applied to them in scan order */
                for( int base = 0; base < xn; base += Coop::W )
                {
                    const int i = base + coop.lane();
                    int sad_mine = MF_COST_MAX;
                    if( i < xn )
                    {
                        /* the reference indexes its x-cost table with the offset from min_x here (me.c:671,688: cost_fpel_mvx[xs[i]],
                         * not cost_fpel_mvx[min_x + xs[i]] as in the ads call), so the SAD stage charges the cost of column xs[i] */
                        sad_mine = mf_sad( p->fenc, p->fenc_stride, p->ref[0] + (long)my * p->stride + min_x + xs[i], p->stride, s->bw, s->bh ) +
                                   p->cost_mv[4 * xs[i] - p->mvp[0]];
                    }
                    // The reference walks the chunk in order: a candidate stays if its SAD is below sad_thresh / 8 of the best SAD so far, and
                    // becomes the best if it is below that.  sad_thresh / 8 >= 1.25, so whatever does not stay cannot be a new best either:
                    // "the best so far" in front of candidate k is the minimum of bsad and of EVERY SAD in front of it -- a prefix minimum
                    // across the lanes -- and the places of those that stay follow from a ballot, as in the ads stage.  (Lane after lane through
                    // a broadcast this loop was a chain of ~500 LDS round trips per search.)
                    int before, chunk_best;
                    coop.min_scan( sad_mine, before, chunk_best );
                    const int best_before = before < bsad ? before : bsad;
                    const bool stay = i < xn && sad_mine < ( best_before * sad_thresh >> 3 );
                    const unsigned long long m = coop.ballot( stay );
                    if( stay )
                    {
                        const int at = nmvsad + mf_popc64( m & ( ( 1ull << coop.lane() ) - 1 ) );
                        mvsads[at].sad = sad_mine + ycost; mvsads[at].mx = min_x + xs[i]; mvsads[at].my = my;
                    }
                    nmvsad += mf_popc64( m );
                    if( chunk_best < bsad ) bsad = chunk_best;
                }
                bsad += ycost;
            }
            // the survivors are thinned to half the range's worth before their full costs are taken (me.c:704-752)
            coop.sync(); // the list was written a lane per entry
            nmvsad = mf_thin_survivors( coop, mvsads, nmvsad, bsad, bsad * sad_thresh >> 3, me_range >> 1 );
            for( int i = 0; i < nmvsad; i++ )
                mef_try_f( s, mvsads[i].mx, mvsads[i].my );
            break;
        }
    }

    /* -> quarter-pel vector, me.c:774-789 */
    int mv[2], cost, cost_mv = 0;
    if( p->subpel_refine < 3 )
    {
        cost_mv = mef_bits_f( s, s->bmx, s->bmy );
        cost = s->bcost;
        if( s->bmx == pmv_q[0] && s->bmy == pmv_q[1] )
            cost += cost_mv;
        mv[0] = 4 * s->bmx; mv[1] = 4 * s->bmy;
    }
    else if( bpred_cost < s->bcost )
    {
        mv[0] = bpred_mx; mv[1] = bpred_my; cost = bpred_cost;
    }
    else
    {
        mv[0] = 4 * s->bmx; mv[1] = 4 * s->bmy; cost = s->bcost;
    }
    if( p->subpel_refine >= 2 )
        mef_refine_subpel( s, mv, &cost, &cost_mv, mef_subpel_iterations[p->subpel_refine][2], mef_subpel_iterations[p->subpel_refine][3] );
    out[0] = mv[0]; out[1] = mv[1]; out[2] = cost; out[3] = cost_mv;
}
} // namespace mefull

#ifdef __HIPCC__
// request table in device memory; candidates per request in mvc[i][MF_MVC_MAX][2]
#define MF_MVC_MAX 10
struct CoopWave
{
    static constexpr int W = 64;
    int16_t *xs_; // in LDS: the ads survivors of the current row, shared by the wave
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ void argmin( int &cost, int &idx ) const
    {
        unsigned long long key = ( (unsigned long long)(unsigned)cost << 32 ) | (unsigned)idx; // costs are non-negative and below 2^28
#pragma unroll
        for( int m = 1; m < 64; m <<= 1 )
        {
            const unsigned long long o = __shfl_xor( key, m, 64 );
            key = o < key ? o : key;
        }
        cost = (int)( key >> 32 ); idx = (int)(unsigned)key;
    }
    __device__ __forceinline__ unsigned long long ballot( bool p ) const { return __builtin_amdgcn_ballot_w64( p ); }
    __device__ __forceinline__ unsigned long long max64( unsigned long long v ) const
    {
#pragma unroll
        for( int m = 1; m < 64; m <<= 1 )
        {
            const unsigned long long o = __shfl_xor( v, m, 64 );
            v = o > v ? o : v;
        }
        return v;
    }
    __device__ __forceinline__ int bcast( int v, int l ) const { return __shfl( v, l, 64 ); }
    // prefix minimum across the wave without leaving the registers: the classic DPP scan (row shifts by 1, 2, 3, then 4 and 8 under bank
    // masks, then the two row broadcasts), the exclusive form one wave shift further; lanes without a source keep the identity
    __device__ __forceinline__ void min_scan( int v, int &before, int &all ) const
    {
        constexpr int ID = 1 << 28;
        auto mn = []( int a, int b ) { return a < b ? a : b; };
        int r = mn( v, __builtin_amdgcn_update_dpp( ID, v, 0x111, 0xF, 0xF, false ) );  // row_shr:1
        r = mn( r, __builtin_amdgcn_update_dpp( ID, v, 0x112, 0xF, 0xF, false ) );      // row_shr:2
        r = mn( r, __builtin_amdgcn_update_dpp( ID, v, 0x113, 0xF, 0xF, false ) );      // row_shr:3
        r = mn( r, __builtin_amdgcn_update_dpp( ID, r, 0x114, 0xF, 0xE, false ) );      // row_shr:4, banks 1-3
        r = mn( r, __builtin_amdgcn_update_dpp( ID, r, 0x118, 0xF, 0xC, false ) );      // row_shr:8, banks 2-3
        r = mn( r, __builtin_amdgcn_update_dpp( ID, r, 0x142, 0xA, 0xF, false ) );      // row_bcast:15 into rows 1 and 3
        r = mn( r, __builtin_amdgcn_update_dpp( ID, r, 0x143, 0xC, 0xF, false ) );      // row_bcast:31 into rows 2 and 3
        before = __builtin_amdgcn_update_dpp( ID, r, 0x138, 0xF, 0xF, false );          // wave_shr:1
        all = __builtin_amdgcn_readlane( r, 63 );
    }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ int16_t *xs() { return xs_; }
};

// The exhaustive methods (ESA, TESA): one wave per request.  The 64 lanes run the whole search in lock step (the predictor stage
// and the sub-pel refinement redundantly, their loads are broadcasts) and share the exhaustive scan: 64 candidates per step,
// ordered compaction of the ads survivors, first-in-scan-order ties -- bit-exact with the one-thread form below, which remains
// the device reference and serves DIA / HEX / UMH requests.
// (MF_COOP_WAVES: waves per SIMD the register allocation aims at.  The wave's life is a chain of ~100 memory round trips, so more of
// them resident hide more of it -- until the registers that no longer fit are memory traffic themselves.)
#ifndef MF_COOP_WAVES
#define MF_COOP_WAVES 5
#endif
template <typename T, int METHODS>
__global__ __launch_bounds__( 64, MF_COOP_WAVES ) void me_full_coop_kernel( const MfReq<T> *reqs, const int16_t *mvc, const int *n_mvc, const int *index, int n, int *out )
{
    __shared__ int16_t xs_lds[( METHODS & 2 ) ? MF_TESA_WIDTH_MAX + 64 : 2];
    if( (int)blockIdx.x >= n )
        return;
    const int i = index[blockIdx.x];
    CoopWave coop;
    coop.xs_ = xs_lds;
    int res[4];
    mefull::mf_me_search_full<T, CoopWave, METHODS>( &reqs[i], (const int16_t( * )[2])( mvc + (long)i * MF_MVC_MAX * 2 ), n_mvc[i], res, coop );
    if( ( threadIdx.x & 63 ) == 0 )
        for( int k = 0; k < 4; k++ )
            out[4 * i + k] = res[k];
}
// The request table of the public entry (x264hip_me_request, include/x264hip.h) turned into the kernels' own records ON the device: a thread
// per request.  The records carry eight pointers each; built on the host they were 200 bytes per request written into pinned memory and
// read back over the host link, twice the size of what the caller handed over.  scratch_off: per-request byte offset of a TESA
// request's candidate list (nullptr: request i takes slot i of `uniform_scratch` bytes); force_method >= 0: every request is searched
// with that method (device-resident tables are declared to hold one class of methods).
struct MeTranslate
{
    const void *fenc_plane, *ref[4];
    long fenc_stride, ref_stride, integral_lower;
    const uint16_t *integral, *cost_mv;
    char *scratch;
    const unsigned long long *scratch_off;
    unsigned long long uniform_scratch;
    int force_method, n;
};
template <typename T>
__global__ __launch_bounds__( 256 ) void me_translate_kernel( const x264hip_me_request *q_all, MeTranslate A, MfReq<T> *table, int16_t *mvc, int *n_mvc, int *index )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i >= A.n )
        return;
    const x264hip_me_request q = q_all[i];
    MfReq<T> r;
    r.i_pixel = q.i_pixel; r.me_method = A.force_method >= 0 ? A.force_method : q.me_method; r.subpel_refine = q.subpel_refine; r.me_range = q.me_range;
    r.mbcmp_satd = q.mbcmp_satd; r.fpelcmp_satd = q.fpelcmp_satd;
    r.fenc = (const T *)A.fenc_plane + (long)q.y * A.fenc_stride + q.x; r.fenc_stride = (int)A.fenc_stride;
#pragma unroll
    for( int k = 0; k < 4; k++ )
        r.ref[k] = (const T *)A.ref[k] + (long)q.y * A.ref_stride + q.x;
    r.stride = (int)A.ref_stride;
    r.integral = A.integral ? A.integral + (long)q.y * A.ref_stride + q.x : nullptr;
    r.integral_lower = A.integral_lower;
#pragma unroll
    for( int k = 0; k < 2; k++ )
    {
        r.mvp[k] = q.mvp[k]; r.lim_min[k] = q.lim_min[k]; r.lim_max[k] = q.lim_max[k];
        r.spel_min[k] = q.spel_min[k]; r.spel_max[k] = q.spel_max[k];
    }
    r.cost_mv = A.cost_mv;
    r.scratch = r.me_method == 4 ? A.scratch + ( A.scratch_off ? A.scratch_off[i] : (unsigned long long)i * A.uniform_scratch ) : nullptr;
    table[i] = r;
    n_mvc[i] = q.n_mvc < 0 ? 0 : q.n_mvc > X264HIP_ME_MVC_MAX ? X264HIP_ME_MVC_MAX : q.n_mvc;
#pragma unroll
    for( int k = 0; k < X264HIP_ME_MVC_MAX; k++ )
    {
        mvc[( (long)i * MF_MVC_MAX + k ) * 2] = q.mvc[k][0];
        mvc[( (long)i * MF_MVC_MAX + k ) * 2 + 1] = q.mvc[k][1];
    }
    if( index )
        index[i] = i;
}
// one thread per request, for the listed requests (index == nullptr: all)
template <typename T>
__global__ __launch_bounds__( 64 ) void me_full_list_kernel( const MfReq<T> *reqs, const int16_t *mvc, const int *n_mvc, const int *index, int n, int *out )
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if( k >= n )
        return;
    const int i = index[k];
    int res[4];
    mefull::mf_me_search_full<T>( &reqs[i], (const int16_t( * )[2])( mvc + (long)i * MF_MVC_MAX * 2 ), n_mvc[i], res );
    for( int j = 0; j < 4; j++ )
        out[4 * i + j] = res[j];
}
template <typename T>
__global__ __launch_bounds__( 64 ) void me_full_kernel( const MfReq<T> *reqs, const int16_t *mvc, const int *n_mvc, int n, int *out )
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if( i >= n )
        return;
    int res[4];
    mefull::mf_me_search_full<T>( &reqs[i], (const int16_t( * )[2])( mvc + (long)i * MF_MVC_MAX * 2 ), n_mvc[i], res );
    for( int k = 0; k < 4; k++ )
        out[4 * i + k] = res[k];
}
#endif
