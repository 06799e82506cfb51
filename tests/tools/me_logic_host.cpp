// TEST INFRASTRUCTURE: compiles x264_amd/csrc/me_logic.h (the block-search logic the device kernel runs per 8-lane group) for
// the host with a scalar evaluator, so that its candidate order, tie-breaking and early exits can be checked against the oracle's
// whole-field search (or{8,10}_search_field) without a GPU.  Pixel costs come from the oracle's own metrics (liboracle.so): only
// the logic is under test here; the device evaluator (loads, DPP reductions) is covered by the -m gpu parity tests.
// A second evaluator (StripEval) fetches the reference samples the way the device does -- out of the strip copy of the padded planes,
// through the index arithmetic of x264_amd/csrc/strip_layout.h that the kernels share -- so the layout (writer and reader side, the
// quarter-pel partner column inside one strip) is checked on the host too.
#include <stdint.h>
#include <string.h>
#include <vector>
#define ME_HD inline
#include "me_logic.h"
#include "strip_layout.h"
extern "C" {
#include "x264_oracle.h"
}

template <typename T> struct Ops;
template <> struct Ops<uint8_t>
{
    static int sad( const uint8_t *a, int sa, const uint8_t *b, int sb ) { return or8_sad( a, sa, b, sb, 8, 8 ); }
    static int satd( const uint8_t *a, int sa, const uint8_t *b, int sb ) { return or8_satd( a, sa, b, sb, 8, 8 ); }
    static void mc( uint8_t *d, const uint8_t *const pl[4], int stride, int x, int y, const or_weight *w ) { or8_mc_luma( d, 8, pl, stride, x, y, 8, 8, w ); }
};
template <> struct Ops<uint16_t>
{
    static int sad( const uint16_t *a, int sa, const uint16_t *b, int sb ) { return or10_sad( a, sa, b, sb, 8, 8 ); }
    static int satd( const uint16_t *a, int sa, const uint16_t *b, int sb ) { return or10_satd( a, sa, b, sb, 8, 8 ); }
    static void mc( uint16_t *d, const uint16_t *const pl[4], int stride, int x, int y, const or_weight *w ) { or10_mc_luma( d, 8, pl, stride, x, y, 8, 8, w ); }
};

template <typename T>
struct HostEval : melogic::ScalarSets<HostEval<T>>
{
    const or_la_cfg *c;
    const T *fenc;          // block origin in the source plane
    const T *ref[4];        // block origin in the four half-pel planes of the reference
    const T *ref_w;         // block origin in the plane full-pel candidates read
    const or_weight *wt;
    int mvpx, mvpy;
    int n_fpel = 0, n_qpel = 0;
    int cmp( int satd, const T *b, int sb ) const { return satd ? Ops<T>::satd( fenc, c->stride, b, sb ) : Ops<T>::sad( fenc, c->stride, b, sb ); }
    int fpel( int x, int y ) { n_fpel++; return cmp( c->fpelcmp_satd, ref_w + y * c->stride + x, c->stride ); }
    int qpel( int qx, int qy, int use_satd )
    {
        T buf[64];
        n_qpel++;
        Ops<T>::mc( buf, ref, c->stride, qx, qy, wt && wt->on ? wt : nullptr );
        return cmp( use_satd, buf, 8 );
    }
    int bits( int qx, int qy ) const { return c->cost_mv[qx - mvpx] + c->cost_mv[qy - mvpy]; }
    bool any( bool v ) const { return v; }
};

// the reference read out of strip copies: every sample access goes through strip_layout::read_off / qpel_taps, a row at a time,
// exactly as a lane of the device evaluator does (me_search.h: GroupEval::fpel / qpel)
#define LA_PAD_HOST 32
template <typename T>
struct StripEval : melogic::ScalarSets<StripEval<T>>
{
    const or_la_cfg *c;
    const T *fenc;          // block origin in the source plane (row-major, as on the device)
    const T *strips;        // strips of the four planes of the reference (plane p at p * 2 * plane_elems)
    const T *strips_w;      // strips of the plane full-pel candidates read (weighted copy of plane 0 or `strips`)
    const or_weight *wt;
    int plane_elems, strip_elems, cx0, row16_0; // padded column of the block, strip-row offset of its row 0
    int mvpx, mvpy;
    int n_fpel = 0, n_qpel = 0;
    int pixel_max;
    int cmp( int satd, const T *b ) const { return satd ? Ops<T>::satd( fenc, c->stride, b, 8 ) : Ops<T>::sad( fenc, c->stride, b, 8 ); }
    int block0( int satd ) // the block itself on the unweighted plane 0 (zero check)
    {
        T buf[64];
        for( int r = 0; r < 8; r++ )
            memcpy( buf + 8 * r, strips + strip_layout::read_off( cx0, row16_0 + 16 * r, strip_elems ), 8 * sizeof( T ) );
        return cmp( satd, buf );
    }
    int fpel( int x, int y )
    {
        T buf[64];
        n_fpel++;
        for( int r = 0; r < 8; r++ )
            memcpy( buf + 8 * r, strips_w + strip_layout::read_off( cx0 + x, row16_0 + 16 * r + ( y << 4 ), strip_elems ), 8 * sizeof( T ) );
        return cmp( c->fpelcmp_satd, buf );
    }
    int qpel( int qx, int qy, int use_satd )
    {
        T buf[64];
        n_qpel++;
        for( int r = 0; r < 8; r++ )
        {
            int oa, ob;
            strip_layout::qpel_taps( plane_elems, strip_layout::read_off( cx0 + ( qx >> 2 ), row16_0 + 16 * r + ( ( qy >> 2 ) << 4 ), strip_elems ), qx, qy, oa, ob );
            for( int i = 0; i < 8; i++ )
            {
                int v = ( strips[oa + i] + strips[ob + i] + 1 ) >> 1;
                if( wt && wt->on )
                {
                    const int off = wt->offset * ( sizeof( T ) == 1 ? 1 : 4 ); // the offset is in 8-bit units (mc.c:117-160)
                    v = wt->denom >= 1 ? ( ( v * wt->scale + ( 1 << ( wt->denom - 1 ) ) ) >> wt->denom ) + off : v * wt->scale + off;
                    v = v < 0 ? 0 : v > pixel_max ? pixel_max : v;
                }
                buf[8 * r + i] = (T)v;
            }
        }
        return cmp( use_satd, buf );
    }
    int bits( int qx, int qy ) const { return c->cost_mv[qx - mvpx] + c->cost_mv[qy - mvpy]; }
    bool any( bool v ) const { return v; }
};

// strip copy of `n_planes` padded planes (origin pointers as the oracle hands them), written the way strips_kernel /
// weight_strips_kernel write it
template <typename T>
static std::vector<T> build_strips( const T *const origin[], int n_planes, int stride, int rows )
{
    const int n_strips = stride >> 3;
    std::vector<T> st( (size_t)n_planes * 2 * stride * rows, 0 );
    for( int p = 0; p < n_planes; p++ )
    {
        const T *pad0 = origin[p] - ( LA_PAD_HOST * stride + LA_PAD_HOST );
        T *dst = st.data() + (size_t)p * 2 * stride * rows;
        for( int k = 0; k < n_strips; k++ )
            for( int Y = 0; Y < rows; Y++ )
                memcpy( dst + strip_layout::row_off( k, Y, rows ), pad0 + (size_t)Y * stride + 8 * k, ( k + 1 < n_strips ? 16 : 8 ) * sizeof( T ) );
    }
    return st;
}

template <typename T, bool STRIPS>
static void search_field( const or_la_cfg *c, const T *fenc0, const T *const ref[4], const T *ref_w, const or_weight *wt,
                          int16_t ( *mvs )[2], int *mv_costs, long *evals )
{
    const int rows = 8 * c->mb_h + 2 * LA_PAD_HOST, plane_elems = c->stride * rows;
    std::vector<T> st, stw;
    if( STRIPS )
    {
        st = build_strips<T>( ref, 4, c->stride, rows );
        if( wt && wt->on )
            stw = build_strips<T>( &ref_w, 1, c->stride, rows );
    }
    const int W = c->mb_w, H = c->mb_h, ns = c->n_slices > 1 ? c->n_slices : 1;
    const bool no_edges = !c->do_edges && W > 2 && H > 2;
    MeCfg C = { c->me_method == OR_ME_HEX, c->subpel_refine >= 3, c->me_range, c->mbcmp_satd, c->fpelcmp_satd };
    std::vector<int> packed( (size_t)W * H, 0 );
    for( int by = H - 1; by >= 0; by-- )
    {
        int band_end = H;
        for( int sl = ns - 1; sl >= 1; sl-- )
        {
            const int start = ( H * sl + ns / 2 ) / ns;
            if( by < start ) band_end = start;
        }
        for( int bx = W - 1; bx >= 0; bx-- )
        {
            const int xy = by * W + bx, off = 8 * ( by * c->stride + bx );
            if( no_edges && !( bx > 0 && bx < W - 1 && by > 0 && by < H - 1 ) )
                continue; // never visited: vectors and costs stay zero
            MeLim L;
            melogic::block_limits( L, bx, by, W, H, c->mv_range );
            const bool has_below = by < band_end - 1;
            int mvcx[4], mvcy[4];
            const int n = melogic::neighbour_list( bx, W, has_below, bx < W - 1 ? packed[xy + 1] : 0, has_below ? packed[xy + W] : 0,
                                                   has_below && bx > 0 ? packed[xy + W - 1] : 0, has_below && bx < W - 1 ? packed[xy + W + 1] : 0, mvcx, mvcy );
            int mvpx, mvpy;
            if( n <= 1 ) { mvpx = mvcx[0]; mvpy = mvcy[0]; }
            else { mvpx = melogic::median3( mvcx[0], mvcx[1], mvcx[2] ); mvpy = melogic::median3( mvcy[0], mvcy[1], mvcy[2] ); }
            int mvx = 0, mvy = 0, cost = 0;
            bool done = false;
            if( STRIPS )
            {
                StripEval<T> ev;
                ev.c = c; ev.fenc = fenc0 + off; ev.wt = wt; ev.mvpx = mvpx; ev.mvpy = mvpy;
                ev.strips = st.data(); ev.strips_w = wt && wt->on ? stw.data() : st.data();
                ev.plane_elems = plane_elems; ev.strip_elems = strip_layout::strip_elems( rows );
                ev.cx0 = 8 * bx + LA_PAD_HOST; ev.row16_0 = ( 8 * by + LA_PAD_HOST ) << 4;
                ev.pixel_max = ( 1 << ( 8 * (int)sizeof( T ) == 8 ? 8 : 10 ) ) - 1;
                if( !( mvpx | mvpy ) )
                {
                    cost = ev.block0( c->mbcmp_satd ); // slicetype.c:684-692
                    done = cost < 64;
                }
                if( !done )
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                if( evals ) { evals[0] += ev.n_fpel; evals[1] += ev.n_qpel; }
            }
            else
            {
                HostEval<T> ev;
                ev.c = c; ev.fenc = fenc0 + off; ev.wt = wt; ev.mvpx = mvpx; ev.mvpy = mvpy;
                for( int k = 0; k < 4; k++ ) ev.ref[k] = ref[k] + off;
                ev.ref_w = wt && wt->on ? ref_w + off : ev.ref[0];
                if( !( mvpx | mvpy ) )
                {
                    cost = ev.cmp( c->mbcmp_satd, ev.ref[0], c->stride ); // slicetype.c:684-692
                    done = cost < 64;
                }
                if( !done )
                    melogic::search( C, L, ev, mvpx, mvpy, n, mvcx, mvcy, mvx, mvy, cost );
                if( evals ) { evals[0] += ev.n_fpel; evals[1] += ev.n_qpel; }
            }
            if( !done )
            {
                cost -= c->cost_mv[0];
                if( mvx | mvy ) cost += 5 * c->lambda;
            }
            packed[xy] = ( mvx & 0xFFFF ) | ( mvy << 16 );
            mvs[xy][0] = (int16_t)mvx; mvs[xy][1] = (int16_t)mvy;
            mv_costs[xy] = cost;
        }
    }
}

extern "C" void mel8_search_field( const or_la_cfg *c, const uint8_t *fenc0, const uint8_t *const ref[4], const uint8_t *ref_w, const or_weight *wt,
                                   int16_t ( *mvs )[2], int *mv_costs, long *evals )
{
    search_field<uint8_t, false>( c, fenc0, ref, ref_w, wt, mvs, mv_costs, evals );
}
extern "C" void mel8_search_field_strips( const or_la_cfg *c, const uint8_t *fenc0, const uint8_t *const ref[4], const uint8_t *ref_w, const or_weight *wt,
                                          int16_t ( *mvs )[2], int *mv_costs, long *evals )
{
    search_field<uint8_t, true>( c, fenc0, ref, ref_w, wt, mvs, mv_costs, evals );
}
extern "C" void mel10_search_field( const or_la_cfg *c, const uint16_t *fenc0, const uint16_t *const ref[4], const uint16_t *ref_w, const or_weight *wt,
                                    int16_t ( *mvs )[2], int *mv_costs, long *evals )
{
    search_field<uint16_t, false>( c, fenc0, ref, ref_w, wt, mvs, mv_costs, evals );
}
extern "C" void mel10_search_field_strips( const or_la_cfg *c, const uint16_t *fenc0, const uint16_t *const ref[4], const uint16_t *ref_w, const or_weight *wt,
                                           int16_t ( *mvs )[2], int *mv_costs, long *evals )
{
    search_field<uint16_t, true>( c, fenc0, ref, ref_w, wt, mvs, mv_costs, evals );
}
