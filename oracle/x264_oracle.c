/* TEST INFRASTRUCTURE ONLY -- see x264_oracle.h.
 *
 * CPU restatement of the x264 lookahead / motion-estimation hot path.  Written from the behaviour of
 * the reference (file:line cited per function, paths relative to /root/reference); compiled twice
 * (-DOR_DEPTH=8 / 10).  Everything here is integer arithmetic except the AQ helper at the end.
 */
#include "x264_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifndef OR_DEPTH
#define OR_DEPTH 8
#endif
#if OR_DEPTH == 8
typedef uint8_t pixel;
typedef int16_t dctcoef;
typedef uint16_t udctcoef;
#define ORN(x) or8_##x
#else
typedef uint16_t pixel;
typedef int32_t dctcoef;
typedef uint32_t udctcoef;
#define ORN(x) or10_##x
#endif
#define PIXEL_MAX ((1 << OR_DEPTH) - 1)
#define DEPTH_SHIFT (OR_DEPTH - 8)
#define COST_MAX (1 << 28)

static inline int clip3( int v, int lo, int hi ) { return v < lo ? lo : v > hi ? hi : v; }
static inline pixel clip_pixel( int v ) { return (pixel)( v < 0 ? 0 : v > PIXEL_MAX ? PIXEL_MAX : v ); }
static inline int imin( int a, int b ) { return a < b ? a : b; }
static inline int imax( int a, int b ) { return a > b ? a : b; }
static inline int median3( int a, int b, int c ) /* common/base.h:232-240 */
{
    int lo = imin( a, b ), hi = imax( a, b );
    return imax( lo, imin( hi, c ) );
}

/* ------------------------------------------------------------------------------------------------
 * M1: half-resolution planes.  common/mc.c:458-507 (x264_frame_init_lowres + frame_init_lowres_core),
 * common/frame.c:535-554,627-631 (border expansion), frame.c:640-666 (mod16 replication).
 * ---------------------------------------------------------------------------------------------- */
static inline int rnd_avg( int a, int b ) { return ( a + b + 1 ) >> 1; }

void ORN(lowres_core)( const pixel *src, pixel *d0, pixel *dh, pixel *dv, pixel *dc,
                       int src_stride, int dst_stride, int w, int h )
{
    for( int y = 0; y < h; y++ )
    {
        const pixel *r0 = src + 2*y*src_stride, *r1 = r0 + src_stride, *r2 = r1 + src_stride;
        for( int x = 0; x < w; x++ )
        {
            int c00 = rnd_avg( r0[2*x],   r1[2*x]   ), c10 = rnd_avg( r0[2*x+1], r1[2*x+1] ), c20 = rnd_avg( r0[2*x+2], r1[2*x+2] );
            int c01 = rnd_avg( r1[2*x],   r2[2*x]   ), c11 = rnd_avg( r1[2*x+1], r2[2*x+1] ), c21 = rnd_avg( r1[2*x+2], r2[2*x+2] );
            d0[y*dst_stride+x] = (pixel)rnd_avg( c00, c10 );
            dh[y*dst_stride+x] = (pixel)rnd_avg( c10, c20 );
            dv[y*dst_stride+x] = (pixel)rnd_avg( c01, c11 );
            dc[y*dst_stride+x] = (pixel)rnd_avg( c11, c21 );
        }
    }
}

/* Half-pel planes of a full-resolution plane, common/mc.c:172-196 (hpel_filter): six-tap (1,-5,20,20,-5,1)
 * vertically (dstv, columns -2 .. width+2), horizontally (dsth), and both (dstc: the horizontal taps run over
 * the unrounded vertical sums, which the reference parks in an int16 row buffer, offset by -10*PIXEL_MAX for
 * the high bit depths so that they fit). */
static inline int tap6( int a, int b, int c, int d, int e, int f ) { return a + f - 5*( b + e ) + 20*( c + d ); }
void ORN(hpel_filter)( pixel *dsth, pixel *dstv, pixel *dstc, const pixel *src, long stride, int width, int height, int16_t *buf )
{
    const int pixel_max = ( 1 << OR_DEPTH ) - 1;
    const int pad = OR_DEPTH > 9 ? -10 * pixel_max : 0;
    for( int y = 0; y < height; y++ )
    {
        for( int x = -2; x < width + 3; x++ )
        {
            int v = tap6( src[x-2*stride], src[x-stride], src[x], src[x+stride], src[x+2*stride], src[x+3*stride] );
            dstv[x] = clip_pixel( ( v + 16 ) >> 5 );
            buf[x+2] = (int16_t)( v + pad );
        }
        for( int x = 0; x < width; x++ )
            dstc[x] = clip_pixel( ( tap6( buf[x], buf[x+1], buf[x+2], buf[x+3], buf[x+4], buf[x+5] ) - 32*pad + 512 ) >> 10 );
        for( int x = 0; x < width; x++ )
            dsth[x] = clip_pixel( ( tap6( src[x-2], src[x-1], src[x], src[x+1], src[x+2], src[x+3] ) + 16 ) >> 5 );
        dsth += stride; dstv += stride; dstc += stride; src += stride;
    }
}

/* Integral image rows for the exhaustive searches, common/mc.c:424-456 (integral_init4h/8h/4v/8v): running column
 * sums of 4- or 8-wide row sums, then differences 4 or 8 rows apart give the box sums; all arithmetic modulo 2^16. */
void ORN(integral_init4h)( uint16_t *sum, const pixel *pix, long stride )
{
    int v = pix[0] + pix[1] + pix[2] + pix[3];
    for( long x = 0; x < stride - 4; x++ )
    {
        sum[x] = (uint16_t)( v + sum[x - stride] );
        v += pix[x + 4] - pix[x];
    }
}
void ORN(integral_init8h)( uint16_t *sum, const pixel *pix, long stride )
{
    int v = 0;
    for( int i = 0; i < 8; i++ ) v += pix[i];
    for( long x = 0; x < stride - 8; x++ )
    {
        sum[x] = (uint16_t)( v + sum[x - stride] );
        v += pix[x + 8] - pix[x];
    }
}
void ORN(integral_init4v)( uint16_t *sum8, uint16_t *sum4, long stride )
{
    for( long x = 0; x < stride - 8; x++ )
        sum4[x] = (uint16_t)( sum8[x + 4*stride] - sum8[x] );
    for( long x = 0; x < stride - 8; x++ )
        sum8[x] = (uint16_t)( sum8[x + 8*stride] + sum8[x + 8*stride + 4] - sum8[x] - sum8[x + 4] );
}
void ORN(integral_init8v)( uint16_t *sum8, long stride )
{
    for( long x = 0; x < stride - 8; x++ )
        sum8[x] = (uint16_t)( sum8[x + 8*stride] - sum8[x] );
}

/* Successive elimination, common/pixel.c:759-803 (ads4/ads2/ads1): candidates i of a row whose lower bound
 * sum|enc_dc - box sum| + cost_mvx[i] stays below thresh, in order. n_dc = 4, 2 or 1. */
int ORN(ads)( int n_dc, const int *enc_dc, const uint16_t *sums, int delta, const uint16_t *cost_mvx, int16_t *mvs, int width, int thresh )
{
    int nmv = 0;
    for( int i = 0; i < width; i++, sums++ )
    {
        int ads = abs( enc_dc[0] - sums[0] ) + cost_mvx[i];
        if( n_dc == 2 )
            ads += abs( enc_dc[1] - sums[delta] );
        else if( n_dc == 4 )
            ads += abs( enc_dc[1] - sums[8] ) + abs( enc_dc[2] - sums[delta] ) + abs( enc_dc[3] - sums[delta + 8] );
        if( ads < thresh )
            mvs[nmv++] = (int16_t)i;
    }
    return nmv;
}

/* src is the picture as handed to the encoder (width x height, not necessarily mod 16).  The
 * reference first replicates the last column/row out to the mod16 size and then one more
 * column/row (mc.c:466-468); both are the same as clamping the source coordinate. */
void ORN(lowres_init)( const pixel *src, int src_stride, int width, int height, int mb_w, int mb_h,
                       pixel *p0, pixel *ph, pixel *pv, pixel *pc, int stride )
{
    const int lw = 8*mb_w, lh = 8*mb_h;
    pixel *planes[4] = { p0, ph, pv, pc };
    for( int y = 0; y < lh; y++ )
    {
        const pixel *r[3];
        for( int k = 0; k < 3; k++ )
            r[k] = src + (size_t)imin( 2*y+k, height-1 ) * src_stride;
        for( int x = 0; x < lw; x++ )
        {
            int x0 = imin( 2*x, width-1 ), x1 = imin( 2*x+1, width-1 ), x2 = imin( 2*x+2, width-1 );
            int c00 = rnd_avg( r[0][x0], r[1][x0] ), c10 = rnd_avg( r[0][x1], r[1][x1] ), c20 = rnd_avg( r[0][x2], r[1][x2] );
            int c01 = rnd_avg( r[1][x0], r[2][x0] ), c11 = rnd_avg( r[1][x1], r[2][x1] ), c21 = rnd_avg( r[1][x2], r[2][x2] );
            p0[y*stride+x] = (pixel)rnd_avg( c00, c10 );
            ph[y*stride+x] = (pixel)rnd_avg( c10, c20 );
            pv[y*stride+x] = (pixel)rnd_avg( c01, c11 );
            pc[y*stride+x] = (pixel)rnd_avg( c11, c21 );
        }
    }
    for( int p = 0; p < 4; p++ )
    {
        pixel *pl = planes[p];
        for( int y = 0; y < lh; y++ )
            for( int k = 1; k <= OR_PAD; k++ )
            {
                pl[y*stride - k] = pl[y*stride];
                pl[y*stride + lw - 1 + k] = pl[y*stride + lw - 1];
            }
        for( int k = 1; k <= OR_PAD; k++ )
        {
            memcpy( pl - k*stride - OR_PAD, pl - OR_PAD, (lw + 2*OR_PAD) * sizeof(pixel) );
            memcpy( pl + (lh-1+k)*stride - OR_PAD, pl + (lh-1)*stride - OR_PAD, (lw + 2*OR_PAD) * sizeof(pixel) );
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * P1/P3/P4/P8: block metrics.  common/pixel.c:55-80 (sad), :85-151 (ssd), :265-332 (satd),
 * :334-381 (sa8d), :183-201 (var).  The reference packs two 16/32-bit lanes per word; the results
 * are the plain sums restated here (no lane overflows for the supported depths).
 * ---------------------------------------------------------------------------------------------- */
int ORN(sad)( const pixel *a, int sa, const pixel *b, int sb, int w, int h )
{
    int s = 0;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
            s += abs( a[y*sa+x] - b[y*sb+x] );
    return s;
}

int ORN(ssd)( const pixel *a, int sa, const pixel *b, int sb, int w, int h )
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

static inline void hadamard4( int *v, int step )
{
    int s01 = v[0] + v[step], d01 = v[0] - v[step], s23 = v[2*step] + v[3*step], d23 = v[2*step] - v[3*step];
    v[0] = s01 + s23; v[step] = d01 + d23; v[2*step] = s01 - s23; v[3*step] = d01 - d23;
}

static int hadamard_abs_4x4( const pixel *a, int sa, const pixel *b, int sb )
{
    int d[16], s = 0;
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            d[4*y+x] = a[y*sa+x] - b[y*sb+x];
    for( int y = 0; y < 4; y++ ) hadamard4( d + 4*y, 1 );
    for( int x = 0; x < 4; x++ ) hadamard4( d + x, 4 );
    for( int i = 0; i < 16; i++ ) s += abs( d[i] );
    return s;
}

/* satd of any WxH made of 4x4 tiles: sum over tiles of (sum |H4 D H4^T|), halved per 8x4 / 4x4 unit
 * exactly as PIXEL_SATD_C composes x264_pixel_satd_8x4 / _4x4 (pixel.c:265-332). */
int ORN(satd)( const pixel *a, int sa, const pixel *b, int sb, int w, int h )
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

static int hadamard_abs_8x8( const pixel *a, int sa, const pixel *b, int sb )
{
    int d[64], s = 0;
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
            d[8*y+x] = a[y*sa+x] - b[y*sb+x];
    for( int pass = 0; pass < 2; pass++ )
    {
        int step = pass ? 8 : 1, line = pass ? 1 : 8;
        for( int l = 0; l < 8; l++ )
        {
            int *v = d + l*line;
            for( int span = 1; span < 8; span <<= 1 )
                for( int i = 0; i < 8; i++ )
                    if( !(i & span) )
                    {
                        int p = v[i*step], q = v[(i+span)*step];
                        v[i*step] = p + q; v[(i+span)*step] = p - q;
                    }
        }
    }
    for( int i = 0; i < 64; i++ ) s += abs( d[i] );
    return s;
}

int ORN(sa8d)( const pixel *a, int sa, const pixel *b, int sb, int w ) /* w = 8 or 16 (square) */
{
    int s = 0;
    for( int y = 0; y < w; y += 8 )
        for( int x = 0; x < w; x += 8 )
            s += hadamard_abs_8x8( a + y*sa + x, sa, b + y*sb + x, sb );
    return ( s + 2 ) >> 2;
}

uint64_t ORN(var)( const pixel *a, int sa, int w, int h )
{
    uint32_t sum = 0, sqr = 0;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
        {
            sum += a[y*sa+x];
            sqr += a[y*sa+x] * a[y*sa+x];
        }
    return sum + ( (uint64_t)sqr << 32 );
}

/* The remaining P8 metrics (main-encode consumers): common/pixel.c:206-231 (var2_8x8 / var2_8x16 on the interleaved chroma
 * halves of the fenc/fdec buffers), :383-435 (hadamard_ac: sums of the absolute AC coefficients of the 4x4 and 8x8
 * Hadamard transforms of the pixels themselves), :716-723 (vsad) and :747-754 (asd8). */
int ORN(var2)( const pixel *fenc, const pixel *fdec, int h, int ssd[2] )
{
    int sum_u = 0, sum_v = 0, sqr_u = 0, sqr_v = 0;
    const int shift = h == 16 ? 7 : 6;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < 8; x++ )
        {
            int du = fenc[y*OR_FENC_STRIDE + x] - fdec[y*OR_FDEC_STRIDE + x];
            int dv = fenc[y*OR_FENC_STRIDE + x + OR_FENC_STRIDE/2] - fdec[y*OR_FDEC_STRIDE + x + OR_FDEC_STRIDE/2];
            sum_u += du; sum_v += dv; sqr_u += du*du; sqr_v += dv*dv;
        }
    ssd[0] = sqr_u; ssd[1] = sqr_v;
    return sqr_u - (int)( (int64_t)sum_u * sum_u >> shift ) + sqr_v - (int)( (int64_t)sum_v * sum_v >> shift );
}
static void hadamard_ac_8x8( const pixel *pix, int stride, uint64_t *sum4, uint64_t *sum8 )
{
    static const pixel zero[8] = { 0 };
    int dc = 0, s4 = 0;
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
            dc += pix[y*stride + x];
    for( int by = 0; by < 8; by += 4 )
        for( int bx = 0; bx < 8; bx += 4 )
            s4 += hadamard_abs_4x4( pix + by*stride + bx, stride, zero, 0 );
    *sum4 += (uint64_t)( s4 - dc );
    *sum8 += (uint64_t)( hadamard_abs_8x8( pix, stride, zero, 0 ) - dc );
}
uint64_t ORN(hadamard_ac)( const pixel *pix, int stride, int w, int h )
{
    uint64_t sum4 = 0, sum8 = 0;
    for( int y = 0; y < h; y += 8 )
        for( int x = 0; x < w; x += 8 )
            hadamard_ac_8x8( pix + y*stride + x, stride, &sum4, &sum8 );
    return ( ( sum8 >> 2 ) << 32 ) + ( (uint32_t)sum4 >> 1 );
}
int ORN(vsad)( const pixel *src, long stride, int height )
{
    int score = 0;
    for( int i = 1; i < height; i++, src += stride )
        for( int j = 0; j < 16; j++ )
            score += abs( src[j] - src[j + stride] );
    return score;
}
int ORN(asd8)( const pixel *a, long sa, const pixel *b, long sb, int height )
{
    int sum = 0;
    for( int y = 0; y < height; y++ )
        for( int x = 0; x < 8; x++ )
            sum += a[y*sa + x] - b[y*sb + x];
    return abs( sum );
}

static inline int mbcmp8x8( const or_la_cfg *c, const pixel *a, int sa, const pixel *b, int sb )
{
    return c->mbcmp_satd ? ORN(satd)( a, sa, b, sb, 8, 8 ) : ORN(sad)( a, sa, b, sb, 8, 8 );
}
static inline int fpelcmp8x8( const or_la_cfg *c, const pixel *a, int sa, const pixel *b, int sb )
{
    return c->fpelcmp_satd ? ORN(satd)( a, sa, b, sb, 8, 8 ) : ORN(sad)( a, sa, b, sb, 8, 8 );
}

/* ------------------------------------------------------------------------------------------------
 * P6/P7: intra predictors used by the lowres intra cost.  common/predict.c:221-308 (8x8 chroma
 * dc/h/v/plane), :632-675 (edge low-pass, all neighbours available), :741-884 (modes 3..8).
 * src points at pixel (0,0) of an OR_FDEC_STRIDE buffer whose row -1 (x = -1..15) and column -1
 * hold the neighbours.  Written per output pixel, following H.264 8.3.2.2 / 8.3.4.
 * ---------------------------------------------------------------------------------------------- */
#define S(x,y) src[(y)*OR_FDEC_STRIDE+(x)]
enum { PRED8C_DC = 0, PRED8C_H = 1, PRED8C_V = 2, PRED8C_P = 3 };

void ORN(predict_8x8c)( int mode, pixel *src )
{
    if( mode == PRED8C_DC )
    {
        int t0 = 0, t1 = 0, l0 = 0, l1 = 0;
        for( int i = 0; i < 4; i++ )
        {
            t0 += S(i,-1); t1 += S(i+4,-1); l0 += S(-1,i); l1 += S(-1,i+4);
        }
        int dc[2][2] = { { (t0+l0+4)>>3, (t1+2)>>2 }, { (l1+2)>>2, (t1+l1+4)>>3 } };
        for( int y = 0; y < 8; y++ )
            for( int x = 0; x < 8; x++ )
                S(x,y) = (pixel)dc[y>>2][x>>2];
    }
    else if( mode == PRED8C_H )
    {
        for( int y = 0; y < 8; y++ )
            for( int x = 0; x < 8; x++ )
                S(x,y) = S(-1,y);
    }
    else if( mode == PRED8C_V )
    {
        for( int y = 0; y < 8; y++ )
            for( int x = 0; x < 8; x++ )
                S(x,y) = S(x,-1);
    }
    else
    {
        int H = 0, V = 0;
        for( int i = 1; i <= 4; i++ )
        {
            H += i * ( S(3+i,-1) - S(3-i,-1) );
            V += i * ( S(-1,3+i) - S(-1,3-i) );
        }
        int a = 16 * ( S(-1,7) + S(7,-1) ), b = ( 17*H + 16 ) >> 5, cc = ( 17*V + 16 ) >> 5;
        for( int y = 0; y < 8; y++ )
            for( int x = 0; x < 8; x++ )
                S(x,y) = clip_pixel( ( a + b*(x-3) + cc*(y-3) + 16 ) >> 5 );
    }
}

/* edge layout of the reference: edge[14-y] = left y (y=0..7), edge[15] = top-left, edge[16+x] = top x
 * (x=0..15), edge[32] = edge[31]; edge[6] = edge[7]. */
void ORN(predict_8x8_filter)( const pixel *src, pixel *edge )
{
#define F3(a,b,c) ( ( (a) + 2*(b) + (c) + 2 ) >> 2 )
    edge[15] = (pixel)F3( S(0,-1), S(-1,-1), S(-1,0) );
    for( int y = 0; y < 7; y++ )
        edge[14-y] = (pixel)F3( S(-1,y-1), S(-1,y), S(-1,y+1) );
    edge[7] = edge[6] = (pixel)( ( S(-1,6) + 3*S(-1,7) + 2 ) >> 2 );
    for( int x = 0; x < 15; x++ )
        edge[16+x] = (pixel)F3( S(x-1,-1), S(x,-1), S(x+1,-1) );
    edge[31] = edge[32] = (pixel)( ( S(14,-1) + 3*S(15,-1) + 2 ) >> 2 );
}

void ORN(predict_8x8)( int mode, pixel *src, const pixel *edge )
{
    /* T(i): filtered top sample i (i = -1 is the corner), L(i): filtered left sample i */
#define T(i) ( (i) < 0 ? edge[15] : edge[16+(i)] )
#define L(i) ( (i) < 0 ? edge[15] : edge[14-(i)] )
#define F2(a,b) ( ( (a) + (b) + 1 ) >> 1 )
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
        {
            int v;
            switch( mode )
            {
                case 3: /* diagonal down-left */
                    v = ( x == 7 && y == 7 ) ? ( T(14) + 3*T(15) + 2 ) >> 2 : F3( T(x+y), T(x+y+1), T(x+y+2) );
                    break;
                case 4: /* diagonal down-right */
                    if( x > y )       v = F3( T(x-y-2), T(x-y-1), T(x-y) );
                    else if( x < y )  v = F3( L(y-x-2), L(y-x-1), L(y-x) );
                    else              v = F3( T(0), T(-1), L(0) );
                    break;
                case 5: /* vertical right */
                {
                    int z = 2*x - y;
                    if( z >= 0 && !(z & 1) )  v = F2( T(x-(y>>1)-1), T(x-(y>>1)) );
                    else if( z >= 0 )         v = F3( T(x-(y>>1)-2), T(x-(y>>1)-1), T(x-(y>>1)) );
                    else if( z == -1 )        v = F3( L(0), T(-1), T(0) );
                    else                      v = F3( L(y-2*x-1), L(y-2*x-2), L(y-2*x-3) );
                    break;
                }
                case 6: /* horizontal down */
                {
                    int z = 2*y - x;
                    if( z >= 0 && !(z & 1) )  v = F2( L(y-(x>>1)-1), L(y-(x>>1)) );
                    else if( z >= 0 )         v = F3( L(y-(x>>1)-2), L(y-(x>>1)-1), L(y-(x>>1)) );
                    else if( z == -1 )        v = F3( L(0), T(-1), T(0) );
                    else                      v = F3( T(x-2*y-1), T(x-2*y-2), T(x-2*y-3) );
                    break;
                }
                case 7: /* vertical left */
                    v = ( y & 1 ) ? F3( T(x+(y>>1)), T(x+(y>>1)+1), T(x+(y>>1)+2) ) : F2( T(x+(y>>1)), T(x+(y>>1)+1) );
                    break;
                default: /* 8: horizontal up */
                {
                    int z = x + 2*y;
                    if( z > 13 )          v = L(7);
                    else if( z == 13 )    v = ( L(6) + 3*L(7) + 2 ) >> 2;
                    else if( z & 1 )      v = F3( L(y+(x>>1)), L(y+(x>>1)+1), L(y+(x>>1)+2) );
                    else                  v = F2( L(y+(x>>1)), L(y+(x>>1)+1) );
                    break;
                }
            }
            S(x,y) = (pixel)v;
        }
#undef T
#undef L
}

/* res order dc, h, v (pixel.c:542-556 INTRA_MBCMP 8x8 chroma) */
void ORN(intra_x3_8x8c)( int satd, const pixel *fenc, pixel *fdec, int res[3] )
{
    for( int m = 0; m < 3; m++ )
    {
        ORN(predict_8x8c)( m, fdec );
        res[m] = satd ? ORN(satd)( fdec, OR_FDEC_STRIDE, fenc, OR_FENC_STRIDE, 8, 8 )
                      : ORN(sad)( fdec, OR_FDEC_STRIDE, fenc, OR_FENC_STRIDE, 8, 8 );
    }
}
#undef S

/* ------------------------------------------------------------------------------------------------
 * M2-M4: quarter-pel fetch from the four half-pel planes, averaging, explicit weights.
 * common/mc.c:198-249 (mc_luma/get_ref), :49-111 (avg), :117-160 (mc_weight),
 * common/tables.c:183-184 (which two planes a quarter-pel phase averages).
 * ---------------------------------------------------------------------------------------------- */
static inline int apply_weight( int v, const or_weight *wt )
{
    int off = wt->offset * ( 1 << DEPTH_SHIFT );
    if( wt->denom >= 1 )
        return clip_pixel( ( ( v * wt->scale + ( 1 << ( wt->denom - 1 ) ) ) >> wt->denom ) + off );
    return clip_pixel( v * wt->scale + off );
}

/* One interpolated sample at lowres integer position (x,y) displaced by quarter-pel (mvx,mvy).
 * Phase (fx,fy): first tap from plane (fx?H:0)+(fy==2?V:0) one row lower when fy==3; second tap
 * from plane (fx==2?H:0)+(fy?V:0) one column further when fx==3; taps equal when fx,fy are even. */
static inline int qpel_sample( const pixel *const planes[4], int stride, int x, int y, int mvx, int mvy, const or_weight *wt )
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
    return wt && wt->on ? apply_weight( v, wt ) : v;
}

void ORN(mc_luma)( pixel *dst, int ds, const pixel *const planes[4], int stride, int mvx, int mvy, int w, int h, const or_weight *wt )
{
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
            dst[y*ds+x] = (pixel)qpel_sample( planes, stride, x, y, mvx, mvy, wt );
}

void ORN(avg)( pixel *dst, int ds, const pixel *a, int sa, const pixel *b, int sb, int w, int h, int weight )
{
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < w; x++ )
            dst[y*ds+x] = weight == 32 ? (pixel)rnd_avg( a[y*sa+x], b[y*sb+x] )
                                       : clip_pixel( ( a[y*sa+x]*weight + b[y*sb+x]*(64-weight) + 32 ) >> 6 );
}

/* whole padded plane 0 weighted: slicetype.c:490-500 + frame.c:825-841 */
void ORN(weight_plane)( pixel *dst, const pixel *src, int stride, int width, int lines, const or_weight *wt )
{
    for( int y = -OR_PAD; y < lines + OR_PAD; y++ )
        for( int x = -OR_PAD; x < width + OR_PAD; x++ )
            dst[y*stride+x] = (pixel)apply_weight( src[y*stride+x], wt );
}

/* weight_cost_luma without the slice-header term: slicetype.c:191-222 */
unsigned ORN(weight_cost)( const or_la_cfg *c, const pixel *fenc0, const pixel *ref0, const or_weight *wt, const uint16_t *intra_cost )
{
    unsigned cost = 0;
    pixel buf[64];
    for( int by = 0; by < c->mb_h; by++ )
        for( int bx = 0; bx < c->mb_w; bx++ )
        {
            const pixel *r = ref0 + 8*( by*c->stride + bx ), *f = fenc0 + 8*( by*c->stride + bx );
            int cmp;
            if( wt && wt->on )
            {
                for( int i = 0; i < 64; i++ )
                    buf[i] = (pixel)apply_weight( r[(i>>3)*c->stride + (i&7)], wt );
                cmp = mbcmp8x8( c, buf, 8, f, c->stride );
            }
            else
                cmp = mbcmp8x8( c, r, c->stride, f, c->stride );
            cost += imin( cmp, intra_cost[by*c->mb_w+bx] );
        }
    return cost;
}

/* ------------------------------------------------------------------------------------------------
 * D1/Q1: forward transforms and quantisers as vtable primitives.  common/dct.c:47-270,332-386,
 * common/quant.c:50-104.  fenc stride 16, fdec stride 32.  Coefficient (u,v) (u horizontal
 * frequency) of a 4x4/8x8 block lands at out[u*N+v], the reference's transposed order.
 * ---------------------------------------------------------------------------------------------- */
static void fdct4_1d( const int *in, int step, int *out, int ostep )
{
    int s03 = in[0] + in[3*step], s12 = in[step] + in[2*step], d03 = in[0] - in[3*step], d12 = in[step] - in[2*step];
    out[0] = s03 + s12; out[ostep] = 2*d03 + d12; out[2*ostep] = s03 - s12; out[3*ostep] = d03 - 2*d12;
}

static void sub_dct4( dctcoef *out, const pixel *fenc, const pixel *fdec )
{
    int d[16], t[16], o[16];
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            d[4*y+x] = fenc[y*OR_FENC_STRIDE+x] - fdec[y*OR_FDEC_STRIDE+x];
    for( int y = 0; y < 4; y++ ) fdct4_1d( d + 4*y, 1, t + y, 4 );   /* t[u*4+y] */
    for( int u = 0; u < 4; u++ ) fdct4_1d( t + 4*u, 1, o + 4*u, 1 ); /* o[u*4+v] */
    for( int i = 0; i < 16; i++ ) out[i] = (dctcoef)o[i];
}

static void fdct8_1d( const int *in, int step, int *out, int ostep )
{
    int s07 = in[0] + in[7*step], s16 = in[step] + in[6*step], s25 = in[2*step] + in[5*step], s34 = in[3*step] + in[4*step];
    int d07 = in[0] - in[7*step], d16 = in[step] - in[6*step], d25 = in[2*step] - in[5*step], d34 = in[3*step] - in[4*step];
    int e0 = s07 + s34, e1 = s16 + s25, e2 = s07 - s34, e3 = s16 - s25;
    int o4 = d16 + d25 + ( d07 + ( d07 >> 1 ) );
    int o5 = d07 - d34 - ( d25 + ( d25 >> 1 ) );
    int o6 = d07 + d34 - ( d16 + ( d16 >> 1 ) );
    int o7 = d16 - d25 + ( d34 + ( d34 >> 1 ) );
    out[0] = e0 + e1;           out[ostep]   = o4 + ( o7 >> 2 );
    out[2*ostep] = e2 + ( e3 >> 1 ); out[3*ostep] = o5 + ( o6 >> 2 );
    out[4*ostep] = e0 - e1;     out[5*ostep] = o6 - ( o5 >> 2 );
    out[6*ostep] = ( e2 >> 1 ) - e3; out[7*ostep] = ( o4 >> 2 ) - o7;
}

static void sub_dct8( dctcoef *out, const pixel *fenc, const pixel *fdec )
{
    int d[64], t[64], o[64];
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
            d[8*y+x] = fenc[y*OR_FENC_STRIDE+x] - fdec[y*OR_FDEC_STRIDE+x];
    /* the reference's first pass runs down the columns (dct.c:364-369), then along the rows */
    for( int x = 0; x < 8; x++ ) fdct8_1d( d + x, 8, t + x, 8 );        /* t[v*8+x] */
    for( int v = 0; v < 8; v++ ) fdct8_1d( t + 8*v, 1, o + v, 8 );      /* o[u*8+v] */
    for( int i = 0; i < 64; i++ ) out[i] = (dctcoef)o[i];
}

static int dc_sum4( const pixel *fenc, const pixel *fdec )
{
    int s = 0;
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            s += fenc[y*OR_FENC_STRIDE+x] - fdec[y*OR_FDEC_STRIDE+x];
    return s;
}

void ORN(dct)( int kind, dctcoef *out, const pixel *fenc, const pixel *fdec )
{
    switch( kind )
    {
        case 0: sub_dct4( out, fenc, fdec ); break;
        case 1:
            for( int i = 0; i < 4; i++ )
                sub_dct4( out + 16*i, fenc + 4*(i&1) + 4*(i>>1)*OR_FENC_STRIDE, fdec + 4*(i&1) + 4*(i>>1)*OR_FDEC_STRIDE );
            break;
        case 2:
            for( int j = 0; j < 4; j++ )
                for( int i = 0; i < 4; i++ )
                    sub_dct4( out + 64*j + 16*i,
                              fenc + 8*(j&1) + 8*(j>>1)*OR_FENC_STRIDE + 4*(i&1) + 4*(i>>1)*OR_FENC_STRIDE,
                              fdec + 8*(j&1) + 8*(j>>1)*OR_FDEC_STRIDE + 4*(i&1) + 4*(i>>1)*OR_FDEC_STRIDE );
            break;
        case 3: sub_dct8( out, fenc, fdec ); break;
        case 4:
            for( int j = 0; j < 4; j++ )
                sub_dct8( out + 64*j, fenc + 8*(j&1) + 8*(j>>1)*OR_FENC_STRIDE, fdec + 8*(j&1) + 8*(j>>1)*OR_FDEC_STRIDE );
            break;
        case 5: /* sub8x8_dct_dc: 2x2 Hadamard of the four 4x4 DC sums */
        {
            int s[4];
            for( int i = 0; i < 4; i++ )
                s[i] = dc_sum4( fenc + 4*(i&1) + 4*(i>>1)*OR_FENC_STRIDE, fdec + 4*(i&1) + 4*(i>>1)*OR_FDEC_STRIDE );
            out[0] = (dctcoef)( s[0]+s[1]+s[2]+s[3] ); out[1] = (dctcoef)( s[0]+s[1]-s[2]-s[3] );
            out[2] = (dctcoef)( s[0]-s[1]+s[2]-s[3] ); out[3] = (dctcoef)( s[0]-s[1]-s[2]+s[3] );
            break;
        }
        case 6: /* sub8x16_dct_dc: 2x4 transform of eight 4x4 DC sums (dct.c:231-270) */
        {
            int a[8], h0[4], h1[4];
            for( int i = 0; i < 8; i++ )
                a[i] = dc_sum4( fenc + 4*(i&1) + 4*(i>>1)*OR_FENC_STRIDE, fdec + 4*(i&1) + 4*(i>>1)*OR_FDEC_STRIDE );
            for( int r = 0; r < 4; r++ ) { h0[r] = a[2*r] + a[2*r+1]; h1[r] = a[2*r] - a[2*r+1]; }
            int p0 = h0[0]+h0[1], p1 = h0[2]+h0[3], p2 = h1[0]+h1[1], p3 = h1[2]+h1[3];
            int q0 = h0[0]-h0[1], q1 = h0[2]-h0[3], q2 = h1[0]-h1[1], q3 = h1[2]-h1[3];
            out[0] = (dctcoef)( p0+p1 ); out[1] = (dctcoef)( p2+p3 ); out[2] = (dctcoef)( p0-p1 ); out[3] = (dctcoef)( p2-p3 );
            out[4] = (dctcoef)( q0-q1 ); out[5] = (dctcoef)( q2-q3 ); out[6] = (dctcoef)( q0+q1 ); out[7] = (dctcoef)( q2+q3 );
            break;
        }
        case 7: /* dct4x4dc in place on out (dct.c:47-77): 4x4 Hadamard, (x+1)>>1 */
        {
            int d[16];
            for( int i = 0; i < 16; i++ ) d[i] = out[i];
            int t[16];
            for( int i = 0; i < 4; i++ )
            {
                int s01 = d[4*i] + d[4*i+1], d01 = d[4*i] - d[4*i+1], s23 = d[4*i+2] + d[4*i+3], d23 = d[4*i+2] - d[4*i+3];
                t[i] = s01 + s23; t[4+i] = s01 - s23; t[8+i] = d01 - d23; t[12+i] = d01 + d23;
            }
            for( int i = 0; i < 4; i++ )
            {
                int s01 = t[4*i] + t[4*i+1], d01 = t[4*i] - t[4*i+1], s23 = t[4*i+2] + t[4*i+3], d23 = t[4*i+2] - t[4*i+3];
                out[4*i]   = (dctcoef)( ( s01 + s23 + 1 ) >> 1 ); out[4*i+1] = (dctcoef)( ( s01 - s23 + 1 ) >> 1 );
                out[4*i+2] = (dctcoef)( ( d01 - d23 + 1 ) >> 1 ); out[4*i+3] = (dctcoef)( ( d01 + d23 + 1 ) >> 1 );
            }
            break;
        }
        case 8: /* dct2x4dc in place on the eight DC values in out[0..7] (dct.c:109-143) */
        {
            int a0 = out[0] + out[1], a1 = out[2] + out[3], a2 = out[4] + out[5], a3 = out[6] + out[7];
            int a4 = out[0] - out[1], a5 = out[2] - out[3], a6 = out[4] - out[5], a7 = out[6] - out[7];
            int b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7, b4 = a0 - a1, b5 = a2 - a3, b6 = a4 - a5, b7 = a6 - a7;
            out[0] = (dctcoef)( b0 + b1 ); out[1] = (dctcoef)( b2 + b3 ); out[2] = (dctcoef)( b0 - b1 ); out[3] = (dctcoef)( b2 - b3 );
            out[4] = (dctcoef)( b4 - b5 ); out[5] = (dctcoef)( b6 - b7 ); out[6] = (dctcoef)( b4 + b5 ); out[7] = (dctcoef)( b6 + b7 );
            break;
        }
    }
}

static inline int quant_one( dctcoef *coef, uint32_t mf, uint32_t bias )
{
    int v = *coef;
    if( v > 0 )
        v = (int)( ( bias + (uint32_t)v ) * mf >> 16 );
    else
        v = -(int32_t)( ( bias + (uint32_t)( -v ) ) * mf >> 16 );
    *coef = (dctcoef)v;
    return *coef;
}

/* kind: 0 quant_4x4, 1 quant_8x8, 2 quant_4x4x4, 3 quant_4x4_dc, 4 quant_2x2_dc */
int ORN(quant)( int kind, dctcoef *coef, const udctcoef *mf, const udctcoef *bias, int mf_dc, int bias_dc )
{
    int nz = 0;
    switch( kind )
    {
        case 0: for( int i = 0; i < 16; i++ ) nz |= quant_one( coef+i, mf[i], bias[i] ); return !!nz;
        case 1: for( int i = 0; i < 64; i++ ) nz |= quant_one( coef+i, mf[i], bias[i] ); return !!nz;
        case 2:
        {
            int mask = 0;
            for( int j = 0; j < 4; j++ )
            {
                nz = 0;
                for( int i = 0; i < 16; i++ ) nz |= quant_one( coef+16*j+i, mf[i], bias[i] );
                mask |= ( !!nz ) << j;
            }
            return mask;
        }
        case 3: for( int i = 0; i < 16; i++ ) nz |= quant_one( coef+i, (uint32_t)mf_dc, (uint32_t)bias_dc ); return !!nz;
        case 4: for( int i = 0; i < 4; i++ )  nz |= quant_one( coef+i, (uint32_t)mf_dc, (uint32_t)bias_dc ); return !!nz;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------------
 * S2 (intra part): lowres intra cost of one 8x8 block.  encoder/slicetype.c:714-757.
 * ---------------------------------------------------------------------------------------------- */
static int intra_cost_block( const or_la_cfg *c, const pixel *fenc0, int bx, int by )
{
    pixel fenc[8*OR_FENC_STRIDE];
    pixel nb[9*OR_FDEC_STRIDE + 8];
    pixel edge[36];
    pixel *pix = nb + 8 + OR_FDEC_STRIDE;
    const pixel *src = fenc0 + 8*( by*c->stride + bx );
    for( int y = 0; y < 8; y++ )
        memcpy( fenc + y*OR_FENC_STRIDE, src + y*c->stride, 8*sizeof(pixel) );
    for( int x = -1; x < 16; x++ )
        pix[-OR_FDEC_STRIDE + x] = src[-c->stride + x];
    for( int y = 0; y < 8; y++ )
        pix[y*OR_FDEC_STRIDE - 1] = src[y*c->stride - 1];

    int res[3];
    /* predict_8x8_filter reads the unfiltered neighbours: take it before predictors overwrite pix */
    if( c->subme > 1 )
        ORN(predict_8x8_filter)( pix, edge );
    ORN(intra_x3_8x8c)( c->mbcmp_satd, fenc, pix, res );
    int best = imin( res[0], imin( res[1], res[2] ) );
    if( c->subme > 1 )
    {
        ORN(predict_8x8c)( PRED8C_P, pix );
        best = imin( best, mbcmp8x8( c, fenc, OR_FENC_STRIDE, pix, OR_FDEC_STRIDE ) );
        for( int m = 3; m < 9; m++ )
        {
            ORN(predict_8x8)( m, pix, edge );
            best = imin( best, mbcmp8x8( c, fenc, OR_FENC_STRIDE, pix, OR_FDEC_STRIDE ) );
        }
    }
    return ( ( best + 5*c->lambda ) >> DEPTH_SHIFT ) + 4;
}

void ORN(intra_costs)( const or_la_cfg *c, const pixel *fenc0, uint16_t *intra_cost )
{
    /* blocks slicetype_slice_cost never visits keep the 0xFFFF of frame.c:288-289 (memset -1 at allocation) */
    const int e = !c->do_edges;
    for( int by = 0; by < c->mb_h; by++ )
        for( int bx = 0; bx < c->mb_w; bx++ )
            intra_cost[by*c->mb_w+bx] = ( e && ( by < 1 || by > c->mb_h - 2 || bx < 1 || bx > c->mb_w - 2 ) ) ? 0xFFFF
                                      : (uint16_t)intra_cost_block( c, fenc0, bx, by );
}

/* ------------------------------------------------------------------------------------------------
 * S3: motion search of one 8x8 lowres block.  encoder/me.c:182-420,774-798 (x264_me_search_ref,
 * DIA and HEX branches) and :865-992 (refine_subpel), with the predictor helpers of
 * common/common.h:774-805.
 * ---------------------------------------------------------------------------------------------- */
typedef struct
{
    const or_la_cfg *c;
    const pixel *fenc;          /* 8x8 block, stride OR_FENC_STRIDE */
    const pixel *ref[4];        /* hpel planes at the block origin */
    const pixel *ref_w;         /* weighted full-pel plane at the block origin (== ref[0] if unweighted) */
    const or_weight *wt;
    int stride;
    int mvp[2];
    int spel_min[2], spel_max[2], fpel_min[2], fpel_max[2];
    /* results */
    int mv[2], cost;
} me_ctx;

static inline int mv_bits( const me_ctx *m, int qx, int qy ) /* qpel units */
{
    return m->c->cost_mv[qx - m->mvp[0]] + m->c->cost_mv[qy - m->mvp[1]];
}
static inline int fpel_cost( const me_ctx *m, int fx, int fy ) /* COST_MV, me.c:63-70 */
{
    return fpelcmp8x8( m->c, m->fenc, OR_FENC_STRIDE, m->ref_w + fy*m->stride + fx, m->stride ) + mv_bits( m, 4*fx, 4*fy );
}
static inline void fetch_block( const me_ctx *m, pixel *dst, int qx, int qy )
{
    ORN(mc_luma)( dst, 8, m->ref, m->stride, qx, qy, 8, 8, m->wt );
}
static inline int qpel_cost_sad( const me_ctx *m, int qx, int qy ) /* COST_MV_HPEL / COST_MV_SAD */
{
    pixel blk[64];
    fetch_block( m, blk, qx, qy );
    return fpelcmp8x8( m->c, m->fenc, OR_FENC_STRIDE, blk, 8 ) + mv_bits( m, qx, qy );
}
static inline int qpel_cost_satd( const me_ctx *m, int qx, int qy ) /* COST_MV_SATD (mbcmp) */
{
    pixel blk[64];
    fetch_block( m, blk, qx, qy );
    return mbcmp8x8( m->c, m->fenc, OR_FENC_STRIDE, blk, 8 ) + mv_bits( m, qx, qy );
}
static inline int in_fpel_range( const me_ctx *m, int fx, int fy ) /* CHECK_MVRANGE, me.c:204-209 */
{
    return fx >= m->fpel_min[0] && fx <= m->fpel_max[0] && fy >= m->fpel_min[1] && fy <= m->fpel_max[1];
}

static void refine_subpel( me_ctx *m, int hpel_iters, int qpel_iters );

static void me_search( me_ctx *m, const int16_t (*mvc)[2], int n_mvc )
{
    const or_la_cfg *c = m->c;
    int bmx, bmy, bcost;
    int bpred_cost = COST_MAX, bpred_mx = 0, bpred_my = 0;
    int pmv_x, pmv_y; /* the "pmv" the reference packs: qpel (subme>=3) or fullpel */
    int cand[8][2], n_cand;

    if( c->subpel_refine >= 3 )
    {
        /* predictor tested at quarter-pel precision (me.c:216-275) */
        bpred_mx = clip3( m->mvp[0], 4*m->fpel_min[0], 4*m->fpel_max[0] );
        bpred_my = clip3( m->mvp[1], 4*m->fpel_min[1], 4*m->fpel_max[1] );
        pmv_x = bpred_mx; pmv_y = bpred_my;
        bpred_cost = qpel_cost_sad( m, bpred_mx, bpred_my );
        int pmv_cost = bpred_cost;
        n_cand = 0;
        for( int i = 0; i < n_mvc; i++ )
        {
            int mx = mvc[i][0], my = mvc[i][1];
            if( ( !mx && !my ) || ( mx == pmv_x && my == pmv_y ) )
                continue;
            cand[n_cand][0] = clip3( mx, 4*m->fpel_min[0], 4*m->fpel_max[0] );
            cand[n_cand][1] = clip3( my, 4*m->fpel_min[1], 4*m->fpel_max[1] );
            n_cand++;
        }
        for( int i = 0; i < n_cand; i++ ) /* first strictly-lower candidate wins; pmv wins ties */
        {
            int cost = qpel_cost_sad( m, cand[i][0], cand[i][1] );
            if( cost < bpred_cost )
            {
                bpred_cost = cost; bpred_mx = cand[i][0]; bpred_my = cand[i][1];
            }
        }
        bmx = ( bpred_mx + 2 ) >> 2;
        bmy = ( bpred_my + 2 ) >> 2;
        if( ( bpred_mx | bpred_my ) & 3 )
        {
            bcost = COST_MAX;
            int cost = fpel_cost( m, bmx, bmy );
            if( cost < bcost ) bcost = cost;
        }
        else
            bcost = bpred_cost;
        if( pmv_x | pmv_y )
        {
            if( bmx | bmy )
            {
                int cost = fpel_cost( m, 0, 0 );
                if( cost < bcost ) { bcost = cost; bmx = 0; bmy = 0; }
            }
        }
        else if( pmv_cost < bcost )
        {
            bcost = pmv_cost; bmx = 0; bmy = 0;
        }
    }
    else
    {
        /* predictor rounded to full-pel (me.c:276-318) */
        bmx = clip3( ( m->mvp[0] + 2 ) >> 2, m->fpel_min[0], m->fpel_max[0] );
        bmy = clip3( ( m->mvp[1] + 2 ) >> 2, m->fpel_min[1], m->fpel_max[1] );
        pmv_x = bmx; pmv_y = bmy;
        bcost = fpelcmp8x8( c, m->fenc, OR_FENC_STRIDE, m->ref_w + bmy*m->stride + bmx, m->stride ); /* no mv bits */
        n_cand = 0;
        for( int i = 0; i < n_mvc; i++ )
        {
            int mx = ( mvc[i][0] + 2 ) >> 2, my = ( mvc[i][1] + 2 ) >> 2;
            if( ( !mx && !my ) || ( mx == pmv_x && my == pmv_y ) )
                continue;
            cand[n_cand][0] = clip3( mx, m->fpel_min[0], m->fpel_max[0] );
            cand[n_cand][1] = clip3( my, m->fpel_min[1], m->fpel_max[1] );
            n_cand++;
        }
        for( int i = 0; i < n_cand; i++ )
        {
            int cost = fpel_cost( m, cand[i][0], cand[i][1] );
            if( cost < bcost ) { bcost = cost; bmx = cand[i][0]; bmy = cand[i][1]; }
        }
        if( pmv_x | pmv_y )
        {
            int cost = fpel_cost( m, 0, 0 );
            if( cost < bcost ) { bcost = cost; bmx = 0; bmy = 0; }
        }
    }

    if( c->me_method == OR_ME_DIA )
    {
        /* radius-1 diamond, candidates in order up, down, left, right; earliest wins ties (me.c:322-342) */
        static const int dia[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
        int iters = c->me_range;
        do
        {
            int best = -1;
            for( int k = 0; k < 4; k++ )
            {
                int cost = fpel_cost( m, bmx + dia[k][0], bmy + dia[k][1] );
                if( cost < bcost ) { bcost = cost; best = k; }
            }
            if( best < 0 )
                break;
            bmx += dia[best][0]; bmy += dia[best][1];
        } while( --iters && in_fpel_range( m, bmx, bmy ) );
    }
    else
    {
        /* hexagon (me.c:344-420): full hexagon once, then half hexagons in the direction of travel,
         * finally the 8-neighbour square. */
        static const int hex[6][2] = { {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2} };
        int dir = -1;
        for( int k = 0; k < 6; k++ )
        {
            int cost = fpel_cost( m, bmx + hex[k][0], bmy + hex[k][1] );
            if( cost < bcost ) { bcost = cost; dir = k; }
        }
        if( dir >= 0 )
        {
            bmx += hex[dir][0]; bmy += hex[dir][1];
            for( int i = ( c->me_range >> 1 ) - 1; i > 0 && in_fpel_range( m, bmx, bmy ); i-- )
            {
                int best = -2;
                for( int k = -1; k <= 1; k++ )
                {
                    int d = ( dir + k + 6 ) % 6;
                    int cost = fpel_cost( m, bmx + hex[d][0], bmy + hex[d][1] );
                    if( cost < bcost ) { bcost = cost; best = k; }
                }
                if( best == -2 )
                    break;
                dir = ( dir + best + 6 ) % 6;
                bmx += hex[dir][0]; bmy += hex[dir][1];
            }
        }
        static const int sq[8][2] = { {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
        int best = -1;
        for( int k = 0; k < 8; k++ )
        {
            int cost = fpel_cost( m, bmx + sq[k][0], bmy + sq[k][1] );
            if( cost < bcost ) { bcost = cost; best = k; }
        }
        if( best >= 0 ) { bmx += sq[best][0]; bmy += sq[best][1]; }
    }

    /* back to quarter-pel (me.c:774-789) */
    if( c->subpel_refine < 3 )
    {
        m->cost = bcost;
        if( bmx == pmv_x && bmy == pmv_y )
            m->cost += mv_bits( m, 4*bmx, 4*bmy );
        m->mv[0] = 4*bmx; m->mv[1] = 4*bmy;
    }
    else if( bpred_cost < bcost )
    {
        m->mv[0] = bpred_mx; m->mv[1] = bpred_my; m->cost = bpred_cost;
    }
    else
    {
        m->mv[0] = 4*bmx; m->mv[1] = 4*bmy; m->cost = bcost;
    }

    if( c->subpel_refine >= 2 )
    {
        /* subpel_iterations[i_subpel_refine][2..3], me.c:38-50: refine 2 -> {1,0}, 4 -> {1,1}...
         * the lookahead only ever uses rows 2 and 4. */
        static const int iters[5][2] = { {0,0}, {0,0}, {1,0}, {1,0}, {1,1} };
        /* NOTE rows: {me_hpel, me_qpel} = subpel_iterations[r][2], [r][3] */
        refine_subpel( m, iters[c->subpel_refine][0], iters[c->subpel_refine][1] );
    }
}

static void refine_subpel( me_ctx *m, int hpel_iters, int qpel_iters )
{
    const or_la_cfg *c = m->c;
    int bmx = m->mv[0], bmy = m->mv[1], bcost = m->cost;

    if( hpel_iters )
    {
        if( c->subpel_refine < 3 ) /* try the sub-pel part of the predictor (me.c:889-895) */
        {
            int mx = clip3( m->mvp[0], m->spel_min[0]+2, m->spel_max[0]-2 );
            int my = clip3( m->mvp[1], m->spel_min[1]+2, m->spel_max[1]-2 );
            if( mx != bmx || my != bmy )
            {
                int cost = qpel_cost_sad( m, mx, my );
                if( cost < bcost ) { bcost = cost; bmx = mx; bmy = my; }
            }
        }
        /* half-pel diamond, order up, down, left, right (me.c:897-922) */
        static const int d2[4][2] = { {0,-2}, {0,2}, {-2,0}, {2,0} };
        for( int i = hpel_iters; i > 0; i-- )
        {
            int best = -1;
            for( int k = 0; k < 4; k++ )
            {
                int cost = qpel_cost_sad( m, bmx + d2[k][0], bmy + d2[k][1] );
                if( cost < bcost ) { bcost = cost; best = k; }
            }
            if( best < 0 )
                break;
            bmx += d2[best][0]; bmy += d2[best][1];
        }
    }

    /* re-cost with mbcmp when it differs from fpelcmp (me.c:925-929) */
    if( c->mbcmp_satd != c->fpelcmp_satd )
        bcost = qpel_cost_satd( m, bmx, bmy );

    /* quarter-pel diamond: skips the point it just came from (me.c:946-963) */
    int bdir = -1;
    static const int d1[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
    for( int i = qpel_iters; i > 0; i-- )
    {
        if( bmy <= m->spel_min[1] || bmy >= m->spel_max[1] || bmx <= m->spel_min[0] || bmx >= m->spel_max[0] )
            break;
        int odir = bdir, omx = bmx, omy = bmy;
        for( int k = 0; k < 4; k++ )
        {
            if( ( k ^ 1 ) == odir )
                continue;
            int cost = qpel_cost_satd( m, omx + d1[k][0], omy + d1[k][1] );
            if( cost < bcost ) { bcost = cost; bmx = omx + d1[k][0]; bmy = omy + d1[k][1]; bdir = k; }
        }
        if( bmx == omx && bmy == omy )
            break;
    }
    m->cost = bcost; m->mv[0] = bmx; m->mv[1] = bmy;
}

/* mv limits of a block: slicetype.c:550-562 */
static void block_limits( const or_la_cfg *c, int bx, int by, me_ctx *m )
{
    int range = 2*c->mv_range;
    m->spel_min[0] = imax( 4*( -8*bx - 12 ), -range );
    m->spel_max[0] = imin( 4*( 8*( c->mb_w - bx - 1 ) + 12 ), range - 1 );
    m->spel_min[1] = imax( 4*( -8*by - 12 ), -range );
    m->spel_max[1] = imin( 4*( 8*( c->mb_h - by - 1 ) + 12 ), range - 1 );
    for( int k = 0; k < 2; k++ )
    {
        m->fpel_min[k] = m->spel_min[k] >> 2;
        m->fpel_max[k] = m->spel_max[k] >> 2;
    }
}

/* S2 (search part) over a whole frame: reverse raster order, predictors from the already searched
 * right / below / below-left / below-right neighbours.  slicetype.c:654-709, :814-834. */
void ORN(search_field)( const or_la_cfg *c, const pixel *fenc0, const pixel *const ref[4], const pixel *ref_w,
                        const or_weight *wt, int16_t (*mvs)[2], int *mv_costs )
{
    const int W = c->mb_w, H = c->mb_h, ns = imax( c->n_slices, 1 ), de = !!c->do_edges;
    pixel fenc[8*OR_FENC_STRIDE];
    /* bands of slicetype.c:917-918; inside a band the rows and columns of slicetype.c:825-833.  Blocks that are not
     * visited keep what the caller put into mvs / mv_costs (zeros: frame.c:283-285) */
    for( int sl = 0; sl < ns; sl++ )
    {
    const int slice_start = ( H*sl + ns/2 ) / ns, slice_end = ( H*( sl + 1 ) + ns/2 ) / ns;
    for( int by = imin( slice_end - 1, H - 2 + de ); by >= imax( slice_start, 1 - de ); by-- )
        for( int bx = W - 2 + de; bx >= 1 - de; bx-- )
        {
            const int xy = by*W + bx, off = 8*( by*c->stride + bx );
            me_ctx m;
            memset( &m, 0, sizeof(m) );
            m.c = c; m.fenc = fenc; m.stride = c->stride; m.wt = wt;
            for( int k = 0; k < 4; k++ ) m.ref[k] = ref[k] + off;
            m.ref_w = ( wt && wt->on ) ? ref_w + off : m.ref[0];
            for( int y = 0; y < 8; y++ )
                memcpy( fenc + y*OR_FENC_STRIDE, fenc0 + off + y*c->stride, 8*sizeof(pixel) );
            block_limits( c, bx, by, &m );

            int16_t mvc[4][2] = { {0,0}, {0,0}, {0,0}, {0,0} };
            int n = 0;
#define ADD(i) { mvc[n][0] = mvs[i][0]; mvc[n][1] = mvs[i][1]; n++; }
            if( bx < W - 1 ) ADD( xy + 1 );
            if( by < slice_end - 1 )
            {
                ADD( xy + W );
                if( bx > 0 ) ADD( xy + W - 1 );
                if( bx < W - 1 ) ADD( xy + W + 1 );
            }
#undef ADD
            if( n <= 1 ) { m.mvp[0] = mvc[0][0]; m.mvp[1] = mvc[0][1]; }
            else
            {
                m.mvp[0] = median3( mvc[0][0], mvc[1][0], mvc[2][0] );
                m.mvp[1] = median3( mvc[0][1], mvc[1][1], mvc[2][1] );
            }
            int done = 0;
            if( !m.mvp[0] && !m.mvp[1] )
            {
                /* near-zero residual shortcut on the unweighted plane (slicetype.c:684-692) */
                m.cost = mbcmp8x8( c, fenc, OR_FENC_STRIDE, m.ref[0], c->stride );
                if( m.cost < 64 ) { m.mv[0] = m.mv[1] = 0; done = 1; }
            }
            if( !done )
            {
                me_search( &m, (const int16_t (*)[2])mvc, n );
                m.cost -= c->cost_mv[0];
                if( m.mv[0] | m.mv[1] )
                    m.cost += 5*c->lambda;
            }
            mvs[xy][0] = (int16_t)m.mv[0]; mvs[xy][1] = (int16_t)m.mv[1];
            mv_costs[xy] = m.cost;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * S2 (mode selection) + S4 (reductions) for one (p0,p1,b): slicetype.c:616-652 (bidir candidates),
 * :708-712, :758-790, :946-985.  Searches are inputs (mvs/costs per list).
 * ---------------------------------------------------------------------------------------------- */
static int bidir_cost( const or_la_cfg *c, const pixel *fenc, const pixel *const r0[4], const pixel *const r1[4],
                       const int mv0[2], const int mv1[2], int bipred_weight )
{
    pixel a[64], b[64], avg[64];
    if( c->subme <= 1 )
    {
        /* half-pel plane pick without interpolation (slicetype.c:582-589) */
        int h0 = ( ( mv0[0] & 2 ) >> 1 ) + ( mv0[1] & 2 ), h1 = ( ( mv1[0] & 2 ) >> 1 ) + ( mv1[1] & 2 );
        const pixel *s0 = r0[h0] + ( mv0[0] >> 2 ) + ( mv0[1] >> 2 ) * c->stride;
        const pixel *s1 = r1[h1] + ( mv1[0] >> 2 ) + ( mv1[1] >> 2 ) * c->stride;
        ORN(avg)( avg, 8, s0, c->stride, s1, c->stride, 8, 8, bipred_weight );
    }
    else
    {
        ORN(mc_luma)( a, 8, r0, c->stride, mv0[0], mv0[1], 8, 8, NULL );
        ORN(mc_luma)( b, 8, r1, c->stride, mv1[0], mv1[1], 8, 8, NULL );
        ORN(avg)( avg, 8, a, 8, b, 8, 8, 8, bipred_weight );
    }
    return mbcmp8x8( c, fenc, OR_FENC_STRIDE, avg, 8 );
}

void ORN(cell)( const or_la_cfg *c, const pixel *fenc0, const pixel *const ref0[4], const pixel *const ref1[4],
                int b_bidir, int dist_scale_factor, const or_weight *wt,
                const int16_t (*mvs0)[2], const int *costs0, const int16_t (*mvs1)[2], const int *costs1,
                const int16_t (*ref1_l0_mvs)[2], const uint16_t *intra_cost, const uint16_t *inv_qscale,
                int with_intra, uint16_t *lowres_costs, int *row_satds, int *row_satds_intra, or_cell_out *out )
{
    (void)wt;
    const int W = c->mb_w, H = c->mb_h;
    const int bipred_weight = c->weighted_bipred ? 64 - ( dist_scale_factor >> 2 ) : 32;
    const int is_intra_only = !mvs0 && !mvs1; /* p0 == p1 */
    memset( out, 0, sizeof(*out) );
    pixel fenc[8*OR_FENC_STRIDE];
    const int de = !!c->do_edges;
    for( int by = H - 1; by >= 0; by-- )
    {
        if( row_satds ) row_satds[by] = 0;
        if( row_satds_intra && with_intra ) row_satds_intra[by] = 0;
        if( !de && ( by < 1 || by > H - 2 ) )
            continue; /* slicetype.c:825-826: the row sums of unvisited rows are the zeros of :934-935 */
        for( int bx = W - 2 + de; bx >= 1 - de; bx-- )
        {
            const int xy = by*W + bx, off = 8*( by*c->stride + bx );
            const int scored = ( bx > 0 && bx < W-1 && by > 0 && by < H-1 ) || W <= 2 || H <= 2;
            int bcost = COST_MAX, list_used = 0;
            if( !is_intra_only )
            {
                me_ctx lim;
                block_limits( c, bx, by, &lim );
                for( int y = 0; y < 8; y++ )
                    memcpy( fenc + y*OR_FENC_STRIDE, fenc0 + off + y*c->stride, 8*sizeof(pixel) );
                const pixel *r0[4], *r1[4];
                for( int k = 0; k < 4; k++ ) { r0[k] = ref0[k] + off; r1[k] = b_bidir ? ref1[k] + off : NULL; }
                if( b_bidir )
                {
                    int dmv[2][2] = { {0,0}, {0,0} };
                    if( ref1_l0_mvs )
                    {
                        int rx = ref1_l0_mvs[xy][0], ry = ref1_l0_mvs[xy][1];
                        dmv[0][0] = ( rx*dist_scale_factor + 128 ) >> 8;
                        dmv[0][1] = ( ry*dist_scale_factor + 128 ) >> 8;
                        dmv[1][0] = dmv[0][0] - rx;
                        dmv[1][1] = dmv[0][1] - ry;
                        for( int l = 0; l < 2; l++ )
                            for( int k = 0; k < 2; k++ )
                            {
                                dmv[l][k] = clip3( dmv[l][k], lim.spel_min[k], lim.spel_max[k] );
                                if( c->subme <= 1 ) dmv[l][k] &= ~1;
                            }
                    }
                    int cost = bidir_cost( c, fenc, r0, r1, dmv[0], dmv[1], bipred_weight );
                    if( cost < bcost ) { bcost = cost; list_used = 3; }
                    if( dmv[0][0] | dmv[0][1] | dmv[1][0] | dmv[1][1] )
                    {
                        static const int zero[2] = { 0, 0 };
                        pixel avg[64];
                        ORN(avg)( avg, 8, r0[0], c->stride, r1[0], c->stride, 8, 8, bipred_weight );
                        (void)zero;
                        cost = mbcmp8x8( c, fenc, OR_FENC_STRIDE, avg, 8 );
                        if( cost < bcost ) { bcost = cost; list_used = 3; }
                    }
                }
                if( costs0[xy] < bcost ) { bcost = costs0[xy]; list_used = 1; }
                if( b_bidir )
                {
                    if( costs1[xy] < bcost ) { bcost = costs1[xy]; list_used = 2; }
                    int mv0[2] = { mvs0[xy][0], mvs0[xy][1] }, mv1[2] = { mvs1[xy][0], mvs1[xy][1] };
                    if( mv0[0] | mv0[1] | mv1[0] | mv1[1] )
                    {
                        int cost = 5*c->lambda + bidir_cost( c, fenc, r0, r1, mv0, mv1, bipred_weight );
                        if( cost < bcost ) { bcost = cost; list_used = 3; }
                    }
                }
            }
            int icost = intra_cost[xy];
            if( with_intra )
            {
                int icost_aq = c->aq_mode ? ( icost * inv_qscale[xy] + 128 ) >> 8 : icost;
                if( row_satds_intra ) row_satds_intra[by] += icost_aq;
                if( scored ) { out->intra_cost_est += icost; out->intra_cost_est_aq += icost_aq; }
            }
            bcost = ( bcost >> DEPTH_SHIFT ) + 4;
            if( !b_bidir )
            {
                int b_intra = icost < bcost;
                if( b_intra ) { bcost = icost; list_used = 0; }
                if( scored ) out->intra_mbs += b_intra;
            }
            if( !is_intra_only )
            {
                int bcost_aq = c->aq_mode ? ( bcost * inv_qscale[xy] + 128 ) >> 8 : bcost;
                if( row_satds ) row_satds[by] += bcost_aq;
                if( scored ) { out->cost_est += bcost; out->cost_est_aq += bcost_aq; }
            }
            lowres_costs[xy] = (uint16_t)( imin( bcost, 0x3FFF ) + ( list_used << 14 ) );
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Adaptive-quant input stage: encoder/ratecontrol.c:225-415, aq-mode 0..3.
 * Produces i_inv_qscale_factor (Q8) per MB and the luma sum / mean-removed ssd that
 * x264_weights_analyse reads (slicetype.c:301-306).  FP32 on purpose, same expression order as
 * the reference (which is built with -ffast-math; tests pin the bits against it).
 * luma/cb/cr are the picture as handed in (not mod16): coordinates clamp like frame.c:640-666.
 * ---------------------------------------------------------------------------------------------- */
/* x264_log2( x ) - bias.  The reference is built with -ffast-math and gcc folds the subtraction into the
 * integer-part term: lut[mantissa] + ( lz_lut[lz] - bias ).  That association is what its bits follow
 * (verified against oracle/_ref: 99/99 macroblocks exact, any other association < 63/99). */
static float lut_log2_minus( uint32_t x, float bias ) /* common/base.h:225-229 + tables.c:66-90 */
{
    static float lut[128];
    static int init = 0;
    if( !init )
    {
        for( int i = 0; i < 128; i++ )
            lut[i] = (float)( floor( log2( 1.0 + i/128.0 ) * 100000.0 + 0.5 ) / 100000.0 );
        init = 1;
    }
    int lz = __builtin_clz( x );
    return lut[( x << lz >> 24 ) & 0x7f] + ( (float)( 31 - lz ) - bias );
}

static int exp2fix8( float x ) /* common/base.h:217-223 + tables.c:58-64 */
{
    static uint8_t lut[64];
    static int init = 0;
    if( !init )
    {
        for( int i = 0; i < 64; i++ )
            lut[i] = (uint8_t)floor( ( pow( 2.0, i/64.0 ) - 1.0 ) * 256.0 + 0.5 );
        init = 1;
    }
    int i = (int)( x * ( -64.f/6.f ) + 512.5f );
    if( i < 0 ) return 0;
    if( i > 1023 ) return 0xffff;
    return ( lut[i & 63] + 256 ) << ( i >> 6 ) >> 8;
}

uint64_t ORN(aq_frame)( const pixel *luma, int stride, int width, int height, int mb_w, int mb_h,
                        const pixel *cb, const pixel *cr, int cstride, int aq_mode, float aq_strength,
                        uint16_t *inv_qscale, float *qp_offset, uint64_t *ssd_out )
{
    return ORN(aq_frame_fmt)( luma, stride, width, height, mb_w, mb_h, cb, cr, cstride, aq_mode, aq_strength, inv_qscale, qp_offset, ssd_out, 1, NULL );
}

/* chroma_format: 1 = 4:2:0 (8x8 chroma per macroblock and plane, shift 6), 2 = 4:2:2 (8x16, shift 7), 3 = 4:4:4 (16x16 like luma,
 * shift 8): ac_energy_plane / ac_energy_mb, ratecontrol.c:238-296 */
uint64_t ORN(aq_frame_fmt)( const pixel *luma, int stride, int width, int height, int mb_w, int mb_h,
                            const pixel *cb, const pixel *cr, int cstride, int aq_mode, float aq_strength,
                            uint16_t *inv_qscale, float *qp_offset, uint64_t *ssd_out, int chroma_format, const float *quant_offsets )
{
    /* quant_offsets: x264_picture_t.prop.quant_offsets, one float per macroblock added to the AQ offset (ratecontrol.c:318-326,396-397) */
    uint64_t sum_y = 0, ssd_y = 0;
    const float strength = aq_strength * 1.0397f;
    const int c444 = chroma_format == 3, c420 = chroma_format != 2 && chroma_format != 3;
    const int cw = c444 ? width : ( width + 1 ) >> 1, chh = c420 ? ( height + 1 ) >> 1 : height;
    const int cbw = c444 ? 16 : 8, cbh = c420 ? 8 : 16, cshift = c444 ? 8 : c420 ? 6 : 7;
    float sum_r4 = 0.f, sum_r8 = 0.f;
    float *av_r8 = malloc( sizeof(float) * mb_w * mb_h );
    for( int my = 0; my < mb_h; my++ )
        for( int mx = 0; mx < mb_w; mx++ )
        {
            uint32_t s = 0, q = 0;
            for( int y = 0; y < 16; y++ )
            {
                const pixel *row = luma + (size_t)imin( 16*my + y, height-1 ) * stride;
                for( int x = 0; x < 16; x++ )
                {
                    uint32_t v = row[imin( 16*mx + x, width-1 )];
                    s += v; q += v*v;
                }
            }
            sum_y += s; ssd_y += q;
            uint32_t energy = q - (uint32_t)( (uint64_t)s * s >> 8 );
            for( int p = 0; p < 2; p++ )
            {
                const pixel *pl = p ? cr : cb;
                if( !pl ) continue;
                uint32_t cs = 0, cq = 0;
                for( int y = 0; y < cbh; y++ )
                {
                    const pixel *row = pl + (size_t)imin( cbh*my + y, chh-1 ) * cstride;
                    for( int x = 0; x < cbw; x++ )
                    {
                        uint32_t v = row[imin( cbw*mx + x, cw-1 )];
                        cs += v; cq += v*v;
                    }
                }
                energy += cq - (uint32_t)( (uint64_t)cs * cs >> cshift );
            }
            if( aq_mode == 1 && aq_strength != 0.f )
            {
                float qp_adj = strength * lut_log2_minus( energy > 1 ? energy : 1, 14.427f + 2*( OR_DEPTH - 8 ) );
                if( quant_offsets ) qp_adj += quant_offsets[my*mb_w+mx];
                if( qp_offset ) qp_offset[my*mb_w+mx] = qp_adj;
                inv_qscale[my*mb_w+mx] = (uint16_t)exp2fix8( qp_adj );
            }
            else if( ( aq_mode == 2 || aq_mode == 3 ) && aq_strength != 0.f )
            {
                /* first pass of the auto-variance modes (ratecontrol.c:354-371) as the reference build computes it
                 * (gcc -O3 -ffast-math, disassembly of oracle/_ref): powf( x, 0.125f ) is three square roots, and the
                 * "qp_adj * qp_adj" that is averaged is the intermediate fourth root, not a product */
                float x = (float)(uint64_t)energy;
#if OR_DEPTH > 8
                x *= 1.f / ( 1 << ( 2*( OR_DEPTH - 8 ) ) );
#endif
                x += 1.f;
                const float r4 = sqrtf( sqrtf( x ) ), r8 = sqrtf( r4 );
                sum_r4 += r4;        /* sequential sums in raster order: the rounding of every step counts */
                sum_r8 += r8;
                av_r8[my*mb_w+mx] = r8;
            }
            else
            {
                /* AQ off, or on with strength 0 (the MB-tree case): only the caller's offsets remain (:316-334) */
                float qp_adj = aq_mode && quant_offsets ? quant_offsets[my*mb_w+mx] : 0.f;
                if( qp_offset ) qp_offset[my*mb_w+mx] = qp_adj;
                inv_qscale[my*mb_w+mx] = aq_mode && quant_offsets ? (uint16_t)exp2fix8( qp_adj ) : 256;
            }
        }
    if( ( aq_mode == 2 || aq_mode == 3 ) && aq_strength != 0.f )
    {
        /* ratecontrol.c:372-377, :380-393 in the operation order of the reference build */
        const float cnt = (float)( mb_w*mb_h );
        const float avg_pow2 = sum_r4 / cnt, avg = sum_r8 / cnt;
        const float str = aq_strength * avg;
        const float avg_adj = ( ( 14.f - avg_pow2 ) * 0.5f ) / avg + avg;
        for( int i = 0; i < mb_w*mb_h; i++ )
        {
            const float q = av_r8[i];
            float qp_adj = ( q - avg_adj ) * str;
            if( aq_mode == 3 )
                qp_adj = ( 1.f - 14.f / ( q*q ) ) * aq_strength + qp_adj;
            if( quant_offsets ) qp_adj += quant_offsets[i];
            if( qp_offset ) qp_offset[i] = qp_adj;
            inv_qscale[i] = (uint16_t)exp2fix8( qp_adj );
        }
    }
    free( av_r8 );
    uint64_t n = (uint64_t)( 16*mb_w ) * ( 16*mb_h );
    /* i_pixel_sum is uint32_t in the reference (common/frame.h:140): the total has wrapped mod 2^32 before ratecontrol.c:410-414
     * squares it (bright 10-bit pictures from ~4.2 Mpx on) */
    const uint64_t s32 = (uint32_t)sum_y;
    if( ssd_out ) *ssd_out = ssd_y - ( s32*s32 + n/2 ) / n;
    return s32; /* i_pixel_sum[0] as the reference stores it */
}

/* slicetype_frame_cost_recalculate (slicetype.c:999-1024): the frame cost of an already evaluated cell under new per-MB
 * quantiser offsets (MB-tree), without touching the searches: cost14 * exp2fix8(qp_offset), row sums, and the frame sum
 * over the same interior blocks as slicetype_frame_cost. */
int ORN(frame_cost_recalculate)( int mb_w, int mb_h, const uint16_t *lowres_costs, const float *qp_offset, int *row_satds )
{
    int score = 0;
    for( int y = mb_h - 1; y >= 0; y-- )
    {
        row_satds[y] = 0;
        for( int x = mb_w - 1; x >= 0; x-- )
        {
            int cost = lowres_costs[y*mb_w + x] & 0x3FFF;
            cost = ( cost * exp2fix8( qp_offset[y*mb_w + x] ) + 128 ) >> 8;
            row_satds[y] += cost;
            if( ( y > 0 && y < mb_h - 1 && x > 0 && x < mb_w - 1 ) || mb_w <= 2 || mb_h <= 2 )
                score += cost;
        }
    }
    return score;
}

/* ================================================================================================
 * SURVEY 8(f) rank 3 groundwork: the main-encode motion search, encoder/me.c:182-798 (x264_me_search_ref with
 * every method: DIA, HEX, UMH, ESA, TESA) and :865-992 (refine_subpel, all subpel_iterations rows), for any
 * partition size, luma only (b_chroma_me off), single reference, no weights.  Pinned against the reference
 * through oracle/ref_harness.c:rh_me_search (tests/test_me_full_vs_ref.py).
 * ================================================================================================ */
static const uint8_t mef_size[7][2] = { {16,16}, {16,8}, {8,16}, {8,8}, {8,4}, {4,8}, {4,4} };
static const uint8_t mef_subpel_iterations[12][4] = /* me.c:38-50 */
    { {0,0,0,0}, {1,1,0,0}, {0,1,1,0}, {0,2,1,0}, {0,2,1,1}, {0,2,1,2}, {0,0,2,2}, {0,0,2,2}, {0,0,4,10}, {0,0,4,10}, {0,0,4,10}, {0,0,4,10} };

typedef struct
{
    const ORN(me_full) *p;
    int bw, bh;
    int bmx, bmy, bcost;
} mef;

static inline int mef_fpelcmp( const mef *s, const pixel *b, int sb )
{
    return s->p->fpelcmp_satd ? ORN(satd)( s->p->fenc, OR_FENC_STRIDE, b, sb, s->bw, s->bh ) : ORN(sad)( s->p->fenc, OR_FENC_STRIDE, b, sb, s->bw, s->bh );
}
static inline int mef_mbcmp( const mef *s, const pixel *b, int sb )
{
    return s->p->mbcmp_satd ? ORN(satd)( s->p->fenc, OR_FENC_STRIDE, b, sb, s->bw, s->bh ) : ORN(sad)( s->p->fenc, OR_FENC_STRIDE, b, sb, s->bw, s->bh );
}
static inline int mef_bits_q( const mef *s, int qx, int qy ) { return s->p->cost_mv[qx - s->p->mvp[0]] + s->p->cost_mv[qy - s->p->mvp[1]]; }
static inline int mef_bits_f( const mef *s, int fx, int fy ) { return mef_bits_q( s, 4*fx, 4*fy ); } /* BITS_MVD */
static inline int mef_cost_f( const mef *s, int fx, int fy ) /* the cost COST_MV computes */
{
    return mef_fpelcmp( s, s->p->ref[0] + (long)fy * s->p->stride + fx, s->p->stride ) + mef_bits_f( s, fx, fy );
}
static inline void mef_try_f( mef *s, int fx, int fy ) /* COST_MV */
{
    int c = mef_cost_f( s, fx, fy );
    if( c < s->bcost ) { s->bcost = c; s->bmx = fx; s->bmy = fy; }
}
static inline int mef_cost_q( const mef *s, int qx, int qy, int use_mbcmp ) /* COST_MV_HPEL / COST_MV_SAD / COST_MV_SATD */
{
    pixel pix[16*16];
    ORN(mc_luma)( pix, 16, s->p->ref, s->p->stride, qx, qy, s->bw, s->bh, NULL );
    return ( use_mbcmp ? mef_mbcmp( s, pix, 16 ) : mef_fpelcmp( s, pix, 16 ) ) + mef_bits_q( s, qx, qy );
}
static inline int mef_in_range( const mef *s, int fx, int fy ) /* CHECK_MVRANGE */
{
    return fx >= s->p->lim_min[0] && fx <= s->p->lim_max[0] && fy >= s->p->lim_min[1] && fy <= s->p->lim_max[1];
}
/* COST_MV_X4 relative to (omx, omy), candidates applied in order */
static void mef_x4( mef *s, int omx, int omy, const int d[4][2] )
{
    for( int k = 0; k < 4; k++ )
        mef_try_f( s, omx + d[k][0], omy + d[k][1] );
}
static void mef_cross( mef *s, int omx, int omy, int start, int x_max, int y_max ) /* CROSS, me.c:139-166 */
{
    const ORN(me_full) *p = s->p;
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

static void mef_hex2( mef *s, int me_range ) /* the HEX branch incl. the square refine, me.c:344-420 */
{
    static const int8_t hex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
    static const uint8_t mod6m1[8] = { 5,0,1,2,3,4,5,0 };
    static const int8_t square1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
    static const int8_t first[6][2] = { {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2} };
    int bmx = s->bmx, bmy = s->bmy, bcost = s->bcost, dir = -1;
    for( int k = 0; k < 6; k++ ) /* packed (cost<<3)+k+2 with COPY1_IF_LT: lowest cost, first on ties */
    {
        int c = mef_cost_f( s, bmx + first[k][0], bmy + first[k][1] );
        if( c < bcost ) { bcost = c; dir = k; }
    }
    if( dir >= 0 )
    {
        bmx += hex2[dir+1][0]; bmy += hex2[dir+1][1];
        s->bmx = bmx; s->bmy = bmy;
        for( int i = ( me_range >> 1 ) - 1; i > 0 && mef_in_range( s, bmx, bmy ); i-- )
        {
            int best = -1;
            for( int k = 0; k < 3; k++ )
            {
                int c = mef_cost_f( s, bmx + hex2[dir+k][0], bmy + hex2[dir+k][1] );
                if( c < bcost ) { bcost = c; best = k; }
            }
            if( best < 0 )
                break;
            dir += best - 1;
            dir = mod6m1[dir+1];
            bmx += hex2[dir+1][0]; bmy += hex2[dir+1][1];
        }
    }
    int sq = 0;
    for( int k = 1; k <= 8; k++ )
    {
        int c = mef_cost_f( s, bmx + square1[k][0], bmy + square1[k][1] );
        if( c < bcost ) { bcost = c; sq = k; }
    }
    s->bmx = bmx + square1[sq][0]; s->bmy = bmy + square1[sq][1]; s->bcost = bcost;
}

typedef struct { int sad; int mx, my; } mef_mvsad;

static void mef_refine_subpel( mef *s, int mv[2], int *cost, int *cost_mv, int hpel_iters, int qpel_iters )
{
    const ORN(me_full) *p = s->p;
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
        static const int d2[4][2] = { {0,-2}, {0,2}, {-2,0}, {2,0} };
        for( int i = hpel_iters; i > 0; i-- )
        {
            int best = -1, omx = bmx, omy = bmy;
            for( int k = 0; k < 4; k++ )
            {
                int c = mef_cost_q( s, omx + d2[k][0], omy + d2[k][1], 0 );
                if( c < bcost ) { bcost = c; best = k; }
            }
            if( best < 0 )
                break;
            bmx = omx + d2[best][0]; bmy = omy + d2[best][1];
        }
    }
    if( p->mbcmp_satd != p->fpelcmp_satd ) /* h->pixf.mbcmp_unaligned[0] != h->pixf.fpelcmp[0] */
        bcost = mef_cost_q( s, bmx, bmy, 1 );
    static const int d1[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
    if( p->subpel_refine != 1 )
    {
        int bdir = -1;
        for( int i = qpel_iters; i > 0; i-- )
        {
            if( bmy <= p->spel_min[1] || bmy >= p->spel_max[1] || bmx <= p->spel_min[0] || bmx >= p->spel_max[0] )
                break;
            int odir = bdir, omx = bmx, omy = bmy;
            for( int k = 0; k < 4; k++ )
            {
                if( ( k ^ 1 ) == odir )
                    continue;
                int c = mef_cost_q( s, omx + d1[k][0], omy + d1[k][1], 1 );
                if( c < bcost ) { bcost = c; bmx = omx + d1[k][0]; bmy = omy + d1[k][1]; bdir = k; }
            }
            if( bmx == omx && bmy == omy )
                break;
        }
    }
    else if( bmy > p->spel_min[1] && bmy < p->spel_max[1] && bmx > p->spel_min[0] && bmx < p->spel_max[0] )
    {
        int omx = bmx, omy = bmy; /* subme 1: one quarter-pel diamond with fpelcmp */
        for( int k = 0; k < 4; k++ )
        {
            int c = mef_cost_q( s, omx + d1[k][0], omy + d1[k][1], 0 );
            if( c < bcost ) { bcost = c; bmx = omx + d1[k][0]; bmy = omy + d1[k][1]; }
        }
    }
    mv[0] = bmx; mv[1] = bmy; *cost = bcost;
    *cost_mv = mef_bits_q( s, bmx, bmy );
}

void ORN(me_search_full)( const ORN(me_full) *p, const int16_t (*mvc)[2], int n_mvc, int out[4] )
{
    mef S, *s = &S;
    s->p = p; s->bw = mef_size[p->i_pixel][0]; s->bh = mef_size[p->i_pixel][1];
    const int mv_x_min = p->lim_min[0], mv_y_min = p->lim_min[1], mv_x_max = p->lim_max[0], mv_y_max = p->lim_max[1];
    int me_range = p->me_range;
    int bpred_cost = COST_MAX, bpred_mx = 0, bpred_my = 0, pmx, pmy, pmv_nonzero;
    s->bcost = COST_MAX; s->bmx = s->bmy = 0;
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
            mef_try_f( s, s->bmx, s->bmy ); /* bcost is COST_MAX here: always taken */
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

    switch( p->me_method )
    {
        case 0: /* DIA, me.c:322-342 */
        {
            static const int d[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
            int i = me_range;
            do
            {
                int best = -1, bmx = s->bmx, bmy = s->bmy;
                for( int k = 0; k < 4; k++ )
                {
                    int c = mef_cost_f( s, bmx + d[k][0], bmy + d[k][1] );
                    if( c < s->bcost ) { s->bcost = c; best = k; }
                }
                if( best < 0 )
                    break;
                s->bmx = bmx + d[best][0]; s->bmy = bmy + d[best][1];
            } while( --i && mef_in_range( s, s->bmx, s->bmy ) );
            break;
        }
        case 1:
            mef_hex2( s, me_range );
            break;
        case 2: /* UMH, me.c:422-618 */
        {
            static const uint8_t pixel_size_shift[7] = { 0, 1, 1, 2, 3, 3, 4 };
            static const int dia1[4][2] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
            int ucost1, ucost2, cross_start = 1, omx, omy;
            ucost1 = s->bcost;
            mef_x4( s, pmx, pmy, dia1 );
            if( pmx | pmy )
                mef_x4( s, 0, 0, dia1 );
            omx = ( pmx | pmy ) ? 0 : pmx; omy = ( pmx | pmy ) ? 0 : pmy; /* what DIA1_ITER left in omx/omy */
            if( p->i_pixel == 6 ) /* PIXEL_4x4 */
            {
                mef_hex2( s, me_range );
                break;
            }
            ucost2 = s->bcost;
            if( ( s->bmx | s->bmy ) && ( ( s->bmx - pmx ) | ( s->bmy - pmy ) ) )
            {
                omx = s->bmx; omy = s->bmy;
                mef_x4( s, omx, omy, dia1 );
            }
            if( s->bcost == ucost2 )
                cross_start = 3;
            omx = s->bmx; omy = s->bmy;
#define MEF_SAD_THRESH( v ) ( s->bcost < ( (v) >> pixel_size_shift[p->i_pixel] ) )
            int done = 0;
            if( s->bcost == ucost2 && MEF_SAD_THRESH( 2000 ) )
            {
                static const int o1[4][2] = { {0,-2}, {-1,-1}, {1,-1}, {-2,0} }, o2[4][2] = { {2,0}, {-1,1}, {1,1}, {0,2} };
                mef_x4( s, omx, omy, o1 );
                mef_x4( s, omx, omy, o2 );
                if( s->bcost == ucost1 && MEF_SAD_THRESH( 500 ) )
                    done = 1;
                else if( s->bcost == ucost2 )
                {
                    int range = ( me_range >> 1 ) | 1;
                    static const int o3[4][2] = { {-1,-2}, {1,-2}, {-2,-1}, {2,-1} }, o4[4][2] = { {-2,1}, {2,1}, {-1,2}, {1,2} };
                    mef_cross( s, omx, omy, 3, range, range );
                    mef_x4( s, omx, omy, o3 );
                    mef_x4( s, omx, omy, o4 );
                    if( s->bcost == ucost2 )
                        done = 1;
                    else
                        cross_start = range + 2;
                }
            }
            if( done )
                break;
            if( n_mvc )
            {
                static const uint8_t range_mul[4][4] = { {3,3,4,4}, {3,4,4,4}, {4,4,4,5}, {4,4,5,6} };
                int mvd, denom = 1;
                if( n_mvc == 1 )
                    mvd = p->i_pixel == 0 ? 25 : abs( p->mvp[0] - mvc[0][0] ) + abs( p->mvp[1] - mvc[0][1] );
                else
                {
                    denom = n_mvc - 1;
                    mvd = 0;
                    if( p->i_pixel != 0 )
                    {
                        mvd = abs( p->mvp[0] - mvc[0][0] ) + abs( p->mvp[1] - mvc[0][1] );
                        denom++;
                    }
                    for( int i = 0; i < n_mvc - 1; i++ ) /* x264_predictor_difference */
                        mvd += abs( mvc[i][0] - mvc[i+1][0] ) + abs( mvc[i][1] - mvc[i+1][1] );
                }
                int sad_ctx = MEF_SAD_THRESH( 1000 ) ? 0 : MEF_SAD_THRESH( 2000 ) ? 1 : MEF_SAD_THRESH( 4000 ) ? 2 : 3;
                int mvd_ctx = mvd < 10*denom ? 0 : mvd < 20*denom ? 1 : mvd < 40*denom ? 2 : 3;
                me_range = me_range * range_mul[mvd_ctx][sad_ctx] >> 2;
            }
#undef MEF_SAD_THRESH
            mef_cross( s, omx, omy, cross_start, me_range, me_range >> 1 );
            {
                static const int o5[4][2] = { {-2,-2}, {-2,2}, {2,-2}, {2,2} };
                mef_x4( s, omx, omy, o5 );
            }
            omx = s->bmx; omy = s->bmy;
            int i = 1;
            do
            {
                static const int8_t hex4[16][2] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                                    {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };
                const int near_edge = 4*i > imin( imin( mv_x_max - omx, omx - mv_x_min ), imin( mv_y_max - omy, omy - mv_y_min ) );
                for( int j = 0; j < 16; j++ )
                {
                    int mx = omx + hex4[j][0]*i, my = omy + hex4[j][1]*i;
                    if( !near_edge || mef_in_range( s, mx, my ) )
                        mef_try_f( s, mx, my );
                }
            } while( ++i <= me_range >> 2 );
            if( s->bmy <= mv_y_max && s->bmy >= mv_y_min && s->bmx <= mv_x_max && s->bmx >= mv_x_min )
                mef_hex2( s, me_range );
            break;
        }
        default: /* ESA (3) / TESA (4), me.c:620-772 */
        {
            const int min_x = imax( s->bmx - me_range, mv_x_min ), min_y = imax( s->bmy - me_range, mv_y_min );
            const int max_x = imin( s->bmx + me_range, mv_x_max ), max_y = imin( s->bmy + me_range, mv_y_max );
            const int width = ( max_x - min_x + 3 ) & ~3;
            if( p->me_method == 3 )
            {
                /* successive elimination only discards candidates that cannot beat the current best (sum|d| >= |sum d|),
                 * so the result is the plain exhaustive scan in its order -- including the up to three columns past
                 * max_x that rounding the width to a multiple of four adds */
                for( int my = min_y; my <= max_y; my++ )
                {
                    if( s->bcost <= p->cost_mv[4*my - p->mvp[1]] )
                        continue;
                    for( int mx = min_x; mx < min_x + width; mx++ )
                        mef_try_f( s, mx, my );
                }
                break;
            }
            /* TESA: ADS threshold, SAD threshold, keep the best few SADs, then SATD */
            const uint16_t *sums_base = p->integral;
            int enc_dc[4];
            const int small = p->i_pixel > 3; /* sad_size: 8x8 quadrants for sizes >= 8x8, else 4x4 */
            int delta = small ? 4 : 8;
            {
                const pixel *f = p->fenc;
                const int q[4][2] = { {0,0}, {delta,0}, {0,delta}, {delta,delta} };
                for( int k = 0; k < 4; k++ )
                {
                    int sum = 0;
                    for( int y = 0; y < delta; y++ )
                        for( int x = 0; x < delta; x++ )
                            sum += f[( q[k][1] + y ) * OR_FENC_STRIDE + q[k][0] + x];
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
            mef_mvsad *mvsads = malloc( sizeof( mef_mvsad ) * (size_t)( width + 4 ) * ( max_y - min_y + 2 ) );
            int16_t *xs = malloc( sizeof( int16_t ) * ( width + 64 ) );
            uint16_t *cost_fpel_mvx = malloc( sizeof( uint16_t ) * ( width + 4 ) );
            for( int x = 0; x < width; x++ )
                cost_fpel_mvx[x] = p->cost_mv[4*( min_x + x ) - p->mvp[0]];
            int nmvsad = 0, limit;
            int sad_thresh = me_range <= 16 ? 10 : me_range <= 24 ? 11 : 12;
            int bsad = ORN(sad)( p->fenc, OR_FENC_STRIDE, p->ref[0] + (long)s->bmy * p->stride + s->bmx, p->stride, s->bw, s->bh ) + mef_bits_f( s, s->bmx, s->bmy );
            for( int my = min_y; my <= max_y; my++ )
            {
                int ycost = p->cost_mv[4*my - p->mvp[1]];
                if( bsad <= ycost )
                    continue;
                bsad -= ycost;
                int xn = ORN(ads)( ads_n, enc_dc, sums_base + min_x + (long)my * p->stride, delta, cost_fpel_mvx, xs, width, bsad * 17 >> 4 );
                for( int i = 0; i < xn; i++ )
                {
                    int mx = min_x + xs[i];
                    /* the reference indexes its x-cost table with the offset from min_x here (me.c:671,688: cost_fpel_mvx[xs[i]],
                     * not cost_fpel_mvx[min_x + xs[i]] as in the ads call), so the SAD stage charges the cost of column xs[i] */
                    int sad = ORN(sad)( p->fenc, OR_FENC_STRIDE, p->ref[0] + (long)my * p->stride + mx, p->stride, s->bw, s->bh ) +
                              p->cost_mv[4 * xs[i] - p->mvp[0]];
                    if( sad < bsad * sad_thresh >> 3 )
                    {
                        if( sad < bsad ) bsad = sad;
                        mvsads[nmvsad].sad = sad + ycost; mvsads[nmvsad].mx = mx; mvsads[nmvsad].my = my;
                        nmvsad++;
                    }
                }
                bsad += ycost;
            }
            limit = me_range >> 1;
            sad_thresh = bsad * sad_thresh >> 3;
            while( nmvsad > limit*2 && sad_thresh > bsad )
            {
                int i = 0;
                sad_thresh = ( sad_thresh + bsad ) >> 1;
                while( i < nmvsad && mvsads[i].sad <= sad_thresh )
                    i++;
                for( int j = i; j < nmvsad; j++ )
                {
                    mvsads[i] = mvsads[j];
                    if( mvsads[j].sad <= sad_thresh ) /* i += (sad - (sad_thresh+1)) >> 31 with the sign trick */
                        i++;
                }
                nmvsad = i;
            }
            while( nmvsad > limit )
            {
                int bi = 0;
                for( int i = 1; i < nmvsad; i++ )
                    if( mvsads[i].sad > mvsads[bi].sad )
                        bi = i;
                nmvsad--;
                mvsads[bi] = mvsads[nmvsad];
            }
            for( int i = 0; i < nmvsad; i++ )
                mef_try_f( s, mvsads[i].mx, mvsads[i].my );
            free( mvsads ); free( xs ); free( cost_fpel_mvx );
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
