// vtable_blocks.h -- the remaining per-macroblock entries of x264_dct_function_t (common/dct.h:29-59), x264_quant_function_t
// (common/quant.h:30-45) and x264_pixel_function_t (var2, ads; common/pixel.h:95-117) in batch form: n independent calls on the
// reference's macroblock-local buffers (fenc rows FENC_STRIDE = 16 samples apart, fdec rows FDEC_STRIDE = 32 apart).
// The arithmetic is BM_HD (shared with the host check in tests/tools/vtable_blocks_host.cpp); the kernels below give one thread a
// call (dct / quant / var2) or one wave a call (ads: ordered compaction of a row of candidates).
#pragma once
#include <stdint.h>

#ifndef BM_HD
#define BM_HD __host__ __device__ __forceinline__
#endif
#include "dct_quant_block.h"

#define VT_FENC_STRIDE 16
#define VT_FDEC_STRIDE 32

// kinds of x264hip_dct_batch == the dctf entries (coefficient counts per call in vt_dct_coefs)
enum { VT_SUB4X4 = 0, VT_SUB8X8 = 1, VT_SUB16X16 = 2, VT_SUB8X8_DCT8 = 3, VT_SUB16X16_DCT8 = 4, VT_SUB8X8_DC = 5, VT_SUB8X16_DC = 6, VT_DCT4X4DC = 7, VT_DCT2X4DC = 8 };
BM_HD int vt_dct_coefs( int kind )
{
    return kind == 0 ? 16 : kind == 1 ? 64 : kind == 2 ? 256 : kind == 3 ? 64 : kind == 4 ? 256 : kind == 5 ? 4 : kind == 6 ? 8 : kind == 7 ? 16 : kind == 8 ? 8 : 0;
}
BM_HD bool vt_dct_in_place( int kind ) { return kind >= 7; } // the DC transforms work on coefficients, not on pixels

BM_HD void vt_fdct4_1d( const int *in, int step, int *out, int ostep )
{
    const int s03 = in[0] + in[3 * step], s12 = in[step] + in[2 * step], d03 = in[0] - in[3 * step], d12 = in[step] - in[2 * step];
    out[0] = s03 + s12; out[ostep] = 2 * d03 + d12; out[2 * ostep] = s03 - s12; out[3 * ostep] = d03 - 2 * d12;
}
// sub4x4_dct (dct.c:157-175): coefficient (u, v), u = horizontal frequency, lands at out[4*u + v]
template <typename T, typename C>
BM_HD void vt_sub4x4( C *out, const T *fenc, const T *fdec )
{
    int d[16], t[16], o[16];
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            d[4 * y + x] = (int)fenc[y * VT_FENC_STRIDE + x] - (int)fdec[y * VT_FDEC_STRIDE + x];
    for( int y = 0; y < 4; y++ ) vt_fdct4_1d( d + 4 * y, 1, t + y, 4 );
    for( int u = 0; u < 4; u++ ) vt_fdct4_1d( t + 4 * u, 1, o + 4 * u, 1 );
    for( int i = 0; i < 16; i++ ) out[i] = (C)o[i];
}
// sub8x8_dct8 (dct.c:332-366): first pass down the columns, then along the rows
template <typename T, typename C>
BM_HD void vt_sub8x8_dct8( C *out, const T *fenc, const T *fdec )
{
    int d[64], t[64], o[64];
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
            d[8 * y + x] = (int)fenc[y * VT_FENC_STRIDE + x] - (int)fdec[y * VT_FDEC_STRIDE + x];
    for( int x = 0; x < 8; x++ ) dq_fdct8_1d( d + x, 8, t + x, 8 );
    for( int v = 0; v < 8; v++ ) dq_fdct8_1d( t + 8 * v, 1, o + v, 8 );
    for( int i = 0; i < 64; i++ ) out[i] = (C)o[i];
}
template <typename T>
BM_HD int vt_dc_sum4( const T *fenc, const T *fdec ) // sub4x4_dct_dc (dct.c:207-214)
{
    int s = 0;
    for( int y = 0; y < 4; y++ )
        for( int x = 0; x < 4; x++ )
            s += (int)fenc[y * VT_FENC_STRIDE + x] - (int)fdec[y * VT_FDEC_STRIDE + x];
    return s;
}

template <typename T, typename C>
BM_HD void vt_dct( int kind, C *out, const T *fenc, const T *fdec )
{
    switch( kind )
    {
        case VT_SUB4X4: vt_sub4x4<T, C>( out, fenc, fdec ); break;
        case VT_SUB8X8: // dct.c:177-183: the four 4x4 blocks in raster order
            for( int i = 0; i < 4; i++ )
                vt_sub4x4<T, C>( out + 16 * i, fenc + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FENC_STRIDE, fdec + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FDEC_STRIDE );
            break;
        case VT_SUB16X16: // dct.c:185-191: four 8x8 quadrants of four blocks each
            for( int j = 0; j < 4; j++ )
                for( int i = 0; i < 4; i++ )
                    vt_sub4x4<T, C>( out + 64 * j + 16 * i, fenc + 8 * ( j & 1 ) + 8 * ( j >> 1 ) * VT_FENC_STRIDE + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FENC_STRIDE,
                                     fdec + 8 * ( j & 1 ) + 8 * ( j >> 1 ) * VT_FDEC_STRIDE + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FDEC_STRIDE );
            break;
        case VT_SUB8X8_DCT8: vt_sub8x8_dct8<T, C>( out, fenc, fdec ); break;
        case VT_SUB16X16_DCT8: // dct.c:368-386
            for( int j = 0; j < 4; j++ )
                vt_sub8x8_dct8<T, C>( out + 64 * j, fenc + 8 * ( j & 1 ) + 8 * ( j >> 1 ) * VT_FENC_STRIDE, fdec + 8 * ( j & 1 ) + 8 * ( j >> 1 ) * VT_FDEC_STRIDE );
            break;
        case VT_SUB8X8_DC: // dct.c:216-229: 2x2 transform of the four DC sums
        {
            int s[4];
            for( int i = 0; i < 4; i++ )
                s[i] = vt_dc_sum4<T>( fenc + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FENC_STRIDE, fdec + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FDEC_STRIDE );
            out[0] = (C)( s[0] + s[1] + s[2] + s[3] ); out[1] = (C)( s[0] + s[1] - s[2] - s[3] );
            out[2] = (C)( s[0] - s[1] + s[2] - s[3] ); out[3] = (C)( s[0] - s[1] - s[2] + s[3] );
            break;
        }
        case VT_SUB8X16_DC: // dct.c:231-270: 2x4 transform of the eight DC sums
        {
            int a[8], h0[4], h1[4];
            for( int i = 0; i < 8; i++ )
                a[i] = vt_dc_sum4<T>( fenc + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FENC_STRIDE, fdec + 4 * ( i & 1 ) + 4 * ( i >> 1 ) * VT_FDEC_STRIDE );
            for( int r = 0; r < 4; r++ ) { h0[r] = a[2 * r] + a[2 * r + 1]; h1[r] = a[2 * r] - a[2 * r + 1]; }
            const int p0 = h0[0] + h0[1], p1 = h0[2] + h0[3], p2 = h1[0] + h1[1], p3 = h1[2] + h1[3];
            const int q0 = h0[0] - h0[1], q1 = h0[2] - h0[3], q2 = h1[0] - h1[1], q3 = h1[2] - h1[3];
            out[0] = (C)( p0 + p1 ); out[1] = (C)( p2 + p3 ); out[2] = (C)( p0 - p1 ); out[3] = (C)( p2 - p3 );
            out[4] = (C)( q0 - q1 ); out[5] = (C)( q2 - q3 ); out[6] = (C)( q0 + q1 ); out[7] = (C)( q2 + q3 );
            break;
        }
        case VT_DCT4X4DC: // dct.c:47-77, in place: 4x4 Hadamard, second pass rounds ( x + 1 ) >> 1
        {
            int d[16], t[16];
            for( int i = 0; i < 16; i++ ) d[i] = out[i];
            for( int i = 0; i < 4; i++ )
            {
                const int s01 = d[4 * i] + d[4 * i + 1], d01 = d[4 * i] - d[4 * i + 1], s23 = d[4 * i + 2] + d[4 * i + 3], d23 = d[4 * i + 2] - d[4 * i + 3];
                t[i] = s01 + s23; t[4 + i] = s01 - s23; t[8 + i] = d01 - d23; t[12 + i] = d01 + d23;
            }
            for( int i = 0; i < 4; i++ )
            {
                const int s01 = t[4 * i] + t[4 * i + 1], d01 = t[4 * i] - t[4 * i + 1], s23 = t[4 * i + 2] + t[4 * i + 3], d23 = t[4 * i + 2] - t[4 * i + 3];
                out[4 * i] = (C)( ( s01 + s23 + 1 ) >> 1 ); out[4 * i + 1] = (C)( ( s01 - s23 + 1 ) >> 1 );
                out[4 * i + 2] = (C)( ( d01 - d23 + 1 ) >> 1 ); out[4 * i + 3] = (C)( ( d01 + d23 + 1 ) >> 1 );
            }
            break;
        }
        case VT_DCT2X4DC: // dct.c:109-143, in place on the eight DC values
        {
            const int a0 = out[0] + out[1], a1 = out[2] + out[3], a2 = out[4] + out[5], a3 = out[6] + out[7];
            const int a4 = out[0] - out[1], a5 = out[2] - out[3], a6 = out[4] - out[5], a7 = out[6] - out[7];
            const int b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7, b4 = a0 - a1, b5 = a2 - a3, b6 = a4 - a5, b7 = a6 - a7;
            out[0] = (C)( b0 + b1 ); out[1] = (C)( b2 + b3 ); out[2] = (C)( b0 - b1 ); out[3] = (C)( b2 - b3 );
            out[4] = (C)( b4 - b5 ); out[5] = (C)( b6 - b7 ); out[6] = (C)( b4 + b5 ); out[7] = (C)( b6 + b7 );
            break;
        }
    }
}

// kinds of x264hip_quant_batch == the quantf entries; coefficient counts per call in vt_quant_coefs
enum { VT_QUANT_4X4 = 0, VT_QUANT_8X8 = 1, VT_QUANT_4X4X4 = 2, VT_QUANT_4X4_DC = 3, VT_QUANT_2X2_DC = 4 };
BM_HD int vt_quant_coefs( int kind ) { return kind == 0 ? 16 : kind == 1 ? 64 : kind == 2 ? 64 : kind == 3 ? 16 : kind == 4 ? 4 : 0; }

template <typename C>
BM_HD int vt_quant_one( C *coef, uint32_t mf, uint32_t bias ) // QUANT_ONE (quant.c:37-48)
{
    int v = *coef;
    if( v > 0 ) v = (int)( ( bias + (uint32_t)v ) * mf >> 16 );
    else v = -(int)( ( bias + (uint32_t)( -v ) ) * mf >> 16 );
    *coef = (C)v;
    return *coef;
}
// returns what the reference entry returns: the non-zero flag, for quant_4x4x4 the 4-bit mask of its four blocks
template <typename C, typename U>
BM_HD int vt_quant( int kind, C *coef, const U *mf, const U *bias, int mf_dc, int bias_dc )
{
    int nz = 0;
    switch( kind )
    {
        case VT_QUANT_4X4: for( int i = 0; i < 16; i++ ) nz |= vt_quant_one( coef + i, mf[i], bias[i] ); return !!nz;
        case VT_QUANT_8X8: for( int i = 0; i < 64; i++ ) nz |= vt_quant_one( coef + i, mf[i], bias[i] ); return !!nz;
        case VT_QUANT_4X4X4: // quant.c:74-84
        {
            int mask = 0;
            for( int j = 0; j < 4; j++ )
            {
                nz = 0;
                for( int i = 0; i < 16; i++ ) nz |= vt_quant_one( coef + 16 * j + i, mf[i], bias[i] );
                mask |= ( !!nz ) << j;
            }
            return mask;
        }
        case VT_QUANT_4X4_DC: for( int i = 0; i < 16; i++ ) nz |= vt_quant_one( coef + i, (uint32_t)mf_dc, (uint32_t)bias_dc ); return !!nz;
        case VT_QUANT_2X2_DC: for( int i = 0; i < 4; i++ ) nz |= vt_quant_one( coef + i, (uint32_t)mf_dc, (uint32_t)bias_dc ); return !!nz;
    }
    return -1;
}

// var2_8x8 / var2_8x16 (pixel.c:206-231): the two chroma halves of the macroblock buffers (U at column 0, V at column stride/2)
template <typename T>
BM_HD int vt_var2( const T *fenc, const T *fdec, int h, int ssd[2] )
{
    int sum_u = 0, sum_v = 0, sqr_u = 0, sqr_v = 0;
    const int shift = h == 16 ? 7 : 6;
    for( int y = 0; y < h; y++ )
        for( int x = 0; x < 8; x++ )
        {
            const int du = (int)fenc[y * VT_FENC_STRIDE + x] - (int)fdec[y * VT_FDEC_STRIDE + x];
            const int dv = (int)fenc[y * VT_FENC_STRIDE + x + VT_FENC_STRIDE / 2] - (int)fdec[y * VT_FDEC_STRIDE + x + VT_FDEC_STRIDE / 2];
            sum_u += du; sum_v += dv; sqr_u += du * du; sqr_v += dv * dv;
        }
    ssd[0] = sqr_u; ssd[1] = sqr_v;
    return sqr_u - (int)( (long long)sum_u * sum_u >> shift ) + sqr_v - (int)( (long long)sum_v * sum_v >> shift );
}

// one candidate of ads1 / ads2 / ads4 (pixel.c:759-803)
BM_HD int vt_ads_one( int n_dc, const int *enc_dc, const uint16_t *sums, int delta, int cost )
{
    int a = enc_dc[0] - sums[0];
    int ads = ( a < 0 ? -a : a ) + cost;
    if( n_dc == 2 ) { a = enc_dc[1] - sums[delta]; ads += a < 0 ? -a : a; }
    else if( n_dc == 4 )
    {
        a = enc_dc[1] - sums[8]; ads += a < 0 ? -a : a;
        a = enc_dc[2] - sums[delta]; ads += a < 0 ? -a : a;
        a = enc_dc[3] - sums[delta + 8]; ads += a < 0 ? -a : a;
    }
    return ads;
}

#ifdef __HIPCC__
template <typename T, typename C>
__global__ __launch_bounds__( 64 ) void vt_dct_kernel( int kind, int n, const T *__restrict__ fenc, const T *__restrict__ fdec, C *__restrict__ coefs )
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if( i >= n )
        return;
    vt_dct<T, C>( kind, coefs + (size_t)i * vt_dct_coefs( kind ), fenc + (size_t)i * 16 * VT_FENC_STRIDE, fdec + (size_t)i * 16 * VT_FDEC_STRIDE );
}
template <typename C, typename U>
__global__ __launch_bounds__( 64 ) void vt_quant_kernel( int kind, int n, C *__restrict__ coefs, const U *__restrict__ mf, const U *__restrict__ bias, int mf_dc, int bias_dc,
                                                         int *__restrict__ nz )
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if( i >= n )
        return;
    nz[i] = vt_quant<C, U>( kind, coefs + (size_t)i * vt_quant_coefs( kind ), mf, bias, mf_dc, bias_dc );
}
template <typename T>
__global__ __launch_bounds__( 64 ) void vt_var2_kernel( int h, int n, const T *__restrict__ fenc, const T *__restrict__ fdec, int *__restrict__ var, int *__restrict__ ssd )
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if( i >= n )
        return;
    int s[2];
    var[i] = vt_var2<T>( fenc + (size_t)i * 16 * VT_FENC_STRIDE, fdec + (size_t)i * 16 * VT_FDEC_STRIDE, h, s );
    ssd[2 * i] = s[0]; ssd[2 * i + 1] = s[1];
}

// ads: one wave per call.  The reference appends the index of every candidate below the threshold in scan order; here 64
// candidates are judged at once and the survivors keep that order through a ballot + prefix population count.
struct VtAdsCall
{
    int n_dc, delta, width, thresh;
    int enc_dc[4];
    long long sums_off, cost_off, mvs_off; // element offsets into the batch's sums / cost_mvx / mvs arrays
};
__global__ __launch_bounds__( 64 ) void vt_ads_kernel( int n, const VtAdsCall *__restrict__ calls, const uint16_t *__restrict__ sums, const uint16_t *__restrict__ cost_mvx,
                                                       int16_t *__restrict__ mvs, int *__restrict__ counts )
{
    const int c = blockIdx.x, lane = threadIdx.x;
    if( c >= n )
        return;
    const VtAdsCall A = calls[c];
    int nmv = 0;
    for( int base = 0; base < A.width; base += 64 )
    {
        const int i = base + lane;
        bool keep = false;
        if( i < A.width )
            keep = vt_ads_one( A.n_dc, A.enc_dc, sums + A.sums_off + i, A.delta, cost_mvx[A.cost_off + i] ) < A.thresh;
        const unsigned long long m = __builtin_amdgcn_ballot_w64( keep );
        if( keep )
            mvs[A.mvs_off + nmv + __popcll( m & ( ( 1ull << lane ) - 1 ) )] = (int16_t)i;
        nmv += __popcll( m );
    }
    if( lane == 0 )
        counts[c] = nmv;
}
#endif
