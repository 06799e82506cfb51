// sub8x8_dct8 + quant_8x8 of one 8x8 block (common/dct.c:332-366, common/quant.c:50-62,64-72) in the frame form of SURVEY 8f
// rank 4: the block of the plane `fenc` minus the block of the prediction plane `fdec`, transformed and quantised.  The
// arithmetic is BM_HD like block_metrics.h: the device kernel below and the host check in tests/tools compile the same function.
#pragma once
#include <stdint.h>

#ifndef BM_HD
#define BM_HD __host__ __device__ __forceinline__
#endif

BM_HD void dq_fdct8_1d( const int *in, int step, int *out, int ostep )
{
    const int s07 = in[0] + in[7 * step], s16 = in[step] + in[6 * step], s25 = in[2 * step] + in[5 * step], s34 = in[3 * step] + in[4 * step];
    const int d07 = in[0] - in[7 * step], d16 = in[step] - in[6 * step], d25 = in[2 * step] - in[5 * step], d34 = in[3 * step] - in[4 * step];
    const int e0 = s07 + s34, e1 = s16 + s25, e2 = s07 - s34, e3 = s16 - s25;
    const int o4 = d16 + d25 + ( d07 + ( d07 >> 1 ) ), o5 = d07 - d34 - ( d25 + ( d25 >> 1 ) );
    const int o6 = d07 + d34 - ( d16 + ( d16 >> 1 ) ), o7 = d16 - d25 + ( d34 + ( d34 >> 1 ) );
    out[0] = e0 + e1; out[ostep] = o4 + ( o7 >> 2 ); out[2 * ostep] = e2 + ( e3 >> 1 ); out[3 * ostep] = o5 + ( o6 >> 2 );
    out[4 * ostep] = e0 - e1; out[5 * ostep] = o6 - ( o5 >> 2 ); out[6 * ostep] = ( e2 >> 1 ) - e3; out[7 * ostep] = ( o4 >> 2 ) - o7;
}

// coefficients land in the reference's order (coefficient (u, v), u = horizontal frequency, at out[8*u + v]); returns quant_8x8's nz
template <typename T, typename C>
BM_HD int dq_block8x8( const T *fenc, long fenc_stride, const T *fdec, long fdec_stride, const uint32_t *mf, const uint32_t *bias, C *out )
{
    int d[64], t[64], o[64];
    for( int y = 0; y < 8; y++ )
        for( int x = 0; x < 8; x++ )
            d[8 * y + x] = (int)fenc[y * fenc_stride + x] - (int)fdec[y * fdec_stride + x];
    for( int x = 0; x < 8; x++ ) dq_fdct8_1d( d + x, 8, t + x, 8 );     // columns: t[8*v + x]
    for( int v = 0; v < 8; v++ ) dq_fdct8_1d( t + 8 * v, 1, o + v, 8 ); // rows:    o[8*u + v]
    int nz = 0;
    for( int k = 0; k < 64; k++ )
    {
        int v = (C)o[k];
        if( v > 0 ) v = (int)( ( bias[k] + (uint32_t)v ) * mf[k] >> 16 );
        else v = -(int)( ( bias[k] + (uint32_t)( -v ) ) * mf[k] >> 16 );
        out[k] = (C)v;
        nz |= out[k];
    }
    return nz != 0;
}

#ifdef __HIPCC__
struct QuantTab8
{
    uint32_t mf[64], bias[64];
};

// one thread per 8x8 block, horizontally adjacent blocks on adjacent threads (a wave's row loads are contiguous)
template <typename T, typename C>
__global__ __launch_bounds__( 64 ) void frame_dct_quant8x8_kernel( const T *__restrict__ fenc, long fenc_stride, const T *__restrict__ fdec, long fdec_stride,
                                                                   int blocks_w, QuantTab8 q, C *__restrict__ coefs, uint8_t *__restrict__ nz_out )
{
    const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y;
    if( bx >= blocks_w )
        return;
    C out[64];
    const int nz = dq_block8x8<T, C>( fenc + (long)8 * by * fenc_stride + 8 * bx, fenc_stride, fdec + (long)8 * by * fdec_stride + 8 * bx, fdec_stride,
                                      q.mf, q.bias, out );
    C *dst = coefs + ( (size_t)by * blocks_w + bx ) * 64;
    for( int k = 0; k < 64; k++ )
        dst[k] = out[k];
    nz_out[(size_t)by * blocks_w + bx] = (uint8_t)nz;
}
#endif
