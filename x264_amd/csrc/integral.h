// Integral (box-sum) planes of a reference frame for the exhaustive searches: what x264_frame_filter builds
// with integral_init4h/8h/4v/8v (common/mc.c:424-456, :757-783; layout common/frame.c:240-256) -- sum8[y][x] = sum of the 8x8 box
// with its top-left sample at (x, y) of the padded luma plane, modulo 2^16, and sum4 likewise for 4x4 boxes, `lower` elements
// further on.  The reference accumulates running sums row by row; on the device every entry is computed independently in two
// separable passes (8 + 8 loads per entry instead of 64).  tests/test_me_full_vs_ref.py pins the reference's planes to exactly
// these box sums wherever a whole box fits; entries whose box would leave the plane are not written (the searches never read them).
#pragma once
#include <stdint.h>

#ifndef BM_HD
#define BM_HD __host__ __device__ __forceinline__
#endif

// pass 1: horizontal sums of 8 and of 4 consecutive samples starting at x (x + 8 <= width resp. x + 4 <= width)
template <typename T>
BM_HD void ii_row_sums( const T *row, int x, int width, uint16_t *h8, uint16_t *h4 )
{
    unsigned s4 = 0, s8 = 0;
    if( x + 4 <= width )
    {
        for( int i = 0; i < 4; i++ ) s4 += row[x + i];
        *h4 = (uint16_t)s4;
    }
    if( x + 8 <= width )
    {
        s8 = s4;
        for( int i = 4; i < 8; i++ ) s8 += row[x + i];
        *h8 = (uint16_t)s8;
    }
}
// pass 2: vertical sum of n row sums
BM_HD uint16_t ii_col_sum( const uint16_t *h, long stride, int n )
{
    unsigned s = 0;
    for( int j = 0; j < n; j++ ) s += h[j * stride];
    return (uint16_t)s;
}

#ifdef __HIPCC__
// Border replication of x264_frame_expand_border / x264_frame_expand_border_filtered (common/frame.c:535-623) for a whole padded
// plane at once: every sample outside the rectangle [x0, x1] x [y0, y1] becomes the nearest sample of the rectangle.  src == dst
// is allowed (only samples outside the rectangle are written, only samples inside it are read); with src != dst the rectangle is
// copied as well (src addressed with its own stride and origin).
template <typename T>
__global__ __launch_bounds__( 256 ) void expand_border_kernel( T *__restrict__ dst, long dst_stride, const T *src, long src_stride, int in_place,
                                                              int x0, int x1, int y0, int y1, int xmin, int xmax, int ymin )
{
    const int x = xmin + blockIdx.x * 256 + threadIdx.x, y = ymin + blockIdx.y;
    if( x > xmax )
        return;
    const int cx = x < x0 ? x0 : x > x1 ? x1 : x, cy = y < y0 ? y0 : y > y1 ? y1 : y;
    if( in_place && cx == x && cy == y )
        return;
    dst[(long)y * dst_stride + x] = src[(long)cy * src_stride + cx];
}

template <typename T>
__global__ __launch_bounds__( 256 ) void integral_rows_kernel( const T *__restrict__ plane, long stride, int width, int height,
                                                               uint16_t *__restrict__ h8, uint16_t *__restrict__ h4 )
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if( x >= width || y >= height )
        return;
    ii_row_sums<T>( plane + (long)y * stride, x, width, h8 + (long)y * stride + x, h4 + (long)y * stride + x );
}
__global__ __launch_bounds__( 256 ) void integral_cols_kernel( const uint16_t *__restrict__ h8, const uint16_t *__restrict__ h4, long stride, int width, int height,
                                                               uint16_t *__restrict__ sum8, uint16_t *__restrict__ sum4 )
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if( x >= width || y >= height )
        return;
    if( x + 8 <= width && y + 8 <= height )
        sum8[(long)y * stride + x] = ii_col_sum( h8 + (long)y * stride + x, stride, 8 );
    if( x + 4 <= width && y + 4 <= height )
        sum4[(long)y * stride + x] = ii_col_sum( h4 + (long)y * stride + x, stride, 4 );
}
#endif
