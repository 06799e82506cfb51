// Block metrics of x264_pixel_function_t that only the main encode calls: ssd (common/pixel.c:85-151),
// sa8d (:334-381), var (:183-201), hadamard_ac (:383-435), vsad (:716-723), asd8 (:747-754), as batched device entries over a
// raster of blocks of a device-resident plane.  One thread per block: horizontally adjacent blocks are adjacent threads, so a
// wave's row loads are one contiguous run of W*64 pixels.  The arithmetic is plain C++ marked BM_HD so that tests/tools can
// compile the very same functions for the host and check them against the oracle without a GPU (the library itself has no
// host path: only the kernels below are reachable from the C ABI).
#pragma once
#include <stdint.h>

#ifndef BM_HD
#define BM_HD __host__ __device__ __forceinline__
#endif

template <typename T, int W, int H>
BM_HD uint64_t bm_ssd( const T *a, long sa, const T *b, long sb )
{
    int s = 0;
    for( int y = 0; y < H; y++ )
        for( int x = 0; x < W; x++ )
        {
            const int d = (int)a[y * sa + x] - (int)b[y * sb + x];
            s += d * d;
        }
    return (uint64_t)(uint32_t)s;
}

template <typename T, int W, int H>
BM_HD uint64_t bm_var( const T *a, long sa )
{
    uint32_t sum = 0, sqr = 0;
    for( int y = 0; y < H; y++ )
        for( int x = 0; x < W; x++ )
        {
            const uint32_t v = a[y * sa + x];
            sum += v; sqr += v * v;
        }
    return sum + ( (uint64_t)sqr << 32 );
}

// in-place 1-D Hadamard butterflies over N = 4 or 8 values spaced `step` apart
template <int N>
BM_HD void bm_hadamard_1d( int *v, int step )
{
    for( int span = 1; span < N; span <<= 1 )
        for( int i = 0; i < N; i++ )
            if( !( i & span ) )
            {
                const int p = v[i * step], q = v[( i + span ) * step];
                v[i * step] = p + q; v[( i + span ) * step] = p - q;
            }
}

// sum of |H_N * D * H_N^T| of the N x N block of differences a - b (b == nullptr: of a itself)
template <typename T, int N>
BM_HD int bm_hadamard_abs( const T *a, long sa, const T *b, long sb )
{
    int d[N * N];
    for( int y = 0; y < N; y++ )
        for( int x = 0; x < N; x++ )
            d[N * y + x] = (int)a[y * sa + x] - ( b ? (int)b[y * sb + x] : 0 );
    for( int y = 0; y < N; y++ ) bm_hadamard_1d<N>( d + N * y, 1 );
    for( int x = 0; x < N; x++ ) bm_hadamard_1d<N>( d + x, N );
    int s = 0;
    for( int i = 0; i < N * N; i++ ) s += d[i] < 0 ? -d[i] : d[i];
    return s;
}

template <typename T, int W> // W = 8 or 16, square
BM_HD uint64_t bm_sa8d( const T *a, long sa, const T *b, long sb )
{
    int s = 0;
    for( int y = 0; y < W; y += 8 )
        for( int x = 0; x < W; x += 8 )
            s += bm_hadamard_abs<T, 8>( a + y * sa + x, sa, b + y * sb + x, sb );
    return (uint64_t)(uint32_t)( ( s + 2 ) >> 2 );
}

template <typename T, int W, int H> // multiples of 8
BM_HD uint64_t bm_hadamard_ac( const T *pix, long stride )
{
    uint64_t sum4 = 0, sum8 = 0;
    for( int by = 0; by < H; by += 8 )
        for( int bx = 0; bx < W; bx += 8 )
        {
            const T *p = pix + by * stride + bx;
            int dc = 0, s4 = 0;
            for( int y = 0; y < 8; y++ )
                for( int x = 0; x < 8; x++ )
                    dc += p[y * stride + x];
            for( int y = 0; y < 8; y += 4 )
                for( int x = 0; x < 8; x += 4 )
                    s4 += bm_hadamard_abs<T, 4>( p + y * stride + x, stride, (const T *)nullptr, 0 );
            sum4 += (uint64_t)( s4 - dc );
            sum8 += (uint64_t)( bm_hadamard_abs<T, 8>( p, stride, (const T *)nullptr, 0 ) - dc );
        }
    return ( ( sum8 >> 2 ) << 32 ) + ( (uint32_t)sum4 >> 1 );
}

template <typename T, int H> // 16 wide, rows 0 .. H-1
BM_HD uint64_t bm_vsad( const T *src, long stride )
{
    int score = 0;
    for( int i = 1; i < H; i++ )
        for( int j = 0; j < 16; j++ )
        {
            const int d = (int)src[( i - 1 ) * stride + j] - (int)src[i * stride + j];
            score += d < 0 ? -d : d;
        }
    return (uint64_t)(uint32_t)score;
}

template <typename T, int H> // 8 wide
BM_HD uint64_t bm_asd8( const T *a, long sa, const T *b, long sb )
{
    int sum = 0;
    for( int y = 0; y < H; y++ )
        for( int x = 0; x < 8; x++ )
            sum += (int)a[y * sa + x] - (int)b[y * sb + x];
    return (uint64_t)(uint32_t)( sum < 0 ? -sum : sum );
}

enum { BM_SSD = 0, BM_SA8D = 1, BM_VAR = 2, BM_HADAMARD_AC = 3, BM_VSAD = 4, BM_ASD8 = 5 };

// one metric of one W x H block: the single switch both the kernel and the host check go through
template <typename T, int METRIC, int W, int H>
BM_HD uint64_t bm_block( const T *a, long sa, const T *b, long sb )
{
    if( METRIC == BM_SSD ) return bm_ssd<T, W, H>( a, sa, b, sb );
    if( METRIC == BM_SA8D ) return bm_sa8d<T, W>( a, sa, b, sb );
    if( METRIC == BM_VAR ) return bm_var<T, W, H>( a, sa );
    if( METRIC == BM_HADAMARD_AC ) return bm_hadamard_ac<T, W, H>( a, sa );
    if( METRIC == BM_VSAD ) return bm_vsad<T, H>( a, sa );
    return bm_asd8<T, H>( a, sa, b, sb );
}

#ifdef __HIPCC__
template <typename T, int METRIC, int W, int H>
__global__ __launch_bounds__( 64 ) void block_metric_kernel( const T *__restrict__ a, const T *__restrict__ b, long stride, int blocks_w,
                                                             unsigned long long *__restrict__ out )
{
    const int bx = blockIdx.x * 64 + threadIdx.x, by = blockIdx.y;
    if( bx >= blocks_w )
        return;
    const long o = (long)by * H * stride + (long)bx * W;
    out[(long)by * blocks_w + bx] = bm_block<T, METRIC, W, H>( a + o, stride, b ? b + o : nullptr, stride );
}
#endif
