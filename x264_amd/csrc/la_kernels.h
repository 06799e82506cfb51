// la_kernels.h -- the streaming / block-parallel kernels around the search: half-resolution planes,
// adaptive-quant statistics, intra costs, weighted planes and costs, mode selection + reductions.
#pragma once
#include "device_common.h"

// ---- M1: lowres planes (common/mc.c:458-507) + border (common/frame.c:535-554,627-631) ----------------
// One thread -> 4 horizontally adjacent output pixels of all four planes, addressed in the PADDED
// domain: border pixels are produced by clamping the lowres coordinate, so there is no second pass
// and every store is a full, aligned 4-pixel vector.  Source coordinates clamp to the picture, which
// reproduces the mod16 replication (frame.c:640-666) and the +1 row/column duplication (mc.c:466-468).
// per-frame ingest descriptor: a batch of frames is one launch of each ingest kernel (blockIdx.z = frame)
struct PutDesc
{
    const void *src;              // full-resolution luma
    const void *cb, *cr;          // optional chroma planes (AQ energy)
    int src_stride, cstride;
    void *planes;                 // 4 padded lowres planes
    uint16_t *inv_qscale;
    uint2 *mb_sums;
    unsigned long long *frame_sums;
    float *qp_aq, *qp;
    uint16_t *intra_cost;
    int aq_on, pad_;
};

template <typename T>
__global__ __launch_bounds__( 256 ) void lowres_kernel( const PutDesc *descs, PutDesc single, int width, int height,
                                                        int plane_elems, int stride, int lw, int lh )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.z ) : single;
    const T *__restrict__ src = (const T *)D.src;
    T *__restrict__ planes = (T *)D.planes;
    const int src_stride = D.src_stride;
    const int pw4 = ( lw + 2 * LA_PAD ) >> 2;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x;
    const int Y = blockIdx.y;
    if( X4 >= pw4 )
        return;
    const int y = iclip3( Y - LA_PAD, 0, lh - 1 );
    const T *r0 = src + (size_t)imin2( 2 * y, height - 1 ) * src_stride;
    const T *r1 = src + (size_t)imin2( 2 * y + 1, height - 1 ) * src_stride;
    const T *r2 = src + (size_t)imin2( 2 * y + 2, height - 1 ) * src_stride;
    T o0[4], oh[4], ov[4], oc[4];
    const int xb = X4 * 4 - LA_PAD;
    if( xb >= 0 && 2 * ( xb + 3 ) + 2 < width )
    {
        // interior: columns 2xb .. 2xb+8 are in range; vector-load 8 pixels + 1 per row
        int a[9], b[9], c[9];
        T va[8], vb[8], vc[8];
        __builtin_memcpy( va, r0 + 2 * xb, 8 * sizeof( T ) );
        __builtin_memcpy( vb, r1 + 2 * xb, 8 * sizeof( T ) );
        __builtin_memcpy( vc, r2 + 2 * xb, 8 * sizeof( T ) );
#pragma unroll
        for( int i = 0; i < 8; i++ ) { a[i] = va[i]; b[i] = vb[i]; c[i] = vc[i]; }
        a[8] = r0[2 * xb + 8]; b[8] = r1[2 * xb + 8]; c[8] = r2[2 * xb + 8];
        int t[9], u[9];
#pragma unroll
        for( int i = 0; i < 9; i++ ) { t[i] = ( a[i] + b[i] + 1 ) >> 1; u[i] = ( b[i] + c[i] + 1 ) >> 1; }
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            o0[i] = (T)( ( t[2 * i] + t[2 * i + 1] + 1 ) >> 1 );
            oh[i] = (T)( ( t[2 * i + 1] + t[2 * i + 2] + 1 ) >> 1 );
            ov[i] = (T)( ( u[2 * i] + u[2 * i + 1] + 1 ) >> 1 );
            oc[i] = (T)( ( u[2 * i + 1] + u[2 * i + 2] + 1 ) >> 1 );
        }
    }
    else
    {
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            const int x = iclip3( xb + i, 0, lw - 1 );
            const int x0 = imin2( 2 * x, width - 1 ), x1 = imin2( 2 * x + 1, width - 1 ), x2 = imin2( 2 * x + 2, width - 1 );
            int t0 = ( r0[x0] + r1[x0] + 1 ) >> 1, t1 = ( r0[x1] + r1[x1] + 1 ) >> 1, t2 = ( r0[x2] + r1[x2] + 1 ) >> 1;
            int u0 = ( r1[x0] + r2[x0] + 1 ) >> 1, u1 = ( r1[x1] + r2[x1] + 1 ) >> 1, u2 = ( r1[x2] + r2[x2] + 1 ) >> 1;
            o0[i] = (T)( ( t0 + t1 + 1 ) >> 1 ); oh[i] = (T)( ( t1 + t2 + 1 ) >> 1 );
            ov[i] = (T)( ( u0 + u1 + 1 ) >> 1 ); oc[i] = (T)( ( u1 + u2 + 1 ) >> 1 );
        }
    }
    const size_t o = (size_t)Y * stride + X4 * 4; // padded origin: plane base + 0 is padded (−32,−32)
    __builtin_memcpy( planes + o, o0, 4 * sizeof( T ) );
    __builtin_memcpy( planes + plane_elems + o, oh, 4 * sizeof( T ) );
    __builtin_memcpy( planes + 2 * (size_t)plane_elems + o, ov, 4 * sizeof( T ) );
    __builtin_memcpy( planes + 3 * (size_t)plane_elems + o, oc, 4 * sizeof( T ) );
}

// The strip copy of the four planes read by the search (me_search.h): strip k of a plane = columns 8k .. 8k+15 of every row, 16
// samples per row, rows one after the other; the strips of plane p start 4 + 2p planes behind the row-major planes.  A wave copies 64
// consecutive rows of one strip: the lanes read the 16 samples of their row out of the row-major plane lowres_kernel has just written
// (L2 hits) and the wave's store is one contiguous kilobyte.  (Writing the strips from lowres_kernel itself, in the 4-sample pieces
// its threads own, reached memory as 13 MB per 1080p frame for 7.5 MB of planes: profiles/r02_traffic.json.)
template <typename T>
__global__ __launch_bounds__( 256 ) void strips_kernel( const PutDesc *descs, PutDesc single, int plane_elems, int stride, int rows )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.z ) : single;
    const T *__restrict__ planes = (const T *)D.planes;
    T *__restrict__ strips = (T *)D.planes + 4 * (size_t)plane_elems;
    const int n_strips = stride >> 3;
    const int sidx = blockIdx.y * 4 + ( threadIdx.x >> 6 ), p = sidx / n_strips, k = sidx - p * n_strips;
    const int Y = blockIdx.x * 64 + ( threadIdx.x & 63 );
    if( p >= 4 || Y >= rows )
        return;
    const T *src = planes + (size_t)p * plane_elems + (size_t)Y * stride + 8 * k;
    T v[16];
    __builtin_memcpy( v, src, 8 * sizeof( T ) );
    if( k + 1 < n_strips )
        __builtin_memcpy( v + 8, src + 8, 8 * sizeof( T ) );
    else
        for( int i = 8; i < 16; i++ ) v[i] = 0; // beyond the plane: never part of a block a search may read
    T *dst = strips + 2 * (size_t)p * plane_elems + strip_layout::row_off( k, Y, rows );
    __builtin_memcpy( dst, v, 16 * sizeof( T ) );
}

// lowres_kernel + strips_kernel in one pass (the default ingest): a workgroup produces a tile of LT_ROWS x (LT_COLS + 8) padded samples of the four
// planes in LDS (the 8 extra columns are the right half of the tile's last strip) and writes it out twice, both times in whole
// contiguous runs -- the row-major planes 16 samples per thread (a row of the tile = one or two full cache lines), the strips 16
// samples per thread with consecutive lanes on consecutive rows (LT_ROWS x 16 samples of a strip = one contiguous run).  The planes
// are not read back, so a frame costs W*H read + 4*S + 8*S written and nothing else (the two-kernel form read the 4*S again, from
// HBM once a batch of frames exceeds the caches).  Columns at and beyond the padded width (< stride) are zero in the strips and
// untouched in the planes; no candidate reads them.
#ifndef LT_ROWS
#define LT_ROWS 32
#endif
#define LT_COLS 128
#define LT_PITCH 144 // samples per tile row in LDS: LT_COLS + 8, rounded up to keep every row 16-byte aligned
template <typename T>
__device__ __forceinline__ void lowres_px4( const T *__restrict__ src, int src_stride, int width, int height, int lw, int lh, int xb, int Y,
                                            T o0[4], T oh[4], T ov[4], T oc[4] )
{
    const int y = iclip3( Y - LA_PAD, 0, lh - 1 );
    const T *r0 = src + (size_t)imin2( 2 * y, height - 1 ) * src_stride;
    const T *r1 = src + (size_t)imin2( 2 * y + 1, height - 1 ) * src_stride;
    const T *r2 = src + (size_t)imin2( 2 * y + 2, height - 1 ) * src_stride;
    // A piece of four samples never straddles the picture's edge (the border is 32 wide, the picture a multiple of 8): a piece of the
    // left / right border repeats sample 0 of the first / sample 3 of the last piece of its row (plane_expand_border, frame.c:535-554).
    // (Until round 5 border pieces recomputed every sample from clamped source columns, 18 single-sample loads per piece, and a tile
    // that touches the border -- a third of them at 1080p -- ran that path for the whole wave.)
    const int xb_req = xb;
    xb = iclip3( xb, 0, lw - 4 );
    if( 2 * xb + 7 < width )
    {
        int a[9], b[9], c[9];
        T va[8], vb[8], vc[8];
        __builtin_memcpy( va, r0 + 2 * xb, 8 * sizeof( T ) );
        __builtin_memcpy( vb, r1 + 2 * xb, 8 * sizeof( T ) );
        __builtin_memcpy( vc, r2 + 2 * xb, 8 * sizeof( T ) );
#pragma unroll
        for( int i = 0; i < 8; i++ ) { a[i] = va[i]; b[i] = vb[i]; c[i] = vc[i]; }
        const int x8 = imin2( 2 * xb + 8, width - 1 ); // the column right of the picture is its last column again (mc.c:466-468)
        a[8] = r0[x8]; b[8] = r1[x8]; c[8] = r2[x8];
        int t[9], u[9];
#pragma unroll
        for( int i = 0; i < 9; i++ ) { t[i] = ( a[i] + b[i] + 1 ) >> 1; u[i] = ( b[i] + c[i] + 1 ) >> 1; }
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            o0[i] = (T)( ( t[2 * i] + t[2 * i + 1] + 1 ) >> 1 );
            oh[i] = (T)( ( t[2 * i + 1] + t[2 * i + 2] + 1 ) >> 1 );
            ov[i] = (T)( ( u[2 * i] + u[2 * i + 1] + 1 ) >> 1 );
            oc[i] = (T)( ( u[2 * i + 1] + u[2 * i + 2] + 1 ) >> 1 );
        }
    }
    else
    {
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            const int x = iclip3( xb + i, 0, lw - 1 );
            const int x0 = imin2( 2 * x, width - 1 ), x1 = imin2( 2 * x + 1, width - 1 ), x2 = imin2( 2 * x + 2, width - 1 );
            int t0 = ( r0[x0] + r1[x0] + 1 ) >> 1, t1 = ( r0[x1] + r1[x1] + 1 ) >> 1, t2 = ( r0[x2] + r1[x2] + 1 ) >> 1;
            int u0 = ( r1[x0] + r2[x0] + 1 ) >> 1, u1 = ( r1[x1] + r2[x1] + 1 ) >> 1, u2 = ( r1[x2] + r2[x2] + 1 ) >> 1;
            o0[i] = (T)( ( t0 + t1 + 1 ) >> 1 ); oh[i] = (T)( ( t1 + t2 + 1 ) >> 1 );
            ov[i] = (T)( ( u0 + u1 + 1 ) >> 1 ); oc[i] = (T)( ( u1 + u2 + 1 ) >> 1 );
        }
    }
    if( xb_req != xb )
    {
        const int k = xb_req < 0 ? 0 : 3;
        const T e0 = o0[k], eh = oh[k], ev = ov[k], ec = oc[k];
#pragma unroll
        for( int i = 0; i < 4; i++ ) { o0[i] = e0; oh[i] = eh; ov[i] = ev; oc[i] = ec; }
    }
}
// NPR: how many of the four planes are also written ROW-MAJOR.  The searches and the B cells read the strip copy; of the row-major planes
// only plane 0 has readers on the device (intra_kernel, weight_cost_kernel, the weighted strip copies): H, V and HV row-major are read by
// x264hip_get_lowres alone, which rebuilds them from the strips when asked (strips_to_plane_kernel).  NPR = 1: 9 S written per frame
// instead of 12 S (0.34 -> 0.28 ms per 160 frames of 1080p); NPR = 4 (X264HIP_ROWMAJOR=4): every plane at ingest, the round-5 form.
template <typename T, int NPR>
__global__ __launch_bounds__( 256 ) void lowres_tiles_kernel( const PutDesc *descs, PutDesc single, int width, int height,
                                                              int plane_elems, int stride, int lw, int lh )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.z ) : single;
    const T *__restrict__ src = (const T *)D.src;
    T *__restrict__ planes = (T *)D.planes;
    T *__restrict__ strips = planes + 4 * (size_t)plane_elems;
    const int pw = lw + 2 * LA_PAD, rows = lh + 2 * LA_PAD, n_strips = stride >> 3;
    const int X0 = blockIdx.x * LT_COLS, Y0 = blockIdx.y * LT_ROWS;
    __shared__ __attribute__( ( aligned( 16 ) ) ) T tile[4][LT_ROWS][LT_PITCH];
    const int tid = threadIdx.x;
    constexpr int Q = ( LT_COLS + 8 ) / 4; // 4-sample pieces per tile row
    for( int item = tid; item < LT_ROWS * Q; item += 256 )
    {
        const int r = item / Q, q = item - r * Q;
        const int Y = Y0 + r, X = X0 + 4 * q;
        T o[4][4];
        if( Y < rows && X < pw )
            lowres_px4<T>( src, D.src_stride, width, height, lw, lh, X - LA_PAD, Y, o[0], o[1], o[2], o[3] );
        else
#pragma unroll
            for( int p = 0; p < 4; p++ )
#pragma unroll
                for( int i = 0; i < 4; i++ ) o[p][i] = 0;
#pragma unroll
        for( int p = 0; p < 4; p++ )
            __builtin_memcpy( &tile[p][r][4 * q], o[p], 4 * sizeof( T ) );
    }
    __syncthreads();
#pragma unroll
    for( int rp = 0; rp < LT_ROWS / 32; rp++ )
    {
        // row-major planes: LT_ROWS rows x 8 pieces of 16 samples
        const int r = rp * 32 + ( tid >> 3 ), c = ( tid & 7 ) * 16;
        const int Y = Y0 + r, X = X0 + c;
        if( Y < rows )
        {
            const size_t o = (size_t)Y * stride + X;
#pragma unroll
            for( int p = 0; p < NPR; p++ )
            {
                T v[16];
                __builtin_memcpy( v, &tile[p][r][c], 16 * sizeof( T ) );
                if( X + 16 <= pw )
                    __builtin_memcpy( planes + (size_t)p * plane_elems + o, v, 16 * sizeof( T ) );
                else if( X + 8 <= pw ) // the padded width is a multiple of 8, not of 16
                    __builtin_memcpy( planes + (size_t)p * plane_elems + o, v, 8 * sizeof( T ) );
            }
        }
    }
    // strips X0 / 8 .. X0 / 8 + 15: a lane per row, 16 samples each
#pragma unroll
    for( int pass = 0; pass < LT_COLS / 8 * LT_ROWS / 256; pass++ )
    {
        const int s = pass * ( 256 / LT_ROWS ) + tid / LT_ROWS, r = tid % LT_ROWS;
        const int k = ( X0 >> 3 ) + s, Y = Y0 + r;
        if( k < n_strips && Y < rows )
        {
#pragma unroll
            for( int p = 0; p < 4; p++ )
            {
                T v[16];
                __builtin_memcpy( v, &tile[p][r][8 * s], 16 * sizeof( T ) );
                if( k + 1 >= n_strips )
#pragma unroll
                    for( int i = 8; i < 16; i++ ) v[i] = 0; // beyond the plane
                __builtin_memcpy( strips + 2 * (size_t)p * plane_elems + strip_layout::row_off( k, Y, rows ), v, 16 * sizeof( T ) );
            }
        }
    }
}

// row-major plane p of a frame rebuilt from its strip copy (x264hip_get_lowres for the H / V / HV planes): a thread per 8 samples
template <typename T>
__global__ __launch_bounds__( 256 ) void strips_to_plane_kernel( T *__restrict__ planes, int p, int plane_elems, int stride, int rows )
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // piece index: row Y, strip k
    const int n_strips = stride >> 3;
    if( i >= rows * n_strips )
        return;
    const int Y = i / n_strips, k = i - Y * n_strips;
    const T *strips = planes + 4 * (size_t)plane_elems;
    T v[8];
    __builtin_memcpy( v, strips + 2 * (size_t)p * plane_elems + strip_layout::row_off( k, Y, rows ), 8 * sizeof( T ) );
    __builtin_memcpy( planes + (size_t)p * plane_elems + (size_t)Y * stride + 8 * k, v, 8 * sizeof( T ) );
}

// plain x264_mc_functions_t.frame_init_lowres_core signature (mc.h:326-327): no borders, caller's layout
template <typename T>
__global__ __launch_bounds__( 256 ) void lowres_core_kernel( const T *__restrict__ src, T *d0, T *dh, T *dv, T *dc,
                                                             long src_stride, long dst_stride, int w, int h )
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if( x >= w )
        return;
    const T *r0 = src + 2 * y * src_stride, *r1 = r0 + src_stride, *r2 = r1 + src_stride;
    int t0 = ( r0[2 * x] + r1[2 * x] + 1 ) >> 1, t1 = ( r0[2 * x + 1] + r1[2 * x + 1] + 1 ) >> 1, t2 = ( r0[2 * x + 2] + r1[2 * x + 2] + 1 ) >> 1;
    int u0 = ( r1[2 * x] + r2[2 * x] + 1 ) >> 1, u1 = ( r1[2 * x + 1] + r2[2 * x + 1] + 1 ) >> 1, u2 = ( r1[2 * x + 2] + r2[2 * x + 2] + 1 ) >> 1;
    d0[y * dst_stride + x] = (T)( ( t0 + t1 + 1 ) >> 1 );
    dh[y * dst_stride + x] = (T)( ( t1 + t2 + 1 ) >> 1 );
    dv[y * dst_stride + x] = (T)( ( u0 + u1 + 1 ) >> 1 );
    dc[y * dst_stride + x] = (T)( ( u1 + u2 + 1 ) >> 1 );
}

// ---- adaptive-quant statistics (encoder/ratecontrol.c:225-415, aq-mode 0..3) ---------------------------
// One wave per 16x16 macroblock: luma sum / sum of squares (+ optional 8x8 chroma), energy -> Q8 inverse
// qscale through the reference's log2 / exp2 look-up tables; frame totals accumulate with 64-bit atomics.
struct AqLuts
{
    float log2_lut[128];
    unsigned char exp2_lut[64];
};

// sum over the wave, the same (wave-uniform) value in every lane: DPP within the 16-lane rows, then the four row totals as scalars
__device__ __forceinline__ unsigned wave_sum_u32( unsigned v )
{
    const int r = reduce16( (int)v );
    return (unsigned)( __builtin_amdgcn_readlane( r, 0 ) + __builtin_amdgcn_readlane( r, 16 ) + __builtin_amdgcn_readlane( r, 32 ) + __builtin_amdgcn_readlane( r, 48 ) );
}

template <typename T>
#define AQ_MBS_PER_WG 8
__global__ __launch_bounds__( 64 ) void aq_kernel( const PutDesc *descs, PutDesc single, int width, int height, int mb_w, int mb_h,
                                                   float strength, float log2_bias, const AqLuts *luts, int aq_mode, float depth_corr, int chroma_format )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.z ) : single;
    const T *__restrict__ luma = (const T *)D.src, *__restrict__ cb = (const T *)D.cb, *__restrict__ cr = (const T *)D.cr;
    const int stride = D.src_stride, cstride = D.cstride, aq_on = D.aq_on;
    uint16_t *inv_qscale = D.inv_qscale;
    uint2 *mb_sums = D.mb_sums;
    float *qp_offset_aq = D.qp_aq, *qp_offset = D.qp;
    // XCD-aware placement: workgroup b runs on XCD b % 8 (observed dispatch order, used for locality only), so the macroblocks are
    // dealt to the XCDs in eight contiguous runs -- the 128-byte lines that horizontally adjacent macroblocks share are then fetched
    // into ONE L2 instead of eight (the launch is 1-D, a multiple of 8 workgroups per frame)
    // one-wave workgroups, AQ_MBS_PER_WG macroblocks each, in three phases so that nothing waits for memory one macroblock at a
    // time: (1) the pixel loads of all macroblocks, (2) their wave sums, (3) the per-macroblock finish with one LANE per
    // macroblock (table look-ups and stores of the eight macroblocks in parallel instead of eight times on lane 0)
    const int lane = lane_id();
    const int wg = (int)( blockIdx.x & 7 ) * (int)( gridDim.x >> 3 ) + (int)( blockIdx.x >> 3 );
    const int ly = lane >> 2, lx = ( lane & 3 ) * 4;
    const int first = wg * AQ_MBS_PER_WG, n_here = imin2( AQ_MBS_PER_WG, mb_w * mb_h - first );
    if( n_here <= 0 )
        return; // wave-uniform
    __shared__ unsigned sh_s[AQ_MBS_PER_WG], sh_q[AQ_MBS_PER_WG], sh_e[AQ_MBS_PER_WG];
    unsigned ps[AQ_MBS_PER_WG], pq[AQ_MBS_PER_WG];
#pragma unroll
    for( int it = 0; it < AQ_MBS_PER_WG; it++ )
    {
        const int logical = first + imin2( it, n_here - 1 );
        const int mx = logical % mb_w, my = logical / mb_w;
        const T *row = luma + (size_t)imin2( 16 * my + ly, height - 1 ) * stride;
        unsigned s = 0, q = 0;
        T px[4];
        if( 16 * mx + 15 < width ) // wave-uniform: the macroblock lies inside the picture, one 4-sample load per lane
            __builtin_memcpy( px, row + 16 * mx + lx, 4 * sizeof( T ) );
        else
        {
#pragma unroll
            for( int i = 0; i < 4; i++ )
                px[i] = row[imin2( 16 * mx + lx + i, width - 1 )];
        }
#pragma unroll
        for( int i = 0; i < 4; i++ )
        {
            const unsigned v = px[i];
            s += v; q += v * v;
        }
        ps[it] = s; pq[it] = q;
    }
#pragma unroll
    for( int it = 0; it < AQ_MBS_PER_WG; it++ )
    {
        const unsigned s = wave_sum_u32( ps[it] ), q = wave_sum_u32( pq[it] );
        if( lane == 0 )
        {
            sh_s[it] = s; sh_q[it] = q;
            sh_e[it] = q - (unsigned)( ( (unsigned long long)s * s ) >> 8 );
        }
    }
    if( cb )
    {
        // chroma part of ac_energy_mb (ratecontrol.c:238-296): per plane an 8x8 block with shift 6 (4:2:0), 8x16 with shift 7
        // (4:2:2) or 16x16 with shift 8 (4:4:4); coordinates clamp like the mod-16 border replication
        const int c444 = chroma_format == 3, c420 = chroma_format < 2;
        const int cw = c444 ? width : ( width + 1 ) >> 1, ch = c420 ? ( height + 1 ) >> 1 : height;
        const int lbw = c444 ? 4 : 3, bw = 1 << lbw, bh = c420 ? 8 : 16, n = ( bw * bh ) >> 6, shift = c444 ? 8 : c420 ? 6 : 7;
        for( int it = 0; it < n_here; it++ )
        {
            const int logical = first + it;
            const int mx = logical % mb_w, my = logical / mb_w;
            unsigned sb = 0, qb = 0, sr = 0, qr = 0;
            for( int k = 0; k < n; k++ )
            {
                const int idx = lane + 64 * k;
                const int cy = imin2( bh * my + ( idx >> lbw ), ch - 1 ), cx = imin2( bw * mx + ( idx & ( bw - 1 ) ), cw - 1 );
                const unsigned vb = cb[(size_t)cy * cstride + cx], vr = cr[(size_t)cy * cstride + cx];
                sb += vb; qb += vb * vb; sr += vr; qr += vr * vr;
            }
            sb = wave_sum_u32( sb ); qb = wave_sum_u32( qb ); sr = wave_sum_u32( sr ); qr = wave_sum_u32( qr );
            if( lane == 0 )
                sh_e[it] += qb - (unsigned)( ( (unsigned long long)sb * sb ) >> shift ) + qr - (unsigned)( ( (unsigned long long)sr * sr ) >> shift );
        }
    }
    __syncthreads(); // one wave: orders the LDS words
    if( lane < n_here )
    {
        const int xy = first + lane; // raster index = my * mb_w + mx
        const unsigned energy = sh_e[lane];
        mb_sums[xy] = make_uint2( sh_s[lane], sh_q[lane] );
        int out = 256;
        float qp_adj = 0.f;
        if( aq_on && aq_mode >= 2 )
        {
            // first pass of the auto-variance modes (ratecontrol.c:354-371) in the reference build's arithmetic (gcc -O3
            // -ffast-math): powf( x, 0.125f ) is three correctly rounded square roots, and the "qp_adj * qp_adj" that is
            // averaged is the intermediate fourth root.  aq_auto_kernel turns the two roots into the final values.
            float x = __uint2float_rn( energy );
            if( depth_corr != 1.f )
                x = __fmul_rn( x, depth_corr );
            x = __fadd_rn( x, 1.f );
            const float r4 = sqrtf( sqrtf( x ) );
            qp_offset_aq[xy] = r4;
            qp_offset[xy] = sqrtf( r4 ); // sqrtf: correctly rounded (the __fsqrt_rn intrinsic is the 1-ulp native one)
            inv_qscale[xy] = 256;
            return;
        }
        if( aq_on )
        {
            unsigned e = energy > 1 ? energy : 1;
            int lz = __clz( e );
            // association of the reference build (gcc -ffast-math): lut[mantissa] + ( lz_part - bias )
            float l2 = __fadd_rn( luts->log2_lut[( e << lz >> 24 ) & 0x7f], __fsub_rn( (float)( 31 - lz ), log2_bias ) );
            qp_adj = __fmul_rn( strength, l2 );
            int i = (int)__fadd_rn( __fmul_rn( qp_adj, -64.f / 6.f ), 512.5f );
            out = i < 0 ? 0 : i > 1023 ? 0xffff : ( ( luts->exp2_lut[i & 63] + 256 ) << ( i >> 6 ) >> 8 );
        }
        inv_qscale[xy] = (uint16_t)out;
        qp_offset_aq[xy] = qp_adj; // f_qp_offset_aq = f_qp_offset = qp_adj (ratecontrol.c:392-396)
        qp_offset[xy] = qp_adj;
    }
}

// Second pass of aq-mode 2 / 3 (ratecontrol.c:372-398) for one frame per workgroup.  The two averages are SEQUENTIAL
// FP32 sums in raster order in the reference (the rounding of every step counts), so one thread adds them, out of LDS
// tiles the whole workgroup stages; everything after the averages is per macroblock and runs on all threads.
#define AQ_AUTO_TILE 4096
__global__ __launch_bounds__( 1024 ) void aq_auto_kernel( const PutDesc *descs, PutDesc single, int n_mb, int aq_mode, float aq_strength,
                                                          const AqLuts *luts )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.x ) : single;
    if( !D.aq_on )
        return;
    __shared__ float t4[AQ_AUTO_TILE], t8[AQ_AUTO_TILE];
    __shared__ float par[2];
    float s4 = 0.f, s8 = 0.f;
    for( int base = 0; base < n_mb; base += AQ_AUTO_TILE )
    {
        const int cnt = imin2( AQ_AUTO_TILE, n_mb - base );
        for( int i = threadIdx.x; i < cnt; i += blockDim.x )
        {
            t4[i] = D.qp_aq[base + i];
            t8[i] = D.qp[base + i];
        }
        __syncthreads();
        if( threadIdx.x == 0 )
            for( int i = 0; i < cnt; i++ )
            {
                s4 = __fadd_rn( s4, t4[i] );
                s8 = __fadd_rn( s8, t8[i] );
            }
        __syncthreads();
    }
    if( threadIdx.x == 0 )
    {
        const float cnt = (float)n_mb;
        const float avg_pow2 = __fdiv_rn( s4, cnt ), avg = __fdiv_rn( s8, cnt );
        par[0] = __fmul_rn( aq_strength, avg );                                                                 // strength
        par[1] = __fadd_rn( __fdiv_rn( __fmul_rn( __fsub_rn( 14.f, avg_pow2 ), 0.5f ), avg ), avg );            // avg_adj
    }
    __syncthreads();
    const float str = par[0], avg_adj = par[1];
    for( int i = threadIdx.x; i < n_mb; i += blockDim.x )
    {
        const float q = D.qp[i];
        float qp_adj = __fmul_rn( __fsub_rn( q, avg_adj ), str );
        if( aq_mode == 3 )
            qp_adj = __fadd_rn( __fmul_rn( __fsub_rn( 1.f, __fdiv_rn( 14.f, __fmul_rn( q, q ) ) ), aq_strength ), qp_adj );
        const int k = (int)__fadd_rn( __fmul_rn( qp_adj, -64.f / 6.f ), 512.5f );
        D.inv_qscale[i] = (uint16_t)( k < 0 ? 0 : k > 1023 ? 0xffff : ( ( luts->exp2_lut[k & 63] + 256 ) << ( k >> 6 ) >> 8 ) );
        D.qp_aq[i] = qp_adj;
        D.qp[i] = qp_adj;
    }
}

// frame totals of the per-MB sums: one workgroup, no atomics
__global__ __launch_bounds__( 1024 ) void aq_reduce_kernel( const PutDesc *descs, PutDesc single, int n )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.x ) : single;
    const uint2 *__restrict__ mb_sums = D.mb_sums;
    unsigned long long *frame_sums = D.frame_sums;
    __shared__ unsigned long long sh[2][16];
    unsigned long long s = 0, q = 0;
    for( int i = threadIdx.x; i < n; i += blockDim.x )
    {
        uint2 v = mb_sums[i];
        s += v.x; q += v.y;
    }
#pragma unroll
    for( int o = 32; o > 0; o >>= 1 )
    {
        s += __shfl_xor( s, o );
        q += __shfl_xor( q, o );
    }
    if( ( threadIdx.x & 63 ) == 0 ) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    if( threadIdx.x == 0 )
    {
        unsigned long long ts = 0, tq = 0;
        for( int i = 0; i < (int)( blockDim.x >> 6 ); i++ ) { ts += sh[0][i]; tq += sh[1][i]; }
        frame_sums[0] = ts; frame_sums[1] = tq;
    }
}

// ---- lowres intra cost (encoder/slicetype.c:714-757; predictors common/predict.c:221-308,632-884) ----
// Four 8x8 blocks per wave (one per 16-lane group), one prediction mode per pass.  The 17+8 neighbours of a block and their
// low-pass filtered versions sit in LDS; every lane derives its 4 predicted pixels directly from the H.264 per-pixel formulas.
struct IntraEdges
{
    int top[18];  // top[i+1] = p[i,-1], i = -1..15 (top[0] is the corner); top[17] pad
    int left[8];  // p[-1,y]
    int ft[18];   // filtered: ft[i+1] = p'[i,-1], ft[0] = p'[-1,-1]
    int fl[8];    // filtered left p'[-1,y]
    int e[28];    // the filtered edge as ONE line, bottom of the left column -> corner -> top row: e[7-i] = p'[-1,i], e[8] = p'[-1,-1],
                  // e[9+i] = p'[i,-1] (i = 0..15).  A directional predictor is then three neighbouring entries whatever side of the
                  // diagonal the pixel is on (the per-pixel case distinctions of predict.c:700-884 become index arithmetic)
};

__device__ __forceinline__ int f3( int a, int b, int c ) { return ( a + 2 * b + c + 2 ) >> 2; }
__device__ __forceinline__ int f2( int a, int b ) { return ( a + b + 1 ) >> 1; }

__device__ __forceinline__ int intra_pred_px( const IntraEdges &E, int mode, int x, int y, int pixel_max )
{
#define TT( i ) E.ft[( i ) + 1]
#define LL( i ) ( ( i ) < 0 ? E.ft[0] : E.fl[( i )] )
    switch( mode )
    {
        case 0: // 8x8 chroma-style DC, four quadrants
        {
            int t0 = E.top[1] + E.top[2] + E.top[3] + E.top[4], t1 = E.top[5] + E.top[6] + E.top[7] + E.top[8];
            int l0 = E.left[0] + E.left[1] + E.left[2] + E.left[3], l1 = E.left[4] + E.left[5] + E.left[6] + E.left[7];
            return y < 4 ? ( x < 4 ? ( t0 + l0 + 4 ) >> 3 : ( t1 + 2 ) >> 2 ) : ( x < 4 ? ( l1 + 2 ) >> 2 : ( t1 + l1 + 4 ) >> 3 );
        }
        case 1: return E.left[y];
        case 2: return E.top[x + 1];
        case 3: // plane
        {
            int H = 0, V = 0;
#pragma unroll
            for( int i = 1; i <= 4; i++ )
            {
                H += i * ( E.top[3 + i + 1] - E.top[3 - i + 1] );
                V += i * ( E.left[3 + i] - ( 3 - i < 0 ? E.top[0] : E.left[3 - i] ) );
            }
            int a = 16 * ( E.left[7] + E.top[8] ), b = ( 17 * H + 16 ) >> 5, c = ( 17 * V + 16 ) >> 5;
            return iclip3( ( a + b * ( x - 3 ) + c * ( y - 3 ) + 16 ) >> 5, 0, pixel_max );
        }
        case 4: // diagonal down-left: the last pixel's third tap repeats p'[15,-1], which is the reference's ( t14 + 3 t15 + 2 ) >> 2
            return f3( E.e[9 + x + y], E.e[10 + x + y], E.e[9 + imin2( x + y + 2, 15 )] );
        case 5: // diagonal down-right
            return f3( E.e[7 + x - y], E.e[8 + x - y], E.e[9 + x - y] );
        case 6: // vertical right
        {
            const int z = 2 * x - y, base = z >= -1 ? 7 + x - ( y >> 1 ) : 8 + z;
            const int e0 = E.e[base], e1 = E.e[base + 1], e2 = E.e[base + 2];
            return ( z >= 0 && !( z & 1 ) ) ? f2( e1, e2 ) : f3( e0, e1, e2 );
        }
        case 7: // horizontal down
        {
            const int z = 2 * y - x, base = z >= -1 ? 7 - y + ( x >> 1 ) : 6 - z;
            const int e0 = E.e[base], e1 = E.e[base + 1], e2 = E.e[base + 2];
            return ( z >= 0 && !( z & 1 ) ) ? f2( e0, e1 ) : f3( e0, e1, e2 );
        }
        case 8: // vertical left
        {
            const int k = 9 + x + ( y >> 1 );
            const int e0 = E.e[k], e1 = E.e[k + 1], e2 = E.e[k + 2];
            return ( y & 1 ) ? f3( e0, e1, e2 ) : f2( e0, e1 );
        }
        default: // 9: horizontal up: beyond the last left sample the edge repeats it (the reference's special cases for z >= 13)
        {
            const int k = y + ( x >> 1 );
            const int e0 = E.e[7 - imin2( k, 7 )], e1 = E.e[7 - imin2( k + 1, 7 )], e2 = E.e[7 - imin2( k + 2, 7 )];
            return ( ( x + 2 * y ) & 1 ) ? f3( e0, e1, e2 ) : f2( e0, e1 );
        }
    }
#undef TT
#undef LL
}

// One-wave workgroups, every wave walks INTRA_BLOCKS_PER_WG consecutive blocks, FOUR AT A TIME: a 16-lane group owns one block (the
// Px4 geometry: a lane holds 4 pixels of a row) and all four groups evaluate the SAME prediction mode in a pass, so the mode switch
// is uniform across the wave.  (One block per wave with four modes side by side ran the four case bodies of a pass one after the
// other at a quarter of the lanes each: ten serial case bodies per block, against ten per FOUR blocks here.)  The workgroups stay
// one wave wide on purpose: beside the search kernel, whose waves fill the register files, a single free wave slot is all such a
// workgroup needs (four-wave workgroups measured 5 % slower end to end with eight contexts in flight).  A workgroup per block made
// the launch dispatch-bound (1.3 M workgroups for 160 frames of 1080p).
// MODES (3: DC / H / V, subme <= 1; 10: all) is a template parameter and the mode loop is unrolled: every pass's case body is then
// straight-line code and the LDS reads of the next mode are issued under the transform of the current one (INTRA_UNROLL=0: the loop as
// a loop, one uniform switch per pass -- round 4's form, for A/B runs: 0.526 ms against 0.336 ms for 160 frames of 1080p,
// scripts/r05_intra.sh).
#ifndef INTRA_UNROLL
#define INTRA_UNROLL 1
#endif
#define INTRA_BLOCKS_PER_WG 8
template <typename T, int MODES>
__global__ __launch_bounds__( 64 ) void intra_kernel( LaP P, const PutDesc *descs, PutDesc single )
{
    const PutDesc D = descs ? load_uniform( descs + blockIdx.z ) : single;
    const T *__restrict__ fenc0 = (const T *)D.planes + LA_PAD * P.stride + LA_PAD;
    uint16_t *intra_cost = D.intra_cost;
    __shared__ IntraEdges E4[4];
    const int lane = lane_id();
    // XCD-aware placement as in aq_kernel: each XCD gets a contiguous run of workgroups (INTRA_BLOCKS_PER_WG consecutive blocks each)
    const int n_blocks = P.mb_w * P.mb_h;
    const int wg = (int)( blockIdx.x & 7 ) * (int)( gridDim.x >> 3 ) + (int)( blockIdx.x >> 3 );
    const int g = lane >> 4, l = lane & 15, q = l >> 2;
    const int tx = ( q & 1 ) * 4, row = ( q >> 1 ) * 4 + ( l & 3 );
    IntraEdges &E = E4[g];
    for( int it = 0; it < INTRA_BLOCKS_PER_WG; it += 4 )
    {
        const int logical = wg * INTRA_BLOCKS_PER_WG + it + g;
        const bool live = logical < n_blocks;
        const int lg = live ? logical : n_blocks - 1; // dead groups redo the last block and write nothing
        const int bx = lg % P.mb_w, by = lg / P.mb_w;
        const T *src = fenc0 + 8 * ( by * P.stride + bx );
        // the 17 + 8 neighbours of the group's block: lane l fetches top[l] and, lanes 0..8, top[16] / left[0..7]
        E.top[l] = src[-P.stride + l - 1];
        if( l == 8 )
            E.top[16] = src[-P.stride + 15];
        else if( l < 8 )
            E.left[l] = src[l * P.stride - 1];
        const Px4 f = load_px4( src + row * P.stride + tx );
        __syncthreads();
        {
            // ft[l] = p'[l-1,-1]: corner, t0..t15 (predict.c:632-675 with all neighbours available); lanes 0..8 also ft[16] / fl[0..7]
            const int i = l - 1;
            const int ftl = i < 0 ? f3( E.top[1], E.top[0], E.left[0] ) : f3( E.top[i], E.top[i + 1], E.top[i + 2] );
            E.ft[l] = ftl; E.e[8 + l] = ftl;
            if( l == 8 )
            {
                const int v = ( E.top[15] + 3 * E.top[16] + 2 ) >> 2;
                E.ft[16] = v; E.e[24] = v;
            }
            else if( l < 8 )
            {
                const int v = l == 7 ? ( E.left[6] + 3 * E.left[7] + 2 ) >> 2 : f3( l == 0 ? E.top[0] : E.left[l - 1], E.left[l], E.left[l + 1] );
                E.fl[l] = v; E.e[7 - l] = v;
            }
        }
        __syncthreads();
        int best = COST_MAX_I;
#if INTRA_UNROLL
#pragma unroll
#else
#pragma nounroll
#endif
        for( int mode = 0; mode < MODES; mode++ )
        {
            int pr[4];
#pragma unroll
            for( int i = 0; i < 4; i++ )
                pr[i] = intra_pred_px( E, mode, tx + i, row, P.pixel_max );
            const Px4 r = px4_from_ints( pr, false );
            const int v = P.mbcmp_satd ? reduce16( satd_partial_px4( f, r ) ) >> 1 : reduce16( sad_partial16( f, r ) );
            best = imin2( best, v );
        }
        if( l == 0 && live )
            intra_cost[by * P.mb_w + bx] = (uint16_t)( ( ( best + 5 * P.lambda ) >> P.depth_shift ) + 4 );
        __syncthreads(); // the edges are rewritten by the next trip
    }
}

// ---- explicit weights: weighted copy of padded plane 0 (slicetype.c:490-500, mc.c:117-160), in the strip layout of me_search.h
template <typename T>
__global__ __launch_bounds__( 256 ) void weight_strips_kernel( const T *__restrict__ src, T *__restrict__ strips, int n, int stride, WtD w, int pixel_max )
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const int Y = i / stride, c = i - Y * stride, k = c >> 3, rows = n / stride;
    const T v = (T)weight_px( src[i], w, pixel_max );
    strips[strip_layout::row_off( k, Y, rows ) + ( c & 7 )] = v;         // left half of strip k ...
    if( k )
        strips[strip_layout::row_off( k - 1, Y, rows ) + 8 + ( c & 7 )] = v; // ... and right half of strip k - 1
}

// The same for every weighted search of a launch at once (blockIdx.y = search): source plane, destination and weight come out of the
// launch's own search descriptors (me_search.h SearchDesc: ref_strips is the strip copy behind the reference's four row-major planes,
// refw_strips where the weighted copy goes).  A launch per weighted search -- 45 small kernels in a row for a fade -- cost each of them
// the queueing of a kernel beside seven other contexts (156 us on average against 3 us alone, profiles/r06_bench_kernel_stats.csv).
template <typename T, typename D>
__global__ __launch_bounds__( 256 ) void weight_strips_multi_kernel( const D *__restrict__ descs, int n, int stride, int pixel_max )
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n )
        return;
    const D d = load_uniform( descs + blockIdx.y );
    const T *src = d.ref_strips - 4 * (size_t)n; // plane 0, row-major: four planes of n samples, then their strip copies
    T *strips = const_cast<T *>( d.refw_strips );
    const int Y = i / stride, c = i - Y * stride, k = c >> 3, rows = n / stride;
    const T v = (T)weight_px( src[i], d.wt, pixel_max );
    strips[strip_layout::row_off( k, Y, rows ) + ( c & 7 )] = v;
    if( k )
        strips[strip_layout::row_off( k - 1, Y, rows ) + 8 + ( c & 7 )] = v;
}

// weight_cost_luma (slicetype.c:191-222): sum over blocks of min( mbcmp, intra_cost ).  256-thread workgroups, four
// blocks per wave, sixteen per workgroup.  blockIdx.y picks the (fenc, ref, weight) job of a batch, blockIdx.z the
// unweighted (0) or weighted (1) sum of a pair.  The workgroup that arrives last publishes the total to pinned host
// memory and re-arms the two device counters of its job.
struct WeightJob
{
    const void *fenc0, *ref0;     // plane-0 origins
    const uint16_t *intra_cost;
    WtD w;
    unsigned *accum;              // device [2][2]: { running sum, arrivals } per z
    unsigned *out_host;           // pinned [2]
};

#define WCOST_BLOCKS_PER_WG 128 // 4 waves x 4 blocks x 8 passes
template <typename T>
__global__ __launch_bounds__( 256 ) void weight_cost_kernel( LaP P, const WeightJob *jobs, WeightJob single, int mode /* 0 unweighted, 1 weighted, 2 both */ )
{
    __shared__ unsigned part[2][4];
    const WeightJob J = jobs ? load_uniform( jobs + blockIdx.y ) : single;
    const T *__restrict__ fenc0 = (const T *)J.fenc0, *__restrict__ ref0 = (const T *)J.ref0;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int g = lane >> 4, l = lane & 15, q = l >> 2;
    const int tx = ( q & 1 ) * 4, row = ( q >> 1 ) * 4 + ( l & 3 );
    const int n_mb = P.mb_w * P.mb_h;
    unsigned tot[2] = { 0, 0 };
    // a workgroup covers WCOST_BLOCKS_PER_WG consecutive blocks (few workgroups per sum = few same-address atomics)
#pragma unroll 2
    for( int pass = 0; pass < WCOST_BLOCKS_PER_WG / 16; pass++ )
    {
        const int first = blockIdx.x * WCOST_BLOCKS_PER_WG + pass * 16 + wave * 4;
        if( first >= n_mb )
            break;
        const int xyc = imin2( first + g, n_mb - 1 );
        const int bx = xyc % P.mb_w, by = xyc / P.mb_w;
        const int off = 8 * ( by * P.stride + bx ) + row * P.stride + tx;
        const Px4 f = load_px4( fenc0 + off );
        const Px4 r = load_px4( ref0 + off );
        // the intra costs as the reference reads them here: after the 14-bit clamp of the [0][0] map (slicetype.c:712)
        // (a block the evaluations never visit keeps the 0xFFFF of its allocation, frame.c:288-289)
        const int icost = la_visited( P, bx, by ) ? imin2( (int)J.intra_cost[xyc], 0x3FFF ) : 0xFFFF;
        // one pass over the pixels serves both sums of a pair
#pragma unroll
        for( int z = 0; z < 2; z++ )
        {
            if( mode != 2 && mode != z )
                continue;
            const Px4 rz = z && J.w.on ? weight_px4<T>( r, J.w, P.pixel_max ) : r;
            const int c = imin2( block_cost8x8<T>( f, rz, P.mbcmp_satd ), icost );
#pragma unroll
            for( int k = 0; k < 4; k++ )
                if( first + k < n_mb )
                    tot[z] += (unsigned)__builtin_amdgcn_readlane( c, 16 * k );
        }
    }
    if( lane == 0 ) { part[0][wave] = tot[0]; part[1][wave] = tot[1]; }
    __syncthreads();
    if( threadIdx.x < 2 && ( mode == 2 || mode == (int)threadIdx.x ) )
    {
        const int z = threadIdx.x;
        unsigned *accum = J.accum + 2 * z;
        atomicAdd( &accum[0], part[z][0] + part[z][1] + part[z][2] + part[z][3] );
        __threadfence();
        if( atomicAdd( &accum[1], 1u ) == gridDim.x - 1 )
        {
            __threadfence();
            const unsigned total = atomicExch( &accum[0], 0u );
            atomicExch( &accum[1], 0u );
            __hip_atomic_store( J.out_host + z, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
        }
    }
}

// ---- mode selection and reductions (slicetype.c:616-652,708-712,758-790,946-985) ----------------------
struct CellArgs
{
    int b_bidir, dist_scale_factor, with_intra, is_intra_only, ref1_l0_valid;
    const unsigned long long *mvq0, *mvq1, *ref1_l0; // granule arrays (low 32 bits = packed mv)
    const int *costs0, *costs1;
    const uint16_t *intra_cost;   // may alias lowres_costs for the intra-only cell (frame.c:283)
    const uint16_t *inv_qscale;
    uint16_t *lowres_costs;
    int *row_satds, *row_satds_intra;
    int *acc;                     // [5]: cost_est, cost_est_aq, intra_mbs, intra_cost_est, intra_cost_est_aq
    int *blk;                     // [n_mb] scratch: final block cost | b_intra << 30, input of cell_reduce_kernel
    const void *fenc0, *ref0_0, *ref1_0; // B cells: strip copies (me_search.h) of the source frame's plane 0 and of the two references' planes
    int sums_only;                // reduce only: intra sums of a frame, no maps written (speculative [0][0] sums)
    int pad_;
    int *acc_dev;                 // device copy of acc[0..4] (what x264hip_export_cells packs for another rank)
    int *work;                    // [8] zero between launches: partial sums [0..4] and the arrival counter [5] of cell_reduce_kernel's workgroups
    // B cells evaluated BOTH ways in one pass (dual != 0, ref1_l0_valid set): the outcome without the list-1 reference's own vectors goes here
    // (the cell's spare storage); its sums are reduced through a descriptor of its own
    uint16_t *lowres_costs2;
    int *blk2;
    int dual, pad2_;
};

__device__ __forceinline__ void cell_finish( const LaP &P, const CellArgs &A, int xy, int bcost, int list_used, bool second = false )
{
    // executed by ONE lane per block: final cost of the block, its map entry, and the word the reduction reads
    const int icost = A.intra_cost[xy];
    int b_intra = 0;
    bcost = ( bcost >> P.depth_shift ) + 4;
    if( !A.b_bidir )
    {
        b_intra = icost < bcost;
        if( b_intra ) { bcost = icost; list_used = 0; }
    }
    ( second ? A.blk2 : A.blk )[xy] = bcost | ( b_intra << 30 );
    ( second ? A.lowres_costs2 : A.lowres_costs )[xy] = (uint16_t)( imin2( bcost, 0x3FFF ) + ( list_used << 14 ) );
}

// Row and frame sums of one evaluation (slicetype.c:746-757,778-788,946-985): gridDim.y workgroups per cell, each owning a contiguous
// band of block rows (a wave owns whole rows: the row sums need no atomics); the five frame sums are integer sums, so the order of the
// workgroups' contributions does not matter: each adds its part to the cell's work words, and the last one to arrive publishes the totals
// (pinned host record + device copy) and leaves the work words zero for the next use.  (One workgroup per cell -- the round 1-3 form --
// took 80 us for a 4K cell whatever else the chip was doing: a cell evaluated on demand waited for it.)
__global__ __launch_bounds__( 256 ) void cell_reduce_kernel( LaP P, const CellArgs *descs, CellArgs single )
{
    // (a sixth sum beside the five of the reference: how many intra costs of the frame exceed the 14 bits a map entry holds -- zero for
    // nearly every frame, and then the clamp an intra-only evaluation applies to the map, slicetype.c:790, is the identity and the host
    // neither launches it nor orders the queued MB-tree lists in front of it: x264hip.hip, frame_cost_t)
    __shared__ int sh[6][4];
    __shared__ int last;
    const CellArgs A = descs ? load_uniform( descs + blockIdx.x ) : single;
    const int W = P.mb_w, H = P.mb_h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    const int row_begin = (int)( (long long)H * blockIdx.y / gridDim.y ), row_end = (int)( (long long)H * ( blockIdx.y + 1 ) / gridDim.y );
    int t[6] = { 0, 0, 0, 0, 0, 0 };
    for( int by = row_begin + wave; by < row_end; by += n_waves )
    {
        int row = 0, row_i = 0;
        for( int bx = lane; bx < W; bx += 64 )
        {
            const int xy = by * W + bx;
            if( A.sums_only )
                t[5] += A.intra_cost[xy] > 0x3FFF;
            if( !la_visited( P, bx, by ) )
                continue; // contributes to no sum (slicetype.c:825-833)
            const bool scored = ( bx > 0 && bx < W - 1 && by > 0 && by < H - 1 ) || W <= 2 || H <= 2;
            const int inv = P.aq_mode ? A.inv_qscale[xy] : 256;
            const int w = A.sums_only ? 0 : A.blk[xy];
            const int bcost = w & 0x3FFFFFFF, b_intra = w >> 30;
            if( A.sums_only )
            {
                const int icost = A.intra_cost[xy];
                const int icost_aq = P.aq_mode ? ( icost * inv + 128 ) >> 8 : icost;
                row_i += icost_aq;
                if( scored ) { t[3] += icost; t[4] += icost_aq; }
                continue;
            }
            if( A.with_intra )
            {
                // for the intra-only cell the map aliases the intra costs and may just have been clamped: the
                // unclamped value is the block word there
                const int icost = A.is_intra_only ? bcost : A.intra_cost[xy];
                const int icost_aq = P.aq_mode ? ( icost * inv + 128 ) >> 8 : icost;
                row_i += icost_aq;
                if( scored ) { t[3] += icost; t[4] += icost_aq; }
            }
            if( !A.b_bidir && scored )
                t[2] += b_intra;
            if( !A.is_intra_only )
            {
                const int bcost_aq = P.aq_mode ? ( bcost * inv + 128 ) >> 8 : bcost;
                row += bcost_aq;
                if( scored ) { t[0] += bcost; t[1] += bcost_aq; }
            }
        }
        row = (int)wave_sum_u32( (unsigned)row ); row_i = (int)wave_sum_u32( (unsigned)row_i );
        if( lane == 0 )
        {
            if( !A.is_intra_only && !A.sums_only ) A.row_satds[by] = row;
            if( A.with_intra || A.sums_only ) A.row_satds_intra[by] = row_i;
        }
    }
#pragma unroll
    for( int k = 0; k < 6; k++ )
    {
        t[k] = (int)wave_sum_u32( (unsigned)t[k] );
        if( lane == 0 ) sh[k][wave] = t[k];
    }
    __syncthreads();
    const int slot = threadIdx.x < 5 ? threadIdx.x : 6; // work[5] is the arrival counter: the sixth sum gathers in work[6]
    if( threadIdx.x < 6 )
    {
        int v = 0;
        for( int i = 0; i < n_waves; i++ ) v += sh[threadIdx.x][i];
        if( gridDim.y == 1 )
        {
            A.acc[threadIdx.x] = v;
            A.acc_dev[threadIdx.x] = v;
        }
        else if( v )
            atomicAdd( &A.work[slot], v );
    }
    if( gridDim.y == 1 )
        return;
    __threadfence();
    __syncthreads();
    if( threadIdx.x == 0 )
        last = atomicAdd( &A.work[5], 1 ) == (int)gridDim.y - 1;
    __syncthreads();
    if( last && threadIdx.x < 6 )
    {
        __threadfence();
        const int v = atomicExch( &A.work[slot], 0 );
        A.acc[threadIdx.x] = v;
        A.acc_dev[threadIdx.x] = v;
        if( threadIdx.x == 0 )
            atomicExch( &A.work[5], 0 );
    }
}

// slicetype_frame_cost_recalculate (slicetype.c:999-1024): the cost of an evaluated cell under the frame's current quantiser
// offsets (f_qp_offset after MB-tree, or f_qp_offset_aq for B frames): cost14 * exp2fix8( qp_offset ), new row sums, and
// the frame sum over the interior blocks.  One workgroup, a wave per block row (like cell_reduce_kernel); exp2fix8 is the
// expression aq_kernel uses (explicit multiply and add: the reference has no FMA here either).
__global__ __launch_bounds__( 256 ) void recalc_kernel( LaP P, const uint16_t *__restrict__ lowres_costs, const float *__restrict__ qp_offset,
                                                        const AqLuts *__restrict__ luts, int *__restrict__ row_satds, int *score_host )
{
    __shared__ int sh[4];
    const int W = P.mb_w, H = P.mb_h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n_waves = blockDim.x >> 6;
    int t = 0;
    for( int by = wave; by < H; by += n_waves )
    {
        int row = 0;
        for( int bx = lane; bx < W; bx += 64 )
        {
            const int xy = by * W + bx;
            const int i = (int)__fadd_rn( __fmul_rn( qp_offset[xy], -64.f / 6.f ), 512.5f );
            const int e = i < 0 ? 0 : i > 1023 ? 0xffff : ( ( luts->exp2_lut[i & 63] + 256 ) << ( i >> 6 ) >> 8 );
            const int cost = ( ( lowres_costs[xy] & 0x3FFF ) * e + 128 ) >> 8;
            row += cost;
            if( ( bx > 0 && bx < W - 1 && by > 0 && by < H - 1 ) || W <= 2 || H <= 2 )
                t += cost;
        }
#pragma unroll
        for( int o = 32; o > 0; o >>= 1 )
            row += __shfl_xor( row, o );
        if( lane == 0 )
            row_satds[by] = row;
    }
#pragma unroll
    for( int o = 32; o > 0; o >>= 1 )
        t += __shfl_xor( t, o );
    if( lane == 0 ) sh[wave] = t;
    __syncthreads();
    if( threadIdx.x == 0 )
    {
        int v = 0;
        for( int k = 0; k < n_waves; k++ ) v += sh[k];
        *score_host = v;
    }
}

// P and intra-only cells: no pixel work, one thread per block
__global__ __launch_bounds__( 256 ) void cell_p_kernel( LaP P, const CellArgs *descs, CellArgs single )
{
    const CellArgs A = descs ? load_uniform( descs + blockIdx.y ) : single;
    const int xy = blockIdx.x * blockDim.x + threadIdx.x;
    if( xy >= P.mb_w * P.mb_h )
        return;
    int bcost = COST_MAX_I, list_used = 0;
    if( !A.is_intra_only )
    {
        int c0 = A.costs0[xy];
        if( c0 < bcost ) { bcost = c0; list_used = 1; }
    }
    cell_finish( P, A, xy, bcost, list_used );
}

// B cells: a wave evaluates CELLB_BPW = 8 consecutive blocks of a row AT ONCE on the geometry of the search (me_search.h): a block is 8 lanes,
// a lane one 8-pixel row, reference samples from the strip copies; the three bidirectional candidates of a block (direct-style vectors,
// zero vectors, searched vectors) follow each other in the same lanes, so one instruction stream serves eight blocks and every lane
// works all the time.  (Rounds 2-3 walked the blocks two at a time with the candidates side by side in four lane-group slots, one of them
// idle: four passes of the whole instruction stream per eight blocks -- 7.1 us per 4K cell in a batch against 4.1 now.)  Everything per block
// that is not pixel work is vector code on lane k for block bx0 + k: fetching the vectors and list costs, deriving the candidate vectors
// (slicetype.c:560-600) and the final choice (:601-652); the pixel lanes of group k pick block k's vectors up with ds_bpermute and hand
// the three candidate costs back the same way.
#define CELLB_BPW 8
__device__ __forceinline__ int pack_mv( int x, int y ) { return ( x & 0xFFFF ) | ( y << 16 ); }
// The wave's life is a chain of memory round trips around ~500 instructions, so the order of the requests is the design: the descriptor
// comes through the scalar cache (load_uniform), then ONE round for everything whose address needs no vector -- the per-block words, the
// source rows, both taps of the zero candidate --, then ONE round for the eight taps of the two candidates with vectors; nothing is read
// at the end (a B cell never looks at the intra cost: slicetype.c:735 applies to P cells only).  Rounds 2-4 first half: nine dependent
// round trips (descriptor, pointer fields, words, the list-1 vector behind its pointer, three candidates one after the other, intra cost
// twice).
template <typename T>
__global__ __launch_bounds__( 64 ) void cell_b_kernel( LaP P, const CellArgs *descs, CellArgs single )
{
    const CellArgs A = descs ? load_uniform( descs + blockIdx.z ) : single;
    const int lane = lane_id();
    const int by = blockIdx.y, bx0 = blockIdx.x * CELLB_BPW;
    const int nb = imin2( CELLB_BPW, P.mb_w - bx0 );
    const int g = lane >> 3, l = lane & 7; // block of the wave, row of the block
    const T *fbase = (const T *)A.fenc0, *s0base = (const T *)A.ref0_0, *s1base = (const T *)A.ref1_0; // strips of the source frame's plane 0 and of the two references
    const int strip_elems = ( P.plane_elems / P.stride ) * 16;
    const int row16 = ( 8 * by + l + LA_PAD ) << 4;
    const int bipred_weight = P.weighted_bipred ? 64 - ( A.dist_scale_factor >> 2 ) : 32;
    const int range = 2 * P.mv_range;
    const int smin_y = imax2( 4 * ( -8 * by - 12 ), -range ), smax_y = imin2( 4 * ( 8 * ( P.mb_h - by - 1 ) + 12 ), range - 1 );
    // lane k: the words and the candidate vectors of block bx0 + k
    const int bx_mine = bx0 + imin2( lane, nb - 1 ), xy_mine = by * P.mb_w + bx_mine;
    int pm0 = 0, pm1 = 0, pd0 = 0, pd1 = 0, c0v = 0, c1v = 0, wr = 0;
    if( lane < nb )
    {
        pm0 = (int)(unsigned)A.mvq0[xy_mine]; pm1 = (int)(unsigned)A.mvq1[xy_mine];
        c0v = A.costs0[xy_mine]; c1v = A.costs1[xy_mine];
        if( A.ref1_l0_valid )
            wr = (int)(unsigned)A.ref1_l0[xy_mine];
    }
    // the pixel lanes of group g work on block min( g, nb - 1 ) (a short row end costs its last block again); the source rows and the zero
    // candidate (plane 0 of both references at the block itself, one tap each) need no vector: requested together with the words
    const int gb = imin2( g, nb - 1 ), from = gb << 2; // ds_bpermute address of lane gb
    const int cx0 = 8 * ( bx0 + gb ) + LA_PAD;
    const int o0 = strip_off( cx0, row16, strip_elems );
    const Px8 f = load_px8_at( fbase, o0 );
    const Px8 z0 = load_px8_at( s0base, o0 ), z1 = load_px8_at( s1base, o0 );
    if( lane < nb && A.ref1_l0_valid )
    {
        const int smin_x = imax2( 4 * ( -8 * bx_mine - 12 ), -range ), smax_x = imin2( 4 * ( 8 * ( P.mb_w - bx_mine - 1 ) + 12 ), range - 1 );
        const int rx = (int)(short)( wr & 0xFFFF ), ry = wr >> 16;
        int d0x = ( rx * A.dist_scale_factor + 128 ) >> 8, d0y = ( ry * A.dist_scale_factor + 128 ) >> 8;
        int d1x = d0x - rx, d1y = d0y - ry;
        d0x = iclip3( d0x, smin_x, smax_x ); d0y = iclip3( d0y, smin_y, smax_y );
        d1x = iclip3( d1x, smin_x, smax_x ); d1y = iclip3( d1y, smin_y, smax_y );
        if( P.subme <= 1 ) { d0x &= ~1; d0y &= ~1; d1x &= ~1; d1y &= ~1; }
        pd0 = pack_mv( d0x, d0y ); pd1 = pack_mv( d1x, d1y );
    }
    const bool dmv_nz = ( pd0 | pd1 ) != 0, mv_nz = ( pm0 | pm1 ) != 0;
    const int qd0 = __builtin_amdgcn_ds_bpermute( from, pd0 ), qd1 = __builtin_amdgcn_ds_bpermute( from, pd1 );
    const int qm0 = __builtin_amdgcn_ds_bpermute( from, pm0 ), qm1 = __builtin_amdgcn_ds_bpermute( from, pm1 );
    auto mix = [&]( const Px8 &ra, const Px8 &rb ) -> Px8 {
        Px8 pred;
        if( bipred_weight == 32 )
        {
            pred.lo = avg_px4( ra.lo, rb.lo, (const T *)nullptr ); pred.hi = avg_px4( ra.hi, rb.hi, (const T *)nullptr );
        }
        else
        {
            // ( a w + b ( 64 - w ) + 32 ) >> 6 on packed 16-bit pairs: w = 64 - ( dist_scale_factor >> 2 ) lies in 4 .. 61, so the sum stays below
            // 2^16 for 10-bit samples too and the result never leaves 0 .. pixel_max (mc.c:61-87 pixel_avg_weight_wxh without its clip doing anything)
            const u16x2 wa = { (unsigned short)bipred_weight, (unsigned short)bipred_weight };
            const u16x2 wb = { (unsigned short)( 64 - bipred_weight ), (unsigned short)( 64 - bipred_weight ) };
            const u16x2 rnd = { 32, 32 }, six = { 6, 6 };
            auto one = [&]( uint32_t x, uint32_t y ) { return as_u32( (u16x2)( ( as_u2( x ) * wa + as_u2( y ) * wb + rnd ) >> six ) ); };
            pred.lo.a = one( ra.lo.a, rb.lo.a ); pred.lo.b = one( ra.lo.b, rb.lo.b );
            pred.hi.a = one( ra.hi.a, rb.hi.a ); pred.hi.b = one( ra.hi.b, rb.hi.b );
            pred.lo.raw = sizeof( T ) == 1 ? __builtin_amdgcn_perm( pred.lo.b, pred.lo.a, 0x06040200u ) : 0;
            pred.hi.raw = sizeof( T ) == 1 ? __builtin_amdgcn_perm( pred.hi.b, pred.hi.a, 0x06040200u ) : 0;
        }
        return pred;
    };
    // tap offsets of a candidate pair (list-0 vector pa into reference 0, list-1 vector pc into reference 1): o[0], o[1] / o[2], o[3]
    auto taps = [&]( int pa, int pc, int o[4] ) {
        int ax = (int)(short)( pa & 0xFFFF ), ay = pa >> 16, cx = (int)(short)( pc & 0xFFFF ), cy = pc >> 16;
        if( P.subme <= 1 ) { ax &= ~1; ay &= ~1; cx &= ~1; cy &= ~1; } // half-pel plane pick (slicetype.c:582-589)
        strip_layout::qpel_taps( P.plane_elems, strip_off( cx0 + ( ax >> 2 ), row16 + ( ( ay >> 2 ) << 4 ), strip_elems ), ax, ay, o[0], o[1] );
        strip_layout::qpel_taps( P.plane_elems, strip_off( cx0 + ( cx >> 2 ), row16 + ( ( cy >> 2 ) << 4 ), strip_elems ), cx, cy, o[2], o[3] );
    };
    int od[4], om[4];
    taps( qd0, qd1, od ); taps( qm0, qm1, om );
    // Where none of the wave's eight blocks has a direct-style vector the candidate IS the zero candidate (both taps of a zero vector are
    // the block itself in plane 0, and the rounded average of a run with itself is the run); where none has a searched vector the third
    // candidate is never looked at (mv_nz below).  Wave-uniform tests: still picture areas cost a third of a moving one's B cell.
    const bool any_d = __builtin_amdgcn_ballot_w64( ( qd0 | qd1 ) != 0 ) != 0ull, any_m = __builtin_amdgcn_ballot_w64( ( qm0 | qm1 ) != 0 ) != 0ull;
    Px8 d0a = z0, d0b = z0, d1a = z1, d1b = z1, m0a = z0, m0b = z0, m1a = z1, m1b = z1;
    if( any_d )
    {
        d0a = load_px8_at( s0base, od[0] ); d0b = load_px8_at( s0base, od[1] ); d1a = load_px8_at( s1base, od[2] ); d1b = load_px8_at( s1base, od[3] );
    }
    if( any_m )
    {
        m0a = load_px8_at( s0base, om[0] ); m0b = load_px8_at( s0base, om[1] ); m1a = load_px8_at( s1base, om[2] ); m1b = load_px8_at( s1base, om[3] );
    }
    auto avg8 = [&]( const Px8 &a, const Px8 &b ) -> Px8 {
        Px8 r;
        r.lo = avg_px4( a.lo, b.lo, (const T *)nullptr ); r.hi = avg_px4( a.hi, b.hi, (const T *)nullptr );
        return r;
    };
    const int v_zero = block_cost8<T>( f, mix( z0, z1 ), P.mbcmp_satd );
    int v_dmv = v_zero, v_mv = v_zero;
    if( any_d )
        v_dmv = block_cost8<T>( f, mix( avg8( d0a, d0b ), avg8( d1a, d1b ) ), P.mbcmp_satd );
    if( any_m )
        v_mv = block_cost8<T>( f, mix( avg8( m0a, m0b ), avg8( m1a, m1b ) ), P.mbcmp_satd );
    // lane k collects the costs of block k (any lane of group k holds them) and chooses
    const int back = ( imin2( lane, 7 ) << 3 ) << 2; // ds_bpermute address of lane 8 k
    const int c_dmv = __builtin_amdgcn_ds_bpermute( back, v_dmv ), c_zero = __builtin_amdgcn_ds_bpermute( back, v_zero ), c_mv = __builtin_amdgcn_ds_bpermute( back, v_mv );
    if( lane < nb )
    {
        // the block's word and map entry (cell_finish without its intra comparison, which a bidirectional cell does not make)
        auto finish = [&]( int bcost, int list_used, int *blk, uint16_t *map ) {
            bcost = ( bcost >> P.depth_shift ) + 4;
            blk[xy_mine] = bcost;
            map[xy_mine] = (uint16_t)( imin2( bcost, 0x3FFF ) + ( list_used << 14 ) );
        };
        int bcost = COST_MAX_I, list_used = 0;
        if( c_dmv < bcost ) { bcost = c_dmv; list_used = 3; }              // the scaled vectors of the list-1 reference (zero without them)
        if( dmv_nz && c_zero < bcost ) { bcost = c_zero; list_used = 3; }  // zero vectors, if those were not zero
        if( c0v < bcost ) { bcost = c0v; list_used = 1; }
        if( c1v < bcost ) { bcost = c1v; list_used = 2; }
        if( mv_nz )
        {
            const int c = 5 * P.lambda + c_mv;
            if( c < bcost ) { bcost = c; list_used = 3; }
        }
        finish( bcost, list_used, A.blk, A.lowres_costs );
        if( A.dual )
        {
            // the same block WITHOUT the list-1 reference's vectors (slicetype.c:629 false): the zero vectors take the first place, the
            // rest of the order is unchanged -- all three candidate costs are the ones above
            int b2 = COST_MAX_I, l2 = 0;
            if( c_zero < b2 ) { b2 = c_zero; l2 = 3; }
            if( c0v < b2 ) { b2 = c0v; l2 = 1; }
            if( c1v < b2 ) { b2 = c1v; l2 = 2; }
            if( mv_nz )
            {
                const int c = 5 * P.lambda + c_mv;
                if( c < b2 ) { b2 = c; l2 = 3; }
            }
            finish( b2, l2, A.blk2, A.lowres_costs2 );
        }
    }
}

// ---- batched vtable primitives: SAD / SATD of every block of a plane against a displaced reference ----
// Streaming layout: a lane owns one 16-sample row segment (one 16-byte load for 8-bit pixels) of a 16x16
// region, 16 lanes are the 16 rows of the region, a wave covers four horizontally adjacent regions, so one
// wave-wide load instruction touches 16 full 64-byte lines of the fenc plane.  A DPP quad is four consecutive
// rows, i.e. the rows of a 4x4 tile: the vertical Hadamard and all three block sizes reduce without LDS
// (4x4 -> quad, 8x8 -> quad + half-row mirror, 16x16 -> + row mirror).
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141

template <typename T>
__device__ __forceinline__ void load_row16( const T *p, Px4 f[4] );
template <>
__device__ __forceinline__ void load_row16<uint8_t>( const uint8_t *p, Px4 f[4] )
{
    uint4 w;
    __builtin_memcpy( &w, p, 16 );
    f[0] = px4_from_raw( w.x ); f[1] = px4_from_raw( w.y ); f[2] = px4_from_raw( w.z ); f[3] = px4_from_raw( w.w );
}
template <>
__device__ __forceinline__ void load_row16<uint16_t>( const uint16_t *p, Px4 f[4] )
{
    uint4 w0, w1;
    __builtin_memcpy( &w0, p, 16 );
    __builtin_memcpy( &w1, p + 8, 16 );
    f[0].a = w0.x; f[0].b = w0.y; f[1].a = w0.z; f[1].b = w0.w;
    f[2].a = w1.x; f[2].b = w1.y; f[3].a = w1.z; f[3].b = w1.w;
    f[0].raw = f[1].raw = f[2].raw = f[3].raw = 0;
}
template <typename T>
__device__ __forceinline__ void load_row8( const T *p, Px4 f[2] );
template <>
__device__ __forceinline__ void load_row8<uint8_t>( const uint8_t *p, Px4 f[2] )
{
    uint2 w;
    __builtin_memcpy( &w, p, 8 );
    f[0] = px4_from_raw( w.x ); f[1] = px4_from_raw( w.y );
}
template <>
__device__ __forceinline__ void load_row8<uint16_t>( const uint16_t *p, Px4 f[2] )
{
    uint4 w;
    __builtin_memcpy( &w, p, 16 );
    f[0].a = w.x; f[0].b = w.y; f[1].a = w.z; f[1].b = w.w;
    f[0].raw = f[1].raw = 0;
}

// All seven partition sizes of x264_pixel_function_t.sad / .satd (common/pixel.h:37-59): BW x BH in {16,8,4} x {16,8,4} with
// 16x4 / 4x16 excluded.  A 16-lane group holds a 16x16 region of the planes, lane = row, four Px4 per lane; a region contains
// (16/BW) x (16/BH) blocks.  Every block of the field has its own full-pel displacement (mv[block], raster order of blocks).
// RR = region rows per lane: a lane walks RR vertically adjacent 16x16 regions with all their loads requested before the first
// cost is computed (the displacement of a block is itself a load the reference address depends on: with one region per lane the
// kernel had two dependent memory round trips and 32 bytes in flight per lane; RR = 4 keeps 128).
template <typename T, int BW, int BH, bool SATD, int RR>
__global__ __launch_bounds__( 256 ) void pixel_cmp_batch_kernel( const T *__restrict__ fenc_, const T *__restrict__ ref_, int stride,
                                                                 int regions_w, int regions_h, const int16_t *__restrict__ mv_, int *__restrict__ out_, int xcd_bands, MultiPtrs M )
{
    const T *__restrict__ fenc = MULTI_PICK( M, 0, const T *, fenc_ );
    const T *__restrict__ ref = MULTI_PICK( M, 1, const T *, ref_ );
    const int16_t *__restrict__ mv = MULTI_PICK( M, 2, const int16_t *, mv_ );
    int *__restrict__ out = MULTI_PICK( M, 3, int *, out_ );
    const int lane = lane_id();
    // XCD k takes the k-th horizontal band of the field (device_common.h xcd_band_block): the reference rows two vertically neighbouring
    // workgroups both read (vectors move a block up to the search range) are fetched into ONE L2 -- 4.1 -> 5.1 TB/s on the 265 MB mosaic
    int wg_x = blockIdx.x, wg_y = blockIdx.y;
    if( xcd_bands )
        xcd_band_block( wg_x, wg_y );
    const int rx0 = ( wg_x * 4 + ( threadIdx.x >> 6 ) ) * 4;
    if( rx0 >= regions_w )
        return; // wave-uniform
    const int rx = rx0 + ( lane >> 4 ), row = lane & 15;
    const bool live = rx < regions_w;
    const int rxc = live ? rx : regions_w - 1; // keep every lane in the DPP exchanges
    constexpr int NX = 16 / BW, NY = 16 / BH;  // blocks of a region per row of blocks / per column
    const int bw = regions_w * NX;             // blocks per row of the field
    int m[RR][NX];
#pragma unroll
    for( int q = 0; q < RR; q++ )
    {
        const int ry = imin2( wg_y * RR + q, regions_h - 1 );
        const int by = ry * NY + row / BH;
#pragma unroll
        for( int k = 0; k < NX; k++ )
            __builtin_memcpy( &m[q][k], mv + 2 * ( by * bw + rxc * NX + k ), 4 );
    }
    Px4 f[RR][4], r[RR][4];
#pragma unroll
    for( int q = 0; q < RR; q++ )
    {
        const int ry = imin2( wg_y * RR + q, regions_h - 1 );
        const size_t o = (size_t)( ry * 16 + row ) * stride + rxc * 16;
        load_row16<T>( fenc + o, f[q] );
#pragma unroll
        for( int k = 0; k < NX; k++ )
        {
            const T *rp = ref + (long)o + ( m[q][k] >> 16 ) * stride + (int16_t)m[q][k] + k * BW;
            if( BW == 16 )
                load_row16<T>( rp, r[q] );
            else if( BW == 8 )
                load_row8<T>( rp, r[q] + 2 * k );
            else
                r[q][k] = load_px4( rp );
        }
    }
#pragma unroll
    for( int q = 0; q < RR; q++ )
    {
        const int ry = wg_y * RR + q;
        const int by = ry * NY + row / BH;
        int part[4];
#pragma unroll
        for( int t = 0; t < 4; t++ )
            part[t] = SATD ? satd_partial_px4( f[q][t], r[q][t] ) : sad_partial_px4( f[q][t], r[q][t], (const T *)nullptr );
        // per block of this row of blocks: the 4-sample columns it spans, then the BH rows (quad, half row, row of 16 lanes)
        int mine = 0;
#pragma unroll
        for( int k = 0; k < NX; k++ )
        {
            int v = 0;
#pragma unroll
            for( int t = 0; t < BW / 4; t++ )
                v += part[k * ( BW / 4 ) + t];
            v = reduce_quad( v );
            if( BH >= 8 ) v += dpp_mov<DPP_ROW_HALF_MIRROR>( v );
            if( BH == 16 ) v += dpp_mov<DPP_ROW_MIRROR>( v );
            if( ( row % BH ) == k ) mine = v; // lane k of the block's rows writes block k (NX <= 4 <= BH)
        }
        if( live && ry < regions_h && ( row % BH ) < NX )
            out[by * bw + rxc * NX + ( row % BH )] = SATD ? mine >> 1 : mine;
    }
}

// ---- hpel_filter (common/mc.c:172-196; x264_mc_functions_t.hpel_filter, mc.h:306-307) -------------------------
// The three half-pel planes of a full-resolution plane: 64x16 output tiles, the source tile (+2/+3 samples each
// way) and the unrounded vertical six-tap sums staged in LDS, so every source sample is read from HBM once per
// tile.  dstv also gets the reference's five extra columns (-2,-1, width..width+2).  The reference's int16 row
// buffer with its -10*PIXEL_MAX offset is only a storage trick: the offset cancels in the six taps (sum 32).
#define HPEL_TW 64
#define HPEL_TH 16
#define HPEL_LW 72 // tile columns x0-2 .. x0+69 (66..69 unused padding), a multiple of four
__device__ __forceinline__ void unpack4( uint32_t w, int v[4], const uint8_t * ) { v[0] = w & 255; v[1] = ( w >> 8 ) & 255; v[2] = ( w >> 16 ) & 255; v[3] = w >> 24; }
template <typename T>
__device__ __forceinline__ void lds_row4( const T *p, int v[4] ) // four consecutive samples from a 4-sample aligned LDS position
{
    if( sizeof( T ) == 1 )
    {
        const uint32_t w = *(const uint32_t *)p;
        v[0] = w & 255; v[1] = ( w >> 8 ) & 255; v[2] = ( w >> 16 ) & 255; v[3] = w >> 24;
    }
    else
    {
        const uint2 w = *(const uint2 *)p;
        v[0] = w.x & 0xFFFF; v[1] = w.x >> 16; v[2] = w.y & 0xFFFF; v[3] = w.y >> 16;
    }
}
template <typename T>
__device__ __forceinline__ void store4( T *p, const int v[4] ) // four consecutive samples to global memory (any alignment)
{
    if( sizeof( T ) == 1 )
    {
        const uint32_t w = (uint32_t)v[0] | ( (uint32_t)v[1] << 8 ) | ( (uint32_t)v[2] << 16 ) | ( (uint32_t)v[3] << 24 );
        __builtin_memcpy( p, &w, 4 );
    }
    else
    {
        const uint2 w = make_uint2( (uint32_t)v[0] | ( (uint32_t)v[1] << 16 ), (uint32_t)v[2] | ( (uint32_t)v[3] << 16 ) );
        __builtin_memcpy( p, &w, 8 );
    }
}

template <typename T>
__global__ __launch_bounds__( 256 ) void hpel_filter_kernel( T *__restrict__ dsth, T *__restrict__ dstv, T *__restrict__ dstc, const T *__restrict__ src,
                                                             long stride, int width, int height, int pixel_max )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) T s_src[HPEL_TH + 5][HPEL_LW];
    __shared__ __attribute__( ( aligned( 16 ) ) ) int s_v[HPEL_TH][HPEL_LW];
    const int x0 = blockIdx.x * HPEL_TW, y0 = blockIdx.y * HPEL_TH;
    const int t = threadIdx.x;
    // source tile: columns x0-2 .. x0+TW+2, rows y0-2 .. y0+TH+2, never beyond what the reference itself reads
    // (four samples per load where all four lie inside what the reference reads; the tile edge sample by sample)
    for( int i = t; i < ( HPEL_TH + 5 ) * ( HPEL_LW / 4 ); i += 256 )
    {
        const int r = i / ( HPEL_LW / 4 ), c = 4 * ( i - r * ( HPEL_LW / 4 ) );
        const int x = x0 - 2 + c, y = imin2( y0 - 2 + r, height + 2 );
        const T *row = src + (long)y * stride;
        if( x + 3 <= width + 2 )
            __builtin_memcpy( &s_src[r][c], row + x, 4 * sizeof( T ) );
        else
#pragma unroll
            for( int k = 0; k < 4; k++ )
                s_src[r][c + k] = row[imin2( x + k, width + 2 )];
    }
    __syncthreads();
    // unrounded vertical six-tap sums, four columns per thread
    for( int i = t; i < HPEL_TH * ( HPEL_LW / 4 ); i += 256 )
    {
        const int r = i / ( HPEL_LW / 4 ), c = 4 * ( i - r * ( HPEL_LW / 4 ) );
        int a[6][4];
#pragma unroll
        for( int k = 0; k < 6; k++ )
            lds_row4<T>( &s_src[r + k][c], a[k] );
        int4 v;
        v.x = a[0][0] + a[5][0] - 5 * ( a[1][0] + a[4][0] ) + 20 * ( a[2][0] + a[3][0] );
        v.y = a[0][1] + a[5][1] - 5 * ( a[1][1] + a[4][1] ) + 20 * ( a[2][1] + a[3][1] );
        v.z = a[0][2] + a[5][2] - 5 * ( a[1][2] + a[4][2] ) + 20 * ( a[2][2] + a[3][2] );
        v.w = a[0][3] + a[5][3] - 5 * ( a[1][3] + a[4][3] ) + 20 * ( a[2][3] + a[3][3] );
        *(int4 *)&s_v[r][c] = v;
    }
    __syncthreads();
    // four outputs per thread: tile columns c+2 .. c+5 need the sums / samples of columns c .. c+8
    const int r = t >> 4, c = 4 * ( t & 15 ), y = y0 + r, x = x0 + c;
    if( y < height && x < width )
    {
        const long row = (long)y * stride;
        __attribute__( ( aligned( 16 ) ) ) int sv[12];
        int ss[12];
        *(int4 *)&sv[0] = *(const int4 *)&s_v[r][c];
        *(int4 *)&sv[4] = *(const int4 *)&s_v[r][c + 4];
        *(int4 *)&sv[8] = *(const int4 *)&s_v[r][c + 8];
        lds_row4<T>( &s_src[r + 2][c], ss );
        lds_row4<T>( &s_src[r + 2][c + 4], ss + 4 );
        lds_row4<T>( &s_src[r + 2][c + 8], ss + 8 );
        int ov[4], oc[4], oh[4];
#pragma unroll
        for( int k = 0; k < 4; k++ )
        {
            ov[k] = iclip3( ( sv[k + 2] + 16 ) >> 5, 0, pixel_max );
            oc[k] = iclip3( ( sv[k] + sv[k + 5] - 5 * ( sv[k + 1] + sv[k + 4] ) + 20 * ( sv[k + 2] + sv[k + 3] ) + 512 ) >> 10, 0, pixel_max );
            oh[k] = iclip3( ( ss[k] + ss[k + 5] - 5 * ( ss[k + 1] + ss[k + 4] ) + 20 * ( ss[k + 2] + ss[k + 3] ) + 16 ) >> 5, 0, pixel_max );
        }
        if( x + 4 <= width )
        {
            store4<T>( dstv + row + x, ov );
            store4<T>( dstc + row + x, oc );
            store4<T>( dsth + row + x, oh );
        }
        else
            for( int k = 0; x + k < width; k++ )
            {
                dstv[row + x + k] = (T)ov[k]; dstc[row + x + k] = (T)oc[k]; dsth[row + x + k] = (T)oh[k];
            }
    }
    // the reference's five extra dstv columns (-2, -1, width .. width+2)
    if( t < 5 * HPEL_TH )
    {
        const int e = t % 5, re = t / 5, ye = y0 + re;
        const bool first_tile = blockIdx.x == 0, last_tile = x0 + HPEL_TW >= width;
        const int xe = e < 2 ? e - 2 : width + e - 2;
        if( ye < height && ( e < 2 ? first_tile : last_tile ) )
            dstv[(long)ye * stride + xe] = (T)iclip3( ( s_v[re][xe - ( x0 - 2 )] + 16 ) >> 5, 0, pixel_max );
    }
}

// 8-bit planes: the same three planes by one wave per strip, no LDS and no barrier.  The tiled kernel above spends ~24 scalar integer
// operations per output sample and is bound by them (607 M lane-operations over a 4K plane = 15 us of the 18 us it takes); here a lane
// holds four neighbouring columns as packed 16-bit pairs: the vertical six-tap sums are v_pk_* arithmetic (two columns per operation, the
// sums fit 16 bits), the two horizontal filters are v_dot2_i32_i16 on pairs of neighbouring columns (exact 32-bit accumulation: the
// nested-shift 16-bit form of the reference's assembly wraps for extreme inputs), and the columns a lane needs from its neighbours
// come through wave_shr / wave_shl DPP moves.  A wave walks HPS_R output rows down a strip of 64 x 4 columns (the outer two lanes only
// feed their neighbours) with every row load of the strip in flight before the first is used.
// Measured on a 4K plane: 4 rows per wave 17.2 us, 8 rows 18.1, 2 rows 17.7, 16 rows 23.1 (the tiled kernel 19.8).
#ifndef HPS_R
#define HPS_R 4
#endif
#define HPS_W 248
typedef short hp_s2 __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ hp_s2 hp_as_s2( unsigned v ) { return __builtin_bit_cast( hp_s2, v ); }
__device__ __forceinline__ unsigned hp_as_u( hp_s2 v ) { return __builtin_bit_cast( unsigned, v ); }
__device__ __forceinline__ unsigned hp_prev( unsigned v ) { return (unsigned)__builtin_amdgcn_mov_dpp( (int)v, 0x138, 0xf, 0xf, true ); } // lane - 1 (wave_shr:1)
__device__ __forceinline__ unsigned hp_next( unsigned v ) { return (unsigned)__builtin_amdgcn_mov_dpp( (int)v, 0x130, 0xf, 0xf, true ); } // lane + 1 (wave_shl:1)
// ( low half of a, low half of b ), ( high, high ), ( high of a, low of b ), ( low of a, high of b ) as one register each
__device__ __forceinline__ unsigned hp_lo_lo( unsigned a, unsigned b ) { return __builtin_amdgcn_perm( b, a, 0x05040100 ); }
__device__ __forceinline__ unsigned hp_hi_hi( unsigned a, unsigned b ) { return __builtin_amdgcn_perm( b, a, 0x07060302 ); }
__device__ __forceinline__ unsigned hp_hi_lo( unsigned a, unsigned b ) { return __builtin_amdgcn_perm( b, a, 0x05040302 ); }
__device__ __forceinline__ unsigned hp_lo_hi( unsigned a, unsigned b ) { return __builtin_amdgcn_perm( b, a, 0x07060100 ); }
// ( t + rnd ) >> sh clipped to 8 bits, both halves
__device__ __forceinline__ hp_s2 hp_round_clip( hp_s2 t, short rnd, int sh )
{
    const hp_s2 r = { rnd, rnd }, zero = { 0, 0 }, top = { 255, 255 };
    return __builtin_elementwise_min( __builtin_elementwise_max( ( t + r ) >> (short)sh, zero ), top );
}
// the six-tap filter over columns x-2 .. x+3 given as three pairs of neighbouring columns, + 512 >> 10, clipped to 8 bits
__device__ __forceinline__ unsigned hp_tap6( unsigned p0, unsigned p1, unsigned p2 )
{
    int acc = __builtin_amdgcn_sdot2( hp_as_s2( p0 ), hp_as_s2( 0xFFFB0001u ), 512, false ); //  1 -5
    acc = __builtin_amdgcn_sdot2( hp_as_s2( p1 ), hp_as_s2( 0x00140014u ), acc, false );     // 20 20
    acc = __builtin_amdgcn_sdot2( hp_as_s2( p2 ), hp_as_s2( 0x0001FFFBu ), acc, false );     // -5  1
    return (unsigned)iclip3( acc >> 10, 0, 255 );
}
__global__ __launch_bounds__( 64 ) void hpel_stream_kernel( uint8_t *__restrict__ dsth_, uint8_t *__restrict__ dstv_, uint8_t *__restrict__ dstc_, const uint8_t *__restrict__ src_,
                                                            int stride, int width, int height, MultiPtrs M )
{
    uint8_t *__restrict__ dsth = MULTI_PICK( M, 0, uint8_t *, dsth_ ), *__restrict__ dstv = MULTI_PICK( M, 1, uint8_t *, dstv_ ), *__restrict__ dstc = MULTI_PICK( M, 2, uint8_t *, dstc_ );
    const uint8_t *__restrict__ src = MULTI_PICK( M, 3, const uint8_t *, src_ );
    const int lane = threadIdx.x;
    int wg_x, wg_y;
    xcd_band_block( wg_x, wg_y ); // the five extra rows of a strip are its vertical neighbours' rows: one L2 per band of strips
    const int x = wg_x * HPS_W + 4 * ( lane - 1 ), y0 = wg_y * HPS_R;
    const bool last_tile = wg_x == (int)gridDim.x - 1;
    // The strip's rows y0-2 .. y0+R+2, never beyond what the reference itself reads (columns -2 .. width+2, rows .. height+2), each
    // split once into its ( c0, c2 ) and ( c1, c3 ) pairs of 16-bit values.  Offsets are relative to sample (-2, -2): never negative.
    const uint8_t *s0 = src - 2 * (long)stride - 2;
    hp_s2 e[HPS_R + 5], o[HPS_R + 5];
    const bool whole = x >= -2 && x + 3 <= width + 2;
#pragma unroll
    for( int r = 0; r < HPS_R + 5; r++ )
    {
        const unsigned row = (unsigned)( imin2( y0 + r, height + 4 ) * stride );
        unsigned w = 0;
        if( whole )
            __builtin_memcpy( &w, s0 + ( row + (unsigned)( x + 2 ) ), 4 );
        else
#pragma unroll
            for( int k = 0; k < 4; k++ )
                w |= (unsigned)s0[row + (unsigned)imin2( imax2( x + k + 2, 0 ), width + 4 )] << ( 8 * k );
        e[r] = hp_as_s2( w & 0x00FF00FFu ); o[r] = hp_as_s2( ( w >> 8 ) & 0x00FF00FFu );
    }
    const hp_s2 m5 = { -5, -5 }, m20 = { 20, 20 };
#pragma unroll
    for( int r = 0; r < HPS_R; r++ )
    {
        const int y = y0 + r;
        // vertical sums of rows y-2 .. y+3
        const hp_s2 ve = ( e[r] + e[r + 5] ) + ( e[r + 1] + e[r + 4] ) * m5 + ( e[r + 2] + e[r + 3] ) * m20;
        const hp_s2 vo = ( o[r] + o[r + 5] ) + ( o[r + 1] + o[r + 4] ) * m5 + ( o[r + 2] + o[r + 3] ) * m20;
        const unsigned ov = hp_as_u( hp_round_clip( ve, 16, 5 ) ) | hp_as_u( hp_round_clip( vo, 16, 5 ) ) << 8;
        // centre plane: the filter across the sums (32-bit accumulation: v_dot2 on pairs of neighbouring columns)
        unsigned oc;
        {
            const unsigned E = hp_as_u( ve ), O = hp_as_u( vo ), pe = hp_prev( E ), po = hp_prev( O ), ne = hp_next( E ), no = hp_next( O );
            const unsigned Pm = hp_hi_hi( pe, po ), P0 = hp_lo_lo( E, O ), P1 = hp_hi_hi( E, O ), P2 = hp_lo_lo( ne, no ); // ( c-2, c-1 ) ( c0, c1 ) ( c2, c3 ) ( c4, c5 )
            const unsigned Qm = hp_hi_lo( po, E ), Q0 = hp_lo_hi( O, E ), Q1 = hp_hi_lo( O, ne ), Q2 = hp_lo_hi( no, ne ); // ( c-1, c0 ) ( c1, c2 ) ( c3, c4 ) ( c5, c6 )
            // (bytes 0 and 1 are not joined directly: ROCm 7.2's compiler turns "clip( a >> s ) | clip( b >> s ) << 8" into v_ashr_pk_u8_i32,
            // which writes only the low half of its destination, and then uses the stale upper half as if it were zero)
            const unsigned even = hp_tap6( Pm, P0, P1 ) | hp_tap6( P0, P1, P2 ) << 16;
            const unsigned odd = hp_tap6( Qm, Q0, Q1 ) | hp_tap6( Q0, Q1, Q2 ) << 16;
            oc = even | odd << 8;
        }
        // horizontal plane: the filter across the samples of row y, whose sums fit 16 bits: two outputs per packed operation
        unsigned oh;
        {
            const hp_s2 E = e[r + 2], O = o[r + 2];
            const unsigned uE = hp_as_u( E ), uO = hp_as_u( O ), pe = hp_prev( uE ), po = hp_prev( uO ), ne = hp_next( uE ), no = hp_next( uO );
            const hp_s2 A = hp_as_s2( hp_hi_lo( pe, uE ) ), B = hp_as_s2( hp_hi_lo( po, uO ) );  // ( c-2, c0 ) ( c-1, c1 )
            const hp_s2 G = hp_as_s2( hp_hi_lo( uE, ne ) ), F = hp_as_s2( hp_hi_lo( uO, no ) );  // ( c2, c4 ) ( c3, c5 )
            const hp_s2 he = ( A + F ) + ( B + G ) * m5 + ( E + O ) * m20;                       // outputs x, x+2
            const hp_s2 ho = ( B + hp_as_s2( ne ) ) + ( E + F ) * m5 + ( O + G ) * m20;          // outputs x+1, x+3
            oh = hp_as_u( hp_round_clip( he, 16, 5 ) ) | hp_as_u( hp_round_clip( ho, 16, 5 ) ) << 8;
        }
        if( y < height )
        {
            const unsigned at = (unsigned)( y * stride + x );
            if( lane >= 1 && lane <= 62 && x < width )
            {
                if( x + 4 <= width )
                {
                    __builtin_memcpy( dstv + at, &ov, 4 ); __builtin_memcpy( dstc + at, &oc, 4 ); __builtin_memcpy( dsth + at, &oh, 4 );
                }
                else
                    for( int k = 0; x + k < width; k++ )
                    {
                        dstv[at + k] = (uint8_t)( ov >> ( 8 * k ) ); dstc[at + k] = (uint8_t)( oc >> ( 8 * k ) ); dsth[at + k] = (uint8_t)( oh >> ( 8 * k ) );
                    }
            }
            // the reference's five extra dstv columns (-2, -1, width .. width+2)
            if( wg_x == 0 && lane == 0 )
            {
                uint8_t *q = dstv + (long)y * stride;
                q[-2] = (uint8_t)( ov >> 16 ); q[-1] = (uint8_t)( ov >> 24 );
            }
            if( last_tile && lane >= 1 )
#pragma unroll
                for( int k = 0; k < 4; k++ )
                    if( x + k >= width && x + k <= width + 2 )
                        dstv[at + k] = (uint8_t)( ov >> ( 8 * k ) );
        }
    }
}

// High-bit-depth planes (uint16 samples, up to 14 bits): the same strip walk as hpel_stream_kernel -- one wave per strip of 62 x 4 columns and
// HPS_R rows, no LDS, no barrier, every row load in flight before the first is used -- with 32-bit sums: a vertical six-tap sum of 10-bit
// samples reaches 42 966 and does not fit the packed 16-bit lanes the 8-bit kernel lives on.  A lane loads its four columns of a row as
// two registers of neighbouring sample pairs ( c0, c1 ) ( c2, c3 ); the symmetric row pairs are added as packed pairs first (two columns per
// add, no overflow below 15 bits), then each column's sum is a - 5 b + 20 c in 32 bits.  The centre plane filters those sums across the
// lanes (neighbours through wave_shr / wave_shl DPP moves); the horizontal plane is v_dot2_i32_i16 on neighbouring sample pairs, the odd
// pairs through v_alignbyte.  (10-bit planes took the LDS-tiled kernel until round 4: ~24 scalar operations per output sample, 1.7 TB/s.)
__device__ __forceinline__ int hp16_lo( unsigned v ) { return (int)( v & 0xFFFFu ); }
__device__ __forceinline__ int hp16_hi( unsigned v ) { return (int)( v >> 16 ); }
__device__ __forceinline__ int hp_prev_i( int v ) { return __builtin_amdgcn_mov_dpp( v, 0x138, 0xf, 0xf, true ); } // lane - 1
__device__ __forceinline__ int hp_next_i( int v ) { return __builtin_amdgcn_mov_dpp( v, 0x130, 0xf, 0xf, true ); } // lane + 1
__device__ __forceinline__ int hp16_tap6( unsigned p0, unsigned p1, unsigned p2, int rnd )
{
    int acc = __builtin_amdgcn_sdot2( hp_as_s2( p0 ), hp_as_s2( 0xFFFB0001u ), rnd, false ); //  1 -5
    acc = __builtin_amdgcn_sdot2( hp_as_s2( p1 ), hp_as_s2( 0x00140014u ), acc, false );     // 20 20
    return __builtin_amdgcn_sdot2( hp_as_s2( p2 ), hp_as_s2( 0x0001FFFBu ), acc, false );    // -5  1
}
__device__ __forceinline__ unsigned hp16_pack( int a, int b ) { return (unsigned)a | ( (unsigned)b << 16 ); }
__global__ __launch_bounds__( 64 ) void hpel_stream16_kernel( uint16_t *__restrict__ dsth_, uint16_t *__restrict__ dstv_, uint16_t *__restrict__ dstc_,
                                                              const uint16_t *__restrict__ src_, int stride, int width, int height, int pixel_max, MultiPtrs M )
{
    uint16_t *__restrict__ dsth = MULTI_PICK( M, 0, uint16_t *, dsth_ ), *__restrict__ dstv = MULTI_PICK( M, 1, uint16_t *, dstv_ ), *__restrict__ dstc = MULTI_PICK( M, 2, uint16_t *, dstc_ );
    const uint16_t *__restrict__ src = MULTI_PICK( M, 3, const uint16_t *, src_ );
    const int lane = threadIdx.x;
    int wg_x, wg_y;
    xcd_band_block( wg_x, wg_y ); // the five extra rows of a strip are its vertical neighbours' rows: one L2 per band of strips
    const int x = wg_x * HPS_W + 4 * ( lane - 1 ), y0 = wg_y * HPS_R;
    const bool last_tile = wg_x == (int)gridDim.x - 1;
    // rows y0-2 .. y0+R+2 of the strip, never beyond what the reference itself reads (columns -2 .. width+2, rows .. height+2); offsets
    // are relative to sample (-2, -2): never negative
    const uint16_t *s0 = src - 2 * (long)stride - 2;
    unsigned d0[HPS_R + 5], d1[HPS_R + 5]; // ( c0, c1 ), ( c2, c3 ) of every row
    const bool whole = x >= -2 && x + 3 <= width + 2;
#pragma unroll
    for( int r = 0; r < HPS_R + 5; r++ )
    {
        const unsigned row = (unsigned)( imin2( y0 + r, height + 4 ) * stride );
        if( whole )
        {
            uint2 w;
            __builtin_memcpy( &w, s0 + ( row + (unsigned)( x + 2 ) ), 8 );
            d0[r] = w.x; d1[r] = w.y;
        }
        else
        {
            unsigned c[4];
#pragma unroll
            for( int k = 0; k < 4; k++ )
                c[k] = s0[row + (unsigned)imin2( imax2( x + k + 2, 0 ), width + 4 )];
            d0[r] = c[0] | c[1] << 16; d1[r] = c[2] | c[3] << 16;
        }
    }
#pragma unroll
    for( int r = 0; r < HPS_R; r++ )
    {
        const int y = y0 + r;
        // vertical sums of rows y-2 .. y+3, four columns: the symmetric pairs as packed adds, then a - 5 b + 20 c per column in 32 bits
        int v[4];
        {
            const u16x2 a0 = as_u2( d0[r] ) + as_u2( d0[r + 5] ), b0 = as_u2( d0[r + 1] ) + as_u2( d0[r + 4] ), c0 = as_u2( d0[r + 2] ) + as_u2( d0[r + 3] );
            const u16x2 a1 = as_u2( d1[r] ) + as_u2( d1[r + 5] ), b1 = as_u2( d1[r + 1] ) + as_u2( d1[r + 4] ), c1 = as_u2( d1[r + 2] ) + as_u2( d1[r + 3] );
            const unsigned A0 = as_u32( a0 ), B0 = as_u32( b0 ), C0 = as_u32( c0 ), A1 = as_u32( a1 ), B1 = as_u32( b1 ), C1 = as_u32( c1 );
            v[0] = hp16_lo( A0 ) - 5 * hp16_lo( B0 ) + 20 * hp16_lo( C0 ); v[1] = hp16_hi( A0 ) - 5 * hp16_hi( B0 ) + 20 * hp16_hi( C0 );
            v[2] = hp16_lo( A1 ) - 5 * hp16_lo( B1 ) + 20 * hp16_lo( C1 ); v[3] = hp16_hi( A1 ) - 5 * hp16_hi( B1 ) + 20 * hp16_hi( C1 );
        }
        int ov[4], oc[4], oh[4];
#pragma unroll
        for( int k = 0; k < 4; k++ )
            ov[k] = iclip3( ( v[k] + 16 ) >> 5, 0, pixel_max );
        // centre plane: the six-tap filter across the sums, columns x-2 .. x+6 (the neighbours' through DPP)
        {
            const int m2 = hp_prev_i( v[2] ), m1 = hp_prev_i( v[3] ), p4 = hp_next_i( v[0] ), p5 = hp_next_i( v[1] ), p6 = hp_next_i( v[2] );
            const int c[9] = { m2, m1, v[0], v[1], v[2], v[3], p4, p5, p6 };
#pragma unroll
            for( int k = 0; k < 4; k++ )
                oc[k] = iclip3( ( ( c[k] + c[k + 5] ) - 5 * ( c[k + 1] + c[k + 4] ) + 20 * ( c[k + 2] + c[k + 3] ) + 512 ) >> 10, 0, pixel_max );
        }
        // horizontal plane: the filter across the samples of row y, neighbouring pairs into v_dot2
        {
            const unsigned P0 = d0[r + 2], P1 = d1[r + 2];                                                        // ( c0, c1 ) ( c2, c3 )
            const unsigned Pm = (unsigned)hp_prev_i( (int)P1 ), P2 = (unsigned)hp_next_i( (int)P0 ), P3 = (unsigned)hp_next_i( (int)P1 ); // ( c-2, c-1 ) ( c4, c5 ) ( c6, c7 )
            const unsigned Qm = __builtin_amdgcn_alignbyte( P0, Pm, 2 ), Q0 = __builtin_amdgcn_alignbyte( P1, P0, 2 );   // ( c-1, c0 ) ( c1, c2 )
            const unsigned Q1 = __builtin_amdgcn_alignbyte( P2, P1, 2 ), Q2 = __builtin_amdgcn_alignbyte( P3, P2, 2 );   // ( c3, c4 ) ( c5, c6 )
            oh[0] = iclip3( hp16_tap6( Pm, P0, P1, 16 ) >> 5, 0, pixel_max );
            oh[1] = iclip3( hp16_tap6( Qm, Q0, Q1, 16 ) >> 5, 0, pixel_max );
            oh[2] = iclip3( hp16_tap6( P0, P1, P2, 16 ) >> 5, 0, pixel_max );
            oh[3] = iclip3( hp16_tap6( Q0, Q1, Q2, 16 ) >> 5, 0, pixel_max );
        }
        if( y < height )
        {
            const unsigned at = (unsigned)( y * stride + x );
            if( lane >= 1 && lane <= 62 && x < width )
            {
                if( x + 4 <= width )
                {
                    const uint2 wv = { hp16_pack( ov[0], ov[1] ), hp16_pack( ov[2], ov[3] ) }, wc = { hp16_pack( oc[0], oc[1] ), hp16_pack( oc[2], oc[3] ) },
                                wh = { hp16_pack( oh[0], oh[1] ), hp16_pack( oh[2], oh[3] ) };
                    __builtin_memcpy( dstv + at, &wv, 8 ); __builtin_memcpy( dstc + at, &wc, 8 ); __builtin_memcpy( dsth + at, &wh, 8 );
                }
                else
#pragma unroll
                    for( int k = 0; k < 4; k++ )
                        if( x + k < width )
                        {
                            dstv[at + k] = (uint16_t)ov[k]; dstc[at + k] = (uint16_t)oc[k]; dsth[at + k] = (uint16_t)oh[k];
                        }
            }
            // the reference's five extra dstv columns (-2, -1, width .. width+2)
            if( wg_x == 0 && lane == 0 )
            {
                uint16_t *q = dstv + (long)y * stride;
                q[-2] = (uint16_t)ov[2]; q[-1] = (uint16_t)ov[3];
            }
            if( last_tile && lane >= 1 )
#pragma unroll
                for( int k = 0; k < 4; k++ )
                    if( x + k >= width && x + k <= width + 2 )
                        dstv[at + k] = (uint16_t)ov[k];
        }
    }
}

// Plain device copy, 16 bytes per lane: the measured HBM rate the SAD/SATD figures are quoted against
// (vendor peak and the build's own copy kernel).  A workgroup moves contiguous chunks of 256 x U x 16 bytes: its U loads
// per lane are requested back to back before the first store; NT = non-temporal loads and stores (the data is touched once).
typedef unsigned copy_v4u __attribute__( ( ext_vector_type( 4 ) ) );
template <int U, bool NT>
__global__ __launch_bounds__( 256 ) void copy16_kernel( const copy_v4u *__restrict__ src, copy_v4u *__restrict__ dst, size_t n16 )
{
    const size_t chunk = (size_t)256 * U;
    for( size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk )
    {
        copy_v4u v[U];
#pragma unroll
        for( int u = 0; u < U; u++ )
        {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if( i < n16 )
                v[u] = NT ? __builtin_nontemporal_load( src + i ) : src[i];
        }
#pragma unroll
        for( int u = 0; u < U; u++ )
        {
            const size_t i = base + (size_t)u * 256 + threadIdx.x;
            if( i < n16 )
            {
                if( NT ) __builtin_nontemporal_store( v[u], dst + i );
                else dst[i] = v[u];
            }
        }
    }
}

// ---- D1/Q1 as batched primitives (common/dct.c:157-205,332-386, common/quant.c:50-104) -----------------
// One thread per 4x4 (or 8x8) block; parity/microbench entry, not a production path.
template <typename T, typename C>
__device__ __forceinline__ void fdct4_1d_dev( const int *in, int step, int *out, int ostep )
{
    int s03 = in[0] + in[3 * step], s12 = in[step] + in[2 * step], d03 = in[0] - in[3 * step], d12 = in[step] - in[2 * step];
    out[0] = s03 + s12; out[ostep] = 2 * d03 + d12; out[2 * ostep] = s03 - s12; out[3 * ostep] = d03 - 2 * d12;
}
__device__ __forceinline__ void fdct8_1d_dev( const int *in, int step, int *out, int ostep )
{
    int s07 = in[0] + in[7 * step], s16 = in[step] + in[6 * step], s25 = in[2 * step] + in[5 * step], s34 = in[3 * step] + in[4 * step];
    int d07 = in[0] - in[7 * step], d16 = in[step] - in[6 * step], d25 = in[2 * step] - in[5 * step], d34 = in[3 * step] - in[4 * step];
    int e0 = s07 + s34, e1 = s16 + s25, e2 = s07 - s34, e3 = s16 - s25;
    int o4 = d16 + d25 + ( d07 + ( d07 >> 1 ) ), o5 = d07 - d34 - ( d25 + ( d25 >> 1 ) );
    int o6 = d07 + d34 - ( d16 + ( d16 >> 1 ) ), o7 = d16 - d25 + ( d34 + ( d34 >> 1 ) );
    out[0] = e0 + e1; out[ostep] = o4 + ( o7 >> 2 ); out[2 * ostep] = e2 + ( e3 >> 1 ); out[3 * ostep] = o5 + ( o6 >> 2 );
    out[4 * ostep] = e0 - e1; out[5 * ostep] = o6 - ( o5 >> 2 ); out[6 * ostep] = ( e2 >> 1 ) - e3; out[7 * ostep] = ( o4 >> 2 ) - o7;
}

template <typename T, typename C, typename U>
__global__ __launch_bounds__( 64 ) void dct_quant_kernel( int is8, int n_blocks, const T *__restrict__ fenc, const T *__restrict__ fdec,
                                                          const U *__restrict__ mf, const U *__restrict__ bias, C *__restrict__ coefs, int *nz_out )
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if( i >= n_blocks )
        return;
    const int N = is8 ? 8 : 4;
    const T *fe = fenc + (size_t)i * N * 16, *fd = fdec + (size_t)i * N * 32;
    int d[64], t[64], o[64];
    for( int y = 0; y < N; y++ )
        for( int x = 0; x < N; x++ )
            d[N * y + x] = fe[y * 16 + x] - fd[y * 32 + x];
    if( is8 )
    {
        for( int x = 0; x < 8; x++ ) fdct8_1d_dev( d + x, 8, t + x, 8 );
        for( int v = 0; v < 8; v++ ) fdct8_1d_dev( t + 8 * v, 1, o + v, 8 );
    }
    else
    {
        for( int y = 0; y < 4; y++ ) fdct4_1d_dev<T, C>( d + 4 * y, 1, t + y, 4 );
        for( int u = 0; u < 4; u++ ) fdct4_1d_dev<T, C>( t + 4 * u, 1, o + 4 * u, 1 );
    }
    int nz = 0;
    for( int k = 0; k < N * N; k++ )
    {
        C c = (C)o[k];
        int v = c;
        unsigned m = mf[k], b = bias[k];
        if( v > 0 ) v = (int)( ( b + (unsigned)v ) * m >> 16 );
        else v = -(int)( ( b + (unsigned)( -v ) ) * m >> 16 );
        c = (C)v;
        nz |= c;
        coefs[(size_t)i * N * N + k] = c;
    }
    nz_out[i] = nz != 0;
}


// Frame form (first piece): sub4x4_dct + quant_4x4 (dct.c:157-175, quant.c:50-62) of EVERY 4x4 block
// of a plane against a prediction plane, both resident on the device.  One thread per block, 64 horizontally adjacent
// blocks per wave: each of the four row loads of a wave covers 256 contiguous bytes of a plane row, the sixteen
// coefficients of a block leave as one 32-byte (8-bit) / 64-byte store.  Coefficients are laid out
// [block_y][block_x][16] in the reference's per-block order; nz[block] = the quant return value.
struct QuantTab
{
    uint32_t mf[16], bias[16];
};

template <typename T, typename C>
__global__ __launch_bounds__( 256 ) void frame_dct_quant4x4_kernel( const T *__restrict__ fenc_, long fenc_stride, const T *__restrict__ fdec_, long fdec_stride,
                                                                    int blocks_w, int blocks_h, QuantTab q, C *__restrict__ coefs_, uint8_t *__restrict__ nz_out_, MultiPtrs M )
{
    const T *__restrict__ fenc = MULTI_PICK( M, 0, const T *, fenc_ ), *__restrict__ fdec = MULTI_PICK( M, 1, const T *, fdec_ );
    C *__restrict__ coefs = MULTI_PICK( M, 2, C *, coefs_ );
    uint8_t *__restrict__ nz_out = MULTI_PICK( M, 3, uint8_t *, nz_out_ );
    const int bx = blockIdx.x * 256 + threadIdx.x, by = blockIdx.y;
    if( bx >= blocks_w )
        return;
    int d[16];
#pragma unroll
    for( int y = 0; y < 4; y++ )
    {
        int a[4], b[4];
        load4( fenc + (long)( 4 * by + y ) * fenc_stride + 4 * bx, a );
        load4( fdec + (long)( 4 * by + y ) * fdec_stride + 4 * bx, b );
#pragma unroll
        for( int x = 0; x < 4; x++ )
            d[4 * y + x] = a[x] - b[x];
    }
    int t[16], o[16];
#pragma unroll
    for( int y = 0; y < 4; y++ ) fdct4_1d_dev<T, C>( d + 4 * y, 1, t + y, 4 );
#pragma unroll
    for( int u = 0; u < 4; u++ ) fdct4_1d_dev<T, C>( t + 4 * u, 1, o + 4 * u, 1 );
    int nz = 0;
    __attribute__( ( aligned( 16 ) ) ) C out[16];
#pragma unroll
    for( int k = 0; k < 16; k++ )
    {
        int v = (C)o[k];
        const unsigned m = q.mf[k], b = q.bias[k];
        if( v > 0 ) v = (int)( ( b + (unsigned)v ) * m >> 16 );
        else v = -(int)( ( b + (unsigned)( -v ) ) * m >> 16 );
        out[k] = (C)v;
        nz |= out[k];
    }
    C *dst = coefs + ( (size_t)by * blocks_w + bx ) * 16;
#pragma unroll
    for( int k = 0; k < (int)( 16 * sizeof( C ) / 16 ); k++ )
        ( (uint4 *)dst )[k] = ( (const uint4 *)out )[k];
    nz_out[(size_t)by * blocks_w + bx] = nz != 0;
}

// ---- descriptor tables: pinned host memory -> device memory ---------------------------------------------------------------------
// Every launch that takes a table of descriptors (ingest, search, cells, MB-tree steps, weight jobs) gets it through this kernel
// instead of hipMemcpyAsync.  Measured (rocprofv3 --hip-runtime-trace, eight contexts): a host-to-device hipMemcpyAsync on a stream
// that has an unresolved hipStreamWaitEvent in front of it does not return until the other stream gets there -- 12-41 ms per call,
// ~28 ms per 160-frame pass of a context -- whereas a kernel launch is queued behind the wait and returns at once.  The pinned
// tables are mapped into the device's address space (hipHostMalloc), so the copy is a read over the host link by the kernel itself.
__global__ __launch_bounds__( 256 ) void upload_kernel( uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, unsigned n_words )
{
    const unsigned n16 = n_words >> 2;
    for( unsigned i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256 )
        ( (uint4 *)dst )[i] = ( (const uint4 *)src )[i];
    if( blockIdx.x == 0 && threadIdx.x < ( n_words & 3 ) )
        dst[4 * n16 + threadIdx.x] = src[4 * n16 + threadIdx.x];
}

// ---- MB-tree: common/mc.c:511-598, encoder/slicetype.c:1029-1089 -------------------------
// The host hands over the ordered step list of one macroblock_tree() call; ONE workgroup walks it (steps depend on
// each other through the propagate buffers, a frame at a time), 1024 threads over the macroblocks of a step.  It
// runs on its own stream beside the search work and occupies a single CU.  Propagate buffers are 32-bit
// accumulators: the reference's saturating uint16 adds (MC_CLIP_ADD) only ever add non-negative amounts, so
// clamping the running sum when it is read gives the same value whatever the order of the atomic adds.
struct MbtOpDev
{
    int type, referenced, bipred_weight, fps_factor_i, b_bidir, barrier_before;
    float fps_factor, weightdelta, strength, padf_;
    int *prop_b, *prop_p0, *prop_p1;
    const uint16_t *intra_cost, *lowres_costs, *inv_qscale;
    const unsigned long long *mvq0, *mvq1;
    const float *qp_aq;
    float *qp;
    int lds_b, lds_p0, lds_p1, pad_; // mbtree_lds_kernel: accumulator slots in LDS
};
#define MBT_LDS_LOAD 5   // mbtree_lds_kernel only: global accumulator prop_b -> LDS slot lds_b
#define MBT_LDS_STORE 6  // LDS slot lds_b -> global accumulator prop_b
#define MBT_NOP 7        // a queued FINISH whose frame a later list of the same launch finishes again (the later one wins): skipped

__device__ __forceinline__ int prop_read( const int *p )
{
    int v = __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); // bypass this CU's L1
    return v < 32767 ? v : 32767;
}
// x264_log2( a ) - x264_log2( b ) + w in the association of the reference build (gcc -O3 -ffast-math, the flags the
// reference configures): ( ( lut[a] - int(b) ) + ( int(a) + w ) ) - lut[b]
__device__ __forceinline__ float lut_log2_diff( const AqLuts *luts, unsigned a, unsigned b, float w )
{
    const int lza = __clz( a ), lzb = __clz( b );
    const float t = __fsub_rn( luts->log2_lut[( a << lza >> 24 ) & 0x7f], (float)( 31 - lzb ) );
    return __fsub_rn( __fadd_rn( t, __fadd_rn( (float)( 31 - lza ), w ) ), luts->log2_lut[( b << lzb >> 24 ) & 0x7f] );
}

// one macroblock of a PROPAGATE step: mbtree_propagate_cost + both mbtree_propagate_list scatters
__device__ __forceinline__ void mbt_propagate_mb( const MbtOpDev &o, int *ref0, int *ref1, int W, int H, int i, int ic, int lc, int inv, int in_cost,
                                                  unsigned w0, unsigned w1 )
{
    const int mx = i % W, my = i / W;
    int inter = lc & 0x3FFF;
    if( inter > ic ) inter = ic;
    const float propagate_intra = (float)( ic * inv );
    const float propagate_amount = __fadd_rn( (float)in_cost, __fmul_rn( propagate_intra, o.fps_factor ) );
    const float num = (float)( ic - inter ), den = (float)ic;
    int amount = (int)__fadd_rn( __fdiv_rn( __fmul_rn( propagate_amount, num ), den ), 0.5f );
    if( amount > 32767 ) amount = 32767;
    const int lists_used = lc >> 14;
#pragma unroll
    for( int list = 0; list < 2; list++ )
    {
        if( list && !o.b_bidir ) break;
        if( !( lists_used & ( 1 << list ) ) ) continue;
        int *ref = list ? ref1 : ref0;
        int la = amount;
        if( lists_used == 3 )
            la = ( la * ( list ? 64 - o.bipred_weight : o.bipred_weight ) + 32 ) >> 6;
        const unsigned w = list ? w1 : w0;
        int x = (int)(short)( w & 0xFFFF ), y = (int)w >> 16;
        if( !( x | y ) )
        {
            atomicAdd( &ref[i], la );
            continue;
        }
        const unsigned mbx = (unsigned)( ( x >> 5 ) + mx ), mby = (unsigned)( ( y >> 5 ) + my );
        const unsigned idx0 = mbx + mby * W, idx2 = idx0 + W;
        x &= 31; y &= 31;
        const int q0 = ( ( 32 - y ) * ( 32 - x ) * la + 512 ) >> 10, q1 = ( ( 32 - y ) * x * la + 512 ) >> 10;
        const int q2 = ( y * ( 32 - x ) * la + 512 ) >> 10, q3 = ( y * x * la + 512 ) >> 10;
        if( mby < (unsigned)H )
        {
            if( mbx < (unsigned)W ) atomicAdd( &ref[idx0], q0 );
            if( mbx + 1 < (unsigned)W ) atomicAdd( &ref[idx0 + 1], q1 );
        }
        if( mby + 1 < (unsigned)H )
        {
            if( mbx < (unsigned)W ) atomicAdd( &ref[idx2], q2 );
            if( mbx + 1 < (unsigned)W ) atomicAdd( &ref[idx2 + 1], q3 );
        }
    }
}

// (macroblocks per thread with their loads in flight together.  8 / 16 measured against 4 with the workgroups sized by the number of open
// contexts, scripts/r05_ab_libs.sh: eight contexts 39.0 k / 37.9 k and 38.1 k / 37.9 k frames/s against 38.8 k / 38.8 k, one context
// 27.7 k and 26.0 k against 28.0 k -- no gain)
#ifndef MBT_UNROLL
#define MBT_UNROLL 4
#endif
#define MBT_WGS 4   // workgroups per list at most (x264hip.hip: mbt_wgs_per_list)
#define MBT_THREADS 1024
#define MBT_MAX_GROUPS 48
// The step lists of up to MBT_MAX_GROUPS macroblock_tree() calls, one after the other in the table; list g is steps [beg[g], beg[g+1])
struct MbtGroups
{
    int n, beg[MBT_MAX_GROUPS + 1];
};
// A launch runs the lists of G.n macroblock_tree() calls side by side, list g on workgroups [g * wgs, (g+1) * wgs).  The calls of a
// stream follow each other as a chain of ~25 dependent phases each, and nothing but the accumulators ties one call to the next (every
// call clears the accumulators it uses before it adds to them, slicetype.c:1108-1135): the host gives each list of a launch its own
// accumulator bank, so the chains overlap instead of queueing (x264hip.hip, mbt_flush).
// The wgs workgroups of a list walk its steps together; where a step reads what earlier steps accumulated they meet at a
// counter barrier (monotonic counter, relaxed agent-scope polling, bounded spin).  Everything exchanged between
// steps lives in the propagate accumulators, which are only touched with agent-scope atomics, so no fences are
// needed beyond draining this wave's outstanding operations before it arrives.
__global__ __launch_bounds__( 1024 ) void mbtree_kernel( LaP P, const MbtOpDev *ops_in, MbtGroups G, int wgs, const AqLuts *luts,
                                                         unsigned *bar_all /* per list: [0] arrivals, [2] exits */, unsigned *err )
{
    const int W = P.mb_w, H = P.mb_h, n_mb = W * H;
    const int g = blockIdx.x / wgs;
    const int tid = ( blockIdx.x - g * wgs ) * blockDim.x + threadIdx.x, nthreads = wgs * blockDim.x;
    const MbtOpDev *ops = ops_in + G.beg[g];
    const int n_ops = G.beg[g + 1] - G.beg[g];
    unsigned *bar = bar_all + 4 * g;
    unsigned n_bar = 0;
    for( int k = 0; k < n_ops; k++ )
    {
        const MbtOpDev o = load_uniform( ops + k ); // the same entry for every thread: through the scalar cache (the table was uploaded by an earlier launch)
        if( o.barrier_before )
        {
            __builtin_amdgcn_s_waitcnt( 0 );
            __syncthreads();
            n_bar++;
            if( threadIdx.x == 0 )
            {
                __hip_atomic_fetch_add( &bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                unsigned spins = 0;
                while( __hip_atomic_load( &bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) < n_bar * wgs )
                {
                    __builtin_amdgcn_s_sleep( 2 );
                    if( ++spins > ( 1u << 24 ) )
                    {
                        __hip_atomic_store( err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
                        break;
                    }
                }
            }
            __syncthreads();
        }
        if( o.type == 0 )
        {
            for( int i = tid; i < n_mb; i += nthreads )
                __hip_atomic_store( &o.prop_b[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        }
        else if( o.type == 1 )
        {
            // MBT_UNROLL macroblocks per thread with all their loads in flight together: the step is one memory
            // round trip long instead of one per macroblock
            for( int base = tid; base < n_mb; base += nthreads * MBT_UNROLL )
            {
                int ic[MBT_UNROLL], lc[MBT_UNROLL], inv[MBT_UNROLL], in_cost[MBT_UNROLL];
                unsigned w0[MBT_UNROLL], w1[MBT_UNROLL];
#pragma unroll
                for( int u = 0; u < MBT_UNROLL; u++ )
                {
                    const int i = base + u * nthreads;
                    const bool ok = i < n_mb;
                    const int ii = ok ? i : 0;
                    ic[u] = o.intra_cost[ii]; lc[u] = o.lowres_costs[ii]; inv[u] = o.inv_qscale[ii];
                    in_cost[u] = o.referenced ? prop_read( &o.prop_b[ii] ) : 0;
                    w0[u] = (unsigned)o.mvq0[ii];
                    w1[u] = o.b_bidir ? (unsigned)o.mvq1[ii] : 0u;
                }
#pragma unroll
                for( int u = 0; u < MBT_UNROLL; u++ )
                {
                    const int i = base + u * nthreads;
                    if( i < n_mb )
                        mbt_propagate_mb( o, o.prop_p0, o.prop_p1, W, H, i, ic[u], lc[u], inv[u], in_cost[u], w0[u], w1[u] );
                }
            }
        }
        else if( o.type == 2 ) // (MBT_NOP: a FINISH a later list of the same launch overrides -- nothing to do)
        {
            for( int i = tid; i < n_mb; i += nthreads )
            {
                const int ic = ( (int)o.intra_cost[i] * (int)o.inv_qscale[i] + 128 ) >> 8;
                if( ic )
                {
                    const int pc = ( prop_read( &o.prop_b[i] ) * o.fps_factor_i + 128 ) >> 8;
                    const float ratio = lut_log2_diff( luts, (unsigned)( ic + pc ), (unsigned)ic, o.weightdelta );
                    o.qp[i] = __fsub_rn( o.qp_aq[i], __fmul_rn( o.strength, ratio ) );
                }
            }
        }
    }
    // the last workgroup of the list to leave re-arms its barrier counters for the next launch that uses this ring entry
    __syncthreads();
    if( threadIdx.x == 0 && __hip_atomic_fetch_add( &bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) == (unsigned)wgs - 1 )
    {
        __hip_atomic_store( &bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        __hip_atomic_store( &bar[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    }
}

// The same lists level by level: level L of a list is what lies between its L-th and (L+1)-th barrier, and the steps of one level --
// of every list of the launch -- depend on nothing but earlier levels.  One launch per level, grid.y = the steps of that level (order[]:
// step indices sorted by level), grid.x = blocks of 256 x MBT_UNROLL macroblocks; the kernel boundary is the barrier.  Against the
// barrier kernel above: no workgroup ever waits, so nothing holds a CU while it does no work (a list of the barrier kernel keeps its
// workgroups resident for ~25 phases of ~10 us each, most of it in the counter barrier and the drain of the atomics; with eight
// contexts three or four such launches were resident at any time, profiles/r04_trace_concurrency.json), and the launch fills the chip
// for the few microseconds a level takes instead of 2 x 16 CUs for a millisecond.
__global__ __launch_bounds__( 256 ) void mbtree_level_kernel( LaP P, const MbtOpDev *ops, const int *order, const AqLuts *luts )
{
    const int W = P.mb_w, H = P.mb_h, n_mb = W * H;
    const int k = load_uniform( order + blockIdx.y );
    const MbtOpDev o = load_uniform( ops + k );
    const int first = blockIdx.x * ( 256 * MBT_UNROLL ) + threadIdx.x;
    if( o.type == 0 )
    {
#pragma unroll
        for( int u = 0; u < MBT_UNROLL; u++ )
            if( first + u * 256 < n_mb )
                o.prop_b[first + u * 256] = 0;
    }
    else if( o.type == 1 )
    {
        int ic[MBT_UNROLL], lc[MBT_UNROLL], inv[MBT_UNROLL], in_cost[MBT_UNROLL];
        unsigned w0[MBT_UNROLL], w1[MBT_UNROLL];
#pragma unroll
        for( int u = 0; u < MBT_UNROLL; u++ )
        {
            const int i = first + u * 256;
            const int ii = i < n_mb ? i : 0;
            ic[u] = o.intra_cost[ii]; lc[u] = o.lowres_costs[ii]; inv[u] = o.inv_qscale[ii];
            in_cost[u] = o.referenced ? prop_read( &o.prop_b[ii] ) : 0;
            w0[u] = (unsigned)o.mvq0[ii];
            w1[u] = o.b_bidir ? (unsigned)o.mvq1[ii] : 0u;
        }
#pragma unroll
        for( int u = 0; u < MBT_UNROLL; u++ )
        {
            const int i = first + u * 256;
            if( i < n_mb )
                mbt_propagate_mb( o, o.prop_p0, o.prop_p1, W, H, i, ic[u], lc[u], inv[u], in_cost[u], w0[u], w1[u] );
        }
    }
    else if( o.type == 2 )
    {
#pragma unroll
        for( int u = 0; u < MBT_UNROLL; u++ )
        {
            const int i = first + u * 256;
            if( i >= n_mb ) continue;
            const int ic = ( (int)o.intra_cost[i] * (int)o.inv_qscale[i] + 128 ) >> 8;
            if( ic )
            {
                const int pc = ( prop_read( &o.prop_b[i] ) * o.fps_factor_i + 128 ) >> 8;
                const float ratio = lut_log2_diff( luts, (unsigned)( ic + pc ), (unsigned)ic, o.weightdelta );
                o.qp[i] = __fsub_rn( o.qp_aq[i], __fmul_rn( o.strength, ratio ) );
            }
        }
    }
}

// The same step list walked by ONE workgroup with the accumulators of the frames in play held in LDS (pictures up to ~12 800
// macroblocks: 1080p and below).  A macroblock_tree() call is a chain of ~25 dependent phases (every mini-GOP: its B-frames, then the
// anchor that closes it); across 16 workgroups each phase boundary costs a device-wide barrier plus the drain of the L2 atomics
// (~10 us), several times the work between two boundaries.  Inside one workgroup a boundary is a __syncthreads(), the adds are LDS
// atomics and the in_cost reads are LDS reads; the inputs (costs, vectors) still stream from memory, eight macroblocks per thread in
// flight.  The host maps the accumulators of the call onto the LDS slots (least-recently-used; MBT_LDS_LOAD / MBT_LDS_STORE steps
// move an accumulator in and out) and writes every touched accumulator back at the end, so the global buffers hold what the
// multi-workgroup kernel would have left there.
//
// Round 5: the lists of a launch side by side, list g on workgroup g (x264hip.hip, the queued form).  What counts there is not the latency
// of a call -- the lists run beside the searches, nothing waits for them -- but what they take from the searches: a list on the
// multi-workgroup kernel holds its CUs for milliseconds (every step a round of returning global atomics and a barrier through memory),
// here for a few hundred microseconds, and the L2's atomic units stay out of it.
#define MBT_LDS_UNROLL 8
__global__ __launch_bounds__( 1024 ) void mbtree_lds_kernel( LaP P, const MbtOpDev *ops, MbtGroups G, const AqLuts *luts )
{
    extern __shared__ __attribute__( ( aligned( 16 ) ) ) int mbt_acc[];
    const int W = P.mb_w, H = P.mb_h, n_mb = W * H;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int k_end = G.beg[blockIdx.x + 1];
    for( int k = G.beg[blockIdx.x]; k < k_end; k++ )
    {
        const MbtOpDev o = load_uniform( ops + k );
        int *A_b = mbt_acc + o.lds_b * n_mb, *A_p0 = mbt_acc + o.lds_p0 * n_mb, *A_p1 = mbt_acc + o.lds_p1 * n_mb;
        if( o.type == MBT_NOP )
            continue; // (uniform: nothing was written, no barrier needed)
        if( o.type == 0 )
        {
            // lds_b < 0: an accumulator that never entered LDS (a frame nothing refers to): its global buffer is cleared for later readers
            if( o.lds_b < 0 )
                for( int i = tid; i < n_mb; i += NT ) o.prop_b[i] = 0;
            else
                for( int i = tid; i < n_mb; i += NT ) A_b[i] = 0;
        }
        else if( o.type == MBT_LDS_LOAD )
            for( int i = tid; i < n_mb; i += NT ) A_b[i] = o.prop_b[i];
        else if( o.type == MBT_LDS_STORE )
            for( int i = tid; i < n_mb; i += NT ) o.prop_b[i] = A_b[i];
        else if( o.type == 1 )
        {
            for( int base = tid; base < n_mb; base += NT * MBT_LDS_UNROLL )
            {
                int ic[MBT_LDS_UNROLL], lc[MBT_LDS_UNROLL], inv[MBT_LDS_UNROLL];
                unsigned w0[MBT_LDS_UNROLL], w1[MBT_LDS_UNROLL];
#pragma unroll
                for( int u = 0; u < MBT_LDS_UNROLL; u++ )
                {
                    const int i = base + u * NT;
                    const int ii = i < n_mb ? i : 0;
                    ic[u] = o.intra_cost[ii]; lc[u] = o.lowres_costs[ii]; inv[u] = o.inv_qscale[ii];
                    w0[u] = (unsigned)o.mvq0[ii];
                    w1[u] = o.b_bidir ? (unsigned)o.mvq1[ii] : 0u;
                }
#pragma unroll
                for( int u = 0; u < MBT_LDS_UNROLL; u++ )
                {
                    const int i = base + u * NT;
                    if( i < n_mb )
                    {
                        int in_cost = 0;
                        if( o.referenced )
                        {
                            in_cost = A_b[i];
                            in_cost = in_cost < 32767 ? in_cost : 32767;
                        }
                        mbt_propagate_mb( o, A_p0, A_p1, W, H, i, ic[u], lc[u], inv[u], in_cost, w0[u], w1[u] );
                    }
                }
            }
        }
        else if( o.type == 2 )
        {
            for( int i = tid; i < n_mb; i += NT )
            {
                const int ic = ( (int)o.intra_cost[i] * (int)o.inv_qscale[i] + 128 ) >> 8;
                if( ic )
                {
                    int pr = A_b[i];
                    pr = pr < 32767 ? pr : 32767;
                    const int pc = ( pr * o.fps_factor_i + 128 ) >> 8;
                    const float ratio = lut_log2_diff( luts, (unsigned)( ic + pc ), (unsigned)ic, o.weightdelta );
                    o.qp[i] = __fsub_rn( o.qp_aq[i], __fmul_rn( o.strength, ratio ) );
                }
            }
        }
        __syncthreads();
    }
}

// ---- row forms of the MB-tree entries of x264_mc_functions_t (common/mc.c:511-598), for the exact-signature vtable members that
// x264hip_mc_fill hands out.  The fused step kernel above is what the lookahead itself uses.
__global__ __launch_bounds__( 256 ) void mbt_cost_row_kernel( int16_t *__restrict__ dst, const uint16_t *__restrict__ propagate_in, const uint16_t *__restrict__ intra_costs,
                                                              const uint16_t *__restrict__ inter_costs, const uint16_t *__restrict__ inv_qscales, float fps_factor, int len )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i >= len )
        return;
    const int ic = intra_costs[i];
    int inter = inter_costs[i] & 0x3FFF; // LOWRES_COST_MASK
    if( inter > ic ) inter = ic;
    const float propagate_intra = (float)( ic * (int)inv_qscales[i] );
    const float propagate_amount = __fadd_rn( (float)propagate_in[i], __fmul_rn( propagate_intra, fps_factor ) );
    int amount = (int)__fadd_rn( __fdiv_rn( __fmul_rn( propagate_amount, (float)( ic - inter ) ), (float)ic ), 0.5f );
    dst[i] = (int16_t)( amount > 32767 ? 32767 : amount );
}
// ref_costs32: the frame's accumulators widened to 32 bits for the duration of the call (saturating adds of non-negative amounts
// commute: the clamp is applied when the array is narrowed again)
__global__ __launch_bounds__( 256 ) void mbt_list_row_kernel( int *__restrict__ ref_costs32, const int16_t *__restrict__ mvs, const int16_t *__restrict__ propagate_amount,
                                                              const uint16_t *__restrict__ lowres_costs, int bipred_weight, int mb_y, int len, int list, int W, int H )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i >= len )
        return;
    const int lists_used = lowres_costs[i] >> 14;
    if( !( lists_used & ( 1 << list ) ) )
        return;
    int la = propagate_amount[i];
    if( lists_used == 3 )
        la = ( la * bipred_weight + 32 ) >> 6;
    int x = mvs[2 * i], y = mvs[2 * i + 1];
    if( !( x | y ) )
    {
        atomicAdd( &ref_costs32[mb_y * W + i], la );
        return;
    }
    const unsigned mbx = (unsigned)( ( x >> 5 ) + i ), mby = (unsigned)( ( y >> 5 ) + mb_y );
    const unsigned idx0 = mbx + mby * W, idx2 = idx0 + W;
    x &= 31; y &= 31;
    const int q0 = ( ( 32 - y ) * ( 32 - x ) * la + 512 ) >> 10, q1 = ( ( 32 - y ) * x * la + 512 ) >> 10;
    const int q2 = ( y * ( 32 - x ) * la + 512 ) >> 10, q3 = ( y * x * la + 512 ) >> 10;
    if( mby < (unsigned)H )
    {
        if( mbx < (unsigned)W ) atomicAdd( &ref_costs32[idx0], q0 );
        if( mbx + 1 < (unsigned)W ) atomicAdd( &ref_costs32[idx0 + 1], q1 );
    }
    if( mby + 1 < (unsigned)H )
    {
        if( mbx < (unsigned)W ) atomicAdd( &ref_costs32[idx2], q2 );
        if( mbx + 1 < (unsigned)W ) atomicAdd( &ref_costs32[idx2 + 1], q3 );
    }
}
__global__ __launch_bounds__( 256 ) void widen_u16_kernel( int *__restrict__ dst, const uint16_t *__restrict__ src, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n ) dst[i] = src[i];
}
__global__ __launch_bounds__( 256 ) void narrow_clip15_kernel( uint16_t *__restrict__ dst, const int *__restrict__ src, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n ) dst[i] = (uint16_t)( src[i] < 32767 ? src[i] : 32767 );
}
// only the entries a row's scatter added to are saturated (MC_CLIP_ADD, common/mc.c:527-598); the others keep the caller's 16 bits
__global__ __launch_bounds__( 256 ) void narrow_changed_kernel( uint16_t *__restrict__ dst, const int *__restrict__ sum32, const uint16_t *__restrict__ before, int n )
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if( i < n )
    {
        const int v = sum32[i], o = before[i];
        dst[i] = (uint16_t)( v != o ? ( v < 32767 ? v : 32767 ) : o );
    }
}
