// device_common.h -- shared device-side helpers for the gfx950 lookahead kernels.
//
// Lane geometry used by every block-matching kernel: a wave64 is four "candidate groups" of 16 lanes.
// Inside a group, lane l owns 4 horizontally adjacent pixels of an 8x8 block:
//     quad q = l >> 2 selects the 4x4 tile (tile x = (q&1)*4, tile y = (q>>1)*4), r = l & 3 the row in it.
// A 4x4 tile therefore lives in one DPP quad: the vertical Hadamard butterflies and the first two
// reduction steps are quad_perm DPP moves, the last two row_ror rotations inside the 16-lane row.
// No LDS and no ds_bpermute on the hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LA_PAD 32
#define COST_MAX_I (1 << 28)

struct LaP
{
    int mb_w, mb_h;
    int stride;       // lowres plane stride (pixels)
    int plane_elems;  // pixels per padded plane
    int lambda;
    int me_method, subpel_refine, me_range, mv_range, subme;
    int mbcmp_satd, fpelcmp_satd, weighted_bipred, aq_mode;
    int depth_shift;  // BIT_DEPTH - 8
    int pixel_max;
    const uint16_t *cost_mv; // centred device table
};

struct WtD
{
    int on, scale, denom, offset; // offset already scaled by 1 << depth_shift
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- DPP helpers ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_mov( int v )
{
    return __builtin_amdgcn_update_dpp( 0, v, CTRL, 0xF, 0xF, false );
}
#define DPP_QUAD_XOR1 0xB1 // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E // quad_perm [2,3,0,1]
#define DPP_ROW_ROR4 0x124
#define DPP_ROW_ROR8 0x128

__device__ __forceinline__ int reduce_quad( int v )
{
    v += dpp_mov<DPP_QUAD_XOR1>( v );
    v += dpp_mov<DPP_QUAD_XOR2>( v );
    return v;
}
// sum over the 16 lanes of a row, result in every lane of the row
__device__ __forceinline__ int reduce16( int v )
{
    v = reduce_quad( v );
    v += dpp_mov<DPP_ROW_ROR4>( v );
    v += dpp_mov<DPP_ROW_ROR8>( v );
    return v;
}
__device__ __forceinline__ int iabs( int v ) { return v < 0 ? -v : v; }
__device__ __forceinline__ int imin2( int a, int b ) { return a < b ? a : b; }
__device__ __forceinline__ int imax2( int a, int b ) { return a > b ? a : b; }
__device__ __forceinline__ int iclip3( int v, int lo, int hi ) { return v < lo ? lo : v > hi ? hi : v; }
__device__ __forceinline__ int median3i( int a, int b, int c )
{
    int lo = imin2( a, b ), hi = imax2( a, b );
    return imax2( lo, imin2( hi, c ) );
}
__device__ __forceinline__ int sel4( int g, int a0, int a1, int a2, int a3 )
{
    return g == 0 ? a0 : g == 1 ? a1 : g == 2 ? a2 : a3;
}

// ---- pixel access --------------------------------------------------------------------------------
__device__ __forceinline__ void load4( const uint8_t *p, int v[4] )
{
    uint32_t w;
    __builtin_memcpy( &w, p, 4 ); // gfx950 global loads are byte-addressable: one global_load_dword
    v[0] = w & 255; v[1] = ( w >> 8 ) & 255; v[2] = ( w >> 16 ) & 255; v[3] = w >> 24;
}
__device__ __forceinline__ void load4( const uint16_t *p, int v[4] )
{
    uint2 w;
    __builtin_memcpy( &w, p, 8 );
    v[0] = w.x & 0xFFFF; v[1] = w.x >> 16; v[2] = w.y & 0xFFFF; v[3] = w.y >> 16;
}

__device__ __forceinline__ int weight_px( int v, const WtD &w, int pixel_max )
{
    int r = w.denom >= 1 ? ( ( v * w.scale + ( 1 << ( w.denom - 1 ) ) ) >> w.denom ) + w.offset : v * w.scale + w.offset;
    return iclip3( r, 0, pixel_max );
}

// Four quarter-pel samples at lowres position (x..x+3, y) displaced by (mvx,mvy): rounded average of
// two of the four half-pel planes (both taps coincide for full/half-pel phases, so the code path is
// branch free).  p0 = plane 0 at the block origin; plane k is plane_elems*k further.
template <typename T>
__device__ __forceinline__ void qpel4( const T *p0, int plane_elems, int stride, int x, int y, int mvx, int mvy, int out[4] )
{
    int fx = mvx & 3, fy = mvy & 3;
    int ix = x + ( mvx >> 2 ), iy = y + ( mvy >> 2 );
    int pa = ( fx ? 1 : 0 ) + ( fy == 2 ? 2 : 0 );
    int pb = ( fx == 2 ? 1 : 0 ) + ( fy ? 2 : 0 );
    const T *a = p0 + (size_t)pa * plane_elems + ( iy + ( fy == 3 ) ) * stride + ix;
    const T *b = p0 + (size_t)pb * plane_elems + iy * stride + ix + ( fx == 3 );
    int t[4];
    load4( a, out );
    load4( b, t );
#pragma unroll
    for( int i = 0; i < 4; i++ )
        out[i] = ( out[i] + t[i] + 1 ) >> 1;
}

// ---- block metrics on the 16-lane layout -----------------------------------------------------------
// d[4]: this lane's 4 differences.  Returns the 8x8 block cost in every lane of the 16-lane row.
__device__ __forceinline__ int satd_tile_partial( const int d[4] )
{
    const int lane = lane_id();
    // horizontal 4-point Hadamard in registers
    int s01 = d[0] + d[1], d01 = d[0] - d[1], s23 = d[2] + d[3], d23 = d[2] - d[3];
    int h[4] = { s01 + s23, d01 + d23, s01 - s23, d01 - d23 };
    // vertical 4-point Hadamard across the quad (rows r = lane & 3)
    int acc = 0;
#pragma unroll
    for( int i = 0; i < 4; i++ )
    {
        int v = h[i];
        int p = dpp_mov<DPP_QUAD_XOR1>( v );
        v = ( lane & 1 ) ? p - v : p + v;
        p = dpp_mov<DPP_QUAD_XOR2>( v );
        v = ( lane & 2 ) ? p - v : p + v;
        acc += iabs( v );
    }
    return acc;
}
__device__ __forceinline__ int block_cost8x8( const int d[4], int use_satd )
{
    if( use_satd )
        return reduce16( satd_tile_partial( d ) ) >> 1;
    return reduce16( iabs( d[0] ) + iabs( d[1] ) + iabs( d[2] ) + iabs( d[3] ) );
}
