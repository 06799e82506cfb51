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
    int no_edges;     // slicetype.c:823 do_edges == 0: the outermost ring of blocks is never evaluated (no MB-tree, no VBV)
    int n_slices;     // param.i_lookahead_threads: bands whose searches do not see each other's vectors (slicetype.c:668,917-918)
    const uint16_t *cost_mv; // centred device table
};

// a block slicetype_slice_cost visits (slicetype.c:825-833)
__device__ __forceinline__ bool la_visited( const LaP &P, int bx, int by )
{
    return !P.no_edges || ( bx > 0 && bx < P.mb_w - 1 && by > 0 && by < P.mb_h - 1 );
}

struct WtD
{
    int on, scale, denom, offset; // offset already scaled by 1 << depth_shift
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// An in-kernel wait gave up: what it was waiting for, into the context's pinned error block (16 words: [0] which wait -- 2 a step of
// me_rows_kernel, 3 / 4 the start / a step of me_latency_kernel --, then search, row (group), step or column, the tag waited for, the tag
// last seen, the vector word last seen); word 0 goes last, the host looks at it first
__device__ __forceinline__ void report_wait_timeout( unsigned *err_host, unsigned code, unsigned search, unsigned row, unsigned where, unsigned tag, unsigned long long seen )
{
    // (the FIRST wave to give up reports: the waves above it in the chain give up moments later, for its sake)
    if( atomicAdd_system( err_host + 8, 1u ) != 0u )
    {
        __hip_atomic_store( err_host + 9, ( row << 16 ) | ( search & 0xFFFF ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ); // the last one to give up
        return;
    }
    __hip_atomic_store( err_host + 1, search, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host + 2, row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host + 3, where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host + 4, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host + 5, (unsigned)( seen >> 32 ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host + 6, (unsigned)seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
    __hip_atomic_store( err_host, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM );
}

// ---- DPP helpers ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_mov( int v )
{
    return __builtin_amdgcn_mov_dpp( v, CTRL, 0xF, 0xF, true );
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
__device__ __forceinline__ int imax3( int a, int b, int c ) { return imax2( imax2( a, b ), c ); } // v_max3_i32
__device__ __forceinline__ int imin3( int a, int b, int c ) { return imin2( imin2( a, b ), c ); } // v_min3_i32
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

// ---- pixel access: four pixels per lane, held as two registers of packed 16-bit values ------------------
// gfx950 packed-math (VOP3P) works on pairs of 16-bit values, which is exactly the range this path needs:
// 10-bit samples, differences, and 4x4 Hadamard coefficients (|c| <= 16*1023) all fit int16.  8-bit
// samples additionally keep their raw 4-byte form for the byte-wise v_sad_u8 / v_lerp_u8 instructions.
typedef short s16x2 __attribute__( ( ext_vector_type( 2 ) ) );
typedef unsigned short u16x2 __attribute__( ( ext_vector_type( 2 ) ) );
__device__ __forceinline__ s16x2 as_s2( uint32_t v ) { return __builtin_bit_cast( s16x2, v ); }
__device__ __forceinline__ u16x2 as_u2( uint32_t v ) { return __builtin_bit_cast( u16x2, v ); }
__device__ __forceinline__ uint32_t as_u32( s16x2 v ) { return __builtin_bit_cast( uint32_t, v ); }
__device__ __forceinline__ uint32_t as_u32( u16x2 v ) { return __builtin_bit_cast( uint32_t, v ); }

struct Px4
{
    uint32_t a, b; // {p0, p1}, {p2, p3} as packed u16
    uint32_t raw;  // the same four samples as bytes (meaningful for 8-bit pixels only)
};

__device__ __forceinline__ Px4 px4_from_raw( uint32_t w )
{
    Px4 r;
    r.raw = w;
    r.a = __builtin_amdgcn_perm( 0, w, 0x0c010c00u );
    r.b = __builtin_amdgcn_perm( 0, w, 0x0c030c02u );
    return r;
}
__device__ __forceinline__ Px4 px4_from_ints( const int v[4], bool with_raw )
{
    Px4 r;
    r.a = (uint32_t)v[0] | ( (uint32_t)v[1] << 16 );
    r.b = (uint32_t)v[2] | ( (uint32_t)v[3] << 16 );
    r.raw = with_raw ? __builtin_amdgcn_perm( r.b, r.a, 0x06040200u ) : 0;
    return r;
}
__device__ __forceinline__ void px4_to_ints( const Px4 &p, int v[4] )
{
    v[0] = p.a & 0xFFFF; v[1] = p.a >> 16; v[2] = p.b & 0xFFFF; v[3] = p.b >> 16;
}

__device__ __forceinline__ Px4 load_px4( const uint8_t *p )
{
    uint32_t w;
    __builtin_memcpy( &w, p, 4 ); // gfx950 global loads are byte-addressable: one global_load_dword
    return px4_from_raw( w );
}
__device__ __forceinline__ Px4 load_px4( const uint16_t *p )
{
    uint2 w;
    __builtin_memcpy( &w, p, 8 );
    Px4 r;
    r.a = w.x; r.b = w.y; r.raw = 0;
    return r;
}
// Loads through a wave-uniform base pointer plus an unsigned 32-bit byte offset: the backend emits the
// `global_load_* v, v_off, s[base:base+1]` form (no 64-bit VALU address arithmetic, no flat aperture check).
#define AS_GLOBAL __attribute__( ( address_space( 1 ) ) )
// Workgroups of a 2-D grid reach the 8 XCDs round robin by their linear index, so vertical neighbours -- which share the rows a filter's
// taps or a displaced block reach into -- sit behind different L2s and each fetches those rows from memory.  This bijection of the linear
// index hands XCD k (indices k, k + 8, ...) ONE contiguous run of the row-major order, i.e. a horizontal band of the field: shared rows are
// fetched once per band border instead of once per workgroup row.  (bx, by) replace blockIdx.x / blockIdx.y; scalar arithmetic only.
// (A 3-D grid -- the multi-plane launches, plane = blockIdx.z -- is dealt to the XCDs by the index over all three dimensions: slice z
// starts at XCD ( z G ) mod 8, so the classes are rotated by that much inside the slice.)
__device__ __forceinline__ void xcd_band_block( int &bx, int &by )
{
    const int G = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int shift = (int)( ( blockIdx.z * (unsigned)G ) & 7u ), xcd = ( id + shift ) & 7;
    int start = 0;
    for( int j = 0; j < xcd; j++ )
        start += ( G - ( ( j - shift ) & 7 ) + 7 ) >> 3;   // workgroups of this slice on XCD j: indices ( j - shift ) mod 8, + 8, ...
    const int id2 = start + ( ( id - ( ( xcd - shift ) & 7 ) ) >> 3 );
    by = id2 / (int)gridDim.x;
    bx = id2 - by * (int)gridDim.x;
}
// The streaming primitives in their multi-plane form (x264hip_*_multi): up to 16 independent planes / plane pairs per launch, blockIdx.z
// picks the set; the pointers travel by value in the kernel arguments (scalar loads).  n == 0: the pointers given as plain arguments.
// A launch over one 4K plane pair lasts 5 us and is half launch and ramp; sixteen of them in one launch run at the rate of a large field.
#define MULTI_PLANES_MAX 16
struct MultiPtrs
{
    int n, pad_;
    void *p[4][MULTI_PLANES_MAX]; // four pointer roles of the kernel at hand, one entry per plane set
};
#define MULTI_PICK( M, role, T, plain ) ( ( M ).n ? (T)( M ).p[role][blockIdx.z] : ( plain ) )

// A descriptor every lane of the wave reads from the same address (a table entry picked by blockIdx): fetched through the scalar cache, so
// its fields -- the pointers above all -- land in SGPRs: one s_load instead of a round of per-lane loads, and every load through one of
// those pointers is a plain global load with a scalar base (no flat aperture check, no v_readfirstlane pair in front of it).  The table
// was written by an earlier launch on the same stream; the scalar cache is invalidated at the start of every dispatch.
template <typename D>
__device__ __forceinline__ D load_uniform( const D *p )
{
    D r;
    __builtin_memcpy( &r, (const __attribute__( ( address_space( 4 ) ) ) D *)p, sizeof( D ) );
    return r;
}
// a pointer every lane agrees on, moved into scalar registers (the compiler cannot prove it for values loaded through a table)
template <typename P>
__device__ __forceinline__ P *uniform_ptr( P *p )
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane( (unsigned)v ), hi = __builtin_amdgcn_readfirstlane( (unsigned)( v >> 32 ) );
    return (P *)( ( (unsigned long long)hi << 32 ) | lo );
}
__device__ __forceinline__ uint32_t gload_u32( const void *ubase, unsigned byte_off )
{
    uint32_t w;
    __builtin_memcpy( &w, (const AS_GLOBAL char *)ubase + byte_off, 4 );
    return w;
}
__device__ __forceinline__ uint2 gload_u64( const void *ubase, unsigned byte_off )
{
    uint2 w;
    __builtin_memcpy( &w, (const AS_GLOBAL char *)ubase + byte_off, 8 );
    return w;
}
typedef unsigned u32x3 __attribute__( ( ext_vector_type( 3 ) ) );
__device__ __forceinline__ u32x3 gload_u96( const void *ubase, unsigned byte_off ) // byte_off a multiple of 4
{
    typedef u32x3 u32x3_a4 __attribute__( ( aligned( 4 ) ) );
    return *(const AS_GLOBAL u32x3_a4 *)( (const AS_GLOBAL char *)ubase + byte_off );
}
__device__ __forceinline__ int gload_u16( const void *ubase, unsigned byte_off )
{
    uint16_t w;
    __builtin_memcpy( &w, (const AS_GLOBAL char *)ubase + byte_off, 2 );
    return w;
}
__device__ __forceinline__ Px4 load_px4_at( const uint8_t *ubase, int elem_off )
{
    return px4_from_raw( gload_u32( ubase, (unsigned)elem_off ) );
}
__device__ __forceinline__ Px4 load_px4_at( const uint16_t *ubase, int elem_off )
{
    const uint2 w = gload_u64( ubase, (unsigned)elem_off << 1 );
    Px4 r;
    r.a = w.x; r.b = w.y; r.raw = 0;
    return r;
}
__device__ __forceinline__ int mad24( int a, int b, int c ) { return __mul24( a, b ) + c; }

__device__ __forceinline__ void load4( const uint8_t *p, int v[4] )
{
    uint32_t w;
    __builtin_memcpy( &w, p, 4 );
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
template <typename T>
__device__ __forceinline__ Px4 weight_px4( const Px4 &p, const WtD &w, int pixel_max )
{
    int v[4];
    px4_to_ints( p, v );
#pragma unroll
    for( int i = 0; i < 4; i++ )
        v[i] = weight_px( v[i], w, pixel_max );
    return px4_from_ints( v, sizeof( T ) == 1 );
}

// rounded average of two sample quads
__device__ __forceinline__ Px4 avg_px4( const Px4 &x, const Px4 &y, const uint8_t * )
{
    return px4_from_raw( __builtin_amdgcn_lerp( x.raw, y.raw, 0x01010101u ) ); // (a + b + 1) >> 1 per byte
}
__device__ __forceinline__ Px4 avg_px4( const Px4 &x, const Px4 &y, const uint16_t * )
{
    const u16x2 one = { 1, 1 };
    Px4 r;
    r.a = as_u32( (u16x2)( ( as_u2( x.a ) + as_u2( y.a ) + one ) >> (u16x2){ 1, 1 } ) );
    r.b = as_u32( (u16x2)( ( as_u2( x.b ) + as_u2( y.b ) + one ) >> (u16x2){ 1, 1 } ) );
    r.raw = 0;
    return r;
}

// ---- block metrics on the 16-lane layout -----------------------------------------------------------
// {v.lo + v.hi, v.lo - v.hi}: the one butterfly of the horizontal transform that crosses register halves,
// as a single VOP3P multiply-add with operand-half selection (hi*{1,-1} + lo)
__device__ __forceinline__ s16x2 cross_half_butterfly( s16x2 v )
{
    uint32_t r;
    const uint32_t k = 0xFFFF0001u; // { +1, -1 }
    asm( "v_pk_mad_i16 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"( r ) : "v"( as_u32( v ) ), "v"( k ) );
    return as_s2( r );
}

// per-lane sum of |4x4 Hadamard coefficients| this lane holds after the quad-wide transform.
// The last butterfly, the absolute values and their sum are ONE v_sad_u16 per register: the rows of the quad carry a bias of 0x8000 in
// every coefficient (added to sample 0 of the even rows' differences: a horizontal transform spreads it over the four coefficients of
// the row, the first vertical step leaves it once in every lane), so the signed coefficients compare as unsigned 16-bit numbers, and
// |partner +- own| = |(+-partner + bias) - (own + bias)| where the lanes that would be subtracted send their value negated
// (-(t + 0x8000) == -t + 0x8000 mod 2^16).  16 instructions per four samples against 22 with a multiply-add step and max( x, -x ).
__device__ __forceinline__ int satd_partial_px4( const Px4 &f, const Px4 &r, unsigned acc = 0u )
{
    const int lane = lane_id();
    const uint32_t bias = ( lane & 1 ) ? 0u : 0x8000u; // f.a ^ bias is loop-invariant wherever the source block is
    const s16x2 d01 = as_s2( f.a ^ bias ) - as_s2( r.a ), d23 = as_s2( f.b ) - as_s2( r.b );
    // horizontal 4-point Hadamard: {d0+d2, d1+d3}, {d0-d2, d1-d3}, then the cross-half butterflies
    s16x2 X = cross_half_butterfly( d01 + d23 ), Y = cross_half_butterfly( d01 - d23 );
    // vertical 4-point Hadamard across the quad (rows r = lane & 3): v' = partner + sign * v
    const s16x2 s1 = ( lane & 1 ) ? (s16x2){ -1, -1 } : (s16x2){ 1, 1 };
    const s16x2 s2 = ( lane & 2 ) ? (s16x2){ -1, -1 } : (s16x2){ 1, 1 };
    X = X * s1 + as_s2( (uint32_t)dpp_mov<DPP_QUAD_XOR1>( (int)as_u32( X ) ) );
    Y = Y * s1 + as_s2( (uint32_t)dpp_mov<DPP_QUAD_XOR1>( (int)as_u32( Y ) ) );
    const uint32_t wx = (uint32_t)dpp_mov<DPP_QUAD_XOR2>( (int)as_u32( X * s2 ) ), wy = (uint32_t)dpp_mov<DPP_QUAD_XOR2>( (int)as_u32( Y * s2 ) );
    return (int)__builtin_amdgcn_sad_u16( wy, as_u32( Y ), __builtin_amdgcn_sad_u16( wx, as_u32( X ), acc ) );
}
__device__ __forceinline__ int sad_partial_px4( const Px4 &f, const Px4 &r, const uint8_t * )
{
    return (int)__builtin_amdgcn_sad_u8( f.raw, r.raw, 0u );
}
__device__ __forceinline__ int sad_partial16( const Px4 &f, const Px4 &r )
{
    const s16x2 d01 = as_s2( f.a ) - as_s2( r.a ), d23 = as_s2( f.b ) - as_s2( r.b );
    const s16x2 a0 = __builtin_elementwise_max( d01, -d01 ), a1 = __builtin_elementwise_max( d23, -d23 );
    return (int)__builtin_amdgcn_udot2( as_u2( as_u32( a0 ) ) + as_u2( as_u32( a1 ) ), (u16x2){ 1, 1 }, 0u, false );
}
__device__ __forceinline__ int sad_partial_px4( const Px4 &f, const Px4 &r, const uint16_t * ) { return sad_partial16( f, r ); }

// cost of the 8x8 block this 16-lane row holds, in every lane of the row
template <typename T>
__device__ __forceinline__ int block_cost8x8( const Px4 &f, const Px4 &r, int use_satd )
{
    if( use_satd )
        return reduce16( satd_partial_px4( f, r ) ) >> 1;
    return reduce16( sad_partial_px4( f, r, (const T *)nullptr ) );
}
