// mem_rates.hip -- what one wave-wide load costs the L1 (TCP) and the LDS of an MI355X CU for the lane -> address shapes the block
// search uses or could use.  Every wave reads an L1-resident 16 KB region over and over, 8 independent loads in flight, 16 waves per CU
// (4 per SIMD, the search kernel's occupancy); wall time by HIP events.  Output: ns and CU-cycles per wave-load instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 -o mem_rates mem_rates.hip && ./mem_rates [clock_GHz]
// (measurement aid: not part of the product, not part of the tests)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define AS1 __attribute__( ( address_space( 1 ) ) )
#define REGION 16384
enum { P_NATURAL8, P_MOD8_8, P_NATURAL12, P_NATURAL16, P_LANE7_ONLY, P_HALF_GROUPS, P_COALESCED8, P_BROADCAST8, P_NATURAL4, P_ROWS32_8, P_ROWS32_16U, P_TWO_LANES, P_AL8_O0, P_AL8_O08, P_DW8, P_DW12, P_ODD8, P_AL16_O08, N_PAT };
static const char *pat_name[N_PAT] = {
    "strip rows 16 B apart, lane = row, 8 B per lane (the search kernel)",
    "same rows, lane l takes the row = l mod 8 (row-interleaved), 8 B",
    "strip rows, lane = row, 12 B per lane at the candidate's byte offset (dwordx3, left+right pair)",
    "strip rows, lane = row, the whole 16-B row (dwordx4, aligned)",
    "strip rows, only lane 7 of every group active (one extra row per block)",
    "strip rows, 8 B, groups 0-3 only (half the wave masked)",
    "64 lanes x 8 B contiguous (512 B)",
    "every lane the same 8 B",
    "strip rows, lane = row, 4 B per lane",
    "rows 32 B apart, lane = row, 8 B",
    "rows 32 B apart, lane = row, 16 B at any byte offset (dwordx4 unaligned)",
    "strip rows, lanes 6 and 7 of every group active (two extra rows)",
    "strip rows, 8 B at byte 0 of the row",
    "strip rows, 8 B at byte 0 or 8 of the row (8-byte aligned)",
    "strip rows, 8 B at byte 0 or 4 of the row (dword aligned)",
    "strip rows, 12 B at byte 0 or 4 of the row (dwordx3, dword aligned)",
    "strip rows, 8 B at an odd byte offset",
    "rows 32 B apart, 16 B at byte 0 or 16 (dwordx4 aligned)",
};

typedef unsigned u1v_u __attribute__( ( aligned( 1 ) ) );
typedef unsigned u2v __attribute__( ( ext_vector_type( 2 ) ) );
typedef unsigned u3v __attribute__( ( ext_vector_type( 3 ) ) );
typedef unsigned u4v __attribute__( ( ext_vector_type( 4 ) ) );
typedef u2v u2v_u __attribute__( ( aligned( 1 ) ) );
typedef u3v u3v_u __attribute__( ( aligned( 1 ) ) );
typedef u4v u4v_u __attribute__( ( aligned( 1 ) ) );
template <int W>
__device__ __forceinline__ unsigned gl( const char *base, unsigned off )
{
    const AS1 char *p = (const AS1 char *)base + off;
    if constexpr( W == 4 ) { return *(const AS1 u1v_u *)p; }
    else if constexpr( W == 8 ) { const u2v w = *(const AS1 u2v_u *)p; return w.x ^ w.y; }
    else if constexpr( W == 12 ) { const u3v w = *(const AS1 u3v_u *)p; return w.x ^ w.y ^ w.z; }
    else { const u4v w = *(const AS1 u4v_u *)p; return w.x ^ w.y ^ w.z ^ w.w; }
}

template <int PAT>
__global__ __launch_bounds__( 64 ) void l1_kernel( const char *buf, unsigned *out, int iters )
{
    const int lane = threadIdx.x, g = lane >> 3, l = lane & 7;
    unsigned acc = 0;
    // a different "candidate" per group and per load: row Y (0..200), byte offset o (0..7) inside the 16-B strip row
    unsigned h = 0x9E3779B9u * ( g + 1 ) + blockIdx.x * 7919u;
    for( int i = 0; i < iters; i++ )
    {
#pragma unroll
        for( int k = 0; k < 8; k++ )
        {
            h = h * 1664525u + 1013904223u;
            const unsigned hh = __builtin_amdgcn_ds_bpermute( ( lane & ~7 ) << 2, (int)h ); // group-uniform
            const unsigned Y = ( hh >> 8 ) % 200u, o = ( hh >> 20 ) & 7u;
            unsigned off;
            bool on = true;
            switch( PAT )
            {
                case P_NATURAL8: case P_NATURAL12: case P_NATURAL4: off = 16 * ( Y + l ) + o; break;
                case P_MOD8_8: off = 16 * ( Y + ( ( l - Y ) & 7 ) ) + o; break;
                case P_NATURAL16: off = 16 * ( Y + l ); break;
                case P_LANE7_ONLY: off = 16 * ( Y + l ) + o; on = l == 7; break;
                case P_TWO_LANES: off = 16 * ( Y + l ) + o; on = l >= 6; break;
                case P_HALF_GROUPS: off = 16 * ( Y + l ) + o; on = g < 4; break;
                case P_COALESCED8: off = ( ( hh >> 8 ) % 24u ) * 512 + 8 * lane; break;
                case P_BROADCAST8: off = 8 * ( ( __builtin_amdgcn_readfirstlane( hh ) >> 8 ) % 2000u ); break;
                case P_ROWS32_8: off = 32 * ( ( Y >> 1 ) + l ) + o; break;
                case P_AL8_O0: off = 16 * ( Y + l ); break;
                case P_AL8_O08: off = 16 * ( Y + l ) + ( o & 1 ) * 8; break;
                case P_DW8: case P_DW12: off = 16 * ( Y + l ) + ( o & 1 ) * 4; break;
                case P_ODD8: off = 16 * ( Y + l ) + ( o | 1 ); break;
                case P_AL16_O08: off = 32 * ( ( Y >> 1 ) + l ) + ( o & 1 ) * 16; break;
                case P_ROWS32_16U: off = 32 * ( ( Y >> 1 ) + l ) + o + ( ( hh >> 24 ) & 8u ); break;
            }
            if( on )
                acc ^= gl < PAT == P_NATURAL12 || PAT == P_DW12 ? 12 : PAT == P_NATURAL16 || PAT == P_ROWS32_16U || PAT == P_AL16_O08 ? 16 : PAT == P_NATURAL4 ? 4 : 8 > ( buf, off );
        }
    }
    if( acc == 0x12345678u ) out[blockIdx.x] = acc;
}

enum { L_B64_ALIGNED, L_B64_ANY, L_B32X3, L_B128, L_BPERMUTE, L_B64_MOD8, N_LDS };
static const char *lds_name[N_LDS] = {
    "ds_read_b64, rows 16 B apart, lane = row, 8-byte aligned",
    "ds_read_b64, rows 16 B apart, lane = row, any byte offset",
    "three ds_read_b32 (aligned) covering 8 bytes at any byte offset (counted as one)",
    "ds_read_b128, the whole 16-B row",
    "ds_bpermute_b32",
    "ds_read_b64 aligned, row-interleaved lanes",
};
template <int PAT>
__global__ __launch_bounds__( 64 ) void lds_kernel( const char *buf, unsigned *out, int iters )
{
    __shared__ __attribute__( ( aligned( 16 ) ) ) char win[4096];
    const int lane = threadIdx.x, g = lane >> 3, l = lane & 7;
    for( int i = lane; i < 1024; i += 64 ) ( (unsigned *)win )[i] = ( (const unsigned *)buf )[i];
    __syncthreads();
    unsigned acc = 0;
    unsigned h = 0x9E3779B9u * ( g + 1 ) + blockIdx.x * 7919u;
    for( int i = 0; i < iters; i++ )
    {
#pragma unroll
        for( int k = 0; k < 8; k++ )
        {
            h = h * 1664525u + 1013904223u;
            const unsigned hh = __builtin_amdgcn_readlane( h, 0 ) + 977u * g; // cheap group-uniform value without an LDS instruction
            const unsigned Y = ( hh >> 8 ) % 200u, o = ( hh >> 20 ) & 7u;
            if( PAT == L_B64_ALIGNED ) { uint2 w; __builtin_memcpy( &w, win + 16 * ( Y + l ) + ( o & 8 ), 8 ); acc ^= w.x ^ w.y; }
            if( PAT == L_B64_MOD8 ) { uint2 w; __builtin_memcpy( &w, win + 16 * ( Y + ( ( l - Y ) & 7 ) ), 8 ); acc ^= w.x ^ w.y; }
            if( PAT == L_B64_ANY ) { uint2 w; __builtin_memcpy( &w, win + 16 * ( Y + l ) + o, 8 ); acc ^= w.x ^ w.y; }
            if( PAT == L_B32X3 )
            {
                const unsigned *p = (const unsigned *)( win + 16 * ( Y + l ) + ( o & 4 ) );
                const unsigned a = p[0], b = p[1], c = p[2];
                acc ^= __builtin_amdgcn_alignbyte( b, a, o & 3 ) ^ __builtin_amdgcn_alignbyte( c, b, o & 3 );
            }
            if( PAT == L_B128 ) { uint4 w; __builtin_memcpy( &w, win + 16 * ( Y + l ), 16 ); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
            if( PAT == L_BPERMUTE ) acc ^= __builtin_amdgcn_ds_bpermute( ( ( lane & ~7 ) + ( ( l + Y ) & 7 ) ) << 2, (int)( acc + k ) );
        }
    }
    if( acc == 0x12345678u ) out[blockIdx.x] = acc;
}

template <class F>
static double time_ms( F launch )
{
    hipEvent_t a, b;
    hipEventCreate( &a ); hipEventCreate( &b );
    launch(); hipDeviceSynchronize();
    hipEventRecord( a ); launch(); hipEventRecord( b ); hipEventSynchronize( b );
    float ms = 0; hipEventElapsedTime( &ms, a, b );
    return ms;
}

int main( int argc, char **argv )
{
    hipDeviceProp_t pr; hipGetDeviceProperties( &pr, 0 );
    const int cus = pr.multiProcessorCount;
    const double ghz = argc > 1 ? atof( argv[1] ) : pr.clockRate * 1e-6;
    char *buf; unsigned *out;
    hipMalloc( &buf, 1 << 20 ); hipMalloc( &out, 1 << 20 );
    std::vector<unsigned> init( ( 1 << 20 ) / 4 );
    for( size_t i = 0; i < init.size(); i++ ) init[i] = (unsigned)( i * 2654435761u );
    hipMemcpy( buf, init.data(), 1 << 20, hipMemcpyHostToDevice );
    const int iters = 2000, waves_per_cu = 16, blocks = cus * waves_per_cu;
    printf( "{\"device\": \"%s\", \"cus\": %d, \"clock_GHz\": %.3f, \"waves_per_cu\": %d, \"loads_per_wave\": %d,\n \"l1\": [\n", pr.name, cus, ghz, waves_per_cu, iters * 8 );
#define RUN_L1( P ) { const double ms = time_ms( [&] { l1_kernel<P><<<blocks, 64>>>( buf, out, iters ); } ); \
        const double ns = ms * 1e6 / ( (double)iters * 8 * waves_per_cu ); \
        printf( "  {\"pattern\": \"%s\", \"ns_per_wave_load_per_cu\": %.3f, \"cu_cycles\": %.2f}%s\n", pat_name[P], ns, ns * ghz, P == N_PAT - 1 ? "" : "," ); }
    RUN_L1( P_NATURAL8 ) RUN_L1( P_MOD8_8 ) RUN_L1( P_NATURAL12 ) RUN_L1( P_NATURAL16 ) RUN_L1( P_LANE7_ONLY ) RUN_L1( P_HALF_GROUPS ) RUN_L1( P_COALESCED8 )
    RUN_L1( P_BROADCAST8 ) RUN_L1( P_NATURAL4 ) RUN_L1( P_ROWS32_8 ) RUN_L1( P_ROWS32_16U ) RUN_L1( P_TWO_LANES ) RUN_L1( P_AL8_O0 ) RUN_L1( P_AL8_O08 ) RUN_L1( P_DW8 ) RUN_L1( P_DW12 ) RUN_L1( P_ODD8 ) RUN_L1( P_AL16_O08 )
    printf( " ],\n \"lds\": [\n" );
#define RUN_LDS( P ) { const double ms = time_ms( [&] { lds_kernel<P><<<blocks, 64>>>( buf, out, iters ); } ); \
        const double ns = ms * 1e6 / ( (double)iters * 8 * waves_per_cu ); \
        printf( "  {\"pattern\": \"%s\", \"ns_per_wave_read_per_cu\": %.3f, \"cu_cycles\": %.2f}%s\n", lds_name[P], ns, ns * ghz, P == N_LDS - 1 ? "" : "," ); }
    RUN_LDS( L_B64_ALIGNED ) RUN_LDS( L_B64_ANY ) RUN_LDS( L_B32X3 ) RUN_LDS( L_B128 ) RUN_LDS( L_BPERMUTE ) RUN_LDS( L_B64_MOD8 )
    printf( " ]}\n" );
    return 0;
}
