// Micro-probe: how many L1 (TCP) cache accesses does one wave-wide global_load_dword cost for a given
// lane -> address pattern?  Run under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD` and divide.
//   hipcc --offload-arch=gfx950 -O3 -o probe probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define STRIDE 1024 // bytes per "row"
template <int PAT>
__device__ __forceinline__ unsigned lane_offset( int lane )
{
    const int g = lane >> 4, l = lane & 15;
    switch( PAT )
    {
        case 0: return 0;                                             // every lane the same dword
        case 1: return 4 * lane;                                      // 64 consecutive dwords (4 lines)
        case 2: return ( ( ( l >> 2 ) >> 1 ) * 4 + ( l & 3 ) ) * STRIDE + ( ( l >> 2 ) & 1 ) * 4 + g * 5;  // search kernel, candidate per 16 lanes (rows = quad)
        case 3: return ( ( g >> 1 ) * 4 + ( lane & 3 ) ) * STRIDE + ( g & 1 ) * 4 + ( ( lane >> 2 ) & 3 ) * 5; // tile per 16 lanes, candidates inside
        case 4: return ( l >> 1 ) * STRIDE + ( l & 1 ) * 4 + g * 5;   // 16 lanes = 8 rows x 2 dwords, row-major lanes
        case 5: return ( lane >> 3 ) * STRIDE + ( lane & 7 ) * 4;     // 8 rows x 32 B
        case 6: return ( lane >> 2 ) * STRIDE + ( lane & 3 ) * 4;     // quad = 16 contiguous bytes, 16 rows
        case 7: return ( lane & 3 ) * STRIDE + ( lane >> 2 ) * 4;     // quad = 4 rows, 16 dwords (one line) per row
        case 8: return ( lane & 1 ) * STRIDE + ( lane >> 1 ) * 4;     // pairs of rows, 128 B per row
        case 9: return ( lane & 7 ) * STRIDE + ( lane >> 3 ) * 4;     // 8 rows, 8 dwords per row, row index fastest
        case 10: return 4 * lane + 2;                                 // 64 consecutive dwords, misaligned by 2 bytes
        case 11: return ( lane >> 4 ) * STRIDE + ( lane & 15 ) * 4 + 62; // 4 rows x 64 B straddling a line boundary
    }
    return 0;
}

__device__ __forceinline__ unsigned gload( const char *base, unsigned off )
{
    unsigned w;
    __builtin_memcpy( &w, (const __attribute__( ( address_space( 1 ) ) ) char *)base + off, 4 );
    return w;
}

template <int PAT>
__global__ __launch_bounds__( 64 ) void pattern( const char *buf, unsigned *out, int iters )
{
    const int lane = threadIdx.x;
    const unsigned off = lane_offset<PAT>( lane );
    unsigned acc = 0;
    for( int i = 0; i < iters; i++ )
    {
        acc += gload( buf, off + ( i & 7 ) * 16 * STRIDE );
    }
    if( acc == 0x12345678u )
        out[blockIdx.x] = acc;
}

template <int PAT>
static void run( const char *buf, unsigned *out )
{
    pattern<PAT><<<1024, 64>>>( buf, out, 256 );
    hipDeviceSynchronize();
}

int main()
{
    char *buf; unsigned *out;
    hipMalloc( &buf, 1 << 20 ); hipMemset( buf, 1, 1 << 20 ); hipMalloc( &out, 4096 * 4 );
    run<0>( buf, out ); run<1>( buf, out ); run<2>( buf, out ); run<3>( buf, out ); run<4>( buf, out ); run<5>( buf, out );
    run<6>( buf, out ); run<7>( buf, out ); run<8>( buf, out ); run<9>( buf, out ); run<10>( buf, out ); run<11>( buf, out );
    printf( "done\n" );
    return 0;
}
