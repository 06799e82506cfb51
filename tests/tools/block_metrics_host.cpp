// TEST TOOL: compiles the arithmetic of x264_amd/csrc/block_metrics.h (the functions the device kernels call) for the host, so
// that tests/test_block_metrics_host.py can check it against the oracle without a GPU.  Not part of libx264hip.so.
#define BM_HD inline
#include "../../x264_amd/csrc/block_metrics.h"
#include "../../x264_amd/csrc/dct_quant_block.h"

template <typename T>
static uint64_t run( int metric, int w, int h, const T *a, long sa, const T *b, long sb )
{
#define CASE( M, W, H ) if( metric == M && w == W && h == H ) return bm_block<T, M, W, H>( a, sa, b, sb );
    CASE( BM_SSD, 16, 16 ) CASE( BM_SSD, 16, 8 ) CASE( BM_SSD, 8, 16 ) CASE( BM_SSD, 8, 8 ) CASE( BM_SSD, 8, 4 ) CASE( BM_SSD, 4, 8 ) CASE( BM_SSD, 4, 4 )
    CASE( BM_SA8D, 16, 16 ) CASE( BM_SA8D, 8, 8 )
    CASE( BM_VAR, 16, 16 ) CASE( BM_VAR, 8, 16 ) CASE( BM_VAR, 8, 8 )
    CASE( BM_HADAMARD_AC, 16, 16 ) CASE( BM_HADAMARD_AC, 16, 8 ) CASE( BM_HADAMARD_AC, 8, 16 ) CASE( BM_HADAMARD_AC, 8, 8 )
    CASE( BM_VSAD, 16, 16 ) CASE( BM_VSAD, 16, 8 )
    CASE( BM_ASD8, 8, 16 ) CASE( BM_ASD8, 8, 8 )
#undef CASE
    return ~0ull;
}
extern "C" uint64_t bm_host_u8( int metric, int w, int h, const uint8_t *a, long sa, const uint8_t *b, long sb ) { return run<uint8_t>( metric, w, h, a, sa, b, sb ); }
extern "C" uint64_t bm_host_u16( int metric, int w, int h, const uint16_t *a, long sa, const uint16_t *b, long sb ) { return run<uint16_t>( metric, w, h, a, sa, b, sb ); }

extern "C" int dq8x8_host_u8( const uint8_t *fe, long fs, const uint8_t *fd, long ds, const uint32_t *mf, const uint32_t *bias, int16_t *out ) { return dq_block8x8<uint8_t, int16_t>( fe, fs, fd, ds, mf, bias, out ); }
extern "C" int dq8x8_host_u16( const uint16_t *fe, long fs, const uint16_t *fd, long ds, const uint32_t *mf, const uint32_t *bias, int32_t *out ) { return dq_block8x8<uint16_t, int32_t>( fe, fs, fd, ds, mf, bias, out ); }
