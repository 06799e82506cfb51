// TEST INFRASTRUCTURE: x264_amd/csrc/vtable_blocks.h (the arithmetic of the batched dct / quant / var2 / ads device entries) compiled
// for the host, so that tests/test_vtable_blocks_host.py can compare it with the oracle without a GPU.
#include <stdint.h>
#define BM_HD inline
#include "vtable_blocks.h"

extern "C" void vtb_dct_u8( int kind, int16_t *out, const uint8_t *fenc, const uint8_t *fdec ) { vt_dct<uint8_t, int16_t>( kind, out, fenc, fdec ); }
extern "C" void vtb_dct_u16( int kind, int32_t *out, const uint16_t *fenc, const uint16_t *fdec ) { vt_dct<uint16_t, int32_t>( kind, out, fenc, fdec ); }
extern "C" int vtb_quant_u8( int kind, int16_t *coef, const uint16_t *mf, const uint16_t *bias, int mf_dc, int bias_dc ) { return vt_quant<int16_t, uint16_t>( kind, coef, mf, bias, mf_dc, bias_dc ); }
extern "C" int vtb_quant_u16( int kind, int32_t *coef, const uint32_t *mf, const uint32_t *bias, int mf_dc, int bias_dc ) { return vt_quant<int32_t, uint32_t>( kind, coef, mf, bias, mf_dc, bias_dc ); }
extern "C" int vtb_var2_u8( const uint8_t *fenc, const uint8_t *fdec, int h, int *ssd ) { return vt_var2<uint8_t>( fenc, fdec, h, ssd ); }
extern "C" int vtb_var2_u16( const uint16_t *fenc, const uint16_t *fdec, int h, int *ssd ) { return vt_var2<uint16_t>( fenc, fdec, h, ssd ); }
extern "C" int vtb_ads( int n_dc, const int *enc_dc, const uint16_t *sums, int delta, const uint16_t *cost_mvx, int16_t *mvs, int width, int thresh )
{
    int n = 0;
    for( int i = 0; i < width; i++ )
        if( vt_ads_one( n_dc, enc_dc, sums + i, delta, cost_mvx[i] ) < thresh )
            mvs[n++] = (int16_t)i;
    return n;
}
