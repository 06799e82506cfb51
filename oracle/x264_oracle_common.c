/* TEST INFRASTRUCTURE ONLY -- depth-independent helpers of the oracle (see x264_oracle.h). */
#include "x264_oracle.h"
#include <math.h>

/* encoder/analyse.c:143-202: cost_mv[+-i] = min( (int)(lambda*logs[i] + .5f), 65535 ),
 * logs[0] = 0.718f, logs[i] = log2f(i+1)*2 + 1.718f.  n = 2*4*mv_range entries each side. */
void or_cost_mv_table( uint16_t *centre, int n, int lambda )
{
    for( int i = 0; i <= n; i++ )
    {
        float l = i ? log2f( (float)( i + 1 ) ) * 2.0f + 1.718f : 0.718f;
        int v = (int)( lambda * l + .5f );
        centre[i] = centre[-i] = (uint16_t)( v < 65535 ? v : 65535 );
    }
}

/* X264_LOOKAHEAD_QP = 12 + 6*(depth-8) (common/common.h:63); x264_lambda_tab there is 1 / 4 */
int or_lambda_for_depth( int bit_depth ) { return bit_depth == 8 ? 1 : bit_depth == 10 ? 4 : -1; }

/* ------------------------------------------------------------------------------------------------
 * MB-tree: common/mc.c:511-598 (mbtree_propagate_cost / _list) and
 * encoder/slicetype.c:1029-1089 (macroblock_tree_finish / _propagate), FP32 like the reference C path.
 * propagate buffers are uint16 with saturation at 32767 (MC_CLIP_ADD, common/mc.h:29).
 * ---------------------------------------------------------------------------------------------- */
#include <string.h>
/* x264_log2( a ) - x264_log2( b ) + w (common/base.h:225-229, slicetype.c:1045) with the association the reference
 * build uses (gcc -O3 -ffast-math, the flags the reference configures): ( ( lut[a] - int(b) ) + ( int(a) + w ) ) - lut[b];
 * w = weightdelta is only non-zero with weightp = FAKE (slicetype.c:462-463).  Pinned against oracle/_ref: 11583 of 11583
 * macroblocks of the f_qp_offset maps exact with this order, < 22 % with the textbook order. */
static float oc_log2_diff( uint32_t a, uint32_t b, float w )
{
    static float lut[128];
    static int init = 0;
    if( !init )
    {
        for( int i = 0; i < 128; i++ )
            lut[i] = (float)( floor( log2( 1.0 + i/128.0 ) * 100000.0 + 0.5 ) / 100000.0 );
        init = 1;
    }
    int lza = __builtin_clz( a ), lzb = __builtin_clz( b );
    float t = lut[( a << lza >> 24 ) & 0x7f] - (float)( 31 - lzb );
    return ( t + ( (float)( 31 - lza ) + w ) ) - lut[( b << lzb >> 24 ) & 0x7f];
}

static inline void clip_add( uint16_t *s, int x )
{
    int t = *s + x;
    *s = (uint16_t)( t < 32767 ? t : 32767 );
}

/* one macroblock_tree_propagate call (slicetype.c:1051-1085) for frame b referencing p0 (list 0) and p1 (list 1) */
void or_mbtree_propagate( int mb_w, int mb_h, const uint16_t *intra_cost, const uint16_t *lowres_costs, const uint16_t *inv_qscale,
                          const uint16_t *propagate_in /* frame b's own buffer, NULL = not referenced (zeros) */,
                          const int16_t (*mvs0)[2], const int16_t (*mvs1)[2], uint16_t *ref0_costs, uint16_t *ref1_costs,
                          int bipred_weight, float fps_factor )
{
    for( int my = 0; my < mb_h; my++ )
        for( int mx = 0; mx < mb_w; mx++ )
        {
            const int i = my*mb_w + mx;
            int ic = intra_cost[i];
            int inter = lowres_costs[i] & 0x3FFF;
            if( inter > ic ) inter = ic;
            float propagate_intra = ic * inv_qscale[i];
            float propagate_amount = ( propagate_in ? propagate_in[i] : 0 ) + propagate_intra*fps_factor;
            float propagate_num = ic - inter;
            float propagate_denom = ic;
            int amount = (int)( propagate_amount * propagate_num / propagate_denom + 0.5f );
            if( amount > 32767 ) amount = 32767;
            amount = (int16_t)amount;
            const int lists_used = lowres_costs[i] >> 14;
            for( int list = 0; list < 2; list++ )
            {
                const int16_t (*mvs)[2] = list ? mvs1 : mvs0;
                uint16_t *ref_costs = list ? ref1_costs : ref0_costs;
                if( !mvs || !ref_costs || !( lists_used & ( 1 << list ) ) )
                    continue;
                int listamount = amount;
                if( lists_used == 3 )
                    listamount = ( listamount * ( list ? 64 - bipred_weight : bipred_weight ) + 32 ) >> 6;
                int x = mvs[i][0], y = mvs[i][1];
                if( !x && !y )
                {
                    clip_add( &ref_costs[i], listamount );
                    continue;
                }
                unsigned mbx = (unsigned)( ( x >> 5 ) + mx ), mby = (unsigned)( ( y >> 5 ) + my );
                unsigned idx0 = mbx + mby*mb_w, idx2 = idx0 + mb_w;
                x &= 31; y &= 31;
                int w0 = ( ( 32-y )*( 32-x ) * listamount + 512 ) >> 10, w1 = ( ( 32-y )*x * listamount + 512 ) >> 10;
                int w2 = ( y*( 32-x ) * listamount + 512 ) >> 10, w3 = ( y*x * listamount + 512 ) >> 10;
                if( mby < (unsigned)mb_h )
                {
                    if( mbx < (unsigned)mb_w ) clip_add( &ref_costs[idx0], w0 );
                    if( mbx+1 < (unsigned)mb_w ) clip_add( &ref_costs[idx0+1], w1 );
                }
                if( mby+1 < (unsigned)mb_h )
                {
                    if( mbx < (unsigned)mb_w ) clip_add( &ref_costs[idx2], w2 );
                    if( mbx+1 < (unsigned)mb_w ) clip_add( &ref_costs[idx2+1], w3 );
                }
            }
        }
}

/* macroblock_tree_finish (slicetype.c:1029-1049) */
void or_mbtree_finish( int n_mb, const uint16_t *intra_cost, const uint16_t *inv_qscale, const uint16_t *propagate_cost,
                       const float *qp_offset_aq, float *qp_offset, int fps_factor, float weightdelta, float strength )
{
    for( int i = 0; i < n_mb; i++ )
    {
        int ic = ( intra_cost[i] * inv_qscale[i] + 128 ) >> 8;
        if( ic )
        {
            int pc = ( propagate_cost[i] * fps_factor + 128 ) >> 8;
            float log2_ratio = oc_log2_diff( ic + pc, ic, weightdelta );
            qp_offset[i] = qp_offset_aq[i] - strength * log2_ratio;
        }
    }
}
