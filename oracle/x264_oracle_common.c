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
