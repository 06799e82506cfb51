/* TEST INFRASTRUCTURE ONLY -- a stand-in for libx264hip.so over the CPU oracle (oracle/liboracle.so), exporting the handful of level-1
 * entry points x264_amd/csrc/slicetype_hip.c binds: open / close / strerror / frame_put / frame_cost / get_mvs / get_lowres_costs /
 * get_intra_costs, 8-bit, same contracts as include/x264hip.h.
 *
 * Why: the reference encoder with its accelerator hook bound to "the library" (oracle/_ref/libx264ref8hip.so) can then run in the CPU
 * suite: tests/test_reference_seam.py points $X264HIP_LIB at this file's .so and checks that a whole x264_encoder_encode run through the
 * hook equals the plain C run -- which tests the binding (what it writes into the reference's arrays, when) without a GPU, and pins the
 * oracle against the reference at the level of a whole encode.  The GPU suite runs the same test on the real library.
 * Never loaded by the product: x264_amd/ has no CPU path. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "x264hip.h"
#include "x264_oracle.h"

#define API __attribute__(( visibility( "default" ) ))

typedef struct
{
    int valid;
    int16_t ( *mvs )[2];
    int *costs;
} field_t;

typedef struct
{
    int valid;
    uint8_t *buf, *plane[4];   /* four padded half-pel planes, plane[i] = pixel (0, 0) */
    uint8_t *wbuf;
    uint16_t *inv_qscale, *intra;
    int *rows_intra;
    field_t *field[2];         /* [list][dist - 1] */
    uint16_t **cell;           /* [(bframes+2)^2] lowres_costs, allocated on first use */
    int **rows;
} slot_t;

struct x264hip_ctx
{
    x264hip_params p;
    or_la_cfg cfg;
    uint16_t *cost_mv;
    int n_mb, plane_elems;
    slot_t *slots;
};

API const char *x264hip_strerror( int code )
{
    return code == 0 ? "ok" : code == X264HIP_EINVAL ? "invalid argument" : code == X264HIP_ENOMEM ? "out of memory" : code == X264HIP_ESTATE ? "not evaluated" : "error";
}

API int x264hip_open( x264hip_ctx **out, int device, const x264hip_params *params )
{
    (void)device;
    if( !out || !params || params->bit_depth != 8 || !params->cost_mv ) return X264HIP_EINVAL;
    x264hip_ctx *c = calloc( 1, sizeof( *c ) );
    if( !c ) return X264HIP_ENOMEM;
    c->p = *params;
    const int mb_w = ( params->width + 15 ) / 16, mb_h = ( params->height + 15 ) / 16, n_tab = 2 * 4 * params->mv_range;
    c->n_mb = mb_w * mb_h;
    c->cost_mv = malloc( ( 2 * n_tab + 1 ) * sizeof( uint16_t ) );
    memcpy( c->cost_mv, params->cost_mv - n_tab, ( 2 * n_tab + 1 ) * sizeof( uint16_t ) );
    or_la_cfg *g = &c->cfg;
    g->mb_w = mb_w; g->mb_h = mb_h; g->stride = ( 8 * mb_w + 2 * OR_PAD + 63 ) / 64 * 64;
    g->lambda = params->lambda; g->me_method = params->me_method; g->subpel_refine = params->subpel_refine; g->me_range = params->me_range;
    g->mv_range = params->mv_range; g->subme = params->subme; g->mbcmp_satd = params->mbcmp_satd; g->fpelcmp_satd = params->fpelcmp_satd;
    g->weighted_bipred = params->weighted_bipred; g->aq_mode = params->aq_mode; g->bframe_bias = params->bframe_bias;
    g->n_slices = params->lookahead_slices > 1 ? params->lookahead_slices : 1;
    g->do_edges = !params->no_edges || mb_w <= 2 || mb_h <= 2;
    g->cost_mv = c->cost_mv + n_tab;
    c->plane_elems = g->stride * ( 8 * mb_h + 2 * OR_PAD );
    c->slots = calloc( params->max_frames, sizeof( slot_t ) );
    *out = c;
    return X264HIP_OK;
}

static void slot_clear( x264hip_ctx *c, slot_t *s )
{
    const int ns = c->p.bframes + 2;
    for( int l = 0; l < 2; l++ )
        for( int d = 0; s->field[l] && d <= c->p.bframes; d++ )
            s->field[l][d].valid = 0;
    for( int i = 0; s->cell && i < ns * ns; i++ )
    {
        free( s->cell[i] ); free( s->rows[i] );
        s->cell[i] = NULL; s->rows[i] = NULL;
    }
}

API void x264hip_close( x264hip_ctx *c )
{
    if( !c ) return;
    for( int i = 0; i < c->p.max_frames; i++ )
    {
        slot_t *s = &c->slots[i];
        if( !s->buf ) continue;
        slot_clear( c, s );
        for( int l = 0; l < 2; l++ )
        {
            for( int d = 0; d <= c->p.bframes; d++ ) { free( s->field[l][d].mvs ); free( s->field[l][d].costs ); }
            free( s->field[l] );
        }
        free( s->buf ); free( s->wbuf ); free( s->inv_qscale ); free( s->intra ); free( s->rows_intra ); free( s->cell ); free( s->rows );
    }
    free( c->slots ); free( c->cost_mv ); free( c );
}

API int x264hip_frame_put( x264hip_ctx *c, int slot, const void *luma, int stride, int is_device, const void *cb, const void *cr, int cstride, const uint16_t *inv_qscale )
{
    if( !c || slot < 0 || slot >= c->p.max_frames || !luma || is_device ) return X264HIP_EINVAL;
    slot_t *s = &c->slots[slot];
    const or_la_cfg *g = &c->cfg;
    const int ns = c->p.bframes + 2;
    if( !s->buf )
    {
        s->buf = calloc( 4 * (size_t)c->plane_elems, 1 ); s->wbuf = calloc( c->plane_elems, 1 );
        s->inv_qscale = malloc( c->n_mb * sizeof( uint16_t ) ); s->intra = malloc( c->n_mb * sizeof( uint16_t ) ); s->rows_intra = malloc( g->mb_h * sizeof( int ) );
        for( int l = 0; l < 2; l++ )
        {
            s->field[l] = calloc( c->p.bframes + 1, sizeof( field_t ) );
            for( int d = 0; d <= c->p.bframes; d++ )
            {
                s->field[l][d].mvs = calloc( c->n_mb, sizeof( int16_t[2] ) ); s->field[l][d].costs = calloc( c->n_mb, sizeof( int ) );
            }
        }
        s->cell = calloc( ns * ns, sizeof( uint16_t * ) ); s->rows = calloc( ns * ns, sizeof( int * ) );
        for( int i = 0; i < 4; i++ )
            s->plane[i] = s->buf + (size_t)i * c->plane_elems + OR_PAD * g->stride + OR_PAD;
    }
    slot_clear( c, s );
    for( int l = 0; l < 2; l++ ) /* mc.c:471-481 + frame.c:283-285: a fresh picture has zero vectors */
        for( int d = 0; d <= c->p.bframes; d++ )
            memset( s->field[l][d].mvs, 0, c->n_mb * sizeof( int16_t[2] ) );
    or8_lowres_init( luma, stride, c->p.width, c->p.height, g->mb_w, g->mb_h, s->plane[0], s->plane[1], s->plane[2], s->plane[3], g->stride );
    if( inv_qscale )
        memcpy( s->inv_qscale, inv_qscale, c->n_mb * sizeof( uint16_t ) );
    else if( !c->p.aq_mode )
        for( int i = 0; i < c->n_mb; i++ ) s->inv_qscale[i] = 256;
    else
    {
        float *qp = malloc( c->n_mb * sizeof( float ) );
        uint64_t ssd;
        or8_aq_frame( luma, stride, c->p.width, c->p.height, g->mb_w, g->mb_h, cb, cr, cstride, c->p.aq_mode, c->p.aq_strength, s->inv_qscale, qp, &ssd );
        free( qp );
    }
    memset( s->intra, 0xFF, c->n_mb * sizeof( uint16_t ) ); /* frame.c:284: blocks never visited keep 0xFFFF */
    or8_intra_costs( g, s->plane[0], s->intra );
    s->valid = 1;
    return X264HIP_OK;
}

API int x264hip_frame_cost( x264hip_ctx *c, int s0, int s1, int sb, int d0, int d1, const int do_search[2], const x264hip_weight *w, int with_intra,
                            int ref1_l0_valid, x264hip_cost *out )
{
    if( !c || !out || sb < 0 || sb >= c->p.max_frames || !c->slots[sb].valid ) return X264HIP_EINVAL;
    const or_la_cfg *g = &c->cfg;
    slot_t *B = &c->slots[sb], *F0 = &c->slots[s0], *F1 = &c->slots[s1];
    const int ns = c->p.bframes + 2, idx = d0 * ns + d1;
    or_cell_out co;
    memset( &co, 0, sizeof( co ) );
    if( !B->cell[idx] && idx )
    {
        B->cell[idx] = calloc( c->n_mb, sizeof( uint16_t ) );
    }
    if( !B->rows[idx] ) B->rows[idx] = calloc( g->mb_h, sizeof( int ) );
    int *rows_i = calloc( g->mb_h, sizeof( int ) );
    if( !d0 && !d1 )
    {
        or8_cell( g, B->plane[0], NULL, NULL, 0, 128, NULL, NULL, NULL, NULL, NULL, NULL, B->intra, B->inv_qscale, !!with_intra, B->intra, B->rows[idx], rows_i, &co );
        if( with_intra ) memcpy( B->rows_intra, rows_i, g->mb_h * sizeof( int ) );
    }
    else
    {
        if( !F0->valid || ( d1 && !F1->valid ) ) { free( rows_i ); return X264HIP_ESTATE; }
        const uint8_t *r0[4] = { F0->plane[0], F0->plane[1], F0->plane[2], F0->plane[3] }, *r1[4] = { F1->plane[0], F1->plane[1], F1->plane[2], F1->plane[3] };
        if( do_search[0] )
        {
            or_weight wt = { 0, 0, 0, 0 };
            const uint8_t *wp = NULL;
            if( w && w->on )
            {
                wt.on = 1; wt.scale = w->scale; wt.denom = w->denom; wt.offset = w->offset;
                /* x264_weight_scale_plane over the padded plane (slicetype.c:493-499); the weighted copy is only read during this search */
                or8_weight_plane( B->wbuf + OR_PAD * g->stride + OR_PAD, F0->plane[0], g->stride, 8 * g->mb_w, 8 * g->mb_h, &wt );
                wp = B->wbuf + OR_PAD * g->stride + OR_PAD;
            }
            field_t *f = &B->field[0][d0 - 1];
            or8_search_field( g, B->plane[0], r0, wp, wt.on ? &wt : NULL, f->mvs, f->costs );
            f->valid = 1;
        }
        if( d1 && do_search[1] )
        {
            field_t *f = &B->field[1][d1 - 1];
            or8_search_field( g, B->plane[0], r1, NULL, NULL, f->mvs, f->costs );
            f->valid = 1;
        }
        field_t *f0 = &B->field[0][d0 - 1], *f1 = d1 ? &B->field[1][d1 - 1] : NULL;
        if( !f0->valid || ( f1 && !f1->valid ) ) { free( rows_i ); return X264HIP_ESTATE; }
        const int dsf = ( d0 * 256 + ( d0 + d1 ) / 2 ) / ( d0 + d1 );
        const int16_t ( *ref1_l0 )[2] = NULL;
        if( d1 && ref1_l0_valid )
        {
            if( !F1->field[0][d0 + d1 - 1].valid ) { free( rows_i ); return X264HIP_ESTATE; }
            ref1_l0 = (const int16_t ( * )[2])F1->field[0][d0 + d1 - 1].mvs;
        }
        or8_cell( g, B->plane[0], r0, d1 ? r1 : NULL, d1 != 0, dsf, NULL, (const int16_t ( * )[2])f0->mvs, f0->costs, f1 ? (const int16_t ( * )[2])f1->mvs : NULL,
                  f1 ? f1->costs : NULL, ref1_l0, B->intra, B->inv_qscale, !!with_intra, B->cell[idx], B->rows[idx], rows_i, &co );
        if( with_intra ) memcpy( B->rows_intra, rows_i, g->mb_h * sizeof( int ) );
    }
    free( rows_i );
    out->cost_est = co.cost_est; out->cost_est_aq = co.cost_est_aq; out->intra_mbs = co.intra_mbs;
    out->intra_cost_est = co.intra_cost_est; out->intra_cost_est_aq = co.intra_cost_est_aq;
    return X264HIP_OK;
}

API int x264hip_get_mvs( x264hip_ctx *c, int slot, int list, int dist_minus1, int16_t *mvs, int *mv_costs )
{
    if( !c || slot < 0 || slot >= c->p.max_frames || list < 0 || list > 1 || dist_minus1 < 0 || dist_minus1 > c->p.bframes || !c->slots[slot].valid ) return X264HIP_EINVAL;
    field_t *f = &c->slots[slot].field[list][dist_minus1];
    if( !f->valid ) return X264HIP_ESTATE;
    if( mvs ) memcpy( mvs, f->mvs, c->n_mb * sizeof( int16_t[2] ) );
    if( mv_costs ) memcpy( mv_costs, f->costs, c->n_mb * sizeof( int ) );
    return X264HIP_OK;
}

API int x264hip_get_lowres_costs( x264hip_ctx *c, int slot, int d0, int d1, uint16_t *costs, int *row_satds )
{
    if( !c || slot < 0 || slot >= c->p.max_frames || !c->slots[slot].valid || d0 < 0 || d1 < 0 || d0 + d1 > c->p.bframes + 1 ) return X264HIP_EINVAL;
    slot_t *s = &c->slots[slot];
    const int idx = d0 * ( c->p.bframes + 2 ) + d1;
    if( !idx )
    {
        if( costs ) memcpy( costs, s->intra, c->n_mb * sizeof( uint16_t ) );
        if( row_satds ) memcpy( row_satds, s->rows_intra, c->cfg.mb_h * sizeof( int ) );
        return X264HIP_OK;
    }
    if( !s->cell[idx] ) return X264HIP_ESTATE;
    if( costs ) memcpy( costs, s->cell[idx], c->n_mb * sizeof( uint16_t ) );
    if( row_satds ) memcpy( row_satds, s->rows[idx], c->cfg.mb_h * sizeof( int ) );
    return X264HIP_OK;
}

API int x264hip_get_intra_costs( x264hip_ctx *c, int slot, uint16_t *intra_costs )
{
    return x264hip_get_lowres_costs( c, slot, 0, 0, intra_costs, NULL );
}
