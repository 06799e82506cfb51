/* TEST INFRASTRUCTURE ONLY.
 *
 * ctypes-callable harness around the REAL reference (jpsdr/x264, C path, no asm).  This file is our
 * own code; it is compiled by oracle/build_ref.sh together with the reference's library sources
 * (taken where they lie under /root/reference) into oracle/_ref/libx264ref{8,10}.so.
 *
 * It includes the reference's encoder/analyse.c as a translation unit (which itself pulls in
 * encoder/slicetype.c, analyse.c:3894) so that the static functions slicetype_frame_cost /
 * lowres_context_init are reachable without touching the reference tree.
 *
 * Only tests/ and bench.py's cpu_baseline leg load the resulting library.  Nothing in the product
 * (x264_amd/) links, loads or calls it.
 */
#include "encoder/analyse.c"

#include <time.h>

/* the templated entry point has no prototype in the reference headers (api.c:34 declares it locally) */
x264_t *x264_encoder_open( x264_param_t *, void * );

#define RH_API __attribute__((visibility("default"), force_align_arg_pointer))
#define RH_MAX_FRAMES 512

typedef struct
{
    x264_t *h;
    x264_param_t param;
    int n_frames;
    x264_frame_t *frames[RH_MAX_FRAMES + 4];
    int n_refs;
    x264_frame_t *refs[16];       /* reconstructed-frame style references (half-pel planes + integral) for rh_me_search */
} rh_ctx;

static void rh_log_quiet( void *p, int level, const char *fmt, va_list ap )
{
    if( level <= X264_LOG_ERROR )
        vfprintf( stderr, fmt, ap );
}

RH_API int rh_bit_depth( void ) { return BIT_DEPTH; }

/* opts: "key=value,key=value" applied with x264_param_parse after the preset. */
RH_API rh_ctx *rh_open( int width, int height, const char *preset, const char *tune, const char *opts )
{
    rh_ctx *c = calloc( 1, sizeof(*c) );
    if( !c )
        return NULL;
    if( x264_param_default_preset( &c->param, preset && preset[0] ? preset : "medium", tune && tune[0] ? tune : NULL ) < 0 )
        goto fail;
    c->param.i_width = width;
    c->param.i_height = height;
    c->param.i_csp = X264_CSP_I420;
    c->param.i_bitdepth = BIT_DEPTH;
    c->param.i_threads = 1;
    c->param.i_lookahead_threads = 1;
    c->param.b_vfr_input = 0;
    c->param.i_fps_num = 25;
    c->param.i_fps_den = 1;
    c->param.pf_log = rh_log_quiet;
    c->param.i_log_level = X264_LOG_ERROR;
    if( opts && opts[0] )
    {
        char *dup = strdup( opts ), *save = NULL;
        for( char *tok = strtok_r( dup, ",", &save ); tok; tok = strtok_r( NULL, ",", &save ) )
        {
            char *eq = strchr( tok, '=' );
            if( eq ) *eq = 0;
            if( !strcmp( tok, "vfr-input" ) ) /* a field of x264_param_t without an option name (x264.h: b_vfr_input) */
            {
                c->param.b_vfr_input = eq ? atoi( eq + 1 ) : 1;
                continue;
            }
            if( !strcmp( tok, "csp" ) && eq ) /* i_csp is set by the CLI from --output-csp, not by x264_param_parse */
            {
                c->param.i_csp = !strcmp( eq + 1, "i444" ) ? X264_CSP_I444 : !strcmp( eq + 1, "i422" ) ? X264_CSP_I422 : X264_CSP_I420;
                continue;
            }
            if( !strcmp( tok, "timebase" ) && eq ) /* i_timebase_num / i_timebase_den: set by the CLI, not an option of x264_param_parse */
            {
                unsigned a = 0, b = 0;
                if( sscanf( eq + 1, "%u/%u", &a, &b ) == 2 ) { c->param.i_timebase_num = a; c->param.i_timebase_den = b; }
                continue;
            }
            if( x264_param_parse( &c->param, tok, eq ? eq + 1 : NULL ) < 0 )
            {
                fprintf( stderr, "rh_open: bad option %s\n", tok );
                free( dup );
                goto fail;
            }
        }
        free( dup );
    }
    c->h = x264_encoder_open( &c->param, NULL );
    if( !c->h )
        goto fail;
    return c;
fail:
    free( c );
    return NULL;
}

RH_API void rh_close( rh_ctx *c )
{
    if( !c ) return;
    for( int i = 0; i < c->n_frames; i++ )
        if( c->frames[i] )
            x264_frame_push_unused( c->h, c->frames[i] );
    for( int i = 0; i < c->n_refs; i++ )
        if( c->refs[i] )
            x264_frame_push_unused( c->h, c->refs[i] );
    x264_encoder_close( c->h );
    free( c );
}

/* effective (validated) parameters, for mirroring into our own context */
RH_API void rh_get_config( rh_ctx *c, int *out )
{
    x264_t *h = c->h;
    x264_mb_analysis_t a;
    lowres_context_init( h, &a );
    out[0]  = h->mb.i_mb_width;
    out[1]  = h->mb.i_mb_height;
    out[2]  = h->param.i_bframe;
    out[3]  = h->param.i_bframe_adaptive;
    out[4]  = h->param.rc.i_lookahead;
    out[5]  = h->param.analyse.i_mv_range;
    out[6]  = h->param.analyse.i_me_range;
    out[7]  = h->mb.i_me_method;
    out[8]  = h->mb.i_subpel_refine;
    out[9]  = a.i_lambda;
    out[10] = h->param.analyse.i_weighted_pred;
    out[11] = h->param.analyse.b_weighted_bipred;
    out[12] = h->param.rc.i_aq_mode;
    out[13] = h->param.rc.b_mb_tree;
    out[14] = h->param.i_scenecut_threshold;
    out[15] = h->param.i_keyint_max;
    out[16] = h->param.i_keyint_min;
    out[17] = h->param.i_bframe_pyramid;
    out[18] = h->param.i_bframe_bias;
    out[19] = h->frames.i_delay;
    out[20] = h->lookahead->i_slicetype_length;
    out[21] = h->param.rc.i_vbv_buffer_size;
    out[22] = h->param.b_open_gop;
    out[23] = h->param.b_intra_refresh;
    out[24] = h->param.analyse.i_subpel_refine;
    out[25] = h->pixf.mbcmp[0] == h->pixf.satd[0];     /* mbcmp is SATD */
    out[26] = h->pixf.fpelcmp[0] == h->pixf.satd[0];   /* fpelcmp is SATD */
    out[27] = h->param.analyse.b_psy;
    out[28] = h->param.rc.i_rc_method;
    out[29] = h->frames.i_bframe_delay;
    out[30] = h->param.i_frame_reference;
    out[31] = (int)(h->param.rc.f_aq_strength * 65536.f);
    out[32] = h->param.i_lookahead_threads;
    out[33] = h->param.i_threads;
}

/* cost_mv table of the lookahead qp, centred: out[i + n] for i in [-n, n], n = 2*4*mv_range */
RH_API int rh_cost_mv( rh_ctx *c, uint16_t *out, int cap )
{
    x264_t *h = c->h;
    int n = 2*4*h->param.analyse.i_mv_range;
    if( out )
    {
        if( cap < 2*n+1 ) return -1;
        for( int i = -n; i <= n; i++ )
            out[i+n] = h->cost_mv[X264_LOOKAHEAD_QP][i];
    }
    return n;
}

static x264_frame_t *rh_make_frame( rh_ctx *c, const pixel *y, const pixel *u, const pixel *v, int idx )
{
    x264_t *h = c->h;
    int w = h->param.i_width, ht = h->param.i_height;
    x264_picture_t pic;
    x264_picture_init( &pic );
    /* planar input in the colour space the context was opened with ("csp=i422" / "csp=i444" in the option string; default I420) */
    int csp = h->param.i_csp & X264_CSP_MASK;
    int cw = csp == X264_CSP_I444 ? w : (w+1)/2, chh = csp == X264_CSP_I420 ? (ht+1)/2 : ht;
    pic.img.i_csp = csp | (BIT_DEPTH > 8 ? X264_CSP_HIGH_DEPTH : 0);
    pic.img.i_plane = 3;
    pixel *grey = NULL;
    if( !u || !v )
    {
        grey = malloc( (size_t)(cw+1)*(chh+1)*sizeof(pixel) );
        for( int i = 0; i < (cw+1)*(chh+1); i++ )
            grey[i] = 1 << (BIT_DEPTH-1);
        u = v = grey;
    }
    pic.img.plane[0] = (uint8_t*)y; pic.img.i_stride[0] = w * sizeof(pixel);
    pic.img.plane[1] = (uint8_t*)u; pic.img.i_stride[1] = cw * sizeof(pixel);
    pic.img.plane[2] = (uint8_t*)v; pic.img.i_stride[2] = cw * sizeof(pixel);
    pic.i_pts = idx;
    x264_frame_t *f = x264_frame_pop_unused( h, 0 );
    if( !f || x264_frame_copy_picture( h, f, &pic ) < 0 )
    {
        free( grey );
        return NULL;
    }
    free( grey );
    if( w != 16*h->mb.i_mb_width || ht != 16*h->mb.i_mb_height )
        x264_frame_expand_border_mod16( h, f );
    f->i_frame = idx;
    f->i_pic_struct = PIC_STRUCT_PROGRESSIVE;
    return f;
}

static const float *rh_quant_offsets = NULL; /* optional [n_frames][mb_count] x264_picture_t.prop.quant_offsets (rh_add_frame: one frame's worth) */
RH_API void rh_set_quant_offsets( const float *q ) { rh_quant_offsets = q; }

/* Add a frame to the evaluation set (index = order of addition): AQ + lowres init, exactly the
 * pre-lookahead steps of x264_encoder_encode (encoder.c:3368-3423). */
RH_API int rh_add_frame( rh_ctx *c, const pixel *y, const pixel *u, const pixel *v )
{
    if( c->n_frames >= RH_MAX_FRAMES ) return -1;
    x264_frame_t *f = rh_make_frame( c, y, u, v, c->n_frames );
    if( !f ) return -1;
    x264_adaptive_quant_frame( c->h, f, (float*)rh_quant_offsets );
    x264_frame_init_lowres( c->h, f );
    c->frames[c->n_frames] = f;
    return c->n_frames++;
}

/* ---- main-encode motion search: x264_me_search_ref with any method on a full-resolution reference ----
 * A reference frame as the encoder keeps it after reconstruction: luma plane with expanded borders, the three half-pel
 * planes (x264_frame_filter -> mc.hpel_filter) and, when the context was opened with me=esa/tesa, the integral planes. */
RH_API int rh_add_ref_frame( rh_ctx *c, const pixel *y )
{
    x264_t *h = c->h;
    if( c->n_refs >= 16 ) return -1;
    x264_frame_t *f = x264_frame_pop_unused( h, 1 );
    if( !f ) return -1;
    int w = h->param.i_width, ht = h->param.i_height;
    for( int r = 0; r < 16*h->mb.i_mb_height; r++ )
    {
        const pixel *src = y + (size_t)X264_MIN( r, ht-1 ) * w;
        pixel *dst = f->plane[0] + (size_t)r * f->i_stride[0];
        for( int x = 0; x < 16*h->mb.i_mb_width; x++ )
            dst[x] = src[X264_MIN( x, w-1 )];
    }
    for( int r = 0; r < 8*h->mb.i_mb_height; r++ )
        for( int x = 0; x < 16*h->mb.i_mb_width; x++ )
            f->plane[1][(size_t)r * f->i_stride[1] + x] = 1 << (BIT_DEPTH-1);
    h->i_threadslice_start = 0;
    h->i_threadslice_end = h->mb.i_mb_height;
    for( int mb_y = 1; mb_y <= h->mb.i_mb_height; mb_y++ )
    {
        int min_y = mb_y - 1, end = mb_y == h->mb.i_mb_height;   /* the order of fdec_filter_row, encoder.c:2471-2479 */
        x264_frame_expand_border( h, f, min_y );
        x264_frame_filter( h, f, min_y, end );
        x264_frame_expand_border_filtered( h, f, min_y, end );
    }
    c->refs[c->n_refs] = f;
    return c->n_refs++;
}

/* fenc: the bw x bh source block in FENC layout (stride 16).  Limits as mb_analyse_init sets them (analyse.c:330-396,
 * single frame thread).  out: mvx, mvy, cost, cost_mv. */
RH_API int rh_me_search( rh_ctx *c, int ref, const pixel *fenc, int mb_x, int mb_y, int xoff, int yoff, int i_pixel, int subpel_refine,
                         int me_range, const int16_t *mvp, const int16_t *mvc_in, int n_mvc, int *out )
{
    x264_t *h = c->h;
    if( ref < 0 || ref >= c->n_refs ) return -1;
    x264_frame_t *f = c->refs[ref];
    ALIGNED_ARRAY_64( pixel, fenc_buf,[16*16] );
    ALIGNED_ARRAY_8( int16_t, mvc,[16],[2] );
    memcpy( fenc_buf, fenc, sizeof(fenc_buf) );
    for( int i = 0; i < n_mvc && i < 16; i++ ) { mvc[i][0] = mvc_in[2*i]; mvc[i][1] = mvc_in[2*i+1]; }
    h->mb.i_mb_x = mb_x; h->mb.i_mb_y = mb_y;
    h->mb.i_qp = X264_LOOKAHEAD_QP;
    h->fenc = c->n_frames ? c->frames[0] : f;    /* ESA reads h->fenc->i_lines[0] (me.c:641) */
    const int i_fmv_range = 4 * h->param.analyse.i_mv_range, i_fpel_border = 6;
    h->mb.mv_min[0] = 4*( -16*mb_x - 24 );
    h->mb.mv_max[0] = 4*( 16*( h->mb.i_mb_width - mb_x - 1 ) + 24 );
    h->mb.mv_min_spel[0] = X264_MAX( h->mb.mv_min[0], -i_fmv_range );
    h->mb.mv_max_spel[0] = X264_MIN( h->mb.mv_max[0], i_fmv_range-1 );
    h->mb.mv_limit_fpel[0][0] = (h->mb.mv_min_spel[0]>>2) + i_fpel_border;
    h->mb.mv_limit_fpel[1][0] = (h->mb.mv_max_spel[0]>>2) - i_fpel_border;
    h->mb.mv_min[1] = 4*( -16*mb_y - 24 );
    h->mb.mv_max[1] = 4*( 16*( h->mb.i_mb_height - mb_y - 1 ) + 24 );
    h->mb.mv_min_spel[1] = X264_MAX( h->mb.mv_min[1], -i_fmv_range );
    h->mb.mv_max_spel[1] = X264_MIN( h->mb.mv_max[1], i_fmv_range-1 );
    h->mb.mv_limit_fpel[0][1] = (h->mb.mv_min_spel[1]>>2) + i_fpel_border;
    h->mb.mv_limit_fpel[1][1] = (h->mb.mv_max_spel[1]>>2) - i_fpel_border;
    h->mb.i_me_method = h->param.analyse.i_me_method;
    h->mb.i_subpel_refine = subpel_refine;
    h->mb.b_chroma_me = 0;
    const int saved_range = h->param.analyse.i_me_range;
    h->param.analyse.i_me_range = me_range;
    x264_me_t m;
    memset( &m, 0, sizeof(m) );
    m.i_pixel = i_pixel;
    m.p_cost_mv = h->cost_mv[X264_LOOKAHEAD_QP];
    m.i_ref_cost = 0;
    m.i_ref = 0;
    m.weight = x264_weight_none;
    m.p_fenc[0] = fenc_buf;
    m.i_stride[0] = f->i_stride[0];
    const intptr_t off = 16*mb_x + xoff + ( 16*mb_y + yoff ) * (intptr_t)f->i_stride[0];
    for( int i = 0; i < 4; i++ )
        m.p_fref[i] = f->filtered[0][i] + off;
    m.p_fref_w = m.p_fref[0];
    m.integral = f->integral ? f->integral + off : NULL;
    m.mvp[0] = mvp[0]; m.mvp[1] = mvp[1];
    x264_me_search_ref( h, &m, mvc, n_mvc, NULL );
    h->param.analyse.i_me_range = saved_range;
    out[0] = m.mv[0]; out[1] = m.mv[1]; out[2] = m.cost; out[3] = m.cost_mv;
    return 0;
}

/* the planes rh_me_search reads, for the checker: plane p (0 full, 1 H, 2 V, 3 HV) incl. PADH/PADV borders, tight */
RH_API int rh_ref_geometry( rh_ctx *c, int *out )
{
    if( !c->n_refs ) return -1;
    x264_frame_t *f = c->refs[0];
    out[0] = f->i_width[0]; out[1] = f->i_lines[0]; out[2] = f->i_stride[0]; out[3] = PADH; out[4] = PADV; out[5] = f->integral != NULL; out[6] = PADH_ALIGN; out[7] = c->h->frames.b_have_sub8x8_esa;
    return 0;
}
RH_API void rh_get_ref_plane( rh_ctx *c, int ref, int p, pixel *out )
{
    x264_frame_t *f = c->refs[ref];
    int w = f->i_width[0] + 2*PADH, hh = f->i_lines[0] + 2*PADV;
    for( int y = 0; y < hh; y++ )
        memcpy( out + (size_t)y*w, f->filtered[0][p] + (y-PADV)*(intptr_t)f->i_stride[0] - PADH, w * sizeof(pixel) );
}
/* integral planes as x264_frame_filter leaves them: (lines + 2*PADV) rows x stride, upper (8x8 sums) then lower (4x4 sums) */
RH_API int rh_get_integral( rh_ctx *c, int ref, uint16_t *out, int cap )
{
    x264_frame_t *f = c->refs[ref];
    if( !f->integral ) return -1;
    int rows = ( f->i_lines[0] + 2*PADV ) * 2, stride = f->i_stride[0];
    if( cap < rows * stride ) return -1;
    memcpy( out, f->integral - PADV*stride - PADH_ALIGN, (size_t)rows * stride * sizeof(uint16_t) );
    return rows * stride;
}
/* Direct call of the reference's static slicetype_frame_cost (slicetype.c:836). */
RH_API int rh_frame_cost( rh_ctx *c, int p0, int p1, int b )
{
    x264_mb_analysis_t a;
    lowres_context_init( c->h, &a );
    return slicetype_frame_cost( c->h, &a, c->frames, p0, p1, b );
}

/* slicetype_frame_cost_recalculate (slicetype.c:999) of an evaluated cell; b_type_b picks f_qp_offset_aq like a B frame */
RH_API int rh_frame_cost_recalculate( rh_ctx *c, int p0, int p1, int b, int b_type_b )
{
    c->frames[b]->i_type = b_type_b ? X264_TYPE_B : X264_TYPE_P;
    return slicetype_frame_cost_recalculate( c->h, c->frames, p0, p1, b );
}
RH_API void rh_get_qp_offsets( rh_ctx *c, int idx, float *qp, float *qp_aq )
{
    x264_frame_t *f = c->frames[idx];
    int n = c->h->mb.i_mb_count;
    if( qp ) memcpy( qp, f->f_qp_offset, n * sizeof(float) );
    if( qp_aq ) memcpy( qp_aq, f->f_qp_offset_aq, n * sizeof(float) );
}
RH_API void rh_set_qp_offsets( rh_ctx *c, int idx, const float *qp )
{
    memcpy( c->frames[idx]->f_qp_offset, qp, c->h->mb.i_mb_count * sizeof(float) );
}

RH_API void rh_weights_analyse( rh_ctx *c, int b, int ref, int *out )
{
    x264_weights_analyse( c->h, c->frames[b], c->frames[ref], 1 );
    x264_weight_t *w = c->frames[b]->weight[0];
    out[0] = w[0].weightfn != NULL; out[1] = w[0].i_scale; out[2] = w[0].i_denom; out[3] = w[0].i_offset;
}

/* Field accessors: copy internal arrays of frame idx into caller buffers. */
RH_API int rh_lowres_geometry( rh_ctx *c, int *out )
{
    x264_frame_t *f = c->frames[0];
    if( !f ) return -1;
    out[0] = f->i_width_lowres; out[1] = f->i_lines_lowres; out[2] = f->i_stride_lowres;
    out[3] = PADH; out[4] = PADV;
    return 0;
}

/* plane p of the lowres pyramid including the padded border: (lines+2*PADV) x (width+2*PADH), tight */
RH_API void rh_get_lowres( rh_ctx *c, int idx, int p, pixel *out )
{
    x264_frame_t *f = c->frames[idx];
    int w = f->i_width_lowres + 2*PADH, hh = f->i_lines_lowres + 2*PADV;
    for( int y = 0; y < hh; y++ )
        memcpy( out + (size_t)y*w, f->lowres[p] + (y-PADV)*f->i_stride_lowres - PADH, w*sizeof(pixel) );
}

RH_API void rh_get_frame_stats( rh_ctx *c, int idx, uint16_t *inv_qscale, uint16_t *intra_cost, uint64_t *sum_ssd )
{
    x264_frame_t *f = c->frames[idx];
    int n = c->h->mb.i_mb_count;
    if( inv_qscale ) memcpy( inv_qscale, f->i_inv_qscale_factor, n*sizeof(uint16_t) );
    if( intra_cost ) memcpy( intra_cost, f->i_intra_cost, n*sizeof(uint16_t) );
    if( sum_ssd ) { sum_ssd[0] = f->i_pixel_sum[0]; sum_ssd[1] = f->i_pixel_ssd[0]; }
}

RH_API int rh_get_mvs( rh_ctx *c, int idx, int list, int dist, int16_t *mvs, int *mv_costs )
{
    x264_frame_t *f = c->frames[idx];
    int n = c->h->mb.i_mb_count;
    if( !f->lowres_mvs[list][dist] ) return -1;
    memcpy( mvs, f->lowres_mvs[list][dist], n*2*sizeof(int16_t) );
    if( mv_costs ) memcpy( mv_costs, f->lowres_mv_costs[list][dist], n*sizeof(int) );
    return 0;
}

RH_API int rh_get_cell( rh_ctx *c, int idx, int d0, int d1, uint16_t *lowres_costs, int *row_satds, int *summary )
{
    x264_frame_t *f = c->frames[idx];
    int n = c->h->mb.i_mb_count;
    if( !f->lowres_costs[d0][d1] ) return -1;
    if( lowres_costs ) memcpy( lowres_costs, f->lowres_costs[d0][d1], n*sizeof(uint16_t) );
    if( row_satds ) memcpy( row_satds, f->i_row_satds[d0][d1], c->h->mb.i_mb_height*sizeof(int) );
    summary[0] = f->i_cost_est[d0][d1];
    summary[1] = f->i_cost_est_aq[d0][d1];
    summary[2] = f->i_intra_mbs[d0];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Lookahead-only driver: the front half of x264_encoder_encode (encoder.c:3368-3454) paced exactly
 * like the encoder (one frame consumed per call once the delay is filled), without the main encode.
 * Per output frame (coded order) records: display index, type, and the frame's whole i_cost_est /
 * i_cost_est_aq matrices + i_intra_mbs as they stand when the frame leaves the lookahead.
 * Returns the number of frames output; *seconds = wall time spent inside (lowres init + lookahead).
 * ------------------------------------------------------------------------------------------ */
#define RH_MAT ((X264_BFRAME_MAX+2)*(X264_BFRAME_MAX+2))

static float *rh_out_qp = NULL; /* optional [n_frames][mb_count] dump of f_qp_offset at the moment a frame leaves the lookahead */

static uint16_t *rh_out_prop = NULL; /* optional [n_frames][mb_count] dump of i_propagate_cost */
RH_API void rh_set_qp_dump( float *buf ) { rh_out_qp = buf; }
RH_API void rh_set_prop_dump( uint16_t *buf ) { rh_out_prop = buf; }
static int *rh_out_planned_type = NULL, *rh_out_planned_satd = NULL; /* optional [n_frames][X264_LOOKAHEAD_MAX+1] */
static int *rh_out_rows = NULL; /* optional [n_frames][(X264_BFRAME_MAX+2)^2][mb_h] dump of i_row_satds (cells that are not allocated stay untouched) */
RH_API void rh_set_vbv_dump( int *planned_type, int *planned_satd, int *rows ) { rh_out_planned_type = planned_type; rh_out_planned_satd = planned_satd; rh_out_rows = rows; }
/* optional: run the real x264_rc_analyse_slice (slicetype.c:1976-2032) on every frame as it leaves, the way rate control does when
 * the frame gets encoded.  For B frames the caller supplies the distances to the nearest references (cells[k] = { b-p0, p1-b } of
 * output k; what fref_nearest gives in the encoder); out[k] = { returned cost, i_row_satd[mb_h], i_row_satds[0][0][mb_h] }. */
static const int *rh_rc_cells = NULL;
static int *rh_out_rc = NULL;
RH_API void rh_set_rc_dump( const int *cells, int *out ) { rh_rc_cells = cells; rh_out_rc = out; }
static const int64_t *rh_pts = NULL; /* optional [n_frames] x264_picture_t.i_pts (default: the frame index) */
RH_API void rh_set_pts( const int64_t *pts ) { rh_pts = pts; }
static const int *rh_forced_types = NULL; /* optional [n_frames] x264_picture_t.i_type of every input picture (x264.h:274-280) */
RH_API void rh_set_forced_types( const int *types ) { rh_forced_types = types; }

static int rh_drain_one( x264_t *h, int *out_idx, int *out_type, int *out_cost, int *out_cost_aq, int *out_imbs, int n_out )
{
    h->i_frame++;
    if( !h->frames.current[0] )
        x264_lookahead_get_frames( h );
    if( !h->frames.current[0] && x264_lookahead_is_empty( h ) )
        return -1;
    x264_frame_t *f = x264_frame_shift( h->frames.current );
    if( out_idx )  out_idx[n_out] = f->i_frame;
    if( out_type ) out_type[n_out] = f->i_type;
    if( out_cost )    memcpy( out_cost    + (size_t)n_out*RH_MAT, f->i_cost_est,    RH_MAT*sizeof(int) );
    if( out_cost_aq ) memcpy( out_cost_aq + (size_t)n_out*RH_MAT, f->i_cost_est_aq, RH_MAT*sizeof(int) );
    if( out_imbs )    memcpy( out_imbs + (size_t)n_out*(X264_BFRAME_MAX+2), f->i_intra_mbs, (X264_BFRAME_MAX+2)*sizeof(int) );
    /* (the AQ / MB-tree arrays only exist with aq-mode != 0, frame.c:286-301: the dump stays zero without them) */
    if( rh_out_qp && f->f_qp_offset )   memcpy( rh_out_qp + (size_t)n_out*h->mb.i_mb_count, f->f_qp_offset, h->mb.i_mb_count*sizeof(float) );
    if( rh_out_prop && f->i_propagate_cost ) memcpy( rh_out_prop + (size_t)n_out*h->mb.i_mb_count, f->i_propagate_cost, h->mb.i_mb_count*sizeof(uint16_t) );
    if( rh_out_planned_type )
        for( int j = 0; j <= X264_LOOKAHEAD_MAX; j++ )
            rh_out_planned_type[(size_t)n_out*(X264_LOOKAHEAD_MAX+1) + j] = f->i_planned_type[j]; /* uint8_t there */
    if( rh_out_planned_satd ) memcpy( rh_out_planned_satd + (size_t)n_out*(X264_LOOKAHEAD_MAX+1), f->i_planned_satd, (X264_LOOKAHEAD_MAX+1)*sizeof(int) );
    if( rh_out_rows && h->frames.b_have_lowres )
        for( int i = 0; i <= h->param.i_bframe+1; i++ )
            for( int j = 0; j <= h->param.i_bframe+1; j++ )
                memcpy( rh_out_rows + ((size_t)n_out*(X264_BFRAME_MAX+2)*(X264_BFRAME_MAX+2) + i*(X264_BFRAME_MAX+2) + j)*h->mb.i_mb_height,
                        f->i_row_satds[i][j], h->mb.i_mb_height*sizeof(int) );
    /* (rate control analyses B frames only with VBV, ratecontrol.c:2472-2474; without it their cell may not even exist) */
    if( rh_out_rc && h->param.rc.i_rc_method != X264_RC_CQP && ( !IS_X264_TYPE_B( f->i_type ) || h->param.rc.i_vbv_buffer_size ) )
    {
        static x264_frame_t near0, near1;
        int mbh = h->mb.i_mb_height;
        int *o = rh_out_rc + (size_t)n_out*(1 + 2*mbh);
        x264_frame_t *fenc_bak = h->fenc, *fdec_bak = h->fdec;
        h->fenc = f; h->fdec = f;
        if( IS_X264_TYPE_B( f->i_type ) )
        {
            near0.i_poc = 0;
            f->i_poc = 2*rh_rc_cells[2*n_out];
            near1.i_poc = 2*(rh_rc_cells[2*n_out] + rh_rc_cells[2*n_out+1]);
            h->fref_nearest[0] = &near0; h->fref_nearest[1] = &near1;
        }
        o[0] = x264_rc_analyse_slice( h );
        memcpy( o + 1, f->i_row_satd, mbh*sizeof(int) );
        memcpy( o + 1 + mbh, f->i_row_satds[0][0], mbh*sizeof(int) );
        h->fenc = fenc_bak; h->fdec = fdec_bak;
    }
    x264_frame_push_unused( h, f );
    return 0;
}

RH_API int rh_lookahead_run( rh_ctx *c, const pixel *yuv, int n_frames, int luma_only,
                             int *out_idx, int *out_type, int *out_cost, int *out_cost_aq, int *out_imbs,
                             double *seconds, double *seconds_prep )
{
    x264_t *h = c->h;
    int w = h->param.i_width, ht = h->param.i_height;
    int csp_ = h->param.i_csp & X264_CSP_MASK;
    size_t ysz = (size_t)w*ht, csz = (size_t)(csp_ == X264_CSP_I444 ? w : (w+1)/2)*(csp_ == X264_CSP_I420 ? (ht+1)/2 : ht);
    size_t fsz = luma_only ? ysz : ysz + 2*csz;
    int n_out = 0;
    double t_la = 0, t_prep = 0;
    struct timespec t0, t1;
    for( int i = 0; i < n_frames; i++ )
    {
        const pixel *y = yuv + (size_t)i*fsz;
        clock_gettime( CLOCK_MONOTONIC, &t0 );
        x264_frame_t *fenc = rh_make_frame( c, y, luma_only ? NULL : y+ysz, luma_only ? NULL : y+ysz+csz, 0 );
        if( !fenc ) return -1;
        fenc->i_frame = h->frames.i_input++;
        fenc->i_pts = rh_pts ? rh_pts[i] : fenc->i_frame;
        if( rh_forced_types ) /* x264_frame_copy_picture, frame.c:392-400 */
            fenc->i_type = fenc->i_forced_type = rh_forced_types[i] < X264_TYPE_AUTO || rh_forced_types[i] > X264_TYPE_KEYFRAME ? X264_TYPE_AUTO : rh_forced_types[i];
        if( fenc->i_frame == 0 )
            h->frames.i_first_pts = fenc->i_pts;
        if( h->frames.i_bframe_delay && fenc->i_frame == h->frames.i_bframe_delay )
            h->frames.i_bframe_delay_time = fenc->i_pts - h->frames.i_first_pts;
        h->frames.i_second_largest_pts = h->frames.i_largest_pts;
        h->frames.i_largest_pts = fenc->i_pts;
        x264_adaptive_quant_frame( h, fenc, rh_quant_offsets ? (float*)rh_quant_offsets + (size_t)i*h->mb.i_mb_count : NULL );
        clock_gettime( CLOCK_MONOTONIC, &t1 );
        t_prep += (t1.tv_sec-t0.tv_sec) + 1e-9*(t1.tv_nsec-t0.tv_nsec);
        clock_gettime( CLOCK_MONOTONIC, &t0 );
        if( h->frames.b_have_lowres ) /* encoder.c:3422-3423 */
            x264_frame_init_lowres( h, fenc );
        x264_lookahead_put_frame( h, fenc );
        if( h->frames.i_input > h->frames.i_delay + 1 - h->i_thread_frames )
        {
            if( rh_drain_one( h, out_idx, out_type, out_cost, out_cost_aq, out_imbs, n_out ) == 0 )
                n_out++;
        }
        clock_gettime( CLOCK_MONOTONIC, &t1 );
        t_la += (t1.tv_sec-t0.tv_sec) + 1e-9*(t1.tv_nsec-t0.tv_nsec);
    }
    clock_gettime( CLOCK_MONOTONIC, &t0 );
    while( n_out < n_frames && rh_drain_one( h, out_idx, out_type, out_cost, out_cost_aq, out_imbs, n_out ) == 0 )
        n_out++;
    clock_gettime( CLOCK_MONOTONIC, &t1 );
    t_la += (t1.tv_sec-t0.tv_sec) + 1e-9*(t1.tv_nsec-t0.tv_nsec);
    if( seconds ) *seconds = t_la;
    if( seconds_prep ) *seconds_prep = t_prep;
    return n_out;
}

/* ------------------------------------------------------------------------------------------
 * A whole x264_encoder_encode run (encoder.c:3368-3740), the reference's own control flow end to end.
 * Per coded frame, read off h->fenc right after the call that coded it (i_threads == 1: the frame object stays intact until a later
 * call reuses it): display number, slice type, every cost cell, and CRC-32s of the per-block maps the lookahead left in it --
 * [0] lowres_costs of every evaluated cell (blocks slicetype_slice_cost visits, slicetype.c:823-833), [1] lowres_mvs of every searched
 * field, [2] lowres_mv_costs of the same (visited blocks), [3] f_qp_offset, [4] i_row_satd as rate control was given it
 * (x264_rc_analyse_slice, slicetype.c:1976-2013), [5] i_planned_satd, [6] i_planned_type (vbv_lookahead), [7] i_intra_mbs.  stream_crc: CRC-32 of every NAL unit except SEI (the
 * version SEI spells out the option list, "opencl=1" included, encoder/set.c + common/base.c:1446-1447).
 * Used to compare the plain C run with the run whose slicetype_frame_cost goes through the accelerator hook (opts "opencl=1" in the
 * build where x264_amd/csrc/slicetype_hip.c stands in for encoder/slicetype-cl.c): everything must be identical.
 * ------------------------------------------------------------------------------------------ */
static uint32_t rh_crc32( uint32_t crc, const void *data, size_t n )
{
    static uint32_t T[256];
    if( !T[1] )
        for( uint32_t i = 0; i < 256; i++ )
        {
            uint32_t v = i;
            for( int k = 0; k < 8; k++ ) v = v & 1 ? 0xEDB88320u ^ ( v >> 1 ) : v >> 1;
            T[i] = v;
        }
    const uint8_t *p = data;
    crc = ~crc;
    while( n-- ) crc = T[( crc ^ *p++ ) & 255] ^ ( crc >> 8 );
    return ~crc;
}

RH_API int rh_accel_state( rh_ctx *c )
{
    /* 1 = the accelerator hook is in use, 0 = it was never asked for or the encoder fell back to the C path, -1 = it failed mid-way */
#if HAVE_OPENCL
    if( c->h->opencl.b_fatal_error ) return -1;
#endif
    return c->h->param.b_opencl;
}

#define RH_ENC_CRCS 8
RH_API int rh_encode_run( rh_ctx *c, const pixel *yuv, int n_frames, int luma_only, int *out_frame, int *out_type, int *out_cost, int *out_cost_aq,
                          uint32_t *out_map_crc, int *out_bytes, uint32_t *stream_crc, double *seconds )
{
    x264_t *h = c->h;
    int w = h->param.i_width, ht = h->param.i_height;
    int csp_ = h->param.i_csp & X264_CSP_MASK;
    size_t ysz = (size_t)w*ht, cw = csp_ == X264_CSP_I444 ? w : (w+1)/2, chh = csp_ == X264_CSP_I420 ? (ht+1)/2 : ht, csz = cw*chh;
    size_t fsz = luma_only ? ysz : ysz + 2*csz;
    pixel *grey = NULL;
    if( luma_only )
    {
        grey = malloc( csz * sizeof(pixel) );
        if( !grey ) return -1;
        for( size_t i = 0; i < csz; i++ ) grey[i] = 1 << (BIT_DEPTH-1);
    }
    int do_edges = h->param.rc.b_mb_tree || h->param.rc.i_vbv_buffer_size || h->mb.i_mb_width <= 2 || h->mb.i_mb_height <= 2;
    int mbw = h->mb.i_mb_width, mbh = h->mb.i_mb_height, n_mb = h->mb.i_mb_count;
    uint32_t crc_stream = 0;
    int n_out = 0;
    struct timespec t0, t1;
    clock_gettime( CLOCK_MONOTONIC, &t0 );
    for( int i = 0; ; i++ )
    {
        x264_picture_t pic, pic_out;
        x264_nal_t *nal;
        int i_nal = 0, size;
        if( i < n_frames )
        {
            const pixel *y = yuv + (size_t)i*fsz;
            x264_picture_init( &pic );
            pic.img.i_csp = h->param.i_csp;
            pic.img.i_plane = 3;
            pic.img.plane[0] = (uint8_t*)y; pic.img.i_stride[0] = w * SIZEOF_PIXEL;
            pic.img.plane[1] = (uint8_t*)( luma_only ? grey : y + ysz ); pic.img.i_stride[1] = cw * SIZEOF_PIXEL;
            pic.img.plane[2] = (uint8_t*)( luma_only ? grey : y + ysz + csz ); pic.img.i_stride[2] = cw * SIZEOF_PIXEL;
            pic.i_pts = rh_pts ? rh_pts[i] : i;
            pic.i_type = rh_forced_types ? rh_forced_types[i] : X264_TYPE_AUTO;
            size = x264_encoder_encode( h, &nal, &i_nal, &pic, &pic_out );
        }
        else
        {
            if( !x264_encoder_delayed_frames( h ) ) break;
            size = x264_encoder_encode( h, &nal, &i_nal, NULL, &pic_out );
        }
        if( size < 0 ) { free( grey ); return -2; }
        if( !size ) continue;
        size = 0; /* (counted without SEI as well) */
        for( int k = 0; k < i_nal; k++ )
            if( nal[k].i_type != NAL_SEI )
            {
                crc_stream = rh_crc32( crc_stream, nal[k].p_payload, nal[k].i_payload );
                size += nal[k].i_payload;
            }
        if( n_out >= n_frames ) { free( grey ); return -3; }
        x264_frame_t *f = h->fenc;
        if( out_frame ) out_frame[n_out] = f->i_frame;
        if( out_type )  out_type[n_out] = f->i_type;
        if( out_bytes ) out_bytes[n_out] = size;
        if( out_cost )    memcpy( out_cost    + (size_t)n_out*RH_MAT, f->i_cost_est,    RH_MAT*sizeof(int) );
        if( out_cost_aq ) memcpy( out_cost_aq + (size_t)n_out*RH_MAT, f->i_cost_est_aq, RH_MAT*sizeof(int) );
        if( out_map_crc )
        {
            uint32_t *o = out_map_crc + (size_t)n_out*RH_ENC_CRCS;
            memset( o, 0, RH_ENC_CRCS * sizeof(uint32_t) );
            if( h->frames.b_have_lowres )
            {
                for( int d0 = 0; d0 <= h->param.i_bframe+1; d0++ )
                    for( int d1 = 0; d1 <= h->param.i_bframe+1; d1++ )
                        if( f->i_cost_est[d0][d1] >= 0 && f->lowres_costs[d0][d1] )
                            for( int y = do_edges ? 0 : 1; y < mbh - !do_edges; y++ )
                                o[0] = rh_crc32( o[0], f->lowres_costs[d0][d1] + y*mbw + !do_edges, ( mbw - 2*!do_edges ) * sizeof(uint16_t) );
                for( int l = 0; l <= !!h->param.i_bframe; l++ )
                    for( int d = 0; d <= h->param.i_bframe; d++ )
                        if( f->lowres_mvs[l][d] && f->lowres_mvs[l][d][0][0] != 0x7FFF )
                        {
                            o[1] = rh_crc32( o[1], f->lowres_mvs[l][d], n_mb * 2 * sizeof(int16_t) );
                            for( int y = do_edges ? 0 : 1; y < mbh - !do_edges; y++ )
                                o[2] = rh_crc32( o[2], f->lowres_mv_costs[l][d] + y*mbw + !do_edges, ( mbw - 2*!do_edges ) * sizeof(int) );
                        }
            }
            if( f->f_qp_offset )
                o[3] = rh_crc32( 0, f->f_qp_offset, n_mb * sizeof(float) );
            /* (row sums are kept only for VBV, slicetype.c:970-980 -- without it the C path never writes them and nothing reads them) */
            if( f->i_row_satd && h->param.rc.i_rc_method != X264_RC_CQP && h->param.rc.i_vbv_buffer_size )
                o[4] = rh_crc32( 0, f->i_row_satd, mbh * sizeof(int) );
            if( h->param.rc.i_vbv_buffer_size && h->param.rc.i_lookahead )
            {
                o[5] = rh_crc32( 0, f->i_planned_satd, ( h->param.rc.i_lookahead + 1 ) * sizeof(int) );
                o[6] = rh_crc32( 0, f->i_planned_type, ( h->param.rc.i_lookahead + 1 ) * sizeof(uint8_t) );
            }
            /* ([0] is written by an intra-only evaluation in the C path and by nobody through the hook; nothing reads it) */
            o[7] = rh_crc32( 0, f->i_intra_mbs + 1, ( h->param.i_bframe + 1 ) * sizeof(int) );
        }
        n_out++;
    }
    clock_gettime( CLOCK_MONOTONIC, &t1 );
    if( seconds ) *seconds = (t1.tv_sec-t0.tv_sec) + 1e-9*(t1.tv_nsec-t0.tv_nsec);
    if( stream_crc ) *stream_crc = crc_stream;
    free( grey );
    return n_out;
}

/* ------------------------------------------------------------------------------------------
 * Primitive access (checkasm-style known answers): the C vtables filled by the reference.
 * ------------------------------------------------------------------------------------------ */
/* kind: 0 sad, 1 satd, 2 ssd, 3 sa8d (size 0..3) */
RH_API int rh_pixel_cmp( rh_ctx *c, int kind, int size, pixel *a, intptr_t sa, pixel *b, intptr_t sb )
{
    x264_pixel_function_t *pf = &c->h->pixf;
    switch( kind )
    {
        case 0: return pf->sad[size]( a, sa, b, sb );
        case 1: return pf->satd[size]( a, sa, b, sb );
        case 2: return pf->ssd[size]( a, sa, b, sb );
        case 3: return pf->sa8d[size]( a, sa, b, sb );
    }
    return -1;
}
/* fenc has FENC_STRIDE; n = 3 or 4; satd != 0 selects satd_x3/x4 */
RH_API void rh_pixel_cmp_xn( rh_ctx *c, int satd, int n, int size, pixel *fenc, pixel *ref, const int *offs, intptr_t stride, int *out )
{
    x264_pixel_function_t *pf = &c->h->pixf;
    if( n == 3 )
        (satd ? pf->satd_x3 : pf->sad_x3)[size]( fenc, ref+offs[0], ref+offs[1], ref+offs[2], stride, out );
    else
        (satd ? pf->satd_x4 : pf->sad_x4)[size]( fenc, ref+offs[0], ref+offs[1], ref+offs[2], ref+offs[3], stride, out );
}
RH_API int rh_var2( rh_ctx *c, int is8x16, pixel *fenc, pixel *fdec, int *ssd ) { return c->h->pixf.var2[is8x16 ? PIXEL_8x16 : PIXEL_8x8]( fenc, fdec, ssd ); }
RH_API uint64_t rh_hadamard_ac( rh_ctx *c, int size, pixel *pix, intptr_t stride ) { return c->h->pixf.hadamard_ac[size]( pix, stride ); }
RH_API int rh_vsad( rh_ctx *c, pixel *src, intptr_t stride, int height ) { return c->h->pixf.vsad( src, stride, height ); }
RH_API int rh_asd8( rh_ctx *c, pixel *a, intptr_t sa, pixel *b, intptr_t sb, int height ) { return c->h->pixf.asd8( a, sa, b, sb, height ); }
RH_API uint64_t rh_pixel_var( rh_ctx *c, int size, pixel *a, intptr_t sa ) { return c->h->pixf.var[size]( a, sa ); }
/* fdec points at the top-left pixel of an FDEC_STRIDE buffer with neighbours filled in */
RH_API void rh_intra_x3_8x8c( rh_ctx *c, int satd, pixel *fenc, pixel *fdec, int *res )
{
    (satd ? c->h->pixf.intra_satd_x3_8x8c : c->h->pixf.intra_sad_x3_8x8c)( fenc, fdec, res );
}
RH_API void rh_predict_8x8c( rh_ctx *c, int mode, pixel *src ) { c->h->predict_8x8c[mode]( src ); }
RH_API void rh_predict_8x8_filter( rh_ctx *c, pixel *src, pixel *edge, int neighbor, int filters ) { c->h->predict_8x8_filter( src, edge, neighbor, filters ); }
RH_API void rh_predict_8x8( rh_ctx *c, int mode, pixel *src, pixel *edge ) { c->h->predict_8x8[mode]( src, edge ); }
RH_API void rh_lowres_core( rh_ctx *c, pixel *src, pixel *d0, pixel *dh, pixel *dv, pixel *dc, intptr_t ss, intptr_t ds, int w, int hh )
{
    c->h->mc.frame_init_lowres_core( src, d0, dh, dv, dc, ss, ds, w, hh );
}
RH_API void rh_integral_init( rh_ctx *c, int kind, uint16_t *sum8, uint16_t *sum4, pixel *pix, intptr_t stride )
{
    x264_t *h = c->h;
    if( kind == 0 ) h->mc.integral_init4h( sum8, pix, stride );
    else if( kind == 1 ) h->mc.integral_init8h( sum8, pix, stride );
    else if( kind == 2 ) h->mc.integral_init4v( sum8, sum4, stride );
    else h->mc.integral_init8v( sum8, stride );
}
/* size_idx: PIXEL_16x16 (ads4), PIXEL_16x8 (ads2), PIXEL_8x8 (ads1) */
RH_API int rh_ads( rh_ctx *c, int size_idx, int *enc_dc, uint16_t *sums, int delta, uint16_t *cost_mvx, int16_t *mvs, int width, int thresh )
{
    return c->h->pixf.ads[size_idx]( enc_dc, sums, delta, cost_mvx, mvs, width, thresh );
}
RH_API void rh_hpel_filter( rh_ctx *c, pixel *dsth, pixel *dstv, pixel *dstc, pixel *src, intptr_t stride, int w, int hh, int16_t *buf )
{
    c->h->mc.hpel_filter( dsth, dstv, dstc, src, stride, w, hh, buf );
}
/* planes: 4 pointers to the same-geometry hpel planes; returns via dst (always materialised) */
RH_API void rh_mc_luma( rh_ctx *c, pixel *dst, intptr_t ds, pixel *p0, pixel *p1, pixel *p2, pixel *p3, intptr_t ss,
                        int mvx, int mvy, int w, int hh, int wt_on, int scale, int denom, int offset )
{
    x264_t *h = c->h;
    pixel *src[4] = { p0, p1, p2, p3 };
    x264_weight_t wt[3];
    memset( wt, 0, sizeof(wt) );
    SET_WEIGHT( wt[0], wt_on, scale, denom, offset );
    h->mc.mc_luma( dst, ds, src, ss, mvx, mvy, w, hh, wt_on ? wt : x264_weight_none );
}
RH_API void rh_avg( rh_ctx *c, int size, pixel *dst, intptr_t ds, pixel *a, intptr_t sa, pixel *b, intptr_t sb, int weight )
{
    c->h->mc.avg[size]( dst, ds, a, sa, b, sb, weight );
}
/* kind: 0 sub4x4_dct 1 sub8x8_dct 2 sub16x16_dct 3 sub8x8_dct8 4 sub16x16_dct8 5 sub8x8_dct_dc 6 sub8x16_dct_dc
 *       7 dct4x4dc (in place) 8 dct2x4dc (in place uses out as dct4x4 array [8][16]) */
RH_API void rh_dct( rh_ctx *c, int kind, dctcoef *out, pixel *fenc, pixel *fdec )
{
    x264_dct_function_t *d = &c->h->dctf;
    switch( kind )
    {
        case 0: d->sub4x4_dct( out, fenc, fdec ); break;
        case 1: d->sub8x8_dct( (void*)out, fenc, fdec ); break;
        case 2: d->sub16x16_dct( (void*)out, fenc, fdec ); break;
        case 3: d->sub8x8_dct8( out, fenc, fdec ); break;
        case 4: d->sub16x16_dct8( (void*)out, fenc, fdec ); break;
        case 5: d->sub8x8_dct_dc( out, fenc, fdec ); break;
        case 6: d->sub8x16_dct_dc( out, fenc, fdec ); break;
        case 7: d->dct4x4dc( out ); break;
        case 8: /* dct2x4dc( dct[8], dct4x4[8][16] ): `out` holds the eight DC values; results come back in place */
        {
            dctcoef blocks[8][16], dc[8];
            memset( blocks, 0, sizeof(blocks) );
            for( int i = 0; i < 8; i++ ) blocks[i][0] = out[i];
            d->dct2x4dc( dc, blocks );
            for( int i = 0; i < 8; i++ ) out[i] = dc[i];
            break;
        }
    }
}
/* kind: 0 quant_4x4 1 quant_8x8 2 quant_4x4x4 3 quant_4x4_dc 4 quant_2x2_dc ; cqm list i_list, qp */
RH_API int rh_quant( rh_ctx *c, int kind, dctcoef *coef, int i_list, int qp, int b_intra_bias )
{
    x264_t *h = c->h;
    x264_quant_function_t *q = &h->quantf;
    switch( kind )
    {
        case 0: return q->quant_4x4( coef, h->quant4_mf[i_list][qp], h->quant4_bias[i_list][qp] );
        case 1: return q->quant_8x8( coef, h->quant8_mf[i_list][qp], h->quant8_bias[i_list][qp] );
        case 2: return q->quant_4x4x4( (void*)coef, h->quant4_mf[i_list][qp], h->quant4_bias[i_list][qp] );
        case 3: return q->quant_4x4_dc( coef, h->quant4_mf[i_list][qp][0]>>1, h->quant4_bias[i_list][qp][0]<<1 );
        case 4: return q->quant_2x2_dc( coef, h->quant4_mf[i_list][qp][0]>>1, h->quant4_bias[i_list][qp][0]<<1 );
    }
    return -1;
}
RH_API void rh_quant_tables( rh_ctx *c, int is8, int i_list, int qp, udctcoef *mf, udctcoef *bias )
{
    x264_t *h = c->h;
    int n = is8 ? 64 : 16;
    memcpy( mf,   is8 ? h->quant8_mf[i_list][qp]   : h->quant4_mf[i_list][qp],   n*sizeof(udctcoef) );
    memcpy( bias, is8 ? h->quant8_bias[i_list][qp] : h->quant4_bias[i_list][qp], n*sizeof(udctcoef) );
}
RH_API void rh_luts( float *log2_lut, float *log2_lz_lut, uint8_t *exp2_lut )
{
    memcpy( log2_lut, x264_log2_lut, 128*sizeof(float) );
    memcpy( log2_lz_lut, x264_log2_lz_lut, 32*sizeof(float) );
    memcpy( exp2_lut, x264_exp2_lut, 64 );
}
RH_API void rh_get_weight( rh_ctx *c, int idx, int *out )
{
    x264_weight_t *w = c->frames[idx]->weight[0];
    out[0] = w[0].weightfn != NULL; out[1] = w[0].i_scale; out[2] = w[0].i_denom; out[3] = w[0].i_offset;
}
