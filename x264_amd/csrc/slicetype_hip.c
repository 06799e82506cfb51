/* slicetype_hip.c -- libx264hip.so behind the reference encoder's own accelerator seam.
 *
 * This is the reference-side binding, C, compiled INTO jpsdr/x264 in a HAVE_OPENCL build in place of encoder/slicetype-cl.c and
 * common/opencl.c (the two files the reference Makefile adds for that build, Makefile:254).  It defines exactly the symbols those two
 * files define and the unmodified reference calls:
 *     x264_opencl_load_library / _close_library     common/opencl.h:796-799   <- encoder/encoder.c:1744-1753, 4575
 *     x264_opencl_lookahead_init / _delete           common/opencl.h:801-804   <- encoder/encoder.c:1797-1799, 4208
 *     x264_opencl_frame_delete                       common/opencl.h:806-807   <- common/frame.c:335-336
 *     x264_opencl_lowres_init, _motionsearch, _finalize_cost, _flush, _slicetype_prep, _slicetype_end
 *                                                    encoder/slicetype-cl.h:29-42 <- encoder/slicetype.c:878-897, 1531, 1741
 * so that with --opencl (param.b_opencl) slicetype_frame_cost's accelerator branch lands on the HIP library while the memo test, the
 * first-trigger flags, x264_weights_analyse, scene cuts, the B-frame decision (slicetype_path), MB-tree, VBV planning and the main encode
 * are the reference's own code, untouched.  Results are written into the reference's own arrays, as slicetype-cl.c does at its flush
 * (slicetype-cl.c:58-70,254-282,513-535,613-649): lowres_costs[0][0] (i_intra_cost), i_row_satds, i_cost_est(_aq), i_intra_mbs,
 * lowres_mvs / lowres_mv_costs of every field searched, lowres_costs of the cell.  Unlike the OpenCL kernels -- a different, hierarchical
 * search -- the HIP path reproduces slicetype_mb_cost bit for bit, so the encoder's output is identical with the hook on and off
 * (tests/test_reference_seam.py, tests/test_gpu_reference_seam.py: slice types, every cost cell, the maps, the bitstream).
 *
 * The library is opened with dlopen (name: $X264HIP_LIB, else libx264hip.so), like common/opencl.c:53-61 opens libOpenCL: the encoder
 * builds and runs without ROCm, and falls back to its C path when the library or a device is missing (x264_opencl_load_library
 * returns NULL -> encoder.c:1748-1752).  A failing call latches h->opencl.b_fatal_error and x264_encoder_encode returns -1
 * (encoder.c:3332-3335), the error behaviour of slicetype-cl.c:44-56.
 *
 * Differences to slicetype-cl.c that keep the C path's results:
 *  - x264_opencl_slicetype_prep only brings the window's frames to the device (and, optionally, lets the library speculate: never
 *    changes results).  The OpenCL version also pre-searches every pair for the trellis with weights analysed in ITS order
 *    (slicetype-cl.c:693-735), which changes which searches are weighted; here every search is triggered by slicetype_frame_cost.
 *  - the two searches and the cell of one slicetype_frame_cost call are ONE library call (x264hip_frame_cost), issued from
 *    x264_opencl_finalize_cost; x264_opencl_motionsearch only records what the call asked for.
 *
 * State lives where the OpenCL path keeps its own: the function table x264_opencl_load_library returns is the head of our per-encoder
 * record (h->opencl.ocl, copied into every frame at frame.c:301-302), a frame's device slot sits in x264_frame_opencl_t.intra_cost. */
#include "common/common.h"
#include "encoder/slicetype-cl.h"

#if HAVE_OPENCL
#include <dlfcn.h>
#include "x264hip.h"

typedef struct
{
    x264_opencl_function_t ocl; /* must be first: h->opencl.ocl / frame->opencl.ocl point here */
    void *lib;
    x264hip_ctx *ctx;
    int n_slots;
    int *free_slots, n_free;
    int speculate;              /* $X264HIP_SEAM_PREFETCH: x264_opencl_slicetype_prep also hands the window to x264hip_prefetch */
    /* the request slicetype_frame_cost is making: filled by x264_opencl_motionsearch, consumed by x264_opencl_finalize_cost */
    int do_search[2];
    x264hip_weight w;
    int16_t *mvs; int *mv_costs; uint16_t *costs; int *rows;
    int ( *open )( x264hip_ctx **, int, const x264hip_params * );
    void ( *close )( x264hip_ctx * );
    const char *( *strerror )( int );
    int ( *frame_put )( x264hip_ctx *, int, const void *, int, int, const void *, const void *, int, const uint16_t * );
    int ( *frame_cost )( x264hip_ctx *, int, int, int, int, int, const int[2], const x264hip_weight *, int, int, x264hip_cost * );
    int ( *get_mvs )( x264hip_ctx *, int, int, int, int16_t *, int * );
    int ( *get_lowres_costs )( x264hip_ctx *, int, int, int, uint16_t *, int * );
    int ( *get_intra_costs )( x264hip_ctx *, int, uint16_t * );
    int ( *prefetch )( x264hip_ctx *, const int *, const int *, int );
} hip_seam_t;

#define SEAM( h ) ( (hip_seam_t *)( h )->opencl.ocl )
#define SLOT_OF( f ) ( (int)(intptr_t)( f )->opencl.intra_cost - 1 ) /* -1: the frame has no slot yet */

static int hip_fail( x264_t *h, const char *what, int rc )
{
    /* slicetype-cl.c:44-56 */
    hip_seam_t *s = SEAM( h );
    h->param.b_opencl = 0;
    h->opencl.b_fatal_error = 1;
    x264_log( h, X264_LOG_ERROR, "x264hip: %s failed: %s\n", what, s && s->strerror ? s->strerror( rc ) : "?" );
    return -1;
}
#define HIPCHECK( call, what ) do { if( h->opencl.b_fatal_error ) return -1; int rc_ = ( call ); if( rc_ ) return hip_fail( h, what, rc_ ); } while( 0 )

x264_opencl_function_t *x264_opencl_load_library( void )
{
    hip_seam_t *s = calloc( 1, sizeof( *s ) );
    if( !s )
        return NULL;
    const char *name = getenv( "X264HIP_LIB" );
    s->lib = dlopen( name && name[0] ? name : "libx264hip.so", RTLD_NOW | RTLD_LOCAL );
    if( !s->lib )
        goto fail;
#define LOAD( field, sym ) if( !( *(void **)&s->field = dlsym( s->lib, sym ) ) ) goto fail
    LOAD( open, "x264hip_open" ); LOAD( close, "x264hip_close" ); LOAD( strerror, "x264hip_strerror" );
    LOAD( frame_put, "x264hip_frame_put" ); LOAD( frame_cost, "x264hip_frame_cost" ); LOAD( get_mvs, "x264hip_get_mvs" );
    LOAD( get_lowres_costs, "x264hip_get_lowres_costs" ); LOAD( get_intra_costs, "x264hip_get_intra_costs" );
#undef LOAD
    *(void **)&s->prefetch = dlsym( s->lib, "x264hip_prefetch" ); /* optional */
    return &s->ocl;
fail:
    if( s->lib )
        dlclose( s->lib );
    free( s );
    return NULL;
}

void x264_opencl_close_library( x264_opencl_function_t *ocl )
{
    hip_seam_t *s = (hip_seam_t *)ocl;
    if( !s )
        return;
    /* (every frame has been deleted by now, encoder.c:4575 comes last) */
    if( s->ctx )
        s->close( s->ctx );
    free( s->free_slots ); free( s->mvs ); free( s->mv_costs ); free( s->costs ); free( s->rows );
    dlclose( s->lib );
    free( s );
}

int x264_opencl_lookahead_init( x264_t *h )
{
    hip_seam_t *s = SEAM( h );
    x264hip_params p;
    memset( &p, 0, sizeof( p ) );
    p.bit_depth = BIT_DEPTH;
    p.width = h->param.i_width; p.height = h->param.i_height;
    p.bframes = h->param.i_bframe;
    p.lambda = x264_lambda_tab[X264_LOOKAHEAD_QP];
    /* lowres_context_init (slicetype.c:45-61) */
    p.me_method = h->param.analyse.i_subpel_refine > 1 ? X264_MIN( X264_ME_HEX, h->param.analyse.i_me_method ) : X264_ME_DIA;
    p.subpel_refine = h->param.analyse.i_subpel_refine > 1 ? 4 : 2;
    p.me_range = h->param.analyse.i_me_range; p.mv_range = h->param.analyse.i_mv_range; p.subme = h->param.analyse.i_subpel_refine;
    p.mbcmp_satd = h->pixf.mbcmp[0] == h->pixf.satd[0];     /* mbcmp_init, encoder.c:1409-1427 */
    p.fpelcmp_satd = h->pixf.fpelcmp[0] == h->pixf.satd[0];
    p.weighted_bipred = h->param.analyse.b_weighted_bipred;
    p.aq_mode = h->param.rc.i_aq_mode; p.aq_strength = h->param.rc.f_aq_strength; /* (the encoder's own x264_adaptive_quant_frame stays authoritative: its factors go in with every frame) */
    p.bframe_bias = h->param.i_bframe_bias;
    p.no_edges = !( h->param.rc.b_mb_tree || h->param.rc.i_vbv_buffer_size );      /* do_edges, slicetype.c:823 */
    p.lookahead_slices = h->param.i_lookahead_threads;                               /* the bands of slicetype.c:917-918 */
    p.chroma_format = 1;
    p.cost_mv = h->cost_mv[X264_LOOKAHEAD_QP];                                        /* analyse.c:151-157,194 */
    /* every x264_frame_t that can pass through the lookahead keeps its slot for life: frames.unused[0] holds i_delay + 3 of them
     * (encoder.c:1635), plus the ones in flight in the encoder threads */
    p.max_frames = s->n_slots = h->frames.i_delay + h->param.i_bframe + h->param.i_threads + 8;
    s->free_slots = malloc( s->n_slots * sizeof( int ) );
    s->mvs = malloc( h->mb.i_mb_count * 2 * sizeof( int16_t ) ); s->mv_costs = malloc( h->mb.i_mb_count * sizeof( int ) );
    s->costs = malloc( h->mb.i_mb_count * sizeof( uint16_t ) ); s->rows = malloc( h->mb.i_mb_height * sizeof( int ) );
    if( !s->free_slots || !s->mvs || !s->mv_costs || !s->costs || !s->rows )
        return -1;
    for( int i = 0; i < s->n_slots; i++ )
        s->free_slots[i] = s->n_slots - 1 - i;
    s->n_free = s->n_slots;
    const char *dev = getenv( "X264HIP_DEVICE" ), *spec = getenv( "X264HIP_SEAM_PREFETCH" );
    s->speculate = spec && atoi( spec ) && s->prefetch;
    int rc = s->open( &s->ctx, h->param.i_opencl_device ? h->param.i_opencl_device : dev ? atoi( dev ) : 0, &p );
    if( rc )
    {
        x264_log( h, X264_LOG_WARNING, "x264hip: %s, using the C lookahead\n", s->strerror( rc ) );
        s->ctx = NULL;
        return -1;
    }
    x264_log( h, X264_LOG_INFO, "x264hip: lookahead on the HIP device (%d frame slots)\n", s->n_slots );
    return 0;
}

void x264_opencl_lookahead_delete( x264_t *h )
{
    /* the context goes with the library (x264_opencl_close_library): frames deleted after this call still give their slots back */
    (void)h;
}

void x264_opencl_frame_delete( x264_frame *frame )
{
    hip_seam_t *s = (hip_seam_t *)frame->opencl.ocl;
    int slot = SLOT_OF( frame );
    if( s && slot >= 0 && s->n_free < s->n_slots )
        s->free_slots[s->n_free++] = slot;
    frame->opencl.intra_cost = NULL;
}

/* x264_opencl_lowres_init (slicetype-cl.c:82-282): the frame's picture to the device (lowres planes are made there), its intra costs,
 * their row sums and frame sums back into the reference's arrays.  Once per picture: b_intra_calculated is the flag, as in the
 * OpenCL path (:84-86). */
int x264_opencl_lowres_init( x264_t *h, x264_frame_t *fenc, int lambda )
{
    if( fenc->b_intra_calculated )
        return 0;
    fenc->b_intra_calculated = 1;
    hip_seam_t *s = SEAM( h );
    int slot = SLOT_OF( fenc );
    if( slot < 0 )
    {
        if( !s->n_free )
            return hip_fail( h, "frame slot allocation", X264HIP_ENOMEM );
        slot = s->free_slots[--s->n_free];
        fenc->opencl.intra_cost = (cl_mem)(intptr_t)( slot + 1 );
    }
    /* plane[0] with its stride is the mod-16 padded picture x264_frame_init_lowres reads (mc.c:458-482); the quantiser factors are the
     * ones x264_adaptive_quant_frame made (ratecontrol.c:304-415), 256 everywhere without AQ (slicetype-cl.c:187-199) */
    HIPCHECK( s->frame_put( s->ctx, slot, fenc->plane[0], fenc->i_stride[0], 0, NULL, NULL, 0,
                            h->param.rc.i_aq_mode && fenc->i_inv_qscale_factor ? fenc->i_inv_qscale_factor : NULL ), "x264hip_frame_put" );
    static const int none[2] = { 0, 0 };
    x264hip_cost c;
    HIPCHECK( s->frame_cost( s->ctx, slot, slot, slot, 0, 0, none, NULL, 1, 0, &c ), "x264hip_frame_cost (intra)" );
    HIPCHECK( s->get_lowres_costs( s->ctx, slot, 0, 0, fenc->lowres_costs[0][0], fenc->i_row_satds[0][0] ), "x264hip_get_lowres_costs (intra)" );
    fenc->i_cost_est[0][0] = c.intra_cost_est;
    fenc->i_cost_est_aq[0][0] = c.intra_cost_est_aq;
    (void)lambda;
    return 0;
}

/* x264_opencl_motionsearch (slicetype-cl.c:350-535): slicetype_frame_cost calls it once per list whose first-trigger flag it has just
 * cleared, always followed by x264_opencl_finalize_cost for the same (p0, p1, b): note the request, the search runs there. */
int x264_opencl_motionsearch( x264_t *h, x264_frame_t **frames, int b, int ref, int b_islist1, int lambda, const x264_weight_t *w )
{
    hip_seam_t *s = SEAM( h );
    if( h->opencl.b_fatal_error )
        return -1;
    s->do_search[!!b_islist1] = 1;
    if( !b_islist1 )
    {
        s->w.on = w && w->weightfn;
        if( s->w.on )
        {
            s->w.scale = w->i_scale; s->w.denom = w->i_denom; s->w.offset = w->i_offset;
        }
    }
    (void)frames; (void)b; (void)ref; (void)lambda;
    return 0;
}

/* x264_opencl_finalize_cost (slicetype-cl.c:537-651) + the searches noted above: one slicetype_frame_cost evaluation (slicetype.c:899-989)
 * on the device, then everything the C path leaves in the frame. */
int x264_opencl_finalize_cost( x264_t *h, int lambda, x264_frame_t **frames, int p0, int p1, int b, int dist_scale_factor )
{
    hip_seam_t *s = SEAM( h );
    x264_frame_t *fenc = frames[b];
    int do_search[2] = { s->do_search[0], s->do_search[1] };
    x264hip_weight w = s->w;
    s->do_search[0] = s->do_search[1] = 0;
    s->w.on = 0;
    if( h->opencl.b_fatal_error )
        return -1;
    int sb = SLOT_OF( fenc ), s0 = SLOT_OF( frames[p0] ), s1 = SLOT_OF( frames[p1] );
    if( sb < 0 || s0 < 0 || s1 < 0 )
        return hip_fail( h, "x264_opencl_finalize_cost (a frame was never initialised)", X264HIP_ESTATE );
    /* slicetype.c:629: the list-1 reference's own list-0 vectors are used when that field has been searched */
    int ref1_l0 = b < p1 && frames[p1]->lowres_mvs[0][p1-p0-1][0][0] != 0x7FFF;
    x264hip_cost c;
    HIPCHECK( s->frame_cost( s->ctx, s0, s1, sb, b-p0, p1-b, do_search, do_search[0] && w.on ? &w : NULL, 0, ref1_l0, &c ), "x264hip_frame_cost" );
    for( int l = 0; l < 2; l++ )
        if( do_search[l] )
        {
            int d = l ? p1-b-1 : b-p0-1;
            HIPCHECK( s->get_mvs( s->ctx, sb, l, d, s->mvs, fenc->lowres_mv_costs[l][d] ), "x264hip_get_mvs" );
            memcpy( fenc->lowres_mvs[l][d], s->mvs, h->mb.i_mb_count * 2 * sizeof( int16_t ) );
        }
    HIPCHECK( s->get_lowres_costs( s->ctx, sb, b-p0, p1-b, fenc->lowres_costs[b-p0][p1-b], fenc->i_row_satds[b-p0][p1-b] ), "x264hip_get_lowres_costs" );
    /* slicetype.c:946-989: the B-frame score is stored scaled, the AQ sum is not */
    fenc->i_cost_est[b-p0][p1-b] = b != p1 ? (int)( (uint64_t)c.cost_est * 100 / ( 120 + h->param.i_bframe_bias ) ) : c.cost_est;
    fenc->i_cost_est_aq[b-p0][p1-b] = c.cost_est_aq;
    if( b == p1 )
        fenc->i_intra_mbs[b-p0] = c.intra_mbs;
    (void)lambda; (void)dist_scale_factor;
    return 0;
}

void x264_opencl_flush( x264_t *h )
{
    /* the getters above have already waited for the device and written the reference's arrays: nothing is deferred */
    (void)h;
}

void x264_opencl_slicetype_prep( x264_t *h, x264_frame_t **frames, int num_frames, int lambda )
{
    if( !h->param.b_opencl )
        return;
    hip_seam_t *s = SEAM( h );
    /* slicetype-cl.c:683-686: the window's frames to the device, their intra costs into the frames */
    for( int i = 0; i <= num_frames; i++ )
        if( x264_opencl_lowres_init( h, frames[i], lambda ) < 0 )
            return;
    if( s->speculate && num_frames > 0 )
    {
        /* let the library search ahead of the decisions (x264hip_prefetch: never changes a result) */
        int slots[X264_LOOKAHEAD_MAX+4], numbers[X264_LOOKAHEAD_MAX+4];
        int n = 0;
        for( int i = 0; i <= num_frames && n < X264_LOOKAHEAD_MAX+4; i++, n++ )
        {
            slots[n] = SLOT_OF( frames[i] );
            numbers[n] = frames[i]->i_frame;
        }
        int rc = s->prefetch( s->ctx, slots, numbers, n );
        if( rc )
            hip_fail( h, "x264hip_prefetch", rc );
    }
}

void x264_opencl_slicetype_end( x264_t *h )
{
    (void)h;
}

#endif /* HAVE_OPENCL */
